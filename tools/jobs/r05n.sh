#!/usr/bin/env bash
cd "$(dirname "$0")/../.."
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_sizes_gpu.py tests/test_dynamics_gpu.py -q -x -k "gemm_h or cfg3 or half or fp16 or bf16" 2>&1 | tail -3
tools/kstats.sh gpurun_out/r05n/stats_red.txt python tools/time_gemm_h_input.py 1
grep -i "skinny\|splitk\|pack_w" gpurun_out/r05n/stats_red.txt | awk '{print substr($1,1,60), $2,$3,$4}'
timeout 600 python tools/bench_u1_block.py cfg3_dense256_fp16 2>&1 | grep "^cfg3" | cut -c1-200
