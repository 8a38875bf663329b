#!/usr/bin/env bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05n
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_sizes_gpu.py -q -x -k "gemm_h or cfg3" 2>&1 | tail -4
tools/kstats.sh gpurun_out/r05n/stats_u1x.txt python tools/time_gemm_h_input.py 1
grep -i "skinny\|splitk\|pack_w" gpurun_out/r05n/stats_u1x.txt | awk '{print substr($1,1,60), $2,$3,$4}'
timeout 300 python tools/time_gemm_h_input.py 0 1 2>&1 | grep "^\["
