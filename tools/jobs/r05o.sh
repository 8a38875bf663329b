#!/usr/bin/env bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05o
timeout 1500 python -m pytest tests/test_sizes_gpu.py tests/test_dynamics_gpu.py tests/test_kernels_gpu.py tests/test_train_gpu.py -q -x -k "cfg3 or half or bf16 or fp16 or heads_update_h or autocast or gemm_h" 2>&1 | tail -5
timeout 600 python tools/bench_u1_block.py cfg3_dense256_fp16 2>&1 | grep "^cfg3" | tee gpurun_out/r05o/bench_cfg3.txt
