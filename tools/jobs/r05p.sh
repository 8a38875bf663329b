#!/usr/bin/env bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05p
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_sizes_gpu.py -q -x -k "u1_force or u1_ops or cfg3 or cfg2" 2>&1 | tail -4
timeout 600 python tools/bench_u1_block.py cfg3_dense256_fp16 2>&1 | grep "^cfg3" | tee gpurun_out/r05p/bench_cfg3.txt
