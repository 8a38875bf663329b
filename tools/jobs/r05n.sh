#!/usr/bin/env bash
# skinny input-layer kernel in the product: U(1) half-precision parity tests + cfg-3 bench block
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05n
timeout 1500 python -m pytest tests/test_sizes_gpu.py tests/test_dynamics_gpu.py tests/test_kernels_gpu.py -q -x -k "cfg3 or half or bf16 or fp16 or gemm_h or autocast" 2>&1 | tail -5
timeout 600 python tools/bench_u1_block.py cfg3_dense256_fp16 2>&1 | grep "^cfg3" | tee gpurun_out/r05n/bench_cfg3.txt
