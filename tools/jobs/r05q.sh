#!/usr/bin/env bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05q
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_sizes_gpu.py tests/test_dynamics_gpu.py -q -x -k "gemm_h or cfg3 or half or fp16 or bf16" 2>&1 | tail -4
python - <<'PY'
import sys, time, torch
sys.path.insert(0, 'l2hmc-qcd_amd')
from l2hmc import _ops as ops, native
m, n, k = 8192, 256, 256
a = torch.randn(m, k, device='cuda').half(); w = (torch.randn(n, k, device='cuda') / 16).half(); b = torch.zeros(n, device='cuda')
for v in (0, 1, 0, 1):
    native.set_tuning('gemm_h_small', v)
    for _ in range(3): ops.gemm_h(a, w, b, act='leaky_relu')
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): ops.gemm_h(a, w, b, act='leaky_relu')
    torch.cuda.synchronize(); print(f'[gemm_h_small={v}] hidden layer 8192x256x256: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us')
PY
timeout 600 python tools/bench_u1_block.py cfg3_dense256_fp16 2>&1 | grep "^cfg3" | tee gpurun_out/r05q/bench_cfg3.txt
