#!/usr/bin/env bash
# The hipBLASLt route of the big plain 16-bit layers (gemm_lt.hip, tuning gemm_h_lt) against the own LDS-DMA kernel:
# the conv stack's Linear alone, then cfg-3 with the reference's default conv network (one box).
cd "$(dirname "$0")/../.."
python - <<'PY'
import sys, torch
sys.path.insert(0, 'l2hmc-qcd_amd')
from l2hmc import _ops as ops, native
M, N, K = 8192, 8192, 51200
a = torch.randn(M, K, device='cuda', dtype=torch.float16)
w = (torch.randn(N, K, device='cuda') / K ** 0.5).to(torch.float16)
b = torch.zeros(N, device='cuda')
for lt in (1, 0, 1, 0):
    native.set_tuning('gemm_h_lt', lt)
    f = lambda: ops.gemm_h(a, w, b, act='leaky_relu', out_dtype=torch.float32)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f'gemm_h 8192 x 8192 x 51200 fp16 (+bias, leaky_relu, fp32 container) gemm_h_lt={lt}: {ms:.3f} ms  {2 * M * N * K / ms / 1e9:.0f} TFLOP/s', flush=True)
PY
for lt in 1 0; do echo "## cfg-3 default conv network, fp16, gemm_h_lt=$lt"; timeout 900 python tools/bench_u1.py --L 64 64 --nb 8192 --beta 6 --steps 2 --no-hmc --no-graph --conv --precision fp16 --tune gemm_h_lt $lt 2>&1 | grep -E "forward|hmc"; done
