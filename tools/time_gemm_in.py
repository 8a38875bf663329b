"""cfg-4 input layer (256 x 256 x 2 x 131072 fp64, split-K) timing; L2Q_LIB_NAME selects an A/B build"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'l2hmc-qcd_amd'))
from l2hmc import _ops as ops
torch.manual_seed(0)
nb, h, K = 256, 256, 131072
xv = torch.randn(nb, K, dtype=torch.float64, device='cuda'); fv = torch.randn(nb, K, dtype=torch.float64, device='cuda')
wx = torch.randn(h, K, dtype=torch.float64, device='cuda') / K ** 0.5; wv = torch.randn(h, K, dtype=torch.float64, device='cuda') / K ** 0.5
bx = torch.randn(h, dtype=torch.float64, device='cuda')
for _ in range(3):
    c = ops.gemm(xv, wx, bx, a2=fv, w2=wv, bias2=bx, act='tanh')
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    c = ops.gemm(xv, wx, bx, a2=fv, w2=wv, bias2=bx, act='tanh')
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
ref = torch.tanh(xv[:8] @ wx.t() + fv[:8] @ wv.t() + 2 * bx)
print(f'{os.environ.get("L2Q_LIB_NAME", "libl2q.so")}: {ms:.4f} ms  {2.0 * nb * h * 2 * K / ms / 1e9:.1f} TFLOP/s  max|err| {float((c[:8] - ref).abs().max()):.2e}', flush=True)
