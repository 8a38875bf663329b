"""8192 x 8192 x 51200 fp16 Linear (gemm_f16_dma.hip) timing; L2Q_LIB_NAME selects an A/B build"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'l2hmc-qcd_amd'))
from l2hmc import _ops as ops
torch.manual_seed(0)
m, n, k = 8192, 8192, 51200
a = torch.randn(m, k, device='cuda').half(); w = (torch.randn(n, k, device='cuda') / k ** 0.5).half()
b = torch.randn(n, device='cuda')
for _ in range(2):
    c = ops.gemm_h(a, w, b, act='leaky_relu', out_dtype=torch.float32)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    c = ops.gemm_h(a, w, b, act='leaky_relu', out_dtype=torch.float32)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
ref = (a[:64].float() @ w[:256].float().t() + b[:256])
ref = torch.where(ref > 0, ref, 0.01 * ref)
print(f'{os.environ.get("L2Q_LIB_NAME", "libl2q.so")}: {ms:.3f} ms  {2.0 * m * n * k / ms / 1e12:.1f} TFLOP/s  max|err| {float((c[:64, :256] - ref).abs().max()):.2e}', flush=True)
