"""cfg-4 input layer (M 256, N 256, K 131072 + 131072): fp64 MFMA layer vs the int8-sliced one"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'l2hmc-qcd_amd'))
from l2hmc import _ops as ops
m, n = int(os.environ.get('M', 256)), int(os.environ.get('NN', 256))
k = int(os.environ.get('K', 131072))
g = torch.Generator(device='cuda').manual_seed(1)
a = (torch.rand(m, k, dtype=torch.float64, device='cuda', generator=g) - 0.5) * 4.0
a2 = (torch.rand(m, k, dtype=torch.float64, device='cuda', generator=g) - 0.5) * 4.0
w = (torch.rand(n, k, dtype=torch.float64, device='cuda', generator=g) - 0.5) / k ** 0.5
w2 = (torch.rand(n, k, dtype=torch.float64, device='cuda', generator=g) - 0.5) / k ** 0.5
b = torch.zeros(n, dtype=torch.float64, device='cuda')
img, img2 = ops.gemm_sliced_build(w), ops.gemm_sliced_build(w2)
ref = ops.gemm(a, w, b, a2=a2, w2=w2, bias2=b, act='tanh')
got = ops.gemm_sliced(a, img, n, b, a2=a2, image2=img2, bias2=b, act='tanh')
print('max |sliced - fp64|', float((got - ref).abs().max()), 'max |ref|', float(ref.abs().max()))
for name, fn in (('fp64 mfma', lambda: ops.gemm(a, w, b, a2=a2, w2=w2, bias2=b, act='tanh')),
                 ('int8 sliced', lambda: ops.gemm_sliced(a, img, n, b, a2=a2, image2=img2, bias2=b, act='tanh'))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f'{name:12s} M {m} N {n} K 2x{k}: {ms:.4f} ms  {2.0 * m * n * 2 * k / ms / 1e9:.1f} fp64-equivalent TFLOP/s', flush=True)
