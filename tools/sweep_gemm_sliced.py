"""Input layer, fp64 MFMA kernel vs the int8-sliced one, over the shapes the SU(3) vnet meets
(N = 256 hidden units; M = chains; K = K2 = 8 V 4 for lattice volume V)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'l2hmc-qcd_amd'))
from l2hmc import _ops as ops


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


n = 256
g = torch.Generator(device='cuda').manual_seed(1)
for k in (2048, 8192, 32768, 131072):
    w = (torch.rand(n, k, dtype=torch.float64, device='cuda', generator=g) - 0.5) / k ** 0.5
    w2 = (torch.rand(n, k, dtype=torch.float64, device='cuda', generator=g) - 0.5) / k ** 0.5
    b = torch.zeros(n, dtype=torch.float64, device='cuda')
    img, img2 = ops.gemm_sliced_build(w), ops.gemm_sliced_build(w2)
    for m in (64, 128, 256, 512, 1024):
        a = (torch.rand(m, k, dtype=torch.float64, device='cuda', generator=g) - 0.5) * 4.0
        a2 = (torch.rand(m, k, dtype=torch.float64, device='cuda', generator=g) - 0.5) * 4.0
        t0 = timeit(lambda: ops.gemm(a, w, b, a2=a2, w2=w2, bias2=b, act='tanh'))
        t1 = timeit(lambda: ops.gemm_sliced(a, img, n, b, a2=a2, image2=img2, bias2=b, act='tanh'))
        d = float((ops.gemm(a, w, b, a2=a2, w2=w2, bias2=b, act='tanh')
                   - ops.gemm_sliced(a, img, n, b, a2=a2, image2=img2, bias2=b, act='tanh')).abs().max())
        print(f'M {m:5d} N {n} K 2x{k:7d}: fp64 {t0:.4f} ms  sliced {t1:.4f} ms  ratio {t1 / t0:.3f}  max diff {d:.1e}',
              flush=True)
