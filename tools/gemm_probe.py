import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
from l2hmc import _ops as ops
from kbench import timeit
nb, h, V = 256, 256, 4096
N_ = 36 * V
z = torch.randn(nb, h, dtype=torch.float64, device='cuda')
w = torch.randn(N_, h, dtype=torch.float64, device='cuda') / 16
b = torch.randn(N_, dtype=torch.float64, device='cuda')
co = torch.zeros(N_, dtype=torch.float64, device='cuda')
fl = 2.0 * nb * h * N_
for name, kw in (('none', {}), ('bias', {'bias': b}), ('tanh', {'bias': b, 'act': 'tanh'}), ('tanh+coeff', {'bias': b, 'act': 'tanh', 'coeff': co}), ('relu', {'bias': b, 'act': 'relu'})):
    bias = kw.pop('bias', None)
    t = timeit(lambda: ops.gemm(z, w, bias, **kw), iters=5, warm=2)
    print(f'head gemm epilogue={name:12s} {t*1e3:7.3f} ms {fl/t/1e12:6.2f} TF')
for M in (64, 128, 256, 512, 1024):
    zz = torch.randn(M, h, dtype=torch.float64, device='cuda')
    t = timeit(lambda: ops.gemm(zz, w), iters=5, warm=2)
    print(f'head gemm M={M:5d} none {t*1e3:7.3f} ms {2.0*M*h*N_/t/1e12:6.2f} TF')
for K in (256, 1024, 4096):
    zz = torch.randn(256, K, dtype=torch.float64, device='cuda'); ww = torch.randn(36 * 1024, K, dtype=torch.float64, device='cuda')
    t = timeit(lambda: ops.gemm(zz, ww), iters=5, warm=2)
    print(f'gemm 256 x {36*1024} x K={K:5d} {t*1e3:7.3f} ms {2.0*256*K*36*1024/t/1e12:6.2f} TF')
a = torch.randn(4096, 4096, dtype=torch.float64, device='cuda'); bb = torch.randn(4096, 4096, dtype=torch.float64, device='cuda')
t = timeit(lambda: ops.gemm(a, bb), iters=3, warm=1); print(f'gemm 4096^3 mine {t*1e3:7.3f} ms {2*4096**3/t/1e12:6.2f} TF')
t = timeit(lambda: a @ bb.t(), iters=3, warm=1); print(f'gemm 4096^3 rocBLAS {t*1e3:7.3f} ms {2*4096**3/t/1e12:6.2f} TF')
