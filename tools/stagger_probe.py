import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from l2hmc import _ops as ops, native
from kbench import timeit
nb, h, V = 256, 256, 4096
N_ = 36 * V
z = torch.randn(nb, h, dtype=torch.float64, device='cuda')
heads = {k: (torch.randn(N_, h, dtype=torch.float64, device='cuda') / 16, torch.randn(N_, dtype=torch.float64, device='cuda'),
             None if k == 't' else torch.ones(N_, dtype=torch.float64, device='cuda')) for k in 'stq'}
v = torch.randn(nb, N_, dtype=torch.complex128, device='cuda'); f = torch.randn(nb, N_, dtype=torch.complex128, device='cuda')
for stg in (0, 2, 4, 6, 8, 12, 16, 0):
    native.set_tuning('heads_stagger', stg)
    t1 = timeit(lambda: ops.vnet_heads_vupdate_(z, heads, (1., 1., 1.), v, f, 0.001, True), iters=8, warm=2)
    t2 = timeit(lambda: ops.vnet_heads_vupdate_pair_(z, heads, (1., 1., 1.), v, f, 0.001, True, False, 0.001, True), iters=8, warm=2)
    print(f'stagger={stg:3d} single {t1*1e3:.3f} ms  pair {t2*1e3:.3f} ms')
