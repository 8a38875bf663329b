import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'l2hmc-qcd_amd'))
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tools'))
from l2hmc import _ops as ops, native
from kbench import timeit
nb, L = 256, (8, 8, 8, 8); V = 4096
torch.manual_seed(0)
xn = ops.su3_project_su_n(torch.randn(nb, 4, 9, V, dtype=torch.complex128, device='cuda'))
for tile in (0, 1, 2):
    for occ in (2, 3, 4):
        native.set_tuning('force_tile', tile); native.set_tuning('force_occ', occ)
        t = timeit(lambda: ops.su3_force_n(xn, 6.0, L))
        print(f'force_tile={tile} occ={occ}: {t*1e3:.3f} ms  {nb*V*1152/t/1e9:.0f} GB/s')
