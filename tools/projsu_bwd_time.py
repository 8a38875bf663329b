import sys, os
sys.path.insert(0, 'l2hmc-qcd_amd'); sys.path.insert(0, 'tools')
import torch
from l2hmc import _ops as ops, native
from kbench import timeit
nb, V = 256, 4096
torch.manual_seed(0)
x = ops.su3_project_su_n(torch.randn(nb, 4, 9, V, dtype=torch.complex128, device='cuda'))
F = torch.randn(nb, 4, 9, V, dtype=torch.complex128, device='cuda')
F = F - F.transpose(-1, -2) if False else F
gvec = torch.randn(nb, 4, 8, V, dtype=torch.float64, device='cuda')
gm = torch.zeros_like(x)
for name, m in (('unitary x', x), ('generic F', F)):
    t = timeit(lambda: ops.su3_projsu_vec8_bwd_(gm, m, gvec), iters=10, warm=3)
    print(f'{os.environ.get("L2Q_LIB_NAME", "libl2q.so"):18s} projsu_vec8_bwd {name}: {t*1e3:.4f} ms', flush=True)
