import sys
sys.path.insert(0, '/root/repo/l2hmc-qcd_amd')
import torch, numpy as np
torch.set_default_dtype(torch.float64)
import l2hmc.configs as cfgs
from l2hmc.dynamics.pytorch.dynamics import Dynamics
from l2hmc.lattice.su3.pytorch.lattice import LatticeSU3
torch.manual_seed(3)
for beta in (0.9, 1.8):
    L, nb = [4, 4, 4, 4], 64
    dc = cfgs.DynamicsConfig(nchains=nb, group='SU3', latvolume=L, nleapfrog=5, eps=0.1, eps_hmc=0.1,
                             verbose=False, use_split_xnets=False, use_separate_networks=False)
    lat = LatticeSU3(nb, L)
    dyn = Dynamics(lat.action, dc, None).eval()
    x = lat.random().to(dyn.device)
    b = torch.tensor(beta)
    ps, acc = [], []
    for i in range(160):
        xo, m = dyn.apply_transition_hmc((x, b), eps=0.1, nleapfrog=10)
        x = dyn.g.compat_proj(xo.reshape(x.shape))
        if i >= 60:
            ps.append(lat.plaqs(x).mean()); acc.append(m['acc'].mean())
    est = float(torch.stack(ps).mean()); err = float(torch.stack(ps).std()) / len(ps) ** 0.5
    print(f'beta {beta}: <plaq> = {est:.5f} +- {err:.5f}  series b/18 + b^2/216 = {beta/18 + beta**2/216:.5f}  acc {float(torch.stack(acc).mean()):.3f}')
