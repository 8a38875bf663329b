#!/usr/bin/env python3
"""Golden vectors for the improved (c1 != 0) SU(3) gauge action from the REAL reference:
    bash tests/golden/setup_reference_env.sh
    PYTHONPATH=/tmp/oracle_stubs:/tmp/oracle/src python3 tests/golden/make_golden_c1.py
Writes tests/golden/su3_c1.npz: action, rectangle traces, autograd force, one plain-HMC
transition and one merged L2HMC trajectory driven through Dynamics with a c1 != 0 potential
(the reference's Dynamics keeps a c1 = 0 lattice for the force, dynamics.py:134-135, 1493-1499)."""
import os

import numpy as np
import torch

torch.set_default_dtype(torch.float64)
import l2hmc.configs as cfgs  # noqa: E402
from l2hmc.dynamics.pytorch.dynamics import Dynamics  # noqa: E402
from l2hmc.lattice.su3.pytorch.lattice import LatticeSU3  # noqa: E402
from l2hmc.network.pytorch.network import NetworkFactory  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.manual_seed(11)
np.random.seed(11)
L = [4, 2, 4, 2]
nb, c1, beta = 2, -0.331, 5.7
lat = LatticeSU3(nb, L, c1=c1)
x = lat.g.random([nb, 4, *L, 3, 3]) if hasattr(lat.g, 'random') else None
out = {'latvolume': np.array(L), 'c1': np.array(c1), 'beta': np.array(beta), 'x': x.numpy()}
b = torch.tensor(beta)
out['action'] = lat.action(x.clone(), b).detach().numpy()
ps, rs = lat._wilson_loops(x.clone(), needs_rect=True)
out['rects'] = rs.detach().numpy()
out['plaq_sum'] = ps.real.sum(tuple(range(2, ps.ndim))).sum(0).numpy()
out['rect_sum'] = rs.real.sum(tuple(range(2, rs.ndim))).sum(0).numpy()
out['force'] = lat.grad_action(x.clone(), b).detach().numpy()
urul, uuud = lat._rectangles(x, 2, 1)
out['rect_21_urul'] = urul.numpy()
out['rect_21_uuud'] = uuud.numpy()

# Dynamics with the improved action as potential_fn: plain HMC with injected momenta
dc = cfgs.DynamicsConfig(nchains=nb, group='SU3', latvolume=L, nleapfrog=2, eps=0.02,
                         eps_hmc=0.05, use_split_xnets=False, use_separate_networks=False,
                         verbose=True)
nc = cfgs.NetworkConfig(units=[4], activation_fn='tanh', dropout_prob=0.0, use_batch_norm=False)
xdim = int(np.prod(L)) * 4 * 8
spec = cfgs.InputSpec(xshape=tuple(dc.xshape), xnet={'x': [xdim], 'v': [xdim]},
                      vnet={'x': [xdim], 'v': [xdim]})
dyn = Dynamics(lat.action, dc, NetworkFactory(spec, nc, cfgs.ConvolutionConfig()))
dyn.eval()
torch.manual_seed(5)
g0 = torch.get_rng_state()
nrm = torch.stack([torch.randn([nb, 4, *L]) for _ in range(8)])
u = torch.rand(nb)
torch.set_rng_state(g0)
xo, m = dyn.apply_transition_hmc((x.clone(), b), eps=0.05, nleapfrog=3)
out.update({'hmc_normals': nrm.numpy(), 'hmc_u': u.numpy(), 'hmc_x_out': xo.detach().numpy(),
            'hmc_acc': m['acc'].detach().numpy(), 'hmc_acc_mask': m['acc_mask'].detach().numpy(),
            'hmc_energy': torch.stack(list(m['energy'])).detach().numpy()
            if isinstance(m['energy'], (list, tuple)) else m['energy'].detach().numpy()})
np.savez(os.path.join(OUT, 'su3_c1.npz'), **out)
print({k: v.shape for k, v in out.items()})
print('action', out['action'], 'acc', out['hmc_acc'])
