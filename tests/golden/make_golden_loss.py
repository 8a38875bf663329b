#!/usr/bin/env python3
"""Golden LatticeLoss values from the REAL reference on the already committed trajectories
(run like make_golden.py, once per default dtype):
    PYTHONPATH=/tmp/oracle_stubs:/tmp/oracle/src python3 tests/golden/make_golden_loss.py su3|u1
"""
import os
import sys

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
WHICH = sys.argv[1]
if WHICH == 'su3':
    torch.set_default_dtype(torch.float64)
import l2hmc.configs as cfgs  # noqa: E402
from l2hmc.loss.pytorch.loss import LatticeLoss  # noqa: E402

out = {}
if WHICH == 'su3':
    from l2hmc.lattice.su3.pytorch.lattice import LatticeSU3
    g = np.load(os.path.join(OUT, 'su3_l2hmc.npz'))
    L = [int(i) for i in g['latvolume']]
    x, xp, acc = (torch.from_numpy(g[k]) for k in ('x', 'x_prop', 'acc'))
    lat = LatticeSU3(x.shape[0], L)
    for name, lc in (('su3', cfgs.LossConfig(use_mixed_loss=True, charge_weight=0.0,
                                             rmse_weight=0.1, plaq_weight=0.1)),
                     ('mix', cfgs.LossConfig(use_mixed_loss=False, charge_weight=0.3,
                                             rmse_weight=1.0, plaq_weight=0.5))):
        lf = LatticeLoss(lat, lc)
        out[f'{name}_loss'] = lf(x, xp, acc).detach().numpy()
        out[f'{name}_plaq'] = lf.plaq_loss(x, xp, acc).detach().numpy()
        out[f'{name}_charge'] = lf.charge_loss(x, xp, acc).detach().numpy()
        out[f'{name}_rmse'] = lf.rmse_loss(x, xp, acc).detach().numpy()
    m = lf.lattice_metrics(x, torch.from_numpy(g['x_out']).reshape(x.shape))
    out.update({'dQint': m['dQint'].numpy(), 'dQsin': m['dQsin'].numpy()})
else:
    from l2hmc.lattice.u1.pytorch.lattice import LatticeU1
    g = np.load(os.path.join(OUT, 'u1_c1.npz'))
    L = [int(i) for i in g['latvolume']]
    x, xp, acc = (torch.from_numpy(g[k]) for k in ('x', 'x_prop', 'acc'))
    lat = LatticeU1(x.shape[0], L)
    lf = LatticeLoss(lat, cfgs.LossConfig(use_mixed_loss=True, charge_weight=0.01))
    out['default_loss'] = lf(x, xp, acc).detach().numpy()
    lf = LatticeLoss(lat, cfgs.LossConfig(use_mixed_loss=False, charge_weight=0.5))
    out['plain_loss'] = lf(x, xp, acc).detach().numpy()
    m = lf.lattice_metrics(x, torch.from_numpy(g['x_out']).reshape(x.shape))
    out.update({'dQint': m['dQint'].numpy(), 'dQsin': m['dQsin'].numpy()})
np.savez(os.path.join(OUT, f'loss_{WHICH}.npz'), **out)
print({k: float(v) for k, v in out.items() if v.ndim == 0})
