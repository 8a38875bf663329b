#!/usr/bin/env python3
"""Calibration helper for tests/test_sizes_gpu.py::test_cfg5_l2hmc_trajectory_16x4_256chains:
acceptance of the product on the test's inputs for candidate (head_scale, eps)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'l2hmc-qcd_amd'), ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import test_sizes_gpu as T  # noqa: E402

torch.set_default_dtype(torch.float64)
L, nb = T.L16, 2
START = sys.argv[1] if len(sys.argv) > 1 else 'warm'
rng = np.random.default_rng(5)
x2 = T._warm(rng, 2, L) if START == 'warm' else T._hot(rng, 2, L)
nrm2 = rng.normal(size=(8, 2, 4, *L))
prev = 1.0
dyn, lat = T._build(L, nb, 1, [256], eps=0.005, head_scale=1.0, seed=12)
dyn.config.verbose = True
for hs in ((0.1, 0.03, 0.01) if START == 'hot' else (0.003,)):
    with torch.no_grad():
        for lin in (dyn.vnet.scale.layer, dyn.vnet.transl, dyn.vnet.transf.layer):
            lin.weight.mul_(hs / prev)
            lin.bias.mul_(hs / prev)
    prev = hs
    for eps in ((0.0015, 0.0025, 0.004) if START == 'hot' else (2e-4, 3e-4, 5e-4)):
        dyn.assign_eps(float(eps))
        dyn._inject = {'normals': nrm2, 'u': np.zeros(2)}
        xo, m = dyn((T.dev(x2), torch.tensor(6.2)))
        e = m['energy']
        print(f'{START} hs={hs} eps={eps} acc={m["acc"].tolist()} dH={(e[0] - e[-1]).tolist()} '
              f'sld={m["sumlogdet"].tolist()}', flush=True)
