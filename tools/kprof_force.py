#!/usr/bin/env python3
"""Target of rocprofv3 --pmc passes for the force kernels alone: force_tile = 7 (plaquette sharing) and 5 (thread per
link) at the cfg-4 size (L2Q_KPROF_LATTICE / L2Q_KPROF_NB for others).
    L2Q_KPROF_SCRIPT=tools/kprof_force.py L2Q_PMC_JSON=gpurun_out/pmc_force.json bash tools/pmc_collect.sh <tag>"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
from l2hmc import _ops as ops, native  # noqa: E402

L = tuple(int(i) for i in os.environ.get('L2Q_KPROF_LATTICE', '8 8 8 8').split())
nb = int(os.environ.get('L2Q_KPROF_NB', 256))
V = L[0] * L[1] * L[2] * L[3]
torch.manual_seed(0)
xn = ops.su3_project_su_n(torch.randn(nb, 4, 9, V, dtype=torch.complex128, device='cuda'))
f = torch.empty_like(xn)
for tile in (7, 5):
    native.set_tuning('force_tile', tile)
    for _ in range(4):
        native.call('l2q_su3_force', xn, 6.0, f, nb, *L)
torch.cuda.synchronize()
