#!/usr/bin/env python3
"""PMC target: stencil kernel variants only."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
from l2hmc import _ops as ops, native  # noqa: E402
nb, L, V = 256, (8, 8, 8, 8), 4096
torch.manual_seed(0)
xn = ops.su3_project_su_n(torch.randn(nb, 4, 9, V, dtype=torch.complex128, device='cuda'))
f = torch.empty_like(xn)
for _ in range(3):
    for sweep in (0, 1):
        native.set_tuning('plaq_sweep', sweep)
        ops.su3_plaq_sums_n(xn, L)
    for tile in (0, 1):
        native.set_tuning('force_tile', tile)
        native.call('l2q_su3_force', xn, 6.0, f, nb, *L)
torch.cuda.synchronize()
