#!/usr/bin/env python3
"""How many torch threads should bench.py's cpu_baseline use on this host?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import torch_cpu as tc
L = (8, 8, 8, 8); nbc = 16
gen = torch.Generator().manual_seed(1)
z = torch.randn((nbc, 4, *L, 3, 3, 2), dtype=torch.float64, generator=gen)
x = tc.project_su(torch.view_as_complex(z))
nrm = torch.randn((8, nbc, 4, *L), dtype=torch.float64, generator=gen)
u = torch.rand(nbc, dtype=torch.float64, generator=gen)
import numpy as np
sim = tc.TorchSU3Dynamics(L, 1, [0.01], [0.01], [np.zeros(36 * 4096)], None)
for nt in (4, 8, 16, 32, 64, 128):
    if nt > (os.cpu_count() or 1):
        break
    torch.set_num_threads(nt)
    sim.apply_transition_hmc(x[:2], 6.0, nrm[:, :2], u[:2], 0.01, 1)
    t0 = time.perf_counter()
    sim.apply_transition_hmc(x, 6.0, nrm, u, 0.01, 2)
    dt = time.perf_counter() - t0
    print(f'threads {nt}: {nbc * 2 / dt:.2f} chain*LF/s (HMC, {nbc} chains x 2 LF in {dt:.2f} s)', flush=True)
