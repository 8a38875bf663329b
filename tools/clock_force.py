#!/usr/bin/env python3
"""Shader clock / socket power (rocm-smi) while ONE force kernel runs back to back: the shipping thread-per-link kernel
(force_tile 5), the plaquette-sharing kernel (7), and -- with L2Q_LIB_NAME=libl2q_pq15.so, a timing build of
tools/ab_build.sh ... -DL2Q_PQ_EXP=15 -- the same kernel without memory traffic.  Answers whether "memory time adds
to the arithmetic" is the power cap (clock drops when HBM is busy) or a stall."""
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
from l2hmc import _ops as ops, native  # noqa: E402


def smi():
    out = subprocess.run('rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|Power"',
                         shell=True, capture_output=True, text=True).stdout
    return ' | '.join(x.split(':', 1)[1].strip() for x in out.strip().splitlines() if ':' in x)


nb, L = 256, (8, 8, 8, 8)
V = 4096
torch.manual_seed(0)
xn = ops.su3_project_su_n(torch.randn(nb, 4, 9, V, dtype=torch.complex128, device='cuda'))
f = torch.empty_like(xn)
for tile in (5, 7):
    native.set_tuning('force_tile', tile)
    res, stop = [], [False]

    def poll():
        time.sleep(1.5)
        while not stop[0]:
            res.append(smi())
            time.sleep(0.5)
    th = threading.Thread(target=poll)
    th.start()
    t0, n = time.time(), 0
    while time.time() - t0 < 4.5:
        for _ in range(50):
            native.call('l2q_su3_force', xn, 6.0, f, nb, *L)
        n += 50
        torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    stop[0] = True
    th.join()
    print(f'== {os.environ.get("L2Q_LIB_NAME", "libl2q.so")} force_tile={tile} {native.kernel_name("l2q_su3_force", L)}: {dt * 1e3:.4f} ms per call')
    for r in res[:5]:
        print('   ', r)
