#!/usr/bin/env python3
"""Shader clock and socket power as the SMI reports them while the cfg-4 trajectory (bench.py's workload)
runs in steady state, against the same for the fp64 heads kernel alone and the force kernel alone."""
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
import bench  # noqa: E402
from l2hmc import native  # noqa: E402


def smi():
    out = subprocess.run('rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|fclk|mclk|Power"',
                         shell=True, capture_output=True, text=True).stdout
    return ' | '.join(x.split(':', 1)[1].strip() for x in out.strip().splitlines() if ':' in x)


def sample(label, fn, secs=4.0):
    stop = [False]
    res = []

    def poll():
        time.sleep(1.0)
        while not stop[0]:
            res.append(smi())
            time.sleep(0.5)
    th = threading.Thread(target=poll)
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < secs:
        fn()
        n += 1
        if n % 4 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    stop[0] = True
    th.join()
    print(f'== {label}: {dt * 1e3:.3f} ms per call')
    for r in res:
        print('   ', r)


sys.argv = [sys.argv[0]]
args = bench.parse()
dyn, lat = bench.build(args, 9992)
x = bench.hot_start(args, seed=9992)
beta = torch.tensor(args.beta)
st = {'x': x}


def traj():
    st['x'], _ = dyn((st['x'], beta))


sample('L2HMC trajectory (cfg-4)', traj)
nb, L = args.nchains, tuple(args.lattice)
xn = dyn._pack_input(st['x']) if hasattr(dyn, '_pack_input') else None
from l2hmc import _ops as ops  # noqa: E402
V = L[0] * L[1] * L[2] * L[3]
xn = ops.su3_project_su_n(torch.randn(nb, 4, 9, V, dtype=torch.complex128, device='cuda'))
f = torch.empty_like(xn)
sample('force kernel alone', lambda: native.call('l2q_su3_force', xn, 6.0, f, nb, *L))
