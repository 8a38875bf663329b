#!/usr/bin/env python3
"""Where a cfg-4 train step spends its wall time with the tape's heads on the TAPE instances of the sliced
kernel vs on three fp64 GEMMs + v_update (one process, settings interleaved; phases separated by device
synchronisations: forward trajectory, loss + seeds, reverse sweep, rest = arena / all-reduce / Adam)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
import torch  # noqa: E402
import bench  # noqa: E402
from l2hmc.dynamics.pytorch import training as T  # noqa: E402

args = argparse.Namespace(lattice=[8, 8, 8, 8], units=[256], nchains=256, nleapfrog=4, micro_batch=None,
                          fp64_train_heads=False, beta=6.0)
torch.set_default_dtype(torch.float64)
tr = bench.build_trainer(args, 9992)
x = bench.hot_start(args, 1)
phase = {}


def wrap(name):
    fn = getattr(T, name)

    def w(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        phase[name] = phase.get(name, 0.0) + time.perf_counter() - t0
        return r
    return fn, w


for _ in range(2):
    x, m = tr.train_step((x, args.beta))
for rep in range(2):
    for sliced in (True, False):
        tr.dynamics.sliced_train_heads = sliced
        x, m = tr.train_step((x, args.beta))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            x, m = tr.train_step((x, args.beta))
        torch.cuda.synchronize()
        whole = (time.perf_counter() - t0) / 3
        saved = {}
        for name in ('trajectory_fb_train', 'loss_and_seeds', 'backward', '_native_begin'):
            saved[name], w = wrap(name)
            setattr(T, name, w)
        phase.clear()
        t0 = time.perf_counter()
        for _ in range(3):
            x, m = tr.train_step((x, args.beta))
        torch.cuda.synchronize()
        tot = (time.perf_counter() - t0) / 3
        for name, fn in saved.items():
            setattr(T, name, fn)
        ph = {k: round(v / 3 * 1e3, 2) for k, v in phase.items()}
        print(f'sliced_train_heads={sliced}: step {whole * 1e3:.2f} ms | with phase syncs {tot * 1e3:.2f} ms: {ph} '
              f'rest {tot * 1e3 - sum(ph.values()):.2f}', flush=True)
