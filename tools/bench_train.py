#!/usr/bin/env python3
"""Training-step timing (Trainer.train_step: trajectory tape + reverse sweep + fused Adam) on the
U(1) configs of BASELINE.json.  Prints ms/step and chain*LF/s (2*nlf executed LF steps)."""
import argparse, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
import l2hmc.configs as cfgs  # noqa: E402
from l2hmc.trainers.pytorch.trainer import Trainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--L', type=int, nargs=2, default=[16, 16])
ap.add_argument('--nb', type=int, default=2048)
ap.add_argument('--nlf', type=int, default=8)
ap.add_argument('--beta', type=float, default=4.0)
ap.add_argument('--conv', action='store_true')
ap.add_argument('--steps', type=int, default=5)
ap.add_argument('--warmup', type=int, default=2)
a = ap.parse_args()
torch.manual_seed(9992); np.random.seed(9992)
cfg = cfgs.get_config(['dynamics.group=U1', f'dynamics.latvolume=[{a.L[0]},{a.L[1]}]',
                       f'dynamics.nchains={a.nb}', f'dynamics.nleapfrog={a.nlf}',
                       'dynamics.verbose=false'] + ([] if a.conv else ['conv=none']))
tr = Trainer(cfg)
x = tr.lattice.random()
for _ in range(a.warmup):
    x, m = tr.train_step((x, a.beta))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    x, m = tr.train_step((x, a.beta))
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
print(f'U(1) {a.L} nb={a.nb} nlf={a.nlf} conv={a.conv} train_step: {dt*1e3:.2f} ms/step '
      f'{a.nb * 2 * a.nlf / dt:.3e} chain*LF/s  params={tr.arena.numel()} '
      f'loss={m["loss"]:.4g} acc={float(m["acc"].mean()):.3f}')
