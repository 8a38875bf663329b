#!/usr/bin/env python3
"""SU(3) training-step timing (Trainer.train_step) at BASELINE cfg-4 scale by default:
8^4, 256 chains, nleapfrog 4 (8 LF steps merged), vnet units [256], fp64."""
import argparse, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
import l2hmc.configs as cfgs  # noqa: E402
from l2hmc.trainers.pytorch.trainer import Trainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--L', type=int, nargs=4, default=[8, 8, 8, 8])
ap.add_argument('--nb', type=int, default=256)
ap.add_argument('--nlf', type=int, default=4)
ap.add_argument('--units', type=int, default=256)
ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--warmup', type=int, default=1)
ap.add_argument('--micro-batch', type=int, default=None, help='chains per tape micro-batch')
ap.add_argument('--beta', type=float, default=6.0)
a = ap.parse_args()
torch.manual_seed(9992); np.random.seed(9992)
L = ','.join(str(i) for i in a.L)
cfg = cfgs.get_config(['dynamics.group=SU3', f'dynamics.latvolume=[{L}]', f'dynamics.nchains={a.nb}',
                       f'dynamics.nleapfrog={a.nlf}', 'dynamics.eps=0.01', 'dynamics.verbose=false',
                       'dynamics.use_split_xnets=false', 'dynamics.use_separate_networks=false',
                       f'network.units=[{a.units}]', 'network.activation_fn=tanh',
                       'network.dropout_prob=0.0', 'network.use_batch_norm=false', 'conv=none',
                       'loss.plaq_weight=0.1', 'loss.rmse_weight=0.1', 'loss.charge_weight=0.0'])
tr = Trainer(cfg)
tr.micro_batch = a.micro_batch
x = tr.lattice.random()
for _ in range(a.warmup):
    x, m = tr.train_step((x, a.beta))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    x, m = tr.train_step((x, a.beta))
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
print(f'SU(3) {a.L} nb={a.nb} nlf={a.nlf} units={a.units} micro_batch={a.micro_batch} train_step: {dt*1e3:.1f} ms/step '
      f'{a.nb * 2 * a.nlf / dt:.3e} chain*LF/s  params={tr.arena.numel()} loss={m["loss"]:.4g} '
      f'acc={float(m["acc"].mean()):.3f}  peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB')
