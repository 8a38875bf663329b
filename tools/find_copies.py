#!/usr/bin/env python3
"""Where do the device-to-device copies of one cfg-4 trajectory come from?  Wraps Tensor.clone /
copy_ / contiguous / to and prints the caller of every call that moves >= 64 MB."""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
import bench  # noqa: E402

counts = collections.Counter()


def wrap(name):
    orig = getattr(torch.Tensor, name)

    def f(self, *a, **k):
        out = orig(self, *a, **k)
        big = self.numel() * self.element_size() >= 64 << 20
        moved = name in ('clone', 'copy_') or (isinstance(out, torch.Tensor) and out.data_ptr() != self.data_ptr())
        if big and moved and self.is_cuda:
            fr = traceback.extract_stack(limit=4)[:-1]
            counts[(name, ' <- '.join(f'{os.path.basename(x.filename)}:{x.lineno}' for x in reversed(fr)))] += 1
        return out
    setattr(torch.Tensor, name, f)


for n in ('clone', 'copy_', 'contiguous', 'to'):
    wrap(n)


import argparse
sys.argv = [sys.argv[0]]
args = bench.parse()
dyn, lat = bench.build(args, 9992)
x = bench.hot_start(args, seed=9992)
beta = torch.tensor(args.beta)
for _ in range(2):
    xo, m = dyn((x, beta))
torch.cuda.synchronize()
counts.clear()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    xo, m = dyn((x, beta))
    torch.cuda.synchronize()
seen = collections.Counter()
for ev in prof.events():
    if ev.device_time_total > 150 and ('copy' in ev.name.lower() or 'clone' in ev.name.lower() or 'Memcpy' in ev.name):
        st = [f for f in (ev.stack or []) if 'l2hmc' in f or 'bench' in f][:3]
        seen[(ev.name, round(ev.device_time_total), ' <- '.join(x.split('/')[-1] for x in st))] += 1
for k, n in sorted(seen.items(), key=lambda kv: -kv[1])[:30]:
    print(n, k)
for (name, where), n in sorted(counts.items(), key=lambda kv: -kv[1]):
    print(f'{n:3d} x {name:10s} {where}')
