#!/usr/bin/env python3
"""Where do the device-to-device copies of one SU(3) cfg-4 train step come from? (torch profiler)"""
import collections, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
import bench
sys.argv = [sys.argv[0], '--mode', 'train']
args = bench.parse()
tr = bench.build_trainer(args, 9992)
x = bench.hot_start(args, seed=9992)
for _ in range(2):
    x, m = tr.train_step((x, args.beta))
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    x, m = tr.train_step((x, args.beta))
    torch.cuda.synchronize()
seen = collections.Counter(); tot = collections.Counter()
for ev in prof.events():
    if ev.device_time_total > 50 and ('Memcpy' in ev.name or ev.name in ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::index_select', 'aten::index_add_', 'aten::add_', 'aten::mul', 'aten::zero_', 'aten::fill_', 'aten::add', 'aten::sum')):
        st = [f for f in (ev.stack or []) if 'l2hmc' in f][:2]
        k = (ev.name, ' <- '.join(x.split('/')[-1] for x in st))
        seen[k] += 1; tot[k] += ev.device_time_total
for k, t in sorted(tot.items(), key=lambda kv: -kv[1])[:25]:
    print(f'{t/1e3:8.2f} ms  {seen[k]:3d} x  {k[0]:18s} {k[1]}')
