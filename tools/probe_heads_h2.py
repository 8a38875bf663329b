#!/usr/bin/env python3
"""fused half-precision heads + update at the cfg-3 shape under the tuning knobs (stagger, tile, order)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
from l2hmc import _ops as ops, native  # noqa: E402
M, N, K = 8192, 8192, 256
hd = torch.float16
torch.manual_seed(0)
dev = 'cuda'
z = (torch.randn(M, K, device=dev) * 0.5).to(hd)
W = {k: (torch.randn(N, K, device=dev) / 16).to(hd) for k in 'stq'}
b = {k: torch.randn(N, device=dev) * 0.1 for k in 'stq'}
one = torch.ones(N, device=dev)
heads = {'s': (W['s'], b['s'], one), 't': (W['t'], b['t'], None), 'q': (W['q'], b['q'], one)}
v = torch.randn(M, N, device=dev)
f = torch.randn(M, N, device=dev)
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for bm in (128, 64):
    native.set_tuning('heads_h_bm', bm)
    for order in (0, 1):
        native.set_tuning('heads_h_order', order)
        for stg in (0, 1, 2, 4, 8):
            native.set_tuning('heads_stagger', stg)
            t = timeit(lambda: ops.u1_heads_update_h_(z, heads, 1.0, v, f, 0.05, True))
            print(f'bm {bm} order {order} stagger {stg}: {t:.4f} ms', flush=True)
