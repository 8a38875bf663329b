#!/usr/bin/env python3
"""fused half-precision heads + v-update at the cfg-3 shape on four rotating operand sets (> MALL);
run under L2Q_LIB_NAME=libl2q_hh<n>.so (-DL2Q_HH_SKIP=n: 1 no K-loop, 2 no epilogue math, 4 no field traffic)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
from l2hmc import _ops as ops, native  # noqa: E402
M, N, K = 8192, 8192, 256
for kv in sys.argv[1:]:                      # key=value tuning knobs
    k_, v_ = kv.split('=')
    assert native.set_tuning(k_, int(v_)) >= 0, kv
hd = torch.float16
torch.manual_seed(0)
dev = 'cuda'
z = (torch.randn(M, K, device=dev) * 0.5).to(hd)
W = {k: (torch.randn(N, K, device=dev) / 16).to(hd) for k in 'stq'}
b = {k: torch.randn(N, device=dev) * 0.1 for k in 'stq'}
one = torch.ones(N, device=dev)
heads = {'s': (W['s'], b['s'], one), 't': (W['t'], b['t'], None), 'q': (W['q'], b['q'], one)}
sets = [(torch.randn(M, N, device=dev), torch.randn(M, N, device=dev)) for _ in range(4)]
mask = (torch.rand(N, device=dev) > 0.5).float()
def timeit(fn, reps=12):
    for i in range(4): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
tv = timeit(lambda i: ops.u1_heads_update_h_(z, heads, 1.0, sets[i & 3][0], sets[i & 3][1], 0.05, True))
tx = timeit(lambda i: ops.u1_heads_update_h_(z, heads, 1.0, sets[i & 3][0], sets[i & 3][1], 0.05, True, mask=mask))
print(f'{os.environ.get("L2Q_LIB_NAME", "libl2q.so"):18s} {" ".join(sys.argv[1:]):18s} v-update {tv:.4f} ms   x-update (NCP) {tx:.4f} ms')
