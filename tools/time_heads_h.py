"""cfg-3 heads + update alone (8192 chains x 8192 entries, K = 256, fp16): v- and x-update per value of the
`heads_h_stream` tuning (0 tile kernel, 1 stream kernel, 2 K-split stream kernel).  Interleaved rounds, median;
four rotating operand sets (the fields are 268 MB each: beyond the 256 MB L3)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
from l2hmc import _ops as ops, native  # noqa: E402

m, n, k = 8192, 8192, 256
hd = torch.float16
g = torch.Generator(device='cuda').manual_seed(1)
z = torch.randn(m, k, device='cuda', generator=g).to(hd)
heads = {}
for nm in 'stq':
    w = (torch.randn(n, k, device='cuda', generator=g) / k ** 0.5).to(hd)
    b = 0.1 * torch.randn(n, device='cuda', generator=g)
    c = None if nm == 't' else 0.7 * torch.exp(0.3 * torch.randn(n, device='cuda', generator=g))
    heads[nm] = (w, b, c)
mask = (torch.rand(n, device='cuda', generator=g) < 0.5).float()
sets = [(torch.randn(m, n, device='cuda', generator=g), torch.randn(m, n, device='cuda', generator=g)) for _ in range(3)]


def run(xupd, reps=12):
    def once(i):
        a, b = sets[i % len(sets)]
        ops.u1_heads_update_h_(z, heads, 0.9, a, b, 0.01, True, mask=mask if xupd else None, complement=False)
    once(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        once(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


variants = [int(a) for a in sys.argv[1:]] or [0, 1, 2]
res = {(v, x): [] for v in variants for x in (0, 1)}
for rnd in range(5):
    for v in variants:
        native.set_tuning('heads_h_stream', v)
        for x in (0, 1):
            res[(v, x)].append(run(bool(x)))
tag = os.environ.get('L2Q_LIB_NAME', 'libl2q.so')
for v in variants:
    tv, tx = sorted(res[(v, 0)])[2], sorted(res[(v, 1)])[2]
    print(f'[{tag} heads_h_stream={v}] v-update {tv:7.1f} us ({3 * m * n * 4 / tv * 1e-6:.2f} TB/s)   '
          f'x-update {tx:7.1f} us ({3 * m * n * 4 / tx * 1e-6:.2f} TB/s)')
