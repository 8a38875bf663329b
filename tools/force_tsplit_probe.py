"""SU(3) force at the bench shape (8^4 x 256 chains) per value of the `force_tsplit` tuning: t-range chunks per chain
(1: a workgroup sweeps all T slices of its 64-site spatial tile; T: one slice per workgroup, a chain's 64 workgroups
then fill the 64 slots of one XCD and the chain (2.4 MB) fits its 4 MB L2)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
from l2hmc import _ops as ops, native  # noqa: E402

L = [int(a) for a in os.environ.get('LAT', '8 8 8 8').split()]
nb = int(os.environ.get('NB', '256'))
V = L[0] * L[1] * L[2] * L[3]
torch.manual_seed(0)
xn = ops.su3_project_su_n(torch.randn(nb, 4, 9, V, dtype=torch.complex128, device='cuda'))
f = torch.empty_like(xn)
ref = None
res = {}
for rnd in range(3):
    for ts in (0, 1, 2, 4, 8):
        native.set_tuning('force_tsplit', ts)
        for _ in range(3):
            native.call('l2q_su3_force', xn, 6.0, f, nb, *L)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            native.call('l2q_su3_force', xn, 6.0, f, nb, *L)
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(ts, []).append(e0.elapsed_time(e1) / 20)
        if ref is None:
            ref = f.clone()
        assert torch.equal(f, ref), ts
for ts, v in res.items():
    ms = sorted(v)[1]
    print(f'[force_tsplit={ts}] {ms:.4f} ms  {nb * V * 1152 / ms * 1e-9:.2f} TB/s algorithmic ({nb * V * 1152 / ms * 1e-9 / 8:.3f} of 8 TB/s)')
