"""SU(3) force at the bench shape per value of the launch-order tunings (stagger of the second resident set, XCD remap,
t-range chunks): interleaved rounds, median."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'l2hmc-qcd_amd'))
from l2hmc import _ops as ops, native
L = [8, 8, 8, 8]; nb = 256; V = 4096
torch.manual_seed(0)
xn = ops.su3_project_su_n(torch.randn(nb, 4, 9, V, dtype=torch.complex128, device='cuda'))
f = torch.empty_like(xn)
def t():
    for _ in range(3): native.call('l2q_su3_force', xn, 6.0, f, nb, *L)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): native.call('l2q_su3_force', xn, 6.0, f, nb, *L)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 30
res = {}
for rnd in range(3):
    for key, vals in (('force_stagger', (0, 1, 2, 4, 8, 16)), ('xcd_swizzle', (1, 0)), ('force_tsplit', (0, 4))):
        for v in vals:
            native.set_tuning(key, v)
            res.setdefault((key, v), []).append(t())
        native.set_tuning(key, vals[0])
for k, v in res.items():
    print(f'[sweep {k[0]}={k[1]}] {sorted(v)[1]:.4f} ms')
