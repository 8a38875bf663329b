#!/usr/bin/env python3
"""fp64 heads + v-update kernel from Python (torch tensors) at the cfg-4 shape: back-to-back
launches vs launches interleaved with the kernels that surround it in a trajectory."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
from l2hmc import _ops as ops, native  # noqa: E402

nb, L, V, h = 256, (8, 8, 8, 8), 4096, 256
torch.manual_seed(0)
xn = ops.su3_project_su_n(torch.randn(nb, 4, 9, V, dtype=torch.complex128, device='cuda'))
vn = ops.su3_assemble_tah_n(torch.randn(8, nb, 4, V, dtype=torch.float64, device='cuda'))
f = torch.empty_like(xn)
native.call('l2q_su3_force', xn, 6.0, f, nb, *L)
z = torch.randn(nb, h, dtype=torch.float64, device='cuda')
N_ = 36 * V
heads = {k: (torch.randn(N_, h, dtype=torch.float64, device='cuda') / 16,
             torch.randn(N_, dtype=torch.float64, device='cuda'),
             None if k == 't' else torch.ones(N_, dtype=torch.float64, device='cuda')) for k in 'stq'}
K = 32 * V
xv = torch.randn(nb, K, dtype=torch.float64, device='cuda')
fv = torch.randn(nb, K, dtype=torch.float64, device='cuda')
wx = torch.randn(h, K, dtype=torch.float64, device='cuda') / K ** 0.5
wv = torch.randn(h, K, dtype=torch.float64, device='cuda') / K ** 0.5
bx = torch.randn(h, dtype=torch.float64, device='cuda')


def pair():
    ops.vnet_heads_vupdate_pair_(z, heads, (1., 1., 1.), vn.reshape(nb, -1), f.reshape(nb, -1), 0.01, True, False, 0.01, True)


def timed(fn, pre=None, reps=10):
    tot = 0.0
    for i in range(reps + 3):
        if pre is not None:
            pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if i >= 3:
            tot += e0.elapsed_time(e1)
    return tot / reps


def surround():
    native.call('l2q_su3_force', xn, 6.0, f, nb, *L)
    ops.su3_projsu_vec8_n(f)
    ops.gemm(xv, wx, bx, a2=fv, w2=wv, bias2=bx, act='tanh')


print(f'pair, back to back:           {timed(pair):.4f} ms  (includes memset + finalize launches)')
print(f'pair, after force+vec8+gemm:  {timed(pair, surround):.4f} ms')
for _ in range(3):
    pair()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    pair()
e1.record()
torch.cuda.synchronize()
print(f'pair, 20 launches in a row:   {e0.elapsed_time(e1) / 20:.4f} ms')

# sustained load: do the clocks hold?  Blocks of 100 launches, with the SMI's view of the clocks and
# the power draw sampled from the host while the queue drains.
if '--sustained' in sys.argv:
    import subprocess
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    evs[0].record()
    for b in range(20):
        for _ in range(100):
            pair()
        evs[b + 1].record()
    smi = subprocess.run('rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|Power" | head -6',
                         shell=True, capture_output=True, text=True).stdout
    torch.cuda.synchronize()
    print('pair, 20 blocks of 100 launches (ms per launch):',
          ' '.join(f'{evs[b].elapsed_time(evs[b + 1]) / 100:.3f}' for b in range(20)))
    print(smi)
