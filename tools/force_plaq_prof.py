#!/usr/bin/env python3
"""In-kernel cycle totals of su3_force_plaq_kernel (run with L2Q_LIB_NAME=libl2q_pqprof.so, a -DL2Q_PQ_PROF=1 build of
tools/ab_build.sh): per wavefront of workgroup 0 -- products, wait at the first barrier, gather / refresh, wait at the
second barrier -- summed over its 9 iterations."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
from l2hmc import _ops as ops, native  # noqa: E402

nb, L, V = 256, (8, 8, 8, 8), 4096
torch.manual_seed(0)
xn = ops.su3_project_su_n(torch.randn(nb, 4, 9, V, dtype=torch.complex128, device='cuda'))
f = torch.empty_like(xn)
native.set_tuning('force_tile', 7)
for _ in range(3):
    native.call('l2q_su3_force', xn, 6.0, f, nb, *L)
torch.cuda.synchronize()
# workgroup 0 (after the XCD swizzle: logical block 0 = chain 0, tile 0) dumps over the first 16 complex of its chain
raw = torch.view_as_real(f.reshape(-1)[:16]).cpu().reshape(8, 4)
names = ['(t,y)', '(t,z)', '(y,z)', 'helper A', '(t,x)', '(x,y)', '(x,z)', 'helper B']
print('wavefront   products  wait B1   gather   wait B2   total   (cycles per iteration, 9 iterations)')
for w in range(8):
    t = raw[w] / 9.0
    print(f'{w} {names[w]:9s} {t[0]:8.0f} {t[1]:8.0f} {t[2]:8.0f} {t[3]:8.0f} {t.sum():8.0f}')
