#!/usr/bin/env python3
"""The kernels of the SU(3) reverse sweep at the cfg-4 size (8^4 x 256 chains), a few launches each -- the target
of rocprofv3 --pmc passes (tools/pmc_collect.sh with L2Q_KPROF_SCRIPT=tools/kprof_train.py)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
from l2hmc import _ops as ops, native  # noqa: E402

L = (8, 8, 8, 8)
nb, V = 256, 4096
n = 36 * V
torch.manual_seed(0)
c128 = lambda *s: torch.randn(*s, dtype=torch.complex128, device='cuda')
f64 = lambda *s: torch.randn(*s, dtype=torch.float64, device='cuda')
xn = ops.su3_project_su_n(c128(nb, 4, 9, V))
vn = ops.su3_assemble_tah_n(f64(8, nb, 4, V))
F, g, gv, gx = c128(nb, 4, 9, V), c128(nb, 4, 9, V), c128(nb, 4, 9, V), c128(nb, 4, 9, V)
mask = (torch.rand(n, device='cuda') > 0.5).float()
s, t, q = (0.3 * f64(nb, n) for _ in range(3))
gl = f64(nb)
gvec = f64(nb, 4, 8, V)
coeff = 0.1 * f64(n)
bg, cg = torch.zeros(n, dtype=torch.float64, device='cuda'), torch.zeros(n, dtype=torch.float64, device='cuda')
z = torch.tanh(f64(nb, 256))
W = [f64(n, 256) / 16 for _ in range(3)]
b = [0.1 * f64(n) for _ in range(3)]
cs = torch.exp(0.1 * f64(n))
image, ok = ops.heads_sliced_build_into(*W)
assert ok
for _ in range(3):
    ops.su3_expm_mul2_bwd_n(xn, vn, 0.01, mask, False, g, gv)
    ops.su3_expm_mul_bwd_n(xn, vn, 0.01, mask, False, g, gv)
    ops.v_update_bwd_pair_c128(vn.reshape(nb, -1), F.reshape(nb, -1), g.reshape(nb, -1), s, t, q, 0.01, True, 0.01,
                               True, False, gv.reshape(nb, -1), gl)
    ops.v_update_bwd_c128(vn.reshape(nb, -1), g.reshape(nb, -1), s, t, q, 0.01, True, gv.reshape(nb, -1), gl)
    ops.su3_projsu_vec8_bwd_(gx, F, gvec)
    ops.su3_projsu_vec8_bwd_(gx, xn, gvec)
    native.call('l2q_su3_force_bwd', xn, g, 6.0, gx, nb, *L)
    ops.scaled_tanh_bwd_sums(s, t, coeff, 1.0, bg, cg)
    ops.vnet_heads_vupdate_sliced_tape(z, image, b[0], cs, b[1], 1.0, b[2], cs, vn.reshape(nb, -1),
                                       F.reshape(nb, -1), 0.01, True)
    ops.v_update(vn.reshape(nb, -1), F.reshape(nb, -1), s, t, q, 0.01, True)
torch.cuda.synchronize()
print('kprof_train done')
