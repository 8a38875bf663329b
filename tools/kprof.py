#!/usr/bin/env python3
"""Run each hot kernel a few times at the cfg-4 size (or, with L2Q_KPROF_LATTICE="16 16 16 16"
L2Q_KPROF_NB=256, at the cfg-5 per-GPU shard) -- the target of rocprofv3 --pmc passes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
from l2hmc import _ops as ops, native  # noqa: E402

L = tuple(int(i) for i in os.environ.get('L2Q_KPROF_LATTICE', '8 8 8 8').split())
nb = int(os.environ.get('L2Q_KPROF_NB', 256))
V = L[0] * L[1] * L[2] * L[3]
torch.manual_seed(0)
xn = ops.su3_project_su_n(torch.randn(nb, 4, 9, V, dtype=torch.complex128, device='cuda'))
vn = ops.su3_assemble_tah_n(torch.randn(8, nb, 4, V, dtype=torch.float64, device='cuda'))
f = torch.empty_like(xn)
h = 256
z = torch.randn(nb, h, dtype=torch.float64, device='cuda')
N_ = 36 * V
heads = {k: (torch.randn(N_, h, dtype=torch.float64, device='cuda') / 16,
             torch.randn(N_, dtype=torch.float64, device='cuda'),
             None if k == 't' else torch.ones(N_, dtype=torch.float64, device='cuda')) for k in 'stq'}
sl = dict(heads)
sl['sliced'] = ops.heads_sliced_build(heads)      # int8 slice image (csrc/heads_sliced.hip)
K = 32 * V
mask = (torch.rand(36 * V, device='cuda') > 0.5).float()
xv = torch.randn(nb, K, dtype=torch.float64, device='cuda')
fv = torch.randn(nb, K, dtype=torch.float64, device='cuda')
wx = torch.randn(h, K, dtype=torch.float64, device='cuda') / K ** 0.5
wv = torch.randn(h, K, dtype=torch.float64, device='cuda') / K ** 0.5
bx = torch.randn(h, dtype=torch.float64, device='cuda')
# the int8-sliced input layer (csrc/gemm_sliced.hip): inputs bounded like su3_to_vec(projectSU(.)) (< 4)
img = None
if ops.gemm_sliced_pays(nb, h, K, K):
    img = (ops.gemm_sliced_build(wx), ops.gemm_sliced_build(wv))
    xvb, fvb = xv.clamp(-3.9, 3.9), fv.clamp(-3.9, 3.9)
for _ in range(3):
    ops.su3_plaq_sums_n(xn, L)
    native.call('l2q_su3_force_kick', xn, 6.0, -0.005, vn, nb, *L)
    ops.su3_expm_mul2_vec8_n(xn, vn, 0.01, mask, False)
    native.call('l2q_su3_force', xn, 6.0, f, nb, *L)
    ops.su3_expm_mul_n(xn, vn, 0.01)
    ops.su3_projsu_vec8_n(xn)
    ops.gemm(xv, wx, bx, a2=fv, w2=wv, bias2=bx, act='tanh')
    if img is not None:
        ops.gemm_sliced(xvb, img[0], h, bx, a2=fvb, image2=img[1], bias2=bx, act='tanh')
    ops.vnet_heads_vupdate_(z, heads, (1., 1., 1.), vn.reshape(nb, -1), f.reshape(nb, -1), 0.01, True)
    ops.vnet_heads_vupdate_pair_(z, heads, (1., 1., 1.), vn.reshape(nb, -1), f.reshape(nb, -1), 0.01, True, False, 0.01, True)
    ops.vnet_heads_vupdate_(z, sl, (1., 1., 1.), vn.reshape(nb, -1), f.reshape(nb, -1), 0.01, True)
    ops.vnet_heads_vupdate_pair_(z, sl, (1., 1., 1.), vn.reshape(nb, -1), f.reshape(nb, -1), 0.01, True, False, 0.01, True)
    ops.su3_expm_mul2_n(xn, vn, 0.01, mask, False)
# the two-x-plane force kernel (su3_force_pair.hip, tuning force_tile = 6), for its own traffic counters
if os.environ.get('L2Q_KPROF_PAIR', '1') == '1':
    native.set_tuning('force_tile', 6)
    for _ in range(3):
        native.call('l2q_su3_force', xn, 6.0, f, nb, *L)
        native.call('l2q_su3_force_kick', xn, 6.0, -0.005, vn, nb, *L)
    native.set_tuning('force_tile', 5)
torch.cuda.synchronize()
print('kprof done')
