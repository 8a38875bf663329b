"""The cfg-3 (U(1) 64 x 64, 8192 chains, fp16 layers) kernels at their bench shapes, three launches each: the
target of tools/pmc_collect.sh (L2Q_KPROF_SCRIPT=tools/kprof_u1_cfg3.py) and of tools/kstats.sh."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
from l2hmc import _ops as ops  # noqa: E402

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
x = (torch.rand(m, n, device='cuda', generator=g) * 6.28 - 3.14)
v = torch.randn(m, n, device='cuda', generator=g)
f = torch.randn(m, n, device='cuda', generator=g)
wx = (torch.randn(k, n, device='cuda', generator=g) / n ** 0.5).to(hd)
wv = (torch.randn(k, n, device='cuda', generator=g) / n ** 0.5).to(hd)
wx2 = (torch.randn(k, 2 * n, device='cuda', generator=g) / n ** 0.5).to(hd)
bz = torch.zeros(k, device='cuda')
for _ in range(3):
    ops.u1_heads_update_h_(z, heads, 0.9, v, f, 0.01, True)                                   # v-update
    ops.u1_heads_update_h_(z, heads, 0.9, x, v, 0.01, True, mask=mask, complement=False)      # x-update
    ops.gemm_h(x, wx, bz, a2=f, w2=wv, bias2=bz, act='leaky_relu')                            # vnet input layer
    ops.gemm_h_u1x(x, mask, False, wx2, bz, v, wv, bz, 'leaky_relu')                          # xnet input layer
    ops.u1_force(x.reshape(m, 2, 64, 64), 6.0, (64, 64))
torch.cuda.synchronize()
print('kprof_u1_cfg3 done')
