#!/usr/bin/env python3
"""Per-shape timing of the backward GEMMs of one cfg-4 vnet call (LeapfrogLayer.backward -> _linear_bwd ->
l2q_gemm_ex): which of the four shapes holds the 0.56-of-peak average."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from l2hmc import _ops as ops  # noqa: E402
from kbench import timeit  # noqa: E402

nb, h = 256, 256
NH, NI = 36 * 4096, 32 * 4096
R = lambda *s: torch.randn(*s, dtype=torch.float64, device='cuda')
dpre_h, z, Wh = R(nb, NH), R(nb, h), R(NH, h) / 16
gWh = torch.zeros(NH, h, dtype=torch.float64, device='cuda')
dpre_i, xf, Wx = R(nb, h), R(nb, NI), R(h, NI) / 64
gWx = torch.zeros(h, NI, dtype=torch.float64, device='cuda')
cases = [
    ('heads dW += dpre^T z      [147456 x 256], K = 256 chains', 2.0 * NH * h * nb,
     lambda: ops.gemm_ex(dpre_h, z, a_trans=True, w_trans=True, out=gWh, accumulate=True)),
    ('heads dz  = dpre W        [256 x 256],    K = 147456    ', 2.0 * nb * h * NH,
     lambda: ops.gemm_ex(dpre_h, Wh, w_trans=True)),
    ('input dW += dpre^T x      [256 x 131072], K = 256 chains', 2.0 * h * NI * nb,
     lambda: ops.gemm_ex(dpre_i, xf, a_trans=True, w_trans=True, out=gWx, accumulate=True)),
    ('input dx  = dpre W        [256 x 131072], K = 256 units ', 2.0 * nb * NI * h,
     lambda: ops.gemm_ex(dpre_i, Wx, w_trans=True)),
]
for name, fl, fn in cases:
    t = timeit(fn, iters=10, warm=3)
    print(f'{name}: {t * 1e3:7.3f} ms  {fl / t / 1e12:6.2f} TFLOP/s  ({fl / t / 78.6e12:.2f} of the fp64 MFMA peak)', flush=True)

# the nine calls of a step as ONE GEMM over K = 9 x 256 chains (cotangents and activations kept until the end of
# the reverse sweep): what deferring the weight-gradient GEMMs would buy
K9 = 9 * nb
dpre9, z9 = R(K9, NH), R(K9, h)
t = timeit(lambda: ops.gemm_ex(dpre9, z9, a_trans=True, w_trans=True, out=gWh, accumulate=True), iters=5, warm=2)
fl = 2.0 * NH * h * K9
print(f'heads dW, K = {K9}: {t * 1e3:7.3f} ms = {t * 1e3 / 9:.3f} per call  {fl / t / 1e12:6.2f} TFLOP/s', flush=True)
del dpre9
dpi9, xf9 = R(K9, h), R(K9, NI)
t = timeit(lambda: ops.gemm_ex(dpi9, xf9, a_trans=True, w_trans=True, out=gWx, accumulate=True), iters=5, warm=2)
fl = 2.0 * h * NI * K9
print(f'input dW, K = {K9}: {t * 1e3:7.3f} ms = {t * 1e3 / 9:.3f} per call  {fl / t / 1e12:6.2f} TFLOP/s', flush=True)
