#!/usr/bin/env python3
"""Time l2q_u1_heads_update_h / l2q_gemm_h alone at the cfg-3 shape (8192 chains, 64x64, units 256)."""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
from l2hmc import _ops as ops, native  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument('--m', type=int, default=8192)
ap.add_argument('--n', type=int, default=8192)
ap.add_argument('--k', type=int, default=256)
ap.add_argument('--tune', nargs=2, action='append', default=[])
a = ap.parse_args()
for k_, v_ in a.tune:
    assert native.set_tuning(k_, int(v_)) >= 0
hd = torch.float16
z = torch.randn(a.m, a.k, device='cuda').to(hd)
heads = {}
for nm in 'stq':
    heads[nm] = ((torch.randn(a.n, a.k, device='cuda') / a.k ** 0.5).to(hd),
                 torch.zeros(a.n, device='cuda'), torch.ones(a.n, device='cuda'))
v = torch.randn(a.m, a.n, device='cuda'); f = torch.randn(a.m, a.n, device='cuda')
x = torch.rand(a.m, a.n, device='cuda'); mask = (torch.rand(a.n, device='cuda') < 0.5).float()


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


gf = 3 * 2 * a.m * a.n * a.k / 1e9
t = timeit(lambda: ops.u1_heads_update_h_(z, heads, 1.0, v, f, 0.1, True))
print(f'heads+v-update: {t:.1f} us  {gf / t * 1e-3:.1f} TFLOP/s  {3 * a.m * a.n * 4 / t * 1e-6:.2f} TB/s')
t = timeit(lambda: ops.u1_heads_update_h_(z, heads, 1.0, x, v, 0.1, True, mask=mask))
print(f'heads+x-update: {t:.1f} us  {gf / t * 1e-3:.1f} TFLOP/s')
t = timeit(lambda: ops.gemm_h(z, heads['s'][0], heads['s'][1], coeff=heads['s'][2], act='tanh', out_dtype=torch.float32))
print(f'one head gemm_h (fp32 out): {t:.1f} us  {gf / 3 / t * 1e-3:.1f} TFLOP/s')
t = timeit(lambda: ops.gemm_h(z, heads['s'][0], heads['s'][1], act='tanh'))
print(f'one head gemm_h (16-bit out): {t:.1f} us  {gf / 3 / t * 1e-3:.1f} TFLOP/s')

if a.m >= 4096:
    K2 = 51200
    zz = torch.randn(a.m, K2, device='cuda').to(hd)
    ww = (torch.randn(a.n, K2, device='cuda') / K2 ** 0.5).to(hd)
    bb = torch.zeros(a.n, device='cuda')
    t = timeit(lambda: ops.gemm_h(zz, ww, bb, act='leaky_relu', out_dtype=torch.float32), n=3)
    print(f'conv-stack Linear {a.m}x{a.n}x{K2} gemm_h: {t:.1f} us  {2 * a.m * a.n * K2 / t * 1e-6:.1f} TFLOP/s')
