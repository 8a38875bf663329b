#!/usr/bin/env python3
"""Time the layers of the default U(1) conv stack at the cfg-3 shape on the 16-bit path."""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
from l2hmc import _ops as ops, native  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument('--nb', type=int, default=8192)
ap.add_argument('--L', type=int, default=64)
ap.add_argument('--tune', nargs=2, action='append', default=[])
a = ap.parse_args()
for k_, v_ in a.tune:
    assert native.set_tuning(k_, int(v_)) >= 0
hd = torch.float16


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


H = a.L
x = torch.randn(a.nb, 4, H, H, device='cuda')
layout, cin = 'nchw', 4
for (cout, k, pool) in ((8, 5, 1), (16, 3, 1), (32, 3, 2), (64, 3, 1), (128, 2, 2)):
    w = torch.randn(cout, cin, k, k, device='cuda') / (cin * k * k) ** 0.5
    w16 = (w if layout == 'nchw' else w.permute(0, 2, 3, 1)).to(hd).contiguous()
    b = torch.zeros(cout, device='cuda')
    t = timeit(lambda: ops.conv2d_periodic_gemm_h(x, layout, w16, b, 1, 'leaky_relu'))
    Ho = H + k - 1
    gf = 2 * a.nb * Ho * Ho * cin * k * k * cout / 1e9
    inb = x.numel() * x.element_size() / 1e6
    outb = a.nb * Ho * Ho * cout * 2 / 1e6
    print(f'conv {cin:3d}->{cout:3d} k={k} on {H}x{H}: {t:7.3f} ms  {gf / t:8.1f} GFLOP/ms  in {inb:.0f} MB out {outb:.0f} MB')
    y = ops.conv2d_periodic_gemm_h(x, layout, w16, b, pool, 'leaky_relu')
    if pool > 1:
        yy = ops.conv2d_periodic_gemm_h(x, layout, w16, b, 1, None)
        t = timeit(lambda: native.call('l2q_maxpool_act_nhwc_h', 0, yy.reshape(-1, cout), a.nb, Ho, Ho, cout, pool, 3,
                                       torch.empty_like(y)))
        print(f'   maxpool {pool}: {t:7.3f} ms')
        del yy
    x, layout, cin, H = y, 'nhwc', cout, y.shape[1]
