#!/usr/bin/env python3
"""Micro-benchmark of the individual HIP kernels at the BASELINE cfg-4 size
(SU(3) 8^4, 256 chains, fp64).  Prints achieved algorithmic GB/s."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
from l2hmc import _ops as ops, native  # noqa: E402


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--nb', type=int, default=256)
    ap.add_argument('--L', type=int, nargs=4, default=[8, 8, 8, 8])
    ap.add_argument('--gemm', action='store_true')
    a = ap.parse_args()
    L, nb = tuple(a.L), a.nb
    V = L[0] * L[1] * L[2] * L[3]
    dev = 'cuda'
    torch.manual_seed(0)
    xn = torch.randn(nb, 4, 9, V, dtype=torch.complex128, device=dev)
    xn = ops.su3_project_su_n(xn)
    vn = ops.su3_assemble_tah_n(torch.randn(8, nb, 4, V, dtype=torch.float64, device=dev))
    sites = nb * V
    mask = (torch.rand(36 * V, device=dev) > 0.5).float()
    stq = [0.01 * torch.randn(nb, 36 * V, dtype=torch.float64, device=dev) for _ in range(3)]
    rows = []

    def rec(name, t, bytes_per_site, flop_per_site=0):
        gbs = sites * bytes_per_site / t / 1e9
        tf = sites * flop_per_site / t / 1e12
        rows.append((name, t * 1e3, gbs, tf))
        print(f'{name:34s} {t*1e3:8.3f} ms  {gbs:8.1f} GB/s  ({gbs/8000*100:5.1f}% of 8 TB/s)'
              + (f'  {tf:6.2f} TFLOP/s' if flop_per_site else ''), flush=True)

    for sweep in (0, 1, 2):
        native.set_tuning('plaq_sweep', sweep)
        for occ in (2, 3, 4):
            native.set_tuning('plaq_occ', occ)
            rec(f'su3_plaq_reduce sweep={sweep} occ={occ}', timeit(lambda: ops.su3_plaq_sums_n(xn, L)), 576, 2800)
    native.set_tuning('plaq_occ', 2)
    native.set_tuning('plaq_sweep', 2)
    native.set_tuning('xcd_swizzle', 0)
    rec('su3_plaq_reduce occ=2 noswz', timeit(lambda: ops.su3_plaq_sums_n(xn, L)), 576, 2800)
    native.set_tuning('xcd_swizzle', 1)
    f = torch.empty_like(xn)
    for tile in (0, 1, 2):
        native.set_tuning('force_tile', tile)
        for occ in ((2, 3) if tile < 2 else (2,)):
            native.set_tuning('force_occ', occ)
            rec(f'su3_force tile={tile} occ={occ}', timeit(lambda: native.call(
                'l2q_su3_force', xn, 6.0, f, nb, *L)), 1152, 11200)
    native.set_tuning('force_occ', 2)
    native.set_tuning('force_tile', 1)
    native.set_tuning('xcd_swizzle', 0)
    rec('su3_force occ=2 noswz', timeit(lambda: native.call(
        'l2q_su3_force', xn, 6.0, f, nb, *L)), 1152, 11200)
    native.set_tuning('xcd_swizzle', 1)
    rec('su3_force_kick', timeit(lambda: ops.su3_force_kick_n(xn, 6.0, -0.005, vn, L)), 576 * 3, 11200)
    out = torch.empty_like(xn)
    rec('su3_expm_mul (no mask)', timeit(lambda: ops.su3_expm_mul_n(xn, vn, 0.01, out=out)), 1728)
    rec('su3_expm_mul (masked)', timeit(lambda: ops.su3_expm_mul_n(xn, vn, 0.01, mask, False, out=out)), 1728)
    rec('su3_projsu_vec8', timeit(lambda: ops.su3_projsu_vec8_n(xn)), 832)
    rec('su3_project_su', timeit(lambda: ops.su3_project_su_n(xn)), 1152)
    rec('su3_kinetic_reduce', timeit(lambda: ops.su3_kinetic_n(vn)), 576)
    v2 = vn.clone()
    rec('v_update (complex)', timeit(lambda: ops.v_update_(v2, f, *stq, 0.01, True)), 2592)
    rec('su3_pack', timeit(lambda: ops.su3_pack(xn.reshape(nb, -1))), 1152)
    nrm = torch.randn(8, nb, 4, V, dtype=torch.float64, device=dev)
    rec('su3_assemble_tah', timeit(lambda: ops.su3_assemble_tah_n(nrm)), 4 * (64 + 144))
    # reference points: device copy bandwidth
    a_ = torch.empty(2 ** 28, dtype=torch.float32, device=dev); b_ = torch.empty_like(a_)
    t = timeit(lambda: b_.copy_(a_))
    print(f'{"torch copy 1 GiB":34s} {t*1e3:8.3f} ms  {2*a_.numel()*4/t/1e9:8.1f} GB/s', flush=True)
    if a.gemm:
        h = 256
        K = 32 * V
        xv = torch.randn(nb, K, dtype=torch.float64, device=dev)
        fv = torch.randn(nb, K, dtype=torch.float64, device=dev)
        wx = torch.randn(h, K, dtype=torch.float64, device=dev) / K ** 0.5
        wv = torch.randn(h, K, dtype=torch.float64, device=dev) / K ** 0.5
        bx = torch.randn(h, dtype=torch.float64, device=dev)
        t = timeit(lambda: ops.gemm(xv, wx, bx, a2=fv, w2=wv, bias2=bx, act='tanh'), iters=5, warm=2)
        fl = 2.0 * nb * h * 2 * K
        print(f'{"gemm in  [nb,2*32V]x[h,..]":34s} {t*1e3:8.3f} ms  {fl/t/1e12:7.2f} TFLOP/s fp64', flush=True)
        t = timeit(lambda: torch.addmm(bx, xv, wx.t()), iters=5, warm=2)
        print(f'{"  (rocBLAS addmm, half of it)":34s} {t*1e3:8.3f} ms  {fl/2/t/1e12:7.2f} TFLOP/s fp64', flush=True)
        z = torch.randn(nb, h, dtype=torch.float64, device=dev)
        N_ = 36 * V
        ws_ = torch.randn(N_, h, dtype=torch.float64, device=dev) / h ** 0.5
        bs = torch.randn(N_, dtype=torch.float64, device=dev)
        co = torch.zeros(N_, dtype=torch.float64, device=dev)
        t = timeit(lambda: ops.gemm(z, ws_, bs, coeff=co, act='tanh'), iters=5, warm=2)
        fl = 2.0 * nb * h * N_
        print(f'{"gemm head [nb,h]x[36V,h]":34s} {t*1e3:8.3f} ms  {fl/t/1e12:7.2f} TFLOP/s fp64', flush=True)
        t = timeit(lambda: torch.addmm(bs, z, ws_.t()), iters=5, warm=2)
        print(f'{"  (rocBLAS addmm)":34s} {t*1e3:8.3f} ms  {fl/t/1e12:7.2f} TFLOP/s fp64', flush=True)
        heads = {'s': (ws_, bs, co.exp()), 't': (ws_.clone(), bs, None), 'q': (ws_.clone(), bs, co.exp())}
        vv = vn.reshape(nb, -1).clone(); ff = f.reshape(nb, -1)
        t = timeit(lambda: ops.vnet_heads_vupdate_(z, heads, (1., 1., 1.), vv, ff, 0.01, True), iters=5, warm=2)
        print(f'{"fused 3 heads + v_update":34s} {t*1e3:8.3f} ms  {3*fl/t/1e12:7.2f} TFLOP/s fp64', flush=True)


if __name__ == '__main__':
    main()
