#!/usr/bin/env python3
"""cfg-3 shape (M 8192 chains, N 8192 entries, K 256, fp16): the fused heads + update kernel against the
pieces of a two-kernel alternative (16-bit heads through the GEMM kernels, then a streaming update)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
from l2hmc import _ops as ops, native  # noqa: E402

M, N, K = 8192, 8192, 256
hd = torch.float16
torch.manual_seed(0)
dev = 'cuda'
z = (torch.randn(M, K, device=dev) * 0.5).to(hd)
W = {k: (torch.randn(N, K, device=dev) / 16).to(hd) for k in 'stq'}
b = {k: torch.randn(N, device=dev) * 0.1 for k in 'stq'}
cs = torch.ones(N, device=dev)
cq = torch.ones(N, device=dev)
heads = {'s': (W['s'], b['s'], cs), 't': (W['t'], b['t'], None), 'q': (W['q'], b['q'], cq)}
v = torch.randn(M, N, device=dev)
f = torch.randn(M, N, device=dev)
x = (torch.rand(M, N, device=dev) - 0.5) * 6.0
mask = (torch.rand(N, device=dev) > 0.5).float()


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print(f'fused heads + v-update      {timeit(lambda: ops.u1_heads_update_h_(z, heads, 1.0, v, f, 0.05, True)):.4f} ms')
print(f'fused heads + x-update(NCP) {timeit(lambda: ops.u1_heads_update_h_(z, heads, 1.0, x, v, 0.05, True, mask=mask)):.4f} ms')
for dma in (1, 0):
    native.set_tuning('gemm_h_dma', dma)
    print(f'gemm_h one head -> fp16, tanh (gemm_h_dma={dma})   '
          f'{timeit(lambda: ops.gemm_h(z, W["s"], b["s"], act="tanh")):.4f} ms')
    print(f'gemm_h one head -> fp32, tanh (gemm_h_dma={dma})   '
          f'{timeit(lambda: ops.gemm_h(z, W["s"], b["s"], act="tanh", out_dtype=torch.float32)):.4f} ms')
    Wc = torch.cat([W['s'], W['t'], W['q']], 0).contiguous()
    bc = torch.cat([b['s'], b['t'], b['q']], 0).contiguous()
    print(f'gemm_h three heads concatenated -> fp16 (gemm_h_dma={dma})   '
          f'{timeit(lambda: ops.gemm_h(z, Wc, bc, act="tanh")):.4f} ms')
native.set_tuning('gemm_h_dma', 1)
s32 = torch.randn(M, N, device=dev) * 0.1
print(f'l2q_v_update with fp32 s, t, q (1.6 GB)  {timeit(lambda: ops.v_update_(v, f, s32, s32, s32, 0.05, True)):.4f} ms')
print(f'l2q_u1_x_update with fp32 s, t, q        {timeit(lambda: ops.u1_x_update_(x, v, s32, s32, s32, mask, False, 0.05, True)):.4f} ms')
a16 = torch.randn(M, 3 * N, device=dev).to(hd)
out = torch.empty_like(v)
print(f'torch: read 3 fp16 + 2 fp32, write fp32 (1.2 GB; a streaming bound) '
      f'{timeit(lambda: torch.add(v, f, out=out).add_(a16[:, :N]).add_(a16[:, N:2*N]).add_(a16[:, 2*N:])):.4f} ms (4 passes)')
print(f'torch: v.add_(f) (805 MB)  {timeit(lambda: v.add_(f)):.4f} ms')
