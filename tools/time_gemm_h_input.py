"""cfg-3 input layers alone: vnet (x, F fp32 [8192, 8192] each -> 256) and xnet (cos / sin of x formed in
the loader + v), per value of the `gemm_h_skinny` tuning (0: register-staged 128 x 256 tile, 1: streaming
kernel with its own split count, 2 / 4 / 8: that split count).  Interleaved rounds, median."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
from l2hmc import _ops as ops, native  # noqa: E402

m, n, k = 8192, 256, 8192
hd = torch.float16
x = torch.rand(m, k, device='cuda')
f = torch.randn(m, k, device='cuda')
wx = (torch.randn(n, k, device='cuda') / k ** 0.5).to(hd)
wv = (torch.randn(n, k, device='cuda') / k ** 0.5).to(hd)
wx2 = (torch.randn(n, 2 * k, device='cuda') / k ** 0.5).to(hd)
b = torch.zeros(n, device='cuda')
mask = (torch.rand(k, device='cuda') < 0.5).float()


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


variants = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4, 8]
res = {(v, w): [] for v in variants for w in 'vx'}
for rnd in range(5):
    for v in variants:
        native.set_tuning('gemm_h_skinny', v)
        res[(v, 'v')].append(timeit(lambda: ops.gemm_h(x, wx, b, a2=f, w2=wv, bias2=b, act='leaky_relu')))
        res[(v, 'x')].append(timeit(lambda: ops.gemm_h_u1x(x, mask, False, wx2, b, f, wv, b, 'leaky_relu')))
tag = os.environ.get('L2Q_LIB_NAME', 'libl2q.so')
for v in variants:
    tv = sorted(res[(v, 'v')])[2]
    tx = sorted(res[(v, 'x')])[2]
    print(f'[{tag} skinny={v}] vnet input {tv:7.1f} us ({2 * m * k * 4 / tv * 1e-6:.2f} TB/s)   '
          f'xnet input {tx:7.1f} us ({2 * m * k * 4 / tx * 1e-6:.2f} TB/s)')
