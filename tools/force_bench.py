#!/usr/bin/env python3
"""Force-kernel A/B on one box: every force_tile variant at the bench.py shape (and the cfg-5
per-GPU shape with --big), timed with HIP events; results compared with variant 2."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
from l2hmc import _ops as ops, native  # noqa: E402


def run(nb, L, reps=20):
    V = L[0] * L[1] * L[2] * L[3]
    torch.manual_seed(0)
    xn = ops.su3_project_su_n(torch.randn(nb, 4, 9, V, dtype=torch.complex128, device='cuda'))
    vn = ops.su3_assemble_tah_n(torch.randn(8, nb, 4, V, dtype=torch.float64, device='cuda'))
    f = torch.empty_like(xn)
    ref = None
    for tile in ((2, 7, 5) if '--plaq' in sys.argv else (2, 7, 6, 5, 4, 3, 1, 0)):
        native.set_tuning('force_tile', tile)
        for kick in ((False,) if tile == 7 else (False, True)):
            name = native.kernel_name('l2q_su3_force_kick' if kick else 'l2q_su3_force', L)

            def go():
                if kick:
                    native.call('l2q_su3_force_kick', xn, 6.0, -0.005, vn, nb, *L)
                else:
                    native.call('l2q_su3_force', xn, 6.0, f, nb, *L)
            for _ in range(3):
                go()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                go()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            alg = nb * V * (1728 if kick else 1152)
            extra = ''
            if not kick:
                if ref is None:
                    ref = f.clone()
                extra = f'  max|dF| vs variant 2: {float((f - ref).abs().max()):.2e}'
            print(f'{"x".join(map(str, L))} x {nb}: force_tile={tile} {name:45s} {ms:.4f} ms '
                  f'{alg / ms / 1e6:8.1f} GB/s  frac {alg / ms / 1e6 / 8000:.3f}{extra}', flush=True)
    native.set_tuning('force_tile', 5)


if __name__ == '__main__':
    run(256, (8, 8, 8, 8))
    if '--quick' in sys.argv:
        sys.exit(0)
    if '--big' in sys.argv:
        run(64, (16, 16, 16, 16), reps=5)
