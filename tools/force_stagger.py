#!/usr/bin/env python3
"""force_tile = 5 (su3_force_link.hip) with the second resident workgroup set delayed by
force_stagger x ~2k cycles: are the two workgroups of a CU in lock-step?"""
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
    native.set_tuning('force_tile', 5)
    for rnd in range(2):
        for stg in (0, 1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 24):
            native.set_tuning('force_stagger', stg)
            out = []
            for kick in (False, True):
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
                out.append(e0.elapsed_time(e1) / reps)
            print(f'{"x".join(map(str, L))} x {nb}: round {rnd} force_stagger={stg:2d}  force {out[0]:.4f} ms  '
                  f'kick {out[1]:.4f} ms', flush=True)
    native.set_tuning('force_stagger', 0)


if __name__ == '__main__':
    run(256, (8, 8, 8, 8))
    run(64, (16, 16, 16, 16), reps=5)
