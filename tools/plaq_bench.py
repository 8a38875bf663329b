import os, sys, torch
sys.path.insert(0, '/root/repo/l2hmc-qcd_amd')
from l2hmc import _ops as ops, native
def run(nb, L, reps=20):
    V = L[0]*L[1]*L[2]*L[3]
    torch.manual_seed(0)
    xn = ops.su3_project_su_n(torch.randn(nb, 4, 9, V, dtype=torch.complex128, device='cuda'))
    ref = None
    for sweep in (3, 2, 1, 0):
        native.set_tuning('plaq_sweep', sweep)
        name = native.kernel_name('l2q_su3_plaq_reduce', L)
        for _ in range(3): s = ops.su3_plaq_sums_n(xn, L)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): s = ops.su3_plaq_sums_n(xn, L)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        if ref is None: ref = s.clone()
        alg = nb * V * 576
        print(f'{"x".join(map(str, L))} x {nb}: plaq_sweep={sweep} {name:36s} {ms:.4f} ms {alg/ms/1e6:8.1f} GB/s frac {alg/ms/1e6/8000:.3f}  max|d| vs 3: {float((s-ref).abs().max()):.2e} (|s| {float(ref.abs().max()):.1e})', flush=True)
    native.set_tuning('plaq_sweep', 3)
run(256, (8,8,8,8))
run(64, (16,16,16,16), reps=5)
run(4, (4,8,8,8))
