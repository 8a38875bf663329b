"""x-update kernels at cfg-4 (l2q_su3_expm_mul2_vec8, l2q_su3_expm_mul); L2Q_LIB_NAME selects an A/B build"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'l2hmc-qcd_amd'))
from l2hmc import _ops as ops
nb, V = 256, 4096
torch.manual_seed(0)
xn = ops.su3_project_su_n(torch.randn(nb, 4, 9, V, dtype=torch.complex128, device='cuda'))
vn = ops.su3_assemble_tah_n(torch.randn(8, nb, 4, V, dtype=torch.float64, device='cuda'))
mask = (torch.rand(36 * V, device='cuda') > 0.5).float()


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


a = t(lambda: ops.su3_expm_mul2_vec8_n(xn, vn, 0.01, mask, False))
b = t(lambda: ops.su3_expm_mul_n(xn, vn, 0.01))
y = ops.su3_expm_mul_n(xn, vn, 0.01)
print(f'{os.environ.get("L2Q_LIB_NAME", "libl2q.so")}: expm_mul2_vec8 {a:.4f} ms  expm_mul {b:.4f} ms  checksum {float(y.real.sum()):.15e}', flush=True)
