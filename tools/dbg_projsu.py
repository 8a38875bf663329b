import os, sys, torch
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, os.path.join(R, 'l2hmc-qcd_amd')); sys.path.insert(0, os.path.join(R, 'tests'))
import emu_native as E
from l2hmc import native
g = torch.Generator().manual_seed(5)
c = lambda *s: torch.complex(torch.randn(*s, dtype=torch.float64, generator=g), torch.randn(*s, dtype=torch.float64, generator=g))
nf, V = 2, 64
m = c(nf, V, 3, 3)
mn = E._native(m); gvec = torch.randn(nf, 8, V, dtype=torch.float64, generator=g)
gm = torch.zeros(nf, 9, V, dtype=torch.complex128).cuda()
native.call('l2q_su3_projsu_vec8_bwd', mn.cuda(), gvec.cuda(), gm, nf, V)
ref = torch.zeros(nf, 9, V, dtype=torch.complex128)
E.call('l2q_su3_projsu_vec8_bwd', mn, gvec, ref, nf, V)
d = (gm.cpu() - ref).abs().amax(1)      # [nf, V]
print('per-link max err: min %.3e median %.3e max %.3e' % (float(d.min()), float(d.median()), float(d.max())))
print('fraction of links with err > 1e-9:', float((d > 1e-9).double().mean()))
# forward check of the vec8 kernel on same input
out = torch.empty(nf, 8, V, dtype=torch.float64).cuda()
native.call('l2q_su3_projsu_vec8', mn.cuda(), out, nf, V)
fw = E._to_vec8(E._proj_su(m)).transpose(-1, -2)
print('forward vec8 err', float((out.cpu() - fw).abs().max()))
bad = torch.nonzero(d > 1e-9)
for (f, s) in bad.tolist()[:8]:
    M = m[f, s]
    w, Vv = torch.linalg.eigh(M.conj().T @ M)
    U = M @ ((Vv * (1 / torch.sqrt(w))) @ Vv.conj().T)
    det = torch.linalg.det(U)
    print(f, s, 'err %.3e' % float(d[f, s]), 'eig', [round(float(x), 4) for x in w], 'arg det U %.4f' % float(torch.angle(det)),
          'rel err', float((gm.cpu()[f, :, s] - ref[f, :, s]).abs().max() / ref[f, :, s].abs().max()))
good = torch.nonzero(d <= 1e-9)[:5]
for (f, s) in good.tolist():
    M = m[f, s]; w, Vv = torch.linalg.eigh(M.conj().T @ M); U = M @ ((Vv * (1 / torch.sqrt(w))) @ Vv.conj().T)
    print('good', f, s, 'arg det U %.4f' % float(torch.angle(torch.linalg.det(U))))
print('lib g (1,4):', gm.cpu()[1, :, 4])
print('ref g (1,4):', ref[1, :, 4])
# run the kernel again on only field 1 (nf=1) and on a copy with V=128 padding
gm2 = torch.zeros(1, 9, V, dtype=torch.complex128).cuda()
native.call('l2q_su3_projsu_vec8_bwd', mn[1:2].contiguous().cuda(), gvec[1:2].contiguous().cuda(), gm2, 1, V)
print('nf=1 rerun err at link 4:', float((gm2.cpu()[0, :, 4] - ref[1, :, 4]).abs().max()))
