"""int8-sliced heads kernel against the fp64 MFMA kernel and a long-double reference (GPU box)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'l2hmc-qcd_amd'))
from l2hmc import _ops as ops, native as N

dev = 'cuda'
torch.manual_seed(0)


sys.path.insert(0, os.path.dirname(__file__))
from probe_heads_sliced_lib import make


def ref_ld(z, heads, v, f, eps, fwd):
    """long-double reference of one update (numpy, CPU)"""
    L = np.longdouble
    zz = z.cpu().numpy().astype(L)
    out = {}
    for nm in 'stq':
        w, b, c = heads[nm]
        y = zz @ w.cpu().numpy().astype(L).T + b.cpu().numpy().astype(L)
        out[nm] = y
    s = heads['s'][2].cpu().numpy().astype(L) * np.tanh(out['s'])
    q = heads['q'][2].cpu().numpy().astype(L) * np.tanh(out['q'])
    t = out['t']
    vv = v.cpu().numpy()
    ff = f.cpu().numpy()
    vr, vi = vv.real.astype(L), vv.imag.astype(L)
    fr, fi = ff.real.astype(L), ff.imag.astype(L)
    h = L(0.5) * L(eps)
    if fwd:
        es, eq = np.exp(h * s), np.exp(L(eps) * q)
        vr = es * vr - h * (fr * eq + t); vi = es * vi - h * (fi * eq)
        ld = (h * s).sum(1)
    else:
        es, eq = np.exp(-h * s), np.exp(L(eps) * q)
        vr = es * (vr + h * (fr * eq + t)); vi = es * (vi + h * (fi * eq))
        ld = (-h * s).sum(1)
    return vr, vi, ld


def run(z, heads, v, f, sliced, kind, eps=0.1, fwd=True):
    ops.USE_SLICED_HEADS[0] = sliced
    vv = v.clone()
    if kind == 'single':
        ld = ops.vnet_heads_vupdate_(z, heads, (1.0, 1.0, 1.0), vv, f, eps, fwd)
        return vv, ld
    if kind == 'to':
        out = torch.empty_like(v)
        ld = ops.vnet_heads_vupdate_(z, heads, (1.0, 1.0, 1.0), out, f, eps, fwd, v)
        return out, ld
    if kind == 'pair':
        ld = ops.vnet_heads_vupdate_pair_(z, heads, (1.0, 1.0, 1.0), vv, f, eps, fwd, True, 0.07, not fwd)
        return vv, ld
    ld, ld1, ke = ops.vnet_heads_vupdate_pair_mid_(z, heads, (1.0, 1.0, 1.0), vv, f, eps, fwd, False, eps, fwd)
    return vv, torch.stack([ld, ld1, ke])


# ---- accuracy against long double (small)
for cplx in (True, False):
    for (m, n) in ((64, 48), (37, 50), (130, 16)):
        z, heads, v, f = make(m, n, cplx=cplx)
        heads['sliced'] = ops.heads_sliced_build(heads)
        assert heads['sliced'] is not None
        for fwd in (True, False):
            vr, vi, ldr = ref_ld(z, heads, v, f, 0.1, fwd)
            res = {}
            for sl in (False, True):
                o, ld = run(z, heads, v, f, sl, 'single', 0.1, fwd)
                oc = o.cpu().numpy()
                e = max(float(np.abs(oc.real.astype(np.longdouble) - vr).max()),
                        float(np.abs(oc.imag.astype(np.longdouble) - vi).max()) if cplx else 0.0)
                el = float(np.abs(ld.cpu().numpy().astype(np.longdouble) - ldr).max())
                res[sl] = (e, el)
            print(f'cplx {cplx} M {m} N {n} fwd {fwd}: |dv| vs long double  fp64 {res[False][0]:.2e} sliced {res[True][0]:.2e}'
                  f'   |dlogdet| fp64 {res[False][1]:.2e} sliced {res[True][1]:.2e}', flush=True)

# ---- variants, sliced vs fp64
z, heads, v, f = make(200, 1000)
heads['sliced'] = ops.heads_sliced_build(heads)
for kind in ('single', 'to', 'pair', 'mid'):
    a, la = run(z, heads, v, f, False, kind)
    b, lb = run(z, heads, v, f, True, kind)
    print(f'{kind}: max|dv| {float((a - b).abs().max()):.2e}  max|d aux| {float((la - lb).abs().max()):.2e} (|aux| {float(la.abs().max()):.2e})', flush=True)

# ---- ill-conditioned weights are refused
zb, hb, _, _ = make(64, 64)
hb['s'][0][5, :] *= 1e-9
hb['s'][0][5, 7] = 1.0
print('ill-conditioned column refused:', ops.heads_sliced_build(hb) is None, flush=True)

# ---- cfg-4 size timing
for (m, n) in ((256, 147456), (128, 147456), (2048, 2048 * 9)):
    z, heads, v, f = make(m, n)
    t0 = time.time()
    heads['sliced'] = ops.heads_sliced_build(heads)
    torch.cuda.synchronize()
    tb = time.time() - t0
    for kind in ('single', 'pair'):
        for sl in (False, True):
            for _ in range(3):
                run(z, heads, v, f, sl, kind)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ops.USE_SLICED_HEADS[0] = sl
            vv = v.clone()
            e0.record()
            for _ in range(10):
                if kind == 'single':
                    ops.vnet_heads_vupdate_(z, heads, (1.0, 1.0, 1.0), vv, f, 1e-3, True)
                else:
                    ops.vnet_heads_vupdate_pair_(z, heads, (1.0, 1.0, 1.0), vv, f, 1e-3, True, False, 1e-3, True)
            e1.record(); torch.cuda.synchronize()
            print(f'M {m} N {n} {kind} sliced {sl}: {e0.elapsed_time(e1) / 10:.4f} ms (incl. split of Z + finalize)', flush=True)
    a, la = run(z, heads, v, f, False, 'pair')
    b, lb = run(z, heads, v, f, True, 'pair')
    print(f'  build {tb * 1e3:.1f} ms; pair max|dv| {float((a - b).abs().max()):.2e} max|dlogdet| {float((la - lb).abs().max()):.2e} of {float(la.abs().max()):.2e}', flush=True)
