"""TEST INFRASTRUCTURE ONLY -- a torch-CPU stand-in for the libl2q.so entry points used by the
U(1) sampling + training path.

Purpose: (1) run the *host orchestration* of the training step (tape, reverse sweep, arena,
Adam) on the CPU-only build container against the reference's gradient fixtures, and
(2) serve as the checker for each backward kernel on the GPU (``tests/test_train_gpu.py``).
Forward entry points are restated with plain torch ops; every backward entry point is obtained
by ``torch.autograd`` of that restatement -- nothing here shares code or derivations with the
HIP kernels.  The product never imports this module (``native.call`` raises on CPU tensors);
tests install it with ``emu_native.install(monkeypatch)``.
"""
from __future__ import annotations

import math

import torch

ACT = {0: None, 1: 'tanh', 2: 'relu', 3: 'leaky_relu', 4: 'elu', 5: 'swish'}


def _act(z, act):
    a = ACT[act] if isinstance(act, int) else act
    if a is None:
        return z
    if a == 'tanh':
        return torch.tanh(z)
    if a == 'relu':
        return torch.relu(z)
    if a == 'leaky_relu':
        return torch.nn.functional.leaky_relu(z, 0.01)
    if a == 'elu':
        return torch.nn.functional.elu(z)
    if a == 'swish':
        return torch.nn.functional.silu(z)
    raise ValueError(a)


def _wrap(x):
    return torch.remainder(x + math.pi, 2 * math.pi) - math.pi


def _theta(x):                                   # x [nb, 2, T, X]
    return x[:, 0] + torch.roll(x[:, 1], -1, 1) - torch.roll(x[:, 0], -1, 2) - x[:, 1]


def _force(x, beta):
    s = torch.sin(_theta(x))
    return torch.stack([beta * (s - torch.roll(s, 1, 2)), beta * (-s + torch.roll(s, 1, 1))], 1)


def _keep(mask, complement, like):
    k = mask.to(like.dtype).reshape(1, -1)
    return (1 - k) if complement else k


def _x_update(x, v, s, t, q, mask, complement, eps, forward, ncp):
    """x, v, s, t, q: [nb, n] -> (x', logdet)"""
    keep = _keep(mask, complement, x)
    mb = 1 - keep
    S = eps * s if forward else -eps * s
    es, eq = torch.exp(S), torch.exp(eps * q)
    tr = v * eq + t
    if ncp:
        h = x / 2
        x1 = 2 * torch.atan(torch.tan(h) * es)
        xp = x1 + eps * tr if forward else x1 - es * eps * tr
        ld = torch.log(es / (torch.cos(h) ** 2 + (es * torch.sin(h)) ** 2))
    else:
        xp = x * es + eps * tr if forward else es * (x - eps * tr)
        ld = S
    return _wrap(keep * x + mb * xp), (mb * ld).sum(1)


def _v_update(v, f, s, t, q, eps, forward):
    S = 0.5 * eps * s if forward else -0.5 * eps * s
    es, eq = torch.exp(S), torch.exp(eps * q)
    fq = f * eq + t
    vn = es * v - 0.5 * eps * fq if forward else es * (v + 0.5 * eps * fq)
    return vn, S.sum(1)


def _im2col(x4, k):
    """x4 [nb, C, H, W] -> col [nb*Ho*Wo, C*k*k] with periodic padding k-1 on both sides."""
    nb, C, H, W = x4.shape
    p = k - 1
    xp = torch.cat([x4[:, :, -p:, :], x4, x4[:, :, :p, :]], 2) if p > 0 else x4
    xp = torch.cat([xp[:, :, :, -p:], xp, xp[:, :, :, :p]], 3) if p > 0 else xp
    cols = torch.nn.functional.unfold(xp, k)                   # [nb, C*k*k, Ho*Wo]
    return cols.transpose(1, 2).reshape(-1, C * k * k)


def _as_nchw(x, sn, sc, sh, sw, nb, C, H, W):
    if sw == 1:
        return x.reshape(nb, C, H, W)
    return x.reshape(nb, H, W, C).permute(0, 3, 1, 2)


def _maxpool_act(y, nb, H, W, C, pool, act):
    y4 = y.reshape(nb, H, W, C).permute(0, 3, 1, 2)
    o = torch.nn.functional.max_pool2d(y4, pool)
    return _act(o, act).permute(0, 2, 3, 1).contiguous()


def _vjp(fn, inputs, cotangents):
    """fn(*inputs) -> tuple of outputs; returns grads wrt inputs (zeros where unused)."""
    ins = [i.detach().clone().requires_grad_(True) for i in inputs]
    with torch.enable_grad():
        outs = fn(*ins)
    if isinstance(outs, torch.Tensor):
        outs = (outs,)
    pairs = [(o, c) for o, c in zip(outs, cotangents) if c is not None]
    grads = torch.autograd.grad([o for o, _ in pairs], ins, [c.reshape(o.shape) for o, c in pairs],
                                allow_unused=True)
    return [torch.zeros_like(i) if g is None else g for g, i in zip(grads, ins)]


# ---------------------------------------------------------------------------- entry points
def l2q_transpose(a, out, batch, rows, cols, esz):
    out.copy_(a.reshape(batch, rows, cols).transpose(1, 2).reshape(out.shape))


def _gemm(A, W, M, N, K, A2, W2, K2, bias, bias2, coeff, scale, act, C, ws, wsn):
    z = A.reshape(M, K) @ W.reshape(N, K).T
    if A2 is not None:
        z = z + A2.reshape(M, K2) @ W2.reshape(N, K2).T
    if bias is not None:
        z = z + bias
    if bias2 is not None:
        z = z + bias2
    y = _act(z, act)
    y = scale * torch.exp(coeff) * y if coeff is not None else scale * y
    C.copy_(y.reshape(C.shape))


l2q_gemm_f32 = _gemm
l2q_gemm_f64 = _gemm


def l2q_gemm_ex(A, ta, W, tw, M, N, K, esz, accumulate, C, ws, wsn):
    a = A.reshape(K, M).T if ta else A.reshape(M, K)
    w = W.reshape(K, N).T if tw else W.reshape(N, K)
    y = (a @ w.T).reshape(C.shape)
    if accumulate:
        C.add_(y)
    else:
        C.copy_(y)


def l2q_gemm_h(ht, A, a32, W, M, N, K, A2, W2, K2, bias, bias2, coeff, scale, act, C, c32, ws, wsn):
    """autocast rounding points of include/l2q.h: l2q_gemm_h, products accumulated in fp32."""
    hd = torch.float16 if ht == 0 else torch.bfloat16
    r16 = lambda t: t.to(hd).float()
    z = r16(A.reshape(M, K).float()) @ W.reshape(N, K).float().T
    if A2 is not None:
        z = z + r16(A2.reshape(M, K2).float()) @ W2.reshape(N, K2).float().T
    if bias is not None:
        z = z + bias
    if bias2 is not None:
        z = z + bias2
    y = r16(z)
    if act:
        y = r16(_act(y, act))
    y = scale * torch.exp(coeff) * y if coeff is not None else r16(scale * y)
    C.copy_(y.reshape(C.shape).to(C.dtype))


def l2q_u1_plaq_reduce(x, nb, T, X, esz, out):
    th = _theta(x.reshape(nb, 2, T, X))
    out[:, 0] = torch.cos(th).sum((1, 2))
    out[:, 1] = torch.sin(th).sum((1, 2))
    out[:, 2] = (th - 2 * math.pi * torch.floor((th + math.pi) / (2 * math.pi))).sum((1, 2))


def l2q_u1_force(x, beta, force, v, coef, nb, T, X, esz):
    f = _force(x.reshape(nb, 2, T, X), beta)
    if force is not None:
        force.copy_(f.reshape(force.shape))
    if v is not None:
        v.add_(coef * f.reshape(v.shape))


def l2q_u1_x_update(x, v, s, t, q, mask, complement, eps, forward, ncp, esz, nb, n, logdet):
    xn, ld = _x_update(x.reshape(nb, n), v.reshape(nb, n), s, t, q, mask, complement, eps,
                       bool(forward), bool(ncp))
    x.copy_(xn.reshape(x.shape))
    logdet.copy_(ld)


def l2q_v_update(v, force, s, t, q, eps, forward, cplx, esz, nb, n, logdet, ws, wsn):
    assert not cplx
    vn, ld = _v_update(v.reshape(nb, n), force.reshape(nb, n), s, t, q, eps, bool(forward))
    v.copy_(vn.reshape(v.shape))
    logdet.copy_(ld)


def l2q_u1_wrap(x, y, n, esz):
    y.copy_(_wrap(x))


def l2q_u1_kinetic_reduce(v, nb, n, esz, out):
    out.copy_(0.5 * (v.reshape(nb, n) ** 2).sum(1))


def l2q_axpy(p, alpha, x, n, esz):
    x.add_(alpha * p.reshape(x.shape))


def l2q_u1_masked_cos_sin(x, mask, complement, out, nb, n, esz):
    a = _keep(mask, complement, x) * x.reshape(nb, n)
    out.copy_(torch.cat([torch.cos(a), torch.sin(a)], 1).reshape(out.shape))


def l2q_accept(h_init, h_prop, sld, u, acc, mask, nb, esz):
    a = torch.exp(torch.minimum(h_init - h_prop + sld, torch.zeros_like(h_init)))
    acc.copy_(a)
    mask.copy_((a > u).float())


def l2q_select_rows(a, b, mask, out, nb, row_bytes):
    m = mask.reshape(nb, *([1] * (a.dim() - 1))).bool()
    out.copy_(torch.where(m, a, b))


def _clast_cols(col, C, k, to_clast):
    """reorder the K columns between (ci, i, j) and (i, j, ci)"""
    M = col.shape[0]
    if to_clast:
        return col.reshape(M, C, k, k).permute(0, 2, 3, 1).reshape(M, -1)
    return col.reshape(M, k, k, C).permute(0, 3, 1, 2).reshape(M, -1)


def l2q_im2col_periodic_f32(x, sn, sc, sh, sw, nb, C, H, W, k, clast, col):
    c = _im2col(_as_nchw(x, sn, sc, sh, sw, nb, C, H, W), k)
    col.copy_(_clast_cols(c, C, k, True) if clast else c)


def l2q_conv_gemm_periodic_f32(x, sn, sc, sh, sw, nb, C, H, W, k, w, clast, b, cout, act, out):
    col = _im2col(_as_nchw(x, sn, sc, sh, sw, nb, C, H, W), k)
    w = w.reshape(cout, k, k, C).permute(0, 3, 1, 2) if clast else w.reshape(cout, C, k, k)
    out.copy_(_act(col @ w.reshape(cout, -1).T + b, act).reshape(out.shape))


def l2q_maxpool_act_nhwc_f32(y, nb, H, W, C, pool, act, out):
    out.copy_(_maxpool_act(y, nb, H, W, C, pool, act))


def _fused_net(xin, vin, wxT, wvT, b0, hidden, units, nl, ws, bs, cs, wt, bt, scale_t, wq, bq, cq,
               act):
    units = [int(units[i]) for i in range(nl)]
    z = _act(xin @ wxT + vin @ wvT + b0, act)
    off = 0
    for l in range(1, nl):
        ui, uo = units[l - 1], units[l]
        W = hidden[off:off + uo * ui].reshape(uo, ui); off += uo * ui
        b = hidden[off:off + uo]; off += uo
        z = _act(z @ W.T + b, act)
    s = cs * torch.tanh(z @ ws.T + bs)
    t = scale_t * (z @ wt.T + bt)
    q = cq * torch.tanh(z @ wq.T + bq)
    return s, t, q


def l2q_u1_vstep_f32(x, v, beta, eps, forward, nb, T, X, *net_and_out):
    *net, accumulate, logdet = net_and_out
    n = 2 * T * X
    f = _force(x.reshape(nb, 2, T, X), beta).reshape(nb, n)
    s, t, q = _fused_net(x.reshape(nb, n), f, *net)
    vn, ld = _v_update(v.reshape(nb, n), f, s, t, q, eps, bool(forward))
    v.copy_(vn.reshape(v.shape))
    logdet.copy_(logdet + ld if accumulate else ld)


def l2q_u1_xstep_f32(x, v, mask, complement, eps, forward, ncp, nb, n, *net_and_out):
    *net, accumulate, logdet = net_and_out
    a = _keep(mask, complement, x) * x.reshape(nb, n)
    s, t, q = _fused_net(torch.cat([torch.cos(a), torch.sin(a)], 1), v.reshape(nb, n), *net)
    xn, ld = _x_update(x.reshape(nb, n), v.reshape(nb, n), s, t, q, mask, complement, eps,
                       bool(forward), bool(ncp))
    x.copy_(xn.reshape(x.shape))
    logdet.copy_(logdet + ld if accumulate else ld)


# ---- training entry points (VJPs by autograd of the restatements above)
def l2q_act_fwd(x, act, n, esz, y):
    y.copy_(_act(x, act))


def l2q_act_bwd(dy, y, act, n, esz, dx):
    a = ACT[act]
    if a == 'swish':                       # y holds the pre-activation
        sg = torch.sigmoid(y)
        dx.copy_(dy * sg * (1 + y * (1 - sg)))
        return
    if a == 'tanh':
        d = 1 - y * y
    elif a == 'relu':
        d = (y > 0).to(y.dtype)
    elif a == 'leaky_relu':
        d = torch.where(y > 0, torch.ones_like(y), torch.full_like(y, 0.01))
    elif a == 'elu':
        d = torch.where(y > 0, torch.ones_like(y), y + 1)
    else:
        d = torch.ones_like(y)
    dx.copy_(dy * d)


def l2q_mul(a, b, alpha, n, esz, out):
    out.copy_(alpha * a * b)


def l2q_axpy_rows(x, a, nb, n, esz, y):
    y.add_((a.reshape(nb, 1) * x.reshape(nb, n)).reshape(y.shape))


def l2q_colsum(a, b, M, N, alpha, accumulate, esz, out, ws=None, wsn=0):
    r = alpha * ((a * b) if b is not None else a).reshape(M, N).double().sum(0).to(a.dtype)
    if accumulate:
        out.add_(r.reshape(out.shape))
    else:
        out.copy_(r.reshape(out.shape))


def l2q_scaled_tanh_bwd(ds, s, coeff, scale, M, N, esz, dpre):
    if coeff is None:
        dpre.copy_(scale * ds)
        return
    g = scale * torch.exp(coeff).reshape(1, N)
    th = torch.where(g != 0, s / g, torch.zeros_like(s))
    dpre.copy_(ds * g * (1 - th * th))


def l2q_scaled_tanh_bwd_sums(ds, s, coeff, scale, M, N, esz, dpre, bgrad, cgrad, ws=None, wsn=0):
    l2q_scaled_tanh_bwd(ds, s, coeff, scale, M, N, esz, dpre)
    l2q_colsum(dpre, None, M, N, 1.0, 1, esz, bgrad)
    if coeff is not None:
        l2q_colsum(ds, s, M, N, 1.0, 1, esz, cgrad)


def l2q_bn_train_fwd(x, gamma, beta, eps, momentum, rm, rv, M, N, esz, y, mean, invstd):
    mu = x.mean(0)
    var = x.var(0, unbiased=False)
    mean.copy_(mu)
    invstd.copy_(1 / torch.sqrt(var + eps))
    y.copy_((x - mu) * invstd * gamma + beta)
    if rm is not None:
        rm.mul_(1 - momentum).add_(momentum * mu)
        rv.mul_(1 - momentum).add_(momentum * x.var(0, unbiased=True))


def l2q_bn_bwd(dy, x, mean, invstd, gamma, M, N, esz, dx, dgamma, dbeta):
    eps_vec = (1 / invstd ** 2 - x.var(0, unbiased=False)).detach()

    def f(x_, g_, b_):
        mu = x_.mean(0)
        return (x_ - mu) / torch.sqrt(x_.var(0, unbiased=False) + eps_vec) * g_ + b_
    gx, gg, gb = _vjp(f, [x, gamma, torch.zeros_like(gamma)], [dy])
    dx.copy_(gx)
    dgamma.add_(gg)
    dbeta.add_(gb)


def l2q_col2im_periodic_f32(dcol, sn, sc, sh, sw, nb, C, H, W, k, clast, dx):
    if clast:
        dcol = _clast_cols(dcol.reshape(-1, C * k * k), C, k, False).contiguous()
    (g,) = _vjp(lambda a: _im2col(a, k), [torch.zeros(nb, C, H, W, dtype=dcol.dtype)], [dcol])
    if sw == 1:
        dx.copy_(g.reshape(dx.shape))
    else:
        dx.copy_(g.permute(0, 2, 3, 1).reshape(dx.shape))


def l2q_maxpool_act_nhwc_bwd_f32(dout, out, y, nb, H, W, C, pool, act, din):
    (g,) = _vjp(lambda a: _maxpool_act(a, nb, H, W, C, pool, act), [y.reshape(nb, H, W, C)], [dout])
    din.copy_(g.reshape(din.shape))


def l2q_u1_force_bwd(x, dF, beta, nb, T, X, esz, dx):
    (g,) = _vjp(lambda a: _force(a, beta), [x.reshape(nb, 2, T, X)], [dF])
    dx.add_(g.reshape(dx.shape))


def l2q_u1_plaq_bwd(x, gcos, gsin, nb, T, X, esz, dx):
    def f(a):
        th = _theta(a)
        return torch.cos(th).sum((1, 2)), torch.sin(th).sum((1, 2))
    (g,) = _vjp(f, [x.reshape(nb, 2, T, X)], [gcos, gsin])
    dx.add_(g.reshape(dx.shape))


def l2q_u1_x_update_bwd(x, v, s, t, q, mask, complement, eps, forward, ncp, gx, gl, esz, nb, n,
                        dx, dv, ds, dt, dq, deps):
    e = torch.tensor(float(eps), dtype=x.dtype).expand(nb).clone()

    def f(x_, v_, s_, t_, q_, e_):
        return _x_update(x_, v_, s_, t_, q_, mask, complement, e_.reshape(nb, 1), bool(forward),
                         bool(ncp))
    g = _vjp(f, [x.reshape(nb, n), v.reshape(nb, n), s, t, q, e], [gx.reshape(nb, n), gl])
    dx.copy_(g[0].reshape(dx.shape))
    dv.add_(g[1].reshape(dv.shape))
    ds.copy_(g[2]); dt.copy_(g[3]); dq.copy_(g[4]); deps.copy_(g[5])


def l2q_v_update_bwd(v, force, s, t, q, eps, forward, gv, gl, esz, nb, n, dv, dF, ds, dt, dq,
                     deps):
    e = torch.tensor(float(eps), dtype=v.dtype).expand(nb).clone()

    def f(v_, f_, s_, t_, q_, e_):
        return _v_update(v_, f_, s_, t_, q_, e_.reshape(nb, 1), bool(forward))
    g = _vjp(f, [v.reshape(nb, n), force.reshape(nb, n), s, t, q, e], [gv.reshape(nb, n), gl])
    dv.copy_(g[0].reshape(dv.shape)); dF.copy_(g[1].reshape(dF.shape))
    ds.copy_(g[2]); dt.copy_(g[3]); dq.copy_(g[4]); deps.copy_(g[5])


def l2q_u1_masked_cos_sin_bwd(x, mask, complement, dout, nb, n, esz, dx):
    def f(a):
        m = _keep(mask, complement, a) * a
        return torch.cat([torch.cos(m), torch.sin(m)], 1)
    (g,) = _vjp(f, [x.reshape(nb, n)], [dout.reshape(nb, 2 * n)])
    dx.add_(g.reshape(dx.shape))


def l2q_adam(p, g, m, v, n, lr, b1, b2, eps, step, gscale, esz):
    gi = g * gscale
    m.mul_(b1).add_((1 - b1) * gi)
    v.mul_(b2).add_((1 - b2) * gi * gi)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    p.sub_((lr / bc1) * m / (v.sqrt() / math.sqrt(bc2) + eps))


def l2q_sumsq(a, n, esz, out, ws, wsn):
    out[0] = (a.double() ** 2).sum()



# ============================================================================ SU(3) (native layout)
C128 = torch.complex128


def _adj(a):
    return a.conj().transpose(-1, -2)


def _mats(xn):
    """xn[..., 9, V] -> [..., V, 3, 3]"""
    V = xn.shape[-1]
    return xn.transpose(-1, -2).reshape(*xn.shape[:-2], V, 3, 3).contiguous()


def _native(m):
    """[..., V, 3, 3] -> [..., 9, V]"""
    V = m.shape[-3]
    return m.reshape(*m.shape[:-3], V, 9).transpose(-1, -2).contiguous()


def _roll(a, mu, sh):                      # a [nb, T, X, Y, Z, 3, 3]; value at s + sh * mu
    return torch.roll(a, -sh, dims=1 + mu)


def _su3_planes(x):
    """x [nb, 4, T, X, Y, Z, 3, 3] -> per-plane sums of tr P, [nb, 6] complex
    (lattice/su3/pytorch/lattice.py:157-199, planes (u > v) in loop order)"""
    out = []
    for u in range(1, 4):
        for v in range(u):
            P = x[:, u] @ _roll(x[:, v], u, 1) @ _adj(x[:, v] @ _roll(x[:, u], v, 1))
            out.append(torch.diagonal(P, dim1=-2, dim2=-1).sum(-1).sum((1, 2, 3, 4)))
    return torch.stack(out, 1)


def _tah(m):
    r = 0.5 * (m - _adj(m))
    tr = torch.diagonal(r, dim1=-2, dim2=-1).sum(-1) / 3
    return r - tr[..., None, None] * torch.eye(3, dtype=m.dtype)


def _su3_force(x, beta):
    """TAH(D x^H) with D = d action / dx held constant (lattice.py:299-308)."""
    xd = x.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        S = -(beta / 3.0) * _su3_planes(xd).real.sum()
        (D,) = torch.autograd.grad(S, xd)
    return _tah(D.detach() @ _adj(x))


def _proj_su(M):
    """polar factor with the determinant phase removed (utils.py:332-346), via eigh"""
    w, V = torch.linalg.eigh(_adj(M) @ M)
    U = M @ ((V * (1 / torch.sqrt(w))[..., None, :]) @ _adj(V))
    d = torch.linalg.det(U)
    th = -torch.atan2(d.imag, d.real) / 3
    return U * torch.complex(torch.cos(th), torch.sin(th))[..., None, None]


def _to_vec8(x):
    s3 = 1 / math.sqrt(3.0)
    x00, x01, x02 = x[..., 0, 0], x[..., 0, 1], x[..., 0, 2]
    x11, x12, x22 = x[..., 1, 1], x[..., 1, 2], x[..., 2, 2]
    return torch.stack([-2 * x01.imag, -2 * x01.real, x11.imag - x00.imag, -2 * x02.imag,
                        -2 * x02.real, -2 * x12.imag, -2 * x12.real,
                        s3 * (2 * x22.imag - x11.imag - x00.imag)], -1)


def _keep_n(mask_n, complement, nb, V):
    if mask_n is None:
        return None
    k = mask_n.reshape(1, 4, 9, V).double()
    return (1 - k) if complement else k


def _expm_mul(xn, vn, eps, keep):
    """native in / native out"""
    E = torch.matrix_exp(eps * _mats(vn))
    if keep is None:
        return _native(E @ _mats(xn))
    return keep * xn + _native(E @ _mats((1 - keep) * xn))


def l2q_su3_pack(x, out, nb, V):
    out.copy_(_native(x.reshape(nb, 4, V, 3, 3)))


def l2q_su3_unpack(xn, out, nb, V):
    out.copy_(_mats(xn.reshape(nb, 4, 9, V)).reshape(out.shape))


def l2q_su3_unpack_select(an, bn, mask, out, nb, V):
    pick = (mask.reshape(nb, 1, 1, 1) != 0)
    xn = torch.where(pick, an.reshape(nb, 4, 9, V), bn.reshape(nb, 4, 9, V))
    out.copy_(_mats(xn).reshape(out.shape))


def l2q_su3_plaq_reduce(xn, nb, T, X, Y, Z, out, ws, wsn):
    p = _su3_planes(_mats(xn.reshape(nb, 4, 9, -1)).reshape(nb, 4, T, X, Y, Z, 3, 3)).sum(1)
    out[:, 0] = p.real
    out[:, 1] = p.imag


def l2q_su3_plaq_planes(xn, nb, T, X, Y, Z, out, ws, wsn):
    p = _su3_planes(_mats(xn.reshape(nb, 4, 9, -1)).reshape(nb, 4, T, X, Y, Z, 3, 3))
    out[:, :, 0] = p.real
    out[:, :, 1] = p.imag


def l2q_su3_force(xn, beta, f, nb, T, X, Y, Z):
    x = _mats(xn.reshape(nb, 4, 9, -1)).reshape(nb, 4, T, X, Y, Z, 3, 3)
    f.copy_(_native(_su3_force(x, beta).reshape(nb, 4, -1, 3, 3)).reshape(f.shape))


def l2q_su3_force_kick(xn, beta, coef, vn, nb, T, X, Y, Z):
    x = _mats(xn.reshape(nb, 4, 9, -1)).reshape(nb, 4, T, X, Y, Z, 3, 3)
    vn.add_(coef * _native(_su3_force(x, beta).reshape(nb, 4, -1, 3, 3)).reshape(vn.shape))


def l2q_su3_force_kick_to(xn, beta, coef, vin, vout, nb, T, X, Y, Z):
    vout.copy_(vin)
    l2q_su3_force_kick(xn, beta, coef, vout, nb, T, X, Y, Z)


def l2q_su3_expm_mul(xn, vn, eps, mask_n, complement, out, nb, V):
    keep = _keep_n(mask_n, complement, nb, V)
    out.copy_(_expm_mul(xn.reshape(nb, 4, 9, V), vn.reshape(nb, 4, 9, V), eps, keep).reshape(out.shape))


def l2q_su3_expm_mul2(xn, vn, eps, mask_n, complement_first, out, nb, V):
    k1 = _keep_n(mask_n, complement_first, nb, V)
    x1 = _expm_mul(xn.reshape(nb, 4, 9, V), vn.reshape(nb, 4, 9, V), eps, k1)
    out.copy_(_expm_mul(x1, vn.reshape(nb, 4, 9, V), eps, 1 - k1).reshape(out.shape))


def l2q_su3_expm_mul2_vec8(xn, vn, eps, mask_n, complement_first, out, vec, nb, V):
    l2q_su3_expm_mul2(xn, vn, eps, mask_n, complement_first, out, nb, V)
    l2q_su3_projsu_vec8(out, vec, nb * 4, V)


def l2q_su3_project_su(xn, out, nf, V):
    out.copy_(_native(_proj_su(_mats(xn.reshape(nf, 9, V)))).reshape(out.shape))


def l2q_su3_projsu_vec8(xn, out, nf, V):
    v = _to_vec8(_proj_su(_mats(xn.reshape(nf, 9, V))))            # [nf, V, 8]
    out.copy_(v.transpose(-1, -2).reshape(out.shape))


def l2q_su3_kinetic_reduce(vn, nb, V, out, ws, wsn):
    out.copy_(0.5 * ((vn.reshape(nb, -1).abs() ** 2).sum(1) - 8.0 * 4 * V))


def l2q_su3_assemble_tah(normals, out, nf, V):
    n = normals.reshape(8, nf, V)
    h = math.sqrt(0.5)
    r3, r8 = h * n[0], h * n[1] / math.sqrt(3.0)
    r01, r02, r12, i01, i02, i12 = (h * n[k] for k in range(2, 8))
    m = torch.zeros(nf, V, 3, 3, dtype=C128)
    m[..., 0, 0] = 1j * (r8 + r3); m[..., 1, 1] = 1j * (r8 - r3); m[..., 2, 2] = 1j * (-2 * r8)
    m[..., 0, 1] = torch.complex(r01, i01); m[..., 1, 0] = torch.complex(-r01, i01)
    m[..., 0, 2] = torch.complex(r02, i02); m[..., 2, 0] = torch.complex(-r02, i02)
    m[..., 1, 2] = torch.complex(r12, i12); m[..., 2, 1] = torch.complex(-r12, i12)
    out.copy_(_native(m).reshape(out.shape))


def l2q_diff_norm2_reduce(a, b, nb, n, out, ws, wsn):
    out.copy_(((a - b).reshape(nb, -1).abs() ** 2).sum(1))


def l2q_scale_f64(x, alpha, y, n):
    y.copy_(alpha * x)


_v_update_real = l2q_v_update


def l2q_v_update(v, force, s, t, q, eps, forward, cplx, esz, nb, n, logdet, ws, wsn):   # noqa: F811
    if not cplx:
        return _v_update_real(v, force, s, t, q, eps, forward, cplx, esz, nb, n, logdet, ws, wsn)
    vn, ld = _v_update(v.reshape(nb, n), force.reshape(nb, n), s, t, q, eps, bool(forward))
    v.copy_(vn.reshape(v.shape))
    logdet.copy_(ld)


def l2q_v_update_to(vin, vout, force, s, t, q, eps, forward, cplx, esz, nb, n, logdet, ws, wsn):
    vout.copy_(vin)
    return l2q_v_update(vout, force, s, t, q, eps, forward, cplx, esz, nb, n, logdet, ws, wsn)


# ---- SU(3) training entry points
def l2q_su3_expm_mul_bwd(xn, vn, eps, mask_n, complement, gxnew, gx, gv, deps, nb, V, ws, wsn):
    keep = _keep_n(mask_n, complement, nb, V)
    e = torch.full((nb,), float(eps), dtype=torch.float64)
    g = _vjp(lambda x_, v_, e_: _expm_mul(x_, e_.reshape(nb, 1, 1, 1) * v_, 1.0, keep),
             [xn.reshape(nb, 4, 9, V), vn.reshape(nb, 4, 9, V), e], [gxnew.reshape(nb, 4, 9, V)])
    gx.copy_(g[0].reshape(gx.shape))
    gv.add_(g[1].reshape(gv.shape))
    deps.copy_(g[2])


def l2q_su3_expm_mul2_bwd(xn, vn, eps, mask_n, complement_first, gxnew, gx, gv, deps, nb, V, ws, wsn):
    k1 = _keep_n(mask_n, complement_first, nb, V)
    e = torch.full((nb,), float(eps), dtype=torch.float64)

    def f(x_, v_, e_):
        a = e_.reshape(nb, 1, 1, 1) * v_
        return _expm_mul(_expm_mul(x_, a, 1.0, k1), a, 1.0, 1 - k1)
    g = _vjp(f, [xn.reshape(nb, 4, 9, V), vn.reshape(nb, 4, 9, V), e], [gxnew.reshape(nb, 4, 9, V)])
    gx.copy_(g[0].reshape(gx.shape))
    gv.add_(g[1].reshape(gv.shape))
    deps.copy_(g[2])


def l2q_su3_projsu_vec8_bwd(xn, gvec, gm, nf, V):
    (g,) = _vjp(lambda a: _to_vec8(_proj_su(_mats(a))).transpose(-1, -2),
                [xn.reshape(nf, 9, V)], [gvec.reshape(nf, 8, V)])
    gm.add_(g.reshape(gm.shape))


def l2q_su3_force_bwd(xn, gf, beta, gx, nb, T, X, Y, Z):
    def f(a):
        x = _mats(a).reshape(nb, 4, T, X, Y, Z, 3, 3)
        return _native(_su3_force(x, beta).reshape(nb, 4, -1, 3, 3))
    (g,) = _vjp(f, [xn.reshape(nb, 4, 9, -1)], [gf.reshape(nb, 4, 9, -1)])
    gx.add_(g.reshape(gx.shape))


def l2q_su3_plaq_bwd(xn, w, gx, nb, T, X, Y, Z):
    wc = torch.complex(w.reshape(nb, 6, 2)[..., 0], w.reshape(nb, 6, 2)[..., 1])

    def f(a):
        x = _mats(a).reshape(nb, 4, T, X, Y, Z, 3, 3)
        return (wc.conj() * _su3_planes(x)).real.sum()
    (g,) = _vjp(f, [xn.reshape(nb, 4, 9, -1)], [torch.ones(())])
    gx.add_(g.reshape(gx.shape))


def _su3_rect_sum(x):
    """x [nb, 4, T, X, Y, Z, 3, 3] -> [nb]: sum Re tr of the 12 planar 2x1 loops per site
    (lattice/su3/pytorch/lattice.py:180-196, 262)"""
    tot = 0.0
    for u in range(4):
        for v in range(4):
            if u == v:
                continue
            a = x[:, u] @ _roll(x[:, u], u, 1) @ _roll(x[:, v], u, 2)
            b = x[:, v] @ _roll(x[:, u], v, 1) @ _roll(_roll(x[:, u], v, 1), u, 1)
            tot = tot + (a * b.conj()).sum((-1, -2)).real.sum((1, 2, 3, 4))
    return tot


def l2q_su3_rect_reduce(xn, nb, T, X, Y, Z, out, ws, wsn):
    x = _mats(xn.reshape(nb, 4, 9, -1)).reshape(nb, 4, T, X, Y, Z, 3, 3)
    out.copy_(_su3_rect_sum(x))


def _rect_grad(xn, nb, T, X, Y, Z, w):
    def f(a):
        x = _mats(a).reshape(nb, 4, T, X, Y, Z, 3, 3)
        return (w * _su3_rect_sum(x)).sum()
    (g,) = _vjp(f, [xn.reshape(nb, 4, 9, -1)], [torch.ones(())])
    return g


def l2q_su3_rect_force_add(xn, coef, fn, nb, T, X, Y, Z):
    # TAH(U A) = -TAH(A^H U^H) with A^H = d(sum Re tr R)/dU in torch's convention
    g = _mats(_rect_grad(xn, nb, T, X, Y, Z, torch.ones(nb, dtype=torch.float64)))
    u = _mats(xn.reshape(nb, 4, 9, -1))
    fn.add_(_native(-coef * _tah(g @ _adj(u))).reshape(fn.shape))


def l2q_su3_rect_bwd(xn, w, gx, nb, T, X, Y, Z):
    gx.add_(_rect_grad(xn, nb, T, X, Y, Z, w.reshape(nb)).reshape(gx.shape))


def l2q_v_update_bwd_c128(v, force, s, t, q, eps, forward, gv, gl, nb, n, dv, dF, ds, dt, dq,
                          deps, ws, wsn):
    e = torch.full((nb,), float(eps), dtype=torch.float64)

    def f(v_, f_, s_, t_, q_, e_):
        return _v_update(v_, f_, s_, t_, q_, e_.reshape(nb, 1), bool(forward))
    g = _vjp(f, [v.reshape(nb, n), force.reshape(nb, n), s, t, q, e], [gv.reshape(nb, n), gl])
    dv.copy_(g[0].reshape(dv.shape)); dF.copy_(g[1].reshape(dF.shape))
    ds.copy_(g[2]); dt.copy_(g[3]); dq.copy_(g[4]); deps.copy_(g[5])


def l2q_v_update_bwd_pair_c128(v1, vmid, force, s, t, q, eps1, fwd1, eps2, fwd2, flip, gv, gl, nb, n, dv, dF, ds,
                               dt, dq, deps1, deps2, ws, wsn):
    z = lambda a: torch.empty_like(a)
    dvm, dF2, ds2, dt2, dq2 = z(dv), z(dF), z(ds), z(dt), z(dq)
    l2q_v_update_bwd_c128(vmid, force, s, t, q, eps2, fwd2, gv, gl, nb, n, dvm, dF2, ds2, dt2, dq2, deps2, ws, wsn)
    if flip:
        dvm = -dvm
    l2q_v_update_bwd_c128(v1, force, s, t, q, eps1, fwd1, dvm, gl, nb, n, dv, dF, ds, dt, dq, deps1, ws, wsn)
    dF.add_(dF2); ds.add_(ds2); dt.add_(dt2); dq.add_(dq2)


def l2q_v_update_bwd_acc_c128(v, force, s, t, q, eps, forward, gv, gl, nb, n, aF, as_, at, aq, dv, dF, ds,
                              dt, dq, deps, ws, wsn):
    l2q_v_update_bwd_c128(v, force, s, t, q, eps, forward, gv, gl, nb, n, dv, dF, ds, dt, dq, deps, ws, wsn)
    dF.add_(aF.reshape(dF.shape)); ds.add_(as_); dt.add_(at); dq.add_(aq)


def l2q_diff_bwd_f64(x, y, a, nb, n, gx):
    if x.is_complex():                       # n counts doubles
        x, y, gx = torch.view_as_real(x), torch.view_as_real(y), torch.view_as_real(gx)
    gx.add_((2.0 * a.reshape(nb, 1) * (x.reshape(nb, n) - y.reshape(nb, n))).reshape(gx.shape))


def l2q_gemm_h_u1x(ht, x, mask, complement, W, M, N, xdim, A2, W2, K2, bias, bias2, act, C, ws, wsn):
    keep = (1.0 - mask if complement else mask).reshape(1, xdim)
    a = keep * x.reshape(M, xdim)
    A = torch.cat([torch.cos(a), torch.sin(a)], 1).contiguous()
    l2q_gemm_h(ht, A, 1, W, M, N, 2 * xdim, A2, W2, K2, bias, bias2, None, 1.0, act, C, 0, ws, wsn)


def l2q_conv_gemm_periodic_h(ht, x, x32, sn, sc, sh, sw, nb, C, H, W, k, w, clast, b, cout, act, out):
    hd = torch.float16 if ht == 0 else torch.bfloat16
    r16 = lambda t: t.to(hd).float()
    col = _im2col(r16(_as_nchw(x.float(), sn, sc, sh, sw, nb, C, H, W)), k)
    w = w.float()
    w = w.reshape(cout, k, k, C).permute(0, 3, 1, 2) if clast else w.reshape(cout, C, k, k)
    y = r16(col @ w.reshape(cout, -1).T + b)
    if act:
        y = r16(_act(y, act))
    out.copy_(y.reshape(out.shape).to(out.dtype))


def l2q_conv_pool_gemm_periodic_h(ht, x, x32, sn, sc, sh, sw, nb, C, H, W, k, w, clast, b, cout, act, out):
    """conv (no activation) -> MaxPool2d(2) -> activation: the two entry points it replaces, composed"""
    hd = torch.float16 if ht == 0 else torch.bfloat16
    Ho, Wo = H + k - 1, W + k - 1
    y = torch.empty(nb * Ho * Wo, cout, dtype=hd)
    l2q_conv_gemm_periodic_h(ht, x, x32, sn, sc, sh, sw, nb, C, H, W, k, w, clast, b, cout, 0, y)
    l2q_maxpool_act_nhwc_h(ht, y, nb, Ho, Wo, cout, 2, act, out)


def l2q_nchw_to_nhwc_pad_f32(x, nb, C, H, W, cpad, out):
    out.zero_()
    out.reshape(nb, H, W, cpad)[..., :C] = x.reshape(nb, C, H, W).permute(0, 2, 3, 1)


def l2q_nchw_to_nhwc_pad_h(ht, x, nb, C, H, W, cpad, out):
    out.zero_()
    out.reshape(nb, H, W, cpad)[..., :C] = x.reshape(nb, C, H, W).permute(0, 2, 3, 1).to(out.dtype)


def l2q_maxpool_act_nhwc_h(ht, y, nb, H, W, C, pool, act, out):
    hd = torch.float16 if ht == 0 else torch.bfloat16
    y4 = y.float().reshape(nb, H, W, C).permute(0, 3, 1, 2)
    o = _act(torch.nn.functional.max_pool2d(y4, pool), act).permute(0, 2, 3, 1)
    out.copy_(o.to(hd).reshape(out.shape))


def l2q_u1_heads_update_h(ht, Z, M, K, N, Ws, bs, cs, Wt, bt, scale_t, Wq, bq, cq, xupd, a, b, mask,
                          complement, eps, forward, ncp, logdet, accumulate, ws, wsn):
    sv, tv, qv = (torch.empty(M, N) for _ in range(3))
    one = torch.zeros(N)
    l2q_gemm_h(ht, Z, 0, Ws, M, N, K, None, None, 0, bs, None, one, 1.0, 1, sv, 1, None, 0)
    l2q_gemm_h(ht, Z, 0, Wt, M, N, K, None, None, 0, bt, None, None, scale_t, 0, tv, 1, None, 0)
    l2q_gemm_h(ht, Z, 0, Wq, M, N, K, None, None, 0, bq, None, one, 1.0, 1, qv, 1, None, 0)
    sv, qv = cs * sv, cq * qv
    if xupd:
        an, ld = _x_update(a.reshape(M, N), b.reshape(M, N), sv, tv, qv, mask, complement, eps,
                           bool(forward), bool(ncp))
    else:
        an, ld = _v_update(a.reshape(M, N), b.reshape(M, N), sv, tv, qv, eps, bool(forward))
    a.copy_(an.reshape(a.shape))
    if accumulate:
        logdet.add_(ld.to(logdet.dtype))
    else:
        logdet.copy_(ld)


_TABLE = {k: v for k, v in globals().items() if k.startswith('l2q_')}


def call(name, *args):
    fn = _TABLE.get(name)
    if fn is None:
        raise NotImplementedError(f'emu_native: {name} is not emulated')
    with torch.no_grad():
        fn(*args)


def install(monkeypatch):
    """Route l2hmc.native.call to the emulation (CPU tensors)."""
    from l2hmc import native
    monkeypatch.setattr(native, 'call', call)
