"""numpy oracle: SU(3) group numerics and the 4D Wilson plaquette action / force.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Arrays use the reference's public
layout ``x[nb, 4, T, X, Y, Z, 3, 3]`` complex128.  Citations are into
``/root/reference/src/l2hmc``.
"""
from __future__ import annotations

import numpy as np

SQRT1by2 = np.sqrt(0.5)
SQRT1by3 = np.sqrt(1.0 / 3.0)
SQRT3 = np.sqrt(3.0)
EPS = 1e-12


def adj(x: np.ndarray) -> np.ndarray:
    return np.conj(np.swapaxes(x, -1, -2))


def trace(x: np.ndarray) -> np.ndarray:
    """group/su3/pytorch/group.py:71-72"""
    return np.einsum('...ii->...', x)


def eye_like(x: np.ndarray) -> np.ndarray:
    return np.eye(3, dtype=x.dtype).reshape((1,) * (x.ndim - 2) + (3, 3))


def project_tah(x: np.ndarray) -> np.ndarray:
    """R = (X - X^H)/2 - tr(X - X^H)/(2N).  group/su3/pytorch/group.py:92-103."""
    r = 0.5 * (x - adj(x))
    d = trace(r) / 3.0
    return r - d[..., None, None] * eye_like(x)


def det3(m: np.ndarray) -> np.ndarray:
    return (
        m[..., 0, 0] * (m[..., 1, 1] * m[..., 2, 2] - m[..., 1, 2] * m[..., 2, 1])
        - m[..., 0, 1] * (m[..., 1, 0] * m[..., 2, 2] - m[..., 1, 2] * m[..., 2, 0])
        + m[..., 0, 2] * (m[..., 1, 0] * m[..., 2, 1] - m[..., 1, 1] * m[..., 2, 0])
    )


def expm(a: np.ndarray) -> np.ndarray:
    """Matrix exponential of general complex 3x3 matrices.

    The reference calls ``torch.matrix_exp`` (group/su3/pytorch/group.py:50,90), a library
    routine (scaling & squaring with Taylor polynomials).  Restated here as scaling &
    squaring with a degree-18 Taylor polynomial evaluated by Horner's rule; agreement with
    torch is pinned by the golden vectors (abs err ~1e-15).
    """
    a = np.asarray(a, dtype=np.complex128)
    nrm = np.abs(a).sum(-2).max(-1)                      # 1-norm
    with np.errstate(divide='ignore'):
        s = np.where(nrm > 0.5, np.ceil(np.log2(np.maximum(nrm, 1e-300) / 0.5)), 0.0)
    s = np.maximum(s, 0).astype(np.int64)
    a = a / (2.0 ** s)[..., None, None]
    eye = eye_like(a)
    r = eye + a / 18.0
    for k in range(17, 0, -1):
        r = eye + (a @ r) / float(k)
    smax = int(s.max()) if s.size else 0
    for i in range(smax):
        sq = r @ r
        r = np.where((s > i)[..., None, None], sq, r)
    return r


def eigs3x3(tr, p2, det):
    """Closed-form eigenvalues of a 3x3 PHM.  group/su3/pytorch/utils.py:227-283."""
    tr3 = tr / 3.0
    p23 = p2 / 3.0
    tr32 = tr3 * tr3
    q = np.abs(0.5 * (p23 - tr32))
    r = 0.25 * tr3 * (5 * tr32 - p2) - 0.5 * det
    sq = np.sqrt(q)
    sq3 = q * sq
    with np.errstate(divide='ignore', invalid='ignore'):
        isq3 = 1.0 / sq3
    isq3c = np.minimum(3e38, np.maximum(-3e38, isq3))
    rsq3c = r * isq3c
    rsq3 = np.minimum(1.0, np.maximum(-1.0, rsq3c))
    rsq3 = np.clip(rsq3, -1.0 + EPS, 1.0 - EPS)
    t = (1.0 / 3.0) * np.arccos(rsq3)
    st = np.sin(t)
    ct = np.cos(t)
    sqc = sq * ct
    sqs = SQRT3 * sq * st
    ll = tr3 + sqc
    return tr3 - 2 * sqc, ll + sqs, ll - sqs


def rsqrt_phm3f(tr, p2, det):
    """group/su3/pytorch/utils.py:286-317"""
    e0, e1, e2 = eigs3x3(tr, p2, det)
    se0, se1, se2 = np.sqrt(np.abs(e0)), np.sqrt(np.abs(e1)), np.sqrt(np.abs(e2))
    u = se0 + se1 + se2
    w = se0 * se1 * se2
    d = w * (se0 + se1) * (se0 + se2) * (se1 + se2)
    with np.errstate(divide='ignore', invalid='ignore'):
        di = 1.0 / d
    c0 = di * (w * u * u + e0 * se0 * (e1 + e2) + e1 * se1 * (e0 + e2)
               + e2 * se2 * (e0 + e1))
    c1 = -(tr * u + w) * di
    c2 = u * di
    return c0, c1, c2


def rsqrt_phm3(x: np.ndarray) -> np.ndarray:
    """group/su3/pytorch/utils.py:320-329"""
    tr = trace(x).real
    x2 = x @ x
    p2 = trace(x2).real
    det = det3(x).real
    c0, c1, c2 = rsqrt_phm3f(tr, p2, det)
    return (c0[..., None, None] * eye_like(x) + c1[..., None, None] * x
            + c2[..., None, None] * x2)


def project_u(x: np.ndarray) -> np.ndarray:
    """x (x^H x)^{-1/2}.  group/su3/pytorch/utils.py:332-338"""
    return x @ rsqrt_phm3(adj(x) @ x)


def project_su(x: np.ndarray) -> np.ndarray:
    """group/su3/pytorch/utils.py:341-346"""
    m = project_u(x)
    d = det3(m)
    p = (-1.0 / 3.0) * np.arctan2(d.imag, d.real)
    return m * (np.cos(p) + 1j * np.sin(p))[..., None, None]


def su3_to_vec(x: np.ndarray) -> np.ndarray:
    """8 real components.  group/su3/pytorch/utils.py:394-420"""
    c = -2.0
    x00, x01, x02 = x[..., 0, 0], x[..., 0, 1], x[..., 0, 2]
    x11, x12, x22 = x[..., 1, 1], x[..., 1, 2], x[..., 2, 2]
    return np.stack([
        c * x01.imag, c * x01.real, x11.imag - x00.imag,
        c * x02.imag, c * x02.real, c * x12.imag, c * x12.real,
        SQRT1by3 * (2 * x22.imag - x11.imag - x00.imag),
    ], axis=-1)


def vec_to_su3(v: np.ndarray) -> np.ndarray:
    """group/su3/pytorch/utils.py:423-445 (no projection)."""
    c = -0.5
    x01 = c * (v[..., 1] + 1j * v[..., 0])
    x02 = c * (v[..., 4] + 1j * v[..., 3])
    x12 = c * (v[..., 6] + 1j * v[..., 5])
    x2i = SQRT1by3 * v[..., 7]
    x0i = c * (x2i + v[..., 2])
    x1i = c * (x2i - v[..., 2])
    out = np.zeros(v.shape[:-1] + (3, 3), dtype=np.complex128)
    out[..., 0, 0] = 1j * x0i
    out[..., 1, 1] = 1j * x1i
    out[..., 2, 2] = 1j * x2i
    out[..., 0, 1] = x01
    out[..., 0, 2] = x02
    out[..., 1, 2] = x12
    out[..., 1, 0] = -np.conj(x01)
    out[..., 2, 0] = -np.conj(x02)
    out[..., 2, 1] = -np.conj(x12)
    return out


def group_to_vec(x: np.ndarray) -> np.ndarray:
    """su3_to_vec(projectSU(x)).  group/su3/pytorch/group.py:138-147"""
    return su3_to_vec(project_su(x))


def rand_tah3(normals: np.ndarray) -> np.ndarray:
    """Assemble traceless anti-Hermitian momenta from the 8 standard-normal fields the
    reference draws, in its draw order r3, r8, r01, r02, r12, i01, i02, i12.
    group/su3/pytorch/utils.py:171-195.  normals: [8, ...] -> [..., 3, 3]
    """
    r3 = SQRT1by2 * normals[0]
    r8 = SQRT1by2 * SQRT1by3 * normals[1]
    r01, r02, r12 = (SQRT1by2 * normals[k] for k in (2, 3, 4))
    i01, i02, i12 = (SQRT1by2 * normals[k] for k in (5, 6, 7))
    out = np.zeros(normals.shape[1:] + (3, 3), dtype=np.complex128)
    out[..., 0, 0] = 1j * (r8 + r3)
    out[..., 1, 1] = 1j * (r8 - r3)
    out[..., 2, 2] = 1j * (-2 * r8)
    out[..., 0, 1] = r01 + 1j * i01
    out[..., 1, 0] = -r01 + 1j * i01
    out[..., 0, 2] = r02 + 1j * i02
    out[..., 2, 0] = -r02 + 1j * i02
    out[..., 1, 2] = r12 + 1j * i12
    out[..., 2, 1] = -r12 + 1j * i12
    return out


def kinetic_energy(p: np.ndarray) -> np.ndarray:
    """0.5 * sum(|p|_F^2 - 8) per link.  group/su3/pytorch/group.py:125-126"""
    n2 = (np.abs(p) ** 2).sum((-2, -1))
    return 0.5 * (n2 - 8.0).reshape(p.shape[0], -1).sum(1)


def check_su(x: np.ndarray):
    """group/su3/pytorch/utils.py:376-391 -> (avg, max) per chain"""
    d = (np.abs(adj(x) @ x - eye_like(x)) ** 2).sum((-2, -1))
    d = d + np.abs(-1 + det3(x)) ** 2
    d = d.reshape(x.shape[0], -1)
    c = 2.0 * (3 * 3 + 1)
    return np.sqrt(d.mean(1) / c), np.sqrt(d.max(1) / c)


# ----------------------------------------------------------------------------- lattice
def wilson_loops(x: np.ndarray) -> np.ndarray:
    """tr of the 6 plaquettes per site -> [6, nb, T, X, Y, Z] complex.

    lattice/su3/pytorch/lattice.py:157-199 (c1 == 0 branch).  ``roll(-1, axis=u+1)`` on
    ``x[:, v]`` is the forward neighbour in direction u.
    """
    out = []
    for u in range(1, 4):
        for v in range(0, u):
            xu, xv = x[:, u], x[:, v]
            yuv = xu @ np.roll(xv, -1, axis=u + 1)
            yvu = xv @ np.roll(xu, -1, axis=v + 1)
            out.append(np.einsum('...ij,...ij->...', yuv, np.conj(yvu)))
    return np.stack(out)


def plaq_sums(x: np.ndarray):
    """(sum Re tr P, sum Im tr P) per chain."""
    w = wilson_loops(x)
    s = w.reshape(6, x.shape[0], -1).sum(-1).sum(0)
    return s.real, s.imag


def action(x: np.ndarray, beta: float) -> np.ndarray:
    """-(beta/3) sum Re tr P.  lattice/su3/pytorch/lattice.py:252-269"""
    re, _ = plaq_sums(x)
    return beta * re * (-1.0 / 3.0)


def volume(x: np.ndarray) -> int:
    return int(np.prod(x.shape[2:6]))


def plaqs(x: np.ndarray) -> np.ndarray:
    """lattice/su3/pytorch/lattice.py:201-206"""
    re, _ = plaq_sums(x)
    return re / (6 * 3 * volume(x))


def sin_charges(x: np.ndarray) -> np.ndarray:
    """lattice/su3/pytorch/lattice.py:237-240"""
    _, im = plaq_sums(x)
    return im / (6 * 3 * volume(x))


def int_charges(x: np.ndarray) -> np.ndarray:
    """lattice/su3/pytorch/lattice.py:232-235"""
    _, im = plaq_sums(x)
    return im / (32 * np.pi ** 2)


def staples(x: np.ndarray) -> np.ndarray:
    """A_mu(x) = sum_{nu != mu} [ U_nu(x+mu) U_mu(x+nu)^H U_nu(x)^H
                                 + U_nu(x+mu-nu)^H U_mu(x-nu)^H U_nu(x-nu) ]"""
    a = np.zeros_like(x)
    for mu in range(4):
        for nu in range(4):
            if nu == mu:
                continue
            xm, xn = x[:, mu], x[:, nu]
            xn_pmu = np.roll(xn, -1, axis=mu + 1)
            xm_pnu = np.roll(xm, -1, axis=nu + 1)
            up = xn_pmu @ adj(xm_pnu) @ adj(xn)
            xn_pmu_mnu = np.roll(xn_pmu, +1, axis=nu + 1)
            xm_mnu = np.roll(xm, +1, axis=nu + 1)
            xn_mnu = np.roll(xn, +1, axis=nu + 1)
            dn = adj(xn_pmu_mnu) @ adj(xm_mnu) @ xn_mnu
            a[:, mu] += up + dn
    return a


def grad_action(x: np.ndarray, beta: float) -> np.ndarray:
    """The reference takes autograd dS/dx and returns projectTAH(dS/dx @ x^H)
    (lattice/su3/pytorch/lattice.py:299-308).  In closed form:

        F_mu(x) = (beta/3) * TAH( U_mu(x) A_mu(x) ),   A = sum of the 6 staples.

    Checked against the reference's autograd result to 1.8e-15 (tests/golden).
    """
    return (beta / 3.0) * project_tah(x @ staples(x))


# ----------------------------------------------------------------------------- c1 != 0
def coeffs(beta: float, c1: float) -> dict:
    """lattice/su3/pytorch/lattice.py:83-91"""
    return {'plaq': beta * (1.0 - 8.0 * c1), 'rect': beta * c1}


def rect_loops(x: np.ndarray) -> np.ndarray:
    """The two rectangle traces per plane of _wilson_loops(needs_rect=True),
    lattice/su3/pytorch/lattice.py:180-196 -> [12, nb, T, X, Y, Z] complex, in the reference's
    order (tr_urul_, tr_uuud_ for (u, v) = (1,0), (2,0), (2,1), (3,0), (3,1), (3,2))."""
    out = []
    for u in range(1, 4):
        for v in range(0, u):
            xu, xv = x[:, u], x[:, v]
            yuv = xu @ np.roll(xv, -1, axis=u + 1)
            yvu = xv @ np.roll(xu, -1, axis=v + 1)
            yu = np.roll(xu, -1, axis=v + 1)
            yv = np.roll(xv, -1, axis=u + 1)
            uu = adj(xv) @ yuv
            ur = adj(xu) @ yvu
            ul = yuv @ adj(yu)
            ud = yvu @ adj(yv)
            ul_ = np.roll(ul, -1, axis=u + 1)
            ud_ = np.roll(ud, -1, axis=v + 1)
            out.append(np.einsum('...ij,...ij->...', ur, np.conj(ul_)))
            out.append(np.einsum('...ij,...ij->...', uu, np.conj(ud_)))
    return np.stack(out)


def rect_sums(x: np.ndarray) -> np.ndarray:
    """sum Re tr R per chain (rs.real.sum of lattice.py:262)."""
    return rect_loops(x).reshape(12, x.shape[0], -1).sum(-1).sum(0).real


def action_c1(x: np.ndarray, beta: float, c1: float) -> np.ndarray:
    """lattice/su3/pytorch/lattice.py:252-269 with c1 != 0."""
    c = coeffs(beta, c1)
    re, _ = plaq_sums(x)
    return (c['plaq'] * re + c['rect'] * rect_sums(x)) * (-1.0 / 3.0)


def _shift(f: np.ndarray, hops) -> np.ndarray:
    """f(x + sum of hops); hops = [(direction, +-n), ...]"""
    for mu, n in hops:
        f = np.roll(f, -n, axis=mu + 1)
    return f


def rect_staples(x: np.ndarray) -> np.ndarray:
    """Sum of the 18 rest-of-loop products of the 2x1 rectangles through each link, oriented so
    that tr(U_mu(x) A) is the loop (what autograd of sum Re tr R produces, conjugate-transposed)."""
    a = np.zeros_like(x)
    for mu in range(4):
        for nu in range(4):
            if nu == mu:
                continue
            um, un = x[:, mu], x[:, nu]
            M = lambda *h: _shift(um, h)      # noqa: E731
            Nn = lambda *h: _shift(un, h)     # noqa: E731
            m, n = mu, nu
            s1 = M((m, 1)) @ Nn((m, 2)) @ adj(M((m, 1), (n, 1))) @ adj(M((n, 1))) @ adj(Nn())
            s2 = Nn((m, 1)) @ adj(M((n, 1))) @ adj(M((m, -1), (n, 1))) @ adj(Nn((m, -1))) @ M((m, -1))
            s3 = M((m, 1)) @ adj(Nn((m, 2), (n, -1))) @ adj(M((m, 1), (n, -1))) @ adj(M((n, -1))) @ Nn((n, -1))
            s4 = adj(Nn((m, 1), (n, -1))) @ adj(M((n, -1))) @ adj(M((m, -1), (n, -1))) @ Nn((m, -1), (n, -1)) @ M((m, -1))
            s5 = Nn((m, 1)) @ Nn((m, 1), (n, 1)) @ adj(M((n, 2))) @ adj(Nn((n, 1))) @ adj(Nn())
            s6 = adj(Nn((m, 1), (n, -1))) @ adj(Nn((m, 1), (n, -2))) @ adj(M((n, -2))) @ Nn((n, -2)) @ Nn((n, -1))
            a[:, mu] += s1 + s2 + s3 + s4 + s5 + s6
    return a


def grad_action_c1(x: np.ndarray, beta: float, c1: float) -> np.ndarray:
    """projectTAH(dS/dx @ x^H) for the improved action (lattice.py:299-308 with c1 != 0):
    (1/3) TAH( U (c_plaq A_plaq + c_rect A_rect) ).  Pinned to the reference's autograd result
    in tests/golden/su3_c1.npz."""
    c = coeffs(beta, c1)
    return (1.0 / 3.0) * project_tah(x @ (c['plaq'] * staples(x) + c['rect'] * rect_staples(x)))
