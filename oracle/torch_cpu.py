"""torch-CPU oracle of the SU(3) hot path, all host cores.  TEST INFRASTRUCTURE ONLY.

A second restatement of the reference's SU(3) trajectory beside the numpy one (oracle/su3.py,
oracle/dynamics.py), written with the *method* of the reference so that it can serve as the
``cpu_baseline`` of bench.py on the GPU box (the reference's own Python cannot travel there):

* Wilson loops by ``torch.roll`` + batched 3x3 matmuls   (lattice/su3/pytorch/lattice.py:157-199)
* force = autograd of the action, then projectTAH(dS/dx x^H)             (lattice.py:299-308)
* ``torch.linalg.matrix_exp`` for the link update            (group/su3/pytorch/group.py:45-50)
* projectSU through the closed-form 3x3 eigenvalues            (group/su3/pytorch/utils.py:227-346)
* the vnet as ``nn.functional.linear`` layers                  (network/pytorch/network.py:430-551)
* generalised leapfrog / merged trajectory / accept            (dynamics/pytorch/dynamics.py:956-1297)

torch's intra-op thread pool parallelises the matmuls / elementwise passes over all cores
(``torch.get_num_threads()`` is what bench.py reports as ``cores``).  Pinned to the reference by
tests/test_oracle_golden.py (su3_ops / su3_hmc / su3_l2hmc fixtures).  Citations are into
``/root/reference/src/l2hmc``.
"""
from __future__ import annotations

import math

import torch

C128 = torch.complex128


def adj(a):
    return a.mH


def tah(a):
    """group.py:92-103"""
    r = 0.5 * (a - a.mH)
    tr = torch.diagonal(r, dim1=-2, dim2=-1).sum(-1) / 3.0
    return r - tr[..., None, None] * torch.eye(3, dtype=a.dtype)


def det3(m):
    return (m[..., 0, 0] * (m[..., 1, 1] * m[..., 2, 2] - m[..., 1, 2] * m[..., 2, 1])
            - m[..., 0, 1] * (m[..., 1, 0] * m[..., 2, 2] - m[..., 1, 2] * m[..., 2, 0])
            + m[..., 0, 2] * (m[..., 1, 0] * m[..., 2, 1] - m[..., 1, 1] * m[..., 2, 0]))


def _inv_sqrt_coeffs(tr, p2, det):
    """coefficients of (PHM)^(-1/2) on {1, M, M^2}; utils.py:227-317 (same clamps)"""
    tr3, p23 = tr / 3.0, p2 / 3.0
    tr32 = tr3 * tr3
    q = (0.5 * (p23 - tr32)).abs()
    r = 0.25 * tr3 * (5.0 * tr32 - p2) - 0.5 * det
    sq = q.sqrt()
    isq3 = (1.0 / (q * sq)).clamp(-3e38, 3e38)
    c = (r * isq3).clamp(-1.0, 1.0).clamp(-1.0 + 1e-12, 1.0 - 1e-12)
    t = torch.acos(c) / 3.0
    sqc, sqs = sq * torch.cos(t), math.sqrt(3.0) * sq * torch.sin(t)
    ll = tr3 + sqc
    e0, e1, e2 = tr3 - 2.0 * sqc, ll + sqs, ll - sqs
    s0, s1, s2 = e0.abs().sqrt(), e1.abs().sqrt(), e2.abs().sqrt()
    u, w = s0 + s1 + s2, s0 * s1 * s2
    di = 1.0 / (w * (s0 + s1) * (s0 + s2) * (s1 + s2))
    c0 = di * (w * u * u + e0 * s0 * (e1 + e2) + e1 * s1 * (e0 + e2) + e2 * s2 * (e0 + e1))
    return c0, -(tr * u + w) * di, u * di


def project_su(x):
    """utils.py:320-346"""
    h = x.mH @ x
    h2 = h @ h
    tr = torch.diagonal(h, dim1=-2, dim2=-1).sum(-1).real
    p2 = torch.diagonal(h2, dim1=-2, dim2=-1).sum(-1).real
    c0, c1, c2 = _inv_sqrt_coeffs(tr, p2, det3(h).real)
    eye = torch.eye(3, dtype=x.dtype)
    m = x @ (c0[..., None, None] * eye + c1[..., None, None] * h + c2[..., None, None] * h2)
    d = det3(m)
    ph = torch.atan2(d.imag, d.real) * (-1.0 / 3.0)
    return m * torch.complex(torch.cos(ph), torch.sin(ph))[..., None, None]


def su3_to_vec(x):
    """utils.py:394-420"""
    a, b, c = x[..., 0, 1], x[..., 0, 2], x[..., 1, 2]
    d0, d1, d2 = x[..., 0, 0].imag, x[..., 1, 1].imag, x[..., 2, 2].imag
    return torch.stack([-2 * a.imag, -2 * a.real, d1 - d0, -2 * b.imag, -2 * b.real,
                        -2 * c.imag, -2 * c.real, (2 * d2 - d1 - d0) / math.sqrt(3.0)], -1)


def rand_tah3(n):
    """n: [8, ...] standard normals in the reference's draw order (utils.py:171-195)"""
    h = math.sqrt(0.5)
    r3, r8 = h * n[0], (h * math.sqrt(1.0 / 3.0)) * n[1]
    out = torch.zeros(n.shape[1:] + (3, 3), dtype=C128)
    z = torch.zeros_like(r3)
    out[..., 0, 0] = torch.complex(z, r8 + r3)
    out[..., 1, 1] = torch.complex(z, r8 - r3)
    out[..., 2, 2] = torch.complex(z, -2 * r8)
    for (i, j), (kr, ki) in {(0, 1): (2, 5), (0, 2): (3, 6), (1, 2): (4, 7)}.items():
        out[..., i, j] = torch.complex(h * n[kr], h * n[ki])
        out[..., j, i] = torch.complex(-h * n[kr], h * n[ki])
    return out


def plaq_trace_sum(x):
    """per-chain sum over sites and the 6 planes of tr P (complex)   lattice.py:157-199"""
    tot = 0
    for u in range(1, 4):
        for v in range(u):
            xu, xv = x[:, u], x[:, v]
            yuv = xu @ xv.roll(-1, dims=u + 1)
            yvu = xv @ xu.roll(-1, dims=v + 1)
            tot = tot + (yuv * yvu.conj()).sum((-1, -2)).flatten(1).sum(1)
    return tot


def action(x, beta):
    """-(beta/3) sum Re tr P     lattice.py:252-269"""
    return plaq_trace_sum(x).real * (-float(beta) / 3.0)


def grad_action(x, beta):
    """autograd dS/dx -> projectTAH(dS/dx x^H)     lattice.py:299-308"""
    x = x.detach().requires_grad_(True)
    with torch.enable_grad():
        s = action(x, beta)
        g, = torch.autograd.grad(s, x, grad_outputs=torch.ones_like(s))
    return tah(g @ x.detach().mH)


def kinetic(v):
    """group.py:125-126"""
    return 0.5 * ((v.abs() ** 2).sum((-1, -2)) - 8.0).flatten(1).sum(1)


def leapfrog_layer(xv, fv, w, nunits, act=torch.tanh):
    """LeapfrogLayer.forward, eval mode, no conv / batch norm   network.py:430-551"""
    lin = torch.nn.functional.linear
    z = act(lin(xv, w['input_layer.xlayer.weight'], w['input_layer.xlayer.bias'])
            + lin(fv, w['input_layer.vlayer.weight'], w['input_layer.vlayer.bias']))
    for i in range(nunits - 1):
        z = act(lin(z, w[f'hidden_layers.{i}.weight'], w[f'hidden_layers.{i}.bias']))
    s = torch.exp(w['scale.coeff']) * torch.tanh(lin(z, w['scale.layer.weight'], w['scale.layer.bias']))
    t = lin(z, w['transl.weight'], w['transl.bias'])
    q = torch.exp(w['transf.coeff']) * torch.tanh(lin(z, w['transf.layer.weight'], w['transf.layer.bias']))
    return s, t, q


class TorchSU3Dynamics:
    """Merged L2HMC / plain-HMC transitions on [nb, 4, T, X, Y, Z, 3, 3] complex128 CPU tensors.
    xeps / veps: raw parameters (eps = p / (1 + p), dynamics.py:82-83); masks: [nlf][xdim] with
    the kept entries = 1; weights: state_dict of the vnet as torch tensors (None: no network)."""

    def __init__(self, latvolume, nleapfrog, xeps, veps, masks, weights=None, nunits=1):
        self.L = tuple(int(i) for i in latvolume)
        self.nlf = int(nleapfrog)
        self.xeps = [float(p) / (1.0 + float(p)) for p in xeps]
        self.veps = [float(p) / (1.0 + float(p)) for p in veps]
        shape = (1, 4, *self.L, 3, 3)
        self.masks = [torch.as_tensor(m, dtype=torch.float64).reshape(shape) for m in masks]
        self.w, self.nunits = weights, nunits

    def _v(self, st, x, v, beta, fwd):
        """dynamics.py:1266-1297"""
        f = grad_action(x, beta)
        eps = self.veps[st]
        nb = x.shape[0]
        if self.w is None:
            s = t = q = torch.zeros((nb, v[0].numel()), dtype=torch.float64)
        else:
            s, t, q = leapfrog_layer(su3_to_vec(project_su(x)).reshape(nb, -1),
                                     su3_to_vec(project_su(f)).reshape(nb, -1), self.w, self.nunits)
        lj = (0.5 * eps * s) if fwd else (-0.5 * eps * s)
        es, eq, t = torch.exp(lj).reshape(v.shape), torch.exp(eps * q).reshape(v.shape), t.reshape(v.shape)
        fn = f * eq + t
        vn = es * v - 0.5 * eps * fn if fwd else es * (v + 0.5 * eps * fn)
        return vn, lj.sum(1)

    def _x(self, st, x, v, keep, fwd):
        """x' = keep x + expm(+-eps v) ((1 - keep) x)    dynamics.py:1420-1425, 1468-1474"""
        e = torch.linalg.matrix_exp((self.xeps[st] if fwd else -self.xeps[st]) * v)
        return keep * x + e @ ((1.0 - keep) * x)

    def _lf(self, step, x, v, beta, fwd):
        """dynamics.py:1187-1228"""
        st = step if fwd else self.nlf - step - 1
        m = self.masks[st]
        first, second = (m, 1.0 - m) if fwd else (1.0 - m, m)
        v, l0 = self._v(st, x, v, beta, fwd)
        x = self._x(st, x, v, first, fwd)
        x = self._x(st, x, v, second, fwd)
        v, l1 = self._v(st, x, v, beta, fwd)
        return x, v, l0 + l1

    def hamiltonian(self, x, v, beta):
        return kinetic(v) + action(x, beta)

    @staticmethod
    def _finish(x, v, xp, vp, h0, h1, sld, u):
        acc = torch.exp(torch.minimum(h0 - h1 + sld, torch.zeros_like(h0)))
        ma = (acc > torch.as_tensor(u, dtype=acc.dtype)).to(torch.float64)
        sel = ma.reshape(-1, *([1] * (x.dim() - 1)))
        xo = sel * xp + (1.0 - sel) * x
        return xo, {'acc': acc, 'acc_mask': ma, 'sumlogdet': ma * sld, 'x_prop': xp, 'v_prop': vp,
                    'v_init': v}

    @torch.no_grad()
    def apply_transition_fb(self, x, beta, normals, u):
        """dynamics.py:660-702, 956-1029"""
        v = rand_tah3(normals)
        h0 = self.hamiltonian(x, v, beta)
        xp, vp, sld = x, v, torch.zeros(x.shape[0], dtype=torch.float64)
        for step in range(self.nlf):
            xp, vp, ld = self._lf(step, xp, vp, beta, True)
            sld = sld + ld
        vp = -vp
        for step in range(self.nlf):
            xp, vp, ld = self._lf(step, xp, vp, beta, False)
            sld = sld + ld
        return self._finish(x, v, xp, vp, h0, self.hamiltonian(xp, vp, beta), sld, u)

    @torch.no_grad()
    def apply_transition_hmc(self, x, beta, normals, u, eps, nleapfrog):
        """dynamics.py:632-658, 900-954"""
        v = rand_tah3(normals)
        h0 = self.hamiltonian(x, v, beta)
        xp, vp = x, v
        for _ in range(nleapfrog):
            vp = vp - 0.5 * eps * grad_action(xp, beta)
            xp = torch.linalg.matrix_exp(eps * vp) @ xp
            vp = vp - 0.5 * eps * grad_action(xp, beta)
        sld = torch.zeros(x.shape[0], dtype=torch.float64)
        return self._finish(x, v, xp, vp, h0, self.hamiltonian(xp, vp, beta), sld, u)
