"""SU(3) matrix utilities on the GPU -- API of the reference's
``src/l2hmc/group/su3/pytorch/utils.py`` (projectSU :341-346, projectU :332-338, projectTAH
:349-359, randTAH3 :171-195, su3_to_vec :394-420, vec_to_su3 :423-445, norm2 :157-168,
checkU / checkSU :362-391, eyeOf :134-141).

Every function takes tensors whose last two dims are the 3x3 matrix (reference layout),
converts once to the native plane layout (``l2q_transpose``) and runs the register-resident
HIP kernels of ``csrc/su3_kernels.hip``.  No PyTorch arithmetic fallback exists.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from l2hmc import DEVICE
from l2hmc import _ops as ops

Tensor = torch.Tensor
C128 = torch.complex128

NP_SQRT1by2 = np.sqrt(1. / 2.)
NP_SQRT1by3 = np.sqrt(1. / 3.)
ONE_HALF = 1. / 2.
ONE_THIRD = 1. / 3.


def _native(x: Tensor) -> Tensor:
    """[..., 3, 3] -> [1, 1, 9, nlinks] native planes (one anonymous field)."""
    x = x.to(device=DEVICE, dtype=C128).contiguous()
    n = x.numel() // 9
    return ops.transpose(x.reshape(1, n, 9), 1, n, 9).reshape(1, 1, 9, n)


def _reference(xn: Tensor, shape: Sequence[int]) -> Tensor:
    n = xn.shape[-1]
    return ops.transpose(xn.reshape(1, 9, n), 1, 9, n).reshape(*shape)


def eyeOf(x: Tensor) -> Tensor:
    batch_dims = [1] * (len(x.shape) - 2)
    eye = torch.zeros(batch_dims + [*x.shape[-2:]], device=x.device)
    eye[-2:] = torch.eye(x.shape[-1], device=x.device)
    return eye


def norm2(x: Tensor, axis: Sequence[int] = (-2, -1),
          exclude: Optional[Sequence[int]] = None) -> Tensor:
    """No reduction if axis is empty (utils.py:157-168)."""
    if x.is_complex():
        x = x.abs()
    n = x.square()
    if exclude is None:
        return n if len(axis) == 0 else n.sum(tuple(axis))
    return n.sum([i for i in range(len(n.shape)) if i not in exclude])


def randTAH3(shape: Sequence[int]) -> Tensor:
    """Traceless anti-Hermitian matrices from 8 x randn(shape) drawn on the CPU torch
    generator in the reference's order (parity is defined against the CPU path)."""
    shape = tuple(int(i) for i in shape)
    normals = torch.stack([torch.randn(shape, dtype=torch.float64) for _ in range(8)])
    n = int(np.prod(shape))
    vn = ops.su3_assemble_tah_n(normals.reshape(8, 1, n).to(DEVICE))
    return _reference(vn.reshape(1, 1, 9, n), (*shape, 3, 3))


def eigs3x3(tr: Tensor, p2: Tensor, det: Tensor) -> tuple[Tensor, Tensor, Tensor]:
    """Eigenvalues of a 3x3 positive Hermitian matrix from tr X, tr X^2 and det X by the
    trigonometric (Cardano) form -- the function the reference exports as utils.py:227-283,
    including its clamp of the acos argument to +-(1 - 1e-12).  Host-side utility: the kernels
    have their own device version (csrc/su3_math.hpp m3_eigs)."""
    tr3, p23 = tr / 3.0, p2 / 3.0
    q = (0.5 * (p23 - tr3 * tr3)).abs()
    r = 0.25 * tr3 * (5.0 * tr3 * tr3 - p2) - 0.5 * det
    sq = torch.sqrt(q)
    ratio = (r / (q * sq).clamp(min=1e-300)).clamp(-3e38, 3e38)
    t = torch.acos(ratio.real.clamp(-1.0 + 1e-12, 1.0 - 1e-12)) / 3.0
    sqc, sqs = sq * torch.cos(t), (3.0 ** 0.5) * sq * torch.sin(t)
    ll = tr3 + sqc
    return tr3 - 2.0 * sqc, ll + sqs, ll - sqs


def rsqrtPHM3f(tr: Tensor, p2: Tensor, det: Tensor) -> tuple[Tensor, Tensor, Tensor]:
    """Coefficients c0, c1, c2 with X^{-1/2} = c0 + c1 X + c2 X^2 (utils.py:286-317)."""
    e0, e1, e2 = eigs3x3(tr, p2, det)
    s0, s1, s2 = e0.abs().sqrt(), e1.abs().sqrt(), e2.abs().sqrt()
    u, w = s0 + s1 + s2, s0 * s1 * s2
    di = 1.0 / (w * (s0 + s1) * (s0 + s2) * (s1 + s2))
    c0 = di * (w * u * u + e0 * s0 * (e1 + e2) + e1 * s1 * (e0 + e2) + e2 * s2 * (e0 + e1))
    return c0, -(tr * u + w) * di, u * di


def rsqrtPHM3(x: Tensor) -> Tensor:
    """X^{-1/2} of a positive Hermitian 3x3 X (utils.py:320-329)."""
    tr = torch.diagonal(x, dim1=-2, dim2=-1).sum(-1).real
    x2 = x @ x
    p2 = torch.diagonal(x2, dim1=-2, dim2=-1).sum(-1).real
    c0, c1, c2 = (c.reshape(c.shape + (1, 1)).to(x.dtype)
                  for c in rsqrtPHM3f(tr, p2, torch.linalg.det(x).real))
    return c0 * torch.eye(3, dtype=x.dtype, device=x.device) + c1 * x + c2 * x2


def projectU(x: Tensor) -> Tensor:
    """x (x^H x)^{-1/2}"""
    return _reference(ops.su3_project_u_n(_native(x)), x.shape)


def projectSU(x: Tensor) -> Tensor:
    return _reference(ops.su3_project_su_n(_native(x)), x.shape)


def projectTAH(x: Tensor) -> Tensor:
    """R = 1/2 (X - X^H) - 1/(2 N) tr(X - X^H)"""
    return _reference(ops.su3_project_tah_n(_native(x)), x.shape)


def checkSU(x: Tensor) -> tuple[Tensor, Tensor]:
    """(average, maximum) deviation of x^H x and det x from SU(3), per chain."""
    nb = x.shape[0]
    n = x.numel() // (9 * nb)
    if n % 4:
        raise ValueError('checkSU expects [nb, 4, ..., 3, 3]')
    r = ops.su3_check_su_n(ops.su3_pack(x.to(DEVICE)))
    return r[:, 0], r[:, 1]


def checkU(x: Tensor) -> tuple[Tensor, Tensor]:
    d = norm2(torch.matmul(x.adjoint(), x) - eyeOf(x).to(x.dtype))
    d_ = d.flatten(1)
    c = 2 * (3 * 3 + 1)
    return (d_.mean(-1) / c).sqrt(), (d_.max(-1)[0] / c).sqrt()


def su3_to_vec(x: Tensor) -> Tensor:
    """8 real numbers X^a with X = X^a T^a (anti-Hermitian input); pure indexing."""
    c = -2
    x00, x01, x02 = x[..., 0, 0], x[..., 0, 1], x[..., 0, 2]
    x11, x12, x22 = x[..., 1, 1], x[..., 1, 2], x[..., 2, 2]
    return torch.stack([
        c * x01.imag, c * x01.real, x11.imag - x00.imag, c * x02.imag, c * x02.real,
        c * x12.imag, c * x12.real,
        NP_SQRT1by3 * ((2 * x22.imag) - x11.imag - x00.imag),
    ], dim=-1)


def vec_to_su3(v: Tensor) -> Tensor:
    s3 = NP_SQRT1by3
    c = -0.5
    zero = torch.zeros_like(v[..., 0])
    x01 = c * torch.complex(v[..., 1], v[..., 0])
    x02 = c * torch.complex(v[..., 4], v[..., 3])
    x12 = c * torch.complex(v[..., 6], v[..., 5])
    x2i = s3 * v[..., 7]
    x0i = c * (x2i + v[..., 2])
    x1i = c * (x2i - v[..., 2])
    v00, v11, v22 = (torch.complex(zero, x0i), torch.complex(zero, x1i),
                     torch.complex(zero, x2i))
    return torch.stack([
        torch.stack([v00, -x01.conj(), -x02.conj()], -1),
        torch.stack([x01, v11, -x12.conj()], -1),
        torch.stack([x02, x12, v22], -1),
    ], -1)


# ------------------------------------------------------------------ module-level names a notebook may import
# (utils.py:26-47, 144-154, 448-514 of the reference; none of them is on the leapfrog path)
SQRT1by2, SQRT1by3, SQRT3 = NP_SQRT1by2, NP_SQRT1by3, float(np.sqrt(3.0))
TWO_PI = 2.0 * np.pi
EPS = 1e-12
# non-zero structure constants of [T^a, T^b] = f^abc T^c in the basis of su3_to_vec / vec_to_su3
f012, f036, f045, f135, f146, f234, f256 = +1.0, +0.5, -0.5, +0.5, +0.5, +0.5, -0.5
f347 = f567 = float(np.sqrt(0.75))
_F_NONZERO = {(0, 1, 2): f012, (0, 3, 6): f036, (0, 4, 5): f045, (1, 3, 5): f135, (1, 4, 6): f146,
              (2, 3, 4): f234, (2, 5, 6): f256, (3, 4, 7): f347, (5, 6, 7): f567}


def _structure_constants(dtype, device) -> Tensor:
    """f^abc [8, 8, 8], totally antisymmetric, from the nine independent non-zero entries."""
    f = torch.zeros(8, 8, 8, dtype=dtype, device=device)
    for (a, b, c), val in _F_NONZERO.items():
        for (i, j, k), sgn in (((a, b, c), 1), ((b, c, a), 1), ((c, a, b), 1),
                               ((b, a, c), -1), ((a, c, b), -1), ((c, b, a), -1)):
            f[i, j, k] = sgn * val
    return f


def su3fabc(v: Tensor) -> Tensor:
    """f^{abc} v[..., c] as an [..., 8, 8] matrix, laid out like the reference's nested stack
    (utils.py:448-488: entry [..., a, b] holds f^{abc} v_c; checked against the reference's table)."""
    f = _structure_constants(v.dtype, v.device)
    return torch.einsum('abc,...c->...ab', f, v)


def eye_like(x: Tensor) -> Tensor:
    """identity with the (2-D) shape, dtype and device of x (utils.py:144-145)"""
    return torch.eye(*x.size(), out=torch.empty_like(x)).to(DEVICE)


def expm(m: Tensor, order: int = 12) -> Tensor:
    """Taylor polynomial of exp(m) of the given order, evaluated by Horner's rule (utils.py:148-154) --
    the reference's own truncated series, NOT the integrator's exponential (the x-update uses
    torch.matrix_exp there and l2q_su3_expm_mul here)."""
    eye = eyeOf(m).to(m.dtype)
    acc = eye + m / order
    for i in range(order - 1, 0, -1):
        acc = eye + torch.matmul(m, acc) / i
    return acc


def SU3Gradient(f, x: Tensor, create_graph: bool = True) -> tuple[Tensor, Tensor]:
    """(f(x), df/dx) by autograd for a per-chain real f (utils.py:491-514).  With f = LatticeSU3.action the
    derivative comes from l2q_su3_plaq_bwd through l2hmc/_autograd.py; those nodes are first-order, so
    `create_graph` only matters for an f written in differentiable torch ops."""
    x.requires_grad_(True)
    y = f(x)
    ones = torch.ones(x.shape[0], device=y.device, dtype=y.dtype)
    dydx, = torch.autograd.grad(y, x, create_graph=create_graph, retain_graph=True, grad_outputs=ones)
    return y, dydx
