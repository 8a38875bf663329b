"""torch.autograd nodes for the lattice-sized reductions of the path.

The reference gets every derivative from torch.autograd over ATen ops (``loss.backward()`` in
trainers/pytorch/trainer.py:1284-1314).  Here the reductions are HIP kernels without a graph; these
``torch.autograd.Function`` wrappers give them one, so that an UNMODIFIED caller --
``loss = LatticeLoss(...)(x_init, x_prop, acc); loss.backward()`` -- back-propagates through
``l2q_su3_plaq_bwd`` / ``l2q_u1_plaq_bwd`` / ``l2q_diff_bwd_f64`` into the trajectory node of
``dynamics/pytorch/autograd.py``.  Cotangents follow torch's convention for complex tensors
(dL/dRe + i dL/dIm), which is also what the l2q backward kernels produce.

Only tensors that require a gradient (with grad mode on) take these routes; everything else stays on
the plain kernels.
"""
from __future__ import annotations

from typing import Sequence

import torch

from l2hmc import _ops as ops

Tensor = torch.Tensor


def wants_grad(*ts) -> bool:
    return torch.is_grad_enabled() and any(isinstance(t, Tensor) and t.requires_grad for t in ts)


def native_of(x: Tensor):
    """The native-layout original a transition attached to the reference-layout tensor it returned
    (`attach_native`), if the tensor has not been written since."""
    c = getattr(x, '_l2q_native', None)
    if c is not None and c[0] == x._version and c[1].shape[0] == x.shape[0]:
        return c[1]
    return None


def attach_native(x: Tensor, xn: Tensor) -> Tensor:
    x._l2q_native = (x._version, xn)
    return x


def su3_pack_cached(x: Tensor) -> Tensor:
    xn = native_of(x)
    return ops.su3_pack(x.detach().reshape(x.shape[0], -1)) if xn is None else xn


class SU3PlaqPlanes(torch.autograd.Function):
    """x [nb, 4, T, X, Y, Z, 3, 3] (any shape flattenable to it) -> [nb, 6, 2] per-plane
    (sum Re tr P, sum Im tr P): `l2q_su3_plaq_planes` forward, `l2q_su3_plaq_bwd` backward."""

    @staticmethod
    def forward(ctx, x: Tensor, lat: Sequence[int]):
        xn = su3_pack_cached(x)
        ctx.save_for_backward(xn)
        ctx.lat, ctx.shape = tuple(int(i) for i in lat), x.shape
        return ops.su3_plaq_planes_n(xn, lat)

    @staticmethod
    def backward(ctx, g):
        (xn,) = ctx.saved_tensors
        gx = torch.zeros_like(xn)
        ops.su3_plaq_bwd_(gx, xn, g.contiguous(), ctx.lat)
        return ops.su3_unpack(gx, ctx.lat).reshape(ctx.shape), None


class SU3WilsonLoops(torch.autograd.Function):
    """x -> [6, nb, T, X, Y, Z] complex: the trace field of the reference's `wilson_loops`
    (lattice/su3/pytorch/lattice.py:242-244); `l2q_su3_wilson_loops` / `l2q_su3_wilson_loops_bwd`."""

    @staticmethod
    def forward(ctx, x: Tensor, lat: Sequence[int]):
        xn = su3_pack_cached(x)
        ctx.save_for_backward(xn)
        ctx.lat, ctx.shape = tuple(int(i) for i in lat), x.shape
        return ops.su3_wilson_loops_n(xn, lat)

    @staticmethod
    def backward(ctx, g):
        (xn,) = ctx.saved_tensors
        gx = torch.zeros_like(xn)
        ops.su3_wilson_loops_bwd_(gx, xn, g, ctx.lat)
        return ops.su3_unpack(gx, ctx.lat).reshape(ctx.shape), None


class U1WilsonLoops(torch.autograd.Function):
    """x [nb, 2, T, X] -> theta [nb, T, X] (lattice/u1/pytorch/lattice.py:154-159), a linear map:
    `l2q_u1_wilson_loops` forward, its adjoint `l2q_u1_wilson_loops_bwd` backward."""

    @staticmethod
    def forward(ctx, x: Tensor, lat: Sequence[int]):
        ctx.lat, ctx.shape, ctx.dtype = tuple(int(i) for i in lat), x.shape, x.dtype
        return ops.u1_wilson_loops(x.detach().contiguous(), lat)

    @staticmethod
    def backward(ctx, g):
        dx = torch.zeros((g.shape[0], 2, *ctx.lat), dtype=ctx.dtype, device=g.device)
        ops.u1_wilson_loops_bwd_(dx, g, ctx.lat)
        return dx.reshape(ctx.shape), None


class SU3RectSums(torch.autograd.Function):
    """x -> [nb] sum Re tr R over the 12 planar 2x1 loops per site (c1 != 0 actions)."""

    @staticmethod
    def forward(ctx, x: Tensor, lat: Sequence[int]):
        xn = su3_pack_cached(x)
        ctx.save_for_backward(xn)
        ctx.lat, ctx.shape = tuple(int(i) for i in lat), x.shape
        return ops.su3_rect_sums_n(xn, lat)

    @staticmethod
    def backward(ctx, g):
        (xn,) = ctx.saved_tensors
        gx = torch.zeros_like(xn)
        ops.su3_rect_bwd_(gx, xn, g.contiguous(), ctx.lat)
        return ops.su3_unpack(gx, ctx.lat).reshape(ctx.shape), None


class U1PlaqSums(torch.autograd.Function):
    """x [nb, 2, T, X] -> [nb, 3] (sum cos theta, sum sin theta, sum project_angle(theta)).  The third
    column is piecewise constant in x (every link enters two plaquettes with opposite signs), so only
    the first two carry a gradient (`l2q_u1_plaq_bwd`)."""

    @staticmethod
    def forward(ctx, x: Tensor, lat: Sequence[int]):
        xx = x.detach().contiguous()
        ctx.save_for_backward(xx)
        ctx.lat, ctx.shape = tuple(int(i) for i in lat), x.shape
        return ops.u1_plaq_sums(xx, lat)

    @staticmethod
    def backward(ctx, g):
        (xx,) = ctx.saved_tensors
        dx = torch.zeros_like(xx)
        ops.u1_plaq_bwd_(dx, xx, g[:, 0].contiguous(), g[:, 1].contiguous(), ctx.lat)
        return dx.reshape(ctx.shape), None


class DiffNorm2(torch.autograd.Function):
    """(a, b) -> [nb] sum |a - b|^2 over float64 / complex128 tensors (`l2q_diff_norm2_reduce`,
    `l2q_diff_bwd_f64`)."""

    @staticmethod
    def forward(ctx, a: Tensor, b: Tensor):
        a_, b_ = a.detach().contiguous(), b.detach().contiguous()
        ctx.save_for_backward(a_, b_)
        return ops.diff_norm2(a_, b_)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = ops.diff_bwd_(torch.zeros_like(a), a, b, g.contiguous())
        if ctx.needs_input_grad[1]:
            gb = ops.diff_bwd_(torch.zeros_like(b), b, a, g.contiguous())
        return ga, gb


class SelectRows(torch.autograd.Function):
    """out[c] = a[c] if mask[c] else b[c] (`l2q_select_rows`; the accept / reject step
    x_out = ma x_prop + mr x_init of dynamics.py:677-682); the mask carries no gradient."""

    @staticmethod
    def forward(ctx, a: Tensor, b: Tensor, mask: Tensor):
        nb = a.shape[0]
        ctx.save_for_backward(mask)
        out = ops.select_rows(a.detach().reshape(nb, -1), b.detach().reshape(nb, -1), mask)
        return out.reshape(a.shape)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        nb = g.shape[0]
        g2 = g.contiguous().reshape(nb, -1)
        z = torch.zeros_like(g2)
        ga = ops.select_rows(g2, z, mask).reshape(g.shape) if ctx.needs_input_grad[0] else None
        gb = ops.select_rows(z, g2, mask).reshape(g.shape) if ctx.needs_input_grad[1] else None
        return ga, gb, None
