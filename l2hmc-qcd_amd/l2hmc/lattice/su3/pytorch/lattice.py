"""``LatticeSU3`` -- API of src/l2hmc/lattice/su3/pytorch/lattice.py:39-349 on HIP kernels.

The reference builds ~18 lattice-sized temporaries per action call (roll / bmm / stack) and
gets the force by autograd; here ``action``, ``plaqs`` and the charges come from ONE pass of
``l2q_su3_plaq_reduce`` (576 B per chain-site) and the force from ``l2q_su3_force``
(explicit staples + TAH, 1152 B per chain-site).  c1 != 0 (Iwasaki / DBW2, lattice.py:83-112,
180-196) adds ``l2q_su3_rect_reduce`` / ``l2q_su3_rect_force_add`` for the 2x1 rectangles.
"""
from __future__ import annotations

import logging
from typing import Optional

import numpy as np
import torch

import l2hmc.group.su3.pytorch.group as g
from l2hmc import DEVICE
from l2hmc import _autograd as AG
from l2hmc import _ops as ops
from l2hmc.configs import Charges
from l2hmc.lattice.lattice import Lattice

log = logging.getLogger(__name__)
Tensor = torch.Tensor
PI = np.pi
TWO_PI = 2. * np.pi


def _beta(beta) -> float:
    return float(beta.item()) if isinstance(beta, torch.Tensor) else float(beta)


def pbc(tup: tuple[int], shape: tuple[int]) -> list:
    return np.mod(tup, shape).tolist()


def mat_adj(mat: np.ndarray) -> np.ndarray:
    return mat.conj().T


class PlaqSums:
    """Per-chain (sum Re tr P, sum Im tr P): what the sampler's own consumers (action, plaquette,
    charges) reduce the reference's ``wloops`` field to.  ``LatticeSU3.plaq_sums`` returns it from ONE
    pass of `l2q_su3_plaq_reduce`; ``wilson_loops`` returns the field itself as the reference does, and
    every ``_plaqs / _charges / ...`` helper below accepts either."""

    def __init__(self, sums: Tensor):
        self.re = sums[:, 0].contiguous()
        self.im = sums[:, 1].contiguous()


def _site_sum(w: Tensor) -> Tensor:
    """the reference's reduction of a trace field [nplanes, nb, T, X, Y, Z] to [nb] (lattice.py:209, 228)"""
    return w.sum(tuple(range(2, len(w.shape)))).sum(0)


class LatticeSU3(Lattice):
    """4D lattice with SU(3) links: x.shape = [nb, 4, nt, nx, ny, nz, 3, 3] complex128."""
    dim = 4

    def __init__(self, nchains: int, shape: list[int], c1: float = 0.0) -> None:
        assert len(shape) == 4
        self.g = g.SU3()
        self.nt, self.nx, self.ny, self.nz = shape
        self.c1 = c1
        super().__init__(group=self.g, nchains=nchains, shape=list(shape))
        self.volume = self.nt * self.nx * self.ny * self.nz

    # ------------------------------------------------------------ native-layout core
    def pack(self, x: Tensor) -> Tensor:
        # (a tensor a transition returned still carries its native-layout original)
        return AG.su3_pack_cached(x.to(DEVICE))

    def unpack(self, xn: Tensor) -> Tensor:
        return ops.su3_unpack(xn, self._lattice_shape)

    def plaq_sums_n(self, xn: Tensor) -> Tensor:
        return ops.su3_plaq_sums_n(xn, self._lattice_shape)

    def rect_sums_n(self, xn: Tensor) -> Tensor:
        """[nb]: sum Re tr R over the 12 planar 2x1 loops per site (c1 != 0 actions)."""
        return ops.su3_rect_sums_n(xn, self._lattice_shape)

    def action_n(self, xn: Tensor, beta) -> Tensor:
        """-(1/3) [beta (1 - 8 c1) sum Re tr P + beta c1 sum Re tr R] (lattice.py:252-269)"""
        b = _beta(beta)
        if self.c1 == 0.0:
            return (-b / 3.0) * self.plaq_sums_n(xn)[:, 0]
        return (-b * (1.0 - 8.0 * self.c1) / 3.0) * self.plaq_sums_n(xn)[:, 0] \
            + (-b * self.c1 / 3.0) * self.rect_sums_n(xn)

    def grad_action_n(self, xn: Tensor, beta) -> Tensor:
        """(1/3) TAH(U (c_plaq A_plaq + c_rect A_rect)): staples of the plaquettes and, for
        c1 != 0, of the 18 rectangles through each link."""
        b = _beta(beta)
        if self.c1 == 0.0:
            return ops.su3_force_n(xn, b, self._lattice_shape)
        f = ops.su3_force_n(xn, b * (1.0 - 8.0 * self.c1), self._lattice_shape)
        return ops.su3_rect_force_add_n(xn, b * self.c1 / 3.0, f, self._lattice_shape)

    # ------------------------------------------------------------ reference API
    def coeffs(self, beta: Tensor) -> dict[str, Tensor]:
        return {'plaq': beta * (1.0 - 8.0 * self.c1), 'rect': beta * self.c1}

    def wilson_loops(self, x: Tensor) -> Tensor:
        """tr P for the 6 planes (u > v) and every site: [6, nb, T, X, Y, Z] complex, the tensor the
        reference returns (lattice.py:242-244), from `l2q_su3_wilson_loops` (differentiable:
        `l2q_su3_wilson_loops_bwd`).  The sampler itself never materialises it (`plaq_sums`)."""
        if AG.wants_grad(x):
            return AG.SU3WilsonLoops.apply(x.to(DEVICE), self._lattice_shape)
        return ops.su3_wilson_loops_n(self.pack(x), self._lattice_shape)

    def plaq_sums(self, x: Tensor) -> PlaqSums:
        """the per-chain reductions of `wilson_loops(x)` in one fused pass (no field written)"""
        if AG.wants_grad(x):
            # differentiable route (loss.backward() of an autograd caller): l2q_su3_plaq_bwd behind it
            return PlaqSums(AG.SU3PlaqPlanes.apply(x.to(DEVICE), self._lattice_shape).sum(1))
        return PlaqSums(self.plaq_sums_n(self.pack(x)))

    def _wilson_loops(self, x: Tensor, needs_rect: bool = False) -> tuple[Tensor, Tensor]:
        """(plaquette traces [6, nb, T, X, Y, Z], rectangle traces [12, nb, T, X, Y, Z]) like the
        reference (lattice.py:157-199); without `needs_rect` the second is zeros (a broadcast view)."""
        ps = self.wilson_loops(x)
        if not needs_rect:
            return ps, torch.zeros((), dtype=ps.dtype, device=ps.device).expand(12, *ps.shape[1:])
        rects = []
        for u in range(1, self.dim):
            for v in range(u):
                rects.extend(self.g.trace(r) for r in self._rectangles(x, u, v))
        return ps, torch.stack(rects)

    def _plaquettes(self, x: Tensor) -> Tensor:
        return self._plaqs(self.plaq_sums(x))

    def plaqs(self, x: Optional[Tensor] = None, wloops=None) -> Tensor:
        return self._plaqs(self.plaq_sums(x) if wloops is None else wloops)

    def charges(self, x: Optional[Tensor] = None, wloops=None) -> Charges:
        return self._charges(self.plaq_sums(x) if wloops is None else wloops)

    def int_charges(self, x: Optional[Tensor] = None, wloops=None) -> Tensor:
        return self._int_charges(self.plaq_sums(x) if wloops is None else wloops)

    def sin_charges(self, x: Optional[Tensor] = None, wloops=None) -> Tensor:
        return self._sin_charges(self.plaq_sums(x) if wloops is None else wloops)

    # the helpers take the reference's trace field (a Tensor, lattice.py:208-240) or a PlaqSums
    @staticmethod
    def _re_sum(wloops) -> Tensor:
        return wloops.re if isinstance(wloops, PlaqSums) else _site_sum(wloops.real)

    @staticmethod
    def _im_sum(wloops) -> Tensor:
        return wloops.im if isinstance(wloops, PlaqSums) else _site_sum(wloops.imag)

    def _plaqs(self, wloops) -> Tensor:
        return self._re_sum(wloops) / (6 * 3 * self.volume)

    def _charges(self, wloops) -> Charges:
        return Charges(intQ=self._int_charges(wloops), sinQ=self._sin_charges(wloops))

    def _int_charges(self, wloops) -> Tensor:
        return self._im_sum(wloops) / (32 * (np.pi ** 2))

    def _sin_charges(self, wloops) -> Tensor:
        return self._im_sum(wloops) / (6 * 3 * self.volume)

    def kinetic_energy(self, v: Tensor) -> Tensor:
        return self.g.kinetic_energy(v)

    # ---- per-site plaquette matrices (observables / debugging helpers, lattice.py:93-155);
    # the sampler itself only ever needs the reductions above
    def _link_staple_op(self, link: Tensor, staple: Tensor) -> Tensor:
        return self.g.mul(link, staple)

    def _plaquette(self, x: Tensor, u: int, v: int) -> Tensor:
        """U_u(x) U_v(x+u) U_u(x+v)^H U_v(x)^H as a matrix field [nb, T, X, Y, Z, 3, 3]"""
        x = x.to(DEVICE).reshape(x.shape[0], *self._shape[1:])
        xu, xv = x[:, u], x[:, v]
        xuv = self.g.mul(xu, xv.roll(shifts=-1, dims=(u + 1)))
        xvu = self.g.mul(xv, xu.roll(shifts=-1, dims=(v + 1)))
        return self.g.mul(xuv, xvu, adjoint_b=True)

    def _trace_plaquette(self, x: Tensor, u: int, v: int) -> Tensor:
        return self.g.trace(self._plaquette(x, u, v))

    def _plaquette_field(self, x: Tensor, needs_rect: bool = False):
        """matrix fields of the 6 plaquettes (and the 12 rectangles), lattice.py:132-155"""
        plaqs = [self._plaquette(x, u, v) for u in range(1, self.dim) for v in range(u)]
        rects = None
        if needs_rect:
            rects = []
            for u in range(1, self.dim):
                for v in range(u):
                    rects.extend(self._rectangles(x, u, v))
            rects = torch.stack(rects)
        return torch.stack(plaqs), rects

    def _rectangles(self, x: Tensor, u: int, v: int) -> tuple[Tensor, Tensor]:
        """The two 2x1 loops of plane (u, v) as matrix fields (lattice.py:96-112): an
        observables / debugging helper, the sampler only needs rect_sums_n."""
        x = x.to(DEVICE).reshape(x.shape[0], *self._shape[1:])
        xu, xv = x[:, u], x[:, v]
        xuv = self.g.mul(xu, xv.roll(shifts=-1, dims=(u + 1)))
        xvu = self.g.mul(xv, xu.roll(shifts=-1, dims=(v + 1)))
        yu = xu.roll(-1, dims=v + 1)
        yv = xv.roll(-1, dims=u + 1)
        uu = self.g.mul(xv, xuv, adjoint_a=True)
        ur = self.g.mul(xu, xvu, adjoint_a=True)
        ul = self.g.mul(xuv, yu, adjoint_b=True)
        ud = self.g.mul(xvu, yv, adjoint_b=True)
        ul_ = ul.roll(-1, dims=u + 1)
        ud_ = ud.roll(-1, dims=v + 1)
        return self.g.mul(ur, ul_, adjoint_b=True), self.g.mul(uu, ud_, adjoint_b=True)

    def _action(self, wloops, beta: Tensor) -> Tensor:
        """The reference's unused opposite-sign variant (lattice.py:271-285): +coeff sum Re tr P / 3."""
        ps, rs = wloops if isinstance(wloops, tuple) else (wloops, None)
        c = self.coeffs(torch.as_tensor(_beta(beta)))
        action = c['plaq'] * self._re_sum(ps)
        if self.c1 != 0 and rs is not None:
            action = action + c['rect'] * (_site_sum(rs.real) if rs.dim() > 1 else rs)
        return action / 3.0

    def plaq_loss(self, acc: Tensor, x1=None, x2=None, wloops1=None, wloops2=None):
        log.error('TODO')                       # a stub in the reference as well (lattice.py:351-359)

    def charge_loss(self, acc: Tensor, x1=None, x2=None, wloops1=None, wloops2=None):
        log.error('TODO')                       # (lattice.py:361-369)

    def action(self, x: Tensor, beta: Tensor) -> Tensor:
        """-(1/3) (c_plaq sum Re tr P + c_rect sum Re tr R) (lattice.py:252-269)"""
        if AG.wants_grad(x):
            b = _beta(beta)
            s = (-b * (1.0 - 8.0 * self.c1) / 3.0) * AG.SU3PlaqPlanes.apply(
                x.to(DEVICE), self._lattice_shape)[:, :, 0].sum(1)
            if self.c1 != 0.0:
                s = s + (-b * self.c1 / 3.0) * AG.SU3RectSums.apply(x.to(DEVICE), self._lattice_shape)
            return s
        return self.action_n(self.pack(x), beta)

    def action_with_grad(self, x: Tensor, beta: Tensor) -> tuple[Tensor, Tensor]:
        xn = self.pack(x)
        return self.action_n(xn, beta), self.unpack(self.grad_action_n(xn, beta))

    def grad_action(self, x: Tensor, beta: Tensor) -> Tensor:
        """(beta/3) TAH(U * staples)  ==  projectTAH(autograd dS/dx @ x^H) (lattice.py:299-308)"""
        return self.unpack(self.grad_action_n(self.pack(x), beta))

    def calc_metrics(self, x: Tensor, beta: Optional[Tensor] = None,
                     xinit: Optional[Tensor] = None) -> dict[str, Tensor]:
        w = self.plaq_sums(x)
        q = self._charges(w)
        metrics = {'plaqs': self._plaqs(w), 'sinQ': q.sinQ, 'intQ': q.intQ}
        if beta is not None:
            s, dsdx = self.action_with_grad(x, beta)
            metrics.update({'action': s, 'dsdx': dsdx})
            if xinit is not None:
                s_, dsdx_ = self.action_with_grad(xinit, beta)
                metrics.update({'daction': (s - s_).abs(), 'dsdx': (dsdx - dsdx_).abs()})
        if xinit is not None:
            w_ = self.plaq_sums(xinit)
            q_ = self._charges(w_)
            metrics.update({'dplaqs': (metrics['plaqs'] - self._plaqs(w_)).abs(),
                            'dQint': (q.intQ - q_.intQ).abs(),
                            'dQsin': (q.sinQ - q_.sinQ).abs()})
        return metrics
