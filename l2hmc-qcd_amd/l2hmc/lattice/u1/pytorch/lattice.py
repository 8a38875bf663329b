"""``LatticeU1`` -- API of src/l2hmc/lattice/u1/pytorch/lattice.py:50-317 on HIP kernels."""
from __future__ import annotations

from math import pi as PI
from typing import Optional

import torch
from torch.special import i0, i1

import l2hmc.group.u1.pytorch.group as g
from l2hmc import DEVICE
from l2hmc import _autograd as AG
from l2hmc import _ops as ops
from l2hmc.configs import Charges, LatticeMetrics
from l2hmc.lattice.lattice import Lattice

TWOPI = 2. * PI
Tensor = torch.Tensor


def area_law(beta: float, nplaqs: int):
    return (i1(beta) / i0(beta)) ** nplaqs


def plaq_exact(beta: float | Tensor):
    """I1(beta)/I0(beta), computed in fp32 like the reference (lattice.py:37-42)."""
    beta = torch.as_tensor(beta, dtype=torch.float32)
    return (i1(beta) / i0(beta)).to(torch.get_default_dtype())


def project_angle(x: Tensor) -> Tensor:
    """For x in [-4pi, 4pi], returns x in [-pi, pi]."""
    return x - TWOPI * torch.floor((x + PI) / TWOPI)


def _beta(beta) -> float:
    return float(beta.item()) if isinstance(beta, torch.Tensor) else float(beta)


class PlaqSumsU1:
    """per-chain sums of cos / sin / project_angle of the plaquette angle: what the sampler's own
    consumers reduce the reference's ``wloops`` field to (``LatticeU1.plaq_sums``, one pass of
    `l2q_u1_plaq_reduce`).  ``wilson_loops`` returns the field itself as the reference does; every
    ``_plaqs / _charges / ...`` helper accepts either."""

    def __init__(self, sums: Tensor):
        self.cos, self.sin, self.proj = (sums[:, i].contiguous() for i in range(3))


class LatticeU1(Lattice):
    def __init__(self, nchains: int, shape: list[int]):
        assert len(shape) == 2
        self.g = g.U1Phase()
        self.nt, self.nx = shape
        self.nplaqs = self.nt * self.nx
        super().__init__(group=self.g, nchains=nchains, shape=list(shape))
        self.volume = self.nt * self.nx

    def _x(self, x: Tensor) -> Tensor:
        return x.to(DEVICE).reshape(-1, *self.xshape).contiguous()

    def wilson_loops(self, x: Tensor) -> Tensor:
        """theta = U0(t,x) + U1(t+1,x) - U0(t,x+1) - U1(t,x): [nb, T, X], the tensor the reference
        returns (lattice.py:154-159), from `l2q_u1_wilson_loops` (differentiable: its adjoint kernel)."""
        if AG.wants_grad(x):
            return AG.U1WilsonLoops.apply(x.to(DEVICE).reshape(-1, *self.xshape), self._lattice_shape)
        return ops.u1_wilson_loops(self._x(x), self._lattice_shape)

    def plaq_sums(self, x: Tensor) -> PlaqSumsU1:
        """the per-chain reductions of `wilson_loops(x)` in one fused pass (no field written)"""
        if AG.wants_grad(x):
            # differentiable route (loss.backward() of an autograd caller): l2q_u1_plaq_bwd behind it
            return PlaqSumsU1(AG.U1PlaqSums.apply(x.to(DEVICE).reshape(-1, *self.xshape),
                                                  self._lattice_shape))
        return PlaqSumsU1(ops.u1_plaq_sums(self._x(x), self._lattice_shape))

    def _get_wloops(self, x: Optional[Tensor] = None) -> PlaqSumsU1:
        if x is None:
            raise ValueError('One of `x` or `wloops` must be specified.')
        return self.plaq_sums(x)

    def draw_uniform_batch(self, requires_grad: bool = True) -> Tensor:
        """uniform in (-pi, pi) (lattice.py:67-71)"""
        return TWOPI * torch.rand(self._shape, requires_grad=requires_grad) - PI

    def kinetic_energy(self, v: Tensor) -> Tensor:
        return 0.5 * v.flatten(1) ** 2

    def action(self, x: Tensor, beta: Tensor) -> Tensor:
        """beta * sum(1 - cos theta) (lattice.py:80-86)"""
        return self._action(self.plaq_sums(x), beta)

    def _action(self, wloops, beta: Tensor) -> Tensor:
        if isinstance(wloops, Tensor):                   # the reference's field (lattice.py:80-86)
            return _beta(beta) * (1. - wloops.cos()).sum((1, 2))
        return _beta(beta) * (self.volume - wloops.cos)

    def action_with_grad(self, x: Tensor, beta: Tensor) -> tuple[Tensor, Tensor]:
        return self.action(x, beta), self.grad_action(x, beta)

    def grad_action(self, x: Tensor, beta: Tensor, create_graph: bool = True) -> Tensor:
        """F0 = beta[sin th - sin th(t,x-1)], F1 = beta[-sin th + sin th(t-1,x)]: the closed
        form of the reference's autograd (lattice.py:102-117)."""
        xx = self._x(x)
        return ops.u1_force(xx, _beta(beta), self._lattice_shape).reshape(x.shape)

    def plaqs_diff(self, beta: float, x: Optional[Tensor] = None,
                   wloops: Optional[PlaqSumsU1] = None) -> Tensor:
        wloops = self._get_wloops(x) if wloops is None else wloops
        plaqs = self._plaqs(wloops)
        return plaq_exact(torch.as_tensor(beta)).to(plaqs.device) * torch.ones_like(plaqs) - plaqs

    def calc_metrics(self, x: Tensor, beta: Optional[Tensor] = None) -> dict[str, Tensor]:
        w = self.plaq_sums(x)
        return {'plaqs': self._plaqs(w), 'intQ': self._int_charges(w),
                'sinQ': self._sin_charges(w)}

    def observables(self, x: Tensor) -> LatticeMetrics:
        w = self.plaq_sums(x)
        return LatticeMetrics(p4x4=self.plaqs4x4(x=x), plaqs=self._plaqs(w),
                              charges=self._charges(w))

    def plaqs(self, x: Optional[Tensor] = None, wloops: Optional[PlaqSumsU1] = None) -> Tensor:
        if wloops is None:
            if x is None:
                raise ValueError('One of `x` or `wloops` must be specified.')
            wloops = self.plaq_sums(x)
        return self._plaqs(wloops)

    def _plaqs(self, wloops) -> Tensor:
        if isinstance(wloops, Tensor):                   # (lattice.py:200-203)
            return wloops.cos().mean((1, 2))
        return wloops.cos / self.volume

    def wilson_loops4x4(self, x: Tensor) -> Tensor:
        """4x4 Wilson loops (lattice.py:161-186); an observable outside the leapfrog path,
        kept as plain tensor indexing."""
        x = x.reshape(-1, *self.xshape)
        xu, xv = x[:, 0], x[:, 1]
        return (
            xu + xu.roll(-1, dims=2) + xu.roll(-2, dims=2) + xu.roll(-3, dims=2)
            + xu.roll(-4, dims=2) + xv.roll((-4, -1), dims=(2, 1))
            + xv.roll((-4, -2), dims=(2, 1)) + xv.roll((-4, -3), dims=(2, 1))
            - xu.roll((-3, -4), dims=(2, 1)) - xu.roll((-2, -4), dims=(2, 1))
            - xu.roll((-1, -4), dims=(2, 1)) - xv.roll(-4, dims=1) - xv.roll(-3, dims=1)
            - xv.roll(-2, dims=1) - xv.roll(-1, dims=1) - xv
        ).T

    def plaqs4x4(self, x: Optional[Tensor] = None,
                 wloops4x4: Optional[Tensor] = None) -> Tensor:
        if wloops4x4 is None:
            if x is None:
                raise ValueError('One of `x` or `wloops` must be specified.')
            wloops4x4 = self.wilson_loops4x4(x)
        return self._plaqs4x4(wloops4x4)

    def _plaqs4x4(self, wloops4x4: Tensor) -> Tensor:
        return wloops4x4.cos().mean((1, 2))

    def plaq_loss(self, acc: Tensor, x1: Optional[Tensor] = None, x2: Optional[Tensor] = None,
                  wl1=None, wl2=None) -> Tensor:
        """-(acc * sum 2 (1 - cos(theta_2 - theta_1)) + 1e-4).mean()  (lattice.py:278-292).  From
        configurations: the plaquette angle is linear in the links, so theta_2 - theta_1 = theta(x2 - x1)
        and one reduction over the difference field gives the sum.  From `wilson_loops` tensors: the
        reference's expression."""
        if isinstance(wl1, Tensor) or isinstance(wl2, Tensor):
            w1 = self.wilson_loops(x1) if wl1 is None else wl1
            w2 = self.wilson_loops(x2) if wl2 is None else wl2
            ploss = acc.to(w1.device) * (2. * (1. - (w2 - w1).cos())).sum((1, 2)) + 1e-4
            return -ploss.mean(0)
        if x1 is None or x2 is None:
            raise ValueError('plaq_loss needs the configurations x1, x2 or `wilson_loops` tensors '
                             '(per-chain PlaqSumsU1 do not determine it)')
        d = self._x(x2) - self._x(x1)
        cosd = ops.u1_plaq_sums(d.contiguous(), self._lattice_shape)[:, 0]
        ploss = acc.to(cosd.device) * (2. * (self.volume - cosd)) + 1e-4
        return -ploss.mean(0)

    def charge_loss(self, acc: Tensor, x1: Optional[Tensor] = None, x2: Optional[Tensor] = None,
                    wl1: Optional[PlaqSumsU1] = None, wl2: Optional[PlaqSumsU1] = None) -> Tensor:
        """-(acc * (sinQ_2 - sinQ_1)^2 + 1e-4).mean()  (lattice.py:294-308)"""
        w1 = self._get_wloops(x1) if wl1 is None else wl1
        w2 = self._get_wloops(x2) if wl2 is None else wl2
        dq = (self._sin_charges(w2) - self._sin_charges(w1)) ** 2
        return -(acc.to(dq.device) * dq + 1e-4).mean(0)

    def _sin_charges(self, wloops) -> Tensor:
        if isinstance(wloops, Tensor):                   # (lattice.py:221-224)
            return wloops.sin().sum((1, 2)) / TWOPI
        return wloops.sin / TWOPI

    def _int_charges(self, wloops) -> Tensor:
        if isinstance(wloops, Tensor):                   # (lattice.py:226-228)
            return project_angle(wloops).sum((1, 2)) / TWOPI
        return wloops.proj / TWOPI

    def sin_charges(self, x: Optional[Tensor] = None,
                    wloops: Optional[PlaqSumsU1] = None) -> Tensor:
        wloops = self._get_wloops(x) if wloops is None else wloops
        return self._sin_charges(wloops)

    def int_charges(self, x: Optional[Tensor] = None,
                    wloops: Optional[PlaqSumsU1] = None) -> Tensor:
        wloops = self._get_wloops(x) if wloops is None else wloops
        return self._int_charges(wloops)

    def charges(self, x: Optional[Tensor] = None,
                wloops: Optional[PlaqSumsU1] = None) -> Charges:
        wloops = self._get_wloops(x) if wloops is None else wloops
        return self._charges(wloops)

    def _charges(self, wloops: PlaqSumsU1) -> Charges:
        return Charges(intQ=self._int_charges(wloops), sinQ=self._sin_charges(wloops))
