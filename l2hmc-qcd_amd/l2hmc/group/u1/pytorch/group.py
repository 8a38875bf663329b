"""``U1Phase`` group object -- API of src/l2hmc/group/u1/pytorch/group.py:62-165."""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from l2hmc import DEVICE
from l2hmc import _ops as ops
from l2hmc.group.group import Group

PI = torch.pi
TWO_PI = torch.pi * 2.
Tensor = torch.Tensor


def eyeOf(x: Tensor) -> Tensor:
    """identity broadcastable against x's trailing matrix axes (group.py:50-57)"""
    batch_dims = [1] * (len(x.shape) - 1)
    eye = torch.zeros(batch_dims + [*x.shape[-1:]]).to(x.device)
    eye[-1:] = torch.eye(x.shape[-1])
    return eye


def rand_unif(shape: Sequence[int], a: float, b: float, requires_grad: bool = True):
    """x ~ U(a, b) with shape `shape` (group.py:23-38)"""
    rand = (a - b) * torch.rand(tuple(shape)) + b
    return rand.clone().detach().requires_grad_(requires_grad)


def random_angle(shape: Sequence[int], requires_grad: bool = True) -> Tensor:
    return rand_unif(shape, -PI, PI, requires_grad=requires_grad)


class U1Phase(Group):
    def __init__(self) -> None:
        super().__init__(dim=2, shape=[1], dtype=torch.get_default_dtype())

    def floormod(self, x: Tensor | float, y: Tensor | float) -> Tensor:
        return (x - torch.floor_divide(x, y) * y)

    def phase_to_coords(self, phi: Tensor) -> Tensor:
        return torch.cat([phi.cos(), phi.sin()], -1)

    def coords_to_phase(self, x: Tensor) -> Tensor:
        assert x.shape[-1] == 2
        return torch.atan2(x[..., -1], x[..., -2])

    @staticmethod
    def group_to_vec(x: Tensor) -> Tensor:
        """cat([cos x, sin x], dim=1) (group.py:86-89) -> l2q_u1_masked_cos_sin, mask = 1"""
        nb = x.shape[0]
        xd = x.to(DEVICE).contiguous()
        n = xd.numel() // nb
        ones = torch.ones(n, dtype=torch.float32, device=xd.device)
        out = ops.u1_masked_cos_sin(xd.reshape(nb, 2, -1), ones, False, (1, n // 2))
        return out.reshape(nb, 2 * x.shape[1], *x.shape[2:])

    @staticmethod
    def vec_to_group(x: Tensor) -> Tensor:
        if x.is_complex():
            return torch.atan2(x.imag, x.real)
        return torch.atan2(x[..., -1], x[..., -2])

    def exp(self, x: Tensor) -> Tensor:
        return torch.complex(x.cos(), x.sin())

    def update_gauge(self, x: Tensor, p: Tensor) -> Tensor:
        """x + p (group.py:102-103) -> l2q_axpy"""
        out = x.to(DEVICE).clone().contiguous()
        return ops.axpy_(out, p.to(DEVICE).reshape(out.shape).contiguous(), 1.0)

    def mul(self, a: Tensor, b: Tensor, adjoint_a: Optional[bool] = None,
            adjoint_b: Optional[bool] = None) -> Tensor:
        if adjoint_a and adjoint_b:
            return -a - b
        if adjoint_a:
            return -a + b
        if adjoint_b:
            return a - b
        return a + b

    def adjoint(self, x: Tensor) -> Tensor:
        return -x

    def trace(self, x: Tensor) -> Tensor:
        return torch.cos(x)

    def diff_trace(self, x: Tensor) -> Tensor:
        return -torch.sin(x)

    def diff2trace(self, x: Tensor) -> Tensor:
        return -torch.cos(x)

    def compat_proj(self, x: Tensor) -> Tensor:
        """((x + pi) mod 2 pi) - pi (group.py:137-138) -> l2q_u1_wrap"""
        return ops.u1_wrap(x.to(DEVICE).contiguous())

    def projectTAH(self, x: Tensor) -> Tensor:
        return x

    def projectSU(self, x: Tensor) -> Tensor:
        return self.compat_proj(x)

    def random(self, shape: Sequence[int]) -> Tensor:
        return self.compat_proj(TWO_PI * torch.rand(*shape))

    def random_momentum(self, shape: Sequence[int]) -> Tensor:
        return torch.randn(*shape).reshape(shape[0], -1).to(DEVICE)

    def kinetic_energy(self, p: Tensor) -> Tensor:
        """0.5 sum p^2 per chain (group.py:164-165)"""
        return ops.u1_kinetic(p.to(DEVICE).reshape(p.shape[0], -1).contiguous())
