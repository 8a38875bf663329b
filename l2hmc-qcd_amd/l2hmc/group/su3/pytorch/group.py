"""``SU3`` group object -- API of src/l2hmc/group/su3/pytorch/group.py:33-227 on HIP kernels."""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from l2hmc import DEVICE
from l2hmc import _ops as ops
from l2hmc.group.group import Group
from l2hmc.group.su3.pytorch.utils import (
    _native, _reference, checkSU, checkU, eyeOf, norm2, projectSU, projectTAH, projectU,
    randTAH3, su3_to_vec, vec_to_su3,
)

Tensor = torch.Tensor
C128 = torch.complex128


class SU3(Group):
    def __init__(self) -> None:
        super().__init__(dim=4, shape=[3, 3], dtype=torch.complex128, name='SU3')

    def update_gauge(self, x: Tensor, p: Tensor) -> Tensor:
        """matrix_exp(p) @ x  (group.py:45-50) -> l2q_su3_expm_mul"""
        xn, pn = _native(x), _native(p)
        return _reference(ops.su3_expm_mul_n(xn, pn, 1.0), x.shape)

    def checkSU(self, x: Tensor):
        return checkSU(x)

    def checkU(self, x: Tensor):
        return checkU(x)

    def mul(self, a: Tensor, b: Tensor, adjoint_a: bool = False,
            adjoint_b: bool = False) -> Tensor:
        a, b = torch.broadcast_tensors(a, b)
        return _reference(ops.su3_mul_n(_native(a), _native(b), adjoint_a, adjoint_b), a.shape)

    def adjoint(self, x: Tensor) -> Tensor:
        return x.adjoint()

    def trace(self, x: Tensor) -> Tensor:
        return torch.diagonal(x, dim1=-2, dim2=-1).sum(-1)

    def exp(self, x: Tensor) -> Tensor:
        xn = _native(x)
        eye = torch.zeros_like(xn)
        eye[..., (0, 4, 8), :] = 1.0
        return _reference(ops.su3_expm_mul_n(eye, xn, 1.0), x.shape)

    def projectTAH(self, x: Tensor) -> Tensor:
        return projectTAH(x)

    def projectSU(self, x: Tensor) -> Tensor:
        return projectSU(x)

    def projectU(self, x: Tensor) -> Tensor:
        return projectU(x)

    def compat_proj(self, x: Tensor) -> Tensor:
        return projectSU(x)

    def random(self, shape: Sequence[int]) -> Tensor:
        """projectSU(randn + i randn), drawn on the CPU generator (group.py:113-119)."""
        r = torch.randn(*shape, dtype=torch.float64)
        i = torch.randn(*shape, dtype=torch.float64)
        return projectSU(torch.complex(r, i).to(DEVICE))

    def random_momentum(self, shape: Sequence[int]) -> Tensor:
        return randTAH3(shape[:-2])

    def kinetic_energy(self, p: Tensor) -> Tensor:
        """0.5 * sum(|p|_F^2 - 8) per chain (group.py:125-126)"""
        return ops.su3_kinetic_n(ops.su3_pack(p.to(DEVICE)))

    def vec_to_group(self, x: Tensor) -> Tensor:
        return self.compat_proj(vec_to_su3(x))

    def group_to_vec(self, x: Tensor) -> Tensor:
        """su3_to_vec(projectSU(x)) -> [..., 8]  (group.py:138-147), one fused kernel."""
        xn = _native(x)
        n = xn.shape[-1]
        v = ops.su3_projsu_vec8_n(xn)
        return ops.transpose(v.reshape(1, 8, n), 1, 8, n).reshape(*x.shape[:-2], 8)

    def norm2(self, x: Tensor, axis: Sequence[int] = (-2, -1),
              exclude: Optional[Sequence[int]] = None) -> Tensor:
        return norm2(x, axis, exclude)

