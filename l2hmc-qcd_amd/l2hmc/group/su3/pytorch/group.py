"""``SU3`` group object -- API of src/l2hmc/group/su3/pytorch/group.py:33-227 on HIP kernels."""
from __future__ import annotations

import logging
from typing import Optional, Sequence

import torch

from l2hmc import DEVICE
from l2hmc import _ops as ops
from l2hmc.group.group import Group
from l2hmc.group.su3.pytorch.utils import (
    _native, _reference, checkSU, checkU, eyeOf, norm2, projectSU, projectTAH, projectU,
    randTAH3, rsqrtPHM3, rsqrtPHM3f, su3_to_vec, vec_to_su3,
)

log = logging.getLogger(__name__)
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

    def diff_trace(self, x: Tensor) -> Tensor:
        log.error('TODO')                       # stubs in the reference too (group.py:80-86)
        return x

    def diff2trace(self, x: Tensor) -> Tensor:
        log.error('TODO')
        return x

    def projectTAH(self, x: Tensor) -> Tensor:
        return projectTAH(x)

    def compat_proju(self, u: Tensor, x: Tensor) -> Tensor:
        """group.py:149-166 verbatim in behaviour: project solve(u, x)[0] to the traceless
        anti-Hermitian algebra and return u @ B (small-batch utility, plain tensor ops)."""
        _, n, _ = x.shape
        algebra_elem = torch.linalg.solve(u, x)[0]
        B = (algebra_elem - algebra_elem.conj().transpose(-2, -1)) / 2.
        trace = torch.einsum('bii->b', B)
        B = B - ((1 / n) * trace.unsqueeze(-1).unsqueeze(-1)
                 * torch.eye(n).repeat(x.shape[0], 1, 1))
        assert torch.abs(torch.mean(torch.einsum('bii->b', B))) < 1e-6
        return B

    @staticmethod
    def rsqrtPHM3f(tr: Tensor, p2: Tensor, det: Tensor):
        return rsqrtPHM3f(tr, p2, det)

    def rsqrtPHM3(self, x: Tensor) -> Tensor:
        return rsqrtPHM3(x)

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

