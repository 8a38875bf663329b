"""``LatticeLoss`` -- API of src/l2hmc/loss/pytorch/loss.py:21-210.

The value is computed from per-chain reductions done by the HIP kernels
(``l2q_su3_plaq_planes``, ``l2q_u1_plaq_reduce``, ``l2q_diff_norm2_reduce``).  When an input
requires a gradient (the proposal of a train-mode ``Dynamics.forward``), the reductions run as
``torch.autograd.Function`` nodes (``l2hmc/_autograd.py``) and ``loss.backward()`` works like in the
reference (trainers/pytorch/trainer.py:1284-1314); otherwise no graph is attached (``eval_step`` /
``hmc_step`` report ``loss`` exactly like the reference).
"""
from __future__ import annotations

from typing import Optional

import torch

from l2hmc import DEVICE
from l2hmc import _autograd as AG
from l2hmc import _ops as ops
from l2hmc.configs import LossConfig
from l2hmc.group.su3.pytorch.group import SU3
from l2hmc.group.u1.pytorch.group import U1Phase
from l2hmc.lattice.su3.pytorch.lattice import LatticeSU3
from l2hmc.lattice.u1.pytorch.lattice import LatticeU1

Tensor = torch.Tensor


class LatticeLoss:
    def __init__(self, lattice: LatticeU1 | LatticeSU3, loss_config: LossConfig):
        self.lattice = lattice
        self.config = loss_config
        self.xshape = self.lattice.xshape
        self.plaq_weight = torch.tensor(self.config.plaq_weight, dtype=torch.float)
        self.charge_weight = torch.tensor(self.config.charge_weight, dtype=torch.float)
        self.rmse_weight = torch.tensor(self.config.rmse_weight, dtype=torch.float)
        if isinstance(self.lattice, LatticeU1):
            self.g = U1Phase()
        elif isinstance(self.lattice, LatticeSU3):
            self.g = SU3()
        else:
            raise ValueError(f'Unexpected lattice: {type(lattice)}')

    def __call__(self, x_init: Tensor, x_prop: Tensor, acc: Tensor) -> Tensor:
        return self.calc_loss(x_init=x_init, x_prop=x_prop, acc=acc)

    @staticmethod
    def mixed_loss(loss: Tensor, weight: Tensor) -> Tensor:
        return (weight / loss) - (loss / weight)

    # per-plane real sums [6, nb] (SU3) -- what `w.real.sum(range(2, ndim))` gives (loss.py:64-65)
    def _plane_sums(self, x: Tensor) -> Tensor:
        assert isinstance(self.lattice, LatticeSU3)
        if AG.wants_grad(x):
            s = AG.SU3PlaqPlanes.apply(x.to(DEVICE), self.lattice._lattice_shape)
        else:
            s = ops.su3_plaq_planes_n(self.lattice.pack(x), self.lattice._lattice_shape)
        return s[:, :, 0].transpose(0, 1)

    def _mixed(self, loss: Tensor, weight: Tensor, use_mixed_loss: Optional[bool]) -> Tensor:
        use_mixed = self.config.use_mixed_loss if use_mixed_loss is None else use_mixed_loss
        weight = weight.to(loss.device)
        if use_mixed:
            return self.mixed_loss(loss + 1e-4, weight).mean()
        return (-loss / weight).mean()

    def plaq_loss(self, x_init: Tensor, x_prop: Tensor, acc: Tensor,
                  use_mixed_loss: Optional[bool] = None) -> Tensor:
        if not isinstance(self.lattice, LatticeSU3):
            raise NotImplementedError('U(1) plaq_loss broadcasts acc[nb] against [nb, T] in the '
                                      'reference (loss.py:64-66) and is unused (plaq_weight = 0)')
        p1, p2 = self._plane_sums(x_init), self._plane_sums(x_prop)
        ploss = acc.to(DEVICE) * (p2 - p1) ** 2
        # NB the reference's _plaq_loss takes use_mixed_loss literally (None -> not mixed)
        if use_mixed_loss:
            return self.mixed_loss(ploss + 1e-4, self.plaq_weight.to(DEVICE)).mean()
        return (-ploss / self.plaq_weight.to(DEVICE)).mean()

    def charge_loss(self, x_init: Tensor, x_prop: Tensor, acc: Tensor,
                    use_mixed_loss: Optional[bool] = None) -> Tensor:
        q1 = self.lattice._sin_charges(self.lattice.plaq_sums(x_init))
        q2 = self.lattice._sin_charges(self.lattice.plaq_sums(x_prop))
        return self._mixed(acc.to(DEVICE) * (q2 - q1) ** 2, self.charge_weight, use_mixed_loss)

    def rmse_loss(self, x_init: Tensor, x_prop: Tensor, acc: Tensor,
                  use_mixed_loss: Optional[bool] = None) -> Tensor:
        nb = x_init.shape[0]
        a = x_prop.to(DEVICE).reshape(nb, -1)
        b = x_init.to(DEVICE).reshape(nb, -1)
        if not a.is_complex():
            # same failure as the reference, whose rmse_loss takes `dx.imag` (loss.py:139)
            raise RuntimeError('imag is not implemented for tensors with non-complex dtypes.')
        if AG.wants_grad(a, b):
            d2 = AG.DiffNorm2.apply(a.to(torch.complex128), b.to(torch.complex128))
        else:
            d2 = ops.diff_norm2(a.to(torch.complex128), b.to(torch.complex128))
        nelem = a.shape[1]
        return self._mixed(acc.to(DEVICE) * (d2 / nelem).to(acc.dtype), self.rmse_weight,
                           use_mixed_loss)

    # ---- the reference's wloops-based helpers (loss.py:57-92, 166-192): w1, w2 are the tensors
    # `lattice.wilson_loops` returns (or, for the charge term, the per-chain PlaqSums of `plaq_sums`)
    def _plaq_loss(self, w1: Tensor, w2: Tensor, acc: Tensor,
                   use_mixed_loss: Optional[bool] = None) -> Tensor:
        """(loss.py:57-70) on `wilson_loops` tensors; `use_mixed_loss` is taken literally (None -> not
        mixed) like the reference"""
        if not isinstance(w1, Tensor) or not isinstance(w2, Tensor):
            raise TypeError('_plaq_loss takes the tensors of lattice.wilson_loops (per-chain PlaqSums do '
                            'not determine the per-plane sums): or use plaq_loss(x_init, x_prop, acc)')
        p1 = w1.real.sum(list(range(2, len(w1.shape))))
        p2 = w2.real.sum(list(range(2, len(w2.shape))))
        ploss = acc.to(p1.device) * (p2 - p1) ** 2
        if use_mixed_loss:
            return self.mixed_loss(ploss + 1e-4, self.plaq_weight.to(ploss.device)).mean()
        return (-ploss / self.plaq_weight.to(ploss.device)).mean()

    def _charge_loss(self, w1, w2, acc: Tensor, use_mixed_loss: Optional[bool] = None) -> Tensor:
        dq = (self.lattice._sin_charges(w2) - self.lattice._sin_charges(w1)) ** 2
        return self._mixed(acc.to(dq.device) * dq, self.charge_weight, use_mixed_loss)

    def general_loss(self, x_init: Tensor, x_prop: Tensor, acc: Tensor,
                     plaq_weight: Optional[float] = None, charge_weight: Optional[float] = None,
                     use_mixed_loss: Optional[bool] = None):
        pw = self.plaq_weight if plaq_weight is None else plaq_weight
        qw = self.charge_weight if charge_weight is None else charge_weight
        loss = 0.0
        if pw > 0:
            loss = loss + pw * self.plaq_loss(x_init, x_prop, acc, use_mixed_loss=use_mixed_loss)
        if qw > 0:
            loss = loss + qw * self.charge_loss(x_init, x_prop, acc, use_mixed_loss=use_mixed_loss)
        return loss

    def loss_from_sums(self, cos_init: Tensor, sin_init: Tensor, cos_prop: Tensor,
                       sin_prop: Tensor, acc: Tensor) -> Tensor:
        """U(1) training loss from the per-chain plaquette sums (sum cos theta, sum sin theta)
        of the initial / proposed configurations: the same value as `calc_loss`, written with
        differentiable torch ops on [nb] vectors only (dynamics/pytorch/training.py seeds the
        lattice-sized cotangent kernels with its gradient).  loss.py:100-148, 194-210."""
        if not isinstance(self.lattice, LatticeU1):
            raise NotImplementedError('training loss from sums: U(1) only')
        if self.plaq_weight > 0:
            raise NotImplementedError('U(1) plaq_loss is ill-formed in the reference '
                                      '(loss.py:64-66) and unused (plaq_weight = 0)')
        if self.rmse_weight > 0:
            raise RuntimeError('imag is not implemented for tensors with non-complex dtypes.')
        total = torch.zeros((), dtype=acc.dtype, device=acc.device)
        if self.charge_weight > 0:
            two_pi = 2.0 * torch.pi
            dq = sin_prop / two_pi - sin_init / two_pi
            total = total + self._mixed(acc * dq ** 2, self.charge_weight, None)
        return total

    def loss_from_sums_su3(self, planes_init: Tensor, planes_prop: Tensor, d2: Tensor, acc: Tensor,
                           nelem: int) -> Tensor:
        """SU(3) training loss from per-chain reductions: planes_* [nb, 6, 2] = per-plane
        (sum Re tr P, sum Im tr P), d2 [nb] = sum |x' - x|^2.  Same value as `calc_loss`
        (loss.py:56-148, 194-210), differentiable torch ops on small tensors only."""
        assert isinstance(self.lattice, LatticeSU3)
        total = torch.zeros((), dtype=acc.dtype, device=acc.device)
        if self.rmse_weight > 0:
            total = total + self._mixed(acc * (d2 / nelem).to(acc.dtype), self.rmse_weight, None)
        if self.plaq_weight > 0:
            p1, p2 = planes_init[:, :, 0].transpose(0, 1), planes_prop[:, :, 0].transpose(0, 1)
            ploss = acc * (p2 - p1) ** 2                                 # [6, nb]
            # the reference's _plaq_loss takes use_mixed_loss=None literally -> not mixed
            total = total + (-ploss / self.plaq_weight.to(acc.device)).mean()
        if self.charge_weight > 0:
            norm = 6 * 3 * self.lattice.volume
            q1, q2 = planes_init[:, :, 1].sum(1) / norm, planes_prop[:, :, 1].sum(1) / norm
            total = total + self._mixed(acc * (q2 - q1) ** 2, self.charge_weight, None)
        return total

    def lattice_metrics(self, xinit: Tensor, xout: Optional[Tensor] = None) -> dict[str, Tensor]:
        metrics = self.lattice.calc_metrics(x=xinit)
        if xout is not None:
            w = self.lattice.plaq_sums(xout.reshape(xinit.shape))
            metrics.update({'dQint': (self.lattice._int_charges(w) - metrics['intQ']).abs(),
                            'dQsin': (self.lattice._sin_charges(w) - metrics['sinQ']).abs()})
        return metrics

    def calc_loss(self, x_init: Tensor, x_prop: Tensor, acc: Tensor) -> Tensor:
        """plaq + charge + rmse terms (loss.py:194-210)"""
        zero = torch.zeros((), dtype=acc.dtype, device=DEVICE)
        rmse = self.rmse_loss(x_init, x_prop, acc) if self.rmse_weight > 0 else zero
        plaq = self.plaq_loss(x_init, x_prop, acc) if self.plaq_weight > 0 else zero
        charge = self.charge_loss(x_init, x_prop, acc) if self.charge_weight > 0 else zero
        return plaq + charge + rmse
