"""numpy oracle: the L2HMC / HMC leapfrog integrator.  TEST INFRASTRUCTURE ONLY.

Restates ``dynamics/pytorch/dynamics.py`` of the reference (citations below are
file:line into ``/root/reference/src/l2hmc``).  All randomness (momentum normals,
accept uniforms, masks) is an explicit input so results are a pure function of data.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import numpy as np

from . import su3 as _su3
from . import u1 as _u1


def sigmoid_log(p):
    """eps = sigmoid(log p) = 1/(1+exp(-log p)).  dynamics.py:82-83,1270"""
    p = np.asarray(p)
    return 1.0 / (1.0 + np.exp(-np.log(p)))


class DynamicsOracle:
    """group: 'SU3' | 'U1';  latvolume: lattice extents;  xeps/veps: raw parameter values
    (one per leapfrog index);  masks: list of float arrays [xdim] with xdim//2 ones;
    vnet(step, x_in, f_in) / xnet(step, first, x_in, v_in) -> (s, t, q) each [nb, xdim]
    (None => the reference's ``dummy_network``: zeros)."""

    def __init__(self, group: str, latvolume: Sequence[int], nleapfrog: int,
                 xeps, veps, masks, vnet: Optional[Callable] = None,
                 xnet: Optional[Callable] = None, use_ncp: bool = True,
                 merge_directions: bool = True, dtype=np.float64):
        self.group = group.upper()
        self.latvolume = tuple(latvolume)
        self.nlf = nleapfrog
        self.xeps = [np.asarray(e, dtype=dtype) for e in xeps]
        self.veps = [np.asarray(e, dtype=dtype) for e in veps]
        self.masks = [np.asarray(m).reshape(-1) for m in masks]
        self.vnet, self.xnet = vnet, xnet
        self.use_ncp = use_ncp
        self.merge_directions = merge_directions
        self.dtype = dtype
        if self.group == 'SU3':
            self.link_shape = (4, *self.latvolume, 3, 3)
        else:
            self.link_shape = (2, *self.latvolume)
        self.xdim = int(np.prod(self.link_shape))

    # ------------------------------------------------------------ physics
    def unflatten(self, x):
        return x.reshape(x.shape[0], *self.link_shape)

    def grad_potential(self, x, beta):
        """dynamics.py:1493-1499 -> lattice.grad_action"""
        x = self.unflatten(x)
        return _su3.grad_action(x, beta) if self.group == 'SU3' else _u1.grad_action(x, beta)

    def potential_energy(self, x, beta):
        x = self.unflatten(x)
        return _su3.action(x, beta) if self.group == 'SU3' else _u1.action(x, beta)

    def kinetic_energy(self, v):
        """dynamics.py:1485-1487 (the *group's* kinetic energy)"""
        return _su3.kinetic_energy(self.unflatten(v)) if self.group == 'SU3' \
            else _u1.kinetic_energy(v)

    def hamiltonian(self, x, v, beta):
        """dynamics.py:1479-1483"""
        return self.kinetic_energy(v) + self.potential_energy(x, beta)

    def update_gauge(self, x, p):
        """SU3: expm(p) @ x (group/su3/pytorch/group.py:45-50); U1: x + p"""
        return _su3.expm(p) @ x if self.group == 'SU3' else x + p

    @staticmethod
    def accept_prob(h_init, h_prop, sumlogdet):
        """exp(min(0, H_i - H_p + sum logdet)).  dynamics.py:1065-1079"""
        dh = h_init - h_prop + sumlogdet
        return np.exp(np.minimum(dh, 0.0))

    # ------------------------------------------------------------ plain HMC
    def leapfrog_hmc(self, x, v, beta, eps):
        """dynamics.py:900-913"""
        x_ = x.reshape(v.shape)
        f1 = self.grad_potential(x_, beta).reshape(v.shape)
        v1 = v - 0.5 * eps * f1
        xp = self.update_gauge(x_, eps * v1)
        f2 = self.grad_potential(xp, beta).reshape(v.shape)
        v2 = v1 - 0.5 * eps * f2
        return xp, v2

    def transition_kernel_hmc(self, x, v, beta, eps, nleapfrog=None, history=False):
        """dynamics.py:915-954"""
        nlf = 2 * self.nlf if self.merge_directions else self.nlf
        nleapfrog = nlf if nleapfrog is None else nleapfrog
        x_, v_ = x, v
        energies = [self.hamiltonian(x_, v_, beta)] if history else []
        for _ in range(nleapfrog):
            x_, v_ = self.leapfrog_hmc(x_, v_, beta, eps)
            if history:
                energies.append(self.hamiltonian(x_, v_, beta))
        sumlogdet = np.zeros(x.shape[0], dtype=self.dtype)
        acc = self.accept_prob(self.hamiltonian(x, v, beta),
                               self.hamiltonian(x_, v_, beta), sumlogdet)
        out = {'acc': acc, 'sumlogdet': sumlogdet}
        if history:
            out['energy'] = np.stack(energies)
        return x_, v_, out

    # ------------------------------------------------------------ L2HMC sub-updates
    def _call_vnet(self, step, x, force):
        """dynamics.py:1142-1159"""
        if self.vnet is None:
            z = np.zeros((x.shape[0], self.xdim), dtype=self.dtype)
            return z, z, z
        if self.group == 'SU3':
            x = _su3.group_to_vec(self.unflatten(x))
            force = _su3.group_to_vec(self.unflatten(force))
        return self.vnet(step, x, force)

    def _call_xnet(self, step, first, xm, v):
        """dynamics.py:1161-1185 (U1 only: SU3 never calls it, :1420-1425)"""
        if self.xnet is None:
            z = np.zeros((xm.shape[0], self.xdim), dtype=self.dtype)
            return z, z, z
        return self.xnet(step, first, _u1.group_to_vec(xm), v)

    def update_v(self, step, x, v, beta, forward: bool):
        """dynamics.py:1266-1280 (fwd) / :1282-1297 (bwd)"""
        force = self.grad_potential(x, beta)
        eps = sigmoid_log(self.veps[step])
        s, t, q = self._call_vnet(step, x, force)
        logjac = (eps * s / 2.0) if forward else (-eps * s / 2.0)
        logdet = logjac.reshape(logjac.shape[0], -1).sum(1)
        exp_s = np.exp(logjac).reshape(v.shape)
        exp_q = np.exp(eps * q).reshape(v.shape)
        t = t.reshape(v.shape)
        force = force.reshape(v.shape)
        fnew = force * exp_q + t
        if forward:
            vn = exp_s * v - 0.5 * eps * fnew
        else:
            vn = exp_s * (v + 0.5 * eps * fnew)
        return vn, np.real(logdet)

    def update_x(self, step, x, v, m, first: bool, forward: bool):
        """dynamics.py:1386-1428 (fwd) / :1430-1477 (bwd).  m: flat [xdim] mask that is
        *kept*; the complement (1-m) is updated."""
        eps = sigmoid_log(self.xeps[step])
        m = m.reshape(1, *self.link_shape).astype(np.float32)
        mb = (np.ones_like(m) - m)
        x = self.unflatten(x)
        xm = m * x
        nb = x.shape[0]
        if self.group == 'SU3':
            sgn = 1.0 if forward else -1.0
            vv = self.unflatten(v)
            xn = xm + _su3.expm(sgn * eps * vv) @ (mb * x)
            return xn, np.zeros(nb, dtype=self.dtype)
        xf_ = x.reshape(nb, -1)
        vf_ = v.reshape(nb, -1)
        s, t, q = self._call_xnet(step, first, xm.astype(x.dtype), v)
        s = (eps * s) if forward else (-eps * s)
        q = eps * q
        exp_s, exp_q = np.exp(s), np.exp(q)
        mbf = mb.reshape(1, -1)
        if self.use_ncp:
            halfx = xf_ / 2.0
            x1 = 2.0 * np.arctan(np.tan(halfx) * exp_s)
            if forward:
                xp = x1 + eps * (vf_ * exp_q + t)
            else:
                xp = x1 - exp_s * eps * (vf_ * exp_q + t)
            cterm = np.cos(halfx) ** 2
            sterm = (exp_s * np.sin(halfx)) ** 2
            logdet = (mbf * np.log(exp_s / (cterm + sterm))).sum(1)
        else:
            if forward:
                xp = xf_ * exp_s + eps * (vf_ * exp_q + t)
            else:
                xp = exp_s * (xf_ - eps * (vf_ * exp_q + t))
            logdet = (mbf * s).sum(1)
        xn = xm + mb * self.unflatten(xp)
        xn = _u1.compat_proj(xn.astype(x.dtype))
        return xn, np.real(logdet).astype(self.dtype)

    def forward_lf(self, step, x, v, beta):
        """dynamics.py:1187-1206"""
        m = self.masks[step]
        mb = 1.0 - m
        v, ld = self.update_v(step, x, v, beta, True)
        x, l1 = self.update_x(step, x, v, m, True, True)
        x, l2 = self.update_x(step, x, v, mb, False, True)
        v, l3 = self.update_v(step, x, v, beta, True)
        return x, v, ld + l1 + l2 + l3

    def backward_lf(self, step, x, v, beta):
        """dynamics.py:1208-1228"""
        step_r = self.nlf - step - 1
        m = self.masks[step_r]
        mb = 1.0 - m
        v, ld = self.update_v(step_r, x, v, beta, False)
        x, l1 = self.update_x(step_r, x, v, mb, False, False)
        x, l2 = self.update_x(step_r, x, v, m, True, False)
        v, l3 = self.update_v(step_r, x, v, beta, False)
        return x, v, ld + l1 + l2 + l3

    def transition_kernel_fb(self, x, v, beta, history=False):
        """dynamics.py:956-1029"""
        nb = x.shape[0]
        sumlogdet = np.zeros(nb, dtype=self.dtype)
        x_, v_ = x, v
        energies = [self.hamiltonian(x_, v_, beta)] if history else []
        logdets = [sumlogdet.copy()] if history else []
        for step in range(self.nlf):
            x_, v_, ld = self.forward_lf(step, x_, v_, beta)
            sumlogdet = sumlogdet + ld
            if history:
                energies.append(self.hamiltonian(x_, v_, beta))
                logdets.append(sumlogdet.copy())
        v_ = -v_
        for step in range(self.nlf):
            x_, v_, ld = self.backward_lf(step, x_, v_, beta)
            sumlogdet = sumlogdet + ld
            if history:
                energies.append(self.hamiltonian(x_, v_, beta))
                logdets.append(sumlogdet.copy())
        acc = self.accept_prob(self.hamiltonian(x, v, beta),
                               self.hamiltonian(x_, v_, beta), sumlogdet)
        out = {'acc': acc, 'sumlogdet': sumlogdet}
        if history:
            out['energy'] = np.stack(energies)
            out['logdet'] = np.stack(logdets)
        return x_, v_, out

    def transition_kernel(self, x, v, beta, forward: bool, history=False):
        """Single-direction kernel, dynamics.py:1031-1063 -- including its call
        ``compute_accept_prob(state_init=state, state_prop=sinit, ...)`` (:1053-1057), i.e. the
        FINAL state in the `init` slot: dh = H(final) - H(start) + sumlogdet, the opposite sign
        convention of transition_kernel_fb (:1023).  Restated as written."""
        nb = x.shape[0]
        lf = self.forward_lf if forward else self.backward_lf
        sumlogdet = np.zeros(nb, dtype=self.dtype)
        x_, v_ = x, v
        energies = [self.hamiltonian(x_, v_, beta)] if history else []
        logdets = [sumlogdet.copy()] if history else []
        for step in range(self.nlf):
            x_, v_, ld = lf(step, x_, v_, beta)
            sumlogdet = sumlogdet + ld
            if history:
                energies.append(self.hamiltonian(x_, v_, beta))
                logdets.append(sumlogdet.copy())
        acc = self.accept_prob(self.hamiltonian(x_, v_, beta),
                               self.hamiltonian(x, v, beta), sumlogdet)
        out = {'acc': acc, 'sumlogdet': sumlogdet}
        if history:
            out['energy'] = np.stack(energies)
            out['logdet'] = np.stack(logdets)
        return x_, v_, out

    # ------------------------------------------------------------ full transitions
    def random_momentum(self, normals):
        """SU3: normals [8, nb, 4, T, X, Y, Z] -> TAH matrices; U1: [nb, 2, T, X] -> flat.
        dynamics.py:844 -> group.random_momentum"""
        if self.group == 'SU3':
            return _su3.rand_tah3(normals)
        return normals.reshape(normals.shape[0], -1)

    def _select(self, x, xp, v, vp, acc, u):
        """dynamics.py:1081-1087, :665-682"""
        ma = (acc > u).astype(np.float32)
        mr = 1.0 - ma
        nb = x.shape[0]
        xo = ma[:, None] * xp.reshape(nb, -1) + mr[:, None] * x.reshape(nb, -1)
        vo = ma[:, None] * vp.reshape(nb, -1) + mr[:, None] * v.reshape(nb, -1)
        return xo, vo, ma

    def apply_transition_fb(self, x, beta, normals, u, history=False):
        """dynamics.py:660-702"""
        v = self.random_momentum(normals)
        xp, vp, m = self.transition_kernel_fb(x, v, beta, history=history)
        xo, vo, ma = self._select(x, xp, v, vp, m['acc'], u)
        m.update({'acc_mask': ma, 'sumlogdet': ma * m['sumlogdet'],
                  'v_init': v, 'x_prop': xp, 'v_prop': vp, 'v_out': vo})
        return xo, m

    def apply_transition(self, x, beta, forward: bool, normals, u, history=False):
        """merge_directions=False: dynamics.py:704-742 with the direction already drawn
        (`torch.rand(1) > 0.5`, :709)."""
        v = self.random_momentum(normals)
        xp, vp, m = self.transition_kernel(x, v, beta, forward, history=history)
        xo, vo, ma = self._select(x, xp, v, vp, m['acc'], u)
        m.update({'acc_mask': ma, 'sumlogdet': ma * m['sumlogdet'],
                  'v_init': v, 'x_prop': xp, 'v_prop': vp, 'v_out': vo})
        return xo, m

    def apply_transition_hmc(self, x, beta, normals, u, eps, nleapfrog=None,
                             history=False):
        """dynamics.py:632-658"""
        v = self.random_momentum(normals)
        xp, vp, m = self.transition_kernel_hmc(x, v, beta, eps, nleapfrog, history=history)
        xo, vo, ma = self._select(x, xp, v, vp, m['acc'], u)
        m.update({'acc_mask': ma, 'v_init': v, 'x_prop': xp, 'v_prop': vp, 'v_out': vo})
        return xo, m
