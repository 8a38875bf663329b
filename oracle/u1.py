"""numpy oracle: 2D U(1) lattice (angles).  TEST INFRASTRUCTURE ONLY.

Layout ``x[nb, 2, T, X]`` real (float32 or float64).  Citations are into
``/root/reference/src/l2hmc``.
"""
from __future__ import annotations

import numpy as np

PI = np.pi
TWO_PI = 2.0 * np.pi


def compat_proj(x):
    """((x + pi) mod 2pi) - pi.  group/u1/pytorch/group.py:137-138"""
    dt = x.dtype
    return (np.mod(x + dt.type(PI), dt.type(TWO_PI)) - dt.type(PI)).astype(dt)


def wilson_loops(x):
    """theta = U0(t,x) + U1(t+1,x) - U0(t,x+1) - U1(t,x).
    lattice/u1/pytorch/lattice.py:154-159"""
    xu, xv = x[:, 0], x[:, 1]
    return xu + np.roll(xv, -1, axis=1) - np.roll(xu, -1, axis=2) - xv


def action(x, beta):
    """beta * sum(1 - cos theta).  lattice/u1/pytorch/lattice.py:80-86"""
    w = wilson_loops(x)
    return (x.dtype.type(beta) * (1.0 - np.cos(w)).sum((1, 2))).astype(x.dtype)


def grad_action(x, beta):
    """autograd dS/dx of the above (lattice/u1/pytorch/lattice.py:102-117), in closed form
    F0 = beta [sin th - sin th(t, x-1)],  F1 = beta [-sin th + sin th(t-1, x)]."""
    s = np.sin(wilson_loops(x))
    b = x.dtype.type(beta)
    f0 = b * (s - np.roll(s, 1, axis=2))
    f1 = b * (-s + np.roll(s, 1, axis=1))
    return np.stack([f0, f1], axis=1).astype(x.dtype)


def plaqs(x):
    """mean cos theta.  lattice/u1/pytorch/lattice.py:188-203"""
    return np.cos(wilson_loops(x)).mean((1, 2))


def sin_charges(x):
    """sum sin theta / 2pi.  lattice/u1/pytorch/lattice.py:221-224"""
    return np.sin(wilson_loops(x)).sum((1, 2)) / TWO_PI


def project_angle(w):
    """lattice/u1/pytorch/lattice.py:45-47"""
    return w - TWO_PI * np.floor((w + PI) / TWO_PI)


def int_charges(x):
    """lattice/u1/pytorch/lattice.py:226-228"""
    return project_angle(wilson_loops(x)).sum((1, 2)) / TWO_PI


def kinetic_energy(p):
    """0.5 sum p^2.  group/u1/pytorch/group.py:164-165"""
    return 0.5 * (p.reshape(p.shape[0], -1) ** 2).sum(-1)


def group_to_vec(x):
    """cat([cos x, sin x], dim=1).  group/u1/pytorch/group.py:86-89"""
    return np.concatenate([np.cos(x), np.sin(x)], axis=1)
