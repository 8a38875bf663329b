"""``Dynamics.forward`` as a node of a torch.autograd graph.

The reference trains by calling ``dynamics((x, beta))`` in train mode, forming
``LatticeLoss(x_init, mc_states.proposed.x, acc)`` and calling ``loss.backward()`` followed by
``torch.optim.Adam(dynamics.parameters()).step()`` (trainers/pytorch/trainer.py:1266-1367), optionally
through ``GradScaler`` (:1303-1313) and ``DistributedDataParallel`` (:246-257).  Those callers need
outputs with a ``grad_fn`` and ``p.grad`` filled by autograd's own accumulation nodes.

``Transition`` is that node.  Its forward records the trajectory on the tape of
``dynamics/pytorch/training.py`` (every sub-update a HIP kernel, states and activations kept in HBM)
and returns the proposal ``(x', v', sum logdet, acc)``; its backward turns the cotangents of those four
into the seeds of the hand-written reverse sweep (the energy terms of ``acc`` are differentiated
here: kinetic energy -> v', Wilson action -> ``l2q_*_plaq_bwd``), runs the sweep into FRESH gradient
buffers and hands them to autograd as the gradients of the parameters -- so ``p.grad`` accumulation,
``GradScaler.unscale_``, DDP's reducer hooks and ``torch.autograd.grad`` all see what they expect.

The gradient with respect to the input configuration x is not produced (the reference sets
``x.requires_grad_(True)`` but never reads ``x.grad``, trainer.py:1271).
"""
from __future__ import annotations

import weakref
from typing import Optional

import torch

from l2hmc import _autograd as AG
from l2hmc import _ops as ops
from l2hmc.dynamics.pytorch import training as T

Tensor = torch.Tensor


class _Session:
    """Native-order weight shadows (LeapfrogLayer.native_train_begin) shared by the recorded
    trajectories of one optimiser step: begun by the first forward, ended when the last outstanding
    trajectory has been reversed -- or dropped without a backward (its token is collected)."""

    def __init__(self, dyn, nb: int):
        self.nets = T._native_begin(dyn, nb)
        self.stamp = _param_stamp(dyn)
        self.nb = nb
        self.live = 0

    def join(self):
        tok = _Token()
        self.live += 1
        # a trajectory whose graph is dropped without a backward releases through the collector
        weakref.finalize(tok, self.release, tok.state)
        return tok

    def release(self, state: dict) -> None:
        if state['released']:
            return
        state['released'] = True
        self.live -= 1
        if self.live == 0:
            for n in self.nets:
                n.native_train_end(scatter=False)       # (gradients were flushed by each backward)
            self.nets = []


class _Token:
    __slots__ = ('__weakref__', 'state')

    def __init__(self):
        self.state = {'released': False, 'done': False}


def _param_stamp(dyn) -> tuple:
    return (ops.PARAM_GENERATION[0], sum(p._version for p in dyn.parameters()))


def trained_parameters(dyn) -> list:
    """The parameters a trajectory can send a gradient to, in `dyn.parameters()` order: everything that
    requires a gradient except the SU(3) xnet (built, never called: dynamics.py:1420-1425 -- `grad`
    stays None like under the reference's autograd, hence its find_unused_parameters=True)."""
    skip = set()
    if dyn.group == 'SU3' and dyn._networks_built:
        skip = {id(p) for p in dyn.xnet.parameters()}
    return [p for p in dyn.parameters() if p.requires_grad and id(p) not in skip]


def _session(dyn, nb: int) -> _Session:
    s: Optional[_Session] = getattr(dyn, '_ag_session', None)
    if s is not None and s.live > 0 and s.stamp == _param_stamp(dyn):
        if s.nb != nb:
            # the deferred-gradient arenas of the live session are keyed and sized by its batch: a second
            # batch size would re-key them under the first tape's slots
            raise RuntimeError(f'Dynamics.forward: a recorded trajectory of {s.nb} chains has not been reversed '
                               f'yet; reverse (or drop) it before recording one of {nb} chains')
        return s
    if s is not None and s.live > 0:
        # parameters changed under a recorded trajectory that was never reversed: its shadows are stale
        for n in s.nets:
            n.native_train_end(scatter=False)
        s.nets = []
    s = _Session(dyn, nb)
    dyn._ag_session = s
    return s


class Transition(torch.autograd.Function):
    """(x, *parameters) -> (x_prop, v_prop, sumlogdet, acc) of one L2HMC trajectory in train mode."""

    @staticmethod
    def forward(ctx, dyn, x: Tensor, beta: float, direction: Optional[bool], side: dict, *params):
        nb = x.shape[0]
        xn = dyn._pack(x.detach())
        vn = dyn._momentum_n(nb)
        sess = _session(dyn, nb)
        tok = sess.join()
        if dyn.config.merge_directions:
            x_, v_, hist, tape = T.trajectory_fb_train(dyn, xn, vn, beta)
        else:
            x_, v_, hist, tape = T.trajectory_train(dyn, xn, vn, beta, bool(direction))
        ctx.dyn, ctx.tape, ctx.sess, ctx.tok, ctx.beta = dyn, tape, sess, tok, beta
        ctx.x_, ctx.v_, ctx.acc, ctx.sld_fwd = x_, v_, hist['acc'], hist['sumlogdet']
        ctx.params = params
        ctx.set_materialize_grads(False)
        # what the caller sees: reference layout, plus the native originals for the wrappers
        # (U(1): the reference layout IS the native one -- hand out copies, so that an in-place wrap / projection
        # of the proposal before loss.backward() cannot corrupt the fields the reverse sweep differentiates)
        su3 = dyn.group == 'SU3'
        xp = AG.attach_native(dyn._unpack(x_) if su3 else dyn._unpack(x_).clone(), x_)
        vp = dyn._unpack(v_) if su3 else v_.reshape(nb, -1).clone()
        # native originals / per-step history for the caller (Dynamics._forward_train), not kept here
        side.update({'xn': xn, 'vn': vn, 'x_': x_, 'v_': v_, 'hist': hist})
        return xp, vp, hist['sumlogdet'].clone(), hist['acc'].clone()

    @staticmethod
    def backward(ctx, gxp, gvp, gsld, gacc):
        dyn, tape, b = ctx.dyn, ctx.tape, ctx.beta
        if ctx.tok.state['done']:
            raise RuntimeError('Dynamics.forward: this trajectory has been reversed already (its tape is '
                               'released after the first backward; retain_graph is not supported)')
        x_, v_, acc = ctx.x_, ctx.v_, ctx.acc
        nb = x_.shape[0]
        su3 = dyn.group == 'SU3'
        rdt = torch.float64 if su3 else x_.dtype
        dev = x_.device
        # ---- seeds: cotangents of (x', v', sum logdet) including what arrives through acc =
        # exp(min(0, dh)), dh = H_init - H(x', v') + sum logdet (swapped for the single-direction kernel)
        gl = torch.zeros(nb, dtype=rdt, device=dev) if gsld is None else gsld.to(rdt).clone()
        gx = torch.zeros_like(x_) if gxp is None else dyn._pack(gxp)
        if gvp is None:
            gv = torch.zeros_like(v_)
        else:
            gv = dyn._pack(gvp) if su3 else gvp.to(rdt).reshape(v_.shape).clone()
        if gacc is not None:
            swapped = bool(getattr(tape, 'swapped', False))
            h0, h1 = tape.h_init.to(rdt), tape.h_prop.to(rdt)
            dh = (h1 - h0 if swapped else h0 - h1) + ctx.sld_fwd.to(rdt)
            g_dh = torch.where(dh < 0, gacc.to(rdt) * acc.to(rdt), torch.zeros_like(dh))
            g_h = g_dh if swapped else -g_dh                      # cotangent of H(x', v')
            gl = gl + g_dh
            if su3:
                c1 = dyn.potential_c1
                w = torch.zeros(nb, 6, 2, dtype=torch.float64, device=dev)
                w[:, :, 0] = ((-b * (1.0 - 8.0 * c1) / 3.0) * g_h).reshape(nb, 1)
                ops.su3_plaq_bwd_(gx, x_, w, dyn.latvolume)
                if c1 != 0.0:
                    ops.su3_rect_bwd_(gx, x_, (-b * c1 / 3.0) * g_h, dyn.latvolume)
                ops.axpy_rows_(torch.view_as_real(gv), g_h.contiguous(), torch.view_as_real(v_))
            else:
                z = torch.zeros(nb, dtype=rdt, device=dev)
                ops.u1_plaq_bwd_(gx.reshape(x_.shape), x_, (-b) * g_h, z, dyn.latvolume)
                ops.axpy_rows_(gv.reshape(nb, -1), g_h.contiguous(), v_.reshape(nb, -1))
        # ---- reverse sweep into fresh gradient buffers; autograd accumulates them into p.grad
        params = ctx.params
        saved = [p.grad for p in params]
        try:
            for p in params:
                p.grad = torch.zeros_like(p)
            T.backward(dyn, tape, gx, gv, gl.contiguous(), b)
            for n in ctx.sess.nets:
                n.native_train_end(keep_active=True)          # flush the native-order gradients
            grads = [p.grad for p in params]
        finally:
            for p, g in zip(params, saved):
                p.grad = g
            ctx.tok.state['done'] = True
            ctx.sess.release(ctx.tok.state)
            ctx.tape = ctx.x_ = ctx.v_ = None
        return (None, None, None, None, None, *grads)


def transition(dyn, x: Tensor, beta, direction: Optional[bool] = None, side: Optional[dict] = None):
    """Train-mode trajectory with an autograd graph: (x_prop, v_prop, sumlogdet, acc), reference
    layout, each with a grad_fn that leads to `dyn`'s parameters.  `side` receives the native-layout
    states and the per-step history of the trajectory."""
    from l2hmc.dynamics.pytorch.dynamics import _beta
    params = trained_parameters(dyn)
    if not params:
        raise RuntimeError('Dynamics.forward (train mode): no parameter requires a gradient')
    return Transition.apply(dyn, x, _beta(beta), direction, {} if side is None else side, *params)
