"""Training-gradient path of the U(1) L2HMC sampler: one merged forward/backward trajectory
recorded on a tape, the loss, and the reverse sweep that turns the loss cotangents into
parameter gradients -- what ``loss.backward()`` does in the reference's
``Trainer.train_step`` (trainers/pytorch/trainer.py:1284-1367) through torch.autograd.

There is no autograd graph here.  Each sub-update of the integrator (dynamics.py:1187-1477)
has a hand-written cotangent kernel in ``csrc/train_kernels.hip`` (``l2q_v_update_bwd``,
``l2q_u1_x_update_bwd``, ``l2q_u1_force_bwd`` -- the reference differentiates *through* the
force, ``create_graph=True`` -- ``l2q_u1_masked_cos_sin_bwd``, ``l2q_u1_plaq_bwd``) and the
networks have explicit backward passes (network.py ``LeapfrogLayer.backward``).  The host keeps
the tape: per sub-update the states it started from and the network activations
(HBM is 288 GB; a 64 x 64 / 8192-chain trajectory tape is ~30 GB).  Only the per-chain scalars
of the loss / acceptance probability ([nb] vectors) go through torch.autograd.

Parameters and gradients live in one flat arena per dtype (``ParamArena``): the optimiser is a
single fused Adam launch and the data-parallel gradient exchange a single all-reduce.

SU(3) uses the same tape with the cotangent kernels of ``csrc/su3_train_kernels.hip``:
``l2q_su3_expm_mul_bwd`` (Frechet derivative of the matrix exponential), ``l2q_su3_projsu_vec8_bwd``
(polar-projection derivative solved in a Jacobi eigenbasis), ``l2q_su3_force_bwd`` (the reference
differentiates the force only through the ``x^H`` factor of ``TAH(dS/dx x^H)``,
lattice/su3/pytorch/lattice.py:299-308), ``l2q_su3_plaq_bwd`` and ``l2q_v_update_bwd_c128``.
The vnet runs on reference-ordered 8-component inputs (transposes at its boundary) so that the
ordinary ``LeapfrogLayer.backward`` accumulates into the checkpoint-ordered parameters.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from l2hmc import DEVICE
from l2hmc import _ops as ops

Tensor = torch.Tensor


# ------------------------------------------------------------------------------ arena
class ParamArena:
    """All trainable parameters of a module re-homed into ONE contiguous buffer per dtype,
    with a matching flat gradient buffer (each ``p.grad`` is a view of it) and flat Adam
    moments.  MI355X-first replacement of per-tensor optimiser loops and bucketed DDP:
    one `l2q_adam` launch, one RCCL all-reduce."""

    def __init__(self, module: nn.Module, skip=None):
        """skip: parameters that can never receive a gradient (the SU(3) xnet is built but never
        called, dynamics.py:1420-1425): they stay outside the arena, keep ``grad = None`` like
        under the reference's autograd, and cost no gradient / moment memory (2/3 of all
        parameters at SU(3) 8^4, units [256])."""
        self.groups: dict = {}
        skip_ids = {id(p) for p in (skip or ())}
        params = [p for p in module.parameters() if p.requires_grad and id(p) not in skip_ids]
        by_dtype: dict = {}
        for p in params:
            by_dtype.setdefault(p.dtype, []).append(p)
        for dt, ps in by_dtype.items():
            # every parameter starts on a 16-byte boundary (vector loads / transposed-operand
            # GEMMs read the views directly); the padding stays zero in all four buffers
            q = 16 // torch.empty((), dtype=dt).element_size()
            n = sum(-(-p.numel() // q) * q for p in ps)
            flat = torch.zeros(n, dtype=dt, device=DEVICE)
            grad = torch.zeros(n, dtype=dt, device=DEVICE)
            off = 0
            with torch.no_grad():
                for p in ps:
                    k = p.numel()
                    flat[off:off + k].copy_(p.detach().reshape(-1).to(DEVICE))
                    p.data = flat[off:off + k].view(p.shape)
                    p.grad = grad[off:off + k].view(p.shape)
                    off += -(-k // q) * q
            self.groups[dt] = {'params': ps, 'flat': flat, 'grad': grad,
                               'm': torch.zeros_like(flat), 'v': torch.zeros_like(flat)}
        self.step_count = 0
        ops.PARAM_GENERATION[0] += 1

    def reset_state(self) -> None:
        """forget the Adam moments and the step counter (a fresh optimiser)"""
        for g in self.groups.values():
            g['m'].zero_()
            g['v'].zero_()
        self.step_count = 0

    def zero_grad(self) -> None:
        for g in self.groups.values():
            g['grad'].zero_()

    def numel(self) -> int:
        """number of trained parameters (without the alignment padding of the flat buffers)"""
        return sum(p.numel() for g in self.groups.values() for p in g['params'])

    def all_reduce(self, comm=None, force: bool = False) -> float:
        """Sum the flat gradients over ranks (RCCL over xGMI: one collective per dtype);
        returns the 1/world factor to fold into the optimiser step (DDP averages).
        `comm`: a `utils.dist.NativeComm` (l2q_allreduce_grads through the C ABI) instead of the
        torch.distributed group; `force`: run the collective even with a single rank (the RCCL
        smoke test / bench probe of a 1-GPU box)."""
        import torch.distributed as dist
        if comm is not None:
            if comm.world_size == 1 and not force:
                return 1.0
            for g in self.groups.values():
                comm.all_reduce_(g['grad'])
            return 1.0 / comm.world_size
        if not (dist.is_available() and dist.is_initialized()):
            return 1.0
        if dist.get_world_size() == 1 and not force:
            return 1.0
        for g in self.groups.values():
            dist.all_reduce(g['grad'], op=dist.ReduceOp.SUM)
        return 1.0 / dist.get_world_size()

    def grad_norm(self, scale: float = 1.0) -> float:
        tot = 0.0
        for g in self.groups.values():
            tot += float(ops.sumsq(g['grad']).item())
        return scale * tot ** 0.5

    # ---- torch.optim.Adam-compatible (de)serialisation (`optimizer_state_dict` of the
    # reference's checkpoints, trainers/pytorch/trainer.py:573-614)
    def _param_slices(self):
        for g in self.groups.values():
            base = g['flat'].data_ptr()
            for p in g['params']:
                off = (p.data_ptr() - base) // g['flat'].element_size()
                yield p, g, off, p.numel()

    def state_dict(self, params_in_order, lr: float, betas=(0.9, 0.999), eps: float = 1e-8) -> dict:
        """Layout of torch.optim.Adam(params_in_order).state_dict() after `step_count` steps."""
        index = {id(p): i for i, p in enumerate(params_in_order)}
        state = {}
        if self.step_count > 0:
            for p, g, off, n in self._param_slices():
                if id(p) not in index:
                    continue
                state[index[id(p)]] = {
                    'step': torch.tensor(float(self.step_count)),
                    'exp_avg': g['m'][off:off + n].view(p.shape).clone(),
                    'exp_avg_sq': g['v'][off:off + n].view(p.shape).clone()}
        group = {'lr': lr, 'betas': tuple(betas), 'eps': eps, 'weight_decay': 0, 'amsgrad': False,
                 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False,
                 'fused': None, 'decoupled_weight_decay': False,
                 'params': list(range(len(params_in_order)))}
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd: dict, params_in_order) -> None:
        """Accepts a torch.optim.Adam state_dict over the same parameter order (parameters the
        optimiser never stepped, e.g. the SU(3) xnet, have no entry and keep zero moments)."""
        steps = {int(float(v['step'])) for v in sd['state'].values()}
        if len(steps) > 1:
            raise ValueError(f'per-parameter step counts differ: {sorted(steps)}')
        self.step_count = steps.pop() if steps else 0
        where = {id(p): (g, off, n) for p, g, off, n in self._param_slices()}
        for g in self.groups.values():
            g['m'].zero_()
            g['v'].zero_()
        for i, st in sd['state'].items():
            p = params_in_order[int(i)]
            if id(p) not in where:
                continue                      # a parameter this build never trains
            g, off, n = where[id(p)]
            g['m'][off:off + n].copy_(st['exp_avg'].reshape(-1).to(g['m']))
            g['v'][off:off + n].copy_(st['exp_avg_sq'].reshape(-1).to(g['v']))

    def adam_step(self, lr: float, betas=(0.9, 0.999), eps: float = 1e-8,
                  grad_scale: float = 1.0) -> None:
        self.step_count += 1
        for g in self.groups.values():
            ops.adam_step_(g['flat'], g['grad'], g['m'], g['v'], lr, betas[0], betas[1], eps,
                           self.step_count, grad_scale)
        ops.PARAM_GENERATION[0] += 1


class GradReducer:
    """The data-parallel gradient exchange of ONE optimiser step, overlapped with the reverse sweep.

    The reference gets this from DistributedDataParallel: bucketed all-reduces that start while
    `loss.backward()` is still running (trainers/pytorch/trainer.py:246-257, 1296-1304).  Here the
    gradients live in the flat `ParamArena`; whenever the sweep has FINISHED a set of parameters --
    a network whose last tape entry has been reversed, a weight matrix whose deferred-gradient GEMM
    and native-order scatter are done -- `ready()` starts the all-reduce of their contiguous arena
    range on the communication stream (RCCL over xGMI), and the sweep carries on.  `finish()` reduces
    what is left in as few contiguous ranges as possible and makes the compute stream wait for all
    of it: only the last slab's exchange is exposed.  Sums are element-wise, so the result is the one
    blocking all-reduce's bit for bit."""

    _paths_checked = False          # the once-per-process comparison of the ranks' memory-gated paths
    _paths_differ = False

    def __init__(self, arena: 'ParamArena', comm=None, force: bool = False):
        import torch.distributed as dist
        self.arena, self.comm = arena, comm
        self.world = 1
        if comm is not None:
            self.world = comm.world_size
        elif dist.is_available() and dist.is_initialized():
            self.world = dist.get_world_size()
        self.active = self.world > 1 or force
        self.done: dict = {dt: [] for dt in arena.groups}        # dtype -> disjoint [lo, hi) ranges already launched
        self.handles: list = []
        self.side = None
        self.launched = 0                                        # collectives started before finish()
        self.blocking = False                                    # ranks disagree on their paths: one exchange in finish()
        self._where = None
        self._slices: dict = {}

    def _ranges(self, params) -> dict:
        if self._where is None:
            self._where = {id(p): (g['flat'].dtype, off, n) for p, g, off, n in self.arena._param_slices()}
            for dt, off, n in self._where.values():
                self._slices.setdefault(dt, []).append((off, off + n))
            for v in self._slices.values():
                v.sort()
        out: dict = {}
        for p in params:
            w = self._where.get(id(p))
            if w is not None:
                out.setdefault(w[0], []).append((w[1], w[1] + w[2]))
        return out

    def _gap_is_padding(self, dt, a: int, b: int) -> bool:
        """no arena parameter of this dtype lies in [a, b): the gap is alignment padding (zeros)"""
        import bisect
        sl = self._slices.get(dt, [])
        i = bisect.bisect_left(sl, (a, a))
        if i > 0 and sl[i - 1][1] > a:
            return False
        return not (i < len(sl) and sl[i][0] < b)

    def _merge(self, dt, ranges: list, slack: int) -> list:
        """sorted, with neighbours joined when the gap between them is nothing but the arena's alignment
        padding (at most `slack` elements that belong to no parameter); overlapping ranges are joined too"""
        merged: list = []
        for lo, hi in sorted(ranges):
            if merged and (lo <= merged[-1][1]
                           or (lo <= merged[-1][1] + slack and self._gap_is_padding(dt, merged[-1][1], lo))):
                merged[-1][1] = max(merged[-1][1], hi)
            else:
                merged.append([lo, hi])
        return [(a, b) for a, b in merged]

    def _uncovered(self, dt, lo: int, hi: int) -> list:
        """the parts of [lo, hi) that no launched range covers (every element is exchanged exactly once)"""
        out, pos = [], lo
        for a, b in sorted(self.done[dt]):
            if b <= pos:
                continue
            if a >= hi:
                break
            if a > pos:
                out.append((pos, a))
            pos = max(pos, b)
        if pos < hi:
            out.append((pos, hi))
        return out

    def _check_paths_once(self) -> None:
        """Before the FIRST overlapped collective of the process: every rank reports which memory-gated paths it
        took (`ops.mem_gate_signature`; the gates themselves already decide by consensus).  If the tables differ
        the number, order and size of the per-range collectives could differ as well: this step and all later
        ones then use ONE blocking exchange of the whole arena in `finish()`, which is rank-invariant."""
        if GradReducer._paths_checked:
            self.blocking = GradReducer._paths_differ
            return
        GradReducer._paths_checked, GradReducer._paths_differ = True, False
        import torch.distributed as dist
        if self.comm is None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            sigs = [None] * dist.get_world_size()
            dist.all_gather_object(sigs, ops.mem_gate_signature())
            if len(set(sigs)) > 1:
                import logging
                logging.getLogger('l2hmc').warning(
                    'ranks took different memory-gated paths %s: gradient exchange falls back to one blocking '
                    'all-reduce per step', sigs)
                GradReducer._paths_differ = True
        self.blocking = GradReducer._paths_differ

    def _launch(self, view: Tensor) -> None:
        import torch.distributed as dist
        if self.comm is not None:
            # C-ABI route: the collective runs on a side stream that waits for the producer kernels
            if view.is_cuda:
                if self.side is None:
                    self.side = torch.cuda.Stream(device=view.device)
                self.side.wait_stream(torch.cuda.current_stream(view.device))
                with torch.cuda.stream(self.side):
                    self.comm.all_reduce_(view)
            else:
                self.comm.all_reduce_(view)
            return
        if dist.is_available() and dist.is_initialized():
            self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))

    def ready(self, params) -> None:
        """the gradients of `params` are final for this step: start their exchange"""
        if not self.active:
            return
        if self.launched == 0 and not self.blocking:
            self._check_paths_once()
        if self.blocking:
            return
        for dt, rs in self._ranges(params).items():
            g = self.arena.groups[dt]
            q = 16 // g['flat'].element_size()
            for lo, hi in self._merge(dt, rs, q):
                for a, b in self._uncovered(dt, lo, hi):
                    self._launch(g['grad'][a:b])
                    self.done[dt].append((a, b))
                    self.launched += 1

    def finish(self) -> float:
        """exchange the rest, wait for everything; returns the 1 / world factor for the optimiser"""
        if not self.active:
            return 1.0
        for dt, g in self.arena.groups.items():
            n = g['grad'].numel()
            pos = 0
            for lo, hi in self._merge(dt, self.done[dt], 0) + [(n, n)]:
                if lo > pos:
                    self._launch(g['grad'][pos:lo])
                pos = max(pos, hi)
        for h in self.handles:
            h.wait()
        self.handles = []
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        return 1.0 / self.world


class LossScaler:
    """torch.amp.GradScaler's bookkeeping (defaults included: init 2^16, growth x2 every 2000 clean steps,
    backoff x0.5) for the flat-arena trainer: the reference scales the loss of its mixed-precision steps,
    un-scales the gradients, SKIPS the optimiser step when any of them is inf / nan and adapts the scale
    (trainers/pytorch/trainer.py:256-257, 1303-1313).  The scale is a power of two, so scaling the seeds of
    the fp32 reverse sweep and folding 1 / scale into the fused Adam changes no bit unless something
    overflows.  (The reference's 16-bit backward overflows earlier than this fp32 sweep: its first steps at
    2^16 are typically skipped, tests/golden/u1_train_fp16_overflow.npz; here they are taken.)"""

    def __init__(self, init_scale: float = 2.0 ** 16, growth_factor: float = 2.0,
                 backoff_factor: float = 0.5, growth_interval: int = 2000, enabled: bool = True):
        self.scale = float(init_scale)
        self.growth_factor, self.backoff_factor = float(growth_factor), float(backoff_factor)
        self.growth_interval, self.enabled = int(growth_interval), bool(enabled)
        self.growth_tracker = 0
        self.skipped = 0

    def get_scale(self) -> float:
        return self.scale if self.enabled else 1.0

    def update(self, found_inf: bool) -> None:
        if not self.enabled:
            return
        if found_inf:
            self.scale *= self.backoff_factor
            self.growth_tracker = 0
            self.skipped += 1
        else:
            self.growth_tracker += 1
            if self.growth_tracker == self.growth_interval:
                self.scale *= self.growth_factor
                self.growth_tracker = 0

    def state_dict(self) -> dict:
        return {'scale': self.scale, 'growth_factor': self.growth_factor,
                'backoff_factor': self.backoff_factor, 'growth_interval': self.growth_interval,
                '_growth_tracker': self.growth_tracker}

    def load_state_dict(self, sd: dict) -> None:
        self.scale = float(sd['scale'])
        self.growth_factor, self.backoff_factor = float(sd['growth_factor']), float(sd['backoff_factor'])
        self.growth_interval = int(sd['growth_interval'])
        self.growth_tracker = int(sd['_growth_tracker'])


# ------------------------------------------------------------------------------ forward (tape)
def _eps_and_slope(p: Tensor) -> tuple[float, float]:
    """eps = sigmoid(log p) = p / (1 + p) and d eps / d p = 1 / (1 + p)^2 (dynamics.py:82-83)"""
    pv = float(p.detach().cpu())
    return pv / (1.0 + pv), 1.0 / (1.0 + pv) ** 2


class Tape:
    def __init__(self):
        self.entries: list = []
        self.eps_grads: dict = {}          # ('x'|'v', step) -> list of [nb] tensors


def _v_step(dyn, tape: Tape, st: int, x: Tensor, v: Tensor, beta: float, forward: bool):
    nb = x.shape[0]
    eps = dyn._eps('v', st)
    vnet = dyn._get_vnet(st)
    F = ops.u1_force(x, beta, dyn.latvolume)
    s, t, q, ctx = vnet.forward_train(x, F)
    v_new, ld = ops.v_update(v, F.reshape(nb, -1), s, t, q, eps, forward)
    tape.entries.append({'kind': 'v', 'step': st, 'forward': forward, 'x': x, 'v': v, 'F': F,
                         's': s, 't': t, 'q': q, 'ctx': ctx, 'net': vnet, 'eps': eps})
    return v_new, ld


def _x_step(dyn, tape: Tape, st: int, x: Tensor, v: Tensor, mask: Tensor, complement: bool,
            forward: bool, first: bool):
    nb = x.shape[0]
    eps = dyn._eps('x', st)
    xnet = dyn._get_xnet(st, first)
    xm = ops.u1_masked_cos_sin(x, mask, complement, dyn.latvolume)
    s, t, q, ctx = xnet.forward_train(xm, v)
    x_new = x.clone()
    ld = ops.u1_x_update_(x_new.reshape(nb, -1), v, s, t, q, mask, complement, eps, forward,
                          dyn.config.use_ncp)
    tape.entries.append({'kind': 'x', 'step': st, 'forward': forward, 'x': x, 'v': v, 's': s,
                         't': t, 'q': q, 'ctx': ctx, 'net': xnet, 'mask': mask,
                         'complement': complement, 'eps': eps})
    return x_new, ld


def _v_step_su3(dyn, tape: Tape, st: int, x: Tensor, v: Tensor, beta: float, forward: bool,
                share: Optional[dict] = None):
    """x, v native [nb, 4, 9, V] complex128.

    `share` (trajectory-local): the closing v-update of one leapfrog step and the opening one of
    the next (also across the momentum flip) act on the SAME x, hence -- with one shared vnet,
    the SU(3) default -- on the same force and the same (s, t, q).  The second of such a pair
    reuses the first one's force / network evaluation (the eval path does the same,
    Dynamics.reuse_v_inputs) and is recorded as a tape entry that points at its `primary`: the
    reverse sweep adds its (dF, ds, dt, dq) cotangents to the primary's before ONE backward pass
    through the network, the projections and the force.  9 instead of 16 network / force
    evaluations forward AND backward in a merged nleapfrog = 4 trajectory; exact (the cotangents
    are summed either way)."""
    nb, _, _, V = x.shape
    eps = dyn._eps('v', st)
    vnet = dyn._get_vnet(st)
    hit = (share is not None and share.get('x') is x and share.get('net') is vnet
           and not vnet.training_needs_fresh_forward())
    if hit:
        F, sn, tn, qn = share['F'], share['s'], share['t'], share['q']
        v_new, ld = ops.v_update(v, F.reshape(nb, -1), sn, tn, qn, eps, forward)
        tape.entries.append({'kind': 'v', 'step': st, 'forward': forward, 'x': x, 'v': v, 'F': F,
                             's': sn, 't': tn, 'q': qn, 'ctx': None, 'net': vnet, 'eps': eps,
                             'primary': share['idx']})
        return v_new, ld
    F = ops.su3_force_n(x, beta, dyn.latvolume)
    # native-order training: the network inputs of all calls of the step side by side in one arena, so that the
    # weight gradients are ONE GEMM per matrix at the end of the reverse sweep (LeapfrogLayer.defer_slot)
    slot = None
    if vnet.native_active() and getattr(dyn, 'defer_weight_grads', True):
        slot = vnet.defer_slot(nb, 32 * V, 32 * V, getattr(tape, 'defer_cap', 0))
    if slot is not None:
        xv = ops.su3_projsu_vec8_n(x, out=slot[1]).reshape(nb, -1)
        fv = ops.su3_projsu_vec8_n(F, out=slot[2]).reshape(nb, -1)
    else:
        xv = ops.su3_projsu_vec8_n(x).reshape(nb, -1)
        fv = ops.su3_projsu_vec8_n(F).reshape(nb, -1)
    sie = ops.SLICED_INPUT_EXP if getattr(dyn, 'sliced_train_input', False) else None  # |vec8(projectSU(.))| < 4
    if (vnet.native_active() and getattr(dyn, 'sliced_train_heads', True) and ops.USE_SLICED_HEADS[0]
            and vnet.sliced_train_image() is not None):
        # the heads and this v-update in one launch on the int8 matrix cores (TAPE kernel: s, t, q are
        # stored for the reverse sweep); the image is rebuilt once per optimiser step
        z, ctx = vnet.forward_train(xv, fv, hidden_only=True, sliced_input_exp=sie)
        if slot is not None:
            ctx['defer_idx'] = slot[0]
        v_new, ld, sn, tn, qn = vnet.heads_vupdate_train_sliced(z, ctx, v.reshape(nb, -1), F.reshape(nb, -1),
                                                                 eps, forward)
        v_new = v_new.reshape(v.shape)
        tape.entries.append({'kind': 'v', 'step': st, 'forward': forward, 'x': x, 'v': v, 'F': F,
                             's': sn, 't': tn, 'q': qn, 'ctx': ctx, 'net': vnet, 'eps': eps})
        if share is not None:
            share.clear()
            share.update({'x': x, 'net': vnet, 'F': F, 's': sn, 't': tn, 'q': qn,
                          'idx': len(tape.entries) - 1})
        return v_new, ld
    if vnet.native_active():
        # native-order weight shadows (LeapfrogLayer.native_train_begin): no activation transposes
        sn, tn, qn, ctx = vnet.forward_train(xv, fv, sliced_input_exp=sie)
        if slot is not None:
            ctx['defer_idx'] = slot[0]
    else:
        # the network sees the reference's entry order (mu, site, component)
        s, t, q, ctx = vnet.forward_train(ops.unpack_entries(xv, V, 8), ops.unpack_entries(fv, V, 8))
        sn, tn, qn = (ops.pack_entries(a, V, 9) for a in (s, t, q))
    v_new, ld = ops.v_update(v, F.reshape(nb, -1), sn, tn, qn, eps, forward)
    tape.entries.append({'kind': 'v', 'step': st, 'forward': forward, 'x': x, 'v': v, 'F': F,
                         's': sn, 't': tn, 'q': qn, 'ctx': ctx, 'net': vnet, 'eps': eps})
    if share is not None:
        share.clear()
        share.update({'x': x, 'net': vnet, 'F': F, 's': sn, 't': tn, 'q': qn,
                      'idx': len(tape.entries) - 1})
    return v_new, ld


def _x_step_su3(dyn, tape: Tape, st: int, x: Tensor, v: Tensor, mask: Tensor, complement: bool,
                forward: bool):
    """x' = keep (.) x + expm(+-eps v) @ ((1 - keep) (.) x); no network, logdet = 0
    (dynamics.py:1420-1425, 1468-1474)."""
    eps = dyn._eps('x', st)
    seps = eps if forward else -eps
    x_new = ops.su3_expm_mul_n(x, v, seps, mask, complement)
    tape.entries.append({'kind': 'x', 'step': st, 'forward': forward, 'x': x, 'v': v,
                         'mask': mask, 'complement': complement, 'eps': seps})
    return x_new


def _lf_train(dyn, tape: Tape, step: int, x, v, beta, forward: bool, share: Optional[dict] = None):
    """One generalised leapfrog step (dynamics.py:1187-1228), functional (new tensors)."""
    if forward:
        st, order = step, ((False, True), (True, False))         # (complement, first)
    else:
        st = dyn.config.nleapfrog - step - 1
        order = ((True, False), (False, True))
    m = dyn._native_masks()[st]
    if dyn.group == 'SU3':
        v, ld = _v_step_su3(dyn, tape, st, x, v, beta, forward, share)
        if getattr(dyn, 'fuse_x_halves_train', True):
            # both masked half-updates in one kernel forward (ONE expm(eps v)) and one kernel in the reverse
            # sweep (one Frechet derivative for both: it is linear in its direction)
            eps = dyn._eps('x', st)
            seps = eps if forward else -eps
            x_new = ops.su3_expm_mul2_n(x, v, seps, m, order[0][0])
            tape.entries.append({'kind': 'x2', 'step': st, 'forward': forward, 'x': x, 'v': v, 'mask': m,
                                 'complement_first': order[0][0], 'eps': seps})
            x = x_new
        else:
            for comp, _first in order:
                x = _x_step_su3(dyn, tape, st, x, v, m, comp, forward)
        v, l = _v_step_su3(dyn, tape, st, x, v, beta, forward, share)
        return x, v, ld + l
    v, ld = _v_step(dyn, tape, st, x, v, beta, forward)
    for comp, first in order:
        x, l = _x_step(dyn, tape, st, x, v, m, comp, forward, first)
        ld = ld + l
    v, l = _v_step(dyn, tape, st, x, v, beta, forward)
    return x, v, ld + l


def trajectory_fb_train(dyn, xn: Tensor, vn: Tensor, beta: float):
    """Merged forward + backward trajectory (dynamics.py:956-1029) recording the tape.
    Returns (x_prop, v_prop, history, tape)."""
    if not dyn._networks_built:
        raise RuntimeError('training needs networks (Dynamics(network_factory=...))')
    tape = Tape()
    nb = xn.shape[0]
    x, v = xn, vn
    sumlogdet = dyn._zeros_nb(nb)
    h_init = dyn._hamiltonian_n(xn, vn, beta)
    history: dict = {}
    verbose = dyn.config.verbose
    sldf = torch.zeros_like(sumlogdet)
    sldb = torch.zeros_like(sumlogdet)
    if verbose:
        dyn.update_history({'energy': h_init, 'logprob': h_init - sumlogdet, 'logdet': sumlogdet,
                            'sldf': sldf, 'sldb': sldb, 'sld': sumlogdet,
                            'xeps': dyn.xeps[0], 'veps': dyn.veps[0]}, history)
    nlf = dyn.config.nleapfrog
    share = {} if (dyn.group == 'SU3' and getattr(dyn, 'reuse_v_inputs', True)) else None
    # network calls of this trajectory -- arenas for the deferred weight gradients only with ONE shared vnet (the
    # SU(3) default): separate networks see four calls each, not worth an arena per network
    one_net = len({id(dyn._get_vnet(s)) for s in range(nlf)}) == 1
    tape.defer_cap = ((2 * nlf + 1) if share is not None else 4 * nlf) if one_net else 0
    for step in range(nlf):
        x, v, ld = _lf_train(dyn, tape, step, x, v, beta, True, share)
        sumlogdet = sumlogdet + ld
        if verbose:
            sldf = sldf + ld
            dyn.update_history(dyn._metrics_n(x, v, beta, sumlogdet, step,
                                              {'sldf': sldf, 'sldb': sldb, 'sld': sumlogdet}),
                               history)
    v = ops.scale(v, -1.0) if dyn.group == 'SU3' else -v
    tape.entries.append({'kind': 'flip'})
    for step in range(nlf):
        x, v, ld = _lf_train(dyn, tape, step, x, v, beta, False, share)
        sumlogdet = sumlogdet + ld
        if verbose:
            sldb = sldb + ld
            dyn.update_history(dyn._metrics_n(x, v, beta, sumlogdet, nlf - step - 1,
                                              {'sldf': torch.zeros_like(sldb), 'sldb': sldb,
                                               'sld': sumlogdet}), history)
    h_prop = dyn._hamiltonian_n(x, v, beta)
    acc = dyn._accept_prob_n(h_init, h_prop, sumlogdet)
    history.update({'acc': acc, 'sumlogdet': sumlogdet})
    if verbose:
        history = dyn._stack_history(history)
    tape.h_init, tape.h_prop = h_init, h_prop
    return x, v, history, tape


def trajectory_train(dyn, xn: Tensor, vn: Tensor, beta: float, forward: bool):
    """Single-direction trajectory of `merge_directions=False` (dynamics.py:1031-1063) recording the
    tape: nleapfrog forward (or backward) generalised leapfrog steps, no momentum flip, and the
    accept probability with the reference's SWAPPED arguments (`compute_accept_prob(state_init=
    state, state_prop=sinit)`, :1053-1057): dh = H(final) - H(start) + sumlogdet -- `tape.swapped`
    tells loss_and_seeds.  Returns (x_prop, v_prop, history, tape)."""
    if not dyn._networks_built:
        raise RuntimeError('training needs networks (Dynamics(network_factory=...))')
    tape = Tape()
    nb = xn.shape[0]
    x, v = xn, vn
    sumlogdet = dyn._zeros_nb(nb)
    h_init = dyn._hamiltonian_n(xn, vn, beta)
    history: dict = {}
    verbose = dyn.config.verbose
    if verbose:
        dyn.update_history(dyn._metrics_n(x, v, beta, sumlogdet, None, None), history)
    nlf = dyn.config.nleapfrog
    share = {} if (dyn.group == 'SU3' and getattr(dyn, 'reuse_v_inputs', True)) else None
    one_net = len({id(dyn._get_vnet(s)) for s in range(nlf)}) == 1
    tape.defer_cap = ((nlf + 1) if share is not None else 2 * nlf) if one_net else 0
    for step in range(nlf):
        x, v, ld = _lf_train(dyn, tape, step, x, v, beta, forward, share)
        sumlogdet = sumlogdet + ld
        if verbose:
            dyn.update_history(dyn._metrics_n(x, v, beta, sumlogdet, step, None), history)
    h_fin = dyn._hamiltonian_n(x, v, beta)
    acc = dyn._accept_prob_n(h_fin, h_init, sumlogdet)
    history.update({'acc': acc, 'sumlogdet': sumlogdet})
    if verbose:
        history = dyn._stack_history(history)
    tape.h_init, tape.h_prop = h_init, h_fin
    tape.swapped = True
    return x, v, history, tape


# ------------------------------------------------------------------------------ loss + seeds
def loss_and_seeds(dyn, loss_fn, xn_init: Tensor, x_prop: Tensor, v_prop: Tensor, tape: Tape,
                   sumlogdet: Tensor, beta: float):
    """Loss value (reference: LatticeLoss.calc_loss on (x_init, x_prop, acc)) and the
    cotangents it sends into the trajectory: (loss, gx_prop, gv_prop, g_sumlogdet).
    The lattice-sized reductions are the l2q kernels; only [nb]-vectors see autograd."""
    if dyn.group == 'SU3':
        return _loss_and_seeds_su3(dyn, loss_fn, xn_init, x_prop, v_prop, tape, sumlogdet, beta)
    lat = dyn.latvolume
    V = dyn.volume
    sp = ops.u1_plaq_sums(x_prop, lat)
    si = ops.u1_plaq_sums(xn_init, lat)
    ke = ops.u1_kinetic(v_prop)
    cos_p = sp[:, 0].clone().requires_grad_(True)
    sin_p = sp[:, 1].clone().requires_grad_(True)
    ke_p = ke.clone().requires_grad_(True)
    sld = sumlogdet.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        h_prop = ke_p + beta * (V - cos_p)
        dh = tape.h_init.detach() - h_prop + sld
        if getattr(tape, 'swapped', False):            # single-direction kernel: see trajectory_train
            dh = h_prop - tape.h_init.detach() + sld
        acc = torch.exp(torch.minimum(dh, torch.zeros_like(dh)))
        loss = loss_fn.loss_from_sums(si[:, 0], si[:, 1], cos_p, sin_p, acc)
        g_cos, g_sin, g_ke, g_sld = torch.autograd.grad(loss, [cos_p, sin_p, ke_p, sld],
                                                        allow_unused=True)
    nb = x_prop.shape[0]
    gx = torch.zeros_like(x_prop)
    z = torch.zeros(nb, dtype=x_prop.dtype, device=x_prop.device)
    ops.u1_plaq_bwd_(gx, x_prop, z if g_cos is None else g_cos, z if g_sin is None else g_sin, lat)
    gv = torch.zeros_like(v_prop)
    if g_ke is not None:
        ops.axpy_rows_(gv, g_ke, v_prop)
    gl = (z if g_sld is None else g_sld).to(x_prop.dtype).contiguous()
    return loss.detach(), gx, gv, gl


def _loss_and_seeds_su3(dyn, loss_fn, xn_init, x_prop, v_prop, tape, sumlogdet, beta):
    """SU(3): per-plane plaquette sums (l2q_su3_plaq_planes), |x' - x|^2 (l2q_diff_norm2_reduce)
    and the kinetic energy are the only lattice-sized reductions the loss / acceptance need."""
    lat = dyn.latvolume
    nb = x_prop.shape[0]
    pl_i = ops.su3_plaq_planes_n(xn_init, lat)                       # [nb, 6, 2]
    pl_p = ops.su3_plaq_planes_n(x_prop, lat).clone().requires_grad_(True)
    ke_p = ops.su3_kinetic_n(v_prop).clone().requires_grad_(True)
    d2 = ops.diff_norm2(x_prop, xn_init).clone().requires_grad_(True)
    sld = sumlogdet.detach().clone().requires_grad_(True)
    c1 = dyn.potential_c1
    leaves = [pl_p, ke_p, d2, sld]
    if c1 != 0.0:                       # rectangle term of the improved action (potential only)
        rs_p = ops.su3_rect_sums_n(x_prop, lat).clone().requires_grad_(True)
        leaves.append(rs_p)
    with torch.enable_grad():
        h_prop = ke_p + (-beta * (1.0 - 8.0 * c1) / 3.0) * pl_p[:, :, 0].sum(1)
        if c1 != 0.0:
            h_prop = h_prop + (-beta * c1 / 3.0) * rs_p
        dh = tape.h_init.detach() - h_prop + sld
        if getattr(tape, 'swapped', False):
            dh = h_prop - tape.h_init.detach() + sld
        acc = torch.exp(torch.minimum(dh, torch.zeros_like(dh)))
        loss = loss_fn.loss_from_sums_su3(pl_i, pl_p, d2, acc, nelem=x_prop[0].numel())
        grads = torch.autograd.grad(loss, leaves, allow_unused=True)
    g_pl, g_ke, g_d2, g_sld = grads[:4]
    gx = torch.zeros_like(x_prop)
    if g_pl is not None:
        ops.su3_plaq_bwd_(gx, x_prop, g_pl, lat)
    if c1 != 0.0 and grads[4] is not None:
        ops.su3_rect_bwd_(gx, x_prop, grads[4], lat)
    if g_d2 is not None:
        ops.diff_bwd_(gx, x_prop, xn_init, g_d2)
    gv = torch.zeros_like(v_prop)
    if g_ke is not None:                                             # KE = 1/2 sum (|v|^2 - 8)
        ops.axpy_rows_(torch.view_as_real(gv), g_ke, torch.view_as_real(v_prop))
    z = torch.zeros(nb, dtype=torch.float64, device=x_prop.device)
    gl = (z if g_sld is None else g_sld).to(torch.float64).contiguous()
    return loss.detach(), gx, gv, gl


# ------------------------------------------------------------------------------ reverse sweep
def _first_use(tape: Tape) -> dict:
    """tape index -> networks whose FIRST recorded call sits there: once the reverse sweep has passed that
    entry, those networks' gradients are final for this trajectory"""
    first: dict = {}
    for i, e in enumerate(tape.entries):
        n = e.get('net')
        if n is not None and id(n) not in first:
            first[id(n)] = (i, n)
    out: dict = {}
    for i, n in first.values():
        out.setdefault(i, []).append(n)
    return out


def backward(dyn, tape: Tape, gx: Tensor, gv: Tensor, gl: Tensor, beta: float, on_final=None) -> None:
    """Replay the tape in reverse, accumulating every parameter's .grad (networks, xeps, veps).
    gx / gv: cotangents of the proposed state; gl [nb]: cotangent of sum logdet (every
    sub-update's logdet enters the sum with weight 1).  on_final(params): called as soon as the sweep
    has finished a set of parameters (GradReducer.ready: the gradient exchange overlaps the rest)."""
    if dyn.group == 'SU3':
        return _backward_su3(dyn, tape, gx, gv, gl, beta, on_final)
    lat = dyn.latvolume
    nb = gx.shape[0]
    gx = gx.reshape(nb, -1).contiguous()
    gv = gv.reshape(nb, -1).contiguous()
    eps_acc: dict = {}
    finals = _first_use(tape) if on_final is not None else {}
    for idx in range(len(tape.entries) - 1, -1, -1):
        e = tape.entries[idx]
        kind = e['kind']
        if kind == 'flip':
            gv = -gv
            continue
        if kind == 'v':
            x, v, F = e['x'], e['v'], e['F']
            dv, dF, ds, dt, dq, deps = ops.v_update_bwd(v, F.reshape(nb, -1), e['s'], e['t'],
                                                        e['q'], e['eps'], e['forward'], gv, gl)
            dx_net, dF_net = e['net'].backward(e['ctx'], ds, dt, dq)
            ops.add_(dF, dF_net.reshape(nb, -1))
            ops.add_(gx, dx_net.reshape(nb, -1))
            ops.u1_force_bwd_(gx, x, dF, beta, lat)
            gv = dv
            eps_acc.setdefault(('v', e['step']), []).append(deps)
        else:
            x, v = e['x'], e['v']
            dx, ds, dt, dq, deps = ops.u1_x_update_bwd(
                x.reshape(nb, -1), v, e['s'], e['t'], e['q'], e['mask'], e['complement'],
                e['eps'], e['forward'], dyn.config.use_ncp, gx, gl, gv)
            dxm, dv_net = e['net'].backward(e['ctx'], ds, dt, dq)
            ops.add_(gv, dv_net.reshape(nb, -1))
            ops.u1_masked_cos_sin_bwd_(dx, x, e['mask'], e['complement'], dxm.contiguous())
            gx = dx
            eps_acc.setdefault(('x', e['step']), []).append(deps)
        for net in finals.get(idx, ()):
            on_final(list(net.parameters()))
    _accumulate_eps_grads(dyn, eps_acc)


def _backward_su3(dyn, tape: Tape, gx: Tensor, gv: Tensor, gl: Tensor, beta: float, on_final=None) -> None:
    lat = dyn.latvolume
    nb, _, _, V = gx.shape
    gx, gv = gx.contiguous(), gv.contiguous()
    eps_acc: dict = {}
    pend: dict = {}            # primary tape index -> cotangents deferred by the sharing v-update
    paired: dict = {}          # primary tape index -> its cotangents from the pair kernel
    flips_done: set = set()
    fuse_pairs = getattr(dyn, 'fuse_v_pairs_bwd', True)
    for idx in range(len(tape.entries) - 1, -1, -1):
        e = tape.entries[idx]
        kind = e['kind']
        if kind == 'flip':
            if idx not in flips_done:
                gv = ops.scale(gv, -1.0)
            continue
        if kind == 'v' and fuse_pairs and e.get('primary') is not None:
            # the sharing update and its primary are neighbours on the tape (at most the momentum flip between
            # them): both reversed in one pass, F / s / t / q read once, their cotangents written once
            pidx = e['primary']
            between = tape.entries[pidx + 1:idx]
            if all(b['kind'] == 'flip' for b in between):
                pe = tape.entries[pidx]
                flip = len(between) % 2 == 1
                dv, dF, dsn, dtn, dqn, deps1, deps2 = ops.v_update_bwd_pair_c128(
                    pe['v'].reshape(nb, -1), e['v'].reshape(nb, -1), e['F'].reshape(nb, -1), e['s'], e['t'],
                    e['q'], pe['eps'], pe['forward'], e['eps'], e['forward'], flip, gv.reshape(nb, -1), gl)
                eps_acc.setdefault(('v', e['step']), []).append(deps2)
                eps_acc.setdefault(('v', pe['step']), []).append(deps1)
                paired[pidx] = (dv, dF, dsn, dtn, dqn)
                flips_done.update(range(pidx + 1, idx))
                continue
        if kind == 'v' and idx in paired:
            x, v, F = e['x'], e['v'], e['F']
            dv, dF, dsn, dtn, dqn = paired.pop(idx)
            dF = dF.reshape(F.shape)
            if e['ctx'].get('native'):
                dxr, dfr = e['net'].backward(e['ctx'], dsn, dtn, dqn)
                ops.su3_projsu_vec8_bwd_(dF, F, dfr.reshape(nb, -1))
                ops.su3_projsu_vec8_bwd_(gx, x, dxr.reshape(nb, -1))
            else:
                ds, dt, dq = (ops.unpack_entries(a, V, 9) for a in (dsn, dtn, dqn))
                dxr, dfr = e['net'].backward(e['ctx'], ds, dt, dq)
                ops.su3_projsu_vec8_bwd_(dF, F, ops.pack_entries(dfr.reshape(nb, -1), V, 8))
                ops.su3_projsu_vec8_bwd_(gx, x, ops.pack_entries(dxr.reshape(nb, -1), V, 8))
            ops.su3_force_bwd_(gx, x, dF, beta, lat)
            gv = dv.reshape(v.shape)
            continue
        if kind == 'v':
            x, v, F = e['x'], e['v'], e['F']
            # (a primary entry adds the cotangents its secondary deferred to it in the same pass)
            dv, dF, dsn, dtn, dqn, deps = ops.v_update_bwd_c128(
                v.reshape(nb, -1), F.reshape(nb, -1), e['s'], e['t'], e['q'], e['eps'],
                e['forward'], gv.reshape(nb, -1), gl, acc=pend.pop(idx, None))
            if e.get('primary') is not None:
                # same x, force and (s, t, q) as the primary entry: hand the cotangents over
                assert e['primary'] not in pend
                pend[e['primary']] = (dF, dsn, dtn, dqn)
                gv = dv.reshape(v.shape)
                eps_acc.setdefault(('v', e['step']), []).append(deps)
                continue
            dF = dF.reshape(F.shape)
            if e['ctx'].get('native'):
                dxr, dfr = e['net'].backward(e['ctx'], dsn, dtn, dqn)
                ops.su3_projsu_vec8_bwd_(dF, F, dfr.reshape(nb, -1))
                ops.su3_projsu_vec8_bwd_(gx, x, dxr.reshape(nb, -1))
            else:
                ds, dt, dq = (ops.unpack_entries(a, V, 9) for a in (dsn, dtn, dqn))
                dxr, dfr = e['net'].backward(e['ctx'], ds, dt, dq)
                ops.su3_projsu_vec8_bwd_(dF, F, ops.pack_entries(dfr.reshape(nb, -1), V, 8))
                ops.su3_projsu_vec8_bwd_(gx, x, ops.pack_entries(dxr.reshape(nb, -1), V, 8))
            ops.su3_force_bwd_(gx, x, dF, beta, lat)
            gv = dv.reshape(v.shape)
            eps_acc.setdefault(('v', e['step']), []).append(deps)
        elif kind == 'x2':
            gx, deps = ops.su3_expm_mul2_bwd_n(e['x'], e['v'], e['eps'], e['mask'], e['complement_first'],
                                               gx, gv)
            eps_acc.setdefault(('x', e['step']), []).append(deps if e['forward'] else -deps)
        else:
            gx, deps = ops.su3_expm_mul_bwd_n(e['x'], e['v'], e['eps'], e['mask'], e['complement'],
                                              gx, gv)
            # the kernel differentiates w.r.t. the signed step it was called with
            eps_acc.setdefault(('x', e['step']), []).append(deps if e['forward'] else -deps)
    assert not pend and not paired
    for net in {id(e['net']): e['net'] for e in tape.entries if e.get('kind') == 'v'}.values():
        net.flush_deferred()
        if on_final is not None and not net.native_active():
            # (native-order shadows: the matrices become final one by one in native_train_end)
            on_final(list(net.parameters()))
    _accumulate_eps_grads(dyn, eps_acc)


def _accumulate_eps_grads(dyn, eps_acc: dict) -> None:
    """d loss / d (x|v)eps[step] = sum over chains and sub-updates of deps, times d eps / d p"""
    for (which, st), lst in eps_acc.items():
        p = (dyn.xeps if which == 'x' else dyn.veps)[st]
        if not p.requires_grad or p.grad is None:
            continue
        _, slope = _eps_and_slope(p)
        tot = torch.stack(lst).sum()
        p.grad.add_((slope * tot).to(p.dtype).reshape(p.shape))


def _cat_metrics(parts: list, sizes: list) -> dict:
    """Merge the metrics of chain micro-batches: tensors are concatenated along their chain
    axis (the one whose extent is the micro-batch size), everything else is taken from the
    first part."""
    out = {}
    for k, v0 in parts[0].items():
        if isinstance(v0, Tensor) and v0.ndim >= 1:
            ax = [d for d in range(v0.ndim) if v0.shape[d] == sizes[0]]
            if ax and all(isinstance(p[k], Tensor) for p in parts):
                a = ax[0] if v0.ndim == 1 or v0.shape[0] == sizes[0] else ax[-1]
                out[k] = torch.cat([p[k] for p in parts], dim=a)
                continue
        out[k] = v0
    return out


def train_forward_backward_chunked(dyn, loss_fn, x: Tensor, beta, micro_batch: int,
                                   loss_weight: float = 1.0, reducer: Optional[GradReducer] = None):
    """train_forward_backward over micro-batches of chains.  The loss is a mean over chains and
    the chains are independent, so the gradient is the size-weighted sum of the micro-batch
    gradients; the trajectory tape then holds `micro_batch` chains at a time (what makes the
    16^4 shard of BASELINE cfg-5 trainable within 288 GB).  Exact for networks without
    BatchNorm (whose batch statistics would become per-micro-batch)."""
    nb = x.shape[0]
    outs, mets, sizes, loss = [], [], [], 0.0
    inj = dyn._inject
    direction = None
    if not dyn.config.merge_directions:
        # ONE direction per optimiser step, drawn (or injected) here like the unchunked path and the
        # reference do (dynamics.py:709) and handed to every micro-batch: the step must not mix forward and
        # backward trajectories, nor consume the host generator once per micro-batch
        d = inj.get('forward') if inj else None
        direction = bool(torch.rand(1) > 0.5) if d is None else bool(d)
    nets = _native_begin(dyn, min(nb, micro_batch))     # one gather / scatter for all micro-batches
    try:
        for lo in range(0, nb, micro_batch):
            hi = min(nb, lo + micro_batch)
            if inj is not None or direction is not None:
                sl = {}
                if inj is not None and inj.get('normals') is not None:
                    nrm = inj['normals']
                    sl['normals'] = nrm[:, lo:hi] if dyn.group == 'SU3' else nrm[lo:hi]
                if inj is not None and inj.get('u') is not None:
                    sl['u'] = inj['u'][lo:hi]
                if direction is not None:
                    sl['forward'] = direction
                dyn._inject = sl
            xo, m, l = train_forward_backward(dyn, loss_fn, x[lo:hi], beta,
                                              loss_weight=loss_weight * (hi - lo) / nb)
            m.pop('mc_states', None)
            outs.append(xo)
            mets.append(m)
            sizes.append(hi - lo)
            loss = loss + l * ((hi - lo) / nb)
        done = True
    finally:
        dyn._inject = inj
        # (gradients accumulate over the micro-batches: only the closing scatter can hand slabs to the exchange)
        cb = reducer.ready if (reducer is not None and reducer.active and locals().get('done')) else None
        for n in nets:
            n.native_train_end(on_ready=cb)
    return torch.cat(outs, 0), _cat_metrics(mets, sizes), loss


def _native_begin(dyn, nb: Optional[int] = None) -> list:
    """SU(3) dense vnets train on native-order weight shadows (LeapfrogLayer.native_train_begin;
    `dyn.native_training = False` keeps the reference-order path with activation transposes).
    Returns the networks this call switched (the caller ends them); networks that are already in
    native mode (a chunked step begins once for all its micro-batches) are left alone.
    The shadows cost one gather + one scatter of the big matrices (and twice their bytes of HBM) per
    step; they are skipped only when the device is short of memory (`dyn.native_training = 'force'`
    takes them regardless)."""
    if dyn.group != 'SU3' or not getattr(dyn, 'native_training', True):
        return []
    from l2hmc.network.pytorch.network import ConvStack
    p = dyn._perms()
    nets, seen = [], set()
    for st in range(dyn.config.nleapfrog):
        n = dyn._get_vnet(st)
        if id(n) in seen or isinstance(n.input_layer.conv_stack, ConvStack) or n.native_active():
            continue
        seen.add(id(n))
        have = bool(getattr(n, '_nat', None) and n._nat.get('w'))      # shadows of an earlier step: already paid for
        if nb is not None and not have and getattr(dyn, 'native_training', True) != 'force':
            # The shadows (weights + gradients: 2 x the five matrices) and what runs on them -- the sliced tape
            # heads (+ 7/8 of the head weights), the deferred weight gradients -- are taken whenever three times
            # the matrices' bytes fit in half of what the device has free.  (Rounds 2-3 compared the shadows'
            # gather + scatter with the 18 activation transposes they replace and kept the 16^4 shard in
            # 16-chain micro-batches on the reference-order path: 7.08 s per step at 156.7 GiB; with the shadows
            # 4.22 s at 213.5 GiB, `profiles/r04am_*`.)
            il = n.input_layer
            wbytes = sum(t.numel() * t.element_size() for t in (
                il.xlayer.weight, il.vlayer.weight, n.scale.layer.weight, n.transl.weight,
                n.transf.layer.weight))
            dev = il.vlayer.weight.device
            if dev.type == 'cuda' and not ops.mem_gate('native-order weight shadows', 3 * wbytes, 0.5, dev):
                continue
        n.native_train_begin(p['in'], p['out'])
        nets.append(n)
    return nets


def train_forward_backward(dyn, loss_fn, x: Tensor, beta, loss_weight: float = 1.0,
                           reducer: Optional[GradReducer] = None):
    """forward_step + calc_loss + loss.backward() of the reference's train_step for one input
    batch: returns (x_out [nb, xdim], metrics, loss).  Gradients are ACCUMULATED into the
    parameters' .grad (zero them first).  reducer: this call completes the step's gradients -- their
    data-parallel exchange starts slab by slab while the sweep is still running (GradReducer)."""
    from l2hmc.dynamics.pytorch.dynamics import _beta
    b = _beta(beta)
    merged = bool(dyn.config.merge_directions)
    forward = True
    if not merged:
        # the reference trains on what `forward` samples with: `apply_transition`, one direction per
        # step drawn BEFORE the momenta (dynamics.py:709); tests pin it through dyn._inject['forward']
        inj = dyn._inject.get('forward') if dyn._inject else None
        forward = bool(torch.rand(1) > 0.5) if inj is None else bool(inj)
    xn = dyn._pack(x)
    vn = dyn._momentum_n(xn.shape[0])
    nets = _native_begin(dyn, xn.shape[0])
    cb = None
    try:
        if merged:
            x_, v_, hist, tape = trajectory_fb_train(dyn, xn, vn, b)
        else:
            x_, v_, hist, tape = trajectory_train(dyn, xn, vn, b, forward)
        loss, gx, gv, gl = loss_and_seeds(dyn, loss_fn, xn, x_, v_, tape, hist['sumlogdet'], b)
        if loss_weight != 1.0:
            gx, gv, gl = gx * loss_weight, gv * loss_weight, gl * loss_weight
        cb = reducer.ready if (reducer is not None and reducer.active) else None
        backward(dyn, tape, gx, gv, gl, b, cb)
    except BaseException:
        cb = None
        raise
    finally:
        for n in nets:
            n.native_train_end(on_ready=cb if nets else None)
    xout, metrics = dyn._finish(xn, vn, x_, v_, beta, hist, with_sumlogdet=True)
    return xout, metrics, loss
