"""Thin ``Trainer`` counterpart driving the hot path: ``hmc_step`` / ``eval_step`` /
``eval`` / ``warmup`` with the reference's step sequence (per-step ``compat_proj``, loss and
lattice metrics, ``trainers/pytorch/trainer.py:904-956, 1085-1252, 1699-1744``).

``train_step`` (U(1)): hand-written reverse sweep + fused Adam over a flat parameter arena +
one RCCL all-reduce of the flat gradient (dynamics/pytorch/training.py) in place of
autograd / torch.optim / DDP.  Out of scope here: wandb /
aim / rich live displays / checkpoint directories.
"""
from __future__ import annotations

import time
from typing import Optional

import numpy as np
import torch

import l2hmc.configs as cfgs
from l2hmc import DEVICE
from l2hmc.dynamics.pytorch.dynamics import Dynamics
from l2hmc.lattice.su3.pytorch.lattice import LatticeSU3
from l2hmc.lattice.u1.pytorch.lattice import LatticeU1, plaq_exact
from l2hmc.loss.pytorch.loss import LatticeLoss
from l2hmc.network.pytorch.network import NetworkFactory
from l2hmc.utils.history import BaseHistory
from l2hmc.utils.step_timer import StepTimer

Tensor = torch.Tensor


class Trainer:
    def __init__(self, cfg: cfgs.ExperimentConfig | dict, build_networks: bool = True,
                 keep=None, skip=None):
        if isinstance(cfg, dict):
            cfg = cfgs.instantiate(cfg)
        self.config = cfg
        # stored like the reference does (trainers/trainer.py:65-66; it never reads them again)
        self.keep = [keep] if isinstance(keep, str) else keep
        self.skip = [skip] if isinstance(skip, str) else skip
        if cfg.precision == 'float64' or cfg.dynamics.group.upper() == 'SU3':
            torch.set_default_dtype(torch.float64)
        self.device = DEVICE
        self.lattice = self.build_lattice()
        self.g = self.lattice.g
        self.loss_fn = LatticeLoss(lattice=self.lattice, loss_config=cfg.loss)
        self.dynamics = self.build_dynamics(build_networks)
        self.grad_scaler = None
        if cfg.precision in ('fp16', 'bf16') and build_networks:
            # the reference wraps Dynamics.forward in torch.autocast(dtype=precision)
            # (trainer.py:211-219, 1276-1280): Linear layers in 16 bit, lattice arithmetic in fp32 -- in
            # eval_step AND in train_step, whose loss goes through a GradScaler (:256-257, 1303-1313):
            # the tape's forward runs the 16-bit layers, the reverse sweep updates fp32 master weights,
            # `grad_scaler` keeps GradScaler's scale / skip-on-inf bookkeeping.
            if cfg.dynamics.group.upper() != 'U1':
                raise ValueError(f'precision={cfg.precision}: SU(3) is complex128 by definition')
            self.dynamics.set_net_precision(cfg.precision)
            from l2hmc.dynamics.pytorch.training import LossScaler
            self.grad_scaler = LossScaler()
        evals = 2 * cfg.dynamics.nleapfrog if cfg.dynamics.merge_directions \
            else cfg.dynamics.nleapfrog
        self.timers = {k: StepTimer(evals_per_step=evals) for k in ('train', 'eval', 'hmc')}
        # per-job metric stores with the reference's dataset dump (utils/history.py:157-263, 854-909)
        self.histories = {k: BaseHistory(steps=cfg.steps) for k in ('train', 'eval', 'hmc')}
        self._estep = self._hstep = self._gstep = 0
        self.arena = None                 # flat parameter / gradient arena (built on first train_step)
        # chains per training micro-batch (None: all at once).  Not a reference option: bounds the
        # trajectory tape so that large lattices train within HBM (exact without BatchNorm).
        self.micro_batch: Optional[int] = None
        # gradient exchange: slab by slab while the reverse sweep is still running (False: ONE blocking
        # all-reduce of the flat arena after it); `comm`: a utils.dist.NativeComm for the C-ABI route
        self.overlap_grad_exchange = True
        self.comm = None
        self.last_exchange: Optional[dict] = None

    # -- construction (trainer.py:490-562, trainers/trainer.py:292-309)
    def build_lattice(self):
        d = self.config.dynamics
        if d.group.upper() == 'U1':
            return LatticeU1(d.nchains, list(d.latvolume))
        return LatticeSU3(d.nchains, list(d.latvolume), c1=self.config.c1)

    def get_input_spec(self) -> cfgs.InputSpec:
        d = self.config.dynamics
        xshape = d.xshape
        if d.group.upper() == 'U1':
            dims = {'xnet': {'x': [d.xdim, 2], 'v': [d.xdim]},
                    'vnet': {'x': [d.xdim], 'v': [d.xdim]}}
        else:
            xdim = int(np.cumprod(xshape[1:-2])[-1]) * 8
            dims = {'xnet': {'x': [xdim], 'v': [xdim]}, 'vnet': {'x': [xdim], 'v': [xdim]}}
        return cfgs.InputSpec(xshape=tuple(xshape), **dims)

    def build_dynamics(self, build_networks: bool = True) -> Dynamics:
        nf = None
        if build_networks:
            nf = NetworkFactory(input_spec=self.get_input_spec(),
                                network_config=self.config.network,
                                conv_config=self.config.conv,
                                net_weights=self.config.net_weights)
        return Dynamics(potential_fn=self.lattice.action, config=self.config.dynamics,
                        network_factory=nf)

    # -- small parts of the reference's Trainer surface (trainer.py:299-308, 473-571, 703-707,
    #    1254-1264, 1929-1950)
    def count_parameters(self, model: Optional[torch.nn.Module] = None) -> int:
        model = self.dynamics if model is None else model
        return sum(p.numel() for p in model.parameters() if p.requires_grad)

    def draw_x(self) -> Tensor:
        return self.g.random(list(self.config.dynamics.xshape)).flatten(1)

    def draw_v(self) -> Tensor:
        return self.g.random_momentum(list(self.config.dynamics.xshape))

    def build_loss_fn(self) -> LatticeLoss:
        return LatticeLoss(lattice=self.lattice, loss_config=self.config.loss)

    def calc_loss(self, xinit: Tensor, xprop: Tensor, acc: Tensor) -> Tensor:
        return self.loss_fn(xinit, xprop, acc)

    def get_lr(self, step: int) -> float:
        return self.config.learning_rate.lr_init

    def reset_optimizer(self) -> None:
        """Drop the Adam moments and the step counter (trainer.py:483-488)."""
        if self.arena is not None:
            self.arena.reset_state()

    def should_log(self, epoch: int) -> bool:
        return epoch % self.config.steps.log == 0 and self._rank_zero()

    def should_print(self, epoch: int) -> bool:
        return epoch % self.config.steps.print == 0 and self._rank_zero()

    @staticmethod
    def _rank_zero() -> bool:
        from l2hmc import RANK
        return RANK == 0

    def metric_to_numpy(self, metric) -> np.ndarray:
        if isinstance(metric, float):
            return np.array(metric)
        if isinstance(metric, list):
            if isinstance(metric[0], Tensor):
                metric = torch.stack(metric)
            elif isinstance(metric[0], np.ndarray):
                metric = np.stack(metric)
            else:
                raise ValueError(f'Unexpected value encountered: {type(metric)}')
        if isinstance(metric, Tensor):
            return metric.detach().cpu().numpy()
        return np.asarray(metric)

    def train_epoch(self, x: Tensor, beta, era: Optional[int] = None,
                    nepoch: Optional[int] = None, warmup: bool = True, **kw) -> tuple[Tensor, dict]:
        """One era at fixed beta: optional thermalisation by HMC, then `nepoch` train steps
        (trainer.py:1478-1620 without the logging / plotting side effects)."""
        nepoch = self.config.steps.nepoch if nepoch is None else nepoch
        if warmup:
            x = self.warmup(beta=float(beta), x=x)
        out = self.train(x=x, beta=float(beta), nsteps=nepoch)
        return out['x'], out['history']

    # -- steps
    def _prep(self, x: Tensor) -> Tensor:
        """every step starts with compat_proj (U1: mod 2 pi, SU3: projectSU) trainer.py:915-917"""
        return self.g.compat_proj(self.dynamics.unflatten(x.to(self.device)))

    def _finish(self, xi, xo, metrics, timer_key) -> tuple[Tensor, dict]:
        xp = metrics.pop('mc_states').proposed.x
        loss = self.loss_fn(x_init=xi, x_prop=xp, acc=metrics['acc'])
        if self.config.dynamics.verbose:
            metrics.update(self.loss_fn.lattice_metrics(xinit=xi, xout=xo))
        metrics.update({'loss': loss.item()})
        return xo.detach(), metrics

    def hmc_step(self, inputs, eps: Optional[float] = None,
                 nleapfrog: Optional[int] = None) -> tuple[Tensor, dict]:
        self.dynamics.eval()
        xi, beta = inputs
        beta = torch.as_tensor(beta, dtype=torch.get_default_dtype())
        xi = self._prep(xi)
        xo, metrics = self.dynamics.apply_transition_hmc((xi, beta), eps=eps, nleapfrog=nleapfrog)
        self._hstep += 1
        return self._finish(xi, xo, metrics, 'hmc')

    def eval_step(self, inputs) -> tuple[Tensor, dict]:
        self.dynamics.eval()
        xinit, beta = inputs
        beta = torch.as_tensor(beta, dtype=torch.get_default_dtype())
        xinit = self._prep(xinit)
        xout, metrics = self.dynamics((xinit, beta))
        self._estep += 1
        return self._finish(xinit, xout, metrics, 'eval')

    def train_step(self, inputs) -> tuple[Tensor, dict]:
        """One optimisation step (trainer.py:1316-1367): compat_proj -> trajectory in train mode
        -> LatticeLoss(x_init, x_prop, acc) -> gradients (hand-written reverse sweep,
        dynamics/pytorch/training.py) -> data-parallel all-reduce of the flat gradient ->
        clip_grad_norm -> fused Adam.  U(1) and SU(3)."""
        from l2hmc.dynamics.pytorch import training as T
        # (merge_directions=False trains on the single-direction kernel the reference's forward
        # samples with, swapped accept arguments included: training.trajectory_train)
        gas = int(getattr(self.config, 'gradient_accumulation_steps', 1) or 1)
        if gas != 1:
            raise NotImplementedError(
                f'gradient_accumulation_steps={gas}: not implemented (Trainer.micro_batch bounds '
                'the tape memory instead: same gradient, one optimiser step per train_step)')
        if self.arena is None:
            self.arena = self._new_arena()
        self.dynamics.train()
        xinit, beta = inputs
        beta = torch.as_tensor(beta, dtype=torch.get_default_dtype())
        xinit = self._prep(xinit)
        self.arena.zero_grad()
        mb = self.micro_batch

        # data-parallel exchange overlapped with the reverse sweep (what DDP's buckets do for the reference,
        # trainer.py:246-257): the sweep that COMPLETES the step's gradients hands finished slabs to RCCL
        reducer = T.GradReducer(self.arena, comm=self.comm) if self.overlap_grad_exchange else None

        def fwd_bwd(xin, w=1.0, last=True):
            red = reducer if last else None
            if mb is not None and 0 < mb < xin.shape[0]:
                return T.train_forward_backward_chunked(self.dynamics, self.loss_fn, xin, beta, mb,
                                                        loss_weight=w, reducer=red)
            return T.train_forward_backward(self.dynamics, self.loss_fn, xin, beta, loss_weight=w,
                                            reducer=red)
        # mixed precision: `grad_scaler.scale(loss).backward()` = the seeds of the sweep times the scale
        ls = 1.0 if self.grad_scaler is None else self.grad_scaler.get_scale()
        aw = self.config.loss.aux_weight
        xout, metrics, loss = fwd_bwd(xinit, ls, last=not aw > 0)
        loss_tot = loss
        if aw > 0:
            # the reference's `aux_loss += aw * aux_loss` (trainer.py:1343-1353)
            yinit = self.g.random(list(xinit.shape)).to(self.device)
            _, _m, aux = fwd_bwd(yinit, (1.0 + aw) * ls)
            loss_tot = loss + (1.0 + aw) * aux
        if reducer is not None:
            scale = reducer.finish() / ls
            self.last_exchange = {'collectives_overlapped': reducer.launched, 'ranks': reducer.world}
        else:
            scale = self.arena.all_reduce(comm=self.comm) / ls     # `unscale_`: folded into the fused Adam
        clip = float(self.config.learning_rate.clip_norm)
        found_inf = False
        if clip > 0.0 or self.grad_scaler is not None:
            gn = self.arena.grad_norm(scale)
            found_inf = not np.isfinite(gn)
            if clip > 0.0 and not found_inf:
                coef = clip / (gn + 1e-6)
                if coef < 1.0:
                    scale *= coef
        if self.grad_scaler is not None:
            # `grad_scaler.step(optimizer)` skips the update when a gradient is inf / nan; `update()`
            self.grad_scaler.update(found_inf)
            metrics['loss_scale'] = ls
            #  the loss of a skipped step is still finite: say that no update was made
            metrics['step_skipped'] = bool(found_inf)
        if not (found_inf and self.grad_scaler is not None):
            self.arena.adam_step(lr=float(self.config.learning_rate.lr_init), grad_scale=scale)
        if self._gstep == 0:
            self._check_paths_across_ranks()
        metrics.pop('mc_states', None)
        metrics['loss'] = float(loss_tot)
        if self.config.dynamics.verbose:
            metrics.update(self.loss_fn.lattice_metrics(xinit=xinit, xout=xout))
        self._gstep += 1
        return xout.detach(), metrics

    @staticmethod
    def _check_paths_across_ranks() -> None:
        """The memory gates of l2hmc._ops pick between kernels that agree to rounding only; in a data-parallel
        job every rank must have picked alike or the replicas drift apart bit by bit.  Compared once, after the
        first step; a difference is reported with both tables."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        from l2hmc import _ops as ops
        sigs = [None] * dist.get_world_size()
        dist.all_gather_object(sigs, ops.mem_gate_signature())
        if len(set(sigs)) > 1:
            import logging
            logging.getLogger('l2hmc').warning(
                'ranks took different memory-gated kernel paths: %s -- set l2hmc._ops.MEM_GATE_POLICY[0] to '
                "'always' or 'never' to pin the choice", sigs)

    def _new_arena(self):
        from l2hmc.dynamics.pytorch import training as T
        skip = None
        if self.config.dynamics.group.upper() == 'SU3' and self.dynamics._networks_built:
            skip = list(self.dynamics.xnet.parameters())      # built, never called: no gradients
        return T.ParamArena(self.dynamics, skip=skip)

    # -- checkpoints (trainer.py:573-680): same keys / file names as the reference
    def save_ckpt(self, era: int, epoch: int, outdir, metrics: Optional[dict] = None):
        """`ckpt-{era}-{epoch}-{step}.tar` with era / epoch / gstep / xeps / veps /
        model_state_dict / optimizer_state_dict (torch.optim.Adam layout) and
        `model-{era}-{epoch}-{step}.pth` (the bare state_dict).  Rank 0 only."""
        from pathlib import Path
        from l2hmc.dynamics.pytorch import training as T
        from l2hmc import RANK
        if RANK != 0:
            return None
        outdir = Path(outdir)
        outdir.mkdir(parents=True, exist_ok=True)
        if self.arena is None:
            self.arena = self._new_arena()
        params = list(self.dynamics.parameters())
        ckpt = {'era': era, 'epoch': epoch, 'gstep': self._gstep,
                'xeps': [e.detach().cpu().numpy() for e in self.dynamics.xeps],
                'veps': [e.detach().cpu().numpy() for e in self.dynamics.veps],
                'model_state_dict': {k: v.detach().cpu().clone()
                                     for k, v in self.dynamics.state_dict().items()},
                'optimizer_state_dict': self.arena.state_dict(
                    params, lr=float(self.config.learning_rate.lr_init))}
        if metrics is not None:
            ckpt.update(metrics)
        f = outdir.joinpath(f'ckpt-{era}-{epoch}-{self._gstep}.tar')
        torch.save(ckpt, f)
        torch.save(ckpt['model_state_dict'], outdir.joinpath(f'model-{era}-{epoch}-{self._gstep}.pth'))
        return f

    def load_ckpt(self, path) -> dict:
        """Restore parameters, step sizes, optimiser moments and the global step from a
        checkpoint written by `save_ckpt` (or by the reference's trainer for the same model)."""
        from l2hmc.dynamics.pytorch import training as T
        from l2hmc import _ops as ops
        ckpt = torch.load(path, map_location='cpu', weights_only=False)
        if self.arena is None:
            self.arena = self._new_arena()
        sd = ckpt['model_state_dict']
        with torch.no_grad():                       # copy INTO the arena views (keep the aliasing)
            own = self.dynamics.state_dict()
            for k, v in sd.items():
                if k in own:
                    own[k].copy_(v.to(own[k].device))
        ops.PARAM_GENERATION[0] += 1
        self.arena.load_state_dict(ckpt['optimizer_state_dict'], list(self.dynamics.parameters()))
        self._gstep = int(ckpt.get('gstep', ckpt.get('step', 0)))
        return ckpt

    def train(self, x: Optional[Tensor] = None, beta: Optional[float] = None,
              nsteps: Optional[int] = None, nera: Optional[int] = None,
              nepoch: Optional[int] = None) -> dict:
        """Training loop (trainer.py:1369-1697 without logging / checkpoint side effects).
        With `beta` / `nsteps`: that many steps at fixed beta.  Otherwise `nera` eras of `nepoch`
        steps with beta annealed linearly from beta_init to beta_final over the eras
        (configs.py:840-873)."""
        if beta is None and nsteps is None:
            nera = self.config.steps.nera if nera is None else nera
            nepoch = self.config.steps.nepoch if nepoch is None else nepoch
            betas = self.config.annealing_schedule.setup(nera=nera, nepoch=nepoch)
            out: dict = {'history': {}, 'x': x}
            for era in range(nera):
                res = self.train(x=out['x'], beta=float(betas[str(era)]), nsteps=nepoch)
                for k, v in res['history'].items():
                    out['history'].setdefault(k, []).extend(v)
                out['history'].setdefault('era', []).extend([era] * nepoch)
                out['x'], out['timer'] = res['x'], res['timer']
            return out
        beta = self.config.annealing_schedule.beta_init if beta is None else beta
        nsteps = self.config.steps.nepoch if nsteps is None else nsteps
        x = self.lattice.random() if x is None else x
        timer = self.timers['train']
        history: dict[str, list] = {}
        patience, stuck = 10, 0                     # trainer.py:1532, 1594-1600 of the reference
        for step in range(nsteps):
            timer.start()
            x, metrics = self.train_step((x, beta))
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            dt = timer.stop()
            record = {'step': step, 'dt': dt, 'beta': beta}
            record.update({k: v for k, v in metrics.items() if k not in ('beta',)})
            for k, v in record.items():
                if isinstance(v, Tensor):
                    v = v.detach().float().cpu() if not v.is_complex() else v.detach().cpu()
                history.setdefault(k, []).append(v)
            avgs = self.histories['train'].update(record)
            x, stuck = self._redraw_if_stuck(x, avgs, stuck, patience, step)
        return {'history': history, 'x': x, 'timer': timer}

    # -- single steps with timing / bookkeeping, and the loss-driven beta schedule
    #    (trainer.py:1369-1476, 1840-1927 without the wandb / aim / rich side effects)
    def train_step_detailed(self, x: Optional[Tensor] = None, beta=None, era: int = 0, epoch: int = 0,
                            verbose: bool = True, **_unused) -> tuple[Tensor, dict]:
        """One timed train_step; the record (era / epoch / tstep / dt / beta / loss / dQ* / metrics) goes
        into histories['train'] and comes back with its per-key averages under 'avgs'."""
        x = self.lattice.random() if x is None else x
        beta = self.config.annealing_schedule.beta_init if beta is None else beta
        self.timers['train'].start()
        xout, metrics = self.train_step((x, float(beta)))
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        dt = self.timers['train'].stop()
        record = {'era': era, 'epoch': epoch, 'tstep': self._gstep, 'dt': dt, 'beta': float(beta),
                  'loss': metrics.pop('loss', None), 'dQsin': metrics.pop('dQsin', None),
                  'dQint': metrics.pop('dQint', None), **metrics}
        record['avgs'] = self.histories['train'].update({k: v for k, v in record.items() if v is not None})
        return xout, record

    def eval_step_detailed(self, job_type: str, x: Optional[Tensor] = None, beta: Optional[float] = None,
                           verbose: bool = True) -> tuple[Tensor, dict]:
        if job_type not in ('eval', 'hmc'):
            raise ValueError(f'Job type should be eval or hmc, got: {job_type}')
        x = self.lattice.random() if x is None else x
        beta = self.config.annealing_schedule.beta_init if beta is None else beta
        self.timers[job_type].start()
        xout, metrics = (self.eval_step if job_type == 'eval' else self.hmc_step)((x, beta))
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        dt = self.timers[job_type].stop()
        record = {'dt': dt, 'beta': beta, 'loss': metrics.pop('loss', None),
                  'dQsin': metrics.pop('dQsin', None), 'dQint': metrics.pop('dQint', None), **metrics}
        record['avgs'] = self.histories[job_type].update({k: v for k, v in record.items() if v is not None})
        return xout, record

    def train_dynamic(self, x: Optional[Tensor] = None, nera: Optional[int] = None,
                      nepoch: Optional[int] = None, beta=None, **_unused) -> dict:
        """Eras at a beta that follows the loss instead of the fixed ladder (trainer.py:1840-1927): start from
        the schedule's first value; after each era, with `annealing_schedule.dynamic`, beta moves by a tenth of
        itself -- down when the loss rose on average over the era, up when it fell -- until it reaches
        beta_final.  (Without `dynamic` the reference's loop never changes beta; `nera` bounds it here.)"""
        self.dynamics.train()
        nera = self.config.steps.nera if nera is None else nera
        nepoch = self.config.steps.nepoch if nepoch is None else nepoch
        sched = self.config.annealing_schedule
        betas = sched.setup(nera=nera, nepoch=nepoch)
        beta_final = float(sched.beta_final)
        b = float(betas.get('0', beta_final)) if beta is None else float(beta)
        x = self.lattice.random() if x is None else x
        out: dict = {'history': {}, 'betas': [], 'x': x, 'timer': self.timers['train']}
        era = 0
        while b < beta_final and era < nera:
            res = self.train(x=out['x'], beta=b, nsteps=nepoch)
            out['x'] = res['x']
            for k, v in res['history'].items():
                out['history'].setdefault(k, []).extend(v)
            out['history'].setdefault('era', []).extend([era] * nepoch)
            out['betas'].append(b)
            losses = torch.as_tensor([float(l) for l in res['history']['loss'][1:]])
            if getattr(sched, 'dynamic', False) and losses.numel() > 1:
                b = b - b / 10.0 if float((losses[1:] - losses[:-1]).mean()) > 0 else b + b / 10.0
            era += 1
        return out

    def _redraw_if_stuck(self, x: Tensor, avgs: dict, stuck: int, patience: int, step: int):
        """The reference's stuck-chain rescue (trainers/pytorch/trainer.py:1209-1215 in `eval`,
        :1594-1600 in `train_epoch`): at logging steps, a mean acceptance below 1e-5 counts as
        stuck; once `patience` such observations have accumulated the chains are redrawn from
        `lattice.random()` and the counter restarts."""
        if step % max(1, int(self.config.steps.log)) != 0:
            return x, stuck
        if float(np.real(avgs.get('acc', 1.0))) < 1e-5:
            if stuck < patience:
                return x, stuck + 1
            x = self.lattice.random()
            return x, 0
        return x, stuck

    def warmup(self, beta: float, nsteps: int = 100, tol: float = 1e-5,
               x: Optional[Tensor] = None) -> Tensor:
        """<= nsteps HMC steps; U(1) stops when |<plaq> - I1/I0| < tol (trainer.py:1699-1744)."""
        x = self.lattice.random() if x is None else x
        is_u1 = isinstance(self.lattice, LatticeU1)
        pexact = plaq_exact(torch.tensor(beta)) if is_u1 else None
        for _ in range(nsteps):
            x, metrics = self.hmc_step((x, beta))
            if is_u1:
                plaqs = metrics.get('plaqs', self.lattice.plaqs(self._prep(x)))
                if float((plaqs.mean().cpu() - pexact).abs()) < tol:
                    break
        return x

    def eval(self, beta: Optional[float] = None, x: Optional[Tensor] = None,
             job_type: str = 'eval', nsteps: Optional[int] = None, eps: Optional[float] = None,
             nleapfrog: Optional[int] = None, dynamic_step_size: Optional[bool] = None) -> dict:
        """(trainer.py:1085-1252) returns {'history': {key: [per-step tensors]}, 'x': x}.
        Stuck chains are redrawn (patience 5, :1106, 1209-1215); with `dynamic_step_size` the HMC
        step size follows the acceptance towards 0.66 in 10 % steps (:1216-1224)."""
        assert job_type in ('eval', 'hmc')
        beta = self.config.annealing_schedule.beta_final if beta is None else beta
        nsteps = self.config.steps.test if nsteps is None else nsteps
        x = self.lattice.random() if x is None else x
        timer = self.timers[job_type]
        history: dict[str, list] = {}
        patience, stuck = 5, 0
        if job_type == 'hmc' and dynamic_step_size and eps is None:
            eps = self.config.dynamics.eps_hmc
        for step in range(nsteps):
            timer.start()
            if job_type == 'hmc':
                x, metrics = self.hmc_step((x, beta), eps=eps, nleapfrog=nleapfrog)
            else:
                x, metrics = self.eval_step((x, beta))
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            dt = timer.stop()
            record = {'step': step, 'dt': dt, 'beta': beta}
            record.update({k: v for k, v in metrics.items() if k not in ('beta',)})
            if job_type == 'hmc' and dynamic_step_size and eps is not None:
                record['eps'] = eps
            for k, v in record.items():
                if isinstance(v, Tensor):
                    v = v.detach().float().cpu() if not v.is_complex() else v.detach().cpu()
                history.setdefault(k, []).append(v)
            avgs = self.histories[job_type].update(record)
            x, stuck = self._redraw_if_stuck(x, avgs, stuck, patience, step)
            if job_type == 'hmc' and dynamic_step_size and eps is not None \
                    and step % max(1, int(self.config.steps.log)) == 0:
                acc_avg = float(metrics['acc_mask'].float().mean())
                eps = eps - eps / 10. if acc_avg < 0.66 else eps + eps / 10.
        return {'history': history, 'x': x, 'timer': timer}
