"""``Dynamics`` -- the L2HMC / HMC leapfrog integrator, API of the reference's
``src/l2hmc/dynamics/pytorch/dynamics.py`` (State :45-67, MonteCarloStates :70-74,
Dynamics :113-1535) driven by the gfx950 kernels of ``libl2q.so``.

How it differs from the reference *inside* (results are the same):

* A trajectory runs on device-resident state in the kernels' native layout (SU(3):
  ``xn[nb, 4, 9, V]`` complex planes, see include/l2q.h).  ``x`` is packed once on entry and
  the proposal / output unpacked once on exit; nothing in between touches the host or the
  reference layout.  Masks and network weights are permuted to native order once.
* force = explicit staples (no autograd), action/plaquette/charges = one reduction pass,
  ``expm(eps v) @ x`` with element masks = one kernel, ``su3_to_vec(projectSU(.))`` = one
  kernel, the generalised v-update (+ logdet reduction) = one kernel, the vnet layers =
  MFMA GEMMs with fused bias / activation / ScaledTanh epilogues.
* Plain HMC fuses ``v -= eps/2 F`` into the force kernel (F is never written).

The public sub-update methods (``_update_v_fwd`` ...) accept / return reference-layout
``State`` objects like the reference and are what the parity tests call.
"""
from __future__ import annotations

import logging
import os
from dataclasses import dataclass
from math import pi as PI
from pathlib import Path
from typing import Callable, Optional, Sequence, Tuple, Union

import numpy as np
import torch
from torch import nn

import l2hmc.configs as cfgs
from l2hmc import DEVICE
from l2hmc import _ops as ops
from l2hmc.group.su3.pytorch.group import SU3
from l2hmc.group.u1.pytorch.group import U1Phase
from l2hmc.lattice.su3.pytorch.lattice import LatticeSU3
from l2hmc.lattice.u1.pytorch.lattice import LatticeU1
from l2hmc.network.pytorch.network import ConvStack, NetworkFactory, dummy_network

log = logging.getLogger(__name__)

TWO_PI = 2. * PI
Shape = Union[tuple, list]
Tensor = torch.Tensor
Array = np.ndarray

DynamicsInput = Tuple[Tensor, Tensor]
DynamicsOutput = Tuple[Tensor, dict]


class State:
    """(x, v, beta) container with the reference's interface (dynamics.py:45-67).  Inside a
    transition the fields live in the kernels' native layout; ``x`` / ``v`` may therefore be
    given as zero-argument callables that convert to the reference layout on first access
    (most callers only ever read ``mc_states.proposed.x``)."""

    def __init__(self, x, v, beta, xshape=None):
        self._x, self._v, self.beta = x, v, beta
        if xshape is None:
            xshape = self.x.shape
        self.xshape = xshape
        self.nb = xshape[0]

    @property
    def x(self) -> Tensor:
        if callable(self._x):
            self._x = self._x()
        return self._x

    @x.setter
    def x(self, value) -> None:
        self._x = value

    @property
    def v(self) -> Tensor:
        if callable(self._v):
            self._v = self._v()
        return self._v

    @v.setter
    def v(self, value) -> None:
        self._v = value

    def __iter__(self):
        return iter((self.x, self.v, self.beta))

    def __repr__(self) -> str:
        return f'State(nb={self.nb}, xshape={tuple(self.xshape)}, beta={self.beta})'

    def flatten(self) -> 'State':
        return State(x=self.x.flatten(1), v=self.v.flatten(1), beta=self.beta)

    def to_numpy(self):
        return {'x': self.x.detach().cpu().numpy(), 'v': self.v.detach().cpu().numpy(),
                'beta': torch.as_tensor(self.beta).detach().cpu().numpy()}


@dataclass
class MonteCarloStates:
    init: State             # Input state
    proposed: State         # Proposal state
    out: State              # Output state (after acc/rej)


def sigmoid(x: torch.Tensor) -> torch.Tensor:
    return 1. / (1. + torch.exp(-x))


def to_u1(x: Tensor) -> Tensor:
    """x wrapped into [-pi, pi) as a U(1) link angle (dynamics.py:76-78)"""
    return ((x + PI) % TWO_PI) - PI


def rand_unif(shape: Sequence[int], a: float, b: float, requires_grad: bool) -> Tensor:
    """~U[a, b] (dynamics.py:85-93)"""
    return ((a - b) * torch.rand(tuple(shape)) + b).clone().detach().requires_grad_(requires_grad)


def random_angle(shape: Sequence[int], requires_grad: bool = True) -> Tensor:
    """angles in (-pi, pi) (dynamics.py:96-98)"""
    return rand_unif(shape, -PI, PI, requires_grad=requires_grad)


class Mask:
    """m / (1 - m) pair with ``combine(x, y) = m x + (1 - m) y`` (dynamics.py:102-110)"""

    def __init__(self, m: Tensor):
        self.m = m
        self.mb = torch.ones_like(m) - m

    def combine(self, x: Tensor, y: Tensor) -> Tensor:
        return self.m * x + self.mb * y


def _beta(beta) -> float:
    return float(beta.item()) if isinstance(beta, torch.Tensor) else float(beta)


class Dynamics(nn.Module):
    def __init__(self, potential_fn: Callable, config: cfgs.DynamicsConfig,
                 network_factory: Optional[NetworkFactory] = None):
        super().__init__()
        self.config = config
        self.xdim = self.config.xdim
        self.xshape = self.config.xshape
        self.potential_fn = potential_fn
        self.nlf = self.config.nleapfrog
        self._device = DEVICE
        self.device = DEVICE
        self.group = self.config.group.upper()
        if self.group == 'U1':
            self.g = U1Phase()
            self.lattice = LatticeU1(self.config.nchains, self.config.latvolume)
        else:
            self.g = SU3()
            self.lattice = LatticeSU3(self.config.nchains, self.config.latvolume)
        self.latvolume = tuple(int(i) for i in self.config.latvolume)
        self.volume = int(np.prod(self.latvolume))
        self.network_factory = network_factory
        if network_factory is not None:
            self._networks_built = True
            self.networks = self._build_networks(network_factory)
            self.xnet = self.networks['xnet']
            self.vnet = self.networks['vnet']
            self.register_module('xnet', self.networks['xnet'])
            self.register_module('vnet', self.networks['vnet'])
            if self.group == 'SU3':
                # The SU(3) xnet is built (its tensors are state_dict / checkpoint keys) but never
                # called (dynamics.py:1420-1425): 264 M of the 446 M parameters at
                # 8^4 / units [256], 34 GB at 16^4.  It keeps its freshly initialised values.  While it is a
                # small part of the device's headroom (2.1 GB at 8^4) it stays in HBM, so that the module is
                # on ONE device type -- what `DistributedDataParallel(dynamics, find_unused_parameters=True)`,
                # the reference's wrapper (trainers/pytorch/trainer.py:246-257), requires; the 34 GB of the
                # 16^4 shard go to host memory (`dynamics.xnet.to(DEVICE)` brings them back for a DDP wrap).
                nbytes = sum(p.numel() * p.element_size() for p in self.xnet.parameters())
                dev = DEVICE if isinstance(DEVICE, torch.device) else torch.device(DEVICE)
                if not ops.mem_gate('SU(3) xnet (never called) kept on the device', nbytes, 0.05, dev):
                    self.xnet.to('cpu')
        else:
            self._networks_built = False
            self.xnet = dummy_network
            self.vnet = dummy_network
            self.networks = {'xnet': self.xnet, 'vnet': self.vnet}
        self.masks = self._build_masks()
        self._dtype: torch.dtype = torch.get_default_dtype()
        if self._networks_built:
            self._dtype = self._get_vnet(0).input_layer.xlayer.weight.dtype
        rg = (not self.config.eps_fixed)
        self.xeps = nn.ParameterList([
            nn.parameter.Parameter(torch.tensor(self.config.eps, device=DEVICE),
                                   requires_grad=rg)
            for _ in range(self.config.nleapfrog)])
        self.veps = nn.ParameterList([
            nn.parameter.Parameter(torch.tensor(self.config.eps, device=DEVICE),
                                   requires_grad=rg)
            for _ in range(self.config.nleapfrog)])
        # draw momenta / accept uniforms on this device's generator.  The reference draws on
        # l2hmc.DEVICE (utils.py:173-189, dynamics.py:1083-1085); parity against its CPU path
        # uses rng_device = 'cpu' (same generator stream) or injected draws.
        self.rng_device = DEVICE
        self.fuse_heads = True      # SU3: fused heads + v-update kernel (same results)
        self.fuse_x_updates = True  # SU3: both x half-updates of a LF step in one kernel
        self.reuse_v_inputs = True  # force / vec8 / hidden activation once per distinct x
        self.pair_v_updates = True  # adjacent v-updates on the same x in one heads kernel
        self.fuse_x_vec8 = True     # SU3: the x-update also emits the next vnet input vec8(x')
        self.fuse_u1_steps = True   # U1 (small lattices, dense nets): one kernel per sub-update
        # Improved gauge action (c1 != 0): the reference evaluates H with potential_fn (the
        # Trainer's LatticeSU3(c1), trainers/pytorch/trainer.py:499-504) but takes the leapfrog
        # force from its own c1 = 0 lattice (dynamics.py:134-135, 1493-1499).  Reproduced: the
        # rectangle term enters the potential energy only, the force stays (beta/3) TAH(U A_plaq).
        owner = getattr(potential_fn, '__self__', None)
        # The trajectory evaluates H with the built-in Wilson (+ c1 rectangle) action kernels, not
        # by calling potential_fn: that is only equivalent if potential_fn IS the action of a
        # lattice of this group and shape.  Anything else (a lambda, a partial, a custom
        # potential) would be silently ignored in energy / acc / the training seeds -- refuse it.
        want = LatticeSU3 if self.group == 'SU3' else LatticeU1
        if not (isinstance(owner, want)
                and getattr(potential_fn, '__name__', '') in ('action', 'potential_energy')):
            raise ValueError(
                'Dynamics(potential_fn=...): expected the bound `action` of a '
                f'{want.__name__} (got {potential_fn!r}); the HIP trajectory computes the Wilson / '
                'improved gauge action itself and cannot call an arbitrary potential')
        if [int(i) for i in owner._lattice_shape] != [int(i) for i in self.config.latvolume]:
            raise ValueError(f'potential_fn belongs to a lattice of shape {owner._lattice_shape}, '
                             f'config.latvolume is {list(self.config.latvolume)}')
        self.potential_c1 = float(getattr(owner, 'c1', 0.0) or 0.0) if self.group == 'SU3' else 0.0
        self.net_precision = None   # torch.float16 / torch.bfloat16: see set_net_precision
        self.fuse_half_heads = True
        self.merge_hmc_kicks = True  # plain HMC, non-verbose: adjacent half-kicks share one force pass
        self._inject: Optional[dict] = None
        self._eps_cache: dict = {}
        self._masks_native: Optional[list] = None
        self._perm: dict = {}
        self._xcache = None                    # (returned x_out, its version, native original)
        # opt-in (a sampler loop / bench.py that feeds the returned tensor straight back): the entry
        # is validated by tensor identity + torch's version counter only, which writes through raw
        # pointers (the library's own in-place kernels, .data edits) do not bump
        self.cache_native_output = False
        self.pair_v_updates_verbose = True     # verbose=True also pairs adjacent v-updates (mid-point kernel)
        # fp64 heads with K = 256 outside the training tape: products rebuilt from exact int8 slice
        # products on the int8 matrix cores (csrc/heads_sliced.hip; same values to fp64 rounding)
        self.sliced_heads = True
        self.sliced_train_heads = True      # training tape: heads + first v-update on the TAPE instances of that kernel
        # training tape: the vnet input layer on digit images of the step's weights.  Off: at cfg-4 the nine
        # calls save 9 x (0.62 - 0.45) ms and the two image builds + their stream synchronisations cost as much
        # (102.9-104.2 vs 102.5 ms per step, alternating on one box); pays only with more calls per step
        self.sliced_train_input = False
        self.defer_weight_grads = True      # training tape: W.grad of all network calls of a step in one GEMM per matrix
        self.fuse_x_halves_train = True     # training tape (SU3): both x half-updates of a step in one kernel each way
        self.fuse_v_pairs_bwd = True        # reverse sweep (SU3): the two v-updates that share a network call in one pass
        # fp64 input layer of the SU(3) vnet on the int8 matrix cores (csrc/gemm_sliced.hip): its inputs
        # su3_to_vec(projectSU(.)) are bounded by 2.31 entry-wise, which the kernel checks (NaN otherwise)
        self.sliced_input = True
        # train mode + grad mode: forward() records the trajectory and returns tensors with a grad_fn
        # (the reference's forward_step / loss.backward() contract); False keeps the graph-free sampler
        self.autograd_forward = True
        # eval mode, small U(1) lattices: whole transitions replayed from a HIP graph (_auto_graphed)
        self.auto_graph = True
        self.auto_graph_su3 = False        # SU(3): kernel-bound, a replay is worth ~1 % (opt-in)
        # a (shape, beta, step size) is captured the `auto_graph_after`-th time it is seen: a capture costs two
        # warm-up trajectories + the captured one + a device sync, which a sampler that sweeps beta or the HMC step
        # size would pay on every call; the most recently used `AUTO_GRAPH_KEEP` graphs are kept
        self.auto_graph_after = 3
        self._graphs: dict = {}
        self._graph_seen: dict = {}

    def train(self, mode: bool = True):
        """nn.Module.train; leaving train mode also releases the training-only device buffers of the
        networks (native-order shadows, deferred-gradient arenas, tape slice images: ADVICE r04)."""
        was = self.training
        super().train(mode)
        if was and not mode and self._networks_built:
            from l2hmc.network.pytorch.network import LeapfrogLayer
            for m in self.networks.modules():
                if isinstance(m, LeapfrogLayer):
                    m.native_train_release()
        return self

    # ------------------------------------------------------------------ construction
    def set_net_precision(self, precision) -> None:
        """'fp16' / 'bf16' (or the torch dtype): every LeapfrogLayer's Linear layers run in
        half precision with fp32 accumulation, the lattice arithmetic (action, force, cos / sin,
        updates, logdet, accept step) stays fp32 -- BASELINE cfg-3 "fp16 nets / fp32 action",
        the reference's torch.autocast region (trainers/pytorch/trainer.py:211-219, 1276-1280).
        None / 'float32' restores full precision."""
        table = {None: None, 'float32': None, 'fp32': None, 'fp16': torch.float16,
                 'bf16': torch.bfloat16, torch.float16: torch.float16,
                 torch.bfloat16: torch.bfloat16, torch.float32: None}
        if precision not in table:
            raise ValueError(f'set_net_precision: {precision!r}')
        half = table[precision]
        if half is not None and self.group != 'U1':
            raise ValueError('half-precision networks: U(1) only (SU(3) is complex128 by '
                             'definition, group/su3/pytorch/group.py:41)')
        from l2hmc.network.pytorch.network import LeapfrogLayer
        for m in self.networks.modules():
            if isinstance(m, LeapfrogLayer):
                m.set_precision(half)
        self.net_precision = half

    def get_models(self) -> dict:
        if self.config.use_separate_networks:
            xnet, vnet = {}, {}
            for lf in range(self.config.nleapfrog):
                vnet[str(lf)] = self._get_vnet(lf)
                if self.config.use_split_xnets:
                    xnet[str(lf)] = {'0': self._get_xnet(lf, first=True),
                                     '1': self._get_xnet(lf, first=False)}
                else:
                    xnet[str(lf)] = self._get_xnet(lf, first=True)
        else:
            vnet = self._get_vnet(0)
            if self.config.use_split_xnets:
                xnet = {'0': self._get_xnet(0, first=True), '1': self._get_xnet(0, first=False)}
            else:
                xnet = self._get_xnet(0, first=True)
        return {'xnet': xnet, 'vnet': vnet}

    def _build_networks(self, network_factory: NetworkFactory) -> nn.ModuleDict:
        split = self.config.use_split_xnets
        n = self.nlf if self.config.use_separate_networks else 1
        return network_factory.build_networks(n, split, group=self.g)

    def _build_masks(self):
        """nlf binary masks [1, xdim] float32 with xdim // 2 ones (dynamics.py:1101-1110)."""
        masks = []
        for _ in range(self.config.nleapfrog):
            _idx = np.arange(self.xdim)
            idx = np.random.permutation(_idx)[:self.xdim // 2]
            mask = np.zeros((self.xdim,), dtype=np.float32)
            mask[idx] = 1.
            masks.append(torch.from_numpy(mask[None, :]))
        return masks

    def set_masks(self, masks: Sequence) -> None:
        self.masks = [torch.as_tensor(np.asarray(m), dtype=torch.float32).reshape(1, -1)
                      for m in masks]
        self._masks_native = None

    def init_weights(self, method: str = 'xavier_uniform', **kw) -> None:
        """(dynamics.py:372-535) re-initialise every Linear of the networks."""
        fn = getattr(nn.init, method + '_' if not method.endswith('_') else method, None)
        if method in ('zero', 'zeros'):
            fn = nn.init.zeros_
        if fn is None:
            raise ValueError(f'unknown init method {method}')
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, nn.Linear):
                    fn(m.weight)

    # ------------------------------------------------------------------ save / load
    def save(self, outdir: os.PathLike) -> None:
        netdir = Path(outdir).joinpath('networks')
        netdir.mkdir(exist_ok=True, parents=True)
        self.save_eps(outdir=netdir)
        torch.save(self.state_dict(), netdir.joinpath('dynamics.pt').as_posix())

    def save_eps(self, outdir: os.PathLike) -> None:
        netdir = Path(outdir).joinpath('networks')
        netdir.mkdir(exist_ok=True, parents=True)
        xeps = np.array([i.detach().cpu().numpy() for i in self.xeps])
        veps = np.array([i.detach().cpu().numpy() for i in self.veps])
        np.save(netdir.joinpath('xeps.npy'), xeps)
        np.save(netdir.joinpath('veps.npy'), veps)
        np.savetxt(netdir.joinpath('xeps.txt').as_posix(), xeps)
        np.savetxt(netdir.joinpath('veps.txt').as_posix(), veps)

    def load(self, outdir: os.PathLike) -> None:
        netdir = Path(outdir).joinpath('networks')
        self.load_state_dict(torch.load(netdir.joinpath('dynamics.pt')))
        self.assign_eps(self.load_eps(outdir))

    def load_eps(self, outdir: os.PathLike) -> dict[str, dict[str, Tensor]]:
        netdir = Path(outdir).joinpath('networks')
        xe = torch.from_numpy(np.load(netdir.joinpath('xeps.npy')))
        ve = torch.from_numpy(np.load(netdir.joinpath('veps.npy')))
        return {'xeps': {str(lf): xe[lf] for lf in range(self.config.nleapfrog)},
                'veps': {str(lf): ve[lf] for lf in range(self.config.nleapfrog)}}

    def restore_eps(self, outdir: os.PathLike) -> None:
        self.assign_eps(self.load_eps(Path(outdir).joinpath('networks')))

    def assign_eps(self, eps) -> None:
        if isinstance(eps, dict):
            xe, ve = eps['xeps'], eps['veps']
        elif isinstance(eps, tuple):
            xe = {str(n): eps[0] for n in range(self.config.nleapfrog)}
            ve = {str(n): eps[1] for n in range(self.config.nleapfrog)}
        elif isinstance(eps, float):
            xe = {str(n): eps for n in range(self.config.nleapfrog)}
            ve = {str(n): eps for n in range(self.config.nleapfrog)}
        else:
            raise TypeError
        rg = (not self.config.eps_fixed)
        self.xeps = nn.ParameterList()
        self.veps = nn.ParameterList()
        for lf in range(self.config.nleapfrog):
            self.xeps.append(nn.parameter.Parameter(
                torch.as_tensor(xe[str(lf)]).clone().to(DEVICE), requires_grad=rg))
            self.veps.append(nn.parameter.Parameter(
                torch.as_tensor(ve[str(lf)]).clone().to(DEVICE), requires_grad=rg))
        self._eps_cache = {}

    # ------------------------------------------------------------------ helpers
    def flatten(self, x: Tensor) -> Tensor:
        return x.reshape(x.shape[0], -1)

    def unflatten(self, x: Tensor) -> Tensor:
        return x.reshape(x.shape[0], *self.xshape[1:])

    def _eps(self, which: str, step: int) -> float:
        """sigmoid(log(p)) of the step-size parameter, evaluated in the parameter's dtype
        (dynamics.py:82-83, 1270, 1394); cached per parameter version (no per-step sync)."""
        p = (self.xeps if which == 'x' else self.veps)[step]
        key = (which, step)
        ver = (p._version, ops.PARAM_GENERATION[0])
        hit = self._eps_cache.get(key)
        if hit is not None and hit[0] == ver and hit[2] is p:
            return hit[1]
        # refresh every step size with one device -> host copy
        ps = list(self.xeps) + list(self.veps)
        vals = torch.stack([q.detach().reshape(()) for q in ps]).cpu()
        n = len(self.xeps)
        for i, q in enumerate(ps):
            k = ('x', i) if i < n else ('v', i - n)
            self._eps_cache[k] = ((q._version, ops.PARAM_GENERATION[0]),
                                  float(sigmoid(vals[i].log())), q)
        return self._eps_cache[key][1]

    def _get_vnet(self, step: int):
        if not self._networks_built:
            return self.vnet
        if self.config.use_separate_networks:
            return self.vnet.get_submodule(str(step))
        return self.vnet

    def _get_xnet(self, step: int, first: bool = False):
        if not self._networks_built:
            return self.xnet
        if self.config.use_separate_networks:
            xnet = self.xnet.get_submodule(str(step))
            if self.config.use_split_xnets:
                return xnet.get_submodule('first') if first else xnet.get_submodule('second')
            return xnet
        return self.xnet

    def _get_mask(self, step: int) -> tuple[Tensor, Tensor]:
        m = self.masks[step]
        return m, torch.ones_like(m) - m

    def _native_masks(self) -> list:
        """float32 masks on the device in kernel order (SU3: native entry order)."""
        if self._masks_native is None:
            out = []
            for m in self.masks:
                m = m.to(DEVICE).reshape(1, -1).contiguous()
                if self.group == 'SU3':
                    m = ops.pack_entries(m, self.volume)
                out.append(m.reshape(-1).contiguous())
            self._masks_native = out
        return self._masks_native

    def _perms(self):
        if not self._perm:
            self._perm = {'in': ops.native_index(self.volume, 8, DEVICE),
                          'out': ops.native_index(self.volume, 9, DEVICE)}
        return self._perm

    # ---- layout conversion (SU3: reference <-> native planes; U1: identity reshape)
    def _pack(self, a: Tensor) -> Tensor:
        a = a.to(DEVICE)
        if self.group == 'SU3':
            return ops.su3_pack(a.reshape(a.shape[0], -1))
        return a.reshape(a.shape[0], 2, *self.latvolume).contiguous().clone()

    def _pack_input(self, x: Tensor) -> Tensor:
        """_pack for the input of a transition.  A sampler loop feeds the tensor the previous
        transition returned straight back (`x, _ = dynamics((x, beta))`): its native-layout
        original is still at hand, so the 0.24 ms reference -> native transpose is skipped when
        the caller passes that very tensor, unmodified (the transitions never write their input)."""
        c, self._xcache = self._xcache, None         # one use: a miss drops the entry
        if (c is not None and self.group == 'SU3' and x is c[0] and x._version == c[1]
                and self.cache_native_output):
            return c[2]
        return self._pack(x)

    def _unpack(self, an: Tensor) -> Tensor:
        if self.group == 'SU3':
            return ops.su3_unpack(an, self.latvolume)
        return an.reshape(an.shape[0], 2, *self.latvolume)

    # ------------------------------------------------------------------ native physics
    def _force_n(self, xn: Tensor, beta) -> Tensor:
        if self.group == 'SU3':
            return ops.su3_force_n(xn, _beta(beta), self.latvolume)
        return ops.u1_force(xn, _beta(beta), self.latvolume)

    def _kick_n(self, xn: Tensor, vn: Tensor, beta, coef: float,
                v_src: Optional[Tensor] = None) -> None:
        """vn += coef * F(xn), F never materialised (v_src: vn = v_src + coef * F(xn))."""
        if self.group == 'SU3':
            ops.su3_force_kick_n(xn, _beta(beta), coef, vn, self.latvolume, v_src)
        else:
            if v_src is not None:
                vn.copy_(v_src)
            ops.u1_force_kick_(xn, _beta(beta), coef, vn, self.latvolume)

    def _potential_n(self, xn: Tensor, beta) -> Tensor:
        if self.group == 'SU3':
            b, c1 = _beta(beta), self.potential_c1
            pe = (-b * (1.0 - 8.0 * c1) / 3.0) * ops.su3_plaq_sums_n(xn, self.latvolume)[:, 0]
            if c1 != 0.0:
                pe = pe + (-b * c1 / 3.0) * ops.su3_rect_sums_n(xn, self.latvolume)
            return pe
        s = ops.u1_plaq_sums(xn, self.latvolume)
        return _beta(beta) * (self.volume - s[:, 0])

    def _kinetic_n(self, vn: Tensor) -> Tensor:
        return ops.su3_kinetic_n(vn) if self.group == 'SU3' else ops.u1_kinetic(vn)

    def _hamiltonian_n(self, xn: Tensor, vn: Tensor, beta) -> Tensor:
        return self._kinetic_n(vn) + self._potential_n(xn, beta)

    def _momentum_n(self, nb: int) -> Tensor:
        """Fresh momenta (dynamics.py:844): SU3 8 x randn([nb,4,T,X,Y,Z]) in the reference's
        order -> l2q_su3_assemble_tah; U1 randn(nb,2,T,X) flattened."""
        inj = self._inject.get('normals') if self._inject else None
        if self.group == 'SU3':
            shape = (nb, 4, *self.latvolume)
            if inj is not None:
                nrm = torch.as_tensor(inj, dtype=torch.float64).to(DEVICE)
            elif self.rng_device == 'cpu':
                nrm = torch.stack([torch.randn(shape, dtype=torch.float64)
                                   for _ in range(8)]).to(DEVICE)
            else:
                nrm = torch.randn((8, *shape), dtype=torch.float64, device=DEVICE)
            return ops.su3_assemble_tah_n(nrm.reshape(8, nb, 4, self.volume))
        shape = (nb, 2, *self.latvolume)
        if inj is not None:
            v = torch.as_tensor(inj).to(DEVICE)
        else:
            v = torch.randn(shape, device=self.rng_device).to(DEVICE)
        return v.reshape(nb, -1).to(self._real_dtype()).contiguous()

    def _real_dtype(self) -> torch.dtype:
        return torch.float64 if self.group == 'SU3' else self._dtype

    def _uniform(self, acc: Tensor) -> Tensor:
        inj = self._inject.get('u') if self._inject else None
        if inj is not None:
            return torch.as_tensor(inj).to(device=DEVICE, dtype=acc.dtype)
        if self.rng_device == 'cpu':
            return torch.rand(acc.shape, dtype=acc.dtype).to(DEVICE)
        return torch.rand_like(acc)

    # ---- networks on native state
    def _vnet_n(self, step: int, xn: Tensor, fn: Tensor):
        vnet = self._get_vnet(step)
        nb = xn.shape[0]
        if not self._networks_built:
            z = torch.zeros((nb, self.xdim), dtype=self._real_dtype(), device=DEVICE)
            return z, z, z
        if self.group == 'SU3':
            p = self._perms()
            w = vnet.kernel_weights(p['in'], p['out'])
            xv = ops.su3_projsu_vec8_n(xn).reshape(nb, -1)
            fv = ops.su3_projsu_vec8_n(fn).reshape(nb, -1)
            return vnet.forward_flat(xv, fv, w)
        x = xn
        if isinstance(vnet.input_layer.conv_stack, ConvStack):
            x = vnet.input_layer.conv_stack(xn)
        return vnet.forward_flat(x.reshape(nb, -1), fn.reshape(nb, -1))

    def _xnet_n(self, step: int, first: bool, xn: Tensor, vn: Tensor, mask: Tensor,
                complement: bool):
        """U1 only (SU3 never calls its xnet, dynamics.py:1420-1425)."""
        xnet = self._get_xnet(step, first)
        nb = xn.shape[0]
        if not self._networks_built:
            z = torch.zeros((nb, self.xdim), dtype=self._real_dtype(), device=DEVICE)
            return z, z, z
        xm = ops.u1_masked_cos_sin(xn, mask, complement, self.latvolume)
        if isinstance(xnet.input_layer.conv_stack, ConvStack):
            xm = xnet.input_layer.conv_stack(xm)
        return xnet.forward_flat(xm.reshape(nb, -1), vn.reshape(nb, -1))

    # ---- sub-updates on native state (in place on xn / vn)
    def _can_fuse_heads(self, vnet) -> bool:
        return (self.fuse_heads and self._networks_built and self.group == 'SU3'
                and vnet.units[-1] % 2 == 0)

    def _v_inputs_n(self, vnet, xn: Tensor, beta, cache: Optional[dict]):
        """(force, hidden activation z, kernel weights) for the fused heads kernel.  `cache`
        (trajectory-local): the force, the vec8 network inputs and z depend only on x (and the
        network), and x does not change between the closing v-update of one leapfrog step and
        the opening v-update of the next (nor across the momentum flip), so they are computed
        once per distinct x -- 9 instead of 16 evaluations in a merged nlf = 4 trajectory.
        Same inputs, same deterministic kernels: bitwise identical results."""
        nb = xn.shape[0]
        hit = cache is not None and cache.get('valid', False)
        fn = cache['F'] if hit else self._force_n(xn, beta)
        if cache is not None and not hit:
            pre = cache.get('xv_pre')          # vec8(x) emitted by the x-update that produced x
            cache.clear()
            cache.update({'valid': True, 'F': fn})
            if pre is not None:
                cache['xv_pre'] = pre
        if not self._can_fuse_heads(vnet):
            return fn, None, None
        p = self._perms()
        w = vnet.kernel_weights(p['in'], p['out'])
        hs = w['heads_scaled']
        # int8 slice image of the head weights (csrc/heads_sliced.hip); lives in the version-keyed
        # weight cache, so it is rebuilt whenever a parameter changes.  Both knobs are consulted on
        # EVERY call (`use_sliced`), the image is built the first time they are both on.
        hs['use_sliced'] = bool(self._sliced_wanted(vnet) and ops.USE_SLICED_HEADS[0])
        if hs['use_sliced'] and 'sliced' not in hs:
            hs['sliced'] = ops.heads_sliced_build(hs)
        zkey = ('z', id(vnet))
        if cache is not None and zkey in cache:
            return fn, cache[zkey], w
        if cache is not None and 'xv' in cache:
            xv, fv = cache['xv'], cache['fv']
        else:
            pre = cache.pop('xv_pre', None) if cache is not None else None
            xv = pre if pre is not None else ops.su3_projsu_vec8_n(xn).reshape(nb, -1)
            fv = ops.su3_projsu_vec8_n(fn).reshape(nb, -1)
            if cache is not None:
                cache['xv'], cache['fv'] = xv, fv
        z = vnet.hidden_flat(xv, fv, w, sliced_exp=ops.SLICED_INPUT_EXP if self.sliced_input else None)
        if cache is not None:
            cache[zkey] = z
        return fn, z, w

    def _sliced_wanted(self, vnet) -> bool:
        """`sliced_heads`: True -> networks whose last hidden activation is BOUNDED (tanh: every row
        of Z has |z| <= 1 and the fixed-point split of a row keeps >= 46 bits of every entry that
        matters); 'force' -> any activation (an unbounded activation can produce a row with one
        dominant entry, whose small entries the per-row fixed point resolves less finely than fp64:
        error relative to K max|z| max|w| instead of sum |z||w| -- `ops.heads_sliced_zflag()` tells
        whether such a row was seen); False -> the fp64 MFMA kernels."""
        if self.sliced_heads == 'force':
            return True
        return bool(self.sliced_heads) and getattr(vnet, 'act', None) == 'tanh'

    def _fused_u1(self, net) -> Optional[dict]:
        """Weights in the layout of the fused U(1) sub-update kernels, or None when the fused
        path does not apply (SU3, fp64, conv stack, wide layers, large lattice)."""
        if not (self.fuse_u1_steps and self.group == 'U1' and self._networks_built
                and self._dtype == torch.float32 and self.xdim <= ops.u1_fused_max_n()
                and getattr(net, 'half_dtype', None) is None):
            return None
        net._check_mode()
        return net.kernel_weights().get('fused_u1')

    def _half_fused(self, net) -> bool:
        """half-precision layers: heads + update in one kernel (s, t, q stay in registers)."""
        return (self.group == 'U1' and self._networks_built and self.fuse_half_heads
                and getattr(net, 'half_dtype', None) is not None)

    def _update_v_n(self, step: int, xn: Tensor, vn: Tensor, beta, forward: bool,
                    cache: Optional[dict] = None, acc: Optional[Tensor] = None,
                    v_src: Optional[Tensor] = None) -> Tensor:
        """v-update in place (dynamics.py:1266-1297); returns logdet [nb] (the fused U(1) kernel
        adds it into `acc` when given).  `v_src`: the momentum is read from there and only
        written to vn (the heads kernel does that itself; the other paths copy first)."""
        eps = self._eps('v', step)
        nb = xn.shape[0]
        vnet = self._get_vnet(step)
        if v_src is not None and not (self.group == 'SU3' and self._can_fuse_heads(vnet)):
            vn.copy_(v_src)
            v_src = None
        fw = self._fused_u1(vnet)
        if fw is not None:            # force + vnet + update in one launch
            return ops.u1_vstep_(xn, vn, _beta(beta), eps, forward, self.latvolume, fw, acc)
        if self._half_fused(vnet):
            fn = self._force_n(xn, beta)
            x = xn
            if isinstance(vnet.input_layer.conv_stack, ConvStack):
                x = vnet.input_layer.conv_stack(xn)
            w = vnet.kernel_weights()
            z = vnet.hidden_flat_h(x.reshape(nb, -1), fn.reshape(nb, -1), w)
            return ops.u1_heads_update_h_(z, w['h']['heads_scaled'], vnet.nw.t, vn.reshape(nb, -1),
                                          fn.reshape(nb, -1), eps, forward)
        fn, z, w = self._v_inputs_n(vnet, xn, beta, cache)
        if z is not None:
            # heads + momentum update in one kernel: s, t, q never reach HBM
            return ops.vnet_heads_vupdate_(z, w['heads_scaled'], (vnet.nw.s, vnet.nw.t, vnet.nw.q),
                                           vn.reshape(nb, -1), fn.reshape(nb, -1), eps, forward,
                                           None if v_src is None else v_src.reshape(nb, -1))
        if v_src is not None:
            vn.copy_(v_src)
        s, t, q = self._vnet_n(step, xn, fn)
        return ops.v_update_(vn.reshape(nb, -1), fn.reshape(nb, -1), s, t, q, eps, forward)

    def _update_v_pair_n(self, step0: int, forward0: bool, flip: bool, step1: int,
                         forward1: bool, xn: Tensor, vn: Tensor, beta,
                         cache: Optional[dict] = None, mid: Optional[dict] = None) -> Tensor:
        """The closing v-update of one leapfrog step and the opening v-update of the next (same
        x, same network => same s, t, q), optionally with the merged trajectory's v -> -v in
        between (dynamics.py:1001), from ONE evaluation of the heads.  Sum of both logdets.
        With `mid` (per-step metrics, verbose=True): the kernel also returns the first update's
        logdet and the kinetic energy of the momentum between the two updates; they are left in
        mid['ld1'] / mid['ke'] and the return value is the SECOND update's logdet alone."""
        nb = xn.shape[0]
        vnet = self._get_vnet(step1)
        fn, z, w = self._v_inputs_n(vnet, xn, beta, cache)
        args = (z, w['heads_scaled'], (vnet.nw.s, vnet.nw.t, vnet.nw.q), vn.reshape(nb, -1),
                fn.reshape(nb, -1), self._eps('v', step0), forward0, flip, self._eps('v', step1),
                forward1)
        if mid is None:
            return ops.vnet_heads_vupdate_pair_(*args)
        ld, ld1, vn2 = ops.vnet_heads_vupdate_pair_mid_(*args)
        mid['ld1'] = ld1
        # group/su3/pytorch/group.py:125-126 as l2q_su3_kinetic_reduce evaluates it
        mid['ke'] = 0.5 * (vn2 - 8.0 * 4.0 * self.volume)
        return ld - ld1

    def _can_pair_mid(self) -> bool:
        """The mid-point pair kernel (LDS-DMA heads kernel) needs whole 16-wide K-slabs."""
        if self.group != 'SU3' or not self._networks_built:
            return False
        return all(int(self._get_vnet(st).units[-1]) % 16 == 0
                   for st in range(self.config.nleapfrog))

    def _update_x_n(self, step: int, xn: Tensor, vn: Tensor, mask: Tensor, complement: bool,
                    forward: bool, first: bool, acc: Optional[Tensor] = None) -> Optional[Tensor]:
        eps = self._eps('x', step)
        if self.group == 'SU3':
            ops.su3_expm_mul_n(xn, vn, eps if forward else -eps, mask, complement, out=xn)
            return None
        nb = xn.shape[0]
        fw = self._fused_u1(self._get_xnet(step, first))
        if fw is not None:            # masked cos/sin + xnet + update in one launch
            return ops.u1_xstep_(xn.reshape(nb, -1), vn, mask, complement, eps, forward,
                                 self.config.use_ncp, fw, acc)
        xnet = self._get_xnet(step, first)
        if self._half_fused(xnet):
            w = xnet.kernel_weights()
            if isinstance(xnet.input_layer.conv_stack, ConvStack):
                xm = ops.u1_masked_cos_sin(xn, mask, complement, self.latvolume)
                xm = xnet.input_layer.conv_stack(xm)
                z = xnet.hidden_flat_h(xm.reshape(nb, -1), vn.reshape(nb, -1), w)
            else:                      # cos / sin of the masked links formed in the GEMM's loader
                z = xnet.hidden_flat_h_u1x(xn.reshape(nb, -1), mask, complement,
                                           vn.reshape(nb, -1), w)
            return ops.u1_heads_update_h_(z, w['h']['heads_scaled'], xnet.nw.t,
                                          xn.reshape(nb, -1), vn.reshape(nb, -1), eps, forward,
                                          mask=mask, complement=complement,
                                          use_ncp=self.config.use_ncp)
        s, t, q = self._xnet_n(step, first, xn, vn, mask, complement)
        return ops.u1_x_update_(xn.reshape(nb, -1), vn, s, t, q, mask, complement, eps,
                                forward, self.config.use_ncp)

    def _flip_v_n(self, vn: Tensor) -> Tensor:
        if self.group == 'SU3':
            return ops.scale(vn, -1.0, out=vn)
        return vn.neg_()

    def _lf_n(self, step: int, xn: Tensor, vn: Tensor, beta, forward: bool,
              cache: Optional[dict] = None, pend: Optional[dict] = None,
              mid: Optional[dict] = None, x_src: Optional[Tensor] = None,
              v_src: Optional[Tensor] = None) -> Tensor:
        """One generalised leapfrog step in place; returns logdet [nb]
        (dynamics.py:1187-1228).  `pend` (optional, trajectory-local) enables deferral of the
        closing v-update: it is then executed together with the next step's opening v-update
        (`_update_v_pair_n`), its logdet being returned by that next call; the caller flushes
        a last pending update with `_flush_pending_n`.  `mid` (with `pend`, per-step metrics):
        the deferred closing update's logdet and the kinetic energy right after it are left in
        mid['ld1'] / mid['ke'], and the returned logdet covers THIS step's sub-updates only.
        `x_src` (first step of a trajectory, SU(3)): the configuration is READ from x_src and the
        x-update writes it into xn, so the trajectory never copies its input; `v_src` likewise for
        the momentum of the step's opening v-update."""
        if forward:
            st, order = step, ((False, True), (True, False))     # (complement, first)
        else:
            st = self.config.nleapfrog - step - 1
            order = ((True, False), (False, True))
        m = self._native_masks()[st]
        if self.group == 'U1' and self.fuse_u1_steps:
            fv = self._fused_u1(self._get_vnet(st))
            fx = {True: self._fused_u1(self._get_xnet(st, True)),
                  False: self._fused_u1(self._get_xnet(st, False))} if fv is not None else None
            if fv is not None and fx[True] is not None and fx[False] is not None:
                # four one-launch sub-updates, the log-det summed inside the kernels
                nb, b = xn.shape[0], _beta(beta)
                ev, ex = self._eps('v', st), self._eps('x', st)
                ld = ops.u1_vstep_(xn, vn, b, ev, forward, self.latvolume, fv)
                for comp, first in order:
                    ops.u1_xstep_(xn.reshape(nb, -1), vn, m, comp, ex, forward,
                                  self.config.use_ncp, fx[first], ld)
                ops.u1_vstep_(xn, vn, b, ev, forward, self.latvolume, fv, ld)
                return ld
        prev = pend.pop('p', None) if pend is not None else None
        xr = xn if x_src is None else x_src              # where this step reads x before its x-update
        if v_src is not None and prev is not None:
            vn.copy_(v_src)
            v_src = None
        if prev is not None:
            st0, f0, flip = prev
            v0, v1 = self._get_vnet(st0), self._get_vnet(st)
            if v0 is v1 and self._can_fuse_heads(v1):
                ld = self._update_v_pair_n(st0, f0, flip, st, forward, xr, vn, beta, cache, mid)
            else:
                ld = self._update_v_n(st0, xr, vn, beta, f0, cache)
                if mid is not None:
                    mid['ld1'], mid['ke'] = ld, self._kinetic_n(vn)
                if flip:
                    self._flip_v_n(vn)
                l1 = self._update_v_n(st, xr, vn, beta, forward, cache)
                ld = l1 if mid is not None else ld + l1
        else:
            ld = self._update_v_n(st, xr, vn, beta, forward, cache, v_src=v_src)
        if self.group == 'SU3' and self.fuse_x_updates:
            # both half-updates share expm(eps v): one kernel, one pass over x
            eps = self._eps('x', st)
            if (cache is not None and self.fuse_x_vec8 and self.reuse_v_inputs
                    and self._can_fuse_heads(self._get_vnet(st))):
                _, xv = ops.su3_expm_mul2_vec8_n(xr, vn, eps if forward else -eps, m, not forward,
                                                 out=xn)
                cache['xv_pre'] = xv.reshape(xn.shape[0], -1)
            else:
                ops.su3_expm_mul2_n(xr, vn, eps if forward else -eps, m, not forward, out=xn)
        else:
            if x_src is not None:
                xn.copy_(x_src)
            for comp, first in order:
                l = self._update_x_n(st, xn, vn, m, comp, forward, first)
                if l is not None:
                    ld = ld + l
        if cache is not None:
            cache['valid'] = False                     # x changed
        if pend is not None:
            pend['p'] = (st, forward, False)           # closing v-update deferred
            return ld
        return ld + self._update_v_n(st, xn, vn, beta, forward, cache)

    def _flush_pending_n(self, pend: Optional[dict], xn, vn, beta, cache) -> Optional[Tensor]:
        prev = pend.pop('p', None) if pend is not None else None
        if prev is None:
            return None
        st0, f0, flip = prev
        ld = self._update_v_n(st0, xn, vn, beta, f0, cache)
        if flip:
            self._flip_v_n(vn)
        return ld

    # ------------------------------------------------------------------ public sub-updates
    def group_to_vec(self, x: Tensor) -> Tensor:
        return self.g.group_to_vec(self.unflatten(x))

    def vec_to_group(self, x: Tensor) -> Tensor:
        x = self.unflatten(x)
        if self.group == 'SU3':
            return self.g.vec_to_group(x)
        return torch.complex(x[..., 0], x[..., 1])

    def grad_potential(self, x: Tensor, beta: Tensor) -> Tensor:
        return self.lattice.grad_action(x, beta)

    def hamiltonian(self, state: State) -> Tensor:
        return self.kinetic_energy(state.v) + self.potential_energy(state.x, state.beta)

    def kinetic_energy(self, v: Tensor) -> Tensor:
        return self._kinetic_n(self._pack(v) if self.group == 'SU3'
                               else v.to(DEVICE).reshape(v.shape[0], -1).contiguous())

    def potential_energy(self, x: Tensor, beta: Tensor):
        return self.potential_fn(x, beta)

    def _call_vnet(self, step: int, inputs: tuple[Tensor, Tensor]):
        x, force = inputs
        s, t, q = self._vnet_n(step, self._pack(x), self._pack(force))
        return self._heads_to_reference(s, t, q)

    def _call_xnet(self, step: int, inputs: tuple[Tensor, Tensor], first: bool = False):
        """inputs = (m * x, v).  U1: network of [cos, sin] of the masked field."""
        x, v = inputs
        if self.group == 'SU3':
            xnet = self._get_xnet(step, first)
            x = torch.stack([x.real, x.imag], 1)
            v = torch.stack([v.real, v.imag], 1)
            return xnet((x, v))
        ones = torch.ones(self.xdim, dtype=torch.float32, device=DEVICE)
        return self._xnet_n(step, first, self._pack(x), v.to(DEVICE).reshape(x.shape[0], -1),
                            ones, False)

    def _heads_to_reference(self, s, t, q):
        if self.group == 'SU3':
            return tuple(ops.unpack_entries(a, self.volume) for a in (s, t, q))
        return s, t, q

    def _update_v(self, step: int, state: State, forward: bool) -> tuple[State, Tensor]:
        xn = self._pack(state.x)
        vn = self._pack(state.v) if self.group == 'SU3' else \
            state.v.to(DEVICE).reshape(state.v.shape[0], -1).contiguous().clone()
        ld = self._update_v_n(step, xn, vn, state.beta, forward)
        v = self._unpack(vn) if self.group == 'SU3' else vn.reshape(state.v.shape)
        return State(state.x, v, state.beta), ld

    def _update_v_fwd(self, step: int, state: State) -> tuple[State, Tensor]:
        """v' = exp(eps s/2) v - eps/2 (F exp(eps q) + t)  (dynamics.py:1266-1280)"""
        return self._update_v(step, state, True)

    def _update_v_bwd(self, step: int, state: State) -> tuple[State, Tensor]:
        """v' = exp(-eps s/2) (v + eps/2 (F exp(eps q) + t))  (dynamics.py:1282-1297)"""
        return self._update_v(step, state, False)

    def _update_x(self, step: int, state: State, m: Tensor, first: bool, forward: bool):
        xn = self._pack(state.x)
        nb = xn.shape[0]
        m = m.to(DEVICE).reshape(1, -1).float().contiguous()
        if self.group == 'SU3':
            vn = self._pack(state.v)
            mask = ops.pack_entries(m, self.volume).reshape(-1)
        else:
            vn = state.v.to(DEVICE).reshape(nb, -1).contiguous()
            mask = m.reshape(-1)
        ld = self._update_x_n(step, xn, vn, mask, False, forward, first)
        if ld is None:
            ld = torch.zeros(nb, dtype=torch.get_default_dtype(), device=DEVICE)
        return State(x=self._unpack(xn), v=state.v, beta=state.beta), ld

    def _update_x_fwd(self, step: int, state: State, m: Tensor, first: bool):
        """SU3: x' = m x + expm(eps v) @ ((1-m) x); U1: NCP update (dynamics.py:1386-1428)"""
        return self._update_x(step, state, m, first, True)

    def _update_x_bwd(self, step: int, state: State, m: Tensor, first: bool):
        """(dynamics.py:1430-1477)"""
        return self._update_x(step, state, m, first, False)

    def _lf(self, step: int, state: State, forward: bool) -> tuple[State, Tensor]:
        xn = self._pack(state.x)
        nb = xn.shape[0]
        vn = self._pack(state.v) if self.group == 'SU3' else \
            state.v.to(DEVICE).reshape(nb, -1).contiguous().clone()
        ld = self._lf_n(step, xn, vn, state.beta, forward)
        v = self._unpack(vn) if self.group == 'SU3' else vn
        return State(self._unpack(xn), v, state.beta), ld

    def _forward_lf(self, step: int, state: State) -> tuple[State, Tensor]:
        return self._lf(step, state, True)

    def _backward_lf(self, step: int, state: State) -> tuple[State, Tensor]:
        return self._lf(step, state, False)

    def leapfrog_hmc(self, state: State, eps: Optional[float] = None) -> State:
        """v1 = v - eps/2 F(x); x' = update_gauge(x, eps v1); v2 = v1 - eps/2 F(x')
        (dynamics.py:900-913)"""
        eps = self.config.eps if eps is None else eps
        xn = self._pack(state.x)
        nb = xn.shape[0]
        vn = self._pack(state.v) if self.group == 'SU3' else \
            state.v.to(DEVICE).reshape(nb, -1).contiguous().clone()
        self._leapfrog_hmc_n(xn, vn, state.beta, eps)
        v = self._unpack(vn) if self.group == 'SU3' else vn
        return State(x=self._unpack(xn), v=v, beta=state.beta)

    def _leapfrog_hmc_n(self, xn: Tensor, vn: Tensor, beta, eps: float) -> None:
        self._kick_n(xn, vn, beta, -0.5 * eps)
        if self.group == 'SU3':
            ops.su3_expm_mul_n(xn, vn, eps, out=xn)
        else:
            ops.axpy_(xn.reshape(xn.shape[0], -1), vn, eps)
        self._kick_n(xn, vn, beta, -0.5 * eps)

    # ------------------------------------------------------------------ transition kernels
    def _metrics_n(self, xn, vn, beta, logdet, step=None, extras=None) -> dict:
        energy = self._hamiltonian_n(xn, vn, beta)
        m = {'energy': energy, 'logprob': energy - logdet, 'logdet': logdet}
        if extras is not None:
            m.update(extras)
        if step is not None:
            m.update({'xeps': self.xeps[step], 'veps': self.veps[step]})
        return m

    @staticmethod
    def update_history(metrics: dict, history: dict):
        for key, val in metrics.items():
            history.setdefault(key, []).append(val)
        return history

    @staticmethod
    def _stack_history(history: dict) -> dict:
        for key, val in history.items():
            if isinstance(val, list) and isinstance(val[0], Tensor):
                history[key] = torch.stack([v.detach() for v in val])
        return history

    def _zeros_nb(self, nb: int) -> Tensor:
        return torch.zeros(nb, dtype=torch.get_default_dtype(), device=DEVICE)

    def _accept_prob_n(self, h_init: Tensor, h_prop: Tensor, sumlogdet: Tensor) -> Tensor:
        """exp(min(0, H_init - H_prop + sum logdet))  (dynamics.py:1065-1079) -> l2q_accept"""
        acc, _ = ops.accept(h_init, h_prop, sumlogdet, torch.zeros_like(h_init))
        return acc

    def _kernel_hmc_n(self, xn, vn, beta, eps=None, nleapfrog=None):
        """(dynamics.py:915-954) in place on clones; returns (x', v', history)."""
        nb = xn.shape[0]
        v_ = None                                   # cloned below unless the first kick can write a fresh tensor
        x_ = None                                   # cloned below unless the first x-update can write a fresh tensor
        sumlogdet = self._zeros_nb(nb)
        history: dict = {}
        h_init = self._hamiltonian_n(xn, vn, beta)
        if self.config.verbose:
            self.update_history({'energy': h_init, 'logprob': h_init - sumlogdet,
                                 'logdet': sumlogdet}, history)
        eps = self.config.eps_hmc if eps is None else eps
        nlf = (self.config.nleapfrog if not self.config.merge_directions
               else 2 * self.config.nleapfrog)
        if eps is None:
            eps = 1. / nlf
        nleapfrog = nlf if nleapfrog is None else nleapfrog
        h = h_init
        if self.merge_hmc_kicks and not self.config.verbose and nleapfrog > 1:
            # The closing half-kick of step k and the opening one of step k+1 use the same x:
            # v - (eps/2) F - (eps/2) F == v - eps F up to one rounding of v, so one force pass
            # serves both.  nleapfrog + 1 force evaluations instead of 2 nleapfrog.  Only
            # without per-step metrics (they need v at the step boundary).
            if self.group == 'SU3':
                # the opening kick only reads x and v, the first x-update writes a fresh tensor:
                # neither the input configuration nor the momentum is copied
                v_ = torch.empty_like(vn)
                self._kick_n(xn, v_, beta, -0.5 * eps, v_src=vn)
                x_ = ops.su3_expm_mul_n(xn, v_, eps)
            else:
                x_, v_ = xn.clone(), vn.clone()
                self._kick_n(x_, v_, beta, -0.5 * eps)
                ops.axpy_(x_.reshape(nb, -1), v_, eps)
            for i in range(nleapfrog):
                if i > 0:
                    if self.group == 'SU3':
                        ops.su3_expm_mul_n(x_, v_, eps, out=x_)
                    else:
                        ops.axpy_(x_.reshape(nb, -1), v_, eps)
                self._kick_n(x_, v_, beta, -eps if i + 1 < nleapfrog else -0.5 * eps)
            nleapfrog_done = True
        else:
            nleapfrog_done = False
            x_, v_ = xn.clone(), vn.clone()
        for _ in range(0 if nleapfrog_done else nleapfrog):
            self._leapfrog_hmc_n(x_, v_, beta, eps)
            if self.config.verbose:
                h = self._hamiltonian_n(x_, v_, beta)
                self.update_history({'energy': h, 'logprob': h - sumlogdet,
                                     'logdet': sumlogdet}, history)
        if not self.config.verbose or nleapfrog == 0:
            h = self._hamiltonian_n(x_, v_, beta)
        acc = self._accept_prob_n(h_init, h, sumlogdet)
        history.update({'acc': acc, 'sumlogdet': sumlogdet})
        if self.config.verbose:
            history = self._stack_history(history)
        return x_, v_, history

    def _kernel_fb_n(self, xn, vn, beta):
        """Merged forward + backward trajectory (dynamics.py:956-1029)."""
        nb = xn.shape[0]
        # SU(3): the first leapfrog step reads the input configuration and momentum and writes
        # x_ / v_ (x_src, v_src below): no copy of either
        lazy_x = self.group == 'SU3' and self.config.nleapfrog > 0 and self._networks_built
        x_ = torch.empty_like(xn) if lazy_x else xn.clone()
        v_ = torch.empty_like(vn) if lazy_x else vn.clone()
        sumlogdet = self._zeros_nb(nb)
        sldf = torch.zeros_like(sumlogdet)
        sldb = torch.zeros_like(sumlogdet)
        history: dict = {}
        h_init = self._hamiltonian_n(xn, vn, beta)
        verbose = self.config.verbose
        if verbose:
            m = {'energy': h_init, 'logprob': h_init - sumlogdet, 'logdet': sumlogdet,
                 'sldf': sldf, 'sldb': sldb, 'sld': sumlogdet,
                 'xeps': self.xeps[0], 'veps': self.veps[0]}
            self.update_history(m, history)
        h = h_init
        cache = {} if self.reuse_v_inputs else None
        can_pair = (self.pair_v_updates and cache is not None and self.group == 'SU3'
                    and self._networks_built)
        # per-step metrics need the state between the paired updates: with verbose=True the pair
        # kernel returns the first update's logdet and the kinetic energy after it (`mid`), and
        # the metrics of a step are emitted once its closing update has run (one call later)
        vpair = verbose and can_pair and self.pair_v_updates_verbose and self._can_pair_mid()
        pend = {} if (can_pair and (not verbose or vpair)) else None
        nlf = self.config.nleapfrog
        deferred: Optional[dict] = None

        def emit(d, ke):
            energy = ke + d['pe']
            if d['fwd']:
                extras = {'sldf': sldf, 'sldb': sldb, 'sld': sumlogdet}
            else:
                extras = {'sldf': torch.zeros_like(sldb), 'sldb': sldb, 'sld': sumlogdet}
            mt = {'energy': energy, 'logprob': energy - sumlogdet, 'logdet': sumlogdet}
            mt.update(extras)
            mt.update({'xeps': self.xeps[d['step']], 'veps': self.veps[d['step']]})
            self.update_history(mt, history)
            return energy

        for step in range(nlf):
            mid = {} if vpair else None
            logdet = self._lf_n(step, x_, v_, beta, True, cache, pend, mid,
                                x_src=xn if (lazy_x and step == 0) else None,
                                v_src=vn if (lazy_x and step == 0) else None)
            if vpair:
                if deferred is not None:               # the previous step's closing update ran now
                    sumlogdet = sumlogdet + mid['ld1']
                    sldf = sldf + mid['ld1']
                    emit(deferred, mid['ke'])
                sumlogdet = sumlogdet + logdet
                sldf = sldf + logdet
                deferred = {'pe': self._potential_n(x_, beta), 'step': step, 'fwd': True}
                continue
            sumlogdet = sumlogdet + logdet
            if verbose:
                sldf = sldf + logdet
                extras = {'sldf': sldf, 'sldb': sldb, 'sld': sumlogdet}
                self.update_history(self._metrics_n(x_, v_, beta, sumlogdet, step, extras),
                                    history)
        if pend is not None and 'p' in pend:
            st0, f0, _ = pend['p']
            pend['p'] = (st0, f0, True)                # flip happens inside the paired kernel
        else:
            v_ = self._flip_v_n(v_)
        for step in range(nlf):
            mid = {} if vpair else None
            logdet = self._lf_n(step, x_, v_, beta, False, cache, pend, mid)
            if vpair:
                if deferred is not None:
                    sumlogdet = sumlogdet + mid['ld1']
                    if deferred['fwd']:
                        sldf = sldf + mid['ld1']
                    else:
                        sldb = sldb + mid['ld1']
                    emit(deferred, mid['ke'])
                sumlogdet = sumlogdet + logdet
                sldb = sldb + logdet
                deferred = {'pe': self._potential_n(x_, beta), 'step': nlf - step - 1, 'fwd': False}
                continue
            sumlogdet = sumlogdet + logdet
            if verbose:
                sldb = sldb + logdet
                extras = {'sldf': torch.zeros_like(sldb), 'sldb': sldb, 'sld': sumlogdet}
                mt = self._metrics_n(x_, v_, beta, sumlogdet,
                                     self.config.nleapfrog - step - 1, extras)
                h = mt['energy']
                self.update_history(mt, history)
        last = self._flush_pending_n(pend, x_, v_, beta, cache)
        if last is not None:
            sumlogdet = sumlogdet + last
        if vpair and deferred is not None:
            if last is not None:
                if deferred['fwd']:
                    sldf = sldf + last
                else:
                    sldb = sldb + last
            h = emit(deferred, self._kinetic_n(v_))
        if not verbose or self.config.nleapfrog == 0:
            h = self._hamiltonian_n(x_, v_, beta)
        acc = self._accept_prob_n(h_init, h, sumlogdet)
        history.update({'acc': acc, 'sumlogdet': sumlogdet})
        if verbose:
            history = self._stack_history(history)
        return x_, v_, history

    def _kernel_n(self, xn, vn, beta, forward: bool):
        """Single-direction kernel (dynamics.py:1031-1063), including the reference's swapped
        arguments to compute_accept_prob."""
        nb = xn.shape[0]
        x_, v_ = xn.clone(), vn.clone()
        sumlogdet = self._zeros_nb(nb)
        history: dict = {}
        h0 = self._hamiltonian_n(xn, vn, beta)
        if self.config.verbose:
            self.update_history({'energy': h0, 'logprob': h0 - sumlogdet,
                                 'logdet': sumlogdet}, history)
        cache = {} if self.reuse_v_inputs else None
        for step in range(self.config.nleapfrog):
            logdet = self._lf_n(step, x_, v_, beta, forward, cache)
            sumlogdet = sumlogdet + logdet
            if self.config.verbose:
                self.update_history(self._metrics_n(x_, v_, beta, sumlogdet, step), history)
        h1 = self._hamiltonian_n(x_, v_, beta)
        acc = self._accept_prob_n(h1, h0, sumlogdet)          # state_init=final, prop=initial
        history.update({'acc': acc, 'sumlogdet': sumlogdet})
        if self.config.verbose:
            history = self._stack_history(history)
        return x_, v_, history

    # ---- reference-shaped wrappers
    def _state_n(self, state: State):
        xn = self._pack(state.x)
        vn = self._pack(state.v) if self.group == 'SU3' else \
            state.v.to(DEVICE).reshape(xn.shape[0], -1).contiguous()
        return xn, vn

    def _state_from_n(self, xn, vn, beta, lazy: bool = False) -> State:
        if lazy and self.group == 'SU3':
            shape = (xn.shape[0], *self.xshape[1:])
            return State(x=lambda: self._unpack(xn), v=lambda: self._unpack(vn), beta=beta,
                         xshape=shape)
        v = self._unpack(vn) if self.group == 'SU3' else vn
        return State(x=self._unpack(xn), v=v, beta=beta)

    def transition_kernel_hmc(self, state: State, eps: Optional[float] = None,
                              nleapfrog: Optional[int] = None) -> tuple[State, dict]:
        xn, vn = self._state_n(state)
        x_, v_, hist = self._kernel_hmc_n(xn, vn, state.beta, eps, nleapfrog)
        return self._state_from_n(x_, v_, state.beta), hist

    def transition_kernel_fb(self, state: State) -> tuple[State, dict]:
        xn, vn = self._state_n(state)
        x_, v_, hist = self._kernel_fb_n(xn, vn, state.beta)
        return self._state_from_n(x_, v_, state.beta), hist

    def transition_kernel(self, state: State, forward: bool) -> tuple[State, dict]:
        xn, vn = self._state_n(state)
        x_, v_, hist = self._kernel_n(xn, vn, state.beta, forward)
        return self._state_from_n(x_, v_, state.beta), hist

    def compute_accept_prob(self, state_init: State, state_prop: State,
                            sumlogdet: Tensor) -> Tensor:
        return self._accept_prob_n(self.hamiltonian(state_init), self.hamiltonian(state_prop),
                                   sumlogdet.to(DEVICE))

    def _get_accept_masks(self, px: Tensor) -> tuple[Tensor, Tensor]:
        """ma = (px > U[0,1)) as float32, mr = 1 - ma (dynamics.py:1081-1087)"""
        u = self._uniform(px)
        acc = (px > u).to(torch.float)
        return acc, torch.ones_like(acc) - acc

    @staticmethod
    def _get_direction_masks(batch_size: int) -> tuple[Tensor, Tensor]:
        fwd = (torch.rand(batch_size) > 0.5).to(torch.float).to(DEVICE)
        return fwd, torch.ones_like(fwd) - fwd

    # ------------------------------------------------------------------ full transitions
    def _finish(self, xn, vn, x_, v_, beta, hist, with_sumlogdet: bool):
        """accept/reject + select (dynamics.py:660-702); returns (x_out[nb, xdim], metrics)."""
        nb = xn.shape[0]
        acc = hist['acc']
        u = self._uniform(acc)
        ma = (acc > u).to(torch.float32)
        fuse = self.group == 'SU3' and not self.cache_native_output
        if fuse:
            # select + native -> reference transpose in ONE pass (the native x_out is only needed by the
            # opt-in native-output cache)
            xo_n = None
            xout = ops.su3_unpack_select(x_, xn, ma, self.latvolume).reshape(nb, -1)
        else:
            xo_n = ops.select_rows(x_.reshape(nb, -1), xn.reshape(nb, -1), ma).reshape(xn.shape)
            xout = self._unpack(xo_n).reshape(nb, -1)
        # reference-layout copies of the init / proposed / out states are made on first access
        init = self._state_from_n(xn, vn, beta, lazy=True)
        prop = self._state_from_n(x_, v_, beta, lazy=True)
        if self.group == 'SU3':
            # the selected momentum is formed on first access as well (nothing in a sampler loop
            # reads it); v_ / vn are this trajectory's own tensors and are not written again
            out = State(x=xout, v=lambda: self._unpack(ops.select_rows(
                v_.reshape(nb, -1), vn.reshape(nb, -1), ma).reshape(vn.shape)).reshape(nb, -1),
                beta=beta, xshape=xout.shape)
            # (kept only while it is small next to the HBM: the 16^4 shard would pin 9.7 GB)
            keep = (self.cache_native_output and xo_n is not None
                    and xo_n.numel() * xo_n.element_size() <= (1 << 30))
            self._xcache = (xout, xout._version, xo_n) if keep else None
        else:
            vo_n = ops.select_rows(v_.reshape(nb, -1), vn.reshape(nb, -1), ma).reshape(vn.shape)
            out = State(x=xout, v=vo_n.reshape(nb, -1), beta=beta)
        mc_states = MonteCarloStates(init=init, proposed=prop, out=out)
        if with_sumlogdet:
            hist.update({'beta': beta, 'sumlogdet': ma * hist['sumlogdet']})
        hist.update({'acc_mask': ma, 'mc_states': mc_states})
        if getattr(self, '_capturing', False) and self.group == 'SU3':
            hist['_native'] = (xn, vn, x_, v_, ma)     # for _auto_graphed: fresh lazy states per replay
        return xout, hist

    def _follow_autocast(self) -> None:
        """The reference selects 16-bit networks by wrapping this call in `torch.autocast(dtype=...)`
        (trainers/pytorch/trainer.py:211-219, 1276-1280).  An autocast region around `forward` switches the
        U(1) networks to that type (as `set_net_precision` does) and leaving it switches them back; an explicit
        `set_net_precision` is never overridden.  SU(3) is complex128 by definition: ignored."""
        if self.group != 'U1' or not self._networks_built:
            return
        dt = DEVICE.type if isinstance(DEVICE, torch.device) else str(DEVICE).split(':')[0]
        half = None
        if torch.is_autocast_enabled(dt):
            half = torch.get_autocast_dtype(dt)
            half = half if half in (torch.float16, torch.bfloat16) else None
        if half is not None and self.net_precision is None:
            self.set_net_precision(half)
            self._precision_from_autocast = True
        elif half is None and getattr(self, '_precision_from_autocast', False):
            self.set_net_precision(None)
            self._precision_from_autocast = False
        elif half is not None and getattr(self, '_precision_from_autocast', False) \
                and self.net_precision != half:
            self.set_net_precision(half)

    def forward(self, inputs: tuple[Tensor, Tensor]) -> tuple[Tensor, dict]:
        self._follow_autocast()
        if self._records_graph():
            return self._forward_train(inputs)
        return (self.apply_transition_fb(inputs) if self.config.merge_directions
                else self.apply_transition(inputs))

    # ---- train mode: the trajectory as a node of the caller's autograd graph
    def _records_graph(self) -> bool:
        """Train mode with grad mode on and trainable parameters: what the reference's
        `forward_step` + `loss.backward()` (trainers/pytorch/trainer.py:1266-1314) runs under."""
        return (self.training and self._networks_built and self.autograd_forward
                and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()))

    def _forward_train(self, inputs: tuple[Tensor, Tensor]) -> tuple[Tensor, dict]:
        """`forward` whose outputs carry a grad_fn (dynamics/pytorch/autograd.py): x_out,
        metrics['acc'], metrics['sumlogdet'] and the proposed / output states of
        metrics['mc_states'] back-propagate into `self.parameters()` through the hand-written reverse
        sweep.  Same draw order as the sampling path: (direction,) momenta, dropout masks, accept
        uniforms."""
        from l2hmc import _autograd as AG
        from l2hmc.dynamics.pytorch import autograd as G
        x, beta = inputs
        direction = None
        if not self.config.merge_directions:
            inj = self._inject.get('forward') if self._inject else None
            direction = bool(torch.rand(1) > 0.5) if inj is None else bool(inj)
        xd = x.to(DEVICE)
        side: dict = {}
        xp, vp, sld, acc = G.transition(self, xd, beta, direction, side)
        nb = xp.shape[0]
        hist = side['hist']
        xn, vn = side['xn'], side['vn']
        u = self._uniform(acc.detach())
        ma = (acc.detach() > u).to(torch.float32)
        x_init = xd.detach().reshape(xp.shape)
        xout = AG.SelectRows.apply(xp, x_init, ma).reshape(nb, -1)
        su3 = self.group == 'SU3'
        v_init = (lambda: self._unpack(vn)) if su3 else vn
        init = State(x=x.reshape(xp.shape), v=v_init, beta=beta, xshape=tuple(xp.shape))
        prop = State(x=xp, v=vp, beta=beta, xshape=tuple(xp.shape))

        def v_out():
            vi = self._unpack(vn) if su3 else vn
            return AG.SelectRows.apply(vp, vi.reshape(vp.shape), ma).reshape(nb, -1)
        out = State(x=xout, v=v_out, beta=beta, xshape=tuple(xout.shape))
        hist.update({'acc': acc, 'beta': beta, 'sumlogdet': ma * sld, 'acc_mask': ma,
                     'mc_states': MonteCarloStates(init=init, proposed=prop, out=out)})
        return xout, hist

    # ---- transitions replayed from a HIP graph, transparently
    AUTO_GRAPH_MAX_ELEMS = 1 << 21      # U(1): chains x links up to which a trajectory is launch-bound
    AUTO_GRAPH_MAX_BYTES_SU3 = 1 << 30  # SU(3): a graph pins every temporary of a trajectory; not at 16^4

    AUTO_GRAPH_KEEP = 8

    def _auto_graphed(self, mode: str, x: Tensor, beta, eps=None, nleapfrog=None):
        """Eval-mode transitions are captured once per (batch, beta, step size) as a HIP graph
        (`GraphedTransition`: re-captured when the model changes) and replayed behind the ordinary
        `forward` / `apply_transition_hmc` calls.
          * small U(1) lattices (BASELINE cfg-1 / cfg-2, the reference's published 16 x 16 runs) spend more
            host time launching their ~30 us kernels than the GPU spends running them (cfg-2: 3.4 ms per
            trajectory eager for 1.9 ms of kernels): default ON (`auto_graph`);
          * SU(3) 8^4 is kernel-bound: a replay is worth ~1 % (21.2 vs 21.5 ms per trajectory), so it is
            OPT-IN (`auto_graph_su3 = True`).  (A 10 % gain measured earlier in round 5 turned out to hinge on a
            memset node whose presence also corrupted replays after a large device-to-host copy on the null
            stream -- profiles/r05k_graph_memset_node.txt -- and the library records no memset node any more.)
        The caller's contract is unchanged: x_out and the [nb]-sized metrics are COPIED out of the graph's
        static buffers; the SU(3) `mc_states` fields, which are formed on first access anyway, read the
        graph's native buffers and therefore must be read before the NEXT transition of this sampler (they
        raise afterwards instead of returning another trajectory's data).  Not taken with injected /
        host-generator draws (parity runs), in train mode, with the native-output cache, or where the pinned
        temporaries would be large (SU(3) fields > 1 GiB: the 16^4 shard).  `dyn.auto_graph = False`
        restores eager launches everywhere."""
        if not (self.auto_graph and not self.training and self._inject is None
                and (self._networks_built or mode == 'hmc') and not getattr(self, '_capturing', False)
                and isinstance(x, Tensor) and x.is_cuda and self.rng_device == DEVICE
                and not self.cache_native_output
                and not torch.cuda.is_current_stream_capturing()):
            return None
        if self.group == 'U1':
            if x.numel() > self.AUTO_GRAPH_MAX_ELEMS:
                return None
        elif not self.auto_graph_su3 or x.numel() * 16 > self.AUTO_GRAPH_MAX_BYTES_SU3 or mode != 'fb':
            # (plain HMC at 8^4 measures 7 % SLOWER from a graph -- 251k vs 269k chain*LF/s, profiles/r05i_* --:
            # only the L2HMC trajectory is replayed, and only on request)
            return None
        b = _beta(beta)
        key = (mode, tuple(x.shape), b, None if eps is None else float(eps), nleapfrog)
        g = self._graphs.pop(key, None)
        if g is None:
            seen = self._graph_seen.get(key, 0) + 1
            if len(self._graph_seen) > 64:
                self._graph_seen.clear()
            self._graph_seen[key] = seen
            if seen < int(self.auto_graph_after):
                return None                                   # eager until the key has proved recurrent
            while len(self._graphs) >= (self.AUTO_GRAPH_KEEP if self.group == 'U1' else 2):
                self._graphs.pop(next(iter(self._graphs)))     # least recently used (dict order = use order)
            # the warm-up trajectories and the capture draw from the device generator: put its state back
            # afterwards, so that the random stream the caller sees does not depend on WHEN a graph was captured
            rng = torch.cuda.get_rng_state(x.device)
            self._capturing = True
            try:
                g = GraphedTransition(self, x.detach(), b, mode, eps, nleapfrog, warmup=2)
            finally:
                self._capturing = False
                torch.cuda.set_rng_state(rng, x.device)
        self._graphs[key] = g                                  # (re-inserted: most recently used last)
        self._capturing = True
        try:
            xo, m = g(x.detach())
        finally:
            self._capturing = False
        own = lambda t: t.clone() if isinstance(t, Tensor) else t
        out = {k: own(v) for k, v in m.items() if k not in ('mc_states', '_native')}
        xo = own(xo)
        nat = m.get('_native')
        if self.group == 'SU3' and nat is not None:
            xn, vn, x_, v_, ma = nat
            stamp, nb = g.replays, xn.shape[0]

            def lazy(fn):
                def read():
                    if g.replays != stamp:
                        raise RuntimeError(
                            'mc_states of a graph-replayed transition were read after the NEXT transition of '
                            'the same sampler had run (its buffers hold that trajectory now): read them right '
                            'after the call, or set dynamics.auto_graph = False')
                    return fn()
                return read
            shape = (nb, *self.xshape[1:])
            init = State(x=lazy(lambda: self._unpack(xn)), v=lazy(lambda: self._unpack(vn)), beta=beta,
                         xshape=shape)
            prop = State(x=lazy(lambda: self._unpack(x_)), v=lazy(lambda: self._unpack(v_)), beta=beta,
                         xshape=shape)
            outs = State(x=xo, v=lazy(lambda: self._unpack(ops.select_rows(
                v_.reshape(nb, -1), vn.reshape(nb, -1), ma).reshape(vn.shape)).reshape(nb, -1)),
                beta=beta, xshape=xo.shape)
            out['mc_states'] = MonteCarloStates(init=init, proposed=prop, out=outs)
        else:
            v = m['mc_states']
            out['mc_states'] = MonteCarloStates(*(State(x=own(st.x), v=own(st.v), beta=beta)
                                                  for st in (v.init, v.proposed, v.out)))
        if 'beta' in out:
            out['beta'] = beta
        return xo, out

    def apply_transition_hmc(self, inputs: tuple[Tensor, Tensor], eps: Optional[float] = None,
                             nleapfrog: Optional[int] = None) -> tuple[Tensor, dict]:
        x, beta = inputs
        hit = self._auto_graphed('hmc', x, beta, eps, nleapfrog)
        if hit is not None:
            return hit
        xn = self._pack_input(x)
        vn = self._momentum_n(xn.shape[0])
        x_, v_, hist = self._kernel_hmc_n(xn, vn, beta, eps, nleapfrog)
        return self._finish(xn, vn, x_, v_, beta, hist, with_sumlogdet=False)

    def apply_transition_fb(self, inputs: tuple[Tensor, Tensor]) -> tuple[Tensor, dict]:
        x, beta = inputs
        hit = self._auto_graphed('fb', x, beta)
        if hit is not None:
            return hit
        xn = self._pack_input(x)
        vn = self._momentum_n(xn.shape[0])
        x_, v_, hist = self._kernel_fb_n(xn, vn, beta)
        return self._finish(xn, vn, x_, v_, beta, hist, with_sumlogdet=True)

    def apply_transition(self, inputs: tuple[Tensor, Tensor]) -> tuple[Tensor, dict]:
        x, beta = inputs
        forward = bool(torch.rand(1) > 0.5)
        xn = self._pack_input(x)
        vn = self._momentum_n(xn.shape[0])
        x_, v_, hist = self._kernel_n(xn, vn, beta, forward)
        return self._finish(xn, vn, x_, v_, beta, hist, with_sumlogdet=True)

    def apply_transition_both(self, inputs: tuple[Tensor, Tensor]) -> tuple[Tensor, dict]:
        """Forward and backward single-direction proposals from the same x (two momentum draws),
        mixed per chain by a random direction mask, then accept / reject
        (dynamics.py:744-803, including its per-key masking of the metrics)."""
        x, beta = inputs
        nb = x.shape[0]
        xn = self._pack(x)
        vf0 = self._momentum_n(nb)
        xf, vf, mf_hist = self._kernel_n(xn, vf0, beta, True)
        vb0 = self._momentum_n(nb)
        xb, vb, mb_hist = self._kernel_n(xn, vb0, beta, False)
        mf_, mb_ = self._get_direction_masks(batch_size=nb)

        def mix(a, b):
            flat = ops.select_rows(a.reshape(nb, -1), b.reshape(nb, -1), mf_)
            return flat.reshape(a.shape)
        v_init, xp, vp = mix(vf0, vb0), mix(xf, xb), mix(vf, vb)
        logdetp = mf_ * mf_hist['sumlogdet'] + mb_ * mb_hist['sumlogdet']
        acc = mf_ * mf_hist['acc'] + mb_ * mb_hist['acc']
        ma_, mr_ = self._get_accept_masks(acc)
        xo_n = ops.select_rows(xp.reshape(nb, -1), xn.reshape(nb, -1), ma_).reshape(xn.shape)
        vo_n = ops.select_rows(vp.reshape(nb, -1), v_init.reshape(nb, -1), ma_).reshape(vp.shape)
        metrics = {}
        for (key, valf), (_, valb) in zip(mf_hist.items(), mb_hist.items()):
            if isinstance(valf, Tensor) and valf.shape[-1:] == (nb,):
                metrics[key] = ma_ * (mf_ * valf + mb_ * valb)
        mc_states = MonteCarloStates(init=self._state_from_n(xn, v_init, beta, lazy=True),
                                     proposed=self._state_from_n(xp, vp, beta, lazy=True),
                                     out=self._state_from_n(xo_n, vo_n, beta, lazy=True))
        metrics.update({'acc': acc, 'acc_mask': ma_, 'sumlogdet': ma_ * logdetp,
                        'mc_states': mc_states})
        return self._unpack(xo_n).reshape(nb, -1), metrics

    def get_metrics(self, state: State, logdet: Tensor, step: Optional[int] = None,
                    extras: Optional[dict] = None) -> dict:
        """energy / logprob / logdet (+ step sizes) of a reference-layout state (:865-886)."""
        xn, vn = self._state_n(state)
        return self._metrics_n(xn, vn, state.beta, logdet.to(DEVICE), step, extras)

    # plain-HMC pieces with the *trainable* step sizes (dynamics.py:1244-1264)
    def _update_v_fwd_hmc(self, step: int, state: State) -> Tensor:
        xn, vn = self._state_n(state)
        vn = vn.clone()
        self._kick_n(xn, vn, state.beta, -0.5 * self._eps('v', step))
        return self._unpack(vn) if self.group == 'SU3' else vn.reshape(state.v.shape)

    def _update_v_bwd_hmc(self, step: int, state: State) -> Tensor:
        xn, vn = self._state_n(state)
        vn = vn.clone()
        self._kick_n(xn, vn, state.beta, 0.5 * self._eps('v', step))
        return self._unpack(vn) if self.group == 'SU3' else vn.reshape(state.v.shape)

    def _update_x_hmc(self, step: int, state: State, sign: float) -> Tensor:
        xn, vn = self._state_n(state)
        eps = sign * self._eps('x', step)
        if self.group == 'SU3':
            return self._unpack(ops.su3_expm_mul_n(xn, vn, eps))
        xn = xn.clone()
        ops.axpy_(xn.reshape(xn.shape[0], -1), vn, eps)
        return self._unpack(xn)

    def _update_x_fwd_hmc(self, step: int, state: State) -> Tensor:
        return self._update_x_hmc(step, state, +1.0)

    def _update_x_bwd_hmc(self, step: int, state: State) -> Tensor:
        return self._update_x_hmc(step, state, -1.0)

    def _stack_as_xy(self, x: Tensor) -> Tensor:
        """[cos(x), sin(x)] stacked on a new last axis (dynamics.py:1137-1140); U(1) fields."""
        x = x.to(DEVICE).contiguous()
        nb, n = x.shape[0], x[0].numel()
        ones = torch.ones(n, dtype=torch.float32, device=DEVICE)
        cs = ops.u1_masked_cos_sin(x.reshape(nb, n), ones, False, (n // 2,)).reshape(nb, 2, n)
        return torch.stack([cs[:, 0].reshape(x.shape), cs[:, 1].reshape(x.shape)], dim=-1)

    @staticmethod
    def complexify(x: Tensor, dim: int = 1) -> Tensor:
        """real pairs along `dim` -> complex (dynamics.py:1501-1535)"""
        assert len(x.shape) >= 2
        assert x.shape[dim] == 2
        if dim != 1:
            xr, xi = x.transpose(0, dim)
            return torch.complex(xr.transpose(0, dim - 1), xi.transpose(0, dim - 1))
        if len(x.shape) == 2:
            return torch.complex(x[:, 0], x[:, 1])
        return torch.complex(x[:, 0, ...], x[:, 1, ...])

    def generate_proposal_hmc(self, inputs, eps=None, nleapfrog=None) -> dict:
        x, beta = inputs
        xn = self._pack(x)
        vn = self._momentum_n(xn.shape[0])
        x_, v_, hist = self._kernel_hmc_n(xn, vn, beta, eps, nleapfrog)
        return {'init': self._state_from_n(xn, vn, beta),
                'proposed': self._state_from_n(x_, v_, beta), 'metrics': hist}

    def generate_proposal_fb(self, inputs) -> dict:
        x, beta = inputs
        xn = self._pack(x)
        vn = self._momentum_n(xn.shape[0])
        x_, v_, hist = self._kernel_fb_n(xn, vn, beta)
        return {'init': self._state_from_n(xn, vn, beta),
                'proposed': self._state_from_n(x_, v_, beta), 'metrics': hist}

    def generate_proposal(self, inputs, forward: bool) -> dict:
        x, beta = inputs
        xn = self._pack(x)
        vn = self._momentum_n(xn.shape[0])
        x_, v_, hist = self._kernel_n(xn, vn, beta, forward)
        return {'init': self._state_from_n(xn, vn, beta),
                'proposed': self._state_from_n(x_, v_, beta), 'metrics': hist}

    def make_graphed(self, x: Tensor, beta: float, mode: str = 'fb', eps: Optional[float] = None,
                     nleapfrog: Optional[int] = None, warmup: int = 2) -> 'GraphedTransition':
        """Capture one whole transition (momenta, 2N leapfrog steps, accept/select) into a HIP
        graph.  The U(1) configs are launch-bound (~40 small kernels per leapfrog step on a
        16 x 16 lattice); replaying a graph removes the per-launch host cost.  `beta` must be a
        Python float (a tensor would need a host read inside the capture)."""
        return GraphedTransition(self, x, float(beta), mode, eps, nleapfrog, warmup)

    def random_state(self, beta: float) -> State:
        x = self.g.random(list(self.xshape)).to(self.device)
        v = self.g.random_momentum(list(self.xshape)).to(x.device)
        return State(x=x, v=v, beta=torch.tensor(beta).to(self.device))

    def test_reversibility(self) -> dict[str, Array]:
        state = self.random_state(beta=1.)
        state_fwd, _ = self.transition_kernel(state, forward=True)
        state_, _ = self.transition_kernel(state_fwd, forward=False)
        dx = torch.abs(state.x - state_.x.reshape(state.x.shape))
        dv = torch.abs(state.v - state_.v.reshape(state.v.shape))
        return {'dx': dx.detach().cpu().numpy(), 'dv': dv.detach().cpu().numpy()}


class GraphedTransition:
    """A transition of `Dynamics` recorded once as a HIP graph and replayed with new inputs.

        g = dyn.make_graphed(x, beta=4.0)          # 'fb' (Dynamics.forward) or 'hmc'
        x_out, metrics = g(x)                      # same outputs as dyn((x, beta))

    Outputs are views of the graph's static buffers: valid until the next call (clone to keep).
    Random draws come from the device generator (graph-safe philox offsets), i.e. successive
    replays draw fresh momenta / accept uniforms.

    What a captured graph freezes, and how that is kept safe:
    * device pointers of the kernel-order weight copies (`LeapfrogLayer.kernel_weights`), the
      native masks and the reduction workspace; host floats (step sizes, beta) as kernel
      arguments.  The graph keeps references to the weight / mask copies it recorded and pins the
      workspace block (`native.pin_workspace`: a later, larger request allocates a new block
      instead of freeing this one), so a replay never reads freed memory.
    * parameters.  `__call__` compares a signature of everything the capture depended on --
      `ops.PARAM_GENERATION` (bumped by the fused Adam step / checkpoint loads, which bypass
      `Tensor._version`), identity and version of every parameter (incl. xeps / veps: `assign_eps`
      makes new ones), the masks, the network precision and the fusion switches -- and
      RE-CAPTURES when it changed, so a replay after `train_step`, `load_ckpt`, `assign_eps`,
      `set_masks` or `set_net_precision` uses the current model."""

    def __init__(self, dyn: Dynamics, x: Tensor, beta: float, mode: str, eps, nleapfrog,
                 warmup: int):
        assert mode in ('fb', 'hmc')
        if not torch.cuda.is_available():
            raise RuntimeError('GraphedTransition needs a GPU')
        self.dyn, self.beta, self.mode = dyn, beta, mode
        self._eps, self._nleapfrog, self._warmup = eps, nleapfrog, max(1, warmup)
        self.static_x = x.to(DEVICE).clone()
        self.captures = 0
        self.replays = 0
        self._capture()

    def _signature(self) -> tuple:
        d = self.dyn
        # (the module tree is walked once per capture: on the U(1) configs a replay is 0.1-0.3 ms
        # and `d.parameters()` alone cost as much; step sizes are tracked by identity because
        # `assign_eps` replaces them)
        plist = getattr(self, '_plist', None)
        if plist is None:
            plist = self._plist = list(d.parameters())
        eps_ids = tuple(id(p) for p in d.xeps) + tuple(id(p) for p in d.veps)
        params = (eps_ids, tuple(p._version for p in plist))
        masks = tuple(id(m) for m in d.masks)
        flags = (d.fuse_heads, d.fuse_x_updates, d.reuse_v_inputs, d.pair_v_updates,
                 d.fuse_u1_steps, d.fuse_half_heads, d.merge_hmc_kicks, d.sliced_heads,
                 ops.USE_SLICED_HEADS[0], d.sliced_input, ops.USE_SLICED_INPUT[0], d.pair_v_updates_verbose,
                 getattr(d, 'fuse_x_vec8', None), d.net_precision, d.config.verbose, d.training)
        return (ops.PARAM_GENERATION[0], params, masks, flags)

    def _run(self):
        if self.mode == 'fb':
            return self.dyn.apply_transition_fb((self.static_x, self.beta))
        return self.dyn.apply_transition_hmc((self.static_x, self.beta), eps=self._eps,
                                             nleapfrog=self._nleapfrog)

    def _capture(self) -> None:
        from l2hmc import native
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):              # warm-up: caches, workspaces, weight copies
            for _ in range(self._warmup):
                self._run()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # everything the recorded kernels point at, kept alive for the life of this graph
        self._pinned = [native.pin_workspace(), list(self.dyn._native_masks())]
        from l2hmc.network.pytorch.network import LeapfrogLayer
        for m in self.dyn.modules():
            if isinstance(m, LeapfrogLayer):
                self._pinned.append(dict(m._head_cache))
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out_x, self.out_metrics = self._run()
        self._plist = None                         # re-walk the module tree for this capture
        self._sig = self._signature()
        self.captures += 1

    def __call__(self, x: Tensor):
        if self._signature() != self._sig:
            self._capture()                        # the model changed since the capture
        self.static_x.copy_(x.reshape(self.static_x.shape))
        self.graph.replay()
        self.replays += 1
        return self.out_x, self.out_metrics
