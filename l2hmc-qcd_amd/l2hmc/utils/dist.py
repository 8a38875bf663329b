"""Single-node data parallelism for the L2HMC sampler/trainer: independent Markov chains are
sharded across ranks (one process per GPU); the only exchange is one all-reduce(mean) of the
flattened parameter gradient per training step.

Counterpart of the reference's ``src/l2hmc/utils/dist.py`` (mpi4py bootstrap + NCCL/Gloo,
:115-346) and of ``DDP(dynamics)`` in ``trainers/pytorch/trainer.py:246-257``, without mpi4py /
horovod / deepspeed: rank and world size come from the torchrun-style environment
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT); backend ``nccl`` is RCCL over
xGMI on MI355X, ``gloo`` on CPU (tests).
"""
from __future__ import annotations

import os
import random
from typing import Iterable, Optional

import numpy as np
import torch
import torch.distributed as dist


def query_environment() -> dict[str, int]:
    """(dist.py:157-162 of the reference, minus MPI)"""
    return {'rank': int(os.environ.get('RANK', 0)),
            'local_rank': int(os.environ.get('LOCAL_RANK', 0)),
            'world_size': int(os.environ.get('WORLD_SIZE', 1))}


SEEDED = [False]          # set by seed_everything: lets Experiment tell an unseeded direct caller


def seed_everything(seed: int) -> None:
    """random / numpy / torch, like common.py:115-121 of the reference"""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    SEEDED[0] = True


def setup_torch_distributed(backend: Optional[str] = None, port: str = '2345') -> dict[str, int]:
    env = query_environment()
    if env['world_size'] > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(port))
        if backend is None or backend.upper() in ('DDP', 'NCCL', 'RCCL'):
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if torch.cuda.is_available():
            torch.cuda.set_device(env['local_rank'])
        dist.init_process_group(backend=backend.lower(), rank=env['rank'],
                                world_size=env['world_size'])
    return env


def setup_torch(seed: int, backend: Optional[str] = None, port: str = '2345',
                precision: Optional[str] = None) -> int:
    """Initialise the process group, the default dtype (`precision`: float16 / float32 / float64,
    like utils/dist.py:293-341 of the reference) and the RNGs.  Weights and masks must be identical on
    every rank, so the *model* seed is the base seed; per-rank streams (chains, momenta)
    use ``chain_seed`` (the reference seeds everything with seed*(rank+1)*(local_rank+1),
    dist.py:340, which makes the numpy masks differ per rank -- SURVEY.md 8(e))."""
    env = setup_torch_distributed(backend, port)
    if precision is not None:
        torch.set_default_dtype({'float16': torch.float16, 'float32': torch.float32,
                                 'float64': torch.float64}.get(precision, torch.float32))
    seed_everything(seed)
    return env['rank']


def chain_seed(seed: int, rank: Optional[int] = None) -> int:
    """Seed of this rank's chain streams (initial configurations, momenta, accept uniforms):
    distinct from the model seed on EVERY rank -- `seed * (rank + 1)` would replay on rank 0 the
    very stream that drew the weights and masks, and give all ranks the same chains for seed 0."""
    rank = query_environment()['rank'] if rank is None else rank
    return (int(seed) + 1_000_003 * (rank + 1)) % (2 ** 32 - 1)      # numpy wants < 2^32


def shard_chains(nchains_global: int, rank: Optional[int] = None,
                 world_size: Optional[int] = None) -> tuple[int, int]:
    """[start, stop) of this rank's contiguous block of the global chain index; chains never
    straddle ranks and every rank gets the same count (global batch must divide evenly, as with
    the reference's per-rank ``nchains``)."""
    env = query_environment()
    rank = env['rank'] if rank is None else rank
    world_size = env['world_size'] if world_size is None else world_size
    if nchains_global % world_size:
        raise ValueError(f'{nchains_global} chains do not divide over {world_size} ranks')
    per = nchains_global // world_size
    return rank * per, (rank + 1) * per


def flatten_grads(params: Iterable[torch.nn.Parameter]) -> tuple[torch.Tensor, list]:
    """One flat buffer of all existing gradients (params without grad -- e.g. the never-called
    SU(3) xnet -- are skipped, the ``find_unused_parameters`` case of trainer.py:249-251)."""
    ps = [p for p in params if p.grad is not None]
    if not ps:
        return torch.zeros(0), ps
    flat = torch.cat([p.grad.reshape(-1) for p in ps])
    return flat, ps


def allreduce_grads(params: Iterable[torch.nn.Parameter]) -> int:
    """grad <- mean over ranks, as ONE collective on one flat buffer (vs DDP's 25 MB buckets):
    xGMI is a point-to-point mesh, so a single large reduce-scatter + all-gather keeps all 7
    links busy.  Returns the number of elements reduced."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    flat, ps = flatten_grads(params)
    if flat.numel() == 0:
        return 0
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= dist.get_world_size()
    off = 0
    for p in ps:
        n = p.grad.numel()
        p.grad.copy_(flat[off:off + n].reshape(p.grad.shape))
        off += n
    return int(flat.numel())


def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> None:
    """rank-0 parameters / buffers to every rank (DDP constructor semantics)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    nccl = dist.get_backend() == 'nccl'
    for t in list(module.parameters()) + list(module.buffers()):
        if nccl and not t.is_cuda:
            # host-resident tensors (the never-called SU(3) xnet lives in host memory) cannot go
            # through RCCL directly: stage them through the device in <= 1 GiB pieces
            flat = t.data.reshape(-1)
            step = max(1, (1 << 30) // max(1, flat.element_size()))
            for lo in range(0, flat.numel(), step):
                buf = flat[lo:lo + step].cuda()
                dist.broadcast(buf, src=src)
                flat[lo:lo + step].copy_(buf.cpu())
        else:
            dist.broadcast(t.data, src=src)


def model_checksum(module: torch.nn.Module) -> str:
    """Order-dependent digest of every parameter, buffer and (Dynamics) mask of a model: equal on two ranks
    exactly when they hold the same bits (bench.py --gpus N asserts it; the data-parallel step relies on it)."""
    import hashlib
    h = hashlib.sha256()
    with torch.no_grad():
        ts = [p for _n, p in sorted(module.named_parameters())] + [b for _n, b in sorted(module.named_buffers())]
        ts += list(getattr(module, 'masks', []) or [])
        for t in ts:
            t = t.detach()
            # two order-sensitive moments in fp64 instead of shipping the bytes of 1.4 GB matrices to the host
            f = (torch.view_as_real(t) if t.is_complex() else t).double().reshape(-1)
            if f.numel() == 0:
                continue
            if f.numel() > (1 << 24) and not f.is_cuda:
                # (host-resident giants: the never-called SU(3) xnet) the plain sum only
                h.update(repr((tuple(t.shape), float(f.sum()))).encode())
                continue
            w = torch.arange(1, f.numel() + 1, dtype=torch.float64, device=f.device)
            h.update(repr((tuple(t.shape), float(f.sum()), float((f * w).sum() / f.numel()))).encode())
    return h.hexdigest()[:16]


def sync_model(dynamics, src: int = 0) -> None:
    """Make `dynamics` identical on every rank: broadcast rank `src`'s parameters, buffers and
    the numpy-drawn leapfrog masks.  No-op without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    broadcast_parameters(dynamics, src=src)
    masks = getattr(dynamics, 'masks', None)
    if masks:
        dev = next(dynamics.parameters()).device if any(True for _ in dynamics.parameters()) else 'cpu'
        if dist.get_backend() == 'gloo':
            dev = 'cpu'
        stacked = torch.stack([m.reshape(-1).float() for m in masks]).to(dev).contiguous()
        dist.broadcast(stacked, src=src)
        dynamics.set_masks([stacked[i].cpu().numpy() for i in range(stacked.shape[0])])
    from l2hmc import _ops as ops
    ops.PARAM_GENERATION[0] += 1          # cached kernel-order weight copies are stale now


class NativeComm:
    """RCCL communicator behind the C ABI (``l2q_comm_init`` / ``l2q_allreduce_grads``,
    include/l2q.h): the gradient all-reduce without PyTorch's process group on the data path.
    The 128-byte unique id is drawn on rank 0 and reaches the other ranks through the already
    initialised ``torch.distributed`` group (any backend; only the bootstrap uses it).  With
    ``world_size == 1`` nothing else is needed -- which is also how a single-GPU box proves that
    librccl loads and that the collective orders with the kernels on the launch stream."""

    def __init__(self, rank: Optional[int] = None, world_size: Optional[int] = None):
        import ctypes as C
        from l2hmc import native
        env = query_environment()
        if dist.is_available() and dist.is_initialized():
            rank = dist.get_rank() if rank is None else rank
            world_size = dist.get_world_size() if world_size is None else world_size
        self.rank = env['rank'] if rank is None else rank
        self.world_size = env['world_size'] if world_size is None else world_size
        lib = native.load()
        ident = C.create_string_buffer(128)
        if self.rank == 0:
            rc = lib.l2q_comm_unique_id(ident)
            if rc != 0:
                raise native.L2QError(f'l2q_comm_unique_id failed ({rc}): {lib.l2q_last_error().decode()}')
        if self.world_size > 1:
            box = [ident.raw if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            ident = C.create_string_buffer(box[0], 128)
        comm = C.c_void_p()
        rc = lib.l2q_comm_init(ident, self.world_size, self.rank, C.byref(comm))
        if rc != 0:
            raise native.L2QError(f'l2q_comm_init failed ({rc}): {lib.l2q_last_error().decode()}')
        self._comm = comm

    def all_reduce_(self, flat: torch.Tensor) -> torch.Tensor:
        """in-place sum of a contiguous fp32 / fp64 device buffer over ranks, on the current stream"""
        from l2hmc import native
        if flat.dtype not in (torch.float32, torch.float64):
            raise TypeError(f'all_reduce_: {flat.dtype} (fp32 / fp64 gradient buffers only)')
        native.call('l2q_allreduce_grads', self._comm, flat, flat.numel(), flat.element_size())
        return flat

    def close(self, abort: bool = False) -> None:
        """orderly ncclCommDestroy; abort=True: ncclCommAbort (no handshake with the peers)"""
        comm, self._comm = getattr(self, '_comm', None), None
        if comm is not None and comm.value:
            from l2hmc import native
            lib = native.load()
            (lib.l2q_comm_abort if abort else lib.l2q_comm_destroy)(comm)

    # a communicator is a device-side resource: release it with the object / the `with` block
    def __enter__(self) -> 'NativeComm':
        return self

    def __exit__(self, *exc) -> None:
        self.close()

    def __del__(self):
        # garbage collection / interpreter shutdown: the peers may be gone and ncclCommDestroy could wait for
        # them -- abort instead unless this is a lone rank (use the context manager for an orderly close)
        try:
            self.close(abort=getattr(self, 'world_size', 1) > 1)
        except Exception:      # noqa: BLE001  (interpreter shutdown: the library may be gone)
            pass


def cleanup() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
