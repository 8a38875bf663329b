"""world_size-2 gloo tests (CPU) of the N>1 path: chain sharding, per-rank seeds, the single
flat gradient all-reduce and parameter broadcast."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from l2hmc.utils import dist as D
    assert D.setup_torch(seed=1234, backend='gloo') == rank
    lo, hi = D.shard_chains(16)
    # identical model on every rank (base seed), different chains
    model = torch.nn.Sequential(torch.nn.Linear(6, 4), torch.nn.Tanh(), torch.nn.Linear(4, 1))
    unused = torch.nn.Linear(3, 3)                  # never gets a gradient (SU3 xnet case)
    w0 = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    g = torch.Generator().manual_seed(99)           # the same global data on both ranks
    data = torch.randn(16, 6, generator=g)
    loss = model(data[lo:hi]).pow(2).mean()
    loss.backward()
    n = D.allreduce_grads(list(model.parameters()) + list(unused.parameters()))
    grads = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    with torch.no_grad():
        for p in model.parameters():
            p.add_(float(rank + 1))                 # make the replicas diverge
    D.broadcast_parameters(model, src=0)
    w1 = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    torch.save({'lo': lo, 'hi': hi, 'w0': w0, 'grads': grads, 'n': n, 'w1': w1,
                'chain_seed': D.chain_seed(1234)}, os.path.join(out, f'r{rank}.pt'))
    D.cleanup()


def test_two_rank_gloo(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f'r{i}.pt') for i in range(world)]
    assert (r[0]['lo'], r[0]['hi'], r[1]['lo'], r[1]['hi']) == (0, 8, 8, 16)
    assert torch.equal(r[0]['w0'], r[1]['w0'])                     # same weights everywhere
    assert r[0]['chain_seed'] != r[1]['chain_seed']
    assert torch.equal(r[0]['grads'], r[1]['grads'])               # averaged gradient
    # equals the single-process gradient of the global-batch mean loss
    torch.manual_seed(1234)
    model = torch.nn.Sequential(torch.nn.Linear(6, 4), torch.nn.Tanh(), torch.nn.Linear(4, 1))
    g = torch.Generator().manual_seed(99)
    data = torch.randn(16, 6, generator=g)
    model(data).pow(2).mean().backward()
    ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    assert torch.allclose(r[0]['grads'], ref, atol=1e-7)
    assert r[0]['n'] == ref.numel()                                # unused params skipped
    assert torch.equal(r[0]['w1'], r[1]['w1'])                     # broadcast from rank 0
    assert torch.allclose(r[0]['w1'], r[0]['w0'] + 1.0)


def test_shard_chains_errors():
    sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
    from l2hmc.utils import dist as D
    assert D.shard_chains(2048, rank=3, world_size=8) == (768, 1024)
    with pytest.raises(ValueError):
        D.shard_chains(10, rank=0, world_size=4)


# ------------------------------------------------------------------ data-parallel train step
def _train_worker(rank, world, port, out):
    """Each rank: the same Dynamics (base seed), its shard of the chains, one train step through
    ParamArena.all_reduce (the path bench / Trainer use under torchrun).  Kernels are replaced
    by tests/emu_native.py -- this container has no GPU; the collective is real (gloo)."""
    sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import numpy as np
    torch.set_default_dtype(torch.float64)
    import emu_native
    import helpers
    from l2hmc import native
    from l2hmc.dynamics.pytorch import training as T
    from l2hmc.utils import dist as D
    native.call = emu_native.call
    if world > 1:
        assert D.setup_torch(seed=1234, backend='gloo') == rank
    g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'u1_train_f64_plain.npz')))
    nb = g['x'].shape[0] - 1                               # 5 chains in the file -> use 4
    lo, hi = (rank * nb // world, (rank + 1) * nb // world)
    gs = dict(g)
    for k in ('x', 'normals', 'u'):
        gs[k] = g[k][lo:hi]
    dyn, lat, loss_fn = helpers.build_u1_train_dynamics(gs)
    arena = T.ParamArena(dyn)
    arena.zero_grad()
    dyn._inject = {'normals': gs['normals'], 'u': gs['u']}
    x = dyn.g.compat_proj(dyn.unflatten(torch.from_numpy(gs['x'])))
    _, m, loss = T.train_forward_backward(dyn, loss_fn, x, torch.tensor(float(g['beta'])))
    scale = arena.all_reduce()
    grp = arena.groups[torch.float64]
    grads = (grp['grad'] * scale).clone()
    arena.adam_step(lr=1e-3, grad_scale=scale)
    torch.save({'grads': grads, 'params': grp['flat'].clone(), 'loss': float(loss), 'scale': scale},
               os.path.join(out, f't{world}_{rank}.pt'))
    if world > 1:
        D.cleanup()


def _overlap_worker(rank, world, port, out, fixture):
    """One train step twice on every rank: ONE blocking all-reduce after the sweep, then the exchange
    overlapped with the sweep (training.GradReducer: networks / matrices handed over as they finish) -- the
    flat gradient must come out bit-identical, with collectives started BEFORE the sweep's end."""
    sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import numpy as np
    torch.set_default_dtype(torch.float64)
    import emu_native
    import helpers
    from l2hmc import native
    from l2hmc.dynamics.pytorch import training as T
    from l2hmc.utils import dist as D
    native.call = emu_native.call
    import l2hmc._ops as ops
    ops.N.call = emu_native.call
    assert D.setup_torch(seed=1234, backend='gloo') == rank
    g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', fixture + '.npz')))
    su3 = fixture.startswith('su3')
    nb = g['x'].shape[0]
    lo, hi = (0, nb - 1) if rank == 0 else (1, nb)           # overlapping shards of unequal content
    gs = dict(g)
    gs['x'], gs['u'] = g['x'][lo:hi], g['u'][lo:hi]
    gs['normals'] = g['normals'][:, lo:hi] if su3 else g['normals'][lo:hi]
    build = helpers.build_su3_train_dynamics if su3 else helpers.build_u1_train_dynamics
    res = {}
    for mode in ('blocking', 'overlap'):
        dyn, lat, loss_fn = build(gs)
        arena = T.ParamArena(dyn, skip=list(dyn.xnet.parameters()) if su3 else None)
        arena.zero_grad()
        dyn._inject = {'normals': gs['normals'], 'u': gs['u']}
        x = dyn.g.compat_proj(dyn.unflatten(torch.from_numpy(gs['x'])))
        events = []
        if mode == 'overlap':
            red = T.GradReducer(arena)
            ready = red.ready

            def spy(params, _r=ready):
                n0 = red.launched
                _r(params)
                events.append(red.launched - n0)
            red.ready = spy
            T.train_forward_backward(dyn, loss_fn, x, torch.tensor(float(g['beta'])), reducer=red)
            before_finish = red.launched
            scale = red.finish()
        else:
            T.train_forward_backward(dyn, loss_fn, x, torch.tensor(float(g['beta'])))
            before_finish = 0
            scale = arena.all_reduce()
        res[mode] = {'grad': arena.groups[torch.float64]['grad'].clone(), 'scale': scale,
                     'early': before_finish, 'events': events}
    torch.save(res, os.path.join(out, f'o_{fixture}_{rank}.pt'))
    D.cleanup()


@pytest.mark.parametrize('fixture', ['u1_train_f64', 'su3_train'])
def test_overlapped_gradient_exchange_equals_blocking_all_reduce(fixture, tmp_path):
    port = _free_port()
    mp.spawn(_overlap_worker, args=(2, port, str(tmp_path), fixture), nprocs=2, join=True)
    r = [torch.load(tmp_path / f'o_{fixture}_{i}.pt', weights_only=False) for i in range(2)]
    for rk in r:
        assert rk['blocking']['scale'] == 0.5 and rk['overlap']['scale'] == 0.5
        assert torch.equal(rk['blocking']['grad'], rk['overlap']['grad'])          # same bits
        assert float(rk['blocking']['grad'].abs().max()) > 0
        # U(1): one collective per network as the sweep leaves it (nleapfrog 2: 2 vnets + 4 xnets); SU(3): per matrix of the
        # native-order shadows, biggest first
        assert rk['overlap']['early'] >= (6 if fixture.startswith('u1') else 5), rk['overlap']
    assert torch.equal(r[0]['overlap']['grad'], r[1]['overlap']['grad'])


def _consensus_worker(rank, world, port, out):
    """4 ranks whose memory gates see UNEQUAL headroom (rank 2 is short): every gate is decided by all ranks
    together (l2hmc._ops.mem_gate: all-reduce MIN before the first kernel that depends on it), and a rank
    that nevertheless reports another path table makes the overlapped exchange fall back to ONE blocking
    all-reduce on every rank (training.GradReducer._check_paths_once) instead of mismatched collectives."""
    sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import numpy as np
    torch.set_default_dtype(torch.float64)
    import emu_native
    import helpers
    from l2hmc import native
    from l2hmc.dynamics.pytorch import training as T
    from l2hmc.utils import dist as D
    native.call = emu_native.call
    import l2hmc._ops as ops
    ops.N.call = emu_native.call
    assert D.setup_torch(seed=1234, backend='gloo') == rank
    ops.device_headroom = lambda device: (1 << 20) if rank == 2 else (1 << 40)
    cpu = torch.device('cpu')
    res = {'big': ops.mem_gate('test: 1 GiB image', 1 << 30, 0.25, cpu),        # rank 2 cannot -> nobody does
           'small': ops.mem_gate('test: 1 KiB image', 1 << 10, 0.25, cpu),      # everybody can
           'again': ops.mem_gate('test: 1 GiB image', 1 << 30, 0.25, cpu)}      # (kept: no second collective)
    ops.device_headroom = lambda device: 1 << 40
    res['sticky'] = ops.mem_gate('test: 1 GiB image', 1 << 30, 0.25, cpu)       # the agreed answer stays
    res['sig'] = ops.mem_gate_signature()
    g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'u1_train_f64.npz')))
    nb = g['x'].shape[0]
    lo = rank % (nb - 1)
    gs = dict(g)
    for k in ('x', 'u', 'normals'):
        gs[k] = g[k][lo:lo + 2]
    for mode in ('blocking', 'overlap', 'overlap_paths_differ'):
        dyn, lat, loss_fn = helpers.build_u1_train_dynamics(gs)
        arena = T.ParamArena(dyn)
        arena.zero_grad()
        dyn._inject = {'normals': gs['normals'], 'u': gs['u']}
        x = dyn.g.compat_proj(dyn.unflatten(torch.from_numpy(gs['x'])))
        beta = torch.tensor(float(g['beta']))
        if mode == 'blocking':
            T.train_forward_backward(dyn, loss_fn, x, beta)
            early, scale = 0, arena.all_reduce()
        else:
            if mode == 'overlap_paths_differ':
                T.GradReducer._paths_checked = False
                if rank == 2:
                    ops.MEM_GATE_LOG['a gate only rank 2 refused'] = False
            red = T.GradReducer(arena)
            T.train_forward_backward(dyn, loss_fn, x, beta, reducer=red)
            early, scale = red.launched, red.finish()
            res[mode + '_blocking_flag'] = red.blocking
        res[mode] = {'grad': arena.groups[torch.float64]['grad'].clone(), 'scale': scale, 'early': early}
    torch.save(res, os.path.join(out, f'c{rank}.pt'))
    D.cleanup()


def test_world4_unequal_memory_gates_decide_by_consensus(tmp_path):
    port = _free_port()
    mp.spawn(_consensus_worker, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    r = [torch.load(tmp_path / f'c{i}.pt', weights_only=False) for i in range(4)]
    for rk in r:
        assert (rk['big'], rk['small'], rk['again'], rk['sticky']) == (False, True, False, False)
        assert rk['sig'] == r[0]['sig']
        assert rk['blocking']['scale'] == 0.25
        assert rk['overlap']['early'] >= 6 and rk['overlap_blocking_flag'] is False
        assert rk['overlap_paths_differ']['early'] == 0 and rk['overlap_paths_differ_blocking_flag'] is True
        gmax = float(rk['blocking']['grad'].abs().max())
        for mode in ('overlap', 'overlap_paths_differ'):
            # (4-rank ring sums associate differently for different message sizes: rounding-level differences
            # between the schedules, none between the ranks of one schedule)
            assert float((rk['blocking']['grad'] - rk[mode]['grad']).abs().max()) < 1e-13 * gmax
            assert torch.equal(rk[mode]['grad'], r[0][mode]['grad'])
        assert torch.equal(rk['blocking']['grad'], rk['overlap_paths_differ']['grad'])   # the same single exchange


def test_grad_reducer_launches_every_element_once():
    """partial overlaps are not exchanged twice, and ranges are merged across a gap only when the
    gap holds no parameter (a scalar eps between two ready matrices must not be swallowed)."""
    sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
    from l2hmc.dynamics.pytorch import training as T
    a, s1, b, c = (torch.nn.Parameter(torch.zeros(n)) for n in (10, 1, 10, 6))
    offs = {id(a): 0, id(s1): 10, id(b): 12, id(c): 24}        # s1 sits inside the 2-element alignment gap
    grp = {'flat': torch.zeros(32), 'grad': torch.zeros(32)}

    class Arena:
        groups = {torch.float32: grp}

        @staticmethod
        def _param_slices():
            for p in (a, s1, b, c):
                yield p, grp, offs[id(p)], p.numel()
    red = T.GradReducer(Arena(), force=True)
    T.GradReducer._paths_checked, T.GradReducer._paths_differ = True, False
    sent = []
    red._launch = lambda view: sent.append((view.storage_offset(), view.storage_offset() + view.numel()))
    red.ready([a, b])
    assert sent == [(0, 10), (12, 22)], sent                  # NOT (0, 22): that would take s1 along
    red.ready([s1, b, c])                  # b again (covered); s1 and c new, each with the zero padding next to it
    assert sorted(sent) == [(0, 10), (10, 12), (12, 22), (22, 30)], sent
    red.ready([a, s1, b, c])
    assert len(sent) == 4
    red.finish()                                              # the rest: alignment padding only
    cover = torch.zeros(32, dtype=torch.int32)
    for lo, hi in sent:
        cover[lo:hi] += 1
    assert int(cover.max()) == 1 and int(cover.min()) == 1, cover


def test_two_rank_train_step_equals_single_process(tmp_path):
    """2 ranks x 2 chains == 1 process x 4 chains: averaged flat gradient and the parameters
    after the fused Adam step (no BatchNorm in this fixture, so shard statistics don't enter)."""
    port = _free_port()
    mp.spawn(_train_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    old = torch.get_default_dtype()
    sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
    from l2hmc import native
    keep = native.call                       # (the in-process run swaps in the emulator: put it back)
    try:
        _train_worker(0, 1, port, str(tmp_path))
    finally:
        native.call = keep
        torch.set_default_dtype(old)
    r0, r1 = (torch.load(tmp_path / f't2_{i}.pt') for i in range(2))
    one = torch.load(tmp_path / 't1_0.pt')
    assert r0['scale'] == 0.5 and one['scale'] == 1.0
    assert torch.equal(r0['grads'], r1['grads'])
    assert torch.equal(r0['params'], r1['params'])                 # replicas stay in lock-step
    gn = float(one['grads'].abs().max())
    assert float((r0['grads'] - one['grads']).abs().max()) < 1e-12 * max(gn, 1.0)
    assert abs(0.5 * (r0['loss'] + r1['loss']) - one['loss']) < 1e-12 * max(abs(one['loss']), 1.0)
    assert float((r0['params'] - one['params']).abs().max()) < 1e-7


# ------------------------------------------------------------------ Experiment under 2 ranks
def _experiment_worker(rank, world, port, out):
    """`python -m l2hmc`-style construction on every rank: the model (parameters, buffers, masks)
    must come out identical, the chains / momenta / accept uniforms must not."""
    sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import numpy as np
    import emu_native
    from l2hmc import native
    import l2hmc.configs as cfgs
    from l2hmc.experiment.pytorch.experiment import Experiment
    from l2hmc.utils import dist as D
    native.call = emu_native.call
    import l2hmc._ops as ops
    ops.N.call = emu_native.call
    cfg = cfgs.get_config(['dynamics.group=U1', 'dynamics.latvolume=[4,4]', 'dynamics.nchains=6',
                           'dynamics.nleapfrog=2', 'conv=none', 'network.units=[8,8]',
                           'backend=gloo', f'port={port}', 'seed=4321'])
    # a rank-dependent perturbation BEFORE construction: whatever the local RNG state is, the
    # replicas must end up with rank 0's model
    torch.manual_seed(17 + rank)
    np.random.seed(17 + rank)
    ex = Experiment(cfg)
    dyn = ex.trainer.dynamics
    if rank == 1:                      # and an explicit divergence that sync_model has to repair
        with torch.no_grad():
            for p in dyn.parameters():
                p.add_(0.125)
        dyn.set_masks([1.0 - m.numpy().reshape(-1) for m in dyn.masks])
    D.sync_model(dyn)
    params = torch.cat([p.detach().reshape(-1).cpu().double() for p in dyn.parameters()])
    masks = torch.stack([m.reshape(-1) for m in dyn.masks])
    x = ex.lattice.random()
    v = torch.randn(4)
    xo, m = ex.trainer.eval_step((x, 2.0))
    torch.save({'params': params, 'masks': masks, 'x': x.cpu(), 'v': v, 'np': np.random.rand(3),
                'xo': xo.cpu(), 'acc': m['acc'].cpu()}, os.path.join(out, f'e{rank}.pt'))
    D.cleanup()


def test_experiment_two_ranks_same_model_different_chains(tmp_path):
    port = _free_port()
    mp.spawn(_experiment_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f'e{i}.pt', weights_only=False) for i in range(2))
    assert torch.equal(r0['params'], r1['params']) and r0['params'].numel() > 1000
    assert torch.equal(r0['masks'], r1['masks'])
    assert not torch.equal(r0['x'], r1['x'])                        # independent chains
    assert not torch.equal(r0['v'], r1['v']) and not (r0['np'] == r1['np']).all()
    assert not torch.equal(r0['xo'], r1['xo'])


# --------------------------------------------------------------------------- NativeComm bootstrap
def _fake_rccl(tmp):
    """tests/native_host/fake_rccl.c -> a shared library with RCCL's five entry points on HOST memory"""
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    so = os.path.join(tmp, 'libfake_rccl.so')
    subprocess.run(['gcc', '-O1', '-shared', '-fPIC', '-o', so,
                    os.path.join(ROOT, 'tests', 'native_host', 'fake_rccl.c')], check=True)
    return so


def _native_comm_worker(rank, world, port, out, so, mode):
    sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), L2Q_RCCL_LIB=so,
                      FAKE_RCCL_DIR=out, FAKE_RCCL_TIMEOUT_S='4')
    from l2hmc import native
    from l2hmc.utils import dist as D
    native.ptr = lambda t: t.data_ptr()             # host buffers: the fake library sums host memory
    native.stream_ptr = lambda: 0
    D.setup_torch_distributed('gloo', str(port))
    res = {}
    if mode == 'ok':
        with D.NativeComm() as c:                   # id on rank 0 -> broadcast over gloo -> init on both
            assert (c.rank, c.world_size) == (rank, world)
            g64 = torch.arange(5, dtype=torch.float64) * (rank + 1)
            g32 = torch.full((3,), float(rank + 1), dtype=torch.float32)
            c.all_reduce_(g64)
            c.all_reduce_(g32)
            with pytest.raises(TypeError):
                c.all_reduce_(torch.zeros(2, dtype=torch.int32))
            res = {'g64': g64, 'g32': g32}
        assert c._comm is None                      # released by the context manager
        # ParamArena-style use: the flat gradient of a model, averaged like DDP
        c2 = D.NativeComm()
        flat = torch.full((4,), float(10 * (rank + 1)), dtype=torch.float64)
        c2.all_reduce_(flat)
        res['mean'] = flat / world
        del c2                                      # __del__ destroys the communicator
    elif mode == 'mismatch':
        # every rank draws ITS OWN id (a bootstrap that forgot the broadcast): the rendezvous must fail
        # with an error on both ranks -- not hang
        import ctypes as C
        lib = native.load()
        ident = C.create_string_buffer(128)
        assert lib.l2q_comm_unique_id(ident) == 0
        import time
        time.sleep(0.01 * rank)
        comm = C.c_void_p()
        rc = lib.l2q_comm_init(ident, world, rank, C.byref(comm))
        res = {'rc': rc, 'err': lib.l2q_last_error().decode()}
    torch.save(res, os.path.join(out, f'n{rank}.pt'))
    D.cleanup()


def test_native_comm_bootstrap_two_ranks(tmp_path):
    """utils.dist.NativeComm with world_size 2 on the CPU container: the C-ABI route l2q_comm_unique_id
    -> (broadcast of the 128 bytes over the torch.distributed group) -> l2q_comm_init ->
    l2q_allreduce_grads -> l2q_comm_destroy, with librccl replaced by a host-memory fake
    (L2Q_RCCL_LIB).  The first real multi-GPU run then only has RCCL itself left to discover."""
    so = _fake_rccl(str(tmp_path))
    port = _free_port()
    mp.spawn(_native_comm_worker, args=(2, port, str(tmp_path), so, 'ok'), nprocs=2, join=True)
    r = [torch.load(tmp_path / f'n{i}.pt') for i in range(2)]
    for k in ('g64', 'g32', 'mean'):
        assert torch.equal(r[0][k], r[1][k]), k
    assert torch.equal(r[0]['g64'], torch.arange(5, dtype=torch.float64) * 3)
    assert torch.equal(r[0]['g32'], torch.full((3,), 3.0))
    assert torch.equal(r[0]['mean'], torch.full((4,), 15.0, dtype=torch.float64))


def test_native_comm_id_mismatch_fails_instead_of_hanging(tmp_path):
    so = _fake_rccl(str(tmp_path))
    port = _free_port()
    mp.spawn(_native_comm_worker, args=(2, port, str(tmp_path), so, 'mismatch'), nprocs=2, join=True)
    r = [torch.load(tmp_path / f'n{i}.pt') for i in range(2)]
    assert all(x['rc'] != 0 for x in r)
    assert all('ncclCommInitRank' in x['err'] for x in r), [x['err'] for x in r]


def test_rccl_resolution_failure_reports_its_reason_every_time(tmp_path):
    """after a failed resolution every entry point re-states why (no stale message)."""
    import subprocess
    code = (
        "import sys, ctypes as C; sys.path.insert(0, %r)\n"
        "from l2hmc import native\n"
        "lib = native.load(); buf = C.create_string_buffer(128)\n"
        "assert lib.l2q_comm_unique_id(buf) != 0\n"
        "e1 = lib.l2q_last_error().decode()\n"
        "assert lib.l2q_set_tuning(b'no_such_knob', 1) != 0\n"       # another error in between
        "assert lib.l2q_last_error().decode() != e1\n"
        "c = C.c_void_p(); assert lib.l2q_comm_init(buf, 1, 0, C.byref(c)) != 0\n"
        "e2 = lib.l2q_last_error().decode()\n"
        "assert e1 == e2 and 'libdoes_not_exist' in e1, (e1, e2)\n" % os.path.join(ROOT, 'l2hmc-qcd_amd'))
    env = dict(os.environ, L2Q_RCCL_LIB='libdoes_not_exist.so')
    subprocess.run([sys.executable, '-c', code], check=True, env=env)


def test_rccl_abi_version_is_checked(tmp_path):
    """the wrapper declares RCCL's types by hand, so the library it resolves must SAY it speaks that
    ABI -- ncclGetVersion with major 2 is accepted (l2q_comm_version reports the code), anything else refused
    with the reason; l2q_comm_abort tears a communicator down without a handshake."""
    import subprocess
    so = _fake_rccl(str(tmp_path))
    code = (
        "import sys, os, ctypes as C; sys.path.insert(0, %r)\n"
        "from l2hmc import native\n"
        "lib = native.load(); buf = C.create_string_buffer(128)\n"
        "want = os.environ['WANT']\n"
        "rc = lib.l2q_comm_unique_id(buf)\n"
        "if want == 'ok':\n"
        "    assert rc == 0 and lib.l2q_comm_version() == 22105\n"
        "    c = C.c_void_p(); assert lib.l2q_comm_init(buf, 1, 0, C.byref(c)) == 0\n"
        "    assert lib.l2q_comm_abort(c) == 0\n"
        "else:\n"
        "    assert rc != 0 and lib.l2q_comm_version() < 0\n"
        "    assert 'refuses' in lib.l2q_last_error().decode(), lib.l2q_last_error().decode()\n"
        % os.path.join(ROOT, 'l2hmc-qcd_amd'))
    for want, ver in (('ok', None), ('bad', '30100')):
        env = dict(os.environ, L2Q_RCCL_LIB=so, WANT=want)
        if ver:
            env['FAKE_RCCL_VERSION'] = ver
        subprocess.run([sys.executable, '-c', code], check=True, env=env)


def _ddp_worker(rank, world, port, out):
    """The reference's data-parallel wrapper as it uses it (trainers/pytorch/trainer.py:246-257, 1276-1304):
    `DistributedDataParallel(dynamics)`, forward through the wrapper, `loss.backward()` -- the gradient
    all-reduce is DDP's own, fired by the accumulation hooks of the parameters this build's Transition node
    hands its gradients to."""
    sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import numpy as np
    torch.set_default_dtype(torch.float64)
    import emu_native
    import helpers
    from l2hmc import native
    from l2hmc.utils import dist as D
    native.call = emu_native.call
    import l2hmc._ops as ops
    ops.N.call = emu_native.call
    if world > 1:
        assert D.setup_torch(seed=1234, backend='gloo') == rank
    g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'u1_train_f64_plain.npz')))
    nb = g['x'].shape[0] - 1
    lo, hi = (rank * nb // world, (rank + 1) * nb // world)
    gs = dict(g)
    for k in ('x', 'normals', 'u'):
        gs[k] = g[k][lo:hi]
    dyn, lat, loss_fn = helpers.build_u1_train_dynamics(gs)
    model = dyn
    if world > 1:
        from torch.nn.parallel import DistributedDataParallel as DDP
        model = DDP(dyn, find_unused_parameters=True)
    dyn._inject = {'normals': gs['normals'], 'u': gs['u']}
    x = dyn.g.compat_proj(dyn.unflatten(torch.from_numpy(gs['x'])))
    x.requires_grad_(True)
    xout, m = model((x, torch.tensor(float(g['beta']))))
    loss = loss_fn(x, m['mc_states'].proposed.x, m['acc'])
    loss.backward()
    grads = torch.cat([p.grad.reshape(-1) for p in dyn.parameters()])
    torch.save({'grads': grads, 'loss': float(loss)}, os.path.join(out, f'd{world}_{rank}.pt'))
    if world > 1:
        D.cleanup()


def test_reference_ddp_wrapper_drives_dynamics(tmp_path):
    port = _free_port()
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    old = torch.get_default_dtype()
    sys.path.insert(0, os.path.join(ROOT, 'l2hmc-qcd_amd'))
    from l2hmc import native
    import l2hmc._ops as ops
    keep = (native.call, ops.N.call)
    try:
        _ddp_worker(0, 1, port, str(tmp_path))
    finally:
        native.call, ops.N.call = keep
        torch.set_default_dtype(old)
    r0, r1 = (torch.load(tmp_path / f'd2_{i}.pt') for i in range(2))
    one = torch.load(tmp_path / 'd1_0.pt')
    assert torch.equal(r0['grads'], r1['grads'])                   # DDP averaged them
    gn = float(one['grads'].abs().max())
    assert gn > 0 and float((r0['grads'] - one['grads']).abs().max()) < 1e-12 * max(gn, 1.0)
