"""RCCL on the one GPU of the test box: the collective of the path -- the sum
of the flat training-gradient buffer, reference trainers/pytorch/trainer.py:246-257 (DDP wrap) on
the process group of utils/dist.py:126-144 -- executed with a single rank, through both routes the
product has: the C ABI (l2q_comm_init / l2q_allreduce_grads, librccl resolved at run time) and
torch.distributed's 'nccl' backend (= RCCL) on the real ParamArena buffers.  With one rank the sum
is the identity, so what is proven is that librccl loads, that the collective is ordered after
the kernels enqueued on the launch stream before it and before those after it, and that one flat
buffer of the cfg-4 (1.45 GB) and cfg-5 (23 GB) gradient sizes goes through in a single call."""
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.fixture()
def comm():
    from l2hmc.utils.dist import NativeComm
    c = NativeComm(rank=0, world_size=1)
    yield c
    c.close()


@pytest.mark.parametrize('n,dtype', [(1 << 10, torch.float32), (181_403_648, torch.float64),
                                     (2_902_458_368, torch.float64)])
def test_native_allreduce_single_rank(comm, n, dtype):
    """n = the vnet parameter count at cfg-4 (8^4, units [256]) and at cfg-5 (16^4)."""
    from l2hmc import _ops as ops
    need = n * torch.empty((), dtype=dtype).element_size()
    free, total = torch.cuda.mem_get_info()
    if need * 1.3 > free:
        pytest.skip(f'needs {need / 2**30:.1f} GiB of HBM')
    buf = torch.empty(n, dtype=dtype, device='cuda')
    buf.copy_(torch.arange(n, device='cuda') % 1021)                   # enqueued before
    if dtype == torch.float64:
        ops.scale(buf, 0.5, out=buf)            # a library kernel on the launch stream, through raw pointers
    else:
        buf.mul_(0.5)
    comm.all_reduce_(buf)                                              # the collective
    buf.mul_(2.0)                                                      # enqueued after
    want = (torch.arange(n, device='cuda') % 1021).to(dtype)
    assert torch.equal(buf, want)
    del want
    # timing record (printed with -s): algorithmic bandwidth of the single-rank copy path
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(3):
        comm.all_reduce_(buf)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 3
    print(f'l2q_allreduce_grads world=1 {need / 1e9:.2f} GB: {ms:.3f} ms')


def test_native_comm_rejects_bad_arguments(comm):
    from l2hmc import native
    lib = native.load()
    buf = torch.zeros(16, device='cuda')
    assert lib.l2q_allreduce_grads(None, buf.data_ptr(), 16, 4, None) == -1
    assert lib.l2q_allreduce_grads(comm._comm, buf.data_ptr(), 16, 2, None) == -1      # fp16: refused
    assert lib.l2q_allreduce_grads(comm._comm, buf.data_ptr(), 0, 4, None) == -1
    assert lib.l2q_init(torch.cuda.current_device()) == torch.cuda.current_device()
    assert lib.l2q_init(99) == -1
    with pytest.raises(TypeError):
        comm.all_reduce_(torch.zeros(4, dtype=torch.float16, device='cuda'))


def test_torch_nccl_process_group_single_rank_param_arena():
    """init_process_group('nccl', world_size=1) and ParamArena.all_reduce(force=True) on the flat
    gradient buffers of a real (small) training step; the native route gives the same bits."""
    import torch.distributed as dist
    import l2hmc.configs as cfgs
    from l2hmc.trainers.pytorch.trainer import Trainer
    from l2hmc.utils.dist import NativeComm
    assert not dist.is_initialized()
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{_free_port()}', rank=0,
                            world_size=1)
    try:
        assert dist.get_backend() == 'nccl'
        tr = Trainer(cfgs.get_config([
            'dynamics.group=U1', 'dynamics.latvolume=[8,8]', 'dynamics.nchains=16',
            'dynamics.nleapfrog=2', 'conv=none', 'network.units=[16,16]']))
        x = tr.lattice.random()
        tr.train_step((x, 2.0))
        for g in tr.arena.groups.values():
            g['grad'].normal_()
        before = {k: g['grad'].clone() for k, g in tr.arena.groups.items()}
        assert tr.arena.all_reduce() == 1.0                    # single rank: no collective by default
        assert tr.arena.all_reduce(force=True) == 1.0          # the collective, through RCCL
        for k, g in tr.arena.groups.items():
            assert torch.equal(g['grad'], before[k])
        c = NativeComm()                                       # picks rank / world from the group
        assert (c.rank, c.world_size) == (0, 1)
        assert tr.arena.all_reduce(comm=c, force=True) == 1.0
        for k, g in tr.arena.groups.items():
            assert torch.equal(g['grad'], before[k])
        c.close()
        # a cfg-4-sized flat fp64 buffer through torch's group (stream-ordered with a producer)
        n = 181_403_648
        buf = torch.full((n,), 3.0, dtype=torch.float64, device='cuda')
        buf.mul_(2.0)
        dist.all_reduce(buf)
        buf.add_(1.0)
        assert float(buf.min()) == 7.0 and float(buf.max()) == 7.0
    finally:
        dist.destroy_process_group()
