"""CPU tests of the training-gradient path's HOST side (tape, reverse sweep, loss seeds,
network backward sequencing, flat arena, Adam bookkeeping) against the reference's gradient
fixtures, with the libl2q.so entry points replaced by the torch restatement in
tests/emu_native.py (test infrastructure; the product has no CPU path).  The kernels
themselves are checked against the same restatement on the GPU (test_train_gpu.py)."""
import numpy as np
import pytest
import torch

import emu_native
import helpers


@pytest.fixture(autouse=True)
def _f32_default():
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float32)
    yield
    torch.set_default_dtype(old)


@pytest.fixture
def f64():
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(old)


@pytest.mark.skipif(torch.cuda.is_available(), reason='host-logic test for the CPU container')
@pytest.mark.parametrize('autograd', [False, True], ids=['trainer', 'autograd'])
@pytest.mark.parametrize('name', ['u1_train_f64', 'u1_train_f64_plain', 'u1_train_nomerge_fwd',
                                  'u1_train_nomerge_bwd'])
def test_train_step_host_logic_f64(name, autograd, golden, monkeypatch, f64):
    g = golden(name)
    emu_native.install(monkeypatch)
    dyn, lat, loss_fn = helpers.build_u1_train_dynamics(g)
    out = helpers.check_train_step(g, dyn, loss_fn, rtol=1e-9, atol_rel=1e-6, autograd=autograd)
    assert out['grad_rel'] < 1e-7, out
    assert out['param_abs'] < 1e-7, out      # Adam's g / (|g| + 1e-8) amplifies rounding of tiny g


@pytest.mark.skipif(torch.cuda.is_available(), reason='host-logic test for the CPU container')
@pytest.mark.parametrize('autograd', [False, True], ids=['trainer', 'autograd'])
def test_train_step_host_logic_conv_f32(autograd, golden, monkeypatch):
    g = golden('u1_train_conv')
    emu_native.install(monkeypatch)
    dyn, lat, loss_fn = helpers.build_u1_train_dynamics(g)
    out = helpers.check_train_step(g, dyn, loss_fn, rtol=2e-4, atol_rel=1e-3,
                                   adam_min_grad=1e-3, autograd=autograd)
    assert out['grad_rel'] < 2e-2, out
    assert out['param_abs'] < 2e-5, out


@pytest.mark.skipif(torch.cuda.is_available(), reason='host-logic test for the CPU container')
@pytest.mark.parametrize('conv', [False, True])
def test_trainer_train_loop_host_logic(conv, monkeypatch):
    """Trainer.train_step bookkeeping with dropout + BatchNorm (+ conv stack): arena views,
    Adam step counter, cache invalidation after the fused update, eval after training."""
    import l2hmc.configs as cfgs
    from l2hmc.trainers.pytorch.trainer import Trainer
    emu_native.install(monkeypatch)
    old = torch.get_default_dtype()
    try:
        torch.manual_seed(0)
        np.random.seed(0)
        cfg = cfgs.get_config(['dynamics.group=U1', 'dynamics.latvolume=[4,4]',
                               'dynamics.nchains=8', 'dynamics.nleapfrog=2',
                               'dynamics.verbose=false', 'network.units=[8,8]',
                               'network.dropout_prob=0.2', 'network.use_batch_norm=true',
                               'learning_rate.clip_norm=1.0']
                              + ([] if conv else ['conv=none']))
        tr = Trainer(cfg)
        x = tr.warmup(beta=2.0, nsteps=3)
        before = {k: v.detach().clone() for k, v in tr.dynamics.named_parameters()}
        eps0 = tr.dynamics._eps('x', 0)
        out = tr.train(x=x, beta=2.0, nsteps=3)
        assert all(np.isfinite(l) for l in out['history']['loss'])
        moved = sum(int(not torch.equal(p.detach(), before[k]))
                    for k, p in tr.dynamics.named_parameters())
        assert moved >= 0.9 * len(before), (moved, len(before))
        assert tr.arena.step_count == 3
        assert tr.dynamics._eps('x', 0) != eps0          # step-size cache saw the fused update
        # every parameter is a view of the flat arena, every grad a view of the flat gradient
        for grp in tr.arena.groups.values():
            lo, hi = grp['flat'].data_ptr(), grp['flat'].data_ptr() + grp['flat'].numel() * 8
            for p in grp['params']:
                assert lo <= p.data_ptr() < hi
        _, m = tr.eval_step((out['x'], 2.0))
        assert torch.isfinite(m['acc']).all()
    finally:
        torch.set_default_dtype(old)


@pytest.mark.skipif(torch.cuda.is_available(), reason='host-logic test for the CPU container')
@pytest.mark.parametrize('autograd', [False, True], ids=['trainer', 'autograd'])
@pytest.mark.parametrize('name', ['su3_train', 'su3_train_c1', 'su3_train_nomerge'])
def test_su3_train_step_host_logic(name, autograd, golden, monkeypatch, f64):
    """SU(3) tape / reverse sweep / loss seeds against the reference's autograd gradients
    (su3_train_c1: improved action, the rectangle term enters the accept probability)."""
    g = golden(name)
    emu_native.install(monkeypatch)
    dyn, lat, loss_fn = helpers.build_su3_train_dynamics(g)
    out = helpers.check_train_step(g, dyn, loss_fn, rtol=1e-7, atol_rel=1e-6, adam_min_grad=1e-6,
                                   autograd=autograd)
    assert out['grad_rel'] < 1e-5, out
    assert out['param_abs'] < 1e-6, out


@pytest.mark.skipif(torch.cuda.is_available(), reason='host-logic test for the CPU container')
def test_experiment_train_eras_host_logic(monkeypatch):
    """Experiment.train: eras x epochs with the linearly annealed beta."""
    import l2hmc.configs as cfgs
    from l2hmc.experiment.pytorch.experiment import Experiment
    emu_native.install(monkeypatch)
    cfg = cfgs.get_config(['dynamics.group=U1', 'dynamics.latvolume=[4,4]', 'dynamics.nchains=4',
                           'dynamics.nleapfrog=2', 'dynamics.verbose=false', 'network.units=[4]',
                           'conv=none', 'steps.nera=3', 'steps.nepoch=2',
                           'annealing_schedule.beta_init=1.0', 'annealing_schedule.beta_final=3.0'])
    ex = Experiment(cfg)
    out = ex.train()
    assert out['history']['beta'] == [1.0, 1.0, 2.0, 2.0, 3.0, 3.0]
    assert out['history']['era'] == [0, 0, 1, 1, 2, 2]
    assert ex.trainer.arena.step_count == 6


@pytest.mark.skipif(torch.cuda.is_available(), reason='host-logic test for the CPU container')
def test_checkpoint_roundtrip_and_torch_adam_layout(monkeypatch, tmp_path):
    """save_ckpt / load_ckpt: parameters, Adam moments and step survive; the optimizer_state_dict
    loads into a real torch.optim.Adam over the same parameters (the reference's format), and a
    state_dict produced by torch.optim.Adam loads into the arena."""
    import l2hmc.configs as cfgs
    from l2hmc.trainers.pytorch.trainer import Trainer
    emu_native.install(monkeypatch)
    ov = ['dynamics.group=U1', 'dynamics.latvolume=[4,4]', 'dynamics.nchains=4',
          'dynamics.nleapfrog=2', 'dynamics.verbose=false', 'network.units=[4]', 'conv=none']
    torch.manual_seed(1); np.random.seed(1)
    tr = Trainer(cfgs.get_config(ov))
    out = tr.train(beta=2.0, nsteps=3)
    f = tr.save_ckpt(era=0, epoch=3, outdir=tmp_path)
    ck = torch.load(f, weights_only=False)
    assert set(ck) >= {'era', 'epoch', 'gstep', 'xeps', 'veps', 'model_state_dict', 'optimizer_state_dict'}
    assert ck['gstep'] == 3 and len(ck['xeps']) == 2
    # (a) our optimizer_state_dict is a valid torch.optim.Adam state
    ref_params = [torch.nn.Parameter(p.detach().clone()) for p in tr.dynamics.parameters()]
    opt = torch.optim.Adam(ref_params, lr=1e-3)
    opt.load_state_dict(ck['optimizer_state_dict'])
    st = opt.state_dict()['state']
    assert len(st) == len(ref_params) and float(st[0]['step']) == 3.0
    # (b) round trip into a fresh trainer built from another seed
    torch.manual_seed(2); np.random.seed(2)
    tr2 = Trainer(cfgs.get_config(ov))
    tr2.dynamics.set_masks([m.numpy() for m in tr.dynamics.masks])
    tr2.load_ckpt(f)
    for (k, a), (_, b) in zip(tr.dynamics.state_dict().items(), tr2.dynamics.state_dict().items()):
        assert torch.equal(a, b), k
    g1, g2 = tr.arena.groups[torch.float32], tr2.arena.groups[torch.float32]
    assert torch.equal(g1['m'], g2['m']) and torch.equal(g1['v'], g2['v'])
    assert tr2.arena.step_count == 3 and tr2._gstep == 3
    # identical continuation: one more step from the same state and the same draws
    x = out['x']
    for t in (tr, tr2):
        torch.manual_seed(5)
        t.dynamics.rng_device = 'cpu'
        t.train_step((x, 2.0))
    assert torch.equal(g1['flat'], g2['flat'])
    # (c) a state produced by torch.optim.Adam loads into the arena
    for p in ref_params:
        p.grad = torch.ones_like(p)
    opt.step()
    tr2.arena.load_state_dict(opt.state_dict(), list(tr2.dynamics.parameters()))
    assert tr2.arena.step_count == 4


@pytest.mark.skipif(torch.cuda.is_available(), reason='host-logic test for the CPU container')
@pytest.mark.parametrize('prec', ['fp16', 'bf16'])
def test_half_precision_eval_host_logic(prec, monkeypatch):
    """precision=fp16|bf16 (BASELINE cfg-3 "fp16 nets / fp32 action"): the Trainer switches
    every LeapfrogLayer to the 16-bit layers for sampling, the fused fp32 sub-update kernels
    are bypassed, results stay within half-precision distance of the fp32 trajectory, and
    train_step keeps differentiating the fp32 master weights."""
    import l2hmc.configs as cfgs
    from l2hmc.trainers.pytorch.trainer import Trainer
    emu_native.install(monkeypatch)
    torch.manual_seed(0)
    np.random.seed(0)
    ov = ['dynamics.group=U1', 'dynamics.latvolume=[4,4]', 'dynamics.nchains=8',
          'dynamics.nleapfrog=2', 'dynamics.verbose=false', 'network.units=[8,8]',
          'network.dropout_prob=0.0', 'network.use_batch_norm=false', 'conv=none']
    tr = Trainer(cfgs.get_config(ov + [f'precision={prec}']))
    hd = torch.float16 if prec == 'fp16' else torch.bfloat16
    assert tr.dynamics.net_precision == hd
    assert all(m.half_dtype == hd for m in tr.dynamics.networks.modules() if hasattr(m, 'set_precision'))
    assert tr.dynamics._fused_u1(tr.dynamics._get_vnet(0)) is None
    x = tr.lattice.random()
    nrm = torch.randn(8, 32)
    res = {}
    for p in (prec, None):
        tr.dynamics.set_net_precision(p)
        tr.dynamics._inject = {'normals': nrm.numpy(), 'u': np.full(8, 0.5, dtype=np.float32)}
        xo, m = tr.eval_step((x, 2.0))
        res[p] = (xo, m['acc'])
    ulp = 2.0 ** -10 if prec == 'fp16' else 2.0 ** -7
    assert 0 < float((res[prec][1] - res[None][1]).abs().max()) < 200 * ulp
    tr.dynamics.set_net_precision(prec)
    tr.dynamics._inject = None
    _, m = tr.train_step((x, 2.0))
    assert np.isfinite(float(m['loss']))
    with pytest.raises(ValueError):
        Trainer(cfgs.get_config(['dynamics.group=SU3', 'dynamics.latvolume=[2,2,2,2]',
                                 'dynamics.nchains=2', 'dynamics.nleapfrog=1',
                                 'network.units=[4]', f'precision={prec}']))


@pytest.mark.skipif(torch.cuda.is_available(), reason='host-logic test for the CPU container')
def test_trainer_small_surface_host_logic(monkeypatch):
    """count_parameters / draw_x / draw_v / calc_loss / reset_optimizer / train_epoch /
    Experiment.set_net_weights behave like their reference namesakes."""
    import l2hmc.configs as cfgs
    from l2hmc.dynamics.pytorch import dynamics as D
    from l2hmc.experiment.pytorch.experiment import Experiment
    emu_native.install(monkeypatch)
    torch.manual_seed(0)
    np.random.seed(0)
    cfg = cfgs.get_config(['dynamics.group=U1', 'dynamics.latvolume=[4,4]', 'dynamics.nchains=6',
                           'dynamics.nleapfrog=2', 'dynamics.verbose=false', 'network.units=[8]',
                           'network.dropout_prob=0.0', 'network.use_batch_norm=false', 'conv=none',
                           'steps.nepoch=2'])
    ex = Experiment(cfg)
    tr = ex.trainer
    assert tr.count_parameters() == sum(p.numel() for p in tr.dynamics.parameters())
    x, v = tr.draw_x(), tr.draw_v()
    assert x.shape == (6, 32) and float(x.abs().max()) <= np.pi + 1e-6 and v.numel() == 6 * 32
    assert tr.get_lr(0) == tr.config.learning_rate.lr_init
    assert isinstance(tr.metric_to_numpy([torch.ones(2), torch.zeros(2)]), np.ndarray)
    xo, hist = tr.train_epoch(tr.lattice.random(), 2.0, nepoch=2, warmup=False)
    assert len(hist['loss']) == 2 and tr.arena.step_count == 2
    tr.reset_optimizer()
    assert tr.arena.step_count == 0
    assert all(float(g['m'].abs().max()) == 0.0 for g in tr.arena.groups.values())
    ex.set_net_weights(cfgs.NetWeights(x=cfgs.NetWeight(0., 0., 0.), v=cfgs.NetWeight(0., 0., 0.)))
    vnet = tr.dynamics._get_vnet(0)
    assert (vnet.nw.s, vnet.nw.t, vnet.nw.q) == (0., 0., 0.)
    tr.dynamics._inject = None
    _, m = tr.eval_step((xo, 2.0))            # all heads scaled to zero: generic HMC
    assert float(m['sumlogdet'].abs().max()) < 1e-4
    with pytest.raises(NotImplementedError):
        ex.visualize_model()
    a = D.to_u1(torch.tensor([4.0, -4.0]))
    assert float(a.abs().max()) <= np.pi
    assert D.random_angle((3, 2)).shape == (3, 2)
    mk = D.Mask(torch.tensor([1., 0.]))
    assert torch.equal(mk.combine(torch.tensor([2., 2.]), torch.tensor([5., 5.])), torch.tensor([2., 5.]))


@pytest.mark.skipif(torch.cuda.is_available(), reason='host-logic test for the CPU container')
def test_su3_micro_batched_training_host_logic(golden, monkeypatch, f64):
    """Trainer.micro_batch: the gradient of a train step taken in chain micro-batches equals the
    full-batch gradient (independent chains, mean loss); the never-called SU(3) xnet stays
    outside the gradient / moment arena."""
    import l2hmc.configs as cfgs
    from l2hmc.trainers.pytorch.trainer import Trainer
    emu_native.install(monkeypatch)
    ov = ['dynamics.group=SU3', 'dynamics.latvolume=[2,2,2,2]', 'dynamics.nchains=4',
          'dynamics.nleapfrog=1', 'dynamics.eps=0.02', 'dynamics.verbose=false',
          'dynamics.use_split_xnets=false', 'dynamics.use_separate_networks=false',
          'network.units=[4]', 'network.dropout_prob=0.0', 'network.use_batch_norm=false',
          'network.activation_fn=tanh', 'loss.aux_weight=0.0', 'learning_rate.clip_norm=0.0',
          'conv=none']
    grads = {}
    for mb in (None, 2, 3):
        torch.manual_seed(1)
        np.random.seed(1)
        tr = Trainer(cfgs.get_config(ov))
        tr.micro_batch = mb
        x = tr.lattice.random()
        V = 16
        nrm = torch.randn(8, 4, 4, 2, 2, 2, 2, generator=torch.Generator().manual_seed(7))
        tr.dynamics._inject = {'normals': nrm.numpy(), 'u': np.full(4, 0.5)}
        xo, m = tr.train_step((x, 6.0))
        assert xo.shape[0] == 4 and m['acc'].shape == (4,)
        grads[mb] = {k: p.grad.detach().clone() for k, p in tr.dynamics.named_parameters()
                     if p.grad is not None}
        # arena: vnet + eps only
        n_arena = tr.arena.numel()
        n_x = sum(p.numel() for p in tr.dynamics.xnet.parameters())
        assert n_arena + n_x == tr.count_parameters()
        assert all(p.grad is None for p in tr.dynamics.xnet.parameters())
        sd = tr.arena.state_dict(list(tr.dynamics.parameters()), lr=1e-3)
        assert len(sd['state']) == len(grads[mb])
    for mb in (2, 3):
        for k, g in grads[None].items():
            d = float((grads[mb][k] - g).abs().max())
            # 1e-7: the gradient passes through the backward of projectSU(force), whose conditioning
            # (~1e7, see su3_train golden) amplifies the rounding differences of a changed summation
            # order (micro-batch GEMM shapes, native-order weight shadows)
            assert d <= 1e-7 * max(1.0, float(g.abs().max())), (mb, k, d)


@pytest.mark.skipif(torch.cuda.is_available(), reason='host-logic test for the CPU container')
@pytest.mark.parametrize('name', ['u1_bf16', 'u1_bf16_tanh', 'u1_fp16', 'u1_fp16_tanh'])
def test_bf16_rounding_points_vs_reference_autocast(name, golden, monkeypatch):
    """The emulator's restatement of where the 16-bit layers round (tests/emu_native.py,
    l2q_gemm_h) against the REAL reference run under torch.autocast('cpu', bfloat16)
    (tests/golden/make_golden_bf16.py): pins the emulator that the GPU half-precision kernel
    tests compare with, and the product's host logic for precision='bf16'."""
    g = golden(name)
    emu_native.install(monkeypatch)
    dyn, lat = helpers.build_u1_dynamics(g)
    hd = 'fp16' if 'fp16' in name else 'bf16'
    dyn.set_net_precision(hd)
    dyn.fuse_half_heads = False           # gemm_h + fp32 update kernels (the emulated set)
    x = torch.from_numpy(g['x'])
    beta = torch.tensor(float(g['beta']))
    nb = x.shape[0]
    f = dyn.grad_potential(x, beta)
    ulp = 2.0 ** -7 if hd == 'bf16' else 2.0 ** -10
    for a, k in zip(dyn._call_vnet(0, (x, f)), ('vnet_s', 'vnet_t', 'vnet_q')):
        ref = g[k]
        d = np.abs(a.numpy() - ref)
        assert d.max() <= 2.5 * ulp * max(1.0, np.abs(ref).max()), (k, d.max())
        assert (d == 0).mean() > 0.3, (k, (d == 0).mean())
    dyn._inject = {'normals': g['normals'], 'u': g['u']}
    xo, m = dyn((x, beta))
    assert np.abs(m['acc'].numpy() - g['acc']).max() < max(3 * np.abs(g['acc'] - g['acc_fp32']).max(), 5e-3)
    assert np.array_equal(m['acc_mask'].numpy(), g['acc_mask'])


def test_dynamics_rejects_foreign_potential_and_unsupported_training(monkeypatch):
    """a potential_fn the HIP trajectory cannot honour is refused at construction;
    training configurations that are not differentiated raise instead of training something else."""
    import l2hmc.configs as cfgs
    from l2hmc.dynamics.pytorch.dynamics import Dynamics
    from l2hmc.lattice.u1.pytorch.lattice import LatticeU1
    from l2hmc.trainers.pytorch.trainer import Trainer
    dc = cfgs.DynamicsConfig(nchains=2, group='U1', latvolume=[4, 4], nleapfrog=1)
    lat = LatticeU1(2, [4, 4])
    Dynamics(lat.action, dc, None)                                     # fine
    Dynamics(lat.potential_energy, dc, None)
    with pytest.raises(ValueError):
        Dynamics(lambda x, b: lat.action(x, b), dc, None)
    with pytest.raises(ValueError):
        Dynamics(LatticeU1(2, [4, 8]).action, dc, None)                # other lattice shape
    emu_native.install(monkeypatch)
    base = ['dynamics.group=U1', 'dynamics.latvolume=[4,4]', 'dynamics.nchains=2',
            'dynamics.nleapfrog=2', 'conv=none', 'network.units=[4]']
    tr = Trainer(cfgs.get_config(base + ['dynamics.merge_directions=false']))
    x = tr.lattice.random()
    tr.eval_step((x, 2.0))                                             # sampling works
    tr.train_step((x, 2.0))                 # ... and so does training (round 4: single-direction tape)
    tr2 = Trainer(cfgs.get_config(base))
    tr2.config.gradient_accumulation_steps = 2
    with pytest.raises(NotImplementedError):
        tr2.train_step((x, 2.0))


@pytest.mark.skipif(torch.cuda.is_available(), reason='host-logic test for the CPU container')
@pytest.mark.parametrize('act', ['swish', 'tanh', 'leaky_relu'])
def test_leapfrog_layer_backward_host_logic(act, monkeypatch):
    """swish keeps the pre-activations on the tape (VERDICT r01 missing item 6)."""
    emu_native.install(monkeypatch)
    helpers.check_leapfrog_layer_backward(act, 'cpu', torch.float64, 1e-10)


@pytest.mark.skipif(torch.cuda.is_available(), reason='host-logic test for the CPU container')
def test_trainer_train_step_single_direction(monkeypatch):
    """Trainer.train_step with dynamics.merge_directions=False (VERDICT r03 missing #4): trains on
    the transition eval_step samples with -- the direction is drawn from the global generator before
    the momenta, like Dynamics.apply_transition (dynamics.py:709)."""
    import l2hmc.configs as cfgs
    from l2hmc.trainers.pytorch.trainer import Trainer
    from l2hmc.dynamics.pytorch import training as T
    emu_native.install(monkeypatch)
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float32)
    try:
        torch.manual_seed(5)
        np.random.seed(5)
        cfg = cfgs.get_config(['dynamics.group=U1', 'dynamics.latvolume=[4,4]', 'dynamics.nchains=6',
                               'dynamics.nleapfrog=2', 'dynamics.merge_directions=false',
                               'dynamics.verbose=true', 'network.units=[6]', 'conv=none',
                               'network.dropout_prob=0.0', 'network.use_batch_norm=false'])
        tr = Trainer(cfg)
        x = tr.lattice.random()
        seen = []
        real = T.trajectory_train

        def spy(dyn, xn, vn, beta, forward):
            seen.append(forward)
            return real(dyn, xn, vn, beta, forward)
        monkeypatch.setattr(T, 'trajectory_train', spy)
        w0 = tr.dynamics.vnet['0'].transl.weight.detach().clone()
        for i in range(6):
            torch.manual_seed(100 + i)
            want = bool(torch.rand(1) > 0.5)
            torch.manual_seed(100 + i)
            x, m = tr.train_step((x, 2.0))
            assert seen[-1] == want and np.isfinite(m['loss'])
            assert m['energy'].shape == (3, 6)                      # nleapfrog + 1 entries, not 2 nlf + 1
        assert len(set(seen)) == 2                                  # both directions occurred
        assert not torch.equal(tr.dynamics.vnet['0'].transl.weight.detach(), w0)
    finally:
        torch.set_default_dtype(old)


@pytest.mark.skipif(torch.cuda.is_available(), reason='host-logic test for the CPU container')
@pytest.mark.parametrize('name', ['u1_train_f64', 'su3_train', 'su3_train_nomerge'])
def test_autograd_bridge_semantics_host_logic(name, golden, monkeypatch, f64):
    g = golden(name)
    emu_native.install(monkeypatch)
    build = helpers.build_su3_train_dynamics if name.startswith('su3') else helpers.build_u1_train_dynamics
    # (the emulator restates the training kernels; the fused SU(3) sampler kernels run on the GPU tier)
    helpers.check_autograd_bridge_semantics(g, build, tol=1e-7 if name.startswith('su3') else 1e-9,
                                            sampler=not name.startswith('su3'))


@pytest.mark.skipif(torch.cuda.is_available(), reason='host-logic test for the CPU container')
@pytest.mark.parametrize('inject_dir', [True, False, None])
def test_micro_batched_single_direction_training_host_logic(inject_dir, monkeypatch, f64):
    """merge_directions=False in chain micro-batches: ONE direction per optimiser step (injected, or one
    host-generator draw), so the gradient equals the unchunked step's and the generator advances exactly
    as much (every micro-batch used to draw its own direction)."""
    import l2hmc.configs as cfgs
    from l2hmc.trainers.pytorch.trainer import Trainer
    emu_native.install(monkeypatch)
    ov = ['dynamics.group=SU3', 'dynamics.latvolume=[2,2,2,2]', 'dynamics.nchains=4',
          'dynamics.nleapfrog=2', 'dynamics.eps=0.02', 'dynamics.verbose=false',
          'dynamics.merge_directions=false',
          'dynamics.use_split_xnets=false', 'dynamics.use_separate_networks=false',
          'network.units=[4]', 'network.dropout_prob=0.0', 'network.use_batch_norm=false',
          'network.activation_fn=tanh', 'loss.aux_weight=0.0', 'learning_rate.clip_norm=0.0',
          'conv=none']
    grads, after = {}, {}
    for mb in (None, 2, 3):
        torch.manual_seed(1)
        np.random.seed(1)
        tr = Trainer(cfgs.get_config(ov))
        tr.micro_batch = mb
        x = tr.lattice.random()
        nrm = torch.randn(8, 4, 4, 2, 2, 2, 2, generator=torch.Generator().manual_seed(7))
        tr.dynamics._inject = {'normals': nrm.numpy(), 'u': np.full(4, 0.5)}
        if inject_dir is not None:
            tr.dynamics._inject['forward'] = inject_dir
        torch.manual_seed(11)                         # the direction draw (when not injected) comes from here
        xo, m = tr.train_step((x, 6.0))
        after[mb] = float(torch.rand(1))              # where the host generator stands after the step
        grads[mb] = {k: p.grad.detach().clone() for k, p in tr.dynamics.named_parameters()
                     if p.grad is not None}
    for mb in (2, 3):
        assert after[mb] == after[None], (mb, after)
        for k, g in grads[None].items():
            d = float((grads[mb][k] - g).abs().max())
            assert d <= 1e-7 * max(1.0, float(g.abs().max())), (mb, k, d)


@pytest.mark.skipif(torch.cuda.is_available(), reason='host-logic test for the CPU container')
@pytest.mark.parametrize('route', ['trainer', 'autograd'])
@pytest.mark.parametrize('name', ['u1_train_fp16', 'u1_train_fp16_bn', 'u1_train_bf16', 'u1_train_fp16_conv'])
def test_half_precision_train_step_host_logic(name, route, golden, monkeypatch):
    """autocast + GradScaler training (trainers/pytorch/trainer.py:211-219, 1276-1280, 1303-1313) against the
    real reference run that way: accept masks bit-equal, gradients within a multiple of the reference's OWN
    16-bit-vs-fp32 distance (the emulator restates the 16-bit layers' rounding points)."""
    g = golden(name)
    emu_native.install(monkeypatch)
    out = helpers.check_half_train_step(g, route)
    print(name, route, out)
    helpers.assert_half_train_step(g, name, route, out)


@pytest.mark.skipif(torch.cuda.is_available(), reason='host-logic test for the CPU container')
def test_trainer_detailed_steps_and_train_dynamic_host_logic(monkeypatch):
    """Trainer.train_step_detailed / eval_step_detailed / train_dynamic (trainers/pytorch/trainer.py:1369-1476,
    1840-1927): records with dt / loss / averages; beta follows the loss by tenths of itself."""
    import l2hmc.configs as cfgs
    from l2hmc.trainers.pytorch.trainer import Trainer
    emu_native.install(monkeypatch)
    torch.manual_seed(0)
    np.random.seed(0)
    cfg = cfgs.get_config(['dynamics.group=U1', 'dynamics.latvolume=[4,4]', 'dynamics.nchains=4',
                           'dynamics.nleapfrog=2', 'dynamics.verbose=true', 'network.units=[4]',
                           'network.dropout_prob=0.0', 'network.use_batch_norm=false', 'conv=none',
                           'annealing_schedule.beta_init=1.0', 'annealing_schedule.beta_final=1.5',
                           'annealing_schedule.dynamic=true', 'steps.nera=3', 'steps.nepoch=3', 'steps.log=1'])
    tr = Trainer(cfg)
    x, rec = tr.train_step_detailed(era=1, epoch=2)
    assert rec['era'] == 1 and rec['epoch'] == 2 and rec['tstep'] == 1 and rec['dt'] > 0
    assert np.isfinite(rec['loss']) and 'avgs' in rec and 'dQint' in rec
    x, rec = tr.eval_step_detailed('eval', x=x, beta=1.0)
    assert rec['dt'] > 0 and np.isfinite(rec['loss'])
    x, rec = tr.eval_step_detailed('hmc', x=x, beta=1.0)
    assert np.isfinite(rec['loss'])
    with pytest.raises(ValueError):
        tr.eval_step_detailed('train')
    out = tr.train_dynamic(x=x)
    assert 1 <= len(out['betas']) <= 3 and out['betas'][0] == 1.0
    for b0, b1 in zip(out['betas'], out['betas'][1:]):
        assert abs(abs(b1 - b0) - b0 / 10.0) < 1e-12
    assert len(out['history']['loss']) == 3 * len(out['betas'])
