"""GPU tests of the rows SURVEY 8(f) marks "next" that are built so far: LatticeLoss values,
trainer eval/hmc steps, config composition -> Experiment, the `python -m l2hmc` entry."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(autouse=True)
def _restore_dtype():
    old = torch.get_default_dtype()
    yield
    torch.set_default_dtype(old)


def test_lattice_loss_su3(golden):
    torch.set_default_dtype(torch.float64)
    import l2hmc.configs as cfgs
    from l2hmc.lattice.su3.pytorch.lattice import LatticeSU3
    from l2hmc.loss.pytorch.loss import LatticeLoss
    g, want = golden('su3_l2hmc'), golden('loss_su3')
    L = [int(i) for i in g['latvolume']]
    x, xp, acc = dev(g['x']), dev(g['x_prop']), dev(g['acc'])
    lat = LatticeSU3(x.shape[0], L)
    lf = LatticeLoss(lat, cfgs.LossConfig(use_mixed_loss=True, charge_weight=0.0, rmse_weight=0.1,
                                          plaq_weight=0.1))
    assert abs(float(lf(x, xp, acc)) - float(want['su3_loss'])) < 1e-7 * abs(float(want['su3_loss']))
    assert abs(float(lf.plaq_loss(x, xp, acc)) - float(want['su3_plaq'])) < 1e-8
    assert abs(float(lf.rmse_loss(x, xp, acc)) - float(want['su3_rmse'])) < 1e-7
    lf = LatticeLoss(lat, cfgs.LossConfig(use_mixed_loss=False, charge_weight=0.3, rmse_weight=1.0,
                                          plaq_weight=0.5))
    assert abs(float(lf(x, xp, acc)) - float(want['mix_loss'])) < 1e-10
    assert abs(float(lf.charge_loss(x, xp, acc)) - float(want['mix_charge'])) < 1e-14
    m = lf.lattice_metrics(x, dev(g['x_out']))
    assert np.abs(m['dQint'].cpu().numpy() - want['dQint']).max() < 1e-10


def test_lattice_loss_u1(golden):
    torch.set_default_dtype(torch.float32)
    import l2hmc.configs as cfgs
    from l2hmc.lattice.u1.pytorch.lattice import LatticeU1
    from l2hmc.loss.pytorch.loss import LatticeLoss
    g, want = golden('u1_c1'), golden('loss_u1')
    L = [int(i) for i in g['latvolume']]
    x, xp, acc = dev(g['x']), dev(g['x_prop']), dev(g['acc'])
    lat = LatticeU1(x.shape[0], L)
    lf = LatticeLoss(lat, cfgs.LossConfig(use_mixed_loss=True, charge_weight=0.01))
    rel = abs(float(lf(x, xp, acc)) - float(want['default_loss'])) / abs(float(want['default_loss']))
    assert rel < 1e-3, rel
    lf = LatticeLoss(lat, cfgs.LossConfig(use_mixed_loss=False, charge_weight=0.5))
    rel = abs(float(lf(x, xp, acc)) - float(want['plain_loss'])) / abs(float(want['plain_loss']))
    assert rel < 1e-3, rel
    with pytest.raises(RuntimeError):
        LatticeLoss(lat, cfgs.LossConfig(rmse_weight=1.0)).rmse_loss(x, xp, acc)


def test_wilson_loops_tensor_su3(golden):
    """`LatticeSU3.wilson_loops` returns the reference's tensor ([6, nb, T, X, Y, Z] complex,
    lattice/su3/pytorch/lattice.py:242-244): the expressions reference callers apply to it
    (loss/pytorch/loss.py:57-110, lattice.py:208-240) give the reference's values, and it is differentiable."""
    torch.set_default_dtype(torch.float64)
    import l2hmc.configs as cfgs
    from l2hmc.lattice.su3.pytorch.lattice import LatticeSU3, PlaqSums
    from l2hmc.loss.pytorch.loss import LatticeLoss
    go = golden('su3_ops')
    lat0 = LatticeSU3(go['x'].shape[0], [int(i) for i in go['latvolume']])
    w = lat0.wilson_loops(dev(go['x']))
    assert isinstance(w, torch.Tensor) and w.dtype == torch.complex128 and tuple(w.shape) == go['wloops'].shape
    assert np.abs(w.cpu().numpy() - go['wloops']).max() < 1e-13
    assert np.abs(lat0._plaqs(wloops=w).cpu().numpy() - go['plaqs']).max() < 1e-14
    assert np.abs(lat0._sin_charges(wloops=w).cpu().numpy() - go['sinQ']).max() < 1e-14
    assert np.abs(lat0._int_charges(wloops=w).cpu().numpy() - go['intQ']).max() < 1e-13
    assert np.abs(lat0.plaqs(wloops=w).cpu().numpy() - go['plaqs']).max() < 1e-14
    assert np.abs(lat0.charges(wloops=w).intQ.cpu().numpy() - go['intQ']).max() < 1e-13
    ps, rs = lat0._wilson_loops(dev(go['x']))
    assert torch.equal(ps, w) and tuple(rs.shape) == (12, *w.shape[1:]) and float(rs.abs().max()) == 0.0
    assert isinstance(lat0.plaq_sums(dev(go['x'])), PlaqSums)

    g, want = golden('su3_l2hmc'), golden('loss_su3')
    L = [int(i) for i in g['latvolume']]
    x, xp, acc = dev(g['x']), dev(g['x_prop']), dev(g['acc'])
    lat = LatticeSU3(x.shape[0], L)
    w1, w2 = lat.wilson_loops(x), lat.wilson_loops(xp)
    # loss.py:64-70 (`_plaq_loss`, not mixed) written out on the returned tensors
    p1 = w1.real.sum(list(range(2, len(w1.shape))))
    p2 = w2.real.sum(list(range(2, len(w2.shape))))
    ploss = acc * (p2 - p1) ** 2
    pw = torch.tensor(0.1, dtype=torch.float).cuda()              # the reference keeps its weights in fp32
    assert abs(float((-ploss / pw).mean()) - float(want['su3_plaq'])) < 1e-8
    assert abs(float((-ploss / 0.5).mean()) - float(want['mix_plaq'])) < 1e-8
    lf = LatticeLoss(lat, cfgs.LossConfig(use_mixed_loss=False, charge_weight=0.3, rmse_weight=1.0,
                                          plaq_weight=0.5))
    assert abs(float(lf._plaq_loss(w1, w2, acc)) - float(want['mix_plaq'])) < 1e-8
    # loss.py:79-92 (`_charge_loss`)
    q1, q2 = lat._sin_charges(wloops=w1), lat._sin_charges(wloops=w2)
    qw = torch.tensor(0.3, dtype=torch.float).cuda()
    assert abs(float((-(acc * (q2 - q1) ** 2) / qw).mean()) - float(want['mix_charge'])) < 1e-14
    assert abs(float(lf._charge_loss(w1, w2, acc)) - float(want['mix_charge'])) < 1e-14
    # loss.py:100-110 (`lattice_metrics`)
    m = lat.calc_metrics(x=x)
    wo = lat.wilson_loops(x=dev(g['x_out']).reshape(x.shape))
    assert np.abs((lat._int_charges(wloops=wo) - m['intQ']).abs().cpu().numpy() - want['dQint']).max() < 1e-10
    assert np.abs((lat._sin_charges(wloops=wo) - m['sinQ']).abs().cpu().numpy() - want['dQsin']).max() < 1e-10
    # gradients: the reference's loss through the tensor == through the fused per-plane sums; and an arbitrary
    # per-site cotangent against torch autograd over the roll / matmul construction of the same traces
    xa = xp.clone().requires_grad_(True)
    wa = lat.wilson_loops(xa)
    pa = wa.real.sum(list(range(2, len(wa.shape))))
    la = (-(acc * (pa - p1) ** 2) / 0.5).mean() + (-(acc * (lat._sin_charges(wloops=wa) - q1) ** 2) / 0.3).mean()
    la.backward()
    xb = xp.clone().requires_grad_(True)
    lb = lf.plaq_loss(x, xb, acc) + lf.charge_loss(x, xb, acc)
    lb.backward()
    assert abs(float(la) - float(lb)) < 1e-10 * max(1.0, abs(float(lb)))
    assert float((xa.grad - xb.grad).abs().max()) < 1e-10 * max(1.0, float(xb.grad.abs().max()))
    gen = torch.Generator().manual_seed(3)
    cot = torch.complex(torch.randn(w2.shape, generator=gen), torch.randn(w2.shape, generator=gen)).cuda()
    xc = xp.clone().requires_grad_(True)
    (lat.wilson_loops(xc) * cot.conj()).real.sum().backward()
    xd = xp.clone().requires_grad_(True)

    def tr_plaq(xx, u, v):           # plain torch ops (autograd), the reference's construction lattice.py:164-174
        xu, xv = xx[:, u], xx[:, v]
        yuv = xu @ xv.roll(-1, dims=u + 1)
        yvu = xv @ xu.roll(-1, dims=v + 1)
        return torch.diagonal(yuv @ yvu.adjoint(), dim1=-2, dim2=-1).sum(-1)
    ref = torch.stack([tr_plaq(xd, u, v) for u in range(1, 4) for v in range(u)])
    assert float((ref.detach() - w2).abs().max()) < 1e-12
    (ref * cot.conj()).real.sum().backward()
    assert float((xc.grad - xd.grad).abs().max()) < 1e-11 * max(1.0, float(xd.grad.abs().max()))


def test_wilson_loops_tensor_u1(golden):
    """`LatticeU1.wilson_loops` returns the reference's [nb, T, X] plaquette angles (lattice/u1/pytorch/
    lattice.py:154-159); the reference's expressions on it (lattice.py:188-308) and its gradient."""
    torch.set_default_dtype(torch.float32)
    import l2hmc.configs as cfgs
    from l2hmc.lattice.u1.pytorch.lattice import LatticeU1, project_angle
    from l2hmc.loss.pytorch.loss import LatticeLoss
    g, want = golden('u1_c1'), golden('loss_u1')
    L = [int(i) for i in g['latvolume']]
    x, xp, acc = dev(g['x']), dev(g['x_prop']), dev(g['acc'])
    lat = LatticeU1(x.shape[0], L)
    w = lat.wilson_loops(x)
    assert isinstance(w, torch.Tensor) and tuple(w.shape) == g['wloops'].shape
    assert np.array_equal(w.cpu().numpy(), g['wloops'])              # same left-to-right fp32 sum
    assert np.abs(w.cos().mean((1, 2)).cpu().numpy() - g['plaqs']).max() < 1e-6
    assert np.abs(lat._plaqs(wloops=w).cpu().numpy() - g['plaqs']).max() < 1e-6
    assert np.abs(lat._sin_charges(wloops=w).cpu().numpy() - g['sinQ']).max() < 1e-5
    assert np.abs(lat._int_charges(wloops=w).cpu().numpy() - g['intQ']).max() < 1e-5
    assert np.abs((project_angle(w).sum((1, 2)) / (2 * np.pi)).cpu().numpy() - g['intQ']).max() < 1e-5
    assert np.abs(lat._action(w, torch.tensor(float(g['beta']))).cpu().numpy() - g['action']).max() < 1e-3
    assert np.abs(lat.sin_charges(wloops=w).cpu().numpy() - g['sinQ']).max() < 1e-5
    # LatticeLoss._charge_loss / lattice.charge_loss / plaq_loss on tensors (loss.py:72-92, lattice.py:278-308)
    w2 = lat.wilson_loops(xp)
    lf = LatticeLoss(lat, cfgs.LossConfig(use_mixed_loss=False, charge_weight=0.5))
    rel = abs(float(lf._charge_loss(w, w2, acc)) - float(want['plain_loss'])) / abs(float(want['plain_loss']))
    assert rel < 1e-3, rel
    dq = (lat._sin_charges(wloops=w2) - lat._sin_charges(wloops=w)) ** 2
    assert abs(float(lat.charge_loss(acc, wl1=w, wl2=w2)) - float(-(acc * dq + 1e-4).mean(0))) < 1e-6
    pl = -(acc * (2. * (1. - (w2 - w).cos())).sum((1, 2)) + 1e-4).mean(0)
    assert abs(float(lat.plaq_loss(acc, wl1=w, wl2=w2)) - float(pl)) < 1e-5 * abs(float(pl))
    assert abs(float(lat.plaq_loss(acc, x1=x, x2=xp)) - float(pl)) < 1e-4 * abs(float(pl))
    wo = lat.wilson_loops(x=dev(g['x_out']).reshape(x.shape))
    m = lat.calc_metrics(x=x)
    assert np.abs((lat._int_charges(wloops=wo) - m['intQ']).abs().cpu().numpy() - want['dQint']).max() < 1e-4
    assert np.abs((lat._sin_charges(wloops=wo) - m['sinQ']).abs().cpu().numpy() - want['dQsin']).max() < 1e-4
    # gradient of a non-linear functional of the field against torch autograd over the roll construction
    xa = xp.double().clone().requires_grad_(True)
    cot = torch.randn(w.shape, generator=torch.Generator().manual_seed(5)).double().cuda()
    (lat.wilson_loops(xa).sin() * cot).sum().backward()
    xb = xp.double().clone().requires_grad_(True)
    th = xb[:, 0] + xb[:, 1].roll(-1, dims=1) - xb[:, 0].roll(-1, dims=2) - xb[:, 1]
    (th.sin() * cot).sum().backward()
    assert float((xa.grad - xb.grad).abs().max()) < 1e-12


def test_experiment_from_config_su3():
    import l2hmc.configs as cfgs
    from l2hmc.experiment.pytorch.experiment import Experiment
    cfg = cfgs.get_config(['+experiment=su3', 'dynamics.nchains=4', 'network.units=[8]',
                           'annealing_schedule.beta_init=6.0', 'steps.test=3', 'seed=7'])
    ex = Experiment(cfg)
    tr = ex.trainer
    assert tr.dynamics.config.group == 'SU3' and tr.dynamics.vnet.units == [8]
    res = ex.evaluate('eval')
    h = res['history']
    assert len(h['acc']) == 3 and all(torch.isfinite(a).all() for a in h['acc'])
    for key in ('loss', 'plaqs', 'intQ', 'sinQ', 'dQint', 'dQsin', 'acc_mask', 'sumlogdet', 'energy'):
        assert key in h, key
    assert h['energy'][0].shape == (2 * 2 + 1, 4)
    # the trainer re-projects at the start of every step: output of a step is close to SU(3)
    from l2hmc.group.su3.pytorch.utils import checkSU
    avg, mx = checkSU(tr._prep(res['x']))
    assert float(mx.max()) < 1e-12
    res = ex.evaluate('hmc', nsteps=2, eps=0.05, nleapfrog=4)
    assert len(res['history']['acc']) == 2
    rate = res['timer'].get_eval_rate()
    assert rate['num_steps'] == 2 and rate['eval_rate'] > 0
    xt, mt = tr.train_step((res['x'], 6.0))                 # SU(3) training step
    assert np.isfinite(mt['loss']) and xt.shape == res['x'].shape


def test_cli_u1(capsys):
    torch.set_default_dtype(torch.float32)
    from l2hmc.__main__ import main
    out = main(['mode=test', 'dynamics.nchains=16', 'dynamics.latvolume=[8,8]', 'conv=none',
                'steps.nera=2', 'steps.nepoch=3', 'steps.test=2', 'seed=3'])
    assert {'train', 'eval', 'hmc'} <= set(out) and out['eval']['steps'] == 2
    assert out['train']['steps'] == 6 and np.isfinite(out['train']['loss_last'])
    assert out['hmc']['chain_LF_per_s'] > 0
    out = main(['mode=test', 'dynamics.nchains=16', 'dynamics.latvolume=[8,8]', 'conv=none',
                'steps.nera=0', 'steps.test=2', 'seed=3'])
    assert 'train' not in out and {'eval', 'hmc'} <= set(out)


def test_minor_lattice_group_helpers():
    """Small API members outside the sampler's path: U(1) lattice plaq_loss / charge_loss,
    SU(3) per-site plaquette matrices, rsqrtPHM3, eigs3x3."""
    torch.set_default_dtype(torch.float64)
    from l2hmc.group.su3.pytorch import utils as U
    from l2hmc.lattice.su3.pytorch.lattice import LatticeSU3
    from l2hmc.lattice.u1.pytorch.lattice import LatticeU1
    g = torch.Generator().manual_seed(1)
    # U(1)
    lat = LatticeU1(5, [6, 4])
    x1 = (2 * np.pi * torch.rand(5, 2, 6, 4, generator=g) - np.pi).cuda()
    x2 = (2 * np.pi * torch.rand(5, 2, 6, 4, generator=g) - np.pi).cuda()
    acc = torch.rand(5, generator=g).cuda()

    def theta(x):
        return x[:, 0] + torch.roll(x[:, 1], -1, 1) - torch.roll(x[:, 0], -1, 2) - x[:, 1]
    want = -(acc * (2 * (1 - torch.cos(theta(x2) - theta(x1)))).sum((1, 2)) + 1e-4).mean(0)
    assert abs(float(lat.plaq_loss(acc, x1, x2) - want)) < 1e-10
    dq = (torch.sin(theta(x2)).sum((1, 2)) - torch.sin(theta(x1)).sum((1, 2))) / (2 * np.pi)
    assert abs(float(lat.charge_loss(acc, x1, x2) - (-(acc * dq ** 2 + 1e-4).mean(0)))) < 1e-10
    # SU(3): plaquette matrices and their traces against the reduction kernel
    L = [2, 3, 2, 4]
    ls = LatticeSU3(2, L)
    x = ls.random().cuda()
    tot = sum(ls._trace_plaquette(x, u, v).sum((1, 2, 3, 4)) for u in range(1, 4) for v in range(u))
    w = ls.plaq_sums(x)
    assert float((tot.real - w.re).abs().max()) < 1e-9 and float((tot.imag - w.im).abs().max()) < 1e-9
    wt = ls.wilson_loops(x)
    assert float((wt.sum((0, 2, 3, 4, 5)) - tot).abs().max()) < 1e-9
    field, rect = ls._plaquette_field(x)
    assert field.shape == (6, 2, *L, 3, 3) and rect is None
    # X^{-1/2} of a positive Hermitian matrix
    a = torch.complex(torch.randn(7, 3, 3, generator=g), torch.randn(7, 3, 3, generator=g))
    h = a.adjoint() @ a + 0.5 * torch.eye(3)
    r = U.rsqrtPHM3(h)
    assert float((r @ h @ r - torch.eye(3)).abs().max()) < 1e-9
    ev = torch.stack(U.eigs3x3(torch.diagonal(h, dim1=-2, dim2=-1).sum(-1).real,
                               torch.diagonal(h @ h, dim1=-2, dim2=-1).sum(-1).real,
                               torch.linalg.det(h).real), -1)
    assert float((ev.sort(-1).values - torch.linalg.eigvalsh(h)).abs().max()) < 1e-8


@pytest.mark.parametrize('sampler', ['hmc', 'l2hmc', 'l2hmc_fp16'])
def test_u1_samplers_reproduce_exact_plaquette(sampler):
    """Size-independent property: in the 2D U(1) gauge theory <cos theta_P> = I1(beta) / I0(beta)
    (lattice/u1/pytorch/lattice.py:37-42; torus corrections ~ (I1/I0)^V).  Plain HMC and the
    generalised L2HMC update with RANDOM (untrained) networks -- whose Jacobian enters the accept
    step -- both have to sample that distribution; so does the 16-bit-network variant."""
    import l2hmc.configs as cfgs
    from l2hmc.dynamics.pytorch.dynamics import Dynamics
    from l2hmc.lattice.u1.pytorch.lattice import LatticeU1, plaq_exact
    from l2hmc.network.pytorch.network import NetworkFactory
    torch.set_default_dtype(torch.float32)
    torch.manual_seed(11)
    np.random.seed(11)
    L, nb, beta = [8, 8], 1024, 2.0
    dc = cfgs.DynamicsConfig(nchains=nb, group='U1', latvolume=L, nleapfrog=4, eps=0.05,
                             eps_hmc=0.125, verbose=False)
    nc = cfgs.NetworkConfig(units=[16, 16], activation_fn='leaky_relu', dropout_prob=0.0,
                            use_batch_norm=False)
    spec = cfgs.InputSpec(xshape=tuple(dc.xshape), xnet={'x': [dc.xdim, 2], 'v': [dc.xdim]},
                          vnet={'x': [dc.xdim], 'v': [dc.xdim]})
    lat = LatticeU1(nb, L)
    dyn = Dynamics(lat.action, dc, NetworkFactory(spec, nc, cfgs.ConvolutionConfig())).eval()
    if sampler == 'l2hmc_fp16':
        dyn.set_net_precision('fp16')
    x = lat.random().to(dyn.device)
    b = torch.tensor(beta)
    plaqs, accs = [], []
    nsteps, ntherm = 260, 120
    for i in range(nsteps):
        if sampler == 'hmc':
            xo, m = dyn.apply_transition_hmc((x, b), eps=0.125, nleapfrog=8)
        else:
            xo, m = dyn((x, b))
        x = dyn.g.compat_proj(xo.reshape(x.shape))
        if i >= ntherm:
            plaqs.append(lat.plaqs(x).mean())
            accs.append(m['acc'].mean())
    est = float(torch.stack(plaqs).mean())
    acc = float(torch.stack(accs).mean())
    exact = float(plaq_exact(torch.tensor(beta)))
    print(f'{sampler}: <plaq> = {est:.5f}  exact {exact:.5f}  <acc> = {acc:.3f}')
    assert acc > 0.2, acc                      # the chains do move
    assert abs(est - exact) < 4e-3, (sampler, est, exact, acc)


def test_su3_hmc_reproduces_strong_coupling_plaquette():
    """Size-independent property for SU(3): the strong-coupling expansion of the Wilson action,
    <(1/3) Re tr P> = beta/18 + beta^2/216 - 5 beta^4/93312 + O(beta^5), against plain HMC with the
    staple-force / expm / plaquette kernels at beta = 0.9 on 4^4 (64 chains, 100 measured
    trajectories; statistical error ~1e-4)."""
    import l2hmc.configs as cfgs
    from l2hmc.dynamics.pytorch.dynamics import Dynamics
    from l2hmc.lattice.su3.pytorch.lattice import LatticeSU3
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        torch.manual_seed(3)
        beta, L, nb = 0.9, [4, 4, 4, 4], 64
        dc = cfgs.DynamicsConfig(nchains=nb, group='SU3', latvolume=L, nleapfrog=5, eps=0.1,
                                 eps_hmc=0.1, verbose=False, use_split_xnets=False,
                                 use_separate_networks=False)
        lat = LatticeSU3(nb, L)
        dyn = Dynamics(lat.action, dc, None).eval()
        x = lat.random().to(dyn.device)
        b = torch.tensor(beta)
        ps, acc = [], []
        for i in range(160):
            xo, m = dyn.apply_transition_hmc((x, b), eps=0.1, nleapfrog=10)
            x = dyn.g.compat_proj(xo.reshape(x.shape))
            if i >= 60:
                ps.append(lat.plaqs(x).mean())
                acc.append(m['acc'].mean())
        est = float(torch.stack(ps).mean())
        series = beta / 18 + beta ** 2 / 216 - 5 * beta ** 4 / 93312
        print(f'SU(3) HMC beta {beta}: <plaq> = {est:.5f}  series {series:.5f}')
        assert float(torch.stack(acc).mean()) > 0.5
        assert abs(est - series) < 5e-4, (est, series)
    finally:
        torch.set_default_dtype(old)


def test_cli_precision_and_improved_action():
    """`python -m l2hmc` end to end with the non-default switches: 16-bit layers + the default
    conv stack for U(1), the improved gauge action (c1 != 0) for SU(3)."""
    from l2hmc.__main__ import main
    torch.set_default_dtype(torch.float32)
    out = main(['mode=test', 'dynamics.nchains=16', 'dynamics.latvolume=[8,8]', 'precision=fp16',
                'steps.nera=1', 'steps.nepoch=2', 'steps.test=2', 'seed=3'])
    assert {'train', 'eval', 'hmc'} <= set(out) and np.isfinite(out['train']['loss_last'])
    old = torch.get_default_dtype()
    try:
        out = main(['mode=test', 'dynamics.group=SU3', 'dynamics.nchains=4',
                    'dynamics.latvolume=[2,2,2,2]', 'dynamics.nleapfrog=2', 'network.units=[4]',
                    'network.use_batch_norm=false', 'network.dropout_prob=0.0', 'conv=none',
                    'c1=-0.331', 'steps.nera=1', 'steps.nepoch=2', 'steps.test=2', 'seed=3'])
        assert {'train', 'eval', 'hmc'} <= set(out) and np.isfinite(out['train']['loss_last'])
    finally:
        torch.set_default_dtype(old)


def test_train4dsu3_call_sequence():
    """The call sequence of the reference's own SU(3) integration script (train4dSU3.py:61,
    108-190, 200-260) through the product, name for name: seed with setup_torch, load
    conf/su3-min.yaml, `dict_to_list_of_overrides` -> `configs.get_experiment`, `random_state`,
    HMC steps (`trainer.hmc_step`), eval steps, train steps, `BaseHistory.update` +
    `summarize_dict`, `g.checkSU` on the unflattened output."""
    import yaml
    from l2hmc.configs import CONF_DIR, dict_to_list_of_overrides, get_experiment
    from l2hmc.experiment.pytorch.experiment import Experiment
    import l2hmc.group.su3.pytorch.group as g
    from l2hmc.utils.dist import setup_torch
    from l2hmc.utils.history import BaseHistory, summarize_dict
    _ = setup_torch(precision='float64', backend='DDP', seed=4351)
    with (CONF_DIR / 'su3-min.yaml').open('r') as stream:
        conf = dict(yaml.safe_load(stream))
    conf['dynamics']['nchains'] = 8                      # (64 in the file; the sequence is the point)
    overrides = dict_to_list_of_overrides(conf)
    ex = get_experiment(overrides=[*overrides], build_networks=True, keep=['loss'], skip='dt')
    assert isinstance(ex, Experiment) and ex.trainer.keep == ['loss'] and ex.trainer.skip == ['dt']
    assert ex.config.dynamics.group == 'SU3' and list(ex.config.dynamics.latvolume) == [4, 4, 4, 4]
    assert ex.config.network.units == [1] and ex.config.conv.filters in (None, [], '')
    state = ex.trainer.dynamics.random_state(6.0)
    assert isinstance(state.x, torch.Tensor) and isinstance(state.beta, torch.Tensor)
    # HMC(...)  train4dSU3.py:108-149
    hist = BaseHistory()
    x = state.x
    for step in range(3):
        x, metrics_ = ex.trainer.hmc_step((x, state.beta.item()), eps=0.1, nleapfrog=1)
        avgs = hist.update({'hmc_step': step, 'dt': 0.0, **metrics_})
        assert 'acc=' in summarize_dict(avgs)
    xhmc = ex.trainer.dynamics.unflatten(x)
    avg, mx = g.checkSU(xhmc)
    assert float(mx.max()) < 1e-10                   # plain HMC stays on the group manifold
    # eval(...)  :152-193
    hist = BaseHistory()
    x = state.x
    for step in range(3):
        x, metrics_ = ex.trainer.eval_step((x, 6.0))
        avgs = hist.update({'eval_step': step, 'dt': 0.0, **metrics_})
        assert np.isfinite(avgs['loss'])
    assert ex.trainer.dynamics.unflatten(x).shape == state.x.shape
    # the training loop :239-255
    hist = BaseHistory()
    x = state.x
    for step in range(3):
        x, metrics_ = ex.trainer.train_step((x, state.beta))
        avgs = hist.update({'train_step': step, 'dt': 0.0, **metrics_})
        assert np.isfinite(avgs['loss']) and 0.0 <= avgs['acc'] <= 1.0
    assert len(hist.history['loss']) == 3
