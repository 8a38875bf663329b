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
                'steps.test=2', 'seed=3'])
    assert set(out) == {'eval', 'hmc'} and out['eval']['steps'] == 2
    assert out['hmc']['chain_LF_per_s'] > 0
