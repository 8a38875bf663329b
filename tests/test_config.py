"""Config surface: every in-scope file of the reference's conf/ tree composes,
`configs.get_experiment` (configs.py:1008-1034) builds an Experiment, mandatory values and
missing options fail the way hydra's do.  CPU tier (kernels emulated where a model is built)."""
import numpy as np
import pytest
import torch

import l2hmc.configs as cfgs


def test_every_group_option_composes():
    """group=option for every option file shipped under conf/ (28 of round 3 + conv/stack,
    steps/{gpu,hmc,long-debug}, logdir/{default,debug,test}, mode/exp, experiment/beta6-16x16)."""
    n = 0
    for group in sorted(p for p in cfgs.CONF_DIR.iterdir() if p.is_dir()):
        for opt in sorted(group.glob('*.yaml')):
            ov = [f'{"+" if group.name == "experiment" else ""}{group.name}={opt.stem}']
            if opt.stem == 'exp':
                ov.append('name=run1')
            if opt.stem == 'beta6-16x16':
                ov += ['framework=pytorch', 'compression=none']
            cfg = cfgs.get_config(ov)
            ec = cfgs.instantiate(cfg)
            assert ec.dynamics.nleapfrog >= 1 and isinstance(cfg['rundir'], str), ov
            n += 1
    assert n == 34          # 37 files with config.yaml, su3-min.yaml, su3test.yaml
    c = cfgs.instantiate(cfgs.get_config(['conv=stack']))
    assert list(c.conv.filters) == [16, 32, 16] and list(c.conv.sizes) == [3, 5, 3]
    assert cfgs.instantiate(cfgs.get_config(['steps=hmc'])).steps.nera == 0
    assert cfgs.instantiate(cfgs.get_config(['steps=gpu'])).steps.nepoch == 5000
    assert cfgs.instantiate(cfgs.get_config(['steps=long-debug'])).steps.log == 10
    e = cfgs.instantiate(cfgs.get_config(['+experiment=beta6-16x16', 'framework=pytorch',
                                          'compression=none']))
    assert e.name == 'beta6-16x16-pytorch' and e.nchains == 512 and e.steps.test == 5000
    assert e.annealing_schedule.beta_init == 1.0 and e.annealing_schedule.beta_final == 6.0


def test_rundir_interpolation():
    c = cfgs.get_config(['dynamics.latvolume=[8,4]', 'dynamics.nleapfrog=3',
                         'annealing_schedule.beta_final=2.5'])
    assert c['rundir'].startswith('outputs/runs/U1/8x4/nlf-3/beta-2.5/merge_directions-True/pytorch/')
    assert cfgs.get_config(['logdir=debug'])['rundir'].startswith('outputs/debug/runs/pytorch/')
    assert cfgs.get_config(['mode=exp', 'name=abc'])['rundir'].startswith(
        'outputs/experiments/abc/runs/')
    assert cfgs.get_config([], config_name='su3-min')['rundir'].startswith('outputs/runs/SU3/4x4/nlf-1/')


def test_mandatory_values_and_missing_options():
    with pytest.raises(ValueError, match='Missing mandatory value: name'):
        cfgs.instantiate(cfgs.get_config(['mode=exp']))
    with pytest.raises(ValueError, match='framework, compression'):
        cfgs.instantiate(cfgs.get_config(['+experiment=beta6-16x16']))
    # the reference's mode/cpu.yaml and mode/gpu.yaml select options that do not exist in its own
    # tree (dynamics/default_cpu.yaml, steps/gpu.yaml exists but dynamics/gpu.yaml does not):
    # hydra fails on them, and so does selecting them here
    for mode in ('cpu', 'gpu'):
        with pytest.raises(FileNotFoundError):
            cfgs.get_config([f'mode={mode}'])
    with pytest.raises(ValueError, match='tensorflow'):
        cfgs.get_experiment(['framework=tensorflow'])


def test_get_experiment(monkeypatch):
    """configs.get_experiment(overrides, build_networks, keep, skip): no seeding of its own (the
    caller's generator stream continues into the networks, like the reference's)."""
    import emu_native
    emu_native.install(monkeypatch)
    torch.set_default_dtype(torch.float32)
    ov = ['dynamics.group=U1', 'dynamics.latvolume=[4,4]', 'dynamics.nchains=4',
          'dynamics.nleapfrog=2', 'network.units=[6]', 'conv=none', 'seed=77']
    from l2hmc.utils.dist import setup_torch
    setup_torch(seed=123)
    ex = cfgs.get_experiment(ov, build_networks=True, keep='loss', skip=['a', 'b'])
    assert ex.trainer.keep == ['loss'] and ex.trainer.skip == ['a', 'b']
    w1 = ex.trainer.dynamics.vnet['0'].transl.weight.detach().clone()
    after1 = torch.rand(3)
    setup_torch(seed=123)
    ex2 = cfgs.get_experiment(ov)
    assert torch.equal(ex2.trainer.dynamics.vnet['0'].transl.weight.detach(), w1)
    assert torch.equal(torch.rand(3), after1)          # and the stream continues identically
    setup_torch(seed=124)                              # a different caller seed: other weights
    ex3 = cfgs.get_experiment(ov)
    assert not torch.equal(ex3.trainer.dynamics.vnet['0'].transl.weight.detach(), w1)
    ex4 = cfgs.get_experiment(ov, build_networks=False)
    assert not ex4.trainer.dynamics._networks_built
    x, m = ex.trainer.eval_step((ex.lattice.random(), 2.0))
    assert x.shape == (4, 32) and np.isfinite(m['loss'])


def test_config_file_roundtrip_and_steps_update(tmp_path):
    """BaseConfig.to_file / from_file, Steps.update (configs.py:169-178, 371-388)"""
    import l2hmc.configs as cfgs
    s = cfgs.Steps(nera=2, nepoch=10, test=5)
    s2 = s.update(nepoch=20, log=7)
    assert (s2.nera, s2.nepoch, s2.test, s2.log, s2.total) == (2, 20, 5, 7, 40) and s.nepoch == 10
    f = tmp_path / 'steps.json'
    s2.to_file(f)
    s3 = cfgs.Steps(nera=1, nepoch=1, test=1)
    s3.from_file(f)
    assert s3 == s2


def test_su3_utils_module_names():
    """names a notebook may import from group/su3/pytorch/utils.py (reference :39-47, 144-154, 448-514):
    structure constants against the commutators of the basis vec_to_su3 defines, the truncated Taylor expm
    against matrix_exp, eye_like, SU3Gradient on a torch-written function."""
    import torch
    from l2hmc.group.su3.pytorch import utils as U
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        T = [U.vec_to_su3(torch.eye(8)[a]) for a in range(8)]          # generators T^a
        v = torch.randn(8, generator=torch.Generator().manual_seed(0))
        F = U.su3fabc(v)                                                   # [a, b] = f^abc v_c
        for a in range(8):
            for b in range(8):
                comm = T[a] @ T[b] - T[b] @ T[a]                           # = f^abc T^c
                fc = U.su3_to_vec(comm)
                assert abs(float(F[a, b]) - float((fc * v).sum())) < 1e-12, (a, b)
        assert float((F + F.T).abs().max()) == 0.0
        m = 0.2 * U.vec_to_su3(torch.randn(3, 8, generator=torch.Generator().manual_seed(1)))
        assert float((U.expm(m, order=16) - torch.matrix_exp(m)).abs().max()) < 1e-13
        assert torch.equal(U.eye_like(torch.zeros(3, 3)).cpu(), torch.eye(3))
        x = torch.randn(4, 5)
        y, g = U.SU3Gradient(lambda t: (t ** 2).sum(1), x, create_graph=False)
        assert torch.allclose(g, 2 * x.detach()) and y.shape == (4,)
        assert U.f012 == 1.0 and abs(U.f347 - 0.75 ** 0.5) < 1e-15
    finally:
        torch.set_default_dtype(old)
