"""Pin the numpy oracle against golden vectors produced by the real reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import su3, u1
from helpers import su3_oracle, u1_oracle


def close(a, b, tol):
    assert a.shape == b.shape, (a.shape, b.shape)
    err = float(np.abs(a - b).max()) if a.size else 0.0
    assert err <= tol, f'max abs err {err:.3e} > {tol:.1e}'


def test_su3_ops(golden):
    g = golden('su3_ops')
    x, v, beta = g['x'], g['v'], float(g['beta'])
    close(su3.wilson_loops(x), g['wloops'], 1e-13)
    close(su3.action(x, beta), g['action'], 1e-10)
    close(su3.plaqs(x), g['plaqs'], 1e-14)
    close(su3.sin_charges(x), g['sinQ'], 1e-14)
    close(su3.int_charges(x), g['intQ'], 1e-13)
    close(su3.grad_action(x, beta), g['force'], 1e-13)
    close(su3.rand_tah3(g['normals']), v, 0.0)
    close(su3.kinetic_energy(v), g['kinetic'], 1e-10)
    close(su3.expm(float(g['eps_expm']) * v) @ x, g['expm_v_x'], 1e-14)
    close(su3.expm(g['general']), g['expm_general'], 1e-11)
    close(su3.project_su(g['general']), g['projsu_general'], 1e-12)
    close(su3.project_tah(g['general']), g['tah_general'], 1e-15)
    close(su3.group_to_vec(x), g['vec_x'], 1e-12)
    # projectSU of the (anti-Hermitian, traceless) force is ill-conditioned in the
    # reference itself: numpy vs torch differ by ~3e-9 on identical input
    close(su3.group_to_vec(g['force']), g['vec_force'], 1e-7)
    close(su3.su3_to_vec(g['general']), g['vec_general'], 1e-15)
    close(su3.vec_to_su3(np.moveaxis(g['normals'], 0, -1)), g['vec_to_su3'], 1e-15)
    close(np.stack(su3.check_su(x)), g['checksu_x'], 1e-15)


def test_su3_cold_start():
    """Reference-independent known answers."""
    L = (2, 3, 2, 4)
    x = np.zeros((2, 4, *L, 3, 3), dtype=np.complex128)
    x[..., range(3), range(3)] = 1.0
    vol = int(np.prod(L))
    assert np.allclose(su3.action(x, 6.0), -6 * 6.0 * vol)
    assert np.allclose(su3.plaqs(x), 1.0)
    assert np.abs(su3.grad_action(x, 6.0)).max() == 0.0
    assert np.allclose(su3.kinetic_energy(np.zeros_like(x)), -0.5 * 8 * 4 * vol)


def test_su3_hmc(golden):
    g = golden('su3_hmc')
    from oracle.dynamics import DynamicsOracle
    L = tuple(int(i) for i in g['latvolume'])
    d = DynamicsOracle('SU3', L, 2, [0.02] * 2, [0.02] * 2, [np.zeros(1)] * 2)
    xo, m = d.apply_transition_hmc(g['x'], float(g['beta']), g['normals'], g['u'],
                                   float(g['eps']), int(g['nleapfrog']), history=True)
    close(m['v_init'], g['v_init'], 0.0)
    close(m['x_prop'], g['x_prop'], 1e-12)
    close(m['v_prop'], g['v_prop'], 1e-11)
    close(m['energy'], g['energy'], 1e-8)
    close(m['acc'], g['acc'], 1e-8)
    assert np.array_equal(m['acc_mask'], g['acc_mask'])
    close(xo, g['x_out'], 1e-12)


def test_su3_l2hmc(golden):
    g = golden('su3_l2hmc')
    d = su3_oracle(g)
    x, v, beta = g['x'], g['v0'], float(g['beta'])
    f = d.grad_potential(x, beta)
    close(f, g['force0'], 1e-12)
    s, t, q = d._call_vnet(0, x, f)
    tol_net = 1e-6     # inherits the ill-conditioned projectSU(force) of the reference
    close(s, g['s'], tol_net), close(t, g['t'], tol_net), close(q, g['q'], tol_net)
    v1, ld = d.update_v(0, x, v, beta, True)
    close(v1, g['v_fwd'], 1e-7), close(ld, g['logdet_v_fwd'], 1e-7)
    vb, ldb = d.update_v(1, x, v, beta, False)
    close(vb, g['v_bwd'], 1e-7), close(ldb, g['logdet_v_bwd'], 1e-7)
    x1, _ = d.update_x(0, x, g['v_fwd'], g['masks'][0], True, True)
    close(x1, g['x_fwd'], 1e-13)
    xb, _ = d.update_x(1, x, g['v_fwd'], 1.0 - g['masks'][0], False, False)
    close(xb, g['x_bwd'], 1e-13)
    xl, vl, ll = d.forward_lf(0, x, v, beta)
    close(xl, g['lf_fwd_x'], 1e-8), close(vl, g['lf_fwd_v'], 1e-7)
    close(ll, g['lf_fwd_logdet'], 1e-7)
    xo, m = d.apply_transition_fb(x, beta, g['normals'], g['u'], history=True)
    close(m['v_init'], g['v_init'], 0.0)
    close(m['x_prop'], g['x_prop'], 1e-7)
    close(m['v_prop'], g['v_prop'], 1e-6)
    close(m['energy'], g['energy'], 1e-5)
    close(m['logdet'], g['logdet'], 1e-6)
    close(m['acc'], g['acc'], 1e-5)
    assert np.array_equal(m['acc_mask'], g['acc_mask'])
    close(m['sumlogdet'], g['sumlogdet'], 1e-6)
    close(xo, g['x_out'], 1e-7)


@pytest.mark.parametrize('name', ['u1_conv', 'u1_c1'])
def test_u1(golden, name):
    g = golden(name)
    d = u1_oracle(g)
    x, beta = g['x'], float(g['beta'])
    nb = x.shape[0]
    close(u1.wilson_loops(x), g['wloops'], 1e-5)
    close(u1.action(x, beta), g['action'], 2e-4)
    close(u1.grad_action(x, beta), g['force'], 1e-5)
    close(u1.plaqs(x), g['plaqs'], 1e-6)
    close(u1.sin_charges(x), g['sinQ'], 1e-5)
    close(u1.int_charges(x), g['intQ'], 1e-5)
    v = g['normals'].reshape(nb, -1)
    close(u1.kinetic_energy(v), g['kinetic'], 1e-4)
    f = u1.grad_action(x, beta)
    s, t, q = d._call_vnet(0, x, f)
    close(s, g['vnet_s'], 1e-5), close(t, g['vnet_t'], 1e-5), close(q, g['vnet_q'], 1e-5)
    v1, ld = d.update_v(0, x, v, beta, True)
    close(v1, g['v_fwd'], 1e-5), close(ld, g['logdet_v_fwd'], 1e-4)
    m0 = g['masks'][0]
    xm = m0.reshape(1, *x.shape[1:]) * x
    sx, tx, qx = d._call_xnet(0, True, xm, v)
    close(sx, g['xnet_s'], 1e-5), close(tx, g['xnet_t'], 1e-5), close(qx, g['xnet_q'], 1e-5)
    x1, l1 = d.update_x(0, x, v, m0, True, True)
    close(x1, g['x_fwd'], 2e-5), close(l1, g['logdet_x_fwd'], 1e-4)
    xb, lb = d.update_x(0, x, v, 1.0 - m0, False, False)
    close(xb, g['x_bwd'], 2e-5), close(lb, g['logdet_x_bwd'], 1e-4)
    # plain HMC
    xo, m = d.apply_transition_hmc(x, beta, g['hmc_normals'], g['hmc_u'],
                                   float(g['hmc_eps']), int(g['hmc_nleapfrog']), history=True)
    close(m['energy'], g['hmc_energy'], 5e-3)
    close(m['acc'], g['hmc_acc'], 5e-3)
    assert np.array_equal(m['acc_mask'], g['hmc_acc_mask'])
    close(xo, g['hmc_x_out'].reshape(nb, -1), 1e-4)
    # merged L2HMC trajectory
    xo, m = d.apply_transition_fb(x, beta, g['normals'], g['u'], history=True)
    close(m['energy'], g['energy'], 2e-2)
    close(m['acc'], g['acc'], 1e-2)
    assert np.array_equal(m['acc_mask'], g['acc_mask'])
    # angles live on a circle: compare modulo 2 pi
    dx = np.abs(np.angle(np.exp(1j * (xo - g['x_out'].reshape(nb, -1)))))
    assert dx.max() < 2e-3, dx.max()


def test_su3_improved_action_c1(golden):
    """c1 != 0 (Iwasaki / DBW2 rectangles): oracle restatement against the reference's
    rectangle traces, action and autograd force (tests/golden/make_golden_c1.py)."""
    g = golden('su3_c1')
    x, beta, c1 = g['x'], float(g['beta']), float(g['c1'])
    close(su3.rect_loops(x), g['rects'], 1e-13)
    close(su3.rect_sums(x), g['rect_sum'], 1e-11)
    close(su3.plaq_sums(x)[0], g['plaq_sum'], 1e-11)
    close(su3.action_c1(x, beta, c1), g['action'], 1e-11)
    close(su3.grad_action_c1(x, beta, c1), g['force'], 1e-12)
    # c1 -> 0 reduces to the Wilson action / force
    close(su3.action_c1(x, beta, 0.0), su3.action(x, beta), 1e-12)
    close(su3.grad_action_c1(x, beta, 0.0), su3.grad_action(x, beta), 1e-13)
    # the rectangle staples are the derivative of the rectangle sum: finite differences along a
    # random algebra direction, d/dt sum_R Re tr R (e^{tX} U) = Re tr(X U A)
    rng = np.random.default_rng(3)
    X = su3.project_tah(rng.normal(size=x.shape) + 1j * rng.normal(size=x.shape))
    h = 1e-5
    fd = (su3.rect_sums(su3.expm(h * X) @ x) - su3.rect_sums(su3.expm(-h * X) @ x)) / (2 * h)
    an = np.einsum('...ij,...ji->...', X, x @ su3.rect_staples(x)).real.reshape(x.shape[0], -1).sum(1)
    close(fd, an, 1e-5 * max(1.0, float(np.abs(an).max())))


# ---------------------------------------------------------------- torch-CPU oracle (all cores)
def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a))


def test_torch_cpu_oracle_su3_ops(golden):
    """oracle/torch_cpu.py (bench.py's cpu_baseline) against the reference fixtures."""
    import torch
    from oracle import torch_cpu as tc
    g = golden('su3_ops')
    x, v, beta = _t(g['x']), _t(g['v']), float(g['beta'])
    close(tc.action(x, beta).numpy(), g['action'], 1e-10)
    close(tc.grad_action(x, beta).numpy(), g['force'], 1e-13)
    close(tc.rand_tah3(_t(g['normals'])).numpy(), g['v'], 0.0)
    close(tc.kinetic(v).numpy(), g['kinetic'], 1e-10)
    close(tc.project_su(_t(g['general'])).numpy(), g['projsu_general'], 1e-12)
    close(tc.tah(_t(g['general'])).numpy(), g['tah_general'], 1e-15)
    close(tc.su3_to_vec(tc.project_su(x)).numpy(), g['vec_x'], 1e-12)
    close(tc.su3_to_vec(_t(g['general'])).numpy(), g['vec_general'], 1e-15)
    close((torch.linalg.matrix_exp(float(g['eps_expm']) * v) @ x).numpy(), g['expm_v_x'], 1e-14)


def test_torch_cpu_oracle_trajectories(golden):
    from oracle import torch_cpu as tc
    g = golden('su3_hmc')
    L = [int(i) for i in g['latvolume']]
    d = tc.TorchSU3Dynamics(L, 2, [0.02] * 2, [0.02] * 2, [np.zeros(36 * int(np.prod(L)))] * 2)
    xo, m = d.apply_transition_hmc(_t(g['x']), float(g['beta']), _t(g['normals']), g['u'],
                                   float(g['eps']), int(g['nleapfrog']))
    close(m['x_prop'].numpy(), g['x_prop'], 1e-12)
    close(m['acc'].numpy(), g['acc'], 1e-8)
    assert np.array_equal(m['acc_mask'].numpy(), g['acc_mask'])
    close(xo.numpy().reshape(g['x_out'].shape), g['x_out'], 1e-12)
    g = golden('su3_l2hmc')
    L = [int(i) for i in g['latvolume']]
    w = {k[5:]: _t(v) for k, v in g.items() if k.startswith('vnet.')}
    d = tc.TorchSU3Dynamics(L, int(g['nleapfrog']), g['xeps'], g['veps'], g['masks'], w, nunits=1)
    xo, m = d.apply_transition_fb(_t(g['x']), float(g['beta']), _t(g['normals']), g['u'])
    close(m['v_init'].numpy(), g['v_init'], 0.0)
    close(m['x_prop'].numpy(), g['x_prop'], 1e-7)
    close(m['v_prop'].numpy(), g['v_prop'], 1e-6)
    close(m['acc'].numpy(), g['acc'], 1e-5)
    assert np.array_equal(m['acc_mask'].numpy(), g['acc_mask'])
    close(m['sumlogdet'].numpy(), g['sumlogdet'], 1e-6)
    close(xo.numpy().reshape(g['x_out'].shape), g['x_out'], 1e-7)


@pytest.mark.parametrize('name', ['u1_cfg2_dense', 'u1_cfg2_conv'])
def test_u1_cfg2_shape_oracle_vs_reference(golden, name, monkeypatch):
    """BASELINE cfg-2's own shape (16 x 16, beta 4, nleapfrog 8, separate + split networks, default
    conv stack resp. dense units): the oracle against the reference's merged trajectory
    (tests/golden/make_golden_sizes.py).  The weights are the counter-based numbers of
    tests/golden/seeded.py written into the PRODUCT's state_dict (host logic only, no kernels), so
    this also pins that the product's parameter names / shapes are the reference's."""
    import emu_native
    import helpers
    import torch
    emu_native.install(monkeypatch)
    torch.set_default_dtype(torch.float32)
    g = golden(name)
    dyn, lat = helpers.build_u1_seeded_dynamics(g, 2, device='cpu')
    d = helpers.u1_seeded_oracle(g, dyn)
    x, beta = g['x'], float(g['beta'])
    xo, m = d.apply_transition_fb(x, beta, g['normals'], g['u'], history=True)
    close(m['energy'], g['energy'], 2e-2)                 # |H| ~ 4e2, 16 leapfrog steps in fp32
    close(m['acc'], g['acc'], 5e-3)
    assert np.array_equal(m['acc_mask'], g['acc_mask']) and g['acc_mask'].tolist() == [0.0, 1.0]
    dx = np.abs(np.angle(np.exp(1j * (xo - g['x_out'].reshape(2, -1)))))
    assert dx.max() < 1e-3, dx.max()
    dx = np.abs(np.angle(np.exp(1j * (m['x_prop'].reshape(2, -1) - g['x_prop'].reshape(2, -1)))))
    assert dx.max() < 1e-3, dx.max()


import helpers as _helpers  # noqa: E402


@pytest.mark.parametrize('name', _helpers.MODES_U1 + _helpers.MODES_SU3)
def test_modes_oracle_vs_reference(golden, name):
    """The oracle branches the first fixtures never exercised, pinned to the real reference
    (tests/golden/make_golden_modes.py): merge_directions=False -- single-direction kernel with
    the swapped accept arguments, both directions -- shared / per-step / split network lookup,
    SU(3) separate networks, SU(3) BatchNorm."""
    g = golden(name)
    d = _helpers.modes_oracle(g)
    su3_ = str(g['group']) == 'SU3'
    x, beta = g['x'], float(g['beta'])
    nb = x.shape[0]
    if bool(g['merge_directions']):
        xo, m = d.apply_transition_fb(x, beta, g['normals'], g['u'], history=True)
    else:
        xo, m = d.apply_transition(x, beta, bool(g['forward']), g['normals'], g['u'],
                                   history=True)
    te, ta, tx = (1e-5, 1e-5, 1e-7) if su3_ else (2e-2, 1e-2, 2e-3)
    close(m['v_init'].reshape(nb, -1), g['v_init'].reshape(nb, -1), 0.0)
    close(m['energy'], g['energy'], te)
    close(m['logdet'], g['logdet'], te)
    close(m['acc'], g['acc'], ta)
    assert np.array_equal(m['acc_mask'], g['acc_mask'])
    close(m['sumlogdet'], g['sumlogdet'], te)
    if su3_:
        close(m['x_prop'], g['x_prop'], tx)
        close(xo, g['x_out'], tx)
    else:
        dx = np.abs(np.angle(np.exp(1j * (xo - g['x_out'].reshape(nb, -1)))))
        assert dx.max() < tx, dx.max()
