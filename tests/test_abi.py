"""CPU-only checks of the drop-in boundary: libl2q.so loads and exports every symbol that
include/l2q.h declares (no compute calls without a GPU); the ctypes table mirrors the header;
argument validation returns error codes; the product refuses to run without a GPU."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'l2q.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(l2q_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_header():
    from l2hmc import native
    lib = native.load()
    syms = header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in include/l2q.h but not exported'
    assert sorted(native.SIGNATURES) == syms, set(native.SIGNATURES) ^ set(syms)
    assert lib.l2q_version() >= 100


def test_argument_validation_without_gpu():
    from l2hmc import native
    lib = native.load()
    assert lib.l2q_su3_force(None, 6.0, None, 1, 2, 2, 2, 2, None) == -1          # L2Q_EINVAL
    assert b'null pointer' in lib.l2q_last_error()
    assert lib.l2q_transpose(1, 2, 1, 4, 4, 3, None) == -1                       # bad elem size
    assert lib.l2q_set_tuning(b'force_occ', 7) == -1
    assert lib.l2q_set_tuning(b'force_occ', 2) in (2, 3, 4)
    assert lib.l2q_reduce_ws_bytes(4, 1000) > 0 and lib.l2q_gemm_ws_bytes(256, 256, 262144, 0) > 0


def test_half_layer_kernel_plan_host_logic():
    """Host-side choices of the half-precision layer kernels (no GPU work): which shapes the streaming input-layer
    kernel takes and with how many K-splits (csrc/gemm_f16_skinny.hip), and that the workspace the caller is told
    to bring covers its partial sums and the slab-major copy of the weights."""
    from l2hmc import native
    lib = native.load()
    q = lib.l2q_gemm_h_skinny_splits
    prev = lib.l2q_set_tuning(b'gemm_h_skinny', 1)
    try:
        assert q(8192, 256, 8192, 8192, 0) == 4            # BASELINE cfg-3 vnet: 128 row tiles x 4 = one round of 512
        assert q(8192, 256, 16384, 8192, 1) == 4           # xnet: K counts the cos and the sin columns
        assert q(2048, 256, 2048, 2048, 0) == 8            # 32 row tiles: the largest admissible split count
        assert q(65536, 128, 8192, 0, 0) == 1              # more row tiles than workgroup slots: no split-K
        assert q(8192, 256, 8192 + 64, 8192, 0) == 0       # K not a multiple of the 128-column slab
        assert q(8192, 512, 8192, 8192, 0) == 0            # wider than one workgroup's columns
        assert q(512, 256, 8192, 8192, 0) == 0             # too few chains
        assert q(8192, 256, 1024, 1024, 0) == 0            # short K: the tile kernels
        assert q(8192, 256, 16385, 8192, 1) == 0           # odd K for the cos | sin form
        lib.l2q_set_tuning(b'gemm_h_skinny', 8)
        assert q(8192, 256, 8192, 8192, 0) == 8
        need = lib.l2q_gemm_h_ws_bytes(8192, 256, 8192, 8192)
        assert need >= 8 * 8192 * 256 * 4 + 256 * 16384 * 2
        lib.l2q_set_tuning(b'gemm_h_skinny', 0)
        assert q(8192, 256, 8192, 8192, 0) == 0
        assert lib.l2q_gemm_h_ws_bytes(8192, 256, 8192, 8192) < need
    finally:
        lib.l2q_set_tuning(b'gemm_h_skinny', prev if prev >= 0 else 1)
    assert lib.l2q_set_tuning(b'heads_h_stream', 4) == -1 and lib.l2q_set_tuning(b'heads_h_stream', 2) in (0, 1, 2, 3)
    assert lib.l2q_set_tuning(b'gemm_h_small', 1) in (0, 1)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from l2hmc import _ops as ops, native
    with pytest.raises(native.L2QError):
        ops.su3_pack(torch.zeros(1, 4, 2, 2, 2, 2, 3, 3, dtype=torch.complex128))
    with pytest.raises(native.L2QError):
        ops.u1_wrap(torch.zeros(4))


def test_configs_surface():
    import l2hmc.configs as c
    dc = c.DynamicsConfig(nchains=4, group='SU3', latvolume=[4, 4, 4, 4], nleapfrog=2)
    assert dc.xshape == (4, 4, 4, 4, 4, 4, 3, 3) and dc.xdim == 4 * 256 * 9
    du = c.DynamicsConfig(nchains=8, group='U1', latvolume=[8, 8], nleapfrog=4, eps_hmc=None)
    assert du.xdim == 128 and du.eps_hmc == 0.25
    spec = c.InputSpec(xshape=dc.xshape)
    assert spec.vdim == 4 * 256 * 8
    cc = c.ConvolutionConfig(filters=[8, 16], sizes=[5, 3])
    assert cc.pool == [2, 2]
    a = c.AnnealingSchedule(beta_init=2.0, beta_final=4.0)
    a.setup(nera=3, nepoch=10)
    assert list(a.betas) == [2.0, 3.0, 4.0]
    assert c.dict_to_list_of_overrides({'dynamics': {'nchains': 4}}) == ['dynamics.nchains=4']


def test_reference_api_surface_present():
    """Every public member of the reference modules this build replaces exists here (names
    collected from the reference once: `def` statements of dynamics / lattice / group /
    network / loss; deprecated `_*_deprecated` variants and `_build_networks1` excluded)."""
    import torch  # noqa: F401
    import l2hmc.network.pytorch.network as net
    from l2hmc.dynamics.pytorch.dynamics import Dynamics
    from l2hmc.group.su3.pytorch import utils as su3u
    from l2hmc.group.su3.pytorch.group import SU3
    from l2hmc.group.u1.pytorch import group as u1g
    from l2hmc.lattice.su3.pytorch import lattice as lsu3
    from l2hmc.lattice.u1.pytorch.lattice import LatticeU1
    from l2hmc.loss.pytorch.loss import LatticeLoss
    want = {
        Dynamics: 'apply_transition apply_transition_both apply_transition_fb apply_transition_hmc '
                  'generate_proposal generate_proposal_fb generate_proposal_hmc transition_kernel '
                  'transition_kernel_fb transition_kernel_hmc leapfrog_hmc _forward_lf _backward_lf '
                  '_update_v_fwd _update_v_bwd _update_x_fwd _update_x_bwd _update_v_fwd_hmc '
                  '_update_v_bwd_hmc _update_x_fwd_hmc _update_x_bwd_hmc _call_vnet _call_xnet '
                  '_get_vnet _get_xnet _get_mask _build_masks _build_networks _get_accept_masks '
                  '_get_direction_masks compute_accept_prob get_metrics update_history hamiltonian '
                  'kinetic_energy potential_energy grad_potential random_state test_reversibility '
                  'flatten unflatten group_to_vec vec_to_group complexify _stack_as_xy save load '
                  'save_eps load_eps restore_eps assign_eps init_weights get_models',
        lsu3.LatticeSU3: 'action _action grad_action action_with_grad wilson_loops _wilson_loops '
                         '_plaquette _plaquette_field _trace_plaquette _rectangles _link_staple_op '
                         '_plaquettes plaqs _plaqs charges _charges int_charges _int_charges '
                         'sin_charges _sin_charges calc_metrics coeffs kinetic_energy '
                         'potential_energy random random_momentum update_link plaq_loss charge_loss',
        LatticeU1: 'action _action grad_action action_with_grad wilson_loops wilson_loops4x4 plaqs '
                   '_plaqs plaqs4x4 _plaqs4x4 plaqs_diff charges _charges int_charges _int_charges '
                   'sin_charges _sin_charges calc_metrics observables kinetic_energy '
                   'potential_energy random random_momentum update_link draw_uniform_batch '
                   'plaq_loss charge_loss _get_wloops',
        SU3: 'update_gauge checkSU checkU mul adjoint trace exp projectTAH projectSU projectU '
             'compat_proj compat_proju random random_momentum kinetic_energy vec_to_group '
             'group_to_vec norm2 diff_trace diff2trace rsqrtPHM3 rsqrtPHM3f',
        u1g.U1Phase: 'phase_to_coords coords_to_phase group_to_vec exp update_gauge mul adjoint '
                     'trace diff_trace diff2trace floormod compat_proj projectTAH random '
                     'random_momentum kinetic_energy',
        LatticeLoss: 'mixed_loss plaq_loss charge_loss rmse_loss _plaq_loss _charge_loss general_loss '
                     'lattice_metrics calc_loss',
    }
    missing = [f'{cls.__name__}.{n}' for cls, names in want.items() for n in names.split()
               if not hasattr(cls, n)]
    for mod, names in ((su3u, 'eyeOf norm2 randTAH3 projectU projectSU projectTAH checkSU checkU '
                              'su3_to_vec vec_to_su3 eigs3x3 rsqrtPHM3 rsqrtPHM3f'),
                       (u1g, 'eyeOf rand_unif random_angle'), (lsu3, 'pbc mat_adj'),
                       (net, 'nested_children flatten xy_repr dummy_network init_all init_all_by_shape '
                             'init_weights zero_weights calc_output_size get_network '
                             'get_and_call_network PeriodicPadding ScaledTanh ConvStack InputLayer '
                             'LeapfrogLayer NetworkFactory')):
        missing += [f'{mod.__name__}.{n}' for n in names.split() if not hasattr(mod, n)]
    assert not missing, missing


def test_flat_configs_and_history(tmp_path):
    """conf/su3test.yaml / conf/su3-min.yaml (the reference's flat `--config-name` files) compose
    over the defaults; BaseHistory stacks per-step metrics and dumps the dataset (utils/history.py
    :157-263, 854-909 of the reference)."""
    import numpy as np
    import torch
    import l2hmc.configs as c
    from l2hmc.utils.history import BaseHistory
    ec = c.instantiate(c.get_config(['dynamics.nchains=3'], config_name='su3test'))
    assert (ec.dynamics.group, ec.dynamics.nleapfrog, ec.network.units) == ('SU3', 4, [256])
    assert ec.dynamics.nchains == 3 and ec.conv.filters == [] and ec.net_weights.x.s == 0.0
    assert ec.loss.plaq_weight == 0.1 and ec.learning_rate.lr_init == 1e-4 and ec.steps.test == 50
    em = c.instantiate(c.get_config([], config_name='su3-min'))
    assert (em.dynamics.nleapfrog, em.network.units, em.dynamics.eps) == (1, [1], 0.06)
    assert em.net_weights.x.q == 0.0 and em.loss.rmse_weight == 1.0 and not em.loss.use_mixed_loss
    h = BaseHistory(ec.steps)
    for i in range(6):
        avgs = h.update({'era': 0, 'step': i, 'acc': torch.full((4,), 0.25 * (i % 4)),
                         'energy': torch.ones(3, 4) * i, 'loss': -1.5, 'plaqs': {'mean': 0.5}})
    assert avgs['acc'] == 0.25 and avgs['plaqs/mean'] == 0.5 and 'loss=' in h.era_summary(0)
    ds = h.get_dataset(therm_frac=0.5)
    get = (lambda k: ds[k]) if isinstance(ds, dict) else (lambda k: (tuple(ds[k].dims), ds[k].values))
    assert get('acc')[0] == ('chain', 'draw') and get('acc')[1].shape == (4, 3)
    assert get('energy')[0] == ('chain', 'leapfrog', 'draw') and get('energy')[1].shape == (4, 3, 3)
    f = h.save_dataset(tmp_path, 'eval')
    z = np.load(f)
    assert z['energy'].shape == (4, 3, 6) and z['plaqs_mean'].shape == (6,)
    rows = (tmp_path / 'eval_avgs.csv').read_text().strip().splitlines()
    assert len(rows) == 7 and rows[0].split(',')[0] == 'acc'


def test_comm_entry_points_validate_without_gpu():
    """l2q_init / l2q_comm_* / l2q_allreduce_grads (the thin RCCL wrapper of SURVEY 8(b)(ii)):
    exported, argument checks come before any RCCL / HIP call."""
    import torch
    from l2hmc import native
    lib = native.load()
    assert lib.l2q_allreduce_grads(None, None, 4, 8, None) == -1
    assert b'null pointer' in lib.l2q_last_error()
    assert lib.l2q_comm_init(None, 1, 0, None) == -1
    assert lib.l2q_comm_unique_id(None) == -1
    assert lib.l2q_comm_destroy(None) == -1
    if not torch.cuda.is_available():
        assert lib.l2q_init(0) == -3                       # L2Q_EHIP: no device visible


def test_sliced_heads_waits_are_sound():
    """csrc/heads_sliced.hip counts its vmcnt / lgkmcnt waits by hand; the ISA of every template instance
    keeps at least N memory instructions between the LDS-DMA of a period and the `vmcnt(N)` in front of the
    barrier that publishes the image (tools/check_sliced_waits.py; hipcc cross-compiles without a GPU)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('check_sliced_waits',
                                                  os.path.join(ROOT, 'tools', 'check_sliced_waits.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n, bad, report = mod.check(mod.isa())
    assert n == 16 and bad == 0, report            # 12 inference instances + the 4 TAPE ones


def test_gemm_sliced_isa_discipline():
    """csrc/gemm_sliced.hip issues its activation loads and LDS reads as asm with hand-placed waits; the ISA
    must keep the loads in place (8 register tuples, refilled where they were read, never copied), read a
    loaded register only behind its vmcnt wait, and issue every MFMA behind the lgkmcnt wait of its
    fragments (tools/check_gemm_sliced_isa.py; hipcc cross-compiles without a GPU)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('check_gemm_sliced_isa',
                                                  os.path.join(ROOT, 'tools', 'check_gemm_sliced_isa.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad, report = mod.check(mod.isa())
    assert bad == 0, report


def test_force_plaq_dma_wait_is_sound():
    """csrc/su3_force_plaq.hip publishes its LDS-DMA slice refresh with a hand-counted `s_waitcnt vmcnt(27)`: vmcnt
    retires in order, so the wait is sound only if at least 27 vector-memory instructions are issued between the last
    `global_load_lds_dwordx4` of an iteration and the wait (then "at most 27 outstanding" implies every DMA has
    landed).  Checked in the ISA of the shipped source (hipcc cross-compiles without a GPU)."""
    import re
    import subprocess
    import tempfile
    src = os.path.join(ROOT, 'l2hmc-qcd_amd', 'csrc', 'su3_force_plaq.hip')
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
                        '-Wno-unused-function', '-save-temps=obj', '-c', src, '-o', os.path.join(tmp, 'k.o')],
                       check=True, cwd=os.path.dirname(src), capture_output=True)
        isa = [f for f in os.listdir(tmp) if f.endswith('gfx950.s')]
        assert isa, os.listdir(tmp)
        lines = open(os.path.join(tmp, isa[0])).read().split('\n')
    code = [l.strip() for l in lines]
    waits = [i for i, l in enumerate(code) if l.startswith('s_waitcnt vmcnt(27)')]
    dmas = [i for i, l in enumerate(code) if l.startswith('global_load_lds_dwordx4')]
    assert len(waits) == 1 and len(dmas) == 27, (len(waits), len(dmas))
    # the block that holds the wait (the loop is rotated: it sits above the loop head and is entered by a branch)
    label = next(code[j].split(':')[0] for j in range(waits[0], 0, -1) if code[j].startswith('.LBB'))
    assert re.match(r'\.LBB\d+_\d+$', label), label
    after = next(j for j in range(dmas[-1], len(code)) if re.match(r's_c?branch\w*\s+' + re.escape(label) + r'$', code[j]))
    vmem = [l for l in code[dmas[-1] + 1:after] if re.match(r'(buffer|global|scratch)_(load|store)', l)]
    assert len(vmem) >= 27, len(vmem)
    assert code[waits[0] + 1].startswith('s_barrier') or any(c.startswith('s_barrier') for c in code[waits[0]:waits[0] + 6])
