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
