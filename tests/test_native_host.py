"""CPU checks of the per-link device math of the SU(3) training kernels: the very headers the HIP
kernels include (csrc/su3_math.hpp, csrc/su3_train_math.hpp) are compiled for the host with g++
against a stub <hip/hip_runtime.h> (tests/native_host/) and compared with torch.autograd of
the restatement in tests/emu_native.py.  Catches formula / indexing bugs without a GPU."""
import os
import shutil
import subprocess

import pytest
import torch

import emu_native as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C128 = torch.complex128


@pytest.fixture(scope='module')
def host(tmp_path_factory):
    if shutil.which('g++') is None:
        pytest.skip('no g++')
    exe = tmp_path_factory.mktemp('host') / 'link_math'
    subprocess.run(['g++', '-O1', '-std=c++17', '-I', os.path.join(ROOT, 'tests', 'native_host'),
                    '-I', os.path.join(ROOT, 'l2hmc-qcd_amd', 'csrc'),
                    os.path.join(ROOT, 'tests', 'native_host', 'link_math.cpp'), '-o', str(exe)],
                   check=True)

    def run(op, *operands):
        vals = []
        for t in operands:
            t = torch.view_as_real(t) if t.is_complex() else t
            vals += [repr(float(v)) for v in t.reshape(-1)]
        out = subprocess.run([str(exe)], input=op + ' ' + ' '.join(vals), capture_output=True,
                             text=True, check=True).stdout.split()
        v = torch.tensor([float(x) for x in out], dtype=torch.float64).reshape(-1, 9, 2)
        return [torch.complex(m[:, 0], m[:, 1]).reshape(3, 3) for m in v]
    return run


def crnd(g, *shape, scale=1.0):
    return scale * torch.complex(torch.randn(*shape, dtype=torch.float64, generator=g),
                                 torch.randn(*shape, dtype=torch.float64, generator=g))


def test_expm_and_frechet(host):
    g = torch.Generator().manual_seed(3)
    for scale in (0.05, 0.8, 3.0):
        A, G = crnd(g, 3, 3, scale=scale), crnd(g, 3, 3)
        (e,) = host('expm', A)
        ref = torch.matrix_exp(A)
        assert float((e - ref).abs().max()) < 1e-13 * max(1.0, float(ref.abs().max()))
        Ar = A.clone().requires_grad_(True)
        (gA,) = torch.autograd.grad(torch.matrix_exp(Ar), Ar, G)
        # g_A = L_exp(A^H)[G]: the Cayley-Hamilton form that ships and the Taylor recursion kept for A/B
        for op in ('frechet', 'frechet_series'):
            e2, l = host(op, A.conj().T.contiguous(), G)
            assert float((l - gA).abs().max()) < 1e-12 * max(1.0, float(gA.abs().max())), op
            assert float((e2 - ref.conj().T).abs().max()) < 1e-13 * max(1.0, float(ref.abs().max())), op


def test_frechet_in_the_algebra(host):
    """eps * v of the x-update is (nearly) anti-Hermitian and traceless: tr X ~ 0, det X imaginary -- the
    invariants the Cayley-Hamilton tangent recursion is driven by degenerate there; norms on both sides
    of the scaling threshold (0.25) and the squaring chain (up to 2^6)."""
    g = torch.Generator().manual_seed(11)
    worst = 0.0
    for scale in (1e-6, 1e-3, 0.05, 0.1, 0.2, 0.5, 2.0, 10.0):
        for _ in range(6):
            A = scale * E._tah(crnd(g, 3, 3))
            G = crnd(g, 3, 3)
            Ar = A.clone().requires_grad_(True)
            (gA,) = torch.autograd.grad(torch.matrix_exp(Ar), Ar, G)
            _, l = host('frechet', A.conj().T.contiguous(), G)
            worst = max(worst, float((l - gA).abs().max()) / max(1.0, float(gA.abs().max())))
    assert worst < 1e-12, worst


@pytest.mark.parametrize('kind', ['generic', 'near_su3', 'tah'])
def test_projsu_vec8_vjp(host, kind):
    g = torch.Generator().manual_seed(5)
    worst = 0.0
    for _ in range(25):
        A = crnd(g, 3, 3)
        M = A if kind == 'generic' else (E._proj_su(A) + crnd(g, 3, 3, scale=0.05)
                                         if kind == 'near_su3' else E._tah(A))
        gy = torch.randn(8, dtype=torch.float64, generator=g)
        (got,) = host('projsu_vjp', M, gy)
        Mr = M.clone().requires_grad_(True)
        (ref,) = torch.autograd.grad(E._to_vec8(E._proj_su(Mr)), Mr, gy)
        worst = max(worst, float((got - ref).abs().max()) / max(1.0, float(ref.abs().max())))
    assert worst < (1e-9 if kind == 'tah' else 1e-11), worst


def test_projsu_vec8_vjp_at_unitary_links(host):
    """H = 1 (degenerate spectrum): the reference's closed-form autograd returns NaN here; the
    Jacobi-based VJP must equal the true derivative of the projection (finite differences)."""
    g = torch.Generator().manual_seed(7)
    for _ in range(10):
        M = E._proj_su(crnd(g, 3, 3))
        gy = torch.randn(8, dtype=torch.float64, generator=g)
        (got,) = host('projsu_vjp', M, gy)
        dm = crnd(g, 3, 3)
        f = lambda a: E._to_vec8(E._proj_su(a))
        eps = 1e-6
        fd = float((((f(M + eps * dm) - f(M - eps * dm)) / (2 * eps)) * gy).sum())
        an = float((got.conj() * dm).real.sum())
        assert abs(fd - an) < 1e-7 * max(1.0, abs(fd)), (fd, an)
