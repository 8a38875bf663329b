"""Functional wrappers over the C ABI (``include/l2q.h``): tensors in, tensors out.

SU(3) functions ending in ``_n`` take / return fields in the *native* layout
``xn[nb, 4, 9, V]`` complex128 (see l2q.h); ``su3_pack`` / ``su3_unpack`` convert from / to the
reference's public layout ``x[nb, 4, T, X, Y, Z, 3, 3]``.  Nothing here falls back to PyTorch
arithmetic: every function launches HIP kernels from libl2q.so or raises.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch

from . import native as N

C128 = torch.complex128

# bumped whenever parameters are modified by a kernel behind PyTorch's back (the fused Adam
# step on the flat arena does not touch Tensor._version): invalidates cached weight copies
PARAM_GENERATION = [0]


def _vol(lat: Sequence[int]) -> int:
    v = 1
    for n in lat:
        v *= int(n)
    return v


def _ws(nb: int, n_per_chain: int, device) -> torch.Tensor:
    return N.workspace(N.reduce_ws_bytes(nb, n_per_chain), device)


# ---------------------------------------------------------------------------- layout
def transpose(a: torch.Tensor, batch: int, rows: int, cols: int) -> torch.Tensor:
    """[batch, rows, cols] -> [batch, cols, rows] for 4/8/16-byte elements."""
    a = a.contiguous()
    out = torch.empty_like(a)
    N.call('l2q_transpose', a, out, batch, rows, cols, a.element_size())
    return out


def t2d(a: torch.Tensor) -> torch.Tensor:
    """[rows, cols] -> contiguous [cols, rows]."""
    r, c = a.shape
    return transpose(a, 1, r, c).reshape(c, r)


def su3_pack(x: torch.Tensor) -> torch.Tensor:
    """x[nb,4,T,X,Y,Z,3,3] (or any [nb, 4*V*9]-reshapeable) c128 -> xn[nb,4,9,V]."""
    nb = x.shape[0]
    x = x.to(C128).contiguous()
    V = x.numel() // (nb * 36)
    out = torch.empty((nb, 4, 9, V), dtype=C128, device=x.device)
    N.call('l2q_su3_pack', x, out, nb, V)
    return out


def su3_unpack(xn: torch.Tensor, lat: Sequence[int]) -> torch.Tensor:
    nb, _, _, V = xn.shape
    out = torch.empty((nb, 4, *[int(i) for i in lat], 3, 3), dtype=C128, device=xn.device)
    N.call('l2q_su3_unpack', xn.contiguous(), out, nb, V)
    return out


def su3_unpack_select(an: torch.Tensor, bn: torch.Tensor, mask: torch.Tensor,
                      lat: Sequence[int]) -> torch.Tensor:
    """reference-layout x_out[c] = mask[c] ? an[c] : bn[c] from two native fields in one pass"""
    nb, _, _, V = an.shape
    out = torch.empty((nb, 4, *[int(i) for i in lat], 3, 3), dtype=C128, device=an.device)
    N.call('l2q_su3_unpack_select', an.contiguous(), bn.contiguous(), mask.to(torch.float32).contiguous(),
           out, nb, V)
    return out


def pack_entries(a: torch.Tensor, V: int, ncomp: int = 9) -> torch.Tensor:
    """per-entry quantity [nb, 4*V*ncomp] in reference order -> native order [nb, 4*ncomp*V]."""
    nb = a.shape[0]
    return transpose(a.reshape(nb * 4, V, ncomp), nb * 4, V, ncomp).reshape(nb, -1)


def unpack_entries(a: torch.Tensor, V: int, ncomp: int = 9) -> torch.Tensor:
    nb = a.shape[0]
    return transpose(a.reshape(nb * 4, ncomp, V), nb * 4, ncomp, V).reshape(nb, -1)


def native_index(V: int, ncomp: int, device) -> torch.Tensor:
    """idx[j_native] = j_reference for per-entry (ncomp=9) or vec8 (ncomp=8) quantities."""
    return (torch.arange(4 * V * ncomp, device=device).reshape(4, V, ncomp)
            .permute(0, 2, 1).reshape(-1).contiguous())


# ---------------------------------------------------------------------------- SU(3)
def su3_plaq_sums_n(xn: torch.Tensor, lat: Sequence[int]) -> torch.Tensor:
    """[nb, 2]: (sum Re tr P, sum Im tr P) over sites and the 6 planes."""
    nb = xn.shape[0]
    T, X, Y, Z = (int(i) for i in lat)
    out = torch.empty((nb, 2), dtype=torch.float64, device=xn.device)
    ws = _ws(nb, T * X * Y * Z * 36, xn.device)
    N.call('l2q_su3_plaq_reduce', xn, nb, T, X, Y, Z, out, ws, ws.numel())
    return out


def su3_plaq_planes_n(xn: torch.Tensor, lat: Sequence[int]) -> torch.Tensor:
    """[nb, 6, 2]: per-plane (sum Re tr P, sum Im tr P)."""
    nb = xn.shape[0]
    T, X, Y, Z = (int(i) for i in lat)
    out = torch.empty((nb, 6, 2), dtype=torch.float64, device=xn.device)
    ws = N.workspace(N.reduce_ws_bytes(nb, T * X * Y * Z) * 4, xn.device)
    N.call('l2q_su3_plaq_planes', xn, nb, T, X, Y, Z, out, ws, ws.numel())
    return out


def su3_wilson_loops_n(xn: torch.Tensor, lat: Sequence[int]) -> torch.Tensor:
    """[6, nb, T, X, Y, Z] complex128: tr P per plane and site (the reference's `wilson_loops` tensor)."""
    nb = xn.shape[0]
    T, X, Y, Z = (int(i) for i in lat)
    out = torch.empty((6, nb, T, X, Y, Z), dtype=C128, device=xn.device)
    N.call('l2q_su3_wilson_loops', xn, nb, T, X, Y, Z, out)
    return out


def diff_norm2(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """per-chain sum |a - b|^2 (float64 / complex128 tensors of equal shape)."""
    nb = a.shape[0]
    a, b = a.contiguous(), b.contiguous()
    n = a.numel() // nb * (2 if a.is_complex() else 1)
    out = torch.empty(nb, dtype=torch.float64, device=a.device)
    ws = _ws(nb, n, a.device)
    N.call('l2q_diff_norm2_reduce', a, b, nb, n, out, ws, ws.numel())
    return out


def su3_force_n(xn: torch.Tensor, beta: float, lat: Sequence[int]) -> torch.Tensor:
    nb = xn.shape[0]
    T, X, Y, Z = (int(i) for i in lat)
    f = torch.empty_like(xn)
    N.call('l2q_su3_force', xn, float(beta), f, nb, T, X, Y, Z)
    return f


def su3_force_kick_n(xn: torch.Tensor, beta: float, coef: float, vn: torch.Tensor,
                     lat: Sequence[int], v_src: Optional[torch.Tensor] = None) -> torch.Tensor:
    """vn += coef * F(xn) in place; with v_src: vn = v_src + coef * F(xn) (vn only written)."""
    T, X, Y, Z = (int(i) for i in lat)
    if v_src is not None:
        assert v_src.shape == vn.shape and v_src.dtype == vn.dtype and v_src.is_contiguous()
        N.call('l2q_su3_force_kick_to', xn, float(beta), float(coef), v_src, vn, xn.shape[0], T, X, Y, Z)
        return vn
    N.call('l2q_su3_force_kick', xn, float(beta), float(coef), vn, xn.shape[0], T, X, Y, Z)
    return vn


def su3_expm_mul_n(xn: torch.Tensor, vn: torch.Tensor, eps: float,
                   mask_n: Optional[torch.Tensor] = None, complement: bool = False,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    nb, _, _, V = xn.shape
    out = torch.empty_like(xn) if out is None else out
    N.call('l2q_su3_expm_mul', xn, vn, float(eps), mask_n, int(complement), out, nb, V)
    return out


def su3_expm_mul2_n(xn: torch.Tensor, vn: torch.Tensor, eps: float, mask_n: torch.Tensor,
                    complement_first: bool, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Both masked half-updates of one leapfrog step with one expm (== two calls to rounding)."""
    nb, _, _, V = xn.shape
    out = torch.empty_like(xn) if out is None else out
    N.call('l2q_su3_expm_mul2', xn, vn, float(eps), mask_n, int(complement_first), out, nb, V)
    return out


def su3_expm_mul2_vec8_n(xn: torch.Tensor, vn: torch.Tensor, eps: float, mask_n: torch.Tensor,
                         complement_first: bool, out: Optional[torch.Tensor] = None):
    """su3_expm_mul2_n + su3_projsu_vec8_n of its result in one pass: (x', vec8(x'))."""
    nb, _, _, V = xn.shape
    out = torch.empty_like(xn) if out is None else out
    vec = torch.empty((nb, 4, 8, V), dtype=torch.float64, device=xn.device)
    N.call('l2q_su3_expm_mul2_vec8', xn, vn, float(eps), mask_n, int(complement_first), out, vec,
           nb, V)
    return out, vec


def su3_project_su_n(xn: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(xn)
    nf, V = xn.numel() // (9 * xn.shape[-1]), xn.shape[-1]
    N.call('l2q_su3_project_su', xn, out, nf, V)
    return out


def su3_project_tah_n(xn: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(xn)
    nf, V = xn.numel() // (9 * xn.shape[-1]), xn.shape[-1]
    N.call('l2q_su3_project_tah', xn, out, nf, V)
    return out


def su3_project_u_n(xn: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(xn)
    nf, V = xn.numel() // (9 * xn.shape[-1]), xn.shape[-1]
    N.call('l2q_su3_project_u', xn, out, nf, V)
    return out


def su3_mul_n(an: torch.Tensor, bn: torch.Tensor, adjoint_a: bool = False,
              adjoint_b: bool = False) -> torch.Tensor:
    out = torch.empty_like(an)
    nf, V = an.numel() // (9 * an.shape[-1]), an.shape[-1]
    N.call('l2q_su3_mul', an, bn, int(adjoint_a), int(adjoint_b), out, nf, V)
    return out


def su3_projsu_vec8_n(xn: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[..., 9, V] c128 -> [..., 8, V] float64 (native vec8 order).  out: a contiguous float64 buffer of that
    many elements to write instead of a new tensor (the training tape's activation arena)."""
    V = xn.shape[-1]
    nf = xn.numel() // (9 * V)
    shape = (*xn.shape[:-2], 8, V)
    if out is None:
        out = torch.empty(shape, dtype=torch.float64, device=xn.device)
    elif out.numel() != nf * 8 * V or out.dtype != torch.float64 or not out.is_contiguous():
        raise N.L2QError('su3_projsu_vec8_n: bad output buffer')
    N.call('l2q_su3_projsu_vec8', xn, out, nf, V)
    return out.view(shape)


def su3_kinetic_n(vn: torch.Tensor) -> torch.Tensor:
    nb, _, _, V = vn.shape
    out = torch.empty(nb, dtype=torch.float64, device=vn.device)
    ws = _ws(nb, 36 * V, vn.device)
    N.call('l2q_su3_kinetic_reduce', vn, nb, V, out, ws, ws.numel())
    return out


def su3_assemble_tah_n(normals: torch.Tensor) -> torch.Tensor:
    """normals [8, ..., V] float64 -> vn [..., 9, V] (e.g. [8, nb, 4, V] -> [nb, 4, 9, V])."""
    V = normals.shape[-1]
    lead = tuple(normals.shape[1:-1])
    nf = 1
    for i in lead:
        nf *= int(i)
    out = torch.empty((*lead, 9, V), dtype=C128, device=normals.device)
    N.call('l2q_su3_assemble_tah', normals.contiguous(), out, nf, V)
    return out


def su3_check_su_n(xn: torch.Tensor) -> torch.Tensor:
    nb, _, _, V = xn.shape
    out = torch.empty((nb, 2), dtype=torch.float64, device=xn.device)
    ws = _ws(nb, 36 * V, xn.device)
    N.call('l2q_su3_check_su', xn, nb, V, out, ws, ws.numel())
    return out


# ---------------------------------------------------------------------------- shared
def v_update_(v: torch.Tensor, force: torch.Tensor, s: torch.Tensor, t: torch.Tensor,
              q: torch.Tensor, eps: float, forward: bool) -> torch.Tensor:
    """In-place generalised momentum update; returns logdet [nb]."""
    nb = v.shape[0]
    cplx = v.is_complex()
    n = v.numel() // nb
    real_dtype = s.dtype
    logdet = torch.empty(nb, dtype=real_dtype, device=v.device)
    ws = _ws(nb, n, v.device)
    N.call('l2q_v_update', v, force, s, t, q, float(eps), int(forward), int(cplx),
           s.element_size(), nb, n, logdet, ws, ws.numel())
    return logdet


def v_update(v: torch.Tensor, force: torch.Tensor, s: torch.Tensor, t: torch.Tensor,
             q: torch.Tensor, eps: float, forward: bool):
    """Out-of-place generalised momentum update: (v', logdet [nb]); v is not written."""
    nb = v.shape[0]
    n = v.numel() // nb
    out = torch.empty_like(v)
    logdet = torch.empty(nb, dtype=s.dtype, device=v.device)
    ws = _ws(nb, n, v.device)
    N.call('l2q_v_update_to', v, out, force, s, t, q, float(eps), int(forward), int(v.is_complex()),
           s.element_size(), nb, n, logdet, ws, ws.numel())
    return out, logdet


def accept(h_init: torch.Tensor, h_prop: torch.Tensor, sumlogdet: torch.Tensor,
           u: torch.Tensor):
    nb = h_init.shape[0]
    dt = h_init.dtype
    acc = torch.empty(nb, dtype=dt, device=h_init.device)
    mask = torch.empty(nb, dtype=torch.float32, device=h_init.device)
    N.call('l2q_accept', h_init.contiguous(), h_prop.to(dt).contiguous(),
           sumlogdet.to(dt).contiguous(), u.to(dt).contiguous(), acc, mask, nb,
           h_init.element_size())
    return acc, mask


def select_rows(a: torch.Tensor, b: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """out[c] = a[c] if mask[c] else b[c]."""
    nb = a.shape[0]
    a, b = a.contiguous(), b.contiguous()
    row_bytes = a.numel() // nb * a.element_size()
    out = torch.empty_like(a)
    N.call('l2q_select_rows', a, b, mask, out, nb, row_bytes)
    return out


def scale(x: torch.Tensor, alpha: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """alpha * x for float64 / complex128 tensors."""
    out = torch.empty_like(x) if out is None else out
    n = x.numel() * (2 if x.is_complex() else 1)
    N.call('l2q_scale_f64', x, float(alpha), out, n)
    return out


def axpy_(x: torch.Tensor, p: torch.Tensor, alpha: float) -> torch.Tensor:
    N.call('l2q_axpy', p, float(alpha), x, x.numel(), x.element_size())
    return x


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *,
         a2: Optional[torch.Tensor] = None, w2: Optional[torch.Tensor] = None,
         bias2: Optional[torch.Tensor] = None, coeff: Optional[torch.Tensor] = None,
         scale: float = 1.0, act: Optional[str] = None) -> torch.Tensor:
    """epi(a @ w.T (+ a2 @ w2.T) + bias (+ bias2)) on the MFMA GEMM (fp64 or fp32)."""
    m, k = a.shape
    n = w.shape[0]
    k2 = 0 if a2 is None else a2.shape[1]
    if w.shape[1] != k or (a2 is not None and (w2 is None or w2.shape != (n, k2))):
        raise N.L2QError(f'gemm: shape mismatch a{tuple(a.shape)} w{tuple(w.shape)}')
    for other in (w, bias, a2, w2, bias2, coeff):
        if other is not None and other.dtype != a.dtype:
            raise N.L2QError(f'gemm: dtype mismatch {other.dtype} vs {a.dtype}')
    out = torch.empty((m, n), dtype=a.dtype, device=a.device)
    ws = N.workspace(N.gemm_ws_bytes(m, n, k, k2), a.device)
    name = {torch.float64: 'l2q_gemm_f64', torch.float32: 'l2q_gemm_f32'}.get(a.dtype)
    if name is None:
        raise N.L2QError(f'gemm: unsupported dtype {a.dtype}')
    N.call(name, a, w, m, n, k, a2, w2, k2, bias, bias2, coeff, float(scale), N.ACT[act], out,
           ws, ws.numel())
    return out


# int8-sliced fp64 input layer (csrc/gemm_sliced.hip): [True] = use the digit images when the network
# offers them (LeapfrogLayer.kernel_weights builds them for fp64 SU(3) vnets whose shapes qualify)
USE_SLICED_INPUT = [True]
SLICED_INPUT_EXP = 2          # |su3_to_vec(projectSU(.))| < 2.31 < 2^2 for every entry of the vnet inputs


def gemm_sliced_ok(m: int, n: int, k: int, k2: int = 0) -> bool:
    """Shapes csrc/gemm_sliced.hip serves."""
    return m > 0 and m % 64 == 0 and n % 64 == 0 and k > 0 and k % 64 == 0 and k2 % 64 == 0


def gemm_sliced_pays(m: int, n: int, k: int, k2: int = 0) -> bool:
    """Shapes at which the int8-sliced layer is faster than the fp64 MFMA layer (measured at n = 256,
    tools/sweep_gemm_sliced.py, profiles/r04_gemm_sliced_sweep.txt): m (k + k2) >= 2^24, e.g. 256 chains
    with k + k2 >= 65 536 (the 8^4 lattice has 262 144); below it the fixed 64 x 64 tiles leave CUs idle."""
    return gemm_sliced_ok(m, n, k, k2) and m * n * (k + k2) >= 1 << 32


def gemm_sliced_build(w: torch.Tensor):
    """Digit image of one fp64 weight matrix [n, k] (include/l2q.h: l2q_gemm_sliced_build); None when
    the matrix does not qualify (dtype, shape, non-finite entry)."""
    import ctypes
    n, k = w.shape
    if w.dtype != torch.float64 or not w.is_cuda or n % 64 or k % 64:
        return None
    nbytes = int(N.load().l2q_gemm_sliced_bytes(n, k))
    buf = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    usable = ctypes.c_int(0)
    N.call('l2q_gemm_sliced_build', w.contiguous(), n, k, buf, nbytes, ctypes.byref(usable))
    return buf if usable.value else None


def gemm_sliced(a: torch.Tensor, image: torch.Tensor, n: int, bias: Optional[torch.Tensor] = None, *,
                a_exp: int = SLICED_INPUT_EXP, a2: Optional[torch.Tensor] = None,
                image2: Optional[torch.Tensor] = None, a2_exp: int = SLICED_INPUT_EXP,
                bias2: Optional[torch.Tensor] = None, coeff: Optional[torch.Tensor] = None,
                scale: float = 1.0, act: Optional[str] = None) -> torch.Tensor:
    """epi(a @ W.T (+ a2 @ W2.T) + bias (+ bias2)) with W / W2 given as digit images and the fp64
    activations sliced on the fly: every |a| must be < 2^a_exp (else the output is NaN)."""
    m, k = a.shape
    k2 = 0 if a2 is None else a2.shape[1]
    if a.dtype != torch.float64 or not gemm_sliced_ok(m, n, k, k2) or (a2 is not None and image2 is None):
        raise N.L2QError(f'gemm_sliced: a{tuple(a.shape)} n {n} k2 {k2} {a.dtype}')
    out = torch.empty((m, n), dtype=a.dtype, device=a.device)
    ws = N.workspace(int(N.load().l2q_gemm_sliced_ws_bytes(m, n, k, k2)), a.device)
    N.call('l2q_gemm_sliced_f64', a, image, k, int(a_exp), a2, image2, k2, int(a2_exp), m, n, bias, bias2,
           coeff, float(scale), N.ACT[act], out, ws, ws.numel())
    return out


def gemm_ex(a: torch.Tensor, w: torch.Tensor, a_trans: bool = False, w_trans: bool = False,
            out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """C (+)= Aop @ Wop^T without transposed copies (include/l2q.h: l2q_gemm_ex).  a: [m, k], or
    [k, m] with a_trans; w: [n, k], or [k, n] with w_trans.  accumulate needs `out`."""
    (k, m) = a.shape if a_trans else a.shape[::-1]
    (kw, n) = w.shape if w_trans else w.shape[::-1]
    if k != kw or a.dtype != w.dtype or a.dtype not in (torch.float32, torch.float64):
        raise N.L2QError(f'gemm_ex: a{tuple(a.shape)} w{tuple(w.shape)} {a.dtype}/{w.dtype}')
    if out is None:
        if accumulate:
            raise N.L2QError('gemm_ex: accumulate needs an output tensor')
        out = torch.empty((m, n), dtype=a.dtype, device=a.device)
    elif out.numel() != m * n or out.dtype != a.dtype or not out.is_contiguous():
        raise N.L2QError('gemm_ex: bad output tensor')
    aligned = all(t.data_ptr() % 16 == 0 for t in (a, w) if t.device.type != 'cpu')
    if (a_trans or w_trans) and not (
            aligned and (gemm_ex_ok(m, a.dtype) if a_trans else gemm_ex_ok(k, a.dtype))
            and (gemm_ex_ok(n, a.dtype) if w_trans else gemm_ex_ok(k, a.dtype))):
        # rows that are not a multiple of 16 bytes: materialise the transposes instead
        y = gemm(t2d(a) if a_trans else a, t2d(w) if w_trans else w)
        if accumulate:
            return add_(out, y)
        out.copy_(y.reshape(out.shape))
        return out
    ws = N.workspace(N.gemm_ws_bytes(m, n, k, 0), a.device)
    N.call('l2q_gemm_ex', a.contiguous(), int(a_trans), w.contiguous(), int(w_trans), m, n, k,
           a.element_size(), int(accumulate), out, ws, ws.numel())
    return out


def gemm_ex_ok(rows: int, dtype: torch.dtype) -> bool:
    """A transposed operand's row length must be a multiple of 16 bytes."""
    return rows % (2 if dtype == torch.float64 else 4) == 0


HALF_TYPES = {torch.float16: 0, torch.bfloat16: 1}


def gemm_h(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *,
           a2: Optional[torch.Tensor] = None, w2: Optional[torch.Tensor] = None,
           bias2: Optional[torch.Tensor] = None, coeff: Optional[torch.Tensor] = None,
           scale: float = 1.0, act: Optional[str] = None,
           out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """Half-precision layer (include/l2q.h: l2q_gemm_h).  w / w2: float16 or bfloat16 [n, k];
    a / a2: that type or float32 (rounded while staged); bias / bias2 / coeff float32;
    out_dtype: w.dtype (default) or float32 (required with coeff)."""
    m, k = a.shape
    n = w.shape[0]
    k2 = 0 if a2 is None else a2.shape[1]
    hd = w.dtype
    if hd not in HALF_TYPES:
        raise N.L2QError(f'gemm_h: weight dtype {hd} is not a 16-bit float')
    if w.shape[1] != k or (a2 is not None and (w2 is None or w2.shape != (n, k2))):
        raise N.L2QError(f'gemm_h: shape mismatch a{tuple(a.shape)} w{tuple(w.shape)}')
    if a.dtype not in (hd, torch.float32) or (a2 is not None and a2.dtype != a.dtype) or \
            (w2 is not None and w2.dtype != hd):
        raise N.L2QError(f'gemm_h: operand dtypes {a.dtype} / {hd}')
    for other in (bias, bias2, coeff):
        if other is not None and other.dtype != torch.float32:
            raise N.L2QError('gemm_h: bias / coeff must be float32')
    out_dtype = (torch.float32 if coeff is not None else hd) if out_dtype is None else out_dtype
    if out_dtype not in (hd, torch.float32):
        raise N.L2QError(f'gemm_h: output dtype {out_dtype}')
    out = torch.empty((m, n), dtype=out_dtype, device=a.device)
    ws = N.workspace(int(N.load().l2q_gemm_h_ws_bytes(m, n, k, k2)), a.device)
    N.call('l2q_gemm_h', HALF_TYPES[hd], a, int(a.dtype == torch.float32), w, m, n, k, a2, w2, k2,
           bias, bias2, coeff, float(scale), N.ACT[act], out, int(out_dtype == torch.float32),
           ws, ws.numel())
    return out


def gemm_h_u1x(x: torch.Tensor, mask: torch.Tensor, complement: bool, w: torch.Tensor,
               bias: torch.Tensor, a2: torch.Tensor, w2: torch.Tensor, bias2: torch.Tensor,
               act: Optional[str]) -> torch.Tensor:
    """Input layer of the half-precision U(1) xnet with the masked cos / sin formed in the tile
    loader (include/l2q.h: l2q_gemm_h_u1x).  x [nb, xdim] fp32 angles, w [n, 2 xdim] 16-bit,
    a2 = v [nb, k2] fp32, w2 [n, k2] 16-bit -> [nb, n] 16-bit."""
    m, xdim = x.shape
    n = w.shape[0]
    k2 = a2.shape[1]
    hd = w.dtype
    if w.shape[1] != 2 * xdim or w2.shape != (n, k2) or hd not in HALF_TYPES or \
            x.dtype != torch.float32 or a2.dtype != torch.float32:
        raise N.L2QError(f'gemm_h_u1x: x{tuple(x.shape)} w{tuple(w.shape)} w2{tuple(w2.shape)}')
    out = torch.empty((m, n), dtype=hd, device=x.device)
    ws = N.workspace(int(N.load().l2q_gemm_h_ws_bytes(m, n, 2 * xdim, k2)), x.device)
    N.call('l2q_gemm_h_u1x', HALF_TYPES[hd], x.contiguous(), mask.reshape(-1).contiguous(),
           int(complement), w, m, n, xdim, a2.contiguous(), w2, k2, bias, bias2, N.ACT[act], out,
           ws, ws.numel())
    return out


def u1_heads_update_h_(z: torch.Tensor, heads: dict, scale_t: float, a: torch.Tensor,
                       b: torch.Tensor, eps: float, forward: bool, *,
                       mask: Optional[torch.Tensor] = None, complement: bool = False,
                       use_ncp: bool = True, acc: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Half-precision heads + sub-update in one kernel (include/l2q.h: l2q_u1_heads_update_h).
    heads: {'s': (W16, b, cs), 't': (W16, b, None), 'q': (W16, b, cq)} with cs / cq the fp32
    per-entry scales nw * exp(coeff).  mask given: x-update (a = x, b = v), else v-update
    (a = v, b = force); a is updated in place.  Returns logdet [nb] fp32 (acc: added into it)."""
    m, k = z.shape
    ws_, bs, cs = heads['s']
    wt, bt, _ = heads['t']
    wq, bq, cq = heads['q']
    n = ws_.shape[0]
    if z.dtype not in HALF_TYPES or ws_.dtype != z.dtype:
        raise N.L2QError(f'u1_heads_update_h: dtypes {z.dtype} / {ws_.dtype}')
    if a.dtype != torch.float32 or b.dtype != torch.float32 or a.numel() != m * n:
        raise N.L2QError('u1_heads_update_h: the lattice operands are fp32 [nb, n]')
    logdet = acc if acc is not None else torch.empty(m, dtype=torch.float32, device=z.device)
    ws = N.workspace(int(N.load().l2q_u1_heads_update_h_ws_bytes(m, n)), z.device)
    N.call('l2q_u1_heads_update_h', HALF_TYPES[z.dtype], z, m, k, n, ws_, bs, cs, wt, bt,
           float(scale_t), wq, bq, cq, int(mask is not None), a, b, mask, int(complement),
           float(eps), int(forward), int(use_ncp), logdet, int(acc is not None), ws, ws.numel())
    return logdet


# int8-sliced heads (csrc/heads_sliced.hip): [True] = use the slice image when the network offers one
# (`heads['sliced']`, built by network.kernel_weights for fp64 heads with K = 256 outside training)
USE_SLICED_HEADS = [True]
SLICED_K = 256


SLICED_IMAGE_MAX_FREE_FRACTION = [0.25]     # heads_sliced_build: largest share of the free device memory


# ---- memory gates: ONE helper for every "take the faster path only while it fits comfortably" decision
# (ADVICE r04).  The int8-sliced and the fp64 kernels agree to rounding, not bit for bit, so a gate that
# silently flips with the allocator's state would make a run irreproducible from its seed and could put the
# ranks of a data-parallel job on different paths.  Every refusal is therefore RECORDED (MEM_GATE_LOG) and logged
# once per gate; `MEM_GATE_POLICY[0]` makes the choice explicit: 'auto' (default: gate on free + idle memory),
# 'always' (take the path, let the allocator fail loudly if it cannot), 'never' (the fp64 / un-deferred path).
# `mem_gate_signature()` is what a distributed job compares across ranks (Trainer does, and warns).
MEM_GATE_POLICY = ['auto']
MEM_GATE_LOG: dict = {}


def device_headroom(device) -> int:
    """bytes a new allocation can count on: free device memory + the caching allocator's idle blocks"""
    if not (isinstance(device, torch.device) and device.type == 'cuda'):
        return 1 << 62
    free, _total = torch.cuda.mem_get_info(device)
    return int(free + torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device))


# In a data-parallel job the ranks must take the SAME path (the overlapped gradient exchange launches
# one collective per finished parameter range, and which ranges exist depends on these gates -- ranks that
# disagree would issue mismatched all-reduces).  With torch.distributed initialised and more than one rank, an
# 'auto' decision is therefore made ONCE per (gate, size) by all ranks together -- the path is taken only if
# every rank can afford it (all-reduce MIN, before the first kernel that depends on it) -- and kept for the rest
# of the process.  Every rank reaches a gate in the same program order, so the collective matches up.
MEM_GATE_CONSENSUS = [True]
_MEM_GATE_AGREED: dict = {}


def _multi_rank() -> bool:
    import torch.distributed as dist
    return bool(MEM_GATE_CONSENSUS[0] and dist.is_available() and dist.is_initialized()
                and dist.get_world_size() > 1)


def _ranks_agree(ok: bool, device) -> bool:
    """all-reduce MIN of one rank-local yes / no"""
    import torch.distributed as dist
    backend = str(dist.get_backend())
    on_dev = 'nccl' in backend and isinstance(device, torch.device) and device.type == 'cuda'
    if not on_dev and 'gloo' not in backend:
        return ok                                     # (an RCCL-only group and a host-side decision: nothing to compare)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if on_dev else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def mem_gate(what: str, need_bytes: int, fraction: float, device) -> bool:
    """True: `need_bytes` may be taken for `what` (at most `fraction` of the headroom)."""
    pol = MEM_GATE_POLICY[0]
    if pol == 'always':
        ok = True
    elif pol == 'never':
        ok = False
    else:
        key = (what, int(need_bytes))
        if key in _MEM_GATE_AGREED:
            ok = _MEM_GATE_AGREED[key]
        else:
            ok = need_bytes <= fraction * device_headroom(device)
            if _multi_rank():
                ok = _MEM_GATE_AGREED[key] = _ranks_agree(ok, device)
    prev = MEM_GATE_LOG.get(what)
    MEM_GATE_LOG[what] = ok
    if not ok and prev is not False:
        import logging
        logging.getLogger('l2hmc').warning(
            'memory gate: %s not taken (%.2f GiB wanted, policy %s) -- the slower equivalent path runs; '
            'results agree with the faster one to rounding only', what, need_bytes / 2 ** 30, pol)
    return ok


def mem_gate_signature() -> tuple:
    return tuple(sorted(MEM_GATE_LOG.items()))


def heads_sliced_build(heads: dict):
    """int8 slice image of the three head weight matrices (include/l2q.h: l2q_heads_sliced_build).
    Returns the uint8 device buffer, or None when the weights do not qualify (dtype, K, dynamic range)."""
    import ctypes
    ws_, wt, wq = heads['s'][0], heads['t'][0], heads['q'][0]
    n, k = ws_.shape
    if ws_.dtype != torch.float64 or k != SLICED_K or not ws_.is_cuda:
        return None
    nbytes = int(N.load().l2q_heads_sliced_bytes(k, n))
    # the image lives NEXT TO the fp64 weights (+87 % of their size: 0.7 GB per vnet at 8^4, 11 GB at 16^4):
    # built only while it is a small part of what the device still has (free + the allocator's idle blocks)
    if not mem_gate('sliced heads image', nbytes, SLICED_IMAGE_MAX_FREE_FRACTION[0], ws_.device):
        return None
    buf = torch.empty(nbytes, dtype=torch.uint8, device=ws_.device)
    usable = ctypes.c_int(0)
    N.call('l2q_heads_sliced_build', ws_, wt, wq, k, n, buf, nbytes, ctypes.byref(usable))
    return buf if usable.value else None


def heads_sliced_build_into(ws_: torch.Tensor, wt: torch.Tensor, wq: torch.Tensor,
                            buf: Optional[torch.Tensor] = None):
    """The slice image of three [n, 256] fp64 weight matrices, written into `buf` when it has the right size
    (the training loop rebuilds the image every optimiser step: no allocation after the first).
    Returns (buffer, usable)."""
    import ctypes
    n, k = ws_.shape
    nbytes = int(N.load().l2q_heads_sliced_bytes(k, n))
    if buf is None or buf.numel() != nbytes or buf.device != ws_.device:
        if not mem_gate('sliced heads image (training tape)', nbytes, SLICED_IMAGE_MAX_FREE_FRACTION[0],
                        ws_.device):
            return None, False                       # (same memory gate as heads_sliced_build)
        buf = torch.empty(nbytes, dtype=torch.uint8, device=ws_.device)
    usable = ctypes.c_int(0)
    N.call('l2q_heads_sliced_build', ws_, wt, wq, k, n, buf, nbytes, ctypes.byref(usable))
    return buf, bool(usable.value)


def vnet_heads_vupdate_sliced_tape(z: torch.Tensor, image: torch.Tensor, bs: torch.Tensor, cs: torch.Tensor,
                                   bt: torch.Tensor, scale_t: float, bq: torch.Tensor, cq: torch.Tensor,
                                   v: torch.Tensor, force: torch.Tensor, eps: float, forward: bool):
    """Forward pass of the training tape on the int8-sliced heads kernel (include/l2q.h:
    l2q_vnet_heads_vupdate_sliced_tape_f64): (v', logdet [nb], s, t, q); v is not written.
    cs / cq: per-entry scales nw.s exp(coeff_s) / nw.q exp(coeff_q)."""
    m, k = z.shape
    n = bs.numel()
    out = torch.empty_like(v)
    s = torch.empty((m, n), dtype=torch.float64, device=z.device)
    t = torch.empty_like(s)
    q = torch.empty_like(s)
    logdet = torch.empty(m, dtype=torch.float64, device=z.device)
    nbytes = int(N.load().l2q_vnet_heads_sliced_ws_bytes(m, n))
    ws = N.workspace(nbytes, z.device)
    N.call('l2q_vnet_heads_vupdate_sliced_tape_f64', z, m, k, n, image, bs, cs, bt, float(scale_t), bq, cq, v,
           out, force, int(v.is_complex()), float(eps), int(forward), s, t, q, logdet, ws, ws.numel())
    return out, logdet, s, t, q


def heads_sliced_zflag(reset: bool = True) -> int:
    """Number of activation rows handed to the sliced heads kernel since the last reset whose mean
    |entry| was below 2^-6 of their largest one (include/l2q.h: l2q_heads_sliced_zflag; synchronises).
    0 with tanh networks; > 0 means the sliced products of those rows carry the absolute bound
    2^-54 K max|z| max|w| instead of fp64's relative one -- `dyn.sliced_heads = False` selects fp64."""
    import ctypes
    count = ctypes.c_int(0)
    N.call('l2q_heads_sliced_zflag', int(reset), ctypes.byref(count))
    return int(count.value)


def _sliced_call(z, heads, scales, v, force, eps1, forward1, pair, flip, eps2, forward2, v_src=None,
                 mid=False):
    m, k = z.shape
    _, bs, cs = heads['s']
    _, bt, _ = heads['t']
    wq, bq, cq = heads['q']
    n = wq.shape[0]
    out = torch.empty((3 if mid else 1, m), dtype=torch.float64, device=z.device)
    nbytes = int(N.load().l2q_vnet_heads_sliced_ws_bytes(m, n))
    ws = N.workspace(nbytes, z.device)
    N.call('l2q_vnet_heads_vupdate_sliced_f64', z, m, k, n, heads['sliced'], bs, cs, float(scales[0]), bt,
           float(scales[1]), bq, cq, float(scales[2]), v_src, v, force, int(v.is_complex()), float(eps1),
           int(forward1), int(pair), int(flip), float(eps2), int(forward2), out[0],
           out[1] if mid else None, out[2] if mid else None, ws, ws.numel())
    return out


def _use_sliced(z, heads) -> bool:
    # (the sliced kernel takes per-entry scales; a heads dict with scalar scales stays on the fp64 kernel)
    return (USE_SLICED_HEADS[0] and heads.get('use_sliced', True) and heads.get('sliced') is not None
            and z.dtype == torch.float64
            and z.shape[1] == SLICED_K and heads['s'][2] is not None and heads['q'][2] is not None)


def vnet_heads_vupdate_(z: torch.Tensor, heads: dict, scales, v: torch.Tensor,
                        force: torch.Tensor, eps: float, forward: bool,
                        v_src: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Fused (s, t, q) heads + generalised momentum update, v in place; returns logdet [nb].
    heads: {'s': (W, b, colscale|None), 't': (W, b, None), 'q': (W, b, colscale|None)};
    scales = (nw.s, nw.t, nw.q) used where no per-column scale is given.
    v_src: read the momentum from there instead (v is then only written)."""
    m, k = z.shape
    ws_, bs, cs = heads['s']
    wt, bt, _ = heads['t']
    wq, bq, cq = heads['q']
    n = ws_.shape[0]
    if _use_sliced(z, heads):
        if v_src is not None:
            assert v_src.shape == v.shape and v_src.dtype == v.dtype and v_src.is_contiguous()
        return _sliced_call(z, heads, scales, v, force, eps, forward, False, False, 0.0, False, v_src)[0]
    logdet = torch.empty(m, dtype=torch.float64, device=z.device)
    nbytes = int(N.load().l2q_vnet_heads_ws_bytes(m, n))
    ws = N.workspace(nbytes, z.device)
    if v_src is not None:
        assert v_src.shape == v.shape and v_src.dtype == v.dtype and v_src.is_contiguous()
        N.call('l2q_vnet_heads_vupdate_to_f64', z, m, k, n, ws_, bs, cs, float(scales[0]), wt, bt,
               float(scales[1]), wq, bq, cq, float(scales[2]), v_src, v, force,
               int(v.is_complex()), float(eps), int(forward), logdet, ws, ws.numel())
        return logdet
    N.call('l2q_vnet_heads_vupdate_f64', z, m, k, n, ws_, bs, cs, float(scales[0]), wt, bt,
           float(scales[1]), wq, bq, cq, float(scales[2]), v, force, int(v.is_complex()),
           float(eps), int(forward), logdet, ws, ws.numel())
    return logdet


def vnet_heads_vupdate_pair_(z: torch.Tensor, heads: dict, scales, v: torch.Tensor,
                             force: torch.Tensor, eps1: float, forward1: bool, flip: bool,
                             eps2: float, forward2: bool) -> torch.Tensor:
    """Two consecutive v-updates on the same x from one evaluation of the heads (optionally
    with v -> -v in between); returns the summed logdet [nb]."""
    m, k = z.shape
    ws_, bs, cs = heads['s']
    wt, bt, _ = heads['t']
    wq, bq, cq = heads['q']
    n = ws_.shape[0]
    if _use_sliced(z, heads):
        return _sliced_call(z, heads, scales, v, force, eps1, forward1, True, flip, eps2, forward2)[0]
    logdet = torch.empty(m, dtype=torch.float64, device=z.device)
    ws = N.workspace(int(N.load().l2q_vnet_heads_ws_bytes(m, n)), z.device)
    N.call('l2q_vnet_heads_vupdate_pair_f64', z, m, k, n, ws_, bs, cs, float(scales[0]), wt, bt,
           float(scales[1]), wq, bq, cq, float(scales[2]), v, force, int(v.is_complex()),
           float(eps1), int(forward1), int(flip), float(eps2), int(forward2), logdet, ws,
           ws.numel())
    return logdet


def vnet_heads_vupdate_pair_mid_(z: torch.Tensor, heads: dict, scales, v: torch.Tensor,
                                 force: torch.Tensor, eps1: float, forward1: bool, flip: bool,
                                 eps2: float, forward2: bool):
    """vnet_heads_vupdate_pair_ that also returns what per-step metrics need of the momentum between
    the two updates: (logdet_sum, logdet_first, sum |v_mid|^2), each [nb]
    (include/l2q.h: l2q_vnet_heads_vupdate_pair_mid_f64; needs K % 16 == 0)."""
    m, k = z.shape
    ws_, bs, cs = heads['s']
    wt, bt, _ = heads['t']
    wq, bq, cq = heads['q']
    n = ws_.shape[0]
    if _use_sliced(z, heads):
        out = _sliced_call(z, heads, scales, v, force, eps1, forward1, True, flip, eps2, forward2, mid=True)
        return out[0], out[1], out[2]
    out = torch.empty((3, m), dtype=torch.float64, device=z.device)
    ws = N.workspace(int(N.load().l2q_vnet_heads_ws_bytes(m, n)), z.device)
    N.call('l2q_vnet_heads_vupdate_pair_mid_f64', z, m, k, n, ws_, bs, cs, float(scales[0]), wt, bt,
           float(scales[1]), wq, bq, cq, float(scales[2]), v, force, int(v.is_complex()),
           float(eps1), int(forward1), int(flip), float(eps2), int(forward2), out[0], out[1], out[2],
           ws, ws.numel())
    return out[0], out[1], out[2]


# ---------------------------------------------------------------------------- U(1)
def u1_plaq_sums(x: torch.Tensor, lat: Sequence[int]) -> torch.Tensor:
    """[nb, 3]: sum cos(theta), sum sin(theta), sum project_angle(theta)."""
    nb = x.shape[0]
    T, X = (int(i) for i in lat)
    out = torch.empty((nb, 3), dtype=x.dtype, device=x.device)
    N.call('l2q_u1_plaq_reduce', x, nb, T, X, x.element_size(), out)
    return out


def u1_wilson_loops(x: torch.Tensor, lat: Sequence[int]) -> torch.Tensor:
    """[nb, T, X]: the plaquette angles theta (the reference's `wilson_loops` tensor)."""
    T, X = (int(i) for i in lat)
    out = torch.empty((x.shape[0], T, X), dtype=x.dtype, device=x.device)
    N.call('l2q_u1_wilson_loops', x, x.shape[0], T, X, x.element_size(), out)
    return out


def u1_wilson_loops_bwd_(dx: torch.Tensor, g: torch.Tensor, lat: Sequence[int]) -> torch.Tensor:
    """dx [nb, 2, T, X] += adjoint of the plaquette-angle map applied to g [nb, T, X]."""
    T, X = (int(i) for i in lat)
    N.call('l2q_u1_wilson_loops_bwd', g.to(dx.dtype).contiguous(), dx.shape[0], T, X, dx.element_size(), dx)
    return dx


def u1_force(x: torch.Tensor, beta: float, lat: Sequence[int]) -> torch.Tensor:
    T, X = (int(i) for i in lat)
    f = torch.empty_like(x)
    N.call('l2q_u1_force', x, float(beta), f, None, 0.0, x.shape[0], T, X, x.element_size())
    return f


def u1_force_kick_(x: torch.Tensor, beta: float, coef: float, v: torch.Tensor,
                   lat: Sequence[int]) -> torch.Tensor:
    T, X = (int(i) for i in lat)
    N.call('l2q_u1_force', x, float(beta), None, v, float(coef), x.shape[0], T, X,
           x.element_size())
    return v


def u1_x_update_(x: torch.Tensor, v: torch.Tensor, s: torch.Tensor, t: torch.Tensor,
                 q: torch.Tensor, mask: torch.Tensor, complement: bool, eps: float,
                 forward: bool, use_ncp: bool = True) -> torch.Tensor:
    nb = x.shape[0]
    n = x.numel() // nb
    logdet = torch.empty(nb, dtype=x.dtype, device=x.device)
    N.call('l2q_u1_x_update', x, v, s, t, q, mask, int(complement), float(eps), int(forward),
           int(use_ncp), x.element_size(), nb, n, logdet)
    return logdet


def u1_wrap(x: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(x)
    N.call('l2q_u1_wrap', x, out, x.numel(), x.element_size())
    return out


def u1_kinetic(v: torch.Tensor) -> torch.Tensor:
    nb = v.shape[0]
    out = torch.empty(nb, dtype=v.dtype, device=v.device)
    N.call('l2q_u1_kinetic_reduce', v, nb, v.numel() // nb, v.element_size(), out)
    return out


def u1_masked_cos_sin(x: torch.Tensor, mask: torch.Tensor, complement: bool,
                      lat: Sequence[int]) -> torch.Tensor:
    """[nb, 4, T, X]: cat([cos(m x), sin(m x)], dim=1)."""
    nb = x.shape[0]
    n = x.numel() // nb
    out = torch.empty((nb, 4, *[int(i) for i in lat]), dtype=x.dtype, device=x.device)
    N.call('l2q_u1_masked_cos_sin', x, mask, int(complement), out, nb, n, x.element_size())
    return out


def conv2d_periodic(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, pool: int = 1,
                    act: Optional[str] = None) -> torch.Tensor:
    nb, cin, H, W = x.shape
    cout, _, k, _ = w.shape
    pool = max(int(pool), 1)
    Ho, Wo = (H + k - 1) // pool, (W + k - 1) // pool
    out = torch.empty((nb, cout, Ho, Wo), dtype=torch.float32, device=x.device)
    N.call('l2q_conv2d_periodic_f32', x.float().contiguous(), w.float().contiguous(),
           b.float().contiguous(), out, nb, cin, H, W, cout, k, pool, N.ACT[act])
    return out


def conv2d_periodic_gemm(x: torch.Tensor, layout: str, w: torch.Tensor, b: torch.Tensor,
                         pool: int = 1, act: Optional[str] = None,
                         w_clast: Optional[torch.Tensor] = None) -> torch.Tensor:
    """PeriodicPadding(k-1) -> Conv2d(k) -> [MaxPool2d(pool)] -> [act] as implicit GEMM on
    the f32 MFMA kernel.  x: [nb, C, H, W] if layout == 'nchw' else [nb, H, W, C];
    returns NHWC [nb, Ho, Wo, cout].  w_clast: the weight as [cout, k, k, C] (used for NHWC
    inputs: contiguous gathers); made on the fly when not supplied."""
    x = x.contiguous()
    if layout == 'nchw':
        nb, C, H, W = x.shape
        sn, sc, sh, sw = C * H * W, H * W, W, 1
    else:
        nb, H, W, C = x.shape
        sn, sc, sh, sw = H * W * C, 1, W * C, C
    cout, cin, k, _ = w.shape
    assert cin == C
    Ho, Wo, Kc = H + k - 1, W + k - 1, C * k * k
    pool = max(int(pool), 1)
    y = torch.empty((nb * Ho * Wo, cout), dtype=torch.float32, device=x.device)
    # im2col inside the GEMM's A-tile loader: no col matrix in HBM
    clast = layout != 'nchw'
    if clast:
        wk = (w.permute(0, 2, 3, 1) if w_clast is None else w_clast).reshape(cout, Kc).contiguous()
    else:
        wk = w.reshape(cout, Kc).contiguous()
    N.call('l2q_conv_gemm_periodic_f32', x, sn, sc, sh, sw, nb, C, H, W, k, wk, int(clast),
           b.contiguous(), cout, N.ACT[None if pool > 1 else act], y)
    if pool == 1:
        return y.reshape(nb, Ho, Wo, cout)
    out = torch.empty((nb, Ho // pool, Wo // pool, cout), dtype=torch.float32, device=x.device)
    N.call('l2q_maxpool_act_nhwc_f32', y, nb, Ho, Wo, cout, pool, N.ACT[act], out)
    return out


def nchw_to_nhwc_pad(x: torch.Tensor, cpad: int) -> torch.Tensor:
    """fp32 [nb, C, H, W] -> fp32 [nb, H, W, cpad], channels zero-padded."""
    nb, C, H, W = x.shape
    out = torch.empty((nb, H, W, cpad), dtype=torch.float32, device=x.device)
    N.call('l2q_nchw_to_nhwc_pad_f32', x.contiguous(), nb, C, H, W, cpad, out)
    return out


def nchw_to_nhwc_pad_h(x: torch.Tensor, hd: torch.dtype, cpad: int = 8) -> torch.Tensor:
    """fp32 [nb, C, H, W] -> 16-bit [nb, H, W, cpad], channels zero-padded."""
    nb, C, H, W = x.shape
    out = torch.empty((nb, H, W, cpad), dtype=hd, device=x.device)
    N.call('l2q_nchw_to_nhwc_pad_h', HALF_TYPES[hd], x.contiguous(), nb, C, H, W, cpad, out)
    return out


FUSE_CONV_POOL_H = [True]      # switch for A/B and the equality test (same bits either way)


def conv2d_periodic_gemm_h(x: torch.Tensor, layout: str, w16: torch.Tensor, b: torch.Tensor,
                           pool: int = 1, act: Optional[str] = None) -> torch.Tensor:
    """Half-precision PeriodicPadding(k-1) -> Conv2d(k) -> [MaxPool2d] -> [act] (include/l2q.h:
    l2q_conv_gemm_periodic_h).  x: fp32 or 16-bit, [nb, C, H, W] ('nchw') or [nb, H, W, C]
    ('nhwc'); w16: 16-bit weight, [cout, C, k, k] for 'nchw' input, [cout, k, k, C] for 'nhwc'
    input; b fp32.  Returns NHWC 16-bit."""
    x = x.contiguous()
    hd = w16.dtype
    if layout == 'nchw':
        nb, C, H, W = x.shape
        sn, sc, sh, sw = C * H * W, H * W, W, 1
        cout, cin, k, _ = w16.shape
    else:
        nb, H, W, C = x.shape
        sn, sc, sh, sw = H * W * C, 1, W * C, C
        cout, k, _, cin = w16.shape
    if cin != C or hd not in HALF_TYPES or x.dtype not in (hd, torch.float32):
        raise N.L2QError(f'conv2d_periodic_gemm_h: x {tuple(x.shape)} {x.dtype}, w {tuple(w16.shape)} {hd}')
    Ho, Wo = H + k - 1, W + k - 1
    pool = max(int(pool), 1)
    if pool == 2 and FUSE_CONV_POOL_H[0] and Ho >= 2 and Wo >= 2:
        # conv + MaxPool2d(2) + activation in one kernel: the un-pooled image never reaches HBM
        out = torch.empty((nb, Ho // 2, Wo // 2, cout), dtype=hd, device=x.device)
        N.call('l2q_conv_pool_gemm_periodic_h', HALF_TYPES[hd], x, int(x.dtype == torch.float32), sn,
               sc, sh, sw, nb, C, H, W, k, w16.reshape(cout, -1).contiguous(), int(layout != 'nchw'),
               b.contiguous(), cout, N.ACT[act], out)
        return out
    y = torch.empty((nb * Ho * Wo, cout), dtype=hd, device=x.device)
    N.call('l2q_conv_gemm_periodic_h', HALF_TYPES[hd], x, int(x.dtype == torch.float32), sn, sc, sh,
           sw, nb, C, H, W, k, w16.reshape(cout, -1).contiguous(), int(layout != 'nchw'),
           b.contiguous(), cout, N.ACT[None if pool > 1 else act], y)
    if pool == 1:
        return y.reshape(nb, Ho, Wo, cout)
    out = torch.empty((nb, Ho // pool, Wo // pool, cout), dtype=hd, device=x.device)
    N.call('l2q_maxpool_act_nhwc_h', HALF_TYPES[hd], y, nb, Ho, Wo, cout, pool, N.ACT[act], out)
    return out


def _fused_net_args(fw: dict):
    """Network part of the fused-kernel argument list, marshalled once per weight version (the
    tensors stay referenced by `fw`; device pointers are passed as plain ints)."""
    args = fw.get('_args')
    if args is None:
        hs = fw['heads']
        # pre-convert to device pointers (plain ints) on the GPU; tensors otherwise, so that the
        # missing-GPU error still comes from native.call
        p = N.ptr if fw['wxT'].is_cuda else (lambda t: t)
        args = (p(fw['wxT']), p(fw['wvT']), p(fw['b0']), p(fw['hidden']), fw['units_c'], fw['nl'],
                p(hs['s'][0]), p(hs['s'][1]), p(hs['s'][2]), p(hs['t'][0]), p(hs['t'][1]),
                float(fw['scale_t']), p(hs['q'][0]), p(hs['q'][1]), p(hs['q'][2]),
                N.ACT[fw['act']])
        fw['_args'] = args
    return args


def u1_vstep_(x: torch.Tensor, v: torch.Tensor, beta: float, eps: float, forward: bool,
              lat: Sequence[int], fw: dict, acc: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Fused force + vnet + momentum update (v in place); returns logdet [nb] (added into
    `acc` when given).  fw: the 'fused_u1' entry of LeapfrogLayer.kernel_weights()."""
    T, X = (int(i) for i in lat)
    nb = x.shape[0]
    logdet = torch.empty(nb, dtype=torch.float32, device=x.device) if acc is None else acc
    N.call('l2q_u1_vstep_f32', x, v, float(beta), float(eps), int(forward), nb, T, X,
           *_fused_net_args(fw), int(acc is not None), logdet)
    return logdet


def u1_xstep_(x: torch.Tensor, v: torch.Tensor, mask: torch.Tensor, complement: bool, eps: float,
              forward: bool, use_ncp: bool, fw: dict,
              acc: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Fused masked cos/sin + xnet + position update (x in place); returns logdet [nb] (added
    into `acc` when given)."""
    nb = x.shape[0]
    n = x.numel() // nb
    logdet = torch.empty(nb, dtype=torch.float32, device=x.device) if acc is None else acc
    N.call('l2q_u1_xstep_f32', x, v, mask, int(complement), float(eps), int(forward),
           int(use_ncp), nb, n, *_fused_net_args(fw), int(acc is not None), logdet)
    return logdet


def u1_fused_max_n() -> int:
    return int(N.load().l2q_u1_fused_max_n())


# ---------------------------------------------------------------------------- training (VJPs)
def act_fwd(x: torch.Tensor, act: Optional[str]) -> torch.Tensor:
    """act(x) (new tensor)."""
    y = torch.empty_like(x)
    N.call('l2q_act_fwd', x.contiguous(), N.ACT[act], x.numel(), x.element_size(), y)
    return y


def act_bwd(dy: torch.Tensor, y: torch.Tensor, act: Optional[str],
            from_preact: bool = False) -> torch.Tensor:
    """dy * act'(z) in place on dy.  `y` is the activation OUTPUT for every activation whose
    derivative can be written in terms of it; swish needs the PRE-activation z, and the caller must
    say so (`from_preact=True`) -- a post-activation passed for swish would give a silently wrong
    gradient, so that combination raises."""
    if N.ACT[act] == 0:
        return dy
    if (act == 'swish') != bool(from_preact):
        raise ValueError("act_bwd: swish differentiates from the pre-activation (from_preact=True), "
                         "every other activation from its output")
    N.call('l2q_act_bwd', dy, y, N.ACT[act], dy.numel(), dy.element_size(), dy)
    return dy


def mul(a: torch.Tensor, b: torch.Tensor, alpha: float = 1.0,
        out: Optional[torch.Tensor] = None) -> torch.Tensor:
    out = torch.empty_like(a) if out is None else out
    N.call('l2q_mul', a, b, float(alpha), a.numel(), a.element_size(), out)
    return out


def axpy_rows_(y: torch.Tensor, a: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """y[c] += a[c] * x[c]"""
    nb = y.shape[0]
    N.call('l2q_axpy_rows', x, a.to(x.dtype).contiguous(), nb, y.numel() // nb, y.element_size(), y)
    return y


def add_(y: torch.Tensor, x: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
    """y += alpha * x (float32 / float64 / complex128 as pairs of doubles, any shape)."""
    if y.is_complex():
        N.call('l2q_axpy', torch.view_as_real(x), float(alpha), torch.view_as_real(y),
               2 * y.numel(), y.element_size() // 2)
        return y
    N.call('l2q_axpy', x, float(alpha), y, y.numel(), y.element_size())
    return y


def colsum_(out: torch.Tensor, a: torch.Tensor, b: Optional[torch.Tensor] = None,
            alpha: float = 1.0, accumulate: bool = True) -> torch.Tensor:
    """out[n] (+)= alpha * sum_m a[m, n] * (b[m, n] | 1)"""
    m, n = a.shape
    ws = N.workspace(int(N.load().l2q_colsum_ws_bytes(m, n)), a.device)
    N.call('l2q_colsum', a, b, m, n, float(alpha), int(accumulate), a.element_size(), out, ws,
           ws.numel())
    return out


def scaled_tanh_bwd(ds: torch.Tensor, s: Optional[torch.Tensor], coeff: Optional[torch.Tensor],
                    scale: float) -> torch.Tensor:
    m, n = ds.shape
    dpre = torch.empty_like(ds)
    N.call('l2q_scaled_tanh_bwd', ds, s, coeff, float(scale), m, n, ds.element_size(), dpre)
    return dpre


def scaled_tanh_bwd_sums(ds: torch.Tensor, s: Optional[torch.Tensor], coeff: Optional[torch.Tensor],
                         scale: float, bgrad: torch.Tensor, cgrad: Optional[torch.Tensor],
                         out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """scaled_tanh_bwd + `bgrad += colsum(dpre)` + (with coeff) `cgrad += colsum(ds * s)` in one pass.
    out: where dpre goes (a contiguous [m, n] slice of the tape's cotangent arena)."""
    m, n = ds.shape
    dpre = torch.empty_like(ds) if out is None else out
    if dpre.shape != ds.shape or dpre.dtype != ds.dtype or not dpre.is_contiguous():
        raise N.L2QError('scaled_tanh_bwd_sums: bad output buffer')
    ws = N.workspace(2 * int(N.load().l2q_colsum_ws_bytes(m, n)), ds.device)
    N.call('l2q_scaled_tanh_bwd_sums', ds, s, coeff, float(scale), m, n, ds.element_size(), dpre, bgrad,
           cgrad, ws, ws.numel())
    return dpre


def bn_train_fwd(x: torch.Tensor, gamma, beta, eps: float, momentum: float, running_mean,
                 running_var):
    m, n = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(n, dtype=x.dtype, device=x.device)
    invstd = torch.empty(n, dtype=x.dtype, device=x.device)
    N.call('l2q_bn_train_fwd', x, gamma, beta, float(eps), float(momentum), running_mean,
           running_var, m, n, x.element_size(), y, mean, invstd)
    return y, mean, invstd


def bn_bwd(dy, x, mean, invstd, gamma, dgamma, dbeta) -> torch.Tensor:
    m, n = x.shape
    dx = torch.empty_like(x)
    N.call('l2q_bn_bwd', dy, x, mean, invstd, gamma, m, n, x.element_size(), dx, dgamma, dbeta)
    return dx


def u1_force_bwd_(dx: torch.Tensor, x: torch.Tensor, dF: torch.Tensor, beta: float,
                  lat: Sequence[int]) -> torch.Tensor:
    T, X = (int(i) for i in lat)
    N.call('l2q_u1_force_bwd', x, dF, float(beta), x.shape[0], T, X, x.element_size(), dx)
    return dx


def u1_plaq_bwd_(dx: torch.Tensor, x: torch.Tensor, gcos: Optional[torch.Tensor],
                 gsin: Optional[torch.Tensor], lat: Sequence[int]) -> torch.Tensor:
    T, X = (int(i) for i in lat)
    gcos = None if gcos is None else gcos.to(x.dtype).contiguous()
    gsin = None if gsin is None else gsin.to(x.dtype).contiguous()
    N.call('l2q_u1_plaq_bwd', x, gcos, gsin, x.shape[0], T, X, x.element_size(), dx)
    return dx


def u1_x_update_bwd(x, v, s, t, q, mask, complement: bool, eps: float, forward: bool,
                    use_ncp: bool, gx, gl, dv):
    """-> (dx, ds, dt, dq, deps[nb]); dv accumulated in place."""
    nb = x.shape[0]
    n = x.numel() // nb
    dx, ds, dt, dq = (torch.empty_like(s) for _ in range(4))
    deps = torch.empty(nb, dtype=x.dtype, device=x.device)
    N.call('l2q_u1_x_update_bwd', x, v, s, t, q, mask, int(complement), float(eps), int(forward),
           int(use_ncp), gx, gl, x.element_size(), nb, n, dx, dv, ds, dt, dq, deps)
    return dx, ds, dt, dq, deps


def v_update_bwd(v, force, s, t, q, eps: float, forward: bool, gv, gl):
    """-> (dv, dF, ds, dt, dq, deps[nb])"""
    nb = v.shape[0]
    n = v.numel() // nb
    dv, dF, ds, dt, dq = (torch.empty_like(s) for _ in range(5))
    deps = torch.empty(nb, dtype=v.dtype, device=v.device)
    N.call('l2q_v_update_bwd', v, force, s, t, q, float(eps), int(forward), gv, gl,
           v.element_size(), nb, n, dv, dF, ds, dt, dq, deps)
    return dv, dF, ds, dt, dq, deps


def u1_masked_cos_sin_bwd_(dx, x, mask, complement: bool, dout) -> torch.Tensor:
    nb = x.shape[0]
    N.call('l2q_u1_masked_cos_sin_bwd', x, mask, int(complement), dout, nb, x.numel() // nb,
           x.element_size(), dx)
    return dx


def conv2d_periodic_gemm_train(x: torch.Tensor, layout: str, w: torch.Tensor, b: torch.Tensor,
                               pool: int = 1, act: Optional[str] = None,
                               half: Optional[torch.dtype] = None):
    """conv2d_periodic_gemm that also returns what the backward pass needs.
    half = float16 | bfloat16: the layer as torch.autocast runs it in the reference's train step -- input,
    weight and bias rounded to that type, fp32 accumulation (products of 16-bit values are exact in fp32:
    the fp32 MFMA GEMM gives what the 16-bit one gives, to summation order), the convolution's output rounded,
    the activation's output rounded; everything stays in fp32 containers for the backward pass."""
    x = x.contiguous()
    if half is not None:
        r16 = lambda t: t.to(half).float()
        x, w, b = r16(x), r16(w), r16(b)
    if layout == 'nchw':
        nb, C, H, W = x.shape
        strides = (C * H * W, H * W, W, 1)
    else:
        nb, H, W, C = x.shape
        strides = (H * W * C, 1, W * C, C)
    cout, cin, k, _ = w.shape
    Ho, Wo, Kc = H + k - 1, W + k - 1, C * k * k
    # NHWC input: K columns in (i, j, ci) order, so that im2col, the GEMM and col2im all move
    # contiguous runs of channels (col2im was 35 % of the conv training step in (ci, i, j) order)
    clast = layout != 'nchw'
    col = torch.empty((nb * Ho * Wo, Kc), dtype=torch.float32, device=x.device)
    N.call('l2q_im2col_periodic_f32', x, *strides, nb, C, H, W, k, int(clast), col)
    pool = max(int(pool), 1)
    wk = (w.permute(0, 2, 3, 1) if clast else w).reshape(cout, Kc).contiguous()
    if half is not None and pool == 1 and act is not None:
        # rounding point between the convolution and its activation
        y = act_fwd(gemm(col, wk, b.contiguous(), act=None).to(half).float(), act).to(half).float()
    else:
        y = gemm(col, wk, b.contiguous(), act=None if pool > 1 else act)
        if half is not None:
            y = y.to(half).float()
    ctx = {'col': col, 'strides': strides, 'dims': (nb, C, H, W, k, cout), 'pool': pool,
           'act': act, 'y': y, 'clast': clast, 'wk': wk}
    if pool == 1:
        out = y.reshape(nb, Ho, Wo, cout)
    else:
        out = torch.empty((nb, Ho // pool, Wo // pool, cout), dtype=torch.float32, device=x.device)
        N.call('l2q_maxpool_act_nhwc_f32', y, nb, Ho, Wo, cout, pool, N.ACT[act], out)
        if half is not None:
            out = out.to(half).float()
    ctx['out'] = out
    return out, ctx


def conv2d_periodic_gemm_bwd(ctx: dict, dout: torch.Tensor, w: torch.Tensor, dw: torch.Tensor,
                             db: torch.Tensor, need_dx: bool = True):
    """dw, db accumulated (+=); returns dx in the layout / strides of the forward input."""
    nb, C, H, W, k, cout = ctx['dims']
    Ho, Wo, Kc = H + k - 1, W + k - 1, C * k * k
    pool, act = ctx['pool'], ctx['act']
    if act == 'swish':
        # the fused conv kernels keep the activation OUTPUT only; swish' needs the pre-activation
        raise NotImplementedError('conv backward with swish: the conv tape holds post-activations')
    dout = dout.contiguous()
    if pool > 1:
        dy = torch.empty_like(ctx['y'])
        N.call('l2q_maxpool_act_nhwc_bwd_f32', dout, ctx['out'], ctx['y'], nb, Ho, Wo, cout, pool,
               N.ACT[act], dy)
    else:
        dy = act_bwd(dout.reshape(nb * Ho * Wo, cout).clone(), ctx['y'], act)
    dy = dy.reshape(nb * Ho * Wo, cout)
    colsum_(db, dy)
    clast = ctx.get('clast', False)
    dwk = gemm(t2d(dy), t2d(ctx['col']))                             # dW = dy^T col, [cout, Kc]
    if clast:                                                        # (i, j, ci) -> (ci, i, j)
        dw.add_(dwk.reshape(cout, k, k, C).permute(0, 3, 1, 2))
    else:
        add_(dw, dwk.reshape(dw.shape))
    if not need_dx:
        return None
    dcol = gemm(dy, t2d(ctx['wk'] if 'wk' in ctx else w.reshape(cout, Kc)))    # [M, Kc]
    sn, sc, sh, sw = ctx['strides']
    shape = (nb, C, H, W) if sw == 1 else (nb, H, W, C)
    dx = torch.empty(shape, dtype=torch.float32, device=dout.device)
    N.call('l2q_col2im_periodic_f32', dcol, sn, sc, sh, sw, nb, C, H, W, k, int(clast), dx)
    return dx


def adam_step_(p, g, m, v, lr: float, beta1: float, beta2: float, eps: float, step: int,
               grad_scale: float = 1.0) -> None:
    N.call('l2q_adam', p, g, m, v, p.numel(), float(lr), float(beta1), float(beta2), float(eps),
           int(step), float(grad_scale), p.element_size())


def sumsq(a: torch.Tensor) -> torch.Tensor:
    out = torch.empty(1, dtype=torch.float64, device=a.device)
    ws = N.workspace(int(N.load().l2q_sumsq_ws_bytes(a.numel())), a.device)
    N.call('l2q_sumsq', a, a.numel(), a.element_size(), out, ws, ws.numel())
    return out


# ---------------------------------------------------------------------------- SU(3) training (VJPs)
def su3_expm_mul_bwd_n(xn, vn, eps: float, mask_n, complement: bool, gxnew, gv):
    """-> (gx, deps[nb]); gv accumulated in place.  eps: the signed step of the forward call."""
    nb, _, _, V = xn.shape
    gx = torch.empty_like(xn)
    deps = torch.empty(nb, dtype=torch.float64, device=xn.device)
    ws = N.workspace(nb * 4 * ((V + 255) // 256) * 8, xn.device)
    N.call('l2q_su3_expm_mul_bwd', xn, vn, float(eps), mask_n, int(complement), gxnew, gx, gv, deps,
           nb, V, ws, ws.numel())
    return gx, deps


def su3_expm_mul2_bwd_n(xn, vn, eps: float, mask_n, complement_first: bool, gxnew, gv):
    """VJP of su3_expm_mul2_n (both half-updates, one Frechet derivative): -> (gx, deps[nb]); gv += in place."""
    nb, _, _, V = xn.shape
    gx = torch.empty_like(xn)
    deps = torch.empty(nb, dtype=torch.float64, device=xn.device)
    ws = N.workspace(nb * 4 * ((V + 255) // 256) * 8, xn.device)
    N.call('l2q_su3_expm_mul2_bwd', xn, vn, float(eps), mask_n, int(complement_first), gxnew, gx, gv, deps,
           nb, V, ws, ws.numel())
    return gx, deps


def su3_projsu_vec8_bwd_(gm: torch.Tensor, mn: torch.Tensor, gvec: torch.Tensor) -> torch.Tensor:
    """gm += VJP of su3_projsu_vec8_n at mn for the cotangent gvec [..., 8, V]."""
    V = mn.shape[-1]
    nf = mn.numel() // (9 * V)
    N.call('l2q_su3_projsu_vec8_bwd', mn, gvec.contiguous(), gm, nf, V)
    return gm


def su3_force_bwd_(gx: torch.Tensor, xn: torch.Tensor, gf: torch.Tensor, beta: float,
                   lat: Sequence[int]) -> torch.Tensor:
    T, X, Y, Z = (int(i) for i in lat)
    N.call('l2q_su3_force_bwd', xn, gf, float(beta), gx, xn.shape[0], T, X, Y, Z)
    return gx


def su3_plaq_bwd_(gx: torch.Tensor, xn: torch.Tensor, w: torch.Tensor,
                  lat: Sequence[int]) -> torch.Tensor:
    """w [nb, 6, 2] float64: complex plane weights (re, im)."""
    T, X, Y, Z = (int(i) for i in lat)
    N.call('l2q_su3_plaq_bwd', xn, w.to(torch.float64).contiguous(), gx, xn.shape[0], T, X, Y, Z)
    return gx


def su3_wilson_loops_bwd_(gx: torch.Tensor, xn: torch.Tensor, w: torch.Tensor,
                          lat: Sequence[int]) -> torch.Tensor:
    """w [6, nb, T, X, Y, Z] complex128: the cotangent of `su3_wilson_loops_n`'s output."""
    T, X, Y, Z = (int(i) for i in lat)
    N.call('l2q_su3_wilson_loops_bwd', xn, w.to(C128).contiguous(), gx, xn.shape[0], T, X, Y, Z)
    return gx


def su3_rect_sums_n(xn: torch.Tensor, lat: Sequence[int]) -> torch.Tensor:
    """[nb]: sum over sites and the 12 planar 2x1 loops of Re tr R (c1 != 0 actions)."""
    nb = xn.shape[0]
    T, X, Y, Z = (int(i) for i in lat)
    out = torch.empty(nb, dtype=torch.float64, device=xn.device)
    ws = _ws(nb, T * X * Y * Z * 36, xn.device)
    N.call('l2q_su3_rect_reduce', xn, nb, T, X, Y, Z, out, ws, ws.numel())
    return out


def su3_rect_force_add_n(xn: torch.Tensor, coef: float, fn: torch.Tensor,
                         lat: Sequence[int]) -> torch.Tensor:
    """fn += coef * TAH(U * rectangle staples) in place."""
    T, X, Y, Z = (int(i) for i in lat)
    N.call('l2q_su3_rect_force_add', xn, float(coef), fn, xn.shape[0], T, X, Y, Z)
    return fn


def su3_rect_bwd_(gx: torch.Tensor, xn: torch.Tensor, w: torch.Tensor,
                  lat: Sequence[int]) -> torch.Tensor:
    """gx += w[c] * d(sum_R Re tr R)/dx;  w [nb] float64."""
    T, X, Y, Z = (int(i) for i in lat)
    N.call('l2q_su3_rect_bwd', xn, w.to(torch.float64).contiguous(), gx, xn.shape[0], T, X, Y, Z)
    return gx


def v_update_bwd_c128(v, force, s, t, q, eps: float, forward: bool, gv, gl, acc=None):
    """complex momenta: -> (dv, dF, ds, dt, dq, deps[nb]).  acc = (aF, as, at, aq): cotangents added to
    (dF, ds, dt, dq) in the same pass (the v-update that shared this one's force and heads)."""
    nb = v.shape[0]
    n = v.numel() // nb
    dv, dF = torch.empty_like(v), torch.empty_like(v)
    ds, dt, dq = (torch.empty_like(s) for _ in range(3))
    deps = torch.empty(nb, dtype=torch.float64, device=v.device)
    ws = N.workspace(nb * ((n + 255) // 256) * 8, v.device)
    if acc is not None:
        aF, as_, at, aq = acc
        N.call('l2q_v_update_bwd_acc_c128', v, force, s, t, q, float(eps), int(forward), gv, gl, nb, n,
               aF, as_, at, aq, dv, dF, ds, dt, dq, deps, ws, ws.numel())
    else:
        N.call('l2q_v_update_bwd_c128', v, force, s, t, q, float(eps), int(forward), gv, gl, nb, n, dv,
               dF, ds, dt, dq, deps, ws, ws.numel())
    return dv, dF, ds, dt, dq, deps


def v_update_bwd_pair_c128(v1, v_mid, force, s, t, q, eps1: float, forward1: bool, eps2: float, forward2: bool,
                           flip: bool, gv, gl):
    """Two v-updates that share (force, s, t, q), reversed in one pass (include/l2q.h:
    l2q_v_update_bwd_pair_c128): -> (dv, dF, ds, dt, dq, deps1[nb], deps2[nb])."""
    nb = v1.shape[0]
    n = v1.numel() // nb
    dv, dF = torch.empty_like(v1), torch.empty_like(v1)
    ds, dt, dq = (torch.empty_like(s) for _ in range(3))
    deps1 = torch.empty(nb, dtype=torch.float64, device=v1.device)
    deps2 = torch.empty_like(deps1)
    ws = N.workspace(2 * nb * ((n + 255) // 256) * 8, v1.device)
    N.call('l2q_v_update_bwd_pair_c128', v1, v_mid, force, s, t, q, float(eps1), int(forward1), float(eps2),
           int(forward2), int(flip), gv, gl, nb, n, dv, dF, ds, dt, dq, deps1, deps2, ws, ws.numel())
    return dv, dF, ds, dt, dq, deps1, deps2


def diff_bwd_(gx: torch.Tensor, x: torch.Tensor, y: torch.Tensor, a: torch.Tensor) -> torch.Tensor:
    """gx += 2 a[c] (x - y) over float64 / complex128 tensors of equal shape."""
    nb = x.shape[0]
    n = x.numel() // nb * (2 if x.is_complex() else 1)
    N.call('l2q_diff_bwd_f64', x, y, a.to(torch.float64).contiguous(), nb, n, gx)
    return gx
