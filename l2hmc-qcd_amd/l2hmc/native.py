"""ctypes binding of ``libl2q.so`` (C ABI: ``include/l2q.h``).

The product path has NO fallback: if the library is missing, or a tensor is not a contiguous
CUDA/HIP tensor, the call raises.  All launches go on PyTorch's current stream so that
``torch.cuda.Event`` timing and stream ordering with the rest of the program hold.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, '_lib', os.environ.get('L2Q_LIB_NAME', 'libl2q.so'))

ACT = {None: 0, 'none': 0, 'tanh': 1, 'relu': 2, 'leaky_relu': 3, 'elu': 4, 'swish': 5}

P, I, L, D, F, Z = C.c_void_p, C.c_int, C.c_long, C.c_double, C.c_float, C.c_size_t

# name -> (restype, argtypes); mirrors include/l2q.h one to one
SIGNATURES = {
    'l2q_last_error': (C.c_char_p, []),
    'l2q_version': (I, []),
    'l2q_set_tuning': (I, [C.c_char_p, I]),
    'l2q_kernel_name': (I, [C.c_char_p, I, I, I, I, C.c_char_p, Z]),
    'l2q_init': (I, [I]),
    'l2q_comm_unique_id': (I, [P]),
    'l2q_comm_init': (I, [P, I, I, C.POINTER(P)]),
    'l2q_allreduce_grads': (I, [P, P, L, I, P]),
    'l2q_comm_destroy': (I, [P]),
    'l2q_comm_abort': (I, [P]),
    'l2q_comm_version': (I, []),
    'l2q_reduce_ws_bytes': (Z, [I, L]),
    'l2q_transpose': (I, [P, P, L, I, I, I, P]),
    'l2q_su3_pack': (I, [P, P, I, L, P]),
    'l2q_su3_unpack': (I, [P, P, I, L, P]),
    'l2q_su3_unpack_select': (I, [P, P, P, P, I, L, P]),
    'l2q_su3_plaq_reduce': (I, [P, I, I, I, I, I, P, P, Z, P]),
    'l2q_su3_plaq_planes': (I, [P, I, I, I, I, I, P, P, Z, P]),
    'l2q_su3_wilson_loops': (I, [P, I, I, I, I, I, P, P]),
    'l2q_diff_norm2_reduce': (I, [P, P, I, L, P, P, Z, P]),
    'l2q_su3_force': (I, [P, D, P, I, I, I, I, I, P]),
    'l2q_su3_force_kick': (I, [P, D, D, P, I, I, I, I, I, P]),
    'l2q_su3_force_kick_to': (I, [P, D, D, P, P, I, I, I, I, I, P]),
    'l2q_su3_expm_mul': (I, [P, P, D, P, I, P, I, L, P]),
    'l2q_su3_expm_mul2': (I, [P, P, D, P, I, P, I, L, P]),
    'l2q_su3_expm_mul2_vec8': (I, [P, P, D, P, I, P, P, I, L, P]),
    'l2q_su3_project_su': (I, [P, P, L, L, P]),
    'l2q_su3_projsu_vec8': (I, [P, P, L, L, P]),
    'l2q_su3_project_tah': (I, [P, P, L, L, P]),
    'l2q_su3_project_u': (I, [P, P, L, L, P]),
    'l2q_su3_mul': (I, [P, P, I, I, P, L, L, P]),
    'l2q_su3_kinetic_reduce': (I, [P, I, L, P, P, Z, P]),
    'l2q_su3_assemble_tah': (I, [P, P, L, L, P]),
    'l2q_su3_check_su': (I, [P, I, L, P, P, Z, P]),
    'l2q_v_update': (I, [P, P, P, P, P, D, I, I, I, I, L, P, P, Z, P]),
    'l2q_v_update_to': (I, [P, P, P, P, P, P, D, I, I, I, I, L, P, P, Z, P]),
    'l2q_accept': (I, [P, P, P, P, P, P, I, I, P]),
    'l2q_select_rows': (I, [P, P, P, P, I, L, P]),
    'l2q_scale_f64': (I, [P, D, P, L, P]),
    'l2q_gemm_f64': (I, [P, P, I, I, L, P, P, L, P, P, P, D, I, P, P, Z, P]),
    'l2q_gemm_ws_bytes': (Z, [I, I, L, L]),
    'l2q_gemm_sliced_bytes': (Z, [I, L]),
    'l2q_gemm_sliced_build': (I, [P, I, L, P, Z, P, P]),
    'l2q_gemm_sliced_ws_bytes': (Z, [I, I, L, L]),
    'l2q_gemm_sliced_f64': (I, [P, P, L, I, P, P, L, I, I, I, P, P, P, D, I, P, P, Z, P]),
    'l2q_vnet_heads_vupdate_f64': (I, [P, I, I, L, P, P, P, D, P, P, D, P, P, P, D, P, P, I, D, I, P, P, Z, P]),
    'l2q_vnet_heads_vupdate_to_f64': (I, [P, I, I, L, P, P, P, D, P, P, D, P, P, P, D, P, P, P, I, D, I, P, P, Z, P]),
    'l2q_vnet_heads_vupdate_pair_f64': (I, [P, I, I, L, P, P, P, D, P, P, D, P, P, P, D, P, P, I, D, I, I, D, I, P, P, Z, P]),
    'l2q_vnet_heads_vupdate_pair_mid_f64': (I, [P, I, I, L, P, P, P, D, P, P, D, P, P, P, D, P, P, I, D, I, I, D, I, P, P, P, P, Z, P]),
    'l2q_vnet_heads_ws_bytes': (Z, [I, L]),
    'l2q_heads_sliced_bytes': (Z, [I, L]),
    'l2q_heads_sliced_build': (I, [P, P, P, I, L, P, Z, P, P]),
    'l2q_heads_sliced_zflag': (I, [I, P, P]),
    'l2q_vnet_heads_sliced_ws_bytes': (Z, [I, L]),
    'l2q_vnet_heads_vupdate_sliced_f64': (I, [P, I, I, L, P, P, P, D, P, D, P, P, D, P, P, P, I, D, I, I, I, D,
                                              I, P, P, P, P, Z, P]),
    'l2q_vnet_heads_vupdate_sliced_tape_f64': (I, [P, I, I, L, P, P, P, P, D, P, P, P, P, P, I, D, I, P, P, P, P,
                                                   P, Z, P]),
    'l2q_gemm_f32': (I, [P, P, I, I, L, P, P, L, P, P, P, F, I, P, P, Z, P]),
    'l2q_gemm_ex': (I, [P, I, P, I, I, I, L, I, I, P, P, Z, P]),
    'l2q_gemm_h': (I, [I, P, I, P, I, I, L, P, P, L, P, P, P, F, I, P, I, P, Z, P]),
    'l2q_gemm_h_ws_bytes': (Z, [I, I, L, L]),
    'l2q_gemm_h_skinny_splits': (I, [I, I, L, L, I]),
    'l2q_gemm_h_u1x': (I, [I, P, P, I, P, I, I, L, P, P, L, P, P, I, P, P, Z, P]),
    'l2q_u1_heads_update_h': (I, [I, P, I, I, L, P, P, P, P, P, F, P, P, P, I, P, P, P, I, F, I, I, P,
                                  I, P, Z, P]),
    'l2q_u1_heads_update_h_ws_bytes': (Z, [I, L]),
    'l2q_conv_gemm_periodic_h': (I, [I, P, I, L, L, L, L, I, I, I, I, I, P, I, P, I, I, P, P]),
    'l2q_conv_pool_gemm_periodic_h': (I, [I, P, I, L, L, L, L, I, I, I, I, I, P, I, P, I, I, P, P]),
    'l2q_maxpool_act_nhwc_h': (I, [I, P, I, I, I, I, I, I, P, P]),
    'l2q_nchw_to_nhwc_pad_h': (I, [I, P, I, I, I, I, I, P, P]),
    'l2q_u1_plaq_reduce': (I, [P, I, I, I, I, P, P]),
    'l2q_u1_wilson_loops': (I, [P, I, I, I, I, P, P]),
    'l2q_u1_wilson_loops_bwd': (I, [P, I, I, I, I, P, P]),
    'l2q_u1_force': (I, [P, D, P, P, D, I, I, I, I, P]),
    'l2q_u1_x_update': (I, [P, P, P, P, P, P, I, D, I, I, I, I, L, P, P]),
    'l2q_u1_wrap': (I, [P, P, L, I, P]),
    'l2q_u1_kinetic_reduce': (I, [P, I, L, I, P, P]),
    'l2q_axpy': (I, [P, D, P, L, I, P]),
    'l2q_u1_masked_cos_sin': (I, [P, P, I, P, I, L, I, P]),
    'l2q_conv2d_periodic_f32': (I, [P, P, P, P, I, I, I, I, I, I, I, I, P]),
    'l2q_im2col_periodic_f32': (I, [P, L, L, L, L, I, I, I, I, I, I, P, P]),
    'l2q_maxpool_act_nhwc_f32': (I, [P, I, I, I, I, I, I, P, P]),
    'l2q_conv_gemm_periodic_f32': (I, [P, L, L, L, L, I, I, I, I, I, P, I, P, I, I, P, P]),
    'l2q_nchw_to_nhwc_pad_f32': (I, [P, I, I, I, I, I, P, P]),
    'l2q_u1_fused_max_n': (I, []),
    'l2q_u1_vstep_f32': (I, [P, P, D, D, I, I, I, I, P, P, P, P, P, I, P, P, P, P, P, D, P, P, P, I, I, P, P]),
    'l2q_u1_xstep_f32': (I, [P, P, P, I, D, I, I, I, I, P, P, P, P, P, I, P, P, P, P, P, D, P, P, P, I, I, P, P]),
    'l2q_act_fwd': (I, [P, I, L, I, P, P]),
    'l2q_act_bwd': (I, [P, P, I, L, I, P, P]),
    'l2q_mul': (I, [P, P, D, L, I, P, P]),
    'l2q_axpy_rows': (I, [P, P, I, L, I, P, P]),
    'l2q_colsum': (I, [P, P, L, I, D, I, I, P, P, Z, P]),
    'l2q_colsum_ws_bytes': (Z, [L, I]),
    'l2q_scaled_tanh_bwd': (I, [P, P, P, D, I, I, I, P, P]),
    'l2q_scaled_tanh_bwd_sums': (I, [P, P, P, D, I, I, I, P, P, P, P, Z, P]),
    'l2q_bn_train_fwd': (I, [P, P, P, D, D, P, P, I, I, I, P, P, P, P]),
    'l2q_bn_bwd': (I, [P, P, P, P, P, I, I, I, P, P, P, P]),
    'l2q_col2im_periodic_f32': (I, [P, L, L, L, L, I, I, I, I, I, I, P, P]),
    'l2q_maxpool_act_nhwc_bwd_f32': (I, [P, P, P, I, I, I, I, I, I, P, P]),
    'l2q_u1_force_bwd': (I, [P, P, D, I, I, I, I, P, P]),
    'l2q_u1_plaq_bwd': (I, [P, P, P, I, I, I, I, P, P]),
    'l2q_u1_x_update_bwd': (I, [P, P, P, P, P, P, I, D, I, I, P, P, I, I, L, P, P, P, P, P, P, P]),
    'l2q_v_update_bwd': (I, [P, P, P, P, P, D, I, P, P, I, I, L, P, P, P, P, P, P, P]),
    'l2q_u1_masked_cos_sin_bwd': (I, [P, P, I, P, I, L, I, P, P]),
    'l2q_adam': (I, [P, P, P, P, L, D, D, D, D, L, D, I, P]),
    'l2q_sumsq': (I, [P, L, I, P, P, Z, P]),
    'l2q_sumsq_ws_bytes': (Z, [L]),
    'l2q_su3_expm_mul_bwd': (I, [P, P, D, P, I, P, P, P, P, I, L, P, Z, P]),
    'l2q_su3_expm_mul2_bwd': (I, [P, P, D, P, I, P, P, P, P, I, L, P, Z, P]),
    'l2q_su3_projsu_vec8_bwd': (I, [P, P, P, L, L, P]),
    'l2q_su3_force_bwd': (I, [P, P, D, P, I, I, I, I, I, P]),
    'l2q_su3_plaq_bwd': (I, [P, P, P, I, I, I, I, I, P]),
    'l2q_su3_wilson_loops_bwd': (I, [P, P, P, I, I, I, I, I, P]),
    'l2q_su3_rect_reduce': (I, [P, I, I, I, I, I, P, P, Z, P]),
    'l2q_su3_rect_force_add': (I, [P, D, P, I, I, I, I, I, P]),
    'l2q_su3_rect_bwd': (I, [P, P, P, I, I, I, I, I, P]),
    'l2q_v_update_bwd_c128': (I, [P, P, P, P, P, D, I, P, P, I, L, P, P, P, P, P, P, P, Z, P]),
    'l2q_v_update_bwd_pair_c128': (I, [P, P, P, P, P, P, D, I, D, I, I, P, P, I, L, P, P, P, P, P, P, P, P, Z, P]),
    'l2q_v_update_bwd_acc_c128': (I, [P, P, P, P, P, D, I, P, P, I, L, P, P, P, P, P, P, P, P, P, P, P, Z, P]),
    'l2q_diff_bwd_f64': (I, [P, P, P, I, L, P, P]),
}

_lib: Optional[C.CDLL] = None


class L2QError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libl2q.so (once).  Raises if it has not been built -- there is no CPU path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise L2QError(
                f'{LIB_PATH} not found: build it with `python __graft_entry__.py` '
                '(hipcc --offload-arch=gfx950).  The l2hmc hot path has no CPU fallback.')
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        # L2Q_TUNING="force_tile=7,plaq_sweep=2": tuning knobs of include/l2q.h (l2q_set_tuning) for a whole process --
        # A/B runs of bench.py / the tests without code changes (the table is per device: applies to the current one)
        for kv in filter(None, os.environ.get('L2Q_TUNING', '').split(',')):
            k, _, v = kv.partition('=')
            if lib.l2q_set_tuning(k.strip().encode(), int(v)) < 0:      # (returns the previous value)
                raise L2QError(f'L2Q_TUNING: unknown knob or value out of range: {kv!r}')
    return _lib


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]):
    """Device pointer of a contiguous HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise L2QError('l2q kernels need tensors on the GPU (got a CPU tensor); '
                       'the hot path has no CPU fallback')
    if not t.is_contiguous():
        raise L2QError('l2q kernels need contiguous tensors')
    return t.data_ptr()


def call(name: str, *args):
    """Invoke an int-returning entry point; tensors are passed as device pointers and the
    current stream is appended.  Raises L2QError with l2q_last_error() on failure."""
    lib = load()
    conv = [ptr(a) if isinstance(a, torch.Tensor) else a for a in args]
    rc = getattr(lib, name)(*conv, stream_ptr())
    if rc != 0:
        raise L2QError(f'{name} failed ({rc}): {lib.l2q_last_error().decode()}')


def set_tuning(key: str, value: int) -> int:
    return load().l2q_set_tuning(key.encode(), int(value))


def kernel_name(entry: str, lat) -> str:
    """Device-kernel template `entry` dispatches for this lattice under the current tuning."""
    buf = C.create_string_buffer(256)
    T, X, Y, Zz = (int(i) for i in lat)
    rc = load().l2q_kernel_name(entry.encode(), T, X, Y, Zz, buf, 256)
    return buf.value.decode() if rc == 0 else ''


class Workspace:
    """Grow-only scratch buffer for the order-stable reductions and split-K partials.  A block
    that a captured HIP graph points at is pinned (`pin`): growing then allocates a new block
    and keeps the pinned one alive instead of handing its memory back to the allocator."""

    def __init__(self):
        self.buf: Optional[torch.Tensor] = None
        self.pinned: list = []

    def get(self, nbytes: int, device) -> torch.Tensor:
        nbytes = max(int(nbytes), 256)
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != torch.device(device):
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self.buf

    def pin(self) -> Optional[torch.Tensor]:
        if self.buf is not None and not any(b is self.buf for b in self.pinned):
            self.pinned.append(self.buf)
        return self.buf


_WS = Workspace()


def workspace(nbytes: int, device) -> torch.Tensor:
    return _WS.get(nbytes, device)


def pin_workspace() -> Optional[torch.Tensor]:
    """Keep the current workspace block alive for good (a captured HIP graph references it)."""
    return _WS.pin()


def reduce_ws_bytes(nb: int, n_per_chain: int) -> int:
    return int(load().l2q_reduce_ws_bytes(int(nb), int(n_per_chain)))


def gemm_ws_bytes(m: int, n: int, k: int, k2: int = 0) -> int:
    return int(load().l2q_gemm_ws_bytes(int(m), int(n), int(k), int(k2)))
