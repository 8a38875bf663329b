// su3_force_tile.hpp -- addressing helpers shared by the slice-resident force kernels that split a
// link's work over wavefronts (su3_force_rows.hip, su3_force_nu.hip): buffer loads with the
// uniform part of the address in the scalar offset, ds_read_b128 with immediate entry offsets.
#pragma once
#include "l2q_common.hpp"
#include "su3_math.hpp"
#include "su3_links.hpp"

namespace l2q {

constexpr int kRS = 64;                       // sites per tile
constexpr int kEnt = kRS * 16;                // bytes between entries of a link in LDS
constexpr int kPlaneB = 9 * kEnt;             // bytes of one link direction of a tile

typedef unsigned int v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double2 buf_ld(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  union { v4u u; double2 d; } c;
  c.u = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
  return c.d;
}

__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t rs, int voff, int soff, double2 v) {
  union { v4u u; double2 d; } c;
  c.d = v;
  __builtin_amdgcn_raw_buffer_store_b128(c.u, rs, voff, soff, 0);
}

// streaming store (nt): the written line is not kept in L2 / MALL ahead of data that is re-read
__device__ __forceinline__ void buf_st_nt(__amdgpu_buffer_rsrc_t rs, int voff, int soff, double2 v) {
  union { v4u u; double2 d; } c;
  c.d = v;
  __builtin_amdgcn_raw_buffer_store_b128(c.u, rs, voff, soff, 2);
}

__device__ __forceinline__ double2 buf_ld_nt(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  union { v4u u; double2 d; } c;
  c.u = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 2);
  return c.d;
}

struct R3 {
  double re[3], im[3];
};

__device__ __forceinline__ void r3_zero(R3& a) {
#pragma unroll
  for (int k = 0; k < 3; ++k) { a.re[k] = 0.0; a.im[k] = 0.0; }
}

// one periodic hop of a spatial site index q = (x*Y + y)*Z + z in direction dir (1, 2, 3)
__device__ __forceinline__ int hop(int q, int x, int y, int z, int dir, int sgn, const Dims& d) {
  const int n = dir == 1 ? d.X : dir == 2 ? d.Y : d.Z;
  const int st = dir == 1 ? d.Y * d.Z : dir == 2 ? d.Z : 1;
  const int c = dir == 1 ? x : dir == 2 ? y : z;
  if (sgn > 0) return (c + 1 == n) ? q - (n - 1) * st : q + st;
  return (c == 0) ? q + (n - 1) * st : q - st;
}

extern __shared__ __attribute__((aligned(16))) char fr_lds[];

__device__ __forceinline__ double2 lds_ld(int addr) {
  return *reinterpret_cast<const double2*>(fr_lds + addr);
}

// Operand loads.  IN (compile time): the operand's site is inside the tile for every lane ->
// ds_read_b128 from the LDS copy at byte address `lds` (entry e at + e * kEnt); otherwise a
// buffer load from the chain: per-lane site offset `voff`, uniform (link, slice) offset `soff`.
// (ENT: bytes between the entries of a link in LDS = 16 x the sites of the tile)
template <bool IN, int ENT = kEnt>
__device__ __forceinline__ void ld_full(M3& m, int lds, __amdgpu_buffer_rsrc_t rs, int voff, int soff, int V16) {
#pragma unroll
  for (int e = 0; e < 9; ++e) {
    const double2 d = IN ? lds_ld(lds + e * ENT) : buf_ld(rs, voff, soff + e * V16);
    m.re[e] = d.x; m.im[e] = d.y;
  }
}

// row `row` of the link (row is wave-uniform)
template <bool IN, int ENT = kEnt>
__device__ __forceinline__ void ld_row(R3& a, int lds, __amdgpu_buffer_rsrc_t rs, int voff, int soff, int V16, int row) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double2 d = IN ? lds_ld(lds + (3 * row + k) * ENT) : buf_ld(rs, voff, soff + (3 * row + k) * V16);
    a.re[k] = d.x; a.im[k] = d.y;
  }
}

// conj of column `col` of the link = row `col` of its adjoint
template <bool IN>
__device__ __forceinline__ void ld_colc(R3& a, int lds, __amdgpu_buffer_rsrc_t rs, int voff, int soff, int V16, int col) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double2 d = IN ? lds_ld(lds + (3 * k + col) * kEnt) : buf_ld(rs, voff, soff + (3 * k + col) * V16);
    a.re[k] = d.x; a.im[k] = -d.y;
  }
}

// Operand: LDS (IN, compile time) at byte address lds, else the chain buffer at (voff, soff).
template <bool IN, int ENT = kEnt>
struct Opnd {
  int lds, voff, soff;
};

// Row pipelining (L2Q_ROW_PIPE, default on): the next row of a streamed operand is requested BEFORE
// the current row's 36 FMAs, so that the ~100-cycle LDS latency of a row hides behind the arithmetic of
// the previous one instead of stalling the wavefront seven times per staple (ISA before: ds_read x3,
// s_waitcnt, fma ... per row; PMC: the SIMDs issued 73 % of the time with two wavefronts each).
#ifndef L2Q_ROW_PIPE
#define L2Q_ROW_PIPE 1
#endif

// t = A * B^H (ADJ_A = false) or A^H * B^H (ADJ_A = true); B streamed by rows
template <bool ADJ_A, bool IN, int ENT>
__device__ __forceinline__ void mul_xh_stream(M3& t, const M3& a, const Opnd<IN, ENT>& b,
                                              __amdgpu_buffer_rsrc_t rs, int V16) {
  R3 rows[2];
#if L2Q_ROW_PIPE
  ld_row<IN, ENT>(rows[0], b.lds, rs, b.voff, b.soff, V16, 0);
#endif
#pragma unroll
  for (int j = 0; j < 3; ++j) {
#if L2Q_ROW_PIPE
    if (j < 2) {
      ld_row<IN, ENT>(rows[(j + 1) & 1], b.lds, rs, b.voff, b.soff, V16, j + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    const R3& br = rows[j & 1];
#else
    R3& br = rows[0];
    ld_row<IN, ENT>(br, b.lds, rs, b.voff, b.soff, V16, j);
#endif
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double sr = 0.0, si = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double ar = ADJ_A ? a.re[3 * k + i] : a.re[3 * i + k];
        const double ai = ADJ_A ? -a.im[3 * k + i] : a.im[3 * i + k];
        const double xr = br.re[k], xi = -br.im[k];           // conj(B_jk)
        sr = fma(ar, xr, sr); sr = fma(-ai, xi, sr);
        si = fma(ar, xi, si); si = fma(ai, xr, si);
      }
      t.re[3 * i + j] = sr; t.im[3 * i + j] = si;
    }
  }
}

// acc += T * C^H (ADJ_C = true) or T * C (ADJ_C = false); C streamed by rows
template <bool ADJ_C, bool IN, int ENT>
__device__ __forceinline__ void mac_stream(M3& acc, const M3& t, const Opnd<IN, ENT>& c,
                                           __amdgpu_buffer_rsrc_t rs, int V16) {
  R3 rows[2];
#if L2Q_ROW_PIPE
  ld_row<IN, ENT>(rows[0], c.lds, rs, c.voff, c.soff, V16, 0);
#endif
#pragma unroll
  for (int q = 0; q < 3; ++q) {
#if L2Q_ROW_PIPE
    if (q < 2) {
      ld_row<IN, ENT>(rows[(q + 1) & 1], c.lds, rs, c.voff, c.soff, V16, q + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    const R3& cr = rows[q & 1];
#else
    R3& cr = rows[0];
    ld_row<IN, ENT>(cr, c.lds, rs, c.voff, c.soff, V16, q);
#endif
    if (ADJ_C) {
      // acc_iq += sum_k t_ik conj(C_qk)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        double sr = acc.re[3 * i + q], si = acc.im[3 * i + q];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double xr = cr.re[k], xi = -cr.im[k];
          sr = fma(t.re[3 * i + k], xr, sr); sr = fma(-t.im[3 * i + k], xi, sr);
          si = fma(t.re[3 * i + k], xi, si); si = fma(t.im[3 * i + k], xr, si);
        }
        acc.re[3 * i + q] = sr; acc.im[3 * i + q] = si;
      }
    } else {
      // acc_ij += t_iq C_qj
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          double sr = acc.re[3 * i + j], si = acc.im[3 * i + j];
          sr = fma(t.re[3 * i + q], cr.re[j], sr); sr = fma(-t.im[3 * i + q], cr.im[j], sr);
          si = fma(t.re[3 * i + q], cr.im[j], si); si = fma(t.im[3 * i + q], cr.re[j], si);
          acc.re[3 * i + j] = sr; acc.im[3 * i + j] = si;
        }
    }
  }
}

template <bool IN, int ENT>
__device__ __forceinline__ void ld_m(M3& m, const Opnd<IN, ENT>& o, __amdgpu_buffer_rsrc_t rs, int V16) {
  ld_full<IN, ENT>(m, o.lds, rs, o.voff, o.soff, V16);
}

}  // namespace l2q
