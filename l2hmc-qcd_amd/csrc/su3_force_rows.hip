// su3_force_rows.hip -- SU(3) staple force, slice-resident sweep with the 3x3 algebra split by
// ROWS over wavefronts (gfx950).
//
//   F_mu(s) = coef * TAH( U_mu(s) * A_mu(s) ),   A = sum of the 6 staples   (the reference:
//   autograd of the Wilson action + projectTAH, lattice/su3/pytorch/lattice.py:299-308)
//
// Why rows.  A thread that owns a whole link carries acc, t, a, b as 3x3 complex fp64 matrices
// (36 VGPRs each): ~240 live registers => ONE wavefront per SIMD (su3_force_slice_kernel in
// su3_kernels.hip), and with one wavefront nothing hides the LDS / L2 latencies: rocprofv3 shows
// the SIMDs issuing 58 % of the time (45 % VALU), 34 % parked on s_waitcnt / barriers.  Row r of
// every product in a staple chain needs only row r of the LEFT factor:
//     (A B^H C^H)_r = ((A_r B^H) C^H),      (A^H B^H C)_r = ((conj(A_:r) B^H) C)
// so three wavefronts (r = 0, 1, 2) each run the whole chain on 1x3 row vectors: the same FMAs
// in total, ~1/3 of the registers per lane => 3 wavefronts per SIMD, and another wavefront's FMAs
// issue while one waits for its operands.  The price is LDS traffic (every wavefront reads the
// right-hand factors in full: 21 instead of 9 matrix entries per lane and staple).
//
// Workgroup = 64 spatial sites x 4 directions x 3 rows = 12 wavefronts, sweeping t.  LDS holds
// the spatial links of the CURRENT and NEXT time slice of the tile, the t-links of the current
// slice (the next slice's t-links are never an operand), and a double-buffered exchange area
// through which rows 1, 2 of the staple sum reach the row-0 wavefront, which forms U*A, TAH and
// stores the link.  Every link is fetched from HBM once per sweep (+ the tile's halo from L2):
// each thread prefetches its row of the slice after next into registers while it computes.
//
// Addressing is the other half of the design (hipcc otherwise keeps ~100 loop-invariant 64-bit
// addresses alive and spills them): the wavefront index is made provably uniform with
// readfirstlane, so direction / row / slice selects are scalar; global operands are buffer loads
// (chain base in the descriptor, the per-lane site offset in ONE 32-bit VGPR per neighbour, the
// uniform (link, entry, slice) part in the scalar offset); LDS operands are ds_read_b128 with the
// entry offset as immediate.
#include "su3_force_tile.hpp"

namespace l2q {

constexpr int kRowsThreads = kRS * 12;        // 4 directions x 3 rows
constexpr int kOffS0 = 0, kOffS1 = 3 * kPlaneB, kOffT = 6 * kPlaneB, kOffX = 7 * kPlaneB;
constexpr int kXBuf = 4 * 6 * kEnt;           // one exchange buffer: [4 mu][6 entries][kRS]
constexpr int kOffH = kOffX + 2 * kXBuf;       // x-halo (plane x0 + 1) of the current slice: links t, y, z
constexpr int kRowsLds = kOffH;
constexpr int kRowsLdsHalo = kOffH + 3 * kPlaneB;

// out_j = sum_k a_k * conj(B_jk)          (row vector times B^H)
__device__ __forceinline__ void rv_mul_mh(R3& out, const R3& a, const M3& b) {
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    double sr = 0.0, si = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double br = b.re[3 * j + k], bi = -b.im[3 * j + k];
      sr = fma(a.re[k], br, sr); sr = fma(-a.im[k], bi, sr);
      si = fma(a.re[k], bi, si); si = fma(a.im[k], br, si);
    }
    out.re[j] = sr; out.im[j] = si;
  }
}

// acc_j += sum_k t_k * conj(C_jk)
__device__ __forceinline__ void rv_mac_mh(R3& acc, const R3& t, const M3& c) {
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    double sr = acc.re[j], si = acc.im[j];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double cr = c.re[3 * j + k], ci = -c.im[3 * j + k];
      sr = fma(t.re[k], cr, sr); sr = fma(-t.im[k], ci, sr);
      si = fma(t.re[k], ci, si); si = fma(t.im[k], cr, si);
    }
    acc.re[j] = sr; acc.im[j] = si;
  }
}

// acc_j += sum_k t_k * C_kj
__device__ __forceinline__ void rv_mac_m(R3& acc, const R3& t, const M3& c) {
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    double sr = acc.re[j], si = acc.im[j];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double cr = c.re[3 * k + j], ci = c.im[3 * k + j];
      sr = fma(t.re[k], cr, sr); sr = fma(-t.im[k], ci, sr);
      si = fma(t.re[k], ci, si); si = fma(t.im[k], cr, si);
    }
    acc.re[j] = sr; acc.im[j] = si;
  }
}

// keeps hipcc from hoisting the next staple's operand loads above the current staple's
// arithmetic (unfenced it keeps 6 staples of operands in flight and spills ~1000 VGPRs)
#ifdef L2Q_ROWS_FENCE
#define L2Q_STAPLE_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define L2Q_STAPLE_FENCE() do { } while (0)
#endif

// tile residency of the +-dir neighbours (dir = 1, 2, 3 = x, y, z) for EVERY lane, compile time:
// INM bit (dir - 1).  A 64-site tile of consecutive spatial sites holds whole z-rows if Z | 64,
// whole (y,z)-planes if Y Z | 64, whole spatial volumes if X Y Z | 64.
template <int INM>
__device__ __forceinline__ constexpr bool in_dir(int dir) { return dir == 0 ? true : ((INM >> (dir - 1)) & 1) != 0; }

struct RowsCtx {
  __amdgpu_buffer_rsrc_t rs, ro;
  Dims d;
  int V16, Vs16, tile0b, lt, r, t0, t1;
  int sp, px, py, pz;
  double coef;
};

// One wavefront's sweep: direction MU, row c.r.  MODE 0: out = coef * F;  MODE 1: out += coef * F
template <int MODE, int MU, int INM>
__device__ __forceinline__ void force_rows_sweep(const RowsCtx& c) {
  const Dims& d = c.d;
  const int T = d.T, V16 = c.V16, Vs16 = c.Vs16, r = c.r;
  const __amdgpu_buffer_rsrc_t rs = c.rs, ro = c.ro;
  constexpr bool IN_MU = in_dir<INM>(MU);
  // HALO (tile = one whole (y,z)-plane, x the only direction that leaves it -- the 8^4 bench
  // shape): the wavefronts of direction x would chain ~7 serial L2 round trips per slice (the
  // first factor of each of their staples lives at x0 + 1) and every barrier would wait for
  // them.  The links t, y, z of plane x0 + 1 are therefore staged in LDS too (27 KiB), fetched
  // one slice ahead like the tile itself.
  constexpr bool HALO = (INM & 8) != 0;
  constexpr bool A_LDS = IN_MU || (HALO && MU == 1);   // first staple factors at s + mu: LDS?
  // per-lane byte offsets of the neighbour sites (slice-independent)
  const int q_sp = c.sp * 16;
  int q_pmu = q_sp, mx = c.px, my = c.py, mz = c.pz;
  if (MU != 0) {
    int q = hop(c.sp, c.px, c.py, c.pz, MU, +1, d);
    q_pmu = q * 16;
    mz = q % d.Z; q /= d.Z;
    my = q % d.Y; q /= d.Y;
    mx = q;
  }
  int q_pp[4], q_pm[4], q_pmm[4];
#pragma unroll
  for (int nu = 1; nu < 4; ++nu) {
    q_pp[nu] = hop(c.sp, c.px, c.py, c.pz, nu, +1, d) * 16;
    q_pm[nu] = hop(c.sp, c.px, c.py, c.pz, nu, -1, d) * 16;
    q_pmm[nu] = hop(q_pmu / 16, mx, my, mz, nu, -1, d) * 16;
  }
  const int lb = -c.tile0b;                           // LDS address of site q: region + lb + q * 16
  // halo slot of a site of plane x0 + 1 = its (y,z) index; link rho -> halo plane (t, y, z) -> (0, 1, 2)
  auto hslot = [&](int qb) { return ((qb >> 4) & (kRS - 1)) * 16; };
  auto hoff = [&](int rho) { return kOffH + (rho == 0 ? 0 : rho - 1) * kPlaneB; };
  constexpr int HRHO = MU == 0 ? 0 : MU == 1 ? 2 : 3;  // halo link this wavefront keeps fed (MU < 3)
  const int q_hx = (HALO && MU < 3) ? (MU == 1 ? q_pmu : hop(c.sp, c.px, c.py, c.pz, 1, +1, d) * 16) : 0;
  const int own_hrow = kOffH + (MU == 0 ? 0 : MU) * kPlaneB + 3 * r * kEnt + c.lt * 16;
  // byte offset in LDS of this thread's row of its link inside a spatial slot / the t buffer
  const int own_row = (MU == 0 ? 0 : (MU - 1) * kPlaneB) + 3 * r * kEnt + c.lt * 16;
  // prologue: slice ta = t0 - 1 -> (S0, Tl), slice t0 spatial -> S1
  {
    const int ta = (c.t0 - 1 + T) % T;
    const int g0 = (MU * 9 + 3 * r) * V16;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double2 va = buf_ld(rs, q_sp, g0 + k * V16 + ta * Vs16);
      if (MU == 0) {
        *reinterpret_cast<double2*>(fr_lds + kOffT + own_row + k * kEnt) = va;
      } else {
        const double2 vb = buf_ld(rs, q_sp, g0 + k * V16 + (c.t0 % T) * Vs16);
        *reinterpret_cast<double2*>(fr_lds + kOffS0 + own_row + k * kEnt) = va;
        *reinterpret_cast<double2*>(fr_lds + kOffS1 + own_row + k * kEnt) = vb;
      }
    }
  }
  if (HALO && MU < 3) {
    const int ta = (c.t0 - 1 + T) % T;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      *reinterpret_cast<double2*>(fr_lds + own_hrow + k * kEnt) =
          buf_ld(rs, q_hx, (HRHO * 9 + 3 * r + k) * V16 + ta * Vs16);
  }
  __syncthreads();
  int cur = 0;
  R3 carry;                                           // t-direction down staple (spatial MU), row r
  r3_zero(carry);
  const int niter = (c.t1 - c.t0) + 1;
#pragma unroll 1
  for (int it = 0; it < niter; ++it) {
    const int tcur = (c.t0 - 1 + it + T) % T;
    const int tnext = (tcur + 1 == T) ? 0 : tcur + 1;
    const int offSc = cur ? kOffS1 : kOffS0;          // spatial links of slice tcur / tnext
    const int offSn = cur ? kOffS0 : kOffS1;
    const int gcur = tcur * Vs16, gnxt = tnext * Vs16;
    const bool more = it + 1 < niter;
    // LDS region (+ lb) and uniform global offset of link direction rho in slice tcur / tnext
    auto lc = [&](int rho) { return (rho == 0 ? kOffT : offSc + (rho - 1) * kPlaneB) + lb; };
    auto ln = [&](int rho) { return offSn + (rho - 1) * kPlaneB + lb; };
    auto gc = [&](int rho) { return rho * 9 * V16 + gcur; };
    auto gn = [&](int rho) { return rho * 9 * V16 + gnxt; };
    // prefetch this thread's row of the slice that enters LDS after this iteration:
    // t-links of slice tnext, spatial links of the slice after next
    double2 pre[3];
    if (more) {
      const int tp = MU == 0 ? tnext : ((tnext + 1 == T) ? 0 : tnext + 1);
      const int g0 = (MU * 9 + 3 * r) * V16 + tp * Vs16;
#pragma unroll
      for (int k = 0; k < 3; ++k) pre[k] = buf_ld(rs, q_sp, g0 + k * V16);
    }
    double2 preh[3];
    if (HALO && MU < 3 && more) {
#pragma unroll
      for (int k = 0; k < 3; ++k) preh[k] = buf_ld(rs, q_hx, (HRHO * 9 + 3 * r + k) * V16 + gnxt);
    }
    R3 acc;
    r3_zero(acc);
    if (it > 0) {
      if constexpr (MU == 0) {
#pragma unroll
        for (int nu = 1; nu < 4; ++nu) {
          constexpr bool dummy = true; (void)dummy;
          R3 a, t;
          M3 b;
          // up:   U_nu(s+t) U_t(s+nu)^H U_nu(s)^H
          ld_row<true>(a, ln(nu) + q_sp, rs, q_sp, gn(nu), V16, r);
          if (in_dir<INM>(nu)) ld_full<true>(b, lc(0) + q_pp[nu], rs, q_pp[nu], gc(0), V16);
          else ld_full<false>(b, 0, rs, q_pp[nu], gc(0), V16);
          rv_mul_mh(t, a, b);
          ld_full<true>(b, lc(nu) + q_sp, rs, q_sp, gc(nu), V16);
          rv_mac_mh(acc, t, b);
          L2Q_STAPLE_FENCE();
          // down: U_nu(s+t-nu)^H U_t(s-nu)^H U_nu(s-nu)
          if (in_dir<INM>(nu)) {
            ld_colc<true>(a, ln(nu) + q_pm[nu], rs, q_pm[nu], gn(nu), V16, r);
            ld_full<true>(b, lc(0) + q_pm[nu], rs, q_pm[nu], gc(0), V16);
            rv_mul_mh(t, a, b);
            ld_full<true>(b, lc(nu) + q_pm[nu], rs, q_pm[nu], gc(nu), V16);
          } else {
            ld_colc<false>(a, 0, rs, q_pm[nu], gn(nu), V16, r);
            ld_full<false>(b, 0, rs, q_pm[nu], gc(0), V16);
            rv_mul_mh(t, a, b);
            ld_full<false>(b, 0, rs, q_pm[nu], gc(nu), V16);
          }
          rv_mac_m(acc, t, b);
          L2Q_STAPLE_FENCE();
        }
      } else {
        acc = carry;                                  // down staple in the t direction
        {
          R3 a, t;
          M3 b;
          // up (nu = t): U_t(s+mu) U_mu(s+t)^H U_t(s)^H
          ld_row<A_LDS>(a, IN_MU ? lc(0) + q_pmu : hoff(0) + hslot(q_pmu), rs, q_pmu, gc(0), V16, r);
          ld_full<true>(b, ln(MU) + q_sp, rs, q_sp, gn(MU), V16);
          rv_mul_mh(t, a, b);
          ld_full<true>(b, lc(0) + q_sp, rs, q_sp, gc(0), V16);
          rv_mac_mh(acc, t, b);
          L2Q_STAPLE_FENCE();
        }
#pragma unroll
        for (int nu = 1; nu < 4; ++nu) {
          if (nu == MU) continue;
          R3 a, t;
          M3 b;
          // up:   U_nu(s+mu) U_mu(s+nu)^H U_nu(s)^H
          ld_row<A_LDS>(a, IN_MU ? lc(nu) + q_pmu : hoff(nu) + hslot(q_pmu), rs, q_pmu, gc(nu), V16, r);
          if (in_dir<INM>(nu)) ld_full<true>(b, lc(MU) + q_pp[nu], rs, q_pp[nu], gc(MU), V16);
          else ld_full<false>(b, 0, rs, q_pp[nu], gc(MU), V16);
          rv_mul_mh(t, a, b);
          ld_full<true>(b, lc(nu) + q_sp, rs, q_sp, gc(nu), V16);
          rv_mac_mh(acc, t, b);
          L2Q_STAPLE_FENCE();
          // down: U_nu(s+mu-nu)^H U_mu(s-nu)^H U_nu(s-nu)
          if (IN_MU && in_dir<INM>(nu)) ld_colc<true>(a, lc(nu) + q_pmm[nu], rs, q_pmm[nu], gc(nu), V16, r);
          else if (HALO && MU == 1) ld_colc<true>(a, hoff(nu) + hslot(q_pmm[nu]), rs, q_pmm[nu], gc(nu), V16, r);
          else ld_colc<false>(a, 0, rs, q_pmm[nu], gc(nu), V16, r);
          if (in_dir<INM>(nu)) {
            ld_full<true>(b, lc(MU) + q_pm[nu], rs, q_pm[nu], gc(MU), V16);
            rv_mul_mh(t, a, b);
            ld_full<true>(b, lc(nu) + q_pm[nu], rs, q_pm[nu], gc(nu), V16);
          } else {
            ld_full<false>(b, 0, rs, q_pm[nu], gc(MU), V16);
            rv_mul_mh(t, a, b);
            ld_full<false>(b, 0, rs, q_pm[nu], gc(nu), V16);
          }
          rv_mac_m(acc, t, b);
          L2Q_STAPLE_FENCE();
        }
      }
    }
    if (MU != 0 && more) {
      // next iteration's t-direction down staple of link (tnext, sp, mu), all from slice tcur:
      //   U_t(tcur, sp+mu)^H U_mu(tcur, sp)^H U_t(tcur, sp)
      R3 a, t;
      M3 b;
      ld_colc<A_LDS>(a, IN_MU ? lc(0) + q_pmu : hoff(0) + hslot(q_pmu), rs, q_pmu, gc(0), V16, r);
      ld_full<true>(b, lc(MU) + q_sp, rs, q_sp, gc(MU), V16);
      rv_mul_mh(t, a, b);
      ld_full<true>(b, lc(0) + q_sp, rs, q_sp, gc(0), V16);
      r3_zero(carry);
      rv_mac_m(carry, t, b);
      L2Q_STAPLE_FENCE();
    }
    const int xb = kOffX + (it & 1) * kXBuf + MU * 6 * kEnt + c.lt * 16;
    if (it > 0 && r != 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k)
        *reinterpret_cast<double2*>(fr_lds + xb + ((r - 1) * 3 + k) * kEnt) = make_double2(acc.re[k], acc.im[k]);
    }
    __syncthreads();                                  // slice tcur consumed, staple rows published
    if (more) {
      const int dst = (MU == 0 ? kOffT : offSc) + own_row;
#pragma unroll
      for (int k = 0; k < 3; ++k) *reinterpret_cast<double2*>(fr_lds + dst + k * kEnt) = pre[k];
      if (HALO && MU < 3) {
#pragma unroll
        for (int k = 0; k < 3; ++k) *reinterpret_cast<double2*>(fr_lds + own_hrow + k * kEnt) = preh[k];
      }
    }
    cur ^= 1;
    __syncthreads();                                  // next slice in place
    if (it > 0 && r == 0) {
      // U * A with A = (own row 0 | rows 1, 2 from the exchange buffer), then TAH and the store.
      // The link itself comes back from L2 (its LDS copy has just been replaced); it is streamed
      // row by row and TAH is formed entry by entry at the store, so the live set stays at two
      // matrices.
      M3 a, ua;
#pragma unroll
      for (int k = 0; k < 3; ++k) { a.re[k] = acc.re[k]; a.im[k] = acc.im[k]; }
#pragma unroll
      for (int e = 0; e < 6; ++e) { const double2 dd = lds_ld(xb + e * kEnt); a.re[3 + e] = dd.x; a.im[3 + e] = dd.y; }
      const int so = MU * 9 * V16 + gcur;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        double ur[3], ui[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { const double2 dd = buf_ld(rs, q_sp, so + (3 * i + k) * V16); ur[k] = dd.x; ui[k] = dd.y; }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          double sr = 0.0, si = 0.0;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            sr = fma(ur[k], a.re[3 * k + j], sr); sr = fma(-ui[k], a.im[3 * k + j], sr);
            si = fma(ur[k], a.im[3 * k + j], si); si = fma(ui[k], a.re[3 * k + j], si);
          }
          ua.re[3 * i + j] = sr; ua.im[3 * i + j] = si;
        }
      }
      // F = (W - W^H)/2 - tr(W - W^H)/6   (group/su3/pytorch/group.py:92-103), W = U A
      const double tri = (ua.im[0] + ua.im[4] + ua.im[8]) / 3.0;     // the trace term is imaginary
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int e = 3 * i + j, et = 3 * j + i;
          const double fr = 0.5 * (ua.re[e] - ua.re[et]);
          double fi = 0.5 * (ua.im[e] + ua.im[et]);
          if (i == j) fi -= tri;
          double2 v2 = make_double2(c.coef * fr, c.coef * fi);
          if (MODE == 1) {
            const double2 o = buf_ld(ro, q_sp, so + e * V16);
            v2.x += o.x; v2.y += o.y;
          }
          buf_st(ro, q_sp, so + e * V16, v2);
        }
    }
  }
}

template <int MODE, int INM>
__global__ __launch_bounds__(kRowsThreads) void su3_force_rows_kernel(
    const double2* __restrict__ xn, Dims d, int nsb, int tsplit, int swz, double coef,
    double2* __restrict__ out) {
  const long w = xcd_swizzle(blockIdx.x, gridDim.x, swz);
  const int per_chain = nsb * tsplit;
  const long c = w / per_chain;
  const int rr = (int)(w % per_chain);
  const int tc = rr / nsb, sb = rr % nsb;
  const int V = d.V, T = d.T;
  RowsCtx k;
  k.d = d;
  k.V16 = V * 16;
  k.Vs16 = d.X * d.Y * d.Z * 16;
  k.tile0b = sb * kRS * 16;
  k.lt = threadIdx.x & (kRS - 1);
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kRS);     // provably wave-uniform
  const int mu = wv / 3;                              // direction and row of this wavefront
  k.r = wv - 3 * mu;
  const int tlen = (T + tsplit - 1) / tsplit;
  k.t0 = tc * tlen;
  k.t1 = min(T, k.t0 + tlen);
  const int chain_bytes = 36 * k.V16;
  k.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(xn + c * 36L * V), 0, chain_bytes, 0x00020000);
  k.ro = __builtin_amdgcn_make_buffer_rsrc((void*)(out + c * 36L * V), 0, chain_bytes, 0x00020000);
  k.sp = sb * kRS + k.lt;
  {
    int q = k.sp;
    k.pz = q % d.Z; q /= d.Z;
    k.py = q % d.Y; q /= d.Y;
    k.px = q;
  }
  k.coef = coef;
  // every wavefront runs the sweep specialised for its direction (identical barrier sequence)
  switch (mu) {
    case 0: force_rows_sweep<MODE, 0, INM>(k); break;
    case 1: force_rows_sweep<MODE, 1, INM>(k); break;
    case 2: force_rows_sweep<MODE, 2, INM>(k); break;
    default: force_rows_sweep<MODE, 3, INM>(k); break;
  }
}

size_t force_rows_lds_bytes() { return (size_t)kRowsLdsHalo; }

bool force_rows_applicable(const Dims& d) {
  return (d.X * d.Y * d.Z) % kRS == 0 && 36.0 * d.V * 16.0 < 2.0e9;
}

template <int MODE, int INM>
static void launch_rows_variant(const double2* xn, Dims d, int nb, int nsb, int tsplit, double coef,
                                double2* out, hipStream_t st) {
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute((const void*)su3_force_rows_kernel<MODE, INM>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (INM & 8) ? kRowsLdsHalo : kRowsLds);
  }
  hipLaunchKernelGGL((su3_force_rows_kernel<MODE, INM>), dim3((unsigned)((long)nb * nsb * tsplit)),
                     dim3(kRowsThreads), (INM & 8) ? kRowsLdsHalo : kRowsLds, st, xn, d, nsb, tsplit, tuning().xcd_swizzle, coef, out);
}

int force_rows_inmask(const Dims& d) {
  int m = 0;
  if (kRS % d.Z == 0) m |= 4;                               // +-z neighbours stay in the tile
  if (kRS % (d.Y * d.Z) == 0) m |= 2;                       // +-y
  if (kRS % (d.X * d.Y * d.Z) == 0) m |= 1;                 // +-x
  if (m == 6 && d.Y * d.Z == kRS) m |= 8;                   // tile = one (y,z)-plane: x-halo in LDS
  return m;
}

void launch_force_rows(bool kick, const double2* xn, Dims d, int nb, double coef, double2* out,
                       hipStream_t st) {
  const int Vs = d.X * d.Y * d.Z;
  const int nsb = Vs / kRS;
  int tsplit = (int)cdiv(512, (long)nb * nsb);         // >= ~2 resident rounds of 256 CUs
  if (tsplit > d.T) tsplit = d.T;
  if (tsplit < 1) tsplit = 1;
  const int tlen = (int)cdiv(d.T, tsplit);
  tsplit = (int)cdiv(d.T, tlen);
  const int inm = force_rows_inmask(d);
#define L2Q_ROWS_CASE(M)                                                                    \
  case M:                                                                                   \
    if (kick) launch_rows_variant<1, M>(xn, d, nb, nsb, tsplit, coef, out, st);             \
    else launch_rows_variant<0, M>(xn, d, nb, nsb, tsplit, coef, out, st);                  \
    break;
  switch (inm) {
    L2Q_ROWS_CASE(14)
    L2Q_ROWS_CASE(7)
    L2Q_ROWS_CASE(6)
    L2Q_ROWS_CASE(4)
    default:
      if (kick) launch_rows_variant<1, 0>(xn, d, nb, nsb, tsplit, coef, out, st);
      else launch_rows_variant<0, 0>(xn, d, nb, nsb, tsplit, coef, out, st);
  }
#undef L2Q_ROWS_CASE
}

}  // namespace l2q
