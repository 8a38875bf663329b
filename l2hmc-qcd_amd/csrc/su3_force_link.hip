// su3_force_link.hip -- SU(3) staple force, slice-resident sweep, one thread per link with the
// right-hand factors of every 3x3 product STREAMED row by row (gfx950).
//
//   F_mu(s) = coef * TAH( U_mu(s) * A_mu(s) ),   A = sum over nu != mu of the up and the down
//   staple in the (mu, nu) plane    (the reference: autograd of the Wilson action + projectTAH,
//   lattice/su3/pytorch/lattice.py:299-308)
//
// The same arithmetic and tile as su3_force_nu.hip, without its cross-wavefront exchange: a
// wavefront owns direction mu of the tile's 64 sites and walks the three planes of mu itself.
// With the right-hand factors streamed (live set: acc, t, a = three 36-VGPR matrices + one row)
// plus the carried t-staple and the 9-entry prefetch of the thread's own link, a thread needs
// 210-246 registers: TWO wavefronts per SIMD.  A workgroup is 64 sites x 4 directions =
// 4 wavefronts and 63 KiB of LDS (spatial links of the current and next slice, t-links of the
// current slice), so TWO workgroups share a CU and one runs while the other sits at its slice
// barrier; there is no exchange buffer, no finishing wavefront and no third party to wait for.
//
// What makes it compile: hipcc schedules an unrolled straight-line chain of 3x3 fp64 products
// with a live set far above what the data flow needs (this kernel spilled 146-431 VGPRs at the
// 256-register budget; sched_barrier / memory fences between the staples change nothing).  Here
// EVERY staple sits inside its own `if (it >= c.lo)`, where `lo` (= 1: the kernel's `it > 0`
// test) arrives as a kernel ARGUMENT: the compiler can neither fold the conditions nor merge the
// blocks, each block is allocated on its own, and the result is 0 spilled registers.
// (A struct field set from a template constant does not do: it is folded after inlining.)
//
// MI355X, cfg-4 (8^4 x 256 chains): plain force 0.39-0.41 ms (0.37-0.39 of the 8 TB/s roofline; the
// plane-split kernel su3_force_nu.hip: 0.43-0.45 ms), fused kick 0.51 ms (0.44; 0.61 ms);
// 16^4 x 64 chains: 1.89 / 2.39 ms (2.11 / 2.79 ms).  Results are bit-identical to the thread-per-link
// slice kernel (same operation order per link).
#include "su3_force_tile.hpp"

namespace l2q {

// keeps hipcc from starting the next staple's operand loads before the current staple's
// arithmetic has retired its matrices (unfenced it software-pipelines across staples and spills)
#ifdef L2Q_LK_NOFENCE
#define L2Q_LK_FENCE() do { } while (0)
#else
#define L2Q_LK_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#endif

// Timing experiments (tools/ab_build.sh ... -DL2Q_LK_EXP=bits; results are WRONG, only the clock is read):
//   1  halo operands come from the tile's own LDS copy (no neighbour loads from L2 / HBM)
//   2  the output is stored only where it equals a magic value (no store traffic, arithmetic kept)
//   4  no slice refresh: the own-link prefetch and the LDS rewrite are skipped (with 1: no global loads)
//   8  no slice barriers
//  16  plain (write-back) stores instead of streaming ones
//  64  planes in the order z, y, x (the staples whose operands are all in LDS first)
//  32  the output is stored AFTER the slice refresh (the vmcnt wait for the prefetched own link then does
//      not include the acknowledgement of this iteration's stores: vmcnt is in order on gfx9)
#ifndef L2Q_LK_EXP
#define L2Q_LK_EXP 0
#endif
// A/B: 1 = form the real parts of diag(U A) as well (they cancel in the projection; round-3 kernel)
#ifndef L2Q_LK_FULL_DIAG
#define L2Q_LK_FULL_DIAG 0
#endif

constexpr int kLkThreads = kRS * 4;
constexpr int kLkOffS0 = 0, kLkOffS1 = 3 * kPlaneB, kLkOffT = 6 * kPlaneB;
constexpr int kLkLds = 7 * kPlaneB;

template <int INM>
__device__ __forceinline__ constexpr bool lk_in(int dir) { return dir == 0 ? true : ((INM >> (dir - 1)) & 1) != 0; }

struct LkCtx {
  __amdgpu_buffer_rsrc_t rs, ro, rv;   // links, output, the momentum the kick reads (MODE 1)
  Dims d;
  int V16, Vs16, tile0b, lt, t0, t1;
  int sp, px, py, pz;
  double coef;
  int lo;        // = 1, a kernel ARGUMENT: `it >= lo` is the `it > 0` test hipcc cannot fold or merge
};

__host__ __device__ constexpr int lk_other(int mu, int j) { return j + (j >= mu ? 1 : 0); }

// MODE 0: out = coef * F;  MODE 1: out = vin + coef * F  (vin == out: the in-place kick);
// MODE 2 (training, the VJP of the force with the staple sum held constant like the reference's
//   autograd.grad(action) without create_graph, lattice/su3/pytorch/lattice.py:299-308):
//   out += coef * TAH(vin) A^H  with vin = g_F, out = g_x  (the same staple sweep, another epilogue)
template <int MODE, int MU, int INM>
__device__ __forceinline__ void force_link_sweep(const LkCtx& c) {
  constexpr bool IN_MU = lk_in<INM>(MU);
  const Dims& d = c.d;
  const int T = d.T, V16 = c.V16, Vs16 = c.Vs16;
  const __amdgpu_buffer_rsrc_t rs = c.rs, ro = c.ro, rv = c.rv;
  const int q_sp = c.sp * 16;
  int q_pmu = q_sp, mx = c.px, my = c.py, mz = c.pz;          // s + mu (spatial MU)
  if (MU != 0) {
    int q = hop(c.sp, c.px, c.py, c.pz, MU, +1, d);
    q_pmu = q * 16;
    mz = q % d.Z; q /= d.Z;
    my = q % d.Y; q /= d.Y;
    mx = q;
  }
  int q_pp[4], q_pm[4], q_pmm[4];                             // s + nu, s - nu, s + mu - nu
#pragma unroll
  for (int nu = 1; nu < 4; ++nu) {
    q_pp[nu] = hop(c.sp, c.px, c.py, c.pz, nu, +1, d) * 16;
    q_pm[nu] = hop(c.sp, c.px, c.py, c.pz, nu, -1, d) * 16;
    q_pmm[nu] = hop(q_pmu / 16, mx, my, mz, nu, -1, d) * 16;
  }
  const int lb = -c.tile0b;
  const int own = (MU == 0 ? 0 : (MU - 1) * kPlaneB) + c.lt * 16;   // this thread's link in a slot
  {
    const int ta = (c.t0 - 1 + T) % T;
#pragma unroll
    for (int e = 0; e < 9; ++e) {
      const double2 va = buf_ld(rs, q_sp, (MU * 9 + e) * V16 + ta * Vs16);
      if (MU == 0) {
        *reinterpret_cast<double2*>(fr_lds + kLkOffT + own + e * kEnt) = va;
      } else {
        const double2 vb = buf_ld(rs, q_sp, (MU * 9 + e) * V16 + (c.t0 % T) * Vs16);
        *reinterpret_cast<double2*>(fr_lds + kLkOffS0 + own + e * kEnt) = va;
        *reinterpret_cast<double2*>(fr_lds + kLkOffS1 + own + e * kEnt) = vb;
      }
    }
  }
  __syncthreads();
  int cur = 0;
  M3 carry;                                           // spatial MU: t-direction down staple
  if (MU != 0) m3_zero(carry);
  const int niter = (c.t1 - c.t0) + 1;
#pragma unroll 1
  for (int it = 0; it < niter; ++it) {
    const int tcur = (c.t0 - 1 + it + T) % T;
    const int tnext = (tcur + 1 == T) ? 0 : tcur + 1;
    const int offSc = cur ? kLkOffS1 : kLkOffS0;
    const int offSn = cur ? kLkOffS0 : kLkOffS1;
    const int gcur = tcur * Vs16, gnxt = tnext * Vs16;
    const bool more = it + 1 < niter;
    auto oc = [&](int rho, int qb) {
      return Opnd<true>{(rho == 0 ? kLkOffT : offSc + (rho - 1) * kPlaneB) + lb + qb, qb, rho * 9 * V16 + gcur};
    };
    auto on = [&](int rho, int qb) {
      return Opnd<true>{offSn + (rho - 1) * kPlaneB + lb + qb, qb, rho * 9 * V16 + gnxt};
    };
#if L2Q_LK_EXP & 1
    auto gco = [&](int rho, int qb) { return oc(rho, q_sp); };
    auto gno = [&](int rho, int qb) { return on(rho == 0 ? 1 : rho, q_sp); };
#else
    auto gco = [&](int rho, int qb) { return Opnd<false>{0, qb, rho * 9 * V16 + gcur}; };
    auto gno = [&](int rho, int qb) { return Opnd<false>{0, qb, rho * 9 * V16 + gnxt}; };
#endif
    // prefetch the thread's own link of the slice that enters LDS after this iteration
    double2 pre[9];
    if (more && !(L2Q_LK_EXP & 4)) {
      const int tp = MU == 0 ? tnext : ((tnext + 1 == T) ? 0 : tnext + 1);
#pragma unroll
      for (int e = 0; e < 9; ++e) pre[e] = buf_ld(rs, q_sp, (MU * 9 + e) * V16 + tp * Vs16);
    }
    M3 acc;
    double2 outv[9];
    m3_zero(acc);
    if constexpr (MU == 0) {
      if (it >= c.lo) {
#pragma unroll
        for (int nu_ = 1; nu_ < 4; ++nu_) {
          const int nu = (L2Q_LK_EXP & 64) ? 4 - nu_ : nu_;
          M3 a, t;
          if (it >= c.lo) {
          // up:   U_nu(s+t) U_t(s+nu)^H U_nu(s)^H
          ld_m(a, on(nu, q_sp), rs, V16);
          if (lk_in<INM>(nu)) mul_xh_stream<false>(t, a, oc(0, q_pp[nu]), rs, V16);
          else mul_xh_stream<false>(t, a, gco(0, q_pp[nu]), rs, V16);
          mac_stream<true>(acc, t, oc(nu, q_sp), rs, V16);
          }
          L2Q_LK_FENCE();
          if (it < c.lo) continue;
          // down: U_nu(s+t-nu)^H U_t(s-nu)^H U_nu(s-nu)
          if (lk_in<INM>(nu)) {
            ld_m(a, on(nu, q_pm[nu]), rs, V16);
            mul_xh_stream<true>(t, a, oc(0, q_pm[nu]), rs, V16);
            mac_stream<false>(acc, t, oc(nu, q_pm[nu]), rs, V16);
            L2Q_LK_FENCE();
          } else {
            ld_m(a, gno(nu, q_pm[nu]), rs, V16);
            mul_xh_stream<true>(t, a, gco(0, q_pm[nu]), rs, V16);
            mac_stream<false>(acc, t, gco(nu, q_pm[nu]), rs, V16);
            L2Q_LK_FENCE();
          }
        }
      }
    } else {
      {
        // plane (MU, t): the down staple was formed one slice earlier (carry)
        M3 a, t;
        if (IN_MU) ld_m(a, oc(0, q_pmu), rs, V16);            // U_t(tcur, s+mu): both staples
        else ld_m(a, gco(0, q_pmu), rs, V16);
        if (it >= c.lo) {
          acc = carry;
          // up: U_t(s+mu) U_mu(s+t)^H U_t(s)^H
          mul_xh_stream<false>(t, a, on(MU, q_sp), rs, V16);
          mac_stream<true>(acc, t, oc(0, q_sp), rs, V16);
          L2Q_LK_FENCE();
        }
        if (more) {
          // next slice's down staple: U_t(tcur, s+mu)^H U_mu(tcur, s)^H U_t(tcur, s)
          mul_xh_stream<true>(t, a, oc(MU, q_sp), rs, V16);
          m3_zero(carry);
          mac_stream<false>(carry, t, oc(0, q_sp), rs, V16);
          L2Q_LK_FENCE();
        }
      }
      if (it >= c.lo) {
#pragma unroll
        for (int nu_ = 1; nu_ < 4; ++nu_) {
          const int nu = (L2Q_LK_EXP & 64) ? 4 - nu_ : nu_;
          if (nu == MU) continue;
          M3 a, t;
          if (it >= c.lo) {
          // up:   U_nu(s+mu) U_mu(s+nu)^H U_nu(s)^H
          if (IN_MU) ld_m(a, oc(nu, q_pmu), rs, V16);
          else ld_m(a, gco(nu, q_pmu), rs, V16);
          if (lk_in<INM>(nu)) mul_xh_stream<false>(t, a, oc(MU, q_pp[nu]), rs, V16);
          else mul_xh_stream<false>(t, a, gco(MU, q_pp[nu]), rs, V16);
          mac_stream<true>(acc, t, oc(nu, q_sp), rs, V16);
          }
          L2Q_LK_FENCE();
          if (it < c.lo) continue;
          // down: U_nu(s+mu-nu)^H U_mu(s-nu)^H U_nu(s-nu)
          if (IN_MU && lk_in<INM>(nu)) ld_m(a, oc(nu, q_pmm[nu]), rs, V16);
          else ld_m(a, gco(nu, q_pmm[nu]), rs, V16);
          if (lk_in<INM>(nu)) {
            mul_xh_stream<true>(t, a, oc(MU, q_pm[nu]), rs, V16);
            mac_stream<false>(acc, t, oc(nu, q_pm[nu]), rs, V16);
            L2Q_LK_FENCE();
          } else {
            mul_xh_stream<true>(t, a, gco(MU, q_pm[nu]), rs, V16);
            mac_stream<false>(acc, t, gco(nu, q_pm[nu]), rs, V16);
            L2Q_LK_FENCE();
          }
        }
      }
    }
    if (MODE == 2 && it >= c.lo) {
      // G = TAH(g_F(s, mu)) from memory, then entry (i, j) of G A^H = sum_k G_ik conj(A_jk), added to g_x
      const int so = MU * 9 * V16 + gcur;
      M3 gl, g;
#pragma unroll
      for (int e = 0; e < 9; ++e) {
        const double2 o = buf_ld_nt(rv, q_sp, so + e * V16);
        gl.re[e] = o.x; gl.im[e] = o.y;
      }
      m3_tah(g, gl);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          double sr = 0.0, si = 0.0;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const double ar = acc.re[3 * j + k], ai = -acc.im[3 * j + k];
            sr = fma(g.re[3 * i + k], ar, sr); sr = fma(-g.im[3 * i + k], ai, sr);
            si = fma(g.re[3 * i + k], ai, si); si = fma(g.im[3 * i + k], ar, si);
          }
          const double2 o = buf_ld_nt(ro, q_sp, so + (3 * i + j) * V16);
          outv[3 * i + j] = make_double2(fma(c.coef, sr, o.x), fma(c.coef, si, o.y));
        }
#pragma unroll
      for (int e = 0; e < 9; ++e) buf_st_nt(ro, q_sp, so + e * V16, outv[e]);
    }
    if (MODE != 2 && it >= c.lo) {
      // W = U A with U streamed by rows from the tile; F = (W - W^H)/2 - tr(W - W^H)/6
      // (group/su3/pytorch/group.py:92-103), formed entry by entry at the store.  The output (and
      // the momentum read by the kick) is touched once: streaming (nt) accesses keep it from
      // evicting the links the neighbouring tiles are about to re-read from L2 (-4 % measured)
      M3 ua;
      const Opnd<true> uo = oc(MU, q_sp);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        R3 ur;
        ld_row<true>(ur, uo.lds, rs, uo.voff, uo.soff, V16, i);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          double sr = 0.0, si = 0.0;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            // the real parts of the diagonal cancel in (W - W^H) / 2: not formed (18 of the 108 FMAs)
            if (L2Q_LK_FULL_DIAG || i != j) { sr = fma(ur.re[k], acc.re[3 * k + j], sr); sr = fma(-ur.im[k], acc.im[3 * k + j], sr); }
            si = fma(ur.re[k], acc.im[3 * k + j], si); si = fma(ur.im[k], acc.re[3 * k + j], si);
          }
          ua.re[3 * i + j] = sr; ua.im[3 * i + j] = si;
        }
      }
      const int so = MU * 9 * V16 + gcur;
      const double tri = (ua.im[0] + ua.im[4] + ua.im[8]) / 3.0;     // the trace term is imaginary
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int e = 3 * i + j, et = 3 * j + i;
          const double fr = (!L2Q_LK_FULL_DIAG && i == j) ? 0.0 : 0.5 * (ua.re[e] - ua.re[et]);
          double fi = 0.5 * (ua.im[e] + ua.im[et]);
          if (i == j) fi -= tri;
          double2 v2 = make_double2(c.coef * fr, c.coef * fi);
          if (MODE == 1) {
            const double2 o = buf_ld_nt(rv, q_sp, so + e * V16);
            v2.x += o.x; v2.y += o.y;
          }
          outv[e] = v2;
        }
      if (!(L2Q_LK_EXP & 32)) {
#pragma unroll
        for (int e = 0; e < 9; ++e)
          if (!(L2Q_LK_EXP & 2) || outv[e].x == 1.2345e300) {
            if (L2Q_LK_EXP & 16) buf_st(ro, q_sp, so + e * V16, outv[e]);
            else buf_st_nt(ro, q_sp, so + e * V16, outv[e]);
          }
      }
    }
    if (!(L2Q_LK_EXP & 8)) __syncthreads();           // slice tcur consumed
    if (more && !(L2Q_LK_EXP & 4)) {
      const int dst = (MU == 0 ? kLkOffT : offSc) + own;
#pragma unroll
      for (int e = 0; e < 9; ++e) *reinterpret_cast<double2*>(fr_lds + dst + e * kEnt) = pre[e];
    }
    if ((L2Q_LK_EXP & 32) && MODE != 2 && it >= c.lo) {
      const int so = MU * 9 * V16 + gcur;
#pragma unroll
      for (int e = 0; e < 9; ++e) {
        if (L2Q_LK_EXP & 16) buf_st(ro, q_sp, so + e * V16, outv[e]);
        else buf_st_nt(ro, q_sp, so + e * V16, outv[e]);
      }
    }
    cur ^= 1;
    if (!(L2Q_LK_EXP & 8)) __syncthreads();           // next slice in place
  }
}

#ifndef L2Q_LK_OCC
#define L2Q_LK_OCC 2
#endif
template <int MODE, int INM>
__global__ __launch_bounds__(kLkThreads, L2Q_LK_OCC) void su3_force_link_kernel(
    const double2* __restrict__ xn, Dims d, int nsb, int tsplit, int swz, double coef,
    const double2* vin, double2* out, int lo, int stagger) {
  // The two workgroups co-resident on a CU start together, do identical work and would stay in
  // lock-step: both at their slice barriers / halo loads at the same time, both wanting the fp64
  // pipe at the same time.  Delaying the second resident set ONCE (tuning force_stagger, units of
  // ~2k cycles) keeps them out of phase for the rest of the launch.
  if (stagger > 0 && blockIdx.x >= 256 && blockIdx.x < 512) {
    for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(32);
  }
  const long w = xcd_swizzle(blockIdx.x, gridDim.x, swz);
  const int per_chain = nsb * tsplit;
  const long c = w / per_chain;
  const int rr = (int)(w % per_chain);
  const int tc = rr / nsb, sb = rr % nsb;
  const int V = d.V, T = d.T;
  LkCtx k;
  k.d = d;
  k.V16 = V * 16;
  k.Vs16 = d.X * d.Y * d.Z * 16;
  k.tile0b = sb * kRS * 16;
  k.lt = threadIdx.x & (kRS - 1);
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kRS);     // provably wave-uniform
  const int tlen = (T + tsplit - 1) / tsplit;
  k.t0 = tc * tlen;
  k.t1 = min(T, k.t0 + tlen);
  const int chain_bytes = 36 * k.V16;
  k.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(xn + c * 36L * V), 0, chain_bytes, 0x00020000);
  k.ro = __builtin_amdgcn_make_buffer_rsrc((void*)(out + c * 36L * V), 0, chain_bytes, 0x00020000);
  k.rv = __builtin_amdgcn_make_buffer_rsrc((void*)((MODE != 0 ? vin : out) + c * 36L * V), 0, chain_bytes, 0x00020000);
  k.sp = sb * kRS + k.lt;
  {
    int q = k.sp;
    k.pz = q % d.Z; q /= d.Z;
    k.py = q % d.Y; q /= d.Y;
    k.px = q;
  }
  k.coef = coef;
  k.lo = lo;
  switch (wv) {
    case 0: force_link_sweep<MODE, 0, INM>(k); break;
    case 1: force_link_sweep<MODE, 1, INM>(k); break;
    case 2: force_link_sweep<MODE, 2, INM>(k); break;
    default: force_link_sweep<MODE, 3, INM>(k); break;
  }
}

template <int MODE, int INM>
static void launch_link_variant(const double2* xn, Dims d, int nb, int nsb, int tsplit, double coef,
                                const double2* vin, double2* out, hipStream_t st) {
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute((const void*)su3_force_link_kernel<MODE, INM>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLkLds);
  }
  hipLaunchKernelGGL((su3_force_link_kernel<MODE, INM>), dim3((unsigned)((long)nb * nsb * tsplit)),
                     dim3(kLkThreads), kLkLds, st, xn, d, nsb, tsplit, tuning().xcd_swizzle, coef, vin, out, 1,
                     tuning().force_stagger);
}

int force_link_inmask(const Dims& d) {
  int m = 0;
  if (kRS % d.Z == 0) m |= 4;
  if (kRS % (d.Y * d.Z) == 0) m |= 2;
  if (kRS % (d.X * d.Y * d.Z) == 0) m |= 1;
  return m;
}

bool force_link_applicable(const Dims& d) {
  return (d.X * d.Y * d.Z) % kRS == 0 && 36.0 * d.V * 16.0 < 2.0e9;
}

// g_x += coef * TAH(g_F) A^H  (l2q_su3_force_bwd; the staple sum A recomputed by the force sweep)
void launch_force_link_bwd(const double2* xn, Dims d, int nb, double coef, const double2* gf, double2* gx,
                           hipStream_t st) {
  const int Vs = d.X * d.Y * d.Z;
  const int nsb = Vs / kRS;
  int tsplit = (int)cdiv(1024, (long)nb * nsb);
  if (tsplit > d.T) tsplit = d.T;
  if (tsplit < 1) tsplit = 1;
  const int tlen = (int)cdiv(d.T, tsplit);
  tsplit = (int)cdiv(d.T, tlen);
  switch (force_link_inmask(d)) {
    case 7: launch_link_variant<2, 7>(xn, d, nb, nsb, tsplit, coef, gf, gx, st); break;
    case 6: launch_link_variant<2, 6>(xn, d, nb, nsb, tsplit, coef, gf, gx, st); break;
    case 4: launch_link_variant<2, 4>(xn, d, nb, nsb, tsplit, coef, gf, gx, st); break;
    default: launch_link_variant<2, 0>(xn, d, nb, nsb, tsplit, coef, gf, gx, st);
  }
}

// kick: out = vin + coef * F (vin == nullptr or out: in place)
void launch_force_link(bool kick, const double2* xn, Dims d, int nb, double coef, double2* out,
                       hipStream_t st, const double2* vin) {
  if (vin == nullptr) vin = out;
  const int Vs = d.X * d.Y * d.Z;
  const int nsb = Vs / kRS;
  int tsplit = (int)cdiv(1024, (long)nb * nsb);        // >= ~2 resident rounds of 2 x 256 workgroups
  if (tuning().force_tsplit > 0) tsplit = tuning().force_tsplit;
  if (tsplit > d.T) tsplit = d.T;
  if (tsplit < 1) tsplit = 1;
  const int tlen = (int)cdiv(d.T, tsplit);
  tsplit = (int)cdiv(d.T, tlen);
#define L2Q_LK_CASE(M)                                                                    \
  case M:                                                                                 \
    if (kick) launch_link_variant<1, M>(xn, d, nb, nsb, tsplit, coef, vin, out, st);           \
    else launch_link_variant<0, M>(xn, d, nb, nsb, tsplit, coef, vin, out, st);                \
    break;
  switch (force_link_inmask(d)) {
    L2Q_LK_CASE(7)
    L2Q_LK_CASE(6)
    L2Q_LK_CASE(4)
    default:
      if (kick) launch_link_variant<1, 0>(xn, d, nb, nsb, tsplit, coef, vin, out, st);
      else launch_link_variant<0, 0>(xn, d, nb, nsb, tsplit, coef, vin, out, st);
  }
#undef L2Q_LK_CASE
}

}  // namespace l2q
