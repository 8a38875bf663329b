// su3_force_pair.hip -- SU(3) staple force, slice-resident sweep, one thread per link, TWO adjacent
// x-planes per workgroup (gfx950).
//
//   F_mu(s) = coef * TAH( U_mu(s) * A_mu(s) ),   A = sum over nu != mu of the up and the down
//   staple in the (mu, nu) plane    (the reference: autograd of the Wilson action + projectTAH,
//   lattice/su3/pytorch/lattice.py:299-308)
//
// Same arithmetic, operation order and per-thread code as su3_force_link.hip (results are
// bit-identical), different tile.  There a workgroup owns 64 sites of ONE x-plane and every
// x-neighbour comes from L2: 17 neighbour matrices per site and slice next to the 4 own links
// (PMC at cfg-4: 1.43 GB fetched for 0.60 GB of links, the kernel waits on those loads for 29 % of
// its wavefront time with only two wavefronts per SIMD to hide them).  Here a workgroup is 512
// threads = 2 groups of 64 sites x 4 directions; group 1 is group 0 shifted by one x-plane, so for
// group 0 the +x neighbours and for group 1 the -x neighbours are in LDS: half of the x-halo
// requests (8.5 instead of 17 matrices per site and slice) disappear.  LDS: 7 link planes of 128
// sites = 126 KiB, one workgroup per CU -- the same 8 wavefronts and the same registers per CU as
// the two 256-thread workgroups of the link kernel.
//
// A wavefront = (group g, direction mu): whether a neighbour is inside the tile is a template
// constant per wavefront (x: +x for g = 0, -x for g = 1; y, z: when the 64-site group spans the
// whole extent, as in the link kernel).  Lattices: Y*Z a multiple of 64, X even (8^4: a group is a
// whole (y,z) plane; 16^4: four y-rows of a plane).
#include "su3_force_tile.hpp"

namespace l2q {

#define L2Q_PR_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

constexpr int kPrSites = 2 * kRS;                 // sites per workgroup
constexpr int kPrThreads = kPrSites * 4;
constexpr int kPrEnt = kPrSites * 16;             // bytes between entries of a link in LDS
constexpr int kPrPlane = 9 * kPrEnt;
constexpr int kPrGrp = kRS * 16;                  // byte offset of group 1 inside an entry row
constexpr int kPrOffS0 = 0, kPrOffS1 = 3 * kPrPlane, kPrOffT = 6 * kPrPlane;
constexpr int kPrLds = 7 * kPrPlane;              // 129 024 B

// neighbour in direction dir (1 = x, 2 = y, 3 = z) inside the tile?  G = group of the wavefront
template <int INM, int G>
__device__ __forceinline__ constexpr bool pr_in_p(int dir) {
  return dir == 0 ? true : dir == 1 ? (G == 0) : ((INM >> (dir - 1)) & 1) != 0;
}
template <int INM, int G>
__device__ __forceinline__ constexpr bool pr_in_m(int dir) {
  return dir == 0 ? true : dir == 1 ? (G == 1) : ((INM >> (dir - 1)) & 1) != 0;
}

struct PrCtx {
  __amdgpu_buffer_rsrc_t rs, ro, rv;
  Dims d;
  int V16, Vs16, lt, t0, t1;
  int lb_own, lb_oth;     // LDS byte address of a site of the own / the other group = lb + 16 * site
  int sp, px, py, pz;
  double coef;
  int lo;                 // = 1, a kernel ARGUMENT (see su3_force_link.hip)
};

using OpL = Opnd<true, kPrEnt>;
using OpG = Opnd<false, kPrEnt>;

template <int MODE, int MU, int INM, int G>
__device__ __forceinline__ void force_pair_sweep(const PrCtx& c) {
  constexpr bool IN_MU = pr_in_p<INM, G>(MU);
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
  // LDS bases: a hop in x changes the group, hops in y / z stay inside it
  const int lbo = c.lb_own, lbx = c.lb_oth;
  auto lb_hop = [&](int dir) { return dir == 1 ? lbx : lbo; };
  auto lb_hop2 = [&](int da, int db) { return ((da == 1) != (db == 1)) ? lbx : lbo; };
  const int own = (MU == 0 ? 0 : (MU - 1) * kPrPlane) + lbo + q_sp;   // this thread's link in a slot
  {
    const int ta = (c.t0 - 1 + T) % T;
#pragma unroll
    for (int e = 0; e < 9; ++e) {
      const double2 va = buf_ld(rs, q_sp, (MU * 9 + e) * V16 + ta * Vs16);
      if (MU == 0) {
        *reinterpret_cast<double2*>(fr_lds + kPrOffT + own + e * kPrEnt) = va;
      } else {
        const double2 vb = buf_ld(rs, q_sp, (MU * 9 + e) * V16 + (c.t0 % T) * Vs16);
        *reinterpret_cast<double2*>(fr_lds + kPrOffS0 + own + e * kPrEnt) = va;
        *reinterpret_cast<double2*>(fr_lds + kPrOffS1 + own + e * kPrEnt) = vb;
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
    const int offSc = cur ? kPrOffS1 : kPrOffS0;
    const int offSn = cur ? kPrOffS0 : kPrOffS1;
    const int gcur = tcur * Vs16, gnxt = tnext * Vs16;
    const bool more = it + 1 < niter;
    // operands of the current (oc) / next (on) slice in LDS; lb = LDS base of the operand's group
    auto oc = [&](int rho, int qb, int lb) {
      return OpL{(rho == 0 ? kPrOffT : offSc + (rho - 1) * kPrPlane) + lb + qb, qb, rho * 9 * V16 + gcur};
    };
    auto on = [&](int rho, int qb, int lb) {
      return OpL{offSn + (rho - 1) * kPrPlane + lb + qb, qb, rho * 9 * V16 + gnxt};
    };
    auto gco = [&](int rho, int qb) { return OpG{0, qb, rho * 9 * V16 + gcur}; };
    auto gno = [&](int rho, int qb) { return OpG{0, qb, rho * 9 * V16 + gnxt}; };
    // prefetch the thread's own link of the slice that enters LDS after this iteration
    double2 pre[9];
    if (more) {
      const int tp = MU == 0 ? tnext : ((tnext + 1 == T) ? 0 : tnext + 1);
#pragma unroll
      for (int e = 0; e < 9; ++e) pre[e] = buf_ld(rs, q_sp, (MU * 9 + e) * V16 + tp * Vs16);
    }
    M3 acc;
    m3_zero(acc);
    if constexpr (MU == 0) {
      if (it >= c.lo) {
#pragma unroll
        for (int nu = 1; nu < 4; ++nu) {
          M3 a, t;
          if (it >= c.lo) {
          // up:   U_nu(s+t) U_t(s+nu)^H U_nu(s)^H
          ld_m(a, on(nu, q_sp, lbo), rs, V16);
          if (pr_in_p<INM, G>(nu)) mul_xh_stream<false>(t, a, oc(0, q_pp[nu], lb_hop(nu)), rs, V16);
          else mul_xh_stream<false>(t, a, gco(0, q_pp[nu]), rs, V16);
          mac_stream<true>(acc, t, oc(nu, q_sp, lbo), rs, V16);
          }
          L2Q_PR_FENCE();
          if (it < c.lo) continue;
          // down: U_nu(s+t-nu)^H U_t(s-nu)^H U_nu(s-nu)
          if (pr_in_m<INM, G>(nu)) {
            ld_m(a, on(nu, q_pm[nu], lb_hop(nu)), rs, V16);
            mul_xh_stream<true>(t, a, oc(0, q_pm[nu], lb_hop(nu)), rs, V16);
            mac_stream<false>(acc, t, oc(nu, q_pm[nu], lb_hop(nu)), rs, V16);
            L2Q_PR_FENCE();
          } else {
            ld_m(a, gno(nu, q_pm[nu]), rs, V16);
            mul_xh_stream<true>(t, a, gco(0, q_pm[nu]), rs, V16);
            mac_stream<false>(acc, t, gco(nu, q_pm[nu]), rs, V16);
            L2Q_PR_FENCE();
          }
        }
      }
    } else {
      {
        // plane (MU, t): the down staple was formed one slice earlier (carry)
        M3 a, t;
        if (IN_MU) ld_m(a, oc(0, q_pmu, lb_hop(MU)), rs, V16);   // U_t(tcur, s+mu): both staples
        else ld_m(a, gco(0, q_pmu), rs, V16);
        if (it >= c.lo) {
          acc = carry;
          // up: U_t(s+mu) U_mu(s+t)^H U_t(s)^H
          mul_xh_stream<false>(t, a, on(MU, q_sp, lbo), rs, V16);
          mac_stream<true>(acc, t, oc(0, q_sp, lbo), rs, V16);
          L2Q_PR_FENCE();
        }
        if (more) {
          // next slice's down staple: U_t(tcur, s+mu)^H U_mu(tcur, s)^H U_t(tcur, s)
          mul_xh_stream<true>(t, a, oc(MU, q_sp, lbo), rs, V16);
          m3_zero(carry);
          mac_stream<false>(carry, t, oc(0, q_sp, lbo), rs, V16);
          L2Q_PR_FENCE();
        }
      }
      if (it >= c.lo) {
#pragma unroll
        for (int nu = 1; nu < 4; ++nu) {
          if (nu == MU) continue;
          M3 a, t;
          if (it >= c.lo) {
          // up:   U_nu(s+mu) U_mu(s+nu)^H U_nu(s)^H
          if (IN_MU) ld_m(a, oc(nu, q_pmu, lb_hop(MU)), rs, V16);
          else ld_m(a, gco(nu, q_pmu), rs, V16);
          if (pr_in_p<INM, G>(nu)) mul_xh_stream<false>(t, a, oc(MU, q_pp[nu], lb_hop(nu)), rs, V16);
          else mul_xh_stream<false>(t, a, gco(MU, q_pp[nu]), rs, V16);
          mac_stream<true>(acc, t, oc(nu, q_sp, lbo), rs, V16);
          }
          L2Q_PR_FENCE();
          if (it < c.lo) continue;
          // down: U_nu(s+mu-nu)^H U_mu(s-nu)^H U_nu(s-nu)
          if (IN_MU && pr_in_m<INM, G>(nu)) ld_m(a, oc(nu, q_pmm[nu], lb_hop2(MU, nu)), rs, V16);
          else ld_m(a, gco(nu, q_pmm[nu]), rs, V16);
          if (pr_in_m<INM, G>(nu)) {
            mul_xh_stream<true>(t, a, oc(MU, q_pm[nu], lb_hop(nu)), rs, V16);
            mac_stream<false>(acc, t, oc(nu, q_pm[nu], lb_hop(nu)), rs, V16);
            L2Q_PR_FENCE();
          } else {
            mul_xh_stream<true>(t, a, gco(MU, q_pm[nu]), rs, V16);
            mac_stream<false>(acc, t, gco(nu, q_pm[nu]), rs, V16);
            L2Q_PR_FENCE();
          }
        }
      }
    }
    if (it >= c.lo) {
      // W = U A with U streamed by rows from the tile; F = (W - W^H)/2 - tr(W - W^H)/6
      // (group/su3/pytorch/group.py:92-103), formed entry by entry at the store (nt accesses: the
      // output is touched once)
      M3 ua;
      const OpL uo = oc(MU, q_sp, lbo);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        R3 ur;
        ld_row<true, kPrEnt>(ur, uo.lds, rs, uo.voff, uo.soff, V16, i);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          double sr = 0.0, si = 0.0;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            sr = fma(ur.re[k], acc.re[3 * k + j], sr); sr = fma(-ur.im[k], acc.im[3 * k + j], sr);
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
          const double fr = 0.5 * (ua.re[e] - ua.re[et]);
          double fi = 0.5 * (ua.im[e] + ua.im[et]);
          if (i == j) fi -= tri;
          double2 v2 = make_double2(c.coef * fr, c.coef * fi);
          if (MODE == 1) {
            const double2 o = buf_ld_nt(rv, q_sp, so + e * V16);
            v2.x += o.x; v2.y += o.y;
          }
          buf_st_nt(ro, q_sp, so + e * V16, v2);
        }
    }
    __syncthreads();                                  // slice tcur consumed
    if (more) {
      const int dst = (MU == 0 ? kPrOffT : offSc) + own;
#pragma unroll
      for (int e = 0; e < 9; ++e) *reinterpret_cast<double2*>(fr_lds + dst + e * kPrEnt) = pre[e];
    }
    cur ^= 1;
    __syncthreads();                                  // next slice in place
  }
}

template <int MODE, int INM>
__global__ __launch_bounds__(kPrThreads, 1) void su3_force_pair_kernel(
    const double2* __restrict__ xn, Dims d, int npair, int bpp, int tsplit, int swz, double coef,
    const double2* vin, double2* out, int lo) {
  const long w = xcd_swizzle(blockIdx.x, gridDim.x, swz);
  const int per_chain = npair * tsplit;
  const long c = w / per_chain;
  const int rr = (int)(w % per_chain);
  const int tc = rr / npair, pb = rr % npair;
  const int V = d.V, T = d.T;
  const int YZ = d.Y * d.Z;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kRS);     // provably wave-uniform
  const int g = wv >> 2;
  // pair pb = (x-pair xp, block yb of the plane): group 0 at x = 2 xp, group 1 one plane further
  const int xp = pb / bpp, yb = pb % bpp;
  const int site0 = 2 * xp * YZ + yb * kRS;                // first site of group 0
  PrCtx k;
  k.d = d;
  k.V16 = V * 16;
  k.Vs16 = d.X * YZ * 16;
  k.lt = threadIdx.x & (kRS - 1);
  // LDS byte address of site q of group gg inside an entry row: gg * kPrGrp + 16 (q - site0 - gg YZ)
  const int lb0 = -site0 * 16, lb1 = kPrGrp - (site0 + YZ) * 16;
  k.lb_own = g ? lb1 : lb0;
  k.lb_oth = g ? lb0 : lb1;
  const int tlen = (T + tsplit - 1) / tsplit;
  k.t0 = tc * tlen;
  k.t1 = min(T, k.t0 + tlen);
  const int chain_bytes = 36 * k.V16;
  k.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(xn + c * 36L * V), 0, chain_bytes, 0x00020000);
  k.ro = __builtin_amdgcn_make_buffer_rsrc((void*)(out + c * 36L * V), 0, chain_bytes, 0x00020000);
  k.rv = __builtin_amdgcn_make_buffer_rsrc((void*)((MODE == 1 ? vin : out) + c * 36L * V), 0, chain_bytes, 0x00020000);
  k.sp = site0 + g * YZ + k.lt;
  {
    int q = k.sp;
    k.pz = q % d.Z; q /= d.Z;
    k.py = q % d.Y; q /= d.Y;
    k.px = q;
  }
  k.coef = coef;
  k.lo = lo;
  switch (wv) {
    case 0: force_pair_sweep<MODE, 0, INM, 0>(k); break;
    case 1: force_pair_sweep<MODE, 1, INM, 0>(k); break;
    case 2: force_pair_sweep<MODE, 2, INM, 0>(k); break;
    case 3: force_pair_sweep<MODE, 3, INM, 0>(k); break;
    case 4: force_pair_sweep<MODE, 0, INM, 1>(k); break;
    case 5: force_pair_sweep<MODE, 1, INM, 1>(k); break;
    case 6: force_pair_sweep<MODE, 2, INM, 1>(k); break;
    default: force_pair_sweep<MODE, 3, INM, 1>(k); break;
  }
}

template <int MODE, int INM>
static void launch_pair_variant(const double2* xn, Dims d, int nb, int npair, int bpp, int tsplit, double coef,
                                const double2* vin, double2* out, hipStream_t st) {
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute((const void*)su3_force_pair_kernel<MODE, INM>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, kPrLds);
  }
  hipLaunchKernelGGL((su3_force_pair_kernel<MODE, INM>), dim3((unsigned)((long)nb * npair * tsplit)),
                     dim3(kPrThreads), kPrLds, st, xn, d, npair, bpp, tsplit, tuning().xcd_swizzle, coef, vin,
                     out, 1);
}

// bits: 4 = z, 2 = y inside a 64-site group (x is decided per wavefront)
int force_pair_inmask(const Dims& d) {
  int m = 0;
  if (kRS % d.Z == 0) m |= 4;
  if (kRS % (d.Y * d.Z) == 0) m |= 2;
  return m;
}

bool force_pair_applicable(const Dims& d) {
  return (d.Y * d.Z) % kRS == 0 && d.X % 2 == 0 && 36.0 * d.V * 16.0 < 2.0e9;
}

// kick: out = vin + coef * F (vin == nullptr or out: in place)
void launch_force_pair(bool kick, const double2* xn, Dims d, int nb, double coef, double2* out,
                       hipStream_t st, const double2* vin) {
  if (vin == nullptr) vin = out;
  const int bpp = d.Y * d.Z / kRS;
  const int npair = (d.X / 2) * bpp;
  int tsplit = (int)cdiv(1024, (long)nb * npair);      // >= ~4 resident rounds of 256 workgroups
  if (tsplit > d.T) tsplit = d.T;
  if (tsplit < 1) tsplit = 1;
  const int tlen = (int)cdiv(d.T, tsplit);
  tsplit = (int)cdiv(d.T, tlen);
#define L2Q_PR_CASE(M)                                                                         \
  case M:                                                                                      \
    if (kick) launch_pair_variant<1, M>(xn, d, nb, npair, bpp, tsplit, coef, vin, out, st);    \
    else launch_pair_variant<0, M>(xn, d, nb, npair, bpp, tsplit, coef, vin, out, st);         \
    break;
  switch (force_pair_inmask(d)) {
    L2Q_PR_CASE(6)
    L2Q_PR_CASE(4)
    default:
      if (kick) launch_pair_variant<1, 0>(xn, d, nb, npair, bpp, tsplit, coef, vin, out, st);
      else launch_pair_variant<0, 0>(xn, d, nb, npair, bpp, tsplit, coef, vin, out, st);
  }
#undef L2Q_PR_CASE
}

}  // namespace l2q
