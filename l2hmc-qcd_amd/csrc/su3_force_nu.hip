// su3_force_nu.hip -- SU(3) staple force, slice-resident sweep with the six staples of a link
// split by PLANE over wavefronts (gfx950).
//
//   F_mu(s) = coef * TAH( U_mu(s) * A_mu(s) ),   A = sum over nu != mu of the up and the down
//   staple in the (mu, nu) plane    (the reference: autograd of the Wilson action + projectTAH,
//   lattice/su3/pytorch/lattice.py:299-308)
//
// Register pressure decides this kernel.  One thread per link needs acc, t, a, b (3x3 complex
// fp64 = 36 VGPRs each) plus the carried t-staple and the prefetch: ~240 live registers, ONE
// wavefront per SIMD (su3_force_slice_kernel), VALU issuing 45 % of the time.  Splitting the 3x3
// algebra by rows (su3_force_rows.hip) gets 3 wavefronts per SIMD but every row re-reads the
// right-hand factors: 21 instead of 9 LDS reads per 72 FMAs, and tools/microbench/dfma_peak
// shows that mix LDS-bound at 53 % of the fp64 VALU peak.  Here the unit of work is a
// (link, plane) pair: wavefront (mu, j) computes the up + down staple of its links in the plane
// of mu and the j-th other direction -- four full 3x3 products on six operand matrices, the
// right-hand factors streamed row by row (peak: three matrices + one row live) -- so the LDS
// traffic is the minimal 54 entries per 432 FMAs AND three wavefronts fit a SIMD.  The three
// partial staple sums of a link meet in LDS: j = 0, 1 publish, j = 2 adds them, forms U*A (U taken
// from the tile before the barrier), TAH and stores the link.
//
// Workgroup = 64 spatial sites x 4 directions x 3 planes = 12 wavefronts, sweeping t; LDS holds
// the spatial links of the current and next slice, the t-links of the current slice and the
// exchange buffer (135 KiB).  Each thread prefetches one row of one link of the slice after next
// while it computes; every link is fetched from HBM once per sweep (+ the tile's halo from L2).
// Addressing as in su3_force_tile.hpp: everything wave-uniform is scalar, the per-lane part of an
// operand address is ONE 32-bit VGPR per neighbour site.
#include "su3_force_tile.hpp"

namespace l2q {

constexpr int kNuThreads = kRS * 12;
constexpr int kNuOffS0 = 0, kNuOffS1 = 3 * kPlaneB, kNuOffT = 6 * kPlaneB, kNuOffX = 7 * kPlaneB;
constexpr int kNuLds = kNuOffX + 2 * 4 * kPlaneB;       // exchange: [2 publishers][4 mu][9][kRS]

template <int INM>
__device__ __forceinline__ constexpr bool nu_in(int dir) { return dir == 0 ? true : ((INM >> (dir - 1)) & 1) != 0; }

struct NuCtx {
  __amdgpu_buffer_rsrc_t rs, ro;
  Dims d;
  int V16, Vs16, tile0b, lt, t0, t1;
  int sp, px, py, pz;
  double coef;
};

// the J-th direction other than MU, ascending
__host__ __device__ constexpr int other_dir(int mu, int j) { return j + (j >= mu ? 1 : 0); }

// One wavefront's sweep: direction MU, plane partner NU = other_dir(MU, J).
// MODE 0: out = coef * F;  MODE 1: out += coef * F
template <int MODE, int MU, int J, int INM>
__device__ __forceinline__ void force_nu_sweep(const NuCtx& c) {
  constexpr int NU = other_dir(MU, J);
  constexpr bool IN_MU = nu_in<INM>(MU), IN_NU = nu_in<INM>(NU);
  const Dims& d = c.d;
  const int T = d.T, V16 = c.V16, Vs16 = c.Vs16;
  const __amdgpu_buffer_rsrc_t rs = c.rs, ro = c.ro;
  // per-lane byte offsets of the neighbour sites (slice-independent)
  const int q_sp = c.sp * 16;
  int q_pmu = q_sp, mx = c.px, my = c.py, mz = c.pz;          // s + mu (spatial MU)
  if (MU != 0) {
    int q = hop(c.sp, c.px, c.py, c.pz, MU, +1, d);
    q_pmu = q * 16;
    mz = q % d.Z; q /= d.Z;
    my = q % d.Y; q /= d.Y;
    mx = q;
  }
  int q_pp = q_sp, q_pm = q_sp, q_pmm = q_pmu;                  // s + nu, s - nu, s + mu - nu
  if (NU != 0) {
    q_pp = hop(c.sp, c.px, c.py, c.pz, NU, +1, d) * 16;
    q_pm = hop(c.sp, c.px, c.py, c.pz, NU, -1, d) * 16;
    q_pmm = hop(q_pmu / 16, mx, my, mz, NU, -1, d) * 16;
  }
  const int lb = -c.tile0b;                           // LDS address of site q: region + lb + q * 16
  // prefetch duty: row J of link MU of this thread's site
  const int own_row = (MU == 0 ? 0 : (MU - 1) * kPlaneB) + 3 * J * kEnt + c.lt * 16;
  {
    const int ta = (c.t0 - 1 + T) % T;
    const int g0 = (MU * 9 + 3 * J) * V16;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double2 va = buf_ld(rs, q_sp, g0 + k * V16 + ta * Vs16);
      if (MU == 0) {
        *reinterpret_cast<double2*>(fr_lds + kNuOffT + own_row + k * kEnt) = va;
      } else {
        const double2 vb = buf_ld(rs, q_sp, g0 + k * V16 + (c.t0 % T) * Vs16);
        *reinterpret_cast<double2*>(fr_lds + kNuOffS0 + own_row + k * kEnt) = va;
        *reinterpret_cast<double2*>(fr_lds + kNuOffS1 + own_row + k * kEnt) = vb;
      }
    }
  }
  __syncthreads();
  int cur = 0;
  M3 carry;                                           // (MU spatial, NU = t): t-direction down staple
  if (MU != 0 && NU == 0) m3_zero(carry);
  const int niter = (c.t1 - c.t0) + 1;
  // the finishing wavefront of a link is j = 2: never the one that carries the t-staple (j = 0 for
  // spatial mu), so it has 36 registers to hold its link U across the barrier
  constexpr int JF = 2;
  const int xpub = kNuOffX + ((J < JF ? J : J - 1) * 4 + MU) * kPlaneB + c.lt * 16;
#pragma unroll 1
  for (int it = 0; it < niter; ++it) {
    const int tcur = (c.t0 - 1 + it + T) % T;
    const int tnext = (tcur + 1 == T) ? 0 : tcur + 1;
    const int offSc = cur ? kNuOffS1 : kNuOffS0;
    const int offSn = cur ? kNuOffS0 : kNuOffS1;
    const int gcur = tcur * Vs16, gnxt = tnext * Vs16;
    const bool more = it + 1 < niter;
    // operand of link direction rho at site offset qb in slice tcur / tnext
    auto oc = [&](int rho, int qb) {
      return Opnd<true>{(rho == 0 ? kNuOffT : offSc + (rho - 1) * kPlaneB) + lb + qb, qb, rho * 9 * V16 + gcur};
    };
    auto on = [&](int rho, int qb) {
      return Opnd<true>{offSn + (rho - 1) * kPlaneB + lb + qb, qb, rho * 9 * V16 + gnxt};
    };
    auto gco = [&](int rho, int qb) { return Opnd<false>{0, qb, rho * 9 * V16 + gcur}; };
    auto gno = [&](int rho, int qb) { return Opnd<false>{0, qb, rho * 9 * V16 + gnxt}; };
    double2 pre[3];
    if (more) {
      const int tp = MU == 0 ? tnext : ((tnext + 1 == T) ? 0 : tnext + 1);
      const int g0 = (MU * 9 + 3 * J) * V16 + tp * Vs16;
#pragma unroll
      for (int k = 0; k < 3; ++k) pre[k] = buf_ld(rs, q_sp, g0 + k * V16);
    }
    M3 acc;
    m3_zero(acc);
    if constexpr (MU == 0) {
      // t-link, plane (t, NU), NU spatial
      if (it > 0) {
        M3 a, t;
        // up:   U_nu(s+t) U_t(s+nu)^H U_nu(s)^H
        ld_m(a, on(NU, q_sp), rs, V16);
        if (IN_NU) mul_xh_stream<false>(t, a, oc(0, q_pp), rs, V16);
        else mul_xh_stream<false>(t, a, gco(0, q_pp), rs, V16);
        mac_stream<true>(acc, t, oc(NU, q_sp), rs, V16);
        // down: U_nu(s+t-nu)^H U_t(s-nu)^H U_nu(s-nu)
        if (IN_NU) {
          ld_m(a, on(NU, q_pm), rs, V16);
          mul_xh_stream<true>(t, a, oc(0, q_pm), rs, V16);
          mac_stream<false>(acc, t, oc(NU, q_pm), rs, V16);
        } else {
          ld_m(a, gno(NU, q_pm), rs, V16);
          mul_xh_stream<true>(t, a, gco(0, q_pm), rs, V16);
          mac_stream<false>(acc, t, gco(NU, q_pm), rs, V16);
        }
      }
    } else if constexpr (NU == 0) {
      // spatial link, plane (MU, t): the down staple was formed one slice earlier (carry)
      M3 a, t;
      if (IN_MU) ld_m(a, oc(0, q_pmu), rs, V16);              // U_t(tcur, s+mu): both staples
      else ld_m(a, gco(0, q_pmu), rs, V16);
      if (it > 0) {
        acc = carry;
        // up: U_t(s+mu) U_mu(s+t)^H U_t(s)^H
        mul_xh_stream<false>(t, a, on(MU, q_sp), rs, V16);
        mac_stream<true>(acc, t, oc(0, q_sp), rs, V16);
      }
      if (more) {
        // next slice's down staple of link (tnext, s, mu): U_t(tcur, s+mu)^H U_mu(tcur, s)^H U_t(tcur, s)
        mul_xh_stream<true>(t, a, oc(MU, q_sp), rs, V16);
        m3_zero(carry);
        mac_stream<false>(carry, t, oc(0, q_sp), rs, V16);
      }
    } else {
      // spatial link, spatial plane
      if (it > 0) {
        M3 a, t;
        // up:   U_nu(s+mu) U_mu(s+nu)^H U_nu(s)^H
        if (IN_MU) ld_m(a, oc(NU, q_pmu), rs, V16);
        else ld_m(a, gco(NU, q_pmu), rs, V16);
        if (IN_NU) mul_xh_stream<false>(t, a, oc(MU, q_pp), rs, V16);
        else mul_xh_stream<false>(t, a, gco(MU, q_pp), rs, V16);
        mac_stream<true>(acc, t, oc(NU, q_sp), rs, V16);
        // down: U_nu(s+mu-nu)^H U_mu(s-nu)^H U_nu(s-nu)
        if (IN_MU && IN_NU) ld_m(a, oc(NU, q_pmm), rs, V16);
        else ld_m(a, gco(NU, q_pmm), rs, V16);
        if (IN_NU) {
          mul_xh_stream<true>(t, a, oc(MU, q_pm), rs, V16);
          mac_stream<false>(acc, t, oc(NU, q_pm), rs, V16);
        } else {
          mul_xh_stream<true>(t, a, gco(MU, q_pm), rs, V16);
          mac_stream<false>(acc, t, gco(NU, q_pm), rs, V16);
        }
      }
    }
    if (J != JF && it > 0) {
#pragma unroll
      for (int e = 0; e < 9; ++e)
        *reinterpret_cast<double2*>(fr_lds + xpub + e * kEnt) = make_double2(acc.re[e], acc.im[e]);
    }
    // the finisher takes its own link out of LDS now (the slot is overwritten after the barrier):
    // U * A below then needs no trip to L2, which made this wavefront the last one at the next
    // barrier, every slice
    M3 u;
    if (J == JF && it > 0) ld_m(u, oc(MU, q_sp), rs, V16);
    __syncthreads();                                  // slice tcur consumed, partial sums published
    if (J == JF && it > 0) {
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int e = 0; e < 9; ++e) {
          const double2 dd = lds_ld(kNuOffX + (p * 4 + MU) * kPlaneB + c.lt * 16 + e * kEnt);
          acc.re[e] += dd.x; acc.im[e] += dd.y;
        }
    }
    if (more) {
      const int dst = (MU == 0 ? kNuOffT : offSc) + own_row;
#pragma unroll
      for (int k = 0; k < 3; ++k) *reinterpret_cast<double2*>(fr_lds + dst + k * kEnt) = pre[k];
    }
    cur ^= 1;
    __syncthreads();                                  // next slice in place, exchange buffer free
    if (J == JF && it > 0) {
      // W = U A, F = (W - W^H)/2 - tr(W - W^H)/6 (group/su3/pytorch/group.py:92-103), formed
      // entry by entry at the store
      M3 ua;
      const int so = MU * 9 * V16 + gcur;
      m3_mul_nn(ua, u, acc);
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
__global__ __launch_bounds__(kNuThreads) void su3_force_nu_kernel(
    const double2* __restrict__ xn, Dims d, int nsb, int tsplit, int swz, double coef,
    double2* __restrict__ out) {
  const long w = xcd_swizzle(blockIdx.x, gridDim.x, swz);
  const int per_chain = nsb * tsplit;
  const long c = w / per_chain;
  const int rr = (int)(w % per_chain);
  const int tc = rr / nsb, sb = rr % nsb;
  const int V = d.V, T = d.T;
  NuCtx k;
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
  k.sp = sb * kRS + k.lt;
  {
    int q = k.sp;
    k.pz = q % d.Z; q /= d.Z;
    k.py = q % d.Y; q /= d.Y;
    k.px = q;
  }
  k.coef = coef;
  // every wavefront runs the sweep specialised for its (direction, plane): identical barrier
  // sequence in all twelve
  switch (wv) {
    case 0: force_nu_sweep<MODE, 0, 0, INM>(k); break;
    case 1: force_nu_sweep<MODE, 0, 1, INM>(k); break;
    case 2: force_nu_sweep<MODE, 0, 2, INM>(k); break;
    case 3: force_nu_sweep<MODE, 1, 0, INM>(k); break;
    case 4: force_nu_sweep<MODE, 1, 1, INM>(k); break;
    case 5: force_nu_sweep<MODE, 1, 2, INM>(k); break;
    case 6: force_nu_sweep<MODE, 2, 0, INM>(k); break;
    case 7: force_nu_sweep<MODE, 2, 1, INM>(k); break;
    case 8: force_nu_sweep<MODE, 2, 2, INM>(k); break;
    case 9: force_nu_sweep<MODE, 3, 0, INM>(k); break;
    case 10: force_nu_sweep<MODE, 3, 1, INM>(k); break;
    default: force_nu_sweep<MODE, 3, 2, INM>(k); break;
  }
}

template <int MODE, int INM>
static void launch_nu_variant(const double2* xn, Dims d, int nb, int nsb, int tsplit, double coef,
                              double2* out, hipStream_t st) {
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute((const void*)su3_force_nu_kernel<MODE, INM>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, kNuLds);
  }
  hipLaunchKernelGGL((su3_force_nu_kernel<MODE, INM>), dim3((unsigned)((long)nb * nsb * tsplit)),
                     dim3(kNuThreads), kNuLds, st, xn, d, nsb, tsplit, tuning().xcd_swizzle, coef, out);
}

int force_nu_inmask(const Dims& d) {
  int m = 0;
  if (kRS % d.Z == 0) m |= 4;                               // +-z neighbours stay in the tile
  if (kRS % (d.Y * d.Z) == 0) m |= 2;                       // +-y
  if (kRS % (d.X * d.Y * d.Z) == 0) m |= 1;                 // +-x
  return m;
}

bool force_nu_applicable(const Dims& d) {
  return (d.X * d.Y * d.Z) % kRS == 0 && 36.0 * d.V * 16.0 < 2.0e9;
}

void launch_force_nu(bool kick, const double2* xn, Dims d, int nb, double coef, double2* out,
                     hipStream_t st) {
  const int Vs = d.X * d.Y * d.Z;
  const int nsb = Vs / kRS;
  int tsplit = (int)cdiv(512, (long)nb * nsb);         // >= ~2 resident rounds of 256 CUs
  if (tsplit > d.T) tsplit = d.T;
  if (tsplit < 1) tsplit = 1;
  const int tlen = (int)cdiv(d.T, tsplit);
  tsplit = (int)cdiv(d.T, tlen);
#define L2Q_NU_CASE(M)                                                                    \
  case M:                                                                                 \
    if (kick) launch_nu_variant<1, M>(xn, d, nb, nsb, tsplit, coef, out, st);             \
    else launch_nu_variant<0, M>(xn, d, nb, nsb, tsplit, coef, out, st);                  \
    break;
  switch (force_nu_inmask(d)) {
    L2Q_NU_CASE(7)
    L2Q_NU_CASE(6)
    L2Q_NU_CASE(4)
    default:
      if (kick) launch_nu_variant<1, 0>(xn, d, nb, nsb, tsplit, coef, out, st);
      else launch_nu_variant<0, 0>(xn, d, nb, nsb, tsplit, coef, out, st);
  }
#undef L2Q_NU_CASE
}

}  // namespace l2q
