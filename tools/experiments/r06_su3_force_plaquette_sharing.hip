// su3_force_plane.hip -- SU(3) staple force with the plaquettes SHARED between the links they close:
// slice-resident sweep, one wavefront per PLANE of the tile's 64 sites (gfx950).
//
//   F_mu(s) = coef * TAH( U_mu(s) * A_mu(s) ),   A = the six staples of the link
//   (the reference: autograd of the Wilson action + projectTAH, lattice/su3/pytorch/lattice.py:299-308)
//
// The thread-per-link kernels (su3_force_link.hip) form U A from 13 products per link = 52 per site and read
// 19 operand matrices per link (76 per site) from LDS / L2: measured, the fp64 FMAs take 56 % of the SIMD
// cycles and the LDS pipe is as loaded as the VALU.  Here the work is organised by plaquette instead.  With
//     L_ab = U_a(s) U_b(s+a),   L_ba = U_b(s) U_a(s+b),   P = L_ab L_ba^H          (plane {a, b} based at s)
// the FOUR links of the plaquette get their contributions from it (TAH is linear, so every contribution is
// reduced to its 8 real components before it leaves the thread):
//     link (s,   a): + TAH(P)                         link (s,   b): - TAH(P)
//     link (s+b, a): TAH( U_a(s+b) L_ab^H U_b(s) )    link (s+a, b): TAH( U_b(s+a) L_ba^H U_a(s) )
// = 7 products on 4 operand matrices per plane and site: 42 products and 24 operand reads per site.  A
// contribution whose target link lies in the next slice is carried in registers to the next iteration (same
// spatial site); one whose target lies outside the tile (the x direction on 8^4, x and y on 16^4) is not
// sent: the target's workgroup recomputes it from the neighbouring plane (3 products, operands from L2):
// 45 products per site on 8^4, 48 on 16^4, and 10 instead of 17 neighbour matrices per site from L2.
//
// Accumulation is deterministic without atomics: two LDS banks of [dir][8][site] doubles, one for the
// contributions of a link's own site, one for those arriving from a neighbour (or the previous slice), and
// three rounds per slice separated by barriers -- in round r a link receives exactly one deposit per bank,
// from the plane {a, b_r} with b_r the r-th other direction of a.  After the third round the wavefront of
// direction `dir` sums the two banks, scales, expands the 8 components to the 3 x 3 anti-Hermitian matrix
// and stores it.  LDS: links of ONE slice (36 KiB; the next slice's links arrive through the registers of the
// wavefronts that need them as operands anyway) + 2 x 16 KiB banks = 68 KiB: two workgroups of six
// wavefronts per CU = 3 wavefronts per SIMD at <= 168 registers.
//
// Results agree with the thread-per-link kernels to rounding (sum of TAHs instead of TAH of the sum), not bit
// for bit; they are the same bits from run to run.
#include "su3_force_tile.hpp"
#include <type_traits>

namespace l2q {

#ifdef L2Q_PL_NOFENCE
#define L2Q_PL_FENCE() do { } while (0)
#else
#define L2Q_PL_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#endif

// Timing experiments (tools/ab_build.sh ... -DL2Q_PL_EXP=bits; results WRONG, only the clock is read):
//   1 no barriers between the deposit rounds   2 no neighbour-term recomputation (halo products skipped)
//   4 no deposits / gather arithmetic (stores kept)
#ifndef L2Q_PL_EXP
#define L2Q_PL_EXP 0
#endif
#ifndef L2Q_PL_OCC
#define L2Q_PL_OCC 3
#endif
constexpr int kPlThreads = kRS * 6;
constexpr int kPlBankB = 4 * 4 * kEnt;                 // [dir][pair of components][site] double2 = 16 KiB
constexpr int kPlOffBank = 4 * kPlaneB;                // after the links [dir][entry][site]
constexpr int kPlOffCarry = kPlOffBank + 2 * kPlBankB;   // temporal planes: contribution to the NEXT slice [plane][4][site]
constexpr int kPlLds = kPlOffCarry + 3 * 4 * kEnt;

struct T8 {
  double v[8];       // (re, im) of entries (0,1), (0,2), (1,2); Im of entries (0,0), (1,1) (traceless)
};

template <int INM>
__device__ __forceinline__ constexpr bool pl_in(int dir) { return ((INM >> (dir - 1)) & 1) != 0; }

// TAH(X Y) (ADJ_Y = false) or TAH(X Y^H): only the entries the projection keeps are formed (90 of 108 FMAs)
template <bool ADJ_Y>
__device__ __forceinline__ void tah_prod(T8& r, const M3& x, const M3& y) {
  double wr[9], wi[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double sr = 0.0, si = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double ar = x.re[3 * i + k], ai = x.im[3 * i + k];
        const double br = ADJ_Y ? y.re[3 * j + k] : y.re[3 * k + j];
        const double bi = ADJ_Y ? -y.im[3 * j + k] : y.im[3 * k + j];
        if (i != j) { sr = fma(ar, br, sr); sr = fma(-ai, bi, sr); }
        si = fma(ar, bi, si); si = fma(ai, br, si);
      }
      wr[3 * i + j] = sr; wi[3 * i + j] = si;
    }
  const double tri = (wi[0] + wi[4] + wi[8]) / 3.0;
  r.v[0] = 0.5 * (wr[1] - wr[3]); r.v[1] = 0.5 * (wi[1] + wi[3]);
  r.v[2] = 0.5 * (wr[2] - wr[6]); r.v[3] = 0.5 * (wi[2] + wi[6]);
  r.v[4] = 0.5 * (wr[5] - wr[7]); r.v[5] = 0.5 * (wi[5] + wi[7]);
  r.v[6] = wi[0] - tri; r.v[7] = wi[4] - tri;
}

// TAH(X C) with the right factor C streamed by rows from LDS / the chain (row k of C meets column k of X)
template <bool IN>
__device__ __forceinline__ void tah_rstream(T8& r, const M3& x, const Opnd<IN>& c, __amdgpu_buffer_rsrc_t rs,
                                            int V16) {
  double wr[9], wi[9];
#pragma unroll
  for (int e = 0; e < 9; ++e) { wr[e] = 0.0; wi[e] = 0.0; }
  R3 rows[2];
  ld_row<IN>(rows[0], c.lds, rs, c.voff, c.soff, V16, 0);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (k < 2) {
      ld_row<IN>(rows[(k + 1) & 1], c.lds, rs, c.voff, c.soff, V16, k + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    const R3& cr = rows[k & 1];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const double ar = x.re[3 * i + k], ai = x.im[3 * i + k];
        if (i != j) { wr[3 * i + j] = fma(ar, cr.re[j], wr[3 * i + j]); wr[3 * i + j] = fma(-ai, cr.im[j], wr[3 * i + j]); }
        wi[3 * i + j] = fma(ar, cr.im[j], wi[3 * i + j]); wi[3 * i + j] = fma(ai, cr.re[j], wi[3 * i + j]);
      }
  }
  const double tri = (wi[0] + wi[4] + wi[8]) / 3.0;
  r.v[0] = 0.5 * (wr[1] - wr[3]); r.v[1] = 0.5 * (wi[1] + wi[3]);
  r.v[2] = 0.5 * (wr[2] - wr[6]); r.v[3] = 0.5 * (wi[2] + wi[6]);
  r.v[4] = 0.5 * (wr[5] - wr[7]); r.v[5] = 0.5 * (wi[5] + wi[7]);
  r.v[6] = wi[0] - tri; r.v[7] = wi[4] - tri;
}

struct PlCtx {
  __amdgpu_buffer_rsrc_t rs, ro;
  Dims d;
  int V16, Vs16, tile0b, lt, t0, t1;
  int sp, px, py, pz;
  double coef;
  int lo;
};

// bank (kind 0: own site, 1: from a neighbour / the previous slice), direction, tile-local site byte offset
__device__ __forceinline__ void bank_add(int kind, int dir, int ltb, const T8& c, double sgn) {
  const int a = kPlOffBank + kind * kPlBankB + dir * (4 * kEnt) + ltb;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    double2 o = lds_ld(a + k * kEnt);
    o.x = fma(sgn, c.v[2 * k], o.x);
    o.y = fma(sgn, c.v[2 * k + 1], o.y);
    *reinterpret_cast<double2*>(fr_lds + a + k * kEnt) = o;
  }
}

__device__ __forceinline__ void t8_add(T8& a, const T8& b, double sgn) {
#pragma unroll
  for (int k = 0; k < 8; ++k) a.v[k] = fma(sgn, b.v[k], a.v[k]);
}

// One wavefront's sweep: plane {A, B}, A < B; A == 0 is a temporal plane.  W = the wavefront's index (its
// duties in the slice refresh and the final gather).
template <int A, int B, int W, int INM>
__device__ __forceinline__ void force_plane_sweep(const PlCtx& c) {
  constexpr bool TP = A == 0;                                  // temporal plane
  constexpr bool IN_A = TP ? true : pl_in<INM>(A);             // s + a reachable inside the workgroup
  constexpr bool IN_B = pl_in<INM>(B);
  constexpr int R_AB = B - 1;                                  // round of the deposits to direction A
  constexpr int R_BA = A;                                      // round of the deposits to direction B
  const Dims& d = c.d;
  const int T = d.T, V16 = c.V16, Vs16 = c.Vs16;
  const __amdgpu_buffer_rsrc_t rs = c.rs, ro = c.ro;
  const int q_sp = c.sp * 16;
  const int lb = -c.tile0b;
  const int ltb = c.lt * 16;
  // neighbour sites (byte offsets inside a slice)
  const int q_pb = hop(c.sp, c.px, c.py, c.pz, B, +1, d) * 16;
  const int q_mb = hop(c.sp, c.px, c.py, c.pz, B, -1, d) * 16;
  int q_pa = q_sp, q_ma = q_sp, q_mb_pa = q_mb, q_ma_pb = q_sp;
  if (!TP) {
    q_pa = hop(c.sp, c.px, c.py, c.pz, A, +1, d) * 16;
    q_ma = hop(c.sp, c.px, c.py, c.pz, A, -1, d) * 16;
    {
      int q = q_mb / 16;                                       // s - b, then + a
      const int z = q % d.Z; q /= d.Z;
      const int y = q % d.Y; q /= d.Y;
      q_mb_pa = hop(q_mb / 16, q, y, z, A, +1, d) * 16;
    }
    {
      int q = q_ma / 16;                                       // s - a, then + b
      const int z = q % d.Z; q /= d.Z;
      const int y = q % d.Y; q /= d.Y;
      q_ma_pb = hop(q_ma / 16, q, y, z, B, +1, d) * 16;
    }
  }
  // ---- first slice (t0 - 1) into LDS: entries 6 W .. 6 W + 5 of the 36 (dir * 9 + e) of this thread's site
  {
    const int ta = (c.t0 - 1 + T) % T;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double2 va = buf_ld(rs, q_sp, (6 * W + k) * V16 + ta * Vs16);
      *reinterpret_cast<double2*>(fr_lds + (6 * W + k) * kEnt + ltb) = va;
    }
    // banks and carry slots start at zero
    for (int i = threadIdx.x; i < (2 * kPlBankB + 3 * 4 * kEnt) / 16; i += kPlThreads)
      *reinterpret_cast<double2*>(fr_lds + kPlOffBank + i * 16) = make_double2(0.0, 0.0);
  }
  __syncthreads();
  // (temporal planes: the contribution to link (s + t, b) waits in this thread's LDS carry slot for the next iteration)
  const int niter = (c.t1 - c.t0) + 1;
  // (Peeling the prologue and the last iteration -- `full` / `more` as compile-time constants of three copies of
  // the body -- was tried: 320-554 spilled registers instead of 82-136.)
#pragma unroll 1
  for (int it = 0; it < niter; ++it) {
    const int tcur = (c.t0 - 1 + it + T) % T;
    const bool more = it + 1 < niter;
    const bool full = it >= c.lo;
    const int tnext = (tcur + 1 == T) ? 0 : tcur + 1;
    const int tnext2 = (tnext + 1 == T) ? 0 : tnext + 1;
    const int gcur = tcur * Vs16, gnxt = tnext * Vs16;
    auto lo_ = [&](int rho, int qb) { return Opnd<true>{rho * kPlaneB + lb + qb, qb, rho * 9 * V16 + gcur}; };
    auto go_ = [&](int rho, int qb) { return Opnd<false>{0, qb, rho * 9 * V16 + gcur}; };
    auto gn_ = [&](int rho, int qb) { return Opnd<false>{0, qb, rho * 9 * V16 + gnxt}; };
    // slice refresh duties (written to LDS after the compute phase): the spatial planes' wavefronts W = 3, 4, 5 fetch
    // the next slice's links of direction W - 2, the temporal planes' wavefronts three entries each of its
    // t-links.  (Requesting the 18 registers of a spatial wavefront later in the iteration -- a second call site --
    // makes hipcc spill 700+ registers: conditionally defined arrays.)
    double2 pt[TP ? 3 : 9];
    auto refresh_req = [&]() {
      if (more) {
        if (TP) {
#pragma unroll
          for (int k = 0; k < 3; ++k) pt[k] = buf_ld(rs, q_sp, (3 * W + k) * V16 + gnxt);
        } else {
#pragma unroll
          for (int e = 0; e < 9; ++e) pt[e] = buf_ld(rs, q_sp, ((W - 2) * 9 + e) * V16 + gnxt);
        }
      }
    };
    refresh_req();
    // Left factors live in registers, the right factor of every product is streamed by rows (LDS or L2), and a
    // sched_barrier closes every product: what is live at any point is what the next products still need.
    T8 p, d1, d2;
    if (TP) {
      if (more || full) {
        M3 lba;
        {
          M3 ub;
          ld_m(ub, lo_(B, q_sp), rs, V16);
          m3_zero(lba);
          if (IN_B) mac_stream<false>(lba, ub, lo_(0, q_pb), rs, V16);      // L_ba = U_b(s) U_t(s+b)
          else mac_stream<false>(lba, ub, go_(0, q_pb), rs, V16);
        }
        L2Q_PL_FENCE();
        if (full) {
          M3 lab;
          {
            M3 ua;
            ld_m(ua, lo_(0, q_sp), rs, V16);
            m3_zero(lab);
            mac_stream<false>(lab, ua, gn_(B, q_sp), rs, V16);              // L_ab = U_t(s) U_b(s+t)
          }
          L2Q_PL_FENCE();
          tah_prod<true>(p, lab, lba);                                      // TAH(P)
          L2Q_PL_FENCE();
          if (IN_B) {
            M3 t;
            {
              M3 uapb;
              ld_m(uapb, lo_(0, q_pb), rs, V16);
              m3_mul_na(t, uapb, lab);                                      // U_t(s+b) L_ab^H
            }
            L2Q_PL_FENCE();
            tah_rstream(d1, t, lo_(B, q_sp), rs, V16);                      // . U_b(s) -> link (s+b, t)
            L2Q_PL_FENCE();
          } else {
            // the contribution link (s, t) would receive from the plane based at s' = s - b (another tile):
            // TAH( U_t(s) (U_t(s') U_b(s'+t))^H U_b(s') )
            M3 t;
            {
              M3 lh;
              {
                M3 x;
                ld_m(x, go_(0, q_mb), rs, V16);
                m3_zero(lh);
                mac_stream<false>(lh, x, gn_(B, q_mb), rs, V16);
              }
              L2Q_PL_FENCE();
              M3 ua;
              ld_m(ua, lo_(0, q_sp), rs, V16);
              m3_mul_na(t, ua, lh);
            }
            L2Q_PL_FENCE();
            tah_rstream(d1, t, go_(B, q_mb), rs, V16);                      // (merged into the own-site deposit)
            L2Q_PL_FENCE();
          }
        }
        if (more) {
          M3 t;
          {
            M3 nxt;
            ld_m(nxt, gn_(B, q_sp), rs, V16);
            m3_mul_na(t, nxt, lba);                                         // U_b(s+t) L_ba^H
          }
          L2Q_PL_FENCE();
          tah_rstream(d2, t, lo_(0, q_sp), rs, V16);                        // . U_t(s) -> link (s+t, b), next iteration
          L2Q_PL_FENCE();
        }
      }
    } else {
      if (full) {
        M3 lab, lba;
        {
          M3 ua;
          ld_m(ua, lo_(A, q_sp), rs, V16);
          m3_zero(lab);
          if (IN_A) mac_stream<false>(lab, ua, lo_(B, q_pa), rs, V16);      // L_ab = U_a(s) U_b(s+a)
          else mac_stream<false>(lab, ua, go_(B, q_pa), rs, V16);
        }
        L2Q_PL_FENCE();
        {
          M3 ub;
          ld_m(ub, lo_(B, q_sp), rs, V16);
          m3_zero(lba);
          if (IN_B) mac_stream<false>(lba, ub, lo_(A, q_pb), rs, V16);      // L_ba = U_b(s) U_a(s+b)
          else mac_stream<false>(lba, ub, go_(A, q_pb), rs, V16);
        }
        L2Q_PL_FENCE();
        tah_prod<true>(p, lab, lba);
        L2Q_PL_FENCE();
        if (IN_B) {
          M3 t;
          {
            M3 uapb;
            ld_m(uapb, lo_(A, q_pb), rs, V16);
            m3_mul_na(t, uapb, lab);
          }
          L2Q_PL_FENCE();
          tah_rstream(d1, t, lo_(B, q_sp), rs, V16);                        // -> link (s+b, a)
          L2Q_PL_FENCE();
        } else {
          // from s' = s - b: TAH( U_a(s) (U_a(s') U_b(s'+a))^H U_b(s') )
          M3 t;
          {
            M3 lh;
            {
              M3 x;
              ld_m(x, go_(A, q_mb), rs, V16);
              m3_zero(lh);
              mac_stream<false>(lh, x, go_(B, q_mb_pa), rs, V16);
            }
            L2Q_PL_FENCE();
            M3 ua;
            ld_m(ua, lo_(A, q_sp), rs, V16);
            m3_mul_na(t, ua, lh);
          }
          L2Q_PL_FENCE();
          tah_rstream(d1, t, go_(B, q_mb), rs, V16);
          L2Q_PL_FENCE();
        }
        if (IN_A) {
          M3 t;
          {
            M3 ubpa;
            ld_m(ubpa, lo_(B, q_pa), rs, V16);
            m3_mul_na(t, ubpa, lba);
          }
          L2Q_PL_FENCE();
          tah_rstream(d2, t, lo_(A, q_sp), rs, V16);                        // -> link (s+a, b)
          L2Q_PL_FENCE();
        } else {
          // from s'' = s - a: TAH( U_b(s) (U_b(s'') U_a(s''+b))^H U_a(s'') )
          M3 t;
          {
            M3 lh;
            {
              M3 x;
              ld_m(x, go_(B, q_ma), rs, V16);
              m3_zero(lh);
              mac_stream<false>(lh, x, go_(A, q_ma_pb), rs, V16);
            }
            L2Q_PL_FENCE();
            M3 ub;
            ld_m(ub, lo_(B, q_sp), rs, V16);
            m3_mul_na(t, ub, lh);
          }
          L2Q_PL_FENCE();
          tah_rstream(d2, t, go_(A, q_ma), rs, V16);
          L2Q_PL_FENCE();
        }
      }
    }
    // what this thread deposits at its own site: link (s, a) gets +p (+ the recomputed neighbour term when b
    // leaves the tile), link (s, b) gets -p (+ likewise for a): the neighbour terms are folded into d1 / d2
    if (full) {
      if (!IN_B) t8_add(d1, p, 1.0);                           // d1 := p + H1
      if (!TP && !IN_A) t8_add(d2, p, -1.0);                   // d2 := -p + H2
    }
    __syncthreads();                                           // slice tcur consumed by every plane
    // ---- slice refresh (single buffer: nobody reads links between here and the next iteration)
    if (more) {
      if (TP) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
          *reinterpret_cast<double2*>(fr_lds + (3 * W + k) * kEnt + ltb) = pt[k];
      } else {
#pragma unroll
        for (int e = 0; e < 9; ++e)
          *reinterpret_cast<double2*>(fr_lds + (W - 2) * kPlaneB + e * kEnt + ltb) = pt[e];
      }
    }
    // ---- three deposit rounds: one deposit per link and bank in each
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      if (full) {
        if (r == R_AB) {
          if (IN_B) {
            bank_add(0, A, ltb, p, 1.0);
            bank_add(1, A, lb + q_pb, d1, 1.0);
          } else {
            bank_add(0, A, ltb, d1, 1.0);                      // p + the recomputed neighbour term
          }
        }
        if (r == R_BA) {
          if (TP) {
            bank_add(0, B, ltb, p, -1.0);
            // formed one slice earlier at this site: from the carry slot into the neighbour bank
            const int cs = kPlOffCarry + W * (4 * kEnt) + ltb;
            const int a1 = kPlOffBank + kPlBankB + B * (4 * kEnt) + ltb;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const double2 cv = lds_ld(cs + k * kEnt);
              double2 o = lds_ld(a1 + k * kEnt);
              o.x += cv.x; o.y += cv.y;
              *reinterpret_cast<double2*>(fr_lds + a1 + k * kEnt) = o;
            }
          } else if (IN_A) {
            bank_add(0, B, ltb, p, -1.0);
            bank_add(1, B, lb + q_pa, d2, 1.0);
          } else {
            bank_add(0, B, ltb, d2, 1.0);                      // -p + the recomputed neighbour term
          }
        }
      }
      if (TP && r == 0 && more) {
        // the new contribution to link (s + t, b) takes the slot's place (read just above by this very thread)
        const int cs = kPlOffCarry + W * (4 * kEnt) + ltb;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          *reinterpret_cast<double2*>(fr_lds + cs + k * kEnt) = make_double2(d2.v[2 * k], d2.v[2 * k + 1]);
      }
      if (!(L2Q_PL_EXP & 1) || r == 2) __syncthreads();
    }
    // ---- gather: the wavefront of direction W sums the banks, clears them and stores F
    if (W < 4 && full) {
      const int a0 = kPlOffBank + W * (4 * kEnt) + ltb;
      double f[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double2 u = lds_ld(a0 + k * kEnt);
        const double2 w = lds_ld(a0 + kPlBankB + k * kEnt);
        f[2 * k] = c.coef * (u.x + w.x);
        f[2 * k + 1] = c.coef * (u.y + w.y);
        *reinterpret_cast<double2*>(fr_lds + a0 + k * kEnt) = make_double2(0.0, 0.0);
        *reinterpret_cast<double2*>(fr_lds + a0 + kPlBankB + k * kEnt) = make_double2(0.0, 0.0);
      }
      const int so = W * 9 * V16 + gcur;
      const double d22 = -(f[6] + f[7]);
      buf_st_nt(ro, q_sp, so + 0 * V16, make_double2(0.0, f[6]));
      buf_st_nt(ro, q_sp, so + 1 * V16, make_double2(f[0], f[1]));
      buf_st_nt(ro, q_sp, so + 2 * V16, make_double2(f[2], f[3]));
      buf_st_nt(ro, q_sp, so + 3 * V16, make_double2(-f[0], f[1]));
      buf_st_nt(ro, q_sp, so + 4 * V16, make_double2(0.0, f[7]));
      buf_st_nt(ro, q_sp, so + 5 * V16, make_double2(f[4], f[5]));
      buf_st_nt(ro, q_sp, so + 6 * V16, make_double2(-f[2], f[3]));
      buf_st_nt(ro, q_sp, so + 7 * V16, make_double2(-f[4], f[5]));
      buf_st_nt(ro, q_sp, so + 8 * V16, make_double2(0.0, d22));
    }
  }
}

template <int INM>
__global__ __launch_bounds__(kPlThreads, L2Q_PL_OCC) void su3_force_plane_kernel(
    const double2* __restrict__ xn, Dims d, int nsb, int tsplit, int swz, double coef, double2* out, int lo) {
  const long w = xcd_swizzle(blockIdx.x, gridDim.x, swz);
  const int per_chain = nsb * tsplit;
  const long c = w / per_chain;
  const int rr = (int)(w % per_chain);
  const int tc = rr / nsb, sb = rr % nsb;
  const int V = d.V, T = d.T;
  PlCtx k;
  k.d = d;
  k.V16 = V * 16;
  k.Vs16 = d.X * d.Y * d.Z * 16;
  k.tile0b = sb * kRS * 16;
  k.lt = threadIdx.x & (kRS - 1);
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kRS);
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
  k.lo = lo;
  // identical barrier sequence in all six wavefronts
  switch (wv) {
    case 0: force_plane_sweep<0, 1, 0, INM>(k); break;
    case 1: force_plane_sweep<0, 2, 1, INM>(k); break;
    case 2: force_plane_sweep<0, 3, 2, INM>(k); break;
    case 3: force_plane_sweep<1, 2, 3, INM>(k); break;
    case 4: force_plane_sweep<1, 3, 4, INM>(k); break;
    default: force_plane_sweep<2, 3, 5, INM>(k); break;
  }
}

template <int INM>
static void launch_plane_variant(const double2* xn, Dims d, int nb, int nsb, int tsplit, double coef,
                                 double2* out, hipStream_t st) {
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute((const void*)su3_force_plane_kernel<INM>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, kPlLds);
  }
  hipLaunchKernelGGL((su3_force_plane_kernel<INM>), dim3((unsigned)((long)nb * nsb * tsplit)),
                     dim3(kPlThreads), kPlLds, st, xn, d, nsb, tsplit, tuning().xcd_swizzle, coef, out, 1);
}

int force_link_inmask(const Dims& d);
bool force_link_applicable(const Dims& d);

bool force_plane_applicable(const Dims& d) { return force_link_applicable(d) && d.T >= 2; }

void launch_force_plane(const double2* xn, Dims d, int nb, double coef, double2* out, hipStream_t st) {
  const int Vs = d.X * d.Y * d.Z;
  const int nsb = Vs / kRS;
  int tsplit = (int)cdiv(1024, (long)nb * nsb);
  if (tuning().force_tsplit > 0) tsplit = tuning().force_tsplit;
  if (tsplit > d.T) tsplit = d.T;
  if (tsplit < 1) tsplit = 1;
  const int tlen = (int)cdiv(d.T, tsplit);
  tsplit = (int)cdiv(d.T, tlen);
  switch (force_link_inmask(d)) {
    case 7: launch_plane_variant<7>(xn, d, nb, nsb, tsplit, coef, out, st); break;
    case 6: launch_plane_variant<6>(xn, d, nb, nsb, tsplit, coef, out, st); break;
    case 4: launch_plane_variant<4>(xn, d, nb, nsb, tsplit, coef, out, st); break;
    default: launch_plane_variant<0>(xn, d, nb, nsb, tsplit, coef, out, st);
  }
}

}  // namespace l2q
