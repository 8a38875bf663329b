// su3_force_plaq.hip -- SU(3) staple force with the plaquettes SHARED between the four links they close:
// slice-resident sweep, ONE eight-wavefront workgroup per CU, a wavefront per plane + two helpers (gfx950).
//
//   F_mu(s) = coef * TAH( U_mu(s) * A_mu(s) ),   A = the six staples of the link
//   (the reference: autograd of the Wilson action + projectTAH, lattice/su3/pytorch/lattice.py:299-308)
//
// The thread-per-link kernel (su3_force_link.hip) forms U A from 13 products per link = 52 per site on 76 operand
// matrices per site; its fp64 FMAs take 56 % of the SIMD cycles and the LDS pipe is as loaded as the VALU.  Here
// the work is organised by plaquette.  With
//     L_ab = U_a(s) U_b(s+a),   L_ba = U_b(s) U_a(s+b),   P = L_ab L_ba^H          (plane {a, b} based at s)
// all four links of the plaquette get their contribution from one thread (TAH is linear: a contribution is
// reduced to its 8 real components before it leaves the thread):
//     link (s,   a): + TAH(P)                         link (s,   b): - TAH(P)
//     link (s+b, a): TAH( U_a(s+b) L_ab^H U_b(s) )    link (s+a, b): TAH( U_b(s+a) L_ba^H U_a(s) )
// = 7 products on 4 operand matrices per plane and site.  The tile is a (y, z) plane of 64 sites swept along t:
// contributions to the next slice wait one iteration; those that would cross the tile in x are not sent -- the
// receiving workgroup forms them itself from the neighbouring plane (3 products each): 45 products per site
// instead of 52, 10 instead of 17 neighbour matrices per site from L2.
//
// Every one of the 6 contributions of a link has its OWN 64-byte LDS slot and exactly one writer per slice, so
// the sum is order-fixed (bit-reproducible) without atomics and without ordering barriers: compute -> barrier ->
// all eight wavefronts gather (6 slots -> F, 9 stores per link) and the helpers write the next slice's links ->
// barrier.  Wavefronts (SIMD = index % 4, so that each SIMD carries ~12 products per slice):
//     0 (t,y)  1 (t,z)  2 (y,z)  3 helper A: the x-neighbour term of link t + the spatial links of the refresh
//     4 (t,x)  5 (x,y)  6 (x,z)  7 helper B: the x-neighbour terms of links y and z
// LDS: links of ONE slice 36 KiB + 4 x 6 slots x 4 KiB = 132 KiB; 256 registers per thread (hipcc needs 172-202
// for a sweep: the two-workgroup variant of this design spilled, tools/experiments/README.md).
//
// Results agree with the thread-per-link kernels to rounding (sum of TAHs instead of TAH of the sum).
//
// MEASURED (MI355X, 8^4 x 256 chains; tools/force_bench.py, profiles/r06_force_plaq_ab.txt): 0.372 ms stand-alone
// (0.405 of the 8 TB/s roofline) against 0.397 ms for su3_force_link_kernel on the same box; inside the trajectory
// 0.389 against 0.385 ms with the TRAJECTORY 0.5 % shorter (20.79 vs 20.90 ms: less power and HBM traffic left for
// the neighbours).  That is inside the box-to-box spread, so it is the opt-in tuning force_tile = 7, not the default.  What it does achieve is the traffic: 1.42 GB
// of HBM traffic per launch = 1.17 x algorithmic (link kernel 2.02 GB = 1.68 x), L2 hit rate 0.62 (0.46).  In-kernel
// cycle counters (-DL2Q_PQ_PROF, tools/force_plaq_prof.py): the product phase of a SIMD is 11-13 products of 108
// four-cycle fp64 FMA instructions = 5600-7100 cycles of VALU issue, the gather ~1500, barrier slack ~600; with the
// stalls of the first versions gone (chain operands a whole iteration ahead, one gather group per wavefront, LDS-DMA
// refresh) the kernel draws enough for the socket's 1400 W cap to set the clock, which is what bounds the
// thread-per-link kernel as well (profiles/r06_force_clock_power.txt).
#include "su3_force_tile.hpp"

namespace l2q {

#ifdef L2Q_PQ_NOFENCE
#define L2Q_PQ_FENCE() do { } while (0)
#else
#define L2Q_PQ_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#endif

// Timing experiments (tools/ab_build.sh ... -DL2Q_PQ_EXP=bits; results WRONG, only the clock is read):
//   1 no output stores (the gather's slot reads become dead code too)   2 helpers form no neighbour terms   4 the planes' chain operands are requested once, not
//   per slice   8 no slice refresh loads   16 write-back instead of streaming (nt) stores
#ifndef L2Q_PQ_EXP
#define L2Q_PQ_EXP 0
#endif
// Workgroup barrier that waits for this wavefront's LDS traffic only.  __syncthreads() also drains vmcnt: every global
// load and STORE in flight (the gather's stores, the operands requested for the next slice) would have to be
// acknowledged at each of the two barriers of a slice.  Nothing is communicated through global memory inside the
// kernel, so LDS ordering is all the barrier has to give.
#ifdef L2Q_PQ_SYNCTHREADS
__device__ __forceinline__ void pqf_barrier() { __syncthreads(); }
#else
__device__ __forceinline__ void pqf_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif
// -DL2Q_PQ_PROF: per-wavefront cycle totals (products, wait at the first barrier, gather / refresh, wait at the second
// barrier) of workgroup 0, written over the first entries of its output (a timing build: F is wrong there)
#ifndef L2Q_PQ_PROF
#define L2Q_PQ_PROF 0
#endif
struct PqfProf {
  long long t[4] = {0, 0, 0, 0};
  long long last = 0;
  __device__ __forceinline__ void start() { if (L2Q_PQ_PROF) last = clock64(); }
  __device__ __forceinline__ void mark(int k) {
    if (L2Q_PQ_PROF) { const long long now = clock64(); t[k] += now - last; last = now; }
  }
  __device__ __forceinline__ void dump(double2* out, int wv, int lane) const {
    if (L2Q_PQ_PROF && blockIdx.x == 0 && lane == 0) {
      out[2 * wv] = make_double2((double)t[0], (double)t[1]);
      out[2 * wv + 1] = make_double2((double)t[2], (double)t[3]);
    }
  }
};
constexpr int kPqfThreads = kRS * 8;
// LDS: t-links of the current slice [9][site], then TWO buffers of spatial links [3 dirs][9][site] (the next slice's
// arrive by LDS-DMA in the buffer the current one does not use), then the slots
constexpr int kPqfOffSlot = 7 * kPlaneB;
constexpr int kPqfSlotB = 4 * kEnt;                      // one slot: [4 component pairs][site] double2 = 4 KiB
constexpr int kPqfLds = kPqfOffSlot + 4 * 6 * kPqfSlotB; // 63 + 96 KiB = 162 816 B of the CU's 163 840
// byte offset of link direction rho in the buffer pair (cur: which spatial buffer holds the current slice)
__device__ __forceinline__ int pqf_link_base(int rho, int cur) {
  return rho == 0 ? 0 : (1 + 3 * cur + (rho - 1)) * kPlaneB;
}

struct T8 {
  double v[8];       // TWICE: (re, im) of entries (0,1), (0,2), (1,2); Im of entries (0,0), (1,1) (traceless)
};

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
  // TWICE the projection's components: the 1/2 of (W - W^H)/2 is folded into the gather's coefficient; the trace
  // term by a multiplication (an fp64 division is ~14 instructions, and there is one per contribution)
  const double tri2 = (wi[0] + wi[4] + wi[8]) * (2.0 / 3.0);
  r.v[0] = wr[1] - wr[3]; r.v[1] = wi[1] + wi[3];
  r.v[2] = wr[2] - wr[6]; r.v[3] = wi[2] + wi[6];
  r.v[4] = wr[5] - wr[7]; r.v[5] = wi[5] + wi[7];
  r.v[6] = fma(2.0, wi[0], -tri2); r.v[7] = fma(2.0, wi[4], -tri2);
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
  // TWICE the projection's components: the 1/2 of (W - W^H)/2 is folded into the gather's coefficient; the trace
  // term by a multiplication (an fp64 division is ~14 instructions, and there is one per contribution)
  const double tri2 = (wi[0] + wi[4] + wi[8]) * (2.0 / 3.0);
  r.v[0] = wr[1] - wr[3]; r.v[1] = wi[1] + wi[3];
  r.v[2] = wr[2] - wr[6]; r.v[3] = wi[2] + wi[6];
  r.v[4] = wr[5] - wr[7]; r.v[5] = wi[5] + wi[7];
  r.v[6] = fma(2.0, wi[0], -tri2); r.v[7] = fma(2.0, wi[4], -tri2);
}


struct PqfCtx {
  __amdgpu_buffer_rsrc_t rs, ro;
  Dims d;
  int V16, Vs16, tile0b, lt, t0, t1;
  int sp, px, py, pz;
  double coef;
  int wv;
  void* outp;      // (the chain's output: only the L2Q_PQ_PROF dump uses the raw pointer)
  const char* chain;   // the chain's links (LDS-DMA takes a flat address)
};

// slot j of link (site byte offset ltb, direction dir): j = r (own-site term of plane {dir, r-th other direction})
// or 3 + r (the term arriving from the neighbour in that direction / the previous slice)
__device__ __forceinline__ void slot_put(int dir, int j, int ltb, const T8& c, double sgn) {
  const int a = kPqfOffSlot + (dir * 6 + j) * kPqfSlotB + ltb;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    *reinterpret_cast<double2*>(fr_lds + a + k * kEnt) =
        sgn < 0.0 ? make_double2(-c.v[2 * k], -c.v[2 * k + 1]) : make_double2(c.v[2 * k], c.v[2 * k + 1]);
}

__host__ __device__ constexpr int pqf_rank(int c, int b) { return b < c ? b : b - 1; }   // b among the others of c

// Gather of one group: 64 half-links (direction dir; half 0: entries (0,1), (0,2) and their mirror images, half 1:
// entry (1,2), its mirror image and the diagonal): the six slots summed in a fixed order and scaled (pqf_sum), then
// stored (pqf_store).  (Measured: gather + stores cost 20 % of the kernel's time wherever the stores are issued -- right
// here, after the next slice's operand requests, or deferred into the next slice's products; the slot traffic --
// 72 ds_write_b128 + 96 ds_read_b128 per site next to ~200 operand reads -- is the likelier half of that.)
struct PqfOut { double f0, f1, f2, f3; };
__device__ __forceinline__ PqfOut pqf_sum(const PqfCtx& c, int dir, int half, int ltb) {
  const int a0 = kPqfOffSlot + dir * 6 * kPqfSlotB + (2 * half) * kEnt + ltb;
  double f0 = 0.0, f1 = 0.0, f2 = 0.0, f3 = 0.0;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const double2 u = lds_ld(a0 + j * kPqfSlotB);
    const double2 w = lds_ld(a0 + j * kPqfSlotB + kEnt);
    f0 += u.x; f1 += u.y; f2 += w.x; f3 += w.y;
  }
  const double hc = 0.5 * c.coef;                           // (the contributions carry twice the TAH components)
  return PqfOut{f0 * hc, f1 * hc, f2 * hc, f3 * hc};
}
#if L2Q_PQ_EXP & 16
#define PQF_ST buf_st
#else
#define PQF_ST buf_st_nt
#endif
__device__ __forceinline__ void pqf_store(const PqfCtx& c, int dir, int half, int q_sp, int gslice, const PqfOut& o) {
  if (L2Q_PQ_EXP & 1) return;
  const int so = dir * 9 * c.V16 + gslice;
  const __amdgpu_buffer_rsrc_t ro = c.ro;
  if (half == 0) {
    PQF_ST(ro, q_sp, so + 1 * c.V16, make_double2(o.f0, o.f1));
    PQF_ST(ro, q_sp, so + 2 * c.V16, make_double2(o.f2, o.f3));
    PQF_ST(ro, q_sp, so + 3 * c.V16, make_double2(-o.f0, o.f1));
    PQF_ST(ro, q_sp, so + 6 * c.V16, make_double2(-o.f2, o.f3));
  } else {
    PQF_ST(ro, q_sp, so + 5 * c.V16, make_double2(o.f0, o.f1));
    PQF_ST(ro, q_sp, so + 7 * c.V16, make_double2(-o.f0, o.f1));
    PQF_ST(ro, q_sp, so + 0 * c.V16, make_double2(0.0, o.f2));
    PQF_ST(ro, q_sp, so + 4 * c.V16, make_double2(0.0, o.f3));
    PQF_ST(ro, q_sp, so + 8 * c.V16, make_double2(0.0, -(o.f2 + o.f3)));
  }
}

// Plane wavefront {A, B}, A < B.  Direction 1 (x) leaves the tile, 2 and 3 stay inside, 0 (t) is the sweep.
// G0: the gather group (dir = g & 3, half = g >> 2) this wavefront sums and stores.  A wavefront's
// operands that come from L2 / HBM are requested one iteration AHEAD (before its gather stores).
template <int A, int B, int G0, int RE0 = 0, int RNE = 0>
__device__ __forceinline__ void pqf_plane(const PqfCtx& c) {
  constexpr bool TP = A == 0;
  constexpr bool IN_A = A != 1;                                // (t counts as inside: carried)
  constexpr bool IN_B = B != 1;
  constexpr bool PRE_A = TP || !IN_A;                          // U_b(s+a) comes from the chain
  constexpr bool PRE_B = !IN_B;                                // U_a(s+b) comes from the chain: plane (t, x)
  constexpr bool EARLY = !IN_A || !IN_B;                       // an operand crosses the tile in x
  constexpr int R_AB = pqf_rank(A, B), R_BA = pqf_rank(B, A);
  const Dims& d = c.d;
  const int T = d.T, V16 = c.V16, Vs16 = c.Vs16;
  const __amdgpu_buffer_rsrc_t rs = c.rs;
  const int q_sp = c.sp * 16, lb = -c.tile0b, ltb = c.lt * 16;
  const int q_pb = hop(c.sp, c.px, c.py, c.pz, B, +1, d) * 16;
  const int q_pa = TP ? q_sp : hop(c.sp, c.px, c.py, c.pz, TP ? 1 : A, +1, d) * 16;
  T8 carry;
#pragma unroll
  for (int k = 0; k < 8; ++k) carry.v[k] = 0.0;
  // slice of the operand U_b(s+a) of iteration `it`: the NEXT slice for a temporal plane, the current one else
  M3 ubpa_pre, uapb_pre;
  {
    const int tc0 = (c.t0 - 1 + T) % T;
    const int ts = TP ? (tc0 + 1) % T : tc0;
    if (PRE_A) ld_m(ubpa_pre, Opnd<false>{0, q_pa, B * 9 * V16 + ts * Vs16}, rs, V16);
    if (PRE_B) ld_m(uapb_pre, Opnd<false>{0, q_pb, A * 9 * V16 + tc0 * Vs16}, rs, V16);
  }
  PqfProf prof;
  prof.start();
  const int niter = (c.t1 - c.t0) + 1;
#pragma unroll 1
  for (int it = 0; it < niter; ++it) {
    const int tcur = (c.t0 - 1 + it + T) % T;
    const int tnext = (tcur + 1 == T) ? 0 : tcur + 1;
    const int tnext2 = (tnext + 1 == T) ? 0 : tnext + 1;
    const int gcur = tcur * Vs16;
    const bool more = it + 1 < niter;
    const bool full = it >= 1;
    const int cur = it & 1;
    auto lo_ = [&](int rho, int qb) { return Opnd<true>{pqf_link_base(rho, cur) + lb + qb, qb, rho * 9 * V16 + gcur}; };
    auto prefetch_next = [&]() {
      if (L2Q_PQ_EXP & 4) return;
      const int ts = TP ? tnext2 : tnext;      // (the last iteration re-requests a valid slice: harmless)
      if (PRE_A) ld_m(ubpa_pre, Opnd<false>{0, q_pa, B * 9 * V16 + ts * Vs16}, rs, V16);
      if (PRE_B) ld_m(uapb_pre, Opnd<false>{0, q_pb, A * 9 * V16 + tnext * Vs16}, rs, V16);
    };
    // this wavefront's share of the slice refresh (RNE entries from entry RE0 of the next slice's 36)
    double2 pt[RNE > 0 ? RNE : 1];
    if (RNE > 0 && more) {
#pragma unroll
      for (int k = 0; k < RNE; ++k) pt[k] = buf_ld(rs, q_sp, (RE0 + k) * V16 + tnext * Vs16);
    }
    // the term formed one slice earlier for link (s, b) of THIS slice: its slot is free again (the previous
    // gather is behind the last barrier)
    if (TP && full) slot_put(B, 3 + R_BA, ltb, carry, 1.0);
    if (TP ? (more || full) : full) {
      // all four operands of the plaquette are requested up front (each is read ONCE: 4 instead of 6 matrix reads) and
      // every product below runs on registers: no LDS latency inside the chain of products
      M3 ub, ua;
      M3& uapb = uapb_pre;                                                  // (chain operand, or filled from LDS here)
      M3& ubpa = ubpa_pre;
      ld_m(ub, lo_(B, q_sp), rs, V16);
      if (!PRE_B) ld_m(uapb, lo_(A, q_pb), rs, V16);
      ld_m(ua, lo_(A, q_sp), rs, V16);
      if (!PRE_A) ld_m(ubpa, lo_(B, q_pa), rs, V16);
      M3 lba, t, lab;
      m3_mul_nn(lba, ub, uapb);                                             // L_ba = U_b(s) U_a(s+b)
      L2Q_PQ_FENCE();
      m3_mul_na(t, ubpa, lba);                                              // U_b(s+a) L_ba^H
      L2Q_PQ_FENCE();
      // Wavefronts whose operands cross the tile in x ((t,x), (x,y), (x,z)): the chain operands have their last use
      // here -- request the NEXT slice's now, a whole iteration ahead (they take ~6000 cycles to arrive under this
      // kernel's own traffic: in-kernel counters, tools/force_plaq_prof.py).  The temporal planes (t,y), (t,z) read
      // U_b(s + t), which the refresh requests have already pulled into L2: their request goes out after the first
      // barrier (EARLY = false; requesting it here costs them 34 spilled registers).
      if (EARLY) {
        if (full && IN_B) m3_mul_nn(lab, ua, ubpa);                         // L_ab = U_a(s) U_b(s+a)
        L2Q_PQ_FENCE();
        prefetch_next();
        L2Q_PQ_FENCE();
      }
      if (full) {
        T8 p;
        tah_prod<false>(p, ua, t);                                          // TAH(P), P = U_a(s) U_b(s+a) L_ba^H
        slot_put(A, R_AB, ltb, p, 1.0);
        slot_put(B, R_BA, ltb, p, -1.0);
      }
      L2Q_PQ_FENCE();
      if (TP ? more : IN_A) {
        T8 d2;
        tah_prod<false>(d2, t, ua);                                         // -> link (s+a, b)
        if (TP) carry = d2;
        else slot_put(B, 3 + R_BA, lb + q_pa, d2, 1.0);
      }
      L2Q_PQ_FENCE();
      if (full && IN_B) {
        // -> link (s+b, a): TAH( U_a(s+b) L_ab^H U_b(s) )
        if (!EARLY) m3_mul_nn(lab, ua, ubpa);                               // L_ab = U_a(s) U_b(s+a)
        L2Q_PQ_FENCE();
        M3 t1;
        m3_mul_na(t1, uapb, lab);
        L2Q_PQ_FENCE();
        T8 d1;
        tah_prod<false>(d1, t1, ub);
        slot_put(A, 3 + R_AB, lb + q_pb, d1, 1.0);
      }
      L2Q_PQ_FENCE();
    } else if (EARLY) {
      prefetch_next();                                                      // (prologue of a spatial plane)
    }
    prof.mark(0);
    pqf_barrier();                                           // every contribution of this slice is in its slot
    prof.mark(1);
    if (!EARLY) prefetch_next();
    if (RNE > 0 && more) {
#pragma unroll
      for (int k = 0; k < RNE; ++k)
        *reinterpret_cast<double2*>(fr_lds + (RE0 + k) * kEnt + ltb) = pt[k];       // (t-links: entries 0..8)
    }
    if (full) pqf_store(c, G0 & 3, G0 >> 2, q_sp, gcur, pqf_sum(c, G0 & 3, G0 >> 2, ltb));
    prof.mark(2);
    pqf_barrier();                                           // slots free, next slice's links in place
    prof.mark(3);
  }
  __syncthreads();                                           // (every store of the workgroup is out)
  prof.dump(reinterpret_cast<double2*>(c.outp), c.wv, c.lt);
}

// Helper wavefront: the x-neighbour terms of links c in {C0, C1} (C1 < 0: one term), and NE entries starting at
// entry E0 of the 36 of the next slice's links.
//   link (s, c) receives  TAH( U_c(s) (U_c(s') U_x(s'+c))^H U_x(s') ),  s' = s - x   (slot 3 + rank_c(x))
template <int C0, int C1, int E0, int NE, int G0>
__device__ __forceinline__ void pqf_helper(const PqfCtx& c) {
  const Dims& d = c.d;
  const int T = d.T, V16 = c.V16, Vs16 = c.Vs16;
  const __amdgpu_buffer_rsrc_t rs = c.rs;
  const int q_sp = c.sp * 16, lb = -c.tile0b, ltb = c.lt * 16;
  const int q_mx = hop(c.sp, c.px, c.py, c.pz, 1, -1, d) * 16;             // s' = s - x
  int q_mx_pc[2] = {q_mx, q_mx};                                          // s' + c (c spatial)
  {
    int q = q_mx / 16;
    const int z = q % d.Z; q /= d.Z;
    const int y = q % d.Y; q /= d.Y;
    if (C0 > 0) q_mx_pc[0] = hop(q_mx / 16, q, y, z, C0, +1, d) * 16;
    if (C1 > 0) q_mx_pc[1] = hop(q_mx / 16, q, y, z, C1 > 0 ? C1 : 2, +1, d) * 16;
  }
  PqfProf prof;
  prof.start();
  M3 ux, x0, y0, x1, y1;                 // chain operands of the coming slice (requested at the end of the previous one)
  m3_zero(ux); m3_zero(x0); m3_zero(y0); m3_zero(x1); m3_zero(y1);
  const int niter = (c.t1 - c.t0) + 1;
#pragma unroll 1
  for (int it = 0; it < niter; ++it) {
    const int tcur = (c.t0 - 1 + it + T) % T;
    const int tnext = (tcur + 1 == T) ? 0 : tcur + 1;
    const int gcur = tcur * Vs16, gnxt = tnext * Vs16;
    const bool more = it + 1 < niter;
    const bool full = it >= 1;
    const int cur = it & 1;
    auto lo_ = [&](int rho, int qb) { return Opnd<true>{pqf_link_base(rho, cur) + lb + qb, qb, rho * 9 * V16 + gcur}; };
    // slice refresh: the next slice's spatial links (entries 9..35) go straight from the chain into the spatial
    // buffer the current slice does not use -- LDS-DMA, no registers, the whole iteration to arrive
    if (NE > 0 && more && !(L2Q_PQ_EXP & 8)) {
      const char* src = c.chain + (long)gnxt + q_sp;
      char* dst = fr_lds + (1 + 3 * (cur ^ 1)) * kPlaneB;
#pragma unroll
      for (int k = 0; k < NE; ++k)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long)(E0 + k) * V16),
                                         (__attribute__((address_space(3))) void*)(dst + (E0 - 9 + k) * kEnt), 16, 0, 0);
    }
    if (full && !(L2Q_PQ_EXP & 2)) {
      // the chain operands of both terms were requested a whole iteration ago (below)
      {
        M3 lh, t, uc;
        ld_m(uc, lo_(C0, q_sp), rs, V16);
        m3_mul_nn(lh, x0, y0);
        L2Q_PQ_FENCE();
        m3_mul_na(t, uc, lh);
        L2Q_PQ_FENCE();
        T8 h;
        tah_prod<false>(h, t, ux);
        slot_put(C0, 3 + pqf_rank(C0, 1), ltb, h, 1.0);
        L2Q_PQ_FENCE();
      }
      if (C1 >= 0) {
        constexpr int CC = C1 >= 0 ? C1 : 0;
        M3 lh, t, uc;
        ld_m(uc, lo_(CC, q_sp), rs, V16);
        m3_mul_nn(lh, x1, y1);
        L2Q_PQ_FENCE();
        m3_mul_na(t, uc, lh);
        L2Q_PQ_FENCE();
        T8 h;
        tah_prod<false>(h, t, ux);
        slot_put(CC, 3 + pqf_rank(CC, 1), ltb, h, 1.0);
        L2Q_PQ_FENCE();
      }
    }
    // the NEXT slice's chain operands, a whole iteration ahead (s' = s - x; the last iteration re-requests a valid slice)
    if (!(L2Q_PQ_EXP & 2)) {
      const int gn2 = ((tnext + 1 == T) ? 0 : tnext + 1) * Vs16;
      ld_m(ux, Opnd<false>{0, q_mx, 1 * 9 * V16 + gnxt}, rs, V16);                       // U_x(s')
      ld_m(x0, Opnd<false>{0, q_mx, C0 * 9 * V16 + gnxt}, rs, V16);                      // U_c(s')
      if (C0 == 0) ld_m(y0, Opnd<false>{0, q_mx, 1 * 9 * V16 + gn2}, rs, V16);           // U_x(s' + t)
      else ld_m(y0, Opnd<false>{0, q_mx_pc[0], 1 * 9 * V16 + gnxt}, rs, V16);            // U_x(s' + c)
      if (C1 >= 0) {
        ld_m(x1, Opnd<false>{0, q_mx, (C1 >= 0 ? C1 : 0) * 9 * V16 + gnxt}, rs, V16);
        ld_m(y1, Opnd<false>{0, q_mx_pc[1], 1 * 9 * V16 + gnxt}, rs, V16);
      }
    }
    prof.mark(0);
    pqf_barrier();
    prof.mark(1);
    if (full) pqf_store(c, G0 & 3, G0 >> 2, q_sp, gcur, pqf_sum(c, G0 & 3, G0 >> 2, ltb));
    {
      // slice refresh (single buffer: every plane has passed the barrier above, nobody reads links before the next)
      // The LDS-DMA of the refresh must have landed before the barrier.  vmcnt retires in order: AFTER the DMA this
      // wavefront has issued the next slice's chain operands (9 requests per matrix, behind scheduling fences, so
      // they cannot move above the DMA) and, in a full iteration, 4-5 gather stores.  Waiting until at most the
      // operand requests are outstanding is exact without the stores and waits for the oldest few operand
      // requests (issued thousands of cycles ago) with them -- never for fewer than all of the DMA.
      if (NE > 0) {
        constexpr int kAfter = ((C1 >= 0) ? 5 : 3) * 9;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kAfter) : "memory");
      }
    }
    prof.mark(2);
    pqf_barrier();
    prof.mark(3);
  }
  __syncthreads();
  prof.dump(reinterpret_cast<double2*>(c.outp), c.wv, c.lt);
}

__global__ __launch_bounds__(kPqfThreads, 2) void su3_force_plaq_kernel(
    const double2* __restrict__ xn, Dims d, int nsb, int tsplit, int swz, double coef, double2* out) {
  const long w = xcd_swizzle(blockIdx.x, gridDim.x, swz);
  const int per_chain = nsb * tsplit;
  const long c = w / per_chain;
  const int rr = (int)(w % per_chain);
  const int tc = rr / nsb, sb = rr % nsb;
  const int V = d.V, T = d.T;
  PqfCtx k;
  k.d = d;
  k.V16 = V * 16;
  k.Vs16 = d.X * d.Y * d.Z * 16;
  k.tile0b = sb * kRS * 16;
  k.lt = threadIdx.x & (kRS - 1);
  k.wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kRS);
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
  k.outp = (void*)(out + c * 36L * V);
  k.chain = reinterpret_cast<const char*>(xn + c * 36L * V);
  // first slice (t0 - 1) into LDS: 36 entries per site over the 512 threads; wavefront w takes 4 or 5 of them
  {
    const int ta = (k.t0 - 1 + T) % T;
    const int q_sp = k.sp * 16;
    for (int e = k.wv; e < 36; e += 8) {                         // (iteration 0 reads spatial buffer 0)
      const double2 va = buf_ld(k.rs, q_sp, e * k.V16 + ta * k.Vs16);
      *reinterpret_cast<double2*>(fr_lds + e * kEnt + k.lt * 16) = va;
    }
  }
  __syncthreads();
  // identical barrier sequence in all eight wavefronts
  switch (k.wv) {
    // (gather group g = wavefront index: direction g & 3, half g >> 2)
    case 0: pqf_plane<0, 2, 0>(k); break;
    case 1: pqf_plane<0, 3, 1>(k); break;
    case 2: pqf_plane<2, 3, 2>(k); break;
    case 3: pqf_helper<0, -1, 9, 27, 3>(k); break;             // x-term of link t; the spatial links of the refresh (LDS-DMA)
    case 4: pqf_plane<0, 1, 4, 0, 9>(k); break;                // (+ the t-links of the refresh)
    case 5: pqf_plane<1, 2, 5>(k); break;
    case 6: pqf_plane<1, 3, 6>(k); break;
    default: pqf_helper<2, 3, 0, 0, 7>(k); break;              // x-terms of links y and z
  }
}

int force_link_inmask(const Dims& d);
bool force_link_applicable(const Dims& d);

// the (y, z) plane must be the 64-site tile (8^4-like lattices); everything else stays on su3_force_link.hip
bool force_plaq_applicable(const Dims& d) {
  return force_link_applicable(d) && d.Y * d.Z == kRS && d.T >= 2 && d.X >= 2;
}

void launch_force_plaq(const double2* xn, Dims d, int nb, double coef, double2* out, hipStream_t st) {
  const int Vs = d.X * d.Y * d.Z;
  const int nsb = Vs / kRS;
  int tsplit = (int)cdiv(512, (long)nb * nsb);                 // one workgroup per CU: >= 2 rounds of 256
  if (tuning().force_tsplit > 0) tsplit = tuning().force_tsplit;
  if (tsplit > d.T) tsplit = d.T;
  if (tsplit < 1) tsplit = 1;
  const int tlen = (int)cdiv(d.T, tsplit);
  tsplit = (int)cdiv(d.T, tlen);
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute((const void*)su3_force_plaq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              kPqfLds);
  }
  hipLaunchKernelGGL(su3_force_plaq_kernel, dim3((unsigned)((long)nb * nsb * tsplit)), dim3(kPqfThreads), kPqfLds,
                     st, xn, d, nsb, tsplit, tuning().xcd_swizzle, coef, out);
}

}  // namespace l2q
