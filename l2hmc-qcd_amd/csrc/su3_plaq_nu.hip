// su3_plaq_nu.hip -- SU(3) plaquette sums, slice-resident sweep with the six planes of a site split
// over wavefronts (gfx950).
//
//   sum over sites s and planes u > v of  tr[ U_u(s) U_v(s+u) (U_v(s) U_u(s+v))^H ]
//   (the reference: _wilson_loops / _plaquettes, lattice/su3/pytorch/lattice.py:157-269)
//
// The thread-per-site kernel (su3_plaq_slice_kernel) keeps the four links of the next slice in
// registers (144 VGPRs) next to three working matrices: one wavefront per SIMD, so every LDS and L2
// latency of its six planes is exposed (10 us per slice for 3.5 us of arithmetic; 46 % of the HBM
// roofline).  Here a wavefront owns ONE plane of 64 sites -- two 3x3 products and a trace on four
// operand matrices, ~120 VGPRs -- and a workgroup is 64 sites x 6 planes = 6 wavefronts.  LDS holds
// the links of the current and the next slice (2 x 36 KiB), every thread carries 6 of the 36
// link entries of its site for the slice after next in registers while it computes; two
// workgroups share a CU = 12 wavefronts, 3 per SIMD.  Every link is fetched from HBM once per sweep;
// the only operands that are not LDS reads are those at s + x (the tile is a (y, z) plane on the
// 8^4 / 16^4 lattices), which the neighbouring workgroup of the same XCD has just pulled into L2.
// Addressing as in su3_force_tile.hpp: uniform parts scalar, ONE 32-bit VGPR per neighbour site.
//
// MEASURED (MI355X, tools/plaq_bench.py): 0.207 ms at 8^4 x 256 chains vs 0.178 ms for the
// thread-per-site kernel (16^4 x 64: 0.87 vs 0.72 ms) -- three wavefronts per SIMD do not pay for
// 64-site tiles (every s + x operand from L2 instead of half of them) and two barriers per two
// products.  Kept as tuning plaq_sweep = 3; the default stays the thread-per-site kernel.
#include "su3_force_tile.hpp"

namespace l2q {

constexpr int kPqThreads = kRS * 6;
constexpr int kPqBuf = 4 * kPlaneB;                   // one slice: 4 directions x 9 entries x 64 sites
constexpr int kPqRed = 2 * kPqBuf;
constexpr int kPqLds = kPqRed + 128;

template <int INM>
__device__ __forceinline__ constexpr bool pq_in(int dir) { return ((INM >> (dir - 1)) & 1) != 0; }

struct PqCtx {
  __amdgpu_buffer_rsrc_t rs;
  Dims d;
  int V16, Vs16, tile0b, lt, t0, t1;
  int sp, px, py, pz;
  int lo;
};

// One wavefront's sweep: plane (U, V), U > V; V == 0 is a temporal plane.
template <int U, int V, int P, int INM>
__device__ __forceinline__ void plaq_nu_sweep(const PqCtx& c, double& sr, double& si) {
  constexpr bool IN_U = pq_in<INM>(U);
  const Dims& d = c.d;
  const int T = d.T, V16 = c.V16, Vs16 = c.Vs16;
  const __amdgpu_buffer_rsrc_t rs = c.rs;
  const int q_sp = c.sp * 16;
  const int q_pu = hop(c.sp, c.px, c.py, c.pz, U, +1, d) * 16;                  // s + u
  const int q_pv = V == 0 ? q_sp : hop(c.sp, c.px, c.py, c.pz, V == 0 ? 1 : V, +1, d) * 16;   // s + v
  const int lb = -c.tile0b;                           // LDS address of site q: buffer + rho * kPlaneB + lb + q * 16
  // prefetch duty: entries 6 P .. 6 P + 5 of the 36 (rho * 9 + e) of this thread's site
  const int own = 6 * P * kEnt + c.lt * 16;
  {
    const int ta = c.t0 % T, tb = (c.t0 + 1) % T;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double2 va = buf_ld(rs, q_sp, (6 * P + k) * V16 + ta * Vs16);
      const double2 vb = buf_ld(rs, q_sp, (6 * P + k) * V16 + tb * Vs16);
      *reinterpret_cast<double2*>(fr_lds + own + k * kEnt) = va;
      *reinterpret_cast<double2*>(fr_lds + kPqBuf + own + k * kEnt) = vb;
    }
  }
  __syncthreads();
  int cur = 0;
#pragma unroll 1
  for (int t = c.t0; t < c.t1; ++t) {
    const int tn = (t + 1 == T) ? 0 : t + 1;
    const int tp = (tn + 1 == T) ? 0 : tn + 1;
    const bool more = t + 1 < c.t1;
    const int bc = cur ? kPqBuf : 0, bn = cur ? 0 : kPqBuf;
    const int gcur = t * Vs16;
    double2 pre[6];
    if (more) {
#pragma unroll
      for (int k = 0; k < 6; ++k) pre[k] = buf_ld(rs, q_sp, (6 * P + k) * V16 + tp * Vs16);
    }
    auto oc = [&](int rho, int qb) { return Opnd<true>{bc + rho * kPlaneB + lb + qb, qb, rho * 9 * V16 + gcur}; };
    auto on = [&](int rho, int qb) { return Opnd<true>{bn + rho * kPlaneB + lb + qb, qb, 0}; };
    auto gc = [&](int rho, int qb) { return Opnd<false>{0, qb, rho * 9 * V16 + gcur}; };
    if (t >= c.lo) {   // always true; `lo` is a kernel argument, so the block stays conditional
      // Y = U_u(s) U_v(s+u), W = U_v(s) U_u(s+v): the left factor in registers, the right one
      // streamed by rows (the live set stays at three matrices + one row, as in su3_force_nu.hip)
      M3 a, y, w;
      m3_zero(y);
      ld_m(a, oc(U, q_sp), rs, V16);
      if (IN_U) mac_stream<false>(y, a, oc(V, q_pu), rs, V16);
      else mac_stream<false>(y, a, gc(V, q_pu), rs, V16);
      __builtin_amdgcn_sched_barrier(0);                // (hipcc otherwise interleaves both products: 350 spilled VGPRs)
      m3_zero(w);
      ld_m(a, oc(V, q_sp), rs, V16);
      if constexpr (V == 0) {
        mac_stream<false>(w, a, on(U, q_sp), rs, V16);          // s + t: the next slice
      } else {
        if (pq_in<INM>(V == 0 ? 1 : V)) mac_stream<false>(w, a, oc(U, q_pv), rs, V16);
        else mac_stream<false>(w, a, gc(U, q_pv), rs, V16);
      }
      __builtin_amdgcn_sched_barrier(0);
      // tr Y W^H
#pragma unroll
      for (int e = 0; e < 9; ++e) {
        sr = fma(y.re[e], w.re[e], sr); sr = fma(y.im[e], w.im[e], sr);
        si = fma(y.im[e], w.re[e], si); si = fma(-y.re[e], w.im[e], si);
      }
    }
    if (more) {
      __syncthreads();                                  // slice t consumed by every plane
#pragma unroll
      for (int k = 0; k < 6; ++k) *reinterpret_cast<double2*>(fr_lds + bc + own + k * kEnt) = pre[k];
      cur ^= 1;
      __syncthreads();                                  // slice t + 2 in place
    }
  }
}

template <int INM>
__global__ __launch_bounds__(kPqThreads, 3) void su3_plaq_nu_kernel(
    const double2* __restrict__ xn, Dims d, int nsb, int tsplit, int swz, double* __restrict__ partial,
    int lo) {
  const long w = xcd_swizzle(blockIdx.x, gridDim.x, swz);
  const int per_chain = nsb * tsplit;
  const long c = w / per_chain;
  const int rr = (int)(w % per_chain);
  const int tc = rr / nsb, sb = rr % nsb;
  const int V = d.V, T = d.T;
  PqCtx k;
  k.d = d;
  k.V16 = V * 16;
  k.Vs16 = d.X * d.Y * d.Z * 16;
  k.tile0b = sb * kRS * 16;
  k.lt = threadIdx.x & (kRS - 1);
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kRS);
  const int tlen = (T + tsplit - 1) / tsplit;
  k.t0 = tc * tlen;
  k.t1 = min(T, k.t0 + tlen);
  k.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(xn + c * 36L * V), 0, 36 * k.V16, 0x00020000);
  k.sp = sb * kRS + k.lt;
  {
    int q = k.sp;
    k.pz = q % d.Z; q /= d.Z;
    k.py = q % d.Y; q /= d.Y;
    k.px = q;
  }
  k.lo = lo;
  double sr = 0.0, si = 0.0;
  // identical barrier sequence in all six wavefronts
  switch (wv) {
    case 0: plaq_nu_sweep<2, 1, 0, INM>(k, sr, si); break;
    case 1: plaq_nu_sweep<3, 1, 1, INM>(k, sr, si); break;
    case 2: plaq_nu_sweep<3, 2, 2, INM>(k, sr, si); break;
    case 3: plaq_nu_sweep<1, 0, 3, INM>(k, sr, si); break;
    case 4: plaq_nu_sweep<2, 0, 4, INM>(k, sr, si); break;
    default: plaq_nu_sweep<3, 0, 5, INM>(k, sr, si); break;
  }
  // fixed-order block reduction: wave butterfly, then the six wave partials through LDS
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { sr += __shfl_down(sr, off, 64); si += __shfl_down(si, off, 64); }
  double* red = reinterpret_cast<double*>(fr_lds + kPqRed);
  __syncthreads();
  if (k.lt == 0) { red[2 * wv] = sr; red[2 * wv + 1] = si; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ar = 0.0, ai = 0.0;
    for (int i = 0; i < 6; ++i) { ar += red[2 * i]; ai += red[2 * i + 1]; }
    partial[(c * per_chain + rr) * 2 + 0] = ar;
    partial[(c * per_chain + rr) * 2 + 1] = ai;
  }
}

template <int INM>
static void launch_pq_variant(const double2* xn, Dims d, int nb, int nsb, int tsplit, double* partial,
                              hipStream_t st) {
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute((const void*)su3_plaq_nu_kernel<INM>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, kPqLds);
  }
  hipLaunchKernelGGL((su3_plaq_nu_kernel<INM>), dim3((unsigned)((long)nb * nsb * tsplit)),
                     dim3(kPqThreads), kPqLds, st, xn, d, nsb, tsplit, tuning().xcd_swizzle, partial, 0);
}

int force_nu_inmask(const Dims& d);
bool force_nu_applicable(const Dims& d);

bool plaq_nu_applicable(const Dims& d) { return force_nu_applicable(d); }

// per_chain = (Vs / 64) * tsplit partial pairs per chain
long plaq_nu_per_chain(const Dims& d, int nb) {
  const int nsb = d.X * d.Y * d.Z / kRS;
  int tsplit = (int)cdiv(1024, (long)nb * nsb);
  if (tsplit > d.T) tsplit = d.T;
  if (tsplit < 1) tsplit = 1;
  const int tlen = (int)cdiv(d.T, tsplit);
  tsplit = (int)cdiv(d.T, tlen);
  return (long)nsb * tsplit;
}

void launch_plaq_nu(const double2* xn, Dims d, int nb, double* partial, hipStream_t st) {
  const int nsb = d.X * d.Y * d.Z / kRS;
  const int tsplit = (int)(plaq_nu_per_chain(d, nb) / nsb);
  switch (force_nu_inmask(d)) {
    case 7: launch_pq_variant<7>(xn, d, nb, nsb, tsplit, partial, st); break;
    case 6: launch_pq_variant<6>(xn, d, nb, nsb, tsplit, partial, st); break;
    case 4: launch_pq_variant<4>(xn, d, nb, nsb, tsplit, partial, st); break;
    default: launch_pq_variant<0>(xn, d, nb, nsb, tsplit, partial, st);
  }
}

}  // namespace l2q
