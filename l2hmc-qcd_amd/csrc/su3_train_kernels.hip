// su3_train_kernels.hip -- reverse-mode (VJP) kernels of the SU(3) L2HMC training step, gfx950.
//
// Cotangent convention (the one torch uses for complex tensors): for a real loss L and a
// complex quantity z, g_z = dL/dRe z + i dL/dIm z, i.e. dL = Re tr(g_z^H dz).  With it
//   W = A B      =>  g_A = g_W B^H,  g_B = A^H g_W
//   E = expm(A)  =>  g_A = L_exp(A^H)[g_E]            (Frechet derivative at A^H)
//   F = TAH(M)   =>  g_M = TAH(g_F)                   (orthogonal projector, self-adjoint)
// The reference obtains all of these from torch.autograd over torch.matrix_exp, the closed-form
// projectSU (group/su3/pytorch/utils.py:227-346) and autograd.grad(action) (lattice.py:299-308).
#include "l2q_common.hpp"
#include "su3_links.hpp"
#include "su3_train_math.hpp"

namespace l2q {

#ifndef L2Q_EXPB_STASH
#define L2Q_EXPB_STASH 1
#endif

// ------------------------------------------------------------------ x half-update (expm) VJP
// forward (l2q_su3_expm_mul): x' = keep (.) x + expm(eps v) @ ((1 - keep) (.) x)
//   g_x = keep (.) g + (1 - keep) (.) (E^H g);   g_E = g y^H,  y = (1 - keep) (.) x
//   g_A = L_exp((eps v)^H)[g_E];  g_v += eps g_A;  d eps = sum Re tr(g_A^H v)
__global__ __launch_bounds__(kBlock, 2) void su3_expm_mul_bwd_kernel(
    const double2* __restrict__ xn, const double2* __restrict__ vn, double eps,
    const float* __restrict__ mask, int complement, const double2* __restrict__ gxn, double2* gx,
    double2* gv, int V, long nblk, double* __restrict__ partial) {
  __shared__ double lds[4];
#if L2Q_EXPB_STASH
  // g and v are needed before AND after the Frechet derivative and cannot stay in registers across it: each
  // thread parks its copies in its own LDS slots (entry-major: conflict-free 16-byte accesses, no barrier --
  // nobody else touches them) instead of fetching them a second time (the in-flight working set of an XCD,
  // ~7 MB, does not fit its 4 MB L2: the second fetch went to the fabric).  72 KiB: two workgroups per CU.
  extern __shared__ __attribute__((aligned(16))) double2 stash[];
  double2* sg = stash + threadIdx.x;
  double2* sv = stash + 9 * kBlock + threadIdx.x;
#endif
  const long f = blockIdx.x / nblk, blk = blockIdx.x % nblk;     // f = chain*4 + mu
  const int s = (int)blk * kBlock + threadIdx.x;
  const int mu = (int)(f & 3);
  double de = 0.0;
  if (s < V) {
    // Register budget: the Frechet derivative keeps six 3x3 matrices live (216 VGPRs); x, v, g and
    // the mask are therefore NOT carried across it but read again afterwards (L2 hits) -- two
    // wavefronts per SIMD instead of one.
    auto keep_of = [&](int i) -> double {
      double k = 0.0;
      if (mask != nullptr) {
        k = (double)mask[(mu * 9 + i) * (long)V + s];
        if (complement) k = 1.0 - k;
      }
      return k;
    };
    M3 gE, B;
    {
      M3 x, g;
      load_link(x, xn + f * 9L * V, V, s);
      load_link(g, gxn + f * 9L * V, V, s);
#if L2Q_EXPB_STASH
#pragma unroll
      for (int i = 0; i < 9; ++i) sg[i * kBlock] = make_double2(g.re[i], g.im[i]);
#endif
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const double k1 = 1.0 - keep_of(i);
        x.re[i] *= k1; x.im[i] *= k1;                    // y = (1 - keep) (.) x
      }
      m3_mul_na(gE, g, x);                               // g y^H
    }
    {
      M3 v;
      load_link(v, vn + f * 9L * V, V, s);
#if L2Q_EXPB_STASH
#pragma unroll
      for (int i = 0; i < 9; ++i) sv[i * kBlock] = make_double2(v.re[i], v.im[i]);
#endif
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {                    // B = (eps v)^H
          B.re[3 * i + j] = eps * v.re[3 * j + i]; B.im[3 * i + j] = -eps * v.im[3 * j + i];
        }
    }
    M3 EB, gA;
#ifdef L2Q_FRECHET_SERIES
    m3_expm_frechet_series(EB, gA, B, gE);             // (A/B build: the matrix-valued Taylor recursion)
#else
    m3_expm_frechet(EB, gA, B, gE);                    // EB = expm(eps v)^H
#endif
    {
      M3 g, gy;
#if L2Q_EXPB_STASH
#pragma unroll
      for (int i = 0; i < 9; ++i) { const double2 d = sg[i * kBlock]; g.re[i] = d.x; g.im[i] = d.y; }
#else
      load_link(g, gxn + f * 9L * V, V, s);
#endif
      m3_mul_nn(gy, EB, g);                              // E^H g
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const double k = keep_of(i);
        gy.re[i] = k * g.re[i] + (1.0 - k) * gy.re[i];
        gy.im[i] = k * g.im[i] + (1.0 - k) * gy.im[i];
      }
      store_link(gx + f * 9L * V, V, s, gy);
    }
    {
      M3 v;
#if L2Q_EXPB_STASH
#pragma unroll
      for (int i = 0; i < 9; ++i) { const double2 d = sv[i * kBlock]; v.re[i] = d.x; v.im[i] = d.y; }
#else
      load_link(v, vn + f * 9L * V, V, s);
#endif
      de = m3_inner(gA, v);
    }
    double2* gvf = gv + f * 9L * V;
#pragma unroll
    for (int e = 0; e < 9; ++e) {
      double2 o = gvf[e * (long)V + s];
      o.x = fma(eps, gA.re[e], o.x); o.y = fma(eps, gA.im[e], o.y);
      gvf[e * (long)V + s] = o;
    }
  }
  const double r = block_sum(de, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// ------------------------------------------------------------------ both x half-updates of a leapfrog step, VJP
// forward (l2q_su3_expm_mul2): x' = k1 (.) x + E y1,  x'' = k2 (.) x' + E y2   with E = expm(eps v) ONCE,
//   k1 = the mask (or its complement), k2 = 1 - k1, y1 = (1 - k1) (.) x, y2 = (1 - k2) (.) x'.
// The Frechet derivative is linear in its direction: both halves share ONE evaluation,
//   g'  = k2 (.) g  + (1 - k2) (.) (E^H g),   g'' = k1 (.) g' + (1 - k1) (.) (E^H g')   (= g_x),
//   g_A = L_exp((eps v)^H)[ g y2^H + g' y1^H ],   g_v += eps g_A,   d eps = sum Re tr(g_A^H v).
// One exponential (Cayley-Hamilton) + one derivative instead of two of each; x and v are parked in LDS
// (each thread its own slots) between their two uses.
__global__ __launch_bounds__(kBlock, 2) void su3_expm_mul2_bwd_kernel(
    const double2* __restrict__ xn, const double2* __restrict__ vn, double eps,
    const float* __restrict__ mask, int complement_first, const double2* __restrict__ gxn, double2* gx,
    double2* gv, int V, long nblk, double* __restrict__ partial) {
  __shared__ double lds[4];
  extern __shared__ __attribute__((aligned(16))) double2 stash[];
  double2* sx = stash + threadIdx.x;
  double2* sv = stash + 9 * kBlock + threadIdx.x;
  const long f = blockIdx.x / nblk, blk = blockIdx.x % nblk;     // f = chain*4 + mu
  const int s = (int)blk * kBlock + threadIdx.x;
  const int mu = (int)(f & 3);
  double de = 0.0;
  if (s < V) {
    float k1[9];                                       // 0 / 1: exact in fp32
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const float k = mask[(mu * 9 + i) * (long)V + s];
      k1[i] = complement_first ? 1.0f - k : k;
    }
    M3 E;
    {
      M3 v, a;
      load_link(v, vn + f * 9L * V, V, s);
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        sv[i * kBlock] = make_double2(v.re[i], v.im[i]);
        a.re[i] = eps * v.re[i]; a.im[i] = eps * v.im[i];
      }
      m3_expm(E, a);
    }
    M3 gE, gp;                                         // g y2^H (+ g' y1^H), g'
    {
      M3 x, y, xp;
      load_link(x, xn + f * 9L * V, V, s);
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        sx[i * kBlock] = make_double2(x.re[i], x.im[i]);
        const double m1 = 1.0 - (double)k1[i];
        y.re[i] = m1 * x.re[i]; y.im[i] = m1 * x.im[i];
      }
      m3_mul_nn(xp, E, y);                               // x' = k1 x + E y1
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const double k = (double)k1[i];
        xp.re[i] = fma(k, x.re[i], xp.re[i]); xp.im[i] = fma(k, x.im[i], xp.im[i]);
        // y2 = (1 - k2) x' = k1 x'
        y.re[i] = k * xp.re[i]; y.im[i] = k * xp.im[i];
      }
      M3 g, t;
      load_link(g, gxn + f * 9L * V, V, s);
      m3_mul_na(gE, g, y);                               // g y2^H
      m3_mul_an(t, E, g);                                // E^H g
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const double k2 = 1.0 - (double)k1[i];
        gp.re[i] = k2 * g.re[i] + (1.0 - k2) * t.re[i];
        gp.im[i] = k2 * g.im[i] + (1.0 - k2) * t.im[i];
      }
    }
    {
      M3 y, t;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const double2 d = sx[i * kBlock];
        const double m1 = 1.0 - (double)k1[i];
        y.re[i] = m1 * d.x; y.im[i] = m1 * d.y;          // y1
      }
      m3_mac_na(gE, gp, y);                              // + g' y1^H
      m3_mul_an(t, E, gp);                               // E^H g'
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const double k = (double)k1[i];
        t.re[i] = k * gp.re[i] + (1.0 - k) * t.re[i];
        t.im[i] = k * gp.im[i] + (1.0 - k) * t.im[i];
      }
      store_link(gx + f * 9L * V, V, s, t);
    }
    M3 B;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {                      // B = (eps v)^H
        const double2 d = sv[(3 * j + i) * kBlock];
        B.re[3 * i + j] = eps * d.x; B.im[3 * i + j] = -eps * d.y;
      }
    M3 EB, gA;
    m3_expm_frechet(EB, gA, B, gE);
    {
      M3 v;
#pragma unroll
      for (int i = 0; i < 9; ++i) { const double2 d = sv[i * kBlock]; v.re[i] = d.x; v.im[i] = d.y; }
      de = m3_inner(gA, v);
    }
    double2* gvf = gv + f * 9L * V;
#pragma unroll
    for (int e = 0; e < 9; ++e) {
      double2 o = gvf[e * (long)V + s];
      o.x = fma(eps, gA.re[e], o.x); o.y = fma(eps, gA.im[e], o.y);
      gvf[e * (long)V + s] = o;
    }
  }
  const double r = block_sum(de, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// ------------------------------------------------------------------ projectSU -> vec8 VJP
#ifndef L2Q_PVB_OCC
#define L2Q_PVB_OCC 2      // two wavefronts per SIMD, no spills: 0.762-0.772 ms against 0.775-0.813 at three (16 spilled registers), 0.85-0.88 at four
#endif
__global__ __launch_bounds__(kBlock, L2Q_PVB_OCC) void su3_projsu_vec8_bwd_kernel(
    const double2* __restrict__ in, const double* __restrict__ gvec, double2* gm, int V,
    long nblk) {
  const long f = blockIdx.x / nblk, blk = blockIdx.x % nblk;
  const int s = (int)blk * kBlock + threadIdx.x;
  if (s >= V) return;
  M3 m, g;
  load_link(m, in + f * 9L * V, V, s);
  double gy[8];
#pragma unroll
  for (int a = 0; a < 8; ++a) gy[a] = gvec[(f * 8 + a) * (long)V + s];
  m3_projsu_vec8_vjp(g, m, gy);
  double2* o = gm + f * 9L * V;
#pragma unroll
  for (int e = 0; e < 9; ++e) {
    double2 r = o[e * (long)V + s];
    r.x += g.re[e]; r.y += g.im[e];
    o[e * (long)V + s] = r;
  }
}

// su3_force_link.hip
bool force_link_applicable(const Dims& d);
void launch_force_link_bwd(const double2* xn, Dims d, int nb, double coef, const double2* gf, double2* gx,
                           hipStream_t st);

// ------------------------------------------------------------------ staple-type VJPs
// Both use the up / down staples of link (s, mu) in direction nu
//   S_up = U_nu(s+mu) U_mu(s+nu)^H U_nu(s)^H,  S_dn = U_nu(s+mu-nu)^H U_mu(s-nu)^H U_nu(s-nu).
// MODE 0 (force VJP; the reference's force is TAH(D x^H) with D = dS/dx held constant,
//   lattice/su3/pytorch/lattice.py:299-308):  g_x += coef TAH(g_F) (sum S)^H
// MODE 1 (plaquette-sum VJP, L = sum_p Re(conj(w_p) tr P_p), planes p = (u > v) in the order
//   (1,0) (2,0) (2,1) (3,0) (3,1) (3,2)):
//   mu > nu: g_x += w_p S_up^H + conj(w_p) S_dn^H;   mu < nu: g_x += conj(w_p) S_up^H + w_p S_dn^H
// MODE 2 (VJP of the per-site trace FIELD of LatticeSU3.wilson_loops, lattice.py:242-244): as MODE 1 with a
//   weight per (plane, chain, site), w[(p nb + c) V + site] complex: the up staple closes the plaquette based
//   at s, the down staple the one based at s - nu
__device__ __forceinline__ int plane_index(int u, int v) { return u * (u - 1) / 2 + v; }

__device__ __forceinline__ void mac_scaled_adjoint(M3& acc, double wr, double wi, const M3& sm) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double ar = sm.re[3 * j + i], ai = -sm.im[3 * j + i];     // (S^H)_ij
      acc.re[3 * i + j] += wr * ar - wi * ai;
      acc.im[3 * i + j] += wr * ai + wi * ar;
    }
}

template <int MODE>
__global__ __launch_bounds__(kBlock, 2) void su3_staple_bwd_kernel(
    const double2* __restrict__ xn, Dims d, long nblk, int swz, double coef,
    const double2* __restrict__ gf, const double* __restrict__ w, double2* gx) {
  const long total = (long)gridDim.x;
  const long wk = xcd_swizzle(blockIdx.x, total, swz);
  const int mu = (int)(wk & 3);
  const long cb = wk >> 2;
  const long c = cb / nblk, blk = cb % nblk;
  const int s = (int)blk * kBlock + threadIdx.x;
  if (s >= d.V) return;
  const int V = d.V;
  const double2* xc = xn + c * 36L * V;
  const double2* fm = xc + mu * 9 * V;
  const Site p = site_coords(s, d);
  const int cmu = coord_of(p, mu);
  const int s_pmu = fwd(s, cmu, d, mu);
  M3 acc;
  m3_zero(acc);
#pragma unroll 1
  for (int nu = 0; nu < 4; ++nu) {
    if (nu == mu) continue;
    const double2* fn = xc + nu * 9 * V;
    const int cnu = coord_of(p, nu);
    const int s_pnu = fwd(s, cnu, d, nu);
    const int s_mnu = bwd(s, cnu, d, nu);
    const int s_pmu_mnu = bwd(s_pmu, cnu, d, nu);
    double wr = 1.0, wi = 0.0;
    double wdr = 1.0, wdi = 0.0;                       // the down staple's weight (MODE 2: another site)
    if (MODE == 1) {
      const int pl = mu > nu ? plane_index(mu, nu) : plane_index(nu, mu);
      wr = w[(c * 6 + pl) * 2 + 0];
      wi = w[(c * 6 + pl) * 2 + 1];
      if (mu < nu) wi = -wi;                           // conj(w) multiplies S_up^H when mu < nu
      wdr = wr; wdi = wi;
    }
    if (MODE == 2) {
      const int pl = mu > nu ? plane_index(mu, nu) : plane_index(nu, mu);
      const long nbl = (long)(total / (4 * nblk));
      const double* wp = w + ((pl * nbl + c) * V) * 2;
      wr = wp[2L * s]; wi = wp[2L * s + 1];
      wdr = wp[2L * s_mnu]; wdi = wp[2L * s_mnu + 1];
      if (mu < nu) { wi = -wi; wdi = -wdi; }
    }
    M3 a, b, t, st;
    load_link(a, fn, V, s_pmu);
    load_link(b, fm, V, s_pnu);
    m3_mul_na(t, a, b);
    load_link(a, fn, V, s);
    m3_mul_na(st, t, a);                               // S_up
    mac_scaled_adjoint(acc, wr, wi, st);
    load_link(a, fn, V, s_pmu_mnu);
    load_link(b, fm, V, s_mnu);
    m3_mul_aa(t, a, b);
    load_link(a, fn, V, s_mnu);
    m3_mul_nn(st, t, a);                               // S_dn
    mac_scaled_adjoint(acc, wdr, -wdi, st);
  }
  M3 g;
  if (MODE == 0) {
    M3 gfl, tg;
    load_link(gfl, gf + (c * 4 + mu) * 9L * V, V, s);
    m3_tah(tg, gfl);
    m3_mul_nn(g, tg, acc);                             // TAH(g_F) (sum S)^H  (acc holds the adjoint sum)
#pragma unroll
    for (int i = 0; i < 9; ++i) { g.re[i] *= coef; g.im[i] *= coef; }
  } else {
    g = acc;
  }
  double2* o = gx + (c * 4 + mu) * 9L * V;
#pragma unroll
  for (int e = 0; e < 9; ++e) {
    double2 r = o[e * (long)V + s];
    r.x += g.re[e]; r.y += g.im[e];
    o[e * (long)V + s] = r;
  }
}

// ------------------------------------------------------------------ complex v-update VJP
// v, F, g_v complex [nb][n]; s, t, q real [nb][n] (dynamics.py:1266-1297 on complex momenta)
// ACC: (dF, ds, dt, dq) = this update's cotangents + those of the v-update that SHARED its force and heads
// (aF, as, at, aq; the tape's primary / secondary pairs, training.py): one pass instead of four axpy launches
template <bool FWD, bool ACC = false>
__global__ __launch_bounds__(kBlock) void v_update_bwd_cplx_kernel(
    const double2* __restrict__ v, const double2* __restrict__ force, const double* __restrict__ s,
    const double* __restrict__ t, const double* __restrict__ q, double eps,
    const double2* __restrict__ gv, const double* __restrict__ gl, long n, long nblk,
    double2* __restrict__ dv, double2* __restrict__ dF, double* __restrict__ ds,
    double* __restrict__ dt, double* __restrict__ dq, double* __restrict__ partial,
    const double2* __restrict__ aF = nullptr, const double* __restrict__ as = nullptr,
    const double* __restrict__ at = nullptr, const double* __restrict__ aq = nullptr) {
  __shared__ double lds[4];
  const long c = blockIdx.x / nblk, blk = blockIdx.x % nblk;
  const long j = blk * kBlock + threadIdx.x;
  double de = 0.0;
  if (j < n) {
    const long o = c * n + j;
    const double glc = gl ? gl[c] : 0.0;
    const double2 vj = v[o], fj = force[o], g = gv[o];
    const double sj = s[o], tj = t[o], qj = q[o];
    const double S = FWD ? 0.5 * eps * sj : -0.5 * eps * sj;
    const double es = exp(S), eq = exp(eps * qj);
    const double fqr = fj.x * eq + tj, fqi = fj.y * eq;
    double dS, dBr, dBi;
    if (FWD) {
      dS = es * (g.x * vj.x + g.y * vj.y) + glc;
      dBr = -g.x; dBi = -g.y;
    } else {
      const double wr = vj.x + 0.5 * eps * fqr, wi = vj.y + 0.5 * eps * fqi;
      dS = es * (g.x * wr + g.y * wi) + glc;
      dBr = g.x * es; dBi = g.y * es;
    }
    dv[o] = make_double2(g.x * es, g.y * es);
    const double dQ = 0.5 * eps * eq * (dBr * fj.x + dBi * fj.y);
    double2 oF = make_double2(dBr * 0.5 * eps * eq, dBi * 0.5 * eps * eq);
    double ot = 0.5 * eps * dBr, os = FWD ? 0.5 * eps * dS : -0.5 * eps * dS, oq = eps * dQ;
    if (ACC) {
      // (same order as the axpy it replaces: x += p)
      const double2 pF = aF[o];
      oF.x += pF.x; oF.y += pF.y;
      os += as[o]; ot += at[o]; oq += aq[o];
    }
    dF[o] = oF;
    dt[o] = ot;
    ds[o] = os;
    dq[o] = oq;
    de = 0.5 * (dBr * fqr + dBi * fqi) + (FWD ? 0.5 * sj * dS : -0.5 * sj * dS) + qj * dQ;
  }
  const double r = block_sum(de, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// One complex v-update VJP at one entry (the arithmetic of v_update_bwd_cplx_kernel, same order).
template <bool FWD>
__device__ __forceinline__ void vub_entry(double2 vj, double2 fj, double sj, double tj, double qj, double eps,
                                          double2 g, double glc, double2& dv, double2& dF, double& ds,
                                          double& dt, double& dq, double& de) {
  const double S = FWD ? 0.5 * eps * sj : -0.5 * eps * sj;
  const double es = exp(S), eq = exp(eps * qj);
  const double fqr = fj.x * eq + tj, fqi = fj.y * eq;
  double dS, dBr, dBi;
  if (FWD) {
    dS = es * (g.x * vj.x + g.y * vj.y) + glc;
    dBr = -g.x; dBi = -g.y;
  } else {
    const double wr = vj.x + 0.5 * eps * fqr, wi = vj.y + 0.5 * eps * fqi;
    dS = es * (g.x * wr + g.y * wi) + glc;
    dBr = g.x * es; dBi = g.y * es;
  }
  dv = make_double2(g.x * es, g.y * es);
  const double dQ = 0.5 * eps * eq * (dBr * fj.x + dBi * fj.y);
  dF = make_double2(dBr * 0.5 * eps * eq, dBi * 0.5 * eps * eq);
  dt = 0.5 * eps * dBr;
  ds = FWD ? 0.5 * eps * dS : -0.5 * eps * dS;
  dq = eps * dQ;
  de = 0.5 * (dBr * fqr + dBi * fqi) + (FWD ? 0.5 * sj * dS : -0.5 * sj * dS) + qj * dQ;
}

// The two v-updates that share one force and one network evaluation (the closing update of a leapfrog step and
// the opening one of the next, optionally with the momentum flip between them), reversed in ONE pass:
//   v_mid = U1(v1), [v_mid <- -v_mid], v_out = U2(v_mid):   g -> (U2 VJP at v_mid) -> [sign] -> (U1 VJP at v1).
// F, s, t, q are read once and (dF, ds, dt, dq) written once as the sum of both updates' cotangents (first
// update's + second's, the order of the hand-over it replaces): 144 instead of 296 bytes per entry.
template <bool FWD1, bool FWD2>
__global__ __launch_bounds__(kBlock) void v_update_bwd_pair_cplx_kernel(
    const double2* __restrict__ v1, const double2* __restrict__ vmid, const double2* __restrict__ force,
    const double* __restrict__ s, const double* __restrict__ t, const double* __restrict__ q, double eps1,
    double eps2, int flip, const double2* __restrict__ gv, const double* __restrict__ gl, long n, long nblk,
    double2* __restrict__ dv, double2* __restrict__ dF, double* __restrict__ ds, double* __restrict__ dt,
    double* __restrict__ dq, double* __restrict__ partial1, double* __restrict__ partial2) {
  __shared__ double lds[8];
  const long c = blockIdx.x / nblk, blk = blockIdx.x % nblk;
  const long j = blk * kBlock + threadIdx.x;
  double de1 = 0.0, de2 = 0.0;
  if (j < n) {
    const long o = c * n + j;
    const double glc = gl ? gl[c] : 0.0;
    const double2 fj = force[o];
    const double sj = s[o], tj = t[o], qj = q[o];
    double2 dvm, dF2, dv1, dF1;
    double ds2, dt2, dq2, ds1, dt1, dq1;
    vub_entry<FWD2>(vmid[o], fj, sj, tj, qj, eps2, gv[o], glc, dvm, dF2, ds2, dt2, dq2, de2);
    if (flip) { dvm.x = -dvm.x; dvm.y = -dvm.y; }
    vub_entry<FWD1>(v1[o], fj, sj, tj, qj, eps1, dvm, glc, dv1, dF1, ds1, dt1, dq1, de1);
    dv[o] = dv1;
    dF[o] = make_double2(dF1.x + dF2.x, dF1.y + dF2.y);
    ds[o] = ds1 + ds2;
    dt[o] = dt1 + dt2;
    dq[o] = dq1 + dq2;
  }
  const double r1 = block_sum(de1, lds);
  const double r2 = block_sum(de2, lds + 4);
  if (threadIdx.x == 0) { partial1[blockIdx.x] = r1; partial2[blockIdx.x] = r2; }
}

// g_x += 2 a[c] (x - y)  (cotangent of sum |x - y|^2, the rmse term of LatticeLoss)
__global__ void diff_bwd_kernel(const double* __restrict__ x, const double* __restrict__ y,
                                const double* __restrict__ a, long n, double* gx) {
  const long c = blockIdx.y;
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) gx[c * n + j] += 2.0 * a[c] * (x[c * n + j] - y[c * n + j]);
}

}  // namespace l2q

using namespace l2q;

static bool dims_ok4(int nb, int T, int X, int Y, int Z) {
  return nb > 0 && T > 0 && X > 0 && Y > 0 && Z > 0 && (double)T * X * Y * Z * 36.0 < 2.0e9;
}

extern "C" {

int l2q_su3_expm_mul_bwd(const void* xn, const void* vn, double eps, const float* mask_n,
                         int complement, const void* gxnew, void* gx, void* gv, double* deps,
                         int nb, long V, void* ws, size_t ws_bytes, void* stream) {
  L2Q_REQUIRE(xn && vn && gxnew && gx && gv && deps && ws, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && V > 0 && V <= 200000000L, L2Q_EINVAL, "bad size");
  const long nblk = cdiv(V, kBlock);
  L2Q_REQUIRE(ws_bytes >= (size_t)nb * 4 * nblk * sizeof(double), L2Q_EINVAL, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const size_t stash_bytes = L2Q_EXPB_STASH ? 2 * 9 * kBlock * sizeof(double2) : 0;
  static PerDeviceOnce attr_once;
  if (L2Q_EXPB_STASH && attr_once.first())
    (void)hipFuncSetAttribute((const void*)su3_expm_mul_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)stash_bytes);
  hipLaunchKernelGGL(su3_expm_mul_bwd_kernel, dim3((unsigned)(nb * 4L * nblk)), dim3(kBlock), stash_bytes, st,
                     (const double2*)xn, (const double2*)vn, eps, mask_n, complement,
                     (const double2*)gxnew, (double2*)gx, (double2*)gv, (int)V, nblk, (double*)ws);
  launch_finalize((const double*)ws, deps, nb, 4 * nblk, 1, 1.0, 0.0, st);
  return check_launch("l2q_su3_expm_mul_bwd");
}

int l2q_su3_expm_mul2_bwd(const void* xn, const void* vn, double eps, const float* mask_n,
                          int complement_first, const void* gxnew, void* gx, void* gv, double* deps,
                          int nb, long V, void* ws, size_t ws_bytes, void* stream) {
  L2Q_REQUIRE(xn && vn && mask_n && gxnew && gx && gv && deps && ws, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && V > 0 && V <= 200000000L, L2Q_EINVAL, "bad size");
  const long nblk = cdiv(V, kBlock);
  L2Q_REQUIRE(ws_bytes >= (size_t)nb * 4 * nblk * sizeof(double), L2Q_EINVAL, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const size_t stash_bytes = 2 * 9 * kBlock * sizeof(double2);
  static PerDeviceOnce attr_once;
  if (attr_once.first())
    (void)hipFuncSetAttribute((const void*)su3_expm_mul2_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)stash_bytes);
  hipLaunchKernelGGL(su3_expm_mul2_bwd_kernel, dim3((unsigned)(nb * 4L * nblk)), dim3(kBlock), stash_bytes, st,
                     (const double2*)xn, (const double2*)vn, eps, mask_n, complement_first,
                     (const double2*)gxnew, (double2*)gx, (double2*)gv, (int)V, nblk, (double*)ws);
  launch_finalize((const double*)ws, deps, nb, 4 * nblk, 1, 1.0, 0.0, st);
  return check_launch("l2q_su3_expm_mul2_bwd");
}

int l2q_su3_projsu_vec8_bwd(const void* in, const double* gvec, void* gm, long nfields, long V,
                            void* stream) {
  L2Q_REQUIRE(in && gvec && gm, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nfields > 0 && V > 0 && V <= 200000000L, L2Q_EINVAL, "bad size");
  const long nblk = cdiv(V, kBlock);
  hipLaunchKernelGGL(su3_projsu_vec8_bwd_kernel, dim3((unsigned)(nfields * nblk)), dim3(kBlock), 0,
                     (hipStream_t)stream, (const double2*)in, gvec, (double2*)gm, (int)V, nblk);
  return check_launch("l2q_su3_projsu_vec8_bwd");
}

int l2q_su3_force_bwd(const void* xn, const void* gf, double beta, void* gx, int nb, int T, int X,
                      int Y, int Z, void* stream) {
  L2Q_REQUIRE(xn && gf && gx, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(dims_ok4(nb, T, X, Y, Z), L2Q_EINVAL, "non-positive size");
  Dims d{T, X, Y, Z, T * X * Y * Z};
  // the slice-resident force sweep with the VJP epilogue (su3_force_link.hip, MODE 2) where the force itself
  // runs on it: the staple sum of the link from LDS-resident slices instead of 18 flat matrix loads per link
  if (tuning().force_tile >= 5 && force_link_applicable(d)) {
    launch_force_link_bwd((const double2*)xn, d, nb, beta / 3.0, (const double2*)gf, (double2*)gx,
                          (hipStream_t)stream);
    return check_launch("l2q_su3_force_bwd");
  }
  const long nblk = cdiv(d.V, kBlock);
  hipLaunchKernelGGL(su3_staple_bwd_kernel<0>, dim3((unsigned)(nb * nblk * 4)), dim3(kBlock), 0,
                     (hipStream_t)stream, (const double2*)xn, d, nblk, tuning().xcd_swizzle,
                     beta / 3.0, (const double2*)gf, (const double*)nullptr, (double2*)gx);
  return check_launch("l2q_su3_force_bwd");
}

int l2q_su3_plaq_bwd(const void* xn, const double* w, void* gx, int nb, int T, int X, int Y, int Z,
                     void* stream) {
  L2Q_REQUIRE(xn && w && gx, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(dims_ok4(nb, T, X, Y, Z), L2Q_EINVAL, "non-positive size");
  Dims d{T, X, Y, Z, T * X * Y * Z};
  const long nblk = cdiv(d.V, kBlock);
  hipLaunchKernelGGL(su3_staple_bwd_kernel<1>, dim3((unsigned)(nb * nblk * 4)), dim3(kBlock), 0,
                     (hipStream_t)stream, (const double2*)xn, d, nblk, tuning().xcd_swizzle, 1.0,
                     (const double2*)nullptr, w, (double2*)gx);
  return check_launch("l2q_su3_plaq_bwd");
}

int l2q_su3_wilson_loops_bwd(const void* xn, const void* w, void* gx, int nb, int T, int X, int Y, int Z,
                             void* stream) {
  L2Q_REQUIRE(xn && w && gx, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(dims_ok4(nb, T, X, Y, Z), L2Q_EINVAL, "non-positive size");
  Dims d{T, X, Y, Z, T * X * Y * Z};
  const long nblk = cdiv(d.V, kBlock);
  hipLaunchKernelGGL(su3_staple_bwd_kernel<2>, dim3((unsigned)(nb * nblk * 4)), dim3(kBlock), 0,
                     (hipStream_t)stream, (const double2*)xn, d, nblk, tuning().xcd_swizzle, 1.0,
                     (const double2*)nullptr, (const double*)w, (double2*)gx);
  return check_launch("l2q_su3_wilson_loops_bwd");
}

int l2q_v_update_bwd_c128(const void* v, const void* force, const double* s, const double* t,
                          const double* q, double eps, int forward, const void* gv,
                          const double* gl, int nb, long n, void* dv, void* dF, double* ds,
                          double* dt, double* dq, double* deps, void* ws, size_t ws_bytes,
                          void* stream) {
  L2Q_REQUIRE(v && force && s && t && q && gv && dv && dF && ds && dt && dq && deps && ws,
              L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && n > 0, L2Q_EINVAL, "non-positive size");
  const long nblk = cdiv(n, kBlock);
  L2Q_REQUIRE(ws_bytes >= (size_t)nb * nblk * sizeof(double), L2Q_EINVAL, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)(nb * nblk)), block(kBlock);
  if (forward)
    hipLaunchKernelGGL((v_update_bwd_cplx_kernel<true, false>), grid, block, 0, st, (const double2*)v,
                       (const double2*)force, s, t, q, eps, (const double2*)gv, gl, n, nblk,
                       (double2*)dv, (double2*)dF, ds, dt, dq, (double*)ws, (const double2*)nullptr,
                       (const double*)nullptr, (const double*)nullptr, (const double*)nullptr);
  else
    hipLaunchKernelGGL((v_update_bwd_cplx_kernel<false, false>), grid, block, 0, st, (const double2*)v,
                       (const double2*)force, s, t, q, eps, (const double2*)gv, gl, n, nblk,
                       (double2*)dv, (double2*)dF, ds, dt, dq, (double*)ws, (const double2*)nullptr,
                       (const double*)nullptr, (const double*)nullptr, (const double*)nullptr);
  launch_finalize((const double*)ws, deps, nb, nblk, 1, 1.0, 0.0, st);
  return check_launch("l2q_v_update_bwd_c128");
}

int l2q_v_update_bwd_acc_c128(const void* v, const void* force, const double* s, const double* t,
                              const double* q, double eps, int forward, const void* gv, const double* gl,
                              int nb, long n, const void* acc_dF, const double* acc_ds, const double* acc_dt,
                              const double* acc_dq, void* dv, void* dF, double* ds, double* dt, double* dq,
                              double* deps, void* ws, size_t ws_bytes, void* stream) {
  L2Q_REQUIRE(v && force && s && t && q && gv && dv && dF && ds && dt && dq && deps && ws, L2Q_EINVAL,
              "null pointer");
  L2Q_REQUIRE(acc_dF && acc_ds && acc_dt && acc_dq, L2Q_EINVAL, "null pointer (cotangents to add)");
  L2Q_REQUIRE(nb > 0 && n > 0, L2Q_EINVAL, "non-positive size");
  const long nblk = cdiv(n, kBlock);
  L2Q_REQUIRE(ws_bytes >= (size_t)nb * nblk * sizeof(double), L2Q_EINVAL, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)(nb * nblk)), block(kBlock);
  if (forward)
    hipLaunchKernelGGL((v_update_bwd_cplx_kernel<true, true>), grid, block, 0, st, (const double2*)v,
                       (const double2*)force, s, t, q, eps, (const double2*)gv, gl, n, nblk, (double2*)dv,
                       (double2*)dF, ds, dt, dq, (double*)ws, (const double2*)acc_dF, acc_ds, acc_dt, acc_dq);
  else
    hipLaunchKernelGGL((v_update_bwd_cplx_kernel<false, true>), grid, block, 0, st, (const double2*)v,
                       (const double2*)force, s, t, q, eps, (const double2*)gv, gl, n, nblk, (double2*)dv,
                       (double2*)dF, ds, dt, dq, (double*)ws, (const double2*)acc_dF, acc_ds, acc_dt, acc_dq);
  launch_finalize((const double*)ws, deps, nb, nblk, 1, 1.0, 0.0, st);
  return check_launch("l2q_v_update_bwd_acc_c128");
}

int l2q_v_update_bwd_pair_c128(const void* v1, const void* v_mid, const void* force, const double* s,
                               const double* t, const double* q, double eps1, int forward1, double eps2,
                               int forward2, int flip_between, const void* gv, const double* gl, int nb, long n,
                               void* dv, void* dF, double* ds, double* dt, double* dq, double* deps1,
                               double* deps2, void* ws, size_t ws_bytes, void* stream) {
  L2Q_REQUIRE(v1 && v_mid && force && s && t && q && gv && dv && dF && ds && dt && dq && deps1 && deps2 && ws,
              L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && n > 0, L2Q_EINVAL, "non-positive size");
  const long nblk = cdiv(n, kBlock);
  L2Q_REQUIRE(ws_bytes >= 2 * (size_t)nb * nblk * sizeof(double), L2Q_EINVAL, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)(nb * nblk)), block(kBlock);
  double* p1 = (double*)ws;
  double* p2 = p1 + (size_t)nb * nblk;
#define L2Q_VP(A, B)                                                                                      \
  hipLaunchKernelGGL((v_update_bwd_pair_cplx_kernel<A, B>), grid, block, 0, st, (const double2*)v1,        \
                     (const double2*)v_mid, (const double2*)force, s, t, q, eps1, eps2, flip_between,      \
                     (const double2*)gv, gl, n, nblk, (double2*)dv, (double2*)dF, ds, dt, dq, p1, p2)
  if (forward1 && forward2) L2Q_VP(true, true);
  else if (forward1) L2Q_VP(true, false);
  else if (forward2) L2Q_VP(false, true);
  else L2Q_VP(false, false);
#undef L2Q_VP
  launch_finalize(p1, deps1, nb, nblk, 1, 1.0, 0.0, st);
  launch_finalize(p2, deps2, nb, nblk, 1, 1.0, 0.0, st);
  return check_launch("l2q_v_update_bwd_pair_c128");
}

int l2q_diff_bwd_f64(const double* x, const double* y, const double* a, int nb, long n, double* gx,
                     void* stream) {
  L2Q_REQUIRE(x && y && a && gx, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && n > 0, L2Q_EINVAL, "non-positive size");
  hipLaunchKernelGGL(diff_bwd_kernel, dim3((unsigned)cdiv(n, kBlock), (unsigned)nb), dim3(kBlock), 0,
                     (hipStream_t)stream, x, y, a, n, gx);
  return check_launch("l2q_diff_bwd_f64");
}

}  // extern "C"
