// su3_train_math.hpp -- per-link cotangent algebra of the SU(3) training kernels (device code;
// also compiled for the host by tests/native_host/ to check the formulas without a GPU).
#pragma once
#include "su3_math.hpp"

namespace l2q {

#ifdef __HIP_DEVICE_COMPILE__
__device__ __forceinline__ bool l2q_wave_any(bool p) { return __any(p) != 0; }
// 1 / d and 1 / sqrt(x) for finite positive arguments from the hardware seeds (v_rcp_f64 / v_rsq_f64, ~2^-26)
// and two Newton steps each: ~1 ulp in 5 / 8 instructions where the IEEE division and square root sequences take
// ~13 and ~20 (the Jacobi rotations of the projectSU VJP spend a third of their instructions there)
__device__ __forceinline__ double l2q_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(r, fma(-d, r, 1.0), r);
  return fma(r, fma(-d, r, 1.0), r);
}
__device__ __forceinline__ double l2q_rsqrt(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = r * fma(-0.5 * x, r * r, 1.5);
  return r * fma(-0.5 * x, r * r, 1.5);
}
#else
__device__ __forceinline__ bool l2q_wave_any(bool p) { return p; }      // host pass / host test build
__device__ __forceinline__ double l2q_rcp(double d) { return 1.0 / d; }
__device__ __forceinline__ double l2q_rsqrt(double x) { return 1.0 / sqrt(x); }
#endif

__device__ __forceinline__ void m3_adjoint(M3& r, const M3& a) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) { r.re[3 * i + j] = a.re[3 * j + i]; r.im[3 * i + j] = -a.im[3 * j + i]; }
}

// sum_ij Re(conj(a_ij) b_ij)
__device__ __forceinline__ double m3_inner(const M3& a, const M3& b) {
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 9; ++i) s += a.re[i] * b.re[i] + a.im[i] * b.im[i];
  return s;
}

// E = exp(B) and L = L_exp(B)[G] (Frechet derivative of the exponential at B in direction G) through the
// Cayley-Hamilton form the forward kernel uses (m3_expm): with X = B / 2^s, |X|_F < 0.25,
//   exp(X) = f0 I + f1 X + f2 X^2,   f_i analytic in the invariants p2 = tr X, p1 = (p2^2 - tr X^2) / 2, p0 = det X,
//   L_exp(X)[G] = df0 I + df1 X + df2 X^2 + f1 G + f2 (X G + G X),
//   dp2 = tr G,  dp1 = p2 tr G - tr(X G),  dp0 = tr(adj(X) G) = tr(X^2 G) - p2 tr(X G) + p1 tr G,
// where (df0, df1, df2) is the tangent of the scalar recursion A^(n+1)/(n+1)! = [c p0, a - c p1, b + c p2] / (n+1)
// carried next to its state: three 3x3 products and 12 scalar steps instead of the 13 x 3 products of the
// matrix-valued Taylor recursion below (4.2k -> 1.5k fp64 FMAs per link before the squarings), and four
// live matrices instead of six.  Squarings: L <- E L + L E, E <- E E.  Valid for any complex 3x3 B.
__device__ __forceinline__ void m3_expm_frechet(M3& E, M3& L, const M3& B, const M3& G) {
  double n2 = 0.0;
#pragma unroll
  for (int i = 0; i < 9; ++i) n2 += B.re[i] * B.re[i] + B.im[i] * B.im[i];
  const double nrm = sqrt(n2);
  int s = 0;
  double scale = 1.0;
  if (nrm > 0.25) {
    int ex;
    (void)frexp(nrm, &ex);
    s = ex + 2;                                      // nrm / 2^s in [0.125, 0.25)
    if (s > 60) s = 60;
    scale = ldexp(1.0, -s);
  }
  M3 X, Gs, X2;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    X.re[i] = B.re[i] * scale; X.im[i] = B.im[i] * scale;
    Gs.re[i] = G.re[i] * scale; Gs.im[i] = G.im[i] * scale;
  }
  m3_mul_nn(X2, X, X);
  const double p2r = X.re[0] + X.re[4] + X.re[8], p2i = X.im[0] + X.im[4] + X.im[8];
  const double t2r = X2.re[0] + X2.re[4] + X2.re[8], t2i = X2.im[0] + X2.im[4] + X2.im[8];
  double sqr, sqi;
  cmul(sqr, sqi, p2r, p2i, p2r, p2i);
  const double p1r = 0.5 * (sqr - t2r), p1i = 0.5 * (sqi - t2i);
  double p0r, p0i;
  m3_det(p0r, p0i, X);
  // tr G, tr(X G), tr(X^2 G)
  const double g0r = Gs.re[0] + Gs.re[4] + Gs.re[8], g0i = Gs.im[0] + Gs.im[4] + Gs.im[8];
  double g1r = 0.0, g1i = 0.0, g2r = 0.0, g2i = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int e = 3 * i + j, et = 3 * j + i;
      g1r = fma(X.re[e], Gs.re[et], g1r); g1r = fma(-X.im[e], Gs.im[et], g1r);
      g1i = fma(X.re[e], Gs.im[et], g1i); g1i = fma(X.im[e], Gs.re[et], g1i);
      g2r = fma(X2.re[e], Gs.re[et], g2r); g2r = fma(-X2.im[e], Gs.im[et], g2r);
      g2i = fma(X2.re[e], Gs.im[et], g2i); g2i = fma(X2.im[e], Gs.re[et], g2i);
    }
  double xr, xi, yr, yi, zr, zi;
  const double q2r = g0r, q2i = g0i;                                   // dp2
  cmul(xr, xi, p2r, p2i, g0r, g0i);
  const double q1r = xr - g1r, q1i = xi - g1i;                         // dp1
  cmul(yr, yi, p2r, p2i, g1r, g1i);
  cmul(zr, zi, p1r, p1i, g0r, g0i);
  const double q0r = g2r - yr + zr, q0i = g2i - yi + zi;               // dp0
  // state (a, b, c) = coefficients of X^n / n! on {I, X, X^2}, tangent (da, db, dc); n = 2: X^2 / 2
  double ar = 0.0, ai = 0.0, br = 0.0, bi = 0.0, cr = 0.5, ci = 0.0;
  double dar = 0.0, dai = 0.0, dbr = 0.0, dbi = 0.0, dcr = 0.0, dci = 0.0;
  double f0r = 1.0, f0i = 0.0, f1r = 1.0, f1i = 0.0, f2r = 0.5, f2i = 0.0;
  double d0r = 0.0, d0i = 0.0, d1r = 0.0, d1i = 0.0, d2r = 0.0, d2i = 0.0;
  // 0.25^n / n! < 3e-18 from n = 13 on; the tangent term of order n is ~ 0.25^(n-1) / (n-1)!: terms up to n = 14
  static constexpr double kInvN[16] = {1.0 / 1, 1.0 / 2, 1.0 / 3, 1.0 / 4, 1.0 / 5, 1.0 / 6, 1.0 / 7, 1.0 / 8,
                                       1.0 / 9, 1.0 / 10, 1.0 / 11, 1.0 / 12, 1.0 / 13, 1.0 / 14, 1.0 / 15, 1.0 / 16};
#pragma unroll 1
  for (int n = 2; n < 14; ++n) {
    const double inv = kInvN[n];                                       // 1 / (n + 1)
    double ur, ui, vr, vi, wr, wi;
    // tangent first (it reads the old state)
    cmul(xr, xi, dcr, dci, p0r, p0i); cmul(ur, ui, cr, ci, q0r, q0i);
    cmul(yr, yi, dcr, dci, p1r, p1i); cmul(vr, vi, cr, ci, q1r, q1i);
    cmul(zr, zi, dcr, dci, p2r, p2i); cmul(wr, wi, cr, ci, q2r, q2i);
    const double ndar = (xr + ur) * inv, ndai = (xi + ui) * inv;
    const double ndbr = (dar - yr - vr) * inv, ndbi = (dai - yi - vi) * inv;
    const double ndcr = (dbr + zr + wr) * inv, ndci = (dbi + zi + wi) * inv;
    cmul(xr, xi, cr, ci, p0r, p0i);
    cmul(yr, yi, cr, ci, p1r, p1i);
    cmul(zr, zi, cr, ci, p2r, p2i);
    const double nar = xr * inv, nai = xi * inv;
    const double nbr = (ar - yr) * inv, nbi = (ai - yi) * inv;
    const double ncr = (br + zr) * inv, nci = (bi + zi) * inv;
    ar = nar; ai = nai; br = nbr; bi = nbi; cr = ncr; ci = nci;
    dar = ndar; dai = ndai; dbr = ndbr; dbi = ndbi; dcr = ndcr; dci = ndci;
    f0r += ar; f0i += ai; f1r += br; f1i += bi; f2r += cr; f2i += ci;
    d0r += dar; d0i += dai; d1r += dbr; d1i += dbi; d2r += dcr; d2i += dci;
  }
  M3 T;
  m3_mul_nn(T, X, Gs);
  m3_mac_nn(T, Gs, X);                                                 // X G + G X
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    cmul(xr, xi, f1r, f1i, X.re[i], X.im[i]);
    cmul(yr, yi, f2r, f2i, X2.re[i], X2.im[i]);
    E.re[i] = xr + yr; E.im[i] = xi + yi;
    cmul(xr, xi, d1r, d1i, X.re[i], X.im[i]);
    cmul(yr, yi, d2r, d2i, X2.re[i], X2.im[i]);
    cmul(zr, zi, f1r, f1i, Gs.re[i], Gs.im[i]);
    double wr, wi;
    cmul(wr, wi, f2r, f2i, T.re[i], T.im[i]);
    L.re[i] = (xr + yr) + (zr + wr); L.im[i] = (xi + yi) + (zi + wi);
  }
  E.re[0] += f0r; E.re[4] += f0r; E.re[8] += f0r;
  E.im[0] += f0i; E.im[4] += f0i; E.im[8] += f0i;
  L.re[0] += d0r; L.re[4] += d0r; L.re[8] += d0r;
  L.im[0] += d0i; L.im[4] += d0i; L.im[8] += d0i;
#pragma unroll 1
  for (int k = 0; k < s; ++k) {
    M3 t;
    m3_mul_nn(t, E, L);
    m3_mac_nn(t, L, E);
    L = t;
    m3_mul_nn(t, E, E);
    E = t;
  }
}

// The round-2 form of the same pair (kept for A/B, -DL2Q_FRECHET_SERIES, and as the cross-check of the
// host test): E = exp(B) and L = L_exp(B)[G]
// scaling & squaring on the Taylor series, M_n = M_{n-1} X + X^{n-1} G,
// L_0 = sum M_n / n!, then L <- E L + L E, E <- E E per squaring.
__device__ __forceinline__ void m3_expm_frechet_series(M3& E, M3& L, const M3& B, const M3& G) {
  double n2 = 0.0;
#pragma unroll
  for (int i = 0; i < 9; ++i) n2 += B.re[i] * B.re[i] + B.im[i] * B.im[i];
  const double nrm = sqrt(n2);
  int s = 0;
  double scale = 1.0;
  if (nrm > 0.25) {
    int ex;
    (void)frexp(nrm, &ex);
    s = ex + 2;                                      // nrm / 2^s in [0.125, 0.25)
    if (s > 60) s = 60;
    scale = ldexp(1.0, -s);
  }
  M3 X, Gs, P, M;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    X.re[i] = B.re[i] * scale; X.im[i] = B.im[i] * scale;
    Gs.re[i] = G.re[i] * scale; Gs.im[i] = G.im[i] * scale;
  }
  m3_identity(E);
  m3_zero(L);
  m3_identity(P);
  m3_zero(M);
  // 1 / n! by the same successive divisions as before, evaluated at compile time (a rolled loop cannot fold
  // `f /= n` and pays an IEEE fp64 division sequence per step)
  struct InvFact {
    double v[14];
    constexpr InvFact() : v{} {
      double f = 1.0;
      v[0] = 1.0;
      for (int n = 1; n <= 13; ++n) { f /= (double)n; v[n] = f; }
    }
  };
  static constexpr InvFact kF{};
#pragma unroll 1
  for (int n = 1; n <= 13; ++n) {
    // M <- M X + P Gs and P <- P X, row by row IN PLACE: row i of either result depends on row i
    // of M and P only, so a three-entry temporary replaces a full scratch matrix (six live
    // matrices instead of seven: what takes the reverse x-update kernel from one to two
    // wavefronts per SIMD)
    const double f = kF.v[n];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double tr[3], ti[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double sr = 0.0, si = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          sr = fma(M.re[3 * i + k], X.re[3 * k + j], sr); sr = fma(-M.im[3 * i + k], X.im[3 * k + j], sr);
          si = fma(M.re[3 * i + k], X.im[3 * k + j], si); si = fma(M.im[3 * i + k], X.re[3 * k + j], si);
          sr = fma(P.re[3 * i + k], Gs.re[3 * k + j], sr); sr = fma(-P.im[3 * i + k], Gs.im[3 * k + j], sr);
          si = fma(P.re[3 * i + k], Gs.im[3 * k + j], si); si = fma(P.im[3 * i + k], Gs.re[3 * k + j], si);
        }
        tr[j] = sr; ti[j] = si;
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        M.re[3 * i + j] = tr[j]; M.im[3 * i + j] = ti[j];
        L.re[3 * i + j] = fma(f, tr[j], L.re[3 * i + j]); L.im[3 * i + j] = fma(f, ti[j], L.im[3 * i + j]);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double sr = 0.0, si = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          sr = fma(P.re[3 * i + k], X.re[3 * k + j], sr); sr = fma(-P.im[3 * i + k], X.im[3 * k + j], sr);
          si = fma(P.re[3 * i + k], X.im[3 * k + j], si); si = fma(P.im[3 * i + k], X.re[3 * k + j], si);
        }
        tr[j] = sr; ti[j] = si;
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        P.re[3 * i + j] = tr[j]; P.im[3 * i + j] = ti[j];
        E.re[3 * i + j] = fma(f, tr[j], E.re[3 * i + j]); E.im[3 * i + j] = fma(f, ti[j], E.im[3 * i + j]);
      }
    }
  }
#pragma unroll 1
  for (int k = 0; k < s; ++k) {
    M3 t;
    m3_mul_nn(t, E, L);
    m3_mac_nn(t, L, E);
    L = t;
    m3_mul_nn(t, E, E);
    E = t;
  }
}

// ------------------------------------------------------------------ projectSU -> vec8 VJP
// J^H A J and V J for the complex Jacobi rotation in the (P, Q) plane:
//   J_PP = cs, J_PQ = sn, J_QP = -sn ph, J_QQ = cs ph   (ph = e^{-i arg h_PQ})
// Convergence threshold of a rotation, relative to the two diagonal entries.  2^-54 (a quarter ulp of the
// diagonal): what is left off the diagonal then perturbs M^H M below its own rounding, and the functions of H
// taken afterwards (1/h, 1/(h_i + h_j)) are smooth, so the result moves by ~1e-16 whatever the gaps are.  The
// round-2 value 1e-19 kept links that are already unitary rotating for all six sweeps: their M^H M is 1 + 1e-16
// noise with O(1) rotation angles inside the degenerate cluster, and the off-diagonal only falls linearly there.
// Measured (tools/projsu_bwd_time.py, 8^4 x 256 chains): 0.707 -> 0.685 ms for unitary links, unchanged for
// generic matrices (the slowest of a wavefront's 64 lanes sets the sweep count: five or six either way);
// without any rotation the kernel takes 0.465 ms, i.e. the sweeps are ~0.26 of its 0.73 ms.
#ifndef L2Q_JACOBI_TOL
#define L2Q_JACOBI_TOL 0x1p-54
#endif
// (returns whether this lane rotated at all)
template <int P, int Q>
__device__ __forceinline__ bool jacobi_rotate(M3& H, M3& Vm) {
  const double cr = H.re[3 * P + Q], ci = H.im[3 * P + Q];
  const double c_2 = cr * cr + ci * ci;
  const double irc = l2q_rsqrt(c_2 > 1e-290 ? c_2 : 1.0);            // 1 / |c|
  const double ac = c_2 > 1e-290 ? c_2 * irc : 0.0;
  const double a = H.re[3 * P + P], b = H.re[3 * Q + Q];
  // converged (relative to the diagonal) or so small that cr^2 + ci^2 is denormal and the
  // phase conj(c)/|c| would no longer have unit modulus: identity rotation (branch-free, the
  // lanes of a wavefront converge at different sweeps)
  const bool live = (ac > L2Q_JACOBI_TOL * (fabs(a) + fabs(b))) && (ac > 1e-140);
  const double acs = live ? ac : 1.0;
  // t = sgn(tau) / (|tau| + sqrt(1 + tau^2)) with tau = (b - a) / (2 |c|), written without forming tau,
  // and the reciprocal of |c| (from its reciprocal square root) for the phase
  const double d = b - a, c2 = 2.0 * acs;
  const double r2 = d * d + c2 * c2;                                  // > 0: c2 >= 2e-140 (or 2 when not live)
  const double t = (d >= 0.0 ? c2 : -c2) * l2q_rcp(fabs(d) + r2 * l2q_rsqrt(r2));
  const double cs = live ? l2q_rsqrt(1.0 + t * t) : 1.0;
  const double sn = live ? t * cs : 0.0;
  const double iac = live ? irc : 1.0;
  const double pr = live ? cr * iac : 1.0, pi = live ? -ci * iac : 0.0;   // ph = conj(c) / |c|
  // right-multiply by J: columns P, Q of H and of V
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    {
      const double xr = H.re[3 * r + P], xi = H.im[3 * r + P], yr = H.re[3 * r + Q], yi = H.im[3 * r + Q];
      const double zr = yr * pr - yi * pi, zi = yr * pi + yi * pr;     // y * ph
      H.re[3 * r + P] = cs * xr - sn * zr; H.im[3 * r + P] = cs * xi - sn * zi;
      H.re[3 * r + Q] = sn * xr + cs * zr; H.im[3 * r + Q] = sn * xi + cs * zi;
    }
    {
      const double xr = Vm.re[3 * r + P], xi = Vm.im[3 * r + P], yr = Vm.re[3 * r + Q], yi = Vm.im[3 * r + Q];
      const double zr = yr * pr - yi * pi, zi = yr * pi + yi * pr;
      Vm.re[3 * r + P] = cs * xr - sn * zr; Vm.im[3 * r + P] = cs * xi - sn * zi;
      Vm.re[3 * r + Q] = sn * xr + cs * zr; Vm.im[3 * r + Q] = sn * xi + cs * zi;
    }
  }
  // left-multiply by J^H: rows P, Q of H
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double xr = H.re[3 * P + k], xi = H.im[3 * P + k], yr = H.re[3 * Q + k], yi = H.im[3 * Q + k];
    const double zr = yr * pr + yi * pi, zi = -yr * pi + yi * pr;      // y * conj(ph)
    H.re[3 * P + k] = cs * xr - sn * zr; H.im[3 * P + k] = cs * xi - sn * zi;
    // row Q: sn * x + cs * conj(ph) * y
    H.re[3 * Q + k] = sn * xr + cs * zr; H.im[3 * Q + k] = sn * xi + cs * zi;
  }
  H.re[3 * P + Q] = 0.0; H.im[3 * P + Q] = 0.0;
  H.re[3 * Q + P] = 0.0; H.im[3 * Q + P] = 0.0;
  H.im[3 * P + P] = 0.0; H.im[3 * Q + Q] = 0.0;
  return live;
}

// y = su3_to_vec(projectSU(M)):  g_M += VJP(g_y).  projectSU(M) = U e^{i theta},
// U = M (M^H M)^{-1/2} (polar factor), theta = -arg(det U) / 3.
//   g_W from g_y (adjoint of the linear 8-component map), c = Re tr(g_W^H i W),
//   g_U = e^{-i theta} g_W - (c / 3) i U;   Z = U^H g_U;  H K + K H = Z with H = U^H M;
//   g_M = U (K - K^H).
// U and the Sylvester equation are computed in the eigenbasis of H (cyclic complex Jacobi, 6 sweeps):
// robust for the degenerate H ~ 1 of a link that is already unitary, where a polynomial-in-H
// solution (and the reference's closed-form eigenvalue derivative) breaks down.
__device__ __forceinline__ void m3_projsu_vec8_vjp(M3& g, const M3& m, const double (&gy)[8]) {
  // polar factor from the Jacobi eigen-decomposition of M^H M = V diag(w) V^H (the closed-form
  // eigenvalues of the forward kernel lose half the digits when the spectrum is degenerate,
  // which is the normal case for links that are already unitary)
  M3 a2, vm;
  m3_mul_an(a2, m, m);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = i; j < 3; ++j) {                      // Hermitian part
      const double ar = 0.5 * (a2.re[3 * i + j] + a2.re[3 * j + i]);
      const double ai = 0.5 * (a2.im[3 * i + j] - a2.im[3 * j + i]);
      a2.re[3 * i + j] = ar; a2.im[3 * i + j] = ai;
      a2.re[3 * j + i] = ar; a2.im[3 * j + i] = -ai;
    }
  m3_identity(vm);
#pragma unroll 1
  for (int sweep = 0; sweep < 6; ++sweep) {
    bool live = jacobi_rotate<0, 1>(a2, vm);
    live |= jacobi_rotate<0, 2>(a2, vm);
    live |= jacobi_rotate<1, 2>(a2, vm);
    // a sweep in which no lane of the wavefront rotated leaves every later sweep an identity as well
    // (same bits): links that are already unitary -- the x of every v-net call -- stop after one or two
    if (!l2q_wave_any(live)) break;
  }
  const double ev[3] = {sqrt(fabs(a2.re[0])), sqrt(fabs(a2.re[4])), sqrt(fabs(a2.re[8]))};   // eig(H)
  const double iev[3] = {1.0 / ev[0], 1.0 / ev[1], 1.0 / ev[2]};
  // Everything after the decomposition stays in the eigenbasis (round 4: six 3x3 products instead of nine):
  //   P = U V = M V diag(1/h);   theta from det M (det H > 0, so arg det U = arg det M);
  //   G' = g_W V;   c = Re tr(g_W^H i W) = -Im(phi tr(G'^H P));   Q = g_U V = conj(phi) G' - (c/3) i P;
  //   V^H (U^H g_U) V = P^H Q;   Kt_ij = (P^H Q)_ij / (h_i + h_j);   g_M = U (K - K^H) = P (Kt - Kt^H) V^H.
  M3 pm;
  m3_mul_nn(pm, m, vm);                                // M V
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) { pm.re[3 * i + j] *= iev[j]; pm.im[3 * i + j] *= iev[j]; }
  double dr, di;
  m3_det(dr, di, m);
  const double theta = -atan2(di, dr) / 3.0;
  const double pr = cos(theta), pi = sin(theta);
  // g_W: adjoint of m3_to_vec8
  M3 gw;
  m3_zero(gw);
  const double s3 = 0.57735026918962584;
  gw.re[1] = -2.0 * gy[1]; gw.im[1] = -2.0 * gy[0];
  gw.re[2] = -2.0 * gy[4]; gw.im[2] = -2.0 * gy[3];
  gw.re[5] = -2.0 * gy[6]; gw.im[5] = -2.0 * gy[5];
  gw.im[0] = -gy[2] - s3 * gy[7];
  gw.im[4] = gy[2] - s3 * gy[7];
  gw.im[8] = 2.0 * s3 * gy[7];
  M3 gv_;
  m3_mul_nn(gv_, gw, vm);                              // G' = g_W V
  double sr = 0.0, si = 0.0;                           // tr(G'^H P)
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    sr += gv_.re[i] * pm.re[i] + gv_.im[i] * pm.im[i];
    si += gv_.re[i] * pm.im[i] - gv_.im[i] * pm.re[i];
  }
  const double c3 = -(pr * si + pi * sr) / 3.0;        // c / 3
  M3 q;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    // conj(phi) G' - (c/3) i P
    q.re[i] = gv_.re[i] * pr + gv_.im[i] * pi + c3 * pm.im[i];
    q.im[i] = -gv_.re[i] * pi + gv_.im[i] * pr - c3 * pm.re[i];
  }
  M3 zt;
  m3_mul_an(zt, pm, q);                                // P^H Q
  const double i01 = 1.0 / (ev[0] + ev[1]), i02 = 1.0 / (ev[0] + ev[2]), i12 = 1.0 / (ev[1] + ev[2]);
  const double inv[9] = {0.5 * iev[0], i01, i02, i01, 0.5 * iev[1], i12, i02, i12, 0.5 * iev[2]};
#pragma unroll
  for (int i = 0; i < 9; ++i) { zt.re[i] *= inv[i]; zt.im[i] *= inv[i]; }
  M3 ka;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {                      // Kt - Kt^H
      ka.re[3 * i + j] = zt.re[3 * i + j] - zt.re[3 * j + i];
      ka.im[3 * i + j] = zt.im[3 * i + j] + zt.im[3 * j + i];
    }
  M3 t;
  m3_mul_nn(t, pm, ka);
  m3_mul_na(g, t, vm);                                 // P (Kt - Kt^H) V^H
}

}  // namespace l2q
