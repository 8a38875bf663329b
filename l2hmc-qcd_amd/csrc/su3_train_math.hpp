// su3_train_math.hpp -- per-link cotangent algebra of the SU(3) training kernels (device code;
// also compiled for the host by tests/native_host/ to check the formulas without a GPU).
#pragma once
#include "su3_math.hpp"

namespace l2q {

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

// E = exp(B) and L = L_exp(B)[G] (Frechet derivative of the exponential at B in direction G):
// scaling & squaring on the Taylor series, M_n = M_{n-1} X + X^{n-1} G,
// L_0 = sum M_n / n!, then L <- E L + L E, E <- E E per squaring.
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
template <int P, int Q>
__device__ __forceinline__ void jacobi_rotate(M3& H, M3& Vm) {
  const double cr = H.re[3 * P + Q], ci = H.im[3 * P + Q];
  const double ac = sqrt(cr * cr + ci * ci);
  const double a = H.re[3 * P + P], b = H.re[3 * Q + Q];
  // converged (relative to the diagonal) or so small that cr^2 + ci^2 is denormal and the
  // phase conj(c)/|c| would no longer have unit modulus: identity rotation (branch-free, the
  // lanes of a wavefront converge at different sweeps)
  const bool live = (ac > 1e-19 * (fabs(a) + fabs(b))) && (ac > 1e-140);
  const double acs = live ? ac : 1.0;
  const double tau = (b - a) / (2.0 * acs);
  const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
  const double cs = live ? 1.0 / sqrt(1.0 + t * t) : 1.0;
  const double sn = live ? t * cs : 0.0;
  const double pr = live ? cr / acs : 1.0, pi = live ? -ci / acs : 0.0;   // ph = conj(c) / |c|
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
    jacobi_rotate<0, 1>(a2, vm);
    jacobi_rotate<0, 2>(a2, vm);
    jacobi_rotate<1, 2>(a2, vm);
  }
  const double ev[3] = {sqrt(fabs(a2.re[0])), sqrt(fabs(a2.re[4])), sqrt(fabs(a2.re[8]))};   // eig(H)
  M3 t, u;
  m3_mul_nn(t, m, vm);                                 // M V
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) { t.re[3 * i + j] /= ev[j]; t.im[3 * i + j] /= ev[j]; }
  m3_mul_na(u, t, vm);                                 // U = M V diag(1/h) V^H
  double dr, di;
  m3_det(dr, di, u);
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
  // c = Re tr(g_W^H i W), W = u * phi
  double c = 0.0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const double wr = u.re[i] * pr - u.im[i] * pi, wi = u.re[i] * pi + u.im[i] * pr;
    c += gw.re[i] * (-wi) + gw.im[i] * wr;            // i W = (-wi, wr)
  }
  M3 gu;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    // conj(phi) g_W - (c/3) i U
    gu.re[i] = gw.re[i] * pr + gw.im[i] * pi + (c / 3.0) * u.im[i];
    gu.im[i] = -gw.re[i] * pi + gw.im[i] * pr - (c / 3.0) * u.re[i];
  }
  M3 z, zt;
  m3_mul_an(z, u, gu);                                 // Z = U^H g_U
  m3_mul_an(t, vm, z);                                 // V^H Z
  m3_mul_nn(zt, t, vm);                                // V^H Z V
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double inv = 1.0 / (ev[i] + ev[j]);
      zt.re[3 * i + j] *= inv; zt.im[3 * i + j] *= inv;
    }
  M3 k;
  m3_mul_nn(t, vm, zt);
  m3_mul_na(k, t, vm);                                 // K = V Kt V^H
  M3 ka;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {                      // K - K^H
      ka.re[3 * i + j] = k.re[3 * i + j] - k.re[3 * j + i];
      ka.im[3 * i + j] = k.im[3 * i + j] + k.im[3 * j + i];
    }
  m3_mul_nn(g, u, ka);
}

}  // namespace l2q
