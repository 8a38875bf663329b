// su3_math.hpp -- register-resident 3x3 complex fp64 algebra for gfx950 device code.
//
// Everything here is per-lane: one lane owns one 3x3 complex matrix (18 doubles = 36 VGPRs).
// Formulae follow the reference's PyTorch expressions so that results agree to rounding:
//   projectSU / eigs3x3 / rsqrtPHM3f : src/l2hmc/group/su3/pytorch/utils.py:227-346
//   projectTAH                        : src/l2hmc/group/su3/pytorch/group.py:92-103
//   su3_to_vec                        : src/l2hmc/group/su3/pytorch/utils.py:394-420
//   matrix_exp (torch library call)   : src/l2hmc/group/su3/pytorch/group.py:45-50
#pragma once
#include <hip/hip_runtime.h>

namespace l2q {

struct M3 {
  double re[9];
  double im[9];
};

__device__ __forceinline__ void m3_zero(M3& a) {
#pragma unroll
  for (int i = 0; i < 9; ++i) { a.re[i] = 0.0; a.im[i] = 0.0; }
}

__device__ __forceinline__ void m3_identity(M3& a) {
  m3_zero(a);
  a.re[0] = a.re[4] = a.re[8] = 1.0;
}

// C = A * B
__device__ __forceinline__ void m3_mul_nn(M3& c, const M3& a, const M3& b) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double sr = 0.0, si = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double ar = a.re[3 * i + k], ai = a.im[3 * i + k];
        const double br = b.re[3 * k + j], bi = b.im[3 * k + j];
        sr = fma(ar, br, sr); sr = fma(-ai, bi, sr);
        si = fma(ar, bi, si); si = fma(ai, br, si);
      }
      c.re[3 * i + j] = sr; c.im[3 * i + j] = si;
    }
}

// C = A * B^H
__device__ __forceinline__ void m3_mul_na(M3& c, const M3& a, const M3& b) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double sr = 0.0, si = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double ar = a.re[3 * i + k], ai = a.im[3 * i + k];
        const double br = b.re[3 * j + k], bi = -b.im[3 * j + k];
        sr = fma(ar, br, sr); sr = fma(-ai, bi, sr);
        si = fma(ar, bi, si); si = fma(ai, br, si);
      }
      c.re[3 * i + j] = sr; c.im[3 * i + j] = si;
    }
}

// C = A^H * B
__device__ __forceinline__ void m3_mul_an(M3& c, const M3& a, const M3& b) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double sr = 0.0, si = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double ar = a.re[3 * k + i], ai = -a.im[3 * k + i];
        const double br = b.re[3 * k + j], bi = b.im[3 * k + j];
        sr = fma(ar, br, sr); sr = fma(-ai, bi, sr);
        si = fma(ar, bi, si); si = fma(ai, br, si);
      }
      c.re[3 * i + j] = sr; c.im[3 * i + j] = si;
    }
}

// C = A^H * B^H
__device__ __forceinline__ void m3_mul_aa(M3& c, const M3& a, const M3& b) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double sr = 0.0, si = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double ar = a.re[3 * k + i], ai = -a.im[3 * k + i];
        const double br = b.re[3 * j + k], bi = -b.im[3 * j + k];
        sr = fma(ar, br, sr); sr = fma(-ai, bi, sr);
        si = fma(ar, bi, si); si = fma(ai, br, si);
      }
      c.re[3 * i + j] = sr; c.im[3 * i + j] = si;
    }
}

// acc += A * B^H
__device__ __forceinline__ void m3_mac_na(M3& c, const M3& a, const M3& b) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double sr = c.re[3 * i + j], si = c.im[3 * i + j];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double ar = a.re[3 * i + k], ai = a.im[3 * i + k];
        const double br = b.re[3 * j + k], bi = -b.im[3 * j + k];
        sr = fma(ar, br, sr); sr = fma(-ai, bi, sr);
        si = fma(ar, bi, si); si = fma(ai, br, si);
      }
      c.re[3 * i + j] = sr; c.im[3 * i + j] = si;
    }
}

// acc += A * B
__device__ __forceinline__ void m3_mac_nn(M3& c, const M3& a, const M3& b) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double sr = c.re[3 * i + j], si = c.im[3 * i + j];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double ar = a.re[3 * i + k], ai = a.im[3 * i + k];
        const double br = b.re[3 * k + j], bi = b.im[3 * k + j];
        sr = fma(ar, br, sr); sr = fma(-ai, bi, sr);
        si = fma(ar, bi, si); si = fma(ai, br, si);
      }
      c.re[3 * i + j] = sr; c.im[3 * i + j] = si;
    }
}

// tr( Y * (A * B)^H ) accumulated into (sr, si) without materialising A * B
__device__ __forceinline__ void m3_trace_y_abh(double& sr, double& si, const M3& y, const M3& a,
                                               const M3& b) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double pr = 0.0, pi = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double ar = a.re[3 * i + k], ai = a.im[3 * i + k];
        const double br = b.re[3 * k + j], bi = b.im[3 * k + j];
        pr = fma(ar, br, pr); pr = fma(-ai, bi, pr);
        pi = fma(ar, bi, pi); pi = fma(ai, br, pi);
      }
      const double yr = y.re[3 * i + j], yi = y.im[3 * i + j];
      sr = fma(yr, pr, sr); sr = fma(yi, pi, sr);
      si = fma(yi, pr, si); si = fma(-yr, pi, si);
    }
}

__device__ __forceinline__ void m3_add(M3& a, const M3& b) {
#pragma unroll
  for (int i = 0; i < 9; ++i) { a.re[i] += b.re[i]; a.im[i] += b.im[i]; }
}

// tr(A * B^H) = sum_ij A_ij conj(B_ij)
__device__ __forceinline__ void m3_trace_mul_na(double& tr, double& ti, const M3& a, const M3& b) {
  double sr = 0.0, si = 0.0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    sr = fma(a.re[i], b.re[i], sr); sr = fma(a.im[i], b.im[i], sr);
    si = fma(a.im[i], b.re[i], si); si = fma(-a.re[i], b.im[i], si);
  }
  tr = sr; ti = si;
}

__device__ __forceinline__ void cmul(double& cr, double& ci, double ar, double ai, double br,
                                     double bi) {
  cr = ar * br - ai * bi;
  ci = ar * bi + ai * br;
}

// det by cofactor expansion along the first row
__device__ __forceinline__ void m3_det(double& dr, double& di, const M3& m) {
  double t0r, t0i, t1r, t1i, ar, ai, br, bi;
  // m00 * (m11 m22 - m12 m21)
  cmul(ar, ai, m.re[4], m.im[4], m.re[8], m.im[8]);
  cmul(br, bi, m.re[5], m.im[5], m.re[7], m.im[7]);
  cmul(t0r, t0i, m.re[0], m.im[0], ar - br, ai - bi);
  // m01 * (m10 m22 - m12 m20)
  cmul(ar, ai, m.re[3], m.im[3], m.re[8], m.im[8]);
  cmul(br, bi, m.re[5], m.im[5], m.re[6], m.im[6]);
  cmul(t1r, t1i, m.re[1], m.im[1], ar - br, ai - bi);
  dr = t0r - t1r; di = t0i - t1i;
  // m02 * (m10 m21 - m11 m20)
  cmul(ar, ai, m.re[3], m.im[3], m.re[7], m.im[7]);
  cmul(br, bi, m.re[4], m.im[4], m.re[6], m.im[6]);
  cmul(t0r, t0i, m.re[2], m.im[2], ar - br, ai - bi);
  dr += t0r; di += t0i;
}

// R = (X - X^H)/2 - tr(X - X^H)/6 * I      (group.py:92-103)
__device__ __forceinline__ void m3_tah(M3& r, const M3& x) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      r.re[3 * i + j] = 0.5 * (x.re[3 * i + j] - x.re[3 * j + i]);
      r.im[3 * i + j] = 0.5 * (x.im[3 * i + j] + x.im[3 * j + i]);
    }
  const double dr = (r.re[0] + r.re[4] + r.re[8]) / 3.0;
  const double di = (r.im[0] + r.im[4] + r.im[8]) / 3.0;
  r.re[0] -= dr; r.re[4] -= dr; r.re[8] -= dr;
  r.im[0] -= di; r.im[4] -= di; r.im[8] -= di;
}

// exp(A) for a general complex 3x3 A.  Scaling & squaring; the scaled exponential is summed
// through the Cayley-Hamilton recursion A^(n+1) = p0 A^(n-2) - p1 A^(n-1) + p2 A^n carried
// on the three scalar coefficients of {I, A, A^2}, so it costs one matmul (A^2), 20 scalar
// recursion steps and s squarings.
__device__ __forceinline__ void m3_expm(M3& out, const M3& ain) {
  double n2 = 0.0;
#pragma unroll
  for (int i = 0; i < 9; ++i) n2 += ain.re[i] * ain.re[i] + ain.im[i] * ain.im[i];
  const double nrm = sqrt(n2);                       // Frobenius >= spectral norm
  int s = 0;
  double scale = 1.0;
  if (nrm > 0.5) {
    int ex;
    (void)frexp(nrm, &ex);                           // nrm = f * 2^ex, f in [0.5, 1)
    s = ex + 1;                                      // nrm / 2^s in [0.25, 0.5)
    if (s > 60) s = 60;
    scale = ldexp(1.0, -s);
  }
  M3 a;
#pragma unroll
  for (int i = 0; i < 9; ++i) { a.re[i] = ain.re[i] * scale; a.im[i] = ain.im[i] * scale; }
  M3 a2;
  m3_mul_nn(a2, a, a);
  // characteristic polynomial  A^3 = p2 A^2 - p1 A + p0 I
  const double p2r = a.re[0] + a.re[4] + a.re[8], p2i = a.im[0] + a.im[4] + a.im[8];
  const double t2r = a2.re[0] + a2.re[4] + a2.re[8], t2i = a2.im[0] + a2.im[4] + a2.im[8];
  double sqr, sqi;
  cmul(sqr, sqi, p2r, p2i, p2r, p2i);
  const double p1r = 0.5 * (sqr - t2r), p1i = 0.5 * (sqi - t2i);
  double p0r, p0i;
  m3_det(p0r, p0i, a);
  // coefficients of A^n / n!  on {I, A, A^2}
  double ar = 0.0, ai = 0.0, br = 0.0, bi = 0.0, cr = 0.5, ci = 0.0;   // n = 2: A^2/2
  double f0r = 1.0, f0i = 0.0, f1r = 1.0, f1i = 0.0, f2r = 0.5, f2i = 0.0;
  // 1 / (n + 1) from a table of correctly rounded constants (scalar loads): written as a division the rolled
  // loop carries an IEEE fp64 division sequence per step (13 of its ~45 instructions), and fully unrolled the
  // kernels that inline this go from 158 to 168 VGPRs and spill
#ifndef L2Q_EXPM_DIV
#define L2Q_EXPM_DIV 0
#endif
  static constexpr double kInv[22] = {
      1.0 / 1, 1.0 / 2, 1.0 / 3, 1.0 / 4, 1.0 / 5, 1.0 / 6, 1.0 / 7, 1.0 / 8, 1.0 / 9, 1.0 / 10, 1.0 / 11,
      1.0 / 12, 1.0 / 13, 1.0 / 14, 1.0 / 15, 1.0 / 16, 1.0 / 17, 1.0 / 18, 1.0 / 19, 1.0 / 20, 1.0 / 21, 1.0 / 22};
#pragma unroll 1
  for (int n = 2; n < 22; ++n) {
    const double inv = L2Q_EXPM_DIV ? 1.0 / (double)(n + 1) : kInv[n];
    double xr, xi, yr, yi, zr, zi;
    cmul(xr, xi, cr, ci, p0r, p0i);                  // c p0
    cmul(yr, yi, cr, ci, p1r, p1i);                  // c p1
    cmul(zr, zi, cr, ci, p2r, p2i);                  // c p2
    const double nar = xr * inv, nai = xi * inv;
    const double nbr = (ar - yr) * inv, nbi = (ai - yi) * inv;
    const double ncr = (br + zr) * inv, nci = (bi + zi) * inv;
    ar = nar; ai = nai; br = nbr; bi = nbi; cr = ncr; ci = nci;
    f0r += ar; f0i += ai; f1r += br; f1i += bi; f2r += cr; f2i += ci;
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    double xr, xi, yr, yi;
    cmul(xr, xi, f1r, f1i, a.re[i], a.im[i]);
    cmul(yr, yi, f2r, f2i, a2.re[i], a2.im[i]);
    out.re[i] = xr + yr; out.im[i] = xi + yi;
  }
  out.re[0] += f0r; out.re[4] += f0r; out.re[8] += f0r;
  out.im[0] += f0i; out.im[4] += f0i; out.im[8] += f0i;
#pragma unroll 1
  for (int k = 0; k < s; ++k) {
    M3 t;
    m3_mul_nn(t, out, out);
    out = t;
  }
}

// Closed-form eigenvalues of a 3x3 positive Hermitian matrix from (tr, tr(M^2), det).
// utils.py:227-283 (same clamps: +-3e38 on 1/q^1.5, acos argument to +-(1 - 1e-12)).
__device__ __forceinline__ void eigs3x3(double& e0, double& e1, double& e2, double tr, double p2,
                                        double det) {
  const double tr3 = (1.0 / 3.0) * tr;
  const double p23 = (1.0 / 3.0) * p2;
  const double tr32 = tr3 * tr3;
  const double q = fabs(0.5 * (p23 - tr32));
  const double r = 0.25 * tr3 * (5.0 * tr32 - p2) - 0.5 * det;
  const double sq = sqrt(q);
  const double sq3 = q * sq;
  double isq3 = 1.0 / sq3;
  isq3 = fmin(3e38, fmax(-3e38, isq3));
  double rsq3 = r * isq3;
  rsq3 = fmin(1.0, fmax(-1.0, rsq3));
  rsq3 = fmin(1.0 - 1e-12, fmax(-1.0 + 1e-12, rsq3));
  const double t = (1.0 / 3.0) * acos(rsq3);
  double st, ct;
  sincos(t, &st, &ct);
  const double sqc = sq * ct;
  const double sqs = 1.7320508075688772 * sq * st;
  const double ll = tr3 + sqc;
  e0 = tr3 - 2.0 * sqc;
  e1 = ll + sqs;
  e2 = ll - sqs;
}

// m = x (x^H x)^(-1/2)                                 utils.py:286-338
__device__ __forceinline__ void m3_project_u(M3& m, const M3& x) {
  M3 t, t2;
  m3_mul_an(t, x, x);
  m3_mul_nn(t2, t, t);
  const double tr = t.re[0] + t.re[4] + t.re[8];
  const double p2 = t2.re[0] + t2.re[4] + t2.re[8];
  double detr, deti;
  m3_det(detr, deti, t);
  double e0, e1, e2;
  eigs3x3(e0, e1, e2, tr, p2, detr);
  const double se0 = sqrt(fabs(e0)), se1 = sqrt(fabs(e1)), se2 = sqrt(fabs(e2));
  const double u = se0 + se1 + se2;
  const double w = se0 * se1 * se2;
  const double d = w * (se0 + se1) * (se0 + se2) * (se1 + se2);
  const double di = 1.0 / d;
  const double c0 = di * (w * u * u + e0 * se0 * (e1 + e2) + e1 * se1 * (e0 + e2)
                          + e2 * se2 * (e0 + e1));
  const double c1 = -(tr * u + w) * di;
  const double c2 = u * di;
  M3 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    r.re[i] = c1 * t.re[i] + c2 * t2.re[i];
    r.im[i] = c1 * t.im[i] + c2 * t2.im[i];
  }
  r.re[0] += c0; r.re[4] += c0; r.re[8] += c0;
  m3_mul_nn(m, x, r);
}

// out = projectU(x) * exp(-i arg(det)/3)                utils.py:341-346
__device__ __forceinline__ void m3_project_su(M3& out, const M3& x) {
  M3 m;
  m3_project_u(m, x);
  double detr, deti;
  m3_det(detr, deti, m);
  const double p = (-1.0 / 3.0) * atan2(deti, detr);
  double sp, cp;
  sincos(p, &sp, &cp);
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    out.re[i] = m.re[i] * cp - m.im[i] * sp;
    out.im[i] = m.re[i] * sp + m.im[i] * cp;
  }
}

// 8 real adjoint components (utils.py:394-420)
__device__ __forceinline__ void m3_to_vec8(double v[8], const M3& x) {
  v[0] = -2.0 * x.im[1];
  v[1] = -2.0 * x.re[1];
  v[2] = x.im[4] - x.im[0];
  v[3] = -2.0 * x.im[2];
  v[4] = -2.0 * x.re[2];
  v[5] = -2.0 * x.im[5];
  v[6] = -2.0 * x.re[5];
  v[7] = 0.57735026918962584 * (2.0 * x.im[8] - x.im[4] - x.im[0]);
}

}  // namespace l2q
