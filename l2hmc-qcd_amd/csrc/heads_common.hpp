// heads_common.hpp -- what the kernels of the fused (s, t, q) heads + momentum update share: the
// argument block and the fp64 transcendental chain of the epilogue (gemm.hip: fp64 MFMA kernels;
// heads_sliced.hip: the int8-sliced kernel).
#pragma once
#include "l2q_common.hpp"

namespace l2q {

// 1 / d for finite positive d: v_rcp_f64 seed (~2^-26) + two Newton steps (5 instructions instead
// of the ~12 of the IEEE division sequence); error ~1 ulp
__device__ __forceinline__ double rcp_nr(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(r, fma(-d, r, 1.0), r);
  r = fma(r, fma(-d, r, 1.0), r);
  return r;
}

__device__ __forceinline__ double fast_tanh(double x) {
  // tanh(x) = 1 - 2 / (exp(2x) + 1).  |x| is clamped to 20 (tanh(20) rounds to 1 in fp64) so that
  // exp stays finite for the Newton reciprocal; NaN is passed through.
  const double c = fmin(fmax(x, -20.0), 20.0);
  const double t = 1.0 - 2.0 * rcp_nr(exp(2.0 * c) + 1.0);
  return (x != x) ? x : t;
}

// exp(x) for the step-size-scaled arguments of the momentum update (|eps s / 2|, |eps q| ~ 1e-2):
// degree-11 Taylor polynomial for |x| < 1/8 (remainder < 2e-18), libm otherwise.
__device__ __forceinline__ double exp_small(double x) {
  if (fabs(x) < 0.125) {
    double r = 1.0 / 39916800.0;
    r = fma(r, x, 1.0 / 3628800.0);
    r = fma(r, x, 1.0 / 362880.0);
    r = fma(r, x, 1.0 / 40320.0);
    r = fma(r, x, 1.0 / 5040.0);
    r = fma(r, x, 1.0 / 720.0);
    r = fma(r, x, 1.0 / 120.0);
    r = fma(r, x, 1.0 / 24.0);
    r = fma(r, x, 1.0 / 6.0);
    r = fma(r, x, 0.5);
    r = fma(r, x, 1.0);
    return fma(r, x, 1.0);
  }
  return exp(x);
}
// Branch-free exp for the LDS-DMA heads kernel: x = k ln2 + r, |r| <= ln2/2, degree-12 Taylor
// polynomial (remainder 1.7e-16 relative), ldexp.  ~19 fp64 instructions and no divergent
// fall-back path: on gfx950 the fp64 VALU instructions of a wavefront are paid in full by the
// fp64 MFMA stream of the other wavefront on the SIMD (tools/microbench/mfma_valu_overlap.hip).
// (exp_bf_core: |x| <= 708 guaranteed by the caller; no clamp, NaN travels through the arithmetic)
__device__ __forceinline__ double exp_bf_core(double xc) {
  const double k = __builtin_rint(xc * 1.4426950408889634);
  double r = fma(-k, 6.93147180369123816490e-01, xc);
  r = fma(-k, 1.90821492927058770002e-10, r);
  double p = 1.0 / 479001600.0;
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return __builtin_amdgcn_ldexp(p, (int)k);
}
__device__ __forceinline__ double exp_bf(double x) {
  const double y = exp_bf_core(fmin(fmax(x, -708.0), 709.0));
  return (x != x) ? x : y;
}
__device__ __forceinline__ double tanh_bf(double x) {
  const double c = fmin(fmax(x, -20.0), 20.0);
  const double t = 1.0 - 2.0 * rcp_nr(exp_bf_core(2.0 * c) + 1.0);
  return (x != x) ? x : t;
}

struct HeadsArgs {
  const double* Z;        // [M][K]
  const double* W[3];     // s, t, q weights [N][K]
  const double* b[3];     // biases [N]
  const double* cs;       // per-column scale of s: nw.s * exp(coeff_s[n])   (may be null -> ss)
  const double* cq;       // per-column scale of q
  double ss, st, sq;      // scalar scales (used where the vector is null; st always)
  double eps;
  double eps2;            // second update of a pair (PAIR kernels)
  int fwd2, flip;         // its direction; v -> -v between the two updates
  double* v;              // [M][N] (x2 if complex): the updated momentum
  const double* vin;      // the momentum read (= v for the in-place update)
  const double* F;        // [M][N] (x2 if complex)
  double* logdet_part;    // [M][ncols_part]
  double* ld1_part;       // MID kernels: log-Jacobian of the first update alone, [M][ncols_part]
  double* ke_part;        // MID kernels: sum |v|^2 after the first update,        [M][ncols_part]
  int M, N, K, ncols_part;
  // TAPE kernels (heads_sliced.hip): the heads themselves, [M][N] each, for the reverse sweep of training
  double* tape_s = nullptr;
  double* tape_t = nullptr;
  double* tape_q = nullptr;
};

}  // namespace l2q
