// heads_h_common.hpp -- shared by the half-precision heads + update kernels of the U(1) LeapfrogLayer
// (gemm_f16.hip: tile and stream kernels; heads_kstream_f16.hip: K-split stream kernel): arguments, the hardware
// transcendentals of the 16-bit epilogue, and the per-entry arithmetic (autocast's rounding points -> v- or x-update).
#pragma once
#include "half_common.hpp"
#include "u1_math.hpp"

namespace l2q {

__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
__device__ __forceinline__ float fast_tanh_h(float x) {
  return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f);      // saturates to +-1; v_rcp_f32 (1 ulp),
                                                                       // not the 10-instruction IEEE division
}

// atan2(y, x) for x >= 0 (the half-angle form of the x-update: x = cos(theta / 2) of a wrapped angle; a cosine that
// rounds to -1e-7 at theta = +-pi is handled by the same formula to first order).  atan(t) = t P(t^2) on [0, 1],
// P of degree 7 fitted at Chebyshev nodes: |error| < 1.5e-7 in fp32 Horner form (checked on 2e5 points); the
// libm atan2f is ~60 instructions with an IEEE division, and the epilogue of the x-update is VALU-bound.
__device__ __forceinline__ float fast_atan2_px(float y, float x) {
  const float ay = fabsf(y);
  const bool big = ay > x;                         // (compare + select: fmaxf / fminf canonicalise their operands first)
  const float mx = big ? ay : x, mn = big ? x : ay;
  const float t = mn * __builtin_amdgcn_rcpf(mx);
  const float u = t * t;
  float p = -0.00455979211255908f;
  p = fmaf(p, u, 0.023780519142746925f);
  p = fmaf(p, u, -0.05882975459098816f);
  p = fmaf(p, u, 0.09868865460157394f);
  p = fmaf(p, u, -0.14003290235996246f);
  p = fmaf(p, u, 0.19966961443424225f);
  p = fmaf(p, u, -0.3333181142807007f);
  p = fmaf(p, u, 0.9999998807907104f);
  float r = p * t;
  r = big ? 1.5707963267948966f - r : r;
  return copysignf(r, y);
}

// ((x + pi) mod 2 pi) - pi for |x| of a few pi (a wrapped angle plus one leapfrog increment) on v_fract_f32: the
// result is in [-pi, pi) by construction (fract < 1), no fix-up compares; differs from the exact fmodf loop of
// wrap_angle<float> (u1_math.hpp) by the rounding of one fp32 multiply (~5e-7).
__device__ __forceinline__ float fast_wrap_angle(float x) {
  const float pi = 3.14159265358979323846f, two_pi = 6.28318530717958647692f;
  const float y = (x + pi) * 0.15915494309189535f;
  return fmaf(two_pi, __builtin_amdgcn_fractf(y), -pi);
}

// timing-only builds of u1_heads_update_h_kernel (tools/ab_build.sh -DL2Q_HH_SKIP=n): 1 no K-loop,
// 2 no epilogue math, 4 no field traffic.  0 in the product.
#ifndef L2Q_HH_SKIP
#define L2Q_HH_SKIP 0
#endif

struct HeadsHArgs {
  const void* Z;          // [M][K]  16 bit
  const void* W[3];       // s, t, q weights [N][K]  16 bit
  const float* b[3];      // biases [N]
  const float* cs;        // nw.s * exp(coeff_s[n])
  const float* cq;
  float st;               // nw.t
  float eps;
  float* a;               // v (v-update) or x (x-update), [M][N], in place
  const float* bsrc;      // force (v-update) or v (x-update)
  const float* mask;      // x-update: [N] keep mask (complement flips it)
  int complement;
  double* logdet_part;    // [M][ncols_part]
  int M, N, K, ncols_part;
};

// One entry of the half-precision heads + update epilogue (the arithmetic both kernels below share):
// head pre-activations (fp32 accumulators) -> s, t, q with autocast's rounding points -> v- or x-update.
// Returns the new field value; `ldterm` is the entry's contribution to the chain's log-Jacobian.
template <typename HT, bool XUPD, bool FWD, bool NCP>
__device__ __forceinline__ float hh_element(float as, float at, float aq, float bs, float bt, float bq,
                                            float cs, float cq, float st, float eps, float a0, float b0,
                                            float keep, float& ldterm) {
  const float s = cs * rnd<HT>(fast_tanh_h(rnd<HT>(as + bs)));
  const float t = rnd<HT>(st * rnd<HT>(at + bt));
  const float q = cq * rnd<HT>(fast_tanh_h(rnd<HT>(aq + bq)));
  if (!XUPD) {
    const float lj = FWD ? (eps * s * 0.5f) : (-eps * s * 0.5f);
    ldterm = lj;
    const float es = fast_exp(lj), eq = fast_exp(eps * q);
    const float f = b0 * eq + t;
    return FWD ? (es * a0 - 0.5f * eps * f) : (es * (a0 + 0.5f * eps * f));
  }
  const float xj = a0, mb = 1.f - keep;
  const float sj = FWD ? eps * s : -eps * s;
  const float es = fast_exp(sj), eq = fast_exp(eps * q);
  const float tr = b0 * eq + t;
  float xp, l;
  if (NCP) {
    const float hx = xj * 0.5f;                    // |hx| <= pi/2 (x is wrapped)
    const float ch = __cosf(hx), sh = es * __sinf(hx);
    const float x1 = 2.f * fast_atan2_px(sh, ch);  // = 2 atan(tan(hx) es), no division
    xp = FWD ? (x1 + eps * tr) : (x1 - es * eps * tr);
    // log(es / (ch^2 + sh^2)); the argument is never denormal: the bare v_log_f32 (log2) without __logf's scaling
    l = sj - 0.6931471805599453f * __builtin_amdgcn_logf(ch * ch + sh * sh);
  } else {
    xp = FWD ? (xj * es + eps * tr) : (es * (xj - eps * tr));
    l = sj;
  }
  ldterm = mb * l;
  return fast_wrap_angle(keep * xj + mb * xp);
}

// ---- the same arithmetic on TWO entries at a time (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 issue two fp32
// operations per lane and cycle; the transcendentals, conversions and selects stay per entry).  Used by the stream
// kernel, whose x-update is bound by the VALU issue of its epilogue.  Not bit-identical to hh_element (the compiler
// contracts other mul / add pairs; exp(2 x) and exp(eps q) take their scale factors in one packed multiply).
typedef float v2f __attribute__((ext_vector_type(2)));
#define L2Q_V2(expr_x, expr_y) ((v2f){(expr_x), (expr_y)})
// both entries through one v_cvt_pk_{f16,bf16}_f32 (round to nearest even, like the scalar conversion)
template <typename HT> __device__ __forceinline__ v2f rnd2(v2f x) {
  typedef HT h2 __attribute__((ext_vector_type(2)));
  return __builtin_convertvector(__builtin_convertvector(x, h2), v2f);
}
__device__ __forceinline__ v2f exp2_2(v2f x) { return L2Q_V2(__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)); }
__device__ __forceinline__ v2f rcp_2(v2f x) { return L2Q_V2(__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)); }
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat2(float a) { return L2Q_V2(a, a); }
__device__ __forceinline__ v2f tanh2_h(v2f x) {                       // 1 - 2 / (exp(2 x) + 1)
  const v2f e = exp2_2(x * splat2(2.885390081777927f));                // 2 log2(e)
  return fma2(splat2(-2.f), rcp_2(e + splat2(1.f)), splat2(1.f));
}
__device__ __forceinline__ v2f atan2_px_2(v2f y, v2f x) {
  const v2f ay = L2Q_V2(fabsf(y.x), fabsf(y.y));
  // (compare + select instead of fmaxf / fminf: those canonicalise both operands first -- two more VALU instructions)
  const bool bx = ay.x > x.x, by = ay.y > x.y;
  const v2f mx = L2Q_V2(bx ? ay.x : x.x, by ? ay.y : x.y), mn = L2Q_V2(bx ? x.x : ay.x, by ? x.y : ay.y);
  const v2f t = mn * rcp_2(mx);
  const v2f u = t * t;
  v2f p = splat2(-0.00455979211255908f);
  p = fma2(p, u, splat2(0.023780519142746925f));
  p = fma2(p, u, splat2(-0.05882975459098816f));
  p = fma2(p, u, splat2(0.09868865460157394f));
  p = fma2(p, u, splat2(-0.14003290235996246f));
  p = fma2(p, u, splat2(0.19966961443424225f));
  p = fma2(p, u, splat2(-0.3333181142807007f));
  p = fma2(p, u, splat2(0.9999998807907104f));
  const v2f r = p * t, rc = splat2(1.5707963267948966f) - r;
  return L2Q_V2(copysignf(bx ? rc.x : r.x, y.x), copysignf(by ? rc.y : r.y, y.y));
}

template <typename HT, bool XUPD, bool FWD, bool NCP>
__device__ __forceinline__ v2f hh_element2(v2f as, v2f at, v2f aq, v2f bs, v2f bt, v2f bq, v2f cs, v2f cq, float st,
                                           float eps, v2f a0, v2f b0, v2f keep, v2f& ldterm) {
  const float l2e = 1.4426950408889634f;
  const v2f s = cs * rnd2<HT>(tanh2_h(rnd2<HT>(as + bs)));
  const v2f t = rnd2<HT>(splat2(st) * rnd2<HT>(at + bt));
  const v2f q = cq * rnd2<HT>(tanh2_h(rnd2<HT>(aq + bq)));
  const v2f eq = exp2_2(q * splat2(eps * l2e));
  if (!XUPD) {
    const v2f lj = s * splat2(FWD ? 0.5f * eps : -0.5f * eps);
    ldterm = lj;
    const v2f es = exp2_2(lj * splat2(l2e));
    const v2f f = fma2(b0, eq, t);
    return FWD ? fma2(es, a0, splat2(-0.5f * eps) * f) : es * fma2(splat2(0.5f * eps), f, a0);
  }
  const v2f xj = a0, mb = splat2(1.f) - keep;
  const v2f sj = s * splat2(FWD ? eps : -eps);
  const v2f es = exp2_2(sj * splat2(l2e));
  const v2f tr = fma2(b0, eq, t);
  v2f xp, l;
  if (NCP) {
    const v2f hx = xj * splat2(0.5f);
    const v2f ch = L2Q_V2(__cosf(hx.x), __cosf(hx.y));
    const v2f sh = es * L2Q_V2(__sinf(hx.x), __sinf(hx.y));
    const v2f x1 = splat2(2.f) * atan2_px_2(sh, ch);
    xp = FWD ? fma2(splat2(eps), tr, x1) : fma2(es * splat2(-eps), tr, x1);
    const v2f d = fma2(ch, ch, sh * sh);           // cos^2 + es^2 sin^2: never denormal -> the bare v_log_f32
    l = fma2(splat2(-0.6931471805599453f), L2Q_V2(__builtin_amdgcn_logf(d.x), __builtin_amdgcn_logf(d.y)), sj);
  } else {
    xp = FWD ? fma2(xj, es, splat2(eps) * tr) : es * fma2(splat2(-eps), tr, xj);
    l = sj;
  }
  ldterm = mb * l;
  const v2f w = fma2(keep, xj, mb * xp);
  const float pi = 3.14159265358979323846f, two_pi = 6.28318530717958647692f;
  const v2f y = (w + splat2(pi)) * splat2(0.15915494309189535f);
  return fma2(splat2(two_pi), L2Q_V2(__builtin_amdgcn_fractf(y.x), __builtin_amdgcn_fractf(y.y)), splat2(-pi));
}

}  // namespace l2q
