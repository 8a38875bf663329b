// u1_math.hpp -- scalar helpers shared by the U(1) forward and backward kernels.
#pragma once
#include "l2q_common.hpp"

namespace l2q {

template <typename T> struct Math;
template <> struct Math<float> {
  static __device__ __forceinline__ float sin(float x) { return sinf(x); }
  static __device__ __forceinline__ float cos(float x) { return cosf(x); }
  static __device__ __forceinline__ float tan(float x) { return tanf(x); }
  static __device__ __forceinline__ float atan(float x) { return atanf(x); }
  static __device__ __forceinline__ float exp(float x) { return expf(x); }
  static __device__ __forceinline__ float log(float x) { return logf(x); }
  static __device__ __forceinline__ float floor(float x) { return floorf(x); }
  static __device__ __forceinline__ float fmod(float x, float y) { return fmodf(x, y); }
};
template <> struct Math<double> {
  static __device__ __forceinline__ double sin(double x) { return ::sin(x); }
  static __device__ __forceinline__ double cos(double x) { return ::cos(x); }
  static __device__ __forceinline__ double tan(double x) { return ::tan(x); }
  static __device__ __forceinline__ double atan(double x) { return ::atan(x); }
  static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
  static __device__ __forceinline__ double log(double x) { return ::log(x); }
  static __device__ __forceinline__ double floor(double x) { return ::floor(x); }
  static __device__ __forceinline__ double fmod(double x, double y) { return ::fmod(x, y); }
};

// theta(t,x) = U0(t,x) + U1(t+1,x) - U0(t,x+1) - U1(t,x)      (lattice.py:154-159)
template <typename T>
__device__ __forceinline__ T plaq_angle(const T* __restrict__ xc, int t, int x, int Tn, int Xn) {
  const int V = Tn * Xn;
  const int tp = (t + 1 == Tn) ? 0 : t + 1;
  const int xp = (x + 1 == Xn) ? 0 : x + 1;
  return xc[t * Xn + x] + xc[V + tp * Xn + x] - xc[t * Xn + xp] - xc[V + t * Xn + x];
}

// python-style modulo wrap: ((x + pi) mod 2pi) - pi with the sign of the divisor
template <typename T>
__device__ __forceinline__ T wrap_angle(T x) {
  const T pi = (T)3.14159265358979323846, two_pi = (T)6.28318530717958647692;
  T r = Math<T>::fmod(x + pi, two_pi);
  if (r < (T)0) r += two_pi;
  return r - pi;
}

__device__ __forceinline__ float act_f32(float z, int act) {
  switch (act) {
    case L2Q_ACT_TANH: return tanhf(z);
    case L2Q_ACT_RELU: return fmaxf(z, 0.0f);
    case L2Q_ACT_LEAKY_RELU: return z > 0.0f ? z : 0.01f * z;
    case L2Q_ACT_ELU: return z > 0.0f ? z : expm1f(z);
    case L2Q_ACT_SWISH: return z / (1.0f + expf(-z));
    default: return z;
  }
}

#define L2Q_DISPATCH_T(elem_bytes, CALL)                                  \
  if ((elem_bytes) == 4) { using T = float; CALL; }                       \
  else if ((elem_bytes) == 8) { using T = double; CALL; }                 \
  else { set_error("%s: elem_bytes must be 4 or 8", __func__); return L2Q_EINVAL; }

}  // namespace l2q
