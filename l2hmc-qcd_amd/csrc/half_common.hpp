// half_common.hpp -- shared by the half-precision layer kernels (gemm_f16.hip, conv_patch_f16.hip):
// MFMA wrappers for fp16 / bf16 operands, autocast rounding points of the epilogue.
#pragma once
#include "l2q_common.hpp"

namespace l2q {

typedef float v4f32 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int HBK = 64;            // K-slab: two MFMA K-steps of 32
constexpr int HLD = HBK + 8;       // row stride 144 B: conflict-free ds_read_b128 fragments

template <typename HT> struct MfmaH;
typedef float v16f32 __attribute__((ext_vector_type(16)));
template <> struct MfmaH<_Float16> {
  using vec_t = f16x8;
  static __device__ __forceinline__ v4f32 run(vec_t a, vec_t b, v4f32 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  // 32x32x16: A lane l holds row l & 31, k = 8 (l >> 5) .. +7; D col = l & 31,
  // row = (r & 3) + 8 (r >> 2) + 4 (l >> 5).  Half the LDS fragment reads per flop of 16x16x32.
  static __device__ __forceinline__ v16f32 run32(vec_t a, vec_t b, v16f32 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct MfmaH<__bf16> {
  using vec_t = bf16x8;
  static __device__ __forceinline__ v4f32 run(vec_t a, vec_t b, v4f32 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ v16f32 run32(vec_t a, vec_t b, v16f32 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

template <typename HT> __device__ __forceinline__ float rnd(float x) { return (float)(HT)x; }

__device__ __forceinline__ float act_h(float z, int act) {
  switch (act) {
    case L2Q_ACT_TANH: return tanhf(z);
    case L2Q_ACT_RELU: return z > 0.f ? z : 0.f;
    case L2Q_ACT_LEAKY_RELU: return z > 0.f ? z : 0.01f * z;
    case L2Q_ACT_ELU: return z > 0.f ? z : expm1f(z);
    case L2Q_ACT_SWISH: return z / (1.f + expf(-z));
    default: return z;
  }
}

struct EpiH {
  const float* bias;
  const float* bias2;
  const float* coeff;
  float scale;
  int act;
};

// y = scale * exp(coeff[n]) * r16(act(r16(acc + bias)))   (see the header of this file)
template <typename HT>
__device__ __forceinline__ float epilogue_h(float acc, float cb, float cs, bool has_coeff, int act) {
  float y = rnd<HT>(acc + cb);
  if (act != L2Q_ACT_NONE) y = rnd<HT>(act_h(y, act));
  y *= cs;
  return has_coeff ? y : rnd<HT>(y);
}

static inline bool al16(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// geometry of PeriodicPadding(k-1) -> Conv2d(k) as an implicit GEMM (gemm_f16.hip)
struct ConvGeomH {
  long sn, sc, sh, sw;
  int C, H, W, k, Ho, Wo, Kc;
  int clast;            // K order: 0 (ci, i, j) -- nn.Conv2d's flatten order, 1 (i, j, ci)
  long M;               // GEMM rows: output pixels, or (pool == 2) 4 x pooled pixels
  int pool = 1;         // 2: MaxPool2d(2) fused -- row m is window position m & 3 of pooled pixel m >> 2
  int Hp = 0, Wp = 0;   // pooled extent (Ho / 2, Wo / 2)
};

}  // namespace l2q
