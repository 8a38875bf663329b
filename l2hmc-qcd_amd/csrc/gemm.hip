// gemm.hip -- dense layers of the xnet / vnet leapfrog networks on the CDNA4 matrix cores.
//
//   C[M][N] = epi( A[M][K] . W[N][K]^T  (+ A2[M][K2] . W2[N][K2]^T) + bias (+ bias2) )
//
// Both operands are K-contiguous (nn.Linear weight layout), so one LDS-tiled "NT" kernel
// serves every layer.  fp64 uses v_mfma_f64_16x16x4_f64, fp32 v_mfma_f32_16x16x4_f32 (exact
// IEEE fma chains, no reduced-precision path).  256-thread workgroups = 2x2 wavefronts, each
// owning a (BM/2)x(BN/2) quadrant of 16x16 MFMA tiles; operands are staged through LDS in
// K-slabs of 16 with 16-byte global loads and a register prefetch of the next slab.
// The L2HMC layers are skinny (M = #chains <= a few hundred, K or N = 32V..36V ~ 1e5), so
// the input layer splits K across workgroups with a fixed-order second-stage reduce
// (deterministic, no atomics), and the three output heads (s, t, q) are computed together
// and consumed in registers by the generalised momentum update (fused_heads_vupdate):
// s, t, q are never written to HBM.
#include "l2q_common.hpp"
#include "heads_common.hpp"
#include <type_traits>

namespace l2q {

#ifndef L2Q_BK
#define L2Q_BK 16
#endif
#ifndef L2Q_GD_SPREAD
#define L2Q_GD_SPREAD 0      // 1: LDS-DMA pieces of gemm_dma_f64_kernel spread over the slab's MFMAs (input layer 0.652-0.655
                           // against 0.644-0.645 ms back to back: not the limiter; tools/time_gemm_in.py)
#endif
#ifndef L2Q_HEADS_OCC
#define L2Q_HEADS_OCC 2
#endif
constexpr int BK = L2Q_BK, LDP = BK + 2;   // +2: conflict-free ds_read_b64 / ds_read_b32 fragments

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef float v4f32 __attribute__((ext_vector_type(4)));

template <typename T> struct Mfma;
template <> struct Mfma<double> {
  using acc_t = v4f64;
  using vec_t = double2;
  static constexpr int VEC = 2;
  static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <> struct Mfma<float> {
  using acc_t = v4f32;
  using vec_t = float4;
  static constexpr int VEC = 4;
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = 4 * (lane >> 4) + reg
  static __device__ __forceinline__ int row(int lane, int r) { return 4 * (lane >> 4) + r; }
};

__device__ __forceinline__ float fast_tanh(float x) { return tanhf(x); }

template <typename T>
__device__ __forceinline__ T apply_act(T z, int act) {
  switch (act) {
    case L2Q_ACT_TANH: return fast_tanh(z);
    case L2Q_ACT_RELU: return z > (T)0 ? z : (T)0;
    case L2Q_ACT_LEAKY_RELU: return z > (T)0 ? z : (T)0.01 * z;
    case L2Q_ACT_ELU: return z > (T)0 ? z : expm1(z);
    case L2Q_ACT_SWISH: return z / ((T)1 + exp(-z));
    default: return z;
  }
}

// y = scale * exp(coeff[n]) * act(acc + bias[n] + bias2[n])
template <typename T>
struct Epilogue {
  const T* bias;
  const T* bias2;
  const T* coeff;
  T scale;
  int act;
  int accumulate;        // C += result (gradient accumulation) instead of C = result
  __device__ __forceinline__ T colscale(int n) const {
    return coeff ? scale * exp(coeff[n]) : scale;
  }
  __device__ __forceinline__ T colbias(int n) const {
    T b = (T)0;
    if (bias) b += bias[n];
    if (bias2) b += bias2[n];
    return b;
  }
};

// Operand tile loader: ROWS x BK elements of the virtual K-concatenated matrix [P | P2]
// (row stride K resp. K2), zero outside [kbeg, kend) and beyond nrows.  VECLOAD: 16-byte
// loads (needs K, K2, kbeg multiples of VEC and 16-byte aligned bases), else scalar.
template <typename T, int ROWS, bool VECLOAD>
struct TileLoader {
  using vec_t = typename Mfma<T>::vec_t;
  static constexpr int VEC = Mfma<T>::VEC;
  static constexpr int VPR = BK / VEC;                 // vectors per row
  static constexpr int RPP = kBlock / VPR;             // rows per pass
  static constexpr int NP = (ROWS + RPP - 1) / RPP;    // passes
  static constexpr int SRPP = kBlock / BK;             // scalar path: rows per pass
  static constexpr int SNP = (ROWS + SRPP - 1) / SRPP;
  T reg[VECLOAD ? NP * VEC : SNP];

  __device__ __forceinline__ void fetch(const T* __restrict__ p, const T* __restrict__ p2,
                                        long row0, long nrows, long k0, long K, long K2,
                                        long kend) {
    const int tid = threadIdx.x;
    if (VECLOAD) {
      const int kv = (tid % VPR) * VEC, r = tid / VPR;
      const long kk = k0 + kv;
      const T* base = p;
      long ld = K, kc = kk;
      if (kk >= K) { base = p2; ld = K2; kc = kk - K; }
      const bool kin = kk < kend;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const long row = row0 + r + (long)i * RPP;
        vec_t v;
        if (kin && row < nrows && (r + i * RPP) < ROWS) {
          v = *reinterpret_cast<const vec_t*>(base + row * ld + kc);
        } else {
          v = vec_t{};
        }
        const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) reg[i * VEC + j] = e[j];
      }
    } else {
      const int kq = tid % BK, r = tid / BK;
      const long kk = k0 + kq;
#pragma unroll
      for (int i = 0; i < SNP; ++i) {
        const long row = row0 + r + (long)i * SRPP;
        T v = (T)0;
        if (kk < kend && row < nrows && (r + i * SRPP) < ROWS) {
          if (kk < K) v = p[row * K + kk];
          else if (kk - K < K2) v = p2[row * K2 + (kk - K)];
        }
        reg[i] = v;
      }
    }
  }

  __device__ __forceinline__ void store(T (*lds)[LDP]) const {
    const int tid = threadIdx.x;
    if (VECLOAD) {
      const int kv = (tid % VPR) * VEC, r = tid / VPR;
#pragma unroll
      for (int i = 0; i < NP; ++i)
        if (r + i * RPP < ROWS) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) lds[r + i * RPP][kv + j] = reg[i * VEC + j];
        }
    } else {
      const int kq = tid % BK, r = tid / BK;
#pragma unroll
      for (int i = 0; i < SNP; ++i)
        if (r + i * SRPP < ROWS) lds[r + i * SRPP][kq] = reg[i];
    }
  }
};

// Same tile from a TRANSPOSED source: element (row r, column k) lives at p[k * ld + r] (the
// operand is stored [K][rows], rows contiguous) -- the backward GEMMs dW = dY^T X and
// dX = dY W read their operands this way, so no transposed copy is ever written to HBM.
// 16-byte loads run along the rows (needs ld % VEC == 0 and an aligned base), the LDS image is
// the same [ROWS][BK] as TileLoader's.
template <typename T, int ROWS>
struct TileLoaderT {
  using vec_t = typename Mfma<T>::vec_t;
  static constexpr int VEC = Mfma<T>::VEC;
  static constexpr int RV = ROWS / VEC;                // row vectors per k
  static constexpr int KPP = kBlock / RV;              // k per pass
  static constexpr int NP = BK / KPP;
  T reg[NP * VEC];

  __device__ __forceinline__ void fetch(const T* __restrict__ p, long row0, long nrows, long k0,
                                        long ld, long kend) {
    const int rv = threadIdx.x % RV, kk = threadIdx.x / RV;
    const long row = row0 + (long)rv * VEC;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const long k = k0 + kk + i * KPP;
      if (k < kend && row + VEC <= nrows) {
        const vec_t v = *reinterpret_cast<const vec_t*>(p + k * ld + row);
        const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) reg[i * VEC + j] = e[j];
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j)
          reg[i * VEC + j] = (k < kend && row + j < nrows) ? p[k * ld + row + j] : (T)0;
      }
    }
  }

  __device__ __forceinline__ void store(T (*lds)[LDP]) const {
    const int rv = threadIdx.x % RV, kk = threadIdx.x / RV;
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
      for (int j = 0; j < VEC; ++j) lds[rv * VEC + j][kk + i * KPP] = reg[i * VEC + j];
  }
};

// ---------------------------------------------------------------------------------------
// Generic layer: grid x = N tiles, y = M tiles, z = K splits.  FUSED: splits == 1 and the
// epilogue is applied here; otherwise raw partial sums go to part[z][M][N].
// TA / TW: the A / W operand is stored transposed ([K][M] resp. [K][N]; needs K2 == 0).
template <typename T, bool FUSED, bool VECLOAD, bool TA = false, bool TW = false>
__global__ __launch_bounds__(kBlock, 2) void gemm_nt_kernel(
    const T* __restrict__ A, const T* __restrict__ W, const T* __restrict__ A2,
    const T* __restrict__ W2, int M, int N, long K, long K2, long kchunk, Epilogue<T> epi,
    T* __restrict__ C, T* __restrict__ part) {
  constexpr int BM = 128, BN = 128;
  using acc_t = typename Mfma<T>::acc_t;
  __shared__ T As[BM][LDP];
  __shared__ T Ws[BN][LDP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const long m0 = (long)blockIdx.y * BM, n0 = (long)blockIdx.x * BN;
  const long Kt = K + K2;
  const long kbeg = (long)blockIdx.z * kchunk;
  long kend = kbeg + kchunk;
  if (kend > Kt) kend = Kt;

  acc_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (acc_t){0, 0, 0, 0};

  typename std::conditional<TA, TileLoaderT<T, BM>, TileLoader<T, BM, VECLOAD>>::type la;
  typename std::conditional<TW, TileLoaderT<T, BN>, TileLoader<T, BN, VECLOAD>>::type lw;
#define L2Q_FETCH_AW(K0)                                                   \
  do {                                                                     \
    if constexpr (TA) la.fetch(A, m0, M, (K0), (long)M, kend);            \
    else la.fetch(A, A2, m0, M, (K0), K, K2, kend);                        \
    if constexpr (TW) lw.fetch(W, n0, N, (K0), (long)N, kend);            \
    else lw.fetch(W, W2, n0, N, (K0), K, K2, kend);                        \
  } while (0)
  L2Q_FETCH_AW(kbeg);
  for (long k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();                                 // previous slab fully consumed
    la.store(As);
    lw.store(Ws);
    __syncthreads();
    if (k0 + BK < kend) {                            // overlap the next slab's latency
      L2Q_FETCH_AW(k0 + BK);
    }
#pragma unroll
    for (int ks = 0; ks < BK; ks += 4) {
      T fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = As[wm + 16 * i + (lane & 15)][ks + (lane >> 4)];
        fb[i] = Ws[wn + 16 * i + (lane & 15)][ks + (lane >> 4)];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = Mfma<T>::run(fa[i], fb[j], acc[i][j]);
    }
  }

  T* dst = FUSED ? C : part + (long)blockIdx.z * M * N;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long n = n0 + wn + 16 * j + (lane & 15);
    if (n >= N) continue;
    T cs = (T)1, cb = (T)0;
    if (FUSED) { cs = epi.colscale((int)n); cb = epi.colbias((int)n); }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long m = m0 + wm + 16 * i + Mfma<T>::row(lane, r);
        if (m < M) {
          const T v = acc[i][j][r];
          if (FUSED) {
            const T y = cs * apply_act<T>(v + cb, epi.act);
            dst[m * N + n] = epi.accumulate ? dst[m * N + n] + y : y;
          } else {
            dst[m * N + n] = v;
          }
        }
      }
  }
}

// ---------------------------------------------------------------------------------------
// Periodic conv layer as an implicit GEMM: the im2col matrix is never materialised.  Row m of
// the virtual A operand is output pixel (b, ho, wo), column kk = (ci*k + i)*k + j reads
// in[b, ci, (ho+i-(k-1)) mod H, (wo+j-(k-1)) mod W] through generic element strides (NCHW for
// the first layer, the previous layer's NHWC output afterwards).  The (b, ho, wo) decomposition
// of the tile's 128 rows is done once per workgroup (LDS), the (ci, i, j) decomposition of a
// thread's K column once per slab and divides by the compile-time kernel size only.
// Materialised, the five col matrices of the default U(1) conv stack are 2.1 GB per call at
// cfg-2 (written, then read again by the GEMM).
struct ConvGeom {
  long sn, sc, sh, sw;
  int C, H, W, k, Ho, Wo, Kc;
  int clast;            // column order of the virtual A / the weight rows: 0: (ci, i, j), 1: (i, j, ci)
  long M;
};

// BN = 32 / 64 / 128 output-channel tile: the conv layers have 8..128 channels, a fixed 128-wide
// tile would spend most MFMAs on padding.
// VEC4: NHWC input, (i, j, ci) K order, C % 4 == 0: a thread gathers four consecutive input
// channels of one tap with one 16-byte load (4x fewer gather instructions than element-wise).
template <int KS, int BN, bool VEC4>
__global__ __launch_bounds__(kBlock, 2) void conv_gemm_kernel(const float* __restrict__ in,
                                                              ConvGeom g,
                                                              const float* __restrict__ Wt, int N,
                                                              Epilogue<float> epi,
                                                              float* __restrict__ C) {
  using T = float;
  constexpr int BM = 128;
  constexpr int WN = BN >= 64 ? 2 : 1, WM = 4 / WN;       // wavefront grid over the tile
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16;
  using acc_t = Mfma<T>::acc_t;
  __shared__ T As[BM][LDP];
  __shared__ T Ws[BN][LDP];
  __shared__ long rbase[BM];
  __shared__ int rr0[BM], rc0[BM];
  const int k = KS > 0 ? KS : g.k;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave / WN) * TM, wn = (wave % WN) * TN;
  const long m0 = (long)blockIdx.y * BM, n0 = (long)blockIdx.x * BN;
  for (int r = tid; r < BM; r += kBlock) {
    const long m = m0 + r;
    long base = -1;
    int r0 = 0, c0 = 0;
    if (m < g.M) {
      // 32-bit index arithmetic (M < 2^31, checked at launch): a 64-bit division is ~100 instructions
      const unsigned mu = (unsigned)m;
      const unsigned t = mu / (unsigned)g.Wo;
      const int wo = (int)(mu - t * (unsigned)g.Wo);
      const unsigned bq = t / (unsigned)g.Ho;
      const int ho = (int)(t - bq * (unsigned)g.Ho);
      base = (long)bq * g.sn;
      r0 = (ho - (k - 1)) % g.H; if (r0 < 0) r0 += g.H;
      c0 = (wo - (k - 1)) % g.W; if (c0 < 0) c0 += g.W;
    }
    rbase[r] = base; rr0[r] = r0; rc0[r] = c0;
  }
  __syncthreads();

  acc_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (acc_t){0, 0, 0, 0};

  constexpr int AV = VEC4 ? 4 : 1;
  constexpr int SRPP = kBlock / (BK / AV), SNP = BM / SRPP;       // rows per pass, passes
  const int kq = (tid % (BK / AV)) * AV, rq = tid / (BK / AV);
  T areg[SNP][AV];
// gather this thread's K column(s) (kk = K0 + kq ..) for its SNP rows of the tile
#define L2Q_CONV_FETCH_A(K0)                                                            \
  do {                                                                                  \
    const unsigned kk_ = (unsigned)(K0) + (unsigned)kq;      /* 32-bit: Kc = C k k is small */ \
    const bool kin_ = kk_ < (unsigned)g.Kc;                                             \
    int j_, i_, ci_;                                                                    \
    if (g.clast) { const unsigned ij_ = kk_ / (unsigned)g.C; ci_ = (int)(kk_ - ij_ * (unsigned)g.C); \
                   i_ = (int)(ij_ / (unsigned)k); j_ = (int)(ij_ - (unsigned)i_ * (unsigned)k); }    \
    else { const unsigned ij_ = kk_ / (unsigned)k; j_ = (int)(kk_ - ij_ * (unsigned)k);               \
           ci_ = (int)(ij_ / (unsigned)k); i_ = (int)(ij_ - (unsigned)ci_ * (unsigned)k); }           \
    const long coff_ = (long)ci_ * g.sc;                                                \
    _Pragma("unroll") for (int p = 0; p < SNP; ++p) {                                   \
      const int row_ = rq + p * SRPP;                                                   \
      const long base_ = rbase[row_];                                                   \
      _Pragma("unroll") for (int e = 0; e < AV; ++e) areg[p][e] = (T)0;                 \
      if (kin_ && base_ >= 0) {                                                         \
        int r_ = rr0[row_] + i_; if (r_ >= g.H) r_ -= g.H; if (r_ >= g.H) r_ %= g.H;    \
        int c_ = rc0[row_] + j_; if (c_ >= g.W) c_ -= g.W; if (c_ >= g.W) c_ %= g.W;    \
        const T* src_ = in + base_ + coff_ + r_ * g.sh + c_ * g.sw;                     \
        if (VEC4) {                                                                     \
          const float4 v_ = *reinterpret_cast<const float4*>(src_);                     \
          areg[p][0] = v_.x; areg[p][AV > 1 ? 1 : 0] = v_.y;                            \
          areg[p][AV > 2 ? 2 : 0] = v_.z; areg[p][AV > 3 ? 3 : 0] = v_.w;               \
        } else {                                                                        \
          areg[p][0] = src_[0];                                                         \
        }                                                                               \
      }                                                                                 \
    }                                                                                   \
  } while (0)
  TileLoader<T, BN, false> lw;
  L2Q_CONV_FETCH_A(0);
  lw.fetch(Wt, Wt, n0, N, 0, g.Kc, 0, g.Kc);      // (p2 unused: K2 = 0; a literal nullptr crashes hipcc 7.2 at -O2+)
  for (long k0 = 0; k0 < g.Kc; k0 += BK) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < SNP; ++p)
#pragma unroll
      for (int e = 0; e < AV; ++e) As[rq + p * SRPP][kq + e] = areg[p][e];
    lw.store(Ws);
    __syncthreads();
    if (k0 + BK < g.Kc) {
      L2Q_CONV_FETCH_A(k0 + BK);
      lw.fetch(Wt, Wt, n0, N, k0 + BK, g.Kc, 0, g.Kc);
    }
#pragma unroll
    for (int ks = 0; ks < BK; ks += 4) {
      T fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = As[wm + 16 * i + (lane & 15)][ks + (lane >> 4)];
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[j] = Ws[wn + 16 * j + (lane & 15)][ks + (lane >> 4)];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = Mfma<T>::run(fa[i], fb[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const long n = n0 + wn + 16 * j + (lane & 15);
    if (n >= N) continue;
    const T cs = epi.colscale((int)n), cb = epi.colbias((int)n);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long m = m0 + wm + 16 * i + Mfma<T>::row(lane, r);
        if (m < g.M) C[m * N + n] = cs * apply_act<T>(acc[i][j][r] + cb, epi.act);
      }
  }
#undef L2Q_CONV_FETCH_A
}

// Fixed-order reduction of the split-K partials + epilogue.  A workgroup owns 64 consecutive
// outputs; its four wavefronts each sum a contiguous quarter of the splits (eight loads in flight
// per lane), the quarters meet in LDS and are added in order 0..3 -- the same order every run.
// (One thread per output summing all 128 partials of the SU(3) input layer one after the other was
// a 128-deep chain of dependent HBM round trips: 43 us per call.)
template <typename T>
__global__ __launch_bounds__(kBlock) void splitk_reduce_kernel(const T* __restrict__ part,
                                                               int splits, long MN, int N,
                                                               Epilogue<T> epi, T* __restrict__ C) {
  __shared__ T quarter[4][64];
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + o;
  const int per = (splits + 3) / 4;
  const int z0 = g * per, z1 = min(splits, z0 + per);
  T s = (T)0;
  if (i < MN) {
    int z = z0;
    for (; z + 8 <= z1; z += 8) {
      T v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(long)(z + u) * MN + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; z < z1; ++z) s += part[(long)z * MN + i];
  }
  quarter[g][o] = s;
  __syncthreads();
  if (g != 0 || i >= MN) return;
  s = ((quarter[0][o] + quarter[1][o]) + quarter[2][o]) + quarter[3][o];
  const int n = (int)(i % N);
  const T y = epi.colscale(n) * apply_act<T>(s + epi.colbias(n), epi.act);
  C[i] = epi.accumulate ? C[i] + y : y;
}

// ---------------------------------------------------------------------------------------
// Fused output heads + generalised momentum update (dynamics.py:1266-1297 with
// network.py:547-551):  for every chain m and entry n
//     s = cs[n] tanh(z.Ws[n] + bs[n]),  t = ct (z.Wt[n] + bt[n]),  q = cq[n] tanh(z.Wq[n] + bq[n])
//     forward : v' = exp(eps s/2) v - eps/2 (F exp(eps q) + t)
//     backward: v' = exp(-eps s/2) (v + eps/2 (F exp(eps q) + t))
//     logdet_part[m][.] = +- sum_n eps s / 2
// Tile 64 (chains) x 64 (entries), each wavefront 32 x 32 = 2x2 MFMA tiles for each of the
// three heads (48 fp64 accumulators / lane, no spills at 2 waves/SIMD).  The three W tiles share the staged Z tile.
// CPLX: v, F are complex (SU(3)); the real heads act on both parts, t on the real part.
constexpr int kHeadsBN = 64;


// PAIR: the closing v-update of one leapfrog step and the opening v-update of the next act on
// the same x, hence the same (s, t, q): both are applied here from one evaluation of the heads
// (optionally with the momentum flip of the merged trajectory in between).
template <bool CPLX, bool FWD, bool PAIR>
__global__ __launch_bounds__(kBlock, L2Q_HEADS_OCC) void fused_heads_vupdate_kernel(HeadsArgs a, int swz, int stagger) {
  constexpr int BM = 64, BN = kHeadsBN, NJ = BN / 32;
  using T = double;
  using acc_t = v4f64;
  __shared__ T Zs[BM][LDP];
  __shared__ T Ws[3][BN][LDP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * (BN / 2);
  // logical block order: m fastest, so the M/64 blocks that share a W tile are adjacent and
  // (after the XCD swizzle) on the same XCD's L2
  const long mt = (a.M + BM - 1) / BM;
  // The two workgroups co-resident on a CU start together and would otherwise run their
  // MFMA-free epilogues at the same time; delaying the second resident set once keeps one
  // block's epilogue under the other's MFMA main loop for the rest of the launch.
  if (stagger > 0 && blockIdx.x >= 256 && blockIdx.x < 512) {
    for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }
  const long w = xcd_swizzle(blockIdx.x, gridDim.x, swz);
  const long m0 = (w % mt) * BM, n0 = (w / mt) * BN;

  acc_t acc[3][2][NJ];
#pragma unroll
  for (int h = 0; h < 3; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[h][i][j] = (acc_t){0, 0, 0, 0};

  TileLoader<T, BM, true> lz;
  TileLoader<T, BN, true> lw0, lw1, lw2;
  const long K = a.K;
  lz.fetch(a.Z, nullptr, m0, a.M, 0, K, 0, K);
  lw0.fetch(a.W[0], nullptr, n0, a.N, 0, K, 0, K);
  lw1.fetch(a.W[1], nullptr, n0, a.N, 0, K, 0, K);
  lw2.fetch(a.W[2], nullptr, n0, a.N, 0, K, 0, K);
  for (long k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();
    lz.store(Zs);
    lw0.store(Ws[0]);
    lw1.store(Ws[1]);
    lw2.store(Ws[2]);
    __syncthreads();
    if (k0 + BK < K) {
      lz.fetch(a.Z, nullptr, m0, a.M, k0 + BK, K, 0, K);
      lw0.fetch(a.W[0], nullptr, n0, a.N, k0 + BK, K, 0, K);
      lw1.fetch(a.W[1], nullptr, n0, a.N, k0 + BK, K, 0, K);
      lw2.fetch(a.W[2], nullptr, n0, a.N, k0 + BK, K, 0, K);
    }
#pragma unroll
    for (int ks = 0; ks < BK; ks += 4) {
      T fa[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = Zs[wm + 16 * i + (lane & 15)][ks + (lane >> 4)];
#pragma unroll
      for (int h = 0; h < 3; ++h) {
        T fb[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[j] = Ws[h][wn + 16 * j + (lane & 15)][ks + (lane >> 4)];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[h][i][j] = Mfma<T>::run(fa[i], fb[j], acc[h][i][j]);
      }
    }
  }

  // ---- epilogue: heads -> momentum update in registers
  const double eps = a.eps, heps = 0.5 * a.eps;
  double ld[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) ld[i][r] = 0.0;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const long n = n0 + wn + 16 * j + (lane & 15);
    if (n >= a.N) continue;
    const double bs = a.b[0][n], bt = a.b[1][n], bq = a.b[2][n];
    const double cs = a.cs ? a.cs[n] : a.ss;
    const double cq = a.cq ? a.cq[n] : a.sq;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long m = m0 + wm + 16 * i + Mfma<T>::row(lane, r);
        if (m >= a.M) continue;
        const double s = cs * fast_tanh(acc[0][i][j][r] + bs);
        const double t = a.st * (acc[1][i][j][r] + bt);
        const double q = cq * fast_tanh(acc[2][i][j][r] + bq);
        const double lj = FWD ? heps * s : -heps * s;
        ld[i][r] += lj;
        const double es = exp_small(lj), eq = exp_small(eps * q);
        const long o = m * (long)a.N + n;
        double vr, vi = 0.0, fr0, fi0 = 0.0;
        if (CPLX) {
          const double2 vv = reinterpret_cast<const double2*>(a.vin)[o];
          const double2 ff = reinterpret_cast<const double2*>(a.F)[o];
          vr = vv.x; vi = vv.y; fr0 = ff.x; fi0 = ff.y;
        } else {
          vr = a.vin[o]; fr0 = a.F[o];
        }
        {
          const double fr = fr0 * eq + t, fi = fi0 * eq;
          if (FWD) { vr = es * vr - heps * fr; vi = es * vi - heps * fi; }
          else { vr = es * (vr + heps * fr); vi = es * (vi + heps * fi); }
        }
        if (PAIR) {
          if (a.flip) { vr = -vr; vi = -vi; }
          const double h2 = 0.5 * a.eps2;
          const double lj2 = a.fwd2 ? h2 * s : -h2 * s;
          ld[i][r] += lj2;
          const double es2 = exp_small(lj2), eq2 = exp_small(a.eps2 * q);
          const double fr = fr0 * eq2 + t, fi = fi0 * eq2;
          if (a.fwd2) { vr = es2 * vr - h2 * fr; vi = es2 * vi - h2 * fi; }
          else { vr = es2 * (vr + h2 * fr); vi = es2 * (vi + h2 * fi); }
        }
        if (CPLX) reinterpret_cast<double2*>(a.v)[o] = make_double2(vr, vi);
        else a.v[o] = vr;
      }
  }
  // row partial of logdet over this wave's columns: butterfly over the 16 column lanes
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double x = ld[i][r];
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
      const long m = m0 + wm + 16 * i + Mfma<T>::row(lane, r);
      if ((lane & 15) == 0 && m < a.M) {
        const long col = (n0 / BN) * 2 + (wave & 1);
        a.logdet_part[m * a.ncols_part + col] = x;
      }
    }
}

// ---------------------------------------------------------------------------------------
// The same tile (64 chains x 64 entries x 3 heads, 2 x 2 wavefronts of 32 x 32) with the operand
// staging rebuilt for gfx950:
//  * Z and the three W tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging
//    VGPRs, no ds_write pass, two 32 KB stages, ONE barrier per K-slab;
//  * tile rows are 128 bytes, XOR-swizzled in 16-byte chunks (chunk ^ (row & 7)): LDS-DMA writes
//    lane-linear, so the swizzle is applied to each lane's SOURCE address and again to the
//    fragment reads -- ds_read_b64 fragments hit every bank pair exactly four times (the minimum);
//  * no address arithmetic in the loop (per-lane 32-bit offsets fixed up front, the K offset is
//    scalar): the shipped loop's ~100 VALU instructions per slab (64-bit row * K products and
//    bounds predicates per load) are gone.  That matters more than usual on this part: a
//    wavefront's fp64 MFMAs and ANY VALU instruction of the other wavefront on the SIMD do not
//    overlap (tools/microbench/mfma_valu_overlap.hip: fp64 / fp32 / int VALU time adds to the
//    MFMA time), so every VALU instruction is paid in MFMA time;
//  * epilogue operands (v, F) are requested in batches of four, one batch ahead, and the
//    transcendental chain is branch-free (exp_bf / tanh_bf).
// cfg-4 (M 256, K 256, N 147456): K-loop alone 0.85 ms = 68 TFLOP/s (0.99 ms before), whole kernel
// 1.14-1.16 ms (1.32-1.47 ms before); tools/lab/heads_lab.hip holds the A/B harness.
// Needs K % 16 == 0 (heads_launch falls back to fused_heads_vupdate_kernel otherwise).
typedef __attribute__((address_space(3))) void* lds_ptr_t;
// 16-byte chunk swizzle of the 128-byte LDS rows: position = chunk ^ L2Q_SWZ(row).  A ds_read_b64
// fragment read takes 32 lanes per LDS cycle (rows 0..15 x the two 8-byte halves of one chunk);
// the bank group of a row is 32 (row & 1) + 4 position, so the swizzle must separate the eight
// rows of equal parity: (row >> 1) & 7.  (row & 7, the first version, put rows r and r + 8 on
// the same banks: SQ_LDS_BANK_CONFLICT = 47 % of SQ_LDS_IDX_ACTIVE, tools/microbench/
// lds_b64_conflict.hip modes 2 and 5.)
#ifndef L2Q_SWZ
#define L2Q_SWZ(r) (((r) >> 1) & 7)
#endif
// Qualifier of the fragment reads of the LDS-DMA kernels.  Plain loads are merged by the compiler into
// ds_read2st64_b64, whose lane groups and 32-bank model differ from ds_read_b64's (MI355X_MICROARCH.md, LDS
// table): rows r and r ^ 1 of the swizzle above then share a bank -- PMC: 47-50 % of the LDS-active cycles of
// these kernels are bank conflicts.  `volatile` keeps them single ds_read_b64 (conflict-free, 2 cycles each).
// Measured (tools/gpu_job_lds.sh, same box, two rounds): input layer 0.649 -> 0.643 ms, heads pair 1.335 -> 1.318 ms:
// the LDS was not what these kernels wait for, the conflict-free form is simply never slower.
#ifndef L2Q_LDS_Q
#define L2Q_LDS_Q volatile
#endif
template <typename T>
__device__ __forceinline__ T lds_frag(const char* p) {
  return *(const L2Q_LDS_Q __attribute__((address_space(3))) T*)(const __attribute__((address_space(3))) char*)p;
}

// MID (pair kernels): per-step metrics need the state BETWEEN the two updates -- the kernel also
// returns the first update's log-Jacobian and sum |v|^2 of the intermediate momentum (per-row
// partials, reduced batch by batch so that the extra accumulators do not cost registers), which
// is all `Dynamics` needs of it (kinetic energy); verbose=True then runs 9 instead of 16 heads
// kernels per merged nleapfrog = 4 trajectory like verbose=False.
template <bool CPLX, bool FWD, bool PAIR, bool MID = false>
__global__ __launch_bounds__(kBlock, 2) void fused_heads_dma_kernel(HeadsArgs a, int swz) {
  constexpr int BM = 64, BN = 64, NJ = 2;
  constexpr int ROWB = BK * 8;                    // 128 bytes per tile row per slab
  constexpr int STAGE = (BM + 3 * BN) * ROWB;     // 32 KB
  using acc_t = v4f64;
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const long mt = (a.M + BM - 1) / BM;
  const long w = xcd_swizzle(blockIdx.x, gridDim.x, swz);
  const long m0 = (w % mt) * BM, n0 = (w / mt) * BN;
  const long K = a.K;

  // ---- loader: instruction q of this wavefront fills tile rows (4q + wave) * 8 + (lane >> 3),
  // 16-byte position lane & 7 of the row, with source chunk (lane & 7) ^ (row & 7)
  const char* ubase[4];
  ubase[0] = reinterpret_cast<const char*>(a.Z) + m0 * K * 8;
#pragma unroll
  for (int h = 0; h < 3; ++h) ubase[h + 1] = reinterpret_cast<const char*>(a.W[h]) + n0 * K * 8;
  unsigned voff[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int R = (4 * q + wave) * 8 + (lane >> 3);       // tile row 0..255
    const int r = R & 63;                                  // row within the operand
    long lim = (q < 2 ? (long)a.M - m0 : (long)a.N - n0) - 1;
    const int rc = r <= lim ? r : (int)lim;                // clamp: edge tiles re-read a valid row
    const int c = (lane & 7) ^ L2Q_SWZ(R);
    voff[q] = (unsigned)(rc * (int)K * 8 + c * 16);
  }
  auto issue = [&](int stage, long k0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const char* g = ubase[q >> 1] + k0 * 8 + (unsigned long)voff[q];
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (lds_ptr_t)(lds + stage * STAGE + (4 * q + wave) * 1024), 16, 0, 0);
    }
  };

  // ---- fragment read offsets (bytes within a stage), one per k-quad
  unsigned offA[4], offB[4];
#pragma unroll
  for (int kq = 0; kq < 4; ++kq) {
    const unsigned sw = ((((kq * 2) + (lane >> 5)) ^ L2Q_SWZ(lane)) << 4) + ((lane >> 4) & 1) * 8;
    offA[kq] = (wm + (lane & 15)) * ROWB + sw;
    offB[kq] = (BM + wn + (lane & 15)) * ROWB + sw;
  }

  acc_t acc[3][2][NJ];
#pragma unroll
  for (int h = 0; h < 3; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[h][i][j] = (acc_t){0, 0, 0, 0};

  issue(0, 0);
  const int nslab = (int)(K / BK);
  for (int s = 0; s < nslab; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < nslab) issue((s + 1) & 1, (long)(s + 1) * BK);
    const char* sb = lds + (s & 1) * STAGE;
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      double fa[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = lds_frag<double>(sb + offA[kq] + i * 16 * ROWB);
#pragma unroll
      for (int h = 0; h < 3; ++h) {
        double fb[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          fb[j] = lds_frag<double>(sb + offB[kq] + (h * BN + j * 16) * ROWB);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[h][i][j] = Mfma<double>::run(fa[i], fb[j], acc[h][i][j]);
      }
    }
  }
  // ---- epilogue: batches b = (j, i), 4 elements (r) each, operands one batch ahead
  const double eps = a.eps, heps = 0.5 * a.eps;
  double ld[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) ld[i][r] = 0.0;
  constexpr int NB = 2 * NJ;
  double2 vv[2][4], ff[2][4];
  const long mrow = m0 + wm + (lane >> 4);
  const long ncol = n0 + wn + (lane & 15);
  auto elem = [&](int b, int r, long& o, bool& ok) {
    const int j = b >> 1, i = b & 1;
    const long m = mrow + 16 * i + 4 * r, n = ncol + 16 * j;
    ok = (m < a.M) && (n < a.N);
    o = ok ? m * (long)a.N + n : 0;
  };
  auto fetch = [&](int b, int slot) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      long o; bool ok;
      elem(b, r, o, ok);
      if (CPLX) {
        vv[slot][r] = reinterpret_cast<const double2*>(a.vin)[o];
        ff[slot][r] = reinterpret_cast<const double2*>(a.F)[o];
      } else {
        vv[slot][r] = make_double2(a.vin[o], 0.0);
        ff[slot][r] = make_double2(a.F[o], 0.0);
      }
    }
  };
  double cb[NJ][5];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const long n = ncol + 16 * j;
    const long nc = n < a.N ? n : 0;
    cb[j][0] = a.b[0][nc]; cb[j][1] = a.b[1][nc]; cb[j][2] = a.b[2][nc];
    cb[j][3] = a.cs ? a.cs[nc] : a.ss;
    cb[j][4] = a.cq ? a.cq[nc] : a.sq;
  }
  fetch(0, 0);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int slot = b & 1, j = b >> 1, i = b & 1;
    if (b + 1 < NB) fetch(b + 1, slot ^ 1);
    const double bs = cb[j][0], bt = cb[j][1], bq = cb[j][2], cs = cb[j][3], cq = cb[j][4];
    double m1[4], m2[4], mk[4];                      // MID: this batch's per-row terms
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      long o; bool ok;
      elem(b, r, o, ok);
      const double s = cs * tanh_bf(acc[0][i][j][r] + bs);
      const double t = a.st * (acc[1][i][j][r] + bt);
      const double q = cq * tanh_bf(acc[2][i][j][r] + bq);
      const double lj = FWD ? heps * s : -heps * s;
      if (!MID) { if (ok) ld[i][r] += lj; }
      const double es = exp_bf(lj), eq = exp_bf(eps * q);
      double vr = vv[slot][r].x, vi = vv[slot][r].y;
      const double fr0 = ff[slot][r].x, fi0 = ff[slot][r].y;
      {
        const double fr = fr0 * eq + t, fi = fi0 * eq;
        if (FWD) { vr = es * vr - heps * fr; vi = es * vi - heps * fi; }
        else { vr = es * (vr + heps * fr); vi = es * (vi + heps * fi); }
      }
      if (PAIR) {
        if (MID) { m1[r] = ok ? lj : 0.0; mk[r] = ok ? fma(vr, vr, vi * vi) : 0.0; }
        if (a.flip) { vr = -vr; vi = -vi; }
        const double h2 = 0.5 * a.eps2;
        const double lj2 = a.fwd2 ? h2 * s : -h2 * s;
        if (!MID) { if (ok) ld[i][r] += lj2; }
        else m2[r] = ok ? lj + lj2 : 0.0;
        // same step size and direction (adjacent updates of an un-trained or eps_fixed model, i.e.
        // six of the seven pairs of a merged nleapfrog = 4 trajectory): lj2 == lj bit for bit, the
        // two exponentials are the first update's (wave-uniform branch; pair 1.21 -> 1.15 ms)
        double es2, eq2;
        if (a.eps2 == a.eps && a.fwd2 == (int)FWD) { es2 = es; eq2 = eq; }
        else { es2 = exp_bf(lj2); eq2 = exp_bf(a.eps2 * q); }
        const double fr = fr0 * eq2 + t, fi = fi0 * eq2;
        if (a.fwd2) { vr = es2 * vr - h2 * fr; vi = es2 * vi - h2 * fi; }
        else { vr = es2 * (vr + h2 * fr); vi = es2 * (vi + h2 * fi); }
      }
      if (ok) {
        if (CPLX) reinterpret_cast<double2*>(a.v)[o] = make_double2(vr, vi);
        else a.v[o] = vr;
      }
    }
    if (PAIR && MID) {
      // row partials of this batch over the 16 column lanes; partial column = (tile, wave half, j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double x1 = m1[r], x2 = m2[r], xk = mk[r];
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
          x1 += __shfl_xor(x1, off, 64); x2 += __shfl_xor(x2, off, 64); xk += __shfl_xor(xk, off, 64);
        }
        const long m = m0 + wm + 16 * i + Mfma<double>::row(lane, r);
        if ((lane & 15) == 0 && m < a.M) {
          const long col = (n0 / BN) * 4 + (wave & 1) * 2 + j;
          a.logdet_part[m * a.ncols_part + col] = x2;
          a.ld1_part[m * a.ncols_part + col] = x1;
          a.ke_part[m * a.ncols_part + col] = xk;
        }
      }
    }
  }
  if (PAIR && MID) return;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double x = ld[i][r];
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
      const long m = m0 + wm + 16 * i + Mfma<double>::row(lane, r);
      if ((lane & 15) == 0 && m < a.M) {
        const long col = (n0 / BN) * 2 + (wave & 1);
        a.logdet_part[m * a.ncols_part + col] = x;
      }
    }
}

// ---------------------------------------------------------------------------------------
// fp64 layer with LDS-DMA operand staging (see fused_heads_dma_kernel for the scheme): 128 x 128
// tile, 2 x 2 wavefronts of 64 x 64, two 32 KB stages, one barrier per K-slab, no address
// arithmetic in the loop.  A K-contiguous operand ([rows][K]) is staged as 128-byte XOR-swizzled
// rows; a TRANSPOSED operand ([K][rows], the backward GEMMs dW = dY^T X, dX = dY W) is staged as
// its own image [16 k][128 rows] -- one 1 KB LDS-DMA instruction per k, fragments are 128-byte
// contiguous runs, no swizzle needed.  Serves 16-byte aligned operands with K, K2 and the split
// boundaries multiples of BK (every layer of the SU(3) vnet, forward and backward); everything else
// stays on gemm_nt_kernel.  Rows past M / N re-read valid rows and are masked at the store.
template <bool FUSED, bool TA, bool TW>
__global__ __launch_bounds__(kBlock, 2) void gemm_dma_f64_kernel(
    const double* __restrict__ A, const double* __restrict__ W, const double* __restrict__ A2,
    const double* __restrict__ W2, int M, int N, long K, long K2, long kchunk, Epilogue<double> epi,
    double* __restrict__ C, double* __restrict__ part, int swz) {
  constexpr int BM = 128, BN = 128;
  constexpr int ROWB = BK * 8;
  constexpr int OPB = BM * ROWB;                 // 16 KB per operand per stage
  constexpr int STAGE = 2 * OPB;
  using T = double;
  using acc_t = v4f64;
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  // Logical block order: m-tiles fastest, then n-tiles, then K-splits, and each XCD gets a
  // contiguous range of it (hardware block b runs on XCD b % 8): the blocks that read the same
  // W rows / the same K-chunk of A run back to back on ONE XCD and share its L2.  Without this
  // the 2 x 2 tiles of a K-chunk of the input layer (256 x 256 x 262144, 128 splits) land on four
  // XCDs and A and W are each read twice from HBM (PMC: 2.21 GB per launch for 1.14 GB of operands).
  const long nbx = gridDim.x, nby = gridDim.y;
  const long lin = blockIdx.x + nbx * (blockIdx.y + nby * (long)blockIdx.z);
  const long wlog = xcd_swizzle(lin, nbx * nby * gridDim.z, swz);
  const long by = wlog % nby, bx = (wlog / nby) % nbx, bz = wlog / (nbx * nby);
  const long m0 = by * BM, n0 = bx * BN;
  const long Kt = K + K2;
  const long kbeg = bz * kchunk;
  long kend = kbeg + kchunk;
  if (kend > Kt) kend = Kt;

  // loader: this wavefront issues instructions g = 4q + wave, q = 0..7; g < 16 belongs to A, the
  // rest to W.  K-contiguous operand: instruction fills tile rows (g & 15) * 8 + (lane >> 3),
  // 16-byte position lane & 7, source chunk (lane & 7) ^ (row & 7); per-lane byte offsets are
  // relative to the block's first row, one set per operand pair.  Transposed operand: instruction
  // fills k-row g & 15 with rows 2 lane, 2 lane + 1; the per-lane offset does not depend on q.
  unsigned vo1[8], vo2[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const bool isA = q < 4;
    const long lim = (isA ? (long)M - m0 : (long)N - n0) - 1;     // last valid row of this block
    if ((isA && TA) || (!isA && TW)) {
      long r = 2 * lane;
      if (r + 1 > lim) r = lim - 1;                  // clamp the pair (M, N even and >= 2)
      vo1[q] = (unsigned)(r * 8);
      vo2[q] = vo1[q];
    } else {
      const int R = ((4 * q + wave) & 15) * 8 + (lane >> 3);
      const int rc = R <= lim ? R : (int)lim;
      const int c = (lane & 7) ^ L2Q_SWZ(R);
      vo1[q] = (unsigned)(rc * K * 8 + c * 16);
      vo2[q] = (unsigned)(rc * K2 * 8 + c * 16);
    }
  }
  // block bases: K-contiguous operands start at row m0 / n0; transposed ones at column m0 / n0
  const char* a1 = reinterpret_cast<const char*>(A) + (TA ? m0 * 8 : m0 * K * 8);
  const char* w1 = reinterpret_cast<const char*>(W) + (TW ? n0 * 8 : n0 * K * 8);
  const char* a2 = reinterpret_cast<const char*>(A2) + m0 * K2 * 8;      // K2 != 0 only without TA / TW
  const char* w2 = reinterpret_cast<const char*>(W2) + n0 * K2 * 8;
  auto issue = [&](int stage, long k0, int q0 = 0, int q1 = 8) {
    const bool second = k0 >= K;                       // wave-uniform: a slab never straddles K
    const long kk = second ? k0 - K : k0;
#pragma unroll
    for (int q = q0; q < q1; ++q) {
      const int g = 4 * q + wave;
      const char* src;
      if (q < 4) src = TA ? a1 + (kk + (g & 15)) * (long)M * 8 : (second ? a2 : a1) + kk * 8;
      else src = TW ? w1 + (kk + (g & 15)) * (long)N * 8 : (second ? w2 : w1) + kk * 8;
      src += (unsigned long)(second ? vo2[q] : vo1[q]);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (lds_ptr_t)(lds + stage * STAGE + g * 1024), 16, 0, 0);
    }
  };
  unsigned offA[4], offB[4];
#pragma unroll
  for (int kq = 0; kq < 4; ++kq) {
    const unsigned sw = ((((kq * 2) + (lane >> 5)) ^ L2Q_SWZ(lane)) << 4) + ((lane >> 4) & 1) * 8;
    const unsigned tr = (4 * kq + (lane >> 4)) * (BM * 8) + (lane & 15) * 8;
    offA[kq] = TA ? tr + wm * 8 : (wm + (lane & 15)) * ROWB + sw;
    offB[kq] = OPB + (TW ? tr + wn * 8 : (wn + (lane & 15)) * ROWB + sw);
  }
  constexpr int stepA = TA ? 16 * 8 : 16 * ROWB, stepB = TW ? 16 * 8 : 16 * ROWB;
  acc_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (acc_t){0, 0, 0, 0};

  issue(0, kbeg);
  int st = 0;
  for (long k0 = kbeg; k0 < kend; k0 += BK, st ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!L2Q_GD_SPREAD && k0 + BK < kend) issue(st ^ 1, k0 + BK);
    // (L2Q_GD_SPREAD: the next slab's eight LDS-DMA pieces go out two per k-quad, behind eight MFMAs each,
    // instead of back to back; past the last slab the last one is fetched again into the idle stage)
    const long kn = k0 + BK < kend ? k0 + BK : k0;
    const char* sb = lds + st * STAGE;
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      T fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = lds_frag<T>(sb + offA[kq] + i * stepA);
        fb[i] = lds_frag<T>(sb + offB[kq] + i * stepB);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = Mfma<T>::run(fa[i], fb[j], acc[i][j]);
        if (L2Q_GD_SPREAD && (i & 1)) {
          __builtin_amdgcn_sched_barrier(0);
          issue(st ^ 1, kn, 2 * kq + (i >> 1), 2 * kq + (i >> 1) + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }

  T* dst = FUSED ? C : part + bz * M * N;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long n = n0 + wn + 16 * j + (lane & 15);
    if (n >= N) continue;
    T cs = (T)1, cb = (T)0;
    if (FUSED) { cs = epi.colscale((int)n); cb = epi.colbias((int)n); }
    // accumulate (dW += ...): all sixteen previous values of this column first -- C is read and
    // written through one pointer, element by element the loads would each wait for the
    // previous element's store (64 dependent HBM round trips per lane)
    T prev[4][4];
    if (FUSED && epi.accumulate) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long m = m0 + wm + 16 * i + Mfma<T>::row(lane, r);
          prev[i][r] = m < M ? dst[m * N + n] : (T)0;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long m = m0 + wm + 16 * i + Mfma<T>::row(lane, r);
        if (m < M) {
          const T v = acc[i][j][r];
          if (FUSED) {
            const T y = cs * apply_act<T>(v + cb, epi.act);
            dst[m * N + n] = epi.accumulate ? prev[i][r] + y : y;
          } else {
            dst[m * N + n] = v;
          }
        }
      }
  }
}

static int pick_splits(int M, int N, long Kt) {
  const long tiles = cdiv(M, 128) * cdiv(N, 128);
  if (tiles >= 256 || Kt <= 8 * BK) return 1;
  long s = cdiv(512, tiles);
  const long maxs = Kt / (4 * BK) > 0 ? Kt / (4 * BK) : 1;
  if (s > maxs) s = maxs;
  // (one or two output tiles -- a 16-chain micro-batch against a 256-wide layer: two workgroups per CU, each
  // issuing the MFMAs of a full 128-row tile whatever M is, are what sets the streaming rate of the weights)
  if (s > (tiles <= 2 ? 256 : 128)) s = tiles <= 2 ? 256 : 128;
  return (int)(s < 1 ? 1 : s);
}

template <typename T>
static bool vec_ok(const T* A, const T* W, const T* A2, const T* W2, long K, long K2) {
  constexpr long V = Mfma<T>::VEC;
  auto al = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  return K % V == 0 && K2 % V == 0 && al(A) && al(W) && al(A2) && al(W2);
}

// ta / tw: operand stored transposed (see TileLoaderT); accumulate: C += result.
template <typename T>
static int gemm_launch(const T* A, const T* W, int M, int N, long K, const T* A2, const T* W2,
                       long K2, const T* bias, const T* bias2, const T* coeff, T scale, int act,
                       T* C, void* ws, size_t ws_bytes, hipStream_t st, int ta = 0, int tw = 0,
                       int accumulate = 0) {
  const long Kt = K + K2;
  int splits = pick_splits(M, N, Kt);
  long kchunk = cdiv(cdiv(Kt, splits), BK) * BK;
  splits = (int)cdiv(Kt, kchunk);
  Epilogue<T> epi{bias, bias2, coeff, scale, act, accumulate};
  const dim3 grid((unsigned)cdiv(N, 128), (unsigned)cdiv(M, 128), (unsigned)splits);
  constexpr long V = Mfma<T>::VEC;
  auto al = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  // per-operand vector-load conditions: K-contiguous operands need K % VEC, transposed ones
  // their leading dimension (M resp. N) % VEC
  const bool veca = ta ? (M % V == 0 && al(A)) : (K % V == 0 && K2 % V == 0 && al(A) && al(A2));
  const bool vecw = tw ? (N % V == 0 && al(W)) : (K % V == 0 && K2 % V == 0 && al(W) && al(W2));
  if ((ta || tw) && (K2 != 0 || !veca || !vecw)) {
    set_error("l2q_gemm: transposed operands need K2 == 0 and 16-byte aligned rows");
    return L2Q_ESHAPE;
  }
  const bool vec = veca && vecw;
  T* part = nullptr;
  if (splits > 1) {
    const size_t need = (size_t)splits * M * N * sizeof(T);
    if (!ws || ws_bytes < need) {
      set_error("l2q_gemm: split-K workspace too small (%zu < %zu)", ws_bytes, need);
      return L2Q_ESHAPE;
    }
    part = (T*)ws;
  }
#define L2Q_GEMM(F, V, TA_, TW_)                                                               \
  hipLaunchKernelGGL((gemm_nt_kernel<T, F, V, TA_, TW_>), grid, dim3(kBlock), 0, st, A, W, A2, \
                     W2, M, N, K, K2, kchunk, epi, C, part)
#define L2Q_GEMM_F(F)                                                        \
  do {                                                                       \
    if (ta && tw) L2Q_GEMM(F, true, true, true);                             \
    else if (tw) L2Q_GEMM(F, true, false, true);                             \
    else if (ta) L2Q_GEMM(F, true, true, false);                             \
    else if (vec) L2Q_GEMM(F, true, false, false);                           \
    else L2Q_GEMM(F, false, false, false);                                   \
  } while (0)
  // fp64, 16-byte aligned operands, whole K-slabs: LDS-DMA staged kernel
  bool dma = false;
  if constexpr (std::is_same<T, double>::value) {
    dma = tuning().heads_dma && vec && K % BK == 0 && K2 % BK == 0 && kchunk % BK == 0 &&
          (ta ? (M % 2 == 0 && M >= 2) : 128 * K * 8 < (1L << 32)) &&
          (tw ? (N % 2 == 0 && N >= 2) : 128 * K * 8 < (1L << 32)) && 128 * K2 * 8 < (1L << 32);
    if (dma) {
      const double* A2p = K2 ? A2 : A;            // never dereferenced when K2 == 0
      const double* W2p = K2 ? W2 : W;
#define L2Q_GEMM_DMA(F, TA_, TW_)                                                                   \
  hipLaunchKernelGGL((gemm_dma_f64_kernel<F, TA_, TW_>), grid, dim3(kBlock), 0, st, A, W, A2p, W2p, \
                     M, N, K, K2, kchunk, epi, C, part, tuning().xcd_swizzle)
#define L2Q_GEMM_DMA_F(F)                                  \
  do {                                                     \
    if (ta && tw) L2Q_GEMM_DMA(F, true, true);             \
    else if (tw) L2Q_GEMM_DMA(F, false, true);             \
    else if (ta) L2Q_GEMM_DMA(F, true, false);             \
    else L2Q_GEMM_DMA(F, false, false);                    \
  } while (0)
      if (splits == 1) L2Q_GEMM_DMA_F(true);
      else L2Q_GEMM_DMA_F(false);
#undef L2Q_GEMM_DMA_F
#undef L2Q_GEMM_DMA
    }
  }
  if (dma) {
  } else if (splits == 1) L2Q_GEMM_F(true);
  else L2Q_GEMM_F(false);
#undef L2Q_GEMM_F
#undef L2Q_GEMM
  if (splits > 1) {
    const long MN = (long)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3((unsigned)cdiv(MN, 64)), dim3(kBlock), 0,
                       st, (const T*)ws, splits, MN, N, epi, C);
  }
  return check_launch("l2q_gemm");
}

}  // namespace l2q

using namespace l2q;

extern "C" {

size_t l2q_gemm_ws_bytes(int M, int N, long K, long K2) {
  if (M <= 0 || N <= 0 || K + K2 <= 0) return 0;
  const int splits = pick_splits(M, N, K + K2);
  return splits == 1 ? 0 : (size_t)(splits + 1) * M * N * sizeof(double);
}

int l2q_gemm_f64(const double* A, const double* W, int M, int N, long K, const double* A2,
                 const double* W2, long K2, const double* bias, const double* bias2,
                 const double* coeff, double scale, int act, double* C, void* ws, size_t ws_bytes,
                 void* stream) {
  L2Q_REQUIRE(A && W && C, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(M > 0 && N > 0 && K > 0 && K2 >= 0, L2Q_EINVAL, "non-positive size");
  L2Q_REQUIRE(K2 == 0 || (A2 && W2), L2Q_EINVAL, "second operand pair missing");
  L2Q_REQUIRE(act >= L2Q_ACT_NONE && act <= L2Q_ACT_SWISH, L2Q_EINVAL, "bad activation");
  return gemm_launch<double>(A, W, M, N, K, A2, W2, K2, bias, bias2, coeff, scale, act, C, ws,
                             ws_bytes, (hipStream_t)stream);
}

int l2q_gemm_f32(const float* A, const float* W, int M, int N, long K, const float* A2,
                 const float* W2, long K2, const float* bias, const float* bias2,
                 const float* coeff, float scale, int act, float* C, void* ws, size_t ws_bytes,
                 void* stream) {
  L2Q_REQUIRE(A && W && C, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(M > 0 && N > 0 && K > 0 && K2 >= 0, L2Q_EINVAL, "non-positive size");
  L2Q_REQUIRE(K2 == 0 || (A2 && W2), L2Q_EINVAL, "second operand pair missing");
  L2Q_REQUIRE(act >= L2Q_ACT_NONE && act <= L2Q_ACT_SWISH, L2Q_EINVAL, "bad activation");
  return gemm_launch<float>(A, W, M, N, K, A2, W2, K2, bias, bias2, coeff, scale, act, C, ws,
                            ws_bytes, (hipStream_t)stream);
}

int l2q_gemm_ex(const void* A, int a_trans, const void* W, int w_trans, int M, int N, long K,
                int elem_bytes, int accumulate, void* C, void* ws, size_t ws_bytes, void* stream) {
  L2Q_REQUIRE(A && W && C, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(M > 0 && N > 0 && K > 0, L2Q_EINVAL, "non-positive size");
  L2Q_REQUIRE(elem_bytes == 4 || elem_bytes == 8, L2Q_EINVAL, "elem_bytes must be 4 or 8");
  hipStream_t st = (hipStream_t)stream;
  if (elem_bytes == 8)
    return gemm_launch<double>((const double*)A, (const double*)W, M, N, K, nullptr, nullptr, 0,
                               nullptr, nullptr, nullptr, 1.0, L2Q_ACT_NONE, (double*)C, ws, ws_bytes,
                               st, a_trans, w_trans, accumulate);
  return gemm_launch<float>((const float*)A, (const float*)W, M, N, K, nullptr, nullptr, 0, nullptr,
                            nullptr, nullptr, 1.0f, L2Q_ACT_NONE, (float*)C, ws, ws_bytes, st,
                            a_trans, w_trans, accumulate);
}

int l2q_conv_gemm_periodic_f32(const float* in, long sn, long sc, long sh, long sw, int nb, int C,
                               int H, int W, int k, const float* weight, int channels_last_cols,
                               const float* bias, int cout, int act, float* out, void* stream) {
  L2Q_REQUIRE(in && weight && out, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && C > 0 && H > 0 && W > 0 && k > 0 && cout > 0, L2Q_EINVAL,
              "non-positive size");
  L2Q_REQUIRE(act >= L2Q_ACT_NONE && act <= L2Q_ACT_SWISH, L2Q_EINVAL, "bad activation");
  ConvGeom g;
  g.sn = sn; g.sc = sc; g.sh = sh; g.sw = sw; g.C = C; g.H = H; g.W = W; g.k = k;
  g.Ho = H + k - 1; g.Wo = W + k - 1; g.Kc = C * k * k;
  g.clast = channels_last_cols ? 1 : 0;
  g.M = (long)nb * g.Ho * g.Wo;
  L2Q_REQUIRE(cdiv(g.M, 128) < 65536L * 16 && g.M < (1L << 31), L2Q_ESHAPE, "too many output pixels");
  Epilogue<float> epi{bias, nullptr, nullptr, 1.0f, act, 0};
  const int bn = cout <= 32 ? 32 : cout <= 64 ? 64 : 128;
  const dim3 grid((unsigned)cdiv(cout, bn), (unsigned)cdiv(g.M, 128)), block(kBlock);
  hipStream_t st = (hipStream_t)stream;
  // 16-byte channel gathers: NHWC input, (i, j, ci) order, C % 4 == 0, aligned
  const bool vec4 = g.clast && sc == 1 && C % 4 == 0 && sw % 4 == 0 && sh % 4 == 0 && sn % 4 == 0 &&
                    (reinterpret_cast<uintptr_t>(in) & 15) == 0;
#define L2Q_CGB(KS, BNV)                                                                          \
  do {                                                                                            \
    if (vec4) hipLaunchKernelGGL((conv_gemm_kernel<KS, BNV, true>), grid, block, 0, st, in, g,    \
                                 weight, cout, epi, out);                                         \
    else hipLaunchKernelGGL((conv_gemm_kernel<KS, BNV, false>), grid, block, 0, st, in, g,        \
                            weight, cout, epi, out);                                              \
  } while (0)
#define L2Q_CG(KS)                                                     \
  do {                                                                 \
    if (bn == 32) L2Q_CGB(KS, 32);                                     \
    else if (bn == 64) L2Q_CGB(KS, 64);                                \
    else L2Q_CGB(KS, 128);                                             \
  } while (0)
  switch (k) {
    case 1: L2Q_CG(1); break;
    case 2: L2Q_CG(2); break;
    case 3: L2Q_CG(3); break;
    case 4: L2Q_CG(4); break;
    case 5: L2Q_CG(5); break;
    default: L2Q_CG(0); break;
  }
#undef L2Q_CG
#undef L2Q_CGB
  return check_launch("l2q_conv_gemm_periodic_f32");
}

size_t l2q_vnet_heads_ws_bytes(int M, long N) {
  if (M <= 0 || N <= 0) return 0;
  // (three partial arrays of 4 columns per tile for the pair kernel with mid-point outputs)
  return (size_t)M * (size_t)(cdiv(N, kHeadsBN) * 4) * 3 * sizeof(double) + 256;
}

static int heads_launch(const double* Z, int M, int K, long N, const double* Ws, const double* bs,
                        const double* cs, double scale_s, const double* Wt, const double* bt,
                        double scale_t, const double* Wq, const double* bq, const double* cq,
                        double scale_q, void* v, const void* force, int is_complex, double eps,
                        int forward, int pair, double eps2, int forward2, int flip,
                        double* logdet, void* ws, size_t ws_bytes, void* stream,
                        double* logdet1 = nullptr, double* vnorm2_mid = nullptr,
                        const void* v_in = nullptr) {
  L2Q_REQUIRE(Z && Ws && bs && Wt && bt && Wq && bq && v && force && logdet && ws, L2Q_EINVAL,
              "null pointer");
  L2Q_REQUIRE(M > 0 && K > 0 && N > 0 && N < 2000000000L, L2Q_EINVAL, "bad size");
  L2Q_REQUIRE(K % 2 == 0, L2Q_ESHAPE, "K (last hidden width) must be even");
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  L2Q_REQUIRE(al(Z) && al(Ws) && al(Wt) && al(Wq) && al(v) && al(force), L2Q_ESHAPE,
              "operands must be 16-byte aligned");
  const long ntile = cdiv(N, kHeadsBN), mtile = cdiv(M, 64);
  const bool mid = logdet1 != nullptr;
  L2Q_REQUIRE(!mid || (pair && vnorm2_mid), L2Q_EINVAL, "mid-point outputs belong to the pair kernel");
  const int ncols = (int)(ntile * (mid ? 4 : 2));
  L2Q_REQUIRE(ws_bytes >= (size_t)M * ncols * (mid ? 3 : 1) * sizeof(double), L2Q_ESHAPE,
              "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  HeadsArgs a;
  a.Z = Z; a.W[0] = Ws; a.W[1] = Wt; a.W[2] = Wq; a.b[0] = bs; a.b[1] = bt; a.b[2] = bq;
  a.cs = cs; a.cq = cq; a.ss = scale_s; a.st = scale_t; a.sq = scale_q; a.eps = eps;
  a.eps2 = eps2; a.fwd2 = forward2; a.flip = flip;
  a.v = (double*)v; a.vin = v_in ? (const double*)v_in : (const double*)v;
  L2Q_REQUIRE(al(a.vin), L2Q_ESHAPE, "operands must be 16-byte aligned");
  a.F = (const double*)force; a.logdet_part = (double*)ws;
  a.ld1_part = (double*)ws + (size_t)M * ncols;
  a.ke_part = (double*)ws + 2 * (size_t)M * ncols;
  a.M = M; a.N = (int)N; a.K = K; a.ncols_part = ncols;
  const dim3 grid((unsigned)(ntile * mtile)), block(kBlock);
  const int swz = tuning().xcd_swizzle;
  const int stg = tuning().heads_stagger;
  // partial columns of wave tiles that fall entirely beyond N are never written: clear first
  launch_zero(ws, (size_t)M * ncols * (mid ? 3 : 1) * sizeof(double), st);
  // LDS-DMA kernel whenever the K-slabs are whole (tuning heads_dma = 0 keeps the older kernel)
  const bool dma_ok = (K % BK == 0) && K >= BK && K <= (1 << 20);
  const bool dma = tuning().heads_dma && dma_ok;
#define L2Q_HEADS(C, F, P)                                                                      \
  do {                                                                                          \
    if (dma) hipLaunchKernelGGL((fused_heads_dma_kernel<C, F, P>), grid, block, 0, st, a, swz); \
    else hipLaunchKernelGGL((fused_heads_vupdate_kernel<C, F, P>), grid, block, 0, st, a, swz, stg); \
  } while (0)
  if (mid) {
    // (only the LDS-DMA kernel has the mid-point variant: it runs whatever `heads_dma` says --
    // results never depend on the knobs)
    L2Q_REQUIRE(dma_ok, L2Q_ESHAPE, "mid-point outputs need the LDS-DMA kernel (K % 16 == 0)");
#define L2Q_HEADS_MID(C, F) \
  hipLaunchKernelGGL((fused_heads_dma_kernel<C, F, true, true>), grid, block, 0, st, a, swz)
    if (is_complex) { if (forward) L2Q_HEADS_MID(true, true); else L2Q_HEADS_MID(true, false); }
    else { if (forward) L2Q_HEADS_MID(false, true); else L2Q_HEADS_MID(false, false); }
#undef L2Q_HEADS_MID
  } else if (pair) {
    if (is_complex) { if (forward) L2Q_HEADS(true, true, true); else L2Q_HEADS(true, false, true); }
    else { if (forward) L2Q_HEADS(false, true, true); else L2Q_HEADS(false, false, true); }
  } else {
    if (is_complex) { if (forward) L2Q_HEADS(true, true, false); else L2Q_HEADS(true, false, false); }
    else { if (forward) L2Q_HEADS(false, true, false); else L2Q_HEADS(false, false, false); }
  }
#undef L2Q_HEADS
  launch_finalize((const double*)ws, logdet, M, ncols, 1, 1.0, 0.0, st);
  if (mid) {
    launch_finalize(a.ld1_part, logdet1, M, ncols, 1, 1.0, 0.0, st);
    launch_finalize(a.ke_part, vnorm2_mid, M, ncols, 1, 1.0, 0.0, st);
  }
  return check_launch("l2q_vnet_heads_vupdate_f64");
}

int l2q_vnet_heads_vupdate_f64(const double* Z, int M, int K, long N, const double* Ws,
                               const double* bs, const double* cs, double scale_s,
                               const double* Wt, const double* bt, double scale_t,
                               const double* Wq, const double* bq, const double* cq,
                               double scale_q, void* v, const void* force, int is_complex,
                               double eps, int forward, double* logdet, void* ws, size_t ws_bytes,
                               void* stream) {
  return heads_launch(Z, M, K, N, Ws, bs, cs, scale_s, Wt, bt, scale_t, Wq, bq, cq, scale_q, v,
                      force, is_complex, eps, forward, 0, 0.0, 0, 0, logdet, ws, ws_bytes, stream);
}

int l2q_vnet_heads_vupdate_to_f64(const double* Z, int M, int K, long N, const double* Ws,
                                  const double* bs, const double* cs, double scale_s,
                                  const double* Wt, const double* bt, double scale_t,
                                  const double* Wq, const double* bq, const double* cq,
                                  double scale_q, const void* v_in, void* v_out, const void* force,
                                  int is_complex, double eps, int forward, double* logdet, void* ws,
                                  size_t ws_bytes, void* stream) {
  L2Q_REQUIRE(v_in, L2Q_EINVAL, "null pointer");
  return heads_launch(Z, M, K, N, Ws, bs, cs, scale_s, Wt, bt, scale_t, Wq, bq, cq, scale_q, v_out,
                      force, is_complex, eps, forward, 0, 0.0, 0, 0, logdet, ws, ws_bytes, stream,
                      nullptr, nullptr, v_in);
}

int l2q_vnet_heads_vupdate_pair_f64(const double* Z, int M, int K, long N, const double* Ws,
                                    const double* bs, const double* cs, double scale_s,
                                    const double* Wt, const double* bt, double scale_t,
                                    const double* Wq, const double* bq, const double* cq,
                                    double scale_q, void* v, const void* force, int is_complex,
                                    double eps1, int forward1, int flip_between, double eps2,
                                    int forward2, double* logdet, void* ws, size_t ws_bytes,
                                    void* stream) {
  return heads_launch(Z, M, K, N, Ws, bs, cs, scale_s, Wt, bt, scale_t, Wq, bq, cq, scale_q, v,
                      force, is_complex, eps1, forward1, 1, eps2, forward2, flip_between, logdet,
                      ws, ws_bytes, stream);
}

int l2q_vnet_heads_vupdate_pair_mid_f64(const double* Z, int M, int K, long N, const double* Ws,
                                        const double* bs, const double* cs, double scale_s,
                                        const double* Wt, const double* bt, double scale_t,
                                        const double* Wq, const double* bq, const double* cq,
                                        double scale_q, void* v, const void* force, int is_complex,
                                        double eps1, int forward1, int flip_between, double eps2,
                                        int forward2, double* logdet, double* logdet1,
                                        double* vnorm2_mid, void* ws, size_t ws_bytes, void* stream) {
  L2Q_REQUIRE(logdet1 && vnorm2_mid, L2Q_EINVAL, "null pointer");
  return heads_launch(Z, M, K, N, Ws, bs, cs, scale_s, Wt, bt, scale_t, Wq, bq, cq, scale_q, v,
                      force, is_complex, eps1, forward1, 1, eps2, forward2, flip_between, logdet,
                      ws, ws_bytes, stream, logdet1, vnorm2_mid);
}

}  // extern "C"
