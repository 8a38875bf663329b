// gemm.hip -- dense layers of the xnet / vnet leapfrog networks on the CDNA4 matrix cores.
//
//   C[M][N] = epi( A[M][K] . W[N][K]^T  (+ A2[M][K2] . W2[N][K2]^T) + bias (+ bias2) )
//
// Both operands are K-contiguous (nn.Linear weight layout), so one LDS-tiled "NT" kernel
// serves every layer.  fp64 uses v_mfma_f64_16x16x4_f64, fp32 v_mfma_f32_16x16x4_f32 (exact
// IEEE fma chains, no reduced-precision path).  Tile 128x128x16 per 256-thread workgroup,
// each of the 4 wavefronts owns a 64x64 quadrant = 4x4 MFMA tiles (64 accumulators/lane).
// The L2HMC layers are skinny (M = #chains <= a few hundred, K or N = 32V..36V ~ 1e5), so
// the K loop is split across workgroups (split-K) with a fixed-order second-stage reduce:
// deterministic, no atomics.
#include "l2q_common.hpp"

namespace l2q {

constexpr int BM = 128, BN = 128, BK = 16, LDP = BK + 2;   // +2: conflict-free ds_read_b64/b32

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef float v4f32 __attribute__((ext_vector_type(4)));

template <typename T> struct Mfma;
template <> struct Mfma<double> {
  using acc_t = v4f64;
  static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <> struct Mfma<float> {
  using acc_t = v4f32;
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = 4 * (lane >> 4) + reg
  static __device__ __forceinline__ int row(int lane, int r) { return 4 * (lane >> 4) + r; }
};

template <typename T>
__device__ __forceinline__ T apply_act(T z, int act) {
  switch (act) {
    case L2Q_ACT_TANH: return tanh(z);
    case L2Q_ACT_RELU: return z > (T)0 ? z : (T)0;
    case L2Q_ACT_LEAKY_RELU: return z > (T)0 ? z : (T)0.01 * z;
    case L2Q_ACT_ELU: return z > (T)0 ? z : expm1(z);
    case L2Q_ACT_SWISH: return z / ((T)1 + exp(-z));
    default: return z;
  }
}

template <typename T>
struct Epilogue {
  const T* bias;
  const T* bias2;
  const T* coeff;
  T scale;
  int act;
  __device__ __forceinline__ T operator()(T acc, int n) const {
    T z = acc;
    if (bias) z += bias[n];
    if (bias2) z += bias2[n];
    z = apply_act<T>(z, act);
    return coeff ? scale * exp(coeff[n]) * z : scale * z;
  }
};

// element (row, kk) of the virtual K-concatenated operand [P | P2], zero outside
template <typename T>
__device__ __forceinline__ T load_cat(const T* __restrict__ p, const T* __restrict__ p2, long row,
                                      long nrows, long kk, long K, long K2) {
  if (row >= nrows) return (T)0;
  if (kk < K) return p[row * K + kk];
  kk -= K;
  if (kk < K2) return p2[row * K2 + kk];
  return (T)0;
}

// grid: x = N tiles, y = M tiles, z = K splits.  FUSED: splits == 1, epilogue applied here;
// otherwise raw partial sums go to part[z][M][N].
template <typename T, bool FUSED>
__global__ __launch_bounds__(kBlock, 2) void gemm_nt_kernel(
    const T* __restrict__ A, const T* __restrict__ W, const T* __restrict__ A2,
    const T* __restrict__ W2, int M, int N, long K, long K2, long kchunk, Epilogue<T> epi,
    T* __restrict__ C, T* __restrict__ part) {
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

  // global -> register staging: thread owns column kq of 8 rows (rq + 16 i)
  const int kq = tid & 15, rq = tid >> 4;
  T ra[8], rw[8];
  auto fetch = [&](long k0) {
    const long kk = k0 + kq;
    const bool kin = kk < kend;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      ra[i] = kin ? load_cat<T>(A, A2, m0 + rq + 16 * i, M, kk, K, K2) : (T)0;
      rw[i] = kin ? load_cat<T>(W, W2, n0 + rq + 16 * i, N, kk, K, K2) : (T)0;
    }
  };
  fetch(kbeg);
  for (long k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();                                 // previous tile fully consumed
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      As[rq + 16 * i][kq] = ra[i];
      Ws[rq + 16 * i][kq] = rw[i];
    }
    __syncthreads();
    if (k0 + BK < kend) fetch(k0 + BK);              // overlap next tile's HBM/L2 latency
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
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long m = m0 + wm + 16 * i + Mfma<T>::row(lane, r);
        const long n = n0 + wn + 16 * j + (lane & 15);
        if (m < M && n < N) {
          const T v = acc[i][j][r];
          dst[m * N + n] = FUSED ? epi(v, (int)n) : v;
        }
      }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void splitk_reduce_kernel(const T* __restrict__ part,
                                                               int splits, long MN, int N,
                                                               Epilogue<T> epi, T* __restrict__ C) {
  const long i = (long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= MN) return;
  T s = (T)0;
  for (int z = 0; z < splits; ++z) s += part[(long)z * MN + i];     // fixed order
  C[i] = epi(s, (int)(i % N));
}

static int pick_splits(int M, int N, long Kt) {
  const long tiles = cdiv(M, BM) * cdiv(N, BN);
  if (tiles >= 256 || Kt <= 8 * BK) return 1;
  long s = cdiv(512, tiles);
  const long maxs = Kt / (4 * BK) > 0 ? Kt / (4 * BK) : 1;
  if (s > maxs) s = maxs;
  if (s > 128) s = 128;
  return (int)(s < 1 ? 1 : s);
}

template <typename T>
static int gemm_launch(const T* A, const T* W, int M, int N, long K, const T* A2, const T* W2,
                       long K2, const T* bias, const T* bias2, const T* coeff, T scale, int act,
                       T* C, void* ws, size_t ws_bytes, hipStream_t st) {
  const long Kt = K + K2;
  int splits = pick_splits(M, N, Kt);
  long kchunk = cdiv(cdiv(Kt, splits), BK) * BK;
  splits = (int)cdiv(Kt, kchunk);
  Epilogue<T> epi{bias, bias2, coeff, scale, act};
  const dim3 grid((unsigned)cdiv(N, BN), (unsigned)cdiv(M, BM), (unsigned)splits);
  if (splits == 1) {
    hipLaunchKernelGGL((gemm_nt_kernel<T, true>), grid, dim3(kBlock), 0, st, A, W, A2, W2, M, N, K,
                       K2, kchunk, epi, C, (T*)nullptr);
  } else {
    const size_t need = (size_t)splits * M * N * sizeof(T);
    if (!ws || ws_bytes < need) {
      set_error("l2q_gemm: split-K workspace too small (%zu < %zu)", ws_bytes, need);
      return L2Q_ESHAPE;
    }
    hipLaunchKernelGGL((gemm_nt_kernel<T, false>), grid, dim3(kBlock), 0, st, A, W, A2, W2, M, N,
                       K, K2, kchunk, epi, (T*)nullptr, (T*)ws);
    const long MN = (long)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3((unsigned)cdiv(MN, kBlock)), dim3(kBlock), 0,
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

}  // extern "C"
