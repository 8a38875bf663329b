// gemm_f16_small.hip -- the HIDDEN layers of the half-precision networks on many chains:
//
//   C[M][N] = epi( A[M][K] . W[N][K]^T + bias )      A, W 16-bit, K <= 256, N <= 256, M large
//
// (network.py:283-326: LeapfrogLayer's hidden Linear layers, units[i] -> units[i + 1]; at BASELINE cfg-3 8192 x 256 x
// 256, 64 launches per trajectory.)  gemm_nt_h_kernel runs this shape as 128 tiles of 128 x 128 on half of the CUs,
// four K-slabs each behind two barriers: 39 us for 1 GFLOP.  Here a workgroup owns 32 chains and ALL N columns (one
// workgroup per CU at 8192 chains): the A tile (32 x K, <= 16 KB) goes through LDS once, one barrier; each
// wavefront requests its 64 rows of W straight into MFMA operands (W is <= 128 KB: L2-resident after the first
// workgroup), two K-steps ahead.  Same rounding points as gemm_nt_h_kernel (half_common.hpp); K is summed in one
// accumulator in k order (no split-K).
#include "half_common.hpp"

namespace l2q {

constexpr int kSmBM = 32, kSmNT = 256, kSmKMax = 256;

template <typename HT, typename CT>
__global__ __launch_bounds__(kSmNT, 2) void gemm_small_h_kernel(const HT* __restrict__ A, const HT* __restrict__ W,
                                                                int M, int N, int K, EpiH epi, CT* __restrict__ C) {
  using vec_t = typename MfmaH<HT>::vec_t;
  __shared__ __attribute__((aligned(16))) char as[kSmBM * kSmKMax * 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, grp = lane >> 4;
  const long m0 = (long)blockIdx.x * kSmBM;
  const int rowb = K * 2, cpr = K >> 3;            // bytes / 16-byte chunks per A row
  // chunk c of row r sits at slot c ^ (r & swm), swm + 1 = the largest power of two <= 16 that divides cpr (the XOR
  // must stay inside the row): conflict-free ds_read_b128 fragments for K = 128, 256; fewer bank groups otherwise
  const int swm = ((cpr & -cpr) < 16 ? (cpr & -cpr) : 16) - 1;
  // ---- A tile: 32 rows x cpr chunks, thread t carries chunks t, t + 256, ... (at most 4 at K = 256)
  const int nchunk = kSmBM * cpr;
  uint4 z0 = make_uint4(0, 0, 0, 0), z1 = z0, z2 = z0, z3 = z0;
  auto ld = [&](int p) {
    if (p >= nchunk) p = nchunk - 1;
    const int row = p / cpr, c = p - row * cpr;
    long m = m0 + row;
    if (m >= M) m = M - 1;
    return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(A) + m * (long)rowb + (c << 4));
  };
  z0 = ld(tid); z1 = ld(tid + 256); z2 = ld(tid + 512); z3 = ld(tid + 768);
  // ---- W operands: lane (row wn + 16 j + l15, k-group grp) holds k = 32 kk + 8 grp .. + 7
  const int wn = wave * 64;
  long wrow[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = wn + 16 * j + l15;
    wrow[j] = (long)(n < N ? n : N - 1) * K + 8 * grp;
  }
  const int ksteps = K >> 5;
  vec_t rw[2][4];
  auto fetch_w = [&](int e, int kk) {
    if (kk >= ksteps) kk = ksteps - 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) rw[e][j] = *reinterpret_cast<const vec_t*>(W + wrow[j] + 32 * kk);
  };
  fetch_w(0, 0);
  fetch_w(1, 1);
  auto st = [&](int p, const uint4& v) {
    if (p < nchunk) {
      const int row = p / cpr, c = p - row * cpr;
      *reinterpret_cast<uint4*>(as + row * rowb + ((c ^ (row & swm)) << 4)) = v;
    }
  };
  st(tid, z0); st(tid + 256, z1); st(tid + 512, z2); st(tid + 768, z3);
  __syncthreads();

  v4f32 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (v4f32){0.f, 0.f, 0.f, 0.f};
  const int fsw = l15 & swm;
  for (int kb = 0; kb < ksteps; kb += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int kk = kb + u;
      if (kk < ksteps) {
        const int chunk = (4 * kk + grp) ^ fsw;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const vec_t fa = *reinterpret_cast<const vec_t*>(as + (l15 + 16 * i) * rowb + (chunk << 4));
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = MfmaH<HT>::run(rw[u][j], fa, acc[i][j]);
        }
      }
      fetch_w(u, kk + 2);
    }
  }
  // W was the MFMA row operand: lane owns chain 16 i + l15 and outputs 16 j + 4 grp + r
  const bool vecc = (N % 4) == 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long nb4 = wn + 16 * j + 4 * grp;
    if (nb4 >= N) continue;
    float cs[4], cb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long n = nb4 + r < N ? nb4 + r : N - 1;
      cs[r] = epi.coeff ? epi.scale * expf(epi.coeff[n]) : epi.scale;
      cb[r] = 0.f;
      if (epi.bias) cb[r] += epi.bias[n];
      if (epi.bias2) cb[r] += epi.bias2[n];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long m = m0 + 16 * i + l15;
      if (m >= M) continue;
      typedef CT cv __attribute__((ext_vector_type(4)));
      cv o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (CT)epilogue_h<HT>(acc[i][j][r], cb[r], cs[r], epi.coeff != nullptr, epi.act);
      CT* dst = C + m * N + nb4;
      if (vecc) {
        *reinterpret_cast<cv*>(dst) = o;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (nb4 + r < N) dst[r] = o[r];
      }
    }
  }
}

// true: launched.  false: not this kernel's case
template <typename HT>
bool gemm_h_small_launch(const void* A, const void* W, int M, int N, long K, const EpiH& epi, void* C, int c_is_f32,
                         hipStream_t st) {
  if (K > kSmKMax || K < 32 || K % 32 != 0 || N > 256 || N < 16 || M < 2048) return false;
  if (!al16(A) || !al16(W) || !al16(C)) return false;
  const dim3 grid((unsigned)cdiv(M, kSmBM)), block(kSmNT);
  if (c_is_f32)
    hipLaunchKernelGGL((gemm_small_h_kernel<HT, float>), grid, block, 0, st, (const HT*)A, (const HT*)W, M, N, (int)K,
                       epi, (float*)C);
  else
    hipLaunchKernelGGL((gemm_small_h_kernel<HT, HT>), grid, block, 0, st, (const HT*)A, (const HT*)W, M, N, (int)K,
                       epi, (HT*)C);
  return true;
}

template bool gemm_h_small_launch<_Float16>(const void*, const void*, int, int, long, const EpiH&, void*, int, hipStream_t);
template bool gemm_h_small_launch<__bf16>(const void*, const void*, int, int, long, const EpiH&, void*, int, hipStream_t);

}  // namespace l2q
