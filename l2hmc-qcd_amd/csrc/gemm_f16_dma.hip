// gemm_f16_dma.hip -- the large half-precision Linear layers on LDS-DMA staging (gfx950):
//
//   C[M][N] = epi( A[M][K] . W[N][K]^T + bias )      A, W 16-bit, K-contiguous, fp32 accumulate
//
// for M, N multiples of 256 and K a multiple of 64 -- the conv stack's Linear of the U(1) networks
// (network.py:283-326; 8192 x 8192 x 51 200 at BASELINE cfg-3) and the other big dense layers.
// Same rounding points as gemm_nt_h_kernel (half_common.hpp); the accumulation order over K differs
// from the register-staged kernel only in that there is no split-K here.
//
// 256 x 256 tile, 512 threads = 8 wavefronts (4 x 2), wavefront tile 64 x 128 on
// v_mfma_f32_16x16x32_{f16,bf16}: 32 accumulator tiles.  A K-slab is 64 halves = ONE 128-byte row
// per tile row: global_load_lds_dwordx4 writes it into LDS with the 16-byte chunks XOR-swizzled by
// (row >> 1) & 7 (applied to the SOURCE address; 16 consecutive rows of a fragment read then hit 16
// different bank groups), two 64 KB stages, one barrier per K-slab, no address arithmetic in the
// loop.  Blocks are ordered so that the 32 tiles an XCD runs at a time form an 8 x 4 patch: 12 operand
// panels per K-slab through its L2 instead of 64.
//
// MI355X, 8192 x 8192 x 51 200 fp16 (6.87 TFLOP): 5.3-5.4 ms = 1.27-1.30 PFLOP/s (0.51-0.52 of the 2.5 PFLOP/s
// dense peak); the register-staged 128 x 256 kernel: 8.7 ms (0.31).  178 VGPRs, no spills, two
// wavefronts per SIMD; per K-slab a CU reads 192 KB of fragments from LDS and the DMA writes 64 KB
// (~2000 of the slab's ~3900 cycles of LDS time): wider wavefront tiles are the next step.  hipcc batches the
// fragment reads in front of each group of 16 MFMAs behind an s_waitcnt lgkmcnt(0) (four exposed LDS
// latencies per slab, covered only by the SIMD's other wavefront); requesting both K-steps' fragments up
// front (sched_barrier) ends in ONE lgkmcnt(0) in front of all 64 MFMAs -- it does not emit the partial
// count -- so that variant was not kept.  PMC (profiles/r02j_pmc_gemm_h_dma.txt): no LDS bank conflicts,
// L2 hit rate 0.81 (10.1 GB through the fabric for 1.68 GB of operands), 48 % of the wavefront cycles
// waiting and only 3 % of them on LDS: the wait is the vmcnt(0) + barrier for the NEXT slab, one slab
// (~1.7 us) being a short prefetch distance for the fifth of the requests that miss L2.  A third stage
// needs a smaller stage.  Measured (HD_BN = 128, HD_STAGES = 3: 256 x 128 tiles, three 48 KB stages, two
// slabs in flight): 6.75 ms against 5.19 ms -- the 1.5x operand traffic and the doubled barrier rate cost more
// than the longer prefetch distance buys; 32-wide K-slabs in a four-deep ring (HD_BK = 32, HD_STAGES = 4: three
// slabs in flight, 64-byte rows swizzled by (row >> 2) & 3): 6.42 ms.  Both deeper rings lose to the doubled
// barrier rate: 64-wide slabs, two stages it is.
#include "half_common.hpp"

namespace l2q {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
extern __shared__ __attribute__((aligned(1024))) char hd_lds[];

#ifndef HD_BN
#define HD_BN 256          // N-tile width and ring depth (A/B builds: tools/ab_build.sh ... -DHD_BN=128 -DHD_STAGES=3)
#define HD_STAGES 2
#endif
#ifndef HD_SPREAD
#define HD_SPREAD 0        // LDS-DMA pieces of the next slab: 0 back to back at the top of the slab; 1 one per MFMA group
                           // over the whole slab (6.28 ms against 5.76-5.83: the late pieces have not landed at the
                           // next barrier); 2 two per group over the first K-step (5.75: within noise).  Round 3.
#endif
#ifndef HD_BK
#define HD_BK 64           // K-slab in halves: 64 (128-byte rows, 8 chunks) or 32 (64-byte rows, 4 chunks)
#endif
constexpr int kHdBM = 256, kHdBN = HD_BN, kHdBK = HD_BK, kHdStages = HD_STAGES;
constexpr int kHdRow = kHdBK * 2;                  // bytes of a tile row in LDS
constexpr int kHdRpi = 1024 / kHdRow;              // tile rows one wavefront DMA instruction fills
constexpr int kHdOp = kHdBM * kHdRow;              // bytes of the A tile per stage
constexpr int kHdStage = kHdOp + kHdBN * kHdRow;
// chunk swizzle key of a row: 16 consecutive rows of a fragment read must cover 16 bank groups
__host__ __device__ constexpr int hd_key(int row) { return kHdBK == 64 ? (row >> 1) & 7 : (row >> 2) & 3; }

// wavefront grid over the tile: 4 x 2 (wavefront tile 64 x 128) 5.30 ms, 2 x 4 5.43 ms; 2 x 2 with the
// 256 accumulator registers in AGPRs and one wavefront per SIMD: 34 ms (nothing hides the LDS latency)
constexpr int kHdWM = 4, kHdWN = 2, kHdWaves = kHdWM * kHdWN;

template <typename HT, typename CT>
__global__ __launch_bounds__(64 * kHdWaves, kHdWaves / 4) void gemm_h_dma_kernel(const HT* __restrict__ A,
                                                                   const HT* __restrict__ W, int M, int N,
                                                                   long K, EpiH epi, CT* __restrict__ C,
                                                                   int patched) {
  using vec_t = typename MfmaH<HT>::vec_t;
  constexpr int MI = kHdBM / (16 * kHdWM), NI = kHdBN / (16 * kHdWN);
  constexpr int GA = kHdBM / kHdRpi, GW = kHdBN / kHdRpi, LQ = (GA + GW) / kHdWaves, LQA = GA / kHdWaves;   // loader instructions
  constexpr int CPR = kHdRow / 16, KS = kHdBK / 32;             // 16-byte chunks per row, MFMA K-steps per slab
  const int tid = threadIdx.x, lane = tid & 63, grp = lane >> 4, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave / kHdWN) * (16 * MI), wn = (wave % kHdWN) * (16 * NI);
  const long tm = M / kHdBM, tn = N / kHdBN, total = tm * tn;
  long w = xcd_swizzle(blockIdx.x, total, 1);
  long bm, bn;
  if (patched) {                                   // 8 x 4 patches of tiles, m fastest inside a patch
    const long patch = w >> 5, in = w & 31, ppr = tm >> 3;
    bm = (patch % ppr) * 8 + (in & 7);
    bn = (patch / ppr) * 4 + (in >> 3);
  } else {
    bm = w % tm; bn = w / tm;
  }
  const long m0 = bm * kHdBM, n0 = bn * kHdBN;
  // loader: wavefront instruction g = 8 q + wave (q < 4: A, else W) fills tile rows (g & 31) * 8 +
  // (lane >> 3), 16-byte position lane & 7 with source chunk (lane & 7) ^ ((row >> 1) & 7)
  unsigned vo[LQ];
#pragma unroll
  for (int q = 0; q < LQ; ++q) {
    const int gq = kHdWaves * q + wave;
    const int R = (gq < GA ? gq : gq - GA) * kHdRpi + lane / CPR;
    const int c = (lane % CPR) ^ hd_key(R);
    vo[q] = (unsigned)(R * K * 2 + c * 16);
  }
  const char* a1 = reinterpret_cast<const char*>(A) + m0 * K * 2;
  const char* w1 = reinterpret_cast<const char*>(W) + n0 * K * 2;
  auto issue = [&](int stage, long k0) {
#pragma unroll
    for (int q = 0; q < LQ; ++q) {
      const int g = kHdWaves * q + wave;
      const char* src = (q < LQA ? a1 : w1) + k0 * 2 + (unsigned long)vo[q];
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (lds_ptr_t)(hd_lds + stage * kHdStage + g * 1024), 16, 0, 0);
    }
  };
  unsigned offA[KS], offB[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const unsigned sw = (unsigned)(((4 * s + grp) ^ hd_key(l15)) << 4);
    offA[s] = (wm + l15) * kHdRow + sw;
    offB[s] = kHdOp + (wn + l15) * kHdRow + sw;
  }
  v4f32 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (v4f32){0, 0, 0, 0};

  // ring of kHdStages stages: slabs k + 1 .. k + kHdStages - 1 are in flight while slab k is computed
  const long nslab = K / kHdBK;
#pragma unroll
  for (int p = 0; p < kHdStages - 1; ++p)
    if (p < nslab) issue(p, (long)p * kHdBK);
  int st = 0;
  for (long ks = 0; ks < nslab; ++ks) {
    // slab ks has landed when at most the younger slabs' requests are outstanding
    if (kHdStages > 2 && ks + kHdStages - 2 < nslab)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kHdStages - 2) * LQ) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const long kn = ks + kHdStages - 1;
    int sn = st + kHdStages - 1;
    if (sn >= kHdStages) sn -= kHdStages;
    // past the last slab the last one is fetched again (into the stage nobody reads any more): no branch
    const long kq = (kn < nslab ? kn : nslab - 1) * kHdBK;
    if (!HD_SPREAD) {
      if (kn < nslab) issue(sn, kn * kHdBK);
    }
    const char* sb = hd_lds + st * kHdStage;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      vec_t fb[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[j] = *reinterpret_cast<const vec_t*>(sb + offB[s] + j * 16 * kHdRow);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const vec_t fa = *reinterpret_cast<const vec_t*>(sb + offA[s] + i * 16 * kHdRow);
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = MfmaH<HT>::run(fb[j], fa, acc[i][j]);
        if (HD_SPREAD) {
          // one piece behind each group of NI MFMAs: a piece holds the wavefront's issue for ~76 cycles;
          // back to back at the top of the slab (and on both wavefronts of a SIMD at once, right after
          // the barrier) they leave the matrix pipe idle
#pragma unroll
          for (int q = (HD_SPREAD == 2 ? (s == 0 ? i * LQ / MI : LQ) : (s * MI + i) * LQ / (KS * MI));
               q < (HD_SPREAD == 2 ? (s == 0 ? (i + 1) * LQ / MI : LQ) : (s * MI + i + 1) * LQ / (KS * MI)); ++q) {
            __builtin_amdgcn_sched_barrier(0);
            const int g = kHdWaves * q + wave;
            const char* src = (q < LQA ? a1 : w1) + kq * 2 + (unsigned long)vo[q];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (lds_ptr_t)(hd_lds + sn * kHdStage + g * 1024), 16, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    if (++st == kHdStages) st = 0;
  }
  // W was the MFMA row operand: lane owns chain m = l15 of tile i, outputs 4 grp + r of tile j
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const long nb4 = n0 + wn + 16 * j + 4 * grp;
    float cs[4], cb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long n = nb4 + r;
      cs[r] = epi.coeff ? epi.scale * expf(epi.coeff[n]) : epi.scale;
      cb[r] = 0.f;
      if (epi.bias) cb[r] += epi.bias[n];
      if (epi.bias2) cb[r] += epi.bias2[n];
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const long m = m0 + wm + 16 * i + l15;
      typedef CT cv __attribute__((ext_vector_type(4)));
      cv o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (CT)epilogue_h<HT>(acc[i][j][r], cb[r], cs[r], epi.coeff != nullptr, epi.act);
      *reinterpret_cast<cv*>(C + m * N + nb4) = o;
    }
  }
}

// true: launched.  false: the shape does not fit (the caller uses gemm_nt_h_kernel)
template <typename HT>
bool gemm_h_dma_launch(const void* A, const void* W, int M, int N, long K, const EpiH& epi, void* C,
                       int c_is_f32, hipStream_t st) {
  if (M % kHdBM || N % kHdBN || K % kHdBK || K < 4 * kHdBK) return false;
  if ((long)(M / kHdBM) * (N / kHdBN) < 128) return false;  // few tiles: the split-K kernels fill the chip
  if (!al16(A) || !al16(W) || !al16(C) || (double)kHdBM * K * 2.0 >= 4.0e9) return false;
  const long tm = M / kHdBM, tn = N / kHdBN;
  const int patched = (tm % 8 == 0 && tn % 4 == 0) ? 1 : 0;     // (row-major tile order: 5.72 instead of 5.46 ms)
  const dim3 grid((unsigned)(tm * tn)), block(64 * kHdWaves);
  const size_t lds = (size_t)kHdStages * kHdStage;
#define L2Q_HD(CTV)                                                                                  \
  do {                                                                                               \
    static PerDeviceOnce attr_once;                                                                  \
    if (attr_once.first()) {                                                                                 \
      (void)hipFuncSetAttribute((const void*)gemm_h_dma_kernel<HT, CTV>,                             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
    }                                                                                                \
    hipLaunchKernelGGL((gemm_h_dma_kernel<HT, CTV>), grid, block, lds, st, (const HT*)A, (const HT*)W, M, \
                       N, K, epi, (CTV*)C, patched);                                                 \
  } while (0)
  if (c_is_f32) L2Q_HD(float);
  else L2Q_HD(HT);
#undef L2Q_HD
  return true;
}

template bool gemm_h_dma_launch<_Float16>(const void*, const void*, int, int, long, const EpiH&, void*, int,
                                          hipStream_t);
template bool gemm_h_dma_launch<__bf16>(const void*, const void*, int, int, long, const EpiH&, void*, int,
                                        hipStream_t);

}  // namespace l2q
