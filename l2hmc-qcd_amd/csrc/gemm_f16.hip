// gemm_f16.hip -- the dense layers of the leapfrog networks in half precision: BASELINE cfg-3
// "fp16 nets / fp32 action" (reference: torch.autocast around Dynamics.forward,
// trainers/pytorch/trainer.py:211-219 -- nn.Linear runs in fp16 / bf16, everything else in fp32).
//
//   C[M][N] = epi( A[M][K] . W[N][K]^T  (+ A2[M][K2] . W2[N][K2]^T) + bias (+ bias2) )
//
// W, W2 are 16-bit (converted once per weight version on the host side), A / A2 are either the
// 16-bit output of the previous layer or fp32 lattice data (cos/sin of the links, momenta) which
// the tile loader rounds on its way into LDS -- there is no separate cast pass over HBM.
// Products accumulate in fp32 on v_mfma_f32_16x16x32_{f16,bf16} (gfx950's double-K form).
// Rounding points follow autocast: the Linear output (accumulator + bias) is rounded to 16 bit,
// the activation is evaluated on that value and rounded again; exp(coeff) and the net-weight
// scale are applied in fp32 (torch promotes fp32 tensor x fp16 tensor to fp32).  The input
// layer's two products share one accumulator here (autocast rounds xlayer(x) and vlayer(v)
// separately before adding: one rounding fewer, <= 2^-11 relative).
#include <type_traits>
#include "half_common.hpp"
#include "u1_math.hpp"
#include "heads_h_common.hpp"

namespace l2q {

// ROWS x HBK tile of the virtual K-concatenated matrix [P | P2] -> registers -> LDS (as HT).
// S: element type in HBM (HT or float).  vec: every 16-byte vector is aligned and inside one
// segment (K, K2 multiples of the vector length); otherwise element-wise loads.
template <typename HT, typename S, int ROWS = 128, int NT = kBlock>
struct TileH {
  static constexpr int VEC = 16 / sizeof(S);          // 8 halves or 4 floats
  static constexpr int VPR = HBK / VEC;               // vectors per row: 8 or 16
  static constexpr int RPP = NT / VPR;                // rows per pass: 32 or 16 (256 threads)
  static constexpr int NP = ROWS / RPP;               // passes: 4 or 8 for 128 rows
  typedef HT hv __attribute__((ext_vector_type(VEC)));
  hv reg[NP];                                         // packed: VEC/2 VGPRs per vector

  // cs_mask != nullptr (fp32 sources only): the first operand is VIRTUAL -- column kk < K / 2
  // is cos(keep[kk] p[row][kk]), column K / 2 + kk is sin(.), keep = mask or 1 - mask: the U(1)
  // xnet's [cos(m x), sin(m x)] input (dynamics.py:1161-1185) formed inside the tile loader from
  // the link angles, instead of by a kernel that writes 2 x the field to HBM for this GEMM to read
  // back.  p then has row stride K / 2.  (v_sin / v_cos: |argument| <= pi, error ~1e-6, far inside
  // the 16-bit rounding that follows.)
  __device__ __forceinline__ void fetch(const S* __restrict__ p, const S* __restrict__ p2,
                                        long row0, long nrows, long k0, long K, long K2,
                                        long kend, bool vec, const float* __restrict__ cs_mask = nullptr,
                                        int cs_compl = 0) {
    const int tid = threadIdx.x;
    const int kv = (tid % VPR) * VEC, r = tid / VPR;
    const long kk = k0 + kv;
    if (sizeof(S) == 4 && cs_mask != nullptr && kk < K) {
      const long xd = K >> 1;
      const bool is_sin = kk >= xd;
      const long col = is_sin ? kk - xd : kk;
      float keep[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float m = col + j < xd ? cs_mask[col + j] : 0.f;
        keep[j] = cs_compl ? 1.f - m : m;
      }
      const bool kin = kk < kend;
      const bool whole = vec && (xd % VEC) == 0;         // vector entirely inside one half
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const long row = row0 + r + (long)i * RPP;
        if (whole) {
          typedef S sv __attribute__((ext_vector_type(VEC)));
          sv xv;
#pragma unroll
          for (int j = 0; j < VEC; ++j) xv[j] = (S)0;
          if (kin && row < nrows) xv = *reinterpret_cast<const sv*>(p + row * xd + col);
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const float a = keep[j] * (float)xv[j];
            reg[i][j] = (HT)((kin && row < nrows) ? (is_sin ? __sinf(a) : __cosf(a)) : 0.f);
          }
          continue;
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float v = 0.f;
          if (kin && row < nrows && col + j < xd) {
            const float a = keep[j] * (float)p[row * xd + col + j];
            v = is_sin ? __sinf(a) : __cosf(a);
          } else if (kin && row < nrows && !is_sin && col + j >= xd) {
            // a vector straddling the cos | sin boundary (xdim not a multiple of VEC)
            const long c2 = col + j - xd;
            const float m2 = cs_mask[c2];
            v = __sinf((cs_compl ? 1.f - m2 : m2) * (float)p[row * xd + c2]);
          }
          reg[i][j] = (HT)v;
        }
      }
      return;
    }
    if (vec) {
      const S* base = p;
      long ld = K, kc = kk;
      if (kk >= K) { base = p2; ld = K2; kc = kk - K; }
      const bool kin = kk < kend;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const long row = row0 + r + (long)i * RPP;
        if (kin && row < nrows) {
          typedef S sv __attribute__((ext_vector_type(VEC)));
          const sv v = *reinterpret_cast<const sv*>(base + row * ld + kc);
#pragma unroll
          for (int j = 0; j < VEC; ++j) reg[i][j] = (HT)v[j];
        } else {
#pragma unroll
          for (int j = 0; j < VEC; ++j) reg[i][j] = (HT)0.f;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const long row = row0 + r + (long)i * RPP;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const long k = kk + j;
          float v = 0.f;
          if (row < nrows && k < kend) {
            if (k < K) v = (float)p[row * K + k];
            else if (k - K < K2) v = (float)p2[row * K2 + (k - K)];
          }
          reg[i][j] = (HT)v;
        }
      }
    }
  }

  __device__ __forceinline__ void store(HT (*lds)[HLD]) const {
    const int tid = threadIdx.x;
    const int kv = (tid % VPR) * VEC, r = tid / VPR;
#pragma unroll
    for (int i = 0; i < NP; ++i) *reinterpret_cast<hv*>(&lds[r + i * RPP][kv]) = reg[i];
  }
};

// grid x = N tiles, y = M tiles, z = K splits.  FUSED: splits == 1, epilogue applied here;
// otherwise raw fp32 partial sums go to part[z][M][N].  AS: element type of A / A2 in HBM,
// CT: element type of C.
// NT = 256: 2 x 2 wavefronts, tile 128 x 128.  NT = 512: 2 x 4 wavefronts, tile 128 x 256 -- for
// the wide-K input layer, where N = units[0] is a few hundred: every A element (fp32 lattice
// data, the bulk of the traffic) is then read by one workgroup only.
template <typename HT, typename AS, typename CT, bool FUSED, int NT>
__global__ __launch_bounds__(NT, NT == 256 ? 2 : 1) void gemm_nt_h_kernel(
    const AS* __restrict__ A, const HT* __restrict__ W, const AS* __restrict__ A2,
    const HT* __restrict__ W2, int M, int N, long K, long K2, long kchunk, int splits, EpiH epi,
    int veca, int vecw, CT* __restrict__ C, float* __restrict__ part, int patch,
    const float* __restrict__ cs_mask, int cs_compl) {
  using vec_t = typename MfmaH<HT>::vec_t;
  constexpr int BN = NT / 2, WN = NT / 128;            // tile width, wavefronts along N
  __shared__ __attribute__((aligned(16))) HT As[128][HLD];
  __shared__ __attribute__((aligned(16))) HT Ws[BN][HLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave / WN) * 64, wn = (wave % WN) * 64;
  // 1-D grid.  Hardware block b runs on XCD b % 8; the N-tiles of one (M-tile, K-split) read the
  // same A tile, so they are made neighbours in ONE XCD's queue (the second read hits that
  // XCD's L2) instead of neighbours in launch order (which lands them on different XCDs).
  const long ntn = (N + BN - 1) / BN, ntm = (M + 127) / 128;
  const long nmz = ntm * splits;
  long mz, nt;
  if (splits == 1 && patch && ntm % 8 == 0 && ntn % 8 == 0 && ((ntm / 8) * (ntn / 8)) % kXcds == 0) {
    // Big layers stream both operands from HBM / L2, so the ~64 workgroups an XCD runs at a time
    // should share as many operand tiles as possible: they are made an 8 x 8 patch of tiles
    // (8 A tiles + 8 W tiles for 64 workgroups instead of 1 + 64).
    const long xcd = blockIdx.x % kXcds, seq = blockIdx.x / kXcds;
    const long p = (seq / 64) * kXcds + xcd, local = seq % 64;
    const long pn = ntn / 8;
    mz = (p / pn) * 8 + local / 8;
    nt = (p % pn) * 8 + local % 8;
  } else if (nmz % kXcds == 0) {
    const long xcd = blockIdx.x % kXcds, seq = blockIdx.x / kXcds;
    mz = (seq / ntn) * kXcds + xcd;
    nt = seq % ntn;
  } else {
    mz = blockIdx.x / ntn;
    nt = blockIdx.x % ntn;
  }
  const long zsplit = mz / ntm;
  const long m0 = (mz % ntm) * 128, n0 = nt * BN;
  const long Kt = K + K2;
  const long kbeg = zsplit * kchunk;
  long kend = kbeg + kchunk;
  if (kend > Kt) kend = Kt;

  // wavefront tile 64 x 64 = 2 x 2 MFMA tiles of 32 x 32 (64 fp32 accumulators / lane)
  v16f32 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  TileH<HT, AS, 128, NT> la;
  TileH<HT, HT, BN, NT> lw;
  la.fetch(A, A2, m0, M, kbeg, K, K2, kend, veca != 0, cs_mask, cs_compl);
  lw.fetch(W, W2, n0, N, kbeg, K, K2, kend, vecw != 0);
  for (long k0 = kbeg; k0 < kend; k0 += HBK) {
    __syncthreads();
    la.store(As);
    lw.store(Ws);
    __syncthreads();
    if (k0 + HBK < kend) {
      la.fetch(A, A2, m0, M, k0 + HBK, K, K2, kend, veca != 0, cs_mask, cs_compl);
      lw.fetch(W, W2, n0, N, k0 + HBK, K, K2, kend, vecw != 0);
    }
#pragma unroll
    for (int ks = 0; ks < HBK; ks += 16) {
      vec_t fa[2], fb[2];
      const int kq = ks + 8 * (lane >> 5);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i] = *reinterpret_cast<const vec_t*>(&As[wm + 32 * i + (lane & 31)][kq]);
        fb[i] = *reinterpret_cast<const vec_t*>(&Ws[wn + 32 * i + (lane & 31)][kq]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = MfmaH<HT>::run32(fb[j], fa[i], acc[i][j]);
    }
  }

  // W was the MFMA "row" operand: lane holds row m = lane & 31 of tile i and, per register quad q,
  // the four consecutive columns n = 32 j + 8 q + 4 (lane >> 5) + (r & 3): 8- / 16-byte stores.
  const bool vecc = (N % 4) == 0;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long nb4 = n0 + wn + 32 * j + 8 * q + 4 * (lane >> 5);
      if (nb4 >= N) continue;
      float cs[4], cb[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long n = nb4 + r < N ? nb4 + r : N - 1;
        cs[r] = 1.f; cb[r] = 0.f;
        if (FUSED) {
          cs[r] = epi.coeff ? epi.scale * expf(epi.coeff[n]) : epi.scale;
          if (epi.bias) cb[r] += epi.bias[n];
          if (epi.bias2) cb[r] += epi.bias2[n];
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const long m = m0 + wm + 32 * i + (lane & 31);
        if (m >= M) continue;
        if (FUSED) {
          typedef CT cv __attribute__((ext_vector_type(4)));
          cv o;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            o[r] = (CT)epilogue_h<HT>(acc[i][j][4 * q + r], cb[r], cs[r], epi.coeff != nullptr, epi.act);
          CT* dst = C + m * N + nb4;
          if (vecc) *reinterpret_cast<cv*>(dst) = o;
          else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (nb4 + r < N) dst[r] = o[r];
          }
        } else {
          float* dst = part + zsplit * M * N + m * N + nb4;
          if (vecc) {
            *reinterpret_cast<v4f32*>(dst) = (v4f32){acc[i][j][4 * q], acc[i][j][4 * q + 1],
                                                     acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (nb4 + r < N) dst[r] = acc[i][j][4 * q + r];
          }
        }
      }
    }
}

template <typename HT, typename CT>
__global__ __launch_bounds__(kBlock) void splitk_reduce_h_kernel(const float* __restrict__ part,
                                                                 int splits, long MN, int N,
                                                                 EpiH epi, CT* __restrict__ C) {
  const long i = (long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= MN) return;
  float s = 0.f;
  for (int z = 0; z < splits; ++z) s += part[(long)z * MN + i];     // fixed order
  const int n = (int)(i % N);
  const float cs = epi.coeff ? epi.scale * expf(epi.coeff[n]) : epi.scale;
  float cb = 0.f;
  if (epi.bias) cb += epi.bias[n];
  if (epi.bias2) cb += epi.bias2[n];
  C[i] = (CT)epilogue_h<HT>(s, cb, cs, epi.coeff != nullptr, epi.act);
}

// four consecutive outputs per thread (N % 4 == 0: one row): 16-byte loads of every partial, one 8- / 16-byte store
template <typename HT, typename CT>
__global__ __launch_bounds__(kBlock) void splitk_reduce4_h_kernel(const float* __restrict__ part, int splits, long MN,
                                                                  int N, EpiH epi, CT* __restrict__ C) {
  const long i = ((long)blockIdx.x * kBlock + threadIdx.x) * 4;
  if (i >= MN) return;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < splits; ++z) {                               // fixed order
    const float4 p = *reinterpret_cast<const float4*>(part + (long)z * MN + i);
    s[0] += p.x; s[1] += p.y; s[2] += p.z; s[3] += p.w;
  }
  const int n = (int)(i % N);
  typedef CT cv __attribute__((ext_vector_type(4)));
  cv o;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float cs = epi.coeff ? epi.scale * expf(epi.coeff[n + r]) : epi.scale;
    float cb = 0.f;
    if (epi.bias) cb += epi.bias[n + r];
    if (epi.bias2) cb += epi.bias2[n + r];
    o[r] = (CT)epilogue_h<HT>(s[r], cb, cs, epi.coeff != nullptr, epi.act);
  }
  *reinterpret_cast<cv*>(C + i) = o;
}

template <typename HT, typename CT>
static void launch_splitk_reduce_h(const float* part, int splits, long MN, int N, const EpiH& epi, CT* C, hipStream_t st) {
  if (N % 4 == 0 && (reinterpret_cast<uintptr_t>(part) & 15) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0)
    hipLaunchKernelGGL((splitk_reduce4_h_kernel<HT, CT>), dim3((unsigned)cdiv(MN / 4, kBlock)), dim3(kBlock), 0, st, part,
                       splits, MN, N, epi, C);
  else
    hipLaunchKernelGGL((splitk_reduce_h_kernel<HT, CT>), dim3((unsigned)cdiv(MN, kBlock)), dim3(kBlock), 0, st, part,
                       splits, MN, N, epi, C);
}

static int pick_splits_h(int M, int N, long Kt) {
  const long tiles = cdiv(M, 128) * cdiv(N, 128);
  if (tiles >= 256 || Kt <= 8 * HBK) return 1;
  long s = cdiv(512, tiles);
  const long maxs = Kt / (4 * HBK) > 0 ? Kt / (4 * HBK) : 1;
  if (s > maxs) s = maxs;
  if (s > 64) s = 64;
  return (int)(s < 1 ? 1 : s);
}

// wide: the 128 x 256 tile (512 threads, one workgroup per CU), always through split-K partials.
static int pick_config_h(int M, int N, long Kt, bool* wide) {
  *wide = N > 128 && Kt >= 32 * HBK && cdiv(M, 128) * cdiv(N, 128) < 256;
  if (!*wide && tuning().gemm_h_wide_fused && N >= 1024 && M >= 1024 && Kt >= 16 * HBK) {
    *wide = true;                  // large square-ish layer: 128 x 256 tiles, epilogue in-kernel
    return 1;
  }
  if (!*wide) return pick_splits_h(M, N, Kt);
  const long tiles = cdiv(M, 128) * cdiv(N, 256);
  long s = cdiv(256, tiles);
  const long maxs = Kt / (8 * HBK);
  if (s > maxs) s = maxs;
  if (s > 64) s = 64;
  return (int)(s < 1 ? 1 : s);
}


// gemm_f16_skinny.hip: the streaming kernel of the wide-K fp32-operand input layer (N <= 256); false -> not its case
int gemm_h_skinny_splits(int M, int N, long raw1, long K2, int want);
size_t gemm_h_skinny_ws_bytes(int M, int N, long K, long K2);
template <typename HT>
bool gemm_h_skinny_launch(const float* A, const void* W, int M, int N, long K, const float* A2,
                          const void* W2, long K2, void* ws, size_t ws_bytes, hipStream_t st,
                          const float* cs_mask, int cs_compl, int want, int* splits_out);

template <typename HT, typename AS, typename CT>
static int gemm_h_launch(const void* A_, const void* W_, int M, int N, long K, const void* A2_,
                         const void* W2_, long K2, EpiH epi, void* C_, void* ws, size_t ws_bytes,
                         hipStream_t st, const float* cs_mask = nullptr, int cs_compl = 0) {
  const AS* A = (const AS*)A_;
  const AS* A2 = (const AS*)A2_;
  const HT* W = (const HT*)W_;
  const HT* W2 = (const HT*)W2_;
  CT* C = (CT*)C_;
  const long Kt = K + K2;
  if constexpr (std::is_same<AS, float>::value) {
    const int sk = tuning().gemm_h_skinny;
    int S = 0;
    if (sk != 0 && gemm_h_skinny_launch<HT>((const float*)A_, W_, M, N, K, (const float*)A2_, W2_, K2, ws,
                                            ws_bytes, st, cs_mask, cs_compl, sk == 1 ? 0 : sk, &S)) {
      launch_splitk_reduce_h<HT, CT>((const float*)ws, S, (long)M * N, N, epi, C, st);
      return check_launch("l2q_gemm_h");
    }
  }
  bool wide = false;
  int splits = pick_config_h(M, N, Kt, &wide);
  long kchunk = cdiv(cdiv(Kt, splits), HBK) * HBK;
  splits = (int)cdiv(Kt, kchunk);
  constexpr long VA = 16 / sizeof(AS);
  const int veca = K % VA == 0 && K2 % VA == 0 && al16(A) && al16(A2);
  const int vecw = K % 8 == 0 && K2 % 8 == 0 && al16(W) && al16(W2);
  const int patch = tuning().gemm_h_patch;
  const bool wide_fused = wide && splits == 1 && (long)M * N >= 1024L * 1024L;
  float* part = nullptr;
  if (splits > 1 || (wide && !wide_fused)) {
    const size_t need = (size_t)splits * M * N * sizeof(float);
    if (!ws || ws_bytes < need) {
      set_error("l2q_gemm_h: split-K workspace too small (%zu < %zu)", ws_bytes, need);
      return L2Q_ESHAPE;
    }
    part = (float*)ws;
  }
  if (wide_fused) {
    const dim3 grid((unsigned)(cdiv(N, 256) * cdiv(M, 128)));
    hipLaunchKernelGGL((gemm_nt_h_kernel<HT, AS, CT, true, 512>), grid, dim3(512), 0, st, A, W, A2,
                       W2, M, N, K, K2, kchunk, splits, epi, veca, vecw, C, part, patch, cs_mask, cs_compl);
  } else if (wide) {
    const dim3 grid((unsigned)(cdiv(N, 256) * cdiv(M, 128) * splits));
    hipLaunchKernelGGL((gemm_nt_h_kernel<HT, AS, CT, false, 512>), grid, dim3(512), 0, st, A, W, A2,
                       W2, M, N, K, K2, kchunk, splits, epi, veca, vecw, C, part, patch, cs_mask, cs_compl);
  } else if (splits == 1) {
    const dim3 grid((unsigned)(cdiv(N, 128) * cdiv(M, 128)));
    hipLaunchKernelGGL((gemm_nt_h_kernel<HT, AS, CT, true, 256>), grid, dim3(kBlock), 0, st, A, W,
                       A2, W2, M, N, K, K2, kchunk, splits, epi, veca, vecw, C, part, patch, cs_mask, cs_compl);
  } else {
    const dim3 grid((unsigned)(cdiv(N, 128) * cdiv(M, 128) * splits));
    hipLaunchKernelGGL((gemm_nt_h_kernel<HT, AS, CT, false, 256>), grid, dim3(kBlock), 0, st, A, W,
                       A2, W2, M, N, K, K2, kchunk, splits, epi, veca, vecw, C, part, patch, cs_mask, cs_compl);
  }
  if (splits > 1 || (wide && !wide_fused)) {
    launch_splitk_reduce_h<HT, CT>((const float*)part, splits, (long)M * N, N, epi, C, st);
  }
  return check_launch("l2q_gemm_h");
}

template <typename HT>
static int gemm_h_dispatch(const void* A, int a_f32, const void* W, int M, int N, long K,
                           const void* A2, const void* W2, long K2, EpiH epi, void* C, int c_f32,
                           void* ws, size_t ws_bytes, hipStream_t st,
                           const float* cs_mask = nullptr, int cs_compl = 0) {
  if (a_f32) {
    return c_f32 ? gemm_h_launch<HT, float, float>(A, W, M, N, K, A2, W2, K2, epi, C, ws, ws_bytes,
                                                   st, cs_mask, cs_compl)
                 : gemm_h_launch<HT, float, HT>(A, W, M, N, K, A2, W2, K2, epi, C, ws, ws_bytes, st,
                                                cs_mask, cs_compl);
  }
  return c_f32 ? gemm_h_launch<HT, HT, float>(A, W, M, N, K, A2, W2, K2, epi, C, ws, ws_bytes, st)
               : gemm_h_launch<HT, HT, HT>(A, W, M, N, K, A2, W2, K2, epi, C, ws, ws_bytes, st);
}


// ---------------------------------------------------------------------------------------
// The three output heads of a U(1) LeapfrogLayer AND the sub-update that consumes them, in one
// kernel: s, t, q ([chains][xdim] fp32 each -- 3 x 268 MB at cfg-3) never reach HBM.
//   s = cs[n] r16(tanh(r16(Z.Ws[n] + bs[n]))),  t = r16(ct r16(Z.Wt[n] + bt[n])),  q like s
//   XUPD = false: generalised momentum update (dynamics.py:1266-1297), a = v, b = force
//   XUPD = true : masked link update (dynamics.py:1386-1477), a = x, b = v, keep = mask[n]
// Tile 128 chains x 64 entries; each wavefront owns 64 x 32 = 4 x 2 MFMA tiles per head (96
// fp32 accumulators / lane); the three W tiles share the staged Z tile.  The per-chain logdet
// goes through per-(chain, column-group) partials and the fixed-order finalize.
// Epilogue math of the half-precision path on the hardware transcendentals (v_exp / v_log /
// v_sin / v_cos / v_rcp, ~1e-6 absolute): three orders of magnitude inside the 16-bit rounding
// the heads already carry, and ~4x fewer VALU cycles than the libm forms, which matter here
// because the epilogue (not the K = units[-1] MFMA loop) is the long part of this kernel.
// (HeadsHArgs, fast_exp / fast_tanh_h, hh_element: heads_h_common.hpp)

template <typename HT, bool XUPD, bool FWD, bool NCP, int BM>
__global__ __launch_bounds__(kBlock, BM == 128 ? 2 : 3) void u1_heads_update_h_kernel(HeadsHArgs a, int swz,
                                                                      int nfast, int stagger) {
  constexpr int BN = 64, MI = BM / 32;        // MI: 16-row MFMA tiles per wavefront along m
  using vec_t = typename MfmaH<HT>::vec_t;
  // tuning `heads_h_stagger` (A/B only): the g-th workgroup placed on a CU starts g * stagger * ~3.5 us
  // late, so that co-resident workgroups are in different phases (staging / MFMA / epilogue)
  if (stagger > 0 && blockIdx.x < 1024) {
    const int gen = blockIdx.x >> 8;
    for (int i = 0; i < gen * stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }
  __shared__ __attribute__((aligned(16))) HT Zs[BM][HLD];
  __shared__ __attribute__((aligned(16))) HT Ws[3][BN][HLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * 32;
  const long mt = (a.M + BM - 1) / BM;
  const long w = xcd_swizzle(blockIdx.x, gridDim.x, swz);
  const long nt = (a.N + BN - 1) / BN;
  const long m0 = nfast ? (w / nt) * BM : (w % mt) * BM;
  const long n0 = nfast ? (w % nt) * BN : (w / mt) * BN;
  const HT* Z = (const HT*)a.Z;
  const HT* W0 = (const HT*)a.W[0];
  const HT* W1 = (const HT*)a.W[1];
  const HT* W2 = (const HT*)a.W[2];
  const long K = a.K;
  const bool vec = (K % 8) == 0;

  v4f32 acc[3][MI][2];
#pragma unroll
  for (int h = 0; h < 3; ++h)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[h][i][j] = (v4f32){0, 0, 0, 0};

  // The lattice operands of the update (a = v or x, b = force or v) do not depend on the K-loop, and
  // the update is in place: once the first tile has been stored, hipcc keeps every later load of
  // `a` behind that store.  They are therefore requested up front, by ROW tile i (a wavefront's
  // 16 x 32 sub-tile is one whole 128-byte line per row when both column halves j = 0, 1 are
  // requested together): row tiles 0, 1 travel during the K-loop, 2, 3 are requested before the
  // first store of the epilogue.  Measured neutral at cfg-3 (508 vs 513 us per launch), and so was
  // an LDS-DMA K-loop on 64-chain tiles (506 vs 494 us, same box; not kept): neither the operand
  // latency of the epilogue nor the four global -> register -> LDS round trips of the K = 256
  // loop is what bounds this kernel at 3x its HBM time (805 MB per launch).
  const bool vec4 = (a.N & 3) == 0;
  float4 pav[MI][2], pbv[MI][2];
  auto fetch_ab = [&](int i) {
    if (L2Q_HH_SKIP & 4) {
#pragma unroll
      for (int j = 0; j < 2; ++j) { pav[i][j] = make_float4(1.f, 2.f, 3.f, 4.f); pbv[i][j] = pav[i][j]; }
      return;
    }
    const long m = m0 + wm + 16 * i + (lane & 15);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long nb4 = n0 + wn + 16 * j + 4 * (lane >> 4);
      const bool ok = nb4 < a.N && m < a.M;
      const long o = ok ? m * (long)a.N + nb4 : 0;
      pav[i][j] = *reinterpret_cast<const float4*>(a.a + o);
      pbv[i][j] = *reinterpret_cast<const float4*>(a.bsrc + o);
    }
  };
  if (vec4) {
#pragma unroll
    for (int i = 0; i < MI / 2; ++i) fetch_ab(i);
  }

  TileH<HT, HT, BM> lz;
  TileH<HT, HT, BN> l0, l1, l2;
  lz.fetch(Z, Z, m0, a.M, 0, K, 0, K, vec);
  l0.fetch(W0, W0, n0, a.N, 0, K, 0, K, vec);
  l1.fetch(W1, W1, n0, a.N, 0, K, 0, K, vec);
  l2.fetch(W2, W2, n0, a.N, 0, K, 0, K, vec);
  for (long k0 = 0; k0 < ((L2Q_HH_SKIP & 1) ? 0 : K); k0 += HBK) {
    __syncthreads();
    lz.store(Zs);
    l0.store(Ws[0]);
    l1.store(Ws[1]);
    l2.store(Ws[2]);
    __syncthreads();
    if (k0 + HBK < K) {
      lz.fetch(Z, Z, m0, a.M, k0 + HBK, K, 0, K, vec);
      l0.fetch(W0, W0, n0, a.N, k0 + HBK, K, 0, K, vec);
      l1.fetch(W1, W1, n0, a.N, k0 + HBK, K, 0, K, vec);
      l2.fetch(W2, W2, n0, a.N, k0 + HBK, K, 0, K, vec);
    }
#pragma unroll
    for (int ks = 0; ks < HBK; ks += 32) {
      const int kq = ks + 8 * (lane >> 4);
      vec_t fa[MI];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        fa[i] = *reinterpret_cast<const vec_t*>(&Zs[wm + 16 * i + (lane & 15)][kq]);
#pragma unroll
      for (int h = 0; h < 3; ++h) {
        vec_t fb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
          fb[j] = *reinterpret_cast<const vec_t*>(&Ws[h][wn + 16 * j + (lane & 15)][kq]);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[h][i][j] = MfmaH<HT>::run(fb[j], fa[i], acc[h][i][j]);
      }
    }
  }

  // ---- epilogue: heads -> update, in registers.  The MFMAs ran with W as the "row" operand:
  // lane holds chain m = lane & 15 of tile i and the four consecutive entries
  // n = 16 j + 4 (lane >> 4) + r of tile j: 16-byte accesses to v / x / force / mask / biases.
  const float eps = a.eps;
  if (vec4) {
#pragma unroll
    for (int i = MI / 2; i < MI; ++i) fetch_ab(i);       // before any store of the epilogue
  }
  float* __restrict__ pa = a.a;
  const float* __restrict__ pb = a.bsrc;
  float ld[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) ld[i] = 0.f;
  // Per-column parameters (biases, exp(coeff) scales, mask) of this wavefront's two 16-column groups:
  // they depend on j only, so they are fetched ONCE, as one 16-byte load per array and group -- 10-12
  // loads per wavefront.  (Fetched per element inside the row loop they were 160 scalar gathers per
  // wavefront and tile against 24 loads / stores of actual field traffic: the texture addresser, not
  // HBM, was what the epilogue waited for.)
  float pbs[2][4], pbt[2][4], pbq[2][4], pcs[2][4], pcq[2][4], pkeep[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const long nb4 = n0 + wn + 16 * j + 4 * (lane >> 4);
    if (vec4 && nb4 < a.N) {
      const float4 v0 = *reinterpret_cast<const float4*>(a.b[0] + nb4);
      const float4 v1 = *reinterpret_cast<const float4*>(a.b[1] + nb4);
      const float4 v2 = *reinterpret_cast<const float4*>(a.b[2] + nb4);
      const float4 v3 = *reinterpret_cast<const float4*>(a.cs + nb4);
      const float4 v4 = *reinterpret_cast<const float4*>(a.cq + nb4);
      pbs[j][0] = v0.x; pbs[j][1] = v0.y; pbs[j][2] = v0.z; pbs[j][3] = v0.w;
      pbt[j][0] = v1.x; pbt[j][1] = v1.y; pbt[j][2] = v1.z; pbt[j][3] = v1.w;
      pbq[j][0] = v2.x; pbq[j][1] = v2.y; pbq[j][2] = v2.z; pbq[j][3] = v2.w;
      pcs[j][0] = v3.x; pcs[j][1] = v3.y; pcs[j][2] = v3.z; pcs[j][3] = v3.w;
      pcq[j][0] = v4.x; pcq[j][1] = v4.y; pcq[j][2] = v4.z; pcq[j][3] = v4.w;
      if (XUPD) {
        const float4 v5 = *reinterpret_cast<const float4*>(a.mask + nb4);
        pkeep[j][0] = v5.x; pkeep[j][1] = v5.y; pkeep[j][2] = v5.z; pkeep[j][3] = v5.w;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        long n = nb4 + r < a.N ? nb4 + r : a.N - 1;
        if (n < 0) n = 0;
        pbs[j][r] = a.b[0][n]; pbt[j][r] = a.b[1][n]; pbq[j][r] = a.b[2][n];
        pcs[j][r] = a.cs[n]; pcq[j][r] = a.cq[n];
        if (XUPD) pkeep[j][r] = a.mask[n];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (!XUPD) pkeep[j][r] = 0.f;
      else if (a.complement) pkeep[j][r] = 1.f - pkeep[j][r];
    }
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const long m = m0 + wm + 16 * i + (lane & 15);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long nb4 = n0 + wn + 16 * j + 4 * (lane >> 4);
      if (nb4 >= a.N || m >= a.M) continue;
      float bs[4], bt[4], bq[4], cs[4], cq[4], keep[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        bs[r] = pbs[j][r]; bt[r] = pbt[j][r]; bq[r] = pbq[j][r];
        cs[r] = pcs[j][r]; cq[r] = pcq[j][r]; keep[r] = pkeep[j][r];
      }
      const long o = m * (long)a.N + nb4;
      float av[4], bv[4], out[4];
      if (vec4) {
        const float4 t0 = pav[i][j];
        const float4 t1 = pbv[i][j];
        av[0] = t0.x; av[1] = t0.y; av[2] = t0.z; av[3] = t0.w;
        bv[0] = t1.x; bv[1] = t1.y; bv[2] = t1.z; bv[3] = t1.w;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool in = nb4 + r < a.N;
          av[r] = in ? pa[o + r] : 0.f;
          bv[r] = in ? pb[o + r] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (L2Q_HH_SKIP & 2) {
          out[r] = av[r] + bv[r] + acc[0][i][j][r] + acc[1][i][j][r] + acc[2][i][j][r] + bs[r] + cq[r] + keep[r];
          continue;
        }
        const bool in = nb4 + r < a.N;
        float ldt;
        out[r] = hh_element<HT, XUPD, FWD, NCP>(acc[0][i][j][r], acc[1][i][j][r], acc[2][i][j][r], bs[r], bt[r],
                                                bq[r], cs[r], cq[r], a.st, eps, av[r], bv[r], keep[r], ldt);
        if (in) ld[i] += ldt;
      }
      if ((L2Q_HH_SKIP & 4) && out[0] + out[1] + out[2] + out[3] != 12345.678f) continue;
      if (vec4) {
        *reinterpret_cast<float4*>(pa + o) = make_float4(out[0], out[1], out[2], out[3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (nb4 + r < a.N) pa[o + r] = out[r];
      }
    }
  }
  // chain partial of logdet over this wave's 32 entries: the four lane groups hold 4 n each
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    double x = (double)ld[i];
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    const long m = m0 + wm + 16 * i + (lane & 15);
    if (lane < 16 && m < a.M) {
      const long col = (n0 / BN) * 2 + (wave & 1);
      a.logdet_part[m * a.ncols_part + col] = x;
    }
  }
}


// ---------------------------------------------------------------------------------------
// The same operation as u1_heads_update_h_kernel, organised as a STREAM over the chains with the
// weights stationary (round 3).  Same-box decomposition of the tile kernel at cfg-3 (8192 chains x
// 8192 entries, K = 256; tools/probe_heads_h3.py on -DL2Q_HH_SKIP builds): whole kernel 0.41 ms =
// staging + MFMA 0.20 ms (every 128 x 64 tile stages 160 KB of Z and W through registers into LDS in
// four dependent round trips) + field traffic 0.16-0.25 ms (805 MB) + epilogue arithmetic 0.06-0.09 ms,
// and the first two do not overlap.  Here
//  * a workgroup owns 64 entries (columns) and a range of chains; each of its four wavefronts keeps the
//    three heads' weights of ITS 16 columns in registers for the whole sweep (3 x K / 32 MFMA operand
//    fragments = 96 VGPRs at K = 256; W is read once per workgroup) together with the per-column
//    biases / scales / mask;
//  * the chains stream by in steps of 32 rows: their Z rows (32 x K 16-bit = 16 KB, four 16-byte chunks
//    per thread) and the fp32 field operands (a, b) of the NEXT step are requested while the current
//    step's epilogue runs; Z goes registers -> LDS (two stages, one barrier per step, chunk-swizzled so
//    that the ds_read_b128 fragments are conflict-free).  (An LDS-DMA version was written first: hipcc
//    cannot tell its vmcnt traffic from the operand prefetches and drains vmcnt to 0 around every
//    global_load_lds, which serialises the step; plain loads keep its scoreboard exact.)
//  * per step a wavefront issues 6 K / 32 MFMAs (32 rows x its 16 columns x 3 heads), runs the shared
//    hh_element arithmetic on 2 x 4 entries per lane and stores two float4.
// Same MFMA instruction, operand roles and k order as the tile kernel: identical accumulators; the fp32
// epilogue is contracted differently by hipcc in the two kernels (rare 1-ulp16 flips of a head); the
// per-chain log-det is summed in a different (fixed) order.
// Needs K in {32, 64, 128, 256}, N % 4 == 0 and 16-byte aligned operands (heads_h_launch falls back).
// MEASURED (cfg-3, same box, four rotating operand sets): 0.475 ms (v) / 0.56 ms (x) against the tile
// kernel's 0.420 / 0.48 ms -- with one step of prefetch the operand latency of every 32-row step is still
// exposed (hipcc waits vmcnt(0) at the top of a step; a second prefetch stage does not fit the 256
// registers next to the 96 of the stationary weights).  Kept as tuning `heads_h_stream = 1`, off by default.
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <typename HT, bool XUPD, bool FWD, bool NCP>
__global__ __launch_bounds__(kBlock, 2) void u1_heads_stream_h_kernel(HeadsHArgs a, int swz, int rows_per_wg) {
  constexpr int KS_MAX = 8;                       // K <= 256
  constexpr int ROWS = 32;                        // chains per step
  constexpr int STAGE = ROWS * 512;               // bytes (K = 256); smaller K uses a prefix
  using vec_t = typename MfmaH<HT>::vec_t;
  __shared__ __attribute__((aligned(1024))) char zs[2 * STAGE];
  __shared__ float red[2][4][ROWS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.K;
  const int ksteps = K >> 5;                      // MFMA k-steps of 32
  const int cpr = K >> 3;                         // 16-byte chunks per Z row
  const int rowb = K * 2;                         // bytes per Z row
  const int swm = (cpr < 16 ? cpr : 16) - 1;      // chunk swizzle mask
  const long ntiles = (a.N + 63) / 64;
  const long w = xcd_swizzle(blockIdx.x, gridDim.x, swz);
  const long n0 = (w % ntiles) * 64;              // n-tiles fastest: neighbours walk the same rows
  const long mbeg = (w / ntiles) * rows_per_wg;
  long mend = mbeg + rows_per_wg;
  if (mend > a.M) mend = a.M;
  if (mbeg >= a.M) return;
  const int nstep = (int)((mend - mbeg + ROWS - 1) / ROWS);

  // ---- stationary operands of this wavefront: columns nw0 .. nw0 + 15
  const long nw0 = n0 + 16 * wave;
  const long ncol = nw0 + (lane & 15);            // W row this lane's fragments come from
  const long nrow = ncol < a.N ? ncol : a.N - 1;
  vec_t wf[3][KS_MAX];
#pragma unroll
  for (int h = 0; h < 3; ++h) {
    const HT* W = (const HT*)a.W[h] + nrow * (long)K + 8 * (lane >> 4);
#pragma unroll
    for (int kk = 0; kk < KS_MAX; ++kk) {
      // unconditional (k-steps past K re-read the last one and are never used): a load behind a branch
      // makes hipcc guard every later use with s_waitcnt vmcnt(0), which would drain the prefetches
      const int ko = kk < ksteps ? 32 * kk : K - 32;
      wf[h][kk] = *reinterpret_cast<const vec_t*>(W + ko);
    }
  }
  const long nb4 = nw0 + 4 * (lane >> 4);         // this lane's four consecutive entries
  const bool ncok = nb4 < a.N;                    // (N % 4 == 0: all four or none)
  const long nq = ncok ? nb4 : 0;
  float bs[4], bt[4], bq[4], cs[4], cq[4], keep[4];
  {
    const float4 v0 = *reinterpret_cast<const float4*>(a.b[0] + nq);
    const float4 v1 = *reinterpret_cast<const float4*>(a.b[1] + nq);
    const float4 v2 = *reinterpret_cast<const float4*>(a.b[2] + nq);
    const float4 v3 = *reinterpret_cast<const float4*>(a.cs + nq);
    const float4 v4 = *reinterpret_cast<const float4*>(a.cq + nq);
    bs[0] = v0.x; bs[1] = v0.y; bs[2] = v0.z; bs[3] = v0.w;
    bt[0] = v1.x; bt[1] = v1.y; bt[2] = v1.z; bt[3] = v1.w;
    bq[0] = v2.x; bq[1] = v2.y; bq[2] = v2.z; bq[3] = v2.w;
    cs[0] = v3.x; cs[1] = v3.y; cs[2] = v3.z; cs[3] = v3.w;
    cq[0] = v4.x; cq[1] = v4.y; cq[2] = v4.z; cq[3] = v4.w;
#pragma unroll
    for (int r = 0; r < 4; ++r) keep[r] = 0.f;
    if (XUPD) {
      const float4 v5 = *reinterpret_cast<const float4*>(a.mask + nq);
      keep[0] = v5.x; keep[1] = v5.y; keep[2] = v5.z; keep[3] = v5.w;
      if (a.complement) {
#pragma unroll
        for (int r = 0; r < 4; ++r) keep[r] = 1.f - keep[r];
      }
    }
  }

  // ---- Z stream: a step's 32 rows are 32 * cpr 16-byte chunks; thread t carries chunks t, t + 256, ...
  // (at most four) in registers for one step and writes them to the stage when their turn comes.
  // Chunk c of row r sits at slot c ^ (r & swm) of its row: the ds_read_b128 fragments below (16 rows, one
  // chunk index) then touch 16 different bank groups.
  const int nchunk = ROWS * cpr;                  // 1024 (K = 256) .. 128 (K = 32)
  const char* zbase = reinterpret_cast<const char*>(a.Z);
  uint4 zr[4];
  auto zfetch = [&](long r0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int p = tid + 256 * q;
      if (p >= nchunk) p = nchunk - 1;            // unconditional loads (a load behind a branch is waited for
                                                  // on the spot); the surplus copies are never stored
      const int row = p / cpr, c = p - row * cpr;
      long m = r0 + row;
      if (m >= a.M) m = a.M - 1;                  // rows past the end re-read a valid one (masked later)
      zr[q] = *reinterpret_cast<const uint4*>(zbase + m * (long)rowb + (c << 4));
    }
  };
  auto zstore = [&](int st) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p = tid + 256 * q;
      if (p < nchunk) {
        const int row = p / cpr, c = p - row * cpr;
        *reinterpret_cast<uint4*>(zs + st * STAGE + row * rowb + ((c ^ (row & swm)) << 4)) = zr[q];
      }
    }
  };
  // fragment read offsets of this lane: row (lane & 15) [+ 16 i], chunk 4 kk + (lane >> 4)
  const int frow = lane & 15;
  const int fsw = frow & swm;                     // (row + 16 i) & swm == row & swm (swm <= 15)

  // ---- field operands: lane holds chain 16 i + (lane & 15) of the step and entries nb4 .. nb4 + 3
  float* __restrict__ pa = a.a;
  const float* __restrict__ pb = a.bsrc;
  float4 av[2][2], bv[2][2];                      // [slot][i]
  auto fetch = [&](int slot, long r0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long m = r0 + 16 * i + (lane & 15);
      const bool ok = ncok && m < mend;
      const long o = ok ? m * (long)a.N + nb4 : 0;
      av[slot][i] = *reinterpret_cast<const float4*>(pa + o);
      bv[slot][i] = *reinterpret_cast<const float4*>(pb + o);
    }
  };
  const float eps = a.eps;
  zfetch(mbeg);
  fetch(0, mbeg);
  auto step = [&](auto CUR, int s) {
    constexpr int cur = decltype(CUR)::value;      // stage / operand slot of this step (compile time)
    const long r0 = mbeg + (long)s * ROWS;
    zstore(cur);                                   // stage `cur` was last read two steps ago
    __syncthreads();
    // previous step's row sums of the log-det: the four wavefronts' partials meet here
    if (s > 0 && tid < ROWS) {
      const long m = r0 - ROWS + tid;
      if (m < mend) {
        const int pr = (s - 1) & 1;
        const double x = ((double)red[pr][0][tid] + (double)red[pr][1][tid]) +
                         ((double)red[pr][2][tid] + (double)red[pr][3][tid]);
        a.logdet_part[m * a.ncols_part + (n0 >> 6)] = x;
      }
    }
    v4f32 acc[3][2];
#pragma unroll
    for (int h = 0; h < 3; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[h][i] = (v4f32){0, 0, 0, 0};
    const char* sb = zs + cur * STAGE;
#pragma unroll
    for (int kk = 0; kk < KS_MAX; ++kk) {
      if (kk < ksteps) {
        vec_t fa[2];
        const int chunk = (4 * kk + (lane >> 4)) ^ fsw;
#pragma unroll
        for (int i = 0; i < 2; ++i)
          fa[i] = *reinterpret_cast<const vec_t*>(sb + (frow + 16 * i) * rowb + (chunk << 4));
#pragma unroll
        for (int h = 0; h < 3; ++h)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[h][i] = MfmaH<HT>::run(wf[h][kk], fa[i], acc[h][i]);
      }
    }
    // next step's operands (Z rows into registers, field values into the other slot): in flight during
    // this step's epilogue, consumed one step later
    if (s + 1 < nstep) {
      zfetch(r0 + ROWS);
      fetch(cur ^ 1, r0 + ROWS);
    }
    float ld[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long m = r0 + 16 * i + (lane & 15);
      const bool ok = ncok && m < mend;
      const float4 ta = av[cur][i];
      const float4 tb = bv[cur][i];
      const float a4[4] = {ta.x, ta.y, ta.z, ta.w}, b4[4] = {tb.x, tb.y, tb.z, tb.w};
      float out[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float ldt;
        out[r] = hh_element<HT, XUPD, FWD, NCP>(acc[0][i][r], acc[1][i][r], acc[2][i][r], bs[r], bt[r], bq[r],
                                                cs[r], cq[r], a.st, eps, a4[r], b4[r], keep[r], ldt);
        if (ok) ld[i] += ldt;
      }
      if (ok) *reinterpret_cast<float4*>(pa + m * (long)a.N + nb4) = make_float4(out[0], out[1], out[2], out[3]);
    }
    // row sums over this wavefront's 16 columns (four lane groups of 4 entries)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float x = ld[i];
      x += __shfl_xor(x, 16, 64);
      x += __shfl_xor(x, 32, 64);
      if (lane < 16) red[cur][wave][16 * i + lane] = x;
    }
  };
  for (int s = 0; s < nstep; s += 2) {
    step(std::integral_constant<int, 0>{}, s);
    if (s + 1 < nstep) step(std::integral_constant<int, 1>{}, s + 1);
  }
  __syncthreads();
  if (tid < ROWS) {
    const long m = mbeg + (long)(nstep - 1) * ROWS + tid;
    if (m < mend) {
      const int pr = (nstep - 1) & 1;
      const double x = ((double)red[pr][0][tid] + (double)red[pr][1][tid]) +
                       ((double)red[pr][2][tid] + (double)red[pr][3][tid]);
      a.logdet_part[m * a.ncols_part + (n0 >> 6)] = x;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Periodic conv layer of the U(1) ConvStack in half precision (autocast runs nn.Conv2d in 16
// bit as well): implicit GEMM as gemm.hip's conv_gemm_kernel -- row m of the virtual A operand
// is output pixel (b, ho, wo), never materialised -- on the 16-bit MFMA.  For the NHWC 16-bit
// activations of layers 2.. the K order is (i, j, ci) and C % 8 == 0, so a thread gathers eight
// consecutive input channels with ONE 16-byte load (VEC8); the first layer reads the fp32 NCHW
// lattice data ([cos, sin] of the links) element-wise and rounds it while staging.
// Output: NHWC 16-bit, r16(acc + bias) then r16(act(.)) -- autocast's rounding points.
// POOL (round 3): MaxPool2d(2) and the activation fused.  The GEMM row index then runs over (pooled
// pixel, window position): row m = 4 P + 2 dy + dx is conv pixel (2 hp + dy, 2 wp + dx) of pooled pixel
// P = (b, hp, wp), so the four pixels of a window are four neighbouring rows of one MFMA tile = four
// neighbouring lanes of the accumulator layout; the window maximum is two quad shuffles, and only the
// pooled, activated tile (a quarter of the pixels, contiguous in memory) is written.  The un-pooled
// intermediate -- 3.2 GB at cfg-3's 128-channel layer, written at ~1 TB/s and read back by the pool
// kernel -- never exists; pixels the floor-mode pool drops are not computed.  Rounding points as in the
// two-kernel path: r16(acc + bias) -> max -> r16(act(.)) (max commutes with the monotone rounding).
// wavefronts per SIMD the latency-bound variants are compiled for (pooled: the output tile in LDS is a
// quarter; <= 64 channels: 44 KB of LDS): 2 -> 3 took the pooled 128-channel layer of cfg-3 from 3.3 to 2.6 ms
#ifndef L2Q_CONV_POOL_OCC
#define L2Q_CONV_POOL_OCC 3
#endif
template <typename HT, typename IT, int KS, int BN, bool VEC8, bool POOL = false>
__global__ __launch_bounds__(kBlock, (POOL || BN <= 64) ? L2Q_CONV_POOL_OCC : 2) void conv_gemm_h_kernel(const IT* __restrict__ in,
                                                                ConvGeomH g,
                                                                const HT* __restrict__ Wt, int N,
                                                                const float* __restrict__ bias,
                                                                int act, HT* __restrict__ C) {
  constexpr int BM = 128;
  constexpr int WN = BN >= 64 ? 2 : 1, WM = 4 / WN;
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16;
  using vec_t = typename MfmaH<HT>::vec_t;
  __shared__ __attribute__((aligned(16))) HT As[BM][HLD];
  __shared__ __attribute__((aligned(16))) HT Ws[BN][HLD];
  __shared__ long rbase[BM];
  __shared__ int rr0[BM], rc0[BM];
  __shared__ __attribute__((aligned(16))) HT Cs[(POOL ? BM / 4 : BM) * BN];   // output tile, row stride N <= BN
  const int k = KS > 0 ? KS : g.k;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave / WN) * TM, wn = (wave % WN) * TN;
  const long m0 = (long)blockIdx.y * BM, n0 = (long)blockIdx.x * BN;
  for (int r = tid; r < BM; r += kBlock) {
    const long m = m0 + r;
    long base = -1;
    int r0 = 0, c0 = 0;
    if (m < g.M) {
      int wo, ho;
      long bidx;
      // 32-bit index arithmetic (M < 2^31, checked at launch): a 64-bit division is ~100 instructions
      const unsigned mu = (unsigned)m;
      if (POOL) {
        const unsigned P = mu >> 2, d = mu & 3;
        const unsigned t = P / (unsigned)g.Wp;
        const unsigned wp = P - t * (unsigned)g.Wp;
        const unsigned bq = t / (unsigned)g.Hp;
        ho = 2 * (int)(t - bq * (unsigned)g.Hp) + (int)(d >> 1);
        wo = 2 * (int)wp + (int)(d & 1);
        bidx = bq;
      } else {
        const unsigned t = mu / (unsigned)g.Wo;
        wo = (int)(mu - t * (unsigned)g.Wo);
        const unsigned bq = t / (unsigned)g.Ho;
        ho = (int)(t - bq * (unsigned)g.Ho);
        bidx = bq;
      }
      base = bidx * g.sn;
      r0 = (ho - (k - 1)) % g.H; if (r0 < 0) r0 += g.H;
      c0 = (wo - (k - 1)) % g.W; if (c0 < 0) c0 += g.W;
    }
    rbase[r] = base; rr0[r] = r0; rc0[r] = c0;
  }
  __syncthreads();

  v4f32 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (v4f32){0, 0, 0, 0};

  // A gather: VEC8: 8 threads x 8 channels per row, 32 rows per pass; else 64 threads x 1 column
  constexpr int AV = VEC8 ? 8 : 1;
  constexpr int ARPP = kBlock / (HBK / AV), ANP = BM / ARPP;      // 32 rows x 4 or 4 rows x 32
  typedef HT av_t __attribute__((ext_vector_type(AV)));
  av_t areg[ANP];
  const int akv = (tid % (HBK / AV)) * AV, arq = tid / (HBK / AV);
#define L2Q_CONVH_FETCH_A(K0)                                                           \
  do {                                                                                  \
    const unsigned kk_ = (unsigned)(K0) + (unsigned)akv;     /* 32-bit: Kc = C k k is small */ \
    const bool kin_ = kk_ < (unsigned)g.Kc;                                             \
    int j_, i_, ci_;                                                                    \
    if (g.clast) { const unsigned ij_ = kk_ / (unsigned)g.C; ci_ = (int)(kk_ - ij_ * (unsigned)g.C); \
                   i_ = (int)(ij_ / (unsigned)k); j_ = (int)(ij_ - (unsigned)i_ * (unsigned)k); }    \
    else { const unsigned ij_ = kk_ / (unsigned)k; j_ = (int)(kk_ - ij_ * (unsigned)k);               \
           ci_ = (int)(ij_ / (unsigned)k); i_ = (int)(ij_ - (unsigned)ci_ * (unsigned)k); }           \
    const long coff_ = (long)ci_ * g.sc;                                                \
    _Pragma("unroll") for (int p = 0; p < ANP; ++p) {                                   \
      const int row_ = arq + p * ARPP;                                                  \
      const long base_ = rbase[row_];                                                   \
      av_t v_;                                                                          \
      _Pragma("unroll") for (int e = 0; e < AV; ++e) v_[e] = (HT)0.f;                   \
      if (kin_ && base_ >= 0) {                                                         \
        int r_ = rr0[row_] + i_; if (r_ >= g.H) r_ -= g.H; if (r_ >= g.H) r_ %= g.H;    \
        int c_ = rc0[row_] + j_; if (c_ >= g.W) c_ -= g.W; if (c_ >= g.W) c_ %= g.W;    \
        const IT* src_ = in + base_ + coff_ + r_ * g.sh + c_ * g.sw;                    \
        if (VEC8) v_ = *reinterpret_cast<const av_t*>(src_);                            \
        else v_[0] = (HT)(float)src_[0];                                                \
      }                                                                                 \
      areg[p] = v_;                                                                     \
    }                                                                                   \
  } while (0)
  TileH<HT, HT, BN> lw;
  const bool vecw = (g.Kc % 8) == 0;
  L2Q_CONVH_FETCH_A(0);
  lw.fetch(Wt, Wt, n0, N, 0, g.Kc, 0, g.Kc, vecw);
  for (long k0 = 0; k0 < g.Kc; k0 += HBK) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < ANP; ++p) *reinterpret_cast<av_t*>(&As[arq + p * ARPP][akv]) = areg[p];
    lw.store(Ws);
    __syncthreads();
    if (k0 + HBK < g.Kc) {
      L2Q_CONVH_FETCH_A(k0 + HBK);
      lw.fetch(Wt, Wt, n0, N, k0 + HBK, g.Kc, 0, g.Kc, vecw);
    }
#pragma unroll
    for (int ks = 0; ks < HBK; ks += 32) {
      const int kq = ks + 8 * (lane >> 4);
      vec_t fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        fa[i] = *reinterpret_cast<const vec_t*>(&As[wm + 16 * i + (lane & 15)][kq]);
#pragma unroll
      for (int j = 0; j < NI; ++j)
        fb[j] = *reinterpret_cast<const vec_t*>(&Ws[wn + 16 * j + (lane & 15)][kq]);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = MfmaH<HT>::run(fb[j], fa[i], acc[i][j]);
    }
  }
#undef L2Q_CONVH_FETCH_A
  // W was the MFMA row operand: lane owns pixel m = lane & 15 of tile i, channels 4 (lane >> 4) + r
  // One N-tile (cout <= BN, every layer of the default stack): the workgroup's output
  // C[m0 .. m0+127][0 .. N) is ONE contiguous range of memory.  The lanes' 4-channel pieces
  // (8 bytes, 32-byte runs per wavefront store: ~1.2 TB/s measured) are therefore assembled in
  // LDS and written as a flat 16-byte-per-lane stream.
  const bool staged = (POOL || BN == 128) && gridDim.x == 1 && (N % 8) == 0;   // (un-pooled narrower tiles: no gain measured)
  const bool vecc = (N % 4) == 0;
  if (staged) __syncthreads();                       // all fragment reads of As / Ws are done
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const long nb4 = n0 + wn + 16 * j + 4 * (lane >> 4);
    if (!POOL && nb4 >= N) continue;
    float cb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) cb[r] = (bias && nb4 < N) ? bias[nb4 + r < N ? nb4 + r : N - 1] : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int ml = wm + 16 * i + (lane & 15);
      const long m = m0 + ml;
      typedef HT cv __attribute__((ext_vector_type(4)));
      cv o;
      if (POOL) {
        // every lane takes part in the quad shuffles; rows past M hold bias only and are never stored
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = rnd<HT>(acc[i][j][r] + cb[r]);
          v = fmaxf(v, __shfl_xor(v, 1, 64));
          v = fmaxf(v, __shfl_xor(v, 2, 64));
          o[r] = (HT)(act != L2Q_ACT_NONE ? act_h(v, act) : v);
        }
        if ((lane & 3) != 0 || m >= g.M || nb4 >= N) continue;
        const int pl = ml >> 2;                       // pooled pixel within the tile
        if (staged) {
          *reinterpret_cast<cv*>(&Cs[(long)pl * N + nb4]) = o;
          continue;
        }
        HT* dst = C + ((m0 >> 2) + pl) * N + nb4;
        if (vecc) *reinterpret_cast<cv*>(dst) = o;
        else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (nb4 + r < N) dst[r] = o[r];
        }
        continue;
      }
      if (m >= g.M) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (HT)epilogue_h<HT>(acc[i][j][r], cb[r], 1.f, false, act);
      if (staged) {
        *reinterpret_cast<cv*>(&Cs[(long)ml * N + nb4]) = o;
        continue;
      }
      HT* dst = C + m * N + nb4;
      if (vecc) *reinterpret_cast<cv*>(dst) = o;
      else {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (nb4 + r < N) dst[r] = o[r];
      }
    }
  }
  if (staged) {
    __syncthreads();
    long rows = g.M - m0 < BM ? g.M - m0 : BM;
    long row0 = m0;
    if (POOL) { rows >>= 2; row0 >>= 2; }             // pooled pixels of this tile: contiguous in memory
    const long total = rows * N;                      // halves, multiple of 8
    typedef HT v8 __attribute__((ext_vector_type(8)));
    HT* dst = C + row0 * N;
    for (long idx = (long)tid * 8; idx < total; idx += (long)kBlock * 8)
      *reinterpret_cast<v8*>(dst + idx) = *reinterpret_cast<const v8*>(&Cs[idx]);
  }
}

// out[b, ho, wo, c] = r16(act(max over the pool x pool window of in[b, ., ., c])), NHWC 16-bit.
// VEC channels per thread (8 = one 16-byte access when C % 8 == 0, else 1).
template <typename HT, int VEC>
__global__ __launch_bounds__(kBlock) void maxpool_act_nhwc_h_kernel(const HT* __restrict__ in, int H,
                                                                    int W, int C, int pool, int act,
                                                                    int Ho, int Wo, long total,
                                                                    HT* __restrict__ out) {
  typedef HT hv __attribute__((ext_vector_type(VEC)));
  const long idx = (long)blockIdx.x * kBlock + threadIdx.x;     // over [b, ho, wo, C / VEC]
  if (idx >= total) return;
  const int cv = C / VEC;
  const int c = (int)(idx % cv) * VEC;
  long t = idx / cv;
  const int wo = (int)(t % Wo); t /= Wo;
  const int ho = (int)(t % Ho);
  const long b = t / Ho;
  float m[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) m[e] = -INFINITY;
  for (int i = 0; i < pool; ++i)
    for (int j = 0; j < pool; ++j) {
      const hv v = *reinterpret_cast<const hv*>(in + ((b * H + ho * pool + i) * W + wo * pool + j) * C + c);
#pragma unroll
      for (int e = 0; e < VEC; ++e) m[e] = fmaxf(m[e], (float)v[e]);
    }
  hv o;
#pragma unroll
  for (int e = 0; e < VEC; ++e) o[e] = (HT)act_h(m[e], act);
  *reinterpret_cast<hv*>(out + ((b * Ho + ho) * Wo + wo) * (long)C + c) = o;
}

// fp32 NCHW -> 16-bit NHWC with the channel count padded to CP (zeros): gives the first conv
// layer (2 or 4 input channels) the 16-byte channel gathers of the later layers.
template <typename HT>
__global__ __launch_bounds__(kBlock) void nchw_to_nhwc_pad_h_kernel(const float* __restrict__ in,
                                                                    int C, long HW, int CP,
                                                                    long total, HT* __restrict__ out) {
  const long idx = (long)blockIdx.x * kBlock + threadIdx.x;      // over [b, hw]
  if (idx >= total) return;
  const long b = idx / HW, p = idx % HW;
  const float* src = in + b * C * HW + p;
  HT* dst = out + idx * CP;
  for (int c = 0; c < CP; ++c) dst[c] = c < C ? (HT)src[c * HW] : (HT)0.f;
}

// gemm_f16_small.hip: K, N <= 256 on many rows; false -> not its case
template <typename HT>
bool gemm_h_small_launch(const void* A, const void* W, int M, int N, long K, const EpiH& epi, void* C, int c_is_f32,
                         hipStream_t st);
// gemm_lt.hip: hipBLASLt for plain layers with M, N, K in the thousands; false -> not taken
bool gemm_h_lt_shape(int M, int N, long K);
size_t gemm_h_lt_ws_bytes(int M, int N, long K);
template <typename HT>
bool gemm_h_lt_launch(const void* A, const void* W, int M, int N, long K, const EpiH& epi, void* C, int c_is_f32,
                      void* ws, size_t ws_bytes, hipStream_t st);
// gemm_f16_dma.hip: 256 x 256 LDS-DMA kernel when the shape fits it; false -> the kernels here
template <typename HT>
bool gemm_h_dma_launch(const void* A, const void* W, int M, int N, long K, const EpiH& epi, void* C,
                       int c_is_f32, hipStream_t st);

// conv_patch_f16.hip: LDS-patch kernel when the layer fits it; false -> use the gather kernel here
template <typename HT>
bool conv_patch_launch(const void* in, const ConvGeomH& g, const void* w, const float* bias,
                       int cout, int act, void* out, hipStream_t st);

// conv_stream_f16.hip: persistent whole-K kernel when the layer fits it; false -> the gather kernel here
template <typename HT>
bool conv_stream_launch(const void* in, const ConvGeomH& g, const void* w, const float* bias, int cout,
                        int act, void* out, hipStream_t st);

template <typename HT, typename IT>
static int conv_h_launch(const void* in_, ConvGeomH g, const void* w_, const float* bias, int cout,
                         int act, void* out_, hipStream_t st) {
  const IT* in = (const IT*)in_;
  const HT* weight = (const HT*)w_;
  HT* out = (HT*)out_;
  const int bn = cout <= 32 ? 32 : cout <= 64 ? 64 : 128;
  const dim3 grid((unsigned)cdiv(cout, bn), (unsigned)cdiv(g.M, 128)), block(kBlock);
  // 16-byte channel gathers: 16-bit NHWC input, (i, j, ci) order, C % 8 == 0, aligned
  const bool vec8 = sizeof(IT) == 2 && g.clast && g.sc == 1 && g.C % 8 == 0 && g.sw % 8 == 0 &&
                    g.sh % 8 == 0 && g.sn % 8 == 0 && al16(in);
  if (g.pool != 2 && vec8 && tuning().conv_patch &&
      conv_patch_launch<HT>(in_, g, w_, bias, cout, act, out_, st))
    return check_launch("l2q_conv_gemm_periodic_h");
  if (vec8 && tuning().conv_stream && cout > (g.pool == 2 ? 0 : 16) &&
      conv_stream_launch<HT>(in_, g, w_, bias, cout, act, out_, st))
    return check_launch("l2q_conv_gemm_periodic_h");
#define L2Q_CHB(KS, BNV)                                                                         \
  do {                                                                                           \
    if (g.pool == 2 && vec8)                                                                     \
      hipLaunchKernelGGL((conv_gemm_h_kernel<HT, IT, KS, BNV, sizeof(IT) == 2, true>), grid,     \
                         block, 0, st, in, g, weight, cout, bias, act, out);                     \
    else if (g.pool == 2)                                                                        \
      hipLaunchKernelGGL((conv_gemm_h_kernel<HT, IT, KS, BNV, false, true>), grid, block, 0, st, \
                         in, g, weight, cout, bias, act, out);                                   \
    else if (vec8)                                                                               \
      hipLaunchKernelGGL((conv_gemm_h_kernel<HT, IT, KS, BNV, sizeof(IT) == 2>), grid, block, 0, \
                         st, in, g, weight, cout, bias, act, out);                               \
    else                                                                                         \
      hipLaunchKernelGGL((conv_gemm_h_kernel<HT, IT, KS, BNV, false>), grid, block, 0, st, in,   \
                         g, weight, cout, bias, act, out);                                       \
  } while (0)
#define L2Q_CH(KS)                                                     \
  do {                                                                 \
    if (bn == 32) L2Q_CHB(KS, 32);                                     \
    else if (bn == 64) L2Q_CHB(KS, 64);                                \
    else L2Q_CHB(KS, 128);                                             \
  } while (0)
  switch (g.k) {
    case 2: L2Q_CH(2); break;
    case 3: L2Q_CH(3); break;
    case 5: L2Q_CH(5); break;
    default: L2Q_CH(0); break;
  }
#undef L2Q_CH
#undef L2Q_CHB
  return check_launch("l2q_conv_gemm_periodic_h");
}

__global__ void cast_f64_f32_kernel(const double* __restrict__ in, float* __restrict__ out, int n,
                                    int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = accumulate ? out[i] + (float)in[i] : (float)in[i];
}

// heads_kstream_f16.hip: K-split stream kernel (tuning heads_h_stream = 2); false -> not its case
template <typename HT>
bool heads_h_kstream_launch(HeadsHArgs a, int xupd, int forward, int use_ncp, int swz, float* logdet,
                            int accumulate, hipStream_t st, bool any_length);

template <typename HT>
static int heads_h_launch(HeadsHArgs a, int xupd, int forward, int use_ncp, float* logdet,
                          int accumulate, void* ws, hipStream_t st) {
  const int bm = tuning().heads_h_bm;
  const long ntile = cdiv(a.N, 64), mtile = cdiv(a.M, bm);
  const dim3 grid((unsigned)(ntile * mtile)), block(kBlock);
  const int swz = tuning().xcd_swizzle;
  const int nfast = tuning().heads_h_order;
  const int stg = tuning().heads_stagger;
  double* part = (double*)ws;
  double* tmp = part + (size_t)a.M * a.ncols_part;
  a.logdet_part = part;
  // K-split stream kernel (writes every partial it sums: no zeroing, its own finalize)
  if (tuning().heads_h_stream >= 2 &&
      heads_h_kstream_launch<HT>(a, xupd, forward, use_ncp, swz, logdet, accumulate, st, tuning().heads_h_stream == 3))
    return check_launch("l2q_u1_heads_update_h");
  launch_zero(part, (size_t)a.M * a.ncols_part * sizeof(double), st);
  // weights-stationary stream kernel wherever its shape conditions hold (tuning heads_h_stream = 0: tile kernel)
  const bool stream = tuning().heads_h_stream == 1 && (a.K == 32 || a.K == 64 || a.K == 128 || a.K == 256) &&
                      (a.N & 3) == 0 && al16(a.a) && al16(a.bsrc) && al16(a.b[0]) && al16(a.b[1]) &&
                      al16(a.b[2]) && al16(a.cs) && al16(a.cq) && (!xupd || al16(a.mask)) && al16(a.Z) &&
                      al16(a.W[0]) && al16(a.W[1]) && al16(a.W[2]);
  if (stream) {
    const long nt = cdiv(a.N, 64);
    long msplit = cdiv(768, nt);
    const long maxsplit = cdiv(a.M, 32);
    if (msplit > maxsplit) msplit = maxsplit;
    if (msplit < 1) msplit = 1;
    const int rows_per_wg = (int)(cdiv(cdiv(a.M, msplit), 32) * 32);
    msplit = cdiv(a.M, rows_per_wg);
    const dim3 sgrid((unsigned)(nt * msplit));
#define L2Q_HS(X, F, C) \
  hipLaunchKernelGGL((u1_heads_stream_h_kernel<HT, X, F, C>), sgrid, block, 0, st, a, swz, rows_per_wg)
    if (!xupd) { if (forward) L2Q_HS(false, true, false); else L2Q_HS(false, false, false); }
    else if (use_ncp) { if (forward) L2Q_HS(true, true, true); else L2Q_HS(true, false, true); }
    else { if (forward) L2Q_HS(true, true, false); else L2Q_HS(true, false, false); }
#undef L2Q_HS
    launch_finalize(part, tmp, a.M, a.ncols_part, 1, 1.0, 0.0, st);
    hipLaunchKernelGGL(cast_f64_f32_kernel, dim3((unsigned)cdiv(a.M, 64)), dim3(64), 0, st, tmp, logdet,
                       a.M, accumulate);
    return check_launch("l2q_u1_heads_update_h");
  }
#define L2Q_HH(X, F, C)                                                                          \
  do {                                                                                           \
    if (bm == 128)                                                                               \
      hipLaunchKernelGGL((u1_heads_update_h_kernel<HT, X, F, C, 128>), grid, block, 0, st, a,    \
                         swz, nfast, stg);                                                  \
    else                                                                                         \
      hipLaunchKernelGGL((u1_heads_update_h_kernel<HT, X, F, C, 64>), grid, block, 0, st, a,     \
                         swz, nfast, stg);                                                  \
  } while (0)
  if (!xupd) { if (forward) L2Q_HH(false, true, false); else L2Q_HH(false, false, false); }
  else if (use_ncp) { if (forward) L2Q_HH(true, true, true); else L2Q_HH(true, false, true); }
  else { if (forward) L2Q_HH(true, true, false); else L2Q_HH(true, false, false); }
#undef L2Q_HH
  launch_finalize(part, tmp, a.M, a.ncols_part, 1, 1.0, 0.0, st);
  hipLaunchKernelGGL(cast_f64_f32_kernel, dim3((unsigned)cdiv(a.M, 64)), dim3(64), 0, st, tmp, logdet,
                     a.M, accumulate);
  return check_launch("l2q_u1_heads_update_h");
}

}  // namespace l2q

using namespace l2q;

extern "C" {

size_t l2q_gemm_h_ws_bytes(int M, int N, long K, long K2) {
  if (M <= 0 || N <= 0 || K + K2 <= 0) return 0;
  bool wide = false;
  const int splits = pick_config_h(M, N, K + K2, &wide);
  size_t need = (splits == 1 && !wide) ? 0 : (size_t)(splits + 1) * M * N * sizeof(float);
  if (wide && splits == 1 && (long)M * N >= 1024L * 1024L) need = 0;     // fused wide tile
  // the streaming input-layer kernel (fp32 operands, N <= 256): up to 8 K-splits of partial sums.  The
  // element type and the U1X form are not known here: sized for any of them.
  if (tuning().gemm_h_skinny != 0 && N <= 256 && K + K2 >= 4096 && M >= 1024) {
    const size_t sk = gemm_h_skinny_ws_bytes(M, N, K, K2);     // partial sums + the slab-major copy of W
    if (sk > need) need = sk;
  }
  if (K2 == 0 && gemm_h_lt_shape(M, N, K) && gemm_h_lt_ws_bytes(M, N, K) > need) need = gemm_h_lt_ws_bytes(M, N, K);
  return need;
}

int l2q_gemm_h_skinny_splits(int M, int N, long K, long K2, int u1x) {
  const int sk = tuning().gemm_h_skinny;
  if (sk == 0 || M <= 0 || N <= 0 || K <= 0 || K2 < 0 || (u1x && (K & 1))) return 0;
  return gemm_h_skinny_splits(M, N, u1x ? K / 2 : K, K2, sk == 1 ? 0 : sk);
}

int l2q_gemm_h(int half_type, const void* A, int a_is_f32, const void* W, int M, int N, long K,
               const void* A2, const void* W2, long K2, const float* bias, const float* bias2,
               const float* coeff, float scale, int act, void* C, int c_is_f32, void* ws,
               size_t ws_bytes, void* stream) {
  L2Q_REQUIRE(A && W && C, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(M > 0 && N > 0 && K > 0 && K2 >= 0, L2Q_EINVAL, "non-positive size");
  L2Q_REQUIRE(K2 == 0 || (A2 && W2), L2Q_EINVAL, "second operand pair missing");
  L2Q_REQUIRE(act >= L2Q_ACT_NONE && act <= L2Q_ACT_SWISH, L2Q_EINVAL, "bad activation");
  L2Q_REQUIRE(half_type == L2Q_HALF_F16 || half_type == L2Q_HALF_BF16, L2Q_EINVAL, "bad half type");
  L2Q_REQUIRE(coeff == nullptr || c_is_f32, L2Q_EINVAL, "exp(coeff) scaling needs an fp32 output");
  const EpiH epi{bias, bias2, coeff, scale, act};
  const hipStream_t st = (hipStream_t)stream;
  // hidden layers on many chains (K, N <= 256): one pass, all N columns per workgroup (gemm_f16_small.hip)
  if (tuning().gemm_h_small && !a_is_f32 && K2 == 0) {
    const bool done = half_type == L2Q_HALF_F16
                          ? gemm_h_small_launch<_Float16>(A, W, M, N, K, epi, C, c_is_f32, st)
                          : gemm_h_small_launch<__bf16>(A, W, M, N, K, epi, C, c_is_f32, st);
    if (done) return check_launch("l2q_gemm_h");
  }
  // plain layers with every dimension in the thousands: hipBLASLt (gemm_lt.hip), when it is there and takes the shape
  if (!a_is_f32 && K2 == 0 && gemm_h_lt_shape(M, N, K)) {
    const bool done = half_type == L2Q_HALF_F16
                          ? gemm_h_lt_launch<_Float16>(A, W, M, N, K, epi, C, c_is_f32, ws, ws_bytes, st)
                          : gemm_h_lt_launch<__bf16>(A, W, M, N, K, epi, C, c_is_f32, ws, ws_bytes, st);
    if (done) return check_launch("l2q_gemm_h");
  }
  // big 16-bit x 16-bit layers: 256 x 256 tiles on LDS-DMA staging (gemm_f16_dma.hip)
  if (tuning().gemm_h_dma && !a_is_f32 && K2 == 0) {
    const bool done = half_type == L2Q_HALF_F16
                          ? gemm_h_dma_launch<_Float16>(A, W, M, N, K, epi, C, c_is_f32, st)
                          : gemm_h_dma_launch<__bf16>(A, W, M, N, K, epi, C, c_is_f32, st);
    if (done) return check_launch("l2q_gemm_h");
  }
  if (half_type == L2Q_HALF_F16)
    return gemm_h_dispatch<_Float16>(A, a_is_f32, W, M, N, K, A2, W2, K2, epi, C, c_is_f32, ws,
                                     ws_bytes, st);
  return gemm_h_dispatch<__bf16>(A, a_is_f32, W, M, N, K, A2, W2, K2, epi, C, c_is_f32, ws,
                                 ws_bytes, st);
}

size_t l2q_u1_heads_update_h_ws_bytes(int M, long N) {
  if (M <= 0 || N <= 0) return 0;
  return ((size_t)M * (size_t)(cdiv(N, 64) * 2) + (size_t)M) * sizeof(double);
}

int l2q_u1_heads_update_h(int half_type, const void* Z, int M, int K, long N, const void* Ws,
                          const float* bs, const float* cs, const void* Wt, const float* bt,
                          float scale_t, const void* Wq, const float* bq, const float* cq,
                          int x_update, float* a, const float* b, const float* mask,
                          int complement, float eps, int forward, int use_ncp, float* logdet,
                          int accumulate, void* ws, size_t ws_bytes, void* stream) {
  L2Q_REQUIRE(Z && Ws && bs && cs && Wt && bt && Wq && bq && cq && a && b && logdet && ws,
              L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(!x_update || mask, L2Q_EINVAL, "x-update needs the mask");
  L2Q_REQUIRE(M > 0 && K > 0 && N > 0 && N < 2000000000L, L2Q_EINVAL, "bad size");
  L2Q_REQUIRE(half_type == L2Q_HALF_F16 || half_type == L2Q_HALF_BF16, L2Q_EINVAL, "bad half type");
  L2Q_REQUIRE(K % 8 != 0 || (al16(Z) && al16(Ws) && al16(Wt) && al16(Wq)), L2Q_ESHAPE,
              "operands must be 16-byte aligned");
  L2Q_REQUIRE(ws_bytes >= l2q_u1_heads_update_h_ws_bytes(M, N), L2Q_ESHAPE, "workspace too small");
  HeadsHArgs h;
  h.Z = Z; h.W[0] = Ws; h.W[1] = Wt; h.W[2] = Wq; h.b[0] = bs; h.b[1] = bt; h.b[2] = bq;
  h.cs = cs; h.cq = cq; h.st = scale_t; h.eps = eps; h.a = a; h.bsrc = b; h.mask = mask;
  h.complement = complement; h.logdet_part = nullptr;
  h.M = M; h.N = (int)N; h.K = K; h.ncols_part = (int)(cdiv(N, 64) * 2);
  const hipStream_t st = (hipStream_t)stream;
  if (half_type == L2Q_HALF_F16)
    return heads_h_launch<_Float16>(h, x_update, forward, use_ncp, logdet, accumulate, ws, st);
  return heads_h_launch<__bf16>(h, x_update, forward, use_ncp, logdet, accumulate, ws, st);
}

int l2q_gemm_h_u1x(int half_type, const float* x, const float* mask, int complement, const void* W,
                   int M, int N, long xdim, const float* A2, const void* W2, long K2,
                   const float* bias, const float* bias2, int act, void* C, void* ws,
                   size_t ws_bytes, void* stream) {
  L2Q_REQUIRE(x && mask && W && C, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(M > 0 && N > 0 && xdim > 0 && K2 >= 0, L2Q_EINVAL, "non-positive size");
  L2Q_REQUIRE(K2 == 0 || (A2 && W2), L2Q_EINVAL, "second operand pair missing");
  L2Q_REQUIRE(act >= L2Q_ACT_NONE && act <= L2Q_ACT_SWISH, L2Q_EINVAL, "bad activation");
  L2Q_REQUIRE(half_type == L2Q_HALF_F16 || half_type == L2Q_HALF_BF16, L2Q_EINVAL, "bad half type");
  const EpiH epi{bias, bias2, nullptr, 1.f, act};
  const hipStream_t st = (hipStream_t)stream;
  if (half_type == L2Q_HALF_F16)
    return gemm_h_dispatch<_Float16>(x, 1, W, M, N, 2 * xdim, A2, W2, K2, epi, C, 0, ws, ws_bytes, st,
                                     mask, complement);
  return gemm_h_dispatch<__bf16>(x, 1, W, M, N, 2 * xdim, A2, W2, K2, epi, C, 0, ws, ws_bytes, st,
                                 mask, complement);
}

int l2q_conv_gemm_periodic_h(int half_type, const void* in, int in_is_f32, long sn, long sc, long sh,
                             long sw, int nb, int C, int H, int W, int k, const void* weight,
                             int channels_last_cols, const float* bias, int cout, int act,
                             void* out, void* stream) {
  L2Q_REQUIRE(in && weight && out, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && C > 0 && H > 0 && W > 0 && k > 0 && cout > 0, L2Q_EINVAL,
              "non-positive size");
  L2Q_REQUIRE(act >= L2Q_ACT_NONE && act <= L2Q_ACT_SWISH, L2Q_EINVAL, "bad activation");
  L2Q_REQUIRE(half_type == L2Q_HALF_F16 || half_type == L2Q_HALF_BF16, L2Q_EINVAL, "bad half type");
  ConvGeomH g;
  g.sn = sn; g.sc = sc; g.sh = sh; g.sw = sw; g.C = C; g.H = H; g.W = W; g.k = k;
  g.Ho = H + k - 1; g.Wo = W + k - 1; g.Kc = C * k * k;
  g.clast = channels_last_cols ? 1 : 0;
  g.M = (long)nb * g.Ho * g.Wo;
  L2Q_REQUIRE(cdiv(g.M, 128) < 65536L * 16 && g.M < (1L << 31), L2Q_ESHAPE, "too many output pixels");
  const hipStream_t st = (hipStream_t)stream;
  if (half_type == L2Q_HALF_F16) {
    return in_is_f32 ? conv_h_launch<_Float16, float>(in, g, weight, bias, cout, act, out, st)
                     : conv_h_launch<_Float16, _Float16>(in, g, weight, bias, cout, act, out, st);
  }
  return in_is_f32 ? conv_h_launch<__bf16, float>(in, g, weight, bias, cout, act, out, st)
                   : conv_h_launch<__bf16, __bf16>(in, g, weight, bias, cout, act, out, st);
}

int l2q_conv_pool_gemm_periodic_h(int half_type, const void* in, int in_is_f32, long sn, long sc,
                                  long sh, long sw, int nb, int C, int H, int W, int k,
                                  const void* weight, int channels_last_cols, const float* bias,
                                  int cout, int act, void* out, void* stream) {
  L2Q_REQUIRE(in && weight && out, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && C > 0 && H > 0 && W > 0 && k > 0 && cout > 0, L2Q_EINVAL,
              "non-positive size");
  L2Q_REQUIRE(act >= L2Q_ACT_NONE && act <= L2Q_ACT_SWISH, L2Q_EINVAL, "bad activation");
  L2Q_REQUIRE(half_type == L2Q_HALF_F16 || half_type == L2Q_HALF_BF16, L2Q_EINVAL, "bad half type");
  ConvGeomH g;
  g.sn = sn; g.sc = sc; g.sh = sh; g.sw = sw; g.C = C; g.H = H; g.W = W; g.k = k;
  g.Ho = H + k - 1; g.Wo = W + k - 1; g.Kc = C * k * k;
  g.clast = channels_last_cols ? 1 : 0;
  g.pool = 2; g.Hp = g.Ho / 2; g.Wp = g.Wo / 2;
  L2Q_REQUIRE(g.Hp > 0 && g.Wp > 0, L2Q_ESHAPE, "image smaller than the pooling window");
  g.M = 4L * nb * g.Hp * g.Wp;
  L2Q_REQUIRE(cdiv(g.M, 128) < 65536L * 16 && g.M < (1L << 31), L2Q_ESHAPE, "too many output pixels");
  const hipStream_t st = (hipStream_t)stream;
  if (half_type == L2Q_HALF_F16) {
    return in_is_f32 ? conv_h_launch<_Float16, float>(in, g, weight, bias, cout, act, out, st)
                     : conv_h_launch<_Float16, _Float16>(in, g, weight, bias, cout, act, out, st);
  }
  return in_is_f32 ? conv_h_launch<__bf16, float>(in, g, weight, bias, cout, act, out, st)
                   : conv_h_launch<__bf16, __bf16>(in, g, weight, bias, cout, act, out, st);
}

int l2q_nchw_to_nhwc_pad_h(int half_type, const float* in, int nb, int C, int H, int W, int cpad,
                           void* out, void* stream) {
  L2Q_REQUIRE(in && out, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && C > 0 && H > 0 && W > 0 && cpad >= C, L2Q_EINVAL, "bad size");
  L2Q_REQUIRE(half_type == L2Q_HALF_F16 || half_type == L2Q_HALF_BF16, L2Q_EINVAL, "bad half type");
  const long HW = (long)H * W, total = (long)nb * HW;
  const dim3 grid((unsigned)cdiv(total, kBlock)), block(kBlock);
  const hipStream_t st = (hipStream_t)stream;
  if (half_type == L2Q_HALF_F16)
    hipLaunchKernelGGL(nchw_to_nhwc_pad_h_kernel<_Float16>, grid, block, 0, st, in, C, HW, cpad, total,
                       (_Float16*)out);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_pad_h_kernel<__bf16>, grid, block, 0, st, in, C, HW, cpad, total,
                       (__bf16*)out);
  return check_launch("l2q_nchw_to_nhwc_pad_h");
}

int l2q_maxpool_act_nhwc_h(int half_type, const void* in, int nb, int H, int W, int C, int pool,
                           int act, void* out, void* stream) {
  L2Q_REQUIRE(in && out, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && H > 0 && W > 0 && C > 0 && pool > 0, L2Q_EINVAL, "non-positive size");
  L2Q_REQUIRE(half_type == L2Q_HALF_F16 || half_type == L2Q_HALF_BF16, L2Q_EINVAL, "bad half type");
  const int Ho = H / pool, Wo = W / pool;
  L2Q_REQUIRE(Ho > 0 && Wo > 0, L2Q_ESHAPE, "pooling window larger than the image");
  const int vec = (C % 8 == 0 && al16(in) && al16(out)) ? 8 : 1;
  const long total = (long)nb * Ho * Wo * (C / vec);
  const dim3 grid((unsigned)cdiv(total, kBlock)), block(kBlock);
  const hipStream_t st = (hipStream_t)stream;
#define L2Q_MP(HT, V)                                                                              \
  hipLaunchKernelGGL((maxpool_act_nhwc_h_kernel<HT, V>), grid, block, 0, st, (const HT*)in, H, W, C, \
                     pool, act, Ho, Wo, total, (HT*)out)
  if (half_type == L2Q_HALF_F16) { if (vec == 8) L2Q_MP(_Float16, 8); else L2Q_MP(_Float16, 1); }
  else { if (vec == 8) L2Q_MP(__bf16, 8); else L2Q_MP(__bf16, 1); }
#undef L2Q_MP
  return check_launch("l2q_maxpool_act_nhwc_h");
}

}  // extern "C"
