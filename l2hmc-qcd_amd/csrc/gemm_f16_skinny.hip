// gemm_f16_skinny.hip -- the INPUT layer of the half-precision U(1) networks on large lattices:
//
//   part[z][M][N] = A[M][K-range z] . W[N][K-range z]^T + A2[M][K2-range z] . W2[N][K2-range z]^T
//
// A, A2 are fp32 lattice data (links or cos/sin of links, momenta, force: xdim = 2 L^2 columns each), N =
// units[0] <= 256, W / W2 16-bit.  At BASELINE cfg-3 (64 x 64, 8192 chains) the layer is 8192 x 256 x
// 16 384..24 576: 537 MB of fp32 operand against 8-12 MB of weights and 69-103 GFLOP -- a STREAM of A with
// a matrix product attached (HBM time ~0.1 ms, MFMA time ~0.03 ms), which gemm_nt_h_kernel's 128 x 256 tile
// (one workgroup per CU, one slab requested at a time behind two barriers) ran at 2.1 TB/s.  Here:
//   * all N columns per workgroup, so A is read from HBM exactly once (as in the 128 x 256 tile);
//   * 64 chains x 256 threads per workgroup (wavefront w owns output columns 64 w .. 64 w + 63 of all 64
//     chains), three workgroups per CU with their own barriers;
//   * A in slabs of 128 K-columns: 512 contiguous bytes per chain and request round (a tile of 64 rows x
//     128 B streams at 3.9 TB/s on this machine, 64 x 512 B at 5.5: tools/probes/tile_stream_probe.hip), two
//     slabs deep in registers, rounded to 16 bit on the way into LDS, ONE barrier per slab; no predication
//     in the loop (row / slab indices are clamped instead, so that hipcc's s_waitcnt counts stay exact);
//   * W never goes through LDS: the W slab of 32 K-columns -- 16 KB, four times the A tile it multiplies --
//     is requested by each wavefront directly in the layout of the MFMA operand (v_mfma_f32_16x16x32: lane =
//     (row n & 15, k-group)), from a copy of W re-ordered slab-major ([K / 32][N][32], pack_w_kernel, once per
//     launch: 8-12 MB through L2) so that one dwordx4 instruction reads 1 KB contiguous.  A first version
//     staged W through LDS like the tile kernels do: with every global request and every MFMA removed its
//     loop still took 95 us of the 184 (LDS stores of W + barrier); requesting the row-major W directly
//     (16 row pieces of 64 B per instruction) cost 125 us for W alone against 87 slab-major;
//   * split-K with the split index = blockIdx % splits: one XCD sees one K-range of W (<= 3 MB at cfg-3,
//     resident in its 4 MB L2) and every A tile once;
//   * U1X (the xnet's [cos(m x), sin(m x)] input, dynamics.py:1161-1185): ONE request of a slab of links makes two
//     LDS tiles, cos and sin, multiplied by the W slabs K columns apart (the K-ordered loader of gemm_nt_h_kernel
//     read x from HBM twice: 805 MB per launch instead of 537; so did a first version of this kernel with the cos
//     and sin slabs adjacent in the K order -- 886 MB of fabric reads in the PMC pass: two slabs later x has left
//     the 4 MB L2 of an XCD that streams 64 workgroups x 32 KB per slab).
// Measured at cfg-3 (vnet / xnet input layer, kernel alone): 138 / 190 us (+ 7 us re-ordering W + 11 us split-K
// reduce) against gemm_nt_h_kernel's 243 / 420; by parts (SK_SKIP builds): loop skeleton 21 / 45 us, W + MFMA 74, A +
// MFMA 109, A + W without MFMA 105 (= the speed of a plain float4 read of A on this machine), everything 138.  A
// variant with the A stream on two producer wavefronts and the MFMAs on four consumers (so that W requests do not
// queue behind HBM misses in a wavefront's in-order vmcnt) was measured slower (149 / 194 us) and is not kept.
// fp32 partial sums go to part[splits][M][N]; splitk_reduce_h_kernel (gemm_f16.hip) adds them in fixed
// order and applies the epilogue with autocast's rounding points.  The accumulation order over K differs
// from gemm_nt_h_kernel's (fp32 accumulators, both inside the 16-bit rounding that follows).
#include "half_common.hpp"

namespace l2q {

#ifndef SK_OCC
#define SK_OCC 2           // workgroups per CU the register budget is set for
#endif
#ifndef SK_SKIP
#define SK_SKIP 0          // timing-only builds (tools/ab_build.sh): 1 no MFMA, 2 no W requests, 4 no A requests
#endif
#ifndef SK_RH
#define SK_RH 1            // row halves per workgroup: 1 = 64 chains x 4 wavefronts; 2 = 128 chains x 8 wavefronts (the two
                           // wavefronts of a column group request the same W slabs, half the L2 traffic of W if L1
                           // catches the second request) -- measured slower at cfg-3: 158 / 201 us against 150 / 191
#endif
constexpr int kSkRH = SK_RH, kSkBM = 64 * kSkRH, kSkBN = 256, kSkBK = 32, kSkNT = 256 * kSkRH;
constexpr int kSkBKA = 128, kSkSub = kSkBKA / kSkBK;     // A slab: 128 columns = 4 W slabs of 32
constexpr int kSkLD = kSkBKA + 16;           // LDS row stride 288 B: conflict-free ds_read_b128 of 16-row fragments

struct SkArgs {
  const float* A;          // [M][K]  (U1X: the link angles, K = xdim)
  const float* A2;         // [M][K2] or nullptr
  const void* Wp;          // slab-major copy of [W | W2]: [(Kw + K2) / 32][N][32], Kw = K (U1X: 2 K)
  const float* mask;       // U1X: [K] site mask
  int complement;
  int M, N;
  long K, K2;
  int splits;
  float* part;             // [splits][M][N]
};

// Wp[g][n][32] = [W | W2][n][32 g .. 32 g + 31]; one thread per 16-byte piece, k fastest (contiguous reads)
template <typename HT>
__global__ __launch_bounds__(kBlock) void pack_w_kernel(const HT* __restrict__ W, long Kw, const HT* __restrict__ W2,
                                                        long K2, int N, HT* __restrict__ Wp) {
  typedef HT hv8 __attribute__((ext_vector_type(8)));
  const long cpr = (Kw + K2) / 8;                       // pieces per row
  const long i = (long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= cpr * N) return;
  const long n = i / cpr, col = (i % cpr) * 8;
  const hv8 v = col < Kw ? *reinterpret_cast<const hv8*>(W + n * Kw + col)
                         : *reinterpret_cast<const hv8*>(W2 + n * K2 + (col - Kw));
  *reinterpret_cast<hv8*>(Wp + ((col / kSkBK) * N + n) * kSkBK + (col % kSkBK)) = v;
}

template <typename HT, bool U1X>
__global__ __launch_bounds__(kSkNT, SK_OCC) void gemm_skinny_h_kernel(SkArgs a) {
  using vec_t = typename MfmaH<HT>::vec_t;
  typedef HT hv4 __attribute__((ext_vector_type(4)));
  constexpr int NT = kSkNT, BM = kSkBM;
  constexpr int AR = NT / 32, AP = BM / AR;         // A slab: 32 lanes x 16 B per row, AP = 8 passes of AR = 8 rows
  constexpr int NTL = U1X ? 2 : 1;                  // LDS tiles one A slab turns into (U1X: cos and sin of the links)
  __shared__ __attribute__((aligned(16))) HT As[2][NTL][BM][kSkLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, grp = lane >> 4;
  const int wn = (wave & 3) * 64, rbase = (wave >> 2) * 64;
  const int S = a.splits;
  const long z = blockIdx.x % S, mt = blockIdx.x / S;
  const long m0 = mt * BM;
  const long q1 = a.K / ((long)kSkBKA * S), q2 = a.K2 / ((long)kSkBKA * S);  // this split's A slabs per segment
  const long nq = q1 + q2;
  // W slabs (32 columns) of this split in the order they are multiplied: segment-1 slab Q contributes NTL x 4 (U1X:
  // its four cos slabs, then the four sin slabs K columns further), segment-2 slabs 4 each
  const long nv1 = (long)NTL * kSkSub * q1, nv = nv1 + kSkSub * q2;
  const HT* Wp = (const HT*)a.Wp;
  const long kw = (long)NTL * a.K;                  // columns of W in front of W2 in the packed copy

  const int ac = (tid & 31) * 4;
  int wrow[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = wn + 16 * j + l15;
    wrow[j] = (n < a.N ? n : a.N - 1) * kSkBK + 8 * grp;
  }

  float4 ra[2][AP];
  vec_t rw[2][4];
  float4 rk[2];

  // A slab Q of this split: segment 1 (Q < q1), then segment 2
  auto fetch_a = [&](int e, long Q) {
    if (Q >= nq) Q = nq - 1;                       // past the end: the last slab again (nobody reads it)
    const bool s1 = Q < q1;
    const long acol = s1 ? (q1 * z + Q) * kSkBKA : (q2 * z + (Q - q1)) * kSkBKA;
    const float* Ab = s1 ? a.A : a.A2;
    const long lda = s1 ? a.K : a.K2;
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      if (SK_SKIP & 4) { ra[e][p] = make_float4((float)Q, 1.f, 2.f, (float)acol); continue; }
      long m = m0 + (tid >> 5) + p * AR;
      m = m < a.M ? m : a.M - 1;                   // clamped, not predicated: rows >= M are never stored
      ra[e][p] = *reinterpret_cast<const float4*>(Ab + m * lda + acol + ac);
    }
    if (U1X) rk[e] = *reinterpret_cast<const float4*>(a.mask + (s1 ? acol : 0) + ac);
  };
  // W slab v as MFMA operands: lane (row wn + 16 j + l15, k-group grp) holds k = 8 grp .. 8 grp + 7
  auto fetch_w = [&](int e, long v) {
    if (v >= nv) v = nv - 1;
    long wcol;                                     // first column of the slab in [W | W2]
    if (v < nv1) {
      const long Q = v / (NTL * kSkSub), r = v % (NTL * kSkSub);   // r >= 4: the sin slabs
      wcol = (q1 * z + Q) * kSkBKA + (r & 3) * kSkBK + (r >> 2) * a.K;
    } else {
      const long w = v - nv1;
      wcol = kw + (q2 * z + (w >> 2)) * kSkBKA + (w & 3) * kSkBK;
    }
    const HT* src = Wp + (wcol / kSkBK) * ((long)a.N * kSkBK);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (SK_SKIP & 2) {
#pragma unroll
        for (int q = 0; q < 8; ++q) rw[e][j][q] = (HT)(float)(v + q);
        continue;
      }
      rw[e][j] = *reinterpret_cast<const vec_t*>(src + wrow[j]);
    }
  };

  // registers of slab Q -> LDS buffer `buf`, rounded to 16 bit; U1X segment 1: tile 0 = cos(keep x), tile 1 = sin(keep x)
  // from ONE request of x
  auto store_a = [&](int e, long Q, int buf) {
    const bool trig = U1X && Q < q1;                               // wavefront-uniform
    float keep[4] = {1.f, 1.f, 1.f, 1.f};
    if (U1X) {
      const float m[4] = {rk[e].x, rk[e].y, rk[e].z, rk[e].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) keep[j] = a.complement ? 1.f - m[j] : m[j];
    }
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const float f[4] = {ra[e][p].x, ra[e][p].y, ra[e][p].z, ra[e][p].w};
      const int row = (tid >> 5) + p * AR;
      if (trig) {
        float c[4], sn[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { c[j] = __cosf(keep[j] * f[j]); sn[j] = __sinf(keep[j] * f[j]); }
        const hv4 hc = {(HT)c[0], (HT)c[1], (HT)c[2], (HT)c[3]};
        const hv4 hs = {(HT)sn[0], (HT)sn[1], (HT)sn[2], (HT)sn[3]};
        *reinterpret_cast<hv4*>(&As[buf][0][row][ac]) = hc;
        *reinterpret_cast<hv4*>(&As[buf][NTL - 1][row][ac]) = hs;
      } else {
        const hv4 h = {(HT)f[0], (HT)f[1], (HT)f[2], (HT)f[3]};
        *reinterpret_cast<hv4*>(&As[buf][0][row][ac]) = h;
      }
    }
  };

  // wavefront tile 64 chains x 64 columns = 4 x 4 MFMA tiles of 16 x 16; W is the MFMA "row" operand: lane owns
  // chain 16 i + l15 of tile row i and columns 16 j + 4 grp + r of tile column j
  v4f32 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (v4f32){0.f, 0.f, 0.f, 0.f};

  auto mfma = [&](int buf, int tile, int sub, int e) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const vec_t fa = *reinterpret_cast<const vec_t*>(&As[buf][tile][rbase + 16 * i + l15][kSkBK * sub + 8 * grp]);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = MfmaH<HT>::run(rw[e][j], fa, acc[i][j]);
    }
  };

  // A ring, entry Q & 1: slab Q until it has been stored to LDS buffer Q & 1 (during slab Q - 1), then slab
  // Q + 2.  W ring, entry v & 1: W slab v until it has been multiplied, then v + 2 (every A slab takes an even
  // number of W slabs, so the entry of its sub-slab `sub` is sub & 1).
  fetch_a(0, 0);
  fetch_a(1, 1);
  fetch_w(0, 0);
  fetch_w(1, 1);
  store_a(0, 0, 0);
  fetch_a(0, 2);
  __syncthreads();
  long v = 0;                                       // next W slab to multiply
  for (long qb = 0; qb < nq; qb += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long Q = qb + u;
      const int ntile = (U1X && Q < q1) ? 2 : 1;    // wavefront-uniform
      if (Q < nq) {
        for (int tile = 0; tile < ntile; ++tile) {
#pragma unroll
          for (int sub = 0; sub < kSkSub; ++sub) {
            if (!(SK_SKIP & 1)) mfma(u, U1X ? tile : 0, sub, sub & 1);
            fetch_w(sub & 1, v + 2);
            ++v;
          }
        }
      }
      store_a(u ^ 1, Q + 1, u ^ 1);
      fetch_a(u ^ 1, Q + 3);
      __syncthreads();
    }
  }

  const bool vecc = (a.N % 4) == 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long nb4 = wn + 16 * j + 4 * grp;
    if (nb4 >= a.N) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long m = m0 + rbase + 16 * i + l15;
      if (m >= a.M) continue;
      float* dst = a.part + (z * a.M + m) * a.N + nb4;
      if (vecc) {
        *reinterpret_cast<v4f32*>(dst) = acc[i][j];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (nb4 + r < a.N) dst[r] = acc[i][j][r];
      }
    }
  }
}

// splits the skinny kernel would use for this shape, 0: the shape is not its case.  `raw1`: K-columns of the
// first operand as stored (U1X: xdim).
int gemm_h_skinny_splits(int M, int N, long raw1, long K2, int want) {
  if (N > kSkBN || N < 64 || M < 1024 || raw1 % kSkBKA != 0 || K2 % kSkBKA != 0 || raw1 + K2 < 4096) return 0;
  const long s1 = raw1 / kSkBKA, s2 = K2 / kSkBKA;
  const long mt = cdiv(M, kSkBM);
  const long slots = 256L * 2 / kSkRH;                 // a full first round of workgroups (8 wavefronts per CU)
  int fits = 0, smallest = 0;
  for (int s = 8; s >= 1; s >>= 1) {
    if (s1 % s != 0 || s2 % s != 0) continue;
    if ((s1 + s2) / s < 4) continue;                  // too few slabs per split to amortise the pipeline fill
    if (s == want) return s;
    if (fits == 0 && mt * s <= slots) fits = s;       // the largest split count that is one round of workgroups
    smallest = s;
  }
  if (want > 0) return 0;
  return fits ? fits : smallest;
}

size_t gemm_h_skinny_ws_bytes(int M, int N, long K, long K2) {
  return (size_t)8 * M * N * sizeof(float) + (size_t)N * (K + K2) * 2;
}

// true: launched (partials in ws; the caller runs splitk_reduce_h_kernel).  false: not this kernel's case.
template <typename HT>
bool gemm_h_skinny_launch(const float* A, const void* W, int M, int N, long K, const float* A2,
                          const void* W2, long K2, void* ws, size_t ws_bytes, hipStream_t st,
                          const float* cs_mask, int cs_compl, int want, int* splits_out) {
  const long raw1 = cs_mask ? K / 2 : K;
  if (cs_mask && (K & 1)) return false;
  const int S = gemm_h_skinny_splits(M, N, raw1, K2, want);
  if (S == 0) return false;
  if (!al16(A) || !al16(A2) || !al16(W) || !al16(W2) || !al16(cs_mask)) return false;
  const size_t part_bytes = (size_t)S * M * N * sizeof(float);
  if (!ws || part_bytes + (size_t)N * (K + K2) * 2 > ws_bytes) return false;
  HT* Wp = (HT*)((char*)ws + part_bytes);
  const long pieces = (K + K2) / 8 * N;
  hipLaunchKernelGGL((pack_w_kernel<HT>), dim3((unsigned)cdiv(pieces, kBlock)), dim3(kBlock), 0, st,
                     (const HT*)W, K, (const HT*)W2, K2, N, Wp);
  SkArgs a;
  a.A = A; a.A2 = A2; a.Wp = Wp; a.mask = cs_mask; a.complement = cs_compl;
  a.M = M; a.N = N; a.K = raw1; a.K2 = K2; a.splits = S; a.part = (float*)ws;
  const dim3 grid((unsigned)(cdiv(M, kSkBM) * S));
  if (cs_mask)
    hipLaunchKernelGGL((gemm_skinny_h_kernel<HT, true>), grid, dim3(kSkNT), 0, st, a);
  else
    hipLaunchKernelGGL((gemm_skinny_h_kernel<HT, false>), grid, dim3(kSkNT), 0, st, a);
  *splits_out = S;
  return true;
}

template bool gemm_h_skinny_launch<_Float16>(const float*, const void*, int, int, long, const float*,
                                             const void*, long, void*, size_t, hipStream_t, const float*,
                                             int, int, int*);
template bool gemm_h_skinny_launch<__bf16>(const float*, const void*, int, int, long, const float*,
                                           const void*, long, void*, size_t, hipStream_t, const float*, int,
                                           int, int*);

}  // namespace l2q
