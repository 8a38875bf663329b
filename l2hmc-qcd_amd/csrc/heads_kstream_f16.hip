// heads_kstream_f16.hip -- the three 16-bit heads of a U(1) LeapfrogLayer + the sub-update that consumes them
// (the operation of u1_heads_update_h_kernel, gemm_f16.hip) as a STREAM over the chains with the weights
// stationary in registers and the K dimension split over wavefront pairs.  K = units[-1] = 256, 128 or 64.
//
// Why: the tile kernel stages 160 KB of Z and W through registers into LDS for every 128 x 64 tile (1.3 GB of
// L2 -> LDS traffic per launch against 805 MB of HBM traffic) and runs at 0.25 of the HBM roofline; round 3's
// stream kernel (u1_heads_stream_h_kernel) keeps a wavefront's 16 columns x 3 heads x K = 256 of W in 96
// VGPRs, which leaves no registers for a second stage of operand prefetch (0.475 ms against 0.42).  Here
//   * a workgroup = 8 wavefronts owns 64 entries (columns) and a range of chains.  Wavefront (cg, kh) keeps the
//     three heads' weights of columns 16 cg .. 16 cg + 15 for K-half kh: 3 x 4 MFMA operands = 48 VGPRs;
//   * the chains stream by in steps of 32 rows.  Per step a wavefront multiplies both 16-row tiles by its
//     K-half (24 MFMAs), hands the partial sums of the row tile it does NOT finish to its partner (cg, 1 - kh)
//     through LDS (12 floats per lane), adds the partner's, and runs the epilogue on 16 rows x 16 columns: 4
//     entries per lane;
//   * operand requests are two steps ahead: Z rows (L2) first, then the fp32 field operands (HBM) -- a
//     wavefront's loads return in order, so a Z request behind a field request would wait for HBM; field
//     operands sit in a three-slot register ring (8 VGPRs per slot), Z goes registers -> LDS one step before
//     it is multiplied (two 16 KB stages, chunk-swizzled: conflict-free ds_read_b128 fragments);
//   * ONE barrier per step: the epilogue of step s - 1 runs in the iteration that multiplies step s (the
//     partial sums wait in a second LDS buffer, the wavefront's own in registers), so the matrix pipe and the
//     VALU work at the same time.  (First version, two barriers per step and the epilogue behind the exchange:
//     0.285 / 0.408 ms per v- / x-update, of which 0.116 ms were barriers, LDS hand-overs and their latencies.)
//   * the epilogue handles two entries at a time on packed fp32 math (hh_element2, heads_h_common.hpp): the x-update
//     is bound by the VALU issue of its epilogue (97 instructions per entry, 10 of them quarter-rate, two wavefronts
//     per SIMD in lockstep); with packed conversions, the bare v_log_f32 and compare-select min / max: 399 -> 259 VALU
//     instructions per 4 entries, 0.324 -> 0.274 ms.
// Same MFMA instruction and operand roles as the other two kernels; the K = 256 sum is formed as (k < 128) +
// (k >= 128) in fp32 instead of one running accumulator: a rounding-level difference in front of the 16-bit
// rounding of the head (rare 1-ulp16 flips, as between the tile and the stream kernel).
#include <type_traits>
#include "heads_h_common.hpp"

namespace l2q {

#ifndef KS_PACKED
#define KS_PACKED 1        // epilogue on two entries at a time (packed fp32 math); 0: the scalar hh_element
#endif
#ifndef KS_NCG
#define KS_NCG 4           // column groups of 16 entries per workgroup (4: 8 wavefronts, one workgroup per CU; 2: 4
                           // wavefronts, two workgroups per CU with their own barriers -- measured slower at cfg-3,
                           // 0.259 / 0.347 ms against 0.216 / 0.323: 128-byte field rows and twice the Z staging)
#endif
constexpr int kKsNCG = KS_NCG, kKsCols = 16 * kKsNCG, kKsWaves = 2 * kKsNCG;
constexpr int kKsRows = 32, kKsNT = 64 * kKsWaves;

// KH: MFMA k-steps (of 32) per wavefront = K / 64: K = 256, 128, 64
template <typename HT, bool XUPD, bool FWD, bool NCP, int KH>
__global__ __launch_bounds__(kKsNT, 8 / kKsWaves) void u1_heads_kstream_h_kernel(HeadsHArgs a, int swz, int rows_per_wg) {
  using vec_t = typename MfmaH<HT>::vec_t;
  constexpr int KK = 64 * KH;                      // K
  constexpr int ROWS = kKsRows, ROWB = KK * 2, STAGE = ROWS * ROWB;
  __shared__ __attribute__((aligned(1024))) char zs[2 * STAGE];
  __shared__ __attribute__((aligned(16))) float4 xb[2][kKsWaves][3][64];   // partial sums for the partner wavefront
  __shared__ float red[2][kKsWaves][16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = wave % kKsNCG, kh = wave / kKsNCG;
  const int l15 = lane & 15, grp = lane >> 4;
  const long ntiles = (a.N + kKsCols - 1) / kKsCols;
  const long w = xcd_swizzle(blockIdx.x, gridDim.x, swz);
  const long n0 = (w % ntiles) * kKsCols;         // n-tiles fastest: neighbours walk the same rows
  const long mbeg = (w / ntiles) * rows_per_wg;
  long mend = mbeg + rows_per_wg;
  if (mend > a.M) mend = a.M;
  if (mbeg >= a.M) return;
  const int nstep = (int)((mend - mbeg + ROWS - 1) / ROWS);

  // ---- stationary operands: columns nw0 .. nw0 + 15, k = 128 kh .. 128 kh + 127
  const long nw0 = n0 + 16 * cg;
  const long ncol = nw0 + l15;
  const long nrow = ncol < a.N ? ncol : a.N - 1;
  vec_t wf[3][KH];
#pragma unroll
  for (int h = 0; h < 3; ++h) {
    const HT* W = (const HT*)a.W[h] + nrow * (long)KK + 32 * KH * kh + 8 * grp;
#pragma unroll
    for (int kk = 0; kk < KH; ++kk) wf[h][kk] = *reinterpret_cast<const vec_t*>(W + 32 * kk);
  }
  const long nb4 = nw0 + 4 * grp;                 // this lane's four consecutive entries
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

  // ---- Z stream: a step's 32 rows x 512 B; thread t carries 32 contiguous bytes (chunks 2 (t & 15), + 1) of row
  // t >> 4.  Chunk c of row r sits at slot c ^ (r & 15) of its row: the ds_read_b128 fragments (16 rows, one
  // chunk index) touch 16 different bank groups.
  const char* zbase = reinterpret_cast<const char*>(a.Z);
  constexpr int TPR = KK / 16;                     // threads per Z row (32 bytes each)
  constexpr int RPP = kKsNT / TPR;                 // rows per pass
  constexpr int ZP = RPP >= ROWS ? 1 : ROWS / RPP; // passes (1 | 2); RPP > ROWS: the surplus threads carry copies
  constexpr int SWM = (KK / 8 < 16 ? KK / 8 : 16) - 1;     // chunk swizzle mask (chunks per row: a power of two)
  const bool zact = tid / TPR < ROWS;
  const int zrow = zact ? tid / TPR : ROWS - 1, zc = 2 * (tid % TPR);
  uint4 zr0, zr1, zr2, zr3;                       // (scalars: hipcc moved a `uint4 zr[2]` to LDS and waited vmcnt(0)
                                                  // behind every request)
  auto zfetch = [&](int s) {
    long m = mbeg + (long)s * ROWS + zrow;
    if (m >= a.M) m = a.M - 1;                    // rows past the end re-read a valid one (never stored)
    const char* p = zbase + m * (long)ROWB + (zc << 4);
    zr0 = *reinterpret_cast<const uint4*>(p);
    zr1 = *reinterpret_cast<const uint4*>(p + 16);
    if (ZP == 2) {
      long m2 = mbeg + (long)s * ROWS + zrow + RPP;
      if (m2 >= a.M) m2 = a.M - 1;
      const char* p2 = zbase + m2 * (long)ROWB + (zc << 4);
      zr2 = *reinterpret_cast<const uint4*>(p2);
      zr3 = *reinterpret_cast<const uint4*>(p2 + 16);
    }
  };
  auto zstore = [&](int st) {
    char* row = zs + st * STAGE + zrow * ROWB;
    if (!zact) return;
    *reinterpret_cast<uint4*>(row + ((zc ^ (zrow & SWM)) << 4)) = zr0;
    *reinterpret_cast<uint4*>(row + (((zc + 1) ^ (zrow & SWM)) << 4)) = zr1;
    if (ZP == 2) {                                // row zrow + 16: the same swizzle key
      *reinterpret_cast<uint4*>(row + RPP * ROWB + ((zc ^ (zrow & SWM)) << 4)) = zr2;
      *reinterpret_cast<uint4*>(row + RPP * ROWB + (((zc + 1) ^ (zrow & SWM)) << 4)) = zr3;
    }
  };

  // ---- field operands of the row tile this wavefront finishes: chain 16 kh + l15 of the step, entries nb4 .. + 3
  float* __restrict__ pa = a.a;
  const float* __restrict__ pb = a.bsrc;
  float4 av[3], bv[3];
  auto fetch = [&](int slot, int s) {
    const long m = mbeg + (long)s * ROWS + 16 * kh + l15;
    const bool ok = ncok && m < mend;
    const long o = ok ? m * (long)a.N + nb4 : 0;  // (clamped, not predicated: a load behind a branch is waited
                                                  // for on the spot)
    if (L2Q_HH_SKIP & 4) { av[slot] = make_float4(0.1f, 0.2f, 0.3f, (float)s); bv[slot] = av[slot]; return; }
    av[slot] = *reinterpret_cast<const float4*>(pa + o);
    bv[slot] = *reinterpret_cast<const float4*>(pb + o);
  };
  const float eps = a.eps;
  const int frow = l15, fsw = l15 & SWM;          // fragment rows frow + 16 i; (row + 16 i) & SWM == row & SWM

  zfetch(0);
  fetch(0, 0);                                    // slot of step q: q % 3
  zstore(0);
  zfetch(1);

  // Software pipeline, ONE barrier per step: iteration s multiplies step s (MFMA on stage s & 1, partial sums of the
  // other row tile -> xb[s & 1]) and runs the epilogue of step s - 1 (partner's partial sums from xb[(s - 1) & 1] +
  // the ones carried in `mine`), so the matrix pipe works on step s while the VALU does the arithmetic of step
  // s - 1.  Field operands of step q are requested in iteration q - 1 and used in iteration q + 1.
  v4f32 mine[3];
#pragma unroll
  for (int h = 0; h < 3; ++h) mine[h] = (v4f32){0, 0, 0, 0};
  auto logdet_out = [&](int q) {                  // row sums of step q: the four column groups meet
    if (tid < ROWS) {
      const long m = mbeg + (long)q * ROWS + tid;
      if (m < mend) {
        const int pr = q & 1, hw = kKsNCG * (tid >> 4), rr = tid & 15;
        double x = (double)red[pr][hw][rr] + (double)red[pr][hw + 1][rr];
        if (kKsNCG == 4) x += (double)red[pr][hw + 2][rr] + (double)red[pr][hw + 3][rr];
        a.logdet_part[m * a.ncols_part + n0 / kKsCols] = x;
      }
    }
  };
  // SLOT: field slot of step s - 1 (compile time); MUL / EPI: the two halves of the iteration
  auto iter = [&](auto SLOT, auto MUL, auto EPI, int s) {
    constexpr int slot = decltype(SLOT)::value;
    constexpr bool mul = decltype(MUL)::value, epi = decltype(EPI)::value;
    __syncthreads();                              // stage (s + 1) & 1, xb[s & 1], red[(s - 1) & 1] free; xb[(s - 1) & 1] complete
    if (mul) {
      zstore((s + 1) & 1);
      zfetch(s + 2);                              // L2 first, then HBM (in-order return)
      fetch((slot + 2) % 3, s + 1);
    }
    if (s >= 2) logdet_out(s - 2);
    v4f32 acc[3][2];
    if (mul) {
#pragma unroll
      for (int h = 0; h < 3; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[h][i] = (v4f32){0, 0, 0, 0};
      const char* sb = zs + (s & 1) * STAGE;
#pragma unroll
      for (int kk = 0; kk < ((L2Q_HH_SKIP & 1) ? 0 : KH); ++kk) {
        vec_t fa[2];
        const int chunk = (4 * (KH * kh + kk) + grp) ^ fsw;
#pragma unroll
        for (int i = 0; i < 2; ++i)      // accumulator i = row tile i ^ kh: acc[h][0] is the tile this wavefront finishes
          fa[i] = *reinterpret_cast<const vec_t*>(sb + (frow + 16 * (i ^ kh)) * ROWB + (chunk << 4));
#pragma unroll
        for (int h = 0; h < 3; ++h)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[h][i] = MfmaH<HT>::run(wf[h][kk], fa[i], acc[h][i]);
      }
    }
    if (epi) {
      float pre[3][4];
#pragma unroll
      for (int h = 0; h < 3; ++h) {
        const float4 q = xb[(s - 1) & 1][wave ^ kKsNCG][h][lane];
        // (k < 128) + (k >= 128): fp32 addition commutes exactly, so both wavefronts of a pair form the same sum
#pragma unroll
        for (int r = 0; r < 4; ++r) pre[h][r] = mine[h][r] + (r == 0 ? q.x : r == 1 ? q.y : r == 2 ? q.z : q.w);
      }
      const long m = mbeg + (long)(s - 1) * ROWS + 16 * kh + l15;
      const bool ok = ncok && m < mend;
      const float4 ta = av[slot], tb = bv[slot];
      const float a4[4] = {ta.x, ta.y, ta.z, ta.w}, b4[4] = {tb.x, tb.y, tb.z, tb.w};
      const float okf = ok ? 1.f : 0.f;
      float out[4], ld = 0.f;
#if KS_PACKED
#pragma unroll
      for (int r = 0; r < 4; r += 2) {              // two entries per call: packed fp32 math (heads_h_common.hpp)
        v2f ldt;
        const v2f o = hh_element2<HT, XUPD, FWD, NCP>(
            L2Q_V2(pre[0][r], pre[0][r + 1]), L2Q_V2(pre[1][r], pre[1][r + 1]), L2Q_V2(pre[2][r], pre[2][r + 1]),
            L2Q_V2(bs[r], bs[r + 1]), L2Q_V2(bt[r], bt[r + 1]), L2Q_V2(bq[r], bq[r + 1]), L2Q_V2(cs[r], cs[r + 1]),
            L2Q_V2(cq[r], cq[r + 1]), a.st, eps, L2Q_V2(a4[r], a4[r + 1]), L2Q_V2(b4[r], b4[r + 1]),
            L2Q_V2(keep[r], keep[r + 1]), ldt);
        out[r] = o.x; out[r + 1] = o.y;
        ld = fmaf(okf, ldt.x + ldt.y, ld);
      }
#else
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float ldt;
        if (L2Q_HH_SKIP & 2) {
          out[r] = pre[0][r] + pre[1][r] + pre[2][r] + a4[r] + b4[r];
          ldt = out[r];
        } else {
          out[r] = hh_element<HT, XUPD, FWD, NCP>(pre[0][r], pre[1][r], pre[2][r], bs[r], bt[r], bq[r], cs[r],
                                                  cq[r], a.st, eps, a4[r], b4[r], keep[r], ldt);
        }
        ld = fmaf(okf, ldt, ld);
      }
#endif
      if ((L2Q_HH_SKIP & 4) && out[0] + out[1] + out[2] + out[3] != 12345.678f) {
      } else if (ok) *reinterpret_cast<float4*>(pa + m * (long)a.N + nb4) = make_float4(out[0], out[1], out[2], out[3]);
      // row sums over this wavefront's 16 columns (four lane groups of 4 entries)
      ld += __shfl_xor(ld, 16, 64);
      ld += __shfl_xor(ld, 32, 64);
      if (lane < 16) red[(s - 1) & 1][wave][lane] = ld;
    }
    if (mul) {
      // partial sums of the other row tile -> partner (cg, 1 - kh); this wavefront's own tile is carried
#pragma unroll
      for (int h = 0; h < 3; ++h) {
        xb[s & 1][wave][h][lane] = make_float4(acc[h][1][0], acc[h][1][1], acc[h][1][2], acc[h][1][3]);
        mine[h] = acc[h][0];
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using T = std::true_type;
  using F = std::false_type;
  iter(I2{}, T{}, F{}, 0);                        // (slot + 2) % 3 = 1: the request of step 1
  int s = 1;
  for (; s + 2 < nstep; s += 3) {
    iter(I0{}, T{}, T{}, s);
    iter(I1{}, T{}, T{}, s + 1);
    iter(I2{}, T{}, T{}, s + 2);
  }
  // s = 1 (mod 3) here: the remaining multiply iterations, then the epilogue of the last step
  if (s < nstep) { iter(I0{}, T{}, T{}, s); ++s; }
  if (s < nstep) { iter(I1{}, T{}, T{}, s); ++s; }
  switch ((nstep - 1) % 3) {
    case 0: iter(I0{}, F{}, T{}, nstep); break;
    case 1: iter(I1{}, F{}, T{}, nstep); break;
    default: iter(I2{}, F{}, T{}, nstep); break;
  }
  __syncthreads();
  logdet_out(nstep - 1);
}

// logdet[m] (+)= sum over the column blocks of part[m][0 .. ncols), one wavefront per chain, fixed order
__global__ __launch_bounds__(kBlock) void kstream_finalize_kernel(const double* __restrict__ part, int ncols, int M,
                                                                  float* __restrict__ logdet, int accumulate) {
  const int lane = threadIdx.x & 63;
  const long m = (long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (m >= M) return;
  double x = 0.0;
  for (int c = lane; c < ncols; c += 64) x += part[m * ncols + c];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
  if (lane == 0) logdet[m] = accumulate ? logdet[m] + (float)x : (float)x;
}

// true: launched.  false: not this kernel's case (K != 256, ragged N, unaligned operands)
template <typename HT>
bool heads_h_kstream_launch(HeadsHArgs a, int xupd, int forward, int use_ncp, int swz, float* logdet,
                            int accumulate, hipStream_t st, bool any_length) {
  // (short streams -- fewer than ~8 steps per workgroup -- stay on the tile kernel: the stationary weights and the
  // pipeline fill are paid per workgroup)
  if ((a.K != 256 && a.K != 128 && a.K != 64) || (a.N & 3) != 0) return false;
  if (!any_length && (a.M < 1024 || (long)a.M * cdiv(a.N, 64) < 65536)) return false;
  const long slots = 256L * (8 / kKsWaves);       // workgroups the chip runs at a time
  if (!al16(a.a) || !al16(a.bsrc) || !al16(a.b[0]) || !al16(a.b[1]) || !al16(a.b[2]) || !al16(a.cs) ||
      !al16(a.cq) || (xupd && !al16(a.mask)) || !al16(a.Z) || !al16(a.W[0]) || !al16(a.W[1]) || !al16(a.W[2]))
    return false;
  const long nt = cdiv(a.N, kKsCols);
  a.ncols_part = (int)nt;                         // one partial per (chain, column block), each written exactly once
  // one workgroup per CU at a time (8 wavefronts, > 128 VGPRs): rounds of 256
  long msplit = cdiv(slots, nt);
  if (nt < slots && slots % nt == 0) msplit = slots / nt;
  const long maxsplit = cdiv(a.M, 4 * kKsRows);
  if (msplit > maxsplit) msplit = maxsplit;
  if (msplit < 1) msplit = 1;
  const int rows_per_wg = (int)(cdiv(cdiv(a.M, msplit), kKsRows) * kKsRows);
  msplit = cdiv(a.M, rows_per_wg);
  const dim3 grid((unsigned)(nt * msplit)), block(kKsNT);
#define L2Q_KS(X, F, C)                                                                                             \
  do {                                                                                                              \
    if (a.K == 256) hipLaunchKernelGGL((u1_heads_kstream_h_kernel<HT, X, F, C, 4>), grid, block, 0, st, a, swz, rows_per_wg);      \
    else if (a.K == 128) hipLaunchKernelGGL((u1_heads_kstream_h_kernel<HT, X, F, C, 2>), grid, block, 0, st, a, swz, rows_per_wg); \
    else hipLaunchKernelGGL((u1_heads_kstream_h_kernel<HT, X, F, C, 1>), grid, block, 0, st, a, swz, rows_per_wg);                 \
  } while (0)
  if (!xupd) { if (forward) L2Q_KS(false, true, false); else L2Q_KS(false, false, false); }
  else if (use_ncp) { if (forward) L2Q_KS(true, true, true); else L2Q_KS(true, false, true); }
  else { if (forward) L2Q_KS(true, true, false); else L2Q_KS(true, false, false); }
#undef L2Q_KS
  hipLaunchKernelGGL(kstream_finalize_kernel, dim3((unsigned)cdiv(a.M, kBlock / 64)), dim3(kBlock), 0, st,
                     (const double*)a.logdet_part, a.ncols_part, a.M, logdet, accumulate);
  return true;
}

template bool heads_h_kstream_launch<_Float16>(HeadsHArgs, int, int, int, int, float*, int, hipStream_t, bool);
template bool heads_h_kstream_launch<__bf16>(HeadsHArgs, int, int, int, int, float*, int, hipStream_t, bool);

}  // namespace l2q
