// conv_stream_f16.hip -- half-precision periodic convolution as an implicit GEMM by PERSISTENT workgroups
// with the weights resident in LDS and whole-K pixel tiles prefetched two tiles ahead (round 3; the
// wide layers of the U(1) ConvStack, network.py:151-172, 283-326, optionally with MaxPool2d(2) + the
// activation fused).  Dispatched from l2q_conv_gemm_periodic_h / l2q_conv_pool_gemm_periodic_h.
//
// Why: conv_gemm_h_kernel (gemm_f16.hip) walks K = C k k (144 .. 288 for the default stack) in slabs
// of 64 with the next slab requested only one slab ahead, and a slab's MFMA work is ~0.25 us: the
// ~2 us gather latency is exposed four to five times per 128-pixel tile, plus the row-table prologue
// and the store epilogue -- 18 us per tile at two workgroups per CU where the MFMAs need 0.9 us
// (cfg-3, 128-channel layer: 3.3 ms; three workgroups per CU: 2.6 ms).  Here
//   * a workgroup lives for the whole launch: the weight matrix [cout][K] is loaded into LDS once;
//   * a tile is 64 GEMM rows (pixels, or -- pooled -- 16 pooling windows x 4) over ALL of K:
//     64 K / 8 sixteen-byte gathers, K / 32 per thread, whose (row, tap, channel group) never changes
//     from tile to tile, so tap offsets are computed once per launch;
//   * the gathers of tile t + 2 are issued as soon as tile t's registers have been written to LDS:
//     every load has two tile times to arrive;
//   * per tile: one LDS store pass, two barriers, K / 32 MFMA steps straight from LDS, epilogue.
// Same MFMA instruction, operand roles and k order as conv_gemm_h_kernel (zero padding adds exact
// zeros): identical bits.
// MEASURED (cfg-3, fp16): pooled 128-channel layer 3.7 ms, un-pooled 64-channel layer 1.8 ms against 2.6 /
// 1.2 ms of the gather kernel at three workgroups per CU: with the weights resident (68 KB) only one
// workgroup of four wavefronts fits a CU, and the ~1700 instructions of address arithmetic, predication
// and epilogue per 64-row tile then run on one wavefront per SIMD (first version, with the libm
// activations inlined 32 x and spills sharing vmcnt with the prefetch: 5.4 ms).  Kept as tuning
// `conv_stream = 1`, off by default; next step would be 512-thread workgroups.  16-bit NHWC input with C % 8 == 0 and (i, j, ci) weight order only.
#include <type_traits>
#include "half_common.hpp"

namespace l2q {

struct ConvStreamArgs {
  const void* in;
  const void* Wt;
  const float* bias;
  void* out;
  ConvGeomH g;
  long ntiles;
  int N, act, ksteps, KP;      // KP: LDS row stride in halves (32 ksteps + 8)
  int nv;                      // 16-byte vectors per GEMM row (4 ksteps)
  int off_a, off_c, off_t;     // byte offsets of the A tile, the output tile and the row tables
};

extern __shared__ __attribute__((aligned(16))) char cs_lds[];

// The persistent loop body must stay inside the instruction cache: with act_h's libm branches (tanhf, expm1f,
// expf) inlined into the 32-element unrolled epilogue of both tile instances it was 17 000 instructions.
// max over the four lanes of a quad: two DPP quad_perm moves (xor 1: [1,0,3,2], xor 2: [2,3,0,1])
__device__ __forceinline__ float cs_quad_max(float v) {
  int x = __float_as_int(v);
  float o = __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false));
  v = fmaxf(v, o);
  x = __float_as_int(v);
  o = __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false));
  return fmaxf(v, o);
}
__device__ __noinline__ float cs_act_slow(float z, int act) { return act_h(z, act); }
__device__ __forceinline__ float cs_act(float z, int act) {
  if (act == L2Q_ACT_NONE) return z;
  if (act == L2Q_ACT_RELU) return z > 0.f ? z : 0.f;
  if (act == L2Q_ACT_LEAKY_RELU) return z > 0.f ? z : 0.01f * z;
  return cs_act_slow(z, act);
}

constexpr int kCsBM = 64;      // GEMM rows per tile
constexpr int kCsQ = 10;       // gathers per thread and tile at most: K <= 320 (template QN: 4, 6, 8, 10)

// (BN * QN >= 512: the resident weights + A tile leave room for one workgroup per CU anyway -- compiled for
// one wavefront per SIMD, 512 registers: a spill here is fatal, scratch traffic shares vmcnt with the prefetch)
template <typename HT, int BN, bool POOL, int QN>
__global__ __launch_bounds__(kBlock, BN * QN >= 512 ? 1 : 2) void conv_stream_h_kernel(ConvStreamArgs a) {
  constexpr int BM = kCsBM;
  constexpr int WN = BN >= 64 ? 2 : 1, WM = 4 / WN;          // wavefront grid
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16;
  using vec_t = typename MfmaH<HT>::vec_t;
  HT* Ws = reinterpret_cast<HT*>(cs_lds);
  HT* As = reinterpret_cast<HT*>(cs_lds + a.off_a);
  HT* Cs = reinterpret_cast<HT*>(cs_lds + a.off_c);
  long* rbase = reinterpret_cast<long*>(cs_lds + a.off_t);               // [2][BM]
  int* rr0 = reinterpret_cast<int*>(cs_lds + a.off_t + 2 * BM * 8);     // [2][BM]
  int* rc0 = rr0 + 2 * BM;
  const ConvGeomH& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave / WN) * TM, wn = (wave % WN) * TN;
  const int k = g.k, KP = a.KP, N = a.N, nv = a.nv;
  const HT* in = (const HT*)a.in;
  HT* C = (HT*)a.out;

  // ---- weights -> LDS [BN][KP], zero beyond (N, Kc)
  {
    const HT* Wt = (const HT*)a.Wt;
    for (int idx = tid; idx < BN * nv; idx += kBlock) {
      const int n = idx / nv, kv = (idx - n * nv) * 8;
      vec_t v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (HT)0.f;
      if (n < N && kv < g.Kc) v = *reinterpret_cast<const vec_t*>(Wt + (long)n * g.Kc + kv);
      *reinterpret_cast<vec_t*>(Ws + n * KP + kv) = v;
    }
  }
  // ---- this thread's gathers: vector v = tid + 256 q of a tile = (row v / nv, k = 8 (v % nv)): fixed
  int grow[QN], gtap[QN], glds[QN];                // gtap: tap row | tap column << 8 | channel offset << 16;
                                                   // glds: the vector's place in the LDS tile (halves)
#pragma unroll
  for (int q = 0; q < QN; ++q) {
    const int v = tid + kBlock * q;
    int row = v / nv;
    const int kk = (v - row * nv) * 8;
    bool ok = row < BM && kk < g.Kc;
    if (!ok) row = -1;
    const int kc = ok ? kk : 0;
    const int ij = kc / g.C;                       // K order (i, j, ci)
    grow[q] = row;
    const int di = ij / k, dj = ij - di * k, co = kc - ij * g.C;
    gtap[q] = di | (dj << 8) | (co << 16);
    glds[q] = (v / nv) * KP + kk;
  }

  // row table of a tile (64 threads): image base and the wrapped top-left tap of each GEMM row
  auto rows = [&](long tile, int slot) {
    if (tid < BM) {
      const long m = tile * BM + tid;
      long base = -1;
      int r0 = 0, c0 = 0;
      if (m < g.M) {
        const unsigned mu = (unsigned)m;
        int ho, wo;
        unsigned bq;
        if (POOL) {
          const unsigned P = mu >> 2, d = mu & 3;
          const unsigned t = P / (unsigned)g.Wp;
          bq = t / (unsigned)g.Hp;
          ho = 2 * (int)(t - bq * (unsigned)g.Hp) + (int)(d >> 1);
          wo = 2 * (int)(P - t * (unsigned)g.Wp) + (int)(d & 1);
        } else {
          const unsigned t = mu / (unsigned)g.Wo;
          wo = (int)(mu - t * (unsigned)g.Wo);
          bq = t / (unsigned)g.Ho;
          ho = (int)(t - bq * (unsigned)g.Ho);
        }
        base = (long)bq * g.sn;
        r0 = (ho - (k - 1)) % g.H; if (r0 < 0) r0 += g.H;
        c0 = (wo - (k - 1)) % g.W; if (c0 < 0) c0 += g.W;
      }
      rbase[slot * BM + tid] = base; rr0[slot * BM + tid] = r0; rc0[slot * BM + tid] = c0;
    }
  };
  // Gathers are UNCONDITIONAL (a load behind a branch is waited for on the spot by hipcc, which would undo the
  // prefetch): vectors past the tile / past K / of rows past M read a valid address and are zeroed when the
  // registers are written to LDS.
  vec_t areg[2][QN];
  auto gather = [&](auto SLOT, int tslot) {
    constexpr int slot = decltype(SLOT)::value;
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      const int row = grow[q] >= 0 ? grow[q] : 0;
      const long base = rbase[tslot * BM + row];
      int r = rr0[tslot * BM + row] + (gtap[q] & 255); if (r >= g.H) r -= g.H;
      int c = rc0[tslot * BM + row] + ((gtap[q] >> 8) & 255); if (c >= g.W) c -= g.W;
      const HT* src = in + (base >= 0 ? base : 0) + (gtap[q] >> 16) + (long)r * g.sh + (long)c * g.sw;
      areg[slot][q] = *reinterpret_cast<const vec_t*>(src);
    }
  };
  auto stash = [&](auto SLOT, long t) {
    constexpr int slot = decltype(SLOT)::value;
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      if (tid + kBlock * q < BM * nv) {
        vec_t x = areg[slot][q];
        if (grow[q] < 0 || t * BM + grow[q] >= g.M) { // (no table read: its slot is being rewritten)
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = (HT)0.f;
        }
        *reinterpret_cast<vec_t*>(As + glds[q]) = x;
      }
    }
  };

  const long t0 = blockIdx.x, dt = gridDim.x;
  // prologue: row tables of the first two tiles, their gathers
  rows(t0, 0);
  rows(t0 + dt, 1);
  __syncthreads();
  gather(std::integral_constant<int, 0>{}, 0);
  gather(std::integral_constant<int, 1>{}, 1);

  auto tile = [&](auto SLOT, long t) {
    constexpr int slot = decltype(SLOT)::value;
    __syncthreads();                                   // previous tile: fragment reads / table reads done
    stash(SLOT, t);                                    // this tile's gathers (issued two tiles ago) -> LDS
    rows(t + 2 * dt, slot);                            // row table of the tile after next (same slot)
    __syncthreads();
    if (t + 2 * dt < a.ntiles) gather(SLOT, slot);     // ... and its gathers: in flight for two tiles
    v4f32 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = (v4f32){0, 0, 0, 0};
#pragma unroll 2
    for (int kk = 0; kk < a.ksteps; ++kk) {
      const int kq = 32 * kk + 8 * (lane >> 4);
      vec_t fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        fa[i] = *reinterpret_cast<const vec_t*>(As + (wm + 16 * i + (lane & 15)) * KP + kq);
#pragma unroll
      for (int j = 0; j < NI; ++j)
        fb[j] = *reinterpret_cast<const vec_t*>(Ws + (wn + 16 * j + (lane & 15)) * KP + kq);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = MfmaH<HT>::run(fb[j], fa[i], acc[i][j]);
    }
    // ---- epilogue (W was the MFMA row operand: lane owns row lane & 15 of tile i, channels 4 (lane >> 4) + r)
    const long m0 = t * BM;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const long nb4 = wn + 16 * j + 4 * (lane >> 4);
      float cb[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) cb[r] = (a.bias && nb4 < N) ? a.bias[nb4 + r < N ? nb4 + r : N - 1] : 0.f;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int ml = wm + 16 * i + (lane & 15);
        const long m = m0 + ml;
        typedef HT cv __attribute__((ext_vector_type(4)));
        cv o;
        int lrow;                                      // output pixel (pooled or not) within the tile
        if (POOL) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = rnd<HT>(acc[i][j][r] + cb[r]);
            o[r] = (HT)cs_quad_max(v);                 // activation: in the copy loop below
          }
          if ((lane & 3) != 0) continue;
          lrow = ml >> 2;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (HT)(acc[i][j][r] + cb[r]);   // r16(acc + bias); act below
          lrow = ml;
        }
        if (m >= g.M || nb4 >= N) continue;
        *reinterpret_cast<cv*>(Cs + (long)lrow * N + nb4) = o;        // cout % 8 == 0 (launch): always staged
      }
    }
    {
      __syncthreads();
      long nrows = g.M - m0 < BM ? g.M - m0 : BM;
      long row0 = m0;
      if (POOL) { nrows >>= 2; row0 >>= 2; }
      const long total = nrows * N;                    // halves, multiple of 8: one contiguous range
      typedef HT v8 __attribute__((ext_vector_type(8)));
      HT* dst = C + row0 * N;
      // r16(act(.)) of the staged (rounded, pooled) values on the way out: ONE copy of the activation code
      // per tile instance instead of one per accumulator element
      for (long idx = (long)tid * 8; idx < total; idx += (long)kBlock * 8) {
        v8 x = *reinterpret_cast<const v8*>(Cs + idx);
        if (a.act != L2Q_ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = (HT)cs_act((float)x[e], a.act);
        }
        *reinterpret_cast<v8*>(dst + idx) = x;
      }
    }
  };
  for (long t = t0; t < a.ntiles; t += 2 * dt) {
    tile(std::integral_constant<int, 0>{}, t);
    if (t + dt < a.ntiles) tile(std::integral_constant<int, 1>{}, t + dt);
  }
}

// stream kernel when the layer fits it; false -> the caller uses the gather kernel
template <typename HT>
bool conv_stream_launch(const void* in, const ConvGeomH& g, const void* w, const float* bias, int cout,
                        int act, void* out, hipStream_t st) {
  if (g.C % 8 != 0 || !g.clast || g.sc != 1 || g.Kc > 32 * kCsQ || cout > 128) return false;
  if (g.sw % 8 != 0 || g.sh % 8 != 0 || g.sn % 8 != 0 || !al16(in) || !al16(w) || !al16(out)) return false;
  if (g.M >= (1L << 31) || g.k - 1 > g.H || g.k - 1 > g.W || cout % 8 != 0) return false;
  ConvStreamArgs a;
  a.in = in; a.Wt = w; a.bias = bias; a.out = out; a.g = g; a.N = cout; a.act = act;
  a.ksteps = (int)cdiv(g.Kc, 32);
  a.KP = 32 * a.ksteps + 8;
  a.nv = 4 * a.ksteps;
  a.ntiles = cdiv(g.M, kCsBM);
  const int bn = cout <= 32 ? 32 : cout <= 64 ? 64 : 128;
  const size_t wbytes = (size_t)bn * a.KP * 2, abytes = (size_t)kCsBM * a.KP * 2;
  const size_t cbytes = (size_t)(g.pool == 2 ? kCsBM / 4 : kCsBM) * bn * 2;
  a.off_a = (int)wbytes;
  a.off_c = a.off_a + (int)abytes;
  a.off_t = a.off_c + (int)((cbytes + 15) & ~(size_t)15);
  const size_t lds = (size_t)a.off_t + 2 * kCsBM * (8 + 4 + 4);
  if (lds > 150 * 1024) return false;
  int per_cu = (int)((160 * 1024) / lds);
  per_cu = per_cu < 1 ? 1 : per_cu > 2 ? 2 : per_cu;        // compiled for two wavefronts per SIMD
  const long nwg = a.ntiles < 256L * per_cu ? a.ntiles : 256L * per_cu;
#define L2Q_CS4(BNV, PV, QV)                                                                         \
  do {                                                                                               \
    static PerDeviceOnce attr_once;                                                                  \
    if (attr_once.first())                                                                           \
      (void)hipFuncSetAttribute((const void*)conv_stream_h_kernel<HT, BNV, PV, QV>,                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);             \
    hipLaunchKernelGGL((conv_stream_h_kernel<HT, BNV, PV, QV>), dim3((unsigned)nwg), dim3(kBlock), lds, st, a); \
  } while (0)
#define L2Q_CS(BNV, PV)                                                  \
  do {                                                                   \
    if (a.ksteps <= 4) L2Q_CS4(BNV, PV, 4);                              \
    else if (a.ksteps <= 6) L2Q_CS4(BNV, PV, 6);                         \
    else if (a.ksteps <= 8) L2Q_CS4(BNV, PV, 8);                         \
    else L2Q_CS4(BNV, PV, 10);                                           \
  } while (0)
  if (g.pool == 2) {
    if (bn == 32) L2Q_CS(32, true); else if (bn == 64) L2Q_CS(64, true); else L2Q_CS(128, true);
  } else {
    if (bn == 32) L2Q_CS(32, false); else if (bn == 64) L2Q_CS(64, false); else L2Q_CS(128, false);
  }
#undef L2Q_CS
#undef L2Q_CS4
  return true;
}

template bool conv_stream_launch<_Float16>(const void*, const ConvGeomH&, const void*, const float*, int, int,
                                           void*, hipStream_t);
template bool conv_stream_launch<__bf16>(const void*, const ConvGeomH&, const void*, const float*, int, int,
                                         void*, hipStream_t);

}  // namespace l2q
