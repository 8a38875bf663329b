// conv_patch_f16.hip -- half-precision periodic convolution with the input patch staged in LDS by
// persistent workgroups (the conv stack of the U(1) networks, BASELINE cfg-3 with the reference's default
// network; network.py:151-172, 283-326).  Dispatched from l2q_conv_gemm_periodic_h (gemm_f16.hip).
#include "half_common.hpp"

namespace l2q {

// ---------------------------------------------------------------------------------------------
// The same convolution with the input staged ONCE per output tile (PeriodicPadding(k-1) -> Conv2d(k),
// network.py:151-172, 283-326; 16-bit NHWC input, C a power of two >= 8, cout % 4 == 0).
//
// conv_gemm_h_kernel gathers every A element from global memory once per tap: k^2 16-byte requests
// per input pixel and channel group (25 for the first layer), all through the vector L1.  Here a
// persistent workgroup
//   * keeps the whole weight matrix [cout][K] in LDS for its lifetime,
//   * per tile of 64 MI consecutive output pixels of ONE image loads the (rows + k - 1) x (Wo + k - 1)
//     input patch it needs -- periodic wrap resolved at load time -- with coalesced 16-byte loads,
//   * forms the MFMA pixel fragments straight from the patch: the 8 consecutive K entries of a lane
//     are 8 channels of one tap (K order (i, j, ci), C % 8 == 0), i.e. one ds_read_b128 at
//     patch[(row + i) * PW + (col + j)][ci] -- no A tile, no barriers inside the K loop.
// Pixel stride in the patch is C + 8 halves for C >= 16 (16 lanes x 16 B at stride 2 C bytes would
// hit 2-8 banks), weight row stride K32 + 8 halves: both fragment reads are conflict-free.
struct ConvPatchArgs {
  const void* in;
  const void* Wt;
  const float* bias;
  void* out;
  long sn;              // halves per image
  long ntiles;
  int C, cshift;        // input channels (power of two), log2(C / 8)
  int H, W, Ho, Wo, k, K, N, act;
  int tiles_per_img, PW, CP, KP, ksteps;
  int off_patch, off_tab;   // byte offsets of the patch and the tap table in dynamic LDS
};

extern __shared__ __attribute__((aligned(16))) char cp_lds[];

template <typename HT, int BN, int MI>
__global__ __launch_bounds__(kBlock, BN >= 128 ? 2 : 3) void conv_patch_h_kernel(ConvPatchArgs a) {
  constexpr int NI = BN / 16;
  constexpr int TP = 64 * MI;                             // output pixels per tile (16 MI per wavefront)
  constexpr int PRE = 8;                                  // patch vectors a thread carries in registers
  using vec_t = typename MfmaH<HT>::vec_t;
  HT* Ws = reinterpret_cast<HT*>(cp_lds);
  HT* patch = reinterpret_cast<HT*>(cp_lds + a.off_patch);
  int* tab = reinterpret_cast<int*>(cp_lds + a.off_tab);           // [ksteps][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 4, l15 = lane & 15;
  const HT* in = (const HT*)a.in;
  const HT* Wt = (const HT*)a.Wt;
  HT* out = (HT*)a.out;
  const int k = a.k, C = a.C, CP = a.CP, PW = a.PW, KP = a.KP;
  // weights: [BN][KP], zero beyond (N, K)
  {
    const int vpr = KP / 8;
    for (int idx = tid; idx < BN * vpr; idx += kBlock) {
      const int n = idx / vpr, kv = (idx - n * vpr) * 8;
      vec_t v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (HT)0.f;
      if (n < a.N && kv < a.K) v = *reinterpret_cast<const vec_t*>(Wt + (long)n * a.K + kv);
      *reinterpret_cast<vec_t*>(Ws + n * KP + kv) = v;
    }
    // tap table: patch offset (halves) of the 8 K entries starting at kq = 32 s + 8 g
    for (int idx = tid; idx < a.ksteps * 4; idx += kBlock) {
      const int kq = 32 * (idx >> 2) + 8 * (idx & 3);
      int off = 0;
      if (kq < a.K) {
        const int t = kq / C, ci = kq - t * C;
        const int it = t / k, jt = t - it * k;
        off = (it * PW + jt) * CP + ci;
      }
      tab[idx] = off;
    }
  }
  const int HoWo = a.Ho * a.Wo;
  const int cmask = (1 << a.cshift) - 1;
  // patch of tile `t`: vector `idx` -> (source address, LDS address); up to PRE vectors per thread
  // travel through registers (requested before the K loop of the previous tile), the rest --
  // patches larger than PRE * 256 vectors -- are copied after it
  auto geom = [&](long t, long& img, int& p0, int& ho_min, int& nvec) {
    img = t / a.tiles_per_img;
    p0 = (int)(t - img * a.tiles_per_img) * TP;
    const int plast = min(p0 + TP - 1, HoWo - 1);
    ho_min = p0 / a.Wo;
    nvec = ((plast / a.Wo - ho_min + k) * PW) << a.cshift;
  };
  auto src_of = [&](long img, int ho_min, int idx) {
    const int cv = idx & cmask, pix = idx >> a.cshift;
    const int pr = pix / PW, pc = pix - pr * PW;
    int r = ho_min - (k - 1) + pr; r += r < 0 ? a.H : r >= a.H ? -a.H : 0;    // k - 1 <= H, W (host check)
    int c = pc - (k - 1); c += c < 0 ? a.W : c >= a.W ? -a.W : 0;
    return in + img * a.sn + ((long)r * a.W + c) * C + cv * 8;
  };
  auto dst_of = [&](int idx) { return patch + (long)(idx >> a.cshift) * CP + (idx & cmask) * 8; };
  vec_t pre[PRE];
  long tile = blockIdx.x;
  if (tile < a.ntiles) {
    long img; int p0, ho_min, nvec;
    geom(tile, img, p0, ho_min, nvec);
#pragma unroll
    for (int q = 0; q < PRE; ++q) {
      const int idx = tid + q * kBlock;
      if (idx < nvec) pre[q] = *reinterpret_cast<const vec_t*>(src_of(img, ho_min, idx));
    }
  }
  for (; tile < a.ntiles; tile += gridDim.x) {
    long img; int p0, ho_min, nvec;
    geom(tile, img, p0, ho_min, nvec);
    __syncthreads();                                      // the previous tile's LDS traffic is done
#pragma unroll
    for (int q = 0; q < PRE; ++q) {
      const int idx = tid + q * kBlock;
      if (idx < nvec) *reinterpret_cast<vec_t*>(dst_of(idx)) = pre[q];
    }
    for (int idx = tid + PRE * kBlock; idx < nvec; idx += kBlock)
      *reinterpret_cast<vec_t*>(dst_of(idx)) = *reinterpret_cast<const vec_t*>(src_of(img, ho_min, idx));
    __syncthreads();
    {                                                     // request the next tile's patch
      const long tn = tile + gridDim.x;
      if (tn < a.ntiles) {
        long img2; int p02, ho2, nv2;
        geom(tn, img2, p02, ho2, nv2);
#pragma unroll
        for (int q = 0; q < PRE; ++q) {
          const int idx = tid + q * kBlock;
          if (idx < nv2) pre[q] = *reinterpret_cast<const vec_t*>(src_of(img2, ho2, idx));
        }
      }
    }
    int pbase[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int p = min(p0 + wave * (16 * MI) + 16 * i + l15, HoWo - 1);
      const int ho = p / a.Wo, wo = p - ho * a.Wo;
      pbase[i] = ((ho - ho_min) * PW + wo) * CP;
    }
    v4f32 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = (v4f32){0, 0, 0, 0};
    const HT* wrow = Ws + l15 * KP + 8 * grp;
#pragma unroll 1
    for (int s = 0; s < a.ksteps; ++s) {
      const int off = tab[4 * s + grp];
      vec_t fb[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[j] = *reinterpret_cast<const vec_t*>(wrow + 16 * j * KP + 32 * s);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const vec_t fa = *reinterpret_cast<const vec_t*>(patch + pbase[i] + off);
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = MfmaH<HT>::run(fb[j], fa, acc[i][j]);
      }
    }
    // lane owns pixel l15 of tile i, channels 16 j + 4 grp + r.  The tile's pixels x N channels
    // are ONE contiguous range of the NHWC output: wide layers (N >= 32, where a wavefront's 8-byte
    // pieces are 32-byte runs 2 N bytes apart) assemble it in LDS -- over the patch, which is no
    // longer needed -- and write it as a flat 16-byte-per-lane stream
    const bool staged = BN >= 32 && (a.N & 7) == 0;
    const int SP = a.N + 4;                               // staged pixel stride (halves): spreads the banks
    HT* dst0 = out + (img * (long)HoWo + p0) * a.N;
    if (staged) __syncthreads();                          // every wavefront is done with the patch
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int nb4 = 16 * j + 4 * grp;
      if (nb4 >= a.N) continue;
      float cb[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) cb[r] = a.bias ? a.bias[nb4 + r] : 0.f;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int pl = wave * (16 * MI) + 16 * i + l15;
        if (p0 + pl >= HoWo) continue;
        typedef HT cv4 __attribute__((ext_vector_type(4)));
        cv4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (HT)epilogue_h<HT>(acc[i][j][r], cb[r], 1.f, false, a.act);
        if (staged) *reinterpret_cast<cv4*>(patch + pl * SP + nb4) = o;
        else *reinterpret_cast<cv4*>(dst0 + (long)pl * a.N + nb4) = o;
      }
    }
    if (staged) {
      __syncthreads();
      const int rows = min(TP, HoWo - p0);
      const int vpp = a.N >> 3;                           // 16-byte vectors per pixel
      for (int idx = tid; idx < rows * vpp; idx += kBlock) {
        const int pl = idx / vpp, cv = idx - pl * vpp;
        typedef HT cv4 __attribute__((ext_vector_type(4)));
        // (the staged stride N + 4 keeps 8-byte alignment only)
        const cv4 lo = *reinterpret_cast<const cv4*>(patch + pl * SP + cv * 8);
        const cv4 hi = *reinterpret_cast<const cv4*>(patch + pl * SP + cv * 8 + 4);
        vec_t v;
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
        *reinterpret_cast<vec_t*>(dst0 + (long)idx * 8) = v;
      }
    }
  }
}

// LDS-patch kernel when the layer fits it; false -> the caller uses the gather kernel
template <typename HT>
bool conv_patch_launch(const void* in, const ConvGeomH& g, const void* w, const float* bias,
                              int cout, int act, void* out, hipStream_t st) {
  const int C = g.C;
  if (C < 8 || (C & (C - 1)) != 0 || C > 128 || cout % 4 != 0 || cout > 128) return false;
  // Measured at cfg-3 (8192 chains, 64 x 64, filters [8, 16, 32, 64, 128]; profiles/r02_conv_patch_ab.txt):
  // layers with up to 16 output channels 1.06 ms against 2.04-2.35 ms of the gather kernel; wider layers
  // 2.5 / 1.7 / 3.9 ms against 2.0 / 1.5 / 3.4 ms -- those stay with the gather kernel unless
  // tuning "conv_patch" = 2 asks for this one everywhere it fits (tests do).
  if (cout > 16 && tuning().conv_patch < 2) return false;
  if (g.sw != C || g.sh != (long)g.W * C || !al16(w) || !al16(out)) return false;
  const int HoWo = g.Ho * g.Wo;
  ConvPatchArgs a;
  a.in = in; a.Wt = w; a.bias = bias; a.out = out; a.sn = g.sn;
  a.C = C; a.cshift = 0;
  while ((8 << a.cshift) < C) ++a.cshift;
  a.H = g.H; a.W = g.W; a.Ho = g.Ho; a.Wo = g.Wo; a.k = g.k; a.K = g.Kc; a.N = cout; a.act = act;
  const int bn = cout <= 16 ? 16 : cout <= 32 ? 32 : cout <= 64 ? 64 : 128;
  const int mi = bn <= 32 ? 8 : bn == 64 ? 2 : 2, tp = 64 * mi;
  if (g.k - 1 > g.H || g.k - 1 > g.W) return false;
  a.tiles_per_img = (int)cdiv(HoWo, tp);
  a.ntiles = (long)(g.M / HoWo) * a.tiles_per_img;
  a.PW = g.Wo + g.k - 1;
  a.CP = C >= 16 ? C + 8 : C;
  a.ksteps = (int)cdiv(g.Kc, 32);
  a.KP = a.ksteps * 32 + 8;
  const int rows_max = (int)cdiv(tp, g.Wo) + 1 + g.k - 1;
  const size_t wbytes = (size_t)bn * a.KP * 2;
  size_t pbytes = (size_t)rows_max * a.PW * a.CP * 2;
  if (bn >= 32 && cout % 8 == 0 && (size_t)tp * (cout + 4) * 2 > pbytes) pbytes = (size_t)tp * (cout + 4) * 2;
  pbytes = (pbytes + 15) & ~(size_t)15;
  a.off_patch = (int)((wbytes + 15) & ~(size_t)15);
  a.off_tab = a.off_patch + (int)pbytes;
  const size_t lds = (size_t)a.off_tab + (size_t)a.ksteps * 4 * sizeof(int);
  if (lds > 150 * 1024 || a.ksteps > 64) return false;
  int per_cu = (int)((160 * 1024) / lds);
  per_cu = per_cu < 1 ? 1 : per_cu > 8 ? 8 : per_cu;
  const long nwg = a.ntiles < 256L * per_cu ? a.ntiles : 256L * per_cu;
#define L2Q_CP(BNV, MIV)                                                                            \
  do {                                                                                           \
    static PerDeviceOnce attr_once;                                                              \
    if (attr_once.first()) {                                                                             \
      (void)hipFuncSetAttribute((const void*)conv_patch_h_kernel<HT, BNV, MIV>,                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);         \
    }                                                                                            \
    hipLaunchKernelGGL((conv_patch_h_kernel<HT, BNV, MIV>), dim3((unsigned)nwg), dim3(kBlock), lds, st, a); \
  } while (0)
  if (bn == 16) L2Q_CP(16, 8);
  else if (bn == 32) L2Q_CP(32, 8);
  else if (bn == 64) L2Q_CP(64, 2);
  else L2Q_CP(128, 2);
#undef L2Q_CP
  return true;
}

template bool conv_patch_launch<_Float16>(const void*, const ConvGeomH&, const void*, const float*, int, int,
                                          void*, hipStream_t);
template bool conv_patch_launch<__bf16>(const void*, const ConvGeomH&, const void*, const float*, int, int,
                                        void*, hipStream_t);

}  // namespace l2q
