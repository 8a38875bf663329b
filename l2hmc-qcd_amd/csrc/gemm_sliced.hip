// gemm_sliced.hip -- the fp64 input layer of the SU(3) vnet,
//     C[M][N] = epilogue( A[M][K] . W[N][K]^T + A2[M][K2] . W2[N][K2]^T + bias + bias2 )
// (network/pytorch/network.py:430-451: xlayer(x) + vlayer(v), activation), with the fp64 products
// formed on the INT8 matrix cores by error-free slicing, like the output heads (heads_sliced.hip) -- but
// here BOTH operands stream (K = 2 x 131 072 at cfg-4 against M = N = 256), so the activations are sliced
// ON THE FLY by helper wavefronts of the GEMM itself: no slice copy of A is ever written to memory and
// the kernels that produce A do not change.
//
// Numerics.  A value v with |v| < 2^e becomes the 54-bit fixed-point integer X = rint(v 2^(54 - e)) and is
// recoded in balanced base 256, X = sum_{s=0..6} d_s 256^(6-s), d_s in [-128, 127] (|d_0| <= 64).  Weights: one
// exponent per output row (row maximum, build time); activations: ONE exponent e for the whole operand,
// given by the caller (the vnet inputs are su3_to_vec(projectSU(.)): |entry| < 2.31, e = 2) and CHECKED by the
// slicer -- a value outside (-2^e, 2^e) or a NaN raises a flag and the whole output becomes NaN.  The
// rounding X = rint(.) is read off the mantissas of two sums with magic constants (1.5 2^76: the high 30
// bits; 1.5 2^52: the exact remainder); adding 0x80 to every byte turns the balanced digits into the plain
// bytes of the sum, and XOR 0x80 turns those into int8 -- 10 instructions per value, then byte permutes
// gather the digit planes (16 consecutive k of one row = one lane's 16-byte MFMA fragment piece per plane).
// A dot product is sum_k X_k Y_k = sum_{s,t} 256^(12-s-t) P_st with P_st = sum_k d_s[k] d'_t[k] computed
// exactly by v_mfma_i32_16x16x64_i8; the 28 pairs with s + t <= 6 are kept, summed by g = s + t in int32
// over at most 16 384 k (7 x 2^14 x 2^14 < 2^31), then combined in fp64 by a Horner chain and accumulated
// in fp64 across k-ranges.  Error per product relative to 2^e_a 2^e_w: 2^-55 (input rounding, each operand)
// + 6 x 2^-54 at worst (dropped pairs s + t = 7), like the heads kernel; here e_a is the operand's declared
// bound rather than the row's own maximum.  (First version: 50-bit integers from ONE magic-number FMA --
// the top digit then holds 3 bits and the dropped pairs weigh 2^-43: 2e-13 against the long-double value
// where the fp64 layer has 1.5e-15; tests/test_kernels_gpu.py::test_gemm_sliced caught it.)
//
// Kernel.  512-thread workgroups, one per CU: four "matrix" wavefronts (a 64 x 64 output tile as 2 x 2
// wavefronts of 32 x 32 = four 16 x 16 MFMA tiles x 7 int32 group accumulators = 112 VGPRs) and four
// "helper" wavefronts (one per SIMD next to a matrix wavefront).  Per 64-wide k-slab: the weights' 28
// fragments (pre-sliced at build time in fragment order) come by LDS-DMA, issued by the matrix wavefronts
// two slabs ahead; the activations' 28 fragments are loaded as fp64 (16 consecutive k per lane, one slab
// ahead, asm loads with explicit vmcnt), sliced and written to LDS by the helpers; two activation + three
// weight stages of 28 KB, one barrier per slab.  A matrix wavefront issues 28 ds_read_b128 (asm, explicit
// lgkmcnt) and 112 MFMAs per slab, the last 12 of them after the next barrier (they cover the barrier and
// the first LDS reads).  A workgroup owns (tile, k-range group); partial sums [group][M][N] are added in a
// fixed order by the reduce kernel, which applies bias / activation / ScaledTanh scale.
//
// Measured (cfg-4: M = N = 256, K = 2 x 131 072; tools/time_gemm_sliced.py, -DL2Q_GS_EXP=64 prints the shader
// clock and the barrier waits): 0.486 ms against 0.63 ms for the fp64 MFMA layer on the same box = 71
// fp64-equivalent TFLOP/s; 3500 clocks per slab at 2.1-2.2 GHz, where the 112 MFMAs alone take 1792.  The
// matrix wavefronts wait ~1100 clocks per slab at the barrier for the helpers: with no memory instructions
// and no conversion the slab takes 2450-2500 clocks, the conversion arithmetic adds ~450 (the helper's
// ~260 VALU instructions run at ~10 clocks each beside a busy matrix pipe:
// tools/microbench/mfma_valu_overlap.hip, "convert"), the loads and LDS-DMA ~550 more, whoever issues them.
#include "heads_common.hpp"

namespace l2q {

typedef int gs_v4i __attribute__((ext_vector_type(4)));
typedef unsigned gs_v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* gs_lds_ptr_t;

#ifndef L2Q_GS_EXP
#define L2Q_GS_EXP 0
#endif
constexpr int GS_NS = 7;                       // int8 digits per value
constexpr int GS_BITS = 54;                    // fixed-point bits below the operand's exponent
constexpr int GS_FRAG = 1024;                  // one MFMA operand fragment: 64 lanes x 16 bytes
constexpr int GS_T = 64;                       // output tile (rows = columns)
constexpr int GS_OPER = 4 * GS_NS * GS_FRAG;   // one operand of a stage: 4 row tiles x 7 digits = 28 KB
constexpr int GS_RANGE = 16384;                // k per int32 accumulation
constexpr double GS_MAGIC = 6755399441055744.0;   // 1.5 * 2^52

// 16 values -> 7 digit planes of 16 bytes (plane s: byte j = digit s of x[j]); returns nonzero when a
// value is outside (-lim, lim) or not a number.  X = rint(x sc) (|X| <= 2^54) = H 2^24 + L in two exact
// steps, each read off the mantissa of a sum with a "magic" constant: t1 = fma(x, sc, 1.5 2^76) has ulp
// 2^24, so t1 - 1.5 2^76 = H 2^24 and its low mantissa dword is H; r = x sc - H 2^24 is exact (one FMA),
// |r| <= 2^23, and r + 1.5 2^52 carries L = rint(r).  Balanced digits: the bytes of (value + 0x80..80) are
// digit + 128 -- done on L (3 bytes, its carry goes into H) and on H (4 bytes); XOR 0x80 makes them int8.
// ~10 instructions per value.  hook(j) runs after value j is converted (the GEMM's helpers issue their
// memory instructions there, spread through the arithmetic; called for j = 3, 7, 11, 15); get(j) supplies
// value j when its turn comes.
struct GsNoHook { __device__ __forceinline__ void operator()(int) const {} };
template <bool PIN, class Get, class Hook>
__device__ __forceinline__ int gs_slice16_impl(Get get, double sc, double lim, gs_v4u (&out)[GS_NS], Hook hook) {
  unsigned lo[16], hi[16];               // lo: digits 6, 5, 4 in bytes 0, 1, 2; hi: digits 3, 2, 1, 0 in bytes 0..3
  unsigned long long bad = 0;            // wavefront mask (scalar registers): |x| >= lim or NaN
  constexpr double M24 = 113336795588871485128704.0;          // 1.5 * 2^76
  // four values at a time, stage by stage: four independent dependency chains side by side (written value by
  // value the compiler emits each chain serially and every instruction waits for the one before it)
#pragma unroll
  for (int j0 = 0; j0 < 16; j0 += 4) {
    double xj[4], t1[4], r[4], t2[4];
    int Lb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xj[i] = get(j0 + i);
#if defined(L2Q_GS_EXP) && (L2Q_GS_EXP & 1)
    // timing experiment: no conversion arithmetic (the raw bits go through the transposes)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      lo[j0 + i] = (unsigned)__double_as_longlong(xj[i]) ^ 0x00808080u;       // (an instruction reads the loaded register)
      hi[j0 + i] = (unsigned)(__double_as_longlong(xj[i]) >> 32) ^ 0x80808080u;
    }
    if (PIN)
      asm volatile("" : "+v"(lo[j0]), "+v"(hi[j0]), "+v"(lo[j0 + 1]), "+v"(hi[j0 + 1]), "+v"(lo[j0 + 2]), "+v"(hi[j0 + 2]),
                   "+v"(lo[j0 + 3]), "+v"(hi[j0 + 3]));
    hook(j0 + 3);
    continue;
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) t1[i] = fma(xj[i], sc, M24);
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = M24 - t1[i];             // -H 2^24 exactly
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = fma(xj[i], sc, r[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) bad |= __builtin_amdgcn_fcmp(fabs(xj[i]), lim, 11);   // 11 = unordered or >=
#pragma unroll
    for (int i = 0; i < 4; ++i) t2[i] = r[i] + GS_MAGIC;
#pragma unroll
    for (int i = 0; i < 4; ++i) Lb[i] = (int)(unsigned)__double_as_longlong(t2[i]) + 0x00808080;   // in [0x8080, 0x1008080]: bit 24 = carry into H
#pragma unroll
    for (int i = 0; i < 4; ++i) lo[j0 + i] = (unsigned)Lb[i] ^ 0x00808080u;                         // (bytes 0..2 are used)
#pragma unroll
    for (int i = 0; i < 4; ++i) Lb[i] >>= 24;
#pragma unroll
    for (int i = 0; i < 4; ++i) hi[j0 + i] = (unsigned)__double_as_longlong(t1[i]) + (unsigned)Lb[i] + 0x80808080u;
#pragma unroll
    for (int i = 0; i < 4; ++i) hi[j0 + i] ^= 0x80808080u;
    if (PIN)                                                    // the conversions stay in front of the hook
      asm volatile("" : "+v"(lo[j0]), "+v"(hi[j0]), "+v"(lo[j0 + 1]), "+v"(hi[j0 + 1]), "+v"(lo[j0 + 2]), "+v"(hi[j0 + 2]),
                   "+v"(lo[j0 + 3]), "+v"(hi[j0 + 3]));
    hook(j0 + 3);
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const unsigned a0 = lo[4 * g], a1 = lo[4 * g + 1], a2 = lo[4 * g + 2], a3 = lo[4 * g + 3];
    const unsigned h0 = hi[4 * g], h1 = hi[4 * g + 1], h2 = hi[4 * g + 2], h3 = hi[4 * g + 3];
    // v_perm_b32(hi_src, lo_src, sel): selector byte 0..3 -> lo_src byte, 4..7 -> hi_src byte
    // pairs (bytes 0, 1) and (bytes 2, 3) of two values at a time
    const unsigned p01a = __builtin_amdgcn_perm(a1, a0, 0x05010400u);   // [a0.b0, a1.b0, a0.b1, a1.b1]
    const unsigned p23a = __builtin_amdgcn_perm(a3, a2, 0x05010400u);
    const unsigned p01b = __builtin_amdgcn_perm(a1, a0, 0x07030602u);   // [a0.b2, a1.b2, a0.b3, a1.b3]
    const unsigned p23b = __builtin_amdgcn_perm(a3, a2, 0x07030602u);
    const unsigned q01a = __builtin_amdgcn_perm(h1, h0, 0x05010400u);
    const unsigned q23a = __builtin_amdgcn_perm(h3, h2, 0x05010400u);
    const unsigned q01b = __builtin_amdgcn_perm(h1, h0, 0x07030602u);
    const unsigned q23b = __builtin_amdgcn_perm(h3, h2, 0x07030602u);
    out[6][g] = __builtin_amdgcn_perm(p23a, p01a, 0x05040100u);         // L byte 0
    out[5][g] = __builtin_amdgcn_perm(p23a, p01a, 0x07060302u);         // L byte 1
    out[4][g] = __builtin_amdgcn_perm(p23b, p01b, 0x05040100u);         // L byte 2
    out[3][g] = __builtin_amdgcn_perm(q23a, q01a, 0x05040100u);         // H byte 0
    out[2][g] = __builtin_amdgcn_perm(q23a, q01a, 0x07060302u);         // H byte 1
    out[1][g] = __builtin_amdgcn_perm(q23b, q01b, 0x05040100u);         // H byte 2
    out[0][g] = __builtin_amdgcn_perm(q23b, q01b, 0x07060302u);         // H byte 3
  }
  return bad != 0;
}
__device__ __forceinline__ int gs_slice16(const double (&x)[16], double sc, double lim, gs_v4u (&out)[GS_NS]) {
  return gs_slice16_impl<false>([&](int j) { return x[j]; }, sc, lim, out, GsNoHook());
}

// ---- build: per-row exponent, then the digit image in fragment order ----------------------------------
// exps[n] = e with max_k |W[n][k]| < 2^e (0 for an all-zero row); flag |= 1 for a non-finite entry
__global__ __launch_bounds__(256) void gs_rowexp_kernel(const double* __restrict__ W, long K, int* __restrict__ exps,
                                                         int* __restrict__ flag) {
  __shared__ double red[4];
  const double* row = W + (long)blockIdx.x * K;
  double mx = 0.0;
  int bad = 0;
  for (long k = threadIdx.x * 2L; k < K; k += 512) {
    const double2 v = *reinterpret_cast<const double2*>(row + k);
    const double a = fabs(v.x), b = fabs(v.y);
    bad |= !(a <= 1.7976931348623157e308) | !(b <= 1.7976931348623157e308);
    mx = fmax(mx, fmax(a, b));
  }
  const double m = block_max(mx, red);
  if (__syncthreads_or(bad) && threadIdx.x == 0) atomicOr(flag, 1);
  if (threadIdx.x == 0) {
    int e = 0;
    if (m > 0.0 && m <= 1.7976931348623157e308) (void)frexp(m, &e);      // m = f 2^e, f in [0.5, 1)
    exps[blockIdx.x] = e;
  }
}

// K-order inside a 64-wide slab.  An MFMA operand fragment gives lane (row r = lane & 15, group g = lane >> 4)
// 16 k-positions; WHICH 16 is free as long as both operands agree.  Byte b of group g holds
//     k = 8 (b >> 1) + 2 g + (b & 1):
// the 16-byte load j of a thread then fetches k = 8 j + 2 g, + 1, so the four threads of a row read 64
// contiguous bytes per instruction (16 consecutive k per lane would make every lane of a load instruction
// touch its own cache line: measured, the activation loads were 37 % of the kernel).  The four threads of a
// row are ADJACENT lanes of the loading wavefront (row = lane >> 2, g = lane & 3); their digits go to the
// fragment position r + 16 g.
// image[(kslab * NT + ntile) * 7 + digit][fragment lane][16 B]
__global__ __launch_bounds__(256) void gs_build_kernel(const double* __restrict__ W, long K, int NT,
                                                        const int* __restrict__ exps, char* __restrict__ image,
                                                        double* __restrict__ wsc) {
  const int lane = threadIdx.x & 63;
  const long frag = blockIdx.x * 4L + (threadIdx.x >> 6);         // (kslab, ntile)
  const long nslab = K / 64;
  if (frag >= nslab * NT) return;
  const long ks = frag / NT;
  const int nt = (int)(frag % NT);
  const int r = lane >> 2, g = lane & 3;
  const int n = nt * 16 + r;
  const int e = exps[n];
  const double sc = ldexp(1.0, GS_BITS - e);
  const double2* src = reinterpret_cast<const double2*>(W + (long)n * K + ks * 64);
  double x[16];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const double2 v = src[4 * j + g];
    x[2 * j] = v.x; x[2 * j + 1] = v.y;
  }
  gs_v4u out[GS_NS];
  (void)gs_slice16(x, sc, 1.7976931348623157e308, out);      // (|w| <= 2^e by construction; non-finite rows are flagged above)
  char* dst = image + frag * (long)(GS_NS * GS_FRAG) + (r + 16 * g) * 16;
#pragma unroll
  for (int s = 0; s < GS_NS; ++s) *reinterpret_cast<gs_v4u*>(dst + s * GS_FRAG) = out[s];
  if (ks == 0 && g == 0) wsc[n] = ldexp(1.0, e - 2 * GS_BITS + 48);   // 256^6 2^-(54 - e_w) 2^-54; x 2^e_a at run time
}

// timing experiments (tools/ab_build.sh ... -DL2Q_GS_EXP=bits; wrong results): 1 the helpers store the raw
// loaded bits instead of digits (no slicing arithmetic), 2 one MFMA per digit pair instead of four,
// 4 no weight LDS-DMA after the prologue, 8 no activation loads after the prologue, 64 print the shader
// clock measured over the kernel

struct GsArgs {
  const double* A[2];       // activations [M][K]
  const char* img[2];       // digit images of the weights
  const double* wsc[2];     // [N] column scales
  long K[2];
  double sc[2], lim[2], post[2];   // 2^(50 - e_a), 2^e_a, 2^e_a
  int groups0;              // k-range groups of operand 0 (the rest belong to operand 1)
  long klen;                // k per group (a multiple of 64)
  int M, N;
  double* part;             // [groups][M][N]
  int* flag;
};

// compile-time loop and LDS read / wait with literal operands (asm wants immediates)
template <int I> using gs_c = std::integral_constant<int, I>;
template <int I, int N, class F>
__device__ __forceinline__ void gs_for(F f) {
  if constexpr (I < N) {
    f(gs_c<I>());
    gs_for<I + 1, N>(f);
  }
}
template <int OFF>
__device__ __forceinline__ void gs_dsr(gs_v4i& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void gs_wait(gs_v4i& a0, gs_v4i& a1, gs_v4i& b0, gs_v4i& b1) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1) : "n"(N));
}

// (matrix wavefronts: at B(p) the pieces of slab p, issued in period p - 2, must have landed; the seven of
// period p - 1 may still be in flight)
__device__ __forceinline__ void gs_barrier() { asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(512, 1) void gemm_sliced_kernel(GsArgs a, int swz) {
  // two stages of activation digits (written by the helpers one slab ahead), three of weight digits (LDS-DMA
  // by the matrix wavefronts, two slabs ahead: a piece has more than a period to land)
  __shared__ __attribute__((aligned(1024))) char lds[5 * GS_OPER];           // 140 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tn_count = a.N / GS_T;
  const int tiles = (a.M / GS_T) * tn_count;
  const long w = xcd_swizzle(blockIdx.x, gridDim.x, swz);
  const int grp = (int)(w / tiles), tile = (int)(w % tiles);
  const int tm = tile / tn_count, tn = tile % tn_count;
  const int op = grp < a.groups0 ? 0 : 1;
  const int gl = op ? grp - a.groups0 : grp;
  const long K = a.K[op];
  const long kbeg = (long)gl * a.klen;
  const long kend = kbeg + a.klen < K ? kbeg + a.klen : K;
  const int nslab = (int)((kend - kbeg) / 64);
  const int NT = a.N / 16;

  const char* bsrc = a.img[op] + ((kbeg / 64) * NT + tn * 4) * (long)(GS_NS * GS_FRAG) + lane * 16;
  const long bstep = (long)NT * (GS_NS * GS_FRAG);               // next k-slab of the image
  // experiment 64: shader clock over the kernel (s_memtime counts core clocks, s_memrealtime 100 MHz)
  unsigned long long t0c = 0, t0r = 0, twait = 0, twait2 = 0;
  if (L2Q_GS_EXP & 64) { t0c = __builtin_amdgcn_s_memtime(); t0r = __builtin_amdgcn_s_memrealtime(); }
  if (wave < 4) {
    // ================================================================ matrix wavefronts
    const int wm = wave >> 1, wn = wave & 1;
#ifndef L2Q_GS_PRIO
#define L2Q_GS_PRIO 3
#endif
    // the MFMA stream is the critical path: the arbiter should serve it first and fit the helper's conversion
    // arithmetic (same SIMD) into the gaps
    __builtin_amdgcn_s_setprio(L2Q_GS_PRIO);
    gs_v4i acc[4][GS_NS];
    double racc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int g = 0; g < GS_NS; ++g) acc[t][g] = (gs_v4i){0, 0, 0, 0};
#pragma unroll
      for (int r = 0; r < 4; ++r) racc[t][r] = 0.0;
    }
    // The activation fragments are XOR-swizzled in LDS: piece (row r, k-chunk g) of a fragment sits in slot
    // 16 g + (r ^ 2 g) instead of 16 g + r.  A helper's ds_write_b128 (8 consecutive lanes = 2 rows x 4 chunks)
    // then covers all 32 banks once instead of hitting four of them 4 times (PMC before: half of the LDS-active
    // cycles of the kernel were bank conflicts), and the matrix wavefronts' ds_read_b128 lane groups stay
    // conflict-free (both checked by enumeration over the bank model of MI355X_MICROARCH.md).
    const char* abase = lds + (2 * wm) * (GS_NS * GS_FRAG) + ((lane & 48) | ((lane & 15) ^ ((lane >> 4) * 2))) * 16;
    const char* bbase = lds + 2 * GS_OPER + (2 * wn) * (GS_NS * GS_FRAG) + lane * 16;
    int p3 = 0;                                                // p % 3
    // The weight digits come in by LDS-DMA, issued by the MATRIX wavefronts (the helpers are the longer path:
    // a memory instruction costs the issuing wavefront 60-180 clocks, and these wavefronts otherwise wait
    // ~1400 clocks per slab at the barrier): wavefront w moves fragments w, w + 4, ..., w + 24 of a slab, one
    // after each MFMA group of row 0, two slabs ahead (three stages: a piece has more than a period to land).
    auto dma_b = [&](int q, int f) {
      if (L2Q_GS_EXP & 4) return;
      const int qc = q < nslab ? q : nslab - 1;                 // past the end: the last slab again, into a free stage
      const int frag = wave + 4 * f;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(bsrc + (long)qc * bstep + frag * GS_FRAG),
          (gs_lds_ptr_t)(lds + (2 + q % 3) * GS_OPER + frag * GS_FRAG), 16, 0, 0);
    };
#pragma unroll
    for (int f = 0; f < GS_NS; ++f) dma_b(0, f);
#pragma unroll
    for (int f = 0; f < GS_NS; ++f) dma_b(1, f);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    gs_v4i bf[2][GS_NS], af[2][2], afd[2][2];
    bool pending = false;                                      // rows 5, 6 of the previous slab not issued yet
    for (int p = 0; p < nslab; ++p) {
      unsigned long long tb = 0;
      if (L2Q_GS_EXP & 64) tb = __builtin_amdgcn_s_memtime();
      gs_barrier();                                            // B(p): the stages of slab p are complete
      if (L2Q_GS_EXP & 64) twait += __builtin_amdgcn_s_memtime() - tb;
      const int soa = (p & 1) * GS_OPER, sob = p3 * GS_OPER;
      p3 = p3 == 2 ? 0 : p3 + 1;
      // Reads and MFMAs in an explicit order, the reads as asm with explicit lgkmcnt waits (LDS returns in
      // order): left to the compiler, all 18 reads of the first row are hoisted and waited for with
      // lgkmcnt(0) -- ~600 clocks of LDS time with four wavefronts reading -- before the first MFMA.  Row 0
      // starts on four reads and pulls the other weight fragments in two groups ahead; the activation
      // fragments of row s + 1 are read while row s runs.  A fragment is only touched through its wait.
      const unsigned aaddr = (unsigned)(unsigned long)(gs_lds_ptr_t)(abase + soa);
      const unsigned baddr = (unsigned)(unsigned long)(gs_lds_ptr_t)(bbase + sob);
      auto rd_a = [&](auto sc, gs_v4i (&dst)[2]) {
        constexpr int S = decltype(sc)::value;
        gs_dsr<S * GS_FRAG>(dst[0], aaddr);
        gs_dsr<(GS_NS + S) * GS_FRAG>(dst[1], aaddr);
      };
      auto rd_b = [&](auto tc) {
        constexpr int T = decltype(tc)::value;
        gs_dsr<T * GS_FRAG>(bf[0][T], baddr);
        gs_dsr<(GS_NS + T) * GS_FRAG>(bf[1][T], baddr);
      };
      auto mm = [&](const gs_v4i (&a2)[2], int s, int t) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (!(L2Q_GS_EXP & 2) || (i == 0 && j == 0))
              acc[2 * i + j][s + t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2[i], bf[j][t], acc[2 * i + j][s + t], 0, 0, 0);
      };
      // The last two rows of a slab (12 MFMAs, fragments kept in afd and in the weight fragments 0 and 1) are
      // issued after the NEXT barrier, behind the first reads of the next slab: they cover the barrier and the
      // LDS latency, during which the matrix pipe would idle.  Row 0 runs t = 6 ... 0 so that its first weight
      // fragments go into registers the old slab no longer needs.
      auto tail = [&]() {
        mm(afd[0], 5, 0);
        mm(afd[0], 5, 1);
        mm(afd[1], 6, 0);
      };
      rd_a(gs_c<0>(), af[0]);
      rd_b(gs_c<6>());
      rd_b(gs_c<5>());
      rd_b(gs_c<4>());
      if (pending) {
        gs_wait<8>(afd[0][0], afd[0][1], afd[1][0], afd[1][1]);   // (complete since the barrier; orders the MFMAs)
        tail();
      }
      rd_b(gs_c<3>());
      gs_for<0, GS_NS>([&](auto uc) {                       // row 0: 7 groups of 4 MFMAs
        constexpr int T = GS_NS - 1 - decltype(uc)::value;
        // outstanding behind the fragments of group T: three more pairs (two for T = 1, one for T = 0)
        gs_wait<(T >= 2 ? 6 : T == 1 ? 4 : 2)>(af[0][0], af[0][1], bf[0][T], bf[1][T]);
        mm(af[0], 0, T);
        dma_b(p + 2, GS_NS - 1 - T);
        if constexpr (T >= 4) rd_b(gs_c<T - 4>());
        if constexpr (T == 3) rd_a(gs_c<1>(), af[1]);
      });
      gs_for<1, 5>([&](auto sc) {                           // rows 1..4: 24, 20, 16, 12 MFMAs
        constexpr int S = decltype(sc)::value;
        gs_wait<0>(af[S & 1][0], af[S & 1][1], bf[0][0], bf[1][0]);
        if constexpr (S < 4) rd_a(gs_c<S + 1>(), af[(S + 1) & 1]);
        if constexpr (S == 4) {
          rd_a(gs_c<5>(), afd[0]);
          rd_a(gs_c<6>(), afd[1]);
        }
#pragma unroll
        for (int t = 0; S + t < GS_NS; ++t) mm(af[S & 1], S, t);
      });
      pending = true;
      if (((p + 1) & (GS_RANGE / 64 - 1)) == 0 || p + 1 == nslab) {
        // end of an int32 range: the last rows now, then sum_g 256^(6-g) S_g in fp64, accumulate, clear
        gs_wait<0>(afd[0][0], afd[0][1], afd[1][0], afd[1][1]);
        tail();
        pending = false;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            double x = (double)acc[t][0][r];
#pragma unroll
            for (int g = 1; g < GS_NS; ++g) x = fma(x, 256.0, (double)acc[t][g][r]);
            racc[t][r] += x;
          }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int g = 0; g < GS_NS; ++g) acc[t][g] = (gs_v4i){0, 0, 0, 0};
      }
    }
    if ((L2Q_GS_EXP & 64) && wave == 0 && lane == 0 && (blockIdx.x & 63) == 0) {
      const unsigned long long c = __builtin_amdgcn_s_memtime() - t0c, r = __builtin_amdgcn_s_memrealtime() - t0r;
      printf("block %d: %llu core clocks, %llu x 10 ns -> %.0f MHz, %.0f clocks per slab; matrix wavefront 0 at the barrier %.0f per slab\n",
             (int)blockIdx.x, c, r, (double)c / ((double)r * 0.01), (double)c / nslab, (double)twait / nslab);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (no DMA may outlive the workgroup's LDS)
    // C/D layout of v_mfma_i32_16x16x64_i8: col = lane & 15, row = 4 (lane >> 4) + reg
    double* part = a.part + (long)grp * a.M * a.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = tn * GS_T + (2 * wn + j) * 16 + (lane & 15);
        const double cs = a.wsc[op][n] * a.post[op];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = tm * GS_T + (2 * wm + i) * 16 + 4 * (lane >> 4) + r;
          part[(long)m * a.N + n] = racc[2 * i + j][r] * cs;
        }
      }
    return;
  }

  // ================================================================== helper wavefronts
  const int h = wave - 4;                                       // row tile of the A operand
  const double sc = a.sc[op], lim = a.lim[op];
  const int hr = lane >> 2, hg = lane & 3;                      // row of the tile / k-group this thread slices
  const double* arow = a.A[op] + (long)(tm * GS_T + h * 16 + hr) * K + kbeg;
  int bad = 0;
  // The activation loads are written as asm: the compiler's own wait placement for register loads that cross
  // the loop edge is a vmcnt(0) at the top of every period (seen in the ISA), which would expose the latency
  // of the loads issued a moment earlier.  Here the waits are explicit (gs_wait below); ld is only read
  // through them.
  typedef double gs_v2d __attribute__((ext_vector_type(2)));
  auto load1 = [&](gs_v2d& dst, const double* src) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src) : "memory");
  };
  auto load_a = [&](int q, gs_v2d (&ld)[8]) {
    const int qc = q < nslab ? q : nslab - 1;                   // past the end: the last slab again (uniform counts)
    const double* src = arow + (long)qc * 64 + 2 * hg;
#pragma unroll
    for (int j = 0; j < 8; ++j) load1(ld[j], src + 8 * j);
  };
  // digits of slab q from the registers ld into A stage q & 1.  With `next`, the loads of slab q + 1 are issued
  // between the conversions, each into the registers of the pair just consumed, so that the texture path
  // (1 KB per instruction, the issue costs this wavefront 60-180 clocks) works while the wavefront converts.
  // Four values (two loads) at a time: four independent chains for the scheduler.  vmcnt is in order and a
  // period issues load_0 ... load_7: when group g (loads 2g, 2g + 1) is about to be converted, 6 - 2g older
  // loads follow them and 2g were issued in this period, so vmcnt(6) proves them complete, for every g.
  auto slice_store = [&](int q, gs_v2d (&ld)[8], bool next) {
    gs_v4u out[GS_NS];
    const int qn = q + 1 < nslab ? q + 1 : nslab - 1;            // past the end: the last slab again
    const double* asrc = arow + (long)qn * 64 + 2 * hg;
    auto get = [&](int j) {
      if (!(j & 3)) {
        if (next) asm volatile("s_waitcnt vmcnt(6)" : "+v"(ld[j >> 1]), "+v"(ld[(j >> 1) + 1]));
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(ld[j >> 1]), "+v"(ld[(j >> 1) + 1]));
      }
      return (double)ld[j >> 1][j & 1];
    };
    auto hook = [&](int j) {
      if (!next || (j & 3) != 3) return;
      const int f = j >> 1;                                    // (odd: the second load of the group of four values)
      __builtin_amdgcn_sched_barrier(0);
      load1(ld[f - 1], asrc + 8 * (f - 1));
      load1(ld[f], asrc + 8 * f);
      __builtin_amdgcn_sched_barrier(0);
    };
    bad |= gs_slice16_impl<true>(get, sc, lim, out, hook);
    char* dst = lds + (q & 1) * GS_OPER + h * (GS_NS * GS_FRAG) + ((hr ^ (2 * hg)) + 16 * hg) * 16;   // (swizzled, see abase)
#pragma unroll
    for (int s = 0; s < GS_NS; ++s) *reinterpret_cast<gs_v4u*>(dst + s * GS_FRAG) = out[s];
  };
  if (nslab > 0) {
    // Period p (between B(p) and B(p + 1)) converts slab p + 1 and issues the loads of slab p + 2, which have
    // about one period to arrive.
    gs_v2d ld[8];
    load_a(0, ld);
    slice_store(0, ld, false);                                   // (waits for everything issued so far)
    load_a(1, ld);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                 // B(0)
    for (int p = 0; p + 1 < nslab; ++p) {
      slice_store(p + 1, ld, true);
      unsigned long long tb = 0;
      if (L2Q_GS_EXP & 64) {
        tb = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        twait2 += __builtin_amdgcn_s_memtime() - tb;
        tb = __builtin_amdgcn_s_memtime();
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");               // B(p + 1)
      if (L2Q_GS_EXP & 64) twait += __builtin_amdgcn_s_memtime() - tb;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if ((L2Q_GS_EXP & 64) && wave == 4 && lane == 0 && (blockIdx.x & 63) == 0)
    printf("block %d helper 0: at the barrier %.0f per slab, waiting for its LDS stores before it %.0f\n", (int)blockIdx.x,
           (double)twait / nslab, (double)twait2 / nslab);
  if (__any(bad) && lane == 0) atomicOr(a.flag, 1);
}

// out[m][n] = scale exp(coeff[n]) act( sum_g part[g][m][n] + bias[n] + bias2[n] ), groups added in order;
// a raised flag (operand outside its declared range / NaN) poisons the output
__device__ __forceinline__ double gs_act(double z, int act) {
  switch (act) {
    case L2Q_ACT_TANH: return fast_tanh(z);       // (the fp64 layer's tanh, heads_common.hpp)
    case L2Q_ACT_RELU: return z > 0.0 ? z : 0.0;
    case L2Q_ACT_LEAKY_RELU: return z > 0.0 ? z : 0.01 * z;
    case L2Q_ACT_ELU: return z > 0.0 ? z : expm1(z);
    case L2Q_ACT_SWISH: return z / (1.0 + exp(-z));
    default: return z;
  }
}

// The "operand out of range" flag of a call and the ticket that clears it: device symbols (zero at module load), NOT a
// workspace word reset by the launcher with hipMemsetAsync.  Captured into a HIP graph that memset is a memset NODE,
// and on ROCm 7.0 such a node makes the replayed graph unsafe: after a >= 512 KB device-to-host copy on the null
// stream every later replay of a captured SU(3) trajectory came out NaN, even with this self-cleaning flag in place
// (the node itself scribbles), while the same graph ran 10 % FASTER than eager launches -- an effect that exists only
// with the node present (19.2 vs 21.2-21.8 ms per trajectory with a zeroing kernel, with no reset at all, or with the
// node anywhere else: profiles/r05k_graph_memset_node.txt) and is therefore not something to build on.  The flag cleans
// up after itself instead: the LAST block of gs_reduce_kernel to have read it puts flag and ticket back to zero for
// the next call.  One flag per device: calls on different streams of one device would share it (the library
// launches on one stream).
__device__ int gs_flag_dev[2];
static int* gs_flag_ptr() {
  static int* cache[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!cache[dev]) {
    int* p = nullptr;
    if (hipGetSymbolAddress((void**)&p, HIP_SYMBOL(gs_flag_dev)) != hipSuccess) return nullptr;
    cache[dev] = p;
  }
  return cache[dev];
}

__global__ __launch_bounds__(256) void gs_reduce_kernel(const double* __restrict__ part, int groups, long MN, int N,
                                                         const double* __restrict__ bias,
                                                         const double* __restrict__ bias2,
                                                         const double* __restrict__ coeff, double scale, int act,
                                                         int* flag, double* __restrict__ C) {
  const long i = blockIdx.x * 256L + threadIdx.x;
  const int raised = __atomic_load_n(flag, __ATOMIC_RELAXED);
  if (i < MN) {
    double s = 0.0;
    for (int g = 0; g < groups; ++g) s += part[(long)g * MN + i];
    const int n = (int)(i % N);
    double b = 0.0;
    if (bias) b += bias[n];
    if (bias2) b += bias2[n];
    double y = (coeff ? scale * exp(coeff[n]) : scale) * gs_act(s + b, act);
    if (raised) y = __longlong_as_double(0x7ff8000000000000LL);
    C[i] = y;
  }
  // every thread of the block has read the flag: take a ticket; the last block clears flag and ticket
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(flag + 1, 1) == (int)gridDim.x - 1) {
      __atomic_store_n(flag, 0, __ATOMIC_RELAXED);
      __atomic_store_n(flag + 1, 0, __ATOMIC_RELAXED);
    }
  }
}

static inline size_t gs_image_bytes(int N, long K) {
  return (size_t)(K / 64) * (size_t)(N / 16) * GS_NS * GS_FRAG;
}
static inline size_t gs_align(size_t x) { return (x + 255) & ~(size_t)255; }

// k per group: about 256 workgroups in all (one round of one per CU), at least 4096 k (64 slabs: the
// prologue and the partial-sum traffic stay small), at most 64 int32 ranges
static long gs_pick_klen(int M, int N, long K, long K2) {
  const long tiles = (long)(M / GS_T) * (N / GS_T);
#ifndef L2Q_GS_TARGET
#define L2Q_GS_TARGET 256      // (A/B at cfg-4, same box: 256 -> 0.450 ms, 512 -> 0.454-0.464, 1024 -> 0.471-0.476)
#endif
  long klen = cdiv(cdiv((K + K2) * tiles, L2Q_GS_TARGET), 1024) * 1024;
  if (klen < 4096) klen = 4096;
  if (klen > 64L * GS_RANGE) klen = 64L * GS_RANGE;
  return klen;
}

}  // namespace l2q

using namespace l2q;

extern "C" {

size_t l2q_gemm_sliced_bytes(int N, long K) {
  if (N <= 0 || K <= 0 || N % GS_T != 0 || K % 64 != 0) return 0;
  return gs_align(gs_image_bytes(N, K)) + gs_align((size_t)N * sizeof(double)) + gs_align((size_t)N * sizeof(int)) + 256;
}

int l2q_gemm_sliced_build(const double* W, int N, long K, void* image, size_t image_bytes, int* usable,
                          void* stream) {
  L2Q_REQUIRE(W && image && usable, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(N > 0 && K > 0 && N % GS_T == 0 && K % 64 == 0, L2Q_ESHAPE,
              "the sliced layer serves N % 64 == 0, K % 64 == 0");
  L2Q_REQUIRE(image_bytes >= l2q_gemm_sliced_bytes(N, K), L2Q_ESHAPE, "buffer too small");
  L2Q_REQUIRE((reinterpret_cast<uintptr_t>(W) & 15) == 0 && (reinterpret_cast<uintptr_t>(image) & 255) == 0,
              L2Q_ESHAPE, "W must be 16-byte aligned, the image 256-byte");
  hipStream_t st = (hipStream_t)stream;
  char* buf = (char*)image;
  double* wsc = (double*)(buf + gs_align(gs_image_bytes(N, K)));
  int* exps = (int*)((char*)wsc + gs_align((size_t)N * sizeof(double)));
  int* flag = (int*)((char*)exps + gs_align((size_t)N * sizeof(int)));
  (void)hipMemsetAsync(flag, 0, sizeof(int), st);
  hipLaunchKernelGGL(gs_rowexp_kernel, dim3((unsigned)N), dim3(256), 0, st, W, K, exps, flag);
  const long nfrag = (K / 64) * (long)(N / 16);
  hipLaunchKernelGGL(gs_build_kernel, dim3((unsigned)cdiv(nfrag, 4)), dim3(256), 0, st, W, K, N / 16, exps, buf, wsc);
  int hflag = 0;
  if (hipMemcpyAsync(&hflag, flag, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) {
    set_error("l2q_gemm_sliced_build: %s", hipGetErrorString(hipGetLastError()));
    return L2Q_EHIP;
  }
  *usable = hflag ? 0 : 1;
  return check_launch("l2q_gemm_sliced_build");
}

size_t l2q_gemm_sliced_ws_bytes(int M, int N, long K, long K2) {
  if (M <= 0 || N <= 0 || K <= 0 || K2 < 0) return 0;
  const long klen = gs_pick_klen(M, N, K, K2);
  const long groups = cdiv(K, klen) + (K2 > 0 ? cdiv(K2, klen) : 0);
  return (size_t)groups * M * N * sizeof(double) + 512;
}

int l2q_gemm_sliced_f64(const double* A, const void* image, long K, int a_exp, const double* A2,
                        const void* image2, long K2, int a2_exp, int M, int N, const double* bias,
                        const double* bias2, const double* coeff, double scale, int act, double* C, void* ws,
                        size_t ws_bytes, void* stream) {
  L2Q_REQUIRE(A && image && C && ws, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(K2 == 0 || (A2 && image2), L2Q_EINVAL, "second operand pair missing");
  L2Q_REQUIRE(M > 0 && N > 0 && K > 0 && K2 >= 0, L2Q_EINVAL, "non-positive size");
  L2Q_REQUIRE(M % GS_T == 0 && N % GS_T == 0 && K % 64 == 0 && K2 % 64 == 0, L2Q_ESHAPE,
              "the sliced layer serves M, N % 64 == 0 and K, K2 % 64 == 0");
  L2Q_REQUIRE(a_exp > -900 && a_exp < 900 && a2_exp > -900 && a2_exp < 900, L2Q_EINVAL, "bad operand exponent");
  L2Q_REQUIRE(act >= L2Q_ACT_NONE && act <= L2Q_ACT_SWISH, L2Q_EINVAL, "bad activation");
  L2Q_REQUIRE(ws_bytes >= l2q_gemm_sliced_ws_bytes(M, N, K, K2), L2Q_ESHAPE, "workspace too small");
  auto al = [](const void* p, uintptr_t m) { return (reinterpret_cast<uintptr_t>(p) & m) == 0; };
  L2Q_REQUIRE(al(A, 15) && (!A2 || al(A2, 15)) && al(image, 255) && (!image2 || al(image2, 255)) && al(ws, 255),
              L2Q_ESHAPE, "operands must be 16-byte aligned (images and workspace 256-byte)");
  hipStream_t st = (hipStream_t)stream;
  GsArgs a;
  const long klen = gs_pick_klen(M, N, K, K2);
  const int g0 = (int)cdiv(K, klen), g1 = K2 > 0 ? (int)cdiv(K2, klen) : 0;
  a.A[0] = A; a.A[1] = A2;
  a.img[0] = (const char*)image; a.img[1] = (const char*)image2;
  a.wsc[0] = (const double*)((const char*)image + gs_align(gs_image_bytes(N, K)));
  a.wsc[1] = K2 > 0 ? (const double*)((const char*)image2 + gs_align(gs_image_bytes(N, K2))) : nullptr;
  a.K[0] = K; a.K[1] = K2;
  const int ex[2] = {a_exp, a2_exp};
  for (int o = 0; o < 2; ++o) {
    a.sc[o] = ldexp(1.0, GS_BITS - ex[o]);
    a.lim[o] = ldexp(1.0, ex[o]);
    a.post[o] = ldexp(1.0, ex[o]);
  }
  a.groups0 = g0; a.klen = klen; a.M = M; a.N = N;
  const int groups = g0 + g1;
  a.part = (double*)ws;
  a.flag = gs_flag_ptr();
  L2Q_REQUIRE(a.flag, L2Q_EHIP, "device symbol gs_flag_dev not found");
  const int tiles = (M / GS_T) * (N / GS_T);
  hipLaunchKernelGGL(gemm_sliced_kernel, dim3((unsigned)(groups * tiles)), dim3(512), 0, st, a, tuning().xcd_swizzle);
  const long MN = (long)M * N;
  hipLaunchKernelGGL(gs_reduce_kernel, dim3((unsigned)cdiv(MN, 256)), dim3(256), 0, st, (const double*)ws, groups, MN,
                     N, bias, bias2, coeff, scale, act, a.flag, C);
  return check_launch("l2q_gemm_sliced_f64");
}

}  // extern "C"
