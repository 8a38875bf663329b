// heads_sliced.hip -- the three output heads of the SU(3) / U(1) fp64 vnet + the generalised momentum
// update (network.py:547-551 + dynamics.py:1266-1297) with the fp64 products formed on the INT8
// matrix cores by error-free slicing (the "Ozaki scheme"): gfx950 issues v_mfma_i32_16x16x64_i8 at
// ~50x the rate of v_mfma_f64_16x16x4_f64 (3.9 POP/s against 78.6 TFLOP/s), so an fp64 dot product
// rebuilt from 28 exact int8 x int8 -> int32 slice products costs about half the matrix-core time of
// the native fp64 instruction -- and unlike it, leaves the VALU free for the epilogue (a wavefront's
// fp64 MFMAs block every VALU instruction of the other wavefront on the SIMD,
// tools/microbench/mfma_valu_overlap.hip).
//
// Numerics.  Each vector v (a row of Z = one chain's last hidden activations, or a row of W = one
// output entry's weights; K = 256 numbers) gets ONE exponent e with |v_k| < 2^e and is rounded to the
// 54-bit fixed-point integer X_k = rint(v_k 2^(54 - e)), then recoded in balanced base 256:
//     X_k = sum_{s=0..6} d_s[k] 256^(6-s),   d_s in [-128, 127]   (d_0 in [-65, 65]).
// A dot product is sum_k X_k Y_k = sum_{s,t} 256^(12-s-t) P_st with P_st = sum_k d_s[k] d'_t[k]
// computed EXACTLY by the int8 MFMA (|P_st| <= 2^22); the 28 pairs with s + t <= 6 are kept, summed
// by group g = s + t in int32 (<= 7 2^22 < 2^25, exact) and combined in fp64 by a Horner chain.
// Error of z relative to |Z_m|_max |W_n|_max K: 2^-55 (input rounding, each operand) + 6 x 2^-54 at
// worst, ~2^-55 typically (dropped pairs s + t = 7), i.e. that of ONE fp64 rounding per product --
// the native kernel makes K of them.  Against a long-double reference the sliced product is as
// accurate as the native fp64 one when the entries of a vector are within a few orders of magnitude
// of its largest one (tests/test_kernels_gpu.py::test_heads_sliced_*: both ~3e-17 of sum |z||w| on
// average); a weight vector whose mean |entry| is below 2^-6 of its largest (a few dominant
// entries: the small ones would keep too few bits) is refused at build time
// (`usable = 0`: the caller keeps the fp64 kernel).  NaN / Inf in a vector poison its scale, hence
// every output that uses it, as in fp64.  The kernel is an INFERENCE path: the training tape keeps
// the fp64 kernels (weights change every step and the reverse sweep reuses their products).
//
// Kernel (details at heads_sliced_kernel).  One 512-thread workgroup per CU: four "matrix" wavefronts
// (16 chains each, their 7 x 4 int8 A-fragments = 112 VGPRs stationary in registers) and four "update"
// wavefronts, one of each per SIMD.  A workgroup owns 64 chains and walks a contiguous range of
// 16-entry output tiles; per tile the three heads' weight slices (3 chunks of 28 KB, stored
// pre-swizzled in MFMA fragment order by the build kernel) stream global -> LDS by LDS-DMA through
// three images, one barrier per chunk.  Per chunk a matrix wavefront issues 28 ds_read_b128 and 112
// MFMAs and hands its 7 int32 group sums to its update wavefront through an LDS ring; that one runs
// the Horner chain and, after the third chunk of a tile, the same epilogue as the fp64 kernel
// (heads_common.hpp).  The RG = M / 64 workgroups that need the same chunks run on ONE XCD (hardware
// block b -> XCD b % 8) and share its L2: W is read once from HBM (PMC: 1.005 x algorithmic).
// (Round 4: the eight per-column epilogue parameters through one 1 KB LDS-DMA piece per tile instead of
// eight 8-byte loads per update wavefront -- 32 fewer registers, 0 spills also in the MID variants, but 3 %
// SLOWER on one box (profiles/r04h_heads_ab.txt: 0.957 vs 0.930 ms): the loads were L1 hits; reverted.)
// (First version: 256-thread workgroups of four symmetric wavefronts, two per CU: 40-140 spilled
// VGPRs -- the A fragments, the fp64 constants of the epilogue and its operands do not fit one
// wavefront's 256 registers, and scratch traffic shares vmcnt with the LDS-DMA.)
#include "heads_common.hpp"
#include <type_traits>
#include <vector>

namespace l2q {
// (kernels outside an anonymous namespace: rocprofv3 tables then carry their names)

constexpr int OZ_NS = 7;                                  // int8 slices per operand
constexpr int OZ_BITS = 54;                               // fixed-point bits below the vector's exponent
constexpr int OZ_KB = 4;                                  // K = 256 = 4 k-blocks of one 16x16x64 MFMA
constexpr int OZ_K = 64 * OZ_KB;
constexpr int OZ_FRAG = 1024;                             // one operand fragment: 64 lanes x 16 bytes
constexpr int OZ_CHUNK = OZ_NS * OZ_KB * OZ_FRAG;         // 16 vectors x 256 x 7 slices = 28 KB

#ifndef L2Q_SL_PROF
#define L2Q_SL_PROF 0      // 1: per-wavefront cycle counters (barrier wait, total) into SlicedArgs::dbg
#endif
#ifndef L2Q_SL_V2
#define L2Q_SL_V2 1        // A/B: 0 = both tanh evaluations in the q head's period, degree-12 step-size exponentials
#endif
#ifndef L2Q_SL_SKIP
#define L2Q_SL_SKIP 0      // timing experiments (bits): 1 no epilogue arithmetic, 2 one MFMA per fragment group,
                           // 4 no LDS-DMA after the first images, 8 no Horner chain, 16 no ring writes
#endif
typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// ---- slicing: one wavefront per vector of K = 256 doubles (4 per lane) --------------------------
// out: chunk (r / 16) * cstride + coff, fragment layout [slice][kb][lane = (r % 16) + 16 (k % 64) / 16][k % 16]
// scale[r] = 2^(e - 30) (the product of a row and a column scale restores 2^(eZ + eW - 2 * 54 + 48)),
// 0 for padding vectors, NaN for a vector with a non-finite entry.
__global__ __launch_bounds__(256) void oz_split_kernel(const double* __restrict__ X, long R, long Rpad,
                                                        int cstride, int coff, char* __restrict__ out,
                                                        double* __restrict__ scale, int scale_pad,
                                                        int* __restrict__ flag, double rho_min) {
  const int lane = threadIdx.x & 63;
  const long r = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (r >= Rpad) return;
  double x[4] = {0.0, 0.0, 0.0, 0.0};
  if (r < R) {
    const double2 p0 = reinterpret_cast<const double2*>(X + r * OZ_K)[2 * lane];
    const double2 p1 = reinterpret_cast<const double2*>(X + r * OZ_K)[2 * lane + 1];
    x[0] = p0.x; x[1] = p0.y; x[2] = p1.x; x[3] = p1.y;
  }
  double mx = 0.0, sm = 0.0;
  int bad = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const double ax = fabs(x[c]);
    bad |= !(ax <= 1.7976931348623157e308);
    mx = fmax(mx, ax);
    sm += ax;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mx = fmax(mx, __shfl_xor(mx, off, 64));
    sm += __shfl_xor(sm, off, 64);
    bad |= __shfl_xor(bad, off, 64);
  }
  int e = 0;
  if (mx > 0.0 && !bad) (void)frexp(mx, &e);               // mx = f 2^e, f in [0.5, 1)
  unsigned pk[OZ_NS];
#pragma unroll
  for (int s = 0; s < OZ_NS; ++s) pk[s] = 0u;
  if (!bad) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      long long q = __double2ll_rn(ldexp(x[c], OZ_BITS - e));
#pragma unroll
      for (int s = OZ_NS - 1; s > 0; --s) {
        const int d = (int)((q + 128) & 0xFF) - 128;
        q = (q - d) >> 8;
        pk[s] |= (unsigned)(d & 0xFF) << (8 * c);
      }
      pk[0] |= (unsigned)((int)q & 0xFF) << (8 * c);
    }
  }
  const long chunk = (r >> 4) * cstride + coff;
  const int kb = lane >> 4;
  const int fl = (int)(r & 15) + 16 * ((lane & 15) >> 2);
  char* base = out + chunk * (long)OZ_CHUNK + kb * OZ_FRAG + fl * 16 + 4 * (lane & 3);
#pragma unroll
  for (int s = 0; s < OZ_NS; ++s) *reinterpret_cast<unsigned*>(base + s * OZ_KB * OZ_FRAG) = pk[s];
  if (lane == 0) {
    double sc = 0.0;
    if (r < R) sc = bad ? __longlong_as_double(0x7ff8000000000000LL) : ldexp(1.0, e - 30);
    if (r < R || scale_pad) scale[r] = sc;          // (W: [3][N] packed, no room for the padding vectors)
    if (flag && r < R && !bad && sm < rho_min * OZ_K * mx) atomicAdd(flag, 1);
  }
}

// rows of Z (activation vectors) whose mean |entry| was below 2^-6 of their largest one since the
// counter was last reset: the conditioning guard the weights get at build time, as a diagnostic for
// the per-call operand (one counter per device; l2q_heads_sliced_zflag reads it)
__device__ int oz_zflag_dev;
static int* oz_zflag_ptr() {
  static int* cache[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!cache[dev]) {
    int* p = nullptr;
    if (hipGetSymbolAddress((void**)&p, HIP_SYMBOL(oz_zflag_dev)) != hipSuccess) return nullptr;
    cache[dev] = p;
  }
  return cache[dev];
}

struct SlicedArgs {
  const char* Zs;          // [Mpad / 16] chunks
  const double* zscale;    // [Mpad]
  const char* Wsl;         // [ceil(N / 16)][3] chunks
  const double* wscale;    // [3][N]
  int ncw;                 // column workers = partial columns per chain
  int tiles_per_cw;
  int rg;                  // row groups of 64 chains
  long long* dbg;          // -DL2Q_SL_PROF builds: [block][wave][4] cycle counters
};

// Same element update as fused_heads_dma_kernel (gemm.hip), one element per call.
template <bool FWD, bool PAIR, bool MID>
__device__ __forceinline__ void heads_element(const HeadsArgs& a, double zs, double zt, double zq, double cs,
                                              double cq, double& vr, double& vi, double fr0, double fi0,
                                              bool ok, double& ld, double& ld1, double& ke) {
  const double eps = a.eps, heps = 0.5 * a.eps;
  const double s = cs * tanh_bf(zs);
  const double t = a.st * zt;
  const double q = cq * tanh_bf(zq);
  const double lj = FWD ? heps * s : -heps * s;
  const double es = exp_bf(lj), eq = exp_bf(eps * q);
  {
    const double fr = fr0 * eq + t, fi = fi0 * eq;
    if (FWD) { vr = es * vr - heps * fr; vi = es * vi - heps * fi; }
    else { vr = es * (vr + heps * fr); vi = es * (vi + heps * fi); }
  }
  if (!PAIR) {
    if (ok) ld += lj;
    return;
  }
  if (MID) { if (ok) { ld1 += lj; ke += fma(vr, vr, vi * vi); } }
  if (a.flip) { vr = -vr; vi = -vi; }
  const double h2 = 0.5 * a.eps2;
  const double lj2 = a.fwd2 ? h2 * s : -h2 * s;
  if (ok) ld += lj + lj2;
  double es2, eq2;
  if (a.eps2 == a.eps && a.fwd2 == (int)FWD) { es2 = es; eq2 = eq; }
  else { es2 = exp_bf(lj2); eq2 = exp_bf(a.eps2 * q); }
  const double fr = fr0 * eq2 + t, fi = fi0 * eq2;
  if (a.fwd2) { vr = es2 * vr - h2 * fr; vi = es2 * vi - h2 * fi; }
  else { vr = es2 * (vr + h2 * fr); vi = es2 * (vi + h2 * fi); }
}

// One 512-thread workgroup per CU = 8 wavefronts, two per SIMD with SEPARATE roles (the register file
// gives each 256 VGPRs; the two roles' live sets never meet):
//   * wavefronts 0-3, "matrix": 16 chains each, the 7 x 4 A-fragments stationary in registers; per chunk
//     28 ds_read_b128 (two fragments ahead), 112 MFMAs, 7 ds_write_b128 of the raw int32 group sums;
//     they also issue the LDS-DMA of the next chunk.  Nothing else: the MFMA pipe of the SIMD only
//     drains at the chunk barrier.
//   * wavefronts 4-7, "update": wavefront 4 + p takes the group sums of wavefront p one barrier later,
//     runs the Horner chain, and -- once the third head of a tile is in -- the epilogue of its
//     16 x 16 elements in three parts, one per chunk period, so that it arrives at each barrier before
//     the matrix wavefronts do (VALU per tile ~5000 cycles against 5376 MFMA cycles).  Its operands
//     (v, F, per-column parameters) are requested one tile ahead.
// All eight wavefronts pass one s_barrier per chunk.
#ifndef L2Q_SL_PRIO
#define L2Q_SL_PRIO 3
#endif
#ifndef L2Q_SL_NPD
#define L2Q_SL_NPD 4          // LDS-DMA pieces (of a wavefront pair's 7 per chunk) issued by the matrix wavefront
#endif
#ifndef L2Q_SL_CVT
#define L2Q_SL_CVT 0
#endif
// int32 -> fp64, exact
__device__ __forceinline__ double i2d(int x) {
#if L2Q_SL_CVT
  return __hiloint2double(0x43300000, x ^ 0x80000000) - 4503601774854144.0;   // 2^52 + 2^31 + x, minus the bias
#else
  return (double)x;
#endif
}
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// TAPE: the three heads (s, t, q as the update uses them) are stored as well -- the forward pass of the
// training tape; 12 more stores per lane and tile, issued with the tile's other stores (they are older
// than the next `fetch`, so the hand-counted vmcnt waits do not change).
template <bool CPLX, bool FWD, bool PAIR, bool MID, bool TAPE = false>
__global__ __launch_bounds__(512, 1) void heads_sliced_kernel(HeadsArgs a, SlicedArgs o) {
  constexpr int RING = OZ_NS * OZ_FRAG;                       // 7 KB: one wavefront's group sums of a chunk
  __shared__ __attribute__((aligned(1024))) char lds[3 * OZ_CHUNK + 2 * 4 * RING];           // 140 KB
  char* const ring = lds + 3 * OZ_CHUNK;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // hardware block b runs on XCD b % 8: the RG row groups of one column worker share an XCD
  const int xcd = blockIdx.x & 7, qb = blockIdx.x >> 3;
  const int rg = qb % o.rg, cw = xcd * (o.ncw >> 3) + qb / o.rg;
  const long NT = ((long)a.N + 15) >> 4;
  const long t0 = (long)cw * o.tiles_per_cw;
  long t1 = t0 + o.tiles_per_cw;
  if (t1 > NT) t1 = NT;
  const long ntl = t1 > t0 ? t1 - t0 : 0;
  const long nq = ntl * 3;
  const int p = wave & 3;
  const long rt = (long)rg * 4 + p;                           // this pair's 16 chains
  // C/D layout of v_mfma_i32_16x16x64_i8: col = lane & 15, row = 4 (lane >> 4) + reg
  const long mrow = rt * 16 + 4 * (lane >> 4);

  if (wave < 4) {
    // ================================================================ matrix wavefronts
    // Nothing but ds_read / MFMA / ds_write: a wavefront issues in order, so every other instruction
    // that takes longer than an MFMA's 16 cycles (an LDS-DMA piece: ~60, a cluster of ds_reads) drains
    // the matrix pipe.  The LDS-DMA of the weight images is issued by the update wavefronts.
    __builtin_amdgcn_s_setprio(L2Q_SL_PRIO);
    v4i32 A[OZ_NS][OZ_KB];
#pragma unroll
    for (int s = 0; s < OZ_NS; ++s)
#pragma unroll
      for (int kb = 0; kb < OZ_KB; ++kb)
        A[s][kb] = *reinterpret_cast<const v4i32*>(o.Zs + (rt * OZ_NS + s) * (long)(OZ_KB * OZ_FRAG) +
                                                   kb * OZ_FRAG + lane * 16);
    // B fragments of the even / odd k-block, read and waited for by hand: fragment r of the NEXT block
    // is requested after MFMA 4 r + 3 of the current one (28 MFMAs per block, slice j = 0 .. 6 in
    // groups of 7 - j), and group j waits with the lgkmcnt that leaves exactly the younger requests
    // outstanding (LDS returns in order).  The wait carries its fragment as an operand so that no
    // MFMA using it can be scheduled above.
    v4i32 bfa[OZ_NS], bfb[OZ_NS];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + lane * 16;
#define L2Q_SL_READ(dst, ad, j) \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(ad), "n"((j) * OZ_KB * OZ_FRAG))
#if L2Q_SL_PROF
    long long prof_lds = 0, prof_bar = 0;
    const long long prof_t0 = clock64();
#endif
    auto wait_for = [&](v4i32& x, auto cnt) {
      asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(decltype(cnt)::value));
    };
    using std::integral_constant;
    // one k-block: 28 MFMAs on `cur`, the seven fragments of the next block into `nxt` from LDS address nad
    // (dsrc != nullptr, first block of a chunk: also L2Q_SL_NPD pieces of the LDS-DMA of chunk qi + 2,
    // one every eight MFMAs; the update wavefront issues the rest -- the split that balances the two)
    auto block = [&](v4i32 (&acc)[OZ_NS], v4i32 (&cur)[OZ_NS], v4i32 (&nxt)[OZ_NS], int kb, unsigned nad,
                     const char* dsrc, char* ddst) {
      int idx = 0;
#pragma unroll
      for (int j = 0; j < OZ_NS; ++j) {
        if (j == 0 || j == 1 || j == OZ_NS - 1) wait_for(cur[j], integral_constant<int, 6>{});
        else wait_for(cur[j], integral_constant<int, 7>{});
#pragma unroll
        for (int i = 0; i + j < OZ_NS; ++i, ++idx) {
          if (!(L2Q_SL_SKIP & 2) || i == 0)
            acc[i + j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[i][kb], cur[j], acc[i + j], 0, 0, 0);
          if (dsrc && (idx & 7) == 1 && (idx >> 3) < L2Q_SL_NPD) {
            __builtin_amdgcn_sched_barrier(0);
            const int frag = p + 4 * (idx >> 3);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dsrc + frag * OZ_FRAG),
                                             (lds_ptr_t)(ddst + frag * OZ_FRAG), 16, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
          if ((idx & 3) == 3) {
            __builtin_amdgcn_sched_barrier(0);
            switch (idx >> 2) {
              case 0: L2Q_SL_READ(nxt[0], nad, 0); break;
              case 1: L2Q_SL_READ(nxt[1], nad, 1); break;
              case 2: L2Q_SL_READ(nxt[2], nad, 2); break;
              case 3: L2Q_SL_READ(nxt[3], nad, 3); break;
              case 4: L2Q_SL_READ(nxt[4], nad, 4); break;
              case 5: L2Q_SL_READ(nxt[5], nad, 5); break;
              default: L2Q_SL_READ(nxt[6], nad, 6); break;
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    };
    const char* wsrc = o.Wsl + t0 * 3 * (long)OZ_CHUNK + lane * 16;
    wg_barrier();                                             // B(0): images 0 and 1 are in LDS
    int st = 0;                                               // image of chunk qi = qi % 3
    if (nq > 0) {
#pragma unroll
      for (int j = 0; j < OZ_NS; ++j) {
        switch (j) {
          case 0: L2Q_SL_READ(bfa[0], lds0, 0); break;
          case 1: L2Q_SL_READ(bfa[1], lds0, 1); break;
          case 2: L2Q_SL_READ(bfa[2], lds0, 2); break;
          case 3: L2Q_SL_READ(bfa[3], lds0, 3); break;
          case 4: L2Q_SL_READ(bfa[4], lds0, 4); break;
          case 5: L2Q_SL_READ(bfa[5], lds0, 5); break;
          default: L2Q_SL_READ(bfa[6], lds0, 6); break;
        }
      }
    }
    for (long qi = 0; qi < nq; ++qi) {
      v4i32 acc[OZ_NS];
#pragma unroll
      for (int g = 0; g < OZ_NS; ++g) acc[g] = (v4i32){0, 0, 0, 0};
      const unsigned sb = lds0 + st * OZ_CHUNK;
      const int stn = st == 2 ? 0 : st + 1;
      // the next chunk's first block (its image was complete at B(qi)); the last chunk re-reads its own
      const unsigned nb0 = lds0 + (qi + 1 < nq ? stn : st) * OZ_CHUNK;
      // DMA(qi + 2) into the image chunk qi - 1 occupied (everybody left it before B(qi)); past the last
      // chunk the last one is fetched again into an image nobody reads
      const long qd = qi + 2 < nq ? qi + 2 : nq - 1;
      const int std_ = st == 0 ? 2 : st - 1;
      block(acc, bfa, bfb, 0, sb + 1 * OZ_FRAG, wsrc + qd * (long)OZ_CHUNK, lds + std_ * OZ_CHUNK);
      block(acc, bfb, bfa, 1, sb + 2 * OZ_FRAG, nullptr, nullptr);
      block(acc, bfa, bfb, 2, sb + 3 * OZ_FRAG, nullptr, nullptr);
      block(acc, bfb, bfa, 3, nb0, nullptr, nullptr);
      char* rb = ring + ((qi & 1) * 4 + p) * RING + lane * 16;
#pragma unroll
      for (int g = 0; g < OZ_NS; ++g)
        if (!((L2Q_SL_SKIP & 16) && g < 5)) *reinterpret_cast<v4i32*>(rb + g * OZ_FRAG) = acc[g];
#if L2Q_SL_PROF
      { const long long c0 = clock64(); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const long long c1 = clock64(); asm volatile("s_barrier" ::: "memory");
        prof_lds += c1 - c0; prof_bar += clock64() - c1; }
#else
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");        // B(qi + 1)
#endif
      st = stn;
    }
#undef L2Q_SL_READ
#if L2Q_SL_PROF
    if (o.dbg && lane == 0) {
      long long* d = o.dbg + ((long)blockIdx.x * 8 + wave) * 4;
      d[0] = clock64() - prof_t0; d[1] = prof_bar; d[2] = 0; d[3] = prof_lds;
    }
#endif
    return;
  }

  // ================================================================== update wavefronts
  double rs[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) rs[r] = o.zscale[mrow + r];
  double ld[4] = {0.0, 0.0, 0.0, 0.0}, ld1[4] = {0.0, 0.0, 0.0, 0.0}, ke[4] = {0.0, 0.0, 0.0, 0.0};
  // group sums of chunk qi -> sum_g 256^(6-g) P_g, scaled by the chain's 2^(e - 30)
  auto conv = [&](long qi, double (&S)[4]) {
    if (L2Q_SL_SKIP & 8) {
#pragma unroll
      for (int r = 0; r < 4; ++r) S[r] = rs[r] * (double)qi;
      return;
    }
    const char* rb = ring + ((qi & 1) * 4 + p) * RING + lane * 16;
    v4i32 gg[OZ_NS];
#pragma unroll
    for (int g = 0; g < OZ_NS; ++g) gg[g] = *reinterpret_cast<const v4i32*>(rb + g * OZ_FRAG);
    __builtin_amdgcn_sched_barrier(0);
    double x[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = i2d(gg[0][r]);
#pragma unroll
    for (int g = 1; g < OZ_NS; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) x[r] = fma(x[r], 256.0, i2d(gg[g][r]));
#pragma unroll
    for (int r = 0; r < 4; ++r) S[r] = x[r] * rs[r];
  };
  struct Operands {
    double2 vv[4], ff[4];
    double b0, b1, b2, pcs, pcq, w0, w1, w2;
  };
  // rows past M re-read the last valid chain, columns past N the last valid entry (never stored)
  long rowoff[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) rowoff[r] = ((mrow + r) < a.M ? (mrow + r) : (long)a.M - 1) * (long)a.N;
  auto fetch = [&](long t, Operands& q) {
    const long n = t * 16 + (lane & 15);
    const long nc = n < a.N ? n : (long)a.N - 1;
    q.b0 = a.b[0][nc]; q.b1 = a.b[1][nc]; q.b2 = a.b[2][nc];
    q.pcs = a.cs[nc];
    q.pcq = a.cq[nc];
    q.w0 = o.wscale[nc]; q.w1 = o.wscale[(long)a.N + nc]; q.w2 = o.wscale[2 * (long)a.N + nc];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long oo = rowoff[r] + nc;
      if (CPLX) {
        q.vv[r] = reinterpret_cast<const double2*>(a.vin)[oo];
        q.ff[r] = reinterpret_cast<const double2*>(a.F)[oo];
      } else {
        q.vv[r] = make_double2(a.vin[oo], 0.0);
        q.ff[r] = make_double2(a.F[oo], 0.0);
      }
    }
  };
  // The epilogue of a tile in three stages, one per chunk period, each over all four elements at once:
  // a single wavefront per SIMD has nobody to hide its dependent fp64 chains behind, so the stages
  // are cut ACROSS the elements (eight independent chains each) rather than element by element
  // (same operations in the same order as heads_element).
  double es_[4], eq_[4], es2_[4], eq2_[4], s_[4], q_[4], t_[4];
  const double eps = a.eps, heps = 0.5 * a.eps, h2 = 0.5 * a.eps2;
  const bool same2 = PAIR && a.eps2 == a.eps && a.fwd2 == (int)FWD;
#if L2Q_SL_V2
  // The three periods of a tile end at barriers shared with the matrix wavefronts, so the LONGEST period sets
  // the pace: with both tanh evaluations in the period of the q head it was 2 x 112 + 24 + the 56 of the
  // Horner chain = 304 fp64 instructions against 176 / 112 in the other two.  The s head of a tile arrives two
  // periods before its q head: its tanh moves into the (lightest) period right after its arrival -- the last
  // period of the PREVIOUS tile -- and rides in sn_ until the tile's own epilogue: 192 / 176 / 224.
  double sn_[4];
  auto tanh_s = [&](const Operands& q, const double (&fs)[4], double (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double zs = fma(fs[r], q.w0, q.b0);
      out[r] = (L2Q_SL_SKIP & 1) ? zs : q.pcs * tanh_bf(zs);
    }
  };
  auto stage_a = [&](const Operands& q, const double (&fs)[4], const double (&ft)[4], const double (&fq)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double zt = fma(ft[r], q.w1, q.b1);
      const double zq = fma(fq[r], q.w2, q.b2);
      s_[r] = sn_[r];
      if (L2Q_SL_SKIP & 1) { q_[r] = zq; t_[r] = zt; continue; }
      t_[r] = a.st * zt;
      q_[r] = q.pcq * tanh_bf(zq);
    }
  };
#else
  auto stage_a = [&](const Operands& q, const double (&fs)[4], const double (&ft)[4], const double (&fq)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double zs = fma(fs[r], q.w0, q.b0);
      const double zt = fma(ft[r], q.w1, q.b1);
      const double zq = fma(fq[r], q.w2, q.b2);
      if (L2Q_SL_SKIP & 1) { s_[r] = zs; q_[r] = zq; t_[r] = zt; continue; }
      s_[r] = q.pcs * tanh_bf(zs);
      t_[r] = a.st * zt;
      q_[r] = q.pcq * tanh_bf(zq);
    }
  };
#endif
  // exp of the step-size-scaled arguments: when every lane's |x| < 0.34 (k = rint(x log2 e) = 0) the
  // range reduction of exp_bf is the identity -- the polynomial alone gives the same bits
  auto exp_poly = [](double r) {
    double p = 1.0 / 479001600.0;
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    return fma(p, r, 1.0);
  };
#if L2Q_SL_V2
  // step-size-scaled arguments are ~1e-2: below 2^-6 the degree-7 polynomial is exact to 9e-20
  auto exp_poly7 = [](double r) {
    double p = 1.0 / 5040.0;
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    return fma(p, r, 1.0);
  };
#endif
  auto exp4x2 = [&](const double (&xa)[4], const double (&xb)[4], double (&ya)[4], double (&yb)[4]) {
    bool small = true;
#pragma unroll
    for (int r = 0; r < 4; ++r) small = small && fabs(xa[r]) < 0.34 && fabs(xb[r]) < 0.34;
#if L2Q_SL_V2
    bool tiny = true;
#pragma unroll
    for (int r = 0; r < 4; ++r) tiny = tiny && fabs(xa[r]) < 0x1p-6 && fabs(xb[r]) < 0x1p-6;
    if (__all(tiny)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { ya[r] = exp_poly7(xa[r]); yb[r] = exp_poly7(xb[r]); }
      return;
    }
#endif
    if (__all(small)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { ya[r] = exp_poly(xa[r]); yb[r] = exp_poly(xb[r]); }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) { ya[r] = exp_bf(xa[r]); yb[r] = exp_bf(xb[r]); }
    }
  };
  auto stage_b = [&] {
    if (L2Q_SL_SKIP & 1) return;
    double xa[4], xb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xa[r] = FWD ? heps * s_[r] : -heps * s_[r];
      xb[r] = eps * q_[r];
    }
    exp4x2(xa, xb, es_, eq_);
    if (PAIR && !same2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xa[r] = a.fwd2 ? h2 * s_[r] : -h2 * s_[r];
        xb[r] = a.eps2 * q_[r];
      }
      exp4x2(xa, xb, es2_, eq2_);
    }
  };
  auto stage_c = [&](long t, const Operands& q, double2 (&outv)[4]) {
    const long n = t * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = n < a.N && (mrow + r) < a.M;
      double vr = q.vv[r].x, vi = q.vv[r].y;
      const double fr0 = q.ff[r].x, fi0 = q.ff[r].y;
      if (L2Q_SL_SKIP & 1) { vr += s_[r] + t_[r] + fr0; vi += q_[r] + fi0; }
      else {
        const double s = s_[r], tt = t_[r];
        const double lj = FWD ? heps * s : -heps * s;
        {
          const double fr = fr0 * eq_[r] + tt, fi = fi0 * eq_[r];
          if (FWD) { vr = es_[r] * vr - heps * fr; vi = es_[r] * vi - heps * fi; }
          else { vr = es_[r] * (vr + heps * fr); vi = es_[r] * (vi + heps * fi); }
        }
        if (!PAIR) {
          if (ok) ld[r] += lj;
        } else {
          if (MID) { if (ok) { ld1[r] += lj; ke[r] += fma(vr, vr, vi * vi); } }
          if (a.flip) { vr = -vr; vi = -vi; }
          const double lj2 = a.fwd2 ? h2 * s : -h2 * s;
          if (ok) ld[r] += lj + lj2;
          const double e2 = same2 ? es_[r] : es2_[r], g2 = same2 ? eq_[r] : eq2_[r];
          const double fr = fr0 * g2 + tt, fi = fi0 * g2;
          if (a.fwd2) { vr = e2 * vr - h2 * fr; vi = e2 * vi - h2 * fi; }
          else { vr = e2 * (vr + h2 * fr); vi = e2 * (vi + h2 * fi); }
        }
      }
      outv[r] = make_double2(vr, vi);
    }
  };
  auto store = [&](long t, const double2 (&outv)[4]) {
    const long n = t * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (n < a.N && (mrow + r) < a.M) {
        const long oo = rowoff[r] + n;
        if (CPLX) reinterpret_cast<double2*>(a.v)[oo] = outv[r];
        else a.v[oo] = outv[r].x;
        if (TAPE) { a.tape_s[oo] = s_[r]; a.tape_t[oo] = t_[r]; a.tape_q[oo] = q_[r]; }
      }
    }
  };
  Operands cur, nxt;
  // LDS-DMA of the weight images (this wavefront's 7 of the 28 pieces of a chunk), one chunk per period:
  // DMA(k + 2) is issued right after B(k) and waited for before B(k + 1), so every image is complete a
  // whole chunk before the matrix wavefronts read it.  Past the last chunk the last one is fetched
  // again (into an image nobody reads): no branch in the instruction stream.
  const char* wsrc = o.Wsl + t0 * 3 * (long)OZ_CHUNK + lane * 16;
  auto issue = [&](long qi, int f0) {
    const long qc = qi < nq ? qi : nq - 1;
    const int stage = (int)(qi % 3);
    const char* src = wsrc + qc * (long)OZ_CHUNK;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = f0; f < OZ_NS; ++f) {
      const int frag = p + 4 * f;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + frag * OZ_FRAG),
                                       (lds_ptr_t)(lds + stage * OZ_CHUNK + frag * OZ_FRAG), 16, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
#if L2Q_SL_PROF
  long long prof_bar = 0, prof_mem = 0, prof_conv = 0;
  const long long prof_t0 = clock64();
#define L2Q_SL_T(acc, ...) do { const long long c0_ = clock64(); __VA_ARGS__; acc += clock64() - c0_; } while (0)
#define L2Q_SL_CBAR(vm) do { const long long c0_ = clock64();                                            \
    asm volatile("s_waitcnt vmcnt(" #vm ") lgkmcnt(0)\n\ts_barrier" ::: "memory"); prof_bar += clock64() - c0_; } while (0)
#else
#define L2Q_SL_T(acc, ...) do { __VA_ARGS__; } while (0)
#define L2Q_SL_CBAR(vm) asm volatile("s_waitcnt vmcnt(" #vm ") lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif
  // period k = between B(k) and B(k + 1): [stores of the previous tile] DMA(k + 2), the Horner chain of
  // chunk k - 1, one stage of the epilogue.  The barrier that ends a period waits for its DMA with the
  // vmcnt that leaves exactly the younger requests outstanding: the 16 operand loads of `fetch` in the
  // first period of a tile, nothing otherwise (a tile's stores are issued BEFORE the next period's DMA).
  double fs[4], ft[4], fq[4], gs[4], gt[4];
  double2 outv[4];
  if (ntl > 0) {
    issue(0, 0);
    issue(1, 0);
    fetch(t0, cur);
    L2Q_SL_CBAR(16);                                   // B(0)
    issue(2, L2Q_SL_NPD);
    L2Q_SL_CBAR(0);                                    // B(1): chunk 0 is in the ring
    issue(3, L2Q_SL_NPD);
    conv(0, fs);
#if L2Q_SL_V2
    tanh_s(cur, fs, sn_);
#endif
    L2Q_SL_CBAR(0);                                    // B(2)
    issue(4, L2Q_SL_NPD);
    conv(1, ft);
    for (long u = 0; u < ntl; ++u) {
      const long t = t0 + u;
      const bool more = u + 1 < ntl;
      L2Q_SL_CBAR(0);                                  // B(3u + 3)
      L2Q_SL_T(prof_mem, if (u > 0) store(t - 1, outv); issue(3 * u + 5, L2Q_SL_NPD));
      L2Q_SL_T(prof_conv, conv(3 * u + 2, fq));
      L2Q_SL_T(prof_mem, fetch(more ? t + 1 : t, nxt));
      stage_a(cur, fs, ft, fq);
      if (more) {
        L2Q_SL_CBAR(16);                               // B(3u + 4)
        L2Q_SL_T(prof_mem, issue(3 * u + 6, L2Q_SL_NPD));
        L2Q_SL_T(prof_conv, conv(3 * u + 3, gs));
      }
      stage_b();
      if (more) {
        L2Q_SL_CBAR(0);                                // B(3u + 5)
        L2Q_SL_T(prof_mem, issue(3 * u + 7, L2Q_SL_NPD));
        L2Q_SL_T(prof_conv, conv(3 * u + 4, gt));
      }
      stage_c(t, cur, outv);
#if L2Q_SL_V2
      if (more) tanh_s(nxt, gs, sn_);
#endif
      if (more) {
        cur = nxt;
#pragma unroll
        for (int r = 0; r < 4; ++r) { fs[r] = gs[r]; ft[r] = gt[r]; }
      }
    }
    store(t1 - 1, outv);
  } else {
    L2Q_SL_CBAR(0);                                    // B(0) = B(nq)
  }
#undef L2Q_SL_CBAR
#undef L2Q_SL_T
#if L2Q_SL_PROF
  if (o.dbg && lane == 0) {
    long long* d = o.dbg + ((long)blockIdx.x * 8 + wave) * 4;
    d[0] = clock64() - prof_t0; d[1] = prof_bar; d[2] = prof_mem; d[3] = prof_conv;
  }
#endif
  // (the trailing LDS-DMA pieces of the last tile land in images nobody reads: let them land before the
  // workgroup's LDS can be handed to the next one)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // ---- per-chain partials of this column worker (fixed order: deterministic)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double x = ld[r], x1 = ld1[r], xk = ke[r];
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
      x += __shfl_xor(x, off, 64);
      if (MID) { x1 += __shfl_xor(x1, off, 64); xk += __shfl_xor(xk, off, 64); }
    }
    const long m = mrow + r;
    if ((lane & 15) == 0 && m < a.M) {
      a.logdet_part[m * a.ncols_part + cw] = x;
      if (MID) {
        a.ld1_part[m * a.ncols_part + cw] = x1;
        a.ke_part[m * a.ncols_part + cw] = xk;
      }
    }
  }
}

static inline long sliced_chunks(long N) { return cdiv(N, 16) * 3; }
static inline size_t sliced_scale_off(long N) { return ((size_t)sliced_chunks(N) * OZ_CHUNK + 255) & ~(size_t)255; }
static inline int sliced_rg(int M) { return (int)cdiv(M, 64); }
static inline int sliced_ncw(int M) {
  const int per_xcd = 32 / sliced_rg(M);        // one workgroup per CU, 32 CUs per XCD
  return 8 * (per_xcd > 0 ? per_xcd : 1);
}

}  // namespace l2q

using namespace l2q;

static int sliced_launch(const char* who, double* tape_s, double* tape_t, double* tape_q, const double* Z, int M,
                         int K, long N, const void* sliced,
                                      const double* bs, const double* cs, double scale_s, const double* bt,
                                      double scale_t, const double* bq, const double* cq, double scale_q,
                                      const void* v_in, void* v, const void* force, int is_complex, double eps1,
                                      int forward1, int pair, int flip_between, double eps2, int forward2,
                                      double* logdet, double* logdet1, double* vnorm2_mid, void* ws,
                                      size_t ws_bytes, void* stream) {
  L2Q_REQUIRE(Z && sliced && bs && bt && bq && cs && cq && v && force && logdet && ws, L2Q_EINVAL,
              "null pointer (the sliced kernel takes per-entry scales cs / cq)");
  L2Q_REQUIRE(K == OZ_K, L2Q_ESHAPE, "the sliced heads kernel serves K = 256");
  L2Q_REQUIRE(M > 0 && N > 0 && N < 2000000000L, L2Q_EINVAL, "bad size");
  const bool mid = logdet1 != nullptr;
  L2Q_REQUIRE(!mid || (pair && vnorm2_mid), L2Q_EINVAL, "mid-point outputs belong to the pair kernel");
  L2Q_REQUIRE(ws_bytes >= l2q_vnet_heads_sliced_ws_bytes(M, N), L2Q_ESHAPE, "workspace too small");
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const void* vin = v_in ? v_in : v;
  L2Q_REQUIRE(al(Z) && al(v) && al(vin) && al(force) && (reinterpret_cast<uintptr_t>(sliced) & 255) == 0 &&
                  (reinterpret_cast<uintptr_t>(ws) & 255) == 0,
              L2Q_ESHAPE, "operands must be 16-byte aligned (slice buffer and workspace 256-byte)");
  hipStream_t st = (hipStream_t)stream;
  const int rg = sliced_rg(M), ncw = sliced_ncw(M);
  const long mpad = (long)rg * 64;
  char* zs = (char*)ws;
  double* zscale = (double*)(zs + (mpad / 16) * OZ_CHUNK);
  double* part = zscale + mpad;
  hipLaunchKernelGGL(oz_split_kernel, dim3((unsigned)cdiv(mpad, 4)), dim3(256), 0, st, Z, (long)M, mpad, 1, 0,
                     zs, zscale, 1, oz_zflag_ptr(), 0x1p-6);
  HeadsArgs a;
  a.tape_s = tape_s; a.tape_t = tape_t; a.tape_q = tape_q;
  a.Z = Z; a.W[0] = a.W[1] = a.W[2] = nullptr;
  a.b[0] = bs; a.b[1] = bt; a.b[2] = bq; a.cs = cs; a.cq = cq;
  a.ss = scale_s; a.st = scale_t; a.sq = scale_q; a.eps = eps1; a.eps2 = eps2; a.fwd2 = forward2;
  a.flip = flip_between;
  a.v = (double*)v; a.vin = (const double*)vin; a.F = (const double*)force;
  a.logdet_part = part; a.ld1_part = part + (size_t)M * ncw; a.ke_part = part + 2 * (size_t)M * ncw;
  a.M = M; a.N = (int)N; a.K = K; a.ncols_part = ncw;
  SlicedArgs o;
  o.Zs = zs; o.zscale = zscale; o.Wsl = (const char*)sliced;
  o.wscale = (const double*)((const char*)sliced + sliced_scale_off(N));
  o.ncw = ncw; o.rg = rg; o.dbg = nullptr;
#if L2Q_SL_PROF
  static long long* dbg = nullptr;
  if (!dbg) (void)hipMalloc(&dbg, 4096 * 8 * 4 * sizeof(long long));
  (void)hipMemsetAsync(dbg, 0, 4096 * 8 * 4 * sizeof(long long), st);
  o.dbg = dbg;
#endif
  o.tiles_per_cw = (int)cdiv(cdiv(N, 16), ncw);
  const dim3 grid((unsigned)(rg * ncw)), block(512);
#define L2Q_SL(C, F, P, MD) hipLaunchKernelGGL((heads_sliced_kernel<C, F, P, MD>), grid, block, 0, st, a, o)
#define L2Q_SL_F(C, P, MD) do { if (forward1) L2Q_SL(C, true, P, MD); else L2Q_SL(C, false, P, MD); } while (0)
#define L2Q_SL_C(P, MD) do { if (is_complex) L2Q_SL_F(true, P, MD); else L2Q_SL_F(false, P, MD); } while (0)
  if (tape_s) {
    if (is_complex) {
      if (forward1) hipLaunchKernelGGL((heads_sliced_kernel<true, true, false, false, true>), grid, block, 0, st, a, o);
      else hipLaunchKernelGGL((heads_sliced_kernel<true, false, false, false, true>), grid, block, 0, st, a, o);
    } else {
      if (forward1) hipLaunchKernelGGL((heads_sliced_kernel<false, true, false, false, true>), grid, block, 0, st, a, o);
      else hipLaunchKernelGGL((heads_sliced_kernel<false, false, false, false, true>), grid, block, 0, st, a, o);
    }
  } else if (mid) L2Q_SL_C(true, true);
  else if (pair) L2Q_SL_C(true, false);
  else L2Q_SL_C(false, false);
#undef L2Q_SL_C
#undef L2Q_SL_F
#undef L2Q_SL
#if L2Q_SL_PROF
  {
    (void)hipStreamSynchronize(st);
    const int nblk = rg * ncw;
    std::vector<long long> h((size_t)nblk * 8 * 4);
    (void)hipMemcpy(h.data(), o.dbg, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double tot[2] = {0, 0}, bar[2] = {0, 0}, vm = 0, ldsw = 0, mx[2] = {0, 0}, umem = 0, uconv = 0;
    for (int b = 0; b < nblk; ++b)
      for (int w = 0; w < 8; ++w) {
        const long long* d = &h[((size_t)b * 8 + w) * 4];
        const int c = w >= 4;
        tot[c] += d[0]; bar[c] += d[1]; if (!c) { vm += d[2]; ldsw += d[3]; }
        else { umem += d[2]; uconv += d[3]; }
        if (d[0] > mx[c]) mx[c] = (double)d[0];
      }
    const double n = nblk * 4.0;
    fprintf(stderr, "[sliced prof] matrix waves: total %.0f (max %.0f) barrier %.0f vmcnt %.0f lgkmcnt %.0f | update waves: total %.0f barrier %.0f dma+fetch+store issue %.0f conv %.0f  (clock64 ticks, mean per wave)\n",
            tot[0] / n, mx[0], bar[0] / n, vm / n, ldsw / n, tot[1] / n, bar[1] / n, umem / n, uconv / n);
  }
#endif
  launch_finalize(a.logdet_part, logdet, M, ncw, 1, 1.0, 0.0, st);
  if (mid) {
    launch_finalize(a.ld1_part, logdet1, M, ncw, 1, 1.0, 0.0, st);
    launch_finalize(a.ke_part, vnorm2_mid, M, ncw, 1, 1.0, 0.0, st);
  }
  return check_launch(who);
}

extern "C" {

size_t l2q_heads_sliced_bytes(int K, long N) {
  if (K != OZ_K || N <= 0) return 0;
  return sliced_scale_off(N) + (size_t)3 * N * sizeof(double) + 256;
}

int l2q_heads_sliced_build(const double* Ws, const double* Wt, const double* Wq, int K, long N, void* sliced,
                           size_t sliced_bytes, int* usable, void* stream) {
  L2Q_REQUIRE(Ws && Wt && Wq && sliced && usable, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(K == OZ_K, L2Q_ESHAPE, "the sliced heads kernel serves K = 256");
  L2Q_REQUIRE(N > 0 && N < 2000000000L, L2Q_EINVAL, "bad size");
  L2Q_REQUIRE(sliced_bytes >= l2q_heads_sliced_bytes(K, N), L2Q_ESHAPE, "buffer too small");
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  L2Q_REQUIRE(al(Ws) && al(Wt) && al(Wq) && (reinterpret_cast<uintptr_t>(sliced) & 255) == 0, L2Q_ESHAPE,
              "operands must be 16-byte aligned (the slice buffer 256-byte)");
  hipStream_t st = (hipStream_t)stream;
  char* buf = (char*)sliced;
  double* wscale = (double*)(buf + sliced_scale_off(N));
  int* flag = (int*)(wscale + 3 * N);
  (void)hipMemsetAsync(flag, 0, sizeof(int), st);
  const long Npad = cdiv(N, 16) * 16;
  const double* W[3] = {Ws, Wt, Wq};
  for (int h = 0; h < 3; ++h)
    hipLaunchKernelGGL(oz_split_kernel, dim3((unsigned)cdiv(Npad, 4)), dim3(256), 0, st, W[h], N, Npad, 3, h,
                       buf, wscale + (long)h * N, 0, flag, 0x1p-6);
  int hflag = 0;
  if (hipMemcpyAsync(&hflag, flag, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) {
    set_error("l2q_heads_sliced_build: %s", hipGetErrorString(hipGetLastError()));
    return L2Q_EHIP;
  }
  *usable = hflag ? 0 : 1;
  return check_launch("l2q_heads_sliced_build");
}

int l2q_heads_sliced_zflag(int reset, int* count, void* stream) {
  L2Q_REQUIRE(count, L2Q_EINVAL, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  int* p = oz_zflag_ptr();
  L2Q_REQUIRE(p, L2Q_EHIP, "device symbol not found");
  if (hipMemcpyAsync(count, p, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
      (reset && hipMemsetAsync(p, 0, sizeof(int), st) != hipSuccess) ||
      hipStreamSynchronize(st) != hipSuccess) {
    set_error("l2q_heads_sliced_zflag: %s", hipGetErrorString(hipGetLastError()));
    return L2Q_EHIP;
  }
  return L2Q_OK;
}

size_t l2q_vnet_heads_sliced_ws_bytes(int M, long N) {
  if (M <= 0 || N <= 0) return 0;
  const size_t mpad = (size_t)sliced_rg(M) * 64;
  return (mpad / 16) * OZ_CHUNK + mpad * sizeof(double) + (size_t)M * sliced_ncw(M) * 3 * sizeof(double) + 512;
}

int l2q_vnet_heads_vupdate_sliced_f64(const double* Z, int M, int K, long N, const void* sliced,
                                      const double* bs, const double* cs, double scale_s, const double* bt,
                                      double scale_t, const double* bq, const double* cq, double scale_q,
                                      const void* v_in, void* v, const void* force, int is_complex, double eps1,
                                      int forward1, int pair, int flip_between, double eps2, int forward2,
                                      double* logdet, double* logdet1, double* vnorm2_mid, void* ws,
                                      size_t ws_bytes, void* stream) {
  return sliced_launch("l2q_vnet_heads_vupdate_sliced_f64", nullptr, nullptr, nullptr, Z, M, K, N, sliced, bs, cs,
                       scale_s, bt, scale_t, bq, cq, scale_q, v_in, v, force, is_complex, eps1, forward1, pair,
                       flip_between, eps2, forward2, logdet, logdet1, vnorm2_mid, ws, ws_bytes, stream);
}

int l2q_vnet_heads_vupdate_sliced_tape_f64(const double* Z, int M, int K, long N, const void* sliced,
                                           const double* bs, const double* cs, const double* bt, double scale_t,
                                           const double* bq, const double* cq, const void* v_in, void* v,
                                           const void* force, int is_complex, double eps, int forward,
                                           double* s_out, double* t_out, double* q_out, double* logdet, void* ws,
                                           size_t ws_bytes, void* stream) {
  L2Q_REQUIRE(s_out && t_out && q_out, L2Q_EINVAL, "null pointer");
  return sliced_launch("l2q_vnet_heads_vupdate_sliced_tape_f64", s_out, t_out, q_out, Z, M, K, N, sliced, bs, cs, 1.0,
                       bt, scale_t, bq, cq, 1.0, v_in, v, force, is_complex, eps, forward, 0, 0, 0.0, 0, logdet,
                       nullptr, nullptr, ws, ws_bytes, stream);
}

}  // extern "C"
