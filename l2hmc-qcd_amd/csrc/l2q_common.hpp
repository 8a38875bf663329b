// l2q_common.hpp -- error plumbing, launch helpers and order-stable block reductions.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#include "../../include/l2q.h"

namespace l2q {

void set_error(const char* fmt, ...);

struct Tuning {
  int plaq_occ = 2;
  int force_occ = 2;
  int xcd_swizzle = 1;
  int plaq_sweep = 2;     // 2: slice-resident thread-per-site kernel (LDS + register prefetch), 3: slice-resident
                          // with the six planes split over wavefronts (su3_plaq_nu.hip; measured slower),
                          // 1: L2 t-sweep, 0: flat
  int heads_dma = 1;      // heads + v-update: LDS-DMA staged kernel (0: register-staged kernel of round 1)
  int force_tsplit = 0;   // su3_force_link.hip: 0 = t-range chunks chosen by the launcher, n = that many chunks per chain
  int force_stagger = 0;  // x ~2k cycles initial delay of the 2nd resident workgroup set (su3_force_link.hip)
  int heads_stagger = 0;  // x ~8k cycles initial delay of the 2nd resident block set (heads kernel)
  int force_tile = 5;     // 7: plaquettes shared between their four links, one 8-wavefront workgroup per CU (su3_force_plaq.hip;
                          //    plain force on lattices whose (y, z) plane is the 64-site tile: 1.17x HBM traffic, not faster),
                          // 6: as 5 with two adjacent x-planes per workgroup (su3_force_pair.hip: half the x-halo),
                          // 5: slice-resident thread-per-link, streamed factors, 2 workgroups / CU
                          // (su3_force_link.hip), 4: staples split by plane over wavefronts (su3_force_nu.hip), 3: rows
                          // split over wavefronts (su3_force_rows.hip), 2: slice-resident
                          // thread-per-link, 1: LDS-tiled (64 sites x 4 mu), 0: flat
  int u1_fused_ch = 0;    // chains per workgroup of the fused U(1) kernels (0: auto; 1, 2, 4, 8)
  int gemm_h_wide_fused = 0;  // large half GEMMs: 128 x 256 tile with 512 threads (measured slower)
  int conv_patch = 1;         // half conv: input patch staged in LDS by persistent workgroups for layers with
                              // <= 16 output channels (2: every layer it fits, 0: gather kernel only)
  int conv_stream = 0;        // half conv: 1 = persistent whole-K kernel (conv_stream_f16.hip; measured slower at cfg-3:
                              // 3.7 / 1.8 ms against the gather kernel's 2.6 / 1.2 ms), 0 = gather kernel
  int gemm_h_lt = 1;          // plain 16-bit layers with M, N, K >= 2048: hipBLASLt (gemm_lt.hip; 0: own kernels only)
  int gemm_h_dma = 1;         // big half GEMMs (M, N % 256 == 0, K % 64 == 0): LDS-DMA 256 x 256 kernel (0: off)
  int gemm_h_skinny = 1;      // wide-K fp32-operand input layer, N <= 256: streaming kernel (gemm_f16_skinny.hip);
                              // 0 off, 1 on (split count chosen), 2 / 4 / 8: that split count (A/B)
  int gemm_h_small = 1;       // 16-bit layers with K, N <= 256 on >= 2048 rows: one-pass kernel (gemm_f16_small.hip; 0: off)
  int gemm_h_patch = 1;       // half GEMM: 8 x 8 tile patches per XCD (0: row-major tile order)
  int heads_h_stream = 2; // half-precision heads+update: 2 = K-split stream kernel where its shape conditions hold
                          // (heads_kstream_f16.hip: K = 256 / 128 / 64, long streams; cfg-3: 0.22 / 0.32 ms per v- / x-update
                          // against the tile kernel's 0.36 / 0.42), 1 = round 3's weights-stationary stream kernel
                          // (0.475 ms), 0 = tile kernel only, 3 = as 2 for streams of any length (tests)
  int heads_h_bm = 128;   // chains per workgroup of the half-precision heads+update kernel (64 | 128)
  int heads_h_order = 1;  // half-precision heads+update kernel: 0 m-tiles fastest, 1 n-tiles fastest (consecutive
                          // workgroups walk along the rows of the fp32 field: 0.50 -> 0.41 ms at cfg-3)
};
Tuning& tuning();

// One-time per-DEVICE set-up of a kernel (hipFuncSetAttribute is a property of the function on
// the current device): `static PerDeviceOnce once; if (once.first()) { ... }`.
struct PerDeviceOnce {
  bool done[16] = {};
  bool first() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return true;
    if (done[dev]) return false;
    done[dev] = true;
    return true;
  }
};

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return L2Q_EHIP;
  }
  return L2Q_OK;
}

#define L2Q_REQUIRE(cond, code, msg)          \
  do {                                        \
    if (!(cond)) {                            \
      ::l2q::set_error("%s: %s", __func__, msg); \
      return code;                            \
    }                                         \
  } while (0)

constexpr int kBlock = 256;          // 4 wavefronts of 64
constexpr int kXcds = 8;             // MI355X: 8 XCDs, block b is observed on XCD b % 8

// XCD-aware remap of a 1-D grid: hardware block `b` runs on XCD b % 8, so give each XCD a
// contiguous range of logical work (neighbouring site blocks of the same chains share the
// XCD-private L2).  Speed only -- any placement is correct.
__device__ __forceinline__ long xcd_swizzle(long b, long total, int enable = 1) {
  if (!enable || total % kXcds != 0) return b;
  const long per = total / kXcds;
  return (b % kXcds) * per + b / kXcds;
}

// Sum of `v` over the 256 threads of a block, returned on thread 0.  Fixed tree: wave
// butterfly via DPP/shuffles, then 4 wave partials through LDS -- same order every run.
__device__ __forceinline__ double block_sum(double v, double* lds /* >= 4 doubles */) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw; ++w) r += lds[w];
  }
  return r;
}

__device__ __forceinline__ double block_max(double v, double* lds) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    r = lds[0];
    for (int w = 1; w < nw; ++w) r = fmax(r, lds[w]);
  }
  return r;
}

inline long cdiv(long a, long b) { return (a + b - 1) / b; }

// second stage of the per-chain reductions: partial[c][nblk][ncomp] -> out[c][ncomp],
// one block per chain, sequential fixed order over blocks inside each lane, then block_sum.
// out[c][k] = scale * sum_b partial[c][b][k] + offset
void launch_finalize(const double* partial, double* out, int nb, long nblk, int ncomp,
                     double scale, double offset, hipStream_t st);

// Zero `bytes` of device memory with a KERNEL.  The launch paths use this instead of hipMemsetAsync: a memset becomes
// a memset NODE when the call is captured into a HIP graph, and on ROCm 7.0 a replayed graph that contains one is
// corrupted by a >= 512 KB device-to-host copy on the null stream (round 5: every later replay of a captured SU(3)
// trajectory came out NaN; profiles/r05k_graph_memset_node.txt).  No launch path of this library records a memset node.
void launch_zero(void* p, size_t bytes, hipStream_t st);

}  // namespace l2q
