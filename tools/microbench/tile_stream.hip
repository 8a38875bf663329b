// tile_stream.hip -- what an in-place update of an fp32 field a[M][N] (+ a second operand b[M][N]) reaches when
// it is walked in (BM rows x BN columns) tiles, the access shape of the fused "heads + update" kernels
// (a = v or x, b = force or v; cfg-3: M = N = 8192).  out: a = a * 1.0001f + b.
//   mode 0: linear (grid-stride float4 over the whole array): the streaming reference
//   mode 1: MFMA-accumulator mapping, direct float4 loads / stores (lane -> row lane & 15, 4 columns of
//           group lane >> 4), one tile per workgroup, tile order n-fastest            (= the shipped kernels)
//   mode 2: as 1, persistent: a workgroup owns BM rows and sweeps a column range, next step's operands
//           requested before the current step is consumed
//   mode 3: persistent, operands staged by LDS-DMA (global_load_lds_dwordx4: whole row segments, no VGPRs),
//           DEPTH steps ahead; consumed from LDS in the MFMA mapping; results written back through LDS as
//           whole row segments
// usage: tile_stream [BM] [BN] [NS] [DEPTH]   (defaults 128 32 8 2).  Buffers: 4 rotating sets (> MALL).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__global__ __launch_bounds__(256) void k_linear(float* a, const float* b, long n4) {
  float4* a4 = reinterpret_cast<float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 x = a4[i];
    const float4 y = b4[i];
    x.x = x.x * 1.0001f + y.x; x.y = x.y * 1.0001f + y.y; x.z = x.z * 1.0001f + y.z; x.w = x.w * 1.0001f + y.w;
    a4[i] = x;
  }
}

// wave w of 4 owns rows [w * BM / 4, ...) of the tile in 16-row MFMA tiles; columns in 16-wide groups
template <int BM, int BN>
__global__ __launch_bounds__(256) void k_tile(float* a, const float* b, int M, int N) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long nt = N / BN;
  const long m0 = (blockIdx.x / nt) * BM, n0 = (blockIdx.x % nt) * BN;
  constexpr int MI = BM / 64, NJ = BN / 16;
  float4 av[MI][NJ], bv[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const long o = (m0 + wave * (BM / 4) + 16 * i + (lane & 15)) * N + n0 + 16 * j + 4 * (lane >> 4);
      av[i][j] = *reinterpret_cast<const float4*>(a + o);
      bv[i][j] = *reinterpret_cast<const float4*>(b + o);
    }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const long o = (m0 + wave * (BM / 4) + 16 * i + (lane & 15)) * N + n0 + 16 * j + 4 * (lane >> 4);
      float4 x = av[i][j];
      const float4 y = bv[i][j];
      x.x = x.x * 1.0001f + y.x; x.y = x.y * 1.0001f + y.y; x.z = x.z * 1.0001f + y.z; x.w = x.w * 1.0001f + y.w;
      *reinterpret_cast<float4*>(a + o) = x;
    }
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void k_sweep(float* a, const float* b, int M, int N, int NS) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long m0 = (blockIdx.x / NS) * BM;
  const int ncols = N / NS, c0 = (blockIdx.x % NS) * ncols, nstep = ncols / BN;
  constexpr int MI = BM / 64, NJ = BN / 16;
  float4 av[2][MI][NJ], bv[2][MI][NJ];
  auto off = [&](int s, int i, int j) {
    return (m0 + wave * (BM / 4) + 16 * i + (lane & 15)) * (long)N + c0 + s * BN + 16 * j + 4 * (lane >> 4);
  };
  auto fetch = [&](int s, int slot) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        av[slot][i][j] = *reinterpret_cast<const float4*>(a + off(s, i, j));
        bv[slot][i][j] = *reinterpret_cast<const float4*>(b + off(s, i, j));
      }
  };
  fetch(0, 0);
  for (int s = 0; s < nstep; s += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int st = s + h;
      if (st >= nstep) break;
      if (st + 1 < nstep) fetch(st + 1, h ^ 1);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          float4 x = av[h][i][j];
          const float4 y = bv[h][i][j];
          x.x = x.x * 1.0001f + y.x; x.y = x.y * 1.0001f + y.y; x.z = x.z * 1.0001f + y.z; x.w = x.w * 1.0001f + y.w;
          *reinterpret_cast<float4*>(a + off(st, i, j)) = x;
        }
    }
  }
}

// LDS-DMA staged: stage = a tile [BM][BN] + b tile [BM][BN] fp32, row-major, (DEPTH + 1) stages in a ring.
// A DMA instruction moves 64 lanes x 16 B = 1 KB = 1024 / (BN * 4) rows.
template <int BM, int BN, int DEPTH>
__global__ __launch_bounds__(256) void k_dma(float* a, const float* b, int M, int N, int NS) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  constexpr int TILE = BM * BN * 4;                 // bytes of one operand tile
  constexpr int STAGE = 2 * TILE;
  constexpr int NST = DEPTH + 1;
  constexpr int RPI = 1024 / (BN * 4);              // rows per DMA instruction
  constexpr int NI = BM / RPI / 4;                  // instructions per wavefront, operand and stage
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long m0 = (blockIdx.x / NS) * BM;
  const int ncols = N / NS, c0 = (blockIdx.x % NS) * ncols, nstep = ncols / BN;
  const int lr = lane / (BN / 4), lc = lane % (BN / 4);   // row within the instruction, 16-byte column
  auto issue = [&](int s) {
    char* st = lds + (s % NST) * STAGE;
#pragma unroll
    for (int q = 0; q < NI; ++q) {
      const int g = q * 4 + wave;                    // instruction index: rows g * RPI ...
      const long o = (m0 + g * RPI + lr) * (long)N + c0 + s * BN + lc * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a + o),
                                       (lds_ptr_t)(st + g * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b + o),
                                       (lds_ptr_t)(st + TILE + g * 1024), 16, 0, 0);
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (d < nstep) issue(d);
  constexpr int MI = BM / 64, NJ = BN / 16;
  for (int s = 0; s < nstep; ++s) {
    // the DMA of step s is the oldest outstanding group: allow the younger ones to stay in flight
    // (in flight at this point: DMA(s) .. DMA(s + DEPTH - 1), 2 NI instructions each, and the NI stores of
    // step s - 1, which may retire out of order with the loads: waiting down to (DEPTH - 1) * 2 NI outstanding
    // guarantees DMA(s) whatever the stores did)
    if (DEPTH == 1 || s + DEPTH - 1 >= nstep) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((DEPTH - 1) * 2 * NI) : "memory");
    __syncthreads();
    if (s + DEPTH < nstep) issue(s + DEPTH);
    char* st = lds + (s % NST) * STAGE;
    // consume in the MFMA mapping, write the result back into the a tile
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int r = wave * (BM / 4) + 16 * i + (lane & 15), c = 16 * j + 4 * (lane >> 4);
        float4 x = *reinterpret_cast<const float4*>(st + (r * BN + c) * 4);
        const float4 y = *reinterpret_cast<const float4*>(st + TILE + (r * BN + c) * 4);
        x.x = x.x * 1.0001f + y.x; x.y = x.y * 1.0001f + y.y; x.z = x.z * 1.0001f + y.z; x.w = x.w * 1.0001f + y.w;
        *reinterpret_cast<float4*>(st + (r * BN + c) * 4) = x;
      }
    __syncthreads();
    // whole row segments back to memory (same lane -> address mapping as the DMA)
#pragma unroll
    for (int q = 0; q < NI; ++q) {
      const int g = q * 4 + wave;
      const long o = (m0 + g * RPI + lr) * (long)N + c0 + s * BN + lc * 4;
      *reinterpret_cast<float4*>(a + o) = *reinterpret_cast<const float4*>(st + g * 1024 + lane * 16);
    }
  }
}

int main(int argc, char** argv) {
  const int M = 8192, N = 8192;
  const int NS = argc > 1 ? atoi(argv[1]) : 8;
  const size_t n = (size_t)M * N;
  float *a[4], *b[4];
  for (int i = 0; i < 4; ++i) {
    CK(hipMalloc(&a[i], n * 4)); CK(hipMalloc(&b[i], n * 4));
    CK(hipMemset(a[i], 0, n * 4)); CK(hipMemset(b[i], 0, n * 4));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 4; ++i) launch(a[i], b[i]);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 12; ++r) launch(a[r & 3], b[r & 3]);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 12;
    printf("%-58s %.4f ms  %.2f TB/s\n", name, ms, 3.0 * n * 4 / ms / 1e9);
    fflush(stdout);
  };
  run("linear float4 (streaming reference)", [&](float* x, float* y) { hipLaunchKernelGGL(k_linear, dim3(2048), dim3(256), 0, 0, x, y, (long)(n / 4)); });
  run("tile 128 x 64, one per workgroup, n-fastest", [&](float* x, float* y) { hipLaunchKernelGGL((k_tile<128, 64>), dim3((M / 128) * (N / 64)), dim3(256), 0, 0, x, y, M, N); });
  run("tile 64 x 64, one per workgroup, n-fastest", [&](float* x, float* y) { hipLaunchKernelGGL((k_tile<64, 64>), dim3((M / 64) * (N / 64)), dim3(256), 0, 0, x, y, M, N); });
  run("sweep 128 x 32 direct, prefetch 1", [&](float* x, float* y) { hipLaunchKernelGGL((k_sweep<128, 32>), dim3((M / 128) * NS), dim3(256), 0, 0, x, y, M, N, NS); });
  run("sweep 128 x 64 direct, prefetch 1", [&](float* x, float* y) { hipLaunchKernelGGL((k_sweep<128, 64>), dim3((M / 128) * NS), dim3(256), 0, 0, x, y, M, N, NS); });
  run("sweep 64 x 64 direct, prefetch 1", [&](float* x, float* y) { hipLaunchKernelGGL((k_sweep<64, 64>), dim3((M / 64) * NS), dim3(256), 0, 0, x, y, M, N, NS); });
#define DMA(BM, BN, D)                                                                                       \
  do {                                                                                                       \
    const size_t lb = (size_t)(D + 1) * 2 * BM * BN * 4;                                                     \
    CK(hipFuncSetAttribute((const void*)k_dma<BM, BN, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb)); \
    char nm[96];                                                                                             \
    snprintf(nm, sizeof nm, "sweep %d x %d LDS-DMA, %d steps ahead (%zu KB LDS)", BM, BN, D, lb / 1024);     \
    run(nm, [&](float* x, float* y) { hipLaunchKernelGGL((k_dma<BM, BN, D>), dim3((M / BM) * NS), dim3(256), lb, 0, x, y, M, N, NS); }); \
  } while (0)
  DMA(128, 32, 1);
  DMA(128, 32, 2);
  DMA(128, 64, 1);
  DMA(64, 64, 1);
  DMA(64, 64, 2);
  DMA(64, 128, 1);
  return 0;
}
