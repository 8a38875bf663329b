// tile_stream_probe.hip -- how fast can a [M][K] fp32 matrix be streamed from HBM when each workgroup reads a
// TILE of it (ROWS rows x CB bytes per step, row stride K * 4 bytes) instead of a contiguous range?  This is the
// access pattern of the A operand of a GEMM (gemm_f16_skinny.hip) and of the field operands of
// u1_heads_update_h_kernel.  hipcc --offload-arch=gfx950 -O3 tile_stream_probe.hip -o tile_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ROWS, int CB, int D, int OCC>
__global__ __launch_bounds__(256, OCC) void tile_read(const float* __restrict__ A, long K, int splits, float* out) {
  constexpr int LPR = CB / 16;             // lanes per row
  constexpr int RPP = 256 / LPR;           // rows per pass
  constexpr int LPT = ROWS / RPP;          // loads per thread per step
  static_assert(LPT >= 1, "tile too small");
  const int tid = threadIdx.x;
  const long z = blockIdx.x % splits, mt = blockIdx.x / splits;
  const long kchunk = K / splits;          // floats
  const long steps = kchunk * 4 / CB;
  const float* base = A + (mt * ROWS + tid / LPR) * K + z * kchunk + (tid % LPR) * 4;
  float4 ring[D][LPT];
  float4 acc = make_float4(0, 0, 0, 0);
  auto fetch = [&](int e, long s) {
    if (s >= steps) s = steps - 1;
#pragma unroll
    for (int p = 0; p < LPT; ++p)
      ring[e][p] = *reinterpret_cast<const float4*>(base + (long)p * RPP * K + s * (CB / 4));
  };
#pragma unroll
  for (int e = 0; e < D; ++e) fetch(e, e);
  for (long sb = 0; sb < steps; sb += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
#pragma unroll
      for (int p = 0; p < LPT; ++p) {
        acc.x += ring[u][p].x; acc.y += ring[u][p].y; acc.z += ring[u][p].z; acc.w += ring[u][p].w;
      }
      fetch(u, sb + u + D);
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x * 256 + tid] = acc.x;
}

// reference: contiguous float4 grid-stride copy-less read
__global__ __launch_bounds__(256) void flat_read(const float4* __restrict__ A, long n4, float* out) {
  float4 acc = make_float4(0, 0, 0, 0);
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride * 4) {
    float4 a = A[i], b = i + stride < n4 ? A[i + stride] : a, c = i + 2 * stride < n4 ? A[i + 2 * stride] : a,
           d = i + 3 * stride < n4 ? A[i + 3 * stride] : a;
    acc.x += a.x + b.x + c.x + d.x; acc.y += a.y + b.y + c.y + d.y;
    acc.z += a.z + b.z + c.z + d.z; acc.w += a.w + b.w + c.w + d.w;
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[threadIdx.x] = acc.x;
}

template <typename F>
static float time_us(F launch, int reps = 10) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / reps;
}

int main(int argc, char** argv) {
  const long M = 16384, K = argc > 1 ? atol(argv[1]) : 8192;      // 537 MB at K = 8192: past the 256 MB L3
  float* A; float* out;
  hipMalloc(&A, M * K * 4);
  hipMalloc(&out, 1 << 24);
  hipMemset(A, 0, M * K * 4);
  const double gb = M * K * 4 * 1e-9;
  {
    float t = time_us([&] { hipLaunchKernelGGL(flat_read, dim3(256 * 8), dim3(256), 0, 0, (const float4*)A, M * K / 4, out); });
    printf("flat float4 read, 2048 WGs                         : %7.1f us  %.2f TB/s\n", t, gb / t * 1e-3 * 1e3);
  }
#define RUN(ROWS, CB, D, OCC, S)                                                                              \
  {                                                                                                           \
    float t = time_us([&] {                                                                                   \
      hipLaunchKernelGGL((tile_read<ROWS, CB, D, OCC>), dim3((unsigned)(M / ROWS * S)), dim3(256), 0, 0, A, K, S, out); \
    });                                                                                                       \
    printf("tile %3d rows x %4d B, depth %d, occ %d, splits %d, %5ld WGs: %7.1f us  %.2f TB/s\n", ROWS, CB, D, OCC, S, \
           (long)(M / ROWS * S), t, gb / t);                                                                  \
  }
  RUN(64, 128, 2, 3, 2) RUN(64, 128, 2, 3, 4) RUN(64, 128, 4, 3, 2) RUN(64, 128, 8, 3, 2)
  RUN(64, 256, 2, 3, 2) RUN(64, 256, 4, 3, 2)
  RUN(64, 512, 1, 3, 2) RUN(64, 512, 2, 3, 2) RUN(64, 512, 4, 2, 2)
  RUN(64, 1024, 1, 3, 2) RUN(64, 1024, 2, 2, 2)
  RUN(32, 1024, 2, 3, 1) RUN(16, 2048, 2, 3, 1) RUN(16, 4096, 1, 3, 1)
  RUN(128, 128, 2, 3, 4) RUN(128, 256, 2, 3, 4) RUN(128, 512, 2, 2, 4)
  RUN(64, 128, 2, 8, 8) RUN(64, 256, 2, 8, 8) RUN(64, 512, 2, 4, 8)
  return 0;
}
