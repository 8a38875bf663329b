// strip_rw.hip -- which C-tile access pattern reaches HBM bandwidth?  v[m][n] = f(v, F) over an
// [M][N] fp32 matrix, tiled like a GEMM epilogue.  Build: hipcc --offload-arch=gfx950 -O3 strip_rw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// P: 0 = MFMA-C pattern: tile BM x BN, wave covers 16 rows x 64 B per instruction (lane&15 = row)
//    1 = row pattern: tile BM x BN, 16 lanes x float4 cover BN=64 floats of a row, 4 rows / instr
//    2 = BN floats of a row contiguous across as many lanes as needed (BN >= 256: 1 row / instr)
template <int P, int BM, int BN>
__global__ __launch_bounds__(256) void k(float* __restrict__ a, const float* __restrict__ b, int M,
                                         int N, int mfast) {
  const int mt = M / BM, nt = N / BN;
  const int w = blockIdx.x;
  const long m0 = (mfast ? w % mt : w / nt) * (long)BM, n0 = (mfast ? w / mt : w % nt) * (long)BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (P == 0) {
    // 4 waves as 2 x 2, wave tile (BM/2) x (BN/2)
    const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
    for (int j = 0; j < BN / 2 / 16; ++j)
      for (int i = 0; i < BM / 2 / 16; ++i) {
        const long o = (m0 + wm + 16 * i + (lane & 15)) * N + n0 + wn + 16 * j + 4 * (lane >> 4);
        float4 x = *(const float4*)(a + o), y = *(const float4*)(b + o);
        x.x = x.x * 1.01f + y.x; x.y = x.y * 1.01f + y.y; x.z = x.z * 1.01f + y.z; x.w = x.w * 1.01f + y.w;
        *(float4*)(a + o) = x;
      }
  } else {
    constexpr int LPR = BN / 4;                 // lanes per row
    constexpr int RPI = 256 / LPR;              // rows per block-instruction
    const int r = tid / LPR, c = (tid % LPR) * 4;
    for (int i = 0; i < BM / RPI; ++i) {
      const long o = (m0 + r + (long)i * RPI) * N + n0 + c;
      float4 x = *(const float4*)(a + o), y = *(const float4*)(b + o);
      x.x = x.x * 1.01f + y.x; x.y = x.y * 1.01f + y.y; x.z = x.z * 1.01f + y.z; x.w = x.w * 1.01f + y.w;
      *(float4*)(a + o) = x;
    }
  }
}

template <int P, int BM, int BN>
void run(const char* name, float* a, float* b, int M, int N, int mfast) {
  const int grid = (M / BM) * (N / BN);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<P, BM, BN>), dim3(grid), dim3(256), 0, 0, a, b, M, N, mfast);
  hipEventRecord(e0);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<P, BM, BN>), dim3(grid), dim3(256), 0, 0, a, b, M, N, mfast);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  printf("%-44s mfast=%d  %8.1f us  %6.2f TB/s\n", name, mfast, us, 3.0 * M * N * 4 / us * 1e-6);
}

int main() {
  const int M = 8192, N = 8192;
  float *a, *b;
  hipMalloc(&a, (size_t)M * N * 4); hipMalloc(&b, (size_t)M * N * 4);
  hipMemset(a, 0, (size_t)M * N * 4); hipMemset(b, 0, (size_t)M * N * 4);
  for (int mf = 0; mf < 2; ++mf) {
    run<0, 128, 64>("MFMA-C lanes, tile 128x64", a, b, M, N, mf);
    run<0, 128, 128>("MFMA-C lanes, tile 128x128", a, b, M, N, mf);
    run<0, 64, 256>("MFMA-C lanes, tile 64x256", a, b, M, N, mf);
    run<1, 128, 64>("row lanes (256 B/row), tile 128x64", a, b, M, N, mf);
    run<1, 128, 128>("row lanes (512 B/row), tile 128x128", a, b, M, N, mf);
    run<1, 64, 256>("row lanes (1 KB/row), tile 64x256", a, b, M, N, mf);
    run<1, 32, 1024>("row lanes (4 KB/row), tile 32x1024", a, b, M, N, mf);
  }
  return 0;
}
