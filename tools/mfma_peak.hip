// Micro-benchmark: sustained issue rate of v_mfma_f64_16x16x4_f64 and v_fma_f64 on gfx950.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_f64(double* out, int iters) {
  v4f64 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (v4f64){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void fma_f64(double* out, int iters) {
  double acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = i;
  double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = fma(acc[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
double timeit(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3;
}

int main() {
  double* out;
  const int blocks = 256 * 8, iters = 4000;
  hipMalloc(&out, blocks * 256 * sizeof(double));
  for (int wpb : {1, 2}) {
    const int nb = blocks * wpb / 2;
    double t = timeit([&] { hipLaunchKernelGGL(mfma_f64<8>, dim3(nb), dim3(256), 0, 0, out, iters); });
    double fl = (double)nb * 4 /*waves*/ * iters * 8 * 2048.0;
    printf("mfma_f64_16x16x4 x8 acc, %d blocks: %.2f TFLOP/s\n", nb, fl / t / 1e12);
  }
  double t = timeit([&] { hipLaunchKernelGGL(mfma_f64<2>, dim3(blocks), dim3(256), 0, 0, out, iters); });
  printf("mfma_f64_16x16x4 x2 acc: %.2f TFLOP/s\n", (double)blocks * 4 * iters * 2 * 2048.0 / t / 1e12);
  t = timeit([&] { hipLaunchKernelGGL(fma_f64, dim3(blocks), dim3(256), 0, 0, out, iters); });
  printf("v_fma_f64 x16 acc: %.2f TFLOP/s\n", (double)blocks * 256 * iters * 16 * 2.0 / t / 1e12);
  return 0;
}
