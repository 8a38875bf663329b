// fp64 FMA issue rate per SIMD against the number of wavefronts on it (gfx950): does a lone wavefront reach
// the full rate?   hipcc --offload-arch=gfx950 -O3 -o dfma_waves dfma_waves.hip && ./dfma_waves
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T>
__global__ void k(T* out, int iters) {
  T acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (T)i;
  const T a = (T)1.0 + (T)threadIdx.x * (T)1e-9, b = (T)1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = acc[i] * a + b;
  }
  T r = 0;
  for (int i = 0; i < 16; ++i) r += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <typename T>
static void run(const char* name, int threads) {
  T* out; hipMalloc(&out, 256 * 1024 * sizeof(T));
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<T>, dim3(256), dim3(threads), 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<T>, dim3(256), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = 16.0 * iters * (threads / 256.0);       // wave-instructions per SIMD
  printf("%s %d waves/SIMD: %.3f ms, %.2f ns per wave-instruction per SIMD, %.1f TFLOP/s\n", name, threads / 256, ms,
         ms * 1e6 / per_simd, 2.0 * 64 * per_simd * 1024 / (ms * 1e-3) / 1e12);
  hipFree(out);
}
int main() {
  for (int t : {256, 512, 1024}) run<double>("fp64 fma", t);
  for (int t : {256, 512, 1024}) run<float>("fp32 fma", t);
  return 0;
}
