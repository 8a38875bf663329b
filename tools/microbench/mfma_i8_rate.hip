// Issue rate of v_mfma_i32_16x16x64_i8 from one / two wavefronts per SIMD, with independent and with
// chained accumulators, and with operand registers that change per instruction (as in heads_sliced.hip).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_i8_rate mfma_i8_rate.hip && ./mfma_i8_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i32 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(int* out, int iters, long long* cyc) {
  const int lane = threadIdx.x & 63;
  v4i32 a[8], b[7], acc[7];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (v4i32){lane + i, lane * 3 + i, 7 * i, lane ^ i};
#pragma unroll
  for (int i = 0; i < 7; ++i) { b[i] = (v4i32){lane * 5 + i, i, lane, 3 * i}; acc[i] = (v4i32){0, 0, 0, 0}; }
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {          // 28 MFMAs, 7 independent accumulators, operands vary
#pragma unroll
      for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int i = 0; i + j < 7; ++i)
          acc[i + j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], b[j], acc[i + j], 0, 0, 0);
    } else if (MODE == 1) {   // 28 MFMAs on ONE accumulator (chained)
#pragma unroll
      for (int j = 0; j < 28; ++j)
        acc[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[j & 7], b[j % 7], acc[0], 0, 0, 0);
    } else {                  // 28 MFMAs, same operands, 7 accumulators round robin
#pragma unroll
      for (int j = 0; j < 28; ++j)
        acc[j % 7] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[0], b[0], acc[j % 7], 0, 0, 0);
    }
  }
  const long long t1 = clock64();
  v4i32 s = acc[0];
#pragma unroll
  for (int i = 1; i < 7; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
static int run(const char* name, int threads) {
  int* out; long long* cyc;
  CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 8));
  const int iters = 20000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, 100, cyc);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  const double n = 28.0 * iters;                    // MFMAs per wavefront
  const double waves_per_simd = threads / 256.0;
  const double tops = 2.0 * 16 * 16 * 64 * n * (threads / 64) * 256 / (ms * 1e-3) / 1e12;
  printf("%-44s %d waves/SIMD: %7.3f ms  %7.1f TOP/s  clock64 ticks per MFMA per SIMD %.2f  (wall ns per MFMA per SIMD %.2f)\n",
         name, (int)waves_per_simd, ms, tops, (double)c / n / waves_per_simd, ms * 1e6 / n / waves_per_simd);
  return 0;
}

int main() {
  for (int t : {256, 512}) {
    if (t == 256) { run<0>("7 accumulators, operands vary", 256); run<1>("one accumulator (chained)", 256); run<2>("7 accumulators, same operands", 256); }
    else { run<0>("7 accumulators, operands vary", 512); run<1>("one accumulator (chained)", 512); run<2>("7 accumulators, same operands", 512); }
  }
  return 0;
}
