// dfma_peak.hip -- what the fp64 VALU / LDS pipes of an MI355X CU deliver in the instruction mix
// of the SU(3) staple kernels.  Build: hipcc --offload-arch=gfx950 -O3 dfma_peak.hip -o dfma_peak
//   mode 0: pure v_fma_f64, 12 independent accumulators per lane
//   mode 1: the row-vector staple pattern: 21 ds_read_b128 + 72 v_fma_f64 per "staple"
//   mode 2: only the 21 ds_read_b128 (+ one add per value so they are not dead)
// grid = 256 CUs x wgs_per_cu workgroups of `threads` threads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>
__global__ void k(double* out, int iters, double seed) {
  extern __shared__ double2 lds[];
  const int lane = threadIdx.x;
  for (int i = threadIdx.x; i < 9 * 64 * 4; i += blockDim.x) lds[i] = make_double2(seed * i, seed);
  __syncthreads();
  double a[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) a[j] = seed * (j + lane);
  double x = seed + lane, y = seed - lane;
  const double2* l = lds + (lane & 63);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int rep = 0; rep < 6; ++rep)
#pragma unroll
        for (int j = 0; j < 12; ++j) a[j] = fma(a[j], x, y);
    } else {
      double2 b[21];
      const int base = (it & 3) * 9 * 64;
#pragma unroll
      for (int e = 0; e < 21; ++e) b[e] = l[base + (e % 9) * 64 + ((e / 9) & 1) * 9 * 64];
      if (MODE == 1) {
#pragma unroll
        for (int rep = 0; rep < 3; ++rep)
#pragma unroll
          for (int j = 0; j < 12; ++j) {
            a[j] = fma(a[j], b[(j + rep * 4) % 21].x, b[(j + 7 + rep) % 21].y);
            a[j] = fma(a[j], b[(j + 3 + rep * 5) % 21].y, b[(j + 11 + rep) % 21].x);
          }
      } else {
#pragma unroll
        for (int e = 0; e < 21; ++e) a[e % 12] += b[e].x;
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int j = 0; j < 12; ++j) s += a[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(int threads, int wgs_per_cu, int iters) {
  const int grid = 256 * wgs_per_cu;
  double* out;
  hipMalloc(&out, sizeof(double) * grid * threads);
  const size_t ldsb = 9 * 64 * 4 * sizeof(double2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<grid, threads, ldsb>>>(out, 10, 1e-9);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<grid, threads, ldsb>>>(out, iters, 1e-9);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)grid * threads / 64;
  const double fma_per_lane = MODE == 2 ? 0 : 72.0 * iters;
  const double tf = 2.0 * fma_per_lane * grid * threads / (ms * 1e-3) / 1e12;
  const double ldsbytes = MODE == 0 ? 0 : 21.0 * 16 * iters * grid * threads;
  printf("mode %d threads %4d wg/cu %d (%.0f waves/SIMD): %.3f ms  %.1f TFLOP/s fp64 (%.0f%% of 78.6)  LDS %.1f TB/s\n",
         MODE, threads, wgs_per_cu, waves / 1024, ms, tf, 100 * tf / 78.6, ldsbytes / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  for (int t : {256, 512, 768, 1024}) run<0>(t, 1, 4000);
  run<0>(256, 2, 4000);
  for (int t : {256, 768, 1024}) run<1>(t, 1, 4000);
  run<1>(768, 2, 4000);
  for (int t : {256, 768, 1024}) run<2>(t, 1, 4000);
  return 0;
}
