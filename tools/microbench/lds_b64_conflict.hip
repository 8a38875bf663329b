// lds_b64_conflict.hip -- what SQ_LDS_BANK_CONFLICT counts for ds_read_b64 / ds_read_b128 on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 lds_b64_conflict.hip -o lds_b64_conflict
// Run under   rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
//   mode 0: ds_read_b32, lane i -> byte 4 i                     (one pass, no conflict possible)
//   mode 1: ds_read_b64, lane i -> byte 8 i                     (512 contiguous bytes = the minimum two passes)
//   mode 2: ds_read_b64, the MFMA-f64 fragment pattern of fused_heads_dma_kernel (128-byte rows,
//           16-byte chunk ^ (row & 7), lane -> row lane & 15, k-pair lane >> 4)
//   mode 3: ds_read_b64 with a real 4-way conflict (lane i -> byte 256 (i & 3) * ... same bank pair)
//   mode 5: mode 2 with chunk ^ ((row >> 1) & 7): rows r and r + 8 no longer share a bank group
//   mode 4: ds_read_b128, lane i -> byte 16 i                   (1 KiB contiguous = the minimum four passes)
// If modes 1 and 2 show the same CONFLICT / IDX_ACTIVE ratio (~0.5) and mode 3 a higher one, the 47 %
// the heads kernel reports is the second pass of a conflict-free 8-byte read, not a layout problem.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>
__global__ void k(double* out, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[32768];
  for (int i = threadIdx.x; i < 32768 / 8; i += blockDim.x) reinterpret_cast<double*>(lds)[i] = i * 0.5;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  unsigned off;
  if (MODE == 0) off = lane * 4;
  else if (MODE == 1) off = lane * 8;
  else if (MODE == 2) {
    const int row = lane & 15;
    off = row * 128 + (((lane >> 5) ^ (lane & 7)) << 4) + ((lane >> 4) & 1) * 8;
  } else if (MODE == 5) {          // mode 2 with the chunk swizzle taken from (row >> 1) & 7
    const int row = lane & 15;
    off = row * 128 + (((lane >> 5) ^ ((lane >> 1) & 7)) << 4) + ((lane >> 4) & 1) * 8;
  } else if (MODE == 3) off = (lane & 15) * 256 + (lane >> 4) * 8;      // 16 rows of 256 B: same banks
  else off = lane * 16;
  double acc = 0.0;
  for (int it = 0; it < iters; ++it) {
    const unsigned o = off + (it & 7) * 2048;
    if (MODE == 0) {
      float v;
      asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(o) : "memory");
      acc += v;
    } else if (MODE == 4) {
      double2 v;
      asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(o) : "memory");
      acc += v.x + v.y;
    } else {
      double v;
      asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(o) : "memory");
      acc += v;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE>
void run(int iters) {
  double* out;
  hipMalloc(&out, sizeof(double) * 512 * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 0, 0, out, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 512.0 * 256 * iters * (MODE == 0 ? 4 : MODE == 4 ? 16 : 8);
  printf("mode %d: %.3f ms, %.1f B/clk/CU at 2.4 GHz\n", MODE, ms, bytes / (ms * 1e-3) / 256 / 2.4e9);
  hipFree(out);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  run<0>(iters); run<1>(iters); run<2>(iters); run<3>(iters); run<4>(iters); run<5>(iters);
  return 0;
}
