// Does VALU work of one wavefront overlap with MFMA work of another wavefront on the same SIMD
// (gfx950)?  First block: v_mfma_f64_16x16x4_f64 (no: the times add); second block (I8 = true):
// v_mfma_i32_16x16x64_i8.  512-thread workgroups, 1 per CU: wavefronts 0-3 (one per SIMD) run a
// v_mfma_f64_16x16x4_f64 loop, wavefronts 4-7 run a VALU loop of the given flavour.
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_valu_overlap.hip -o tools/bin/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef int v4i32 __attribute__((ext_vector_type(4)));

// mode bits: 1 = MFMA waves active, 2 = VALU waves active; flavour: 0 fp64 fma, 1 fp32 fma, 2 int mad
template <int FLAV, bool I8 = false>
__global__ __launch_bounds__(512) void k(double* out, int it_m, int it_v, int mode) {
  const int wave = threadIdx.x >> 6;
  double res = 0;
  if (wave < 4) {
    if ((mode & 1) && I8) {
      v4i32 acc[8], a, b;
      for (int i = 0; i < 8; ++i) acc[i] = (v4i32){0, 0, 0, 0};
      a = (v4i32){(int)threadIdx.x, 3, 5, 7}; b = (v4i32){1, (int)threadIdx.x * 3, 2, 9};
      for (int it = 0; it < it_m * 4; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
      }
      for (int i = 0; i < 8; ++i) res += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if (mode & 1) {
      v4f64 acc[8];
      for (int i = 0; i < 8; ++i) acc[i] = (v4f64){0, 0, 0, 0};
      double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
      for (int it = 0; it < it_m; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      }
      for (int i = 0; i < 8; ++i) res += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
  } else if (mode & 2) {
    if (FLAV == 0) {
      double acc[16];
      for (int i = 0; i < 16; ++i) acc[i] = i;
      double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
      for (int it = 0; it < it_v; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fma(acc[i], a, b);
      }
      for (int i = 0; i < 16; ++i) res += acc[i];
    } else if (FLAV == 1) {
      float acc[16];
      for (int i = 0; i < 16; ++i) acc[i] = i;
      float a = 1.0f + threadIdx.x * 1e-6f, b = 1e-6f;
      for (int it = 0; it < it_v; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fmaf(acc[i], a, b);
      }
      for (int i = 0; i < 16; ++i) res += acc[i];
    } else if (FLAV == 3) {
      // the fp64 -> balanced base-256 digit conversion of csrc/gemm_sliced.hip (10 instructions per value)
      double x[16];
      unsigned accu = 0;
      unsigned long long bad = 0;
      for (int i = 0; i < 16; ++i) x[i] = i * 0.01 + threadIdx.x * 1e-4;
      const double sc = 4503599627370496.0, M24 = 113336795588871485128704.0, M52 = 6755399441055744.0;
      for (int it = 0; it < it_v; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          bad |= __builtin_amdgcn_fcmp(fabs(x[i]), 4.0, 11);
          const double t1 = fma(x[i], sc, M24);
          const double r = fma(x[i], sc, M24 - t1);
          const double t2 = r + M52;
          const int H = (int)(unsigned)__double_as_longlong(t1);
          const int L = (int)(unsigned)__double_as_longlong(t2);
          const int Lb = L + 0x00808080;
          const unsigned lo = (unsigned)Lb ^ 0x00808080u;
          const unsigned hi = ((unsigned)H + (unsigned)(Lb >> 24) + 0x80808080u) ^ 0x80808080u;
          accu += lo ^ hi;
          asm volatile("" : "+v"(x[i]));
        }
      }
      res = accu + (double)bad;
    } else if (FLAV == 4) {
      unsigned acc[16];
      for (int i = 0; i < 16; ++i) acc[i] = i;
      unsigned a = 3 + threadIdx.x;
      for (int it = 0; it < it_v; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_perm(acc[i], a, 0x05010400u);
      }
      for (int i = 0; i < 16; ++i) res += acc[i];
    } else if (FLAV == 5) {
      double acc[16];
      for (int i = 0; i < 16; ++i) acc[i] = i;
      double a = 1.0 + threadIdx.x * 1e-9;
      for (int it = 0; it < it_v; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = acc[i] + a;
      }
      for (int i = 0; i < 16; ++i) res += acc[i];
    } else {
      unsigned acc[16];
      for (int i = 0; i < 16; ++i) acc[i] = i;
      unsigned a = 3 + threadIdx.x, b = 7;
      for (int it = 0; it < it_v; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = acc[i] * a + b;
      }
      for (int i = 0; i < 16; ++i) res += acc[i];
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = res;
}

template <typename F>
double timeit(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f(); f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

template <int FLAV, bool I8 = false>
void run(const char* name, double* out, int it_m, int it_v) {
  const int nb = 256;
  double t[4];
  for (int mode = 1; mode <= 3; ++mode)
    t[mode] = timeit([&] { hipLaunchKernelGGL((k<FLAV, I8>), dim3(nb), dim3(512), 0, 0, out, it_m, it_v, mode); });
  printf("%-10s MFMA alone %.3f ms (%.1f TFLOP/s)  VALU alone %.3f ms  both %.3f ms  (sum %.3f, max %.3f)\n", name, t[1],
         256.0 * 4 * it_m * 8 * 2048.0 / t[1] / 1e9, t[2], t[3], t[1] + t[2], t[1] > t[2] ? t[1] : t[2]);
}

int main() {
  double* out;
  hipMalloc(&out, 256 * 512 * sizeof(double));
  const int it_m = 4000;
  run<0>("fp64 fma", out, it_m, 8000);
  run<1>("fp32 fma", out, it_m, 16000);
  run<2>("int mad", out, it_m, 16000);
  run<0>("fp64 fma/2", out, it_m, 4000);
  printf("-- v_mfma_i32_16x16x64_i8 (the TFLOP/s column does not apply)\n");
  run<0, true>("fp64 fma", out, it_m, 8000);
  run<1, true>("fp32 fma", out, it_m, 16000);
  run<2, true>("int mad", out, it_m, 16000);
  run<3, true>("convert", out, it_m, 1600);
  run<4, true>("v_perm", out, it_m, 16000);
  run<5, true>("fp64 add", out, it_m, 8000);
  return 0;
}
