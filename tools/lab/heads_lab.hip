// heads_lab.hip -- timing laboratory for the fused vnet heads + momentum update kernel at the cfg-4
// shape (M = 256 chains, K = 256, N = 147456 entries, complex v / F).  Baseline = the shipped
// kernel of csrc/gemm.hip (copied into heads_base.inc with a MODE switch: 1 = K-loop only,
// 2 = epilogue only); variants under test live in heads_v2.inc.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I l2hmc-qcd_amd/csrc tools/lab/heads_lab.hip -o tools/bin/heads_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <type_traits>
#include "l2q_common.hpp"

#include "heads_base.inc"

using namespace l2q;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static double urand() { return (double)rand() / RAND_MAX * 2.0 - 1.0; }

template <typename F>
static double timeit(F f, int reps = 10) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}


#include <map>
#include <algorithm>
template <typename L>
static void trace_run(const char* name, L launch, size_t nblocks) {
  unsigned long long* tr;
  CK(hipMalloc(&tr, nblocks * 5 * 8));
  CK(hipMemset(tr, 0, nblocks * 5 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(l2q::g_trace), &tr, sizeof(tr)));
  launch();
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(nblocks * 5);
  CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
  CK(hipFree(tr));
  struct Iv { double t0, t1, t2; size_t b; };
  std::map<unsigned long long, std::vector<Iv>> cus;
  unsigned long long tmin = ~0ull;
  for (size_t b = 0; b < nblocks; ++b) tmin = std::min(tmin, h[4 * b]);
  double sk = 0, se = 0;
  for (size_t b = 0; b < nblocks; ++b) {
    const unsigned long long id = h[4 * b + 3];
    const unsigned long long key = ((id >> 32) << 16) | ((id >> 8) & 0xff);   // xcc, se/sh/cu
    Iv iv{(h[4 * b] - tmin) * 0.01, (h[4 * b + 1] - tmin) * 0.01, (h[4 * b + 2] - tmin) * 0.01, b};
    cus[key].push_back(iv);
    sk += iv.t1 - iv.t0; se += iv.t2 - iv.t1;
  }
  double cyc = 0, wall = 0;
  for (size_t b = 0; b < nblocks; ++b) { cyc += (double)h[4 * nblocks + b]; wall += (h[4 * b + 2] - h[4 * b]) * 0.01; }
  printf("      shader clock %.3f GHz\n", cyc / wall * 1e-3);
  double etot = 0, eov = 0, tend = 0;
  for (auto& kv : cus) {
    auto& v = kv.second;
    for (auto& x : v) {
      etot += x.t2 - x.t1;
      tend = std::max(tend, x.t2);
      // overlap of x's epilogue with other blocks' K-loops on this CU (union approximated by max single overlap)
      double best = 0;
      for (auto& y : v) if (y.b != x.b) best = std::max(best, std::min(x.t2, y.t1) - std::max(x.t1, y.t0));
      eov += std::max(0.0, best);
    }
  }
  printf("trace %-28s CUs %zu blocks/CU %.1f  mean K-loop %.2f us  mean epilogue %.2f us  epilogue covered by a co-resident K-loop: %.1f %%  span %.1f us\n",
         name, cus.size(), (double)nblocks / cus.size(), sk / nblocks, se / nblocks, 100.0 * eov / etot, tend);
  auto& v0 = cus.begin()->second;
  std::sort(v0.begin(), v0.end(), [](const Iv& a, const Iv& b) { return a.t0 < b.t0; });
  for (size_t i = 0; i < std::min<size_t>(v0.size(), 10); ++i)
    printf("    cu0 block %6zu  K [%7.2f, %7.2f]  E [%7.2f, %7.2f]\n", v0[i].b, v0[i].t0, v0[i].t1, v0[i].t1, v0[i].t2);
  fflush(stdout);
}

#include "heads_v2.inc"
#include "heads_v5.inc"
#include "heads_v7.inc"
#include "heads_v8.inc"

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 256, K = 256;
  const long N = argc > 2 ? atol(argv[2]) : 147456;
  srand(1);
  std::vector<double> hZ((size_t)M * K), hW((size_t)3 * N * K), hb(3 * N), hc(2 * N), hv((size_t)2 * M * N), hF((size_t)2 * M * N);
  for (auto& x : hZ) x = urand();
  for (auto& x : hW) x = 0.05 * urand();
  for (auto& x : hb) x = 0.1 * urand();
  for (auto& x : hc) x = 1.0 + 0.1 * urand();
  for (auto& x : hv) x = urand();
  for (auto& x : hF) x = urand();
  double *Z, *W, *b, *c, *v, *F, *v0, *vref, *ws;
  CK(hipMalloc(&Z, hZ.size() * 8)); CK(hipMalloc(&W, hW.size() * 8)); CK(hipMalloc(&b, hb.size() * 8));
  CK(hipMalloc(&c, hc.size() * 8)); CK(hipMalloc(&v, hv.size() * 8)); CK(hipMalloc(&F, hF.size() * 8));
  CK(hipMalloc(&v0, hv.size() * 8)); CK(hipMalloc(&vref, hv.size() * 8));
  const long ntile = cdiv(N, 64), mtile = cdiv(M, 64);
  const int ncols = (int)(ntile * 2);
  CK(hipMalloc(&ws, (size_t)M * ncols * 8 * 2));
  CK(hipMemcpy(Z, hZ.data(), hZ.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hW.data(), hW.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(b, hb.data(), hb.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(c, hc.data(), hc.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(v0, hv.data(), hv.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(F, hF.data(), hF.size() * 8, hipMemcpyHostToDevice));
  HeadsArgs a;
  a.Z = Z; a.W[0] = W; a.W[1] = W + N * K; a.W[2] = W + 2 * N * K;
  a.b[0] = b; a.b[1] = b + N; a.b[2] = b + 2 * N; a.cs = c; a.cq = c + N;
  a.ss = 1; a.st = 1; a.sq = 1; a.eps = 0.01; a.eps2 = (getenv("EQ_EPS") ? 0.01 : 0.012); a.fwd2 = 1; a.flip = 0;
  a.v = v; a.F = F; a.logdet_part = ws; a.M = M; a.N = (int)N; a.K = K; a.ncols_part = ncols;
  const double flop = 2.0 * 3 * M * (double)N * K;
  const dim3 grid((unsigned)(ntile * mtile)), block(kBlock);

  auto report = [&](const char* name, double ms) {
    printf("%-44s %8.4f ms  %6.2f TFLOP/s  frac %.3f\n", name, ms, flop / ms / 1e9, flop / ms / 1e9 / 78.6);
    fflush(stdout);
  };
  if (argc > 3) {   // PMC mode: one launch of each candidate
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL((fused_heads_vupdate_kernel<true, true, false, 0>), grid, block, 0, 0, a, 1, 0);
      hipLaunchKernelGGL((heads_v4_kernel<true, true, false, 0>), grid, block, 0, 0, a, 1);
      hipLaunchKernelGGL((heads_v4_kernel<true, true, false, 1>), grid, block, 0, 0, a, 1);
      hipLaunchKernelGGL((heads_v5_kernel<true, true, false, 0>), grid, block, 0, 0, a, 1);
      hipLaunchKernelGGL((heads_v5_kernel<true, true, false, 4>), grid, block, 0, 0, a, 1);
    }
    CK(hipDeviceSynchronize());
    return 0;
  }
  // reference result (PAIR = false and true)
  for (int pair = 0; pair < 2; ++pair) {
    CK(hipMemcpy(v, v0, hv.size() * 8, hipMemcpyDeviceToDevice));
    if (pair) hipLaunchKernelGGL((fused_heads_vupdate_kernel<true, true, true, 0>), grid, block, 0, 0, a, 1, 0);
    else hipLaunchKernelGGL((fused_heads_vupdate_kernel<true, true, false, 0>), grid, block, 0, 0, a, 1, 0);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(vref, v, hv.size() * 8, hipMemcpyDeviceToDevice));
    std::vector<double> href(hv.size());
    CK(hipMemcpy(href.data(), vref, hv.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> hld((size_t)M * ncols);
    CK(hipMemcpy(hld.data(), ws, hld.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ldref(M, 0.0);
    for (int m = 0; m < M; ++m) for (int j = 0; j < ncols; ++j) ldref[m] += hld[(size_t)m * ncols + j];
    v2_check(a, v0, hv.size(), href, ldref, pair, ws);
    v4_check(a, v0, hv.size(), href, ldref, pair, ws);
    v5_check(a, v0, hv.size(), href, ldref, pair, ws);
    v7_check(a, v0, hv.size(), href, ldref, pair, ws);
    v8_check(a, v0, hv.size(), href, ldref, pair, ws);
  }
  CK(hipMemcpy(v, v0, hv.size() * 8, hipMemcpyDeviceToDevice));
  report("base full (single)", timeit([&] { hipLaunchKernelGGL((fused_heads_vupdate_kernel<true, true, false, 0>), grid, block, 0, 0, a, 1, 0); }));
  report("base full (pair)", timeit([&] { hipLaunchKernelGGL((fused_heads_vupdate_kernel<true, true, true, 0>), grid, block, 0, 0, a, 1, 0); }));
  report("base K-loop only", timeit([&] { hipLaunchKernelGGL((fused_heads_vupdate_kernel<true, true, false, 1>), grid, block, 0, 0, a, 1, 0); }));
  report("base epilogue only (single)", timeit([&] { hipLaunchKernelGGL((fused_heads_vupdate_kernel<true, true, false, 2>), grid, block, 0, 0, a, 1, 0); }));
  report("base epilogue only (pair)", timeit([&] { hipLaunchKernelGGL((fused_heads_vupdate_kernel<true, true, true, 2>), grid, block, 0, 0, a, 1, 0); }));
  trace_run("base single", [&] { hipLaunchKernelGGL((fused_heads_vupdate_kernel<true, true, false, 3>), grid, block, 0, 0, a, 1, 0); }, grid.x);
  report("base full (single) again", timeit([&] { hipLaunchKernelGGL((fused_heads_vupdate_kernel<true, true, false, 0>), grid, block, 0, 0, a, 1, 0); }));
  v2_time(a, report);
  v4_time(a, report);
  v8_time(a, report);
  return 0;
}
