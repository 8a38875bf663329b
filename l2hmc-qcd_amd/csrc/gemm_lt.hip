// gemm_lt.hip -- the LARGE plain 16-bit layers of the reference's autocast region (nn.Linear with both dimensions in
// the thousands: the Linear that closes the U(1) conv stack, 8192 x 8192 x 51 200 at cfg-3;
// network/pytorch/network.py:323-326 under trainers/pytorch/trainer.py:211-219) through hipBLASLt.
//
// This is the one place where the build uses a vendor GEMM: a plain C = A W^T + bias with nothing fused in front
// of it.  The library's kernel for this shape runs at 1.56 PFLOP/s on MI355X, this build's own LDS-DMA 256 x 256
// kernel (gemm_f16_dma.hip) at 1.25 (same box, profiles/r06_gemm_h_lt_ab.txt); everything that has a fused loader or
// epilogue (masked cos / sin inputs, fp32 operands rounded on load, heads + update, conv + pool) stays on the
// hand-written kernels.  The rounding points are autocast's: D = r16(A W^T + bias) in fp32 accumulation (the
// library's bias epilogue), then r16(act(.)) by a small kernel here (also the widening copy when the caller wants
// the 16-bit values in an fp32 container).
//
// libhipblaslt.so is resolved at run time (dlopen, like librccl in comm.hip): the product has no link-time
// dependency on it, and where it is missing -- or refuses the shape -- l2q_gemm_h simply continues to its own
// kernels.  Handle, descriptors and the heuristic's algorithm are cached per (device, type, shape); the call
// allocates nothing per launch: the library's workspace and the 16-bit staging of D are carved out of the
// caller's `ws` (l2q_gemm_h_ws_bytes accounts for them).
#include "l2q_common.hpp"
#include "half_common.hpp"

#include <dlfcn.h>
#include <hipblaslt/hipblaslt.h>

#include <map>
#include <mutex>
#include <tuple>

namespace l2q {

constexpr size_t kLtWorkspace = 64ul << 20;

namespace {

struct LtApi {
  void* so = nullptr;
  bool tried = false, ok = false;
  decltype(&hipblasLtCreate) create = nullptr;
  decltype(&hipblasLtMatmulDescCreate) desc_create = nullptr;
  decltype(&hipblasLtMatmulDescSetAttribute) desc_set = nullptr;
  decltype(&hipblasLtMatrixLayoutCreate) layout_create = nullptr;
  decltype(&hipblasLtMatmulPreferenceCreate) pref_create = nullptr;
  decltype(&hipblasLtMatmulPreferenceSetAttribute) pref_set = nullptr;
  decltype(&hipblasLtMatmulPreferenceDestroy) pref_destroy = nullptr;
  decltype(&hipblasLtMatmulAlgoGetHeuristic) heuristic = nullptr;
  decltype(&hipblasLtMatmul) matmul = nullptr;
};

LtApi& lt_api() {
  static LtApi api;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  if (api.tried) return api;
  api.tried = true;
  for (const char* name : {"libhipblaslt.so.1", "libhipblaslt.so", "/opt/rocm/lib/libhipblaslt.so"}) {
    api.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (api.so) break;
  }
  if (!api.so) return api;
#define L2Q_LT_SYM(field, sym) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.so, #sym))
  L2Q_LT_SYM(create, hipblasLtCreate);
  L2Q_LT_SYM(desc_create, hipblasLtMatmulDescCreate);
  L2Q_LT_SYM(desc_set, hipblasLtMatmulDescSetAttribute);
  L2Q_LT_SYM(layout_create, hipblasLtMatrixLayoutCreate);
  L2Q_LT_SYM(pref_create, hipblasLtMatmulPreferenceCreate);
  L2Q_LT_SYM(pref_set, hipblasLtMatmulPreferenceSetAttribute);
  L2Q_LT_SYM(pref_destroy, hipblasLtMatmulPreferenceDestroy);
  L2Q_LT_SYM(heuristic, hipblasLtMatmulAlgoGetHeuristic);
  L2Q_LT_SYM(matmul, hipblasLtMatmul);
#undef L2Q_LT_SYM
  api.ok = api.create && api.desc_create && api.desc_set && api.layout_create && api.pref_create && api.pref_set &&
           api.pref_destroy && api.heuristic && api.matmul;
  return api;
}

struct LtPlan {
  bool usable = false;
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, ld = nullptr;
  hipblasLtMatmulAlgo_t algo;
};

struct LtDevice {
  hipblasLtHandle_t handle = nullptr;
  std::map<std::tuple<int, int, int, long, int>, LtPlan> plans;      // (half type, M, N, K, bias?) -> plan
};

std::mutex g_lt_mu;
std::map<int, LtDevice> g_lt_dev;

// r16(act(d)) over M * N values, to a 16-bit or an fp32 container
template <typename HT, bool OUT32>
__global__ __launch_bounds__(kBlock) void lt_act_kernel(const HT* __restrict__ d, void* __restrict__ out, long n,
                                                        int act) {
  const long i = ((long)blockIdx.x * kBlock + threadIdx.x) * 8;
  if (i >= n) return;
  union { uint4 u; HT h[8]; } in;
  if (i + 8 <= n) {
    in.u = *reinterpret_cast<const uint4*>(d + i);
  } else {
    for (int k = 0; k < 8; ++k) in.h[k] = i + k < n ? d[i + k] : (HT)0.f;
  }
  float y[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    y[k] = (float)in.h[k];
    if (act != L2Q_ACT_NONE) y[k] = rnd<HT>(act_h(y[k], act));
  }
  if (OUT32) {
    float* o = reinterpret_cast<float*>(out) + i;
    if (i + 8 <= n) {
      *reinterpret_cast<float4*>(o) = make_float4(y[0], y[1], y[2], y[3]);
      *reinterpret_cast<float4*>(o + 4) = make_float4(y[4], y[5], y[6], y[7]);
    } else {
      for (int k = 0; k < 8 && i + k < n; ++k) o[k] = y[k];
    }
  } else {
    HT* o = reinterpret_cast<HT*>(out) + i;
    union { uint4 u; HT h[8]; } r;
#pragma unroll
    for (int k = 0; k < 8; ++k) r.h[k] = (HT)y[k];
    if (i + 8 <= n) {
      *reinterpret_cast<uint4*>(o) = r.u;
    } else {
      for (int k = 0; k < 8 && i + k < n; ++k) o[k] = r.h[k];
    }
  }
}

}  // namespace

// the shapes this route takes: both output dimensions in the thousands, a deep K, nothing fused
bool gemm_h_lt_shape(int M, int N, long K) {
  return tuning().gemm_h_lt != 0 && M >= 2048 && N >= 2048 && K >= 2048 && K < (1L << 31) && M % 8 == 0 &&
         N % 8 == 0 && K % 8 == 0;
}

bool gemm_h_lt_available() { return lt_api().ok; }

size_t gemm_h_lt_ws_bytes(int M, int N, long K) {
  if (!gemm_h_lt_shape(M, N, K)) return 0;
  return kLtWorkspace + (size_t)M * N * 2 + 256;
}

// true: the layer has been enqueued on `st`.  false: not taken (library missing, shape refused): the caller goes on.
template <typename HT>
bool gemm_h_lt_launch(const void* A, const void* W, int M, int N, long K, const EpiH& epi, void* C, int c_is_f32,
                      void* ws, size_t ws_bytes, hipStream_t st) {
  constexpr int half_type = std::is_same<HT, _Float16>::value ? L2Q_HALF_F16 : L2Q_HALF_BF16;
  if (!gemm_h_lt_shape(M, N, K) || epi.bias2 || epi.coeff || epi.scale != 1.0f) return false;
  if (!ws || ws_bytes < gemm_h_lt_ws_bytes(M, N, K)) return false;
  if (!al16(A) || !al16(W) || !al16(C)) return false;
  LtApi& api = lt_api();
  if (!api.ok) return false;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::lock_guard<std::mutex> lock(g_lt_mu);
  LtDevice& D = g_lt_dev[dev];
  if (!D.handle && api.create(&D.handle) != HIPBLAS_STATUS_SUCCESS) { D.handle = nullptr; return false; }
  const auto key = std::make_tuple(half_type, M, N, K, epi.bias ? 1 : 0);
  auto it = D.plans.find(key);
  if (it == D.plans.end()) {
    LtPlan p;
    const hipDataType ht = half_type == L2Q_HALF_F16 ? HIP_R_16F : HIP_R_16BF;
    // row-major C[M][N] = A[M][K] W[N][K]^T  ==  column-major D (N x M) = op_T(Wc: K x N) . (Ac: K x M)
    bool ok = api.desc_create(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) == HIPBLAS_STATUS_SUCCESS;
    const hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
    ok = ok && api.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)) == HIPBLAS_STATUS_SUCCESS;
    ok = ok && api.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)) == HIPBLAS_STATUS_SUCCESS;
    if (ok && epi.bias) {
      const hipblasLtEpilogue_t ep = HIPBLASLT_EPILOGUE_BIAS;
      const hipDataType bt = HIP_R_32F;
      ok = api.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof(ep)) == HIPBLAS_STATUS_SUCCESS &&
           api.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)) == HIPBLAS_STATUS_SUCCESS;
    }
    ok = ok && api.layout_create(&p.la, ht, (uint64_t)K, (uint64_t)N, (int64_t)K) == HIPBLAS_STATUS_SUCCESS;
    ok = ok && api.layout_create(&p.lb, ht, (uint64_t)K, (uint64_t)M, (int64_t)K) == HIPBLAS_STATUS_SUCCESS;
    ok = ok && api.layout_create(&p.ld, ht, (uint64_t)N, (uint64_t)M, (int64_t)N) == HIPBLAS_STATUS_SUCCESS;
    if (ok) {
      hipblasLtMatmulPreference_t pref = nullptr;
      ok = api.pref_create(&pref) == HIPBLAS_STATUS_SUCCESS;
      const uint64_t wsz = kLtWorkspace;
      ok = ok && api.pref_set(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsz, sizeof(wsz)) ==
                     HIPBLAS_STATUS_SUCCESS;
      if (ok) {
        // (the bias pointer must be set for the heuristic of a bias epilogue; it is set again per call)
        if (epi.bias) {
          const void* bp = epi.bias;
          (void)api.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bp, sizeof(bp));
        }
        hipblasLtMatmulHeuristicResult_t res;
        int found = 0;
        ok = api.heuristic(D.handle, p.desc, p.la, p.lb, p.ld, p.ld, pref, 1, &res, &found) ==
                 HIPBLAS_STATUS_SUCCESS && found > 0 && res.workspaceSize <= kLtWorkspace;
        if (ok) p.algo = res.algo;
      }
      if (pref) (void)api.pref_destroy(pref);
    }
    p.usable = ok;
    it = D.plans.emplace(key, p).first;
  }
  LtPlan& p = it->second;
  if (!p.usable) return false;
  char* wsb = reinterpret_cast<char*>(ws);
  void* lt_ws = wsb;
  HT* d16 = reinterpret_cast<HT*>(wsb + kLtWorkspace);
  const bool direct = !c_is_f32 && epi.act == L2Q_ACT_NONE;            // the library's output IS the result
  void* dout = direct ? C : (void*)d16;
  if (epi.bias) {
    const void* bp = epi.bias;
    if (api.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bp, sizeof(bp)) != HIPBLAS_STATUS_SUCCESS)
      return false;
  }
  const float alpha = 1.0f, beta = 0.0f;
  if (api.matmul(D.handle, p.desc, &alpha, W, p.la, A, p.lb, &beta, dout, p.ld, dout, p.ld, &p.algo, lt_ws,
                 kLtWorkspace, st) != HIPBLAS_STATUS_SUCCESS)
    return false;
  if (!direct) {
    const long n = (long)M * N;
    const dim3 grid((unsigned)cdiv(cdiv(n, 8), kBlock)), block(kBlock);
    if (c_is_f32) hipLaunchKernelGGL((lt_act_kernel<HT, true>), grid, block, 0, st, d16, C, n, epi.act);
    else hipLaunchKernelGGL((lt_act_kernel<HT, false>), grid, block, 0, st, d16, C, n, epi.act);
  }
  return true;
}

template bool gemm_h_lt_launch<_Float16>(const void*, const void*, int, int, long, const EpiH&, void*, int, void*,
                                         size_t, hipStream_t);
template bool gemm_h_lt_launch<__bf16>(const void*, const void*, int, int, long, const EpiH&, void*, int, void*,
                                       size_t, hipStream_t);

}  // namespace l2q
