// common.hip -- error text, layout conversion, element-wise L2HMC updates, accept/select.
#include <cstdarg>

#include "l2q_common.hpp"

namespace l2q {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Performance knobs are per DEVICE (the only state the library keeps besides the last-error
// text): one process per GPU is the deployment model, but a process that drives several devices
// gets an independent table for each -- `tuning()` resolves to the table of the current HIP
// device of the calling thread.  Results never depend on the knobs.
Tuning& tuning() {
  static Tuning t[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
  return t[dev];
}

// out[c][k] = scale * sum_b partial[c][b][k] + offset   (fixed order)
__global__ void finalize_sum_kernel(const double* __restrict__ partial, double* __restrict__ out,
                                    long nblk, int ncomp, double scale, double offset) {
  __shared__ double lds[4];
  const long c = blockIdx.x;
  for (int k = 0; k < ncomp; ++k) {
    double s = 0.0;
    for (long b = threadIdx.x; b < nblk; b += blockDim.x) s += partial[(c * nblk + b) * ncomp + k];
    const double t = block_sum(s, lds);
    if (threadIdx.x == 0) out[c * ncomp + k] = scale * t + offset;
  }
}

void launch_finalize(const double* partial, double* out, int nb, long nblk, int ncomp,
                     double scale, double offset, hipStream_t st) {
  hipLaunchKernelGGL(finalize_sum_kernel, dim3(nb), dim3(kBlock), 0, st, partial, out, nblk, ncomp,
                     scale, offset);
}

__global__ void zero_words_kernel(unsigned* __restrict__ p, size_t nwords) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nwords) p[i] = 0u;
}

void launch_zero(void* p, size_t bytes, hipStream_t st) {
  const size_t nwords = (bytes + 3) / 4;
  if (nwords == 0) return;
  hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)cdiv((long)nwords, kBlock)), dim3(kBlock), 0, st,
                     (unsigned*)p, nwords);
}

// ------------------------------------------------------------------ batched transpose
// in[batch][rows][cols] -> out[batch][cols][rows].  cols or rows is small (8/9); a block
// stages a tile of TILE consecutive "long-axis" items through LDS so that both the global
// read and the global write are contiguous runs.
template <typename T>
__global__ __launch_bounds__(kBlock) void transpose_kernel(const T* __restrict__ in,
                                                           T* __restrict__ out, int rows, int cols,
                                                           long tiles_r, long tiles_c) {
  // tile: 32 x 32 with +1 padding
  __shared__ T tile[32][33];
  long b = blockIdx.x;
  const long tc = b % tiles_c; b /= tiles_c;
  const long tr = b % tiles_r; b /= tiles_r;
  const T* src = in + b * (long)rows * cols;
  T* dst = out + b * (long)rows * cols;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long r = tr * 32 + ty + 8 * k, c = tc * 32 + tx;
    if (r < rows && c < cols) tile[ty + 8 * k][tx] = src[r * cols + c];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long c = tc * 32 + ty + 8 * k, r = tr * 32 + tx;
    if (r < rows && c < cols) dst[c * rows + r] = tile[tx][ty + 8 * k];
  }
}

// The accept / reject select of a transition fused into its native -> reference transpose: batch item b
// (a link direction of chain b / per_chain) is read from `a` where mask[chain] != 0 and from `b` otherwise.
// One pass over the field instead of select_rows (read 2, write 1) + transpose (read 1, write 1).
template <typename T>
__global__ __launch_bounds__(kBlock) void transpose_select_kernel(const T* __restrict__ a,
                                                                  const T* __restrict__ bsrc,
                                                                  const float* __restrict__ mask,
                                                                  int per_chain, T* __restrict__ out,
                                                                  int rows, int cols, long tiles_r,
                                                                  long tiles_c) {
  __shared__ T tile[32][33];
  long b = blockIdx.x;
  const long tc = b % tiles_c; b /= tiles_c;
  const long tr = b % tiles_r; b /= tiles_r;
  const T* src = (mask[b / per_chain] != 0.0f ? a : bsrc) + b * (long)rows * cols;
  T* dst = out + b * (long)rows * cols;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long r = tr * 32 + ty + 8 * k, c = tc * 32 + tx;
    if (r < rows && c < cols) tile[ty + 8 * k][tx] = src[r * cols + c];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long c = tc * 32 + ty + 8 * k, r = tr * 32 + tx;
    if (r < rows && c < cols) dst[c * rows + r] = tile[tx][ty + 8 * k];
  }
}

template <typename T>
static int launch_transpose(const void* in, void* out, long batch, int rows, int cols,
                            hipStream_t st) {
  const long tiles_r = cdiv(rows, 32), tiles_c = cdiv(cols, 32);
  const long nblocks = batch * tiles_r * tiles_c;
  if (nblocks > 0x7fffffffL) { set_error("l2q_transpose: grid too large"); return L2Q_ESHAPE; }
  hipLaunchKernelGGL(transpose_kernel<T>, dim3((unsigned)nblocks), dim3(kBlock), 0, st,
                     (const T*)in, (T*)out, rows, cols, tiles_r, tiles_c);
  return check_launch("l2q_transpose");
}

// ------------------------------------------------------------------ v update
// T = double/float; CPLX: v, F are (re, im) pairs, heads are real.
template <typename T, bool CPLX, bool FWD>
__global__ __launch_bounds__(kBlock) void v_update_kernel(const T* vin, T* v, const T* __restrict__ force,
                                                          const T* __restrict__ s,
                                                          const T* __restrict__ t,
                                                          const T* __restrict__ q, T eps, long n,
                                                          long nblk, double* __restrict__ partial) {
  __shared__ double lds[4];
  const long c = blockIdx.x / nblk, blk = blockIdx.x % nblk;
  const long j = blk * kBlock + threadIdx.x;
  double ld = 0.0;
  if (j < n) {
    const long o = c * n + j;
    const T half = (T)0.5;
    const T sj = s[o], tj = t[o], qj = q[o];
    const T lj = FWD ? (eps * sj * half) : (-eps * sj * half);
    ld = (double)lj;
    const T es = exp(lj);
    const T eq = exp(eps * qj);
    if (CPLX) {
      const T vr = vin[2 * o], vi = vin[2 * o + 1];
      const T fr = force[2 * o] * eq + tj, fi = force[2 * o + 1] * eq;
      if (FWD) {
        v[2 * o] = es * vr - half * eps * fr;
        v[2 * o + 1] = es * vi - half * eps * fi;
      } else {
        v[2 * o] = es * (vr + half * eps * fr);
        v[2 * o + 1] = es * (vi + half * eps * fi);
      }
    } else {
      const T f = force[o] * eq + tj;
      v[o] = FWD ? (es * vin[o] - half * eps * f) : (es * (vin[o] + half * eps * f));
    }
  }
  const double r = block_sum(ld, lds);
  if (threadIdx.x == 0) partial[c * nblk + blk] = r;
}

template <typename T>
__global__ void cast_out_kernel(const double* __restrict__ in, T* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (T)in[i];
}

template <typename T>
__global__ void accept_kernel(const T* h0, const T* h1, const T* sld, const T* u, T* acc,
                              float* mask, int nb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nb) return;
  const T dh = h0[i] - h1[i] + sld[i];
  // torch.minimum propagates NaN (dynamics.py:1065-1079): a diverged trajectory (inf - inf in
  // H) must show up as acc = NaN in the metrics and be rejected ((NaN > u) is false), not be
  // reported as a perfectly accepted proposal
  const T a = (dh != dh) ? dh : exp(dh < (T)0 ? dh : (T)0);
  acc[i] = a;
  mask[i] = (a > u[i]) ? 1.0f : 0.0f;
}

template <typename W>
__global__ __launch_bounds__(kBlock) void select_rows_kernel(const W* __restrict__ a,
                                                             const W* __restrict__ b,
                                                             const float* __restrict__ mask,
                                                             W* __restrict__ out, long roww,
                                                             long nblk) {
  const long c = blockIdx.x / nblk, blk = blockIdx.x % nblk;
  const long j = blk * kBlock + threadIdx.x;
  if (j >= roww) return;
  const bool acc = mask[c] != 0.0f;
  out[c * roww + j] = acc ? a[c * roww + j] : b[c * roww + j];
}

__global__ void scale_kernel(const double* x, double alpha, double* y, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = alpha * x[i];
}

template <typename T>
__global__ void axpy_kernel(const T* __restrict__ p, T alpha, T* x, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = x[i] + alpha * p[i];
}

}  // namespace l2q

using namespace l2q;

extern "C" {

const char* l2q_last_error(void) { return g_err; }
int l2q_version(void) { return 100; }

int l2q_set_tuning(const char* key, int value) {
  L2Q_REQUIRE(key, L2Q_EINVAL, "null key");
  Tuning& t = tuning();
  int* slot = nullptr;
  bool ok = false;
  if (!strcmp(key, "plaq_occ")) { slot = &t.plaq_occ; ok = value >= 2 && value <= 4; }
  else if (!strcmp(key, "force_occ")) { slot = &t.force_occ; ok = value >= 2 && value <= 4; }
  else if (!strcmp(key, "xcd_swizzle")) { slot = &t.xcd_swizzle; ok = value == 0 || value == 1; }
  else if (!strcmp(key, "plaq_sweep")) { slot = &t.plaq_sweep; ok = value >= 0 && value <= 3; }
  else if (!strcmp(key, "force_tile")) { slot = &t.force_tile; ok = value >= 0 && value <= 7; }
  else if (!strcmp(key, "gemm_h_wide_fused")) { slot = &t.gemm_h_wide_fused; ok = value == 0 || value == 1; }
  else if (!strcmp(key, "conv_stream")) { slot = &t.conv_stream; ok = value == 0 || value == 1; }
  else if (!strcmp(key, "conv_patch")) { slot = &t.conv_patch; ok = value >= 0 && value <= 2; }
  else if (!strcmp(key, "gemm_h_dma")) { slot = &t.gemm_h_dma; ok = value == 0 || value == 1; }
  else if (!strcmp(key, "gemm_h_lt")) { slot = &t.gemm_h_lt; ok = value == 0 || value == 1; }
  else if (!strcmp(key, "gemm_h_skinny")) { slot = &t.gemm_h_skinny; ok = value == 0 || value == 1 || value == 2 || value == 4 || value == 8; }
  else if (!strcmp(key, "gemm_h_small")) { slot = &t.gemm_h_small; ok = value == 0 || value == 1; }
  else if (!strcmp(key, "gemm_h_patch")) { slot = &t.gemm_h_patch; ok = value == 0 || value == 1; }
  else if (!strcmp(key, "heads_h_bm")) { slot = &t.heads_h_bm; ok = value == 64 || value == 128; }
  else if (!strcmp(key, "heads_h_stream")) { slot = &t.heads_h_stream; ok = value >= 0 && value <= 3; }
  else if (!strcmp(key, "heads_h_order")) { slot = &t.heads_h_order; ok = value == 0 || value == 1; }
  else if (!strcmp(key, "u1_fused_ch")) { slot = &t.u1_fused_ch; ok = value == 0 || value == 1 || value == 2 || value == 4 || value == 8; }
  else if (!strcmp(key, "heads_dma")) { slot = &t.heads_dma; ok = value == 0 || value == 1; }
  else if (!strcmp(key, "force_tsplit")) { slot = &t.force_tsplit; ok = value >= 0 && value <= 64; }
  else if (!strcmp(key, "force_stagger")) { slot = &t.force_stagger; ok = value >= 0 && value <= 64; }
  else if (!strcmp(key, "heads_stagger")) { slot = &t.heads_stagger; ok = value >= 0 && value <= 64; }
  if (!slot || !ok) { set_error("l2q_set_tuning: bad key/value %s=%d", key, value); return L2Q_EINVAL; }
  const int prev = *slot;
  *slot = value;
  return prev;
}

int l2q_transpose(const void* in, void* out, long batch, int rows, int cols, int elem_bytes,
                  void* stream) {
  L2Q_REQUIRE(in && out, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(batch > 0 && rows > 0 && cols > 0, L2Q_EINVAL, "non-positive size");
  L2Q_REQUIRE(in != out, L2Q_EINVAL, "in-place transpose not supported");
  hipStream_t st = (hipStream_t)stream;
  switch (elem_bytes) {
    case 4: return launch_transpose<float>(in, out, batch, rows, cols, st);
    case 8: return launch_transpose<double>(in, out, batch, rows, cols, st);
    case 16: return launch_transpose<double2>(in, out, batch, rows, cols, st);
    default: set_error("l2q_transpose: elem_bytes must be 4, 8 or 16"); return L2Q_EINVAL;
  }
}

int l2q_su3_pack(const void* x_ref, void* x_nat, int nb, long V, void* stream) {
  L2Q_REQUIRE(V > 0 && V <= 0x7fffffffL, L2Q_EINVAL, "bad volume");
  return l2q_transpose(x_ref, x_nat, (long)nb * 4, (int)V, 9, 16, stream);
}

int l2q_su3_unpack(const void* x_nat, void* x_ref, int nb, long V, void* stream) {
  L2Q_REQUIRE(V > 0 && V <= 0x7fffffffL, L2Q_EINVAL, "bad volume");
  return l2q_transpose(x_nat, x_ref, (long)nb * 4, 9, (int)V, 16, stream);
}

int l2q_su3_unpack_select(const void* a_nat, const void* b_nat, const float* mask, void* x_ref, int nb,
                          long V, void* stream) {
  L2Q_REQUIRE(a_nat && b_nat && mask && x_ref, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && V > 0 && V <= 0x7fffffffL, L2Q_EINVAL, "bad size");
  L2Q_REQUIRE(x_ref != a_nat && x_ref != b_nat, L2Q_EINVAL, "in-place transpose not supported");
  const long tiles_r = 1, tiles_c = cdiv(V, 32);
  const long nblocks = (long)nb * 4 * tiles_r * tiles_c;
  if (nblocks > 0x7fffffffL) { set_error("l2q_su3_unpack_select: grid too large"); return L2Q_ESHAPE; }
  hipLaunchKernelGGL(transpose_select_kernel<double2>, dim3((unsigned)nblocks), dim3(kBlock), 0,
                     (hipStream_t)stream, (const double2*)a_nat, (const double2*)b_nat, mask, 4,
                     (double2*)x_ref, 9, (int)V, tiles_r, tiles_c);
  return check_launch("l2q_su3_unpack_select");
}

static int v_update_launch(const void* vin, void* v, const void* force, const void* s, const void* t,
                           const void* q, double eps, int forward, int is_complex, int elem_bytes,
                           int nb, long n, void* logdet, void* ws, size_t ws_bytes, void* stream) {
  L2Q_REQUIRE(vin && v && force && s && t && q && logdet && ws, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && n > 0, L2Q_EINVAL, "non-positive size");
  L2Q_REQUIRE(elem_bytes == 8 || elem_bytes == 4, L2Q_EINVAL, "elem_bytes must be 4 or 8");
  const long nblk = cdiv(n, kBlock);
  L2Q_REQUIRE(ws_bytes >= ((size_t)nb * nblk + nb) * sizeof(double), L2Q_ESHAPE,
              "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  double* partial = (double*)ws;
  const dim3 grid((unsigned)(nb * nblk)), block(kBlock);
#define L2Q_VUPD(T, C, F)                                                                   \
  hipLaunchKernelGGL((v_update_kernel<T, C, F>), grid, block, 0, st, (const T*)vin, (T*)v, (const T*)force, \
                     (const T*)s, (const T*)t, (const T*)q, (T)eps, n, nblk, partial)
  if (elem_bytes == 8) {
    if (is_complex) { if (forward) L2Q_VUPD(double, true, true); else L2Q_VUPD(double, true, false); }
    else { if (forward) L2Q_VUPD(double, false, true); else L2Q_VUPD(double, false, false); }
    launch_finalize(partial, (double*)logdet, nb, nblk, 1, 1.0, 0.0, st);
  } else {
    if (is_complex) { if (forward) L2Q_VUPD(float, true, true); else L2Q_VUPD(float, true, false); }
    else { if (forward) L2Q_VUPD(float, false, true); else L2Q_VUPD(float, false, false); }
    double* tmp = partial + (size_t)nb * nblk;
    launch_finalize(partial, tmp, nb, nblk, 1, 1.0, 0.0, st);
    hipLaunchKernelGGL(cast_out_kernel<float>, dim3(cdiv(nb, 64)), dim3(64), 0, st, tmp,
                       (float*)logdet, nb);
  }
#undef L2Q_VUPD
  return check_launch("l2q_v_update");
}

int l2q_v_update(void* v, const void* force, const void* s, const void* t, const void* q,
                 double eps, int forward, int is_complex, int elem_bytes, int nb, long n,
                 void* logdet, void* ws, size_t ws_bytes, void* stream) {
  return v_update_launch(v, v, force, s, t, q, eps, forward, is_complex, elem_bytes, nb, n, logdet,
                         ws, ws_bytes, stream);
}

int l2q_v_update_to(const void* v_in, void* v_out, const void* force, const void* s, const void* t,
                    const void* q, double eps, int forward, int is_complex, int elem_bytes, int nb,
                    long n, void* logdet, void* ws, size_t ws_bytes, void* stream) {
  return v_update_launch(v_in, v_out, force, s, t, q, eps, forward, is_complex, elem_bytes, nb, n,
                         logdet, ws, ws_bytes, stream);
}

int l2q_accept(const void* h_init, const void* h_prop, const void* sumlogdet, const void* u,
               void* acc, float* mask, int nb, int elem_bytes, void* stream) {
  L2Q_REQUIRE(h_init && h_prop && sumlogdet && u && acc && mask, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0, L2Q_EINVAL, "non-positive size");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)cdiv(nb, 64)), block(64);
  if (elem_bytes == 8) {
    hipLaunchKernelGGL(accept_kernel<double>, grid, block, 0, st, (const double*)h_init,
                       (const double*)h_prop, (const double*)sumlogdet, (const double*)u,
                       (double*)acc, mask, nb);
  } else if (elem_bytes == 4) {
    hipLaunchKernelGGL(accept_kernel<float>, grid, block, 0, st, (const float*)h_init,
                       (const float*)h_prop, (const float*)sumlogdet, (const float*)u, (float*)acc,
                       mask, nb);
  } else {
    set_error("l2q_accept: elem_bytes must be 4 or 8");
    return L2Q_EINVAL;
  }
  return check_launch("l2q_accept");
}

int l2q_select_rows(const void* a, const void* b, const float* mask, void* out, int nb,
                    long row_bytes, void* stream) {
  L2Q_REQUIRE(a && b && mask && out, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && row_bytes > 0, L2Q_EINVAL, "non-positive size");
  L2Q_REQUIRE(row_bytes % 4 == 0, L2Q_ESHAPE, "row_bytes must be a multiple of 4");
  hipStream_t st = (hipStream_t)stream;
#define L2Q_SEL(W)                                                                          \
  do {                                                                                      \
    const long roww = row_bytes / (long)sizeof(W);                                          \
    const long nblk = cdiv(roww, kBlock);                                                   \
    hipLaunchKernelGGL(select_rows_kernel<W>, dim3((unsigned)(nb * nblk)), dim3(kBlock), 0, \
                       st, (const W*)a, (const W*)b, mask, (W*)out, roww, nblk);            \
  } while (0)
  if (row_bytes % 16 == 0) L2Q_SEL(uint4);
  else if (row_bytes % 8 == 0) L2Q_SEL(uint2);
  else L2Q_SEL(unsigned);
#undef L2Q_SEL
  return check_launch("l2q_select_rows");
}

int l2q_scale_f64(const double* x, double alpha, double* y, long n, void* stream) {
  L2Q_REQUIRE(x && y, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(n > 0, L2Q_EINVAL, "non-positive size");
  hipLaunchKernelGGL(scale_kernel, dim3((unsigned)cdiv(n, kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, x, alpha, y, n);
  return check_launch("l2q_scale_f64");
}

int l2q_axpy(const void* p, double alpha, void* x, long n, int elem_bytes, void* stream) {
  L2Q_REQUIRE(p && x, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(n > 0, L2Q_EINVAL, "non-positive size");
  const dim3 grid((unsigned)cdiv(n, kBlock)), block(kBlock);
  if (elem_bytes == 8) {
    hipLaunchKernelGGL(axpy_kernel<double>, grid, block, 0, (hipStream_t)stream, (const double*)p,
                       alpha, (double*)x, n);
  } else if (elem_bytes == 4) {
    hipLaunchKernelGGL(axpy_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)p,
                       (float)alpha, (float*)x, n);
  } else {
    set_error("l2q_axpy: elem_bytes must be 4 or 8");
    return L2Q_EINVAL;
  }
  return check_launch("l2q_axpy");
}

}  // extern "C"
