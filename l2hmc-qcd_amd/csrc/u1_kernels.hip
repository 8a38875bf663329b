// u1_kernels.hip -- 2D U(1) lattice kernels (angles) for gfx950.
//
// Fields keep the reference layout x[chain][2][T][X] (site-contiguous already).  A 2D lattice
// of one chain is tiny (8x8 ... 64x64), so one workgroup owns one chain: the whole lattice
// is read once, per-chain reductions finish inside the block in a fixed order (no atomics,
// no second pass) and 2048-8192 chains fill the 256 CUs.
#include "l2q_common.hpp"
#include "u1_math.hpp"

namespace l2q {

template <typename T>
__global__ __launch_bounds__(kBlock) void u1_plaq_kernel(const T* __restrict__ x, int Tn, int Xn,
                                                         T* __restrict__ out) {
  __shared__ double lds[4];
  const int c = blockIdx.x, V = Tn * Xn;
  const T* xc = x + (long)c * 2 * V;
  const T pi = (T)3.14159265358979323846, two_pi = (T)6.28318530717958647692;
  double sc = 0.0, ss = 0.0, sp = 0.0;
  for (int s = threadIdx.x; s < V; s += kBlock) {
    const T th = plaq_angle(xc, s / Xn, s % Xn, Tn, Xn);
    sc += (double)Math<T>::cos(th);
    ss += (double)Math<T>::sin(th);
    sp += (double)(th - two_pi * Math<T>::floor((th + pi) / two_pi));   // project_angle
  }
  const double a = block_sum(sc, lds);
  const double b = block_sum(ss, lds);
  const double p = block_sum(sp, lds);
  if (threadIdx.x == 0) { out[c * 3 + 0] = (T)a; out[c * 3 + 1] = (T)b; out[c * 3 + 2] = (T)p; }
}

// The plaquette-angle FIELD theta[chain][t][x] itself: the tensor the reference's `LatticeU1.wilson_loops`
// returns (lattice/u1/pytorch/lattice.py:154-159; same left-to-right sum, so fp32 results are the same bits).
template <typename T>
__global__ __launch_bounds__(kBlock) void u1_wloops_kernel(const T* __restrict__ x, int Tn, int Xn,
                                                           long total, T* __restrict__ out) {
  const long i = (long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= total) return;
  const int V = Tn * Xn;
  const long c = i / V;
  const int s = (int)(i % V);
  out[i] = plaq_angle(x + c * 2 * V, s / Xn, s % Xn, Tn, Xn);
}

// its adjoint (theta is linear in the links): dx0(t, x) += g(t, x) - g(t, x-1), dx1(t, x) += g(t-1, x) - g(t, x)
template <typename T>
__global__ __launch_bounds__(kBlock) void u1_wloops_bwd_kernel(const T* __restrict__ g, int Tn, int Xn,
                                                               long total, T* dx) {
  const long i = (long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= total) return;
  const int V = Tn * Xn;
  const long c = i / V;
  const int s = (int)(i % V);
  const int t = s / Xn, xx = s % Xn;
  const int tm = t == 0 ? Tn - 1 : t - 1, xm = xx == 0 ? Xn - 1 : xx - 1;
  const T* gc = g + c * V;
  T* dc = dx + c * 2 * V;
  dc[s] += gc[s] - gc[t * Xn + xm];
  dc[V + s] += gc[tm * Xn + xx] - gc[s];
}

// force written and/or fused kick v += coef * F.  One workgroup per chain: sin(theta) of every
// plaquette is computed ONCE into LDS (each one enters 4 force components), then the force
// is two differences of LDS values.  Lattices beyond kU1MaxLds sites recompute instead.
constexpr int kU1MaxLds = 8192;

template <typename T>
__global__ __launch_bounds__(kBlock) void u1_force_kernel(const T* __restrict__ x, T beta,
                                                          T* __restrict__ force, T* v, T coef,
                                                          int Tn, int Xn) {
  __shared__ T sth[kU1MaxLds];
  const int c = blockIdx.x, V = Tn * Xn;
  const T* xc = x + (long)c * 2 * V;
  const bool cached = V <= kU1MaxLds;
  if (cached) {
    for (int s = threadIdx.x; s < V; s += kBlock)
      sth[s] = Math<T>::sin(plaq_angle(xc, s / Xn, s % Xn, Tn, Xn));
    __syncthreads();
  }
  for (int s = threadIdx.x; s < V; s += kBlock) {
    const int t = s / Xn, xx = s % Xn;
    const int tm = (t == 0) ? Tn - 1 : t - 1;
    const int xm = (xx == 0) ? Xn - 1 : xx - 1;
    T s0, sxm, stm;
    if (cached) {
      s0 = sth[s]; sxm = sth[t * Xn + xm]; stm = sth[tm * Xn + xx];
    } else {
      s0 = Math<T>::sin(plaq_angle(xc, t, xx, Tn, Xn));
      sxm = Math<T>::sin(plaq_angle(xc, t, xm, Tn, Xn));
      stm = Math<T>::sin(plaq_angle(xc, tm, xx, Tn, Xn));
    }
    const T f0 = beta * (s0 - sxm);
    const T f1 = beta * (-s0 + stm);
    const long o = (long)c * 2 * V;
    if (force) { force[o + s] = f0; force[o + V + s] = f1; }
    if (v) { v[o + s] += coef * f0; v[o + V + s] += coef * f1; }
  }
}

// The same force for fp32 lattices of 1024 .. 4096 sites with X % 4 == 0 (32 x 32, 64 x 64: BASELINE cfg-3), staged:
// the chain's links come in as float4 (one pass over HBM, 32 KB into LDS), sin(theta) of every plaquette is formed
// from LDS, and the force leaves as float4 rows.  Same expressions in the same order as u1_force_kernel (identical
// bits); that kernel reads each link with four scalar loads per plaquette and writes scalars: 153 us per launch at
// cfg-3 (537 MB, 3.5 TB/s).
constexpr int kU1StageMax = 4096;
__global__ __launch_bounds__(kBlock) void u1_force_staged_f32_kernel(const float* __restrict__ x, float beta,
                                                                     float* __restrict__ force, float* v,
                                                                     float coef, int Tn, int Xn) {
  __shared__ __attribute__((aligned(16))) float xs[2 * kU1StageMax];
  __shared__ __attribute__((aligned(16))) float sth[kU1StageMax];
  const int c = blockIdx.x, V = Tn * Xn;
  const long o = (long)c * 2 * V;
  const float4* xc4 = reinterpret_cast<const float4*>(x + o);
  for (int i = threadIdx.x; i < V / 2; i += kBlock) reinterpret_cast<float4*>(xs)[i] = xc4[i];
  __syncthreads();
  for (int s = threadIdx.x; s < V; s += kBlock) sth[s] = sinf(plaq_angle<float>(xs, s / Xn, s % Xn, Tn, Xn));
  __syncthreads();
  for (int i = threadIdx.x; i < V / 4; i += kBlock) {
    const int s = 4 * i, t = s / Xn, xx = s - t * Xn;          // four sites of one row
    const int tm = (t == 0) ? Tn - 1 : t - 1;
    const float4 s0 = *reinterpret_cast<const float4*>(sth + s);
    const float4 stm = *reinterpret_cast<const float4*>(sth + tm * Xn + xx);
    const float sl = sth[t * Xn + (xx == 0 ? Xn - 1 : xx - 1)];
    const float4 f0 = make_float4(beta * (s0.x - sl), beta * (s0.y - s0.x), beta * (s0.z - s0.y), beta * (s0.w - s0.z));
    const float4 f1 = make_float4(beta * (-s0.x + stm.x), beta * (-s0.y + stm.y), beta * (-s0.z + stm.z),
                                  beta * (-s0.w + stm.w));
    if (force) {
      *reinterpret_cast<float4*>(force + o + s) = f0;
      *reinterpret_cast<float4*>(force + o + V + s) = f1;
    }
    if (v) {
      float4 a = *reinterpret_cast<float4*>(v + o + s), b = *reinterpret_cast<float4*>(v + o + V + s);
      a.x += coef * f0.x; a.y += coef * f0.y; a.z += coef * f0.z; a.w += coef * f0.w;
      b.x += coef * f1.x; b.y += coef * f1.y; b.z += coef * f1.z; b.w += coef * f1.w;
      *reinterpret_cast<float4*>(v + o + s) = a;
      *reinterpret_cast<float4*>(v + o + V + s) = b;
    }
  }
}

template <typename T, bool FWD, bool NCP>
__global__ __launch_bounds__(kBlock) void u1_x_update_kernel(T* x, const T* __restrict__ v,
                                                             const T* __restrict__ s,
                                                             const T* __restrict__ t,
                                                             const T* __restrict__ q,
                                                             const float* __restrict__ mask,
                                                             int complement, T eps, long n,
                                                             T* __restrict__ logdet) {
  __shared__ double lds[4];
  const long c = blockIdx.x;
  double ld = 0.0;
  for (long j = threadIdx.x; j < n; j += kBlock) {
    const long o = c * n + j;
    T keep = (T)mask[j];
    if (complement) keep = (T)1 - keep;
    const T mb = (T)1 - keep;
    const T xj = x[o];
    const T sj = FWD ? eps * s[o] : -eps * s[o];
    const T es = Math<T>::exp(sj);
    const T eq = Math<T>::exp(eps * q[o]);
    const T tr = v[o] * eq + t[o];
    T xp, l;
    if (NCP) {
      const T hx = xj / (T)2;
      const T x1 = (T)2 * Math<T>::atan(Math<T>::tan(hx) * es);
      xp = FWD ? (x1 + eps * tr) : (x1 - es * eps * tr);
      const T ch = Math<T>::cos(hx), sh = es * Math<T>::sin(hx);
      l = Math<T>::log(es / (ch * ch + sh * sh));
    } else {
      xp = FWD ? (xj * es + eps * tr) : (es * (xj - eps * tr));
      l = sj;
    }
    ld += (double)(mb * l);
    x[o] = wrap_angle<T>(keep * xj + mb * xp);
  }
  const double r = block_sum(ld, lds);
  if (threadIdx.x == 0) logdet[c] = (T)r;
}

template <typename T>
__global__ void u1_wrap_kernel(const T* x, T* y, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = wrap_angle<T>(x[i]);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void u1_kinetic_kernel(const T* __restrict__ v, long n,
                                                            T* __restrict__ out) {
  __shared__ double lds[4];
  const long c = blockIdx.x;
  double acc = 0.0;
  for (long j = threadIdx.x; j < n; j += kBlock) {
    const double p = (double)v[c * n + j];
    acc = fma(p, p, acc);
  }
  const double r = block_sum(acc, lds);
  if (threadIdx.x == 0) out[c] = (T)(0.5 * r);
}

// out[c][0:2][site] = cos(keep * x), out[c][2:4][site] = sin(keep * x)
template <typename T>
__global__ __launch_bounds__(kBlock) void u1_masked_cos_sin_kernel(const T* __restrict__ x,
                                                                   const float* __restrict__ mask,
                                                                   int complement,
                                                                   T* __restrict__ out, long n) {
  const long c = blockIdx.y;
  const long j = (long)blockIdx.x * kBlock + threadIdx.x;
  if (j >= n) return;
  T keep = (T)mask[j];
  if (complement) keep = (T)1 - keep;
  const T a = keep * x[c * n + j];
  out[c * 2 * n + j] = Math<T>::cos(a);
  out[c * 2 * n + n + j] = Math<T>::sin(a);
}

// PeriodicPadding(k-1) -> Conv2d(k) (cross-correlation, stride 1) -> MaxPool(pool) -> act.
// Padded index i of an (H + 2(k-1)) image maps to source (i - (k-1)) mod H, so the conv
// output has H + k - 1 rows (network.py:158-172 + nn.Conv2d).  One thread per pooled output.
__global__ __launch_bounds__(kBlock) void conv2d_periodic_kernel(
    const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, int cin, int H, int W, int cout, int k, int pool, int act, int Ho,
    int Wo, long total) {
  const long idx = (long)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= total) return;
  const int wo = (int)(idx % Wo);
  const int ho = (int)((idx / Wo) % Ho);
  const int f = (int)((idx / ((long)Wo * Ho)) % cout);
  const long b = idx / ((long)Wo * Ho * cout);
  const float* inb = in + b * (long)cin * H * W;
  const int pad = k - 1;
  float best = -3.402823466e38f;
  for (int ph = 0; ph < pool; ++ph)
    for (int pw = 0; pw < pool; ++pw) {
      const int r0 = ho * pool + ph, c0 = wo * pool + pw;     // conv output coordinates
      float acc = bias[f];
      for (int ci = 0; ci < cin; ++ci)
        for (int i = 0; i < k; ++i) {
          int r = (r0 + i - pad) % H; if (r < 0) r += H;
          for (int j = 0; j < k; ++j) {
            int cc = (c0 + j - pad) % W; if (cc < 0) cc += W;
            acc = fmaf(inb[(ci * H + r) * W + cc], w[((f * cin + ci) * k + i) * k + j], acc);
          }
        }
      best = fmaxf(best, acc);
    }
  out[idx] = act_f32(best, act);
}

// ---- conv stack as implicit GEMM: periodic im2col -> MFMA GEMM (gemm.hip) -> pool + act.
// col[m][kk], m = (b*Ho + ho)*Wo + wo, kk = (ci*k + i)*k + j (the flatten order of a Conv2d
// weight [cout][cin][k][k]); source pixel ((ho + i - (k-1)) mod H, (wo + j - (k-1)) mod W).
// Generic input strides so the first layer reads NCHW and later layers the GEMM's NHWC output.
// One thread row (threadIdx.y) per output pixel m: its (b, ho, wo) decomposition costs three
// integer divisions once; the 64 lanes of threadIdx.x then walk the Kc = C k k columns, whose
// (ci, i, j) decomposition divides by the compile-time kernel size only.  (The first version
// decomposed every element separately: ~10 runtime divisions per 4-byte store made this kernel
// 62 % of the conv stack's time.)
template <int KS>
__global__ __launch_bounds__(kBlock) void im2col_periodic_kernel(
    const float* __restrict__ in, long sn, long sc, long sh, long sw, int C, int H, int W, int kr,
    int Ho, int Wo, int Kc, long Mrows, int clast, float* __restrict__ col) {
  const int k = KS > 0 ? KS : kr;
  const long m = (long)blockIdx.x * 4 + threadIdx.y;
  if (m >= Mrows) return;
  const int wo = (int)(m % Wo);
  const long t = m / Wo;
  const int ho = (int)(t % Ho);
  const long b = t / Ho;
  const float* inb = in + b * sn;
  float* out = col + m * Kc;
  // first source row / column of the window, reduced once into [0, H) / [0, W)
  int r0 = (ho - (k - 1)) % H; if (r0 < 0) r0 += H;
  int c0 = (wo - (k - 1)) % W; if (c0 < 0) c0 += W;
  for (int kk = threadIdx.x; kk < Kc; kk += 64) {
    int j, i, ci;                 // column order (ci, i, j), or (i, j, ci) with clast
    if (clast) { ci = kk % C; const int ij = kk / C; j = ij % k; i = ij / k; }
    else { j = kk % k; const int ij = kk / k; i = ij % k; ci = ij / k; }
    int r = r0 + i; if (r >= H) r -= H; if (r >= H) r %= H;
    int c = c0 + j; if (c >= W) c -= W; if (c >= W) c %= W;
    out[kk] = inb[ci * sc + r * sh + c * sw];
  }
}

// NHWC max-pool (floor mode, stride = window) followed by the activation
__global__ __launch_bounds__(kBlock) void nchw_to_nhwc_pad_kernel(const float* __restrict__ in, int C,
                                                                  long HW, int CP, long total,
                                                                  float* __restrict__ out) {
  // out[b][hw][c] = c < C ? in[b][c][hw] : 0: the first conv layer's 2 / 4 lattice channels as
  // NHWC padded to a 16-byte group, so that it uses the vector gathers of the later layers
  const long idx = (long)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= total) return;
  const long b = idx / HW, p = idx % HW;
  const float* src = in + b * C * HW + p;
  float* dst = out + idx * CP;
  for (int c = 0; c < CP; ++c) dst[c] = c < C ? src[c * HW] : 0.f;
}

// VEC channels per thread: 4 = one 16-byte access (C % 4 == 0), else 1
template <int VEC>
__global__ __launch_bounds__(kBlock) void maxpool_act_nhwc_kernel(const float* __restrict__ in,
                                                                  int H, int W, int C, int pool,
                                                                  int act, int Ho, int Wo,
                                                                  long total,
                                                                  float* __restrict__ out) {
  typedef float fv __attribute__((ext_vector_type(VEC)));
  const long idx = (long)blockIdx.x * kBlock + threadIdx.x;      // over [b, ho, wo, C / VEC]
  if (idx >= total) return;
  const int cv = C / VEC;
  const int c = (int)(idx % cv) * VEC;
  const int wo = (int)((idx / cv) % Wo);
  const int ho = (int)((idx / ((long)cv * Wo)) % Ho);
  const long b = idx / ((long)cv * Wo * Ho);
  fv best;
#pragma unroll
  for (int e = 0; e < VEC; ++e) best[e] = -3.402823466e38f;
  for (int ph = 0; ph < pool; ++ph)
    for (int pw = 0; pw < pool; ++pw) {
      const fv v = *reinterpret_cast<const fv*>(in + ((b * H + ho * pool + ph) * W + wo * pool + pw) * C + c);
#pragma unroll
      for (int e = 0; e < VEC; ++e) best[e] = fmaxf(best[e], v[e]);
    }
#pragma unroll
  for (int e = 0; e < VEC; ++e) best[e] = act_f32(best[e], act);
  *reinterpret_cast<fv*>(out + ((b * Ho + ho) * Wo + wo) * (long)C + c) = best;
}

}  // namespace l2q

using namespace l2q;

extern "C" {

int l2q_u1_plaq_reduce(const void* x, int nb, int T_, int X_, int elem_bytes, void* out,
                       void* stream) {
  L2Q_REQUIRE(x && out, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && T_ > 0 && X_ > 0, L2Q_EINVAL, "non-positive size");
  hipStream_t st = (hipStream_t)stream;
  L2Q_DISPATCH_T(elem_bytes, hipLaunchKernelGGL(u1_plaq_kernel<T>, dim3(nb), dim3(kBlock), 0, st,
                                                (const T*)x, T_, X_, (T*)out));
  return check_launch("l2q_u1_plaq_reduce");
}

int l2q_u1_wilson_loops(const void* x, int nb, int T_, int X_, int elem_bytes, void* out, void* stream) {
  L2Q_REQUIRE(x && out, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && T_ > 0 && X_ > 0, L2Q_EINVAL, "non-positive size");
  hipStream_t st = (hipStream_t)stream;
  const long total = (long)nb * T_ * X_;
  L2Q_DISPATCH_T(elem_bytes, hipLaunchKernelGGL(u1_wloops_kernel<T>, dim3((unsigned)cdiv(total, kBlock)),
                                                dim3(kBlock), 0, st, (const T*)x, T_, X_, total, (T*)out));
  return check_launch("l2q_u1_wilson_loops");
}

int l2q_u1_wilson_loops_bwd(const void* g, int nb, int T_, int X_, int elem_bytes, void* dx, void* stream) {
  L2Q_REQUIRE(g && dx, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && T_ > 0 && X_ > 0, L2Q_EINVAL, "non-positive size");
  hipStream_t st = (hipStream_t)stream;
  const long total = (long)nb * T_ * X_;
  L2Q_DISPATCH_T(elem_bytes, hipLaunchKernelGGL(u1_wloops_bwd_kernel<T>, dim3((unsigned)cdiv(total, kBlock)),
                                                dim3(kBlock), 0, st, (const T*)g, T_, X_, total, (T*)dx));
  return check_launch("l2q_u1_wilson_loops_bwd");
}

int l2q_u1_force(const void* x, double beta, void* force, void* v, double coef, int nb, int T_,
                 int X_, int elem_bytes, void* stream) {
  L2Q_REQUIRE(x && (force || v), L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && T_ > 0 && X_ > 0, L2Q_EINVAL, "non-positive size");
  hipStream_t st = (hipStream_t)stream;
  const long V = (long)T_ * X_;
  if (elem_bytes == 4 && X_ % 4 == 0 && V >= 1024 && V <= kU1StageMax &&
      (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(force) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(v) & 15) == 0) {
    hipLaunchKernelGGL(u1_force_staged_f32_kernel, dim3(nb), dim3(kBlock), 0, st, (const float*)x, (float)beta,
                       (float*)force, (float*)v, (float)coef, T_, X_);
    return check_launch("l2q_u1_force");
  }
  L2Q_DISPATCH_T(elem_bytes,
                 hipLaunchKernelGGL(u1_force_kernel<T>, dim3(nb), dim3(kBlock), 0, st, (const T*)x,
                                    (T)beta, (T*)force, (T*)v, (T)coef, T_, X_));
  return check_launch("l2q_u1_force");
}

int l2q_u1_x_update(void* x, const void* v, const void* s, const void* t, const void* q,
                    const float* mask, int complement, double eps, int forward, int use_ncp,
                    int elem_bytes, int nb, long n, void* logdet, void* stream) {
  L2Q_REQUIRE(x && v && s && t && q && mask && logdet, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && n > 0, L2Q_EINVAL, "non-positive size");
  hipStream_t st = (hipStream_t)stream;
#define L2Q_XUPD(F, N)                                                                        \
  hipLaunchKernelGGL((u1_x_update_kernel<T, F, N>), dim3(nb), dim3(kBlock), 0, st, (T*)x,     \
                     (const T*)v, (const T*)s, (const T*)t, (const T*)q, mask, complement,    \
                     (T)eps, n, (T*)logdet)
  L2Q_DISPATCH_T(elem_bytes, {
    if (forward) { if (use_ncp) L2Q_XUPD(true, true); else L2Q_XUPD(true, false); }
    else { if (use_ncp) L2Q_XUPD(false, true); else L2Q_XUPD(false, false); }
  });
#undef L2Q_XUPD
  return check_launch("l2q_u1_x_update");
}

int l2q_u1_wrap(const void* x, void* y, long n, int elem_bytes, void* stream) {
  L2Q_REQUIRE(x && y, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(n > 0, L2Q_EINVAL, "non-positive size");
  hipStream_t st = (hipStream_t)stream;
  L2Q_DISPATCH_T(elem_bytes,
                 hipLaunchKernelGGL(u1_wrap_kernel<T>, dim3((unsigned)cdiv(n, kBlock)),
                                    dim3(kBlock), 0, st, (const T*)x, (T*)y, n));
  return check_launch("l2q_u1_wrap");
}

int l2q_u1_kinetic_reduce(const void* v, int nb, long n, int elem_bytes, void* out, void* stream) {
  L2Q_REQUIRE(v && out, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && n > 0, L2Q_EINVAL, "non-positive size");
  hipStream_t st = (hipStream_t)stream;
  L2Q_DISPATCH_T(elem_bytes, hipLaunchKernelGGL(u1_kinetic_kernel<T>, dim3(nb), dim3(kBlock), 0, st,
                                                (const T*)v, n, (T*)out));
  return check_launch("l2q_u1_kinetic_reduce");
}

int l2q_u1_masked_cos_sin(const void* x, const float* mask, int complement, void* out, int nb,
                          long n, int elem_bytes, void* stream) {
  L2Q_REQUIRE(x && mask && out, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && n > 0, L2Q_EINVAL, "non-positive size");
  hipStream_t st = (hipStream_t)stream;
  L2Q_DISPATCH_T(elem_bytes,
                 hipLaunchKernelGGL(u1_masked_cos_sin_kernel<T>,
                                    dim3((unsigned)cdiv(n, kBlock), (unsigned)nb), dim3(kBlock), 0,
                                    st, (const T*)x, mask, complement, (T*)out, n));
  return check_launch("l2q_u1_masked_cos_sin");
}

int l2q_conv2d_periodic_f32(const float* in, const float* w, const float* bias, float* out, int nb,
                            int cin, int H, int W, int cout, int k, int pool, int act,
                            void* stream) {
  L2Q_REQUIRE(in && w && bias && out, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && cin > 0 && H > 0 && W > 0 && cout > 0 && k > 0, L2Q_EINVAL,
              "non-positive size");
  if (pool < 1) pool = 1;
  const int Hc = H + k - 1, Wc = W + k - 1;         // conv output extent
  const int Ho = Hc / pool, Wo = Wc / pool;
  L2Q_REQUIRE(Ho > 0 && Wo > 0, L2Q_ESHAPE, "pooling window larger than the image");
  const long total = (long)nb * cout * Ho * Wo;
  hipLaunchKernelGGL(conv2d_periodic_kernel, dim3((unsigned)cdiv(total, kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, in, w, bias, out, cin, H, W, cout, k, pool, act, Ho, Wo,
                     total);
  return check_launch("l2q_conv2d_periodic_f32");
}

int l2q_nchw_to_nhwc_pad_f32(const float* in, int nb, int C, int H, int W, int cpad, float* out,
                             void* stream) {
  L2Q_REQUIRE(in && out, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && C > 0 && H > 0 && W > 0 && cpad >= C, L2Q_EINVAL, "bad size");
  const long HW = (long)H * W, total = (long)nb * HW;
  hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel, dim3((unsigned)cdiv(total, kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, in, C, HW, cpad, total, out);
  return check_launch("l2q_nchw_to_nhwc_pad_f32");
}

int l2q_im2col_periodic_f32(const float* in, long sn, long sc, long sh, long sw, int nb, int C,
                            int H, int W, int k, int channels_last_cols, float* col,
                            void* stream) {
  L2Q_REQUIRE(in && col, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && C > 0 && H > 0 && W > 0 && k > 0, L2Q_EINVAL, "non-positive size");
  const int Ho = H + k - 1, Wo = W + k - 1, Kc = C * k * k;
  const long Mrows = (long)nb * Ho * Wo;
  L2Q_REQUIRE(cdiv(Mrows, 4) < 0x7fffffffL, L2Q_ESHAPE, "grid too large");
  const dim3 grid((unsigned)cdiv(Mrows, 4)), block(64, 4);
  hipStream_t st = (hipStream_t)stream;
#define L2Q_I2C(KS)                                                                              \
  hipLaunchKernelGGL(im2col_periodic_kernel<KS>, grid, block, 0, st, in, sn, sc, sh, sw, C, H, W, \
                     k, Ho, Wo, Kc, Mrows, channels_last_cols ? 1 : 0, col)
  switch (k) {
    case 1: L2Q_I2C(1); break;
    case 2: L2Q_I2C(2); break;
    case 3: L2Q_I2C(3); break;
    case 4: L2Q_I2C(4); break;
    case 5: L2Q_I2C(5); break;
    default: L2Q_I2C(0); break;
  }
#undef L2Q_I2C
  return check_launch("l2q_im2col_periodic_f32");
}

int l2q_maxpool_act_nhwc_f32(const float* in, int nb, int H, int W, int C, int pool, int act,
                             float* out, void* stream) {
  L2Q_REQUIRE(in && out, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && H > 0 && W > 0 && C > 0 && pool > 0, L2Q_EINVAL, "non-positive size");
  const int Ho = H / pool, Wo = W / pool;
  L2Q_REQUIRE(Ho > 0 && Wo > 0, L2Q_ESHAPE, "pooling window larger than the image");
  const bool vec = C % 4 == 0 && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  const long total = (long)nb * Ho * Wo * (vec ? C / 4 : C);
  const dim3 grid((unsigned)cdiv(total, kBlock)), block(kBlock);
  if (vec) hipLaunchKernelGGL(maxpool_act_nhwc_kernel<4>, grid, block, 0, (hipStream_t)stream, in, H,
                              W, C, pool, act, Ho, Wo, total, out);
  else hipLaunchKernelGGL(maxpool_act_nhwc_kernel<1>, grid, block, 0, (hipStream_t)stream, in, H, W,
                          C, pool, act, Ho, Wo, total, out);
  return check_launch("l2q_maxpool_act_nhwc_f32");
}

}  // extern "C"
