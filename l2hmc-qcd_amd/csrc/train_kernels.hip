// train_kernels.hip -- reverse-mode (VJP) kernels of the U(1) L2HMC training step, the
// train-mode network layers (BatchNorm1d batch statistics, dropout, conv/pool backward) and
// the fused Adam update, for gfx950.
//
// The reference obtains all of this from torch.autograd (trainers/pytorch/trainer.py:1284-1314,
// loss.backward()); here every sub-update of the leapfrog integrator has an explicit
// cotangent kernel and the host replays the trajectory tape in reverse (dynamics/pytorch/
// training.py).  Per-chain reductions (d eps) finish inside one workgroup in a fixed order.
#include "l2q_common.hpp"
#include "u1_math.hpp"

namespace l2q {

// ------------------------------------------------------------------ element-wise helpers
template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, int act, long n,
                               T* __restrict__ dx) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const T yy = y[i];
  T d;
  switch (act) {
    case L2Q_ACT_TANH: d = (T)1 - yy * yy; break;
    case L2Q_ACT_RELU: d = yy > (T)0 ? (T)1 : (T)0; break;
    case L2Q_ACT_LEAKY_RELU: d = yy > (T)0 ? (T)1 : (T)0.01; break;
    case L2Q_ACT_ELU: d = yy > (T)0 ? (T)1 : yy + (T)1; break;     // y = e^z - 1 -> dy/dz = y + 1
    case L2Q_ACT_SWISH: {                                           // `y` holds the PRE-activation z
      const T sg = (T)1 / ((T)1 + exp(-yy));
      d = sg * ((T)1 + yy * ((T)1 - sg));
      break;
    }
    default: d = (T)1; break;
  }
  dx[i] = dy[i] * d;
}

template <typename T>
__global__ void act_fwd_kernel(const T* __restrict__ x, int act, long n, T* __restrict__ y) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const T z = x[i];
  T r;
  switch (act) {
    case L2Q_ACT_TANH: r = tanh(z); break;
    case L2Q_ACT_RELU: r = z > (T)0 ? z : (T)0; break;
    case L2Q_ACT_LEAKY_RELU: r = z > (T)0 ? z : (T)0.01 * z; break;
    case L2Q_ACT_ELU: r = z > (T)0 ? z : expm1(z); break;
    case L2Q_ACT_SWISH: r = z / ((T)1 + exp(-z)); break;
    default: r = z; break;
  }
  y[i] = r;
}

template <typename T>
__global__ void mul_kernel(const T* __restrict__ a, const T* __restrict__ b, T alpha, long n,
                           T* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = alpha * a[i] * b[i];
}

template <typename T>
__global__ void axpy_rows_kernel(const T* __restrict__ x, const T* __restrict__ a, long n,
                                 T* __restrict__ y) {
  const long c = blockIdx.y;
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) y[c * n + j] += a[c] * x[c * n + j];
}

// out[n] (+)= alpha * sum_m a[m][n] * (b ? b[m][n] : 1), two order-stable stages.
// Stage 1: a block of CX columns x (1024 / CX) row lanes reduces `rpb` consecutive rows into
// partial[r][n] (CX = 8..64 by N so that a wavefront still reads contiguous memory when N is
// small: conv bias gradients have N = 8 and M ~ 10^6 rows).  Stage 2 sums the partials.
template <typename T>
__global__ __launch_bounds__(1024) void colsum_partial_kernel(const T* __restrict__ a,
                                                              const T* __restrict__ b, long M,
                                                              int N, long rpb,
                                                              double* __restrict__ partial) {
  __shared__ double part[1024];
  const int CX = blockDim.x, RY = blockDim.y;
  const int cx = threadIdx.x, ry = threadIdx.y;
  const long col = (long)blockIdx.x * CX + cx;
  const long r0 = (long)blockIdx.y * rpb;
  long r1 = r0 + rpb; if (r1 > M) r1 = M;
  double s = 0.0;
  if (col < N) {
    for (long m = r0 + ry; m < r1; m += RY) {
      const double av = (double)a[m * N + col];
      s += b ? av * (double)b[m * N + col] : av;
    }
  }
  part[ry * CX + cx] = s;
  __syncthreads();
  if (ry == 0 && col < N) {
    double r = 0.0;
    for (int k = 0; k < RY; ++k) r += part[k * CX + cx];
    partial[(long)blockIdx.y * N + col] = r;
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void colsum_final_kernel(const double* __restrict__ partial,
                                                              long R, int N, double alpha,
                                                              int accumulate, T* __restrict__ out) {
  __shared__ double part[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const long col = (long)blockIdx.x * 64 + cx;
  double s = 0.0;
  if (col < N)
    for (long r = ry; r < R; r += 4) s += partial[r * N + col];
  part[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && col < N) {
    const double r = alpha * (part[0][cx] + part[1][cx] + part[2][cx] + part[3][cx]);
    out[col] = accumulate ? (T)((double)out[col] + r) : (T)r;
  }
}

// s = scale * exp(coeff[n]) * tanh(pre)  ->  dpre = ds * g * (1 - th^2), g = scale e^c, th = s / g.
// coeff == NULL: plain linear head, dpre = scale * ds.
template <typename T>
__global__ void scaled_tanh_bwd_kernel(const T* __restrict__ ds, const T* __restrict__ s,
                                       const T* __restrict__ coeff, T scale, int N, long total,
                                       T* __restrict__ dpre) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  if (!coeff) { dpre[i] = scale * ds[i]; return; }
  const T g = scale * Math<T>::exp(coeff[i % N]);
  if (g == (T)0) { dpre[i] = (T)0; return; }
  const T th = s[i] / g;
  dpre[i] = ds[i] * g * ((T)1 - th * th);
}

// The same VJP with the two column sums the head's parameter gradients need formed in the same pass:
// partial_b[r][n] = sum over the block's rows of dpre (bias gradient), partial_c[r][n] = sum of ds * s
// (d s / d coeff = s: the ScaledTanh coefficient gradient).  Row partition, summation order and the
// second stage are those of l2q_colsum: the results are the bits of the three separate passes.
template <typename T>
__global__ __launch_bounds__(1024) void scaled_tanh_bwd_sums_kernel(
    const T* __restrict__ ds, const T* __restrict__ s, const T* __restrict__ coeff, T scale, long M, int N,
    long rpb, T* __restrict__ dpre, double* __restrict__ partial_b, double* __restrict__ partial_c) {
  __shared__ double part[2][1024];
  const int CX = blockDim.x, RY = blockDim.y;
  const int cx = threadIdx.x, ry = threadIdx.y;
  const long col = (long)blockIdx.x * CX + cx;
  const long r0 = (long)blockIdx.y * rpb;
  long r1 = r0 + rpb; if (r1 > M) r1 = M;
  double sb = 0.0, sc = 0.0;
  if (col < N) {
    const T g = coeff ? scale * Math<T>::exp(coeff[col]) : scale;
    for (long m = r0 + ry; m < r1; m += RY) {
      const long i = m * N + col;
      const T d = ds[i];
      T o;
      if (!coeff) o = scale * d;
      else if (g == (T)0) o = (T)0;
      else { const T th = s[i] / g; o = d * g * ((T)1 - th * th); }
      dpre[i] = o;
      sb += (double)o;
      if (coeff) sc += (double)d * (double)s[i];
    }
  }
  part[0][ry * CX + cx] = sb;
  part[1][ry * CX + cx] = sc;
  __syncthreads();
  if (ry == 0 && col < N) {
    double rb = 0.0, rc = 0.0;
    for (int k = 0; k < RY; ++k) { rb += part[0][k * CX + cx]; rc += part[1][k * CX + cx]; }
    partial_b[(long)blockIdx.y * N + col] = rb;
    if (coeff) partial_c[(long)blockIdx.y * N + col] = rc;
  }
}

// ------------------------------------------------------------------ BatchNorm1d (train mode)
// One workgroup per feature column (N is 16..256 here, M = chains).
template <typename T>
__global__ __launch_bounds__(kBlock) void bn_train_fwd_kernel(
    const T* __restrict__ x, const T* __restrict__ gamma, const T* __restrict__ beta, double eps,
    double momentum, T* running_mean, T* running_var, int M, int N, T* __restrict__ y,
    T* __restrict__ save_mean, T* __restrict__ save_invstd) {
  __shared__ double lds[4];
  __shared__ double bc[2];
  const int n = blockIdx.x;
  double s = 0.0;
  for (long m = threadIdx.x; m < M; m += kBlock) s += (double)x[m * N + n];
  const double tot = block_sum(s, lds);
  if (threadIdx.x == 0) bc[0] = tot / M;
  __syncthreads();
  const double mean = bc[0];
  double q = 0.0;
  for (long m = threadIdx.x; m < M; m += kBlock) {
    const double d = (double)x[m * N + n] - mean;
    q = fma(d, d, q);
  }
  const double qt = block_sum(q, lds);
  if (threadIdx.x == 0) {
    const double var = qt / M;
    bc[1] = 1.0 / sqrt(var + eps);
    save_mean[n] = (T)mean;
    save_invstd[n] = (T)bc[1];
    if (running_mean) {
      const double unb = M > 1 ? qt / (M - 1) : var;
      running_mean[n] = (T)((1.0 - momentum) * (double)running_mean[n] + momentum * mean);
      running_var[n] = (T)((1.0 - momentum) * (double)running_var[n] + momentum * unb);
    }
  }
  __syncthreads();
  const double inv = bc[1];
  const double g = (double)gamma[n], b = (double)beta[n];
  for (long m = threadIdx.x; m < M; m += kBlock)
    y[m * N + n] = (T)(((double)x[m * N + n] - mean) * inv * g + b);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void bn_bwd_kernel(
    const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ save_mean,
    const T* __restrict__ save_invstd, const T* __restrict__ gamma, int M, int N,
    T* __restrict__ dx, T* dgamma, T* dbeta) {
  __shared__ double lds[4];
  __shared__ double bc[2];
  const int n = blockIdx.x;
  const double mean = (double)save_mean[n], inv = (double)save_invstd[n];
  double sb = 0.0, sg = 0.0;
  for (long m = threadIdx.x; m < M; m += kBlock) {
    const double d = (double)dy[m * N + n];
    sb += d;
    sg = fma(d, ((double)x[m * N + n] - mean) * inv, sg);
  }
  const double tb = block_sum(sb, lds);
  const double tg = block_sum(sg, lds);
  if (threadIdx.x == 0) {
    bc[0] = tb; bc[1] = tg;
    dbeta[n] = (T)((double)dbeta[n] + tb);
    dgamma[n] = (T)((double)dgamma[n] + tg);
  }
  __syncthreads();
  const double db = bc[0], dg = bc[1], g = (double)gamma[n];
  for (long m = threadIdx.x; m < M; m += kBlock) {
    const double xh = ((double)x[m * N + n] - mean) * inv;
    dx[m * N + n] = (T)(g * inv / M * (M * (double)dy[m * N + n] - db - xh * dg));
  }
}

// ------------------------------------------------------------------ conv stack backward
// Adjoint of im2col_periodic: dx[b][ci][r][c] = sum over (ho, i), (wo, j) whose source pixel
// is (r, c) of dcol[(b, ho, wo)][(ci, i, j)].  Gather form (no atomics): one thread per input
// element enumerates its (ho, i) x (wo, j) pre-images in a fixed order.
__global__ __launch_bounds__(kBlock) void col2im_periodic_kernel(
    const float* __restrict__ dcol, long sn, long sc, long sh, long sw, int C, int H, int W, int k,
    int Ho, int Wo, int Kc, long total, int clast, float* __restrict__ dx) {
  const long idx = (long)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= total) return;
  // thread order follows the fastest index of dcol's columns: (b, ci, r, c) for the (ci, i, j)
  // order, (b, r, c, ci) for (i, j, ci) -- there a wavefront reads runs of consecutive channels
  int c, r, ci;
  long b;
  if (clast) {
    ci = (int)(idx % C);
    c = (int)((idx / C) % W);
    r = (int)((idx / ((long)C * W)) % H);
    b = idx / ((long)C * W * H);
  } else {
    c = (int)(idx % W);
    r = (int)((idx / W) % H);
    ci = (int)((idx / ((long)W * H)) % C);
    b = idx / ((long)W * H * C);
  }
  float acc = 0.0f;
  for (int i = 0; i < k; ++i) {
    int h0 = (r + (k - 1) - i) % H; if (h0 < 0) h0 += H;
    for (int ho = h0; ho < Ho; ho += H)
      for (int j = 0; j < k; ++j) {
        int w0 = (c + (k - 1) - j) % W; if (w0 < 0) w0 += W;
        for (int wo = w0; wo < Wo; wo += W)
          acc += dcol[((b * Ho + ho) * (long)Wo + wo) * Kc +
                      (clast ? (i * k + j) * C + ci : (ci * k + i) * k + j)];
      }
  }
  dx[b * sn + ci * sc + r * sh + c * sw] = acc;
}

// out = act(maxpool(in)) (NHWC, floor mode).  din[window argmax] = dout * act'(out); every
// other input element (and the rows / columns the floor drops) gets 0.  First maximum in
// row-major window order wins, like nn.MaxPool2d.
__global__ __launch_bounds__(kBlock) void maxpool_act_nhwc_bwd_kernel(
    const float* __restrict__ dout, const float* __restrict__ out, const float* __restrict__ in,
    int H, int W, int C, int pool, int act, int Ho, int Wo, long total, float* __restrict__ din) {
  const long idx = (long)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const int w = (int)((idx / C) % W);
  const int h = (int)((idx / ((long)C * W)) % H);
  const long b = idx / ((long)C * W * H);
  const int ho = h / pool, wo = w / pool;
  float g = 0.0f;
  if (ho < Ho && wo < Wo) {
    float best = -3.402823466e38f;
    int bh = 0, bw = 0;
    for (int ph = 0; ph < pool; ++ph)
      for (int pw = 0; pw < pool; ++pw) {
        const float val = in[((b * H + ho * pool + ph) * W + wo * pool + pw) * C + c];
        if (val > best) { best = val; bh = ph; bw = pw; }
      }
    if (ho * pool + bh == h && wo * pool + bw == w) {
      const long o = ((b * Ho + ho) * (long)Wo + wo) * C + c;
      const float y = out[o];
      float d;
      switch (act) {
        case L2Q_ACT_TANH: d = 1.0f - y * y; break;
        case L2Q_ACT_RELU: d = y > 0.0f ? 1.0f : 0.0f; break;
        case L2Q_ACT_LEAKY_RELU: d = y > 0.0f ? 1.0f : 0.01f; break;
        case L2Q_ACT_ELU: d = y > 0.0f ? 1.0f : y + 1.0f; break;
        default: d = 1.0f; break;
      }
      g = dout[o] * d;
    }
  }
  din[idx] = g;
}

// ------------------------------------------------------------------ U(1) physics cotangents
// theta = D x (plaq_angle);   D^T acting on a plaquette field w:
//   (D^T w)_0(t,x) = w(t,x) - w(t,x-1),   (D^T w)_1(t,x) = w(t-1,x) - w(t,x)
// One workgroup per chain, plaquette cotangent staged in LDS.
constexpr int kU1BwdMaxLds = 8192;

// force F = beta D^T sin(D x):  dx += D^T [ cos(theta) * beta * (D dF) ]
// plaq  L(sum cos, sum sin):    dx += D^T [ -gcos sin(theta) + gsin cos(theta) ]
template <typename T, bool FORCE>
__global__ __launch_bounds__(kBlock) void u1_stencil_bwd_kernel(
    const T* __restrict__ x, const T* __restrict__ dF, T beta, const T* __restrict__ gcos,
    const T* __restrict__ gsin, int Tn, int Xn, T* dx) {
  __shared__ T wl[kU1BwdMaxLds];
  const int c = blockIdx.x, V = Tn * Xn;
  const T* xc = x + (long)c * 2 * V;
  T gc = (T)0, gs = (T)0;
  if (!FORCE) { gc = gcos ? gcos[c] : (T)0; gs = gsin ? gsin[c] : (T)0; }
  for (int s = threadIdx.x; s < V; s += kBlock) {
    const int t = s / Xn, xx = s % Xn;
    const T th = plaq_angle(xc, t, xx, Tn, Xn);
    T w;
    if (FORCE) {
      const T* dc = dF + (long)c * 2 * V;
      w = beta * plaq_angle(dc, t, xx, Tn, Xn) * Math<T>::cos(th);
    } else {
      w = -gc * Math<T>::sin(th) + gs * Math<T>::cos(th);
    }
    wl[s] = w;
  }
  __syncthreads();
  T* dxc = dx + (long)c * 2 * V;
  for (int s = threadIdx.x; s < V; s += kBlock) {
    const int t = s / Xn, xx = s % Xn;
    const int tm = (t == 0) ? Tn - 1 : t - 1;
    const int xm = (xx == 0) ? Xn - 1 : xx - 1;
    const T w0 = wl[s];
    dxc[s] += w0 - wl[t * Xn + xm];
    dxc[V + s] += wl[tm * Xn + xx] - w0;
  }
}

// cotangents of one x sub-update (dynamics.py:1386-1477); see DESIGN.md section "training".
template <typename T, bool FWD, bool NCP>
__global__ __launch_bounds__(kBlock) void u1_x_update_bwd_kernel(
    const T* __restrict__ x, const T* __restrict__ v, const T* __restrict__ s,
    const T* __restrict__ t, const T* __restrict__ q, const float* __restrict__ mask,
    int complement, T eps, const T* __restrict__ gx, const T* __restrict__ gl, long n,
    T* __restrict__ dx, T* dv, T* __restrict__ ds, T* __restrict__ dt, T* __restrict__ dq,
    T* __restrict__ deps) {
  __shared__ double lds[4];
  const long c = blockIdx.x;
  const T glc = gl ? gl[c] : (T)0;
  double de = 0.0;
  for (long j = threadIdx.x; j < n; j += kBlock) {
    const long o = c * n + j;
    T keep = (T)mask[j];
    if (complement) keep = (T)1 - keep;
    const T mb = (T)1 - keep;
    const T g = gx[o];
    if (mb == (T)0) {                 // element kept: x' = x, nothing reaches s, t, q, v
      dx[o] = keep * g;
      ds[o] = (T)0; dt[o] = (T)0; dq[o] = (T)0;
      continue;
    }
    const T xj = x[o], vj = v[o], sj = s[o], tj = t[o], qj = q[o];
    const T S = FWD ? eps * sj : -eps * sj;
    const T es = Math<T>::exp(S);
    const T eq = Math<T>::exp(eps * qj);
    const T tr = vj * eq + tj;                       // v e^Q + t
    const T gp = mb * g;                             // cotangent of xp
    const T glm = mb * glc;                          // cotangent of this element's log-det term
    T dxj, dS, dA;                                   // dA: cotangent of A = eps * tr
    if (NCP) {
      const T hx = xj / (T)2;
      const T ch = Math<T>::cos(hx), sh = Math<T>::sin(hx);
      const T D = ch * ch + es * es * sh * sh;
      const T sx = (T)2 * sh * ch;                   // sin x
      const T x1_x = es / D, x1_S = sx * es / D;
      const T ld_x = -(es * es - (T)1) * sx / ((T)2 * D);
      const T ld_S = (T)1 - (T)2 * es * es * sh * sh / D;
      dxj = gp * x1_x + glm * ld_x;
      dS = gp * x1_S + glm * ld_S;
      if (FWD) dA = gp;
      else { dA = -gp * es; dS += -gp * es * eps * tr; }
    } else {
      if (FWD) { dxj = gp * es; dS = gp * xj * es + glm; dA = gp; }
      else {
        dxj = gp * es; dS = gp * es * (xj - eps * tr) + glm; dA = -gp * es;
      }
    }
    dx[o] = keep * g + dxj;
    dv[o] += dA * eps * eq;
    dt[o] = dA * eps;
    const T dQ = dA * eps * vj * eq;
    ds[o] = FWD ? eps * dS : -eps * dS;
    dq[o] = eps * dQ;
    de += (double)(dA * tr + (FWD ? dS * sj : -dS * sj) + dQ * qj);
  }
  const double r = block_sum(de, lds);
  if (threadIdx.x == 0) deps[c] = (T)r;
}

// cotangents of one (real) v sub-update (dynamics.py:1266-1297)
template <typename T, bool FWD>
__global__ __launch_bounds__(kBlock) void v_update_bwd_kernel(
    const T* __restrict__ v, const T* __restrict__ force, const T* __restrict__ s,
    const T* __restrict__ t, const T* __restrict__ q, T eps, const T* __restrict__ gv,
    const T* __restrict__ gl, long n, T* __restrict__ dv, T* __restrict__ dF, T* __restrict__ ds,
    T* __restrict__ dt, T* __restrict__ dq, T* __restrict__ deps) {
  __shared__ double lds[4];
  const long c = blockIdx.x;
  const T glc = gl ? gl[c] : (T)0;
  const T half = (T)0.5;
  double de = 0.0;
  for (long j = threadIdx.x; j < n; j += kBlock) {
    const long o = c * n + j;
    const T vj = v[o], fj = force[o], sj = s[o], tj = t[o], qj = q[o], g = gv[o];
    const T S = FWD ? half * eps * sj : -half * eps * sj;
    const T es = Math<T>::exp(S);
    const T eq = Math<T>::exp(eps * qj);
    const T fq = fj * eq + tj;
    T dS, dB;                                        // B = eps/2 * fq
    if (FWD) { dS = g * vj * es + glc; dB = -g; }
    else { dS = g * es * (vj + half * eps * fq) + glc; dB = g * es; }
    dv[o] = g * es;
    dF[o] = dB * half * eps * eq;
    dt[o] = dB * half * eps;
    const T dQ = dB * half * eps * fj * eq;
    ds[o] = FWD ? half * eps * dS : -half * eps * dS;
    dq[o] = eps * dQ;
    de += (double)(dB * half * fq + (FWD ? half * sj * dS : -half * sj * dS) + dQ * qj);
  }
  const double r = block_sum(de, lds);
  if (threadIdx.x == 0) deps[c] = (T)r;
}

// dx += keep * (-sin(keep x) dcos + cos(keep x) dsin)
template <typename T>
__global__ __launch_bounds__(kBlock) void u1_masked_cos_sin_bwd_kernel(
    const T* __restrict__ x, const float* __restrict__ mask, int complement,
    const T* __restrict__ dout, long n, T* dx) {
  const long c = blockIdx.y;
  const long j = (long)blockIdx.x * kBlock + threadIdx.x;
  if (j >= n) return;
  T keep = (T)mask[j];
  if (complement) keep = (T)1 - keep;
  const T a = keep * x[c * n + j];
  dx[c * n + j] += keep * (-Math<T>::sin(a) * dout[c * 2 * n + j]
                           + Math<T>::cos(a) * dout[c * 2 * n + n + j]);
}

// ------------------------------------------------------------------ optimiser
// torch.optim.Adam (no weight decay, no amsgrad) over one flat parameter arena:
//   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
template <typename T>
__global__ void adam_kernel(T* p, const T* __restrict__ g, T* m, T* v, long n, T lr, T b1, T b2,
                            T eps, T bc1, T sqrt_bc2, T gscale) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const T gi = g[i] * gscale;
  const T mi = b1 * m[i] + ((T)1 - b1) * gi;
  const T vi = b2 * v[i] + ((T)1 - b2) * gi * gi;
  m[i] = mi; v[i] = vi;
  const T denom = sqrt(vi) / sqrt_bc2 + eps;
  p[i] -= (lr / bc1) * mi / denom;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void sumsq_partial_kernel(const T* __restrict__ a, long n,
                                                               double* __restrict__ partial) {
  __shared__ double lds[4];
  double s = 0.0;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) {
    const double d = (double)a[i];
    s = fma(d, d, s);
  }
  const double r = block_sum(s, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

}  // namespace l2q

using namespace l2q;

static inline unsigned grid1(long n, int block = kBlock) { return (unsigned)cdiv(n, block); }

extern "C" {

int l2q_act_bwd(const void* dy, const void* y, int act, long n, int elem_bytes, void* dx,
                void* stream) {
  L2Q_REQUIRE(dy && y && dx, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(n > 0, L2Q_EINVAL, "non-positive size");
  hipStream_t st = (hipStream_t)stream;
  L2Q_DISPATCH_T(elem_bytes, hipLaunchKernelGGL(act_bwd_kernel<T>, dim3(grid1(n)), dim3(kBlock), 0,
                                                st, (const T*)dy, (const T*)y, act, n, (T*)dx));
  return check_launch("l2q_act_bwd");
}

int l2q_act_fwd(const void* x, int act, long n, int elem_bytes, void* y, void* stream) {
  L2Q_REQUIRE(x && y, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(n > 0, L2Q_EINVAL, "non-positive size");
  hipStream_t st = (hipStream_t)stream;
  L2Q_DISPATCH_T(elem_bytes, hipLaunchKernelGGL(act_fwd_kernel<T>, dim3(grid1(n)), dim3(kBlock), 0,
                                                st, (const T*)x, act, n, (T*)y));
  return check_launch("l2q_act_fwd");
}

int l2q_mul(const void* a, const void* b, double alpha, long n, int elem_bytes, void* out,
            void* stream) {
  L2Q_REQUIRE(a && b && out, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(n > 0, L2Q_EINVAL, "non-positive size");
  hipStream_t st = (hipStream_t)stream;
  L2Q_DISPATCH_T(elem_bytes, hipLaunchKernelGGL(mul_kernel<T>, dim3(grid1(n)), dim3(kBlock), 0, st,
                                                (const T*)a, (const T*)b, (T)alpha, n, (T*)out));
  return check_launch("l2q_mul");
}

int l2q_axpy_rows(const void* x, const void* a, int nb, long n, int elem_bytes, void* y,
                  void* stream) {
  L2Q_REQUIRE(x && a && y, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && n > 0, L2Q_EINVAL, "non-positive size");
  hipStream_t st = (hipStream_t)stream;
  L2Q_DISPATCH_T(elem_bytes,
                 hipLaunchKernelGGL(axpy_rows_kernel<T>, dim3(grid1(n), (unsigned)nb), dim3(kBlock),
                                    0, st, (const T*)x, (const T*)a, n, (T*)y));
  return check_launch("l2q_axpy_rows");
}

static long colsum_rows_per_block(long M) {
  long rpb = M >= 4096 ? 256 : 64;
  if (cdiv(M, rpb) > 4096) rpb = cdiv(M, 4096);
  return rpb;
}

size_t l2q_colsum_ws_bytes(long M, int N) {
  if (M <= 0 || N <= 0) return 0;
  return (size_t)cdiv(M, colsum_rows_per_block(M)) * (size_t)N * sizeof(double);
}

int l2q_colsum(const void* a, const void* b, long M, int N, double alpha, int accumulate,
               int elem_bytes, void* out, void* ws, size_t ws_bytes, void* stream) {
  L2Q_REQUIRE(a && out && ws, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(M > 0 && N > 0, L2Q_EINVAL, "non-positive size");
  L2Q_REQUIRE(ws_bytes >= l2q_colsum_ws_bytes(M, N), L2Q_EINVAL, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const long rpb = colsum_rows_per_block(M);
  const long R = cdiv(M, rpb);
  int cx = 64;
  while (cx > 8 && cx / 2 >= N) cx /= 2;
  const dim3 blk(cx, 1024 / cx), grid((unsigned)cdiv(N, cx), (unsigned)R);
  L2Q_DISPATCH_T(elem_bytes, {
    hipLaunchKernelGGL(colsum_partial_kernel<T>, grid, blk, 0, st, (const T*)a, (const T*)b, M, N,
                       rpb, (double*)ws);
    hipLaunchKernelGGL(colsum_final_kernel<T>, dim3(grid1(N, 64)), dim3(kBlock), 0, st,
                       (const double*)ws, R, N, alpha, accumulate, (T*)out);
  });
  return check_launch("l2q_colsum");
}

int l2q_scaled_tanh_bwd(const void* ds, const void* s, const void* coeff, double scale, int M,
                        int N, int elem_bytes, void* dpre, void* stream) {
  L2Q_REQUIRE(ds && dpre && (s || !coeff), L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(M > 0 && N > 0, L2Q_EINVAL, "non-positive size");
  hipStream_t st = (hipStream_t)stream;
  const long total = (long)M * N;
  L2Q_DISPATCH_T(elem_bytes,
                 hipLaunchKernelGGL(scaled_tanh_bwd_kernel<T>, dim3(grid1(total)), dim3(kBlock), 0,
                                    st, (const T*)ds, (const T*)s, (const T*)coeff, (T)scale, N,
                                    total, (T*)dpre));
  return check_launch("l2q_scaled_tanh_bwd");
}

int l2q_scaled_tanh_bwd_sums(const void* ds, const void* s, const void* coeff, double scale, int M, int N,
                             int elem_bytes, void* dpre, void* bgrad, void* cgrad, void* ws, size_t ws_bytes,
                             void* stream) {
  L2Q_REQUIRE(ds && dpre && bgrad && ws && (s || !coeff) && (cgrad || !coeff), L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(M > 0 && N > 0, L2Q_EINVAL, "non-positive size");
  L2Q_REQUIRE(ws_bytes >= 2 * l2q_colsum_ws_bytes(M, N), L2Q_EINVAL, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const long rpb = colsum_rows_per_block(M);
  const long R = cdiv(M, rpb);
  int cx = 64;
  while (cx > 8 && cx / 2 >= N) cx /= 2;
  const dim3 blk(cx, 1024 / cx), grid((unsigned)cdiv(N, cx), (unsigned)R);
  double* pb = (double*)ws;
  double* pc = pb + (size_t)R * N;
  L2Q_DISPATCH_T(elem_bytes, {
    hipLaunchKernelGGL(scaled_tanh_bwd_sums_kernel<T>, grid, blk, 0, st, (const T*)ds, (const T*)s,
                       (const T*)coeff, (T)scale, (long)M, N, rpb, (T*)dpre, pb, pc);
    hipLaunchKernelGGL(colsum_final_kernel<T>, dim3(grid1(N, 64)), dim3(kBlock), 0, st, (const double*)pb, R, N,
                       1.0, 1, (T*)bgrad);
    if (coeff)
      hipLaunchKernelGGL(colsum_final_kernel<T>, dim3(grid1(N, 64)), dim3(kBlock), 0, st, (const double*)pc, R,
                         N, 1.0, 1, (T*)cgrad);
  });
  return check_launch("l2q_scaled_tanh_bwd_sums");
}

int l2q_bn_train_fwd(const void* x, const void* gamma, const void* beta, double eps,
                     double momentum, void* running_mean, void* running_var, int M, int N,
                     int elem_bytes, void* y, void* save_mean, void* save_invstd, void* stream) {
  L2Q_REQUIRE(x && gamma && beta && y && save_mean && save_invstd, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE((running_mean == nullptr) == (running_var == nullptr), L2Q_EINVAL,
              "running_mean / running_var must both be given or both be NULL");
  L2Q_REQUIRE(M > 0 && N > 0, L2Q_EINVAL, "non-positive size");
  hipStream_t st = (hipStream_t)stream;
  L2Q_DISPATCH_T(elem_bytes,
                 hipLaunchKernelGGL(bn_train_fwd_kernel<T>, dim3(N), dim3(kBlock), 0, st,
                                    (const T*)x, (const T*)gamma, (const T*)beta, eps, momentum,
                                    (T*)running_mean, (T*)running_var, M, N, (T*)y, (T*)save_mean,
                                    (T*)save_invstd));
  return check_launch("l2q_bn_train_fwd");
}

int l2q_bn_bwd(const void* dy, const void* x, const void* save_mean, const void* save_invstd,
               const void* gamma, int M, int N, int elem_bytes, void* dx, void* dgamma,
               void* dbeta, void* stream) {
  L2Q_REQUIRE(dy && x && save_mean && save_invstd && gamma && dx && dgamma && dbeta, L2Q_EINVAL,
              "null pointer");
  L2Q_REQUIRE(M > 0 && N > 0, L2Q_EINVAL, "non-positive size");
  hipStream_t st = (hipStream_t)stream;
  L2Q_DISPATCH_T(elem_bytes,
                 hipLaunchKernelGGL(bn_bwd_kernel<T>, dim3(N), dim3(kBlock), 0, st, (const T*)dy,
                                    (const T*)x, (const T*)save_mean, (const T*)save_invstd,
                                    (const T*)gamma, M, N, (T*)dx, (T*)dgamma, (T*)dbeta));
  return check_launch("l2q_bn_bwd");
}

int l2q_col2im_periodic_f32(const float* dcol, long sn, long sc, long sh, long sw, int nb, int C,
                            int H, int W, int k, int channels_last_cols, float* dx,
                            void* stream) {
  L2Q_REQUIRE(dcol && dx, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && C > 0 && H > 0 && W > 0 && k > 0, L2Q_EINVAL, "non-positive size");
  const int Ho = H + k - 1, Wo = W + k - 1, Kc = C * k * k;
  const long total = (long)nb * C * H * W;
  L2Q_REQUIRE(cdiv(total, kBlock) < 0x7fffffffL, L2Q_ESHAPE, "grid too large");
  hipLaunchKernelGGL(col2im_periodic_kernel, dim3(grid1(total)), dim3(kBlock), 0,
                     (hipStream_t)stream, dcol, sn, sc, sh, sw, C, H, W, k, Ho, Wo, Kc, total,
                     channels_last_cols ? 1 : 0, dx);
  return check_launch("l2q_col2im_periodic_f32");
}

int l2q_maxpool_act_nhwc_bwd_f32(const float* dout, const float* out, const float* in, int nb,
                                 int H, int W, int C, int pool, int act, float* din,
                                 void* stream) {
  L2Q_REQUIRE(dout && out && in && din, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && H > 0 && W > 0 && C > 0 && pool > 0, L2Q_EINVAL, "non-positive size");
  L2Q_REQUIRE(act != L2Q_ACT_SWISH, L2Q_EINVAL, "swish is not supported by the training path");
  const int Ho = H / pool, Wo = W / pool;
  L2Q_REQUIRE(Ho > 0 && Wo > 0, L2Q_ESHAPE, "pooling window larger than the image");
  const long total = (long)nb * H * W * C;
  hipLaunchKernelGGL(maxpool_act_nhwc_bwd_kernel, dim3(grid1(total)), dim3(kBlock), 0,
                     (hipStream_t)stream, dout, out, in, H, W, C, pool, act, Ho, Wo, total, din);
  return check_launch("l2q_maxpool_act_nhwc_bwd_f32");
}

int l2q_u1_force_bwd(const void* x, const void* dF, double beta, int nb, int T_, int X_,
                     int elem_bytes, void* dx, void* stream) {
  L2Q_REQUIRE(x && dF && dx, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && T_ > 0 && X_ > 0, L2Q_EINVAL, "non-positive size");
  L2Q_REQUIRE((long)T_ * X_ <= kU1BwdMaxLds, L2Q_ESHAPE, "lattice larger than 8192 sites");
  hipStream_t st = (hipStream_t)stream;
  L2Q_DISPATCH_T(elem_bytes,
                 hipLaunchKernelGGL((u1_stencil_bwd_kernel<T, true>), dim3(nb), dim3(kBlock), 0, st,
                                    (const T*)x, (const T*)dF, (T)beta, (const T*)nullptr,
                                    (const T*)nullptr, T_, X_, (T*)dx));
  return check_launch("l2q_u1_force_bwd");
}

int l2q_u1_plaq_bwd(const void* x, const void* gcos, const void* gsin, int nb, int T_, int X_,
                    int elem_bytes, void* dx, void* stream) {
  L2Q_REQUIRE(x && dx && (gcos || gsin), L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && T_ > 0 && X_ > 0, L2Q_EINVAL, "non-positive size");
  L2Q_REQUIRE((long)T_ * X_ <= kU1BwdMaxLds, L2Q_ESHAPE, "lattice larger than 8192 sites");
  hipStream_t st = (hipStream_t)stream;
  L2Q_DISPATCH_T(elem_bytes,
                 hipLaunchKernelGGL((u1_stencil_bwd_kernel<T, false>), dim3(nb), dim3(kBlock), 0,
                                    st, (const T*)x, (const T*)nullptr, (T)0, (const T*)gcos,
                                    (const T*)gsin, T_, X_, (T*)dx));
  return check_launch("l2q_u1_plaq_bwd");
}

int l2q_u1_x_update_bwd(const void* x, const void* v, const void* s, const void* t, const void* q,
                        const float* mask, int complement, double eps, int forward, int use_ncp,
                        const void* gx, const void* gl, int elem_bytes, int nb, long n, void* dx,
                        void* dv, void* ds, void* dt, void* dq, void* deps, void* stream) {
  L2Q_REQUIRE(x && v && s && t && q && mask && gx && dx && dv && ds && dt && dq && deps,
              L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && n > 0, L2Q_EINVAL, "non-positive size");
  hipStream_t st = (hipStream_t)stream;
#define L2Q_XB(F, N)                                                                            \
  hipLaunchKernelGGL((u1_x_update_bwd_kernel<T, F, N>), dim3(nb), dim3(kBlock), 0, st,          \
                     (const T*)x, (const T*)v, (const T*)s, (const T*)t, (const T*)q, mask,     \
                     complement, (T)eps, (const T*)gx, (const T*)gl, n, (T*)dx, (T*)dv, (T*)ds, \
                     (T*)dt, (T*)dq, (T*)deps)
  L2Q_DISPATCH_T(elem_bytes, {
    if (forward) { if (use_ncp) L2Q_XB(true, true); else L2Q_XB(true, false); }
    else { if (use_ncp) L2Q_XB(false, true); else L2Q_XB(false, false); }
  });
#undef L2Q_XB
  return check_launch("l2q_u1_x_update_bwd");
}

int l2q_v_update_bwd(const void* v, const void* force, const void* s, const void* t,
                     const void* q, double eps, int forward, const void* gv, const void* gl,
                     int elem_bytes, int nb, long n, void* dv, void* dF, void* ds, void* dt,
                     void* dq, void* deps, void* stream) {
  L2Q_REQUIRE(v && force && s && t && q && gv && dv && dF && ds && dt && dq && deps, L2Q_EINVAL,
              "null pointer");
  L2Q_REQUIRE(nb > 0 && n > 0, L2Q_EINVAL, "non-positive size");
  hipStream_t st = (hipStream_t)stream;
#define L2Q_VB(F)                                                                              \
  hipLaunchKernelGGL((v_update_bwd_kernel<T, F>), dim3(nb), dim3(kBlock), 0, st, (const T*)v,  \
                     (const T*)force, (const T*)s, (const T*)t, (const T*)q, (T)eps,           \
                     (const T*)gv, (const T*)gl, n, (T*)dv, (T*)dF, (T*)ds, (T*)dt, (T*)dq,    \
                     (T*)deps)
  L2Q_DISPATCH_T(elem_bytes, { if (forward) L2Q_VB(true); else L2Q_VB(false); });
#undef L2Q_VB
  return check_launch("l2q_v_update_bwd");
}

int l2q_u1_masked_cos_sin_bwd(const void* x, const float* mask, int complement, const void* dout,
                              int nb, long n, int elem_bytes, void* dx, void* stream) {
  L2Q_REQUIRE(x && mask && dout && dx, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && n > 0, L2Q_EINVAL, "non-positive size");
  hipStream_t st = (hipStream_t)stream;
  L2Q_DISPATCH_T(elem_bytes,
                 hipLaunchKernelGGL(u1_masked_cos_sin_bwd_kernel<T>, dim3(grid1(n), (unsigned)nb),
                                    dim3(kBlock), 0, st, (const T*)x, mask, complement,
                                    (const T*)dout, n, (T*)dx));
  return check_launch("l2q_u1_masked_cos_sin_bwd");
}

int l2q_adam(void* p, const void* g, void* m, void* v, long n, double lr, double beta1,
             double beta2, double eps, long step, double grad_scale, int elem_bytes,
             void* stream) {
  L2Q_REQUIRE(p && g && m && v, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(n > 0 && step > 0, L2Q_EINVAL, "non-positive size / step");
  hipStream_t st = (hipStream_t)stream;
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double sbc2 = sqrt(1.0 - pow(beta2, (double)step));
  L2Q_DISPATCH_T(elem_bytes,
                 hipLaunchKernelGGL(adam_kernel<T>, dim3(grid1(n)), dim3(kBlock), 0, st, (T*)p,
                                    (const T*)g, (T*)m, (T*)v, n, (T)lr, (T)beta1, (T)beta2,
                                    (T)eps, (T)bc1, (T)sbc2, (T)grad_scale));
  return check_launch("l2q_adam");
}

size_t l2q_sumsq_ws_bytes(long n) {
  long nblk = cdiv(n, (long)kBlock * 8);
  if (nblk > 1024) nblk = 1024;
  if (nblk < 1) nblk = 1;
  return (size_t)nblk * sizeof(double);
}

int l2q_sumsq(const void* a, long n, int elem_bytes, double* out, void* ws, size_t ws_bytes,
              void* stream) {
  L2Q_REQUIRE(a && out && ws, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(n > 0, L2Q_EINVAL, "non-positive size");
  L2Q_REQUIRE(ws_bytes >= l2q_sumsq_ws_bytes(n), L2Q_EINVAL, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const long nblk = (long)(l2q_sumsq_ws_bytes(n) / sizeof(double));
  L2Q_DISPATCH_T(elem_bytes,
                 hipLaunchKernelGGL(sumsq_partial_kernel<T>, dim3((unsigned)nblk), dim3(kBlock), 0,
                                    st, (const T*)a, n, (double*)ws));
  launch_finalize((const double*)ws, out, 1, nblk, 1, 1.0, 0.0, st);
  return check_launch("l2q_sumsq");
}

}  // extern "C"
