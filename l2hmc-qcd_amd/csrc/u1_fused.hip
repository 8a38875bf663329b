// u1_fused.hip -- one kernel per L2HMC sub-update of the 2D U(1) sampler (eval mode, fp32,
// dense networks): force (v-step) or masked cos/sin (x-step), the whole LeapfrogLayer
// (input layer, hidden layers, the three heads) and the momentum / position update with its
// log-det reduction.  s, t, q and the hidden activations never leave the CU.
//
// Why: on 8x8 .. 16x16 lattices a sub-update is ~12 launches of a few microseconds each
// (GEMMs with N = 16, element-wise kernels); the work itself is ~80 kFLOP and 10 KB per
// chain.  A workgroup takes CH chains (CH * 3n floats of LDS), walks the layers with the
// weights streamed from L2 (they are shared by every workgroup) and applies the update.
// Replaces, per call: l2q_u1_force / l2q_u1_masked_cos_sin, 5-8 x l2q_gemm_f32 and
// l2q_v_update / l2q_u1_x_update (dynamics.py:1142-1185, 1266-1297, 1386-1477;
// network.py:430-451, 522-551).
#include "l2q_common.hpp"
#include "u1_math.hpp"

namespace l2q {

constexpr int kMaxLayers = 8;     // widths of the dense stack (input layer + hidden layers)
constexpr int kMaxWidth = 64;

struct U1Net {
  const float* wxT;      // [Kx][U0]  (transposed xlayer weight)
  const float* wvT;      // [Kv][U0]
  const float* b0;       // [U0]      (xlayer.bias + vlayer.bias)
  const float* hidden;   // for l = 1 .. nl-1: W_l [U_l][U_{l-1}], b_l [U_l], packed
  const float* ws; const float* bs; const float* cs;   // heads [n][UL], [n], per-column scale
  const float* wt; const float* bt;
  const float* wq; const float* bq; const float* cq;
  float scale_t;
  int nl;
  int units[kMaxLayers];
  int act;
};

// dense stack on CH chains whose inputs sit in LDS: xin[CH][Kx], vin[CH][Kv] -> z[CH][UL]
template <int CH>
__device__ __forceinline__ void u1_dense_stack(const U1Net& net, const float* xin, int Kx,
                                               const float* vin, int Kv, float* part, float* z0,
                                               float* z1) {
  const int U0 = net.units[0];
  int up = 1;
  while (up < U0) up <<= 1;                         // threads = up (unit) x S (K slices)
  const int S = kBlock / up;
  const int u = threadIdx.x % up, sl = threadIdx.x / up;
  float acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) acc[c] = 0.0f;
  if (u < U0) {
    int k = sl;
    for (; k + 3 * S < Kx; k += 4 * S) {             // four weight loads in flight
      const float w0 = net.wxT[(long)k * U0 + u], w1 = net.wxT[(long)(k + S) * U0 + u];
      const float w2 = net.wxT[(long)(k + 2 * S) * U0 + u], w3 = net.wxT[(long)(k + 3 * S) * U0 + u];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const float* xc = xin + c * Kx + k;
        acc[c] = fmaf(w0, xc[0], fmaf(w1, xc[S], fmaf(w2, xc[2 * S], fmaf(w3, xc[3 * S], acc[c]))));
      }
    }
    for (; k < Kx; k += S) {
      const float w = net.wxT[(long)k * U0 + u];
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = fmaf(w, xin[c * Kx + k], acc[c]);
    }
    k = sl;
    for (; k + 3 * S < Kv; k += 4 * S) {
      const float w0 = net.wvT[(long)k * U0 + u], w1 = net.wvT[(long)(k + S) * U0 + u];
      const float w2 = net.wvT[(long)(k + 2 * S) * U0 + u], w3 = net.wvT[(long)(k + 3 * S) * U0 + u];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const float* vc = vin + c * Kv + k;
        acc[c] = fmaf(w0, vc[0], fmaf(w1, vc[S], fmaf(w2, vc[2 * S], fmaf(w3, vc[3 * S], acc[c]))));
      }
    }
    for (; k < Kv; k += S) {
      const float w = net.wvT[(long)k * U0 + u];
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = fmaf(w, vin[c * Kv + k], acc[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < CH; ++c) part[(sl * CH + c) * up + u] = acc[c];
  __syncthreads();
  for (int i = threadIdx.x; i < CH * U0; i += kBlock) {
    const int c = i / U0, uu = i % U0;
    float s = net.b0[uu];
    for (int k = 0; k < S; ++k) s += part[(k * CH + c) * up + uu];
    z0[c * kMaxWidth + uu] = act_f32(s, net.act);
  }
  __syncthreads();
  const float* hw = net.hidden;
  float* zin = z0;
  float* zout = z1;
  for (int l = 1; l < net.nl; ++l) {
    const int Ui = net.units[l - 1], Uo = net.units[l];
    for (int i = threadIdx.x; i < CH * Uo; i += kBlock) {
      const int c = i / Uo, uu = i % Uo;
      float s = hw[Uo * Ui + uu];
      for (int k = 0; k < Ui; ++k) s = fmaf(hw[uu * Ui + k], zin[c * kMaxWidth + k], s);
      zout[c * kMaxWidth + uu] = act_f32(s, net.act);
    }
    __syncthreads();
    hw += Uo * Ui + Uo;
    float* tmp = zin; zin = zout; zout = tmp;
  }
  if (zin != z0) {                                   // result always in z0
    for (int i = threadIdx.x; i < CH * kMaxWidth; i += kBlock) z0[i] = zin[i];
    __syncthreads();
  }
}

// pre-activations of the three heads of entry j for all CH chains: the weight rows are read
// once (k outer), the chain loop is unrolled so the accumulators stay in registers
template <int CH>
__device__ __forceinline__ void u1_head_pre(const U1Net& net, int UL, const float* z0, int j,
                                            float (&as)[CH], float (&at)[CH], float (&aq)[CH]) {
  const float bs = net.bs[j], bt = net.bt[j], bq = net.bq[j];
#pragma unroll
  for (int c = 0; c < CH; ++c) { as[c] = bs; at[c] = bt; aq[c] = bq; }
  const float* rs = net.ws + (long)j * UL;
  const float* rt = net.wt + (long)j * UL;
  const float* rq = net.wq + (long)j * UL;
  if ((UL & 3) == 0) {                               // rows are 16-byte aligned: 128-bit loads
    for (int k = 0; k < UL; k += 4) {
      const float4 w_s = *reinterpret_cast<const float4*>(rs + k);
      const float4 w_t = *reinterpret_cast<const float4*>(rt + k);
      const float4 w_q = *reinterpret_cast<const float4*>(rq + k);
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const float4 zz = *reinterpret_cast<const float4*>(z0 + c * kMaxWidth + k);
        as[c] = fmaf(w_s.x, zz.x, fmaf(w_s.y, zz.y, fmaf(w_s.z, zz.z, fmaf(w_s.w, zz.w, as[c]))));
        at[c] = fmaf(w_t.x, zz.x, fmaf(w_t.y, zz.y, fmaf(w_t.z, zz.z, fmaf(w_t.w, zz.w, at[c]))));
        aq[c] = fmaf(w_q.x, zz.x, fmaf(w_q.y, zz.y, fmaf(w_q.z, zz.z, fmaf(w_q.w, zz.w, aq[c]))));
      }
    }
    return;
  }
  for (int k = 0; k < UL; ++k) {
    const float w_s = rs[k], w_t = rt[k], w_q = rq[k];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const float zz = z0[c * kMaxWidth + k];
      as[c] = fmaf(w_s, zz, as[c]); at[c] = fmaf(w_t, zz, at[c]); aq[c] = fmaf(w_q, zz, aq[c]);
    }
  }
}

// ---- v sub-update: F = force(x); (s,t,q) = vnet(x, F); v' as l2q_v_update.  x is not changed.
template <int CH, bool FWD>
__global__ __launch_bounds__(kBlock) void u1_vstep_kernel(const float* __restrict__ x,
                                                          float* v, U1Net net, float beta,
                                                          float eps, int Tn, int Xn, int nb,
                                                          int accumulate, float* logdet) {
  extern __shared__ float sm[];
  const int V = Tn * Xn, n = 2 * V;
  float* xin = sm;                     // [CH][n]
  float* fin = xin + CH * n;           // [CH][n]  the force (second network input)
  float* sn = fin + CH * n;            // [CH][V]  sin(theta)
  float* part = sn + CH * V;           // [S][CH][up] <= 256 * CH floats
  float* z0 = part + kBlock * CH;      // [CH][64]
  float* z1 = z0 + CH * kMaxWidth;
  __shared__ double red[4];
  const int c0 = blockIdx.x * CH;
  for (int i = threadIdx.x; i < CH * n; i += kBlock) {
    const int c = i / n;
    xin[i] = (c0 + c < nb) ? x[(long)(c0 + c) * n + (i - c * n)] : 0.0f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < CH * V; i += kBlock) {
    const int c = i / V, s = i % V;
    sn[i] = sinf(plaq_angle(xin + c * n, s / Xn, s % Xn, Tn, Xn));
  }
  __syncthreads();
  // F0 = beta [sin th - sin th(t, x-1)],  F1 = beta [-sin th + sin th(t-1, x)]
  for (int i = threadIdx.x; i < CH * V; i += kBlock) {
    const int c = i / V, s = i % V, t = s / Xn, xx = s % Xn;
    const int tm = (t == 0) ? Tn - 1 : t - 1, xm = (xx == 0) ? Xn - 1 : xx - 1;
    const float* sc = sn + c * V;
    fin[c * n + s] = beta * (sc[s] - sc[t * Xn + xm]);
    fin[c * n + V + s] = beta * (-sc[s] + sc[tm * Xn + xx]);
  }
  __syncthreads();
  u1_dense_stack<CH>(net, xin, n, fin, n, part, z0, z1);
  const int UL = net.units[net.nl - 1];
  double ld[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) ld[c] = 0.0;
  for (int j = threadIdx.x; j < n; j += kBlock) {
    float as[CH], at[CH], aq[CH];
    u1_head_pre<CH>(net, UL, z0, j, as, at, aq);
    const float cs = net.cs[j], cq = net.cq[j];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (c0 + c >= nb) break;
      const float s = cs * tanhf(as[c]), t = net.scale_t * at[c], q = cq * tanhf(aq[c]);
      const long o = (long)(c0 + c) * n + j;
      const float half = 0.5f;
      const float lj = FWD ? (eps * s * half) : (-eps * s * half);
      ld[c] += (double)lj;
      const float es = expf(lj), eq = expf(eps * q);
      const float f = fin[c * n + j] * eq + t;
      const float vv = v[o];
      v[o] = FWD ? (es * vv - half * eps * f) : (es * (vv + half * eps * f));
    }
  }
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const double r = block_sum(ld[c], red);
    if (threadIdx.x == 0 && c0 + c < nb)
      logdet[c0 + c] = accumulate ? logdet[c0 + c] + (float)r : (float)r;
  }
}

// ---- x sub-update: (s,t,q) = xnet([cos(m x), sin(m x)], v); x' as l2q_u1_x_update
template <int CH, bool FWD, bool NCP>
__global__ __launch_bounds__(kBlock) void u1_xstep_kernel(float* x, const float* __restrict__ v,
                                                          U1Net net, const float* __restrict__ mask,
                                                          int complement, float eps, int n, int nb,
                                                          int accumulate, float* logdet) {
  extern __shared__ float sm[];
  float* xin = sm;                     // [CH][2n]: cos then sin of the kept entries
  float* vin = xin + CH * 2 * n;       // [CH][n]
  float* part = vin + CH * n;
  float* z0 = part + kBlock * CH;
  float* z1 = z0 + CH * kMaxWidth;
  __shared__ double red[4];
  const int c0 = blockIdx.x * CH;
  for (int i = threadIdx.x; i < CH * n; i += kBlock) {
    const int c = i / n, j = i - c * n;
    float keep = mask[j];
    if (complement) keep = 1.0f - keep;
    const bool live = c0 + c < nb;
    const float a = live ? keep * x[(long)(c0 + c) * n + j] : 0.0f;
    xin[c * 2 * n + j] = cosf(a);
    xin[c * 2 * n + n + j] = sinf(a);
    vin[c * n + j] = live ? v[(long)(c0 + c) * n + j] : 0.0f;
  }
  __syncthreads();
  u1_dense_stack<CH>(net, xin, 2 * n, vin, n, part, z0, z1);
  const int UL = net.units[net.nl - 1];
  double ld[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) ld[c] = 0.0;
  for (int j = threadIdx.x; j < n; j += kBlock) {
    float as[CH], at[CH], aq[CH];
    u1_head_pre<CH>(net, UL, z0, j, as, at, aq);
    const float cs = net.cs[j], cq = net.cq[j];
    float keep = mask[j];
    if (complement) keep = 1.0f - keep;
    const float mb = 1.0f - keep;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (c0 + c >= nb) break;
      const float s = cs * tanhf(as[c]), t = net.scale_t * at[c], q = cq * tanhf(aq[c]);
      const long o = (long)(c0 + c) * n + j;
      const float xj = x[o];
      const float sj = FWD ? eps * s : -eps * s;
      const float es = expf(sj), eq = expf(eps * q);
      const float tr = vin[c * n + j] * eq + t;
      float xp, l;
      if (NCP) {
        const float hx = xj * 0.5f;
        const float x1 = 2.0f * atanf(tanf(hx) * es);
        xp = FWD ? (x1 + eps * tr) : (x1 - es * eps * tr);
        const float ch = cosf(hx), sh = es * sinf(hx);
        l = logf(es / (ch * ch + sh * sh));
      } else {
        xp = FWD ? (xj * es + eps * tr) : (es * (xj - eps * tr));
        l = sj;
      }
      ld[c] += (double)(mb * l);
      x[o] = wrap_angle<float>(keep * xj + mb * xp);
    }
  }
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const double r = block_sum(ld[c], red);
    if (threadIdx.x == 0 && c0 + c < nb)
      logdet[c0 + c] = accumulate ? logdet[c0 + c] + (float)r : (float)r;
  }
}

}  // namespace l2q

using namespace l2q;

static int fill_net(U1Net& net, const float* wxT, const float* wvT, const float* b0,
                    const float* hidden, const int* units, int nl, const float* ws, const float* bs,
                    const float* cs, const float* wt, const float* bt, double scale_t,
                    const float* wq, const float* bq, const float* cq, int act) {
  if (!(wxT && wvT && b0 && units && ws && bs && cs && wt && bt && wq && bq && cq)) {
    set_error("l2q_u1_*step_f32: null pointer");
    return L2Q_EINVAL;
  }
  if (nl < 1 || nl > kMaxLayers || (nl > 1 && !hidden)) {
    set_error("l2q_u1_*step_f32: 1..%d dense layers supported", kMaxLayers);
    return L2Q_ESHAPE;
  }
  for (int l = 0; l < nl; ++l) {
    if (units[l] < 1 || units[l] > kMaxWidth) {
      set_error("l2q_u1_*step_f32: layer widths must be 1..%d", kMaxWidth);
      return L2Q_ESHAPE;
    }
    net.units[l] = units[l];
  }
  net.wxT = wxT; net.wvT = wvT; net.b0 = b0; net.hidden = hidden;
  net.ws = ws; net.bs = bs; net.cs = cs; net.wt = wt; net.bt = bt; net.wq = wq; net.bq = bq;
  net.cq = cq; net.scale_t = (float)scale_t; net.nl = nl; net.act = act;
  return L2Q_OK;
}

static int chains_per_block(long floats_per_chain, int nb) {
  // More chains per workgroup amortise the weight stream from L2, fewer give more workgroups
  // per CU to hide the barriers between the layers: take the largest CH that fits LDS
  // (two workgroups per CU) and still leaves >= 4 workgroups per CU.
  const int forced = tuning().u1_fused_ch;
  for (int ch = 8; ch >= 1; ch >>= 1) {
    const long bytes = (long)ch * (floats_per_chain + kBlock + 2 * kMaxWidth) * 4;
    if (bytes > 72 * 1024) continue;
    if (forced) { if (ch <= forced) return ch; continue; }
    if (ch == 1 || cdiv(nb, ch) >= 4 * 256) return ch;
  }
  return 0;
}

extern "C" {

int l2q_u1_fused_max_n(void) { return 2048; }

int l2q_u1_vstep_f32(const float* x, float* v, double beta, double eps, int forward, int nb, int T_,
                     int X_, const float* wxT, const float* wvT, const float* b0,
                     const float* hidden, const int* units, int nl, const float* ws,
                     const float* bs, const float* cs, const float* wt, const float* bt,
                     double scale_t, const float* wq, const float* bq, const float* cq, int act,
                     int accumulate, float* logdet, void* stream) {
  L2Q_REQUIRE(x && v && logdet, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && T_ > 0 && X_ > 0, L2Q_EINVAL, "non-positive size");
  const int n = 2 * T_ * X_;
  L2Q_REQUIRE(n <= l2q_u1_fused_max_n(), L2Q_ESHAPE, "lattice too large for the fused kernel");
  U1Net net;
  const int rc = fill_net(net, wxT, wvT, b0, hidden, units, nl, ws, bs, cs, wt, bt, scale_t, wq,
                          bq, cq, act);
  if (rc != L2Q_OK) return rc;
  const int ch = chains_per_block(2L * n + n / 2, nb);
  L2Q_REQUIRE(ch > 0, L2Q_ESHAPE, "does not fit LDS");
  const size_t lds = (size_t)ch * (2L * n + n / 2 + kBlock + 2 * kMaxWidth) * 4;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)cdiv(nb, ch)), block(kBlock);
#define L2Q_VS(CHN)                                                                            \
  do {                                                                                         \
    if (forward) hipLaunchKernelGGL((u1_vstep_kernel<CHN, true>), grid, block, lds, st, x, v,  \
                                    net, (float)beta, (float)eps, T_, X_, nb, accumulate, logdet); \
    else hipLaunchKernelGGL((u1_vstep_kernel<CHN, false>), grid, block, lds, st, x, v, net,    \
                            (float)beta, (float)eps, T_, X_, nb, accumulate, logdet);          \
  } while (0)
  switch (ch) {
    case 8: L2Q_VS(8); break;
    case 4: L2Q_VS(4); break;
    case 2: L2Q_VS(2); break;
    default: L2Q_VS(1); break;
  }
#undef L2Q_VS
  return check_launch("l2q_u1_vstep_f32");
}

int l2q_u1_xstep_f32(float* x, const float* v, const float* mask, int complement, double eps,
                     int forward, int use_ncp, int nb, int n, const float* wxT, const float* wvT,
                     const float* b0, const float* hidden, const int* units, int nl,
                     const float* ws, const float* bs, const float* cs, const float* wt,
                     const float* bt, double scale_t, const float* wq, const float* bq,
                     const float* cq, int act, int accumulate, float* logdet, void* stream) {
  L2Q_REQUIRE(x && v && mask && logdet, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && n > 0, L2Q_EINVAL, "non-positive size");
  L2Q_REQUIRE(n <= l2q_u1_fused_max_n(), L2Q_ESHAPE, "lattice too large for the fused kernel");
  U1Net net;
  const int rc = fill_net(net, wxT, wvT, b0, hidden, units, nl, ws, bs, cs, wt, bt, scale_t, wq,
                          bq, cq, act);
  if (rc != L2Q_OK) return rc;
  const int ch = chains_per_block(3L * n, nb);
  L2Q_REQUIRE(ch > 0, L2Q_ESHAPE, "does not fit LDS");
  const size_t lds = (size_t)ch * (3L * n + kBlock + 2 * kMaxWidth) * 4;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)cdiv(nb, ch)), block(kBlock);
#define L2Q_XS(CHN, F, N)                                                                      \
  hipLaunchKernelGGL((u1_xstep_kernel<CHN, F, N>), grid, block, lds, st, x, v, net, mask,      \
                     complement, (float)eps, n, nb, accumulate, logdet)
#define L2Q_XS4(CHN)                                                                           \
  do {                                                                                         \
    if (forward) { if (use_ncp) L2Q_XS(CHN, true, true); else L2Q_XS(CHN, true, false); }      \
    else { if (use_ncp) L2Q_XS(CHN, false, true); else L2Q_XS(CHN, false, false); }            \
  } while (0)
  switch (ch) {
    case 8: L2Q_XS4(8); break;
    case 4: L2Q_XS4(4); break;
    case 2: L2Q_XS4(2); break;
    default: L2Q_XS4(1); break;
  }
#undef L2Q_XS4
#undef L2Q_XS
  return check_launch("l2q_u1_xstep_f32");
}

}  // extern "C"
