// su3_rect_kernels.hip -- the 2x1 rectangle term of the improved gauge actions (c1 != 0: Iwasaki,
// DBW2, ...), reference lattice/su3/pytorch/lattice.py:83-112 (coeffs, _rectangles) and
// :180-196, 252-269 (the rectangle traces inside _wilson_loops and their use in action):
//
//   S = -(1/3) [ beta (1 - 8 c1) sum_P Re tr P  +  beta c1 sum_R Re tr R ]
//
// R runs over both orientations of the planar 2x1 loop in all six planes: 12 loops per site.
// A link U_mu(x) lies on 18 of them (per nu != mu: four with the long side along mu, two with the
// long side along nu).  Everything here follows a lattice path link by link, so each kernel is a
// table of hop sequences plus one product routine; the loops over loops are not unrolled (three
// live matrices, as in the plaquette kernels).
#include "su3_links.hpp"

namespace l2q {

namespace {

struct Walker {
  int s;
  int t, x, y, z;
};

__device__ __forceinline__ int get_c(const Walker& w, int mu) {
  return mu == 0 ? w.t : mu == 1 ? w.x : mu == 2 ? w.y : w.z;
}
__device__ __forceinline__ void set_c(Walker& w, int mu, int v) {
  if (mu == 0) w.t = v; else if (mu == 1) w.x = v; else if (mu == 2) w.y = v; else w.z = v;
}
__device__ __forceinline__ void hop_fwd(Walker& w, const Dims& d, int mu) {
  const int c = get_c(w, mu), n = extent_of(d, mu), st = stride_of(d, mu);
  if (c + 1 == n) { w.s -= (n - 1) * st; set_c(w, mu, 0); }
  else { w.s += st; set_c(w, mu, c + 1); }
}
__device__ __forceinline__ void hop_bwd(Walker& w, const Dims& d, int mu) {
  const int c = get_c(w, mu), n = extent_of(d, mu), st = stride_of(d, mu);
  if (c == 0) { w.s += (n - 1) * st; set_c(w, mu, n - 1); }
  else { w.s -= st; set_c(w, mu, c - 1); }
}

// Ordered product of the links along a path of `len` hops from w.  Step code +1 / -1: one hop
// along +mu / -mu, +2 / -2: along +nu / -nu.  A forward hop multiplies by U_dir(site) and then
// moves, a backward hop moves first and multiplies by U_dir(site)^H.
__device__ __forceinline__ void path_product(M3& out, const double2* __restrict__ xc, const Dims& d,
                                             Walker w, int mu, int nu, const signed char* steps,
                                             int len) {
  const int V = d.V;
  M3 p, a;
  {
    const int st = steps[0];
    const int dir = (st == 1 || st == -1) ? mu : nu;
    if (st > 0) { load_link(p, xc + dir * 9 * V, V, w.s); hop_fwd(w, d, dir); }
    else {
      hop_bwd(w, d, dir);
      load_link(a, xc + dir * 9 * V, V, w.s);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) { p.re[3 * i + j] = a.re[3 * j + i]; p.im[3 * i + j] = -a.im[3 * j + i]; }
    }
  }
#pragma unroll 1
  for (int k = 1; k < len; ++k) {
    const int st = steps[k];
    const int dir = (st == 1 || st == -1) ? mu : nu;
    M3 t;
    if (st > 0) {
      load_link(a, xc + dir * 9 * V, V, w.s);
      hop_fwd(w, d, dir);
      m3_mul_nn(t, p, a);
    } else {
      hop_bwd(w, d, dir);
      load_link(a, xc + dir * 9 * V, V, w.s);
      m3_mul_na(t, p, a);
    }
    p = t;
  }
  out = p;
}

// rest-of-loop paths ("staples") of the 6 rectangles through U_mu(x) in the (mu, nu) plane,
// starting at x + mu and ending at x
__constant__ signed char kRectStaple[6][5] = {
    {+1, +2, -1, -1, -2},   // long side along mu, U is its first link, nu up
    {+2, -1, -1, -2, +1},   // long side along mu, U is its second link, nu up
    {+1, -2, -1, -1, +2},   // first link, nu down
    {-2, -1, -1, +2, +1},   // second link, nu down
    {+2, +2, -1, -2, -2},   // long side along nu, up
    {-2, -2, -1, +2, +2},   // long side along nu, down
};
__constant__ signed char kRectHalfA[3] = {+1, +1, +2};   // U_mu(x) U_mu(x+mu) U_nu(x+2mu)
__constant__ signed char kRectHalfB[3] = {+2, +1, +1};   // U_nu(x) U_mu(x+nu) U_mu(x+mu+nu)

}  // namespace

// out[c] = sum_x sum_{mu != nu} Re tr [ U_mu(x) U_mu(x+mu) U_nu(x+2mu) (U_nu(x) U_mu(x+nu) U_mu(x+mu+nu))^H ]
// (= rs.real.sum() of lattice.py:262: the reference's two traces per plane are the Hermitian
// conjugates of these loops, same real part)
__global__ __launch_bounds__(kBlock, 2) void su3_rect_reduce_kernel(const double2* __restrict__ xn,
                                                                    Dims d, long nblk,
                                                                    double* __restrict__ partial) {
  __shared__ double lds[4];
  const long c = blockIdx.x / nblk, blk = blockIdx.x % nblk;
  const int s = (int)blk * kBlock + threadIdx.x;
  double sr = 0.0;
  if (s < d.V) {
    const double2* xc = xn + c * 36L * d.V;
    const Site p = site_coords(s, d);
    const Walker w0{s, p.t, p.x, p.y, p.z};
#pragma unroll 1
    for (int mu = 0; mu < 4; ++mu)
#pragma unroll 1
      for (int nu = 0; nu < 4; ++nu) {
        if (nu == mu) continue;
        M3 a, b;
        path_product(a, xc, d, w0, mu, nu, kRectHalfA, 3);
        path_product(b, xc, d, w0, mu, nu, kRectHalfB, 3);
        double tr, ti;
        m3_trace_mul_na(tr, ti, a, b);
        sr += tr;
      }
  }
  const double r = block_sum(sr, lds);
  if (threadIdx.x == 0) partial[c * nblk + blk] = r;
}

// A = sum over the 18 rectangle staples of link (x, mu).
// MODE 0: out[c][mu] += coef * TAH(U_mu(x) A)      (rectangle part of grad_action, lattice.py:299-308)
// MODE 1: out[c][mu] += w[c] * A^H                 (cotangent of w[c] * sum_R Re tr R)
template <int MODE>
__global__ __launch_bounds__(kBlock, 2) void su3_rect_staple_kernel(const double2* __restrict__ xn,
                                                                    Dims d, long nblk, double coef,
                                                                    const double* __restrict__ wgt,
                                                                    double2* __restrict__ out) {
  const long w = blockIdx.x;
  const int mu = (int)(w & 3);
  const long cb = w >> 2;
  const long c = cb / nblk, blk = cb % nblk;
  const int s = (int)blk * kBlock + threadIdx.x;
  if (s >= d.V) return;
  const int V = d.V;
  const double2* xc = xn + c * 36L * V;
  const Site p = site_coords(s, d);
  Walker w1{s, p.t, p.x, p.y, p.z};
  hop_fwd(w1, d, mu);                                  // every staple starts at x + mu
  M3 acc;
  m3_zero(acc);
#pragma unroll 1
  for (int nu = 0; nu < 4; ++nu) {
    if (nu == mu) continue;
#pragma unroll 1
    for (int k = 0; k < 6; ++k) {
      M3 st;
      path_product(st, xc, d, w1, mu, nu, kRectStaple[k], 5);
      m3_add(acc, st);
    }
  }
  double2* o = out + (c * 4 + mu) * 9L * V;
  if (MODE == 0) {
    M3 u, ua, f;
    load_link(u, xc + mu * 9 * V, V, s);
    m3_mul_nn(ua, u, acc);
    m3_tah(f, ua);
#pragma unroll
    for (int e = 0; e < 9; ++e) {
      double2 r = o[e * V + s];
      r.x += coef * f.re[e]; r.y += coef * f.im[e];
      o[e * V + s] = r;
    }
  } else {
    const double wc = wgt[c];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double2 r = o[(3 * i + j) * V + s];
        r.x += wc * acc.re[3 * j + i]; r.y -= wc * acc.im[3 * j + i];
        o[(3 * i + j) * V + s] = r;
      }
  }
}

static bool rect_dims_ok(int nb, int T, int X, int Y, int Z) {
  return nb > 0 && T > 0 && X > 0 && Y > 0 && Z > 0 && (long)T * X * Y * Z * 36 < 2000000000L;
}

}  // namespace l2q

using namespace l2q;

extern "C" {

int l2q_su3_rect_reduce(const void* xn, int nb, int T, int X, int Y, int Z, double* out, void* ws,
                        size_t ws_bytes, void* stream) {
  L2Q_REQUIRE(xn && out && ws, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(rect_dims_ok(nb, T, X, Y, Z), L2Q_EINVAL, "bad size");
  Dims d{T, X, Y, Z, T * X * Y * Z};
  const long nblk = cdiv(d.V, kBlock);
  L2Q_REQUIRE(ws_bytes >= (size_t)nb * nblk * sizeof(double), L2Q_ESHAPE, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(su3_rect_reduce_kernel, dim3((unsigned)(nb * nblk)), dim3(kBlock), 0, st,
                     (const double2*)xn, d, nblk, (double*)ws);
  launch_finalize((const double*)ws, out, nb, nblk, 1, 1.0, 0.0, st);
  return check_launch("l2q_su3_rect_reduce");
}

int l2q_su3_rect_force_add(const void* xn, double coef, void* fn, int nb, int T, int X, int Y, int Z,
                           void* stream) {
  L2Q_REQUIRE(xn && fn, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(rect_dims_ok(nb, T, X, Y, Z), L2Q_EINVAL, "bad size");
  Dims d{T, X, Y, Z, T * X * Y * Z};
  const long nblk = cdiv(d.V, kBlock);
  hipLaunchKernelGGL(su3_rect_staple_kernel<0>, dim3((unsigned)(nb * nblk * 4)), dim3(kBlock), 0,
                     (hipStream_t)stream, (const double2*)xn, d, nblk, coef, (const double*)nullptr,
                     (double2*)fn);
  return check_launch("l2q_su3_rect_force_add");
}

int l2q_su3_rect_bwd(const void* xn, const double* w, void* gx, int nb, int T, int X, int Y, int Z,
                     void* stream) {
  L2Q_REQUIRE(xn && w && gx, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(rect_dims_ok(nb, T, X, Y, Z), L2Q_EINVAL, "bad size");
  Dims d{T, X, Y, Z, T * X * Y * Z};
  const long nblk = cdiv(d.V, kBlock);
  hipLaunchKernelGGL(su3_rect_staple_kernel<1>, dim3((unsigned)(nb * nblk * 4)), dim3(kBlock), 0,
                     (hipStream_t)stream, (const double2*)xn, d, nblk, 0.0, w, (double2*)gx);
  return check_launch("l2q_su3_rect_bwd");
}

}  // extern "C"
