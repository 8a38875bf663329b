// su3_links.hpp -- link addressing in the native layout xn[chain][mu][e][site] and periodic
// neighbour arithmetic, shared by the SU(3) forward and training kernels.
#pragma once
#include "l2q_common.hpp"
#include "su3_math.hpp"

namespace l2q {

struct Dims {
  int T, X, Y, Z;
  int V;          // sites per chain (36 V complex entries per chain must fit an int)
};

__device__ __forceinline__ void load_link(M3& m, const double2* __restrict__ f, int V, int s) {
#pragma unroll
  for (int e = 0; e < 9; ++e) {
    const double2 d = f[e * V + s];
    m.re[e] = d.x; m.im[e] = d.y;
  }
}

__device__ __forceinline__ void store_link(double2* __restrict__ f, int V, int s, const M3& m) {
#pragma unroll
  for (int e = 0; e < 9; ++e) f[e * V + s] = make_double2(m.re[e], m.im[e]);
}

struct Site {
  int t, x, y, z;
};

__device__ __forceinline__ Site site_coords(int s, const Dims& d) {
  Site r;
  r.z = s % d.Z; s /= d.Z;
  r.y = s % d.Y; s /= d.Y;
  r.x = s % d.X; s /= d.X;
  r.t = s;
  return r;
}

// mu is wave-uniform everywhere below (loop counters / blockIdx), so these selects are
// scalar and nothing is indexed dynamically in VGPR arrays.
__device__ __forceinline__ int coord_of(const Site& p, int mu) {
  return mu == 0 ? p.t : mu == 1 ? p.x : mu == 2 ? p.y : p.z;
}
__device__ __forceinline__ int stride_of(const Dims& d, int mu) {
  return mu == 0 ? d.X * d.Y * d.Z : mu == 1 ? d.Y * d.Z : mu == 2 ? d.Z : 1;
}
__device__ __forceinline__ int extent_of(const Dims& d, int mu) {
  return mu == 0 ? d.T : mu == 1 ? d.X : mu == 2 ? d.Y : d.Z;
}
// site index of the periodic forward / backward neighbour of s (coordinate cm in direction mu)
__device__ __forceinline__ int fwd(int s, int cm, const Dims& d, int mu) {
  const int st = stride_of(d, mu), n = extent_of(d, mu);
  return (cm + 1 == n) ? s - (n - 1) * st : s + st;
}
__device__ __forceinline__ int bwd(int s, int cm, const Dims& d, int mu) {
  const int st = stride_of(d, mu), n = extent_of(d, mu);
  return (cm == 0) ? s + (n - 1) * st : s - st;
}

}  // namespace l2q
