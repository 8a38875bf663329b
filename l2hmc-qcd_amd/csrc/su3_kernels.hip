// su3_kernels.hip -- 4D SU(3) lattice kernels for gfx950 (MI355X).
//
// Native field layout xn[chain][mu][e][site] complex128: lane <-> site, so every load of a
// matrix entry is one coalesced 16 B/lane wavefront load; the 3x3 algebra lives in
// registers (su3_math.hpp).  Stencil neighbours are re-read through L1/L2; the 1-D grids
// are XCD-swizzled so the blocks that share a chain's links share an XCD's L2.
#include "l2q_common.hpp"
#include "su3_math.hpp"
#include "su3_links.hpp"

namespace l2q {

// ------------------------------------------------------------------ plaquette reduce
// sum over the 6 planes u > v of tr[ U_u(s) U_v(s+u) (U_v(s) U_u(s+v))^H ]
// (lattice/su3/pytorch/lattice.py:164-174).  The plane loop is deliberately NOT unrolled:
// it bounds the live ranges (<= 3 matrices) instead of letting the scheduler hoist 144 loads.
template <int OCC>
__global__ __launch_bounds__(kBlock, OCC) void su3_plaq_kernel(const double2* __restrict__ xn,
                                                               Dims d, long nblk, int swz,
                                                               double* __restrict__ partial) {
  __shared__ double lds[8];
  const long total = (long)gridDim.x;
  const long w = xcd_swizzle(blockIdx.x, total, swz);
  const long c = w / nblk, blk = w % nblk;
  const int s = (int)blk * kBlock + threadIdx.x;
  double sr = 0.0, si = 0.0;
  if (s < d.V) {
    const double2* xc = xn + c * 36L * d.V;
    const Site p = site_coords(s, d);
    const int V = d.V;
#pragma unroll 1
    for (int u = 1; u < 4; ++u) {
      const int s_pu = fwd(s, coord_of(p, u), d, u);
#pragma unroll 1
      for (int v = 0; v < u; ++v) {
        const int s_pv = fwd(s, coord_of(p, v), d, v);
        M3 a, b, yuv;
        load_link(a, xc + u * 9 * V, V, s);
        load_link(b, xc + v * 9 * V, V, s_pu);
        m3_mul_nn(yuv, a, b);
        load_link(a, xc + v * 9 * V, V, s);
        load_link(b, xc + u * 9 * V, V, s_pv);
        m3_trace_y_abh(sr, si, yuv, a, b);
      }
    }
  }
  const double br = block_sum(sr, lds);
  const double bi = block_sum(si, lds + 4);
  if (threadIdx.x == 0) {
    partial[(c * nblk + blk) * 2 + 0] = br;
    partial[(c * nblk + blk) * 2 + 1] = bi;
  }
}

// ------------------------------------------------------------------ staple force
// KICK = false: fn[c][mu] = coef * TAH(U A);  KICK = true: vn[c][mu] += coef * TAH(U A)
// A_mu(s) = sum_{nu != mu} [ U_nu(s+mu) U_mu(s+nu)^H U_nu(s)^H
//                           + U_nu(s+mu-nu)^H U_mu(s-nu)^H U_nu(s-nu) ]
template <bool KICK, int OCC>
__global__ __launch_bounds__(kBlock, OCC) void su3_force_kernel(const double2* __restrict__ xn,
                                                                Dims d, long nblk, int swz, double coef,
                                                                double2* __restrict__ out) {
  const long total = (long)gridDim.x;
  const long w = xcd_swizzle(blockIdx.x, total, swz);
  // logical order: chain-major, then site block, then mu (4 consecutive blocks share sites)
  const int mu = (int)(w & 3);
  const long cb = w >> 2;
  const long c = cb / nblk, blk = cb % nblk;
  const int s = (int)blk * kBlock + threadIdx.x;
  if (s >= d.V) return;
  const int V = d.V;
  const double2* xc = xn + c * 36L * V;
  const double2* fm = xc + mu * 9 * V;
  const Site p = site_coords(s, d);
  const int cmu = coord_of(p, mu);
  const int s_pmu = fwd(s, cmu, d, mu);
  M3 acc;
  m3_zero(acc);
#pragma unroll 1
  for (int nu = 0; nu < 4; ++nu) {
    if (nu == mu) continue;
    const double2* fn = xc + nu * 9 * V;
    const int cnu = coord_of(p, nu);
    const int s_pnu = fwd(s, cnu, d, nu);
    const int s_mnu = bwd(s, cnu, d, nu);
    const int s_pmu_mnu = bwd(s_pmu, cnu, d, nu);     // nu-coordinate unchanged by the mu hop
    M3 a, b, t;
    // up:   U_nu(s+mu) U_mu(s+nu)^H U_nu(s)^H
    load_link(a, fn, V, s_pmu);
    load_link(b, fm, V, s_pnu);
    m3_mul_na(t, a, b);
    load_link(a, fn, V, s);
    m3_mac_na(acc, t, a);
    // down: U_nu(s+mu-nu)^H U_mu(s-nu)^H U_nu(s-nu)
    load_link(a, fn, V, s_pmu_mnu);
    load_link(b, fm, V, s_mnu);
    m3_mul_aa(t, a, b);
    load_link(a, fn, V, s_mnu);
    m3_mac_nn(acc, t, a);
  }
  M3 u, ua, f;
  load_link(u, fm, V, s);
  m3_mul_nn(ua, u, acc);
  m3_tah(f, ua);
  double2* o = out + (c * 4 + mu) * 9L * V;
#pragma unroll
  for (int e = 0; e < 9; ++e) {
    double2 r = make_double2(coef * f.re[e], coef * f.im[e]);
    if (KICK) {
      const double2 v = o[e * V + s];
      r.x += v.x; r.y += v.y;
    }
    o[e * V + s] = r;
  }
}

// Per-plane variant for LatticeLoss._plaq_loss (loss/pytorch/loss.py:57-70 sums each of the 6
// planes separately): partial[c][blk][plane][re|im]; same site loop, one block reduction per
// plane.
__global__ __launch_bounds__(kBlock, 2) void su3_plaq_planes_kernel(const double2* __restrict__ xn,
                                                                    Dims d, long nblk,
                                                                    double* __restrict__ partial) {
  __shared__ double lds[8];
  const long c = blockIdx.x / nblk, blk = blockIdx.x % nblk;
  const int s = (int)blk * kBlock + threadIdx.x;
  const bool live = s < d.V;
  const double2* xc = xn + c * 36L * d.V;
  const Site p = site_coords(live ? s : 0, d);
  const int V = d.V, ss = live ? s : 0;
  int plane = 0;
#pragma unroll 1
  for (int u = 1; u < 4; ++u) {
    const int s_pu = fwd(ss, coord_of(p, u), d, u);
#pragma unroll 1
    for (int v = 0; v < u; ++v, ++plane) {
      const int s_pv = fwd(ss, coord_of(p, v), d, v);
      double sr = 0.0, si = 0.0;
      M3 a, b, yuv;
      load_link(a, xc + u * 9 * V, V, ss);
      load_link(b, xc + v * 9 * V, V, s_pu);
      m3_mul_nn(yuv, a, b);
      load_link(a, xc + v * 9 * V, V, ss);
      load_link(b, xc + u * 9 * V, V, s_pv);
      m3_trace_y_abh(sr, si, yuv, a, b);
      if (!live) { sr = 0.0; si = 0.0; }
      const double br = block_sum(sr, lds);
      const double bi = block_sum(si, lds + 4);
      if (threadIdx.x == 0) {
        partial[((c * nblk + blk) * 6 + plane) * 2 + 0] = br;
        partial[((c * nblk + blk) * 6 + plane) * 2 + 1] = bi;
      }
    }
  }
}

// The trace FIELD itself, out[plane][chain][site] = tr P_plane(site) (complex): the tensor the reference's
// `LatticeSU3.wilson_loops` returns, [6, nb, T, X, Y, Z] (lattice/su3/pytorch/lattice.py:157-174, 242-244).
// The sampler never needs it (action / plaquette / charges are the per-chain sums above); callers that
// reduce the field themselves (loss/pytorch/loss.py:57-110) do.  864 B read + 96 B written per site.
__global__ __launch_bounds__(kBlock, 2) void su3_wloops_kernel(const double2* __restrict__ xn, Dims d,
                                                               long nblk, long nb,
                                                               double2* __restrict__ out) {
  const long c = blockIdx.x / nblk, blk = blockIdx.x % nblk;
  const int s = (int)blk * kBlock + threadIdx.x;
  if (s >= d.V) return;
  const double2* xc = xn + c * 36L * d.V;
  const Site p = site_coords(s, d);
  const int V = d.V;
  int plane = 0;
#pragma unroll 1
  for (int u = 1; u < 4; ++u) {
    const int s_pu = fwd(s, coord_of(p, u), d, u);
#pragma unroll 1
    for (int v = 0; v < u; ++v, ++plane) {
      const int s_pv = fwd(s, coord_of(p, v), d, v);
      double sr = 0.0, si = 0.0;
      M3 a, b, yuv;
      load_link(a, xc + u * 9 * V, V, s);
      load_link(b, xc + v * 9 * V, V, s_pu);
      m3_mul_nn(yuv, a, b);
      load_link(a, xc + v * 9 * V, V, s);
      load_link(b, xc + u * 9 * V, V, s_pv);
      m3_trace_y_abh(sr, si, yuv, a, b);
      out[(plane * nb + c) * V + s] = make_double2(sr, si);
    }
  }
}

// per-chain sum |a - b|^2 over n doubles (complex fields pass 2n)
__global__ __launch_bounds__(kBlock) void diff_norm2_kernel(const double* __restrict__ a,
                                                            const double* __restrict__ b, long n,
                                                            long nblk, double* __restrict__ partial) {
  __shared__ double lds[4];
  const long c = blockIdx.x / nblk, blk = blockIdx.x % nblk;
  double acc = 0.0;
  const long base = blk * (4L * kBlock);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long j = base + (long)k * kBlock + threadIdx.x;
    if (j < n) {
      const double dlt = a[c * n + j] - b[c * n + j];
      acc = fma(dlt, dlt, acc);
    }
  }
  const double r = block_sum(acc, lds);
  if (threadIdx.x == 0) partial[c * nblk + blk] = r;
}

// ------------------------------------------------------------------ plaquette, t-sweep
// One workgroup = 256 spatial sites of one chain, sweeping a range of t.  The +t neighbours
// loaded in iteration t (U_x, U_y, U_z at t+1) are the same lines the block asks for as its
// own links one iteration later, i.e. within ~1 us on the same XCD -> L2 hits instead of a
// second trip through the fabric (the flat kernel re-fetched them: 2.7x algorithmic bytes).
template <int OCC>
__global__ __launch_bounds__(kBlock, OCC) void su3_plaq_sweep_kernel(
    const double2* __restrict__ xn, Dims d, int nsb, int tsplit, int swz,
    double* __restrict__ partial) {
  __shared__ double lds[8];
  const long w = xcd_swizzle(blockIdx.x, gridDim.x, swz);
  const int per_chain = nsb * tsplit;
  const long c = w / per_chain;
  const int r = (int)(w % per_chain);
  const int tc = r / nsb, sb = r % nsb;
  const int Vs = d.X * d.Y * d.Z;
  const int sp = sb * kBlock + threadIdx.x;             // spatial site
  const int tlen = (d.T + tsplit - 1) / tsplit;
  const int t0 = tc * tlen, t1 = min(d.T, t0 + tlen);
  double sr = 0.0, si = 0.0;
  if (sp < Vs) {
    const double2* xc = xn + c * 36L * d.V;
    const int V = d.V;
    Site p = site_coords(sp, d);                         // p.t == 0 here (sp < Vs)
#pragma unroll 1
    for (int t = t0; t < t1; ++t) {
      p.t = t;
      const int s = t * Vs + sp;
#pragma unroll 1
      for (int u = 1; u < 4; ++u) {
        const int s_pu = fwd(s, coord_of(p, u), d, u);
        M3 au;
        load_link(au, xc + u * 9 * V, V, s);
#pragma unroll 1
        for (int v = 0; v < u; ++v) {
          const int s_pv = fwd(s, coord_of(p, v), d, v);
          M3 a, b, yuv;
          load_link(b, xc + v * 9 * V, V, s_pu);
          m3_mul_nn(yuv, au, b);
          load_link(a, xc + v * 9 * V, V, s);
          load_link(b, xc + u * 9 * V, V, s_pv);
          m3_trace_y_abh(sr, si, yuv, a, b);
        }
      }
    }
  }
  const double br = block_sum(sr, lds);
  const double bi = block_sum(si, lds + 4);
  if (threadIdx.x == 0) {
    partial[(c * per_chain + r) * 2 + 0] = br;
    partial[(c * per_chain + r) * 2 + 1] = bi;
  }
}

// ------------------------------------------------------------------ plaquette, slice-resident
// One workgroup = 128 spatial sites of one chain, sweeping t.  LDS holds the 4 links of the
// CURRENT time slice for the tile ([4][9][128] complex = 72 KiB); the NEXT slice's links are
// prefetched into registers while the current slice is being computed and become the LDS
// tile of the next iteration.  Every link is therefore fetched from HBM exactly once per
// sweep (plus the x-halo of the tile): own links, +y/+z(/+x) neighbours come from LDS, the
// +t neighbours from the prefetch registers.  (The flat kernel moves 2.7x the algorithmic
// bytes through the fabric and is bound by that.)
constexpr int kSlice = 128;

__device__ __forceinline__ void lds_store_link(double2* tile, int rho, int li, const M3& m) {
#pragma unroll
  for (int e = 0; e < 9; ++e) tile[(rho * 9 + e) * kSlice + li] = make_double2(m.re[e], m.im[e]);
}

// link rho through a (base, stride) pair that points either into the LDS tile (stride 128)
// or into the global slice (stride V): one code path, flat addressing
struct SliceRef {
  const double2* base;   // entry e of link rho at base[(rho * 9 + e) * stride]
  int stride;
};

// own-site link: always in the tile -> plain ds_read_b128
__device__ __forceinline__ void slice_own(M3& m, const double2* tile, int rho, int li) {
  const double2* l = tile + rho * 9 * kSlice + li;
#pragma unroll
  for (int e = 0; e < 9; ++e) {
    const double2 dd = l[e * kSlice];
    m.re[e] = dd.x; m.im[e] = dd.y;
  }
}

// neighbour link: LDS when the (wave-uniform, loop-invariant) direction is known in-tile
template <bool LDS>
__device__ __forceinline__ void slice_nbr(M3& m, const SliceRef& r, const double2* tile, int lnb,
                                          int rho);
template <>
__device__ __forceinline__ void slice_nbr<true>(M3& m, const SliceRef&, const double2* tile,
                                                int lnb, int rho) {
  slice_own(m, tile, rho, lnb);
}

__device__ __forceinline__ void slice_link(M3& m, const SliceRef& r, int rho) {
  const double2* p = r.base + rho * 9 * r.stride;
#pragma unroll
  for (int e = 0; e < 9; ++e) {
    const double2 dd = p[e * r.stride];
    m.re[e] = dd.x; m.im[e] = dd.y;
  }
}

template <>
__device__ __forceinline__ void slice_nbr<false>(M3& m, const SliceRef& r, const double2*, int,
                                                 int rho) {
  slice_link(m, r, rho);
}

// YZ: +y and +z neighbours of every lane are inside the tile (e.g. the tile is a stack of
// whole (y,z) planes) -> plain LDS reads for them; +x stays on the generic flat path.
template <bool YZ>
__global__ __launch_bounds__(kSlice, 1) void su3_plaq_slice_kernel(
    const double2* __restrict__ xn, Dims d, int nsb, int tsplit, int swz,
    double* __restrict__ partial) {
  __shared__ double2 tile[4 * 9 * kSlice];
  __shared__ double red[8];
  const long w = xcd_swizzle(blockIdx.x, gridDim.x, swz);
  const int per_chain = nsb * tsplit;
  const long c = w / per_chain;
  const int r = (int)(w % per_chain);
  const int tc = r / nsb, sb = r % nsb;
  const int Vs = d.X * d.Y * d.Z, V = d.V;
  const int tile0 = sb * kSlice;
  const int li = threadIdx.x;
  const int sp = tile0 + li;                            // spatial site (Vs % 128 == 0)
  const int tlen = (d.T + tsplit - 1) / tsplit;
  const int t0 = tc * tlen, t1 = min(d.T, t0 + tlen);
  const double2* xc = xn + c * 36L * V;
  // periodic spatial neighbours (indices within a slice)
  int sn[4];
  {
    int q = sp;
    const int z = q % d.Z; q /= d.Z;
    const int y = q % d.Y; q /= d.Y;
    const int x = q;
    sn[0] = sp;
    sn[1] = (x + 1 == d.X) ? sp - (d.X - 1) * d.Y * d.Z : sp + d.Y * d.Z;
    sn[2] = (y + 1 == d.Y) ? sp - (d.Y - 1) * d.Z : sp + d.Z;
    sn[3] = (z + 1 == d.Z) ? sp - (d.Z - 1) : sp + 1;
  }
  bool in_tile[4];                                      // wave-uniform, loop-invariant
#pragma unroll
  for (int u = 0; u < 4; ++u) in_tile[u] = __all((unsigned)(sn[u] - tile0) < (unsigned)kSlice);
  M3 nxt[4];
#pragma unroll
  for (int rho = 0; rho < 4; ++rho) load_link(nxt[rho], xc + rho * 9 * V, V, t0 * Vs + sp);
  double sr = 0.0, si = 0.0;
#pragma unroll 1
  for (int t = t0; t < t1; ++t) {
    __syncthreads();                                    // previous slice fully consumed
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) lds_store_link(tile, rho, li, nxt[rho]);
    __syncthreads();
    const int tn = (t + 1 == d.T) ? 0 : t + 1;
    // prefetch the next slice: U_1..U_3 now (needed by the temporal planes below), U_0 after
    // the spatial planes (only needed as next iteration's tile) to cap register pressure
#pragma unroll
    for (int rho = 1; rho < 4; ++rho) load_link(nxt[rho], xc + rho * 9 * V, V, tn * Vs + sp);
    const double2* xs = xc + (long)t * Vs;              // base of slice t (index by spatial site)
    SliceRef nb_[4];                                    // own site and +x, +y, +z neighbours
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      nb_[u].base = in_tile[u] ? (const double2*)(tile + (sn[u] - tile0)) : (xs + sn[u]);
      nb_[u].stride = in_tile[u] ? kSlice : V;
    }
    // spatial-spatial planes (u, v) = (2,1), (3,1), (3,2): everything in the current slice
    {
      M3 a, b, yuv;
      slice_own(a, tile, 2, li); slice_nbr<YZ>(b, nb_[2], tile, sn[2] - tile0, 1); m3_mul_nn(yuv, a, b);
      slice_own(a, tile, 1, li); slice_link(b, nb_[1], 2); m3_trace_y_abh(sr, si, yuv, a, b);
    }
    {
      M3 a, b, yuv;
      slice_own(a, tile, 3, li); slice_nbr<YZ>(b, nb_[3], tile, sn[3] - tile0, 1); m3_mul_nn(yuv, a, b);
      slice_own(a, tile, 1, li); slice_link(b, nb_[1], 3); m3_trace_y_abh(sr, si, yuv, a, b);
    }
    {
      M3 a, b, yuv;
      slice_own(a, tile, 3, li); slice_nbr<YZ>(b, nb_[3], tile, sn[3] - tile0, 2); m3_mul_nn(yuv, a, b);
      slice_own(a, tile, 2, li); slice_nbr<YZ>(b, nb_[2], tile, sn[2] - tile0, 3); m3_trace_y_abh(sr, si, yuv, a, b);
    }
    // temporal planes (u, 0): U_u(s) U_0(s+u) (U_0(s) U_u(s+t))^H, U_u(s+t) = prefetched link
    load_link(nxt[0], xc, V, tn * Vs + sp);
    {
      M3 a, b, yuv;
      slice_own(a, tile, 1, li); slice_link(b, nb_[1], 0); m3_mul_nn(yuv, a, b);
      slice_own(a, tile, 0, li); m3_trace_y_abh(sr, si, yuv, a, nxt[1]);
    }
    {
      M3 a, b, yuv;
      slice_own(a, tile, 2, li); slice_nbr<YZ>(b, nb_[2], tile, sn[2] - tile0, 0); m3_mul_nn(yuv, a, b);
      slice_own(a, tile, 0, li); m3_trace_y_abh(sr, si, yuv, a, nxt[2]);
    }
    {
      M3 a, b, yuv;
      slice_own(a, tile, 3, li); slice_nbr<YZ>(b, nb_[3], tile, sn[3] - tile0, 0); m3_mul_nn(yuv, a, b);
      slice_own(a, tile, 0, li); m3_trace_y_abh(sr, si, yuv, a, nxt[3]);
    }
  }
  // block reduction (2 waves)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { sr += __shfl_down(sr, off, 64); si += __shfl_down(si, off, 64); }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[(threadIdx.x >> 6) * 2] = sr; red[(threadIdx.x >> 6) * 2 + 1] = si; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[(c * per_chain + r) * 2 + 0] = red[0] + red[2];
    partial[(c * per_chain + r) * 2 + 1] = red[1] + red[3];
  }
}

// ------------------------------------------------------------------ staple force, LDS tile
// One workgroup = 64 consecutive sites x 4 directions (wavefront w <-> mu = w).  The 4 own
// links of the 64 sites are staged once in LDS ([4][9][64] complex = 36 KiB); every operand
// whose site falls inside the tile (the site itself and, for 8^4 / 16^4 lattices, all +-y,
// +-z neighbours of a (y,z) plane) is then an LDS read instead of a 9 KiB trip to L2.
// For 8^4 that turns 42 of the 76 matrix loads per site into LDS reads (the flat kernel is
// L2-bandwidth-bound: 12 GB of L2->L1 traffic per launch at cfg-4).
struct TileRef {
  const double2* lds;    // [4][9][64]
  int s0;                // first site of the tile
};

// One code path for both sources: a generic (flat) pointer that is either the lane's slot in
// the LDS tile (stride 64) or the global plane (stride V), chosen wave-uniformly.
__device__ __forceinline__ void load_link_tiled(M3& m, const double2* __restrict__ xc,
                                                const TileRef& tr, int rho, int V, int s) {
  const int li = s - tr.s0;
  const bool in = __all((unsigned)li < 64u);
  const double2* base = in ? (tr.lds + rho * 9 * 64 + li) : (xc + rho * 9 * V + s);
  const int stride = in ? 64 : V;
#pragma unroll
  for (int e = 0; e < 9; ++e) {
    const double2 dd = base[e * stride];
    m.re[e] = dd.x; m.im[e] = dd.y;
  }
}

template <bool KICK, int OCC>
__global__ __launch_bounds__(kBlock, OCC) void su3_force_tile_kernel(
    const double2* __restrict__ xn, Dims d, long ntile, int swz, double coef,
    double2* __restrict__ out) {
  __shared__ double2 tile[4 * 9 * 64];
  const long w = xcd_swizzle(blockIdx.x, gridDim.x, swz);
  const long c = w / ntile, tb = w % ntile;
  const int mu = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int V = d.V;
  const int s0 = (int)tb * 64;
  const bool live = s0 + lane < V;
  const int s = live ? s0 + lane : V - 1;             // clamp: dead lanes still take part
  const double2* xc = xn + c * 36L * V;
  {
    const double2* g = xc + mu * 9 * V;
#pragma unroll
    for (int e = 0; e < 9; ++e) tile[(mu * 9 + e) * 64 + lane] = g[e * V + s];
  }
  __syncthreads();
  TileRef tr{tile, live ? s0 : -1000000};
  if (!__all(live)) tr.s0 = -1000000;                 // ragged last tile: global path only
  const Site p = site_coords(s, d);
  const int cmu = coord_of(p, mu);
  const int s_pmu = fwd(s, cmu, d, mu);
  M3 acc;
  m3_zero(acc);
#pragma unroll 1
  for (int nu = 0; nu < 4; ++nu) {
    if (nu == mu) continue;
    const int cnu = coord_of(p, nu);
    const int s_pnu = fwd(s, cnu, d, nu);
    const int s_mnu = bwd(s, cnu, d, nu);
    const int s_pmu_mnu = bwd(s_pmu, cnu, d, nu);
    M3 a, b, t;
    // up:   U_nu(s+mu) U_mu(s+nu)^H U_nu(s)^H
    load_link_tiled(a, xc, tr, nu, V, s_pmu);
    load_link_tiled(b, xc, tr, mu, V, s_pnu);
    m3_mul_na(t, a, b);
    load_link_tiled(a, xc, tr, nu, V, s);
    m3_mac_na(acc, t, a);
    // down: U_nu(s+mu-nu)^H U_mu(s-nu)^H U_nu(s-nu)
    load_link_tiled(a, xc, tr, nu, V, s_pmu_mnu);
    load_link_tiled(b, xc, tr, mu, V, s_mnu);
    m3_mul_aa(t, a, b);
    load_link_tiled(a, xc, tr, nu, V, s_mnu);
    m3_mac_nn(acc, t, a);
  }
  M3 u, ua, f;
  load_link_tiled(u, xc, tr, mu, V, s);
  m3_mul_nn(ua, u, acc);
  m3_tah(f, ua);
  if (!live) return;
  double2* o = out + (c * 4 + mu) * 9L * V;
#pragma unroll
  for (int e = 0; e < 9; ++e) {
    double2 r = make_double2(coef * f.re[e], coef * f.im[e]);
    if (KICK) {
      const double2 v = o[e * V + s];
      r.x += v.x; r.y += v.y;
    }
    o[e * V + s] = r;
  }
}

// ------------------------------------------------------------------ staple force, slice-resident
// One workgroup = 128 spatial sites x 4 directions (512 threads, wavefront pairs <-> mu),
// sweeping t.  LDS holds TWO time slices of the tile's links (current and next,
// 2 x [4][9][128] complex = 144 KiB); each thread prefetches its own link of the slice after
// next into registers.  The staples that reach back to slice t-1 (the down-staple of a
// spatial link in the t direction, built from slice t-1 links only) are computed one
// iteration early and carried in registers, so slice t-1 is never needed again.
// Per site and sweep the links are read from HBM once (+ the x-halo of the tile); ~17 of the
// 19 operands of a link come from LDS instead of L2 (the flat kernel issues 76 matrix loads
// per site to L2 and leaves the SIMDs idle 53 % of the time waiting for them).
// Register budget decides this kernel.  A workgroup is 64 sites x 4 directions = 4 wavefronts and
// is compiled for ONE wavefront per SIMD (__launch_bounds__(256, 1)): the unified register file
// then gives each wavefront 512 registers, and the ~70 values that do not fit the 256 VGPRs live
// in AGPRs (v_accvgpr moves) instead of scratch memory.  Measured on MI355X at cfg-4:
//   128-site tiles, 2 waves/SIMD (256 registers):  variant 0: 143 spilled VGPRs, 0.97 ms;
//     variant 1: 59 spills, 0.64 ms;  variant 2: 18 spills, 0.56-0.58 ms
//   64-site tiles, 1 wave/SIMD (256 VGPR + 72 AGPR, no scratch):  variant 0: 0.52 ms;
//     variant 1: 0.54 ms;  variant 2: 0.56 ms   (without the scheduling fences: 0.55 ms;
//     staple loops fully unrolled: 512 registers + 660 B scratch)
// 0: carried t-staple + register prefetch of the slice after next   1: no prefetch
// 2: no prefetch, t-staple re-read from slice t-1 through L2
// The plain force runs 128-site tiles with two links per thread at one wavefront per SIMD,
// variant 0 (same-box A/B: 0.51 ms; 64-site one-wave build 0.55 ms; 128-site two-wave variant 2
// 0.59 ms); the fused v += coef F variant (two more HBM streams) keeps the 128-site / two-wave /
// variant-2 build (0.73 ms vs 0.75-0.77 ms for the one-wave builds).
constexpr int kFSPlain = 128, kFSKick = 128, kLptPlain = 2;
// compiler-only fence: keeps hipcc from hoisting the next staple's operand loads above the
// current staple's arithmetic
#define L2Q_SCHED_FENCE() asm volatile("" ::: "memory")

struct SPos {
  int q, x, y, z;        // spatial site index and coordinates
};

// one periodic hop in spatial direction dir (1, 2, 3 = x, y, z); dir and sgn are wave-uniform
__device__ __forceinline__ SPos sp_move(SPos p, int dir, int sgn, const Dims& d) {
  const int n = dir == 1 ? d.X : dir == 2 ? d.Y : d.Z;
  const int st = dir == 1 ? d.Y * d.Z : dir == 2 ? d.Z : 1;
  int c = dir == 1 ? p.x : dir == 2 ? p.y : p.z;
  if (sgn > 0) {
    if (c + 1 == n) { p.q -= (n - 1) * st; c = 0; } else { p.q += st; c += 1; }
  } else {
    if (c == 0) { p.q += (n - 1) * st; c = n - 1; } else { p.q -= st; c -= 1; }
  }
  if (dir == 1) p.x = c; else if (dir == 2) p.y = c; else p.z = c;
  return p;
}

// link rho of spatial site q in a slice: LDS copy if q is inside the tile (wave-uniform),
// else the global slice; one code path through a flat pointer
template <int kFS>
__device__ __forceinline__ void fs_get(M3& m, const double2* slot, const double2* __restrict__ gsl,
                                       int rho, int q, int tile0, int V) {
  const int li = q - tile0;
  const bool in = __all((unsigned)li < (unsigned)kFS);
  const double2* base = in ? (slot + rho * 9 * kFS + li) : (gsl + rho * 9 * V + q);
  const int stride = in ? kFS : V;
#pragma unroll
  for (int e = 0; e < 9; ++e) {
    const double2 dd = base[e * stride];
    m.re[e] = dd.x; m.im[e] = dd.y;
  }
}

template <int kFS>
__device__ __forceinline__ void fs_own(M3& m, const double2* slot, int rho, int lt) {
  const double2* l = slot + rho * 9 * kFS + lt;
#pragma unroll
  for (int e = 0; e < 9; ++e) {
    const double2 dd = l[e * kFS];
    m.re[e] = dd.x; m.im[e] = dd.y;
  }
}

template <int kFS>
__device__ __forceinline__ void fs_put(double2* slot, int rho, int lt, const M3& m) {
#pragma unroll
  for (int e = 0; e < 9; ++e) slot[(rho * 9 + e) * kFS + lt] = make_double2(m.re[e], m.im[e]);
}

// LPT: links (directions) per thread.  2: a workgroup of 2 kFS threads covers mu = g and g + 2
// one after the other -- the 128-site tile (half the x-halo of the 64-site one) at one wavefront
// per SIMD.
template <bool KICK, int kFS, int VARIANT, int LPT>
__global__ __launch_bounds__(4 * kFS / LPT, (kFS == 64 || LPT == 2) ? 1 : 2) void su3_force_slice_kernel(
    const double2* __restrict__ xn, Dims d, int nsb, int tsplit, int swz, double coef,
    double2* __restrict__ out) {
  extern __shared__ double2 fs_lds[];                   // [2][4][9][kFS]
  constexpr int kSlot = 4 * 9 * kFS;
  const long w = xcd_swizzle(blockIdx.x, gridDim.x, swz);
  const int per_chain = nsb * tsplit;
  const long c = w / per_chain;
  const int r = (int)(w % per_chain);
  const int tc = r / nsb, sb = r % nsb;
  const int Vs = d.X * d.Y * d.Z, V = d.V, T = d.T;
  const int tile0 = sb * kFS;
  const int lt = threadIdx.x & (kFS - 1), g = threadIdx.x / kFS;       // g (hence mu) is wave-uniform
  constexpr int MUSTEP = 4 / LPT;
  const int tlen = (T + tsplit - 1) / tsplit;
  const int t0 = tc * tlen, t1 = min(T, t0 + tlen);
  const double2* xc = xn + c * 36L * V;
  SPos p;
  p.q = tile0 + lt;
  {
    int q = p.q;
    p.z = q % d.Z; q /= d.Z;
    p.y = q % d.Y; q /= d.Y;
    p.x = q;
  }
  const int sp = p.q;
  // prologue: slices t0-1 and t0 of the tile -> slots 0, 1 (each thread its own link)
  {
    const int ta = (t0 - 1 + T) % T;
#pragma unroll 1
    for (int k = 0; k < 2 * LPT; ++k) {
      const int sl = (ta + (k & 1)) % T, mu = g + MUSTEP * (k >> 1);
      M3 tmp;
      load_link(tmp, xc + mu * 9 * V, V, sl * Vs + sp);
      fs_put<kFS>(fs_lds + (k & 1) * kSlot, mu, lt, tmp);
    }
  }
  __syncthreads();
  int slot_cur = 0;
  M3 dcarry_[LPT], pre_[LPT];
#pragma unroll
  for (int h = 0; h < LPT; ++h) m3_zero(dcarry_[h]);
  const int niter = (t1 - t0) + 1;
#pragma unroll 1
  for (int it = 0; it < niter; ++it) {
    const int tcur = (t0 - 1 + it + T) % T;
    const int tnext = (tcur + 1) % T;
    const double2* cur = fs_lds + slot_cur * kSlot;
    const double2* nxt = fs_lds + (slot_cur ^ 1) * kSlot;
    const double2* gcur = xc + (long)tcur * Vs;
    const double2* gnxt = xc + (long)tnext * Vs;
    const bool more = it + 1 < niter;
#pragma unroll
    for (int h = 0; h < LPT; ++h) {
    const int mu = g + MUSTEP * h;
    double2* oc = out + (c * 4 + mu) * 9L * V;
    SPos pmu = p;
    if (mu != 0) pmu = sp_move(p, mu, +1, d);
    if (it > 0) {
      M3 acc;
      if (mu == 0) {
        m3_zero(acc);
#pragma unroll 1
        for (int nu = 1; nu < 4; ++nu) {
          const SPos pp = sp_move(p, nu, +1, d), pm = sp_move(p, nu, -1, d);
          M3 a, b, t;
          // up:   U_nu(s+t) U_t(s+nu)^H U_nu(s)^H
          fs_own<kFS>(a, nxt, nu, lt);
          fs_get<kFS>(b, cur, gcur, 0, pp.q, tile0, V);
          m3_mul_na(t, a, b);
          fs_own<kFS>(a, cur, nu, lt);
          m3_mac_na(acc, t, a);
          L2Q_SCHED_FENCE();
          // down: U_nu(s+t-nu)^H U_t(s-nu)^H U_nu(s-nu)
          fs_get<kFS>(a, nxt, gnxt, nu, pm.q, tile0, V);
          fs_get<kFS>(b, cur, gcur, 0, pm.q, tile0, V);
          m3_mul_aa(t, a, b);
          fs_get<kFS>(a, cur, gcur, nu, pm.q, tile0, V);
          m3_mac_nn(acc, t, a);
          L2Q_SCHED_FENCE();
        }
      } else {
        if constexpr (VARIANT != 2) {
          acc = dcarry_[h];                                // down staple in the t direction
        } else {                                       // t-direction down staple from slice t-1 (L2)
          const double2* gprv = xc + (long)((tcur - 1 + T) % T) * Vs;
          M3 a, b, t;
          load_link(a, gprv, V, pmu.q);
          load_link(b, gprv + mu * 9 * V, V, sp);
          m3_mul_aa(t, a, b);
          load_link(a, gprv, V, sp);
          m3_mul_nn(acc, t, a);
          L2Q_SCHED_FENCE();
        }
        {
          M3 a, b, t;
          // up (nu = t): U_t(s+mu) U_mu(s+t)^H U_t(s)^H
          fs_get<kFS>(a, cur, gcur, 0, pmu.q, tile0, V);
          fs_own<kFS>(b, nxt, mu, lt);
          m3_mul_na(t, a, b);
          fs_own<kFS>(a, cur, 0, lt);
          m3_mac_na(acc, t, a);
          L2Q_SCHED_FENCE();
        }
#pragma unroll 1
        for (int nu = 1; nu < 4; ++nu) {
          if (nu == mu) continue;
          const SPos pp = sp_move(p, nu, +1, d), pm = sp_move(p, nu, -1, d);
          const SPos pmm = sp_move(pmu, nu, -1, d);
          M3 a, b, t;
          // up:   U_nu(s+mu) U_mu(s+nu)^H U_nu(s)^H
          fs_get<kFS>(a, cur, gcur, nu, pmu.q, tile0, V);
          fs_get<kFS>(b, cur, gcur, mu, pp.q, tile0, V);
          m3_mul_na(t, a, b);
          fs_own<kFS>(a, cur, nu, lt);
          m3_mac_na(acc, t, a);
          L2Q_SCHED_FENCE();
          // down: U_nu(s+mu-nu)^H U_mu(s-nu)^H U_nu(s-nu)
          fs_get<kFS>(a, cur, gcur, nu, pmm.q, tile0, V);
          fs_get<kFS>(b, cur, gcur, mu, pm.q, tile0, V);
          m3_mul_aa(t, a, b);
          fs_get<kFS>(a, cur, gcur, nu, pm.q, tile0, V);
          m3_mac_nn(acc, t, a);
          L2Q_SCHED_FENCE();
        }
      }
      M3 u, ua, f;
      fs_own<kFS>(u, cur, mu, lt);
      m3_mul_nn(ua, u, acc);
      m3_tah(f, ua);
      const int s = tcur * Vs + sp;
#pragma unroll
      for (int e = 0; e < 9; ++e) {
        double2 rr = make_double2(coef * f.re[e], coef * f.im[e]);
        if (KICK) {
          const double2 v = oc[e * V + s];
          rr.x += v.x; rr.y += v.y;
        }
        oc[e * V + s] = rr;
      }
      L2Q_SCHED_FENCE();
    }
    // prefetch this thread's link of the slice after next (hidden behind the carry staple,
    // the barrier and the partner wavefront's work)
    if constexpr (VARIANT == 0) {
      if (more) load_link(pre_[h], xc + mu * 9 * V, V, ((tnext + 1) % T) * Vs + sp);
    }
    if (VARIANT != 2 && mu != 0 && more) {
      // next iteration's t-direction down staple of link (tnext, sp, mu), all from slice tcur:
      //   U_t(tcur, sp+mu)^H U_mu(tcur, sp)^H U_t(tcur, sp)
      M3 a, b, t;
      fs_get<kFS>(a, cur, gcur, 0, pmu.q, tile0, V);
      fs_own<kFS>(b, cur, mu, lt);
      m3_mul_aa(t, a, b);
      fs_own<kFS>(a, cur, 0, lt);
      m3_mul_nn(dcarry_[h], t, a);
    }
    }
    __syncthreads();                                    // slice tcur fully consumed
    if (more) {
#pragma unroll
      for (int h = 0; h < LPT; ++h) {
        const int mu = g + MUSTEP * h;
        if constexpr (VARIANT != 0) load_link(pre_[h], xc + mu * 9 * V, V, ((tnext + 1) % T) * Vs + sp);
        fs_put<kFS>(fs_lds + slot_cur * kSlot, mu, lt, pre_[h]);
      }
    }
    slot_cur ^= 1;
    __syncthreads();
  }
}

// ------------------------------------------------------------------ per-link kernels
// out = keep (.) x + expm(eps v) @ ((1-keep) (.) x).  TWO: the two half-updates of one
// leapfrog step (keep = mask then keep = 1 - mask, or the reverse order when `complement`)
// applied back to back with ONE expm(eps v) and one pass over x -- both are local to a link.
// VEC8: additionally emit su3_to_vec(projectSU(x')) (the vnet input of the next v-update,
// group/su3/pytorch/group.py:138-147) while x' is in registers -- saves the pass that re-reads it.
#ifndef XU_OCC
#define XU_OCC 3        // wavefronts per SIMD the x-update is compiled for (A/B builds override)
#endif
template <bool TWO, bool VEC8>
__global__ __launch_bounds__(kBlock, XU_OCC) void su3_expm_mul_kernel(const double2* xn,
                                                              const double2* __restrict__ vn,
                                                              double eps,
                                                              const float* __restrict__ mask,
                                                              int complement, double2* out,
                                                              double* __restrict__ out_vec,
                                                              int V, long nblk, int lo) {
  const long f = blockIdx.x / nblk, blk = blockIdx.x % nblk;     // f = chain*4 + mu
  const int s = (int)blk * kBlock + threadIdx.x;
  if (s >= V) return;
  const int mu = (int)(f & 3);
  // The three stages below sit in run-time conditionals that are always taken (`lo` = 0 is a kernel
  // ARGUMENT): hipcc then allocates each stage on its own instead of scheduling the exponential, the
  // masked products and the projection as one straight line with all their temporaries live
  // (218 VGPRs, two wavefronts per SIMD; capped at 168 it spilled 52).  Staged, with the masked copies of
  // x formed inside the product: 182 VGPRs uncapped, 168 with 10 spilled at three wavefronts per SIMD,
  // x-update 0.423 -> 0.402 ms at cfg-4 (the same staging at two wavefronts per SIMD: 0.443 ms).
  M3 x, e;
  load_link(e, vn + f * 9L * V, V, s);
  if (s >= lo) {
    M3 a;
#pragma unroll
    for (int i = 0; i < 9; ++i) { a.re[i] = e.re[i] * eps; a.im[i] = e.im[i] * eps; }
    m3_expm(e, a);
  }
  // (the link is requested only now: it would otherwise be live across the exponential; the other
  // wavefronts of the SIMD cover its latency)
  load_link(x, xn + f * 9L * V, V, s);
  if (mask != nullptr) {
    const float* mk = mask + mu * 9 * V;
    float keep[9];                           // 0 / 1 masks: exact in fp32, half the registers
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const float k = mk[i * V + s];
      keep[i] = complement ? 1.0f - k : k;
    }
    // m <- keep (.) m + e @ ((1 - keep) (.) m), IN PLACE and column by column: entry (i, j) of the
    // result needs column j of m only, so a column is read (masked copy formed on the fly), its
    // three results are formed and written back before the next column is touched -- the live set
    // is e, m, the masks and one column (no second and third copy of the link: that is what took
    // the staged kernel over the 168-register budget of three wavefronts per SIMD, 10 spilled).
    // Same products in the same order as the two-copy form: identical bits.
    auto half = [&](M3& m, bool flip) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double cr[3], ci[3], ki[3];
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
          const double kq = flip ? 1.0 - (double)keep[3 * kk + j] : (double)keep[3 * kk + j];
          cr[kk] = (1.0 - kq) * m.re[3 * kk + j]; ci[kk] = (1.0 - kq) * m.im[3 * kk + j];
          ki[kk] = kq;
        }
        double orr[3], oi[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          double sr = 0.0, si = 0.0;
#pragma unroll
          for (int kk = 0; kk < 3; ++kk) {
            const double ar = e.re[3 * i + kk], ai = e.im[3 * i + kk];
            sr = fma(ar, cr[kk], sr); sr = fma(-ai, ci[kk], sr);
            si = fma(ar, ci[kk], si); si = fma(ai, cr[kk], si);
          }
          orr[i] = fma(ki[i], m.re[3 * i + j], sr);
          oi[i] = fma(ki[i], m.im[3 * i + j], si);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) { m.re[3 * i + j] = orr[i]; m.im[3 * i + j] = oi[i]; }
      }
    };
    if (s >= lo) half(x, false);
    if (TWO) {
      if (s >= lo) half(x, true);
    }
  } else {
    if (s >= lo) {
      M3 r;
      m3_mul_nn(r, e, x);
      x = r;
    }
  }
  M3& r = x;
  store_link(out + f * 9L * V, V, s, r);
  if constexpr (VEC8) {
    if (s >= lo) {
      M3 p;
      m3_project_su(p, r);
      double w[8];
      m3_to_vec8(w, p);
#pragma unroll
      for (int a = 0; a < 8; ++a) out_vec[(f * 8 + a) * (long)V + s] = w[a];
    }
  }
}

// MODE 0: projectSU -> links;  1: projectSU -> vec8;  2: projectTAH -> links;  3: projectU
template <int MODE>
__global__ __launch_bounds__(kBlock) void su3_project_kernel(const double2* in,
                                                             double2* out_links,
                                                             double* __restrict__ out_vec, int V,
                                                             long nblk) {
  const long f = blockIdx.x / nblk, blk = blockIdx.x % nblk;
  const int s = (int)blk * kBlock + threadIdx.x;
  if (s >= V) return;
  M3 x, r;
  load_link(x, in + f * 9L * V, V, s);
  if (MODE == 2) {
    m3_tah(r, x);
  } else if (MODE == 3) {
    m3_project_u(r, x);
  } else {
    m3_project_su(r, x);
  }
  if (MODE == 1) {
    double v[8];
    m3_to_vec8(v, r);
#pragma unroll
    for (int a = 0; a < 8; ++a) out_vec[(f * 8 + a) * (long)V + s] = v[a];
  } else {
    store_link(out_links + f * 9L * V, V, s, r);
  }
}

// out = op(a) * op(b) per link, op = identity or adjoint   (SU3.mul, group.py:56-69)
__global__ __launch_bounds__(kBlock) void su3_mul_kernel(const double2* a, const double2* b,
                                                         int adj_a, int adj_b, double2* out, int V,
                                                         long nblk) {
  const long f = blockIdx.x / nblk, blk = blockIdx.x % nblk;
  const int s = (int)blk * kBlock + threadIdx.x;
  if (s >= V) return;
  M3 x, y, r;
  load_link(x, a + f * 9L * V, V, s);
  load_link(y, b + f * 9L * V, V, s);
  if (adj_a && adj_b) m3_mul_aa(r, x, y);
  else if (adj_a) m3_mul_an(r, x, y);
  else if (adj_b) m3_mul_na(r, x, y);
  else m3_mul_nn(r, x, y);
  store_link(out + f * 9L * V, V, s, r);
}

__global__ __launch_bounds__(kBlock) void su3_assemble_tah_kernel(const double* __restrict__ nrm,
                                                                  double2* __restrict__ vn, long V,
                                                                  long nblk, long nlinks) {
  const long f = blockIdx.x / nblk, blk = blockIdx.x % nblk;
  const long s = blk * kBlock + threadIdx.x;
  if (s >= V) return;
  const long l = f * V + s;
  const double h = 0.70710678118654757;          // sqrt(1/2)
  const double r3 = h * nrm[0 * nlinks + l];
  const double r8 = h * 0.57735026918962573 * nrm[1 * nlinks + l];
  const double r01 = h * nrm[2 * nlinks + l], r02 = h * nrm[3 * nlinks + l];
  const double r12 = h * nrm[4 * nlinks + l], i01 = h * nrm[5 * nlinks + l];
  const double i02 = h * nrm[6 * nlinks + l], i12 = h * nrm[7 * nlinks + l];
  double2* o = vn + f * 9 * V;
  o[0 * V + s] = make_double2(0.0, r8 + r3);
  o[1 * V + s] = make_double2(r01, i01);
  o[2 * V + s] = make_double2(r02, i02);
  o[3 * V + s] = make_double2(-r01, i01);
  o[4 * V + s] = make_double2(0.0, r8 - r3);
  o[5 * V + s] = make_double2(r12, i12);
  o[6 * V + s] = make_double2(-r02, i02);
  o[7 * V + s] = make_double2(-r12, i12);
  o[8 * V + s] = make_double2(0.0, -2.0 * r8);
}

// per-chain sum over all 36 V complex entries of |p|^2; the -8 per link and the 1/2 are
// applied in the finalize stage through `scale` and an offset there.
__global__ __launch_bounds__(kBlock) void su3_norm2_kernel(const double2* __restrict__ vn,
                                                           long n_per_chain, long nblk,
                                                           double* __restrict__ partial) {
  __shared__ double lds[4];
  const long c = blockIdx.x / nblk, blk = blockIdx.x % nblk;
  const double2* v = vn + c * n_per_chain;
  double acc = 0.0;
  // each block owns a contiguous run of 4 * kBlock entries
  const long base = blk * (4L * kBlock);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long j = base + (long)k * kBlock + threadIdx.x;
    if (j < n_per_chain) {
      const double2 d = v[j];
      acc = fma(d.x, d.x, acc);
      acc = fma(d.y, d.y, acc);
    }
  }
  const double r = block_sum(acc, lds);
  if (threadIdx.x == 0) partial[c * nblk + blk] = r;
}

__global__ __launch_bounds__(kBlock) void su3_check_kernel(const double2* __restrict__ xn, long V,
                                                           long nblk, double* __restrict__ psum,
                                                           double* __restrict__ pmax) {
  __shared__ double lds[8];
  const long c = blockIdx.x / nblk, blk = blockIdx.x % nblk;   // blk over 4 V links
  const long l = blk * kBlock + threadIdx.x;
  double dv = 0.0;
  const bool on = l < 4 * V;
  if (on) {
    const long mu = l / V, s = l % V;
    M3 x, t;
    load_link(x, xn + (c * 4 + mu) * 9L * V, (int)V, (int)s);
    m3_mul_an(t, x, x);
    t.re[0] -= 1.0; t.re[4] -= 1.0; t.re[8] -= 1.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) dv += t.re[i] * t.re[i] + t.im[i] * t.im[i];
    double dr, di;
    m3_det(dr, di, x);
    dv += (dr - 1.0) * (dr - 1.0) + di * di;
  }
  const double bs = block_sum(dv, lds);
  const double bm = block_max(dv, lds + 4);
  if (threadIdx.x == 0) { psum[c * nblk + blk] = bs; pmax[c * nblk + blk] = bm; }
}

__global__ void check_finalize_kernel(const double* __restrict__ psum,
                                      const double* __restrict__ pmax, long nblk, double nlinks,
                                      double* __restrict__ out) {
  __shared__ double lds[8];
  const long c = blockIdx.x;
  double s = 0.0, m = 0.0;
  for (long b = threadIdx.x; b < nblk; b += blockDim.x) {
    s += psum[c * nblk + b];
    m = fmax(m, pmax[c * nblk + b]);
  }
  const double ts = block_sum(s, lds);
  const double tm = block_max(m, lds + 4);
  if (threadIdx.x == 0) {
    out[c * 2 + 0] = sqrt(ts / nlinks / 20.0);
    out[c * 2 + 1] = sqrt(tm / 20.0);
  }
}

}  // namespace l2q

namespace l2q {
// su3_force_rows.hip
bool force_rows_applicable(const Dims& d);
int force_rows_inmask(const Dims& d);
// su3_force_nu.hip
bool force_nu_applicable(const Dims& d);
int force_nu_inmask(const Dims& d);
bool plaq_nu_applicable(const Dims& d);
long plaq_nu_per_chain(const Dims& d, int nb);
void launch_plaq_nu(const double2* xn, Dims d, int nb, double* partial, hipStream_t st);
void launch_force_nu(bool kick, const double2* xn, Dims d, int nb, double coef, double2* out,
                     hipStream_t st);
void launch_force_rows(bool kick, const double2* xn, Dims d, int nb, double coef, double2* out,
                       hipStream_t st);
// su3_force_link.hip
bool force_link_applicable(const Dims& d);
int force_link_inmask(const Dims& d);
void launch_force_link(bool kick, const double2* xn, Dims d, int nb, double coef, double2* out,
                       hipStream_t st, const double2* vin = nullptr);
// gemm_lt.hip
bool gemm_h_lt_shape(int M, int N, long K);
bool gemm_h_lt_available();
// su3_force_plaq.hip
bool force_plaq_applicable(const Dims& d);
void launch_force_plaq(const double2* xn, Dims d, int nb, double coef, double2* out, hipStream_t st);
// su3_force_pair.hip
bool force_pair_applicable(const Dims& d);
int force_pair_inmask(const Dims& d);
void launch_force_pair(bool kick, const double2* xn, Dims d, int nb, double coef, double2* out,
                       hipStream_t st, const double2* vin = nullptr);
}  // namespace l2q

using namespace l2q;

// force_tile = 4 picks per lattice (same-box A/B, tools/force_bench.py): the plane-split kernel
// wins when its 64-site tile holds whole (y,z)-planes (8^4: 0.466 vs 0.575 ms) and for the fused
// kick; on 16^4 (tile = 4 z-rows: y AND x leave the tile) the plain force is 7 % faster with the
// 128-site thread-per-link kernel (2.11 vs 2.26 ms at 64 chains).
template <bool KICK>
static bool nu_preferred(const Dims& d) {
  const int Vs_ = d.X * d.Y * d.Z;
  const bool slice_ok = Vs_ % (KICK ? kFSKick : kFSPlain) == 0;
  return KICK || !slice_ok || (force_nu_inmask(d) & 2) != 0;
}

template <bool KICK>
static void launch_force(const double2* xn, Dims d, int nb, long nblk, double coef, double2* out,
                         hipStream_t st) {
  const int Vs_ = d.X * d.Y * d.Z;
  constexpr int kFS = KICK ? kFSKick : kFSPlain;
  constexpr int kVar = KICK ? 2 : 0;
  constexpr int kLpt = KICK ? 1 : kLptPlain;
  if (!KICK && tuning().force_tile == 7 && force_plaq_applicable(d)) {
    launch_force_plaq(xn, d, nb, coef, out, st);               // plaquettes shared between their four links
    return;
  }
  if (tuning().force_tile == 6 && force_pair_applicable(d)) {
    launch_force_pair(KICK, xn, d, nb, coef, out, st);
    return;
  }
  if (tuning().force_tile >= 5 && force_link_applicable(d)) {
    launch_force_link(KICK, xn, d, nb, coef, out, st);
    return;
  }
  if (tuning().force_tile >= 4 && force_nu_applicable(d) && nu_preferred<KICK>(d)) {
    launch_force_nu(KICK, xn, d, nb, coef, out, st);
    return;
  }
  if (tuning().force_tile == 3 && force_rows_applicable(d)) {
    launch_force_rows(KICK, xn, d, nb, coef, out, st);
    return;
  }
  if (tuning().force_tile >= 2 && Vs_ % kFS == 0) {
    const int nsb = Vs_ / kFS;
    int tsplit = (int)cdiv(512, (long)nb * nsb);       // >= ~2 resident rounds of 256 CUs
    if (tsplit > d.T) tsplit = d.T;
    if (tsplit < 1) tsplit = 1;
    const int tlen = (int)cdiv(d.T, tsplit);
    tsplit = (int)cdiv(d.T, tlen);
    const size_t lds = 2ul * 4 * 9 * kFS * sizeof(double2);
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
      (void)hipFuncSetAttribute((const void*)su3_force_slice_kernel<KICK, kFS, kVar, kLpt>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL((su3_force_slice_kernel<KICK, kFS, kVar, kLpt>),
                       dim3((unsigned)((long)nb * nsb * tsplit)), dim3(4 * kFS / kLpt), lds, st, xn, d, nsb,
                       tsplit, tuning().xcd_swizzle, coef, out);
    return;
  }
  if (tuning().force_tile) {
    const long ntile = cdiv(d.V, 64);
    const dim3 grid((unsigned)(nb * ntile)), block(kBlock);
    const int swz = tuning().xcd_swizzle;
    switch (tuning().force_occ) {
      case 4: hipLaunchKernelGGL((su3_force_tile_kernel<KICK, 4>), grid, block, 0, st, xn, d, ntile, swz, coef, out); break;
      case 3: hipLaunchKernelGGL((su3_force_tile_kernel<KICK, 3>), grid, block, 0, st, xn, d, ntile, swz, coef, out); break;
      default: hipLaunchKernelGGL((su3_force_tile_kernel<KICK, 2>), grid, block, 0, st, xn, d, ntile, swz, coef, out); break;
    }
    return;
  }
  const dim3 grid((unsigned)(nb * nblk * 4)), block(kBlock);
  const int swz = tuning().xcd_swizzle;
  switch (tuning().force_occ) {
    case 4: hipLaunchKernelGGL((su3_force_kernel<KICK, 4>), grid, block, 0, st, xn, d, nblk, swz, coef, out); break;
    case 3: hipLaunchKernelGGL((su3_force_kernel<KICK, 3>), grid, block, 0, st, xn, d, nblk, swz, coef, out); break;
    default: hipLaunchKernelGGL((su3_force_kernel<KICK, 2>), grid, block, 0, st, xn, d, nblk, swz, coef, out); break;
  }
}

static bool dims_ok(int nb, int T, int X, int Y, int Z) {
  return nb > 0 && T > 0 && X > 0 && Y > 0 && Z > 0 && (double)T * X * Y * Z * 36.0 < 2.0e9;
}

extern "C" {

int l2q_kernel_name(const char* entry, int T, int X, int Y, int Z, char* buf, size_t buf_bytes) {
  L2Q_REQUIRE(entry && buf && buf_bytes > 0, L2Q_EINVAL, "null pointer");
  const int Vs = X * Y * Z;
  const Tuning& t = tuning();
  buf[0] = 0;
  if (!strcmp(entry, "l2q_su3_plaq_reduce")) {
    const Dims dq{T, X, Y, Z, T * X * Y * Z};
    if (t.plaq_sweep == 3 && plaq_nu_applicable(dq))
      snprintf(buf, buf_bytes, "su3_plaq_nu_kernel<%d>", force_nu_inmask(dq));
    else if (t.plaq_sweep >= 2 && Vs % kSlice == 0)
      snprintf(buf, buf_bytes, "su3_plaq_slice_kernel<%s>", (kSlice % (Y * Z)) == 0 ? "true" : "false");
    else if (t.plaq_sweep == 1) snprintf(buf, buf_bytes, "su3_plaq_sweep_kernel<%d>", t.plaq_occ);
    else snprintf(buf, buf_bytes, "su3_plaq_kernel<%d>", t.plaq_occ);
  } else if (!strcmp(entry, "l2q_su3_force") || !strcmp(entry, "l2q_su3_force_kick")) {
    const bool kick = !strcmp(entry, "l2q_su3_force_kick");
    const int fs = kick ? kFSKick : kFSPlain;
    const Dims dd{T, X, Y, Z, T * X * Y * Z};
    if (!kick && t.force_tile == 7 && force_plaq_applicable(dd))
      snprintf(buf, buf_bytes, "su3_force_plaq_kernel");
    else if (t.force_tile == 6 && force_pair_applicable(dd))
      snprintf(buf, buf_bytes, "su3_force_pair_kernel<%d, %d>", kick ? 1 : 0, force_pair_inmask(dd));
    else if (t.force_tile >= 5 && force_link_applicable(dd))
      snprintf(buf, buf_bytes, "su3_force_link_kernel<%d, %d>", kick ? 1 : 0, force_link_inmask(dd));
    else if (t.force_tile >= 4 && force_nu_applicable(dd) && (kick ? nu_preferred<true>(dd) : nu_preferred<false>(dd)))
      snprintf(buf, buf_bytes, "su3_force_nu_kernel<%d, %d>", kick ? 1 : 0, force_nu_inmask(dd));
    else if (t.force_tile == 3 && Vs % 64 == 0)
      snprintf(buf, buf_bytes, "su3_force_rows_kernel<%d, %d>", kick ? 1 : 0, force_rows_inmask(Dims{T, X, Y, Z, T * X * Y * Z}));
    else if (t.force_tile >= 2 && Vs % fs == 0)
      snprintf(buf, buf_bytes, "su3_force_slice_kernel<%s, %d, %d, %d>", kick ? "true" : "false", fs,
               kick ? 2 : 0, kick ? 1 : kLptPlain);
    else if (t.force_tile) snprintf(buf, buf_bytes, "su3_force_tile_kernel<%s, %d>", kick ? "true" : "false", t.force_occ);
    else snprintf(buf, buf_bytes, "su3_force_kernel<%s, %d>", kick ? "true" : "false", t.force_occ);
  } else if (!strcmp(entry, "l2q_vnet_heads_vupdate_sliced_f64")) {
    snprintf(buf, buf_bytes, "heads_sliced_kernel");
  } else if (!strncmp(entry, "l2q_vnet_heads_vupdate", 22)) {
    // (for shapes with whole 16-wide K-slabs, which every SU(3) vnet has)
    snprintf(buf, buf_bytes, "%s", t.heads_dma ? "fused_heads_dma_kernel" : "fused_heads_vupdate_kernel");
  } else if (!strcmp(entry, "l2q_gemm_h")) {
    // (T, X, Y) carry (M, N, K) here: "hipblaslt" when the plain-layer route of gemm_lt.hip takes the shape
    if (gemm_h_lt_shape(T, X, (long)Y) && gemm_h_lt_available()) snprintf(buf, buf_bytes, "hipblaslt");
  } else if (!strcmp(entry, "l2q_gemm_sliced_f64")) {
    snprintf(buf, buf_bytes, "gemm_sliced_kernel");
  } else if (!strcmp(entry, "l2q_gemm_f64")) {
    snprintf(buf, buf_bytes, "%s", t.heads_dma ? "gemm_dma_f64_kernel" : "gemm_nt_kernel");
  }
  return L2Q_OK;
}

size_t l2q_reduce_ws_bytes(int nb, long n_per_chain) {
  if (nb <= 0 || n_per_chain <= 0) return 0;
  // two partial arrays of up to 2 doubles per 256-item block
  return (size_t)nb * (size_t)cdiv(n_per_chain, kBlock) * 4 * sizeof(double) + 256;
}

int l2q_su3_plaq_reduce(const void* xn, int nb, int T, int X, int Y, int Z, double* out, void* ws,
                        size_t ws_bytes, void* stream) {
  L2Q_REQUIRE(xn && out && ws, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(dims_ok(nb, T, X, Y, Z), L2Q_EINVAL, "non-positive size");
  Dims d{T, X, Y, Z, T * X * Y * Z};
  hipStream_t st = (hipStream_t)stream;
  double* partial = (double*)ws;
  const int swz = tuning().xcd_swizzle;
  if (tuning().plaq_sweep == 3 && plaq_nu_applicable(d)) {
    // six planes of a site over six wavefronts, 3 wavefronts per SIMD (su3_plaq_nu.hip)
    const long per_chain = plaq_nu_per_chain(d, nb);
    L2Q_REQUIRE(ws_bytes >= (size_t)nb * per_chain * 2 * sizeof(double), L2Q_ESHAPE,
                "workspace too small");
    launch_plaq_nu((const double2*)xn, d, nb, partial, st);
    launch_finalize(partial, out, nb, per_chain, 2, 1.0, 0.0, st);
    return check_launch("l2q_su3_plaq_reduce");
  }
  if (tuning().plaq_sweep >= 2 && (X * Y * Z) % kSlice == 0) {
    const int Vs = X * Y * Z;
    const int nsb = Vs / kSlice;
    int tsplit = (int)cdiv(1024, (long)nb * nsb);      // keep >= ~1024 workgroups
    if (tsplit > T) tsplit = T;
    if (tsplit < 1) tsplit = 1;
    const int tlen = (int)cdiv(T, tsplit);
    tsplit = (int)cdiv(T, tlen);
    const long per_chain = (long)nsb * tsplit;
    L2Q_REQUIRE(ws_bytes >= (size_t)nb * per_chain * 2 * sizeof(double), L2Q_ESHAPE,
                "workspace too small");
    // a 128-site tile made of whole (y,z) rows with Y*Z | 128 keeps +y, +z inside the tile
    const bool yz = (kSlice % (Y * Z)) == 0;
    if (yz) hipLaunchKernelGGL(su3_plaq_slice_kernel<true>, dim3((unsigned)(nb * per_chain)), dim3(kSlice), 0, st,
                               (const double2*)xn, d, nsb, tsplit, swz, partial);
    else hipLaunchKernelGGL(su3_plaq_slice_kernel<false>, dim3((unsigned)(nb * per_chain)), dim3(kSlice), 0, st,
                            (const double2*)xn, d, nsb, tsplit, swz, partial);
    launch_finalize(partial, out, nb, per_chain, 2, 1.0, 0.0, st);
    return check_launch("l2q_su3_plaq_reduce");
  }
  if (tuning().plaq_sweep == 1) {
    const int Vs = X * Y * Z;
    const int nsb = (int)cdiv(Vs, kBlock);
    int tsplit = (int)cdiv(1024, (long)nb * nsb);      // keep >= ~1024 workgroups in flight
    if (tsplit > T) tsplit = T;
    if (tsplit < 1) tsplit = 1;
    const int tlen = (int)cdiv(T, tsplit);
    tsplit = (int)cdiv(T, tlen);
    const long per_chain = (long)nsb * tsplit;
    L2Q_REQUIRE(ws_bytes >= (size_t)nb * per_chain * 2 * sizeof(double), L2Q_ESHAPE,
                "workspace too small");
    const dim3 grid((unsigned)(nb * per_chain)), block(kBlock);
    switch (tuning().plaq_occ) {
      case 4: hipLaunchKernelGGL(su3_plaq_sweep_kernel<4>, grid, block, 0, st, (const double2*)xn, d, nsb, tsplit, swz, partial); break;
      case 3: hipLaunchKernelGGL(su3_plaq_sweep_kernel<3>, grid, block, 0, st, (const double2*)xn, d, nsb, tsplit, swz, partial); break;
      default: hipLaunchKernelGGL(su3_plaq_sweep_kernel<2>, grid, block, 0, st, (const double2*)xn, d, nsb, tsplit, swz, partial); break;
    }
    launch_finalize(partial, out, nb, per_chain, 2, 1.0, 0.0, st);
    return check_launch("l2q_su3_plaq_reduce");
  }
  const long nblk = cdiv(d.V, kBlock);
  L2Q_REQUIRE(ws_bytes >= (size_t)nb * nblk * 2 * sizeof(double), L2Q_ESHAPE, "workspace too small");
  const dim3 grid((unsigned)(nb * nblk)), block(kBlock);
  switch (tuning().plaq_occ) {
    case 4: hipLaunchKernelGGL(su3_plaq_kernel<4>, grid, block, 0, st, (const double2*)xn, d, nblk, swz, partial); break;
    case 3: hipLaunchKernelGGL(su3_plaq_kernel<3>, grid, block, 0, st, (const double2*)xn, d, nblk, swz, partial); break;
    default: hipLaunchKernelGGL(su3_plaq_kernel<2>, grid, block, 0, st, (const double2*)xn, d, nblk, swz, partial); break;
  }
  launch_finalize(partial, out, nb, nblk, 2, 1.0, 0.0, st);
  return check_launch("l2q_su3_plaq_reduce");
}

int l2q_su3_wilson_loops(const void* xn, int nb, int T, int X, int Y, int Z, void* out, void* stream) {
  L2Q_REQUIRE(xn && out, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(dims_ok(nb, T, X, Y, Z), L2Q_EINVAL, "non-positive size");
  Dims d{T, X, Y, Z, T * X * Y * Z};
  const long nblk = cdiv(d.V, kBlock);
  hipLaunchKernelGGL(su3_wloops_kernel, dim3((unsigned)(nb * nblk)), dim3(kBlock), 0, (hipStream_t)stream,
                     (const double2*)xn, d, nblk, (long)nb, (double2*)out);
  return check_launch("l2q_su3_wilson_loops");
}

int l2q_su3_plaq_planes(const void* xn, int nb, int T, int X, int Y, int Z, double* out, void* ws,
                        size_t ws_bytes, void* stream) {
  L2Q_REQUIRE(xn && out && ws, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(dims_ok(nb, T, X, Y, Z), L2Q_EINVAL, "non-positive size");
  Dims d{T, X, Y, Z, T * X * Y * Z};
  const long nblk = cdiv(d.V, kBlock);
  L2Q_REQUIRE(ws_bytes >= (size_t)nb * nblk * 12 * sizeof(double), L2Q_ESHAPE, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(su3_plaq_planes_kernel, dim3((unsigned)(nb * nblk)), dim3(kBlock), 0, st,
                     (const double2*)xn, d, nblk, (double*)ws);
  launch_finalize((const double*)ws, out, nb, nblk, 12, 1.0, 0.0, st);
  return check_launch("l2q_su3_plaq_planes");
}

int l2q_diff_norm2_reduce(const double* a, const double* b, int nb, long n, double* out, void* ws,
                          size_t ws_bytes, void* stream) {
  L2Q_REQUIRE(a && b && out && ws, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && n > 0, L2Q_EINVAL, "non-positive size");
  const long nblk = cdiv(n, 4L * kBlock);
  L2Q_REQUIRE(ws_bytes >= (size_t)nb * nblk * sizeof(double), L2Q_ESHAPE, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(diff_norm2_kernel, dim3((unsigned)(nb * nblk)), dim3(kBlock), 0, st, a, b, n,
                     nblk, (double*)ws);
  launch_finalize((const double*)ws, out, nb, nblk, 1, 1.0, 0.0, st);
  return check_launch("l2q_diff_norm2_reduce");
}

int l2q_su3_force(const void* xn, double beta, void* fn, int nb, int T, int X, int Y, int Z,
                  void* stream) {
  L2Q_REQUIRE(xn && fn, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(dims_ok(nb, T, X, Y, Z), L2Q_EINVAL, "non-positive size");
  L2Q_REQUIRE(xn != fn, L2Q_EINVAL, "force output must not alias the gauge field");
  Dims d{T, X, Y, Z, T * X * Y * Z};
  const long nblk = cdiv(d.V, kBlock);
  launch_force<false>((const double2*)xn, d, nb, nblk, beta / 3.0, (double2*)fn, (hipStream_t)stream);
  return check_launch("l2q_su3_force");
}

int l2q_su3_force_kick(const void* xn, double beta, double coef, void* vn, int nb, int T, int X,
                       int Y, int Z, void* stream) {
  L2Q_REQUIRE(xn && vn, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(dims_ok(nb, T, X, Y, Z), L2Q_EINVAL, "non-positive size");
  Dims d{T, X, Y, Z, T * X * Y * Z};
  const long nblk = cdiv(d.V, kBlock);
  launch_force<true>((const double2*)xn, d, nb, nblk, coef * beta / 3.0, (double2*)vn, (hipStream_t)stream);
  return check_launch("l2q_su3_force_kick");
}

int l2q_su3_force_kick_to(const void* xn, double beta, double coef, const void* v_in, void* v_out,
                          int nb, int T, int X, int Y, int Z, void* stream) {
  L2Q_REQUIRE(xn && v_in && v_out, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(dims_ok(nb, T, X, Y, Z), L2Q_EINVAL, "non-positive size");
  Dims d{T, X, Y, Z, T * X * Y * Z};
  hipStream_t st = (hipStream_t)stream;
  if (v_in != v_out && tuning().force_tile == 6 && force_pair_applicable(d)) {
    launch_force_pair(true, (const double2*)xn, d, nb, coef * beta / 3.0, (double2*)v_out, st,
                      (const double2*)v_in);
    return check_launch("l2q_su3_force_kick_to");
  }
  if (v_in != v_out && tuning().force_tile >= 5 && force_link_applicable(d)) {
    launch_force_link(true, (const double2*)xn, d, nb, coef * beta / 3.0, (double2*)v_out, st,
                      (const double2*)v_in);
    return check_launch("l2q_su3_force_kick_to");
  }
  // the other kernels of the family update in place: copy first
  if (v_in != v_out)
    (void)hipMemcpyAsync(v_out, v_in, (size_t)nb * 36 * d.V * sizeof(double2), hipMemcpyDeviceToDevice, st);
  return l2q_su3_force_kick(xn, beta, coef, v_out, nb, T, X, Y, Z, stream);
}

int l2q_su3_expm_mul(const void* xn, const void* vn, double eps, const float* mask_n,
                     int complement, void* out, int nb, long V, void* stream) {
  L2Q_REQUIRE(xn && vn && out, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && V > 0 && V <= 200000000L, L2Q_EINVAL, "bad size");
  const long nblk = cdiv(V, kBlock);
  hipLaunchKernelGGL((su3_expm_mul_kernel<false, false>), dim3((unsigned)(nb * 4L * nblk)), dim3(kBlock),
                     0, (hipStream_t)stream, (const double2*)xn, (const double2*)vn, eps, mask_n,
                     complement, (double2*)out, (double*)nullptr, (int)V, nblk, 0);
  return check_launch("l2q_su3_expm_mul");
}

int l2q_su3_expm_mul2(const void* xn, const void* vn, double eps, const float* mask_n,
                      int complement_first, void* out, int nb, long V, void* stream) {
  L2Q_REQUIRE(xn && vn && out && mask_n, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && V > 0 && V <= 200000000L, L2Q_EINVAL, "bad size");
  const long nblk = cdiv(V, kBlock);
  hipLaunchKernelGGL((su3_expm_mul_kernel<true, false>), dim3((unsigned)(nb * 4L * nblk)), dim3(kBlock),
                     0, (hipStream_t)stream, (const double2*)xn, (const double2*)vn, eps, mask_n,
                     complement_first, (double2*)out, (double*)nullptr, (int)V, nblk, 0);
  return check_launch("l2q_su3_expm_mul2");
}

int l2q_su3_expm_mul2_vec8(const void* xn, const void* vn, double eps, const float* mask_n,
                           int complement_first, void* out, double* vec, int nb, long V,
                           void* stream) {
  L2Q_REQUIRE(xn && vn && out && mask_n && vec, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && V > 0 && V <= 200000000L, L2Q_EINVAL, "bad size");
  const long nblk = cdiv(V, kBlock);
  hipLaunchKernelGGL((su3_expm_mul_kernel<true, true>), dim3((unsigned)(nb * 4L * nblk)), dim3(kBlock),
                     0, (hipStream_t)stream, (const double2*)xn, (const double2*)vn, eps, mask_n,
                     complement_first, (double2*)out, vec, (int)V, nblk, 0);
  return check_launch("l2q_su3_expm_mul2_vec8");
}

int l2q_su3_project_su(const void* in, void* out, long nfields, long V, void* stream) {
  L2Q_REQUIRE(in && out, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nfields > 0 && V > 0 && V <= 200000000L, L2Q_EINVAL, "bad size");
  const long nblk = cdiv(V, kBlock);
  hipLaunchKernelGGL(su3_project_kernel<0>, dim3((unsigned)(nfields * nblk)), dim3(kBlock), 0,
                     (hipStream_t)stream, (const double2*)in, (double2*)out, (double*)nullptr, (int)V,
                     nblk);
  return check_launch("l2q_su3_project_su");
}

int l2q_su3_projsu_vec8(const void* in, double* vec, long nfields, long V, void* stream) {
  L2Q_REQUIRE(in && vec, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nfields > 0 && V > 0 && V <= 200000000L, L2Q_EINVAL, "bad size");
  const long nblk = cdiv(V, kBlock);
  hipLaunchKernelGGL(su3_project_kernel<1>, dim3((unsigned)(nfields * nblk)), dim3(kBlock), 0,
                     (hipStream_t)stream, (const double2*)in, (double2*)nullptr, vec, (int)V, nblk);
  return check_launch("l2q_su3_projsu_vec8");
}

int l2q_su3_project_tah(const void* in, void* out, long nfields, long V, void* stream) {
  L2Q_REQUIRE(in && out, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nfields > 0 && V > 0 && V <= 200000000L, L2Q_EINVAL, "bad size");
  const long nblk = cdiv(V, kBlock);
  hipLaunchKernelGGL(su3_project_kernel<2>, dim3((unsigned)(nfields * nblk)), dim3(kBlock), 0,
                     (hipStream_t)stream, (const double2*)in, (double2*)out, (double*)nullptr, (int)V,
                     nblk);
  return check_launch("l2q_su3_project_tah");
}

int l2q_su3_project_u(const void* in, void* out, long nfields, long V, void* stream) {
  L2Q_REQUIRE(in && out, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nfields > 0 && V > 0 && V <= 200000000L, L2Q_EINVAL, "bad size");
  const long nblk = cdiv(V, kBlock);
  hipLaunchKernelGGL(su3_project_kernel<3>, dim3((unsigned)(nfields * nblk)), dim3(kBlock), 0,
                     (hipStream_t)stream, (const double2*)in, (double2*)out, (double*)nullptr, (int)V,
                     nblk);
  return check_launch("l2q_su3_project_u");
}

int l2q_su3_mul(const void* a, const void* b, int adjoint_a, int adjoint_b, void* out,
                long nfields, long V, void* stream) {
  L2Q_REQUIRE(a && b && out, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nfields > 0 && V > 0 && V <= 200000000L, L2Q_EINVAL, "bad size");
  const long nblk = cdiv(V, kBlock);
  hipLaunchKernelGGL(su3_mul_kernel, dim3((unsigned)(nfields * nblk)), dim3(kBlock), 0,
                     (hipStream_t)stream, (const double2*)a, (const double2*)b, adjoint_a, adjoint_b,
                     (double2*)out, (int)V, nblk);
  return check_launch("l2q_su3_mul");
}

int l2q_su3_kinetic_reduce(const void* vn, int nb, long V, double* out, void* ws, size_t ws_bytes,
                           void* stream) {
  L2Q_REQUIRE(vn && out && ws, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && V > 0 && V <= 200000000L, L2Q_EINVAL, "bad size");
  const long n = 36 * V;
  const long nblk = cdiv(n, 4L * kBlock);
  L2Q_REQUIRE(ws_bytes >= (size_t)nb * nblk * sizeof(double), L2Q_ESHAPE, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(su3_norm2_kernel, dim3((unsigned)(nb * nblk)), dim3(kBlock), 0, st,
                     (const double2*)vn, n, nblk, (double*)ws);
  // 0.5 * (sum |p|^2 - 8 * 4V)
  launch_finalize((const double*)ws, out, nb, nblk, 1, 0.5, -0.5 * 8.0 * 4.0 * (double)V, st);
  return check_launch("l2q_su3_kinetic_reduce");
}

int l2q_su3_assemble_tah(const double* normals, void* vn, long nfields, long V, void* stream) {
  L2Q_REQUIRE(normals && vn, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nfields > 0 && V > 0 && V <= 200000000L, L2Q_EINVAL, "bad size");
  const long nblk = cdiv(V, kBlock);
  hipLaunchKernelGGL(su3_assemble_tah_kernel, dim3((unsigned)(nfields * nblk)), dim3(kBlock), 0,
                     (hipStream_t)stream, normals, (double2*)vn, V, nblk, nfields * V);
  return check_launch("l2q_su3_assemble_tah");
}

int l2q_su3_check_su(const void* xn, int nb, long V, double* out, void* ws, size_t ws_bytes,
                     void* stream) {
  L2Q_REQUIRE(xn && out && ws, L2Q_EINVAL, "null pointer");
  L2Q_REQUIRE(nb > 0 && V > 0 && V <= 200000000L, L2Q_EINVAL, "bad size");
  const long nblk = cdiv(4 * V, kBlock);
  L2Q_REQUIRE(ws_bytes >= (size_t)nb * nblk * 2 * sizeof(double), L2Q_ESHAPE, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  double* psum = (double*)ws;
  double* pmax = psum + (size_t)nb * nblk;
  hipLaunchKernelGGL(su3_check_kernel, dim3((unsigned)(nb * nblk)), dim3(kBlock), 0, st,
                     (const double2*)xn, V, nblk, psum, pmax);
  hipLaunchKernelGGL(check_finalize_kernel, dim3(nb), dim3(kBlock), 0, st, psum, pmax, nblk,
                     4.0 * (double)V, out);
  return check_launch("l2q_su3_check_su");
}

}  // extern "C"
