/* l2q.h -- C ABI of libl2q.so: hand-written gfx950 (MI355X / CDNA4) kernels for the L2HMC /
 * HMC leapfrog integrator over 2D U(1) and 4D SU(3) lattice gauge fields.
 *
 * The reference (saforem2/l2hmc-qcd) is 100 % Python and has no FFI boundary; its hot path
 * is a sequence of ATen calls behind the Python classes `Dynamics`, `LatticeSU3`, `LatticeU1`,
 * `SU3`, `U1Phase`, `LeapfrogLayer`.  Each entry point below replaces one such ATen sequence;
 * the reference lines it replaces are cited (paths relative to src/l2hmc/ of the reference).
 * The Python classes of the same names in `l2hmc-qcd_amd/l2hmc/` call these through ctypes.
 *
 * Conventions
 *   - extern "C", plain C types.  All pointers are DEVICE pointers owned by the caller
 *     (normally PyTorch's allocator).  Nothing is allocated and nothing synchronises inside;
 *     work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream).
 *   - return 0 on success, negative L2Q_E* on failure; l2q_last_error() gives the text.
 *   - "native" SU(3) field layout (what every lattice kernel consumes and produces):
 *         xn[chain][mu][e][site]   complex128 (re, im interleaved, 16 B)
 *     with e = 3*row + col of the 3x3 link matrix and site = ((t*X + x)*Y + y)*Z + z.
 *     One wavefront reading entry e of 64 consecutive sites is a single coalesced 1 KiB load.
 *     The reference's public layout x[chain][mu][t][x][y][z][3][3] is converted at the API
 *     edge by l2q_su3_pack / l2q_su3_unpack; a trajectory stays in native layout throughout.
 *   - native flat index of a real per-entry quantity (masks, s/t/q network heads):
 *         j_n = (mu*9 + e)*V + site        (reference: j = (mu*V + site)*9 + e)
 *     native index of the 8-component algebra vector (vnet input):
 *         k_n = (mu*8 + a)*V + site        (reference: k = (mu*V + site)*8 + a)
 *   - U(1) fields keep the reference layout x[chain][2][T][X] (already site-contiguous).
 *   - per-chain reductions are order-stable (fixed tree, no atomics) so that dH and the
 *     accept/reject mask are reproducible run to run.
 */
#ifndef L2Q_H_
#define L2Q_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define L2Q_OK 0
#define L2Q_EINVAL (-1) /* bad argument (null pointer, non-positive size, bad enum) */
#define L2Q_ESHAPE (-2) /* inconsistent shapes / workspace too small */
#define L2Q_EHIP (-3)   /* HIP runtime error at launch */

/* activation enum for the network kernels (network/pytorch/network.py:40-46) */
#define L2Q_ACT_NONE 0
#define L2Q_ACT_TANH 1
#define L2Q_ACT_RELU 2
#define L2Q_ACT_LEAKY_RELU 3
#define L2Q_ACT_ELU 4
#define L2Q_ACT_SWISH 5

/* 16-bit operand types of l2q_gemm_h */
#define L2Q_HALF_F16 0
#define L2Q_HALF_BF16 1

const char* l2q_last_error(void);
int l2q_version(void);
/* performance knobs (results never depend on them), kept PER DEVICE: the call applies to the
 * calling thread's current HIP device -- the library's only state besides the last-error text
 * is this per-device table: "plaq_occ" / "force_occ" in {2,3,4} pick
 * the register-allocation variant (min waves per SIMD) of the stencil kernels; "xcd_swizzle"
 * in {0,1}; "force_tile" in 0..7 picks the SU(3) force kernel design (5, the default: thread per link,
 * su3_force_link.hip; 7: plaquettes shared between their four links, su3_force_plaq.hip; the others
 * are the earlier designs kept as A/B variants), "plaq_sweep" in 0..3 the plaquette kernel.  Results agree to
 * rounding across variants.  Returns the previous value, or L2Q_EINVAL for an unknown key/value.
 * (The Python binding applies L2Q_TUNING="key=value,..." from the environment when it loads the library.) */
int l2q_set_tuning(const char* key, int value);
/* Name (template instantiation as rocprofv3 prints it, without the l2q:: prefix) of the device
 * kernel that `entry` ("l2q_su3_force", "l2q_su3_force_kick", "l2q_su3_plaq_reduce",
 * "l2q_vnet_heads_vupdate[_pair]_f64", "l2q_gemm_f64": kernel family only) dispatches
 * for a T x X x Y x Z lattice under the current tuning; "" for entry points with a single
 * kernel.  entry "l2q_gemm_h" with (T, X, Y) = (M, N, K): "hipblaslt" when that plain 16-bit layer goes to the
 * vendor library (gemm_lt.hip: library present, shape taken), "" when it runs on this build's kernels.  Lets a profile (rocprofv3 --pmc) be matched to the build that is running. */
int l2q_kernel_name(const char* entry, int T, int X, int Y, int Z, char* buf, size_t buf_bytes);
/* Select and check the device this process drives (one process per GPU): hipSetDevice(device),
 * refuses anything that is not gfx950, touches the device's knob table.  Returns `device` or a
 * negative error.  Optional -- every entry point works on the caller's current HIP device; this is
 * the explicit per-device handle SURVEY.md section 8(b)(ii) lists. */
int l2q_init(int device);

/* ---------------------------------------------------------------- the one collective of the path
 * Sum of the flat training-gradient buffer over ranks: the reference wraps the dynamics in
 * DistributedDataParallel (trainers/pytorch/trainer.py:246-257; bucketed all-reduce during
 * loss.backward(), :1296-1304) on the process group of utils/dist.py:126-144.  Thin wrapper over
 * RCCL (xGMI within a node); librccl is resolved at run time, so the library loads without it.
 * Bootstrap like NCCL: rank 0 calls l2q_comm_unique_id, the L2Q_COMM_ID_BYTES bytes reach the other
 * ranks out of band (a torch.distributed / file / env broadcast), every rank calls l2q_comm_init
 * with its HIP device current.  l2q_allreduce_grads is in place, enqueued on `stream`, elem_bytes
 * 8 = fp64, 4 = fp32 (one call per dtype group of the gradient arena).  Sampling never calls it. */
#define L2Q_COMM_ID_BYTES 128
int l2q_comm_unique_id(void* id_out);
int l2q_comm_init(const void* id, int nranks, int rank, void** comm_out);
int l2q_allreduce_grads(void* comm, void* grad, long n, int elem_bytes, void* stream);
int l2q_comm_destroy(void* comm);
/* teardown without a handshake with the peers (ncclCommAbort; falls back to destroy): for garbage collection
 * and interpreter shutdown, when the other ranks may be gone */
int l2q_comm_abort(void* comm);
/* version code of the RCCL library that was resolved (ncclGetVersion: major * 10000 + minor * 100 + patch), or a
 * negative error.  The wrapper declares the 2.x ABI by hand and refuses a library that reports another major. */
int l2q_comm_version(void);

/* bytes of scratch the reductions need for `nb` chains of `n_per_chain` work items */
size_t l2q_reduce_ws_bytes(int nb, long n_per_chain);

/* ---------------------------------------------------------------- layout conversion */
/* batched transpose in[batch][rows][cols] -> out[batch][cols][rows], elem_bytes in {4,8,16}.
 * pack:   reference [.., site, e] -> native [.., e, site]  (rows = V, cols = 9 or 8)
 * unpack: the same call with rows/cols swapped. */
int l2q_transpose(const void* in, void* out, long batch, int rows, int cols, int elem_bytes,
                  void* stream);
/* x[nb][4][V][3][3] c128 <-> xn[nb][4][9][V] c128 */
int l2q_su3_pack(const void* x_ref, void* x_nat, int nb, long V, void* stream);
int l2q_su3_unpack(const void* x_nat, void* x_ref, int nb, long V, void* stream);
/* x_ref[c] = unpack(mask[c] != 0 ? a_nat[c] : b_nat[c]): the accept / reject select of a transition
 * (x_out = ma x_prop + mr x_init, dynamics.py:677-682) fused into the native -> reference transpose of its
 * result; mask [nb] float32 (the acc_mask of l2q_accept) */
int l2q_su3_unpack_select(const void* a_nat, const void* b_nat, const float* mask, void* x_ref, int nb,
                          long V, void* stream);

/* ---------------------------------------------------------------- SU(3) lattice kernels */
/* Per-chain plaquette sums: out[c][0] = sum_{sites, 6 planes} Re tr P, out[c][1] = Im.
 * Replaces LatticeSU3._wilson_loops + the reductions in action/_plaquettes/_sin_charges/
 * _int_charges (lattice/su3/pytorch/lattice.py:157-269):
 *   action = -(beta/3) out[c][0];  plaqs = out[c][0]/(18 V);  sinQ = out[c][1]/(18 V);
 *   intQ = out[c][1]/(32 pi^2).   Algorithmic traffic: 576 B per (chain, site). */
int l2q_su3_plaq_reduce(const void* xn, int nb, int T, int X, int Y, int Z, double* out,
                        void* ws, size_t ws_bytes, void* stream);
/* The same sums kept per plane: out[c][p][0|1], p = 0..5 in the reference's order
 * (u,v) = (1,0),(2,0),(2,1),(3,0),(3,1),(3,2) -- what LatticeLoss._plaq_loss reduces to
 * (loss/pytorch/loss.py:57-70).  ws >= nb * ceil(V/256) * 12 doubles. */
int l2q_su3_plaq_planes(const void* xn, int nb, int T, int X, int Y, int Z, double* out, void* ws,
                        size_t ws_bytes, void* stream);
/* The trace field itself: out[p][c][site] = tr P_p(site), complex128, planes ordered as above -- the tensor
 * [6, nb, T, X, Y, Z] that LatticeSU3.wilson_loops returns (lattice/su3/pytorch/lattice.py:157-174, 242-244)
 * for callers that reduce it themselves (loss/pytorch/loss.py:57-110). */
int l2q_su3_wilson_loops(const void* xn, int nb, int T, int X, int Y, int Z, void* out, void* stream);
/* out[c] = sum_j (a[c][j] - b[c][j])^2 over n doubles (rmse loss, loss.py:131-148). */
int l2q_diff_norm2_reduce(const double* a, const double* b, int nb, long n, double* out, void* ws,
                          size_t ws_bytes, void* stream);
/* F = (beta/3) TAH(U_mu(x) * sum of 6 staples), written in native layout.
 * Replaces LatticeSU3.grad_action (autograd + projectTAH, lattice.py:299-308).
 * Algorithmic traffic: 1152 B per (chain, site). */
int l2q_su3_force(const void* xn, double beta, void* fn, int nb, int T, int X, int Y, int Z,
                  void* stream);
/* fused plain-HMC half-kick: v += coef * F(x)   (dynamics/pytorch/dynamics.py:903-911,
 * coef = -eps/2); F is never materialised. */
int l2q_su3_force_kick(const void* xn, double beta, double coef, void* vn, int nb, int T,
                       int X, int Y, int Z, void* stream);
/* The same kick out of place: v_out = v_in + coef * (beta/3) TAH(U A)  (v_in == v_out is the call
 * above).  The first kick of a plain-HMC trajectory uses it to leave the momentum it was handed
 * untouched without copying it. */
int l2q_su3_force_kick_to(const void* xn, double beta, double coef, const void* v_in, void* v_out,
                          int nb, int T, int X, int Y, int Z, void* stream);
/* out = keep (.) x + expm(eps * v) @ ((1 - keep) (.) x), element-wise 0/1 mask on matrix
 * entries.  mask_n: float32 [36 V] in native order or NULL (keep = 0 everywhere: the plain
 * `update_gauge`, group/su3/pytorch/group.py:45-50).  keep = mask if !complement else 1-mask.
 * Replaces Dynamics._update_x_fwd/_bwd SU3 branch (dynamics.py:1420-1425, 1468-1474).
 * `out` may alias xn. */
int l2q_su3_expm_mul(const void* xn, const void* vn, double eps, const float* mask_n,
                     int complement, void* out, int nb, long V, void* stream);
/* Both half-updates of one leapfrog step in one pass (dynamics.py:1199-1203 forward:
 * keep = m then keep = 1-m; :1221-1225 backward: keep = 1-m then keep = m, i.e.
 * complement_first = 1), sharing one expm(eps v).  Equal to two l2q_su3_expm_mul calls up to
 * FMA contraction (1e-16).  `out` may alias xn. */
int l2q_su3_expm_mul2(const void* xn, const void* vn, double eps, const float* mask_n,
                      int complement_first, void* out, int nb, long V, void* stream);
/* l2q_su3_expm_mul2 that also writes vec[nb*4][8][V] = su3_to_vec(projectSU(x')) of the updated
 * links (group.py:138-147): the vnet input of the v-update that follows the x-update in a
 * leapfrog step (dynamics.py:1154-1156, 1204-1206), formed while x' is in registers. */
int l2q_su3_expm_mul2_vec8(const void* xn, const void* vn, double eps, const float* mask_n,
                           int complement_first, void* out, double* vec, int nb, long V,
                           void* stream);
/* projectSU on every link (group/su3/pytorch/utils.py:341-346); out may alias in. */
int l2q_su3_project_su(const void* in, void* out, long nfields, long V, void* stream);
/* su3_to_vec(projectSU(.)) -> vec[nfields][8][V] double (group.py:138-147); the vnet
 * GEMM A-operand (dynamics.py:1154-1156). */
int l2q_su3_projsu_vec8(const void* in, double* vec, long nfields, long V, void* stream);
/* projectU: x (x^H x)^(-1/2) without the determinant phase (utils.py:332-338). */
int l2q_su3_project_u(const void* in, void* out, long nfields, long V, void* stream);
/* out = op(a) op(b) per link, op = adjoint if flagged (SU3.mul, group.py:56-69). */
int l2q_su3_mul(const void* a, const void* b, int adjoint_a, int adjoint_b, void* out,
                long nfields, long V, void* stream);
/* projectTAH on every link (group.py:92-103); out may alias in. */
int l2q_su3_project_tah(const void* in, void* out, long nfields, long V, void* stream);
/* out[c] = 0.5 * sum_links (|p|_F^2 - 8)   (group.py:125-126) */
int l2q_su3_kinetic_reduce(const void* vn, int nb, long V, double* out, void* ws,
                           size_t ws_bytes, void* stream);
/* momenta from 8 standard-normal fields normals[8][nfields*V] in the reference's draw order
 * r3, r8, r01, r02, r12, i01, i02, i12 (group/su3/pytorch/utils.py:171-195). */
int l2q_su3_assemble_tah(const double* normals, void* vn, long nfields, long V, void* stream);
/* max over links of |x^H x - 1|_F^2 + |det x - 1|^2 and its mean, per chain
 * (checkSU, utils.py:376-391): out[c][0] = sqrt(mean/20), out[c][1] = sqrt(max/20). */
int l2q_su3_check_su(const void* xn, int nb, long V, double* out, void* ws, size_t ws_bytes,
                     void* stream);

/* ---------------------------------------------------------------- L2HMC momentum update */
/* Generalised v-update with real network heads s, t, q [nb][n] applied entry-wise:
 *   forward : v' = exp(eps s/2) v - (eps/2) (F exp(eps q) + t),  logdet[c] =  sum eps s/2
 *   backward: v' = exp(-eps s/2) (v + (eps/2) (F exp(eps q) + t)), logdet[c] = -sum eps s/2
 * (dynamics.py:1266-1297).  is_complex=1: v, F are complex128 [nb][n] (SU3, heads multiply
 * both parts, t adds to the real part); 0: real.  elem_bytes 8 (f64) or 4 (f32, U1).
 * v is updated in place. */
int l2q_v_update(void* v, const void* force, const void* s, const void* t, const void* q,
                 double eps, int forward, int is_complex, int elem_bytes, int nb, long n,
                 void* logdet, void* ws, size_t ws_bytes, void* stream);
/* The same update out of place: v_out = update(v_in) (v_in is not written; v_out == v_in is the
 * in-place call).  The training tape keeps the momentum BEFORE every update for its reverse sweep:
 * this saves the copy. */
int l2q_v_update_to(const void* v_in, void* v_out, const void* force, const void* s, const void* t,
                    const void* q, double eps, int forward, int is_complex, int elem_bytes, int nb,
                    long n, void* logdet, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------- accept / reject */
/* acc = exp(min(0, h_init - h_prop + sumlogdet)); mask = (acc > u) as float32 0/1
 * (dynamics.py:1065-1087).  elem_bytes of h/sumlogdet/acc/u: 8 or 4. */
int l2q_accept(const void* h_init, const void* h_prop, const void* sumlogdet, const void* u,
               void* acc, float* mask, int nb, int elem_bytes, void* stream);
/* out[c][:] = mask[c] ? a[c][:] : b[c][:]   (dynamics.py:677-682), row_bytes (multiple of 4)
 * per chain */
int l2q_select_rows(const void* a, const void* b, const float* mask, void* out, int nb,
                    long row_bytes, void* stream);
/* y = alpha * x over n doubles (momentum flip, dynamics.py:1001); y may alias x */
int l2q_scale_f64(const double* x, double alpha, double* y, long n, void* stream);

/* ---------------------------------------------------------------- dense network layers */
/* C[M][N] = epilogue( A[M][K] . W[N][K]^T (+ A2[M][K2] . W2[N][K2]^T) + bias[N] (+ bias2[N]) )
 * fp64 on v_mfma_f64_16x16x4_f64.  Replaces nn.Linear + activation / ScaledTanh in
 * LeapfrogLayer.forward (network/pytorch/network.py:430-451, 522-551):
 *   epilogue: y = act(z);  if (coeff) y = scale * exp(coeff[n]) * y  else y = scale * y.
 * A2/W2/bias2 may be NULL (K2 = 0).  ws: split-K scratch (l2q_gemm_ws_bytes). */
int l2q_gemm_f64(const double* A, const double* W, int M, int N, long K, const double* A2,
                 const double* W2, long K2, const double* bias, const double* bias2,
                 const double* coeff, double scale, int act, double* C, void* ws,
                 size_t ws_bytes, void* stream);
size_t l2q_gemm_ws_bytes(int M, int N, long K, long K2);
/* The same layer (fp64 operands and result) with the products rebuilt from exact int8 x int8 -> int32 digit
 * products on v_mfma_i32_16x16x64_i8 (csrc/gemm_sliced.hip) -- for the input layer of the
 * SU(3) vnet, whose K = 64 V runs to 2 x 131 072 at 8^4 (network/pytorch/network.py:430-451, xlayer + vlayer).
 * l2q_gemm_sliced_build turns one weight matrix W [N][K] into its digit image (l2q_gemm_sliced_bytes(N, K)
 * bytes, 256-byte aligned: 7 int8 planes in MFMA fragment order + one power-of-two scale per output row;
 * rebuild when W changes; *usable = 0 for a non-finite entry; the call synchronises the stream).
 * l2q_gemm_sliced_f64 slices the activations ON THE FLY against ONE exponent per operand: every entry of A must
 * satisfy |a| < 2^a_exp (A2: a2_exp) -- the vnet inputs su3_to_vec(projectSU(.)) do with a_exp = 2 --; a value
 * outside the range or a NaN turns the whole output into NaN (as a NaN operand would).  Needs M, N multiples of
 * 64 and K, K2 multiples of 64.  Results agree with l2q_gemm_f64 to the accumulation error of that kernel
 * (not bit for bit).  A2 / image2 may be NULL (K2 = 0). */
size_t l2q_gemm_sliced_bytes(int N, long K);
int l2q_gemm_sliced_build(const double* W, int N, long K, void* image, size_t image_bytes, int* usable,
                          void* stream);
size_t l2q_gemm_sliced_ws_bytes(int M, int N, long K, long K2);
int l2q_gemm_sliced_f64(const double* A, const void* image, long K, int a_exp, const double* A2,
                        const void* image2, long K2, int a2_exp, int M, int N, const double* bias,
                        const double* bias2, const double* coeff, double scale, int act, double* C, void* ws,
                        size_t ws_bytes, void* stream);
/* The three output heads of a LeapfrogLayer AND the generalised momentum update in one kernel
 * (network.py:547-551 + dynamics.py:1266-1297): for chain m, entry n
 *   s = cs[n] tanh(Z.Ws[n] + bs[n]);  t = scale_t (Z.Wt[n] + bt[n]);  q = cq[n] tanh(Z.Wq[n] + bq[n])
 *   v' and logdet as in l2q_v_update.   s, t, q are never written to memory.
 * cs / cq: per-entry scale nw.s * exp(coeff_s[n]) (NULL -> the scalar scale_s / scale_q).
 * Z [M][K] (K even), W* [N][K], v / force [M][N] real or complex128, logdet [M]. */
int l2q_vnet_heads_vupdate_f64(const double* Z, int M, int K, long N, const double* Ws,
                               const double* bs, const double* cs, double scale_s,
                               const double* Wt, const double* bt, double scale_t,
                               const double* Wq, const double* bq, const double* cq,
                               double scale_q, void* v, const void* force, int is_complex,
                               double eps, int forward, double* logdet, void* ws,
                               size_t ws_bytes, void* stream);
/* The same update out of place: the momentum is read from v_in and written to v_out (v_in == v_out
 * is the in-place call; any other overlap is undefined).  A trajectory's first update uses it to
 * leave the momentum it was handed untouched without copying it. */
int l2q_vnet_heads_vupdate_to_f64(const double* Z, int M, int K, long N, const double* Ws,
                                  const double* bs, const double* cs, double scale_s,
                                  const double* Wt, const double* bt, double scale_t,
                                  const double* Wq, const double* bq, const double* cq,
                                  double scale_q, const void* v_in, void* v_out, const void* force,
                                  int is_complex, double eps, int forward, double* logdet, void* ws,
                                  size_t ws_bytes, void* stream);
/* Two consecutive v-updates on the SAME x (closing update of leapfrog step k, opening update
 * of step k+1; optionally the merged trajectory's momentum flip v -> -v in between,
 * dynamics.py:1001) from ONE evaluation of the heads: update (eps1, forward1), [flip],
 * update (eps2, forward2).  logdet receives the sum of both updates' log-Jacobians. */
int l2q_vnet_heads_vupdate_pair_f64(const double* Z, int M, int K, long N, const double* Ws,
                                    const double* bs, const double* cs, double scale_s,
                                    const double* Wt, const double* bt, double scale_t,
                                    const double* Wq, const double* bq, const double* cq,
                                    double scale_q, void* v, const void* force, int is_complex,
                                    double eps1, int forward1, int flip_between, double eps2,
                                    int forward2, double* logdet, void* ws, size_t ws_bytes,
                                    void* stream);
/* The same pair of updates with what per-step metrics need of the state BETWEEN them
 * (dynamics.py:865-886 evaluates the Hamiltonian after every leapfrog step): logdet1 [M] = the first
 * update's log-Jacobian alone (logdet still receives the sum), vnorm2_mid [M] = sum |v|^2 of the
 * momentum after the first update (before the flip, which does not change it).  Needs K % 16 == 0. */
int l2q_vnet_heads_vupdate_pair_mid_f64(const double* Z, int M, int K, long N, const double* Ws,
                                        const double* bs, const double* cs, double scale_s,
                                        const double* Wt, const double* bt, double scale_t,
                                        const double* Wq, const double* bq, const double* cq,
                                        double scale_q, void* v, const void* force, int is_complex,
                                        double eps1, int forward1, int flip_between, double eps2,
                                        int forward2, double* logdet, double* logdet1,
                                        double* vnorm2_mid, void* ws, size_t ws_bytes, void* stream);
size_t l2q_vnet_heads_ws_bytes(int M, long N);
/* The same heads + momentum update with the fp64 products rebuilt from exact int8 x int8 -> int32 slice
 * products on v_mfma_i32_16x16x64_i8 (error-free slicing, csrc/heads_sliced.hip): an
 * inference path for K = 256 (the reference's default width of the last hidden layer of the SU(3) nets).
 * l2q_heads_sliced_build turns the three weight matrices [N][K] into the int8 slice image + per-entry
 * power-of-two scales (sliced: l2q_heads_sliced_bytes(K, N) bytes, 256-byte aligned; rebuild when the
 * weights change); *usable = 0 when a weight vector's dynamic range is too wide for the 54-bit fixed
 * point (mean |w| below 2^-6 of the largest): keep the fp64 kernel then.  The call synchronises the
 * stream.  l2q_vnet_heads_vupdate_sliced_f64 is the four fp64 entry points in one: v_in NULL = in
 * place; pair = 0 ignores flip_between / eps2 / forward2; logdet1 / vnorm2_mid non-NULL (pair only) =
 * the mid-point outputs.  Results agree with the fp64 kernels to fp64 rounding (not bit for bit).
 * The per-call operand Z gets the same conditioning test as the weights, as a DIAGNOSTIC: rows whose
 * mean |z| is below 2^-6 of their largest entry (possible with an unbounded activation; never with
 * tanh) are counted per device -- their products are exact to 2^-54 K max|z| max|w| rather than to
 * fp64 rounding of sum |z||w|.  l2q_heads_sliced_zflag copies the count to *count (host), optionally
 * resets it, and synchronises the stream. */
size_t l2q_heads_sliced_bytes(int K, long N);
int l2q_heads_sliced_zflag(int reset, int* count, void* stream);
int l2q_heads_sliced_build(const double* Ws, const double* Wt, const double* Wq, int K, long N, void* sliced,
                           size_t sliced_bytes, int* usable, void* stream);
size_t l2q_vnet_heads_sliced_ws_bytes(int M, long N);
int l2q_vnet_heads_vupdate_sliced_f64(const double* Z, int M, int K, long N, const void* sliced,
                                      const double* bs, const double* cs, double scale_s, const double* bt,
                                      double scale_t, const double* bq, const double* cq, double scale_q,
                                      const void* v_in, void* v, const void* force, int is_complex, double eps1,
                                      int forward1, int pair, int flip_between, double eps2, int forward2,
                                      double* logdet, double* logdet1, double* vnorm2_mid, void* ws,
                                      size_t ws_bytes, void* stream);
/* The forward pass of the TRAINING tape on the same kernel (network.py:547-551 + dynamics.py:1266-1297 under
 * autograd: trainers/pytorch/trainer.py:1316-1367): one (not paired) out-of-place v-update v_in -> v AND the
 * three heads as the update uses them -- s = cs tanh(.), t = scale_t (.), q = cq tanh(.), [M][N] fp64 each --
 * for the reverse sweep (l2q_v_update_bwd_c128, the heads' VJPs).  The slice image is rebuilt once per
 * optimiser step from that step's weights (l2q_heads_sliced_build).  Same workspace as the call above. */
int l2q_vnet_heads_vupdate_sliced_tape_f64(const double* Z, int M, int K, long N, const void* sliced,
                                           const double* bs, const double* cs, const double* bt, double scale_t,
                                           const double* bq, const double* cq, const void* v_in, void* v,
                                           const void* force, int is_complex, double eps, int forward,
                                           double* s_out, double* t_out, double* q_out, double* logdet, void* ws,
                                           size_t ws_bytes, void* stream);
/* fp32 variant on v_mfma_f32_16x16x4_f32 (U(1) networks). */
int l2q_gemm_f32(const float* A, const float* W, int M, int N, long K, const float* A2,
                 const float* W2, long K2, const float* bias, const float* bias2,
                 const float* coeff, float scale, int act, float* C, void* ws,
                 size_t ws_bytes, void* stream);
/* Backward GEMMs of the dense layers without transposed copies (the reverse sweep of the training
 * step):  C[M][N] (+)= sum_k Aop[m][k] Wop[n][k]  with  Aop[m][k] = a_trans ?
 * A[k][m] (A stored [K][M]) : A[m][k] (A stored [M][K]), Wop likewise ([K][N] resp. [N][K]).
 *   dW += dY^T X : A = dY [nb][out] (a_trans), W = X [nb][in] (w_trans), accumulate
 *   dX  = dY W   : A = dY [nb][out],           W = W [out][in] (w_trans)
 * elem_bytes 4 | 8.  Transposed operands need their row length (M resp. N) to be a multiple of
 * 16 bytes.  ws: l2q_gemm_ws_bytes(M, N, K, 0). */
int l2q_gemm_ex(const void* A, int a_trans, const void* W, int w_trans, int M, int N, long K,
                int elem_bytes, int accumulate, void* C, void* ws, size_t ws_bytes, void* stream);
/* Half-precision network layers ("fp16 nets / fp32 action", BASELINE cfg-3): what the reference
 * gets from torch.autocast around Dynamics.forward (trainers/pytorch/trainer.py:211-219,
 * 1278-1280: nn.Linear in fp16 / bf16, lattice arithmetic in fp32).  W / W2 are 16-bit
 * (half_type: L2Q_HALF_F16 | L2Q_HALF_BF16), A / A2 16-bit or fp32 (a_is_f32: rounded to 16 bit
 * as the tile is staged, no cast pass), C 16-bit or fp32 (c_is_f32); bias / bias2 / coeff fp32.
 * fp32 accumulation on v_mfma_f32_16x16x32_{f16,bf16}.  Rounding points as autocast has them:
 *   y = r16(acc + bias);  y = r16(act(y));  C = coeff ? scale * exp(coeff[n]) * y : r16(scale * y)
 * (coeff needs c_is_f32: torch promotes fp32 x fp16 tensors to fp32).  ws: split-K scratch of
 * l2q_gemm_h_ws_bytes. */
int l2q_gemm_h(int half_type, const void* A, int a_is_f32, const void* W, int M, int N, long K,
               const void* A2, const void* W2, long K2, const float* bias, const float* bias2,
               const float* coeff, float scale, int act, void* C, int c_is_f32, void* ws,
               size_t ws_bytes, void* stream);
size_t l2q_gemm_h_ws_bytes(int M, int N, long K, long K2);
/* K-splits the streaming input-layer kernel (fp32 A / A2, N <= 256, wide K: csrc/gemm_f16_skinny.hip) takes for
 * this shape under the current `gemm_h_skinny` tuning; 0: the tile kernels run it.  K as passed to l2q_gemm_h
 * (u1x: 2 xdim). */
int l2q_gemm_h_skinny_splits(int M, int N, long K, long K2, int u1x);
/* The U(1) xnet's input layer in half precision with its [cos(m x), sin(m x)] input
 * (dynamics.py:1161-1185, network.py:430-451) formed inside the GEMM's tile loader:
 *   C = r16(act(r16([cos(keep x) | sin(keep x)] . W^T + A2 . W2^T + bias + bias2))),
 * x [M][xdim] fp32 link angles, keep = mask[xdim] (complement: 1 - mask), W [N][2 xdim] 16-bit,
 * A2 [M][K2] fp32 (the momenta), W2 [N][K2] 16-bit, C [M][N] 16-bit.
 * ws: l2q_gemm_h_ws_bytes(M, N, 2 xdim, K2). */
int l2q_gemm_h_u1x(int half_type, const float* x, const float* mask, int complement, const void* W,
                   int M, int N, long xdim, const float* A2, const void* W2, long K2,
                   const float* bias, const float* bias2, int act, void* C, void* ws,
                   size_t ws_bytes, void* stream);
/* The three heads of a half-precision U(1) LeapfrogLayer and the sub-update that consumes them
 * in one kernel (network.py:547-551 + dynamics.py:1266-1297 / 1386-1477); s, t, q never reach
 * memory.  Z [M][K] 16-bit (last hidden activation), W* [N][K] 16-bit, b* fp32 [N],
 * cs / cq = nw * exp(coeff) fp32 [N] (rounding points as l2q_gemm_h):
 *   x_update == 0: a = v (in place), b = force:  l2q_v_update's arithmetic
 *   x_update != 0: a = x (in place), b = v, mask [N] / complement:  l2q_u1_x_update's
 * logdet [M] fp32 (accumulate != 0: added to).  ws: l2q_u1_heads_update_h_ws_bytes. */
int l2q_u1_heads_update_h(int half_type, const void* Z, int M, int K, long N, const void* Ws,
                          const float* bs, const float* cs, const void* Wt, const float* bt,
                          float scale_t, const void* Wq, const float* bq, const float* cq,
                          int x_update, float* a, const float* b, const float* mask,
                          int complement, float eps, int forward, int use_ncp, float* logdet,
                          int accumulate, void* ws, size_t ws_bytes, void* stream);
size_t l2q_u1_heads_update_h_ws_bytes(int M, long N);
/* Half-precision ConvStack layers (autocast runs nn.Conv2d in 16 bit too, network.py:283-326):
 * l2q_conv_gemm_periodic_f32's implicit GEMM on the 16-bit MFMA.  in: fp32 (in_is_f32, the first
 * layer's [cos, sin] lattice data, rounded while staged) or 16-bit; weight [cout][C k k] 16-bit in
 * (ci, i, j) or, channels_last_cols != 0, (i, j, ci) order; bias fp32; out NHWC 16-bit
 * = r16(act(r16(conv + bias))).  l2q_maxpool_act_nhwc_h: MaxPool2d(pool) then activation on NHWC
 * 16-bit data. */
int l2q_conv_gemm_periodic_h(int half_type, const void* in, int in_is_f32, long sn, long sc, long sh,
                             long sw, int nb, int C, int H, int W, int k, const void* weight,
                             int channels_last_cols, const float* bias, int cout, int act,
                             void* out, void* stream);
/* l2q_conv_gemm_periodic_h followed by MaxPool2d(2) (floor mode) and the activation, in one kernel
 * (network.py:283-326: Conv2d -> MaxPool2d -> act of the pooled ConvStack layers): out is the pooled
 * NHWC image [nb][Ho / 2][Wo / 2][cout] = r16(act(max_window r16(conv + bias))); the un-pooled image
 * is never written.  Same bits as l2q_conv_gemm_periodic_h (act none) + l2q_maxpool_act_nhwc_h(2). */
int l2q_conv_pool_gemm_periodic_h(int half_type, const void* in, int in_is_f32, long sn, long sc,
                                  long sh, long sw, int nb, int C, int H, int W, int k,
                                  const void* weight, int channels_last_cols, const float* bias,
                                  int cout, int act, void* out, void* stream);
int l2q_maxpool_act_nhwc_h(int half_type, const void* in, int nb, int H, int W, int C, int pool,
                           int act, void* out, void* stream);
/* out[b][h][w][c] = c < C ? r16(in[b][c][h][w]) : 0 for c < cpad: the fp32 NCHW lattice input of
 * the first conv layer as 16-bit NHWC with the channels padded to a 16-byte group. */
int l2q_nchw_to_nhwc_pad_h(int half_type, const float* in, int nb, int C, int H, int W, int cpad,
                           void* out, void* stream);

/* ---------------------------------------------------------------- U(1) lattice kernels */
/* x[nb][2][T][X] angles, elem_bytes 4 or 8.
 * out[c][0] = sum cos(theta), [1] = sum sin(theta), [2] = sum project_angle(theta)
 * (lattice/u1/pytorch/lattice.py:154-159, 80-86, 188-228):
 *   action = beta (V - out0); plaqs = out0/V; sinQ = out1/2pi; intQ = out2/2pi. */
int l2q_u1_plaq_reduce(const void* x, int nb, int T, int X, int elem_bytes, void* out,
                       void* stream);
/* The plaquette-angle field out[c][t][x] = U0(t,x) + U1(t+1,x) - U0(t,x+1) - U1(t,x): the tensor [nb, T, X]
 * that LatticeU1.wilson_loops returns (lattice/u1/pytorch/lattice.py:154-159). */
int l2q_u1_wilson_loops(const void* x, int nb, int T, int X, int elem_bytes, void* out, void* stream);
/* its adjoint (VJP): dx0(t,x) += g(t,x) - g(t,x-1), dx1(t,x) += g(t-1,x) - g(t,x);  g [nb][T][X] */
int l2q_u1_wilson_loops_bwd(const void* g, int nb, int T, int X, int elem_bytes, void* dx, void* stream);
/* F0 = beta [sin th - sin th(t,x-1)], F1 = beta [-sin th + sin th(t-1,x)]
 * (autograd of the action, lattice.py:102-117).  If v != NULL: v += coef * F (fused kick)
 * and `force` may be NULL. */
int l2q_u1_force(const void* x, double beta, void* force, void* v, double coef, int nb, int T,
                 int X, int elem_bytes, void* stream);
/* NCP position update (dynamics.py:1386-1477, use_ncp branch) followed by compat_proj:
 *   forward : x' = m x + (1-m) [2 atan(tan(x/2) e^{eps s}) + eps (v e^{eps q} + t)]
 *   backward: x' = m x + (1-m) [2 atan(tan(x/2) e^{-eps s}) - e^{-eps s} eps (v e^{eps q} + t)]
 *   logdet[c] = sum (1-m) log(e^{+-eps s} / (cos^2(x/2) + e^{+-2 eps s} sin^2(x/2)))
 * mask: float32 [n] (the kept entries; complement flips it).  x updated in place. */
int l2q_u1_x_update(void* x, const void* v, const void* s, const void* t, const void* q,
                    const float* mask, int complement, double eps, int forward, int use_ncp,
                    int elem_bytes, int nb, long n, void* logdet, void* stream);
/* ((x + pi) mod 2 pi) - pi   (group/u1/pytorch/group.py:137-138); y may alias x */
int l2q_u1_wrap(const void* x, void* y, long n, int elem_bytes, void* stream);
/* out[c] = 0.5 sum v^2 (group/u1/pytorch/group.py:164-165) */
int l2q_u1_kinetic_reduce(const void* v, int nb, long n, int elem_bytes, void* out,
                          void* stream);
/* y = x + alpha * p  (U1Phase.update_gauge, group.py:102-103) */
int l2q_axpy(const void* p, double alpha, void* x, long n, int elem_bytes, void* stream);
/* [cos(m x), sin(m x)] channels for the U(1) xnet input (group.py:86-89 applied to the
 * masked field, dynamics.py:1398,1174): out[nb][4][T*X] from x[nb][2][T*X] */
int l2q_u1_masked_cos_sin(const void* x, const float* mask, int complement, void* out, int nb,
                          long n, int elem_bytes, void* stream);
/* periodic-padded conv2d (+ optional max-pool p, + activation), NCHW fp32:
 * PeriodicPadding(k-1) -> Conv2d(k) -> [MaxPool2d(p)] -> [act]
 * (network/pytorch/network.py:151-172, 283-326). */
int l2q_conv2d_periodic_f32(const float* in, const float* w, const float* bias, float* out,
                            int nb, int cin, int H, int W, int cout, int k, int pool, int act,
                            void* stream);

/* The same conv layer as an implicit GEMM (the fast path of ConvStack): periodic im2col
 *   col[(b*Ho + ho)*Wo + wo][(ci*k + i)*k + j] = in[b, ci, (ho+i-k+1) mod H, (wo+j-k+1) mod W]
 * with Ho = H+k-1, Wo = W+k-1 and generic element strides (sn, sc, sh, sw) of `in` (NCHW for
 * the first layer, the GEMM's NHWC output afterwards); then l2q_gemm_f32(col, weight[cout][cin k k])
 * gives the NHWC activation; l2q_maxpool_act_nhwc_f32 applies MaxPool2d(pool) + activation.
 * channels_last_cols != 0: column order (i*k + j)*C + ci instead (weight permuted to
 * [cout][k][k][C]): contiguous runs over the channels of an NHWC input, for im2col and for its
 * adjoint l2q_col2im_periodic_f32 alike. */
int l2q_im2col_periodic_f32(const float* in, long sn, long sc, long sh, long sw, int nb, int C,
                            int H, int W, int k, int channels_last_cols, float* col,
                            void* stream);
int l2q_maxpool_act_nhwc_f32(const float* in, int nb, int H, int W, int C, int pool, int act,
                             float* out, void* stream);
/* The conv layer without the col matrix: im2col happens inside the A-tile loader of the f32
 * MFMA GEMM.  out[(b*Ho + ho)*Wo + wo][cout] = act(conv + bias) (NHWC); weight [cout][C k k] in
 * nn.Conv2d's flatten order (ci, i, j), or -- channels_last_cols != 0 -- permuted to (i, j, ci) so
 * that consecutive K columns of an NHWC input are contiguous in memory; input strides as for
 * l2q_im2col_periodic_f32.  (The eval path uses
 * this; the training path keeps the materialised col matrix because the weight gradient needs it.) */
int l2q_conv_gemm_periodic_f32(const float* in, long sn, long sc, long sh, long sw, int nb, int C,
                               int H, int W, int k, const float* weight, int channels_last_cols,
                               const float* bias, int cout, int act, float* out, void* stream);
/* out[b][h][w][c] = c < C ? in[b][c][h][w] : 0 for c < cpad (fp32 NCHW -> NHWC, channels padded
 * to a 16-byte group): lets the first conv layer use the vector gathers of the later ones. */
int l2q_nchw_to_nhwc_pad_f32(const float* in, int nb, int C, int H, int W, int cpad, float* out,
                             void* stream);

/* ---------------------------------------------------------------- fused U(1) sub-updates (fp32)
 * One launch per L2HMC sub-update on small 2D lattices (n = 2 T X <= l2q_u1_fused_max_n()):
 * the force (v-step) or the masked cos/sin inputs (x-step), the whole dense LeapfrogLayer in
 * eval mode (network.py:430-451, 522-551; BatchNorm folded into the heads by the caller) and
 * the update with its log-det (dynamics.py:1266-1297 / 1386-1477).  s, t, q never reach memory.
 * Network description (device pointers unless noted):
 *   wxT [Kx][U0], wvT [Kv][U0]   transposed xlayer / vlayer weights, b0 [U0] = sum of their biases
 *   hidden                       for l = 1..nl-1: W_l [U_l][U_{l-1}] then b_l [U_l], packed
 *   units (HOST pointer) [nl]    widths U_0..U_{nl-1} (each <= 64, nl <= 8)
 *   ws, wt, wq [n][U_last], bs, bt, bq [n]; cs, cq [n] per-entry scale (nw * exp(coeff));
 *   scale_t; act: L2Q_ACT_* of the dense layers.
 * v-step: Kx = Kv = n (inputs x and force(x)); v updated in place, x untouched.
 * x-step: Kx = 2 n ([cos(m x), sin(m x)]), Kv = n (v); x updated in place (+ compat_proj).
 * accumulate != 0: logdet[c] += the sub-update's log-det (running sum of a leapfrog step). */
int l2q_u1_fused_max_n(void);
int l2q_u1_vstep_f32(const float* x, float* v, double beta, double eps, int forward, int nb, int T,
                     int X, const float* wxT, const float* wvT, const float* b0,
                     const float* hidden, const int* units, int nl, const float* ws,
                     const float* bs, const float* cs, const float* wt, const float* bt,
                     double scale_t, const float* wq, const float* bq, const float* cq, int act,
                     int accumulate, float* logdet, void* stream);
int l2q_u1_xstep_f32(float* x, const float* v, const float* mask, int complement, double eps,
                     int forward, int use_ncp, int nb, int n, const float* wxT, const float* wvT,
                     const float* b0, const float* hidden, const int* units, int nl,
                     const float* ws, const float* bs, const float* cs, const float* wt,
                     const float* bt, double scale_t, const float* wq, const float* bq,
                     const float* cq, int act, int accumulate, float* logdet, void* stream);

/* ================================================================ training-gradient path
 * Reverse-mode (VJP) counterparts of the U(1) sub-updates and network layers, the train-mode
 * BatchNorm1d, and the optimiser step.  The reference gets these from torch.autograd
 * (`loss.backward()`, trainers/pytorch/trainer.py:1284-1314) and torch.optim.Adam (:206-209).
 * Conventions: "g*" / "d*" arguments are cotangents with the shape of the quantity they
 * belong to; arguments documented "+=" are accumulated into, all others are overwritten. */

/* y = act(x) element-wise (ACTIVATION_FNS, network.py:40-46); y may alias x.  The training tape
 * uses it for swish, whose derivative needs the pre-activation (the GEMM epilogues fuse the
 * activation and keep only its output). */
int l2q_act_fwd(const void* x, int act, long n, int elem_bytes, void* y, void* stream);
/* dx = dy * act'(z) from the activation OUTPUT y = act(z) (tanh, relu, leaky_relu, elu, none);
 * for swish (not invertible) `y` must hold the PRE-activation z.  dx may alias dy.
 * (network.py:447-451, 536-538) */
int l2q_act_bwd(const void* dy, const void* y, int act, long n, int elem_bytes, void* dx,
                void* stream);
/* out = alpha * a * b element-wise (nn.Dropout mask, network.py:540-541); out may alias a */
int l2q_mul(const void* a, const void* b, double alpha, long n, int elem_bytes, void* out,
            void* stream);
/* y[c][:] += a[c] * x[c][:]  (cotangent of the kinetic energy 1/2 sum v^2, dynamics.py:1485) */
int l2q_axpy_rows(const void* x, const void* a, int nb, long n, int elem_bytes, void* y,
                  void* stream);
/* out[n] (+)= alpha * sum_m a[m][n] * (b ? b[m][n] : 1), fixed summation order
 * (bias gradients; ScaledTanh.coeff gradient with b = the head's output) */
int l2q_colsum(const void* a, const void* b, long M, int N, double alpha, int accumulate,
               int elem_bytes, void* out, void* ws, size_t ws_bytes, void* stream);
size_t l2q_colsum_ws_bytes(long M, int N);
/* head s = scale * exp(coeff[n]) * tanh(pre) (ScaledTanh, network.py:175-206):
 * dpre = ds * scale e^coeff (1 - tanh^2), tanh recovered from s.  coeff == NULL: linear head
 * t = scale * pre, dpre = scale * ds. */
int l2q_scaled_tanh_bwd(const void* ds, const void* s, const void* coeff, double scale, int M,
                        int N, int elem_bytes, void* dpre, void* stream);
/* l2q_scaled_tanh_bwd with the column sums of the head's parameter gradients formed in the same pass:
 * bgrad[n] += sum_m dpre[m][n] (nn.Linear bias) and, with coeff, cgrad[n] += sum_m ds[m][n] s[m][n]
 * (ScaledTanh.coeff) -- the bits of l2q_scaled_tanh_bwd + two l2q_colsum passes (same row partition and
 * summation order).  ws_bytes >= 2 * l2q_colsum_ws_bytes(M, N). */
int l2q_scaled_tanh_bwd_sums(const void* ds, const void* s, const void* coeff, double scale, int M, int N,
                             int elem_bytes, void* dpre, void* bgrad, void* cgrad, void* ws, size_t ws_bytes,
                             void* stream);
/* nn.BatchNorm1d in train mode over x[M][N] (network.py:543-544): batch mean / biased
 * variance, y = (x - mean) invstd gamma + beta; running stats (may be NULL) updated with
 * `momentum` and the unbiased variance like torch. */
int l2q_bn_train_fwd(const void* x, const void* gamma, const void* beta, double eps,
                     double momentum, void* running_mean, void* running_var, int M, int N,
                     int elem_bytes, void* y, void* save_mean, void* save_invstd, void* stream);
/* dx overwritten; dgamma, dbeta += */
int l2q_bn_bwd(const void* dy, const void* x, const void* save_mean, const void* save_invstd,
               const void* gamma, int M, int N, int elem_bytes, void* dx, void* dgamma,
               void* dbeta, void* stream);
/* adjoint of l2q_im2col_periodic_f32 (gather form, no atomics): dx (strides sn, sc, sh, sw)
 * overwritten with the sum of the dcol entries that read each input pixel */
int l2q_col2im_periodic_f32(const float* dcol, long sn, long sc, long sh, long sw, int nb, int C,
                            int H, int W, int k, int channels_last_cols, float* dx,
                            void* stream);
/* adjoint of l2q_maxpool_act_nhwc_f32: din[nb][H][W][C] overwritten (first maximum of each
 * window receives dout * act'(out)) */
int l2q_maxpool_act_nhwc_bwd_f32(const float* dout, const float* out, const float* in, int nb,
                                 int H, int W, int C, int pool, int act, float* din,
                                 void* stream);
/* VJP of the U(1) force (the reference differentiates through autograd.grad(create_graph=True),
 * lattice/u1/pytorch/lattice.py:102-117):  dx += D^T [cos(theta) beta (D dF)] */
int l2q_u1_force_bwd(const void* x, const void* dF, double beta, int nb, int T, int X,
                     int elem_bytes, void* dx, void* stream);
/* VJP of the per-chain plaquette sums (action, sin-charge; lattice.py:80-86, 221-224):
 * dx += D^T [-gcos[c] sin(theta) + gsin[c] cos(theta)];  gcos / gsin [nb], either may be NULL */
int l2q_u1_plaq_bwd(const void* x, const void* gcos, const void* gsin, int nb, int T, int X,
                    int elem_bytes, void* dx, void* stream);
/* VJP of l2q_u1_x_update.  x: the field BEFORE the update; gx [nb][n]: cotangent of the
 * updated field; gl [nb] (may be NULL): cotangent of logdet.  dx, ds, dt, dq overwritten,
 * dv +=, deps[c] = per-chain d/d eps (sum over chains on the host side). */
int l2q_u1_x_update_bwd(const void* x, const void* v, const void* s, const void* t, const void* q,
                        const float* mask, int complement, double eps, int forward, int use_ncp,
                        const void* gx, const void* gl, int elem_bytes, int nb, long n, void* dx,
                        void* dv, void* ds, void* dt, void* dq, void* deps, void* stream);
/* VJP of l2q_v_update (real fields).  v: momentum BEFORE the update.  All outputs overwritten. */
int l2q_v_update_bwd(const void* v, const void* force, const void* s, const void* t,
                     const void* q, double eps, int forward, const void* gv, const void* gl,
                     int elem_bytes, int nb, long n, void* dv, void* dF, void* ds, void* dt,
                     void* dq, void* deps, void* stream);
/* VJP of l2q_u1_masked_cos_sin: dx += m (-sin(m x) dout[:, 0:2] + cos(m x) dout[:, 2:4]) */
int l2q_u1_masked_cos_sin_bwd(const void* x, const float* mask, int complement, const void* dout,
                              int nb, long n, int elem_bytes, void* dx, void* stream);
/* torch.optim.Adam step (no weight decay / amsgrad) over one flat parameter arena; the
 * gradient is multiplied by grad_scale first (1/world_size, gradient clipping). step >= 1. */
int l2q_adam(void* p, const void* g, void* m, void* v, long n, double lr, double beta1,
             double beta2, double eps, long step, double grad_scale, int elem_bytes,
             void* stream);
/* out[0] = sum a^2 (gradient norm for clip_grad_norm, trainer.py:1297-1301), order-stable */
int l2q_sumsq(const void* a, long n, int elem_bytes, double* out, void* ws, size_t ws_bytes,
              void* stream);
size_t l2q_sumsq_ws_bytes(long n);

/* ---- SU(3) cotangents (fp64 / complex128, native layout xn[nb][4][9][V]).  Cotangent
 * convention of torch for complex tensors: g_z = dL/dRe z + i dL/dIm z. */

/* VJP of l2q_su3_expm_mul (one masked half-update, dynamics.py:1420-1425 / 1468-1474; eps is
 * the signed step the forward was called with).  gx overwritten, gv +=, deps[c] = per-chain
 * dL/d eps.  Uses the Frechet derivative of the matrix exponential (scaling & squaring). */
int l2q_su3_expm_mul_bwd(const void* xn, const void* vn, double eps, const float* mask_n,
                         int complement, const void* gxnew, void* gx, void* gv, double* deps,
                         int nb, long V, void* ws, size_t ws_bytes, void* stream);
/* VJP of l2q_su3_expm_mul2 (BOTH masked half-updates of a leapfrog step, the mask first or -- complement_first --
 * its complement first; dynamics.py:1420-1425 / 1468-1474 under autograd): the derivative of the matrix
 * exponential is linear in its direction, so the two halves share one exponential and ONE Frechet
 * derivative.  xn: the links BEFORE both halves; gxnew: cotangent of the links after both; same outputs as
 * l2q_su3_expm_mul_bwd (deps = the sum over both halves). */
int l2q_su3_expm_mul2_bwd(const void* xn, const void* vn, double eps, const float* mask_n,
                          int complement_first, const void* gxnew, void* gx, void* gv, double* deps,
                          int nb, long V, void* ws, size_t ws_bytes, void* stream);
/* VJP of l2q_su3_projsu_vec8 (su3_to_vec(projectSU(.)), group.py:138-147): gm += .  `in` are
 * the matrices the forward projected; gvec [nfields][8][V]. */
int l2q_su3_projsu_vec8_bwd(const void* in, const double* gvec, void* gm, long nfields, long V,
                            void* stream);
/* VJP of the force as the reference differentiates it: F = TAH(D x^H) with D = dS/dx held
 * constant (lattice/su3/pytorch/lattice.py:299-308, no create_graph):
 * gx += (beta/3) TAH(gf) (staple sum)^H */
int l2q_su3_force_bwd(const void* xn, const void* gf, double beta, void* gx, int nb, int T, int X,
                      int Y, int Z, void* stream);
/* VJP of the per-plane plaquette sums (action, plaq / charge loss terms, loss.py:56-148):
 * L = sum_planes Re(conj(w[c][p]) sum_sites tr P_p), w [nb][6] complex (re, im), planes ordered
 * as l2q_su3_plaq_planes.  gx += dL/dx. */
int l2q_su3_plaq_bwd(const void* xn, const double* w, void* gx, int nb, int T, int X, int Y, int Z,
                     void* stream);
/* VJP of l2q_su3_wilson_loops: L = sum_{p, c, site} Re(conj(w[p][c][site]) tr P_p(site)), w complex128 laid
 * out like that function's output (torch's cotangent of a complex tensor: dL/dRe + i dL/dIm).  gx += dL/dx. */
int l2q_su3_wilson_loops_bwd(const void* xn, const void* w, void* gx, int nb, int T, int X, int Y, int Z,
                             void* stream);
/* ---- improved gauge actions (c1 != 0: Iwasaki / DBW2 rectangles), lattice/su3/pytorch/lattice.py
 * :83-112 (coeffs, _rectangles), :180-196 (rectangle traces of _wilson_loops), :252-269 (action):
 *   S = -(1/3) [ beta (1 - 8 c1) sum_P Re tr P + beta c1 sum_R Re tr R ],  12 planar 2x1 loops R
 * per site.  out[c] = sum_R Re tr R.  ws >= nb * ceil(V / 256) doubles. */
int l2q_su3_rect_reduce(const void* xn, int nb, int T, int X, int Y, int Z, double* out, void* ws,
                        size_t ws_bytes, void* stream);
/* fn += coef * TAH(U (sum of the 18 rectangle staples of the link)): with coef = beta c1 / 3 the
 * rectangle part of grad_action = projectTAH(dS/dx x^H) (lattice.py:299-308) for c1 != 0. */
int l2q_su3_rect_force_add(const void* xn, double coef, void* fn, int nb, int T, int X, int Y, int Z,
                           void* stream);
/* gx += w[c] * d(sum_R Re tr R)/dx  (cotangent of the rectangle term of the action in the
 * Metropolis test of a training step). */
int l2q_su3_rect_bwd(const void* xn, const double* w, void* gx, int nb, int T, int X, int Y, int Z,
                     void* stream);
/* l2q_v_update_bwd for complex128 momenta (SU3): v, force, gv, dv, dF complex [nb][n];
 * s, t, q, ds, dt, dq real [nb][n].  All outputs overwritten. */
int l2q_v_update_bwd_c128(const void* v, const void* force, const double* s, const double* t,
                          const double* q, double eps, int forward, const void* gv,
                          const double* gl, int nb, long n, void* dv, void* dF, double* ds,
                          double* dt, double* dq, double* deps, void* ws, size_t ws_bytes,
                          void* stream);
/* The same VJP with (dF, ds, dt, dq) = this update's cotangents + (acc_dF, acc_ds, acc_dt, acc_dq): the
 * two v-updates either side of a momentum flip or a step boundary act on the same x, hence on ONE force and
 * ONE network evaluation (dynamics.py:1187-1228 with a shared vnet), and the reverse sweep hands the
 * later one's cotangents to the earlier one before the single backward pass through network and force.
 * acc_* may alias nothing that is written. */
int l2q_v_update_bwd_acc_c128(const void* v, const void* force, const double* s, const double* t,
                              const double* q, double eps, int forward, const void* gv, const double* gl,
                              int nb, long n, const void* acc_dF, const double* acc_ds, const double* acc_dt,
                              const double* acc_dq, void* dv, void* dF, double* ds, double* dt, double* dq,
                              double* deps, void* ws, size_t ws_bytes, void* stream);
/* Both v-updates of such a pair reversed in one pass: v_mid = U1(v1) [, v_mid <- -v_mid], v_out = U2(v_mid) with
 * ONE force and ONE (s, t, q); gv = cotangent of v_out.  dv = cotangent of v1; (dF, ds, dt, dq) = the sum of both
 * updates' cotangents; deps1 / deps2 = the per-chain d/d eps of the first / second update.  F, s, t, q are read
 * once (144 instead of 296 bytes per entry for the two calls above).  ws_bytes >= 2 x the single call's. */
int l2q_v_update_bwd_pair_c128(const void* v1, const void* v_mid, const void* force, const double* s,
                               const double* t, const double* q, double eps1, int forward1, double eps2,
                               int forward2, int flip_between, const void* gv, const double* gl, int nb, long n,
                               void* dv, void* dF, double* ds, double* dt, double* dq, double* deps1,
                               double* deps2, void* ws, size_t ws_bytes, void* stream);
/* gx[c][:] += 2 a[c] (x[c][:] - y[c][:]) over n doubles per chain (cotangent of
 * l2q_diff_norm2_reduce, the rmse term of LatticeLoss, loss.py:119-148) */
int l2q_diff_bwd_f64(const double* x, const double* y, const double* a, int nb, long n, double* gx,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* L2Q_H_ */
