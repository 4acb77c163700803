/*
 * ttsk.h -- C ABI of libttsk.so: the MI355X (gfx950) implementation of the
 * streaming TT-sketch hot path of RikVoorhaar/tt-sketch.
 *
 * The reference has exactly one native module (tt_sketch/drm/fast_lazy_gaussian.pyx);
 * everything else on the path is Python plug-in surface (DRM.sketch_* generators and
 * the sketch_omega_* / sketch_psi_* tables).  Each entry point below names the
 * reference interface it replaces (file:line relative to the reference root).
 *
 * Conventions
 *  - every function returns 0 on success or a negative ttsk_status; the message of
 *    the last failure on the calling thread is ttsk_last_error().  No exceptions
 *    cross the boundary.
 *  - "dev" pointers are device addresses obtained from ttsk_malloc; "host" pointers
 *    are ordinary process memory.  The library never keeps or frees host pointers.
 *  - all floating point data is fp64; index data is int64 (as the reference stores
 *    it) unless stated; hashes are uint64.
 *  - calls are asynchronous on the library's stream `stream` (0..TTSK_NUM_STREAMS-1)
 *    unless they take host pointers, in which case they block until the data is valid.
 */
#ifndef TTSK_H
#define TTSK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    TTSK_OK = 0,
    TTSK_ERR_HIP = -1,      /* a HIP runtime call failed (no device, OOM, launch failure) */
    TTSK_ERR_ARG = -2,      /* invalid argument (shape / stride / NULL) -> Python ValueError */
    TTSK_ERR_UNSUPPORTED = -3,
    TTSK_ERR_COMM = -4      /* RCCL failure */
} ttsk_status;

#define TTSK_NUM_STREAMS 8

/* ---- runtime ------------------------------------------------------------ */
int ttsk_init(int device);                    /* selects the device, creates streams; idempotent */
int ttsk_shutdown(void);
const char *ttsk_last_error(void);
int ttsk_device_info(char *name, size_t name_len, int *num_cu, size_t *hbm_bytes);
int ttsk_malloc(void **dev, size_t bytes);
int ttsk_free(void *dev);
int ttsk_memset(void *dev, int value, size_t bytes, int stream);
int ttsk_h2d(void *dev, const void *host, size_t bytes, int stream);   /* blocking */
int ttsk_d2h(void *host, const void *dev, size_t bytes, int stream);   /* blocking */
int ttsk_d2d(void *dst, const void *src, size_t bytes, int stream);
int ttsk_sync(int stream);                    /* stream < 0: all streams */
int ttsk_stream_wait(int waiter, int signaller); /* waiter waits for work queued so far on signaller */
/* hipEvent timing on a library stream (bench.py roofline leg) */
int ttsk_timer_start(int stream);
int ttsk_timer_stop(int stream, float *ms);   /* blocks until the stop event completes */
/* hipGraph capture of a launch sequence on one stream (launch-bound inner loops) */
int ttsk_graph_begin(int stream);
int ttsk_graph_end(int stream, void **graph_exec);
int ttsk_graph_launch(void *graph_exec, int stream);
int ttsk_graph_free(void *graph_exec);

/* ---- generic fp64 contraction (MFMA 16x16x4) -----------------------------
 * C[b,m,n] (+)= alpha * sum_{ko,ki} A[b,m,ko,ki] * B[b,ko,ki,n]
 * with arbitrary element strides.  This is the device counterpart of the
 * np.einsum / tensordot / `@` calls in tt_sketch/drm/tensor_train_drm.py:79-141,
 * tt_sketch/drm/dense_gaussian_drm.py:72-75, tt_sketch/tensor.py:390-397 and
 * tt_sketch/sketching_methods/{tensor_train,cp,dense,tucker}_sketch.py.
 * Optional row scaling of A by a vector (sparse_sketch.py:46, `left * entries`). */
typedef struct {
    int64_t batch, M, N, Ko, Ki;
    int64_t a_b, a_m, a_ko, a_ki;    /* element strides of A */
    int64_t b_b, b_ko, b_ki, b_n;    /* element strides of B */
    int64_t c_b, c_m, c_n;           /* element strides of C */
    double alpha;
    int accumulate;                  /* 0: C = ..., 1: C += ... */
    int split_k;                     /* 0: library chooses; >=1 explicit */
} ttsk_gemm_desc;
int ttsk_gemm(const ttsk_gemm_desc *desc, const double *A, const double *B, double *C,
              const double *k_scale /* NULL or length Ko*Ki, multiplies A along k */,
              int stream);
/* strided copy / permute (<= 5 dims): dst[i0..i4] = src[i0..i4]; Tensor.T materialisation */
int ttsk_copy_strided(double *dst, const double *src, int ndim, const int64_t *shape,
                      const int64_t *dst_strides, const int64_t *src_strides, int stream);
/* y[i] = a*x[i] + b*y[i] on contiguous buffers: SketchContainer.__add__
 * (sketch_container.py:61-69) and the TensorSum accumulators (sketch_dispatch.py:93-136) */
int ttsk_axpby(double *y, const double *x, double a, double b, size_t n, int stream);
/* dst[i] (+)= sum_{b<nb} src[b*stride + i], i < n: the partial sketches of a batch summed into one
 * (the `+=` loop of sum_sketch, sketch_dispatch.py:141-147); 16-byte loads when n and stride are even and the
 * bases 16-byte aligned */
int ttsk_sum_slices(double *dst, const double *src, int nb, size_t stride, size_t n, int accumulate, int stream);

/* ---- TT input x TT DRMs: the whole streaming sketch in one call ---------------
 * general_sketch(TensorTrain, TensorTrainDRM, TensorTrainDRM, streaming)
 * (sketch_dispatch.py:202-275) = right chain + left chain of TensorTrainDRM.sketch_tt
 * (tensor_train_drm.py:71-88, through handle_transpose drm_base.py:122-145 for the right
 * side, without materialising tensor.T), Omega_mu = L_mu^T R_mu and
 * Psi_mu = L_{mu-1}^T X_mu R_mu (tensor_train_sketch.py:8-35).  The first GEMM of a left
 * chain step, T = L_{mu-1}^T X_mu, is shared with Psi_mu.
 *
 *   n[d]              mode sizes
 *   s[d+1]            TT ranks of the input, s[0] = s[d] = 1; X[mu] is (s[mu], n[mu], s[mu+1])
 *   lt[d]             true ranks of the left DRM, lt[0] = 1; DL[mu] is (lt[mu], n[mu], lt[mu+1]), mu < d-1
 *   rt[d]             true ranks of the right DRM in ITS walking order (mode d-1 first), rt[0] = 1;
 *                     DR[j] is (rt[j], n[d-1-j], rt[j+1]), j < d-1
 *   l_lo/l_hi[d-1]    rank_min / rank_max of the left DRM (slice of the yielded contraction, :88)
 *   r_lo/r_hi[d-1]    same for the right DRM, in its walking order
 *   out               packed [Psi_0 .. Psi_{d-1}, Omega_0 .. Omega_{d-2}], Psi_mu (l_{mu-1}, n_mu, r_mu),
 *                     Omega_mu (l_mu, r_mu) with l, r the sliced ranks in user order -- the buffer that
 *                     ttsk_comm_allreduce_sum reduces.  accumulate != 0 adds (TensorSum / `stt + X`,
 *                     sketch_dispatch.py:85-139, sketch.py:292-301).
 * All pointer arrays are host arrays of device pointers. */
int ttsk_tt_sketch(int d, const int64_t *n, const int64_t *s, const int64_t *lt, const int64_t *l_lo,
                   const int64_t *l_hi, const int64_t *rt, const int64_t *r_lo, const int64_t *r_hi,
                   const double *const *X, const double *const *DL, const double *const *DR,
                   double *out, int accumulate, int stream);
/* The same for nb tensors of ONE signature (n, s) against the same DRMs: X[b*d + mu] is core mu of
 * tensor b, sketch b goes to out + b*out_stride (out_stride >= ttsk_tt_sketch_size, ignored for
 * nb == 1).  Every chain product is then a single launch over all nb tensors -- the kernels'
 * fixed costs (launch gap, staging the small operand, pipeline fill and drain) are paid once per
 * batch.  This is the term loop of a TensorSum input (sketch_dispatch.py:85-139) and the
 * throughput mode bench.py measures. */
int ttsk_tt_sketch_batch(int nb, int d, const int64_t *n, const int64_t *s, const int64_t *lt, const int64_t *l_lo,
                         const int64_t *l_hi, const int64_t *rt, const int64_t *r_lo, const int64_t *r_hi,
                         const double *const *X, const double *const *DL, const double *const *DR,
                         double *out, int64_t out_stride, int accumulate, int stream);
/* The sketch of the SUM of nb tensors of one signature (a TensorSum of TTs, sketch_dispatch.py:85-139) as ONE
 * packed sketch at `out`: the chains run per tensor as in ttsk_tt_sketch_batch, Psi_mu and Omega_mu contract over
 * (tensor, TT rank) in one product each -- a sum of TTs is a TT with block-diagonal cores -- so no per-tensor
 * sketch is written or summed.  accumulate != 0 adds to `out`. */
int ttsk_tt_sketch_sum(int nb, int d, const int64_t *n, const int64_t *s, const int64_t *lt, const int64_t *l_lo,
                       const int64_t *l_hi, const int64_t *rt, const int64_t *r_lo, const int64_t *r_hi,
                       const double *const *X, const double *const *DL, const double *const *DR,
                       double *out, int accumulate, int stream);
/* One interior step of a TensorTrainDRM chain for nb tensors of one signature, both products in one launch
 * with the intermediate kept on chip (csrc/chain_fused.h):
 *   Out_b[j, a'] = sum_{c, a, k} W_b[c, a] X_b(j, k, c) E[a, k, a']        (tensor_train_drm.py:81-87)
 * W_b (K1 x A, row stride w_c), X_b addressed by element strides (x_j, x_k, x_c) -- so the transposed core of
 * a right sketch needs no copy --, E (A, n, A2) contiguous, Out_b (J x A2) contiguous.  T != NULL also stores
 * T_b[a, k, j] = sum_c W_b[c, a] X_b(j, k, c) (A x n x J contiguous), the operand Psi_mu shares with the left
 * chain (tensor_train_sketch.py:28-34).  TTSK_ERR_UNSUPPORTED when the shape is outside the kernel's cover
 * (ranks whose 16-tile / 4-strip structure has no instantiation, J > 112, odd A2); ttsk_tt_sketch_batch then
 * uses the two-launch form. */
int ttsk_chain_step(int nb, int n, int K1, int A, int A2, int J, const double *const *W, int64_t w_c,
                    const double *const *X, int64_t x_j, int64_t x_k, int64_t x_c, int64_t x_extent,
                    const double *E, double *const *T, double *const *Out, int stream);
/* The same step (same arguments, same results) on the chunked kernel (csrc/chain_wide.h): the DRM rank A is cut
 * into chunks dealt over workgroups, a wave carries up to two 16-row tiles.  Covers what ttsk_chain_step does not:
 * TT ranks K1 beyond 128 and J up to 176 (rank-150 inputs, scripts/plot_timings.py:28-36), input and output DRM
 * ranks of different tile structure, odd DRM ranks, A2 up to 160, K1 much smaller than A.  TTSK_ERR_UNSUPPORTED
 * outside that; ttsk_tt_sketch_batch tries ttsk_chain_step's kernel, then this one, then the two-launch form. */
int ttsk_chain_step_wide(int nb, int n, int K1, int A, int A2, int J, const double *const *W, int64_t w_c,
                         const double *const *X, int64_t x_j, int64_t x_k, int64_t x_c, int64_t x_extent,
                         const double *E, double *const *T, double *const *Out, int stream);
/* The same step for MANY low-rank tensor trains at once (csrc/chain_sum.h: the summands of a TensorSum,
 * sketch_dispatch.py:85-147, whose chains share the DRM; K1, J <= 20, A, A2 <= 128, A2 even): the rows of several
 * terms are stacked into full 16-row tiles, the work of the second product is dealt over the waves by (row tile,
 * column tile) rectangles.  T (optional) receives the intermediate as T[b * t_b + (a * n + k) * t_ld + j] (t_extent
 * elements addressable): t_b = J, t_ld = nb * J is the operand layout of the Psi of a sum (one product over (term,
 * rank)).  TTSK_ERR_UNSUPPORTED outside the cover; ttsk_tt_sketch_sum / _batch then use the other forms. */
int ttsk_chain_step_sum(int nb, int n, int K1, int A, int A2, int J, const double *const *W, int64_t w_c,
                        const double *const *X, int64_t x_j, int64_t x_k, int64_t x_c, int64_t x_extent,
                        const double *E, double *T, int64_t t_b, int64_t t_ld, int64_t t_extent,
                        double *const *Out, int stream);
/* Dense tensor x DRM MATRICES (dense_gaussian_drm.py:77-80, dense_sketch.py:7-52): the first four left products
 * Z_mu = A_mu X^{<mu+1>}, mu = 0..3, from ONE read of X viewed as (n0, n1, n2, C = n3 * n4) (csrc/dense_left_pass.hip).
 * A0 (l, n0) and A3 (l, n0 n1 n2 n3) as the DRM holds them, A1t (l, n1, n0) / A2t (l, n2, n1, n0) the transposed copies of
 * A_1 / A_2; Z0 (l, n1 n2 C), Z1 (l, n2 C), Z2 (l, C); E3 (l, C) still holds i3: Z_3[a, i4] = sum_{i3} E3[a, (i3, i4)].
 * TTSK_ERR_UNSUPPORTED outside the cover (n0 % 4, C % 512, n4 in {64, 128, 256, 512}, l <= 20). */
int ttsk_dense_left_pass(const double *X, int64_t n0, int64_t n1, int64_t n2, int64_t C, int64_t n4, int l,
                         const double *A0, const double *A1t, const double *A2t, const double *A3, double *Z0, double *Z1,
                         double *Z2, double *E3, int stream);
/* number of doubles ttsk_tt_sketch writes to `out` */
int64_t ttsk_tt_sketch_size(int d, const int64_t *n, const int64_t *l_lo, const int64_t *l_hi,
                            const int64_t *r_lo, const int64_t *r_hi);
/* per-kernel device timing of the launches made by ttsk_tt_sketch (bench.py roofline leg):
 * while enabled every GEMM launch is bracketed by hipEvents on its stream. */
int ttsk_prof_enable(int on);
/* measured ceiling of v_mfma_f64_16x16x4_f64 on this device (register-resident operands, 4 independent
 * accumulators per wave, two waves per SIMD, every CU busy): TFLOP/s.  MI355X_MICROARCH.md lists no fp64 MFMA
 * row, so bench.py states this number next to the 78.6 TF/s data-sheet value (it measures 77.7 at 2.4 GHz:
 * one instruction per 64 cycles and SIMD; sustained kernels run at a lower clock). */
int ttsk_mfma_f64_peak_probe(double *tflops);
/* measured ceiling of the hash-Gaussian sampler's arithmetic on this device (fast_lazy_gaussian.pyx:23-49,91-105: the 64-bit
 * mix, the forced-exponent bits, Cephes ndtri), operands resident, no memory traffic: gsamples[0] with the central / tail
 * split of the sampling kernels (tail samples queued and evaluated a full wave at a time), gsamples[1] every lane through
 * ndtri as it comes.  10^9 samples / s.  bench.py prices the sampling passes of the sparse sketch against [0]. */
int ttsk_ndtri_rate_probe(double *gsamples);
/* class 0/1: right-chain GEMM1 (T = R^T X^T) / GEMM2 (split-K); 2/3: left-chain GEMM1 / GEMM2;
 * 4: Psi GEMM; 5: small products (Omega, first mode); 7: untagged ttsk_gemm calls.  Only the main
 * contraction kernel of each call is bracketed (not the split-K reduce / zero fill) -- except for the fused
 * chain step, which replaces BOTH products of a step: it is filed under class 1 (right) / 3 (left) with the
 * flops of both and its slab reduce inside the bracket. */
int ttsk_prof_read(int cls, int64_t *launches, double *total_ms, double *flops);
/* rocprofv3 name of the contraction-kernel instantiation last launched for class `cls` */
int ttsk_prof_kernel_name(int cls, char *buf, size_t len);

/* ---- hash sampler (the reference's native module) -------------------------
 * Host-pointer twins of the Cython API (fast_lazy_gaussian.pyx:14,53,156,183);
 * idx is (m, N) row-major int64/uint64, shape has m entries. */
int ttsk_hash_u64(uint64_t *host_vals, size_t n);                       /* pyx:13-37, in place */
int ttsk_inds_to_rand_double(const uint64_t *host_idx, const uint64_t *shape, int m, size_t N,
                             int rank_min, int rank_max, uint64_t seed, double *host_out);
int ttsk_inds_to_normal(const int64_t *host_idx, const uint64_t *shape, int m, size_t N,
                        int rank_min, int rank_max, uint64_t seed, double *host_out /* (N,rank) */);
int ttsk_inds_to_sparse_sign(const int64_t *host_idx, const uint64_t *shape, int m, size_t N,
                             int true_rank, int rank_min, int rank_max, int nnz_per_row,
                             uint64_t seed, int16_t *host_out /* (N, rank_max-rank_min) */);
/* Device-resident forms used by SparseGaussianDRM / SparseSignDRM.sketch_sparse
 * (sparse_gaussian_drm.py:29-44, sparse_sign_drm.py:34-51).  idx rows are addressed
 * as dev_idx + row_order[i]*row_stride so that tensor.T (tensor.py:201-204, reversed
 * index rows) needs no copy.  Output (N, rank) row-major fp64. */
int ttsk_sparse_normal_dev(const int64_t *dev_idx, int64_t row_stride, const int *row_order,
                           const uint64_t *shape, int m, size_t N, int rank_min, int rank_max,
                           uint64_t seed, double *dev_out, int stream);
/* the same samples for EVERY possible prefix: row f of dev_out (prod(shape), rank) belongs to the index row whose
 * Fortran-order flat index is f (only without the 32-bit wrap of the reference's running product) */
int ttsk_sparse_normal_table(const uint64_t *shape, int m, int rank_min, int rank_max, uint64_t seed, double *dev_out,
                             int stream);
int ttsk_sparse_sign_dev(const int64_t *dev_idx, int64_t row_stride, const int *row_order,
                         const uint64_t *shape, int m, size_t N, int true_rank, int rank_min,
                         int rank_max, int nnz_per_row, uint64_t seed, double *dev_out, int stream);
/* sign rows for EVERY possible prefix, as ttsk_sparse_normal_table (sparse_sign_drm.py:34-51 on all index rows at once) */
int ttsk_sparse_sign_table(const uint64_t *shape, int m, int true_rank, int rank_min, int rank_max, int nnz_per_row, uint64_t seed,
                           double *dev_out, int stream);
/* counter-based N(0,1) fill for TensorTrainDRM / DenseGaussianDRM sampling
 * (tensor.py:358-371 via utils.py:178-227; dense_gaussian_drm.py:50-55): the
 * reference's streams are not reproducible across hosts, parity is by
 * injection; this generator is the same hash -> ndtri construction keyed by (seed, i). */
int ttsk_fill_normal(double *dev_out, size_t n, uint64_t seed, double scale, int stream);
/* `count` such fills in one launch -- array i gets exactly the samples ttsk_fill_normal(seeds[i], scales[i]) would
 * give it (the d - 1 cores of a TensorTrainDRM, tensor.py:358-371: one launch instead of d - 1).  Host arrays. */
int ttsk_fill_normal_many(int count, double *const *dev_outs, const size_t *ns, const uint64_t *seeds,
                          const double *scales, int stream);

/* ---- sparse-input kernels --------------------------------------------------
 * TensorTrainDRM.sketch_sparse (tensor_train_drm.py:60-69): per nonzero e,
 *   v_e <- v_e * D[:, idx[e], :], out (N, rho') row-major; vin NULL for the first mode */
int ttsk_sparse_ttdrm_step(const double *dev_vin, int64_t rho, const double *dev_core,
                           int64_t n, int64_t rhop, const int64_t *dev_idx_row, size_t N,
                           double *dev_vout, int stream);
/* DenseGaussianDRM.sketch_sparse (dense_gaussian_drm.py:59-66): C-order ravel of the
 * first m index rows, then column gather of mat (rank, cols): out (N, rank) row-major */
int ttsk_sparse_densedrm_gather(const double *dev_mat, int64_t rank, int64_t cols,
                                const int64_t *dev_idx, int64_t row_stride, const int *row_order,
                                const int64_t *shape, int m, size_t N, double *dev_out, int stream);
/* sketch_psi_sparse (sparse_sketch.py:8-36,49-69):
 *   Psi[a, idx[e], c] += val[e] * Lv[e,a] * Rv[e,c]   (Lv/Rv NULL -> rank 1, value 1)
 * Lv (N,l) and Rv (N,r) row-major as produced above; Psi (l,n,r) contiguous, zeroed by caller.
 * The kernel reduces runs of equal mode index in registers and issues one atomic per run and
 * output pair; with `dev_perm` = ttsk_sparse_sort_mode's permutation (nonzeros visited in order of
 * their mode index) that is a segmented reduction.  dev_idx_row == NULL with n == 1 sums over all
 * nonzeros: sketch_omega_sparse (sparse_sketch.py:39-46), Omega = (L * entries) R^T. */
int ttsk_sparse_psi(const double *dev_val, const int64_t *dev_idx_row, const int64_t *dev_perm, size_t N,
                    const double *dev_Lv, int64_t l, const double *dev_Rv, int64_t r,
                    int64_t n, double *dev_psi, int stream);

/* ---- SparseTensor x SparseGaussianDRM without (nnz x rank) panels (csrc/sparse_fused.hip) ----
 * sparse_gaussian_drm.py:29-44 + sparse_sketch.py:8-69 as one pass per mode over a resident, mode-ordered stream. */
/* multipliers of the Fortran-order flat index of fast_lazy_gaussian.pyx:60-71 incl. its 32-bit running product (host) */
int ttsk_sparse_flat_mult(const uint64_t *shape, int m, uint64_t *mult_out);
/* order of one mode's stream: ascending (index of physical row mode_row, flat index of the suffix rows r_rows as in
 * ttsk_sparse_mode_stream); the suffix as the secondary key makes the rows of a right-hand DRM table ascend inside a slice */
int ttsk_sparse_mode_order(const int64_t *dev_idx, int64_t row_stride, size_t N, const int *r_rows, const uint64_t *r_shape, int r_m,
                           int mode_row, int64_t n, int64_t *dev_perm, int stream);
/* stream of one mode: record pos = nonzero perm[pos] (perm: ttsk_sparse_sort_mode; NULL = identity): flat index of the
 * l_m index rows l_rows (prefix, shape l_shape) and of the r_m rows r_rows (suffix as the transposed tensor walks it),
 * the index of physical row mode_row as int32, the entry.  Rows are physical rows of the (d, N) index matrix. */
int ttsk_sparse_mode_stream(const int64_t *dev_idx, int64_t row_stride, const int64_t *dev_perm, size_t N, const int *l_rows,
                            const uint64_t *l_shape, int l_m, const int *r_rows, const uint64_t *r_shape, int r_m, int mode_row,
                            const double *dev_val, uint64_t *dev_fl, uint64_t *dev_fr, int32_t *dev_j, double *dev_v, int stream);
/* the same with 32-bit flat indices (20 instead of 28 bytes per record): TTSK_ERR_UNSUPPORTED when a prefix or suffix extent
 * reaches 2^31 */
int ttsk_sparse_mode_stream_u32(const int64_t *dev_idx, int64_t row_stride, const int64_t *dev_perm, size_t N, const int *l_rows,
                            const uint64_t *l_shape, int l_m, const int *r_rows, const uint64_t *r_shape, int r_m, int mode_row,
                            const double *dev_val, uint32_t *dev_fl, uint32_t *dev_fr, int32_t *dev_j, double *dev_v, int stream);
/* one DRM factor of a pass: kind 0 = ones (width 1), 1 = table[flat][w] (every possible prefix sampled once:
 * ttsk_sparse_normal_table / ttsk_sparse_sign_table; allocated with one spare row: rows are fetched in 16-byte units), 2 = normals sampled in the pass: ndtri(u(hash(flat + hash(rank_min + c)
 * + seed))), 3 = sparse-sign rows sampled in the pass (fast_lazy_gaussian.pyx:121-180; whole row <= 32 entries);
 * flat = src 0: prefix, 1: suffix, 2: prefix + j * mul, 3: suffix + j * mul (the prefix / suffix one mode longer) */
typedef struct {
    int kind, w, rank_min, src;
    uint64_t mul, seed;
    const double *table;
    int full, nnz;          /* kind 3 only: length of the whole sign row, its +-1 entries (sparse_sign_drm.py:34-51) */
} ttsk_sg_factor;
/* Psi[a, j, c] (+)= sum_{e: j_e = j} val_e A[e, a] B[e, c]   (dev_psi (wA, n, wB), zero-initialised by the caller)
 * and, with C != NULL, Omega += sum_e val_e C[e, a] B[e, c] (c_left) or sum_e val_e A[e, a] C[e, c]  (dev_omega,
 * zero-initialised).  Stream in ascending j (ttsk_sparse_mode_stream); dev_j == NULL: one slice.  Widths <= 32 (one
 * 16-column matrix tile per factor up to 16, 2 x 2 tiles per product beyond).  No atomics: the result is bit-reproducible. */
int ttsk_sparse_gauss_pass(const uint64_t *dev_fl, const uint64_t *dev_fr, const int32_t *dev_j, const double *dev_val, size_t N,
                           int64_t n, const ttsk_sg_factor *A, const ttsk_sg_factor *B, const ttsk_sg_factor *C, int c_left,
                           double *dev_psi, double *dev_omega, int stream);
int ttsk_sparse_gauss_pass_u32(const uint32_t *dev_fl, const uint32_t *dev_fr, const int32_t *dev_j, const double *dev_val, size_t N,
                           int64_t n, const ttsk_sg_factor *A, const ttsk_sg_factor *B, const ttsk_sg_factor *C, int c_left,
                           double *dev_psi, double *dev_omega, int stream);
/* stable sort permutation of the nonzeros by one index row (values < n): perm[i] = id of the i-th
 * nonzero in mode-index order.  Independent of the DRM: computed once per tensor and mode. */
int ttsk_sparse_sort_mode(const int64_t *dev_idx_row, size_t N, int64_t n, int64_t *dev_perm, int stream);

/* ---- solves ----------------------------------------------------------------
 * right_mul_pinv / left_mul_pinv (utils.py:98-109; SciPy lstsq -> LAPACK gelsd with
 * cond = eps): P = pinv(Omega) by one-sided Jacobi SVD on the device, singular values
 * below rcond*sigma_max dropped (rcond < 0 -> DBL_EPSILON).  Omega is (l, r)
 * contiguous; P is (r, l) contiguous.  The products A*P / P*B are ttsk_gemm calls. */
int ttsk_pinv(const double *dev_omega, int64_t l, int64_t r, double rcond, double *dev_pinv,
              int *host_rank /* may be NULL */, int stream);
/* The same in two phases for callers with several independent pseudo-inverses (assemble_sketched_tt,
 * sketch.py:400-443: one per mode): `begin` queues the fast path on `stream` and returns at once, `end`
 * waits for that stream, and runs the Jacobi SVD if the fast path was rejected.  One begin per library
 * stream may be outstanding; arguments of `end` repeat those of `begin`. */
int ttsk_pinv_begin(const double *dev_omega, int64_t l, int64_t r, double rcond, double *dev_pinv, int stream);
int ttsk_pinv_end(const double *dev_omega, int64_t l, int64_t r, double rcond, double *dev_pinv,
                  int *host_rank /* may be NULL */, int stream);
/* thin SVD of a small matrix (TensorTrain.round, tensor.py:446-484, and tt_svd.py:21-42, after a QR has
 * reduced the unfolding to its triangular factor): A (m, n) row-major, m >= n, by one-sided Jacobi --
 * n <= 1024 in one workgroup (asynchronous), 1024 < n <= 8192 over all compute units with a grid barrier
 * per round (blocks the host once).  US (m, n) = U diag(S), S (n) descending, Vt (n, n); A = US Vt. */
int ttsk_svd_small(const double *dev_A, int64_t m, int64_t n, double *dev_US, double *dev_S, double *dev_Vt,
                   int stream);
/* zero the strictly lower triangle of A (m, n) row-major: the triangular factor R = Q^T M of a thin
 * QR recovered by a product is upper triangular only up to rounding; np.linalg.qr (tensor.py:568)
 * returns exact zeros there, which is what keeps structurally zero singular values exactly zero. */
int ttsk_triu(double *dev_A, int64_t m, int64_t n, int stream);
/* thin QR of orth_step (sketch_dispatch.py:172, scipy.linalg.qr(mode="economic")):
 * A (m, n) row-major with m >= n is overwritten by Q (m, n); Householder with LAPACK's
 * sign convention.  R is not returned (the reference discards it). */
int ttsk_qr_thin(double *dev_A, int64_t m, int64_t n, int stream);
/* orth_step (sketch_dispatch.py:160-174) as ONE call with no read-back: Q (m, k) = qr_thin(Psi_mat pinv(Omega)), k = l,
 * Psi_mat (m, r2) row-major, Omega (l, r2) -- or Omega == NULL: Q = qr_thin(Psi_mat), k = r2 (hmt_sketch).  Same Q as
 * ttsk_pinv + product + ttsk_qr_thin on their fast paths (normal equations, CholeskyQR2, LAPACK's column signs) for
 * ranks up to 256.  The acceptance tests of those factorisations are NOT waited for: a rejection (Omega rank deficient
 * or kappa > 300, Psi_mat Omega^+ with kappa > 1e6) sets the stream's deferred flag and leaves Q meaningless; the caller
 * reads the flag once when the whole sketch is queued (ttsk_deferred_status) and then repeats it through ttsk_pinv /
 * ttsk_qr_thin.  TTSK_ERR_UNSUPPORTED outside the fast path (ranks > 256, TTSK_FAST_SOLVES=0). */
int ttsk_orth_step(const double *dev_psi, int64_t m, int64_t r2, const double *dev_omega, int64_t l, double *dev_q,
                   int stream);
/* The same in two pieces for orthogonal_sketch, whose d - 1 Omega are all known before its sequential loop over the modes
 * starts: the pseudo-inverses of `count` matrices of ONE shape (l, r) with every stage as one batched launch (fast path
 * as in ttsk_orth_step, min(l, r) <= 128, verdicts in `stream`'s deferred flag), then per mode Q = qr_thin(Psi_mat P)
 * with P (r2, l) given. */
int ttsk_pinv_batch_deferred(int count, const double *const *dev_omegas, int64_t l, int64_t r, double *const *dev_pinvs,
                             int stream);
int ttsk_orth_step_pinv(const double *dev_psi, int64_t m, int64_t r2, const double *dev_pinv, int64_t l, double *dev_q,
                        int stream);
/* *host_flag = 1 if a factorisation queued by ttsk_orth_step on `stream` was rejected since the last call; waits for
 * the stream and clears the flag. */
int ttsk_deferred_status(int stream, int *host_flag);
/* orthogonal_sketch / hmt_sketch of a tensor train with tensor-train DRMs as ONE call (sketch.py:44-151,
 * sketch_dispatch.py:160-193, 202-275 with method = orthogonal / hmt; tensor_train_drm.py:71-88 for the chains):
 * right chain (and, orthogonal, left chain + Omega_mu) on the kernels of ttsk_tt_sketch, the pseudo-inverses batched,
 * then per mode T = (Q_0 .. Q_{mu-1})^T-chain (x) X_mu, Q_mu = qr_thin(T R_mu Omega_mu^+) (hmt: qr_thin(T R_mu)).
 *   n, s, X       as ttsk_tt_sketch
 *   rt, DR        right DRM, true ranks and cores in its walking order (mode d-1 first), rt[0] = 1
 *   lt, DL        left DRM (lt[0] = 1), or DL == NULL: hmt_sketch
 *   cores_out[d]  core mu (k_{mu-1}, n[mu], k_mu) with k_mu = lt[mu+1] (orthogonal) or rt[d-1-mu] (hmt), k_{-1} = k_{d-1} = 1
 *   omega_out     orthogonal: d - 1 matrices (lt[mu+1], rt[d-1-mu])
 * Verdicts deferred as in ttsk_orth_step (ttsk_deferred_status on `stream` afterwards).  TTSK_ERR_UNSUPPORTED: ranks
 * beyond 256, k_{mu-1} n[mu] < k_mu, TTSK_FAST_SOLVES=0.  (Omega of one shape with min(l, r) <= 128: the pseudo-inverses
 * are batched launches; the sign reconstruction runs on stream + 1 and is joined back.) */
int ttsk_tt_orth_sketch(int d, const int64_t *n, const int64_t *s, const int64_t *lt, const int64_t *rt,
                        const double *const *X, const double *const *DL, const double *const *DR,
                        double *const *cores_out, double *const *omega_out, int stream);
/* the same for `count` tensor trains of ONE signature (n, s) against ONE pair of DRMs -- the streaming setting of
 * sketch.py:292-301, and the many same-shaped re-orthogonalisations of tt_gmres.py:293-299.  Up to 16 tensors whose
 * unfoldings are all at least twice as tall as wide, ranks <= 128: ONE chain of launches, every step (chain products,
 * pseudo-inverses, Gram / Cholesky / triangular products, sign reconstruction) over all tensors at once, one verdict for the
 * whole batch.  Otherwise as concurrent chains: tensor b on the library streams (stream + 2 (b mod 4), + 1), forked from and
 * joined back into `stream`, a verdict per tensor.
 *   X, cores_out     count * d pointers, tensor-major;   omega_out   count * (d - 1) pointers (orthogonal)
 *   dev_status       count ints in device memory: 1 = a fast factorisation of that tensor was rejected (repeat it on the
 *                    robust path: ttsk_pinv / ttsk_qr_thin); read after ttsk_sync(stream) */
int ttsk_tt_orth_sketch_batch(int count, int d, const int64_t *n, const int64_t *s, const int64_t *lt, const int64_t *rt,
                              const double *const *X, const double *const *DL, const double *const *DR,
                              double *const *cores_out, double *const *omega_out, int *dev_status, int stream);

/* The two products of a dense-tensor sketch with tensor-train DRMs that read the tensor, from ONE read of it (reference
 * dense_sketch.py:7-52, sketch_omega_dense / sketch_psi_dense, with the DRM matrices of tensor_train_drm.py:109-122):
 *   X (n0, Q, T) C order; C (n0, ll) the left DRM's first core; P (Q, r) the right DRM's matrix of one core less
 *   Z (ll, Q, T) = sum_b C[b, .] X[b, ., .]     the first left product (dense_sketch.py:13-14 at mu = 0)
 *   U (n0, r, T) = sum_q P[q, .] X[., q, .]     Psi_0 before its last core (dense_sketch.py:36-52 at mu = 0)
 * The same call one level down gives Z_1 and Psi_1 from Z_0 (rows (p', i_1), C = the left DRM's second core as a matrix): a
 * first extent beyond 64 runs as blocks of 64 rows (of 32 for ll > 20 or r > 40), Z summed over the blocks.
 * TTSK_ERR_UNSUPPORTED outside the kernel's cover: n0 a multiple of 32, T % 16 == 0, Q % 8 == 0, ll <= 32, r <= 64
 * (an odd r through a padded copy of P), Q T < 2^27, X, P and Z 16-byte aligned (the caller then forms the two
 * products separately). */
int ttsk_dense_first_pass(const double *dev_x, int64_t n0, int64_t Q, int64_t T, const double *dev_c, int64_t ll,
                          const double *dev_p, int64_t r, double *dev_z, double *dev_u, int stream);

/* ttsk_pinv for `count` (<= 32) matrices of ONE shape (l, r), min(l, r) <= 128: every stage of the fast attempt one batched
 * launch, the robust kernel queued behind it per matrix with that matrix's verdict as predicate; no read-back (utils.py:98-109
 * for the d - 1 Omega of an assembly).  TTSK_ERR_UNSUPPORTED outside that cover. */
int ttsk_pinv_batch(int count, const double *const *dev_omegas, int64_t l, int64_t r, double *const *dev_pinvs, int stream);
/* assemble_sketched_tt (sketch.py:400-443) as one call: direction 0 ("right") C_mu = Psi_mu pinv(Omega_mu), mu < d - 1,
 * C_{d-1} = Psi_{d-1}; direction 1 ("left") C_0 = Psi_0, C_{mu+1} = pinv(Omega_mu) Psi_{mu+1}.  lr / rr: the d - 1 sketch
 * ranks; psi[mu] (lr[mu-1], n[mu], rr[mu]) contiguous; omega[mu] (lr[mu], rr[mu]); work[mu]: rr[mu] * lr[mu] doubles that
 * receive pinv(Omega_mu).  Pair mu runs on stream (stream + mu) mod TTSK_NUM_STREAMS with ttsk_pinv's robust fallback
 * queued behind the fast attempt (no read-back); all streams are joined into `stream` before the call returns. */
int ttsk_tt_assemble(int d, const int64_t *n, const int64_t *lr, const int64_t *rr, const double *const *dev_psi,
                     const double *const *dev_omega, double *const *dev_cores_out, double *const *dev_work, int direction,
                     int stream);
/* ---- multi-GPU: one RCCL sum of the packed partial sketch ------------------
 * SketchContainer.__add__ across ranks (sketch_container.py:61-69). */
int ttsk_comm_unique_id(void *host_id128);                 /* rank 0: 128-byte id */
int ttsk_comm_init(const void *host_id128, int rank, int nranks);
int ttsk_comm_allreduce_sum(double *dev_buf, size_t n, int stream);
int ttsk_comm_reduce_sum(double *dev_buf, size_t n, int root, int stream);
/* every rank's n doubles in rank order into dev_recv (n * nranks doubles): placement of the blocks of a
 * rank-sharded sketch (blocked_stream_sketch, sketch.py:364-397,446-473 -- disjoint blocks, no sum) */
int ttsk_comm_allgather(const double *dev_send, double *dev_recv, size_t n, int stream);
int ttsk_comm_allreduce_max(double *dev_buf, size_t n, int stream);   /* elementwise max, in place */
/* A failed ttsk_comm_init leaves no communicator behind (the call can be repeated, and the process
 * exits cleanly); destroy is a no-op without one. */
int ttsk_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* TTSK_H */
