/*
 * corrla_rsvd.h -- C ABI of libcorrla_rsvd.so, the MI355X (gfx950) randomized-SVD engine
 * that drops in behind the RSVD hot path of wgurecky/CORRLA_RS.
 *
 * Every entry point cites the reference interface it replaces (paths relative to the
 * reference checkout).  Signatures use plain pointers and sizes only (no C++/torch types),
 * so the Rust shim, the Python `corrla_rs` module (ctypes) and bench.py
 * bind the same symbols.
 *
 * Conventions
 *   - Input A is described exactly like a faer `MatRef<T>`: (ptr, nrows, ncols, row_stride,
 *     col_stride), strides in ELEMENTS.  numpy C-order arrives as (rs=n, cs=1); a native faer
 *     `Mat` is column-major (rs=1, cs=m).  A is never modified.
 *   - Outputs are column-major like the reference's owned `Mat<T>` results
 *     (random_svd.rs:96-109): U m x k (leading dim ldu >= m), S k values (the k x 1 column),
 *     Vt k x n (leading dim ldvt >= k).  The caller allocates them.
 *   - l = min(rank + n_oversamples, min(m, n))   (random_svd.rs:77)
 *   - Signs: the reference returns whatever signs faer's SVD produced.  Here every triplet (u_i, s_i, v_i)
 *     is normalised so that the largest-magnitude component (first on ties) of the SHORT-side singular
 *     vector (length min(m, n): v_i for tall A, u_i for fat A) is positive.
 *   - Every function returns a corrla_status; corrla_last_error() gives the thread-local
 *     message.  The reference panics instead (random_svd.rs:98-107, mat_utils.rs:170-171).
 *   - `_dev` variants take DEVICE pointers (HIP) for A and the outputs and enqueue on the
 *     context's stream; host variants copy H2D/D2H around the same device path.
 *   - There is NO CPU fallback: without a usable gfx950 device every compute entry point
 *     fails with CORRLA_ENODEV.
 */
#ifndef CORRLA_RSVD_H
#define CORRLA_RSVD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Only the C ABI below is exported; every C++ symbol of the library has hidden visibility. */
#if defined(__GNUC__)
#define CORRLA_API __attribute__((visibility("default")))
#else
#define CORRLA_API
#endif

typedef struct corrla_ctx corrla_ctx;

typedef enum corrla_status {
  CORRLA_OK = 0,
  CORRLA_EINVAL = 1,   /* bad argument (rank == 0, rank > min(m,n), null pointer, bad stride ...) */
  CORRLA_ENOMEM = 2,   /* host or device allocation failed */
  CORRLA_EHIP = 3,     /* HIP runtime error */
  CORRLA_ECOMM = 4,    /* RCCL error / communicator not initialised */
  CORRLA_ENUMERIC = 5, /* non-finite data encountered in the small host factorizations */
  CORRLA_ENODEV = 6    /* no usable gfx950 device */
} corrla_status;

/* opts.flags */
#define CORRLA_OMEGA_ON_DEVICE 0x1u /* opts.omega is a device pointer (only for *_dev entry points) */
/* corrla_pca_*: how the column centring of center_mat_col (mat_utils.rs:482-502) is applied.
 * FUSED: the matrix is never rewritten; every product with the centred matrix is corrected by a rank-1 term,
 *        (A - 1 mu^T) X = A X - 1 (mu^T X), (A - 1 mu^T)^T Y = A^T Y - mu (1^T Y)  (SURVEY section 8 f1).
 * COPY : a centred copy is formed first, as the reference does (one more m x n buffer).
 * Default (neither flag): FUSED for f64, COPY for f32 (the fused form cancels digits when |mean| >> spread). */
#define CORRLA_PCA_CENTER_FUSED 0x2u
#define CORRLA_PCA_CENTER_COPY 0x4u
/* Thin-Q of the sketch (`y_mat.qr().compute_thin_q()`, random_svd.rs:38,57) by Householder TSQR with an explicit Q
 * (row panels reduced in LDS, pairwise tree over the R factors, reflectors applied in reverse) instead of the default
 * CholeskyQR2.  One panel holds l = min(rank + n_oversamples, n) <= 142 (f32) / 99 (f64) columns; wider sketches go
 * through column blocks of that width (projection against the finished blocks + panel, repeated).  On the row-sharded
 * entry points every rank reduces its rows, the root R factors are exchanged (one all-reduce of an nranks * l x l
 * buffer) and their stack is reduced redundantly; a call in which some shard has fewer than l rows keeps the default
 * path.  The environment variable CORRLA_QR=householder sets the flag for every call. */
#define CORRLA_QR_HOUSEHOLDER 0x8u
/* opts.seed is to be used as given even when it is 0.  Without this flag seed == 0 (and opts == NULL) means "no seed":
 * the library then draws a fresh sketch on every call, like the reference's unseeded thread_rng
 * (mat_utils.rs:161-175); on the row-sharded entry points that fresh seed is a function of the number of seedless
 * sharded calls made on the context only, so every rank draws the same Omega. */
#define CORRLA_SEED_EXPLICIT 0x10u
/* One-sweep power iteration: for a row-major f32 A with at most 512 columns, the pair
 * Y = A X, Z = A^T Y of random_svd.rs:42-51 (and of the sketch, :31) is computed as Z = A^T (A X) from ONE pass over A
 * with the 32-row tile of A held in LDS, wherever Y itself is not needed (the iterations without the in-loop thin-Q).
 * Same result up to rounding (the intermediate stays in f32 registers instead of being rounded to memory).  Other
 * shapes / dtypes / layouts ignore the flag.  The environment variable CORRLA_POWER_FUSED=1 sets it for every call. */
#define CORRLA_POWER_FUSED 0x20u
/* corrla_rsvd_sharded_dev_*: the local block is a COLUMN shard (m x n_local) of a fat matrix (m < sum of n_local)
 * instead of a row shard of a tall one.  The algorithm works on the tall view A^T (random_svd.rs:69-74), whose row shard
 * is this block transposed (a stride swap).  Outputs: U m x rank and S replicated on every rank, Vt rank x n_local =
 * this rank's columns of V^T.  opts->omega, when given, is m x l (the short side). */
#define CORRLA_SHARD_COLS 0x40u
/* Mixed-precision tall products (f32 row-major inputs, l <= 144; ignored elsewhere): the
 * products of the whole matrix with an l-wide factor -- power_iter's Y = A Omega, Z = A^T Y, Y = A Z (random_svd.rs:31,
 * 42-51) and the projection B = Q^T A (:80; CORRLA_MIXED_PROJECT=0 keeps that one exact) -- run on the bf16 matrix units
 * with an f32 accumulator, each f32 operand split on the fly into bf16 pieces whose products are exact in f32:
 *   BF16X6: three pieces (24 bits), six products -- the f32 product to f32 rounding (measured 3.5e-7 relative against
 *           3.2e-7 for the exact kernel) at 16/6 of the exact-f32 MFMA rate;
 *   BF16X3: two pieces (16 bits), three products -- 4.5e-6 relative, twice that rate again.
 * The thin-Qs, the core SVD and U = Q U~ stay in exact f32.  Off by default (the exact f32 path is the reference's
 * arithmetic); measured effect on the result: profiles/r03_mixed_accuracy.jsonl, DESIGN.md section 3.  The environment
 * variable CORRLA_SKETCH_MIXED=bf16x3|bf16x6 sets it for every call.  Mutually exclusive. */
#define CORRLA_SKETCH_BF16X3 0x80u
#define CORRLA_SKETCH_BF16X6 0x100u

/*
 * Options block.  Zero-initialise, set struct_size = sizeof(corrla_opts).  NULL opts == defaults.
 *   seed   : seed of the device Philox4x32-10 + Box-Muller generator that replaces
 *            random_mat_normal (mat_utils.rs:161-175; the reference draws from an unseeded
 *            thread_rng, so no seed value can reproduce it -- any N(0,1) draw is equivalent).
 *            A given seed makes the call deterministic; 0 without CORRLA_SEED_EXPLICIT = fresh draw per call.
 *   omega  : optional sketch matrix that replaces the draw at random_svd.rs:24 -- column-major
 *            n_t x l, n_t = min(m, n), leading dimension omega_ld (>= n_t), dtype of A.
 *            This is the test hook that lets the CPU oracle and the GPU share one Omega.
 */
typedef struct corrla_opts {
  uint32_t struct_size;
  uint32_t flags;
  uint64_t seed;
  const void* omega;
  int64_t omega_ld;
} corrla_opts;

/* Phase timings of the last rsvd / pca call on a context, milliseconds of DEVICE time: hipEvents are recorded on the
 * context's stream at the phase boundaries and resolved after the call has completed (no synchronisation inside the
 * call; replaces the SystemTime prints of random_svd.rs:125-140).  The phases add up to total_ms. */
typedef struct corrla_timings {
  double total_ms;      /* first to last event of the call */
  double sketch_ms;     /* Omega draw, Y = A*Omega            random_svd.rs:24,31   */
  double power_ms;      /* q x {Z = A^T Y, Y = A Z, norm}    random_svd.rs:35-56   */
  double qr_ms;         /* thin-Q orthonormalisations        random_svd.rs:38,57   */
  double project_ms;    /* B = Q^T A                         random_svd.rs:80      */
  double small_svd_ms;  /* svd of the l-wide core            random_svd.rs:89      */
  double finalize_ms;   /* U = Q*Ut, signs, output copies    random_svd.rs:92-109  */
  int32_t qr_passes;    /* Gram/whitening passes used by all orthonormalisations */
  int32_t n_collectives; /* all-reduces this rank issued during the call (row-sharded entry points; else 0) */
  double sketch_kernel_ms; /* device time of the sketch GEMM launch(es) of this call: hipEvents recorded on the
                              context's stream around Y = A*Omega (no extra synchronisation) */
  double host_enqueue_ms;  /* host wall clock spent enqueueing the call: the host runs ahead of the device */
  double collective_bytes; /* payload bytes of those all-reduces (n x l factors, l x l Gram matrices, scalars) */
  int32_t n_mixed_products; /* tall products of this call that ran on the bf16-split kernels (CORRLA_SKETCH_BF16X3 / X6) */
  int32_t reserved_;
  double knn_ms;        /* corrla_grad_mat_*: device time of the neighbour scan (hipEvents around its launches) */
  double fit_ms;        /* corrla_grad_mat_*: device time of the local least-squares fits; both 0 after any other call */
} corrla_timings;

/* ---- library / context ------------------------------------------------------------- */

/* "corrla_rsvd <version> gfx950" */
CORRLA_API const char* corrla_version(void);
/* thread-local message of the last failing call on this thread */
CORRLA_API const char* corrla_last_error(void);
/* number of visible HIP devices (0 when none / no driver) */
CORRLA_API int corrla_device_count(void);

/* One context = one device, one stream, one workspace arena (and optionally one RCCL
 * communicator).  Replaces faer's process-global Parallelism (mat_utils.rs:31,
 * random_svd.rs:122).  Calls on one context serialise. */
CORRLA_API corrla_status corrla_ctx_create(int device_ordinal, corrla_ctx** out);
CORRLA_API void corrla_ctx_destroy(corrla_ctx* ctx);
CORRLA_API corrla_status corrla_ctx_synchronize(corrla_ctx* ctx);
/* Per-phase device times cost one hipEvent record per phase boundary (~5 us of idle GPU each, 7 per call).  on = 0:
 * only total_ms, sketch_kernel_ms, the counters and host_enqueue_ms are filled in, the *_ms of the phases read 0.
 * Default: on.  (The reference prints wall-clock phase times unconditionally, random_svd.rs:125-140.) */
CORRLA_API corrla_status corrla_ctx_set_phase_timings(corrla_ctx* ctx, int on);
CORRLA_API corrla_status corrla_ctx_get_timings(corrla_ctx* ctx, corrla_timings* out);
/* Rank and size of the context's RCCL communicator as RCCL reports them (ncclCommUserRank / ncclCommCount);
 * nranks = 0 when corrla_ctx_comm_init has not been called. */
CORRLA_API corrla_status corrla_ctx_comm_info(corrla_ctx* ctx, int* rank, int* nranks);

/* ---- the hot path: random_svd -------------------------------------------------------
 * Replaces  pub fn random_svd<T>(a_mat: MatRef<T>, omega_rank, n_iter, n_oversamples)
 *             -> (Mat<T>, Mat<T>, Mat<T>)                      src/lib_math_utils/random_svd.rs:63-110
 * and, through it, the pyo3 surface corrla_rs.rsvd(a, n_rank, n_iters, n_oversamples)
 *                                                              src/lib_math_utils_py.rs:21-36
 * Host-pointer variants (A, U, S, Vt, opts->omega in host memory). */
CORRLA_API corrla_status corrla_rsvd_f32(corrla_ctx* ctx, const float* a, int64_t m, int64_t n, int64_t row_stride,
                              int64_t col_stride, int64_t rank, int64_t n_iter, int64_t n_oversamples,
                              const corrla_opts* opts, float* u, int64_t ldu, float* s, float* vt, int64_t ldvt);
CORRLA_API corrla_status corrla_rsvd_f64(corrla_ctx* ctx, const double* a, int64_t m, int64_t n, int64_t row_stride,
                              int64_t col_stride, int64_t rank, int64_t n_iter, int64_t n_oversamples,
                              const corrla_opts* opts, double* u, int64_t ldu, double* s, double* vt, int64_t ldvt);
/* Device-pointer variants: A, U, S, Vt are HIP device pointers on the context's device. */
CORRLA_API corrla_status corrla_rsvd_dev_f32(corrla_ctx* ctx, const float* a, int64_t m, int64_t n, int64_t row_stride,
                                  int64_t col_stride, int64_t rank, int64_t n_iter, int64_t n_oversamples,
                                  const corrla_opts* opts, float* u, int64_t ldu, float* s, float* vt, int64_t ldvt);
CORRLA_API corrla_status corrla_rsvd_dev_f64(corrla_ctx* ctx, const double* a, int64_t m, int64_t n, int64_t row_stride,
                                  int64_t col_stride, int64_t rank, int64_t n_iter, int64_t n_oversamples,
                                  const corrla_opts* opts, double* u, int64_t ldu, double* s, double* vt, int64_t ldvt);

/* ---- PCA caller ---------------------------------------------------
 * Replaces  PcaRsvd::new(x_mat, rank)                           src/lib_math_utils/pca_rsvd.rs:56-82
 *   means = mat_mean(x, 1); cx = center_mat_col(x); (_, s, vt) = random_svd(cx, rank, n_iter, n_oversamples)
 * and pyo3  rpca(a, n_rank, n_iters, n_oversamples)             src/lib_math_utils_py.rs:38-55
 * The reference hard-codes n_iter = 20, n_oversamples = min(n, 10) (pca_rsvd.rs:65-66; rpca ignores its last two
 * arguments); this ABI takes them explicitly and the Rust / Python shims pass the reference's values.
 * x: n_samples x n_dim, strided like a MatRef.  Outputs: means (n_dim values), s (rank values, the k x 1
 * column), components (rank x n_dim, column-major, leading dimension ldc >= rank).  The centred matrix is
 * formed on the device (one extra read + write of x); the columns' means come from the same MFMA kernel as
 * A^T Y.  Host-pointer and device-pointer variants. */
CORRLA_API corrla_status corrla_pca_f32(corrla_ctx* ctx, const float* x, int64_t n_samples, int64_t n_dim,
                                        int64_t row_stride, int64_t col_stride, int64_t rank, int64_t n_iter,
                                        int64_t n_oversamples, const corrla_opts* opts, float* means, float* s,
                                        float* components, int64_t ldc);
CORRLA_API corrla_status corrla_pca_f64(corrla_ctx* ctx, const double* x, int64_t n_samples, int64_t n_dim,
                                        int64_t row_stride, int64_t col_stride, int64_t rank, int64_t n_iter,
                                        int64_t n_oversamples, const corrla_opts* opts, double* means, double* s,
                                        double* components, int64_t ldc);
CORRLA_API corrla_status corrla_pca_dev_f32(corrla_ctx* ctx, const float* x, int64_t n_samples, int64_t n_dim,
                                            int64_t row_stride, int64_t col_stride, int64_t rank, int64_t n_iter,
                                            int64_t n_oversamples, const corrla_opts* opts, float* means, float* s,
                                            float* components, int64_t ldc);
CORRLA_API corrla_status corrla_pca_dev_f64(corrla_ctx* ctx, const double* x, int64_t n_samples, int64_t n_dim,
                                            int64_t row_stride, int64_t col_stride, int64_t rank, int64_t n_iter,
                                            int64_t n_oversamples, const corrla_opts* opts, double* means, double* s,
                                            double* components, int64_t ldc);
/* Sample-sharded PCA (one process per GPU, communicator from corrla_ctx_comm_init): x holds THIS rank's samples
 * (m_local x n_dim); the column means are all-reduced partial sums over the global sample count, and both centring
 * forms work unchanged (the rank-1 corrections of the fused form are linear in the rows: each rank corrects its own
 * partial product before the all-reduce).  means, s and components come out replicated on every rank. */
CORRLA_API corrla_status corrla_pca_sharded_dev_f32(corrla_ctx* ctx, const float* x, int64_t m_local, int64_t n, int64_t row_stride,
                                         int64_t col_stride, int64_t rank, int64_t n_iter, int64_t n_oversamples,
                                         const corrla_opts* opts, float* means, float* s, float* components, int64_t ldc);
CORRLA_API corrla_status corrla_pca_sharded_dev_f64(corrla_ctx* ctx, const double* x, int64_t m_local, int64_t n, int64_t row_stride,
                                         int64_t col_stride, int64_t rank, int64_t n_iter, int64_t n_oversamples,
                                         const corrla_opts* opts, double* means, double* s, double* components, int64_t ldc);

/* ---- the range finder: power_iter ---------------------------------------------------
 * Replaces  pub fn power_iter<T>(a_mat: MatRef<T>, omega_rank, n_iter) -> Mat<T>
 *                                                              src/lib_math_utils/random_svd.rs:15-59
 * `width` is the ALREADY OVERSAMPLED sketch width (the reference's `omega_rank` argument of
 * power_iter), width <= n.  Q: column-major m x width, leading dimension ldq.  Requires m >= 1,
 * n >= 1 (no fat->tall transpose here, exactly as in the reference). */
CORRLA_API corrla_status corrla_power_iter_f32(corrla_ctx* ctx, const float* a, int64_t m, int64_t n, int64_t row_stride,
                                    int64_t col_stride, int64_t width, int64_t n_iter, const corrla_opts* opts,
                                    float* q, int64_t ldq);
CORRLA_API corrla_status corrla_power_iter_f64(corrla_ctx* ctx, const double* a, int64_t m, int64_t n, int64_t row_stride,
                                    int64_t col_stride, int64_t width, int64_t n_iter, const corrla_opts* opts,
                                    double* q, int64_t ldq);
CORRLA_API corrla_status corrla_power_iter_dev_f32(corrla_ctx* ctx, const float* a, int64_t m, int64_t n, int64_t row_stride,
                                        int64_t col_stride, int64_t width, int64_t n_iter, const corrla_opts* opts,
                                        float* q, int64_t ldq);
CORRLA_API corrla_status corrla_power_iter_dev_f64(corrla_ctx* ctx, const double* a, int64_t m, int64_t n, int64_t row_stride,
                                        int64_t col_stride, int64_t width, int64_t n_iter, const corrla_opts* opts,
                                        double* q, int64_t ldq);

/* ---- the GEMM shim: par_matmul_helper -----------------------------------------------
 * Replaces  par_matmul_helper(res, lhs, rhs, beta, n_threads)  src/lib_math_utils/mat_utils.rs:20-33
 * for the two shapes the hot path uses it with (random_svd.rs:42-51): a large strided
 * matrix A (m x n, any of the two unit-stride layouts) times / transposed-times a skinny
 * column-major matrix.  res is OVERWRITTEN (alpha = None), res = beta * op(A) * X.
 *   trans == 0 : res (m x l) = beta * A   * X (n x l)
 *   trans == 1 : res (n x l) = beta * A^T * X (m x l)
 * X and res are column-major with leading dimensions ldx / ldres.  DEVICE pointers. */
CORRLA_API corrla_status corrla_matmul_dev_f32(corrla_ctx* ctx, int trans, const float* a, int64_t m, int64_t n,
                                    int64_t row_stride, int64_t col_stride, const float* x, int64_t ldx, int64_t l,
                                    float beta, float* res, int64_t ldres);
CORRLA_API corrla_status corrla_matmul_dev_f64(corrla_ctx* ctx, int trans, const double* a, int64_t m, int64_t n,
                                    int64_t row_stride, int64_t col_stride, const double* x, int64_t ldx, int64_t l,
                                    double beta, double* res, int64_t ldres);

/* ---- the Gaussian generator: random_mat_normal --------------------------------------
 * Replaces  random_mat_normal<T>(n_rows, n_cols)               src/lib_math_utils/mat_utils.rs:161-175
 * Fills a DEVICE matrix with i.i.d. N(0,1) from a counter-based Philox4x32-10 + Box-Muller
 * stream: element (i, j) of the logical matrix depends only on (seed, (row0 + i) * global_cols + j),
 * so any row shard generates its own rows bit-identically.  Storage strides in elements. */
CORRLA_API corrla_status corrla_fill_normal_dev_f32(corrla_ctx* ctx, float* p, int64_t rows, int64_t cols, int64_t row_stride,
                                         int64_t col_stride, uint64_t seed, int64_t row0, int64_t global_cols);
CORRLA_API corrla_status corrla_fill_normal_dev_f64(corrla_ctx* ctx, double* p, int64_t rows, int64_t cols, int64_t row_stride,
                                         int64_t col_stride, uint64_t seed, int64_t row0, int64_t global_cols);

/* ---- active-subspace gradient stage --------------------------------
 * Replaces  PolyGradientEstimator::new / grad_at            src/lib_math_utils/active_subspaces.rs:66-141
 *           ActiveSsRsvd::create_grad_mat                    src/lib_math_utils/active_subspaces.rs:215-229
 *           (linear_fit / quad_fit / jac_from_lin / jac_from_quad, src/lib_math_utils/stats_corr.rs:146-249)
 * x  : n_pts x k support points, row-major contiguous (ld = k); y: n_pts values of the scalar function;
 * xq : n_q x k query points (the reference uses the support points themselves);
 * est_order 1: least-squares hyper-plane through the n_nbrs nearest support points of each query (needs n_pts,
 *              n_nbrs > k + 1); 2: full quadratic [x, x_a x_b (a <= b), 1], build_vandermonde (stats_corr.rs:198-207;
 *              needs n_pts, n_nbrs > k (k + 3) / 2).  Limits: k <= 64, n_nbrs <= 512 (hence order 2 for k <= 30), and one
 *              query's neighbours must fit in 160 KiB of LDS; normal equations that do not fit next to them (order 2
 *              beyond k = 14) live in global memory.
 * g  : out_scale * gradients in the reference's k x n_q column-major layout: n_q rows of k contiguous values, row
 *      stride ldg >= k.  fit_svd (active_subspaces.rs:233-250) passes out_scale = 1 / sqrt(n_q) and hands g to
 *      corrla_rsvd_dev_f64 as the k x n_q matrix (row_stride 1, col_stride ldg).
 * n_regularised (optional): queries whose normal equations were numerically singular (the reference's pinv returns
 *      the minimum-norm fit there; this build retries with a 1e-10 relative ridge, and writes a zero gradient if that
 *      fails too).  Nearest neighbours are exact (brute force); equal distances are ordered by index. */
CORRLA_API corrla_status corrla_grad_mat_f64(corrla_ctx* ctx, const double* x, int64_t n_pts, int64_t k, const double* y,
                                             const double* xq, int64_t n_q, int est_order, int64_t n_nbrs, double out_scale,
                                             double* g, int64_t ldg, int* n_regularised);
CORRLA_API corrla_status corrla_grad_mat_dev_f64(corrla_ctx* ctx, const double* x, int64_t n_pts, int64_t k, const double* y,
                                                 const double* xq, int64_t n_q, int est_order, int64_t n_nbrs,
                                                 double out_scale, double* g, int64_t ldg, int* n_regularised);

/* ---- measurement hook ---------------------------------------------------------------
 * Times `reps` back-to-back launches of the sketch GEMM Y = A * X (random_svd.rs:31) with
 * hipEvents recorded on the context's stream (the stream the kernel runs on) and returns the
 * average kernel-sequence duration in milliseconds.  DEVICE pointers, layouts as in
 * corrla_matmul_dev_*.  Used by bench.py for roofline.achieved. */
CORRLA_API corrla_status corrla_time_sketch_dev_f32(corrla_ctx* ctx, const float* a, int64_t m, int64_t n, int64_t row_stride,
                                         int64_t col_stride, const float* x, int64_t ldx, int64_t l, float* y,
                                         int64_t ldy, int reps, double* avg_ms);
CORRLA_API corrla_status corrla_time_sketch_dev_f64(corrla_ctx* ctx, const double* a, int64_t m, int64_t n, int64_t row_stride,
                                         int64_t col_stride, const double* x, int64_t ldx, int64_t l, double* y,
                                         int64_t ldy, int reps, double* avg_ms);

/* ---- multi-GPU: row-sharded tall matrices ---------------------
 * One process per GPU.  Rank 0 calls corrla_comm_unique_id, the 128 bytes are broadcast by the
 * host program (torch.distributed / MPI / a file), then every rank calls corrla_ctx_comm_init.
 * The communicator is RCCL; collectives are enqueued on the context's stream. */
#define CORRLA_UNIQUE_ID_BYTES 128
CORRLA_API corrla_status corrla_comm_unique_id(void* out128);
CORRLA_API corrla_status corrla_ctx_comm_init(corrla_ctx* ctx, const void* unique_id128, int rank, int nranks);

/* Row-sharded random_svd: this rank holds rows [row0, row0 + m_local) of the TALL matrix
 * A (m_global x n, m_global >= n), unit column stride or unit row stride as above.
 * U_local: m_local x k (this rank's rows of U); S and Vt are replicated on every rank.
 * Exchanges per call: q+1 all-reduces of n x l (Z and B^T), the l x l Gram all-reduces of the
 * orthonormalisations, one scalar per power iteration.  DEVICE pointers. */
CORRLA_API corrla_status corrla_rsvd_sharded_dev_f32(corrla_ctx* ctx, const float* a_local, int64_t m_local, int64_t n,
                                          int64_t row_stride, int64_t col_stride, int64_t rank, int64_t n_iter,
                                          int64_t n_oversamples, const corrla_opts* opts, float* u_local, int64_t ldu,
                                          float* s, float* vt, int64_t ldvt);
CORRLA_API corrla_status corrla_rsvd_sharded_dev_f64(corrla_ctx* ctx, const double* a_local, int64_t m_local, int64_t n,
                                          int64_t row_stride, int64_t col_stride, int64_t rank, int64_t n_iter,
                                          int64_t n_oversamples, const corrla_opts* opts, double* u_local, int64_t ldu,
                                          double* s, double* vt, int64_t ldvt);

#ifdef __cplusplus
}
#endif
#endif /* CORRLA_RSVD_H */
