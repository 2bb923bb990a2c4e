// Algorithm driver for the RSVD hot path: the schedule of random_svd.rs:15-110 expressed over
// a device backend `Dev` that supplies the m-/n-sized operations (tall GEMMs, norms, fills).
// The product instantiates it with the HIP backend (hip_backend.hpp); tests/emu instantiates it
// with a host emulation backend to exercise THIS file's logic (shapes, fat/tall handling,
// orthonormalisation passes, sharded exchange points) without a GPU.  No arithmetic on m- or
// n-sized data happens in this file.
#pragma once
#include <cstdio>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <string>
#include <functional>
#include <vector>

#include "small_linalg.hpp"

namespace corrla {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
enum { ST_OK = 0, ST_EINVAL = 1, ST_ENOMEM = 2, ST_EHIP = 3, ST_ECOMM = 4, ST_ENUMERIC = 5, ST_ENODEV = 6 };

inline int64_t round_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// Skinny matrices (Omega, Y, Z, Q, B^T, small l x l operands) are column-major device
// buffers with a leading dimension padded to 64 elements and a column count padded to the
// kernel's column blocking; ALL padding is kept zero so the GEMM kernels never bounds-check them.
constexpr int kMaxColTiles = 9;  // 16-column MFMA tiles per workgroup column block (144 columns)
constexpr int kLdPad = 64;

struct ColBlocking {
  int tiles, nblk, nt;
  int64_t cols_alloc;
};
inline ColBlocking col_blocking(int64_t cols) {
  ColBlocking b;
  b.tiles = (int)std::max<int64_t>(1, (cols + 15) / 16);
  b.nblk = (b.tiles + kMaxColTiles - 1) / kMaxColTiles;
  b.nt = (b.tiles + b.nblk - 1) / b.nblk;
  b.cols_alloc = (int64_t)b.nblk * b.nt * 16;
  return b;
}

template <class T>
struct Skinny {
  T* p = nullptr;
  int64_t rows = 0, cols = 0, ld = 0, cols_alloc = 0;
  // a caller's buffer used in place as the destination of a product: only `cols` columns exist (no zero padding
  // beyond them), so kernels must not write past column cols - 1 and nothing may use it as a padded operand
  bool external = false;
  Skinny view_cols(int64_t c) const {
    Skinny s = *this;
    s.cols = c;
    return s;
  }
};

// Row-major view of memory: element (r, c) at p[r * ld + c].  cols_readable >= cols is how far
// each row may be read (vector loads); entries in [cols, cols_readable) are zero.
template <class T>
struct Big {
  const T* p = nullptr;
  int64_t rows = 0, cols = 0, ld = 0, cols_readable = 0;
};

// A skinny column-major matrix reinterpreted as the row-major matrix of its transpose.
template <class T>
inline Big<T> as_rowmajor_transposed(const Skinny<T>& y, int64_t ncols) {
  Big<T> b;
  b.p = y.p;
  b.rows = ncols;
  b.cols = y.rows;
  b.ld = y.ld;
  b.cols_readable = y.ld;
  return b;
}

// The TALL matrix A (mt x nt, mt >= nt unless the caller insists otherwise) as it sits in
// memory: either row-major (mem = A) or column-major (mem = A^T as a row-major nt x mt).
template <class T>
struct TallA {
  Big<T> mem;
  bool row_major = true;
  int64_t mt = 0, nt = 0;  // local rows (sharded) x columns
  // Implicit centring (PCA): the operator is A - 1 mu_short^T (mu_short: nt values, device) or
  // A - mu_tall 1^T (mu_tall: mt values); A itself is never rewritten.  At most one of them is set.
  const T* mu_short = nullptr;
  const T* mu_tall = nullptr;
};

struct RunOpts {
  uint64_t seed = 0x5eedull;
  bool seed_explicit = false;   // false: the entry point draws a fresh per-call seed (Dev::fresh_seed)
  const void* omega = nullptr;  // column-major nt x l, dtype T
  int64_t omega_ld = 0;
  bool omega_on_device = false;
  bool sharded = false;  // rows of A are sharded over the communicator
  int pca_center = 0;    // 0 default, 1 fused, 2 centred copy (corrla_pca_* only)
  bool qr_householder = false;  // thin-Q by Householder TSQR instead of CholeskyQR2 (CORRLA_QR_HOUSEHOLDER)
  bool power_fused = false;     // one-sweep Z' = A^T (A Z) where it applies (CORRLA_POWER_FUSED)
  int mixed_planes = 0;         // 0: exact f32 / f64 products; 2 / 3: the tall products of the range finder on the bf16 matrix
                                // pipe, operands split into 2 ("bf16x3") / 3 ("bf16x6") bf16 pieces, f32 accumulate
  bool mixed_project = true;    // ... and the projection B = Q^T A (random_svd.rs:80) as well (CORRLA_MIXED_PROJECT=0: exact)
  int poison_core = 0;          // TEST HOOK (env CORRLA_TEST_POISON_CORE = 1 NaN / 2 inf): one entry of the l x l core of
                                // random_svd.rs:89 is overwritten before its SVD; the call must end with ST_ENUMERIC
};
// bits of the device need_next words besides bit 0 (k::kNeedNonFinite / k::kNeedNullCols in hip_kernels.hpp)
constexpr int kFlagNonFinite = 2, kFlagNullCols = 4;

struct Timings {
  // DEVICE time of each phase: elapsed time between events recorded on the context's stream at the phase boundaries
  // (Dev::phase_mark), resolved after the call has completed -- no synchronisation inside the call
  double total_ms = 0, sketch_ms = 0, power_ms = 0, qr_ms = 0, project_ms = 0, small_svd_ms = 0, finalize_ms = 0;
  double host_enqueue_ms = 0;  // host wall clock spent enqueueing the call (the host runs ahead of the device)
  int qr_passes = 0;
  int n_collectives = 0;        // all-reduces issued by this call on this rank (row-sharded entry points)
  double collective_bytes = 0;  // payload bytes of those all-reduces
  double knn_ms = 0, fit_ms = 0;  // gradient stage (corrla_grad_mat_*): neighbour scan / local fits
  int n_mixed_products = 0;     // tall products that ran on the bf16-split kernels (RunOpts::mixed_planes)
  // breakdown of qr_ms (only filled when phase profiling is on): Gram GEMM, D2H + analysis, host Cholesky /
  // inverse, H2D + apply GEMM
  double qr_gram_ms = 0, qr_down_ms = 0, qr_host_ms = 0, qr_apply_ms = 0;
  // device time of the sketch GEMM Y = A*Omega of this call (events on the compute stream; no extra sync)
  double sketch_kernel_ms = 0;
};

struct PhaseTimer {
  using clk = std::chrono::steady_clock;
  clk::time_point t0;
  PhaseTimer() : t0(clk::now()) {}
  double lap() {
    auto t1 = clk::now();
    double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    t0 = t1;
    return ms;
  }
};

template <class Dev, class T>
struct RsvdDriver {
  Dev& dev;
  Timings tm;
  bool profile_phases;  // synchronise at phase boundaries to attribute time
  bool qr_householder = false;  // set from RunOpts by the entry points (random_svd_tall, power_iter)
  static constexpr const T* kNone = nullptr;

  explicit RsvdDriver(Dev& d, bool profile = false) : dev(d), profile_phases(profile) {}

  // ---- op(A) * skinny ---------------------------------------------------------------
  // Y (mt x L) = scale * A * X (nt x L)                         random_svd.rs:31,47-51
  int mixed_planes_ = 0;  // set from RunOpts while the range finder runs (power_iter), see RunOpts::mixed_planes
  void a_times(const TallA<T>& a, const Skinny<T>& x, Skinny<T>& y, const T* scale_dev) {
    if (mixed_planes_ && dev.template mixed_fits<T>(!a.row_major, a.mem, x, y))
      dev.gemm_mixed(!a.row_major, a.mem, x, y, scale_dev, mixed_planes_), ++tm.n_mixed_products;
    else if (a.row_major)
      dev.gemm_nn(a.mem, x, y, scale_dev);
    else
      dev.gemm_tn(a.mem, x, y, scale_dev);
    if (a.mu_short || a.mu_tall) {
      // (A - 1 mu^T) X = A X - 1 (mu^T X);  (A - mu 1^T) X = A X - mu (1^T X)
      T* v = dev.template alloc_scalar<T>((int)x.cols_alloc);
      dev.weighted_colsum(x, a.nt, a.mu_short, v);
      dev.rank1_sub(y, a.mt, a.mu_tall, v, scale_dev);
    }
  }
  // Z (nt x L) = scale * A^T * Y (mt x L); all-reduced when rows are sharded   random_svd.rs:42-46,80
  void at_times(const TallA<T>& a, const Skinny<T>& y, Skinny<T>& z, const T* scale_dev, bool sharded) {
    if (mixed_planes_ && dev.template mixed_fits<T>(a.row_major, a.mem, y, z))
      dev.gemm_mixed(a.row_major, a.mem, y, z, scale_dev, mixed_planes_), ++tm.n_mixed_products;
    else if (a.row_major)
      dev.gemm_tn(a.mem, y, z, scale_dev);
    else
      dev.gemm_nn(a.mem, y, z, scale_dev);
    if (a.mu_short || a.mu_tall) {
      // (A - 1 mu^T)^T Y = A^T Y - mu (1^T Y);  (A - mu 1^T)^T Y = A^T Y - 1 (mu^T Y).  Row-sharded: the correction is
      // linear in the rows, so every rank corrects its own partial product and the all-reduce below sums both parts.
      T* v = dev.template alloc_scalar<T>((int)y.cols_alloc);
      dev.weighted_colsum(y, a.mt, a.mu_tall, v);
      dev.rank1_sub(z, a.nt, a.mu_short, v, scale_dev);
    }
    // (only the l columns that exist: the padding columns up to cols_alloc are zero on every rank and stay zero)
    if (sharded) dev.allreduce(z.p, (size_t)z.ld * (size_t)z.cols);
  }

  // ---- thin-Q orthonormalisation (replaces y.qr().compute_thin_q(), random_svd.rs:38,57) ----
  // Iterated Cholesky-QR on the device-computed Gram matrix: G = Y^T Y (tall GEMM + optional all-reduce),
  // Cholesky factor and inverse (device kernel when l <= 176 and the optimistic path applies, host f64
  // otherwise), Y <- Y * R^-1 (tall GEMM).  A pass whose Gram was already within 0.25 of I leaves Y orthonormal
  // to O(eps), so well-conditioned sketches take two passes.  Numerically singular Grams get a diagonal shift
  // (shifted CholeskyQR, Fukaya et al. 2020); if three shifted passes do not recover full rank the deficient
  // directions are exact-null and are dropped through an eigen-decomposition of G (orthonormalize_core then
  // returns r < l), after which complete_basis fills columns [r, l) with an orthonormal completion -- the
  // "arbitrary completion" a Householder QR of a rank-deficient matrix carries (random_svd.rs:153-196).
  // Returns the number of orthonormal columns (l unless the matrix has fewer rows than columns).
  // `rough`: stop after the first clean Cholesky pass -- enough for the in-loop re-orthonormalisations
  // (random_svd.rs:37-39), whose only role is to keep the sketch well conditioned; the span is unchanged.
  static constexpr size_t kStatusBytes = 32;  // one device status record (CholStatus)
  int st_slots_ = 0;                          // records in the pool of the optimistic run (sized from n_iter)
  struct Pending {
    int slot, npass;
    bool rough;
    int per_pass;  // status records per pass: 1 (single factorisation) or 2 (2 x 2 blocked)
    void* st = nullptr;  // scratch status records of a device-robust thin-Q (diagnostics only)
    int st_per_pass = 1;
    bool is_svd = false; // the convergence verdict of the core SVD's fixed number of sweeps
    int flag_slot = -1;  // >= 0: a device-robust Cholesky-QR (orthonormalize_device); clean iff flags[flag_slot + npass - 1] == 0
  };
  static constexpr int kRobustPasses = 8;  // most passes a device-robust thin-Q enqueues (2 always, the rest conditional)
  bool robust_needs_more_ = false;         // pending_clean: a device-robust thin-Q ran out of enqueued passes
  bool svd_needs_more_ = false;            // pending_clean: the core SVD ran out of enqueued sweeps
  bool svd_needs_v_ = false;               // pending_clean: the core SVD's W-only shortcut failed its verification
  bool null_cols_seen_ = false;            // pending_clean: some device thin-Q pass re-seeded null columns (rank-deficient sketch)
  int* flags_pool_ = nullptr;              // device words: need_next of every pass of every robust thin-Q of the call
  int flags_cap_ = 0, flags_used_ = 0;
  bool defer_status_ = false;
  bool optimistic_dirty_ = false;  // a pass of the optimistic run took the host-controlled loop: repeat the call
  void* st_pool_ = nullptr;
  int st_used_ = 0;
  std::vector<Pending> pending_;
  // rows of the WHOLE tall matrix (all ranks), fetched lazily by one scalar all-reduce the first time a rank-invariant
  // decision needs it; -1 = unknown.  Unsharded calls use the local row count.
  int64_t m_local_ = 0, m_global_ = -1;
  int64_t m_local_override_ = -1;  // >= 0: this rank's true row count (an empty shard runs on one stand-in zero row)
  int64_t global_rows(bool sharded) {
    if (!sharded) return m_local_;
    if (m_global_ < 0) m_global_ = dev.allreduce_sum_host(m_local_);
    return m_global_;
  }

  // ---- Householder thin-Q (random_svd.rs:38,57 as written; CORRLA_QR_HOUSEHOLDER) ------------------------------------
  // Row-sharded calls: every rank must take the same branch, and a shard with fewer rows than columns cannot carry its
  // own l x l R factor -> one scalar all-reduce per call counts such shards; any -> the call keeps the default path.
  int hh_short_shards_ = -1;
  bool householder_usable(const Skinny<T>& y, bool sharded) {
    if (!qr_householder) return false;
    if (!sharded) return y.rows >= y.cols;
    if (hh_short_shards_ < 0) hh_short_shards_ = (int)dev.allreduce_sum_host(m_local_ < y.cols ? 1 : 0);
    return hh_short_shards_ == 0;
  }
  // One column panel (at most 142 columns in f32 / 99 in f64: 2 l x l must fit in LDS): TSQR over this rank's rows.
  // Row-sharded: the P root R factors are stacked (one all-reduce of a zero-padded P l x l buffer = an all-gather), every
  // rank takes the thin-Q of the stack redundantly (same arithmetic -> same bits) and feeds its own l x l block of it
  // into the down sweep:  Y = diag(Q_r) [R_r] = diag(Q_r) Q' R  =>  Q = [Q_r C_r].  (R-factor exchange.)
  void householder_panel(Skinny<T>& y, Skinny<T>& tmp, bool sharded) {
    const int64_t l = y.cols;
    auto h = dev.householder_up(y, tmp);
    if (!sharded) {
      dev.householder_down(h, y, tmp);
      return;
    }
    const int64_t np = dev.nranks(), rk = dev.rank();
    Skinny<T> stack = dev.template alloc_skinny<T>(np * l, l), t2 = dev.template alloc_skinny<T>(np * l, l);
    Skinny<T> rv, cv;
    rv.p = h.r_root();
    rv.rows = rv.cols = rv.ld = rv.cols_alloc = l;
    dev.copy_block(rv, 0, 0, l, l, stack, rk * l, 0);
    dev.allreduce(stack.p, (size_t)stack.ld * (size_t)stack.cols_alloc);
    dev.householder_thin_q(stack, t2);
    cv.p = (T*)dev.alloc_zeroed_bytes((size_t)l * (size_t)l * sizeof(T));
    cv.rows = cv.cols = cv.ld = cv.cols_alloc = l;
    dev.copy_block(stack, rk * l, 0, l, l, cv, 0, 0);
    dev.householder_down(h, y, tmp, (const T*)cv.p);
  }
  // Any width: column blocks of at most one LDS panel, block j = thin-Q of (I - Q_<j Q_<j^T) Y_j, repeated (block
  // classical Gram-Schmidt with re-orthogonalisation around the Householder panels).  The panel Q is orthonormal
  // whatever the rank of its input; what a repeat restores is orthogonality ACROSS blocks: a panel whose input had an
  // overlap t = Q_<j^T X with sigma_max(t) <= 1/2 leaves (I - P) X with singular values >= 0.87, so its thin-Q is
  // orthogonal to Q_<j to O(eps).  Full-rank sketches stop after the second pass (the first panel's output already is
  // nearly orthogonal); a block that the first projection cancelled down to rounding noise -- noise that lies mostly IN
  // span(Q_<j) -- takes a third.  The overlap is read on the host (this mode runs with the host in the loop anyway) from
  // the all-reduced t, so row-sharded ranks decide alike.  The projections are tall MFMA products.
  void householder_thin_q_any_width(Skinny<T>& y, Skinny<T>& tmp, bool sharded) {
    const int64_t l = y.cols, m = y.rows;
    int64_t wmax = dev.template householder_max_width<T>();
    if (const char* e = std::getenv("CORRLA_HH_BLOCK")) wmax = std::max<int64_t>(1, std::min<int64_t>(wmax, std::atoll(e)));
    if (l <= wmax) {
      householder_panel(y, tmp, sharded);
      return;
    }
    const int64_t nb = (l + wmax - 1) / wmax, w = (l + nb - 1) / nb;
    std::vector<double> th;
    for (int64_t c0 = 0; c0 < l; c0 += w) {
      const int64_t wb = std::min(w, l - c0);
      Skinny<T> yb = dev.template alloc_skinny<T>(m, wb), tb = dev.template alloc_skinny<T>(m, wb);
      Skinny<T> pb = dev.template alloc_skinny<T>(m, wb);
      dev.copy_block(y, 0, c0, m, wb, yb, 0, 0);
      for (int pass = 0; pass < 5; ++pass) {
        double overlap2 = 0.0;
        if (c0 > 0) {
          Skinny<T> t = dev.template alloc_skinny<T>(c0, wb);
          dev.gemm_nn(as_rowmajor_transposed(y, c0), yb, t, kNone);  // Q_<j^T Y_j
          if (sharded) dev.allreduce(t.p, (size_t)t.ld * (size_t)t.cols_alloc);
          if (pass >= 1) {
            th.resize((size_t)(c0 * wb));
            dev.download_skinny(t, c0, wb, th.data());
            for (double v : th) overlap2 += v * v;
          }
          dev.gemm_tn(as_rowmajor_transposed(y, c0), t, pb, kNone);  // Q_<j (Q_<j^T Y_j)
          dev.sub_inplace(yb, pb);
        }
        householder_panel(yb, tb, sharded);
        if (c0 == 0 || (pass >= 1 && overlap2 <= 0.25)) break;
      }
      dev.copy_block(yb, 0, 0, m, wb, y, 0, c0);
    }
  }

  int64_t orthonormalize(Skinny<T>& y, Skinny<T>& tmp, bool sharded, bool rough = false) {
    const int64_t l = y.cols;
    if (householder_usable(y, sharded)) {
      // random_svd.rs:38,57 as written: Householder QR, explicit thin Q (orthonormal for any rank of y)
      PhaseTimer qt0;
      householder_thin_q_any_width(y, tmp, sharded);
      ++tm.qr_passes;
      subphase(tm.qr_gram_ms, qt0);
      return l;
    }
    int64_t r = orthonormalize_core(y, tmp, sharded, rough);
    // A Householder thin-Q (random_svd.rs:38,57) is orthonormal whatever the rank of its input: the directions a
    // rank-deficient sketch does not determine are an arbitrary orthonormal completion.  Reproduce that instead of
    // leaving zero columns (in-loop, the completion re-seeds directions the next product with A can pick up again).
    // Sharded: every rank must take the same branch (complete_basis issues all-reduces), so the test uses the GLOBAL
    // row count -- the local one differs between uneven shards -- and r itself derives from all-reduced Gram matrices.
    if (r < l && !defer_status_ && (sharded ? global_rows(true) : y.rows) >= l) r = complete_basis(y, r, sharded);
    return r;
  }

  // columns [r, l) of y <- orthonormal vectors orthogonal to the first r columns: Gaussian block, two projection
  // passes against Q_r, Cholesky-QR of the remainder
  int64_t complete_basis(Skinny<T>& y, int64_t r, bool sharded) {
    const int64_t l = y.cols, c = l - r, m = y.rows;
    Skinny<T> yc = dev.template alloc_skinny<T>(m, c);
    Skinny<T> pc = dev.template alloc_skinny<T>(m, c);
    Skinny<T> tc = dev.template alloc_skinny<T>(m, c);
    dev.fill_normal(yc.p, m, c, (int64_t)1, yc.ld, (uint64_t)(0x9e3779b97f4a7c15ull ^ (uint64_t)(131 * r + l)), (int64_t)0, c);
    if (r > 0) {
      for (int pass = 0; pass < 2; ++pass) {
        Skinny<T> t = dev.template alloc_skinny<T>(r, c);
        dev.gemm_nn(as_rowmajor_transposed(y, r), yc, t, kNone);  // Q_r^T Yc
        if (sharded) dev.allreduce(t.p, (size_t)t.ld * (size_t)t.cols_alloc);
        dev.gemm_tn(as_rowmajor_transposed(y, r), t, pc, kNone);  // Q_r (Q_r^T Yc)
        dev.sub_inplace(yc, pc);
      }
    }
    const int64_t rc = orthonormalize_core(yc, tc, sharded, false);
    dev.copy_cols(yc, y, r, rc);
    return r + rc;
  }

  // Thin-Q without the host, whatever the conditioning or rank of the sketch (optimistic runs, l <= 144): iterated
  // Cholesky-QR in which the DEVICE decides what a pass does (k::chol_inv_kernel, robust form):
  //   pass 1     G = Y^T Y, shifted Cholesky (G + s I never breaks down; Y R_s^-1 keeps the span and lifts the weak
  //              directions by up to 1 / sqrt(s_rel) relative to the strong ones),
  //   pass 2..   shifted again while the Gram is further than 0.25 from I, plain Cholesky once it is within (the
  //              product of that pass is orthonormal to O(eps)), a first-order (I + E)^(-1/2) when it is within 2e-4;
  //              an exactly zero column (or a pivot that fails despite the shift) is a NULL column: a zero column of
  //              R^-1, replaced by a random column after the product,
  //   pass 3..P  carry the need_next word of the pass before as the run_if of every launch: nothing happens once a
  //              pass has seen a Gram within 0.25 of I and produced no null column.
  // Well-conditioned sketches cost what CholeskyQR2 cost plus a few empty launches; ill-conditioned and rank-deficient
  // ones no longer repeat the whole call through the host-controlled loop (16384^2 f32 with sigma_i = 0.97^i:
  // 11.5 ms, sigma_i = 0.7^i: 37 ms, against 5.1 ms for a flat spectrum).  All products run in place (every
  // workgroup / wave reads exactly the rows it writes, and all of them before it stores).  `rough` (in-loop,
  // random_svd.rs:37-39): pass 1 only.  The verdict -- need_next of the last pass must be 0 -- is read with the
  // status records at the end of the call; a sketch that needs more passes than were enqueued repeats on the old path.
  // `polish`: the input is expected to be orthonormal already up to a small defect (the W / sigma factor of the core
  // SVD): no shift, and every pass after the first is conditional.
  // Wider sketches (two column blocks, l > 144) cannot run their products in place: the unconditional passes ping-pong
  // between y and tmp as before, and the conditional ones come in PAIRS gated by the same word, so that a skipped pair
  // leaves the data where the host expects it.  Factors wider than one chol_inv_kernel (l > 176 f32 / 152 f64) use the
  // 2 x 2 blocked form: gram_inspect decides and adds the shift for the whole Gram, the two diagonal blocks are
  // factorised by the robust kernel (null columns per block), combine_need forms the verdict of the pass.
  int64_t orthonormalize_device(Skinny<T>& y, Skinny<T>& tmp, bool sharded, bool rough, bool polish) {
    const int64_t l = y.cols;
    PhaseTimer qt0;
    const bool inplace = dev.qr_inplace_fits(l);
    const bool single = dev.template device_chol_fits<T>(l);
    // how many passes are enqueued is a property of the CONTEXT: it starts at the two (polish: one) a well-conditioned
    // sketch needs -- no conditional launch at all -- and doubles (2 -> 4 -> 8) whenever a call ends with a thin-Q
    // still asking for more; the call is then repeated on the device with the higher count (random_svd_tall)
    const int level = std::max(2, std::min(dev.robust_passes(), kRobustPasses));
    // in-loop (rough, random_svd.rs:37-39): ONE shifted pass is enough whenever its pivots say the sketch was not too
    // ill-conditioned for it (need_ratio); otherwise the same conditional chain as the final thin-Q lifts what can be
    // lifted and re-seeds the rest (a context at level 2 has none enqueued: it escalates like for the final thin-Q)
    const bool inloop = rough && !polish;
    const int npass_even = level;  // 2, 4, 8
    const int npass = inloop ? (level <= 2 ? 1 : npass_even - 1) : (polish ? (level <= 2 ? 1 : 3) : level);
    const int always = (polish || inloop) ? 1 : 2;
    int* need = flags_pool_ + flags_used_;
    int* null_mask = dev.alloc_flags((int)l);
    int* need_blk = dev.alloc_flags(2);
    void* st_scratch = dev.alloc_zeroed_bytes((size_t)2 * npass * kStatusBytes);
    void* insp = single ? nullptr : dev.template alloc_inspect<T>();
    Skinny<T> gd = dev.template alloc_skinny<T>(l, l);
    Skinny<T> md = dev.template alloc_skinny<T>(l, l);
    const double eps0 = (double)std::numeric_limits<T>::epsilon();
    // shift: as small as the rounding error of the computed Gram allows (~ eps sqrt(m) relative to its largest diagonal
    // entry; never below the 16 eps of the host-controlled path).  A small shift lifts the weak directions further per
    // pass AND keeps them more accurate (16384 x 16384 f32, sigma_i = 0.7^i: the 10th singular value to 1.6e-5 with
    // 16 eps, to 2.3e-4 with 1100 eps); a pivot that fails all the same only costs that column (null -> re-seeded).
    // It must also stay above the rounding error of the factorisation itself, or pivots of nearly dependent columns
    // come out negative and the column -- a direction f64 may still resolve -- is discarded as null: the elimination kernel
    // stays within the 16 eps in practice (thousands of fuzz cases), the Schur complement of the 2 x 2 blocked form is formed
    // by a product and carries ~l eps (tools/fuzz_parity.py seed 22 case 133 lost 2e-4 in a trailing singular value at l = 160 with 16 eps).  (Clamping
    // failed pivots to the shift instead keeps the span too, but the inverse factor of dozens of clamped columns overflows.)
    double shift_rel = eps0 * std::max({16.0, 0.25 * std::sqrt((double)std::max<int64_t>(y.rows, 1)), single ? 0.0 : 2.0 * (double)l});
    if (const char* e = std::getenv("CORRLA_QR_SHIFT_SCALE")) shift_rel *= std::atof(e);  // experiments
    // in-loop: a direction whose residual is below the shift would leave the pass less than half normalised; such
    // half-lifted columns measurably hurt (DMDc snapshots: 7e-7 instead of 1e-14 in the 8th singular vector), so they
    // are re-seeded at random like the completion of the host-controlled path (which drops below 1e-2)
    float nullx = 1.f;
    if (const char* e = std::getenv("CORRLA_QR_NULL_EXCESS")) nullx = (float)std::atof(e);  // experiments
    const int64_t n1 = single ? l : round_up((l + 1) / 2, (int64_t)4), n2 = l - n1;
    T* minus_one = nullptr;
    if (!single) {
      minus_one = dev.template alloc_scalar<T>(1);
      dev.fill_const(minus_one, (int64_t)1, (T)-1);
    }
    for (int pass = 0; pass < npass; ++pass) {
      // gate: in place, the pass before; in pairs, the pass before the PAIR
      const int gate = pass < always ? -1 : (inplace ? pass - 1 : always + ((pass - always) / 2) * 2 - 1);
      dev.set_run_if(gate < 0 ? nullptr : need + gate);
      Skinny<T> yv = y.view_cols(l);
      dev.gemm_nn(as_rowmajor_transposed(y, l), yv, gd, kNone);
      if (sharded) dev.allreduce(gd.p, (size_t)gd.ld * (size_t)gd.cols_alloc);  // unconditional: ranks stay in step
      const int shift_mode = polish ? 2 : (pass == 0 ? 1 : 0);
      // in-loop (rough): directions below the level one shifted pass can lift are re-seeded at random, like the
      // completion of the host-controlled path -- the next products with A pull the re-seeded columns back into range(A)
      // (final thin-Q: the first two passes only lift; what is still below the shift level in the third has no
      // independent information -- amplified rounding noise is not even linearly independent, it would be lifted and
      // crushed again for ever -- and is re-seeded too)
      const float nx = (!polish && pass >= 2) ? nullx : 0.f;
      const float need_ratio = (inloop && pass == 0) ? 1e-2f : 0.f;
      if (single) {
        dev.chol_inv_robust(gd, l, (T)(4.0 * eps0), (float)shift_rel, shift_mode, nx, md, st_scratch, pass, need + pass, null_mask,
                            nullptr, need_ratio);
      } else {
        dev.gram_inspect(gd, l, (float)shift_rel, shift_mode, insp);
        const void* shp = dev.template inspect_shift_ptr<T>(insp);
        Skinny<T> g11 = dev.template alloc_skinny<T>(n1, n1), g12 = dev.template alloc_skinny<T>(n1, n2);
        Skinny<T> g22 = dev.template alloc_skinny<T>(n2, n2), x11 = dev.template alloc_skinny<T>(n1, n1);
        Skinny<T> x22 = dev.template alloc_skinny<T>(n2, n2), r12 = dev.template alloc_skinny<T>(n1, n2);
        Skinny<T> t22 = dev.template alloc_skinny<T>(n2, n2), t12 = dev.template alloc_skinny<T>(n1, n2);
        Skinny<T> x12 = dev.template alloc_skinny<T>(n1, n2);
        dev.copy_block(gd, 0, 0, n1, n1, g11, 0, 0);
        dev.copy_block(gd, 0, n1, n1, n2, g12, 0, 0);
        dev.copy_block(gd, n1, n1, n2, n2, g22, 0, 0);
        dev.chol_inv_robust(g11, n1, (T)(4.0 * eps0), 0.f, 2, nx, x11, st_scratch, 2 * pass, need_blk, null_mask, shp, need_ratio);
        dev.gemm_nn(as_rowmajor_transposed(x11, n1), g12, r12, kNone);  // R12 = X11^T G12 (zero rows for null columns)
        dev.gemm_nn(as_rowmajor_transposed(r12, n2), r12, t22, kNone);  // R12^T R12
        dev.sub_inplace(g22, t22);                                      // Schur complement (carries the shift of G22)
        dev.chol_inv_robust(g22, n2, (T)(4.0 * eps0), 0.f, 2, nx, x22, st_scratch, 2 * pass + 1, need_blk + 1, null_mask + n1, shp,
                            need_ratio);
        dev.gemm_tn(as_rowmajor_transposed(r12, n2), x22, t12, kNone);      // R12 X22
        dev.gemm_tn(as_rowmajor_transposed(x11, n1), t12, x12, minus_one);  // -X11 R12 X22
        dev.copy_block(x11, 0, 0, n1, n1, md, 0, 0);
        dev.copy_block(x12, 0, 0, n1, n2, md, 0, n1);
        dev.copy_block(x22, 0, 0, n2, n2, md, n1, n1);
        dev.template combine_need<T>(need + pass, need_ratio > 0.f ? nullptr : insp, need_blk, need_blk + 1);
      }
      if (inplace) {
        dev.apply_inplace(y, l, md);
      } else {
        dev.gemm_tn(as_rowmajor_transposed(y, l), md, tmp, kNone);
        std::swap(y.p, tmp.p);  // conditional passes come in pairs: a skipped pair swaps twice over untouched data
      }
      // (a re-seeded column needs a following pass to be orthonormalised: none after the last one; in-loop, the next
      // products with A and the next thin-Q take care of it)
      if (pass + 1 < npass || inloop) dev.refill_null(y, l, null_mask, (uint64_t)(0x9e3779b97f4a7c15ull ^ (uint64_t)(977 * (flags_used_ + pass) + l)));
      if (pass < always) ++tm.qr_passes;  // the conditional ones are counted when the flags are read (pending_clean)
    }
    dev.set_run_if(nullptr);
    Pending pd{always, npass, rough, 0};
    pd.st = st_scratch;
    pd.st_per_pass = single ? 1 : 2;
    // in-loop: only the single pass of a level-2 context is verified (so that the context escalates); with a conditional
    // chain enqueued the in-loop thin-Q is best effort -- it only has to keep the sketch well conditioned
    pd.flag_slot = (inloop && npass > 1) ? -1 : flags_used_;
    pending_.push_back(pd);
    flags_used_ += npass;
    subphase(tm.qr_gram_ms, qt0);
    return l;
  }

  int64_t orthonormalize_core(Skinny<T>& y, Skinny<T>& tmp, bool sharded, bool rough, bool polish = false) {
    const int64_t l = y.cols;
    int64_t r = l;
    // (sharded: the LOCAL row count says nothing -- and differs between ranks, which must all take the same branch;
    // a matrix with fewer global rows than l ends with need_next still set and repeats on the host-controlled path)
    if (defer_status_ && (sharded || y.rows >= l) && flags_used_ + kRobustPasses <= flags_cap_ &&
        dev.template device_qr_robust_fits<T>(l))
      return orthonormalize_device(y, tmp, sharded, rough, polish);
    if (dev.template device_chol_fits<T>(l)) {
      // Optimistic CholeskyQR2 entirely on the device: [Gram, Cholesky + inverse, apply] x2 are enqueued
      // back to back and the two status records are read once at the end.  Anything unusual (a failed
      // pivot, a second Gram that is not near I) falls through to the host-controlled robust loop below,
      // which simply continues from the current Y (every applied factor was non-singular, so the span is
      // unchanged; a failed pass applied the identity).
      PhaseTimer qt0;
      const int npass = rough ? 1 : 2;
      Skinny<T> gd0 = dev.template alloc_skinny<T>(l, l);
      Skinny<T> md0 = dev.template alloc_skinny<T>(l, l);
      const bool defer = defer_status_ && st_used_ + npass <= st_slots_;
      void* st_dev = defer ? (void*)((char*)st_pool_ + (size_t)st_used_ * kStatusBytes) : dev.alloc_bytes(2 * kStatusBytes);
      const double eps0 = (double)std::numeric_limits<T>::epsilon();
      for (int pass = 0; pass < npass; ++pass) {
        Skinny<T> yv = y.view_cols(l);
        dev.gemm_nn(as_rowmajor_transposed(y, l), yv, gd0, kNone);
        if (sharded) dev.allreduce(gd0.p, (size_t)gd0.ld * (size_t)gd0.cols_alloc);
        dev.chol_inv(gd0, l, (T)(4.0 * eps0), md0, st_dev, pass);
        dev.gemm_tn(as_rowmajor_transposed(y, l), md0, tmp, kNone);
        std::swap(y.p, tmp.p);
        ++tm.qr_passes;
      }
      if (defer) {
        // verified once at the end of random_svd_tall; a record that is not clean reruns the whole
        // computation through the host-controlled path below
        pending_.push_back({st_used_, npass, rough, 1});
        st_used_ += npass;
        subphase(tm.qr_gram_ms, qt0);
        return l;
      }
      int fail[2] = {0, 0};
      float min_ratio[2], dev_i[2];
      dev.read_chol_status(st_dev, npass, fail, min_ratio, dev_i);
      subphase(tm.qr_gram_ms, qt0);
      for (int pass = 0; pass < npass; ++pass)
        if (fail[pass] == 3) throw Error(ST_ENUMERIC, "non-finite Gram matrix in orthonormalisation");
      if (fail[0] == 2) return 0;  // Y is the zero matrix
      const bool ok = fail[0] == 0 && (rough || (fail[1] == 0 && dev_i[1] <= 0.25f));
      if (ok) return l;
    }
    if (!dev.template device_chol_fits<T>(l) && dev.template device_chol_blocked_fits<T>(l)) {
      // 176 < l <= 352: the same optimistic CholeskyQR2, with the factor-and-invert done 2 x 2 blocked on the device:
      //   G = [G11 G12; G12^T G22],  R11 = chol(G11), R12 = R11^-T G12, R22 = chol(G22 - R12^T R12),
      //   R^-1 = [X11, -X11 R12 X22; 0, X22]  with X = R^-1 of the diagonal blocks (chol_inv_kernel).
      PhaseTimer qt0;
      const int npass = rough ? 1 : 2;
      const int64_t n1 = round_up((l + 1) / 2, (int64_t)4), n2 = l - n1;
      Skinny<T> gd0 = dev.template alloc_skinny<T>(l, l);
      const bool defer = defer_status_ && st_used_ + 2 * npass <= st_slots_;
      void* st_dev = defer ? (void*)((char*)st_pool_ + (size_t)st_used_ * kStatusBytes) : dev.alloc_bytes(4 * kStatusBytes);
      const double eps0 = (double)std::numeric_limits<T>::epsilon();
      T* minus_one = dev.template alloc_scalar<T>(1);
      dev.fill_const(minus_one, (int64_t)1, (T)-1);
      for (int pass = 0; pass < npass; ++pass) {
        Skinny<T> yv = y.view_cols(l);
        dev.gemm_nn(as_rowmajor_transposed(y, l), yv, gd0, kNone);
        if (sharded) dev.allreduce(gd0.p, (size_t)gd0.ld * (size_t)gd0.cols_alloc);
        Skinny<T> g11 = dev.template alloc_skinny<T>(n1, n1), g12 = dev.template alloc_skinny<T>(n1, n2);
        Skinny<T> g22 = dev.template alloc_skinny<T>(n2, n2), x11 = dev.template alloc_skinny<T>(n1, n1);
        Skinny<T> x22 = dev.template alloc_skinny<T>(n2, n2), r12 = dev.template alloc_skinny<T>(n1, n2);
        Skinny<T> t22 = dev.template alloc_skinny<T>(n2, n2), t12 = dev.template alloc_skinny<T>(n1, n2);
        Skinny<T> x12 = dev.template alloc_skinny<T>(n1, n2), md0 = dev.template alloc_skinny<T>(l, l);
        dev.copy_block(gd0, 0, 0, n1, n1, g11, 0, 0);
        dev.copy_block(gd0, 0, n1, n1, n2, g12, 0, 0);
        dev.copy_block(gd0, n1, n1, n2, n2, g22, 0, 0);
        dev.chol_inv(g11, n1, (T)(4.0 * eps0), x11, st_dev, 2 * pass);
        dev.gemm_nn(as_rowmajor_transposed(x11, n1), g12, r12, kNone);  // R12 = X11^T G12
        dev.gemm_nn(as_rowmajor_transposed(r12, n2), r12, t22, kNone);  // R12^T R12
        dev.sub_inplace(g22, t22);                                      // Schur complement
        dev.chol_inv(g22, n2, (T)(4.0 * eps0), x22, st_dev, 2 * pass + 1);
        dev.gemm_tn(as_rowmajor_transposed(r12, n2), x22, t12, kNone);  // R12 X22
        dev.gemm_tn(as_rowmajor_transposed(x11, n1), t12, x12, minus_one);  // -X11 R12 X22
        dev.copy_block(x11, 0, 0, n1, n1, md0, 0, 0);
        dev.copy_block(x12, 0, 0, n1, n2, md0, 0, n1);
        dev.copy_block(x22, 0, 0, n2, n2, md0, n1, n1);
        dev.gemm_tn(as_rowmajor_transposed(y, l), md0, tmp, kNone);
        std::swap(y.p, tmp.p);
        ++tm.qr_passes;
      }
      if (defer) {
        pending_.push_back({st_used_, npass, rough, 2});
        st_used_ += 2 * npass;
        subphase(tm.qr_gram_ms, qt0);
        return l;
      }
      int fail[4] = {0, 0, 0, 0};
      float min_ratio[4], dev_i[4];
      dev.read_chol_status(st_dev, 2 * npass, fail, min_ratio, dev_i);
      subphase(tm.qr_gram_ms, qt0);
      bool ok = true;
      for (int i = 0; i < 2 * npass; ++i) {
        if (fail[i] == 3) throw Error(ST_ENUMERIC, "non-finite Gram matrix in orthonormalisation");
        ok = ok && fail[i] == 0;
      }
      if (ok && !rough) ok = dev_i[2 * (npass - 1)] <= 0.25f && dev_i[2 * (npass - 1) + 1] <= 0.25f;
      if (ok) return l;
    }
    // host-controlled from here on: inside an optimistic run its outcome (a rank below l, columns left zero) is not
    // covered by the deferred status records, so the whole call is repeated with the host in the loop
    if (defer_status_) optimistic_dirty_ = true;
    const double eps = (double)std::numeric_limits<T>::epsilon();
    std::vector<double> g((size_t)l * l), mm((size_t)l * l), uu, ss, vv;
    Skinny<T> gd = dev.template alloc_skinny<T>(l, l);
    Skinny<T> md = dev.template alloc_skinny<T>(l, l);
    int fails = 0;
    // After a clean Cholesky pass whose pivots predict ||Y1^T Y1 - I|| = E small, the polishing pass needs
    // no host: (I + E)^(-1/2) = I - E/2 + 3E^2/8 - 5E^3/16 + O(E^4) is formed on the device.
    bool series_next = false;
    const double series_max_e = sizeof(T) == 4 ? 1e-2 : 1e-4;  // O(E^4) stays below eps
    for (int pass = 0; pass < 12; ++pass) {
      if (r == 0) break;
      // G (r x r) = Y[:, :r]^T Y[:, :r]
      Skinny<T> yv = y.view_cols(r);
      Skinny<T> gv = gd.view_cols(r);
      gv.rows = r;
      PhaseTimer qt;
      dev.gemm_nn(as_rowmajor_transposed(y, r), yv, gv, kNone);
      if (sharded) dev.allreduce(gd.p, (size_t)gd.ld * (size_t)gd.cols_alloc);
      subphase(tm.qr_gram_ms, qt);
      ++tm.qr_passes;
      if (series_next) {
        Skinny<T> mv = md.view_cols(r);
        mv.rows = r;
        dev.inv_sqrt_series(gv, r, mv);
        Skinny<T> out = tmp.view_cols(r);
        dev.gemm_tn(as_rowmajor_transposed(y, r), mv, out, kNone);
        if (r < l) dev.zero_cols(tmp, r, l);
        std::swap(y.p, tmp.p);
        subphase(tm.qr_apply_ms, qt);
        break;
      }
      dev.download_skinny(gv, r, r, g.data());
      double gmax = 0.0, dev_i = 0.0;
      for (int64_t j = 0; j < r; ++j)
        for (int64_t i = 0; i < r; ++i) {
          const double x = g[(size_t)j * r + i];
          if (!std::isfinite(x)) throw Error(ST_ENUMERIC, "non-finite Gram matrix in orthonormalisation");
          dev_i = std::max(dev_i, std::fabs(x - (i == j ? 1.0 : 0.0)));
          if (i == j) gmax = std::max(gmax, x);
        }
      if (gmax == 0.0) {
        r = 0;
        break;
      }
      subphase(tm.qr_down_ms, qt);
      if (dev_i <= 16.0 * eps) break;  // already orthonormal at working precision
      const bool near_i = dev_i <= 0.25;
      int64_t r_new = r;
      bool clean = false;
      double min_ratio_last = 0.0;
      if (fails >= 3) {
        // exact-null directions: drop them.  G = V diag(lam) V^T (Jacobi on the symmetric G).
        uu.resize((size_t)r * r);
        vv.resize((size_t)r * r);
        ss.resize(r);
        if (small::jacobi_svd((int)r, g.data(), (int)r, uu.data(), ss.data(), vv.data(), 1e-15) < 0)
          throw Error(ST_ENUMERIC, "non-finite Gram matrix in orthonormalisation");
        r_new = 0;
        while (r_new < r && ss[r_new] > 1e-2 * ss[0]) ++r_new;
        std::fill(mm.begin(), mm.end(), 0.0);
        for (int64_t j = 0; j < r_new; ++j) {
          const double sc = 1.0 / std::sqrt(ss[j]);
          for (int64_t i = 0; i < r; ++i) mm[(size_t)j * r + i] = vv[(size_t)j * r + i] * sc;
        }
        fails = 0;
      } else {
        std::vector<double> rr(g.begin(), g.begin() + (size_t)r * r);
        double min_ratio = 0.0;
        clean = small::chol_upper((int)r, rr.data(), (int)r, 4.0 * eps, &min_ratio);
        min_ratio_last = min_ratio;
        if (!clean) {
          ++fails;
          double shift = 16.0 * eps * gmax;
          bool ok = false;
          for (int tries = 0; tries < 8 && !ok; ++tries, shift *= 10.0) {
            rr.assign(g.begin(), g.begin() + (size_t)r * r);
            for (int64_t i = 0; i < r; ++i) rr[(size_t)i * r + i] += shift;
            ok = small::chol_upper((int)r, rr.data(), (int)r, 0.0, &min_ratio);
          }
          if (!ok) throw Error(ST_ENUMERIC, "Gram matrix not positive definite even after shifting");
        } else {
          fails = 0;
        }
        small::triu_inverse((int)r, rr.data(), (int)r);
        std::copy(rr.begin(), rr.end(), mm.begin());
      }
      subphase(tm.qr_host_ms, qt);
      // Y[:, :r_new] <- Y[:, :r] * M (r x r_new); columns >= r_new become zero.
      Skinny<T> mv = md.view_cols(r_new);
      mv.rows = r;
      dev.upload_skinny(mm.data(), r, r_new, r, md);
      Skinny<T> out = tmp.view_cols(r_new);
      dev.gemm_tn(as_rowmajor_transposed(y, r), mv, out, kNone);
      if (r_new < l) dev.zero_cols(tmp, r_new, l);
      std::swap(y.p, tmp.p);
      r = r_new;
      subphase(tm.qr_apply_ms, qt);
      if (clean && (near_i || rough)) break;
      // predicted orthogonality defect of the pass just applied: ~ l * eps * kappa^2 <= l * eps / min pivot ratio
      if (clean && fails == 0 && r == l && 4.0 * (double)l * eps / std::max(min_ratio_last, 1e-300) <= series_max_e)
        series_next = true;
    }
    return r;
  }

  // ---- power_iter, random_svd.rs:15-59 -------------------------------------------------
  // Leaves the orthonormal basis in `y` (mt x l) and returns its numerical rank.
  int64_t power_iter(const TallA<T>& a, int64_t l, int64_t n_iter, const RunOpts& o, Skinny<T>& y, Skinny<T>& y2) {
    qr_householder = o.qr_householder;
    m_local_ = m_local_override_ >= 0 ? m_local_override_ : a.mt;
    struct MixedScope {  // the bf16-split products serve the range finder only (random_svd.rs:31, 42-51)
      int& slot;
      int saved;
      MixedScope(int& s, int v) : slot(s), saved(s) { slot = v; }
      ~MixedScope() { slot = saved; }
    } mixed_scope(mixed_planes_, o.mixed_planes);
    PhaseTimer pt;
    Skinny<T> om = dev.template alloc_skinny<T>(a.nt, l);
    if (o.omega) {
      if (o.omega_on_device)
        dev.copy_in_skinny((const T*)o.omega, o.omega_ld, om);
      else
        dev.upload_skinny_native((const T*)o.omega, o.omega_ld, om);
    } else {
      // random_mat_normal(a_ncols, omega_rank), random_svd.rs:24 / mat_utils.rs:161-175
      dev.fill_normal(om.p, a.nt, l, 1, om.ld, o.seed, 0, l);
    }
    Skinny<T> z = dev.template alloc_skinny<T>(a.nt, l);
    double* ss_dev = dev.alloc_f64(1);
    T* inv_dev = dev.template alloc_scalar<T>(1);
    // One-sweep schedule: Y = A X followed by Z = A^T Y is the single product Z = A^T (A X) whenever
    // Y itself is not needed, i.e. for the sketch and the iterations without the in-loop thin-Q (i <= 2, :37-39):
    //   Z(0) = A^T (A Omega), Z(i) = A^T (A Z^(i-1)), i < nf = min(q, 3);  then Y = A Z^(nf-1) and the loop continues
    // with the reference's own steps.  A is read nf + 1 times instead of 2 nf + 1 and no m x l matrix is written.
    int64_t i0 = 0;
    const bool fused = o.power_fused && n_iter > 0 && a.row_major && !a.mu_short && !a.mu_tall &&
                       dev.template ata_fused_fits<T>(a.mem, l);
    if (fused) {
      const int64_t nf = std::min<int64_t>(n_iter, 3);
      for (int64_t i = 0; i < nf; ++i) {
        if (i == 0) dev.event_mark(0);
        dev.ata_fused(a.mem, i == 0 ? om : z, z);
        if (i == 0) dev.event_mark(1);
        if (o.sharded) dev.allreduce(z.p, (size_t)z.ld * (size_t)z.cols);
        dev.inv_norm(z, ss_dev, inv_dev);
        dev.scale_inplace(z, inv_dev);
        if (i == 0) phase(tm.sketch_ms, pt);
      }
      a_times(a, z, y, kNone);
      i0 = nf;
    } else {
      dev.event_mark(0);
      a_times(a, om, y, kNone);  // :31
      dev.event_mark(1);
      phase(tm.sketch_ms, pt);
    }
    for (int64_t i = i0; i < n_iter; ++i) {  // :35
      if (i > 2) {                          // :37-39
        phase(tm.power_ms, pt);
        orthonormalize(y, y2, o.sharded, /*rough=*/true);
        phase(tm.qr_ms, pt);
      }
      at_times(a, y, z, kNone, o.sharded);  // :42-46
      // :53-55 rescales Y by 1 / ||Y||_F after every iteration so that nothing overflows; the span -- all the final
      // thin-Q keeps -- does not depend on WHICH positive factor is used.  Here the factor is 1 / ||Z||_F applied to
      // Z (n x l, a few hundred KB, replicated on every rank) BEFORE Y = A Z: Y comes out at ||A Z^|| <= sigma_1
      // instead of sigma_1^3 (a wider safe range than the reference's own), no pass over the m-sized Y is spent on
      // the norm or the scaling (two sweeps of 3.2 GB each per iteration at 10^7 x 80), and sharded runs need no
      // scalar all-reduce (Z is already all-reduced).
      dev.inv_norm(z, ss_dev, inv_dev);
      dev.scale_inplace(z, inv_dev);
      a_times(a, z, y, kNone);              // :47-51
    }
    phase(tm.power_ms, pt);
    int64_t r = orthonormalize(y, y2, o.sharded);  // :57
    phase(tm.qr_ms, pt);
    return r;
  }

  // ---- random_svd, random_svd.rs:63-110, on the already-tall view ------------------------
  // u_tall: mt x k, v_tall: nt x k (both skinny, allocated by the caller), s_dev: k values (device).
  // emit (optional): enqueues the caller's output copies.  It runs BEFORE the status records of the optimistic run are
  // read, so the copies follow the last kernel without a host round trip in between; if the run has to be repeated
  // it runs again and overwrites them.
  void random_svd_tall(const TallA<T>& a, int64_t k, int64_t l, int64_t n_iter, const RunOpts& o, Skinny<T>& u_tall,
                       T* s_dev, Skinny<T>& v_tall, const std::function<void()>& emit = {}) {
    qr_householder = o.qr_householder;
    m_local_ = m_local_override_ >= 0 ? m_local_override_ : a.mt;
    m_global_ = -1;
    hh_short_shards_ = -1;
    // Householder mode has no status records to defer: the body runs with the host in the loop, which also lets it
    // complete the null vectors of an exactly singular core (see random_svd_tall_body)
    const bool hh = qr_householder;
    if (!hh && (dev.template device_chol_fits<T>(l) || dev.template device_chol_blocked_fits<T>(l))) {
      // Optimistic run: every Cholesky-QR status record is checked once, after the last kernel is enqueued
      // (no host synchronisation inside the call).  A record that is not clean (rank deficiency, zero or
      // non-finite input, ...) repeats the computation with the host in the loop.
      const Timings saved = tm;
      const auto mark = dev.arena_mark();  // a repeated attempt reuses the workspace of the abandoned one
      for (int attempt = 0; attempt < 5; ++attempt) {
        if (attempt > 0) dev.arena_rewind(mark);
        // records: <= 2 per pass x <= 2 passes for each of the max(0, q - 3) in-loop, the final and the B^T thin-Q
        st_slots_ = (int)std::min<int64_t>(4 * (std::max<int64_t>(0, n_iter - 3) + 2) + 1, 4096);  // + the core SVD's
        // status records and verdict words share one zeroed allocation: ONE device-to-host copy reads both at the end
        flags_cap_ = kRobustPasses * (int)std::min<int64_t>(std::max<int64_t>(0, n_iter - 3) + 4, 1024);
        st_pool_ = dev.alloc_zeroed_bytes((size_t)st_slots_ * kStatusBytes + (size_t)flags_cap_ * sizeof(int));
        st_used_ = 0;
        flags_pool_ = (int*)((char*)st_pool_ + (size_t)st_slots_ * kStatusBytes);
        flags_used_ = 0;
        pending_.clear();
        optimistic_dirty_ = false;
        robust_needs_more_ = false;
        svd_needs_more_ = false;
        svd_needs_v_ = false;
        null_cols_seen_ = false;
        defer_status_ = true;
        try {
          random_svd_tall_body(a, k, l, n_iter, o, u_tall, s_dev, v_tall);
        } catch (...) {
          defer_status_ = false;
          throw;
        }
        defer_status_ = false;
        if (emit) emit();
        if (pending_clean()) return;
        tm = saved;
        dev.phase_forget();  // the abandoned run still counts in total_ms, not in the phase slots
        // a thin-Q that only ran out of enqueued passes: enqueue more from now on (this context) and repeat on the device
        if (optimistic_dirty_) break;
        if (robust_needs_more_ && dev.robust_passes() < kRobustPasses) {
          dev.set_robust_passes(std::min(kRobustPasses, 2 * std::max(2, dev.robust_passes())));
        } else if (svd_needs_more_ && dev.svd_more_sweeps()) {
          // the context enqueues more Jacobi sweeps from now on; repeat on the device
        } else if (svd_needs_v_ && dev.svd_force_v()) {
          // the context accumulates V in the sweeps from now on; repeat on the device
        } else {
          break;
        }
      }
      dev.arena_rewind(mark);
    }
    random_svd_tall_body(a, k, l, n_iter, o, u_tall, s_dev, v_tall);
    if (emit) emit();
  }

  bool pending_clean() {
    if (optimistic_dirty_) return false;
    if (pending_.empty()) return true;
    std::vector<int> fail((size_t)st_used_);
    std::vector<float> min_ratio((size_t)st_used_), dev_i((size_t)st_used_);
    std::vector<int> flags((size_t)flags_used_);
    {
      // one copy (and one synchronisation) for the records and the words: [0, st_used_ records) ... [flags)
      struct Rec {
        int fail;
        float min_ratio, dev_i, gmax;
        long long clk, wall;
      };
      static_assert(sizeof(Rec) == kStatusBytes, "status record layout");
      const size_t bytes = (size_t)st_slots_ * kStatusBytes + (size_t)flags_used_ * sizeof(int);
      std::vector<char> host(bytes);
      dev.read_bytes(st_pool_, bytes, host.data());
      for (int i = 0; i < st_used_; ++i) {
        Rec r;
        std::memcpy(&r, host.data() + (size_t)i * kStatusBytes, sizeof(r));
        fail[(size_t)i] = r.fail;
        min_ratio[(size_t)i] = r.min_ratio;
        dev_i[(size_t)i] = r.dev_i;
      }
      if (flags_used_ > 0) std::memcpy(flags.data(), host.data() + (size_t)st_slots_ * kStatusBytes, (size_t)flags_used_ * sizeof(int));
    }
    // a non-finite Gram matrix or core is an error of the INPUT: nothing to escalate, nothing to repeat
    for (int i = 0; i < flags_used_; ++i) {
      if (flags[(size_t)i] & kFlagNonFinite) throw Error(ST_ENUMERIC, "non-finite Gram matrix in orthonormalisation");
      if (flags[(size_t)i] & kFlagNullCols) null_cols_seen_ = true;
    }
    for (const Pending& p : pending_)
      if (p.flag_slot < 0 && p.per_pass > 0)
        for (int i = 0; i < (p.is_svd ? 1 : p.npass * p.per_pass); ++i)
          if (fail[(size_t)p.slot + i] == 3)
            throw Error(ST_ENUMERIC, p.is_svd ? "non-finite core matrix in small SVD" : "non-finite Gram matrix in orthonormalisation");
    for (const Pending& p : pending_) {
      if (p.flag_slot >= 0) {  // device-robust thin-Q: its last enqueued pass must not ask for another one
        // (p.slot = number of unconditional passes; conditional pass i ran iff pass i - 1 asked for it)
        for (int i = p.slot; i < p.npass; ++i)
          if (flags[(size_t)p.flag_slot + i - 1] != 0) ++tm.qr_passes;
        if (std::getenv("CORRLA_DEBUG")) {
          std::fprintf(stderr, "[corrla] device thin-Q: %d passes enqueued, need_next =", p.npass);
          for (int i = 0; i < p.npass; ++i) std::fprintf(stderr, " %d", flags[(size_t)p.flag_slot + i]);
          const int nrec = p.npass * p.st_per_pass;
          std::vector<int> f2((size_t)nrec);
          std::vector<float> mr((size_t)nrec), di((size_t)nrec);
          dev.read_chol_status(p.st, nrec, f2.data(), mr.data(), di.data());
          std::fprintf(stderr, "; ||G - I||_max / min pivot ratio per factorisation:");
          for (int i = 0; i < nrec; ++i) std::fprintf(stderr, " %.2g/%.2g", di[(size_t)i], mr[(size_t)i]);
          std::fprintf(stderr, "\n");
        }
        if (flags[(size_t)p.flag_slot + p.npass - 1] != 0) {
          robust_needs_more_ = true;
          return false;
        }
        continue;
      }
      if (p.rough && p.per_pass == 0) continue;  // one shifted pass in-loop: nothing to verify
      if (std::getenv("CORRLA_DEBUG") && p.flag_slot < 0 && p.per_pass > 0)
        std::fprintf(stderr, "[corrla] status record %d (%s): fail %d min_ratio %.3g dev_i %.3g\n", p.slot, p.is_svd ? "core SVD" : "Cholesky",
                     fail[(size_t)p.slot], min_ratio[(size_t)p.slot], dev_i[(size_t)p.slot]);
      if (p.is_svd) {
        if (fail[(size_t)p.slot] == 0) dev.svd_sweeps_used((int)dev_i[(size_t)p.slot]);
        if (fail[(size_t)p.slot] == 1) {  // not converged within the sweeps that were enqueued
          svd_needs_more_ = true;
          return false;
        }
        if (fail[(size_t)p.slot] == 4) {  // the W-only shortcut was not valid for this core: accumulate V from now on
          svd_needs_v_ = true;
          return false;
        }
      }
      for (int i = 0; i < p.npass * p.per_pass; ++i)
        if (fail[p.slot + i] != 0) return false;
      if (!p.rough)
        for (int i = 0; i < p.per_pass; ++i)
          if (!(dev_i[p.slot + (p.npass - 1) * p.per_pass + i] <= 0.25f)) return false;
    }
    return true;
  }

  void random_svd_tall_body(const TallA<T>& a, int64_t k, int64_t l, int64_t n_iter, const RunOpts& o, Skinny<T>& u_tall,
                            T* s_dev, Skinny<T>& v_tall) {
    PhaseTimer total;
    // both are written by a product before anything reads them (q: the sketch; q2: the first Y * R^-1)
    const bool hh_tmp = qr_householder;  // the Householder path stores reflectors in q2: keep it zero-filled
    Skinny<T> q = dev.template alloc_skinny_out<T>(a.mt, l);
    Skinny<T> q2 = hh_tmp ? dev.template alloc_skinny<T>(a.mt, l) : dev.template alloc_skinny_out<T>(a.mt, l);
    power_iter(a, l, n_iter, o, q, q2);  // :76-77
    PhaseTimer pt;
    // B^T = A^T Q  (n x l)                                                           :80
    Skinny<T> bt = dev.template alloc_skinny<T>(a.nt, l);
    mixed_planes_ = o.mixed_project ? o.mixed_planes : 0;
    at_times(a, q, bt, kNone, o.sharded);
    mixed_planes_ = 0;
    phase(tm.project_ms, pt);
    // SVD of B (l x n), :89.  The reference takes faer's full SVD and slices; here:
    // B^T = Qb C with Qb orthonormal (n x l) and C = Qb^T B^T (l x l); C = Uc S Vc^T on the host;
    // => B = Vc S (Qb Uc)^T, i.e. U~ = Vc and V = Qb Uc.
    Skinny<T> qb = dev.template alloc_skinny<T>(a.nt, l);
    Skinny<T> qb2 = dev.template alloc_skinny<T>(a.nt, l);
    dev.copy_skinny(bt, qb);
    orthonormalize(qb, qb2, false);
    phase(tm.qr_ms, pt);
    // The core is formed TRANSPOSED, C^T = B Qb = (Qb^T B^T)^T: C is the triangular factor R of B^T = Qb R (up to
    // rounding), and one-sided Jacobi on the columns of R^T converges in far fewer sweeps than on the columns of R
    // when the spectrum decays (l = 138, sigma_i = 0.9^i: 9 sweeps against 22; flat spectra: the same 7) -- the
    // Drmac-Veselic preconditioning, free here because the QR has just been done.
    // C^T = Vc S Uc^T, so the roles of the two factors swap: small_svd(X) returns (V_X -> first, U_X -> second).
    Skinny<T> ct = dev.template alloc_skinny<T>(l, l);
    dev.gemm_nn(as_rowmajor_transposed(bt, l), qb, ct, kNone);
    if (o.poison_core) dev.poison_entry(ct, l / 2, l / 3, o.poison_core);  // test hook, see RunOpts
    Skinny<T> m1 = dev.template alloc_skinny<T>(l, k);  // U~[:, :k] = Vc[:, :k]
    Skinny<T> m2 = dev.template alloc_skinny<T>(l, k);  // Uc[:, :k]
    void* svd_st = nullptr;
    if (defer_status_ && st_used_ + 1 <= st_slots_) {
      // kernels that run a fixed number of sweeps report convergence here; checked with the Cholesky records
      svd_st = (void*)((char*)st_pool_ + (size_t)st_used_ * kStatusBytes);
      pending_.push_back({st_used_, 1, true, 1});
      pending_.back().is_svd = true;
      st_used_ += 1;
    }
    dev.small_svd(ct, l, k, m2, m1, s_dev, svd_st);
    int64_t nz_known = k;
    if (defer_status_) {
      // One-sided Jacobi leaves the W / sigma factor (here m1 = U~) orthonormal only to the size of the last
      // rotations it skipped, which for clustered singular values is far above eps (6e-5 in f32 for the top 74 values
      // of a 1.25e6 x 512 Gaussian matrix).  One Cholesky-QR pass restores it to working precision: R is I + O(1e-4), so the
      // triplets move by less than their own uncertainty.  (The accumulated-rotation factor m2 is orthogonal by
      // construction.)  A rank-deficient core fails the pass and the call repeats on the host-controlled path.
      Skinny<T> m1b = dev.template alloc_skinny<T>(l, k);
      orthonormalize_core(m1, m1b, false, /*rough=*/true, /*polish=*/true);
    }
    if (!defer_status_) {
      // Exactly singular core (rank-deficient or zero input; only reachable through the host-controlled path): the
      // vectors w_j / sigma_j of its null triplets (here: columns of Vc, the Jacobi runs on C^T) do not exist.  Give
      // them an orthonormal completion, like the arbitrary-but-orthonormal null vectors of the reference's full SVD.
      std::vector<T> sh((size_t)k);
      dev.copy_values_out(s_dev, k, sh.data(), /*dst_is_host=*/true);
      // numerically null: below eps * 1e-3 of the largest singular value nothing of the direction survives the
      // products that formed the core
      const T null_tol = (T)(1e-3 * (double)std::numeric_limits<T>::epsilon()) * sh[0];
      int64_t nz = 0;
      while (nz < k && sh[(size_t)nz] > null_tol) ++nz;
      if (nz < k) complete_basis(m1, nz, false);
      nz_known = nz;
      if (nz == k) {
        Skinny<T> m1b = dev.template alloc_skinny<T>(l, k);
        orthonormalize_core(m1, m1b, false, /*rough=*/true);
      }
    }
    (void)nz_known;
    phase(tm.small_svd_ms, pt);
    // V = Qb * Uc[:, :k]
    dev.gemm_tn(as_rowmajor_transposed(qb, l), m2, v_tall, kNone);
    // sign convention (the reference fixes none): largest-magnitude component of every v_i positive.  The signs only
    // depend on V (short side), so they are folded into the l x k factor U~ BEFORE the tall product: U needs no
    // sign pass and can be written straight into the caller's buffer.
    dev.fix_signs(v_tall, m1, k);
    // U = Q * U~[:, :k]                                                               :92, :96-109
    dev.gemm_tn(as_rowmajor_transposed(q, l), m1, u_tall, kNone);
    phase(tm.finalize_ms, pt);
    tm.host_enqueue_ms += total.lap();
  }

  // everything enqueued since the previous mark belongs to `slot` (device time, resolved after the call)
  void phase(double& slot, PhaseTimer& pt) {
    if (profile_phases) dev.sync();
    (void)pt.lap();
    dev.phase_mark(&slot);
  }
  // nested breakdown of the orthonormalisation (debugging aid): host wall clock, only with synchronised phases
  void subphase(double& slot, PhaseTimer& pt) {
    if (!profile_phases) return;
    dev.sync();
    slot += pt.lap();
  }
};

// Host SVD of the l x l core (download, one-sided Jacobi in f64, upload the sorted factors): the
// backend-independent fallback behind Dev::small_svd.  C = Uc S Vc^T;  m1 <- Vc[:, :k], m2 <- Uc[:, :k].
template <class Dev, class T>
inline void small_svd_host(Dev& dev, const Skinny<T>& cd, int64_t l, int64_t k, Skinny<T>& m1, Skinny<T>& m2, T* s_dev) {
  std::vector<double> c((size_t)l * l), uc((size_t)l * l), vc((size_t)l * l), sv(l);
  dev.download_skinny(cd, l, l, c.data());
  const double eps = (double)std::numeric_limits<T>::epsilon();
  if (small::jacobi_svd((int)l, c.data(), (int)l, uc.data(), sv.data(), vc.data(), std::max(1e-15, eps * 1e-3)) < 0)
    throw Error(ST_ENUMERIC, "non-finite core matrix in small SVD");
  dev.upload_skinny(vc.data(), l, k, l, m1);
  dev.upload_skinny(uc.data(), l, k, l, m2);
  std::vector<T> st(k);
  for (int64_t i = 0; i < k; ++i) st[i] = (T)sv[i];
  dev.store_values(st.data(), k, s_dev, /*dst_is_host=*/false);
}

// ---- argument handling shared by the C ABI of the product and of the test emulation --------
struct Layout {
  bool fat;        // m < n (strict, random_svd.rs:71): work on A^T
  bool row_major;  // the TALL view is row-major in memory
  bool needs_pack; // neither stride is 1, or vector alignment not met: repack first
  int64_t mt, nt, ld;
};

inline void validate_matrix(const void* a, int64_t m, int64_t n, int64_t rs, int64_t cs) {
  if (!a) throw Error(ST_EINVAL, "a is NULL");
  if (m < 1 || n < 1) throw Error(ST_EINVAL, "matrix must have at least one row and one column");
  if (rs < 0 || cs < 0) throw Error(ST_EINVAL, "negative strides are not supported");
  if (m > 1 && rs == 0) throw Error(ST_EINVAL, "row_stride == 0 with more than one row");
  if (n > 1 && cs == 0) throw Error(ST_EINVAL, "col_stride == 0 with more than one column");
}

// Classify a (m, n, rs, cs) strided matrix into the tall row-/column-major views the kernels take.
inline Layout classify(int64_t m, int64_t n, int64_t rs, int64_t cs) {
  Layout L;
  L.fat = m < n;
  L.mt = L.fat ? n : m;
  L.nt = L.fat ? m : n;
  // strides of the tall view
  const int64_t trs = L.fat ? cs : rs, tcs = L.fat ? rs : cs;
  L.needs_pack = false;
  if ((tcs == 1 || L.nt == 1) && (trs >= L.nt || L.mt == 1)) {
    L.row_major = true;
    L.ld = L.mt == 1 ? L.nt : trs;
  } else if ((trs == 1 || L.mt == 1) && (tcs >= L.mt || L.nt == 1)) {
    L.row_major = false;
    L.ld = L.nt == 1 ? L.mt : tcs;
  } else {
    L.row_major = true;
    L.needs_pack = true;
    L.ld = L.nt;
  }
  return L;
}

inline void validate_rank(int64_t m, int64_t n, int64_t rank, int64_t n_iter, int64_t n_oversamples) {
  if (rank < 1) throw Error(ST_EINVAL, "rank must be >= 1");
  if (n_iter < 0 || n_oversamples < 0) throw Error(ST_EINVAL, "n_iter and n_oversamples must be >= 0");
  // random_svd.rs:98-107: slicing 0..omega_rank past l = min(rank+p, min(m,n)) panics in the reference
  if (rank > std::min(m, n)) throw Error(ST_EINVAL, "rank exceeds min(m, n) (the reference panics here)");
  if (rank + n_oversamples > (int64_t)1 << 20) throw Error(ST_EINVAL, "rank + n_oversamples too large");
}

}  // namespace corrla
