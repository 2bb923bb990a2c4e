// C-ABI glue shared by the product library (HIP backend) and the test-only emulation library:
// argument validation, fat/tall + stride classification, staging of A, output orientation
// (random_svd.rs:69-74, 96-109).  Templated on the backend; contains no m-/n-sized arithmetic.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/corrla_rsvd.h"
#include "driver.hpp"

namespace corrla {

inline std::string& last_error_slot() {
  static thread_local std::string msg;
  return msg;
}

template <class F>
inline corrla_status guarded(F&& f) {
  try {
    f();
    return CORRLA_OK;
  } catch (const Error& e) {
    last_error_slot() = e.what();
    return (corrla_status)e.code;
  } catch (const std::bad_alloc&) {
    last_error_slot() = "host allocation failed";
    return CORRLA_ENOMEM;
  } catch (const std::exception& e) {
    last_error_slot() = e.what();
    return CORRLA_EINVAL;
  }
}

inline RunOpts parse_opts(const corrla_opts* o, bool dev_ptrs) {
  RunOpts r;
  const char* qr_env = std::getenv("CORRLA_QR");
  r.qr_householder = qr_env && std::strcmp(qr_env, "householder") == 0;
  const char* fu_env = std::getenv("CORRLA_POWER_FUSED");
  r.power_fused = fu_env && std::atoi(fu_env) != 0;
  if (const char* pz = std::getenv("CORRLA_TEST_POISON_CORE")) r.poison_core = std::atoi(pz);  // test hook, see RunOpts
  if (const char* mx = std::getenv("CORRLA_SKETCH_MIXED"))
    r.mixed_planes = std::strcmp(mx, "bf16x3") == 0 ? 2 : (std::strcmp(mx, "bf16x6") == 0 ? 3 : 0);
  if (const char* mp = std::getenv("CORRLA_MIXED_PROJECT")) r.mixed_project = std::atoi(mp) != 0;
  if (!o) return r;
  if (o->struct_size != sizeof(corrla_opts)) throw Error(ST_EINVAL, "corrla_opts.struct_size mismatch");
  // seed: used as given when it is non-zero or CORRLA_SEED_EXPLICIT is set (so 0 is a usable seed); otherwise every
  // call draws a fresh sketch like the reference's unseeded thread_rng (mat_utils.rs:161-175) -- see fresh_seed()
  r.seed_explicit = o->seed != 0 || (o->flags & CORRLA_SEED_EXPLICIT) != 0;
  if (r.seed_explicit) r.seed = o->seed;
  r.omega = o->omega;
  r.omega_ld = o->omega_ld;
  r.omega_on_device = (o->flags & CORRLA_OMEGA_ON_DEVICE) != 0;
  if (r.omega_on_device && !dev_ptrs) throw Error(ST_EINVAL, "CORRLA_OMEGA_ON_DEVICE is only valid for *_dev entry points");
  if ((o->flags & CORRLA_PCA_CENTER_FUSED) && (o->flags & CORRLA_PCA_CENTER_COPY))
    throw Error(ST_EINVAL, "CORRLA_PCA_CENTER_FUSED and CORRLA_PCA_CENTER_COPY are mutually exclusive");
  r.pca_center = (o->flags & CORRLA_PCA_CENTER_FUSED) ? 1 : ((o->flags & CORRLA_PCA_CENTER_COPY) ? 2 : 0);
  r.qr_householder = r.qr_householder || (o->flags & CORRLA_QR_HOUSEHOLDER) != 0;
  r.power_fused = r.power_fused || (o->flags & CORRLA_POWER_FUSED) != 0;
  if ((o->flags & CORRLA_SKETCH_BF16X3) && (o->flags & CORRLA_SKETCH_BF16X6))
    throw Error(ST_EINVAL, "CORRLA_SKETCH_BF16X3 and CORRLA_SKETCH_BF16X6 are mutually exclusive");
  if (o->flags & CORRLA_SKETCH_BF16X3) r.mixed_planes = 2;
  if (o->flags & CORRLA_SKETCH_BF16X6) r.mixed_planes = 3;
  return r;
}

// Bring the strided input into one of the two layouts the kernels take and describe it as the
// TALL matrix.  Host pointers are always staged (H2D) into a padded row-major device buffer;
// device pointers are used in place when 16-byte vector loads are legal, else repacked.
template <class Dev, class T>
inline TallA<T> stage_input(Dev& dev, bool host_ptrs, const T* a, int64_t m, int64_t n, int64_t rs, int64_t cs,
                            bool force_tall) {
  validate_matrix(a, m, n, rs, cs);
  Layout L = classify(m, n, rs, cs);
  if (force_tall && L.fat) {
    // sharded / power_iter: never transpose (random_svd.rs:15-59 takes the matrix as given)
    L = classify(m, n, rs, cs);
    L.fat = false;
    L.mt = m;
    L.nt = n;
    if ((cs == 1 || n == 1) && (rs >= n || m == 1)) {
      L.row_major = true;
      L.needs_pack = false;
      L.ld = m == 1 ? n : rs;
    } else if ((rs == 1 || m == 1) && (cs >= m || n == 1)) {
      L.row_major = false;
      L.needs_pack = false;
      L.ld = n == 1 ? m : cs;
    } else {
      L.row_major = true;
      L.needs_pack = true;
      L.ld = n;
    }
  }
  constexpr int64_t VEC = 16 / (int64_t)sizeof(T);
  TallA<T> ta;
  ta.mt = L.mt;
  ta.nt = L.nt;
  ta.row_major = L.row_major;
  const int64_t mem_rows = L.row_major ? L.mt : L.nt;
  const int64_t mem_cols = L.row_major ? L.nt : L.mt;
  // strides of the memory-row-major view in the ORIGINAL array
  int64_t vrs, vcs;
  {
    const int64_t trs = L.fat ? cs : rs, tcs = L.fat ? rs : cs;  // tall view strides
    vrs = L.row_major ? trs : tcs;
    vcs = L.row_major ? tcs : trs;
  }
  const bool aligned = !L.needs_pack && (((uintptr_t)a) % 16 == 0) && (L.ld % VEC == 0) && (mem_cols % VEC == 0);
  if (!host_ptrs && aligned) {
    ta.mem.p = a;
    ta.mem.rows = mem_rows;
    ta.mem.cols = mem_cols;
    ta.mem.ld = L.ld;
    ta.mem.cols_readable = mem_cols;
    return ta;
  }
  const int64_t ldp = round_up(mem_cols, kLdPad);
  T* buf = (T*)dev.alloc_bytes((size_t)mem_rows * (size_t)ldp * sizeof(T));
  dev.memset_zero(buf, (size_t)mem_rows * (size_t)ldp * sizeof(T));
  if (host_ptrs) {
    if (L.needs_pack || vcs != 1) {
      std::vector<T> packed((size_t)mem_rows * (size_t)mem_cols);
      for (int64_t r = 0; r < mem_rows; ++r)
        for (int64_t c = 0; c < mem_cols; ++c) packed[(size_t)r * mem_cols + c] = a[r * vrs + c * vcs];
      dev.h2d_2d(buf, ldp, packed.data(), mem_cols, mem_cols, mem_rows);
    } else {
      dev.h2d_2d(buf, ldp, a, mem_rows == 1 ? mem_cols : vrs, mem_cols, mem_rows);
    }
  } else {
    dev.pack_strided(a, mem_rows, mem_cols, vrs, vcs, buf, ldp);
  }
  ta.mem.p = buf;
  ta.mem.rows = mem_rows;
  ta.mem.cols = mem_cols;
  ta.mem.ld = ldp;
  ta.mem.cols_readable = ldp;
  return ta;
}

template <class Dev, class T>
inline void rsvd_entry(Dev& dev, bool host_ptrs, bool sharded, const T* a, int64_t m, int64_t n, int64_t rs, int64_t cs,
                       int64_t rank, int64_t n_iter, int64_t n_oversamples, const corrla_opts* opts, T* u, int64_t ldu,
                       T* s, T* vt, int64_t ldvt, Timings* tm_out, bool profile) {
  // Sharded calls: the LONG side of the global matrix is split over the ranks.  Default: row shards of a tall matrix
  // (local block m_local x n).  CORRLA_SHARD_COLS: column shards of a FAT matrix (local block m x n_local): the
  // tall view of random_svd.rs:69-74 is then A^T, whose row shard is this block transposed -- a stride swap, no copy.
  const bool shard_cols = sharded && opts && (opts->flags & CORRLA_SHARD_COLS) != 0;
  // An EMPTY shard (m_local == 0 rows, or n_local == 0 columns with CORRLA_SHARD_COLS) is legal on the sharded entry
  // points -- more ranks than row blocks, ragged partitions: the rank takes part in every collective with zero
  // contributions (its block is replaced by ONE zero row of the tall view, which adds nothing to any sum) and writes
  // no row of the sharded output factor.
  const bool empty_shard = sharded && m >= 0 && n >= 0 && (shard_cols ? (n == 0 && m >= 1) : (m == 0 && n >= 1));
  const int64_t k = rank;
  RunOpts ro;
  TallA<T> ta;
  bool fat = false, u_in_place = false;
  int64_t l = 0;
  Skinny<T> ut, vtall;
  T* s_dev = nullptr;
  bool begun = false;
  // Everything that can fail on ONE rank only (arguments, staging, workspace) happens before the first collective;
  // on the sharded entry points its outcome is then agreed by one small all-reduce, so that a rank-local failure ends
  // the call on every rank instead of stranding the peers in a collective (Dev::sharded_handshake).
  auto prepare = [&] {
    // (an empty shard has no rows of its sharded output factor: that pointer may be NULL)
    if ((!u && !(empty_shard && !shard_cols)) || !s || (!vt && !(empty_shard && shard_cols)))
      throw Error(ST_EINVAL, "output pointer is NULL");
    if (!empty_shard) validate_matrix(a, m, n, rs, cs);
    if (!sharded && opts && (opts->flags & CORRLA_SHARD_COLS)) throw Error(ST_EINVAL, "CORRLA_SHARD_COLS is only valid for sharded entry points");
    const int64_t short_side = shard_cols ? m : n;
    if (sharded) {
      if (rank < 1 || rank > short_side) throw Error(ST_EINVAL, "rank must be in [1, short side] for the sharded path");
      if (n_iter < 0 || n_oversamples < 0) throw Error(ST_EINVAL, "n_iter and n_oversamples must be >= 0");
    } else {
      validate_rank(m, n, rank, n_iter, n_oversamples);
    }
    if (ldu < m) throw Error(ST_EINVAL, "ldu < m");
    if (ldvt < rank) throw Error(ST_EINVAL, "ldvt < rank");
    ro = parse_opts(opts, !host_ptrs);
    ro.sharded = sharded;
    if (!ro.seed_explicit && !ro.omega) ro.seed = dev.fresh_seed(/*rank_invariant=*/sharded);
    dev.begin_call();
    begun = true;
    if (empty_shard) {
      const int64_t nt = short_side, ldp = round_up(nt, kLdPad);
      T* zrow = (T*)dev.alloc_bytes((size_t)ldp * sizeof(T));
      dev.memset_zero(zrow, (size_t)ldp * sizeof(T));
      ta.mt = 1;
      ta.nt = nt;
      ta.row_major = true;
      ta.mem.p = zrow;
      ta.mem.rows = 1;
      ta.mem.cols = nt;
      ta.mem.ld = ldp;
      ta.mem.cols_readable = ldp;
    } else {
      ta = shard_cols ? stage_input<Dev, T>(dev, host_ptrs, a, n, m, cs, rs, true)
                      : stage_input<Dev, T>(dev, host_ptrs, a, m, n, rs, cs, sharded);
    }
    fat = sharded ? shard_cols : m < n;
    l = std::min<int64_t>(rank + n_oversamples, ta.nt);  // random_svd.rs:77
    if (ro.omega && ro.omega_ld < ta.nt) throw Error(ST_EINVAL, "omega_ld < min(m, n)");
    // Tall input, device pointers: the m x k factor U is produced directly in the caller's buffer (column-major, ldu)
    // -- no staging copy of the largest output.
    u_in_place = !fat && !host_ptrs && !empty_shard;
    if (u_in_place) {
      ut.p = u;
      ut.rows = ta.mt;
      ut.cols = k;
      ut.ld = ldu;
      ut.cols_alloc = k;
      ut.external = true;
    } else {
      ut = dev.template alloc_skinny<T>(ta.mt, k);
    }
    vtall = dev.template alloc_skinny<T>(ta.nt, k);
    s_dev = dev.template alloc_scalar<T>((int)k);
  };
  if (!sharded) {
    prepare();
  } else {
    if (dev.nranks() < 1) throw Error(ST_ECOMM, "communicator not initialised");
    int local = ST_OK;
    std::string local_msg;
    try {
      prepare();
    } catch (const Error& e) {
      local = e.code;
      local_msg = e.what();
    } catch (const std::bad_alloc&) {
      local = ST_ENOMEM;
      local_msg = "host allocation failed";
    }
    if (!begun) dev.begin_call();
    const int agreed = dev.sharded_handshake(local);
    if (local != ST_OK) throw Error(local, local_msg);
    if (agreed != ST_OK)
      throw Error(agreed, "sharded call abandoned: another rank failed before the first collective (status " + std::to_string(agreed) +
                              "); this rank's arguments were valid");
  }
  RsvdDriver<Dev, T> drv(dev, profile);
  if (empty_shard) drv.m_local_override_ = 0;  // the stand-in zero row is not a row of the matrix
  drv.random_svd_tall(ta, k, l, n_iter, ro, ut, s_dev, vtall, [&] {
    // random_svd.rs:96-109: tall -> (U, S, V^T); fat -> (V, S, U^T) of the transposed problem
    if (!fat) {
      if (!u_in_place && !empty_shard) dev.copy_out(ut, k, u, ldu, /*transpose=*/false, host_ptrs);
      dev.copy_out(vtall, k, vt, ldvt, /*transpose=*/true, host_ptrs);
    } else {
      dev.copy_out(vtall, k, u, ldu, false, host_ptrs);
      if (!empty_shard) dev.copy_out(ut, k, vt, ldvt, true, host_ptrs);
    }
    dev.copy_values_out(s_dev, k, s, host_ptrs);
  });
  PhaseTimer fin;
  drv.phase(drv.tm.finalize_ms, fin);  // output copies enqueued by emit()
  dev.phase_end();
  dev.end_call();
  dev.phase_resolve(&drv.tm.total_ms);
  drv.tm.n_collectives = dev.n_collectives;
  drv.tm.collective_bytes = dev.collective_bytes;
  drv.tm.sketch_kernel_ms = dev.event_elapsed_ms(0, 1);
  if (tm_out) *tm_out = drv.tm;
}

// PcaRsvd::new (pca_rsvd.rs:56-82): means, centring (implicit rank-1 corrections or a centred copy, see
// CORRLA_PCA_CENTER_*), random_svd of the centred matrix; keeps S and V^T.
// sharded: the SAMPLES (rows of x) are sharded over the ranks; means, S and the components come out replicated.  The
// column means and every product with the centred matrix are linear in the rows, so they are all-reduced partial sums.
template <class Dev, class T>
inline void pca_entry(Dev& dev, bool host_ptrs, const T* x, int64_t m, int64_t n, int64_t rs, int64_t cs, int64_t rank,
                      int64_t n_iter, int64_t n_oversamples, const corrla_opts* opts, T* means, T* s, T* comps,
                      int64_t ldc, Timings* tm_out, bool profile, bool sharded = false) {
  RunOpts ro;
  TallA<T> ta;
  bool begun = false;
  // what can fail on one rank only comes before the first collective and is agreed by the handshake (see rsvd_entry);
  // an empty shard is not supported here (the centring has no zero-contribution stand-in): it fails validation, on
  // every rank alike
  auto prepare = [&] {
    if (!means || !s || !comps) throw Error(ST_EINVAL, "output pointer is NULL");
    validate_matrix(x, m, n, rs, cs);
    if (sharded) {
      if (rank < 1 || rank > n) throw Error(ST_EINVAL, "rank must be in [1, n_dim] for the sample-sharded PCA");
      if (n_iter < 0 || n_oversamples < 0) throw Error(ST_EINVAL, "n_iter and n_oversamples must be >= 0");
    } else {
      validate_rank(m, n, rank, n_iter, n_oversamples);
    }
    if (ldc < rank) throw Error(ST_EINVAL, "ldc < rank");
    ro = parse_opts(opts, !host_ptrs);
    ro.sharded = sharded;
    if (!ro.seed_explicit && !ro.omega) ro.seed = dev.fresh_seed(sharded);
    dev.begin_call();
    begun = true;
    ta = stage_input<Dev, T>(dev, host_ptrs, x, m, n, rs, cs, sharded);
  };
  if (!sharded) {
    prepare();
  } else {
    if (dev.nranks() < 1) throw Error(ST_ECOMM, "communicator not initialised");
    int local = ST_OK;
    std::string local_msg;
    try {
      prepare();
    } catch (const Error& e) {
      local = e.code;
      local_msg = e.what();
    } catch (const std::bad_alloc&) {
      local = ST_ENOMEM;
      local_msg = "host allocation failed";
    }
    if (!begun) dev.begin_call();
    const int agreed = dev.sharded_handshake(local);
    if (local != ST_OK) throw Error(local, local_msg);
    if (agreed != ST_OK)
      throw Error(agreed, "sharded call abandoned: another rank failed before the first collective (status " + std::to_string(agreed) +
                              "); this rank's arguments were valid");
  }
  const int64_t m_global = sharded ? dev.allreduce_sum_host(m) : m;  // every rank makes this call (rank-invariant)
  if (m_global < 2) throw Error(ST_EINVAL, "PCA needs at least two samples");
  const bool fat = !sharded && m < n;  // the tall view is x^T: its ROWS are the data columns
  // column means of x = (1/m) x^T 1: one pass of the transposed-GEMM kernel against a ones vector
  RsvdDriver<Dev, T> drv(dev, profile);
  const int64_t samples_dim_tall = fat ? ta.nt : ta.mt;  // n_samples as a dimension of the tall view
  Skinny<T> ones = dev.template alloc_skinny<T>(samples_dim_tall, 1);
  dev.fill_const(ones.p, samples_dim_tall, (T)1);
  Skinny<T> mu = dev.template alloc_skinny<T>(fat ? ta.mt : ta.nt, 1);
  T* inv_m = dev.template alloc_scalar<T>(1);
  const T inv_m_host = (T)(1.0 / (double)m_global);
  dev.store_values(&inv_m_host, (int64_t)1, inv_m, /*dst_is_host=*/false);
  if (!fat)
    drv.at_times(ta, ones, mu, inv_m, sharded);  // mu (n) = x^T 1 / m (all-reduced partial sums when sharded)
  else
    drv.a_times(ta, ones, mu, inv_m);          // tall view = x^T (n x m): mu (n) = x^T 1 / m
  TallA<T> tc = ta;
  const bool fused = ro.pca_center == 1 || (ro.pca_center == 0 && sizeof(T) == 8);
  if (fused) {
    // the centred matrix is never formed.  Tall view (i, j) = x(i, j) for tall inputs (means run
    // along the SHORT side), = x(j, i) for fat inputs (means run along the TALL side).
    if (!fat)
      tc.mu_short = mu.p;
    else
      tc.mu_tall = mu.p;
  } else {
    // centred copy (center_mat_col clones too, mat_utils.rs:484): memory rows/cols of the staged operand
    const int64_t ldp = round_up(ta.mem.cols, kLdPad);
    T* cbuf = (T*)dev.alloc_bytes((size_t)ta.mem.rows * (size_t)ldp * sizeof(T));
    dev.memset_zero(cbuf, (size_t)ta.mem.rows * (size_t)ldp * sizeof(T));
    // data columns run along the memory columns iff (tall & row-major) or (fat & column-major-as-rows ...):
    // tall view element (i, j): row-major memory (i, j), else memory (j, i).  Data column index of x is j for
    // the tall case and i for the fat case.
    const bool mean_along_mem_cols = (ta.row_major != fat);
    dev.center_rows_cols(ta.mem.p, ta.mem.rows, ta.mem.cols, ta.mem.ld, mu.p, mean_along_mem_cols, cbuf, ldp);
    tc.mem.p = cbuf;
    tc.mem.ld = ldp;
    tc.mem.cols_readable = ldp;
  }
  const int64_t k = rank;
  const int64_t l = std::min<int64_t>(rank + n_oversamples, tc.nt);
  if (ro.omega && ro.omega_ld < tc.nt) throw Error(ST_EINVAL, "omega_ld < min(m, n)");
  Skinny<T> ut = dev.template alloc_skinny<T>(tc.mt, k);
  Skinny<T> vtall = dev.template alloc_skinny<T>(tc.nt, k);
  T* s_dev = dev.template alloc_scalar<T>((int)k);
  drv.random_svd_tall(tc, k, l, n_iter, ro, ut, s_dev, vtall, [&] {
    // components_ = vr = V^T (k x n_dim)   pca_rsvd.rs:70-71
    if (!fat)
      dev.copy_out(vtall, k, comps, ldc, /*transpose=*/true, host_ptrs);
    else
      dev.copy_out(ut, k, comps, ldc, true, host_ptrs);
    dev.copy_values_out(s_dev, k, s, host_ptrs);
    dev.copy_values_out(mu.p, n, means, host_ptrs);
  });
  PhaseTimer fin;
  drv.phase(drv.tm.finalize_ms, fin);
  dev.phase_end();
  dev.end_call();
  dev.phase_resolve(&drv.tm.total_ms);
  drv.tm.n_collectives = dev.n_collectives;
  drv.tm.collective_bytes = dev.collective_bytes;
  if (tm_out) *tm_out = drv.tm;
}

template <class Dev, class T>
inline void power_iter_entry(Dev& dev, bool host_ptrs, const T* a, int64_t m, int64_t n, int64_t rs, int64_t cs,
                             int64_t width, int64_t n_iter, const corrla_opts* opts, T* q, int64_t ldq) {
  if (!q) throw Error(ST_EINVAL, "q is NULL");
  validate_matrix(a, m, n, rs, cs);
  if (width < 1 || width > n) throw Error(ST_EINVAL, "width must be in [1, n]");
  if (n_iter < 0) throw Error(ST_EINVAL, "n_iter must be >= 0");
  if (ldq < m) throw Error(ST_EINVAL, "ldq < m");
  RunOpts ro = parse_opts(opts, !host_ptrs);
  if (ro.omega && ro.omega_ld < n) throw Error(ST_EINVAL, "omega_ld < n");
  if (!ro.seed_explicit && !ro.omega) ro.seed = dev.fresh_seed(false);
  dev.begin_call();
  TallA<T> ta = stage_input<Dev, T>(dev, host_ptrs, a, m, n, rs, cs, /*force_tall=*/true);
  RsvdDriver<Dev, T> drv(dev, false);
  Skinny<T> y = dev.template alloc_skinny<T>(ta.mt, width);
  Skinny<T> y2 = dev.template alloc_skinny<T>(ta.mt, width);
  drv.power_iter(ta, width, n_iter, ro, y, y2);
  dev.copy_out(y, width, q, ldq, false, host_ptrs);
  dev.end_call();
}

// res = beta * op(A) * X   (mat_utils.rs:20-33 for the two hot-path shapes)
template <class Dev, class T>
inline void matmul_entry(Dev& dev, int trans, const T* a, int64_t m, int64_t n, int64_t rs, int64_t cs, const T* x,
                         int64_t ldx, int64_t l, T beta, T* res, int64_t ldres, Timings* tm_out = nullptr) {
  if (!x || !res) throw Error(ST_EINVAL, "x or res is NULL");
  validate_matrix(a, m, n, rs, cs);
  if (l < 1) throw Error(ST_EINVAL, "l must be >= 1");
  const int64_t xin = trans ? m : n, xout = trans ? n : m;
  if (ldx < xin || ldres < xout) throw Error(ST_EINVAL, "leading dimension too small");
  dev.begin_call();
  TallA<T> ta = stage_input<Dev, T>(dev, false, a, m, n, rs, cs, true);
  RsvdDriver<Dev, T> drv(dev, false);
  drv.mixed_planes_ = parse_opts(nullptr, true).mixed_planes;  // CORRLA_SKETCH_MIXED (this hook takes no opts)
  Skinny<T> xs = dev.template alloc_skinny<T>(xin, l);
  dev.copy_in_skinny(x, ldx, xs);
  Skinny<T> out = dev.template alloc_skinny<T>(xout, l);
  T* beta_dev = dev.template alloc_scalar<T>(1);
  dev.store_values(&beta, (int64_t)1, beta_dev, /*dst_is_host=*/false);
  if (trans)
    drv.at_times(ta, xs, out, beta_dev, false);
  else
    drv.a_times(ta, xs, out, beta_dev);
  dev.copy_out(out, l, res, ldres, false, false);
  dev.end_call();
  if (tm_out) tm_out->n_mixed_products = drv.tm.n_mixed_products;
}

}  // namespace corrla
