// libcorrla_rsvd.so -- C ABI (include/corrla_rsvd.h) over the HIP backend.  This translation unit
// is the product: it contains no CPU compute path; without a gfx950 device every compute entry
// point returns CORRLA_ENODEV / CORRLA_EHIP.
#include <mutex>
#include <vector>

#include "capi_impl.hpp"
#include "hip_backend.hpp"
#include "grad_kernels.hpp"
#include "knn2_kernels.hpp"

using namespace corrla;

struct corrla_ctx {
  HipDev dev;
  Timings last;
  std::mutex mu;
  bool profile;
  explicit corrla_ctx(int ordinal) : dev(ordinal), profile(env_int("CORRLA_PROFILE_PHASES", 0) != 0) {
    // function attributes apply to the device that is current when they are set: once per context (HipDev's
    // constructor has made `ordinal` current), like every other kernel's
    const hipFuncAttribute attr = hipFuncAttributeMaxDynamicSharedMemorySize;
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::knn_kernel, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::knn_mfma_kernel<4, 4>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::knn_mfma_kernel<4, 8>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::knn_mfma_kernel<4, 16>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::knn_mfma_kernel<2, 4>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::knn_mfma_kernel<2, 8>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::knn_mfma_kernel<2, 16>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::grad_fit_kernel, attr, 160 * 1024));
  }
};

namespace {
inline corrla_ctx* need(corrla_ctx* c) {
  if (!c) throw Error(ST_EINVAL, "ctx is NULL");
  return c;
}
// Runs one entry point under the context lock.  On the exception path the stream is drained before the status is
// returned: kernels that write the caller's outputs, or still read the caller's Omega, may be queued, and the
// caller is free to release those buffers as soon as the call has failed.
template <class F>
inline void locked_call(corrla_ctx* c, F&& f) {
  std::lock_guard<std::mutex> lk(c->mu);
  try {
    f();
  } catch (...) {
    (void)hipSetDevice(c->dev.device);
    if (c->dev.stream) (void)hipStreamSynchronize(c->dev.stream);
    (void)hipGetLastError();
    throw;
  }
}

template <class T>
corrla_status rsvd_c(corrla_ctx* ctx, bool host, bool sharded, const T* a, int64_t m, int64_t n, int64_t rs, int64_t cs,
                     int64_t rank, int64_t n_iter, int64_t p, const corrla_opts* o, T* u, int64_t ldu, T* s, T* vt,
                     int64_t ldvt) {
  return guarded([&] {
    corrla_ctx* c = need(ctx);
    locked_call(c, [&] {
      rsvd_entry<HipDev, T>(c->dev, host, sharded, a, m, n, rs, cs, rank, n_iter, p, o, u, ldu, s, vt, ldvt, &c->last,
                            c->profile);
    });
    if (c->profile && env_int("CORRLA_DEBUG", 0))
      std::fprintf(stderr, "[corrla] qr breakdown (ms): gram %.3f  download+check %.3f  host chol/inv %.3f  upload+apply %.3f  (%d passes)\n",
                   c->last.qr_gram_ms, c->last.qr_down_ms, c->last.qr_host_ms, c->last.qr_apply_ms, c->last.qr_passes);
  });
}
template <class T>
corrla_status pca_c(corrla_ctx* ctx, bool host, const T* x, int64_t m, int64_t n, int64_t rs, int64_t cs, int64_t rank,
                    int64_t n_iter, int64_t p, const corrla_opts* o, T* means, T* s, T* comps, int64_t ldc,
                    bool sharded = false) {
  return guarded([&] {
    corrla_ctx* c = need(ctx);
    locked_call(c, [&] {
      pca_entry<HipDev, T>(c->dev, host, x, m, n, rs, cs, rank, n_iter, p, o, means, s, comps, ldc, &c->last, c->profile,
                           sharded);
    });
  });
}
template <class T>
corrla_status power_c(corrla_ctx* ctx, bool host, const T* a, int64_t m, int64_t n, int64_t rs, int64_t cs, int64_t width,
                      int64_t n_iter, const corrla_opts* o, T* q, int64_t ldq) {
  return guarded([&] {
    corrla_ctx* c = need(ctx);
    locked_call(c, [&] { power_iter_entry<HipDev, T>(c->dev, host, a, m, n, rs, cs, width, n_iter, o, q, ldq); });
  });
}
template <class T>
corrla_status matmul_c(corrla_ctx* ctx, int trans, const T* a, int64_t m, int64_t n, int64_t rs, int64_t cs, const T* x,
                       int64_t ldx, int64_t l, T beta, T* res, int64_t ldres) {
  return guarded([&] {
    corrla_ctx* c = need(ctx);
    locked_call(c, [&] { matmul_entry<HipDev, T>(c->dev, trans, a, m, n, rs, cs, x, ldx, l, beta, res, ldres, &c->last); });
  });
}
template <class T>
corrla_status fill_c(corrla_ctx* ctx, T* p, int64_t rows, int64_t cols, int64_t rs, int64_t cs, uint64_t seed,
                     int64_t row0, int64_t global_cols) {
  return guarded([&] {
    corrla_ctx* c = need(ctx);
    if (!p) throw Error(ST_EINVAL, "p is NULL");
    if (rows < 0 || cols < 0 || rs < 0 || cs < 0 || global_cols < cols) throw Error(ST_EINVAL, "bad fill_normal shape");
    std::lock_guard<std::mutex> lk(c->mu);
    CORRLA_HIP(hipSetDevice(c->dev.device));
    c->dev.fill_normal(p, rows, cols, rs, cs, seed, row0, global_cols);
    c->dev.sync();
  });
}
template <class T>
corrla_status time_sketch_c(corrla_ctx* ctx, const T* a, int64_t m, int64_t n, int64_t rs, int64_t cs, const T* x,
                            int64_t ldx, int64_t l, T* y, int64_t ldy, int reps, double* avg_ms) {
  return guarded([&] {
    corrla_ctx* c = need(ctx);
    if (!x || !y || !avg_ms) throw Error(ST_EINVAL, "NULL argument");
    if (reps < 1 || l < 1 || ldx < n || ldy < m) throw Error(ST_EINVAL, "bad time_sketch arguments");
    std::lock_guard<std::mutex> lk(c->mu);
    HipDev& dev = c->dev;
    dev.begin_call();
    TallA<T> ta = stage_input<HipDev, T>(dev, false, a, m, n, rs, cs, true);
    RsvdDriver<HipDev, T> drv(dev, false);
    drv.mixed_planes_ = parse_opts(nullptr, true).mixed_planes;  // CORRLA_SKETCH_MIXED (the hook takes no opts)
    Skinny<T> xs = dev.alloc_skinny<T>(n, l);
    dev.copy_in_skinny(x, ldx, xs);
    Skinny<T> out = dev.alloc_skinny<T>(m, l);
    *avg_ms = dev.time_on_stream(reps, [&] { drv.a_times(ta, xs, out, nullptr); });
    dev.copy_out(out, l, y, ldy, false, false);
    dev.end_call();
  });
}
}  // namespace

extern "C" {

CORRLA_API const char* corrla_version(void) { return "corrla_rsvd 0.1.0 gfx950"; }
CORRLA_API const char* corrla_last_error(void) { return last_error_slot().c_str(); }
CORRLA_API int corrla_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

CORRLA_API corrla_status corrla_ctx_create(int device_ordinal, corrla_ctx** out) {
  return guarded([&] {
    if (!out) throw Error(ST_EINVAL, "out is NULL");
    *out = nullptr;
    *out = new corrla_ctx(device_ordinal);
  });
}
CORRLA_API void corrla_ctx_destroy(corrla_ctx* ctx) { delete ctx; }
CORRLA_API corrla_status corrla_ctx_synchronize(corrla_ctx* ctx) {
  return guarded([&] { need(ctx)->dev.sync(); });
}
CORRLA_API corrla_status corrla_ctx_comm_info(corrla_ctx* ctx, int* rank, int* nranks) {
  return guarded([&] {
    corrla_ctx* c = need(ctx);
    if (!rank || !nranks) throw Error(ST_EINVAL, "NULL argument");
    std::lock_guard<std::mutex> lk(c->mu);
    c->dev.comm_info(rank, nranks);
  });
}
CORRLA_API corrla_status corrla_ctx_set_phase_timings(corrla_ctx* ctx, int on) {
  return guarded([&] {
    corrla_ctx* c = need(ctx);
    std::lock_guard<std::mutex> lk(c->mu);  // not while a call is in flight: its marks would be left half recorded
    c->dev.set_phase_events(on != 0);
  });
}
CORRLA_API corrla_status corrla_ctx_get_timings(corrla_ctx* ctx, corrla_timings* out) {
  return guarded([&] {
    corrla_ctx* c = need(ctx);
    if (!out) throw Error(ST_EINVAL, "out is NULL");
    const Timings& t = c->last;
    out->total_ms = t.total_ms;
    out->sketch_ms = t.sketch_ms;
    out->power_ms = t.power_ms;
    out->qr_ms = t.qr_ms;
    out->project_ms = t.project_ms;
    out->small_svd_ms = t.small_svd_ms;
    out->finalize_ms = t.finalize_ms;
    out->qr_passes = t.qr_passes;
    out->n_collectives = t.n_collectives;
    out->collective_bytes = t.collective_bytes;
    out->sketch_kernel_ms = t.sketch_kernel_ms;
    out->host_enqueue_ms = t.host_enqueue_ms;
    out->n_mixed_products = t.n_mixed_products;
    out->reserved_ = 0;
    out->knn_ms = t.knn_ms;
    out->fit_ms = t.fit_ms;
  });
}

#define CORRLA_DEFINE(SUF, T)                                                                                          \
  CORRLA_API corrla_status corrla_rsvd_##SUF(corrla_ctx* ctx, const T* a, int64_t m, int64_t n, int64_t rs, int64_t cs,           \
                                  int64_t rank, int64_t n_iter, int64_t p, const corrla_opts* o, T* u, int64_t ldu,    \
                                  T* s, T* vt, int64_t ldvt) {                                                         \
    return rsvd_c<T>(ctx, true, false, a, m, n, rs, cs, rank, n_iter, p, o, u, ldu, s, vt, ldvt);                      \
  }                                                                                                                    \
  CORRLA_API corrla_status corrla_rsvd_dev_##SUF(corrla_ctx* ctx, const T* a, int64_t m, int64_t n, int64_t rs, int64_t cs,       \
                                      int64_t rank, int64_t n_iter, int64_t p, const corrla_opts* o, T* u,             \
                                      int64_t ldu, T* s, T* vt, int64_t ldvt) {                                        \
    return rsvd_c<T>(ctx, false, false, a, m, n, rs, cs, rank, n_iter, p, o, u, ldu, s, vt, ldvt);                     \
  }                                                                                                                    \
  CORRLA_API corrla_status corrla_rsvd_sharded_dev_##SUF(corrla_ctx* ctx, const T* a, int64_t m, int64_t n, int64_t rs,           \
                                              int64_t cs, int64_t rank, int64_t n_iter, int64_t p,                     \
                                              const corrla_opts* o, T* u, int64_t ldu, T* s, T* vt, int64_t ldvt) {    \
    return rsvd_c<T>(ctx, false, true, a, m, n, rs, cs, rank, n_iter, p, o, u, ldu, s, vt, ldvt);                      \
  }                                                                                                                    \
  CORRLA_API corrla_status corrla_pca_##SUF(corrla_ctx* ctx, const T* x, int64_t m, int64_t n, int64_t rs, int64_t cs, \
                                            int64_t rank, int64_t n_iter, int64_t p, const corrla_opts* o, T* means,   \
                                            T* s, T* comps, int64_t ldc) {                                             \
    return pca_c<T>(ctx, true, x, m, n, rs, cs, rank, n_iter, p, o, means, s, comps, ldc);                             \
  }                                                                                                                    \
  CORRLA_API corrla_status corrla_pca_dev_##SUF(corrla_ctx* ctx, const T* x, int64_t m, int64_t n, int64_t rs,         \
                                                int64_t cs, int64_t rank, int64_t n_iter, int64_t p,                   \
                                                const corrla_opts* o, T* means, T* s, T* comps, int64_t ldc) {         \
    return pca_c<T>(ctx, false, x, m, n, rs, cs, rank, n_iter, p, o, means, s, comps, ldc);                            \
  }                                                                                                                    \
  CORRLA_API corrla_status corrla_pca_sharded_dev_##SUF(corrla_ctx* ctx, const T* x, int64_t m, int64_t n, int64_t rs,  \
                                                        int64_t cs, int64_t rank, int64_t n_iter, int64_t p,           \
                                                        const corrla_opts* o, T* means, T* s, T* comps, int64_t ldc) { \
    return pca_c<T>(ctx, false, x, m, n, rs, cs, rank, n_iter, p, o, means, s, comps, ldc, true);                      \
  }                                                                                                                    \
  CORRLA_API corrla_status corrla_power_iter_##SUF(corrla_ctx* ctx, const T* a, int64_t m, int64_t n, int64_t rs, int64_t cs,     \
                                        int64_t width, int64_t n_iter, const corrla_opts* o, T* q, int64_t ldq) {      \
    return power_c<T>(ctx, true, a, m, n, rs, cs, width, n_iter, o, q, ldq);                                           \
  }                                                                                                                    \
  CORRLA_API corrla_status corrla_power_iter_dev_##SUF(corrla_ctx* ctx, const T* a, int64_t m, int64_t n, int64_t rs,             \
                                            int64_t cs, int64_t width, int64_t n_iter, const corrla_opts* o, T* q,     \
                                            int64_t ldq) {                                                             \
    return power_c<T>(ctx, false, a, m, n, rs, cs, width, n_iter, o, q, ldq);                                          \
  }                                                                                                                    \
  CORRLA_API corrla_status corrla_matmul_dev_##SUF(corrla_ctx* ctx, int trans, const T* a, int64_t m, int64_t n, int64_t rs,      \
                                        int64_t cs, const T* x, int64_t ldx, int64_t l, T beta, T* res,                \
                                        int64_t ldres) {                                                               \
    return matmul_c<T>(ctx, trans, a, m, n, rs, cs, x, ldx, l, beta, res, ldres);                                      \
  }                                                                                                                    \
  CORRLA_API corrla_status corrla_fill_normal_dev_##SUF(corrla_ctx* ctx, T* p, int64_t rows, int64_t cols, int64_t rs,            \
                                             int64_t cs, uint64_t seed, int64_t row0, int64_t global_cols) {           \
    return fill_c<T>(ctx, p, rows, cols, rs, cs, seed, row0, global_cols);                                             \
  }                                                                                                                    \
  CORRLA_API corrla_status corrla_time_sketch_dev_##SUF(corrla_ctx* ctx, const T* a, int64_t m, int64_t n, int64_t rs,            \
                                             int64_t cs, const T* x, int64_t ldx, int64_t l, T* y, int64_t ldy,        \
                                             int reps, double* avg_ms) {                                               \
    return time_sketch_c<T>(ctx, a, m, n, rs, cs, x, ldx, l, y, ldy, reps, avg_ms);                                    \
  }

CORRLA_DEFINE(f32, float)
CORRLA_DEFINE(f64, double)

// ---- active-subspace gradient stage ------------------------------------------------------
static corrla_status grad_mat_c(corrla_ctx* ctx, bool host_ptrs, const double* x, int64_t n_pts, int64_t kf, const double* y,
                                const double* xq, int64_t n_q, int est_order, int64_t n_nbrs, double out_scale, double* g,
                                int64_t ldg, int* n_regularised) {
  return guarded([&] {
    corrla_ctx* c = need(ctx);
    if (!x || !y || !xq || !g) throw Error(ST_EINVAL, "NULL argument");
    if (n_pts < 1 || n_q < 1 || kf < 1) throw Error(ST_EINVAL, "empty point set");
    if (est_order != 1 && est_order != 2) throw Error(ST_EINVAL, "est_order must be 1 or 2 (the reference panics otherwise)");
    if (kf > k::kGradMaxDim) throw Error(ST_EINVAL, "more than 64 features are not supported");
    const int64_t need_pts = est_order == 1 ? kf + 1 : kf * (kf + 3) / 2;  // active_subspaces.rs:118-119, 129-130
    if (!(n_pts > need_pts && n_nbrs > need_pts))
      throw Error(ST_EINVAL, "n_pts and n_nbrs must exceed k + 1 (order 1) / k (k + 3) / 2 (order 2)");
    if (n_nbrs > n_pts) throw Error(ST_EINVAL, "n_nbrs exceeds the number of support points");
    if (n_nbrs > k::kGradMaxNbr) throw Error(ST_EINVAL, "more than 512 neighbours are not supported");
    if (k::grad_fit_lds_bytes((int)kf, (int)n_nbrs, est_order, /*m_in_lds=*/false) > (size_t)160 * 1024)
      throw Error(ST_EINVAL, "the neighbours of one query do not fit in 160 KiB of LDS");
    if (ldg < kf) throw Error(ST_EINVAL, "ldg < k");
    if (n_pts > 0x7fffffff || n_q * n_nbrs > ((int64_t)1 << 40)) throw Error(ST_EINVAL, "point set too large");
    locked_call(c, [&] {
    HipDev& dev = c->dev;
    dev.begin_call();
    const int kk = (int)kf, nn = (int)n_nbrs;
    const double* xd = x;
    const double* yd = y;
    const double* qd = xq;
    double* gd = g;
    if (host_ptrs) {
      double* xb = (double*)dev.alloc_bytes(sizeof(double) * (size_t)n_pts * kf);
      double* yb = (double*)dev.alloc_bytes(sizeof(double) * (size_t)n_pts);
      double* qb = (double*)dev.alloc_bytes(sizeof(double) * (size_t)n_q * kf);
      CORRLA_HIP(hipMemcpyAsync(xb, x, sizeof(double) * (size_t)n_pts * kf, hipMemcpyHostToDevice, dev.stream));
      CORRLA_HIP(hipMemcpyAsync(yb, y, sizeof(double) * (size_t)n_pts, hipMemcpyHostToDevice, dev.stream));
      CORRLA_HIP(hipMemcpyAsync(qb, xq, sizeof(double) * (size_t)n_q * kf, hipMemcpyHostToDevice, dev.stream));
      xd = xb;
      yd = yb;
      qd = qb;
      gd = (double*)dev.alloc_bytes(sizeof(double) * (size_t)n_q * kf);
    }
    const int64_t ldt = round_up(n_pts, 64);
    double* xt = (double*)dev.alloc_bytes(sizeof(double) * (size_t)ldt * kf);
    int* nbr = (int*)dev.alloc_bytes(sizeof(int) * (size_t)n_q * nn);
    int* status = (int*)dev.alloc_bytes(sizeof(int) * (size_t)n_q);
    hipLaunchKernelGGL(k::grad_transpose_kernel, dim3((unsigned)((n_pts + 255) / 256)), dim3(256), 0, dev.stream, xd, n_pts, kk, xt,
                       ldt);
    // normal equations in LDS when they fit next to the neighbours, else in a per-workgroup slice of global memory
    const bool m_in_lds = k::grad_fit_lds_bytes(kk, nn, est_order, true) <= (size_t)160 * 1024;
    const size_t lds_fit = k::grad_fit_lds_bytes(kk, nn, est_order, m_in_lds);
    if (lds_fit > (size_t)160 * 1024) throw Error(ST_EINVAL, "problem does not fit in LDS");
    if (n_q > 0x7fffffff) throw Error(ST_EINVAL, "too many query points for one launch");
    // Both kernels spend ~n_nbrs ln(n_pts / n_nbrs) list insertions per query; the MFMA distance tile only pays off
    // once the scan itself dominates (measured: 5e4 points 0.10 s VALU / 0.14 s MFMA, 1e5 0.28 / 0.30, 2e5 0.93 / 0.45)
    dev.event_mark(0);  // corrla_timings.knn_ms / fit_ms: events 0-1 around the scan, 1-2 around the fits
    const int knn_mode = env_int("CORRLA_KNN", 0);  // 1: VALU kernel, 2: f32-MFMA kernel, 3: bf16-filter kernel, 0: by size
    // knn2_kernels.hpp (round 3): bf16x3 MFMA filter + batched bitonic list merges; n_nbrs <= 128.  Small clouds keep the
    // VALU scan (its per-query lists live in LDS and there is too little work to amortise the split of the cloud).
    const bool knn2 = (knn_mode == 3 || (knn_mode == 0 && n_pts >= 8192)) && nn <= k::kK2List;
    if (knn2) {
      const int S = kk <= 32 ? 1 : 2;
      const int64_t nchunks = (n_pts + k::kK2Chunk - 1) / k::kK2Chunk;
      __bf16* pb = (__bf16*)dev.alloc_bytes((size_t)nchunks * (size_t)k::k2_chunk_bytes(S));
      float* pnf = (float*)dev.alloc_bytes((size_t)nchunks * k::kK2Chunk * 4 * sizeof(float));  // -c_p, four copies per point
      double* mean = (double*)dev.alloc_bytes(64 * sizeof(double));
      const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (n_pts + 4095) / 4096));
      const int64_t rpb = (n_pts + nb - 1) / nb;
      double* partial = (double*)dev.alloc_bytes((size_t)nb * 64 * sizeof(double));
      hipLaunchKernelGGL(k::knn2_colsum_kernel, dim3((unsigned)nb), dim3(256), 0, dev.stream, xd, n_pts, kk, rpb, partial);
      hipLaunchKernelGGL(k::knn2_mean_kernel, dim3(1), dim3(64), 0, dev.stream, (const double*)partial, nb, n_pts, kk, mean);
      hipLaunchKernelGGL(k::knn2_prep_kernel, dim3((unsigned)nchunks), dim3(256), 0, dev.stream, xd, n_pts, kk, (const double*)mean, S,
                         pb, pnf);
      k::Knn2Args ka;
      ka.pb = pb;
      ka.pn = pnf;
      ka.x = xd;
      ka.xq = qd;
      ka.mean = mean;
      ka.n_pts = n_pts;
      ka.n_q = n_q;
      ka.nchunks = nchunks;
      ka.ntiles = (n_q + k::kK2Q - 1) / k::kK2Q;
      ka.k = kk;
      ka.n_nbrs = nn;
      const int64_t wgs = std::min<int64_t>(ka.ntiles, (int64_t)dev.num_cus * std::max(1, env_int("CORRLA_KNN2_WGS_PER_CU", 1)));
      ka.cand = (int*)dev.alloc_bytes((size_t)wgs * k::kK2Q * k::kK2Cap * sizeof(int));
      ka.list_d = (double*)dev.alloc_bytes((size_t)wgs * k::kK2Q * k::kK2List * sizeof(double));
      ka.list_i = (int*)dev.alloc_bytes((size_t)wgs * k::kK2Q * k::kK2List * sizeof(int));
      ka.nbr = nbr;
      ka.prof = nullptr;
      const bool k2prof = env_int("CORRLA_KNN2_PROF", 0) != 0;
      if (k2prof) {
        ka.prof = (unsigned long long*)dev.alloc_bytes(4 * sizeof(unsigned long long));
        CORRLA_HIP(hipMemsetAsync(ka.prof, 0, 4 * sizeof(unsigned long long), dev.stream));
      }
      if (S == 1)
        hipLaunchKernelGGL((k::knn2_kernel<1>), dim3((unsigned)wgs), dim3(64 * k::kK2Waves), k::k2_lds_bytes(1), dev.stream, ka);
      else
        hipLaunchKernelGGL((k::knn2_kernel<2>), dim3((unsigned)wgs), dim3(64 * k::kK2Waves), k::k2_lds_bytes(2), dev.stream, ka);
      CORRLA_HIP(hipGetLastError());
      if (k2prof) {  // diagnostic only: synchronises
        unsigned long long h[4];
        CORRLA_HIP(hipMemcpyAsync(h, ka.prof, sizeof(h), hipMemcpyDeviceToHost, dev.stream));
        CORRLA_HIP(hipStreamSynchronize(dev.stream));
        std::fprintf(stderr, "knn2 prof (wave 0 of %lld workgroups, 100 MHz ticks): total %llu, flushes %llu (%.1f %%), chunk waits %llu (%.1f %%), "
                     "%llu merge batches\n", (long long)wgs, h[2], h[0], 100.0 * h[0] / (double)h[2], h[1], 100.0 * h[1] / (double)h[2], h[3]);
      }
    } else if (knn_mode == 1 || (knn_mode == 0 && n_pts < 131072) ||
               k::knn_mfma_lds_bytes(kk, nn, 2) > (size_t)160 * 1024) {
      // (also when the MFMA scan's per-query lists outgrow LDS: n_nbrs > ~400 at k = 64 -- the VALU scan's fit up to the
      //  interface's 512)
      const size_t lds_knn = k::knn_lds_bytes(kk, nn);
      if (lds_knn > (size_t)160 * 1024) throw Error(ST_EINVAL, "problem does not fit in LDS");
      const int64_t knn_blocks = (n_q + k::kKnnQueries - 1) / k::kKnnQueries;
      hipLaunchKernelGGL(k::knn_kernel, dim3((unsigned)knn_blocks), dim3(64 * k::kKnnWaves), lds_knn, dev.stream, (const double*)xt,
                         ldt, n_pts, kk, qd, n_q, nn, nbr);
    } else {
      double* pnorm = (double*)dev.alloc_bytes(sizeof(double) * (size_t)n_pts);
      hipLaunchKernelGGL(k::point_norms_kernel, dim3((unsigned)((n_pts + 255) / 256)), dim3(256), 0, dev.stream, (const double*)xt, ldt,
                         n_pts, kk, pnorm);
      const int waves = k::knn_mfma_lds_bytes(kk, nn, 4) <= (size_t)160 * 1024 ? 4 : 2;
      const size_t lds_knn = k::knn_mfma_lds_bytes(kk, nn, waves);
      if (lds_knn > (size_t)160 * 1024) throw Error(ST_EINVAL, "problem does not fit in LDS");
      const int64_t knn_blocks = (n_q + 16 * waves - 1) / (16 * waves);
      const int nks = k::knn_mfma_slices(kk);
#define CORRLA_KNN_LAUNCH(W_, S_)                                                                                              \
  hipLaunchKernelGGL((k::knn_mfma_kernel<W_, S_>), dim3((unsigned)knn_blocks), dim3(64 * W_), lds_knn, dev.stream, (const double*)xt, \
                     ldt, (const double*)pnorm, n_pts, kk, qd, n_q, nn, nbr)
      if (waves == 4) {
        if (nks == 4) CORRLA_KNN_LAUNCH(4, 4);
        else if (nks == 8) CORRLA_KNN_LAUNCH(4, 8);
        else CORRLA_KNN_LAUNCH(4, 16);
      } else {
        if (nks == 4) CORRLA_KNN_LAUNCH(2, 4);
        else if (nks == 8) CORRLA_KNN_LAUNCH(2, 8);
        else CORRLA_KNN_LAUNCH(2, 16);
      }
#undef CORRLA_KNN_LAUNCH
    }
    const int64_t ldgd = host_ptrs ? kf : ldg;
    dev.event_mark(1);
    // order 1: the MFMA-built normal equations (round 3; CORRLA_FIT=1 keeps the general kernel)
    if (est_order == 1 && env_int("CORRLA_FIT", 0) != 1) {
      const size_t lds_lin = k::grad_fit_lin_lds_bytes(kk, nn);
      const int ntt = (kk + 2 + 15) / 16;
      const k::FitRowTab rtab = k::grad_fit_lin_row_table(kk + 1);
      unsigned long long* fprof = nullptr;
      if (env_int("CORRLA_KNN2_PROF", 0)) {
        fprof = (unsigned long long*)dev.alloc_bytes(4 * sizeof(unsigned long long));
        CORRLA_HIP(hipMemsetAsync(fprof, 0, 4 * sizeof(unsigned long long), dev.stream));
      }
#define CORRLA_FIT_LAUNCH(N_)                                                                                               \
  hipLaunchKernelGGL((k::grad_fit_lin_kernel<N_>), dim3((unsigned)n_q), dim3(64), lds_lin, dev.stream, xd, yd, kk, qd, n_q, \
                     (const int*)nbr, nn, out_scale, gd, ldgd, status, fprof, rtab)
      switch (ntt) {
        case 1: CORRLA_FIT_LAUNCH(1); break;
        case 2: CORRLA_FIT_LAUNCH(2); break;
        case 3: CORRLA_FIT_LAUNCH(3); break;
        case 4: CORRLA_FIT_LAUNCH(4); break;
        default: CORRLA_FIT_LAUNCH(5); break;
      }
#undef CORRLA_FIT_LAUNCH
      if (fprof) {  // diagnostic only: synchronises
        unsigned long long h[4];
        CORRLA_HIP(hipMemcpyAsync(h, fprof, sizeof(h), hipMemcpyDeviceToHost, dev.stream));
        CORRLA_HIP(hipStreamSynchronize(dev.stream));
        const double tot = (double)(h[0] + h[1] + h[2]);
        std::fprintf(stderr, "fit prof (%llu queries, 100 MHz ticks per query): gather + normal equations %.0f (%.0f %%), Cholesky %.0f (%.0f %%), "
                     "solves %.0f (%.0f %%)\n", h[3], h[0] / (double)h[3], 100.0 * h[0] / tot, h[1] / (double)h[3], 100.0 * h[1] / tot,
                     h[2] / (double)h[3], 100.0 * h[2] / tot);
      }
    } else if (m_in_lds) {
      hipLaunchKernelGGL(k::grad_fit_kernel, dim3((unsigned)n_q), dim3(64), lds_fit, dev.stream, xd, yd, kk, qd, n_q, (const int*)nbr,
                         nn, est_order, out_scale, gd, ldgd, status, (double*)nullptr, (int64_t)0);
    } else {
      const int64_t wgs = std::min<int64_t>(n_q, 2 * (int64_t)dev.num_cus);
      const size_t melems = k::grad_fit_m_elems(kk, est_order);
      double* mg = (double*)dev.alloc_bytes((size_t)wgs * melems * sizeof(double));
      hipLaunchKernelGGL(k::grad_fit_kernel, dim3((unsigned)wgs), dim3(64), lds_fit, dev.stream, xd, yd, kk, qd, n_q, (const int*)nbr,
                         nn, est_order, out_scale, gd, ldgd, status, mg, (int64_t)melems);
    }
    CORRLA_HIP(hipGetLastError());
    // how many queries needed the ridge / failed: a short reduction on the host (n_q ints)
    std::vector<int> hs((size_t)n_q);
    CORRLA_HIP(hipMemcpyAsync(hs.data(), status, sizeof(int) * (size_t)n_q, hipMemcpyDeviceToHost, dev.stream));
    if (host_ptrs)
      CORRLA_HIP(hipMemcpy2DAsync(g, sizeof(double) * (size_t)ldg, gd, sizeof(double) * (size_t)kf, sizeof(double) * (size_t)kf,
                                  (size_t)n_q, hipMemcpyDeviceToHost, dev.stream));
    dev.event_mark(2);
    dev.end_call();
    c->last = Timings();
    c->last.knn_ms = dev.event_elapsed_ms(0, 1);
    c->last.fit_ms = dev.event_elapsed_ms(1, 2);
    c->last.total_ms = c->last.knn_ms + c->last.fit_ms;
    int bad = 0;
    for (int v : hs) bad += v != 0;
    if (n_regularised) *n_regularised = bad;
    });
  });
}
CORRLA_API corrla_status corrla_grad_mat_f64(corrla_ctx* ctx, const double* x, int64_t n_pts, int64_t k, const double* y,
                                             const double* xq, int64_t n_q, int est_order, int64_t n_nbrs, double out_scale,
                                             double* g, int64_t ldg, int* n_regularised) {
  return grad_mat_c(ctx, true, x, n_pts, k, y, xq, n_q, est_order, n_nbrs, out_scale, g, ldg, n_regularised);
}
CORRLA_API corrla_status corrla_grad_mat_dev_f64(corrla_ctx* ctx, const double* x, int64_t n_pts, int64_t k, const double* y,
                                                 const double* xq, int64_t n_q, int est_order, int64_t n_nbrs,
                                                 double out_scale, double* g, int64_t ldg, int* n_regularised) {
  return grad_mat_c(ctx, false, x, n_pts, k, y, xq, n_q, est_order, n_nbrs, out_scale, g, ldg, n_regularised);
}

CORRLA_API corrla_status corrla_comm_unique_id(void* out128) {
  return guarded([&] {
    if (!out128) throw Error(ST_EINVAL, "out128 is NULL");
    ncclUniqueId id;
    CORRLA_NCCL(ncclGetUniqueId(&id));
    std::memset(out128, 0, CORRLA_UNIQUE_ID_BYTES);
    std::memcpy(out128, &id, sizeof(id));
  });
}
CORRLA_API corrla_status corrla_ctx_comm_init(corrla_ctx* ctx, const void* unique_id128, int rank, int nranks) {
  return guarded([&] {
    corrla_ctx* c = need(ctx);
    if (!unique_id128 || nranks < 1 || rank < 0 || rank >= nranks) throw Error(ST_EINVAL, "bad communicator arguments");
    std::lock_guard<std::mutex> lk(c->mu);
    c->dev.comm_init(unique_id128, rank, nranks);
  });
}

}  // extern "C"
