// TEST INFRASTRUCTURE ONLY -- never linked into libcorrla_rsvd.so, never shipped.
//
// A host emulation of the device backend interface (hip_backend.hpp) so that the `-m "not gpu"`
// tests can drive the REAL algorithm driver (corrla_rs_amd/csrc/driver.hpp) and the REAL C-ABI glue
// (capi_impl.hpp: validation, fat/tall + stride classification, output orientation, the
// orthonormalisation pass logic, the sharded exchange points) on a machine without a GPU.
// It restates nothing from the reference and is not an alternative product path: the product
// library has no CPU fallback and returns CORRLA_ENODEV without a gfx950 device.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <vector>

#include "../../corrla_rs_amd/csrc/capi_impl.hpp"

using namespace corrla;

typedef void (*emu_allreduce_fn)(void* buf, uint64_t count, int is_f64);
static emu_allreduce_fn g_allreduce = nullptr;
static int g_nranks = 1;
static int g_rank = 0;

static inline void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0, c[1] = n1, c[2] = n2, c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
static inline float normal_f32(uint64_t idx, uint64_t seed) {
  uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), 0u, 0u};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const float u1 = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = ((float)(c[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  return std::sqrt(-2.0f * std::log(u1)) * (float)std::cos(2.0 * M_PI * (double)u2);
}
static inline double normal_f64(uint64_t idx, uint64_t seed) {
  uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), 0u, 0u};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const uint64_t a = (((uint64_t)c[0] << 32) | c[1]) >> 11;
  const uint64_t b = (((uint64_t)c[2] << 32) | c[3]) >> 11;
  const double u1 = ((double)a + 0.5) * (1.0 / 9007199254740992.0);
  const double u2 = ((double)b + 0.5) * (1.0 / 9007199254740992.0);
  return std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2);
}
template <class T>
static inline T normal_from_index(uint64_t idx, uint64_t seed);
template <>
inline float normal_from_index<float>(uint64_t i, uint64_t s) { return normal_f32(i, s); }
template <>
inline double normal_from_index<double>(uint64_t i, uint64_t s) { return normal_f64(i, s); }

class EmuDev {
 public:
  std::vector<std::unique_ptr<char[]>> blocks;
  int n_collectives = 0;
  double collective_bytes = 0;
  int nranks() const { return g_nranks; }
  uint64_t fresh_seed(bool) { return 0x5eedull; }   // the emulation stays deterministic
  void begin_call() { blocks.clear(); }
  void end_call() {}
  void sync() {}
  void event_mark(int) {}
  void phase_mark(double*) {}
  void phase_resolve(double*) {}
  void phase_forget() {}
  double event_elapsed_ms(int, int) { return 0.0; }
  void* alloc_bytes(size_t bytes) {
    blocks.emplace_back(new char[bytes + 64]);
    char* p = blocks.back().get();
    p += (64 - ((uintptr_t)p % 64)) % 64;
    return p;
  }
  void memset_zero(void* p, size_t bytes) { std::memset(p, 0, bytes); }
  template <class T>
  Skinny<T> alloc_skinny(int64_t rows, int64_t cols) {
    Skinny<T> s;
    s.rows = rows;
    s.cols = cols;
    s.ld = round_up(std::max<int64_t>(rows, 1), kLdPad);
    s.cols_alloc = col_blocking(cols).cols_alloc;
    const size_t bytes = (size_t)s.ld * s.cols_alloc * sizeof(T);
    s.p = (T*)alloc_bytes(bytes);
    std::memset(s.p, 0, bytes);
    return s;
  }
  template <class T>
  Skinny<T> alloc_skinny_out(int64_t rows, int64_t cols) {
    Skinny<T> s = alloc_skinny<T>(rows, cols);
    // poison everything a product must overwrite, so that a consumer of stale "zeros" shows up in the CPU tests
    for (int64_t c = 0; c < s.cols_alloc; ++c)
      for (int64_t r = 0; r < rows; ++r) s.p[c * s.ld + r] = std::numeric_limits<T>::quiet_NaN();
    return s;
  }
  double* alloc_f64(int n) {
    double* p = (double*)alloc_bytes(sizeof(double) * n);
    std::memset(p, 0, sizeof(double) * n);
    return p;
  }
  template <class T>
  T* alloc_scalar(int n) {
    T* p = (T*)alloc_bytes(sizeof(T) * n);
    std::memset(p, 0, sizeof(T) * n);
    return p;
  }
  template <class T>
  void h2d_2d(T* dst, int64_t dp, const T* src, int64_t sp, int64_t width, int64_t rows) {
    for (int64_t r = 0; r < rows; ++r) std::memcpy(dst + r * dp, src + r * sp, sizeof(T) * width);
  }
  // Same contracts as the HIP kernels, including the padded-column writes.
  template <class T>
  void gemm_nn(const Big<T>& r, const Skinny<T>& x, Skinny<T>& out, const T* scale) {
    if (x.rows != r.cols) throw Error(ST_EINVAL, "gemm_nn: inner dimensions differ");
    if (skipped()) return;
    const ColBlocking cb = col_blocking(x.cols);
    if (cb.cols_alloc > x.cols_alloc || (!out.external && cb.cols_alloc > out.cols_alloc)) throw Error(ST_EINVAL, "emu: column padding");
    if (out.rows != r.rows) throw Error(ST_EINVAL, "emu: gemm output shape mismatch");
    const double sc = scale ? (double)*scale : 1.0;
    for (int64_t c = 0; c < (out.external ? out.cols : cb.cols_alloc); ++c)
      for (int64_t i = 0; i < r.rows; ++i) {
        double s = 0.0;
        for (int64_t kk = 0; kk < r.cols; ++kk) s += (double)r.p[i * r.ld + kk] * (double)x.p[c * x.ld + kk];
        out.p[c * out.ld + i] = (T)(s * sc);
      }
  }
  template <class T>
  void gemm_tn(const Big<T>& r, const Skinny<T>& x, Skinny<T>& out, const T* scale) {
    if (x.rows != r.rows) throw Error(ST_EINVAL, "gemm_tn: inner dimensions differ");
    if (skipped()) return;
    const ColBlocking cb = col_blocking(x.cols);
    if (cb.cols_alloc > x.cols_alloc || (!out.external && cb.cols_alloc > out.cols_alloc)) throw Error(ST_EINVAL, "emu: column padding");
    if (out.rows != r.cols) throw Error(ST_EINVAL, "emu: gemm output shape mismatch");
    const double sc = scale ? (double)*scale : 1.0;
    for (int64_t c = 0; c < (out.external ? out.cols : cb.cols_alloc); ++c)
      for (int64_t j = 0; j < r.cols; ++j) {
        double s = 0.0;
        for (int64_t i = 0; i < r.rows; ++i) s += (double)r.p[i * r.ld + j] * (double)x.p[c * x.ld + i];
        out.p[c * out.ld + j] = (T)(s * sc);
      }
  }
  // same contract as HipDev::ata_fused: z = A^T (A x), the intermediate rounded to T like the kernel's f32 T tile
  template <class T>
  bool ata_fused_fits(const Big<T>& a, int64_t l) const {
    return sizeof(T) == 4 && a.cols <= 512 && a.cols >= 16 && l <= 80 && a.rows >= 4096;
  }
  template <class T>
  void ata_fused(const Big<T>& a, const Skinny<T>& x, Skinny<T>& z) {
    const ColBlocking cb = col_blocking(x.cols);
    std::vector<double> acc((size_t)a.cols);
    for (int64_t c = 0; c < cb.cols_alloc; ++c) {
      std::fill(acc.begin(), acc.end(), 0.0);
      for (int64_t i = 0; i < a.rows; ++i) {
        double t = 0.0;
        for (int64_t kk = 0; kk < a.cols; ++kk) t += (double)a.p[i * a.ld + kk] * (double)x.p[c * x.ld + kk];
        const double tr = (double)(T)t;
        for (int64_t kk = 0; kk < a.cols; ++kk) acc[(size_t)kk] += (double)a.p[i * a.ld + kk] * tr;
      }
      for (int64_t kk = 0; kk < a.cols; ++kk) z.p[c * z.ld + kk] = (T)acc[(size_t)kk];
    }
  }
  // the bf16-split products exist on the GPU only: the emulation keeps every product exact (the flags are accepted)
  template <class T>
  bool mixed_fits(bool, const Big<T>&, const Skinny<T>&, const Skinny<T>&) const { return false; }
  template <class T>
  void gemm_mixed(bool, const Big<T>&, const Skinny<T>&, Skinny<T>&, const T*, int) { throw Error(ST_EINVAL, "emu: no bf16-split products"); }
  template <class T>
  void allreduce(T* p, size_t count) {
    if (g_nranks <= 1) return;
    if (!g_allreduce) throw Error(ST_ECOMM, "emu: no allreduce callback");
    g_allreduce(p, count, sizeof(T) == 8);
    ++n_collectives;
    collective_bytes += (double)count * sizeof(T);
  }
  void allreduce_f64(double* p, size_t count) { allreduce<double>(p, count); }
  // HipDev::sharded_handshake: one MAX all-reduce of (status, schedule state) at the start of a sharded call
  // (callback flag bit 1 = max instead of sum)
  int sharded_handshake(int local_status) {
    if (g_nranks <= 1) return local_status;
    if (!g_allreduce) throw Error(ST_ECOMM, "emu: no allreduce callback");
    double h[8] = {(double)local_status, (double)robust_passes_, 0, 0, 0, 0, 0, 0};
    g_allreduce(h, 8, 1 | 2);
    ++n_collectives;
    collective_bytes += sizeof(h);
    robust_passes_ = (int)h[1];
    return (int)h[0];
  }
  int arena_mark() const { return 0; }
  void arena_rewind(int) {}
  template <class T>
  void poison_entry(Skinny<T>& s, int64_t i, int64_t j, int kind) {
    s.p[j * s.ld + i] = kind == 2 ? std::numeric_limits<T>::infinity() : std::numeric_limits<T>::quiet_NaN();
  }
  int64_t allreduce_sum_host(int64_t v) {
    if (g_nranks <= 1) return v;
    double h = (double)v;
    allreduce<double>(&h, 1);
    return (int64_t)(h + 0.5);
  }
  template <class T>
  void download_skinny(const Skinny<T>& s, int64_t rows, int64_t cols, double* host) {
    for (int64_t j = 0; j < cols; ++j)
      for (int64_t i = 0; i < rows; ++i) host[j * rows + i] = (double)s.p[j * s.ld + i];
  }
  template <class T>
  void upload_skinny(const double* host, int64_t rows, int64_t cols, int64_t ldh, Skinny<T>& dst) {
    std::memset(dst.p, 0, (size_t)dst.ld * dst.cols_alloc * sizeof(T));
    for (int64_t j = 0; j < cols; ++j)
      for (int64_t i = 0; i < rows; ++i) dst.p[j * dst.ld + i] = (T)host[j * ldh + i];
  }
  template <class T>
  void upload_skinny_native(const T* host, int64_t ldh, Skinny<T>& dst) {
    for (int64_t j = 0; j < dst.cols; ++j) std::memcpy(dst.p + j * dst.ld, host + j * ldh, sizeof(T) * dst.rows);
  }
  template <class T>
  void copy_in_skinny(const T* src, int64_t lds, Skinny<T>& dst) { upload_skinny_native(src, lds, dst); }
  template <class T>
  void copy_skinny(const Skinny<T>& src, Skinny<T>& dst) {
    std::memcpy(dst.p, src.p, (size_t)src.ld * src.cols_alloc * sizeof(T));
  }
  template <class T>
  void copy_cols(const Skinny<T>& src, Skinny<T>& dst, int64_t c0, int64_t n) {
    if (src.ld != dst.ld) throw Error(ST_EINVAL, "internal: copy_cols needs equal leading dimensions");
    if (n > 0) std::memcpy(dst.p + c0 * dst.ld, src.p, (size_t)n * src.ld * sizeof(T));
  }
  template <class T>
  void sub_inplace(Skinny<T>& y, const Skinny<T>& p) {
    const int64_t n = y.ld * std::min(y.cols_alloc, p.cols_alloc);
    for (int64_t i = 0; i < n; ++i) y.p[i] -= p.p[i];
  }
  template <class T>
  void zero_cols(Skinny<T>& s, int64_t c0, int64_t c1) {
    if (c1 > c0) std::memset(s.p + c0 * s.ld, 0, (size_t)(c1 - c0) * s.ld * sizeof(T));
  }
  template <class T>
  void store_values(const T* src, int64_t n, T* dst, bool) { std::memcpy(dst, src, sizeof(T) * n); }
  // same status semantics as k::chol_inv_kernel (0 ok, 1 pivot failure -> identity, 2 zero, 3 non-finite)
  struct EmuCholStatus {
    int fail;
    float min_ratio, dev_i, gmax;
    long long clk, wall;  // same 32-byte record as the device kernel's
  };
  template <class T>
  bool device_chol_fits(int64_t l) const {
    return l <= (sizeof(T) == 8 ? 152 : 176) && !std::getenv("CORRLA_EMU_NO_DEVICE_CHOL");
  }
  template <class T>
  bool device_chol_blocked_fits(int64_t l) const {
    return l > 8 && l <= 276 && !std::getenv("CORRLA_EMU_NO_DEVICE_CHOL");
  }
  // ---- Householder TSQR with explicit thin Q: the same panel partition, pairwise tree and reverse application as
  // csrc/tsqr_kernels.hpp / HipDev::householder_thin_q, written with plain loops --------------------------------
  template <class T>
  bool householder_fits(int64_t l) const {
    // one 2 l x l panel plus the blocked form's 16 x 16 T / parked-R blocks and tau (528 elements) in 160 KB of LDS
    return l >= 1 && (size_t)((2 * l + 3) / 4 * 4) * (size_t)l * sizeof(T) + 64 + 528 * sizeof(T) <= (size_t)160 * 1024 && 2 * l <= 320;
  }
  template <class T>
  static void hh_factor_panel(std::vector<T>& P, int rows, int l, T* tau) {  // P column-major rows x l
    for (int j = 0; j < l; ++j) {
      T* cj = P.data() + (size_t)j * rows;
      T sigma = 0;
      for (int r = j + 1; r < rows; ++r) sigma += cj[r] * cj[r];
      const T alpha = cj[j];
      T t = 0, beta = alpha, scale = 0;
      if (sigma > (T)0) {
        beta = -std::copysign(std::sqrt(alpha * alpha + sigma), alpha);
        t = (beta - alpha) / beta;
        scale = (T)1 / (alpha - beta);
      }
      for (int r = j + 1; r < rows; ++r) cj[r] *= scale;
      cj[j] = beta;
      tau[j] = t;
      for (int c = j + 1; c < l; ++c) {
        T* cc = P.data() + (size_t)c * rows;
        T d = cc[j];
        for (int r = j + 1; r < rows; ++r) d += cj[r] * cc[r];
        const T w = t * d;
        cc[j] -= w;
        for (int r = j + 1; r < rows; ++r) cc[r] -= w * cj[r];
      }
    }
  }
  template <class T>
  static void hh_apply_panel(std::vector<T>& P, int rows, int l, const T* V, int64_t ldv, const T* tau) {
    for (int j = l - 1; j >= 0; --j) {
      if (tau[j] == (T)0) continue;
      const T* v = V + (int64_t)j * ldv;
      for (int c = 0; c < l; ++c) {
        T* cc = P.data() + (size_t)c * rows;
        T d = cc[j];
        for (int r = j + 1; r < rows; ++r) d += v[r] * cc[r];
        const T w = tau[j] * d;
        cc[j] -= w;
        for (int r = j + 1; r < rows; ++r) cc[r] -= w * v[r];
      }
    }
  }
  // HipDev::householder_up / householder_down: the two sweeps, so that a cross-rank TSQR can put its exchange between
  template <class T>
  struct HhState {
    int64_t m = 0;
    int l = 0, nleaf = 0, levels = 0;
    std::vector<int> n_at;
    std::vector<std::vector<T>> rbuf, cbuf, taub, vbuf;
    T* r_root() { return rbuf[levels].data(); }
  };
  template <class T>
  HhState<T> householder_up(Skinny<T>& y, Skinny<T>& tmp) {
    HhState<T> h;
    const int64_t m = y.rows;
    const int l = (int)y.cols;
    if (m < l) throw Error(ST_EINVAL, "householder_thin_q: fewer rows than columns");
    const int64_t br = 2 * (int64_t)l;
    const int nleaf = (int)(m <= br ? 1 : (m + br - 1) / br);
    auto row0 = [&](int i) { return (m * (int64_t)i) / nleaf; };
    std::vector<int>& n_at = h.n_at;
    n_at = {nleaf};
    while (n_at.back() > 1) n_at.push_back((n_at.back() + 1) / 2);
    const int levels = (int)n_at.size() - 1;
    h.m = m;
    h.l = l;
    h.nleaf = nleaf;
    h.levels = levels;
    const size_t ll = (size_t)l * l;
    auto &rbuf = h.rbuf, &cbuf = h.cbuf, &taub = h.taub, &vbuf = h.vbuf;
    rbuf.resize(levels + 1);
    cbuf.resize(levels + 1);
    taub.resize(levels + 1);
    vbuf.resize(levels + 1);
    for (int k = 0; k <= levels; ++k) {
      rbuf[k].assign(ll * n_at[k], 0);
      cbuf[k].assign(ll * n_at[k], 0);
      taub[k].assign((size_t)l * n_at[k], 0);
      if (k >= 1) vbuf[k].assign(2 * ll * n_at[k], 0);
    }
    for (int i = 0; i < nleaf; ++i) {  // leaves (up)
      const int64_t r0 = row0(i);
      const int rows = (int)(row0(i + 1) - r0);
      std::vector<T> P((size_t)rows * l);
      for (int c = 0; c < l; ++c)
        for (int r = 0; r < rows; ++r) P[(size_t)c * rows + r] = y.p[(int64_t)c * y.ld + r0 + r];
      hh_factor_panel(P, rows, l, taub[0].data() + (size_t)i * l);
      for (int c = 0; c < l; ++c)
        for (int r = 0; r < rows; ++r) {
          tmp.p[(int64_t)c * tmp.ld + r0 + r] = P[(size_t)c * rows + r];
          if (r < l) rbuf[0][(size_t)i * ll + (size_t)c * l + r] = r <= c ? P[(size_t)c * rows + r] : (T)0;
        }
    }
    for (int k = 1; k <= levels; ++k)  // tree (up)
      for (int t = 0; t < n_at[k]; ++t) {
        const int a = 2 * t, b = 2 * t + 1;
        if (b >= n_at[k - 1]) {
          std::copy(rbuf[k - 1].begin() + (size_t)a * ll, rbuf[k - 1].begin() + (size_t)(a + 1) * ll, rbuf[k].begin() + (size_t)t * ll);
          continue;  // tau stays 0
        }
        const int rows = 2 * l;
        std::vector<T> P((size_t)rows * l);
        for (int c = 0; c < l; ++c)
          for (int r = 0; r < l; ++r) {
            P[(size_t)c * rows + r] = rbuf[k - 1][(size_t)a * ll + (size_t)c * l + r];
            P[(size_t)c * rows + l + r] = rbuf[k - 1][(size_t)b * ll + (size_t)c * l + r];
          }
        hh_factor_panel(P, rows, l, taub[k].data() + (size_t)t * l);
        std::copy(P.begin(), P.end(), vbuf[k].begin() + (size_t)t * 2 * ll);
        for (int c = 0; c < l; ++c)
          for (int r = 0; r < l; ++r) rbuf[k][(size_t)t * ll + (size_t)c * l + r] = r <= c ? P[(size_t)c * rows + r] : (T)0;
      }
    return h;
  }
  template <class T>
  void householder_down(HhState<T>& h, Skinny<T>& y, Skinny<T>& tmp, const T* root_coef = nullptr) {
    const int64_t m = h.m;
    const int l = h.l, nleaf = h.nleaf, levels = h.levels;
    const size_t ll = (size_t)l * l;
    auto row0 = [&](int i) { return (m * (int64_t)i) / nleaf; };
    auto &n_at = h.n_at;
    auto &cbuf = h.cbuf, &taub = h.taub, &vbuf = h.vbuf;
    auto coeff = [&](int k, int t, int r, int c) -> T {  // root: identity unless a coefficient block is given
      if (k == levels) return root_coef ? root_coef[(size_t)c * l + r] : (r == c ? (T)1 : (T)0);
      return cbuf[k][(size_t)t * ll + (size_t)c * l + r];
    };
    for (int k = levels; k >= 1; --k)  // tree (down)
      for (int t = 0; t < n_at[k]; ++t) {
        const int a = 2 * t, b = 2 * t + 1;
        if (b >= n_at[k - 1]) {
          for (int c = 0; c < l; ++c)
            for (int r = 0; r < l; ++r) cbuf[k - 1][(size_t)a * ll + (size_t)c * l + r] = coeff(k, t, r, c);
          continue;
        }
        const int rows = 2 * l;
        std::vector<T> P((size_t)rows * l, (T)0);
        for (int c = 0; c < l; ++c)
          for (int r = 0; r < l; ++r) P[(size_t)c * rows + r] = coeff(k, t, r, c);
        hh_apply_panel(P, rows, l, vbuf[k].data() + (size_t)t * 2 * ll, (int64_t)rows, taub[k].data() + (size_t)t * l);
        for (int c = 0; c < l; ++c)
          for (int r = 0; r < l; ++r) {
            cbuf[k - 1][(size_t)a * ll + (size_t)c * l + r] = P[(size_t)c * rows + r];
            cbuf[k - 1][(size_t)b * ll + (size_t)c * l + r] = P[(size_t)c * rows + l + r];
          }
      }
    for (int i = 0; i < nleaf; ++i) {  // leaves (down)
      const int64_t r0 = row0(i);
      const int rows = (int)(row0(i + 1) - r0);
      std::vector<T> P((size_t)rows * l, (T)0);
      for (int c = 0; c < l; ++c)
        for (int r = 0; r < l && r < rows; ++r) P[(size_t)c * rows + r] = coeff(0, i, r, c);
      hh_apply_panel(P, rows, l, tmp.p + r0, tmp.ld, taub[0].data() + (size_t)i * l);
      for (int c = 0; c < l; ++c)
        for (int r = 0; r < rows; ++r) y.p[(int64_t)c * y.ld + r0 + r] = P[(size_t)c * rows + r];
    }
  }
  template <class T>
  void householder_thin_q(Skinny<T>& y, Skinny<T>& tmp) {
    auto h = householder_up(y, tmp);
    householder_down(h, y, tmp);
  }
  template <class T>
  int householder_max_width() const {
    int w = 1;
    while (householder_fits<T>(w + 1)) ++w;
    return w;
  }
  int rank() const { return g_rank; }

  template <class T>
  void copy_block(const Skinny<T>& src, int64_t r0, int64_t c0, int64_t rows, int64_t cols, Skinny<T>& dst, int64_t dr0,
                  int64_t dc0) {
    for (int64_t j = 0; j < cols; ++j)
      for (int64_t i = 0; i < rows; ++i) dst.p[(dc0 + j) * dst.ld + dr0 + i] = src.p[(c0 + j) * src.ld + r0 + i];
  }
  template <class T>
  void chol_inv(const Skinny<T>& g, int64_t r, T piv_rel, Skinny<T>& m_out, void* st_dev, int slot) {
    EmuCholStatus* st = (EmuCholStatus*)st_dev + slot;
    std::vector<double> a((size_t)r * r);
    double dv = 0.0, gm = 0.0;
    bool finite = true;
    for (int64_t j = 0; j < r; ++j)
      for (int64_t i = 0; i < r; ++i) {
        const double v = (double)g.p[j * g.ld + i];
        a[j * r + i] = v;
        finite = finite && std::isfinite(v);
        dv = std::max(dv, std::fabs(v - (i == j ? 1.0 : 0.0)));
        if (i == j) gm = std::max(gm, v);
      }
    std::memset(m_out.p, 0, (size_t)m_out.ld * m_out.cols_alloc * sizeof(T));
    st->dev_i = (float)dv;
    st->gmax = (float)gm;
    st->min_ratio = 1.f;
    if (!finite) {
      st->fail = 3;
      return;
    }
    if (!(gm > 0.0)) {
      st->fail = 2;
      return;
    }
    if (dv <= (sizeof(T) == 4 ? 2.0e-4 : 1.0e-8)) {  // polishing pass: (I + E)^(-1/2) = I - E/2 to below eps
      for (int64_t j = 0; j < r; ++j)
        for (int64_t i = 0; i < r; ++i)
          m_out.p[j * m_out.ld + i] = (T)((i == j ? 1.0 : 0.0) - 0.5 * (a[j * r + i] - (i == j ? 1.0 : 0.0)));
      st->fail = 0;
      return;
    }
    double mr = 0.0;
    // the device kernel factorizes in T precision: round G through T first
    if (!small::chol_upper((int)r, a.data(), (int)r, (double)piv_rel, &mr)) {
      st->fail = 1;
      for (int64_t i = 0; i < r; ++i) m_out.p[i * m_out.ld + i] = (T)1;
      return;
    }
    small::triu_inverse((int)r, a.data(), (int)r);
    for (int64_t j = 0; j < r; ++j)
      for (int64_t i = 0; i <= j; ++i) m_out.p[j * m_out.ld + i] = (T)a[j * r + i];
    st->fail = 0;
    st->min_ratio = (float)mr;
  }
  // ---- device-robust Cholesky-QR (same contracts as HipDev; the host emulation reads the run_if word directly) ----
  const int* run_if_ = nullptr;
  bool skipped() const { return run_if_ && *run_if_ == 0; }
  void set_run_if(const int* p) { run_if_ = p; }
  void phase_end() {}
  int robust_passes_ = std::getenv("CORRLA_ROBUST_PASSES") ? std::atoi(std::getenv("CORRLA_ROBUST_PASSES")) : 2;
  int robust_passes() const { return robust_passes_; }
  void set_robust_passes(int n) { robust_passes_ = n; }
  bool svd_more_sweeps() { return false; }
  bool svd_force_v() { return false; }
  void svd_sweeps_used(int) {}
  int* alloc_flags(int n) {
    int* p = (int*)alloc_bytes(sizeof(int) * (size_t)std::max(n, 1));
    std::memset(p, 0, sizeof(int) * (size_t)std::max(n, 1));
    return p;
  }
  void* alloc_zeroed_bytes(size_t bytes) {
    void* p = alloc_bytes(bytes);
    std::memset(p, 0, bytes);
    return p;
  }
  void read_flags(const int* dev_p, int n, int* host) { std::memcpy(host, dev_p, sizeof(int) * (size_t)n); }
  void read_bytes(const void* dev_p, size_t bytes, void* host) { std::memcpy(host, dev_p, bytes); }
  template <class T>
  bool device_qr_robust_fits(int64_t l) const {
    return (device_chol_fits<T>(l) || device_chol_blocked_fits<T>(l)) && !std::getenv("CORRLA_EMU_NO_ROBUST_QR");
  }
  bool qr_inplace_fits(int64_t l) const { return col_blocking(l).nblk == 1; }
  template <class T>
  struct EmuInspect {
    T shift;
    int shifted, bad;
    float d2, gmax;
  };
  template <class T>
  void* alloc_inspect() { return alloc_zeroed_bytes(sizeof(EmuInspect<T>)); }
  template <class T>
  const void* inspect_shift_ptr(const void* insp) const { return &((const EmuInspect<T>*)insp)->shift; }
  template <class T>
  void gram_inspect(Skinny<T>& g, int64_t l, float shift_rel, int shift_mode, void* insp) {
    if (skipped()) return;
    EmuInspect<T>* o = (EmuInspect<T>*)insp;
    double dv = 0.0, gm = 0.0;
    bool finite = true;
    for (int64_t j = 0; j < l; ++j)
      for (int64_t i = 0; i < l; ++i) {
        const double v = (double)g.p[j * g.ld + i];
        finite = finite && std::isfinite(v);
        dv = std::max(dv, std::fabs(v - (i == j ? 1.0 : 0.0)));
        if (i == j) gm = std::max(gm, v);
      }
    const bool shifted = finite && gm > 0.0 && shift_rel > 0.f && (shift_mode == 1 || (shift_mode == 0 && dv > 0.25));
    o->shift = shifted ? (T)((double)shift_rel * gm) : (T)0;
    o->shifted = shifted ? 1 : 0;
    o->bad = finite ? 0 : 1;
    o->d2 = (float)dv;
    o->gmax = (float)gm;
    if (shifted)
      for (int64_t i = 0; i < l; ++i) g.p[i * g.ld + i] += o->shift;
  }
  template <class T>
  void combine_need(int* need, const void* insp, const int* na, const int* nb) {
    if (skipped()) return;
    const EmuInspect<T>* o = (const EmuInspect<T>*)insp;
    const int sub = *na | *nb;
    *need = (((o && (o->shifted || o->bad || o->d2 > 0.05f)) || sub) ? 1 : 0) | ((o && o->bad) ? kFlagNonFinite : 0) |
            (sub & (kFlagNonFinite | kFlagNullCols));
  }
  // k::chol_inv_kernel with a CholRobust record: shifted factorisation, failed pivots are null columns (zero columns of
  // R^-1, the factor is that of the Gram with those rows and columns deleted), need_next / null_mask outputs
  template <class T>
  void chol_inv_robust(const Skinny<T>& g, int64_t r, T piv_rel, float shift_rel, int shift_mode, float null_excess,
                       Skinny<T>& m_out, void* st_dev, int slot, int* need_next, int* null_mask,
                       const void* abs_shift = nullptr, float need_ratio = 0.f) {
    if (skipped()) return;
    EmuCholStatus* st = (EmuCholStatus*)st_dev + slot;
    std::vector<double> a((size_t)r * r);
    double dv = 0.0, gm = 0.0;
    bool finite = true;
    for (int64_t j = 0; j < r; ++j)
      for (int64_t i = 0; i < r; ++i) {
        const double v = (double)g.p[j * g.ld + i];
        a[j * r + i] = v;
        finite = finite && std::isfinite(v);
        dv = std::max(dv, std::fabs(v - (i == j ? 1.0 : 0.0)));
        if (i == j) gm = std::max(gm, v);
      }
    std::memset(m_out.p, 0, (size_t)m_out.ld * m_out.cols_alloc * sizeof(T));
    st->dev_i = (float)dv;
    st->gmax = (float)gm;
    st->min_ratio = 1.f;
    st->fail = 0;
    if (!finite) {
      st->fail = 3;
      *need_next = 1 | kFlagNonFinite;
      return;
    }
    if (!(gm > 0.0)) {  // the zero matrix: every column is null
      for (int64_t j = 0; j < r; ++j) null_mask[j] = 1;
      *need_next = 1 | kFlagNullCols;
      return;
    }
    if (dv <= (sizeof(T) == 4 ? 2.0e-4 : 1.0e-8)) {
      for (int64_t j = 0; j < r; ++j) {
        null_mask[j] = 0;
        for (int64_t i = 0; i < r; ++i)
          m_out.p[j * m_out.ld + i] = (T)((i == j ? 1.0 : 0.0) - 0.5 * (a[j * r + i] - (i == j ? 1.0 : 0.0)));
      }
      *need_next = 0;
      return;
    }
    std::vector<double> diag0((size_t)r);
    for (int64_t j = 0; j < r; ++j) diag0[j] = a[j * r + j];
    // blocked form: the shift is in the diagonal already (gram_inspect) and only feeds the null test
    const double pre_sh = abs_shift ? (double)*(const T*)abs_shift : 0.0;
    const bool shifted = abs_shift ? pre_sh > 0.0 : (shift_rel > 0.f && (shift_mode == 1 || (shift_mode == 0 && dv > 0.25)));
    const double sh = abs_shift ? pre_sh : (shifted ? (double)shift_rel * gm : 0.0);
    if (!abs_shift)
      for (int64_t j = 0; j < r; ++j) a[j * r + j] += sh;
    // upper factor R (column-major: R(p, k) at rr[k * r + p]) over the surviving index set
    std::vector<double> rr((size_t)r * r, 0.0);
    int64_t nnull = 0;
    double mr = 1.0;
    for (int64_t j = 0; j < r; ++j) {
      double d = a[j * r + j];
      for (int64_t p = 0; p < j; ++p) d -= rr[j * r + p] * rr[j * r + p];
      if (!(d > (double)piv_rel * diag0[j]) || !(diag0[j] > 0.0) || (shifted && null_excess > 0.f && d - sh <= (double)null_excess * sh)) {
        null_mask[j] = 1;
        ++nnull;
        for (int64_t p = 0; p < j; ++p) rr[j * r + p] = 0.0;  // the column takes no part in anything
        continue;
      }
      null_mask[j] = 0;
      mr = std::min(mr, d / diag0[j]);
      const double rjj = std::sqrt(d);
      rr[j * r + j] = rjj;
      for (int64_t k = j + 1; k < r; ++k) {
        double v = a[k * r + j];
        for (int64_t p = 0; p < j; ++p) v -= rr[j * r + p] * rr[k * r + p];
        rr[k * r + j] = v / rjj;
      }
    }
    // R^-1 on the surviving set (back substitution column by column); null rows and columns stay zero
    std::vector<double> inv((size_t)r * r, 0.0);
    for (int64_t c = 0; c < r; ++c) {
      if (null_mask[c]) continue;
      inv[c * r + c] = 1.0 / rr[c * r + c];
      for (int64_t i = c - 1; i >= 0; --i) {
        if (null_mask[i]) continue;
        double v = 0.0;
        for (int64_t p = i + 1; p <= c; ++p)
          if (!null_mask[p]) v += rr[p * r + i] * inv[c * r + p];
        inv[c * r + i] = -v / rr[i * r + i];
      }
    }
    for (int64_t c = 0; c < r; ++c)
      for (int64_t i = 0; i <= c; ++i) m_out.p[c * m_out.ld + i] = (T)inv[c * r + i];
    st->min_ratio = nnull > 0 ? 0.f : (float)mr;
    if (need_ratio > 0.f)
      *need_next = (nnull > 0 || mr < (double)need_ratio) ? 1 : 0;
    else
      *need_next = (nnull > 0 || dv > 0.05 || (shifted && !abs_shift)) ? 1 : 0;
    if (nnull > 0) *need_next |= kFlagNullCols;
  }
  template <class T>
  void apply_inplace(Skinny<T>& y, int64_t l, const Skinny<T>& m) {
    if (skipped()) return;
    std::vector<double> row((size_t)l);
    const ColBlocking cb = col_blocking(l);
    for (int64_t i = 0; i < y.rows; ++i) {
      for (int64_t c = 0; c < l; ++c) row[c] = (double)y.p[c * y.ld + i];
      for (int64_t c = 0; c < cb.cols_alloc; ++c) {
        double s = 0.0;
        if (c < l)
          for (int64_t kk = 0; kk < l; ++kk) s += row[kk] * (double)m.p[c * m.ld + kk];
        y.p[c * y.ld + i] = (T)s;
      }
    }
  }
  template <class T>
  void refill_null(Skinny<T>& y, int64_t l, const int* null_mask, uint64_t seed) {
    if (skipped()) return;
    const T sc = (T)(1.0 / std::sqrt((double)y.rows));
    for (int64_t j = 0; j < l; ++j)
      if (null_mask[j])
        for (int64_t i = 0; i < y.rows; ++i) y.p[j * y.ld + i] = sc * normal_from_index<T>((uint64_t)(i * l + j), seed);
  }

  void read_chol_status(const void* st_dev, int n, int* fail, float* min_ratio, float* dev_i) {
    const EmuCholStatus* st = (const EmuCholStatus*)st_dev;
    for (int i = 0; i < n; ++i) {
      fail[i] = st[i].fail;
      min_ratio[i] = st[i].min_ratio;
      dev_i[i] = st[i].dev_i;
    }
  }
  template <class T>
  void inv_sqrt_series(Skinny<T>& g, int64_t r, Skinny<T>& m_out) {
    std::vector<double> e((size_t)r * r), e2((size_t)r * r, 0.0), e3((size_t)r * r, 0.0);
    for (int64_t j = 0; j < r; ++j)
      for (int64_t i = 0; i < r; ++i) e[j * r + i] = (double)g.p[j * g.ld + i] - (i == j ? 1.0 : 0.0);
    for (int64_t j = 0; j < r; ++j)
      for (int64_t i = 0; i < r; ++i)
        for (int64_t kk = 0; kk < r; ++kk) e2[j * r + i] += e[kk * r + i] * e[j * r + kk];
    for (int64_t j = 0; j < r; ++j)
      for (int64_t i = 0; i < r; ++i)
        for (int64_t kk = 0; kk < r; ++kk) e3[j * r + i] += e[kk * r + i] * e2[j * r + kk];
    std::memset(m_out.p, 0, (size_t)m_out.ld * m_out.cols_alloc * sizeof(T));
    for (int64_t j = 0; j < r; ++j)
      for (int64_t i = 0; i < r; ++i)
        m_out.p[j * m_out.ld + i] = (T)((i == j ? 1.0 : 0.0) - 0.5 * e[j * r + i] + 0.375 * e2[j * r + i] - 0.3125 * e3[j * r + i]);
  }
  template <class T>
  void copy_values_out(const T* src, int64_t n, T* dst, bool) { std::memcpy(dst, src, sizeof(T) * n); }
  template <class T>
  void small_svd(const Skinny<T>& c, int64_t l, int64_t k, Skinny<T>& m1, Skinny<T>& m2, T* s_dev, void* conv_status) {
    if (conv_status) std::memset(conv_status, 0, sizeof(EmuCholStatus));
    small_svd_host(*this, c, l, k, m1, m2, s_dev);
  }
  template <class T>
  void copy_out(const Skinny<T>& src, int64_t ncols, T* dst, int64_t ldd, bool transpose, bool) {
    for (int64_t c = 0; c < ncols; ++c)
      for (int64_t r = 0; r < src.rows; ++r) {
        if (transpose)
          dst[r * ldd + c] = src.p[c * src.ld + r];
        else
          dst[c * ldd + r] = src.p[c * src.ld + r];
      }
  }
  template <class T>
  void pack_strided(const T* src, int64_t rows, int64_t cols, int64_t rs, int64_t cs, T* dst, int64_t ldd) {
    for (int64_t r = 0; r < rows; ++r)
      for (int64_t c = 0; c < cols; ++c) dst[r * ldd + c] = src[r * rs + c * cs];
  }
  template <class T>
  void fix_signs(Skinny<T>& v_ref, Skinny<T>& other, int64_t k) {
    for (int64_t j = 0; j < k; ++j) {
      T best = (T)-1;
      int64_t bi = 0;
      for (int64_t i = 0; i < v_ref.rows; ++i) {
        const T a = std::fabs(v_ref.p[j * v_ref.ld + i]);
        if (a > best) {
          best = a;
          bi = i;
        }
      }
      if (best > (T)0 && v_ref.p[j * v_ref.ld + bi] < (T)0) {
        for (int64_t i = 0; i < v_ref.rows; ++i) v_ref.p[j * v_ref.ld + i] = -v_ref.p[j * v_ref.ld + i];
        for (int64_t i = 0; i < other.rows; ++i) other.p[j * other.ld + i] = -other.p[j * other.ld + i];
      }
    }
  }
  template <class T>
  void fill_const(T* p, int64_t n, T v) {
    for (int64_t i = 0; i < n; ++i) p[i] = v;
  }
  template <class T>
  void center_rows_cols(const T* in, int64_t rows, int64_t cols, int64_t ldi, const T* mu, bool along_cols, T* out,
                        int64_t ldo) {
    for (int64_t r = 0; r < rows; ++r)
      for (int64_t c = 0; c < cols; ++c) out[r * ldo + c] = in[r * ldi + c] - (along_cols ? mu[c] : mu[r]);
  }
  // v[c] = sum_{r < rows} (w ? w[r] : 1) x(r, c) for every allocated column of x
  template <class T>
  void weighted_colsum(const Skinny<T>& x, int64_t rows, const T* w, T* v) {
    for (int64_t c = 0; c < x.cols_alloc; ++c) {
      double s = 0.0;
      for (int64_t r = 0; r < rows; ++r) s += (w ? (double)w[r] : 1.0) * (double)x.p[c * x.ld + r];
      v[c] = (T)s;
    }
  }
  // out(i, c) -= scale * (u ? u[i] : 1) * v[c]
  template <class T>
  void rank1_sub(Skinny<T>& out, int64_t rows, const T* u, const T* v, const T* scale) {
    const T sc = scale ? *scale : (T)1;
    for (int64_t c = 0; c < out.cols_alloc; ++c)
      for (int64_t r = 0; r < rows; ++r) out.p[c * out.ld + r] -= sc * (u ? u[r] : (T)1) * v[c];
  }
  template <class T>
  void sumsq(const Skinny<T>& y, double* out) {
    double s = 0.0;
    for (int64_t i = 0; i < y.ld * y.cols_alloc; ++i) s += (double)y.p[i] * (double)y.p[i];
    *out = s;
  }
  template <class T>
  void rsqrt_scalar(const double* ss, T* out) { *out = (T)(*ss > 0.0 ? 1.0 / std::sqrt(*ss) : 0.0); }
  template <class T>
  void inv_norm(const Skinny<T>& y, double* ss, T* inv) {
    sumsq(y, ss);
    rsqrt_scalar(ss, inv);
  }
  template <class T>
  void scale_inplace(Skinny<T>& y, const T* sc) {
    for (int64_t i = 0; i < y.ld * y.cols_alloc; ++i) y.p[i] *= *sc;
  }
  template <class T>
  void fill_normal(T* p, int64_t rows, int64_t cols, int64_t rs, int64_t cs, uint64_t seed, int64_t row0, int64_t gcols) {
    for (int64_t i = 0; i < rows; ++i)
      for (int64_t j = 0; j < cols; ++j) p[i * rs + j * cs] = normal_from_index<T>((uint64_t)((row0 + i) * gcols + j), seed);
  }
};

#define EMU_API __attribute__((visibility("default")))
extern "C" {
EMU_API const char* corrla_emu_last_error(void) { return last_error_slot().c_str(); }
EMU_API void corrla_emu_set_comm(emu_allreduce_fn fn, int nranks) {
  g_allreduce = fn;
  g_nranks = nranks;
}
EMU_API void corrla_emu_set_rank(int rank) { g_rank = rank; }
#define EMU_DEFINE(SUF, T)                                                                                             \
  EMU_API int corrla_emu_rsvd_##SUF(const T* a, int64_t m, int64_t n, int64_t rs, int64_t cs, int64_t rank, int64_t n_iter,    \
                            int64_t p, const corrla_opts* o, T* u, int64_t ldu, T* s, T* vt, int64_t ldvt,             \
                            int* qr_passes) {                                                                          \
    return guarded([&] {                                                                                               \
      EmuDev dev;                                                                                                      \
      Timings tm;                                                                                                      \
      rsvd_entry<EmuDev, T>(dev, true, false, a, m, n, rs, cs, rank, n_iter, p, o, u, ldu, s, vt, ldvt, &tm, false);   \
      if (qr_passes) *qr_passes = tm.qr_passes;                                                                        \
    });                                                                                                                \
  }                                                                                                                    \
  EMU_API int corrla_emu_rsvd_sharded_##SUF(const T* a, int64_t m, int64_t n, int64_t rs, int64_t cs, int64_t rank,            \
                                    int64_t n_iter, int64_t p, const corrla_opts* o, T* u, int64_t ldu, T* s, T* vt,   \
                                    int64_t ldvt) {                                                                    \
    return guarded([&] {                                                                                               \
      EmuDev dev;                                                                                                      \
      rsvd_entry<EmuDev, T>(dev, true, true, a, m, n, rs, cs, rank, n_iter, p, o, u, ldu, s, vt, ldvt, nullptr,        \
                            false);                                                                                    \
    });                                                                                                                \
  }                                                                                                                    \
  EMU_API int corrla_emu_pca_##SUF(const T* a, int64_t m, int64_t n, int64_t rs, int64_t cs, int64_t rank,             \
                                   int64_t n_iter, int64_t p, const corrla_opts* o, T* means, T* s, T* comps,          \
                                   int64_t ldc) {                                                                      \
    return guarded([&] {                                                                                               \
      EmuDev dev;                                                                                                      \
      pca_entry<EmuDev, T>(dev, true, a, m, n, rs, cs, rank, n_iter, p, o, means, s, comps, ldc, nullptr, false);      \
    });                                                                                                                \
  }                                                                                                                    \
  EMU_API int corrla_emu_pca_sharded_##SUF(const T* a, int64_t m, int64_t n, int64_t rs, int64_t cs, int64_t rank,     \
                                           int64_t n_iter, int64_t p, const corrla_opts* o, T* means, T* s, T* comps,  \
                                           int64_t ldc) {                                                              \
    return guarded([&] {                                                                                               \
      EmuDev dev;                                                                                                      \
      pca_entry<EmuDev, T>(dev, true, a, m, n, rs, cs, rank, n_iter, p, o, means, s, comps, ldc, nullptr, false, true); \
    });                                                                                                                \
  }                                                                                                                    \
  EMU_API int corrla_emu_power_iter_##SUF(const T* a, int64_t m, int64_t n, int64_t rs, int64_t cs, int64_t width,             \
                                  int64_t n_iter, const corrla_opts* o, T* q, int64_t ldq) {                           \
    return guarded([&] {                                                                                               \
      EmuDev dev;                                                                                                      \
      power_iter_entry<EmuDev, T>(dev, true, a, m, n, rs, cs, width, n_iter, o, q, ldq);                               \
    });                                                                                                                \
  }                                                                                                                    \
  EMU_API int corrla_emu_matmul_##SUF(int trans, const T* a, int64_t m, int64_t n, int64_t rs, int64_t cs, const T* x,         \
                              int64_t ldx, int64_t l, T beta, T* res, int64_t ldres) {                                 \
    return guarded([&] {                                                                                               \
      EmuDev dev;                                                                                                      \
      matmul_entry<EmuDev, T>(dev, trans, a, m, n, rs, cs, x, ldx, l, beta, res, ldres);                               \
    });                                                                                                                \
  }                                                                                                                    \
  EMU_API int corrla_emu_fill_normal_##SUF(T* p, int64_t rows, int64_t cols, int64_t rs, int64_t cs, uint64_t seed,            \
                                   int64_t row0, int64_t gcols) {                                                      \
    return guarded([&] {                                                                                               \
      EmuDev dev;                                                                                                      \
      dev.fill_normal(p, rows, cols, rs, cs, seed, row0, gcols);                                                       \
    });                                                                                                                \
  }
EMU_DEFINE(f32, float)
EMU_DEFINE(f64, double)

// direct hooks for the host-side small dense routines (product code in small_linalg.hpp)
EMU_API int corrla_emu_chol_upper(int n, double* g, int ld, double piv_rel, double* min_ratio) {
  return small::chol_upper(n, g, ld, piv_rel, min_ratio) ? 1 : 0;
}
EMU_API void corrla_emu_triu_inverse(int n, double* r, int ld) { small::triu_inverse(n, r, ld); }
EMU_API int corrla_emu_jacobi_svd(int n, const double* c, int ld, double* u, double* s, double* v, double tol) {
  return small::jacobi_svd(n, c, ld, u, s, v, tol);
}
}
