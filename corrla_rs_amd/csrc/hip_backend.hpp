// HIP backend of the RSVD driver: one device, one stream, one cached workspace arena and
// (optionally) one RCCL communicator per context.  Implements the `Dev` interface that
// driver.hpp / capi_impl.hpp are written against.  Every operation is enqueued on the context's
// stream; the only host synchronisations are the small l x l downloads the host factorizations need.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <random>
#include <string>
#include <vector>

#include "driver.hpp"
#include "hip_kernels.hpp"
#include "tsqr_kernels.hpp"
#include "jacobi_mc_kernels.hpp"
#include "ata_kernels.hpp"
#include "tall_kernels.hpp"
#include "mixed_kernels.hpp"

namespace corrla {

#define CORRLA_HIP(call)                                                                                       \
  do {                                                                                                         \
    hipError_t e_ = (call);                                                                                    \
    if (e_ != hipSuccess)                                                                                      \
      throw Error(ST_EHIP, std::string(#call) + " failed: " + hipGetErrorString(e_) + " (" __FILE__ ":" +       \
                               std::to_string(__LINE__) + ")");                                                \
  } while (0)
#define CORRLA_NCCL(call)                                                                                      \
  do {                                                                                                         \
    ncclResult_t r_ = (call);                                                                                  \
    if (r_ != ncclSuccess) throw Error(ST_ECOMM, std::string(#call) + " failed: " + ncclGetErrorString(r_));   \
  } while (0)

inline int env_int(const char* name, int dflt) {
  const char* v = std::getenv(name);
  return v && *v ? std::atoi(v) : dflt;
}

template <class T>
struct NcclType;
template <>
struct NcclType<float> {
  static constexpr ncclDataType_t v = ncclFloat;
};
template <>
struct NcclType<double> {
  static constexpr ncclDataType_t v = ncclDouble;
};

class HipDev {
 public:
  int device = 0;
  int num_cus = 256;
  hipStream_t stream = nullptr;
  ncclComm_t comm = nullptr;
  int comm_rank = 0, comm_size = 1;
  int n_collectives = 0;        // since begin_call
  double collective_bytes = 0;

  explicit HipDev(int dev_ordinal) : device(dev_ordinal) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
      throw Error(ST_ENODEV, "no HIP device visible (libcorrla_rsvd has no CPU fallback)");
    if (dev_ordinal < 0 || dev_ordinal >= count) throw Error(ST_EINVAL, "device ordinal out of range");
    CORRLA_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    CORRLA_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
      throw Error(ST_ENODEV, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    num_cus = prop.multiProcessorCount;
    CORRLA_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    CORRLA_HIP(hipMalloc(&zero_page_, 256));
    CORRLA_HIP(hipHostMalloc(&pinned_, kPinnedBytes, hipHostMallocDefault));
    for (auto& e : events_) CORRLA_HIP(hipEventCreate(&e));
    CORRLA_HIP(hipMemsetAsync(zero_page_, 0, 256, stream));
    set_lds_attrs<float>();
    set_lds_attrs<double>();
    set_jacobi_attrs<float>();
    set_jacobi_attrs<double>();
    set_ring_attrs<float, 20>();
    set_ring_attrs<double, 18>();
    set_tsqr_attrs<float>();
    set_tsqr_attrs<double>();
    set_jmc_attrs<float>();
    set_jmc_attrs<double>();
    ata_set_attrs<2>();
    ata_set_attrs<4>();
    ata_set_attrs<8>();
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jacobi_svd_split_kernel<float>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jacobi_svd_split_kernel<double>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jacobi_block_round_kernel<float>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jacobi_block_round_kernel<double>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    {
      std::random_device rd;
      entropy_ = ((uint64_t)rd() << 32) ^ (uint64_t)rd();
    }
    split_nn_override_ = env_int("CORRLA_SPLIT_NN", 0);
    split_tn_override_ = env_int("CORRLA_SPLIT_TN", 0);
    mw_override_ = env_int("CORRLA_MW", 0);
    no_device_chol_ = env_int("CORRLA_HOST_CHOL", 0) != 0;
    jmc_min_l_ = env_int("CORRLA_JMC_MIN_L", 96);  // below: the single-workgroup ring kernel + replay is as fast (one launch)
    jmc_max_b_ = std::min(32, std::max(2, env_int("CORRLA_JMC_MAX_B", 24)));
    gemm_xcd_remap_ = env_int("CORRLA_GEMM_XCD", 1);
    tall_min_rows_ = env_int("CORRLA_TALL_MIN_ROWS", 65536);  // 0: the general kernels everywhere
    robust_passes_ = std::max(2, env_int("CORRLA_ROBUST_PASSES", 2));
    robust_qr_ = env_int("CORRLA_DEVICE_ROBUST_QR", 1) != 0;  // 0: the round-1 optimistic CholeskyQR2 + host-controlled repeat
    persist_max_tiles_ = env_int("CORRLA_GEMM_PERSIST_TILES", 16);  // 0: one workgroup per outer tile everywhere
    gemm_debug_flags_ = env_int("CORRLA_GEMM_DEBUG", 0);  // timing-only ablations, results are wrong
    if (const char* e = std::getenv("CORRLA_MIXED_MIN_WORK")) mixed_min_work_ = std::atof(e);
    gemm_wide_mode_ = env_int("CORRLA_GEMM_WIDE", gemm_wide_mode_);
    f64_mfma_waves_ = env_int("CORRLA_F64_WAVES", f64_mfma_waves_);
    gemm_wide_max_red_ = env_int("CORRLA_GEMM_WIDE_MAX_RED", (int)gemm_wide_max_red_);
  }
  ~HipDev() {
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    if (comm) (void)ncclCommDestroy(comm);
    for (auto& c : chunks_) (void)hipFree(c.p);
    for (auto& c : zchunks_) (void)hipFree(c.p);
    if (zero_page_) (void)hipFree(zero_page_);
    if (pinned_) (void)hipHostFree(pinned_);
    for (auto& e : events_)
      if (e) (void)hipEventDestroy(e);
    for (auto& e : ev_pool_) (void)hipEventDestroy(e);
    if (stream) (void)hipStreamDestroy(stream);
  }
  HipDev(const HipDev&) = delete;
  HipDev& operator=(const HipDev&) = delete;

  int nranks() const { return comm_size; }

  // Seed of a call that names none: the reference draws every sketch from an unseeded thread_rng
  // (mat_utils.rs:161-175), so repeated calls must not share one Omega.  Per-context entropy (taken once, at
  // creation) mixed with a call counter; rank_invariant (row-sharded calls: every rank must draw the SAME Omega)
  // leaves the entropy out, so ranks that make the same sequence of calls agree.
  uint64_t fresh_seed(bool rank_invariant) {
    uint64_t z = (rank_invariant ? 0x5eedull : entropy_) + 0x9e3779b97f4a7c15ull * (uint64_t)(++(rank_invariant ? calls_sharded_ : calls_));
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;   // splitmix64 finaliser
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    z ^= z >> 31;
    return z ? z : 0x5eedull;
  }

  void comm_init(const void* id128, int rank, int nranks_) {
    CORRLA_HIP(hipSetDevice(device));
    ncclUniqueId id;
    static_assert(sizeof(ncclUniqueId) <= CORRLA_UNIQUE_ID_BYTES, "unique id size");
    std::memcpy(&id, id128, sizeof(id));
    if (comm) {
      (void)ncclCommDestroy(comm);
      comm = nullptr;
    }
    CORRLA_NCCL(ncclCommInitRank(&comm, nranks_, id, rank));
    comm_rank = rank;
    comm_size = nranks_;
  }
  // what RCCL itself reports for the communicator (not what the caller asked for)
  void comm_info(int* rank_out, int* nranks_out) {
    if (!comm) {
      *rank_out = 0;
      *nranks_out = 0;
      return;
    }
    CORRLA_NCCL(ncclCommUserRank(comm, rank_out));
    CORRLA_NCCL(ncclCommCount(comm, nranks_out));
  }

  // ---- memory ----------------------------------------------------------------------------
  void begin_call() {
    CORRLA_HIP(hipSetDevice(device));
    events_set_[0] = events_set_[1] = events_set_[2] = false;
    n_collectives = 0;
    collective_bytes = 0;
    marks_.clear();
    ev_used_ = 0;
    phase_mark(nullptr);  // start of the call
    for (auto& c : chunks_) c.used = 0;
    // zero pool: the part the previous call used is cleared by ONE memset per chunk (capped), instead of one
    // small memset per workspace allocation
    for (auto& c : zchunks_) {
      c.zeroed = std::min<size_t>(c.used, (size_t)256 << 20);
      c.used = 0;
      if (c.zeroed) CORRLA_HIP(hipMemsetAsync(c.p, 0, c.zeroed, stream));
    }
  }
  void end_call() { sync(); }
  void sync() { CORRLA_HIP(hipStreamSynchronize(stream)); }

  void* alloc_bytes(size_t bytes) {
    bytes = (bytes + 255) / 256 * 256;
    if (bytes == 0) bytes = 256;
    for (auto& c : chunks_)
      if (c.size - c.used >= bytes) {
        void* p = (char*)c.p + c.used;
        c.used += bytes;
        return p;
      }
    Chunk c;
    c.size = std::max<size_t>(bytes, (size_t)64 << 20);
    if (hipMalloc(&c.p, c.size) != hipSuccess) {
      (void)hipGetLastError();
      throw Error(ST_ENOMEM, "device allocation of " + std::to_string(c.size) + " bytes failed");
    }
    c.used = bytes;
    chunks_.push_back(c);
    return c.p;
  }
  void memset_zero(void* p, size_t bytes) { CORRLA_HIP(hipMemsetAsync(p, 0, bytes, stream)); }
  // zero-filled workspace from the zero pool
  void* alloc_zeroed(size_t bytes) {
    bytes = (bytes + 255) / 256 * 256;
    if (bytes == 0) bytes = 256;
    Chunk* hit = nullptr;
    for (auto& c : zchunks_)
      if (c.size - c.used >= bytes) {
        hit = &c;
        break;
      }
    if (!hit) {
      Chunk c;
      c.size = std::max<size_t>(bytes, (size_t)128 << 20);
      if (hipMalloc(&c.p, c.size) != hipSuccess) {
        (void)hipGetLastError();
        throw Error(ST_ENOMEM, "device allocation of " + std::to_string(c.size) + " bytes failed");
      }
      zchunks_.push_back(c);
      hit = &zchunks_.back();
    }
    char* p = (char*)hit->p + hit->used;
    const size_t end = hit->used + bytes;
    if (end > hit->zeroed) {  // beyond what begin_call cleared
      const size_t from = std::max(hit->used, hit->zeroed);
      CORRLA_HIP(hipMemsetAsync((char*)hit->p + from, 0, end - from, stream));
    }
    hit->used = end;
    return p;
  }

  template <class T>
  Skinny<T> alloc_skinny(int64_t rows, int64_t cols) {
    Skinny<T> s;
    s.rows = rows;
    s.cols = cols;
    s.ld = round_up(std::max<int64_t>(rows, 1), kLdPad);
    s.cols_alloc = col_blocking(cols).cols_alloc;
    const size_t bytes = (size_t)s.ld * (size_t)s.cols_alloc * sizeof(T);
    s.p = (T*)alloc_zeroed(bytes);
    return s;
  }
  // a skinny matrix that a product is about to overwrite completely (every allocated column, rows [0, rows)): only
  // the padding rows [rows, ld) are cleared instead of the whole buffer (four m x l work matrices of a 10^7-row call
  // are 12.8 GB of memset otherwise)
  template <class T>
  Skinny<T> alloc_skinny_out(int64_t rows, int64_t cols) {
    Skinny<T> s;
    s.rows = rows;
    s.cols = cols;
    s.ld = round_up(std::max<int64_t>(rows, 1), kLdPad);
    s.cols_alloc = col_blocking(cols).cols_alloc;
    const size_t bytes = (size_t)s.ld * (size_t)s.cols_alloc * sizeof(T);
    if (bytes < ((size_t)4 << 20)) {
      s.p = (T*)alloc_zeroed(bytes);
      return s;
    }
    s.p = (T*)alloc_bytes(bytes);
    if (s.ld > rows)
      CORRLA_HIP(hipMemset2DAsync(s.p + rows, (size_t)s.ld * sizeof(T), 0, (size_t)(s.ld - rows) * sizeof(T), (size_t)s.cols_alloc,
                                  stream));
    // the padding columns too: an uneven column blocking does not write its all-zero tail tile
    if (s.cols_alloc > cols) memset_zero(s.p + cols * s.ld, (size_t)(s.cols_alloc - cols) * s.ld * sizeof(T));
    return s;
  }
  double* alloc_f64(int n) {
    return (double*)alloc_zeroed(sizeof(double) * n);
  }
  template <class T>
  T* alloc_scalar(int n) {
    return (T*)alloc_zeroed(sizeof(T) * n);
  }

  void h2d_2d(void* dst, int64_t dpitch_e, const void* src, int64_t spitch_e, int64_t width_e, int64_t rows, size_t esz) {
    CORRLA_HIP(hipMemcpy2DAsync(dst, (size_t)dpitch_e * esz, src, (size_t)spitch_e * esz, (size_t)width_e * esz,
                                (size_t)rows, hipMemcpyHostToDevice, stream));
    sync();
  }
  template <class T>
  void h2d_2d(T* dst, int64_t dpitch_e, const T* src, int64_t spitch_e, int64_t width_e, int64_t rows) {
    h2d_2d((void*)dst, dpitch_e, (const void*)src, spitch_e, width_e, rows, sizeof(T));
  }

  // ---- GEMMs -----------------------------------------------------------------------------
  template <class T>
  void gemm_nn(const Big<T>& r, const Skinny<T>& x, Skinny<T>& out, const T* scale_dev) {
    if (x.rows != r.cols) throw Error(ST_EINVAL, "gemm_nn: inner dimensions differ");
    if (wide_exact_wanted<T>(false, r, x, out)) return gemm_mixed<T>(false, r, x, out, scale_dev, 0);
    launch_gemm<T>(false, r, x, out, scale_dev, r.rows, r.cols);
  }
  template <class T>
  void gemm_tn(const Big<T>& r, const Skinny<T>& x, Skinny<T>& out, const T* scale_dev) {
    if (x.rows != r.rows) throw Error(ST_EINVAL, "gemm_tn: inner dimensions differ");
    if (wide_exact_wanted<T>(true, r, x, out)) return gemm_mixed<T>(true, r, x, out, scale_dev, 0);
    launch_gemm<T>(true, r, x, out, scale_dev, r.cols, r.rows);
  }

  // ---- bf16-split tall products (mixed_kernels.hpp): f32 operands, bf16 MFMA, f32 accumulate ----------
  // np = 2 ("bf16x3": hi hi + hi lo + lo hi) or 3 ("bf16x6").  Serves row-major f32 big operands against one column
  // block (<= 144 columns) when the product is large enough to be worth the extra launches; everything else keeps the
  // exact f32 kernels (the caller asks mixed_fits first).
  template <class T>
  bool mixed_fits(bool tn, const Big<T>& r, const Skinny<T>& x, const Skinny<T>& out) const {
    if constexpr (!std::is_same<T, float>::value) {
      return false;
    } else {
      const int64_t outer_n = tn ? r.cols : r.rows, red_n = tn ? r.rows : r.cols;
      if (x.external || col_blocking(x.cols).nblk != 1) return false;
      if (((uintptr_t)r.p % 16) || (r.ld % 4) || (r.cols_readable % 4) || ((uintptr_t)x.p % 16) || (x.ld % 64)) return false;
      if (x.ld < round_up(red_n, k::kMxKT) || out.ld < outer_n || out.rows != outer_n) return false;
      if ((const void*)r.p == (const void*)x.p) return false;  // Gram products stay exact
      // (CORRLA_MIXED_MIN_WORK: tests drive small shapes through the kernels)
      return outer_n >= 1 && red_n >= 1 && (double)outer_n * (double)red_n >= (double)mixed_min_work_;
    }
  }
  // np = 0 runs the same 8-wave / 256-outer-index skeleton with EXACT f32 MFMAs (mixed_kernels.hpp): the skinny operand is
  // restaged half as often as in the 4-wave kernels of gemm_kernels.hpp, which is what bounds the short-reduction
  // products (CORRLA_GEMM_WIDE: 0 = never, 1 = whenever the operands fit, 2 = by shape; see wide_exact_wanted).
  template <class T>
  bool wide_exact_wanted(bool tn, const Big<T>& r, const Skinny<T>& x, const Skinny<T>& out) const {
    if constexpr (!std::is_same<T, float>::value) {
      return false;
    } else {
      if (gemm_wide_mode_ == 0 || !mixed_fits<float>(tn, r, x, out)) return false;
      if (gemm_wide_mode_ == 1) return true;
      const int64_t red_n = tn ? r.rows : r.cols;
      return red_n <= gemm_wide_max_red_;
    }
  }
  template <int NT, int NP, bool TN>
  void mixed_launch_one(dim3 grid, const k::MxArgs& g) {
    static bool attr_set = false;  // per instantiation; contexts are created under a process-wide lock
    const int lds = k::mx_lds_bytes(NT, NP);
    if (!attr_set) {
      CORRLA_HIP(hipFuncSetAttribute((const void*)k::gemm_bf16s_kernel<NT, NP, TN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      attr_set = true;
    }
    hipLaunchKernelGGL((k::gemm_bf16s_kernel<NT, NP, TN>), grid, dim3(64 * (k::kMxWaves + k::kMxLoaders)), lds, stream, g);
  }
  template <int NP, bool TN>
  void mixed_launch_nt(int nt, dim3 grid, const k::MxArgs& g) {
    switch (nt) {
      case 1: mixed_launch_one<1, NP, TN>(grid, g); break;
      case 2: mixed_launch_one<2, NP, TN>(grid, g); break;
      case 3: mixed_launch_one<3, NP, TN>(grid, g); break;
      case 4: mixed_launch_one<4, NP, TN>(grid, g); break;
      case 5: mixed_launch_one<5, NP, TN>(grid, g); break;
      case 6: mixed_launch_one<6, NP, TN>(grid, g); break;
      case 7: mixed_launch_one<7, NP, TN>(grid, g); break;
      case 8: mixed_launch_one<8, NP, TN>(grid, g); break;
      default: mixed_launch_one<9, NP, TN>(grid, g); break;
    }
  }
  template <class T>
  void gemm_mixed(bool tn, const Big<T>& r, const Skinny<T>& x, Skinny<T>& out, const T* scale_dev, int np) {
    if constexpr (std::is_same<T, float>::value)
      gemm_mixed_f32(tn, r, x, out, scale_dev, np);
    else
      throw Error(ST_EINVAL, "internal: the bf16-split products are f32 only");
  }
  void gemm_mixed_f32(bool tn, const Big<float>& r, const Skinny<float>& x, Skinny<float>& out, const float* scale_dev, int np) {
    if (np != 0 && np != 2 && np != 3) throw Error(ST_EINVAL, "internal: bf16 split takes 2 or 3 planes (0 = exact f32)");
    if (!mixed_fits<float>(tn, r, x, out)) throw Error(ST_EINVAL, "internal: operands outside the bf16-split kernels' domain");
    const int64_t outer_n = tn ? r.cols : r.rows, red_n = tn ? r.rows : r.cols;
    const ColBlocking cb = col_blocking(x.cols);
    if (cb.cols_alloc > x.cols_alloc || (!out.external && cb.cols_alloc > out.cols_alloc) || (out.external && out.cols < x.cols))
      throw Error(ST_EINVAL, "internal: skinny column padding too small for the column blocking");
    // the skinny operand in np bf16 planes, reduction index in MFMA fragment order
    const int64_t plane_stride = x.ld * cb.cols_alloc;
    __bf16* planes = np ? (__bf16*)alloc_bytes((size_t)np * (size_t)plane_stride * 2) : (__bf16*)x.p;  // exact: X as it is
    if (np) {
      const int64_t slots = plane_stride / 8;
      const dim3 sg((unsigned)std::max<int64_t>(1, std::min<int64_t>(4096, (slots + 255) / 256)));
      if (np == 3)
        hipLaunchKernelGGL((k::split_planes_kernel<3>), sg, dim3(256), 0, stream, (const float*)x.p, x.ld, cb.cols_alloc, planes, plane_stride, run_if_);
      else
        hipLaunchKernelGGL((k::split_planes_kernel<2>), sg, dim3(256), 0, stream, (const float*)x.p, x.ld, cb.cols_alloc, planes, plane_stride, run_if_);
      CORRLA_HIP(hipGetLastError());
    }
    const int64_t tiles64 = (red_n + k::kMxKT - 1) / k::kMxKT;
    if (tiles64 > 0x7fffffff) throw Error(ST_EINVAL, "reduction dimension too large");
    const int tiles_total = (int)tiles64;
    const int64_t outer_tiles = (outer_n + k::kMxOuter - 1) / k::kMxOuter;
    if (outer_tiles > 0x7fffffff) throw Error(ST_EINVAL, "outer dimension too large");
    // one workgroup per CU (120-155 KB of LDS): split the reduction until the grid fills the chip
    int nsplit = 1;
    if (outer_tiles < num_cus) nsplit = (int)std::min<int64_t>((num_cus + outer_tiles / 2) / outer_tiles, std::max(1, tiles_total / 16));
    if (const int ov = env_int("CORRLA_MIXED_SPLIT", 0)) nsplit = ov;
    nsplit = std::max(1, std::min(std::min(nsplit, tiles_total), 65535));
    k::MxArgs a;
    a.r = r.p;
    a.r_rows = r.rows;
    a.r_cols = r.cols;
    a.r_ld = r.ld;
    a.r_cols_readable = r.cols_readable;
    a.planes = planes;
    a.x_ld = x.ld;
    a.plane_stride = plane_stride;
    a.out = out.p;
    a.out_ld = out.ld;
    a.out_cols = out.external ? out.cols : cb.cols_alloc;
    a.slab = nullptr;
    a.slab_stride = (int64_t)out.ld * cb.cols_alloc;
    if (nsplit > 1) a.slab = (float*)alloc_bytes((size_t)nsplit * (size_t)a.slab_stride * sizeof(float));
    a.scale = scale_dev;
    a.zero = (const float*)zero_page_;
    a.tiles_total = tiles_total;
    a.tiles_per_split = (tiles_total + nsplit - 1) / nsplit;
    a.nsplit = nsplit;
    a.run_if = run_if_;
    a.vec_store = ((out.ld % 4) == 0 && ((uintptr_t)out.p % 16) == 0) ? 1 : 0;
    a.debug_flags = gemm_debug_flags_;
    const dim3 grid((unsigned)outer_tiles, 1, (unsigned)nsplit);
    check_grid(grid);
    if (np == 0) {
      if (tn) mixed_launch_nt<0, true>(cb.nt, grid, a); else mixed_launch_nt<0, false>(cb.nt, grid, a);
    } else if (np == 3) {
      if (tn) mixed_launch_nt<3, true>(cb.nt, grid, a); else mixed_launch_nt<3, false>(cb.nt, grid, a);
    } else {
      if (tn) mixed_launch_nt<2, true>(cb.nt, grid, a); else mixed_launch_nt<2, false>(cb.nt, grid, a);
    }
    CORRLA_HIP(hipGetLastError());
    if (nsplit >= 8) {
      dim3 rg((unsigned)((outer_n + 63) / 64), (unsigned)cb.cols_alloc);
      check_grid(rg);
      hipLaunchKernelGGL((k::slab_reduce_deep_kernel<float>), rg, dim3(256), 0, stream, (const float*)a.slab, a.slab_stride, nsplit,
                         out.p, out.ld, outer_n, a.out_cols, scale_dev, run_if_);
      CORRLA_HIP(hipGetLastError());
    } else if (nsplit > 1) {
      dim3 rg((unsigned)((outer_n + 255) / 256), (unsigned)cb.cols_alloc);
      check_grid(rg);
      hipLaunchKernelGGL((k::slab_reduce_kernel<float>), rg, dim3(256), 0, stream, (const float*)a.slab, a.slab_stride, nsplit,
                         out.p, out.ld, outer_n, a.out_cols, scale_dev, run_if_);
      CORRLA_HIP(hipGetLastError());
    }
  }

  // ---- one-sweep Z' = A^T (A Z) (ata_kernels.hpp): row-major f32 A with n <= 512, l <= 80 -----------
  template <class T>
  bool ata_fused_fits(const Big<T>& a, int64_t l) const {
    return std::is_same<T, float>::value && a.cols <= 512 && a.cols >= 16 && l <= 80 && a.rows >= 4096;
  }
  template <int NK>
  void ata_launch_nct(int nct, dim3 grid, const k::AtaArgs& g) {
    const dim3 block(256);
    switch (nct) {
      case 1: hipLaunchKernelGGL((k::ata_fused_kernel<NK, 1>), grid, block, k::ata_lds_bytes(NK, 1), stream, g); break;
      case 2: hipLaunchKernelGGL((k::ata_fused_kernel<NK, 2>), grid, block, k::ata_lds_bytes(NK, 2), stream, g); break;
      case 3: hipLaunchKernelGGL((k::ata_fused_kernel<NK, 3>), grid, block, k::ata_lds_bytes(NK, 3), stream, g); break;
      case 4: hipLaunchKernelGGL((k::ata_fused_kernel<NK, 4>), grid, block, k::ata_lds_bytes(NK, 4), stream, g); break;
      default: hipLaunchKernelGGL((k::ata_fused_kernel<NK, 5>), grid, block, k::ata_lds_bytes(NK, 5), stream, g); break;
    }
  }
  template <int NK>
  void ata_set_attrs() {
    const hipFuncAttribute attr = hipFuncAttributeMaxDynamicSharedMemorySize;
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::ata_fused_kernel<NK, 1>, attr, k::ata_lds_bytes(NK, 1)));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::ata_fused_kernel<NK, 2>, attr, k::ata_lds_bytes(NK, 2)));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::ata_fused_kernel<NK, 3>, attr, k::ata_lds_bytes(NK, 3)));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::ata_fused_kernel<NK, 4>, attr, k::ata_lds_bytes(NK, 4)));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::ata_fused_kernel<NK, 5>, attr, k::ata_lds_bytes(NK, 5)));
  }
  template <class T>
  void ata_fused(const Big<T>& a, const Skinny<T>& x, Skinny<T>& z) {
    if constexpr (!std::is_same<T, float>::value) {
      throw Error(ST_EINVAL, "internal: the one-sweep power iteration is f32 only");
    } else {
      if (x.rows != a.cols || z.rows != a.cols) throw Error(ST_EINVAL, "ata_fused: shapes");
      const ColBlocking cb = col_blocking(x.cols);
      if (cb.cols_alloc > x.cols_alloc || cb.cols_alloc > z.cols_alloc || x.ld != z.ld || cb.cols_alloc > 80)
        throw Error(ST_EINVAL, "internal: ata_fused operands must share the padded layout (<= 80 columns)");
      if (((uintptr_t)a.p % 16) || (a.ld % 4) || (a.cols_readable % 4)) throw Error(ST_EINVAL, "internal: operand not vector aligned");
      // reduction segments: n padded to 128 / 256 / 512 (three instantiation families)
      const int nk = a.cols <= 128 ? 2 : (a.cols <= 256 ? 4 : 8);
      const int nct = (int)(cb.cols_alloc / 16);
      // one workgroup per CU; every workgroup gets at least a few tiles
      const int64_t blocks = (a.rows + k::kAtaRows - 1) / k::kAtaRows;
      const int nrg = (int)std::max<int64_t>(1, std::min<int64_t>(num_cus, blocks / 8));
      const int64_t rows_per_group = ((blocks + nrg - 1) / nrg) * k::kAtaRows;
      k::AtaArgs g;
      g.a = a.p;
      g.m = a.rows;
      g.n = a.cols;
      g.lda = a.ld;
      g.n_readable = a.cols_readable;
      g.z = x.p;
      g.z_ld = x.ld;
      g.out_ld = z.ld;
      g.slab_stride = (int64_t)z.ld * cb.cols_alloc;
      g.slab = (float*)alloc_zeroed((size_t)nrg * (size_t)g.slab_stride * sizeof(float));  // empty groups contribute zeros
      g.rows_per_group = rows_per_group;
      g.nrowgroups = nrg;
      g.zero = (const float*)zero_page_;
      const dim3 grid((unsigned)nrg);
      if (nk == 2) ata_launch_nct<2>(nct, grid, g);
      else if (nk == 4) ata_launch_nct<4>(nct, grid, g);
      else ata_launch_nct<8>(nct, grid, g);
      CORRLA_HIP(hipGetLastError());
      dim3 rgd((unsigned)((a.cols + 63) / 64), (unsigned)cb.cols_alloc);
      check_grid(rgd);
      hipLaunchKernelGGL((k::slab_reduce_deep_kernel<float>), rgd, dim3(256), 0, stream, (const float*)g.slab, g.slab_stride, nrg,
                         z.p, z.ld, a.cols, cb.cols_alloc, (const float*)nullptr, (const int*)nullptr);
      CORRLA_HIP(hipGetLastError());
    }
  }

  // ---- collectives (RCCL over xGMI, on the compute stream) ---------------------------------
  template <class T>
  void allreduce(T* p, size_t count) {
    // CORRLA_FORCE_ALLREDUCE=1: issue the collective on a one-rank communicator too (an identity), so that a 1-GPU
    // box exercises the very RCCL calls -- datatype, count, in-place buffer, stream -- the N > 1 ranks make
    if (comm_size <= 1 && !(comm && env_int("CORRLA_FORCE_ALLREDUCE", 0))) return;
    if (!comm) throw Error(ST_ECOMM, "communicator not initialised");
    CORRLA_NCCL(ncclAllReduce(p, p, count, NcclType<T>::v, ncclSum, comm, stream));
    ++n_collectives;
    collective_bytes += (double)count * sizeof(T);
  }
  void allreduce_f64(double* p, size_t count) { allreduce<double>(p, count); }
  // sum of one host integer over the ranks (exact in f64 up to 2^53); synchronises.  Used for rank-invariant
  // decisions that need the global row count of a sharded matrix.
  int64_t allreduce_sum_host(int64_t v) {
    if (comm_size <= 1) return v;
    if (!comm) throw Error(ST_ECOMM, "communicator not initialised");
    double* d = alloc_f64(1);
    double h = (double)v;
    CORRLA_HIP(hipMemcpyAsync(d, &h, sizeof(double), hipMemcpyHostToDevice, stream));
    CORRLA_NCCL(ncclAllReduce(d, d, 1, ncclDouble, ncclSum, comm, stream));
    ++n_collectives;
    collective_bytes += sizeof(double);
    CORRLA_HIP(hipMemcpyAsync(&h, d, sizeof(double), hipMemcpyDeviceToHost, stream));
    sync();
    return (int64_t)(h + 0.5);
  }

  // ---- small transfers -------------------------------------------------------------------
  // device skinny (rows x cols leading block) -> host f64 column-major (ld = rows); synchronises
  template <class T>
  void download_skinny(const Skinny<T>& s, int64_t rows, int64_t cols, double* host) {
    const size_t n = (size_t)rows * cols;
    std::vector<T> heap;
    T* tmp = (T*)pinned_;
    if (n * sizeof(T) > kPinnedBytes) {
      heap.resize(n);
      tmp = heap.data();
    }
    CORRLA_HIP(hipMemcpy2DAsync(tmp, (size_t)rows * sizeof(T), s.p, (size_t)s.ld * sizeof(T), (size_t)rows * sizeof(T),
                                (size_t)cols, hipMemcpyDeviceToHost, stream));
    sync();
    for (size_t i = 0; i < n; ++i) host[i] = (double)tmp[i];
  }
  // host f64 column-major (rows x cols, ld_host) -> device skinny; the whole allocation is
  // rewritten so padding stays zero
  template <class T>
  void upload_skinny(const double* host, int64_t rows, int64_t cols, int64_t ld_host, Skinny<T>& dst) {
    const size_t n = (size_t)dst.ld * dst.cols_alloc;
    std::vector<T> heap;
    T* tmp = (T*)pinned_;
    if (n * sizeof(T) > kPinnedBytes) {
      heap.resize(n);
      tmp = heap.data();
    }
    sync();  // the staging buffer may still feed an earlier copy
    std::memset(tmp, 0, n * sizeof(T));
    for (int64_t j = 0; j < cols; ++j)
      for (int64_t i = 0; i < rows; ++i) tmp[(size_t)j * dst.ld + i] = (T)host[(size_t)j * ld_host + i];
    CORRLA_HIP(hipMemcpyAsync(dst.p, tmp, n * sizeof(T), hipMemcpyHostToDevice, stream));
    if (!heap.empty()) sync();
  }
  template <class T>
  void upload_skinny_native(const T* host, int64_t ld_host, Skinny<T>& dst) {
    h2d_2d(dst.p, dst.ld, host, ld_host, dst.rows, dst.cols);
  }
  template <class T>
  void copy_in_skinny(const T* src_dev, int64_t ld_src, Skinny<T>& dst) {
    CORRLA_HIP(hipMemcpy2DAsync(dst.p, (size_t)dst.ld * sizeof(T), src_dev, (size_t)ld_src * sizeof(T),
                                (size_t)dst.rows * sizeof(T), (size_t)dst.cols, hipMemcpyDeviceToDevice, stream));
  }
  template <class T>
  void copy_skinny(const Skinny<T>& src, Skinny<T>& dst) {
    CORRLA_HIP(hipMemcpyAsync(dst.p, src.p, (size_t)src.ld * src.cols_alloc * sizeof(T), hipMemcpyDeviceToDevice, stream));
  }
  // dst columns [c0, c0 + n) <- src columns [0, n)  (same row count, hence the same leading dimension)
  template <class T>
  void copy_cols(const Skinny<T>& src, Skinny<T>& dst, int64_t c0, int64_t n) {
    if (src.ld != dst.ld) throw Error(ST_EINVAL, "internal: copy_cols needs equal leading dimensions");
    if (n > 0)
      CORRLA_HIP(hipMemcpyAsync(dst.p + c0 * dst.ld, src.p, (size_t)n * src.ld * sizeof(T), hipMemcpyDeviceToDevice, stream));
  }
  // y -= p over the whole (padded) allocation
  template <class T>
  void sub_inplace(Skinny<T>& y, const Skinny<T>& p) {
    const int64_t n = y.ld * std::min(y.cols_alloc, p.cols_alloc);
    const int blocks = (int)std::min<int64_t>(4096, std::max<int64_t>(1, (n + 255) / 256));
    hipLaunchKernelGGL((k::sub_kernel<T>), dim3(blocks), dim3(256), 0, stream, y.p, (const T*)p.p, n);
    CORRLA_HIP(hipGetLastError());
  }
  template <class T>
  void zero_cols(Skinny<T>& s, int64_t c0, int64_t c1) {
    if (c1 > c0) memset_zero(s.p + c0 * s.ld, (size_t)(c1 - c0) * s.ld * sizeof(T));
  }
  template <class T>
  void store_values(const T* host_src, int64_t n, T* dst, bool dst_is_host) {
    if (dst_is_host) {
      std::memcpy(dst, host_src, sizeof(T) * n);
    } else {
      CORRLA_HIP(hipMemcpyAsync(dst, host_src, sizeof(T) * n, hipMemcpyHostToDevice, stream));
      sync();
    }
  }
  template <class T>
  void copy_values_out(const T* src_dev, int64_t n, T* dst, bool dst_is_host) {
    CORRLA_HIP(hipMemcpyAsync(dst, src_dev, sizeof(T) * n, dst_is_host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice,
                              stream));
    if (dst_is_host) sync();
  }
  // ---- device Cholesky + inverse (optimistic CholeskyQR2) ------------------------------------------
  template <class T>
  bool device_chol_fits(int64_t l) const {
    return l <= 4096 && k::chol_inv_fits((int)l, sizeof(T)) && !no_device_chol_;
  }
  template <class T>
  bool device_chol_blocked_fits(int64_t l) const {
    const int64_t n1 = round_up((l + 1) / 2, (int64_t)4);
    return l > 8 && l <= 4096 && k::chol_inv_fits((int)n1, sizeof(T)) && !no_device_chol_;
  }
  // dst(dr0 + i, dc0 + j) <- src(r0 + i, c0 + j), i < rows, j < cols
  template <class T>
  void copy_block(const Skinny<T>& src, int64_t r0, int64_t c0, int64_t rows, int64_t cols, Skinny<T>& dst, int64_t dr0,
                  int64_t dc0) {
    if (rows <= 0 || cols <= 0) return;
    CORRLA_HIP(hipMemcpy2DAsync(dst.p + dc0 * dst.ld + dr0, (size_t)dst.ld * sizeof(T), src.p + c0 * src.ld + r0,
                                (size_t)src.ld * sizeof(T), (size_t)rows * sizeof(T), (size_t)cols, hipMemcpyDeviceToDevice,
                                stream));
  }
  template <class T>
  void chol_inv(const Skinny<T>& g, int64_t r, T piv_rel, Skinny<T>& m_out, void* st_dev, int slot) {
    // m_out comes zero-filled from alloc_skinny and only its upper triangle is ever written
    hipLaunchKernelGGL((k::chol_inv_kernel<T>), dim3(1), dim3(k::chol_inv_threads((int)r)),
                       k::chol_inv_lds_bytes((int)r, sizeof(T)), stream, (const T*)g.p, g.ld, (int)r, piv_rel, m_out.p,
                       m_out.ld, (k::CholStatus*)st_dev + slot);
    CORRLA_HIP(hipGetLastError());
  }
  // ---- device-robust Cholesky-QR (driver.hpp: orthonormalize_device) ----
  // every launch enqueued while a run_if word is set does nothing when that device word is 0
  void set_run_if(const int* p) { run_if_ = p; }
  // passes a device-robust thin-Q enqueues on this context (driver.hpp: orthonormalize_device); grows on demand
  int robust_passes() const { return robust_passes_; }
  void set_robust_passes(int n) { robust_passes_ = n; }
  // the core SVD of a call did not converge within the sweeps enqueued: enqueue 8 more from now on (false: at the cap)
  bool svd_more_sweeps() {
    if (jmc_extra_sweeps_ >= 24) return false;
    jmc_extra_sweeps_ += 8;
    jmc_sweeps_hint_ = 0;
    return true;
  }
  // the W-only shortcut of the block Jacobi failed its verification: accumulate V from now on (false: already does)
  bool svd_force_v() {
    if (jmc_force_v_) return false;
    jmc_force_v_ = true;
    return true;
  }
  // sweeps the last converged core SVD of this context used: the next call enqueues two more than that instead of the
  // default (the sweeps enqueued beyond convergence are launches that only test a flag: 15 x 4.6 us at C2)
  void svd_sweeps_used(int n) { jmc_sweeps_hint_ = std::max(jmc_sweeps_hint_ - 1, n); }
  int* alloc_flags(int n) { return (int*)alloc_zeroed(sizeof(int) * (size_t)std::max(n, 1)); }
  void* alloc_zeroed_bytes(size_t bytes) { return alloc_zeroed(bytes); }
  void read_bytes(const void* dev_p, size_t bytes, void* host) {
    CORRLA_HIP(hipMemcpyAsync(host, dev_p, bytes, hipMemcpyDeviceToHost, stream));
    sync();
  }
  // TEST HOOK (CORRLA_TEST_POISON_CORE): entry (i, j) of a skinny matrix <- NaN (kind 1) / +inf (kind 2)
  template <class T>
  void poison_entry(Skinny<T>& s, int64_t i, int64_t j, int kind) {
    hipLaunchKernelGGL((k::poison_entry_kernel<T>), dim3(1), dim3(1), 0, stream, s.p + j * s.ld + i, kind);
    CORRLA_HIP(hipGetLastError());
  }
  // ---- workspace marks: a repeated attempt of a call (driver.hpp: random_svd_tall) reuses the workspace of the
  // abandoned one instead of growing the arena (an ill-conditioned 10^7 x 80 call held four 3.2 GB buffers per attempt)
  struct ArenaMark {
    std::vector<size_t> used, zused;
  };
  ArenaMark arena_mark() const {
    ArenaMark mk;
    for (const auto& c : chunks_) mk.used.push_back(c.used);
    for (const auto& c : zchunks_) mk.zused.push_back(c.used);
    return mk;
  }
  void arena_rewind(const ArenaMark& mk) {
    for (size_t i = 0; i < chunks_.size(); ++i) chunks_[i].used = i < mk.used.size() ? mk.used[i] : 0;
    for (size_t i = 0; i < zchunks_.size(); ++i) {
      Chunk& c = zchunks_[i];
      const size_t keep = i < mk.zused.size() ? mk.zused[i] : 0;
      // what the abandoned attempt dirtied goes back to the pool as zeros (stream order: after its kernels)
      if (c.used > keep) CORRLA_HIP(hipMemsetAsync((char*)c.p + keep, 0, c.used - keep, stream));
      c.zeroed = std::max(c.zeroed, c.used);
      c.used = keep;
    }
  }
  // ---- start of a sharded call: ONE small all-reduce (max) carries (a) this rank's validation / staging status, so that
  // a rank-local failure ends the call on EVERY rank instead of leaving the peers blocked in the first collective, and
  // (b) the adaptive schedule state of the context (thin-Q passes enqueued, extra Jacobi sweeps, sweep hint, V mode),
  // which decides how many collectives a call enqueues and so must not differ between ranks with different histories.
  // Returns the largest status over the ranks.  Synchronises (the stream is idle at this point of a call).
  int sharded_handshake(int local_status) {
    if (comm_size <= 1 && !(comm && env_int("CORRLA_FORCE_ALLREDUCE", 0))) return local_status;
    if (!comm) throw Error(ST_ECOMM, "communicator not initialised");
    double h[8] = {(double)local_status, (double)robust_passes_, (double)jmc_extra_sweeps_, (double)jmc_sweeps_hint_,
                   jmc_force_v_ ? 1.0 : 0.0, 0.0, 0.0, 0.0};
    double* d = (double*)alloc_bytes(sizeof(h));
    CORRLA_HIP(hipMemcpyAsync(d, h, sizeof(h), hipMemcpyHostToDevice, stream));
    CORRLA_NCCL(ncclAllReduce(d, d, 8, ncclDouble, ncclMax, comm, stream));
    ++n_collectives;
    collective_bytes += sizeof(h);
    CORRLA_HIP(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, stream));
    sync();
    robust_passes_ = (int)h[1];
    jmc_extra_sweeps_ = (int)h[2];
    jmc_sweeps_hint_ = (int)h[3];
    jmc_force_v_ = h[4] != 0.0;
    return (int)h[0];
  }
  void read_flags(const int* dev_p, int n, int* host) {
    CORRLA_HIP(hipMemcpyAsync(host, dev_p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, stream));
    sync();
  }
  template <class T>
  bool device_qr_robust_fits(int64_t l) const {
    return robust_qr_ && (device_chol_fits<T>(l) || device_chol_blocked_fits<T>(l));
  }
  // one column block: the products of a pass may run in place (see apply_inplace)
  bool qr_inplace_fits(int64_t l) const { return col_blocking(l).nblk == 1; }
  // 2 x 2 blocked robust factorisation (l > 176 / 152): the whole Gram is inspected (shift decision, ||G - I||) and shifted
  // once, the two diagonal-block factorisations take the shift from the record, combine_need forms the pass verdict
  template <class T>
  void* alloc_inspect() { return alloc_zeroed(sizeof(k::GramInspect<T>)); }
  template <class T>
  const void* inspect_shift_ptr(const void* insp) const { return &((const k::GramInspect<T>*)insp)->shift; }
  template <class T>
  void gram_inspect(Skinny<T>& g, int64_t l, float shift_rel, int shift_mode, void* insp) {
    hipLaunchKernelGGL((k::gram_inspect_kernel<T>), dim3(1), dim3(1024), 0, stream, g.p, g.ld, (int)l, shift_rel, shift_mode,
                       (k::GramInspect<T>*)insp, run_if_);
    CORRLA_HIP(hipGetLastError());
  }
  template <class T>
  void combine_need(int* need, const void* insp, const int* na, const int* nb) {
    hipLaunchKernelGGL((k::combine_need_kernel<T>), dim3(1), dim3(1), 0, stream, need, (const k::GramInspect<T>*)insp, na, nb,
                       run_if_);
    CORRLA_HIP(hipGetLastError());
  }
  template <class T>
  void chol_inv_robust(const Skinny<T>& g, int64_t r, T piv_rel, float shift_rel, int shift_mode, float null_excess,
                       Skinny<T>& m_out, void* st_dev, int slot, int* need_next, int* null_mask,
                       const void* abs_shift = nullptr, float need_ratio = 0.f) {
    k::CholRobust rb{shift_rel, shift_mode, null_excess, need_next, null_mask, run_if_, need_ratio, abs_shift};
    hipLaunchKernelGGL((k::chol_inv_kernel<T>), dim3(1), dim3(k::chol_inv_threads((int)r)),
                       k::chol_inv_lds_bytes((int)r, sizeof(T)), stream, (const T*)g.p, g.ld, (int)r, piv_rel, m_out.p,
                       m_out.ld, (k::CholStatus*)st_dev + slot, rb);
    CORRLA_HIP(hipGetLastError());
  }
  // y <- y * m (m: l x l): in place -- a workgroup / wave of the product kernels reads exactly the rows it writes,
  // all of them before its first store (one column block only: device_qr_robust_fits)
  template <class T>
  void apply_inplace(Skinny<T>& y, int64_t l, const Skinny<T>& m) {
    Skinny<T> out = y.view_cols(l);
    Skinny<T> mv = m.view_cols(l);
    mv.rows = l;
    launch_gemm<T>(true, as_rowmajor_transposed(y, l), mv, out, (const T*)nullptr, y.rows, l);
  }
  template <class T>
  void refill_null(Skinny<T>& y, int64_t l, const int* null_mask, uint64_t seed) {
    const int bx = (int)std::min<int64_t>(64, (y.rows + 255) / 256);
    hipLaunchKernelGGL((k::refill_null_kernel<T>), dim3((unsigned)std::max(bx, 1), (unsigned)l), dim3(256), 0, stream, y.p, y.ld,
                       y.rows, (int)l, null_mask, seed, run_if_);
    CORRLA_HIP(hipGetLastError());
  }

  void read_chol_status(const void* st_dev, int n, int* fail, float* min_ratio, float* dev_i) {
    static_assert(sizeof(k::CholStatus) == 32, "driver.hpp assumes 32-byte status records");
    static_assert(k::kNeedNonFinite == kFlagNonFinite && k::kNeedNullCols == kFlagNullCols, "need_next bits: kernels vs driver");
    std::vector<k::CholStatus> h((size_t)std::max(n, 1));
    CORRLA_HIP(hipMemcpyAsync(h.data(), st_dev, sizeof(k::CholStatus) * n, hipMemcpyDeviceToHost, stream));
    sync();
    for (int i = 0; i < n; ++i) {
      fail[i] = h[i].fail;
      min_ratio[i] = h[i].min_ratio;
      dev_i[i] = h[i].dev_i;
      if (env_int("CORRLA_DEBUG", 0) >= 2)
        std::fprintf(stderr, "[corrla] chol_inv[%d]: fail %d min_ratio %.3g dev_i %.3g clk %lld wall %lld (%.0f MHz)\n", i,
                     h[i].fail, h[i].min_ratio, h[i].dev_i, h[i].clk, h[i].wall,
                     h[i].wall > 0 ? 100.0 * (double)h[i].clk / (double)h[i].wall : 0.0);
    }
  }

  // m_out (r x r) = (I + E)^(-1/2) with G = I + E given in g (overwritten by E); everything on the device
  template <class T>
  void inv_sqrt_series(Skinny<T>& g, int64_t r, Skinny<T>& m_out) {
    Skinny<T> e2 = alloc_skinny<T>(r, r), e3 = alloc_skinny<T>(r, r);
    if (e2.ld != g.ld || m_out.ld != g.ld) throw Error(ST_EINVAL, "internal: series operands must share a leading dimension");
    dim3 grid((unsigned)((r + 63) / 64), (unsigned)r);
    hipLaunchKernelGGL((k::series_prep_kernel<T>), grid, dim3(64), 0, stream, g.p, g.ld, (int)r);
    Big<T> eb;  // E is symmetric: its column-major image is also its row-major image
    eb.p = g.p;
    eb.rows = r;
    eb.cols = r;
    eb.ld = g.ld;
    eb.cols_readable = g.ld;
    Skinny<T> gv = g.view_cols(r);
    gv.rows = r;
    gemm_nn(eb, gv, e2, (const T*)nullptr);
    gemm_nn(eb, e2, e3, (const T*)nullptr);
    memset_zero(m_out.p, (size_t)m_out.ld * m_out.cols_alloc * sizeof(T));
    hipLaunchKernelGGL((k::series_combine_kernel<T>), grid, dim3(64), 0, stream, (const T*)g.p, (const T*)e2.p,
                       (const T*)e3.p, g.ld, (int)r, m_out.p, m_out.ld);
    CORRLA_HIP(hipGetLastError());
  }

  template <class T>
  void set_tsqr_attrs() {
    const hipFuncAttribute attr = hipFuncAttributeMaxDynamicSharedMemorySize;
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::hh_leaf_factor_kernel<T, false>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::hh_tree_factor_kernel<T, false>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::hh_tree_apply_kernel<T, false>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::hh_leaf_apply_kernel<T, false>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::hh_leaf_factor_kernel<T, true>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::hh_tree_factor_kernel<T, true>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::hh_tree_apply_kernel<T, true>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::hh_leaf_apply_kernel<T, true>, attr, 160 * 1024));
  }

  // ---- Householder TSQR with explicit thin Q (tsqr_kernels.hpp) -------------------------------------------
  // one 2 l x l panel (plus the 16 x 16 T and Gram blocks of the blocked form) must fit in LDS: l <= 142 (f32) / 99 (f64)
  bool hh_wy_ = env_int("CORRLA_HH_WY", 1) != 0;  // blocked compact-WY panels on the MFMA units (0: the unblocked panels)
  template <class T>
  size_t hh_panel_lds(int rows, int l) const {
    return hh_wy_ ? k::hh_wy_lds_bytes(rows, l, sizeof(T)) : k::hh_lds_bytes(rows, l, sizeof(T));
  }
  template <class T>
  bool householder_fits(int64_t l) const {
    return l >= 1 && l <= 4096 && hh_panel_lds<T>((int)(2 * l), (int)l) <= (size_t)160 * 1024 && 2 * l <= 64 * k::kHhMaxRowsPerLane;
  }
  // Up sweep of the TSQR of y (m x l, m >= l): leaf reflectors -> tmp (same shape as y), tree reflectors and the root's
  // R factor (l x l, column-major, ld = l) stay in call-lifetime device buffers named by the returned state.
  template <class T>
  struct HhState {
    int64_t m = 0;
    int l = 0, nleaf = 0, levels = 0, max_rows = 0;
    std::vector<int> n_at;
    std::vector<T*> rbuf, cbuf, taub, vbuf, tbuf;  // tbuf: the T factors of the 16-column blocks (blocked form)
    T* r_root() const { return rbuf[levels]; }
  };
  template <class T>
  HhState<T> householder_up(Skinny<T>& y, Skinny<T>& tmp) {
    HhState<T> h;
    const int64_t m = y.rows;
    const int l = (int)y.cols;
    if (m < l) throw Error(ST_EINVAL, "householder_thin_q: fewer rows than columns");
    if (!householder_fits<T>(l)) throw Error(ST_EINVAL, "householder_thin_q: panel does not fit in LDS");
    const int64_t br = 2 * (int64_t)l;
    const int64_t nleaf64 = m <= br ? 1 : (m + br - 1) / br;
    if (nleaf64 > 0x3fffffff) throw Error(ST_EINVAL, "householder_thin_q: too many panels");
    h.m = m;
    h.l = l;
    h.nleaf = (int)nleaf64;
    h.max_rows = (int)std::min<int64_t>(m, br);
    const size_t lds = hh_panel_lds<T>(h.max_rows, l);
    h.n_at = {h.nleaf};
    while (h.n_at.back() > 1) h.n_at.push_back((h.n_at.back() + 1) / 2);
    h.levels = (int)h.n_at.size() - 1;
    const size_t ll = (size_t)l * l;
    h.rbuf.resize(h.levels + 1);
    h.cbuf.resize(h.levels + 1);
    h.taub.resize(h.levels + 1);
    h.vbuf.assign(h.levels + 1, nullptr);
    h.tbuf.assign(h.levels + 1, nullptr);
    const size_t tsz = (size_t)k::hh_wy_panels(l) * k::kWyNb * k::kWyNb;
    for (int k_ = 0; k_ <= h.levels; ++k_) {
      if (hh_wy_) h.tbuf[k_] = (T*)alloc_bytes(tsz * h.n_at[k_] * sizeof(T));
      h.rbuf[k_] = (T*)alloc_bytes(ll * h.n_at[k_] * sizeof(T));
      h.cbuf[k_] = (T*)alloc_bytes(ll * h.n_at[k_] * sizeof(T));
      h.taub[k_] = (T*)alloc_bytes((size_t)l * h.n_at[k_] * sizeof(T));
      if (k_ >= 1) h.vbuf[k_] = (T*)alloc_bytes(2 * ll * h.n_at[k_] * sizeof(T));
    }
    auto leaf = hh_wy_ ? k::hh_leaf_factor_kernel<T, true> : k::hh_leaf_factor_kernel<T, false>;
    auto tree = hh_wy_ ? k::hh_tree_factor_kernel<T, true> : k::hh_tree_factor_kernel<T, false>;
    hipLaunchKernelGGL(leaf, dim3((unsigned)h.nleaf), dim3(k::kHhThreads), lds, stream, (const T*)y.p, y.ld, m, l, h.nleaf, tmp.p,
                       tmp.ld, h.taub[0], h.rbuf[0], h.tbuf[0]);
    const size_t lds_tree = hh_panel_lds<T>(2 * l, l);
    for (int k_ = 1; k_ <= h.levels; ++k_)
      hipLaunchKernelGGL(tree, dim3((unsigned)h.n_at[k_]), dim3(k::kHhThreads), lds_tree, stream, (const T*)h.rbuf[k_ - 1],
                         h.n_at[k_ - 1], l, h.vbuf[k_], h.taub[k_], h.rbuf[k_], h.tbuf[k_]);
    CORRLA_HIP(hipGetLastError());
    return h;
  }
  // Down sweep: y <- Q [C; 0] with Q the orthogonal factor of the up sweep and C = root_coef (l x l, column-major,
  // ld = l; nullptr = the identity, i.e. y <- the explicit thin Q).  A cross-rank TSQR passes its block of the top-level
  // Q here (driver.hpp householder_panel).
  template <class T>
  void householder_down(HhState<T>& h, Skinny<T>& y, Skinny<T>& tmp, const T* root_coef = nullptr) {
    const int l = h.l;
    const size_t lds = k::hh_lds_bytes(h.max_rows, l, sizeof(T));
    const size_t lds_tree = k::hh_lds_bytes(2 * l, l, sizeof(T));
    auto tree = hh_wy_ ? k::hh_tree_apply_kernel<T, true> : k::hh_tree_apply_kernel<T, false>;
    auto leaf = hh_wy_ ? k::hh_leaf_apply_kernel<T, true> : k::hh_leaf_apply_kernel<T, false>;
    for (int k_ = h.levels; k_ >= 1; --k_)
      hipLaunchKernelGGL(tree, dim3((unsigned)h.n_at[k_]), dim3(k::kHhThreads), lds_tree, stream,
                         (const T*)(k_ == h.levels ? root_coef : h.cbuf[k_]), (const T*)h.vbuf[k_], (const T*)h.taub[k_],
                         h.n_at[k_ - 1], l, h.cbuf[k_ - 1], (const T*)h.tbuf[k_]);
    hipLaunchKernelGGL(leaf, dim3((unsigned)h.nleaf), dim3(k::kHhThreads), lds, stream,
                       (const T*)(h.levels == 0 ? root_coef : h.cbuf[0]), (const T*)tmp.p, tmp.ld, (const T*)h.taub[0], h.m, l,
                       h.nleaf, y.p, y.ld, (const T*)h.tbuf[0]);
    CORRLA_HIP(hipGetLastError());
  }
  // y (m x l, m >= l) <- thin Q of its Householder QR; tmp (same shape) receives the leaf reflectors
  template <class T>
  void householder_thin_q(Skinny<T>& y, Skinny<T>& tmp) {
    HhState<T> h = householder_up(y, tmp);
    householder_down(h, y, tmp);
  }
  // widest panel one workgroup can hold: 142 (f32) / 99 (f64)
  template <class T>
  int householder_max_width() const {
    int w = 1;
    while (householder_fits<T>(w + 1)) ++w;
    return w;
  }
  int rank() const { return comm_rank; }

  template <class T>
  void set_jmc_attrs() {
    const hipFuncAttribute attr = hipFuncAttributeMaxDynamicSharedMemorySize;
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jmc_step_kernel<T, 1, 16>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jmc_step_kernel<T, 2, 16>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jmc_step_kernel<T, 3, 16>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jmc_step_kernel<T, 4, 16>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jmc_step_kernel<T, 5, 16>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jmc_step_kernel<T, 6, 16>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jmc_step_kernel<T, 7, 16>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jmc_step_kernel<T, 8, 16>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jmc_step_kernel<T, 9, 16>, attr, 160 * 1024));
  }

  // ---- multi-workgroup block Jacobi (jacobi_mc_kernels.hpp) ------------------------------------------------
  // lanes per Jacobi processor.  (8-lane processors -- half the waves for the same block pair, twice the column per
  // lane -- measured 6 % slower at l = 138 f32: the rounds are bound by the per-lane column traffic, not by the
  // rotation arithmetic they would amortise; the kernel keeps the template parameter, nothing instantiates 8.)
  template <class T>
  int jmc_lanes(int64_t) const {
    return 16;
  }
  // geometry for an l x l core: chunk rows NC, workgroups NP, block width b (even, <= 32); false when it does not fit
  template <class T>
  bool jmc_geometry(int64_t l, int* nc_out, int* np_out, int* b_out) const {
    if (l < 2 || l > 288) return false;
    const int lanes = jmc_lanes<T>(l);
    const int nc = (int)((l + 2 * lanes - 1) / (2 * lanes));
    // block width: a multiple of four columns (= whole waves of four 16-lane processors, whole sub-blocks of the
    // wave-local schedule); CORRLA_JMC_LOCAL=0 keeps the round-2 rule (even)
    const int local = env_int("CORRLA_JMC_LOCAL", 1);
    auto width = [&](int np_) {
      int bb = (int)((l + 2 * np_ - 1) / (2 * np_));
      return local ? (bb + 3) / 4 * 4 : bb + (bb & 1);
    };
    int np = env_int("CORRLA_JMC_NP", 0);
    if (np <= 0) {
      // fewest workgroups whose block pair fits one CU (<= 32 processors, LDS): fewer, larger steps per sweep
      np = 2;
      while (np < 128 && (width(np) > jmc_max_b_ || k::jmc_lds_bytes(nc, width(np), sizeof(T), lanes) > (size_t)160 * 1024)) ++np;
    }
    const int b = width(np);
    if (np < 1 || b < 2 || b > 32 || k::jmc_lds_bytes(nc, b, sizeof(T), lanes) > (size_t)160 * 1024) return false;
    *nc_out = nc;
    *np_out = np;
    *b_out = b;
    return true;
  }
  template <class T, int NC>
  void jmc_launch_step(int lanes, int np, int b, T* w, T* v, int nblocks, int step, int sweep, T tol, T tol_early, T floor2, k::JmcCtl* ctl) {
    const size_t lds = k::jmc_lds_bytes(NC, b, sizeof(T), lanes);
    const unsigned threads = (unsigned)((b * lanes + 63) / 64 * 64);
    hipLaunchKernelGGL((k::jmc_step_kernel<T, NC, 16>), dim3((unsigned)np), dim3(threads), lds, stream, w, v, b, nblocks, step,
                       sweep, step == 0 ? 1 : 0, tol, tol_early, floor2, ctl, jmc_local_);
  }
  // conv_status: device CholStatus record that receives the convergence verdict of the fixed number of sweeps
  // enqueued without any synchronisation (the caller checks it later); nullptr: sweeps are enqueued in groups and the
  // host waits for each group until the iteration has converged
  template <class T>
  void small_svd_mc(const Skinny<T>& c, int64_t l, int64_t k, Skinny<T>& m1, Skinny<T>& m2, T* s_dev, void* conv_status) {
    int nc = 0, np = 0, b = 0;
    if (!jmc_geometry<T>(l, &nc, &np, &b)) throw Error(ST_EINVAL, "internal: core too large for the multi-workgroup Jacobi");
    jmc_local_ = env_int("CORRLA_JMC_LOCAL", 1);  // 1: wave-local sub-block schedule (jacobi_mc_kernels.hpp), 0: ring schedule
    // global column pitch = LDS column pitch: a block of b columns is one contiguous byte range in both
    const int lanes = jmc_lanes<T>(l);
    const int rp = k::jmc_pitch(nc, (int)sizeof(T), lanes), nblocks = 2 * np, ncols_pad = nblocks * b;
    T* wj = (T*)alloc_bytes((size_t)rp * ncols_pad * sizeof(T));
    T* vj = (T*)alloc_bytes((size_t)rp * ncols_pad * sizeof(T));
    k::JmcCtl* ctl = (k::JmcCtl*)alloc_bytes(sizeof(k::JmcCtl));
    k::CholStatus* st = conv_status ? (k::CholStatus*)conv_status : (k::CholStatus*)alloc_bytes(sizeof(k::CholStatus));
    // optimistic calls (conv_status given) may use the W-only mode when the core is well conditioned; the
    // host-controlled repeat always accumulates V
    const int force_v = (conv_status == nullptr || jmc_force_v_ || env_int("CORRLA_JMC_FORCE_V", 0)) ? 1 : 0;
    hipLaunchKernelGGL((k::jmc_init_kernel<T>), dim3(1), dim3(1024), 0, stream, (const T*)c.p, c.ld, (int)l, wj, vj, rp,
                       ncols_pad, force_v, ctl);
    const double eps = (double)std::numeric_limits<T>::epsilon();
    const T tol = (T)(std::sqrt((double)l) * eps);
    // The iteration ends with the sweep in which no pair exceeded sqrt(eps) (quadratic convergence).  Clustered
    // singular values do not converge quadratically: the W / sigma factor of a 1.25e6 x 512 Gaussian sketch came out
    // orthonormal to 6e-5 only.  Running to a sweep without any rotation costs two more sweeps; the driver instead
    // re-orthonormalises that factor with one Cholesky-QR pass (a first-order (I + E)^-1/2 here), which is cheaper.
    const T tol_early = env_int("CORRLA_JACOBI_STRICT", 0) ? tol : (T)std::sqrt(eps);
    const T floor2 = (T)((double)l * eps * eps);  // squared norm of a numerically zero column (see the kernel)
    auto enqueue_sweeps = [&](int s0, int s1) {
      for (int sw = s0; sw < s1; ++sw)
        for (int step = 0; step < nblocks - 1; ++step) switch (nc) {
            case 1: jmc_launch_step<T, 1>(lanes, np, b, wj, vj, nblocks, step, sw, tol, tol_early, floor2, ctl); break;
            case 2: jmc_launch_step<T, 2>(lanes, np, b, wj, vj, nblocks, step, sw, tol, tol_early, floor2, ctl); break;
            case 3: jmc_launch_step<T, 3>(lanes, np, b, wj, vj, nblocks, step, sw, tol, tol_early, floor2, ctl); break;
            case 4: jmc_launch_step<T, 4>(lanes, np, b, wj, vj, nblocks, step, sw, tol, tol_early, floor2, ctl); break;
            case 5: jmc_launch_step<T, 5>(lanes, np, b, wj, vj, nblocks, step, sw, tol, tol_early, floor2, ctl); break;
            case 6: jmc_launch_step<T, 6>(lanes, np, b, wj, vj, nblocks, step, sw, tol, tol_early, floor2, ctl); break;
            case 7: jmc_launch_step<T, 7>(lanes, np, b, wj, vj, nblocks, step, sw, tol, tol_early, floor2, ctl); break;
            case 8: jmc_launch_step<T, 8>(lanes, np, b, wj, vj, nblocks, step, sw, tol, tol_early, floor2, ctl); break;
            default: jmc_launch_step<T, 9>(lanes, np, b, wj, vj, nblocks, step, sw, tol, tol_early, floor2, ctl); break;
          }
      CORRLA_HIP(hipGetLastError());
    };
    const size_t lds_fin = (size_t)(l + 2) * sizeof(T) + (size_t)(l + 2) * sizeof(int) + 64;
    auto finish = [&](int nsw) {
      hipLaunchKernelGGL((k::jmc_finish_kernel<T>), dim3(1), dim3(1024), lds_fin, stream, (const T*)wj, (const T*)vj, rp, (int)l,
                         nsw, (const k::JmcCtl*)ctl, m1.p, m1.ld, m2.p, m2.ld, s_dev, (int)k, st);
      // W-only mode: the accumulated-rotation factor is recovered from the core itself (no-op otherwise)
      const unsigned groups = (unsigned)(l * k);
      hipLaunchKernelGGL((k::jmc_other_factor_kernel<T>), dim3((groups * 16 + 255) / 256), dim3(256), 0, stream, (const T*)c.p,
                         c.ld, (int)l, (int)k, (const T*)m2.p, m2.ld, (const T*)s_dev, (const k::JmcCtl*)ctl, m1.p, m1.ld);
      CORRLA_HIP(hipGetLastError());
    };
    if (conv_status) {
      const int nsw_default = std::max(1, env_int("CORRLA_JMC_SWEEPS", sizeof(T) == 4 ? 10 : 13));
      const int nsw = std::min(k::kJmcMaxSweeps, (jmc_sweeps_hint_ > 0 ? std::min(nsw_default, jmc_sweeps_hint_ + 2) : nsw_default) + jmc_extra_sweeps_);
      enqueue_sweeps(0, nsw);
      finish(nsw);
      if (env_int("CORRLA_DEBUG", 0)) {
        k::JmcCtl hd;
        CORRLA_HIP(hipMemcpyAsync(&hd, ctl, sizeof(hd), hipMemcpyDeviceToHost, stream));
        sync();
        int used = 0;
        while (used < nsw && hd.rot[used] && hd.big[used]) ++used;
        std::fprintf(stderr, "[corrla] jacobi_svd (multi-workgroup, %d sweeps enqueued, %s) l=%d np=%d b=%d nc=%d sweeps run=%d rounds(wg0)=%llu "
                             "cycles/round=%.0f ns/round=%.0f (%.0f MHz)\n", nsw, hd.with_v ? "V accumulated" : "W only", (int)l, np, b, nc, std::min(used + 1, nsw),
                     hd.rounds, hd.rounds ? (double)hd.clk / hd.rounds : 0.0, hd.rounds ? 10.0 * hd.wall / hd.rounds : 0.0,
                     hd.wall ? 100.0 * hd.clk / hd.wall : 0.0);
        if (hd.steps)
          std::fprintf(stderr, "[corrla]   per step (wg0, us): load %.2f norms %.2f rounds %.2f store %.2f total %.2f over %llu steps\n",
                       0.01 * hd.t_load / hd.steps, 0.01 * hd.t_norm / hd.steps, 0.01 * hd.wall / hd.steps,
                       0.01 * hd.t_store / hd.steps, 0.01 * hd.t_total / hd.steps, hd.steps);
        if (hd.steps > 1)
          std::fprintf(stderr, "[corrla]   span first-start..last-end over all workgroups: %.2f us per step\n",
                       0.01 * hd.t_span / (hd.steps - 1));
      }
      return;
    }
    int done = 0;
    k::JmcCtl h;
    while (done < k::kJmcMaxSweeps) {
      const int s1 = std::min(k::kJmcMaxSweeps, done + 8);
      enqueue_sweeps(done, s1);
      CORRLA_HIP(hipMemcpyAsync(&h, ctl, sizeof(h), hipMemcpyDeviceToHost, stream));
      sync();
      bool conv = false;
      for (int s_ = 0; s_ < s1; ++s_) conv = conv || !(h.rot[s_] && h.big[s_]);
      done = s1;
      if (conv || h.bad) break;
    }
    if (h.bad) throw Error(ST_ENUMERIC, "non-finite core matrix in small SVD");
    finish(done);
    if (env_int("CORRLA_DEBUG", 0)) {
      int used = 0;
      while (used < done && h.rot[used] && h.big[used]) ++used;
      std::fprintf(stderr, "[corrla] jacobi_svd (multi-workgroup) l=%d np=%d b=%d sweeps=%d\n", (int)l, np, b, used + 1);
    }
  }

  template <class T, int BIG_E>
  void set_ring_attrs() {
    const hipFuncAttribute attr = hipFuncAttributeMaxDynamicSharedMemorySize;
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jacobi_ring_kernel<T, 8>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jacobi_ring_kernel<T, 12>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jacobi_ring_kernel<T, 16>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jacobi_ring_kernel<T, BIG_E>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jacobi_ring_w_kernel<T, 8, 8>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jacobi_ring_w_kernel<T, 12, 8>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jacobi_ring_w_kernel<T, 16, 8>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jacobi_ring_w_kernel<T, BIG_E, 8>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jacobi_ring_w_kernel<T, 24, 4>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jacobi_ring_w_kernel<T, 32, 4>, attr, 160 * 1024));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jacobi_ring_w_kernel<T, 36, 4>, attr, 160 * 1024));
  }

  // SVD of the l x l core (random_svd.rs:89).  Default: single-workgroup LDS-resident Jacobi when W fits
  // in LDS, block Jacobi over many waves otherwise (any l up to 1024).  CORRLA_SVD=block / host force the
  // block kernel / the f64 host Jacobi.
  template <class T>
  void small_svd(const Skinny<T>& c, int64_t l, int64_t k, Skinny<T>& m1, Skinny<T>& m2, T* s_dev, void* conv_status) {
    // conv_status (optional): a device status record; kernels that run a FIXED number of sweeps report there
    // whether they converged, the others (loop to convergence inside one launch) leave it cleared = converged
    if (conv_status) memset_zero(conv_status, sizeof(k::CholStatus));
    const char* mode = std::getenv("CORRLA_SVD");
    {
      int nc_, np_, b_;
      const bool want_mc = mode && std::strcmp(mode, "mc") == 0;
      const bool want_other = mode && !want_mc;
      if (!want_other && !env_int("CORRLA_HOST_SVD", 0) && (want_mc || l >= jmc_min_l_) && jmc_geometry<T>(l, &nc_, &np_, &b_)) {
        small_svd_mc(c, l, k, m1, m2, s_dev, conv_status);
        return;
      }
    }
    const bool want_host = (mode && std::strcmp(mode, "host") == 0) || env_int("CORRLA_HOST_SVD", 0);
    const bool want_lds = mode && std::strcmp(mode, "lds") == 0;
    if (want_host || l > 1024) {
      small_svd_host(*this, c, l, k, m1, m2, s_dev);
      return;
    }
    // The single-workgroup and block Jacobi kernels carry no status word: a non-finite core would come back as a
    // triplet of zeros.  One small launch scans the core first: optimistic runs find fail = 3 in the status record at
    // the end of the call, host-controlled ones read the word now.
    {
      int* bad = conv_status ? nullptr : alloc_flags(1);
      hipLaunchKernelGGL((k::core_finite_check_kernel<T>), dim3(1), dim3(1024), 0, stream, (const T*)c.p, c.ld, (int)l,
                         (k::CholStatus*)conv_status, bad);
      CORRLA_HIP(hipGetLastError());
      if (bad) {
        int h = 0;
        read_flags(bad, 1, &h);
        if (h) throw Error(ST_ENUMERIC, "non-finite core matrix in small SVD");
      }
    }
    const bool want_block = mode && std::strcmp(mode, "block") == 0;
    (void)want_lds;
    if (want_block) {
      small_svd_block(c, l, k, m1, m2, s_dev);
      return;
    }
    small_svd_lds(c, l, k, m1, m2, s_dev);  // falls through to the block kernel when W does not fit in LDS
  }

  template <class T>
  void small_svd_block(const Skinny<T>& c, int64_t l, int64_t k, Skinny<T>& m1, Skinny<T>& m2, T* s_dev) {
    const int nb = (int)(2 * ((l + 15) / 16));          // even number of 8-column blocks
    const int cols_pad = nb * 8;
    const int rows_pad = (int)round_up(l, 16);
    const int64_t ld = rows_pad;
    T* wj = (T*)alloc_bytes((size_t)ld * cols_pad * sizeof(T));
    T* vj = (T*)alloc_bytes((size_t)ld * cols_pad * sizeof(T));
    k::JacobiCtl* ctl = (k::JacobiCtl*)alloc_bytes(sizeof(k::JacobiCtl));
    hipLaunchKernelGGL((k::jacobi_init_kernel<T>), dim3(64), dim3(256), 0, stream, (const T*)c.p, c.ld, (int)l, wj, ld, vj,
                       ld, cols_pad, rows_pad, ctl);
    const double eps = (double)std::numeric_limits<T>::epsilon();
    const float tol_early = (float)std::sqrt(eps);
    const size_t lds = (size_t)(2 * 16 * (rows_pad + 1) + 7 * 16 * 17 + 32) * sizeof(T) + 16 * sizeof(int) + 64;
    const int max_sweeps = env_int("CORRLA_JACOBI_SWEEPS", 12), inner = env_int("CORRLA_JACOBI_INNER", 1);
    for (int sw = 0; sw < max_sweeps; ++sw) {
      for (int round = 0; round < nb - 1; ++round)
        hipLaunchKernelGGL((k::jacobi_block_round_kernel<T>), dim3(nb / 2), dim3(256), lds, stream, wj, ld, vj, ld, rows_pad,
                           nb, round, inner, ctl);
      hipLaunchKernelGGL(k::jacobi_sweep_end_kernel, dim3(1), dim3(1), 0, stream, ctl, tol_early);
    }
    const size_t lds_fin = (size_t)(l + 2) * sizeof(T) + (size_t)(l + 2) * sizeof(int) + 64;
    hipLaunchKernelGGL((k::jacobi_finish_kernel<T>), dim3(1), dim3(1024), lds_fin, stream, (const T*)wj, ld, (const T*)vj,
                       ld, (int)l, m1.p, m1.ld, m2.p, m2.ld, s_dev, (int)k);
    CORRLA_HIP(hipGetLastError());
    if (env_int("CORRLA_DEBUG", 0)) {
      k::JacobiCtl h;
      CORRLA_HIP(hipMemcpyAsync(&h, ctl, sizeof(h), hipMemcpyDeviceToHost, stream));
      sync();
      std::fprintf(stderr, "[corrla] block jacobi l=%d nb=%d sweeps=%u done=%u\n", (int)l, nb, h.sweeps, h.done);
    }
  }

  template <class T>
  void small_svd_lds(const Skinny<T>& c, int64_t l, int64_t k, Skinny<T>& m1, Skinny<T>& m2, T* s_dev) {
    constexpr size_t kLdsMax = (size_t)160 * 1024;
    const size_t lds2 = k::jacobi_lds_bytes((int)l, sizeof(T), true);
    const size_t lds1 = k::jacobi_lds_bytes((int)l, sizeof(T), false);
    const bool ring_ok = l >= 2 && l <= 144 && !env_int("CORRLA_JACOBI_NORING", 0) &&
                         k::jacobi_ring_w_lds_bytes((int)l, l <= 96 ? 96 : (l <= 128 ? 128 : 144), sizeof(T)) <= kLdsMax;
    // the LDS-resident kernels with V in global memory are far slower than the block kernel
    if (!ring_ok && (lds2 > kLdsMax || l > k::kJacobiMaxL)) {
      small_svd_block(c, l, k, m1, m2, s_dev);
      return;
    }
    const int64_t ldv = round_up(k::jacobi_pitch((int)l, (int)(16 / sizeof(T))), 16);
    T* vg = (T*)alloc_bytes((size_t)ldv * l * sizeof(T));
    int* info = (int*)alloc_bytes(sizeof(int) * 4);
    const double eps = (double)std::numeric_limits<T>::epsilon();
    const T tol = (T)(std::sqrt((double)l) * eps);
    // quadratic convergence: a sweep that starts below sqrt(eps) ends below tol -- except for clustered singular
    // values, whose W / sigma factor the driver re-orthonormalises afterwards (see small_svd_mc)
    const T tol_early = env_int("CORRLA_JACOBI_STRICT", 0) ? tol : (T)std::sqrt(eps);
    const bool v_lds = lds2 <= kLdsMax;
    const size_t lds = v_lds ? lds2 : lds1;
    // ring kernel: columns resident in registers (l <= 144)
    // ring kernels: rows per column slot = G lanes x E rows.  G = 4 (fewer, longer waves) measured ~12 % slower
    // than G = 8 at l = 96..144 (latency is hidden by the extra waves); kept behind CORRLA_RING_G4 for experiments
    const bool ring_replay = !env_int("CORRLA_JACOBI_NOREPLAY", 0);
    const int ring_g = (ring_replay && l > 64 && env_int("CORRLA_RING_G4", 0)) ? 4 : 8;
    const int ring_e8 = l <= 64 ? 8 : (l <= 96 ? 12 : (l <= 128 ? 16 : (sizeof(T) == 4 ? 20 : 18)));
    const int ring_e4 = l <= 96 ? 24 : (l <= 128 ? 32 : 36);
    const int ring_rs = ring_g == 4 ? 4 * ring_e4 : 8 * ring_e8;
    const size_t ring_lds = ring_replay ? k::jacobi_ring_w_lds_bytes((int)l, ring_rs, sizeof(T))
                                        : k::jacobi_ring_lds_bytes((int)l, ring_e8, sizeof(T));
    if (l >= 2 && l <= 144 && ring_lds <= kLdsMax && !env_int("CORRLA_JACOBI_NORING", 0)) {
      const int np = (int)((l + 1) / 2);
      const dim3 block((unsigned)(np * ring_g));  // a partial last wave: no idle processors, no LDS slots for them
      const int max_sw = env_int("CORRLA_JACOBI_SWEEPS", 40);
      const bool replay = ring_replay;
      const int n2 = 2 * np;
      k::RotEntry<T>* rot = nullptr;
      int* rank_g = nullptr;
      if (replay) {
        rot = (k::RotEntry<T>*)alloc_bytes((size_t)max_sw * n2 * k::kRingProcPad * sizeof(k::RotEntry<T>));
        rank_g = (int*)alloc_bytes(sizeof(int) * (size_t)n2);
      }
#define CORRLA_RING_W(EE, GG)                                                                                       \
  hipLaunchKernelGGL((k::jacobi_ring_w_kernel<T, EE, GG>), dim3(1), block, ring_lds, stream, (const T*)c.p, c.ld, (int)l, \
                     m2.p, m2.ld, s_dev, (int)k, tol, tol_early, max_sw, rot, rank_g, info)
#define CORRLA_RING(EE)                                                                                                  \
  hipLaunchKernelGGL((k::jacobi_ring_kernel<T, EE>), dim3(1), block, ring_lds, stream, (const T*)c.p, c.ld, (int)l, m1.p, \
                     m1.ld, m2.p, m2.ld, s_dev, (int)k, tol, tol_early, max_sw, info)
      constexpr int kBigE = sizeof(T) == 4 ? 20 : 18;
      if (replay && ring_g == 4) {
        if (l <= 96) CORRLA_RING_W(24, 4);
        else if (l <= 128) CORRLA_RING_W(32, 4);
        else CORRLA_RING_W(36, 4);
      } else if (replay) {
        if (l <= 64) CORRLA_RING_W(8, 8);
        else if (l <= 96) CORRLA_RING_W(12, 8);
        else if (l <= 128) CORRLA_RING_W(16, 8);
        else CORRLA_RING_W(kBigE, 8);
      } else {
        if (l <= 64) CORRLA_RING(8);
        else if (l <= 96) CORRLA_RING(12);
        else if (l <= 128) CORRLA_RING(16);
        else CORRLA_RING(kBigE);
      }
#undef CORRLA_RING_W
#undef CORRLA_RING
      CORRLA_HIP(hipGetLastError());
      if (replay) {
        hipLaunchKernelGGL((k::jacobi_replay_v_kernel<T>), dim3((unsigned)((l + 256 / k::kReplayLanes - 1) / (256 / k::kReplayLanes))), dim3(256), 0,
                           stream,
                           (const k::RotEntry<T>*)rot, (const int*)info, (const int*)rank_g, (int)l, (int)k, m1.p, m1.ld);
        CORRLA_HIP(hipGetLastError());
      }
      if (env_int("CORRLA_DEBUG", 0)) {
        int h[4] = {0, 0, 0, 0};
        CORRLA_HIP(hipMemcpyAsync(h, info, sizeof(int), hipMemcpyDeviceToHost, stream));
        sync();
        std::fprintf(stderr, "[corrla] jacobi_svd (ring) l=%d sweeps=%d\n", (int)l, h[0]);
      }
      return;
    }
    // role-split kernel: W updates and V updates on different waves (needs both images in LDS, <= 72 pairs,
    // <= 36 sixteen-byte chunks per column)
    const size_t lds_split = k::jacobi_split_lds_bytes((int)l, sizeof(T));
    const int nchunk_s = k::jacobi_pitch((int)l, (int)(16 / sizeof(T))) / (int)(16 / sizeof(T));
    if (lds_split <= kLdsMax && (l + 1) / 2 <= 72 && nchunk_s <= 36 && !env_int("CORRLA_JACOBI_NOSPLIT", 0)) {
      hipLaunchKernelGGL((k::jacobi_svd_split_kernel<T>), dim3(1), dim3(1024), lds_split, stream, (const T*)c.p, c.ld,
                         (int)l, m1.p, m1.ld, m2.p, m2.ld, s_dev, (int)k, tol, tol_early, 40, info);
      CORRLA_HIP(hipGetLastError());
      if (env_int("CORRLA_DEBUG", 0)) {
        int h[4] = {0, 0, 0, 0};
        CORRLA_HIP(hipMemcpyAsync(h, info, sizeof(int), hipMemcpyDeviceToHost, stream));
        sync();
        std::fprintf(stderr, "[corrla] jacobi_svd (split) l=%d sweeps=%d\n", (int)l, h[0]);
      }
      return;
    }
#define CORRLA_JACOBI(VL, G, E)                                                                                      \
  hipLaunchKernelGGL((k::jacobi_svd_kernel<T, VL, G, E>), dim3(1), dim3(1024), lds, stream, (const T*)c.p, c.ld, (int)l, \
                     vg, ldv, m1.p, m1.ld, m2.p, m2.ld, s_dev, (int)k, tol, tol_early, 40, info)
    // chunks per column = pitch / VW; G lanes x E chunks per lane must cover them; npairs <= 1024 / G for one round
    const int nchunk = k::jacobi_pitch((int)l, (int)(16 / sizeof(T))) / (int)(16 / sizeof(T));
    if (l <= 128 && nchunk <= 32) {
      if (v_lds) CORRLA_JACOBI(true, 16, 2); else CORRLA_JACOBI(false, 16, 2);
    } else if (nchunk <= 40) {
      if (v_lds) CORRLA_JACOBI(true, 8, 5); else CORRLA_JACOBI(false, 8, 5);
    } else if (nchunk <= 72) {
      if (v_lds) CORRLA_JACOBI(true, 8, 9); else CORRLA_JACOBI(false, 8, 9);
    } else {
      small_svd_block(c, l, k, m1, m2, s_dev);
      return;
    }
#undef CORRLA_JACOBI
    CORRLA_HIP(hipGetLastError());
    if (env_int("CORRLA_DEBUG", 0)) {
      int h[4] = {0, 0, 0, 0};
      CORRLA_HIP(hipMemcpyAsync(h, info, sizeof(int), hipMemcpyDeviceToHost, stream));
      sync();
      std::fprintf(stderr, "[corrla] jacobi_svd l=%d sweeps=%d v_in_lds=%d\n", (int)l, h[0], (int)v_lds);
    }
  }
  // skinny (rows x ncols) -> column-major destination, optionally transposed (ncols x rows)
  template <class T>
  void copy_out(const Skinny<T>& src, int64_t ncols, T* dst, int64_t ldd, bool transpose, bool to_host) {
    const int64_t rows = src.rows;
    if (!to_host) {
      launch_copy_out(src.p, src.ld, rows, ncols, dst, ldd, transpose);
      return;
    }
    if (!transpose) {
      CORRLA_HIP(hipMemcpy2DAsync(dst, (size_t)ldd * sizeof(T), src.p, (size_t)src.ld * sizeof(T),
                                  (size_t)rows * sizeof(T), (size_t)ncols, hipMemcpyDeviceToHost, stream));
    } else {
      T* tmp = (T*)alloc_bytes((size_t)rows * ncols * sizeof(T));
      launch_copy_out(src.p, src.ld, rows, ncols, tmp, ncols, true);
      CORRLA_HIP(hipMemcpy2DAsync(dst, (size_t)ldd * sizeof(T), tmp, (size_t)ncols * sizeof(T),
                                  (size_t)ncols * sizeof(T), (size_t)rows, hipMemcpyDeviceToHost, stream));
    }
    sync();
  }
  template <class T>
  void pack_strided(const T* src, int64_t rows, int64_t cols, int64_t rs, int64_t cs, T* dst, int64_t ldd) {
    const int64_t tiles_c = (cols + 31) / 32, tiles_r = (rows + 31) / 32;
    if (tiles_c * tiles_r > 0x7fffffff) throw Error(ST_EINVAL, "problem too large for the launch grid");
    dim3 grid((unsigned)(tiles_c * tiles_r));
    hipLaunchKernelGGL((k::pack_strided_kernel<T>), grid, dim3(256), 0, stream, src, rows, cols, rs, cs, dst, ldd, tiles_c);
    CORRLA_HIP(hipGetLastError());
  }

  // ---- implicit centring ---------------------------------------------------------
  template <class T>
  void weighted_colsum(const Skinny<T>& x, int64_t rows, const T* w, T* v) {
    const int ncols = (int)x.cols_alloc;
    const int nblk = (int)std::max<int64_t>(1, std::min<int64_t>(64, (rows + 4095) / 4096));
    double* partial = (double*)alloc_bytes(sizeof(double) * (size_t)nblk * (size_t)ncols);
    hipLaunchKernelGGL((k::wcolsum_partial_kernel<T>), dim3((unsigned)nblk, (unsigned)ncols), dim3(256), 0, stream,
                       (const T*)x.p, x.ld, rows, w, partial, ncols);
    hipLaunchKernelGGL((k::wcolsum_final_kernel<T>), dim3((unsigned)((ncols + 63) / 64)), dim3(64), 0, stream,
                       (const double*)partial, nblk, ncols, v);
    CORRLA_HIP(hipGetLastError());
  }
  template <class T>
  void rank1_sub(Skinny<T>& out, int64_t rows, const T* u, const T* v, const T* scale) {
    dim3 grid((unsigned)((rows + 255) / 256), (unsigned)out.cols_alloc);
    check_grid(grid);
    hipLaunchKernelGGL((k::rank1_sub_kernel<T>), grid, dim3(256), 0, stream, out.p, out.ld, rows, u, v, scale);
    CORRLA_HIP(hipGetLastError());
  }

  // sign convention: flip (u_i, v_i) so the largest-magnitude entry of v_i is positive
  template <class T>
  void fix_signs(Skinny<T>& v_ref, Skinny<T>& other, int64_t k) {
    hipLaunchKernelGGL((k::column_sign_apply_kernel<T>), dim3((unsigned)k), dim3(256), 0, stream, v_ref.p, v_ref.ld, v_ref.rows,
                       other.p, other.ld, other.rows);
    CORRLA_HIP(hipGetLastError());
  }
  template <class T>
  void fill_const(T* p, int64_t n, T v) {
    const int blocks = (int)std::min<int64_t>(2048, std::max<int64_t>(1, (n + 255) / 256));
    hipLaunchKernelGGL((k::fill_const_kernel<T>), dim3(blocks), dim3(256), 0, stream, p, n, v);
    CORRLA_HIP(hipGetLastError());
  }
  template <class T>
  void center_rows_cols(const T* in, int64_t rows, int64_t cols, int64_t ldi, const T* mu, bool along_cols, T* out,
                        int64_t ldo) {
    dim3 grid((unsigned)((cols + 255) / 256), (unsigned)std::min<int64_t>(rows, 4096));
    hipLaunchKernelGGL((k::center_kernel<T>), grid, dim3(256), 0, stream, in, rows, cols, ldi, mu, along_cols ? 1 : 0, out,
                       ldo);
    CORRLA_HIP(hipGetLastError());
  }

  // ---- elementwise / reductions ------------------------------------------------------------
  template <class T>
  void sumsq(const Skinny<T>& y, double* out_dev) {
    const int64_t n = y.ld * y.cols_alloc;  // padding is zero
    const int blocks = (int)std::min<int64_t>(1024, std::max<int64_t>(1, (n + 255) / 256));
    double* partial = (double*)alloc_bytes(sizeof(double) * blocks);
    hipLaunchKernelGGL((k::sumsq_partial_kernel<T>), dim3(blocks), dim3(256), 0, stream, y.p, n, partial);
    hipLaunchKernelGGL(k::sum_partials_kernel, dim3(1), dim3(64), 0, stream, partial, blocks, out_dev);
    CORRLA_HIP(hipGetLastError());
  }
  // ss_dev <- sum of squares of y, inv_dev <- 1 / sqrt(ss): the partial sums and ONE finishing launch
  template <class T>
  void inv_norm(const Skinny<T>& y, double* ss_dev, T* inv_dev) {
    const int64_t n = y.ld * y.cols_alloc;  // padding is zero
    const int blocks = (int)std::min<int64_t>(1024, std::max<int64_t>(1, (n + 255) / 256));
    double* partial = (double*)alloc_bytes(sizeof(double) * blocks);
    hipLaunchKernelGGL((k::sumsq_partial_kernel<T>), dim3(blocks), dim3(256), 0, stream, y.p, n, partial);
    hipLaunchKernelGGL((k::sum_partials_rsqrt_kernel<T>), dim3(1), dim3(64), 0, stream, (const double*)partial, blocks, ss_dev, inv_dev);
    CORRLA_HIP(hipGetLastError());
  }
  template <class T>
  void rsqrt_scalar(const double* ss_dev, T* out_dev) {
    hipLaunchKernelGGL((k::rsqrt_scalar_kernel<T>), dim3(1), dim3(1), 0, stream, ss_dev, out_dev);
    CORRLA_HIP(hipGetLastError());
  }
  template <class T>
  void scale_inplace(Skinny<T>& y, const T* scale_dev) {
    const int64_t n = y.ld * y.cols_alloc;
    const int blocks = (int)std::min<int64_t>(2048, std::max<int64_t>(1, (n + 255) / 256));
    hipLaunchKernelGGL((k::scale_kernel<T>), dim3(blocks), dim3(256), 0, stream, y.p, n, scale_dev);
    CORRLA_HIP(hipGetLastError());
  }
  template <class T>
  void fill_normal(T* p, int64_t rows, int64_t cols, int64_t rs, int64_t cs, uint64_t seed, int64_t row0,
                   int64_t global_cols) {
    const int64_t n = rows * cols;
    if (n <= 0) return;
    const int blocks = (int)std::min<int64_t>(8192, (n + 255) / 256);
    hipLaunchKernelGGL((k::fill_normal_kernel<T>), dim3(blocks), dim3(256), 0, stream, p, rows, cols, rs, cs, seed, row0,
                       global_cols, cs <= rs ? 1 : 0);
    CORRLA_HIP(hipGetLastError());
  }

  // ---- per-phase device times (corrla_timings) ------------------------------------------------------------
  // An event is recorded on the compute stream at every phase boundary; after the call has completed,
  // phase_resolve() adds the elapsed device time between consecutive events to the slot named at the later one.
  // phase_events_ off (corrla_ctx_set_phase_timings): only the first and the last event of a call are recorded --
  // total_ms stays, the per-phase slots stay zero -- because every event record is ~5 us of idle GPU between kernels
  // (9 of them per call: 1 % of a 5 ms step).  The two events around the sketch launch are always recorded.
  void set_phase_events(bool on) { phase_events_ = on; }
  void phase_end() {
    if (!phase_events_) phase_mark_impl(nullptr);
  }
  void phase_mark(double* slot) {
    if (!phase_events_ && slot != nullptr) return;
    phase_mark_impl(slot);
  }
  void phase_mark_impl(double* slot) {
    if (ev_used_ == ev_pool_.size()) {
      hipEvent_t e;
      CORRLA_HIP(hipEventCreate(&e));
      ev_pool_.push_back(e);
    }
    hipEvent_t e = ev_pool_[ev_used_++];
    CORRLA_HIP(hipEventRecord(e, stream));
    marks_.push_back({e, slot});
  }
  void phase_forget() {
    for (auto& m_ : marks_) m_.slot = nullptr;
  }
  // call after end_call(); *total (optional) receives first-mark -> last-mark
  void phase_resolve(double* total) {
    for (size_t i = 1; i < marks_.size(); ++i) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, marks_[i - 1].ev, marks_[i].ev) != hipSuccess) {
        (void)hipGetLastError();
        continue;
      }
      if (marks_[i].slot) *marks_[i].slot += (double)ms;
    }
    if (total && marks_.size() >= 2) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, marks_.front().ev, marks_.back().ev) == hipSuccess)
        *total = (double)ms;
      else
        (void)hipGetLastError();
    }
    marks_.clear();
  }

  // hipEvents on the compute stream around a kernel sequence of the current call (read after end_call)
  void event_mark(int i) {
    CORRLA_HIP(hipEventRecord(events_[i], stream));
    events_set_[i] = true;
  }
  double event_elapsed_ms(int i, int j) {
    if (!events_set_[i] || !events_set_[j]) return 0.0;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, events_[i], events_[j]) != hipSuccess) {
      (void)hipGetLastError();
      return 0.0;
    }
    return (double)ms;
  }

  // hipEvent timing of `reps` back-to-back sketch products on this stream
  template <class F>
  double time_on_stream(int reps, F&& f) {
    hipEvent_t e0, e1;
    CORRLA_HIP(hipEventCreate(&e0));
    CORRLA_HIP(hipEventCreate(&e1));
    f();  // warm-up (also pages in the code object)
    CORRLA_HIP(hipEventRecord(e0, stream));
    for (int i = 0; i < reps; ++i) f();
    CORRLA_HIP(hipEventRecord(e1, stream));
    CORRLA_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    CORRLA_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return (double)ms / reps;
  }

 private:
  struct Chunk {
    void* p = nullptr;
    size_t size = 0, used = 0;
    size_t zeroed = 0;  // zero pool only: bytes cleared by begin_call
  };
  std::vector<Chunk> chunks_;
  std::vector<Chunk> zchunks_;  // zero pool (alloc_zeroed)
  void* zero_page_ = nullptr;
  void* pinned_ = nullptr;  // staging for the small l x l transfers
  hipEvent_t events_[3] = {nullptr, nullptr, nullptr};
  bool events_set_[3] = {false, false, false};
  struct PhaseMark {
    hipEvent_t ev;
    double* slot;
  };
  std::vector<hipEvent_t> ev_pool_;
  size_t ev_used_ = 0;
  std::vector<PhaseMark> marks_;
  static constexpr size_t kPinnedBytes = (size_t)8 << 20;
  int split_nn_override_ = 0, split_tn_override_ = 0, mw_override_ = 0, gemm_debug_flags_ = 0;
  uint64_t entropy_ = 0, calls_ = 0, calls_sharded_ = 0;
  bool no_device_chol_ = false;
  int jmc_min_l_ = 96, jmc_max_b_ = 24, jmc_local_ = 1;
  int gemm_xcd_remap_ = 1;  // CORRLA_GEMM_XCD=0: plain block mapping in gemm_tn (see GemmArgs::xcd_remap)
  int persist_max_tiles_ = 16;
  const int* run_if_ = nullptr;
  bool phase_events_ = true;
  bool robust_qr_ = true;
  int robust_passes_ = 2;
  int jmc_extra_sweeps_ = 0, jmc_sweeps_hint_ = 0;
  bool jmc_force_v_ = false;
  int64_t tall_min_rows_ = 65536;
  int f64_mfma_waves_ = 8;
  int gemm_wide_mode_ = 0;
  int64_t gemm_wide_max_red_ = 2048;
  double mixed_min_work_ = 16777216.0;  // outer x reduction elements below which a product keeps the exact kernels

  static void check_grid(const dim3& g) {
    if (g.y > 65535u || g.z > 65535u) throw Error(ST_EINVAL, "problem too large for the launch grid");
  }

  template <class T, int NT>
  static void set_lds_attr_one() {
    const auto attr = hipFuncAttributeMaxDynamicSharedMemorySize;
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::gemm_nn_kernel<T, 1, NT>, attr, k::gemm_lds_bytes(1, NT)));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::gemm_tn_kernel<T, 1, NT>, attr, k::gemm_lds_bytes(1, NT)));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::gemm_nn_kernel<T, 2, NT>, attr, k::gemm_lds_bytes(2, NT)));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::gemm_tn_kernel<T, 2, NT>, attr, k::gemm_lds_bytes(2, NT)));
    if constexpr (NT <= 8)
      CORRLA_HIP(hipFuncSetAttribute((const void*)k::gemm_nn_kernel<T, 2, NT, true>, attr, 3 * k::big_tile_bytes(2)));
    if constexpr (std::is_same<T, double>::value) {  // two MFMA waves per SIMD on the MW = 2 tile (launch_mw)
      CORRLA_HIP(hipFuncSetAttribute((const void*)k::gemm_nn_kernel<T, 1, NT, false, 8>, attr, k::gemm_lds_bytes(2, NT)));
      CORRLA_HIP(hipFuncSetAttribute((const void*)k::gemm_tn_kernel<T, 1, NT, 8>, attr, k::gemm_lds_bytes(2, NT)));
    }
  }
  template <class T>
  static void set_jacobi_attrs() {
    const int lds = 160 * 1024;
    const auto attr = hipFuncAttributeMaxDynamicSharedMemorySize;
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jacobi_svd_kernel<T, true, 16, 2>, attr, lds));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jacobi_svd_kernel<T, false, 16, 2>, attr, lds));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jacobi_svd_kernel<T, true, 8, 5>, attr, lds));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jacobi_svd_kernel<T, false, 8, 5>, attr, lds));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jacobi_svd_kernel<T, true, 8, 9>, attr, lds));
    CORRLA_HIP(hipFuncSetAttribute((const void*)k::jacobi_svd_kernel<T, false, 8, 9>, attr, lds));
  }
  template <class T>
  static void set_lds_attrs() {
    set_lds_attr_one<T, 1>();
    set_lds_attr_one<T, 2>();
    set_lds_attr_one<T, 3>();
    set_lds_attr_one<T, 4>();
    set_lds_attr_one<T, 5>();
    set_lds_attr_one<T, 6>();
    set_lds_attr_one<T, 7>();
    set_lds_attr_one<T, 8>();
    set_lds_attr_one<T, 9>();
  }

  // Launch geometry: MW (16-wide outer tiles per wave), nsplit (split of the reduction into slabs).
  // One workgroup is resident per CU at the large column blockings, so aim for >= num_cus
  // workgroups; prefer the MW = 2 shape (fewer skinny-operand bytes per MFMA) whenever the
  // reduction is long enough to make up the workgroup count by splitting it.
  void choose_geometry(bool tn, int64_t outer_n, int nblk, int tiles_total, int* mw_out, int* nsplit_out) const {
    const int ov = tn ? split_tn_override_ : split_nn_override_;
    int mw = (outer_n >= 256 && tiles_total >= 8) ? 2 : 1;  // small outputs (Gram, core) use the MW = 1 instantiation
    if (mw_override_ > 0) mw = mw_override_;
    const int64_t outer_tiles = (outer_n + 64 * mw - 1) / (64 * mw);
    const int64_t wgs = outer_tiles * nblk;
    int ns = 1;
    if (ov > 0) {
      ns = ov;
    } else if (wgs < num_cus) {
      ns = (int)((num_cus + wgs - 1) / wgs);
      if (wgs * ns < 2 * (int64_t)num_cus && wgs < num_cus / 4) ns *= 2;  // small grids: two waves of WGs
      ns = std::min(ns, std::max(1, tiles_total / 4));
    }
    ns = std::max(1, std::min(ns, tiles_total));
    *mw_out = mw;
    *nsplit_out = std::min(ns, 65535);
  }

  template <class T, int MW, int NT, int NW = 4>
  void launch_one(bool tn, dim3 grid, const k::GemmArgs<T>& a) {
    constexpr int GW = MW * NW / 4;  // row tiles per SIMD: the tile geometry (hip_kernels.hpp: gemm_nn_kernel)
    // the kernels only touch ring buffers [0, min(tiles per workgroup, stages)): a short reduction (the l-deep
    // products Y * R^-1 and U = Q * U~ have 2-3 tiles) asks for less LDS, so several workgroups share a CU and one's
    // load latency hides behind another's MFMAs and stores
    const int64_t per_wg = (int64_t)std::max(1, a.tiles_per_split) * ((a.outer_blocks + (int64_t)grid.x - 1) / grid.x);
    const int lds = (int)std::min<int64_t>(k::gemm_stages(GW, NT), per_wg) * k::stage_bytes(GW, NT);
    const dim3 block(64 * (NW + k::kLoaders));  // MFMA waves + loader waves
    if (tn)
      hipLaunchKernelGGL((k::gemm_tn_kernel<T, MW, NT, NW>), grid, block, lds, stream, a);
    else
      hipLaunchKernelGGL((k::gemm_nn_kernel<T, MW, NT, false, NW>), grid, block, lds, stream, a);
  }
  template <class T>
  void launch_nt(bool tn, int mw, int nt, dim3 grid, const k::GemmArgs<T>& a) {
    switch (nt) {
      case 1: launch_mw<T, 1>(tn, mw, grid, a); break;
      case 2: launch_mw<T, 2>(tn, mw, grid, a); break;
      case 3: launch_mw<T, 3>(tn, mw, grid, a); break;
      case 4: launch_mw<T, 4>(tn, mw, grid, a); break;
      case 5: launch_mw<T, 5>(tn, mw, grid, a); break;
      case 6: launch_mw<T, 6>(tn, mw, grid, a); break;
      case 7: launch_mw<T, 7>(tn, mw, grid, a); break;
      case 8: launch_mw<T, 8>(tn, mw, grid, a); break;
      case 9: launch_mw<T, 9>(tn, mw, grid, a); break;
      default: throw Error(ST_EINVAL, "internal: bad column blocking");
    }
  }
  template <class T, int NT>
  void launch_alias(dim3 grid, const k::GemmArgs<T>& a) {
    const dim3 block(64 * (4 + k::kLoaders));
    hipLaunchKernelGGL((k::gemm_nn_kernel<T, 2, NT, true>), grid, block, 3 * k::big_tile_bytes(2), stream, a);
  }
  template <class T, int NT>
  void launch_mw(bool tn, int mw, dim3 grid, const k::GemmArgs<T>& a) {
    if (mw == 2) {
      // f64: the same 128-index tile with EIGHT MFMA waves of one row tile each -- two waves per SIMD keep the f64 matrix
      // pipe busier than one can (77.8 vs 60.5 TF register-only); CORRLA_F64_WAVES=4 keeps round 2's shape
      if constexpr (std::is_same<T, double>::value) {
        if (f64_mfma_waves_ == 8) return launch_one<T, 1, NT, 8>(tn, grid, a);
      }
      launch_one<T, 2, NT>(tn, grid, a);
    } else {
      launch_one<T, 1, NT>(tn, grid, a);
    }
  }

  // ---- tall_kernels.hpp: Y M and Y^T Y of a very tall sketch with l <= 96 (f32) / 64 (f64) ----
  template <class T, int K>
  void launch_tall_apply(dim3 grid, const k::TallApplyArgs<T>& g) {
    hipLaunchKernelGGL((k::tall_apply_kernel<T, K, K>), grid, dim3(256), 0, stream, g);
  }
  template <class T, int NCT>
  void launch_tall_gram(dim3 grid, const k::TallGramArgs<T>& g) {
    static bool attr_set = false;  // per instantiation; contexts are created under a process-wide lock
    const int lds = k::gram_lds_bytes(NCT, (int)sizeof(T));
    if (!attr_set) {
      CORRLA_HIP(hipFuncSetAttribute((const void*)k::tall_gram_kernel<T, NCT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      attr_set = true;
    }
    hipLaunchKernelGGL((k::tall_gram_kernel<T, NCT>), grid, dim3(256), lds, stream, g);
  }
  template <class T>
  bool launch_tall(bool tn, const Big<T>& r, const Skinny<T>& x, Skinny<T>& out, const T* scale_dev, int64_t outer_n,
                   int64_t red_n, const ColBlocking& cb) {
    constexpr int kMaxL = sizeof(T) == 4 ? 96 : 64;  // register budget of the B fragments
    constexpr int kVecElems = 16 / (int)sizeof(T);
    if (tall_min_rows_ <= 0) return false;
    if (tn) {
      // out (m x n2) = R^T X with R = Y^T stored row-major kdim x m: the columns of Y are contiguous
      const int64_t m = outer_n, kdim = red_n, n2 = x.cols;
      if (kdim > kMaxL || n2 > kMaxL || m < tall_min_rows_ || r.rows != kdim) return false;
      if (r.ld < round_up(m, 64) || (r.ld % kVecElems) || ((uintptr_t)r.p % 16)) return false;
      if (x.external || x.ld < kdim) return false;
      if (out.rows != m || out.ld < m || out.cols < n2) throw Error(ST_EINVAL, "internal: gemm output shape mismatch");
      const int kt = (int)std::max((kdim + 15) / 16, (n2 + 15) / 16);
      k::TallApplyArgs<T> g;
      g.y = r.p;
      g.m = m;
      g.ld_y = r.ld;
      g.kdim = (int)kdim;
      g.mat = x.p;
      g.ld_m = x.ld;
      g.n2 = (int)n2;
      g.out = out.p;
      g.ld_o = out.ld;
      g.out_cols = (int)(out.external ? out.cols : std::min<int64_t>(out.cols_alloc, 16 * kt));
      g.scale = scale_dev;
      g.vec_store = ((out.ld % kVecElems) == 0 && ((uintptr_t)out.p % 16) == 0) ? 1 : 0;
      g.run_if = run_if_;
      const int64_t nblocks = (m + 16 * kVecElems - 1) / (16 * kVecElems);
      dim3 grid((unsigned)std::min<int64_t>((nblocks + 3) / 4, num_cus));
      switch (kt) {
        case 1: launch_tall_apply<T, 1>(grid, g); break;
        case 2: launch_tall_apply<T, 2>(grid, g); break;
        case 3: launch_tall_apply<T, 3>(grid, g); break;
        case 4: launch_tall_apply<T, 4>(grid, g); break;
        default:
          if constexpr (sizeof(T) == 4) {
            if (kt == 5)
              launch_tall_apply<T, 5>(grid, g);
            else
              launch_tall_apply<T, 6>(grid, g);
          }
          break;
      }
      CORRLA_HIP(hipGetLastError());
      return true;
    }
    // G (l x l) = Y^T Y: both operands are the same column-major m x l memory
    const int64_t l = outer_n, m = red_n;
    if ((const void*)r.p != (const void*)x.p || r.ld != x.ld || l != x.cols || l > kMaxL || m < tall_min_rows_) return false;
    if ((r.ld % kVecElems) || ((uintptr_t)r.p % 16) || x.external || out.external) return false;
    const int nct = (int)((l + 15) / 16);
    if (out.ld < 16 * nct || out.cols_alloc < 16 * nct || cb.cols_alloc < 16 * nct) return false;
    if (out.rows != l) throw Error(ST_EINVAL, "internal: gemm output shape mismatch");
    constexpr int kRows = k::gram_rows<T>();
    const int64_t rows = x.ld;  // the padding rows are zero and may be read
    const int64_t want = std::max<int64_t>(1, std::min<int64_t>(num_cus, rows / (4 * kRows)));
    const int64_t rpg = round_up((rows + want - 1) / want, kRows);
    const int64_t ngroups = (rows + rpg - 1) / rpg;
    k::TallGramArgs<T> g;
    g.y = x.p;
    g.m = m;
    g.ld = x.ld;
    g.l = (int)l;
    g.slab_stride = (int64_t)out.ld * out.cols_alloc;
    g.slab = (T*)alloc_bytes((size_t)ngroups * (size_t)g.slab_stride * sizeof(T));
    g.out_ld = out.ld;
    g.rows_per_group = rpg;
    g.zero = (const T*)zero_page_;
    g.run_if = run_if_;
    dim3 grid((unsigned)ngroups);
    switch (nct) {
      case 1: launch_tall_gram<T, 1>(grid, g); break;
      case 2: launch_tall_gram<T, 2>(grid, g); break;
      case 3: launch_tall_gram<T, 3>(grid, g); break;
      case 4: launch_tall_gram<T, 4>(grid, g); break;
      default:
        if constexpr (sizeof(T) == 4) {
          if (nct == 5)
            launch_tall_gram<T, 5>(grid, g);
          else
            launch_tall_gram<T, 6>(grid, g);
        }
        break;
    }
    CORRLA_HIP(hipGetLastError());
    dim3 rg((unsigned)((l + 63) / 64), (unsigned)(16 * nct));
    hipLaunchKernelGGL((k::slab_reduce_deep_kernel<T>), rg, dim3(256), 0, stream, (const T*)g.slab, g.slab_stride, (int)ngroups,
                       out.p, out.ld, l, (int64_t)(16 * nct), scale_dev, run_if_);
    CORRLA_HIP(hipGetLastError());
    return true;
  }

  // out (outer_n x L) = scale * op(R) * X; `outer_n` = surviving dimension of R, `red_n` = reduced one
  template <class T>
  void launch_gemm(bool tn, const Big<T>& r, const Skinny<T>& x, Skinny<T>& out, const T* scale_dev, int64_t outer_n,
                   int64_t red_n) {
    constexpr int KT = k::MT<T>::KT;
    constexpr int VEC = k::MT<T>::VEC;
    const ColBlocking cb = col_blocking(x.cols);
    if (x.external) throw Error(ST_EINVAL, "internal: an external buffer cannot be a padded operand");
    if (cb.cols_alloc > x.cols_alloc || (!out.external && cb.cols_alloc > out.cols_alloc) || (out.external && out.cols < x.cols))
      throw Error(ST_EINVAL, "internal: skinny column padding too small for the column blocking");
    if (out.rows != outer_n || out.ld < outer_n) throw Error(ST_EINVAL, "internal: gemm output shape mismatch");
    if (((uintptr_t)r.p % 16) || (r.ld % VEC) || (r.cols_readable % VEC) || ((uintptr_t)x.p % 16))
      throw Error(ST_EINVAL, "internal: operand not 16-byte vector aligned");
    if (launch_tall<T>(tn, r, x, out, scale_dev, outer_n, red_n, cb)) return;
    const int64_t tiles64 = (red_n + KT - 1) / KT;
    if (tiles64 > 0x7fffffff) throw Error(ST_EINVAL, "reduction dimension too large");
    const int tiles_total = (int)tiles64;
    if (x.ld < (int64_t)tiles_total * KT) throw Error(ST_EINVAL, "internal: skinny leading dimension too small");
    int mw = 1, nsplit = 1;
    // an uneven column blocking (see below) runs its wide and its narrow blocks as two launches: each must fill the
    // chip by itself, so the reduction split is sized for the blocks of ONE launch
    const int n_wide0 = cb.tiles - cb.nblk * (cb.nt - 1);
    // (only where it pays: the second launch costs ~10 us, the skipped tile 1/18 of a product's time)
    const bool uneven0 = cb.nblk > 1 && n_wide0 < cb.nblk && cb.nt >= 2 && !env_int("CORRLA_EVEN_BLOCKS", 0) &&
                         (double)outer_n * (double)red_n * (double)cb.cols_alloc >= 1.0e10;
    choose_geometry(tn, outer_n, uneven0 ? std::max(1, std::min(n_wide0, cb.nblk - n_wide0)) : cb.nblk, tiles_total, &mw, &nsplit);
    // Gram matrix G = Y^T Y: both operands are the same memory and one outer tile (MW = 2: 128 indices) holds every
    // column -> the aliased instantiation stages Y once per tile
    const bool alias = !tn && (const void*)r.p == (const void*)x.p && r.ld == x.ld && cb.nblk == 1 && outer_n <= 128 &&
                       cb.nt <= 8 && outer_n == x.cols && !env_int("CORRLA_NO_GRAM_ALIAS", 0);
    if (alias) {
      mw = 2;
      const int64_t wgs1 = 1;
      nsplit = (int)std::min<int64_t>(std::max<int64_t>(1, tiles_total / 4), 2 * (int64_t)num_cus / wgs1);
      if (split_nn_override_ > 0) nsplit = std::min(split_nn_override_, tiles_total);
    }
    const int64_t outer_tiles = (outer_n + 64 * mw - 1) / (64 * mw);
    if (outer_tiles > 0x7fffffff) throw Error(ST_EINVAL, "outer dimension too large");
    k::GemmArgs<T> a;
    a.r = r.p;
    a.r_rows = r.rows;
    a.r_cols = r.cols;
    a.r_ld = r.ld;
    a.r_cols_readable = r.cols_readable;
    a.x = x.p;
    a.x_ld = x.ld;
    a.out = out.p;
    a.out_ld = out.ld;
    // columns this product may write: a caller's buffer has exactly `cols`; an uneven column blocking (below) never
    // computes the all-zero tail tile, which therefore stays as allocated (zero)
    const int n_wide = cb.tiles - cb.nblk * (cb.nt - 1);  // column blocks that really have cb.nt tiles
    const bool uneven = !alias && uneven0;
    a.out_cols = out.external ? out.cols : (uneven ? (int64_t)cb.tiles * 16 : cb.cols_alloc);
    a.scale = scale_dev;
    a.zero = (const T*)zero_page_;
    a.tiles_total = tiles_total;
    a.tiles_per_split = (tiles_total + nsplit - 1) / nsplit;
    a.nsplit = nsplit;
    a.debug_flags = gemm_debug_flags_;
    a.slab = nullptr;
    a.slab_stride = (int64_t)out.ld * cb.cols_alloc;
    if (nsplit > 1) a.slab = (T*)alloc_bytes((size_t)nsplit * (size_t)a.slab_stride * sizeof(T));
    a.outer_blocks = (int)outer_tiles;
    a.run_if = run_if_;
    a.vec_store = ((out.ld % 4) == 0 && ((uintptr_t)out.p % 16) == 0) ? 1 : 0;
    a.rotate = (!tn && !alias && a.tiles_per_split <= 32 && a.tiles_per_split > 1 && !env_int("CORRLA_GEMM_NO_ROTATE", 0)) ? 1 : 0;
    // Short reductions (A Z with n = 512: 8 tiles; Y R^-1: 2): a workgroup per outer tile spends a fifth of its life
    // waiting for its first tile.  A persistent launch -- as many workgroups as fit the chip at once, each walking its
    // outer tiles with the DMA ring running on across the boundaries -- pays that latency once.
    int64_t gx = outer_tiles;
    if (!alias && a.tiles_per_split <= persist_max_tiles_) {
      const int64_t lds_full = (int64_t)k::gemm_stages(mw, cb.nt) * k::stage_bytes(mw, cb.nt);
      const int64_t per_cu = std::max<int64_t>(1, std::min<int64_t>(4, (160 * 1024) / lds_full));
      const int64_t slots = per_cu * num_cus / ((int64_t)cb.nblk * nsplit);
      if (slots >= 1 && outer_tiles >= 3 * slots) gx = slots;
    }
    dim3 grid((unsigned)gx, (unsigned)cb.nblk, (unsigned)nsplit);
    check_grid(grid);
    // gemm_tn with a few outer tiles and a long, split reduction (A^T Y at n = 512): the outer tiles of one slab on one XCD
    // (per launch: the kernel's remap assumes gridDim.y == 1 -- the two launches of an uneven blocking qualify one by one)
    auto xcd_ok = [&](unsigned gy) {
      return (tn && gemm_xcd_remap_ && gy == 1 && gx == outer_tiles && outer_tiles >= 2 && outer_tiles <= 32 && nsplit >= 8) ? 1 : 0;
    };
    a.xcd_remap = xcd_ok((unsigned)cb.nblk);
    a.col_base = 0;
    // Uneven column blocking: `tiles` 16-column tiles over nblk blocks need not all be cb.nt wide -- 17 tiles (l = 266)
    // are 9 + 8, not 9 + 9: the narrower blocks run the next-smaller instantiation in a second launch and skip the
    // all-zero padding tile (5.5 % of the MFMA work of every tall product at l = 266).
    if (uneven) {
      dim3 g1((unsigned)gx, (unsigned)n_wide, (unsigned)nsplit);
      dim3 g2((unsigned)gx, (unsigned)(cb.nblk - n_wide), (unsigned)nsplit);
      a.xcd_remap = xcd_ok(g1.y);
      launch_nt<T>(tn, mw, cb.nt, g1, a);
      a.col_base = (int64_t)n_wide * cb.nt * 16;
      a.xcd_remap = xcd_ok(g2.y);
      launch_nt<T>(tn, mw, cb.nt - 1, g2, a);
    } else if (alias) {
      switch (cb.nt) {
        case 1: launch_alias<T, 1>(grid, a); break;
        case 2: launch_alias<T, 2>(grid, a); break;
        case 3: launch_alias<T, 3>(grid, a); break;
        case 4: launch_alias<T, 4>(grid, a); break;
        case 5: launch_alias<T, 5>(grid, a); break;
        case 6: launch_alias<T, 6>(grid, a); break;
        case 7: launch_alias<T, 7>(grid, a); break;
        default: launch_alias<T, 8>(grid, a); break;
      }
    } else {
      launch_nt<T>(tn, mw, cb.nt, grid, a);
    }
    CORRLA_HIP(hipGetLastError());
    if (nsplit >= 8) {
      dim3 rg((unsigned)((outer_n + 63) / 64), (unsigned)cb.cols_alloc);
      check_grid(rg);
      hipLaunchKernelGGL((k::slab_reduce_deep_kernel<T>), rg, dim3(256), 0, stream, (const T*)a.slab, a.slab_stride,
                         nsplit, out.p, out.ld, outer_n, a.out_cols, scale_dev, run_if_);
      CORRLA_HIP(hipGetLastError());
    } else if (nsplit > 1) {
      dim3 rg((unsigned)((outer_n + 255) / 256), (unsigned)cb.cols_alloc);
      check_grid(rg);
      hipLaunchKernelGGL((k::slab_reduce_kernel<T>), rg, dim3(256), 0, stream, (const T*)a.slab, a.slab_stride, nsplit,
                         out.p, out.ld, outer_n, a.out_cols, scale_dev, run_if_);
      CORRLA_HIP(hipGetLastError());
    }
  }

  template <class T>
  void launch_copy_out(const T* src, int64_t lds_, int64_t rows, int64_t cols, T* dst, int64_t ldd, bool transpose) {
    const int64_t tiles_c = (cols + 31) / 32, tiles_r = (rows + 31) / 32;
    if (tiles_c * tiles_r > 0x7fffffff) throw Error(ST_EINVAL, "problem too large for the launch grid");
    dim3 grid((unsigned)(tiles_c * tiles_r));
    hipLaunchKernelGGL((k::copy_out_kernel<T>), grid, dim3(256), 0, stream, src, lds_, rows, cols, dst, ldd,
                       transpose ? 1 : 0, tiles_c);
    CORRLA_HIP(hipGetLastError());
  }
};

}  // namespace corrla
