// gfx950 (MI355X / CDNA4) kernels of the RSVD hot path.  Written for wave64 + MFMA directly; no
// portability layer.
//
// Two GEMM kernels carry every m- or n-sized product of random_svd.rs:15-110:
//
//   gemm_nn :  Out (M x L, col-major) = R (M x K, row-major, streamed once) * X (K x L, col-major)
//   gemm_tn :  Out (K x L, col-major) = R^T * X,  R (M x K row-major, streamed once), X (M x L col-major)
//
// R is the big operand (A, or A^T's memory when A is column-major); X/Out are "skinny" matrices
// (L = rank + oversamples, padded to 16-column MFMA tiles) kept column-major with zero padding.
// With those two, A*Omega / A*Z (random_svd.rs:31,47-51), A^T*Y (:42-46), B^T = A^T*Q (:80), the
// Gram matrices of the orthonormalisation (:38,57) and U = Q*U~ (:92) are all covered for both
// memory layouts of A (see driver.hpp).
//
// MFMA: v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64.  Each workgroup = 4 waves (one per
// SIMD); wave w owns 16 consecutive "outer" indices (rows of R for nn, columns of R for tn) and
// ALL NT 16-column tiles of the skinny operand, so R is read from HBM exactly once per column
// block.  The accumulator tile is D[l-index][outer-index] (skinny operand on the MFMA A side),
// which makes the epilogue stores contiguous along the column-major output.
//
// LDS: both operands are staged by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction)
// into double-buffered images of 256-byte rows (64 f32 / 32 f64 along the reduction index), so
// every fragment read is one ds_read_b128 feeding 4 (f32) / 2 (f64) MFMA k-steps.  The DMA writes
// LDS linearly, so the bank-conflict swizzle (16-byte slot ^= row & 15) is applied to the per-lane
// SOURCE address and again on the read (cdna_hip_programming.md rule 21).  Out-of-range lanes of
// the big operand read a 16-byte zero page instead of being masked, which gives exact zero fill
// on both the reduction tail and the outer tail.  The k index inside an MFMA is a dummy
// summation index, so lane group kq of k-step j is fed element 16g+4kq+j (f32) / 8g+2kq+j (f64)
// of the tile row for BOTH operands.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace corrla {
namespace k {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <class T>
struct MT;
template <>
struct MT<float> {
  typedef f32x4 acc_t;
  typedef f32x4 vec_t;
  static constexpr int VEC = 4;  // elements per 16 bytes
  static constexpr int KT = 64;  // reduction elements per 256-byte LDS row
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = 4 * (lane >> 4) + reg
  static __device__ __forceinline__ int drow(int lane, int j) { return 4 * (lane >> 4) + j; }
  // tn-kernel big-operand tile: 64 reduction rows x 256 B; fragment rows of one 32-lane half are
  // 4 apart -> flip the 64-byte chunk bit
  static __device__ __forceinline__ int tswz(int row) { return ((row >> 2) & 1) << 2; }
};
template <>
struct MT<double> {
  typedef f64x4 acc_t;
  typedef f64x2 vec_t;
  static constexpr int VEC = 2;
  static constexpr int KT = 32;
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  static __device__ __forceinline__ int drow(int lane, int j) { return (lane >> 4) + 4 * j; }
  // tn-kernel big-operand tile: 32 reduction rows x 512 B; fragment rows of one half are 2 apart
  // -> flip the 128-byte chunk bit
  static __device__ __forceinline__ int tswz(int row) { return ((row >> 1) & 1) << 3; }
};

constexpr int kRowBytes = 256;            // LDS row of the k-contiguous images
constexpr int kOuterTile = 64;            // outer indices per workgroup (4 waves x 16)
constexpr int kBigTileBytes = 64 * 256;   // 16 KiB: big-operand tile per stage (both kernels, both dtypes)

__host__ __device__ constexpr int stage_bytes(int nt) { return kBigTileBytes + nt * 16 * kRowBytes; }
__host__ __device__ constexpr int gemm_lds_bytes(int nt) { return 2 * stage_bytes(nt); }

template <class T>
struct GemmArgs {
  const T* r;         // big operand, row-major
  int64_t r_rows, r_cols, r_ld, r_cols_readable;
  const T* x;         // skinny operand, column-major, zero padded
  int64_t x_ld;
  T* out;             // skinny result, column-major
  int64_t out_ld;
  T* slab;            // partial results when nsplit > 1: slab[z][col][outer]
  int64_t slab_stride;
  const T* scale;     // optional device scalar applied to the result (nsplit == 1 only)
  const T* zero;      // >= 16 bytes of zeros
  int tiles_total;    // reduction tiles (of KT elements)
  int tiles_per_split;
  int nsplit;
};

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Stage NT*16 rows (one per skinny column) x 256 B of a column-major skinny operand: row = column
// index, contiguous along the reduction index.  Never out of range (padding is allocated + zero).
template <class T, int NT>
__device__ __forceinline__ void stage_skinny(char* xt, const T* x, int64_t x_ld, int64_t col0, int64_t k0, int wave,
                                             int lane) {
  constexpr int VEC = MT<T>::VEC;
#pragma unroll
  for (int i = 0; i < NT; ++i) {  // NT*4 one-KiB chunks over 4 waves
    const int c = wave + 4 * i;
    const int row = 4 * c + (lane >> 4);
    const int ls = (lane & 15) ^ (row & 15);
    const T* src = x + (col0 + row) * x_ld + k0 + ls * VEC;
    glds16(src, xt + c * 1024);
  }
}

template <class T, int NT>
__device__ __forceinline__ void store_tile(const GemmArgs<T>& g, const typename MT<T>::acc_t (&acc)[NT], int64_t outer0,
                                           int64_t outer_limit, int64_t col0, int wave, int lane) {
  const int64_t outer = outer0 + 16 * wave + (lane & 15);
  if (outer >= outer_limit) return;
  T* dst;
  T sc = (T)1;
  if (g.nsplit > 1) {
    dst = g.slab + (int64_t)blockIdx.z * g.slab_stride;
  } else {
    dst = g.out;
    if (g.scale) sc = *g.scale;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t col = col0 + 16 * t + MT<T>::drow(lane, j);
      dst[col * g.out_ld + outer] = acc[t][j] * sc;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// gemm_nn: grid = (ceil(R_rows/64), column blocks, nsplit)
// ---------------------------------------------------------------------------------------------
template <class T, int NT>
__global__ __launch_bounds__(256) void gemm_nn_kernel(GemmArgs<T> g) {
  typedef typename MT<T>::acc_t acc_t;
  typedef typename MT<T>::vec_t vec_t;
  constexpr int VEC = MT<T>::VEC;
  constexpr int KT = MT<T>::KT;
  constexpr int STAGE = stage_bytes(NT);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t row0 = (int64_t)blockIdx.x * kOuterTile;
  const int64_t col0 = (int64_t)blockIdx.y * (NT * 16);
  const int t_begin = blockIdx.z * g.tiles_per_split;
  const int t_end = min(t_begin + g.tiles_per_split, g.tiles_total);
  const int nk = t_end - t_begin;

  acc_t acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (acc_t){0, 0, 0, 0};

  auto stage = [&](int buf, int kt) {
    char* rt = smem + buf * STAGE;
    const int64_t k0 = (int64_t)kt * KT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // 16 one-KiB chunks over 4 waves
      const int c = wave + 4 * i;
      const int row = 4 * c + (lane >> 4);
      const int ls = (lane & 15) ^ (row & 15);
      const int64_t grow = row0 + row;
      const int64_t kk = k0 + ls * VEC;
      const T* src = (grow < g.r_rows && kk < g.r_cols_readable) ? g.r + grow * g.r_ld + kk : g.zero;
      glds16(src, rt + c * 1024);
    }
    stage_skinny<T, NT>(rt + kBigTileBytes, g.x, g.x_ld, col0, k0, wave, lane);
  };

  auto compute = [&](int buf) {
    const char* rt = smem + buf * STAGE;
    const char* xt = rt + kBigTileBytes;
    const int c = lane & 15, kq = lane >> 4;
    const char* brow = rt + (16 * wave + c) * kRowBytes;
    const char* arow = xt + c * kRowBytes;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int so = ((4 * gq + kq) ^ c) << 4;
      const vec_t b = *(const vec_t*)(brow + so);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const vec_t a = *(const vec_t*)(arow + t * 16 * kRowBytes + so);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[t] = MT<T>::mma(a[j], b[j], acc[t]);
      }
    }
  };

  if (nk > 0) stage(0, t_begin);
  for (int i = 0; i < nk; ++i) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile i has landed for every wave; everyone is done reading buffer (i+1)&1
    if (i + 1 < nk) stage((i + 1) & 1, t_begin + i + 1);
    compute(i & 1);
  }
  store_tile<T, NT>(g, acc, row0, g.r_rows, col0, wave, lane);
}

// ---------------------------------------------------------------------------------------------
// gemm_tn: grid = (ceil(R_cols/64), column blocks, nsplit); reduction over the rows of R
// ---------------------------------------------------------------------------------------------
template <class T, int NT>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmArgs<T> g) {
  typedef typename MT<T>::acc_t acc_t;
  typedef typename MT<T>::vec_t vec_t;
  constexpr int VEC = MT<T>::VEC;
  constexpr int KT = MT<T>::KT;
  constexpr int STAGE = stage_bytes(NT);
  constexpr int RBT = 64 * (int)sizeof(T);  // bytes per row of the big tile (64 outer columns)
  constexpr int LPR = RBT / 16;             // lanes per row in one DMA instruction
  constexpr int RPC = 64 / LPR;             // rows per 1-KiB DMA chunk
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t n0 = (int64_t)blockIdx.x * kOuterTile;
  const int64_t col0 = (int64_t)blockIdx.y * (NT * 16);
  const int t_begin = blockIdx.z * g.tiles_per_split;
  const int t_end = min(t_begin + g.tiles_per_split, g.tiles_total);
  const int nk = t_end - t_begin;

  acc_t acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (acc_t){0, 0, 0, 0};

  auto stage = [&](int buf, int mt) {
    char* rt = smem + buf * STAGE;
    const int64_t m0 = (int64_t)mt * KT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = wave + 4 * i;
      const int row = RPC * c + lane / LPR;
      const int ls = (lane % LPR) ^ MT<T>::tswz(row);
      const int64_t grow = m0 + row;
      const int64_t nn = n0 + ls * VEC;
      const T* src = (grow < g.r_rows && nn < g.r_cols_readable) ? g.r + grow * g.r_ld + nn : g.zero;
      glds16(src, rt + c * 1024);
    }
    stage_skinny<T, NT>(rt + kBigTileBytes, g.x, g.x_ld, col0, m0, wave, lane);
  };

  auto compute = [&](int buf) {
    const char* rt = smem + buf * STAGE;
    const char* xt = rt + kBigTileBytes;
    const int c = lane & 15, kq = lane >> 4;
    const int ncol = 16 * wave + c;           // outer column inside the tile
    const int nls = ncol / VEC;               // its logical 16-byte slot
    const int noff = (ncol % VEC) * (int)sizeof(T);
    const char* arow = xt + c * kRowBytes;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int slot = 4 * gq + kq;
      T b[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const int mloc = slot * VEC + j;
        b[j] = *(const T*)(rt + mloc * RBT + ((nls ^ MT<T>::tswz(mloc)) << 4) + noff);
      }
      const int so = (slot ^ c) << 4;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const vec_t a = *(const vec_t*)(arow + t * 16 * kRowBytes + so);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[t] = MT<T>::mma(a[j], b[j], acc[t]);
      }
    }
  };

  if (nk > 0) stage(0, t_begin);
  for (int i = 0; i < nk; ++i) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (i + 1 < nk) stage((i + 1) & 1, t_begin + i + 1);
    compute(i & 1);
  }
  store_tile<T, NT>(g, acc, n0, g.r_cols, col0, wave, lane);
}

// out[col][i] = scale * sum_z slab[z][col][i], i < limit, col < ncols; fixed z order (deterministic)
template <class T>
__global__ void slab_reduce_kernel(const T* slab, int64_t slab_stride, int nsplit, T* out, int64_t ld, int64_t limit,
                                   int64_t ncols, const T* scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t col = blockIdx.y;
  if (i >= limit || col >= ncols) return;
  const int64_t off = col * ld + i;
  T s = 0;
  for (int z = 0; z < nsplit; ++z) s += slab[(int64_t)z * slab_stride + off];
  if (scale) s *= *scale;
  out[off] = s;
}

// ---- Frobenius norm pieces (random_svd.rs:53-55) ----------------------------------------------
template <class T>
__global__ void sumsq_partial_kernel(const T* y, int64_t n, double* partial) {
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double v = (double)y[i];
    s += v * v;
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  __shared__ double ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ void sum_partials_kernel(const double* partial, int n, double* out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (threadIdx.x == 0) *out = s;
}
template <class T>
__global__ void rsqrt_scalar_kernel(const double* ss, T* out) {
  const double v = *ss;
  *out = (T)(v > 0.0 ? 1.0 / sqrt(v) : 0.0);
}
template <class T>
__global__ void scale_kernel(T* y, int64_t n, const T* scale) {
  const T sc = *scale;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] *= sc;
}

// ---- random_mat_normal (mat_utils.rs:161-175): Philox4x32-10 + Box-Muller -------------------
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0;
    c[1] = n1;
    c[2] = n2;
    c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
template <class T>
__device__ __forceinline__ T normal_from_index(uint64_t idx, uint64_t seed);
template <>
__device__ __forceinline__ float normal_from_index<float>(uint64_t idx, uint64_t seed) {
  uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), 0u, 0u};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const float u1 = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = ((float)(c[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2);
}
template <>
__device__ __forceinline__ double normal_from_index<double>(uint64_t idx, uint64_t seed) {
  uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), 0u, 0u};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const uint64_t a = (((uint64_t)c[0] << 32) | c[1]) >> 11;
  const uint64_t b = (((uint64_t)c[2] << 32) | c[3]) >> 11;
  const double u1 = ((double)a + 0.5) * (1.0 / 9007199254740992.0);
  const double u2 = ((double)b + 0.5) * (1.0 / 9007199254740992.0);
  return sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
}
// element (i, j) -> p[i * rs + j * cs] = N(0,1) keyed by (seed, (row0 + i) * global_cols + j);
// consecutive threads walk the unit-stride direction
template <class T>
__global__ void fill_normal_kernel(T* p, int64_t rows, int64_t cols, int64_t rs, int64_t cs, uint64_t seed, int64_t row0,
                                   int64_t global_cols, int cols_fast) {
  const int64_t total = rows * cols;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t i, j;
    if (cols_fast) {
      i = t / cols;
      j = t - i * cols;
    } else {
      j = t / rows;
      i = t - j * rows;
    }
    p[i * rs + j * cs] = normal_from_index<T>((uint64_t)((row0 + i) * global_cols + j), seed);
  }
}

// ---- SVD of the l x l core (random_svd.rs:89) on the device -------------------------------------
// One-sided (Hestenes) Jacobi in ONE workgroup of 1024 threads: W = C lives in LDS (column-major,
// odd pitch), the accumulated right rotations V live in global memory (L2-resident, same CU).  A
// round-robin tournament gives n/2 disjoint column pairs per step; each pair is rotated by a 16-lane
// group (64 pairs in flight), with the three dot products reduced by xor-shuffles inside the group.
// On exit the columns of W are U_c * sigma and V = V_c with C = U_c diag(sigma) V_c^T.  The kernel
// sorts sigma descending and writes sigma[:k], V_c[:, :k] (-> m1) and U_c[:, :k] (-> m2) directly into
// the zero-padded skinny operands of the GEMMs that follow (U = Q * m1, V = Qb * m2), so the final
// stage needs no host round trip.
template <class T>
__global__ __launch_bounds__(1024) void jacobi_svd_kernel(const T* __restrict__ c, int64_t ldc, int l, T* vg, int64_t ldv,
                                                          T* m1, int64_t ld1, T* m2, int64_t ld2, T* s_out, int k, T tol,
                                                          int max_sweeps, int* info) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int LP = l | 1;
  T* w = (T*)smem;
  T* sigma = w + (size_t)l * LP;
  int* order = (int*)(sigma + l + 2);
  int* flag = order + l + 2;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < l * l; idx += 1024) {
    const int j = idx / l, i = idx - j * l;
    w[j * LP + i] = c[(int64_t)j * ldc + i];
    vg[(int64_t)j * ldv + i] = (i == j) ? (T)1 : (T)0;
  }
  __syncthreads();
  const int n = (l + 1) & ~1;  // players in the tournament (one dummy when l is odd)
  const int npairs = n / 2;
  const int group = tid >> 4, gl = tid & 15;
  int sweep = 0;
  for (; sweep < max_sweeps; ++sweep) {
    if (tid == 0) *flag = 0;
    __syncthreads();
    for (int step = 0; step < n - 1; ++step) {
      for (int pr = group; pr < npairs; pr += 64) {
        int p, q;
        if (pr == 0) {
          p = n - 1;
          q = step;
        } else {
          p = (step + pr) % (n - 1);
          q = (step - pr + (n - 1)) % (n - 1);
        }
        if (p > q) {
          const int t_ = p;
          p = q;
          q = t_;
        }
        if (q >= l) continue;  // dummy player
        T* wp = w + p * LP;
        T* wq = w + q * LP;
        T a = 0, b = 0, g = 0;
        for (int i = gl; i < l; i += 16) {
          const T x = wp[i], y = wq[i];
          a += x * x;
          b += y * y;
          g += x * y;
        }
#pragma unroll
        for (int msk = 1; msk < 16; msk <<= 1) {
          a += __shfl_xor(a, msk, 16);
          b += __shfl_xor(b, msk, 16);
          g += __shfl_xor(g, msk, 16);
        }
        const T ab = sqrt(a * b);
        if (!(fabs(g) > tol * ab) || ab == (T)0) continue;
        const T zeta = (b - a) / ((T)2 * g);
        const T t = (zeta >= (T)0 ? (T)1 : (T)-1) / (fabs(zeta) + sqrt((T)1 + zeta * zeta));
        const T cs = (T)1 / sqrt((T)1 + t * t);
        const T sn = cs * t;
        T* vp = vg + (int64_t)p * ldv;
        T* vq = vg + (int64_t)q * ldv;
        for (int i = gl; i < l; i += 16) {
          const T x = wp[i], y = wq[i];
          wp[i] = cs * x - sn * y;
          wq[i] = sn * x + cs * y;
          const T vx = vp[i], vy = vq[i];
          vp[i] = cs * vx - sn * vy;
          vq[i] = sn * vx + cs * vy;
        }
        if (gl == 0) *flag = 1;
      }
      __syncthreads();
    }
    const int rotated = *flag;
    __syncthreads();
    if (!rotated) break;
  }
  // singular values and descending order
  for (int j = group; j < l; j += 64) {
    T a = 0;
    for (int i = gl; i < l; i += 16) {
      const T x = w[j * LP + i];
      a += x * x;
    }
#pragma unroll
    for (int msk = 1; msk < 16; msk <<= 1) a += __shfl_xor(a, msk, 16);
    if (gl == 0) sigma[j] = sqrt(a);
  }
  __syncthreads();
  for (int j = tid; j < l; j += 1024) {
    const T sj = sigma[j];
    int r = 0;
    for (int i = 0; i < l; ++i) {
      const T si = sigma[i];
      r += (si > sj || (si == sj && i < j)) ? 1 : 0;
    }
    order[r] = j;
  }
  __syncthreads();
  for (int r = group; r < k; r += 64) {
    const int j = order[r];
    const T sj = sigma[j];
    const T inv = sj > (T)0 ? (T)1 / sj : (T)0;
    for (int i = gl; i < l; i += 16) {
      m2[(int64_t)r * ld2 + i] = w[j * LP + i] * inv;
      m1[(int64_t)r * ld1 + i] = vg[(int64_t)j * ldv + i];
    }
    if (gl == 0) s_out[r] = sj;
  }
  if (tid == 0) info[0] = sweep;
}
__host__ __device__ inline size_t jacobi_lds_bytes(int l, size_t esz) {
  return (size_t)l * (l | 1) * esz + (size_t)(l + 2) * esz + (size_t)(l + 2) * sizeof(int) + 64;
}

// ---- layout helpers --------------------------------------------------------------------------
// dst[r * ldd + c] = src[r * rs + c * cs]   (repack any strided matrix to padded row-major)
template <class T>
__global__ void pack_strided_kernel(const T* src, int64_t rows, int64_t cols, int64_t rs, int64_t cs, T* dst, int64_t ldd,
                                    int64_t tiles_c) {
  __shared__ T tile[32][33];
  // 32x32 tiles through LDS so both sides stay coalesced whichever stride is the unit one
  const int64_t r0 = ((int64_t)blockIdx.x / tiles_c) * 32, c0 = ((int64_t)blockIdx.x % tiles_c) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty = 0..7
  const bool col_fast = cs <= rs;
  for (int q = ty; q < 32; q += 8) {
    // read with the fast source direction on tx
    const int64_t r = col_fast ? r0 + q : r0 + tx;
    const int64_t c = col_fast ? c0 + tx : c0 + q;
    if (r < rows && c < cols) tile[r - r0][c - c0] = src[r * rs + c * cs];
  }
  __syncthreads();
  for (int q = ty; q < 32; q += 8) {
    const int64_t r = r0 + q, c = c0 + tx;
    if (r < rows && c < cols) dst[r * ldd + c] = tile[q][tx];
  }
}
// dst (col-major, ldd) <- src (col-major skinny, lds): plain copy or transpose
template <class T>
__global__ void copy_out_kernel(const T* src, int64_t lds_, int64_t rows, int64_t cols, T* dst, int64_t ldd, int transpose,
                                int64_t tiles_c) {
  __shared__ T tile[32][33];
  const int64_t r0 = ((int64_t)blockIdx.x / tiles_c) * 32, c0 = ((int64_t)blockIdx.x % tiles_c) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  if (!transpose) {
    for (int q = ty; q < 32; q += 8) {
      const int64_t r = r0 + tx, c = c0 + q;
      if (r < rows && c < cols) dst[c * ldd + r] = src[c * lds_ + r];
    }
    return;
  }
  for (int q = ty; q < 32; q += 8) {
    const int64_t r = r0 + tx, c = c0 + q;
    if (r < rows && c < cols) tile[q][tx] = src[c * lds_ + r];
  }
  __syncthreads();
  // dst is cols x rows column-major: dst[r * ldd + c]
  for (int q = ty; q < 32; q += 8) {
    const int64_t r = r0 + q, c = c0 + tx;
    if (r < rows && c < cols) dst[r * ldd + c] = tile[tx][q];
  }
}

}  // namespace k
}  // namespace corrla
