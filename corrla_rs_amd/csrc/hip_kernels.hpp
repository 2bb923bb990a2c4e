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

#include <limits>
#include <stdint.h>

#include <type_traits>

namespace corrla {
namespace k {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

#ifndef CORRLA_F32_BIG_IS_A
#define CORRLA_F32_BIG_IS_A true  // see MT<float>::kBigIsA
#endif
template <class T>
struct MT;
template <>
struct MT<float> {
  typedef f32x4 acc_t;
  typedef f32x4 vec_t;
  static constexpr int VEC = 4;  // elements per 16 bytes
  static constexpr int KT = 64;  // reduction elements per 256-byte LDS row
  // The tall GEMMs feed the BIG operand as MFMA A and the skinny one as B: a lane's four D registers are then four
  // consecutive OUTER indices of one result column, i.e. one 16-byte store (a quarter of the store instructions of
  // the other order; 10^7 x 80 results made the stores 10 % of the product's time).
  static constexpr bool kBigIsA = CORRLA_F32_BIG_IS_A;
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
  // f64 D registers are 4 rows apart: the skinny operand stays MFMA A, so that 16 lanes store 128 contiguous bytes
  static constexpr bool kBigIsA = false;
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  static __device__ __forceinline__ int drow(int lane, int j) { return (lane >> 4) + 4 * j; }
  // tn-kernel big-operand tile: 32 reduction rows x 512 B; fragment rows of one half are 2 apart
  // -> flip the 128-byte chunk bit
  static __device__ __forceinline__ int tswz(int row) { return ((row >> 1) & 1) << 3; }
};

constexpr int kRowBytes = 256;  // LDS row of the k-contiguous images
#ifndef CORRLA_PD
#define CORRLA_PD 4
#endif
#ifndef CORRLA_GEMM_DEFER
#define CORRLA_GEMM_DEFER 2  // pipeline steps whose MFMAs are issued after the next tile's barrier (0 = off)
#endif
constexpr int kPrefetchSteps = CORRLA_PD;  // LDS fragment reads run this many MFMA steps ahead
constexpr int kLoaders = 4;      // LDS-DMA loader waves per workgroup (besides the 4 MFMA waves); must divide 4
// A workgroup owns 64*MW outer indices (4 waves x MW 16-wide MFMA tiles each).  MW = 2 halves the
// skinny-operand bytes staged per MFMA (the per-CU global->LDS fill rate, ~11 B/clk, is what bounds
// the MW = 1 shape at 144 columns: 52 KiB per 4608 MFMA cycles); MW = 1 keeps small problems spread
// over more workgroups.
__host__ __device__ constexpr int outer_tile(int mw) { return 64 * mw; }
__host__ __device__ constexpr int big_tile_bytes(int mw) { return 64 * 256 * mw; }
__host__ __device__ constexpr int stage_bytes(int mw, int nt) { return big_tile_bytes(mw) + nt * 16 * kRowBytes; }
// LDS ring depth: 3 stages (the loaders run two tiles ahead, which hides the higher memory latency of the
// chip's low-clock state between bursts) whenever they fit in 160 KiB, else 2.
__host__ __device__ constexpr int gemm_stages(int mw, int nt) { return 3 * stage_bytes(mw, nt) <= 160 * 1024 ? 3 : 2; }
__host__ __device__ constexpr int gemm_lds_bytes(int mw, int nt) { return gemm_stages(mw, nt) * stage_bytes(mw, nt); }
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wg_barrier() { asm volatile("s_barrier" ::: "memory"); }

template <class T>
struct GemmArgs {
  const T* r;         // big operand, row-major
  int64_t r_rows, r_cols, r_ld, r_cols_readable;
  const T* x;         // skinny operand, column-major, zero padded
  int64_t x_ld;
  T* out;             // skinny result, column-major
  int64_t out_ld;
  int64_t out_cols;   // columns of `out` that may be written (the padded column count, or fewer for a caller's buffer)
  int64_t col_base;   // first column of this launch's column blocks (uneven column blockings take two launches)
  T* slab;            // partial results when nsplit > 1: slab[z][col][outer]
  int64_t slab_stride;
  const T* scale;     // optional device scalar applied to the result (nsplit == 1 only)
  const T* zero;      // >= 16 bytes of zeros
  int tiles_total;    // reduction tiles (of KT elements)
  int tiles_per_split;
  int nsplit;
  int debug_flags;    // timing-only ablation (wrong results): 1 = no DMA after the first tile
  const int* run_if;  // optional device word: the launch does nothing when it is 0 (conditional passes of the
                      // device-side Cholesky-QR, driver.hpp: orthonormalize_device)
  int vec_store;      // out (and the slabs) are 16-byte aligned with out_ld % 4 == 0
  int rotate;         // gemm_nn: workgroup x walks the reduction tiles from tile x mod (tiles per split), wrapping
  int outer_blocks;   // outer tiles in all; workgroup x of gridDim.x takes x, x + gridDim.x, ... (persistent launches of
                      // short reductions: the DMA ring runs on across the tile boundary, so only the first outer tile
                      // of a workgroup pays the fill latency)
  int xcd_remap;      // gemm_tn, gridDim.y == 1: the gridDim.x outer tiles of one reduction slab run on ONE XCD (see the
                      // kernel): they stream the same slab of the skinny operand, which then crosses the fabric once
};

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}


template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// A lean LDS-DMA stream for the loader wave: NVAR per-lane source pointers (one per variant of the
// repeating swizzle pattern), issued NQ times each with a uniform byte stride between repeats; per DMA
// the wave spends one 64-bit pointer add.  Chunk i = p + NVAR*q lands at lds + i KiB.
template <int NVAR, int NQ, int CS>
struct DmaStream {
  const char* ptr[NVAR];
  int64_t step;  // bytes between repeat q and q+1 of a variant
  int64_t adv;   // bytes per reduction tile
  // lds = address of this loader's first chunk; its chunks are CS KiB apart (CS = number of loader waves)
  __device__ __forceinline__ void issue(char* lds) {
    static_for<0, NQ>([&](auto iq) {
      static_for<0, NVAR>([&](auto ip) {
        constexpr int q = decltype(iq)::value, pv = decltype(ip)::value;
        glds16(ptr[pv], lds + CS * (pv + NVAR * q) * 1024);
        ptr[pv] += step;
      });
    });
    const int64_t back = adv - (int64_t)NQ * step;
#pragma unroll
    for (int pv = 0; pv < NVAR; ++pv) ptr[pv] += back;
  }
};

// ---- explicit LDS fragment pipeline (cdna_hip_programming.md 5.7) --------------------------------
// hipcc (ROCm 7.2) brackets every LDS fragment read of this loop with s_waitcnt lgkmcnt(0), which
// exposes the full LDS latency to the single wave per SIMD.  The fragment reads are therefore
// issued as inline-asm ds_read_* PD steps ahead of their MFMAs, and retired by hand-counted
// s_waitcnt lgkmcnt(N) statements that take the fragment as a "+v" operand: the data dependence
// keeps every consumer below its wait, and volatile asm keeps reads and waits in program order.
template <int OFF, class V>
__device__ __forceinline__ void lds_read_b128(V& d, unsigned addr) {
  static_assert(sizeof(V) == 16 && OFF >= 0 && OFF < 65536, "ds_read_b128 operand");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read_elem(float& d, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read_b32 offset");
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read_elem(double& d, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read_b64 offset");
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int N, class V>
__device__ __forceinline__ void lds_wait(V& frag) {
  static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag) : "n"(N) : "memory");
}
template <class V>
__device__ __forceinline__ void lds_tie(V& frag) {  // frag was retired by an earlier (in-order) wait
  asm volatile("" : "+v"(frag));
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(uintptr_t)((const __attribute__((address_space(3))) char*)p);
}

// skinny-operand element x, big-operand element r of one k index -> acc (see MT<T>::kBigIsA)
template <class T>
__device__ __forceinline__ typename MT<T>::acc_t gemm_mma(T x, T r, typename MT<T>::acc_t c) {
  if constexpr (MT<T>::kBigIsA)
    return MT<T>::mma(r, x, c);
  else
    return MT<T>::mma(x, r, c);
}

template <class T, int NT>
__device__ __forceinline__ void store_tile(const GemmArgs<T>& g, const typename MT<T>::acc_t (&acc)[NT], int64_t outer0,
                                           int64_t outer_limit, int64_t col0, int lane, int slab_z = (int)blockIdx.z) {
  if (g.debug_flags & 8) return;  // timing-only ablation: no result stores
  T* dst;
  T sc = (T)1;
  if (g.nsplit > 1) {
    dst = g.slab + (int64_t)slab_z * g.slab_stride;
  } else {
    dst = g.out;
    if (g.scale) {
      sc = *g.scale;
      // retire the load HERE: the stores below sit in branches of their own, and hipcc, not knowing across blocks whether
      // `sc` has landed, put an s_waitcnt vmcnt(0) in front of every one of them -- which also waits for the store before
      asm volatile("" : "+v"(sc));
    }
  }
  if constexpr (MT<T>::kBigIsA) {
    // D: column = lane & 15 (result column), rows 4 (lane >> 4) + j (outer indices)
    const int64_t outer = outer0 + 4 * (lane >> 4);
    if (outer >= outer_limit) return;
    const bool whole = g.vec_store && outer + 3 < outer_limit;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int64_t col = col0 + 16 * t + (lane & 15);
      if (g.nsplit > 1 || col < g.out_cols) {
        T* p = dst + col * g.out_ld + outer;
        if (whole) {
          *(typename MT<T>::acc_t*)p = acc[t] * sc;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (outer + j < outer_limit) p[j] = acc[t][j] * sc;
        }
      }
    }
  } else {
    const int64_t outer = outer0 + (lane & 15);
    if (outer >= outer_limit) return;
    // f64: D register j of column tile t is result column col0 + 16 t + 4 j + (lane >> 4), i.e. the 4 NT stores of a lane are
    // 4 out_ld elements apart: ONE running pointer.  (With `dst[col * out_ld + outer]` hipcc computed the 36 addresses
    // ahead of the reduction loop and parked them in scratch; every store then sat behind a scratch reload and an
    // s_waitcnt vmcnt(0) that also waited for the STORE before it -- ~50 us per outer tile, a third of the short-reduction
    // products of the f64 thin-Q.)
    static_assert(sizeof(T) == 8, "the scalar-store epilogue is the f64 layout (drow = (lane >> 4) + 4 j)");
    const int64_t colb = col0 + (lane >> 4);
    T* p = dst + colb * g.out_ld + outer;
    const int64_t step = 4 * g.out_ld;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t col = colb + 16 * t + 4 * j;
        if (g.nsplit > 1 || col < g.out_cols) *p = acc[t][j] * sc;
        p += step;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// gemm_nn: grid = (ceil(R_rows/64), column blocks, nsplit)
// ---------------------------------------------------------------------------------------------
// ALIAS (Gram matrices, G = Y^T Y: the big operand's memory IS the skinny operand's, and the single outer tile
// holds every column): the skinny fragments are read from the big tile's LDS image -- same rows, same swizzle -- so
// the operand is staged ONCE per tile instead of twice (the Gram of a 10^7 x 80 sketch read Y twice: 6.4 GB).
// NW = MFMA waves per workgroup: 4 (one per SIMD) or 8 (two per SIMD, each with MW row tiles; the tile geometry is that of
// GW = MW * NW / 4 row tiles per SIMD).  Two f64 MFMA waves per SIMD reach 77.8 TF where one reaches 60.5
// (tools/microbench/mfma_chain.hip): the f64 products run <double, 1, NT, false, 8> on the 128-row tile of <double, 2, NT>.
template <class T, int MW, int NT, bool ALIAS = false, int NW = 4>
__global__ __launch_bounds__(64 * (NW + kLoaders)) void gemm_nn_kernel(GemmArgs<T> g) {
  constexpr int GW = MW * NW / 4;
  static_assert(NW == 4 || (NW == 8 && !ALIAS), "4 or 8 MFMA waves");
  typedef typename MT<T>::acc_t acc_t;
  typedef typename MT<T>::vec_t vec_t;
  constexpr int VEC = MT<T>::VEC;
  constexpr int KT = MT<T>::KT;
  constexpr int STAGE = ALIAS ? big_tile_bytes(GW) : stage_bytes(GW, NT);
  constexpr int BIG = big_tile_bytes(GW);
  static_assert(!ALIAS || 16 * NT <= 64 * GW, "alias: the outer tile must hold every column");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (g.run_if && *g.run_if == 0) return;  // uniform over the grid
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t row_first = (int64_t)blockIdx.x * outer_tile(GW);
  const int64_t col0 = g.col_base + (int64_t)blockIdx.y * (NT * 16);
  const int t_begin = blockIdx.z * g.tiles_per_split;
  const int t_end = min(t_begin + g.tiles_per_split, g.tiles_total);
  const int nk = t_end - t_begin;
  // outer tiles of this workgroup: blockIdx.x, blockIdx.x + gridDim.x, ... (one when the grid covers them all)
  const int nob = (g.outer_blocks - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int64_t ob_rows = (int64_t)gridDim.x * outer_tile(GW);

  // Waves 0..3 are MFMA waves; waves 4.. (kLoaders of them) are LOADER waves that only issue the LDS-DMA of the next tile
  // (a global_load_lds costs its issuing wave ~100 cycles, which an in-order MFMA wave cannot hide: with
  // the DMA issued from the MFMA waves this kernel lost 14 %).  Chunk c (1 KiB = 4 tile rows): lane
  // (r = lane >> 4, s = lane & 15) fills physical 16-byte slot s of row 4c + r with logical slot
  // s ^ (row & 15).
  if (wave >= NW) {
    __builtin_amdgcn_s_setprio(3);  // few instructions, but they gate everyone: win the issue arbitration
    // generic (bounds-checked) issue of one tile: used for edge tiles only
    int64_t row0 = row_first;  // outer tile being STAGED (runs ahead of the one being multiplied)
    auto stage_checked = [&](int buf, int kt) {
      char* rt = smem + buf * STAGE;
      const int64_t k0 = (int64_t)kt * KT;
      for (int c = wave - NW; c < (ALIAS ? 4 * NT : 16 * GW); c += kLoaders) {
        const int row = 4 * c + (lane >> 4);
        const int ls = (lane & 15) ^ (row & 15);
        const int64_t grow = row0 + row;
        const int64_t kk = k0 + ls * VEC;
        const T* src = (grow < g.r_rows && kk < g.r_cols_readable) ? g.r + grow * g.r_ld + kk : g.zero;
        glds16(src, rt + c * 1024);
      }
      if constexpr (!ALIAS)
        for (int c = wave - NW; c < 4 * NT; c += kLoaders) {
          const int row = 4 * c + (lane >> 4);
          const int ls = (lane & 15) ^ (row & 15);
          glds16(g.x + (col0 + row) * g.x_ld + k0 + ls * VEC, rt + BIG + c * 1024);
        }
    };
    // pattern streams: chunk c covers tile rows 4c..4c+3 and (row & 15) repeats every 4 chunks; loader lw
    // of NL owns chunks c = lw + NL*i, i.e. NV = 4/NL pattern variants, each repeating every 16 rows
    constexpr int NL = kLoaders, NV = 4 / NL;
    const int lw = wave - NW;
    // alias: only the 16 * NT rows that exist as (zero padded) columns of the sketch are staged; the rows above them
    // feed accumulators whose outer index is >= r_rows and is never stored
    DmaStream<NV, ALIAS ? NT : 4 * GW, NL> big;
    DmaStream<NV, NT, NL> sk;
    // rotated reduction order (g.rotate): workgroup x starts at tile x mod nk and wraps.  With a short row (n = 512:
    // 2 KB) every workgroup of a launch would otherwise sit on the same 256-byte phase of its rows at the same time,
    // i.e. on one eighth of the memory channels.
    const int rot = g.rotate ? (int)(blockIdx.x % (unsigned)nk) : 0;
#pragma unroll
    for (int pv = 0; pv < NV; ++pv) {
      const int row = 4 * (lw + NL * pv) + (lane >> 4);
      const int ls = (lane & 15) ^ (row & 15);
      big.ptr[pv] = (const char*)(g.r + (row0 + row) * g.r_ld + (int64_t)(t_begin + rot) * KT + ls * VEC);
      sk.ptr[pv] = (const char*)(g.x + (col0 + row) * g.x_ld + (int64_t)(t_begin + rot) * KT + ls * VEC);
    }
    big.step = 16 * g.r_ld * (int64_t)sizeof(T);
    big.adv = KT * (int64_t)sizeof(T);
    sk.step = 16 * g.x_ld * (int64_t)sizeof(T);
    sk.adv = KT * (int64_t)sizeof(T);
    // alias: the host guarantees 16 * NT allocated columns (x.cols_alloc), all of them readable
    bool rows_inside = ALIAS ? true : (row0 + outer_tile(GW) <= g.r_rows);
    auto stage_tile = [&](int buf, int kt) {
      char* rt = smem + buf * STAGE;
      if (rows_inside && (int64_t)(kt + 1) * KT <= g.r_cols_readable) {
        if (g.debug_flags & 6) {  // timing-only ablations: 2 = no skinny-operand DMA, 4 = no big-operand DMA
          if (!(g.debug_flags & 4)) big.issue(rt + lw * 1024);
          if constexpr (!ALIAS)
            if (!(g.debug_flags & 2)) sk.issue(rt + BIG + lw * 1024);
          return;
        }
        big.issue(rt + lw * 1024);
        if constexpr (!ALIAS) sk.issue(rt + BIG + lw * 1024);
      } else {
        stage_checked(buf, kt);
#pragma unroll
        for (int pv = 0; pv < NV; ++pv) {  // keep the streams in step with the tile counter
          big.ptr[pv] += big.adv;
          sk.ptr[pv] += sk.adv;
        }
      }
    };
    // ring of NSTAGE buffers: tiles i+1 .. i+NSTAGE-1 are in flight while the MFMA waves work on tile i
    constexpr int NSTAGE = ALIAS ? 3 : gemm_stages(GW, NT);
    static_assert((16 * GW) % kLoaders == 0 && (4 * NT) % kLoaders == 0, "chunks must split evenly over the loaders");
    constexpr int DPL = ALIAS ? NT : (16 * GW + 4 * NT) / kLoaders;  // DMA instructions per loader wave per tile
    // the tiles of all outer tiles of this workgroup form ONE sequence through the ring
    const int nflat = nob * nk;
    const int64_t wrap = -(int64_t)nk * KT * (int64_t)sizeof(T);  // back to the first tile of the reduction range
    const int64_t big_jump = ob_rows * g.r_ld * (int64_t)sizeof(T);
    int s_k = rot, s_n = 0;
    auto stage_next = [&](int buf) {
      stage_tile(buf, t_begin + s_k);
      if (++s_k == nk) {
        s_k = 0;
#pragma unroll
        for (int pv = 0; pv < NV; ++pv) {
          big.ptr[pv] += wrap;
          sk.ptr[pv] += wrap;
        }
      }
      if (++s_n == nk) {  // on to the next outer tile (the streams are back at tile `rot`)
        s_n = 0;
        row0 += ob_rows;
#pragma unroll
        for (int pv = 0; pv < NV; ++pv) big.ptr[pv] += big_jump;
        if constexpr (!ALIAS) rows_inside = row0 + outer_tile(GW) <= g.r_rows;
      }
    };
    for (int t = 0; t < NSTAGE - 1 && t < nflat; ++t) stage_next(t % NSTAGE);
    for (int i = 0; i < nflat; ++i) {
      // tile i must have landed; the (NSTAGE-2) younger tiles may stay in flight (vmcnt counts in issue order)
      if (NSTAGE > 2 && i + NSTAGE - 2 < nflat) {
        if (g.debug_flags & 2)
          wait_vmcnt<(NSTAGE - 2) * (16 * GW / kLoaders)>();
        else if (g.debug_flags & 4)
          wait_vmcnt<(NSTAGE - 2) * (4 * NT / kLoaders)>();
        else
          wait_vmcnt<(NSTAGE - 2) * DPL>();
      } else {
        wait_vmcnt<0>();
      }
      wg_barrier();  // tile i visible to the MFMA waves; they are done reading buffer (i-1) % NSTAGE
      if (i + NSTAGE - 1 < nflat && !(g.debug_flags & 1)) stage_next((i + NSTAGE - 1) % NSTAGE);
    }
    return;
  }

  acc_t acc[MW][NT];
  // Fragment read addresses (bytes inside a stage), lane-invariant across tiles: for fragment group
  // gq this lane reads 16-byte slot (4*gq + kq) ^ c of its row in both images.
  const int fc = lane & 15, fkq = lane >> 4;
  const unsigned lds0 = lds_addr(smem);
  unsigned a_off[4], b_off[4];
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const unsigned so = (unsigned)(((4 * gq + fkq) ^ fc) << 4);
    a_off[gq] = lds0 + (ALIAS ? 0 : BIG) + fc * kRowBytes + so;  // alias: column c of the sketch = row c of the big tile
    b_off[gq] = lds0 + (16 * MW * wave + fc) * kRowBytes + so;
  }

  // Step st = gq*NT + t consumes one skinny fragment a(st) and the MW big-operand fragments b(gq) in
  // MW*VEC MFMAs.  Reads for step st+PD are issued before the MFMAs of step st; the wait for step
  // st leaves exactly the younger reads (steps st+1..st+PD) in flight.
  // The MFMAs of the last kDefer steps of a tile are issued AFTER the next tile's barrier and first fragment reads:
  // their operands are already in registers, so they fill the bubble (barrier -> LDS latency -> first MFMA) that
  // every tile otherwise pays.
  constexpr int NS = 4 * NT;
  constexpr int PD = kPrefetchSteps;
  constexpr int kDefer = (NT >= 2 && NS - CORRLA_GEMM_DEFER >= PD) ? (CORRLA_GEMM_DEFER < NT ? CORRLA_GEMM_DEFER : NT) : 0;
  vec_t afr[NS];
  vec_t bfr[4][MW];
  auto mfma_step = [&](auto ic) {
    constexpr int st = decltype(ic)::value;
    constexpr int gq = st / NT, t = st % NT;
#pragma unroll
    for (int mw = 0; mw < MW; ++mw)
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[mw][t] = gemm_mma<T>(afr[st][j], bfr[gq][mw][j], acc[mw][t]);
  };
  auto compute = [&](int buf, bool have_prev) {
    const unsigned sb = (unsigned)(buf * STAGE);
    // fragments of the previous tile's deferred steps: copied so that this tile's reads may reuse the arrays
    vec_t da[kDefer > 0 ? kDefer : 1];
    vec_t db[MW];
    if constexpr (kDefer > 0) {
#pragma unroll
      for (int d = 0; d < kDefer; ++d) da[d] = afr[NS - kDefer + d];
#pragma unroll
      for (int mw = 0; mw < MW; ++mw) db[mw] = bfr[3][mw];
    }
    auto read_step = [&](auto ic) {  // all reads that step `st` needs and that are not issued yet
      constexpr int st = decltype(ic)::value;
      constexpr int gq = st / NT, t = st % NT;
      if constexpr (t == 0) {
        static_for<0, MW>([&](auto im) {
          constexpr int mw = decltype(im)::value;
          lds_read_b128<mw * 16 * kRowBytes>(bfr[gq][mw], b_off[gq] + sb);
        });
      }
      lds_read_b128<t * 16 * kRowBytes>(afr[st], a_off[gq] + sb);
    };
    static_for<0, (PD < NS ? PD : NS)>(read_step);
    if constexpr (kDefer > 0) {
      if (have_prev) {
        static_for<0, kDefer>([&](auto id) {
          constexpr int d = decltype(id)::value;
          constexpr int t = (NS - kDefer + d) % NT;
          lds_tie(da[d]);  // pins these MFMAs after the barrier and the reads above
#pragma unroll
          for (int mw = 0; mw < MW; ++mw)
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[mw][t] = gemm_mma<T>(da[d][j], db[mw][j], acc[mw][t]);
        });
      }
    }
    static_for<0, NS>([&](auto ic) {
      constexpr int st = decltype(ic)::value;
      constexpr int gq = st / NT, t = st % NT;
      if constexpr (st + PD < NS) read_step(std::integral_constant<int, st + PD>{});
      // reads younger than a(st): steps st+1 .. min(st+PD, NS-1), each 1 read (+MW when it opens a group)
      constexpr int last = (st + PD < NS) ? st + PD : NS - 1;
      constexpr int groups = last / NT - gq;  // fragment groups opened by the younger steps
      constexpr int younger = (last - st) + groups * MW;
      lds_wait<(younger < 15 ? younger : 15)>(afr[st]);  // a smaller count only waits for more
      if constexpr (t == 0) {
        static_for<0, MW>([&](auto im) { lds_tie(bfr[gq][decltype(im)::value]); });
      }
      if constexpr (st < NS - kDefer) mfma_step(ic);
    });
  };

  constexpr int NSTAGE_C = ALIAS ? 3 : gemm_stages(GW, NT);
  int buf = 0;
  int64_t row0 = row_first;
  for (int ob = 0; ob < nob; ++ob, row0 += ob_rows) {
#pragma unroll
    for (int mw = 0; mw < MW; ++mw)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[mw][t] = (acc_t){0, 0, 0, 0};
    for (int i = 0; i < nk; ++i) {
      wg_barrier();  // matches the loaders' barrier: tile i is in LDS (all of this wave's LDS reads are retired)
      compute(buf, i > 0);
      buf = buf + 1 == NSTAGE_C ? 0 : buf + 1;
    }
    if constexpr (kDefer > 0) {
      if (nk > 0) static_for<NS - kDefer, NS>(mfma_step);
    }
#pragma unroll
    for (int mw = 0; mw < MW; ++mw)
      store_tile<T, NT>(g, acc[mw], row0 + 16 * MW * wave + 16 * mw, g.r_rows, col0, lane);
  }
}

// ---------------------------------------------------------------------------------------------
// gemm_tn: grid = (ceil(R_cols/64), column blocks, nsplit); reduction over the rows of R
// ---------------------------------------------------------------------------------------------
template <class T, int MW, int NT, int NW = 4>
__global__ __launch_bounds__(64 * (NW + kLoaders)) void gemm_tn_kernel(GemmArgs<T> g) {
  constexpr int GW = MW * NW / 4;  // see gemm_nn_kernel
  static_assert(NW == 4 || NW == 8, "4 or 8 MFMA waves");
  typedef typename MT<T>::acc_t acc_t;
  typedef typename MT<T>::vec_t vec_t;
  constexpr int VEC = MT<T>::VEC;
  constexpr int KT = MT<T>::KT;
  constexpr int STAGE = stage_bytes(GW, NT);
  constexpr int BIG = big_tile_bytes(GW);
  constexpr int RBT = 64 * GW * (int)sizeof(T);  // bytes per row of the big tile (64*GW outer columns)
  constexpr int LPR = RBT / 16;                  // lanes per row in one DMA instruction (16..128)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (g.run_if && *g.run_if == 0) return;  // uniform over the grid
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Which (outer tile, reduction slab) this workgroup takes.  Workgroups go to the eight XCDs round-robin in linear
  // block order (x fastest), so with the plain mapping the few outer tiles of ONE slab land on different XCDs and each
  // pulls that slab of the skinny operand through its own L2: at 1.25M x 512 (C4) Y crossed the fabric four times
  // (FETCH 4.16 GB for 2.96 GB algorithmic, 4.7 TB/s).  xcd_remap: of every 8 * gridDim.x consecutive workgroups, the
  // gridDim.x that share an XCD (ids congruent mod 8) take the outer tiles of one slab.
  int bx = (int)blockIdx.x, bz = (int)blockIdx.z;
  if (g.xcd_remap) {
    const int gx = (int)gridDim.x, per = 8 * gx;
    const int lin = bx + gx * bz, grp = lin / per, r = lin - grp * per;
    if ((grp + 1) * 8 <= (int)gridDim.z) {  // (a last, partial group of slabs keeps the plain mapping)
      bx = r >> 3;
      bz = grp * 8 + (r & 7);
    }
  }
  const int64_t n_first = (int64_t)bx * outer_tile(GW);
  const int64_t col0 = g.col_base + (int64_t)blockIdx.y * (NT * 16);
  const int t_begin = bz * g.tiles_per_split;
  const int t_end = min(t_begin + g.tiles_per_split, g.tiles_total);
  const int nk = t_end - t_begin;
  // outer tiles of this workgroup: bx, bx + gridDim.x, ... (see gemm_nn_kernel)
  const int nob = (g.outer_blocks - bx + (int)gridDim.x - 1) / (int)gridDim.x;
  const int64_t ob_cols = (int64_t)gridDim.x * outer_tile(GW);

  // The loader wave (4) issues the LDS-DMA, waves 0..3 run the MFMAs (see gemm_nn).  The big tile is a
  // row-linear LDS image of KT reduction rows x (64*MW outer columns); 16-byte slot lin = 64c + lane of
  // chunk c holds logical slot (lin % LPR) ^ tswz(row) of row lin / LPR.
  if (wave >= NW) {
    __builtin_amdgcn_s_setprio(3);
    int64_t n0 = n_first;  // outer tile being STAGED
    auto stage_checked = [&](int buf, int mt) {
      char* rt = smem + buf * STAGE;
      const int64_t m0 = (int64_t)mt * KT;
      for (int c = wave - NW; c < 16 * GW; c += kLoaders) {
        const int lin = c * 64 + lane;
        const int row = lin / LPR;
        const int lsb = (lin % LPR) ^ MT<T>::tswz(row);
        const int64_t grow = m0 + row;
        const int64_t nn = n0 + (int64_t)lsb * VEC;
        const T* src = (grow < g.r_rows && nn < g.r_cols_readable) ? g.r + grow * g.r_ld + nn : g.zero;
        glds16(src, rt + c * 1024);
      }
      for (int c = wave - NW; c < 4 * NT; c += kLoaders) {
        const int row = 4 * c + (lane >> 4);
        const int ls = (lane & 15) ^ (row & 15);
        glds16(g.x + (col0 + row) * g.x_ld + m0 + ls * VEC, rt + BIG + c * 1024);
      }
    };
    // big tile: the (row, swizzle) pattern of chunk c repeats every 8 chunks = 8*64/LPR tile rows; loader lw
    // of NL owns chunks c = lw + NL*i
    constexpr int NL = kLoaders, NVB = 8 / NL, NV = 4 / NL;
    const int lw = wave - NW;
    DmaStream<NVB, 2 * GW, NL> big;
    DmaStream<NV, NT, NL> sk;
#pragma unroll
    for (int pv = 0; pv < NVB; ++pv) {
      const int lin = (lw + NL * pv) * 64 + lane;
      const int row = lin / LPR;
      const int lsb = (lin % LPR) ^ MT<T>::tswz(row);
      big.ptr[pv] = (const char*)(g.r + ((int64_t)t_begin * KT + row) * g.r_ld + n0 + (int64_t)lsb * VEC);
    }
#pragma unroll
    for (int pv = 0; pv < NV; ++pv) {
      const int row = 4 * (lw + NL * pv) + (lane >> 4);
      const int ls = (lane & 15) ^ (row & 15);
      sk.ptr[pv] = (const char*)(g.x + (col0 + row) * g.x_ld + (int64_t)t_begin * KT + ls * VEC);
    }
    big.step = (int64_t)(8 * 64 / LPR) * g.r_ld * (int64_t)sizeof(T);
    big.adv = (int64_t)KT * g.r_ld * (int64_t)sizeof(T);
    sk.step = 16 * g.x_ld * (int64_t)sizeof(T);
    sk.adv = KT * (int64_t)sizeof(T);
    bool cols_inside = n0 + outer_tile(GW) <= g.r_cols_readable;
    auto stage_tile = [&](int buf, int mt) {
      char* rt = smem + buf * STAGE;
      if (cols_inside && (int64_t)(mt + 1) * KT <= g.r_rows) {
        big.issue(rt + lw * 1024);
        sk.issue(rt + BIG + lw * 1024);
      } else {
        stage_checked(buf, mt);
#pragma unroll
        for (int pv = 0; pv < NVB; ++pv) big.ptr[pv] += big.adv;
#pragma unroll
        for (int pv = 0; pv < NV; ++pv) sk.ptr[pv] += sk.adv;
      }
    };
    // ring of NSTAGE buffers: tiles i+1 .. i+NSTAGE-1 are in flight while the MFMA waves work on tile i
    constexpr int NSTAGE = gemm_stages(GW, NT);
    static_assert((16 * GW) % kLoaders == 0 && (4 * NT) % kLoaders == 0, "chunks must split evenly over the loaders");
    constexpr int DPL = (16 * GW + 4 * NT) / kLoaders;  // DMA instructions per loader wave per tile
    const int nflat = nob * nk;
    const int64_t big_jump = (ob_cols - (int64_t)nk * KT * g.r_ld) * (int64_t)sizeof(T);
    const int64_t sk_jump = -(int64_t)nk * KT * (int64_t)sizeof(T);
    int s_k = 0;
    auto stage_next = [&](int buf) {
      stage_tile(buf, t_begin + s_k);
      if (++s_k == nk) {  // on to the next outer tile
        s_k = 0;
        n0 += ob_cols;
#pragma unroll
        for (int pv = 0; pv < NVB; ++pv) big.ptr[pv] += big_jump;
#pragma unroll
        for (int pv = 0; pv < NV; ++pv) sk.ptr[pv] += sk_jump;
        cols_inside = n0 + outer_tile(GW) <= g.r_cols_readable;
      }
    };
    for (int t = 0; t < NSTAGE - 1 && t < nflat; ++t) stage_next(t % NSTAGE);
    for (int i = 0; i < nflat; ++i) {
      // tile i must have landed; the (NSTAGE-2) younger tiles may stay in flight (vmcnt counts in issue order)
      if (NSTAGE > 2 && i + NSTAGE - 2 < nflat)
        wait_vmcnt<(NSTAGE - 2) * DPL>();
      else
        wait_vmcnt<0>();
      wg_barrier();  // tile i visible to the MFMA waves; they are done reading buffer (i-1) % NSTAGE
      if (i + NSTAGE - 1 < nflat && !(g.debug_flags & 1)) stage_next((i + NSTAGE - 1) % NSTAGE);
    }
    return;
  }
  acc_t acc[MW][NT];

  // Fragment read addresses (bytes inside a stage), lane-invariant across tiles.  Skinny image: as in
  // gemm_nn.  Big image: element (reduction row mloc, outer column ncol) sits at
  // mloc*RBT + ((ncol/VEC) ^ tswz(mloc))*16 + (ncol%VEC)*sizeof(T); for this lane mloc =
  // (4*gq + kq)*VEC + j, and tswz(mloc) only depends on kq, so one base per mw plus immediates.
  const int fc = lane & 15, fkq = lane >> 4;
  const unsigned lds0 = lds_addr(smem);
  unsigned a_off[4], b_off[MW];
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) a_off[gq] = lds0 + BIG + fc * kRowBytes + (unsigned)(((4 * gq + fkq) ^ fc) << 4);
#pragma unroll
  for (int mw = 0; mw < MW; ++mw) {
    const int ncol = 16 * MW * wave + 16 * mw + fc;
    b_off[mw] = lds0 + (unsigned)(fkq * VEC * RBT + (((ncol / VEC) ^ MT<T>::tswz(fkq * VEC)) << 4) +
                                  (ncol % VEC) * (int)sizeof(T));
  }

  // same deferral of the last steps' MFMAs across the tile barrier as in gemm_nn_kernel
  constexpr int NS = 4 * NT;
  constexpr int PD = kPrefetchSteps;
  constexpr int kDefer = (NT >= 2 && NS - CORRLA_GEMM_DEFER >= PD) ? (CORRLA_GEMM_DEFER < NT ? CORRLA_GEMM_DEFER : NT) : 0;
  vec_t afr[NS];
  T bfr[4][MW][VEC];
  auto mfma_step = [&](auto ic) {
    constexpr int st = decltype(ic)::value;
    constexpr int gq = st / NT, t = st % NT;
#pragma unroll
    for (int mw = 0; mw < MW; ++mw)
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[mw][t] = gemm_mma<T>(afr[st][j], bfr[gq][mw][j], acc[mw][t]);
  };
  auto compute = [&](int buf, bool have_prev) {
    const unsigned sb = (unsigned)(buf * STAGE);
    vec_t da[kDefer > 0 ? kDefer : 1];
    T db[MW][VEC];
    if constexpr (kDefer > 0) {
#pragma unroll
      for (int d = 0; d < kDefer; ++d) da[d] = afr[NS - kDefer + d];
#pragma unroll
      for (int mw = 0; mw < MW; ++mw)
#pragma unroll
        for (int j = 0; j < VEC; ++j) db[mw][j] = bfr[3][mw][j];
    }
    auto read_step = [&](auto ic) {
      constexpr int st = decltype(ic)::value;
      constexpr int gq = st / NT, t = st % NT;
      if constexpr (t == 0) {
        static_for<0, MW * VEC>([&](auto iq) {
          constexpr int mw = decltype(iq)::value / VEC, j = decltype(iq)::value % VEC;
          lds_read_elem<(4 * gq * VEC + j) * RBT>(bfr[gq][mw][j], b_off[mw] + sb);
        });
      }
      lds_read_b128<t * 16 * kRowBytes>(afr[st], a_off[gq] + sb);
    };
    static_for<0, (PD < NS ? PD : NS)>(read_step);
    if constexpr (kDefer > 0) {
      if (have_prev) {
        static_for<0, kDefer>([&](auto id) {
          constexpr int d = decltype(id)::value;
          constexpr int t = (NS - kDefer + d) % NT;
          lds_tie(da[d]);
#pragma unroll
          for (int mw = 0; mw < MW; ++mw)
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[mw][t] = gemm_mma<T>(da[d][j], db[mw][j], acc[mw][t]);
        });
      }
    }
    static_for<0, NS>([&](auto ic) {
      constexpr int st = decltype(ic)::value;
      constexpr int gq = st / NT, t = st % NT;
      if constexpr (st + PD < NS) read_step(std::integral_constant<int, st + PD>{});
      constexpr int last = (st + PD < NS) ? st + PD : NS - 1;
      constexpr int groups = last / NT - gq;
      constexpr int younger = (last - st) + groups * MW * VEC;
      lds_wait<(younger < 15 ? younger : 15)>(afr[st]);  // a smaller count only waits for more
      if constexpr (t == 0) {
        static_for<0, MW * VEC>([&](auto iq) { lds_tie(bfr[gq][decltype(iq)::value / VEC][decltype(iq)::value % VEC]); });
      }
      if constexpr (st < NS - kDefer) mfma_step(ic);
    });
  };

  int buf = 0;
  int64_t n0 = n_first;
  for (int ob = 0; ob < nob; ++ob, n0 += ob_cols) {
#pragma unroll
    for (int mw = 0; mw < MW; ++mw)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[mw][t] = (acc_t){0, 0, 0, 0};
    for (int i = 0; i < nk; ++i) {
      wg_barrier();
      compute(buf, i > 0);
      buf = buf + 1 == gemm_stages(GW, NT) ? 0 : buf + 1;
    }
    if constexpr (kDefer > 0) {
      if (nk > 0) static_for<NS - kDefer, NS>(mfma_step);
    }
#pragma unroll
    for (int mw = 0; mw < MW; ++mw)
      store_tile<T, NT>(g, acc[mw], n0 + 16 * MW * wave + 16 * mw, g.r_cols, col0, lane, bz);
  }
}

// out[col][i] = scale * sum_z slab[z][col][i], i < limit, col < ncols; fixed summation tree (deterministic)
template <class T>
__global__ void slab_reduce_kernel(const T* slab, int64_t slab_stride, int nsplit, T* out, int64_t ld, int64_t limit,
                                   int64_t ncols, const T* scale, const int* run_if = nullptr) {
  if (run_if && *run_if == 0) return;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t col = blockIdx.y;
  if (i >= limit || col >= ncols) return;
  const int64_t off = col * ld + i;
  T s = 0;
  for (int z = 0; z < nsplit; ++z) s += slab[(int64_t)z * slab_stride + off];
  if (scale) s *= *scale;
  out[off] = s;
}
// many slabs, small output (the Gram matrices): 64 outputs x 4 slab groups per block, four loads in flight each
template <class T>
__global__ __launch_bounds__(256) void slab_reduce_deep_kernel(const T* slab, int64_t slab_stride, int nsplit, T* out,
                                                               int64_t ld, int64_t limit, int64_t ncols, const T* scale,
                                                               const int* run_if = nullptr) {
  if (run_if && *run_if == 0) return;
  __shared__ T part[4][64];
  const int li = threadIdx.x & 63, zg = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + li;
  const int64_t col = blockIdx.y;
  const int per = (nsplit + 3) / 4;
  const int z0 = zg * per, z1 = min(nsplit, z0 + per);
  T s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  if (i < limit && col < ncols) {
    const T* src = slab + col * ld + i;
    int z = z0;
    for (; z + 3 < z1; z += 4) {
      s0 += src[(int64_t)z * slab_stride];
      s1 += src[(int64_t)(z + 1) * slab_stride];
      s2 += src[(int64_t)(z + 2) * slab_stride];
      s3 += src[(int64_t)(z + 3) * slab_stride];
    }
    for (; z < z1; ++z) s0 += src[(int64_t)z * slab_stride];
  }
  part[zg][li] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (zg == 0 && i < limit && col < ncols) {
    T s = (part[0][li] + part[1][li]) + (part[2][li] + part[3][li]);
    if (scale) s *= *scale;
    out[col * ld + i] = s;
  }
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
// sum of the block partials and 1 / sqrt of it in one launch (the Z-side rescale of every power iteration)
template <class T>
__global__ void sum_partials_rsqrt_kernel(const double* partial, int n, double* ss_out, T* inv_out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (threadIdx.x == 0) {
    *ss_out = s;
    *inv_out = (T)(s > 0.0 ? 1.0 / sqrt(s) : 0.0);
  }
}
template <class T>
__global__ void rsqrt_scalar_kernel(const double* ss, T* out) {
  const double v = *ss;
  *out = (T)(v > 0.0 ? 1.0 / sqrt(v) : 0.0);
}
template <class T>
__global__ void sub_kernel(T* y, const T* __restrict__ p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] -= p[i];
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
// odd pitch) and so does the accumulated right-rotation matrix V when both fit (else V sits in
// global memory, L2-resident, same CU).  A round-robin tournament gives n/2 disjoint column pairs
// per step; each pair is rotated by a G-lane group (G = 16: 64 pairs in flight; G = 8: 128 pairs, the
// whole step in ONE round), lane gl owning the 16-byte chunks gl + G*e, e < E, of the four columns
// involved (ds_read_b128 / ds_write_b128: the scalar version was bound by LDS instruction issue).
// They are loaded once with independent LDS reads, the three dot products are reduced by xor-
// shuffles inside the group, and the rotation is applied in registers before the write-back.
// Convergence is quadratic, so the sweep in which every |w_p.w_q| / (|w_p||w_q|) was already below
// sqrt(tol) is the last one.  On exit the columns of W are U_c * sigma and V = V_c with
// C = U_c diag(sigma) V_c^T.  The kernel sorts sigma descending and writes sigma[:k], V_c[:, :k]
// (-> m1) and U_c[:, :k] (-> m2) directly into the zero-padded skinny operands of the GEMMs that
// follow (U = Q * m1, V = Qb * m2), so the final stage needs no host round trip.
// Sum over a G-lane group (G = 8 or 16, groups aligned to G lanes), result in every lane.  f32 uses DPP
// (quad_perm xor-1 / xor-2, row_half_mirror, row_mirror) on the VALU; __shfl_xor would go through the LDS
// crossbar (ds_bpermute), which the Jacobi kernels cannot afford.  f64 keeps the shuffles.
template <int G>
__device__ __forceinline__ float group_sum(float x) {
  static_assert(G == 8 || G == 16, "group size");
  auto dpp = [](float v, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  x += dpp(x, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
  x += dpp(x, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
  x += dpp(x, std::integral_constant<int, 0x141>{});  // row_half_mirror: lane i <-> 7 - i
  if constexpr (G == 16) x += dpp(x, std::integral_constant<int, 0x140>{});  // row_mirror: lane i <-> 15 - i
  return x;
}
template <int G>
__device__ __forceinline__ double group_sum(double x) {
#pragma unroll
  for (int msk = 1; msk < G; msk <<= 1) x += __shfl_xor(x, msk, G);
  return x;
}
// sqrt of a squared column norm for the final ranking of the Jacobi kernels: a non-finite value (non-finite input, reported
// through the status word) becomes 0, because comparisons with NaN are all false -- ranks would collide and the order
// table would keep uninitialised entries, i.e. wild column indices in the output loop
template <class T>
__device__ __forceinline__ T jacobi_safe_sigma(T a) {
  return (a >= (T)0 && a < (T)3.0e38) ? (T)sqrt(a) : (T)0;
}
// Jacobi rotation (cos, sin) that annihilates the off-diagonal g of [[a, g], [g, b]]; `rel` receives
// |g| / sqrt(a b).  f32 uses the single-instruction reciprocal / rsqrt (~1 ulp; the Jacobi kernels are
// VALU-issue bound and the IEEE sqrt/div sequences were ~1/4 of their instruction stream); f64 stays IEEE.
__device__ __forceinline__ bool jacobi_rotation(float a, float b, float g, float tol, float& cs, float& sn, float& rel,
                                                float& t) {
  // |g| / sqrt(a b) without forming a * b (two tiny columns would underflow it and never be rotated)
  const float rs = (a > 0.f && b > 0.f) ? __builtin_amdgcn_rsqf(a) * __builtin_amdgcn_rsqf(b) : 0.f;
  rel = fabsf(g) * rs;
  if (!(rel > tol)) return false;
  const float zeta = (b - a) * 0.5f * __builtin_amdgcn_rcpf(g);
  const float den = fabsf(zeta) + __builtin_amdgcn_sqrtf(1.f + zeta * zeta);
  t = copysignf(__builtin_amdgcn_rcpf(den), zeta);
  cs = __builtin_amdgcn_rsqf(1.f + t * t);
  sn = cs * t;
  // two columns of subnormal size (exact-arithmetic null directions of an integer-valued core): rcp(g) overflows
  // and (b - a) * inf is NaN or inf.  Such a pair carries no information; leave it alone rather than poison W.
  if (!(fabsf(sn) <= 1.f && cs <= 1.f)) return false;
  return true;
}
__device__ __forceinline__ bool jacobi_rotation(float a, float b, float g, float tol, float& cs, float& sn, float& rel) {
  float t;
  return jacobi_rotation(a, b, g, tol, cs, sn, rel, t);
}
// f64: the hardware reciprocal / reciprocal square root (~26 bits) refined by one Newton step each instead of the IEEE
// division / square-root sequences (3 + 3 of them per rotation, ~2/3 of the dependent chain of a Jacobi round in f64).
// The rotation only has to be orthogonal to rounding -- cs^2 + sn^2 = cs^2 (1 + t^2) is, whatever the last bits of t.
__device__ __forceinline__ double jr_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  return y * (2.0 - x * y);
}
__device__ __forceinline__ double jr_rsq(double x) {
  double y = __builtin_amdgcn_rsq(x);
  return y * (1.5 - 0.5 * x * y * y);
}
__device__ __forceinline__ bool jacobi_rotation(double a, double b, double g, double tol, double& cs, double& sn,
                                                double& rel, double& t) {
  // |g| / sqrt(a b) without forming a * b (as in f32)
  const double rs = (a > 0.0 && b > 0.0) ? jr_rsq(a) * jr_rsq(b) : 0.0;
  rel = fabs(g) * rs;
  if (!(rel > tol)) return false;
  const double zeta = (b - a) * 0.5 * jr_rcp(g);
  const double w = 1.0 + zeta * zeta;
  const double den = fabs(zeta) + w * jr_rsq(w);  // sqrt(w)
  t = copysign(jr_rcp(den), zeta);
  cs = jr_rsq(1.0 + t * t);
  sn = cs * t;
  if (!(fabs(sn) <= 1.0 && cs <= 1.0)) return false;  // subnormal-size columns / overflow of zeta: leave the pair alone
  return true;
}
__device__ __forceinline__ bool jacobi_rotation(double a, double b, double g, double tol, double& cs, double& sn,
                                                double& rel) {
  double t;
  return jacobi_rotation(a, b, g, tol, cs, sn, rel, t);
}
// pair (p < q) of slot `pr` in round `step` of the round-robin tournament on n players (no integer division)
__device__ __forceinline__ void tournament_pair(int n, int step, int pr, int& p, int& q) {
  if (pr == 0) {
    p = n - 1;
    q = step;
  } else {
    p = step + pr;
    if (p >= n - 1) p -= n - 1;
    q = step - pr;
    if (q < 0) q += n - 1;
  }
  if (p > q) {
    const int t_ = p;
    p = q;
    q = t_;
  }
}
// column pitch of the LDS-resident Jacobi images: multiple of the 16-byte vector width, and an odd number
// of 16-byte slots so consecutive columns start on different bank groups
__host__ __device__ inline int jacobi_pitch(int l, int vw) {
  int slots = (l + vw - 1) / vw;
  slots |= 1;
  return slots * vw;
}
// W + V images, sigma, order, flags
__host__ __device__ inline size_t jacobi_lds_bytes_fwd(int l, size_t esz) {
  const int vw = (int)(16 / esz);
  return (size_t)l * jacobi_pitch(l, vw) * esz * 2 + (size_t)(l + 2) * esz + (size_t)(l + 2) * sizeof(int) + 64;
}
template <class T, bool V_IN_LDS, int G, int E>
__global__ __launch_bounds__(1024) void jacobi_svd_kernel(const T* __restrict__ c, int64_t ldc, int l, T* vg, int64_t ldv,
                                                          T* m1, int64_t ld1, T* m2, int64_t ld2, T* s_out, int k, T tol,
                                                          T tol_early, int max_sweeps, int* info) {
  typedef typename MT<T>::vec_t vec_t;
  constexpr int VW = MT<T>::VEC;  // elements per 16-byte LDS access
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NG = 1024 / G;  // groups = pairs in flight
  // Column pitch: a multiple of the vector width (16-byte aligned columns); the pad rows stay zero, so
  // whole 16-byte chunks are loaded, rotated and stored without per-element masks.
  const int LP = jacobi_pitch(l, VW);
  const int nchunk = LP / VW;
  T* w = (T*)smem;
  T* sigma = w + (size_t)l * LP * (V_IN_LDS ? 2 : 1);
  int* order = (int*)(sigma + l + 2);
  int* flag = order + l + 2;  // flag[0]: rotated this sweep, flag[1]: some pair above tol_early
  if (V_IN_LDS) {
    vg = w + (size_t)l * LP;
    ldv = LP;
  }
  const int tid = threadIdx.x;
  for (int idx = tid; idx < l * LP; idx += 1024) {
    const int j = idx / LP, i = idx - j * LP;
    w[j * LP + i] = (i < l) ? c[(int64_t)j * ldc + i] : (T)0;
    vg[(int64_t)j * ldv + i] = (i == j) ? (T)1 : (T)0;
  }
  __syncthreads();
  const int n = (l + 1) & ~1;  // players in the tournament (one dummy when l is odd)
  const int npairs = n / 2;
  const int group = tid / G, gl = tid % G;
  int sweep = 0;
  for (; sweep < max_sweeps; ++sweep) {
    if (tid < 2) flag[tid] = 0;
    __syncthreads();
    for (int step = 0; step < n - 1; ++step) {
      for (int pr = group; pr < npairs; pr += NG) {
        int p, q;
        tournament_pair(n, step, pr, p, q);
        if (q >= l) continue;  // the dummy player (uniform within the group)
        vec_t* wp = (vec_t*)(w + p * LP);
        vec_t* wq = (vec_t*)(w + q * LP);
        vec_t* vp = (vec_t*)(vg + (int64_t)p * ldv);
        vec_t* vq = (vec_t*)(vg + (int64_t)q * ldv);
        vec_t x[E], y[E], vx[E], vy[E];
        vec_t zero;
#pragma unroll
        for (int z = 0; z < VW; ++z) zero[z] = (T)0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int ch = gl + G * e;
          const bool in = ch < nchunk;
          x[e] = in ? wp[ch] : zero;
          y[e] = in ? wq[ch] : zero;
          vx[e] = in ? vp[ch] : zero;
          vy[e] = in ? vq[ch] : zero;
        }
        T a = 0, b = 0, g = 0;
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
          for (int z = 0; z < VW; ++z) {
            a += x[e][z] * x[e][z];
            b += y[e][z] * y[e][z];
            g += x[e][z] * y[e][z];
          }
        a = group_sum<G>(a);
        b = group_sum<G>(b);
        g = group_sum<G>(g);
        T cs, sn, rel;
        if (jacobi_rotation(a, b, g, tol, cs, sn, rel)) {
#pragma unroll
          for (int e = 0; e < E; ++e) {
            const int ch = gl + G * e;
            if (ch < nchunk) {
              wp[ch] = cs * x[e] - sn * y[e];
              wq[ch] = sn * x[e] + cs * y[e];
              vp[ch] = cs * vx[e] - sn * vy[e];
              vq[ch] = sn * vx[e] + cs * vy[e];
            }
          }
          if (gl == 0) {
            flag[0] = 1;
            if (rel > tol_early) flag[1] = 1;
          }
        }
      }
      __syncthreads();
    }
    const int rotated = flag[0], big = flag[1];
    __syncthreads();
    if (!rotated || !big) {
      if (rotated) ++sweep;  // this sweep did (small, final) rotations
      break;
    }
  }
  // singular values and descending order
  for (int j = group; j < l; j += NG) {
    T a = 0;
    for (int i = gl; i < l; i += G) {
      const T xx = w[j * LP + i];
      a += xx * xx;
    }
#pragma unroll
    for (int msk = 1; msk < G; msk <<= 1) a += __shfl_xor(a, msk, G);
    if (gl == 0) sigma[j] = jacobi_safe_sigma(a);  // NaN-safe: the ranking below must stay a permutation
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
  for (int r = group; r < k; r += NG) {
    const int j = order[r];
    const T sj = sigma[j];
    const T inv = sj > (T)0 ? (T)1 / sj : (T)0;
    for (int i = gl; i < l; i += G) {
      m2[(int64_t)r * ld2 + i] = w[j * LP + i] * inv;
      m1[(int64_t)r * ld1 + i] = vg[(int64_t)j * ldv + i];
    }
    if (gl == 0) s_out[r] = sj;
  }
  if (tid == 0) info[0] = sweep;
}
// Role-split variant of jacobi_svd_kernel for l <= 144 with W and V both in LDS: V never feeds back into
// the iteration, so waves 0..8 (72 eight-lane groups) compute the rotations and update W only, publishing
// (cos, sin) per pair in a double-buffered LDS table, while waves 9..15 (112 four-lane groups) apply the
// PREVIOUS step's rotations to V.  The per-step critical path is the W half; the V half runs beside it.
template <class T>
__global__ __launch_bounds__(1024) void jacobi_svd_split_kernel(const T* __restrict__ c, int64_t ldc, int l, T* m1,
                                                                int64_t ld1, T* m2, int64_t ld2, T* s_out, int k, T tol,
                                                                T tol_early, int max_sweeps, int* info) {
  typedef typename MT<T>::vec_t vec_t;
  constexpr int VW = MT<T>::VEC;
  constexpr int GW = 8, EW = 5;   // W groups: 8 lanes x 5 chunks  (>= 36 chunks of 16 bytes)
  constexpr int GV = 4, EV = 9;   // V groups: 4 lanes x 9 chunks
  constexpr int NWG = 72;         // W groups (waves 0..8)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int LP = jacobi_pitch(l, VW);
  const int nchunk = LP / VW;
  T* w = (T*)smem;
  T* v = w + (size_t)l * LP;
  T* sigma = v + (size_t)l * LP;
  int* order = (int*)(sigma + l + 2);
  int* flag = order + l + 2;
  T* rot = (T*)(((uintptr_t)(flag + 4) + 15) & ~(uintptr_t)15);  // [2][NWG][2]
  const int tid = threadIdx.x;
  for (int idx = tid; idx < l * LP; idx += 1024) {
    const int j = idx / LP, i = idx - j * LP;
    w[idx] = (i < l) ? c[(int64_t)j * ldc + i] : (T)0;
    v[idx] = (i == j) ? (T)1 : (T)0;
  }
  __syncthreads();
  const int n = (l + 1) & ~1;
  const int npairs = n / 2;  // <= NWG
  const bool is_w = tid < NWG * GW;
  const int wgroup = tid / GW, wl = tid % GW;
  const int vgroup = (tid - NWG * GW) / GV, vl = (tid - NWG * GW) % GV;
  auto pair_of = [&](int step, int pr, int& p, int& q) { tournament_pair(n, step, pr, p, q); };
  vec_t zero;
#pragma unroll
  for (int z = 0; z < VW; ++z) zero[z] = (T)0;
  // V worker: apply the rotations published for (pstep) from table buffer tb
  auto v_pass = [&](int pstep, int tb) {
    if (is_w || vgroup >= npairs) return;
    int p, q;
    pair_of(pstep, vgroup, p, q);
    if (q >= l) return;
    const T cs = rot[(tb * NWG + vgroup) * 2], sn = rot[(tb * NWG + vgroup) * 2 + 1];
    if (sn == (T)0) return;
    vec_t* vp = (vec_t*)(v + p * LP);
    vec_t* vq = (vec_t*)(v + q * LP);
    vec_t vx[EV], vy[EV];
#pragma unroll
    for (int e = 0; e < EV; ++e) {
      const int ch = vl + GV * e;
      const bool in = ch < nchunk;
      vx[e] = in ? vp[ch] : zero;
      vy[e] = in ? vq[ch] : zero;
    }
#pragma unroll
    for (int e = 0; e < EV; ++e) {
      const int ch = vl + GV * e;
      if (ch < nchunk) {
        vp[ch] = cs * vx[e] - sn * vy[e];
        vq[ch] = sn * vx[e] + cs * vy[e];
      }
    }
  };
  int sweep = 0;
  int gs = 0;  // global step counter (table parity)
  int last_step = -1;
  for (; sweep < max_sweeps; ++sweep) {
    if (tid < 2) flag[tid] = 0;
    __syncthreads();
    for (int step = 0; step < n - 1; ++step, ++gs) {
      if (is_w) {
        if (wgroup < npairs) {
          int p, q;
          pair_of(step, wgroup, p, q);
          T cs = (T)1, sn = (T)0;
          if (q < l) {
            vec_t* wp = (vec_t*)(w + p * LP);
            vec_t* wq = (vec_t*)(w + q * LP);
            vec_t x[EW], y[EW];
#pragma unroll
            for (int e = 0; e < EW; ++e) {
              const int ch = wl + GW * e;
              const bool in = ch < nchunk;
              x[e] = in ? wp[ch] : zero;
              y[e] = in ? wq[ch] : zero;
            }
            T a = 0, b = 0, g = 0;
#pragma unroll
            for (int e = 0; e < EW; ++e)
#pragma unroll
              for (int z = 0; z < VW; ++z) {
                a += x[e][z] * x[e][z];
                b += y[e][z] * y[e][z];
                g += x[e][z] * y[e][z];
              }
            a = group_sum<GW>(a);
            b = group_sum<GW>(b);
            g = group_sum<GW>(g);
            T rel;
            if (jacobi_rotation(a, b, g, tol, cs, sn, rel)) {
#pragma unroll
              for (int e = 0; e < EW; ++e) {
                const int ch = wl + GW * e;
                if (ch < nchunk) {
                  wp[ch] = cs * x[e] - sn * y[e];
                  wq[ch] = sn * x[e] + cs * y[e];
                }
              }
              if (wl == 0) {
                flag[0] = 1;
                if (rel > tol_early) flag[1] = 1;
              }
            } else {
              cs = (T)1;
              sn = (T)0;
            }
          }
          if (wl == 0) {
            rot[((gs & 1) * NWG + wgroup) * 2] = cs;
            rot[((gs & 1) * NWG + wgroup) * 2 + 1] = sn;
          }
        }
      } else if (last_step >= 0) {
        v_pass(last_step, (gs - 1) & 1);
      }
      last_step = step;
      __syncthreads();
    }
    const int rotated = flag[0], big = flag[1];
    __syncthreads();
    if (!rotated || !big) {
      if (rotated) ++sweep;
      break;
    }
  }
  // drain: the rotations of the last step have not reached V yet
  if (last_step >= 0) v_pass(last_step, (gs - 1) & 1);
  __syncthreads();
  // singular values, order, outputs
  const int group = tid >> 4, gl = tid & 15;
  for (int j = group; j < l; j += 64) {
    T a = 0;
    for (int i = gl; i < l; i += 16) {
      const T xx = w[j * LP + i];
      a += xx * xx;
    }
#pragma unroll
    for (int msk = 1; msk < 16; msk <<= 1) a += __shfl_xor(a, msk, 16);
    if (gl == 0) sigma[j] = jacobi_safe_sigma(a);  // NaN-safe: the ranking below must stay a permutation
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
      m1[(int64_t)r * ld1 + i] = v[j * LP + i];
    }
    if (gl == 0) s_out[r] = sj;
  }
  if (tid == 0) info[0] = sweep;
}
__host__ __device__ inline size_t jacobi_split_lds_bytes(int l, size_t esz) {
  return jacobi_lds_bytes_fwd(l, esz) + 16 + 2 * 72 * 2 * esz;
}

__host__ __device__ inline size_t jacobi_lds_bytes(int l, size_t esz, bool v_in_lds) {
  const int vw = (int)(16 / esz);
  return (size_t)l * jacobi_pitch(l, vw) * esz * (v_in_lds ? 2 : 1) + (size_t)(l + 2) * esz +
         (size_t)(l + 2) * sizeof(int) + 64;
}
// ---- ring Jacobi: columns resident in registers --------------------------------------------------
// Same one-sided Jacobi, but the two columns a processor (8 lanes) works on stay in REGISTERS (W and V,
// lane g owning the 16-byte row chunks g, g + 8, ...), and the pairs follow the odd-even transposition
// ordering: processor i holds the columns at line positions (2i, 2i + 1); after every rotation the two
// columns swap positions, and the window of every processor slides by one position back and forth, so
// only ONE column per processor crosses LDS per round (half the traffic of the LDS-resident kernels, no
// address arithmetic or predication in the loop).  n rounds make every pair of columns meet exactly once.
// Even round: rotate (P, Q), send Q to processor i - 1, receive Q from i + 1.  Odd round: rotate (P, Q),
// send P to i + 1, receive P from i - 1; the last processor then holds (position n - 1, wrapped position 0),
// which are not a pair of the line ordering: it applies (cs, sn) = (0, 1), i.e. P <- -Q, Q <- P, a swap that
// flips the sign of one singular-vector pair.  V lags W by half a round so its update overlaps the exchange.
__device__ __forceinline__ float ring_sum8(float x) { return group_sum<8>(x); }
__device__ __forceinline__ double ring_sum8(double x) {
  auto dpp = [](double v, auto ctrl) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, decltype(ctrl)::value, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, decltype(ctrl)::value, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
  };
  x += dpp(x, std::integral_constant<int, 0xB1>{});
  x += dpp(x, std::integral_constant<int, 0x4E>{});
  x += dpp(x, std::integral_constant<int, 0x141>{});
  return x;
}
template <int G>
__device__ __forceinline__ float ring_sum(float x) {
  static_assert(G == 4 || G == 8, "lanes per processor");
  auto dpp = [](float v, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  x += dpp(x, std::integral_constant<int, 0xB1>{});  // quad_perm [1,0,3,2]
  x += dpp(x, std::integral_constant<int, 0x4E>{});  // quad_perm [2,3,0,1]
  if constexpr (G == 8) x += dpp(x, std::integral_constant<int, 0x141>{});  // row_half_mirror
  return x;
}
template <int G>
__device__ __forceinline__ double ring_sum(double x) {
  static_assert(G == 4 || G == 8, "lanes per processor");
  auto dpp = [](double v, auto ctrl) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, decltype(ctrl)::value, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, decltype(ctrl)::value, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
  };
  x += dpp(x, std::integral_constant<int, 0xB1>{});
  x += dpp(x, std::integral_constant<int, 0x4E>{});
  if constexpr (G == 8) x += dpp(x, std::integral_constant<int, 0x141>{});
  return x;
}
constexpr int kRingMaxThreads = 576;  // 72 processors x 8 lanes
template <class T, int E>
__global__ __launch_bounds__(kRingMaxThreads) void jacobi_ring_kernel(const T* __restrict__ c, int64_t ldc, int l, T* m1,
                                                                      int64_t ld1, T* m2, int64_t ld2, T* s_out, int k,
                                                                      T tol, T tol_early, int max_sweeps, int* info) {
  typedef typename MT<T>::vec_t vec_t;
  constexpr int VW = MT<T>::VEC;
  static_assert(E % VW == 0, "whole 16-byte chunks per lane");
  constexpr int NC = E / VW;
  constexpr int RS = 8 * E;  // padded column length
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n2 = (l + 1) & ~1, np = n2 >> 1;
  T* xw = (T*)smem;                  // [np][RS] W column in flight
  T* xv = xw + (size_t)np * RS;      // [np][RS] V column in flight
  T* sigma = xv + (size_t)np * RS;   // [n2]
  T* xn = sigma + n2;                // [np] squared norm of the W column in flight
  int* rank = (int*)(xn + np);       // [n2]
  int* flag = rank + n2;             // [4]
  const int tid = threadIdx.x, proc = tid >> 3, g = tid & 7;
  const bool act = proc < np;
  const bool last = proc == np - 1;
  vec_t pw[NC], qw[NC], pv[NC], qv[NC];
  {
    const int colp = 2 * proc, colq = 2 * proc + 1;
#pragma unroll
    for (int cc = 0; cc < NC; ++cc)
#pragma unroll
      for (int z = 0; z < VW; ++z) {
        const int row = (cc * 8 + g) * VW + z;
        const bool ok = act && row < l;
        pw[cc][z] = (ok && colp < l) ? c[(int64_t)colp * ldc + row] : (T)0;
        qw[cc][z] = (ok && colq < l) ? c[(int64_t)colq * ldc + row] : (T)0;
        pv[cc][z] = (ok && row == colp) ? (T)1 : (T)0;
        qv[cc][z] = (ok && row == colq) ? (T)1 : (T)0;
      }
  }
  const int my_off = proc * RS + g * VW;
  const int up_off = (proc + 1 >= np ? 0 : proc + 1) * RS + g * VW;
  const int dn_off = (proc == 0 ? np - 1 : proc - 1) * RS + g * VW;
  auto store = [&](T* buf, const vec_t (&x)[NC]) {
    if (act) {
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) *(vec_t*)(buf + my_off + cc * 8 * VW) = x[cc];
    }
  };
  auto load = [&](const T* buf, int off, vec_t (&x)[NC]) {
    if (act) {
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) x[cc] = *(const vec_t*)(buf + off + cc * 8 * VW);
    }
  };
  // squared norms of the two resident W columns: recomputed from the registers at the start of every sweep,
  // updated analytically by each rotation in between (a' = a - t g, b' = b + t g), and travelling with their
  // column through xn, so a round needs ONE dot product instead of three
  T na = (T)0, nb = (T)0;
  auto recompute_norms = [&]() {
    vec_t va, vb;
#pragma unroll
    for (int z = 0; z < VW; ++z) va[z] = vb[z] = (T)0;
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
      va += pw[cc] * pw[cc];
      vb += qw[cc] * qw[cc];
    }
    T a = va[0], b = vb[0];
#pragma unroll
    for (int z = 1; z < VW; ++z) {
      a += va[z];
      b += vb[z];
    }
    na = ring_sum8(a);
    nb = ring_sum8(b);
  };
  // rotation of the W pair in registers; forced: the wrap-around pseudo pair of the last processor
  auto rotate_w = [&](bool forced, T& cs, T& sn) {
    vec_t vg;
#pragma unroll
    for (int z = 0; z < VW; ++z) vg[z] = (T)0;
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) vg += pw[cc] * qw[cc];
    T gg = vg[0];
#pragma unroll
    for (int z = 1; z < VW; ++z) gg += vg[z];
    gg = ring_sum8(gg);
    T rel = (T)0, t = (T)0;
    cs = (T)1;
    sn = (T)0;
    const bool rot = !forced && jacobi_rotation(na, nb, gg, tol, cs, sn, rel, t);
    if (rot) {
      if (g == 0) {
        flag[0] = 1;
        if (rel > tol_early) flag[1] = 1;
      }
      na -= t * gg;
      nb += t * gg;
    } else {
      cs = forced ? (T)0 : (T)1;
      sn = forced ? (T)1 : (T)0;
      if (forced) {
        const T tmp = na;
        na = nb;
        nb = tmp;
      }
    }
    if (sn != (T)0) {
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) {
        const vec_t x = pw[cc], y = qw[cc];
        pw[cc] = cs * x - sn * y;
        qw[cc] = sn * x + cs * y;
      }
    }
  };
  auto apply_v = [&](T cs, T sn) {
    if (sn != (T)0) {
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) {
        const vec_t x = pv[cc], y = qv[cc];
        pv[cc] = cs * x - sn * y;
        qv[cc] = sn * x + cs * y;
      }
    }
  };
  const int up_proc = proc + 1 >= np ? 0 : proc + 1, dn_proc = proc == 0 ? np - 1 : proc - 1;
  int sweep = 0;
  for (; sweep < max_sweeps; ++sweep) {
    if (tid < 2) flag[tid] = 0;
    recompute_norms();
    __syncthreads();
    for (int r2 = 0; r2 < np; ++r2) {
      T cs, sn;
      // even round
      rotate_w(false, cs, sn);
      store(xw, qw);
      if (act && g == 0) xn[proc] = nb;
      __syncthreads();
      apply_v(cs, sn);
      store(xv, qv);
      load(xw, up_off, qw);
      if (act) nb = xn[up_proc];
      __syncthreads();
      load(xv, up_off, qv);
      // odd round
      rotate_w(last, cs, sn);
      store(xw, pw);
      if (act && g == 0) xn[proc] = na;
      __syncthreads();
      apply_v(cs, sn);
      store(xv, pv);
      load(xw, dn_off, pw);
      if (act) na = xn[dn_proc];
      __syncthreads();
      load(xv, dn_off, pv);
    }
    __syncthreads();
    const int rotated = flag[0], big = flag[1];
    __syncthreads();
    if (!rotated || !big) {
      if (rotated) ++sweep;
      break;
    }
  }
  // singular values; the zero padding column of an odd l (its V column is zero too) sorts last
  {
    T a = 0, b = 0, va = 0, vb = 0;
#pragma unroll
    for (int cc = 0; cc < NC; ++cc)
#pragma unroll
      for (int z = 0; z < VW; ++z) {
        a += pw[cc][z] * pw[cc][z];
        b += qw[cc][z] * qw[cc][z];
        va += pv[cc][z] * pv[cc][z];
        vb += qv[cc][z] * qv[cc][z];
      }
    a = ring_sum8(a);
    b = ring_sum8(b);
    va = ring_sum8(va);
    vb = ring_sum8(vb);
    if (act && g == 0) {
      sigma[2 * proc] = va > (T)0 ? jacobi_safe_sigma(a) : (T)-1;
      sigma[2 * proc + 1] = vb > (T)0 ? jacobi_safe_sigma(b) : (T)-1;
    }
  }
  __syncthreads();
  for (int j = tid; j < n2; j += blockDim.x) {
    const T sj = sigma[j];
    int r = 0;
    for (int i = 0; i < n2; ++i) {
      const T si = sigma[i];
      r += (si > sj || (si == sj && i < j)) ? 1 : 0;
    }
    rank[j] = r;
  }
  __syncthreads();
  if (act) {
    const int rp = rank[2 * proc], rq = rank[2 * proc + 1];
    const T sp = sigma[2 * proc], sq = sigma[2 * proc + 1];
    const T ip = sp > (T)0 ? (T)1 / sp : (T)0, iq = sq > (T)0 ? (T)1 / sq : (T)0;
#pragma unroll
    for (int cc = 0; cc < NC; ++cc)
#pragma unroll
      for (int z = 0; z < VW; ++z) {
        const int row = (cc * 8 + g) * VW + z;
        if (row < l) {
          if (rp < k) {
            m2[(int64_t)rp * ld2 + row] = pw[cc][z] * ip;
            m1[(int64_t)rp * ld1 + row] = pv[cc][z];
          }
          if (rq < k) {
            m2[(int64_t)rq * ld2 + row] = qw[cc][z] * iq;
            m1[(int64_t)rq * ld1 + row] = qv[cc][z];
          }
        }
      }
    if (g == 0) {
      if (rp < k) s_out[rp] = sp > (T)0 ? sp : (T)0;
      if (rq < k) s_out[rq] = sq > (T)0 ? sq : (T)0;
    }
  }
  if (tid == 0) info[0] = sweep;
}
// ---- ring Jacobi, W only + replay of the rotation stream onto V ------------------------------------
// The ring kernel above is VALU-issue bound and ~45 % of its instructions accumulate V.  The rows of V are
// independent and need nothing but the (cs, sn) of every round, so: jacobi_ring_w_kernel keeps only W in
// registers, records every round's rotations in a global stream (rot[round][processor] = (cs, sn), identity for
// skipped pairs) and -- with the V buffer gone from LDS -- double-buffers the W exchange: ONE barrier per
// round.  jacobi_replay_v_kernel then applies the recorded stream to V = I with 8 lanes per ROW of V (18
// positions per lane in registers), 32 rows per workgroup, on as many CUs as there are row groups.  Line
// positions after S sweeps are known in closed form (always-swap odd-even transposition reverses the order
// every sweep), so the W kernel only has to publish rank[position] for the replay to scatter V_c[:, :k].
constexpr int kRingProcPad = 72;  // processors per stream row (padded)
template <class T>
struct RotEntry {
  T cs, sn;
};
template <class T, int E, int G>
__global__ __launch_bounds__(kRingProcPad * G) void jacobi_ring_w_kernel(const T* __restrict__ c, int64_t ldc, int l, T* m2,
                                                                        int64_t ld2, T* s_out, int k, T tol, T tol_early,
                                                                        int max_sweeps, RotEntry<T>* rot, int* rank_g,
                                                                        int* info) {
  typedef typename MT<T>::vec_t vec_t;
  constexpr int VW = MT<T>::VEC;
  static_assert(E % VW == 0, "whole 16-byte chunks per lane");
  constexpr int NC = E / VW;
  constexpr int RS = G * E;  // padded column length
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n2 = (l + 1) & ~1, np = n2 >> 1;
  const int nproc = blockDim.x / G;      // processors incl. the idle ones of the last wave: all own LDS slots, so
                                         // the round body needs no "active" guards
  T* xw = (T*)smem;                      // [2][nproc][RS] W column in flight (double-buffered)
  T* sigma = xw + (size_t)2 * nproc * RS;  // [n2]
  T* xn = sigma + n2;                    // [2][nproc] squared norm of the column in flight
  int* rank = (int*)(xn + 2 * nproc);    // [n2]
  int* flag = rank + n2;                 // [4]
  const int tid = threadIdx.x, proc = tid / G, g = tid % G;
  const bool act = proc < np;
  const bool last = proc == np - 1;
  vec_t pw[NC], qw[NC];
  {
    const int colp = 2 * proc, colq = 2 * proc + 1;
#pragma unroll
    for (int cc = 0; cc < NC; ++cc)
#pragma unroll
      for (int z = 0; z < VW; ++z) {
        const int row = (cc * G + g) * VW + z;
        const bool ok = act && row < l;
        pw[cc][z] = (ok && colp < l) ? c[(int64_t)colp * ldc + row] : (T)0;
        qw[cc][z] = (ok && colq < l) ? c[(int64_t)colq * ldc + row] : (T)0;
      }
  }
  // Scale invariance: the core is multiplied by an exact power of two that brings its largest entry into [1, 2)
  // (squared column norms of a tiny or huge core would under- / overflow); sigma is scaled back on output.
  int sexp = 0;
  {
    T mx = (T)0;
#pragma unroll
    for (int cc = 0; cc < NC; ++cc)
#pragma unroll
      for (int z = 0; z < VW; ++z) mx = fmax(mx, fmax(fabs(pw[cc][z]), fabs(qw[cc][z])));
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
    if ((tid & 63) == 0) sigma[tid >> 6] = mx;   // sigma is scratch until the end
    __syncthreads();
    mx = (T)0;
    for (int wv = 0; wv < (int)((blockDim.x + 63) >> 6); ++wv) mx = fmax(mx, sigma[wv]);
    __syncthreads();
    if (mx > (T)0 && mx <= std::numeric_limits<T>::max()) {  // finite, non-zero
      (void)frexp((double)mx, &sexp);
      sexp = 1 - sexp;  // mx * 2^sexp in [1, 2)
#pragma unroll
      for (int cc = 0; cc < NC; ++cc)
#pragma unroll
        for (int z = 0; z < VW; ++z) {
          pw[cc][z] = (T)ldexp((double)pw[cc][z], sexp);
          qw[cc][z] = (T)ldexp((double)qw[cc][z], sexp);
        }
    }
  }
  const int my_off = proc * RS + g * VW;
  // idle processors (proc >= np) form their own harmless ring of zero columns
  const int up_proc = act ? (proc + 1 >= np ? 0 : proc + 1) : proc, dn_proc = act ? (proc == 0 ? np - 1 : proc - 1) : proc;
  const int up_off = up_proc * RS + g * VW, dn_off = dn_proc * RS + g * VW;
  T na = (T)0, nb = (T)0;
  auto recompute_norms = [&]() {
    vec_t va, vb;
#pragma unroll
    for (int z = 0; z < VW; ++z) va[z] = vb[z] = (T)0;
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
      va += pw[cc] * pw[cc];
      vb += qw[cc] * qw[cc];
    }
    T a = va[0], b = vb[0];
#pragma unroll
    for (int z = 1; z < VW; ++z) {
      a += va[z];
      b += vb[z];
    }
    na = ring_sum<G>(a);
    nb = ring_sum<G>(b);
  };
  // one round: rotate (P, Q), record the rotation, send `snd` (with its norm) and receive it from `src`
  auto round = [&](bool forced, RotEntry<T>* rot_row, T* xwb, T* xnb, vec_t (&snd)[NC], T& nsnd, int src_off, int src_proc) {
    vec_t vg;
#pragma unroll
    for (int z = 0; z < VW; ++z) vg[z] = (T)0;
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) vg += pw[cc] * qw[cc];
    T gg = vg[0];
#pragma unroll
    for (int z = 1; z < VW; ++z) gg += vg[z];
    gg = ring_sum<G>(gg);
    T rel = (T)0, t = (T)0, cs = (T)1, sn = (T)0;
    const bool rot_now = !forced && jacobi_rotation(na, nb, gg, tol, cs, sn, rel, t);
    if (rot_now) {
      if (g == 0) {
        flag[0] = 1;
        if (rel > tol_early) flag[1] = 1;
      }
      na -= t * gg;
      nb += t * gg;
    } else {
      cs = (T)1;
      sn = (T)0;
    }
    if (g == 0) rot_row[proc] = RotEntry<T>{cs, sn};  // the pseudo pair is recorded as the identity
    if (forced) {  // P <- -Q, Q <- P
      cs = (T)0;
      sn = (T)1;
      const T tmp = na;
      na = nb;
      nb = tmp;
    }
    // unconditional: (cs, sn) = (1, 0) is exact, and a branch would only pay off when all 8 processors of a wave skip
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
      const vec_t x = pw[cc], y = qw[cc];
      pw[cc] = cs * x - sn * y;
      qw[cc] = sn * x + cs * y;
    }
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) *(vec_t*)(xwb + my_off + cc * G * VW) = snd[cc];
    if (g == 0) xnb[proc] = nsnd;
    // LDS-only barrier: the rotation-stream store above must stay in flight (__syncthreads waits on vmcnt too)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) snd[cc] = *(const vec_t*)(xwb + src_off + cc * G * VW);
    nsnd = xnb[src_proc];
  };
  int sweep = 0;
  int rounds = 0;
  for (; sweep < max_sweeps; ++sweep) {
    if (tid < 2) flag[tid] = 0;
    recompute_norms();
    __syncthreads();
    for (int r2 = 0; r2 < np; ++r2, rounds += 2) {
      // even round: send Q to processor i - 1, receive Q from i + 1
      round(false, rot + (size_t)rounds * kRingProcPad, xw, xn, qw, nb, up_off, up_proc);
      // odd round: send P to i + 1, receive P from i - 1 (other buffer: one barrier per round is enough)
      round(last, rot + (size_t)(rounds + 1) * kRingProcPad, xw + (size_t)nproc * RS, xn + nproc, pw, na, dn_off, dn_proc);
    }
    __syncthreads();
    const int rotated = flag[0], big = flag[1];
    __syncthreads();
    if (!rotated || !big) {
      ++sweep;  // this sweep's rounds are in the stream whether or not it rotated anything
      break;
    }
  }
  // Column ids by position: every completed sweep reverses the line order.  The zero padding column of an odd l
  // (id l) must sort last even against exact-zero singular values.
  {
    T a = 0, b = 0;
#pragma unroll
    for (int cc = 0; cc < NC; ++cc)
#pragma unroll
      for (int z = 0; z < VW; ++z) {
        a += pw[cc][z] * pw[cc][z];
        b += qw[cc][z] * qw[cc][z];
      }
    a = ring_sum<G>(a);
    b = ring_sum<G>(b);
    const int nsw = rounds / n2;
    const int idp = (nsw & 1) ? n2 - 1 - 2 * proc : 2 * proc;
    const int idq = (nsw & 1) ? n2 - 2 - 2 * proc : 2 * proc + 1;
    if (act && g == 0) {
      sigma[2 * proc] = idp < l ? jacobi_safe_sigma(a) : (T)-1;
      sigma[2 * proc + 1] = idq < l ? jacobi_safe_sigma(b) : (T)-1;
    }
  }
  __syncthreads();
  for (int j = tid; j < n2; j += blockDim.x) {
    const T sj = sigma[j];
    int r = 0;
    for (int i = 0; i < n2; ++i) {
      const T si = sigma[i];
      r += (si > sj || (si == sj && i < j)) ? 1 : 0;
    }
    rank[j] = r;
    rank_g[j] = r;
  }
  __syncthreads();
  if (act) {
    const int rp = rank[2 * proc], rq = rank[2 * proc + 1];
    const T sp = sigma[2 * proc], sq = sigma[2 * proc + 1];
    const T ip = sp > (T)0 ? (T)1 / sp : (T)0, iq = sq > (T)0 ? (T)1 / sq : (T)0;
#pragma unroll
    for (int cc = 0; cc < NC; ++cc)
#pragma unroll
      for (int z = 0; z < VW; ++z) {
        const int row = (cc * G + g) * VW + z;
        if (row < l) {
          if (rp < k) m2[(int64_t)rp * ld2 + row] = pw[cc][z] * ip;
          if (rq < k) m2[(int64_t)rq * ld2 + row] = qw[cc][z] * iq;
        }
      }
    if (g == 0) {
      if (rp < k) s_out[rp] = sp > (T)0 ? (T)ldexp((double)sp, -sexp) : (T)0;
      if (rq < k) s_out[rq] = sq > (T)0 ? (T)ldexp((double)sq, -sexp) : (T)0;
    }
  }
  if (tid == 0) {
    info[0] = sweep;
    info[1] = rounds;
  }
}
// rs = G * E: rows per column slot
__host__ __device__ inline size_t jacobi_ring_w_lds_bytes(int l, int rs, size_t esz) {
  const int n2 = (l + 1) & ~1;
  const int nproc = n2 / 2;  // launched with exactly G * np threads
  return (size_t)2 * nproc * rs * esz + (size_t)n2 * (esz + sizeof(int)) + (size_t)2 * nproc * esz + 64;
}

// V_c[:, :k] from the recorded rotation stream: 8 lanes per row of V, 18 line positions per lane, 32 rows per
// workgroup.  Round semantics (identical to the ring kernels): pair (first, second) = positions (2i, 2i + 1) in
// even rounds, (2i + 1, 2i + 2) in odd rounds; first <- sn x + cs y, second <- cs x - sn y (rotation, then the two
// columns swap positions); in odd rounds positions n - 1 and 0 are idle and position 0 changes sign (the pseudo
// pair of the last processor).
constexpr int kReplayChunk = 16;  // rounds staged in LDS at a time
constexpr int kReplayLanes = 16;  // lanes per row of V (one DPP row): 10 line positions per lane (32 lanes: 6)
// value of the previous / next lane inside the 16-lane DPP row (the 8-lane groups are row-aligned)
__device__ __forceinline__ float dpp_from_prev(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_from_next(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x101, 0xf, 0xf, true));
}
__device__ __forceinline__ double dpp_from_prev(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x111, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x111, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_from_next(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x101, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x101, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <class T>
__global__ __launch_bounds__(256) void jacobi_replay_v_kernel(const RotEntry<T>* __restrict__ rot, const int* __restrict__ info,
                                                              const int* __restrict__ rank_g, int l, int k, T* m1,
                                                              int64_t ld1) {
  constexpr int GL = kReplayLanes, NPL = GL == 16 ? 10 : 6, NPR = NPL / 2;  // GL * NPL >= 144 positions
  static_assert(GL * NPL >= 2 * kRingProcPad, "positions");
  constexpr int ROWS = 256 / GL;                             // rows of V per workgroup
  constexpr int kReplayRow = GL * (NPR + 1);                 // staged entries per round (one pad entry per lane)
  typedef typename MT<T>::vec_t vec_t;
  constexpr int VW = MT<T>::VEC;
  constexpr int EV = (2 * (NPR + 1)) / VW;  // 16-byte vectors holding one lane's staged entries
  static_assert((2 * (NPR + 1)) % VW == 0, "lane entries must fill whole 16-byte vectors");
  __shared__ __attribute__((aligned(16))) RotEntry<T> stage[2][kReplayChunk][kReplayRow];
  const int n2 = (l + 1) & ~1, np = n2 >> 1;
  const int nrounds = info[1];
  const int tid = threadIdx.x, ln = tid % GL, row = blockIdx.x * ROWS + tid / GL;
  T v[NPL];
#pragma unroll
  for (int j = 0; j < NPL; ++j) v[j] = (ln * NPL + j == row && row < l) ? (T)1 : (T)0;
  const int nchunks = (nrounds + kReplayChunk - 1) / kReplayChunk;
  constexpr int PER_THREAD = (kReplayChunk * kRingProcPad + 255) / 256;
  RotEntry<T> pre[PER_THREAD];
  auto fetch = [&](int chunk) {
#pragma unroll
    for (int q = 0; q < PER_THREAD; ++q) {
      const int idx = tid + 256 * q;
      const int64_t gidx = (int64_t)chunk * kReplayChunk * kRingProcPad + idx;
      const bool ok = idx < kReplayChunk * kRingProcPad && gidx < (int64_t)nrounds * kRingProcPad;
      pre[q] = ok ? rot[gidx] : RotEntry<T>{(T)1, (T)0};
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int q = 0; q < PER_THREAD; ++q) {
      const int idx = tid + 256 * q;
      if (idx < kReplayChunk * kRingProcPad) {
        const int rr = idx / kRingProcPad, i = idx - rr * kRingProcPad;
        stage[buf][rr][(i / NPR) * (NPR + 1) + i % NPR] = pre[q];
      }
    }
  };
  // validity of this lane's pairs (even rounds: i < np; odd rounds: i < np - 1)
  const int i0 = ln * NPR;
  if (nchunks > 0) {
    fetch(0);
    stash(0);
  }
  __syncthreads();
  for (int ch = 0; ch < nchunks; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nchunks) fetch(ch + 1);
    const int r_hi = min(kReplayChunk, nrounds - ch * kReplayChunk);
    // Rounds come in (even, odd) pairs (chunks start on an even round); the entries of the next round are
    // loaded into the other register set while this one is applied, so no LDS latency sits between rounds.
    vec_t ea[EV], eb[EV];
    T acs = (T)1, asn = (T)0, bcs = (T)1, bsn = (T)0;  // boundary pair of the previous lane
    auto load_round = [&](int rr, vec_t (&dst)[EV], T& pc, T& ps) {
      const vec_t* src = (const vec_t*)&stage[buf][rr][ln * (NPR + 1)];
#pragma unroll
      for (int q = 0; q < EV; ++q) dst[q] = src[q];
      const RotEntry<T> pb = stage[buf][rr][ln > 0 ? (ln - 1) * (NPR + 1) + NPR - 1 : 0];
      pc = pb.cs;
      ps = pb.sn;
    };
    auto even_round = [&](const vec_t (&e)[EV]) {
#pragma unroll
      for (int j = 0; j < NPR; ++j) {
        const T cs = e[(2 * j) / VW][(2 * j) % VW], sn = e[(2 * j + 1) / VW][(2 * j + 1) % VW];
        const bool ok = i0 + j < np;
        const T x = v[2 * j], y = v[2 * j + 1];
        const T nx = sn * x + cs * y, ny = cs * x - sn * y;
        v[2 * j] = ok ? nx : x;
        v[2 * j + 1] = ok ? ny : y;
      }
    };
    auto odd_round = [&](const vec_t (&e)[EV], T pc, T ps) {
      const T my_last = v[NPL - 1], my_first = v[0];
      // neighbours' boundary values: DPP inside a 16-lane row, the LDS crossbar for 32-lane groups (issued
      // first, consumed after the in-lane pairs)
      const T y_next = GL <= 16 ? dpp_from_next(my_first) : __shfl_down(my_first, 1, GL);
      const T x_prev = GL <= 16 ? dpp_from_prev(my_last) : __shfl_up(my_last, 1, GL);
#pragma unroll
      for (int j = 0; j < NPR - 1; ++j) {
        const T cs = e[(2 * j) / VW][(2 * j) % VW], sn = e[(2 * j + 1) / VW][(2 * j + 1) % VW];
        const bool ok = i0 + j < np - 1;
        const T x = v[2 * j + 1], y = v[2 * j + 2];
        const T nx = sn * x + cs * y, ny = cs * x - sn * y;
        v[2 * j + 1] = ok ? nx : x;
        v[2 * j + 2] = ok ? ny : y;
      }
      const T lcs = e[(2 * (NPR - 1)) / VW][(2 * (NPR - 1)) % VW], lsn = e[(2 * NPR - 1) / VW][(2 * NPR - 1) % VW];
      const T nl = lsn * my_last + lcs * y_next;
      v[NPL - 1] = (ln < GL - 1 && i0 + NPR - 1 < np - 1) ? nl : my_last;
      const T nf = pc * x_prev - ps * my_first;
      v[0] = ln == 0 ? -my_first : ((i0 - 1 < np - 1) ? nf : my_first);
    };
    if (r_hi > 0) load_round(0, ea, acs, asn);
    for (int rr = 0; rr < r_hi; rr += 2) {
      if (rr + 1 < r_hi) load_round(rr + 1, eb, bcs, bsn);
      even_round(ea);
      if (rr + 1 < r_hi) {
        if (rr + 2 < r_hi) load_round(rr + 2, ea, acs, asn);
        odd_round(eb, bcs, bsn);
      }
    }
    if (ch + 1 < nchunks) stash(buf ^ 1);
    __syncthreads();
  }
  if (row < l) {
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int pos = ln * NPL + j;
      if (pos < n2) {
        const int r = rank_g[pos];
        if (r < k) m1[(int64_t)r * ld1 + row] = v[j];
      }
    }
  }
}

__host__ __device__ inline size_t jacobi_ring_lds_bytes(int l, int e, size_t esz) {
  const int n2 = (l + 1) & ~1;
  return (size_t)2 * (n2 / 2) * 8 * e * esz + (size_t)n2 * (esz + sizeof(int)) + (size_t)(n2 / 2) * esz + 64;
}

constexpr int kJacobiMaxL = 256;  // 8-lane groups x 8 chunks x 4 elements

// ---- block Jacobi SVD of the l x l core for any l (random_svd.rs:89) -----------------------------
// W (= C on entry) and V (= I) live in global memory (L2-resident, a few hundred KiB).  The columns are
// cut into nb blocks of 8; a round-robin tournament over the blocks gives nb/2 disjoint block pairs per
// round, one kernel launch per round, ONE WAVE per block pair:
//   1. stage the 16 columns of W and V into LDS,
//   2. G = W_S^T W_S (16 x 16) with MFMAs (the same LDS element feeds both operands),
//   3. two-sided Jacobi on G in LDS (tournament of 8 disjoint rotations per step, all 64 lanes apply
//      them) accumulating the 16 x 16 rotation J -- in exact arithmetic this is one-sided Jacobi on the
//      16 columns without touching the long columns,
//   4. W_S <- W_S J and V_S <- V_S J with MFMAs, stored straight from the accumulators.
// Every element of W and V is read and written once per ROUND (2*nb/8 times fewer than per column-pair
// step).  Convergence: each wave publishes max |g_pq| / sqrt(g_pp g_qq) of its block pair (before
// rotating) with atomicMax; a one-thread kernel per sweep raises `done` when the whole sweep stayed
// below tol_early (quadratic convergence), and every later launch returns immediately.
struct JacobiCtl {
  unsigned done;
  unsigned max_bits;  // float bits of the sweep's max relative off-diagonal (non-negative floats order as uints)
  unsigned sweeps;
  unsigned pad;
};

template <class T>
__global__ __launch_bounds__(256) void jacobi_block_round_kernel(T* w, int64_t ldw, T* v, int64_t ldv, int rows_pad, int nb,
                                                                 int round, int inner_sweeps, JacobiCtl* ctl) {
  typedef typename MT<T>::acc_t acc_t;
  if (ctl->done) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int LP = rows_pad + 1;
  T* wt = (T*)smem;          // [16][LP]
  T* vt = wt + 16 * LP;      // [16][LP]
  T* gm = vt + 16 * LP;      // [16][17]
  T* g2 = gm + 16 * 17;      // [16][17]
  T* jm = g2 + 16 * 17;      // [16][17]
  T* gpart = jm + 16 * 17;   // [4][16][17] partial Grams of the 4 waves
  T* alpha = gpart + 4 * 16 * 17;  // [16]
  T* beta = alpha + 16;            // [16]
  int* part = (int*)(beta + 16);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int x = lane & 15, kq = lane >> 4;
  // block pair of this workgroup
  int bi, bj;
  tournament_pair(nb, round, blockIdx.x, bi, bj);
  auto gcol = [&](int c) { return (c < 8) ? bi * 8 + c : bj * 8 + (c - 8); };
  // 1. stage the 16 columns of W and V (all 4 waves)
  for (int e = tid; e < 16 * rows_pad; e += 256) {
    const int c = e / rows_pad, r = e - c * rows_pad;
    const int64_t gc = gcol(c);
    wt[c * LP + r] = w[gc * ldw + r];
    vt[c * LP + r] = v[gc * ldv + r];
  }
  __syncthreads();
  // 2. Gram: wave `wave` sums the 16-row groups it = wave, wave + 4, ... (independent MFMA chains per wave)
  const int nit = rows_pad / 16;
  {
    acc_t acc = (acc_t){0, 0, 0, 0};
    for (int it = wave; it < nit; it += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const T a = wt[x * LP + 16 * it + 4 * kq + j];
        acc = MT<T>::mma(a, a, acc);
      }
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) gpart[(wave * 16 + MT<T>::drow(lane, jj)) * 17 + x] = acc[jj];
  }
  __syncthreads();
  {
    const int r = tid >> 4, cx = tid & 15;  // 256 threads = 16 x 16 entries
    const T gsum = gpart[(0 * 16 + r) * 17 + cx] + gpart[(1 * 16 + r) * 17 + cx] + gpart[(2 * 16 + r) * 17 + cx] +
                   gpart[(3 * 16 + r) * 17 + cx];
    gm[r * 17 + cx] = gsum;
    jm[r * 17 + cx] = (r == cx) ? (T)1 : (T)0;
  }
  __syncthreads();
  // 3. inner two-sided Jacobi on the 16 x 16 Gram, wave 0 only (the others wait at the barrier below)
  if (wave == 0) {
    {  // convergence measure of this block pair (before rotating)
      float mx = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int idx = lane + 64 * e, r = idx >> 4, cx = idx & 15;
        if (r != cx) {
          const T d = gm[r * 17 + r] * gm[cx * 17 + cx];
          if (d > (T)0) mx = fmaxf(mx, (float)(fabs(gm[r * 17 + cx]) / sqrt(d)));
        }
      }
      for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_down(mx, off, 64));
      if (lane == 0) atomicMax(&ctl->max_bits, __float_as_uint(mx));
    }
    const T tiny = (T)4 * (T)(sizeof(T) == 4 ? 1.1920929e-07 : 2.220446049250313e-16);
    for (int sw = 0; sw < inner_sweeps; ++sw) {
      for (int step = 0; step < 15; ++step) {
        if (lane < 8) {
          int p, q;
          tournament_pair(16, step, lane, p, q);
          T cs = (T)1, sn = (T)0, rel;
          if (!jacobi_rotation(gm[p * 17 + p], gm[q * 17 + q], gm[p * 17 + q], tiny, cs, sn, rel)) {
            cs = (T)1;
            sn = (T)0;
          }
          alpha[p] = cs;
          beta[p] = -sn;
          part[p] = q;
          alpha[q] = cs;
          beta[q] = sn;
          part[q] = p;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // G <- R^T G R in one pass through g2, J <- J R
        T gn[4], jn[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int idx = lane + 64 * e, r = idx >> 4, cx = idx & 15, pc = part[cx], pr_ = part[r];
          const T ar = alpha[r], br = beta[r], ac = alpha[cx], bc = beta[cx];
          gn[e] = ar * (ac * gm[r * 17 + cx] + bc * gm[r * 17 + pc]) + br * (ac * gm[pr_ * 17 + cx] + bc * gm[pr_ * 17 + pc]);
          jn[e] = ac * jm[r * 17 + cx] + bc * jm[r * 17 + pc];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int idx = lane + 64 * e, r = idx >> 4, cx = idx & 15;
          gm[r * 17 + cx] = gn[e];
          jm[r * 17 + cx] = jn[e];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  __syncthreads();
  // 4. apply J to the W and V column blocks: wave `wave` owns the row groups it = wave, wave + 4, ...
  T jb[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) jb[ks] = jm[(4 * ks + kq) * 17 + x];
  const int64_t gx = gcol(x);
  for (int it = wave; it < nit; it += 4) {
    acc_t aw = (acc_t){0, 0, 0, 0}, av = (acc_t){0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      aw = MT<T>::mma(wt[(4 * ks + kq) * LP + 16 * it + x], jb[ks], aw);
      av = MT<T>::mma(vt[(4 * ks + kq) * LP + 16 * it + x], jb[ks], av);
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int r = 16 * it + MT<T>::drow(lane, jj);
      w[gx * ldw + r] = aw[jj];
      v[gx * ldv + r] = av[jj];
    }
  }
}

__global__ void jacobi_sweep_end_kernel(JacobiCtl* ctl, float tol_early) {
  if (ctl->done) return;
  ctl->sweeps += 1;
  if (__uint_as_float(ctl->max_bits) <= tol_early) ctl->done = 1;
  ctl->max_bits = 0;
}

template <class T>
__global__ void jacobi_init_kernel(const T* c, int64_t ldc, int l, T* w, int64_t ldw, T* v, int64_t ldv, int cols_pad,
                                   int rows_pad, JacobiCtl* ctl) {
  const int64_t total = (int64_t)cols_pad * rows_pad;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(e / rows_pad), i = (int)(e - (int64_t)j * rows_pad);
    const bool in = i < l && j < l;
    w[(int64_t)j * ldw + i] = in ? c[(int64_t)j * ldc + i] : (T)0;
    v[(int64_t)j * ldv + i] = (in && i == j) ? (T)1 : (T)0;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ctl->done = 0;
    ctl->max_bits = 0;
    ctl->sweeps = 0;
    ctl->pad = 0;
  }
}

// sigma_j = ||w_j||, descending order, outputs (see jacobi_svd_kernel)
template <class T>
__global__ __launch_bounds__(1024) void jacobi_finish_kernel(const T* w, int64_t ldw, const T* v, int64_t ldv, int l, T* m1,
                                                             int64_t ld1, T* m2, int64_t ld2, T* s_out, int k) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sigma = (T*)smem;
  int* order = (int*)(sigma + l + 2);
  const int tid = threadIdx.x, group = tid >> 4, gl = tid & 15;
  for (int j = group; j < l; j += 64) {
    T a = 0;
    for (int i = gl; i < l; i += 16) {
      const T xx = w[(int64_t)j * ldw + i];
      a += xx * xx;
    }
#pragma unroll
    for (int msk = 1; msk < 16; msk <<= 1) a += __shfl_xor(a, msk, 16);
    if (gl == 0) sigma[j] = jacobi_safe_sigma(a);  // NaN-safe: the ranking below must stay a permutation
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
      m2[(int64_t)r * ld2 + i] = w[(int64_t)j * ldw + i] * inv;
      m1[(int64_t)r * ld1 + i] = v[(int64_t)j * ldv + i];
    }
    if (gl == 0) s_out[r] = sj;
  }
}

// ---- device Cholesky + triangular inverse for the Cholesky-QR passes --------------------------------
// One workgroup: G (r x r, symmetric, column-major) -> M = R^-1 with G = R^T R (upper R), written into the
// zero-padded skinny operand of the following apply GEMM.  G and M live in LDS.  status: fail = 0 ok / 1 =
// pivot failure (M is then the identity: the apply becomes a no-op copy) / 2 = zero matrix / 3 = non-finite; [1] = min pivot
// ratio d_j / g_jj, [2] = max |G - I| (both as float bits).  Lets the two passes of a CholeskyQR2 be enqueued
// without any host round trip; the host inspects the status words once at the end.
__device__ inline float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ inline double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  return y * (2.0 - x * y);
}
__device__ inline float fast_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ inline double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  return y * (1.5 - 0.5 * x * y * y);
}
struct CholStatus {
  int fail;
  float min_ratio;
  float dev_i;
  float gmax;
  long long clk, wall;  // shader-clock and 100 MHz wall-clock ticks spent in the elimination loop
};
// Robust form (rq.need_next != nullptr), used by the device-side Cholesky-QR that never asks the host:
//  * the factorisation is that of G + shift_rel * max_i g_ii * I whenever the Gram is further than 0.25 from I (or
//    always: shift_mode 1; never: shift_mode 2).  A shifted factor stays non-singular however ill-conditioned the
//    sketch is, the product Y R_s^-1 keeps the span of Y and AMPLIFIES its weak directions by up to 1 / sqrt(shift_rel)
//    relative to the strong ones -- they come from Y itself, not from the (squared) Gram -- so a few shifted passes
//    unfold condition numbers up to 1 / eps like a Householder QR does (shifted CholeskyQR3, Fukaya et al. 2020);
//  * a pivot that fails the test all the same (an exactly zero column, or a shift below the Gram's rounding error)
//    is a NULL column instead of a failure: it is skipped, column j of R^-1 is zero (the applied product leaves a zero
//    column that refill_null_kernel replaces by a random one) and null_mask[j] = 1;
//  * need_next <- 1 when the product of this pass cannot be orthonormal to working precision yet (it was shifted, has
//    null columns, or its Gram was further than 0.25 from I), else 0: the kernels of the next pass carry it as their
//    run_if word;
//  * run_if: the whole launch does nothing when *run_if == 0.
// bits of a need_next word besides bit 0 ("run another pass"): what the pass saw, for the host's end-of-call verdict
constexpr int kNeedNonFinite = 2;  // a non-finite Gram matrix: the call ends with CORRLA_ENUMERIC, nothing to escalate
constexpr int kNeedNullCols = 4;   // null columns were found (and re-seeded at random): the sketch is rank deficient
struct CholRobust {
  float shift_rel;
  int shift_mode;  // 0: shift iff ||G - I||_max > 0.25, 1: always, 2: never
  float null_excess;  // > 0 (shifted passes only): a column whose pivot exceeds the shift by less than null_excess * shift
                      // -- it lies below the level one shifted pass can lift -- is a null column as well (in-loop
                      // re-orthonormalisations: such directions are re-seeded at random rather than kept as noise)
  int* need_next;
  int* null_mask;  // r words
  const int* run_if;
  float need_ratio;  // > 0 (the single pass of an in-loop thin-Q): need_next <- null columns or a pivot ratio d_j / g_jj below
                     // this, i.e. the sketch was too ill-conditioned for one shifted pass to tame; 0: the standard rule
  const void* abs_shift;  // optional (2 x 2 blocked factorisation): device scalar T = the shift gram_inspect_kernel has
                          // ALREADY added to the diagonal of the whole Gram; decides `shifted` and feeds the null test
};
// what gram_inspect_kernel leaves for the blocked factorisation of one pass
template <class T>
struct GramInspect {
  T shift;      // absolute shift added to the diagonal (0: none)
  int shifted;
  int bad;      // non-finite entries
  float d2;     // ||G - I||_max before the shift
  float gmax;
};
// One workgroup: inspects the l x l Gram (column-major, ld), decides the shift like the robust chol_inv_kernel does
// (mode 0: iff ||G - I||_max > 0.25, 1: always, 2: never), adds it to the diagonal and leaves the record.
template <class T>
__global__ __launch_bounds__(1024) void gram_inspect_kernel(T* g, int64_t ld, int l, float shift_rel, int shift_mode,
                                                            GramInspect<T>* out, const int* run_if) {
  if (run_if && *run_if == 0) return;
  __shared__ float rd[16], rg[16];
  __shared__ int sbad;
  const int tid = threadIdx.x;
  if (tid == 0) sbad = 0;
  float dv = 0.f, gm = 0.f;
  int bad = 0;
  for (int idx = tid; idx < l * l; idx += 1024) {
    const int j = idx / l, i = idx - j * l;
    const T x = g[(int64_t)j * ld + i];
    if (!((float)fabs(x) < 3.0e38f)) bad = 1;
    dv = fmaxf(dv, (float)fabs(x - (i == j ? (T)1 : (T)0)));
    if (i == j) gm = fmaxf(gm, (float)x);
  }
  for (int off = 32; off > 0; off >>= 1) {
    dv = fmaxf(dv, __shfl_down(dv, off, 64));
    gm = fmaxf(gm, __shfl_down(gm, off, 64));
  }
  __syncthreads();
  if ((tid & 63) == 0) {
    rd[tid >> 6] = dv;
    rg[tid >> 6] = gm;
  }
  if (bad) sbad = 1;
  __syncthreads();
  dv = 0.f;
  gm = 0.f;
  for (int i = 0; i < 16; ++i) {
    dv = fmaxf(dv, rd[i]);
    gm = fmaxf(gm, rg[i]);
  }
  const bool shifted = !sbad && gm > 0.f && shift_rel > 0.f && (shift_mode == 1 || (shift_mode == 0 && dv > 0.25f));
  const T sh = shifted ? (T)shift_rel * (T)gm : (T)0;
  if (shifted)
    for (int i = tid; i < l; i += 1024) g[(int64_t)i * ld + i] += sh;
  if (tid == 0) {
    out->shift = sh;
    out->shifted = shifted ? 1 : 0;
    out->bad = sbad;
    out->d2 = dv;
    out->gmax = gm;
  }
}
template <class T>
__global__ void combine_need_kernel(int* need, const GramInspect<T>* insp, const int* na, const int* nb, const int* run_if) {
  if (run_if && *run_if == 0) return;
  // insp == nullptr (the single pass of an in-loop thin-Q): the verdict is that of the two block factorisations alone
  // bit 0: another pass is needed; bit 1 (kNeedNonFinite): a non-finite Gram; bit 2 (kNeedNullCols): null columns were
  // re-seeded -- the host reads the words once at the end of the call (driver.hpp: pending_clean)
  const int sub = *na | *nb;
  const int bad = (insp && insp->bad) ? kNeedNonFinite : 0;
  const int need1 = ((insp && (insp->shifted || insp->bad || insp->d2 > 0.05f)) || sub) ? 1 : 0;
  *need = need1 | bad | (sub & (kNeedNonFinite | kNeedNullCols));
}
// threads: one per 4 x 4 tile of the upper triangle.  f64 keeps 2 x 16 doubles of tile data per thread: at 1024 threads
// (128 VGPRs) the compiler spilled 22 of them into the elimination loop (287 us at r = 138 against 62 us in f32), so
// the f64 instantiation is bounded at 768 threads (170 VGPRs, r <= 152; wider f64 factors take the 2 x 2 blocked form)
template <class T>
__host__ __device__ constexpr int chol_inv_max_threads() { return sizeof(T) == 8 ? 768 : 1024; }
template <class T>
__global__ __launch_bounds__(chol_inv_max_threads<T>()) void chol_inv_kernel(const T* __restrict__ g, int64_t ldg, int r, T piv_rel, T* m,
                                                        int64_t ldm, CholStatus* st, CholRobust rq = CholRobust{0.f, 0, 0.f, nullptr, nullptr, nullptr, 0.f, nullptr}) {
  if (rq.run_if && *rq.run_if == 0) return;
  const bool robust = rq.need_next != nullptr;
  // Register-resident Gaussian elimination of [G | I] in one sweep of r steps, one barrier per step.
  // Thread t owns the 4x4 tile (ti <= tk) of the upper triangle of G (v) and the same tile of W (w), where
  // W(x, y) = L^-1(y, x) for the unit-lower factor G = L U.  Step j: the owners of row j of the reduced G
  // publish it (rb), the owners of column j of W publish it (cb); then
  //     G(i, k) -= rb[i] rb[k] / d_j           (i > j),
  //     W(x, y) -= cb[x]  rb[y] / d_j          (x <= j < y).
  // With d_j the pivots, R = D^-1/2 U and R^-1(i, c) = W(i, c) / sqrt(d_c).  Buffers are double-buffered by
  // the parity of j so one barrier per step is enough.
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nt = (r + 3) >> 2, r4 = nt * 4;
  T* rowbuf = (T*)smem;          // [2][r4]
  T* colbuf = rowbuf + 2 * r4;   // [2][r4]
  T* diag0 = colbuf + 2 * r4;    // [r4] original diagonal
  T* dis = diag0 + r4;           // [r4] 1 / sqrt(d_j)
  float* red = (float*)(dis + r4);  // [32]
  const int tid = threadIdx.x, nwave = blockDim.x >> 6;
  const int ntri = nt * (nt + 1) / 2;
  const bool own = tid < ntri;
  int ti = 0, tk = 0;
  if (own) {
    // row-major over the upper triangle of tiles (ti ascending), so that whole waves retire from the
    // G-update once tj passes their rows
    const int e = ntri - 1 - tid;
    int c = (int)((sqrtf(8.f * (float)e + 1.f) - 1.f) * 0.5f);
    while (c * (c + 1) / 2 > e) --c;
    while ((c + 1) * (c + 2) / 2 <= e) ++c;
    ti = nt - 1 - c;
    tk = nt - 1 - (e - c * (c + 1) / 2);
  }
  const int i0 = 4 * ti, k0 = 4 * tk;
  T v[4][4], w[4][4];
  float dv = 0.f, gm = 0.f;
  int bad = 0;
  // the tile's 16 loads are unconditional (clamped indices) and issued together: one branch per element put an
  // s_waitcnt vmcnt(0) behind every load -- 16 dependent L2 round trips in front of the elimination
#pragma unroll
  for (int aa = 0; aa < 4; ++aa)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int i = min(i0 + aa, r - 1), k = min(k0 + b, r - 1);
      v[aa][b] = g[(int64_t)k * ldg + i];
    }
#pragma unroll
  for (int aa = 0; aa < 4; ++aa)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int i = i0 + aa, k = k0 + b;
      const bool in = own && i < r && k < r;
      const T x = in ? v[aa][b] : (T)0;
      if (in) {
        if (!((float)fabs(x) < 3.0e38f)) bad = 1;
        dv = fmaxf(dv, (float)fabs(x - (i == k ? (T)1 : (T)0)));
        if (i == k) gm = fmaxf(gm, (float)x);
      }
      v[aa][b] = x;
      w[aa][b] = (own && i == k && i < r) ? (T)1 : (T)0;
    }
  for (int idx = tid; idx < 4 * r4; idx += blockDim.x) rowbuf[idx] = (T)0;  // rowbuf and colbuf
  if (own && ti == tk) {
#pragma unroll
    for (int aa = 0; aa < 4; ++aa) diag0[i0 + aa] = v[aa][aa];
  }
  for (int off = 32; off > 0; off >>= 1) {
    dv = fmaxf(dv, __shfl_down(dv, off, 64));
    gm = fmaxf(gm, __shfl_down(gm, off, 64));
  }
  if ((tid & 63) == 0) {
    red[tid >> 6] = dv;
    red[16 + (tid >> 6)] = gm;
  }
  bad = __syncthreads_or(bad);
  float d2 = 0.f, g2 = 0.f;
  for (int i = 0; i < nwave; ++i) {
    d2 = fmaxf(d2, red[i]);
    g2 = fmaxf(g2, red[16 + i]);
  }
  int fl = bad ? 3 : (!(g2 > 0.f) ? 2 : 0);
  float min_ratio = 1.f;
  int nnull = 0;
  if (robust && fl == 2) {
    // the zero matrix: every column is null (M = 0, all of them are refilled)
    for (int j = tid; j < r; j += blockDim.x) rq.null_mask[j] = 1;
    if (own) {
#pragma unroll
      for (int aa = 0; aa < 4; ++aa)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int i = i0 + aa, k = k0 + b;
          if (i <= k && k < r) m[(int64_t)k * ldm + i] = (T)0;
        }
    }
    if (tid == 0) {
      *rq.need_next = 1 | kNeedNullCols;
      st->fail = 0;
      st->min_ratio = 0.f;
      st->dev_i = d2;
      st->gmax = 0.f;
      st->clk = 0;
      st->wall = 0;
    }
    return;
  }
  const long long clk0 = clock64(), wall0 = wall_clock64();
  // G = I + E with a tiny E (the polishing pass of CholeskyQR2): (I + E)^(-1/2) = I - E/2 + 3E^2/8 - ..., and
  // 3 ||E||^2 / 8 is below eps / 4, so the symmetric first-order factor replaces the elimination (uniform branch).
  const float series_tol = sizeof(T) == 4 ? 2.0e-4f : 1.0e-8f;
  if (fl == 0 && d2 <= series_tol) {
    if (own) {
#pragma unroll
      for (int aa = 0; aa < 4; ++aa)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int i = i0 + aa, k = k0 + b;
          if (i < r && k < r) {
            const T val = (i == k ? (T)1 : (T)0) - (T)0.5 * (v[aa][b] - (i == k ? (T)1 : (T)0));
            m[(int64_t)k * ldm + i] = val;
            m[(int64_t)i * ldm + k] = val;
          }
        }
    }
    if (robust)
      for (int j = tid; j < r; j += blockDim.x) rq.null_mask[j] = 0;
    if (tid == 0) {
      if (robust) *rq.need_next = 0;
      st->fail = 0;
      st->min_ratio = 1.f;
      st->dev_i = d2;
      st->gmax = g2;
      st->clk = 0;
      st->wall = 0;
    }
    return;
  }
  const T pre_sh = rq.abs_shift ? *(const T*)rq.abs_shift : (T)0;  // blocked form: the shift is in the diagonal already
  const bool shifted = rq.abs_shift ? pre_sh > (T)0
                                    : (robust && rq.shift_rel > 0.f && (rq.shift_mode == 1 || (rq.shift_mode == 0 && d2 > 0.25f)));
  const T sh = rq.abs_shift ? pre_sh : (shifted ? (T)rq.shift_rel * (T)g2 : (T)0);
  if (fl == 0 && shifted && !rq.abs_shift) {
    // shifted factorisation: G + s I, s relative to the largest diagonal entry (the diagonal tiles own the diagonal)
    if (own && ti == tk) {
#pragma unroll
      for (int aa = 0; aa < 4; ++aa)
        if (i0 + aa < r) v[aa][aa] += sh;
    }
  }
  if (fl == 0) {
    typedef T V4 __attribute__((ext_vector_type(4)));
    for (int tj = 0; tj < nt && fl == 0; ++tj) {
#pragma unroll
      for (int aj = 0; aj < 4; ++aj) {  // unrolled: the published register row / column is static
        const int j = 4 * tj + aj;
        if (j >= r) break;
        T* rb = rowbuf + (j & 1) * r4;
        T* cb = colbuf + (j & 1) * r4;
        if (own && ti == tj) *(V4*)&rb[k0] = V4{v[aj][0], v[aj][1], v[aj][2], v[aj][3]};
        if (own && tk == tj) *(V4*)&cb[i0] = V4{w[0][aj], w[1][aj], w[2][aj], w[3][aj]};
        __syncthreads();
        // every LDS read of the step is issued before the pivot test (one LDS round trip per step)
        const T d = rb[j];
        const T g0 = diag0[j];
        const V4 rk4 = *(const V4*)&rb[k0];
        const V4 ri4 = *(const V4*)&rb[i0];
        const V4 ci4 = *(const V4*)&cb[i0];
        if (!(d > piv_rel * g0) || !(g0 > (T)0) ||
            (shifted && rq.null_excess > 0.f && !(d - sh > (T)rq.null_excess * sh))) {  // uniform: every thread reads the same d
          if (!robust) {
            fl = 1;
            break;
          }
          // null column: no elimination step; its column of R^-1 is zero (dis = 0) and, the step being skipped, no
          // other column of R^-1 receives a contribution from it
          // (dis[j] = 0 marks it: the mask goes to global memory after the loop -- a global store inside it would be
          // waited for at every barrier)
          ++nnull;
          if (tid == 0) dis[j] = (T)0;
          continue;
        }
        if (tid < 64) {
          min_ratio = fminf(min_ratio, (float)d * fast_rcp((float)g0));
          if (tid == 0) {
            T rs = fast_rsqrt(d);
            rs = rs * ((T)1.5 - (T)0.5 * d * rs * rs);
            dis[j] = rs;
          }
        }
        if (own && tk >= tj) {  // tiles left of the pivot column are finished
          T rinv = fast_rcp(d);
          rinv = rinv * ((T)2 - d * rinv);
          T rk[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) rk[q] = rk4[q] * rinv;
          if (ti >= tj) {
            T ri[4] = {ri4[0], ri4[1], ri4[2], ri4[3]};
            if (ti == tj) {
#pragma unroll
              for (int q = 0; q < 4; ++q) ri[q] = (q > aj) ? ri[q] : (T)0;  // rows at or above the pivot stay
            }
#pragma unroll
            for (int aa = 0; aa < 4; ++aa)
#pragma unroll
              for (int b = 0; b < 4; ++b) v[aa][b] -= ri[aa] * rk[b];
          }
          if (ti <= tj) {
            T ci[4] = {ci4[0], ci4[1], ci4[2], ci4[3]};
            if (ti == tj) {
#pragma unroll
              for (int q = 0; q < 4; ++q) ci[q] = (q <= aj) ? ci[q] : (T)0;  // W(x, y): x <= j
            }
            if (tk == tj) {
#pragma unroll
              for (int q = 0; q < 4; ++q) rk[q] = (q > aj) ? rk[q] : (T)0;  // W(x, y): y > j
            }
#pragma unroll
            for (int aa = 0; aa < 4; ++aa)
#pragma unroll
              for (int b = 0; b < 4; ++b) w[aa][b] -= ci[aa] * rk[b];
          }
        }
      }
    }
  }
  const long long clk1 = clock64(), wall1 = wall_clock64();
  __syncthreads();
  if (robust && fl == 0)
    for (int j = tid; j < r; j += blockDim.x) rq.null_mask[j] = dis[j] == (T)0 ? 1 : 0;
  if (own) {
#pragma unroll
    for (int aa = 0; aa < 4; ++aa)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int i = i0 + aa, k = k0 + b;
        if (i <= k && k < r)
          m[(int64_t)k * ldm + i] = (fl == 0) ? w[aa][b] * dis[k] : ((i == k && fl == 1) ? (T)1 : (T)0);
      }
  }
  if (tid == 0) {
    // a plain pass on a Gram within 0.05 of I leaves the product orthonormal to a few eps * sqrt(m)
    if (robust) {
      int need;
      if (rq.need_ratio > 0.f)
        need = (nnull > 0 || min_ratio < rq.need_ratio || fl != 0) ? 1 : 0;
      else
        need = (nnull > 0 || d2 > 0.05f || (shifted && !rq.abs_shift) || fl != 0) ? 1 : 0;
      *rq.need_next = need | (fl == 3 ? kNeedNonFinite : 0) | (nnull > 0 ? kNeedNullCols : 0);
    }
    st->fail = fl;
    st->min_ratio = nnull > 0 ? 0.f : min_ratio;
    st->dev_i = d2;
    st->gmax = g2;
    st->clk = clk1 - clk0;
    st->wall = wall1 - wall0;
  }
}
// columns of y (m x l, column-major) marked in null_mask <- N(0, 1) / sqrt(m): the replacement of the null columns of
// a device Cholesky-QR pass (any orthonormal completion serves, random_svd.rs:38,57 -- a Householder thin-Q returns
// an arbitrary one as well); the following pass orthonormalises them against the rest.  grid = (blocks, l)
template <class T>
__global__ void refill_null_kernel(T* y, int64_t ld, int64_t m, int l, const int* null_mask, uint64_t seed, const int* run_if) {
  if (run_if && *run_if == 0) return;
  const int j = blockIdx.y;
  if (j >= l || null_mask[j] == 0) return;
  const T sc = (T)rsqrt((double)m);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x)
    y[(int64_t)j * ld + i] = sc * normal_from_index<T>((uint64_t)(i * l + j), seed);
}
__host__ __device__ inline int chol_inv_threads(int r) {
  const int nt = (r + 3) / 4;
  return ((nt * (nt + 1) / 2 + 63) / 64) * 64;
}
__host__ __device__ inline size_t chol_inv_lds_bytes(int r, size_t esz) {
  return (size_t)6 * (((size_t)r + 3) / 4 * 4) * esz + 32 * sizeof(float) + 64;
}
__host__ __device__ inline bool chol_inv_fits(int r, size_t esz) {
  return r >= 1 && chol_inv_threads(r) <= (esz == 8 ? 768 : 1024);  // r <= 176 (f32) / 152 (f64)
}

// ---- (I + E)^(-1/2) by its Taylor series, for the polishing pass of the Cholesky-QR ---------------
// g (r x r, column-major, ld) holds G = I + E on entry, E on exit
template <class T>
__global__ void series_prep_kernel(T* g, int64_t ld, int r) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i < r && j < r && i == j) g[(int64_t)j * ld + i] -= (T)1;
}
// m = I - E/2 + 3 E^2 / 8 - 5 E^3 / 16
template <class T>
__global__ void series_combine_kernel(const T* e1, const T* e2, const T* e3, int64_t ld, int r, T* m, int64_t ldm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i >= r || j >= r) return;
  const int64_t o = (int64_t)j * ld + i;
  m[(int64_t)j * ldm + i] = (i == j ? (T)1 : (T)0) - (T)0.5 * e1[o] + (T)0.375 * e2[o] - (T)0.3125 * e3[o];
}

// ---- non-finite cores -------------------------------------------------------------------------------
// One workgroup scans the l x l core of the small SVD (random_svd.rs:89) before a kernel family that has no status
// word of its own (ring / LDS / split / block Jacobi): *bad <- 1 and, when a status record is given, st->fail <- 3,
// so that a non-finite core ends the call with CORRLA_ENUMERIC instead of a triplet of zeros.  (The multi-workgroup
// Jacobi reports the same through jmc_init_kernel / jmc_finish_kernel.)
template <class T>
__global__ __launch_bounds__(1024) void core_finite_check_kernel(const T* __restrict__ c, int64_t ld, int l, CholStatus* st, int* bad_out) {
  int bad = 0;
  for (int idx = threadIdx.x; idx < l * l; idx += 1024) {
    const int j = idx / l, i = idx - j * l;
    if (!((float)fabs(c[(int64_t)j * ld + i]) < 3.0e38f)) bad = 1;
  }
  bad = __syncthreads_or(bad);
  if (threadIdx.x == 0 && bad) {
    if (st) st->fail = 3;
    if (bad_out) *bad_out = 1;
  }
}
// TEST HOOK (CORRLA_TEST_POISON_CORE, tests/test_gpu_parity.py): one entry of a device matrix <- NaN (kind 1) / +inf (2)
template <class T>
__global__ void poison_entry_kernel(T* p, int kind) {
  *p = kind == 2 ? std::numeric_limits<T>::infinity() : std::numeric_limits<T>::quiet_NaN();
}

// ---- PCA caller: centring (center_mat_col, mat_utils.rs:482-502) -----------------------------------
template <class T>
__global__ void fill_const_kernel(T* p, int64_t n, T v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
// out[r][c] = in[r][c] - (along_cols ? mu[c] : mu[r]); row-major in/out with leading dimensions ldi / ldo
template <class T>
__global__ void center_kernel(const T* in, int64_t rows, int64_t cols, int64_t ldi, const T* mu, int along_cols, T* out,
                              int64_t ldo) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  const T mc = along_cols ? mu[c] : (T)0;
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
    const T m_ = along_cols ? mc : mu[r];
    out[r * ldo + c] = in[r * ldi + c] - m_;
  }
}

// ---- implicit centring: rank-1 corrections of the products with A - 1 mu^T ----------------
// partial[b][c] = sum over this block's rows of (w ? w[r] : 1) * x(r, c), f64 accumulation, fixed order
template <class T>
__global__ __launch_bounds__(256) void wcolsum_partial_kernel(const T* __restrict__ x, int64_t ld, int64_t rows,
                                                              const T* __restrict__ w, double* partial, int ncols) {
  __shared__ double red[256];
  const int c = blockIdx.y;
  const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * per, r1 = min(rows, r0 + per);
  double s = 0.0;
  for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) s += (w ? (double)w[r] : 1.0) * (double)x[(int64_t)c * ld + r];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * ncols + c] = red[0];
}
template <class T>
__global__ void wcolsum_final_kernel(const double* partial, int nblk, int ncols, T* v) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncols) return;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += partial[(int64_t)b * ncols + c];
  v[c] = (T)s;
}
// out(i, c) -= scale * (u ? u[i] : 1) * v[c]
template <class T>
__global__ void rank1_sub_kernel(T* out, int64_t ld, int64_t rows, const T* __restrict__ u, const T* __restrict__ v,
                                 const T* __restrict__ scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  const T sc = scale ? *scale : (T)1;
  out[(int64_t)blockIdx.y * ld + i] -= sc * (u ? u[i] : (T)1) * v[blockIdx.y];
}

// ---- sign convention of the singular triplets ---------------------------------------------------
// The reference leaves the signs of (u_i, v_i) to faer's SVD.  Here triplet i is normalised so that the
// largest-magnitude component (first one on ties) of column i of the SHORT-side factor V_tall (n_t rows,
// replicated on every rank of a sharded run) is positive.  One workgroup per column finds the sign, then
// both factors are flipped.
// One workgroup per column finds the sign and flips its column of both factors (one launch: round 2 used a sign kernel
// and two apply kernels, 27 us of the C2 step's tail).
template <class T>
__global__ __launch_bounds__(256) void column_sign_apply_kernel(T* v, int64_t ld, int64_t rows, T* other, int64_t ld_o, int64_t rows_o) {
  const int j = blockIdx.x;
  T* col = v + (int64_t)j * ld;
  T best = (T)-1;
  int64_t best_i = 0;
  for (int64_t i0 = threadIdx.x; i0 < rows; i0 += 8 * (int64_t)blockDim.x) {  // eight loads in flight
    // (clamped, unconditional loads: behind a branch each one gets an s_waitcnt vmcnt(0) of its own)
    T a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t i = i0 + u * (int64_t)blockDim.x;
      a[u] = col[i < rows ? i : rows - 1];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = i0 + u * (int64_t)blockDim.x < rows ? (T)fabs(a[u]) : (T)-1;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (a[u] > best) {  // increasing i: the first maximum wins
        best = a[u];
        best_i = i0 + u * (int64_t)blockDim.x;
      }
  }
  __shared__ T sb[256];
  __shared__ int64_t si[256];
  __shared__ int flip;
  sb[threadIdx.x] = best;
  si[threadIdx.x] = best_i;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      const T ob = sb[threadIdx.x + off];
      const int64_t oi = si[threadIdx.x + off];
      if (ob > sb[threadIdx.x] || (ob == sb[threadIdx.x] && oi < si[threadIdx.x])) {
        sb[threadIdx.x] = ob;
        si[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) flip = (sb[0] > (T)0 && col[si[0]] < (T)0) ? 1 : 0;
  __syncthreads();
  if (!flip) return;
  for (int64_t i = threadIdx.x; i < rows; i += blockDim.x) col[i] = -col[i];
  T* oc = other + (int64_t)j * ld_o;
  for (int64_t i = threadIdx.x; i < rows_o; i += blockDim.x) oc[i] = -oc[i];
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
