// Exact k-nearest-neighbour scan of the active-subspace gradient stage, second generation (SURVEY 8 f2;
// PolyGradientEstimator::nearest_points, src/lib_math_utils/active_subspaces.rs:86-97): same results as
// knn_mfma_kernel (grad_kernels.hpp) -- the n nearest support points of every query by squared Euclidean distance in
// f64, equal distances by lower index -- at a fraction of its time.  What round 2's kernel spent (1e6 x 64, 80
// neighbours, 4.6 s for the stage): 64 f32 MFMAs per 64-point chunk and wave (2048 of ~5500 cycles per chunk), two
// workgroup barriers per chunk with one wave per SIMD, and ~755 wave-serial list insertions per query at ~1.3 us each.
//
// Here:
//  * FILTER on the bf16 matrix pipe: d^2(q, p) = |q|^2 + |p|^2 - 2 q.p with q.p from v_mfma_f32_16x16x32_bf16, both
//    operands split into two bf16 pieces of the CENTRED coordinates (hi hi + hi lo + lo hi: 2^-16 relative per product;
//    distances are translation invariant, and centring keeps |q|^2 + |p|^2 -- which the rounding error scales with --
//    as small as the data allow): 24 MFMAs of 16 cycles per chunk and wave instead of 64 of 32.  The support points are
//    split ONCE (knn2_prep_kernel) into MFMA B-fragment order, so a chunk is 16 KiB of linear LDS-DMA and every fragment
//    read is one conflict-free ds_read_b128.  A pair passes when  d^2_filter - margin (|q|^2 + |p|^2) < tau_q  (tau_q = the
//    query's current n-th exact distance; margin 2e-4 bounds every rounding of the filter) -- conservative, never exact;
//    the point's share of that test rides in as the MFMA's C operand, so the test is one compare per pair.
//  * SURVIVORS are only appended (4-byte index, ballot + popcount, no sorting) to a per-query candidate buffer.
//  * FLUSHES are batched and ALIGNED: after chunks 1, 2, 4, 8, ... (the expected number of survivors per query between
//    chunk c and 2c is n ln 2) every wave empties the buffers of its 32 queries at the same time -- the waves of a
//    workgroup share every staged chunk, so a flush that stalled one of them would stall them all.  A flush takes 64
//    candidates at a time, lane = candidate: exact f64 distance from the uncentred rows, a 64-key bitonic sort of
//    (distance, index), and a two-stage bitonic merge into the query's sorted list of 128 -- ~40 compare-exchange steps
//    for 64 candidates instead of 64 serial insertions.  A buffer that fills between flush points flushes on the spot.
//  * 16 scanning waves x 32 queries (two MFMA row tiles: every fragment read serves both) per workgroup share every staged
//    chunk, one workgroup per CU, persistent over the query tiles.  (8 waves x 16 queries sat on the LDS-DMA fill rate:
//    2 TB staged for 1e6 x 1e6 at ~6.4 TB/s; 16 waves x 16 queries on the fragment reads: 256 KiB per chunk and CU.)
// The neighbour SETS are exact whatever the filter does (a pair the filter drops is farther than the n-th exact
// distance by more than the bound on the filter's error); the ORDER inside a set is (exact distance, index).
// Limits: k <= 64, n_nbrs <= 128 (the host falls back to knn_mfma_kernel beyond).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "mixed_kernels.hpp"

namespace corrla {
namespace k {

constexpr int kK2Waves = 12;            // scanning waves per workgroup = 3 per SIMD (168 VGPRs: at 4 per SIMD the query fragments spill;
                                        // 8 waves x 4 row tiles at 256 VGPRs measured 446 vs 436 ms at 1e6 points; two chunks
                                        // per ring slot and barrier 455 ms: the longer live ranges put scratch into the loop)
constexpr int kK2RowTiles = 2;          // 16-query MFMA row tiles per wave: every B fragment read from LDS serves both
                                        // (with one, the 16 waves' fragment reads -- 256 KiB per chunk and CU at 128 B/clk
                                        // -- outweighed the MFMAs)
constexpr int kK2WQ = 16 * kK2RowTiles; // queries per wave
constexpr int kK2Q = kK2WQ * kK2Waves;  // queries per workgroup tile
constexpr int kK2Cap = 256;             // candidate slots per query between flushes
constexpr int kK2List = 128;            // list entries per query (n_nbrs <= 128)
constexpr int kK2Chunk = 64;            // support points per chunk
constexpr float kK2Margin = 2.0e-4f;    // bound on |d^2_filter - d^2| / (|q|^2 + |p|^2), see the header

__host__ __device__ constexpr int k2_chunk_bytes(int s) { return s * 8192; }            // s = 32-dimension MFMA steps
__host__ __device__ constexpr int k2_stage_bytes(int s) { return k2_chunk_bytes(s) + 1024; }  // + 64 x 4 f32: -c_p, replicated
// two stages + one 64-coordinate f64 row per wave (the query whose candidates are being re-checked)
constexpr int kK2Stages = 4;            // ring of staged chunks: three in flight behind the one being scanned (with one, every
                                        // chunk waited for its own DMA: 29 % of the scan at 1e6 points, CORRLA_KNN2_PROF)
__host__ __device__ constexpr int k2_lds_bytes(int s) { return kK2Stages * k2_stage_bytes(s) + kK2Waves * 512 + 1024; }

struct Knn2Args {
  const __bf16* pb;   // [chunk][s][plane (hi, lo)][tile t of 16 points][lane][8]: B fragments of the centred points
  const float* pn;    // [chunk][64][4]: -c_p = -(1 - margin) |x_p - mean|^2 / 2 in f32, four copies (one 16-byte MFMA C operand
                      // per point column); -inf for the padding points of the last chunk
  const double* x;    // support points, row-major n_pts x k
  const double* xq;   // queries, row-major n_q x k
  const double* mean; // [k]
  int64_t n_pts, n_q, nchunks, ntiles;
  int k, n_nbrs;
  int* cand;          // [gridDim.x][kK2Q][kK2Cap]
  double* list_d;     // [gridDim.x][kK2Q][kK2List]
  int* list_i;        // [gridDim.x][kK2Q][kK2List]
  int* nbr;           // out: [n_q][n_nbrs], nearest first
  unsigned long long* prof;  // optional (CORRLA_KNN2_PROF): 100 MHz ticks of wave 0 of every workgroup, summed:
                             // [0] in flushes, [1] waiting for the chunk (DMA + barrier), [2] in all, [3] list merges (batches)
};

// column means of x (n x k row-major), two stages in fixed order: partial[b][d] = sum over the rows of block b
__global__ __launch_bounds__(256) void knn2_colsum_kernel(const double* __restrict__ x, int64_t n, int k, int64_t rows_per_block,
                                                          double* partial) {
  __shared__ double red[256];
  const int d = threadIdx.x & 63, sub = threadIdx.x >> 6;  // 4 row phases x 64 dimensions
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
  double s = 0.0;
  if (d < k)
    for (int64_t r = r0 + sub; r < r1; r += 4) {
      // non-finite coordinates stay out of the mean: ONE NaN in the cloud would otherwise make every centred coordinate NaN,
      // every pair would pass the filter, and the scan would degenerate into the exact re-check of all N x N_q pairs
      // (the centre only has to be near the cloud; a non-finite point itself still goes through the re-check and sorts last)
      const double v = x[r * k + d];
      if (fabs(v) < 1.0e300) s += v;
    }
  red[threadIdx.x] = s;
  __syncthreads();
  if (sub == 0 && d < k) partial[(int64_t)blockIdx.x * 64 + d] = (red[d] + red[64 + d]) + (red[128 + d] + red[192 + d]);
}
__global__ void knn2_mean_kernel(const double* partial, int nblocks, int64_t n, int k, double* mean) {
  const int d = threadIdx.x;
  if (d >= 64) return;
  double s = 0.0;
  if (d < k)
    for (int b = 0; b < nblocks; ++b) s += partial[(int64_t)b * 64 + d];
  mean[d] = d < k ? s / (double)n : 0.0;
}

// one 256-thread block per chunk of 64 points: centred coordinates -> two bf16 pieces in B-fragment order, + norms.
// Fragment of v_mfma_f32_16x16x32_bf16: lane (col = lane & 15, g = lane >> 4) holds B[k = 8 g + j][col], j = 0..7.
__global__ __launch_bounds__(256) void knn2_prep_kernel(const double* __restrict__ x, int64_t n_pts, int k, const double* __restrict__ mean,
                                                        int nsteps, __bf16* pb, float* pn) {
  const int64_t c = blockIdx.x;
  const int tid = threadIdx.x;
  __bf16* dst = pb + c * (int64_t)(nsteps * 8192 / 2);
  for (int item = tid; item < nsteps * 4 * 64; item += 256) {  // (s, t, lane)
    const int lane = item & 63, t = (item >> 6) & 3, s = item >> 8;
    const int64_t p = c * kK2Chunk + 16 * t + (lane & 15);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int d = 32 * s + 8 * (lane >> 4) + j;
      v[j] = (p < n_pts && d < k) ? (float)(x[p * k + d] - mean[d]) : 0.0f;
    }
    // (the f32 rounding of the centred coordinate, 2^-24 relative, is part of the margin)
    bf16x8 fr[2];
    mx_split8<2>(v, fr);
    *(bf16x8*)(dst + ((s * 2 + 0) * 4 + t) * 512 + lane * 8) = fr[0];
    *(bf16x8*)(dst + ((s * 2 + 1) * 4 + t) * 512 + lane * 8) = fr[1];
  }
  if (tid < kK2Chunk) {
    const int64_t p = c * kK2Chunk + tid;
    float out = -__builtin_huge_valf();
    if (p < n_pts) {
      double s2 = 0.0;
      for (int d = 0; d < k; ++d) {
        const double df = x[p * k + d] - mean[d];
        s2 += df * df;
      }
      out = -(0.5f * (1.0f - kK2Margin)) * (float)s2;
    }
    *(f32x4*)(pn + (c * kK2Chunk + tid) * 4) = (f32x4){out, out, out, out};
  }
}

// ---- (distance, index) keys across the 64 lanes ----------------------------------------------------------------------
struct K2Key {
  double d;
  int i;
};
__device__ __forceinline__ bool k2_less(const K2Key& a, const K2Key& b) { return a.d < b.d || (a.d == b.d && a.i < b.i); }
// Lane exchanges of the sort / merge networks without the LDS crossbar: DPP inside a row of 16 lanes (quad permutes, row
// mirrors; lane ^ 4 and lane ^ 8 as two mirrors), gfx950's v_permlane16_swap / v_permlane32_swap across rows.  A network
// stage on ds_bpermute (what __shfl_xor compiles to) waits ~130 cycles for the crossbar; these are register moves.
typedef unsigned k2_u32x2 __attribute__((ext_vector_type(2)));
template <int CTRL>
__device__ __forceinline__ int k2_dpp(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
// value of lane ^ M (M one of 1, 2, 3, 4, 7, 8, 15, 16, 31, 32, 63)
template <int M>
__device__ __forceinline__ int k2_xchg(int v, int lane) {
  if constexpr (M == 1) return k2_dpp<0xB1>(v);        // quad_perm [1, 0, 3, 2]
  else if constexpr (M == 2) return k2_dpp<0x4E>(v);   // quad_perm [2, 3, 0, 1]
  else if constexpr (M == 3) return k2_dpp<0x1B>(v);   // quad_perm [3, 2, 1, 0]
  else if constexpr (M == 7) return k2_dpp<0x141>(v);  // row_half_mirror
  else if constexpr (M == 15) return k2_dpp<0x140>(v); // row_mirror
  else if constexpr (M == 4) return k2_dpp<0x1B>(k2_dpp<0x141>(v));   // 7 ^ 3
  else if constexpr (M == 8) return k2_dpp<0x141>(k2_dpp<0x140>(v));  // 15 ^ 7
  else if constexpr (M == 16) {
    // v_permlane16_swap: odd rows of the first operand <-> even rows of the second
    const k2_u32x2 r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    return (int)((lane & 16) ? r[0] : r[1]);
  } else if constexpr (M == 32) {
    // v_permlane32_swap: lanes 32..63 of the first operand <-> lanes 0..31 of the second
    const k2_u32x2 r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return (int)((lane & 32) ? r[0] : r[1]);
  } else if constexpr (M == 31) return k2_xchg<16>(k2_dpp<0x140>(v), lane);
  else {
    static_assert(M == 63, "unsupported lane exchange");
    return k2_xchg<32>(k2_xchg<16>(k2_dpp<0x140>(v), lane), lane);
  }
}
template <int M>
__device__ __forceinline__ K2Key k2_xchg_key(const K2Key& a, int lane) {
  K2Key r;
  r.d = __hiloint2double(k2_xchg<M>(__double2hiint(a.d), lane), k2_xchg<M>(__double2loint(a.d), lane));
  r.i = k2_xchg<M>(a.i, lane);
  return r;
}
// one compare-exchange stage with partner lane ^ M; `lower` lanes keep the smaller key
template <int M>
__device__ __forceinline__ void k2_stage(K2Key& a, int lane, bool lower) {
  const K2Key o = k2_xchg_key<M>(a, lane);
  if (k2_less(o, a) == lower) a = o;
}
// a bitonic sequence over the lanes -> ascending
__device__ __forceinline__ void k2_bitonic_merge(K2Key& a, int lane) {
  k2_stage<32>(a, lane, (lane & 32) == 0);
  k2_stage<16>(a, lane, (lane & 16) == 0);
  k2_stage<8>(a, lane, (lane & 8) == 0);
  k2_stage<4>(a, lane, (lane & 4) == 0);
  k2_stage<2>(a, lane, (lane & 2) == 0);
  k2_stage<1>(a, lane, (lane & 1) == 0);
}
// any sequence -> ascending: for every block size 2, 4, .. 64 a "flip" stage (partner = the mirror position inside the
// block) followed by the half-cleaners lane ^ (block / 4) .. lane ^ 1
__device__ __forceinline__ void k2_sort(K2Key& a, int lane) {
  k2_stage<1>(a, lane, (lane & 1) == 0);
  k2_stage<3>(a, lane, (lane & 2) == 0);
  k2_stage<1>(a, lane, (lane & 1) == 0);
  k2_stage<7>(a, lane, (lane & 4) == 0);
  k2_stage<2>(a, lane, (lane & 2) == 0);
  k2_stage<1>(a, lane, (lane & 1) == 0);
  k2_stage<15>(a, lane, (lane & 8) == 0);
  k2_stage<4>(a, lane, (lane & 4) == 0);
  k2_stage<2>(a, lane, (lane & 2) == 0);
  k2_stage<1>(a, lane, (lane & 1) == 0);
  k2_stage<31>(a, lane, (lane & 16) == 0);
  k2_stage<8>(a, lane, (lane & 8) == 0);
  k2_stage<4>(a, lane, (lane & 4) == 0);
  k2_stage<2>(a, lane, (lane & 2) == 0);
  k2_stage<1>(a, lane, (lane & 1) == 0);
  k2_stage<63>(a, lane, (lane & 32) == 0);
  k2_stage<16>(a, lane, (lane & 16) == 0);
  k2_stage<8>(a, lane, (lane & 8) == 0);
  k2_stage<4>(a, lane, (lane & 4) == 0);
  k2_stage<2>(a, lane, (lane & 2) == 0);
  k2_stage<1>(a, lane, (lane & 1) == 0);
}
// the keys in reverse lane order
__device__ __forceinline__ K2Key k2_reverse(const K2Key& a, int lane) { return k2_xchg_key<63>(a, lane); }

// Per-lane state of a wave's 32 queries, element e = 4 (row tile) + r for the query at accumulator register r of the lane's
// group.  (The flush was tried OUT OF LINE, to keep its ~100 registers of sort / merge / re-check state out of the scan
// loop's allocation: the values live across the call then lived in scratch for the whole loop -- worse.  What works is
// the loop structure of the kernel: an inner scan loop without any flush code, the flush between two runs of it.)
struct K2WaveState {
  float qn[4 * kK2RowTiles], cq[4 * kK2RowTiles];
  int cnt[4 * kK2RowTiles];
};
// Flush every query of this wave whose buffer holds at least `least` candidates: batch-merge them into the query's sorted
// list (64 at a time, lane = candidate: exact f64 distance, bitonic sort, two-stage bitonic merge) and refresh its filter
// threshold.  qs: the wave's 512-byte LDS row.
// (the kernel's argument block is passed field by field: taking its address moves the whole block to scratch)
__device__ __forceinline__ void k2_flush_wave(const double* __restrict__ x, const double* __restrict__ xq, int kdim, int n_nbrs,
                                                       K2WaveState* st, int least, int64_t q0, int* cand_w, double* ld_w,
                                                       int* li_w, double* qs, unsigned long long* n_batches) {
  const int lane = threadIdx.x & 63, fg = lane >> 4;
  const double inf = __builtin_huge_val();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's candidate stores have left it
#pragma unroll 1
  for (int qi = 0; qi < kK2WQ; ++qi) {
    const int gq = (qi >> 2) & 3, e = 4 * (qi >> 4) + (qi & 3);  // lane group and state element of the query
    const int ncand = __shfl(st->cnt[e], 16 * gq, 64);
    if (ncand < least || ncand == 0) continue;  // uniform
    const int64_t q = q0 + qi;
    K2Key l0, l1;
    l0.d = __hip_atomic_load(ld_w + qi * kK2List + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    l0.i = __hip_atomic_load(li_w + qi * kK2List + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    l1.d = __hip_atomic_load(ld_w + qi * kK2List + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    l1.i = __hip_atomic_load(li_w + qi * kK2List + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (the first 64 candidate indices travel with the list: one memory round trip fewer per query)
    const int cand0 = lane < ncand ? __hip_atomic_load(cand_w + qi * kK2Cap + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7fffffff;
    // the query's row goes to LDS once (wave-private): the re-check reads it as broadcasts instead of holding it in
    // registers next to the candidate's row.  (Loading list, row and the first candidate indices one query ahead was
    // tried: the flush got no faster and the scan loop's register allocation suffered -- 435 -> 532 ms.)
    qs[lane] = lane < kdim ? xq[q * kdim + lane] : 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int b0 = 0; b0 < ncand; b0 += 64) {
      if (n_batches) ++*n_batches;
      K2Key c;
      c.d = inf;
      c.i = 0x7fffffff;
      if (b0 + lane < ncand) {
        c.i = b0 == 0 ? cand0 : __hip_atomic_load(cand_w + qi * kK2Cap + b0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double* pp = x + (int64_t)c.i * kdim;
        double s0 = 0.0, s1 = 0.0;
        int d = 0;
        // 32, then 16 coordinates per memory round trip (a plain loop waits for every pair: 32 dependent round trips per
        // batch at k = 64); same summation order: even coordinates into s0, odd ones into s1, ascending
        for (; d + 32 <= kdim; d += 32) {
          double pv[32];
#pragma unroll
          for (int j = 0; j < 32; ++j) pv[j] = pp[d + j];
#pragma unroll
          for (int j = 0; j < 32; j += 2) {
            const double a0 = pv[j] - qs[d + j], a1 = pv[j + 1] - qs[d + j + 1];
            s0 += a0 * a0;
            s1 += a1 * a1;
          }
        }
        for (; d + 16 <= kdim; d += 16) {
          double pv[16];
#pragma unroll
          for (int j = 0; j < 16; ++j) pv[j] = pp[d + j];
#pragma unroll
          for (int j = 0; j < 16; j += 2) {
            const double a0 = pv[j] - qs[d + j], a1 = pv[j + 1] - qs[d + j + 1];
            s0 += a0 * a0;
            s1 += a1 * a1;
          }
        }
        for (; d + 1 < kdim; d += 2) {
          const double a0 = pp[d] - qs[d], a1 = pp[d + 1] - qs[d + 1];
          s0 += a0 * a0;
          s1 += a1 * a1;
        }
        if (d < kdim) {
          const double a0 = pp[d] - qs[d];
          s0 += a0 * a0;
        }
        c.d = s0 + s1;
        if (!(c.d == c.d)) c.d = inf;  // a non-finite distance sorts last, like numpy's argsort of a NaN
      }
      k2_sort(c, lane);
      // the 64 smallest of L1 and the candidates (ascending with descending: the elementwise minimum is bitonic) ...
      K2Key m = k2_reverse(c, lane);
      if (k2_less(l1, m)) m = l1;
      k2_bitonic_merge(m, lane);
      // ... then L0 against them: minima = the new L0, maxima = the new L1
      const K2Key rm = k2_reverse(m, lane);
      K2Key lo = l0, hi = rm;
      if (k2_less(rm, l0)) {
        lo = rm;
        hi = l0;
      }
      k2_bitonic_merge(lo, lane);
      k2_bitonic_merge(hi, lane);
      l0 = lo;
      l1 = hi;
    }
    __hip_atomic_store(ld_w + qi * kK2List + lane, l0.d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(li_w + qi * kK2List + lane, l0.i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(ld_w + qi * kK2List + 64 + lane, l1.d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(li_w + qi * kK2List + 64 + lane, l1.i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int nn = n_nbrs;
    const double tsel = nn <= 64 ? l0.d : l1.d;  // (nn is uniform: readlane)
    const int tl = nn <= 64 ? nn - 1 : nn - 65;
    const double tau = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(tsel), tl), __builtin_amdgcn_readlane(__double2loint(tsel), tl));
    // the filter compares in f32: round the threshold UP (never below the exact n-th distance); the few ulps the f32
    // evaluation of cq loses are part of the margin
    float tf = (float)tau;
    if ((double)tf < tau) tf = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, tf) + 1u);  // tau >= 0: next float up
    if (fg == gq) {
      st->cq[e] = tf < __builtin_huge_valf() ? 0.5f * ((1.0f - kK2Margin) * st->qn[e] - tf) : -__builtin_huge_valf();
      st->cnt[e] = 0;
    }
  }
}

template <int S>
__global__ __launch_bounds__(64 * kK2Waves, kK2Waves / 4) void knn2_kernel(Knn2Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STG = k2_stage_bytes(S);
  constexpr int NDMA = S * 8;  // 1 KiB LDS-DMA instructions per chunk (fragments); + one 256-byte one for the norms
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int kdim = g.k;
  int* const cand_w = g.cand + ((int64_t)blockIdx.x * kK2Q + wave * kK2WQ) * kK2Cap;
  double* const ld_w = g.list_d + ((int64_t)blockIdx.x * kK2Q + wave * kK2WQ) * kK2List;
  int* const li_w = g.list_i + ((int64_t)blockIdx.x * kK2Q + wave * kK2WQ) * kK2List;
  const double inf = __builtin_huge_val();

  // every wave issues the same number of DMA instructions per chunk (DPW: fragments, the norms, dummies into a scratch
  // KiB), so that a counted s_waitcnt leaves exactly the youngest chunk in flight (loads retire in order)
  constexpr int DPW = (NDMA + 1 + kK2Waves - 1) / kK2Waves;
  char* const dma_scratch = smem + kK2Stages * STG + kK2Waves * 512;
  auto stage = [&](int buf, int64_t c) __attribute__((always_inline)) {
    char* st = smem + buf * STG;
    const char* src = (const char*)g.pb + c * (int64_t)k2_chunk_bytes(S);
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
      const int i = wave + j * kK2Waves;
      if (i < NDMA)
        glds16(src + i * 1024 + lane * 16, st + i * 1024);
      else if (i == NDMA)
        glds16((const char*)(g.pn + c * (kK2Chunk * 4)) + lane * 16, st + k2_chunk_bytes(S));
      else
        glds16((const char*)g.pb + lane * 16, dma_scratch);  // (any readable KiB)
    }
  };

  unsigned long long t_flush = 0, t_wait = 0, n_batches = 0;
  const unsigned long long t_begin = g.prof ? wall_clock64() : 0;
  for (int64_t tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
    const int64_t q0 = tile * kK2Q + wave * kK2WQ;  // this wave's 32 queries: local query qi = 16 mw + (row of tile mw)
    // ---- A fragments: query q0 + 16 mw + fr, dimensions 32 s + 8 fg + j, centred, two bf16 pieces; |q - mean|^2 ----
    bf16x8 ah[kK2RowTiles][S], al[kK2RowTiles][S];
    float qn_mine[kK2RowTiles];
#pragma unroll
    for (int mw = 0; mw < kK2RowTiles; ++mw) {
      const int64_t q = q0 + 16 * mw + fr;
      const bool qv = q < g.n_q;
      float qn = 0.f;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int d = 32 * s + 8 * fg + j;
          v[j] = (qv && d < kdim) ? (float)(g.xq[q * kdim + d] - g.mean[d]) : 0.0f;
          qn += v[j] * v[j];
        }
        bf16x8 f2[2];
        mx_split8<2>(v, f2);
        ah[mw][s] = f2[0];
        al[mw][s] = f2[1];
      }
      qn += __shfl_xor(qn, 16, 64);
      qn += __shfl_xor(qn, 32, 64);  // every lane with this fr now holds |q_fr|^2 (f32: inside the margin)
      qn_mine[mw] = qn;
    }
    // D layout of row tile mw: column (point) = lane & 15, row (query) = 16 mw + 4 fg + r; per-query state at [4 mw + r]
    // (ext-vector registers with constant indices: plain arrays captured by the flush lambdas ended up in scratch)
    typedef int i32x8 __attribute__((ext_vector_type(4 * kK2RowTiles)));
    typedef float f32x8 __attribute__((ext_vector_type(4 * kK2RowTiles)));
    // The filter per pair:  d^2_filter - margin (qn + pn) < tau   <=>   q.p - cp > cq  with
    //   cq = ((1 - margin) qn - tau) / 2  per query (changes at a flush),  cp = (1 - margin) pn / 2  per point, which the
    // accumulators start from (see the chunk loop): one compare per pair, writing the lane mask the slow path needs anyway.
    // cq = -inf while the list is not full (everything passes), +inf for the padding queries of the last tile
    // (nothing passes); padding points carry -cp = -inf.
    f32x8 qn_r, cq_r;
    i32x8 cnt_r;
#pragma unroll
    for (int mw = 0; mw < kK2RowTiles; ++mw)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        qn_r[4 * mw + r] = __shfl(qn_mine[mw], 4 * fg + r, 64);
        cq_r[4 * mw + r] = (q0 + 16 * mw + 4 * fg + r < g.n_q) ? -__builtin_huge_valf() : __builtin_huge_valf();
        cnt_r[4 * mw + r] = 0;
      }
    // empty lists
    for (int e = lane; e < kK2WQ * kK2List; e += 64) {
      __hip_atomic_store(ld_w + e, inf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(li_w + e, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    if (g.nchunks > 0) stage(0, 0);
    if (g.nchunks > 1) stage(1, 1);
    if (g.nchunks > 2) stage(2, 2);
    int buf = 0;
    // Two loop levels: the INNER loop scans chunks and holds no flush code at all (so the register allocator keeps the
    // scan's values in registers and parks what only the flush needs outside it); it ends when a flush is due -- an aligned
    // flush point, or a buffer that could overflow in the next chunk -- and the flush runs between two runs of it.
    int64_t c = 0;
    while (c < g.nchunks) {
    int least = 0;
#pragma unroll 1
    for (; c < g.nchunks && least == 0; ++c) {
      const unsigned long long tw0 = g.prof ? wall_clock64() : 0;
      // This wave's share of chunk c has landed; chunks c + 1 and c + 2 (the youngest 2 DPW loads) may stay in flight.
      // The DMA of chunk c + 3 is issued at the END of the iteration, after the candidate appends: stores count in vmcnt
      // too, and appends issued behind a DMA made the counted wait cover that DMA as well (one chunk of prefetch lost).
      // Appends in between can only make the wait longer, never let it pass early: loads retire in order.
      if (c + 2 < g.nchunks)
        wait_vmcnt<2 * DPW>();
      else if (c + 1 < g.nchunks)
        wait_vmcnt<DPW>();
      else
        wait_vmcnt<0>();
      __syncthreads();                                    // chunk c is complete; every wave is done with chunk c - 1
      if (g.prof) t_wait += wall_clock64() - tw0;
      const char* st = smem + buf * STG;
      const int buf_prev = buf >= 1 ? buf - 1 : kK2Stages - 1;  // the slot of chunk c - 1 = that of chunk c + 3
      buf = buf + 1 == kK2Stages ? 0 : buf + 1;
      // The accumulators start from -c_p (the C operand of the first MFMA of every tile: one 16-byte read per point column,
      // shared by both row tiles), so that the filter test is ONE compare per pair:  q.p - c_p > c_q.  Padding points carry
      // -inf and never pass (-inf > c_q is false for every c_q, -inf included).
      f32x4 acc[kK2RowTiles][4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 ncp = *(const f32x4*)(st + k2_chunk_bytes(S) + (16 * t + fr) * 16);
#pragma unroll
        for (int mw = 0; mw < kK2RowTiles; ++mw) acc[mw][t] = ncp;
      }
      // The fragments of step (s, t + 1) are read under the MFMAs of step (s, t): inline-asm ds_reads one step ahead, retired
      // by hand-counted waits (hip_kernels.hpp: hipcc otherwise issues one of the two reads late and waits for it at once,
      // and a sched_group_barrier sequence lands one read off).
      {
        const unsigned fa = lds_addr(st) + (unsigned)lane * 16u;
        bf16x8 fh[2], fl[2];
        lds_read_b128<0>(fh[0], fa);
        lds_read_b128<4 * 1024>(fl[0], fa);
        static_for<0, S * 4>([&](auto ic) {
          constexpr int i = decltype(ic)::value, s_ = i >> 2, t = i & 3, cur = i & 1, nxt = cur ^ 1;
          if constexpr (i + 1 < S * 4) {
            constexpr int s2 = (i + 1) >> 2, t2 = (i + 1) & 3;
            lds_read_b128<((s2 * 2 + 0) * 4 + t2) * 1024>(fh[nxt], fa);
            lds_read_b128<((s2 * 2 + 1) * 4 + t2) * 1024>(fl[nxt], fa);
          }
          lds_wait<(i + 1 < S * 4) ? 2 : 0>(fh[cur]);
          lds_tie(fl[cur]);
#pragma unroll
          for (int mw = 0; mw < kK2RowTiles; ++mw) {
            acc[mw][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mw][s_], fh[cur], acc[mw][t], 0, 0, 0);
            acc[mw][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mw][s_], fl[cur], acc[mw][t], 0, 0, 0);
            acc[mw][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mw][s_], fh[cur], acc[mw][t], 0, 0, 0);
          }
        });
      }
      const int base = (int)(c * kK2Chunk);
#pragma unroll
      for (int mw = 0; mw < kK2RowTiles; ++mw) {
        // survivors are sparse (~2 per wave and chunk once the lists are full): the masks are OR-ed per column tile, so
        // that a chunk with one survivor costs ~10 scalar tests instead of one (taken) branch per (tile, register)
        unsigned long long hm[4][4], at[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            // negated comparison: a NaN (a non-finite point or query) goes through to the exact re-check
            hm[t][r] = __ballot(!(acc[mw][t][r] <= cq_r[4 * mw + r]));
          at[t] = (hm[t][0] | hm[t][1]) | (hm[t][2] | hm[t][3]);
        }
        if (((at[0] | at[1]) | (at[2] | at[3])) != 0) {  // uniform
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            if (at[t] == 0) continue;  // uniform
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const unsigned long long m = hm[t][r];
              if (m == 0) continue;  // uniform
              const unsigned gm = (unsigned)(m >> (16 * fg)) & 0xffffu;  // the 16 points of MY query (16 mw + 4 fg + r)
              const bool h = (gm >> fr) & 1u;
              const int slot = cnt_r[4 * mw + r] + __popc(gm & ((1u << fr) - 1u));
              if (h && slot < kK2Cap) cand_w[(16 * mw + 4 * fg + r) * kK2Cap + slot] = base + 16 * t + fr;
              cnt_r[4 * mw + r] += __popc(gm);
            }
          }
        }
      }
      if (c + 3 < g.nchunks) stage(buf_prev, c + 3);
      // aligned flush points: after chunks 1, 2, 4, 8, ... and the last one; in between only a buffer that could
      // overflow in the next chunk (64 more candidates) is flushed
      const int64_t done = c + 1;
      const bool point = (done & (done - 1)) == 0 || done == g.nchunks;
      int cmax = cnt_r[0];
#pragma unroll
      for (int u = 1; u < 4 * kK2RowTiles; ++u) cmax = max(cmax, cnt_r[u]);
      least = point ? 1 : (__any(cmax > kK2Cap - kK2Chunk) ? kK2Cap - kK2Chunk + 1 : 0);
    }
      if (least) {
        const unsigned long long tf0 = g.prof ? wall_clock64() : 0;
        K2WaveState stv;
#pragma unroll
        for (int u = 0; u < 4 * kK2RowTiles; ++u) {
          stv.qn[u] = qn_r[u];
          stv.cq[u] = cq_r[u];
          stv.cnt[u] = cnt_r[u];
        }
        k2_flush_wave(g.x, g.xq, kdim, g.n_nbrs, &stv, least, q0, cand_w, ld_w, li_w,
                      (double*)(smem + kK2Stages * STG) + wave * 64, g.prof ? &n_batches : nullptr);
#pragma unroll
        for (int u = 0; u < 4 * kK2RowTiles; ++u) {
          cq_r[u] = stv.cq[u];
          cnt_r[u] = stv.cnt[u];
        }
        if (g.prof) t_flush += wall_clock64() - tf0;
      }
    }
    // ---- results: the first n_nbrs entries of every list, nearest first ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
    for (int qi = 0; qi < kK2WQ; ++qi) {
      const int64_t q = q0 + qi;
      if (q >= g.n_q) break;  // uniform
      for (int e = lane; e < g.n_nbrs; e += 64)
        g.nbr[q * g.n_nbrs + e] = __hip_atomic_load(li_w + qi * kK2List + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();  // the stage buffers are free for the next tile
  }
  if (g.prof && threadIdx.x == 0) {
    atomicAdd(g.prof + 0, t_flush);
    atomicAdd(g.prof + 1, t_wait);
    atomicAdd(g.prof + 2, wall_clock64() - t_begin);
    atomicAdd(g.prof + 3, n_batches);
  }
}

}  // namespace k
}  // namespace corrla
