// Householder TSQR with an explicit thin Q (gfx950) -- the optional orthonormalisation behind CORRLA_QR_HOUSEHOLDER.
//
// The reference orthonormalises its sketch with faer's Householder QR, `y_mat.qr().compute_thin_q()`
// (src/lib_math_utils/random_svd.rs:38,57).  The default path of this library is CholeskyQR2 (three GEMM-shaped
// passes on the MFMA units); this file is the communication-avoiding Householder form of the same operation:
//   up   : the m x l sketch is cut into row panels of at most 2 l rows.  One workgroup per panel keeps its panel in
//          LDS and reduces it column by column with Householder reflectors (each wave owns columns of the trailing
//          update, the 64 lanes of a wave split the rows of a column, dot products are wave reductions); the l x l
//          R factors are then combined pairwise, [R_a; R_b] -> R, level by level, by the same panel kernel.
//   down : the explicit Q is formed by applying the stored reflectors in reverse: the root turns [I; 0] into two
//          l x l coefficient blocks, every inner node turns its coefficient block into two, every leaf turns its
//          block into its rows of Q.  In this direction a wave owns its columns for the whole panel, so the
//          apply loop needs no barriers at all.
// Two panel codes: the unblocked one (hh_factor_panel / hh_apply_panel: every reflector applied to the whole trailing
// panel by the VALU) and the blocked compact-WY one (hh_wy_*: reflector steps inside 16-column blocks, trailing update
// and down sweep on the MFMA units), selected per context (CORRLA_HH_WY, default blocked).
// A Householder thin-Q is orthonormal whatever the rank of the sketch (tau = 0 for a column that is already reduced),
// which is exactly the behaviour of the reference on rank-deficient inputs.  One panel (2 l x l) must fit in the
// 160 KB LDS of a CU: l <= 142 (f32) / l <= 99 (f64).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <limits>
#include <type_traits>

#include "hip_kernels.hpp"  // MT<T>: the MFMA wrappers and their register layouts

namespace corrla {
namespace k {

constexpr int kHhThreads = 1024;  // 16 waves per panel: four per SIMD hide the load -> reduce -> update chain of a column group
constexpr int kHhWaves = kHhThreads / 64;
constexpr int kHhGroup = 3;  // columns of one wave in flight together (138 columns = 16 waves x 3 groups of 3)
constexpr int kHhMaxRowsPerLane = 5;  // ceil(2 * 138 / 64): panel rows a lane may own in one column
constexpr int kHhTriSlices = 3;       // ceil(138 / 64): rows l .. l + j of a stacked-triangle panel

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float hh_dpp_f(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xf, true));
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double hh_dpp_f(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float hh_readlane63(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}
__device__ __forceinline__ double hh_readlane63(double x) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 63), __builtin_amdgcn_readlane(__double2loint(x), 63));
}
// sum over the 64 lanes, result in every lane: DPP inside the 16-lane rows, then the GFX9 row broadcasts (row 0 -> 1,
// row 2 -> 3, then rows 0+1 -> 2, 3; masked-off rows receive 0 through bound_ctrl) leave the total in lane 63, which
// is read back through a scalar register -- no LDS-crossbar exchange on the dependent chain
template <class T>
__device__ __forceinline__ T hh_wave_sum(T x) {
  x += hh_dpp_f<0xB1>(x);        // quad_perm [1,0,3,2]
  x += hh_dpp_f<0x4E>(x);        // quad_perm [2,3,0,1]
  x += hh_dpp_f<0x141>(x);       // row_half_mirror
  x += hh_dpp_f<0x140>(x);       // row_mirror: every lane holds the sum of its row
  x += hh_dpp_f<0x142, 0xa>(x);  // row_bcast:15 into rows 1 and 3
  x += hh_dpp_f<0x143, 0xc>(x);  // row_bcast:31 into rows 2 and 3
  return hh_readlane63(x);
}

// Barrier that orders LDS traffic only.  The panel loops store tau / T values to global memory that this kernel never
// reads back; a __syncthreads() waits for every outstanding global store as well (vmcnt(0)), i.e. a full write round trip
// (~2 us) per reflector step -- which was the whole cost of a step.
__device__ __forceinline__ void hh_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__host__ __device__ inline int hh_pitch(int rows) { return (rows + 3) & ~3; }
__host__ __device__ inline size_t hh_lds_bytes(int max_rows, int l, size_t esz) {
  return (size_t)hh_pitch(max_rows) * (size_t)l * esz + 64;
}
// rows of leaf i when m rows are cut into nleaf panels
__host__ __device__ inline int64_t hh_leaf_row0(int64_t m, int nleaf, int i) { return (m * (int64_t)i) / nleaf; }

// In-LDS Householder reduction of the rows x l panel P (column pitch RP).  On return the upper triangle holds R and
// column j holds v_j below the diagonal (v_j(j) = 1 implied); tau[j] is written to global memory by wave 0.
// TRI: the panel is two stacked upper triangles [R_a; R_b] (rows = 2 l).  Reflector j then only involves row j and rows
// l .. l + j (everything else of column j below the diagonal is, and stays, zero): NS = 3 row slices instead of 5.
template <class T, int NS, bool TRI>
__device__ void hh_factor_panel(T* P, int RP, int rows, int l, T* tau_out) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int j = 0; j < l; ++j) {
    T* cj = P + (size_t)j * RP;
    const int lo = TRI ? l : j + 1, hi = TRI ? l + j + 1 : rows;  // rows [lo, hi) carry v below the diagonal
    // every wave forms the reflector redundantly (identical arithmetic -> identical values): no exchange needed
    T xr[NS];
    T part = (T)0;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int r = lo + lane + 64 * i;
      const T val = cj[min(r, hi - 1)];  // unconditional load (clamped), then select: the five loads overlap
      xr[i] = r < hi ? val : (T)0;
      part += xr[i] * xr[i];
    }
    const T sigma = hh_wave_sum(part);
    const T alpha = cj[j];
    T tau = (T)0, beta = alpha, scale = (T)0;
    if (sigma > (T)0) {  // LAPACK xLARFG
      beta = -copysign(sqrt(alpha * alpha + sigma), alpha);
      tau = (beta - alpha) / beta;
      scale = (T)1 / (alpha - beta);
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) xr[i] *= scale;  // v below the diagonal
    hh_lds_barrier();  // everyone has read column j
    if (wave == 0) {
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int r = lo + lane + 64 * i;
        if (r < hi) cj[r] = xr[i];
      }
      if (lane == 0) {
        cj[j] = beta;
        tau_out[j] = tau;
      }
    }
    // trailing update, kHhGroup columns of this wave at a time: every LDS load is unconditional (clamped row / column
    // index; v is zero on the rows that do not exist) so that the twenty loads of a group are in flight together,
    // the four reductions overlap, and the update is made from registers
    int rr[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) rr[i] = min(lo + lane + 64 * i, hi - 1);
    // columns are dealt round-robin (column j + 1 + wave + 16 i), kHhGroup of a wave's columns at a time
    for (int c0 = j + 1 + wave; c0 < l; c0 += kHhGroup * kHhWaves) {
      T pv[kHhGroup][NS], pj[kHhGroup], d[kHhGroup];
#pragma unroll
      for (int u = 0; u < kHhGroup; ++u) {
        const T* cc = P + (size_t)min(c0 + u * kHhWaves, l - 1) * RP;
        pj[u] = cc[j];
#pragma unroll
        for (int i = 0; i < NS; ++i) pv[u][i] = cc[rr[i]];
      }
#pragma unroll
      for (int u = 0; u < kHhGroup; ++u) {
        d[u] = (T)0;
#pragma unroll
        for (int i = 0; i < NS; ++i) d[u] += xr[i] * pv[u][i];
      }
#pragma unroll
      for (int u = 0; u < kHhGroup; ++u) d[u] = hh_wave_sum(d[u]);
#pragma unroll
      for (int u = 0; u < kHhGroup; ++u) {
        const int c = c0 + u * kHhWaves;
        if (c < l) {
          T* cc = P + (size_t)c * RP;
          const T w = tau * (d[u] + pj[u]);
#pragma unroll
          for (int i = 0; i < NS; ++i) {
            const int r = lo + lane + 64 * i;
            if (r < hi) cc[r] = pv[u][i] - w * xr[i];
          }
          if (lane == 0) cc[j] = pj[u] - w;
        }
      }
    }
    hh_lds_barrier();  // column j + 1 is complete before the next reflector is formed
  }
}

// P <- H_0 H_1 ... H_{l-1} P for the rows x l panel P in LDS; reflectors (V, leading dimension ldv) and tau in global
// memory.  A wave owns its columns throughout: no barriers.
template <class T, int NS, bool TRI>
__device__ void hh_apply_panel(T* P, int RP, int rows, int l, const T* __restrict__ V, int64_t ldv,
                               const T* __restrict__ tau_in) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  T vn[NS];
  auto load_v = [&](int j) {
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int lo_ = TRI ? l : j + 1, hi_ = TRI ? l + max(j, 0) + 1 : rows;
      const int r = lo_ + lane + 64 * i;
      const T val = V[(int64_t)max(j, 0) * ldv + min(r, hi_ - 1)];  // unconditional, clamped
      vn[i] = (j >= 0 && r < hi_) ? val : (T)0;
    }
  };
  load_v(l - 1);
  for (int j = l - 1; j >= 0; --j) {
    T v[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) v[i] = vn[i];
    const T tau = tau_in[j];
    load_v(j - 1);  // the next reflector travels while this one is applied
    if (tau == (T)0) continue;
    const int lo = TRI ? l : j + 1, hi = TRI ? l + j + 1 : rows;
    int rr[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) rr[i] = min(lo + lane + 64 * i, hi - 1);
    for (int c0 = wave; c0 < l; c0 += kHhGroup * kHhWaves) {
      T pv[kHhGroup][NS], pj[kHhGroup], d[kHhGroup];
#pragma unroll
      for (int u = 0; u < kHhGroup; ++u) {
        const T* cc = P + (size_t)min(c0 + u * kHhWaves, l - 1) * RP;
        pj[u] = cc[j];
#pragma unroll
        for (int i = 0; i < NS; ++i) pv[u][i] = cc[rr[i]];
      }
#pragma unroll
      for (int u = 0; u < kHhGroup; ++u) {
        d[u] = (T)0;
#pragma unroll
        for (int i = 0; i < NS; ++i) d[u] += v[i] * pv[u][i];
      }
#pragma unroll
      for (int u = 0; u < kHhGroup; ++u) d[u] = hh_wave_sum(d[u]);
#pragma unroll
      for (int u = 0; u < kHhGroup; ++u) {
        const int c = c0 + u * kHhWaves;
        if (c < l) {
          T* cc = P + (size_t)c * RP;
          const T w = tau * (d[u] + pj[u]);
#pragma unroll
          for (int i = 0; i < NS; ++i) {
            const int r = lo + lane + 64 * i;
            if (r < hi) cc[r] = pv[u][i] - w * v[i];
          }
          if (lane == 0) cc[j] = pj[u] - w;
        }
      }
    }
  }
}

// ---- blocked (compact WY) panels: reflector steps inside 16-column blocks, everything else on the MFMA units ------------
// H_0 ... H_{15} of one block = I - V T V^T (V: unit lower trapezoidal, T: 16 x 16 upper triangular, LAPACK xLARFT).
//   factor : (a) the block's columns live in the registers of one wave each; reflector jj is formed by its owner, written
//                to LDS, and applied by the waves of the later columns of the block: ONE barrier per reflector and no
//                work outside the block;
//            (b) wave 0 takes G = V^T V with one MFMA chain and runs the T recurrence (lane i = row i of T);
//            (c) trailing columns, one 16-column tile per wave:  W = V^T A (MFMA, K = rows),  TW = T^T W,  A -= V TW.
//                W and TW never leave the registers: the D registers of one product are the B operand of the next, with
//                the reduction index permuted to the D layout (step i of a 16-deep product takes k = drow(lane, i)).
//   apply  : P <- (I - V T V^T) P block by block in reverse, V and T streamed from global memory (L1 / L2 resident; the
//            panel fills the LDS); a wave owns its 16-column tiles for the whole panel: no barriers.
// tau = 0 (a column that is already reduced) gives a zero column in T: the reflector drops out, as in the unblocked code.
constexpr int kWyNb = 16;
// (A variant in which ONE wave owns a whole 16-column block of a tree node in registers -- no barrier between reflector
// steps -- measured slower, 142 -> 184 us per node in f32 and worse in f64, and was removed in round 3: CHANGELOG.md.)
constexpr int kWyPipe = 8;      // k-steps whose operand loads are issued before their MFMAs
constexpr int kWyRowTiles = 3;  // 16-row tiles of the rank-16 update in flight together
// down sweep: V streams from global memory (L1 / L2)
constexpr int kWyApplyPipe = 8;   // (16 / 6 measured slower: the masked last trip grows with the depth)
constexpr int kWyApplyRowTiles = 3;
static_assert(kHhWaves >= kWyNb, "one wave per column of a block");
static_assert(kHhMaxRowsPerLane * 64 / 2 / 16 + 1 <= kHhWaves - 1, "trailing tiles of a block: one wave each, the last wave builds T");
constexpr int kWyExtra = 2 * kWyNb * kWyNb + kWyNb;  // Ts, Rs (parked diagonal block), tau_s (elements) in front of the panel
__host__ __device__ inline size_t hh_wy_lds_bytes(int max_rows, int l, size_t esz) {
  return hh_lds_bytes(max_rows, l, esz) + (size_t)kWyExtra * esz;
}
__host__ __device__ inline int hh_wy_panels(int l) { return (l + kWyNb - 1) / kWyNb; }

// entry (r, c) of the block's V as the MFMA operands need it.  V is stored CLEAN (explicit 1 on the diagonal, 0 above
// it: the factor kernels park the R entries of the block's 16 x 16 diagonal block in LDS while the block is being applied,
// and write clean reflectors to global memory for the down sweep), so only the range masks remain.  Columns past the
// block's end are read clamped (finite values) and meet zero rows / columns of T.
template <class T>
__device__ __forceinline__ T wy_v(const T* __restrict__ V, int64_t ldv, int rows, int r, int c, int cend) {
  const T val = V[(int64_t)min(c, cend - 1) * ldv + min(r, rows - 1)];  // unconditional (clamped) load, then select
  return r < rows ? val : (T)0;
}
// which D register / lane group of an MFMA result holds row k (inverse of MT<T>::drow)
template <class T>
struct WyD;
template <>
struct WyD<float> {
  static constexpr int reg(int k) { return k & 3; }
  static constexpr int grp(int k) { return k >> 2; }
};
template <>
struct WyD<double> {
  static constexpr int reg(int k) { return k >> 2; }
  static constexpr int grp(int k) { return k & 3; }
};
// reciprocal / square root of the reflector scalars: hardware approximation + Newton steps to (nearly) the last bit
// instead of the IEEE division / square-root sequences, which sat on the critical path of every reflector step
__device__ __forceinline__ float hh_rcp(float x) {
  const float y = __builtin_amdgcn_rcpf(x);
  return fmaf(y, fmaf(-x, y, 1.0f), y);
}
__device__ __forceinline__ double hh_rcp(double x) {
  const double y = jr_rcp(x);
  return fma(y, fma(-x, y, 1.0), y);
}
__device__ __forceinline__ float hh_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ double hh_sqrt(double x) {
  const double y = jr_rsq(x), s0 = x * y;
  return fma(0.5 * y, fma(-s0, s0, x), s0);
}
__device__ __forceinline__ float hh_readlane(float x, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
}
__device__ __forceinline__ double hh_readlane(double x, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}

// w0 + sum over rows [k0, rows) of A(r) B(r)^T contributions of an MFMA product whose operands are r-contiguous columns
// (acol / bcol: this lane's column; lane group g supplies row 4 s + g of step s).  kWyPipe steps per trip, their loads
// issued together before their MFMAs; whole trips read through one address with immediate offsets (no clamps, no
// masks) and only the last partial trip is masked.  (Prefetching the next trip across the MFMAs measured slower: the
// register copies and the second accumulator cost more than the exposed LDS latency.)
template <class T, int PIPE>
__device__ __forceinline__ typename MT<T>::acc_t wy_kdot(const T* acol, const T* bcol, int k0, int rows, int g,
                                                         typename MT<T>::acc_t w0) {
  typedef MT<T> M;
  constexpr int kWyPipe = PIPE;  // (shadows the default depth)
  auto load = [&](int kk, T(&av)[kWyPipe], T(&bv)[kWyPipe]) {
    if (kk + 4 * kWyPipe <= rows) {
      const T* qa = acol + kk + g;
      const T* qb = bcol + kk + g;
#pragma unroll
      for (int u = 0; u < kWyPipe; ++u) {
        av[u] = qa[4 * u];
        bv[u] = qb[4 * u];
      }
    } else {
#pragma unroll
      for (int u = 0; u < kWyPipe; ++u) {
        const int r = kk + 4 * u + g, rc = min(r, rows - 1);
        const T xa = acol[rc], xb = bcol[rc];
        av[u] = r < rows ? xa : (T)0;
        bv[u] = r < rows ? xb : (T)0;
      }
    }
  };
  for (int kk = k0; kk < rows; kk += 4 * kWyPipe) {
    T a0[kWyPipe], b0[kWyPipe];
    load(kk, a0, b0);
#pragma unroll
    for (int u = 0; u < kWyPipe; ++u) w0 = M::mma(a0[u], b0[u], w0);
  }
  return w0;
}

// This lane's tile column pc (rows [rbeg, rows)) -= V(:, block) TW: 16-row tiles, kWyRowTiles per trip (loads, MFMAs,
// stores); whole trips unmasked, the last partial trip masked.
template <class T, int RT>
__device__ __forceinline__ void wy_update(T* pc, bool cok, const T* V, int64_t ldv, int j0, int cend, int rbeg, int rows,
                                          typename MT<T>::acc_t tw, int lane) {
  typedef MT<T> M;
  typedef typename M::acc_t acc_t;
  constexpr int STEP = 16 * RT;
  const int n16 = lane & 15;
  const T* vc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) vc[i] = V + (int64_t)min(j0 + M::drow(lane, i), cend - 1) * ldv;
  auto load = [&](int r0, acc_t(&c)[RT], T(&va)[RT][4]) {
    if (r0 + STEP <= rows) {
#pragma unroll
      for (int u = 0; u < RT; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          c[u][i] = pc[r0 + 16 * u + M::drow(lane, i)];
          va[u][i] = -vc[i][r0 + 16 * u + n16];
        }
    } else {
#pragma unroll
      for (int u = 0; u < RT; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = r0 + 16 * u + M::drow(lane, i), rv = r0 + 16 * u + n16;
          const T xc = pc[min(r, rows - 1)], xv = vc[i][min(rv, rows - 1)];
          c[u][i] = r < rows ? xc : (T)0;
          va[u][i] = rv < rows ? -xv : (T)0;
        }
    }
  };
  for (int r0 = rbeg; r0 < rows; r0 += STEP) {
    acc_t c0[RT];
    T v0[RT][4];
    load(r0, c0, v0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int u = 0; u < RT; ++u) c0[u] = M::mma(v0[u][i], tw[i], c0[u]);
    if (cok) {
      if (r0 + STEP <= rows) {
#pragma unroll
        for (int u = 0; u < RT; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) pc[r0 + 16 * u + M::drow(lane, i)] = c0[u][i];
      } else {
#pragma unroll
        for (int u = 0; u < RT; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = r0 + 16 * u + M::drow(lane, i);
            if (r < rows) pc[r] = c0[u][i];
          }
      }
    }
  }
}

// TRI: the panel is two stacked upper triangles [R_a; R_b] (rows = 2 l).  Column j of V then lives in row j and rows
// l .. l + j only (nothing else of a column is, or becomes, non-zero): a wave keeps the block's 16 top rows and the
// bottom triangle's rows (NS - 1 slices) of its column, and the MFMA phases run over rows [j0, j0 + 16) and
// [l, l + cend) instead of [j0, 2 l).
template <class T, int NS, bool TRI>
__device__ void hh_wy_factor_panel(T* P, int RP, int rows, int l, T* tau_out, T* t_out, T* Ts, T* Rs, T* tau_s) {
  typedef MT<T> M;
  typedef typename M::acc_t acc_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, n16 = lane & 15;
#ifdef CORRLA_HH_TIMING
  long long tacc[4] = {0, 0, 0, 0}, tlw[2] = {0, 0}, tprev = clock64();
#define HH_TICK(i) { const long long tn = clock64(); tacc[i] += tn - tprev; tprev = tn; }
#else
#define HH_TICK(i)
#endif
  for (int j0 = 0, pi = 0; j0 < l; j0 += kWyNb, ++pi) {
    const int nbk = min(kWyNb, l - j0), cend = j0 + nbk;
    // row held by (slice i, this lane); `rows` = none
    auto rowof = [&](int i) -> int {
      if constexpr (TRI) return i < NS - 1 ? l + lane + 64 * i : ((lane < kWyNb && j0 + lane < l) ? j0 + lane : rows);
      return lane + 64 * i;
    };
    // TRI: slices of the bottom triangle this block's reflectors can touch (uniform) -- its rows below l + cend are, and
    // stay, zero.  (Skipping the finished rows above the block in a dense panel measured slower: 193 -> 203 us.)
    auto live = [&](int i) -> bool {
      if constexpr (TRI) return i == NS - 1 || 64 * i < cend;
      return true;
    };
    HH_TICK(3)
    // (a) reflectors of the block
    T x[NS];
    {
      const T* mc = P + (size_t)min(j0 + wave, l - 1) * RP;
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        x[i] = (T)0;
        if (live(i)) {
          const int r = rowof(i);
          const T val = mc[min(r, rows - 1)];
          x[i] = r < rows ? val : (T)0;
        }
      }
    }
    for (int jj = 0; jj < nbk; ++jj) {
      const int j = j0 + jj;
      T* cj = P + (size_t)j * RP;
      if (wave == jj) {
        T s2 = (T)0, xs = x[0];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          const int r = rowof(i);
          if (live(i)) s2 += r > j ? x[i] * x[i] : (T)0;
          if (i > 0) xs = (TRI ? NS - 1 : (j >> 6)) == i ? x[i] : xs;
        }
        const T sigma = hh_wave_sum(s2);
        const T alpha = hh_readlane(xs, TRI ? jj : (j & 63));  // the diagonal entry: slice j / 64, lane j % 64 (TRI: top slice)
        T tau = (T)0, beta = alpha, scale = (T)0;
        // LAPACK xLARFG.  The hardware sqrt / rcp flush denormals: a column whose squared norm is below the smallest
        // normal number (entries below ~1e-19 in f32 -- where sigma itself underflows) counts as already reduced
        // (tau = 0, its sub-diagonal entries dropped) instead of producing 1 / 0.
        const T n2 = alpha * alpha + sigma;
        if (sigma > (T)0 && n2 >= std::numeric_limits<T>::min()) {
          beta = -copysign(hh_sqrt(n2), alpha);
          tau = (beta - alpha) * hh_rcp(beta);
          scale = hh_rcp(alpha - beta);
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          const int r = rowof(i);
          if (live(i) && r < rows) cj[r] = r > j ? x[i] * scale : (r == j ? beta : x[i]);
        }
        if (lane == 0) {
          tau_s[jj] = tau;
          tau_out[j] = tau;
        }
      }
      hh_lds_barrier();
      if (wave > jj && wave < nbk) {
        T vf[NS];
        T d = (T)0;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          vf[i] = (T)0;
          if (live(i)) {
            const int r = rowof(i);
            const T val = cj[min(r, rows - 1)];
            vf[i] = (r < rows && r > j) ? val : (r == j ? (T)1 : (T)0);
            d += vf[i] * x[i];
          }
        }
        const T w = tau_s[jj] * hh_wave_sum(d);
#pragma unroll
        for (int i = 0; i < NS; ++i)
          if (live(i)) x[i] -= w * vf[i];
      }
    }
    if (tid >= nbk && tid < kWyNb) tau_s[tid] = (T)0;
    hh_lds_barrier();
    HH_TICK(0)
    // clean V in place for the MFMA phases: the R entries of the 16 x 16 diagonal block are parked in Rs and come back
    // when the next block is parked (or after the last block); rows above j0 are never read by this block's products
    if (tid < kWyNb * kWyNb) {
      const int ra = tid & 15, cb = tid >> 4;
      if (ra <= cb) {
        if (j0 > 0) P[(size_t)(j0 - kWyNb + cb) * RP + (j0 - kWyNb + ra)] = Rs[cb * kWyNb + ra];
        if (cb < nbk) {
          T* e = P + (size_t)(j0 + cb) * RP + (j0 + ra);
          Rs[cb * kWyNb + ra] = *e;
          *e = ra == cb ? (T)1 : (T)0;
        }
      }
    }
    hh_lds_barrier();
    // (b) T of the block on the last wave (G = V^T V by one MFMA chain, then the xLARFT recurrence with lane i = row i),
    //     while (c1) the other waves take W = V^T A for their trailing tiles (at most 8 tiles: the last wave has none)
    const int ntile = (l - cend + 15) / 16;
    acc_t w = {0, 0, 0, 0};
    const int cc = cend + 16 * wave + n16;
    const bool cok = wave < ntile && cc < l;
    T* pc = P + (size_t)min(cc, l - 1) * RP;
    const T* vcol = P + (size_t)min(j0 + n16, cend - 1) * RP;  // this lane's column of V in the V^T products
    const int s1e = TRI ? min(j0 + kWyNb, l) : rows, s2e = l + cend;  // V is non-zero in rows [j0, s1e) (and [l, s2e))
    if (wave == kHhWaves - 1) {
      acc_t acc = wy_kdot<T, kWyPipe>(vcol, vcol, j0, s1e, g, acc_t{0, 0, 0, 0});
      if constexpr (TRI) acc = wy_kdot<T, kWyPipe>(vcol, vcol, l, s2e, g, acc);
#ifdef CORRLA_HH_TIMING
      const long long tg1 = clock64();
      tlw[0] += tg1 - tprev;  // on the last wave: the Gram chain
#endif
      // xLARFT recurrence, lane i = row i of T: T(0:j, j) = -tau_j T(0:j, 0:j) G(0:j, j); G(k, j) sits in D register
      // WyD::reg(k) of lane 16 WyD::grp(k) + j and is broadcast through a scalar register
      T trow[kWyNb];
#pragma unroll
      for (int j = 0; j < kWyNb; ++j) {
        T a2 = (T)0;
#pragma unroll
        for (int kx = 0; kx < j; ++kx) a2 += trow[kx] * hh_readlane(acc[WyD<T>::reg(kx)], 16 * WyD<T>::grp(kx) + j);
        const T tj = tau_s[j];
        trow[j] = lane == j ? tj : (lane < j ? -tj * a2 : (T)0);
      }
      if (lane < kWyNb) {
#pragma unroll
        for (int j = 0; j < kWyNb; ++j) {
          Ts[lane * kWyNb + j] = trow[j];
          t_out[(size_t)pi * kWyNb * kWyNb + lane * kWyNb + j] = trow[j];
        }
      }
#ifdef CORRLA_HH_TIMING
      tlw[1] += clock64() - tg1;  // on the last wave: the recurrence and the stores of T
#endif
    } else if (wave < ntile) {
      // (a tile column past l reads column l - 1: finite values whose results are never stored)
      w = wy_kdot<T, kWyPipe>(vcol, (const T*)pc, j0, s1e, g, w);
      if constexpr (TRI) w = wy_kdot<T, kWyPipe>(vcol, (const T*)pc, l, s2e, g, w);
    }
    hh_lds_barrier();
    HH_TICK(1)
    // (c2) TW = T^T W and A -= V TW
    if (wave < ntile) {
      acc_t tw = {0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < 4; ++i) tw = M::mma(Ts[M::drow(lane, i) * kWyNb + n16], w[i], tw);
      wy_update<T, kWyRowTiles>(pc, cok, (const T*)P, (int64_t)RP, j0, cend, j0, s1e, tw, lane);
      if constexpr (TRI) wy_update<T, kWyRowTiles>(pc, cok, (const T*)P, (int64_t)RP, j0, cend, l, s2e, tw, lane);
    }
    hh_lds_barrier();
    HH_TICK(2)
  }
  if (tid < kWyNb * kWyNb) {
    const int ra = tid & 15, cb = tid >> 4, jl = (hh_wy_panels(l) - 1) * kWyNb;
    if (ra <= cb && jl + cb < l) P[(size_t)(jl + cb) * RP + (jl + ra)] = Rs[cb * kWyNb + ra];
  }
  hh_lds_barrier();
#ifdef CORRLA_HH_TIMING
  if (blockIdx.x == 0 && tid == 64 * (kHhWaves - 1)) printf("  last wave: gram %lld  recurrence %lld\n", tlw[0], tlw[1]);
  if (blockIdx.x == 0 && tid == 0)
    printf("hh_wy_factor rows %d l %d: reflectors %lld  T+W %lld  update %lld  other %lld cycles\n", rows, l, tacc[0], tacc[1], tacc[2],
           tacc[3]);
#endif
}

// P <- H_0 ... H_{l-1} P with the blocks' V (global, leading dimension ldv) and T (global, 256 per block)
template <class T, bool TRI>
__device__ void hh_wy_apply_panel(T* P, int RP, int rows, int l, const T* __restrict__ V, int64_t ldv, const T* __restrict__ t_in) {
  typedef MT<T> M;
  typedef typename M::acc_t acc_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, n16 = lane & 15;
  const int ntile = (l + 15) / 16;
  for (int t = wave; t < ntile; t += kHhWaves) {
    const int cc = 16 * t + n16;
    const bool cok = cc < l;
    T* pc = P + (size_t)min(cc, l - 1) * RP;
    for (int pi = hh_wy_panels(l) - 1; pi >= 0; --pi) {
      const int j0 = pi * kWyNb, cend = min(j0 + kWyNb, l);
      const T* tp = t_in + (size_t)pi * kWyNb * kWyNb;
      const T* vcol = V + (int64_t)min(j0 + n16, cend - 1) * ldv;
      const int s1e = TRI ? min(j0 + kWyNb, l) : rows, s2e = l + cend;
      acc_t w = wy_kdot<T, kWyApplyPipe>(vcol, (const T*)pc, j0, s1e, g, acc_t{0, 0, 0, 0});
      if constexpr (TRI) w = wy_kdot<T, kWyApplyPipe>(vcol, (const T*)pc, l, s2e, g, w);
      acc_t tw = {0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < 4; ++i) tw = M::mma(tp[n16 * kWyNb + M::drow(lane, i)], w[i], tw);  // T W
      wy_update<T, kWyApplyRowTiles>(pc, cok, V, ldv, j0, cend, j0, s1e, tw, lane);
      if constexpr (TRI) wy_update<T, kWyApplyRowTiles>(pc, cok, V, ldv, j0, cend, l, s2e, tw, lane);
    }
  }
}

// ---- up sweep -----------------------------------------------------------------------------------------------------
// leaves: panel i = rows [row0(i), row0(i + 1)) of y (column-major, ld ldy).  Reflectors -> v (same layout as y),
// tau -> tau[i * l ..], R -> rout[i * l * l ..] (l x l column-major, zeros below the diagonal).
template <class T, bool WY>
__global__ __launch_bounds__(kHhThreads) void hh_leaf_factor_kernel(const T* __restrict__ y, int64_t ldy, int64_t m, int l,
                                                                    int nleaf, T* __restrict__ v, int64_t ldv, T* tau,
                                                                    T* rout, T* tbuf) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* P = (T*)smem + (WY ? kWyExtra : 0);
  const int node = blockIdx.x;
  const int64_t r0 = hh_leaf_row0(m, nleaf, node);
  const int rows = (int)(hh_leaf_row0(m, nleaf, node + 1) - r0);
  const int RP = hh_pitch(rows);
  for (int idx = threadIdx.x; idx < rows * l; idx += kHhThreads) {
    const int c = idx / rows, r = idx - c * rows;
    P[(size_t)c * RP + r] = y[(int64_t)c * ldy + r0 + r];
  }
  __syncthreads();
  if constexpr (WY)
    hh_wy_factor_panel<T, kHhMaxRowsPerLane, false>(P, RP, rows, l, tau + (size_t)node * l,
                                                    tbuf + (size_t)node * hh_wy_panels(l) * kWyNb * kWyNb, (T*)smem,
                                                    (T*)smem + kWyNb * kWyNb, (T*)smem + 2 * kWyNb * kWyNb);
  else
    hh_factor_panel<T, kHhMaxRowsPerLane, false>(P, RP, rows, l, tau + (size_t)node * l);
  for (int idx = threadIdx.x; idx < rows * l; idx += kHhThreads) {
    const int c = idx / rows, r = idx - c * rows;
    const T pv = P[(size_t)c * RP + r];
    v[(int64_t)c * ldv + r0 + r] = WY ? (r > c ? pv : (r == c ? (T)1 : (T)0)) : pv;  // blocked form: clean reflectors
  }
  T* ro = rout + (size_t)node * l * l;
  for (int idx = threadIdx.x; idx < l * l; idx += kHhThreads) {
    const int c = idx / l, r = idx - c * l;
    ro[idx] = (r <= c && r < rows) ? P[(size_t)c * RP + r] : (T)0;
  }
}

// inner level: node t combines rin[2 t] and rin[2 t + 1] (an unpaired last R is passed through, tau = 0).
// Reflectors -> v[t] (2 l x l, ld 2 l), tau -> tau[t * l ..], R -> rout[t].
template <class T, bool WY>
__global__ __launch_bounds__(kHhThreads) void hh_tree_factor_kernel(const T* __restrict__ rin, int n_in, int l, T* __restrict__ v,
                                                                    T* tau, T* rout, T* tbuf) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* P = (T*)smem + (WY ? kWyExtra : 0);
  const int node = blockIdx.x;
  const int a = 2 * node, b = 2 * node + 1;
  T* ro = rout + (size_t)node * l * l;
  if (b >= n_in) {
    for (int idx = threadIdx.x; idx < l * l; idx += kHhThreads) ro[idx] = rin[(size_t)a * l * l + idx];
    for (int idx = threadIdx.x; idx < l; idx += kHhThreads) tau[(size_t)node * l + idx] = (T)0;
    return;
  }
  const int rows = 2 * l, RP = hh_pitch(rows);
  for (int idx = threadIdx.x; idx < l * l; idx += kHhThreads) {
    const int c = idx / l, r = idx - c * l;
    P[(size_t)c * RP + r] = rin[(size_t)a * l * l + idx];
    P[(size_t)c * RP + l + r] = rin[(size_t)b * l * l + idx];
  }
  __syncthreads();
  if constexpr (WY)
    hh_wy_factor_panel<T, kHhTriSlices + 1, true>(P, RP, rows, l, tau + (size_t)node * l,
                                                  tbuf + (size_t)node * hh_wy_panels(l) * kWyNb * kWyNb, (T*)smem,
                                                  (T*)smem + kWyNb * kWyNb, (T*)smem + 2 * kWyNb * kWyNb);
  else
    hh_factor_panel<T, kHhTriSlices, true>(P, RP, rows, l, tau + (size_t)node * l);
  T* vo = v + (size_t)node * rows * l;
  for (int idx = threadIdx.x; idx < rows * l; idx += kHhThreads) {
    const int c = idx / rows, r = idx - c * rows;
    const T pv = P[(size_t)c * RP + r];
    vo[idx] = WY ? (r > c ? pv : (r == c ? (T)1 : (T)0)) : pv;
  }
  for (int idx = threadIdx.x; idx < l * l; idx += kHhThreads) {
    const int c = idx / l, r = idx - c * l;
    ro[idx] = r <= c ? P[(size_t)c * RP + r] : (T)0;
  }
}

// ---- down sweep ---------------------------------------------------------------------------------------------------
// inner level: node t turns its coefficient block cin[t] (l x l; the identity when cin == nullptr: the root) into
// the blocks of its children cout[2 t], cout[2 t + 1].
template <class T, bool WY>
__global__ __launch_bounds__(kHhThreads) void hh_tree_apply_kernel(const T* __restrict__ cin, const T* __restrict__ v,
                                                                   const T* __restrict__ tau, int n_children, int l, T* cout,
                                                                   const T* __restrict__ tbuf) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* P = (T*)smem;
  const int node = blockIdx.x;
  const int a = 2 * node, b = 2 * node + 1;
  if (b >= n_children) {
    for (int idx = threadIdx.x; idx < l * l; idx += kHhThreads) {
      const int c = idx / l, r = idx - c * l;
      cout[(size_t)a * l * l + idx] = cin ? cin[(size_t)node * l * l + idx] : (r == c ? (T)1 : (T)0);
    }
    return;
  }
  const int rows = 2 * l, RP = hh_pitch(rows);
  for (int idx = threadIdx.x; idx < l * l; idx += kHhThreads) {
    const int c = idx / l, r = idx - c * l;
    P[(size_t)c * RP + r] = cin ? cin[(size_t)node * l * l + idx] : (r == c ? (T)1 : (T)0);
    P[(size_t)c * RP + l + r] = (T)0;
  }
  __syncthreads();
  if constexpr (WY)
    hh_wy_apply_panel<T, true>(P, RP, rows, l, v + (size_t)node * rows * l, (int64_t)rows,
                         tbuf + (size_t)node * hh_wy_panels(l) * kWyNb * kWyNb);
  else
    hh_apply_panel<T, kHhTriSlices, true>(P, RP, rows, l, v + (size_t)node * rows * l, (int64_t)rows, tau + (size_t)node * l);
  __syncthreads();
  for (int idx = threadIdx.x; idx < l * l; idx += kHhThreads) {
    const int c = idx / l, r = idx - c * l;
    cout[(size_t)a * l * l + idx] = P[(size_t)c * RP + r];
    cout[(size_t)b * l * l + idx] = P[(size_t)c * RP + l + r];
  }
}

// leaves: rows of Q = H_0 ... H_{l-1} [cin[i]; 0]  (cin == nullptr: a single leaf, coefficient block = identity)
template <class T, bool WY>
__global__ __launch_bounds__(kHhThreads) void hh_leaf_apply_kernel(const T* __restrict__ cin, const T* __restrict__ v, int64_t ldv,
                                                                   const T* __restrict__ tau, int64_t m, int l, int nleaf,
                                                                   T* __restrict__ q, int64_t ldq, const T* __restrict__ tbuf) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* P = (T*)smem;
  const int node = blockIdx.x;
  const int64_t r0 = hh_leaf_row0(m, nleaf, node);
  const int rows = (int)(hh_leaf_row0(m, nleaf, node + 1) - r0);
  const int RP = hh_pitch(rows);
  for (int idx = threadIdx.x; idx < rows * l; idx += kHhThreads) {
    const int c = idx / rows, r = idx - c * rows;
    T val = (T)0;
    if (r < l) val = cin ? cin[(size_t)node * l * l + (size_t)c * l + r] : (r == c ? (T)1 : (T)0);
    P[(size_t)c * RP + r] = val;
  }
  __syncthreads();
  if constexpr (WY)
    hh_wy_apply_panel<T, false>(P, RP, rows, l, v + r0, ldv, tbuf + (size_t)node * hh_wy_panels(l) * kWyNb * kWyNb);
  else
    hh_apply_panel<T, kHhMaxRowsPerLane, false>(P, RP, rows, l, v + r0, ldv, tau + (size_t)node * l);
  __syncthreads();
  for (int idx = threadIdx.x; idx < rows * l; idx += kHhThreads) {
    const int c = idx / rows, r = idx - c * rows;
    q[(int64_t)c * ldq + r0 + r] = P[(size_t)c * RP + r];
  }
}

}  // namespace k
}  // namespace corrla
