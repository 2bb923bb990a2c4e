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
// A Householder thin-Q is orthonormal whatever the rank of the sketch (tau = 0 for a column that is already reduced),
// which is exactly the behaviour of the reference on rank-deficient inputs.  One panel (2 l x l) must fit in the
// 160 KB LDS of a CU: l <= 138 (f32) / l <= 97 (f64).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

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
    __syncthreads();  // everyone has read column j
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
    __syncthreads();  // column j + 1 is complete before the next reflector is formed
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

// ---- up sweep -----------------------------------------------------------------------------------------------------
// leaves: panel i = rows [row0(i), row0(i + 1)) of y (column-major, ld ldy).  Reflectors -> v (same layout as y),
// tau -> tau[i * l ..], R -> rout[i * l * l ..] (l x l column-major, zeros below the diagonal).
template <class T>
__global__ __launch_bounds__(kHhThreads) void hh_leaf_factor_kernel(const T* __restrict__ y, int64_t ldy, int64_t m, int l,
                                                                    int nleaf, T* __restrict__ v, int64_t ldv, T* tau,
                                                                    T* rout) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* P = (T*)smem;
  const int node = blockIdx.x;
  const int64_t r0 = hh_leaf_row0(m, nleaf, node);
  const int rows = (int)(hh_leaf_row0(m, nleaf, node + 1) - r0);
  const int RP = hh_pitch(rows);
  for (int idx = threadIdx.x; idx < rows * l; idx += kHhThreads) {
    const int c = idx / rows, r = idx - c * rows;
    P[(size_t)c * RP + r] = y[(int64_t)c * ldy + r0 + r];
  }
  __syncthreads();
  hh_factor_panel<T, kHhMaxRowsPerLane, false>(P, RP, rows, l, tau + (size_t)node * l);
  for (int idx = threadIdx.x; idx < rows * l; idx += kHhThreads) {
    const int c = idx / rows, r = idx - c * rows;
    v[(int64_t)c * ldv + r0 + r] = P[(size_t)c * RP + r];
  }
  T* ro = rout + (size_t)node * l * l;
  for (int idx = threadIdx.x; idx < l * l; idx += kHhThreads) {
    const int c = idx / l, r = idx - c * l;
    ro[idx] = (r <= c && r < rows) ? P[(size_t)c * RP + r] : (T)0;
  }
}

// inner level: node t combines rin[2 t] and rin[2 t + 1] (an unpaired last R is passed through, tau = 0).
// Reflectors -> v[t] (2 l x l, ld 2 l), tau -> tau[t * l ..], R -> rout[t].
template <class T>
__global__ __launch_bounds__(kHhThreads) void hh_tree_factor_kernel(const T* __restrict__ rin, int n_in, int l, T* __restrict__ v,
                                                                    T* tau, T* rout) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* P = (T*)smem;
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
  hh_factor_panel<T, kHhTriSlices, true>(P, RP, rows, l, tau + (size_t)node * l);
  T* vo = v + (size_t)node * rows * l;
  for (int idx = threadIdx.x; idx < rows * l; idx += kHhThreads) {
    const int c = idx / rows, r = idx - c * rows;
    vo[idx] = P[(size_t)c * RP + r];
  }
  for (int idx = threadIdx.x; idx < l * l; idx += kHhThreads) {
    const int c = idx / l, r = idx - c * l;
    ro[idx] = r <= c ? P[(size_t)c * RP + r] : (T)0;
  }
}

// ---- down sweep ---------------------------------------------------------------------------------------------------
// inner level: node t turns its coefficient block cin[t] (l x l; the identity when cin == nullptr: the root) into
// the blocks of its children cout[2 t], cout[2 t + 1].
template <class T>
__global__ __launch_bounds__(kHhThreads) void hh_tree_apply_kernel(const T* __restrict__ cin, const T* __restrict__ v,
                                                                   const T* __restrict__ tau, int n_children, int l, T* cout) {
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
  hh_apply_panel<T, kHhTriSlices, true>(P, RP, rows, l, v + (size_t)node * rows * l, (int64_t)rows, tau + (size_t)node * l);
  __syncthreads();
  for (int idx = threadIdx.x; idx < l * l; idx += kHhThreads) {
    const int c = idx / l, r = idx - c * l;
    cout[(size_t)a * l * l + idx] = P[(size_t)c * RP + r];
    cout[(size_t)b * l * l + idx] = P[(size_t)c * RP + l + r];
  }
}

// leaves: rows of Q = H_0 ... H_{l-1} [cin[i]; 0]  (cin == nullptr: a single leaf, coefficient block = identity)
template <class T>
__global__ __launch_bounds__(kHhThreads) void hh_leaf_apply_kernel(const T* __restrict__ cin, const T* __restrict__ v, int64_t ldv,
                                                                   const T* __restrict__ tau, int64_t m, int l, int nleaf,
                                                                   T* __restrict__ q, int64_t ldq) {
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
  hh_apply_panel<T, kHhMaxRowsPerLane, false>(P, RP, rows, l, v + r0, ldv, tau + (size_t)node * l);
  __syncthreads();
  for (int idx = threadIdx.x; idx < rows * l; idx += kHhThreads) {
    const int c = idx / rows, r = idx - c * rows;
    q[(int64_t)c * ldq + r0 + r] = P[(size_t)c * RP + r];
  }
}

}  // namespace k
}  // namespace corrla
