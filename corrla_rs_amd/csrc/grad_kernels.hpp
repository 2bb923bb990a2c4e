// Active-subspace gradient stage on the device, f64 like the reference:
//   PolyGradientEstimator::{nearest_points, est_grad_lin, est_grad_quad}   src/lib_math_utils/active_subspaces.rs:66-141
//   linear_fit / quad_fit / build_vandermonde / jac_from_lin / jac_from_quad src/lib_math_utils/stats_corr.rs:110-249
//   ActiveSsRsvd::create_grad_mat                                           src/lib_math_utils/active_subspaces.rs:215-229
// The reference walks a kd-tree per sample and takes an SVD-based pseudo-inverse per sample, serially.  Here:
//   knn_kernel      : exact n nearest neighbours by squared Euclidean distance, brute force (in k = 64 dimensions a
//                     kd-tree degenerates to that anyway): 16 queries per workgroup share every LDS-staged chunk of
//                     64 support points; each wave keeps a sorted top-n list per query in LDS (ties -> lower index);
//   grad_fit_kernel : one wave per query: gathers the neighbours, forms the normal equations of the reference's
//                     design matrix ([x, 1] for order 1, [x, x_a x_b (a <= b), 1] for order 2: linear_fit and
//                     build_vandermonde, stats_corr.rs:146-159, 198-207) in LDS, Cholesky-solves them and writes the
//                     gradient.  Both polynomial spaces are translation invariant, so the fit is formed in the
//                     coordinates x - x0 (well-conditioned normal equations; the same fitted polynomial whenever
//                     the design has full column rank) and the gradient at x0 is just the linear coefficients
//                     (order 2: the reference differentiates its quadratic by forward differences with eps = 1e-10,
//                     which agree with the analytic gradient to ~1e-6 relative).
// The gradient matrix is written in the reference's k x N column-major layout (N rows of k contiguous values), which
// is the row-major tall matrix the RSVD kernels take directly.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

namespace corrla {
namespace k {

constexpr int kGradMaxDim = 64;    // features k
constexpr int kGradMaxNbr = 512;   // neighbours per query (register arrays of the list insertion; LDS is checked per call)
// design-matrix columns: k + 1 (order 1), k + k (k + 1) / 2 + 1 (order 2); no fixed cap: the neighbours of a query live in
// LDS (grad_fit_lds_bytes(k, n_nbrs, order, false) <= 160 KiB), its normal equations next to them when they fit (order 2
// up to k = 14) and in a per-workgroup slice of global memory otherwise (order 2 up to k = 30, where n_nbrs <= 512 binds)
constexpr int kKnnQueriesPerWave = 4, kKnnWaves = 4, kKnnQueries = kKnnQueriesPerWave * kKnnWaves;

// xt (k x ldt, dimension-major) <- x (n x k, row-major)
__global__ void grad_transpose_kernel(const double* __restrict__ x, int64_t n, int k, double* xt, int64_t ldt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int d = 0; d < k; ++d) xt[(int64_t)d * ldt + i] = x[i * k + d];
}

// nbr[q][0..n_nbrs) = indices of the n_nbrs nearest support points of query q, nearest first
__global__ __launch_bounds__(64 * kKnnWaves) void knn_kernel(const double* __restrict__ xt, int64_t ldt, int64_t n_pts, int k,
                                                             const double* __restrict__ xq, int64_t n_q, int n_nbrs,
                                                             int* __restrict__ nbr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* pts = (double*)smem;                                    // [k][64]
  double* qv = pts + (size_t)k * 64;                              // [kKnnQueries][k]
  double* ld = qv + (size_t)kKnnQueries * k;                      // [kKnnQueries][n_nbrs] sorted distances
  int* li = (int*)(ld + (size_t)kKnnQueries * n_nbrs);            // [kKnnQueries][n_nbrs] indices
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t q0 = (int64_t)blockIdx.x * kKnnQueries;
  for (int idx = tid; idx < kKnnQueries * k; idx += blockDim.x) {
    const int qq = idx / k, d = idx - qq * k;
    qv[idx] = (q0 + qq < n_q) ? xq[(q0 + qq) * k + d] : 0.0;
  }
  for (int idx = tid; idx < kKnnQueries * n_nbrs; idx += blockDim.x) {
    ld[idx] = __builtin_huge_val();
    li[idx] = -1;
  }
  const int64_t nchunks = (n_pts + 63) / 64;
  for (int64_t c = 0; c < nchunks; ++c) {
    __syncthreads();  // the previous chunk has been consumed (and the initialisation above is visible)
    const int64_t base = c * 64;
    for (int idx = tid; idx < k * 64; idx += blockDim.x) {
      const int d = idx >> 6, j = idx & 63;
      pts[idx] = (base + j < n_pts) ? xt[(int64_t)d * ldt + base + j] : 0.0;
    }
    __syncthreads();
    const bool valid = base + lane < n_pts;
#pragma unroll
    for (int qi = 0; qi < kKnnQueriesPerWave; ++qi) {
      const int qq = wave * kKnnQueriesPerWave + qi;
      if (q0 + qq >= n_q) continue;  // uniform per wave
      const double* qp = qv + (size_t)qq * k;
      double dist = 0.0;
      for (int d = 0; d < k; ++d) {
        const double df = pts[d * 64 + lane] - qp[d];
        dist += df * df;
      }
      double* qd = ld + (size_t)qq * n_nbrs;
      int* qix = li + (size_t)qq * n_nbrs;
      double tau = qd[n_nbrs - 1];
      unsigned long long mask = __ballot(valid && dist < tau);
      while (mask) {  // uniform loop: candidates in increasing index order
        const int b = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        const double cd = __shfl(dist, b, 64);
        if (!(cd < tau)) continue;
        // insertion position = number of entries <= cd (equal distances keep the lower index first)
        int pos = 0;
        for (int e0 = 0; e0 < n_nbrs; e0 += 64) {
          const int e = e0 + lane;
          pos += __popcll(__ballot(e < n_nbrs && qd[e] <= cd));
        }
        // shift [pos, n - 1) up by one: every read is issued before any write (in-order LDS queue of the wave)
        double sd[(kGradMaxNbr + 63) / 64];
        int si[(kGradMaxNbr + 63) / 64];
#pragma unroll
        for (int s = 0; s < (kGradMaxNbr + 63) / 64; ++s) {
          const int e = s * 64 + lane;
          if (e < n_nbrs && e > pos) {
            sd[s] = qd[e - 1];
            si[s] = qix[e - 1];
          }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < (kGradMaxNbr + 63) / 64; ++s) {
          const int e = s * 64 + lane;
          if (e < n_nbrs && e > pos) {
            qd[e] = sd[s];
            qix[e] = si[s];
          }
        }
        if (lane == 0) {
          qd[pos] = cd;
          qix[pos] = (int)(base + b);
        }
        __builtin_amdgcn_wave_barrier();
        tau = qd[n_nbrs - 1];
      }
    }
  }
  __syncthreads();
  for (int idx = tid; idx < kKnnQueries * n_nbrs; idx += blockDim.x) {
    const int qq = idx / n_nbrs;
    if (q0 + qq < n_q) nbr[(q0 + qq) * n_nbrs + (idx - qq * n_nbrs)] = li[idx];
  }
}
inline size_t knn_lds_bytes(int k, int n_nbrs) {
  return (size_t)k * 64 * 8 + (size_t)kKnnQueries * k * 8 + (size_t)kKnnQueries * n_nbrs * 12 + 64;
}

// ---- k-NN with the distance tile on the f64 MFMA ---------------------------------------------------------
// d^2(q, p) = |q|^2 + |p|^2 - 2 q.p: each wave owns 16 queries whose MFMA A-fragments (16 x 4 slices of the query
// block, f32) stay in registers for the whole scan; per chunk of 64 points it issues 4 x k/4 v_mfma_f32_16x16x4_f32
// against an f32 copy of the LDS-staged points (row pitch 80: the four dimension rows of a fragment read fall in
// disjoint banks); the norms are f64.  The f32 matrix unit runs at twice the f64 rate on MI355X and the MFMA value is
// only a FILTER (margin 1e-5 (|q|^2 + |p|^2) covers the f32 rounding); a candidate that passes gets its exact
// distance sum_d (p_d - q_d)^2 recomputed across the lanes before it may enter the sorted list, so the neighbour sets
// and their order are those of the exact search.
typedef float knn_f32x4 __attribute__((ext_vector_type(4)));
constexpr int kKnnPitch = 80;
__device__ __forceinline__ double wave_max_f64(double x) {
  auto dpp = [](double v, auto ctrl) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, decltype(ctrl)::value, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, decltype(ctrl)::value, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
  };
  x = fmax(x, dpp(x, std::integral_constant<int, 0xB1>{}));
  x = fmax(x, dpp(x, std::integral_constant<int, 0x4E>{}));
  x = fmax(x, dpp(x, std::integral_constant<int, 0x141>{}));
  x = fmax(x, dpp(x, std::integral_constant<int, 0x140>{}));
  x = fmax(x, __shfl_xor(x, 16, 64));
  x = fmax(x, __shfl_xor(x, 32, 64));
  return x;
}
// sum over the 64 lanes, result in every lane: DPP inside the 16-lane rows, two cross-row exchanges
__device__ __forceinline__ double wave_sum_f64(double x) {
  auto dpp = [](double v, auto ctrl) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, decltype(ctrl)::value, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, decltype(ctrl)::value, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
  };
  x += dpp(x, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
  x += dpp(x, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
  x += dpp(x, std::integral_constant<int, 0x141>{});  // row_half_mirror
  x += dpp(x, std::integral_constant<int, 0x140>{});  // row_mirror
  x += __shfl_xor(x, 16, 64);
  x += __shfl_xor(x, 32, 64);
  return x;
}
// NKS = number of 4-dimension MFMA slices compiled in (4, 8 or 16: k <= 16, 32, 64; unused slices multiply zeros).
// A run-time slice count inside the unrolled loop made the compiler shuttle the sixteen accumulator registers between
// AccVGPRs and VGPRs around every slice (~100 moves per slice, 6.7x the MFMA issue time of a chunk).
template <int W, int NKS>
__global__ __launch_bounds__(64 * W) void knn_mfma_kernel(const double* __restrict__ xt, int64_t ldt,
                                                          const double* __restrict__ pnorm, int64_t n_pts, int k,
                                                          const double* __restrict__ xq, int64_t n_q, int n_nbrs,
                                                          int* __restrict__ nbr) {
  constexpr int QT = 16 * W;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int k4 = (k + 3) & ~3;
  constexpr int KD = 4 * NKS;                       // staged dimensions (rows k .. KD - 1 are zero)
  double* qv = (double*)smem;                       // [QT][k4]
  double* qn = qv + (size_t)QT * k4;                // [QT]
  double* pts = qn + QT;                            // [KD][kKnnPitch]  (exact re-check)
  float* ptsf = (float*)(pts + (size_t)KD * kKnnPitch);   // [KD][kKnnPitch]  f32 copy: the MFMA filter's B operand
  double* pn = (double*)(ptsf + (size_t)KD * kKnnPitch);  // [64]
  double* ld = pn + 64;                             // [QT][n_nbrs]
  int* li = (int*)(ld + (size_t)QT * n_nbrs);       // [QT][n_nbrs]
  double* qmax = (double*)(li + (size_t)QT * n_nbrs + ((QT * n_nbrs) & 1));  // [QT] current n-th distance
  int* qmaxpos = (int*)(qmax + QT);                 // [QT] its slot in the (unsorted) list
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t q0 = (int64_t)blockIdx.x * QT;
  for (int idx = tid; idx < QT * k4; idx += blockDim.x) {
    const int qq = idx / k4, d = idx - qq * k4;
    qv[idx] = (q0 + qq < n_q && d < k) ? xq[(q0 + qq) * k + d] : 0.0;
  }
  for (int idx = tid; idx < QT * n_nbrs; idx += blockDim.x) {
    ld[idx] = __builtin_huge_val();
    li[idx] = -1;
  }
  if (tid < QT) {
    qmax[tid] = __builtin_huge_val();
    qmaxpos[tid] = n_nbrs - 1;
  }
  __syncthreads();
  if (tid < QT) {
    double s = 0.0;
    for (int d = 0; d < k4; ++d) s += qv[tid * k4 + d] * qv[tid * k4 + d];
    qn[tid] = s;
  }
  __syncthreads();
  // A fragments of this wave's 16 queries: lane (i = lane & 15, kk = lane >> 4) holds Q[i][4 ks + kk], in f32
  float afr[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int d = 4 * ks + (lane >> 4);
    afr[ks] = d < k4 ? (float)qv[(wave * 16 + (lane & 15)) * k4 + d] : 0.0f;
  }
  // D layout of v_mfma_f32_16x16x4_f32: column (point) = lane & 15, row (query) = 4 (lane >> 4) + r
  double qn_r[4], tau_r[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    qn_r[r] = qn[wave * 16 + 4 * (lane >> 4) + r];
    tau_r[r] = __builtin_huge_val();
  }
  const int64_t nchunks = (n_pts + 63) / 64;
  // the next chunk of points travels global -> registers while the MFMAs work on the current one
  constexpr int PER = 64 / W;  // (64 dims x 64 points) / (64 W threads)
  double pre[PER];
  double pre_n = 0.0;
  auto fetch = [&](int64_t cc) {
    const int64_t b2 = cc * 64;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = tid + i * 64 * W;
      const int d = idx >> 6, j = idx & 63;
      pre[i] = (d < k && b2 + j < n_pts) ? xt[(int64_t)d * ldt + b2 + j] : 0.0;
    }
    if (tid < 64) pre_n = b2 + tid < n_pts ? pnorm[b2 + tid] : 0.0;
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = tid + i * 64 * W;
      const int d = idx >> 6, j = idx & 63;
      if (d < KD) {
        pts[d * kKnnPitch + j] = pre[i];
        ptsf[d * kKnnPitch + j] = (float)pre[i];
      }
    }
    if (tid < 64) pn[tid] = pre_n;
  };
  if (nchunks > 0) fetch(0);
  for (int64_t c = 0; c < nchunks; ++c) {
    __syncthreads();  // everyone is done with the previous chunk
    const int64_t base = c * 64;
    stash();
    __syncthreads();
    if (c + 1 < nchunks) fetch(c + 1);
    knn_f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (knn_f32x4){0, 0, 0, 0};
    const float* bp = ptsf + (lane >> 4) * kKnnPitch + (lane & 15);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[ks], bp[(4 * ks) * kKnnPitch + 16 * t], acc[t], 0, 0, 0);
    }
    unsigned hits = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int pl = 16 * t + (lane & 15);
      const double pnv = pn[pl];
      const bool pvalid = base + pl < n_pts;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double nsum = qn_r[r] + pnv;
        const double dm = nsum - 2.0 * (double)acc[t][r];
        // f32 products and sums over <= 64 dimensions: |error of 2 q.p| < 4e-6 (|q|^2 + |p|^2); the negated comparison
        // lets a non-finite value (f32 overflow) through to the exact re-check as well
        if (pvalid && !(dm - 1e-5 * nsum >= tau_r[r])) hits |= 1u << (4 * t + r);
      }
    }
    if (__any(hits != 0)) {
      // slow path (rare once the lists have warmed up).  Point tile t outer, lanes ascending: the candidates of one
      // query arrive in increasing point index, so equal distances keep the lower index first.
#pragma unroll 1
      for (int t = 0; t < 4; ++t) {
        unsigned long long mask = __ballot(((hits >> (4 * t)) & 0xFu) != 0);
        while (mask) {
          const int b = __ffsll((long long)mask) - 1;
          mask &= mask - 1;
          unsigned rb = ((unsigned)__shfl((int)hits, b, 64) >> (4 * t)) & 0xFu;
          const int pl = 16 * t + (b & 15);
          while (rb) {
            const int r = __ffs((int)rb) - 1;
            rb &= rb - 1;
            const int qq = wave * 16 + 4 * (b >> 4) + r;
            if (q0 + qq >= n_q) continue;  // padding query rows
            double df = 0.0;
            if (lane < k) df = pts[lane * kKnnPitch + pl] - qv[qq * k4 + lane];
            const double cd = wave_sum_f64(df * df);
            double* qd = ld + (size_t)qq * n_nbrs;
            int* qix = li + (size_t)qq * n_nbrs;
            if (!(cd < qmax[qq])) continue;  // equal distance: the earlier (lower) index stays
            // unsorted list: the candidate replaces the current n-th entry, then the new maximum is located
            // (largest distance, among equal ones the largest index)
            const int slot = qmaxpos[qq];
            if (lane == 0) {
              qd[slot] = cd;
              qix[slot] = (int)(base + pl);
            }
            __builtin_amdgcn_wave_barrier();
            double mv = -1.0;
            int mi = -2, mp = 0;
            for (int e = lane; e < n_nbrs; e += 64) {
              const double v = qd[e];
              const int ix = qix[e];
              if (v > mv || (v == mv && ix > mi)) {
                mv = v;
                mi = ix;
                mp = e;
              }
            }
            const double wmax = wave_max_f64(mv);
            unsigned long long who = __ballot(mv == wmax);
            if (__popcll(who) > 1) {  // several lanes hold the maximum value (duplicates, empty slots): largest index
              int wi = (mv == wmax) ? mi : -2;
              for (int off = 32; off > 0; off >>= 1) wi = max(wi, __shfl_xor(wi, off, 64));
              who = __ballot(mv == wmax && mi == wi);
            }
            const int wl = __ffsll((long long)who) - 1;
            const int wp = __shfl(mp, wl, 64);
            if (lane == 0) {
              qmax[qq] = wmax;
              qmaxpos[qq] = wp;
            }
            __builtin_amdgcn_wave_barrier();
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) tau_r[r] = qmax[wave * 16 + 4 * (lane >> 4) + r];
    }
  }
  __syncthreads();
  // nearest first, equal distances by index: rank of every entry inside its list
  for (int idx = tid; idx < QT * n_nbrs; idx += blockDim.x) {
    const int qq = idx / n_nbrs, e = idx - qq * n_nbrs;
    if (q0 + qq >= n_q) continue;
    const double* qd = ld + (size_t)qq * n_nbrs;
    const int* qix = li + (size_t)qq * n_nbrs;
    const double v = qd[e];
    const int ix = qix[e];
    int rank = 0;
    for (int f = 0; f < n_nbrs; ++f) rank += (qd[f] < v || (qd[f] == v && qix[f] < ix)) ? 1 : 0;
    nbr[(q0 + qq) * n_nbrs + rank] = ix;
  }
}
inline int knn_mfma_slices(int k) { return k <= 16 ? 4 : (k <= 32 ? 8 : 16); }
inline size_t knn_mfma_lds_bytes(int k, int n_nbrs, int waves) {
  const int k4 = (k + 3) & ~3, qt = 16 * waves, kd = 4 * knn_mfma_slices(k);
  return ((size_t)qt * k4 + qt + (size_t)kd * kKnnPitch + 64 + (size_t)qt * n_nbrs + qt) * 8 +
         ((size_t)kd * kKnnPitch + (size_t)qt * n_nbrs + 1 + qt) * 4 + 64;
}
// |p|^2 of every support point (from the dimension-major copy)
__global__ void point_norms_kernel(const double* __restrict__ xt, int64_t ldt, int64_t n, int k, double* pn) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int d = 0; d < k; ++d) s += xt[(int64_t)d * ldt + i] * xt[(int64_t)d * ldt + i];
  pn[i] = s;
}

// g[q * ldg + m] = out_scale * d(fit)/dx_m at query q.  status[q]: 0 ok, 1 ridge-regularised (rank-deficient design).
__global__ __launch_bounds__(64) void grad_fit_kernel(const double* __restrict__ x, const double* __restrict__ y, int k,
                                                      const double* __restrict__ xq, int64_t n_q,
                                                      const int* __restrict__ nbr, int n_nbrs, int order, double out_scale,
                                                      double* g, int64_t ldg, int* status, double* m_glob = nullptr,
                                                      int64_t m_stride = 0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int P = order == 1 ? k + 1 : k + k * (k + 1) / 2 + 1;  // design columns (the constant is the last one)
  const int LM = (P + 1) | 1;                 // row pitch of M (the right-hand side is column P): ODD, so that the 64 rows
                                              // the lanes of the factorisation walk in step fall in different banks
  double* xn = (double*)smem;                 // [n_nbrs][k] neighbour coordinates minus x0
  double* yn = xn + (size_t)n_nbrs * k;       // [n_nbrs]
  // M: [P][LM] normal equations, lower triangle -> Cholesky factor.  In LDS when it fits next to the neighbours; quadratic
  // fits of more than 14 features (231 columns at k = 20: 430 KB) keep it in a per-workgroup slice of global memory
  // (m_glob; L2-resident) and a bounded grid walks the queries -- slower per query, but a fit instead of CORRLA_EINVAL
  double* M = m_glob ? m_glob + (int64_t)blockIdx.x * m_stride : yn + n_nbrs;
  double* x0 = yn + n_nbrs + (m_glob ? (size_t)0 : (size_t)P * LM);  // [k]
  double* beta = x0 + k;                      // [P]
  double* dinv = beta + P;                    // [P] 1 / L(i, i)
  int* pa = (int*)(dinv + P);                 // [P] column -> (a, b); b = -1: linear term a; a = -1: constant
  int* pb = pa + P;
  int* flag = pb + P;
  const int lane = threadIdx.x;
  for (int64_t q = blockIdx.x; q < n_q; q += gridDim.x) {  // (one query per workgroup unless M lives in global memory)
  for (int d = lane; d < k; d += 64) x0[d] = xq[q * k + d];
  for (int c = lane; c < P; c += 64) {
    if (c < k) {
      pa[c] = c;
      pb[c] = -1;
    } else if (c == P - 1) {
      pa[c] = -1;
      pb[c] = -1;
    }
  }
  if (order == 2 && lane == 0) {  // mat_col_interactions order: a-major, b >= a (stats_corr.rs:112-143)
    int c = k;
    for (int a = 0; a < k; ++a)
      for (int b = a; b < k; ++b) {
        pa[c] = a;
        pb[c] = b;
        ++c;
      }
  }
  if (lane == 0) *flag = 0;
  __syncthreads();
  // gather: the neighbour list first (one coalesced load), then the rows with eight independent loads in flight
  // (a load that depends on a freshly loaded index per element would serialise two global latencies 80 times)
  int* nidx = (int*)(flag + 2);  // [n_nbrs]
  for (int r = lane; r < n_nbrs; r += 64) nidx[r] = nbr[q * n_nbrs + r];
  __syncthreads();
  const int total = n_nbrs * k;
  for (int i0g = 0; i0g < total; i0g += 64 * 8) {
    double val[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = i0g + u * 64 + lane;
      val[u] = 0.0;
      if (idx < total) {
        const int r = idx / k, d = idx - r * k;
        val[u] = x[(int64_t)nidx[r] * k + d] - x0[d];
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = i0g + u * 64 + lane;
      if (idx < total) xn[idx] = val[u];
    }
  }
  for (int r = lane; r < n_nbrs; r += 64) yn[r] = y[nidx[r]];
  __syncthreads();
  // sum_r v(r, i) v(r, j)  (j == P: the right-hand side sum_r v(r, i) y_r).  The column descriptors are hoisted out
  // of the loop over the neighbours, so its LDS reads are independent of one another and pipeline.
  auto gram_entry = [&](int i, int j) -> double {
    const int ai = pa[i], bi = pb[i];
    const int aj = j < P ? pa[j] : -2, bj = j < P ? pb[j] : -1;
    double s0 = 0.0, s1 = 0.0;
    int r = 0;
    auto term = [&](int rr) -> double {
      const double* row = xn + rr * k;
      double vi = ai < 0 ? 1.0 : row[ai];
      if (bi >= 0) vi *= row[bi];
      double vj = aj == -2 ? yn[rr] : (aj < 0 ? 1.0 : row[aj]);
      if (bj >= 0) vj *= row[bj];
      return vi * vj;
    };
    for (; r + 1 < n_nbrs; r += 2) {
      s0 += term(r);
      s1 += term(r + 1);
    }
    if (r < n_nbrs) s0 += term(r);
    return s0 + s1;
  };
  // normal equations: M(i, j) = sum_r v(r, i) v(r, j) for j <= i, M(i, P) = sum_r v(r, i) y_r
  const int npair = P * (P + 1) / 2 + P;
  for (int e = lane; e < npair; e += 64) {
    int i, j;
    if (e < P * (P + 1) / 2) {
      i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
      while (i * (i + 1) / 2 > e) --i;
      while ((i + 1) * (i + 2) / 2 <= e) ++i;
      j = e - i * (i + 1) / 2;
    } else {
      i = e - P * (P + 1) / 2;
      j = P;
    }
    M[i * LM + j] = gram_entry(i, j);
  }
  __syncthreads();
  // Cholesky (lower, in place) with a relative pivot test; a failed pivot restarts once with a ridge.
  // LEFT-looking (Crout), lane = row: column j is  L(i, j) = (M(i, j) - sum_{p < j} L(i, p) L(j, p)) / L(j, j)  for the rows
  // i >= j -- every lane runs the same j-long dot product over its own row (odd pitch: conflict-free) against the
  // broadcast row j, one barrier per column and no read-modify-write of the trailing matrix.  (Round 2's right-looking
  // form spent ~1.5 us per pivot on its unbalanced rank-1 update and two more barriers: ~100 us of a 460-us query.)
  double dmax = 0.0;
  for (int i = 0; i < P; ++i) dmax = fmax(dmax, M[i * LM + i]);
  double ridge = 0.0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    bool ok = true;
    for (int j = 0; j < P; ++j) {
      const double* rj = M + j * LM;
      double sjj = 0.0;  // the pivot, formed redundantly by every lane (uniform control flow, no broadcast needed)
      {
        double t0 = 0.0, t1 = 0.0;
        int p2 = 0;
        for (; p2 + 1 < j; p2 += 2) {
          t0 += rj[p2] * rj[p2];
          t1 += rj[p2 + 1] * rj[p2 + 1];
        }
        if (p2 < j) t0 += rj[p2] * rj[p2];
        sjj = rj[j] + ridge - (t0 + t1);
      }
      if (!(sjj > 1e-13 * dmax)) {
        ok = false;
        break;  // uniform
      }
      // 1 / sqrt(pivot): hardware estimate + two Newton steps (the IEEE sqrt and division sequences cost more than the
      // whole dot product of a column)
      double rinv = __builtin_amdgcn_rsq(sjj);
      rinv = rinv * (1.5 - 0.5 * sjj * rinv * rinv);
      rinv = rinv * (1.5 - 0.5 * sjj * rinv * rinv);
      auto col_entry = [&](int i) -> double {
        const double* ri = M + i * LM;
        double t0 = 0.0, t1 = 0.0;
        int p2 = 0;
        for (; p2 + 1 < j; p2 += 2) {
          t0 += ri[p2] * rj[p2];
          t1 += ri[p2 + 1] * rj[p2 + 1];
        }
        if (p2 < j) t0 += ri[p2] * rj[p2];
        return (ri[j] - (t0 + t1)) * rinv;
      };
      // rows j + 1 + lane and + 64 (P <= 129 covers order 2 up to k = 14) are held back until every lane has read the
      // old column; longer columns write their further rows at once (no other lane reads a row that is not its own)
      const int i0 = j + 1 + lane, i1 = i0 + 64;
      double c0 = 0.0, c1 = 0.0;
      if (i0 < P) c0 = col_entry(i0);
      if (i1 < P) c1 = col_entry(i1);
      for (int i = i1 + 64; i < P; i += 64) M[i * LM + j] = col_entry(i);
      __syncthreads();  // every read of row j (the pivot's old value included) is done
      if (i0 < P) M[i0 * LM + j] = c0;
      if (i1 < P) M[i1 * LM + j] = c1;
      if (lane == 0) {
        M[j * LM + j] = sjj * rinv;  // L(j, j)
        dinv[j] = rinv;
      }
      __syncthreads();
    }
    if (ok) break;
    if (attempt == 1) {  // still singular: give up on this query (zero gradient, status 2)
      if (lane == 0) *flag = 2;
      break;
    }
    // rebuild the lower triangle (the factorisation overwrote part of it) and retry with a ridge
    __syncthreads();
    for (int e = lane; e < P * (P + 1) / 2; e += 64) {
      int i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
      while (i * (i + 1) / 2 > e) --i;
      while ((i + 1) * (i + 2) / 2 <= e) ++i;
      const int j = e - i * (i + 1) / 2;
      M[i * LM + j] = gram_entry(i, j);
    }
    ridge = 1e-10 * dmax;
    if (lane == 0) *flag = 1;
    __syncthreads();
  }
  __syncthreads();
  const int fl = *flag;
  if (fl != 2) {
    // L z = rhs, L^T beta = z in the COLUMN-oriented form: once an unknown is known its multiple of column i of L (of
    // row i for the transposed solve) leaves the remaining right-hand side, lane = row -- no reductions and no divisions
    // (round 2 took a 64-lane f64 reduction and an IEEE division per unknown: ~200 us per query)
    for (int r = lane; r < P; r += 64) beta[r] = M[r * LM + P];
    __syncthreads();
    for (int i = 0; i < P; ++i) {
      const double zi = beta[i] * dinv[i];
      __syncthreads();
      if (lane == 0) beta[i] = zi;
      for (int r = i + 1 + lane; r < P; r += 64) beta[r] -= M[r * LM + i] * zi;
      __syncthreads();
    }
    for (int i = P - 1; i >= 0; --i) {
      const double bi = beta[i] * dinv[i];
      __syncthreads();
      if (lane == 0) beta[i] = bi;
      for (int r = lane; r < i; r += 64) beta[r] -= M[i * LM + r] * bi;
      __syncthreads();
    }
  }
  // gradient at x0
  for (int m = lane; m < k; m += 64) {
    double gm = 0.0;
    if (fl != 2) {
      gm = beta[m];  // d/dx_m of the polynomial in (x - x0) at x0: the quadratic terms vanish there
    }
    g[q * ldg + m] = out_scale * gm;
  }
  if (lane == 0 && status) status[q] = fl;
  __syncthreads();  // the LDS images are free for the next query of this workgroup
  }
}
// ---- order-1 fits, second generation (round 3) ---------------------------------------------------------------------
// est_grad_lin (active_subspaces.rs:99-120; linear_fit, stats_corr.rs:146-159) for k <= 64 features: the same normal
// equations of the design [x - x0, 1] and the same Cholesky solution as grad_fit_kernel, rebuilt around what bounded
// that kernel at BASELINE config 5 (1e6 queries, k = 64, 80 neighbours: ~0.45 ms per query on ONE wave with 75 KB of LDS,
// two queries per CU; 1 s of the stage): every one of its phases walked LDS one dependent read at a time.  Here
//  * the neighbours never touch LDS: lane (c = lane & 15, g = lane >> 4) loads x(nbr[4 s + g], 16 t + c) - x0 straight into
//    the fragment of v_mfma_f64_16x16x4_f64 it feeds (the A and B fragments of that instruction have the same lane
//    layout, so ONE register serves both sides of G = D^T D), loads of several neighbour groups in flight;
//  * the design carries two more columns, the constant and y itself, so the right-hand side D^T y is row k + 1 of the
//    same product; only the lower-triangle tiles are formed (15 for k = 64), in registers;
//  * LDS holds the packed lower triangle of the (k + 1) x (k + 1) system and its right-hand side only (17 KB at k = 64:
//    eight queries per CU), factorised left-looking with eight products in flight per lane, solved column-oriented.
// NTT = 16-column tiles of the design including the constant and y: ceil((k + 2) / 16).
// Rows of the packed lower triangle of grad_fit_lin_kernel (rows 0 .. P; row P = the right-hand side): row i holds columns
// 0 .. i in an even number of doubles, so every row starts on a 16-byte LDS slot; the rows are stored in a PERMUTED order
// chosen so that row i starts at a slot congruent to i modulo 16.  Lane r owns row r, and the 16 lanes a ds_read_b128 is
// serviced for together have distinct lane numbers modulo 16, so their reads of one column fall into 16 different slots of
// the 256-byte bank line: conflict-free (in index order the starts are m (m + 1) or (m + 1)^2 slots, which take 4 - 8
// residues: 4-way conflicts on every read, 12 % of the kernel's LDS cycles).  The greedy below always finds a row of the
// residue it needs while rows of every residue are left (no padding at P = 65; 2 % at P = 50).
struct FitRowTab {
  unsigned short off[68];  // start of row i in doubles
  unsigned short total;    // doubles in all
};
inline FitRowTab grad_fit_lin_row_table(int P) {
  FitRowTab t{};
  bool placed[68] = {};
  int end = 0;  // in 16-byte slots
  for (int n = 0; n <= P; ++n) {
    int pick = -1, pad = 0;
    for (pad = 0; pad < 16 && pick < 0; ++pad) {
      const int res = (end + pad) & 15;
      for (int r = P; r >= 0; --r)  // the longest unplaced row of that residue
        if (!placed[r] && (r & 15) == res) {
          pick = r;
          break;
        }
      if (pick >= 0) break;
    }
    placed[pick] = true;
    t.off[pick] = (unsigned short)(2 * (end + pad));
    end += pad + ((pick + 2) >> 1);
  }
  t.total = (unsigned short)(2 * end);
  return t;
}
template <int NTT>
__global__ __launch_bounds__(64, 2) void grad_fit_lin_kernel(const double* __restrict__ x, const double* __restrict__ y, int k,
                                                          const double* __restrict__ xq, int64_t n_q,
                                                          const int* __restrict__ nbr, int n_nbrs, double out_scale, double* g,
                                                          int64_t ldg, int* status, unsigned long long* prof, FitRowTab tab) {
  typedef double f64x4v __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int P = k + 1;             // unknowns: k slopes and the constant
  // lower triangle of the AUGMENTED normal equations [G b; b^T .] (P + 1 rows: row P is the right-hand side b = D^T y),
  // packed by rows of even length (16-byte aligned rows: the dot products read two doubles per LDS instruction) -- half
  // the LDS of a square image (18 KB at k = 64), so several queries share a CU and hide each other's barriers and loads
  double* M = (double*)smem;                          // [tab.total]
  double* beta = M + tab.total;                       // [P]  the solution
  double* dinv = beta + P;                            // [P]
  int* nidx = (int*)(dinv + P);                       // [n_nbrs rounded up to 4]
  // row starts: a lane keeps those of its own rows (lane, 64 + lane) in registers -- a pivot row's start is a readlane
  // away (a per-pivot load from the argument block cost more than the bank conflicts the table removes) -- and a copy in
  // LDS serves the per-lane lookups of the store after the Gram products
  unsigned short* offl = (unsigned short*)(nidx + (((n_nbrs + 15) & ~15) + 16));  // [128]
  const int myoff0 = tab.off[threadIdx.x & 63], myoff1 = (threadIdx.x & 63) + 64 <= P ? tab.off[(threadIdx.x & 63) + 64] : 0;
  offl[threadIdx.x & 63] = (unsigned short)myoff0;
  offl[64 + (threadIdx.x & 63)] = (unsigned short)myoff1;
  auto row = [&](int i) __attribute__((always_inline)) -> double* { return M + offl[i]; };            // any i (LDS lookup)
  auto row_u = [&](int i) __attribute__((always_inline)) -> double* {                                  // uniform i
    return M + (i < 64 ? __builtin_amdgcn_readlane(myoff0, i) : __builtin_amdgcn_readlane(myoff1, i - 64));
  };
  const int lane = threadIdx.x, fr = lane & 15, fg = lane >> 4;
  const int64_t q = blockIdx.x;
  if (q >= n_q) return;
  // neighbour groups of 4 (one MFMA k-step), padded with -1 (a zero row of the design) to whole rounds of four groups
  // plus one round the pipeline below reads ahead
  const int n16 = (n_nbrs + 15) & ~15;
  for (int r = lane; r < n16 + 16; r += 64) nidx[r] = r < n_nbrs ? nbr[q * n_nbrs + r] : -1;
  // this lane's columns of the design: x0 for the coordinate columns
  double x0c[NTT];
#pragma unroll
  for (int t = 0; t < NTT; ++t) x0c[t] = (16 * t + fr < k) ? xq[q * k + 16 * t + fr] : 0.0;
  __syncthreads();
  int fl = 0;
  double ridge = 0.0, dmax = 0.0;
  // optional (CORRLA_KNN2_PROF): 100 MHz ticks per phase, summed over the queries: [0] gather + normal equations,
  // [1] Cholesky, [2] triangular solves, [3] queries
  unsigned long long tp0 = prof ? wall_clock64() : 0, tp1 = 0, tp2 = 0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    // ---- G = D^T D over the lower-triangle tiles, D = [x - x0, 1, y] (rows = neighbours) ----
    f64x4v acc[NTT][NTT];
#pragma unroll
    for (int a = 0; a < NTT; ++a)
#pragma unroll
      for (int b = 0; b < NTT; ++b) acc[a][b] = (f64x4v){0, 0, 0, 0};
    const int nsteps = n16 >> 2;  // a multiple of 4
    // The gather: branch-free (every lane loads a clamped row at a clamped column, and y of that row) and RAW -- a group's
    // NTT + 1 loads are issued back to back and nothing touches the values until its products are due, four groups later.
    // (Round 3's first version subtracted x0 right behind each load: hipcc then waited for every single load, 100
    // dependent DRAM round trips per query at 80 neighbours, and the conditional tail of the pipeline kept the waits at
    // vmcnt(0).  Padding the neighbour list to whole rounds makes the pipeline branch-free.)
    struct Group {
      double xr[NTT], yr;
      int idx;
    };
    auto fragment = [&](int s, Group& gsrc) __attribute__((always_inline)) {
      gsrc.idx = nidx[4 * s + fg];
      const int64_t id = gsrc.idx >= 0 ? gsrc.idx : 0;
#pragma unroll
      for (int t = 0; t < NTT; ++t) {
        const int col = 16 * t + fr;
        gsrc.xr[t] = x[id * k + (col < k ? col : k - 1)];
      }
      gsrc.yr = y[id];
    };
    auto products = [&](const Group& gsrc) __attribute__((always_inline)) {
      double fv[NTT];
#pragma unroll
      for (int t = 0; t < NTT; ++t) {
        const int col = 16 * t + fr;
        double v = gsrc.xr[t] - x0c[t];
        v = col < k ? v : (col == k ? 1.0 : (col == k + 1 ? gsrc.yr : 0.0));
        fv[t] = gsrc.idx >= 0 ? v : 0.0;
      }
#pragma unroll
      for (int a = 0; a < NTT; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fv[a], fv[b], acc[a][b], 0, 0, 0);
    };
    // four neighbour groups in flight: the loads of step s + 4 are issued right after the MFMAs of step s (the last round
    // reads one round of padding ahead: rows of point 0, never used)
    Group f0, f1, f2, f3;
    fragment(0, f0);
    fragment(1, f1);
    fragment(2, f2);
    fragment(3, f3);
    for (int s = 0; s < nsteps; s += 4) {
      products(f0);
      fragment(s + 4, f0);
      products(f1);
      fragment(s + 5, f1);
      products(f2);
      fragment(s + 6, f2);
      products(f3);
      fragment(s + 7, f3);
    }
    // D layout of v_mfma_f64_16x16x4_f64: column = lane & 15, row = (lane >> 4) + 4 reg
#pragma unroll
    for (int a = 0; a < NTT; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int i = 16 * a + fg + 4 * rg, j = 16 * b + fr;
          const double v = acc[a][b][rg];
          if (i < P && j <= i) row(i)[j] = v + ((i == j) ? ridge : 0.0);
          if (i == P && j < P) row(P)[j] = v;  // row k + 1 of G = D^T y
        }
    __syncthreads();
    if (attempt == 0) {
      double dm = 0.0;
      for (int i = lane; i < P; i += 64) dm = fmax(dm, row(i)[i]);
      for (int off = 32; off > 0; off >>= 1) dm = fmax(dm, __shfl_xor(dm, off, 64));
      dmax = dm;
    }
    // ---- Cholesky of the augmented system, BLOCKED left-looking, fixed row ownership: lane r keeps rows r and 64 + r of
    // the P + 1 <= 66 rows (row P is the right-hand side: its factor row is the forward-solved z = L^-1 b, only the back
    // substitution is left afterwards).
    //  * Panels of 16 pivot columns.  Before a panel, the contribution of ALL earlier panels is taken off its columns by
    //    v_mfma_f64_16x16x4_f64, one 16 x 16 tile of rows at a time:  C(I, J) -= sum_p L(I, p) L(J, p)^T  -- both operands
    //    are read the same way (lane (c, g) of step s: L[16 T + c][16 p + 4 s + g]) straight from the row-packed image.
    //  * Inside a panel, per pivot j a lane computes ONE dot product of at most 15 terms (its row against row j over the
    //    panel's columns: broadcast reads, two doubles per LDS instruction) and keeps the running diagonal of its row in
    //    a register, from which pivot r is read by v_readlane when its turn comes.  One barrier per pivot.
    // (Tick counters, CORRLA_KNN2_PROF: the unblocked form -- dot products over all earlier columns, 32 half-rate v_fma_f64
    //  per lane and pivot on average -- was 59 % of the kernel.)
    if (prof && attempt == 0) tp1 = wall_clock64();
    const int nrows = P + 1;
    const int r0 = lane, r1 = 64 + lane;
    double* const row0 = M + myoff0;
    double* const row1 = M + myoff1;
    double dg0 = 0.0, dg1 = 0.0;
    auto bcast = [&](double v, int src) __attribute__((always_inline)) -> double {  // src uniform
      const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
      return __hiloint2double(hi, lo);
    };
    typedef double f64x2v __attribute__((ext_vector_type(2)));
    bool ok = true;
    for (int J = 0; J < NTT && ok; ++J) {
      const int c0 = 16 * J, c1 = min(c0 + 16, P);
      if (c0 >= P) break;  // (a last tile that only holds the right-hand side row has no pivots)
      if (J > 0) {
        // C(I, J) -= sum_{p < J} L(I, p) L(J, p)^T for the row tiles I >= J
        const int bj = c0 + fr;                                    // pivot row this lane feeds as the B operand
        const double* bptr = row(bj < P ? bj : 0);
        for (int I = J; I < NTT; ++I) {
          const int ai = 16 * I + fr;
          const double* aptr = row(ai <= P ? ai : 0);
          f64x4v acc4 = {0.0, 0.0, 0.0, 0.0};
          for (int pk = 0; pk < 4 * J; ++pk) {                     // k-steps of 4 columns over the earlier panels
            const int col = 4 * pk + fg;
            const double av = ai <= P ? -aptr[col] : 0.0;
            const double bv = bj < P ? bptr[col] : 0.0;
            acc4 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc4, 0, 0, 0);
          }
          // D: column = lane & 15, row = (lane >> 4) + 4 reg
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int i = 16 * I + fg + 4 * rg, jj = c0 + fr;
            if (jj < P && ((i < P && jj <= i) || i == P)) row(i)[jj] += acc4[rg];
          }
        }
        __syncthreads();
      }
      // running diagonals of the panel's rows start from the updated image
      if (r0 >= c0 && r0 < c1) dg0 = row0[r0];
      if (r1 >= c0 && r1 < c1) dg1 = row1[r1];
      for (int j = c0; j < c1; ++j) {
        const double sjj = j < 64 ? bcast(dg0, j) : bcast(dg1, j - 64);
        if (!(sjj > 1e-13 * dmax)) {
          ok = false;
          break;  // uniform
        }
        double rinv = __builtin_amdgcn_rsq(sjj);
        rinv = rinv * (1.5 - 0.5 * sjj * rinv * rinv);
        rinv = rinv * (1.5 - 0.5 * sjj * rinv * rinv);
        double* const rj = row_u(j);
        auto dot = [&](const double* ri) __attribute__((always_inline)) -> double {  // over the panel's columns c0 .. j - 1
          f64x2v t0 = {0.0, 0.0}, t1 = {0.0, 0.0};
          int p2 = c0;
          for (; p2 + 3 < j; p2 += 4) {
            t0 += *(const f64x2v*)(ri + p2) * *(const f64x2v*)(rj + p2);
            t1 += *(const f64x2v*)(ri + p2 + 2) * *(const f64x2v*)(rj + p2 + 2);
          }
          for (; p2 + 1 < j; p2 += 2) t0 += *(const f64x2v*)(ri + p2) * *(const f64x2v*)(rj + p2);
          double tail = 0.0;
          if (p2 < j) tail = ri[p2] * rj[p2];
          const f64x2v tt = t0 + t1;
          return (tt[0] + tt[1]) + tail;
        };
        const bool a0 = r0 > j && r0 < nrows, a1 = r1 > j && r1 < nrows;
        if (a0) {
          const double c = (row0[j] - dot(row0)) * rinv;
          row0[j] = c;
          dg0 -= c * c;
        }
        if (a1) {
          const double c = (row1[j] - dot(row1)) * rinv;
          row1[j] = c;
          dg1 -= c * c;
        }
        if (lane == (j & 63)) {  // the owner of row j
          rj[j] = sjj * rinv;
          dinv[j] = rinv;
        }
        __syncthreads();  // column j of the factor is in LDS before pivot j + 1 reads row j + 1 up to it
      }
    }
    if (ok) break;
    if (attempt == 1) {
      fl = 2;  // still singular: zero gradient
      break;
    }
    fl = 1;  // numerically singular design: once more with a 1e-10 relative ridge (the documented deviation)
    ridge = 1e-10 * dmax;
    __syncthreads();
  }
  if (prof) tp2 = wall_clock64();
  if (fl != 2) {
    for (int r = lane; r < P; r += 64) beta[r] = row_u(P)[r];  // z = L^-1 b, from the factorisation
    __syncthreads();
    for (int i = P - 1; i >= 0; --i) {
      const double bi = beta[i] * dinv[i];
      __syncthreads();
      if (lane == 0) beta[i] = bi;
      const double* rowi = row_u(i);
      for (int r = lane; r < i; r += 64) beta[r] -= rowi[r] * bi;
      __syncthreads();
    }
  }
  for (int m = lane; m < k; m += 64) g[q * ldg + m] = out_scale * (fl != 2 ? beta[m] : 0.0);
  if (lane == 0 && status) status[q] = fl;
  if (prof && lane == 0) {
    const unsigned long long tp3 = wall_clock64();
    atomicAdd(prof + 0, tp1 - tp0);
    atomicAdd(prof + 1, tp2 - tp1);
    atomicAdd(prof + 2, tp3 - tp2);
    atomicAdd(prof + 3, 1ull);
  }
}
inline size_t grad_fit_lin_lds_bytes(int k, int n_nbrs) {
  const int P = k + 1;
  return ((size_t)grad_fit_lin_row_table(P).total + 2 * P) * 8 + (size_t)(((n_nbrs + 15) & ~15) + 16) * 4 + 128 * 2 + 64;
}

// m_in_lds = false: the normal equations live in global memory (grad_fit_kernel's m_glob)
inline size_t grad_fit_lds_bytes(int k, int n_nbrs, int order, bool m_in_lds = true) {
  const int P = order == 1 ? k + 1 : k + k * (k + 1) / 2 + 1;
  const int LM = (P + 1) | 1;
  return ((size_t)n_nbrs * k + n_nbrs + (m_in_lds ? (size_t)P * LM : (size_t)0) + k + 2 * P) * 8 + (size_t)(2 * P + 4 + n_nbrs) * 4 + 64;
}
inline size_t grad_fit_m_elems(int k, int order) {
  const int P = order == 1 ? k + 1 : k + k * (k + 1) / 2 + 1;
  return (size_t)P * (size_t)((P + 1) | 1);
}

}  // namespace k
}  // namespace corrla
