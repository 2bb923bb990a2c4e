// Active-subspace gradient stage on the device, f64 like the reference:
//   PolyGradientEstimator::{nearest_points, est_grad_lin, est_grad_quad}   src/lib_math_utils/active_subspaces.rs:66-141
//   linear_fit / quad_fit / build_vandermonde / jac_from_lin / jac_from_quad src/lib_math_utils/stats_corr.rs:110-249
//   ActiveSsRsvd::create_grad_mat                                           src/lib_math_utils/active_subspaces.rs:215-229
// The reference walks a kd-tree per sample and takes an SVD-based pseudo-inverse per sample, serially.  Here:
//   knn_kernel      : exact n nearest neighbours by squared Euclidean distance, brute force (in k = 64 dimensions a
//                     kd-tree degenerates to that anyway): 16 queries per workgroup share every LDS-staged chunk of
//                     64 support points; each wave keeps a sorted top-n list per query in LDS (ties -> lower index);
//   grad_fit_kernel : one wave per query: gathers the neighbours, forms the normal equations of the reference's
//                     design matrix ([x - x0, 1] for order 1 -- the slopes do not depend on the shift --, [x, x_a x_b]
//                     without a constant for order 2, exactly build_vandermonde) in LDS, Cholesky-solves them and
//                     writes the gradient (order 2: the analytic gradient of the fitted quadratic; the reference
//                     takes forward differences with eps = 1e-10, which agree to ~1e-6 relative).
// The gradient matrix is written in the reference's k x N column-major layout (N rows of k contiguous values), which
// is the row-major tall matrix the RSVD kernels take directly.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace corrla {
namespace k {

constexpr int kGradMaxDim = 64;    // features k
constexpr int kGradMaxNbr = 160;   // neighbours per query
constexpr int kGradMaxCols = 65;   // design-matrix columns: k + 1 (order 1), k + k (k + 1) / 2 (order 2)
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

// g[q * ldg + m] = out_scale * d(fit)/dx_m at query q.  status[q]: 0 ok, 1 ridge-regularised (rank-deficient design).
__global__ __launch_bounds__(64) void grad_fit_kernel(const double* __restrict__ x, const double* __restrict__ y, int k,
                                                      const double* __restrict__ xq, int64_t n_q,
                                                      const int* __restrict__ nbr, int n_nbrs, int order, double out_scale,
                                                      double* g, int64_t ldg, int* status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int P = order == 1 ? k + 1 : k + k * (k + 1) / 2;  // design columns
  const int LM = P + 1;                                    // row pitch of M (the right-hand side is column P)
  double* xn = (double*)smem;                 // [n_nbrs][k] neighbour coordinates (minus x0 for order 1)
  double* yn = xn + (size_t)n_nbrs * k;       // [n_nbrs]
  double* M = yn + n_nbrs;                    // [P][LM] normal equations, lower triangle -> Cholesky factor
  double* x0 = M + (size_t)P * LM;            // [k]
  double* beta = x0 + k;                      // [P]
  int* pa = (int*)(beta + P);                 // [P] column -> (a, b); b = -1: linear term a; a = -1: constant
  int* pb = pa + P;
  int* flag = pb + P;
  const int lane = threadIdx.x;
  const int64_t q = blockIdx.x;
  if (q >= n_q) return;
  for (int d = lane; d < k; d += 64) x0[d] = xq[q * k + d];
  for (int c = lane; c < P; c += 64) {
    if (c < k) {
      pa[c] = c;
      pb[c] = -1;
    } else if (order == 1) {
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
  for (int idx = lane; idx < n_nbrs * k; idx += 64) {
    const int r = idx / k, d = idx - r * k;
    const int p = nbr[q * n_nbrs + r];
    xn[idx] = x[(int64_t)p * k + d] - (order == 1 ? x0[d] : 0.0);
  }
  for (int r = lane; r < n_nbrs; r += 64) yn[r] = y[nbr[q * n_nbrs + r]];
  __syncthreads();
  auto design = [&](int r, int c) -> double {
    const int a = pa[c], b = pb[c];
    if (a < 0) return 1.0;
    const double va = xn[r * k + a];
    return b < 0 ? va : va * xn[r * k + b];
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
    double s = 0.0;
    for (int r = 0; r < n_nbrs; ++r) s += design(r, i) * (j == P ? yn[r] : design(r, j));
    M[i * LM + j] = s;
  }
  __syncthreads();
  // Cholesky (lower, in place) with a relative pivot test; a failed pivot restarts once with a ridge
  double dmax = 0.0;
  for (int i = 0; i < P; ++i) dmax = fmax(dmax, M[i * LM + i]);
  double ridge = 0.0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    bool ok = true;
    for (int j = 0; j < P; ++j) {
      // row j of L: L(j, c) for c < j is final; pivot
      double s = 0.0;
      for (int c = lane; c < j; c += 64) s += M[j * LM + c] * M[j * LM + c];
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
      const double piv = M[j * LM + j] + ridge - s;
      if (!(piv > 1e-13 * dmax)) {
        ok = false;
        break;  // uniform
      }
      const double ljj = sqrt(piv);
      __syncthreads();
      if (lane == 0) M[j * LM + j] = ljj;
      // column j below the diagonal: L(i, j) = (M(i, j) - sum_c L(i, c) L(j, c)) / ljj
      for (int i = j + 1 + lane; i < P; i += 64) {
        double t = M[i * LM + j];
        for (int c = 0; c < j; ++c) t -= M[i * LM + c] * M[j * LM + c];
        M[i * LM + j] = t / ljj;
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
      double s = 0.0;
      for (int r = 0; r < n_nbrs; ++r) s += design(r, i) * design(r, j);
      M[i * LM + j] = s;
    }
    ridge = 1e-10 * dmax;
    if (lane == 0) *flag = 1;
    __syncthreads();
  }
  __syncthreads();
  const int fl = *flag;
  if (fl != 2) {
    // L z = rhs (forward), L^T beta = z (backward); one unknown at a time, dot products across the lanes
    for (int i = 0; i < P; ++i) {
      double s = 0.0;
      for (int c = lane; c < i; c += 64) s += M[i * LM + c] * beta[c];
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
      __syncthreads();
      if (lane == 0) beta[i] = (M[i * LM + P] - s) / M[i * LM + i];
      __syncthreads();
    }
    for (int i = P - 1; i >= 0; --i) {
      double s = 0.0;
      for (int c = i + 1 + lane; c < P; c += 64) s += M[c * LM + i] * beta[c];
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
      __syncthreads();
      if (lane == 0) beta[i] = (beta[i] - s) / M[i * LM + i];
      __syncthreads();
    }
  }
  // gradient at x0
  for (int m = lane; m < k; m += 64) {
    double gm = 0.0;
    if (fl != 2) {
      gm = beta[m];
      if (order == 2) {
        for (int c = k; c < P; ++c) {
          const int a = pa[c], b = pb[c];
          if (a == m) gm += beta[c] * x0[b];
          if (b == m) gm += beta[c] * x0[a];
        }
      }
    }
    g[q * ldg + m] = out_scale * gm;
  }
  if (lane == 0 && status) status[q] = fl;
}
inline size_t grad_fit_lds_bytes(int k, int n_nbrs, int order) {
  const int P = order == 1 ? k + 1 : k + k * (k + 1) / 2;
  return ((size_t)n_nbrs * k + n_nbrs + (size_t)P * (P + 1) + k + P) * 8 + (size_t)(2 * P + 4) * 4 + 64;
}

}  // namespace k
}  // namespace corrla
