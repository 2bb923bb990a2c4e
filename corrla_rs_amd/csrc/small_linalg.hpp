// Host-side dense kernels for the l x l core (l = rank + oversamples, a few hundred at most).
// The north star keeps "the final (k+p) x (k+p) SVD on host"; everything m- or n-sized runs in
// HIP kernels.  All routines are f64, column-major, and written from the textbook algorithms
// (no LAPACK dependency).  They replace, for the small core only, the faer calls at
// random_svd.rs:38,57 (qr -> here: Cholesky of the device-computed Gram) and random_svd.rs:89 (svd).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <vector>

namespace corrla {
namespace small {

// In-place upper Cholesky of the symmetric n x n matrix G (column-major, leading dim ld):
// on success the upper triangle holds R with G = R^T R (strict lower triangle is zeroed).
// Fails (returns false) as soon as a pivot drops to <= piv_rel * G(j,j): the Gram matrix is
// numerically singular at working precision.  min_ratio receives min_j R(j,j)^2 / G(j,j).
inline bool chol_upper(int n, double* g, int ld, double piv_rel, double* min_ratio) {
  double mr = 1.0;
  for (int j = 0; j < n; ++j) {
    double* cj = g + (size_t)j * ld;
    const double gjj = cj[j];
    for (int i = 0; i < j; ++i) {
      const double* ci = g + (size_t)i * ld;
      double s = cj[i];
      for (int k = 0; k < i; ++k) s -= ci[k] * cj[k];
      cj[i] = s / ci[i];
    }
    double d = gjj;
    for (int k = 0; k < j; ++k) d -= cj[k] * cj[k];
    if (!(d > piv_rel * gjj) || !(gjj > 0.0) || !std::isfinite(d)) {
      if (min_ratio) *min_ratio = (gjj > 0.0) ? d / gjj : 0.0;
      return false;
    }
    mr = std::min(mr, d / gjj);
    cj[j] = std::sqrt(d);
    for (int i = j + 1; i < n; ++i) cj[i] = 0.0;
  }
  if (min_ratio) *min_ratio = mr;
  return true;
}

// In-place inverse of an upper-triangular n x n matrix (column-major).
inline void triu_inverse(int n, double* r, int ld) {
  for (int j = 0; j < n; ++j) {
    double* cj = r + (size_t)j * ld;
    const double djj = 1.0 / cj[j];
    // solve R(0:j,0:j) x = -R(0:j,j) * djj using the already inverted leading block:
    // inv(0:j, j) = -inv(0:j,0:j) * R(0:j,j) * djj
    std::vector<double> tmp(cj, cj + j);
    for (int i = 0; i < j; ++i) cj[i] = 0.0;
    for (int k = 0; k < j; ++k) {
      const double* ck = r + (size_t)k * ld;  // column k of the inverted block (upper)
      const double t = tmp[k];
      for (int i = 0; i <= k; ++i) cj[i] += ck[i] * t;
    }
    for (int i = 0; i < j; ++i) cj[i] *= -djj;
    cj[j] = djj;
  }
}

// One-sided (Hestenes) Jacobi SVD of a general n x n matrix C (column-major, ld):
//   C = U diag(S) V^T, S descending and non-negative.
// U, V: n x n column-major (ld = n), S: n.  Columns of U belonging to exactly-zero singular
// values are zero vectors.  tol is the relative off-orthogonality at which a column pair is
// considered converged (use ~eps of the DATA precision).  Returns the number of sweeps, or -1
// when the input holds non-finite values.
inline int jacobi_svd(int n, const double* c, int ld, double* u, double* s, double* v, double tol) {
  std::vector<double> w((size_t)n * n);
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) {
      const double x = c[(size_t)j * ld + i];
      if (!std::isfinite(x)) return -1;
      w[(size_t)j * n + i] = x;
    }
  std::vector<double> vv((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) vv[(size_t)i * n + i] = 1.0;
  std::vector<double> nrm2(n);
  int sweeps = 0;
  const int max_sweeps = 60;
  for (; sweeps < max_sweeps; ++sweeps) {
    for (int j = 0; j < n; ++j) {
      const double* wj = &w[(size_t)j * n];
      double a = 0.0;
      for (int i = 0; i < n; ++i) a += wj[i] * wj[i];
      nrm2[j] = a;
    }
    bool rotated = false;
    for (int p = 0; p < n - 1; ++p) {
      double* wp = &w[(size_t)p * n];
      double* vp = &vv[(size_t)p * n];
      for (int q = p + 1; q < n; ++q) {
        double* wq = &w[(size_t)q * n];
        const double alpha = nrm2[p], beta = nrm2[q];
        if (alpha == 0.0 || beta == 0.0) continue;
        double gamma = 0.0;
        for (int i = 0; i < n; ++i) gamma += wp[i] * wq[i];
        if (std::fabs(gamma) <= tol * std::sqrt(alpha * beta)) continue;
        rotated = true;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / std::sqrt(1.0 + t * t);
        const double sn = cs * t;
        for (int i = 0; i < n; ++i) {
          const double x = wp[i], y = wq[i];
          wp[i] = cs * x - sn * y;
          wq[i] = sn * x + cs * y;
        }
        double* vq = &vv[(size_t)q * n];
        for (int i = 0; i < n; ++i) {
          const double x = vp[i], y = vq[i];
          vp[i] = cs * x - sn * y;
          vq[i] = sn * x + cs * y;
        }
        nrm2[p] = std::max(0.0, alpha - t * gamma);
        nrm2[q] = std::max(0.0, beta + t * gamma);
      }
    }
    if (!rotated) break;
  }
  // singular values, ordering
  std::vector<double> sv(n);
  for (int j = 0; j < n; ++j) {
    const double* wj = &w[(size_t)j * n];
    double a = 0.0;
    for (int i = 0; i < n; ++i) a += wj[i] * wj[i];
    sv[j] = std::sqrt(a);
  }
  std::vector<int> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return sv[a] > sv[b]; });
  for (int jj = 0; jj < n; ++jj) {
    const int j = order[jj];
    s[jj] = sv[j];
    const double inv = sv[j] > 0.0 ? 1.0 / sv[j] : 0.0;
    for (int i = 0; i < n; ++i) {
      u[(size_t)jj * n + i] = w[(size_t)j * n + i] * inv;
      v[(size_t)jj * n + i] = vv[(size_t)j * n + i];
    }
  }
  return sweeps;
}

}  // namespace small
}  // namespace corrla
