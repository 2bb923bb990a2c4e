"""CPU oracle for the active-subspace gradient stage -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatements of
  PolyGradientEstimator::new / nearest_points / est_grad_lin / est_grad_quad
                                         src/lib_math_utils/active_subspaces.rs:66-141
  linear_fit / jac_from_lin / build_vandermonde / mat_col_interactions / quad_fit / quad_eval / jac_from_quad
                                         src/lib_math_utils/stats_corr.rs:110-249
  mat_pinv                               src/lib_math_utils/mat_utils.rs:37-53   (1 / (s_i + 1e-14), no truncation)
  ActiveSsRsvd::create_grad_mat / fit_svd / fit, FittedActiveSsRsvd
                                         src/lib_math_utils/active_subspaces.rs:143-277
Third-party pieces the reference relies on: `kdtree 0.7.0` nearest(x0, n, squared_euclidean) (exact n nearest
neighbours by squared Euclidean distance; the order among equal distances is unspecified) -- restated as a
brute-force search with ties broken by the lower index; faer SVD / eigendecomposition -- numpy.linalg.

Pinned by the reference's own tests (active_subspaces.rs:280-395, run here on seeded samples):
  test_grad_est : y = x1^2 + x2^2, 100 samples, order 2, 14 neighbours: grad(0,0) = (0,0) @1e-2, grad(1,0) = (2,0)
                  @1e-2, grad(-1,0) = -grad(1,0) @1e-2;
  test_active_ss: y = 0.2 x1 + 0.5 x2^2 + 0.1 x3 x1: first component dominated by x2, sigma_0 > sigma_1,
                  grad(0,1,0) = (0.2, 1, 0) @1e-1, x2 dominates var_diag_evd_sensi.
fit_svd itself has no reference test."""
import numpy as np

from . import rsvd_oracle as orc


def mat_pinv(x):
    """mat_utils.rs:37-53: V diag(1 / (s_i + 1e-14)) U^T over ALL singular values."""
    u, s, vt = np.linalg.svd(np.asarray(x, np.float64), full_matrices=False)
    return (vt.T * (1.0 / (s + 1.0e-14))) @ u.T


def nearest_indices(x_mat, x0, n_nbrs):
    """kd_tree.nearest(x0, n_nbrs, squared_euclidean) (active_subspaces.rs:89-92), brute force; ties -> lower index."""
    d2 = np.sum((np.asarray(x_mat, np.float64) - np.asarray(x0, np.float64).reshape(1, -1)) ** 2, axis=1)
    order = np.lexsort((np.arange(d2.size), d2))
    return order[: min(n_nbrs, d2.size)]


def mat_col_interactions(x, include_self):
    """stats_corr.rs:112-143: columns x_a * x_b for a <= b (a < b without self interactions), a-major order."""
    x = np.asarray(x, np.float64)
    cols = []
    for a in range(x.shape[1]):
        for b in range(a, x.shape[1]):
            if a == b and not include_self:
                continue
            cols.append(x[:, a] * x[:, b])
    return np.stack(cols, axis=1) if cols else np.zeros((x.shape[0], 0))


def build_vandermonde(x, include_self=True):
    """stats_corr.rs:198-207: hstack(x, interactions, ones) -- the constant column is the LAST one."""
    x = np.asarray(x, np.float64)
    return np.hstack([x, mat_col_interactions(x, include_self), np.ones((x.shape[0], 1))])


def linear_fit(x, y):
    """stats_corr.rs:146-159: pinv([x 1]) y."""
    x = np.asarray(x, np.float64)
    return mat_pinv(np.hstack([x, np.ones((x.shape[0], 1))])) @ np.asarray(y, np.float64).reshape(-1, 1)


def jac_from_lin(x, y):
    """stats_corr.rs:164-169: the k slopes as a 1 x k row."""
    return linear_fit(x, y)[: np.asarray(x).shape[1], :].T.copy()


def quad_fit(x, y):
    """stats_corr.rs:213-219."""
    return mat_pinv(build_vandermonde(x, True)) @ np.asarray(y, np.float64).reshape(-1, 1)


def quad_eval(x, coeffs):
    """stats_corr.rs:222-226."""
    return build_vandermonde(x, True) @ coeffs


def jac_from_quad(x0, coeffs, eps=1.0e-10):
    """stats_corr.rs:230-249: forward differences with eps = 1e-10 of the fitted quadratic."""
    x0 = np.asarray(x0, np.float64).reshape(1, -1)
    y0 = quad_eval(x0, coeffs)
    out = np.zeros_like(x0)
    for k in range(x0.shape[1]):
        xp = x0.copy()
        xp[0, k] += eps
        out[0, k] = ((quad_eval(xp, coeffs) - y0) * (1.0 / eps))[0, 0]
    return out


def jac_of_quad_exact(x0, coeffs):
    """NOT in the reference: the exact gradient of the quadratic quad_fit returned (columns as in build_vandermonde), used
    by the tests to separate the fit itself from the ~1e-6 |y| rounding noise of the reference's forward differences."""
    x0 = np.asarray(x0, np.float64).ravel()
    k = x0.size
    c = np.asarray(coeffs, np.float64).ravel()
    g = c[:k].copy()
    col = k
    for a in range(k):
        for b in range(a, k):
            g[a] += c[col] * x0[b]
            g[b] += c[col] * x0[a]
            col += 1
    return g.reshape(1, -1)


class PolyGradientEstimator:
    """active_subspaces.rs:21-141."""

    def __init__(self, x_mat, y, est_order, n_nbrs):
        self.x_mat = np.asarray(x_mat, np.float64)
        self.y = np.asarray(y, np.float64).reshape(-1, 1)
        self.est_order, self.n_nbrs, self.k = int(est_order), int(n_nbrs), self.x_mat.shape[1]
        if self.est_order not in (1, 2):
            raise ValueError("Not implemented est order")   # the reference panics (:60)
        self.exact_quad_gradient = False   # test hook, see jac_of_quad_exact

    def nearest_points(self, x0):
        idx = nearest_indices(self.x_mat, x0, self.n_nbrs)
        return self.x_mat[idx], self.y[idx]

    def grad_at(self, x0):
        k = self.k
        if self.est_order == 1:
            if not (self.x_mat.shape[0] > k + 1 and self.n_nbrs > k + 1):   # asserts at :118-119
                raise ValueError("linear fit needs more than k + 1 samples and neighbours")
            xn, yn = self.nearest_points(x0)
            return jac_from_lin(xn, yn)
        need = k * (k + 3) // 2
        if not (self.x_mat.shape[0] > need and self.n_nbrs > need):         # asserts at :129-130
            raise ValueError("quadratic fit needs more than k (k + 3) / 2 samples and neighbours")
        xn, yn = self.nearest_points(x0)
        if self.exact_quad_gradient:
            return jac_of_quad_exact(x0, quad_fit(xn, yn))
        return jac_from_quad(x0, quad_fit(xn, yn))


def create_grad_mat(grad_est, x_mat):
    """active_subspaces.rs:215-229: G (k x N), column i = gradient at row i of x_mat."""
    x = np.asarray(x_mat, np.float64)
    g = np.zeros((x.shape[1], x.shape[0]))
    for i in range(x.shape[0]):
        g[:, i] = grad_est.grad_at(x[i]).ravel()
    return g


def fit_svd(grad_est, x_mat, n_comps, n_iter=8, n_oversamples=10, omega=None):
    """active_subspaces.rs:233-250.  Returns (components k x r, singular values as an r x r diagonal matrix)."""
    x = np.asarray(x_mat, np.float64)
    g = create_grad_mat(grad_est, x) * (1.0 / np.sqrt(float(x.shape[0])))
    u, s, _vt = orc.random_svd(g, min(x.shape[1], n_comps), n_iter, n_oversamples, omega=omega)
    return u, np.diag(s.ravel())


def fit(grad_est, x_mat):
    """active_subspaces.rs:252-277: EVD of G G^T / N, eigenpairs sorted descending (all k of them)."""
    x = np.asarray(x_mat, np.float64)
    g = create_grad_mat(grad_est, x)
    c = g @ g.T * (1.0 / x.shape[0])
    w, v = np.linalg.eigh(c)
    order = np.argsort(-w, kind="stable")
    return v[:, order], np.diag(w[order])


def var_diag_evd_sensi(components, singular_vals):
    """active_subspaces.rs:159-170: diag(components^T * S * components)  (as written in the reference)."""
    m = components.T @ singular_vals @ components
    return np.diag(m).copy()


def sample_mv_normal(cov, n, rng):
    """stats_corr.rs:46-58: rows cov @ z, z ~ N(0, I)  (note: cov itself, not its Cholesky factor)."""
    cov = np.asarray(cov, np.float64)
    return (cov @ rng.standard_normal((cov.shape[0], n))).T.copy()
