"""The f2 oracle (oracle/active_ss_oracle.py) against the reference's own assertions for the gradient stage
(src/lib_math_utils/active_subspaces.rs:280-395), on seeded samples."""
import numpy as np
import pytest

from oracle import active_ss_oracle as aso


def test_reference_test_grad_est():
    # active_subspaces.rs:286-321
    rng = np.random.default_rng(20241008)
    x = aso.sample_mv_normal([[0.9, 0.5], [0.5, 0.9]], 100, rng)
    y = x[:, 0] ** 2 + x[:, 1] ** 2
    est = aso.PolyGradientEstimator(x, y, 2, 14)
    g0 = est.grad_at([0.0, 0.0])
    assert g0.shape == (1, 2) and np.allclose(g0, [[0.0, 0.0]], atol=1e-2)
    g1 = est.grad_at([1.0, 0.0])
    g2 = est.grad_at([-1.0, 0.0])
    assert np.allclose(g1, [[2.0, 0.0]], atol=1e-2)
    assert np.allclose(g1, -g2, atol=1e-2)


def test_reference_test_active_ss():
    # active_subspaces.rs:324-394
    rng = np.random.default_rng(7)
    cov = [[0.9, 0.5, 0.5], [0.5, 0.9, 0.5], [0.5, 0.5, 0.9]]
    x = aso.sample_mv_normal(cov, 100, rng)
    y = 0.2 * x[:, 0] + 0.5 * x[:, 1] ** 2 + 0.10 * x[:, 2] * x[:, 0]
    est = aso.PolyGradientEstimator(x, y, 2, 14)
    comps, sv = aso.fit(est, x)
    n_comps = 2
    assert abs(comps[0, 0]) < abs(comps[1, 0])           # first component dominated by x2
    assert sv[0, 0] > sv[1, 1]
    assert np.allclose(est.grad_at([0.0, 1.0, 0.0]), [[0.2, 1.0, 0.0]], atol=1e-1)
    tr = x @ comps[:, :n_comps]
    assert tr.shape == (100, n_comps) and (tr @ comps[:, :n_comps].T).shape == (100, 3)
    sens = aso.var_diag_evd_sensi(comps, sv)
    assert sens.shape == (3,) and sens[1] > sens[0] and sens[1] > sens[2]
    # fit_svd (no reference test): the RSVD of G / sqrt(N) spans the same leading subspace with the same spectrum
    u, s = aso.fit_svd(est, x, n_comps, omega=np.random.default_rng(1).standard_normal((3, 3)))
    assert np.allclose(np.diag(s) ** 2, np.diag(sv)[:n_comps], rtol=1e-8)
    assert np.linalg.norm(u @ u.T - comps[:, :n_comps] @ comps[:, :n_comps].T) < 1e-6


def test_linear_estimator_is_exact_for_affine_functions():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((400, 6))
    w = np.arange(1.0, 7.0)
    y = x @ w + 0.5
    est = aso.PolyGradientEstimator(x, y, 1, 12)
    g = aso.create_grad_mat(est, x[:25])
    assert g.shape == (6, 25) and np.allclose(g, np.tile(w.reshape(-1, 1), (1, 25)), atol=1e-9)
    with pytest.raises(ValueError):
        aso.PolyGradientEstimator(x, y, 1, 7).grad_at(x[0])     # n_nbrs must exceed k + 1 (asserts :118-119)
    with pytest.raises(ValueError):
        aso.PolyGradientEstimator(x, y, 3, 40)                  # the reference panics on other orders (:60)


def test_quadratic_gradient_matches_analytic_gradient_of_a_quadratic():
    rng = np.random.default_rng(5)
    k = 4
    x = rng.standard_normal((300, k))
    q = rng.standard_normal((k, k))
    q = q + q.T
    b = rng.standard_normal(k)
    y = 0.5 * np.einsum("ni,ij,nj->n", x, q, x) + x @ b + 3.75   # build_vandermonde carries the constant column
    est = aso.PolyGradientEstimator(x, y, 2, 40)
    for x0 in x[:5]:
        # forward differences with eps = 1e-10: rounding noise ~ |y| * 2e-16 / 1e-10
        assert np.allclose(est.grad_at(x0).ravel(), q @ x0 + b, rtol=0, atol=1e-4)
    est.exact_quad_gradient = True
    for x0 in x[:5]:
        assert np.allclose(est.grad_at(x0).ravel(), q @ x0 + b, rtol=0, atol=1e-9)


def test_vandermonde_layout_of_the_reference():
    """stats_corr.rs:112-143, 198-207: [x, x_a x_b (a <= b, a-major), 1]; linear_fit's design is [x, 1] (:146-159)."""
    x = np.array([[2.0, 3.0, 5.0]])
    assert aso.build_vandermonde(x, True).tolist() == [[2, 3, 5, 4, 6, 10, 9, 15, 25, 1]]
    assert aso.build_vandermonde(x, False).tolist() == [[2, 3, 5, 6, 10, 15, 1]]
