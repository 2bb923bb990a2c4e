"""Pins oracle/callers_oracle.py with the reference's own test_dmdc (dmd_rom.rs:233-310) and checks the POD /
active-subspace restatements against exact linear algebra.  CPU only."""
import numpy as np
import pytest

from oracle import callers_oracle as co


@pytest.mark.parametrize("nx,nt", [(20, 40), (50, 40), (500, 40)])   # dmd_rom.rs:235-240
def test_reference_test_dmdc_on_the_oracle(nx, nt):
    snaps, u = co.dmdc_reference_test_data(nx, nt)
    m = co.DMDcOracle(snaps, u, 1.0, 14, 40)
    assert m.est_a_til().shape == (nx, nx) and m.est_b_til().shape[0] == nx     # :281-283
    pred = m.predict_multiple(snaps[:, 0:1], u)
    assert m.lambdas.shape[0] == 14                                                # :308
    assert np.max(np.abs(pred[:, 19] - snaps[:, 20])) < 5e-2                       # :309


def test_pod_modes_span_the_dominant_right_singular_vectors():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((20, 6)) * [10, 7, 5, 1e-2, 1e-3, 1e-4]) @ rng.standard_normal((6, 300))
    modes = co.pod_modes(x, 3)
    assert modes.shape == (300, 3)
    vt = np.linalg.svd(x, full_matrices=False)[2][:3]
    assert np.linalg.norm(modes @ modes.T - vt.T @ vt) < 1e-6


def test_active_ss_fit_svd_matches_gram_eigendecomposition():
    # fit_svd (RSVD of G / sqrt(N)) and fit (EVD of G G^T / N, active_subspaces.rs:252-277) agree: s_i^2 = lambda_i
    rng = np.random.default_rng(1)
    g = (rng.standard_normal((8, 8)) * [5, 3, 2, 1, .1, .01, .001, .0001]) @ rng.standard_normal((8, 4000))
    u, sd = co.active_ss_fit_svd(g, 4)
    lam = np.sort(np.linalg.eigvalsh(g @ g.T / 4000.0))[::-1][:4]
    assert u.shape == (8, 4) and sd.shape == (4, 4)
    assert np.allclose(np.diag(sd) ** 2, lam, rtol=1e-8)
