"""Pins the CPU oracle (oracle/rsvd_oracle.py) against:
  * the reference's known-answer test test_rsvd_lowrank (random_svd.rs:153-196),
  * outputs of the reference author's numpy rsvd (examples/benchmark_rsvd.py:16-54)
    recorded with a shared Omega by tests/golden/make_golden.py,
  * exact SVD truth.
CPU only (no GPU, no /root/reference at run time)."""
import numpy as np
import pytest

from oracle import rsvd_oracle as orc
from tests.helpers import align_signs, check_factorization, load_golden, orth_err
from tests.conftest import golden_names


def test_known_answer_lowrank_k5_and_k3():
    # random_svd.rs:170-195: S == diag(3, 2.2360679, 2, 0, 0) to 1e-3
    rng = np.random.default_rng(0)
    for k in (5, 3):
        u, s, vt = orc.random_svd(orc.KNOWN_ANSWER_A, k, 12, 10, rng=rng)
        assert s.shape == (k, 1)
        assert np.allclose(s.ravel(), orc.KNOWN_ANSWER_S[:k], atol=1e-3)


def test_shape_contract_10000x100():
    # random_svd.rs:119-151 (test_rsvd_shape): 10000x100, k=4, q=12, p=10
    rng = np.random.default_rng(1)
    a = rng.standard_normal((10000, 100))
    u, s, vt = orc.random_svd(a, 4, 12, 10, rng=rng)
    rec = (u * s.ravel()) @ vt
    assert rec.shape == a.shape


@pytest.mark.parametrize("name", golden_names())
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_oracle_vs_golden(name, dtype):
    g = load_golden(name)
    a = g["A"].astype(dtype)
    k, q, p = g["k"], g["q"], g["p"]
    u, s, vt = orc.random_svd(a, k, q, p, omega=g["omega"].astype(dtype))
    assert u.dtype == dtype and s.dtype == dtype and vt.dtype == dtype
    check_factorization(a, u, s, vt, k, 1e-5)
    eps = np.finfo(dtype).eps
    s1 = g["exact_s"][0]
    if "ref_s" in g:
        # same Omega, flat spectrum / q<=3: the two schedules agree to rounding
        tol = 200 * eps * s1
        assert np.max(np.abs(s.ravel() - g["ref_s"])) <= tol
        re = orc.relerr(g["A"], u, s, vt)
        assert abs(re - float(g["ref_relerr"])) <= (1e-12 if dtype == np.float64 else 1e-5)
        u2, vt2 = align_signs(u, vt, g["ref_u"], g["ref_vt"])
        # vectors of well-separated singular values agree; compare the rank-k projector-free product
        rec = (u2 * s.ravel().astype(np.float64)) @ vt2
        rec_ref = (g["ref_u"] * g["ref_s"]) @ g["ref_vt"]
        assert np.linalg.norm(rec - rec_ref) <= (1e-9 if dtype == np.float64 else 2e-3) * np.linalg.norm(rec_ref)
    if name.startswith("known5x5"):
        assert np.allclose(s.ravel(), orc.KNOWN_ANSWER_S[:k], atol=1e-3)
    if name in ("lowrank256x96", "rankdef96x40", "fat20x500_pod"):
        # decaying spectrum, q>3: the Rust schedule (re-orthonormalised) matches exact SVD
        nz = g["exact_s"][:k]
        rtol = 1e-6 if dtype == np.float64 else 2e-3
        assert np.max(np.abs(s.ravel() - nz)) <= rtol * s1
    # orthonormality of the returned factors for the numerically non-zero part
    nnz = int(np.sum(s.ravel() > 1e-6 * s.ravel()[0]))
    assert orth_err(u[:, :nnz]) <= 100 * eps * np.sqrt(a.shape[0])
    assert orth_err(vt[:nnz, :].T) <= 100 * eps * np.sqrt(a.shape[1])


def test_pyo3_surface_rsvd_f64_only_and_arg_order():
    # lib_math_utils_py.rs:21-36: rsvd(a, n_rank, n_iters, n_oversamples), S is (k,1)
    g = load_golden("tall64x48")
    u, s, vt = orc.rsvd(g["A"], g["k"], g["q"], g["p"], omega=g["omega"])
    assert u.dtype == np.float64 and s.shape == (g["k"], 1)
    assert np.max(np.abs(s.ravel() - g["ref_s"])) < 1e-10
    # a non-f64 input is converted to f64 (PyReadonlyArray2<f64>), never computed in f32
    u32, s32, _ = orc.rsvd(g["A"].astype(np.float32), g["k"], g["q"], g["p"], omega=g["omega"])
    assert u32.dtype == np.float64 and np.max(np.abs(s32.ravel() - g["ref_s"])) < 1e-5


def test_rank_larger_than_min_dim_is_error():
    # random_svd.rs:98-107 panics when slicing 0..omega_rank past l
    with pytest.raises(ValueError):
        orc.random_svd(np.ones((6, 4)), 5, 1, 2)


def test_matmul_known_answers():
    # mat_utils.rs:642-684: res = lhs*rhs (alpha=None => overwrite, beta=1)
    import os
    from tests.helpers import GOLDEN_DIR
    d = np.load(os.path.join(GOLDEN_DIR, "matmul_known.npz"))
    assert np.allclose(d["lhs"] @ d["rhs_vec"], d["out_vec"], atol=1e-6)
    assert np.allclose(d["lhs"] @ d["rhs_mat"], d["out_mat"], atol=1e-6)


def test_algorithmic_flops_c2():
    # C2 = 4.464e11 (GEMM part 4.445e11)
    f = orc.algorithmic_flops(16384, 16384, 128, 2, 10)
    assert abs(f - 4.464e11) / 4.464e11 < 2e-3
