"""Host logic of the product (driver.hpp + capi_impl.hpp + small_linalg.hpp) exercised through the
test-only emulation backend, against the CPU oracle.  No GPU, no compute call into libcorrla_rsvd.so."""
import ctypes as C

import numpy as np
import pytest

from oracle import rsvd_oracle as orc
from tests.conftest import golden_names
from tests.emu_harness import emu, emu_fill_normal, emu_matmul, emu_pca, emu_power_iter, emu_rsvd
from tests.helpers import align_signs, check_factorization, load_golden, orth_err


def _compare(a, k, q, p, omega, dtype, s_rtol, rec_rtol):
    a = a.astype(dtype)
    om = omega.astype(dtype)
    u, s, vt = emu_rsvd(a, k, q, p, omega=om)
    uo, so, vto = orc.random_svd(a, k, q, p, omega=om)
    check_factorization(a, u, s, vt, k, 0)
    s1 = max(float(so[0, 0]), 1e-300)
    assert np.max(np.abs(s.ravel().astype(np.float64) - so.ravel())) <= s_rtol * s1
    # relerr parity: within 1e-5 of the CPU restatement on the same A and Omega
    assert abs(orc.relerr(a, u, s, vt) - orc.relerr(a, uo, so, vto)) <= 1e-5
    # rank-k reconstructions agree (sign-free)
    rec = (u.astype(np.float64) * s.ravel()) @ vt.astype(np.float64)
    reco = (uo.astype(np.float64) * so.ravel()) @ vto.astype(np.float64)
    assert np.linalg.norm(rec - reco) <= rec_rtol * max(np.linalg.norm(reco), 1e-300)
    nnz = int(np.sum(so.ravel() > 1e-5 * s1))
    eps = np.finfo(dtype).eps
    assert orth_err(u[:, :nnz]) <= 200 * eps * np.sqrt(a.shape[0])
    assert orth_err(vt[:nnz, :].T) <= 200 * eps * np.sqrt(a.shape[1])
    return u, s, vt


@pytest.mark.parametrize("name", golden_names())
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_emu_driver_vs_oracle_golden(name, dtype):
    g = load_golden(name)
    f64 = dtype == np.float64
    tight = name not in ("lowrank256x96", "fat20x500_pod", "rankdef96x40")
    s_rtol = (1e-10 if f64 else 2e-5) if tight else (1e-7 if f64 else 2e-3)
    rec_rtol = (1e-8 if f64 else 1e-3) if tight else (1e-6 if f64 else 2e-2)
    u, s, vt = _compare(g["A"], g["k"], g["q"], g["p"], g["omega"], dtype, s_rtol, rec_rtol)
    if name.startswith("known5x5"):
        # random_svd.rs:170-195
        assert np.allclose(s.ravel(), orc.KNOWN_ANSWER_S[: g["k"]], atol=1e-3)


@pytest.mark.parametrize("order", ["C", "F", "strided", "strided_cols"])
@pytest.mark.parametrize("shape", [(70, 33), (33, 70), (64, 64), (1, 9), (9, 1), (5, 5)])
def test_emu_layouts_and_shapes(order, shape):
    rng = np.random.default_rng(7)
    m, n = shape
    base = rng.standard_normal((2 * m, 2 * n))
    if order == "C":
        a = np.ascontiguousarray(base[:m, :n])
    elif order == "F":
        a = np.asfortranarray(base[:m, :n])
    elif order == "strided":
        a = base[::2, ::2]
    else:
        a = base[:m, ::2]
    k = max(1, min(m, n) // 3)
    q, p = 2, 4
    nt = min(m, n)
    l = min(k + p, nt)
    omega = rng.standard_normal((nt, l))
    u, s, vt = emu_rsvd(a, k, q, p, omega=omega)
    uo, so, vto = orc.random_svd(np.array(a), k, q, p, omega=omega)
    assert np.allclose(s, so, rtol=0, atol=1e-9 * so[0, 0])
    assert abs(orc.relerr(a, u, s, vt) - orc.relerr(a, uo, so, vto)) < 1e-9


def test_emu_power_iter_matches_oracle_range():
    rng = np.random.default_rng(3)
    a = rng.standard_normal((120, 40))
    om = rng.standard_normal((40, 12))
    for q in (0, 2, 5):
        qe = emu_power_iter(a, 12, q, omega=om)
        qo = orc.power_iter(a, om, q)
        assert orth_err(qe) < 1e-12
        # same range: projectors agree
        assert np.linalg.norm(qe @ qe.T - qo @ qo.T) < 1e-8


def test_emu_invalid_arguments():
    a = np.ones((6, 4))
    with pytest.raises(ValueError):
        emu_rsvd(a, 5, 1, 2)   # rank > min(m,n): reference panics (random_svd.rs:98-107)
    with pytest.raises(ValueError):
        emu_rsvd(a, 0, 1, 2)
    with pytest.raises(ValueError):
        emu_rsvd(a, 2, -1, 2)


def test_emu_zero_matrix():
    u, s, vt = emu_rsvd(np.zeros((12, 7)), 3, 2, 2, omega=np.ones((7, 5)))
    assert np.all(s == 0) and np.all(np.isfinite(u)) and np.all(np.isfinite(vt))


def test_emu_qr_passes_well_conditioned():
    # well-conditioned sketch: 2 Gram passes per orthonormalisation (Y and B^T) = 4, plus the single polishing pass
    # on the l x k factor that the core SVD returns as W / sigma
    rng = np.random.default_rng(5)
    a = rng.standard_normal((300, 80))
    *_, passes = emu_rsvd(a, 8, 2, 8, omega=rng.standard_normal((80, 16)), return_passes=True)
    assert passes == 5


def test_emu_matmul_known_answers():
    # mat_utils.rs:642-684
    d = np.load(__import__("os").path.join(__import__("tests.helpers", fromlist=["x"]).GOLDEN_DIR, "matmul_known.npz"))
    assert np.allclose(emu_matmul(d["lhs"], d["rhs_vec"], False), d["out_vec"], atol=1e-6)
    assert np.allclose(emu_matmul(d["lhs"], d["rhs_mat"], False), d["out_mat"], atol=1e-6)
    rng = np.random.default_rng(0)
    a = rng.standard_normal((37, 21))
    x = rng.standard_normal((37, 5))
    assert np.allclose(emu_matmul(a, x, True, beta=0.5), 0.5 * a.T @ x, atol=1e-12)
    assert np.allclose(emu_matmul(np.asfortranarray(a), x, True), a.T @ x, atol=1e-12)


def test_small_linalg_routines():
    e = emu()
    rng = np.random.default_rng(11)
    n = 37
    x = rng.standard_normal((90, n))
    g = np.asfortranarray(x.T @ x)
    r = g.copy(order="F")
    mr = C.c_double()
    ok = e.corrla_emu_chol_upper(n, r.ctypes.data_as(C.c_void_p), n, C.c_double(1e-14), C.byref(mr))
    assert ok == 1 and np.allclose(np.triu(r).T @ np.triu(r), g, atol=1e-10)
    assert np.allclose(np.tril(r, -1), 0)
    rinv = r.copy(order="F")
    e.corrla_emu_triu_inverse(n, rinv.ctypes.data_as(C.c_void_p), n)
    assert np.allclose(np.triu(rinv) @ np.triu(r), np.eye(n), atol=1e-9)
    # singular Gram -> Cholesky reports failure
    gs = np.asfortranarray(np.ones((4, 4)))
    assert e.corrla_emu_chol_upper(4, gs.ctypes.data_as(C.c_void_p), 4, C.c_double(1e-12), C.byref(mr)) == 0
    # Jacobi SVD incl. a rank-deficient matrix
    for c in (rng.standard_normal((n, n)), rng.standard_normal((n, 5)) @ rng.standard_normal((5, n))):
        cf = np.asfortranarray(c)
        u = np.empty((n, n), order="F"); v = np.empty((n, n), order="F"); s = np.empty(n)
        e.corrla_emu_jacobi_svd.restype = C.c_int
        sweeps = e.corrla_emu_jacobi_svd(n, cf.ctypes.data_as(C.c_void_p), n, u.ctypes.data_as(C.c_void_p),
                                         s.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), C.c_double(1e-15))
        assert 0 < sweeps < 40
        assert np.allclose((u * s) @ v.T, c, atol=1e-10)
        assert np.allclose(s, np.linalg.svd(c, compute_uv=False), atol=1e-10)
        assert np.all(np.diff(s) <= 1e-12)
        assert orth_err(v) < 1e-12


def test_fill_normal_is_counter_based_and_standard_normal():
    full = emu_fill_normal(64, 48, seed=20241008)
    # any row shard regenerates its rows bit-identically (SURVEY 8d synthetic inputs)
    shard = emu_fill_normal(16, 48, seed=20241008, row0=32, global_cols=48)
    assert np.array_equal(full[32:48], shard)
    colmajor = emu_fill_normal(64, 48, seed=20241008, order="F")
    assert np.array_equal(full, colmajor)
    big = emu_fill_normal(400, 500, seed=1)
    assert abs(big.mean()) < 0.01 and abs(big.std() - 1) < 0.01
    from scipy import stats
    assert stats.kstest(big.ravel()[:50000], "norm").pvalue > 1e-3
    assert not np.array_equal(emu_fill_normal(8, 8, seed=1), emu_fill_normal(8, 8, seed=2))


@pytest.mark.parametrize("shape", [(100, 10), (60, 90), (300, 40)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_emu_pca_fused_centring_equals_centred_copy(shape, dtype):
    """(A - 1 mu^T) X = A X - 1 (mu^T X) and its transpose, against the centred copy the reference
    forms (center_mat_col, mat_utils.rs:482-502); tall and fat inputs take different correction vectors."""
    rng = np.random.default_rng(sum(shape) + 1)
    m, n = shape
    x = (rng.standard_normal((m, n)) * (0.8 ** np.arange(n)) + rng.standard_normal((1, n)) * 0.5).astype(dtype)
    k, p = 4, min(n, 10)
    nt = min(m, n)
    omega = rng.standard_normal((nt, min(k + p, nt))).astype(dtype)
    mf, sf, cf = emu_pca(x, k, 20, p, omega=omega, center="fused")
    mc, sc, cc = emu_pca(x, k, 20, p, omega=omega, center="copy")
    f64 = dtype == np.float64
    assert np.array_equal(mf, mc)
    assert np.allclose(sf, sc, rtol=1e-10 if f64 else 2e-4)
    assert np.linalg.norm(cf.T.astype(np.float64) @ cf - cc.T.astype(np.float64) @ cc) < (1e-8 if f64 else 5e-3)
    with pytest.raises(ValueError):
        emu_pca(x, k, 20, p, omega=omega, center="both")


@pytest.mark.parametrize("shape", [(100, 10), (60, 90), (300, 40)])
@pytest.mark.parametrize("order", ["C", "F"])
def test_emu_pca_matches_oracle(shape, order):
    """PcaRsvd::new (pca_rsvd.rs:56-82) through the real glue: means, centring, random_svd(cx, k, 20, min(n,10))."""
    rng = np.random.default_rng(sum(shape))
    m, n = shape
    x = rng.standard_normal((m, n)) * (0.8 ** np.arange(n)) + rng.standard_normal((1, n)) * 3.0
    x = np.asfortranarray(x) if order == "F" else np.ascontiguousarray(x)
    k = 4
    p = min(n, 10)
    nt = min(m, n)
    omega = rng.standard_normal((nt, min(k + p, nt)))
    means, s, comps = emu_pca(x, k, 20, p, omega=omega)
    mo, so, co, ev = orc.pca_rsvd(x, k, omega=omega)
    assert np.allclose(means, mo, atol=1e-12)
    assert np.allclose(s, so, rtol=1e-9)
    proj, projo = comps.T @ comps, co.T @ co   # sign-free comparison of the component subspace
    assert np.linalg.norm(proj - projo) < 1e-7
    # explained variance matches the eigenvalues of the sample covariance (what PCA means)
    cov_eigs = np.sort(np.linalg.eigvalsh(np.cov(x, rowvar=False)))[::-1][:k]
    assert np.allclose((s * s / (m - 1.0)).ravel(), cov_eigs, rtol=1e-6)


def test_sign_convention_short_side_vector_largest_component_positive():
    rng = np.random.default_rng(12)
    for shape in ((80, 30), (30, 80)):
        a = rng.standard_normal(shape)
        u, s, vt = emu_rsvd(a, 5, 2, 5, omega=rng.standard_normal((30, 10)))
        short = vt.T if shape[0] >= shape[1] else u     # length min(m, n)
        for i in range(5):
            j = int(np.argmax(np.abs(short[:, i])))
            assert short[j, i] > 0


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_rank_deficient_input_still_gets_orthonormal_factors(dtype):
    """A Householder thin-Q (random_svd.rs:38,57) is orthonormal whatever the rank of the sketch; the directions the
    data does not determine are an arbitrary completion with zero singular values.  U and V^T of the reference's own
    rank-3 5x5 example (random_svd.rs:155-168) and of a rank-5 60x40 matrix must be orthonormal, not zero-padded."""
    rng = np.random.default_rng(8)
    cases = [(orc.KNOWN_ANSWER_A.astype(dtype), 5, 12, 10, orc.KNOWN_ANSWER_S),
             ((rng.standard_normal((60, 5)) @ rng.standard_normal((5, 40))).astype(dtype), 12, 2, 6, None),
             (np.zeros((30, 20), dtype=dtype), 4, 2, 3, None)]
    for a, k, q, p, s_known in cases:
        nt = min(a.shape)
        om = rng.standard_normal((nt, min(k + p, nt))).astype(dtype)
        u, s, vt = emu_rsvd(a, k, q, p, omega=om)
        tol = 1e-10 if dtype == np.float64 else 2e-4
        assert np.max(np.abs(u.T.astype(np.float64) @ u - np.eye(k))) < tol
        assert np.max(np.abs(vt.astype(np.float64) @ vt.T - np.eye(k))) < tol
        ex = np.linalg.svd(a.astype(np.float64), compute_uv=False)[:k]
        assert np.allclose(s.ravel(), ex, atol=(1e-9 if dtype == np.float64 else 2e-4) * max(ex[0], 1.0))
        if s_known is not None:
            assert np.allclose(s.ravel(), s_known, atol=1e-3)           # the reference's own assertion
        rec = (u.astype(np.float64) * s.ravel()) @ vt.astype(np.float64)
        assert np.linalg.norm(rec - a) <= (1e-9 if dtype == np.float64 else 1e-3) * max(np.linalg.norm(a), 1.0)


@pytest.mark.parametrize("l", [177, 190, 266])
def test_emu_blocked_cholesky_qr_for_wide_sketches(l):
    """176 < l <= 352 takes the 2 x 2 blocked device factor-and-invert (driver.hpp: orthonormalize_core); the result
    must match the oracle like every other width, and a rank-deficient input must fall back cleanly."""
    rng = np.random.default_rng(l)
    m, n = 420, 280
    a = rng.standard_normal((m, n)) * (0.995 ** np.arange(n))
    k, p = l - 10, 10
    om = rng.standard_normal((n, l))
    u, s, vt = emu_rsvd(a, k, 2, p, omega=om)
    uo, so, vto = orc.random_svd(a, k, 2, p, omega=om)
    assert np.max(np.abs(s - so)) <= 1e-10 * so[0, 0]
    assert abs(orc.relerr(a, u, s, vt) - orc.relerr(a, uo, so, vto)) <= 1e-9
    assert np.max(np.abs(u.T @ u - np.eye(k))) < 1e-11 and np.max(np.abs(vt @ vt.T - np.eye(k))) < 1e-11
    low = rng.standard_normal((m, 9)) @ rng.standard_normal((9, n))
    u, s, vt = emu_rsvd(low, k, 1, p, omega=om)
    ex = np.linalg.svd(low, compute_uv=False)[:k]
    assert np.allclose(s.ravel(), ex, atol=1e-9 * ex[0])
    assert np.max(np.abs(u.T @ u - np.eye(k))) < 1e-10


# ---- Householder TSQR thin-Q (CORRLA_QR_HOUSEHOLDER): same panel partition / pairwise tree / reverse application as
# csrc/tsqr_kernels.hpp, run by the emulation backend through the real driver --------------------------------------
@pytest.mark.parametrize("m,n,width", [(40, 12, 12), (700, 30, 17), (1301, 64, 40), (97, 48, 48), (5000, 20, 9)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_householder_tsqr_power_iter_spans_the_oracle_basis(m, n, width, dtype):
    """panel counts 1, 2, odd (pass-through nodes), many; Q is orthonormal and spans the oracle's Householder Q."""
    rng = np.random.default_rng(m + width)
    a = rng.standard_normal((m, n)).astype(dtype)
    om = rng.standard_normal((n, width)).astype(dtype)
    q = emu_power_iter(a, width, 2, omega=om, qr="householder")
    qo = orc.power_iter(a.astype(np.float64), om.astype(np.float64), 2)
    tol = 5e-5 if dtype == np.float32 else 1e-10
    assert np.max(np.abs(q.T.astype(np.float64) @ q - np.eye(width))) < tol
    assert np.linalg.norm(q.astype(np.float64) @ (q.T.astype(np.float64) @ qo) - qo) < tol * 50


def test_householder_tsqr_rsvd_matches_the_oracle_and_cholesky_path():
    rng = np.random.default_rng(11)
    a = (rng.standard_normal((900, 60)) * np.logspace(0, -6, 60)) @ rng.standard_normal((60, 60))
    om = rng.standard_normal((60, 22))
    uh, sh, vh = emu_rsvd(a, 12, 3, 10, omega=om, qr="householder")
    uc, sc, vc = emu_rsvd(a, 12, 3, 10, omega=om)
    uo, so, vo = orc.random_svd(a, 12, 3, 10, omega=om)
    assert np.allclose(sh, so, rtol=1e-9) and np.allclose(sh, sc, rtol=1e-9)
    assert abs(orc.relerr(a, uh, sh, vh) - orc.relerr(a, uo, so, vo)) < 1e-9
    assert np.max(np.abs(uh.T @ uh - np.eye(12))) < 1e-12 and np.max(np.abs(vh @ vh.T - np.eye(12))) < 1e-12


def test_householder_tsqr_is_orthonormal_for_rank_deficient_sketches():
    """A Householder thin-Q is orthonormal whatever the rank (random_svd.rs:38,57): exact rank 3, sketch width 10."""
    rng = np.random.default_rng(12)
    a = rng.standard_normal((400, 3)) @ rng.standard_normal((3, 40))
    q = emu_power_iter(a, 10, 1, omega=rng.standard_normal((40, 10)), qr="householder")
    assert np.max(np.abs(q.T @ q - np.eye(10))) < 1e-12
    u, s, vt = emu_rsvd(a, 6, 2, 4, omega=rng.standard_normal((40, 10)), qr="householder")
    assert np.allclose(s[:3, 0], np.linalg.svd(a, compute_uv=False)[:3], rtol=1e-10) and np.all(s[3:, 0] < 1e-10 * s[0, 0])
    assert np.max(np.abs(u.T @ u - np.eye(6))) < 1e-10


def test_householder_flag_on_the_sharded_entry_point():
    """Row-sharded calls run the cross-rank TSQR (world size 1 here: the stack of root R factors is the one R, its thin-Q
    a diagonal of signs): same factorisation as the unsharded Householder call and as the default path."""
    rng = np.random.default_rng(14)
    a = rng.standard_normal((300, 40))
    om = rng.standard_normal((40, 14))
    u1, s1, vt1 = emu_rsvd(a, 8, 2, 6, omega=om, sharded=True, qr="householder")
    u2, s2, vt2 = emu_rsvd(a, 8, 2, 6, omega=om, qr="householder")
    u0, s0, vt0 = emu_rsvd(a, 8, 2, 6, omega=om, sharded=True)
    for (u, s, vt) in ((u2, s2, vt2), (u0, s0, vt0)):
        assert np.allclose(s, s1, rtol=1e-12)
        assert np.linalg.norm((u * s.ravel()) @ vt - (u1 * s1.ravel()) @ vt1) <= 1e-11 * np.linalg.norm(a)
    assert np.max(np.abs(u1.T @ u1 - np.eye(8))) < 1e-13


@pytest.mark.parametrize("dtype,width", [(np.float64, 99), (np.float64, 100), (np.float64, 200), (np.float32, 143), (np.float32, 300)])
def test_householder_wider_than_one_panel_goes_through_column_blocks(dtype, width):
    """l > 142 (f32) / 99 (f64): one 2 l x l panel no longer fits in LDS; column blocks of at most one panel, each the
    thin-Q of (I - Q Q^T) Y_j taken twice around the Householder panels.  Orthonormal to O(eps) and spanning what the
    oracle's QR spans, also for a sketch of rank 5."""
    rng = np.random.default_rng(width)
    m, n = 900, 320
    eps = np.finfo(dtype).eps
    a = (rng.standard_normal((m, n)) * (0.995 ** np.arange(n))).astype(dtype)
    om = rng.standard_normal((n, width)).astype(dtype)
    q = emu_power_iter(a, width, 4, omega=om, qr="householder").astype(np.float64)
    assert np.max(np.abs(q.T @ q - np.eye(width))) < 200 * eps
    qo = orc.power_iter(a.astype(np.float64), om.astype(np.float64), 4)
    assert np.linalg.norm(qo - q @ (q.T @ qo)) <= (1e-9 if dtype == np.float64 else 2e-2) * np.sqrt(width)
    low = (rng.standard_normal((m, 5)) @ rng.standard_normal((5, n))).astype(dtype)
    q = emu_power_iter(low, width, 1, omega=om, qr="householder").astype(np.float64)
    assert np.max(np.abs(q.T @ q - np.eye(width))) < 200 * eps


@pytest.mark.parametrize("q", [0, 1, 2, 3, 5])
def test_emu_one_sweep_power_iteration_schedule(q):
    """SURVEY 8 f4 (CORRLA_POWER_FUSED): Z = A^T (A X) in one product wherever Y is not needed (the sketch and the
    iterations without the in-loop thin-Q), the reference's own steps from there on.  Same factorisation as the
    two-product schedule and as the oracle, for q below / at / above the i > 2 boundary."""
    rng = np.random.default_rng(q)
    m, n, k, p = 4200, 96, 10, 6          # >= 4096 rows: the fused kernel's domain; f32, row-major
    a = (rng.standard_normal((m, n)) * (0.95 ** np.arange(n))).astype(np.float32)
    om = rng.standard_normal((n, k + p)).astype(np.float32)
    u1, s1, vt1 = emu_rsvd(a, k, q, p, omega=om, fused=True)
    u0, s0, vt0 = emu_rsvd(a, k, q, p, omega=om, fused=False)
    uo, so, vto = orc.random_svd(a.astype(np.float64), k, q, p, omega=om.astype(np.float64))
    assert np.max(np.abs(s1 - s0)) <= 2e-5 * s0[0, 0] and np.max(np.abs(s1 - so)) <= 2e-5 * so[0, 0]
    assert abs(orc.relerr(a, u1, s1, vt1) - orc.relerr(a, uo, so, vto)) <= 1e-5
    assert orth_err(u1) < 2e-4 and orth_err(vt1.T) < 2e-4
    # f64 and column-major inputs ignore the flag (outside the kernel's domain): identical to the plain schedule
    a64 = a.astype(np.float64)
    assert np.array_equal(emu_rsvd(a64, k, q, p, omega=om.astype(np.float64), fused=True)[1],
                          emu_rsvd(a64, k, q, p, omega=om.astype(np.float64), fused=False)[1])
    af = np.asfortranarray(a)
    assert np.array_equal(emu_rsvd(af, k, q, p, omega=om, fused=True)[1], emu_rsvd(af, k, q, p, omega=om, fused=False)[1])
