"""CPU oracle for the RSVD hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a numpy restatement of the reference algorithm
(wgurecky/CORRLA_RS @ 2024_10_08).  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / reported baseline.  The product path
(``corrla_rs_amd`` -> ``libcorrla_rsvd.so`` -> HIP kernels) never imports,
links or executes anything in ``oracle/``.

Parity pin (see tests/test_oracle_golden.py, tests/golden/make_golden.py):
  * the reference's only known-answer test, ``test_rsvd_lowrank``
    (src/lib_math_utils/random_svd.rs:153-196): S of the fixed 5x5 matrix is
    (3, 2.2360679, 2, 0, 0) to 1e-3;
  * outputs of the reference author's own numpy restatement
    (examples/benchmark_rsvd.py:16-54), imported in the authoring container
    with a stub ``corrla_rs`` and a shared Omega, stored as fixtures under
    tests/golden/ (flat-spectrum, low-q cases where both schedules agree to
    rounding -- SURVEY.md section 8c);
  * exact ``numpy.linalg.svd`` truth for the decaying-spectrum cases.

The Rust/faer reference itself cannot be built here (no cargo/rustc, crates
not vendored).  Third-party arithmetic it relies on: faer 0.19.x matmul /
Householder QR (thin Q) / SVD / norm_l2; here numpy/LAPACK supplies the same
mathematical operations.

Each function cites the reference lines it follows (paths relative to the
reference checkout).
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "thin_q",
    "power_iter",
    "random_svd",
    "rsvd",
    "pca_rsvd",
    "relerr",
    "algorithmic_flops",
    "KNOWN_ANSWER_A",
    "KNOWN_ANSWER_S",
]

# src/lib_math_utils/random_svd.rs:155-168 -- the reference's only
# known-answer fixture for this path.
KNOWN_ANSWER_A = np.array(
    [
        [1.0, 0.0, 0.0, 0.0, 2.0],
        [0.0, 0.0, 3.0, 0.0, 0.0],
        [0.0, 0.0, 0.0, 0.0, 0.0],
        [0.0, 0.0, 0.0, 0.0, 0.0],
        [0.0, 2.0, 0.0, 0.0, 0.0],
    ]
)
KNOWN_ANSWER_S = np.array([3.0, 2.2360679, 2.0, 0.0, 0.0])


def thin_q(y: np.ndarray) -> np.ndarray:
    """``y.qr().compute_thin_q()`` (random_svd.rs:38,57): Householder QR,
    explicit thin Q (m x l), R discarded.  numpy's 'reduced' mode is LAPACK
    geqrf + orgqr, i.e. the same Householder construction."""
    q, _ = np.linalg.qr(y, mode="reduced")
    return q


def power_iter(a: np.ndarray, omega: np.ndarray, n_iter: int) -> np.ndarray:
    """Range finder, random_svd.rs:15-59.

    ``omega`` (n x l) replaces ``random_mat_normal(a_ncols, omega_rank)``
    (random_svd.rs:24, mat_utils.rs:161-175) so the caller can share the
    sketch with the GPU run; the reference draws it unseeded.
    """
    a = np.asarray(a)
    omega = np.asarray(omega, dtype=a.dtype)
    assert omega.shape[0] == a.shape[1]
    y = a @ omega  # random_svd.rs:31
    for i in range(n_iter):  # :35
        if i > 2:  # :37-39  (QR only from the 4th iteration on)
            y = thin_q(y)
        z = a.T @ y  # :42-46  par_matmul_helper(o_mat_res, a^T, y)
        y = a @ z  # :47-51  par_matmul_helper(y, a, o_mat_res)
        # :53-55  y * (1 / y.norm_l2())  -- Frobenius norm of the matrix
        nrm = np.linalg.norm(y)
        y = y * (a.dtype.type(1.0) / a.dtype.type(nrm))
    return thin_q(y)  # :57


def random_svd(a: np.ndarray, omega_rank: int, n_iter: int, n_oversamples: int,
               omega: np.ndarray | None = None, rng: np.random.Generator | None = None):
    """``random_svd(a_mat, omega_rank, n_iter, n_oversamples)``,
    random_svd.rs:63-110.  Returns (U m0 x k, S k x 1, Vt k x n0) in the dtype
    of ``a`` (the reference is generic over f32/f64).

    ``omega``: optional (min(m0,n0)-side) x l sketch matrix, l =
    min(omega_rank + n_oversamples, ncols-after-transpose) (:77).
    """
    a = np.asarray(a)
    if a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float64)
    m0, n0 = a.shape
    fat = m0 < n0  # :71 (strict <: square counts as tall)
    aa = a.T if fat else a  # :73 (a view, no copy)
    ncols = aa.shape[1]
    l = min(omega_rank + n_oversamples, ncols)  # :77
    if omega_rank > l:
        # random_svd.rs:98-107: slicing 0..omega_rank past l panics
        raise ValueError("omega_rank exceeds min(m, n): the reference panics here")
    if omega is None:
        rng = rng or np.random.default_rng()
        omega = rng.standard_normal((ncols, l)).astype(a.dtype)
    omega = np.asarray(omega, dtype=a.dtype)
    if omega.shape != (ncols, l):
        raise ValueError(f"omega must be {(ncols, l)}, got {omega.shape}")
    q = power_iter(aa, omega, n_iter)  # :76-77
    b = q.T @ aa  # :80
    # :89  faer .svd() is the full SVD; only the first l columns of V are ever
    # read (:98-107 slice 0..omega_rank <= l), so the thin SVD returns the
    # same slices.
    ut, s, vt = np.linalg.svd(b, full_matrices=False)
    u = q @ ut  # :92
    k = omega_rank
    if fat:  # :96-102
        return vt[:k, :].T.copy(), s[:k].reshape(k, 1).copy(), u[:, :k].T.copy()
    return u[:, :k].copy(), s[:k].reshape(k, 1).copy(), vt[:k, :].copy()  # :103-109


def rsvd(a_mat: np.ndarray, n_rank: int, n_iters: int, n_oversamples: int, omega=None):
    """pyo3 surface ``corrla_rs.rsvd`` (src/lib_math_utils_py.rs:21-36):
    f64 only, positional order (rank, iters, oversamples), S returned 2-D
    (k x 1)."""
    a = np.asarray(a_mat, dtype=np.float64)
    return random_svd(a, n_rank, n_iters, n_oversamples, omega=omega)


def pca_rsvd(x_mat: np.ndarray, rank: int, omega=None):
    """``PcaRsvd::new(x_mat, rank)`` (src/lib_math_utils/pca_rsvd.rs:56-82): column means (mat_mean axis 1,
    mat_utils.rs:87-119), centred copy (center_mat_col, mat_utils.rs:482-502), then
    ``random_svd(cx, rank, 20, min(n_dim, 10))`` (pca_rsvd.rs:65-66).  Returns (means 1 x n, singular values
    k x 1, components k x n, explained variance k x 1 = s^2 / (n_samples - 1), pca_rsvd.rs:91-99).
    This is also what pyo3 ``rpca`` returns (it ignores its n_iters / n_oversamples, lib_math_utils_py.rs:39,48)."""
    x = np.asarray(x_mat, dtype=np.float64)
    n_samples, n_dim = x.shape
    means = x.mean(axis=0, keepdims=True)
    cx = x - means
    _u, s, vt = random_svd(cx, rank, 20, min(n_dim, 10), omega=omega)
    return means, s, vt, s * s / (n_samples - 1.0)


def relerr(a: np.ndarray, u: np.ndarray, s: np.ndarray, vt: np.ndarray) -> float:
    """||A - U diag(S) Vt||_F / ||A||_F in f64 (the accuracy gate of
    SURVEY.md section 8d; mirrors the reconstruction in random_svd.rs:143-147)."""
    a64 = np.asarray(a, dtype=np.float64)
    rec = (np.asarray(u, np.float64) * np.asarray(s, np.float64).reshape(1, -1)) @ np.asarray(vt, np.float64)
    return float(np.linalg.norm(a64 - rec) / np.linalg.norm(a64))


def algorithmic_flops(m: int, n: int, k: int, q: int, p: int) -> float:
    """(4q+4) m n l + 2 m l^2 + (1+max(0,q-3)) (4 m l^2 - 4/3 l^3),
    with m >= n after the fat->tall transpose and the UNPADDED l."""
    if m < n:
        m, n = n, m
    l = min(k + p, n)
    return (4 * q + 4) * m * n * l + 2.0 * m * l * l + (1 + max(0, q - 3)) * (4.0 * m * l * l - 4.0 / 3.0 * l ** 3)
