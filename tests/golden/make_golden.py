#!/usr/bin/env python3
"""Generate the committed golden fixtures in tests/golden/*.npz.

Runs ONLY in the authoring container (needs /root/reference); the fixtures it
writes are plain data (inputs + expected outputs) and travel with the repo.

Sources of expected values
  ref_numpy : the reference author's own numpy restatement
              examples/benchmark_rsvd.py:16-54 (``power_iteration`` / ``rsvd``),
              imported from /root/reference with an empty stub for the
              ``corrla_rs`` extension module it imports at :13 (the Rust
              module cannot be built here: no cargo/rustc).  Omega is shared by
              patching ``np.random.randn`` (the draw at :46).
  exact     : numpy.linalg.svd of A (truth for singular values / relerr).
  known     : the constants of test_rsvd_lowrank, random_svd.rs:155-168.

SURVEY.md section 8c: the numpy restatement differs from the Rust schedule
(no per-iteration rescale, no in-loop QR, l uncapped); on flat spectra / low q
both agree to rounding, so `ref_numpy` outputs are recorded only for such
cases.  Decaying-spectrum / q>3 cases record `exact` only.
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/examples/benchmark_rsvd.py"


def load_reference_numpy_rsvd():
    sys.modules.setdefault("corrla_rs", types.ModuleType("corrla_rs"))
    spec = importlib.util.spec_from_file_location("ref_benchmark_rsvd", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)  # __main__ guard at :57 keeps the benchmark from running
    return mod


def ref_rsvd_shared_omega(mod, a, k, q, p, omega):
    """Call the reference's rsvd(A, omega_rank, n_oversamples, power_iter)
    (note ITS argument order, benchmark_rsvd.py:26) with our Omega."""
    orig = np.random.randn
    try:
        def fake_randn(*shape):
            assert tuple(shape) == omega.shape, (shape, omega.shape)
            return omega.astype(np.float64)
        np.random.randn = fake_randn
        u, s, vt = mod.rsvd(a.astype(np.float64), omega_rank=k, n_oversamples=p, power_iter=q)
    finally:
        np.random.randn = orig
    return u, s, vt


def make_cases():
    rng = np.random.default_rng(20241008)
    cases = {}

    def gauss(m, n):
        return rng.standard_normal((m, n))

    def decaying(m, n, r, base, noise):
        g1 = np.linalg.qr(rng.standard_normal((m, r)))[0]
        g2 = np.linalg.qr(rng.standard_normal((n, r)))[0]
        s = base ** np.arange(r)
        return (g1 * s) @ g2.T + noise * rng.standard_normal((m, n))

    # name: (A, k, q, p, use_ref_numpy)
    known = np.array([[1, 0, 0, 0, 2], [0, 0, 3, 0, 0], [0, 0, 0, 0, 0], [0, 0, 0, 0, 0], [0, 2, 0, 0, 0]], float)
    cases["known5x5_k5"] = (known, 5, 12, 10, False)   # random_svd.rs:170-182
    cases["known5x5_k3"] = (known, 3, 12, 10, False)   # random_svd.rs:184-195
    cases["tall64x48"] = (gauss(64, 48), 8, 2, 10, True)
    cases["fat48x64"] = (gauss(48, 64), 8, 2, 10, True)
    cases["square40_lcap"] = (gauss(40, 40), 36, 2, 10, True)      # l = min(46, 40) = 40 == n
    cases["gauss512x256"] = (gauss(512, 256), 16, 3, 8, True)
    cases["gauss300x70_q0"] = (gauss(300, 70), 10, 0, 10, True)    # no power iterations
    cases["lowrank256x96"] = (decaying(256, 96, 24, 0.9, 1e-3), 12, 6, 10, False)   # q>3: in-loop QR
    cases["rankdef96x40"] = ((gauss(96, 6) @ gauss(6, 40)), 10, 4, 6, False)        # exact rank 6 < l=16
    cases["fat20x500_pod"] = (decaying(500, 20, 20, 0.7, 0.0).T.copy(), 4, 10, 10, False)  # POD shape, pod_rom.rs:56
    return cases, rng


def main():
    mod = load_reference_numpy_rsvd()
    # sanity: the reference numpy oracle reproduces the Rust known answers
    u, s, vt = mod.rsvd(np.array([[1, 0, 0, 0, 2], [0, 0, 3, 0, 0], [0, 0, 0, 0, 0], [0, 0, 0, 0, 0], [0, 2, 0, 0, 0]], float),
                        omega_rank=3, n_oversamples=2, power_iter=2)
    assert np.allclose(s, [3.0, 2.2360679, 2.0], atol=1e-6), s

    cases, rng = make_cases()
    for name, (a, k, q, p, use_ref) in cases.items():
        m0, n0 = a.shape
        ncols = min(m0, n0)
        l = min(k + p, ncols)
        omega = rng.standard_normal((ncols, l))
        out = dict(A=a, omega=omega, k=k, q=q, p=p)
        out["exact_s"] = np.linalg.svd(a, compute_uv=False)
        if use_ref:
            # the reference numpy code does not cap l; only record when k+p <= ncols
            # or emulate the cap by shrinking p (same Omega width)
            p_eff = l - k
            u, s, vt = ref_rsvd_shared_omega(mod, a, k, q, p_eff, omega)
            out["ref_u"], out["ref_s"], out["ref_vt"] = u, s, vt
            rec = (u * s) @ vt
            out["ref_relerr"] = np.linalg.norm(a - rec) / np.linalg.norm(a)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(f"{name}: A{a.shape} k={k} q={q} p={p} l={l} ref={use_ref}")

    # GEMM-shim known answers, mat_utils.rs:642-684 (2x2 identity cases)
    np.savez_compressed(
        os.path.join(HERE, "matmul_known.npz"),
        lhs=np.eye(2), rhs_vec=np.array([[3.0], [2.0]]), out_vec=np.array([[3.0], [2.0]]),
        rhs_mat=np.array([[3.0, 0.0], [2.0, 0.0]]), out_mat=np.array([[3.0, 0.0], [2.0, 0.0]]),
    )


if __name__ == "__main__":
    main()
