"""N > 1 path on CPU: world_size-2 gloo run of the row-sharded random_svd.  The
sharded result must equal the single-rank result on the same A and Omega to rounding, S / Vt must be
replicated, and the number of exchanges must be what DESIGN.md states."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from oracle import rsvd_oracle as orc
from tests.emu_harness import emu_rsvd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(300)
def test_row_sharded_world2_matches_single_rank():
    rng = np.random.default_rng(42)
    m, n, k, q, p = 301, 64, 10, 4, 6          # odd m -> uneven shards; q=4 -> one in-loop QR (i > 2)
    a = rng.standard_normal((m, n))
    l = min(k + p, n)
    omega = rng.standard_normal((n, l))
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "input.npz"), A=a, omega=omega, k=k, q=q, p=p)
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
               "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "tests", "_sharded_worker.py"), td]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        for dtype, tol in ((np.float64, 1e-10), (np.float32, 5e-5)):
            name = np.dtype(dtype).name
            outs = [np.load(os.path.join(td, f"out_{name}_rank{r_}.npz")) for r_ in range(2)]
            u = np.vstack([o["u"] for o in outs])
            s, vt = outs[0]["s"], outs[0]["vt"]
            assert np.array_equal(outs[0]["s"], outs[1]["s"]) and np.array_equal(outs[0]["vt"], outs[1]["vt"])
            u1, s1, vt1 = emu_rsvd(a.astype(dtype), k, q, p, omega=omega.astype(dtype))
            assert np.max(np.abs(s - s1)) <= tol * s1[0, 0]
            rec = (u.astype(np.float64) * s.ravel()) @ vt
            rec1 = (u1.astype(np.float64) * s1.ravel()) @ vt1
            assert np.linalg.norm(rec - rec1) <= 100 * tol * np.linalg.norm(rec1)
            uo, so, vto = orc.random_svd(a.astype(dtype), k, q, p, omega=omega.astype(dtype))
            assert abs(orc.relerr(a, u, s, vt) - orc.relerr(a, uo, so, vto)) <= 1e-5
            # exchanges per call: q + 1 all-reduces of the n x l factors (Z per iteration, B^T), one l x l Gram
            # all-reduce per in-loop orthonormalisation (i > 2, single pass) and two for the final thin-Q (well-conditioned
            # input: the context never had to enqueue conditional passes, whose Gram all-reduces would be unconditional).
            # No scalar all-reduce: the per-iteration rescale uses ||Z||_F, and Z is already replicated.
            # + 1: the handshake at the start of every sharded call (status + schedule state, one MAX all-reduce of 8 words)
            n_ar = int(outs[0]["n_allreduce"])
            assert n_ar == 1 + (q + 1) + max(0, q - 3) + 2


@pytest.mark.timeout(300)
def test_row_sharded_rank_deficient_with_a_shard_shorter_than_l():
    """A rank-deficient matrix sends every rank through the host-controlled path and `complete_basis`, which issues
    all-reduces: the decision to enter it must be rank-invariant.  Here rank 0 holds 10 rows < l = 16 and rank 1 holds
    40, so a test on the LOCAL row count would split the ranks (and hang the collectives); the driver uses the global
    row count, all-reduced once."""
    rng = np.random.default_rng(7)
    m, n, k, q, p = 50, 30, 8, 5, 8
    a = rng.standard_normal((m, 4)) @ rng.standard_normal((4, n))      # exact rank 4 < l = 16
    omega = rng.standard_normal((n, k + p))
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "input.npz"), A=a, omega=omega, k=k, q=q, p=p, splits=np.array([0, 10, 50]))
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29543")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
               "127.0.0.1", "--master-port", "29543", os.path.join(ROOT, "tests", "_sharded_worker.py"), td]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        ex = np.linalg.svd(a, compute_uv=False)[:k]
        for dtype, tol in ((np.float64, 1e-9), (np.float32, 2e-4)):
            name = np.dtype(dtype).name
            outs = [np.load(os.path.join(td, f"out_{name}_rank{r_}.npz")) for r_ in range(2)]
            assert int(outs[0]["hi"]) - int(outs[0]["lo"]) == 10 and int(outs[1]["hi"]) - int(outs[1]["lo"]) == 40
            u = np.vstack([o["u"] for o in outs]).astype(np.float64)
            s, vt = outs[0]["s"].astype(np.float64), outs[0]["vt"].astype(np.float64)
            assert np.array_equal(outs[0]["s"], outs[1]["s"]) and np.array_equal(outs[0]["vt"], outs[1]["vt"])
            assert np.allclose(s.ravel(), ex, atol=tol * ex[0])
            assert np.max(np.abs(u.T @ u - np.eye(k))) < 100 * tol        # orthonormal completion across the shards
            assert np.max(np.abs(vt @ vt.T - np.eye(k))) < 100 * tol
            assert np.linalg.norm((u * s.ravel()) @ vt - a) <= 100 * tol * np.linalg.norm(a)


@pytest.mark.timeout(300)
def test_column_sharded_fat_matrix_world2():
    """CORRLA_SHARD_COLS: a FAT matrix sharded along its long side (columns).  The reference works on the tall view A^T
    (random_svd.rs:69-74); every rank passes its m x n_local block, gets U and S replicated and its own columns of Vt.
    Uneven shards; result equal to the single-rank call on the whole fat matrix with the same Omega (m x l)."""
    rng = np.random.default_rng(11)
    m, n, k, q, p = 40, 301, 8, 4, 6
    a = rng.standard_normal((m, n)) * (0.93 ** np.arange(m))[:, None]
    omega = rng.standard_normal((m, k + p))
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "input.npz"), A=a, omega=omega, k=k, q=q, p=p, splits=np.array([0, 100, 301]), shard_cols=1)
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29545")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
               "127.0.0.1", "--master-port", "29545", os.path.join(ROOT, "tests", "_sharded_worker.py"), td]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        for dtype, tol in ((np.float64, 1e-10), (np.float32, 5e-5)):
            name = np.dtype(dtype).name
            outs = [np.load(os.path.join(td, f"out_{name}_rank{r_}.npz")) for r_ in range(2)]
            assert outs[0]["u"].shape == (m, k) and outs[0]["vt"].shape == (k, 100) and outs[1]["vt"].shape == (k, 201)
            assert np.array_equal(outs[0]["u"], outs[1]["u"]) and np.array_equal(outs[0]["s"], outs[1]["s"])
            u, s = outs[0]["u"], outs[0]["s"]
            vt = np.hstack([o["vt"] for o in outs])
            u1, s1, vt1 = emu_rsvd(a.astype(dtype), k, q, p, omega=omega.astype(dtype))
            assert np.max(np.abs(s - s1)) <= tol * s1[0, 0]
            rec = (u.astype(np.float64) * s.ravel()) @ vt
            rec1 = (u1.astype(np.float64) * s1.ravel()) @ vt1
            assert np.linalg.norm(rec - rec1) <= 100 * tol * np.linalg.norm(rec1)
            uo, so, vto = orc.random_svd(a.astype(dtype), k, q, p, omega=omega.astype(dtype))
            assert abs(orc.relerr(a, u, s, vt) - orc.relerr(a, uo, so, vto)) <= 1e-5


@pytest.mark.timeout(300)
def test_sample_sharded_pca_world2():
    """corrla_pca_sharded_*: samples sharded over two ranks (uneven), fused and copy centring: means all-reduced over
    the global sample count, the rank-1 corrections of the fused form applied per rank before each all-reduce.  Result
    equal to the single-rank PCA of the whole matrix and to the oracle (pca_rsvd.rs:56-82)."""
    from tests.emu_harness import emu_pca
    rng = np.random.default_rng(21)
    m, n, k, q, p = 211, 24, 5, 4, 8
    x = rng.standard_normal((m, n)) * (0.8 ** np.arange(n)) + rng.standard_normal((1, n)) * 2.0
    omega = rng.standard_normal((n, k + p))
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "input.npz"), A=x, omega=omega, k=k, q=q, p=p, splits=np.array([0, 60, 211]), pca=1)
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
               "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "tests", "_sharded_worker.py"), td]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        for dtype, tol in ((np.float64, 1e-9), (np.float32, 2e-4)):
            name = np.dtype(dtype).name
            m1, s1, c1 = emu_pca(x.astype(dtype), k, q, p, omega=omega.astype(dtype), center="copy")
            for center in ("fused", "copy"):
                outs = [np.load(os.path.join(td, f"pca_{name}_{center}_rank{r_}.npz")) for r_ in range(2)]
                for f in ("means", "s", "comps"):
                    assert np.array_equal(outs[0][f], outs[1][f]), (center, f)       # replicated
                assert np.allclose(outs[0]["means"], x.mean(axis=0, keepdims=True), atol=tol * 10)
                assert np.allclose(outs[0]["s"], s1, rtol=tol, atol=tol * s1[0, 0])
                cg = outs[0]["comps"].astype(np.float64)
                assert np.linalg.norm(cg.T @ cg - c1.astype(np.float64).T @ c1.astype(np.float64)) < 200 * tol


@pytest.mark.timeout(300)
@pytest.mark.parametrize("case", ["one_panel", "column_blocks", "rank_deficient"])
def test_row_sharded_householder_tsqr_world2(case):
    """CORRLA_QR_HOUSEHOLDER on the row-sharded entry point (SURVEY 8e, R-factor exchange): every rank reduces its rows
    by TSQR, the two root R factors are stacked by one all-reduce, the thin-Q of the stack is taken redundantly and each
    rank applies its block of it on the way down.  `column_blocks`: a sketch wider than one LDS panel (l = 110 > 97 in
    f64) goes through column blocks with two projection + panel passes each.  `rank_deficient`: exact rank 4 < l, the
    Householder Q is orthonormal whatever the rank (random_svd.rs:38,57), across the shards too."""
    rng = np.random.default_rng(5)
    if case == "one_panel":
        m, n, k, q, p = 301, 64, 10, 4, 6
        a = rng.standard_normal((m, n))
        port = "29549"
    elif case == "column_blocks":
        m, n, k, q, p = 700, 130, 100, 4, 10
        a = rng.standard_normal((m, n)) * (0.998 ** np.arange(n))
        port = "29551"
    else:
        m, n, k, q, p = 90, 30, 8, 5, 8
        a = rng.standard_normal((m, 4)) @ rng.standard_normal((4, n))
        port = "29553"
    l = min(k + p, n)
    omega = rng.standard_normal((n, l))
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "input.npz"), A=a, omega=omega, k=k, q=q, p=p, householder=1,
                 splits=np.array([0, m // 3, m]))
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
               "127.0.0.1", "--master-port", port, os.path.join(ROOT, "tests", "_sharded_worker.py"), td]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        ex = np.linalg.svd(a, compute_uv=False)
        for dtype, tol in ((np.float64, 1e-10), (np.float32, 5e-5)):
            name = np.dtype(dtype).name
            outs = [np.load(os.path.join(td, f"out_{name}_rank{r_}.npz")) for r_ in range(2)]
            u = np.vstack([o["u"] for o in outs]).astype(np.float64)
            s, vt = outs[0]["s"].astype(np.float64), outs[0]["vt"].astype(np.float64)
            assert np.array_equal(outs[0]["s"], outs[1]["s"]) and np.array_equal(outs[0]["vt"], outs[1]["vt"])
            assert np.max(np.abs(u.T @ u - np.eye(k))) < 100 * tol
            assert np.max(np.abs(vt @ vt.T - np.eye(k))) < 100 * tol
            if case == "rank_deficient":
                assert np.allclose(s.ravel()[:4], ex[:4], rtol=100 * tol) and np.all(s.ravel()[4:] < 100 * tol * ex[0])
                assert np.linalg.norm((u * s.ravel()) @ vt - a) <= 100 * tol * np.linalg.norm(a)
                continue
            # same factorisation as the single-rank Householder call and as the oracle
            u1, s1, vt1 = emu_rsvd(a.astype(dtype), k, q, p, omega=omega.astype(dtype), qr="householder")
            assert np.max(np.abs(s - s1)) <= tol * s1[0, 0]
            rec = (u * s.ravel()) @ vt
            rec1 = (u1.astype(np.float64) * s1.ravel()) @ vt1
            assert np.linalg.norm(rec - rec1) <= 100 * tol * np.linalg.norm(rec1)
            uo, so, vto = orc.random_svd(a.astype(dtype), k, q, p, omega=omega.astype(dtype))
            assert abs(orc.relerr(a, u, s, vt) - orc.relerr(a, uo, so, vto)) <= 1e-5


def _run_world(td, world, port, extra_env=None, per_rank_env=None):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", port, os.path.join(ROOT, "tests", "_sharded_worker.py"), td]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)


@pytest.mark.timeout(300)
def test_row_sharded_world4_with_an_empty_shard():
    """More ranks than row blocks: rank 1 holds NO rows (m_local = 0).  It takes part in every collective with zero
    contributions and returns a 0 x k block of U; the other three ranks' blocks stack to the single-rank result."""
    rng = np.random.default_rng(17)
    m, n, k, q, p = 180, 48, 9, 4, 7
    a = rng.standard_normal((m, n)) * (0.97 ** np.arange(n))
    omega = rng.standard_normal((n, k + p))
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "input.npz"), A=a, omega=omega, k=k, q=q, p=p, splits=np.array([0, 50, 50, 121, 180]))
        r = _run_world(td, 4, "29561")
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        for dtype, tol in ((np.float64, 1e-10), (np.float32, 5e-5)):
            name = np.dtype(dtype).name
            outs = [np.load(os.path.join(td, f"out_{name}_rank{r_}.npz")) for r_ in range(4)]
            assert outs[1]["u"].shape == (0, k)
            for o in outs[1:]:
                assert np.array_equal(outs[0]["s"], o["s"]) and np.array_equal(outs[0]["vt"], o["vt"])
            assert len({int(o["n_allreduce"]) for o in outs}) == 1          # every rank issued the same collectives
            u = np.vstack([o["u"] for o in outs])
            s, vt = outs[0]["s"], outs[0]["vt"]
            u1, s1, vt1 = emu_rsvd(a.astype(dtype), k, q, p, omega=omega.astype(dtype))
            assert np.max(np.abs(s - s1)) <= tol * s1[0, 0]
            rec = (u.astype(np.float64) * s.ravel()) @ vt
            rec1 = (u1.astype(np.float64) * s1.ravel()) @ vt1
            assert np.linalg.norm(rec - rec1) <= 100 * tol * np.linalg.norm(rec1)
            uo, so, vto = orc.random_svd(a.astype(dtype), k, q, p, omega=omega.astype(dtype))
            assert abs(orc.relerr(a, u, s, vt) - orc.relerr(a, uo, so, vto)) <= 1e-5


@pytest.mark.timeout(300)
def test_a_rank_local_failure_ends_the_call_on_every_rank_world4():
    """Rank 2 passes a bad leading dimension (ldu < m_local): without the handshake its three peers would sit in the
    first all-reduce for ever.  Every rank must return an error: the failing one its own, the others 'another rank
    failed'."""
    rng = np.random.default_rng(18)
    m, n, k, q, p = 120, 32, 6, 2, 6
    a = rng.standard_normal((m, n))
    omega = rng.standard_normal((n, k + p))
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "input.npz"), A=a, omega=omega, k=k, q=q, p=p, fail_rank=2)
        r = _run_world(td, 4, "29563")
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        outcomes = [open(os.path.join(td, f"outcome_rank{r_}.txt")).read() for r_ in range(4)]
        assert outcomes[2].startswith("ValueError") and "ldu < m" in outcomes[2], outcomes
        for r_ in (0, 1, 3):
            assert outcomes[r_].startswith("ValueError") and "another rank failed" in outcomes[r_], outcomes


@pytest.mark.timeout(300)
def test_ranks_with_different_context_histories_enqueue_the_same_collectives():
    """The number of thin-Q passes a call enqueues (each with a Gram all-reduce) is adaptive context state.  Rank 0
    starts at 8 passes (CORRLA_ROBUST_PASSES, standing for a context that an earlier ill-conditioned call escalated),
    rank 1 at the default 2: the handshake at the start of the call levels the state, so both ranks issue the same
    sequence of collectives (mismatched sequences hang or mis-pair buffers)."""
    rng = np.random.default_rng(19)
    m, n, k, q, p = 160, 40, 8, 5, 6
    a = rng.standard_normal((m, n)) * (0.9 ** np.arange(n))
    omega = rng.standard_normal((n, k + p))
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "input.npz"), A=a, omega=omega, k=k, q=q, p=p, escalate_rank=0)
        r = _run_world(td, 2, "29565")
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        for dtype, tol in ((np.float64, 1e-9), (np.float32, 2e-4)):
            name = np.dtype(dtype).name
            outs = [np.load(os.path.join(td, f"out_{name}_rank{r_}.npz")) for r_ in range(2)]
            assert int(outs[0]["n_allreduce"]) == int(outs[1]["n_allreduce"])
            assert np.array_equal(outs[0]["s"], outs[1]["s"]) and np.array_equal(outs[0]["vt"], outs[1]["vt"])
            u = np.vstack([o["u"] for o in outs])
            uo, so, vto = orc.random_svd(a.astype(dtype), k, q, p, omega=omega.astype(dtype))
            assert abs(orc.relerr(a, u, outs[0]["s"], outs[0]["vt"]) - orc.relerr(a, uo, so, vto)) <= 1e-5
