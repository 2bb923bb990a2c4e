"""GPU parity tests added in round 3 (run with -m gpu on an MI355X): non-finite cores through every core-SVD kernel
family, sketches wider than the device Cholesky serves (l > 352), the rank-deficient f32 cases the fuzz sweep of
round 2 left outside its bound, empty shards and the sharded handshake on a one-rank RCCL communicator."""
import numpy as np
import pytest

from oracle import rsvd_oracle as orc
from tests.helpers import check_factorization, orth_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import corrla_rs_amd as cr
    return cr.Context(0)


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


# ---- a non-finite l x l core must end the call with CORRLA_ENUMERIC, whatever kernel family takes its SVD ------------
# (round 2: a non-finite core made the final ranking of the Jacobi kernels keep uninitialised column indices -> a wild
#  read and an abort; fixed in the kernels, pinned here.  CORRLA_TEST_POISON_CORE overwrites entry (l/2, l/3) of the core
#  formed at random_svd.rs:89 with NaN (1) / +inf (2), AFTER both thin-Qs, so nothing upstream can catch it.)
_POISON_CASES = [
    # (l, environment that selects the kernel family)
    (24, {}),                                            # ring (W only) + replay, one wave
    (75, {}),                                            # ring (W only) + replay, odd pair count
    (75, {"CORRLA_JACOBI_NOREPLAY": "1"}),               # ring with V accumulated in the kernel
    (75, {"CORRLA_JACOBI_NORING": "1"}),                 # role-split LDS kernel
    (75, {"CORRLA_JACOBI_NORING": "1", "CORRLA_JACOBI_NOSPLIT": "1"}),  # LDS-resident kernel
    (96, {}),                                            # multi-workgroup block Jacobi, smallest
    (138, {}),                                           # ... the C2 width
    (138, {"CORRLA_JMC_FORCE_V": "1"}),                  # ... with V accumulated in the sweeps
    (266, {}),                                           # ... the C3 width (2 x 2 blocked Cholesky around it)
    (138, {"CORRLA_SVD": "lds"}),                        # ring kernel at a width the default gives to the block Jacobi
    (138, {"CORRLA_SVD": "lds", "CORRLA_JACOBI_NORING": "1"}),   # split kernel, widest
    (170, {"CORRLA_SVD": "lds"}),                        # LDS-resident kernel beyond the ring / split widths
    (138, {"CORRLA_SVD": "block"}),                      # one launch per round (MFMA block kernel)
    (300, {}),                                           # beyond the multi-workgroup geometry: block kernel
    (138, {"CORRLA_SVD": "host"}),                       # host Jacobi
    (138, {"CORRLA_DEVICE_ROBUST_QR": "0"}),             # round-1 optimistic CholeskyQR2 records around the same core
    (138, {"CORRLA_HOST_CHOL": "1"}),                    # host-controlled thin-Q (no deferred status at all)
]


@pytest.mark.parametrize("kind", [1, 2])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("case", range(len(_POISON_CASES)))
def test_non_finite_core_is_an_error_in_every_svd_kernel_family(ctx, monkeypatch, case, dtype, kind):
    from corrla_rs_amd._lib import CorrlaError
    l, env = _POISON_CASES[case]
    rng = np.random.default_rng(100 + case)
    m, n = max(3 * l, 200), l + 25
    a = rng.standard_normal((m, n)).astype(dtype)
    k, p = l - 8, 8
    om = rng.standard_normal((n, l)).astype(dtype)
    for name, val in env.items():
        monkeypatch.setenv(name, val)
    monkeypatch.setenv("CORRLA_TEST_POISON_CORE", str(kind))
    with pytest.raises(CorrlaError) as e:
        ctx.rsvd(a, k, 2, p, omega=om)
    assert e.value.code == 5, str(e.value)  # CORRLA_ENUMERIC
    # ... and the context is as good as new afterwards: same call without the poison, against the oracle
    monkeypatch.delenv("CORRLA_TEST_POISON_CORE")
    u, s, vt = ctx.rsvd(a, k, 2, p, omega=om)
    uo, so, vto = orc.random_svd(a, k, 2, p, omega=om)
    assert np.all(np.isfinite(u)) and np.all(np.isfinite(s)) and np.all(np.isfinite(vt))
    assert np.max(np.abs(s.ravel().astype(np.float64) - so.ravel())) <= (3e-5 if dtype == np.float32 else 1e-10) * so[0, 0]


# ---- sketches wider than the device Cholesky serves (l > 352): the host-controlled thin-Q, block / host core SVD ------
# random_svd.rs:76-77 puts no limit on l = min(rank + p, n); rank 512 of a few-thousand-column matrix is an ordinary call
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("m,n,k,p", [(1400, 640, 390, 10),      # l = 400
                                     (2048, 800, 502, 10),      # l = 512
                                     (2600, 1300, 1024, 10)])   # l = 1034: beyond the block Jacobi (host SVD)
def test_sketches_wider_than_352_columns(ctx, m, n, k, p, dtype):
    rng = np.random.default_rng(m + k)
    a = (rng.standard_normal((m, n)) * (0.999 ** np.arange(n))).astype(dtype)
    l = min(k + p, n)
    om = rng.standard_normal((n, l)).astype(dtype)
    u, s, vt = ctx.rsvd(a, k, 2, p, omega=om)
    uo, so, vto = orc.random_svd(a, k, 2, p, omega=om)
    check_factorization(a, u, s, vt, k, 0)
    f64 = dtype == np.float64
    s1 = float(so[0, 0])
    assert np.max(np.abs(s.ravel().astype(np.float64) - so.ravel())) <= (1e-10 if f64 else 2e-5) * s1
    assert abs(orc.relerr(a, u, s, vt) - orc.relerr(a, uo, so, vto)) <= 1e-5      # the north star's gate
    rec = (u.astype(np.float64) * s.ravel()) @ vt.astype(np.float64)
    reco = (uo.astype(np.float64) * so.ravel()) @ vto.astype(np.float64)
    assert np.linalg.norm(rec - reco) <= (1e-8 if f64 else 1e-3) * np.linalg.norm(reco)
    eps = np.finfo(dtype).eps
    assert orth_err(u) <= 200 * eps * np.sqrt(m) and orth_err(vt.T) <= 200 * eps * np.sqrt(n)


def test_wide_sketch_device_tensor_and_fat_orientation(ctx, torch):
    """l = 420 through the device-pointer entry, fat input (the tall view is A^T, random_svd.rs:69-74)."""
    rng = np.random.default_rng(77)
    m, n, k, p = 520, 1500, 410, 10
    a = rng.standard_normal((m, n))
    om = rng.standard_normal((m, k + p))
    u, s, vt = ctx.rsvd(torch.tensor(a, device="cuda"), k, 1, p, omega=om)
    u, s, vt = u.cpu().numpy(), s.cpu().numpy(), vt.cpu().numpy()
    uo, so, vto = orc.random_svd(a, k, 1, p, omega=om)
    assert np.max(np.abs(s - so)) <= 1e-10 * so[0, 0]
    assert abs(orc.relerr(a, u, s, vt) - orc.relerr(a, uo, so, vto)) <= 1e-9


# ---- empty shards and the handshake on a one-rank RCCL communicator ---------------------------------------------------
def test_sharded_handshake_and_empty_shard_world_size_1(torch, monkeypatch):
    """World size 1 with every collective issued to RCCL (CORRLA_FORCE_ALLREDUCE): the MAX all-reduce of the handshake
    at the start of a sharded call is a real ncclAllReduce; a rank-local argument error still comes back as this rank's
    own error; a call on an empty block is refused on a one-rank communicator only because the GLOBAL matrix is then
    empty (rank > rows), not by a crash."""
    import corrla_rs_amd as cr
    c = cr.Context(0)
    c.comm_init(cr.Context.unique_id(), 0, 1)
    monkeypatch.setenv("CORRLA_FORCE_ALLREDUCE", "1")
    g = torch.Generator(device="cuda").manual_seed(3)
    a = torch.randn((2000, 96), dtype=torch.float64, device="cuda", generator=g)
    om = np.random.default_rng(5).standard_normal((96, 20))
    u1, s1, vt1 = c.rsvd_sharded(a, 12, 2, 8, omega=om)
    n_coll = c.timings()["n_collectives"]
    assert n_coll == 1 + 3 + 2          # handshake + (q + 1) n x l all-reduces + two Gram all-reduces of the final thin-Q
    u0, s0, vt0 = c.rsvd(a, 12, 2, 8, omega=om)
    assert torch.equal(s0, s1) and torch.equal(u0, u1) and torch.equal(vt0, vt1)
    with pytest.raises(ValueError):
        c.rsvd_sharded(a, 200, 2, 8)        # rank > n: the local error, not 'another rank failed'
    # an empty block on the only rank: the stand-in zero row keeps every kernel launch well formed; the matrix has rank 0,
    # so S = 0 and the call completes (orthonormal completion), no fault
    e = torch.empty((0, 96), dtype=torch.float64, device="cuda")
    ue, se, vte = c.rsvd_sharded(e, 12, 2, 8, omega=om)
    assert ue.shape == (0, 12) and float(se.abs().max()) == 0.0 and bool(torch.isfinite(vte).all())
    c.close()


# ---- the rank-deficient f32 cases round 2's fuzz sweep left outside its 2e-3 bound --------------------------------------
# All five are exactly rank-deficient f32 inputs with q = 3: the reference schedule runs its first three power iterations
# without any re-orthonormalisation (random_svd.rs:37), the sketch's columns spread like (sigma_1 / sigma_r)^7 > 1 / eps_f32,
# and the trailing directions of range(A) are rounding noise in EVERY f32 implementation of the schedule -- including the
# reference's own arithmetic: the CPU restatement run in f32 (LAPACK Householder QR, exactly the reference's algorithm) is
# itself 3e-3 .. 9e-3 away from the f64 oracle in relerr on these inputs (measured: gpurun_out/r03_batch1/replay.log,
# DESIGN section 4).  The sweep compared f32 results with the f64 oracle at 2e-3, which no f32 run can hold here; what CAN
# be held, and is tested, is that the GPU path loses no more than the reference algorithm does in the same arithmetic:
#   deviation_gpu(f64 oracle)  <=  max(2e-3, 3 x deviation_cpu_f32(f64 oracle))  for S,   2 x  for relerr.
# (The Householder thin-Q does NOT help: measured 1.3 - 1.8 x further from the oracle than the default path on every
#  one of these cases; 61:757 and 62:373 are the two the sweep had drawn in Householder mode and run in it here too.)
_RANKDEF_CASES = [(12, 416, False), (32, 420, True), (51, 426, True), (61, 757, False), (62, 373, False)]


@pytest.mark.parametrize("seed,want,wide", _RANKDEF_CASES)
def test_rank_deficient_f32_fuzz_cases_lose_no_more_than_the_reference_algorithm_in_f32(ctx, seed, want, wide):
    from tools.fuzz_parity import cases
    hit = None
    for case, a, om, k, q, p, l, kind, dtype, hh in cases(want + 1, seed, wide):
        if case == want:
            hit = (a, om, k, q, p, l, kind, dtype, hh)
    assert hit is not None
    a, om, k, q, p, l, kind, dtype, hh = hit
    assert dtype == np.float32 and kind == "rankdef" and q == 3
    ref = orc.random_svd(a.astype(np.float64), k, q, p, omega=om.astype(np.float64))
    cpu32 = orc.random_svd(a, k, q, p, omega=om)

    def dev(usv):
        u, s, vt = usv
        ds = float(np.max(np.abs(s.ravel().astype(np.float64) - ref[1].ravel())) / ref[1][0, 0])
        return ds, abs(orc.relerr(a, u, s, vt) - orc.relerr(a, *ref))

    ds_cpu, re_cpu = dev(cpu32)
    gpu = ctx.rsvd(a, k, q, p, omega=om, qr="householder" if hh else None)
    ds_gpu, re_gpu = dev(gpu)
    print(f"{seed}:{want} l={l}: dS gpu {ds_gpu:.2e} cpu-f32 {ds_cpu:.2e}; |d relerr| gpu {re_gpu:.2e} cpu-f32 {re_cpu:.2e}")
    assert np.all(np.isfinite(gpu[1])) and np.all(np.diff(gpu[1].ravel()) <= 1e-6 * gpu[1][0, 0])
    assert ds_gpu <= max(2e-3, 3.0 * ds_cpu)
    assert re_gpu <= max(2e-3, 2.0 * re_cpu)
    # the factors themselves stay orthonormal over the numerically non-null triplets
    keep = ref[1].ravel() > 1e-4 * ref[1][0, 0]
    uu = gpu[0][:, keep].astype(np.float64)
    assert np.max(np.abs(uu.T @ uu - np.eye(uu.shape[1]))) <= 2e-3


# ---- the second-generation k-NN scan (csrc/knn2_kernels.hpp: bf16x3 MFMA filter, batched bitonic list merges) ----------
def _ref_gradients(x, y, xq, n_nbrs):
    """order-1 gradients from brute-force numpy neighbours (distance, index order) -- the oracle's algebra with its own
    neighbour search, vectorised so that thousands of queries stay cheap"""
    from oracle import active_ss_oracle as aso
    est = aso.PolyGradientEstimator(x, y, 1, n_nbrs)
    return aso.create_grad_mat(est, xq)


@pytest.mark.parametrize("n,k,n_nbrs,nq,kind", [
    (9000, 64, 80, 300, "gauss"),          # two MFMA steps, BASELINE config 5's dimensions
    (9000, 17, 30, 300, "gauss"),          # one MFMA step, odd dimension
    (12000, 33, 128, 200, "gauss"),        # the largest list the kernel serves
    (10000, 8, 12, 500, "offset"),         # a cloud far from the origin: centring keeps the filter's margin small
    (8200, 24, 60, 700, "dups"),           # repeated points: equal distances -> lower index
    (8193, 5, 70, 129, "clustered"),       # tight clusters: many near-ties, query tile tail (129 = 128 + 1)
    (20000, 64, 67, 64, "gauss"),          # list just beyond one 64-entry half
])
def test_knn2_neighbour_sets_are_exact(ctx, monkeypatch, n, k, n_nbrs, nq, kind):
    """CORRLA_KNN=3 (the default from 8192 points on): gradients equal to those of the exact search -- i.e. the same
    neighbour sets in the same order -- on Gaussian, offset, duplicated and clustered clouds, against the numpy oracle and
    against the VALU scan of round 1 (CORRLA_KNN=1)."""
    rng = np.random.default_rng(n + k)
    x = rng.standard_normal((n, k))
    if kind == "offset":
        x = x * 0.01 + 1000.0
    elif kind == "dups":
        x[n // 2: n // 2 + 2000] = x[:2000]
    elif kind == "clustered":
        centres = rng.standard_normal((40, k)) * 5.0
        x = centres[rng.integers(0, 40, n)] + 1e-3 * rng.standard_normal((n, k))
    w = rng.standard_normal(k)
    y = np.sin((x - x.mean(axis=0)) @ w * 0.3) + 0.05 * ((x - x.mean(axis=0)) ** 2).sum(axis=1)
    xq = x[rng.permutation(n)[:nq]] if kind != "gauss" else np.vstack([x[: nq // 2], rng.standard_normal((nq - nq // 2, k))])
    monkeypatch.setenv("CORRLA_KNN", "3")
    g3, nreg3 = ctx.grad_mat(x, y, 1, n_nbrs, xq)
    monkeypatch.setenv("CORRLA_KNN", "1")
    g1, nreg1 = ctx.grad_mat(x, y, 1, n_nbrs, xq)
    monkeypatch.delenv("CORRLA_KNN")
    scale = np.abs(g1).max()
    assert nreg3 == nreg1
    assert np.max(np.abs(g3 - g1)) <= 1e-9 * scale, float(np.max(np.abs(g3 - g1)) / scale)
    if kind in ("gauss", "offset"):
        go = _ref_gradients(x, y, xq[:60], n_nbrs)
        assert np.max(np.abs(g3[:, :60] - go)) <= 1e-8 * np.abs(go).max()


def test_knn2_is_the_default_scan_and_handles_few_queries_many_points(ctx, torch):
    """A device-resident cloud of 300 000 points, 1000 queries: a handful of query tiles, 4688 chunks, every flush point of
    the schedule; affine function -> exact slopes whatever the neighbours, and the sampled queries match the oracle."""
    g = torch.Generator(device="cuda").manual_seed(11)
    n, k = 300_000, 32
    x = torch.randn((n, k), dtype=torch.float64, device="cuda", generator=g)
    w = torch.linspace(1.0, 2.0, k, dtype=torch.float64, device="cuda")
    y = torch.sin(x @ w * 0.1)
    xq = x[:1000]
    gm, nreg = ctx.grad_mat(x, y, 1, 48, xq)
    assert nreg == 0 and gm.shape == (k, 1000)
    tm = ctx.timings()   # corrla_timings after a gradient call: the scan's and the fits' device time, nothing else
    assert tm["knn_ms"] > 0 and tm["fit_ms"] > 0 and tm["sketch_kernel_ms"] == 0 and tm["n_mixed_products"] == 0
    from oracle import active_ss_oracle as aso
    xs, ys = x.cpu().numpy(), y.cpu().numpy()
    go = aso.create_grad_mat(aso.PolyGradientEstimator(xs, ys, 1, 48), xs[:12])
    assert np.max(np.abs(gm[:, :12].cpu().numpy() - go)) <= 1e-9 * np.abs(go).max()


@pytest.mark.parametrize("k,n,n_nbrs", [(20, 3000, 260), (26, 4000, 400)])
def test_quadratic_fits_beyond_the_lds_limit_of_the_normal_equations(ctx, k, n, n_nbrs):
    """est_grad_quad (active_subspaces.rs:122-141, stats_corr.rs:198-249) for k = 20 / 26 features: 231 / 378 design columns,
    normal equations of 430 KB / 1.1 MB per query -- beyond LDS (round 2: CORRLA_EINVAL above k = 14), kept in global memory
    now.  Same neighbour sets and the same fitted quadratics as the oracle."""
    from oracle import active_ss_oracle as aso
    rng = np.random.default_rng(k)
    x = rng.standard_normal((n, k)) + 0.5
    qm = rng.standard_normal((k, k)) * 0.1
    qm = qm + qm.T
    b = rng.standard_normal(k)
    y = 0.5 * np.einsum("ni,ij,nj->n", x, qm, x) + x @ b + 3.0 + 1e-3 * np.sin(x @ b)
    nq = 24
    g, nreg = ctx.grad_mat(x, y, 2, n_nbrs, x[:nq])
    assert g.shape == (k, nq) and nreg == 0
    est = aso.PolyGradientEstimator(x, y, 2, n_nbrs)
    est.exact_quad_gradient = True       # the same fitted quadratics differentiated exactly
    go = aso.create_grad_mat(est, x[:nq])
    assert np.max(np.abs(g - go)) <= 1e-7 * np.abs(go).max()
    # an exact quadratic with an offset is recovered exactly
    y2 = 0.5 * np.einsum("ni,ij,nj->n", x, qm, x) + x @ b + 3.0
    g2, _ = ctx.grad_mat(x, y2, 2, n_nbrs, x[:nq])
    assert np.max(np.abs(g2 - (x[:nq] @ qm + b).T)) <= 1e-6 * np.abs(g2).max()


# ---- f64 tall products with two MFMA waves per SIMD (round 3) --------------------------------------------------------
@pytest.mark.parametrize("m,n,l", [
    (4096, 1024, 138),     # 9 column tiles, both kernels on the 128-index tile
    (1000, 700, 266),      # ragged edges, two column blocks (9 + 8 tiles)
    (65536, 512, 40),      # persistent launch over the outer tiles (short reduction)
    (300, 20000, 17),      # long reduction split over workgroups (slab reduce)
])
def test_f64_products_with_eight_mfma_waves_are_bitwise_those_of_four(torch, monkeypatch, m, n, l):
    """The f64 instantiations <double, 1, NT, 8 waves> (default) and <double, 2, NT, 4 waves> (CORRLA_F64_WAVES=4, round 2)
    share the tile geometry and every accumulator's summation order, so A X and A^T Y must agree to the last bit; both
    must agree with torch to rounding."""
    import corrla_rs_amd as cr
    g = torch.Generator(device="cuda").manual_seed(m + n + l)
    a = torch.randn((m, n), dtype=torch.float64, device="cuda", generator=g)
    x = torch.randn((n, l), dtype=torch.float64, device="cuda", generator=g)
    y = torch.randn((m, l), dtype=torch.float64, device="cuda", generator=g)
    monkeypatch.setenv("CORRLA_F64_WAVES", "8")
    c8 = cr.Context(0)
    monkeypatch.setenv("CORRLA_F64_WAVES", "4")
    c4 = cr.Context(0)
    monkeypatch.delenv("CORRLA_F64_WAVES")
    for trans, rhs in ((False, x), (True, y)):
        z8 = c8.matmul(a, rhs, trans=trans)
        z4 = c4.matmul(a, rhs, trans=trans)
        ref = (a.t() if trans else a) @ rhs
        assert torch.equal(z8, z4), float((z8 - z4).abs().max())
        assert float((z8 - ref).abs().max()) <= 1e-11 * float(ref.abs().max())


def test_many_neighbours_on_a_large_cloud_take_the_scan_whose_lists_fit(ctx):
    """n_nbrs = 480 on 140 000 points: beyond the 128-entry lists of the bf16-filter scan, and the per-query lists of round
    2's MFMA scan would need 190 KB of LDS -- the call used to end with CORRLA_EINVAL ("does not fit in LDS") although the
    interface promises n_nbrs <= 512 (found by tools/fuzz_grad.py).  The VALU scan serves it."""
    from oracle import active_ss_oracle as aso
    rng = np.random.default_rng(480)
    n, k, n_nbrs, nq = 140_000, 12, 480, 40
    x = rng.standard_normal((n, k))
    y = np.cos(x @ rng.standard_normal(k) * 0.2) + 0.1 * (x ** 2).sum(axis=1)
    xq = x[rng.permutation(n)[:nq]]
    g, nreg = ctx.grad_mat(x, y, 1, n_nbrs, xq)
    assert nreg == 0
    go = aso.create_grad_mat(aso.PolyGradientEstimator(x, y, 1, n_nbrs), xq[:6])
    assert np.max(np.abs(g[:, :6] - go)) <= 1e-9 * np.abs(go).max()


def test_non_finite_support_points_never_become_neighbours_and_do_not_slow_the_scan(ctx, monkeypatch):
    """A few NaN / inf rows in the cloud: their distances are non-finite and sort last (like numpy's argsort of a NaN), so
    the gradients are those of the cloud without them; and they must not enter the centre the filter works around -- one
    NaN there would make every pair pass the filter, i.e. turn the scan into the exact re-check of all pairs."""
    import time
    rng = np.random.default_rng(31)
    n, k, n_nbrs, nq = 60_000, 16, 40, 512
    x = rng.standard_normal((n, k))
    y = np.sin(x @ rng.standard_normal(k) * 0.3)
    bad = rng.permutation(n)[:7]
    xb = x.copy()
    xb[bad[:3], 2] = np.nan
    xb[bad[3:5], 0] = np.inf
    xb[bad[5:], 5] = -np.inf
    keep = np.setdiff1d(np.arange(n), bad)
    xq = x[keep[rng.permutation(keep.size)[:nq]]]
    monkeypatch.setenv("CORRLA_KNN", "3")
    ctx.grad_mat(x, y, 1, n_nbrs, xq[:8])                     # warm-up (code object, workspaces)
    t0 = time.perf_counter()
    g_bad, nreg = ctx.grad_mat(xb, y, 1, n_nbrs, xq)
    t_bad = time.perf_counter() - t0
    t0 = time.perf_counter()
    g_ref, nreg_ref = ctx.grad_mat(x[keep], y[keep], 1, n_nbrs, xq)
    t_ref = time.perf_counter() - t0
    monkeypatch.delenv("CORRLA_KNN")
    assert nreg == nreg_ref == 0
    assert np.max(np.abs(g_bad - g_ref)) <= 1e-9 * np.abs(g_ref).max()
    assert t_bad < 5.0 * t_ref + 0.5, (t_bad, t_ref)


# ---- multi-workgroup Jacobi: the wave-local sub-block schedule against the ring schedule and the oracle ---------------
# (jacobi_mc_kernels.hpp: same pairs per sweep, different order; widths chosen so that the block width b is and is not a
#  multiple of four, the sub-block count is odd and even, and the last sub-block is ragged)
@pytest.mark.parametrize("force_v", ["0", "1"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_jacobi_wave_local_schedule_matches_ring_schedule_and_oracle(ctx, monkeypatch, dtype, force_v):
    rng = np.random.default_rng(31)
    f64 = dtype == np.float64
    m, n = 1200, 600
    a = (rng.standard_normal((m, n)) * (0.996 ** np.arange(n))).astype(dtype)
    monkeypatch.setenv("CORRLA_JMC_FORCE_V", force_v)
    for l in (96, 97, 101, 109, 117, 125, 133, 138, 141, 150, 163, 171, 200, 233, 266, 288):  # l << n: f32 stays determined
        p = 10
        k = l - p
        om = rng.standard_normal((n, l)).astype(dtype)
        uo, so, vto = orc.random_svd(a.astype(np.float64), k, 2, p, omega=om.astype(np.float64))
        res = {}
        for local in ("1", "0"):
            monkeypatch.setenv("CORRLA_JMC_LOCAL", local)
            u, s, vt = ctx.rsvd(a, k, 2, p, omega=om)
            res[local] = (u, s, vt)
            assert np.max(np.abs(s.astype(np.float64) - so)) <= (1e-10 if f64 else 1e-4) * so[0, 0], (l, local)
            assert abs(orc.relerr(a, u, s, vt) - orc.relerr(a, uo, so, vto)) <= (1e-9 if f64 else 1e-5), (l, local)
            assert orth_err(u) < (1e-11 if f64 else 2e-4) and orth_err(vt.T) < (1e-11 if f64 else 2e-4), (l, local)
            assert np.all(np.diff(s.ravel()) <= 0) and np.all(s >= 0), (l, local)
        s1, s0 = res["1"][1].astype(np.float64), res["0"][1].astype(np.float64)
        assert np.max(np.abs(s1 - s0)) <= (1e-12 if f64 else 2e-5) * so[0, 0], l
    monkeypatch.delenv("CORRLA_JMC_LOCAL")


# ---- gemm_tn: the XCD-aware block mapping moves work between compute units, not bits ---------------------------------
@pytest.mark.parametrize("dtype,m,n,k,split", [(np.float32, 300_000, 512, 64, None),   # C4's shape family: 4 outer tiles x 128 slabs
                                               (np.float64, 40_000, 2048, 256, None),   # C3's: uneven column blocking, 16 outer tiles
                                               (np.float32, 300_000, 512, 64, "20"),    # a last, partial group of slabs (20 = 2 x 8 + 4)
                                               (np.float32, 300_000, 384, 64, "9")])    # 3 outer tiles, 9 slabs
def test_gemm_tn_xcd_mapping_is_bitwise_neutral(monkeypatch, torch, dtype, m, n, k, split):
    import corrla_rs_amd as cr
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    res = []
    if split:
        monkeypatch.setenv("CORRLA_SPLIT_TN", split)
    for flag in ("1", "0"):
        monkeypatch.setenv("CORRLA_GEMM_XCD", flag)
        c = cr.Context()
        a = torch.empty((m, n), dtype=tdt, device="cuda")
        c.fill_normal(a, seed=9)
        u, s, vt = c.rsvd(a, k, 2, 10, seed=4)
        res.append((u.cpu().numpy(), s.cpu().numpy(), vt.cpu().numpy()))
        c.close()
    monkeypatch.delenv("CORRLA_GEMM_XCD")
    for x, y in zip(res[0], res[1]):
        assert np.array_equal(x, y)
    assert np.all(np.isfinite(res[0][1])) and res[0][1][0] > 0
