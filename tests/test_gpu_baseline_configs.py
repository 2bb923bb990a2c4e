"""GPU parity tests at BASELINE.json's own configurations (C1, C1b, C3, C4, C5) -- the sizes the
metric is quoted on, not scaled-down families of them.  C2 lives in tests/test_gpu_parity.py.  Every call goes through
the C ABI of libcorrla_rsvd.so; the oracle (numpy restatement of random_svd.rs:15-110) runs on the GPU box's host
cores with the SAME matrix (copied D2H) and the SAME Omega wherever it finishes in seconds, and size-independent
properties cover the rest.  Run with -m gpu on an MI355X."""
import numpy as np
import pytest

from oracle import rsvd_oracle as orc

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


def _blocked_props(torch, a, u, s, vt, k, eps, step=8192):
    """Orthonormality of U and V, Rayleigh consistency s_i = u_i^T A v_i, ordering, and the relative Frobenius error
    through ||A - U S Vt||^2 = ||A||^2 - 2 sum s_i (u_i^T A v_i) + sum s_i^2 (exact for orthonormal factors); f64
    accumulation over row blocks so the 20 GB case never materialises a second copy."""
    vd, sd = vt.double().t().contiguous(), s.double().ravel()
    eye = torch.eye(k, dtype=torch.float64, device=a.device)
    m = a.shape[0]
    utu = torch.zeros((k, k), dtype=torch.float64, device=a.device)
    ray = torch.zeros((k,), dtype=torch.float64, device=a.device)
    fro2 = 0.0
    for r0 in range(0, m, step):
        ab = a[r0:r0 + step].double()
        ub = u[r0:r0 + step].double()
        utu += ub.t() @ ub
        ray += (ub * (ab @ vd)).sum(dim=0)
        fro2 += float((ab * ab).sum().item())
    tol = 50 * eps * np.sqrt(k) * 4
    assert (utu - eye).abs().max().item() < tol
    assert (vd.t() @ vd - eye).abs().max().item() < tol
    assert ((ray - sd).abs().max() / sd[0]).item() < 1e3 * eps
    assert torch.all(sd[:-1] >= sd[1:] - 1e-6 * sd[0]) and torch.all(sd >= 0)
    err2 = fro2 - 2 * float((sd * ray).sum().item()) + float((sd ** 2).sum().item())
    return np.sqrt(max(err2, 0.0) / fro2)


def _align(u, vt, u_ref, vt_ref):
    """flip (u_i, v_i) pairs so they match the reference's signs (the reference fixes none)"""
    sg = np.sign(np.sum(vt * vt_ref, axis=1))
    sg[sg == 0] = 1.0
    return u * sg, vt * sg[:, None]


def test_c1_exact(ctx, torch):
    """BASELINE configs[0]: 1024 x 1024 f64, rank 32, n_iters 4, n_oversamples 8; shared Omega, vs the oracle.  q = 4
    runs one in-loop re-orthonormalisation (i = 3 > 2, random_svd.rs:37-39)."""
    m = n = 1024
    k, q, p = 32, 4, 8
    a = torch.empty((m, n), dtype=torch.float64, device="cuda")
    ctx.fill_normal(a, seed=20241008)
    a_h = a.cpu().numpy()
    om = np.random.default_rng(1).standard_normal((n, k + p))
    for inp in (a, a_h):          # device-pointer and host-pointer entry points
        u, s, vt = ctx.rsvd(inp, k, q, p, omega=om)
        if hasattr(u, "cpu"):
            u, s, vt = u.cpu().numpy(), s.cpu().numpy(), vt.cpu().numpy()
        uo, so, vto = orc.random_svd(a_h, k, q, p, omega=om)
        assert u.shape == (m, k) and s.shape == (k, 1) and vt.shape == (k, n)
        assert np.max(np.abs(s - so)) <= 1e-10 * so[0, 0]
        assert abs(orc.relerr(a_h, u, s, vt) - orc.relerr(a_h, uo, so, vto)) <= 1e-10
        ua, vta = _align(u, vt, uo, vto)
        # singular vectors: gaps between neighbouring values of a Gaussian matrix are ~1e-3 relative, so vectors agree
        # to ~eps / gap; the subspace (projector) agrees to rounding
        assert np.linalg.norm(ua @ ua.T @ uo - uo) <= 1e-9
        assert np.max(np.abs(u.T @ u - np.eye(k))) < 1e-12 and np.max(np.abs(vt @ vt.T - np.eye(k))) < 1e-12


def test_c1b_the_benchmark_scripts_real_shape(ctx, torch):
    """SURVEY section 0 / 8 (config 1b): examples/benchmark_rsvd.py:62,65-66,101 really runs 100000 x 10000 f64, rank 4,
    8 power iterations, 10 oversamples (8 GB).  Properties on the device, then the oracle on the host on the full
    matrix with the same Omega (l = 14: 18 skinny passes over 8 GB)."""
    m, n, k, q, p = 100_000, 10_000, 4, 8, 10
    a = torch.empty((m, n), dtype=torch.float64, device="cuda")
    ctx.fill_normal(a, seed=20241008)
    om = torch.empty((n, k + p), dtype=torch.float64, device="cuda")
    ctx.fill_normal(om, seed=1)
    u, s, vt = ctx.rsvd(a, k, q, p, omega=om)
    re = _blocked_props(torch, a, u, s, vt, k, 2.2e-16)
    assert 0.99 < re < 1.0
    a_h, om_h = a.cpu().numpy(), om.cpu().numpy()
    del a
    torch.cuda.empty_cache()
    uo, so, vto = orc.random_svd(a_h, k, q, p, omega=om_h)
    assert np.max(np.abs(s.cpu().numpy() - so)) <= 1e-10 * so[0, 0]
    ua, vta = _align(u.cpu().numpy(), vt.cpu().numpy(), uo, vto)
    assert np.max(np.abs(vta - vto)) <= 1e-8


def test_c3_full_size_pod_schedule_f64(ctx, torch):
    """BASELINE configs[2]: POD-by-RSVD, 65536 x 4096 f64 snapshot matrix, rank 256, at its REAL size and the
    reference's schedule -- PodI::_modes calls random_svd(x, n_modes, 10, 10) (pod_rom.rs:56): q = 10, so the seven
    in-loop re-orthonormalisations (i = 3..9) run at l = 266 through the 2 x 2 blocked device Cholesky, then the final
    thin-Q and the l = 266 core SVD.  Device property checks + the full oracle on the host with the same Omega."""
    m, n, k, q, p = 65536, 4096, 256, 10, 10
    a = torch.empty((m, n), dtype=torch.float64, device="cuda")
    ctx.fill_normal(a, seed=7)
    om = torch.empty((n, k + p), dtype=torch.float64, device="cuda")
    ctx.fill_normal(om, seed=2)
    u, s, vt = ctx.rsvd(a, k, q, p, omega=om)
    re = _blocked_props(torch, a, u, s, vt, k, 2.2e-16)
    assert 0.8 < re < 0.999
    a_h, om_h = a.cpu().numpy(), om.cpu().numpy()
    uo, so, vto = orc.random_svd(a_h, k, q, p, omega=om_h)
    s_h = s.cpu().numpy()
    assert np.max(np.abs(s_h - so)) <= 1e-10 * so[0, 0]
    # relerr identity for the oracle's factors vs ours (north star: within 1e-5; f64 gives far better)
    fro2 = float(np.sum(a_h.astype(np.float64) ** 2))
    re_o = np.sqrt(max(fro2 - float(np.sum(so ** 2)), 0.0) / fro2)    # orthonormal factors, s_i = u_i^T A v_i
    assert abs(re - re_o) <= 1e-9
    # row-sampled comparison of U = Q U~ against the oracle on a 4096-row slice, up to the sign of each triplet
    rows = np.random.default_rng(3).choice(m, 4096, replace=False)
    ua, vta = _align(u.cpu().numpy(), vt.cpu().numpy(), uo, vto)
    assert np.max(np.abs(vta - vto)) <= 1e-8
    assert np.max(np.abs(ua[rows] - uo[rows])) <= 1e-8
    # POD modes = V (N x k): the caller's surface returns the same matrix
    import corrla_rs_amd as cr
    assert hasattr(cr, "pod_modes")


def test_c4_whole_matrix_on_one_gpu_f32(ctx, torch):
    """BASELINE configs[3]: 10,000,000 x 512 f32, rank 64 (q = 2, p = 10) -- the WHOLE 20.5 GB matrix on one MI355X
    through the size-independent properties (the N > 1 split of the same matrix is the sharded entry point, next test
    and tests/test_sharded_gloo.py)."""
    m, n, k, q, p = 10_000_000, 512, 64, 2, 10
    a = torch.empty((m, n), dtype=torch.float32, device="cuda")
    ctx.fill_normal(a, seed=11)
    u, s, vt = ctx.rsvd(a, k, q, p, seed=3)
    re = _blocked_props(torch, a, u, s, vt, k, 1.2e-7, step=262144)
    assert 0.8 < re < 0.999
    # Gaussian 1e7 x 512: sigma_1 ~ sqrt(m) + sqrt(n)
    assert 0.98 * (np.sqrt(m) + np.sqrt(n)) < s[0, 0].item() < 1.02 * (np.sqrt(m) + np.sqrt(n))
    # linearity in A (same seed): singular values scale, factors do not move
    a *= 0.5
    u2, s2, vt2 = ctx.rsvd(a, k, q, p, seed=3)
    assert torch.allclose(s2, 0.5 * s, rtol=2e-5)
    assert (vt2 - vt).abs().max().item() < 5e-4


def test_c4_shard_sharded_entry_with_rccl_and_oracle(torch, monkeypatch):
    """One rank's 1,250,000 x 512 shard of config 4 through corrla_rsvd_sharded_dev_f32 on a one-rank RCCL
    communicator with every all-reduce actually issued (CORRLA_FORCE_ALLREDUCE=1: the calls, datatypes, counts and
    stream the 8 ranks make), against the plain entry point and against the oracle on the host with the same Omega."""
    import corrla_rs_amd as cr
    c = cr.Context(0)
    c.comm_init(cr.Context.unique_id(), 0, 1)
    m, n, k, q, p = 1_250_000, 512, 64, 2, 10
    a = torch.empty((m, n), dtype=torch.float32, device="cuda")
    c.fill_normal(a, seed=11, row0=3 * m, global_cols=n)       # rows of rank 3 of 8
    om = np.random.default_rng(5).standard_normal((n, k + p)).astype(np.float32)
    u0, s0, vt0 = c.rsvd(a, k, q, p, omega=om)
    monkeypatch.setenv("CORRLA_FORCE_ALLREDUCE", "1")
    u1, s1, vt1 = c.rsvd_sharded(a, k, q, p, omega=om)
    monkeypatch.delenv("CORRLA_FORCE_ALLREDUCE")
    assert torch.equal(s0, s1) and torch.equal(vt0, vt1) and torch.equal(u0, u1)
    a_h = a.cpu().numpy()
    uo, so, vto = orc.random_svd(a_h, k, q, p, omega=om)
    s_h = s0.cpu().numpy().astype(np.float64)
    assert np.max(np.abs(s_h - so)) <= 2e-5 * so[0, 0]
    fro2 = float(np.sum(a_h.astype(np.float64) ** 2))
    re_g = _blocked_props(torch, a, u0, s0, vt0, k, 1.2e-7, step=131072)
    uo64, vto64 = uo.astype(np.float64), vto.astype(np.float64)
    ray_o = np.einsum("ik,ik->k", uo64, a_h.astype(np.float64) @ vto64.T)
    re_o = np.sqrt(max(fro2 - 2 * float(np.sum(so.ravel() * ray_o)) + float(np.sum(so ** 2)), 0.0) / fro2)
    assert abs(re_g - re_o) <= 1e-5           # north star tolerance
    c.close()


def test_c5_gradient_stage_and_fit_svd_at_1e6_points(ctx, torch):
    """BASELINE configs[4]: active-subspace sensitivity on 1,000,000 x 64 samples.  The gradient stage (exact k-NN +
    local linear fits, active_subspaces.rs:66-141, 215-229) runs for all 10^6 queries on the GPU; 48 sampled queries are
    checked against the oracle's brute-force neighbours + pinv fit on the same cloud, then fit_svd's RSVD of
    G / sqrt(N) (active_subspaces.rs:233-250: rank min(k, n_comps), q = 8, p = 10) is compared with the oracle's
    random_svd of the SAME gradient matrix with the same Omega."""
    from oracle import active_ss_oracle as aso
    n, k, n_nbrs = 1_000_000, 64, 80
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((n, k), dtype=torch.float64, device="cuda", generator=g)
    w = torch.linspace(1.0, 0.05, k, dtype=torch.float64, device="cuda")
    y = torch.sin(x @ w * 0.2) + 0.05 * ((x * w) ** 2).sum(dim=1)
    gm, nreg = ctx.grad_mat(x, y, 1, n_nbrs, scale=1.0 / np.sqrt(n))       # k x n, on the device
    assert gm.shape == (k, n) and nreg == 0 and bool(torch.isfinite(gm).all())
    xs, ys = x.cpu().numpy(), y.cpu().numpy()
    est = aso.PolyGradientEstimator(xs, ys, 1, n_nbrs)
    rows = np.random.default_rng(9).choice(n, 48, replace=False)
    go = aso.create_grad_mat(est, xs[rows]) / np.sqrt(n)
    gg = gm[:, torch.as_tensor(rows, device="cuda")].cpu().numpy()
    assert np.max(np.abs(gg - go)) <= 1e-9 * np.abs(go).max()
    n_comps = 32
    om = np.random.default_rng(2).standard_normal((k, min(n_comps + 10, k)))
    u, s, vt = ctx.rsvd(gm, n_comps, 8, 10, omega=om)                       # fat: works on the n x k tall view
    gm_h = gm.cpu().numpy()
    uo, so, vto = orc.random_svd(gm_h, n_comps, 8, 10, omega=om)
    s_h, u_h = s.cpu().numpy(), u.cpu().numpy()
    assert np.max(np.abs(s_h - so)) <= 1e-10 * so[0, 0]
    assert np.linalg.norm(u_h @ (u_h.T @ uo) - uo) <= 1e-8                  # components: same subspace
    # the dominant direction follows the largest weights of the synthetic function
    assert abs(u_h[0, 0]) > abs(u_h[k - 1, 0])
