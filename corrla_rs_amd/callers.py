"""Host-side mirrors of the reference callers that sit on the RSVD hot path.
Every m- or n-sized decomposition goes through the GPU `rsvd`; what remains here is the k-wide algebra the
reference also does after its random_svd calls (k = n_modes, a few tens), written with numpy.

  pod_modes(x, n_modes)                       <- PodI::_modes            src/lib_math_utils/pod_rom.rs:53-58
  active_ss_fit_svd(grad_mat, n_comps, ...)   <- ActiveSsRsvd::fit_svd   src/lib_math_utils/active_subspaces.rs:233-250
  DMDc(x, u, dt, n_modes, n_iters)            <- DMDc::new               src/lib_math_utils/dmd_rom.rs:45-226
                                                 (pyo3 PyDMDc, src/lib_math_utils_py.rs:222-283); CUDA tensors in ->
                                                 device-resident factors and a factored predictor
  PodI(x, t, n_modes) / RbfInterp             <- PodI, RbfInterp         src/lib_math_utils/pod_rom.rs:36-117,
                                                 src/lib_math_utils/interp_utils.rs:11-160 (pyo3 PyPodI, PyRbfInterp)
  PolyGradientEstimator / ActiveSsRsvd / FittedActiveSsRsvd
                                              <- src/lib_math_utils/active_subspaces.rs:21-277 (the
                                                 neighbour search and the local fits run on the GPU)
Every product with an n_x- or N-sized dimension goes through the library's HIP GEMMs (``Context.matmul``); only
snapshot-count- and n_modes-sized matrices are touched by numpy."""
import numpy as np

from .api import _is_torch, default_context, rsvd

__all__ = ["pod_modes", "active_ss_fit_svd", "DMDc", "PodI", "RbfInterp", "PolyGradientEstimator", "ActiveSsRsvd",
           "FittedActiveSsRsvd"]


def _on_gpu(x):
    return _is_torch(x) and x.is_cuda


def pod_modes(x_data, n_modes, *, seed=None, omega=None, ctx=None):
    """(_u, _s, v) = random_svd(x_data, n_modes, 10, 10); modes = v^T, shape (N, n_modes)."""
    _u, _s, vt = (ctx or default_context()).rsvd(np.asarray(x_data, np.float64), n_modes, 10, 10, seed=seed, omega=omega)
    return np.ascontiguousarray(vt.T)


def active_ss_fit_svd(grad_mat, n_comps, n_iter=8, n_oversamples=10, *, seed=None, omega=None, ctx=None):
    """RSVD variant of the active-subspace fit, given the k x N gradient matrix: scale by 1/sqrt(N),
    random_svd(., min(k, n_comps), n_iter, n_oversamples).  Returns (components U (k, r), diag(S) (r, r))."""
    g = np.asarray(grad_mat, np.float64)
    k_features, n_samples = g.shape
    u, s, _vt = (ctx or default_context()).rsvd(g * (1.0 / np.sqrt(float(n_samples))), min(k_features, n_comps), n_iter,
                                                n_oversamples, seed=seed, omega=omega)
    return u, np.diag(s.ravel())


class PolyGradientEstimator:
    """Mirror of ``PolyGradientEstimator`` (active_subspaces.rs:21-141): local polynomial gradient estimates over a
    point cloud.  The nearest-neighbour search and the per-point least-squares fits run on the GPU
    (``corrla_grad_mat_f64``); ``grad_at`` returns the reference's 1 x k row."""

    def __init__(self, x_mat, y, est_order, n_nbrs, *, ctx=None):
        self.x_mat = np.ascontiguousarray(np.asarray(x_mat, dtype=np.float64))
        self.y = np.asarray(y, dtype=np.float64).reshape(-1)
        self.est_order, self.n_nbrs, self.k = int(est_order), int(n_nbrs), self.x_mat.shape[1]
        self._ctx = ctx
        if self.est_order not in (1, 2):
            raise ValueError("Not implemented est order")   # the reference panics (active_subspaces.rs:60)

    def grad_mat(self, x_query=None, scale=1.0):
        """k x n_q gradient matrix (create_grad_mat, active_subspaces.rs:215-229)."""
        g, self.n_regularised = (self._ctx or default_context()).grad_mat(self.x_mat, self.y, self.est_order, self.n_nbrs,
                                                                           x_query, scale=scale)
        return g

    def grad_at(self, x0):
        return self.grad_mat(np.asarray(x0, dtype=np.float64).reshape(1, -1)).T.copy()


class FittedActiveSsRsvd:
    """active_subspaces.rs:41-47, 143-201."""

    def __init__(self, components, singular_vals, n_comps):
        self.components_, self.singular_vals_, self.n_comps = components, singular_vals, int(n_comps)

    def components(self):
        return self.components_[:, : self.n_comps]

    def singular_vals(self):
        return self.singular_vals_[:, : self.n_comps]

    def transform(self, x_mat):
        return np.asarray(x_mat, dtype=np.float64) @ self.components()

    def inv_transform(self, x_mat):
        x = np.asarray(x_mat, dtype=np.float64)
        if x.shape[1] != self.n_comps:
            raise ValueError("x_mat must have n_comps columns")     # assert at active_subspaces.rs:186
        return x @ self.components().T

    def var_diag_evd_sensi(self):
        m = self.components_.T @ self.singular_vals_ @ self.components_   # as written at active_subspaces.rs:162-164
        return np.diag(m).copy()


class ActiveSsRsvd:
    """Mirror of ``ActiveSsRsvd`` (active_subspaces.rs:35-277): gradient matrix on the GPU, then either the RSVD of
    G / sqrt(N) (``fit_svd``, on the GPU) or the eigendecomposition of the k x k matrix G G^T / N (``fit``; the Gram
    product runs on the GPU through the RSVD library's GEMM, the k x k symmetric eigenproblem in numpy)."""

    def __init__(self, grad_est, n_comps, *, ctx=None):
        self.grad_est, self.n_comps, self._ctx = grad_est, int(n_comps), ctx

    def fit_svd(self, x_mat, n_iter=None, n_oversamples=None, *, seed=None, omega=None):
        x = np.asarray(x_mat, dtype=np.float64)
        g = self.grad_est.grad_mat(x, scale=1.0 / np.sqrt(float(x.shape[0])))       # :238-239
        u, s, _vt = rsvd(g, min(x.shape[1], self.n_comps), 8 if n_iter is None else n_iter,
                         10 if n_oversamples is None else n_oversamples, seed=seed, omega=omega, ctx=self._ctx)   # :242-244
        return FittedActiveSsRsvd(u, np.diag(s.ravel()), self.n_comps)

    def fit_svd_sharded(self, x_mat, rank, world, n_iter=None, n_oversamples=None, *, seed=1):
        """``fit_svd`` with the sample points sharded over `world` GPUs, one process per GPU (BASELINE config 5): the
        support cloud is replicated, rank r estimates the gradients of its contiguous slice of the samples (no
        exchange), and the RSVD of G / sqrt(N) runs on the row-sharded N x k tall view (``rsvd_sharded``: all-reduces of
        k x l blocks only).  The context must carry a communicator (``Context.comm_init``).  Every rank returns the
        same fitted object."""
        import torch
        x = np.asarray(x_mat, dtype=np.float64)
        n = x.shape[0]
        lo, hi = (n * rank) // world, (n * (rank + 1)) // world
        ctx = self._ctx or default_context()
        dev = torch.device(f"cuda:{ctx.device}")
        xs = torch.as_tensor(np.ascontiguousarray(self.grad_est.x_mat), device=dev)
        ys = torch.as_tensor(self.grad_est.y, device=dev)
        g_loc, self.grad_est.n_regularised = ctx.grad_mat(xs, ys, self.grad_est.est_order, self.grad_est.n_nbrs,
                                                          torch.as_tensor(np.ascontiguousarray(x[lo:hi]), device=dev),
                                                          scale=1.0 / np.sqrt(float(n)))
        k_comp = min(x.shape[1], self.n_comps)
        _u_loc, s, vt = ctx.rsvd_sharded(g_loc.t(), k_comp, 8 if n_iter is None else n_iter,
                                         10 if n_oversamples is None else n_oversamples, seed=seed)
        # the tall view is G^T (N x k): its right singular vectors are the k x r components `ur` of fit_svd
        return FittedActiveSsRsvd(vt.t().cpu().numpy().copy(), np.diag(s.cpu().numpy().ravel()), self.n_comps)

    def fit(self, x_mat):
        import torch
        x = np.asarray(x_mat, dtype=np.float64)
        ctx = self._ctx or self.grad_est._ctx or default_context()
        dev = torch.device(f"cuda:{ctx.device}")
        ge = self.grad_est
        g, ge.n_regularised = ctx.grad_mat(torch.as_tensor(ge.x_mat, device=dev), torch.as_tensor(ge.y, device=dev),
                                           ge.est_order, ge.n_nbrs, torch.as_tensor(np.ascontiguousarray(x), device=dev))
        gt = g.t()                                                                   # N x k row-major, on the device
        c = ctx.matmul(gt, gt, trans=True).cpu().numpy() * (1.0 / x.shape[0])        # G G^T / N  (k x k), :256
        c = 0.5 * (c + c.T)
        w, v = np.linalg.eigh(c)
        order = np.argsort(-w, kind="stable")                                        # sort_evd, mat_utils.rs:459-478
        return FittedActiveSsRsvd(v[:, order], np.diag(w[order]), self.n_comps)


def _pinv_diag(d):  # mat_pinv_diag, mat_utils.rs:386-402
    out = np.zeros_like(d)
    idx = np.arange(d.shape[1])
    v = d[idx, idx]
    big = np.abs(v) >= 1e-20
    out[idx[big], idx[big]] = 1.0 / (v[big] + 1e-20)
    return out


class DMDc:
    """Dynamic mode decomposition with control (Proctor et al.), dmd_rom.rs:20-226: two randomized SVDs with 12
    oversamples (input space [x; u][:, :-1] and output space x[:, 1:]) on the GPU, then the n_modes-wide
    operator algebra and the complex eigendecomposition of the n_modes x n_modes A~ on the host."""

    def __init__(self, x_data, u_data, dt, n_modes, n_iters, *, seed=None, omega_x=None, omega_y=None, ctx=None):
        c = ctx or default_context()
        self.on_device = _on_gpu(x_data)
        if self.on_device:
            self._init_device(c, x_data, u_data, float(dt), int(n_modes), int(n_iters), seed, omega_x, omega_y)
            return
        x_data = np.asarray(x_data, np.float64)
        u_data = np.asarray(u_data, np.float64)
        self.n_snapshots, self.n_x, self.n_u = x_data.shape[1], x_data.shape[0], u_data.shape[0]
        self.n_modes, self.dt_snapshots = int(n_modes), float(dt)
        omega = np.vstack([x_data, u_data])
        xin, yout = omega[:, :-1], omega[: self.n_x, 1:]
        u_til, s_til, vt_til = c.rsvd(xin, n_modes, n_iters, 12, seed=seed, omega=omega_x)
        u_hat, _s_hat, _vt_hat = c.rsvd(yout, n_modes, n_iters, 12, seed=None if seed is None else seed + 1, omega=omega_y)
        v_til = vt_til.T
        u1, u2 = u_til[: self.n_x], u_til[self.n_x:]
        s_inv = _pinv_diag(np.diag(s_til.ravel()))
        tmp = u_hat.T @ yout @ v_til @ s_inv          # eq. 29
        self._A = tmp @ u1.T @ u_hat
        self._B = u_hat @ (tmp @ u2.T)                # eq. 30, lifted back
        lam, w = np.linalg.eig(self._A)
        self.lambdas = lam.reshape(-1, 1)
        scale = yout @ (v_til @ (s_inv @ (u1.T @ u_hat)))   # eq. 36
        modes = scale @ w
        self.modes_re, self.modes_im = np.real(modes).copy(), np.imag(modes).copy()

    # ---- device-resident variant: x_data / u_data are CUDA tensors ------------------------------
    def _init_device(self, c, x_data, u_data, dt, n_modes, n_iters, seed, omega_x, omega_y):
        """Same algebra as the host path with every n_x-sized factor left on the GPU: the two randomized SVDs return
        device tensors, the products that contract or produce an n_x-sized dimension run through ``Context.matmul``
        (the library's HIP GEMMs), and only snapshot-count x n_modes and n_modes x n_modes matrices visit the host
        (diag pinv, the complex eigendecomposition of A~).  The dense n_x x n_x operator of ``est_a_til`` is never
        needed for prediction: A = Re(Phi Lambda Phi^+) = Pc K Pc^T with Pc = [Re Phi | Im Phi] (n_x x 2 n_modes) and
        a 2 n_modes x 2 n_modes real K, so ``predict_multiple`` runs its recurrence in 2 n_modes dimensions."""
        import torch
        dev = torch.device(f"cuda:{c.device}")
        self._ctx, self._dev = c, dev
        x = x_data.to(device=dev, dtype=torch.float64)
        u = torch.as_tensor(u_data, dtype=torch.float64, device=dev)
        self.n_snapshots, self.n_x, self.n_u = x.shape[1], x.shape[0], u.shape[0]
        self.n_modes, self.dt_snapshots = n_modes, dt
        k = n_modes
        omega = torch.cat([x, u], dim=0)                         # mat_vstack, dmd_rom.rs:66
        xin, yout = omega[:, :-1], omega[: self.n_x, 1:]         # _X / _Y: strided views, no copies
        u_til, s_til, vt_til = c.rsvd(xin, k, n_iters, 12, seed=seed, omega=omega_x)
        u_hat, _s, _v = c.rsvd(yout, k, n_iters, 12, seed=None if seed is None else seed + 1, omega=omega_y)
        u1, u2 = u_til[: self.n_x], u_til[self.n_x:]
        up = lambda h: torch.as_tensor(np.ascontiguousarray(h), dtype=torch.float64, device=dev)  # noqa: E731
        s_inv = _pinv_diag(np.diag(s_til.cpu().numpy().ravel()))
        v_til = vt_til.t().cpu().numpy()                         # (n_t - 1, k)
        ytu = c.matmul(yout, u_hat, trans=True).cpu().numpy()    # Y^T U^ (n_t - 1, k)
        tmp = ytu.T @ v_til @ s_inv                              # eq. 29: U^^T Y V~ S~^-1
        m1 = c.matmul(u1, u_hat, trans=True).cpu().numpy()       # U~1^T U^ (k, k)
        self._A = tmp @ m1
        b_til = c.matmul(u2, up(tmp.T))                          # (n_u, k) = U~2 tmp^T = (tmp U~2^T)^T, eq. 30
        self._B = c.matmul(u_hat, b_til.t()).contiguous()        # U^ b~  (n_x, n_u), on the device
        lam, w = np.linalg.eig(self._A)
        self.lambdas = lam.reshape(-1, 1)
        scale = c.matmul(yout, up(v_til @ (s_inv @ m1)))         # eq. 36 (n_x, k)
        self._pc = c.matmul(scale, up(np.hstack([w.real, w.imag]))).contiguous()   # [Re Phi | Im Phi]
        self.modes_re, self.modes_im = self._pc[:, :k], self._pc[:, k:]
        # Phi^+ = (Phi^H Phi)^+ Phi^H from the 2k x 2k Gram of Pc.  Columns of Phi are scaled to unit length first
        # (a Gram matrix squares the condition number; the scaling removes the part of it that is mere column
        # scaling); modes whose norm is below 1e-8 of the largest are rounding noise of null eigen-directions of A~
        # (n_modes beyond the rank of the data) and are dropped, as the truncated pseudo-inverse would.
        gc = c.matmul(self._pc, self._pc, trans=True).cpu().numpy()
        g = (gc[:k, :k] + gc[k:, k:]) + 1j * (gc[:k, k:] - gc[k:, :k])
        d = np.sqrt(np.maximum(np.real(np.diag(g)), 0.0))
        keep = d > 1e-8 * max(d.max(), 1e-300)
        dinv = np.where(keep, 1.0 / np.where(keep, d, 1.0), 0.0)
        g_inv = (dinv[:, None] * np.linalg.pinv(g * np.outer(dinv, dinv), hermitian=True)) * dinv[None, :]
        h = np.diag(lam) @ g_inv                                 # Lambda (Phi^H Phi)^+ ;  A = Re(Phi h Phi^H)
        self._K = np.block([[h.real, h.imag], [-h.imag, h.real]])
        self._gc = gc
        self._pcb = torch.cat([self._pc, self._B], dim=1).contiguous()   # [Pc | B]: one GEMM per prediction batch

    def _est_a_til_device(self):
        c = self._ctx
        import torch
        t = c.matmul(self._pc, torch.as_tensor(self._K, dtype=torch.float64, device=self._dev))
        return c.matmul(t, self._pc.t())

    def _predict_multiple_device(self, x_0, u_seq):
        import torch
        c, dev = self._ctx, self._dev
        x0 = torch.as_tensor(x_0, dtype=torch.float64, device=dev).reshape(-1, 1)
        useq = u_seq.detach().cpu().numpy() if _is_torch(u_seq) else np.asarray(u_seq, np.float64)
        useq = useq.reshape(self.n_u, -1).astype(np.float64)
        if x0.shape[0] != self.n_x:
            raise ValueError("x_0 must have n_x rows")           # assert_eq at dmd_rom.rs:199
        z = c.matmul(self._pc, x0, trans=True).cpu().numpy()     # Pc^T x_0           (2k, 1)
        bz = c.matmul(self._pc, self._B, trans=True).cpu().numpy()   # Pc^T B         (2k, n_u)
        n_s = useq.shape[1]
        coef = np.empty((self._K.shape[0] + self.n_u, n_s))
        for j in range(n_s):                                     # x_{j+1} = Pc (K Pc^T x_j) + B u_j
            kz = self._K @ z
            coef[:, j] = np.concatenate([kz.ravel(), useq[:, j]])
            z = self._gc @ kz + bz @ useq[:, j:j + 1]
        return c.matmul(self._pcb, torch.as_tensor(coef, dtype=torch.float64, device=dev))

    def est_a_til(self):
        if self.on_device:
            return self._est_a_til_device()
        modes = self.modes_re + 1j * self.modes_im
        return np.real(modes @ np.diag(self.lambdas.ravel()) @ np.linalg.pinv(modes))

    def est_b_til(self):
        return self._B

    def predict(self, x_0, u_input):
        if self.on_device:
            u1 = u_input.detach().cpu().numpy() if _is_torch(u_input) else np.asarray(u_input, np.float64)
            return self._predict_multiple_device(x_0, u1.reshape(self.n_u, 1))
        return self.est_a_til() @ np.asarray(x_0, np.float64).reshape(-1, 1) + self._B @ np.asarray(u_input, np.float64).reshape(-1, 1)

    def predict_multiple(self, x_0, u_seq):
        if self.on_device:
            return self._predict_multiple_device(x_0, u_seq)
        a = self.est_a_til()
        u_seq = np.asarray(u_seq, np.float64)
        x = np.asarray(x_0, np.float64).reshape(-1, 1)
        out = np.zeros((self.n_x, u_seq.shape[1]))
        for j in range(u_seq.shape[1]):
            x = a @ x + self._B @ u_seq[:, j:j + 1]
            out[:, j] = x[:, 0]
        return out


# ---- POD with interpolated mode weights ---------------------------------------------------------------
def _poly_block(x, degree):
    """build_full_vandermonde (stats_corr.rs:183-207): [x, 1] below degree 2, else [x, x_a x_b (a <= b), 1]."""
    cols = [x]
    if degree >= 2:
        d = x.shape[1]
        cols += [x[:, a:a + 1] * x[:, b:b + 1] for a in range(d) for b in range(a, d)]
    return np.hstack(cols + [np.ones((x.shape[0], 1))])


def _pinv_reg(m):  # mat_pinv, mat_utils.rs:37-53: every singular value inverted as 1 / (s + 1e-14)
    u, sv, vt = np.linalg.svd(m, full_matrices=False)
    return (vt.T * (1.0 / (sv + 1.0e-14))) @ u.T


class RbfInterp:
    """Radial-basis-function interpolant with a polynomial tail (interp_utils.rs:11-160; pyo3 ``PyRbfInterp(kernel_type,
    kernel_param, dim, poly_degree)``: 1 linear r, 2 multiquadric sqrt(1 + (eps r)^2), 3 cubic r^3, otherwise Gaussian
    exp(-(eps r)^2)).  Sized by the number of support points (snapshots), so it stays on the host; ``fit`` accepts several
    right-hand-side columns at once (PodI fits one interpolant per mode weight over the same support points)."""

    def __init__(self, kernel_type, kernel_param, dim, poly_degree):
        self.kernel_type, self.eps, self.dim, self.poly_degree = int(kernel_type), float(kernel_param), int(dim), int(poly_degree)
        self.x_known = self.coeffs = None

    def _phi(self, r):
        if self.kernel_type == 1:
            return r
        if self.kernel_type == 2:
            return np.sqrt(1.0 + (self.eps * r) ** 2)
        if self.kernel_type == 3:
            return r * r * r
        return np.exp(-((r * self.eps) ** 2))

    def _upper(self, x):
        r = np.sqrt(((x[:, None, :] - self.x_known[None, :, :]) ** 2).sum(axis=2))
        return np.hstack([self._phi(r), _poly_block(x, self.poly_degree)])

    def fit(self, x_in, y_in):
        x = np.asarray(x_in, np.float64)
        y = np.asarray(y_in, np.float64).reshape(x.shape[0], -1)
        if x.shape[1] != self.dim:
            raise ValueError("x_in must have `dim` columns")     # assert_eq at interp_utils.rs:133
        self.x_known = x.copy()
        upper = self._upper(x)
        p = upper[:, x.shape[0]:]
        kp = np.vstack([upper, np.hstack([p.T, np.zeros((p.shape[1], p.shape[1]))])])
        self.coeffs = _pinv_reg(kp) @ np.vstack([y, np.zeros((p.shape[1], y.shape[1]))])

    def predict(self, x_query):
        xq = np.asarray(x_query, np.float64)
        if xq.shape[1] != self.dim:
            raise ValueError("x_query must have `dim` columns")  # assert_eq at interp_utils.rs:149
        return self._upper(xq) @ self.coeffs


class PodI:
    """Proper orthogonal decomposition with interpolated mode weights (pod_rom.rs:36-117; pyo3 ``PyPodI(x, t, n_modes)``):
    x_data is n_snapshots x N, t is n_snapshots x dim.  Modes = right singular vectors of random_svd(x, n_modes, 10, 10) on
    the GPU; the weights of snapshot i are pinv(modes) x_i^T -- the modes are orthonormal, so that is x_i modes, one HIP
    GEMM X * modes instead of the reference's N-sized pseudo-inverse; every weight column gets a linear-kernel RBF
    interpolant over t (host, n_snapshots-sized).  ``predict(t_query)`` returns modes * w(t_query), N x 1, as a numpy
    array, or as a CUDA tensor when x_data was one."""

    def __init__(self, x_data, t, n_modes, *, seed=None, omega=None, ctx=None):
        import torch
        c = ctx or default_context()
        self._ctx, self.on_device = c, _on_gpu(x_data)
        dev = torch.device(f"cuda:{c.device}")
        x = (x_data if self.on_device else torch.as_tensor(np.asarray(x_data, np.float64))).to(device=dev, dtype=torch.float64)
        t = np.asarray(t.detach().cpu().numpy() if _is_torch(t) else t, np.float64)
        if t.shape[0] != x.shape[0]:
            raise ValueError("t and x_data must have the same number of rows")   # assert_eq at pod_rom.rs:38
        self.n_snapshots, self.n_modes, self.t_abscissa = x.shape[0], int(n_modes), t
        _u, _s, vt = c.rsvd(x, self.n_modes, 10, 10, seed=seed, omega=omega)     # pod_rom.rs:53-58
        self._modes = vt.t()                                                     # (N, n_modes), on the device
        self.mode_weights = c.matmul(x, self._modes).cpu().numpy()               # (n_snapshots, n_modes)
        self._interp = RbfInterp(1, 0.0, t.shape[1], 1)                          # pod_rom.rs:78-95
        self._interp.fit(t, self.mode_weights)

    @property
    def modes(self):
        return self._modes if self.on_device else self._modes.cpu().numpy()

    def predict(self, t_query):
        tq = np.asarray(t_query.detach().cpu().numpy() if _is_torch(t_query) else t_query, np.float64)
        if tq.ndim != 2 or tq.shape[0] != 1:
            raise ValueError("t_query must be a single row")                     # assert_eq at pod_rom.rs:105
        return self.predict_many(tq)

    def predict_many(self, t_queries):
        """Additive: N x n_q fields for n_q query rows at once (one GEMM)."""
        import torch
        w = self._interp.predict(np.asarray(t_queries, np.float64)).T           # (n_modes, n_q)
        out = self._ctx.matmul(self._modes, torch.as_tensor(np.ascontiguousarray(w), dtype=torch.float64, device=self._modes.device))
        return out if self.on_device else out.cpu().numpy()
