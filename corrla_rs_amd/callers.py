"""Host-side mirrors of the reference callers that sit on the RSVD hot path.
Every m- or n-sized decomposition goes through the GPU `rsvd`; what remains here is the k-wide algebra the
reference also does after its random_svd calls (k = n_modes, a few tens), written with numpy.

  pod_modes(x, n_modes)                       <- PodI::_modes            src/lib_math_utils/pod_rom.rs:53-58
  active_ss_fit_svd(grad_mat, n_comps, ...)   <- ActiveSsRsvd::fit_svd   src/lib_math_utils/active_subspaces.rs:233-250
  DMDc(x, u, dt, n_modes, n_iters)            <- DMDc::new               src/lib_math_utils/dmd_rom.rs:45-226
                                                 (pyo3 PyDMDc, src/lib_math_utils_py.rs:222-283)
  PolyGradientEstimator / ActiveSsRsvd / FittedActiveSsRsvd
                                              <- src/lib_math_utils/active_subspaces.rs:21-277 (the
                                                 neighbour search and the local fits run on the GPU)
Not built here: PodI's weights and RBF interpolation (out of scope)."""
import numpy as np

from .api import default_context, rsvd

__all__ = ["pod_modes", "active_ss_fit_svd", "DMDc", "PolyGradientEstimator", "ActiveSsRsvd", "FittedActiveSsRsvd"]


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
        x = np.asarray(x_mat, dtype=np.float64)
        g = self.grad_est.grad_mat(x)
        c = (g @ g.T) * (1.0 / x.shape[0])                                           # :256
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

    def est_a_til(self):
        modes = self.modes_re + 1j * self.modes_im
        return np.real(modes @ np.diag(self.lambdas.ravel()) @ np.linalg.pinv(modes))

    def est_b_til(self):
        return self._B

    def predict(self, x_0, u_input):
        return self.est_a_til() @ np.asarray(x_0, np.float64).reshape(-1, 1) + self._B @ np.asarray(u_input, np.float64).reshape(-1, 1)

    def predict_multiple(self, x_0, u_seq):
        a = self.est_a_til()
        u_seq = np.asarray(u_seq, np.float64)
        x = np.asarray(x_0, np.float64).reshape(-1, 1)
        out = np.zeros((self.n_x, u_seq.shape[1]))
        for j in range(u_seq.shape[1]):
            x = a @ x + self._B @ u_seq[:, j:j + 1]
            out[:, j] = x[:, 0]
        return out
