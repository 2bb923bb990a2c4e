"""Host-side mirrors of the reference callers that sit on the RSVD hot path.
Every m- or n-sized decomposition goes through the GPU `rsvd`; what remains here is the k-wide algebra the
reference also does after its random_svd calls (k = n_modes, a few tens), written with numpy.

  pod_modes(x, n_modes)                       <- PodI::_modes            src/lib_math_utils/pod_rom.rs:53-58
  active_ss_fit_svd(grad_mat, n_comps, ...)   <- ActiveSsRsvd::fit_svd   src/lib_math_utils/active_subspaces.rs:233-250
  DMDc(x, u, dt, n_modes, n_iters)            <- DMDc::new               src/lib_math_utils/dmd_rom.rs:45-226
                                                 (pyo3 PyDMDc, src/lib_math_utils_py.rs:222-283)
Not built here: the kd-tree / local-regression gradient stage of active subspaces, PodI's weights
and RBF interpolation (out of scope)."""
import numpy as np

from .api import default_context

__all__ = ["pod_modes", "active_ss_fit_svd", "DMDc"]


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
