"""CPU oracle for the callers that sit on the RSVD hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatements of
  PodI::_modes                      src/lib_math_utils/pod_rom.rs:53-58
  ActiveSsRsvd::fit_svd (RSVD part) src/lib_math_utils/active_subspaces.rs:233-250
  DMDc::new / _calc_dmdc_modes / _calc_eigs / _calc_modes / est_a_til / predict_multiple
                                    src/lib_math_utils/dmd_rom.rs:45-226
each on top of oracle.rsvd_oracle.random_svd.  Pinned by the reference's own test_dmdc
(dmd_rom.rs:233-310: 20-step prediction within 5e-2 for (nx, nt) = (20, 40), (50, 40), (500, 40)); fit_svd has
no reference test (parity unpinned) and PodI's test asserts nothing."""
import numpy as np

from . import rsvd_oracle as orc


def pod_modes(x_data, n_modes, omega=None):
    """pod_rom.rs:53-58: (_u, _s, v) = random_svd(x_data, n_modes, 10, 10); return v^T (N x n_modes)."""
    _u, _s, vt = orc.random_svd(np.asarray(x_data, np.float64), n_modes, 10, 10, omega=omega)
    return vt.T.copy()


def active_ss_fit_svd(grad_mat, n_comps, n_iter=8, n_oversamples=10, omega=None):
    """active_subspaces.rs:233-250 given the k x N gradient matrix (create_grad_mat is SURVEY 8 f2):
    scale by 1/sqrt(N), random_svd(., min(k, n_comps), n_iter, n_oversamples); components = U, singular
    values returned as a diagonal matrix (mat_colvec_to_diag)."""
    g = np.asarray(grad_mat, np.float64)
    k_features, n_samples = g.shape
    u, s, _vt = orc.random_svd(g / np.sqrt(float(n_samples)), min(k_features, n_comps), n_iter, n_oversamples, omega=omega)
    return u, np.diag(s.ravel())


def pinv_diag(d):
    """mat_pinv_diag (mat_utils.rs:386-402): |v| < 1e-20 -> 0 else 1 / (v + 1e-20)."""
    out = np.zeros_like(d)
    for i in range(d.shape[1]):
        v = d[i, i]
        out[i, i] = 0.0 if abs(v) < 1e-20 else 1.0 / (v + 1e-20)
    return out


def mat_pinv_comp(x):
    """mat_pinv_comp (mat_utils.rs:56-72): V diag(1 / (s_i + (1e-16 + 1e-16 i))) U^H over ALL singular values."""
    u, s, vh = np.linalg.svd(np.asarray(x, np.complex128), full_matrices=False)
    return (vh.conj().T * (1.0 / (s + (1.0e-16 + 1.0e-16j)))) @ u.conj().T


def mat_pinv(x):
    """mat_pinv (mat_utils.rs:37-53): V diag(1 / (s_i + 1e-14)) U^T over ALL singular values."""
    u, s, vt = np.linalg.svd(np.asarray(x, np.float64), full_matrices=False)
    return (vt.T * (1.0 / (s + 1.0e-14))) @ u.T


class DMDcOracle:
    def __init__(self, x_data, u_data, dt, n_modes, n_iters, omega_x=None, omega_y=None):
        x_data = np.asarray(x_data, np.float64)
        u_data = np.asarray(u_data, np.float64)
        self.n_x, self.n_u, self.n_modes = x_data.shape[0], u_data.shape[0], n_modes
        om = np.vstack([x_data, u_data])                      # dmd_rom.rs:66
        xin = om[:, :-1]                                      # _X  :148-153
        yout = om[: self.n_x, 1:]                             # _Y  :156-162
        u_til, s_til, vt_til = orc.random_svd(xin, n_modes, n_iters, 12, omega=omega_x)   # :72
        v_til = vt_til.T
        u1, u2 = u_til[: self.n_x], u_til[self.n_x:]          # :75-79
        u_hat, _s, _v = orc.random_svd(yout, n_modes, n_iters, 12, omega=omega_y)         # :82
        s_inv = pinv_diag(np.diag(s_til.ravel()))             # :86-87
        tmp = u_hat.T @ yout @ v_til @ s_inv                  # :90-94
        self.a_til = tmp @ u1.T @ u_hat                       # :95-97
        b_til = tmp @ u2.T                                    # :100-102
        self.b_op = u_hat @ b_til                             # :106
        lam, w = np.linalg.eig(self.a_til)                    # :115
        self.lambdas = lam.reshape(-1, 1)
        scale = yout @ (v_til @ (s_inv @ (u1.T @ u_hat)))     # :134-139
        self.modes = scale @ w                                # :140-145 (re + i im)

    def est_a_til(self):                                      # :165-176
        return np.real(self.modes @ np.diag(self.lambdas.ravel()) @ mat_pinv_comp(self.modes))

    def est_b_til(self):
        return self.b_op

    def predict_multiple(self, x0, u_seq):                    # :197-225
        a = self.est_a_til()
        x = np.asarray(x0, np.float64).reshape(-1, 1)
        out = np.zeros((self.n_x, u_seq.shape[1]))
        for j in range(u_seq.shape[1]):
            x = a @ x + self.b_op @ u_seq[:, j:j + 1]
            out[:, j] = x[:, 0]
        return out


def dmdc_reference_test_data(nx, nt):
    """The synthetic data of run_test_dmdc (dmd_rom.rs:243-270)."""
    xp = np.linspace(0.0, 10.0, nx)
    tp = np.linspace(0.0, 10.0, nt)
    u_seq = np.exp(0.2 * tp)
    snaps = np.sin(xp[:, None] + 0.2 * tp[None, :]) * u_seq[None, :]
    return snaps, u_seq.reshape(1, -1)


# ---- POD with interpolated mode weights (pod_rom.rs:36-117) and its RBF interpolant (interp_utils.rs:11-160) ----------
RBF_KERNELS = {
    1: lambda r, eps: r,                                    # RbfKernelLin        interp_utils.rs:37-41
    2: lambda r, eps: np.sqrt(1.0 + (eps * r) ** 2),        # RbfKernelMultiQuad  :63-67
    3: lambda r, eps: r * r * r,                            # RbfKernelCubic      :50-54
    4: lambda r, eps: np.exp(-((r * eps) ** 2)),            # RbfKernelGauss      :76-80
}


class RbfInterpOracle:
    """RbfInterp (interp_utils.rs:82-160); kernel ids as in PyRbfInterp::new (lib_math_utils_py.rs:187-198: 1 linear,
    2 multiquadric, 3 cubic, anything else Gaussian)."""

    def __init__(self, kernel_type, kernel_param, dim, poly_degree):
        self.kernel = RBF_KERNELS.get(int(kernel_type), RBF_KERNELS[4])
        self.eps, self.dim, self.poly_degree = float(kernel_param), int(dim), int(poly_degree)

    def _build_p(self, x):   # build_full_vandermonde, stats_corr.rs:183-196: [x, 1] below degree 2, else the full quadratic
        x = np.asarray(x, np.float64)
        if self.poly_degree < 2:
            return np.hstack([x, np.ones((x.shape[0], 1))])
        from .active_ss_oracle import build_vandermonde
        return build_vandermonde(x, True)

    def _build_kp(self, x, full):   # :96-129
        x = np.asarray(x, np.float64)
        r = np.sqrt(((x[:, None, :] - self.x_known[None, :, :]) ** 2).sum(axis=2))
        upper = np.hstack([self.kernel(r, self.eps), self._build_p(x)])
        if not full:
            return upper
        pt = self._build_p(x).T
        return np.vstack([upper, np.hstack([pt, np.zeros((pt.shape[0], pt.shape[0]))])])

    def fit(self, x, y):            # :131-144
        x = np.asarray(x, np.float64)
        assert x.shape[1] == self.dim
        self.x_known = x.copy()
        kp = self._build_kp(x, True)
        rhs = np.vstack([np.asarray(y, np.float64).reshape(-1, 1), np.zeros((kp.shape[0] - x.shape[0], 1))])
        self.coeffs = mat_pinv(kp) @ rhs

    def predict(self, xq):          # :146-153
        xq = np.asarray(xq, np.float64)
        assert xq.shape[1] == self.dim
        return self._build_kp(xq, False) @ self.coeffs


class PodIOracle:
    """PodI::new / predict (pod_rom.rs:36-117): modes = V of random_svd(x, n_modes, 10, 10); weights row i =
    pinv(modes) x_i^T; one linear-kernel RBF interpolant (dim = t.ncols, polynomial degree 1) per mode weight."""

    def __init__(self, x_data, t, n_modes, omega=None):
        x = np.asarray(x_data, np.float64)
        t = np.asarray(t, np.float64)
        assert t.shape[0] == x.shape[0]
        self.n_modes = int(n_modes)
        self.modes = pod_modes(x, n_modes, omega=omega)                       # :53-58  (N x n_modes)
        self.mode_weights = (mat_pinv(self.modes) @ x.T).T                    # :61-75  (n_snapshots x n_modes)
        self.interp = []
        for j in range(self.n_modes):                                         # :78-95
            f = RbfInterpOracle(1, 0.0, t.shape[1], 1)
            f.fit(t, self.mode_weights[:, j:j + 1])
            self.interp.append(f)

    def predict(self, t_query):                                               # :103-116
        tq = np.asarray(t_query, np.float64)
        assert tq.shape[0] == 1
        w = np.array([[f.predict(tq)[0, 0]] for f in self.interp])
        return self.modes @ w


def pod_reference_test_data(nx=100, n_snapshots=20, sigma=0.25):
    """The pressure-field snapshots of test_pod (pod_rom.rs:127-150): row n = 0.5 t_n exp(-(x - t_n)^2 / sigma^2)."""
    xp = np.linspace(0.0, 10.0, nx)
    tp = np.linspace(1.0, 9.0, n_snapshots)
    snaps = (0.5 * tp[:, None]) * np.exp(-((xp[None, :] - tp[:, None]) ** 2) / sigma ** 2)
    return snaps, tp.reshape(-1, 1)
