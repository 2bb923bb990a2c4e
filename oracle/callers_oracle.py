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
        return np.real(self.modes @ np.diag(self.lambdas.ravel()) @ np.linalg.pinv(self.modes))

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
