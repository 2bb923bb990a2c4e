"""Drop-in module name of the reference's pyo3 extension (`import corrla_rs as hrl;
hrl.rsvd(A, 4, 8, 10)`, examples/benchmark_rsvd.py:13,101; `from corrla_rs import PyDMDc`, examples/benchmark_dmd.py:12).
The RSVD hot path and its callers are provided -- rpca (PCA), PyDMDc, PyPodI (with the PyRbfInterp its mode weights
are interpolated by) -- and everything m- or n-sized is forwarded to corrla_rs_amd (HIP, gfx950).  active_ss
(gradient stage on the GPU) is provided too; the samplers (cs_*_sample) of the pyo3 module are outside this build's
scope."""
import numpy as _np

from corrla_rs_amd.api import PcaRsvd, power_iter, random_svd, rpca, rsvd  # noqa: F401
from corrla_rs_amd.callers import DMDc as _DMDc
from corrla_rs_amd.callers import PodI as _PodI
from corrla_rs_amd.callers import RbfInterp as _RbfInterp

__all__ = ["rsvd", "rpca", "random_svd", "power_iter", "PcaRsvd", "PyDMDc", "PyPodI", "PyRbfInterp", "active_ss"]


def active_ss(a_mat, y, order, n_nbr, n_comps):
    """pyo3 ``active_ss(a_mat, y, order, n_nbr, n_comps)`` (src/lib_math_utils_py.rs:57-86): local polynomial gradients
    (GPU), eigendecomposition of G G^T / N (``ActiveSsRsvd::fit``); returns (components, singular values as a diagonal
    matrix, diagonal sensitivity)."""
    from corrla_rs_amd.callers import ActiveSsRsvd, PolyGradientEstimator
    x = _np.asarray(a_mat, dtype=_np.float64)
    fit = ActiveSsRsvd(PolyGradientEstimator(x, _np.asarray(y, dtype=_np.float64), int(order), int(n_nbr)), int(n_comps)).fit(x)
    return fit.components(), fit.singular_vals(), fit.var_diag_evd_sensi()


class PyRbfInterp:
    """pyo3 ``PyRbfInterp(kernel_type, kernel_param, dim, poly_degree)`` / ``fit(x, y)`` / ``predict(x)``
    (src/lib_math_utils_py.rs:178-220)."""

    def __init__(self, kernel_type, kernel_param, dim, poly_degree):
        self.rbfi = _RbfInterp(int(kernel_type), float(kernel_param), int(dim), int(poly_degree))

    def fit(self, x_np, y_np):
        self.rbfi.fit(_np.asarray(x_np, dtype=_np.float64), _np.asarray(y_np, dtype=_np.float64))

    def predict(self, x_np):
        return self.rbfi.predict(_np.asarray(x_np, dtype=_np.float64))


class PyPodI:
    """pyo3 ``PyPodI(x, t, n_modes)`` / ``predict(t)`` (src/lib_math_utils_py.rs:222-250): the randomized SVD and the
    N-sized products on the GPU."""

    def __init__(self, x_np, t_np, n_modes):
        self.pod = _PodI(_np.asarray(x_np, dtype=_np.float64), _np.asarray(t_np, dtype=_np.float64), int(n_modes))

    def predict(self, t_np):
        return self.pod.predict(_np.asarray(t_np, dtype=_np.float64))


class PyDMDc:
    """pyo3 ``PyDMDc(x, u, n_modes, n_iters)`` / ``predict(x0, u)`` (src/lib_math_utils_py.rs:254-283): DMDc with dt = 1,
    both randomized SVDs on the GPU; ``predict`` is ``DMDc::predict_multiple``."""

    def __init__(self, x_np, u_np, n_modes, n_iters):
        self.dmd = _DMDc(_np.asarray(x_np, dtype=_np.float64), _np.asarray(u_np, dtype=_np.float64), 1.0, int(n_modes),
                         int(n_iters))

    def predict(self, x0_np, u_np):
        return self.dmd.predict_multiple(_np.asarray(x0_np, dtype=_np.float64), _np.asarray(u_np, dtype=_np.float64))
