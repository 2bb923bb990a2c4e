"""Drop-in module name of the reference's pyo3 extension (`import corrla_rs as hrl;
hrl.rsvd(A, 4, 8, 10)`, examples/benchmark_rsvd.py:13,101; `from corrla_rs import PyDMDc`, examples/benchmark_dmd.py:12).
The RSVD hot path, its PCA caller (rpca) and the DMDc class are provided; everything is forwarded to corrla_rs_amd
(HIP, gfx950).  PyRbfInterp / PyPodI (RBF interpolation) are outside this build's scope."""
import numpy as _np

from corrla_rs_amd.api import PcaRsvd, power_iter, random_svd, rpca, rsvd  # noqa: F401
from corrla_rs_amd.callers import DMDc as _DMDc

__all__ = ["rsvd", "rpca", "random_svd", "power_iter", "PcaRsvd", "PyDMDc"]


class PyDMDc:
    """pyo3 ``PyDMDc(x, u, n_modes, n_iters)`` / ``predict(x0, u)`` (src/lib_math_utils_py.rs:254-283): DMDc with dt = 1,
    both randomized SVDs on the GPU; ``predict`` is ``DMDc::predict_multiple``."""

    def __init__(self, x_np, u_np, n_modes, n_iters):
        self.dmd = _DMDc(_np.asarray(x_np, dtype=_np.float64), _np.asarray(u_np, dtype=_np.float64), 1.0, int(n_modes),
                         int(n_iters))

    def predict(self, x0_np, u_np):
        return self.dmd.predict_multiple(_np.asarray(x0_np, dtype=_np.float64), _np.asarray(u_np, dtype=_np.float64))
