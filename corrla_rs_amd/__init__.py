"""corrla_rs_amd -- MI355X (gfx950) randomized-SVD engine behind the RSVD hot path of
wgurecky/CORRLA_RS.  The compute path is libcorrla_rsvd.so (hand-written HIP kernels); this package
is the thin host-side mirror of the reference's Python/Rust surfaces.  No CPU fallback."""
from .api import (Context, PcaRsvd, algorithmic_flops, default_context, power_iter, random_svd, rpca,  # noqa: F401
                  rsvd)

__version__ = "0.1.0"
from .callers import (ActiveSsRsvd, DMDc, FittedActiveSsRsvd, PodI, PolyGradientEstimator, RbfInterp,  # noqa: E402,F401
                      active_ss_fit_svd, pod_modes)
