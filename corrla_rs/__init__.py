"""Drop-in module name of the reference's pyo3 extension (`import corrla_rs as hrl;
hrl.rsvd(A, 4, 8, 10)`, examples/benchmark_rsvd.py:13,101).  Only the RSVD hot path and its PCA caller (rpca) are provided;
everything is forwarded to corrla_rs_amd (HIP, gfx950)."""
from corrla_rs_amd.api import PcaRsvd, power_iter, random_svd, rpca, rsvd  # noqa: F401

__all__ = ["rsvd", "rpca", "random_svd", "power_iter", "PcaRsvd"]
