#!/usr/bin/env python3
"""what the bf16-split range finder does to the RESULT.  For Gaussian and decaying spectra, every mode
(exact f32, bf16x6, bf16x3, each with and without the projection B = Q^T A on the split kernels) against the f64 oracle on
the same A and the same Omega, next to the CPU restatement run in f32 (what the reference algorithm gets in that
arithmetic).  One JSON line per (spectrum, seed, mode):  python tools/mixed_accuracy.py > profiles/..."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import corrla_rs_amd as cr  # noqa: E402
from oracle import rsvd_oracle as orc  # noqa: E402


def matrix(rng, m, n, decay):
    if decay is None:
        return rng.standard_normal((m, n)).astype(np.float32)
    u, _ = np.linalg.qr(rng.standard_normal((m, n)))
    v, _ = np.linalg.qr(rng.standard_normal((n, n)))
    return ((u * (decay ** np.arange(n))) @ v.T).astype(np.float32)


def main():
    os.environ["CORRLA_MIXED_MIN_WORK"] = "1"
    ctx = cr.Context(0)
    m, n, k, q, p = 4096, 1024, 128, 2, 10
    for decay in (None, 0.995, 0.99, 0.97, 0.9, 0.7):
        for seed in (17, 18):
            rng = np.random.default_rng(seed)
            a = matrix(rng, m, n, decay)
            om = rng.standard_normal((n, k + p)).astype(np.float32)
            ref = orc.random_svd(a.astype(np.float64), k, q, p, omega=om.astype(np.float64))
            re_ref = orc.relerr(a, *ref)

            def dev(usv):
                u, s, vt = usv
                ds = float(np.max(np.abs(s.ravel().astype(np.float64) - ref[1].ravel())) / ref[1][0, 0])
                return ds, abs(orc.relerr(a, u, s, vt) - re_ref)

            rows = [("cpu_f32_restatement", dev(orc.random_svd(a, k, q, p, omega=om)))]
            for mode in (None, "bf16x6", "bf16x3"):
                for proj in ((False,) if mode is None else (False, True)):
                    os.environ["CORRLA_MIXED_PROJECT"] = "1" if proj else "0"
                    rows.append(((mode or "gpu_f32") + ("+proj" if proj else ""), dev(ctx.rsvd(a, k, q, p, omega=om, mixed=mode))))
            for name, (ds, dre) in rows:
                print(json.dumps({"spectrum": "gaussian" if decay is None else f"{decay}^i", "seed": seed, "shape": [m, n], "k": k, "q": q,
                                  "p": p, "mode": name, "max_dS_over_s1_vs_f64_oracle": ds, "abs_d_relerr_vs_f64_oracle": dre,
                                  "relerr_f64_oracle": re_ref}), flush=True)
    os.environ.pop("CORRLA_MIXED_PROJECT", None)


if __name__ == "__main__":
    main()
