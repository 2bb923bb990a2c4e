#!/usr/bin/env python3
"""Times the l x l core SVD (device time of the small_svd phase: C^T GEMM + Jacobi + outputs) for several l and
kernel choices through rsvd on a 4096 x 4096 matrix, flat (Gaussian) and decaying spectra.
Usage: bench_core_svd.py [f32|f64 ...]   env: CORRLA_SVD=mc|lds|block|host, CORRLA_JMC_NP, CORRLA_JMC_MAX_B"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import corrla_rs_amd as cr  # noqa: E402

ctx = cr.Context(0)
dts = {"f32": torch.float32, "f64": torch.float64}
ls = [int(x) for x in os.environ.get("LS", "32,64,96,128,138,144,200,266").split(",")]
for dname in (sys.argv[1:] or ["f32", "f64"]):
    g = torch.Generator(device="cuda").manual_seed(3)
    a = torch.empty((4096, 4096), dtype=dts[dname], device="cuda")
    ctx.fill_normal(a, seed=3)
    decay = a * (0.97 ** torch.arange(4096, device="cuda", dtype=dts[dname]))
    for spec, mat in (("flat", a), ("decay", decay)):
        for l in ls:
            k = l - 10
            for mode in os.environ.get("MODES", "default,mc").split(","):
                if mode == "default":
                    os.environ.pop("CORRLA_SVD", None)
                else:
                    os.environ["CORRLA_SVD"] = mode
                best = None
                print(f"# {dname} {spec} l={l} mode={mode}", file=sys.stderr, flush=True)
                for _ in range(3):
                    u, s, vt = ctx.rsvd(mat, k, 2, 10, seed=1)
                    tm = ctx.timings()
                    best = tm["small_svd_ms"] if best is None else min(best, tm["small_svd_ms"])
                print(json.dumps({"dtype": dname, "spectrum": spec, "l": l, "mode": mode, "small_svd_ms": round(best, 3),
                                  "qr_ms": round(tm["qr_ms"], 3), "s0": float(s[0, 0])}), flush=True)
    os.environ.pop("CORRLA_SVD", None)
