#!/usr/bin/env python3
"""Times the l x l core SVD (small_svd phase) for several l through rsvd on a 4096 x 4096 matrix
(CORRLA_PROFILE_PHASES=1 must be set so that phase timings are device-synchronised)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import corrla_rs_amd as cr  # noqa: E402

ctx = cr.Context(0)
dts = {"f32": torch.float32, "f64": torch.float64}
for dname in (sys.argv[1:] or ["f32", "f64"]):
    a = torch.empty((4096, 4096), dtype=dts[dname], device="cuda")
    ctx.fill_normal(a, seed=3)
    for l in (32, 64, 96, 128, 138, 144):
        k = l - 10
        best = None
        for _ in range(4):
            ctx.rsvd(a, k, 2, 10, seed=1)
            tm = ctx.timings()
            best = tm["small_svd_ms"] if best is None else min(best, tm["small_svd_ms"])
        print(json.dumps({"dtype": dname, "l": l, "small_svd_ms": round(best, 3), "qr_ms": round(tm["qr_ms"], 3)}), flush=True)
