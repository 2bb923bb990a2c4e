#!/usr/bin/env python3
"""PCA caller (PcaRsvd::new, pca_rsvd.rs:56-82) timing on device-resident data: n_samples x n_dim f64/f32,
rank k, the reference's hard-coded q = 20, p = min(n_dim, 10).  Prints one JSON line per config."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import corrla_rs_amd as cr  # noqa: E402

ctx = cr.Context(0)
for (m, n, dt, k) in ((100000, 1024, torch.float64, 16), (1000000, 256, torch.float32, 16), (16384, 16384, torch.float32, 32)):
    x = torch.empty((m, n), dtype=dt, device="cuda")
    ctx.fill_normal(x, seed=11)
    x *= torch.linspace(3.0, 0.2, n, device="cuda", dtype=dt)
    for center in ("copy", "fused"):
        for _ in range(2):
            out = ctx.pca(x, k, seed=3, center=center)
        torch.cuda.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            means, s, comps = ctx.pca(x, k, seed=3, center=center)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        q, p = 20, min(n, 10)
        fl = cr.algorithmic_flops(m, n, k, q, p) + 2.0 * m * n  # + means and centring
        ev = (s.double() ** 2 / (m - 1)).ravel()[:3].tolist()
        print(json.dumps({"workload": f"PcaRsvd::new {m}x{n} {str(dt).split('.')[-1]} rank {k} (q=20, p={p})",
                          "center": center, "ms": round(ms, 3), "GFLOPs": round(fl / ms / 1e6, 1),
                          "explained_var_top3": [round(v, 4) for v in ev]}), flush=True)
    del x
    torch.cuda.empty_cache()
