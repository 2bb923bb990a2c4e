#!/usr/bin/env python3
"""random_svd time against the spectrum's decay: A = G * diag(d^i) (G Gaussian).  A sketch whose condition number
(sigma_1 / sigma_l)^(2 min(q,3) + 1) exceeds ~1/sqrt(eps) breaks the first Cholesky-QR pass; this shows what that costs.
Usage: bench_decay.py [f32|f64] [m n k q]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import corrla_rs_amd as cr  # noqa: E402

dt = torch.float64 if (len(sys.argv) > 1 and sys.argv[1] == "f64") else torch.float32
m, n, k, q = (int(x) for x in sys.argv[2:6]) if len(sys.argv) >= 6 else (16384, 16384, 128, 2)
p = 10
ctx = cr.Context(0)
g = torch.empty((m, n), dtype=dt, device="cuda")
ctx.fill_normal(g, seed=3)
for d in [1.0, 0.999, 0.99, 0.97, 0.9, 0.7]:
    a = g * (d ** torch.arange(n, device="cuda", dtype=dt))
    for _ in range(2):
        u, s, vt = ctx.rsvd(a, k, q, p, seed=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        u, s, vt = ctx.rsvd(a, k, q, p, seed=1)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    tm = ctx.timings()
    eye = torch.eye(k, dtype=torch.float64, device="cuda")
    orth = (u.double().t() @ u.double() - eye).abs().max().item()
    print(json.dumps({"dtype": str(dt), "shape": [m, n], "k": k, "q": q, "decay": d, "sigma_l_over_sigma_1": d ** (k + p - 1),
                      "ms": round(ms, 3), "orthU": orth,
                      "phases": {k_: round(v, 3) if isinstance(v, float) else v for k_, v in tm.items()}}), flush=True)
    del a
