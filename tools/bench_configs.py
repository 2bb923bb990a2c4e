#!/usr/bin/env python3
"""Times random_svd on the other BASELINE.json configs (device-resident A, device RNG) and prints phase
timings.  Not the judged bench line (bench.py is); used to find per-config problems."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import corrla_rs_amd as cr  # noqa: E402

CONFIGS = {
    "C1": (1024, 1024, torch.float64, 32, 4, 8),
    "C2": (16384, 16384, torch.float32, 128, 2, 10),
    "C3q2": (65536, 4096, torch.float64, 256, 2, 10),
    "C3": (65536, 4096, torch.float64, 256, 10, 10),
    "C4shard": (1_250_000, 512, torch.float32, 64, 2, 10),
    "C4full": (10_000_000, 512, torch.float32, 64, 2, 10),     # the whole 10^7 x 512 matrix (20.5 GB) on ONE GPU
    "C5": (1_000_000, 64, torch.float64, 32, 8, 10),
    "C2x4": (32768, 32768, torch.float32, 128, 2, 10),          # 4.3 GB
    "C2x16": (65536, 65536, torch.float32, 128, 2, 10),         # 17 GB, 2^32 elements: 64-bit indexing everywhere
    "C2x16tall": (1_048_576, 4096, torch.float32, 128, 2, 10),  # same size, tall
    "C2col": (16384, 16384, torch.float32, 128, 2, 10),
}
FUSED = os.environ.get("FUSED", "0") == "1"
names = sys.argv[1:] or ["C1", "C3q2", "C4shard", "C5", "C2col"]
ctx = cr.Context(0)
for name in names:
    m, n, dt, k, q, p = CONFIGS[name]
    a = torch.empty((m, n), dtype=dt, device="cuda")
    if name == "C2col":
        a = torch.empty((n, m), dtype=dt, device="cuda").t()   # column-major storage
    ctx.fill_normal(a, seed=20241008)
    for _ in range(2):
        u, s, vt = ctx.rsvd(a, k, q, p, seed=1, fused=FUSED)
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        u, s, vt = ctx.rsvd(a, k, q, p, seed=1, fused=FUSED)
    torch.cuda.synchronize()
    dt_ms = (time.perf_counter() - t0) / reps * 1e3
    fl = cr.algorithmic_flops(m, n, k, q, p)
    tm = ctx.timings()
    eye = torch.eye(k, dtype=torch.float64, device="cuda")
    orth = (u.double().t() @ u.double() - eye).abs().max().item()
    print(json.dumps({"config": name + ("+fused" if FUSED else ""), "shape": [m, n], "dtype": str(dt), "k": k, "q": q, "p": p, "ms": round(dt_ms, 3),
                      "TFLOPs": round(fl / dt_ms / 1e9, 2), "orthU": orth, "s0": s[0, 0].item(),
                      "phases": {k_: round(v, 3) if isinstance(v, float) else v for k_, v in tm.items()}}), flush=True)
    del a, u, s, vt
    torch.cuda.empty_cache()
