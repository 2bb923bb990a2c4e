#!/usr/bin/env python3
"""Times the tall product Y = A X (gemm_nn) alone for a few widths, with hipEvents on the library's stream.
Usage: bench_gemm_nn.py m n l [l ...]   env: CORRLA_GEMM_DEBUG (timing-only ablations), CORRLA_GEMM_PERSIST_TILES"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import corrla_rs_amd as cr  # noqa: E402

m, n = int(sys.argv[1]), int(sys.argv[2])
ctx = cr.Context(0)
a = torch.empty((m, n), dtype=torch.float32, device="cuda")
ctx.fill_normal(a, seed=5)
for l in [int(x) for x in sys.argv[3:]]:
    om = torch.empty((n, l), dtype=torch.float32, device="cuda")
    ctx.fill_normal(om, seed=1)
    ctx.time_sketch(a, om, reps=3)
    ms, _ = ctx.time_sketch(a, om, reps=10)
    print(json.dumps({"m": m, "n": n, "l": l, "debug": os.environ.get("CORRLA_GEMM_DEBUG", "0"), "ms": round(ms, 4),
                      "TFLOPs": round(2.0 * m * n * l / ms / 1e9, 1), "A_TBps": round(m * n * 4 / ms / 1e9, 2)}), flush=True)
