"""Prints the multi-workgroup Jacobi's own per-step timing (CORRLA_DEBUG) for both schedules at a C2-like and a C3-like core.
usage: jmc_debug.py [f32|f64] [max_b ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import corrla_rs_amd as cr

def run(dtype, m, n, k, p, local, max_b):
    os.environ["CORRLA_JMC_LOCAL"] = local
    if max_b:
        os.environ["CORRLA_JMC_MAX_B"] = str(max_b)
    ctx = cr.Context()
    a = torch.empty((m, n), dtype=dtype, device="cuda")
    ctx.fill_normal(a, seed=5)
    os.environ.pop("CORRLA_DEBUG", None)
    for _ in range(3):
        ctx.rsvd(a, k, 2, p, seed=3)
    tm = ctx.last_timings() if hasattr(ctx, "last_timings") else None
    os.environ["CORRLA_DEBUG"] = os.environ.get("DEBUG_LEVEL", "1")
    sys.stderr.write(f"== local={local} max_b={max_b} {dtype} l={k + p} timings={tm}\n")
    sys.stderr.flush()
    ctx.rsvd(a, k, 2, p, seed=3)
    os.environ.pop("CORRLA_DEBUG", None)

which = sys.argv[1] if len(sys.argv) > 1 else "f32"
bs = [int(v) for v in sys.argv[2:]] or [0]
for mb in bs:
    for local in ("1", "0"):
        if which == "f32":
            run(torch.float32, 8192, 4096, 128, 10, local, mb)
        else:
            run(torch.float64, 8192, 2048, 256, 10, local, mb)
