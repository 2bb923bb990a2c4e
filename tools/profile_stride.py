#!/usr/bin/env python3
"""Sketch GEMM timing vs leading dimension of A (power-of-two pitch vs padded), same logical matrix."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import corrla_rs_amd as cr  # noqa: E402
m = n = 16384
l = 138
ctx = cr.Context(0)
om = torch.empty((n, l), dtype=torch.float32, device="cuda")
ctx.fill_normal(om, seed=1)
for pad in (0, 0, 0, 64, 0, 0):
    base = torch.empty((m, n + pad), dtype=torch.float32, device="cuda")
    a = base[:, :n]
    ctx.fill_normal(a, seed=20241008)
    ms, y = ctx.time_sketch(a, om, reps=20)
    print(f"ld = n + {pad}: sketch {ms:.4f} ms  {2.0 * m * n * l / ms / 1e9:.1f} TFLOP/s", flush=True)
    del base, a
