#!/usr/bin/env python3
"""Runs only the sketch GEMM Y = A * Omega (random_svd.rs:31) of BASELINE config 2 a few times, for
rocprofv3 --pmc / --kernel-trace passes.  Usage: rocprofv3 ... -- python3 tools/profile_sketch.py [reps] [dtype]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import corrla_rs_amd as cr  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dtype = torch.float64 if len(sys.argv) > 2 and sys.argv[2] == "f64" else torch.float32
m = n = 16384
l = 138
if dtype == torch.float64:
    m, n, l = 65536, 4096, 266
ctx = cr.Context(0)
a = torch.empty((m, n), dtype=dtype, device="cuda")
ctx.fill_normal(a, seed=20241008)
om = torch.empty((n, l), dtype=dtype, device="cuda")
ctx.fill_normal(om, seed=1)
ms, y = ctx.time_sketch(a, om, reps=reps)
flops = 2.0 * m * n * l
print(f"sketch {m}x{n}x{l} {dtype}: {ms:.4f} ms  {flops / ms / 1e9:.1f} TFLOP/s")
# transposed product too (A^T Y), the other half of the power iteration
yy = torch.empty((m, l), dtype=dtype, device="cuda")
ctx.fill_normal(yy, seed=2)
import time
z = ctx.matmul(a, yy, trans=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    z = ctx.matmul(a, yy, trans=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps * 1e3
print(f"A^T Y (incl. staging copies): {dt:.4f} ms  {flops / dt / 1e9:.1f} TFLOP/s")
