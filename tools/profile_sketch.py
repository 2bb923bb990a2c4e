#!/usr/bin/env python3
"""Runs only the tall products of the range finder a few times, for rocprofv3 --pmc / --kernel-trace passes:
the sketch GEMM Y = A * Omega (random_svd.rs:31) back to back, then A^T Y (the other half of the power iteration).
Usage: rocprofv3 ... -- python3 tools/profile_sketch.py [reps] [f32|f64|c4] [bf16x6|bf16x3]
  f32 (default): BASELINE config 2 (16384^2 x 138);  f64: config 3 (65536 x 4096 x 266);  c4: one 1/8 shard of config 4
  (1,250,000 x 512 x 74);  a third argument routes the products through the bf16-split kernels."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
what = sys.argv[2] if len(sys.argv) > 2 else "f32"
mixed = sys.argv[3] if len(sys.argv) > 3 else None
if mixed:
    os.environ["CORRLA_SKETCH_MIXED"] = mixed
import torch  # noqa: E402
import corrla_rs_amd as cr  # noqa: E402

dtype = torch.float64 if what == "f64" else torch.float32
m, n, l = {"f32": (16384, 16384, 138), "f64": (65536, 4096, 266), "c4": (1_250_000, 512, 74)}[what]
ctx = cr.Context(0)
a = torch.empty((m, n), dtype=dtype, device="cuda")
ctx.fill_normal(a, seed=20241008)
om = torch.empty((n, l), dtype=dtype, device="cuda")
ctx.fill_normal(om, seed=1)
ms, y = ctx.time_sketch(a, om, reps=reps)
flops = 2.0 * m * n * l
print(f"sketch {m}x{n}x{l} {dtype} {mixed or 'exact'}: {ms:.4f} ms  {flops / ms / 1e9:.1f} TFLOP/s  {m * n * a.element_size() / ms / 1e6:.0f} GB/s on A")
# transposed product too (A^T Y), the other half of the power iteration
yy = torch.empty((m, l), dtype=dtype, device="cuda")
ctx.fill_normal(yy, seed=2)
z = ctx.matmul(a, yy, trans=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    z = ctx.matmul(a, yy, trans=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps * 1e3
print(f"A^T Y (incl. staging copies): {dt:.4f} ms  {flops / dt / 1e9:.1f} TFLOP/s")
