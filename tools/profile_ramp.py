#!/usr/bin/env python3
"""How the sketch GEMM's duration evolves under sustained load and after idle gaps (DVFS ramp)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import corrla_rs_amd as cr  # noqa: E402
m = n = 16384
l = 138
ctx = cr.Context(0)
a = torch.empty((m, n), dtype=torch.float32, device="cuda")
ctx.fill_normal(a, seed=20241008)
om = torch.empty((n, l), dtype=torch.float32, device="cuda")
ctx.fill_normal(om, seed=1)
def burst(tag, groups, reps):
    out = []
    for _ in range(groups):
        ms, _y = ctx.time_sketch(a, om, reps=reps)
        out.append(round(ms, 3))
    print(tag, out, flush=True)
burst("cold, 12 groups of 4:", 12, 4)
for gap in (0.001, 0.003, 0.010, 0.050, 0.5):
    time.sleep(gap)
    burst(f"after {gap*1e3:.0f} ms idle, 6 groups of 4:", 6, 4)
# a small single-workgroup kernel-like gap: run tiny kernels for 3 ms instead of sleeping
small = torch.empty((64, 64), device="cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.003:
    small.add_(1.0)
torch.cuda.synchronize()
burst("after 3 ms of tiny kernels, 6 groups of 4:", 6, 4)
