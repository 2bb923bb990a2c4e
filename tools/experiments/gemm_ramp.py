"""Experiment: is the in-step sketch GEMM slower than the same launch back to back because of a clock ramp?
Runs three C2 steps, then 16 back-to-back sketch launches, then (after a host-side pause) 16 more; meant to be run under
`rocprofv3 --kernel-trace --output-format csv` -- tools/experiments/gemm_ramp_report.py prints the durations in order."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import corrla_rs_amd as cr

ctx = cr.Context()
a = torch.empty((16384, 16384), dtype=torch.float32, device="cuda")
ctx.fill_normal(a, seed=1)
x = torch.empty((16384, 138), dtype=torch.float32, device="cuda")
ctx.fill_normal(x, seed=2)
for _ in range(3):
    ctx.rsvd(a, 128, 2, 10, seed=3)
torch.cuda.synchronize()
ms, _ = ctx.time_sketch(a, x, reps=16)
print("b2b after steps:", ms)
time.sleep(0.5)
ms, _ = ctx.time_sketch(a, x, reps=16)
print("b2b after 0.5 s idle:", ms)
for _ in range(2):
    ctx.rsvd(a, 128, 2, 10, seed=3)
torch.cuda.synchronize()
