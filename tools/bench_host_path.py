import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import corrla_rs_amd as cr
ctx = cr.Context(0)
rng = np.random.default_rng(0)
a = rng.standard_normal((16384, 16384), dtype=np.float32)
for _ in range(2):
    u, s, vt = ctx.rsvd(a, 128, 2, 10, seed=1)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); u, s, vt = ctx.rsvd(a, 128, 2, 10, seed=1); ts.append(time.perf_counter() - t0)
fl = cr.algorithmic_flops(16384, 16384, 128, 2, 10)
t = sorted(ts)[len(ts)//2]
print("host-pointer path (numpy in, numpy out, pageable memory): median %.1f ms = %.1f TFLOP/s PCIe-inclusive; A = 1.07 GB" % (t*1e3, fl/t/1e12))
