#!/usr/bin/env python3
"""f64 random_svd (the dtype of the reference's Python surface) under rocprofv3 --kernel-trace --stats."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import corrla_rs_amd as cr  # noqa: E402

ctx = cr.Context(0)
a = torch.empty((16384, 8192), dtype=torch.float64, device="cuda")
ctx.fill_normal(a, seed=1)
for _ in range(6):
    u, s, vt = ctx.rsvd(a, 128, 2, 10, seed=1)
torch.cuda.synchronize()
