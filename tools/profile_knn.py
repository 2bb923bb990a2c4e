#!/usr/bin/env python3
"""Separates the scan cost of the MFMA k-NN kernel from its list-update cost: same cloud, different neighbour counts
(run under rocprofv3 --kernel-trace --stats)."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import corrla_rs_amd as cr  # noqa: E402

ctx = cr.Context(0)
n, k = int(sys.argv[1]) if len(sys.argv) > 1 else 262144, 64
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn((n, k), dtype=torch.float64, device="cuda", generator=g)
y = (x * x).sum(dim=1)
for nn in (66, 80, 160):
    ctx.grad_mat(x, y, 1, nn, x[: n // 2])
torch.cuda.synchronize()
