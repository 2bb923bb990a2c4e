"""Isolates which stage loses orthogonality of U at tall-skinny shapes (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import corrla_rs_amd as cr
ctx = cr.Context(0)
def orth(x):
    x = x.double()
    k = x.shape[1]
    g = torch.zeros((k, k), dtype=torch.float64, device=x.device)
    for r0 in range(0, x.shape[0], 1 << 18):
        b = x[r0:r0 + (1 << 18)]
        g += b.t() @ b
    return float((g - torch.eye(k, dtype=torch.float64, device=x.device)).abs().max())
for m in (200_000, 1_250_000):
    a = torch.empty((m, 512), dtype=torch.float32, device="cuda")
    ctx.fill_normal(a, seed=11)
    for env in ({}, {"CORRLA_NO_GRAM_ALIAS": "1"}, {"CORRLA_SVD": "host"}, {"CORRLA_SVD": "host", "CORRLA_NO_GRAM_ALIAS": "1"},
                {"CORRLA_JACOBI_NOREPLAY": "1"}):
        for k_, v_ in env.items():
            os.environ[k_] = v_
        u, s, vt = ctx.rsvd(a, 64, 2, 10, seed=3)
        print(m, env, "orthU %.2e orthV %.2e" % (orth(u), orth(vt.t().contiguous())), flush=True)
        for k_ in env:
            os.environ.pop(k_)
