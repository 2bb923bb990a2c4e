#!/usr/bin/env python3
"""Active-subspace gradient stage timing: exact k-NN + local linear fits for every sample of an
n x k point cloud on the GPU (device-resident data), then fit_svd's RSVD of G / sqrt(N).  Prints one JSON line per
size with the numpy restatement timed on a bounded sample of queries beside it (BASELINE config 5 is n = 1e6, k = 64)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import corrla_rs_amd as cr  # noqa: E402
from oracle import active_ss_oracle as aso  # noqa: E402  (reported baseline only)

ctx = cr.Context(0)
k, n_nbrs = 64, 80
sizes = [int(s) for s in sys.argv[1:]] or [20000, 100000]
for n in sizes:
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((n, k), dtype=torch.float64, device="cuda", generator=g)
    w = torch.linspace(1.0, 0.05, k, dtype=torch.float64, device="cuda")
    y = torch.sin(x @ w * 0.2) + 0.05 * ((x * w) ** 2).sum(dim=1)
    ctx.grad_mat(x[:4096], y[:4096], 1, n_nbrs)      # warm-up (module load, LDS attributes)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gm, nreg = ctx.grad_mat(x, y, 1, n_nbrs, scale=1.0 / np.sqrt(n))
    torch.cuda.synchronize()
    t_grad = time.perf_counter() - t0
    t0 = time.perf_counter()
    u, s, vt = ctx.rsvd(gm, 32, 8, 10, seed=1)       # k x n, fat: the RSVD works on the n x k tall view
    torch.cuda.synchronize()
    t_svd = time.perf_counter() - t0
    # numpy restatement on a bounded sample of queries
    xs, ys = x.cpu().numpy(), y.cpu().numpy()
    est = aso.PolyGradientEstimator(xs, ys, 1, n_nbrs)
    nq = 200 if n <= 100000 else 40
    t0 = time.perf_counter()
    go = aso.create_grad_mat(est, xs[:nq])
    t_cpu = (time.perf_counter() - t0) / nq
    err = float(np.max(np.abs(gm[:, :nq].cpu().numpy() * np.sqrt(n) - go)) / np.abs(go).max())
    print(json.dumps({"workload": f"create_grad_mat {n} x {k} f64, order 1, {n_nbrs} neighbours + fit_svd rank 32 (q=8, p=10)",
                      "grad_stage_s": round(t_grad, 4), "queries_per_s": round(n / t_grad, 1),
                      "knn_pair_evals_per_s": round(float(n) * n / t_grad, 1), "rsvd_ms": round(t_svd * 1e3, 3),
                      "n_regularised": nreg, "max_rel_dev_vs_oracle_sample": err,
                      "cpu_restatement_s_per_query": round(t_cpu, 5), "cpu_restatement_est_total_s": round(t_cpu * n, 1),
                      "cpu_sample": f"{nq} queries, numpy brute-force neighbours + pinv fit"}), flush=True)
    del x, y, gm
    torch.cuda.empty_cache()
