#!/usr/bin/env python3
"""Randomised sweep of the active-subspace gradient stage (corrla_grad_mat_f64) against the oracle: dimensions,
cloud sizes, neighbour counts, both orders, the three k-NN kernels (CORRLA_KNN = 1 / 2 / 3), both order-1 fit kernels
(CORRLA_FIT), and now and then an order-2 design too large for LDS (k up to 30: normal equations in global memory)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import corrla_rs_amd as cr  # noqa: E402
from oracle import active_ss_oracle as aso  # noqa: E402


def run(n_cases=60, seed=0, ctx=None, verbose=True):
    rng = np.random.default_rng(seed)
    ctx = ctx or cr.Context(0)
    bad = 0
    worst = 0.0
    for case in range(n_cases):
        order = int(rng.integers(1, 3))
        big2 = order == 2 and rng.random() < 0.15
        k = int(rng.integers(1, 65)) if order == 1 else (int(rng.integers(15, 31)) if big2 else int(rng.integers(1, 11)))
        need = k + 1 if order == 1 else k * (k + 3) // 2
        n_nbrs = int(rng.integers(need + 1, min(512 if big2 else 160, need + 40) + 1))
        n = int(rng.integers(n_nbrs + 5, 3000))
        x = rng.standard_normal((n, k)) * rng.uniform(0.1, 10.0) + rng.uniform(-3, 3)
        w = rng.standard_normal(k)
        y = np.sin(x @ w * 0.1) + 0.05 * (x ** 2).sum(axis=1) + rng.uniform(-5, 5)
        nq = int(rng.integers(1, 8 if big2 else 60))
        xq = x[rng.choice(n, size=nq, replace=False)] if rng.random() < 0.5 else rng.standard_normal((nq, k))
        os.environ["CORRLA_KNN"] = str(int(rng.integers(1, 4)))   # 3 = the bf16-filter scan (n_nbrs <= 128, else round 2's)
        os.environ["CORRLA_FIT"] = str(int(rng.integers(0, 2)))   # 1 = the general fit kernel for order 1 too
        try:
            g, nreg = ctx.grad_mat(x, y, order, n_nbrs, xq)
        except Exception as e:  # noqa: BLE001
            if verbose:
                print("EXCEPTION", case, order, k, n, n_nbrs, repr(e)[:160])
            bad += 1
            continue
        est = aso.PolyGradientEstimator(x, y, order, n_nbrs)
        go = aso.create_grad_mat(est, xq)
        dev = float(np.max(np.abs(g - go)) / max(np.abs(go).max(), 1e-300))
        # ill-conditioned local designs amplify the forward-difference / rounding differences of the two solvers
        tol = 1e-7 if order == 1 else 1e-3
        worst = max(worst, dev)
        if nreg == 0 and not (dev <= tol):
            if verbose:
                print("VIOLATION", case, "order", order, "k", k, "n", n, "nbrs", n_nbrs, "nq", nq, "dev %.2e" % dev)
            bad += 1
    os.environ.pop("CORRLA_KNN", None)
    os.environ.pop("CORRLA_FIT", None)
    return bad, worst


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    bad, worst = run(n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print("cases", n_cases, "violations", bad, "worst deviation %.2e" % worst)
    sys.exit(1 if bad else 0)
