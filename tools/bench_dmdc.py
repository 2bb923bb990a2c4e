#!/usr/bin/env python3
"""DMDc and PodI with the state-sized factors resident on the device.  Times the device-resident build
(two randomized SVDs + the n_x-sized GEMMs), the factored multi-step predictor, and -- at a size numpy finishes
quickly -- the oracle (CPU restatement of dmd_rom.rs) for reference.  One JSON line per case."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import corrla_rs_amd as cr  # noqa: E402


def main():
    ctx = cr.Context(0)
    dev = torch.device("cuda:0")
    for n_x, n_t, n_u, k, q in [(100_000, 200, 4, 20, 4), (1_000_000, 200, 4, 20, 4), (4_000_000, 100, 2, 16, 4)]:
        g = torch.Generator(device=dev).manual_seed(1)
        basis = torch.randn((n_x, k), dtype=torch.float64, device=dev, generator=g) / np.sqrt(n_x)
        rng = np.random.default_rng(0)
        az = np.diag(np.linspace(0.6, 0.98, k)) + 0.02 * np.triu(rng.standard_normal((k, k)), 1)
        bz = rng.standard_normal((k, n_u))
        u = rng.standard_normal((n_u, n_t))
        z = rng.standard_normal((k, 1))
        zs = []
        for t in range(n_t):
            zs.append(z[:, 0].copy())
            z = az @ z + bz @ u[:, t:t + 1]
        x = basis @ torch.as_tensor(np.array(zs).T, device=dev)        # (n_x, n_t), on the device
        ud = torch.as_tensor(u, device=dev)
        m = cr.DMDc(x, ud, 1.0, k + n_u, q, seed=1, ctx=ctx)          # warm-up (arena growth, clocks)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = cr.DMDc(x, ud, 1.0, k + n_u, q, seed=1, ctx=ctx)
        torch.cuda.synchronize()
        t_build = time.perf_counter() - t0
        t0 = time.perf_counter()
        pred = m.predict_multiple(x[:, 0:1], u[:, :n_t - 1])
        torch.cuda.synchronize()
        t_pred = time.perf_counter() - t0
        err = float((pred - x[:, 1:]).abs().max() / x.abs().max())
        # host-boundary variant of the same class: numpy in / numpy out (factors cross PCIe, k-wide algebra in numpy)
        t_host = None
        if n_x <= 1_000_000:
            xh = x.cpu().numpy()
            t0 = time.perf_counter()
            mh = cr.DMDc(xh, u, 1.0, k + n_u, q, seed=1, ctx=ctx)
            t_host = time.perf_counter() - t0
            del mh, xh
        out = {"case": f"DMDc n_x={n_x} n_t={n_t} n_u={n_u} n_modes={k + n_u} n_iters={q}", "build_device_s": round(t_build, 4),
               "predict_all_steps_s": round(t_pred, 4), "pred_rel_err_vs_data": err,
               "build_numpy_boundary_s": None if t_host is None else round(t_host, 4),
               "dense_A_bytes_avoided": 8 * n_x * n_x}
        print(json.dumps(out), flush=True)
        del x, m, pred
        torch.cuda.empty_cache()
    # PodI: N = 4e6 field points, 64 snapshots
    n_snap, n_pts = 64, 4_000_000
    tt = np.linspace(1.0, 9.0, n_snap).reshape(-1, 1)
    xs = torch.linspace(0.0, 10.0, n_pts, dtype=torch.float64, device=dev)
    field = (0.5 * torch.as_tensor(tt, device=dev)) * torch.exp(-((xs[None, :] - torch.as_tensor(tt, device=dev)) ** 2) / 1.5 ** 2)
    p = cr.PodI(field, tt, 8, seed=2, ctx=ctx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    p = cr.PodI(field, tt, 8, seed=2, ctx=ctx)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    y = p.predict_many(np.linspace(1.5, 8.5, 32).reshape(-1, 1))
    torch.cuda.synchronize()
    t_pred = time.perf_counter() - t0
    print(json.dumps({"case": f"PodI n_snapshots={n_snap} N={n_pts} n_modes=8", "build_device_s": round(t_build, 4),
                      "predict_32_queries_s": round(t_pred, 4), "out_shape": list(y.shape)}), flush=True)


if __name__ == "__main__":
    main()
