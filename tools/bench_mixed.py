#!/usr/bin/env python3
"""random_svd with the bf16-split range finder against the exact-f32 one (same A, same seed), per phase.
   python tools/bench_mixed.py [C2] [C4shard] [C2x4]   -> one JSON line per (config, mode)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import corrla_rs_amd as cr  # noqa: E402

CFG = {"C2": (16384, 16384, 128, 2, 10), "C4shard": (1_250_000, 512, 64, 2, 10), "C2x4": (32768, 32768, 128, 2, 10),
       "C2q4": (16384, 16384, 128, 4, 10)}


def main():
    names = [a for a in sys.argv[1:] if a in CFG] or ["C2"]
    ctx = cr.Context(0)
    for name in names:
        m, n, k, q, p = CFG[name]
        a = torch.empty((m, n), dtype=torch.float32, device="cuda")
        ctx.fill_normal(a, seed=20241008)
        flops = cr.algorithmic_flops(m, n, k, q, p)
        ref = None
        for mode in (None, "bf16x6", "bf16x3"):
            for proj in ((False,) if mode is None else (False, True)):
                os.environ["CORRLA_MIXED_PROJECT"] = "1" if proj else "0"
                ctx.set_phase_timings(False)
                for _ in range(3):
                    out = ctx.rsvd(a, k, q, p, seed=1, mixed=mode)
                torch.cuda.synchronize()
                import time
                t0 = time.perf_counter()
                reps = 10
                sk = 0.0
                for _ in range(reps):
                    out = ctx.rsvd(a, k, q, p, seed=1, mixed=mode)
                    sk += ctx.timings()["sketch_kernel_ms"]
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / reps * 1e3
                ctx.set_phase_timings(True)
                out = ctx.rsvd(a, k, q, p, seed=1, mixed=mode)
                tm = ctx.timings()
                s = out[1].double().ravel()
                if ref is None:
                    ref = s
                l = min(k + p, n)
                print(json.dumps({"config": name, "mode": mode or "f32", "project_mixed": proj, "ms": round(ms, 4),
                                  "TFLOPs_algorithmic": round(flops / ms / 1e9, 2), "sketch_kernel_ms": round(sk / reps, 4),
                                  "sketch_TFLOPs": round(2.0 * m * n * l / (sk / reps) / 1e9, 1),
                                  "sketch_A_GBps": round(m * n * 4 / (sk / reps) / 1e6, 0),
                                  "power_ms": round(tm["power_ms"], 3), "project_ms": round(tm["project_ms"], 3),
                                  "qr_ms": round(tm["qr_ms"], 3), "small_svd_ms": round(tm["small_svd_ms"], 3),
                                  "n_mixed": tm["n_mixed_products"],
                                  "max_dS_over_s1_vs_f32_run": float((s - ref).abs().max() / ref[0])}), flush=True)
        del a
    os.environ.pop("CORRLA_MIXED_PROJECT", None)


if __name__ == "__main__":
    main()
