#!/usr/bin/env python3
"""bench.py -- RSVD throughput on MI355X (BASELINE.json metric: "RSVD GFLOP/s on 16k x 16k f32 rank-128;
% of MFMA peak at 1/2/4/8 GPUs").

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one complete random_svd (random_svd.rs:63-110) of the synthetic matrix, inputs resident in
HBM when the timed region starts: A ~ N(0,1) generated on device (Philox4x32-10 + Box-Muller, seed
20241008, counter = row * n + col), Omega drawn on device by the library (seed 1).
N = 1: BASELINE config 2 -- 16384 x 16384 f32, rank 128, 2 power iterations, 10 oversamples.
N > 1: the same block per GPU, row-sharded (weak scaling): rank r holds rows [16384 r, 16384 (r+1)) of the
       (16384 N) x 16384 matrix; RCCL all-reduces of the n x l and l x l factors.
value = algorithmic GFLOP/s over all ranks: ((4q+4) m n l + 2 m l^2 + (4 m l^2 - 4/3 l^3)) / step time,
with the UNPADDED l = 138; the host-side l x l SVD is inside the timed step.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

M_PER_GPU, N_COLS, RANK, N_ITER, N_OVER = 16384, 16384, 128, 2, 10
SEED_A, SEED_OMEGA = 20241008, 1
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32 matrix peak
CPU_SAMPLE = 16384             # cpu_baseline runs the oracle on the CPU_SAMPLE^2 corner (= the whole matrix)
CPU_THREADS = 16               # a 1-GPU box's CPU share; OpenBLAS is pinned to this many threads
# HBM bytes per sketch launch from rocprofv3 PMC passes on this kernel/config (not collected live):
# FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md section HBM) + WRITE_SIZE, separate --pmc runs
TRAFFIC_BYTES_PER_LAUNCH = 1.152e9 + 18.5e6 + 19.0e6 + 9.2e6  # gemm_nn FETCH x2 + WRITE, slab_reduce FETCH x2 + WRITE
TRAFFIC_SOURCE = "profiles/r01_pmc_sketch_gemm_summary.txt"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(a_dev, l):
    """Reference CPU path beside the GPU number: the oracle (numpy restatement of random_svd.rs) on the
    GPU box's host cores, bounded sample, all cores (numpy/OpenBLAS threading)."""
    import numpy as np
    from threadpoolctl import threadpool_limits
    from oracle import rsvd_oracle as orc
    s = min(CPU_SAMPLE, a_dev.shape[0], a_dev.shape[1])
    a = a_dev[:s, :s].contiguous().cpu().numpy()
    rng = np.random.default_rng(SEED_OMEGA)
    omega = rng.standard_normal((s, l)).astype(np.float32)
    avail = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        pass
    threads = min(CPU_THREADS, avail)
    with threadpool_limits(limits=threads):
        orc.random_svd(a[:1024, :1024], RANK, N_ITER, N_OVER, omega=omega[:1024])  # warm the BLAS threads
        t0 = time.perf_counter()
        uo, so, vto = orc.random_svd(a, RANK, N_ITER, N_OVER, omega=omega)
        dt = time.perf_counter() - t0
    flops = orc.algorithmic_flops(s, s, RANK, N_ITER, N_OVER)
    base = {"value": round(flops / dt / 1e9, 2), "unit": "GFLOP/s", "cores": threads, "kind": "port",
            "sample": f"oracle/rsvd_oracle.py random_svd (numpy restatement of random_svd.rs:15-110) on the {s}x{s} f32 "
                      f"matrix of the same workload, rank {RANK}, q={N_ITER}, p={N_OVER}: {dt:.2f} s on {threads} "
                      f"OpenBLAS threads ({avail} logical CPUs visible)"}
    return base, (a, omega, uo, so, vto)


def relerr_device(torch, a_dev, u, s, vt):
    """||A - U diag(S) Vt||_F / ||A||_F accumulated in f64, row blocks on the device (checker, untimed)."""
    num = den = 0.0
    us = u.double() * s.double().ravel()
    vtd = vt.double()
    for r0 in range(0, a_dev.shape[0], 2048):
        blk = a_dev[r0:r0 + 2048].double()
        num += float(((blk - us[r0:r0 + 2048] @ vtd) ** 2).sum().item())
        den += float((blk ** 2).sum().item())
    return (num / den) ** 0.5


def accuracy_gate(torch, ctx, a_dev, cpu_pack):
    """North-star parity at the full benchmark size: same A, same Omega, GPU path vs the CPU restatement."""
    a_host, omega, uo, so, vto = cpu_pack
    if a_host.shape != tuple(a_dev.shape):
        return None
    u, s, vt = ctx.rsvd(a_dev, RANK, N_ITER, N_OVER, omega=omega)
    re_gpu = relerr_device(torch, a_dev, u, s, vt)
    dev = a_dev.device
    re_cpu = relerr_device(torch, a_dev, torch.as_tensor(uo, device=dev), torch.as_tensor(so, device=dev),
                           torch.as_tensor(vto, device=dev))
    ds = float((s.double().ravel().cpu() - torch.as_tensor(so).double().ravel()).abs().max().item()) / float(so[0, 0])
    return {"relerr_gpu": re_gpu, "relerr_cpu_restatement": re_cpu, "abs_diff": abs(re_gpu - re_cpu),
            "gate_abs_diff": 1e-5, "passed": bool(abs(re_gpu - re_cpu) <= 1e-5), "max_abs_dS_over_s1": ds,
            "same_A_same_Omega": True}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import corrla_rs_amd as cr

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    dist = None
    # CORRLA_BENCH_FORCE_SHARDED=1: exercise the process-group + RCCL + row-sharded entry point at any
    # world size (used to rehearse the N > 1 code path on a 1-GPU box)
    force_sharded = os.environ.get("CORRLA_BENCH_FORCE_SHARDED", "0") == "1"
    use_dist = world > 1 or force_sharded
    if use_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    ctx = cr.Context(local_rank)
    if use_dist:
        ids = [cr.Context.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        ctx.comm_init(ids[0], rank, world)

    m_loc, n, k, q, p = M_PER_GPU, N_COLS, RANK, N_ITER, N_OVER
    l = min(k + p, n)
    a = torch.empty((m_loc, n), dtype=torch.float32, device=dev)
    ctx.fill_normal(a, seed=SEED_A, row0=rank * m_loc, global_cols=n)
    m_glob = m_loc * world
    flops = cr.algorithmic_flops(m_glob, n, k, q, p)

    def step():
        if use_dist:
            return ctx.rsvd_sharded(a, k, q, p, seed=SEED_OMEGA)
        return ctx.rsvd(a, k, q, p, seed=SEED_OMEGA)

    def fence():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        out = step()
    fence()
    sketch_ms_in_steps = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
        # device time of this step's sketch GEMM: hipEvents the library recorded on its own stream around the
        # launch; reading them back costs no GPU work and no extra synchronisation (the call has completed)
        sketch_ms_in_steps.append(ctx.timings()["sketch_kernel_ms"])
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = flops * args.steps / dt / 1e9

    # accuracy gate reported with the timing (rank-local orthonormality; full parity lives in tests/)
    u, s, vt = out
    tm = ctx.timings()

    result = None
    if rank == 0:
        # dominant kernel: the sketch GEMM Y = A * Omega (random_svd.rs:31), hipEvents on the library's stream
        om = torch.empty((n, l), dtype=torch.float32, device=dev)
        ctx.fill_normal(om, seed=SEED_OMEGA)
        # (a) the judged number: average duration of the sketch launch INSIDE the K timed steps
        sk_ms = sum(sketch_ms_in_steps) / len(sketch_ms_in_steps)
        # (b) for information: the same launch back-to-back after the chip's clocks have ramped up (the first
        #     ~25 ms of sustained load run ~15 % slower on MI355X; profiles/r01_dvfs_ramp.txt)
        ctx.time_sketch(a, om, reps=60)
        sk_warm_ms, _ = ctx.time_sketch(a, om, reps=40)
        sk_flops = 2.0 * m_loc * n * l   # algorithmic: unpadded l = 138 (the kernel computes 144 columns)
        achieved = sk_flops / (sk_ms * 1e-3) / 1e12
        roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": TRAFFIC_BYTES_PER_LAUNCH,
                    "traffic_unit": "bytes per launch (algorithmic: 1.074e9 A + 9.4e6 Omega + 1.9e7 out)",
                    "traffic_source": TRAFFIC_SOURCE + " (rocprofv3 --pmc, separate passes; not collected live)",
                    "kernel": "gemm_nn_kernel<float,2,9> + slab_reduce (sketch Y = A*Omega, 16384x16384x138)",
                    "avg_launch_ms": round(sk_ms, 4),
                    "measured": "hipEvents on the library's stream around the sketch launch of each timed step",
                    "steady_state": {"avg_launch_ms": round(sk_warm_ms, 4),
                                     "achieved": round(sk_flops / (sk_warm_ms * 1e-3) / 1e12, 2),
                                     "frac": round(sk_flops / (sk_warm_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                     "note": "same launch back-to-back after >= 60 warm launches (clock ramp)"},
                    "hbm_GBps_on_A_read": round(m_loc * n * 4 / (sk_ms * 1e-3) / 1e9, 1)}
        log(f"[bench] step {ms_per_step:.3f} ms  value {value:.0f} GFLOP/s  sketch in-step {sk_ms:.3f} ms = {achieved:.1f} TF "
            f"({100 * achieved / PEAK_F32_MFMA_TFLOPS:.1f}% of f32 MFMA peak); warmed-up {sk_warm_ms:.3f} ms = "
            f"{sk_flops / (sk_warm_ms * 1e-3) / 1e12:.1f} TF")
        log(f"[bench] last-call phases (ms): {json.dumps({k_: round(v, 3) if isinstance(v, float) else v for k_, v in tm.items()})}")
        eye = torch.eye(k, dtype=torch.float64, device=dev)
        vo = (vt.double() @ vt.double().t() - eye).abs().max().item()
        log(f"[bench] s[0]={s[0, 0].item():.3f} s[k-1]={s[-1, 0].item():.3f}  |VVt-I|max={vo:.2e}")
        result = {
            "metric": "RSVD GFLOP/s on 16k x 16k f32 rank-128",
            "value": round(value, 1), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"random_svd of a {m_glob}x{n} f32 Gaussian matrix ({m_loc}x{n} per GPU, row-sharded), "
                                   f"rank={k}, n_iter={q}, n_oversamples={p} (l={l}); BASELINE.json configs[1]",
                       "m": m_glob, "n": n, "rank": k, "n_iter": q, "n_oversamples": p,
                       "parallelism": f"row-shard x{world}" if world > 1 else "single GPU",
                       "algorithmic_flops_per_step": flops,
                       "pct_of_f32_mfma_peak_whole_job": round(100 * value / 1e3 / (PEAK_F32_MFMA_TFLOPS * world), 2)},
            "roofline": roofline,
            "phases_ms_last_step": {k_: (round(v, 3) if isinstance(v, float) else v) for k_, v in tm.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"], pack = cpu_baseline(a, l)
            result["accuracy"] = accuracy_gate(torch, ctx, a, pack)
            log(f"[bench] accuracy gate (same A, same Omega): {json.dumps(result['accuracy'])}")
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
