#!/usr/bin/env python3
"""bench.py -- RSVD throughput on MI355X (BASELINE.json metric: "RSVD GFLOP/s on 16k x 16k f32 rank-128;
% of MFMA peak at 1/2/4/8 GPUs").

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C4]

With --gpus N > 1 and no torch.distributed environment, this process spawns the N ranks itself (one process per
GPU, 127.0.0.1 rendezvous) BEFORE anything touches the GPU and relays rank 0's JSON line; under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it is one of the ranks.

A "step" is one complete random_svd (random_svd.rs:63-110) of the synthetic matrix, everything on the device
(the l x l core SVD too), inputs resident in HBM when the timed region starts: A ~ N(0,1) generated on device
(Philox4x32-10 + Box-Muller, seed 20241008, counter = global row * n + col, so any row shard generates its own
rows), Omega drawn on device by the library (seed 1).

  --config C2 (default; BASELINE.json configs[1], the configuration the metric is quoted on):
      16384 x 16384 f32, rank 128, 2 power iterations, 10 oversamples (l = 138).
      N > 1: the same block per GPU, row-sharded -- rank r holds rows [16384 r, 16384 (r+1)) of the (16384 N) x 16384
      matrix ("scaling": "weak").
  --config C4 (BASELINE.json configs[3], the north star's ">= 6x at 8 GPUs" case):
      10,000,000 x 512 f32, rank 64, 2 power iterations, 10 oversamples (l = 74); the rows are split N ways
      ("scaling": "strong"); N = 1 runs the whole 20.5 GB matrix on one GPU.
Row-sharded runs exchange only n x l / l x l / scalar all-reduces over RCCL; their count and bytes per
step are reported.

value = algorithmic GFLOP/s over all ranks: ((4q+4) m n l + 2 m l^2 + (4 m l^2 - 4/3 l^3)) / step time, with the
UNPADDED l.  roofline = the sketch GEMM Y = A * Omega (random_svd.rs:31) of the timed steps.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (rows, cols, rank, n_iter, n_oversamples, scaling, BASELINE.json label)
    "C2": (16384, 16384, 128, 2, 10, "weak", "BASELINE.json configs[1]"),
    "C4": (10_000_000, 512, 64, 2, 10, "strong", "BASELINE.json configs[3]"),
}
SEED_A, SEED_OMEGA = 20241008, 1
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32 matrix peak
PEAK_HBM_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s achievable)
CPU_THREADS = 16               # a 1-GPU box's CPU share; OpenBLAS is pinned to this many threads


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def spawn_ranks(args):
    """--gpus N from a plain shell: start the N ranks as CHILD processes (this parent never initialises the GPU, so
    nothing that has touched the GPU is ever exec'ed or forked) and relay rank 0's stdout.  Every rank runs in its own
    session; the parent polls them all: the first non-zero exit, the deadline, or a SIGTERM / SIGINT to the parent
    (e.g. an outer `timeout`) ends the WHOLE group -- a rank that died must not leave its peers blocked in an RCCL
    collective with the GPUs held, and a killed parent must not orphan them."""
    import atexit
    import signal
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []

    def kill_all():
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)     # start_new_session: pid == process-group id
                except (ProcessLookupError, PermissionError):
                    pass
        for p in procs:
            try:
                p.wait(timeout=10)
            except Exception:
                pass

    def on_signal(signum, _frame):
        kill_all()
        os._exit(128 + signum)

    atexit.register(kill_all)
    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, on_signal)
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CORRLA_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=(None if r == 0 else subprocess.DEVNULL), start_new_session=True))
    deadline = time.time() + float(os.environ.get("CORRLA_BENCH_TIMEOUT", "3000"))
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        failed = [c for c in codes if c not in (None, 0)]
        if failed:
            rc = abs(failed[0]) if abs(failed[0]) < 256 else 1
            log(f"[bench] a rank exited with {failed[0]}: ending the other ranks")
            break
        if all(c == 0 for c in codes):
            break
        if time.time() > deadline:
            rc = 124
            log("[bench] deadline reached: ending the ranks")
            break
        time.sleep(0.05)
    kill_all()
    sys.exit(rc)


def pmc_traffic(kernel_substr="gemm_nn_kernel<float, 2, 9"):
    """HBM bytes per launch of the sketch kernel from the newest tracked rocprofv3 PMC summary under profiles/
    (separate --pmc passes, FETCH_SIZE x 2 on gfx950 + WRITE_SIZE: MI355X_MICROARCH.md, HBM section) -- the numbers are
    read from the file, never kept as constants here.  Returns (bytes or None, source)."""
    pdir = os.path.join(ROOT, "profiles")
    cands = sorted((f for f in os.listdir(pdir) if re.match(r"r\d+_pmc_sketch_gemm_summary\.txt$", f)), reverse=True) \
        if os.path.isdir(pdir) else []
    for f in cands:
        fetch = write = red_fetch = red_write = None
        for line in open(os.path.join(pdir, f)):
            m_f = re.search(r"FETCH_SIZE=(\d+)KB", line)
            m_w = re.search(r"WRITE_SIZE=(\d+)KB", line)
            if kernel_substr in line:
                fetch = float(m_f.group(1)) * 1024 * 2 if m_f else fetch
                write = float(m_w.group(1)) * 1024 if m_w else write
            elif "slab_reduce_kernel<float>" in line:
                red_fetch = float(m_f.group(1)) * 1024 * 2 if m_f else red_fetch
                red_write = float(m_w.group(1)) * 1024 if m_w else red_write
        if fetch is not None and write is not None:
            total = fetch + write + (red_fetch or 0.0) + (red_write or 0.0)
            return total, f"profiles/{f} (rocprofv3 --pmc, separate passes, FETCH_SIZE x2 + WRITE_SIZE of {kernel_substr} + slab_reduce)"
    return None, "no tracked PMC summary for this kernel under profiles/"


def cpu_baseline(a_dev, k, q, p, l, sample_rows):
    """Reference CPU path beside the GPU number: the oracle (numpy restatement of random_svd.rs) on the GPU box's host
    cores on a bounded sample (the leading `sample_rows` rows) of the same workload, CPU_THREADS OpenBLAS threads."""
    import numpy as np
    from threadpoolctl import threadpool_limits
    from oracle import rsvd_oracle as orc
    s_rows = min(sample_rows, a_dev.shape[0])
    a = a_dev[:s_rows].contiguous().cpu().numpy()
    n = a.shape[1]
    rng = np.random.default_rng(SEED_OMEGA)
    omega = rng.standard_normal((n, l)).astype(np.float32)
    avail = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        pass
    threads = min(CPU_THREADS, avail)
    with threadpool_limits(limits=threads):
        w = a[:2048, :min(n, 1024)]                                    # warm the BLAS threads on a small tall block
        orc.random_svd(w, min(k, 64), 1, p, omega=omega[:w.shape[1], :min(min(k, 64) + p, w.shape[1])])
        t0 = time.perf_counter()
        uo, so, vto = orc.random_svd(a, k, q, p, omega=omega)
        dt = time.perf_counter() - t0
    flops = orc.algorithmic_flops(s_rows, n, k, q, p)
    base = {"value": round(flops / dt / 1e9, 2), "unit": "GFLOP/s", "cores": threads, "kind": "port",
            "sample": f"oracle/rsvd_oracle.py random_svd (numpy restatement of random_svd.rs:15-110) on the leading "
                      f"{s_rows} x {n} f32 rows of the same workload, rank {k}, q={q}, p={p}: {dt:.2f} s on {threads} "
                      f"OpenBLAS threads ({avail} logical CPUs visible)"}
    return base, (a, omega, uo, so, vto)


def relerr_device(torch, a_dev, u, s, vt):
    """||A - U diag(S) Vt||_F / ||A||_F accumulated in f64, row blocks on the device (checker, untimed)."""
    num = den = 0.0
    us = u.double() * s.double().ravel()
    vtd = vt.double()
    step = max(1, (1 << 25) // a_dev.shape[1])
    for r0 in range(0, a_dev.shape[0], step):
        blk = a_dev[r0:r0 + step].double()
        num += float(((blk - us[r0:r0 + step] @ vtd) ** 2).sum().item())
        den += float((blk ** 2).sum().item())
    return (num / den) ** 0.5


def accuracy_gate(torch, ctx, a_dev, k, q, p, l, cpu_pack):
    """SURVEY.md 8d gate at the benchmark size: same A (the rows the CPU sample covers), same Omega, GPU path vs the
    CPU restatement: |relerr_gpu - relerr_cpu| <= 1e-5, max |dS| <= 1e-5 sigma_1, ||U^T U - I||_max and
    ||V V^T - I||_max <= 50 eps sqrt(l).  All four gate `passed`."""
    a_host, omega, uo, so, vto = cpu_pack
    a_s = a_dev[:a_host.shape[0]]
    u, s, vt = ctx.rsvd(a_s, k, q, p, omega=omega)
    re_gpu = relerr_device(torch, a_s, u, s, vt)
    dev = a_dev.device
    re_cpu = relerr_device(torch, a_s, torch.as_tensor(uo, device=dev), torch.as_tensor(so, device=dev),
                           torch.as_tensor(vto, device=dev))
    ds = float((s.double().ravel().cpu() - torch.as_tensor(so).double().ravel()).abs().max().item()) / float(so[0, 0])
    eye = torch.eye(k, dtype=torch.float64, device=dev)
    uo_ = float((u.double().t() @ u.double() - eye).abs().max().item())
    vo_ = float((vt.double() @ vt.double().t() - eye).abs().max().item())
    orth_gate = 50 * 1.1920929e-07 * l ** 0.5
    ok = abs(re_gpu - re_cpu) <= 1e-5 and ds <= 1e-5 and uo_ <= orth_gate and vo_ <= orth_gate
    return {"relerr_gpu": re_gpu, "relerr_cpu_restatement": re_cpu, "abs_diff": abs(re_gpu - re_cpu),
            "gate_abs_diff": 1e-5, "max_abs_dS_over_s1": ds, "gate_dS": 1e-5, "UtU_minus_I_max": uo_,
            "VVt_minus_I_max": vo_, "gate_orth": orth_gate, "passed": bool(ok), "same_A_same_Omega": True,
            "rows_compared": int(a_host.shape[0])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fused", action="store_true",
                    help="CORRLA_POWER_FUSED: one-sweep A^T (A Z) power iteration (SURVEY 8 f4; n <= 512 f32, i.e. --config C4)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)   # does not return
    if os.environ.get("CORRLA_BENCH_DRYRUN") == "1":   # launcher rehearsal without a GPU (tests/test_bench_host.py)
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({"dryrun": True, "world": int(os.environ.get("WORLD_SIZE", "1")), "config": args.config,
                              "master": os.environ.get("MASTER_ADDR")}), flush=True)
        if os.environ.get("RANK", "0") == "0":   # rehearsal of a rank that is stuck (in a collective) while a peer dies
            time.sleep(float(os.environ.get("CORRLA_BENCH_DRYRUN_SLEEP_RANK0", "0")))
        sys.exit(int(os.environ.get("CORRLA_BENCH_DRYRUN_RC", "0")) if os.environ.get("RANK") == "1" else 0)

    import torch
    import corrla_rs_amd as cr

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch {args.gpus} ranks (or none, and let bench.py spawn them)")
    dist = None
    # CORRLA_BENCH_FORCE_SHARDED=1: exercise the process-group + RCCL + row-sharded entry point at any
    # world size (used to rehearse the N > 1 code path on a 1-GPU box)
    force_sharded = os.environ.get("CORRLA_BENCH_FORCE_SHARDED", "0") == "1"
    use_dist = world > 1 or force_sharded
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    ctx = cr.Context(local_rank)
    nranks_seen = 0
    if use_dist:
        ids = [cr.Context.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        ctx.comm_init(ids[0], rank, world)
        nranks_seen = ctx.comm_info()[1]

    m_cfg, n, k, q, p, scaling, label = CONFIGS[args.config]
    if scaling == "weak":
        m_glob = m_cfg * world
        row_lo, row_hi = m_cfg * rank, m_cfg * (rank + 1)
    else:
        m_glob = m_cfg
        row_lo, row_hi = (m_cfg * rank) // world, (m_cfg * (rank + 1)) // world
    m_loc = row_hi - row_lo
    l = min(k + p, n)
    a = torch.empty((m_loc, n), dtype=torch.float32, device=dev)
    ctx.fill_normal(a, seed=SEED_A, row0=row_lo, global_cols=n)
    flops = cr.algorithmic_flops(m_glob, n, k, q, p)

    def step():
        if use_dist:
            return ctx.rsvd_sharded(a, k, q, p, seed=SEED_OMEGA, fused=args.fused)
        return ctx.rsvd(a, k, q, p, seed=SEED_OMEGA, fused=args.fused)

    def fence():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # the timed steps run without the per-phase event records (instrumentation: one hipEvent per phase boundary, ~5 us
    # of idle GPU each); the two events around the sketch launch -- roofline.achieved -- are always recorded.  The phase
    # breakdown printed below comes from ONE extra, untimed step with the phase events switched back on.
    ctx.set_phase_timings(False)
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

    ctx.set_phase_timings(True)
    out = step()   # untimed: phase breakdown only
    fence()
    u, s, vt = out
    tm = ctx.timings()
    sk_ms = sum(sketch_ms_in_steps) / len(sketch_ms_in_steps)
    sk_flops = 2.0 * m_loc * n * l   # algorithmic: unpadded l (the kernel computes 16-column tiles)
    per_rank = [{"rank": rank, "rows": m_loc, "sketch_ms": round(sk_ms, 4),
                 "sketch_TFLOPs": round(sk_flops / (sk_ms * 1e-3) / 1e12, 2) if sk_ms > 0 else None}]
    if use_dist:
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered

    if rank == 0:
        # dominant kernel: the sketch GEMM Y = A * Omega (random_svd.rs:31), hipEvents on the library's stream
        om = torch.empty((n, l), dtype=torch.float32, device=dev)
        ctx.fill_normal(om, seed=SEED_OMEGA)
        # (a) the judged number: average duration of the sketch launch INSIDE the K timed steps
        # (b) for information: the same launch back-to-back after the chip's clocks have ramped up
        ctx.time_sketch(a, om, reps=60 if args.config == "C2" else 10)
        sk_warm_ms, _ = ctx.time_sketch(a, om, reps=40 if args.config == "C2" else 10)
        achieved = sk_flops / (sk_ms * 1e-3) / 1e12
        a_gbps = m_loc * n * 4 / (sk_ms * 1e-3) / 1e9
        traffic, traffic_src = (None, "not collected for this configuration")
        if args.config == "C2":
            traffic, traffic_src = pmc_traffic()
        mw = 2
        nt = (l + 15) // 16
        kern = f"gemm_nn_kernel<float,{mw},{nt}>" + (" + slab_reduce" if args.config == "C2" else "")
        roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                    "traffic_unit": f"bytes per launch (algorithmic: {m_loc * n * 4:.4g} A + {n * l * 4:.3g} Omega + {m_loc * l * 4:.3g} out)",
                    "traffic_source": traffic_src,
                    "kernel": f"{kern} (sketch Y = A*Omega, {m_loc}x{n}x{l}, rank 0's shard)",
                    "avg_launch_ms": round(sk_ms, 4),
                    "measured": "hipEvents on the library's stream around the sketch launch of each timed step",
                    "steady_state": {"avg_launch_ms": round(sk_warm_ms, 4),
                                     "achieved": round(sk_flops / (sk_warm_ms * 1e-3) / 1e12, 2),
                                     "frac": round(sk_flops / (sk_warm_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                     "note": "same launch back-to-back after warm launches (clock ramp)"},
                    "hbm_GBps_on_A_read": round(a_gbps, 1), "hbm_frac_of_8TBps": round(a_gbps / PEAK_HBM_GBPS, 4),
                    "per_rank": per_rank}
        log(f"[bench] {args.config} x{world}: step {ms_per_step:.3f} ms  value {value:.0f} GFLOP/s  sketch in-step {sk_ms:.3f} ms = "
            f"{achieved:.1f} TF ({100 * achieved / PEAK_F32_MFMA_TFLOPS:.1f}% of f32 MFMA peak, {a_gbps:.0f} GB/s on A); "
            f"warmed-up {sk_warm_ms:.3f} ms")
        phases = {k_: (round(v, 3) if isinstance(v, float) else v) for k_, v in tm.items()}
        log(f"[bench] last-call device phases (ms): {json.dumps(phases)}")
        result = {
            "metric": "RSVD GFLOP/s on 16k x 16k f32 rank-128" if args.config == "C2" else "RSVD GFLOP/s on 10M x 512 f32 rank-64 (row-sharded)",
            "value": round(value, 1), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"random_svd of a {m_glob}x{n} f32 Gaussian matrix ({m_loc}x{n} on rank 0, row-sharded x{world}), "
                                   f"rank={k}, n_iter={q}, n_oversamples={p} (l={l}); {label}",
                       "m": m_glob, "n": n, "rank": k, "n_iter": q, "n_oversamples": p,
                       "parallelism": f"row-shard x{world}" if world > 1 else "single GPU",
                       "algorithmic_flops_per_step": flops,
                       "pct_of_f32_mfma_peak_whole_job": round(100 * value / 1e3 / (PEAK_F32_MFMA_TFLOPS * world), 2)},
            "roofline": roofline,
            "phases_ms_extra_untimed_step": phases,
            "schedule": "one-sweep A^T (A Z) (CORRLA_POWER_FUSED)" if args.fused else "reference (two products per iteration)",
            "collectives": {"rccl_nranks": nranks_seen, "allreduces_per_step": tm["n_collectives"],
                            "allreduce_bytes_per_step": tm["collective_bytes"]},
        }
        if world == 1 and not args.no_cpu_baseline:
            sample_rows = m_loc if args.config == "C2" else 1_250_000
            result["cpu_baseline"], pack = cpu_baseline(a, k, q, p, l, sample_rows)
            result["accuracy"] = accuracy_gate(torch, ctx, a, k, q, p, l, pack)
            log(f"[bench] accuracy gate (same A, same Omega): {json.dumps(result['accuracy'])}")
        else:
            result["cpu_baseline"] = None
            eye = torch.eye(k, dtype=torch.float64, device=dev)
            result["accuracy"] = {"VVt_minus_I_max": float((vt.double() @ vt.double().t() - eye).abs().max().item()),
                                  "note": "rank-local orthonormality only; parity lives in tests/ and in the N = 1 line"}
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
