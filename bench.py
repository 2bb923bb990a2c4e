#!/usr/bin/env python3
"""bench.py -- RSVD throughput on MI355X (BASELINE.json metric: "RSVD GFLOP/s on 16k x 16k f32 rank-128;
% of MFMA peak at 1/2/4/8 GPUs").

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C3|C4|C5] [--mixed bf16x6|bf16x3]

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
  --config C3 (BASELINE.json configs[2], POD-by-RSVD): 65536 x 4096 f64, rank 256, PodI's schedule q = 10, p = 10
      (pod_rom.rs:56; l = 266), dtype f64, roofline against the f64 MFMA peak.  N > 1: weak, like C2.
  --config C4 (BASELINE.json configs[3], the north star's ">= 6x at 8 GPUs" case):
      10,000,000 x 512 f32, rank 64, 2 power iterations, 10 oversamples (l = 74); the rows are split N ways
      ("scaling": "strong"); N = 1 runs the whole 20.5 GB matrix on one GPU.
  --config C5 (BASELINE.json configs[4], active-subspace sensitivity): 1,000,000 x 64 f64 samples; a step = the gradient
      stage (exact k-NN + local linear fits, active_subspaces.rs:66-141,215-229) + fit_svd's RSVD of the gradient matrix
      (:233-250); value = samples / s.  N = 1 only.
  --mixed bf16x6 | bf16x3 (C2 / C4): the same step with the range finder's tall products on the bf16-split kernels
      (off by default).  A SECOND line for information: the judged line is the exact-f32 one.
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
    # name: (rows, cols, rank, n_iter, n_oversamples, scaling, BASELINE.json label, dtype)
    "C2": (16384, 16384, 128, 2, 10, "weak", "BASELINE.json configs[1]", "f32"),
    "C3": (65536, 4096, 256, 10, 10, "weak", "BASELINE.json configs[2] (PodI schedule: q = 10, p = 10, pod_rom.rs:56)", "f64"),
    "C4": (10_000_000, 512, 64, 2, 10, "strong", "BASELINE.json configs[3]", "f32"),
    "C5": (1_000_000, 64, 32, 8, 10, "weak", "BASELINE.json configs[4]", "f64"),
}
SEED_A, SEED_OMEGA = 20241008, 1
PEAK_MFMA_TFLOPS = {"f32": 157.3, "f64": 78.6, "bf16": 2500.0}   # MI355X_MICROARCH.md: dense f32 / f64 / bf16 matrix peaks
PEAK_F32_MFMA_TFLOPS = PEAK_MFMA_TFLOPS["f32"]
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


def pmc_traffic(kernel_substrs=("gemm_nn_kernel<float, 2, 9",), summary=r"r\d+_pmc_sketch_gemm_summary\.txt$",
                reduce_substr="slab_reduce_kernel<float>"):
    """HBM bytes per launch of the sketch kernel(s) from the newest tracked rocprofv3 PMC summary under profiles/
    (separate --pmc passes, FETCH_SIZE x 2 on gfx950 + WRITE_SIZE: MI355X_MICROARCH.md, HBM section) -- the numbers are
    read from the file, never kept as constants here.  Several kernels (an uneven column blocking runs two launches
    per product) are summed.  Returns (bytes or None, source)."""
    if isinstance(kernel_substrs, str):
        kernel_substrs = (kernel_substrs,)
    pdir = os.path.join(ROOT, "profiles")
    cands = sorted((f for f in os.listdir(pdir) if re.match(summary, f)), reverse=True) if os.path.isdir(pdir) else []
    for f in cands:
        fetch = {k_: None for k_ in kernel_substrs}
        write = {k_: None for k_ in kernel_substrs}
        red_fetch = red_write = None
        for line in open(os.path.join(pdir, f)):
            m_f = re.search(r"FETCH_SIZE=(\d+)KB", line)
            m_w = re.search(r"WRITE_SIZE=(\d+)KB", line)
            hit = next((k_ for k_ in kernel_substrs if k_ in line), None)
            if hit is not None:
                fetch[hit] = float(m_f.group(1)) * 1024 * 2 if m_f else fetch[hit]
                write[hit] = float(m_w.group(1)) * 1024 if m_w else write[hit]
            elif reduce_substr and reduce_substr in line:
                red_fetch = float(m_f.group(1)) * 1024 * 2 if m_f else red_fetch
                red_write = float(m_w.group(1)) * 1024 if m_w else red_write
        if all(v is not None for v in fetch.values()) and all(v is not None for v in write.values()):
            total = sum(fetch.values()) + sum(write.values()) + (red_fetch or 0.0) + (red_write or 0.0)
            return total, (f"profiles/{f} (rocprofv3 --pmc, separate passes, FETCH_SIZE x2 + WRITE_SIZE of "
                           f"{' + '.join(kernel_substrs)}" + (" + slab_reduce" if reduce_substr else "") + ")")
    return None, "no tracked PMC summary for this kernel under profiles/"


def measured_f64_ceiling():
    """What a register-only v_mfma_f64_16x16x4_f64 stream sustains on this part (tools/microbench/mfma_chain.hip), parsed
    from the newest tracked profiles/rNN_mfma_register_only_ceiling.txt; None when absent."""
    pdir = os.path.join(ROOT, "profiles")
    for f in sorted((f for f in os.listdir(pdir) if re.match(r"r\d+_mfma_register_only_ceiling\.txt$", f)), reverse=True):
        best = None
        for line in open(os.path.join(pdir, f)):
            m_ = re.match(r"f64 16x16x4\b.*?([\d.]+) TFLOP/s", line)
            if m_:
                best = max(best or 0.0, float(m_.group(1)))
        if best:
            return best, f"profiles/{f}"
    return None, None


def host_threads():
    avail = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        pass
    return min(CPU_THREADS, avail), avail


def cpu_baseline(a_dev, k, q, p, l, sample_rows, all_cores=True):
    """Reference CPU path beside the GPU number: the oracle (numpy restatement of random_svd.rs) on the GPU box's host
    cores on a bounded sample (the leading `sample_rows` rows) of the same workload: CPU_THREADS OpenBLAS threads (a
    1-GPU box's CPU share) -- the judged figure -- and, for information, once more on every visible core."""
    import numpy as np
    from threadpoolctl import threadpool_limits
    from oracle import rsvd_oracle as orc
    s_rows = min(sample_rows, a_dev.shape[0])
    a = a_dev[:s_rows].contiguous().cpu().numpy()
    n = a.shape[1]
    rng = np.random.default_rng(SEED_OMEGA)
    omega = rng.standard_normal((n, l)).astype(a.dtype)
    threads, avail = host_threads()
    flops = orc.algorithmic_flops(s_rows, n, k, q, p)
    with threadpool_limits(limits=threads):
        w = a[:2048, :min(n, 1024)]                                    # warm the BLAS threads on a small tall block
        orc.random_svd(w, min(k, 64), 1, p, omega=omega[:w.shape[1], :min(min(k, 64) + p, w.shape[1])])
        t0 = time.perf_counter()
        uo, so, vto = orc.random_svd(a, k, q, p, omega=omega)
        dt = time.perf_counter() - t0
    base = {"value": round(flops / dt / 1e9, 2), "unit": "GFLOP/s", "cores": threads, "kind": "port",
            "sample": f"oracle/rsvd_oracle.py random_svd (numpy restatement of random_svd.rs:15-110) on the leading "
                      f"{s_rows} x {n} {a.dtype.name} rows of the same workload, rank {k}, q={q}, p={p}: {dt:.2f} s on {threads} "
                      f"OpenBLAS threads ({avail} logical CPUs visible)"}
    if all_cores and avail > threads and dt < 30.0:
        with threadpool_limits(limits=avail):
            t0 = time.perf_counter()
            orc.random_svd(a, k, q, p, omega=omega)
            dt_all = time.perf_counter() - t0
        base["all_cores"] = {"value": round(flops / dt_all / 1e9, 2), "unit": "GFLOP/s", "cores": avail,
                             "note": f"the same sample with OpenBLAS on every visible logical CPU: {dt_all:.2f} s"}
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


def accuracy_gate(torch, ctx, a_dev, k, q, p, l, cpu_pack, mixed=None):
    """SURVEY.md 8d gate at the benchmark size: same A (the rows the CPU sample covers), same Omega, GPU path vs the
    CPU restatement: |relerr_gpu - relerr_cpu| <= 1e-5, max |dS| <= 1e-5 sigma_1 (f64: 1e-10), ||U^T U - I||_max and
    ||V V^T - I||_max <= 50 eps sqrt(l).  All four gate `passed`."""
    a_host, omega, uo, so, vto = cpu_pack
    a_s = a_dev[:a_host.shape[0]]
    u, s, vt = ctx.rsvd(a_s, k, q, p, omega=omega, mixed=mixed)
    re_gpu = relerr_device(torch, a_s, u, s, vt)
    dev = a_dev.device
    re_cpu = relerr_device(torch, a_s, torch.as_tensor(uo, device=dev), torch.as_tensor(so, device=dev),
                           torch.as_tensor(vto, device=dev))
    ds = float((s.double().ravel().cpu() - torch.as_tensor(so).double().ravel()).abs().max().item()) / float(so[0, 0])
    eye = torch.eye(k, dtype=torch.float64, device=dev)
    uo_ = float((u.double().t() @ u.double() - eye).abs().max().item())
    vo_ = float((vt.double() @ vt.double().t() - eye).abs().max().item())
    f64 = a_host.dtype.itemsize == 8
    eps = 2.220446049250313e-16 if f64 else 1.1920929e-07
    orth_gate = 50 * eps * l ** 0.5
    gate_ds = 1e-10 if f64 else 1e-5
    ok = abs(re_gpu - re_cpu) <= 1e-5 and ds <= gate_ds and uo_ <= orth_gate and vo_ <= orth_gate
    return {"relerr_gpu": re_gpu, "relerr_cpu_restatement": re_cpu, "abs_diff": abs(re_gpu - re_cpu),
            "gate_abs_diff": 1e-5, "max_abs_dS_over_s1": ds, "gate_dS": gate_ds, "UtU_minus_I_max": uo_,
            "VVt_minus_I_max": vo_, "gate_orth": orth_gate, "passed": bool(ok), "same_A_same_Omega": True,
            "rows_compared": int(a_host.shape[0])}


def run_c5(args):
    """BASELINE config 5: active-subspace sensitivity at 1,000,000 x 64 f64 samples.  A step = ActiveSsRsvd::create_grad_mat
    (exact k-NN + local linear least-squares gradients, active_subspaces.rs:66-141, 215-229) + fit_svd's RSVD of the
    k x N gradient matrix (:233-250; rank 32, q = 8, p = 10), everything on the device."""
    import numpy as np
    import torch
    import corrla_rs_amd as cr
    n_pts, kf, rank, q, p = CONFIGS["C5"][0], CONFIGS["C5"][1], CONFIGS["C5"][2], CONFIGS["C5"][3], CONFIGS["C5"][4]
    n_nbrs = 80
    steps = args.steps if args.steps is not None else 2
    warmup = args.warmup if args.warmup is not None else 1
    dev = torch.device("cuda:0")
    ctx = cr.Context(0)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((n_pts, kf), dtype=torch.float64, device=dev, generator=g)
    w = torch.linspace(1.0, 0.05, kf, dtype=torch.float64, device=dev)
    y = torch.sin(x @ w * 0.2) + 0.05 * ((x * w) ** 2).sum(dim=1)

    def step():
        gm, nreg = ctx.grad_mat(x, y, 1, n_nbrs, scale=1.0 / np.sqrt(n_pts))
        u, s, vt = ctx.rsvd(gm, rank, q, p, seed=SEED_OMEGA)     # k x N, fat: the RSVD works on the N x k tall view
        return gm, nreg, u, s, vt

    ctx.grad_mat(x[:4096], y[:4096], 1, n_nbrs)                   # code-object load, LDS attributes
    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        gm, nreg, u, s, vt = step()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    ms_per_step = dt / steps * 1e3
    t1 = time.perf_counter()
    ctx.grad_mat(x, y, 1, n_nbrs, scale=1.0 / np.sqrt(n_pts))
    torch.cuda.synchronize(dev)
    t_grad = time.perf_counter() - t1
    tm = ctx.timings()   # hipEvents the library records on its own stream around the scan and around the fits
    knn_ms, fit_ms = float(tm["knn_ms"]), float(tm["fit_ms"])
    # Dominant kernel: the k-NN scan (knn2_kernels.hpp).  Its filter evaluates every (query, point) pair once as a
    # bf16x3 product on v_mfma_f32_16x16x32_bf16 (hi hi + hi lo + lo hi of the centred coordinates, the dimension padded
    # to a multiple of 32): algorithmic flops = N * N_q * 2 * k_pad * 3, priced against the dense bf16 MFMA peak.
    k_pad = -(-kf // 32) * 32
    pair_flops = float(n_pts) * n_pts * 2 * k_pad * 3
    achieved = pair_flops / (knn_ms * 1e-3) / 1e12 if knn_ms > 0 else 0.0
    roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_MFMA_TFLOPS["bf16"], "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_MFMA_TFLOPS["bf16"], 4), "traffic": None,
                "traffic_source": "not collected for this configuration (the scan re-reads a 0.5 GB bf16 image of the cloud "
                                  "once per query tile through L2 / LDS-DMA; it is issue-bound, DESIGN.md section 7)",
                "kernel": "knn2_kernel<2> (+ its prep kernels): exact k-NN, bf16x3 MFMA filter on centred coordinates, every "
                          "survivor re-checked in f64",
                "avg_launch_ms": round(knn_ms, 2),
                "algorithmic": f"N x N_q = {n_pts}^2 point pairs x 2 x {k_pad} flop x 3 bf16 products of the split operands",
                "pair_evaluations_per_s": round(float(n_pts) * n_pts / (knn_ms * 1e-3), 1) if knn_ms > 0 else None,
                "fit_kernel_ms": round(fit_ms, 2)}
    from oracle import active_ss_oracle as aso
    from threadpoolctl import threadpool_limits
    threads, avail = host_threads()
    nq = 24
    xs, ys = x.cpu().numpy(), y.cpu().numpy()
    with threadpool_limits(limits=threads):
        est = aso.PolyGradientEstimator(xs, ys, 1, n_nbrs)
        t0 = time.perf_counter()
        go = aso.create_grad_mat(est, xs[:nq])
        t_cpu = (time.perf_counter() - t0) / nq
    err = float(np.max(np.abs(gm[:, :nq].cpu().numpy() * np.sqrt(n_pts) - go)) / np.abs(go).max())
    result = {
        "metric": "active-subspace samples/s on 1M x 64 f64 (gradient stage + fit_svd RSVD)",
        "value": round(n_pts / (ms_per_step * 1e-3), 1), "unit": "samples/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
        "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"ActiveSsRsvd: create_grad_mat of {n_pts} x {kf} f64 samples (order 1, {n_nbrs} neighbours) + fit_svd "
                               f"(RSVD of the {kf} x {n_pts} gradient matrix, rank {rank}, q={q}, p={p}); {CONFIGS['C5'][6]}",
                   "n_samples": n_pts, "n_features": kf, "n_nbrs": n_nbrs, "rank": rank, "parallelism": "single GPU"},
        "roofline": roofline,
        "phases_ms": {"gradient_stage": round(t_grad * 1e3, 2), "knn_scan": round(knn_ms, 2), "local_fits": round(fit_ms, 2),
                      "fit_svd_rsvd": round(ms_per_step - t_grad * 1e3, 2)},
        "cpu_baseline": {"value": round(1.0 / t_cpu, 3), "unit": "samples/s", "cores": threads, "kind": "port",
                         "sample": f"oracle/active_ss_oracle.py create_grad_mat (numpy restatement of active_subspaces.rs:66-141,"
                                   f"215-229) on {nq} queries against the same {n_pts}-point cloud: {t_cpu:.3f} s per query on "
                                   f"{threads} OpenBLAS threads ({avail} logical CPUs visible)"},
        "accuracy": {"max_rel_dev_of_gradients_vs_oracle_sample": err, "gate": 1e-10, "passed": bool(err <= 1e-10),
                     "n_regularised": int(nreg), "queries_compared": nq},
    }
    log(f"[bench] C5: step {ms_per_step:.1f} ms (gradient stage {t_grad * 1e3:.1f} ms: scan {knn_ms:.1f} ms = {achieved:.0f} TF bf16, fits {fit_ms:.1f} ms), "
        f"{result['value']:.0f} samples/s; gradients vs oracle {err:.2e}")
    print(json.dumps(result), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fused", action="store_true",
                    help="CORRLA_POWER_FUSED: one-sweep A^T (A Z) power iteration (SURVEY 8 f4; n <= 512 f32, i.e. --config C4)")
    ap.add_argument("--mixed", choices=["bf16x6", "bf16x3"], default=None,
                    help="CORRLA_SKETCH_BF16X6 / X3: the range finder's tall products on the bf16-split kernels (SURVEY 8 f4; "
                         "f32 configs).  Informational second line: the judged line is the exact-f32 one")
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

    if args.config == "C5":
        if args.gpus != 1:
            raise SystemExit("--config C5 is a single-GPU line (the sharded form is tests' fit_svd_sharded)")
        return run_c5(args)
    # Defaults.  The untimed warm-up is sized in TIME, not in steps: the first process on a cold board measures ~3 %
    # slower than the same command a few seconds later whatever it runs (profiles/r03_warmup_experiment.txt: C2 4.91 ms
    # with 3 warm-up steps on a fresh box, 4.78 / 4.77 / 4.74 ms with 50 / 200 / 1000, 4.78 ms with 3 again afterwards),
    # and under a sustained run of tall products the board's power management takes ~10 ms to settle
    # (tools/experiments/gemm_ramp.py).  ~0.25-0.6 s of warm-up per configuration; explicit --steps / --warmup win.
    if args.steps is None:
        args.steps = {"C2": 50, "C4": 20, "C3": 5}.get(args.config, 20)
    if args.warmup is None:
        args.warmup = {"C2": 50, "C4": 10, "C3": 5}.get(args.config, 3)

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

    m_cfg, n, k, q, p, scaling, label, dtname = CONFIGS[args.config]
    if args.mixed and dtname != "f32":
        raise SystemExit("--mixed applies to the f32 configurations (C2, C4)")
    tdt = torch.float32 if dtname == "f32" else torch.float64
    esz = 4 if dtname == "f32" else 8
    peak = PEAK_MFMA_TFLOPS[dtname]
    if scaling == "weak":
        m_glob = m_cfg * world
        row_lo, row_hi = m_cfg * rank, m_cfg * (rank + 1)
    else:
        m_glob = m_cfg
        row_lo, row_hi = (m_cfg * rank) // world, (m_cfg * (rank + 1)) // world
    m_loc = row_hi - row_lo
    l = min(k + p, n)
    a = torch.empty((m_loc, n), dtype=tdt, device=dev)
    ctx.fill_normal(a, seed=SEED_A, row0=row_lo, global_cols=n)
    flops = cr.algorithmic_flops(m_glob, n, k, q, p)

    def step():
        if use_dist:
            return ctx.rsvd_sharded(a, k, q, p, seed=SEED_OMEGA, fused=args.fused, mixed=args.mixed)
        return ctx.rsvd(a, k, q, p, seed=SEED_OMEGA, fused=args.fused, mixed=args.mixed)

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
        om = torch.empty((n, l), dtype=tdt, device=dev)
        ctx.fill_normal(om, seed=SEED_OMEGA)
        # (a) the judged number: average duration of the sketch launch INSIDE the K timed steps
        # (b) for information: the same launch back-to-back after the chip's clocks have ramped up
        if args.mixed:
            os.environ["CORRLA_SKETCH_MIXED"] = args.mixed       # the timing hook takes no opts: environment
        reps_a, reps_b = (60, 40) if args.config == "C2" else (10, 10)
        ctx.time_sketch(a, om, reps=reps_a)
        sk_warm_ms, _ = ctx.time_sketch(a, om, reps=reps_b)
        os.environ.pop("CORRLA_SKETCH_MIXED", None)
        achieved = sk_flops / (sk_ms * 1e-3) / 1e12
        a_gbps = m_loc * n * esz / (sk_ms * 1e-3) / 1e9
        alg_bytes = m_loc * n * esz + n * l * esz + m_loc * l * esz
        tname = "float" if dtname == "f32" else "double"
        nt_tiles = (l + 15) // 16
        traffic, traffic_src = (None, "not collected for this configuration")
        if args.mixed:
            np_ = 3 if args.mixed == "bf16x6" else 2
            kern = f"gemm_bf16s_kernel<{nt_tiles},{np_},false> + split_planes + slab_reduce"
            traffic, traffic_src = pmc_traffic((f"gemm_bf16s_kernel<{nt_tiles}, {np_}, false>",), r"r\d+_pmc_mixed_gemm_summary\.txt$")
        elif args.config == "C2":
            kern = f"gemm_nn_kernel<{tname},2,{nt_tiles}> + slab_reduce"
            traffic, traffic_src = pmc_traffic()
        elif args.config == "C3":
            # f64: eight MFMA waves of one row tile each on the 128-index tile (round 3; CORRLA_F64_WAVES=4: <double, 2, NT>)
            w8 = os.environ.get("CORRLA_F64_WAVES", "8") != "4"
            names = ("gemm_nn_kernel<double, 1, 9, false, 8>", "gemm_nn_kernel<double, 1, 8, false, 8>") if w8 else \
                    ("gemm_nn_kernel<double, 2, 9", "gemm_nn_kernel<double, 2, 8")
            kern = " + ".join(n_ if n_.endswith(">") else n_ + ">" for n_ in names) + " (17 column tiles = 9 + 8, two launches)"
            traffic, traffic_src = pmc_traffic(names, r"r\d+_pmc_f64_gemm_summary\.txt$", reduce_substr=None)
        else:
            kern = f"gemm_nn_kernel<{tname},2,{nt_tiles}>"
            traffic, traffic_src = pmc_traffic((f"gemm_nn_kernel<{tname}, 2, {nt_tiles}",), r"r\d+_pmc_c4_gemm_summary\.txt$",
                                               reduce_substr=None)
            if traffic is not None:
                # the PMC passes ran on ONE 1/8 row shard (1,250,000 rows, tools/profile_sketch.py c4); the kernel streams its
                # rows once, so the bytes of this launch scale with the rows this rank holds
                traffic = traffic * (m_loc / 1_250_000.0)
                traffic_src += f"; measured on a 1,250,000-row shard, scaled by rows to this rank's {m_loc}"
        steady = {"avg_launch_ms": round(sk_warm_ms, 4), "note": "same launch back-to-back after warm launches (clock ramp)"}
        if args.mixed:
            # the split kernels are no longer bound by the matrix pipe: the roof is the stream of A from HBM
            ach_gbps = alg_bytes / (sk_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "achieved": round(ach_gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                        "frac": round(ach_gbps / PEAK_HBM_GBPS, 4), "traffic": traffic,
                        "f32_equivalent_TFLOPs": round(achieved, 2)}
            steady.update({"achieved": round(alg_bytes / (sk_warm_ms * 1e-3) / 1e9, 1),
                           "frac": round(alg_bytes / (sk_warm_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4)})
        else:
            roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(achieved / peak, 4), "traffic": traffic}
            steady.update({"achieved": round(sk_flops / (sk_warm_ms * 1e-3) / 1e12, 2),
                           "frac": round(sk_flops / (sk_warm_ms * 1e-3) / 1e12 / peak, 4)})
            if dtname == "f64":
                ceil, ceil_src = measured_f64_ceiling()
                if ceil:
                    roofline["measured_stream_ceiling"] = {"TFLOPs": ceil, "frac_of_it": round(achieved / ceil, 4), "source": ceil_src,
                                                           "note": "best register-only v_mfma_f64_16x16x4_f64 stream on this part (two waves per SIMD; one wave: 60.5)"}
        roofline.update({
            "traffic_unit": f"bytes per launch (algorithmic: {m_loc * n * esz:.4g} A + {n * l * esz:.3g} Omega + {m_loc * l * esz:.3g} out)",
            "traffic_source": traffic_src,
            "kernel": f"{kern} (sketch Y = A*Omega, {m_loc}x{n}x{l}, rank 0's shard)",
            "avg_launch_ms": round(sk_ms, 4),
            "measured": "hipEvents on the library's stream around the sketch launch of each timed step",
            "steady_state": steady,
            "hbm_GBps_on_A_read": round(a_gbps, 1), "hbm_frac_of_8TBps": round(a_gbps / PEAK_HBM_GBPS, 4),
            "per_rank": per_rank})
        log(f"[bench] {args.config} x{world}{' ' + args.mixed if args.mixed else ''}: step {ms_per_step:.3f} ms  value {value:.0f} GFLOP/s  "
            f"sketch in-step {sk_ms:.3f} ms = {achieved:.1f} TF ({100 * achieved / peak:.1f}% of {dtname} MFMA peak, {a_gbps:.0f} GB/s on A); "
            f"warmed-up {sk_warm_ms:.3f} ms")
        phases = {k_: (round(v, 3) if isinstance(v, float) else v) for k_, v in tm.items()}
        log(f"[bench] last-call device phases (ms): {json.dumps(phases)}")
        metric = {"C2": "RSVD GFLOP/s on 16k x 16k f32 rank-128", "C3": "RSVD GFLOP/s on 65536 x 4096 f64 rank-256 (POD schedule q=10)",
                  "C4": "RSVD GFLOP/s on 10M x 512 f32 rank-64 (row-sharded)"}[args.config]
        result = {
            "metric": metric,
            "value": round(value, 1), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": dtname if not args.mixed else f"f32 operands split into bf16 pieces ({args.mixed}), f32 accumulate -- "
                                                   "informational line, the judged line is exact f32",
            "data": "synthetic",
            "config": {"workload": f"random_svd of a {m_glob}x{n} {dtname} Gaussian matrix ({m_loc}x{n} on rank 0, row-sharded x{world}), "
                                   f"rank={k}, n_iter={q}, n_oversamples={p} (l={l}); {label}",
                       "m": m_glob, "n": n, "rank": k, "n_iter": q, "n_oversamples": p,
                       "parallelism": f"row-shard x{world}" if world > 1 else "single GPU",
                       "algorithmic_flops_per_step": flops,
                       f"pct_of_{dtname}_mfma_peak_whole_job": round(100 * value / 1e3 / (peak * world), 2)},
            "roofline": roofline,
            "phases_ms_extra_untimed_step": phases,
            "schedule": ("one-sweep A^T (A Z) (CORRLA_POWER_FUSED)" if args.fused else "reference (two products per iteration)") +
                        (f"; range finder on the {args.mixed} kernels ({tm['n_mixed_products']} products)" if args.mixed else ""),
            "collectives": {"rccl_nranks": nranks_seen, "allreduces_per_step": tm["n_collectives"],
                            "allreduce_bytes_per_step": tm["collective_bytes"]},
        }
        if world == 1 and not args.no_cpu_baseline:
            sample_rows = {"C2": m_loc, "C3": 16384, "C4": 1_250_000}[args.config]
            result["cpu_baseline"], pack = cpu_baseline(a, k, q, p, l, sample_rows)
            result["accuracy"] = accuracy_gate(torch, ctx, a, k, q, p, l, pack, mixed=args.mixed)
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
