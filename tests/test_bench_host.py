"""Host-side pieces of bench.py that run without a GPU: argument handling and the PMC-summary parser that feeds
`roofline.traffic` (the number must come from a tracked profiles/ file, never from a constant in bench.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_help_and_configs():
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], text=True)
    assert "--gpus" in out and "--steps" in out and "--warmup" in out and "--config" in out


def test_traffic_is_parsed_from_the_tracked_pmc_summary():
    sys.path.insert(0, ROOT)
    import bench
    traffic, src = bench.pmc_traffic()
    assert src.startswith("profiles/r") and "pmc_sketch_gemm_summary" in src
    # FETCH_SIZE x 2 + WRITE_SIZE of the sketch GEMM at C2: a little above the 1.074e9 bytes of A
    assert 1.074e9 < traffic < 1.4e9
    none, why = bench.pmc_traffic("no_such_kernel")
    assert none is None and "no tracked" in why
    import inspect
    assert "TRAFFIC_BYTES_PER_LAUNCH" not in inspect.getsource(bench)


def test_self_launch_spawns_n_ranks_and_relays_rank0(monkeypatch):
    """`python bench.py --gpus N` from a plain shell (no torch.distributed.run): the parent spawns N rank processes
    before touching the GPU, rank 0's JSON line is the parent's stdout, a failing rank fails the run."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["CORRLA_BENCH_DRYRUN"] = "1"
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--config", "C4"],
                                  text=True, env=env, timeout=120)
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d == {"dryrun": True, "world": 4, "config": "C4", "master": "127.0.0.1"}
    env["CORRLA_BENCH_DRYRUN_RC"] = "3"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, timeout=120)
    assert r.returncode == 3


def test_a_dead_rank_ends_the_whole_launch_quickly():
    """Rank 1 exits with an error at once while rank 0 'hangs' (sleeps 120 s, standing for a rank blocked in an RCCL
    collective): the parent must notice the failure, end rank 0 and return rank 1's code within seconds -- not wait for
    rank 0 first (bench.py used to wait on the ranks in order, up to 50 minutes with the GPUs held)."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(CORRLA_BENCH_DRYRUN="1", CORRLA_BENCH_DRYRUN_RC="3", CORRLA_BENCH_DRYRUN_SLEEP_RANK0="120")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, timeout=100)
    assert r.returncode == 3 and time.time() - t0 < 60


def test_sigterm_to_the_launcher_takes_the_ranks_down():
    """`timeout` (or the driver) signals the parent: no rank process may survive it."""
    import signal
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(CORRLA_BENCH_DRYRUN="1", CORRLA_BENCH_DRYRUN_SLEEP_RANK0="120", CORRLA_BENCH_TAG="sigterm-test-%d" % os.getpid())
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE)
    time.sleep(3.0)      # the ranks are up (rank 0 asleep)
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=30) == 128 + signal.SIGTERM
    time.sleep(0.5)
    tag = env["CORRLA_BENCH_TAG"].encode()
    left = []
    for pid in os.listdir("/proc"):
        if pid.isdigit():
            try:
                if tag in open(f"/proc/{pid}/environ", "rb").read():
                    left.append(pid)
            except OSError:
                pass
    assert not left, f"rank processes survived the launcher: {left}"
