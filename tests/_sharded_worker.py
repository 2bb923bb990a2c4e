"""Worker for tests/test_sharded_gloo.py: one rank of the row-sharded random_svd, run through the REAL
driver (driver.hpp / capi_impl.hpp) on the test-only emulation backend, with the all-reduce exchange
points served by torch.distributed (gloo) -- the same exchange points RCCL serves on GPUs."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.emu_harness import ALLREDUCE_FN, emu, emu_rsvd  # noqa: E402


def main():
    out_dir = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    calls = {"n": 0, "bytes": 0}

    def allreduce(buf, count, flags):
        is_f64, is_max = bool(flags & 1), bool(flags & 2)   # bit 1: MAX (the handshake at the start of a sharded call)
        dt = np.float64 if is_f64 else np.float32
        arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_double if is_f64 else C.c_float)), shape=(count,))
        t = torch.from_numpy(arr)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if is_max else dist.ReduceOp.SUM)  # in place (shares memory with the C buffer)
        calls["n"] += 1
        calls["bytes"] += count * np.dtype(dt).itemsize

    cb = ALLREDUCE_FN(allreduce)
    emu().corrla_emu_set_comm(cb, world)
    emu().corrla_emu_set_rank(rank)

    d = np.load(os.path.join(out_dir, "input.npz"))
    if "escalate_rank" in d.files and int(d["escalate_rank"]) == rank:
        os.environ["CORRLA_ROBUST_PASSES"] = "8"   # this rank's context starts escalated (read when the backend is created)
    a, omega = d["A"], d["omega"]
    k, q, p = int(d["k"]), int(d["q"]), int(d["p"])
    m = a.shape[0]
    lo, hi = rank * m // world, (rank + 1) * m // world
    if "splits" in d.files:          # explicit (uneven) row split: splits[r] .. splits[r + 1]
        lo, hi = int(d["splits"][rank]), int(d["splits"][rank + 1])
    if "pca" in d.files and int(d["pca"]) == 1:
        from tests.emu_harness import emu_pca
        for dtype in (np.float64, np.float32):
            for center in ("fused", "copy"):
                a_loc = np.ascontiguousarray(a[lo:hi].astype(dtype))
                means, s, comps = emu_pca(a_loc, k, q, p, omega=omega.astype(dtype), center=center, sharded=True)
                np.savez(os.path.join(out_dir, f"pca_{np.dtype(dtype).name}_{center}_rank{rank}.npz"), means=means, s=s, comps=comps,
                         n_allreduce=calls["n"])
                calls["n"] = 0
        dist.barrier()
        dist.destroy_process_group()
        return
    shard_cols = "shard_cols" in d.files and int(d["shard_cols"]) == 1
    qr = "householder" if "householder" in d.files and int(d["householder"]) == 1 else None
    # fail_rank: that rank passes ldu = m_local - 1 (a rank-LOCAL argument error); every rank must come back with an
    # error instead of blocking in its first collective
    fail_rank = int(d["fail_rank"]) if "fail_rank" in d.files else -1
    if fail_rank >= 0:
        a_loc = np.ascontiguousarray(a[lo:hi].astype(np.float64))
        try:
            emu_rsvd(a_loc, k, q, p, omega=omega, sharded=True, bad_ldu=(rank == fail_rank))
            outcome = "ok"
        except ValueError as e:
            outcome = "ValueError: " + str(e)
        except RuntimeError as e:
            outcome = "RuntimeError: " + str(e)
        with open(os.path.join(out_dir, f"outcome_rank{rank}.txt"), "w") as f:
            f.write(outcome)
        dist.barrier()
        dist.destroy_process_group()
        return
    for dtype in (np.float64, np.float32):
        if shard_cols:     # fat matrix, this rank's COLUMNS (lo/hi index the columns)
            n = a.shape[1]
            lo, hi = (int(d["splits"][rank]), int(d["splits"][rank + 1])) if "splits" in d.files else (rank * n // world, (rank + 1) * n // world)
            a_loc = np.ascontiguousarray(a[:, lo:hi].astype(dtype))
        else:
            a_loc = np.ascontiguousarray(a[lo:hi].astype(dtype))
        u, s, vt = emu_rsvd(a_loc, k, q, p, omega=omega.astype(dtype), sharded=True, shard_cols=shard_cols, qr=qr)
        np.savez(os.path.join(out_dir, f"out_{np.dtype(dtype).name}_rank{rank}.npz"), u=u, s=s, vt=vt, lo=lo, hi=hi,
                 n_allreduce=calls["n"])
        calls["n"] = 0
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
