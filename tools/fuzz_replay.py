#!/usr/bin/env python3
"""Replays single cases of tools/fuzz_parity.py (same generator stream):
     fuzz_replay.py <seed>:<case>[:wide] [...]   [--modes default,householder]
For each case and mode prints the deviation of the GPU result from the f64 oracle next to the deviation of the SAME
reference algorithm run in the case's own precision on the CPU (oracle/rsvd_oracle.py in f32: LAPACK Householder QR) --
what a reference-faithful implementation in that arithmetic gets.  The run is configured by the environment (e.g.
CORRLA_DEVICE_ROBUST_QR=0 for the host-controlled thin-Q)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import corrla_rs_amd as cr  # noqa: E402
from oracle import rsvd_oracle as orc  # noqa: E402
from tools.fuzz_parity import cases  # noqa: E402


def dev(a, usv, ref):
    u, s, vt = usv
    uo, so, vto = ref
    d = np.abs(s.ravel().astype(np.float64) - so.ravel()) / so[0, 0]
    return float(d.max()), int(d.argmax()), abs(orc.relerr(a, u, s, vt) - orc.relerr(a, uo, so, vto))


def main():
    modes = ["default", "householder"]
    specs = []
    args = sys.argv[1:]
    while args:
        x = args.pop(0)
        if x == "--modes":
            modes = args.pop(0).split(",")
        else:
            specs.append(x)
    ctx = cr.Context(0)
    for spec in specs:
        parts = spec.split(":")
        seed, want, wide = int(parts[0]), int(parts[1]), len(parts) > 2 and parts[2] == "wide"
        for case, a, om, k, q, p, l, kind, dtype, hh in cases(want + 1, seed, wide):
            if case != want:
                continue
            ref = orc.random_svd(a.astype(np.float64), k, q, p, omega=om.astype(np.float64))
            same = orc.random_svd(a, k, q, p, omega=om)      # the reference algorithm in the case's own precision (CPU)
            ds0, at0, re0 = dev(a, same, ref)
            print(f"case {seed}:{case} {a.shape} {np.dtype(dtype).name} {kind} k {k} q {q} p {p} l {l} (sweep ran it in "
                  f"{'householder' if hh else 'default'} mode)")
            print(f"  CPU restatement in {np.dtype(dtype).name} vs f64 oracle : max dS/s1 {ds0:.2e} at {at0}; |d relerr| {re0:.2e}")
            for mode in modes:
                usv = ctx.rsvd(a, k, q, p, omega=om, qr="householder" if mode == "householder" else None)
                ds, at, re = dev(a, usv, ref)
                print(f"  GPU {mode:11s} vs f64 oracle              : max dS/s1 {ds:.2e} at {at}; |d relerr| {re:.2e}")


if __name__ == "__main__":
    main()
