#!/usr/bin/env python3
"""Replays one case of tools/fuzz_parity.py (same generator stream): fuzz_replay.py <seed> <case>.  Prints the deviations
of the run as configured by the environment (e.g. CORRLA_DEVICE_ROBUST_QR=0 for the host-controlled thin-Q)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import corrla_rs_amd as cr  # noqa: E402
from oracle import rsvd_oracle as orc  # noqa: E402

seed, want = int(sys.argv[1]), int(sys.argv[2])
wide = len(sys.argv) > 3 and sys.argv[3] == "wide"
rng = np.random.default_rng(seed)
ctx = cr.Context(0)
for case in range(want + 1):
    m = int(rng.integers(1, 1500 if wide else 700)); n = int(rng.integers(1, 700 if wide else 400))
    dtype = np.float64 if rng.random() < 0.5 else np.float32
    kind = rng.choice(["flat", "decay", "rankdef", "scaled"])
    a = rng.standard_normal((m, n))
    if kind == "decay":
        a = a * (rng.uniform(0.9, 0.995) ** np.arange(n))
    elif kind == "rankdef":
        r = int(rng.integers(1, max(2, min(m, n) // 2 + 1)))
        a = rng.standard_normal((m, r)) @ rng.standard_normal((r, n))
    elif kind == "scaled":
        a = a * 10.0 ** rng.uniform(-6, 6)
    a = a.astype(dtype)
    if rng.random() < 0.3:
        a = np.asfortranarray(a)
    nt = min(m, n)
    k = min(int(rng.integers(1, nt + 1)), 340 if wide else 160)
    p = int(rng.integers(0, 12)); q = int(rng.integers(0, 7))
    l = min(k + p, nt)
    om = rng.standard_normal((nt, l)).astype(dtype)
    sv = np.linalg.svd(a.astype(np.float64), compute_uv=False)
    eps = np.finfo(dtype).eps
    if sv[0] == 0 or kind == "rankdef":
        pass
    elif (sv[l - 1] / sv[0]) ** (2 * min(q, 3) + 1) < 1e4 * eps:
        continue
    hh = rng.random() < 0.33
    if case != want:
        continue
    u, s, vt = ctx.rsvd(a, k, q, p, omega=om, qr="householder" if hh else None)
    uo, so, vto = orc.random_svd(a.astype(np.float64), k, q, p, omega=om.astype(np.float64))
    rank = int(np.sum(sv > 1e-10 * sv[0]))
    d = np.abs(s.ravel().astype(np.float64) - so.ravel()) / so[0, 0]
    print("case", case, (m, n), dtype.__name__, kind, "k", k, "q", q, "p", p, "l", l, "householder" if hh else "", "rank", rank,
          "sigma_rank/sigma_1 %.2e" % (sv[rank - 1] / sv[0]))
    print("  max dS/s1 %.2e at index %d; relerr gpu %.3e oracle %.3e" % (d.max(), int(d.argmax()), orc.relerr(a, u, s, vt), orc.relerr(a, uo, so, vto)))
