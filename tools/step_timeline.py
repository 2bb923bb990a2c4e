#!/usr/bin/env python3
"""Per-kernel timeline of ONE timed step from a rocprofv3 --kernel-trace run of bench.py (rocpd .db output).

Usage: python tools/step_timeline.py gpurun_out/<dir>/<name>_results.db [--stats]
Prints every dispatch of the last full rsvd step (start ns relative, duration us, gap to the previous kernel)
and, with --stats, the per-kernel totals over the whole run."""
import re
import sqlite3
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"corrla::k::", "", n)
    n = re.sub(r"^void ", "", n)
    return n[:70]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, start, end from kernels order by start"))
    if "--stats" in sys.argv:
        tot = defaultdict(lambda: [0, 0])
        for n, s, e in rows:
            tot[short(n)][0] += 1
            tot[short(n)][1] += e - s
        allns = sum(v[1] for v in tot.values())
        print(f"{'kernel':72s} {'calls':>6s} {'total_us':>10s} {'avg_us':>9s} {'pct':>6s}")
        for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
            print(f"{n:72s} {c:6d} {t / 1e3:10.1f} {t / 1e3 / c:9.2f} {100.0 * t / allns:6.2f}")
        return
    # a step starts with the Philox fill of Omega when present, else with the first big gemm_nn after a copy_out
    big = [i for i, r in enumerate(rows) if "gemm_nn" in r[0] and (r[2] - r[1]) > 300000]
    ends = [i for i, r in enumerate(rows) if "copy_out_kernel" in r[0]]
    if not big or not ends:
        print("no rsvd step found")
        return
    last_end = ends[-1]
    prev_end = max(i for i in ends if i < last_end - 5)
    seg = rows[prev_end + 1:last_end + 1]
    t0 = seg[0][1]
    pe = seg[0][1]
    total = 0
    for n, s, e in seg:
        print(f"{(s - t0) / 1e3:9.1f} us  {short(n):72s} dur {(e - s) / 1e3:8.1f} us  gap {(s - pe) / 1e3:6.1f}")
        pe = e
        total += e - s
    print(f"step span {(seg[-1][2] - t0) / 1e3:.1f} us, kernel time {total / 1e3:.1f} us, {len(seg)} dispatches")


if __name__ == "__main__":
    main()
