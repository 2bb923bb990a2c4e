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
    # one rsvd call ends with column_sign_apply (round 2: column_sign + apply_column_sign x2) + copy_out x2; --call N picks
    # the N-th call
    # (default: the 5th, a timed step of the default bench run after its 3 warm-up calls)
    call = 5
    if "--call" in sys.argv:
        call = int(sys.argv[sys.argv.index("--call") + 1])
    signs = [i for i, r in enumerate(rows) if "column_sign_apply_kernel" in r[0] or ("column_sign_kernel" in r[0] and "apply" not in r[0])]
    if len(signs) < call or call < 2:
        print("not enough rsvd calls in the trace")
        return
    def call_end(i):
        # after the sign kernel: sign application, ONE tall product (U = Q U~, written in place), output copies
        j, gemms = i, 0
        while j + 1 < len(rows):
            n = rows[j + 1][0]
            if "copy_out" in n or "apply_column_sign" in n or "copyBuffer" in n:
                j += 1
            elif "gemm_tn_kernel" in n and gemms == 0:
                gemms += 1
                j += 1
            elif "slab_reduce" in n and gemms == 1:
                j += 1
            else:
                break
        return j
    first = call_end(signs[call - 2]) + 1
    last = call_end(signs[call - 1])
    seg = rows[first:last + 1]
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
