#!/usr/bin/env python
"""GPU-busy time per training step from a rocprofv3 kernel trace (rocpd database): steps are delimited by a marker kernel
(one launch per step); for the last `n` steps print the wall time per step, the summed kernel durations and the time at
least one kernel was running (union of intervals) - the gap is host / launch latency.

    python tools/train_busy.py gpurun_out/x_train_prof/t_results.db [marker substring] [n]"""
import sqlite3
import sys


def main(path, marker="atl_overlap_kernel", n=8):
    db = sqlite3.connect(path)
    rows = list(db.execute("select name, start, end from kernels order by start"))
    marks = [s for name, s, e in rows if marker in name]
    if len(marks) < n + 1:
        raise SystemExit("only %d marker launches" % len(marks))
    t0, t1 = marks[-n - 1], marks[-1]
    sel = [(s, e, name) for name, s, e in rows if t0 <= s < t1]
    total = sum(e - s for s, e, _ in sel)
    busy, cur_s, cur_e = 0, None, None
    for s, e, _ in sorted(sel):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    wall = t1 - t0
    print("last %d steps: wall %.2f ms/step, kernels summed %.2f ms/step, GPU busy (union) %.2f ms/step = %.0f %% of wall, "
          "%d launches/step" % (n, wall / n / 1e6, total / n / 1e6, busy / n / 1e6, 100.0 * busy / wall, len(sel) // n))
    by = {}
    for s, e, name in sel:
        k = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        v = by.setdefault(k, [0, 0])
        v[0] += 1
        v[1] += e - s
    for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
        print("  %-60s %5.0f/step %8.1f us/step %6.2f us avg" % (k[:60], c / n, t / n / 1e3, t / c / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3]), *([int(sys.argv[3])] if len(sys.argv) > 3 else []))
