#!/usr/bin/env python
"""Turn a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME`, the default output
format of ROCm 7.2) into the per-kernel summary committed under profiles/: calls, total / average
duration and share of kernel time.

    python tools/rocpd_summary.py gpurun_out/r1a_prof/r1a_results.db "title" > profiles/r01a_kernel_stats.md
"""
import sqlite3
import sys


def main(path, title=""):
    db = sqlite3.connect(path)
    rows = list(db.execute("select * from top_kernels"))
    total = sum(r[2] for r in rows)
    print("# rocprofv3 --kernel-trace --stats summary%s\n" % (": " + title if title else ""))
    print("source: `%s` (durations in microseconds; %d kernel names, %.1f ms of kernel time)\n"
          % (path, len(rows), total / 1e3))
    print("| kernel | calls | total us | avg us | % |")
    print("|---|---:|---:|---:|---:|")
    for name, calls, tot, avg, pct in rows:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")
        if len(short) > 110:
            short = short[:107] + "..."
        print("| `%s` | %d | %.1f | %.2f | %.2f |" % (short, calls, tot, avg, pct))


if __name__ == "__main__":
    main(sys.argv[1], " ".join(sys.argv[2:]))
