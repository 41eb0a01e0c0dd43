#!/usr/bin/env python
"""Timeline statistics of a `rocprofv3 --kernel-trace --output-format csv` run of the timed mode (`bench.py --only-timed`):
per HW queue the busy share and the gaps between consecutive kernels, over all queues the concurrency histogram, and per
kernel family the average duration in the mix.  What the four-frames-in-flight schedule really does with the chip.

    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python3 bench.py --only-timed --steps 60
    python tools/timeline_stats.py out > profiles/r05_timeline.md
"""
import csv
import glob
import os
import re
import sys


def main(directory, skip_frac=0.35):
    paths = glob.glob(os.path.join(directory, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for p in paths:
        with open(p, newline="") as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"]))
    rows.sort()
    t0, t1 = rows[0][0], rows[-1][1]
    cut = t0 + int((t1 - t0) * skip_frac)          # drop the warm-up / capture part: keep the steady replay phase
    rows = [r for r in rows if r[0] >= cut]
    t0, t1 = rows[0][0], rows[-1][1]
    span = t1 - t0
    print("# Timeline of the timed mode (hipGraph replays, 4 frames in flight), steady phase: %.1f ms, %d dispatches\n" % (span / 1e6, len(rows)))
    queues = {}
    for s, e, q, n in rows:
        queues.setdefault(q, []).append((s, e, n))
    print("| HW queue | dispatches | busy share | median gap us | mean gap us | gaps > 5 us |")
    print("|---|---|---|---|---|---|")
    for q, ev in sorted(queues.items()):
        busy = sum(e - s for s, e, _ in ev)
        gaps = sorted(max(0, ev[i + 1][0] - ev[i][1]) for i in range(len(ev) - 1))
        if not gaps:
            continue
        print("| %s | %d | %.2f | %.2f | %.2f | %d |" % (q, len(ev), busy / span, gaps[len(gaps) // 2] / 1e3, sum(gaps) / len(gaps) / 1e3,
                                                   sum(1 for g in gaps if g > 5000)))
    # concurrency histogram: time with k kernels running
    events = []
    for s, e, _, _ in rows:
        events.append((s, 1))
        events.append((e, -1))
    events.sort()
    hist, cur, last = {}, 0, events[0][0]
    for t, d in events:
        hist[cur] = hist.get(cur, 0) + (t - last)
        cur += d
        last = t
    print("\n| kernels running at once | share of time |")
    print("|---|---|")
    for k in sorted(hist):
        print("| %d | %.3f |" % (k, hist[k] / span))
    fam = {}
    for s, e, _, n in rows:
        m = re.search(r"(conv_igemm\w*<[^>]*>|conv_\w+|wino_\w+|roi_\w+|nms_\w+|topk_\w+|\w+_kernel)", n)
        key = m.group(1) if m else n.split("(")[0][:40]
        c, d = fam.get(key, (0, 0))
        fam[key] = (c + 1, d + (e - s))
    tot = sum(d for _, d in fam.values())
    print("\n| kernel | dispatches | avg us in the mix | share of kernel time |")
    print("|---|---|---|---|")
    for k, (c, d) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:18]:
        print("| `%s` | %d | %.1f | %.3f |" % (k, c, d / c / 1e3, d / tot))
    print("\nsum of dispatch durations / wall = %.2f kernels in flight on average" % (tot / span))


if __name__ == "__main__":
    main(sys.argv[1])
