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


def conv_cross_check(path, frames=5, launches_per_frame=105, flops_per_frame=563.85e9):
    """bench.py measures its `roofline` over its LAST `frames` eager frames (after the timed region): sum the
    conv_igemm* dispatches of exactly those frames (plus the split-K second passes that follow them) from the
    per-dispatch table, for comparison with bench.py's kernel_ms_per_frame / avg_launch_us."""
    db = sqlite3.connect(path)
    rows = list(db.execute("select name, start, end from kernels where name like '%conv_igemm%' or name like "
                           "'%conv_splitk_epilogue%' or name like '%wino_%' order by start"))
    main = [r for r in rows if "conv_igemm" in r[0]]
    if len(main) < frames * launches_per_frame:
        return
    first = main[-frames * launches_per_frame][1]
    sel = [r for r in rows if r[1] >= first]
    igemm = sum(e - s for n, s, e in sel if "conv_igemm" in n) / 1e3
    epi = sum(e - s for n, s, e in sel if "splitk" in n) / 1e3
    wino = sum(e - s for n, s, e in sel if "wino_" in n) / 1e3
    print("\n## cross-check with bench.py's roofline (last %d eager frames, %d conv launches each)\n" % (frames, launches_per_frame))
    print("| quantity | value |\n|---|---:|")
    print("| conv_igemm* kernel time per frame | %.1f us |" % (igemm / frames))
    print("| + conv_splitk_epilogue per frame | %.1f us |" % (epi / frames))
    print("| + Winograd transform kernels per frame | %.1f us |" % (wino / frames))
    print("| average per frcnn_conv2d_fwd call (all its launches) | %.2f us |"
          % ((igemm + epi + wino) / (frames * launches_per_frame)))
    print("| => conv rate at %.1f GFLOP/frame (the convolutions as launched, bench.py roofline.flops_per_frame) | %.1f TFLOP/s |"
          % (flops_per_frame / 1e9, flops_per_frame / ((igemm + epi + wino) / frames * 1e-6) / 1e12))
    print("| => conv rate at 628.4 algorithmic GFLOP/frame (the reference's order of operations, SURVEY.md section 8(d)) | %.1f TFLOP/s |"
          % (628.4e9 / ((igemm + epi + wino) / frames * 1e-6) / 1e12))


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    main(args[0], " ".join(args[1:]))
    if "--conv-cross-check" in sys.argv:
        flops = [float(a.split("=", 1)[1]) for a in sys.argv[1:] if a.startswith("--flops-per-frame=")]
        calls = [int(a.split("=", 1)[1]) for a in sys.argv[1:] if a.startswith("--conv-calls=")]
        kw = {"flops_per_frame": flops[0]} if flops else {}
        if calls:
            kw["launches_per_frame"] = calls[0]
        conv_cross_check(args[0], **kw)
