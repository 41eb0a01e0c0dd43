#!/usr/bin/env python
"""Fold two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, one counter per run, CSV output) into the per-launch HBM
traffic file bench.py reads (profiles/r01_pmc_traffic.json).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_FETCH_SIZE -o pmc -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_WRITE_SIZE -o pmc -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE [--mfma=gpurun_out/pmc_MfmaUtil] \
        "<command>" > profiles/r01_pmc_traffic.json

Both counters are reported in KB.  FETCH_SIZE is doubled: gfx950 counts half of the wide (16 B per lane) streaming
reads these kernels issue (MI355X_MICROARCH.md, HBM / rocprofv3 section).
"""
import csv
import glob
import json
import os
import re
import sys

# family -> substring of the kernel name (all template instances of the conv kernel count as one family, like
# bench.py's per-frcnn_conv2d_fwd-launch timing)
# The RoIAlign launches with the store epilogue (", true>": the two projected maps of a frame, Network._layer4_projected) are a
# family of their own: "roi_align_fwd" stays the reference's operation on the 1024-channel map (bench.py roi_align_timing).
FAMILIES = {"conv_igemm": "conv_igemm", "roi_align_fwd_affine": "roi_align_fwd_planned<4, true>",
            "roi_align_fwd": "roi_align_fwd", "conv_wgrad_f32": "conv_wgrad_f32"}
# kernels whose bytes are ADDED to a family without counting as launches of it: one RoIAlign operation = the plan kernel
# + the pooling kernel, reported per operation
# (same for the other launches of one frcnn_conv2d_fwd call: the split-K second pass and the Winograd transforms)
COMPANIONS = {"conv_splitk_epilogue": "conv_igemm", "wino_": "conv_igemm"}


# bench.py tunes its conv plans during the first frames (extra candidate launches); its roofline is timed over the
# LAST 5 eager frames x 106 conv launches, so the conv traffic is averaged over exactly those dispatches.
LAST = {"conv_igemm": 5 * 105}      # overridden by --conv-calls=N (bench.py roofline.conv_calls_per_frame)


def per_kernel(directory, counter):
    paths = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not paths:
        raise SystemExit("no *counter_collection.csv under %s" % directory)
    vals, extra = {}, {}
    plans = []
    with open(paths[0], newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            if "roi_plan_kernel" in row["Kernel_Name"]:       # belongs to the pooling launch that follows it
                plans.append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
                continue
            for sub, fam in COMPANIONS.items():
                if sub in row["Kernel_Name"]:
                    extra.setdefault(fam, []).append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
            for fam, sub in FAMILIES.items():
                if sub in row["Kernel_Name"]:
                    vals.setdefault(fam, []).append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
                    break
    pools = sorted((d, fam) for fam in ("roi_align_fwd", "roi_align_fwd_affine") for d, _ in vals.get(fam, []))
    for d, v in plans:
        nxt = [fam for pd, fam in pools if pd > d]
        if nxt:
            extra.setdefault(nxt[0], []).append((d, v))
    acc = {}
    for fam, rows in vals.items():
        rows.sort()
        if fam in LAST:
            rows = rows[-LAST[fam]:]
        first = rows[0][0]
        # companions of the selected dispatches only (a companion may precede its family kernel by a few dispatches:
        # the Winograd input transform runs before the grouped GEMM of the same call)
        comp = sum(v for d, v in extra.get(fam, []) if d >= first - 2)
        acc[fam] = (len(rows), sum(v for _, v in rows) + comp)
    return acc


def mfma_util(directory):
    """Duration-weighted MfmaUtil (rocprofv3 derived counter: SQ_VALU_MFMA_BUSY_CYCLES summed over the SIMDs /
    (GRBM_GUI_ACTIVE x SIMD count)) of the conv launches of the roofline frames, plus the per-instantiation averages."""
    paths = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not paths:
        return None
    rows = []
    with open(paths[0], newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == "MfmaUtil" and "conv_igemm" in row["Kernel_Name"]:
                rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"], float(row["Counter_Value"]),
                             int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    rows.sort()
    rows = rows[-LAST["conv_igemm"]:]
    if not rows:
        return None
    total = sum(r[3] for r in rows)
    by = {}
    for _, name, val, dur in rows:
        m = re.search(r"conv_igemm\w*<[^>]*>", name)
        short = m.group(0) if m else name
        n, sv, sd = by.get(short, (0, 0.0, 0))
        by[short] = (n + 1, sv + val * dur, sd + dur)
    return {"launches": len(rows), "mfma_util_percent": sum(r[2] * r[3] for r in rows) / total,
            "by_kernel": {k: {"launches": n, "mfma_util_percent": sv / sd, "time_share": sd / total}
                          for k, (n, sv, sd) in sorted(by.items())}}


def plans_sha(path):
    """bench.plans_sha of a saved plan table (hash independent of row order)."""
    import hashlib
    with open(path) as f:
        rows = json.load(f)
    return hashlib.sha1(json.dumps(sorted([int(v) for v in r] for r in rows)).encode()).hexdigest()[:16]


def main(fetch_dir, write_dir, command="", mfma_dir=None, plans=None, head=None):
    fetch = per_kernel(fetch_dir, "FETCH_SIZE")
    write = per_kernel(write_dir, "WRITE_SIZE")
    out = {"command": command + " (one counter per pass)", "collected_at_head": head,
           "plans_sha": plans_sha(plans) if plans else None,
           "method": "per-dispatch FETCH_SIZE / WRITE_SIZE (KB) summed per kernel family and divided by its launch "
                     "count; FETCH_SIZE doubled (gfx950 reports half of wide 16 B/lane streaming reads, "
                     "MI355X_MICROARCH.md HBM section); Infinity-Cache hits are included in FETCH_SIZE",
           "kernels": {}}
    for fam in FAMILIES:
        if fam not in fetch or fam not in write:
            continue
        n, kb = fetch[fam]
        nw, kbw = write[fam]
        fb = 2.0 * kb * 1024.0 / n
        wb = kbw * 1024.0 / nw
        out["kernels"][fam] = {"launches": n, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                               "traffic_bytes_per_launch": fb + wb}
    if mfma_dir:
        util = mfma_util(mfma_dir)
        if util:
            out["kernels"].setdefault("conv_igemm", {})["mfma"] = util
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    for a in sys.argv[1:]:
        if a.startswith("--conv-calls="):
            LAST["conv_igemm"] = 5 * int(a.split("=", 1)[1])
    sys.argv = [a for a in sys.argv if not a.startswith("--conv-calls=")]
    opts = {a.split("=", 1)[0][2:]: a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--") and "=" in a}
    args = [a for a in sys.argv[1:] if not (a.startswith("--") and "=" in a)]
    main(args[0], args[1], " ".join(args[2:]), opts.get("mfma"), opts.get("plans"), opts.get("head"))
