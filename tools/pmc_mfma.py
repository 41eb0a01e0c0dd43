#!/usr/bin/env python
"""Duration-weighted MfmaUtil per kernel from ONE rocprofv3 PMC pass (CSV output) — used for the training step
(BASELINE.json configs[3] names a "rocprof MFMA capture"):

    rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d gpurun_out/train_pmc -o pmc -- \
        python3 tools/bench_configs.py --train --steps 3
    python tools/pmc_mfma.py gpurun_out/train_pmc "<command>" [--last-frac=0.5] > profiles/r02_train_pmc.json

MfmaUtil = rocprofv3's derived counter: matrix-pipe busy cycles summed over the SIMDs / (GRBM_GUI_ACTIVE x SIMD count).
``--last-frac`` keeps only the last fraction of the dispatches (skips plan autotuning and warm-up launches)."""
import csv
import glob
import json
import os
import re
import sys


def main(directory, command, last_frac):
    paths = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not paths:
        raise SystemExit("no *counter_collection.csv under %s" % directory)
    rows = []
    with open(paths[0], newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == "MfmaUtil":
                rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"], float(row["Counter_Value"]),
                             int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    rows.sort()
    rows = rows[int(len(rows) * (1.0 - last_frac)):]
    fam = {"forward / data-gradient convolution (conv_igemm*)": "conv_igemm", "filter gradient (conv_wgrad_f32)": "conv_wgrad"}
    total_all = sum(r[3] for r in rows)
    out = {"command": command, "counter": "MfmaUtil (percent of SIMD cycles with the matrix pipe busy), duration-weighted",
           "dispatches": len(rows), "kernel_time_ms": total_all / 1e6, "families": {}, "by_kernel": {}}
    for label, sub in fam.items():
        sel = [r for r in rows if sub in r[1]]
        if not sel:
            continue
        t = sum(r[3] for r in sel)
        out["families"][label] = {"launches": len(sel), "mfma_util_percent": sum(r[2] * r[3] for r in sel) / t,
                                  "share_of_kernel_time": t / total_all}
    mm = [r for r in rows if "conv_igemm" in r[1] or "conv_wgrad" in r[1]]
    if mm:
        t = sum(r[3] for r in mm)
        out["matrix_kernels_overall"] = {"launches": len(mm), "mfma_util_percent": sum(r[2] * r[3] for r in mm) / t,
                                         "share_of_kernel_time": t / total_all}
    by = {}
    for _, name, val, dur in mm:
        m = re.search(r"conv_\w+<[^>]*>", name)
        short = m.group(0) if m else name.split("(")[0]
        n, sv, sd = by.get(short, (0, 0.0, 0))
        by[short] = (n + 1, sv + val * dur, sd + dur)
    out["by_kernel"] = {k: {"launches": n, "mfma_util_percent": sv / sd, "share_of_kernel_time": sd / total_all}
                        for k, (n, sv, sd) in sorted(by.items(), key=lambda kv: -kv[1][2])}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    frac = [float(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--last-frac=")]
    main(args[0], " ".join(args[1:]), frac[0] if frac else 1.0)
