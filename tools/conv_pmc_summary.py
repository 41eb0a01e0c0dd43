#!/usr/bin/env python
"""Fold the rocprofv3 counter CSVs of tools/conv_pmc_probe.py runs (one directory per --pmc pass) into one table: per case the
conv_igemm* dispatches of its last launch, every counter summed over them.

    python tools/conv_pmc_summary.py probe.json dir1 dir2 ... > table.md
"""
import csv
import glob
import json
import os
import sys


def load(directory):
    paths = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    rows = {}
    for path in paths:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                d = int(row["Dispatch_Id"])
                ent = rows.setdefault(d, {"kernel": row["Kernel_Name"], "grid": row.get("Grid_Size", ""), "wg": row.get("Workgroup_Size", ""),
                                          "vgpr": row.get("VGPR_Count", ""), "lds": row.get("LDS_Block_Size", ""), "c": {}})
                ent["c"][row["Counter_Name"]] = float(row["Counter_Value"])
    return rows


def main():
    meta = json.load(open(sys.argv[1]))
    cases, reps = meta["cases"], meta["reps"]
    merged = None
    for d in sys.argv[2:]:
        rows = load(d)
        ids = sorted(rows)
        if merged is None:
            merged = {i: rows[i] for i in ids}
        else:
            for i in ids:
                if i in merged and merged[i]["kernel"] == rows[i]["kernel"]:
                    merged[i]["c"].update(rows[i]["c"])
    # split the dispatch stream into launches of conv2d_nhwc: a launch = [wino_input?] conv_igemm* [wino_output?]; take conv_igemm only
    convs = [(i, merged[i]) for i in sorted(merged) if "conv_igemm" in merged[i]["kernel"]]
    assert len(convs) == len(cases) * reps, (len(convs), len(cases), reps)
    names = sorted({k for _, e in convs for k in e["c"]})
    print("| case | kernel | grid | " + " | ".join(names) + " |")
    print("|---|---|---|" + "---|" * len(names))
    out = []
    for ci, (name, flops) in enumerate(cases):
        i, e = convs[ci * reps + reps - 1]
        kern = e["kernel"].split("(")[0]
        print("| %s | %s | %s/%s | " % (name, kern, e["grid"], e["wg"]) + " | ".join("%.4g" % e["c"].get(n, float("nan")) for n in names) + " |")
        out.append({"case": name, "flops": flops, "kernel": kern, "grid": e["grid"], "counters": e["c"]})
    print()
    print("```json")
    print(json.dumps(out))
    print("```")


if __name__ == "__main__":
    main()
