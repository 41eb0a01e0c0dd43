#!/usr/bin/env python
"""Effective shader clock per kernel from one rocprofv3 pass with --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv:
GRBM_GUI_ACTIVE counts cycles the GPU was busy during the dispatch; divided by the dispatch's duration it is the clock the kernel ran at.
    python tools/pmc_clock.py DIR [substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(directory, sub=""):
    dur = {}
    for path in glob.glob(os.path.join(directory, "**", "*kernel_trace.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                dur[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != "GRBM_GUI_ACTIVE" or sub not in row["Kernel_Name"]:
                    continue
                d = dur.get(row["Dispatch_Id"])
                if d:
                    name = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
                    acc[name].append((float(row["Counter_Value"]), d))
    for name, v in sorted(acc.items()):
        cyc = sum(c for c, _ in v)
        us = sum(d for _, d in v)
        print("%-44s n=%-5d avg %.1f us   GRBM_GUI_ACTIVE/us = %.0f (per-XCD sum; /8 = %.0f MHz)" % (name, len(v), us / len(v), cyc / us, cyc / us / 8))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
