#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 PMC passes (CSV output): python tools/pmc_kernel.py DIR [substring]
prints, for every kernel whose name contains `substring`, the mean of each collected counter over its dispatches."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(directory, sub=""):
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if sub in row["Kernel_Name"]:
                    name = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
                    acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name, counters in sorted(acc.items()):
        print(name)
        for c, vals in sorted(counters.items()):
            print("    %-28s n=%-5d mean %.4g  min %.4g  max %.4g" % (c, len(vals), sum(vals) / len(vals), min(vals), max(vals)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
