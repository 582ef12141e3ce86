#!/usr/bin/env python3
"""Condenses the CSV files written by tools/profile.sh into one readable summary:
per-kernel launch count / mean duration from the kernel trace, and per-kernel mean PMC values."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.match(r"(?:void )?([A-Za-z_0-9]+)", name)
    return m.group(1) if m else name


def main(out):
    print("== kernel trace ==")
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
        dur = defaultdict(list)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                dur[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        tot = sum(sum(v) for v in dur.values())
        for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            print("%-28s calls %5d  mean %10.1f us  min %10.1f us  max %10.1f us  total %8.2f ms  %5.1f %%" % (
                k, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3, sum(v) / 1e6, 100.0 * sum(v) / tot))
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        print("-- rocprofv3 kernel_stats.csv --")
        print(open(f).read())
    print("== PMC (mean per dispatch, per kernel) ==")
    for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(list))
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
            print("-- %s --" % os.path.basename(d))
            for k in acc:
                vals = "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items()))
                n = len(next(iter(acc[k].values())))
                print("%-28s n=%-4d %s" % (k, n, vals))


if __name__ == "__main__":
    main(sys.argv[1])
