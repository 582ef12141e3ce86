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


def main(out, all_dirs=False):
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
    for d in sorted(glob.glob(os.path.join(out, "*" if all_dirs else "pmc_*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(list))
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
            print("-- %s --" % os.path.basename(d))
            if all_dirs and any(k == "walk" for k in acc):
                # the gather microbenchmark launches one kernel several times with different arguments:
                # list its dispatches one by one, in launch order
                rows = defaultdict(dict)
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        rows[int(row["Dispatch_Id"])][row["Counter_Name"]] = float(row["Counter_Value"])
                for disp in sorted(rows):
                    print("walk dispatch %-3d %s" % (disp, "  ".join("%s=%.4g" % kv for kv in sorted(rows[disp].items()))))
                continue
            for k in acc:
                vals = "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items()))
                n = len(next(iter(acc[k].values())))
                print("%-28s n=%-4d %s" % (k, n, vals))


if __name__ == "__main__":
    main(sys.argv[1], len(sys.argv) > 2 and sys.argv[2] == "pmc_all")
