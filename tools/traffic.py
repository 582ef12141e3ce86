#!/usr/bin/env python3
"""Turns the passes of tools/traffic.sh into traffic.json: memory-side bytes per path-kernel launch.

FETCH_SIZE and WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE under-reports wide reads, by a factor
that depends on the access width, so the read side is multiplied by the factor measured here on a
known byte count in the same access pattern (whole 64-byte records at unrelated addresses of a
4 GiB table: every record read has to come from memory)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def per_kernel(d):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = re.match(r"(?:void )?([A-Za-z_0-9]+)", row["Kernel_Name"]).group(1)
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def main(out):
    calib = {}
    for mode, label in ((0, "stream16"), (1, "record64"), (2, "record48"), (3, "record16")):
        log = open(os.path.join(out, "calib%d.log" % mode)).read()
        info = json.loads(re.search(r"^\{.*\}$", log, re.M).group(0))
        k = per_kernel(os.path.join(out, "calib%d" % mode))
        name = [n for n in k if n.endswith("_kernel")][0]
        kib = sum(k[name]["FETCH_SIZE"]) / len(k[name]["FETCH_SIZE"])
        calib[label] = {"algorithmic_bytes": info["algorithmic_bytes"], "fetch_size_bytes": kib * 1024,
                        "true_over_reported": info["algorithmic_bytes"] / (kib * 1024), "GBps": info["GBps"]}
    log = open(os.path.join(out, "pmc_FETCH_SIZE.log")).read()
    bench = json.loads([l for l in log.splitlines() if l.startswith("{")][-1])
    rd = per_kernel(os.path.join(out, "pmc_FETCH_SIZE"))
    wr = per_kernel(os.path.join(out, "pmc_WRITE_SIZE"))
    res = {"workload": bench["config"]["workload"], "calibration": calib, "kernels": {}}
    factor = calib["record64"]["true_over_reported"]
    for name in rd:
        f = rd[name]["FETCH_SIZE"]
        w = wr.get(name, {}).get("WRITE_SIZE", [0.0])
        res["kernels"][name] = {
            "launches": len(f), "fetch_size_bytes_reported": sum(f) / len(f) * 1024,
            "write_size_bytes": sum(w) / len(w) * 1024,
        }
    pk = res["kernels"]["path_kernel"]
    res["read_factor_applied"] = factor
    res["path_kernel_hbm_bytes_per_launch"] = pk["fetch_size_bytes_reported"] * factor + pk["write_size_bytes"]
    res["note"] = ("L2 memory-side request bytes (Infinity Cache hits are counted, not excluded); reads scaled by the "
                   "record64 calibration factor")
    with open(os.path.join(out, "traffic.json"), "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
