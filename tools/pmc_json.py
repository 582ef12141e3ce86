#!/usr/bin/env python3
"""Condenses the counter passes of tools/profile.sh into one JSON file that bench.py reads for its
roofline object: per kernel the mean counter values per launch, plus, for the path kernel, the
utilisation of every unit that could bound it.

usage: tools/pmc_json.py <profile directory written by tools/profile.sh> <out.json>

What the derived figures mean (CUs = 256, SIMDs = 1024; counters with _sum are summed over the chip):
  cycles             GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8 XCDs): shader cycles the launch took
  ta_busy_frac       TA_TA_BUSY_sum / (256 * cycles): share of the launch during which a CU's texture addresser
                     (the unit every vector-memory instruction passes through) was busy.  Reads 0.97-1.00 in
                     tools/ubench/gather_rate, whatever the coherence of the lanes.
  valu_busy_low/high SQ_INSTS_VALU * 2 resp. SQ_ACTIVE_INST_VALU * 4 over (1024 * cycles): a plain f32 VALU
                     instruction occupies its SIMD for 2 cycles and a packed-f32 one for 4
                     (tools/ubench/valu_rate); the counters do not tell the two kinds apart.
  l2_read_bytes      TCP_TCC_READ_REQ_sum * 64: bytes the vector L1s requested from the L2s
  hbm_side_bytes     FETCH_SIZE * 1024 * read factor + WRITE_SIZE * 1024 (memory side of the L2s; hits in
                     the Infinity Cache are included, so an upper bound on HBM bytes)
"""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CUS, SIMDS = 256, 1024


def short(name):
    m = re.match(r"(?:void )?([A-Za-z_0-9]+)", name)
    return m.group(1) if m else name


def source_hash():
    h = hashlib.sha256()
    for f in ("pt_kernels.hip", "pt_internal.h", "pt_exact_div.h", "pt_portable_math.h"):
        h.update(open(os.path.join(ROOT, "path-tracing_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def main(prof, out):
    kernels = defaultdict(dict)
    for d in sorted(glob.glob(os.path.join(prof, "pmc_*"))):
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(list))
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
            for k, ctrs in acc.items():
                for c, v in ctrs.items():
                    kernels[k][c] = sum(v) / len(v)
    for f in glob.glob(os.path.join(prof, "trace", "**", "*kernel_trace.csv"), recursive=True):
        dur = defaultdict(list)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                dur[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        for k, v in dur.items():
            kernels[k]["trace_mean_ms"] = sum(v) / len(v) / 1e6
            kernels[k]["trace_launches"] = len(v)
    workload = None
    for log in glob.glob(os.path.join(prof, "trace.log")):
        for line in open(log):
            if line.startswith("{"):
                workload = json.loads(line)["config"]["workload"]
    read_factor = 1.0
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tj):
        read_factor = json.load(open(tj)).get("read_factor_applied", 1.0)
    res = {"workload": workload, "source_hash": source_hash(), "fetch_read_factor": read_factor, "kernels": kernels}
    pk = kernels.get("path_kernel")
    if pk and "GRBM_GUI_ACTIVE" in pk:
        cyc = pk["GRBM_GUI_ACTIVE"] / 8.0
        d = {"cycles": cyc, "trace_mean_ms": pk.get("trace_mean_ms"),
             "clock_ghz": cyc / (pk["trace_mean_ms"] * 1e6) if pk.get("trace_mean_ms") else None}
        if "TA_TA_BUSY_sum" in pk:
            d["ta_busy_frac"] = pk["TA_TA_BUSY_sum"] / (CUS * cyc)
            d["ta_busy_cycles_per_cu"] = pk["TA_TA_BUSY_sum"] / CUS
        if "TD_TD_BUSY_sum" in pk:
            d["td_busy_frac"] = pk["TD_TD_BUSY_sum"] / (CUS * cyc)
        if "SQ_INSTS_VALU" in pk:
            d["valu_busy_low"] = pk["SQ_INSTS_VALU"] * 2.0 / (SIMDS * cyc)
            d["valu_busy_high"] = pk["SQ_ACTIVE_INST_VALU"] * 4.0 / (SIMDS * cyc)
            d["valu_lane_utilisation"] = pk["SQ_THREAD_CYCLES_VALU"] / (pk["SQ_ACTIVE_INST_VALU"] * 64.0)
        if "SQ_INSTS_VMEM_RD" in pk:
            d["vmem_read_instructions"] = pk["SQ_INSTS_VMEM_RD"]
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in pk:
            d["l1_tag_lookups"] = pk["TCP_TOTAL_CACHE_ACCESSES_sum"]
            d["l1_hit_rate"] = 1.0 - pk["TCP_TCC_READ_REQ_sum"] / pk["TCP_TOTAL_CACHE_ACCESSES_sum"]
            d["l2_read_bytes"] = pk["TCP_TCC_READ_REQ_sum"] * 64.0
        if "TCC_HIT_sum" in pk:
            d["l2_hit_rate"] = pk["TCC_HIT_sum"] / (pk["TCC_HIT_sum"] + pk["TCC_MISS_sum"])
        if "FETCH_SIZE" in pk and "WRITE_SIZE" in pk:
            d["hbm_side_read_bytes"] = pk["FETCH_SIZE"] * 1024.0 * read_factor
            d["hbm_side_write_bytes"] = pk["WRITE_SIZE"] * 1024.0
            d["hbm_side_bytes"] = d["hbm_side_read_bytes"] + d["hbm_side_write_bytes"]
        ks = os.path.join(prof, "kstats.json")
        if os.path.exists(ks):            # lane diagnostics of an instrumented run of the same workload (tools/kernel_stats.py)
            d["lane_stats"] = json.load(open(ks))
        res["path_kernel"] = d
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    print(json.dumps(res.get("path_kernel"), indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
