#!/usr/bin/env python
"""profiles/summarize_pmc.py <tag>: condense gpurun_out/prof_<tag>/ (written by collect_pmc.sh) into
profiles/<tag>_kernel_stats.csv, profiles/<tag>_bench_under_rocprof.json, profiles/<tag>_pmc_traffic.json
and profiles/<tag>_pmc_summary.json.

HBM-side traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes): on gfx950 FETCH_SIZE tallies
the 128-byte requests of 16-byte-per-lane coalesced reads at 64 bytes, WRITE_SIZE is exact."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")


def per_kernel(pattern):
    """{kernel family: {counter: [values per dispatch]}} from a counter_collection csv"""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    files = sorted(glob.glob(os.path.join(src, pattern, "*", "*counter_collection.csv")), key=os.path.getmtime)
    for f in files[-1:]:                 # the most recent pass only (gpurun merges new files next to old ones)
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            if "branch_kernel" in name or "dense_kernel" in name:
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
                agg[name]["_dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return agg


def mean(v):
    return sum(v) / len(v) if v else None


def b51(agg):
    for k, v in agg.items():
        if "branch_kernel" in k and "51" in k:
            return k, v
    return None, None


traffic, summary = {}, {}
for prec in ("f32", "f16x3", "bf16"):
    kf, f = b51(per_kernel("fetch_" + prec))
    kw, w = b51(per_kernel("write_" + prec))
    km, m = b51(per_kernel("mfma_" + prec))
    if not f or not w:
        continue
    try:
        line = json.loads(open(os.path.join(src, "fetch_%s.json" % prec)).read().strip().splitlines()[-1])
        spl = line["roofline"]["sites_per_launch"]
    except Exception:
        spl = None
    fetch_kib, write_kib = mean(f["FETCH_SIZE"]), mean(w["WRITE_SIZE"])
    traffic[prec] = {"kernel": kf, "launches": len(f["FETCH_SIZE"]), "sites_per_launch": spl,
                     "FETCH_SIZE_KiB_per_launch": fetch_kib, "WRITE_SIZE_KiB_per_launch": write_kib,
                     "traffic_bytes_per_launch": (2 * fetch_kib + write_kib) * 1024}
    if m:
        dur = mean(m["_dur_ns"])
        clk = mean(m["GRBM_GUI_ACTIVE"]) / 8 / dur          # GHz: the counter sums the 8 XCDs
        busy = mean(m["SQ_VALU_MFMA_BUSY_CYCLES"])
        summary[prec] = {"kernel": km, "avg_launch_us_under_pmc": dur / 1e3, "clock_ghz": clk,
                         "SQ_INSTS_MFMA_per_launch": mean(m["SQ_INSTS_MFMA"]),
                         "SQ_VALU_MFMA_BUSY_CYCLES_per_launch": busy,
                         "mfma_busy_frac_of_simd_cycles": busy / (1024 * dur * clk)}
json.dump(traffic, open(os.path.join(dst, tag + "_pmc_traffic.json"), "w"), indent=1)
json.dump(summary, open(os.path.join(dst, tag + "_pmc_summary.json"), "w"), indent=1)
for f in sorted(glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")), key=os.path.getmtime)[-1:]:
    shutil.copy(f, os.path.join(dst, tag + "_kernel_stats.csv"))
tb = os.path.join(src, "trace_bench.json")
if os.path.exists(tb):
    shutil.copy(tb, os.path.join(dst, tag + "_bench_under_rocprof.json"))
print(json.dumps(traffic, indent=1))
print(json.dumps(summary, indent=1))
