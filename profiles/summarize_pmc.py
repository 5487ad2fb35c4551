#!/usr/bin/env python
"""profiles/summarize_pmc.py <tag>: condense gpurun_out/prof_<tag>/ (written by collect_pmc.sh) into
  profiles/<tag>_kernel_stats.csv, <tag>_bench_under_rocprof.json     kernel trace of the default bench
  profiles/<tag>_pmc_traffic.json, <tag>_pmc_summary.json             W=51 scoring kernel per precision
  profiles/<tag>_train_b<batch>_serial_kernel_stats.csv, <tag>_train_b<batch>_bench_under_rocprof.json
  profiles/<tag>_train_pmc_traffic.json, <tag>_train_pmc_summary.json  W=51 backward pass of the training step
  profiles/<tag>_probe_insts.json                                      what SQ_INSTS_VALU counts (coexec_probe)
  profiles/<tag>_pmc_assemble.json                                     HBM-side traffic of the window-assembly kernel
  profiles/<tag>_issue_cost.txt, <tag>_pmc_classes.json                issue cost per instruction class (issue_cost_probe)
                                                                       and the W=51 scoring kernel's instructions per class

HBM-side traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes): on gfx950 FETCH_SIZE tallies
the 128-byte requests of 16-byte-per-lane coalesced reads at 64 bytes, WRITE_SIZE is exact.

Issue budget of a kernel: a SIMD's issue port is held 4 clocks by a vector instruction and for its passes by
a matrix instruction (SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA), and the two do not overlap
(tools/coexec_probe.hip); SQ_INSTS_VALU counts both kinds (the probe's matrix-only kernel: see
<tag>_probe_insts.json), so vector instructions = SQ_INSTS_VALU - SQ_INSTS_MFMA."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")


def per_kernel(pattern, want):
    """{kernel family: {counter: [value per dispatch, in dispatch order]}} from the newest counter_collection csv of a
    pass; "_dur_ns" holds the dispatches' durations in the same order"""
    rows = collections.defaultdict(lambda: collections.defaultdict(dict))
    files = sorted(glob.glob(os.path.join(src, pattern, "*", "*counter_collection.csv")), key=os.path.getmtime)
    for f in files[-1:]:                 # the most recent pass only (gpurun merges new files next to old ones)
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            if want(name):
                d = rows[name][int(r["Dispatch_Id"])]
                d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                d["_dur_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for name, by_id in rows.items():
        for i in sorted(by_id):
            for c, v in by_id[i].items():
                agg[name][c].append(v)
    return agg


def mean(v):
    return sum(v) / len(v) if v else None


def pick(agg, *needles):
    for k, v in agg.items():
        if all(n in k for n in needles):
            return k, v
    return None, None


def bench_line(name):
    try:
        return json.loads(open(os.path.join(src, name)).read().strip().splitlines()[-1])
    except Exception:
        return None


def vector_includes_matrix():
    """from the probe pass: the first kernel of tools/coexec_probe issues matrix instructions only"""
    agg = per_kernel("probe", lambda n: "probe" in n)
    out = {}
    for k, v in agg.items():
        if v.get("SQ_INSTS_VALU") and v.get("SQ_INSTS_MFMA"):
            out[k] = {"SQ_INSTS_VALU": max(v["SQ_INSTS_VALU"]), "SQ_INSTS_MFMA": max(v["SQ_INSTS_MFMA"]),
                      "SQ_ACTIVE_INST_VALU": max(v.get("SQ_ACTIVE_INST_VALU", [0])),
                      "SQ_VALU_MFMA_BUSY_CYCLES": max(v.get("SQ_VALU_MFMA_BUSY_CYCLES", [0]))}
    return out


def issue_summary(k, m, units_per_launch):
    dur = mean(m["_dur_ns"])
    clk = mean(m["GRBM_GUI_ACTIVE"]) / 8 / dur          # GHz: the counter sums the 8 XCDs
    busy = mean(m["SQ_VALU_MFMA_BUSY_CYCLES"])
    n_mfma = mean(m["SQ_INSTS_MFMA"])
    rec = {"kernel": k, "launches": len(m["_dur_ns"]), "avg_launch_us_under_pmc": dur / 1e3, "clock_ghz": clk,
           "SQ_INSTS_MFMA_per_launch": n_mfma, "SQ_VALU_MFMA_BUSY_CYCLES_per_launch": busy,
           "mfma_busy_frac_of_simd_cycles": busy / (1024 * dur * clk)}
    if m.get("SQ_INSTS_VALU"):
        n_valu = mean(m["SQ_INSTS_VALU"]) - n_mfma
        issue = 4.0 * n_valu + busy
        rec.update({"SQ_INSTS_VALU_per_launch": mean(m["SQ_INSTS_VALU"]),
                    "SQ_ACTIVE_INST_VALU_per_launch": mean(m.get("SQ_ACTIVE_INST_VALU", [])),
                    "SQ_INSTS_LDS_per_launch": mean(m.get("SQ_INSTS_LDS", [])),
                    "SQ_WAIT_INST_ANY_per_launch": mean(m.get("SQ_WAIT_INST_ANY", [])),
                    "mfma_cycles_per_inst": busy / n_mfma if n_mfma else None,
                    "issue_cycles_per_launch": issue,
                    "issue_busy_frac_of_simd_cycles": issue / (1024 * dur * clk)})
        if units_per_launch:
            rec.update({"units_per_launch": units_per_launch, "valu_insts_per_site": n_valu / units_per_launch,
                        "mfma_insts_per_site": n_mfma / units_per_launch, "issue_cycles_per_site": issue / units_per_launch})
    return rec


# ---------------------------------------------------------------- scoring kernels, per precision
is_score = lambda n: "branch_kernel" in n or "dense_kernel" in n
traffic, summary = {}, {}
for prec in ("f32", "f16x3", "f16", "bf16"):
    kf, f = pick(per_kernel("fetch_" + prec, is_score), "branch_kernel", "51")
    kw, w = pick(per_kernel("write_" + prec, is_score), "branch_kernel", "51")
    km, m = pick(per_kernel("mfma_" + prec, is_score), "branch_kernel", "51")
    if not f or not w:
        continue
    line = bench_line("fetch_%s.json" % prec)
    spl = line["roofline"]["sites_per_launch"] if line else None
    fetch_kib, write_kib = mean(f["FETCH_SIZE"]), mean(w["WRITE_SIZE"])
    traffic[prec] = {"kernel": kf, "launches": len(f["FETCH_SIZE"]), "sites_per_launch": spl,
                     "FETCH_SIZE_KiB_per_launch": fetch_kib, "WRITE_SIZE_KiB_per_launch": write_kib,
                     "traffic_bytes_per_launch": (2 * fetch_kib + write_kib) * 1024}
    if m:
        summary[prec] = issue_summary(km, m, spl)
if traffic:
    json.dump(traffic, open(os.path.join(dst, tag + "_pmc_traffic.json"), "w"), indent=1)
    json.dump(summary, open(os.path.join(dst, tag + "_pmc_summary.json"), "w"), indent=1)
for f in sorted(glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")), key=os.path.getmtime)[-1:]:
    shutil.copy(f, os.path.join(dst, tag + "_kernel_stats.csv"))
tb = os.path.join(src, "trace_bench.json")
if os.path.exists(tb) and bench_line("trace_bench.json"):
    shutil.copy(tb, os.path.join(dst, tag + "_bench_under_rocprof.json"))

# ---------------------------------------------------------------- training step
# The backward pass of the W=51 branch's TransformerBlocks + pooling is ONE launch at small batches
# (tb_bwd_pair_kernel<51> up to 112 sites, tb_bwd_kernel<51, 256, 0> above) and FIVE at large ones: tb_bwd_kernel<51, 512, 1> (pooling), then per block
# tb_row_bwd_kernel (row-wise half; one kernel for all window sizes -- the W=51 launches are the longest third of its
# launches) and tb_bwd_kernel<51, 512, 2> (attention half).  Counters and durations are summed over those launches:
# "per launch" below means per backward pass of the branch.
is_train = lambda n: "tb_" in n or "conv_" in n


def bwd51(agg):
    """(label, {counter: [per-pass value]}, launches per pass) of the W=51 backward pass"""
    parts = []
    for k, v in agg.items():
        if "tb_bwd_kernel<51" in k or "tb_bwd_pair_kernel<51" in k:
            parts.append((k, v, None))
        elif "tb_row_bwd_kernel" in k:
            parts.append((k, v, "top_third"))
    if not parts:
        return None, None, 0
    passes = min(len(v["_dur_ns"]) for k, v, sel in parts if sel is None and (k.endswith(", 1>") or k.endswith(", 0>") or "pair" in k))
    out = collections.defaultdict(float)
    launches = 0
    for k, v, sel in parts:
        n = len(v["_dur_ns"])
        keep = range(n)
        if sel == "top_third":
            keep = sorted(range(n), key=lambda i: -v["_dur_ns"][i])[:n // 3]
        launches += len(keep)
        for c, vals in v.items():
            out[c] += sum(vals[i] for i in keep) / passes
    label = " + ".join(sorted(k for k, _, _ in parts))
    return label, {c: [x] for c, x in out.items()}, launches // passes


ttraffic, tsummary = {}, {}
for batch in (8192, 64):
    b = "b%d" % batch
    for f in sorted(glob.glob(os.path.join(src, "train_trace_" + b, "*", "*kernel_stats.csv")), key=os.path.getmtime)[-1:]:
        shutil.copy(f, os.path.join(dst, "%s_train_%s_serial_kernel_stats.csv" % (tag, b)))
    if bench_line("train_trace_%s.json" % b):
        shutil.copy(os.path.join(src, "train_trace_%s.json" % b), os.path.join(dst, "%s_train_%s_bench_under_rocprof.json" % (tag, b)))
    kf, f, nl = bwd51(per_kernel("train_fetch_" + b, is_train))
    kw, w, _ = bwd51(per_kernel("train_write_" + b, is_train))
    km, m, _ = bwd51(per_kernel("train_mfma_" + b, is_train))
    if f and w:
        fetch_kib, write_kib = mean(f["FETCH_SIZE"]), mean(w["WRITE_SIZE"])
        ttraffic["batch%d" % batch] = {"kernel": kf, "launches_per_pass": nl, "sites_per_launch": batch,
                                       "FETCH_SIZE_KiB_per_launch": fetch_kib, "WRITE_SIZE_KiB_per_launch": write_kib,
                                       "traffic_bytes_per_launch": (2 * fetch_kib + write_kib) * 1024}
    if m:
        tsummary["batch%d" % batch] = issue_summary(km, m, batch)
        tsummary["batch%d" % batch]["launches_per_pass"] = nl
if ttraffic:
    json.dump(ttraffic, open(os.path.join(dst, tag + "_train_pmc_traffic.json"), "w"), indent=1)
if tsummary:
    json.dump(tsummary, open(os.path.join(dst, tag + "_train_pmc_summary.json"), "w"), indent=1)

# ---------------------------------------------------------------- vector instructions per class, priced
# Cost of one wavefront instruction in units of v_fma_f32 (tools/issue_cost_probe: 32 independent instructions per trip,
# two wavefronts per SIMD).  add / mul / fma issue at the base rate, transcendentals at ~3.2x, and everything the class
# counters do not single out (max, compare, select, conversion, integer, packed, cross-lane) is priced at the cost of
# v_max3_f32 (~1.6x; v_mov, which issues at the base rate, is over-priced by that).
cost_file = os.path.join(src, "issue_cost.txt")
classes = {}
if os.path.exists(cost_file):
    shutil.copy(cost_file, os.path.join(dst, tag + "_issue_cost.txt"))
    cost, base_clk = {}, None
    for ln in open(cost_file):
        if "clk per wave-instruction" in ln:
            base_clk = float(ln.split("=")[1].split("clk")[0])
        elif " x v_fma_f32" in ln:
            name = ln.split()[0] if not ln.startswith("v_cndmask_b32_e64") else " ".join(ln.split()[:2])
            cost[name] = float(ln.split("ms")[1].split("x")[0])
    mf = {"f32": cost.get("v_mfma_32x32x2_f32"), "f16x3": cost.get("v_mfma_32x32x16_bf16"), "f16": cost.get("v_mfma_32x32x16_bf16"),
          "bf16": cost.get("v_mfma_32x32x16_bf16")}
    for prec in ("f32", "f16x3", "f16", "bf16"):
        km, m = pick(per_kernel("cls_" + prec, is_score), "branch_kernel", "51")
        line = bench_line("cls_%s.json" % prec)
        if not m or not line or base_clk is None or mf[prec] is None:
            continue
        spl = line["roofline"]["sites_per_launch"]
        g = lambda c: mean(m.get(c, [0.0])) / spl
        n_mfma = g("SQ_INSTS_MFMA")
        n_vec = g("SQ_INSTS_VALU") - n_mfma
        n_add, n_mul, n_fma, n_tr = g("SQ_INSTS_VALU_ADD_F32"), g("SQ_INSTS_VALU_MUL_F32"), g("SQ_INSTS_VALU_FMA_F32"), g("SQ_INSTS_VALU_TRANS_F32")
        n_other = n_vec - n_add - n_mul - n_fma - n_tr
        units = (n_add * cost["v_add_f32"] + n_mul * cost["v_mul_f32"] + n_fma * 1.0 + n_tr * cost["v_exp_f32"] +
                 n_other * cost["v_max3_f32"] + n_mfma * mf[prec])
        classes[prec] = {"kernel": km, "sites_per_launch": spl, "per_site": {
            "vector": n_vec, "add_f32": n_add, "mul_f32": n_mul, "fma_f32": n_fma, "trans_f32": n_tr, "cvt": g("SQ_INSTS_VALU_CVT"),
            "int32": g("SQ_INSTS_VALU_INT32"), "other": n_other, "mfma": n_mfma},
            "cost_x_fma": {"add_f32": cost["v_add_f32"], "mul_f32": cost["v_mul_f32"], "fma_f32": 1.0, "trans_f32": cost["v_exp_f32"],
                           "other": cost["v_max3_f32"], "mfma": mf[prec]},
            "fma_clk_at_2p4ghz": base_clk, "issue_units_per_site": units, "issue_clk_per_site_at_2p4ghz": units * base_clk,
            "mfma_share_of_issue": n_mfma * mf[prec] / units, "trans_share_of_issue": n_tr * cost["v_exp_f32"] / units,
            "issue_bound_sites_per_s": 1024 * 2.4e9 / (units * base_clk)}
    if classes:
        json.dump(classes, open(os.path.join(dst, tag + "_pmc_classes.json"), "w"), indent=1)

# ---------------------------------------------------------------- window assembly (bench.py --proteome)
# the launches over ALL sites (bench.py times five of them) are the ones with the largest grid
asm = {}
fa = per_kernel("asm_fetch", lambda n: "assemble_kernel" in n)
wa = per_kernel("asm_write", lambda n: "assemble_kernel" in n)
line = bench_line("asm_fetch.json")
if fa and wa and line:
    f = list(fa.values())[0]
    w = list(wa.values())[0]
    big_f = [v for v in f["FETCH_SIZE"] if v >= 0.5 * max(f["FETCH_SIZE"])]
    big_w = [v for v in w["WRITE_SIZE"] if v >= 0.5 * max(w["WRITE_SIZE"])]
    sites = line["proteome"]["assemble_one_launch"]["sites"]
    asm = {"kernel": "assemble_kernel", "sites_per_launch": sites, "launches": len(big_f),
           "FETCH_SIZE_KiB_per_launch": mean(big_f), "WRITE_SIZE_KiB_per_launch": mean(big_w),
           "traffic_bytes_per_launch": (2 * mean(big_f) + mean(big_w)) * 1024,
           "algorithmic_bytes_per_launch": sites * 117}
    json.dump(asm, open(os.path.join(dst, tag + "_pmc_assemble.json"), "w"), indent=1)

probe = vector_includes_matrix()
if probe:
    json.dump(probe, open(os.path.join(dst, tag + "_probe_insts.json"), "w"), indent=1)
print(json.dumps({"traffic": traffic, "summary": summary, "train_traffic": ttraffic, "train_summary": tsummary, "probe": probe, "classes": classes, "assemble": asm}, indent=1))
