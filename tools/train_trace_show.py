"""Top kernels and per-shape GEMM durations of gpurun_out/train_trace (written by tools/train_trace.sh)."""
import collections, csv, glob
f = glob.glob("gpurun_out/train_trace/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print("%-60s %6s %9.1f us %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
f = glob.glob("gpurun_out/train_trace/*/*kernel_trace.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "tgemm" in n:
        agg[(n[:34], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:16]:
    print(k, len(v), "avg %.1f us" % (sum(v) / len(v)), "total %.1f ms" % (sum(v) / 1e3))
