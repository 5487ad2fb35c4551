#!/bin/bash
# usage (on the MI355X box): bash tools/small_call_trace.sh [sites]   kernel timeline of the LAST device-resident small call
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/small_call_trace
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/small_call_trace -- python3 /root/repo/tools/small_call_timeline.py ${1:-53} > /root/repo/gpurun_out/small_call_trace.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/root/repo/gpurun_out/small_call_trace/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
heads = [i for i, r in enumerate(rows) if "dense_small_kernel<2>" in r["Kernel_Name"] or "head" in r["Kernel_Name"].lower()]
# one call = everything from the kernel after the previous call's last kernel (the head layer) to this call's head layer
ends = [i for i, r in enumerate(rows) if "Li2" in r["Kernel_Name"] or "<2>" in r["Kernel_Name"] or "dense_kernel<1, 2>" in r["Kernel_Name"]]
if len(ends) < 3:
    ends = list(range(len(rows)))
lo, hi = ends[-3] + 1, ends[-2] + 1
call = rows[lo:hi]
t0 = int(call[0]["Start_Timestamp"])
busy = t0
print("kernels in call:", len(call), " span %.1f us" % ((int(call[-1]["End_Timestamp"]) - t0) / 1e3))
for r in call:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  +%6.1f us  gap %6.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - busy) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:80]))
    busy = max(busy, e)
PY
