"""Timeline of ONE training step from gpurun_out/train_trace (tools/train_trace.sh): every kernel of one timed step
(the bench's last step is its profiled one, branches on one stream; the step before it is the last overlapped one)
with its start offset, duration and queue, and the idle gaps on the critical path.
usage: python tools/train_timeline.py [steps back from the end, default 1]"""
import csv, glob, sys
f = glob.glob("gpurun_out/train_trace/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps are separated by the AdamW kernel
ends = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
back = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lo, hi = ends[-2 - back] + 1, ends[-1 - back] + 1
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"])
busy_until = t0
print("kernels in step:", len(step), " span %.1f us" % ((int(step[-1]["End_Timestamp"]) - t0) / 1e3))
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - busy_until) / 1e3
    busy_until = max(busy_until, e)
    print("%8.1f us  +%6.1f us  gap %6.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
