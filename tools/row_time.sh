#!/bin/bash
# usage: tools/row_time.sh lib...   average duration of the training kernels per build (rocprofv3 kernel trace, batch 8192, serial branches)
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  lib=$spec
  rm -rf /root/repo/gpurun_out/rt_${lib}
  DPTM_LIBRARY=/root/repo/deepptmpred_amd/$lib DPTM_TRAIN_SERIAL=1 timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/rt_${lib} -- python3 /root/repo/bench.py --train --batch 8192 --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  f=$(ls -t /root/repo/gpurun_out/rt_${lib}/*/*kernel_stats.csv | head -1)
  python3 - "$f" "$spec" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = []
for r in rows:
    n = r["Name"]
    if "tb_row" in n or "tb_bwd" in n:
        out.append("%s %.4f ms x%s" % (n.split("(")[0].replace("void dptm::", "").replace("dptm::", ""), float(r["AverageNs"]) / 1e6, r["Calls"]))
print(sys.argv[2], " | ".join(sorted(out)))
PY
done
