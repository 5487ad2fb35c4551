#!/bin/bash
# usage (on the MI355X box): bash tools/pmc_train.sh <tag> <kernel-substring> counters...   -- SQ counters of one training kernel
tag=$1; shift; export PMC_KERNEL=$1; shift
cd /tmp && export TMPDIR=/tmp
export DPTM_TRAIN_SERIAL=1
timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d /root/repo/gpurun_out/pmc_$tag -- python3 /root/repo/bench.py --train --batch 8192 --steps 2 --warmup 1 > /root/repo/gpurun_out/pmc_$tag.log 2>&1
python3 - <<PY
import csv, collections, glob, os
k = os.environ["PMC_KERNEL"]
agg = collections.defaultdict(list)
for f in glob.glob("/root/repo/gpurun_out/pmc_$tag/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if k in r["Kernel_Name"] and r["Grid_Size"] == max_grid if False else k in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(agg.items()):
    print("$tag", c, "mean %.3g over %d launches" % (sum(v) / len(v), len(v)))
PY
