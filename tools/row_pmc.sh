#!/bin/bash
# usage: [KERNEL=substring] tools/row_pmc.sh lib...   wave-time split of a training kernel (default tb_row_bwd_kernel) per build (three --pmc passes each: wave cycles by state, instruction counts, L1 requests)
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
P3="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
P2="SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS"
for spec in "$@"; do
  lib=$spec
  n=0
  for P in "$P1" "$P2" "$P3"; do
    n=$((n+1))
    d=/root/repo/gpurun_out/rp_${lib}_$n
    rm -rf $d
    DPTM_LIBRARY=/root/repo/deepptmpred_amd/$lib DPTM_TRAIN_SERIAL=1 timeout -s KILL 200 rocprofv3 --pmc $P --output-format csv -d $d -- python3 /root/repo/bench.py --train --batch 8192 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || exit 1
    f=$(ls -t $d/*/*counter_collection.csv | head -1)
    python3 - "$f" "$spec" "${KERNEL:-tb_row}" <<'PY'
import csv, sys, collections
tot = collections.Counter(); calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[3] in r["Kernel_Name"]:
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); calls[r["Counter_Name"]] += 1
print(sys.argv[2], " ".join("%s=%.4g" % (k.replace("SQ_", ""), v / max(calls[k], 1)) for k, v in sorted(tot.items())))
PY
  done
done
