#!/bin/bash
# usage: tools/pair_sweep.sh "batches"   step time per batch size: default policy | one-site pair tiles forced, three streams | forced, one launch for the three branches
run() { env "$@" timeout -k 10 200 python -u bench.py --train --batch $B --steps 50 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.3f' % d['ms_per_step'], end=' ')" || exit 1; }
for B in $1; do
  echo -n "batch $B: default "; run X=1
  echo -n " pair/streams "; run DPTM_TRAIN_TILE=2
  echo -n " pair/merged "; run DPTM_TRAIN_TILE=2 DPTM_TRAIN_MERGE_WGS=100000
  echo
done
