#!/bin/bash
# usage: tools/ab_bench.sh "precisions" lib...  : default 1M-site bench per library
PRECS="$1"; shift
for lib in "$@"; do for prec in $PRECS; do
  echo "== $lib $prec"
  DPTM_LIBRARY=$PWD/deepptmpred_amd/$lib timeout -k 10 300 python -u bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-variants --precision $prec 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['frac'],4), d['ms_per_step'])" || exit 1
done; done
