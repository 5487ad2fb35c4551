#!/bin/bash
# usage: tools/ab_kernels.sh [-p "f32 f16x3 ..."] lib1.so [lib2.so ...]  (A/B kernel timings of several builds / precisions on the GPU box)
PRECS="f32"
if [ "$1" = "-p" ]; then PRECS="$2"; shift 2; fi
for lib in "$@"; do
  for prec in $PRECS; do
    echo "== $lib $prec"
    DPTM_LIBRARY=$PWD/deepptmpred_amd/$lib timeout -k 10 300 python -u bench.py --steps 2 --warmup 1 --sites 262144 --no-cpu-baseline --no-variants --precision $prec 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['frac'],4), d['kernel_ms_per_step'])" || exit 1
  done
done
