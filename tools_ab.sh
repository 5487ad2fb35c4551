#!/bin/bash
# usage: tools_ab.sh lib1.so lib2.so ...   (A/B branch-kernel timings of several builds on the GPU box)
for lib in "$@"; do
  echo "== $lib"
  DPTM_LIBRARY=$PWD/deepptmpred_amd/$lib timeout -k 10 300 python -u bench.py --steps 2 --warmup 1 --sites 262144 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['frac'],4), d['kernel_ms_per_step'])" || exit 1
done
