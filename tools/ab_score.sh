#!/bin/bash
# usage: tools/ab_score.sh lib1.so [lib2.so ...]   A/B of the fp32 scoring path (1,000,000 resident sites, 3 steps) with several builds
for lib in "$@"; do
  DPTM_LIBRARY=$PWD/deepptmpred_amd/$lib timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'],4), d['kernel_ms_per_step'])" || exit 1
done
