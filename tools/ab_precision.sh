#!/bin/bash
# usage: tools/ab_precision.sh lib1.so [lib2.so ...]   A/B of the half-precision scoring modes (1,000,000 resident sites, 3 steps) with several builds
for lib in "$@"; do
  for prec in bf16 f16 f16x3; do
    DPTM_LIBRARY=$PWD/deepptmpred_amd/$lib timeout -k 10 200 python -u bench.py --precision $prec --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', '$prec', round(d['value']), round(d['ms_per_step'],2), d['kernel_ms_per_step'])" || exit 1
  done
done
