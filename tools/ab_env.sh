#!/bin/bash
# usage: tools/ab_env.sh VAR "v1 v2 ..." [batches]   A/B of the training step under values of one environment switch (GPU box)
VAR=$1; VALS=$2; BATCHES=${3:-64}
for v in $VALS; do
  for b in $BATCHES; do
    steps=5; [ "$b" -le 512 ] && steps=50
    env $VAR=$v timeout -k 10 200 python -u bench.py --train --batch $b --steps $steps --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_serial_step']
print('$VAR=$v', 'batch', $b, 'ms/step %.3f' % d['ms_per_step'], 'tb_bwd51 %.4f tb_fwd51 %.4f' % (k['tb_bwd51'], k['tb_fwd51']), 'bwd %.3f fwd %.3f conv %.3f' % (sum(v for n,v in k.items() if 'tb_bwd' in n), sum(v for n,v in k.items() if 'tb_fwd' in n), sum(v for n,v in k.items() if 'conv' in n)))" || exit 1
  done
done
