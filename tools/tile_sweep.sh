#!/bin/bash
# usage: tools/tile_sweep.sh ["batches"] ["candidates"]  (candidate -1 = the library's own choice)   serial kernel times of the training step per batch size and forced tile candidate (GPU box)
for b in ${1:-128 256 512 1024 2048 4096}; do for t in ${2:-0 1 2 3}; do
DPTM_TRAIN_TILE=$t timeout -k 10 200 python -u bench.py --train --batch $b --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_serial_step']
g=lambda w: sum(v for n,v in k.items() if n.endswith(str(w)))
print('cand', $t, 'batch', $b, 'ms/step %.3f' % d['ms_per_step'], ' '.join('%s %.3f'%(n,v) for n,v in k.items() if 'tb_' in n), 'branch21 %.3f branch33 %.3f branch51 %.3f' % (g(21), g(33), g(51)))" || exit 1
done; done
