for b in 256 512 1024 2048 4096; do for s in 1 0; do
DPTM_TRAIN_SPLIT=$s timeout -k 10 200 python -u bench.py --train --batch $b --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_serial_step']
print('split', $s, 'batch', $b, 'ms/step %.3f' % d['ms_per_step'], ' '.join('%s %.3f'%(n,v) for n,v in k.items() if 'tb_' in n))" || exit 1
done; done
