#!/bin/bash
# usage (on the MI355X box): bash tools/train_trace.sh [batch] [serial]   -- per-kernel stats of the training step;
# "serial" puts the three branches on the caller's stream so that kernel durations are not inflated by overlap
cd /tmp && export TMPDIR=/tmp
if [ "$2" = "serial" ]; then export DPTM_TRAIN_SERIAL=1; fi
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/train_trace -- python3 /root/repo/bench.py --train --batch ${1:-64} --steps 20 --warmup 3 > /root/repo/gpurun_out/train_trace.log 2>&1
