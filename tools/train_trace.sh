#!/bin/bash
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/train_trace -- python3 /root/repo/bench.py --train --batch ${1:-64} --steps 20 --warmup 3 > /root/repo/gpurun_out/train_trace.log 2>&1
