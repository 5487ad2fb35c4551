#!/bin/bash
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /root/repo/gpurun_out/small_trace -- python3 /root/repo/tools/latency_small_call.py > /root/repo/gpurun_out/small_trace.log 2>&1
