#!/bin/bash
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /root/repo/gpurun_out/host_trace -- python3 /root/repo/scratch/host_path.py 131072 > /root/repo/gpurun_out/host_trace.log 2>&1
