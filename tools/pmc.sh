#!/bin/bash
# usage: tools/pmc.sh <precision> <tag> [counters...] ; collects SQ counters for the branch kernels
prec=$1; tag=$2; shift 2
ctrs="$@"
[ -z "$ctrs" ] && ctrs="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d /root/repo/gpurun_out/pmc_$tag -- python3 /root/repo/bench.py --steps 1 --warmup 0 --sites 65536 --no-cpu-baseline --no-variants --precision $prec > /root/repo/gpurun_out/pmc_$tag.log 2>&1
