#!/bin/bash
# Run on the MI355X box from the repository root:  profiles/collect_pmc.sh [tag]
# 1. rocprofv3 --kernel-trace --stats of the default bench command (fp32 headline)
# 2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE, MFMA busy) of the same command per precision
# Raw output goes to gpurun_out/prof_<tag>/; profiles/summarize_pmc.py turns it into the
# committed summaries.
set -e
tag=${1:-r1}
repo=$(pwd)
out=$repo/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > $out/trace_bench.json 2> $out/trace.log
for prec in f32 f16x3 bf16; do
  timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch_$prec -- python3 $repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-variants --precision $prec > $out/fetch_$prec.json 2> $out/fetch_$prec.log
  timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write_$prec -- python3 $repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-variants --precision $prec > $out/write_$prec.json 2> $out/write_$prec.log
  timeout -s KILL 180 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA --output-format csv -d $out/mfma_$prec -- python3 $repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-variants --precision $prec > $out/mfma_$prec.json 2> $out/mfma_$prec.log
done
echo collected under $out
