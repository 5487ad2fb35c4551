#!/bin/bash
# Run on the MI355X box from the repository root:  profiles/collect_pmc.sh [tag] [what]
#   what = score : 1. rocprofv3 --kernel-trace --stats of the default bench command (fp32 headline)
#                  2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE, issue counters) of the same command per precision
#   what = train : the same for `bench.py --train` at 8,192 and 64 sites per step (kernel trace with the branches
#                  on one stream, DPTM_TRAIN_SERIAL=1, so that a duration is the kernel's own)
#   what = assemble : FETCH_SIZE / WRITE_SIZE of the window-assembly kernel in `bench.py --proteome`
#   what = classes : vector instructions per class (add / mul / fma / transcendental / conversion / integer) of the
#                  scoring kernels per precision, and the issue cost of each class (tools/issue_cost_probe)
#   what = probe : do SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU count matrix instructions?  (tools/coexec_probe)
# Raw output goes to gpurun_out/prof_<tag>/; profiles/summarize_pmc.py turns it into the committed summaries.
# Every pass is a run of its own under its own time limit; the script stops at the first pass that fails.
set -e
tag=${1:-r2}
what=${2:-score}
repo=$(pwd)
out=$repo/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
ISSUE="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY"
if [ "$what" = score ]; then
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-extras > $out/trace_bench.json 2> $out/trace.log
  echo trace done
  for prec in f32 f16x3 f16 bf16; do
    B="python3 $repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-variants --no-extras --precision $prec"
    timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch_$prec -- $B > $out/fetch_$prec.json 2> $out/fetch_$prec.log
    timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write_$prec -- $B > $out/write_$prec.json 2> $out/write_$prec.log
    timeout -s KILL 180 rocprofv3 --pmc $ISSUE --output-format csv -d $out/mfma_$prec -- $B > $out/mfma_$prec.json 2> $out/mfma_$prec.log
    echo $prec done
  done
elif [ "$what" = train ]; then
  export DPTM_TRAIN_SERIAL=1
  for batch in 8192 64; do
    B="python3 $repo/bench.py --train --batch $batch --steps 20 --warmup 3 --no-cpu-baseline"
    timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/train_trace_b$batch -- $B > $out/train_trace_b$batch.json 2> $out/train_trace_b$batch.log
    B="python3 $repo/bench.py --train --batch $batch --steps 3 --warmup 1 --no-cpu-baseline"
    timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/train_fetch_b$batch -- $B > $out/train_fetch_b$batch.json 2> $out/train_fetch_b$batch.log
    timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/train_write_b$batch -- $B > $out/train_write_b$batch.json 2> $out/train_write_b$batch.log
    timeout -s KILL 180 rocprofv3 --pmc $ISSUE --output-format csv -d $out/train_mfma_b$batch -- $B > $out/train_mfma_b$batch.json 2> $out/train_mfma_b$batch.log
    echo train batch $batch done
  done
elif [ "$what" = assemble ]; then
  B="python3 $repo/bench.py --proteome --steps 1 --warmup 0 --no-cpu-baseline"
  timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/asm_fetch -- $B > $out/asm_fetch.json 2> $out/asm_fetch.log
  timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/asm_write -- $B > $out/asm_write.json 2> $out/asm_write.log
  echo assemble done
elif [ "$what" = classes ]; then
  CLS="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32"
  timeout -k 10 60 $repo/tools/issue_cost_probe > $out/issue_cost.txt 2> $out/issue_cost.log
  for prec in f32 f16x3 f16 bf16; do
    B="python3 $repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-variants --no-extras --precision $prec"
    timeout -s KILL 180 rocprofv3 --pmc $CLS --output-format csv -d $out/cls_$prec -- $B > $out/cls_$prec.json 2> $out/cls_$prec.log
    echo $prec done
  done
elif [ "$what" = probe ]; then
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $out/probe -- $repo/tools/coexec_probe > $out/probe.txt 2> $out/probe.log
  echo probe done
fi
echo collected under $out
