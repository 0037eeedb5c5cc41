#!/bin/bash
# usage: tools/prof.sh <tag>   -- runs on the GPU box from the repo root; writes under gpurun_out/prof_<tag>/
set -o pipefail
TAG=${1:-r1}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-split --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; }
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1 || { echo pmc_sq failed; tail -5 $OUT/pmc_sq.log; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1 || { echo pmc_fetch failed; tail -5 $OUT/pmc_fetch.log; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1 || { echo pmc_write failed; tail -5 $OUT/pmc_write.log; }
find $OUT -name "*.csv" | head -30
du -sh $OUT
