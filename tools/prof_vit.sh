#!/bin/bash
# usage: tools/prof_vit.sh <tag>  -- on the GPU box from the repo root: kernel trace + HBM-traffic PMC passes (one counter set per run)
# of the frame-synchronous Viterbi decoder on the real-size synthetic graph (tools/bench_viterbi.py)
set -o pipefail
TAG=${1:-vit}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 tools/bench_viterbi.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1 || { echo pmc_fetch failed; tail -5 $OUT/pmc_fetch.log; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1 || { echo pmc_write failed; tail -5 $OUT/pmc_write.log; }
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- $B > $OUT/pmc_tcc.log 2>&1 || { echo pmc_tcc failed; tail -5 $OUT/pmc_tcc.log; }
# where the waves' time goes: parked on s_waitcnt / barriers (WAIT_ANY), stalled at issue (WAIT_INST_ANY), issuing (ACTIVE_INST_ANY); occupancy = WAVE_CYCLES / BUSY_CYCLES
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1 || { echo pmc_sq failed; tail -5 $OUT/pmc_sq.log; }
f=$(find $OUT/trace -name "*_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv
for d in pmc_fetch pmc_write pmc_tcc pmc_sq; do python3 tools/pmc_by_kernel.py $OUT/$d > $OUT/$d.txt 2>&1; done
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_tcc $OUT/pmc_sq
grep -h "viterbi" $OUT/pmc_*.txt | cut -c1-300; grep viterbi $OUT/kernel_stats.csv | cut -c1-200
