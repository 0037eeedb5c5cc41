#!/bin/bash
# usage: tools/prof_lstm.sh <tag> [AM_LSTM_PERSIST mode]  -- on the GPU box from the repo root: kernel trace + PMC passes (each in
# its own run) of the 3x512 recurrent configuration; summaries under gpurun_out/prof_<tag>/ (copied to profiles/ by the caller)
set -o pipefail
TAG=${1:-lstm}
export AM_LSTM_PERSIST=${2:-2}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 tools/bench_configs.py --which ${PROF_WHICH:-lstm} --steps 2"   # PROF_WHICH=lstm1024: the streamed persistent form on 4 x 1024
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; }
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1 || { echo pmc_sq failed; tail -5 $OUT/pmc_sq.log; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1 || { echo pmc_fetch failed; tail -5 $OUT/pmc_fetch.log; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1 || { echo pmc_write failed; tail -5 $OUT/pmc_write.log; }
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- $B > $OUT/pmc_tcc.log 2>&1 || { echo pmc_tcc failed; tail -5 $OUT/pmc_tcc.log; }
f=$(find $OUT/trace -name "*_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv
for d in pmc_sq pmc_fetch pmc_write pmc_tcc; do python3 tools/pmc_by_kernel.py $OUT/$d > $OUT/$d.txt 2>&1; done
rm -rf $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_tcc   # the per-dispatch tables are large; the per-kernel averages are what is kept
rm -rf $OUT/trace
grep -h "persist\|wreg\|thin\|stream" $OUT/pmc_*.txt | cut -c1-400
