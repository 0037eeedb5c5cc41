#!/bin/bash
# rocprofv3 passes of the split-f16 mode: python3 tools/bench_configs.py --which ff196 --precision split_f16
set -o pipefail
TAG=${1:-r1c_split}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; }
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1 || { echo pmc_sq failed; tail -5 $OUT/pmc_sq.log; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1 || { echo pmc_fetch failed; tail -5 $OUT/pmc_fetch.log; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1 || { echo pmc_write failed; tail -5 $OUT/pmc_write.log; }
du -sh $OUT
