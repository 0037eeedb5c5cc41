#!/bin/bash
# usage: tools/prof_all.sh <tag>  -- on the GPU box from the repo root: the bench.py passes of tools/prof.sh plus kernel-trace
# statistics of the secondary configs (recurrent cells, TDNN, decoder); summaries are copied to profiles/ by the caller.
set -o pipefail
TAG=${1:-r1}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
./tools/prof.sh $TAG > $OUT/prof_sh.log 2>&1 || echo "prof.sh failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rec -- python3 tools/bench_configs.py --which lstm --steps 3 > $OUT/rec.log 2>&1 || echo "rec trace failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cells -- python3 tools/bench_configs.py --which cells --steps 2 > $OUT/cells.log 2>&1 || echo "cells trace failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tdnn -- python3 tools/bench_configs.py --which tdnn --steps 2 > $OUT/tdnn.log 2>&1 || echo "tdnn trace failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vit -- python3 tools/bench_viterbi.py > $OUT/vit.log 2>&1 || echo "viterbi trace failed"
for d in rec cells tdnn vit; do f=$(find $OUT/$d -name "*_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${d}_kernel_stats.csv; done
rm -rf $OUT/rec $OUT/cells $OUT/tdnn $OUT/vit   # the raw traces are large; the stats tables are what is kept
du -sh $OUT
