#!/bin/bash
# kernel-trace stats of a short bench run: bash scripts/prof_quick.sh <tag> [bench args...]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/kt_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/bench.py --steps 300 --warmup 40 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/err.log
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cat $f | cut -c1-160
