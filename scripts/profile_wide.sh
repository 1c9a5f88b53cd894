#!/bin/bash
# rocprofv3 kernel stats of the wide tiling path: gpurun -- 'bash scripts/profile_wide.sh r03'
TAG=${1:-dev}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_${TAG}_tiling_wide
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
WIDE_NO_PROFILE=1 WIDE_STEPS=60 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $REPO/scripts/time_tiling_wide.py 5000 60 > $OUT/run.json 2> $OUT/kt.err
WIDE_NO_PROFILE=1 WIDE_STEPS=10 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $REPO/scripts/time_tiling_wide.py 5000 60 > $OUT/run_pmc.json 2> $OUT/pmc.err
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); grep -E "Name|bean::" $f | cut -c1-170
