#!/bin/bash
# rocprofv3 passes for one bench configuration; run on the GPU box via gpurun:
#   gpurun -- 'bash scripts/profile.sh r02 [metric|tiling|survival]'
# Writes CSVs under gpurun_out/prof_<tag>[_<config>]/ ; scripts/summarize_prof.py condenses them
# into profiles/.  Counters are collected in their own passes (kernel-trace only beside --pmc);
# FETCH_SIZE and WRITE_SIZE do not fit in one pass.
set -e
TAG=${1:-dev}
CONFIG=${2:-metric}     # metric | tiling | survival, or <config>_acc: the same workload with --scale-by-acc
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
SUF=""; [ "$CONFIG" != "metric" ] && SUF="_$CONFIG"
OUT=$REPO/gpurun_out/prof_$TAG$SUF
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BASE=${CONFIG%_acc}; ACC=""; [ "$BASE" != "$CONFIG" ] && ACC="--scale-by-acc"
B="python3 $REPO/bench.py --config $BASE $ACC --no-cpu-baseline --no-strong --no-other-configs"
# pass 1: per-kernel time (graph replay, as the bench runs it)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $B --steps 400 --warmup 50 > $OUT/bench_kt.json 2> $OUT/kt.err
# pass 2..4: PMC counters, eager launches
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- $B --steps 20 --warmup 4 --graph-chunk 0 > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $B --steps 20 --warmup 4 --graph-chunk 0 > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $B --steps 20 --warmup 4 --graph-chunk 0 > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
# (summarised where the files are kept: python3 scripts/summarize_prof.py gpurun_out/prof_<tag>[_<config>] <tag> <config>)
cd $REPO && python3 scripts/summarize_prof.py $OUT $TAG $CONFIG > $OUT/summary.log 2>&1 || true
