#!/bin/bash
# rocprofv3 passes for the metric configuration when it steps through k_svi_async (one launch = one report window of
# 100 steps): as scripts/profile.sh, with every launch of the counter passes 100 steps long so that the summary can
# give counters per SVI step.   gpurun -- 'bash scripts/profile_async.sh r05'
set -e
TAG=${1:-dev}
CONFIG=${2:-metric}     # metric | metric_acc (the same workload with --scale-by-acc)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
SUF=""; [ "$CONFIG" != "metric" ] && SUF="_$CONFIG"
OUT=$REPO/gpurun_out/prof_$TAG$SUF
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ACC=""; [ "$CONFIG" = "metric_acc" ] && ACC="--scale-by-acc"
B="python3 $REPO/bench.py --config metric $ACC --no-cpu-baseline --no-strong --no-other-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $B --steps 400 --warmup 100 > $OUT/bench_kt.json 2> $OUT/kt.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- $B --steps 200 --warmup 100 > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $B --steps 200 --warmup 100 > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $B --steps 200 --warmup 100 > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
cd $REPO && ASYNC_STEPS_PER_LAUNCH=100 python3 scripts/summarize_prof.py $OUT $TAG $CONFIG > $OUT/summary.log 2>&1 || true
