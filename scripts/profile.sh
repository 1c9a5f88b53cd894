#!/bin/bash
# rocprofv3 passes for the bench workload; run on the GPU box via gpurun:
#   gpurun -- 'bash scripts/profile.sh r01'
# Writes CSVs under gpurun_out/prof_<tag>/ ; copy the summaries to profiles/.
set -e
TAG=${1:-dev}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
EXTRA=${BENCH_EXTRA:-}
# pass 1: per-kernel time (graph replay, as the bench runs it)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $REPO/bench.py --steps 400 --warmup 50 --no-cpu-baseline $EXTRA > $OUT/bench_kt.json 2> $OUT/kt.err
# pass 2..4: PMC counters, eager launches, separate passes (FETCH_SIZE and WRITE_SIZE do not fit together)
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py --steps 20 --warmup 4 --graph-chunk 0 --no-cpu-baseline $EXTRA > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 20 --warmup 4 --graph-chunk 0 --no-cpu-baseline $EXTRA > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 20 --warmup 4 --graph-chunk 0 --no-cpu-baseline $EXTRA > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
find $OUT -name "*.csv" | head -40
