#!/bin/bash
# End-of-round measurement pass: profiles of the three bench configurations (and their +Acc instantiations), the
# BASELINE configuration timings, the exchange-path and wide-tiling timings, and the bench lines themselves.
#   gpurun --timeout 1100 -- 'bash scripts/final_round.sh r04 1'   profiles of metric / tiling / survival
#   gpurun --timeout 1100 -- 'bash scripts/final_round.sh r04 2'   profiles of the +Acc instantiations, wide tiling
#   gpurun --timeout 1100 -- 'bash scripts/final_round.sh r04 3'   configuration timings and the bench lines
# Every step is bounded by its own timeout; a step that was killed ends the pass (no further GPU work).
# Afterwards, here: for c in metric tiling survival metric_acc tiling_acc survival_acc; do
#   python3 scripts/summarize_prof.py gpurun_out/prof_<tag>[_$c] <tag> $c; done
TAG=${1:-r04}
PART=${2:-1}
mkdir -p gpurun_out
step() { echo "== $*" >&2; timeout -k 10 "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping" >&2; exit $rc; fi; }
if [ "$PART" = "1" ]; then
  for cfg in metric tiling survival; do
    step 300 bash scripts/profile.sh $TAG $cfg > gpurun_out/prof_$cfg.log 2>&1
  done
elif [ "$PART" = "2" ]; then
  for cfg in metric_acc tiling_acc survival_acc; do
    step 300 bash scripts/profile.sh $TAG $cfg > gpurun_out/prof_$cfg.log 2>&1
  done
  step 300 python scripts/time_tiling_wide.py 5000 60 > gpurun_out/tiling_wide.log 2>&1
  step 300 bash scripts/profile_wide.sh $TAG > gpurun_out/prof_wide.log 2>&1
else
  step 500 python scripts/time_configs.py > gpurun_out/time_configs.log 2>&1
  XCHG_CASES=tiling/8 python scripts/time_exchange.py warm > /dev/null 2>&1   # (the first process on a fresh box pages the image in)
  rm -f gpurun_out/exchange_1rank.json   # one case per process (a host-bound loop is sensitive to what ran before it)
  for cs in tiling/1 tiling/8 survival/1 survival/8; do
    XCHG_CASES=$cs step 200 python scripts/time_exchange.py >> gpurun_out/exchange.log 2>&1
  done
  step 400 python bench.py > gpurun_out/bench_metric.json 2> gpurun_out/bench_metric.err
  step 300 python bench.py --config tiling --steps 500 --warmup 50 > gpurun_out/bench_tiling.json 2> gpurun_out/bench_tiling.err
  step 300 python bench.py --config survival --steps 1000 --warmup 50 > gpurun_out/bench_survival.json 2> gpurun_out/bench_survival.err
  step 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_short.json 2> gpurun_out/bench_short.err
  tail -c 700 gpurun_out/bench_metric.json; echo; tail -c 300 gpurun_out/bench_short.json
fi
