#!/bin/bash
# End-of-round measurement pass (one gpurun call): profiles of the three bench configurations, the
# BASELINE configuration timings, the exchange-path and wide-tiling timings, and the bench lines themselves.
# Every step is bounded by its own timeout; a step that was killed ends the pass (no further GPU work).
TAG=${1:-r04}
mkdir -p gpurun_out
step() { echo "== $*" >&2; timeout -k 10 "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping"; exit $rc; fi; }
for cfg in metric tiling survival; do
  step 400 bash scripts/profile.sh $TAG $cfg > gpurun_out/prof_$cfg.log 2>&1
done
step 500 python scripts/time_configs.py > gpurun_out/time_configs.log 2>&1
step 400 python scripts/time_exchange.py > gpurun_out/exchange.log 2>&1
step 300 python scripts/time_tiling_wide.py 5000 60 > gpurun_out/tiling_wide.log 2>&1
step 300 python bench.py > gpurun_out/bench_metric.json 2> gpurun_out/bench_metric.err
step 300 python bench.py --config tiling --steps 500 --warmup 50 > gpurun_out/bench_tiling.json 2> gpurun_out/bench_tiling.err
step 300 python bench.py --config survival --steps 1000 --warmup 50 > gpurun_out/bench_survival.json 2> gpurun_out/bench_survival.err
step 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_short.json 2> gpurun_out/bench_short.err
tail -c 700 gpurun_out/bench_metric.json; echo; tail -c 300 gpurun_out/bench_short.json
