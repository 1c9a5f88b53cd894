#!/bin/bash
# End-of-round measurement pass (one gpurun call): profiles of the three bench configurations, the
# BASELINE configuration timings, and the bench lines themselves.
set -x
TAG=${1:-r02}
mkdir -p gpurun_out
for cfg in metric tiling survival; do
  bash scripts/profile.sh $TAG $cfg > gpurun_out/prof_$cfg.log 2>&1 || echo "profile $cfg failed"
done
python scripts/time_configs.py > gpurun_out/time_configs.log 2>&1 || echo "time_configs failed"
python bench.py > gpurun_out/bench_metric.json 2> gpurun_out/bench_metric.err
python bench.py --config tiling --steps 500 --warmup 50 > gpurun_out/bench_tiling.json 2> gpurun_out/bench_tiling.err
python bench.py --config survival --steps 1000 --warmup 50 > gpurun_out/bench_survival.json 2> gpurun_out/bench_survival.err
python bench.py --scaling strong --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/bench_strong1.json 2> gpurun_out/bench_strong1.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_short.json 2> gpurun_out/bench_short.err
tail -c 600 gpurun_out/bench_metric.json; echo; tail -c 300 gpurun_out/bench_short.json
