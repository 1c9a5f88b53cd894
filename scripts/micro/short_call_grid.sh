#!/bin/bash
# the driver's 20-step call of the metric shape under other grids of k_svi_async (item waves, finisher waves, split finish)
cd "$(dirname "$0")/../.."
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strong --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), 'us per step (20-step call); sustained', round(d['sustained']['ms_per_step']*1e3,2) if d.get('sustained') else None)"; }
for rep in 1 2; do
echo "default            $(run)"
echo "blocks=3072        $(BEAN_HIP_ASYNC_BLOCKS=3072 run)"
echo "blocks=2048 fin=0  $(BEAN_HIP_ASYNC_BLOCKS=2048 BEAN_HIP_ASYNC_FIN=0 run)"
echo "blocks=3072 fin=0  $(BEAN_HIP_ASYNC_BLOCKS=3072 BEAN_HIP_ASYNC_FIN=0 run)"
echo "pair path          $(BEAN_HIP_STEP=pair run)"
done
