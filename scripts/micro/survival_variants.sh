#!/bin/bash
# step time of libbean_hip variants (build/variants/libbean_hip_<name>.so) at BASELINE config 5 (bench.py --config survival)
cd "$(dirname "$0")/../.."
for rep in 1 2; do for v in "$@"; do
  r=$(BEAN_HIP_LIB=build/variants/libbean_hip_$v.so python bench.py --config survival --steps 1000 --warmup 100 --no-cpu-baseline --no-strong 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), 'us per step; kernel', round(d['roofline']['kernel_ms']*1e3,2), 'loss', d['config']['final_loss'])")
  echo "$v $r"
done; done
