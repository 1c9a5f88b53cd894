#!/bin/bash
# What the pieces of k_guide_tiling_rep cost in place at BASELINE config 3 (wrong results: -DBEAN_TL_DIAG=n builds,
# bash scripts/build_variants.sh d1:"-mllvm -disable-machine-licm -DBEAN_TL_DIAG=1" d2:... d3:...): 1 without the rejection
# loops of the draw, 2 without the implicit-gradient calls, 3 without the lgamma / digamma differences.
cd "$(dirname "$0")/../.."
G=${1:-50000}; STEPS=${2:-300}
echo "full $(ORDERS=ordered python scripts/micro/tiling_sorted.py $G $STEPS | tail -1)"
for v in d1 d2 d3; do
  echo "$v $(BEAN_HIP_LIB=build/variants/libbean_hip_$v.so ORDERS=ordered python scripts/micro/tiling_sorted.py $G $STEPS | tail -1)"
done
