#!/bin/bash
# step / kernel time of libbean_hip variants (build/variants/libbean_hip_<name>.so) at BASELINE config 3, guides ordered
cd "$(dirname "$0")/../.."
G=${G:-50000}; STEPS=${STEPS:-400}
for rep in 1 2; do for v in "$@"; do
  echo "$v $(BEAN_HIP_LIB=build/variants/libbean_hip_$v.so ORDERS=ordered python scripts/micro/tiling_sorted.py $G $STEPS | tail -1)"
done; done
