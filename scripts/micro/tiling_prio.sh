#!/bin/bash
# k_guide_tiling_rep: issue priority by progress (BEAN_HIP_TILING_PRIO: 0 off, 1 on = the default)
cd "$(dirname "$0")/../.."
G=${G:-50000}; STEPS=${STEPS:-600}; LIB=${LIB:-crispr-bean_amd/lib/libbean_hip.so}
for rep in 1 2; do for m in 0 1; do
  echo "prio=$m $(BEAN_HIP_LIB=$LIB BEAN_HIP_TILING_PRIO=$m ORDERS=ordered python scripts/micro/tiling_sorted.py $G $STEPS | tail -1)"
done; done
