#!/bin/bash
# k_guide_tiling_rep: which guides a workgroup takes (BEAN_HIP_TILING_MAP: 0 in order; 1 an XCD takes contiguous runs,
# one per quarter of the allele-count order)
cd "$(dirname "$0")/../.."
G=${G:-50000}; STEPS=${STEPS:-600}; LIB=${LIB:-crispr-bean_amd/lib/libbean_hip.so}
for rep in 1 2; do for m in 0 1; do
  echo "map=$m $(BEAN_HIP_LIB=$LIB BEAN_HIP_TILING_MAP=$m ORDERS=ordered python scripts/micro/tiling_sorted.py $G $STEPS | tail -1)"
done; done
