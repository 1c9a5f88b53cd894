#!/bin/bash
# k_param's allele blocks: how many guide blocks are dispatched ahead of them (BEAN_HIP_ALLELE_AHEAD = guide blocks ahead;
# auto: all of them, the allele blocks are the grid's last blocks)
cd "$(dirname "$0")/../.."
G=${1:-50000}; STEPS=${2:-600}
for v in split auto 0 500 1000 1323; do
  if [ $v = split ]; then r=$(BEAN_HIP_ALLELE=split ORDERS=ordered python scripts/micro/tiling_sorted.py $G $STEPS | tail -1)
  elif [ $v = auto ]; then r=$(ORDERS=ordered python scripts/micro/tiling_sorted.py $G $STEPS | tail -1)
  else r=$(BEAN_HIP_ALLELE_AHEAD=$v ORDERS=ordered python scripts/micro/tiling_sorted.py $G $STEPS | tail -1); fi
  echo "$v $r"
done
