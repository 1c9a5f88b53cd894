#!/bin/bash
# run-to-run spread of the tiling step at a small size: N processes each of k_param's allele blocks and of k_allele as a launch
cd "$(dirname "$0")/../.."
G=${1:-6250}; N=${2:-8}; STEPS=${3:-400}
for mode in blocks split; do
  for i in $(seq $N); do
    if [ $mode = split ]; then r=$(BEAN_HIP_ALLELE=split ORDERS=ordered python scripts/micro/tiling_sorted.py $G $STEPS | tail -1); else r=$(ORDERS=ordered python scripts/micro/tiling_sorted.py $G $STEPS | tail -1); fi
    echo "$mode $(echo $r | python -c "import json,sys; d=json.loads(sys.stdin.read().split(' ',1)[1]); print(d['us_per_step'], d['kernel_us'])")"
  done
done
