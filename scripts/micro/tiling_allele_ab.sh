#!/bin/bash
# k_param's allele blocks against k_allele as a launch of its own (BEAN_HIP_ALLELE=split); optionally a build of
# k_param<..., 3> held to seven waves per SIMD (bash scripts/build_variants.sh kp7:"-mllvm -disable-machine-licm -DBEAN_KP3_WAVES=7").
# Same process order for every variant; ORDERS=ordered = the guides as run_inference hands them over.
cd "$(dirname "$0")/../.."
G=${1:-50000}; STEPS=${2:-600}
for rep in 1 2; do
  echo "split   $(BEAN_HIP_ALLELE=split ORDERS=ordered python scripts/micro/tiling_sorted.py $G $STEPS | tail -1)"
  echo "blocks  $(ORDERS=ordered python scripts/micro/tiling_sorted.py $G $STEPS | tail -1)"
  if [ -f build/variants/libbean_hip_kp7.so ]; then
    echo "blocks7 $(BEAN_HIP_LIB=build/variants/libbean_hip_kp7.so ORDERS=ordered python scripts/micro/tiling_sorted.py $G $STEPS | tail -1)"
  fi
done
