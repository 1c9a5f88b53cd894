#!/bin/bash
# time the tiling config with library variants: bash scripts/time_tiling_variants.sh lib1.so lib2.so ...
for lib in "$@"; do
  echo "== $lib"
  BEAN_HIP_LIB=$lib timeout -k 10 300 python scripts/time_configs.py tiling 2>&1 | grep -E "us_per_step|kernel_us"
done
