#!/bin/bash
# Build kernel A/B variants of libbean_hip.so into build/variants/ (git-ignored, travels with gpurun).
#   bash scripts/build_variants.sh name1:"-DFLAG=1 ..." name2:"..." ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build/variants
mkdir -p $OUT
pids=()
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  [ "$flags" = "$spec" ] && flags=""
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -munsafe-fp-atomics $flags \
      $ROOT/crispr-bean_amd/csrc/bean_hip.hip -o $OUT/libbean_hip_$name.so && echo "built $name [$flags]" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
ls -la $OUT
