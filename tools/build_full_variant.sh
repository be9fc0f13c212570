#!/bin/bash
# tools/build_full_variant.sh <name> <extra flags...>: builds tools/_variants/libminppo_<name>.so with EVERY source recompiled with the extra
# flags (A/B builds of switches that several translation units must agree on, e.g. a data layout); never loaded by the product.
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p tools/_variants/full_$NAME
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iminppo_amd/csrc -Iinclude -Wno-unused-result"
pids=""
for f in $(cat minppo_amd/csrc/SOURCES.txt minppo_amd/csrc/SOURCES_DEVICE_ONLY.txt); do
  STEM=$(basename "$f" .hip)
  FP="-ffp-contract=fast"; if [ "$STEM" = "k_physics" ]; then FP="-ffp-contract=off"; fi
  /opt/rocm/bin/hipcc $BASE $FP "$@" -c minppo_amd/csrc/$f -o tools/_variants/full_$NAME/$STEM.o &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_variants/libminppo_${NAME}.so tools/_variants/full_$NAME/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo "built tools/_variants/libminppo_${NAME}.so"
