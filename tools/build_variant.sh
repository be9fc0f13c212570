#!/bin/bash
# tools/build_variant.sh <name> <source.hip> <extra flags...>: builds tools/_variants/libminppo_<name>.so with ONE source file recompiled
# with the extra flags and every other object taken from the product build (profiling / A-B variants; never loaded by the product).
# The recompiled file is built with -DMPPO_EXPERIMENTS: the measurement switches (MPPO_FUSED_SKIP, MPPO_EXTRA_LAUNCHES, MPPO_WGRAD_DBG,
# MPPO_NO_FUSED / _SHADOW / _PREGATHER, MPPO_KSPLIT, MPPO_GEMM_IMPL, MPPO_PEER_POLL_RMW, MPPO_PEER_ALLOC) exist in such a file only.
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; shift 2
mkdir -p tools/_variants
STEM=$(basename "$SRC" .hip)
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Iminppo_amd/csrc -Iinclude -Wno-unused-result -DMPPO_EXPERIMENTS"
if [ "$STEM" = "k_physics" ]; then BASE="$BASE -ffp-contract=off"; fi
/opt/rocm/bin/hipcc $BASE "$@" -c minppo_amd/csrc/$SRC -o tools/_variants/${STEM}_${NAME}.o
OBJS=$(ls minppo_amd/csrc/_build/*.o | grep -v "/${STEM}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_variants/libminppo_${NAME}.so $OBJS tools/_variants/${STEM}_${NAME}.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo "built tools/_variants/libminppo_${NAME}.so"
