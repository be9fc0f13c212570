#!/bin/bash
# times env_kernel for every library under tools/_variants (A/B builds made with tools/build_variant.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
for f in tools/_variants/libminppo_*.so; do echo "== $f"; timeout 200 python tools/env_time.py $f 2>&1 | grep "us per"; done
