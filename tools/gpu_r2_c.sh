#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_ppo.py tests/test_engine.py -q -m gpu 2>&1 | grep -E "^E  |FAILED|passed|failed" | head -20
bash tools/gpu_wgrad_dbg.sh 2>&1 | grep -E "dbg="
