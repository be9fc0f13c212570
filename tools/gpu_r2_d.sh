#!/bin/bash
# round 2, after the contact-frame / geom-pair change: GPU suite, env kernel time, bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "^E  |FAILED|passed|failed" | head -20
timeout 300 python tools/env_time.py 2>&1 | tail -3
timeout 600 python bench.py > gpurun_out/bench_d.json 2> gpurun_out/bench_d.err; tail -c 600 gpurun_out/bench_d.json
