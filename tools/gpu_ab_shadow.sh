#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2; do
for v in 0 1; do echo -n "MPPO_NO_SHADOW=$v "; MPPO_NO_SHADOW=$v python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['us_per_launch'])"; done; done
cd /tmp; export TMPDIR=/tmp
for v in 0 1; do MPPO_NO_SHADOW=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ab$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1; echo "== NO_SHADOW=$v"; head -6 $(ls $GRAFT_REPO_ROOT/gpurun_out/ab$v/*/*kernel_stats.csv | head -1) | cut -d, -f1-4 | cut -c1-110; done
