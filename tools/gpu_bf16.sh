#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_kernels_ppo.py tests/test_engine.py tests/test_golden.py -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1
for v in 0 1; do echo -n "bf16 MPPO_NO_SHADOW=$v "; MPPO_NO_SHADOW=$v python bench.py --no-cpu-baseline --steps 30 --set training.mlp_dtype=bf16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['us_per_launch'])"; done
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/bf16prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --set training.mlp_dtype=bf16 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,os
fs=sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/bf16prof/*/*kernel_stats.csv'), key=os.path.getmtime)
for r in list(csv.DictReader(open(fs[-1])))[:6]:
    print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,2))
PY
