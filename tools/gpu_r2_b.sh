#!/bin/bash
# A/B of the fused weight-gradient + reduction kernel (k_wgrad.hip) against round 1's two launches
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "=== pytest -m gpu (ppo kernels, engine, golden)"; timeout 1200 python -m pytest tests/test_kernels_ppo.py tests/test_engine.py tests/test_golden.py tests/test_train_surface.py tests/test_c_example.py -q -m gpu 2>&1 | grep -E "^E  |FAILED|passed|failed" | head -30
for v in 0 1; do
  echo "=== bench MPPO_OLD_WGRAD=$v"; MPPO_OLD_WGRAD=$v timeout 600 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['sanity'])"
done
echo "=== rocprof"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2b_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2b_bench_prof.json 2>/dev/null
cd $GRAFT_REPO_ROOT; python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/r2b_prof/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print("%-58s calls %5s avg %8.1f us  total %7.2f ms"%(r["Name"].split("(")[0][:58], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
