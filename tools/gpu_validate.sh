#!/bin/bash
# round-end style validation: GPU tests, smoke, default bench, rocprof kernel stats of the bench command
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "=== pytest -m gpu"; timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|^FAILED" gpurun_out/pytest_gpu.log | tail -5
echo "=== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "=== bench (default flags)"; timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 2500 gpurun_out/bench_final.json
echo "=== rocprof"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_final -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.json 2>/dev/null
cd $GRAFT_REPO_ROOT; python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_final/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print("%-58s calls %5s avg %8.1f us  total %7.2f ms"%(r["Name"].split("(")[0][:58], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
