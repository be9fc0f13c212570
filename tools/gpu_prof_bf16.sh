#!/bin/bash
# rocprofv3 kernel stats of the bf16 bench line (BASELINE configs[3])
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bf16 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --set training.mlp_dtype=bf16 > $GRAFT_REPO_ROOT/gpurun_out/bench_prof_bf16.json 2>/dev/null
cd $GRAFT_REPO_ROOT; cp $(ls gpurun_out/prof_bf16/*/*kernel_stats.csv | head -1) gpurun_out/kernel_stats_bf16.csv
python3 - <<'PY'
import csv
for r in list(csv.DictReader(open("gpurun_out/kernel_stats_bf16.csv")))[:8]:
    print("%-60s calls %5s avg %8.2f us  total %7.2f ms" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
