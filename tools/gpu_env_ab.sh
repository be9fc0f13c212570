#!/bin/bash
# A/B of HIP runtime settings that touch the per-launch floor (every kernel inside the hipGraph shows >= 4.6 us in rocprofv3,
# a one-workgroup kernel included): kernel arguments in device memory, graph packet capture, fence scope, eager launches.
# One short bench line per setting; results in gpurun_out/env_ab.txt
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/env_ab.txt; : > $OUT
run() {
  local tag="$1"; shift
  local line
  line=$(env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ${EXTRA} 2>/dev/null | tail -1)
  python3 - "$tag" "$line" >> $OUT <<'PY'
import json, sys
tag, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    print("%-44s %8.3f M env-steps/s  %7.3f ms/update  row pass %6.2f us" % (tag, d["value"] / 1e6, d["ms_per_step"], d["roofline"].get("us_per_launch", -1)))
except Exception as e:
    print("%-44s FAILED (%s) %s" % (tag, e, line[:200]))
PY
  tail -1 $OUT
}
EXTRA=""
run "default" MPPO_AB=0
run "HIP_FORCE_DEV_KERNARG=1" HIP_FORCE_DEV_KERNARG=1
run "HIP_FORCE_DEV_KERNARG=0" HIP_FORCE_DEV_KERNARG=0
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run "AMD_OPT_FLUSH=0" AMD_OPT_FLUSH=0
run "AMD_OPT_FLUSH=1" AMD_OPT_FLUSH=1
run "ROC_SYSTEM_SCOPE_SIGNAL=0" ROC_SYSTEM_SCOPE_SIGNAL=0
run "DEV_KERNARG=1 + PACKET_CAPTURE=1" HIP_FORCE_DEV_KERNARG=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
EXTRA="--no-graph"
run "eager" MPPO_AB=0
run "eager HIP_FORCE_DEV_KERNARG=1" HIP_FORCE_DEV_KERNARG=1
run "eager HIP_FORCE_DEV_KERNARG=0" HIP_FORCE_DEV_KERNARG=0
run "eager AMD_OPT_FLUSH=0" AMD_OPT_FLUSH=0
echo "=== rocprof of the eager run (kernel durations outside a graph)"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_eager -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_eager/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-58s calls %5s avg %8.2f us min %8.2f" % (r["Name"].split("(")[0][:58], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
