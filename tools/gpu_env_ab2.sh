#!/bin/bash
# second pass of tools/gpu_env_ab.sh: ramp shape of a fresh process, then longer runs (60 timed updates after 30 warm-up), each setting twice, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/env_ab2.txt; : > $OUT
echo "=== ramp (graph)"; timeout 300 python tools/ramp_probe.py 60 2>&1 | tail -4 | tee -a $OUT
echo "=== ramp (graph), again"; timeout 300 python tools/ramp_probe.py 60 2>&1 | tail -4 | tee -a $OUT
echo "=== ramp (eager)"; timeout 300 python tools/ramp_probe.py 60 --no-graph 2>&1 | tail -4 | tee -a $OUT
echo "=== ramp (eager, HIP_FORCE_DEV_KERNARG=1)"; HIP_FORCE_DEV_KERNARG=1 timeout 300 python tools/ramp_probe.py 60 --no-graph 2>&1 | tail -4 | tee -a $OUT
run() {
  local tag="$1"; shift
  local line
  line=$(env "$@" timeout 300 python bench.py --steps 60 --warmup 30 --no-cpu-baseline ${EXTRA} 2>/dev/null | tail -1)
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
for rep in 1 2; do
EXTRA=""
run "default" MPPO_AB=0
run "HIP_FORCE_DEV_KERNARG=1" HIP_FORCE_DEV_KERNARG=1
run "HIP_FORCE_DEV_KERNARG=0" HIP_FORCE_DEV_KERNARG=0
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run "ROC_SYSTEM_SCOPE_SIGNAL=0" ROC_SYSTEM_SCOPE_SIGNAL=0
EXTRA="--no-graph"
run "eager" MPPO_AB=0
run "eager HIP_FORCE_DEV_KERNARG=1" HIP_FORCE_DEV_KERNARG=1
run "eager HIP_FORCE_DEV_KERNARG=0" HIP_FORCE_DEV_KERNARG=0
done
echo "=== default flags, three times (what the driver runs)"
for k in 1 2 3; do timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default-flags run: %.3f M  %.3f ms' % (d['value']/1e6, d['ms_per_step']))" | tee -a $OUT; done
