#!/bin/bash
# first GPU contact: smoke, bench (graph / eager), rocprofv3 kernel trace
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -20
echo "=== bench eager"; timeout 900 python bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline 2>&1 | tail -5
echo "=== bench graph"; timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -5
echo "=== rocprof"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline 2>&1 | tail -5
cd $GRAFT_REPO_ROOT; find gpurun_out/prof1 -name "*stats*" | head; for f in $(find gpurun_out/prof1 -name "*kernel_stats.csv"); do head -30 $f; done
