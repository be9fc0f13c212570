#!/bin/bash
# kernel-argument preload (-mllvm -amdgpu-kernarg-preload-count=16) on k_ppo.hip: adam_kernel's in-situ duration, product vs variant (rocprofv3 kernel stats)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
LIBS="${@:-product tools/_variants/libminppo_kpre.so}"
for lib in $LIBS; do
  tag=$(basename $lib .so)
  cd /tmp
  if [ "$lib" = "product" ]; then timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_$tag.json 2>/dev/null
  else timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/tools/bench_with_lib.py $GRAFT_REPO_ROOT/$lib --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_$tag.json 2>/dev/null; fi
  cd $GRAFT_REPO_ROOT
  echo "=== $tag: $(python3 -c "import json;d=json.loads(open('gpurun_out/bench_$tag.json').read().strip().splitlines()[-1]);print('%.3f M  %.3f ms  rowpass %.2f us'%(d['value']/1e6,d['ms_per_step'],d['roofline']['us_per_launch']))")"
  python3 - $tag <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/prof_%s/*/*kernel_stats.csv" % sys.argv[1])[0]
for r in list(csv.DictReader(open(f)))[:7]:
    print("  %-60s calls %5s avg %8.2f us min %8.2f" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
