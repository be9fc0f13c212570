#!/bin/bash
# product library against a variant build (tools/_variants/libminppo_<name>.so): 40-update bench lines, three times each, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
VAR=${1:-prev}; shift
OUT=gpurun_out/ab_$VAR.txt; : > $OUT
run() {
  local tag="$1"; local lib="$2"; shift 2
  local line
  if [ -z "$lib" ]; then line=$(timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | tail -1)
  else line=$(timeout 300 python tools/bench_with_lib.py $lib --steps 40 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | tail -1); fi
  python3 - "$tag" "$line" >> $OUT <<'PY'
import json, sys
tag, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    print("%-34s %8.3f M env-steps/s  %7.3f ms/update  row pass %6.2f us  frac %.3f" % (tag, d["value"] / 1e6, d["ms_per_step"], d["roofline"].get("us_per_launch", -1), d["roofline"]["frac"]))
except Exception as e:
    print("%-34s FAILED (%s) %s" % (tag, e, line[:200]))
PY
  tail -1 $OUT
}
for rep in 1 2 3; do
run "product" "" "$@"
run "variant $VAR" tools/_variants/libminppo_$VAR.so "$@"
done
