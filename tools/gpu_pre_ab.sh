#!/bin/bash
# pre-gathered rows A/B: GPU tests, then bench lines with and without (MPPO_NO_PREGATHER=1), twice, interleaved; bf16 and stompy_full too
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/pre_ab.txt; : > $OUT
timeout 1200 python -m pytest tests -q -m gpu -x > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
grep -q " passed" gpurun_out/pytest_gpu.log || exit 1
grep -q "failed" gpurun_out/pytest_gpu.log && exit 1
run() {
  local tag="$1"; shift
  local line
  line=$(env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline ${EXTRA} 2>/dev/null | tail -1)
  python3 - "$tag" "$line" >> $OUT <<'PY'
import json, sys
tag, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    print("%-44s %8.3f M env-steps/s  %7.3f ms/update  row pass %6.2f us  frac %.3f" % (tag, d["value"] / 1e6, d["ms_per_step"], d["roofline"].get("us_per_launch", -1), d["roofline"]["frac"]))
except Exception as e:
    print("%-44s FAILED (%s) %s" % (tag, e, line[:200]))
PY
  tail -1 $OUT
}
for rep in 1 2; do
EXTRA=""
run "f32 pre-gather" MPPO_AB=0
run "f32 no pre-gather" MPPO_NO_PREGATHER=1
EXTRA="--set training.mlp_dtype=bf16"
run "bf16 pre-gather" MPPO_AB=0
run "bf16 no pre-gather" MPPO_NO_PREGATHER=1
EXTRA="--config stompy_full --envs-per-gpu 8192"
run "stompy_full pre-gather" MPPO_AB=0
run "stompy_full no pre-gather" MPPO_NO_PREGATHER=1
done
