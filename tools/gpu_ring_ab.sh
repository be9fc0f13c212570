#!/bin/bash
# ring depth of the row pass's weight stream (k_fused.hip GemmPipe): product (MPPO_RING_TRAIN = 5) against variants 3 and 4; bench lines, twice
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/ring_ab.txt; : > $OUT
timeout 900 python -m pytest tests/test_kernels_ppo.py tests/test_engine.py tests/test_golden.py -q -m gpu -x > gpurun_out/pytest_gpu_ppo.log 2>&1; tail -2 gpurun_out/pytest_gpu_ppo.log
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
for rep in 1 2; do
run "f32 ring 5 (product)" ""
run "f32 ring 4" tools/_variants/libminppo_ring4.so
run "f32 ring 3" tools/_variants/libminppo_ring3.so
run "bf16 ring 5 (product)" "" --set training.mlp_dtype=bf16
run "bf16 ring 3" tools/_variants/libminppo_ring3.so --set training.mlp_dtype=bf16
run "stompy_full ring 5 (product)" "" --config stompy_full --envs-per-gpu 8192
run "stompy_full ring 3" tools/_variants/libminppo_ring3.so --config stompy_full --envs-per-gpu 8192
done
