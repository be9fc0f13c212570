#!/bin/bash
# tools/soak8.sh <updates> <log> [envs] [config overrides...] - world = 8 on ONE MI355X: four processes of two ranks each (the GPU box allows six GPU processes), 512
# environments per rank (4096 in all), the peer exchange inside every rank's hipGraph, the engine's own random streams; at the end every
# rank's parameters must be bit-identical and no wait may have timed out (tests/dist_worker.py run_process_of_ranks).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
UPD=${1:-500}; LOG=${2:-gpurun_out/soak8.log}; ENVS=${3:-4096}; EXTRA="${@:4}"   # ENVS = environments in all (32768 = BASELINE configs[2]: 8 x 4096)
TMP=$(mktemp -d); PORT=$((20000 + RANDOM % 20000))
export MPPO_TEST_SOAK=1 MPPO_ALLREDUCE=peer HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
pids=()
for p in 0 1 2 3; do
  python tests/dist_worker.py procs $p 4 2 $PORT $UPD $TMP/r training.num_envs=$ENVS training.total_timesteps=2000000000 $EXTRA > $TMP/p$p.log 2>&1 &
  pids+=($!)
done
rc=0
for pid in "${pids[@]}"; do wait $pid || rc=1; done
{ echo "== tools/soak8.sh $UPD: exit $rc"; cat $TMP/p0.log | grep -v "^\[Gloo\]"; for p in 1 2 3; do grep -i "error\|timed out\|Traceback" $TMP/p$p.log; done
  python3 - $TMP <<'PY'
import sys, numpy as np
d = sys.argv[1]
r = [np.load(f"{d}/r{k}.npz") for k in range(8)]
same = all(np.array_equal(r[0]["params"], x["params"]) for x in r[1:])
print("replicas bit-identical over 8 ranks:", same, "| hipGraph on every rank:", all(bool(x["graph"]) for x in r), "| finite:", bool(np.isfinite(r[0]["params"]).all()))
sys.exit(0 if same else 1)
PY
} > "$LOG" 2>&1 || rc=1
cat "$LOG"; rm -rf $TMP; exit $rc
