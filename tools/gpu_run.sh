#!/bin/bash
# tools/gpu_run.sh <label> -- one GPU-box session: runs the steps given on stdin (one shell command per line, '#' comments), each under its
# own `timeout -k 10`, logging to gpurun_out/<label>/NN.log; a step that hits its time limit ends the session (nothing further touches
# the GPU), any other failure is recorded and the session goes on.  Line format:  <seconds> <command ...>
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
LABEL=${1:-session}
OUT=gpurun_out/$LABEL
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
n=0; rc_all=0
while IFS= read -r line; do
  case "$line" in ''|\#*) continue;; esac
  n=$((n + 1))
  secs=${line%% *}; cmd=${line#* }
  log=$(printf "%s/%02d.log" "$OUT" $n)
  echo "== step $n (limit ${secs}s): $cmd" | tee "$log"
  timeout -k 10 "$secs" bash -o pipefail -c "$cmd" >> "$log" 2>&1
  rc=$?
  echo "== step $n exit $rc" | tee -a "$log"
  tail -n 12 "$log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== step $n hit its time limit: session ends here"; exit 1; fi
  [ $rc -ne 0 ] && rc_all=1
done
exit $rc_all
