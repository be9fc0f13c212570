#!/bin/bash
# where does wgrad_kernel's time go: rocprofv3 average per launch for a set of MPPO_WGRAD_DBG / MPPO_KSPLIT settings
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for cfg in "0 8"; do
  set -- $cfg
  rm -rf /tmp/wg_prof; cd /tmp
  MPPO_WGRAD_DBG=$1 MPPO_KSPLIT=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wg_prof -- python3 $GRAFT_REPO_ROOT/tools/kernel_probe.py learn 1 > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  python3 - "$1" "$2" <<PY
import csv,glob,sys
f=glob.glob("/tmp/wg_prof/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "wgrad" in r["Name"] or "grad_reduce" in r["Name"] or "fused_mlp_kernel<false, false" in r["Name"] or "adam" in r["Name"]:
        print("dbg=%s ksplit=%s  %-44s calls %5s avg %8.2f us"%(sys.argv[1], sys.argv[2], r["Name"].split("(")[0][:44], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
