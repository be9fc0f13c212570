#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf /tmp/wg_prof; cd /tmp
MPPO_WGRAD_DBG=4 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/wg_prof -- python3 $GRAFT_REPO_ROOT/tools/kernel_probe.py learn 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/wg_prof/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "wgrad" in r["Kernel_Name"]]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
first=d[0::2]; second=d[1::2]
import statistics as st
print("wgrad first launch (cold operands): median %.2f us; immediate relaunch (warm): median %.2f us; n=%d"%(st.median(first[4:]), st.median(second[4:]), len(first)))
PY
