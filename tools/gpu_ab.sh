#!/bin/bash
# A/B of env-var-selected variants: usage  gpu_ab.sh "<probe args>" "VAR=a VAR2=b" "VAR=c" ...   (each config = one rocprofv3 --stats run)
export TMPDIR=/tmp
cd /tmp
PROBE="$1"; shift
i=0
for cfg in "$@"; do
  i=$((i+1))
  OUT=$GRAFT_REPO_ROOT/gpurun_out/ab/$i
  mkdir -p $OUT
  ( for kv in $cfg; do export "$kv"; done
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/kernel_probe.py $PROBE > $OUT/log.txt 2>&1 )
  echo "== [$i] $cfg"
  python3 - "$OUT" <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        n=r['Name']
        if 'mppo' in n and float(r['Percentage'])>0.5: print('   %-40s calls=%-5s avg=%8.2f us  %5s%%'%(n.split('(')[0].replace('void ','').replace('mppo::','')[:40], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
done
