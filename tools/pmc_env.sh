#!/bin/bash
# tools/pmc_env.sh <tag> - SQ counters of the environment kernel (both BASELINE robots, tools/env_time.py as the workload), one rocprofv3 --pmc
# pass per counter set (never combined with other trace domains); means per launch and per wave -> gpurun_out/<tag>/env_pmc.txt
TAG=${1:-pmc}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU"; do
  i=$((i + 1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc$i" -- python3 "$ROOT/tools/env_time.py" > "$OUT/pmc$i.log" 2>&1
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $i hit its time limit: stopping"; exit 1; fi
done
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/pmc*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'env_kernel' not in k or ', 1>' not in k:
            continue
        agg[k.split('(')[0].replace('void mppo::', '')][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/env_pmc.txt', 'w') as fh:
    for k, d in sorted(agg.items()):
        waves = sum(d['SQ_WAVES']) / len(d['SQ_WAVES']) if 'SQ_WAVES' in d else float('nan')
        fh.write('== %s  (mean over %d launches, %.0f waves per launch)\n' % (k, len(next(iter(d.values()))), waves))
        for c in sorted(d):
            m = sum(d[c]) / len(d[c])
            fh.write('   %-26s mean=%.4g  per-wave=%.1f\n' % (c, m, m / waves))
print(open(out + '/env_pmc.txt').read())
PY
