#!/bin/bash
# tools/pmc_cache.sh <tag> [workload args] - L1 (TCP) / L2 (TCC) counters of the PPO update's kernels with tools/kernel_probe.py as the workload:
# bytes a CU takes in from L2 per launch (the weight stream of the row pass, the operand bands of the weight-gradient launch) and L2 hit rates.
# One rocprofv3 --pmc pass per counter set, never combined with other trace domains, the program directly after `--`.
TAG=${1:-pmc_cache}
shift
ARGS=${*:-"learn 4"}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" \
           "TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN2_sum"; do
  i=$((i + 1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc$i" -- python3 "$ROOT/tools/kernel_probe.py" $ARGS > "$OUT/pmc$i.log" 2>&1
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $i hit its time limit: stopping"; exit 1; fi
  echo "pass $i exit $rc"
done
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(out + '/pmc*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'mppo::' not in k:
            continue
        k = k.split('(')[0].replace('void mppo::', '')
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob(out + '/pmc1/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void mppo::', '')
        dur[k].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
with open(out + '/cache_pmc.txt', 'w') as fh:
    fh.write('rocprofv3 --kernel-trace --pmc <set> -- python3 tools/kernel_probe.py (one pass per counter set; means per launch)\n')
    for k, d in sorted(agg.items()):
        n = len(next(iter(d.values())))
        us = (sum(dur[k]) / len(dur[k]) / 1e3) if dur.get(k) else float('nan')
        fh.write('== %s  (mean over %d launches, %.2f us per launch under the counter pass)\n' % (k, n, us))
        for c in sorted(d):
            fh.write('   %-34s mean=%.5g\n' % (c, sum(d[c]) / len(d[c])))
print(open(out + '/cache_pmc.txt').read()[:6000])
PY
