#!/bin/bash
# HBM-side traffic of the hot kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (TCC slots), --kernel-trace only.
export TMPDIR=/tmp
cd /tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic
mkdir -p $OUT
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python3 $GRAFT_REPO_ROOT/tools/kernel_probe.py learn 1 > $OUT/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections, json
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/traffic/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:80]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
out={}
for k,d in agg.items():
    out[k]={c:{"n":len(v),"mean":sum(v)/len(v)} for c,v in d.items()}
    print('==',k, {c:round(x["mean"],2) for c,x in out[k].items()})
json.dump(out, open('gpurun_out/traffic/summary.json','w'), indent=1)
PY
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_stdout_only.json 2> gpurun_out/bench_stderr.log; wc -l gpurun_out/bench_stdout_only.json
