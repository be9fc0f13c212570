#!/bin/bash
# Round-2 evidence run: GPU tests, bench lines (configs[1], [3], [4]), rocprofv3 kernel stats of the bench command, HBM-side traffic
# counters (FETCH_SIZE / WRITE_SIZE in separate passes).  Everything lands under gpurun_out/r2p_<tag>/ ; copy what is judged into profiles/.
TAG=${1:-a}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2p_$TAG
mkdir -p $OUT; export TMPDIR=/tmp
echo "=== pytest -m gpu"; timeout 1800 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
echo "=== bench default"; timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 600 $OUT/bench_default.json
echo "=== bench bf16 (configs[3])"; timeout 600 python bench.py --set training.mlp_dtype=bf16 --no-cpu-baseline > $OUT/bench_config3_bf16.json 2>/dev/null; python -c "import json;d=json.load(open('$OUT/bench_config3_bf16.json'));print(d['value'],d['ms_per_step'])"
echo "=== bench stompy_full 8192 (configs[4])"; timeout 600 python bench.py --config stompy_full --envs-per-gpu 8192 --no-cpu-baseline > $OUT/bench_config5_stompy_full_8192.json 2>/dev/null; python -c "import json;d=json.load(open('$OUT/bench_config5_stompy_full_8192.json'));print(d['value'],d['ms_per_step'])"
echo "=== rocprof kernel stats"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_prof.json 2>/dev/null
cd $GRAFT_REPO_ROOT; cp $(ls $OUT/prof/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv; head -12 $OUT/kernel_stats.csv | cut -c1-150
echo "=== traffic"; cd /tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/traffic/$tag -- python3 $GRAFT_REPO_ROOT/tools/kernel_probe.py all 8 > $OUT/traffic_$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - "$OUT" <<'PY'
import csv, glob, collections, json, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/traffic/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('mppo::', '')[:80]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
res = {}
for k, d in sorted(agg.items()):
    if 'FETCH_SIZE' not in d or 'WRITE_SIZE' not in d:
        continue
    fetch_raw = sum(d['FETCH_SIZE']) / len(d['FETCH_SIZE']) * 1024.0   # KiB -> bytes
    write = sum(d['WRITE_SIZE']) / len(d['WRITE_SIZE']) * 1024.0
    r32 = sum(d.get('TCC_EA0_RDREQ_32B_sum', [0])) / max(1, len(d.get('TCC_EA0_RDREQ_32B_sum', [0])))
    res[k] = {"launches": len(d['FETCH_SIZE']), "fetch_size_raw_bytes": round(fetch_raw), "fetch_bytes_corrected": round(2 * fetch_raw), "write_bytes": round(write),
              "hbm_bytes_per_launch": round(2 * fetch_raw + write), "rdreq_32B_per_launch": r32}
json.dump({"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum (separate passes, tools/gpu_r2_profiles.sh) -- python3 tools/kernel_probe.py all 8; means per launch. FETCH_SIZE / WRITE_SIZE are KiB; gfx950 correction per MI355X_MICROARCH.md: FETCH_SIZE x2 (128-byte requests tallied at 64 B; valid while TCC_EA0_RDREQ_32B = 0), WRITE_SIZE exact. Workload: stompy_pro, 4096 envs, minibatch 1280 rows.",
           "kernels": res}, open(out + '/hbm_traffic.json', 'w'), indent=1)
for k, v in res.items():
    print('%-46s n=%4d  fetch %8.2f MB  write %8.2f MB  total %8.2f MB' % (k, v['launches'], v['fetch_bytes_corrected'] / 1e6, v['write_bytes'] / 1e6, v['hbm_bytes_per_launch'] / 1e6))
PY
