#!/bin/bash
# tools/profiles.sh <tag> [parts] - the evidence run of a round on the GPU box:  gpurun -- 'bash tools/profiles.sh r03_a'
# parts (default "tests bench stats traffic"): tests = pytest -m gpu | bench = the three BASELINE bench lines (configs[1], [3], [4]) |
# stats = rocprofv3 --kernel-trace --stats of the default bench command with --no-probe | traffic (trafficbf16: the bf16 network) = FETCH_SIZE / WRITE_SIZE / RDREQ in SEPARATE
# --pmc passes over tools/kernel_probe.py (never combined with other trace domains) | bf16stats = kernel stats of the bf16 bench.
# Everything lands under gpurun_out/<tag>/ ; copy what is judged into profiles/ (named <tag>_*).
TAG=${1:-r00}
PARTS=${2:-"tests bench stats traffic"}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
step() { local secs=$1; shift; timeout -k 10 "$secs" "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== a step hit its time limit: stopping"; exit 1; fi; return $rc; }
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
if has tests; then echo "=== pytest -m gpu"; step 1500 python -m pytest tests -q -m gpu > "$OUT/pytest_gpu.log" 2>&1; tail -3 "$OUT/pytest_gpu.log"; fi
if has bench; then
  echo "=== bench default (configs[1])"; step 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; tail -c 700 "$OUT/bench_default.json"
  echo "=== bench bf16 (configs[3])"; step 600 python bench.py --set training.mlp_dtype=bf16 --no-cpu-baseline > "$OUT/bench_config3_bf16.json" 2>/dev/null
  echo "=== bench stompy_full 8192 (configs[4])"; step 600 python bench.py --config stompy_full --envs-per-gpu 8192 --no-cpu-baseline > "$OUT/bench_config5_stompy_full_8192.json" 2>/dev/null
  for f in bench_default bench_config3_bf16 bench_config5_stompy_full_8192; do python3 -c "import json;d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]);print('$f', round(d['value']/1e6,3),'M', round(d['ms_per_step'],3),'ms', round(d['roofline']['us_per_launch'],2),'us', round(d['roofline']['frac'],4))"; done
fi
if has stats; then
  echo "=== rocprof kernel stats"; cd /tmp && step 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-probe > "$OUT/bench_prof.json" 2>/dev/null
  cd "$ROOT"; cp "$(ls $OUT/prof/*/*kernel_stats.csv | head -1)" "$OUT/kernel_stats.csv"; head -12 "$OUT/kernel_stats.csv" | cut -c1-150
  python3 tools/profile_meta.py "$OUT/kernel_stats.csv" --command "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-probe"
fi
if has bf16stats; then
  cd /tmp && step 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_bf16" -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-probe --set training.mlp_dtype=bf16 > "$OUT/bench_prof_bf16.json" 2>/dev/null
  cd "$ROOT"; cp "$(ls $OUT/prof_bf16/*/*kernel_stats.csv | head -1)" "$OUT/kernel_stats_config3_bf16.csv"; head -8 "$OUT/kernel_stats_config3_bf16.csv" | cut -c1-150
  python3 tools/profile_meta.py "$OUT/kernel_stats_config3_bf16.csv" --command "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-probe --set training.mlp_dtype=bf16"
fi
for TV in traffic trafficbf16; do if has $TV; then
  if [ $TV = trafficbf16 ]; then SUF=_bf16; XARG="training.mlp_dtype=bf16"; else SUF=; XARG=; fi
  echo "=== $TV"; cd /tmp
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    tag=$(echo $set | cut -d' ' -f1)
    step 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/traffic$SUF/$tag" -- python3 "$ROOT/tools/kernel_probe.py" all 8 $XARG > "$OUT/traffic${SUF}_$tag.log" 2>&1
  done
  cd "$ROOT"
  python3 - "$OUT" "$SUF" <<'PY'
import csv, glob, collections, json, sys
out = sys.argv[1]; suf = sys.argv[2] if len(sys.argv) > 2 else ''
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/traffic' + suf + '/*/*/*counter_collection.csv'):
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
json.dump({"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum (separate passes, tools/profiles.sh) -- python3 tools/kernel_probe.py all 8; means per launch. FETCH_SIZE / WRITE_SIZE are KiB; gfx950 correction per MI355X_MICROARCH.md: FETCH_SIZE x2 (128-byte requests tallied at 64 B; valid while TCC_EA0_RDREQ_32B = 0), WRITE_SIZE exact. Workload: stompy_pro, 4096 envs, minibatch 1280 rows" + (", training.mlp_dtype=bf16" if suf else "") + ".",
           "kernels": res}, open(out + '/hbm_traffic' + suf + '.json', 'w'), indent=1)
for k, v in res.items():
    print('%-46s n=%4d  fetch %8.2f MB  write %8.2f MB  total %8.2f MB' % (k, v['launches'], v['fetch_bytes_corrected'] / 1e6, v['write_bytes'] / 1e6, v['hbm_bytes_per_launch'] / 1e6))
PY
  python3 tools/profile_meta.py "$OUT/hbm_traffic$SUF.json" --command "tools/profiles.sh $TAG $TV"
fi; done
