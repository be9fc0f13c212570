#!/bin/bash
# tools/refresh_evidence.sh <new label, e.g. r5_ac> <new profiles prefix, e.g. r05_ac> <old profiles prefix, e.g. r05_ab>
# One GPU session (tests + in-situ kernel stats + HBM traffic + SQ counters, float and bf16, then the three bench lines - last, so that they quote this session's summaries) on the CURRENT tree, the
# summaries copied into profiles/ under the new prefix, the old prefix's files removed and its name replaced in DESIGN.md / README.md /
# profiles/README.md.  The NUMBERS quoted in those files are not touched: read the printed summary and edit them.  Needs a clean, built tree.
cd "$(dirname "$0")/.." || exit 1
L=$1; NEW=$2; OLD=$3
[ -n "$L" ] && [ -n "$NEW" ] && [ -n "$OLD" ] || { echo "usage: $0 <label> <new prefix> <old prefix>"; exit 2; }
printf '1500 bash tools/profiles.sh %s "tests stats bf16stats traffic trafficbf16"\n900 bash tools/pmc_mlp.sh %s_pmc all 4\n900 bash tools/pmc_mlp.sh %s_pmc_bf16 all 4 training.mlp_dtype=bf16\n60 bash tools/collect_profiles.sh %s %s partial\n1500 bash tools/profiles.sh %s bench\n' "$L" "$L" "$L" "$L" "$NEW" "$L" > tools/steps/$L.txt
tools/gpurun.sh --timeout 1200 -- "bash tools/gpu_run.sh $L < tools/steps/$L.txt" > /tmp/gpurun_$L.log 2>&1 || { tail -5 /tmp/gpurun_$L.log; exit 1; }
grep "gpurun\] status\|left this round" /tmp/gpurun_$L.log
tail -n 2 gpurun_out/$L/pytest_gpu.log
git rm -q profiles/${OLD}_* 2>/dev/null
tools/collect_profiles.sh "$L" "$NEW" || exit 1
sed -i "s/${OLD}_/${NEW}_/g; s/tools\/profiles.sh ${OLD/r0/r}/tools\/profiles.sh $L/g; s/tools\/pmc_mlp.sh ${OLD/r0/r}_pmc/tools\/pmc_mlp.sh ${L}_pmc/g" DESIGN.md README.md profiles/README.md
python3 - "$NEW" <<'PY'
import csv, json, sys
sys.path.insert(0, ".")
import bench
n = sys.argv[1]
print("kernel sources", bench.kernel_sources_sha(), "summaries", json.load(open(f"profiles/{n}_kernel_stats.csv.meta.json"))["kernel_sources_sha"])
for f in ("bench_default", "bench_config3_bf16", "bench_config5_stompy_full_8192"):
    d = json.loads(open(f"profiles/{n}_{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"]), round(d["ms_per_step"], 3), (d.get("roofline") or {}).get("frac"), (d.get("cpu_baseline") or {}).get("value"))
for f in ("kernel_stats_config3_bf16", "kernel_stats"):
    for r in list(csv.DictReader(open(f"profiles/{n}_{f}.csv")))[:7]:
        if "at::native" not in r["Name"]:
            print("  %-70s %6s %9.2f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
for f in ("mlp_pmc", "mlp_pmc_bf16"):
    d = json.load(open(f"profiles/{n}_{f}.json"))["kernels"]
    print(f, {k[:28]: round(v["mfma_busy_frac"], 3) for k, v in d.items() if v["mfma_busy_frac"] > 0})
PY
