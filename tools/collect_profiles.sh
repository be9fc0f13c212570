#!/bin/bash
# tools/collect_profiles.sh <session label, e.g. r5_z> <profiles prefix, e.g. r05_z> - copies the summaries of a `tools/profiles.sh` + two
# `tools/pmc_mlp.sh` sessions (gpurun_out/<label>, <label>_pmc, <label>_pmc_bf16) into profiles/ under the round's naming, sidecars included.
# A third argument "partial" skips what is not there yet: tools/refresh_evidence.sh runs it ON THE GPU BOX between the profiler passes and the bench
# lines, so that the bench lines committed with a set of summaries quote those summaries (bench.py quotes summaries of its own tree only).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
L=$1; P=profiles/$2; PARTIAL=$3
for f in bench_default.json bench_config3_bf16.json bench_config5_stompy_full_8192.json hbm_traffic.json hbm_traffic.json.meta.json hbm_traffic_bf16.json \
         hbm_traffic_bf16.json.meta.json kernel_stats.csv kernel_stats.csv.meta.json kernel_stats_config3_bf16.csv kernel_stats_config3_bf16.csv.meta.json pytest_gpu.log; do
  [ -n "$PARTIAL" ] && [ ! -f "gpurun_out/$L/$f" ] && continue
  cp "gpurun_out/$L/$f" "${P}_$f" || exit 1
done
for f in mlp_pmc.json mlp_pmc.json.meta.json mlp_pmc.txt mlp_pmc.txt.meta.json; do
  cp "gpurun_out/${L}_pmc/$f" "${P}_$f" || exit 1
  cp "gpurun_out/${L}_pmc_bf16/$f" "${P}_${f/mlp_pmc/mlp_pmc_bf16}" || exit 1
done
ls profiles | grep -c "^$2_"
