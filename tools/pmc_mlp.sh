#!/bin/bash
# tools/pmc_mlp.sh <tag> [workload args] - SQ counters of the PPO update's kernels (fused_mlp_kernel training + rollout, wgrad_kernel, adam_kernel)
# with tools/kernel_probe.py as the workload (one rollout + one learn phase, eager launches), one rocprofv3 --pmc pass per counter set (8 SQ
# slots per pass; never combined with other trace domains; the program goes directly after `--`).  Means per launch and per wave, and the
# matrix-core utilisation MFMA-busy cycles / (4 SIMDs x CUs x kernel cycles) -> gpurun_out/<tag>/mlp_pmc.txt + mlp_pmc.json
TAG=${1:-pmc_mlp}
shift
ARGS=${*:-"all 4"}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i + 1))
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc$i" -- python3 "$ROOT/tools/kernel_probe.py" $ARGS > "$OUT/pmc$i.log" 2>&1
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $i hit its time limit: stopping"; exit 1; fi
  echo "pass $i exit $rc"
done
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, collections, json, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(out + '/pmc*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'mppo::' not in k:
            continue
        k = k.split('(')[0].replace('void mppo::', '')
        if not any(s in k for s in ('fused_mlp_kernel', 'bf16_rowpass_kernel', 'wgrad_kernel', 'adam_kernel', 'env_kernel', 'gae_kernel')):
            continue
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob(out + '/pmc1/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void mppo::', '')
        dur[k].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
CUS, SIMDS = 256, 4
res = {}
with open(out + '/mlp_pmc.txt', 'w') as fh:
    fh.write('rocprofv3 --kernel-trace --pmc <set> -- python3 tools/kernel_probe.py (separate passes per counter set; SQ_*_CYCLES / WAIT / ACTIVE are quad-cycles summed over waves,\n'
             'SQ_VALU_MFMA_BUSY_CYCLES = cycles summed over the matrix pipes of all SIMDs, SQ_BUSY_CYCLES = cycles summed over the 32 shader engines)\n')
    for k, d in sorted(agg.items()):
        mean = {c: sum(v) / len(v) for c, v in d.items()}
        waves = mean.get('SQ_WAVES', float('nan'))
        n = len(next(iter(d.values())))
        us = (sum(dur[k]) / len(dur[k]) / 1e3) if dur.get(k) else float('nan')
        fh.write('== %s  (mean over %d launches, %.0f waves per launch, %.2f us per launch under the counter pass)\n' % (k, n, waves, us))
        for c in sorted(mean):
            fh.write('   %-30s mean=%.4g  per-wave=%.1f\n' % (c, mean[c], mean[c] / waves))
        busy = mean.get('SQ_VALU_MFMA_BUSY_CYCLES')
        sqb = mean.get('SQ_BUSY_CYCLES')
        entry = {'launches': n, 'waves': waves, 'us_under_pmc': us}
        if busy is not None and us == us:
            # one MFMA pipe per SIMD, 1024 on the chip.  (a) against the launch's duration at the 2.4 GHz the 157.3 TFLOP/s peak is priced at;
            # (b) against the cycles the shader engines were busy (SQ_BUSY_CYCLES is summed over the 32 shader engines), which leaves the
            # launch ramp out.  GRBM_GUI_ACTIVE brackets the counter start / stop as well and is not used.
            entry['mfma_busy_frac'] = busy / (SIMDS * CUS * us * 1e-6 * 2.4e9)
            fh.write('   -> MFMA busy / (1024 SIMD pipes x launch duration x 2.4 GHz) = %.4f\n' % entry['mfma_busy_frac'])
            if sqb:
                entry['mfma_busy_over_sq_busy'] = busy / (SIMDS * CUS * sqb / 32.0)
                fh.write('   -> MFMA busy / (1024 SIMD pipes x SQ_BUSY_CYCLES / 32 shader engines) = %.4f\n' % entry['mfma_busy_over_sq_busy'])
        wc = mean.get('SQ_WAVE_CYCLES')
        if wc:
            for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_LDS'):
                if c in mean:
                    entry[c + '_over_wave_cycles'] = mean[c] / wc
            fh.write('   -> of a wave\'s life: ' + ', '.join('%s %.1f %%' % (c.replace('SQ_', ''), 100 * mean[c] / wc) for c in
                     ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS') if c in mean) + '\n')
        entry['counters'] = mean
        res[k] = entry
json.dump({'_how': 'tools/pmc_mlp.sh: rocprofv3 --kernel-trace --pmc, three separate passes over tools/kernel_probe.py; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMD pipes x launch duration under the counter pass x 2.4 GHz)',
           'kernels': res}, open(out + '/mlp_pmc.json', 'w'), indent=1)
print(open(out + '/mlp_pmc.txt').read())
PY
python3 tools/profile_meta.py "$OUT/mlp_pmc.json" "$OUT/mlp_pmc.txt" --command "tools/pmc_mlp.sh $TAG $ARGS"
