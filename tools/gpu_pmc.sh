#!/bin/bash
# PMC passes over the two hot kernels (separate rocprofv3 runs, --pmc only together with --kernel-trace)
export TMPDIR=/tmp
cd /tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -o -E "\bSQ_[A-Z0-9_]+\b" | sort -u | tr '\n' ' ' > $OUT/sq_counters.txt
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_WAVE32_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python3 $GRAFT_REPO_ROOT/tools/kernel_probe.py learn 1 > $OUT/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections, os
root='gpurun_out/pmc'
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root+'/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:60]
        if 'fused_mlp' in k or 'gemm_tn' in k or 'grad_reduce' in k:
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in agg.items():
    print('==',k)
    for c,v in sorted(d.items()): print('   %-28s n=%d mean=%.4g'%(c,len(v),sum(v)/len(v)))
PY
