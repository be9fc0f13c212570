#!/bin/bash
# Adam A/B: product library vs a variant (tools/_variants/libminppo_$1.so): adam_kernel average from rocprofv3 of the bench command
cd /tmp; export TMPDIR=/tmp
for lib in minppo_amd/libminppo_hip.so tools/_variants/libminppo_$1.so; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/adbg -- python3 $GRAFT_REPO_ROOT/tools/bench_with_lib.py $GRAFT_REPO_ROOT/$lib --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  echo "== $lib"; grep -h "adam_kernel\|fused_mlp_kernel<false, false" $(ls -t $GRAFT_REPO_ROOT/gpurun_out/adbg/*/*kernel_stats.csv | head -1) | awk -F, '{print substr($1,1,40), $(NF-6), $(NF-4)}'
done
