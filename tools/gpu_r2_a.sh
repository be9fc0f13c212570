#!/bin/bash
# round 2, first GPU pass: GPU tests (tightened physics bounds, stompy_full, graph+RCCL), smoke, bench (incl. forced single-rank communicator in the graph),
# the bare `--gpus 2` invocation on a 1-GPU box (must be a clean error), rocprof kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "=== pytest -m gpu"; timeout 1800 python -m pytest tests -q -m gpu -x > gpurun_out/r2a_pytest_gpu.log 2>&1; tail -15 gpurun_out/r2a_pytest_gpu.log
echo "=== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "=== bench default"; timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err; tail -c 1500 gpurun_out/r2a_bench.json
echo "=== bench, single-rank communicator inside the graph"; MPPO_FORCE_COMM=1 timeout 900 python bench.py --no-cpu-baseline --steps 10 > gpurun_out/r2a_bench_comm_graph.json 2> gpurun_out/r2a_bench_comm_graph.err; tail -c 700 gpurun_out/r2a_bench_comm_graph.json; tail -3 gpurun_out/r2a_bench_comm_graph.err
echo "=== bench, single-rank communicator, eager"; MPPO_FORCE_COMM=1 MPPO_GRAPH_COMM=0 timeout 900 python bench.py --no-cpu-baseline --steps 10 > gpurun_out/r2a_bench_comm_eager.json 2> gpurun_out/r2a_bench_comm_eager.err; tail -c 700 gpurun_out/r2a_bench_comm_eager.json
echo "=== bare --gpus 2 on this box"; timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1; echo "rc=$?"
echo "=== torchrun form at N=1"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -c 400
echo "=== rocprof"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2a_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2a_bench_prof.json 2>/dev/null
cd $GRAFT_REPO_ROOT; python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/r2a_prof/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print("%-58s calls %5s avg %8.1f us  total %7.2f ms"%(r["Name"].split("(")[0][:58], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
