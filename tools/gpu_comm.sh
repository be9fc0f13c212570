#!/bin/bash
# RCCL path on one GPU (single-rank communicator) + a bench run through it (no hipGraph, eager launches as at N>1)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_engine.py -m gpu -x -q -k rccl 2>&1 | tail -5 > gpurun_out/comm_test.log
MPPO_FORCE_COMM=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_forcecomm.json 2> gpurun_out/bench_forcecomm.err
# and the launcher form the driver uses for N>1, at N=1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_torchrun1.json 2> gpurun_out/bench_torchrun1.err
tail -3 gpurun_out/comm_test.log; tail -c 600 gpurun_out/bench_forcecomm.json; tail -3 gpurun_out/bench_forcecomm.err; tail -c 400 gpurun_out/bench_torchrun1.json; tail -3 gpurun_out/bench_torchrun1.err
