#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_ppo.py tests/test_engine.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --set training.mlp_dtype=bf16 > gpurun_out/bench_bf16.json 2> gpurun_out/bench_bf16.err; tail -2 gpurun_out/bench_bf16.err; cat gpurun_out/bench_bf16.json
