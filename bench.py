#!/usr/bin/env python3
"""bench.py — env-steps/sec of the full PPO training loop (rollout + update) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one `_update_step` of the reference (minppo/train.py:146-283): T = 10 policy + physics
steps for every environment, the bootstrap value, GAE and E x M = 4 x 32 minibatch optimizer steps.
Workload: BASELINE.json configs[1] — the stompy_pro stand-in (synth_stompy_pro: O = 225, A = 10), 4096
environments PER GPU (weak scaling: configs[2] is 8 x 4096), fp32.  Inputs are synthetic and resident in
HBM (the environment state itself); weights are random-init.  Rank 0 prints ONE JSON line.

The line also carries
  roofline     : the dominant kernel (the f32-MFMA batched GEMM of the hidden layers) timed live with HIP
                 events on the engine's stream, algorithmic FLOPs per launch / average duration against the
                 dense f32 MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md)
  cpu_baseline : the NumPy oracle (oracle/, a "port" of the reference's algorithm; the JAX reference
                 cannot run here) timed on this box's host cores on a bounded sample of the same workload.
"""

from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: "Peak FP32 (matrix)"


def parse() -> argparse.Namespace:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--config", default="stompy_pro")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-envs", type=int, default=4096)
    return ap.parse_args()


def gemm_probe(tr, reps: int = 200):
    """Average duration of the dominant kernel: the hidden-layer forward GEMM of one minibatch
    (both networks in one launch: 2 x [mb,256] = [mb,256] . [256,256], bias + activation), HIP events on
    the engine stream.  Returns (seconds per launch, algorithmic flops per launch, description)."""
    import torch
    from minppo_amd import _native as nat

    mb, H = tr.minibatch_size // tr.world_size, tr.H
    dev = tr.device
    a = torch.randn(2, mb, H, device=dev)
    w = torch.randn(2, H, H, device=dev) * 0.06
    b = torch.zeros(2, H, device=dev)
    c = torch.empty(2, mb, H, device=dev)
    descs = (nat.GemmDesc * 2)()
    for i in range(2):
        descs[i] = nat.GemmDesc(a[i].data_ptr(), w[i].data_ptr(), c[i].data_ptr(), b[i].data_ptr(), 0, 0, 0, mb, H, H, H, H, H, 0, 1 if i == 0 else 2)
    s = tr.stream
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(20):
        tr.lib.gemm_batch(descs, 2, 0, 1, 0, 0, s.cuda_stream)
    s.synchronize()
    ev0.record(s)
    for _ in range(reps):
        tr.lib.gemm_batch(descs, 2, 0, 1, 0, 0, s.cuda_stream)
    ev1.record(s)
    s.synchronize()
    sec = ev0.elapsed_time(ev1) * 1e-3 / reps
    flops = 2.0 * 2 * mb * H * H
    return sec, flops, f"gemm_kernel<fwd> 2x[{mb},{H}]x[{H},{H}] f32 MFMA"


def cpu_baseline(config_name: str, n_envs: int):
    """Times one update of the NumPy oracle (oracle/env_oracle.update_step) on the host."""
    import numpy as np
    from minppo_amd.config import load_config_from_cli
    from minppo_amd.model import load_model
    from oracle import ppo_oracle as po
    from oracle.env_oracle import EnvOracle, default_hp, update_step

    cfg = load_config_from_cli([config_name, f"training.num_envs={n_envs}"])
    cm = load_model(cfg.environment.model or cfg.kscale_id)
    env = EnvOracle(cm.t, dtype=np.float32)
    N, T, A, H, M, E = n_envs, cfg.training.num_steps, cm.nu, cfg.model.hidden_size, cfg.training.num_minibatches, cfg.training.update_epochs
    O = env.observation_size
    rng = np.random.default_rng(0)
    p = po.named_to_flat(po.init_params(cfg.training.seed, O, A, H, np.float32), O, A, H)
    opt = po.OptState(np.zeros_like(p), np.zeros_like(p), 0)
    es = env.reset(N)
    hp = default_hp(cfg)
    noise = rng.standard_normal((T, N, A)).astype(np.float32)
    perms = np.stack([rng.permutation(N * T) for _ in range(E)])
    t0 = time.perf_counter()
    update_step(env, p, opt, es, es["obs"], noise, perms, O=O, A=A, H=H, num_minibatches=M, hp=hp)
    dt = time.perf_counter() - t0
    try:
        from threadpoolctl import threadpool_info

        threads = max([i.get("num_threads", 1) for i in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    return {"value": N * T / dt, "unit": "env-steps/s", "cores": int(threads), "kind": "port",
            "sample": f"1 update of the NumPy float32 oracle: {N} envs x {T} steps + {E}x{M} minibatches ({dt:.1f} s); BLAS threads = cores, element-wise NumPy is single-threaded"}


def main() -> None:
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU); see the docstring")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from minppo_amd.config import load_config_from_cli
    from minppo_amd.train import Trainer

    n_global = args.envs_per_gpu * world
    cfg = load_config_from_cli([args.config, f"training.num_envs={n_global}"])
    tr = Trainer(cfg, device=f"cuda:{local_rank}", rank=rank, world_size=world, use_graph=not args.no_graph)
    tr.init_comm()
    tr.reset()
    for _ in range(args.warmup):
        tr.update()

    def fence():
        tr.stream.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.update()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    steps_total = tr.T * n_global * args.steps
    stats = tr.rollout_stats()
    lossm = tr.losses().reshape(-1, 4).mean(0)

    out = None
    if rank == 0:
        sec, flops, desc = gemm_probe(tr)
        achieved = flops / sec / 1e12
        out = {
            "metric": "env-steps/sec (whole node), stompy_pro 4096 envs, 1/2/4/8 MI355X",
            "value": steps_total / dt,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (stand-in robot synth_stompy_pro, random-init weights, Philox action noise)",
            "config": {"workload": f"{args.config}: {args.envs_per_gpu} envs/GPU x T={tr.T} rollout + {tr.E}x{tr.M} minibatch PPO update, O={tr.O} A={tr.A} H={tr.H}, fp32 (BASELINE configs[1]; configs[2] at 8 GPUs)",
                       "global_envs": n_global, "parallelism": f"env-sharded dp{world}, RCCL gradient all-reduce per optimizer step" if world > 1 else "single GPU",
                       "hipgraph": bool(not args.no_graph and world == 1)},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                         "traffic": None, "kernel": desc, "us_per_launch": sec * 1e6,
                         "whole_update_mlp_tflops": (26.0 * (2 * tr.O * tr.H + 2 * tr.H * tr.H + tr.H * (tr.A + 1)) + 0.2 * (tr.O * tr.H + tr.H * tr.H + tr.H)) * steps_total / world / dt / 1e12},
            "sanity": {"mean_reward": stats["mean_reward"], "done_fraction": stats["done_fraction"], "mean_total_loss": float(lossm[0])},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.config, args.cpu_baseline_envs)
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
    tr.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
