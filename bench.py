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
  roofline     : the dominant kernel (`fused_mlp_kernel`, the row-local forward + backward of one minibatch on the
                 f32 matrix cores) timed live with HIP events on a hipGraph replay of 50 launches, algorithmic
                 FLOPs per launch / average duration against the dense f32 MFMA peak (157.3 TFLOP/s,
                 MI355X_MICROARCH.md); `in_situ`: the same kernel inside the update, from the newest committed
                 `rocprofv3 --kernel-trace --stats` summary of this command (profiles/)
  cpu_baseline : the CPU restatement SURVEY.md 8(d) specifies (a "port": the JAX reference cannot run here) timed on this
                 box's host cores on a bounded sample of the same workload (whole updates at 4096 envs): the environment
                 step by a C++17 / OpenMP float32 twin, one environment per thread on all cores (oracle/cpu_twin/), the
                 policy and the PPO update by torch-CPU float32 with BLAS threads; `cores` = the threads actually used.

The timed region contains no episode resets: with the reference's `height_min_z = -0.2` (config.py:38) a fallen robot is still
"healthy", and no episode of the stand-in robot ends (`sanity.done_fraction` 0.0; profiles/r03_f_training_run_1B.log: none in 10^9
steps).  That is the reference's behaviour, not a shortcut; the auto-reset path is covered by the injected-termination tests
(tests/test_kernels_physics.py, tests/test_golden.py) and costs nothing extra by construction (the reset record is loaded only for
an environment that ended).
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
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: dense bf16 MFMA ~2.5 PFLOP/s (no sparsity)


def parse() -> argparse.Namespace:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--config", default="stompy_pro")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE", help="config override (dot-list), e.g. training.mlp_dtype=bf16")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-envs", type=int, default=4096)
    return ap.parse_args()


def rowpass_probe(tr, launches: int = 64, replays: int = 4):
    """Average duration of the dominant kernel, `fused_mlp_kernel` (row-local forward + backward of one minibatch:
    both hidden layers, heads, loss terms, dZ2, dZ1 of actor and critic), on the engine's own buffers.  `launches`
    back-to-back launches are captured into a hipGraph on the engine stream and replayed, bracketed by HIP events recorded
    on that stream, so the host is not in the loop.  Returns (seconds per launch, algorithmic flops per launch, text)."""
    import torch
    from minppo_amd import _native as nat

    T, N, E, M = tr.T, tr.N, tr.E, tr.M
    mb = T * N // M
    reg = {k: tr.region(k) for k in ("params", "obs", "action", "value", "log_prob", "adv", "target", "perm", "adv_stats", "grad_ws")}
    batch = nat.Batch(reg["obs"].data_ptr(), tr.OP, reg["action"].data_ptr(), tr.A, reg["value"].data_ptr(), reg["log_prob"].data_ptr(),
                      reg["adv"].data_ptr(), reg["target"].data_ptr())
    lc = tr.ecfg.loss
    wsb = tr.lib.grad_ws_bytes(C.byref(tr.net), mb)
    s = tr.stream

    # exactly what the engine launches: the row pass reading the W2^T shadow copies of the gradient workspace (refreshed here once),
    # its observation rows pre-gathered by the previous launch (the first launch's rows by mppo_gather_rows), extra workgroups
    # gathering the next minibatch's rows
    tr.lib.shadow_refresh(C.byref(tr.net), reg["params"].data_ptr(), mb, reg["grad_ws"].data_ptr(), wsb, s.cuda_stream)
    pre = os.environ.get("MPPO_NO_PREGATHER", "0") != "1"
    if pre:
        tr.lib.gather_rows(C.byref(tr.net), C.byref(batch), reg["perm"].data_ptr(), mb, reg["grad_ws"].data_ptr(), wsb, 0, s.cuda_stream)

    def launch(k):
        idx = reg["perm"].data_ptr() + 4 * (k % M) * mb
        if pre:
            tr.lib.minibatch_rowpass_pre(C.byref(tr.net), reg["params"].data_ptr(), C.byref(batch), idx, reg["perm"].data_ptr() + 4 * ((k + 1) % M) * mb, mb,
                                         reg["adv_stats"].data_ptr() + 8 * (k % M), 1.0 / mb, C.byref(lc), reg["grad_ws"].data_ptr(), wsb, k & 1, s.cuda_stream)
        else:
            tr.lib.minibatch_rowpass_shadow(C.byref(tr.net), reg["params"].data_ptr(), C.byref(batch), idx, mb,
                                            reg["adv_stats"].data_ptr() + 8 * (k % M), 1.0 / mb, C.byref(lc), reg["grad_ws"].data_ptr(), wsb, s.cuda_stream)

    for k in range(4):  # (an even number: the two row buffers alternate)
        launch(k)
    if pre:  # buffer 0 holds the rows of minibatch 0 again when the captured sequence starts (and when it restarts: launches % M == 0)
        tr.lib.gather_rows(C.byref(tr.net), C.byref(batch), reg["perm"].data_ptr(), mb, reg["grad_ws"].data_ptr(), wsb, 0, s.cuda_stream)
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for k in range(launches):
            launch(k)
    g.replay()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    ev0.record(cur)
    for _ in range(replays):
        g.replay()
    ev1.record(cur)
    torch.cuda.synchronize()
    sec = ev0.elapsed_time(ev1) * 1e-3 / (launches * replays)
    O, A, H = tr.O, tr.A, tr.H
    macs_row = (2 * O * H + 2 * H * H + H * (A + 1)) + (2 * H * H + H * (A + 1))  # forward (both nets) + dZ2, dZ1 (both nets)
    flops = 2.0 * macs_row * mb
    return sec, flops, f"fused_mlp_kernel: row pass of one minibatch (mb={mb}, O={O}, H={H}, A={A}, actor+critic; {'bf16 MFMA 16x16x16, f32 accumulate' if tr.net.bf16 else 'f32 MFMA 16x16x4'})"


def _usable_cores() -> int:
    """Host cores this process may actually use: the scheduler affinity mask and the cgroup CPU quota (a GPU box hands a container the
    CPU share of its GPUs - 16 cores per GPU on this pool - while os.cpu_count() still reports every core of the machine)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def _cpu_baseline_worker(config_name: str, overrides, n_envs: int, updates: int) -> dict:
    """Runs in a fresh interpreter without a GPU: `updates` + 1 whole updates (rollout + PPO update) of the CPU restatement - the environment
    step by the C++17 / OpenMP float32 twin (oracle/cpu_twin/env_twin.cpp: one environment per thread, all host cores, -O3 -march=native,
    built on this machine), the policy and the PPO update by torch-CPU float32 with BLAS on all cores (oracle/cpu_twin/ppo_torch.py).
    The first update is warm-up (thread pools, page faults); the others are timed."""
    import numpy as np
    import torch

    from minppo_amd.config import load_config_from_cli
    from minppo_amd.model import load_model
    from oracle import ppo_oracle as po
    from oracle.cpu_twin import Twin, ppo_torch as pt
    from oracle.env_oracle import default_hp

    cores = int(os.environ.get("MPPO_BENCH_CPU_THREADS", "0")) or _usable_cores()
    torch.set_num_threads(cores)
    cfg = load_config_from_cli([config_name, f"training.num_envs={n_envs}", *overrides])
    cm = load_model(cfg.environment.model or cfg.kscale_id)
    tw = Twin(cm, include_c_vals=bool(cfg.environment.include_c_vals), threads=cores)
    N, T, A, H, M, E = n_envs, cfg.training.num_steps, cm.nu, cfg.model.hidden_size, cfg.training.num_minibatches, cfg.training.update_epochs
    O = tw.obs_dim
    named = po.init_params(cfg.training.seed, O, A, H, np.float32)
    p = {k: torch.tensor(np.asarray(v, np.float32)) for k, v in named.items()}
    hp = default_hp(cfg)
    opt = pt.Adam(p, hp["lr_train"] if hp["anneal_lr"] else hp["lr_opt"], hp["max_grad_norm"], hp["anneal_lr"], N * T // M, E, max(hp["num_updates"], 1))
    last_obs = torch.from_numpy(tw.reset(N)[:, :O].copy())
    gen = torch.Generator().manual_seed(1337)
    times, rollout_s = [], []
    mean_reward = 0.0
    u = 0
    while True:  # one warm-up update, then ~10 s of timed updates: at least `updates` of them (one, if an update takes more than 15 s), at most 40
        if u >= 2 and times[0] > 15.0:
            break
        if u >= updates + 1 and (sum(times[1:]) >= 10.0 or u >= 41):
            break
        u += 1
        noise = torch.randn(T, N, A, generator=gen)
        perms = torch.stack([torch.randperm(N * T, generator=gen) for _ in range(E)])
        t0 = time.perf_counter()
        last_obs, mean_reward = pt.one_update(tw, p, opt, last_obs, noise, perms, M, hp, bool(cfg.model.use_tanh))
        times.append(time.perf_counter() - t0)
    # the environment step alone (same states, fresh actions): how the update's time splits
    acts = [np.random.default_rng(5).standard_normal((N, A)).astype(np.float32) for _ in range(3)]
    t0 = time.perf_counter()
    for a_ in acts:
        tw.step(a_)
    env_ms = 1e3 * (time.perf_counter() - t0) / len(acts)
    dt = float(np.mean(times[1:]))
    return {"seconds_per_update": dt, "env_step_ms": env_ms, "omp_threads": tw.threads, "torch_threads": torch.get_num_threads(), "host_cores": os.cpu_count() or 1, "usable_cores": cores, "N": N, "T": T,
            "E": E, "M": M, "updates_timed": len(times) - 1, "mean_reward": mean_reward, "finite": bool(all(torch.isfinite(v).all() for v in p.values()))}


def cpu_baseline(config_name: str, overrides, n_envs: int):
    """`cpu_baseline` of the JSON line: the CPU restatement SURVEY.md 8(d) specifies - a C++ / OpenMP float32 twin of the environment step on
    all host cores plus the PPO update on torch-CPU / BLAS - timed on a bounded sample of the same workload (whole updates at `n_envs`
    environments) in a child process that never sees the GPU.  The reference's own JAX-CPU path cannot run here (no jax / brax / mujoco on
    the image, no network): BASELINE.md.  Baseline only: a GPU / CPU ratio says nothing about kernel quality, the roofline fraction does."""
    import subprocess

    code = ("import sys, json; sys.path.insert(0, %r); import bench; "
            "print(json.dumps(bench._cpu_baseline_worker(sys.argv[1], json.loads(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))))" % str(ROOT))
    env = {k: v for k, v in os.environ.items() if k not in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS")}
    env.update(HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", OMP_PROC_BIND="false")
    updates = int(os.environ.get("MPPO_BENCH_CPU_UPDATES", "3"))
    try:
        r = subprocess.run([sys.executable, "-c", code, config_name, json.dumps(list(overrides)), str(n_envs), str(updates)], env=env, capture_output=True, text=True,
                           timeout=float(os.environ.get("MPPO_BENCH_CPU_TIMEOUT", "150")))
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "env-steps/s", "cores": _usable_cores(), "kind": "port", "sample": "the CPU restatement did not finish its bounded sample within the time limit"}
    if r.returncode != 0:
        raise RuntimeError("the CPU-baseline worker failed:\n" + r.stderr[-2000:])
    w = json.loads(r.stdout.strip().splitlines()[-1])
    return {"value": w["N"] * w["T"] / w["seconds_per_update"], "unit": "env-steps/s", "cores": int(max(w["omp_threads"], w["torch_threads"])), "kind": "port",
            "port_kind": "restatement-c++: environment step = C++17 / OpenMP float32 twin (oracle/cpu_twin/env_twin.cpp, g++ -O3 -march=native, one environment per thread); "
                         "policy + PPO update = torch-CPU float32 with BLAS threads (oracle/cpu_twin/ppo_torch.py); checked against oracle/ and the golden fixtures (tests/test_cpu_twin.py)",
            "sample": f"{w['updates_timed']} whole updates (after one warm-up update) at {w['N']} envs x {w['T']} steps + {w['E']}x{w['M']} minibatch steps: "
                      f"{w['seconds_per_update']:.3f} s per update, of which the {w['T']} environment steps {w['T'] * w['env_step_ms'] / 1e3:.3f} s "
                      f"({w['env_step_ms']:.1f} ms per step of {w['N']} envs on {w['omp_threads']} OpenMP threads; torch: {w['torch_threads']} threads)",
            "host_cores": int(w["host_cores"]), "usable_cores": int(w["usable_cores"]), "sanity": {"mean_reward": w["mean_reward"], "finite": w["finite"]}}


def spawn_ranks(args: argparse.Namespace) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: this process becomes a supervisor that starts
    one fresh child per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, exactly what
    torch.distributed.run would set) and forwards rank 0's JSON line.  It never touches the GPU itself
    (`torch.cuda.device_count()` does not initialise HIP on this image), so nothing that has initialised a GPU is
    ever re-executed.  If the ranks fail or exceed the time limit with the default transport of the gradients (the engine's
    peer-to-peer exchange inside the hipGraph, csrc/peer.h), the exact child PIDs are killed and the run is repeated once
    with RCCL all-reduces launched eagerly (MPPO_ALLREDUCE=rccl).  MPPO_BENCH_SHARE_GPU=1: all ranks on GPU 0 (how the
    multi-rank path is measured on a one-GPU box; rendezvous over gloo)."""
    import socket
    import subprocess

    import torch

    have = torch.cuda.device_count()
    share = os.environ.get("MPPO_BENCH_SHARE_GPU") == "1"
    if have < (1 if share else args.gpus):
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but this machine exposes {have} GPU(s); nothing was run\n")
        return 2

    def free_port() -> int:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        p = s.getsockname()[1]
        s.close()
        return p

    def attempt(extra_env: dict, limit_s: float):
        port = free_port()
        procs = []
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(0 if share else r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                       MPPO_BENCH_WORKER="1", **extra_env)  # the children measure; this process is already their supervisor
            procs.append(subprocess.Popen([sys.executable, os.environ.get("MPPO_BENCH_WORKER_SCRIPT", str(Path(__file__).resolve())), *sys.argv[1:]], env=env,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None))
        deadline = time.monotonic() + limit_s
        line, ok = b"", True
        try:
            out0, _ = procs[0].communicate(timeout=limit_s)
            line = out0
            for p in procs[1:]:
                p.wait(timeout=max(1.0, deadline - time.monotonic()))
        except subprocess.TimeoutExpired:
            ok = False
        for p in procs:  # exactly the PIDs started above, never a pattern
            if p.poll() is None:
                p.kill()
                p.wait()
        ok = ok and all(p.returncode == 0 for p in procs)
        return ok, line

    limit = float(os.environ.get("MPPO_BENCH_RANK_TIMEOUT", "420"))
    ok, line = attempt({}, limit)
    if not ok and os.environ.get("MPPO_ALLREDUCE", "peer") != "rccl" and not share:
        sys.stderr.write("bench.py: ranks failed with the peer-to-peer exchange; repeating with eager RCCL all-reduces (MPPO_ALLREDUCE=rccl)\n")
        ok, line = attempt({"MPPO_ALLREDUCE": "rccl", "MPPO_GRAPH_COMM": "0"}, limit)
    if not ok:
        sys.stderr.write("bench.py: a rank failed or timed out\n")
        return 1
    js = [l for l in line.decode(errors="replace").splitlines() if l.startswith("{")]
    if not js:
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
        return 1
    sys.stdout.write(js[-1] + "\n")
    sys.stdout.flush()
    return 0


def supervise_rank(args: argparse.Namespace) -> int:
    """A rank started by `python -m torch.distributed.run ... bench.py --gpus N` (the driver's spelling).  The rank process
    itself stays off the GPU and runs the measurement in ONE child (same file, MPPO_BENCH_WORKER=1), so that a failure or
    a hang of the first attempt - the engine's peer-to-peer exchange inside the hipGraph, the default - can be answered the same way
    `spawn_ranks` answers it: every rank kills exactly its own child and the run is repeated once with eager RCCL all-reduces
    (MPPO_ALLREDUCE=rccl).  The ranks agree through status files in a per-job directory (one node: nnodes = 1); the retry
    rendezvous on a fresh port picked by rank 0 instead of the launcher's store (which still holds the first attempt's keys)."""
    import signal
    import socket
    import subprocess
    import tempfile

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    # one directory per job: the ranks of a launcher share its pid as their parent (tests name the job themselves)
    job = "mppo_bench_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("MPPO_BENCH_JOB", str(os.getppid())))
    d = Path(os.environ.get("MPPO_BENCH_STATUS_DIR", tempfile.gettempdir())) / job
    d.mkdir(parents=True, exist_ok=True)
    limit = float(os.environ.get("MPPO_BENCH_RANK_TIMEOUT", "300"))
    child_argv = [sys.executable, os.environ.get("MPPO_BENCH_WORKER_SCRIPT", str(Path(__file__).resolve())), *sys.argv[1:]]
    state = {"proc": None}

    def on_term(signum, frame):  # the launcher is tearing the job down: take the child along (its exact PID)
        p = state["proc"]
        if p is not None and p.poll() is None:
            p.kill()
        os._exit(128 + signum)

    signal.signal(signal.SIGTERM, on_term)
    signal.signal(signal.SIGINT, on_term)

    def status_file(attempt: int, r: int) -> Path:
        return d / f"attempt{attempt}.rank{r}"

    def run_attempt(attempt: int, extra_env: dict):
        env = dict(os.environ, MPPO_BENCH_WORKER="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), **extra_env)
        p = subprocess.Popen(child_argv, env=env, stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, stderr=None)
        state["proc"] = p
        deadline = time.monotonic() + limit
        ok, out = True, b""
        while True:
            try:
                o, _ = p.communicate(timeout=1.0)
                out = o or b""
                ok = p.returncode == 0
                break
            except subprocess.TimeoutExpired:
                peer_failed = any(status_file(attempt, r).exists() and status_file(attempt, r).read_text().strip() != "ok" for r in range(world))
                if peer_failed or time.monotonic() > deadline:
                    p.kill()
                    o, _ = p.communicate()
                    ok = False
                    break
        state["proc"] = None
        tmp = status_file(attempt, rank).with_suffix(f".rank{rank}.tmp")
        tmp.write_text("ok" if ok else "fail")
        tmp.rename(status_file(attempt, rank))
        # everybody's verdict (a rank that never reports counts as failed)
        wait_until = time.monotonic() + limit + 60
        while time.monotonic() < wait_until and not all(status_file(attempt, r).exists() for r in range(world)):
            time.sleep(0.2)
        all_ok = all(status_file(attempt, r).exists() and status_file(attempt, r).read_text().strip() == "ok" for r in range(world))
        return all_ok, out

    ok, out = run_attempt(0, {})
    if not ok and os.environ.get("MPPO_ALLREDUCE", "peer") != "rccl":
        if rank == 0:
            sys.stderr.write("bench.py: ranks failed with the peer-to-peer exchange; repeating with eager RCCL all-reduces (MPPO_ALLREDUCE=rccl)\n")
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
            s.close()
            tmp = d / "retry_port.tmp"
            tmp.write_text(str(port))
            tmp.rename(d / "retry_port")
        wait_until = time.monotonic() + 120
        while time.monotonic() < wait_until and not (d / "retry_port").exists():
            time.sleep(0.2)
        if (d / "retry_port").exists():
            port = (d / "retry_port").read_text().strip()
            ok, out = run_attempt(1, {"MPPO_ALLREDUCE": "rccl", "MPPO_GRAPH_COMM": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port, "TORCHELASTIC_USE_AGENT_STORE": "False"})
    # leave nothing behind: every rank has read every verdict by now (run_attempt returns after all status files exist or the wait
    # expired); rank 0 removes the directory a moment later, whatever is still in it
    if rank == 0:
        import shutil
        time.sleep(1.0)
        shutil.rmtree(d, ignore_errors=True)
    if not ok:
        sys.stderr.write(f"bench.py: rank {rank}: the run failed or timed out\n")
        return 1
    if rank == 0:
        js = [l for l in out.decode(errors="replace").splitlines() if l.startswith("{")]
        if not js:
            sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
            return 1
        sys.stdout.write(js[-1] + "\n")
        sys.stdout.flush()
    return 0


def main() -> None:
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))  # before anything in this process touches a GPU
    if args.gpus > 1 and os.environ.get("MPPO_BENCH_WORKER") != "1" and os.environ.get("MPPO_BENCH_SUPERVISE", "1") != "0":
        raise SystemExit(supervise_rank(args))  # under a launcher: this process supervises, its child measures
    # The contract is ONE JSON line on stdout.  RCCL prints a version banner through C stdio on stdout (flushed at
    # exit, i.e. AFTER anything Python printed), so fd 1 is pointed at stderr for the life of the process and the JSON
    # line is written to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    share = os.environ.get("MPPO_BENCH_SHARE_GPU") == "1"  # every rank on GPU 0 (measurement of the multi-rank path on a one-GPU box)
    if share:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) are visible")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)  # RCCL refuses two ranks on one device
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from minppo_amd.config import load_config_from_cli
    from minppo_amd.train import Trainer

    n_global = args.envs_per_gpu * world
    # MPPO_BENCH_INDEPENDENT=1 (measurement only, with MPPO_BENCH_SHARE_GPU=1): the ranks train independent replicas side by side, no
    # gradient exchange - what the sharing of one GPU by `world` processes costs by itself, the baseline of the exchange's price
    indep = world > 1 and os.environ.get("MPPO_BENCH_INDEPENDENT") == "1"
    if indep:
        cfg = load_config_from_cli([args.config, f"training.num_envs={args.envs_per_gpu}", *args.set])
        tr = Trainer(cfg, device=f"cuda:{local_rank}", rank=0, world_size=1, seed=1337 + rank, use_graph=not args.no_graph)
        transport = "independent replicas (no exchange)"
    else:
        cfg = load_config_from_cli([args.config, f"training.num_envs={n_global}", *args.set])
        tr = Trainer(cfg, device=f"cuda:{local_rank}", rank=rank, world_size=world, use_graph=not args.no_graph)
        transport = tr.init_comm()  # "peer" (default; csrc/peer.h) or "rccl" ($MPPO_ALLREDUCE), "none" for one rank
    tr.reset()
    # A fresh process stalls ONCE for 70-90 ms some 30-40 ms after its first GPU work (measured per update by
    # tools/ramp_probe.py, graph replay and eager launches alike: profiles/r02_g_ramp.txt); with W = 3 warm-up updates
    # (21 ms) that stall would land in the timed region every few runs.  It is absorbed here, before the W warm-up steps,
    # by 0.3 s of unrelated device work (plumbing: torch elementwise passes over 64 MB); the W + K steps below are exactly the contract's.
    pre_warm_s = float(os.environ.get("MPPO_BENCH_PREWARM_S", "0.3"))
    if pre_warm_s > 0:
        xw = torch.zeros(16 << 20, device=f"cuda:{local_rank}")  # 64 MB, elementwise passes (no BLAS library is pulled in)
        t_pw = time.perf_counter()
        while time.perf_counter() - t_pw < pre_warm_s:
            for _ in range(20):
                xw.mul_(0.999).add_(1.0)
            torch.cuda.synchronize()
        del xw
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
    replicas_identical = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        t = t if share else t.to(f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # replicas must hold bit-identical parameters after the same updates: a checksum of the parameter bits, compared over the ranks
    if world > 1 and not indep:
        tr.check_peers()
        import numpy as _np
        bits = tr.params_flat().view(_np.uint32).astype(_np.uint64)
        chk = torch.tensor([int(bits.sum()), int((bits * (_np.arange(bits.size, dtype=_np.uint64) % 65521 + 1)).sum() % (1 << 62))], dtype=torch.int64)
        chk = chk if share else chk.to(f"cuda:{local_rank}")
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_identical = bool((lo == hi).all().item())
    steps_total = tr.T * n_global * args.steps
    stats = tr.rollout_stats()
    lossm = tr.losses().reshape(-1, 4).mean(0)

    out = None
    if rank == 0:
        sec, flops, desc = rowpass_probe(tr)
        achieved = flops / sec / 1e12
        # HBM-side bytes per launch of the roofline kernel: PMC counters need rocprofv3 around the process, so they are
        # collected by tools/gpu_traffic.sh (separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 fetch correction) and
        # the committed summary is read here; null when the workload is not the one the summary was taken on.
        baseline_cfg = {"stompy_pro": "BASELINE configs[1]; configs[2] at 8 GPUs", "stompy_full": "BASELINE configs[4]"}.get(args.config, "not a BASELINE config")
        bf16 = cfg.training.mlp_dtype == "bf16"
        if bf16:
            baseline_cfg = "BASELINE configs[3]" if args.config == "stompy_pro" else baseline_cfg + ", bf16 MFMA"
        peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
        traffic, traffic_src = None, None
        tfs = sorted((ROOT / "profiles").glob("r*_hbm_traffic.json"))  # the newest committed counter summary
        if tfs and args.config == "stompy_pro" and args.envs_per_gpu == 4096 and not bf16:
            k = json.loads(tfs[-1].read_text())["kernels"]
            # the training row pass of a float network: <BF16 = false, ROLLOUT = false, OT = 1, W2T shadow = true> (the engine's choice)
            key = next((n for n in k if n.endswith("fused_mlp_kernel<false, false, 1, true, true>")), None) or \
                next((n for n in k if n.endswith("fused_mlp_kernel<false, false, 1, true>")), None) or \
                next((n for n in k if n.endswith("fused_mlp_kernel<false, false, 1>")), None)
            if key:
                traffic = k[key]["hbm_bytes_per_launch"]
                traffic_src = f"profiles/{tfs[-1].name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 corrections)"
        # matrix-core utilisation of the same kernel from SQ counters (tools/pmc_mlp.sh: rocprofv3 --pmc passes of their own): the newest
        # committed summary; MFMA-busy cycles / (1024 SIMD pipes x launch duration x 2.4 GHz)
        mfma_busy, mfma_src = None, None
        pfs = sorted((ROOT / "profiles").glob("r*_mlp_pmc_bf16.json" if bf16 else "r*_mlp_pmc.json"))
        if pfs and args.config == "stompy_pro" and args.envs_per_gpu == 4096:
            kk = json.loads(pfs[-1].read_text())["kernels"]
            name = "fused_mlp_kernel<true, false, 1, true, true>" if bf16 else "fused_mlp_kernel<false, false, 1, true, true>"
            if name in kk and "mfma_busy_frac" in kk[name]:
                mfma_busy = kk[name]["mfma_busy_frac"]
                mfma_src = f"profiles/{pfs[-1].name} (SQ_VALU_MFMA_BUSY_CYCLES over 1024 SIMD pipes x the launch's duration under the counter pass x 2.4 GHz)"
        # the same kernel INSIDE the update, from the newest committed `rocprofv3 --kernel-trace --stats` summary of this command (the probe above
        # replays the row pass back to back; in the update it follows an Adam launch and runs ~1 us longer): reported beside the probe's figure
        in_situ = None
        ks = sorted((ROOT / "profiles").glob("r*_kernel_stats_config3_bf16.csv" if bf16 else "r*_kernel_stats.csv"))
        if ks and args.config == "stompy_pro" and args.envs_per_gpu == 4096:
            import csv
            want = "fused_mlp_kernel<true, false, 1, true, true>" if bf16 else "fused_mlp_kernel<false, false, 1, true, true>"
            for row in csv.DictReader(ks[-1].open()):
                if want in row.get("Name", ""):
                    us = float(row["AverageNs"]) * 1e-3
                    in_situ = {"us_per_launch": us, "achieved": flops / (us * 1e-6) / 1e12, "frac": flops / (us * 1e-6) / 1e12 / peak, "launches": int(row["Calls"]),
                               "source": f"profiles/{ks[-1].name} (rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline)"}
                    break
        out = {
            "metric": ("env-steps/sec (whole node), stompy_pro 4096 envs, 1/2/4/8 MI355X" if args.config == "stompy_pro" and args.envs_per_gpu == 4096
                       else f"env-steps/sec (whole node), {args.config} {args.envs_per_gpu} envs per GPU, {world} MI355X"),
            "value": steps_total / dt,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16" if bf16 else "f32",
            "data": f"synthetic (stand-in robot {tr.cm.name}, random-init weights, Philox action noise)",
            "config": {"workload": f"{args.config}: {args.envs_per_gpu} envs/GPU x T={tr.T} rollout + {tr.E}x{tr.M} minibatch PPO update, O={tr.O} A={tr.A} H={tr.H}, {'bf16-in/f32-acc MLP products, f32 elsewhere' if bf16 else 'fp32'} ({baseline_cfg})",
                       "global_envs": n_global, "parallelism": (f"env-sharded dp{world}, gradients summed per optimizer step: " +
                                                                           ("peer-to-peer exchange over hipIpc-mapped buffers fused into the weight-gradient and Adam launches (csrc/peer.h)" if transport == "peer"
                                                                            else "RCCL all-reduce") + (", all ranks on ONE GPU" if share else "")) if world > 1 else "single GPU",
                       "allreduce": transport + (f" ({tr.peer_form()})" if transport == "peer" else ""), "replicas_identical": replicas_identical,
                       "hipgraph": bool(tr.graph_active()),
                       "pre_warm": f"{pre_warm_s:.1f} s of unrelated device work before the {args.warmup} warm-up steps (one-time start-up stall of the device, see bench.py)"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src, "mfma_busy_frac": mfma_busy, "mfma_busy_source": mfma_src, "kernel": desc, "us_per_launch": sec * 1e6, "in_situ": in_situ,
                         "whole_update_mlp_tflops": (26.0 * (2 * tr.O * tr.H + 2 * tr.H * tr.H + tr.H * (tr.A + 1)) + 0.2 * (tr.O * tr.H + tr.H * tr.H + tr.H)) * steps_total / world / dt / 1e12},
            "sanity": {"mean_reward": stats["mean_reward"], "done_fraction": stats["done_fraction"], "mean_total_loss": float(lossm[0])},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.config, args.set, args.cpu_baseline_envs)
            out["cpu_baseline"]["engine_over_cpu"] = out["value"] / out["cpu_baseline"]["value"] if out["cpu_baseline"]["value"] else None  # (reported, not a target: the roofline fraction is what describes the kernels)
        else:
            out["cpu_baseline"] = None
    tr.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
