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


def kernel_sources_sha() -> str:
    """Identity of the kernels a measurement describes: sha256 over the text of everything the shipped library is compiled from
    (minppo_amd/csrc/*.hip|*.h|*.inc, include/minppo_hip.h, the flags in minppo_amd/build.py).  Profile summaries under profiles/ record it
    (tools/profile_meta.py writes `<summary>.meta.json` next to each one, on the GPU box, where there is no .git); this file quotes a
    committed summary only if its recorded hash equals the hash of the tree it runs from."""
    import hashlib

    h = hashlib.sha256()
    files = sorted((ROOT / "minppo_amd" / "csrc").glob("*.hip")) + sorted((ROOT / "minppo_amd" / "csrc").glob("*.h")) + sorted((ROOT / "minppo_amd" / "csrc").glob("*.inc"))
    files += [ROOT / "include" / "minppo_hip.h", ROOT / "minppo_amd" / "build.py"]
    for f in files:
        h.update(f.name.encode() + b"\0" + f.read_bytes() + b"\0")
    return h.hexdigest()[:16]


def committed_summary(pattern: str):
    """The newest committed profile summary matching `pattern` whose sidecar (`<name>.meta.json`, tools/profile_meta.py) says it was
    taken on THESE kernel sources; (path, meta) or (None, why).  A summary without a sidecar, or with another tree's hash, describes
    other kernels and is not quoted (round-4 review: the line carried counters of a kernel that had changed since)."""
    here = kernel_sources_sha()
    cands = sorted((ROOT / "profiles").glob(pattern))
    why = None  # (the reason the NEWEST candidate is not quoted: that is the one a reader would look for)
    for f in reversed(cands):
        side = f.with_name(f.name + ".meta.json")
        if not side.exists():
            why = why or f"profiles/{f.name} has no .meta.json (taken before summaries recorded their tree): not quoted"
            continue
        meta = json.loads(side.read_text())
        if meta.get("kernel_sources_sha") != here:
            why = why or f"profiles/{f.name} was taken on kernel sources {meta.get('kernel_sources_sha')} (commit {meta.get('git_head')}), this tree is {here}: stale, not quoted"
            continue
        return f, meta
    return None, why or f"no profiles/{pattern}"


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
    ap.add_argument("--no-probe", action="store_true", help="do not run the stand-alone row-pass probe (profiling runs: the rocprofv3 summary of the "
                    "update then contains the update's launches of the kernel only; the line's `roofline` is null)")
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
    return sec, flops, f"{'bf16_rowpass_kernel' if tr.net.bf16 else 'fused_mlp_kernel'}: row pass of one minibatch (mb={mb}, O={O}, H={H}, A={A}, actor+critic; {'bf16 MFMA 16x16x16, f32 accumulate' if tr.net.bf16 else 'f32 MFMA 16x16x4'})"


def rowpass_kernel_name(bf16: bool) -> str:
    """Name (as rocprofv3 prints it) of the training row pass the engine launches for the headline geometry."""
    return "bf16_rowpass_kernel<1, 8, true>" if bf16 else "fused_mlp_kernel<false, false, 1, true, true, 1>"  # (<BF16, ROLLOUT, OT, W2T, PRE, RT>)


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
        last_obs, mean_reward, _ = pt.one_update(tw, p, opt, last_obs, noise, perms, M, hp, bool(cfg.model.use_tanh))
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


def _attempt_limits():
    """Time limits of a multi-rank run, chosen so that a hung first attempt + teardown + the RCCL repeat fit the driver's 600 s command
    limit with room to spare (<= 540 s): every rank process reports `started` once it has imported torch, loaded the library and
    touched its GPU (a fresh box pages the image in first: that is not the transport's fault and gets its own limit, `start`), and
    from the moment ALL ranks have started an attempt may take `peer` seconds (attempt 0: rendezvous, hipIpc mapping, the <= 10 s
    self-test, hipGraph capture, W + K updates: well under 30 s when healthy) or `rccl` seconds (the repeat).  Worst case
    start + peer + teardown + start' + rccl = 90 + 90 + ~5 + ~15 (warm page cache) + 300 < 540.  $MPPO_BENCH_RANK_TIMEOUT overrides
    `peer`, $MPPO_BENCH_RETRY_TIMEOUT `rccl`, $MPPO_BENCH_START_TIMEOUT `start`."""
    return {"start": float(os.environ.get("MPPO_BENCH_START_TIMEOUT", "90")), "peer": float(os.environ.get("MPPO_BENCH_RANK_TIMEOUT", "90")),
            "rccl": float(os.environ.get("MPPO_BENCH_RETRY_TIMEOUT", "300"))}


def _started_file(d, attempt: int, r: int) -> Path:
    return Path(d) / f"attempt{attempt}.started.rank{r}"


class _AttemptClock:
    """Deadline of one attempt as the supervisors see it: `start` seconds until every rank has reported `started`, then `run` seconds."""

    def __init__(self, d, attempt: int, world: int, start_s: float, run_s: float):
        self.d, self.attempt, self.world, self.run_s = d, attempt, world, run_s
        self.t_start_deadline = time.monotonic() + start_s
        self.t_run_deadline = None

    def expired(self) -> str:
        """'' while the attempt may go on, else the reason it may not."""
        now = time.monotonic()
        if self.t_run_deadline is None:
            if all(_started_file(self.d, self.attempt, r).exists() for r in range(self.world)):
                self.t_run_deadline = now + self.run_s
            elif now > self.t_start_deadline:
                return "a rank did not start (import torch, load the library, touch its GPU) within its limit"
            return ""
        return f"the ranks did not finish within {self.run_s:.0f} s of having started" if now > self.t_run_deadline else ""


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

    import shutil
    import tempfile

    lim = _attempt_limits()
    sdir = Path(tempfile.mkdtemp(prefix="mppo_bench_"))

    def attempt(no: int, extra_env: dict, run_s: float):
        port = free_port()
        procs = []
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(0 if share else r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                       MPPO_BENCH_WORKER="1", MPPO_BENCH_MILESTONE_DIR=str(sdir), MPPO_BENCH_ATTEMPT=str(no), **extra_env)  # the children measure; this process is already their supervisor
            procs.append(subprocess.Popen([sys.executable, os.environ.get("MPPO_BENCH_WORKER_SCRIPT", str(Path(__file__).resolve())), *sys.argv[1:]], env=env,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None))
        clock = _AttemptClock(sdir, no, args.gpus, lim["start"], run_s)
        line, why, got0 = b"", "", False
        while True:
            if got0:
                time.sleep(0.2)
            else:
                try:
                    line, _ = procs[0].communicate(timeout=0.5)  # (may be called again after a TimeoutExpired: nothing read so far is lost)
                    got0 = True
                except subprocess.TimeoutExpired:
                    pass
            if all(p.poll() is not None for p in procs):
                break
            dead = [r for r, p in enumerate(procs) if p.poll() is not None and p.returncode != 0]
            why = f"rank {dead[0]} exited with status {procs[dead[0]].returncode}" if dead else clock.expired()
            if why:
                break
        for p in procs:  # exactly the PIDs started above, never a pattern
            if p.poll() is None:
                p.kill()
                p.wait()
        if not why and not all(p.returncode == 0 for p in procs):
            why = "a rank exited with a non-zero status"
        return (not why), line, why

    try:
        ok, line, why = attempt(0, {}, lim["peer"])
        if not ok and os.environ.get("MPPO_ALLREDUCE", "peer") != "rccl" and not share:
            sys.stderr.write(f"bench.py: the first attempt (peer-to-peer exchange) failed: {why}; repeating with eager RCCL all-reduces (MPPO_ALLREDUCE=rccl)\n")
            ok, line, why = attempt(1, {"MPPO_ALLREDUCE": "rccl", "MPPO_GRAPH_COMM": "0"}, lim["rccl"])
    finally:
        shutil.rmtree(sdir, ignore_errors=True)
    if not ok:
        sys.stderr.write(f"bench.py: a rank failed or timed out: {why}\n")
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
    lim = _attempt_limits()
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

    def run_attempt(attempt: int, extra_env: dict, run_s: float):
        env = dict(os.environ, MPPO_BENCH_WORKER="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   MPPO_BENCH_MILESTONE_DIR=str(d), MPPO_BENCH_ATTEMPT=str(attempt), **extra_env)
        p = subprocess.Popen(child_argv, env=env, stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, stderr=None)
        state["proc"] = p
        clock = _AttemptClock(d, attempt, world, lim["start"], run_s)
        ok, out, why = True, b"", ""
        while True:
            try:
                o, _ = p.communicate(timeout=0.5)
                out = o or b""
                ok = p.returncode == 0
                why = "" if ok else f"rank {rank}'s worker exited with status {p.returncode}"
                break
            except subprocess.TimeoutExpired:
                failed = [r for r in range(world) if status_file(attempt, r).exists() and status_file(attempt, r).read_text().strip() != "ok"]
                why = f"rank {failed[0]} reported a failure" if failed else clock.expired()
                if why:
                    p.kill()
                    o, _ = p.communicate()
                    ok = False
                    break
        state["proc"] = None
        if not ok:
            sys.stderr.write(f"bench.py: rank {rank}: attempt {attempt} ended: {why}\n")
        tmp = status_file(attempt, rank).with_suffix(f".rank{rank}.tmp")
        tmp.write_text("ok" if ok else "fail")
        tmp.rename(status_file(attempt, rank))
        # everybody's verdict: the other supervisors watch the same clock and the same files, so they report within moments of this one
        # (a supervisor that never reports counts as failed)
        wait_until = time.monotonic() + 20.0 + (0.0 if not ok else lim["start"] + run_s)
        while time.monotonic() < wait_until and not all(status_file(attempt, r).exists() for r in range(world)):
            time.sleep(0.2)
        all_ok = all(status_file(attempt, r).exists() and status_file(attempt, r).read_text().strip() == "ok" for r in range(world))
        return all_ok, out, why

    ok, out, why = run_attempt(0, {}, lim["peer"])
    if not ok and os.environ.get("MPPO_ALLREDUCE", "peer") != "rccl":
        if rank == 0:
            sys.stderr.write(f"bench.py: the first attempt (peer-to-peer exchange) failed ({why or 'another rank reported a failure'}); repeating with eager RCCL all-reduces (MPPO_ALLREDUCE=rccl)\n")
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
            s.close()
            tmp = d / "retry_port.tmp"
            tmp.write_text(str(port))
            tmp.rename(d / "retry_port")
        wait_until = time.monotonic() + 30
        while time.monotonic() < wait_until and not (d / "retry_port").exists():
            time.sleep(0.2)
        if (d / "retry_port").exists():
            port = (d / "retry_port").read_text().strip()
            ok, out, why = run_attempt(1, {"MPPO_ALLREDUCE": "rccl", "MPPO_GRAPH_COMM": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port, "TORCHELASTIC_USE_AGENT_STORE": "False"}, lim["rccl"])
    # leave nothing behind: every rank has read every verdict by now (run_attempt returns after all status files exist or the wait
    # expired); rank 0 removes the directory a moment later, whatever is still in it
    if rank == 0:
        import shutil
        time.sleep(1.0)
        shutil.rmtree(d, ignore_errors=True)
    if not ok:
        sys.stderr.write(f"bench.py: rank {rank}: the run failed or timed out ({why or 'another rank reported a failure'})\n")
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
        from minppo_amd import _native as _nat

        torch.zeros(1, device=f"cuda:{local_rank}").add_(1).item()  # the GPU touched (torch's own HIP runtime first, as everywhere else in this file)
        _nat.load()  # the library (and RCCL behind it) paged in: this rank has STARTED (see _attempt_limits)
        if os.environ.get("MPPO_BENCH_MILESTONE_DIR"):
            _started_file(os.environ["MPPO_BENCH_MILESTONE_DIR"], int(os.environ.get("MPPO_BENCH_ATTEMPT", "0")), rank).write_text("started")
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
    t_link = None
    if not indep:
        cfg = load_config_from_cli([args.config, f"training.num_envs={n_global}", *args.set])
        tr = Trainer(cfg, device=f"cuda:{local_rank}", rank=rank, world_size=world, use_graph=not args.no_graph)
        transport = tr.init_comm()  # "peer" (default; csrc/peer.h) or "rccl" ($MPPO_ALLREDUCE), "none" for one rank
        if world > 1 and rank == 0:
            sys.stderr.write(f"bench.py: gradient transport: {transport}" + (f" ({tr.peer_form()})" if transport == "peer" else "") + f" - {tr.comm_note}\n")
        # the measured one-way latency of a system-scope flag between rank 0 and every peer (collective; about 1 ms per peer): the t_link of the
        # efficiency model in DESIGN.md 7.2, for the first run on distinct GPUs to put a number where the model has an assumption
        t_link = tr.peer_latencies() if transport == "peer" else None
        if t_link is not None and rank == 0:
            sys.stderr.write("bench.py: one-way flag latency rank 0 <-> rank 1.." + str(world - 1) + " [us]: " + ", ".join(f"{x:.2f}" for x in t_link) + "\n")
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
        bf16 = cfg.training.mlp_dtype == "bf16"
        peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
        baseline_cfg = {"stompy_pro": "BASELINE configs[1]; configs[2] at 8 GPUs", "stompy_full": "BASELINE configs[4]"}.get(args.config, "not a BASELINE config")
        if bf16:
            baseline_cfg = "BASELINE configs[3]" if args.config == "stompy_pro" else baseline_cfg + ", bf16 MFMA"
        headline_shape = args.config == "stompy_pro" and args.envs_per_gpu == 4096
        roofline = None
        if not args.no_probe:
            sec, flops, desc = rowpass_probe(tr)
            achieved = flops / sec / 1e12
            here = kernel_sources_sha()
            # the training row pass the engine launches for this network: <BF16, ROLLOUT = false, OT = 1, W2T shadow = true, PRE = true>
            kname = rowpass_kernel_name(bf16)
            # HBM-side bytes per launch of the roofline kernel: PMC counters need rocprofv3 around the process, so they are collected by
            # tools/profiles.sh (separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 fetch correction) and the committed summary is read
            # here - ONLY a summary taken on these very kernel sources (committed_summary); null otherwise, with the reason
            traffic, traffic_src = None, None
            if headline_shape:
                f, meta = committed_summary("r*_hbm_traffic_bf16.json" if bf16 else "r*_hbm_traffic.json")
                if f is None:
                    traffic_src = meta
                else:
                    k = json.loads(f.read_text())["kernels"]
                    key = next((n for n in k if n.endswith(kname)), None)
                    if key:
                        traffic = k[key]["hbm_bytes_per_launch"]
                        traffic_src = f"profiles/{f.name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 corrections; commit {meta.get('git_head')}, kernel sources {here})"
            # matrix-core utilisation of the same kernel from SQ counters (tools/pmc_mlp.sh: rocprofv3 --pmc passes of their own):
            # MFMA-busy cycles / (1024 SIMD pipes x launch duration x 2.4 GHz)
            mfma_busy, mfma_src = None, None
            if headline_shape:
                f, meta = committed_summary("r*_mlp_pmc_bf16.json" if bf16 else "r*_mlp_pmc.json")
                if f is None:
                    mfma_src = meta
                else:
                    kk = json.loads(f.read_text())["kernels"]
                    key = next((n for n in kk if n.endswith(kname)), None)
                    if key and "mfma_busy_frac" in kk[key]:
                        mfma_busy = kk[key]["mfma_busy_frac"]
                        mfma_src = (f"profiles/{f.name} (SQ_VALU_MFMA_BUSY_CYCLES over 1024 SIMD pipes x the launch's duration under the counter pass x 2.4 GHz; "
                                    f"commit {meta.get('git_head')}, kernel sources {here})")
            # the same kernel INSIDE the update: the committed `rocprofv3 --kernel-trace --stats` summary of `bench.py --no-probe` (no
            # stand-alone probe launches in the profiled process: Calls must be a whole number of updates x E x M launches), taken on these
            # kernel sources; the probe above replays the row pass back to back, in the update it follows an Adam launch
            in_situ = None
            if headline_shape:
                f, meta = committed_summary("r*_kernel_stats_config3_bf16.csv" if bf16 else "r*_kernel_stats.csv")
                if f is None:
                    in_situ = {"us_per_launch": None, "why": meta}
                else:
                    import csv
                    for row in csv.DictReader(f.open()):
                        if kname in row.get("Name", ""):
                            us, calls = float(row["AverageNs"]) * 1e-3, int(row["Calls"])
                            if calls % (tr.E * tr.M) != 0:
                                in_situ = {"us_per_launch": None, "why": f"profiles/{f.name}: {calls} launches of the row pass is not a whole number of updates x {tr.E * tr.M}: "
                                                                            "the profiled process launched it outside the update as well; not quoted"}
                            else:
                                in_situ = {"us_per_launch": us, "achieved": flops / (us * 1e-6) / 1e12, "frac": flops / (us * 1e-6) / 1e12 / peak, "launches": calls,
                                           "updates": calls // (tr.E * tr.M), "git_head": meta.get("git_head"), "kernel_sources_sha": here,
                                           "source": f"profiles/{f.name} ({meta.get('command', 'rocprofv3 --kernel-trace --stats -- python3 bench.py --no-probe')})"}
                            break
            # Headline figures of the object: the kernel INSIDE the update when a summary of this very tree is committed (it follows an Adam
            # launch that has just rewritten the weights: every XCD's L2 fetches them again, the back-to-back replay never sees that), the
            # live stand-alone replay otherwise - `measured` says which, the other one rides along under `standalone` / `in_situ`.
            standalone = {"us_per_launch": sec * 1e6, "achieved": achieved, "frac": achieved / peak,
                          "how": "64 launches of exactly this minibatch row pass replayed back to back from a hipGraph on the engine stream between HIP events, live in this run"}
            use_situ = bool(in_situ and in_situ.get("us_per_launch"))
            head = in_situ if use_situ else standalone
            # a bf16 network's row pass is not bound by its matrix work (1 us of MFMA in a 10 us launch): what bounds it is the weight stream
            # through the CUs' address paths and the latency-bound phases between - say so, and give the stream's rate
            bound = "mfma"
            stream = None
            if bf16:
                bound = "latency/address-path (the MFMA fraction is reported for the record: DESIGN.md 3.1b)"
                wbytes = 2.0 * (tr.OP * tr.H + 2 * tr.H * tr.H)  # one network's bf16 fragments a workgroup streams per launch: W1, W2, W2^T
                stream = {"weight_bytes_per_workgroup": wbytes, "GB_per_s_per_CU": wbytes / (head["us_per_launch"] * 1e-6) / 1e9,
                          "note": "the weights a workgroup streams divided by the WHOLE launch (the stream is one of its phases: while it runs, a CU's vector-memory path takes 64 B per clock = 150 - 180 GB/s warm, 100 - 120 GB/s behind an optimizer step, tools/wstream_probe.hip)"}
            roofline = {"bound": bound, "achieved": head["achieved"], "peak": peak, "unit": "TFLOP/s", "frac": head["frac"],
                        "measured": ("in situ: rocprofv3 --kernel-trace --stats of `bench.py --no-probe` on this tree's kernels, " + in_situ["source"]) if use_situ
                                    else "stand-alone replay, live (no in-situ summary of this tree's kernels is committed: see in_situ.why)",
                        "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src, "mfma_busy_frac": mfma_busy, "mfma_busy_source": mfma_src, "kernel": desc,
                        "us_per_launch": head["us_per_launch"], "standalone": standalone, "in_situ": in_situ, "weight_stream": stream, "kernel_sources_sha": here,
                        "whole_update_mlp_tflops": (26.0 * (2 * tr.O * tr.H + 2 * tr.H * tr.H + tr.H * (tr.A + 1)) + 0.2 * (tr.O * tr.H + tr.H * tr.H + tr.H)) * steps_total / world / dt / 1e12}
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
            "config": {"workload": f"{args.config}{'' if not cfg.environment.model else ' with robot ' + str(cfg.environment.model)}: {args.envs_per_gpu} envs/GPU x T={tr.T} rollout + {tr.E}x{tr.M} minibatch PPO update, O={tr.O} A={tr.A} H={tr.H}, {'bf16-in/f32-acc MLP products, f32 elsewhere' if bf16 else 'fp32'} ({baseline_cfg})",
                       "global_envs": n_global, "parallelism": (f"env-sharded dp{world}, gradients summed per optimizer step: " +
                                                                           ("peer-to-peer exchange over hipIpc-mapped buffers fused into the weight-gradient and Adam launches (csrc/peer.h)" if transport == "peer"
                                                                            else "RCCL all-reduce") + (", all ranks on ONE GPU" if share else "")) if world > 1 else "single GPU",
                       "allreduce": transport + (f" ({tr.peer_form()})" if transport == "peer" else ""), "replicas_identical": replicas_identical,
                       "t_link_us": t_link,  # one-way latency of a system-scope flag, rank 0 <-> rank q (device-timed ping-pong through the exchange buffers; null: no peer exchange)
                       "hipgraph": bool(tr.graph_active()), "env_kernel": tr.env_kernel,
                       "pre_warm": f"{pre_warm_s:.1f} s of unrelated device work before the {args.warmup} warm-up steps (one-time start-up stall of the device, see bench.py)"},
            "roofline": roofline,
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
