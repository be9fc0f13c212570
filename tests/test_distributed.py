"""Env-sharded data parallelism on the CPU: world_size = 2 over gloo, emulator kernels.

Two ranks with N/2 environments each — gradients summed per optimizer step, advantage statistics summed per
update, rows weighted 1/(mb*world), LR schedule on the global minibatch size — must reproduce the single-process
run on the union of the shards with the block-structured permutation ("sharded oracle" test)."""
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np

from backends import get_backend
from dist_worker import make_inputs
from minppo_amd.config import make_config

HERE = Path(__file__).resolve().parent
OVR = ["training.num_envs=8", "training.num_steps=4", "rl.num_env_steps=4", "training.num_minibatches=2", "training.update_epochs=2", "model.hidden_size=32",
       "training.total_timesteps=3200"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


import os

import pytest


@pytest.mark.parametrize("world,transport", [(2, "peer"), (4, "peer"), (3, "peer"), (8, "peer"), (2, "rccl"), (4, "rccl"), (8, "rccl"), (2, "host"),
                                             (2, "peer_selftest_fails")])
def test_ranks_equal_one_process_on_the_union(tmp_path, world, transport):
    """world ranks through `mppo_engine_learn` (engine.hip do_learn) == one process on the union, for both transports of the
    gradient: "peer" - the exchange fused into the weight-gradient and Adam launches (csrc/peer.h; the emulator's exchange buffers
    are shared-memory segments mapped by every rank process, as hipIpc maps them on the GPU), "rccl" - the communicator branch (the
    emulator's stand-in for ncclAllReduce); "host": the same stages driven from Python (`Trainer.learn_host_driven`) with gloo
    all-reduces.  Three ranks: slices of the gradient that do not divide evenly.  Eight ranks: BASELINE configs[2]'s world size (the
    8-way slice / piece arithmetic of csrc/peer.h, every one of the eight arrival counters and flag lines).  "peer_selftest_fails": rank 1's self-test of the
    mapped buffers (Trainer.init_comm) reports failure - all ranks drop the exchange and continue on the communicator."""
    updates = 2
    port = _free_port()
    host_driven = transport == "host"
    ovr = [o if not o.startswith("training.num_envs=") else f"training.num_envs={ {3: 12, 8: 16}.get(world, 8)}" for o in OVR]
    env = dict(os.environ, MPPO_TEST_HOST_DRIVEN="1" if host_driven else "0", MPPO_ALLREDUCE="rccl" if host_driven else transport)
    if transport == "peer_selftest_fails":
        env.update(MPPO_ALLREDUCE="peer", MPPO_TEST_SELFTEST_FAIL_RANK="1")
    procs = [subprocess.Popen([sys.executable, str(HERE / "dist_worker.py"), str(r), str(world), str(port), str(updates), str(tmp_path / f"r{r}.npz"), *ovr],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    ranks = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    # replicas stay bit-identical (same reduced gradient, same Adam arithmetic)
    for r in range(1, world):
        np.testing.assert_array_equal(ranks[0]["params"], ranks[r]["params"])

    be = get_backend("emu")
    cfg = make_config({"kscale_id": "5eb3cb7f23232298", "visualization": {"camera_name": "track"}}, ovr)
    tr = be.trainer(cfg, external_random=True, use_graph=False)
    tr.reset()
    p0 = tr.params_flat()
    N, T, A, E, M = tr.N, tr.T, tr.A, tr.E, tr.M
    for u in range(updates):
        noise, _, glob = make_inputs(N, T, A, E, M, world, seed=100 + u)
        tr.region("noise", (T, N, A))[:] = noise
        tr.region("perm", (E, T * N))[:] = glob
        tr.update()
    single = tr.params_flat()
    step = np.abs(single - p0).max()
    assert step > 1e-4
    assert np.abs(ranks[0]["params"] - single).max() < 1e-3 * step + 1e-7, (np.abs(ranks[0]["params"] - single).max(), step)
    # per-rank loss partial sums add up to the single-process losses
    np.testing.assert_allclose(sum(r["losses"] for r in ranks), tr.losses(), rtol=1e-4, atol=1e-5)
    # environments are independent: each shard's rewards are the corresponding slice of the union run (last update:
    # the parameters differ by summation order after the first one, so equality is to rounding, not bitwise)
    rew = np.array(tr.region("reward", (T, N)))
    np.testing.assert_allclose(np.concatenate([r["reward"] for r in ranks], 1), rew, atol=5e-3)
    tr.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world,form", [(2, "shared"), (4, "shared"), (2, "fused")])
def test_peer_exchange_between_processes_on_one_gpu(tmp_path, world, form):
    """The peer-to-peer gradient exchange on hardware: `world` rank PROCESSES share cuda:0 (hipIpc maps a buffer of the same device as
    readily as a peer's; RCCL would refuse the duplicate device), each with N / world environments, the exchange's kernels captured
    in every rank's hipGraph.  Ranks == one process on the union of the shards; replicas bit-identical; no wait timed out.
    "shared": the form the engine picks by itself when ranks share a GPU (the two waits of a step are one-wave kernels, so that a
    waiting rank never keeps the peer's kernels off the CUs); "fused": the form for ranks on distinct GPUs - three launches per
    step, the Adam launch waits - forced here at two ranks, where its 64-register waiting workgroups still leave every CU room for
    the peer's largest workgroup (at four ranks on one GPU they would not: csrc/peer.h)."""
    updates = 3
    port = _free_port()
    ovr = ["training.num_envs=256", "training.num_minibatches=4", "training.update_epochs=2", "training.total_timesteps=100000000"]
    env = dict(os.environ, MPPO_TEST_BACKEND="hip", MPPO_TEST_GRAPH="1", MPPO_ALLREDUCE="peer", MPPO_TEST_HOST_DRIVEN="0", MPPO_TEST_PEER_FORM=form,
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    if form == "fused":
        env["MPPO_PEER_MODE"] = "fused"
    else:
        env.pop("MPPO_PEER_MODE", None)
    procs = [subprocess.Popen([sys.executable, str(HERE / "dist_worker.py"), str(r), str(world), str(port), str(updates), str(tmp_path / f"r{r}.npz"), *ovr],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
    try:
        outs = [p.communicate(timeout=600)[0] for p in procs]
    finally:
        for p in procs:  # exactly the PIDs started above
            if p.poll() is None:
                p.kill()
                p.wait()
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    ranks = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    assert all(bool(r["graph"]) for r in ranks), "a rank did not replay its hipGraph"
    for r in range(1, world):
        np.testing.assert_array_equal(ranks[0]["params"], ranks[r]["params"])

    be = get_backend("hip")
    cfg = make_config({"kscale_id": "5eb3cb7f23232298", "visualization": {"camera_name": "track"}}, ovr)
    tr = be.trainer(cfg, external_random=True, use_graph=True)
    tr.reset()
    p0 = tr.params_flat()
    N, T, A, E, M = tr.N, tr.T, tr.A, tr.E, tr.M
    for u in range(updates):
        noise, _, glob = make_inputs(N, T, A, E, M, world, seed=100 + u)
        be.put(tr.region("noise", (T, N, A)), noise)
        be.put(tr.region("perm", (E, T * N)), glob)
        tr.update()
        tr._sync()
    single = tr.params_flat()
    step = np.abs(single - p0).max()
    assert step > 1e-4
    # the gradient's summation order differs (per-rank partial sums added in rank order): rounding-level drift over 24 Adam steps, whose
    # normalisation by sqrt(v) turns a rounding difference of a near-zero gradient into a fraction of one step (lr = 3e-4) for that element
    diff = np.abs(ranks[0]["params"] - single)
    assert diff.max() < 0.1 * step + 1e-7 and np.median(diff) < 1e-3 * step, (diff.max(), np.median(diff), step)
    np.testing.assert_allclose(sum(r["losses"] for r in ranks), tr.losses(), rtol=2e-3, atol=2e-4)
    tr.close()


def _spawn(argv_of, n, env, timeout):
    procs = [subprocess.Popen([sys.executable, str(HERE / "dist_worker.py"), *argv_of(i)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
             for i in range(n)]
    try:
        outs = [p.communicate(timeout=timeout)[0] for p in procs]
    finally:
        for p in procs:  # exactly the PIDs started above
            if p.poll() is None:
                p.kill()
                p.wait()
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    return outs


def _assert_first_step_per_tensor(rank0, m_single, v_single, O, A, H):
    """After ONE optimizer step from zero moments m = (1 - b1) s g and v = (1 - b2) (s g)^2 with s the clip scale: linear / quadratic in
    the gradient, before Adam's 1 / sqrt(v) can turn a rounding difference into a fraction of a step.  A tensor whose rows were weighted
    with anything but 1 / (mb G), or summed over the wrong ranks, is off by a factor here - per tensor, relative to that tensor's scale."""
    from minppo_amd.train import param_slices

    sl, _ = param_slices(O, A, H)
    for name, (off, shape) in sl.items():
        n = int(np.prod(shape))
        for key, single, tol in (("adam_m", m_single, 1e-5), ("adam_v", v_single, 2e-5)):
            a, b = rank0[key][off:off + n], single[off:off + n]
            scale = np.abs(b).max()
            assert scale > 0, (name, key)
            assert np.abs(a - b).max() <= tol * scale, (name, key, np.abs(a - b).max() / scale)


FIRST_STEP = ["training.num_minibatches=1", "training.update_epochs=1"]


@pytest.mark.parametrize("world,transport", [(2, "peer"), (8, "peer"), (8, "rccl")])
def test_first_optimizer_step_matches_the_union_per_tensor(tmp_path, world, transport):
    """One minibatch = the whole batch, one epoch, one update: the ranks' reduced gradient against one process on the union, tensor by
    tensor at 1e-5 of the tensor's scale (emulator; the hardware form is test_first_optimizer_step_per_tensor_on_one_gpu)."""
    port = _free_port()
    ovr = [o for o in OVR if not o.startswith(("training.num_minibatches=", "training.update_epochs=", "training.num_envs="))] + FIRST_STEP + ["training.num_envs=16"]
    env = dict(os.environ, MPPO_TEST_HOST_DRIVEN="0", MPPO_ALLREDUCE=transport)
    _spawn(lambda r: [str(r), str(world), str(port), "1", str(tmp_path / f"r{r}.npz"), *ovr], world, env, 900)
    ranks = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    for r in range(1, world):
        np.testing.assert_array_equal(ranks[0]["adam_m"], ranks[r]["adam_m"])
    be = get_backend("emu")
    cfg = make_config({"kscale_id": "5eb3cb7f23232298", "visualization": {"camera_name": "track"}}, ovr)
    tr = be.trainer(cfg, external_random=True, use_graph=False)
    tr.reset()
    N, T, A, E, M = tr.N, tr.T, tr.A, tr.E, tr.M
    noise, _, glob = make_inputs(N, T, A, E, M, world, seed=100)
    tr.region("noise", (T, N, A))[:] = noise
    tr.region("perm", (E, T * N))[:] = glob
    tr.update()
    _assert_first_step_per_tensor(ranks[0], np.array(tr.region("adam_m")), np.array(tr.region("adam_v")), tr.O, tr.A, tr.H)
    tr.close()


def _union_on_gpu(ovr, world, updates):
    be = get_backend("hip")
    cfg = make_config({"kscale_id": "5eb3cb7f23232298", "visualization": {"camera_name": "track"}}, ovr)
    tr = be.trainer(cfg, external_random=True, use_graph=True)
    tr.reset()
    p0 = tr.params_flat()
    N, T, A, E, M = tr.N, tr.T, tr.A, tr.E, tr.M
    for u in range(updates):
        noise, _, glob = make_inputs(N, T, A, E, M, world, seed=100 + u)
        be.put(tr.region("noise", (T, N, A)), noise)
        be.put(tr.region("perm", (E, T * N)), glob)
        tr.update()
        tr._sync()
    return be, tr, p0


def _gpu_env(**kw):
    return dict(os.environ, MPPO_TEST_BACKEND="hip", MPPO_TEST_GRAPH="1", MPPO_ALLREDUCE="peer", MPPO_TEST_HOST_DRIVEN="0",
                HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("world,per_proc", [(2, 1), (4, 1), (8, 2)])
def test_first_optimizer_step_per_tensor_on_one_gpu(tmp_path, world, per_proc):
    """The tightened hardware check of the env-sharded path: after the FIRST optimizer step (one minibatch, one epoch) the ranks' Adam
    moments - linear / quadratic in the reduced gradient - equal those of one process on the union, tensor by tensor at 1e-5 of the
    tensor's scale, and the replicas are bit-identical.  world = 8 runs as four processes of two ranks each (the box allows six GPU
    processes; tests/dist_worker.py run_process_of_ranks)."""
    port = _free_port()
    ovr = ["training.num_envs=1024", *FIRST_STEP, "training.total_timesteps=100000000"]
    env = _gpu_env()
    env.pop("MPPO_PEER_MODE", None)
    if per_proc == 1:
        _spawn(lambda r: [str(r), str(world), str(port), "1", str(tmp_path / f"r{r}.npz"), *ovr], world, env, 600)
    else:
        nproc = world // per_proc
        _spawn(lambda p: ["procs", str(p), str(nproc), str(per_proc), str(port), "1", str(tmp_path / "r"), *ovr], nproc, env, 600)
    ranks = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    assert all(bool(r["graph"]) for r in ranks), "a rank did not replay its hipGraph"
    for r in range(1, world):
        np.testing.assert_array_equal(ranks[0]["params"], ranks[r]["params"])
        np.testing.assert_array_equal(ranks[0]["adam_m"], ranks[r]["adam_m"])
    be, tr, _ = _union_on_gpu(ovr, world, 1)
    _assert_first_step_per_tensor(ranks[0], be.host(tr.region("adam_m")), be.host(tr.region("adam_v")), tr.O, tr.A, tr.H)
    np.testing.assert_allclose(sum(r["losses"] for r in ranks), tr.losses(), rtol=2e-3, atol=2e-4)
    tr.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["small", "configs2"])
def test_eight_ranks_on_one_gpu(tmp_path, shape):
    """BASELINE configs[2]'s world size on the hardware there is: eight ranks on cuda:0 as four processes of two ranks (the box's process
    limit is six), the exchange inside every rank's hipGraph: ranks == one process on the union, replicas bit-identical, no wait timed out.
    "small": 8 x 512 environments, three updates of 2 x 4 optimizer steps.  "configs2": the EXACT shapes of BASELINE configs[2] -
    32 768 environments as 8 x 4096, 4 epochs x 32 minibatches of 1280 rows per rank (every rank launches what it would launch on a GPU of
    its own: the 256-tile weight-gradient launch with the eight-way slice masks, nA pieces per slice of the 250 140-float gradient) - two
    updates = 256 exchanges against one process that trains on all 32 768 environments with minibatches of 10 240 rows."""
    world, per_proc = 8, 2
    updates = 3 if shape == "small" else 2
    port = _free_port()
    ovr = (["training.num_envs=4096", "training.num_minibatches=4", "training.update_epochs=2", "training.total_timesteps=100000000"] if shape == "small" else
           ["training.num_envs=32768", "training.total_timesteps=2000000000"])
    env = _gpu_env()
    env.pop("MPPO_PEER_MODE", None)
    nproc = world // per_proc
    _spawn(lambda p: ["procs", str(p), str(nproc), str(per_proc), str(port), str(updates), str(tmp_path / "r"), *ovr], nproc, env, 600)
    ranks = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    assert all(bool(r["graph"]) for r in ranks)
    for r in range(1, world):
        np.testing.assert_array_equal(ranks[0]["params"], ranks[r]["params"])
    be, tr, p0 = _union_on_gpu(ovr, world, updates)
    single = tr.params_flat()
    step = np.abs(single - p0).max()
    diff = np.abs(ranks[0]["params"] - single)
    assert step > 1e-4 and diff.max() < 0.1 * step + 1e-7 and np.median(diff) < 1e-3 * step, (diff.max(), np.median(diff), step)
    np.testing.assert_allclose(sum(r["losses"] for r in ranks), tr.losses(), rtol=2e-3, atol=2e-4)
    tr.close()


def test_replica_check_and_flag_latency(tmp_path):
    """Round 6, first contact with a multi-GPU node hardened further: (1) the connect-time self-test is a soak (one fully checked exchange,
    then MPPO_PEER_SOAK more with every element compared on the device: 6 on the emulator, 2 000 on hardware); (2) `Trainer.peer_latencies`
    - the device-timed ping-pong of one flag between rank 0 and every peer that `bench.py --gpus N` reports as config.t_link_us - returns
    one positive figure per peer; (3) `Trainer.check_replicas` (called by make_train before every checkpoint and at the end) passes on
    bit-identical replicas and raises on EVERY rank when one rank's parameters differ by one ulp in one element."""
    world, port = 3, _free_port()
    env = dict(os.environ, MPPO_TEST_HOST_DRIVEN="0", MPPO_ALLREDUCE="peer", MPPO_TEST_LATENCY="1", MPPO_TEST_REPLICA_MISMATCH="1", MPPO_PEER_SOAK="5")
    ovr = [o if not o.startswith("training.num_envs=") else "training.num_envs=12" for o in OVR]
    _spawn(lambda r: [str(r), str(world), str(port), "2", str(tmp_path / f"r{r}.npz"), *ovr], world, env, 300)
    lat = np.load(str(tmp_path / "r0.npz") + ".latency.npy")
    assert lat.shape == (world - 1,) and (lat > 0).all()


def test_make_train_with_two_ranks(tmp_path):
    """`make_train(config)(seed)` - the host loop a user runs (minppo_amd/train.py, reference train.py:92-291) - as two env-sharded ranks:
    the peer exchange is set up by `init_comm`, every second update a checkpoint is written behind the COLLECTIVE check of the exchange
    and followed by a barrier, log lines carry statistics reduced over the ranks.  Both ranks end with bit-identical parameters and
    identical (reduced) metrics; their checkpoint files agree on the number of updates done."""
    import json

    world, port = 2, _free_port()
    env = dict(os.environ, MPPO_ALLREDUCE="peer")
    _spawn(lambda r: ["train", str(r), str(world), str(port), str(tmp_path), *OVR], world, env, 900)
    a, b = (np.load(tmp_path / f"train_r{r}.npz") for r in range(world))
    np.testing.assert_array_equal(a["params"], b["params"])
    np.testing.assert_array_equal(a["mean_reward"], b["mean_reward"])
    np.testing.assert_array_equal(a["total_loss"], b["total_loss"])
    assert len(a["mean_reward"]) == 5 and np.isfinite(a["total_loss"]).all()
    metas = []
    for r in range(world):
        with np.load(tmp_path / f"ck.npz.rank{r}") as z:
            metas.append(json.loads(z["__meta__"].tobytes().decode()))
    assert metas[0]["updates_done"] == metas[1]["updates_done"] == 5 and [m["rank"] for m in metas] == [0, 1]
