"""Worker for tests/test_distributed.py: one rank of an env-sharded run on the CPU (emulator kernels; the engine's
collectives go through the emulator build's shared-memory communicator, the rendezvous through gloo)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def make_inputs(N, T, A, E, M, world, seed=0):
    """Noise for all N envs and a block-structured permutation: global minibatch k = union over ranks of local minibatch k."""
    rng = np.random.default_rng(seed)
    Nl = N // world
    noise = rng.standard_normal((T, N, A)).astype(np.float32)
    Bl, mbl = T * Nl, T * Nl // M
    local = np.stack([np.stack([rng.permutation(Bl) for _ in range(E)]) for _ in range(world)])  # [world, E, Bl]
    glob = np.zeros((E, T * N), np.int64)
    for e in range(E):
        rows = []
        for k in range(M):
            for r in range(world):
                li = local[r, e, k * mbl:(k + 1) * mbl]
                t, nl = li // Nl, li % Nl
                rows.append(t * N + r * Nl + nl)
        glob[e] = np.concatenate(rows)
    return noise, local.astype(np.int32), glob.astype(np.int32)


def _save(tr, be, out_path, T, Nl):
    np.savez(out_path, params=tr.params_flat(), losses=tr.losses(), reward=be.host(tr.region("reward", (T, Nl))), graph=np.array(tr.graph_active()),
             adam_m=be.host(tr.region("adam_m")), adam_v=be.host(tr.region("adam_v")))


def run_rank(rank, world, port, overrides, updates, out_path):
    import torch
    import torch.distributed as dist

    from backends import get_backend
    from minppo_amd.config import make_config

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # rendezvous on the CPU: the ranks may share one GPU, which RCCL refuses
    be = get_backend(os.environ.get("MPPO_TEST_BACKEND", "emu"))  # "hip": every rank on cuda:0 (tests/test_distributed.py, -m gpu)
    use_graph = os.environ.get("MPPO_TEST_GRAPH") == "1"
    cfg = make_config({"kscale_id": "5eb3cb7f23232298", "visualization": {"camera_name": "track"}}, overrides)
    tr = be.trainer(cfg, rank=rank, world_size=world, external_random=True, use_graph=use_graph)
    host_driven = os.environ.get("MPPO_TEST_HOST_DRIVEN") == "1"
    if not host_driven:
        want = os.environ.get("MPPO_ALLREDUCE", "peer")
        fail_rank = os.environ.get("MPPO_TEST_SELFTEST_FAIL_RANK")
        if fail_rank is not None:  # this rank's self-test of the mapped buffers reports failure: every rank must end up on the communicator
            want = "rccl"
            if int(fail_rank) == rank:
                real = tr.lib.engine_peer_selftest

                def failing(engine, stream, ok_ref):
                    real(engine, stream, ok_ref)  # (collective: the peers wait for this rank's contribution)
                    ok_ref._obj.value = 0

                tr.lib.engine_peer_selftest = failing
        got = tr.init_comm()  # $MPPO_ALLREDUCE: the engine's peer-to-peer exchange or its communicator (emulator build: shared memory either way)
        assert got == want == tr.comm_mode(), (got, want, tr.comm_mode())
        want_form = os.environ.get("MPPO_TEST_PEER_FORM")
        assert not want_form or tr.peer_form() == want_form, (tr.peer_form(), want_form)
        if os.environ.get("MPPO_TEST_LATENCY") == "1":  # the flag ping-pong between rank 0 and every peer (collective)
            lat = tr.peer_latencies(iters=40)
            assert (lat is None) == (got != "peer"), (lat, got)
            if lat is not None:
                assert len(lat) == world - 1 and all(0.0 < x < 1e6 for x in lat), lat
                np.save(out_path + ".latency.npy", np.asarray(lat))
    tr.reset()
    N, Nl, T, A, E, M = cfg.training.num_envs, tr.N, tr.T, tr.A, tr.E, tr.M

    def allreduce_sum(view):
        t = torch.from_numpy(view)
        dist.all_reduce(t)

    for u in range(updates):
        noise, local, _ = make_inputs(N, T, A, E, M, world, seed=100 + u)
        be.put(tr.region("noise", (T, Nl, A)), noise[:, rank * Nl:(rank + 1) * Nl])
        be.put(tr.region("perm", (E, T * Nl)), local[rank])
        if use_graph:
            tr.update()  # mppo_engine_update: rollout + learn replayed from the hipGraph, the exchange's kernels inside it
            tr._sync()
        else:
            tr.rollout()
            if host_driven:
                tr.learn_host_driven(allreduce_sum)  # the stage calls driven from Python with gloo collectives
            else:
                tr.learn()  # mppo_engine_learn: csrc/engine.hip do_learn, peer-to-peer or communicator branch
    tr.check_peers()
    tr.check_replicas()  # (collective) bit-identical replicas
    if os.environ.get("MPPO_TEST_REPLICA_MISMATCH") == "1":  # one rank's parameters disturbed by one ulp-sized step: EVERY rank must hear about it
        if rank == world - 1:
            p = be.host(tr.region("params")).copy()
            p[7] = np.nextafter(p[7], np.float32(1e9))
            be.put(tr.region("params"), p)
        try:
            tr.check_replicas()
            raised = False
        except RuntimeError as exc:
            raised = "replicas differ" in str(exc)
        assert raised, f"rank {rank}: check_replicas did not raise"
    _save(tr, be, out_path, T, Nl)
    dist.barrier()  # nobody unmaps an exchange buffer a peer might still read
    tr.close()
    dist.barrier()
    dist.destroy_process_group()


def run_process_of_ranks(proc, nproc, per_proc, port, overrides, updates, out_paths):
    """`per_proc` ranks in THIS process (ranks proc * per_proc ...), `nproc` such processes: how world = 8 runs on a one-GPU box that
    allows six GPU processes (hardware only, shared-GPU form of the exchange).  Same library calls as Trainer.init_comm makes - export,
    gather the handles, connect, barrier, self-test, agreement - with the ranks of a process reaching each other's exchange buffers
    through the engine's same-process registry (csrc/k_peer.hip).  A rank's self-test blocks until its peers have run theirs, so the
    ranks of a process run it from one thread each; updates are enqueued rank after rank (graph launches do not block) and awaited
    together."""
    import ctypes as C
    import threading

    import torch
    import torch.distributed as dist

    from backends import get_backend
    from minppo_amd.config import make_config

    world = nproc * per_proc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=proc, world_size=nproc)
    be = get_backend("hip")
    cfg = make_config({"kscale_id": "5eb3cb7f23232298", "visualization": {"camera_name": "track"}}, overrides)
    mine = [proc * per_proc + j for j in range(per_proc)]
    soak = os.environ.get("MPPO_TEST_SOAK") == "1"  # many updates on the engine's own random streams (no inputs written from here)
    trs = [be.trainer(cfg, rank=r, world_size=world, external_random=not soak, use_graph=True) for r in mine]
    handles = np.zeros((per_proc, 64), np.uint8)
    for j, tr in enumerate(trs):
        tr.lib.engine_peer_export(tr._engine, handles[j].ctypes.data)
    gathered = [torch.zeros(per_proc * 64, dtype=torch.uint8) for _ in range(nproc)]
    dist.all_gather(gathered, torch.from_numpy(handles.reshape(-1)))
    allh = np.concatenate([g.numpy() for g in gathered]).astype(np.uint8)  # rank order: process p holds ranks p * per_proc ...
    for tr in trs:
        tr.lib.engine_peer_connect(tr._engine, allh.ctypes.data, 1)  # every rank drives cuda:0
    dist.barrier()
    oks = [C.c_int32(0) for _ in trs]
    errs = [None] * len(trs)

    def selftest(j):
        try:
            trs[j].lib.engine_peer_selftest(trs[j]._engine, trs[j]._stream_ptr, C.byref(oks[j]))
        except Exception as exc:  # noqa: BLE001
            errs[j] = exc

    th = [threading.Thread(target=selftest, args=(j,)) for j in range(per_proc)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    good = torch.tensor([int(all(e is None for e in errs) and all(o.value == 1 for o in oks))], dtype=torch.int32)
    dist.all_reduce(good, op=dist.ReduceOp.MIN)
    assert int(good.item()) == 1, ("the self-test of the exchange failed on some rank", errs, [o.value for o in oks])
    assert all(tr.comm_mode() == "peer" and tr.peer_form() == "shared" for tr in trs), [(tr.comm_mode(), tr.peer_form()) for tr in trs]
    for tr in trs:
        tr.reset()
    N, Nl, T, A, E, M = cfg.training.num_envs, trs[0].N, trs[0].T, trs[0].A, trs[0].E, trs[0].M
    for tr in trs:  # Trainer.prepare, with ONE barrier for the process
        tr.lib.engine_prepare(tr._engine, tr._stream_ptr)
        tr._sync()
        tr._prepared = True
    dist.barrier()
    import time
    t0 = time.perf_counter()
    for u in range(updates):
        if not soak:
            noise, local, _ = make_inputs(N, T, A, E, M, world, seed=100 + u)
            for tr, r in zip(trs, mine):
                be.put(tr.region("noise", (T, Nl, A)), noise[:, r * Nl:(r + 1) * Nl])
                be.put(tr.region("perm", (E, T * Nl)), local[r])
        for tr in trs:
            tr.update()
        if not soak or (u + 1) % 8 == 0:  # (a soak keeps a few updates in flight)
            for tr in trs:
                tr._sync()
    for tr in trs:
        tr._sync()
    dt = time.perf_counter() - t0
    for tr in trs:
        tr.check_peers(collective=False)
    dist.barrier()
    for tr, path in zip(trs, out_paths):
        _save(tr, be, path, T, Nl)
    if proc == 0:
        print(f"[ranks-in-process] world {world} = {nproc} processes x {per_proc} ranks on cuda:0: {updates} updates in {dt:.2f} s "
              f"({1e3 * dt / max(updates, 1):.2f} ms per update, {N * T * updates / dt:.0f} env-steps/s)", flush=True)
    dist.barrier()
    for tr in trs:
        tr.close()
    dist.barrier()
    dist.destroy_process_group()


def run_make_train(rank, world, port, outdir, overrides):
    """`make_train(config)(seed)` as one rank of an env-sharded job (emulator kernels, gloo rendezvous): the whole host loop of
    minppo_amd/train.py with several ranks - init_comm, periodic checkpoints behind the collective check of the exchange, the barrier
    after them, reduced statistics for the log lines."""
    import torch.distributed as dist

    from backends import get_backend
    from minppo_amd import train as T
    from minppo_amd.config import make_config

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    be = get_backend("emu")
    cfg = make_config({"kscale_id": "5eb3cb7f23232298", "visualization": {"camera_name": "track"}},
                      [*overrides, f"training.checkpoint_path={outdir}/ck.npz", "training.checkpoint_every=2"])
    out = T.make_train(cfg, lib=be.lib, xp="numpy", use_graph=False, rank=rank, world_size=world)(1337, max_updates=5, log_every=1)
    flat = T.tree_to_flat(out.runner_state.train_state.params, *_dims(out))
    np.savez(f"{outdir}/train_r{rank}.npz", params=flat, mean_reward=out.metrics["mean_reward"], total_loss=out.metrics["total_loss"])
    dist.barrier()
    dist.destroy_process_group()


def _dims(out):
    t = out.runner_state.train_state.params["params"]
    k1 = t["MLP_0"]["Dense_0"]["kernel"]
    return k1.shape[0], t["log_std"].shape[0], k1.shape[1]


if __name__ == "__main__":
    if sys.argv[1] == "train":  # train <rank> <world> <port> <outdir> overrides...
        run_make_train(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6:])
    elif sys.argv[1] == "procs":  # procs <proc> <nproc> <per_proc> <port> <updates> <out prefix> overrides...
        proc, nproc, per_proc, port, updates = (int(x) for x in sys.argv[2:7])
        prefix = sys.argv[7]
        run_process_of_ranks(proc, nproc, per_proc, port, sys.argv[8:], updates, [f"{prefix}{proc * per_proc + j}.npz" for j in range(per_proc)])
    else:
        run_rank(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[6:], int(sys.argv[4]), sys.argv[5])
