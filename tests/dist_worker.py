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

                def failing(engine, ok_ref):
                    real(engine, ok_ref)  # (collective: the peers wait for this rank's contribution)
                    ok_ref._obj.value = 0

                tr.lib.engine_peer_selftest = failing
        got = tr.init_comm()  # $MPPO_ALLREDUCE: the engine's peer-to-peer exchange or its communicator (emulator build: shared memory either way)
        assert got == want == tr.comm_mode(), (got, want, tr.comm_mode())
        want_form = os.environ.get("MPPO_TEST_PEER_FORM")
        assert not want_form or tr.peer_form() == want_form, (tr.peer_form(), want_form)
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
    graph = tr.graph_active()
    np.savez(out_path, params=tr.params_flat(), losses=tr.losses(), reward=be.host(tr.region("reward", (T, Nl))), graph=np.array(graph))
    dist.barrier()  # nobody unmaps an exchange buffer a peer might still read
    tr.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    run_rank(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[6:], int(sys.argv[4]), sys.argv[5])
