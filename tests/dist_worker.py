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
    dist.init_process_group("gloo", rank=rank, world_size=world)
    be = get_backend("emu")
    cfg = make_config({"kscale_id": "5eb3cb7f23232298", "visualization": {"camera_name": "track"}}, overrides)
    tr = be.trainer(cfg, rank=rank, world_size=world, external_random=True, use_graph=False)
    host_driven = os.environ.get("MPPO_TEST_HOST_DRIVEN") == "1"
    if not host_driven:
        tr.init_comm()  # the engine's own communicator (emulator build: shared memory between the rank processes)
    tr.reset()
    N, Nl, T, A, E, M = cfg.training.num_envs, tr.N, tr.T, tr.A, tr.E, tr.M

    def allreduce_sum(view):
        t = torch.from_numpy(view)
        dist.all_reduce(t)

    for u in range(updates):
        noise, local, _ = make_inputs(N, T, A, E, M, world, seed=100 + u)
        tr.region("noise", (T, Nl, A))[:] = noise[:, rank * Nl:(rank + 1) * Nl]
        tr.region("perm", (E, T * Nl))[:] = local[rank]
        tr.rollout()
        if host_driven:
            tr.learn_host_driven(allreduce_sum)  # the stage calls driven from Python with gloo collectives
        else:
            tr.learn()  # mppo_engine_learn: csrc/engine.hip do_learn with its communicator branch
    np.savez(out_path, params=tr.params_flat(), losses=tr.losses(), reward=np.array(tr.region("reward", (T, Nl))))
    tr.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    run_rank(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[6:], int(sys.argv[4]), sys.argv[5])
