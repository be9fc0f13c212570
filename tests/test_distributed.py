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


@pytest.mark.parametrize("world,host_driven", [(2, False), (4, False), (2, True)])
def test_ranks_equal_one_process_on_the_union(tmp_path, world, host_driven):
    """world ranks through `mppo_engine_learn` (engine.hip do_learn, communicator branch) == one process on the union;
    `host_driven`: the same stages driven from Python (`Trainer.learn_host_driven`) with gloo all-reduces."""
    updates = 2
    port = _free_port()
    env = dict(os.environ, MPPO_TEST_HOST_DRIVEN="1" if host_driven else "0")
    procs = [subprocess.Popen([sys.executable, str(HERE / "dist_worker.py"), str(r), str(world), str(port), str(updates), str(tmp_path / f"r{r}.npz"), *OVR],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    ranks = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    # replicas stay bit-identical (same reduced gradient, same Adam arithmetic)
    for r in range(1, world):
        np.testing.assert_array_equal(ranks[0]["params"], ranks[r]["params"])

    be = get_backend("emu")
    cfg = make_config({"kscale_id": "5eb3cb7f23232298", "visualization": {"camera_name": "track"}}, OVR)
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
