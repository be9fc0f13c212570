"""Engine with training.rng_impl=threefry, eager then hipGraph: every epoch's index array must be a permutation (progress lines are flushed: a fault shows where)."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1])); sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
from backends import HipBackend
from minppo_amd.config import make_config
be = HipBackend()
BASE = {"kscale_id": "5eb3cb7f23232298", "visualization": {"camera_name": "track"}}
for graph in (False, True):
    for envs in (256, 4096):
        cfg = make_config(BASE, [f"training.num_envs={envs}", "training.num_minibatches=8", "training.update_epochs=2", "training.total_timesteps=100000000", "training.rng_impl=threefry"])
        tr = be.trainer(cfg, use_graph=graph)
        tr.reset()
        for u in range(3):
            tr.update(); tr._sync()
            perm = be.host(tr.region("perm", (tr.E, tr.T * tr.N)))
            ok = all((np.sort(perm[e]) == np.arange(tr.T * tr.N)).all() for e in range(tr.E))
            print(f"graph={graph} envs={envs} update {u}: permutations valid {ok}, graph active {tr.graph_active()}", flush=True)
        tr.close()
