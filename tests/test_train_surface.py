"""The `minppo.train` surface kept by minppo_amd.train: make_train / TrainOutput / save_model / param tree."""
import pickle

import numpy as np
import pytest

from backends import get_backend
from minppo_amd import train as T
from minppo_amd.config import make_config

BASE = {"kscale_id": "5eb3cb7f23232298", "visualization": {"camera_name": "track"}}
SMALL = ["training.num_envs=8", "training.num_steps=2", "rl.num_env_steps=2", "training.num_minibatches=2", "training.update_epochs=1",
         "model.hidden_size=16", "training.total_timesteps=48"]


def test_param_tree_layout_and_init():
    flat = T.init_flat_params(1337, 225, 10, 256)
    assert flat.size == 250133  # P = 512*O + 258*A + 132353
    tree = T.flat_to_tree(flat, 225, 10, 256)
    p = tree["params"]
    assert set(p) == {"MLP_0", "MLP_1", "log_std"} and set(p["MLP_0"]) == {"Dense_0", "Dense_1", "Dense_2"}
    assert p["MLP_0"]["Dense_0"]["kernel"].shape == (225, 256) and p["MLP_0"]["Dense_2"]["kernel"].shape == (256, 10)
    assert p["MLP_1"]["Dense_2"]["kernel"].shape == (256, 1) and p["log_std"].shape == (10,)
    w = p["MLP_0"]["Dense_1"]["kernel"].astype(np.float64)
    np.testing.assert_allclose(w.T @ w, 2 * np.eye(256), atol=1e-5)  # orthogonal, gain sqrt(2)  (train.py:63)
    w3 = p["MLP_1"]["Dense_2"]["kernel"].astype(np.float64)
    np.testing.assert_allclose(w3.T @ w3, [[1e-4]], rtol=1e-5)  # gain 0.01 on both heads (train.py:68)
    assert all((p[m][f"Dense_{i}"]["bias"] == 0).all() for m in ("MLP_0", "MLP_1") for i in range(3)) and (p["log_std"] == 0).all()
    np.testing.assert_array_equal(T.tree_to_flat(tree, 225, 10, 256), flat)
    with pytest.raises(ValueError, match="shape"):
        bad = T.flat_to_tree(flat, 225, 10, 256); bad["params"]["log_std"] = np.zeros(3); T.tree_to_flat(bad, 225, 10, 256)


def test_make_train_runs_and_saves(tmp_path):
    be = get_backend("emu")
    cfg = make_config(BASE, SMALL + [f"training.model_save_path={tmp_path / 'sub' / 'm.pkl'}"])
    train = T.make_train(cfg, lib=be.lib, xp="numpy", use_graph=False)
    out = train(1337, log_every=1)
    assert isinstance(out, T.TrainOutput)
    ts = out.runner_state.train_state
    assert ts.step == 3 * 2  # num_updates = 48 // 2 // 8 = 3 (train.py:93), 2 optimizer steps each
    assert out.runner_state.last_obs.shape == (8, 225)
    assert len(out.metrics["mean_reward"]) == 3 and np.isfinite(out.metrics["total_loss"]).all()
    T.save_model(ts.params, cfg.training.model_save_path)
    with open(cfg.training.model_save_path, "rb") as f:
        back = pickle.load(f)
    np.testing.assert_array_equal(back["params"]["MLP_0"]["Dense_0"]["kernel"], ts.params["params"]["MLP_0"]["Dense_0"]["kernel"])
    # a 2-word key is accepted in place of an int seed
    assert T._seed_from_rng(np.array([0, 1337], np.uint32)) != T._seed_from_rng(np.array([0, 1338], np.uint32))
    with pytest.raises(ValueError, match="batch_size"):
        T.make_train(make_config(BASE, ["training.num_envs=6", "training.num_minibatches=4", "training.num_steps=1"]))
