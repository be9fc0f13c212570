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
    assert flat.size == 250140  # P = 512*O + 258*A + 132353 = 250133 parameters + 7 alignment words of the flat layout
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


def test_make_train_returns_the_references_metrics_when_asked():
    """`training.keep_metrics_history=true`: `TrainOutput.metrics` is what the reference returns (train.py:283,287-289) - an `EnvMetrics` whose six
    fields are [num_updates, T, N] arrays a caller can index `out.metrics.episode_returns[u, t, n]`; refused (ValueError) when the history
    would exceed its byte budget.  The default stays the per-update reductions."""
    from minppo_amd.env import EnvMetrics

    be = get_backend("emu")
    train = T.make_train(make_config(BASE, SMALL + ["training.keep_metrics_history=true"]), lib=be.lib, xp="numpy", use_graph=False)
    out = train(1337)
    assert isinstance(out.metrics, EnvMetrics)
    for f in EnvMetrics._fields:
        assert getattr(out.metrics, f).shape == (3, 2, 8), f  # num_updates = 48 // 2 // 8, T = 2, N = 8
    np.testing.assert_array_equal(out.metrics.timestep[:, :, 0].reshape(-1), np.arange(1, 7))  # env.py:191: one more per step, carried over the updates
    assert (out.metrics.episode_lengths == out.metrics.timestep).all() and not out.metrics.returned_episode.any()  # nobody falls in six steps
    np.testing.assert_allclose(out.metrics.episode_returns[2, 1], out.metrics.episode_returns[2, 0] + (out.metrics.episode_returns[2, 1] - out.metrics.episode_returns[2, 0]))
    assert np.isfinite(out.metrics.episode_returns).all() and (np.abs(out.metrics.episode_returns[-1, -1]) > 0).any()
    with pytest.raises(ValueError, match="keep_metrics_history"):
        T.make_train(make_config(BASE, ["training.keep_metrics_history=true", "training.num_envs=4096", "training.total_timesteps=10000000000"]))


def _resume_case(be, tmp_path, trainer_kw):
    """4 updates in one go == 2 updates, checkpoint, NEW trainer, resume, 2 updates - bit for bit (parameters, Adam moments,
    counters, environment states, episode metrics); the engine's own RNG streams are positioned by the restored counters."""
    over = ["training.num_envs=8", "training.num_steps=4", "rl.num_env_steps=4", "training.num_minibatches=2", "training.update_epochs=2",
            "model.hidden_size=32", "training.total_timesteps=100000"] if be.name == "emu" else \
           ["training.num_envs=256", "training.num_minibatches=8", "training.update_epochs=2", "training.total_timesteps=100000000"]
    cfg = make_config(BASE, over)
    ck = str(tmp_path / "ck" / "state.npz")
    a = be.trainer(cfg, **trainer_kw)
    a.reset()
    for _ in range(2):
        a.update()
    a.save_checkpoint(ck)
    for _ in range(2):
        a.update()
    names = T.Trainer._CKPT_REGIONS
    a._sync()  # the engine runs on its own stream: wait before reading the arena
    want = {n: a._to_host(a.region(n)).copy() for n in names}
    a.close()
    b = be.trainer(cfg, **trainer_kw)
    b.load_checkpoint(ck)
    assert b.updates_done == 2
    for _ in range(2):
        b.update()
    b._sync()
    for n in names:
        np.testing.assert_array_equal(b._to_host(b.region(n)), want[n], err_msg=n)
    assert int(b._to_host(b.region("count"))[1]) == 4
    b.close()
    # a checkpoint of another shape / seed is refused
    c = be.trainer(make_config(BASE, over + ["training.seed=7"]), **trainer_kw)
    with pytest.raises(ValueError, match="seed"):
        c.load_checkpoint(ck)
    c.close()
    # ... and so is one written with the other random-stream implementation (its "jax_rng" region would be all zeros)
    d = be.trainer(make_config(BASE, over + ["training.rng_impl=threefry"]), **trainer_kw)
    with pytest.raises(ValueError, match="rng_impl"):
        d.load_checkpoint(ck)
    d.close()


def test_infer_surface(tmp_path):
    """`minppo.infer.load_model` (reference infer.py:17-19) reads what `save_model` wrote; `main` is a stub upstream too."""
    from minppo_amd import infer

    tree = {"params": {"log_std": np.zeros(3, np.float32)}}
    T.save_model(tree, str(tmp_path / "sub" / "m.pkl"))
    got = infer.load_model(str(tmp_path / "sub" / "m.pkl"))
    np.testing.assert_array_equal(got["params"]["log_std"], tree["params"]["log_std"])
    with pytest.raises(NotImplementedError):
        infer.main([])


def test_checkpoint_resume_is_bit_exact(tmp_path):
    _resume_case(get_backend("emu"), tmp_path, dict(use_graph=False))


@pytest.mark.gpu
def test_checkpoint_resume_is_bit_exact_on_gpu_with_hipgraph(tmp_path):
    _resume_case(get_backend("hip"), tmp_path, dict(use_graph=True))


def test_make_train_checkpoints_and_resumes(tmp_path):
    be = get_backend("emu")
    ck = tmp_path / "run.npz"
    common = SMALL[:-1] + ["training.total_timesteps=64", f"training.model_save_path={tmp_path / 'm.pkl'}", f"training.checkpoint_path={ck}"]
    full = T.make_train(make_config(BASE, common), lib=be.lib, xp="numpy", use_graph=False)(1337)          # 4 updates
    T.make_train(make_config(BASE, common), lib=be.lib, xp="numpy", use_graph=False)(1337, max_updates=2)  # stops after 2, leaves a checkpoint
    rest = T.make_train(make_config(BASE, common + [f"training.resume_from={ck}"]), lib=be.lib, xp="numpy", use_graph=False)(1337)
    assert rest.runner_state.train_state.step == full.runner_state.train_state.step
    np.testing.assert_array_equal(T.tree_to_flat(rest.runner_state.train_state.params, 225, 10, 16), T.tree_to_flat(full.runner_state.train_state.params, 225, 10, 16))


def test_param_tree_follows_num_layers():
    """`MLP([hidden_size] * num_layers + [out])` (reference train.py:79,82): Dense_0 .. Dense_{num_layers} per network."""
    for L in (1, 3):
        flat = T.init_flat_params(7, 21, 4, 16, L)
        tree = T.flat_to_tree(flat, 21, 4, 16, L)
        assert sorted(tree["params"]["MLP_0"]) == [f"Dense_{i}" for i in range(L + 1)]
        assert tree["params"]["MLP_0"]["Dense_0"]["kernel"].shape == (21, 16) and tree["params"]["MLP_0"][f"Dense_{L}"]["kernel"].shape == (16, 4)
        assert tree["params"]["MLP_1"][f"Dense_{L}"]["kernel"].shape == (16, 1)
        if L > 1:
            assert tree["params"]["MLP_1"]["Dense_1"]["kernel"].shape == (16, 16)
        np.testing.assert_array_equal(T.tree_to_flat(tree, 21, 4, 16, L), flat)
        # same flat layout as the oracle's
        from oracle import ppo_oracle as po
        assert {k: v for k, v in T.param_slices(21, 4, 16, L)[0].items()} == po.param_slices(21, 4, 16, L) and flat.size == po.flat_size(21, 4, 16, L)
