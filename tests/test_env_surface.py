"""`minppo.env` surface kept by minppo_amd.env (GPU) and the stub / CLI modules (CPU)."""
import pickle
import sys

import numpy as np
import pytest

from minppo_amd.config import make_config

BASE = {"kscale_id": "5eb3cb7f23232298", "visualization": {"camera_name": "track"}}


def test_cli_follows_the_reference(tmp_path, monkeypatch):
    """`minppo {train,env,infer}` (reference cli.py:12-26); `infer` is a stub upstream too (infer.py:22-27)."""
    from minppo_amd import cli

    monkeypatch.setattr(sys, "argv", ["minppo", "infer", "stompy_pro"])
    with pytest.raises(NotImplementedError):
        cli.main()
    monkeypatch.setattr(sys, "argv", ["minppo", "bogus"])
    with pytest.raises(SystemExit):
        cli.main()


@pytest.mark.gpu
def test_humanoid_env_matches_env_oracle_on_gpu():
    import torch

    from minppo_amd.env import EnvState, HumanoidEnv
    from oracle.env_oracle import EnvOracle, RewardCfg

    cfg = make_config(BASE, ["reward.height_min_z=0.95"])
    env = HumanoidEnv(cfg)
    assert env.observation_size == 225 and env.action_size == 10 and env.dt == pytest.approx(0.002)
    assert env.actuator_ctrlrange.shape == (10, 2) and env.initial_qpos.shape == (17,)
    N = 5
    es = env.reset(num_envs=N)
    assert isinstance(es, EnvState) and es.obs.shape == (N, 225) and not es.done.any()
    orc = EnvOracle(env.cm.t, RewardCfg(height_min_z=0.95))
    eo = orc.reset(N)
    np.testing.assert_allclose(es.obs.cpu().numpy(), eo["obs"], atol=1e-5)
    rng = np.random.default_rng(0)
    for t in range(3):
        a = (0.5 * rng.standard_normal((N, 10))).astype(np.float32)
        es2 = env.step(es, torch.from_numpy(a).cuda())
        eo = orc.step(eo, a.astype(np.float64))
        assert (es2.done.cpu().numpy() == eo["done"]).all()
        np.testing.assert_allclose(es2.reward.cpu().numpy(), eo["reward"], atol=0.3)
        # host-side restatement of compute_reward on the records agrees with the kernel's reward
        np.testing.assert_allclose(env.compute_reward(es.pipeline_state, es2.pipeline_state, torch.from_numpy(a).cuda()).cpu().numpy(),
                                   es2.reward.cpu().numpy(), atol=2e-3)
        assert (env.is_done(es2.pipeline_state).cpu().numpy() == False).all()  # noqa: E712  (auto-reset already applied)
        assert int(es2.metrics.timestep[0]) == t + 1
        es = es2
    with pytest.raises(ValueError, match="shape"):
        env.step(es, torch.zeros(N, 3).cuda())
    env.close()
