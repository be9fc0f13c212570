"""The C++ / OpenMP float32 twin of the environment step (oracle/cpu_twin: the CPU baseline of bench.py) against the committed fixtures and
the NumPy oracle - the same checks, at the same tolerances, that pin the HIP kernel (tests/test_golden.py, tests/test_kernels_physics.py)."""
from pathlib import Path

import numpy as np

from minppo_amd.model import load_model
from oracle.cpu_twin import RewardCfg as TwinReward, Twin
from oracle.env_oracle import EnvOracle, RewardCfg

G = Path(__file__).parent / "golden"
f32 = np.float32


def test_twin_reproduces_the_physics_golden():
    g = np.load(G / "physics_steps.npz")
    cm = load_model("synth_stompy_pro")
    tw = Twin(cm, reward=TwinReward(float(g["height_min_z"]), 2.0, 2.0, 0.2, 0.5, 0.1, 4.0, 1.0, 1.25))
    K, N = g["qpos"].shape[:2]
    O, OP, R, nv, nq = tw.obs_dim, tw.obs_pad, tw.rec_dim, cm.nv, cm.nq
    tw.reset(N)
    for t in range(K):
        rec = np.zeros((N, R), f32)
        rec[:, :nq] = g["qpos"][t]; rec[:, nq:nq + nv] = g["qvel"][t]
        rec[:, nq + nv:nq + nv + 110] = g["cinert"][t][:, 1:].reshape(N, -1)
        rec[:, nq + nv + 110:nq + nv + 176] = g["cvel"][t][:, 1:].reshape(N, -1)
        rec[:, nq + nv + 176:O] = g["qact"][t]
        rec[:, OP:OP + nv] = g["warm"][t]; rec[:, OP + nv] = g["comx"][t]; rec[:, OP + nv + 1] = g["time"][t]
        tw.state[...] = rec
        obs, rew, done = tw.step(g["action"][t].astype(f32))
        assert (done.astype(bool) == g["done"][t]).all()
        np.testing.assert_allclose(obs[:, :O], g["obs"][t], atol=1e-4)
        np.testing.assert_allclose(rew, g["reward"][t], atol=1e-2)
        np.testing.assert_allclose(tw.state[:, :nq], g["qpos1"][t], atol=2e-3)
    assert g["done"].any()
    tw.close()


def test_twin_follows_the_oracle_over_a_rollout():
    """Both BASELINE robots, a free-running rollout from the reset state with the same actions: the reset observation to 1e-5, rewards and
    the stepped state within the float32-solver envelope the kernel is held to, `done` exactly."""
    for name, steps in (("synth_stompy_pro", 12), ("synth_stompy_full", 6)):
        cm = load_model(name)
        tw = Twin(cm)
        env = EnvOracle(cm.t, RewardCfg())
        N = 16
        obs0 = tw.reset(N).copy()
        es = env.reset(N)
        np.testing.assert_allclose(obs0[:, :tw.obs_dim], es["obs"], atol=1e-5)
        rng = np.random.default_rng(3)
        for t in range(steps):
            a = 0.5 * rng.standard_normal((N, cm.nu))
            es = env.step(es, a)
            obs, rew, done = tw.step(a.astype(f32))
            assert (done.astype(bool) == es["done"]).all()
            np.testing.assert_allclose(rew, es["reward"], atol=0.05)
            q = es["pipeline_state"].qpos
            assert np.abs(tw.state[:, :cm.nq] - q).max() < 5e-3, (name, t, np.abs(tw.state[:, :cm.nq] - q).max())
        tw.close()


def test_torch_ppo_reproduces_the_ppo_golden():
    """oracle/cpu_twin/ppo_torch.py (the PPO half of the CPU baseline: torch-CPU float32, autograd) against the float64 fixture that pins
    the NumPy oracle and the kernels: GAE, the losses of all E x M minibatch steps, the parameters and first moments after them."""
    import torch

    from oracle import ppo_oracle as po
    from oracle.cpu_twin import ppo_torch as pt

    g = np.load(G / "ppo_small.npz")
    N, T, O, A, H, M, E = [int(x) for x in g["dims"]]
    named = po.flat_to_named(g["params0"], O, A, H)
    p = {k: torch.tensor(np.asarray(v, f32)) for k, v in named.items()}
    tt = lambda k: torch.tensor(np.asarray(g[k], f32))
    adv, tgt = pt.gae(torch.tensor(g["done"]), tt("value"), tt("reward"), tt("last_val"), 0.99, 0.95)
    np.testing.assert_allclose(adv.numpy(), g["adv"], rtol=1e-5, atol=1e-5)
    opt = pt.Adam(p, 3e-4, 0.5, True, N * T // M, E, 1000)
    traj = {k: tt(k) for k in ("obs", "action", "value", "log_prob")}
    losses = pt.update_epochs(p, opt, traj, adv, tgt, torch.tensor(g["perms"]), M, dict(clip_eps=0.2, vf_coef=0.5, ent_coef=0.0))
    np.testing.assert_allclose(losses, g["losses"].reshape(-1, 4), rtol=2e-4, atol=1e-5)
    flat1 = po.named_to_flat({k: v.numpy().astype(np.float64) for k, v in p.items()}, O, A, H)
    step = np.abs(g["params1"] - g["params0"]).max()
    assert np.abs(flat1 - g["params1"]).max() < 1e-3 * step
    m1 = po.named_to_flat({k: v.numpy().astype(np.float64) for k, v in opt.m.items()}, O, A, H)
    np.testing.assert_allclose(m1, g["adam_m"], rtol=1e-3, atol=1e-7)


def test_twin_follows_the_oracle_on_the_other_collider_kinds():
    """The model classes beyond the BASELINE robots: geom-geom pairs (sphere / capsule self-collision candidates), colliding free bodies,
    and MJCF robots with box and mesh (convex hull) colliders - the twin picks the same hull vertices, slot by slot, or `done` / rewards diverge."""
    import os

    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for name, steps, amp in (("synth_stompy_pro_sc", 8, 0.6), ("synth_tumblers", 20, 0.0), ("synth_can", 60, 0.3), ("synth_pile", 30, 0.0), ("synth_stompy_frames", 6, 0.4), (os.path.join(here, "hand_leg.xml"), 25, 0.3),
                             (os.path.join(here, "mesh_foot.xml"), 40, 0.3),
                             (os.path.join(here, "export_biped", "robot.xml"), 12, 0.3)):  # 28 bodies, 33 dofs, export-style multi-file MJCF
        cm = load_model(name)
        tw = Twin(cm)
        env = EnvOracle(cm.t, RewardCfg())
        N = 8
        obs0 = tw.reset(N).copy()
        es = env.reset(N)
        np.testing.assert_allclose(obs0[:, :tw.obs_dim], es["obs"], atol=1e-5)
        rng = np.random.default_rng(9)
        worst = 0.0
        for t in range(steps):
            a = amp * rng.standard_normal((N, cm.nu))
            # re-seed the twin from the oracle's state every step (identical inputs): qpos, qvel, warm start
            s = es["pipeline_state"]
            tw.state[:, :cm.nq] = s.qpos
            tw.state[:, cm.nq:cm.nq + cm.nv] = s.qvel
            tw.state[:, tw.obs_pad:tw.obs_pad + cm.nv] = s.qacc_warmstart
            tw.state[:, tw.obs_pad + cm.nv] = s.subtree_com[:, 1, 0]
            es = env.step(es, a)
            obs, rew, done = tw.step(a.astype(f32))
            assert (done.astype(bool) == es["done"]).all(), (name, t)
            np.testing.assert_allclose(rew, es["reward"], atol=2e-2, err_msg=f"{name} step {t}")
            worst = max(worst, float(np.abs(tw.state[:, :cm.nq] - es["pipeline_state"].qpos).max()))
        assert worst < (1.2e-2 if "export_biped" in name else 2e-3), (name, worst)  # (the biped's light end bodies: tests/test_kernels_physics.py)
        tw.close()
