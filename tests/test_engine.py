"""The engine (`mppo_engine_*`: one `_update_step`, reference minppo/train.py:146-283) against the oracle.

Stage-wise "identical inputs" parity: noise and permutations are written into the arena; the physics is
chaotic at float32 (see test_kernels_physics.py), so the PPO half is checked on the engine's OWN
trajectory, and the rollout is checked step-by-step against the env oracle driven by the engine's actions."""
import numpy as np
import pytest

from minppo_amd import _native as nat
from minppo_amd.config import make_config
from oracle import ppo_oracle as po
from oracle.env_oracle import EnvOracle, RewardCfg, default_hp
from test_kernels_ppo import _host_permutation

BASE = {"kscale_id": "5eb3cb7f23232298", "visualization": {"camera_name": "track"}}


def _cfg(*over):
    return make_config(BASE, list(over))


def _small(be):
    if be.name == "emu":
        return ["training.num_envs=8", "training.num_steps=4", "rl.num_env_steps=4", "training.num_minibatches=2", "training.update_epochs=2",
                "model.hidden_size=32", "training.total_timesteps=100000"]
    return ["training.num_envs=256", "training.num_minibatches=8", "training.update_epochs=2", "training.total_timesteps=100000000"]


def _random_engine_config(seed, emu):
    """A random configuration of the whole engine (tests of the hand-picked ones sit on known edges): environments, rollout length, minibatch
    count (a divisor of the batch), epochs, hidden width, depth, robot, observation kind (float products, tanh, one frame per step: what the
    oracle side of the test below is written for)."""
    rng = np.random.default_rng(4000 + seed)
    T = int(rng.integers(1, 5 if emu else 12))
    N = int(rng.integers(6, 16 if emu else 700))   # (the free-running physics comparison below is a statistic over environments)
    div = [m for m in range(1, 17) if (T * N) % m == 0]
    M = int(rng.choice(div))
    layers = int(rng.choice([1, 2, 2, 2, 3]))
    H = int(rng.choice([32, 64] if emu else [32, 64, 96, 128, 160, 256, 40, 200]))
    over = [f"training.num_envs={N}", f"training.num_steps={T}", f"rl.num_env_steps={T}", f"training.num_minibatches={M}", f"training.update_epochs={int(rng.integers(1, 4))}",
            f"model.hidden_size={H}", f"model.num_layers={layers}", "training.total_timesteps=100000000"]
    if rng.random() < 0.3:
        over.append("environment.model=synth_stompy_full")
    if rng.random() < 0.25:
        over.append("environment.include_c_vals=false")
    return over


import os

_FUZZ = [f"fuzz{k}" for k in range(int(os.environ.get("MPPO_FUZZ_ENGINES", "3")))]   # (MPPO_FUZZ_ENGINES=60 on the GPU: DESIGN.md section 5)


@pytest.mark.parametrize("robot", ["stompy_pro", "stompy_full", "stompy_pro_no_c_vals", "stompy_pro_1_layer", "stompy_pro_3_layers", "stompy_pro_ragged", "row_tiles_32"] + _FUZZ)
def test_update_matches_oracle_stage_by_stage(be, robot):
    """Both BASELINE robots: configs[1] (synth_stompy_pro, O = 225, A = 10) and configs[4] (synth_stompy_full, O = 415, A = 20);
    the short observation of `environment.include_c_vals=false` (qpos, qvel, qfrc_actuator: O = 49; reference env.py:254-259);
    and `model.num_layers` = 1 / 3 (reference config.py:53, train.py:79,82; the layer-wise kernels)."""
    cfg = _cfg(*(_small(be) if not robot.startswith("fuzz") else _random_engine_config(int(robot[4:]), be.name == "emu")),
               *(["environment.model=synth_stompy_full"] if robot == "stompy_full" else []),
               *(["environment.include_c_vals=false"] if robot == "stompy_pro_no_c_vals" else []),
               *(["model.num_layers=1"] if robot == "stompy_pro_1_layer" else []), *(["model.num_layers=3"] if robot == "stompy_pro_3_layers" else []),
               # minibatches of 9 rows (emulator) / 190 rows (GPU): not a multiple of the 4-row quads nor of the 16-row tiles - the pre-gathered
               # row buffers and the weight-gradient operands end in zero rows, and a minibatch starts in the middle of a quad of the permutation
               *((["training.num_envs=6", "training.num_steps=3", "rl.num_env_steps=3"] if be.name == "emu" else ["training.num_envs=152"]) if robot == "stompy_pro_ragged" else []),
               # the 32-row form of the float row pass (two 16-row tiles per workgroup share every weight stage; taken when the 16-row tiling has more
               # workgroups than the chip has CUs): on the GPU BASELINE configs[4]'s minibatch shape - 2560 rows of the 20-actuator robot, 2 x 160
               # workgroups of 16 rows -> 2 x 80 of 32; on the emulator (threshold 4 workgroups) 64-row minibatches
               *((["training.num_envs=32"] if be.name == "emu" else ["environment.model=synth_stompy_full", "training.num_envs=2048"]) if robot == "row_tiles_32" else []))
    tr = be.trainer(cfg, external_random=True, use_graph=False)
    if robot == "row_tiles_32":
        import ctypes as C_

        rows = C_.c_int32(0)
        be.lib.minibatch_rows_per_workgroup(C_.byref(tr.net), tr.N * tr.T // tr.M, 1, C_.byref(rows))
        assert rows.value == 32, rows.value
    tr.reset()
    N, T, A, H, O, OP, E, M = tr.N, tr.T, tr.A, tr.H, tr.O, tr.OP, tr.E, tr.M
    rng = np.random.default_rng(0)
    env = EnvOracle(tr.cm.t, RewardCfg(), include_c_vals=cfg.environment.include_c_vals)
    assert O == env.observation_size
    hp = default_hp(cfg)
    p = tr.params_flat().astype(np.float64)
    opt = po.OptState(np.zeros_like(p), np.zeros_like(p), 0)
    for u in range(2):
        noise = rng.standard_normal((T, N, A)).astype(np.float32)
        perms = np.stack([rng.permutation(N * T) for _ in range(E)]).astype(np.int32)
        be.put(tr.region("noise", (T, N, A)), noise)
        be.put(tr.region("perm", (E, N * T)), perms)
        tr.rollout()
        tr._sync()
        tj = {k: be.host(v).copy() for k, v in tr.traj().items()}
        obs = tj["obs"][:, :, :O].astype(np.float64)
        named = po.flat_to_named(p, O, A, H)
        # policy on the engine's observations (every t), sample + log-prob with the given noise
        for t in range(T):
            mean, ls, val = po.actor_critic_forward(named, obs[t])
            act = po.mvn_sample(mean, ls, noise[t].astype(np.float64))
            np.testing.assert_allclose(tj["action"][t], act, atol=5e-5)
            np.testing.assert_allclose(tj["value"][t], val, atol=5e-5)
            np.testing.assert_allclose(tj["log_prob"][t], po.mvn_log_prob(act, mean, ls), atol=2e-4)
        _, _, lv = po.actor_critic_forward(named, obs[T])
        np.testing.assert_allclose(tj["last_val"], lv, atol=5e-5)
        # environment: the oracle driven by the engine's actions tracks the engine's rollout
        if u == 0:
            es = env.reset(N)
            np.testing.assert_allclose(obs[0], es["obs"], atol=1e-5)
            for t in range(T):
                es = env.step(es, tj["action"][t].astype(np.float64))
                assert (tj["done"][t].astype(bool) == es["done"]).all()
                # obs(t+1) shows state t (one-step lag): exact-to-rounding at t = 0.  From then on the two trajectories run
                # FREELY (float32 kernel, float64 oracle, same actions) and separate at the rate the unconverged solver
                # allows (test_kernels_physics.py).  Measured for float32 vs float64 of the oracle itself, 256 envs,
                # unit-variance actions: max |dqvel| 0.13 at t = 0 growing to 0.7 .. 1.0 at t = 9, median over envs of the
                # per-env max 1e-3 -> 1.6e-2 (the MI355X kernel, with its fast reciprocal square roots: 3e-2), |dqpos| <= 4.4e-3 at t = 9,
                # |dreward| <= 1.6e-2.  Bounds = that envelope x 1.5 .. 2:
                if t == 0:
                    np.testing.assert_allclose(obs[1], es["obs"], atol=1e-4)
                else:
                    nq, nv = tr.cm.nq, tr.cm.nv
                    np.testing.assert_allclose(obs[t + 1][:, :nq], es["obs"][:, :nq], atol=1e-3 * t)
                    dv = np.abs(obs[t + 1][:, nq:nq + nv] - es["obs"][:, nq:nq + nv]).max(1)
                    assert dv.max() <= 0.15 * (t + 1) and np.median(dv) <= 0.06, (t, dv.max(), np.median(dv))
                np.testing.assert_allclose(tj["reward"][t], es["reward"], atol=0.05)
        # GAE on the engine's trajectory
        adv, tgt = po.calculate_gae(tj["done"].astype(bool), tj["value"].astype(np.float64), tj["reward"].astype(np.float64), tj["last_val"].astype(np.float64),
                                    cfg.rl.gamma, cfg.rl.gae_lambda)
        np.testing.assert_allclose(tj["adv"], adv, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(tj["target"], tgt, rtol=1e-4, atol=1e-4)
        # E x M optimizer steps
        tr.learn()
        traj = dict(obs=obs[:T], action=tj["action"].astype(np.float64), value=tj["value"].astype(np.float64), log_prob=tj["log_prob"].astype(np.float64))
        p_new, opt, losses = po.update_epochs_on_batch(p, opt, traj, adv, tgt, perms, O=O, A=A, H=H, num_minibatches=M, hp=hp)
        step = np.abs(p_new - p).max()
        got = tr.params_flat()
        assert np.abs(got - p_new).max() < 2e-2 * step + 1e-7, (np.abs(got - p_new).max(), step)
        np.testing.assert_allclose(tr.losses(), losses, rtol=2e-3, atol=2e-4)
        cnt = be.host(tr.region("count"))
        assert cnt[0] == (u + 1) * E * M and cnt[1] == u + 1
        # carried last_obs: slot 0 of the next rollout is slot T of this one
        np.testing.assert_array_equal(be.host(tr.region("obs", (T + 1, N, OP)))[0], tj["obs"][T])
        p = got.astype(np.float64)  # continue from the engine's parameters (identical inputs for the next update)
        opt = po.OptState(be.host(tr.region("adam_m")).astype(np.float64), be.host(tr.region("adam_v")).astype(np.float64), opt.count)
    st = tr.rollout_stats()
    assert np.isfinite(st["mean_reward"]) and st["done_fraction"] == 0.0 and st["episodes"] == 0 and st["mean_episode_return"] == 0.0
    np.testing.assert_allclose(st["mean_reward"], tj["reward"].astype(np.float64).mean(), rtol=1e-6)
    tr.close()


def test_rollout_statistics_match_the_env_metrics_history(be):
    """the per-update reduction of `EnvMetrics` (reference env.py:183-194, stacked into `Memory.info` at
    train.py:170 and returned at :283).  The height window is set so tightly (the standing robot bobs by ~1 mm) that episodes end inside the rollouts; the reduced
    sums must equal the same sums over the [T, N] history the env oracle's bookkeeping produces from the engine's own
    reward / done arrays, across two consecutive rollouts (episodes spanning an update boundary carry over), and the
    per-environment metrics the kernel keeps must agree with that history at the end."""
    cfg = _cfg(*_small(be), "reward.height_max_z=1.0101")
    tr = be.trainer(cfg, external_random=True, use_graph=False)
    tr.reset()
    N, T, A, E = tr.N, tr.T, tr.A, tr.E
    rng = np.random.default_rng(3)
    ep_ret, ep_len = np.zeros(N, np.float32), np.zeros(N, np.int64)
    total_done = 0
    for u in range(3):
        be.put(tr.region("noise", (T, N, A)), (3.0 * rng.standard_normal((T, N, A))).astype(np.float32))
        be.put(tr.region("perm", (E, N * T)), np.stack([rng.permutation(N * T) for _ in range(E)]).astype(np.int32))
        tr.rollout()
        tr._sync()
        rew, done = be.host(tr.region("reward", (T, N))).copy(), be.host(tr.region("done", (T, N))).astype(bool)
        s_ret = s_len = 0.0
        for t in range(T):  # env.py:183-194 on the engine's reward / done history
            ep_ret = ep_ret + rew[t]
            ep_len = ep_len + 1
            s_ret += float(ep_ret[done[t]].astype(np.float64).sum())
            s_len += float(ep_len[done[t]].sum())
            ep_ret = np.where(done[t], np.float32(0), ep_ret).astype(np.float32)
            ep_len = np.where(done[t], 0, ep_len)
        st = tr.rollout_stats()
        nd = int(done.sum())
        total_done += nd
        assert st["episodes"] == nd
        np.testing.assert_allclose(st["mean_reward"], rew.astype(np.float64).mean(), rtol=1e-6)
        if nd:
            np.testing.assert_allclose(st["mean_episode_return"], s_ret / nd, rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(st["mean_episode_length"], s_len / nd, rtol=1e-6)
        np.testing.assert_array_equal(be.host(tr.region("episode_lengths")), ep_len)
        np.testing.assert_allclose(be.host(tr.region("episode_returns")), ep_ret, rtol=1e-6, atol=1e-6)
        raw = be.host(tr.region("rollout_stats"))
        st2 = tr.rollout_stats()
        assert st == st2 and raw[6] == 0 and raw[7] == 0
        tr.learn()
    assert total_done >= 2, "the test must see episodes end (raise the noise or the height threshold)"
    tr.close()


def test_a_permutation_bucket_overflow_is_surfaced(be):
    """Round-4 advisor: the two-launch permutation (csrc/k_perm.hip) records a bucket overflow in a word that the end-of-update kernel zeroes
    again; nothing read it.  Now that kernel makes it sticky in count[3] first and `Trainer.check_status()` raises.  The overflow itself is
    unreachable (22 sigma at B <= 131072), so the mark is planted by hand where the scatter kernel would set it.  (The emulator build
    replaces the two-launch form by a host sort - tests/emu/emu_stubs.cpp - so this is a hardware test.)"""
    if be.name != "hip":
        pytest.skip("the two-launch permutation exists in the HIP build only")
    cfg = _cfg(*_small(be))
    tr = be.trainer(cfg, use_graph=False)
    tr.reset()
    tr.update()
    tr.check_status()  # a healthy update: nothing recorded
    assert int(be.host(tr.region("count"))[3]) == 0
    ws = tr.region("perm_ws")
    words = ws.view(np.int32) if be.name == "emu" else ws.view(__import__("torch").int32)
    mark = tr.E * 256  # the word behind the E x 256 bucket counters
    assert int(be.host(words[:mark + 1]).sum()) == 0  # the counters are back at zero between updates
    one = np.ones(1, np.int32)
    be.put(words[mark:mark + 1], one)
    tr.update()  # (this update's scatter finds the mark already set; its own buckets are fine)
    assert int(be.host(tr.region("count"))[3]) == 1 and int(be.host(words[mark:mark + 1])[0]) == 0
    with pytest.raises(RuntimeError, match="permutation"):
        tr.check_status()
    tr.reset()
    tr.check_status()  # a reset run starts clean
    tr.close()


def test_reference_shaped_metrics_history(be):
    """`training.keep_metrics_history` / `Trainer.keep_metrics_history()`: the six `EnvMetrics` fields of EVERY step of a rollout, [T, N] each -
    what the reference keeps as `Memory.info` (train.py:170,172) and returns as `TrainOutput.metrics` (train.py:283,287-289).  Against the
    oracle's bookkeeping (oracle/env_oracle.py:metrics_step = env.py:183-194) applied in float32 to the engine's own reward / done arrays,
    over three updates with episodes ending inside the rollouts (carry-over across the update boundary): every field of every step
    exactly; and the history's last step is what the kernel left in the per-environment metric words."""
    from oracle.env_oracle import metrics_step

    cfg = _cfg(*_small(be), "reward.height_max_z=1.0101")
    tr = be.trainer(cfg, external_random=True, use_graph=False)
    tr.reset()
    tr.keep_metrics_history()
    N, T, A, E = tr.N, tr.T, tr.A, tr.E
    rng = np.random.default_rng(3)
    m = dict(episode_returns=np.zeros(N, np.float32), episode_lengths=np.zeros(N, np.int32), returned_episode_returns=np.zeros(N, np.float32),
             returned_episode_lengths=np.zeros(N, np.int32), timestep=np.zeros(N, np.int32), returned_episode=np.zeros(N, bool))
    total_done = 0
    for u in range(3):
        be.put(tr.region("noise", (T, N, A)), (3.0 * rng.standard_normal((T, N, A))).astype(np.float32))
        be.put(tr.region("perm", (E, N * T)), np.stack([rng.permutation(N * T) for _ in range(E)]).astype(np.int32))
        tr.update()
        h = tr.metrics_history()
        rew, done = be.host(tr.region("reward", (T, N))).copy(), be.host(tr.region("done", (T, N))).astype(bool)
        total_done += int(done.sum())
        for t in range(T):
            m = metrics_step(m, rew[t], done[t], np.float32)
            for k in m:
                np.testing.assert_array_equal(np.asarray(getattr(h, k)[t]), np.asarray(m[k]).astype(getattr(h, k).dtype), err_msg=f"update {u} step {t} {k}")
        assert h.timestep.shape == (T, N) and (h.timestep[T - 1] == (u + 1) * T).all()
        for k in ("episode_returns", "episode_lengths", "returned_episode_returns", "returned_episode_lengths", "timestep", "returned_episode"):
            np.testing.assert_array_equal(np.asarray(getattr(h, k)[T - 1]), be.host(tr.region(k)).astype(getattr(h, k).dtype), err_msg=k)
    assert total_done >= 2, "the test must see episodes end"
    tr.close()


def _random_bf16_config(seed, emu):
    """Random engine configurations a bf16 network can have: two hidden layers, a width the fused kernels cover."""
    rng = np.random.default_rng(4500 + seed)
    T = int(rng.integers(1, 5 if emu else 12))
    N = int(rng.integers(6, 16 if emu else 700))
    M = int(rng.choice([m for m in range(1, 17) if (T * N) % m == 0]))
    over = [f"training.num_envs={N}", f"training.num_steps={T}", f"rl.num_env_steps={T}", f"training.num_minibatches={M}", f"training.update_epochs={int(rng.integers(1, 4))}",
            f"model.hidden_size={int(rng.choice([32, 64] if emu else [32, 64, 96, 128, 160, 192, 224, 256]))}", "training.total_timesteps=100000000"]
    if rng.random() < 0.3:
        over.append("environment.model=synth_stompy_full")
    if rng.random() < 0.25:
        over.append("environment.include_c_vals=false")
    return over


@pytest.mark.parametrize("case", ["configs3"] + [f"fuzz{k}" for k in range(int(os.environ.get("MPPO_FUZZ_ENGINES", "2")))])
def test_update_bf16_mlp_tracks_the_bf16_oracle(be, case):
    """BASELINE configs[3] through the whole engine: training.mlp_dtype = "bf16" (bf16-in / f32-accumulate MFMA in the MLP
    products; GAE, loss, clip and Adam in f32).  The PPO half of one update against the oracle run with the same operand
    rounding (hp["mlp_bf16"]), on the engine's own trajectory.  Also on random configurations (MPPO_FUZZ_ENGINES widens)."""
    cfg = _cfg(*(_small(be) if case == "configs3" else _random_bf16_config(int(case[4:]), be.name == "emu")), "training.mlp_dtype=bf16")
    tr = be.trainer(cfg, external_random=True, use_graph=False)
    tr.reset()
    N, T, A, H, O, E, M = tr.N, tr.T, tr.A, tr.H, tr.O, tr.E, tr.M
    rng = np.random.default_rng(0)
    hp = dict(default_hp(cfg), mlp_bf16=True)
    p = tr.params_flat().astype(np.float64)
    opt = po.OptState(np.zeros_like(p), np.zeros_like(p), 0)
    noise = rng.standard_normal((T, N, A)).astype(np.float32)
    perms = np.stack([rng.permutation(N * T) for _ in range(E)]).astype(np.int32)
    be.put(tr.region("noise", (T, N, A)), noise)
    be.put(tr.region("perm", (E, N * T)), perms)
    tr.rollout()
    tr._sync()
    tj = {k: be.host(v).copy() for k, v in tr.traj().items()}
    obs = tj["obs"][:, :, :O].astype(np.float64)
    named = po.flat_to_named(p, O, A, H)
    for t in range(T):
        mean, ls, val = po.actor_critic_forward(named, obs[t], bf16=True)
        mean_x, _, val_x = po.actor_critic_forward(named, obs[t])
        act = po.mvn_sample(mean, ls, noise[t].astype(np.float64))
        np.testing.assert_allclose(tj["action"][t], act, atol=2e-3 * np.abs(mean_x).max() + 1e-5)
        np.testing.assert_allclose(tj["value"][t], val, atol=2e-3 * np.abs(val_x).max() + 1e-5)
    adv, tgt = po.calculate_gae(tj["done"].astype(bool), tj["value"].astype(np.float64), tj["reward"].astype(np.float64), tj["last_val"].astype(np.float64),
                                cfg.rl.gamma, cfg.rl.gae_lambda)
    np.testing.assert_allclose(tj["adv"], adv, rtol=1e-4, atol=1e-4)  # GAE stays f32 on f32 values
    tr.learn()
    traj = dict(obs=obs[:T], action=tj["action"].astype(np.float64), value=tj["value"].astype(np.float64), log_prob=tj["log_prob"].astype(np.float64))
    p_new, opt, losses = po.update_epochs_on_batch(p, opt, traj, adv, tgt, perms, O=O, A=A, H=H, num_minibatches=M, hp=hp)
    p_exact, _, losses_exact = po.update_epochs_on_batch(p, po.OptState(np.zeros_like(p), np.zeros_like(p), 0), traj, adv, tgt, perms, O=O, A=A, H=H,
                                                         num_minibatches=M, hp=default_hp(cfg))
    step = np.abs(p_new - p).max()
    got = tr.params_flat()
    err_b, err_x = np.abs(got - p_new).max(), np.abs(got - p_exact).max()
    # against the same-rounding oracle: a few percent of the update's step (rounding-boundary flips are amplified by Adam's
    # normalisation where gradients are tiny); the exact oracle is visibly further away, i.e. the rounding is really applied
    assert err_b < 0.15 * step, (err_b, step)
    np.testing.assert_allclose(tr.losses()[..., :2], losses[..., :2], rtol=5e-3, atol=5e-4)
    np.testing.assert_allclose(tr.losses()[..., :2], losses_exact[..., :2], rtol=2e-2, atol=2e-3)  # SURVEY 8c bound
    assert np.abs(p_new - p_exact).max() > 0 and np.isfinite(got).all()
    tr.close()


def test_internal_rng_update_is_deterministic_and_learns_something(be):
    cfg = _cfg(*_small(be))
    outs = []
    for rep in range(2):
        tr = be.trainer(cfg, use_graph=False)
        tr.reset()
        p0 = tr.params_flat()
        for _ in range(2):
            tr.update()
        outs.append(tr.params_flat())
        perm = be.host(tr.region("perm", (tr.E, tr.T * tr.N)))
        assert all((np.sort(perm[e]) == np.arange(tr.T * tr.N)).all() for e in range(tr.E))
        noise = be.host(tr.region("noise"))
        assert abs(noise.mean()) < 0.2 and 0.8 < noise.std() < 1.2
        if rep == 0:
            # the E epoch permutations of an update come out of ONE sort of (epoch, key) pairs; each must equal the single-permutation
            # entry point on the same stream (stream id "PERM" << 24 + epoch; these are the second update's: update counter 1 -> not
            # reproducible through mppo_permutation, whose counter word is 0, so check a FRESH trainer's first update)
            tr2 = be.trainer(cfg, use_graph=False)
            tr2.reset(); tr2.update()
            got = be.host(tr2.region("perm", (tr2.E, tr2.T * tr2.N)))
            B = tr2.T * tr2.N
            wsb = be.lib.permutation_ws_bytes(B)
            pws, one = be.zeros((wsb // 4 + 1,)), be.zeros((B,), np.int32)
            for e in range(tr2.E):
                be.lib.permutation(tr2.seed, (0x5045524D << 24) + e, B, be.ptr(one), be.ptr(pws), wsb, be.stream)
                np.testing.assert_array_equal(got[e], be.host(one), err_msg=f"epoch {e}")
            tr2.close()
        lo = tr.losses()
        assert np.isfinite(lo).all() and np.allclose(lo[..., 3], 0.5 * tr.A * (1 + np.log(2 * np.pi)), atol=0.5)
        tr.close()
    np.testing.assert_array_equal(outs[0], outs[1])  # same seed -> bitwise the same parameters
    assert np.abs(outs[0] - p0).max() > 1e-5


def test_engine_errors(be):
    with pytest.raises(ValueError, match="batch_size"):  # the reference's ValueError (train.py:253-255)
        be.trainer(_cfg("training.num_envs=6", "training.num_minibatches=4", "training.num_steps=1", "rl.num_env_steps=1"))
    with pytest.raises(ValueError, match="num_env_steps"):
        be.trainer(_cfg("rl.num_env_steps=1000"))  # the README's own example would break the reference too (quirk C-1)
    tr = be.trainer(_cfg(*_small(be)))
    with pytest.raises(nat.NativeError, match="reset"):
        tr.update()
    with pytest.raises(nat.NativeError, match="no region"):
        tr.region("no_such_region")
    with pytest.raises(ValueError, match="parameters"):
        tr.set_params_flat(np.zeros(3, np.float32))
    tr.close()
    with pytest.raises(ValueError, match="num_layers"):
        be.trainer(_cfg(*_small(be), "model.num_layers=5"))
    with pytest.raises(nat.NativeError, match="bf16"):  # bf16 products live in the fused kernels: two hidden layers
        be.trainer(_cfg(*_small(be), "model.num_layers=3", "training.mlp_dtype=bf16"))


def test_engine_with_a_robots_matrices_in_global_memory(be, monkeypatch):
    """Round 6: the export-style biped (33 dofs, 19 contact slots) keeps its contact Jacobian and M in per-environment records in global
    memory (k_physics.hip).  An engine takes those records from its arena (region `env_scratch`: nothing is allocated
    inside the hipGraph capture); whole updates equal those of an engine forced to keep everything in LDS (MPPO_ENV_SPILL=0: no such
    region) bit for bit."""
    from pathlib import Path

    robot = str(Path(__file__).parent / "golden" / "export_biped" / "robot.xml")
    over = (["training.num_envs=8", "training.num_steps=3", "rl.num_env_steps=3", "training.num_minibatches=2", "training.update_epochs=1", "model.hidden_size=32"]
            if be.name == "emu" else ["training.num_envs=200", "training.num_minibatches=4", "training.update_epochs=1"])
    cfg = _cfg(*over, "training.total_timesteps=100000000", f"environment.model={robot}")
    res = []
    for spill in (None, "0"):
        if spill is None:
            monkeypatch.delenv("MPPO_ENV_SPILL", raising=False)
        else:
            monkeypatch.setenv("MPPO_ENV_SPILL", spill)
        tr = be.trainer(cfg) if be.name == "emu" else be.trainer(cfg, use_graph=True)
        if spill is None:
            assert tr.region("env_scratch").numel() > 0 if be.xp == "torch" else tr.region("env_scratch").size > 0
        else:
            with pytest.raises(nat.NativeError):
                tr.region("env_scratch")
        tr.reset()
        for _ in range(2):
            tr.update()
        res.append((tr.params_flat(), be.host(tr.region("reward")).copy(), be.host(tr.region("state")).copy()))
        tr.close()
    monkeypatch.delenv("MPPO_ENV_SPILL", raising=False)
    for a_, b_ in zip(res[0], res[1]):
        np.testing.assert_array_equal(a_, b_)


@pytest.mark.gpu
def test_hipgraph_replay_equals_eager_launches():
    """The captured update replays the same launch sequence: parameters are bitwise equal to the eager run."""
    from backends import get_backend

    be = get_backend("hip")
    cfg = _cfg("training.num_envs=512", "training.num_minibatches=8", "training.update_epochs=2", "training.total_timesteps=100000000")
    res = []
    for graph in (False, True):
        tr = be.trainer(cfg, use_graph=graph)
        tr.reset()
        for _ in range(3):
            tr.update()
        res.append((tr.params_flat(), be.host(tr.region("count")).copy(), be.host(tr.region("reward")).copy()))
        tr.close()
    np.testing.assert_array_equal(res[0][1], res[1][1])
    np.testing.assert_array_equal(res[0][2], res[1][2])
    np.testing.assert_array_equal(res[0][0], res[1][0])


@pytest.mark.gpu
@pytest.mark.parametrize("robot,num_envs,dtype", [("synth_stompy_pro", 4096, "f32"), ("synth_stompy_full", 8192, "f32"), ("synth_stompy_pro", 4096, "bf16")])
def test_full_size_properties_on_gpu(robot, num_envs, dtype):
    """BASELINE configs[1] (stompy_pro stand-in, 4096 envs), configs[4] (20-actuator stand-in, 8192 envs) and configs[3] (bf16-in /
    f32-accumulate MLP products, 4096 envs) at full size, through the engine and its hipGraph: size-independent properties of whole updates."""
    from backends import get_backend

    be = get_backend("hip")
    cfg = _cfg(f"training.num_envs={num_envs}", f"environment.model={robot}", f"training.mlp_dtype={dtype}")
    tr = be.trainer(cfg, use_graph=True)
    tr.reset()
    for _ in range(2):
        tr.update()
    tr._sync()
    T, N, E, M = tr.T, tr.N, tr.E, tr.M
    tj = {k: be.host(v) for k, v in tr.traj().items()}
    assert all(np.isfinite(tj[k]).all() for k in ("obs", "action", "value", "reward", "log_prob", "adv", "target"))
    np.testing.assert_allclose(tj["target"], tj["adv"] + tj["value"], rtol=1e-6, atol=1e-6)  # targets = adv + value (train.py:205)
    perm = be.host(tr.region("perm", (E, T * N)))
    for e in range(E):
        assert (np.sort(perm[e]) == np.arange(T * N)).all()  # every sample used exactly once per epoch
    stats = be.host(tr.region("adv_stats", (E * M, 2)))
    mb = T * N // M
    adv = tj["adv"].reshape(-1)
    for k in (0, E * M - 1):
        g = adv[perm.reshape(-1)[k * mb:(k + 1) * mb]].astype(np.float64)
        np.testing.assert_allclose(stats[k], [g.mean(), 1 / (g.std() + 1e-8)], rtol=1e-5)
    assert be.host(tr.region("count"))[0] == 2 * E * M
    ts = be.host(tr.region("timestep"))
    assert (ts == 2 * T).all()  # every env stepped exactly T times per update
    lo = tr.losses()
    assert np.isfinite(lo).all()
    np.testing.assert_allclose(lo[..., 0], lo[..., 2] + cfg.rl.vf_coef * lo[..., 1] - cfg.rl.ent_coef * lo[..., 3], rtol=1e-5, atol=1e-5)
    # the statistics reduction equals the same sums over the rollout's own history; GAE recomputed from the arrays (train.py:185-205)
    st = tr.rollout_stats()
    np.testing.assert_allclose(st["mean_reward"], tj["reward"].astype(np.float64).mean(), rtol=1e-6)
    assert st["episodes"] == int(tj["done"].astype(bool).sum())
    adv, tgt = po.calculate_gae(tj["done"].astype(bool), tj["value"].astype(np.float64), tj["reward"].astype(np.float64), tj["last_val"].astype(np.float64),
                                cfg.rl.gamma, cfg.rl.gae_lambda)
    np.testing.assert_allclose(tj["adv"], adv, rtol=1e-4, atol=1e-4)
    tr.close()


@pytest.mark.gpu
def test_graph_with_collectives_equals_eager_with_collectives(monkeypatch):
    """The RCCL calls captured INSIDE the hipGraph (what multi-rank runs replay) against eager launches with the same
    single-rank communicator (MPPO_FORCE_COMM=1): the same kernels, the same all-reduces, the same order - bit-equal
    parameters after three updates; and the graph really was built (the engine reports it)."""
    from backends import get_backend

    be = get_backend("hip")
    cfg = _cfg("training.num_envs=256", "training.num_minibatches=8", "training.update_epochs=2", "training.total_timesteps=100000000")
    monkeypatch.setenv("MPPO_FORCE_COMM", "1")
    monkeypatch.setenv("MPPO_GRAPH_COMM", "1")  # (RCCL calls inside the graph are opt-in: never verified on more than one GPU)
    res = []
    for use_graph in (False, True):
        tr = be.trainer(cfg, use_graph=use_graph)
        tr.init_comm()
        tr.reset()
        for _ in range(3):
            tr.update()
        res.append(tr.params_flat())
        assert tr.graph_active() == use_graph
        tr.close()
    assert np.isfinite(res[0]).all()
    np.testing.assert_array_equal(res[0], res[1])


@pytest.mark.gpu
def test_rccl_path_at_world_size_one_is_the_identity(monkeypatch):
    """Hardware check of the collective path: with MPPO_FORCE_COMM=1 a single-rank RCCL communicator is created and the
    engine issues its all-reduces (f64 advantage sums per update, f32 gradient per optimizer step) on the compute stream;
    at world size 1 they are identities.  The only arithmetic difference is where the gradient's sum of squares is
    taken (inside the reduce kernel without a communicator, in its own pass after the all-reduce with one), i.e. a
    different f32 summation order of the global norm: parameters agree to rounding, not bitwise."""
    from backends import get_backend

    be = get_backend("hip")
    cfg = _cfg("training.num_envs=256", "training.num_minibatches=8", "training.update_epochs=2", "training.total_timesteps=100000000")
    res = []
    for force in ("0", "1"):
        monkeypatch.setenv("MPPO_FORCE_COMM", force)
        tr = be.trainer(cfg, use_graph=False)
        tr.init_comm()
        tr.reset()
        for _ in range(2):
            tr.update()
        res.append(tr.params_flat())
        tr.close()
    assert np.isfinite(res[1]).all()
    # measured on MI355X: 3 % of the elements differ, by at most 6e-5 (one Adam step is 3e-4): rounding-level drift
    np.testing.assert_allclose(res[0], res[1], rtol=1e-3, atol=3e-4)



@pytest.mark.gpu
@pytest.mark.parametrize("envs,steps,minibatches,epochs", [(6, 3, 2, 2), (64, 10, 32, 4), (1000, 7, 4, 3), (13107, 10, 2, 1)])
def test_two_launch_permutations_equal_the_sort_at_odd_sizes(envs, steps, minibatches, epochs):
    """The same at batch sizes that fill the 256 buckets unevenly or hardly at all (B = 18: most buckets empty; B = 640: BASELINE configs[0];
    B = 7 000: not a multiple of the scatter launch's 4 096 values per workgroup; B = 131 070: the largest batches the two launches take,
    512 values per bucket on average = the LDS bitonic sort instead of the rank sort)."""
    from backends import get_backend

    be = get_backend("hip")
    cfg = _cfg(f"training.num_envs={envs}", f"training.num_steps={steps}", f"rl.num_env_steps={steps}", f"training.num_minibatches={minibatches}",
               f"training.update_epochs={epochs}", "training.total_timesteps=1000000000")
    tr = be.trainer(cfg, use_graph=False)
    tr.reset()
    E, B = tr.E, tr.T * tr.N
    assert (E, B) == (epochs, envs * steps)
    prev = None
    for u in range(3):
        tr.update()
        tr._sync()
        perm = be.host(tr.region("perm", (E, B))).copy()
        for e in range(E):
            assert (np.sort(perm[e]) == np.arange(B)).all(), (u, e)
        if u == 0:
            wsb = be.lib.permutation_ws_bytes(B)
            pws, one = be.zeros((wsb // 4 + 1,)), be.zeros((B,), np.int32)
            for e in range(E):
                be.lib.permutation(tr.seed, (0x5045524D << 24) + e, B, be.ptr(one), be.ptr(pws), wsb, be.stream)
                np.testing.assert_array_equal(perm[e], be.host(one), err_msg=f"epoch {e}")
                np.testing.assert_array_equal(perm[e], _host_permutation(tr.seed, (0x5045524D << 24) + e, B), err_msg=f"epoch {e} against the host's stable argsort of the Philox keys")
        else:
            assert (perm != prev).mean() > 0.8
        prev = perm
    tr.close()


@pytest.mark.gpu
def test_two_launch_permutations_survive_a_rewound_update_index():
    """The bucket counters of the two-launch permutation are taken back to zero by the launch that read them, so they do not depend on the
    history of the update index: an update repeated with the index written back through the arena (what a C-ABI caller restoring its own
    snapshot of the `count` region does, without mppo_engine_reset) draws the very same permutations again."""
    from backends import get_backend

    be = get_backend("hip")
    cfg = _cfg("training.num_envs=1024", "training.total_timesteps=1000000000")
    tr = be.trainer(cfg, use_graph=True)
    tr.reset()
    E, B = tr.E, tr.T * tr.N
    tr.update(); tr.update()
    tr._sync()
    count = be.host(tr.region("count")).copy()
    tr.update()
    tr._sync()
    first = be.host(tr.region("perm", (E, B))).copy()
    be.put(tr.region("count"), count)  # same update index (and Adam step) again: same parity as the launch before
    tr.update()
    tr._sync()
    again = be.host(tr.region("perm", (E, B))).copy()
    np.testing.assert_array_equal(first, again)
    for e in range(E):
        assert (np.sort(again[e]) == np.arange(B)).all()
    tr.update()
    tr._sync()
    nxt = be.host(tr.region("perm", (E, B)))
    for e in range(E):
        assert (np.sort(nxt[e]) == np.arange(B)).all()
    tr.close()


@pytest.mark.gpu
@pytest.mark.parametrize("config,envs", [("stompy_pro", 4096), ("stompy_full", 8192), ("stompy_pro", 32768)])
def test_two_launch_permutations_equal_the_sort_at_full_size(config, envs):
    """The engine's permutations at BASELINE sizes (configs[1]: B = 40 960 samples, configs[4]: B = 81 920 - more than 16 index bits; E = 4 epochs): two launches - scatter of (key, index) values into 256
    buckets, one LDS sort per bucket (csrc/k_perm.hip).  Round 6: also 32 768 environments on ONE rank (B = 327 680: 1 024 buckets - configs[2]'s
    global batch without the sharding).  The first update's permutations equal `mppo_permutation` (the same two launches for one stream) and the
    host's stable argsort of the same Philox keys (tests/test_kernels_ppo.py _host_permutation) epoch by epoch, bit for bit; the following
    updates (the double-buffered bucket counters of both parities, replayed from the hipGraph) stay bijective and differ from update to update."""
    from backends import get_backend

    be = get_backend("hip")
    from minppo_amd.config import load_config_from_cli

    cfg = load_config_from_cli([config, f"training.num_envs={envs}"])
    tr = be.trainer(cfg, use_graph=True)
    tr.reset()
    E, B = tr.E, tr.T * tr.N
    assert (E, B) == (4, 10 * envs)
    seen = []
    for u in range(4):
        tr.update()
        tr._sync()
        perm = be.host(tr.region("perm", (E, B))).copy()
        for e in range(E):
            assert (np.sort(perm[e]) == np.arange(B)).all(), (u, e)
        seen.append(perm)
        if u == 0:
            wsb = be.lib.permutation_ws_bytes(B)
            pws, one = be.zeros((wsb // 4 + 1,)), be.zeros((B,), np.int32)
            for e in range(E):
                be.lib.permutation(tr.seed, (0x5045524D << 24) + e, B, be.ptr(one), be.ptr(pws), wsb, be.stream)
                np.testing.assert_array_equal(perm[e], be.host(one), err_msg=f"epoch {e}")
                np.testing.assert_array_equal(perm[e], _host_permutation(tr.seed, (0x5045524D << 24) + e, B), err_msg=f"epoch {e} against the host's stable argsort of the Philox keys")
    assert tr.graph_active()
    for u in range(1, 4):
        assert (seen[u] != seen[u - 1]).mean() > 0.99
    tr.close()
