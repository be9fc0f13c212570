"""Long-horizon equivalence of the engine and the CPU twin (oracle/cpu_twin: C++ / OpenMP environment step + torch-CPU PPO update).

Two-update parity tests cannot see a slow bias - a wrong `1 - beta2^t` at large t, a schedule that drifts, episode metrics that roll over
wrongly, a statistic normalised by the wrong count - because after two updates such errors are below every tolerance.  Here both sides
train for 300 updates (2 400 optimizer steps) from the same initial parameters on the same Philox-independent random inputs (the action
noise and the minibatch permutations are drawn on the host and written into both: `external_random`).  Trajectories of a rigid-body
simulation under a stochastic policy are chaotic, so the two float32 implementations part ways after a few updates and cannot be compared
point by point; what must agree is everything a slow bias would move:

  * the first updates, before the divergence has grown: rewards and losses per update, tightly;
  * the learning curves (mean reward, value loss, entropy via log_std) smoothed over windows of 20 updates, within the spread that a
    SECOND twin run shows against the first when only its rounding differs (parameters perturbed by one part in 1e7) - the envelope is
    measured, not assumed - times a safety factor;
  * the parameters: per tensor the two runs' norms within a few per cent, and the distance between the runs small against the distance
    both have travelled from the common start.
"""
import numpy as np
import pytest

from minppo_amd.config import make_config

BASE = {"kscale_id": "5eb3cb7f23232298", "visualization": {"camera_name": "track"}}
N, T, M, E, H_, UPDATES = 256, 10, 4, 2, 64, 300  # 2 400 optimizer steps; a 64-wide network keeps the two CPU runs inside the test's minute


def _cfg():
    return make_config(BASE, [f"training.num_envs={N}", f"training.num_minibatches={M}", f"training.update_epochs={E}", f"model.hidden_size={H_}",
                              f"training.total_timesteps={UPDATES * T * N}"])


def _random_inputs(A):
    rng = np.random.default_rng(2025)
    for _ in range(UPDATES):
        yield rng.standard_normal((T, N, A)).astype(np.float32), np.stack([rng.permutation(N * T) for _ in range(E)]).astype(np.int32)


def _twin_run(cfg, flat0, O, A, H, scale=1.0):
    import torch

    from minppo_amd.model import load_model
    from oracle import ppo_oracle as po
    from oracle.cpu_twin import Twin, ppo_torch as pt
    from oracle.env_oracle import default_hp

    import bench  # (the GPU box reports all 256 host cores but grants this container 16: bench._usable_cores reads the cgroup quota)

    cores = max(1, min(8, bench._usable_cores()))
    torch.set_num_threads(cores)  # 320-row minibatches of a 64-wide network: more threads only add synchronisation
    cm = load_model(cfg.environment.model or cfg.kscale_id)
    tw = Twin(cm, include_c_vals=bool(cfg.environment.include_c_vals), threads=cores)
    hp = default_hp(cfg)
    named = po.flat_to_named(flat0.astype(np.float64) * scale, O, A, H)
    p = {k: torch.tensor(np.asarray(v, np.float32)) for k, v in named.items()}
    opt = pt.Adam(p, hp["lr_train"] if hp["anneal_lr"] else hp["lr_opt"], hp["max_grad_norm"], hp["anneal_lr"], N * T // M, E, max(hp["num_updates"], 1))
    last_obs = torch.from_numpy(tw.reset(N)[:, :O].copy())
    rew, los = [], []
    for noise, perms in _random_inputs(A):
        last_obs, r, l = pt.one_update(tw, p, opt, last_obs, torch.from_numpy(noise), torch.from_numpy(perms.astype(np.int64)), M, hp, bool(cfg.model.use_tanh))
        rew.append(r)
        los.append(l.mean(0))
    tw.close()
    flat = po.named_to_flat({k: v.numpy().astype(np.float64) for k, v in p.items()}, O, A, H)
    return np.asarray(rew), np.asarray(los), flat, opt.count


def _smooth(x, w=20):
    return np.convolve(x, np.ones(w) / w, mode="valid")


@pytest.mark.gpu
def test_a_long_training_run_tracks_the_cpu_twin():
    import torch

    from backends import get_backend
    from oracle import ppo_oracle as po

    be = get_backend("hip")
    cfg = _cfg()
    tr = be.trainer(cfg, external_random=True, use_graph=True)
    tr.reset()
    O, A, H = tr.O, tr.A, tr.H
    flat0 = tr.params_flat().copy()
    rew, los = [], []
    noise_r, perm_r = tr.region("noise", (T, N, A)), tr.region("perm", (E, N * T))
    for noise, perms in _random_inputs(A):
        noise_r.copy_(torch.from_numpy(noise))
        perm_r.copy_(torch.from_numpy(perms))
        torch.cuda.synchronize()
        tr.update()
        rew.append(tr.rollout_stats()["mean_reward"])
        los.append(tr.losses().reshape(-1, 4).mean(0))
    tr.check_status()
    rew, los, flat_g = np.asarray(rew), np.asarray(los), tr.params_flat().astype(np.float64)
    count = int(be.host(tr.region("count"))[0])
    tr.close()

    rew_a, los_a, flat_a, count_a = _twin_run(cfg, flat0, O, A, H)
    rew_b, los_b, flat_b, _ = _twin_run(cfg, flat0, O, A, H, scale=1.0 + 1e-7)  # the same run, differently rounded: the envelope
    assert count == count_a == UPDATES * E * M  # optimizer steps (the LR schedule's position, train.py:98-101)
    assert np.isfinite(rew).all() and np.isfinite(los).all() and np.isfinite(flat_g).all()

    # (1) before the divergence has grown: per update
    np.testing.assert_allclose(rew[:3], rew_a[:3], rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(los[:3, 1], los_a[:3, 1], rtol=2e-2, atol=1e-3)  # value loss
    np.testing.assert_allclose(los[:3, 3], los_a[:3, 3], rtol=1e-5, atol=1e-5)  # entropy: a function of log_std alone

    # (2) the curves, smoothed, inside the measured envelope (x 4) + a floor of 2 % of the curve's range
    for name, g, a, b in (("mean reward", rew, rew_a, rew_b), ("value loss", los[:, 1], los_a[:, 1], los_b[:, 1]), ("entropy", los[:, 3], los_a[:, 3], los_b[:, 3])):
        sg, sa, sb = _smooth(g), _smooth(a), _smooth(b)
        env = 4.0 * np.abs(sa - sb).max() + 0.02 * (np.ptp(sa) + np.abs(sa).mean() * 0.1)
        worst = np.abs(sg - sa).max()
        assert worst <= env, f"{name}: smoothed curves differ by {worst:.4g}, envelope {env:.4g} (twin vs perturbed twin: {np.abs(sa - sb).max():.4g})"
    # the entropy curve is where an optimizer bias shows first (log_std is moved by Adam alone): it must have MOVED, and the same way
    assert abs(los[-1, 3] - los[0, 3]) > 1e-3 and np.sign(los[-1, 3] - los[0, 3]) == np.sign(los_a[-1, 3] - los_a[0, 3])
    np.testing.assert_allclose(los[-1, 3] - los[0, 3], los_a[-1, 3] - los_a[0, 3], rtol=0.25)

    # (3) the parameters: per tensor, the norms of the two runs within 2 % (+ the twin pair's own spread), and the runs closer to each other than
    # to the start
    sl = po.param_slices(O, A, H)
    rows, bad = [], []
    for k, (o, shp) in sl.items():
        n = int(np.prod(shp))
        g, a, b, z = flat_g[o:o + n], flat_a[o:o + n], flat_b[o:o + n], flat0[o:o + n].astype(np.float64)
        if not np.any(a != z):
            continue
        # distances travelled from the common start (for the zero-initialised biases and log_std that IS the norm): a biased step size, a wrong
        # bias correction or schedule shows as a different path length; the runs must also be closer to each other than to the start
        tg, ta, tb = np.linalg.norm(g - z), np.linalg.norm(a - z), np.linalg.norm(b - z)
        dga, dab = np.linalg.norm(g - a), np.linalg.norm(a - b)
        rows.append(f"{k:8s} travelled: engine {tg:.4g} twin {ta:.4g} twin' {tb:.4g} | engine-twin {dga:.4g} twin-twin' {dab:.4g}")
        if abs(tg - ta) > 0.05 * ta + 4.0 * abs(ta - tb) + 1e-6 or dga > max(1.0 * 0.5 * (tg + ta), 4.0 * dab) + 1e-6:
            bad.append(k)
    print("\n".join(rows))
    assert not bad, "tensors whose path from the common start differs between the engine and the twin: " + ", ".join(bad) + "\n" + "\n".join(rows)
