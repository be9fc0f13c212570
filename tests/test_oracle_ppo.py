"""Pins the PPO oracle: analytic known answers and torch float64 autograd.

The reference ships no tests or golden vectors ("parity unpinned"), so the oracle is anchored on
closed forms and on an independent autograd implementation of the same loss (reference
`minppo/train.py:218-243`)."""

import math

import numpy as np
import pytest
import torch

from oracle import ppo_oracle as po

O, A, H = 12, 3, 16


def _batch(rng, n, dtype=np.float64):
    return dict(
        obs=rng.standard_normal((n, O)).astype(dtype),
        action=rng.standard_normal((n, A)).astype(dtype),
        value=rng.standard_normal(n).astype(dtype),
        log_prob=(-4 + 0.3 * rng.standard_normal(n)).astype(dtype),
        gae=rng.standard_normal(n).astype(dtype),
        tgt=rng.standard_normal(n).astype(dtype),
    )


def _torch_loss(named, b, clip_eps, vf_coef, ent_coef, use_tanh):
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in named.items()}
    x = torch.tensor(b["obs"])
    act = torch.tanh if use_tanh else torch.relu
    L = po.n_hidden(named)  # `MLP([hidden_size] * num_layers + [out])` (reference train.py:56-68,79,82)
    h = c = x
    for i in range(1, L + 1):
        h = act(h @ tp[f"a_w{i}"] + tp[f"a_b{i}"])
        c = torch.relu(c @ tp[f"c_w{i}"] + tp[f"c_b{i}"])
    mean = h @ tp[f"a_w{L + 1}"] + tp[f"a_b{L + 1}"]
    value = (c @ tp[f"c_w{L + 1}"] + tp[f"c_b{L + 1}"])[:, 0]
    dist = torch.distributions.Independent(torch.distributions.Normal(mean, torch.exp(tp["log_std"])), 1)
    logp = dist.log_prob(torch.tensor(b["action"]))
    ov, tg, g, olp = (torch.tensor(b[k]) for k in ("value", "tgt", "gae", "log_prob"))
    vclip = ov + (value - ov).clamp(-clip_eps, clip_eps)
    vl = 0.5 * torch.maximum((value - tg) ** 2, (vclip - tg) ** 2).mean()
    ratio = torch.exp(logp - olp)
    gn = (g - g.mean()) / (g.std(unbiased=False) + 1e-8)
    la = -torch.minimum(ratio * gn, ratio.clamp(1 - clip_eps, 1 + clip_eps) * gn).mean()
    ent = dist.entropy().mean()
    total = la + vf_coef * vl - ent_coef * ent
    total.backward()
    return (total.item(), vl.item(), la.item(), ent.item()), {k: v.grad.numpy() for k, v in tp.items()}


@pytest.mark.parametrize("use_tanh", [True, False])
@pytest.mark.parametrize("ent_coef,layers", [(0.0, 2), (0.01, 2), (0.01, 1), (0.0, 3), (0.01, 4)])
def test_loss_grad_matches_torch_autograd(use_tanh, ent_coef, layers):
    rng = np.random.default_rng(3)
    named = po.init_params(7, O, A, H, L=layers)
    assert po.n_hidden(named) == layers and po.named_to_flat(named, O, A, H).size == po.flat_size(O, A, H, layers) >= po.param_count(O, A, H, layers)
    for k in named:  # move off the init point so every path carries gradient
        named[k] = named[k] + 0.3 * rng.standard_normal(named[k].shape)
    b = _batch(rng, 200)
    # widen the ratio / value-clip spread so clipped and unclipped branches are all exercised
    mean, ls, _ = po.actor_critic_forward(named, b["obs"], use_tanh)
    b["log_prob"] = po.mvn_log_prob(b["action"], mean, ls) + 0.4 * rng.standard_normal(200)
    lo, grads = po.loss_and_grad(named, b["obs"], b["action"], b["value"], b["log_prob"], b["gae"], b["tgt"], 0.2, 0.5,
                                 ent_coef, use_tanh)
    ref, tgr = _torch_loss(named, b, 0.2, 0.5, ent_coef, use_tanh)
    np.testing.assert_allclose(lo, ref, rtol=1e-12, atol=1e-12)
    for k in grads:
        np.testing.assert_allclose(grads[k], tgr[k], rtol=1e-9, atol=1e-12, err_msg=k)
    ratio = np.exp(po.mvn_log_prob(b["action"], mean, ls) - b["log_prob"])
    assert (ratio > 1.2).any() and (ratio < 0.8).any() and ((ratio > 0.8) & (ratio < 1.2)).any()


def test_first_minibatch_known_answers():
    """before any update ratio == 1, L_pi = -mean(g_hat) ~ 0, entropy closed form."""
    rng = np.random.default_rng(0)
    A10 = 10
    named = po.init_params(1337, O, A10, H)
    obs = rng.standard_normal((64, O))
    mean, ls, value = po.actor_critic_forward(named, obs)
    action = po.mvn_sample(mean, ls, rng.standard_normal((64, A10)))
    logp = po.mvn_log_prob(action, mean, ls)
    gae, tgt = rng.standard_normal(64), rng.standard_normal(64)
    lo, _ = po.loss_and_grad(named, obs, action, value, logp, gae, tgt)
    assert abs(lo.actor_loss) < 1e-12
    assert lo.value_loss == pytest.approx(0.5 * np.mean((value - tgt) ** 2), rel=1e-12)
    assert lo.entropy == pytest.approx(14.189385332046727, abs=1e-12)
    assert po.mvn_log_prob(mean, mean, ls)[0] == pytest.approx(-0.5 * A10 * math.log(2 * math.pi))


def test_gae_known_answers():
    rng = np.random.default_rng(1)
    T, N = 5, 7
    v, r, lv = rng.standard_normal((T, N)), rng.standard_normal((T, N)), rng.standard_normal(N)
    ones, zeros = np.ones((T, N), bool), np.zeros((T, N), bool)
    adv, tgt = po.calculate_gae(ones, v, r, lv, 0.99, 0.95)
    np.testing.assert_allclose(adv, r - v, atol=1e-15)  # done everywhere
    np.testing.assert_allclose(tgt, r, atol=1e-15)
    adv, _ = po.calculate_gae(zeros, v, r, lv, 0.0, 0.95)
    np.testing.assert_allclose(adv, r - v, atol=1e-15)  # gamma = 0
    adv, _ = po.calculate_gae(zeros, v, r, lv, 0.99, 0.0)  # lambda = 0 -> delta
    vn = np.concatenate([v[1:], lv[None]])
    np.testing.assert_allclose(adv, r + 0.99 * vn - v, atol=1e-15)
    # hand-computed T=3, single env, done at t=1
    v3 = np.array([[1.0], [2.0], [3.0]]); r3 = np.array([[0.5], [1.0], [-1.0]]); d3 = np.array([[False], [True], [False]])
    adv, tgt = po.calculate_gae(d3, v3, r3, np.array([4.0]), 0.9, 0.8)
    a2 = -1.0 + 0.9 * 4.0 - 3.0
    a1 = 1.0 - 2.0
    a0 = 0.5 + 0.9 * 2.0 - 1.0 + 0.9 * 0.8 * a1
    np.testing.assert_allclose(adv[:, 0], [a0, a1, a2], atol=1e-15)
    np.testing.assert_allclose(tgt, adv + v3)


def test_clip_adam_schedule_known_answers():
    g = np.array([0.6, 0.8])  # norm 1
    c, n = po.clip_by_global_norm(g, 0.5)
    np.testing.assert_allclose(c, g / 2); assert n == pytest.approx(1.0)
    c, _ = po.clip_by_global_norm(g * 0.4, 0.5)
    np.testing.assert_allclose(c, g * 0.4)
    p, m, v = po.adam_step(np.zeros(2), np.zeros(2), np.zeros(2), g, 0, 1e-3)
    np.testing.assert_allclose(p, -1e-3 * g / (np.abs(g) + 1e-5), rtol=1e-12)
    # schedule (quirk C-2): divisor is minibatch_size*update_epochs
    mb, E, nu, lr = 1280, 4, 24414, 3e-4
    assert po.linear_schedule(0, lr, mb, E, nu) == lr
    assert po.linear_schedule(mb * E - 1, lr, mb, E, nu) == lr
    assert po.linear_schedule(mb * E, lr, mb, E, nu) == pytest.approx(lr * (1 - 1 / nu))


def test_param_packing_roundtrip_and_tree():
    named = po.init_params(5, O, A, H)
    flat = po.named_to_flat(named, O, A, H)
    assert po.param_count(O, A, H) == 2 * (O * H + H + H * H + H) + H * A + 2 * A + H + 1
    assert flat.size == po.flat_size(O, A, H) >= po.param_count(O, A, H) and flat.size % 4 == 0
    assert all(o % 4 == 0 for o, _ in po.param_slices(O, A, H).values())  # every tensor starts on a 16-byte boundary
    assert np.count_nonzero(flat) <= po.param_count(O, A, H)  # the alignment words are zero
    back = po.flat_to_named(flat, O, A, H)
    for k in named:
        np.testing.assert_array_equal(named[k], back[k])
    tree = po.named_to_tree(named)
    assert set(tree["params"]) == {"MLP_0", "MLP_1", "log_std"}
    assert tree["params"]["MLP_0"]["Dense_2"]["kernel"].shape == (H, A)
    assert tree["params"]["MLP_1"]["Dense_2"]["kernel"].shape == (H, 1)
    named2 = po.tree_to_named(tree)
    np.testing.assert_array_equal(po.named_to_flat(named2, O, A, H), flat)
    # orthogonal init: columns orthonormal up to gain
    w = named["a_w2"]
    np.testing.assert_allclose(w.T @ w, 2.0 * np.eye(H), atol=1e-12)
    # SURVEY: P = 512*O + 258*A + 132353 at H=256
    assert po.param_count(225, 10, 256) == 250133
    assert po.param_count(415, 20, 256) == 349993
    assert po.flat_size(225, 10, 256) == 250140 and po.flat_size(415, 20, 256) == 349996  # + alignment words


def test_epoch_driver_matches_manual_steps():
    rng = np.random.default_rng(2)
    T, N, M = 4, 8, 4
    traj = dict(obs=rng.standard_normal((T, N, O)), action=rng.standard_normal((T, N, A)),
                value=rng.standard_normal((T, N)), log_prob=-4 + rng.standard_normal((T, N)) * 0.1)
    adv, tgt = rng.standard_normal((T, N)), rng.standard_normal((T, N))
    p0 = po.named_to_flat(po.init_params(9, O, A, H), O, A, H)
    hp = dict(clip_eps=0.2, vf_coef=0.5, ent_coef=0.0, max_grad_norm=0.5, anneal_lr=True, lr_train=3e-4, lr_opt=1e-3,
              update_epochs=2, num_updates=100)
    perms = np.stack([rng.permutation(T * N) for _ in range(2)])
    p1, opt, losses = po.update_epochs_on_batch(p0, po.OptState(np.zeros_like(p0), np.zeros_like(p0), 0), traj, adv, tgt,
                                                perms, O=O, A=A, H=H, num_minibatches=M, hp=hp)
    assert opt.count == 2 * M and losses.shape == (2, M, 4)
    # row = t*N + n flattening: first minibatch of epoch 0 by hand
    idx = perms[0][: T * N // M]
    t_i, n_i = idx // N, idx % N
    lo, _ = po.loss_and_grad(po.flat_to_named(p0, O, A, H), traj["obs"][t_i, n_i], traj["action"][t_i, n_i],
                             traj["value"][t_i, n_i], traj["log_prob"][t_i, n_i], adv[t_i, n_i], tgt[t_i, n_i])
    np.testing.assert_allclose(losses[0, 0], lo, rtol=1e-13)
    assert np.abs(p1 - p0).max() > 0
