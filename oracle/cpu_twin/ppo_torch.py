"""The PPO half of the CPU baseline on torch-CPU float32 (BLAS threads = host cores).          TEST / BENCH INFRASTRUCTURE.

PARITY UNPINNED (oracle/ppo_oracle.py).  The same update the reference runs under `jax.jit` on a CPU (minppo/train.py:146-283):
policy forward + sample (`:157-160`), bootstrap value (`:182`), GAE (`:185-205`), E epochs of M shuffled minibatches of
`value_and_grad(_loss_fn)` (`:218-247`) and `clip_by_global_norm + adam` with the linear schedule (`:98-124,248`), written the way one
writes it for a CPU: dense float32 matmuls on all cores, autograd for the backward pass.  Checked against oracle/ppo_oracle.py in
tests/test_cpu_twin.py.  Only bench.py's `cpu_baseline` leg and that test import it."""

from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

LOG_2PI = math.log(2.0 * math.pi)


def forward(p: Dict[str, torch.Tensor], x: torch.Tensor, use_tanh: bool = True):
    """ActorCritic.__call__ (train.py:71-83): actor MLP (tanh if model.use_tanh), critic MLP (always ReLU), two hidden layers each."""
    act = torch.tanh if use_tanh else torch.relu
    h = act(x @ p["a_w1"] + p["a_b1"])
    h = act(h @ p["a_w2"] + p["a_b2"])
    mean = h @ p["a_w3"] + p["a_b3"]
    c = torch.relu(x @ p["c_w1"] + p["c_b1"])
    c = torch.relu(c @ p["c_w2"] + p["c_b2"])
    value = (c @ p["c_w3"] + p["c_b3"]).squeeze(-1)
    return mean, value


def log_prob(a, mean, log_std):
    z = (a - mean) * torch.exp(-log_std)
    return -0.5 * (z * z).sum(-1) - log_std.sum() - 0.5 * a.shape[-1] * LOG_2PI


def loss_fn(p, obs, action, old_value, old_logp, gae, targets, clip_eps=0.2, vf_coef=0.5, ent_coef=0.0, use_tanh=True):
    """_loss_fn (train.py:218-243)."""
    mean, value = forward(p, obs, use_tanh)
    lp = log_prob(action, mean, p["log_std"])
    v_clip = old_value + (value - old_value).clamp(-clip_eps, clip_eps)
    value_loss = 0.5 * torch.maximum((value - targets) ** 2, (v_clip - targets) ** 2).mean()
    ratio = torch.exp(lp - old_logp)
    g = (gae - gae.mean()) / (gae.std(unbiased=False) + 1e-8)
    actor_loss = -torch.minimum(ratio * g, ratio.clamp(1.0 - clip_eps, 1.0 + clip_eps) * g).mean()
    entropy = 0.5 * action.shape[-1] * (1.0 + LOG_2PI) + p["log_std"].sum()
    total = actor_loss + vf_coef * value_loss - ent_coef * entropy
    return total, (value_loss, actor_loss, entropy)


def gae(done, value, reward, last_val, gamma: float, lam: float):
    """_calculate_gae (train.py:185-205), reverse over t."""
    T = reward.shape[0]
    adv = torch.zeros_like(reward)
    g = torch.zeros_like(last_val)
    nv = last_val
    for t in range(T - 1, -1, -1):
        nd = 1.0 - done[t].to(reward.dtype)
        delta = reward[t] + gamma * nv * nd - value[t]
        g = delta + gamma * lam * nd * g
        adv[t] = g
        nv = value[t]
    return adv, adv + value


class Adam:
    """optax.chain(clip_by_global_norm(c), adam(lr(count), eps=1e-5)) (train.py:115-124) with linear_schedule (train.py:98-101)."""

    def __init__(self, params, lr, max_grad_norm, anneal, minibatch_size, update_epochs, num_updates):
        self.p = params
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.count = 0
        self.lr, self.c, self.anneal, self.div, self.nu = lr, max_grad_norm, anneal, minibatch_size * update_epochs, num_updates

    @torch.no_grad()
    def step(self, grads):
        norm = torch.sqrt(sum((g * g).sum() for g in grads.values()))
        scale = 1.0 if float(norm) < self.c else self.c / float(norm)
        lr = self.lr * (1.0 - (self.count // self.div) / self.nu) if self.anneal else self.lr
        t = self.count + 1
        bc1, bc2 = 1.0 - 0.9 ** t, 1.0 - 0.999 ** t
        for k, g in grads.items():
            g = g * scale
            self.m[k].mul_(0.9).add_(g, alpha=0.1)
            self.v[k].mul_(0.999).addcmul_(g, g, value=0.001)
            self.p[k].sub_(lr * (self.m[k] / bc1) / (torch.sqrt(self.v[k] / bc2) + 1e-5))
        self.count += 1


def update_epochs(p, opt: Adam, traj, adv, tgt, perms, num_minibatches: int, hp: dict, use_tanh: bool = True):
    """_update_epoch x E (train.py:209-270): flatten [T, N, ...] -> [B, ...], take(perm), M minibatches of value_and_grad + apply_gradients."""
    B = adv.numel()
    flat = {k: traj[k].reshape((B,) + tuple(traj[k].shape[2:])) for k in ("obs", "action", "value", "log_prob")}
    a, tg = adv.reshape(B), tgt.reshape(B)
    mb = B // num_minibatches
    losses = []
    for e in range(perms.shape[0]):
        perm = perms[e]
        for k in range(num_minibatches):
            idx = perm[k * mb:(k + 1) * mb]
            for v in p.values():
                v.requires_grad_(True)
                v.grad = None
            total, aux = loss_fn(p, flat["obs"][idx], flat["action"][idx], flat["value"][idx], flat["log_prob"][idx], a[idx], tg[idx], hp["clip_eps"], hp["vf_coef"],
                                 hp["ent_coef"], use_tanh)
            total.backward()
            grads = {k2: v.grad for k2, v in p.items()}
            for v in p.values():
                v.requires_grad_(False)
            opt.step(grads)
            losses.append((float(total.detach()), float(aux[0].detach()), float(aux[1].detach()), float(aux[2].detach())))
    return np.asarray(losses)


def one_update(twin, p: Dict[str, torch.Tensor], opt: Adam, last_obs: torch.Tensor, noise: torch.Tensor, perms: torch.Tensor, num_minibatches: int, hp: dict,
               use_tanh: bool = True):
    """One `_update_step` (train.py:146-283) on the CPU: rollout through the C++ / OpenMP environment twin, then the update above.
    Returns (last_obs, mean reward, losses [E * M, 4] = total / value / actor / entropy of every minibatch step)."""
    T, N = noise.shape[0], noise.shape[1]
    O = twin.obs_dim
    obs_l, act_l, val_l, rew_l, lp_l, done_l = [], [], [], [], [], []
    with torch.no_grad():
        for t in range(T):
            mean, value = forward(p, last_obs, use_tanh)
            action = mean + torch.exp(p["log_std"]) * noise[t]
            lp = log_prob(action, mean, p["log_std"])
            o, r, d = twin.step(action.numpy())
            obs_l.append(last_obs); act_l.append(action); val_l.append(value); lp_l.append(lp)
            rew_l.append(torch.from_numpy(r.copy())); done_l.append(torch.from_numpy(d.astype(np.bool_)))
            last_obs = torch.from_numpy(o[:, :O].copy())
        _, last_val = forward(p, last_obs, use_tanh)
        traj = dict(obs=torch.stack(obs_l), action=torch.stack(act_l), value=torch.stack(val_l), log_prob=torch.stack(lp_l))
        reward, done = torch.stack(rew_l), torch.stack(done_l)
        adv, tgt = gae(done, traj["value"], reward, last_val, hp["gamma"], hp["gae_lambda"])
    losses = update_epochs(p, opt, traj, adv, tgt, perms, num_minibatches, hp, use_tanh)
    return last_obs, float(reward.mean()), losses
