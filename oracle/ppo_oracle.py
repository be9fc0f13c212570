"""CPU oracle for the PPO half of the minppo hot path.            TEST INFRASTRUCTURE.

PARITY UNPINNED: the reference (kscalelabs/minppo @ 2024-10-16) ships no tests
or golden vectors and its arithmetic lives in JAX / Flax / Optax / Distrax,
none of which is installable here.  This file is a NumPy
restatement written from the cited reference lines plus the published
semantics of those libraries; it is cross-checked against torch-CPU float64
autograd (tests/test_oracle_ppo.py), analytic known-answer tests, and the
committed fixtures under tests/golden/.

Only tests/, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may
import this package.  The product path (minppo_amd/) never does.

Every function is dtype-generic: pass float64 arrays for the reference
answer, float32 to mimic the reference's working precision.
All citations are relative to /root/reference/.
"""

from __future__ import annotations

import math
from typing import Dict, NamedTuple, Tuple

import numpy as np

LOG_2PI = math.log(2.0 * math.pi)


# ---------------------------------------------------------------------------
# parameters: tree <-> flat vector
# ---------------------------------------------------------------------------
# Flat layout (shared with the HIP engine, include/minppo_hip.h):
#   actor : W1[O,H] b1[H] W2[H,H] b2[H] W3[H,A] b3[A]   log_std[A]
#   critic: W1[O,H] b1[H] W2[H,H] b2[H] W3[H,1] b3[1]
# Tree layout = what the reference pickles (minppo/train.py:314; Flax naming,
# SURVEY Appendix A): {"params": {"MLP_0": {"Dense_i": {kernel,bias}}, "log_std", "MLP_1": {...}}}
# `model.num_layers` hidden layers per MLP (config.py:53, default 2): tensors a_w1 .. a_w{L+1}, the last one the output layer.


def param_count(O: int, A: int, H: int, L: int = 2) -> int:
    """Number of parameters of the model (P = 512 O + 258 A + 132 353 at H = 256, two hidden layers)."""
    return 2 * (O * H + H + (L - 1) * (H * H + H)) + H * A + A + A + H + 1


def n_hidden(p: Dict[str, np.ndarray]) -> int:
    """Hidden layers of a named parameter set (`model.num_layers`, train.py:79,82): a_w1 .. a_w{L} hidden, a_w{L+1} the output layer."""
    return sum(1 for k in p if k.startswith("a_w")) - 1


def _slices(O: int, A: int, H: int, L: int = 2):
    def mlp(pref, last):
        for i in range(L + 1):
            n_in, n_out = (O if i == 0 else H), (last if i == L else H)
            yield f"{pref}_w{i + 1}", (n_in, n_out)
            yield f"{pref}_b{i + 1}", (n_out,)
    out = {}
    off = 0
    for name, shape in (*mlp("a", A), ("log_std", (A,)), *mlp("c", 1)):
        out[name] = (off, shape)
        off = (off + int(np.prod(shape)) + 3) & ~3
    return out, off


def flat_size(O: int, A: int, H: int, L: int = 2) -> int:
    """Length of the FLAT vector the engine and this oracle exchange (include/minppo_hip.h): the parameters with every
    tensor starting on a 16-byte boundary; the alignment words hold zeros (zero gradient, zero update)."""
    return _slices(O, A, H, L)[1]


def param_slices(O: int, A: int, H: int, L: int = 2) -> Dict[str, Tuple[int, Tuple[int, ...]]]:
    """name -> (offset, shape) in the flat vector."""
    return _slices(O, A, H, L)[0]


def flat_to_named(flat: np.ndarray, O: int, A: int, H: int, L: int = 0) -> Dict[str, np.ndarray]:
    """L = 0: the depth whose flat size is len(flat) (1 .. 4 hidden layers)."""
    if L == 0:
        L = next(l for l in (2, 1, 3, 4) if flat_size(O, A, H, l) == flat.shape[0])
    return {k: flat[o:o + int(np.prod(s))].reshape(s) for k, (o, s) in param_slices(O, A, H, L).items()}


def named_to_flat(named: Dict[str, np.ndarray], O: int, A: int, H: int) -> np.ndarray:
    dt = named["a_w1"].dtype
    L = n_hidden(named)
    flat = np.zeros(flat_size(O, A, H, L), dt)
    for k, (o, s) in param_slices(O, A, H, L).items():
        flat[o:o + int(np.prod(s))] = np.asarray(named[k], dt).reshape(-1)
    return flat


def named_to_tree(p: Dict[str, np.ndarray]) -> dict:
    L = n_hidden(p)

    def mlp(pref):
        return {f"Dense_{i}": {"kernel": p[f"{pref}_w{i + 1}"], "bias": p[f"{pref}_b{i + 1}"]} for i in range(L + 1)}

    return {"params": {"MLP_0": mlp("a"), "log_std": p["log_std"], "MLP_1": mlp("c")}}


def tree_to_named(tree: dict) -> Dict[str, np.ndarray]:
    t = tree["params"]
    out = {"log_std": np.asarray(t["log_std"])}
    for pref, key in (("a", "MLP_0"), ("c", "MLP_1")):
        for i in range(len(t[key])):
            out[f"{pref}_w{i + 1}"] = np.asarray(t[key][f"Dense_{i}"]["kernel"])
            out[f"{pref}_b{i + 1}"] = np.asarray(t[key][f"Dense_{i}"]["bias"])
    return out


def orthogonal(rng: np.random.Generator, shape: Tuple[int, int], scale: float, dtype=np.float64) -> np.ndarray:
    """`flax.linen.initializers.orthogonal(scale)` semantics: QR of a
    normal matrix, sign-corrected by sign(diag(R)), transposed when in < out, times scale.
    (The random stream is NumPy's, not JAX's threefry: init parity is by explicit weights.)"""
    n_in, n_out = shape
    rows, cols = max(n_in, n_out), min(n_in, n_out)
    a = rng.standard_normal((rows, cols))
    q, r = np.linalg.qr(a)
    q = q * np.sign(np.diag(r))[None, :]
    if n_in < n_out:
        q = q.T
    return (scale * q).astype(dtype)


def init_params(seed: int, O: int, A: int, H: int, dtype=np.float64, L: int = 2) -> Dict[str, np.ndarray]:
    """Weights as `ActorCritic.init` makes them (train.py:63,68,80): hidden gain sqrt(2), heads 0.01, biases 0, log_std 0;
    `L` hidden layers per MLP (`MLP([hidden_size] * num_layers + [out])`, train.py:79,82)."""
    rng = np.random.default_rng(seed)
    g = math.sqrt(2.0)
    p = {}
    for pref, last in (("a", A), ("c", 1)):
        for i in range(L):
            p[f"{pref}_w{i + 1}"] = orthogonal(rng, (O if i == 0 else H, H), g, dtype)
            p[f"{pref}_b{i + 1}"] = np.zeros(H, dtype)
        p[f"{pref}_w{L + 1}"] = orthogonal(rng, (H, last), 0.01, dtype)
        p[f"{pref}_b{L + 1}"] = np.zeros(last, dtype)
    p["log_std"] = np.zeros(A, dtype)
    return p


# ---------------------------------------------------------------------------
# network (train.py:56-83)
# ---------------------------------------------------------------------------


def round_bf16(a: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even to bfloat16 precision (8 significant bits), result in the input's dtype.

    BASELINE configs[3] ("bf16 MLP MFMA path with fp32 GAE/Adam"): the engine feeds its matrix cores bf16 operands and
    accumulates in f32.  This is the operand rounding, so the oracle can follow the same arithmetic."""
    a32 = np.ascontiguousarray(a, dtype=np.float32)
    u = a32.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(a32.shape).astype(a.dtype)


def _mm(a, b, bf16: bool):
    """Matrix product of the MLP; with bf16=True both operands are rounded to bf16 first (products and sums stay in the
    array dtype, like bf16-in / f32-accumulate MFMA)."""
    return round_bf16(a) @ round_bf16(b) if bf16 else a @ b


def actor_critic_forward(p: Dict[str, np.ndarray], x: np.ndarray, use_tanh: bool = True, keep: bool = False, bf16: bool = False):
    """Returns (mean [n,A], log_std [A], value [n]) and, with keep=True, the hidden activations.

    Actor: tanh if use_tanh else relu (train.py:79); critic: always relu (train.py:82, quirk C-4).
    bf16: operand rounding of the two hidden-layer products (the engine keeps the small output-layer products in f32)."""
    act_a = np.tanh if use_tanh else (lambda z: np.maximum(z, 0))
    L = n_hidden(p)
    ha, hc = [], []  # `for feat in self.features[:-1]` (train.py:62-67), then the linear output layer (train.py:68)
    h = x
    for i in range(L):
        h = act_a(_mm(h, p[f"a_w{i + 1}"], bf16) + p[f"a_b{i + 1}"])
        ha.append(h)
    mean = h @ p[f"a_w{L + 1}"] + p[f"a_b{L + 1}"]
    h = x
    for i in range(L):
        h = np.maximum(_mm(h, p[f"c_w{i + 1}"], bf16) + p[f"c_b{i + 1}"], 0)
        hc.append(h)
    value = (h @ p[f"c_w{L + 1}"] + p[f"c_b{L + 1}"])[..., 0]
    if keep:
        return mean, p["log_std"], value, (ha, hc)
    return mean, p["log_std"], value


def mvn_log_prob(x, mean, log_std):
    """`distrax.MultivariateNormalDiag(mean, exp(log_std)).log_prob(x)` (train.py:81,160,223)."""
    z = (x - mean) * np.exp(-log_std)
    A = x.shape[-1]
    return -0.5 * np.sum(z * z, axis=-1) - np.sum(log_std) - 0.5 * A * LOG_2PI


def mvn_entropy(log_std):
    A = log_std.shape[-1]
    return 0.5 * A * (1.0 + LOG_2PI) + np.sum(log_std)


def mvn_sample(mean, log_std, eps):
    """sample = mean + std * eps, eps ~ N(0,1) supplied by the caller (RNG contract)."""
    return mean + np.exp(log_std) * eps


# ---------------------------------------------------------------------------
# GAE (train.py:185-205)
# ---------------------------------------------------------------------------


def calculate_gae(done, value, reward, last_val, gamma: float, lam: float):
    """done/value/reward [T,N], last_val [N] -> (advantages, targets) [T,N]."""
    T = value.shape[0]
    dt = value.dtype
    adv = np.zeros_like(value)
    gae = np.zeros_like(last_val)
    next_value = last_val
    g = dt.type(gamma)
    gl = dt.type(gamma) * dt.type(lam)
    for t in range(T - 1, -1, -1):
        nd = (1 - done[t].astype(np.int32)).astype(dt)  # bool -> int -> float (quirk C-11)
        delta = reward[t] + g * next_value * nd - value[t]
        gae = delta + gl * nd * gae
        adv[t] = gae
        next_value = value[t]
    return adv, adv + value


# ---------------------------------------------------------------------------
# loss and its gradient (train.py:218-243, 246)
# ---------------------------------------------------------------------------


class LossOut(NamedTuple):
    total: float
    value_loss: float
    actor_loss: float
    entropy: float


def loss_and_grad(p, obs, action, old_value, old_logp, gae, targets, clip_eps=0.2, vf_coef=0.5, ent_coef=0.0,
                  use_tanh=True, adv_mean=None, adv_std=None, inv_count=None, bf16=False):
    """Clipped-PPO loss on one minibatch and d(total)/d(params) as a named dict.

    Hand-derived backward (the HIP kernels implement the same formulas); the test-suite
    checks it against torch autograd.  `adv_mean/adv_std/inv_count` override the
    per-minibatch statistics for the sharded (multi-GPU) equivalence test: there the
    statistics and the 1/mb factor are global over ranks.
    bf16: bf16 operand rounding in the hidden-layer products, the dZ.W2^T back-product and all weight-gradient products
    (what the engine does with training.mlp_dtype = "bf16"); the output-layer products h2.W3 and dOut.W3^T stay exact."""
    dt = obs.dtype
    n = obs.shape[0]
    mean, log_std, value, (ha, hc) = actor_critic_forward(p, obs, use_tanh, keep=True, bf16=bf16)
    L = len(ha)
    A = action.shape[-1]
    inv_std = np.exp(-log_std)
    z = (action - mean) * inv_std
    logp = -0.5 * np.sum(z * z, -1) - np.sum(log_std) - dt.type(0.5 * A * LOG_2PI)
    inv_n = dt.type(1.0 / n) if inv_count is None else dt.type(inv_count)

    # value loss (train.py:226-231)
    v_clip = old_value + np.clip(value - old_value, -clip_eps, clip_eps)
    vl1 = (value - targets) ** 2
    vl2 = (v_clip - targets) ** 2
    value_loss = dt.type(0.5) * np.sum(np.maximum(vl1, vl2)) * inv_n

    # actor loss (train.py:234-239); population std, eps added to std (quirk C-8)
    ratio = np.exp(logp - old_logp)
    m = gae.mean() if adv_mean is None else dt.type(adv_mean)
    s = gae.std() if adv_std is None else dt.type(adv_std)
    g = (gae - m) / (s + dt.type(1e-8))
    la1 = ratio * g
    la2 = np.clip(ratio, 1.0 - clip_eps, 1.0 + clip_eps) * g
    actor_loss = -np.sum(np.minimum(la1, la2)) * inv_n
    entropy = mvn_entropy(log_std)
    total = actor_loss + vf_coef * value_loss - ent_coef * entropy

    # ---- backward ----
    # d(-min(la1,la2))/d ratio: inside the clip range both arms are equal (JAX splits the
    # tie 0.5/0.5 and the clip passes gradient -> g in total); outside, the clipped arm is
    # selected exactly when it is the smaller one, and it carries no gradient.
    unclipped = (ratio >= 1.0 - clip_eps) & (ratio <= 1.0 + clip_eps)
    use1 = unclipped | (la1 < la2)
    dlogp = np.where(use1, -g * ratio, 0.0).astype(dt) * inv_n
    # d(0.5*max(vl1,vl2))/d value
    vin = np.abs(value - old_value) <= clip_eps
    dv = np.where(vin | (vl1 > vl2), value - targets, 0.0).astype(dt) * inv_n * dt.type(vf_coef)

    dmean = dlogp[:, None] * z * inv_std  # d logp / d mean = (a-mean)/std^2
    dlog_std = np.sum(dlogp[:, None] * (z * z - 1.0), axis=0) - dt.type(ent_coef)

    grads = {}
    # bf16 model: the engine's row pass hands dZ (and the activations) to the weight-gradient launch ROUNDED to bf16 - they are MFMA
    # operands there - and the bias gradients are column sums of those same stored values
    bsum = (lambda d: round_bf16(d).sum(0)) if bf16 else (lambda d: d.sum(0))
    # output layers: exact products in the engine (module docstring); then dZ_l = (dZ_{l+1} . W_{l+1}^T) * act'(h_l) down the hidden layers
    for pref, hs, dout, tanh_net in (("a", ha, dmean, use_tanh), ("c", hc, dv[:, None], False)):
        grads[f"{pref}_w{L + 1}"] = _mm(hs[L - 1].T, dout, bf16)
        grads[f"{pref}_b{L + 1}"] = bsum(dout)
        dz = (dout @ p[f"{pref}_w{L + 1}"].T) * ((1 - hs[L - 1] * hs[L - 1]) if tanh_net else (hs[L - 1] > 0))
        for i in range(L - 1, -1, -1):  # hidden layer i (0-based): weights {pref}_w{i+1}
            prev = hs[i - 1] if i > 0 else obs
            grads[f"{pref}_w{i + 1}"] = _mm(prev.T, dz, bf16)
            grads[f"{pref}_b{i + 1}"] = bsum(dz)
            if i > 0:
                dz = _mm(dz, p[f"{pref}_w{i + 1}"].T, bf16) * ((1 - hs[i - 1] * hs[i - 1]) if tanh_net else (hs[i - 1] > 0))
    grads["log_std"] = dlog_std
    grads = {k: v.astype(dt) for k, v in grads.items()}
    return LossOut(float(total), float(value_loss), float(actor_loss), float(entropy)), grads


# ---------------------------------------------------------------------------
# optimizer (train.py:98-101, 115-124)
# ---------------------------------------------------------------------------


def linear_schedule(count: int, lr: float, minibatch_size: int, update_epochs: int, num_updates: int) -> float:
    """train.py:98-101 as written: the divisor is minibatch_size*update_epochs (quirk C-2)."""
    frac = 1.0 - (count // (minibatch_size * update_epochs)) / num_updates
    return lr * frac


def clip_by_global_norm(grad_flat: np.ndarray, max_norm: float) -> Tuple[np.ndarray, float]:
    """optax.clip_by_global_norm: g if ||g|| < c else g*c/||g||."""
    dt = grad_flat.dtype
    norm = np.sqrt(np.sum(grad_flat * grad_flat))
    if norm < max_norm:
        return grad_flat, float(norm)
    return grad_flat * (dt.type(max_norm) / norm), float(norm)


def adam_step(p, m, v, g, count: int, lr: float, b1=0.9, b2=0.999, eps=1e-5):
    """optax.adam (scale_by_adam + scale_by_learning_rate): `count` is the pre-increment step index."""
    dt = p.dtype
    t = count + 1
    m = dt.type(b1) * m + dt.type(1 - b1) * g
    v = dt.type(b2) * v + dt.type(1 - b2) * g * g
    mhat = m / dt.type(1 - b1 ** t)
    vhat = v / dt.type(1 - b2 ** t)
    p = p - dt.type(lr) * mhat / (np.sqrt(vhat) + dt.type(eps))
    return p, m, v


class OptState(NamedTuple):
    m: np.ndarray
    v: np.ndarray
    count: int


def optimizer_update(flat_p, opt: OptState, flat_g, *, max_grad_norm, anneal_lr, lr_train, lr_opt,
                     minibatch_size, update_epochs, num_updates):
    g, _ = clip_by_global_norm(flat_g, max_grad_norm)
    lr = linear_schedule(opt.count, lr_train, minibatch_size, update_epochs, num_updates) if anneal_lr else lr_opt
    p, m, v = adam_step(flat_p, opt.m, opt.v, g, opt.count, lr)
    return p, OptState(m, v, opt.count + 1)


# ---------------------------------------------------------------------------
# one epoch / one update on a stored trajectory (train.py:209-274)
# ---------------------------------------------------------------------------


def update_epochs_on_batch(flat_p, opt: OptState, traj: Dict[str, np.ndarray], adv, targets, perms: np.ndarray,
                           *, O, A, H, num_minibatches, hp: dict, use_tanh=True):
    """Runs `len(perms)` epochs of shuffled-minibatch updates.

    traj arrays are time-major [T,N,...]; flattening is row = t*N + n (train.py:260);
    minibatch k = rows perm[k*mb:(k+1)*mb] (train.py:261-265).  Returns new params,
    optimizer state and the per-step losses [E,M,4] (total, value, actor, entropy)."""
    T, N = traj["value"].shape
    B = T * N
    mb = B // num_minibatches
    flat = {k: traj[k].reshape((B,) + traj[k].shape[2:]) for k in ("obs", "action", "value", "log_prob")}
    adv_f, tgt_f = adv.reshape(B), targets.reshape(B)
    losses = np.zeros((len(perms), num_minibatches, 4))
    for e, perm in enumerate(perms):
        for k in range(num_minibatches):
            idx = perm[k * mb:(k + 1) * mb]
            named = flat_to_named(flat_p, O, A, H)
            lo, grads = loss_and_grad(named, flat["obs"][idx], flat["action"][idx], flat["value"][idx],
                                      flat["log_prob"][idx], adv_f[idx], tgt_f[idx],
                                      hp["clip_eps"], hp["vf_coef"], hp["ent_coef"], use_tanh, bf16=bool(hp.get("mlp_bf16", False)))
            g = named_to_flat(grads, O, A, H)
            flat_p, opt = optimizer_update(flat_p, opt, g, max_grad_norm=hp["max_grad_norm"], anneal_lr=hp["anneal_lr"],
                                           lr_train=hp["lr_train"], lr_opt=hp["lr_opt"], minibatch_size=mb,
                                           update_epochs=hp["update_epochs"], num_updates=hp["num_updates"])
            losses[e, k] = lo
    return flat_p, opt, losses
