"""Differential test of the PPO minibatch step over RANDOM shapes: observation width, action dimension, hidden width, minibatch size (ragged:
not a multiple of the 16-row tile, smaller than a tile, a single row), batch size, activation, entropy coefficient, float and bf16 products -
`mppo_minibatch_grad` (fused row pass + weight-gradient kernel where the geometry allows it, the layer-wise kernels elsewhere) against the
NumPy float64 oracle (`oracle/ppo_oracle.py loss_and_grad`, reference train.py:218-247).  The hand-picked shapes of tests/test_kernels_ppo.py
sit on the known edges; these land where nobody looked (a 3-wide observation, 29 actions on a 224-wide layer, 17 rows ...)."""
import ctypes as C
import os

import numpy as np
import pytest

from minppo_amd import _native as nat
from oracle import ppo_oracle as po
from test_kernels_ppo import _net, _params

f32 = np.float32


def _case(seed):
    rng = np.random.default_rng(7000 + seed)
    O = int(rng.choice([rng.integers(3, 40), rng.integers(40, 260)]))
    A = int(rng.choice([1, 2, rng.integers(3, 12), rng.integers(12, 33)]))
    H = int(rng.choice([32, 64, 96, 128, 160, 192, 224, 256, 40, 72, 200]))
    mb = int(rng.choice([1, rng.integers(2, 16), 16, 17, rng.integers(18, 64), rng.integers(64, 150)]))
    return O, A, H, mb + int(rng.integers(0, 40)), mb, int(rng.integers(0, 2)), float(rng.choice([0.0, 0.01])), bool(rng.random() < 0.35)


@pytest.mark.parametrize("seed", range(int(os.environ.get("MPPO_FUZZ_SHAPES", "20"))))   # (MPPO_FUZZ_SHAPES=300: the hunt DESIGN.md section 5 reports)
def test_minibatch_step_on_a_random_shape(be, seed):
    O, A, H, B, mb, tanh, ent, bf16 = _case(seed)
    rng = np.random.default_rng(seed)
    net = _net(O, A, H, tanh)
    net.bf16 = 1 if bf16 else 0
    OP = net.OP
    flat, n64 = _params(rng, O, A, H)
    bobs = np.zeros((B, OP), f32); bobs[:, :O] = rng.standard_normal((B, O))
    bact = rng.standard_normal((B, A)).astype(f32)
    m_, ls_, v_ = po.actor_critic_forward(n64, bobs[:, :O].astype(np.float64), bool(tanh))
    bval = (v_ + 0.3 * rng.standard_normal(B)).astype(f32)
    blp = (po.mvn_log_prob(bact.astype(np.float64), m_, ls_) + 0.3 * rng.standard_normal(B)).astype(f32)
    badv = (rng.standard_normal(B) * 3 + 1).astype(f32)
    btgt = rng.standard_normal(B).astype(f32)
    idx = rng.permutation(B)[:mb].astype(np.int32)
    d = {k: be.arr(v) for k, v in dict(flat=flat, obs=bobs, act=bact, val=bval, lp=blp, adv=badv, tgt=btgt, idx=idx).items()}
    g = badv[idx].astype(np.float64)
    stats_np = np.array([g.mean(), 1 / (g.std() + 1e-8)], f32)
    stats = be.arr(stats_np)
    batch = nat.Batch(be.ptr(d["obs"]), OP, be.ptr(d["act"]), A, be.ptr(d["val"]), be.ptr(d["lp"]), be.ptr(d["adv"]), be.ptr(d["tgt"]))
    lc = nat.LossCfg(0.2, 0.5, ent)
    grad, loss4 = be.full((flat.size,), np.nan), be.zeros((4,))
    wsb = be.lib.grad_ws_bytes(C.byref(net), mb)
    ws = be.full((wsb // 4 + 4,), np.nan)
    call = lambda: be.lib.minibatch_grad(C.byref(net), be.ptr(d["flat"]), C.byref(batch), be.ptr(d["idx"]), mb, be.ptr(stats), 1.0 / mb, C.byref(lc), be.ptr(grad),
                                         be.ptr(loss4), be.ptr(ws), wsb, be.stream)
    if bf16 and not (H % 32 == 0 and H <= 256 and A <= 32):
        with pytest.raises(nat.NativeError, match="bf16 products need the fused kernels"):   # the layer-wise path is float only: said up front
            call()
        return
    call()
    lo, gr = po.loss_and_grad(n64, bobs[idx][:, :O].astype(np.float64), bact[idx].astype(np.float64), bval[idx].astype(np.float64), blp[idx].astype(np.float64),
                              g, btgt[idx].astype(np.float64), 0.2, 0.5, ent, bool(tanh), bf16=bf16)
    got, g64 = be.host(grad), po.named_to_flat(gr, O, A, H)
    assert not np.isnan(got).any(), (O, A, H, mb, bf16)
    # (bf16: the tolerances of test_bf16_mfma_path - operands rounded on both sides, only boundary flips of single operands remain; float: a
    # three-row minibatch's loss sums carry a few 1e-6 of cancellation)
    np.testing.assert_allclose(be.host(loss4), lo, rtol=2e-3 if bf16 else 5e-5, atol=1e-5 if bf16 else 5e-6, err_msg=str((O, A, H, mb, bf16)))
    for k, (o, s) in po.param_slices(O, A, H).items():
        sz = int(np.prod(s))
        tol = (5e-3 if bf16 else 1e-4) * np.abs(g64[o:o + sz]).max() + 1e-7
        if bf16:   # a single operand whose float32 and float64 values round to different bf16 neighbours moves single gradient entries by ~1 %
            err = np.abs(got[o:o + sz] - g64[o:o + sz])
            assert (err > tol).mean() <= 2e-3 and err.max() <= 6 * tol, (k, (O, A, H, mb), float((err > tol).mean()), float(err.max() / tol))
        else:
            np.testing.assert_allclose(got[o:o + sz], g64[o:o + sz], rtol=0, atol=tol, err_msg=f"{k} at {(O, A, H, mb, bf16)}")

