"""Committed fixtures (tests/golden/, made by tests/golden/make_golden.py from the float64 oracle):
  * the oracle still reproduces them (CPU; pins the oracle),
  * the kernels reproduce them through the C ABI (emulator on CPU, MI355X with -m gpu)."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

from minppo_amd import _native as nat
from minppo_amd.model import load_model
from oracle import ppo_oracle as po
from oracle.env_oracle import EnvOracle, RewardCfg

G = Path(__file__).parent / "golden"
f32 = np.float32
HP = dict(clip_eps=0.2, vf_coef=0.5, ent_coef=0.0, max_grad_norm=0.5, anneal_lr=True, lr_train=3e-4, lr_opt=3e-4, update_epochs=2, num_updates=1000)


def test_oracle_reproduces_ppo_golden():
    g = np.load(G / "ppo_small.npz")
    N, T, O, A, H, M, E = g["dims"]
    adv, tgt = po.calculate_gae(g["done"], g["value"], g["reward"], g["last_val"], 0.99, 0.95)
    np.testing.assert_allclose(adv, g["adv"], rtol=1e-13); np.testing.assert_allclose(tgt, g["target"], rtol=1e-13)
    traj = {k: g[k] for k in ("obs", "action", "value", "log_prob")}
    p1, opt, losses = po.update_epochs_on_batch(g["params0"], po.OptState(np.zeros_like(g["params0"]), np.zeros_like(g["params0"]), 0), traj, adv, tgt,
                                                g["perms"], O=O, A=A, H=H, num_minibatches=M, hp=HP)
    np.testing.assert_allclose(p1, g["params1"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-12)
    np.testing.assert_allclose(opt.m, g["adam_m"], rtol=1e-11, atol=1e-16)


def test_oracle_reproduces_physics_golden():
    g = np.load(G / "physics_steps.npz")
    cm = load_model("synth_stompy_pro")
    env = EnvOracle(cm.t, RewardCfg(height_min_z=float(g["height_min_z"])))
    N = g["qpos"].shape[1]
    es = env.reset(N)
    for t in range(g["qpos"].shape[0]):
        s = es["pipeline_state"]
        if t == 3:
            s["qvel"][2, 2] = -30.0
        np.testing.assert_allclose(s.qpos, g["qpos"][t], atol=1e-11)
        es = env.step(es, g["action"][t])
        np.testing.assert_allclose(es["obs"], g["obs"][t], atol=1e-9)
        np.testing.assert_allclose(es["reward"], g["reward"][t], atol=1e-8)
        assert (es["done"] == g["done"][t]).all()
    assert g["done"].any()


def test_kernels_reproduce_ppo_golden(be):
    g = np.load(G / "ppo_small.npz")
    N, T, O, A, H, M, E = [int(x) for x in g["dims"]]
    B, mb = N * T, N * T // M
    net = nat.Net(O, O, A, H, 1, 0)
    d = {k: be.arr(g[k].astype(f32)) for k in ("reward", "value", "last_val", "log_prob", "obs", "action", "params0")}
    d["done"] = be.arr(g["done"].astype(np.uint8))
    adv, tgt = be.zeros((T, N)), be.zeros((T, N))
    be.lib.gae(T, N, 0.99, 0.95, be.ptr(d["reward"]), be.ptr(d["value"]), be.ptr(d["done"]), be.ptr(d["last_val"]), be.ptr(adv), be.ptr(tgt), be.stream)
    np.testing.assert_allclose(be.host(adv), g["adv"], rtol=1e-5, atol=1e-5)
    perms = be.arr(g["perms"].astype(np.int32))
    sums, stats = be.zeros((E * M * 2,), np.float64), be.zeros((E * M * 2,))
    be.lib.adv_sums(be.ptr(adv), be.ptr(perms), E * M, mb, be.ptr(sums), be.stream)
    be.lib.adv_stats_finalize(be.ptr(sums), E * M, float(mb), be.ptr(stats), be.stream)
    batch = nat.Batch(be.ptr(d["obs"]), O, be.ptr(d["action"]), A, be.ptr(d["value"]), be.ptr(d["log_prob"]), be.ptr(adv), be.ptr(tgt))
    lc = nat.LossCfg(0.2, 0.5, 0.0)
    P = g["params0"].size
    p, m, v, grad, loss4 = be.arr(g["params0"].astype(f32)), be.zeros((P,)), be.zeros((P,)), be.zeros((P,)), be.zeros((E * M, 4))
    cnt = be.zeros((4,), np.int32)
    wsb = be.lib.grad_ws_bytes(C.byref(net), mb)
    awsb = be.lib.adam_ws_bytes(P)
    ws, aws = be.zeros((wsb // 4 + 4,)), be.zeros((awsb // 4,))
    cfg = nat.AdamCfg(3e-4, 0.5, 0.9, 0.999, 1e-5, 1, mb * E, 1000)
    l4h = be.host(loss4)
    for e in range(E):
        for k in range(M):
            st = e * M + k
            off = (e * B + k * mb) * 4
            be.lib.minibatch_grad(C.byref(net), be.ptr(p), C.byref(batch), be.ptr(perms) + off, mb, be.ptr(stats) + 8 * st, 1.0 / mb, C.byref(lc),
                                  be.ptr(grad), be.ptr(loss4) + 16 * st, be.ptr(ws), wsb, be.stream)
            if st == 0:
                np.testing.assert_allclose(be.host(grad), g["grad0"], rtol=0, atol=2e-5 * np.abs(g["grad0"]).max())
            be.lib.clip_adam(P, be.ptr(p), be.ptr(m), be.ptr(v), be.ptr(grad), be.ptr(cnt), st, C.byref(cfg), be.ptr(aws), awsb, be.stream)
    np.testing.assert_allclose(be.host(loss4), g["losses"].reshape(-1, 4), rtol=2e-4, atol=1e-5)
    step = np.abs(g["params1"] - g["params0"]).max()
    assert np.abs(be.host(p) - g["params1"]).max() < 1e-3 * step  # params after E*M = 8 Adam steps
    np.testing.assert_allclose(be.host(m), g["adam_m"], rtol=1e-3, atol=1e-7)


def test_kernels_reproduce_physics_golden(be):
    g = np.load(G / "physics_steps.npz")
    cm = load_model("synth_stompy_pro")
    h, dims, _keep = be.model(cm)
    K, N = g["qpos"].shape[:2]
    O, OP, R, nv, nu, nq = dims.obs_dim, dims.obs_pad, dims.rec_dim, cm.nv, cm.nu, cm.nq
    state, reset_rec, obs = be.zeros((N, R)), be.zeros((R,)), be.zeros((N, OP))
    rew, done = be.zeros((N,)), be.zeros((N,), np.uint8)
    be.lib.env_reset(h, N, be.ptr(state), be.ptr(reset_rec), be.ptr(obs), OP, be.ptr(rew), be.ptr(done), None, be.stream)
    rc = nat.RewardCfg(float(g["height_min_z"]), 2.0, 2.0, 0.2, 0.5, 0.1, 4.0, 1.0, 1.25)
    for t in range(K):
        rec = np.zeros((N, R), f32)
        rec[:, :nq] = g["qpos"][t]; rec[:, nq:nq + nv] = g["qvel"][t]
        rec[:, nq + nv:nq + nv + 110] = g["cinert"][t][:, 1:].reshape(N, -1)
        rec[:, nq + nv + 110:nq + nv + 176] = g["cvel"][t][:, 1:].reshape(N, -1)
        rec[:, nq + nv + 176:O] = g["qact"][t]
        rec[:, OP:OP + nv] = g["warm"][t]; rec[:, OP + nv] = g["comx"][t]; rec[:, OP + nv + 1] = g["time"][t]
        be.put(state, rec)
        a = be.arr(g["action"][t].astype(f32))
        be.lib.env_step(h, N, 1, C.byref(rc), be.ptr(state), be.ptr(reset_rec), be.ptr(a), nu, be.ptr(obs), OP, be.ptr(rew), be.ptr(done), None, be.stream)
        assert (be.host(done).astype(bool) == g["done"][t]).all()
        np.testing.assert_allclose(be.host(obs)[:, :O], g["obs"][t], atol=1e-4)
        np.testing.assert_allclose(be.host(rew), g["reward"][t], atol=1e-2)
        np.testing.assert_allclose(be.host(state)[:, :nq], g["qpos1"][t], atol=2e-3)
        # forward intermediates at the same state
        outs = {k: be.zeros(g[k][t].shape) for k in ("qM", "efc_J", "efc_aref", "efc_D", "qacc_smooth", "qfrc_bias")}
        qacc = be.zeros((N, nv))
        pr = nat.ForwardProbe(**{k: be.ptr(v) for k, v in outs.items()}, qacc=be.ptr(qacc))
        ins = [be.arr(g[k][t].astype(f32)) for k in ("qpos", "qvel", "action", "warm")]
        be.lib.physics_forward(h, N, *[be.ptr(x) for x in ins], C.byref(pr), be.stream)
        for k, tol in (("qM", 1e-5), ("efc_J", 1e-5), ("efc_aref", 5e-4), ("efc_D", 5e-4), ("qacc_smooth", 2e-4), ("qfrc_bias", 1e-4)):
            ref = g[k][t]
            assert np.abs(be.host(outs[k]) - ref).max() <= tol * (np.abs(ref).max() + 1e-6), k
        # solver: cost reached (float64 evaluation with the golden matrices)
        q = be.host(qacc).astype(np.float64)
        jar = np.einsum("nrv,nv->nr", g["efc_J"][t], q) - g["efc_aref"][t]
        Ma = np.einsum("nij,nj->ni", g["qM"][t], q)
        qfs = np.einsum("nij,nj->ni", g["qM"][t], g["qacc_smooth"][t])
        cost = 0.5 * np.sum(g["efc_D"][t] * jar * jar * (jar < 0), -1) + 0.5 * np.sum((Ma - qfs) * (q - g["qacc_smooth"][t]), -1)
        np.testing.assert_allclose(cost, g["cost"][t], rtol=5e-2, atol=1e-3)
    be.lib.model_close(h)
