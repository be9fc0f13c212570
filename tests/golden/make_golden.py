"""Generates the committed fixtures under tests/golden/ (run from the repo root: python tests/golden/make_golden.py).

The reference ships no golden vectors and cannot be imported here, so these fixtures are
produced by the float64 oracle (oracle/) — they pin the oracle against regressions and give the GPU
parity tests fixed inputs / expected outputs that travel with the repository — plus one fixture that pins
the engine's own Philox stream (generated with the emulator build of the kernel sources).
"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
OUT = Path(__file__).resolve().parent

from minppo_amd.model import load_model  # noqa: E402
from oracle import ppo_oracle as po  # noqa: E402
from oracle.env_oracle import EnvOracle, RewardCfg  # noqa: E402
from oracle.physics_oracle import Physics  # noqa: E402


def ppo_small():
    """N=8, T=10, O=12, A=3, H=16 — one epoch-set of shuffled minibatch updates."""
    rng = np.random.default_rng(20241016)
    N, T, O, A, H, M, E = 8, 10, 12, 3, 16, 4, 2
    named = po.init_params(1337, O, A, H)
    flat = po.named_to_flat(named, O, A, H)
    traj = dict(obs=rng.standard_normal((T, N, O)), action=rng.standard_normal((T, N, A)), reward=rng.standard_normal((T, N)),
                done=rng.random((T, N)) < 0.15)
    mean, ls, val = po.actor_critic_forward(named, traj["obs"].reshape(-1, O))
    traj["value"] = (val + 0.2 * rng.standard_normal(T * N)).reshape(T, N)
    traj["log_prob"] = (po.mvn_log_prob(traj["action"].reshape(-1, A), mean, ls) + 0.2 * rng.standard_normal(T * N)).reshape(T, N)
    last_val = rng.standard_normal(N)
    adv, tgt = po.calculate_gae(traj["done"], traj["value"], traj["reward"], last_val, 0.99, 0.95)
    perms = np.stack([rng.permutation(T * N) for _ in range(E)])
    hp = dict(clip_eps=0.2, vf_coef=0.5, ent_coef=0.0, max_grad_norm=0.5, anneal_lr=True, lr_train=3e-4, lr_opt=3e-4, update_epochs=E, num_updates=1000)
    idx0 = perms[0][:T * N // M]
    fo = {k: traj[k].reshape((T * N,) + traj[k].shape[2:]) for k in ("obs", "action", "value", "log_prob")}
    lo, gr = po.loss_and_grad(named, fo["obs"][idx0], fo["action"][idx0], fo["value"][idx0], fo["log_prob"][idx0], adv.reshape(-1)[idx0], tgt.reshape(-1)[idx0])
    p1, opt, losses = po.update_epochs_on_batch(flat, po.OptState(np.zeros_like(flat), np.zeros_like(flat), 0), traj, adv, tgt, perms, O=O, A=A, H=H,
                                                num_minibatches=M, hp=hp)
    np.savez_compressed(OUT / "ppo_small.npz", params0=flat, obs=traj["obs"], action=traj["action"], reward=traj["reward"], done=traj["done"],
                        value=traj["value"], log_prob=traj["log_prob"], last_val=last_val, adv=adv, target=tgt, perms=perms,
                        loss0=np.array(lo), grad0=po.named_to_flat(gr, O, A, H), params1=p1, adam_m=opt.m, adam_v=opt.v, losses=losses,
                        dims=np.array([N, T, O, A, H, M, E]))


def physics_steps():
    """synth_stompy_pro, 6 envs: a short walk, then K single steps each starting from a recorded state."""
    cm = load_model("synth_stompy_pro")
    rcfg = RewardCfg(height_min_z=0.95)
    env = EnvOracle(cm.t, rcfg)
    N, K = 6, 6
    rng = np.random.default_rng(7)
    es = env.reset(N)
    recs = dict(qpos=[], qvel=[], warm=[], cinert=[], cvel=[], qact=[], comx=[], time=[], action=[], obs=[], reward=[], done=[], qpos1=[], qvel1=[],
                cost=[], qM=[], efc_J=[], efc_aref=[], efc_D=[], qacc_smooth=[], qfrc_bias=[])
    for t in range(K):
        a = 0.6 * rng.standard_normal((N, cm.nu))
        s = es["pipeline_state"]
        if t == 3:
            s["qvel"][2, 2] = -30.0
        for k, v in (("qpos", s.qpos), ("qvel", s.qvel), ("warm", s.qacc_warmstart), ("cinert", s.cinert), ("cvel", s.cvel), ("qact", s.qfrc_actuator),
                     ("comx", s.subtree_com[:, 1, 0]), ("time", s.time), ("action", a)):
            recs[k].append(np.array(v))
        # forward quantities at this state (before integration), for the probe entry point
        f = s.copy(); f["ctrl"] = a; env.ph.forward(f)
        jar = np.einsum("nrv,nv->nr", f.efc_J, f.qacc) - f.efc_aref
        Ma = np.einsum("nij,nj->ni", f.qM, f.qacc)
        recs["cost"].append(0.5 * np.sum(f.efc_D * jar * jar * (jar < 0), -1) + 0.5 * np.sum((Ma - f.qfrc_smooth) * (f.qacc - f.qacc_smooth), -1))
        for k in ("qM", "efc_J", "efc_aref", "efc_D", "qacc_smooth", "qfrc_bias"):
            recs[k].append(np.array(f[k]))
        es = env.step(es, a)
        recs["obs"].append(es["obs"]); recs["reward"].append(es["reward"]); recs["done"].append(es["done"])
        recs["qpos1"].append(es["pipeline_state"].qpos); recs["qvel1"].append(es["pipeline_state"].qvel)
    np.savez_compressed(OUT / "physics_steps.npz", **{k: np.stack(v) for k, v in recs.items()}, height_min_z=np.array(rcfg.height_min_z))


def philox():
    from backends import get_backend

    be = get_backend("emu")
    z = be.zeros((8,)); be.lib.normal_fill(1337, 7, 8, be.ptr(z), None)
    idx = be.zeros((16,), np.int32); wsb = be.lib.permutation_ws_bytes(16)
    be.lib.permutation(1337, 3, 16, be.ptr(idx), be.ptr(be.zeros((wsb // 4 + 4,), np.int32)), wsb, None)
    np.savez(OUT / "philox.npz", normal8=np.array(z), perm16=np.array(idx))


if __name__ == "__main__":
    ppo_small(); physics_steps(); philox()
    for f in sorted(OUT.glob("*.npz")):
        print(f.name, f.stat().st_size, "bytes")
