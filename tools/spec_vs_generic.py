"""Specialised against run-time-sized environment kernel on the GPU, field by field of the forward probe, then env steps (argv: library, model, [N]).
For a specialised variant build (MPPO_SPECIALIZE=<model> python -m minppo_amd.build) that tests/test_kernels_physics.py does not cover."""
import ctypes as C, os, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
from minppo_amd import _native as nat
nat.HIP_LIB_PATH = Path(sys.argv[1]).resolve()
from backends import HipBackend
from minppo_amd.model import load_model
from test_kernels_physics import _probe, _walk
be = HipBackend()
cm = load_model(sys.argv[2])
N = int(sys.argv[3]) if len(sys.argv) > 3 else 9
ph, d, rng = _walk(cm, N, 5, 8)
ctrl = 0.4 * rng.standard_normal((N, max(cm.nu, 1)))
q32 = [x.astype(np.float32) for x in (d.qpos, d.qvel, ctrl, d.qacc_warmstart)]
res = []
for generic in (False, True):
    for waves in ((None, "1") if not generic else (None,)):
        os.environ.pop("MPPO_ENV_GENERIC", None); os.environ.pop("MPPO_ENV_WAVES", None)
        if generic: os.environ["MPPO_ENV_GENERIC"] = "1"
        if waves: os.environ["MPPO_ENV_WAVES"] = waves
        h, dims, _keep = be.model(cm)
        flag = C.c_int32(-1); be.lib.model_is_specialized(h, C.byref(flag))
        got = _probe(be, h, cm, *q32)
        OP, R = dims.obs_pad, dims.rec_dim
        state, reset_rec, obs = be.zeros((N, R)), be.zeros((R,)), be.zeros((N, OP))
        rew, done = be.zeros((N,)), be.zeros((N,), np.uint8)
        be.lib.env_reset(h, N, be.ptr(state), be.ptr(reset_rec), be.ptr(obs), OP, be.ptr(rew), be.ptr(done), None, be.stream)
        got["reset_state"] = be.host(state).copy()
        rc = nat.RewardCfg(-0.2, 2.0, 2.0, 0.2, 0.5, 0.1, 4.0, 1.0, 1.25)
        r2 = np.random.default_rng(3)
        for _ in range(3):
            act = be.arr((0.8 * r2.standard_normal((N, max(cm.nu, 1)))).astype(np.float32))
            be.lib.env_step(h, N, 1, C.byref(rc), be.ptr(state), be.ptr(reset_rec), be.ptr(act), max(cm.nu, 1), be.ptr(obs), OP, be.ptr(rew), be.ptr(done), None, be.stream)
            be.sync()
        got.update(state=be.host(state).copy(), done=be.host(done).copy())
        res.append((f"specialised={flag.value} waves={waves} lds={dims.lds_bytes}", got))
        be.lib.model_close(h)
ref = res[-1][1]
for name, got in res[:-1]:
    print(name)
    for k in got:
        a, b = np.asarray(got[k], np.float64), np.asarray(ref[k], np.float64)
        neq = ~((a == b) | (np.isnan(a) & np.isnan(b)))
        print(f"   {k:14s} differing {int(neq.sum()):7d} of {a.size:7d}   nan {int(np.isnan(a).sum())} / {int(np.isnan(b).sum())}   max |diff| {np.nanmax(np.abs(a - b)) if a.size else 0:.3e}")
if os.environ.get("SPEC_DEBUG"):
    a, b = res[0][1], ref
    nq, nv = cm.nq, cm.nv
    print("done spec", a["done"], "generic", b["done"])
    print("spec state == reset record:", np.array_equal(a["state"], a["reset_state"]))
    for nm, g_ in (("spec", a), ("generic", b)):
        s0 = g_["state"][0]
        print(nm, "qpos[:7]", s0[:7], "qvel[:6]", s0[nq:nq + 6], "tail", s0[-8:])
    d = np.abs(a["state"] - b["state"])[0]
    print("largest differences of env 0 at record indices", np.argsort(-d)[:12], d[np.argsort(-d)[:12]])
