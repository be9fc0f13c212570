"""A/B of the fixed-size and the run-time-sized env_kernel instantiations on the GPU: first step / field where they differ."""
import ctypes as C, os, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1])); sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
from backends import get_backend
from minppo_amd import _native as nat
from minppo_amd.model import load_model
be = get_backend("hip")
f32 = np.float32
model = sys.argv[1] if len(sys.argv) > 1 else "synth_stompy_pro"
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cm = load_model(model)
N = 9
hs = []
for generic in ((False, False) if os.environ.get("DBG_SAME") else (False, True)):
    if generic: os.environ["MPPO_ENV_GENERIC"] = "1"
    else: os.environ.pop("MPPO_ENV_GENERIC", None)
    h, dims, keep = be.model(cm)
    hs.append((h, dims, keep))
OP, R = hs[0][1].obs_pad, hs[0][1].rec_dim
st = [dict(state=be.zeros((N, R)), reset=be.zeros((R,)), obs=be.zeros((N, OP)), rew=be.zeros((N,)), done=be.zeros((N,), np.uint8)) for _ in range(2)]
for (h, _, _), s in zip(hs, st):
    be.lib.env_reset(h, N, be.ptr(s["state"]), be.ptr(s["reset"]), be.ptr(s["obs"]), OP, be.ptr(s["rew"]), be.ptr(s["done"]), None, be.stream)
print("reset equal:", {k: bool(np.array_equal(be.host(st[0][k]), be.host(st[1][k]), equal_nan=True)) for k in st[0]})
rc = nat.RewardCfg(0.95, 2.0, 2.0, 0.2, 0.5, 0.1, 4.0, 1.0, 1.25)
r2 = np.random.default_rng(3)
nq, nv = cm.nq, cm.nv
for t in range(8):
    a = (0.8 * r2.standard_normal((N, cm.nu))).astype(f32)
    # same input state for both: copy A's state into B
    if os.environ.get("DBG_RESEED", "1") == "1":
        be.put(st[1]["state"], be.host(st[0]["state"]))
    for (h, _, _), s in zip(hs, st):
        act = be.arr(a)
        be.lib.env_step(h, N, nf, C.byref(rc), be.ptr(s["state"]), be.ptr(s["reset"]), be.ptr(act), cm.nu, be.ptr(s["obs"]), OP, be.ptr(s["rew"]), be.ptr(s["done"]), None, be.stream)
    A, B = be.host(st[0]["state"]), be.host(st[1]["state"])
    neq = A != B
    print(f"step {t}: equal={not neq.any()} rows={np.unique(np.argwhere(neq)[:, 0]).tolist()} cols={np.unique(np.argwhere(neq)[:, 1]).tolist()[:24]} max|d|={np.abs(A - B).max():.3e}")
    # forward probe from the shared input state of the NEXT step
