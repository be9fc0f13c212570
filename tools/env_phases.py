"""Where does env_kernel's time go: phase breakdown of one wave (wave 0 of workgroup 0; s_memtime stamps, 100 MHz constant clock)
from a -DMPPO_PHYS_TIMERS build of the library (tools/build_variant.sh timers k_physics.hip -DMPPO_PHYS_TIMERS).
usage: python tools/env_phases.py tools/_variants/libminppo_timers.so [model] [N]"""
import ctypes as C, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from minppo_amd import _native as nat
nat.HIP_LIB_PATH = Path(sys.argv[1]).resolve()
from minppo_amd.model import load_model
lib = nat.load()
model = sys.argv[2] if len(sys.argv) > 2 else "synth_stompy_pro"
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
cm = load_model(model)
blob = np.frombuffer(cm.to_blob(), np.uint8); dblob = torch.from_numpy(blob.copy()).cuda()
h = C.c_void_p(); lib.model_open(blob.ctypes.data, blob.size, dblob.data_ptr(), C.byref(h))
dims = nat.ModelDims(); lib.model_get_dims(h, C.byref(dims))
state = torch.zeros(N, dims.rec_dim, device="cuda"); reset = torch.zeros(dims.rec_dim, device="cuda"); obs = torch.zeros(N, dims.obs_pad, device="cuda")
s = torch.cuda.current_stream().cuda_stream
lib.env_reset(h, N, state.data_ptr(), reset.data_ptr(), obs.data_ptr(), dims.obs_pad, 0, 0, None, s)
g = torch.Generator(device="cuda"); g.manual_seed(0)
rew = torch.zeros(N, device="cuda"); done = torch.zeros(N, dtype=torch.uint8, device="cuda")
rc = nat.RewardCfg(-0.2, 2.0, 2.0, 0.2, 0.5, 0.1, 4.0, 1.0, 1.25)
dll = C.CDLL(str(nat.HIP_LIB_PATH))
names = {0: "load state", 1: "kinematics", 2: "com + contacts", 3: "cinert, cdof", 4: "crb (M)", 5: "cholesky", 6: "tri inverse", 7: "com_vel, cdofdot", 8: "rne/bias/actuation",
         9: "qacc_smooth", 10: "make_constraint", 11: "solver init (3 ctx + grad)", 12: "(loop entry)", 13: "CG it 0", 14: "CG it 1", 15: "CG it 2", 16: "CG it 3", 17: "CG it 4",
         18: "CG it 5", 20: "probe outputs", 21: "euler + integrate", 22: "epilogue"}
acc = {}
for k in range(12):
    act = torch.randn(N, max(dims.nu, 1), device="cuda", generator=g)
    lib.env_step(h, N, 1, C.byref(rc), state.data_ptr(), reset.data_ptr(), act.data_ptr(), max(dims.nu, 1), obs.data_ptr(), dims.obs_pad, rew.data_ptr(), done.data_ptr(), None, s)
    torch.cuda.synchronize()
    t = (C.c_ulonglong * 40)()
    assert dll.mppo_debug_phys_timers(t) == 0
    t = list(t)
    if k < 2:
        continue
    idx = [i for i in sorted(names) if t[i]] + [23]
    for a, b in zip(idx[:-1], idx[1:]):
        acc.setdefault(a, []).append(float(t[b] - t[a]))  # shader-clock ticks
    sub = [13, 24, 25, 26, 27, 28, 29, 30, 14]
    if all(t[i] for i in sub):
        for a, b in zip(sub[:-1], sub[1:]):
            acc.setdefault(100 + a, []).append(float(t[b] - t[a]))
subnames = {113: "  it0: norms, M.s, J.s", 124: "  it0: reductions", 125: "  it0: p0 + first Newton point", 126: "  it0: line-search iterations", 127: "  it0: take step",
            128: "  it0: constraint + gradient update (J^T f)", 129: "  it0: M^-1 grad", 130: "  it0: Polak-Ribiere, new direction"}
names.update(subnames)
tot = 0.0
for i in sorted(acc):
    v = float(np.median(acc[i])); tot += v if i < 100 else 0.0
    print(f"{names[i]:28s} {v:9.0f} ticks {100 * v / sum(float(np.median(x)) for k, x in acc.items() if k < 100):5.1f} %")
print(f"total {tot:9.0f} ticks")
