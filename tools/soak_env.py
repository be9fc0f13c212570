"""tools/soak_env.py <model[:N[:steps]]> ... - a long free run of the environment kernel under random actions: are the states finite and bounded at the end,
how many episodes ended (height band as in the reference's config, NaN guard), did any environment leave the scene?  For the colliders of round 5
(synth_pile: hull pairs + cylinder + plane_convex at rest; synth_can; the export biped with its foot-mesh / shin pairs)."""
import ctypes as C, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from minppo_amd import _native as nat
from minppo_amd.model import load_model
lib = nat.load()
for case in sys.argv[1:] or ["synth_pile:1024:20000", "synth_can:1024:20000", "tests/golden/export_biped/robot.xml:1024:5000"]:
    parts = case.split(":")
    model, N, steps = parts[0], int(parts[1]) if len(parts) > 1 else 1024, int(parts[2]) if len(parts) > 2 else 5000
    cm = load_model(model)
    blob = np.frombuffer(cm.to_blob(), np.uint8)
    dblob = torch.from_numpy(blob.copy()).cuda()
    h = C.c_void_p()
    lib.model_open(blob.ctypes.data, blob.size, dblob.data_ptr(), C.byref(h))
    dims = nat.ModelDims(); lib.model_get_dims(h, C.byref(dims))
    state = torch.zeros(N, dims.rec_dim, device="cuda"); reset = torch.zeros(dims.rec_dim, device="cuda"); obs = torch.zeros(N, dims.obs_pad, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    keep = (torch.zeros(N, device="cuda"), torch.zeros(N, dtype=torch.int32, device="cuda"), torch.zeros(N, device="cuda"), torch.zeros(N, dtype=torch.int32, device="cuda"),
            torch.zeros(N, dtype=torch.int32, device="cuda"), torch.zeros(N, dtype=torch.uint8, device="cuda"))
    met = nat.EnvMetrics(*[t.data_ptr() for t in keep])
    lib.env_reset(h, N, state.data_ptr(), reset.data_ptr(), obs.data_ptr(), dims.obs_pad, 0, 0, C.byref(met), s)
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    nu = max(dims.nu, 1)
    rew = torch.zeros(N, device="cuda"); done = torch.zeros(N, dtype=torch.uint8, device="cuda")
    rc = nat.RewardCfg(-0.2, 2.0, 2.0, 0.2, 0.5, 0.1, 4.0, 1.0, 1.25)
    ndone = 0
    zmin, zmax, vmax = 1e9, -1e9, 0.0
    for k in range(steps):
        act = 0.7 * torch.randn(N, nu, device="cuda", generator=g)
        lib.env_step(h, N, 1, C.byref(rc), state.data_ptr(), reset.data_ptr(), act.data_ptr(), nu, obs.data_ptr(), dims.obs_pad, rew.data_ptr(), done.data_ptr(), C.byref(met), s)
        if k % 500 == 499 or k == steps - 1:
            torch.cuda.synchronize()
            st = state[:, :cm.nq + cm.nv]
            assert torch.isfinite(st).all(), (model, k)
            zmin, zmax = min(zmin, float(st[:, 2].min())), max(zmax, float(st[:, 2].max()))
            vmax = max(vmax, float(st[:, cm.nq:].abs().max()))
        ndone += int(done.sum()) if k % 50 == 0 else 0
    print(f"{model}: {N} environments x {steps} steps: finite; root height {zmin:.3f} .. {zmax:.3f}, max |qvel| at the checkpoints {vmax:.2f}, episodes seen ending at every 50th step {ndone}, "
          f"mean reward at the end {float(rew.mean()):.3f}")
    lib.model_close(h)
