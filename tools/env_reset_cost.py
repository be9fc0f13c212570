"""What terminations cost the environment kernel: env_step time per launch (HIP events, 40 launches) with the healthy band narrowed so that a
given share of the robots terminates and is reset inside the launch (reference env.py:165-181: `select(done, reset_state, next_state)` on
every leaf; csrc/k_physics.hip: the done environments' records are overwritten from the reset record by the same wave).  The reference's
own band (-0.2 .. 2.0) never terminates a robot in 1 B steps (profiles/r03_f_training_run_1B.log), so bench.py's timed region has none."""
import ctypes as C, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from minppo_amd import _native as nat
from minppo_amd.model import load_model
lib = nat.load()
for model, N in (("synth_stompy_pro", 4096), ("synth_stompy_full", 8192)):
    cm = load_model(model)
    blob = np.frombuffer(cm.to_blob(), np.uint8)
    dblob = torch.from_numpy(blob.copy()).cuda()
    h = C.c_void_p()
    lib.model_open(blob.ctypes.data, blob.size, dblob.data_ptr(), C.byref(h))
    dims = nat.ModelDims(); lib.model_get_dims(h, C.byref(dims))
    z0 = float(cm.qpos0[2]) if hasattr(cm, "qpos0") else None
    s = torch.cuda.current_stream().cuda_stream
    for band in ((-0.2, 2.0), (None, 0.004), (None, 0.001), (None, 0.0)):
        state = torch.zeros(N, dims.rec_dim, device="cuda"); reset = torch.zeros(dims.rec_dim, device="cuda")
        obs = torch.zeros(N, dims.obs_pad, device="cuda")
        met_keep = (torch.zeros(N, device="cuda"), torch.zeros(N, dtype=torch.int32, device="cuda"), torch.zeros(N, device="cuda"),
                    torch.zeros(N, dtype=torch.int32, device="cuda"), torch.zeros(N, dtype=torch.int32, device="cuda"), torch.zeros(N, dtype=torch.uint8, device="cuda"))
        met = nat.EnvMetrics(*[t.data_ptr() for t in met_keep])
        lib.env_reset(h, N, state.data_ptr(), reset.data_ptr(), obs.data_ptr(), dims.obs_pad, 0, 0, C.byref(met), s)
        torch.cuda.synchronize()
        zinit = float(obs[0, 2])  # qpos[2] of the standing pose (the observation starts with qpos)
        lo, hi = band if band[0] is not None else (zinit - band[1], zinit + band[1])
        rc = nat.RewardCfg(lo, hi, 2.0, 0.2, 0.5, 0.1, 4.0, 1.0, 1.25)
        g = torch.Generator(device="cuda"); g.manual_seed(0)
        acts = [torch.randn(N, dims.nu, device="cuda", generator=g) for _ in range(10)]
        rew = torch.zeros(N, device="cuda"); done = torch.zeros(N, dtype=torch.uint8, device="cuda")
        def step(k):
            lib.env_step(h, N, 1, C.byref(rc), state.data_ptr(), reset.data_ptr(), acts[k % 10].data_ptr(), dims.nu, obs.data_ptr(), dims.obs_pad, rew.data_ptr(), done.data_ptr(), C.byref(met), s)
        for k in range(10): step(k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dsum = 0.0
        e0.record()
        for k in range(40): step(k)
        e1.record(); torch.cuda.synchronize()
        for k in range(20):
            step(k); dsum += float(done.float().mean())
        print(f"{model} N={N} healthy band z in [{lo:+.4f}, {hi:+.4f}] (standing pose z = {zinit:.4f}): {e0.elapsed_time(e1) / 40 * 1e3:6.1f} us per env_step launch, {100 * dsum / 20:5.1f} % of the robots terminate and are reset per step")
    lib.model_close(h)
