"""env_kernel time as a function of the solver's iteration budget (where do the 160 us go?)."""
import ctypes as C, sys
from dataclasses import replace
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from minppo_amd import _native as nat
from minppo_amd.model import synth_stompy_pro, compile_model
lib = nat.load()
N = 4096
for iters, ls in ((6, 6), (3, 6), (1, 6), (6, 3), (6, 1), (1, 1)):
    cm = compile_model(replace(synth_stompy_pro(), iterations=iters, ls_iterations=ls))
    blob = np.frombuffer(cm.to_blob(), np.uint8)
    dblob = torch.from_numpy(blob.copy()).cuda()
    h = C.c_void_p()
    lib.model_open(blob.ctypes.data, blob.size, dblob.data_ptr(), C.byref(h))
    dims = nat.ModelDims(); lib.model_get_dims(h, C.byref(dims))
    state = torch.zeros(N, dims.rec_dim, device="cuda"); reset = torch.zeros(dims.rec_dim, device="cuda")
    obs = torch.zeros(N, dims.obs_pad, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    met = nat.EnvMetrics(*[t.data_ptr() for t in (torch.zeros(N, device="cuda"), torch.zeros(N, dtype=torch.int32, device="cuda"), torch.zeros(N, device="cuda"),
                                                   torch.zeros(N, dtype=torch.int32, device="cuda"), torch.zeros(N, dtype=torch.int32, device="cuda"), torch.zeros(N, dtype=torch.uint8, device="cuda"))])
    lib.env_reset(h, N, state.data_ptr(), reset.data_ptr(), obs.data_ptr(), dims.obs_pad, 0, 0, C.byref(met), s)
    act = 0.3 * torch.randn(N, dims.nu, device="cuda"); rew = torch.zeros(N, device="cuda"); done = torch.zeros(N, dtype=torch.uint8, device="cuda")
    rc = nat.RewardCfg(-0.2, 2.0, 2.0, 0.2, 0.5, 0.1, 4.0, 1.0, 1.25)
    def step():
        lib.env_step(h, N, 1, C.byref(rc), state.data_ptr(), reset.data_ptr(), act.data_ptr(), dims.nu, obs.data_ptr(), dims.obs_pad, rew.data_ptr(), done.data_ptr(), C.byref(met), s)
    for _ in range(5): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): step()
    e1.record(); torch.cuda.synchronize()
    print(f"iterations={iters} ls_iterations={ls}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per env_step launch (incl. ~3 us launch)")
    lib.model_close(h)
