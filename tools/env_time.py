"""env_kernel time per launch (HIP events, 30 launches after warm-up) for a given build of the library (argv[1], default: the product).
States are walking states: the policy's unit-variance actions drive the robot for 10 steps first."""
import ctypes as C, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from minppo_amd import _native as nat
if len(sys.argv) > 1:
    nat.HIP_LIB_PATH = Path(sys.argv[1]).resolve()
from minppo_amd.model import load_model
lib = nat.load()
print("library:", lib.path)
only = sys.argv[2] if len(sys.argv) > 2 else None  # restrict to one model (needed for single-model experiment builds)
import os
cases = [("synth_stompy_pro", 4096), ("synth_stompy_full", 8192)]
if os.environ.get("MPPO_ENV_TIME_CASES"):  # e.g. "synth_stompy_pro:8192,synth_stompy_pro:16384"
    cases = [(c.split(":")[0], int(c.split(":")[1])) for c in os.environ["MPPO_ENV_TIME_CASES"].split(",")]
for model, N in cases:
    if only and model != only:
        continue
    if model.startswith("random"):  # random<seed>: a robot of tests/test_model_fuzz.py (e.g. random17: 130 contact slots, one environment per wave)
        sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
        from test_model_fuzz import random_model
        from minppo_amd.model import compile_model
        cm = compile_model(random_model(int(model[6:])))
    elif model == "forty_dof":  # the 40-dof / 34-actuator robot of tests/test_kernels_physics.py (three matrix rows per lane when it gets a kernel of its own)
        sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
        from test_kernels_physics import _many_dof_robot
        cm = _many_dof_robot()
    else:
        cm = load_model(model)
    blob = np.frombuffer(cm.to_blob(), np.uint8)
    dblob = torch.from_numpy(blob.copy()).cuda()
    h = C.c_void_p()
    lib.model_open(blob.ctypes.data, blob.size, dblob.data_ptr(), C.byref(h))
    kind = C.c_int32(0); lib.model_is_specialized(h, C.byref(kind)); kind = kind.value
    if os.environ.get("MPPO_ENV_TIME_JIT") == "1":  # a kernel compiled for this robot now (minppo_amd/jit.py), where the library has none
        from minppo_amd import jit
        kind = jit.specialize(lib, h, cm, verbose=True)
    dims = nat.ModelDims(); lib.model_get_dims(h, C.byref(dims))
    state = torch.zeros(N, dims.rec_dim, device="cuda"); reset = torch.zeros(dims.rec_dim, device="cuda")
    obs = torch.zeros(N, dims.obs_pad, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    met_keep = (torch.zeros(N, device="cuda"), torch.zeros(N, dtype=torch.int32, device="cuda"), torch.zeros(N, device="cuda"),
                torch.zeros(N, dtype=torch.int32, device="cuda"), torch.zeros(N, dtype=torch.int32, device="cuda"), torch.zeros(N, dtype=torch.uint8, device="cuda"))
    met = nat.EnvMetrics(*[t.data_ptr() for t in met_keep])  # (the tensors must outlive the launches: the kernel writes the episode metrics through these pointers)
    lib.env_reset(h, N, state.data_ptr(), reset.data_ptr(), obs.data_ptr(), dims.obs_pad, 0, 0, C.byref(met), s)
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    acts = [torch.randn(N, max(dims.nu, 1), device="cuda", generator=g) for _ in range(10)]  # (a robot without actuators still passes a pointer)
    rew = torch.zeros(N, device="cuda"); done = torch.zeros(N, dtype=torch.uint8, device="cuda")
    rc = nat.RewardCfg(-0.2, 2.0, 2.0, 0.2, 0.5, 0.1, 4.0, 1.0, 1.25)
    def step(k):
        lib.env_step(h, N, 1, C.byref(rc), state.data_ptr(), reset.data_ptr(), acts[k % 10].data_ptr(), max(dims.nu, 1), obs.data_ptr(), dims.obs_pad, rew.data_ptr(), done.data_ptr(), C.byref(met), s)
    for k in range(10): step(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(30): step(k)
    e1.record(); torch.cuda.synchronize()
    nb = C.c_size_t(0); lib.model_scratch_bytes(h, N, C.byref(nb))
    print(f"{model} [{('run-time-sized kernel', 'kernel of the library for this robot', 'kernel compiled at start-up')[kind]}] (nv {cm.nv}, {cm.ncon} contact slots, {cm.nefc} constraint rows, {dims.lds_bytes} bytes of LDS per workgroup = {160 * 1024 // dims.lds_bytes} workgroup(s) per CU, {nb.value >> 10} KB of matrices in global memory) N={N}: {e0.elapsed_time(e1) / 30 * 1e3:.1f} us per env_step launch; state checksum {float(state.double().sum()):.6f} done {int(done.sum())}")
    lib.model_close(h)
