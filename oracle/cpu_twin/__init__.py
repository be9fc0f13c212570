"""ctypes binding of the C++ / OpenMP float32 twin of the environment step (env_twin.cpp).   TEST / BENCH INFRASTRUCTURE.

PARITY UNPINNED (oracle/physics_oracle.py).  Used by `bench.py`'s `cpu_baseline` leg and by tests/test_cpu_twin.py only; nothing under
minppo_amd/ imports it.  The library is compiled with `-march=native`, so it is built on the machine that runs it: one file per CPU
model (`libenv_twin.<tag>.so`, git-ignored), rebuilt when the source is newer."""

from __future__ import annotations

import ctypes as C
import hashlib
import os
import platform
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent


def _cpu_tag() -> str:
    model = platform.processor() or ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith(("model name", "flags")):
                model += line
                if line.startswith("flags"):
                    break
    except OSError:
        pass
    return hashlib.sha1(model.encode()).hexdigest()[:10]


def lib_path() -> Path:
    return HERE / f"libenv_twin.{_cpu_tag()}.so"


def build(force: bool = False) -> Path:
    out = lib_path()
    src = [HERE / "env_twin.cpp", ROOT / "minppo_amd" / "csrc" / "model_view.h"]
    if not force and out.exists() and all(s.stat().st_mtime <= out.stat().st_mtime for s in src):
        return out
    cmd = ["g++", "-std=c++17", "-O3", "-march=native", "-fopenmp", "-fPIC", "-shared", "-Wall", "-Wno-unused-variable", f"-I{HERE / 'stub'}",
           f"-I{ROOT / 'minppo_amd' / 'csrc'}", str(HERE / "env_twin.cpp"), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building the CPU twin failed:\n" + r.stderr[-3000:])
    return out


class RewardCfg(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("height_min_z", "height_max_z", "exp_coefficient", "subtraction_factor", "max_diff_norm", "w_ctrl_cost", "w_original_pos",
                                         "w_is_healthy", "w_velocity")]


class Metrics(C.Structure):
    _fields_ = [("episode_returns", C.c_void_p), ("episode_lengths", C.c_void_p), ("returned_episode_returns", C.c_void_p), ("returned_episode_lengths", C.c_void_p),
                ("timestep", C.c_void_p), ("returned_episode", C.c_void_p)]


class Twin:
    """One compiled robot model on the CPU twin: `reset(N)`, `step(action)` on NumPy arrays (the engine's state-record layout)."""

    def __init__(self, compiled_model, include_c_vals: bool = True, reward: RewardCfg | None = None, threads: int | None = None):
        if threads is not None:
            os.environ["OMP_NUM_THREADS"] = str(threads)  # (read when the OpenMP runtime starts: set before the first call)
        self.dll = C.CDLL(str(build()))
        d = self.dll
        d.twin_model_open.restype = C.c_void_p
        d.twin_model_open.argtypes = [C.c_void_p, C.c_size_t]
        d.twin_model_close.argtypes = [C.c_void_p]
        d.twin_model_dims.argtypes = [C.c_void_p, C.c_void_p]
        d.twin_threads.restype = C.c_int
        d.twin_env_reset.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        d.twin_env_step.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                    C.c_void_p, C.c_void_p]
        blob = np.frombuffer(compiled_model.to_blob(include_c_vals), np.uint8).copy()
        self.h = d.twin_model_open(blob.ctypes.data, blob.size)
        if not self.h:
            raise ValueError("the CPU twin refused the model blob (magic / version / size)")
        dims = np.zeros(8, np.int32)
        d.twin_model_dims(self.h, dims.ctypes.data)
        self.nq, self.nv, self.nu, self.nbody, self.obs_dim, self.obs_pad, self.rec_dim, self.nefc = (int(x) for x in dims)
        self.rc = reward if reward is not None else RewardCfg(-0.2, 2.0, 2.0, 0.2, 0.5, 0.1, 4.0, 1.0, 1.25)
        self.threads = int(d.twin_threads())
        self.N = 0

    def reset(self, N: int) -> np.ndarray:
        self.N = N
        self.state = np.zeros((N, self.rec_dim), np.float32)
        self.reset_rec = np.zeros(self.rec_dim, np.float32)
        self.obs = np.zeros((N, self.obs_pad), np.float32)
        self.reward = np.zeros(N, np.float32)
        self.done = np.zeros(N, np.uint8)
        self.dll.twin_env_reset(self.h, N, self.state.ctypes.data, self.reset_rec.ctypes.data, self.obs.ctypes.data, self.obs_pad, None)
        return self.obs

    def step(self, action: np.ndarray, n_frames: int = 1):
        a = np.ascontiguousarray(action, np.float32)
        self.dll.twin_env_step(self.h, self.N, n_frames, C.byref(self.rc), self.state.ctypes.data, self.reset_rec.ctypes.data, a.ctypes.data, a.shape[1],
                               self.obs.ctypes.data, self.obs_pad, self.reward.ctypes.data, self.done.ctypes.data, None)
        return self.obs, self.reward, self.done

    def close(self) -> None:
        if self.h:
            self.dll.twin_model_close(self.h)
            self.h = None
