"""Two ways of running the C ABI under test.

  hip : minppo_amd/libminppo_hip.so on a real MI355X, torch CUDA tensors as device memory (-m gpu)
  emu : tests/emu/libminppo_emu.so — the SAME kernel sources compiled by g++ against the SIMT
        emulator shim, NumPy arrays as "device" memory.  Test infrastructure only: it lets the
        parity tests exercise the kernels' logic and the engine's host code on the CPU-only box.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

from minppo_amd import _native as nat

ROOT = Path(__file__).resolve().parent.parent
EMU_DIR = ROOT / "tests" / "emu"
EMU_LIB = EMU_DIR / "libminppo_emu.so"


def _emu_stale() -> bool:
    if not EMU_LIB.exists():
        return True
    t = EMU_LIB.stat().st_mtime
    srcs = list((ROOT / "minppo_amd" / "csrc").glob("*.h*")) + list(EMU_DIR.glob("*.h")) + list(EMU_DIR.glob("*.cpp")) + list(
        (EMU_DIR / "hip").glob("*.h")) + [ROOT / "include" / "minppo_hip.h"]
    return any(s.stat().st_mtime > t for s in srcs)


JIT_KINDS = []  # (MPPO_TEST_JIT=1: which kernel each opened robot ended up with - 0 run-time-sized, 1 the library's, 2 compiled at start-up)


class Backend:
    name = "?"

    def model(self, cm, include_c_vals=True):
        """Opens a compiled robot model; returns (handle, dims, keepalive)."""
        blob = np.frombuffer(cm.to_blob(include_c_vals), np.uint8).copy()
        dev = self.arr(blob)
        h = C.c_void_p()
        self.lib.model_open(blob.ctypes.data, blob.size, self.ptr(dev), C.byref(h))
        if os.environ.get("MPPO_TEST_JIT") == "1" and self.name == "hip":
            # every robot of a GPU test run gets the kernel compiled for it at start-up (minppo_amd/jit.py): the parity tests then check THAT kernel
            from minppo_amd import jit

            kind = jit.specialize(self.lib, h, cm)
            JIT_KINDS.append((tuple(jit.dims_of(cm)), kind))
        dims = nat.ModelDims()
        self.lib.model_get_dims(h, C.byref(dims))
        return h, dims, (blob, dev)


class EmuBackend(Backend):
    name = "emu"
    xp = "numpy"
    stream = None

    def __init__(self):
        if _emu_stale():
            r = subprocess.run(["bash", str(EMU_DIR / "build_emu.sh")], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("emulator build failed:\n" + r.stderr[-4000:])
        self.lib = nat.Lib(EMU_LIB)

    def arr(self, x):
        x = np.ascontiguousarray(x)
        raw = np.zeros(x.nbytes + 256, np.uint8)
        o = (-raw.ctypes.data) % 256
        out = raw[o:o + x.nbytes].view(x.dtype).reshape(x.shape)
        out[...] = x
        return out

    def zeros(self, shape, dtype=np.float32):
        return self.arr(np.zeros(shape, dtype))

    def full(self, shape, val, dtype=np.float32):
        return self.arr(np.full(shape, val, dtype))

    def host(self, a):
        return np.array(a)

    def ptr(self, a):
        return 0 if a is None else a.ctypes.data

    def sync(self):
        pass

    def trainer(self, cfg, **kw):
        from minppo_amd.train import Trainer

        kw.setdefault("use_graph", False)
        return Trainer(cfg, lib=self.lib, xp="numpy", **kw)

    def put(self, dst, src):
        dst[...] = src


class HipBackend(Backend):
    name = "hip"
    xp = "torch"

    def __init__(self):
        import torch

        self.torch = torch
        assert torch.cuda.is_available()
        self.lib = nat.load()
        self.dev = torch.device("cuda:0")
        self._stream = torch.cuda.Stream(device=self.dev)
        self.stream = self._stream.cuda_stream

    def arr(self, x):
        return self.torch.from_numpy(np.ascontiguousarray(x).copy()).to(self.dev)

    def zeros(self, shape, dtype=np.float32):
        return self.arr(np.zeros(shape, dtype))

    def full(self, shape, val, dtype=np.float32):
        return self.arr(np.full(shape, val, dtype))

    def host(self, a):
        self.sync()
        return a.detach().cpu().numpy()

    def ptr(self, a):
        return 0 if a is None else a.data_ptr()

    def sync(self):
        self._stream.synchronize()
        self.torch.cuda.synchronize()

    def trainer(self, cfg, **kw):
        from minppo_amd.train import Trainer

        return Trainer(cfg, device="cuda:0", **kw)

    def put(self, dst, src):
        self.torch.cuda.synchronize()
        dst.copy_(self.torch.from_numpy(np.ascontiguousarray(src)))
        self.torch.cuda.synchronize()


_CACHE = {}


def get_backend(name: str) -> Backend:
    if name not in _CACHE:
        _CACHE[name] = EmuBackend() if name == "emu" else HipBackend()
    return _CACHE[name]
