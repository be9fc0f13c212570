"""The environment kernel compiled for ONE robot at start-up: what `MPPO_SPECIALIZE=robot.xml python -m minppo_amd.build` does at build
time, for a robot the library was not built for.

What it stands in for: the reference `jax.jit`s its environment step (env.py:123,147 `@partial(jax.jit, static_argnums=(0,))` on `reset` /
`step`; train.py:133,138 the vmapped wrappers, :306 `jax.jit(make_train(config))`): XLA compiles them for the loaded robot's shapes when
training starts.  Here the library carries a run-time-sized kernel that runs any robot, and kernels
with fixed dimensions for the BASELINE robots (csrc/spec_dims.inc) - 1.2x - 1.6x faster, bit-identical results.  `specialize()` gives any
other robot its own: `hipcc --cuda-device-only` of csrc/k_physics.hip with a one-robot list (about 20 s, once: the code object is cached
under the hash of the kernel sources, the build flags and the dimensions), handed to the library through `mppo_model_attach_kernel`, which
checks it against the run-time-sized kernel on the device (a reset and four steps of 24 environments, bit for bit) before it is used.

A kernel that fails that check (round 6 saw one - a 34-dof / 93-body robot with its Cholesky factors in registers: a per-lane flag the
compiler spilled inside divergent code, since carried as the wave's ballot) is compiled once more with the factors in LDS
(`-DMPPO_REGCHOL_MAX_NV=32`); the failure is remembered beside the cache entry, so the next start goes to the working variant directly.

    from minppo_amd import jit
    used = jit.specialize(lib, model_handle, compiled_model)      # 0: run-time-sized kernel stays, 1: the library's own, 2: attached

`environment.jit_kernel=true` makes the trainer and the environment wrapper call it.  Needs hipcc on the machine (the image has it).
`python -m minppo_amd.jit robot.xml` compiles ahead of time (no GPU needed: hipcc cross-compiles), into the same cache.
"""

from __future__ import annotations

import ctypes as C
import fcntl
import hashlib
import os
import struct
import subprocess
import sys
import tempfile
import time
from pathlib import Path
from typing import List, Optional, Tuple

from minppo_amd import build as _build

CACHE_ENV = "MPPO_JIT_CACHE"
DEFAULT_CACHE = _build.HERE / "_jit_cache"  # in-tree (git-ignored), like the built library: it travels with the tree
_DEFAULT_REGCHOL = 48  # csrc/model_view.h: MPPO_REGCHOL_MAX_NV's default


def _regchol_of_build() -> int:
    for f in _build.FILE_FLAGS.get("k_physics.hip", []):
        if f.startswith("-DMPPO_REGCHOL_MAX_NV="):
            return int(f.split("=", 1)[1])
    return _DEFAULT_REGCHOL


def sources_key() -> str:
    """sha-256 over everything a code object of k_physics.hip depends on: the kernel sources and headers, the C header, the build flags."""
    h = hashlib.sha256()
    files = sorted(_build.CSRC.glob("*.hip")) + sorted(_build.CSRC.glob("*.h")) + [_build.HERE.parent / "include" / "minppo_hip.h"]
    for f in files:
        h.update(f.name.encode() + b"\0" + f.read_bytes() + b"\0")
    h.update(" ".join(_build.FLAGS[:1] + [x for x in _build.FLAGS[1:] if not x.startswith("-I")]).encode())
    h.update(" ".join(x for x in _build.FILE_FLAGS.get("k_physics.hip", []) if not x.startswith("-DMPPO_REGCHOL_MAX_NV")).encode())
    return h.hexdigest()[:20]


def dims_of(cm) -> Tuple[int, ...]:
    t = cm.t
    return tuple((1 if int(t["nhull"]) > 0 else 0) if k == "hull" else int(t[k]) for k in _build._SPEC_KEYS)


def kernel_symbols(image: bytes) -> List[str]:
    """The three env_kernel symbols (modes 0, 1, 2) of a gfx950 code object: its ELF symbol table read directly (no binutils needed)."""
    if image[:4] != b"\x7fELF" or image[4] != 2 or image[5] != 1:
        raise ValueError("not a little-endian ELF64 code object")
    shoff, = struct.unpack_from("<Q", image, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", image, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", image, shoff + i * shentsize) for i in range(shnum)]
    names = set()
    for (_n, typ, _f, _a, off, size, link, _i, _al, entsize) in secs:
        if typ not in (2, 11) or entsize == 0:  # SHT_SYMTAB, SHT_DYNSYM
            continue
        stroff = secs[link][4]
        for k in range(size // entsize):
            st_name, st_info = struct.unpack_from("<IB", image, off + k * entsize)
            if (st_info & 0xF) != 2:  # STT_FUNC
                continue
            end = image.index(b"\0", stroff + st_name)
            names.add(image[stroff + st_name:end].decode())
    out = []
    for mode in range(3):
        tail = f"EELi{mode}EEEvNS_9ModelViewE"
        # (the kernels themselves: `_ZN4mppo10env_kernelI...`; a lambda of the kernel that the compiler did not inline is a function `_ZZN4mppo10env_kernelI...`)
        hits = sorted(n for n in names if n.startswith("_ZN4mppo10env_kernelINS_11StaticModelI") and n.endswith(tail + "NS_7EnvArgsENS_7PhysLdsE"))
        if len(hits) != 1:
            raise ValueError(f"code object holds {len(hits)} environment kernels of mode {mode} (expected one)")
        out.append(hits[0])
    return out


def cache_dir() -> Path:
    d = Path(os.environ.get(CACHE_ENV) or DEFAULT_CACHE)
    d.mkdir(parents=True, exist_ok=True)
    return d


def _extra_flags() -> List[str]:
    # MPPO_JIT_EXTRA_FLAGS: more hipcc flags for the start-up compile (code-generation experiments; part of the cache key)
    import shlex

    return shlex.split(os.environ.get("MPPO_JIT_EXTRA_FLAGS", ""))


def _key(dims: Tuple[int, ...], regchol: int) -> str:
    return hashlib.sha256(f"{sources_key()}|{dims}|{regchol}|{' '.join(_extra_flags())}".encode()).hexdigest()[:24]


def compile_kernel(dims: Tuple[int, ...], regchol: int, *, verbose: bool = False) -> Path:
    """The cached code object of the environment kernel for `dims`, compiled if it is not there (one process compiles, the others wait)."""
    key = _key(dims, regchol)
    out = cache_dir() / f"env_{key}.hsaco"
    if out.exists():
        return out
    with open(cache_dir() / f"env_{key}.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if out.exists():
                return out
            t0 = time.time()
            with tempfile.TemporaryDirectory(dir=cache_dir()) as tmp:
                inc = Path(tmp) / "one_robot.inc"
                inc.write_text(f"MPPO_SPEC_EXTRA({', '.join(map(str, dims))})\n")
                obj = Path(tmp) / "k.hsaco"
                flags = [f for f in _build.FILE_FLAGS.get("k_physics.hip", []) if not f.startswith("-DMPPO_REGCHOL_MAX_NV")]
                cmd = [_build.HIPCC, *_build.FLAGS, *flags, f"-DMPPO_REGCHOL_MAX_NV={regchol}", "--cuda-device-only", "--no-gpu-bundle-output", "-DMPPO_JIT_ONLY",
                       f'-DMPPO_SPEC_INC="{inc}"', *_extra_flags(), "-c", str(_build.CSRC / "k_physics.hip"), "-o", str(obj)]
                try:
                    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1800)
                except FileNotFoundError as e:
                    raise RuntimeError(f"environment.jit_kernel needs hipcc ({_build.HIPCC}; set $HIPCC) to compile the environment kernel for this robot") from e
                if r.returncode != 0:
                    raise RuntimeError(f"hipcc failed on the environment kernel for {dims}:\n{r.stderr[-4000:]}")
                kernel_symbols(obj.read_bytes())  # (a code object without the three kernels is not cached)
                os.replace(obj, out)
            if verbose:
                print(f"[minppo_amd.jit] environment kernel for {dims} (factors in registers up to {regchol} dofs) compiled in {time.time() - t0:.0f} s -> {out}", file=sys.stderr)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return out


def attach(lib, handle, image: bytes, regchol: int) -> bool:
    names = kernel_symbols(image)
    arr = (C.c_char_p * 3)(*[n.encode() for n in names])
    used = C.c_int32(0)
    buf = C.create_string_buffer(image, len(image))
    lib.model_attach_kernel(handle, C.cast(buf, C.c_void_p), len(image), arr, regchol, C.byref(used))
    return bool(used.value)


def specialize(lib, handle, cm, *, verbose: bool = False) -> int:
    """Gives the opened model `handle` (of CompiledModel `cm`) a kernel of its own dimensions if the library has none.
    Returns mppo_model_is_specialized afterwards: 0 run-time-sized kernel, 1 an instantiation of the library, 2 an attached one."""
    state = C.c_int32(0)
    lib.model_is_specialized(handle, C.byref(state))
    if state.value:
        return state.value
    if os.environ.get("MPPO_ENV_GENERIC") == "1" or os.environ.get("MPPO_ENV_SPILL"):
        return 0  # (the switches that force the run-time-sized kernel)
    dims = dims_of(cm)
    nv = dims[1]
    first = _regchol_of_build()
    tries = [first] + ([32] if first > 32 and nv > 32 else [])
    for regchol in tries:
        bad = cache_dir() / f"failed_{_key(dims, regchol)}"
        if bad.exists():
            continue  # (this variant failed the device check before)
        path = compile_kernel(dims, regchol, verbose=verbose)
        if attach(lib, handle, path.read_bytes(), regchol):
            return 2
        lib.model_is_specialized(handle, C.byref(state))
        if state.value:
            return state.value
        bad.write_text("mppo_model_attach_kernel did not take this code object (see stderr of the run that wrote this file)\n")
    return 0


def main(argv: Optional[List[str]] = None) -> int:
    """python -m minppo_amd.jit <robot.xml | built-in name> [...]: compiles (and caches) the robots' environment kernels ahead of time - no GPU needed,
    hipcc cross-compiles - so that the first training start does not have to.  Prints the cache entry of each."""
    from minppo_amd.model import load_model

    names = list(sys.argv[1:] if argv is None else argv)
    if not names:
        print(main.__doc__, file=sys.stderr)
        return 2
    for name in names:
        cm = load_model(name)
        dims = dims_of(cm)
        regchol = _regchol_of_build()
        path = compile_kernel(dims, regchol, verbose=True)
        print(f"{name}: dims {dims} -> {path}")
        if regchol > 32 and dims[1] > 32:  # (the variant specialize() falls back to if the device check refuses the first)
            print(f"{name}: fallback variant -> {compile_kernel(dims, 32, verbose=True)}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
