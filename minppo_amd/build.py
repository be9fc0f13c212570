"""Builds minppo_amd/libminppo_hip.so: every HIP source in csrc/ compiled for gfx950 with hipcc.

    python -m minppo_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects go to minppo_amd/csrc/_build/ (git-ignored); the
shared library stays in-tree so that it travels to the GPU box with the repository snapshot.
"""

from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
BUILD = CSRC / "_build"
LIB = HERE / "libminppo_hip.so"
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-I" + str(CSRC), "-I" + str(HERE.parent / "include"),
         "-Wno-unused-result"]


# Per-file flags.  k_physics.hip is compiled WITHOUT floating-point contraction: measured on MI355X (tools/phys_envelope.py, 64
# walking states), fused multiply-adds everywhere move the constraint solver's result 4x further from the float64 oracle
# (median |dqacc| 3.5e-3 of scale against 9e-4; the solver is an unconverged CG that amplifies where a product was rounded)
# while saving only 4 % of the kernel's time; hot inner products use explicit fmaf where that does not cost parity.
FILE_FLAGS = {"k_physics.hip": ["-ffp-contract=off"]}


def sources() -> list[Path]:
    names = (CSRC / "SOURCES.txt").read_text().split() + (CSRC / "SOURCES_DEVICE_ONLY.txt").read_text().split()
    return [CSRC / n for n in names]


def _newer(target: Path, deps: list[Path]) -> bool:
    if not target.exists():
        return False
    t = target.stat().st_mtime
    return all(d.stat().st_mtime <= t for d in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    BUILD.mkdir(exist_ok=True)
    headers = sorted(CSRC.glob("*.h")) + [HERE.parent / "include" / "minppo_hip.h", Path(__file__)]
    srcs = sources()
    objs = [BUILD / (s.stem + ".o") for s in srcs]

    def compile_one(pair):
        src, obj = pair
        if not force and _newer(obj, [src] + headers):
            return None
        cmd = [HIPCC, *FLAGS, *FILE_FLAGS.get(src.name, []), "-c", str(src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src.name}:\n{r.stderr[-4000:]}")
        return src.name

    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        done = [d for d in ex.map(compile_one, zip(srcs, objs)) if d]
    if done or force or not _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", str(LIB), *map(str, objs), "-L/opt/rocm/lib", "-lrccl",
               "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    if verbose:
        print(f"[minppo_amd.build] compiled {done or 'nothing (up to date)'} -> {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
