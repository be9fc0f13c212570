"""Solver-envelope probe (GPU): float32 kernel qacc against the float64 oracle on walking / limit / rest states, for a given build
of the library (argv[1], default the product library), twice (determinism).  Test infrastructure: uses oracle/."""
import sys, ctypes as C
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch
from minppo_amd import _native as nat
from minppo_amd.model import load_model
from oracle.physics_oracle import Physics, PhysState
import backends
from test_kernels_physics import _probe, _walk, _limit_state, _rest_state, _cost

libpath = sys.argv[1] if len(sys.argv) > 1 else None
if libpath:  # must be the FIRST engine library in the process: its internal calls and kernel stubs bind to the first global definition
    nat.HIP_LIB_PATH = Path(libpath).resolve()
be = backends.HipBackend()
print("library:", be.lib.path)
f32 = np.float32
for model in ("synth_stompy_pro", "synth_stompy_full"):
    cm = load_model(model); h, dims, _k = be.model(cm); ph = Physics(cm.t)
    for name in ("walk", "limits", "rest"):
        rng = np.random.default_rng(1); N = 64
        if name == "walk":
            _, d, rng = _walk(cm, N, 6, 5); st = (d.qpos, d.qvel, 0.4 * rng.standard_normal((N, cm.nu)), d.qacc_warmstart)
        else:
            st = (_limit_state if name == "limits" else _rest_state)(cm, ph, N, rng)
        q32 = [x.astype(f32) for x in st]
        ref = PhysState(qpos=q32[0].astype(np.float64), qvel=q32[1].astype(np.float64), ctrl=q32[2].astype(np.float64), qacc_warmstart=q32[3].astype(np.float64), time=np.zeros(N))
        ph.forward(ref)
        g1 = _probe(be, h, cm, *q32); g2 = _probe(be, h, cm, *q32)
        same = all(np.array_equal(g1[k], g2[k]) for k in g1)
        rel = np.abs(g1["qacc"] - ref.qacc).max(1) / (np.abs(ref.qacc).max(1) + 1e-9)
        cg, cr = _cost(ref, g1["qacc"]), _cost(ref, ref.qacc)
        crel = (cg - cr) / (np.abs(cr) + 1e-9)
        print(f"{model:18s} {name:6s} deterministic={same}  qacc rel: med {np.median(rel):.2e} q75 {np.quantile(rel,0.75):.2e} max {rel.max():.2e} | cost rel diff: min {crel.min():+.3f} med {np.median(crel):+.2e} max {crel.max():+.3f} | niter {np.bincount(g1['niter'], minlength=7)}")
