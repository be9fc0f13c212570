#!/usr/bin/env python3
"""tools/make_mjx_fixtures.py - the socket for an EXTERNAL pin of the physics oracle ("parity unpinned").

This script cannot run in the build container or on the GPU box: it needs `mujoco` and `mujoco.mjx` (+ jax), which are not
installed there and cannot be (no network).  Run it on any machine that has them (pip install mujoco mujoco-mjx "jax[cpu]"),
from the repository root, and commit the files it writes:

    python tools/make_mjx_fixtures.py [synth_stompy_pro synth_stompy_full synth_pile ...]      ->  tests/golden/mjx_<model>.npz

tests/test_mjx_fixtures.py consumes those files when they exist (and skips otherwise): it feeds the recorded (qpos, qvel, ctrl,
qacc_warmstart) to the float64 oracle and - through `mppo_physics_forward` - to the kernel, and compares every intermediate with
what MuJoCo-MJX computed.  That is how `"parity"` could leave "partial".

What is recorded, per model, for S = 10 stepped states of E = 4 environments (random position-actuator targets, seed 7):
  inputs   qpos [S,E,nq], qvel [S,E,nv], ctrl [S,E,nu], qacc_warmstart [S,E,nv]
  outputs of mjx.forward on them with the reference's solver settings (env.py:95-97: CG, iterations 6, ls_iterations 6):
           qM (dense, mjx.full_m), qfrc_bias, qfrc_passive, qfrc_actuator, qacc_smooth, efc_J, efc_D, efc_aref (rows in MJX's
           order: joint limits, then the pyramidal contact rows; inactive rows are all-zero), qacc, cinert, cvel, subtree_com
  outputs of mjx.step: qpos1, qvel1
The MJCF is the repository's own (minppo_amd.mjcf.to_mjcf of the built-in stand-in robot), so both sides see the same model.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main(names):
    import jax
    import jax.numpy as jp
    import mujoco
    from mujoco import mjx

    from minppo_amd import model as mm
    from minppo_amd.mjcf import to_mjcf

    jax.config.update("jax_enable_x64", True)  # the oracle is float64; MJX then computes in float64 too
    for name in names:
        spec = getattr(mm, name)()
        xml = to_mjcf(spec)
        mj = mujoco.MjModel.from_xml_string(xml)
        mj.opt.solver = mujoco.mjtSolver.mjSOL_CG  # env.py:95-97
        mj.opt.iterations = 6
        mj.opt.ls_iterations = 6
        m = mjx.put_model(mj)
        S, E = 10, 4
        rng = np.random.default_rng(7)
        d0 = mjx.make_data(m)
        d0 = d0.replace(qpos=jp.asarray(mj.qpos0))
        datas = [mjx.forward(m, d0) for _ in range(E)]
        rec = {k: [] for k in ("qpos", "qvel", "ctrl", "qacc_warmstart", "qM", "qfrc_bias", "qfrc_passive", "qfrc_actuator", "qacc_smooth", "efc_J", "efc_D",
                               "efc_aref", "qacc", "cinert", "cvel", "subtree_com", "qpos1", "qvel1")}
        step = jax.jit(mjx.step)
        fwd = jax.jit(mjx.forward)
        for s in range(S):
            row = {k: [] for k in rec}
            for e in range(E):
                d = datas[e].replace(ctrl=jp.asarray(0.6 * rng.standard_normal(mj.nu)))
                f = fwd(m, d)
                for k in ("qpos", "qvel", "ctrl", "qacc_warmstart"):
                    row[k].append(np.asarray(getattr(d, k)))
                row["qM"].append(np.asarray(mjx.full_m(m, f)))
                for k in ("qfrc_bias", "qfrc_passive", "qfrc_actuator", "qacc_smooth", "efc_J", "efc_D", "efc_aref", "qacc", "cinert", "cvel", "subtree_com"):
                    row[k].append(np.asarray(getattr(f, k)))
                n = step(m, d)
                row["qpos1"].append(np.asarray(n.qpos)); row["qvel1"].append(np.asarray(n.qvel))
                datas[e] = n
            for k in rec:
                rec[k].append(np.stack(row[k]))
        out = ROOT / "tests" / "golden" / f"mjx_{name}.npz"
        np.savez_compressed(out, **{k: np.stack(v) for k, v in rec.items()}, mujoco_version=np.array(mujoco.__version__), jax_version=np.array(jax.__version__))
        print("wrote", out, {k: np.stack(v).shape for k, v in rec.items()})


if __name__ == "__main__":
    # the two BASELINE robots, then one model per contact routine (pairs of round geoms, box corners, plane_convex, plane_cylinder) and the
    # scene that has them all plus sphere_convex / capsule_convex at rest (synth_pile)
    main(sys.argv[1:] or ["synth_stompy_pro", "synth_stompy_full", "synth_stompy_pro_sc", "synth_brick", "synth_wedge", "synth_can", "synth_pile"])
