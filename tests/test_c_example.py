"""examples/c_engine.c: the whole `_update_step` loop driven from plain C through the C ABI (no Python, no torch in the process).
The C program and the Python host (`minppo_amd.train.Trainer`) configure the same engine, so they must report the same loss."""
import json
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.gpu
def test_c_program_runs_the_engine_and_agrees_with_the_python_host(tmp_path):
    from backends import get_backend
    from minppo_amd.config import make_config
    from minppo_amd.model import load_model
    from minppo_amd.train import init_flat_params

    if shutil.which("gcc") is None:
        pytest.skip("gcc not installed")
    cm = load_model("synth_stompy_pro")
    (tmp_path / "model.blob").write_bytes(cm.to_blob())
    init_flat_params(1337, cm.obs_size(), cm.nu, 256).tofile(tmp_path / "params.f32")
    exe = tmp_path / "c_engine"
    lib_dir = ROOT / "minppo_amd"
    r = subprocess.run(["gcc", "-std=c99", "-D_POSIX_C_SOURCE=199309L", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", f"-I{ROOT / 'include'}",
                        str(ROOT / "examples" / "c_engine.c"), f"-L{lib_dir}", "-lminppo_hip", "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib_dir}",
                        "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    N, updates = 1024, 5
    r = subprocess.run([str(exe), str(tmp_path / "model.blob"), str(tmp_path / "params.f32"), str(N), str(updates)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["optimizer_steps"] == (3 + updates) * 128 and out["updates"] == 3 + updates
    assert np.isfinite([out["total_loss"], out["value_loss"], out["env_steps_per_s"]]).all() and out["env_steps_per_s"] > 1e5
    # same engine configuration from the Python host (same seed, same initial parameters): same last-step loss
    be = get_backend("hip")
    cfg = make_config({"kscale_id": "5eb3cb7f23232298", "visualization": {"camera_name": "track"}}, [f"training.num_envs={N}"])
    tr = be.trainer(cfg, use_graph=True)
    tr.reset()
    for _ in range(3 + updates):
        tr.update()
    lo = tr.losses().reshape(-1, 4)[-1]
    tr.close()
    np.testing.assert_allclose([out["total_loss"], out["value_loss"]], lo[:2], rtol=1e-5, atol=1e-6)
