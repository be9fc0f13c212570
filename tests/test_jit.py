"""The environment kernel compiled for one robot at start-up (minppo_amd/jit.py, mppo_model_attach_kernel): the counterpart of the reference's
jax.jit of its step function.  On the CPU: the code object is built and read (hipcc cross-compiles without a GPU), and the library's checks of
what it is handed; on the GPU: the attached kernel against the run-time-sized one, bit for bit, stand-alone and inside the engine."""
import ctypes as C
import os

import numpy as np
import pytest

from minppo_amd import _native as nat
from minppo_amd import jit
from minppo_amd.model import compile_model, load_model

from backends import get_backend
from test_model_fuzz import random_model

ROBOT = "synth_can"  # (a small robot the library has no instantiation for: compiles in seconds)


def _open(be, cm):
    return be.model(cm)


@pytest.fixture(scope="module")
def can_object(tmp_path_factory):
    os.environ[jit.CACHE_ENV] = str(tmp_path_factory.mktemp("jit_cache"))
    cm = load_model(ROBOT)
    path = jit.compile_kernel(jit.dims_of(cm), 48)
    yield cm, path.read_bytes()
    del os.environ[jit.CACHE_ENV]


def test_code_object_holds_the_three_kernels_of_the_robot(can_object):
    cm, image = can_object
    names = jit.kernel_symbols(image)
    dims = "".join(f"Li{d}E" for d in jit.dims_of(cm))
    for mode, n in enumerate(names):
        assert "env_kernel" in n and f"StaticModelI{dims}EELi{mode}EEEv" in n
    # cached: a second request is the same file, not a second compile
    assert jit.compile_kernel(jit.dims_of(cm), 48).read_bytes() == image
    # a robot whose kernel keeps a lambda as a function of its own (40 dofs, three rows per lane: the solve is not inlined): the function's symbol
    # carries the kernel's name inside its own - it is not one of the three
    from test_kernels_physics import _many_dof_robot

    big = jit.compile_kernel(jit.dims_of(_many_dof_robot()), 48).read_bytes()
    assert all(n.startswith("_ZN4mppo10env_kernelI") for n in jit.kernel_symbols(big))
    with pytest.raises(ValueError):
        jit.kernel_symbols(b"\x7fELF" + bytes(200))
    with pytest.raises(ValueError):
        jit.kernel_symbols(b"not a code object")


def test_kernels_can_be_compiled_ahead_of_time(can_object, capsys):
    """`python -m minppo_amd.jit <robot>`: the cache entry a later `jit.specialize` finds (here: the fixture's own, found again)."""
    cm, image = can_object
    assert jit.main([ROBOT]) == 0
    out = capsys.readouterr().out
    path = out.strip().split(" -> ")[-1]
    assert open(path, "rb").read() == image
    assert jit.main([]) == 2


def test_library_refuses_what_is_not_this_robots_kernel(can_object):
    cm, image = can_object
    be = get_backend("emu")
    h, dims, keep = _open(be, cm)
    names = jit.kernel_symbols(image)
    used = C.c_int32(7)
    arr = (C.c_char_p * 3)(*[n.encode() for n in names])
    buf = C.create_string_buffer(image, len(image))
    # another robot's kernels
    other = [n.replace(f"Li{cm.nv}E", f"Li{cm.nv + 1}E", 1).encode() for n in names]
    with pytest.raises(nat.NativeError, match="not the environment kernel of this robot"):
        be.lib.model_attach_kernel(h, C.cast(buf, C.c_void_p), len(image), (C.c_char_p * 3)(*other), 48, C.byref(used))
    # the modes in the wrong order
    with pytest.raises(nat.NativeError, match="not the environment kernel of this robot"):
        be.lib.model_attach_kernel(h, C.cast(buf, C.c_void_p), len(image), (C.c_char_p * 3)(arr[1], arr[0], arr[2]), 48, C.byref(used))
    with pytest.raises(nat.NativeError, match="null argument"):
        be.lib.model_attach_kernel(h, None, len(image), arr, 48, C.byref(used))
    # the emulator runs no device code: the object does not load, the model stays as it was
    with pytest.raises(nat.NativeError, match="does not load"):
        be.lib.model_attach_kernel(h, C.cast(buf, C.c_void_p), len(image), arr, 48, C.byref(used))
    assert used.value == 0
    kind = C.c_int32(9)
    be.lib.model_is_specialized(h, C.byref(kind))
    assert kind.value == 0
    be.lib.model_close(h)
    # a robot the library was built for keeps the library's kernel
    cm2 = load_model("synth_stompy_pro")
    h2, _, keep2 = _open(be, cm2)
    assert jit.specialize(be.lib, h2, cm2) == 1
    be.lib.model_close(h2)


def _run(lib, h, dims, N, steps, torch, seed=0):
    state = torch.zeros(N, dims.rec_dim, device="cuda")
    reset = torch.zeros(dims.rec_dim, device="cuda")
    obs = torch.zeros(N, dims.obs_pad, device="cuda")
    rew = torch.zeros(N, device="cuda")
    done = torch.zeros(N, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    lib.env_reset(h, N, state.data_ptr(), reset.data_ptr(), obs.data_ptr(), dims.obs_pad, 0, 0, None, s)
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    rc = nat.RewardCfg(-0.2, 2.0, 2.0, 0.2, 0.5, 0.1, 4.0, 1.0, 1.25)
    out = []
    nu = max(dims.nu, 1)
    for _ in range(steps):
        act = torch.randn(N, nu, device="cuda", generator=g)
        lib.env_step(h, N, 1, C.byref(rc), state.data_ptr(), reset.data_ptr(), act.data_ptr(), nu, obs.data_ptr(), dims.obs_pad, rew.data_ptr(), done.data_ptr(), None, s)
        torch.cuda.synchronize()
        out.append((state.cpu().numpy().copy(), obs.cpu().numpy().copy(), rew.cpu().numpy().copy(), done.cpu().numpy().copy()))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("robot", [ROBOT, "synth_stompy_frames", "random4"])
def test_attached_kernel_equals_the_runtime_sized_kernel(robot, tmp_path, monkeypatch):
    import torch

    monkeypatch.setenv(jit.CACHE_ENV, str(tmp_path))
    lib = nat.load()
    cm = compile_model(random_model(int(robot[6:]))) if robot.startswith("random") else load_model(robot)
    blob = np.frombuffer(cm.to_blob(), np.uint8)
    dblob = torch.from_numpy(blob.copy()).cuda()
    hs = []
    for attach in (False, True):
        h = C.c_void_p()
        lib.model_open(blob.ctypes.data, blob.size, dblob.data_ptr(), C.byref(h))
        if attach:
            assert jit.specialize(lib, h, cm) == 2, "the robot did not get a kernel of its own"
            assert jit.specialize(lib, h, cm) == 2  # (asking again changes nothing)
        hs.append(h)
    outs = []
    for h in hs:
        dims = nat.ModelDims()
        lib.model_get_dims(h, C.byref(dims))
        outs.append(_run(lib, h, dims, 300, 12, torch))
    for t, (a, b) in enumerate(zip(*outs)):
        for x, y, what in zip(a, b, ("state", "observation", "reward", "done")):
            assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), f"{robot}: {what} differs at step {t}"
    for h in hs:
        lib.model_close(h)


@pytest.mark.gpu
def test_engine_with_a_kernel_compiled_at_start_up(tmp_path, monkeypatch):
    """Two updates of the whole engine on a robot the library has no kernel for: parameters bit-equal with and without `environment.jit_kernel`
    (the engine's hipGraph captures the attached kernel's launches)."""
    import torch

    from minppo_amd.config import load_config_from_cli
    from minppo_amd.train import Trainer

    monkeypatch.setenv(jit.CACHE_ENV, str(tmp_path))
    res = []
    for flag in ("false", "true"):
        cfg = load_config_from_cli(["stompy_pro", "environment.model=synth_stompy_frames", "training.num_envs=256", "training.num_minibatches=4", "training.update_epochs=2",
                                    f"environment.jit_kernel={flag}", "training.total_timesteps=7680"])
        tr = Trainer(cfg)
        kind = C.c_int32(0)
        tr.lib.model_is_specialized(tr._model, C.byref(kind))
        assert kind.value == (2 if flag == "true" else 0)
        tr.reset()
        for _ in range(3):
            tr.update()
        torch.cuda.synchronize()
        res.append(tr.params_flat())
        tr.close()
    assert res[0].size > 0 and np.isfinite(res[0]).all()
    assert np.array_equal(res[0].view(np.uint8), res[1].view(np.uint8))


@pytest.mark.gpu
def test_environment_wrapper_with_a_kernel_compiled_at_start_up(tmp_path, monkeypatch):
    """`HumanoidEnv` (the reference's env.py surface) with `environment.jit_kernel`: the robot gets its own kernel when the wrapper opens the model,
    and reset / step return the same bits as the wrapper on the run-time-sized kernel."""
    import torch

    from minppo_amd.config import load_config_from_cli
    from minppo_amd.env import HumanoidEnv

    monkeypatch.setenv(jit.CACHE_ENV, str(tmp_path))
    outs = []
    for flag in ("false", "true"):
        cfg = load_config_from_cli(["stompy_pro", f"environment.model={ROBOT}", f"environment.jit_kernel={flag}"])
        env = HumanoidEnv(cfg)
        kind = C.c_int32(-1)
        env.lib.model_is_specialized(env._model, C.byref(kind))
        assert kind.value == (2 if flag == "true" else 0)
        es = env.reset(num_envs=37)
        rng = np.random.default_rng(0)
        rec = [es.obs.cpu().numpy().copy()]
        for _ in range(5):
            a = torch.from_numpy((0.5 * rng.standard_normal((37, env.action_size))).astype(np.float32)).cuda()
            es = env.step(es, a)
            rec += [es.obs.cpu().numpy().copy(), es.reward.cpu().numpy().copy(), es.done.cpu().numpy().copy()]
        outs.append(rec)
    for x, y in zip(*outs):
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8))
