"""The model blob's validator (mppo_model_open) against mutated blobs: the environment kernel follows indices from the blob - bodies, dofs,
hull faces, vertex lists, edges, contact kinds - and a kernel that reads out of range can take a GPU down with it, so every index the kernel
follows must be range-checked on the host BEFORE anything is launched.  Single-word mutations of valid blobs (header dimensions, every
integer table, the hull section, the tags that ride in float tables): the library must either refuse the blob or, if it accepts it, step it
on the emulator build without a fault.  Run under AddressSanitizer with `tools/emu_sanitize.sh tests/test_blob_fuzz.py`:
an accepted blob that makes the kernel read out of range shows up there."""
import ctypes as C
import os

import numpy as np
import pytest

from backends import get_backend
from minppo_amd import _native as nat
from minppo_amd.model import _BLOB_F32, _BLOB_INT, compile_model, load_model

f32 = np.float32


def _models():
    from test_convex_pairs import scene

    return [("cvx_scene", compile_model(scene())), ("synth_can", load_model("synth_can")), ("synth_wedge", load_model("synth_wedge")),
            ("synth_stompy_pro_sc", load_model("synth_stompy_pro_sc"))]


def _step_once(be, h, N=4):
    dims = nat.ModelDims()
    be.lib.model_get_dims(h, C.byref(dims))
    OP, R = dims.obs_pad, dims.rec_dim
    state, reset_rec, obs = be.zeros((N, R)), be.zeros((R,)), be.zeros((N, OP))
    rew, done = be.zeros((N,)), be.zeros((N,), np.uint8)
    be.lib.env_reset(h, N, be.ptr(state), be.ptr(reset_rec), be.ptr(obs), OP, be.ptr(rew), be.ptr(done), None, be.stream)
    rc = nat.RewardCfg(-0.2, 2.0, 2.0, 0.2, 0.5, 0.1, 4.0, 1.0, 1.25)
    a = be.zeros((N, max(dims.nu, 1)))
    be.lib.env_step(h, N, 1, C.byref(rc), be.ptr(state), be.ptr(reset_rec), be.ptr(a), max(dims.nu, 1), be.ptr(obs), OP, be.ptr(rew), be.ptr(done), None, be.stream)


@pytest.mark.parametrize("which", [0, 1, 2, 3])
def test_mutated_blobs_are_refused_or_harmless(which):
    be = get_backend("emu")
    name, cm = _models()[which]
    blob = np.frombuffer(cm.to_blob(True), np.uint8).copy()
    words = blob.view(np.int32)
    total, hull_words = int(words[2]), int(words[35])
    rng = np.random.default_rng(100 + which)
    # where the integers the kernel follows live: header dims, the directory, the integer tables, the hull section's header and index tables,
    # and the float tables that carry tags (pair_geom's hull / slot words are floats: mutated by value below)
    nint = len(_BLOB_INT)
    int_ranges = [(3, 16), (32, 37), (64, 64 + 2 * nint)]
    for k in range(nint):
        off, cnt = int(words[64 + 2 * k]), int(words[64 + 2 * k + 1])
        if cnt:
            int_ranges.append((off, off + cnt))
    if hull_words:
        hs = words[total:total + 5]
        nidx = 8 + sum((n + 3) & ~3 for n in (hs[0] + 1, hs[0] + 1, hs[0] + 1, hs[2] + 1, hs[3], 2 * hs[4]))
        int_ranges.append((total, total + nidx))
    candidates = np.concatenate([np.arange(a, b) for a, b in int_ranges])
    values = [-1, -2, -5, 0, 1, 2, 3, 7, 63, 64, 65, 127, 128, 129, 1000, 2 ** 20, 2 ** 31 - 1, -2 ** 31]
    accepted = refused = 0
    for trial in range(int(os.environ.get("MPPO_FUZZ_TRIALS", "160"))):  # (tools/emu_sanitize.sh runs of the round: 2000)
        w = words.copy()
        i = int(rng.choice(candidates))
        w[i] = int(rng.choice(values)) if rng.random() < 0.7 else int(w[i]) + int(rng.choice([-1, 1]))
        if w[i] == words[i]:
            continue
        raw = w.view(np.uint8)
        dev = be.arr(raw)
        h = C.c_void_p()
        try:
            be.lib.model_open(raw.ctypes.data, raw.size, be.ptr(dev), C.byref(h))
        except nat.NativeError:
            refused += 1
            continue
        accepted += 1
        try:
            _step_once(be, h)          # must not fault (under ASan: must not read or write out of range)
        except nat.NativeError:
            pass                       # a launch-time argument check may still refuse it
        be.lib.model_close(h)
    # tags in float tables: pair_geom[7] (hull + 1) and [15] (slot)
    kpg = nint + _BLOB_F32.index("pair_geom")
    pg_off, pg_cnt = int(words[64 + 2 * kpg]), int(words[64 + 2 * kpg + 1])
    for k in range(pg_cnt // 16):
        for col, vals in ((7, (-1.0, 0.5, 3.0, 100.0, 1e9, float("nan"))), (15, (-1.0, 0.5, 2.0, float("nan")))):
            for v in vals:
                w = words.copy()
                w.view(f32)[pg_off + 16 * k + col] = v
                raw = w.view(np.uint8)
                h = C.c_void_p()
                try:
                    be.lib.model_open(raw.ctypes.data, raw.size, be.ptr(be.arr(raw)), C.byref(h))
                except nat.NativeError:
                    refused += 1
                    continue
                accepted += 1
                _step_once(be, h)
                be.lib.model_close(h)
    assert refused >= 40, (name, accepted, refused)   # (most mutations of an index are out of range; the rest are other valid models)
