"""jax.random-compatible streams.  (i) The host restatement (minppo_amd/jaxrng.py) against the Threefry-2x32-20
known-answer vectors of Random123 and the `jax.random` values that are common knowledge (JAX is not installed here: see the
module's STATUS note).  (ii) The device kernels (csrc/k_rng.hip, k_perm.hip) against the host restatement, through the C ABI.
(iii) The engine with training.rng_impl=threefry: the noise and permutations of two consecutive updates are exactly what the
reference's key plumbing (train.py:110,142,158,163,252,258,285) produces from PRNGKey(seed)."""
import ctypes as C

import numpy as np
import pytest

from minppo_amd import jaxrng as jr
from minppo_amd.config import make_config

BASE = {"kscale_id": "5eb3cb7f23232298", "visualization": {"camera_name": "track"}}


def test_threefry_known_answers():
    for key, ctr, want in (((0, 0), (0, 0), (0x6B200159, 0x99BA4EFE)), ((0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF), (0x1CB996FC, 0xBB002BE7)),
                           ((0x13198A2E, 0x03707344), (0x243F6A88, 0x85A308D3), (0xC4923A9C, 0x483DF7A0))):  # Random123 kat_vectors, threefry2x32 20 rounds
        y0, y1 = jr.threefry2x32(key, np.array([ctr[0]], np.uint32), np.array([ctr[1]], np.uint32))
        assert (int(y0[0]), int(y1[0])) == want


def test_jax_random_values_that_are_common_knowledge():
    k0 = jr.prng_key(0)
    assert k0.tolist() == [0, 0] and jr.prng_key(1337).tolist() == [0, 1337]
    assert jr.split(k0).tolist() == [[4146024105, 967050713], [2718843009, 1272950319]]   # jax.random.split(PRNGKey(0))
    assert float(jr.uniform(k0, 1)[0]) == pytest.approx(0.41845703, abs=1e-8)              # jax.random.uniform(PRNGKey(0))
    assert float(jr.normal(k0, 1)[0]) == pytest.approx(-0.20584226, abs=2e-7)              # jax.random.normal(PRNGKey(0))
    np.testing.assert_allclose(jr.normal(jr.prng_key(42), 3), [0.18693547, -1.2806505, -1.5593132], atol=3e-7)  # jax.random.normal(PRNGKey(42), (3,))


def test_stream_properties():
    key = jr.prng_key(7)
    z = jr.normal(key, 200001)  # odd length: the padded counter
    assert abs(float(z.mean())) < 0.01 and abs(float(z.std()) - 1) < 0.01 and np.isfinite(z).all()
    p = jr.permutation(key, 40960)
    assert jr.permutation_rounds(40960) == 2 and (np.sort(p) == np.arange(40960)).all()
    assert not (p == np.arange(40960)).all()
    rng, act, srt = jr.update_keys(key, 10, 4, 40960)
    assert act.shape == (10, 2) and srt.shape == (4, 2, 2) and len({tuple(k) for k in act.tolist()}) == 10


def test_device_kernels_match_the_host_restatement(be):
    key = jr.prng_key(2024)
    dkey = be.arr(key.copy())
    for n in (1, 2, 7, 40960, 40961):
        bits = be.zeros((n,), np.uint32)
        be.lib.threefry_bits(be.ptr(dkey), n, be.ptr(bits), be.stream)
        np.testing.assert_array_equal(be.host(bits), jr.random_bits(key, n))
        out = be.zeros((n,))
        be.lib.threefry_normal(be.ptr(dkey), n, be.ptr(out), be.stream)
        np.testing.assert_allclose(be.host(out), jr.normal(key, n), rtol=2e-5, atol=3e-7)  # log1p / sqrt differ by an ulp between libraries (amplified in the tails)
    T, E, B = 10, 4, 4097
    rounds = jr.permutation_rounds(B)
    rng = be.arr(key.copy())
    act, srt = be.zeros((T, 2), np.uint32), be.zeros((E, rounds, 2), np.uint32)
    be.lib.threefry_update_keys(be.ptr(rng), T, E, rounds, be.ptr(act), be.ptr(srt), be.stream)
    rng_h, act_h, srt_h = jr.update_keys(key, T, E, B)
    np.testing.assert_array_equal(be.host(rng), rng_h)
    np.testing.assert_array_equal(be.host(act), act_h)
    np.testing.assert_array_equal(be.host(srt), srt_h)
    wsb = be.lib.permutation_ws_bytes(B)
    ws, idx = be.zeros((wsb // 4 + 64,)), be.zeros((B,), np.int32)
    be.lib.threefry_permutation(be.ptr(srt), rounds, B, be.ptr(idx), be.ptr(ws), wsb, be.stream)
    # epoch 0's permutation = jax.random.permutation(epoch key, B) with that key's sort sub-keys
    x = np.arange(B, dtype=np.int32)
    for r in range(rounds):
        x = x[np.argsort(jr.random_bits(srt_h[0, r], B), kind="stable")]
    np.testing.assert_array_equal(be.host(idx), x)


def test_engine_draws_the_references_streams(be):
    over = (["training.num_envs=8", "training.num_steps=4", "rl.num_env_steps=4", "training.num_minibatches=2", "training.update_epochs=2",
             "model.hidden_size=32", "training.total_timesteps=100000"] if be.name == "emu" and not __import__("os").environ.get("MPPO_TEST_GPU_SIZES") else
            ["training.num_envs=256", "training.num_minibatches=8", "training.update_epochs=2", "training.total_timesteps=100000000"])
    cfg = make_config(BASE, over + ["training.rng_impl=threefry", "training.seed=1337"])
    tr = be.trainer(cfg, use_graph=(be.name == "hip"))
    tr.reset()
    N, T, A, E = tr.N, tr.T, tr.A, tr.E
    rng = jr.prng_key(1337)
    rng = jr.split(rng)[0]      # network.init           (train.py:110)
    rng = jr.split(rng)[0]      # reset_fn               (train.py:142)
    rng = jr.split(rng)[1]      # RunnerState(..., _rng) (train.py:285)
    for u in range(2):
        tr.update()
        tr._sync()
        rng, act, srt = jr.update_keys(rng, T, E, N * T)
        noise = be.host(tr.region("noise", (T, N, A)))
        for t in range(T):
            np.testing.assert_allclose(noise[t].reshape(-1), jr.normal(act[t], N * A), rtol=2e-5, atol=3e-7)
        perm = be.host(tr.region("perm", (E, N * T)))
        for e in range(E):
            x = np.arange(N * T, dtype=np.int32)
            for r in range(srt.shape[1]):
                x = x[np.argsort(jr.random_bits(srt[e, r], N * T), kind="stable")]
            np.testing.assert_array_equal(perm[e], x)
        np.testing.assert_array_equal(be.host(tr.region("jax_rng"))[:2].view(np.uint32), rng)
    assert np.isfinite(tr.losses()).all()
    tr.close()
