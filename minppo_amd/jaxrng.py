"""JAX-stream-compatible random numbers without JAX.

The reference draws its randomness with `jax.random` on threefry2x32 keys (`minppo/train.py:110,142,158,163-164,252,258,285,303`).
This module restates, in NumPy, exactly the pieces the training loop uses, so that "identical seeds" can mean JAX seeds:

  prng_key(seed)                    jax.random.PRNGKey        -> uint32[2] = (seed >> 32, seed & 0xffffffff)
  split(key, num)                   jax.random.split          (threefry_split: threefry_2x32(key, iota(2 num)).reshape(num, 2))
  random_bits(key, n)               jax.random.bits / _random_bits (32-bit)
  uniform(key, n, lo, hi)           jax.random.uniform        (mantissa trick: bits >> 9 | 0x3f800000, minus 1)
  normal(key, n)                    jax.random.normal         (sqrt(2) erf_inv(uniform in (-1, 1)), XLA's float32 erf_inv polynomial)
  permutation(key, n)               jax.random.permutation    (ceil(3 ln n / ln(2^32 - 1)) rounds of a stable sort by fresh random bits)
  update_keys(rng, T, E)            the split tree of one `_update_step` (train.py:158,163,252)

Conventions are those of jax 0.4.3x with its defaults (`jax_default_prng_impl = threefry2x32`, `jax_threefry_partitionable = False`),
the versions the reference's un-pinned requirements resolved to in October 2024.  STATUS: written from the published algorithm and
checked against the Threefry-2x32-20 known-answer vectors of Random123 and against the few `jax.random` values that are common
knowledge (tests/test_jaxrng.py); JAX itself is not installed here, so bit-equality of long streams with a real JAX run has not
been measured.  The engine's device kernels (`mppo_threefry_*`, csrc/k_rng.hip) are tested bit for bit against this module.
"""

from __future__ import annotations

import math
from typing import Tuple

import numpy as np

_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))
_PARITY = np.uint32(0x1BD11BDA)


def _rotl(x: np.ndarray, r: int) -> np.ndarray:
    return (x << np.uint32(r)) | (x >> np.uint32(32 - r))


def threefry2x32(key: Tuple[int, int], x0: np.ndarray, x1: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Threefry-2x32, 20 rounds (Salmon et al., SC'11), element-wise over the counter arrays."""
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    ks = (k0, k1, np.uint32(k0 ^ k1 ^ _PARITY))
    with np.errstate(over="ignore"):
        x0 = x0.astype(np.uint32) + ks[0]
        x1 = x1.astype(np.uint32) + ks[1]
        for i in range(5):
            for r in _ROT[i % 2]:
                x0 = x0 + x1
                x1 = _rotl(x1, r) ^ x0
            x0 = x0 + ks[(i + 1) % 3]
            x1 = x1 + ks[(i + 2) % 3] + np.uint32(i + 1)
    return x0, x1


def prng_key(seed: int) -> np.ndarray:
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return np.array([seed >> 32, seed & 0xFFFFFFFF], np.uint32)


def _threefry_counts(key, n: int) -> np.ndarray:
    """threefry_2x32(key, iota(n)): the counter array is split into halves (padded to even length), the cipher runs on the
    pairs (first half, second half) and the two output halves are concatenated."""
    half = (n + 1) // 2
    c = np.arange(2 * half, dtype=np.uint32)
    if n % 2:
        c[-1] = 0  # the padding word
    y0, y1 = threefry2x32((key[0], key[1]), c[:half], c[half:])
    return np.concatenate([y0, y1])[:n]


def split(key, num: int = 2) -> np.ndarray:
    return _threefry_counts(key, 2 * num).reshape(num, 2)


def random_bits(key, n: int) -> np.ndarray:
    return _threefry_counts(key, n)


def uniform(key, n: int, minval: float = 0.0, maxval: float = 1.0) -> np.ndarray:
    bits = random_bits(key, n)
    f = ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)
    lo, hi = np.float32(minval), np.float32(maxval)
    return np.maximum(lo, f * (hi - lo) + lo).astype(np.float32)


_ERFINV_LT = np.array([2.81022636e-08, 3.43273939e-07, -3.5233877e-06, -4.39150654e-06, 0.00021858087, -0.00125372503, -0.00417768164, 0.246640727, 1.50140941], np.float32)
_ERFINV_GE = np.array([-0.000200214257, 0.000100950558, 0.00134934322, -0.00367342844, 0.00573950773, -0.0076224613, 0.00943887047, 1.00167406, 2.83297682], np.float32)


def erf_inv_f32(x: np.ndarray) -> np.ndarray:
    """XLA's float32 erf_inv (Giles' polynomial): w = -log1p(-x x); two branches at w = 5."""
    x = x.astype(np.float32)
    w = -np.log1p(-(x * x)).astype(np.float32)
    lt = w < np.float32(5.0)
    w = np.where(lt, w - np.float32(2.5), np.sqrt(w, dtype=np.float32) - np.float32(3.0)).astype(np.float32)
    p = np.where(lt, _ERFINV_LT[0], _ERFINV_GE[0]).astype(np.float32)
    for i in range(1, 9):
        p = (np.where(lt, _ERFINV_LT[i], _ERFINV_GE[i]).astype(np.float32) + p * w).astype(np.float32)
    out = (p * x).astype(np.float32)
    return np.where(np.abs(x) == np.float32(1.0), np.float32(np.inf) * x, out).astype(np.float32)


def normal(key, n: int) -> np.ndarray:
    lo = np.nextafter(np.float32(-1.0), np.float32(0.0))
    u = uniform(key, n, lo, 1.0)
    return (np.float32(math.sqrt(2.0)) * erf_inv_f32(u)).astype(np.float32)


def permutation_rounds(n: int) -> int:
    return int(math.ceil(3 * math.log(max(1, n)) / math.log(np.iinfo(np.uint32).max)))


def permutation(key, n: int) -> np.ndarray:
    x = np.arange(n, dtype=np.int32)
    key = np.asarray(key, np.uint32)
    for _ in range(permutation_rounds(n)):
        key, sub = split(key)
        x = x[np.argsort(random_bits(sub, n), kind="stable")]
    return x


def update_keys(rng, T: int, E: int, n_perm: int):
    """The split tree of one `_update_step` (reference train.py): per env step `rng, action_rng = split(rng)` (:158) and
    `rng, step_rng = split(rng)` (:163; the per-env step keys are unused by the deterministic environment), per epoch
    `rng, _rng = split(rng)` (:252) followed by `_shuffle`'s `key, subkey = split(key)` per sort round.
    Returns (new rng, action keys [T,2], sort keys [E, rounds, 2])."""
    rng = np.asarray(rng, np.uint32)
    rounds = permutation_rounds(n_perm)
    act = np.zeros((T, 2), np.uint32)
    for t in range(T):
        rng, act[t] = split(rng)
        rng, _step = split(rng)
    srt = np.zeros((E, rounds, 2), np.uint32)
    for e in range(E):
        rng, k = split(rng)
        for r in range(rounds):
            k, srt[e, r] = split(k)
    return rng, act, srt
