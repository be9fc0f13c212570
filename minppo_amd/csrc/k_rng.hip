// k_rng.hip — JAX-stream-compatible random numbers on the device: threefry2x32 keys, `jax.random.split`,
// `jax.random.normal` and the sort keys of `jax.random.permutation`, in the conventions of jax 0.4.3x
// (threefry_partitionable = False): reference minppo/train.py:158 (`pi.sample(seed=action_rng)`), :163, :252, :258.
// The host restatement these kernels are tested against, and the status of the claim, are in minppo_amd/jaxrng.py.
//   threefry_normal_kernel : jax.random.normal(key, (n,)): bits = threefry_2x32(key, iota(n)) with the counter array split in
//                            halves (thread i produces elements i and i + ceil(n/2)), uniform in (-1, 1) by the mantissa trick,
//                            sqrt(2) erf_inv(u) with XLA's float32 polynomial (Giles)
//   threefry_bits_kernel   : jax.random.bits(key, (n,)) + the identity permutation (first sort round)
//   threefry_chain_kernel  : the split tree of one `_update_step`, one thread: T x {action key, step key}, E x {epoch key ->
//                            one sub-key per sort round}; the carried key lives in the arena, so a captured graph draws fresh
//                            keys every replay
#include <wave_ops.h>

#include "mppo_common.h"
#include "ppo_layout.h"

namespace mppo {

struct U2 { unsigned x, y; };

__host__ __device__ inline unsigned rotl32(unsigned v, int r) { return (v << r) | (v >> (32 - r)); }

__host__ __device__ inline U2 threefry2x32(unsigned k0, unsigned k1, unsigned x0, unsigned x1) {
  const unsigned k2 = k0 ^ k1 ^ 0x1BD11BDAu;
  // (written out: indexed key / rotation tables end up in scratch memory on the device)
#define TF_MIX(R) { x0 += x1; x1 = rotl32(x1, R) ^ x0; }
#define TF_ROUNDS_A TF_MIX(13) TF_MIX(15) TF_MIX(26) TF_MIX(6)
#define TF_ROUNDS_B TF_MIX(17) TF_MIX(29) TF_MIX(16) TF_MIX(24)
  x0 += k0; x1 += k1;
  TF_ROUNDS_A x0 += k1; x1 += k2 + 1u;
  TF_ROUNDS_B x0 += k2; x1 += k0 + 2u;
  TF_ROUNDS_A x0 += k0; x1 += k1 + 3u;
  TF_ROUNDS_B x0 += k1; x1 += k2 + 4u;
  TF_ROUNDS_A x0 += k2; x1 += k0 + 5u;
#undef TF_ROUNDS_A
#undef TF_ROUNDS_B
#undef TF_MIX
  return {x0, x1};
}

// jax.random.split(key)[which]: threefry_2x32(key, [0, 1, 2, 3]) -> halves (0, 1 | 2, 3) -> out = [y0(0,2), y0(1,3), y1(0,2), y1(1,3)]
__host__ __device__ inline U2 split_key(U2 key, int which) {
  const U2 a = threefry2x32(key.x, key.y, 0u, 2u), b = threefry2x32(key.x, key.y, 1u, 3u);
  return which == 0 ? U2{a.x, b.x} : U2{a.y, b.y};
}

__device__ __forceinline__ float normal_from_bits(unsigned bits) {
  const float lo = -0.99999994f;  // nextafter(-1, 0)
  const float f = __uint_as_float((bits >> 9) | 0x3F800000u) - 1.0f;
  const float u = fmaxf(lo, f * (1.0f - lo) + lo);
  // XLA ErfInv32
  float w = -log1pf(-(u * u));
  const bool lt = w < 5.0f;
  w = lt ? w - 2.5f : sqrtf(w) - 3.0f;
  float p = lt ? 2.81022636e-08f : -0.000200214257f;
  p = (lt ? 3.43273939e-07f : 0.000100950558f) + p * w;
  p = (lt ? -3.5233877e-06f : 0.00134934322f) + p * w;
  p = (lt ? -4.39150654e-06f : -0.00367342844f) + p * w;
  p = (lt ? 0.00021858087f : 0.00573950773f) + p * w;
  p = (lt ? -0.00125372503f : -0.0076224613f) + p * w;
  p = (lt ? -0.00417768164f : 0.00943887047f) + p * w;
  p = (lt ? 0.246640727f : 1.00167406f) + p * w;
  p = (lt ? 1.50140941f : 2.83297682f) + p * w;
  return 1.41421354f * (p * u);
}

__global__ void __launch_bounds__(256) threefry_normal_kernel(const unsigned* __restrict__ key2, size_t n, float* __restrict__ out) {
  const size_t half = (n + 1) / 2, i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= half) return;
  const size_t j = i + half;
  const U2 y = threefry2x32(key2[0], key2[1], (unsigned)i, j < n ? (unsigned)j : 0u);  // the padding counter of an odd n is 0
  out[i] = normal_from_bits(y.x);
  if (j < n) out[j] = normal_from_bits(y.y);
}

__global__ void __launch_bounds__(256) threefry_bits_kernel(const unsigned* __restrict__ key2, size_t n, unsigned* __restrict__ out, int* __restrict__ iota) {
  const size_t half = (n + 1) / 2, i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= half) return;
  const size_t j = i + half;
  const U2 y = threefry2x32(key2[0], key2[1], (unsigned)i, j < n ? (unsigned)j : 0u);
  out[i] = y.x;
  if (iota) iota[i] = (int)i;
  if (j < n) { out[j] = y.y; if (iota) iota[j] = (int)j; }
}

__global__ void threefry_chain_kernel(unsigned* __restrict__ rng2, int T, int E, int rounds, unsigned* __restrict__ act_keys, unsigned* __restrict__ sort_keys) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  U2 rng{rng2[0], rng2[1]};
  for (int t = 0; t < T; ++t) {
    const U2 act = split_key(rng, 1);  // rng, action_rng = split(rng)   (train.py:158)
    rng = split_key(rng, 0);
    act_keys[2 * t] = act.x; act_keys[2 * t + 1] = act.y;
    rng = split_key(rng, 0);           // rng, step_rng = split(rng)     (train.py:163; the step keys are unused: deterministic env)
  }
  for (int e = 0; e < E; ++e) {
    U2 k = split_key(rng, 1);          // rng, _rng = split(rng)         (train.py:252)
    rng = split_key(rng, 0);
    for (int r = 0; r < rounds; ++r) {  // _shuffle: key, subkey = split(key) per sort round
      const U2 sub = split_key(k, 1);
      k = split_key(k, 0);
      sort_keys[2 * (e * rounds + r)] = sub.x; sort_keys[2 * (e * rounds + r) + 1] = sub.y;
    }
  }
  rng2[0] = rng.x; rng2[1] = rng.y;
}

int32_t threefry_normal(const unsigned* key2, size_t n, float* out, hipStream_t s) {
  hipLaunchKernelGGL(threefry_normal_kernel, dim3(cdiv((long)((n + 1) / 2), 256)), dim3(256), 0, s, key2, n, out);
  MPPO_CHECK_LAUNCH("threefry_normal_kernel");
  return MPPO_OK;
}
int32_t threefry_bits(const unsigned* key2, size_t n, unsigned* out, int* iota, hipStream_t s) {
  hipLaunchKernelGGL(threefry_bits_kernel, dim3(cdiv((long)((n + 1) / 2), 256)), dim3(256), 0, s, key2, n, out, iota);
  MPPO_CHECK_LAUNCH("threefry_bits_kernel");
  return MPPO_OK;
}
int32_t threefry_chain(unsigned* rng2, int T, int E, int rounds, unsigned* act_keys, unsigned* sort_keys, hipStream_t s) {
  hipLaunchKernelGGL(threefry_chain_kernel, dim3(1), dim3(64), 0, s, rng2, T, E, rounds, act_keys, sort_keys);
  MPPO_CHECK_LAUNCH("threefry_chain_kernel");
  return MPPO_OK;
}

}  // namespace mppo

using namespace mppo;

extern "C" int32_t mppo_threefry_normal(const uint32_t* key2, size_t n, float* out, void* stream) {
  MPPO_REQUIRE(key2 && out && n >= 1, "mppo_threefry_normal: bad argument");
  return threefry_normal(key2, n, out, static_cast<hipStream_t>(stream));
}

extern "C" int32_t mppo_threefry_bits(const uint32_t* key2, size_t n, uint32_t* out, void* stream) {
  MPPO_REQUIRE(key2 && out && n >= 1, "mppo_threefry_bits: bad argument");
  return threefry_bits(key2, n, out, nullptr, static_cast<hipStream_t>(stream));
}

extern "C" int32_t mppo_threefry_update_keys(uint32_t* rng2, int32_t T, int32_t E, int32_t rounds, uint32_t* act_keys, uint32_t* sort_keys, void* stream) {
  MPPO_REQUIRE(rng2 && act_keys && sort_keys && T >= 1 && E >= 1 && rounds >= 1, "mppo_threefry_update_keys: bad argument");
  return threefry_chain(rng2, T, E, rounds, act_keys, sort_keys, static_cast<hipStream_t>(stream));
}
