// philox.h — Philox4x32-10, the engine's counter-based generator (action noise, permutation keys: k_ppo.hip, k_perm.hip)
#pragma once
#include <hip/hip_runtime.h>

namespace mppo {

struct U4 { unsigned x, y, z, w; };
__host__ __device__ inline U4 philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return {c0, c1, c2, c3};
}

}  // namespace mppo
