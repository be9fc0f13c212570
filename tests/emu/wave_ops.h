// TEST INFRASTRUCTURE — CPU stand-in for minppo_amd/csrc/wave_ops.h (same API).
// Cross-lane operations go through a scratch buffer bracketed by workgroup barriers.
#pragma once
#include <hip/hip_runtime.h>

#define MPPO_DYN_SMEM(name) unsigned char* name = ::emu::g_dyn_smem

typedef float f32x16 __attribute__((vector_size(64)));
typedef float f32x4 __attribute__((vector_size(16)));
typedef short bf16x8 __attribute__((vector_size(16)));

inline float emu_bf16_to_f32(short h) { unsigned u = ((unsigned)(unsigned short)h) << 16; float f; memcpy(&f, &u, 4); return f; }

// lane l supplies A[i = l&31][k = l>>5], B[k = l>>5][j = l&31]; acc[r] = D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31]
inline void mfma_f32_32x32x2(float a, float b, f32x16& acc) {
  const int t = emu::tid(), lane = t & 63, wave = t >> 6;
  float* A = reinterpret_cast<float*>(emu::g_xchg) + wave * 128;
  float* B = A + 64;
  A[lane] = a; B[lane] = b;
  __syncthreads();
  const int col = lane & 31, hi = lane >> 5;
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float c = acc[r];
    c = fmaf(A[row], B[col], c);            // k = 0
    c = fmaf(A[32 + row], B[32 + col], c);  // k = 1
    acc[r] = c;
  }
  __syncthreads();
}

// lane l (r = l&31, h = l>>5) supplies A[r][k = 8h+j], B[k = 8h+j][r], j = 0..7
inline void mfma_bf16_32x32x16(bf16x8 a, bf16x8 b, f32x16& acc) {
  const int t = emu::tid(), lane = t & 63, wave = t >> 6;
  float* A = reinterpret_cast<float*>(emu::g_xchg) + wave * 1024;  // [k][row]
  float* B = A + 512;                                             // [k][col]
  const int r = lane & 31, h = lane >> 5;
  for (int j = 0; j < 8; ++j) { A[(8 * h + j) * 32 + r] = emu_bf16_to_f32(a[j]); B[(8 * h + j) * 32 + r] = emu_bf16_to_f32(b[j]); }
  __syncthreads();
  for (int q = 0; q < 16; ++q) {
    const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
    float c = acc[q];
    for (int k = 0; k < 16; ++k) c += A[k * 32 + row] * B[k * 32 + r];
    acc[q] = c;
  }
  __syncthreads();
}

typedef short bf16x4 __attribute__((vector_size(8)));
inline short emu_f32_to_bf16(float f) {  // round to nearest even
  unsigned u; memcpy(&u, &f, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (short)(u >> 16);
}
inline bf16x4 pack_bf16x4(float a, float b, float c, float d) {
  bf16x4 r; r[0] = emu_f32_to_bf16(a); r[1] = emu_f32_to_bf16(b); r[2] = emu_f32_to_bf16(c); r[3] = emu_f32_to_bf16(d);
  return r;
}
// lane l supplies A[i = l&15][k = 4*(l>>4) + c], B[k][j = l&15]; acc[r] = D[4*(l>>4) + r][l&15]
inline bf16x4 bf16x4_from_bits(float lo, float hi) {
  bf16x4 r; const float t[2] = {lo, hi}; __builtin_memcpy(&r, t, 8);
  return r;
}
inline float2 bf16x4_bits(bf16x4 v) {
  float t[2]; __builtin_memcpy(t, &v, 8);
  return make_float2(t[0], t[1]);
}
inline float4 bf16x4_unpack(float lo, float hi) {
  const bf16x4 v = bf16x4_from_bits(lo, hi);
  return make_float4(emu_bf16_to_f32(v[0]), emu_bf16_to_f32(v[1]), emu_bf16_to_f32(v[2]), emu_bf16_to_f32(v[3]));
}
inline void mfma_bf16_16x16x16(bf16x4 a, bf16x4 b, f32x4& acc) {
  const int t = emu::tid(), lane = t & 63, wave = t >> 6;
  float* A = reinterpret_cast<float*>(emu::g_xchg) + wave * 512;  // [k][i]
  float* B = A + 256;                                             // [k][j]
  const int i = lane & 15, q = lane >> 4;
  for (int c = 0; c < 4; ++c) { A[(4 * q + c) * 16 + i] = emu_bf16_to_f32(a[c]); B[(4 * q + c) * 16 + i] = emu_bf16_to_f32(b[c]); }
  __syncthreads();
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * q + r;
    float c = acc[r];
    for (int k = 0; k < 16; ++k) c = fmaf(A[k * 16 + row], B[k * 16 + i], c);
    acc[r] = c;
  }
  __syncthreads();
}
// lane l supplies A[i = l&15][k = 8*(l>>4) + c], B[k][j = l&15], c = 0..7 (raw bits of eight bf16); acc[r] = D[4*(l>>4) + r][l&15]
inline void mfma_bf16_16x16x32(const float4& a_bits, const float4& b_bits, f32x4& acc) {
  const int t = emu::tid(), lane = t & 63, wave = t >> 6;
  float* A = reinterpret_cast<float*>(emu::g_xchg) + wave * 1024;  // [k][i]
  float* B = A + 512;                                              // [k][j]
  const int i = lane & 15, q = lane >> 4;
  short av[8], bv[8];
  __builtin_memcpy(av, &a_bits, 16);
  __builtin_memcpy(bv, &b_bits, 16);
  for (int c = 0; c < 8; ++c) { A[(8 * q + c) * 16 + i] = emu_bf16_to_f32(av[c]); B[(8 * q + c) * 16 + i] = emu_bf16_to_f32(bv[c]); }
  __syncthreads();
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * q + r;
    float c = acc[r];
    for (int k = 0; k < 32; ++k) c = fmaf(A[k * 16 + row], B[k * 16 + i], c);
    acc[r] = c;
  }
  __syncthreads();
}
// lane l supplies A[i = l&31][k = 4*(l>>5) + c], B[k][j = l&31]; acc layout as mfma_f32_32x32x2
inline void mfma_bf16_32x32x8(bf16x4 a, bf16x4 b, f32x16& acc) {
  const int t = emu::tid(), lane = t & 63, wave = t >> 6;
  float* A = reinterpret_cast<float*>(emu::g_xchg) + wave * 512;  // [k][i]
  float* B = A + 256;
  const int i = lane & 31, h = lane >> 5;
  for (int c = 0; c < 4; ++c) { A[(4 * h + c) * 32 + i] = emu_bf16_to_f32(a[c]); B[(4 * h + c) * 32 + i] = emu_bf16_to_f32(b[c]); }
  __syncthreads();
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    float c = acc[r];
    for (int k = 0; k < 8; ++k) c = fmaf(A[k * 32 + row], B[k * 32 + i], c);
    acc[r] = c;
  }
  __syncthreads();
}

inline float group16_sum(float x) {
  float* s = reinterpret_cast<float*>(emu::g_xchg);
  const int t = emu::tid();
  s[t] = x;
  __syncthreads();
  float r = 0.f;
  const int base = t & ~15;
  for (int k = 0; k < 16; ++k) r += s[base + k];
  __syncthreads();
  return r;
}
inline float group16_max(float x) {
  float* s = reinterpret_cast<float*>(emu::g_xchg);
  const int t = emu::tid();
  s[t] = x;
  __syncthreads();
  float r = s[t & ~15];
  for (int k = 1; k < 16; ++k) r = fmaxf(r, s[(t & ~15) + k]);
  __syncthreads();
  return r;
}
inline bool emu_any(bool pred, int width) {
  float* s = reinterpret_cast<float*>(emu::g_xchg);
  const int t = emu::tid();
  s[t] = pred ? 1.f : 0.f;
  __syncthreads();
  bool r = false;
  const int base = t & ~(width - 1);
  const int n = (int)(blockDim.x * blockDim.y * blockDim.z);
  for (int k = 0; k < width && base + k < n; ++k) r = r || (s[base + k] != 0.f);
  __syncthreads();
  return r;
}
inline bool wave_any(bool pred) { return emu_any(pred, 64); }
inline bool group16_any(bool pred) { return emu_any(pred, 16); }
// (csrc/wave_ops.h: the wave's ballot as the carrier of a per-environment flag; here a lane simply keeps its row's verdict)
typedef bool group16_flags_t;
inline group16_flags_t group16_flags(bool pred) { return emu_any(pred, 16); }
inline group16_flags_t group16_flags_or(group16_flags_t a, group16_flags_t b) { return a || b; }
inline bool group16_flag_set(group16_flags_t m) { return m; }
inline float wave_sum(float x) {
  float* s = reinterpret_cast<float*>(emu::g_xchg);
  const int t = emu::tid();
  s[t] = x;
  __syncthreads();
  float r = 0.f;
  const int n = (int)(blockDim.x * blockDim.y * blockDim.z);
  for (int k = 0; k < 64 && (t & ~63) + k < n; ++k) r += s[(t & ~63) + k];
  __syncthreads();
  return r;
}
inline double wave_sum_f64(double x) {
  double* s = emu::g_xchg;
  const int t = emu::tid();
  s[t] = x;
  __syncthreads();
  double r = 0.0;
  const int n = (int)(blockDim.x * blockDim.y * blockDim.z);
  for (int k = 0; k < 64 && (t & ~63) + k < n; ++k) r += s[(t & ~63) + k];
  __syncthreads();
  return r;
}
// the value lane `src` (0 .. 15) of this lane's 16-lane row holds
inline float group16_shfl(float x, int src) {
  float* s = reinterpret_cast<float*>(emu::g_xchg);
  const int t = emu::tid();
  s[t] = x;
  __syncthreads();
  const float r = s[(t & ~15) | (src & 15)];
  __syncthreads();
  return r;
}
inline float __shfl_xor(float x, int mask) {
  float* s = reinterpret_cast<float*>(emu::g_xchg);
  const int t = emu::tid();
  s[t] = x;
  __syncthreads();
  const float r = s[((t & 63) ^ mask) | (t & ~63)];
  __syncthreads();
  return r;
}

#define MPPO_SCHED_FENCE()
#define MPPO_WAVE_SYNC() __syncthreads()  // (every thread of the workgroup reaches the same point: the kernels use it workgroup-uniformly)
inline void stream_store(float* p, float v) { *p = v; }
inline void stream_store(float* p, float2 v) { *reinterpret_cast<float2*>(p) = v; }
inline void stream_store(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
inline void wt_store(float* base, size_t idx, float4 v) { *reinterpret_cast<float4*>(base + idx) = v; }
struct BufView { const char* base; unsigned bytes; };
inline BufView make_buf(const float* base, unsigned bytes) { return BufView{reinterpret_cast<const char*>(base), bytes}; }
inline float2 buf_load_f2(const BufView& b, int lane_off_bytes, int uniform_off_bytes) {  // per-dword range check, 0 outside (as the hardware)
  const unsigned off = (unsigned)lane_off_bytes + (unsigned)uniform_off_bytes;
  float2 q = make_float2(0.f, 0.f);
  if (off + 4 <= b.bytes) memcpy(&q.x, b.base + off, 4);
  if (off + 8 <= b.bytes) memcpy(&q.y, b.base + off + 4, 4);
  return q;
}
inline float4 buf_load_f4(const BufView& b, int lane_off_bytes, int uniform_off_bytes) {
  const unsigned long long off = (unsigned long long)(unsigned)lane_off_bytes + (unsigned)uniform_off_bytes;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < 4; ++c) if (off + 4 * c + 4 <= b.bytes) memcpy(&v[c], b.base + off + 4 * c, 4);
  return make_float4(v[0], v[1], v[2], v[3]);
}
#define MPPO_INTERLEAVE_MFMA16(NVALU, NVMEM)

// lane l supplies A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]; acc[r] = D[4*(l>>4) + r][l&15]
inline void mfma_f32_16x16x4(float a, float b, f32x4& acc) {
  const int t = emu::tid(), lane = t & 63, wave = t >> 6;
  float* A = reinterpret_cast<float*>(emu::g_xchg) + wave * 128;
  float* B = A + 64;
  A[lane] = a; B[lane] = b;
  __syncthreads();
  const int col = lane & 15, q = lane >> 4;
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * q + r;
    float c = acc[r];
    for (int k = 0; k < 4; ++k) c = fmaf(A[16 * k + row], B[16 * k + col], c);
    acc[r] = c;
  }
  __syncthreads();
}

// agent-scope exchange between workgroups: the emulator runs workgroups one after another in one thread
inline void agent_store(float* p, float v) { *p = v; }
inline float agent_load(const float* p) { return *p; }
inline float2 agent_load2(const float* p) { return make_float2(p[0], p[1]); }
inline int agent_fetch_add(int* p, int v) { const int o = *p; *p = o + v; return o; }
inline int wave_uniform(int x) { return x; }
inline void wg_release_fence() {}
inline void agent_acquire_fence() {}

// system-scope accesses between rank PROCESSES (csrc/peer.h): the emulator's exchange buffers are POSIX shared memory
#include <sched.h>
#include <time.h>
// (a 16-byte access is two 8-byte single-copy-atomic halves: the tagged exchange of csrc/peer.h keeps a value and its tag in one half)
inline void sys_store_f4(void* base, size_t byte_off, float4 v) {
  unsigned long long* p = reinterpret_cast<unsigned long long*>(static_cast<char*>(base) + byte_off);
  __atomic_store_n(p, (unsigned long long)__float_as_uint(v.x) | ((unsigned long long)__float_as_uint(v.y) << 32), __ATOMIC_RELEASE);
  __atomic_store_n(p + 1, (unsigned long long)__float_as_uint(v.z) | ((unsigned long long)__float_as_uint(v.w) << 32), __ATOMIC_RELEASE);
}
inline float4 sys_load_f4(const void* base, size_t byte_off) {
  const unsigned long long* p = reinterpret_cast<const unsigned long long*>(static_cast<const char*>(base) + byte_off);
  const unsigned long long a = __atomic_load_n(p, __ATOMIC_ACQUIRE), b = __atomic_load_n(p + 1, __ATOMIC_ACQUIRE);
  return make_float4(__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)b), __uint_as_float((unsigned)(b >> 32)));
}
inline void sys_store_f2(float* p, float2 v) {
  __atomic_store_n(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__float_as_uint(v.x) | ((unsigned long long)__float_as_uint(v.y) << 32), __ATOMIC_RELEASE);
}
inline float2 sys_load_f2(const float* p) {
  const unsigned long long a = __atomic_load_n(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_ACQUIRE);
  return make_float2(__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)));
}
inline void sys_store_f32(float* p, float v) { __atomic_store_n(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED); }
inline float sys_load_f32(const float* p) { return __uint_as_float(__atomic_load_n(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED)); }
inline void sys_store_i32(int* p, int v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }  // flags: release / acquire order the payload
inline int sys_load_i32(const int* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
inline int sys_poll_rmw(const int* p) { return __atomic_fetch_or(const_cast<int*>(p), 0, __ATOMIC_ACQ_REL); }
inline void sys_store_f64(double* p, double v) { unsigned long long u; memcpy(&u, &v, 8); __atomic_store_n(reinterpret_cast<unsigned long long*>(p), u, __ATOMIC_RELAXED); }
inline double sys_load_f64(const double* p) { const unsigned long long u = __atomic_load_n(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED); double v; memcpy(&v, &u, 8); return v; }
inline void drain_stores() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline void spin_pause() { sched_yield(); }
inline unsigned long long realtime_ticks() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (unsigned long long)ts.tv_sec * 100000000ull + (unsigned long long)ts.tv_nsec / 10ull; }
