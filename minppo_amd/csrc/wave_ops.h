// wave_ops.h — the only place where gfx950 cross-lane / matrix-core builtins appear.
// Kernels talk to the hardware through these helpers (64-wide wavefronts, DPP row
// operations inside 16-lane rows, f32-input MFMA).
#pragma once
#include <hip/hip_runtime.h>

#define MPPO_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

// D(32x32) += A(32x2) * B(2x32), exact f32 (v_mfma_f32_32x32x2_f32).
// lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
// acc[r] is D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
__device__ __forceinline__ void mfma_f32_32x32x2(float a, float b, f32x16& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
}

// D(32x32) += A(32x16) * B(16x32), bf16 inputs, f32 accumulate (v_mfma_f32_32x32x16_bf16).
// lane l (r = l&31, h = l>>5) supplies A[r][k = 8h + j] and B[k = 8h + j][r], j = 0..7.
__device__ __forceinline__ void mfma_bf16_32x32x16(bf16x8 a, bf16x8 b, f32x16& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}

// ---- bf16-in / f32-accumulate variants (training.mlp_dtype = "bf16", BASELINE configs[3]) ----
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_native __attribute__((ext_vector_type(2)));
typedef float f32x2_native __attribute__((ext_vector_type(2)));

// four floats -> four bf16 (round to nearest even; v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16x4 pack_bf16x4(float a, float b, float c, float d) {
  const bf16x2_native lo = __builtin_convertvector(f32x2_native{a, b}, bf16x2_native);
  const bf16x2_native hi = __builtin_convertvector(f32x2_native{c, d}, bf16x2_native);
  typedef short short2_native __attribute__((ext_vector_type(2)));
  const short2_native l = __builtin_bit_cast(short2_native, lo), h = __builtin_bit_cast(short2_native, hi);
  return bf16x4{l[0], l[1], h[0], h[1]};
}

// D(16x16) += A(16x16) * B(16x16): lane l supplies A[i = l&15][k = 4*(l>>4) + c] and B[k = 4*(l>>4) + c][j = l&15], c = 0..3;
// acc[r] is D[row = 4*(l>>4) + r][col = l&15]  (v_mfma_f32_16x16x16_bf16, 8 passes: 4x the k of the f32 16x16x4 per issue slot)
// four bf16 values that arrived as the raw bits of two floats (a 16-byte load of a bf16 array)
__device__ __forceinline__ bf16x4 bf16x4_from_bits(float lo, float hi) {
  typedef float f32x2_bits __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(bf16x4, f32x2_bits{lo, hi});
}
// the other way round: four bf16 values as the raw bits of two floats (an 8-byte store into a bf16 array), and their values as floats
__device__ __forceinline__ float2 bf16x4_bits(bf16x4 v) {
  typedef float f32x2_bits __attribute__((ext_vector_type(2)));
  const f32x2_bits b = __builtin_bit_cast(f32x2_bits, v);
  return make_float2(b[0], b[1]);
}
__device__ __forceinline__ float4 bf16x4_unpack(float lo, float hi) {
  const unsigned a = __float_as_uint(lo), b = __float_as_uint(hi);
  return make_float4(__uint_as_float(a << 16), __uint_as_float(a & 0xFFFF0000u), __uint_as_float(b << 16), __uint_as_float(b & 0xFFFF0000u));
}
__device__ __forceinline__ void mfma_bf16_16x16x16(bf16x4 a, bf16x4 b, f32x4& acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc, 0, 0, 0);
}

// D(16x16) += A(16x32) * B(32x16): lane l supplies A[i = l&15][k = 8*(l>>4) + c] and B[k = 8*(l>>4) + c][j = l&15], c = 0..7, as the raw
// bits of eight bf16 in a 16-byte register; acc as mfma_bf16_16x16x16  (v_mfma_f32_16x16x32_bf16, gfx950: twice the k of 16x16x16 per
// instruction at ~16 cycles per SIMD - MI355X_MICROARCH.md, cycle constants)
__device__ __forceinline__ void mfma_bf16_16x16x32(const float4& a_bits, const float4& b_bits, f32x4& acc) {
  typedef __bf16 bf16x8_native __attribute__((ext_vector_type(8)));
  typedef float f32x4_bits __attribute__((ext_vector_type(4)));
  const bf16x8_native A = __builtin_bit_cast(bf16x8_native, f32x4_bits{a_bits.x, a_bits.y, a_bits.z, a_bits.w});
  const bf16x8_native B = __builtin_bit_cast(bf16x8_native, f32x4_bits{b_bits.x, b_bits.y, b_bits.z, b_bits.w});
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc, 0, 0, 0);
}

// D(32x32) += A(32x8) * B(8x32): lane l supplies A[i = l&31][k = 4*(l>>5) + c] and B[k = 4*(l>>5) + c][j = l&31];
// acc layout as mfma_f32_32x32x2  (v_mfma_f32_32x32x8_bf16)
__device__ __forceinline__ void mfma_bf16_32x32x8(bf16x4 a, bf16x4 b, f32x16& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, acc, 0, 0, 0);
}

template <int CTRL>
__device__ __forceinline__ float dpp_row(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}

// All-reduce inside each 16-lane DPP row: every lane of the row gets the row's sum (row_ror 8,4,2,1).
__device__ __forceinline__ float group16_sum(float x) {
  x += dpp_row<0x128>(x);
  x += dpp_row<0x124>(x);
  x += dpp_row<0x122>(x);
  x += dpp_row<0x121>(x);
  return x;
}
__device__ __forceinline__ float group16_max(float x) {
  x = fmaxf(x, dpp_row<0x128>(x));
  x = fmaxf(x, dpp_row<0x124>(x));
  x = fmaxf(x, dpp_row<0x122>(x));
  x = fmaxf(x, dpp_row<0x121>(x));
  return x;
}
// the value lane `src` (0 .. 15) of this lane's 16-lane row holds (ds_bpermute: the source may differ from row to row)
__device__ __forceinline__ float group16_shfl(float x, int src) { return __shfl(x, (int)((threadIdx.x & 63) & ~15) | (src & 15), 64); }
// true in every lane of the wave if pred holds in any lane of the wave
__device__ __forceinline__ bool wave_any(bool pred) { return __any(pred); }
// true in every lane of a 16-lane row if pred holds in any lane of that row
__device__ __forceinline__ bool group16_any(bool pred) {
  const unsigned long long m = __ballot(pred);
  const int row = (threadIdx.x & 63) >> 4;
  return ((m >> (16 * row)) & 0xffffull) != 0ull;
}
// A per-environment flag that has to survive a long, register-starved stretch of a kernel: the wave's ballot - ONE scalar value (an SGPR pair) for the four
// 16-lane rows, not a register per lane (k_physics.hip says what happened to the per-lane form).
typedef unsigned long long group16_flags_t;
__device__ __forceinline__ group16_flags_t group16_flags(bool pred) { return __ballot(pred); }
__device__ __forceinline__ group16_flags_t group16_flags_or(group16_flags_t a, group16_flags_t b) { return a | b; }
__device__ __forceinline__ bool group16_flag_set(group16_flags_t m) { return ((m >> (16 * (int)((threadIdx.x & 63) >> 4))) & 0xffffull) != 0ull; }
// full-wave sum, result in every lane
__device__ __forceinline__ float wave_sum(float x) {
  x = group16_sum(x);
  x += __shfl_xor(x, 16);
  x += __shfl_xor(x, 32);
  return x;
}
__device__ __forceinline__ double wave_sum_f64(double x) {
  for (int m = 1; m < 64; m <<= 1) x += __shfl_xor(x, m);
  return x;
}

// Scheduling hint for a block of 16 dependent MFMAs followed/preceded by other work in the same basic block: after every
// MFMA place NVALU vector-ALU and NVMEM vector-memory-read instructions.  The dependent f32 MFMA chain leaves a 64-cycle
// gap after each issue; in-order issue would otherwise run the ~100 address / load instructions of the next k-set
// strictly before the whole chain.  (LLVM sched_group_barrier masks: 0x8 MFMA, 0x2 VALU, 0x20 VMEM read.)
// ---- raw buffer loads: 32-bit lane offset + scalar offset against a range-checked descriptor; a read past `bytes` returns 0
// (the hardware's bounds check does the K-tail of the first layer: no clamp arithmetic, no 64-bit lane addresses).
// The intrinsic is bound by name: clang's __builtin_amdgcn_raw_buffer_load_b64 / _b128 of this ROCm release are lowered to a
// single buffer_load_dword (upper lanes of the result undefined) - measured, see DESIGN.md. ----
typedef int i32x4_rsrc __attribute__((ext_vector_type(4)));
typedef float f32x2_native __attribute__((ext_vector_type(2)));
__device__ f32x2_native mppo_raw_buffer_load_f32x2(i32x4_rsrc rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");

struct BufView {
  i32x4_rsrc r;
};
__device__ __forceinline__ BufView make_buf(const float* base, unsigned bytes) {
  union { i32x4_rsrc v; struct { const float* p; unsigned n; unsigned f; } s; } u;
  u.s.p = base;        // words 0-1: 48-bit base, stride 0 (raw buffer)
  u.s.n = bytes;       // word 2: num_records in bytes
  u.s.f = 0x00020000;  // word 3: 32-bit data format (gfx9 family)
  BufView b;
  b.r = u.v;
  return b;
}
typedef float f32x4_native __attribute__((ext_vector_type(4)));
__device__ f32x4_native mppo_raw_buffer_load_f32x4(i32x4_rsrc rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ __forceinline__ float4 buf_load_f4(const BufView& b, int lane_off_bytes, int uniform_off_bytes) {
  const f32x4_native q = mppo_raw_buffer_load_f32x4(b.r, lane_off_bytes, uniform_off_bytes, 0);
  return make_float4(q.x, q.y, q.z, q.w);
}
__device__ __forceinline__ float2 buf_load_f2(const BufView& b, int lane_off_bytes, int uniform_off_bytes) {
  const f32x2_native q = mppo_raw_buffer_load_f32x2(b.r, lane_off_bytes, uniform_off_bytes, 0);
  return make_float2(q.x, q.y);
}

// write-through stores (sc0 sc1): data written once for ANOTHER kernel leaves this XCD's L2 as it is written instead of staying
// dirty until the end-of-kernel release has to flush it (MI355X_MICROARCH.md, "boundary": + B / 6 TB/s for B dirty bytes; "publish-large":
// write-through wins for tens of KB per workgroup).  A raw buffer store bound by name, cache policy in `aux` (1 = sc0, 2 = nt, 16 = sc1).
#ifndef MPPO_WT_STORES
#define MPPO_WT_STORES 1
#endif
__device__ void mppo_raw_buffer_store_f32x4(f32x4_native data, i32x4_rsrc rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");
// streaming stores: data written once for ANOTHER kernel (possibly on another XCD) goes to memory without staying dirty in this
// XCD's L2 - the end-of-kernel write-back has less to flush (MPPO_NT_STORES=0 at compile time restores plain stores)
#ifndef MPPO_NT_STORES
#define MPPO_NT_STORES 1
#endif
__device__ __forceinline__ void stream_store(float* p, float2 v) {
#if MPPO_NT_STORES
  typedef float f32x2_st __attribute__((ext_vector_type(2)));
  __builtin_nontemporal_store(f32x2_st{v.x, v.y}, reinterpret_cast<f32x2_st*>(p));
#else
  *reinterpret_cast<float2*>(p) = v;
#endif
}
__device__ __forceinline__ void stream_store(float* p, float v) {
#if MPPO_NT_STORES
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
__device__ __forceinline__ void stream_store(float* p, float4 v) {
#if MPPO_NT_STORES
  typedef float f32x4_st __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(f32x4_st{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4_st*>(p));
#else
  *reinterpret_cast<float4*>(p) = v;
#endif
}

// ---- agent-scope accesses for data exchanged between workgroups INSIDE one kernel (k_wgrad.hip).  The eight XCDs have private
// L2s: relaxed atomics at agent scope are write-through stores / cache-bypassing loads (sc1 on gfx950), which is what makes a
// partial result written on one XCD readable on another before the kernel ends.
__device__ __forceinline__ void agent_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float agent_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float2 agent_load2(const float* p) {  // p 8-byte aligned
  const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float2(__builtin_bit_cast(float, (unsigned)u), __builtin_bit_cast(float, (unsigned)(u >> 32)));
}
__device__ __forceinline__ int agent_fetch_add(int* p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wg_release_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }  // own stores acknowledged
__device__ __forceinline__ void agent_acquire_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }   // drop stale cache lines

// a value the program knows to be the same in every lane of the wave (e.g. threadIdx.x >> 6), moved to a scalar register: the
// compiler cannot prove that by itself and would otherwise wrap every buffer load whose scalar offset depends on it in a
// waterfall loop (measured on k_wgrad.hip: 21 us -> 14 us)
__device__ __forceinline__ int wave_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }

// wave-level synchronisation point for LDS words that ONE wavefront writes and reads back (its LDS instructions execute in program order:
// all that is needed is that the compiler keeps that order; the emulator, whose lanes are fibers, yields here instead - tests/emu/wave_ops.h)
#define MPPO_WAVE_SYNC() __builtin_amdgcn_wave_barrier()

// nothing is scheduled across this point (compiler-only; no instruction is emitted)
#define MPPO_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

#define MPPO_INTERLEAVE_MFMA16(NVALU, NVMEM)                      \
  _Pragma("unroll") for (int _i = 0; _i < 16; ++_i) {             \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            \
    __builtin_amdgcn_sched_group_barrier(0x002, NVALU, 0);        \
    __builtin_amdgcn_sched_group_barrier(0x020, NVMEM, 0);        \
  }

// D(16x16) += A(16x4) * B(4x16), exact f32 (v_mfma_f32_16x16x4_f32, 32 cycles).
// lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; acc[r] is D[row = 4*(l>>4) + r][col = l&15].
__device__ __forceinline__ void mfma_f32_16x16x4(float a, float b, f32x4& acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
}

// 16 bytes at base[idx .. idx + 4): `base` must be wave-uniform (it becomes the buffer descriptor), idx * 4 below 2 GB
__device__ __forceinline__ void wt_store(float* base, size_t idx, float4 v) {
#if MPPO_WT_STORES
  const BufView b = make_buf(base, 0x7FFFFFFFu);
  mppo_raw_buffer_store_f32x4(f32x4_native{v.x, v.y, v.z, v.w}, b.r, (int)(idx * 4), 0, 17);  // cache policy bits of this target: 1 = sc0, 2 = nt, 16 = sc1
#else
  stream_store(base + idx, v);
#endif
}

// ---- system-scope accesses for data exchanged between PROCESSES / GPUs (csrc/peer.h: the ranks' gradient exchange through hipIpc
// mappings).  sc0 sc1 on gfx950: a store writes through to the memory that owns the address (a peer's HBM over xGMI for a mapped
// peer buffer) and a load bypasses this GPU's L1 and L2.  `base` must be wave-uniform (it becomes the buffer descriptor).
__device__ __forceinline__ void sys_store_f4(void* base, size_t byte_off, float4 v) {
  const BufView b = make_buf(static_cast<const float*>(base), 0x7FFFFFFFu);
  mppo_raw_buffer_store_f32x4(f32x4_native{v.x, v.y, v.z, v.w}, b.r, (int)byte_off, 0, 17);
}
__device__ __forceinline__ float4 sys_load_f4(const void* base, size_t byte_off) {
  const BufView b = make_buf(static_cast<const float*>(base), 0x7FFFFFFFu);
  const f32x4_native q = mppo_raw_buffer_load_f32x4(b.r, (int)byte_off, 0, 17);
  return make_float4(q.x, q.y, q.z, q.w);
}
// a value and its tag as ONE 8-byte access (p 8-byte aligned)
__device__ __forceinline__ void sys_store_f2(float* p, float2 v) {
  const unsigned long long u = (unsigned long long)__builtin_bit_cast(unsigned, v.x) | ((unsigned long long)__builtin_bit_cast(unsigned, v.y) << 32);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ float2 sys_load_f2(const float* p) {
  const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return make_float2(__builtin_bit_cast(float, (unsigned)u), __builtin_bit_cast(float, (unsigned)(u >> 32)));
}
__device__ __forceinline__ void sys_store_f32(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ float sys_load_f32(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void sys_store_i32(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ int sys_load_i32(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ int sys_poll_rmw(const int* p) { return __hip_atomic_fetch_or(const_cast<int*>(p), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void sys_store_f64(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ double sys_load_f64(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// every store of this wave has been acknowledged by the memory it went to (inline asm: the compiler may not drop or move it,
// MI355X_MICROARCH.md "Compiler hazard")
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void spin_pause() { __builtin_amdgcn_s_sleep(2); }
__device__ __forceinline__ unsigned long long realtime_ticks() { return __builtin_amdgcn_s_memrealtime(); }  // 100 MHz, constant
