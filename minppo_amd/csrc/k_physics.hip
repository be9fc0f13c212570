// k_physics.hip — the vectorised environment step as ONE kernel.
//
// Replaces, for N environments at once, reference `HumanoidEnv.step` / `.reset`
// (minppo/env.py:124-196) including the third-party `pipeline_step` / `pipeline_init` it calls
// (env.py:120,162 -> brax.mjx.pipeline -> mujoco.mjx.step/forward with solver = CG, 6 iterations,
// 6 line-search iterations, env.py:95-97), `compute_reward` (env.py:199-235), `is_done`
// (env.py:238-242), the NaN guard (env.py:173-176), the auto-reset select (env.py:179-180),
// `get_obs` (env.py:245-261) and the episode metrics (env.py:183-194).
//
// Mapping (MI355X): 16 lanes (one DPP row) cooperate on one environment, 4 environments per 64-wide wavefront, one to four
// wavefronts per workgroup (chosen per model in mppo_model_open: whatever puts the most waves on a CU's 160 KB of LDS; 1024
// one-wave workgroups at N = 4096 for the 16-dof stand-in = one wave per SIMD on 256 CUs).  Every per-environment array lives in
// LDS, laid out by lifetime (model_view.h: 6.5 KB per environment for the 16-dof stand-in, 12.9 KB for the 26-dof robot); lanes own
// dofs / bodies / constraint rows with stride 16, reductions run as DPP row rotations (wave_ops.h), and the only sequential parts
// are the kinematic-tree levels and the Cholesky columns.  All tree recursions of MuJoCo (composite inertia, body velocities, RNE)
// are evaluated in closed form over ancestor / subtree bitmasks from the model blob, so they need no level-by-level
// synchronisation.  HBM traffic per environment step: the state record in and out, the action in, the observation out (about
// 2.9 KB at O = 225).  The kernel is bound by the instructions one wave issues, not by bandwidth.
#include "model_view.h"
#include "mppo_common.h"
#include <wave_ops.h>

#include <algorithm>
#include <cstring>
#include <vector>
#include <initializer_list>
#include <type_traits>
#include <utility>

namespace mppo {

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose counter is a compile-time constant in the body
// (the unroll pragma gives up on the 26-step elimination loop of the larger robot; this cannot)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }


struct EnvArgs {
  int N, mode, n_frames;  // mode 0: reset (pipeline_init), 1: step, 2: probe (one forward on given inputs)
  float* state;
  const float* reset_in;
  float* reset_out;
  const float* action;
  int act_ld;
  float* obs;
  int obs_ld;
  float* reward;
  unsigned char* done;
  mppo_env_metrics_t met;
  mppo_reward_cfg_t rc;
  const float *p_qpos, *p_qvel, *p_ctrl, *p_warm;
  mppo_forward_probe_t probe;
  float* scratch;  // per-environment records in global memory for the matrices a large robot keeps out of LDS (PhysLds::gwords floats each; null if none)
};

// wave-level synchronisation point: an environment's LDS arrays are touched by ONE wavefront only, whose LDS instructions
// execute in program order - all that is needed is that the compiler keeps that order (the emulator yields here instead)
#ifdef MPPO_EMU
#define SYNC() __syncthreads()
#else
#define SYNC() __builtin_amdgcn_wave_barrier()
#endif
// ... and for the matrices a large robot keeps in global memory (PhysLds::spill): an environment's record is written and read by ONE
// wavefront, whose vector-memory instructions reach the CU's L1 / the L2 in program order; the fence makes the compiler wait for the
// stores (s_waitcnt) before anything that follows
#ifdef MPPO_EMU
#define GSYNC() __syncthreads()
#else
#define GSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif
#define FOR_G(i, n) for (int i = g; i < (n); i += kGroupLanes)
#ifndef MPPO_DOT_UNROLL_RT
#define MPPO_DOT_UNROLL_RT 8
#endif
// How many terms of a dot product are requested from LDS before the first is consumed (one wave per SIMD: nothing else hides an LDS
// round trip, and a third of a wave's life is spent waiting for LDS).  Measured (tools/env_time.py): 8 / 16 / 32 terms give 89.9 / 85.5 /
// 87.9 us for the 16-dof robot and 547 / 531 / 510 us for the 26-dof one: a whole row of M or J at a time.  The summation order does
// not change.  kDotU is a constant of the enclosing function (model-specialised kernels: by the number of dofs; run-time-sized: 8).
#define DOT_UNROLL _Pragma("unroll kDotU")
constexpr int dot_unroll(int nv_static) { return nv_static == 0 ? MPPO_DOT_UNROLL_RT : nv_static <= 16 ? 16 : 32; }
// the instruction scheduler moves nothing across this point
#ifdef MPPO_EMU
#define SCHED_FENCE() do { } while (0)
#else
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
// Phase timers (profiling builds only, -DMPPO_PHYS_TIMERS: tools/env_phases.py): wave 0 of workgroup 0 stamps s_memtime at the phase
// boundaries of its first frame into a device array that mppo_debug_phys_timers() copies out.
#ifdef MPPO_PHYS_TIMERS
__device__ unsigned long long g_phys_t[40];
#define PT(k) do { if (blockIdx.x == 0 && tid == 0) g_phys_t[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PT(k) do { } while (0)
#endif
// model tables in LDS (see the staging copy at the top of the kernel)
#define TI(name) (tabI + MVO(BI_##name))
#define TF(name) (tabF + MVO(BF_##name))
#define TU(name) (reinterpret_cast<const u64*>(tabI + MVO(BI_##name)))
#define MVO(k) (SD::kStatic ? kSO.o[k] : mv.o[k])  // word offset of table k inside the blob: a constant in a model-specialised kernel

// ---- small vector helpers (registers) -------------------------------------------------------
struct V3 { float x, y, z; };
struct Q4 { float w, x, y, z; };
__device__ __forceinline__ V3 ld3(const float* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ void st3(float* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
__device__ __forceinline__ Q4 ld4(const float* p) { return {p[0], p[1], p[2], p[3]}; }
__device__ __forceinline__ void st4(float* p, Q4 q) { p[0] = q.w; p[1] = q.x; p[2] = q.y; p[3] = q.z; }
__device__ __forceinline__ V3 add3(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 sub3(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 mul3(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross3(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ Q4 qmul(Q4 a, Q4 b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__device__ __forceinline__ Q4 qnormalize(Q4 q) {
  const float n = sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  const float s = n > 0.f ? 1.f / n : 1.f;
  return {q.w * s, q.x * s, q.y * s, q.z * s};
}
// rotation matrix of a (unit) quaternion, row-major in m[9]
__device__ __forceinline__ void qmat(Q4 q, float* m) {
  const float w = q.w, x = q.x, y = q.y, z = q.z;
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2.f * (x * y - w * z);         m[2] = 2.f * (x * z + w * y);
  m[3] = 2.f * (x * y + w * z);         m[4] = w * w - x * x + y * y - z * z; m[5] = 2.f * (y * z - w * x);
  m[6] = 2.f * (x * z - w * y);         m[7] = 2.f * (y * z + w * x);         m[8] = w * w - x * x - y * y + z * z;
}
__device__ __forceinline__ V3 qrot(Q4 q, V3 v) {
  float m[9];
  qmat(q, m);
  return {m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z};
}
// MJX math.normalize_with_norm
__device__ __forceinline__ V3 normalize_norm(V3 v, float& n) {
  n = sqrtf(dot3(v, v));
  return mul3(v, 1.f / (n + (n == 0.f ? 1e-6f : 0.f)));
}
// MJX math.closest_segment_point
__device__ __forceinline__ V3 closest_segment_point(V3 a, V3 b, V3 pt) {
  const V3 ab = sub3(b, a);
  const float t = dot3(sub3(pt, a), ab) / (dot3(ab, ab) + 1e-6f);
  return add3(a, mul3(ab, fminf(fmaxf(t, 0.f), 1.f)));
}
// MJX math.closest_segment_to_segment_points (sphere_capsule / capsule_capsule; a sphere is a segment of length 0)
__device__ __forceinline__ void closest_segment_points(V3 a0, V3 a1, V3 b0, V3 b1, V3& best_a, V3& best_b) {
  float len_a, len_b;
  const V3 dir_a = normalize_norm(sub3(a1, a0), len_a), dir_b = normalize_norm(sub3(b1, b0), len_b);
  const float half_a = 0.5f * len_a, half_b = 0.5f * len_b;
  const V3 a_mid = add3(a0, mul3(dir_a, half_a)), b_mid = add3(b0, mul3(dir_b, half_b));
  const V3 trans = sub3(a_mid, b_mid);
  const float dab = dot3(dir_a, dir_b), dat = dot3(dir_a, trans), dbt = dot3(dir_b, trans);
  const float denom = 1.f - dab * dab;
  float ta = (-dat + dab * dbt) / (denom + 1e-6f);
  float tb = dbt + ta * dab;
  ta = fminf(fmaxf(ta, -half_a), half_a);
  tb = fminf(fmaxf(tb, -half_b), half_b);
  best_a = add3(a_mid, mul3(dir_a, ta));
  best_b = add3(b_mid, mul3(dir_b, tb));
  const V3 new_a = closest_segment_point(a0, a1, best_b), new_b = closest_segment_point(b0, b1, best_a);
  const V3 e1 = sub3(new_a, best_b), e2 = sub3(new_b, best_a);
  if (dot3(e1, e1) < dot3(e2, e2)) best_a = new_a; else best_b = new_b;
}
// second row of MJX math.make_frame(n) for a unit n: y (or z when |n.y| >= 0.5) with its n component removed, normalised
__device__ __forceinline__ V3 frame_tangent(V3 n) {
  V3 b = (n.y > -0.5f && n.y < 0.5f) ? V3{0.f, 1.f, 0.f} : V3{0.f, 0.f, 1.f};
  b = sub3(b, mul3(n, dot3(n, b)));
  const float l = sqrtf(dot3(b, b));
  return mul3(b, l > 0.f ? 1.f / l : 1.f);
}
__device__ __forceinline__ Q4 axis_angle(V3 axis, float angle) {
  float s, c;
  sincosf(0.5f * angle, &s, &c);
  return {c, axis.x * s, axis.y * s, axis.z * s};
}
// ---- a sphere / capsule of one body against a convex hull (box, mesh) of another: MJX collision_convex._sphere_convex / _capsule_convex.
// The 16 lanes of an environment work on ONE pair together: lanes stride over the hull's faces (support) and edges (closest approach),
// the winners are found with DPP row reductions (the first index among equal values, as argmax / argmin give it), the winning face's
// polygon - a handful of vertices - is walked by every lane alike.  Tables come from the hull section in global memory (HullView).
struct HullTabs {
  const int* face_adr; const int* fidx; const int* edge;
  const float* vert; const float* fnormal; const float* enormal;
  const int* udadr; const float* udir;  // per hull: unit edge directions, parallel ones dropped (convex_convex)
};
// first index whose value equals the row's maximum (value `v` is this lane's best over its own indices, `i` its index; a lane without
// candidates passes -inf)
__device__ __forceinline__ int group16_argmax(float v, int i) {
  const float m = group16_max(v);
  return (int)-group16_max(v == m ? -(float)i : -INFINITY);
}
// MJX _clip_edge_to_planes for the side planes of one face (plane k: through p0[k] = the polygon's previous vertex, normal (p1 - p0) x n):
// an end in front of a plane moves to the line's intersection with it, the candidate most along the edge wins; both ends in front of one
// plane, or ends that crossed: the edge is kept and masked out
__device__ __forceinline__ bool clip_segment_to_face(const HullTabs& T, int a0, int a1, V3 n, V3& e0, V3& e1) {
  const V3 dir = sub3(e1, e0), rdir = sub3(e0, e1);
  V3 n0 = e0, n1 = e1;
  float d0 = 0.f, d1 = 0.f;  // (argmax over candidates that start as "p itself, dot 0" at plane 0)
  bool first = true, both = false;
  V3 prev = ld3(T.vert + 3 * T.fidx[a1 - 1]);
  for (int i = a0; i < a1; ++i) {
    const V3 cur = ld3(T.vert + 3 * T.fidx[i]);
    const V3 en = cross3(sub3(cur, prev), n);
    const float s0 = dot3(sub3(e0, prev), en), s1 = dot3(sub3(e1, prev), en);
    const bool in0 = s0 > 1e-6f, in1 = s1 > 1e-6f;
    const float denom = dot3(dir, en);
    const float tt = dot3(sub3(prev, e0), en) / (denom + (denom == 0.f ? 1e-6f : 0.f));
    const V3 cand = add3(e0, mul3(dir, tt));
    const V3 c0 = in0 ? cand : e0, c1 = in1 ? cand : e1;
    const float x0 = dot3(sub3(c0, e0), dir), x1 = dot3(sub3(c1, e1), rdir);
    if (first || x0 > d0) { d0 = x0; n0 = c0; }
    if (first || x1 > d1) { d1 = x1; n1 = c1; }
    first = false;
    both = both || (in0 && in1);
    prev = cur;
  }
  bool mask = !both;
  if (!mask) { n0 = e0; n1 = e1; }
  if (dot3(rdir, sub3(n0, n1)) < 0.f) mask = false;
  e0 = n0; e1 = n1;
  return mask;
}
// -> dist[2], pos[2], nrm[2] in the hull's frame (normal from the sphere / capsule into the hull); a sphere fills slot 0 only
__device__ __forceinline__ void hull_pair_contacts(const HullTabs& T, int f0, int f1, int ed0, int ed1, bool capsule, V3 cp, V3 half, float r, int g,
                                                   float* dist, V3* pos, V3* nrm) {
  const V3 c0 = sub3(cp, half), c1 = add3(cp, half);
  // the face with the least penetration among those the geom's lowest point is behind
  float best = -INFINITY;
  int bi = 0;
  bool allneg = true;
  for (int f = f0 + g; f < f1; f += kGroupLanes) {
    const V3 n = ld3(T.fnormal + 3 * f), v0 = ld3(T.vert + 3 * T.fidx[T.face_adr[f]]);
    float sup = fminf(dot3(sub3(c0, v0), n), dot3(sub3(c1, v0), n)) - r;
    allneg = allneg && sup < 0.f;
    if (sup >= 0.f) sup = -1e12f;
    if (sup > best) { best = sup; bi = f; }
  }
  const int fb = group16_argmax(best, bi);
  const bool has_support = !group16_any(!allneg);
  const V3 n = ld3(T.fnormal + 3 * fb);
  const int a0 = T.face_adr[fb], a1 = T.face_adr[fb + 1];
  const V3 v0 = ld3(T.vert + 3 * T.fidx[a0]);
  if (!capsule) {
    // the centre projected on the face's plane; outside the polygon: on the nearest edge it is in front of
    V3 pt = sub3(cp, mul3(n, dot3(sub3(cp, v0), n)));
    bool inside = true;
    float emin = INFINITY;
    V3 ea = v0, eb = v0;
    bool any = false;
    V3 prev = ld3(T.vert + 3 * T.fidx[a1 - 1]);
    for (int i = a0; i < a1; ++i) {
      const V3 cur = ld3(T.vert + 3 * T.fidx[i]);
      const V3 en = cross3(sub3(cur, prev), n);
      float ed = dot3(sub3(pt, prev), en);
      inside = inside && ed <= 0.f;
      if (ed < 0.f || (en.x == 0.f && en.y == 0.f && en.z == 0.f)) ed = 1e12f;
      if (!any || ed < emin) { emin = ed; ea = prev; eb = cur; any = true; }
      prev = cur;
    }
    if (!inside) pt = closest_segment_point(ea, eb, pt);
    float d;
    const V3 nc = normalize_norm(sub3(pt, cp), d);
    dist[0] = d - r;
    pos[0] = mul3(add3(pt, add3(cp, mul3(nc, r))), 0.5f);
    nrm[0] = nc;
    dist[1] = 1.f; pos[1] = pos[0]; nrm[1] = nc;
    return;
  }
  // the capsule's segment clipped to the face's side planes: each clipped end is a contact against the face
  V3 q0 = c0, q1 = c1;
  const bool mask = clip_segment_to_face(T, a0, a1, n, q0, q1);
  float fpen[2];
  {
    const V3 qs[2] = {sub3(q0, mul3(n, r)), sub3(q1, mul3(n, r))};
    for (int j = 0; j < 2; ++j) {
      const V3 fp = sub3(qs[j], mul3(n, dot3(sub3(qs[j], v0), n)));
      pos[j] = mul3(add3(qs[j], fp), 0.5f);
      fpen[j] = (mask && has_support) ? dot3(sub3(fp, qs[j]), n) : -1.f;
    }
  }
  // the hull edge that comes closest to the segment: a shallow edge contact replaces slot 0
  float dbest = INFINITY;
  int eb_i = ed0;
  for (int e = ed0 + g; e < ed1; e += kGroupLanes) {
    V3 pe, pc;
    closest_segment_points(ld3(T.vert + 3 * T.edge[2 * e]), ld3(T.vert + 3 * T.edge[2 * e + 1]), c0, c1, pe, pc);
    const V3 dlt = sub3(pe, pc);
    const float dd = sqrtf(dot3(dlt, dlt));
    if (dd < dbest) { dbest = dd; eb_i = e; }
  }
  const int ew = group16_argmax(-dbest, eb_i);
  V3 pe, pc;
  closest_segment_points(ld3(T.vert + 3 * T.edge[2 * ew]), ld3(T.vert + 3 * T.edge[2 * ew + 1]), c0, c1, pe, pc);
  const V3 edir = sub3(pe, pc);
  const bool degenerate = dot3(edir, edir) < 1e-6f;
  float edist;
  const V3 eaxis = normalize_norm(edir, edist);
  const bool front = dot3(ld3(T.enormal + 6 * ew), eaxis) < 0.f && dot3(ld3(T.enormal + 6 * ew + 3), eaxis) < 0.f;
  const float epen = (!degenerate && front) ? r - edist : -1.f;
  const bool parallel = fabsf(dot3(eaxis, n)) > 0.99f && !degenerate;
  const float minf = fminf(fpen[0], fpen[1]);
  const bool has_edge = epen > 0.f && (minf > 0.f ? epen < minf : true) && !parallel;
  const V3 nn = {-n.x, -n.y, -n.z};
  if (has_edge) pos[0] = mul3(add3(pe, add3(pc, mul3(eaxis, r))), 0.5f);
  nrm[0] = has_edge ? eaxis : nn;
  nrm[1] = nn;
  dist[0] = -(has_edge ? epen : fpen[0]);
  dist[1] = -(has_edge ? -1.f : fpen[1]);
}
// a value pinned in a vector register (an empty asm the optimiser cannot look through): a select between pinned values stays a select -
// without it the compiler folds `q == 0 ? a[0] : q == 1 ? a[1] : ..` into a load from a[q], and an array indexed by a run-time value lives in scratch
#ifdef MPPO_EMU
#define MPPO_REG_PIN(x) do { } while (0)
#else
#define MPPO_REG_PIN(x) asm volatile("" : "+v"(x))
#endif
// ---- a box / mesh hull against a box / mesh hull of another body (round 6): the form of MJX collision_convex._box_box, for every hull pair.
// Everything in the frame of hull B's body; (R, tr) take hull A's body frame into it.  The environment's 16 lanes stride over the
// separating axes (A's face normals, B's face normals, the cross products of the hulls' edge directions), over the faces (reference /
// incident face) and over the clipped candidate points (four per lane, in registers); winners come from DPP row reductions that keep
// argmax / argmin's "first index among equals", a winner's coordinates from a row shuffle.  No LDS.
struct RigidT { float R[9]; V3 t; };
__device__ __forceinline__ V3 rt_rot(const RigidT& X, V3 v) { return {X.R[0] * v.x + X.R[1] * v.y + X.R[2] * v.z, X.R[3] * v.x + X.R[4] * v.y + X.R[5] * v.z, X.R[6] * v.x + X.R[7] * v.y + X.R[8] * v.z}; }
__device__ __forceinline__ V3 rt_rot_t(const RigidT& X, V3 v) { return {X.R[0] * v.x + X.R[3] * v.y + X.R[6] * v.z, X.R[1] * v.x + X.R[4] * v.y + X.R[7] * v.z, X.R[2] * v.x + X.R[5] * v.y + X.R[8] * v.z}; }
// vertex i of a face (its index list starts at a0) of hull A (transformed) or hull B
struct FacePoly { int a0, m; bool of_a; };
__device__ __forceinline__ V3 poly_vert(const HullTabs& T, const RigidT& X, const FacePoly& P, int i) {
  const V3 v = ld3(T.vert + 3 * T.fidx[P.a0 + i]);
  return P.of_a ? add3(rt_rot(X, v), X.t) : v;
}
// MJX _clip_edge_to_planes: the segment e0-e1 against the side planes of polygon P (plane k through vertex k - 1, normal (v_k - v_{k-1}) x n)
template <class GetV>
__device__ __forceinline__ bool clip_edge_to_poly(GetV getv, int m, V3 n, V3& e0, V3& e1) {
  const V3 dir = sub3(e1, e0), rdir = sub3(e0, e1);
  V3 n0 = e0, n1 = e1;
  float d0 = 0.f, d1 = 0.f;
  bool first = true, both = false;
  V3 prev = getv(m - 1);
  for (int i = 0; i < m; ++i) {
    const V3 cur = getv(i);
    const V3 en = cross3(sub3(cur, prev), n);
    const bool in0 = dot3(sub3(e0, prev), en) > 1e-6f, in1 = dot3(sub3(e1, prev), en) > 1e-6f;
    const float denom = dot3(dir, en);
    const float tt = dot3(sub3(prev, e0), en) / (denom + (denom == 0.f ? 1e-6f : 0.f));
    const V3 cand = add3(e0, mul3(dir, tt));
    const V3 c0 = in0 ? cand : e0, c1 = in1 ? cand : e1;
    const float x0 = dot3(sub3(c0, e0), dir), x1 = dot3(sub3(c1, e1), rdir);
    if (first || x0 > d0) { d0 = x0; n0 = c0; }
    if (first || x1 > d1) { d1 = x1; n1 = c1; }
    first = false;
    both = both || (in0 && in1);
    prev = cur;
  }
  bool mask = !both;
  if (!mask) { n0 = e0; n1 = e1; }
  if (dot3(rdir, sub3(n0, n1)) < 0.f) mask = false;
  e0 = n0; e1 = n1;
  return mask;
}
// -> dist[4] (1: slot unused), pos[4] (B's frame, on the reference face), nrm (from A to B, B's frame); identical in every lane of the row
__device__ __forceinline__ void hull_hull_contacts(const HullTabs& T, const RigidT& X, int fa0, int fa1, int fb0, int fb1, int va0, int va1, int vb0, int vb1,
                                                   int ua0, int ua1, int ub0, int ub1, int g, float* dist, V3* pos, V3& nrm) {
  const int nfa = fa1 - fa0, nfb = fb1 - fb0, nea = ua1 - ua0, neb = ub1 - ub0, nax = nfa + nfb + nea * neb;
  auto axis_of = [&](int k, bool& degenerate) -> V3 {
    degenerate = false;
    if (k < nfa) return rt_rot(X, ld3(T.fnormal + 3 * (fa0 + k)));
    if (k < nfa + nfb) return ld3(T.fnormal + 3 * (fb0 + k - nfa));
    const int e = k - nfa - nfb, j = e / nea, i = e - j * nea;
    const V3 cr = cross3(rt_rot(X, ld3(T.udir + 3 * (ua0 + i))), ld3(T.udir + 3 * (ub0 + j)));
    degenerate = dot3(cr, cr) < 1e-6f;
    float nn;
    return normalize_norm(cr, nn);
  };
  // per axis: the smaller of the two overlaps of the hulls' projections, and which way it points
  auto overlap = [&](V3 ax, bool degenerate, float& sign) -> float {
    const V3 axa = rt_rot_t(X, ax);
    const float off = dot3(X.t, ax);
    float amax = -INFINITY, amin = INFINITY, bmax = -INFINITY, bmin = INFINITY;
    for (int v = va0; v < va1; ++v) { const float p = dot3(ld3(T.vert + 3 * v), axa) + off; amax = fmaxf(amax, p); amin = fminf(amin, p); }
    for (int v = vb0; v < vb1; ++v) { const float p = dot3(ld3(T.vert + 3 * v), ax); bmax = fmaxf(bmax, p); bmin = fminf(bmin, p); }
    const float d1 = amax - bmin, d2 = bmax - amin;
    sign = d1 > d2 ? -1.f : 1.f;
    return degenerate ? 1e6f : fminf(d1, d2);
  };
  float best = INFINITY;
  int bk = 0;
  for (int k = g; k < nax; k += kGroupLanes) {
    bool dg;
    float sg_;
    const V3 ax = axis_of(k, dg);
    const float d = overlap(ax, dg, sg_);
    if (d < best) { best = d; bk = k; }
  }
  const int kbest = group16_argmax(-best, bk);
  bool dgb;
  float sg;
  const V3 axis = axis_of(kbest, dgb);
  (void)overlap(axis, dgb, sg);
  // the reference / incident faces: A's face most aligned with sg * axis, B's face most aligned with - sg * axis
  float va_ = -INFINITY, vb_ = -INFINITY;
  int ia = 0, ib = 0;
  for (int f = g; f < nfa; f += kGroupLanes) { const float x = dot3(rt_rot(X, ld3(T.fnormal + 3 * (fa0 + f))), axis) * sg; if (x > va_) { va_ = x; ia = f; } }
  for (int f = g; f < nfb; f += kGroupLanes) { const float x = dot3(ld3(T.fnormal + 3 * (fb0 + f)), axis) * -sg; if (x > vb_) { vb_ = x; ib = f; } }
  const int fa = fa0 + group16_argmax(va_, ia), fb = fb0 + group16_argmax(vb_, ib);
  const V3 na = rt_rot(X, ld3(T.fnormal + 3 * fa)), nb = ld3(T.fnormal + 3 * fb);
  const bool ref_a = fabsf(dot3(na, axis)) > fabsf(dot3(nb, axis));
  const FacePoly PA{T.face_adr[fa], T.face_adr[fa + 1] - T.face_adr[fa], true}, PB{T.face_adr[fb], T.face_adr[fb + 1] - T.face_adr[fb], false};
  const FacePoly ref = ref_a ? PA : PB, inc = ref_a ? PB : PA;
  const V3 ref_n = ref_a ? na : nb, inc_n = ref_a ? nb : na;
  const V3 ref0 = poly_vert(T, X, ref, 0), inc0 = poly_vert(T, X, inc, 0);
  // MJX _clip: the incident face's edges against the reference face's side planes, then the reference face's edges - projected onto the
  // incident plane along the reference normal - against the incident face's side planes: 2 ms + 2 mc candidate points, four per lane
  const int ms = inc.m, mc = ref.m, ncand = 2 * (ms + mc);
  const float pden = dot3(ref_n, inc_n), pd = dot3(inc0, inc_n);
  auto onto_inc_plane = [&](V3 p) { const float tt = (pd - dot3(p, inc_n)) / (pden + (pden == 0.f ? 1e-6f : 0.f)); return add3(p, mul3(ref_n, tt)); };
  V3 cp[4], cr_[4];     // candidate on the incident plane / projected onto the reference plane
  bool cm[4];
  float pen[4];
  static_for<4>([&](auto qc) {  // (a compile-time loop: the unroll pragma gives up on a body with inner loops, and the arrays would live in scratch)
    constexpr int q = decltype(qc)::value;
    const int c = g + kGroupLanes * q;
    cm[q] = false; cp[q] = ref0; cr_[q] = ref0; pen[q] = 0.f;
    if (c < ncand) {
      V3 e0, e1;
      bool m;
      bool second;
      if (c < 2 * ms) {
        const int i = c < ms ? c : c - ms;
        second = c >= ms;
        e0 = poly_vert(T, X, inc, i == 0 ? ms - 1 : i - 1); e1 = poly_vert(T, X, inc, i);
        m = clip_edge_to_poly([&](int k) { return poly_vert(T, X, ref, k); }, mc, ref_n, e0, e1);
      } else {
        const int c2 = c - 2 * ms, i = c2 < mc ? c2 : c2 - mc;
        second = c2 >= mc;
        e0 = onto_inc_plane(poly_vert(T, X, ref, i == 0 ? mc - 1 : i - 1)); e1 = onto_inc_plane(poly_vert(T, X, ref, i));
        m = clip_edge_to_poly([&](int k) { return poly_vert(T, X, inc, k); }, ms, inc_n, e0, e1);
      }
      const V3 p = second ? e1 : e0;
      const float h = dot3(sub3(p, ref0), ref_n);
      cp[q] = p;
      cr_[q] = sub3(p, mul3(ref_n, h));
      cm[q] = m && (-h > 1e-6f);
      pen[q] = dot3(sub3(p, cr_[q]), V3{-ref_n.x, -ref_n.y, -ref_n.z});
    }
  });
  // MJX _manifold_points over the candidates (values of non-candidates carry -1e6; argmax takes the first maximum in candidate order)
  auto pick = [&](auto value, int& cidx) {
    float bv = -INFINITY;
    int bc = 0;
    static_for<4>([&](auto qc) { constexpr int q = decltype(qc)::value; const int c = g + kGroupLanes * q; if (c < ncand) { const float x = value(qc); if (x > bv) { bv = x; bc = c; } } });
    // (a lane's candidates ascend with q: `>` keeps its first maximum; the row reduction keeps the first index among equal maxima)
    cidx = group16_argmax(bv, bc);
  };
  // (selects over a lane's four registers, never an indexed array: nothing of this lives in scratch)
  V3 r0 = cr_[0], r1 = cr_[1], r2 = cr_[2], r3 = cr_[3];
  MPPO_REG_PIN(r0.x); MPPO_REG_PIN(r0.y); MPPO_REG_PIN(r0.z); MPPO_REG_PIN(r1.x); MPPO_REG_PIN(r1.y); MPPO_REG_PIN(r1.z);
  MPPO_REG_PIN(r2.x); MPPO_REG_PIN(r2.y); MPPO_REG_PIN(r2.z); MPPO_REG_PIN(r3.x); MPPO_REG_PIN(r3.y); MPPO_REG_PIN(r3.z);
  auto fetch_pt = [&](int c) -> V3 {
    const int q = c >> 4, src = c & 15;
    const V3 mine = q == 0 ? r0 : q == 1 ? r1 : q == 2 ? r2 : r3;
    return {group16_shfl(mine.x, src), group16_shfl(mine.y, src), group16_shfl(mine.z, src)};
  };
  float dm[4], cmf[4];
  static_for<4>([&](auto qc) { constexpr int q = decltype(qc)::value; dm[q] = cm[q] ? 0.f : -1e6f; cmf[q] = cm[q] ? 1.f : 0.f; });
  float m0 = cmf[0], m1 = cmf[1], m2 = cmf[2], m3 = cmf[3], e0_ = pen[0], e1_ = pen[1], e2_ = pen[2], e3_ = pen[3];
  MPPO_REG_PIN(m0); MPPO_REG_PIN(m1); MPPO_REG_PIN(m2); MPPO_REG_PIN(m3); MPPO_REG_PIN(e0_); MPPO_REG_PIN(e1_); MPPO_REG_PIN(e2_); MPPO_REG_PIN(e3_);
  auto fetch_mask = [&](int c) -> float { const int q = c >> 4; return group16_shfl(q == 0 ? m0 : q == 1 ? m1 : q == 2 ? m2 : m3, c & 15); };
  auto fetch_pen = [&](int c) -> float { const int q = c >> 4; return group16_shfl(q == 0 ? e0_ : q == 1 ? e1_ : q == 2 ? e2_ : e3_, c & 15); };
  int idx[4];
  pick([&](auto qc) { return dm[decltype(qc)::value]; }, idx[0]);
  const V3 A_ = fetch_pt(idx[0]);
  pick([&](auto qc) { constexpr int q = decltype(qc)::value; const V3 e = sub3(A_, cr_[q]); return dot3(e, e) + dm[q]; }, idx[1]);
  const V3 B_ = fetch_pt(idx[1]);
  const V3 ab = cross3(ref_n, sub3(A_, B_));
  pick([&](auto qc) { constexpr int q = decltype(qc)::value; return fabsf(dot3(sub3(A_, cr_[q]), ab)) + dm[q]; }, idx[2]);
  const V3 C_ = fetch_pt(idx[2]);
  const V3 ac = cross3(ref_n, sub3(A_, C_)), bc = cross3(ref_n, sub3(B_, C_));
  {
    // the fourth point: the first maximum of the list [|(B - p) . bc|, then |(A - p) . ac|] over the candidates
    int i1, i2;
    pick([&](auto qc) { constexpr int q = decltype(qc)::value; return fabsf(dot3(sub3(B_, cr_[q]), bc)) + dm[q]; }, i1);
    pick([&](auto qc) { constexpr int q = decltype(qc)::value; return fabsf(dot3(sub3(A_, cr_[q]), ac)) + dm[q]; }, i2);
    const V3 p1 = fetch_pt(i1), p2 = fetch_pt(i2);
    const float v1 = fabsf(dot3(sub3(B_, p1), bc)) + (fetch_mask(i1) != 0.f ? 0.f : -1e6f), v2 = fabsf(dot3(sub3(A_, p2), ac)) + (fetch_mask(i2) != 0.f ? 0.f : -1e6f);
    idx[3] = v2 > v1 ? i2 : i1;
  }
  static_for<4>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    pos[j] = fetch_pt(idx[j]);
    dist[j] = fetch_mask(idx[j]) != 0.f ? -fetch_pen(idx[j]) : 1.f;
  });
  if (kbest >= nfa + nfb) {  // an edge-edge axis: the deepest point of the manifold alone (selects, not indexed: no scratch)
    float dk = dist[0];
    V3 pk = pos[0];
    static_for<3>([&](auto jc) { constexpr int j = decltype(jc)::value + 1; if (dist[j] < dk) { dk = dist[j]; pk = pos[j]; } });
    static_for<4>([&](auto jc) { constexpr int j = decltype(jc)::value; dist[j] = j == 0 ? dk : 1.f; pos[j] = pk; });
  }
  nrm = mul3(axis, sg);
}
// cinert (10) x spatial motion (6) -> spatial force (6)      (mju_mulInertVec)
__device__ __forceinline__ void inert_mul(const float* i, const float* v, float* r) {
  r[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  r[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  r[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  r[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  r[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  r[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
// spatial motion cross product vel x v (mju_crossMotion)
__device__ __forceinline__ void cross_motion(const float* vel, const float* v, float* r) {
  const V3 w = ld3(vel), l = ld3(vel + 3), a = ld3(v), b = ld3(v + 3);
  st3(r, cross3(w, a));
  st3(r + 3, add3(cross3(w, b), cross3(l, a)));
}
// spatial force cross product vel x* f (mju_crossForce)
__device__ __forceinline__ void cross_force(const float* vel, const float* f, float* r) {
  const V3 w = ld3(vel), l = ld3(vel + 3), a = ld3(f), b = ld3(f + 3);
  st3(r, add3(cross3(w, a), cross3(l, b)));
  st3(r + 3, cross3(w, b));
}

// x = (L L^T)^-1 b from the inverse factor Li, a packed lower triangle (row i at i (i + 1) / 2: Li[i][k], k <= i) - the factor of M
// during the step, of M + h*damp once the step's second factorisation has run.  `tmp` and `x` are nv-vectors in LDS; b may alias
// neither.  Contains two synchronisation points.
// Both triangular products run per 16-row pass over a lane-independent range of k (rows 16 q .. 16 q + 15: k < 16 (q + 1) for L^-1 b,
// k >= 16 q for L^-T t) with the entries outside the triangle masked to 0: a uniform trip count lets the compiler request eight LDS
// operands at a time and wait once, where a per-lane triangular loop waits for every single one (one wave per SIMD: nothing else
// hides an LDS round trip).  The terms left out are exact zeros: the sums equal the full-length ones of the register-resident form
// (solve_regs) bit for bit.
template <int NV>
__device__ __forceinline__ void solve_tri(const float* Li, int nv_rt, const float* b, float* tmp, float* x, int g) {
  const int nv = NV ? NV : nv_rt;
  constexpr int kDotU = dot_unroll(NV);
  for (int i0 = 0; i0 < nv; i0 += kGroupLanes) {
    const int i = i0 + g, ic = i < nv ? i : nv - 1, ti = ic * (ic + 1) / 2;
    const int kend = i0 + kGroupLanes < nv ? i0 + kGroupLanes : nv;
    float s = 0.f;
DOT_UNROLL
    for (int k = 0; k < kend; ++k) {
      const float l = Li[ti + (k <= ic ? k : ic)];
      s += (k <= ic ? l : 0.f) * b[k];
    }
    if (i < nv) tmp[i] = s;
  }
  SYNC();
  for (int i0 = 0; i0 < nv; i0 += kGroupLanes) {
    const int i = i0 + g, ic = i < nv ? i : nv - 1;
    float s = 0.f;
    int tk = i0 * (i0 + 1) / 2;  // row k of the triangle
DOT_UNROLL
    for (int k = i0; k < nv; ++k) {
      const float l = Li[tk + (k >= ic ? ic : k)];
      s += (k >= ic ? l : 0.f) * tmp[k];
      tk += k + 1;
    }
    if (i < nv) x[i] = s;
  }
  SYNC();
}

struct LsPoint { float alpha, cost, d0, d1; };

__device__ __forceinline__ LsPoint ls_make(float alpha, float q0, float q1, float q2) {
  LsPoint p;
  p.alpha = alpha;
  p.cost = alpha * alpha * q2 + alpha * q1 + q0;
  p.d0 = 2.f * alpha * q2 + q1;
  p.d1 = 2.f * q2 + (q2 == 0.f ? MJ_MINVAL : 0.f);
  return p;
}
// a dof's active joint-limit row as one int: +-(row + 1), the sign being the row's Jacobian entry (0: no active limit -> sign 0, row 0)
__device__ __forceinline__ float lim_sign(int dl) { return dl > 0 ? 1.f : dl < 0 ? -1.f : 0.f; }
__device__ __forceinline__ int lim_row(int dl) { return dl > 0 ? dl - 1 : dl < 0 ? -dl - 1 : 0; }
__device__ __forceinline__ bool in_bracket(const LsPoint& x, const LsPoint& y) {
  // bitwise on purpose: four compares and three mask operations, no short-circuit branches (each would be a save / restore of exec)
  return (bool)((int)((x.d0 < y.d0) & (y.d0 < 0.f)) | (int)((x.d0 > y.d0) & (y.d0 > 0.f)));
}
__device__ __forceinline__ LsPoint ls_sel(bool c, const LsPoint& a, const LsPoint& b) { return c ? a : b; }

// impedance / reference (MuJoCo solref + solimp -> k, b, imp), pos = signed distance - margin.  k and b depend on the solver
// parameters only (one pair for joint limits, one for contacts); the impedance depends on the row's violation.
__device__ __forceinline__ void kb_params(const float* solref, const float* solimp, float timestep, float& k, float& b) {
  const float timeconst = fmaxf(solref[0], 2.f * timestep);  // refsafe
  const float dampratio = solref[1];
  const float dmax = fminf(fmaxf(solimp[1], MJ_MINIMP), MJ_MAXIMP);
  k = 1.f / (dmax * dmax * timeconst * timeconst * dampratio * dampratio);
  b = 2.f / (dmax * timeconst);
  if (solref[0] <= 0.f) k = -solref[0] / (dmax * dmax);
  if (solref[1] <= 0.f) b = -solref[1] / dmax;
}
__device__ __forceinline__ float impedance(const float* solimp, float pos) {
  const float dmin = fminf(fmaxf(solimp[0], MJ_MINIMP), MJ_MAXIMP);
  const float dmax = fminf(fmaxf(solimp[1], MJ_MINIMP), MJ_MAXIMP);
  const float width = fmaxf(MJ_MINVAL, solimp[2]);
  const float mid = fminf(fmaxf(solimp[3], MJ_MINIMP), MJ_MAXIMP);
  const float power = fmaxf(1.f, solimp[4]);
  const float x = fabsf(pos) / width;
  float a, c;
  if (power == 2.f) {  // MuJoCo's default solimp power: no transcendental needed
    a = x * x / mid;
    c = 1.f - (1.f - x) * (1.f - x) / (1.f - mid);
  } else {
    a = (1.f / powf(mid, power - 1.f)) * powf(x, power);
    c = 1.f - (1.f / powf(1.f - mid, power - 1.f)) * powf(fabsf(1.f - x), power);
  }
  const float y = x < mid ? a : c;
  float imp = dmin + y * (dmax - dmin);
  imp = fminf(fmaxf(imp, dmin), dmax);
  if (x > 1.f) imp = dmax;
  return imp;
}

// What the kernel knows at compile time.  RuntimeModel: nothing - dims, table offsets and the LDS layout arrive as kernel
// arguments (about 200 scalar values; the kernel has ~100 SGPRs, so most of them are spilled to VGPR lanes and read back with
// v_readlane at every use).  StaticModel<dims...>: the dims, the blob's table offsets and the LDS layout are compile-time
// constants (the latter two are pure functions of the dims, model_view.h), which frees those registers and drops the code of
// absent features: 180 -> 148 us (offsets only) -> 131 us (dims too) per step on synth_stompy_pro, 1306 -> 1012 us on
// synth_stompy_full at 8192 envs; results are bit-identical to the run-time-sized kernel (tests/test_kernels_physics.py).
// Instantiations are listed in spec_dims.inc (generated by minppo_amd/build.py); other models run the RuntimeModel kernel.
struct RuntimeModel {
  static constexpr bool kStatic = false;
  static constexpr BlobDims dims() { return BlobDims{}; }
};
template <int NQ, int NV, int NU, int NBODY, int NJNT, int NCON, int NLIMIT, int NPAIR, int NLEVEL, int NROOT, int NCVX = 0, int NCVXVERT = 0, int HULL = 0, int NCYL = 0>
struct StaticModel {
  static constexpr bool kStatic = true;
  static constexpr BlobDims dims() { return BlobDims{NQ, NV, NU, NBODY, NJNT, NCON, NLIMIT, NPAIR, NLEVEL, NROOT, NCVX, NCVXVERT, HULL, NCYL}; }
};
// MODE (EnvArgs::mode) is a template parameter too: the step kernel carries neither the probe's 17 output pointers nor its stores.
template <class SD, int MODE>
__global__ void __launch_bounds__(64 * kMaxWavesPerBlock) env_kernel(ModelView mv, EnvArgs a, PhysLds Prt) {
  constexpr BlobDims kSD = SD::dims();
  constexpr BlobOffsets kSO = blob_offsets(kSD);
  constexpr bool kDims = SD::kStatic;
  // fixed-size kernel, up to 32 dofs: a lane's rows / columns of the Cholesky factors live in registers (see factor_m below)
  constexpr bool kRegChol = kDims && kSD.nv <= kRegCholMaxNv;
  constexpr int kSpill = kDims ? spill_for(kSD.nq, kSD.nv, kSD.nu, kSD.nbody, kSD.njnt, kSD.ncon, kSD.nlimit + 4 * kSD.ncon, kSD.nroot, kSD.ncvx, kRegChol, kSO.words) : 0;
  constexpr PhysLds kSP = make_phys_lds(kSD.nq, kSD.nv, kSD.nu, kSD.nbody, kSD.njnt, kSD.ncon, kSD.nlimit + 4 * kSD.ncon, kSD.nroot, kSD.ncvx, kRegChol, kSpill);
  const PhysLds P = SD::kStatic ? kSP : Prt;
  constexpr int NV = kDims ? kSD.nv : 0;
  constexpr int kDotU = dot_unroll(NV);
  MPPO_DYN_SMEM(smem_raw);
  const int tid = threadIdx.x;
  const int g = tid & (kGroupLanes - 1);
  const int row = (tid & 63) / kGroupLanes;                 // DPP row inside the wave
  // environments per wave: four; a run-time-sized kernel carries two or one when the robot's working set is too large for four (mppo_model_open)
  const int epw = kDims ? kEnvsPerWave : mv.epw;
  const int el = (tid >> 6) * epw + row;          // environment inside the workgroup
  int env = blockIdx.x * ((int)(blockDim.x >> 6) * epw) + el;  // the launch chooses the waves per workgroup (launch_env)
  const bool valid = env < a.N;
  if (!valid) env = a.N - 1;  // surplus groups shadow the last environment and never store
  // model tables: one coalesced copy of the blob into LDS per workgroup, then every table read is a ds_read
  int* tabI = reinterpret_cast<int*>(smem_raw);
  const float* tabF = reinterpret_cast<const float*>(smem_raw);
  {
    const float4* src = reinterpret_cast<const float4*>(mv.blob);
    float4* dst = reinterpret_cast<float4*>(smem_raw);
    for (int i = tid; i < mv.blob_words / 4; i += (int)blockDim.x) dst[i] = src[i];
  }
  __syncthreads();  // the only workgroup-wide barrier: model tables are shared by the waves of the block
  if (row >= epw) return;  // rows without an environment
  float* S = reinterpret_cast<float*>(smem_raw) + mv.blob_words + (size_t)el * P.total;
  // matrices in global memory (large robots; model_view.h PhysLds::spill): this group's record - one per group of the GRID, so that a
  // surplus group (which shadows the last environment) has a record of its own
  const bool spJ = (P.spill & kSpillJ) != 0, spM = (P.spill & kSpillM) != 0;
  float* G = P.gwords > 0 ? a.scratch + (size_t)(blockIdx.x * ((int)(blockDim.x >> 6) * epw) + el) * P.gwords : nullptr;

  const int nq = kDims ? kSD.nq : mv.nq, nv = kDims ? kSD.nv : mv.nv, nu = kDims ? kSD.nu : mv.nu, nb = kDims ? kSD.nbody : mv.nbody, njnt = kDims ? kSD.njnt : mv.njnt,
            ncon = kDims ? kSD.ncon : mv.ncon, nlim = kDims ? kSD.nlimit : mv.nlimit, nefc = kDims ? kSD.nlimit + 4 * kSD.ncon : mv.nefc;
  const int nlevel = kDims ? kSD.nlevel : mv.nlevel, nroot = kDims ? kSD.nroot : mv.nroot;
  const int ldm = P.ldm, ldj = P.ldj;
  const int nvq = (nv + 3) >> 2;  // dof quads (the global-memory matrices hold four consecutive dofs per 16-byte word)
  const float h = mv.timestep;
  const int O = mv.obs_dim, OP = mv.obs_pad;
  float* qpos = S + P.qpos; float* qvel = (float*)__builtin_assume_aligned(S + P.qvel, 16); float* ctrl = S + P.ctrl; float* warm = S + P.warm;
  float* xpos = S + P.xpos; float* xquat = S + P.xquat; float* xipos = S + P.xipos; float* rootcom = S + P.rootcom;  // (poses: region A1)
  float* cinert = S + P.cinert; float* cdof = S + P.cdof; float* cvel = S + P.cvel;
  int* dlim = reinterpret_cast<int*>(S + P.dlim);
  float* M = (float*)__builtin_assume_aligned(S + P.M, 16); float* LL = (float*)__builtin_assume_aligned(S + P.LL, 16);
  float* qfs = (float*)__builtin_assume_aligned(S + P.qfs, 16); float* qas = (float*)__builtin_assume_aligned(S + P.qas, 16); float* qacc = (float*)__builtin_assume_aligned(S + P.qacc, 16); float* Ma = (float*)__builtin_assume_aligned(S + P.Ma, 16);
  float* grad = (float*)__builtin_assume_aligned(S + P.grad, 16); float* Mgrad = (float*)__builtin_assume_aligned(S + P.Mgrad, 16); float* search = (float*)__builtin_assume_aligned(S + P.search, 16); float* mvv = (float*)__builtin_assume_aligned(S + P.mv, 16); float* qfc = (float*)__builtin_assume_aligned(S + P.qfc, 16);
  float* t0 = (float*)__builtin_assume_aligned(S + P.t0, 16); float* t1 = (float*)__builtin_assume_aligned(S + P.t1, 16);
  float* eD = S + P.D; float* earef = S + P.aref; float* jaref = S + P.jaref; float* jv = S + P.jv; float* force = S + P.force;
  float* conpos = S + P.conpos; float* condist = S + P.condist; float* confr = S + P.confr;
  const int npair = kDims ? kSD.npair : mv.npair, nplane = ncon - npair, ncvx = kDims ? kSD.ncvx : mv.ncvx;
  // (features that select code: constants in a model-specialised kernel - a robot without hull pairs / cylinders carries none of it)
  const bool has_hull = kDims ? kSD.hull != 0 : mv.hull_words > 0, has_cyl = kDims ? kSD.ncyl > 0 : mv.ncyl > 0;
  float* cvxsel = S + P.cvxsel; float* cvxok = S + P.cvxok;
  float* xanchor = S + P.xanchor; float* xaxis = S + P.xaxis;
  float* Cw = S + P.C; float* cdofdot = S + P.cdofdot; float* cfrc = S + P.cfrc; float* J = (float*)__builtin_assume_aligned(S + P.J, 16);

  // M in global memory: Mq[k / 4][i][k % 4] - lane i reads four consecutive columns of its row as one 16-byte word, the lanes of an
  // environment consecutive words.  (Entries k >= nv of the last quad are zeros.)
  float4* Mq = reinterpret_cast<float4*>(G + (spM ? P.gM : 0));
  auto Mset = [&](int i, int k, float v) {
    if (spM) reinterpret_cast<float*>(Mq)[((k >> 2) * nv + i) * 4 + (k & 3)] = v; else M[i * ldm + k] = v;
  };
  auto Mget = [&](int i, int k) -> float {
    return spM ? reinterpret_cast<const float*>(Mq)[((k >> 2) * nv + i) * 4 + (k & 3)] : M[i * ldm + k];
  };
  const float* rec = a.state ? a.state + (size_t)env * mv.rec_dim : nullptr;

  PT(0);
  // ---- P0: load the state ------------------------------------------------------------------
  if (MODE == 1) {
    FOR_G(i, nq) qpos[i] = rec[i];
    FOR_G(i, nv) { qvel[i] = rec[nq + i]; warm[i] = rec[OP + i]; }
  } else if (MODE == 0) {
    FOR_G(i, nq) qpos[i] = TF(qpos0)[i];
    FOR_G(i, nv) { qvel[i] = 0.f; warm[i] = 0.f; }
  } else {
    FOR_G(i, nq) qpos[i] = a.p_qpos[(size_t)env * nq + i];
    FOR_G(i, nv) { qvel[i] = a.p_qvel[(size_t)env * nv + i]; warm[i] = a.p_warm[(size_t)env * nv + i]; }
  }
  // pre-step quantities the reward needs (env.py:212-217, 222)
  float pre_p0 = 0.f, pre_z = 0.f, pre_comx = 0.f, time_in = 0.f;
  if (MODE == 1) {
    float s = 0.f;
    FOR_G(i, nq) { const float d = TF(qpos0)[i] - rec[i]; s += d * d; }
    pre_p0 = sqrtf(group16_sum(s));
    pre_z = rec[2];
    pre_comx = rec[OP + nv];
    time_in = rec[OP + nv + 1];
  }
  // observation = the PRE-step record (env.py:163, quirk C-5): copied out before anything of the new record is written (cinert and
  // cvel of the new state go to the record in the middle of the step); the reset observation replaces it at the end when done
  if (MODE == 1) {
    constexpr int kChunk = 32;  // (one memory round trip for observations of up to 512 words)
    for (int i0 = g; i0 < OP; i0 += kGroupLanes * kChunk) {
      float old[kChunk];
      _Pragma("unroll") for (int u = 0; u < kChunk; ++u) { const int i = i0 + kGroupLanes * u; old[u] = rec[i < OP ? i : OP - 1]; }
      _Pragma("unroll") for (int u = 0; u < kChunk; ++u) { const int i = i0 + kGroupLanes * u; if (valid && i < OP) a.obs[(size_t)env * a.obs_ld + i] = old[u]; }
    }
  }
  float* recw = a.state ? a.state + (size_t)env * mv.rec_dim : nullptr;
  const int o_ci = nq + nv, o_cv = o_ci + (mv.include_c ? 10 * (nb - 1) : 0), o_qa = o_cv + (mv.include_c ? 6 * (nb - 1) : 0);  // the record's fields
  // NaN anywhere in the stepped state (env.py:173-176), accumulated where the values are at hand.  What is found in the middle of the step (qfrc_actuator,
  // cinert, cvel) waits for the end of the step as the wave's BALLOT - one scalar value for the four environments - not as a per-lane register: a per-lane
  // flag that only an any-reduction reads is live in all sixteen lanes across the second factorisation, and in a kernel that spills (three matrix rows per
  // lane, 93 bodies: 250 spilled registers) the compiler saved and restored it inside divergent code - the lanes inactive there came back with garbage, every
  // episode "ended" at its first step (round 6; only on hardware, only in that instantiation; found by mppo_model_open's comparison with the run-time-sized
  // kernel).  A scalar is saved whole whatever lanes are active.  Only the kernels that spill vector registers carry the flag that way - the model-specialised ones
  // beyond 32 dofs with their factors in registers (every other instantiation: 0 spilled vector registers, -Rpass-analysis=kernel-resource-usage): two more live
  // scalars cost the BASELINE kernels, which are short of scalar registers, 1.3 us per step (measured), and a per-lane register that is never spilled is safe.
  constexpr bool kFlagAsBallot = kRegChol && kDims && kSD.nv > 2 * kGroupLanes;
  group16_flags_t bad_mid = group16_flags(false);
  int bad_lane = 0;

  float new_comx = 0.f;
  const int frames = MODE == 1 ? a.n_frames : 1;
  for (int frame = 0; frame < frames; ++frame) {
    PT(1);
    if (g == 0) { st3(xpos, {0.f, 0.f, 0.f}); st4(xquat, {1.f, 0.f, 0.f, 0.f}); }  // the world body (region A1 is rewritten by every frame)
    // the controls: they share the storage of a solver vector (t1, first written in the CG loop), so every frame reads them again
    if (MODE == 1) { FOR_G(i, nu) ctrl[i] = a.action[(size_t)env * a.act_ld + i]; }
    else if (MODE == 0) { FOR_G(i, nu) ctrl[i] = 0.f; }
    else { FOR_G(i, nu) ctrl[i] = a.p_ctrl[(size_t)env * nu + i]; }
    SYNC();  // the state loaded above (or integrated by the previous frame) is visible to every lane
    // ================= fwd_position: kinematics ================================================================
    // The tree is deep and narrow (a level holds one or two bodies of a humanoid), so the level-synchronous sweep runs on one
    // or two lanes.  Only what truly depends on the parent is done there: pose = parent pose o local pose.  Everything else -
    // the joint rotations (sincos), local anchors / axes, rotation matrices, inertial frames - is done for all bodies at once
    // before and after the sweep.  (Same transforms as mj_kinematics / mjx.kinematics, regrouped by associativity.)
    // A: local pose of every body in its parent's frame; local joint anchors / axes
    FOR_G(b, nb) {
      if (b > 0) {
        V3 pl = ld3(TF(body_pos) + 3 * b);
        Q4 ql = ld4(TF(body_quat) + 4 * b);
        const int j0 = TI(body_jntadr)[b], j1 = j0 + TI(body_jntnum)[b];
        for (int j = j0; j < j1; ++j) {
          const int qa = TI(jnt_qposadr)[j], jt = TI(jnt_type)[j];
          if (jt == JNT_FREE) {  // only on children of the world: local = world
            pl = ld3(qpos + qa);
            ql = qnormalize(ld4(qpos + qa + 3));
            st3(xanchor + 3 * j, pl);
            st3(xaxis + 3 * j, qrot(ql, ld3(TF(jnt_axis) + 3 * j)));
          } else {
            const V3 anchor = add3(pl, qrot(ql, ld3(TF(jnt_pos) + 3 * j)));
            const V3 axis = qrot(ql, ld3(TF(jnt_axis) + 3 * j));
            st3(xanchor + 3 * j, anchor);
            st3(xaxis + 3 * j, axis);
            const float disp = qpos[qa] - TF(qpos0)[qa];
            if (jt == JNT_HINGE) {
              ql = qmul(ql, axis_angle(ld3(TF(jnt_axis) + 3 * j), disp));
              pl = sub3(anchor, qrot(ql, ld3(TF(jnt_pos) + 3 * j)));
            } else {
              pl = add3(pl, mul3(axis, disp));
            }
          }
        }
        st3(xpos + 3 * b, pl);
        st4(xquat + 4 * b, ql);
      }
    }
    SYNC();
    // B: the sweep, in place (a level reads its parents' finished poses and its own local ones)
    for (int lv = 0; lv < nlevel; ++lv) {
      const int adr = TI(level_adr)[lv], cnt = TI(level_adr)[lv + 1] - adr;
      FOR_G(ii, cnt) {
        const int b = TI(level_body)[adr + ii];
        const int p = TI(body_parent)[b];
        const Q4 pq = ld4(xquat + 4 * p);
        st3(xpos + 3 * b, add3(ld3(xpos + 3 * p), qrot(pq, ld3(xpos + 3 * b))));
        st4(xquat + 4 * b, qnormalize(qmul(pq, ld4(xquat + 4 * b))));
      }
      SYNC();
    }
    // C: joint anchors / axes into the world frame (through the parent's pose), inertial frames' positions (the rotation matrices
    // of bodies and inertial frames are formed from the quaternions where they are used: cinert and the free joint's cdof below)
    FOR_G(j, njnt) {
      const int p = TI(body_parent)[TI(jnt_bodyid)[j]];
      const Q4 pq = ld4(xquat + 4 * p);
      st3(xanchor + 3 * j, add3(ld3(xpos + 3 * p), qrot(pq, ld3(xanchor + 3 * j))));
      st3(xaxis + 3 * j, qrot(pq, ld3(xaxis + 3 * j)));
    }
    FOR_G(b, nb) {
      if (b > 0) {
        const Q4 quat = ld4(xquat + 4 * b);
        st3(xipos + 3 * b, add3(ld3(xpos + 3 * b), qrot(quat, ld3(TF(body_ipos) + 3 * b))));
      }
    }
    SYNC();
    if (MODE == 2 && valid && a.probe.xpos) FOR_G(i, nb * 3) a.probe.xpos[(size_t)env * nb * 3 + i] = xpos[i];  // (the poses are gone by the end of the step)
    PT(2);
    // ---- com_pos: centre of mass of every kinematic tree; contact candidates --------------------
    for (int r = 0; r < nroot; ++r) {
      const u64 mask = TU(body_subtree_mask)[TI(root_body)[r]], mask_hi = nb > 64 ? TU(body_subtree_mask)[nb + TI(root_body)[r]] : 0ull;
      float sx = 0.f, sy = 0.f, sz = 0.f, sm = 0.f;
      FOR_G(b, nb) {
        if (((b < 64 ? mask : mask_hi) >> (b & 63)) & 1ull) {
          const float m = TF(body_mass)[b];
          sx += m * xipos[3 * b]; sy += m * xipos[3 * b + 1]; sz += m * xipos[3 * b + 2]; sm += m;
        }
      }
      sx = group16_sum(sx); sy = group16_sum(sy); sz = group16_sum(sz); sm = group16_sum(sm);
      const float inv = 1.f / fmaxf(sm, MJ_MINVAL);
      if (g == 0) st3(rootcom + 3 * r, {sx * inv, sy * inv, sz * inv});
      if (r == 0) new_comx = sx * inv;  // subtree_com[1].x: body 1 is the first root (env.py:222-223)
    }
    if (ncvx > 0) {
      // convex (mesh) geoms against the plane, MJX collision_convex.plane_convex: among the hull vertices within 1 mm of the deepest
      // penetrating one, _manifold_points picks four that span the contact patch (a: the first candidate, b: the farthest from a,
      // c: the farthest from the line ab, d: the farthest from the edges ac / bc; ties go to the lower index as argmax does);
      // a vertex picked twice fills only its first slot.  One lane per geom scans its vertices five times.
      FOR_G(k, ncvx) {
        const int b = TI(cvx_body)[k], v0 = TI(cvx_vadr)[k], v1 = TI(cvx_vadr)[k + 1];
        float R[9];
        qmat(ld4(xquat + 4 * b), R);
        const V3 nl = {R[6], R[7], R[8]};                 // the plane's normal (+z) in the body frame
        const float h0 = mv.plane_z - xpos[3 * b + 2];    // support(v) = h0 - nl . vert(v): depth below the plane
        const float* vt = TF(cvx_vert);
        float smax = -INFINITY;
        for (int v = v0; v < v1; ++v) smax = fmaxf(smax, h0 - dot3(nl, ld3(vt + 3 * v)));
        const float thr = fmaxf(0.f, smax - 1e-3f);
        auto dm = [&](int v) { return (h0 - dot3(nl, ld3(vt + 3 * v)) > thr) ? 0.f : -1e6f; };
        int ia = v0;
        { float best = -INFINITY; for (int v = v0; v < v1; ++v) { const float x = dm(v); if (x > best) { best = x; ia = v; } } }
        const V3 A = ld3(vt + 3 * ia);
        int ib = v0;
        { float best = -INFINITY; for (int v = v0; v < v1; ++v) { const V3 e = sub3(A, ld3(vt + 3 * v)); const float x = dot3(e, e) + dm(v); if (x > best) { best = x; ib = v; } } }
        const V3 B = ld3(vt + 3 * ib);
        const V3 ab = cross3(nl, sub3(A, B));
        int ic = v0;
        { float best = -INFINITY; for (int v = v0; v < v1; ++v) { const float x = fabsf(dot3(sub3(A, ld3(vt + 3 * v)), ab)) + dm(v); if (x > best) { best = x; ic = v; } } }
        const V3 Cc = ld3(vt + 3 * ic);
        const V3 ac = cross3(nl, sub3(A, Cc)), bc = cross3(nl, sub3(B, Cc));
        int id = v0;
        {
          float best = -INFINITY;
          for (int v = v0; v < v1; ++v) { const float x = fabsf(dot3(sub3(B, ld3(vt + 3 * v)), bc)) + dm(v); if (x > best) { best = x; id = v; } }
          for (int v = v0; v < v1; ++v) { const float x = fabsf(dot3(sub3(A, ld3(vt + 3 * v)), ac)) + dm(v); if (x > best) { best = x; id = v; } }
        }
        const int idx[4] = {ia, ib, ic, id};
        for (int j = 0; j < 4; ++j) {
          bool first = true;
          for (int i = 0; i < j; ++i) first = first && idx[i] != idx[j];
          st3(cvxsel + 3 * (4 * k + j), ld3(vt + 3 * idx[j]));
          cvxok[4 * k + j] = first ? 1.f : 0.f;
        }
      }
      SYNC();
    }
    FOR_G(c, nplane) {  // ground contacts (MJX plane_sphere / plane_capsule; a box's or a mesh's chosen hull vertices: plane_convex)
      const int b = TI(con_bodyid)[c];
      const Q4 q = ld4(xquat + 4 * b);
      if (has_cyl && TI(con_cvx)[c] <= -2) {
        // a cylinder against the plane, MJX collision_primitive.plane_cylinder: three slots, placed by the lane that owns the first - the
        // point of the lower rim nearest to the plane and two more of that rim 120 degrees to either side; lying on its side
        // (|axis . n| half < 1e-3): the nearest point of the other rim in slot 1
        if (TI(con_cvx)[c] != -2) continue;
        const V3 ctr = add3(ld3(xpos + 3 * b), qrot(q, ld3(TF(con_lpos) + 3 * c)));
        const V3 hv = ld3(TF(con_axis) + 3 * c);
        const float half = sqrtf(dot3(hv, hv)), r = TF(con_radius)[c], height = ctr.z - mv.plane_z;
        V3 axis = qrot(q, mul3(hv, 1.f / half));
        const V3 xaxis = qrot(q, ld3(TF(con_axis) + 3 * (c + 1)));
        float prjaxis = axis.z;
        const float sign = prjaxis < 0.f ? 1.f : -1.f;  // the axis turned towards the plane
        axis = mul3(axis, sign); prjaxis *= sign;
        V3 vec = sub3(mul3(axis, prjaxis), V3{0.f, 0.f, 1.f});
        const float len = sqrtf(dot3(vec, vec));
        vec = len < 1e-12f ? mul3(xaxis, r) : mul3(vec, r / len);
        const float prjvec = vec.z;
        axis = mul3(axis, half); prjaxis *= half;
        float nrm;
        const V3 vec1 = mul3(normalize_norm(cross3(vec, axis), nrm), r * 0.8660254037844386f);
        const float d1 = height + prjaxis + prjvec, d2 = height + prjaxis - 0.5f * prjvec, d3 = height - prjaxis + prjvec;
        const bool side = fabsf(prjaxis) < 1e-3f;
        const V3 hvv = mul3(vec, 0.5f);
        const V3 r0 = add3(axis, vec), r1 = side ? sub3(vec, axis) : sub3(add3(axis, vec1), hvv), r2 = sub3(sub3(axis, vec1), hvv);
        const float dd1 = side ? d3 : d2;
        condist[c] = d1; condist[c + 1] = dd1; condist[c + 2] = d2;
        st3(conpos + 3 * c, {ctr.x + r0.x, ctr.y + r0.y, ctr.z + r0.z - 0.5f * d1});
        st3(conpos + 3 * (c + 1), {ctr.x + r1.x, ctr.y + r1.y, ctr.z + r1.z - 0.5f * dd1});
        st3(conpos + 3 * (c + 2), {ctr.x + r2.x, ctr.y + r2.y, ctr.z + r2.z - 0.5f * d2});
        for (int j = 0; j < 3; ++j) { st3(confr + 6 * (c + j), {0.f, 0.f, 1.f}); st3(confr + 6 * (c + j) + 3, {0.f, 1.f, 0.f}); }
        continue;
      }
      const int cs = ncvx > 0 ? TI(con_cvx)[c] : -1;
      const V3 centre = add3(ld3(xpos + 3 * b), qrot(q, cs >= 0 ? ld3(cvxsel + 3 * cs) : ld3(TF(con_lpos) + 3 * c)));
      const float rad = TF(con_radius)[c];
      float dist = centre.z - mv.plane_z - rad;
      if (cs >= 0 && cvxok[cs] == 0.f) dist = 1.f;  // a duplicate of an earlier slot: switched off as MJX does (dist = 1)
      condist[c] = dist;
      st3(conpos + 3 * c, {centre.x, centre.y, centre.z - (rad + 0.5f * dist)});
      // frame: normal +z; the first tangent follows the capsule axis projected on the plane unless that projection is shorter
      // than 0.5 (spheres and corners carry a zero axis) - then make_frame's +y
      const V3 ax = qrot(q, ld3(TF(con_axis) + 3 * c));
      const float bn = sqrtf(ax.x * ax.x + ax.y * ax.y);
      st3(confr + 6 * c, {0.f, 0.f, 1.f});
      st3(confr + 6 * c + 3, bn < 0.5f ? V3{0.f, 1.f, 0.f} : V3{ax.x / bn, ax.y / bn, 0.f});
    }
    FOR_G(k, npair) {  // geom-geom pairs (MJX sphere_sphere / sphere_capsule / capsule_capsule): one contact each
      const int c = nplane + k;
      const int b1 = TI(pair_body)[2 * k], b2 = TI(pair_body)[2 * k + 1];
      const float* gp = TF(pair_geom) + 16 * k;
      if (has_hull && gp[7] != 0.f) continue;  // geom 2 is a convex hull: below
      const Q4 q1 = ld4(xquat + 4 * b1), q2 = ld4(xquat + 4 * b2);
      const V3 c1 = add3(ld3(xpos + 3 * b1), qrot(q1, ld3(gp))), h1 = qrot(q1, ld3(gp + 3));
      const V3 c2 = add3(ld3(xpos + 3 * b2), qrot(q2, ld3(gp + 8))), h2 = qrot(q2, ld3(gp + 11));
      V3 p1, p2;
      closest_segment_points(sub3(c1, h1), add3(c1, h1), sub3(c2, h2), add3(c2, h2), p1, p2);
      float dist;
      V3 n = normalize_norm(sub3(p2, p1), dist);
      if (dist == 0.f) n = {1.f, 0.f, 0.f};
      dist -= gp[6] + gp[14];
      condist[c] = dist;
      st3(conpos + 3 * c, add3(p1, mul3(n, gp[6] + 0.5f * dist)));
      st3(confr + 6 * c, n);
      st3(confr + 6 * c + 3, frame_tangent(n));
    }
    if (has_hull) {
      // sphere / capsule against a box or a mesh hull of another body (pair rows tagged with a hull; run-time-sized kernel only): one
      // pair at a time, the environment's 16 lanes together.  Geometry in the frame of the hull's body, results back in the world.
      const int* hs = mv.blob + mv.blob_words;
      const HullView hv = hull_view(hs[0], hs[1], hs[2], hs[3], hs[4], hs[5]);
      const float* hf = reinterpret_cast<const float*>(hs);
      const HullTabs T{hs + hv.face_adr, hs + hv.fidx, hs + hv.edge, hf + hv.vert, hf + hv.fnormal, hf + hv.enormal, hs + hv.udadr, hf + hv.udir};
      for (int k = 0; k < npair; ++k) {
        const float* gp = TF(pair_geom) + 16 * k;
        const int hid = (int)gp[7] - 1;
        if (hid < 0 || gp[15] != 0.f) continue;  // (the later slots of a pair with several contacts are filled with its first)
        const int c = nplane + k;
        const int b1 = TI(pair_body)[2 * k], b2 = TI(pair_body)[2 * k + 1];
        const Q4 q1 = ld4(xquat + 4 * b1);
        float R[9];
        qmat(ld4(xquat + 4 * b2), R);
        const V3 x2 = ld3(xpos + 3 * b2);
        auto to_hull = [&](V3 v) { return V3{R[0] * v.x + R[3] * v.y + R[6] * v.z, R[1] * v.x + R[4] * v.y + R[7] * v.z, R[2] * v.x + R[5] * v.y + R[8] * v.z}; };
        auto to_world = [&](V3 v) { return V3{R[0] * v.x + R[1] * v.y + R[2] * v.z, R[3] * v.x + R[4] * v.y + R[5] * v.z, R[6] * v.x + R[7] * v.y + R[8] * v.z}; };
        if (gp[14] != 0.f && gp[3] == 0.f && gp[4] == 0.f && gp[5] == 0.f && gp[6] == 0.f) {
          // geom 1 is a hull too: box / mesh against box / mesh (MJX convex_convex), four slots
          const int hid1 = (int)gp[14] - 1;
          float R1[9];
          qmat(q1, R1);
          RigidT X;  // body 1's frame -> body 2's: R2^T R1, R2^T (x1 - x2)
          _Pragma("unroll") for (int i = 0; i < 3; ++i) _Pragma("unroll") for (int j = 0; j < 3; ++j) X.R[3 * i + j] = R[i] * R1[j] + R[3 + i] * R1[3 + j] + R[6 + i] * R1[6 + j];
          X.t = to_hull(sub3(ld3(xpos + 3 * b1), x2));
          float d4[4];
          V3 p4[4], n4;
          hull_hull_contacts(T, X, hs[hv.fadr + hid1], hs[hv.fadr + hid1 + 1], hs[hv.fadr + hid], hs[hv.fadr + hid + 1], hs[hv.vadr + hid1], hs[hv.vadr + hid1 + 1],
                             hs[hv.vadr + hid], hs[hv.vadr + hid + 1], T.udadr[hid1], T.udadr[hid1 + 1], T.udadr[hid], T.udadr[hid + 1], g, d4, p4, n4);
          V3 q0 = p4[0], q1_ = p4[1], q2 = p4[2], q3 = p4[3];
          float e0 = d4[0], e1 = d4[1], e2 = d4[2], e3 = d4[3];
          MPPO_REG_PIN(q0.x); MPPO_REG_PIN(q0.y); MPPO_REG_PIN(q0.z); MPPO_REG_PIN(q1_.x); MPPO_REG_PIN(q1_.y); MPPO_REG_PIN(q1_.z);
          MPPO_REG_PIN(q2.x); MPPO_REG_PIN(q2.y); MPPO_REG_PIN(q2.z); MPPO_REG_PIN(q3.x); MPPO_REG_PIN(q3.y); MPPO_REG_PIN(q3.z);
          MPPO_REG_PIN(e0); MPPO_REG_PIN(e1); MPPO_REG_PIN(e2); MPPO_REG_PIN(e3);
          if (g < 4) {
            const V3 nw = to_world(n4);
            const V3 pw = g == 0 ? q0 : g == 1 ? q1_ : g == 2 ? q2 : q3;
            condist[c + g] = g == 0 ? e0 : g == 1 ? e1 : g == 2 ? e2 : e3;
            st3(conpos + 3 * (c + g), add3(x2, to_world(pw)));
            st3(confr + 6 * (c + g), nw);
            st3(confr + 6 * (c + g) + 3, frame_tangent(nw));
          }
          continue;
        }
        const bool capsule = gp[3] != 0.f || gp[4] != 0.f || gp[5] != 0.f;
        const V3 cp = to_hull(sub3(add3(ld3(xpos + 3 * b1), qrot(q1, ld3(gp))), x2)), half = to_hull(qrot(q1, ld3(gp + 3)));
        float dist2[2];
        V3 pos2[2], nrm2[2];
        hull_pair_contacts(T, hs[hv.fadr + hid], hs[hv.fadr + hid + 1], hs[hv.eadr + hid], hs[hv.eadr + hid + 1], capsule, cp, half, gp[6], g, dist2, pos2, nrm2);
        if (g < (capsule ? 2 : 1)) {  // (selected, not indexed: a lane-indexed register array would live in scratch)
          const V3 nw = to_world(g == 0 ? nrm2[0] : nrm2[1]);
          condist[c + g] = g == 0 ? dist2[0] : dist2[1];
          st3(conpos + 3 * (c + g), add3(x2, to_world(g == 0 ? pos2[0] : pos2[1])));
          st3(confr + 6 * (c + g), nw);
          st3(confr + 6 * (c + g) + 3, frame_tangent(nw));
        }
      }
    }
    SYNC();
    PT(3);
    // ---- cinert (per body), cdof (per joint) -------------------------------------------------------
    FOR_G(b, nb) {
      float* ci = cinert + 10 * b;
      if (b == 0) {
        for (int k = 0; k < 10; ++k) ci[k] = 0.f;
      } else {
        int ri = 0;
        for (int r = 0; r < nroot; ++r) if (TI(root_body)[r] == TI(body_rootid)[b]) ri = r;
        const V3 off = sub3(ld3(xipos + 3 * b), ld3(rootcom + 3 * ri));
        const float m = TF(body_mass)[b];
        float R[9];  // ximat
        qmat(qmul(ld4(xquat + 4 * b), ld4(TF(body_iquat) + 4 * b)), R);
        const float d0 = TF(body_inertia)[3 * b], d1 = TF(body_inertia)[3 * b + 1], d2 = TF(body_inertia)[3 * b + 2];
        const float oo = dot3(off, off);
        ci[0] = R[0] * R[0] * d0 + R[1] * R[1] * d1 + R[2] * R[2] * d2 + m * (oo - off.x * off.x);
        ci[1] = R[3] * R[3] * d0 + R[4] * R[4] * d1 + R[5] * R[5] * d2 + m * (oo - off.y * off.y);
        ci[2] = R[6] * R[6] * d0 + R[7] * R[7] * d1 + R[8] * R[8] * d2 + m * (oo - off.z * off.z);
        ci[3] = R[0] * R[3] * d0 + R[1] * R[4] * d1 + R[2] * R[5] * d2 - m * off.x * off.y;
        ci[4] = R[0] * R[6] * d0 + R[1] * R[7] * d1 + R[2] * R[8] * d2 - m * off.x * off.z;
        ci[5] = R[3] * R[6] * d0 + R[4] * R[7] * d1 + R[5] * R[8] * d2 - m * off.y * off.z;
        ci[6] = m * off.x; ci[7] = m * off.y; ci[8] = m * off.z; ci[9] = m;
      }
    }
    FOR_G(j, njnt) {
      const int b = TI(jnt_bodyid)[j], da = TI(jnt_dofadr)[j], jt = TI(jnt_type)[j];
      int ri = 0;
      for (int r = 0; r < nroot; ++r) if (TI(root_body)[r] == TI(body_rootid)[b]) ri = r;
      const V3 off = sub3(ld3(rootcom + 3 * ri), ld3(xanchor + 3 * j));
      if (jt == JNT_FREE) {
        float R[9];  // xmat
        qmat(ld4(xquat + 4 * b), R);
        _Pragma("unroll") for (int k = 0; k < 3; ++k) {
          float* c = cdof + 6 * (da + k);
          c[0] = c[1] = c[2] = 0.f;
          c[3] = k == 0 ? 1.f : 0.f; c[4] = k == 1 ? 1.f : 0.f; c[5] = k == 2 ? 1.f : 0.f;
          const V3 ax = {R[k], R[3 + k], R[6 + k]};
          float* cr = cdof + 6 * (da + 3 + k);
          st3(cr, ax);
          st3(cr + 3, cross3(ax, off));
        }
      } else if (jt == JNT_HINGE) {
        const V3 ax = ld3(xaxis + 3 * j);
        st3(cdof + 6 * da, ax);
        st3(cdof + 6 * da + 3, cross3(ax, off));
      } else {
        st3(cdof + 6 * da, {0.f, 0.f, 0.f});
        st3(cdof + 6 * da + 3, ld3(xaxis + 3 * j));
      }
    }
    if (spM) { for (int w = g; w < nvq * nv; w += kGroupLanes) Mq[w] = make_float4(0.f, 0.f, 0.f, 0.f); }
    else FOR_G(i, nv) for (int k = 0; k < nv; ++k) M[i * ldm + k] = 0.f;
    if (spM) GSYNC(); else SYNC();
    PT(4);
    // ---- crb: composite inertia over the subtree mask, dense M ------------------------------------
    FOR_G(i, nv) {
      const int bi = TI(dof_bodyid)[i];
      float crb[10];
      for (int k = 0; k < 10; ++k) crb[k] = 0.f;
      for (int w = 0; w < (nb > 64 ? 2 : 1); ++w) {  // (a second word of the subtree set beyond 64 bodies: run-time-sized kernel only)
        u64 mask = TU(body_subtree_mask)[w * nb + bi];
        while (mask) {
          const int c = 64 * w + __ffsll((long long)mask) - 1;
          mask &= mask - 1;
          for (int k = 0; k < 10; ++k) crb[k] += cinert[10 * c + k];
        }
      }
      float buf[6];
      inert_mul(crb, cdof + 6 * i, buf);
      int j = i;
      while (j >= 0) {
        const float* cj = cdof + 6 * j;
        float v = cj[0] * buf[0] + cj[1] * buf[1] + cj[2] * buf[2] + cj[3] * buf[3] + cj[4] * buf[4] + cj[5] * buf[5];
        if (j == i) v += TF(dof_armature)[i];
        Mset(i, j, v);
        Mset(j, i, v);
        j = TI(dof_parentid)[j];
      }
    }
    if (spM) GSYNC(); else SYNC();
    PT(5);
    // ---- factor_m: Cholesky of M (now) and of M + h*diag(damping) (implicit joint damping: at the END of the step, Euler), and the
    // inverse factors.  The step is wrapped in a two-trip loop around ONE copy of the factorisation code: trip 0 factors M and runs
    // the step's forward dynamics and solver, trip 1 factors M + h D and leaves the loop for the integrator.
    // Fixed-size kernel, up to 32 dofs: a lane's matrix rows (g and g + 16) in registers - a right-looking Cholesky that keeps L_kk^2
    // on the diagonal, one column exchange through LDS per elimination step, the inverse factor by forward substitution, a lane's
    // columns g and g + 16 - and the inverse factor STAYS in registers: lane i holds row i (for L^-1 b) and column i (for L^-T t),
    // the rows reaching their owners through the work copy, transposed.  No LL square in LDS, no LDS reads of the factor in the
    // solver's triangular products (round 4; the arithmetic, element by element and term by term, is that of the run-time-sized
    // branch: tests/test_kernels_physics.py::test_specialised_kernel_equals_the_runtime_sized_kernel).
    constexpr int NVc = kRegChol ? kSD.nv : 1;
    constexpr int RC = (NVc + kGroupLanes - 1) / kGroupLanes;  // rows (and, in the inverse, columns) per lane
    float Lrow[RC][NVc], Lcol[RC][NVc];  // Linv[i][k] of this lane's rows i / Linv[k][c] of its columns c (entries outside the triangle: 0)
    int ir[RC];       // this lane's rows; a surplus slot shadows the last row and never publishes
    bool own[RC];
    _Pragma("unroll") for (int q = 0; q < RC; ++q) { const int i = g + kGroupLanes * q; own[q] = i < NVc; ir[q] = own[q] ? i : NVc - 1; }
    // x = (L L^T)^-1 b with the factor in registers: the two masked products of solve_tri (full-length: the extra terms are exact zeros), the operand vector read
    // once for all of a lane's rows
    auto solve_regs = [&](const float* bvec, float* tmp, float* x) {
      float br[NVc];
      _Pragma("unroll") for (int k = 0; k < NVc; ++k) br[k] = bvec[k];
      _Pragma("unroll") for (int q = 0; q < RC; ++q) {
        float sacc = 0.f;
        _Pragma("unroll") for (int k = 0; k < NVc; ++k) sacc += (k < kGroupLanes * (q + 1) ? Lrow[q][k] : 0.f) * br[k];
        if (own[q]) tmp[ir[q]] = sacc;
      }
      SYNC();
      _Pragma("unroll") for (int k = 0; k < NVc; ++k) br[k] = tmp[k];
      _Pragma("unroll") for (int q = 0; q < RC; ++q) {
        float sacc = 0.f;
        _Pragma("unroll") for (int k = 0; k < NVc; ++k) sacc += (k >= kGroupLanes * q ? Lcol[q][k] : 0.f) * br[k];
        if (own[q]) x[ir[q]] = sacc;
      }
      SYNC();
    };
    auto solveL = [&](auto eul_c, const float* bvec, float* tmp, float* x) {
      if constexpr (kRegChol) { (void)eul_c; solve_regs(bvec, tmp, x); }
      else { (void)eul_c; solve_tri<NV>(LL, nv, bvec, tmp, x, g); }
    };
    _Pragma("unroll 1") for (int pass = 0; pass < 2; ++pass) {
    if (kRegChol) {
      const int ldc = P.ldc;
      const bool eul = pass == 1;
      float* col = t0;  // the column being eliminated, all rows (t0 is free here)
      float c[RC][NVc];
      _Pragma("unroll") for (int q = 0; q < RC; ++q) {
        const int i = ir[q];
        const float hd = eul ? h * TF(dof_damping)[i] : 0.f;
        if (spM) {
          _Pragma("unroll") for (int k4 = 0; k4 < (NVc + 3) / 4; ++k4) {
            const float4 w = Mq[k4 * nv + i];
            const float we[4] = {w.x, w.y, w.z, w.w};
            _Pragma("unroll") for (int e = 0; e < 4; ++e) { const int k = 4 * k4 + e; if (k < NVc) c[q][k] = (eul && k == i) ? we[e] + hd : we[e]; }
          }
        } else {
          _Pragma("unroll") for (int k = 0; k < NVc; ++k) {
            const float v = M[i * ldm + k];
            c[q][k] = (eul && k == i) ? v + hd : v;
          }
        }
      }
      static_for<NVc>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        _Pragma("unroll") for (int q = 0; q < RC; ++q) if (own[q]) col[ir[q]] = c[q][k];
        SYNC();
        const float r = rsqrtf(fmaxf(col[k], MJ_MINVAL));
        float l[NVc];  // the scaled column below the diagonal (rows k + 1 ..)
        _Pragma("unroll") for (int j = k + 1; j < NVc; ++j) l[j] = col[j] * r;
        SYNC();  // the column buffer is rewritten in the next step
        _Pragma("unroll") for (int q = 0; q < RC; ++q) {
          if (kGroupLanes * (q + 1) - 1 > k) {  // (otherwise every row of this slot lies on or above the diagonal by now)
            const bool below = ir[q] > k;
            const float a = c[q][k] * r;
            // entries right of the diagonal (j > i) are updated too: they are never read, and skipping them would cost a compare each
            _Pragma("unroll") for (int j = k + 1; j < NVc; ++j) c[q][j] = below ? c[q][j] - a * l[j] : c[q][j];
            c[q][k] = below ? a : c[q][k];
          }
        }
        SCHED_FENCE();
      });
      // publish the rows (lower triangle + squared diagonal)
      _Pragma("unroll") for (int q = 0; q < RC; ++q) {
        if (own[q]) { _Pragma("unroll") for (int k = 0; k < NVc; ++k) Cw[ir[q] * ldc + k] = c[q][k]; }
      }
      SYNC();
      if (pass == 0) PT(6);
      // inverse factor by forward substitution, a lane's columns: x[k] = 0 above the column's diagonal
      float x[RC][NVc];
      _Pragma("unroll") for (int i = 0; i < NVc; ++i) {
        const float d = rsqrtf(fmaxf(Cw[i * ldc + i], MJ_MINVAL));
        float sq_[RC];
        _Pragma("unroll") for (int q = 0; q < RC; ++q) sq_[q] = 0.f;
        _Pragma("unroll") for (int k = 0; k < i; ++k) {
          const float cik = Cw[i * ldc + k];
          _Pragma("unroll") for (int q = 0; q < RC; ++q) if (kGroupLanes * q <= k) sq_[q] += cik * x[q][k];  // (x[q][k] = 0 for k < 16 q: skipped)
        }
        _Pragma("unroll") for (int q = 0; q < RC; ++q) x[q][i] = i == ir[q] ? d : i > ir[q] ? -sq_[q] * d : 0.f;
        if (RC > 1) SCHED_FENCE();  // (two columns per lane: keep the rows' loads from piling up in registers ahead of their use)
      }
      // the columns stay with their lanes; the rows reach theirs through the work copy (all of its readers are done), transposed:
      // T[i][c] = Linv[i][c], zeros above the diagonal included
      SYNC();
      _Pragma("unroll") for (int q = 0; q < RC; ++q) {
        if (own[q]) { _Pragma("unroll") for (int i = 0; i < NVc; ++i) Cw[i * ldm + ir[q]] = x[q][i]; }
      }
      SYNC();
      _Pragma("unroll") for (int q = 0; q < RC; ++q) {
        _Pragma("unroll") for (int k = 0; k < NVc; ++k) {
          if (k < kGroupLanes * (q + 1)) Lrow[q][k] = Cw[ir[q] * ldm + k];
          if (k >= kGroupLanes * q) Lcol[q][k] = x[q][k];
        }
      }
      SYNC();  // the work copy's storage goes to the velocity / RNE scratch (trip 0)
    } else {
      // run-time-sized kernel / more than 32 dofs: the same two trips with the matrices in LDS.  Work copy `Lp`: a packed lower
      // triangle (row i at i (i + 1) / 2; region A2), factored column by column in LEFT-looking order - entry (i, k) starts as
      // M[i][k] and loses L[i][j] L[k][j] for j = 0 .. k - 1, one after the other: the subtractions the right-looking elimination
      // of the register-resident form applies to it, in the same order (bit-identical), but as a dot product of two rows that
      // stores nothing inside its loop (round 6; the right-looking sweep re-wrote the trailing triangle once per column through
      // LDS: a third of the step of a 33-dof robot).  The diagonal keeps L_kk^2.  Then the inverse factor `Li`, packed the same
      // way, one column per lane by forward substitution.  One factor at a time (M now, M + h D at the end of the step): half
      // the LDS of round 5's two squares.
      const bool eul = pass == 1;
      float* Lp = Cw; float* Li = LL;
      FOR_G(i, nv) {
        const int ti = i * (i + 1) / 2;
        const float hd = eul ? h * TF(dof_damping)[i] : 0.f;
        if (spM) {
          for (int k4 = 0; 4 * k4 <= i; ++k4) {
            const float4 w = Mq[k4 * nv + i];
            const float we[4] = {w.x, w.y, w.z, w.w};
            _Pragma("unroll") for (int e = 0; e < 4; ++e) { const int k = 4 * k4 + e; if (k <= i) Lp[ti + k] = (eul && k == i) ? we[e] + hd : we[e]; }
          }
        } else {
          for (int k = 0; k <= i; ++k) {
            const float v = M[i * ldm + k];
            Lp[ti + k] = (eul && k == i) ? v + hd : v;
          }
        }
      }
      SYNC();
      for (int k = 0; k < nv; ++k) {
        const int tk = k * (k + 1) / 2;
        float vv[4];  // this lane's rows g, g + 16, .. (at most 64 dofs)
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {
          const int i = g + kGroupLanes * q;
          vv[q] = 0.f;
          if (i >= k && i < nv) {
            const int ti = i * (i + 1) / 2;
            float v = Lp[ti + k];
  DOT_UNROLL
            for (int j = 0; j < k; ++j) v -= Lp[ti + j] * Lp[tk + j];
            vv[q] = v;
            if (i == k) Lp[tk + k] = v;
          }
        }
        SYNC();
        const float r = rsqrtf(fmaxf(Lp[tk + k], MJ_MINVAL));
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {
          const int i = g + kGroupLanes * q;
          if (i > k && i < nv) Lp[i * (i + 1) / 2 + k] = vv[q] * r;
        }
        SYNC();
      }
      if (pass == 0) PT(6);
      // triangular inverse, one column per lane (no cross-lane dependency inside a column)
      FOR_G(j, nv) {
        const int tj = j * (j + 1) / 2;
        Li[tj + j] = rsqrtf(fmaxf(Lp[tj + j], MJ_MINVAL));
        int ti = tj + j + 1;  // row j + 1
        for (int i = j + 1; i < nv; ++i) {
          float s1 = 0.f;
          int tkj = tj + j;  // Li[k][j], k = j
  #pragma unroll 4
          for (int k = j; k < i; ++k) { s1 += Lp[ti + k] * Li[tkj]; tkj += k + 1; }
          Li[ti + j] = -s1 * rsqrtf(fmaxf(Lp[ti + i], MJ_MINVAL));
          ti += i + 1;
        }
      }
    }
    SYNC();
    if (pass == 1) break;  // M + h D is factored: on to the integrator
    PT(7);
    // ================= fwd_velocity: com_vel, passive, rne (closed forms over ancestor masks) =======
    FOR_G(b, nb) {
      float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      u64 mask = b ? TU(body_ancdof_mask)[b] : 0ull;
      while (mask) {
        const int d = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        const float qd = qvel[d];
        for (int k = 0; k < 6; ++k) v[k] += cdof[6 * d + k] * qd;
      }
      for (int k = 0; k < 6; ++k) cvel[6 * b + k] = v[k];
    }
    FOR_G(d, nv) {
      const int j = TI(dof_jntid)[d];
      float* out = cdofdot + 6 * d;
      if (TI(jnt_type)[j] == JNT_FREE && d - TI(jnt_dofadr)[j] < 3) {
        for (int k = 0; k < 6; ++k) out[k] = 0.f;
      } else {
        float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        u64 mask = TU(dof_velmask)[d];
        while (mask) {
          const int e = __ffsll((long long)mask) - 1;
          mask &= mask - 1;
          const float qd = qvel[e];
          for (int k = 0; k < 6; ++k) v[k] += cdof[6 * e + k] * qd;
        }
        cross_motion(v, cdof + 6 * d, out);
      }
    }
    SYNC();
    FOR_G(b, nb) {
      float* f = cfrc + 6 * b;
      if (b == 0) {
        for (int k = 0; k < 6; ++k) f[k] = 0.f;
      } else {
        float acc[6] = {0.f, 0.f, 0.f, -TF(gravity)[0], -TF(gravity)[1], -TF(gravity)[2]};
        u64 mask = TU(body_ancdof_mask)[b];
        while (mask) {
          const int d = __ffsll((long long)mask) - 1;
          mask &= mask - 1;
          const float qd = qvel[d];
          for (int k = 0; k < 6; ++k) acc[k] += cdofdot[6 * d + k] * qd;
        }
        float ia[6], iv[6], cf[6];
        inert_mul(cinert + 10 * b, acc, ia);
        inert_mul(cinert + 10 * b, cvel + 6 * b, iv);
        cross_force(cvel + 6 * b, iv, cf);
        for (int k = 0; k < 6; ++k) f[k] = ia[k] + cf[k];
      }
    }
    SYNC();
    PT(8);
    // ---- qfrc_bias, passive, actuation -> qfrc_smooth -------------------------------------------------
    FOR_G(d, nv) {
      float f[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int w = 0; w < (nb > 64 ? 2 : 1); ++w) {
        u64 mask = TU(body_subtree_mask)[w * nb + TI(dof_bodyid)[d]];
        while (mask) {
          const int c = 64 * w + __ffsll((long long)mask) - 1;
          mask &= mask - 1;
          for (int k = 0; k < 6; ++k) f[k] += cfrc[6 * c + k];
        }
      }
      const float* cd = cdof + 6 * d;
      const float bias = cd[0] * f[0] + cd[1] * f[1] + cd[2] * f[2] + cd[3] * f[3] + cd[4] * f[4] + cd[5] * f[5];
      float passive = -TF(dof_damping)[d] * qvel[d];
      const int qa = TI(dof_qposadr)[d];
      if (qa >= 0) {
        const float stiff = TF(jnt_stiffness)[TI(dof_jntid)[d]];
        if (stiff != 0.f) passive -= stiff * (qpos[qa] - TF(qpos_spring)[qa]);
      }
      float act = 0.f;
      for (int u = 0; u < nu; ++u) {
        if (TI(act_dofid)[u] == d) {
          float c = ctrl[u];
          if (TI(act_ctrllimited)[u]) c = fminf(fmaxf(c, TF(act_ctrlrange)[2 * u]), TF(act_ctrlrange)[2 * u + 1]);
          const float gear = TF(act_gear)[u];
          const float len = gear * qpos[TI(act_qposadr)[u]], vel = gear * qvel[d];
          float fo = TF(act_gain)[u] * c + TF(act_bias)[3 * u] + TF(act_bias)[3 * u + 1] * len + TF(act_bias)[3 * u + 2] * vel;
          if (TI(act_forcelimited)[u]) fo = fminf(fmaxf(fo, TF(act_forcerange)[2 * u]), TF(act_forcerange)[2 * u + 1]);
          act += fo * gear;
        }
      }
      // MJCF <joint actuatorfrcrange>: the joint's total actuator force is clamped (MJX fwd_actuation; -FLT_MAX / FLT_MAX when the joint has none);
      // comparisons, not fmin / fmax: a NaN must stay a NaN for the guard below
      {
        const float lo = TF(dof_actfrcrange)[2 * d], hi = TF(dof_actfrcrange)[2 * d + 1];
        act = act < lo ? lo : (act > hi ? hi : act);
      }
      qfs[d] = passive - bias + act;
      // qfrc_actuator is part of the new record / observation (env.py:252) and of the NaN guard, not of the solver: it leaves here
      bad_lane |= (int)isnan(act);
      if (MODE != 2 && frame == frames - 1 && valid) {
        recw[o_qa + d] = act;
        if (MODE == 0) { if (env == 0 && a.reset_out) a.reset_out[o_qa + d] = act; if (a.obs) a.obs[(size_t)env * a.obs_ld + o_qa + d] = act; }
      }
      if (MODE == 2) {
        if (valid && a.probe.qfrc_bias) a.probe.qfrc_bias[(size_t)env * nv + d] = bias;
        if (valid && a.probe.qfrc_passive) a.probe.qfrc_passive[(size_t)env * nv + d] = passive;
        if (valid && a.probe.qfrc_actuator) a.probe.qfrc_actuator[(size_t)env * nv + d] = act;
      }
    }
    // cinert and cvel have done their work in the dynamics (RNE above was the last reader); what remains is their place in the
    // new state record / observation (env.py:246-259) and the NaN guard: both are served here, and their LDS goes to the Jacobian
    // (where the values leave for the record anyway, the NaN test rides on that read: the world body's entries are zeros by construction)
    if (MODE != 2 && frame == frames - 1 && mv.include_c && valid) {
      FOR_G(i, 10 * (nb - 1)) {
        const float v = cinert[10 + i];
        bad_lane |= (int)isnan(v);
        recw[o_ci + i] = v;
        if (MODE == 0) { if (env == 0 && a.reset_out) a.reset_out[o_ci + i] = v; if (a.obs) a.obs[(size_t)env * a.obs_ld + o_ci + i] = v; }
      }
      FOR_G(i, 6 * (nb - 1)) {
        const float v = cvel[6 + i];
        bad_lane |= (int)isnan(v);
        recw[o_cv + i] = v;
        if (MODE == 0) { if (env == 0 && a.reset_out) a.reset_out[o_cv + i] = v; if (a.obs) a.obs[(size_t)env * a.obs_ld + o_cv + i] = v; }
      }
    } else {
      FOR_G(i, 10 * nb) bad_lane |= (int)isnan(cinert[i]);
      FOR_G(i, 6 * nb) bad_lane |= (int)isnan(cvel[i]);
    }
    if (kFlagAsBallot) { bad_mid = group16_flags_or(bad_mid, group16_flags(bad_lane != 0)); bad_lane = 0; }
    if (MODE == 2 && valid) {
      if (a.probe.cinert) FOR_G(i, nb * 10) a.probe.cinert[(size_t)env * nb * 10 + i] = cinert[i];
      if (a.probe.cvel) FOR_G(i, nb * 6) a.probe.cvel[(size_t)env * nb * 6 + i] = cvel[i];
    }
    SYNC();  // cfrc / cdofdot / cvel (region A3) and cinert are dead from here: the Jacobian (A4) may overwrite them
    PT(9);
    solveL(std::false_type{}, qfs, t0, qas);  // fwd_acceleration: qacc_smooth = M^-1 qfrc_smooth
    PT(10);
    // ================= make_constraint ===================================================================
    // Rows: nlim joint limits, then four pyramid rows per contact slot.  A limit row has ONE non-zero entry (+-1 at the joint's
    // dof): it is kept as that sign (dsgn, per dof; 0 = no active limit) and its row index (drow), and only the contact rows are
    // a dense [4 ncon][nv] matrix in LDS.  Products with the limit rows are written out where they occur; they equal what the
    // dense row gave bit for bit (the other terms of that row's sum were exact zeros).
    if (!spJ) FOR_G(r, 4 * ncon) for (int k = 0; k < nv; ++k) J[r * ldj + k] = 0.f;
    FOR_G(i, nv) dlim[i] = 0;
    // Jacobian in global memory (large robots): per contact slot the set of dofs its rows touch - the ancestors of its one or two bodies -
    // if the contact is active, else none; the rows are written and read through these sets only (every other entry is an exact zero)
    u64* jmask = reinterpret_cast<u64*>(S + P.jmask);
    float4* Jc = reinterpret_cast<float4*>(G + (spJ ? P.gJc : 0));  // [ncon][nv]: (rows 4 c .. 4 c + 3) of dof d
    float4* Jq = reinterpret_cast<float4*>(G + (spJ ? P.gJq : 0));  // [nvq][4 ncon]: (dofs 4 q .. 4 q + 3) of row r
    if (spJ) {
      FOR_G(c, ncon) {
        u64 m = TU(body_ancdof_mask)[TI(con_bodyid)[c]];
        if (c >= nplane) m |= TU(body_ancdof_mask)[TI(pair_body)[2 * (c - nplane)]];
        jmask[c] = condist[c] < 0.f ? m : 0ull;
      }
    }
    SYNC();
    FOR_G(r, nlim) {  // joint limits: one row each
      const int jid = TI(lim_jntid)[r];
      const int qa = TI(jnt_qposadr)[jid], da = TI(jnt_dofadr)[jid];
      const float dlo = qpos[qa] - TF(jnt_range)[2 * jid], dhi = TF(jnt_range)[2 * jid + 1] - qpos[qa];
      const float pos = fminf(dlo, dhi);
      const bool act = pos < 0.f;
      if (act) dlim[da] = dlo < dhi ? r + 1 : -(r + 1);
      jv[r] = act ? pos : 0.f;               // pos, parked in jv until the row parameters are built
      jaref[r] = act ? TF(dof_invweight0)[da] : 0.f;  // invweight, parked in jaref
    }
    // the four pyramid rows of contact c at dof d (false: the dof moves neither body - the entries are zeros)
    auto contact_rows = [&](int c, int d, float* r4) -> bool {
      // translational Jacobian of the contact point: body 2 minus body 1 (body 1 = world for a ground contact)
      V3 jp = {0.f, 0.f, 0.f};
      bool any = false;
      for (int side = 0; side < 2; ++side) {
        if (side == 1 && c < nplane) break;
        const int b = side == 0 ? TI(con_bodyid)[c] : TI(pair_body)[2 * (c - nplane)];
        if ((TU(body_ancdof_mask)[b] >> d) & 1ull) {
          int ri = 0;
          for (int r = 0; r < nroot; ++r) if (TI(root_body)[r] == TI(body_rootid)[b]) ri = r;
          const V3 off = sub3(ld3(conpos + 3 * c), ld3(rootcom + 3 * ri));
          const V3 jb = add3(ld3(cdof + 6 * d + 3), cross3(ld3(cdof + 6 * d), off));
          jp = side == 0 ? add3(jp, jb) : sub3(jp, jb);
          any = true;
        }
      }
      if (any) {
        const V3 n = ld3(confr + 6 * c), t1 = ld3(confr + 6 * c + 3), t2 = cross3(n, t1);
        const float jn = dot3(n, jp), jt1 = dot3(t1, jp), jt2 = dot3(t2, jp);
        const float mu = TF(con_friction)[3 * c];
        r4[0] = jn + mu * jt1; r4[1] = jn - mu * jt1; r4[2] = jn + mu * jt2; r4[3] = jn - mu * jt2;
      }
      return any;
    };
    if (spJ) {
      const int nv4 = 4 * nvq, R = 4 * ncon;
      for (int item = g; item < ncon * nv4; item += kGroupLanes) {  // (contact, dof) per item, the last quad's padding included
        const int c = item / nv4, d = item - c * nv4;
        const u64 m = jmask[c];
        if (!((m >> (d & ~3)) & 15ull)) continue;  // nobody reads this quad
        float r4[4] = {0.f, 0.f, 0.f, 0.f};
        const bool any = d < nv && ((m >> d) & 1ull) && contact_rows(c, d, r4);
        if (any) Jc[c * nv + d] = make_float4(r4[0], r4[1], r4[2], r4[3]);
        float* q = reinterpret_cast<float*>(Jq + ((d >> 2) * R + 4 * c)) + (d & 3);
        q[0] = r4[0]; q[4] = r4[1]; q[8] = r4[2]; q[12] = r4[3];
      }
    } else
    for (int item = g; item < ncon * nv; item += kGroupLanes) {  // contacts: 4 pyramid rows, (contact, dof) per item
      // (the same arithmetic as contact_rows, written out as in rounds 2 - 5: the BASELINE robots' kernels keep their instruction stream)
      const int c = item / nv, d = item - c * nv;
      if (condist[c] < 0.f) {
        // translational Jacobian of the contact point: body 2 minus body 1 (body 1 = world for a ground contact)
        V3 jp = {0.f, 0.f, 0.f};
        bool any = false;
        for (int side = 0; side < 2; ++side) {
          if (side == 1 && c < nplane) break;
          const int b = side == 0 ? TI(con_bodyid)[c] : TI(pair_body)[2 * (c - nplane)];
          if ((TU(body_ancdof_mask)[b] >> d) & 1ull) {
            int ri = 0;
            for (int r = 0; r < nroot; ++r) if (TI(root_body)[r] == TI(body_rootid)[b]) ri = r;
            const V3 off = sub3(ld3(conpos + 3 * c), ld3(rootcom + 3 * ri));
            const V3 jb = add3(ld3(cdof + 6 * d + 3), cross3(ld3(cdof + 6 * d), off));
            jp = side == 0 ? add3(jp, jb) : sub3(jp, jb);
            any = true;
          }
        }
        if (any) {
          const V3 n = ld3(confr + 6 * c), t1 = ld3(confr + 6 * c + 3), t2 = cross3(n, t1);
          const float jn = dot3(n, jp), jt1 = dot3(t1, jp), jt2 = dot3(t2, jp);
          const float mu = TF(con_friction)[3 * c];
          const int r0 = 4 * c;
          J[(r0 + 0) * ldj + d] = jn + mu * jt1;
          J[(r0 + 1) * ldj + d] = jn - mu * jt1;
          J[(r0 + 2) * ldj + d] = jn + mu * jt2;
          J[(r0 + 3) * ldj + d] = jn - mu * jt2;
        }
      }
    }
    FOR_G(c, ncon) {
      const bool act = condist[c] < 0.f;
      const float mu = TF(con_friction)[3 * c];
      float tw = TF(body_invweight0)[2 * TI(con_bodyid)[c]];
      if (c >= nplane) tw += TF(body_invweight0)[2 * TI(pair_body)[2 * (c - nplane)]];
      const float iw = (tw + mu * mu * tw) * 2.f * mu * mu / mv.impratio;
      for (int k = 0; k < 4; ++k) {
        jv[nlim + 4 * c + k] = act ? condist[c] : 0.f;
        jaref[nlim + 4 * c + k] = act ? iw : 0.f;
      }
    }
    if (spJ) GSYNC(); else SYNC();
    float k_lim, b_lim, k_con, b_con;
    kb_params(TF(limit_solref), TF(limit_solimp), h, k_lim, b_lim);
    kb_params(TF(contact_solref), TF(contact_solimp), h, k_con, b_con);
    // row r of the constraint Jacobian times an nv-vector / column i times an nefc-vector
    auto jrow_dot = [&](int r, const float* x) {
      if (r < nlim) { const int da = TI(jnt_dofadr)[TI(lim_jntid)[r]]; return lim_sign(dlim[da]) * x[da]; }
      float s = 0.f;
      if (spJ) {
        // the row's dof quads that hold anything, in ascending order: the dense sum's terms minus exact zeros
        const int rc = r - nlim, R = 4 * ncon;
        const u64 m = jmask[rc >> 2];
  #pragma unroll 4
        for (int k4 = 0; k4 < nvq; ++k4) {
          if ((m >> (4 * k4)) & 15ull) {
            const float4 w = Jq[k4 * R + rc];
            const int k = 4 * k4;
            s += w.x * x[k];
            s += w.y * (k + 1 < nv ? x[k + 1] : 0.f);
            s += w.z * (k + 2 < nv ? x[k + 2] : 0.f);
            s += w.w * (k + 3 < nv ? x[k + 3] : 0.f);
          }
        }
        return s;
      }
      const float* jr = J + (r - nlim) * ldj;
      DOT_UNROLL for (int k = 0; k < nv; ++k) s += jr[k] * x[k];
      return s;
    };
    auto jcol_dot = [&](int i, const float* f) {
      float s = 0.f;
      if (nlim > 0) { const int dl = dlim[i]; s = lim_sign(dl) * f[lim_row(dl)]; }
      if (spJ) {
  #pragma unroll 4
        for (int c = 0; c < ncon; ++c) {
          if ((jmask[c] >> i) & 1ull) {
            const float4 w = Jc[c * nv + i];
            const float* fc = f + nlim + 4 * c;
            s += w.x * fc[0]; s += w.y * fc[1]; s += w.z * fc[2]; s += w.w * fc[3];
          }
        }
        return s;
      }
      const int nc4 = 4 * ncon;
      DOT_UNROLL for (int r = 0; r < nc4; ++r) s += J[r * ldj + i] * f[nlim + r];
      return s;
    };
    auto mrow_dot = [&](int i, const float* x) {
      float s = 0.f;
      if (spM) {
        // the same terms in the same order; a quad's entries beyond nv are zeros times a masked operand (x is padded to a quad in LDS)
  #pragma unroll 4
        for (int k4 = 0; k4 < nvq; ++k4) {
          const float4 w = Mq[k4 * nv + i];
          const int k = 4 * k4;
          s += w.x * x[k];
          s += w.y * (k + 1 < nv ? x[k + 1] : 0.f);
          s += w.z * (k + 2 < nv ? x[k + 2] : 0.f);
          s += w.w * (k + 3 < nv ? x[k + 3] : 0.f);
        }
        return s;
      }
      DOT_UNROLL for (int k = 0; k < nv; ++k) s += M[i * ldm + k] * x[k];
      return s;
    };
    auto row_params = [&](int r, float s) {
      const float pos = jv[r], iw = jaref[r];
      const bool act = iw > 0.f;  // inactive rows are inert: J = 0, aref = 0, D = 0
      const bool lim = r < nlim;
      const float k = lim ? k_lim : k_con, b = lim ? b_lim : b_con;
      const float imp = impedance(lim ? TF(limit_solimp) : TF(contact_solimp), pos);
      const float R = fmaxf(iw * (1.f - imp) / imp, MJ_MINVAL);
      eD[r] = act ? 1.f / R : 0.f;
      earef[r] = act ? -b * s - k * imp * pos : 0.f;
    };
    FOR_G(r, nefc) row_params(r, jrow_dot(r, qvel));
    SYNC();
    PT(11);
    // ================= solve: CG (Polak-Ribiere, M^-1 preconditioner) =====================================
    const float scale = mv.meaninertia * (float)(nv > 1 ? nv : 1);
    float cost = 0.f, prev_cost = 0.f, gauss = 0.f;
    int niter = 0;
    if (nefc == 0) {
      FOR_G(i, nv) { qacc[i] = qas[i]; qfc[i] = 0.f; }
      SYNC();
    } else {
      // Context.create(qacc): Ma = M qacc, Jaref = J qacc - aref, cost; only groups with `take` are modified.
      auto ctx_eval = [&](const float* src, bool take, float& gs_out, float& cs_out) {
        if (take) FOR_G(i, nv) qacc[i] = src[i];
        SYNC();
        if (take) {
          FOR_G(i, nv) Ma[i] = mrow_dot(i, qacc);
          // (limit rows and contact rows in loops of their own: 16 lanes on 16 dense rows at a time, not on a mixture)
          FOR_G(r, nlim) jaref[r] = jrow_dot(r, qacc) - earef[r];
          FOR_G(rc, 4 * ncon) jaref[nlim + rc] = jrow_dot(nlim + rc, qacc) - earef[nlim + rc];
        }
        SYNC();
        float gs = 0.f, cs = 0.f;
        FOR_G(i, nv) gs += (Ma[i] - qfs[i]) * (qacc[i] - qas[i]);
        FOR_G(r, nefc) { const float x = jaref[r]; if (x < 0.f) cs += eD[r] * x * x; }
        gs = 0.5f * group16_sum(gs);
        cs = 0.5f * group16_sum(cs) + gs;
        if (take) { gs_out = gs; cs_out = cs; }
        SYNC();
      };
      // warm start: keep qacc_warmstart if its cost beats qacc_smooth's (MJX solver.solve)
      float cost_w = 0.f, gauss_w = 0.f;
      ctx_eval(warm, true, gauss_w, cost_w);
      // fixed-size kernel: the warm-start context (this lane's entries of Ma and Jaref) is parked in registers while the smooth
      // start is evaluated, and put back if it wins - the run-time-sized kernel evaluates it a second time instead
      constexpr int kNvR = kDims ? (kSD.nv + kGroupLanes - 1) / kGroupLanes : 1, kEfR = kDims && kSD.nlimit + 4 * kSD.ncon > 0 ? (kSD.nlimit + 4 * kSD.ncon + kGroupLanes - 1) / kGroupLanes : 1;  // (a model without constraint rows never gets here)
      float keep_ma[kNvR], keep_ja[kEfR];
      if (kDims) {
        _Pragma("unroll") for (int j = 0; j < kNvR; ++j) { const int i = g + kGroupLanes * j; keep_ma[j] = i < nv ? Ma[i] : 0.f; }
        _Pragma("unroll") for (int j = 0; j < kEfR; ++j) { const int r = g + kGroupLanes * j; keep_ja[j] = r < nefc ? jaref[r] : 0.f; }
      }
      ctx_eval(qas, true, gauss, cost);
      const bool use_warm = cost_w < cost;
      if (kDims) {
        if (use_warm) {
          _Pragma("unroll") for (int j = 0; j < kNvR; ++j) { const int i = g + kGroupLanes * j; if (i < nv) { qacc[i] = warm[i]; Ma[i] = keep_ma[j]; } }
          _Pragma("unroll") for (int j = 0; j < kEfR; ++j) { const int r = g + kGroupLanes * j; if (r < nefc) jaref[r] = keep_ja[j]; }
          gauss = gauss_w; cost = cost_w;
        }
        SYNC();
      } else if (wave_any(use_warm)) {
        ctx_eval(warm, use_warm, gauss, cost);
      }
      prev_cost = INFINITY;  // MJX Context.create: cost starts at inf, so the first improvement is inf
      // update_constraint + update_gradient at the starting point
      FOR_G(r, nefc) { const float x = jaref[r]; force[r] = x < 0.f ? -eD[r] * x : 0.f; }
      SYNC();
      FOR_G(i, nv) { const float s = jcol_dot(i, force); qfc[i] = s; grad[i] = Ma[i] - qfs[i] - s; }
      SYNC();
      solveL(std::false_type{}, grad, t0, Mgrad);
      FOR_G(i, nv) search[i] = -Mgrad[i];
      SYNC();
      PT(12);
      for (int it = 0; it < mv.iterations; ++it) {
        float gn = 0.f;
        FOR_G(i, nv) gn += grad[i] * grad[i];
        gn = sqrtf(group16_sum(gn)) / scale;
        const float improvement = (prev_cost - cost) / scale;
        const bool run = !(bool)((int)(niter >= mv.iterations) | (int)(improvement < mv.tolerance) | (int)(gn < mv.tolerance));
        if (!wave_any(run)) break;
        PT(13 + (it < 6 ? it : 6));
        // ---------------- line search ----------------
        float sn = 0.f, sMa = 0.f, sq = 0.f;
        FOR_G(i, nv) mvv[i] = mrow_dot(i, search);
        FOR_G(r, nlim) jv[r] = jrow_dot(r, search);
        FOR_G(rc, 4 * ncon) jv[nlim + rc] = jrow_dot(nlim + rc, search);
        FOR_G(i, nv) { sn += search[i] * search[i]; sMa += search[i] * Ma[i]; sq += search[i] * qfs[i]; }
        SYNC();
        if (it == 0) PT(24);
        float smv = 0.f;
        FOR_G(i, nv) smv += search[i] * mvv[i];
        sn = group16_sum(sn); sMa = group16_sum(sMa); sq = group16_sum(sq); smv = group16_sum(smv);
        const float gtol = mv.tolerance * mv.ls_tolerance * sqrtf(sn) * scale;
        const float qg0 = gauss, qg1 = sMa - sq, qg2 = 0.5f * smv;
        // three trial steps at once: sums of the active rows' quadratics
        // A lane's rows (g, g + 16, ...) do not change during the line search.  With compile-time dims their quadratics live in
        // registers for the whole search (kRows of them per lane); the run-time-sized kernel re-reads them from LDS per trial.
        constexpr int kRows = kDims && kSD.nlimit + 4 * kSD.ncon > 0 ? (kSD.nlimit + 4 * kSD.ncon + kGroupLanes - 1) / kGroupLanes : 1;
        float rja[kRows], rv[kRows], rc0[kRows], rc1[kRows], rc2[kRows];
        if (kDims) {
          _Pragma("unroll") for (int j = 0; j < kRows; ++j) {
            const int r = g + kGroupLanes * j;
            const bool in = r < nefc;
            const float ja = in ? jaref[in ? r : 0] : 0.f, v = in ? jv[in ? r : 0] : 0.f, d = in ? eD[in ? r : 0] : 0.f;
            rja[j] = ja; rv[j] = v;
            rc0[j] = 0.5f * ja * ja * d; rc1[j] = v * ja * d; rc2[j] = 0.5f * v * v * d;  // a row past nefc never tests active: 0 + a 0 < 0 is false
          }
        }
        auto eval3 = [&](float a0, float a1, float a2, LsPoint& o0, LsPoint& o1, LsPoint& o2) {
          float q[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if (kDims) {
            _Pragma("unroll") for (int j = 0; j < kRows; ++j) {
              const float ja = rja[j], v = rv[j], c0 = rc0[j], c1 = rc1[j], c2 = rc2[j];
              if (ja + a0 * v < 0.f) { q[0] += c0; q[1] += c1; q[2] += c2; }
              if (ja + a1 * v < 0.f) { q[3] += c0; q[4] += c1; q[5] += c2; }
              if (ja + a2 * v < 0.f) { q[6] += c0; q[7] += c1; q[8] += c2; }
            }
          } else {
            FOR_G(r, nefc) {
              const float ja = jaref[r], v = jv[r], d = eD[r];
              const float c0 = 0.5f * ja * ja * d, c1 = v * ja * d, c2 = 0.5f * v * v * d;
              if (ja + a0 * v < 0.f) { q[0] += c0; q[1] += c1; q[2] += c2; }
              if (ja + a1 * v < 0.f) { q[3] += c0; q[4] += c1; q[5] += c2; }
              if (ja + a2 * v < 0.f) { q[6] += c0; q[7] += c1; q[8] += c2; }
            }
          }
          for (int k = 0; k < 9; ++k) q[k] = group16_sum(q[k]);
          o0 = ls_make(a0, q[0] + qg0, q[1] + qg1, q[2] + qg2);
          o1 = ls_make(a1, q[3] + qg0, q[4] + qg1, q[5] + qg2);
          o2 = ls_make(a2, q[6] + qg0, q[7] + qg1, q[8] + qg2);
        };
        // one trial step: the starting point and the first Newton point (same row order and reduction as eval3)
        auto eval1 = [&](float a0, LsPoint& o0) {
          float q0 = 0.f, q1 = 0.f, q2 = 0.f;
          if (kDims) {
            _Pragma("unroll") for (int j = 0; j < kRows; ++j) {
              if (rja[j] + a0 * rv[j] < 0.f) { q0 += rc0[j]; q1 += rc1[j]; q2 += rc2[j]; }
            }
          } else {
            FOR_G(r, nefc) {
              const float ja = jaref[r], v = jv[r], d = eD[r];
              if (ja + a0 * v < 0.f) { q0 += 0.5f * ja * ja * d; q1 += v * ja * d; q2 += 0.5f * v * v * d; }
            }
          }
          q0 = group16_sum(q0); q1 = group16_sum(q1); q2 = group16_sum(q2);
          o0 = ls_make(a0, q0 + qg0, q1 + qg1, q2 + qg2);
        };
        if (it == 0) PT(25);
        LsPoint p0, lo, hi;
        eval1(0.f, p0);
        eval1(p0.alpha - p0.d0 / p0.d1, lo);
        {
          const bool lesser = lo.d0 < p0.d0;
          const LsPoint nlo = ls_sel(lesser, lo, p0), nhi = ls_sel(lesser, p0, lo);
          lo = nlo; hi = nhi;
        }
        if (it == 0) PT(26);
        bool swap = true;
        int ls_iter = 0;
        for (int li = 0; li < mv.ls_iterations; ++li) {
          const bool done = (bool)((int)(ls_iter >= mv.ls_iterations) | (int)!swap | (int)((lo.d0 < 0.f) & (lo.d0 > -gtol)) | (int)((hi.d0 > 0.f) & (hi.d0 < gtol)));
          const bool go = (bool)((int)!done & (int)run);
          if (!wave_any(go)) break;
          LsPoint lo_next, hi_next, mid;
          eval3(lo.alpha - lo.d0 / lo.d1, hi.alpha - hi.d0 / hi.d1, 0.5f * (lo.alpha + hi.alpha), lo_next, hi_next, mid);
          LsPoint nlo = lo, nhi = hi;
          const bool s1 = in_bracket(nlo, lo_next); nlo = ls_sel(s1, lo_next, nlo);
          const bool s2 = in_bracket(nlo, mid);     nlo = ls_sel(s2, mid, nlo);
          const bool s3 = in_bracket(nlo, hi_next); nlo = ls_sel(s3, hi_next, nlo);
          const bool s4 = in_bracket(nhi, hi_next); nhi = ls_sel(s4, hi_next, nhi);
          const bool s5 = in_bracket(nhi, mid);     nhi = ls_sel(s5, mid, nhi);
          const bool s6 = in_bracket(nhi, lo_next); nhi = ls_sel(s6, lo_next, nhi);
          if (go) { lo = nlo; hi = nhi; swap = (bool)((int)s1 | (int)s2 | (int)s3 | (int)s4 | (int)s5 | (int)s6); ls_iter += 1; }
        }
        if (it == 0) PT(27);
        const bool improved = (bool)((int)(lo.cost < p0.cost) | (int)(hi.cost < p0.cost));
        const float alpha = (lo.cost < hi.cost) ? lo.alpha : hi.alpha;
        if ((bool)((int)improved & (int)run)) {
          FOR_G(i, nv) { qacc[i] += alpha * search[i]; Ma[i] += alpha * mvv[i]; }
          FOR_G(r, nefc) jaref[r] += alpha * jv[r];
        }
        SYNC();
        if (it == 0) PT(28);
        // ---------------- update_constraint, update_gradient, Polak-Ribiere ----------------
        FOR_G(i, nv) { t1[i] = Mgrad[i]; }  // previous Mgrad (previous grad is re-read below before being overwritten)
        FOR_G(r, nefc) { const float x = jaref[r]; force[r] = x < 0.f ? -eD[r] * x : 0.f; }
        SYNC();
        float gs = 0.f, cs = 0.f, pgm = 0.f;
        FOR_G(i, nv) gs += (Ma[i] - qfs[i]) * (qacc[i] - qas[i]);
        FOR_G(r, nefc) { const float x = jaref[r]; if (x < 0.f) cs += eD[r] * x * x; }
        FOR_G(i, nv) pgm += grad[i] * Mgrad[i];
        gs = 0.5f * group16_sum(gs);
        cs = 0.5f * group16_sum(cs) + gs;
        pgm = group16_sum(pgm);
        if (run) { prev_cost = cost; cost = cs; gauss = gs; }
        FOR_G(i, nv) {
          const float s = jcol_dot(i, force);
          if (run) { qfc[i] = s; grad[i] = Ma[i] - qfs[i] - s; }
        }
        SYNC();
        if (it == 0) PT(29);
        solveL(std::false_type{}, grad, t0, mvv);  // candidate Mgrad (mvv is free again)
        float num = 0.f;
        FOR_G(i, nv) num += grad[i] * (mvv[i] - t1[i]);
        num = group16_sum(num);
        if (it == 0) PT(30);
        const float beta = fmaxf(0.f, num / fmaxf(MJ_MINVAL, pgm));
        if (run) {
          FOR_G(i, nv) { Mgrad[i] = mvv[i]; search[i] = -mvv[i] + beta * search[i]; }
          niter += 1;
        }
        SYNC();
      }
    }
    PT(20);
    // ---- probe outputs (parity tests) ---------------------------------------------------------------------
    if (MODE == 2 && valid) {
      const mppo_forward_probe_t& pr = a.probe;
      if (pr.qM) FOR_G(i, nv) for (int k = 0; k < nv; ++k) pr.qM[((size_t)env * nv + i) * nv + k] = Mget(i, k);
      if (pr.qacc_smooth) FOR_G(i, nv) pr.qacc_smooth[(size_t)env * nv + i] = qas[i];
      if (pr.qacc) FOR_G(i, nv) pr.qacc[(size_t)env * nv + i] = qacc[i];
      if (pr.efc_J) FOR_G(r, nefc) for (int k = 0; k < nv; ++k)
        pr.efc_J[((size_t)env * nefc + r) * nv + k] = r >= nlim ? (spJ ? (((jmask[(r - nlim) >> 2] >> k) & 1ull) ? reinterpret_cast<const float*>(Jc + ((r - nlim) >> 2) * nv + k)[(r - nlim) & 3] : 0.f)
                                                                        : J[(r - nlim) * ldj + k])
                                                            : (k == TI(jnt_dofadr)[TI(lim_jntid)[r]] ? lim_sign(dlim[k]) : 0.f);
      if (pr.efc_D) FOR_G(r, nefc) pr.efc_D[(size_t)env * nefc + r] = eD[r];
      if (pr.efc_aref) FOR_G(r, nefc) pr.efc_aref[(size_t)env * nefc + r] = earef[r];
      if (pr.subtree_com1 && g == 0) pr.subtree_com1[env] = new_comx;
      if (pr.solver_niter && g == 0) pr.solver_niter[env] = niter;
    }
    if (MODE == 0) break;  // pipeline_init = forward only
    PT(21);
    // ================= euler: implicit damping, semi-implicit integration ====================================
    FOR_G(i, nv) t1[i] = qfs[i] + qfc[i];
    SYNC();
    }  // (factorisation trips)
    if (MODE == 0) break;
    solveL(std::true_type{}, t1, t0, mvv);  // mvv = (M + h D)^-1 (qfrc_smooth + qfrc_constraint)
    if (MODE == 2) {
      if (valid && a.probe.qacc_euler) FOR_G(i, nv) a.probe.qacc_euler[(size_t)env * nv + i] = mvv[i];
      break;
    }
    FOR_G(i, nv) { qvel[i] += h * mvv[i]; warm[i] = qacc[i]; }  // qacc_warmstart <- solver qacc
    SYNC();
    FOR_G(j, njnt) {
      const int qa = TI(jnt_qposadr)[j], da = TI(jnt_dofadr)[j];
      if (TI(jnt_type)[j] == JNT_FREE) {
        for (int k = 0; k < 3; ++k) qpos[qa + k] += h * qvel[da + k];
        const V3 w = ld3(qvel + da + 3);
        const float n = sqrtf(dot3(w, w));
        const V3 ax = n > 0.f ? mul3(w, 1.f / n) : w;
        st4(qpos + qa + 3, qnormalize(qmul(ld4(qpos + qa + 3), axis_angle(ax, n * h))));
      } else {
        qpos[qa] += h * qvel[da];
      }
    }
    SYNC();
  }

  if (MODE == 2) return;

  PT(22);
  // ================= epilogue: observation, reward, done, auto-reset, metrics, new record ====================
  // The new record's derived fields (cinert, cvel, qfrc_actuator, subtree_com) are those of the LAST forward
  // pass, i.e. they belong to the pre-integration pose: exactly what the MJX data carries.
  if (MODE == 0) {
    // reset: record = [qpos0, 0, cinert[1:], cvel[1:], qfrc_actuator | pad | qacc_warmstart = qacc | com_x | time = 0]
    // (cinert / cvel were written when RNE had read them, qfrc_actuator when it was computed)
    FOR_G(i, mv.rec_dim) {
      if (i >= o_ci && i < O) continue;  // (cinert, cvel, qfrc_actuator: in place)
      float v = 0.f;
      if (i < nq) v = qpos[i];
      else if (i < o_ci) v = 0.f;
      else if (i < OP) v = 0.f;
      else if (i < OP + nv) v = qacc[i - OP];
      else if (i == OP + nv) v = new_comx;
      if (valid) {
        recw[i] = v;
        if (env == 0 && a.reset_out) a.reset_out[i] = v;
        if (i < OP && a.obs) a.obs[(size_t)env * a.obs_ld + i] = v;
      }
    }
    if (valid && g == 0) {
      if (a.reward) a.reward[env] = 0.f;
      if (a.done) a.done[env] = 0;
      if (a.met.episode_returns) {
        a.met.episode_returns[env] = 0.f; a.met.episode_lengths[env] = 0; a.met.returned_episode_returns[env] = 0.f;
        a.met.returned_episode_lengths[env] = 0; a.met.timestep[env] = 0; a.met.returned_episode[env] = 0;
      }
    }
    return;
  }
  // ---- step ----
  float asq = 0.f;
  FOR_G(i, nu) { const float c = a.action[(size_t)env * a.act_ld + i]; asq += c * c; }
  asq = group16_sum(asq);
  const mppo_reward_cfg_t& rc = a.rc;
  const float pos_r = expf(-rc.exp_coefficient * pre_p0) - rc.subtraction_factor * fminf(fmaxf(pre_p0, 0.f), rc.max_diff_norm);
  float healthy = pre_z < rc.height_min_z ? 0.f : 1.f;
  healthy = pre_z > rc.height_max_z ? 0.f : healthy;
  const float dt_env = h * (float)a.n_frames;
  const float vel = (new_comx - pre_comx) / dt_env;
  const float reward = rc.w_ctrl_cost * (-asq) + rc.w_original_pos * pos_r + rc.w_velocity * vel + rc.w_is_healthy * healthy;
  // done: height window on the POST-step state (env.py:171,238-242) or any NaN in the stepped state (env.py:173-176; cinert and
  // cvel were checked when they left LDS)
  const float z = qpos[2];
  // (bitwise accumulation: a short-circuit || would put every load behind its own branch)
  int badi = kFlagAsBallot ? 0 : bad_lane;
  FOR_G(i, nq) badi |= (int)isnan(qpos[i]);
  FOR_G(i, nv) badi |= (int)isnan(qvel[i]) | (int)isnan(warm[i]);  // (qfrc_actuator was checked when it left for the record)
  badi |= (int)isnan(new_comx);
  const bool bad = badi != 0;
  const bool done = (bool)((int)!((rc.height_min_z < z) & (z < rc.height_max_z)) | (int)group16_any(bad) | (int)(kFlagAsBallot && group16_flag_set(bad_mid)));
  // The new record: qpos, qvel, the warm start, com_x and the time from LDS (cinert / cvel / qfrc_actuator are in place already) - or,
  // when the episode ended, the reset record, which is also the observation to emit (env.py:179-180).  Ended episodes are rare: the
  // reset record is not even loaded otherwise.
  const int rec_dim = mv.rec_dim;
  if (valid) {
    if (done) {
      constexpr int kChunk = 8;
      for (int i0 = g; i0 < rec_dim; i0 += kGroupLanes * kChunk) {
        float rst[kChunk];
        _Pragma("unroll") for (int u = 0; u < kChunk; ++u) { const int i = i0 + kGroupLanes * u; rst[u] = a.reset_in[i < rec_dim ? i : rec_dim - 1]; }
        _Pragma("unroll") for (int u = 0; u < kChunk; ++u) {
          const int i = i0 + kGroupLanes * u;
          if (i < rec_dim) { recw[i] = rst[u]; if (i < OP) a.obs[(size_t)env * a.obs_ld + i] = rst[u]; }
        }
      }
    } else {
      FOR_G(i, nq) recw[i] = qpos[i];
      FOR_G(i, nv) { recw[nq + i] = qvel[i]; recw[OP + i] = warm[i]; }
      for (int i = O + g; i < OP; i += kGroupLanes) recw[i] = 0.f;
      for (int i = OP + nv + 2 + g; i < rec_dim; i += kGroupLanes) recw[i] = 0.f;
      if (g == 0) { recw[OP + nv] = new_comx; recw[OP + nv + 1] = time_in + dt_env; }
    }
  }
  if (valid && g == 0) {
    a.reward[env] = reward;
    a.done[env] = done ? 1 : 0;
    if (a.met.episode_returns) {
      const float nd = done ? 0.f : 1.f;
      const int ndi = done ? 0 : 1;
      const float new_ret = a.met.episode_returns[env] + reward;
      const int new_len = a.met.episode_lengths[env] + 1;
      a.met.episode_returns[env] = new_ret * nd;
      a.met.episode_lengths[env] = new_len * ndi;
      a.met.returned_episode_returns[env] = a.met.returned_episode_returns[env] * nd + new_ret * (done ? 1.f : 0.f);
      a.met.returned_episode_lengths[env] = a.met.returned_episode_lengths[env] * ndi + new_len * (done ? 1 : 0);
      a.met.timestep[env] = a.met.timestep[env] + 1;
      a.met.returned_episode[env] = done ? 1 : 0;
    }
  }
  PT(23);
}

}  // namespace mppo

// =================================================================================================
// host side: model handle + launch wrappers (C ABI)
// =================================================================================================
namespace mppo {
template <class SD>
static int32_t launch_env_t(const ModelView& mv, const EnvArgs& a, const PhysLds& lds, int lds_bytes, int blocks, int waves, hipStream_t stream) {
  void (*kern)(ModelView, EnvArgs, PhysLds) = a.mode == 0 ? &env_kernel<SD, 0> : a.mode == 1 ? &env_kernel<SD, 1> : &env_kernel<SD, 2>;
  // (the run-time-sized instantiation serves robots of different sizes: the limit follows the largest one seen)
  static thread_local int attr_lds[3] = {64 * 1024, 64 * 1024, 64 * 1024};
  if (lds_bytes > attr_lds[a.mode]) {
    MPPO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    attr_lds[a.mode] = lds_bytes;
  }
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * waves), lds_bytes, stream, mv, a, lds);
  MPPO_CHECK_LAUNCH("env_kernel");
  return MPPO_OK;
}

struct SpecEntry {
  BlobDims d;
  int32_t (*launch)(const ModelView&, const EnvArgs&, const PhysLds&, int, int, int, hipStream_t);
  bool extra;  // added by MPPO_SPECIALIZE for this build: not one of the instantiations the test suite holds bit-equal to the run-time-sized
               // kernel on both backends - mppo_model_open checks it against that kernel on the device before trusting it (spec_self_check)
};
#define MPPO_SPEC(...) {BlobDims{__VA_ARGS__}, &launch_env_t<StaticModel<__VA_ARGS__>>, false},
#define MPPO_SPEC_EXTRA(...) {BlobDims{__VA_ARGS__}, &launch_env_t<StaticModel<__VA_ARGS__>>, true},
// (MPPO_SPEC_INC: minppo_amd/jit.py compiles this file once more, device side only, with a list of ONE robot - the code object
// mppo_model_attach_kernel takes)
#ifndef MPPO_SPEC_INC
#define MPPO_SPEC_INC "spec_dims.inc"
#endif
static const SpecEntry kSpecs[] = {
#include MPPO_SPEC_INC
    {BlobDims{}, nullptr, false}};
#undef MPPO_SPEC
#undef MPPO_SPEC_EXTRA

// MPPO_ENV_GENERIC=1 forces the run-time-sized kernel (A/B tests of the two instantiations); so does MPPO_ENV_SPILL, which only the
// run-time-sized kernel can follow (a specialised kernel's choice is compiled in)
static int env_spill_override() {
  const char* e = getenv("MPPO_ENV_SPILL");
  if (!e || !e[0]) return -1;
  const int v = atoi(e);
  return v == 0 ? 0 : v == 1 ? kSpillJ : (kSpillJ | kSpillM);
}
static int find_spec(const BlobDims& d) {
  const char* e = getenv("MPPO_ENV_GENERIC");
  if (e && e[0] == '1') return -1;
  if (env_spill_override() >= 0) return -1;
  for (int i = 0; kSpecs[i].launch; ++i) {
    const BlobDims& s = kSpecs[i].d;
    if (s.nq == d.nq && s.nv == d.nv && s.nu == d.nu && s.nbody == d.nbody && s.njnt == d.njnt && s.ncon == d.ncon && s.nlimit == d.nlimit &&
        s.npair == d.npair && s.nlevel == d.nlevel && s.nroot == d.nroot && s.ncvx == d.ncvx && s.ncvxvert == d.ncvxvert && s.hull == d.hull && s.ncyl == d.ncyl)
      return i;
  }
  return -1;
}
}  // namespace mppo

struct mppo_model {
  mppo::ModelView mv;
  mppo::PhysLds lds;
  int lds_bytes;
  int waves;  // wavefronts per workgroup (mv.epw environments each: 4, fewer for a very large robot; one copy of the model tables per workgroup)
  int spec;  // index into the table of model-specialised kernels (spec_dims.inc), -1: the run-time-sized kernel
  // the records of the matrices a large robot keeps out of LDS (PhysLds::gwords floats per environment group of the grid), for launches
  // through mppo_env_reset / _step / mppo_physics_forward: owned by the handle, grown on demand (the engine passes a region of its arena
  // instead).  One stream at a time may launch through a handle that needs them.
  mutable float* scratch = nullptr;
  mutable size_t scratch_bytes = 0;
  int canon_words = 0;  // the table part's length as it follows from the dims (what a specialised kernel's compile-time layout choice saw)
  // a code object attached at run time (mppo_model_attach_kernel): the environment kernel compiled for exactly this robot's dimensions -
  // what MPPO_SPECIALIZE does at build time, for a robot the library was not built for
  bool jit = false;
  int jit_regchol = 0;  // the MPPO_REGCHOL_MAX_NV the code object was compiled with (its LDS layout follows from it)
  hipModule_t jit_module = nullptr;
  hipFunction_t jit_fn[3] = {nullptr, nullptr, nullptr};
  std::vector<char> jit_image;
};

namespace mppo {
const ModelView& model_view(const mppo_model* m) { return m->mv; }
// bytes of global memory the environment kernel needs beside the state for N environments (0 for a robot whose matrices fit LDS)
size_t model_scratch_bytes(const mppo_model* m, int N) {
  if (m->lds.gwords <= 0) return 0;
  const int per_block = m->mv.epw * m->waves;
  return (size_t)cdiv(N, per_block) * per_block * (size_t)m->lds.gwords * sizeof(float);
}
}

namespace mppo {
static int32_t launch_env(const mppo_model_t* m, EnvArgs a, hipStream_t stream, float* ws = nullptr, size_t ws_bytes = 0);

// LDS layout, matrices in global memory, environments per wave and waves per workgroup of a model whose `spec` is decided
static int32_t finalize_layout(mppo_model* m) {
  ModelView& v = m->mv;
  // (a model-specialised kernel of up to kRegCholMaxNv dofs keeps the inverse Cholesky factor in registers: no factor in its LDS layout; the
  // matrices that leave LDS for global memory - spill_for - are a function of the dims that the specialised kernel evaluated at compile time)
  // (MPPO_ENV_SPILL=0|1|3 overrides the choice - nothing, the Jacobian, the Jacobian and M in global memory - for A/B measurements and
  // for the test that holds the two placements bit-equal)
  auto lds_for = [&](bool li_regs) {
    const int forced = env_spill_override();
    return make_phys_lds(v.nq, v.nv, v.nu, v.nbody, v.njnt, v.ncon, v.nefc, v.nroot, v.ncvx, li_regs,
                         forced >= 0 ? forced : spill_for(v.nq, v.nv, v.nu, v.nbody, v.njnt, v.ncon, v.nefc, v.nroot, v.ncvx, li_regs, m->canon_words));
  };
  const bool fixed = m->spec >= 0 || m->jit;
  m->lds = lds_for(fixed && v.nv <= (m->jit ? m->jit_regchol : kRegCholMaxNv));
  // waves per workgroup: whatever puts the most waves on a CU (160 KB of LDS; every workgroup holds one copy of the model tables and
  // waves x 4 environments), the smaller workgroup on a tie.  MPPO_ENV_WAVES=1..4 overrides (measurements).
  // A robot too large for four environments per wave even with its matrices outside LDS runs two or one per wave on the
  // run-time-sized kernel - three quarters of the lanes idle, but it runs (round 5; before, it was refused).
  v.epw = kEnvsPerWave;
  if (fixed && ((long long)v.blob_words + (long long)m->lds.total * kEnvsPerWave) * 4 > 160 * 1024) {
    // (a specialised kernel carries four environments per wave; a robot too large for that runs the run-time-sized kernel with fewer)
    m->spec = -1;
    m->jit = false;
    m->lds = lds_for(false);
  }
  auto lds_of = [&](int w) { return (int)std::min<long long>(((long long)v.blob_words + (long long)m->lds.total * v.epw * w) * 4, 1 << 30); };
  while (lds_of(1) > 160 * 1024 && m->spec < 0 && !m->jit && v.epw > 1) v.epw /= 2;
  int best = 1, best_per_cu = 0;
  for (int w = 1; w <= kMaxWavesPerBlock; ++w) {
    const int per_cu = lds_of(w) <= 160 * 1024 ? (160 * 1024 / lds_of(w)) * w : 0;
    if (per_cu > best_per_cu) { best = w; best_per_cu = per_cu; }
  }
  if (const char* e = getenv("MPPO_ENV_WAVES")) { const int w = atoi(e); if (w >= 1 && w <= kMaxWavesPerBlock) best = w; }
  m->waves = best;
  m->lds_bytes = lds_of(best);
  if (m->lds_bytes > 160 * 1024) return fail(MPPO_EMODEL, "model needs %d bytes of LDS per workgroup for ONE environment (limit 163840)", m->lds_bytes);
  return MPPO_OK;
}

// A kernel instantiation that MPPO_SPECIALIZE added to this build has never been compared with anything: before it is trusted, a reset and
// four steps of 24 environments under pseudo-random controls must equal the run-time-sized kernel's bit for bit ON THIS DEVICE.  If they do
// not (round 6: a 34-dof / 93-body robot's instantiation, 250 spilled registers, ended every episode at its first step on the GPU while the
// same source was right on the emulator - a per-lane flag spilled inside divergent code: `bad_mid` in env_kernel says how it ended), the model runs the run-time-sized kernel and
// says so on stderr.  A few milliseconds at mppo_model_open; the BASELINE instantiations are held to the same standard by the test suite.
// Work on the device the model's tables are on, whatever the calling thread's current device is (restored on the way out)
struct OnDeviceOf {
  int cur = -1, dev = -1;
  hipError_t err = hipSuccess;
  explicit OnDeviceOf(const void* p) {
#ifndef MPPO_EMU
    hipPointerAttribute_t attr{};
    err = hipGetDevice(&cur);
    if (err == hipSuccess) err = hipPointerGetAttributes(&attr, p);
    if (err == hipSuccess) { dev = attr.device; if (dev != cur) err = hipSetDevice(dev); }
#else
    (void)p;
#endif
  }
  ~OnDeviceOf() {
#ifndef MPPO_EMU
    if (dev != cur && dev >= 0 && cur >= 0) (void)hipSetDevice(cur);
#endif
  }
};
static int32_t spec_self_check_on_device(mppo_model* m);
static int32_t spec_self_check(mppo_model* m) {
  OnDeviceOf where(m->mv.blob);
  MPPO_CHECK_HIP(where.err);
  return spec_self_check_on_device(m);
}
static int32_t spec_self_check_on_device(mppo_model* m) {
  const ModelView& v = m->mv;
  const int N = 24, steps = 4, nu = v.nu > 0 ? v.nu : 1;
  const size_t nstate = (size_t)N * v.rec_dim, nobs = (size_t)N * v.obs_pad, nact = (size_t)N * nu;
  const size_t words = nstate + v.rec_dim + nobs + (size_t)steps * nact + N + N;  // state, reset record, observation, controls, reward, done
  float* dev = nullptr;
  MPPO_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&dev), words * sizeof(float)));
  std::vector<float> act((size_t)steps * nact), got[2];
  unsigned lcg = 12345u;
  for (float& x : act) { lcg = lcg * 1664525u + 1013904223u; x = ((lcg >> 8) & 0xffff) / 32768.f - 1.f; }
  float *state = dev, *reset_rec = state + nstate, *obs = reset_rec + v.rec_dim, *actd = obs + nobs, *rew = actd + (size_t)steps * nact;
  unsigned char* done = reinterpret_cast<unsigned char*>(rew + N);
  mppo_reward_cfg_t rc{};
  rc.height_min_z = -1e9f; rc.height_max_z = 1e9f;
  mppo_model generic = *m;
  generic.spec = -1; generic.jit = false; generic.scratch = nullptr; generic.scratch_bytes = 0;
  int32_t st = finalize_layout(&generic);
  for (int which = 0; which < 2 && st == MPPO_OK; ++which) {
    const mppo_model* mm = which == 0 ? m : &generic;
    hipError_t he = hipMemset(dev, 0, words * sizeof(float));
    if (he == hipSuccess) he = hipMemcpy(actd, act.data(), act.size() * sizeof(float), hipMemcpyHostToDevice);
    if (he != hipSuccess) { st = fail(MPPO_EHIP, "specialised-kernel self-check: %s", hipGetErrorString(he)); break; }
    EnvArgs a{};
    a.N = N; a.mode = 0; a.n_frames = 1; a.state = state; a.reset_out = reset_rec; a.obs = obs; a.obs_ld = v.obs_pad; a.reward = rew; a.done = done;
    st = launch_env(mm, a, nullptr);
    for (int t = 0; t < steps && st == MPPO_OK; ++t) {
      EnvArgs b{};
      b.N = N; b.mode = 1; b.n_frames = 1; b.state = state; b.reset_in = reset_rec; b.action = actd + (size_t)t * nact; b.act_ld = nu;
      b.obs = obs; b.obs_ld = v.obs_pad; b.reward = rew; b.done = done; b.rc = rc;
      st = launch_env(mm, b, nullptr);
    }
    if (st != MPPO_OK) break;
    got[which].resize(words);
    he = hipDeviceSynchronize();
    if (he == hipSuccess) he = hipMemcpy(got[which].data(), dev, words * sizeof(float), hipMemcpyDeviceToHost);
    if (he != hipSuccess) st = fail(MPPO_EHIP, "specialised-kernel self-check: %s", hipGetErrorString(he));
  }
  (void)hipFree(dev);
  if (generic.scratch) (void)hipFree(generic.scratch);
  if (st != MPPO_OK) return st;
  // (the controls are the same bytes in both; everything else is the kernels' output)
  if (memcmp(got[0].data(), got[1].data(), words * sizeof(float)) != 0) {
    size_t bad = 0;
    for (size_t i = 0; i < words; ++i) bad += memcmp(&got[0][i], &got[1][i], 4) != 0;
    fprintf(stderr, "minppo_amd: the environment kernel specialised for this robot (nv %d, %d bodies, %d contact slots) differs from the run-time-sized kernel in %zu of %zu "
                    "words after a reset and %d steps of %d environments on this device: NOT used - the run-time-sized kernel runs instead.  (DESIGN.md 3.3 has the one such kernel met so far; "
                    "minppo_amd/build.py names the build variable that keeps a specialised kernel's factorisation out of registers.)\n", v.nv, v.nbody, v.ncon, bad, words, steps, N);
    if (m->scratch) { (void)hipFree(m->scratch); m->scratch = nullptr; m->scratch_bytes = 0; }
    m->spec = -1;
    m->jit = false;
    return finalize_layout(m);
  }
  return MPPO_OK;
}
}  // namespace mppo

extern "C" int32_t mppo_model_open(const void* host_blob, size_t nbytes, const void* dev_blob, mppo_model_t** out) {
  using namespace mppo;
  if (!host_blob || !dev_blob || !out) return fail(MPPO_EINVAL, "mppo_model_open: null argument");
  if (nbytes < 4 * (size_t)kBlobHeaderWords || (nbytes & 3)) return fail(MPPO_EMODEL, "model blob too small or not word-sized (%zu bytes)", nbytes);
  if ((reinterpret_cast<uintptr_t>(dev_blob) & 15) != 0) return fail(MPPO_EINVAL, "device blob must be 16-byte aligned");
  const uint32_t* w = static_cast<const uint32_t*>(host_blob);
  const int32_t* wi = static_cast<const int32_t*>(host_blob);
  const float* wf = static_cast<const float*>(host_blob);
  if (w[0] != kBlobMagic) return fail(MPPO_EMODEL, "bad model blob magic 0x%08x", w[0]);
  if (w[1] != kBlobVersion) return fail(MPPO_EMODEL, "unsupported model blob version %u", w[1]);
  const size_t total = w[2], hull_words = w[35];  // table part + hull section
  if ((total + hull_words) * 4 != nbytes) return fail(MPPO_EMODEL, "model blob size mismatch: header says %zu + %zu words, got %zu bytes", total, hull_words, nbytes);
  if (wi[32] != BLOB_ARRAY_COUNT) return fail(MPPO_EMODEL, "model blob has %d arrays, engine expects %d", wi[32], (int)BLOB_ARRAY_COUNT);
  mppo_model* m = new mppo_model();
  ModelView& v = m->mv;
  v.nq = wi[3]; v.nv = wi[4]; v.nu = wi[5]; v.nbody = wi[6]; v.njnt = wi[7]; v.ncon = wi[8]; v.nlimit = wi[9];
  v.iterations = wi[10]; v.ls_iterations = wi[11]; v.nlevel = wi[12]; v.nroot = wi[13]; v.include_c = wi[14] ? 1 : 0; v.npair = wi[15];
  v.ncvx = wi[33]; v.ncvxvert = wi[34];
  // every header dimension inside a bound that keeps the size arithmetic below (and in blob_offsets) far from overflow, BEFORE any of it
  // is computed (tests/test_blob_fuzz.py under UBSan: a dimension of INT_MAX overflowed `4 * ncon` here)
  for (int d : {v.nq, v.nv, v.nu, v.nbody, v.njnt, v.ncon, v.nlimit, v.iterations, v.ls_iterations, v.nlevel, v.nroot, v.npair, v.ncvx, v.ncvxvert, wi[36]})
    if (d < 0 || d > (1 << 16)) { delete m; return fail(MPPO_EMODEL, "model blob: header dimension %d out of range", d); }
  v.nefc = v.nlimit + 4 * v.ncon;
  v.timestep = wf[16]; v.tolerance = wf[17]; v.ls_tolerance = wf[18]; v.impratio = wf[19]; v.plane_z = wf[20]; v.meaninertia = wf[21];
  auto bad = [&](const char* what) { delete m; return fail(MPPO_EMODEL, "model blob: %s", what); };
  if (v.nq < 1 || v.nv < 1 || v.nbody < 2 || v.nbody > 128 || v.nv > 64 || v.nq > 128 || v.nu < 0 || v.nu > v.nv || v.njnt < 1 ||
      v.ncon < 0 || v.npair < 0 || v.npair > v.ncon || v.nlimit < 0 || v.nroot < 1 || v.nlevel < 1 || v.iterations < 0 || v.ls_iterations < 0 ||
      v.ncvx < 0 || 4 * v.ncvx > v.ncon - v.npair || v.ncvxvert < 4 * v.ncvx || v.ncvxvert > 64 * 64)
    return bad("dimension out of the supported range (nbody<=128, nv<=64)");
  if (!(v.timestep > 0.f) || !(v.meaninertia > 0.f) || !(v.impratio > 0.f)) return bad("non-positive timestep / meaninertia / impratio");
  const int32_t* dir = wi + kBlobHeaderWords;
  BlobDims bd{v.nq, v.nv, v.nu, v.nbody, v.njnt, v.ncon, v.nlimit, v.npair, v.nlevel, v.nroot, v.ncvx, v.ncvxvert, 0, 0};
  const BlobOffsets canon = blob_offsets(bd);
  const size_t dir_end = kBlobHeaderWords + 2 * (size_t)BLOB_ARRAY_COUNT;
  if (dir_end > total) return bad("directory past the end");
  for (int k = 0; k < BLOB_ARRAY_COUNT; ++k) {
    const long off = dir[2 * k], cnt = dir[2 * k + 1];
    if (off < (long)dir_end || cnt < 0 || (size_t)(off + cnt) > total || (off & 3)) return bad("array directory entry out of range");
    if (cnt != blob_array_len(bd, k)) { delete m; return fail(MPPO_EMODEL, "model blob: array %d has %ld entries, expected %d", k, cnt, blob_array_len(bd, k)); }
    if (off != canon.o[k]) { delete m; return fail(MPPO_EMODEL, "model blob: array %d sits at word %ld, canonical placement is %d", k, off, canon.o[k]); }
  }
  auto HI = [&](int k) { return wi + dir[2 * k]; };
  // index tables are validated here so that the kernel never dereferences out of range
  auto in_range = [&](int k, long lo, long hi_excl) {
    const int32_t* p = HI(k);
    for (long i = 0; i < dir[2 * k + 1]; ++i) if (p[i] < lo || p[i] >= hi_excl) return false;
    return true;
  };
  if (!in_range(BI_body_parent, 0, v.nbody) || !in_range(BI_body_rootid, 0, v.nbody) || !in_range(BI_jnt_bodyid, 1, v.nbody) ||
      !in_range(BI_jnt_qposadr, 0, v.nq) || !in_range(BI_jnt_dofadr, 0, v.nv) || !in_range(BI_dof_bodyid, 1, v.nbody) ||
      !in_range(BI_dof_jntid, 0, v.njnt) || !in_range(BI_dof_parentid, -1, v.nv) || !in_range(BI_dof_qposadr, -1, v.nq) ||
      !in_range(BI_act_dofid, 0, v.nv) || !in_range(BI_act_qposadr, 0, v.nq) || !in_range(BI_con_bodyid, 1, v.nbody) || !in_range(BI_pair_body, 1, v.nbody) ||
      !in_range(BI_lim_jntid, 0, v.njnt) || !in_range(BI_level_adr, 0, v.nbody) || !in_range(BI_level_body, 1, v.nbody) ||
      !in_range(BI_root_body, 1, v.nbody) || !in_range(BI_body_jntnum, 0, v.njnt + 1) || !in_range(BI_body_jntadr, -1, v.njnt) ||
      !in_range(BI_con_cvx, -4, 4 * v.ncvx) || !in_range(BI_cvx_body, 1, v.nbody) || !in_range(BI_cvx_vadr, 0, v.ncvxvert + 1))
    return bad("index table entry out of range");
  {
    const int32_t* va = HI(BI_cvx_vadr);
    for (int k = 0; k < v.ncvx; ++k) if (va[k + 1] < va[k] + 4 || va[k + 1] - va[k] > 64) return bad("a convex geom needs 4 .. 64 hull vertices");
    if (v.ncvx > 0 && (va[0] != 0 || va[v.ncvx] != v.ncvxvert)) return bad("cvx_vadr does not cover the vertex table");
    // cylinders: slots -2, -3, -4 in a row among the ground contacts, a half-axis vector of non-zero length in the first
    const int32_t* kind = HI(BI_con_cvx);
    const float* cax = wf + dir[2 * BF_con_axis];
    int ncyl = 0;
    for (int c = 0; c < v.ncon; ++c) {
      if (kind[c] > -2) continue;
      if (c >= v.ncon - v.npair) return bad("a cylinder slot among the pair contacts");
      if (kind[c] == -2) {
        if (c + 2 >= v.ncon - v.npair || kind[c + 1] != -3 || kind[c + 2] != -4) return bad("a cylinder needs three consecutive ground-contact slots");
        if (!(cax[3 * c] * cax[3 * c] + cax[3 * c + 1] * cax[3 * c + 1] + cax[3 * c + 2] * cax[3 * c + 2] > 0.f)) return bad("a cylinder with a zero half-axis");
        ++ncyl;
      } else if (c == 0 || kind[c - 1] != kind[c] + 1) return bad("a cylinder's second / third slot without its first");
    }
    if (ncyl != wi[36]) return bad("header ncyl does not match the contact table");
    v.ncyl = ncyl;
  }
  {
    // the hull section: every index the kernel follows from a pair row to a hull, its faces, their vertex lists and its edges
    v.hull_words = (int)hull_words;
    const int32_t* hs = wi + total;
    HullView hv{};
    if (hull_words > 0) {
      if (hull_words < 8 || (hull_words & 3)) return bad("hull section too short");
      if (hs[0] < 1 || hs[1] < 4 || hs[2] < 4 || hs[3] < 12 || hs[4] < 6 || hs[0] > 64 || hs[1] > 64 * 64 || hs[2] > 128 * 64 || hs[3] > 6 * 128 * 64 || hs[4] > 192 * 64)
        return bad("hull section: dimension out of range");
      if (hs[5] < 3 * hs[0] || hs[5] > hs[4]) return bad("hull section: number of edge directions out of range");
      hv = hull_view(hs[0], hs[1], hs[2], hs[3], hs[4], hs[5]);
      if ((size_t)hv.words != hull_words) return bad("hull section: length does not follow from its dimensions");
      {
        const int32_t* ua = hs + hv.udadr;
        if (ua[0] != 0 || ua[hv.nhull] != hv.nudir) return bad("hull section: edge-direction ranges do not cover their array");
        for (int h = 0; h < hv.nhull; ++h) if (ua[h + 1] < ua[h] + 3) return bad("a hull needs at least three edge directions");
      }
      const int32_t *va = hs + hv.vadr, *fa = hs + hv.fadr, *ea = hs + hv.eadr, *pa = hs + hv.face_adr, *fi = hs + hv.fidx, *ed = hs + hv.edge;
      if (va[0] != 0 || fa[0] != 0 || ea[0] != 0 || pa[0] != 0 || va[hv.nhull] != hv.nvert || fa[hv.nhull] != hv.nface || ea[hv.nhull] != hv.nedge || pa[hv.nface] != hv.nfidx)
        return bad("hull section: address tables do not cover their arrays");
      for (int h = 0; h < hv.nhull; ++h) {
        if (va[h + 1] < va[h] + 4 || va[h + 1] - va[h] > 64 || fa[h + 1] < fa[h] + 4 || ea[h + 1] < ea[h] + 6) return bad("a hull needs 4 .. 64 vertices, at least 4 faces and 6 edges");
        for (int f = fa[h]; f < fa[h + 1]; ++f) {
          if (pa[f + 1] < pa[f] + 3 || pa[f + 1] - pa[f] > 64) return bad("a hull face needs 3 .. 64 vertices");
          for (int i = pa[f]; i < pa[f + 1]; ++i) if (fi[i] < va[h] || fi[i] >= va[h + 1]) return bad("hull face vertex out of its hull's range");
        }
        for (int e = 2 * ea[h]; e < 2 * ea[h + 1]; ++e) if (ed[e] < va[h] || ed[e] >= va[h + 1]) return bad("hull edge vertex out of its hull's range");
      }
    }
    const float* pg = wf + dir[2 * BF_pair_geom];
    auto v_pair_body = [&](int k) { const int32_t* pb = HI(BI_pair_body); return ((long long)pb[2 * k] << 32) | (long long)(unsigned)pb[2 * k + 1]; };
    for (int k = 0; k < v.npair; ++k) {
      const float* row = pg + 16 * k;
      const float hid = row[7], slot = row[15];
      // a hull pair (box / mesh against box / mesh, four slots): geom 1 carries no shape of its own and geom 2's radius word names geom 1's hull
      const bool hullpair = hid != 0.f && row[14] != 0.f && row[3] == 0.f && row[4] == 0.f && row[5] == 0.f && row[6] == 0.f;
      if (hid != (float)(int)hid || hid < 0.f || hid > (float)hv.nhull || slot != (float)(int)slot || slot < 0.f || slot > (hullpair ? 3.f : 1.f))
        return bad("pair row: hull / slot tag out of range");
      if (hullpair) {
        const float h1 = row[14];
        if (h1 != (float)(int)h1 || h1 < 1.f || h1 > (float)hv.nhull || h1 == hid) return bad("pair row: a hull pair's first hull out of range");
        if (slot == 0.f && k + 3 >= v.npair) return bad("pair row: a hull pair needs four consecutive slots");
        // (the manifold's candidates are kept four to a lane: faces of at most 16 vertices on either side)
        const int32_t *fa = wi + total + hv.fadr, *pa = wi + total + hv.face_adr;
        for (int hh : {(int)h1 - 1, (int)hid - 1})
          for (int f = fa[hh]; f < fa[hh + 1]; ++f) if (pa[f + 1] - pa[f] > 16) return bad("a hull in a hull pair has a face of more than 16 vertices");
      }
      if (slot >= 1.f && (hid == 0.f || k == 0 || pg[16 * (k - 1) + 7] != hid || pg[16 * (k - 1) + 15] != slot - 1.f || pg[16 * (k - 1) + 14] != row[14] ||
                          v_pair_body(k) != v_pair_body(k - 1)))
        return bad("pair row: a later slot must follow its pair's previous one");
    }
  }
  {
    const int32_t *jt = HI(BI_jnt_type), *qa = HI(BI_jnt_qposadr), *da = HI(BI_jnt_dofadr), *jn = HI(BI_body_jntnum), *ja = HI(BI_body_jntadr),
                  *par = HI(BI_body_parent), *dp = HI(BI_dof_parentid), *la = HI(BI_level_adr);
    for (int j = 0; j < v.njnt; ++j) {
      if (jt[j] != JNT_FREE && jt[j] != JNT_HINGE && jt[j] != JNT_SLIDE) return bad("unsupported joint type");
      if (jt[j] == JNT_FREE && (qa[j] + 7 > v.nq || da[j] + 6 > v.nv)) return bad("free joint address out of range");
    }
    for (int b = 1; b < v.nbody; ++b) {
      if (par[b] >= b) return bad("bodies are not topologically ordered");
      if (jn[b] > 0 && (ja[b] < 0 || ja[b] + jn[b] > v.njnt)) return bad("body joint range out of bounds");
    }
    for (int d = 0; d < v.nv; ++d) if (dp[d] >= d) return bad("dof_parentid must point to an earlier dof");
    for (int l = 0; l < v.nlevel; ++l) if (la[l + 1] < la[l] || la[l + 1] > v.nbody - 1) return bad("level_adr not monotone");
    if (HI(BI_root_body)[0] != 1) return bad("body 1 must be the first tree root");
  }
  v.blob = static_cast<const int32_t*>(dev_blob);
  v.blob_words = (int)((total + 3) & ~(size_t)3);
  if ((size_t)v.blob_words != total) return bad("blob length must be a multiple of 4 words");
  for (int k = 0; k < BLOB_ARRAY_COUNT; ++k) v.o[k] = dir[2 * k];
  v.obs_dim = v.nq + 2 * v.nv + (v.include_c ? 16 * (v.nbody - 1) : 0);  // env.py:246-259
  v.obs_pad = (v.obs_dim + 3) & ~3;
  v.rec_dim = v.obs_pad + ((v.nv + 2 + 3) & ~3);
  bd.hull = v.hull_words > 0 ? 1 : 0; bd.ncyl = v.ncyl;
  m->spec = find_spec(bd);
  m->canon_words = canon.words;
  if (int32_t rc = mppo::finalize_layout(m); rc != MPPO_OK) { delete m; return rc; }
  // every specialised instantiation proves itself on the device it is about to run on (a few milliseconds): the ones a build adds (MPPO_SPECIALIZE) always,
  // the default ones too on hardware - the test suite holds them bit-equal on the builder's toolchain, a user's compiler is another one (on the emulator the
  // suite itself is the check)
#ifdef MPPO_EMU
  const bool check = m->spec >= 0 && kSpecs[m->spec].extra;
#else
  const bool check = m->spec >= 0;
#endif
  if (check) {
    if (int32_t rc = mppo::spec_self_check(m); rc != MPPO_OK) { if (m->scratch) (void)hipFree(m->scratch); delete m; return rc; }
  }
  *out = m;
  return MPPO_OK;
}

#ifdef MPPO_PHYS_TIMERS
extern "C" int32_t mppo_debug_phys_timers(unsigned long long* out40) {
  MPPO_CHECK_HIP(hipMemcpyFromSymbol(out40, HIP_SYMBOL(mppo::g_phys_t), sizeof(unsigned long long) * 40));
  return MPPO_OK;
}
#endif

extern "C" int32_t mppo_model_close(mppo_model_t* m) {
  if (m && m->scratch) (void)hipFree(m->scratch);
  if (m && m->jit_module) (void)hipModuleUnload(m->jit_module);
  delete m;
  return MPPO_OK;
}

// The tag a code object of this file carries (`mppo_env_kernel_tag`): the sizes of the three kernel-argument structs and the blob version -
// what has to agree between the library and a code object compiled apart from it for a launch to mean anything.
namespace mppo {
constexpr unsigned kEnvKernelTag = (unsigned)sizeof(ModelView) * 2654435761u ^ (unsigned)sizeof(EnvArgs) * 40503u ^ (unsigned)sizeof(PhysLds) * 2246822519u ^ kBlobVersion * 3266489917u ^
                                   (unsigned)kEnvsPerWave;
}
extern "C" __device__ __attribute__((used)) const unsigned mppo_env_kernel_tag = mppo::kEnvKernelTag;

extern "C" int32_t mppo_model_attach_kernel(mppo_model_t* m, const void* image, size_t nbytes, const char* const* names, int32_t regchol_max_nv, int32_t* used) {
  using namespace mppo;
  if (!m || !image || !nbytes || !names || !names[0] || !names[1] || !names[2] || !used) return fail(MPPO_EINVAL, "mppo_model_attach_kernel: null argument");
  *used = 0;
  if (m->jit) return fail(MPPO_EINVAL, "mppo_model_attach_kernel: a code object is attached to this model already");
  if (regchol_max_nv < 0 || regchol_max_nv > 64) return fail(MPPO_EINVAL, "mppo_model_attach_kernel: regchol_max_nv %d", regchol_max_nv);
  if (m->spec >= 0) return MPPO_OK;  // (the library holds this robot's kernel itself)
  {
    const char* e = getenv("MPPO_ENV_GENERIC");
    if ((e && e[0] == '1') || env_spill_override() >= 0) return MPPO_OK;  // (the switches that force the run-time-sized kernel)
  }
  // the kernels' names spell the dimensions they were compiled for: StaticModel<nq, nv, nu, nbody, njnt, ncon, nlimit, npair, nlevel, nroot, ncvx, ncvxvert, hull, ncyl>, MODE
  const ModelView& v = m->mv;
  const int dims[14] = {v.nq, v.nv, v.nu, v.nbody, v.njnt, v.ncon, v.nlimit, v.npair, v.nlevel, v.nroot, v.ncvx, v.ncvxvert, v.hull_words > 0 ? 1 : 0, v.ncyl};
  char want[256];
  int o = snprintf(want, sizeof want, "StaticModelI");
  for (int d : dims) o += snprintf(want + o, sizeof want - o, "Li%dE", d);
  for (int k = 0; k < 3; ++k) {
    char mode[320];
    snprintf(mode, sizeof mode, "%sEELi%dEEEv", want, k);
    if (!strstr(names[k], "env_kernel") || !strstr(names[k], mode))
      return fail(MPPO_EINVAL, "mppo_model_attach_kernel: kernel %d is named %s - not the environment kernel of this robot's dimensions and mode (%s)", k, names[k], mode);
  }
  OnDeviceOf where(m->mv.blob);  // (a module belongs to the device it was loaded on)
  MPPO_CHECK_HIP(where.err);
  m->jit_image.assign(static_cast<const char*>(image), static_cast<const char*>(image) + nbytes);
  hipModule_t mod = nullptr;
  hipError_t he = hipModuleLoadData(&mod, m->jit_image.data());
  if (he != hipSuccess) { m->jit_image.clear(); return fail(MPPO_EHIP, "mppo_model_attach_kernel: the code object does not load (%s)", hipGetErrorString(he)); }
  auto drop = [&](int32_t rc) { (void)hipModuleUnload(mod); m->jit_module = nullptr; m->jit = false; m->jit_image.clear(); m->jit_image.shrink_to_fit(); return rc; };
  hipDeviceptr_t tag_ptr = nullptr;
  size_t tag_bytes = 0;
  unsigned tag = 0;
  he = hipModuleGetGlobal(&tag_ptr, &tag_bytes, mod, "mppo_env_kernel_tag");
  if (he == hipSuccess && tag_bytes == sizeof tag) he = hipMemcpy(&tag, tag_ptr, sizeof tag, hipMemcpyDeviceToHost);
  if (he != hipSuccess || tag_bytes != sizeof tag) return drop(fail(MPPO_EINVAL, "mppo_model_attach_kernel: the code object carries no mppo_env_kernel_tag (%s)", hipGetErrorString(he)));
  if (tag != kEnvKernelTag) return drop(fail(MPPO_EINVAL, "mppo_model_attach_kernel: the code object was compiled from other kernel sources than this library (tag %08x, library %08x)", tag, kEnvKernelTag));
  for (int k = 0; k < 3; ++k) {
    he = hipModuleGetFunction(&m->jit_fn[k], mod, names[k]);
    if (he != hipSuccess) return drop(fail(MPPO_EINVAL, "mppo_model_attach_kernel: no kernel %s in the code object (%s)", names[k], hipGetErrorString(he)));
  }
  // the layout the specialised kernel computed for itself at compile time; then the same proof a build-time instantiation gives
  m->jit_module = mod;
  m->jit = true;
  m->jit_regchol = regchol_max_nv;
  if (m->scratch) { MPPO_CHECK_HIP(hipDeviceSynchronize()); (void)hipFree(m->scratch); m->scratch = nullptr; m->scratch_bytes = 0; }
  int32_t rc = finalize_layout(m);
  if (rc == MPPO_OK && m->jit) rc = spec_self_check(m);
  if (rc != MPPO_OK) { m->jit = false; (void)finalize_layout(m); return drop(rc); }
  if (!m->jit) return drop(MPPO_OK);  // (too large for four environments per wave, or it failed the check: the run-time-sized kernel stays)
  *used = 1;
  return MPPO_OK;
}

extern "C" int32_t mppo_model_scratch_bytes(const mppo_model_t* m, int32_t N, size_t* out) {
  if (!m || !out || N < 1) return mppo::fail(MPPO_EINVAL, "mppo_model_scratch_bytes: null argument or N < 1");
  *out = mppo::model_scratch_bytes(m, N);
  return MPPO_OK;
}

extern "C" int32_t mppo_model_get_dims(const mppo_model_t* m, mppo_model_dims_t* o) {
  if (!m || !o) return mppo::fail(MPPO_EINVAL, "mppo_model_get_dims: null argument");
  const mppo::ModelView& v = m->mv;
  o->nq = v.nq; o->nv = v.nv; o->nu = v.nu; o->nbody = v.nbody; o->njnt = v.njnt; o->ncon = v.ncon; o->nlimit = v.nlimit; o->nefc = v.nefc;
  o->obs_dim = v.obs_dim; o->obs_pad = v.obs_pad; o->rec_dim = v.rec_dim; o->lds_bytes = m->lds_bytes; o->timestep = v.timestep;
  return MPPO_OK;
}

namespace mppo {
// `ws`: the caller's region for the out-of-LDS matrices (the engine's arena), or null: the handle's own allocation, grown on demand
static int32_t launch_env(const mppo_model_t* m, EnvArgs a, hipStream_t stream, float* ws, size_t ws_bytes) {
  const int blocks = cdiv(a.N, m->mv.epw * m->waves);
  const size_t need = model_scratch_bytes(m, a.N);
  if (need > 0) {
    if (ws) {
      if (ws_bytes < need) return fail(MPPO_EINVAL, "environment kernel: the caller's scratch region holds %zu bytes, %d environments need %zu", ws_bytes, a.N, need);
      a.scratch = ws;
    } else {
      if (m->scratch_bytes < need) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (stream && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
          return fail(MPPO_EINVAL, "environment kernel: this robot keeps %zu bytes of matrices in global memory for %d environments and the handle's allocation would have to grow inside a stream capture: launch once outside the capture first", need, a.N);
        if (m->scratch) { MPPO_CHECK_HIP(hipDeviceSynchronize()); (void)hipFree(m->scratch); m->scratch = nullptr; m->scratch_bytes = 0; }
        MPPO_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&m->scratch), need));
        m->scratch_bytes = need;
      }
      a.scratch = m->scratch;
    }
  }
  if (m->jit) {
    ModelView mv = m->mv;
    PhysLds lds = m->lds;
    void* params[] = {&mv, &a, &lds};
    MPPO_CHECK_HIP(hipModuleLaunchKernel(m->jit_fn[a.mode], blocks, 1, 1, 64 * m->waves, 1, 1, m->lds_bytes, stream, params, nullptr));
    return MPPO_OK;
  }
#ifdef MPPO_JIT_ONLY  // (the device-side compile of minppo_amd/jit.py: one robot's instantiation and nothing else)
  return kSpecs[0].launch(m->mv, a, m->lds, m->lds_bytes, blocks, m->waves, stream);
#else
  return (m->spec >= 0 ? kSpecs[m->spec].launch : &launch_env_t<RuntimeModel>)(m->mv, a, m->lds, m->lds_bytes, blocks, m->waves, stream);
#endif
}
// the engine's entry: mppo_env_step with the out-of-LDS matrices in a region of the engine's arena (hipGraph capture: nothing is allocated)
int32_t env_step_ws(const mppo_model_t* m, int32_t N, int32_t n_frames, const mppo_reward_cfg_t* rc, float* state, const float* reset_rec, const float* action,
                    int32_t act_ld, float* obs, int32_t obs_ld, float* reward, uint8_t* done, const mppo_env_metrics_t* metrics, float* ws, size_t ws_bytes, hipStream_t stream) {
  EnvArgs a{};
  a.N = N; a.mode = 1; a.n_frames = n_frames; a.state = state; a.reset_in = reset_rec; a.action = action; a.act_ld = act_ld;
  a.obs = obs; a.obs_ld = obs_ld; a.reward = reward; a.done = done; a.rc = *rc;
  if (metrics) a.met = *metrics;
  return launch_env(m, a, stream, ws, ws_bytes);
}
int32_t env_reset_ws(const mppo_model_t* m, int32_t N, float* state, float* reset_rec, float* obs, int32_t obs_ld, const mppo_env_metrics_t* metrics, float* ws, size_t ws_bytes,
                     hipStream_t stream) {
  EnvArgs a{};
  a.N = N; a.mode = 0; a.n_frames = 1; a.state = state; a.reset_out = reset_rec; a.obs = obs; a.obs_ld = obs_ld;
  if (metrics) a.met = *metrics;
  return launch_env(m, a, stream, ws, ws_bytes);
}
}  // namespace mppo

extern "C" int32_t mppo_model_is_specialized(const mppo_model_t* m, int32_t* out) {
  if (!m || !out) return mppo::fail(MPPO_EINVAL, "mppo_model_is_specialized: null argument");
  *out = m->spec >= 0 ? 1 : m->jit ? 2 : 0;
  return MPPO_OK;
}

extern "C" int32_t mppo_env_reset(const mppo_model_t* m, int32_t N, float* state, float* reset_rec, float* obs, int32_t obs_ld,
                                  float* reward, uint8_t* done, const mppo_env_metrics_t* metrics, void* stream) {
  using namespace mppo;
  MPPO_REQUIRE(m && state && reset_rec, "mppo_env_reset: null model / state / reset_rec");
  MPPO_REQUIRE(N >= 1, "mppo_env_reset: N = %d", N);
  MPPO_REQUIRE(m->mv.nq >= 3, "mppo_env_reset: the environment reads qpos[2] as the height (env.py:239); nq = %d", m->mv.nq);
  MPPO_REQUIRE(!obs || obs_ld >= m->mv.obs_pad, "mppo_env_reset: obs_ld %d < padded observation width %d", obs_ld, m->mv.obs_pad);
  EnvArgs a{};
  a.N = N; a.mode = 0; a.n_frames = 1; a.state = state; a.reset_out = reset_rec; a.obs = obs; a.obs_ld = obs_ld; a.reward = reward; a.done = done;
  if (metrics) a.met = *metrics;
  return launch_env(m, a, static_cast<hipStream_t>(stream));
}

extern "C" int32_t mppo_env_step(const mppo_model_t* m, int32_t N, int32_t n_frames, const mppo_reward_cfg_t* rc, float* state,
                                 const float* reset_rec, const float* action, int32_t act_ld, float* obs, int32_t obs_ld, float* reward,
                                 uint8_t* done, const mppo_env_metrics_t* metrics, void* stream) {
  using namespace mppo;
  MPPO_REQUIRE(m && rc && state && reset_rec && action && obs && reward && done, "mppo_env_step: null argument");
  MPPO_REQUIRE(N >= 1 && n_frames >= 1, "mppo_env_step: N = %d, n_frames = %d", N, n_frames);
  MPPO_REQUIRE(m->mv.nq >= 3, "mppo_env_step: the environment reads qpos[2] as the height (env.py:239); nq = %d", m->mv.nq);
  MPPO_REQUIRE(act_ld >= m->mv.nu, "mppo_env_step: act_ld %d < nu %d", act_ld, m->mv.nu);
  MPPO_REQUIRE(obs_ld >= m->mv.obs_pad, "mppo_env_step: obs_ld %d < padded observation width %d", obs_ld, m->mv.obs_pad);
  EnvArgs a{};
  a.N = N; a.mode = 1; a.n_frames = n_frames; a.state = state; a.reset_in = reset_rec; a.action = action; a.act_ld = act_ld;
  a.obs = obs; a.obs_ld = obs_ld; a.reward = reward; a.done = done; a.rc = *rc;
  if (metrics) a.met = *metrics;
  return launch_env(m, a, static_cast<hipStream_t>(stream));
}

extern "C" int32_t mppo_physics_forward(const mppo_model_t* m, int32_t N, const float* qpos, const float* qvel, const float* ctrl,
                                        const float* qacc_warmstart, const mppo_forward_probe_t* out, void* stream) {
  using namespace mppo;
  MPPO_REQUIRE(m && qpos && qvel && qacc_warmstart && out, "mppo_physics_forward: null argument");
  MPPO_REQUIRE(ctrl || m->mv.nu == 0, "mppo_physics_forward: ctrl is null but the model has actuators");
  MPPO_REQUIRE(N >= 1, "mppo_physics_forward: N = %d", N);
  EnvArgs a{};
  a.N = N; a.mode = 2; a.n_frames = 1; a.p_qpos = qpos; a.p_qvel = qvel; a.p_ctrl = ctrl; a.p_warm = qacc_warmstart; a.probe = *out;
  return launch_env(m, a, static_cast<hipStream_t>(stream));
}
