// env_twin.cpp — a C++17 / OpenMP float32 twin of the environment step, for the CPU baseline of bench.py.     TEST / BENCH INFRASTRUCTURE.
//
// PARITY UNPINNED (see oracle/physics_oracle.py).  What it restates: reference `HumanoidEnv.reset / step` (minppo/env.py:124-196) with the
// third-party `pipeline_init / pipeline_step` behind them (env.py:120,162 -> mujoco.mjx.forward / step, solver CG 6 / 6, env.py:95-97),
// `compute_reward` (env.py:199-235), `is_done` (env.py:238-242), the NaN guard (env.py:173-176), the auto-reset select (env.py:179-180),
// `get_obs` (env.py:245-261) and the episode metrics (env.py:183-194) - the same published MuJoCo / MJX algorithm as oracle/physics_oracle.py
// and oracle/env_oracle.py, written the way a CPU implementation is: ONE environment per thread (`#pragma omp parallel for` over the
// environments, which are independent: train.py:136,140), scalar float32 loops, the tree recursions leaf -> root / root -> leaf as MuJoCo
// does them.  It reads the same model blob as the HIP engine (minppo_amd/model.py: to_blob) and keeps the engine's per-environment
// state record ([qpos | qvel | cinert[1:] | cvel[1:] | qfrc_actuator | pad | qacc_warmstart | subtree_com[1].x | time]), so the golden
// fixtures check it exactly like they check the kernel (tests/test_cpu_twin.py).
//
// SURVEY.md 8(d) "CPU baseline timing" asks for this: the reference's JAX-CPU path cannot run here or on the GPU box (no jax / brax /
// mujoco), a NumPy port with Python loops over bodies is not what a CPU would be asked to run.  Nothing under minppo_amd/ loads it.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "model_view.h"  // blob layout (array order, header words): the one definition the model compiler, the engine and this twin share

#ifdef _OPENMP
#include <omp.h>
#endif

using namespace mppo;

namespace {

struct V3 { float x, y, z; };
struct Q4 { float w, x, y, z; };
inline V3 ld3(const float* p) { return {p[0], p[1], p[2]}; }
inline void st3(float* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
inline Q4 ld4(const float* p) { return {p[0], p[1], p[2], p[3]}; }
inline void st4(float* p, Q4 q) { p[0] = q.w; p[1] = q.x; p[2] = q.y; p[3] = q.z; }
inline V3 add3(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 sub3(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 mul3(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross3(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline Q4 qmul(Q4 a, Q4 b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
inline Q4 qnormalize(Q4 q) {
  const float n = std::sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  const float s = n > 0.f ? 1.f / n : 1.f;
  return {q.w * s, q.x * s, q.y * s, q.z * s};
}
inline void qmat(Q4 q, float* m) {
  const float w = q.w, x = q.x, y = q.y, z = q.z;
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2.f * (x * y - w * z);         m[2] = 2.f * (x * z + w * y);
  m[3] = 2.f * (x * y + w * z);         m[4] = w * w - x * x + y * y - z * z; m[5] = 2.f * (y * z - w * x);
  m[6] = 2.f * (x * z - w * y);         m[7] = 2.f * (y * z + w * x);         m[8] = w * w - x * x - y * y + z * z;
}
inline V3 qrot(Q4 q, V3 v) {
  float m[9];
  qmat(q, m);
  return {m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z};
}
inline Q4 axis_angle(V3 axis, float angle) {
  const float s = std::sin(0.5f * angle), c = std::cos(0.5f * angle);
  return {c, axis.x * s, axis.y * s, axis.z * s};
}
inline V3 normalize_norm(V3 v, float& n) {  // MJX math.normalize_with_norm
  n = std::sqrt(dot3(v, v));
  return mul3(v, 1.f / (n + (n == 0.f ? 1e-6f : 0.f)));
}
inline V3 closest_segment_point(V3 a, V3 b, V3 pt) {
  const V3 ab = sub3(b, a);
  const float t = dot3(sub3(pt, a), ab) / (dot3(ab, ab) + 1e-6f);
  return add3(a, mul3(ab, std::min(std::max(t, 0.f), 1.f)));
}
inline void closest_segment_points(V3 a0, V3 a1, V3 b0, V3 b1, V3& best_a, V3& best_b) {  // MJX math.closest_segment_to_segment_points
  float len_a, len_b;
  const V3 dir_a = normalize_norm(sub3(a1, a0), len_a), dir_b = normalize_norm(sub3(b1, b0), len_b);
  const float half_a = 0.5f * len_a, half_b = 0.5f * len_b;
  const V3 a_mid = add3(a0, mul3(dir_a, half_a)), b_mid = add3(b0, mul3(dir_b, half_b));
  const V3 trans = sub3(a_mid, b_mid);
  const float dab = dot3(dir_a, dir_b), dat = dot3(dir_a, trans), dbt = dot3(dir_b, trans);
  const float denom = 1.f - dab * dab;
  float ta = (-dat + dab * dbt) / (denom + 1e-6f);
  float tb = dbt + ta * dab;
  ta = std::min(std::max(ta, -half_a), half_a);
  tb = std::min(std::max(tb, -half_b), half_b);
  best_a = add3(a_mid, mul3(dir_a, ta));
  best_b = add3(b_mid, mul3(dir_b, tb));
  const V3 new_a = closest_segment_point(a0, a1, best_b), new_b = closest_segment_point(b0, b1, best_a);
  const V3 e1 = sub3(new_a, best_b), e2 = sub3(new_b, best_a);
  if (dot3(e1, e1) < dot3(e2, e2)) best_a = new_a; else best_b = new_b;
}
// ---- sphere / capsule against a convex hull (MJX collision_convex._sphere_convex / _capsule_convex), everything in the hull's frame.
// Written out the way the NumPy oracle states it: per-plane candidate arrays, then the argmax / argmin over them.
struct HullRef {
  const int* face_adr; const int* fidx; const int* edge; const float* vert; const float* fnormal; const float* enormal;
  int f0, f1, e0, e1;
};
inline int best_face(const HullRef& H, V3 c0, V3 c1, float r, bool& has_support) {
  std::vector<float> sup(H.f1 - H.f0);
  has_support = true;
  for (int f = H.f0; f < H.f1; ++f) {
    const V3 n = ld3(H.fnormal + 3 * f), v0 = ld3(H.vert + 3 * H.fidx[H.face_adr[f]]);
    const float a = dot3(sub3(c0, v0), n) - r, b = dot3(sub3(c1, v0), n) - r;
    const float sp = std::min(a, b);
    if (!(sp < 0.f)) has_support = false;
    sup[f - H.f0] = sp >= 0.f ? -1e12f : sp;
  }
  return H.f0 + (int)(std::max_element(sup.begin(), sup.end()) - sup.begin());  // (the first maximum)
}
inline void sphere_convex(const HullRef& H, V3 sp, float r, float& dist, V3& pos, V3& nrm) {
  bool hs;
  const int f = best_face(H, sp, sp, r, hs);
  const V3 n = ld3(H.fnormal + 3 * f);
  const int a0 = H.face_adr[f], m = H.face_adr[f + 1] - a0;
  auto P = [&](int i) { return ld3(H.vert + 3 * H.fidx[a0 + ((i % m) + m) % m]); };
  V3 pt = sub3(sp, mul3(n, dot3(sub3(sp, P(0)), n)));
  std::vector<float> ed(m);
  bool inside = true;
  for (int i = 0; i < m; ++i) {
    const V3 p0 = P(i - 1), p1 = P(i), en = cross3(sub3(p1, p0), n);
    const float d = dot3(sub3(pt, p0), en);
    if (d > 0.f) inside = false;
    const bool degenerate = en.x == 0.f && en.y == 0.f && en.z == 0.f;
    ed[i] = (degenerate || d < 0.f) ? 1e12f : d;
  }
  if (!inside) {
    const int i = (int)(std::min_element(ed.begin(), ed.end()) - ed.begin());
    pt = closest_segment_point(P(i - 1), P(i), pt);
  }
  float d;
  nrm = normalize_norm(sub3(pt, sp), d);
  dist = d - r;
  pos = mul3(add3(pt, add3(sp, mul3(nrm, r))), 0.5f);
}
inline void capsule_convex(const HullRef& H, V3 cp, V3 half, float r, float* dist, V3* pos, V3* nrm) {
  const V3 c0 = sub3(cp, half), c1 = add3(cp, half);
  bool has_support;
  const int f = best_face(H, c0, c1, r, has_support);
  const V3 n = ld3(H.fnormal + 3 * f), nn = mul3(n, -1.f);
  const int a0 = H.face_adr[f], m = H.face_adr[f + 1] - a0;
  auto P = [&](int i) { return ld3(H.vert + 3 * H.fidx[a0 + ((i % m) + m) % m]); };
  // clip the segment against the face's side planes
  std::vector<V3> cand(m);
  std::vector<char> in0(m), in1(m);
  const V3 dir = sub3(c1, c0);
  bool both = false;
  for (int i = 0; i < m; ++i) {
    const V3 p0 = P(i - 1), en = cross3(sub3(P(i), p0), n);
    in0[i] = dot3(sub3(c0, p0), en) > 1e-6f;
    in1[i] = dot3(sub3(c1, p0), en) > 1e-6f;
    both = both || (in0[i] && in1[i]);
    const float denom = dot3(dir, en);
    cand[i] = add3(c0, mul3(dir, dot3(sub3(p0, c0), en) / (denom + (denom == 0.f ? 1e-6f : 0.f))));
  }
  auto clip_point = [&](V3 p0, V3 p1, const std::vector<char>& in_front) {
    int bi = 0;
    float bd = 0.f;
    for (int i = 0; i < m; ++i) {
      const float d = dot3(sub3(in_front[i] ? cand[i] : p0, p0), sub3(p1, p0));
      if (i == 0 || d > bd) { bd = d; bi = i; }
    }
    return in_front[bi] ? cand[bi] : p0;
  };
  V3 q0 = clip_point(c0, c1, in0), q1 = clip_point(c1, c0, in1);
  bool mask = !both;
  if (!mask) { q0 = c0; q1 = c1; }
  if (dot3(sub3(c0, c1), sub3(q0, q1)) < 0.f) mask = false;
  float fpen[2];
  const V3 q[2] = {sub3(q0, mul3(n, r)), sub3(q1, mul3(n, r))};
  for (int j = 0; j < 2; ++j) {
    const V3 fp = sub3(q[j], mul3(n, dot3(sub3(q[j], P(0)), n)));
    pos[j] = mul3(add3(q[j], fp), 0.5f);
    fpen[j] = (mask && has_support) ? dot3(sub3(fp, q[j]), n) : -1.f;
  }
  // nearest hull edge
  int ew = H.e0;
  float dmin = 0.f;
  for (int e = H.e0; e < H.e1; ++e) {
    V3 pe, pc;
    closest_segment_points(ld3(H.vert + 3 * H.edge[2 * e]), ld3(H.vert + 3 * H.edge[2 * e + 1]), c0, c1, pe, pc);
    const V3 dl = sub3(pe, pc);
    const float d = std::sqrt(dot3(dl, dl));
    if (e == H.e0 || d < dmin) { dmin = d; ew = e; }
  }
  V3 pe, pc;
  closest_segment_points(ld3(H.vert + 3 * H.edge[2 * ew]), ld3(H.vert + 3 * H.edge[2 * ew + 1]), c0, c1, pe, pc);
  const V3 edir = sub3(pe, pc);
  const bool degenerate = dot3(edir, edir) < 1e-6f;
  float edist;
  const V3 eaxis = normalize_norm(edir, edist);
  const bool front = dot3(ld3(H.enormal + 6 * ew), eaxis) < 0.f && dot3(ld3(H.enormal + 6 * ew + 3), eaxis) < 0.f;
  const float epen = (!degenerate && front) ? r - edist : -1.f;
  const bool parallel = std::fabs(dot3(eaxis, n)) > 0.99f && !degenerate;
  const float minf = std::min(fpen[0], fpen[1]);
  const bool has_edge = epen > 0.f && (minf > 0.f ? epen < minf : true) && !parallel;
  if (has_edge) pos[0] = mul3(add3(pe, add3(pc, mul3(eaxis, r))), 0.5f);
  nrm[0] = has_edge ? eaxis : nn;
  nrm[1] = nn;
  dist[0] = -(has_edge ? epen : fpen[0]);
  dist[1] = -(has_edge ? -1.f : fpen[1]);
}
// ---- a box / mesh hull against a box / mesh hull of another body (round 6): the form of MJX collision_convex._box_box for every hull pair,
// written out the way the NumPy oracle states it (oracle/physics_oracle.py convex_convex): A's vertices, normals and edge directions
// taken into B's frame, per-axis overlap arrays, argmin / argmax over them, the clipped candidates as arrays, _manifold_points.
struct HullFull {
  const int* face_adr; const int* fidx; const float* vert; const float* fnormal; const float* udir;
  int v0, v1, f0, f1, u0, u1;
};
inline bool clip_edge_to_planes(const std::vector<V3>& poly, V3 n, V3& e0, V3& e1) {  // MJX _clip_edge_to_planes against the polygon's side planes
  const int m = (int)poly.size();
  const V3 dir = sub3(e1, e0), rdir = sub3(e0, e1);
  V3 n0 = e0, n1 = e1;
  float d0 = 0.f, d1 = 0.f;
  bool both = false;
  for (int i = 0; i < m; ++i) {
    const V3 p0 = poly[(i + m - 1) % m], en = cross3(sub3(poly[i], p0), n);
    const bool in0 = dot3(sub3(e0, p0), en) > 1e-6f, in1 = dot3(sub3(e1, p0), en) > 1e-6f;
    const float denom = dot3(dir, en);
    const V3 cand = add3(e0, mul3(dir, dot3(sub3(p0, e0), en) / (denom + (denom == 0.f ? 1e-6f : 0.f))));
    const V3 c0 = in0 ? cand : e0, c1 = in1 ? cand : e1;
    const float x0 = dot3(sub3(c0, e0), dir), x1 = dot3(sub3(c1, e1), rdir);
    if (i == 0 || x0 > d0) { d0 = x0; n0 = c0; }
    if (i == 0 || x1 > d1) { d1 = x1; n1 = c1; }
    both = both || (in0 && in1);
  }
  bool mask = !both;
  if (!mask) { n0 = e0; n1 = e1; }
  if (dot3(rdir, sub3(n0, n1)) < 0.f) mask = false;
  e0 = n0; e1 = n1;
  return mask;
}
inline void convex_convex(const HullFull& A, const HullFull& B, const float* R, V3 tr, float* dist, V3* pos, V3& nrm) {
  auto rot = [&](V3 v) { return V3{R[0] * v.x + R[1] * v.y + R[2] * v.z, R[3] * v.x + R[4] * v.y + R[5] * v.z, R[6] * v.x + R[7] * v.y + R[8] * v.z}; };
  std::vector<V3> va, vb, na, nb, ea, eb;
  for (int v = A.v0; v < A.v1; ++v) va.push_back(add3(rot(ld3(A.vert + 3 * v)), tr));
  for (int v = B.v0; v < B.v1; ++v) vb.push_back(ld3(B.vert + 3 * v));
  for (int f = A.f0; f < A.f1; ++f) na.push_back(rot(ld3(A.fnormal + 3 * f)));
  for (int f = B.f0; f < B.f1; ++f) nb.push_back(ld3(B.fnormal + 3 * f));
  for (int u = A.u0; u < A.u1; ++u) ea.push_back(rot(ld3(A.udir + 3 * u)));
  for (int u = B.u0; u < B.u1; ++u) eb.push_back(ld3(B.udir + 3 * u));
  const int nfa = (int)na.size(), nfb = (int)nb.size();
  std::vector<V3> axes(na);
  axes.insert(axes.end(), nb.begin(), nb.end());
  std::vector<char> degenerate(nfa + nfb, 0);
  for (size_t j = 0; j < eb.size(); ++j)
    for (size_t i = 0; i < ea.size(); ++i) {
      const V3 cr = cross3(ea[i], eb[j]);
      degenerate.push_back(dot3(cr, cr) < 1e-6f);
      float nn;
      axes.push_back(normalize_norm(cr, nn));
    }
  int best = 0;
  float bsup = 0.f, bsign = 1.f;
  for (size_t k = 0; k < axes.size(); ++k) {
    float amax = -INFINITY, amin = INFINITY, bmax = -INFINITY, bmin = INFINITY;
    for (const V3& v : va) { const float p = dot3(v, axes[k]); amax = std::max(amax, p); amin = std::min(amin, p); }
    for (const V3& v : vb) { const float p = dot3(v, axes[k]); bmax = std::max(bmax, p); bmin = std::min(bmin, p); }
    const float d1 = amax - bmin, d2 = bmax - amin;
    const float sup = degenerate[k] ? 1e6f : std::min(d1, d2);
    if (k == 0 || sup < bsup) { bsup = sup; best = (int)k; bsign = d1 > d2 ? -1.f : 1.f; }
  }
  const V3 axis = axes[best];
  int fa = 0, fb = 0;
  for (int f = 1; f < nfa; ++f) if (dot3(na[f], axis) * bsign > dot3(na[fa], axis) * bsign) fa = f;
  for (int f = 1; f < nfb; ++f) if (dot3(nb[f], axis) * -bsign > dot3(nb[fb], axis) * -bsign) fb = f;
  std::vector<V3> pa, pb;
  for (int i = A.face_adr[A.f0 + fa]; i < A.face_adr[A.f0 + fa + 1]; ++i) pa.push_back(add3(rot(ld3(A.vert + 3 * A.fidx[i])), tr));
  for (int i = B.face_adr[B.f0 + fb]; i < B.face_adr[B.f0 + fb + 1]; ++i) pb.push_back(ld3(B.vert + 3 * B.fidx[i]));
  const bool ref_a = std::fabs(dot3(na[fa], axis)) > std::fabs(dot3(nb[fb], axis));
  const std::vector<V3>& ref = ref_a ? pa : pb;
  const std::vector<V3>& inc = ref_a ? pb : pa;
  const V3 ref_n = ref_a ? na[fa] : nb[fb], inc_n = ref_a ? nb[fb] : na[fa];
  const int ms = (int)inc.size(), mc = (int)ref.size();
  std::vector<V3> cand(2 * (ms + mc));
  std::vector<char> mask(2 * (ms + mc));
  for (int i = 0; i < ms; ++i) {
    V3 e0 = inc[(i + ms - 1) % ms], e1 = inc[i];
    const bool m = clip_edge_to_planes(ref, ref_n, e0, e1);
    cand[i] = e0; cand[ms + i] = e1; mask[i] = mask[ms + i] = m;
  }
  const float pden = dot3(ref_n, inc_n), pd = dot3(inc[0], inc_n);
  auto onto = [&](V3 p) { return add3(p, mul3(ref_n, (pd - dot3(p, inc_n)) / (pden + (pden == 0.f ? 1e-6f : 0.f)))); };
  for (int i = 0; i < mc; ++i) {
    V3 e0 = onto(ref[(i + mc - 1) % mc]), e1 = onto(ref[i]);
    const bool m = clip_edge_to_planes(inc, inc_n, e0, e1);
    cand[2 * ms + i] = e0; cand[2 * ms + mc + i] = e1; mask[2 * ms + i] = mask[2 * ms + mc + i] = m;
  }
  const int nc = (int)cand.size();
  std::vector<V3> on_ref(nc);
  std::vector<float> dm(nc), pen(nc);
  for (int c = 0; c < nc; ++c) {
    const float h = dot3(sub3(cand[c], ref[0]), ref_n);
    on_ref[c] = sub3(cand[c], mul3(ref_n, h));
    mask[c] = mask[c] && (-h > 1e-6f);
    dm[c] = mask[c] ? 0.f : -1e6f;
    pen[c] = dot3(sub3(cand[c], on_ref[c]), mul3(ref_n, -1.f));
  }
  auto argmax = [&](auto value) { int bi = 0; float bv = value(0); for (int c = 1; c < nc; ++c) { const float x = value(c); if (x > bv) { bv = x; bi = c; } } return bi; };
  int idx[4];
  idx[0] = argmax([&](int c) { return dm[c]; });
  const V3 a = on_ref[idx[0]];
  idx[1] = argmax([&](int c) { const V3 e = sub3(a, on_ref[c]); return dot3(e, e) + dm[c]; });
  const V3 b = on_ref[idx[1]];
  const V3 ab = cross3(ref_n, sub3(a, b));
  idx[2] = argmax([&](int c) { return std::fabs(dot3(sub3(a, on_ref[c]), ab)) + dm[c]; });
  const V3 cpt = on_ref[idx[2]];
  const V3 ac = cross3(ref_n, sub3(a, cpt)), bc = cross3(ref_n, sub3(b, cpt));
  const int i1 = argmax([&](int c) { return std::fabs(dot3(sub3(b, on_ref[c]), bc)) + dm[c]; });
  const int i2 = argmax([&](int c) { return std::fabs(dot3(sub3(a, on_ref[c]), ac)) + dm[c]; });
  idx[3] = (std::fabs(dot3(sub3(a, on_ref[i2]), ac)) + dm[i2]) > (std::fabs(dot3(sub3(b, on_ref[i1]), bc)) + dm[i1]) ? i2 : i1;
  for (int j = 0; j < 4; ++j) { pos[j] = on_ref[idx[j]]; dist[j] = mask[idx[j]] ? -pen[idx[j]] : 1.f; }
  if (best >= nfa + nfb) {
    int k = 0;
    for (int j = 1; j < 4; ++j) if (dist[j] < dist[k]) k = j;
    const float dk = dist[k];
    const V3 pk = pos[k];
    for (int j = 0; j < 4; ++j) { dist[j] = j == 0 ? dk : 1.f; pos[j] = pk; }
  }
  nrm = mul3(axis, bsign);
}
inline V3 frame_tangent(V3 n) {  // second row of MJX math.make_frame for a unit n
  V3 b = (n.y > -0.5f && n.y < 0.5f) ? V3{0.f, 1.f, 0.f} : V3{0.f, 0.f, 1.f};
  b = sub3(b, mul3(n, dot3(n, b)));
  const float l = std::sqrt(dot3(b, b));
  return mul3(b, l > 0.f ? 1.f / l : 1.f);
}
inline void inert_mul(const float* i, const float* v, float* r) {  // mju_mulInertVec
  r[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  r[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  r[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  r[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  r[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  r[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
inline void cross_motion(const float* vel, const float* v, float* r) {
  const V3 w = ld3(vel), l = ld3(vel + 3), a = ld3(v), b = ld3(v + 3);
  st3(r, cross3(w, a));
  st3(r + 3, add3(cross3(w, b), cross3(l, a)));
}
inline void cross_force(const float* vel, const float* f, float* r) {
  const V3 w = ld3(vel), l = ld3(vel + 3), a = ld3(f), b = ld3(f + 3);
  st3(r, add3(cross3(w, a), cross3(l, b)));
  st3(r + 3, cross3(w, b));
}

struct RewardCfg {  // = mppo_reward_cfg_t (include/minppo_hip.h)
  float height_min_z, height_max_z, exp_coefficient, subtraction_factor, max_diff_norm, w_ctrl_cost, w_original_pos, w_is_healthy, w_velocity;
};

struct Model {
  std::vector<int32_t> blob;
  int nq, nv, nu, nb, njnt, ncon, nlim, npair, nroot, ncvx, nefc, iterations, ls_iterations, include_c;
  int obs_dim, obs_pad, rec_dim;
  float timestep, tolerance, ls_tolerance, impratio, plane_z, meaninertia;
  int off[BLOB_ARRAY_COUNT];
  int hull_base = 0;  // word offset of the hull section (0: none)
  mppo::HullView hv{};
  const int* HI(int o) const { return blob.data() + hull_base + o; }
  const float* HF(int o) const { return reinterpret_cast<const float*>(blob.data()) + hull_base + o; }
  const int* I(int k) const { return blob.data() + off[k]; }
  const float* F(int k) const { return reinterpret_cast<const float*>(blob.data()) + off[k]; }
};

// per-thread working set of one environment's forward pass
struct Work {
  std::vector<float> xpos, xquat, xipos, ximat, xmat, xanchor, xaxis, subcom, submass, cinert, cdof, crb, M, L, Le, cvel, cdofdot, cacc, cfrc;
  std::vector<float> condist, conpos, confr, cvxsel, cvxok, J, D, aref, jaref, jv, force, qfs, qas, qact, qacc, Ma, grad, Mgrad, search, mv, qfc, tmp, t1, bias;
  std::vector<int> lastdof;
  explicit Work(const Model& m) {
    const int nb = m.nb, nv = m.nv, ne = std::max(m.nefc, 1), nc = std::max(m.ncon, 1);
    xpos.resize(3 * nb); xquat.resize(4 * nb); xipos.resize(3 * nb); ximat.resize(9 * nb); xmat.resize(9 * nb); xanchor.resize(3 * m.njnt); xaxis.resize(3 * m.njnt);
    subcom.resize(3 * nb); submass.resize(nb); cinert.resize(10 * nb); cdof.resize(6 * nv); crb.resize(10 * nb); M.resize(nv * nv); L.resize(nv * nv); Le.resize(nv * nv);
    cvel.resize(6 * nb); cdofdot.resize(6 * nv); cacc.resize(6 * nb); cfrc.resize(6 * nb);
    condist.resize(nc); conpos.resize(3 * nc); confr.resize(9 * nc); cvxsel.resize(12 * std::max(m.ncvx, 1)); cvxok.resize(4 * std::max(m.ncvx, 1));
    J.resize((size_t)ne * nv); D.resize(ne); aref.resize(ne); jaref.resize(ne); jv.resize(ne); force.resize(ne);
    for (auto* v : {&qfs, &qas, &qact, &qacc, &Ma, &grad, &Mgrad, &search, &mv, &qfc, &tmp, &t1, &bias}) v->resize(nv);
    lastdof.resize(nb);
  }
};

void kb_params(const float* solref, const float* solimp, float timestep, float& k, float& b) {
  const float timeconst = std::max(solref[0], 2.f * timestep);
  const float dampratio = solref[1];
  const float dmax = std::min(std::max(solimp[1], MJ_MINIMP), MJ_MAXIMP);
  k = 1.f / (dmax * dmax * timeconst * timeconst * dampratio * dampratio);
  b = 2.f / (dmax * timeconst);
  if (solref[0] <= 0.f) k = -solref[0] / (dmax * dmax);
  if (solref[1] <= 0.f) b = -solref[1] / dmax;
}
float impedance(const float* solimp, float pos) {
  const float dmin = std::min(std::max(solimp[0], MJ_MINIMP), MJ_MAXIMP), dmax = std::min(std::max(solimp[1], MJ_MINIMP), MJ_MAXIMP);
  const float width = std::max(MJ_MINVAL, solimp[2]), mid = std::min(std::max(solimp[3], MJ_MINIMP), MJ_MAXIMP), power = std::max(1.f, solimp[4]);
  const float x = std::fabs(pos) / width;
  const float a = (1.f / std::pow(mid, power - 1.f)) * std::pow(x, power);
  const float c = 1.f - (1.f / std::pow(1.f - mid, power - 1.f)) * std::pow(std::fabs(1.f - x), power);
  float imp = dmin + (x < mid ? a : c) * (dmax - dmin);
  imp = std::min(std::max(imp, dmin), dmax);
  if (x > 1.f) imp = dmax;
  return imp;
}

// in-place lower Cholesky of an nv x nv matrix (row-major copy in L); false: not positive definite (-> NaNs propagate like in MJX)
void cholesky(const float* A, float* L, int n) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      float s = A[i * n + j];
      for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = i == j ? std::sqrt(std::max(s, MJ_MINVAL)) : s / L[j * n + j];
    }
}
void chol_solve(const float* L, int n, const float* b, float* x) {  // x = (L L^T)^-1 b
  for (int i = 0; i < n; ++i) {
    float s = b[i];
    for (int k = 0; k < i; ++k) s -= L[i * n + k] * x[k];
    x[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    float s = x[i];
    for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k];
    x[i] = s / L[i * n + i];
  }
}

struct LsPoint { float alpha, cost, d0, d1; };
inline LsPoint ls_make(float alpha, float q0, float q1, float q2) {
  return {alpha, alpha * alpha * q2 + alpha * q1 + q0, 2.f * alpha * q2 + q1, 2.f * q2 + (q2 == 0.f ? MJ_MINVAL : 0.f)};
}
inline bool in_bracket(const LsPoint& x, const LsPoint& y) { return (x.d0 < y.d0 && y.d0 < 0.f) || (x.d0 > y.d0 && y.d0 > 0.f); }

// mjx.forward on (qpos, qvel, ctrl, qacc_warmstart): everything up to the solver's qacc; w keeps cinert, cvel, qact, M, qfs, qfc;
// returns subtree_com[1].x
float forward(const Model& m, Work& w, const float* qpos, const float* qvel, const float* ctrl, const float* warm) {
  const int nb = m.nb, nv = m.nv, nq = m.nq, njnt = m.njnt, ncon = m.ncon, nlim = m.nlim, npair = m.npair, nplane = ncon - npair, nefc = m.nefc;
  (void)nq;
  const int *body_parent = m.I(BI_body_parent), *body_rootid = m.I(BI_body_rootid), *body_jntadr = m.I(BI_body_jntadr), *body_jntnum = m.I(BI_body_jntnum),
            *body_dofadr = m.I(BI_body_dofadr), *body_dofnum = m.I(BI_body_dofnum), *jnt_type = m.I(BI_jnt_type), *jnt_qposadr = m.I(BI_jnt_qposadr),
            *jnt_dofadr = m.I(BI_jnt_dofadr), *jnt_bodyid = m.I(BI_jnt_bodyid), *dof_bodyid = m.I(BI_dof_bodyid), *dof_jntid = m.I(BI_dof_jntid),
            *dof_parentid = m.I(BI_dof_parentid), *dof_qposadr = m.I(BI_dof_qposadr);
  const float* qpos0 = m.F(BF_qpos0);
  // ---- kinematics (mj_kinematics): bodies are topologically ordered
  st3(&w.xpos[0], {0, 0, 0}); st4(&w.xquat[0], {1, 0, 0, 0});
  for (int b = 1; b < nb; ++b) {
    const int p = body_parent[b];
    const Q4 pq = ld4(&w.xquat[4 * p]);
    V3 pos = add3(ld3(&w.xpos[3 * p]), qrot(pq, ld3(m.F(BF_body_pos) + 3 * b)));
    Q4 quat = qmul(pq, ld4(m.F(BF_body_quat) + 4 * b));
    for (int j = body_jntadr[b]; j < body_jntadr[b] + body_jntnum[b]; ++j) {
      const int qa = jnt_qposadr[j], jt = jnt_type[j];
      if (jt == JNT_FREE) {
        pos = ld3(qpos + qa);
        quat = qnormalize(ld4(qpos + qa + 3));
        st3(&w.xanchor[3 * j], pos);
        st3(&w.xaxis[3 * j], qrot(quat, ld3(m.F(BF_jnt_axis) + 3 * j)));
      } else {
        const V3 anchor = add3(pos, qrot(quat, ld3(m.F(BF_jnt_pos) + 3 * j))), axis = qrot(quat, ld3(m.F(BF_jnt_axis) + 3 * j));
        st3(&w.xanchor[3 * j], anchor); st3(&w.xaxis[3 * j], axis);
        const float disp = qpos[qa] - qpos0[qa];
        if (jt == JNT_HINGE) {
          quat = qmul(quat, axis_angle(ld3(m.F(BF_jnt_axis) + 3 * j), disp));
          pos = sub3(anchor, qrot(quat, ld3(m.F(BF_jnt_pos) + 3 * j)));
        } else {
          pos = add3(pos, mul3(axis, disp));
        }
      }
    }
    quat = qnormalize(quat);
    st3(&w.xpos[3 * b], pos); st4(&w.xquat[4 * b], quat);
    qmat(quat, &w.xmat[9 * b]);
    st3(&w.xipos[3 * b], add3(pos, qrot(quat, ld3(m.F(BF_body_ipos) + 3 * b))));
    qmat(qmul(quat, ld4(m.F(BF_body_iquat) + 4 * b)), &w.ximat[9 * b]);
  }
  // ---- com_pos: subtree centres of mass leaf -> root, cinert about the tree's centre of mass, cdof
  const float* mass = m.F(BF_body_mass);
  for (int b = 0; b < nb; ++b) { w.submass[b] = mass[b]; st3(&w.subcom[3 * b], b ? mul3(ld3(&w.xipos[3 * b]), mass[b]) : V3{0, 0, 0}); }
  for (int b = nb - 1; b > 0; --b) {
    const int p = body_parent[b];
    st3(&w.subcom[3 * p], add3(ld3(&w.subcom[3 * p]), ld3(&w.subcom[3 * b])));
    w.submass[p] += w.submass[b];
  }
  for (int b = 0; b < nb; ++b) st3(&w.subcom[3 * b], mul3(ld3(&w.subcom[3 * b]), 1.f / std::max(w.submass[b], MJ_MINVAL)));
  const float comx = w.subcom[3 * 1];
  for (int k = 0; k < 10; ++k) w.cinert[k] = 0.f;
  for (int b = 1; b < nb; ++b) {
    float* ci = &w.cinert[10 * b];
    const V3 off = sub3(ld3(&w.xipos[3 * b]), ld3(&w.subcom[3 * body_rootid[b]]));
    const float mb = mass[b];
    const float* R = &w.ximat[9 * b];
    const float d0 = m.F(BF_body_inertia)[3 * b], d1 = m.F(BF_body_inertia)[3 * b + 1], d2 = m.F(BF_body_inertia)[3 * b + 2];
    const float oo = dot3(off, off);
    ci[0] = R[0] * R[0] * d0 + R[1] * R[1] * d1 + R[2] * R[2] * d2 + mb * (oo - off.x * off.x);
    ci[1] = R[3] * R[3] * d0 + R[4] * R[4] * d1 + R[5] * R[5] * d2 + mb * (oo - off.y * off.y);
    ci[2] = R[6] * R[6] * d0 + R[7] * R[7] * d1 + R[8] * R[8] * d2 + mb * (oo - off.z * off.z);
    ci[3] = R[0] * R[3] * d0 + R[1] * R[4] * d1 + R[2] * R[5] * d2 - mb * off.x * off.y;
    ci[4] = R[0] * R[6] * d0 + R[1] * R[7] * d1 + R[2] * R[8] * d2 - mb * off.x * off.z;
    ci[5] = R[3] * R[6] * d0 + R[4] * R[7] * d1 + R[5] * R[8] * d2 - mb * off.y * off.z;
    ci[6] = mb * off.x; ci[7] = mb * off.y; ci[8] = mb * off.z; ci[9] = mb;
  }
  for (int j = 0; j < njnt; ++j) {
    const int b = jnt_bodyid[j], da = jnt_dofadr[j], jt = jnt_type[j];
    const V3 off = sub3(ld3(&w.subcom[3 * body_rootid[b]]), ld3(&w.xanchor[3 * j]));
    if (jt == JNT_FREE) {
      const float* R = &w.xmat[9 * b];
      for (int k = 0; k < 3; ++k) {
        float* c = &w.cdof[6 * (da + k)];
        c[0] = c[1] = c[2] = 0.f; c[3] = k == 0; c[4] = k == 1; c[5] = k == 2;
        const V3 ax = {R[k], R[3 + k], R[6 + k]};
        st3(&w.cdof[6 * (da + 3 + k)], ax);
        st3(&w.cdof[6 * (da + 3 + k) + 3], cross3(ax, off));
      }
    } else if (jt == JNT_HINGE) {
      const V3 ax = ld3(&w.xaxis[3 * j]);
      st3(&w.cdof[6 * da], ax); st3(&w.cdof[6 * da + 3], cross3(ax, off));
    } else {
      st3(&w.cdof[6 * da], {0, 0, 0}); st3(&w.cdof[6 * da + 3], ld3(&w.xaxis[3 * j]));
    }
  }
  // ---- crb: composite inertias leaf -> root, dense M, Cholesky
  std::copy(w.cinert.begin(), w.cinert.end(), w.crb.begin());
  for (int b = nb - 1; b > 0; --b) for (int k = 0; k < 10; ++k) w.crb[10 * body_parent[b] + k] += w.crb[10 * b + k];
  std::fill(w.M.begin(), w.M.end(), 0.f);
  for (int i = 0; i < nv; ++i) {
    float buf[6];
    inert_mul(&w.crb[10 * dof_bodyid[i]], &w.cdof[6 * i], buf);
    for (int j = i; j >= 0; j = dof_parentid[j]) {
      const float* cj = &w.cdof[6 * j];
      float v = cj[0] * buf[0] + cj[1] * buf[1] + cj[2] * buf[2] + cj[3] * buf[3] + cj[4] * buf[4] + cj[5] * buf[5];
      if (j == i) v += m.F(BF_dof_armature)[i];
      w.M[i * nv + j] = v; w.M[j * nv + i] = v;
    }
  }
  cholesky(w.M.data(), w.L.data(), nv);
  // ---- collision: ground contacts (plane-sphere / plane-capsule end / a box's or a mesh's chosen hull vertices / cylinder rims), geom-geom pairs
  for (int k = 0; k < m.ncvx; ++k) {  // MJX collision_convex.plane_convex + _manifold_points
    const int b = m.I(BI_cvx_body)[k], v0 = m.I(BI_cvx_vadr)[k], v1 = m.I(BI_cvx_vadr)[k + 1];
    float R[9];
    qmat(ld4(&w.xquat[4 * b]), R);
    const V3 nl = {R[6], R[7], R[8]};
    const float h0 = m.plane_z - w.xpos[3 * b + 2];
    const float* vt = m.F(BF_cvx_vert);
    float smax = -INFINITY;
    for (int v = v0; v < v1; ++v) smax = std::max(smax, h0 - dot3(nl, ld3(vt + 3 * v)));
    const float thr = std::max(0.f, smax - 1e-3f);
    auto dm = [&](int v) { return (h0 - dot3(nl, ld3(vt + 3 * v)) > thr) ? 0.f : -1e6f; };
    auto argmax = [&](auto&& f) { int best = v0; float bv = -INFINITY; for (int v = v0; v < v1; ++v) { const float x = f(v); if (x > bv) { bv = x; best = v; } } return best; };
    const int ia = argmax([&](int v) { return dm(v); });
    const V3 A = ld3(vt + 3 * ia);
    const int ib = argmax([&](int v) { const V3 e = sub3(A, ld3(vt + 3 * v)); return dot3(e, e) + dm(v); });
    const V3 B = ld3(vt + 3 * ib);
    const V3 ab = cross3(nl, sub3(A, B));
    const int ic = argmax([&](int v) { return std::fabs(dot3(sub3(A, ld3(vt + 3 * v)), ab)) + dm(v); });
    const V3 Cc = ld3(vt + 3 * ic);
    const V3 ac = cross3(nl, sub3(A, Cc)), bc = cross3(nl, sub3(B, Cc));
    int id = v0;
    {
      float best = -INFINITY;
      for (int v = v0; v < v1; ++v) { const float x = std::fabs(dot3(sub3(B, ld3(vt + 3 * v)), bc)) + dm(v); if (x > best) { best = x; id = v; } }
      for (int v = v0; v < v1; ++v) { const float x = std::fabs(dot3(sub3(A, ld3(vt + 3 * v)), ac)) + dm(v); if (x > best) { best = x; id = v; } }
    }
    const int idx[4] = {ia, ib, ic, id};
    for (int j = 0; j < 4; ++j) {
      bool first = true;
      for (int i = 0; i < j; ++i) first = first && idx[i] != idx[j];
      st3(&w.cvxsel[3 * (4 * k + j)], ld3(vt + 3 * idx[j]));
      w.cvxok[4 * k + j] = first ? 1.f : 0.f;
    }
  }
  for (int c = 0; c < nplane; ++c) {
    const int b = m.I(BI_con_bodyid)[c];
    const Q4 q = ld4(&w.xquat[4 * b]);
    const int kind = m.I(BI_con_cvx)[c];
    if (kind <= -2) {
      // a cylinder against the plane (MJX plane_cylinder): three slots, all of them placed when the first comes up
      if (kind != -2) continue;
      const V3 ctr = add3(ld3(&w.xpos[3 * b]), qrot(q, ld3(m.F(BF_con_lpos) + 3 * c)));
      const V3 hv = ld3(m.F(BF_con_axis) + 3 * c);
      const float half = std::sqrt(dot3(hv, hv)), r = m.F(BF_con_radius)[c], height = ctr.z - m.plane_z;
      V3 axis = qrot(q, mul3(hv, 1.f / half));
      const V3 xaxis = qrot(q, ld3(m.F(BF_con_axis) + 3 * (c + 1))), up = {0, 0, 1};
      float prjaxis = axis.z;
      const float sign = prjaxis < 0.f ? 1.f : -1.f;  // turned towards the plane
      axis = mul3(axis, sign); prjaxis *= sign;
      V3 vec = sub3(mul3(axis, prjaxis), up);
      const float len = std::sqrt(dot3(vec, vec));
      vec = len < 1e-12f ? mul3(xaxis, r) : mul3(vec, r / len);
      const float prjvec = vec.z;
      axis = mul3(axis, half); prjaxis *= half;
      float nrm;
      const V3 vec1 = mul3(normalize_norm(cross3(vec, axis), nrm), r * std::sqrt(3.f) * 0.5f);
      const float d1 = height + prjaxis + prjvec, d2 = height + prjaxis - 0.5f * prjvec, d3 = height - prjaxis + prjvec;
      const bool side = std::fabs(prjaxis) < 1e-3f;
      const float dist[3] = {d1, side ? d3 : d2, d2};
      const V3 hvv = mul3(vec, 0.5f);
      const V3 rel[3] = {add3(axis, vec), side ? sub3(vec, axis) : sub3(add3(axis, vec1), hvv), sub3(sub3(axis, vec1), hvv)};
      for (int j = 0; j < 3; ++j) {
        w.condist[c + j] = dist[j];
        const V3 p = add3(ctr, rel[j]);
        st3(&w.conpos[3 * (c + j)], {p.x, p.y, p.z - 0.5f * dist[j]});
        st3(&w.confr[9 * (c + j)], up); st3(&w.confr[9 * (c + j) + 3], {0, 1, 0}); st3(&w.confr[9 * (c + j) + 6], {-1, 0, 0});
      }
      continue;
    }
    const int cs = m.ncvx > 0 ? kind : -1;
    const V3 centre = add3(ld3(&w.xpos[3 * b]), qrot(q, cs >= 0 ? ld3(&w.cvxsel[3 * cs]) : ld3(m.F(BF_con_lpos) + 3 * c)));
    const float rad = m.F(BF_con_radius)[c];
    float dist = centre.z - m.plane_z - rad;
    if (cs >= 0 && w.cvxok[cs] == 0.f) dist = 1.f;
    w.condist[c] = dist;
    st3(&w.conpos[3 * c], {centre.x, centre.y, centre.z - (rad + 0.5f * dist)});
    const V3 ax = qrot(q, ld3(m.F(BF_con_axis) + 3 * c));
    const float bn = std::sqrt(ax.x * ax.x + ax.y * ax.y);
    const V3 n = {0, 0, 1}, t1 = bn < 0.5f ? V3{0, 1, 0} : V3{ax.x / bn, ax.y / bn, 0.f};
    st3(&w.confr[9 * c], n); st3(&w.confr[9 * c + 3], t1); st3(&w.confr[9 * c + 6], cross3(n, t1));
  }
  for (int k = 0; k < npair; ++k) {
    const int c = nplane + k, b1 = m.I(BI_pair_body)[2 * k], b2 = m.I(BI_pair_body)[2 * k + 1];
    const float* gp = m.F(BF_pair_geom) + 16 * k;
    const Q4 q1 = ld4(&w.xquat[4 * b1]), q2 = ld4(&w.xquat[4 * b2]);
    const V3 c1 = add3(ld3(&w.xpos[3 * b1]), qrot(q1, ld3(gp))), h1 = qrot(q1, ld3(gp + 3));
    if (gp[7] != 0.f) {  // geom 2 is a convex hull fixed to b2: in b2's frame, then back
      if (gp[15] != 0.f) continue;  // (a later slot of a pair with several contacts: filled with the first)
      const int hid = (int)gp[7] - 1;
      if (gp[14] != 0.f && gp[3] == 0.f && gp[4] == 0.f && gp[5] == 0.f && gp[6] == 0.f) {  // geom 1 is a hull too: four slots
        const int hid1 = (int)gp[14] - 1;
        auto full = [&](int h) {
          return HullFull{m.HI(m.hv.face_adr), m.HI(m.hv.fidx), m.HF(m.hv.vert), m.HF(m.hv.fnormal), m.HF(m.hv.udir), m.HI(m.hv.vadr)[h], m.HI(m.hv.vadr)[h + 1],
                          m.HI(m.hv.fadr)[h], m.HI(m.hv.fadr)[h + 1], m.HI(m.hv.udadr)[h], m.HI(m.hv.udadr)[h + 1]};
        };
        float R1[9], R2[9], Rr[9];
        qmat(q1, R1); qmat(q2, R2);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rr[3 * i + j] = R2[i] * R1[j] + R2[3 + i] * R1[3 + j] + R2[6 + i] * R1[6 + j];
        const V3 x2 = ld3(&w.xpos[3 * b2]), dx = sub3(ld3(&w.xpos[3 * b1]), x2);
        const V3 tr = {R2[0] * dx.x + R2[3] * dx.y + R2[6] * dx.z, R2[1] * dx.x + R2[4] * dx.y + R2[7] * dx.z, R2[2] * dx.x + R2[5] * dx.y + R2[8] * dx.z};
        float d4[4];
        V3 p4[4], n4;
        convex_convex(full(hid1), full(hid), Rr, tr, d4, p4, n4);
        const V3 n = qrot(q2, n4), t1 = frame_tangent(n);
        for (int j = 0; j < 4; ++j) {
          w.condist[c + j] = d4[j];
          st3(&w.conpos[3 * (c + j)], add3(x2, qrot(q2, p4[j])));
          st3(&w.confr[9 * (c + j)], n); st3(&w.confr[9 * (c + j) + 3], t1); st3(&w.confr[9 * (c + j) + 6], cross3(n, t1));
        }
        continue;
      }
      const HullRef H{m.HI(m.hv.face_adr), m.HI(m.hv.fidx), m.HI(m.hv.edge), m.HF(m.hv.vert), m.HF(m.hv.fnormal), m.HF(m.hv.enormal),
                      m.HI(m.hv.fadr)[hid], m.HI(m.hv.fadr)[hid + 1], m.HI(m.hv.eadr)[hid], m.HI(m.hv.eadr)[hid + 1]};
      const Q4 q2i = {q2.w, -q2.x, -q2.y, -q2.z};
      const V3 x2 = ld3(&w.xpos[3 * b2]);
      const V3 cp = qrot(q2i, sub3(c1, x2)), hh = qrot(q2i, h1);
      float dist[2];
      V3 pos[2], nrm[2];
      int cnt = 1;
      if (gp[3] != 0.f || gp[4] != 0.f || gp[5] != 0.f) { capsule_convex(H, cp, hh, gp[6], dist, pos, nrm); cnt = 2; }
      else sphere_convex(H, cp, gp[6], dist[0], pos[0], nrm[0]);
      for (int j = 0; j < cnt; ++j) {
        const V3 n = qrot(q2, nrm[j]), t1 = frame_tangent(n);
        w.condist[c + j] = dist[j];
        st3(&w.conpos[3 * (c + j)], add3(x2, qrot(q2, pos[j])));
        st3(&w.confr[9 * (c + j)], n); st3(&w.confr[9 * (c + j) + 3], t1); st3(&w.confr[9 * (c + j) + 6], cross3(n, t1));
      }
      continue;
    }
    const V3 c2 = add3(ld3(&w.xpos[3 * b2]), qrot(q2, ld3(gp + 8))), h2 = qrot(q2, ld3(gp + 11));
    V3 p1, p2;
    closest_segment_points(sub3(c1, h1), add3(c1, h1), sub3(c2, h2), add3(c2, h2), p1, p2);
    float dist;
    V3 n = normalize_norm(sub3(p2, p1), dist);
    if (dist == 0.f) n = {1, 0, 0};
    dist -= gp[6] + gp[14];
    w.condist[c] = dist;
    st3(&w.conpos[3 * c], add3(p1, mul3(n, gp[6] + 0.5f * dist)));
    const V3 t1 = frame_tangent(n);
    st3(&w.confr[9 * c], n); st3(&w.confr[9 * c + 3], t1); st3(&w.confr[9 * c + 6], cross3(n, t1));
  }
  // ---- make_constraint: limits (one row each), contacts (four pyramid rows each); inactive rows are inert
  for (int b = 1; b < nb; ++b) w.lastdof[b] = body_dofnum[b] > 0 ? body_dofadr[b] + body_dofnum[b] - 1 : w.lastdof[body_parent[b]];
  w.lastdof[0] = -1;
  std::fill(w.J.begin(), w.J.end(), 0.f);
  float k_lim, b_lim, k_con, b_con;
  kb_params(m.F(BF_limit_solref), m.F(BF_limit_solimp), m.timestep, k_lim, b_lim);
  kb_params(m.F(BF_contact_solref), m.F(BF_contact_solimp), m.timestep, k_con, b_con);
  auto finish_row = [&](int r, bool act, float pos, float iw, bool lim) {
    const float imp = impedance(lim ? m.F(BF_limit_solimp) : m.F(BF_contact_solimp), act ? pos : 0.f);
    float s = 0.f;
    for (int k = 0; k < nv; ++k) s += w.J[(size_t)r * nv + k] * qvel[k];
    const float R = std::max((act ? iw : 0.f) * (1.f - imp) / imp, MJ_MINVAL);
    w.D[r] = act ? 1.f / R : 0.f;
    w.aref[r] = act ? -(lim ? b_lim : b_con) * s - (lim ? k_lim : k_con) * imp * pos : 0.f;
  };
  for (int r = 0; r < nlim; ++r) {
    const int jid = m.I(BI_lim_jntid)[r], qa = jnt_qposadr[jid], da = jnt_dofadr[jid];
    const float dlo = qpos[qa] - m.F(BF_jnt_range)[2 * jid], dhi = m.F(BF_jnt_range)[2 * jid + 1] - qpos[qa];
    const float pos = std::min(dlo, dhi);
    const bool act = pos < 0.f;
    if (act) w.J[(size_t)r * nv + da] = dlo < dhi ? 1.f : -1.f;
    finish_row(r, act, pos, m.F(BF_dof_invweight0)[da], true);
  }
  for (int c = 0; c < ncon; ++c) {
    const bool act = w.condist[c] < 0.f;
    const float mu = m.F(BF_con_friction)[3 * c];
    const int b2 = m.I(BI_con_bodyid)[c], b1 = c >= nplane ? m.I(BI_pair_body)[2 * (c - nplane)] : 0;
    float tw = m.F(BF_body_invweight0)[2 * b2];
    if (c >= nplane) tw += m.F(BF_body_invweight0)[2 * b1];
    const float iw = (tw + mu * mu * tw) * 2.f * mu * mu / m.impratio;
    const int r0 = nlim + 4 * c;
    if (act) {
      const V3 n = ld3(&w.confr[9 * c]), t1 = ld3(&w.confr[9 * c + 3]), t2 = ld3(&w.confr[9 * c + 6]);
      for (int side = 0; side < 2; ++side) {  // mj_jac of the contact point on body 2, minus the same on body 1 (world for ground contacts)
        const int b = side == 0 ? b2 : b1;
        if (b == 0) continue;
        const V3 off = sub3(ld3(&w.conpos[3 * c]), ld3(&w.subcom[3 * body_rootid[b]]));
        for (int d = w.lastdof[b]; d >= 0; d = dof_parentid[d]) {
          const V3 jb = add3(ld3(&w.cdof[6 * d + 3]), cross3(ld3(&w.cdof[6 * d]), off));
          const float sg = side == 0 ? 1.f : -1.f;
          const float jn = sg * dot3(n, jb), jt1 = sg * dot3(t1, jb), jt2 = sg * dot3(t2, jb);
          w.J[(size_t)(r0 + 0) * nv + d] += jn + mu * jt1;
          w.J[(size_t)(r0 + 1) * nv + d] += jn - mu * jt1;
          w.J[(size_t)(r0 + 2) * nv + d] += jn + mu * jt2;
          w.J[(size_t)(r0 + 3) * nv + d] += jn - mu * jt2;
        }
      }
    }
    for (int k = 0; k < 4; ++k) finish_row(r0 + k, act, w.condist[c], iw, false);
  }
  // ---- com_vel: cvel, cdof_dot root -> leaf
  for (int k = 0; k < 6; ++k) w.cvel[k] = 0.f;
  for (int b = 1; b < nb; ++b) {
    float v[6];
    for (int k = 0; k < 6; ++k) v[k] = w.cvel[6 * body_parent[b] + k];
    for (int j = body_jntadr[b]; j < body_jntadr[b] + body_jntnum[b]; ++j) {
      const int da = jnt_dofadr[j];
      if (jnt_type[j] == JNT_FREE) {
        for (int d = 0; d < 3; ++d) { for (int k = 0; k < 6; ++k) { w.cdofdot[6 * (da + d) + k] = 0.f; v[k] += w.cdof[6 * (da + d) + k] * qvel[da + d]; } }
        for (int d = 3; d < 6; ++d) cross_motion(v, &w.cdof[6 * (da + d)], &w.cdofdot[6 * (da + d)]);
        for (int d = 3; d < 6; ++d) for (int k = 0; k < 6; ++k) v[k] += w.cdof[6 * (da + d) + k] * qvel[da + d];
      } else {
        cross_motion(v, &w.cdof[6 * da], &w.cdofdot[6 * da]);
        for (int k = 0; k < 6; ++k) v[k] += w.cdof[6 * da + k] * qvel[da];
      }
    }
    for (int k = 0; k < 6; ++k) w.cvel[6 * b + k] = v[k];
  }
  // ---- rne: cacc root -> leaf, cfrc leaf -> root, qfrc_bias; passive; actuation
  for (int k = 0; k < 3; ++k) { w.cacc[k] = 0.f; w.cacc[3 + k] = -m.F(BF_gravity)[k]; }
  for (int k = 0; k < 6; ++k) w.cfrc[k] = 0.f;
  for (int b = 1; b < nb; ++b) {
    float a[6];
    for (int k = 0; k < 6; ++k) a[k] = w.cacc[6 * body_parent[b] + k];
    for (int d = body_dofadr[b]; d < body_dofadr[b] + body_dofnum[b]; ++d) for (int k = 0; k < 6; ++k) a[k] += w.cdofdot[6 * d + k] * qvel[d];
    for (int k = 0; k < 6; ++k) w.cacc[6 * b + k] = a[k];
    float ia[6], iv[6], cf[6];
    inert_mul(&w.cinert[10 * b], a, ia);
    inert_mul(&w.cinert[10 * b], &w.cvel[6 * b], iv);
    cross_force(&w.cvel[6 * b], iv, cf);
    for (int k = 0; k < 6; ++k) w.cfrc[6 * b + k] = ia[k] + cf[k];
  }
  for (int b = nb - 1; b > 0; --b) for (int k = 0; k < 6; ++k) w.cfrc[6 * body_parent[b] + k] += w.cfrc[6 * b + k];
  for (int k = 0; k < 6; ++k) w.cfrc[k] = 0.f;
  std::fill(w.qact.begin(), w.qact.end(), 0.f);
  for (int u = 0; u < m.nu; ++u) {
    const int d = m.I(BI_act_dofid)[u];
    float c = ctrl[u];
    if (m.I(BI_act_ctrllimited)[u]) c = std::min(std::max(c, m.F(BF_act_ctrlrange)[2 * u]), m.F(BF_act_ctrlrange)[2 * u + 1]);
    const float gear = m.F(BF_act_gear)[u];
    const float len = gear * qpos[m.I(BI_act_qposadr)[u]], vel = gear * qvel[d];
    float fo = m.F(BF_act_gain)[u] * c + m.F(BF_act_bias)[3 * u] + m.F(BF_act_bias)[3 * u + 1] * len + m.F(BF_act_bias)[3 * u + 2] * vel;
    if (m.I(BI_act_forcelimited)[u]) fo = std::min(std::max(fo, m.F(BF_act_forcerange)[2 * u]), m.F(BF_act_forcerange)[2 * u + 1]);
    w.qact[d] += fo * gear;
  }
  for (int d = 0; d < nv; ++d) {  // <joint actuatorfrcrange>: the joint's total actuator force, clamped
    const float lo = m.F(BF_dof_actfrcrange)[2 * d], hi = m.F(BF_dof_actfrcrange)[2 * d + 1];
    if (w.qact[d] < lo) w.qact[d] = lo; else if (w.qact[d] > hi) w.qact[d] = hi;
  }
  for (int d = 0; d < nv; ++d) {
    const float* cd = &w.cdof[6 * d];
    const float* f = &w.cfrc[6 * dof_bodyid[d]];
    const float bias = cd[0] * f[0] + cd[1] * f[1] + cd[2] * f[2] + cd[3] * f[3] + cd[4] * f[4] + cd[5] * f[5];
    float passive = -m.F(BF_dof_damping)[d] * qvel[d];
    const int qa = dof_qposadr[d];
    if (qa >= 0) {
      const float stiff = m.F(BF_jnt_stiffness)[dof_jntid[d]];
      if (stiff != 0.f) passive -= stiff * (qpos[qa] - m.F(BF_qpos_spring)[qa]);
    }
    w.bias[d] = bias;
    w.qfs[d] = passive - bias + w.qact[d];
  }
  chol_solve(w.L.data(), nv, w.qfs.data(), w.qas.data());  // qacc_smooth
  // ---- solve: CG, Polak-Ribiere, M^-1 preconditioner, bracketed Newton line search (MJX solver.py)
  if (nefc == 0) {
    for (int i = 0; i < nv; ++i) { w.qacc[i] = w.qas[i]; w.qfc[i] = 0.f; }
    return comx;
  }
  auto matvec = [&](const float* A, int rows, const float* x, float* y) {
    for (int r = 0; r < rows; ++r) { float s = 0.f; const float* a = A + (size_t)r * nv; for (int k = 0; k < nv; ++k) s += a[k] * x[k]; y[r] = s; }
  };
  float cost = 0.f, gauss = 0.f;
  auto ctx = [&](const float* src, float& gs, float& cs) {  // Ma, Jaref, gauss, cost at qacc = src
    for (int i = 0; i < nv; ++i) w.qacc[i] = src[i];
    matvec(w.M.data(), nv, w.qacc.data(), w.Ma.data());
    matvec(w.J.data(), nefc, w.qacc.data(), w.jaref.data());
    for (int r = 0; r < nefc; ++r) w.jaref[r] -= w.aref[r];
    gs = 0.f; cs = 0.f;
    for (int i = 0; i < nv; ++i) gs += (w.Ma[i] - w.qfs[i]) * (w.qacc[i] - w.qas[i]);
    for (int r = 0; r < nefc; ++r) if (w.jaref[r] < 0.f) cs += w.D[r] * w.jaref[r] * w.jaref[r];
    gs *= 0.5f; cs = 0.5f * cs + gs;
  };
  float gw, cw;
  ctx(warm, gw, cw);
  ctx(w.qas.data(), gauss, cost);
  if (cw < cost) ctx(warm, gauss, cost);
  float prev_cost = INFINITY;
  auto update_constraint_gradient = [&]() {
    for (int r = 0; r < nefc; ++r) w.force[r] = w.jaref[r] < 0.f ? -w.D[r] * w.jaref[r] : 0.f;
    for (int i = 0; i < nv; ++i) { float s = 0.f; for (int r = 0; r < nefc; ++r) s += w.J[(size_t)r * nv + i] * w.force[r]; w.qfc[i] = s; w.grad[i] = w.Ma[i] - w.qfs[i] - s; }
    chol_solve(w.L.data(), nv, w.grad.data(), w.Mgrad.data());
  };
  update_constraint_gradient();
  for (int i = 0; i < nv; ++i) w.search[i] = -w.Mgrad[i];
  const float scale = m.meaninertia * (float)std::max(nv, 1);
  for (int it = 0; it < m.iterations; ++it) {
    float gn = 0.f;
    for (int i = 0; i < nv; ++i) gn += w.grad[i] * w.grad[i];
    gn = std::sqrt(gn) / scale;
    if ((prev_cost - cost) / scale < m.tolerance || gn < m.tolerance) break;
    // line search
    matvec(w.M.data(), nv, w.search.data(), w.mv.data());
    matvec(w.J.data(), nefc, w.search.data(), w.jv.data());
    float sn = 0.f, sMa = 0.f, sq = 0.f, smv = 0.f;
    for (int i = 0; i < nv; ++i) { sn += w.search[i] * w.search[i]; sMa += w.search[i] * w.Ma[i]; sq += w.search[i] * w.qfs[i]; smv += w.search[i] * w.mv[i]; }
    const float gtol = m.tolerance * m.ls_tolerance * std::sqrt(sn) * scale;
    const float qg0 = gauss, qg1 = sMa - sq, qg2 = 0.5f * smv;
    auto eval = [&](float a) {
      float q0 = 0.f, q1 = 0.f, q2 = 0.f;
      for (int r = 0; r < nefc; ++r) {
        const float ja = w.jaref[r], v = w.jv[r], d = w.D[r];
        if (ja + a * v < 0.f) { q0 += 0.5f * ja * ja * d; q1 += v * ja * d; q2 += 0.5f * v * v * d; }
      }
      return ls_make(a, q0 + qg0, q1 + qg1, q2 + qg2);
    };
    const LsPoint p0 = eval(0.f);
    LsPoint lo = eval(p0.alpha - p0.d0 / p0.d1), hi;
    if (lo.d0 < p0.d0) hi = p0; else { hi = lo; lo = p0; }
    bool swap = true;
    for (int li = 0; li < m.ls_iterations; ++li) {
      if (!swap || (lo.d0 < 0.f && lo.d0 > -gtol) || (hi.d0 > 0.f && hi.d0 < gtol)) break;
      const LsPoint lo_next = eval(lo.alpha - lo.d0 / lo.d1), hi_next = eval(hi.alpha - hi.d0 / hi.d1), mid = eval(0.5f * (lo.alpha + hi.alpha));
      LsPoint nlo = lo, nhi = hi;
      const bool s1 = in_bracket(nlo, lo_next); if (s1) nlo = lo_next;
      const bool s2 = in_bracket(nlo, mid);     if (s2) nlo = mid;
      const bool s3 = in_bracket(nlo, hi_next); if (s3) nlo = hi_next;
      const bool s4 = in_bracket(nhi, hi_next); if (s4) nhi = hi_next;
      const bool s5 = in_bracket(nhi, mid);     if (s5) nhi = mid;
      const bool s6 = in_bracket(nhi, lo_next); if (s6) nhi = lo_next;
      lo = nlo; hi = nhi; swap = s1 || s2 || s3 || s4 || s5 || s6;
    }
    if (lo.cost < p0.cost || hi.cost < p0.cost) {
      const float alpha = lo.cost < hi.cost ? lo.alpha : hi.alpha;
      for (int i = 0; i < nv; ++i) { w.qacc[i] += alpha * w.search[i]; w.Ma[i] += alpha * w.mv[i]; }
      for (int r = 0; r < nefc; ++r) w.jaref[r] += alpha * w.jv[r];
    }
    // update_constraint, update_gradient, Polak-Ribiere
    float gs = 0.f, cs = 0.f, pgm = 0.f;
    for (int i = 0; i < nv; ++i) { w.t1[i] = w.Mgrad[i]; pgm += w.grad[i] * w.Mgrad[i]; gs += (w.Ma[i] - w.qfs[i]) * (w.qacc[i] - w.qas[i]); }
    for (int r = 0; r < nefc; ++r) if (w.jaref[r] < 0.f) cs += w.D[r] * w.jaref[r] * w.jaref[r];
    prev_cost = cost; gauss = 0.5f * gs; cost = 0.5f * cs + gauss;
    update_constraint_gradient();
    float num = 0.f;
    for (int i = 0; i < nv; ++i) num += w.grad[i] * (w.Mgrad[i] - w.t1[i]);
    const float beta = std::max(0.f, num / std::max(MJ_MINVAL, pgm));
    for (int i = 0; i < nv; ++i) w.search[i] = -w.Mgrad[i] + beta * w.search[i];
  }
  return comx;
}

inline bool any_nan(const float* p, int n) { for (int i = 0; i < n; ++i) if (std::isnan(p[i])) return true; return false; }

}  // namespace

extern "C" {

struct twin_metrics {  // = mppo_env_metrics_t
  float* episode_returns; int32_t* episode_lengths; float* returned_episode_returns; int32_t* returned_episode_lengths; int32_t* timestep; uint8_t* returned_episode;
};

void* twin_model_open(const void* host_blob, size_t nbytes) {
  const int32_t* wi = static_cast<const int32_t*>(host_blob);
  const uint32_t* wu = static_cast<const uint32_t*>(host_blob);
  const float* wf = static_cast<const float*>(host_blob);
  if (nbytes < 4 * (size_t)kBlobHeaderWords || wu[0] != kBlobMagic || wu[1] != kBlobVersion || wi[32] != BLOB_ARRAY_COUNT || ((size_t)wu[2] + (size_t)wu[35]) * 4 != nbytes) return nullptr;
  Model* m = new Model();
  if (wu[35] > 0) {
    m->hull_base = (int)wu[2];
    const int32_t* hs = wi + wu[2];
    m->hv = hull_view(hs[0], hs[1], hs[2], hs[3], hs[4], hs[5]);
    if ((uint32_t)m->hv.words != wu[35]) { delete m; return nullptr; }
  }
  m->blob.assign(wi, wi + nbytes / 4);
  m->nq = wi[3]; m->nv = wi[4]; m->nu = wi[5]; m->nb = wi[6]; m->njnt = wi[7]; m->ncon = wi[8]; m->nlim = wi[9]; m->iterations = wi[10]; m->ls_iterations = wi[11];
  m->nroot = wi[13]; m->include_c = wi[14] ? 1 : 0; m->npair = wi[15]; m->ncvx = wi[33];
  m->nefc = m->nlim + 4 * m->ncon;
  m->timestep = wf[16]; m->tolerance = wf[17]; m->ls_tolerance = wf[18]; m->impratio = wf[19]; m->plane_z = wf[20]; m->meaninertia = wf[21];
  for (int k = 0; k < BLOB_ARRAY_COUNT; ++k) m->off[k] = wi[kBlobHeaderWords + 2 * k];
  m->obs_dim = m->nq + 2 * m->nv + (m->include_c ? 16 * (m->nb - 1) : 0);
  m->obs_pad = (m->obs_dim + 3) & ~3;
  m->rec_dim = m->obs_pad + ((m->nv + 2 + 3) & ~3);
  return m;
}
void twin_model_close(void* h) { delete static_cast<Model*>(h); }
void twin_model_dims(const void* h, int32_t* out8) {
  const Model& m = *static_cast<const Model*>(h);
  out8[0] = m.nq; out8[1] = m.nv; out8[2] = m.nu; out8[3] = m.nb; out8[4] = m.obs_dim; out8[5] = m.obs_pad; out8[6] = m.rec_dim; out8[7] = m.nefc;
}
int twin_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

// state record of the reset state (pipeline_init of qpos0, qvel = 0, ctrl = 0: env.py:115-121 with reset_noise_scale = 0)
static void reset_record(const Model& m, Work& w, float* rec) {
  const int nq = m.nq, nv = m.nv, nb = m.nb, O = m.obs_dim, OP = m.obs_pad;
  std::vector<float> qvel(nv, 0.f), ctrl(std::max(m.nu, 1), 0.f), warm(nv, 0.f);
  const float comx = forward(m, w, m.F(BF_qpos0), qvel.data(), ctrl.data(), warm.data());
  std::fill(rec, rec + m.rec_dim, 0.f);
  std::copy(m.F(BF_qpos0), m.F(BF_qpos0) + nq, rec);
  int o = nq + nv;
  if (m.include_c) {
    std::copy(w.cinert.begin() + 10, w.cinert.begin() + 10 * nb, rec + o); o += 10 * (nb - 1);
    std::copy(w.cvel.begin() + 6, w.cvel.begin() + 6 * nb, rec + o); o += 6 * (nb - 1);
  }
  std::copy(w.qact.begin(), w.qact.end(), rec + o);
  (void)O;
  std::copy(w.qacc.begin(), w.qacc.end(), rec + OP);  // qacc_warmstart <- qacc of the forward pass
  rec[OP + nv] = comx;
  rec[OP + nv + 1] = 0.f;
}

void twin_env_reset(const void* h, int32_t N, float* state, float* reset_rec, float* obs, int32_t obs_ld, const twin_metrics* met) {
  const Model& m = *static_cast<const Model*>(h);
  Work w(m);
  reset_record(m, w, reset_rec);
#pragma omp parallel for schedule(static)
  for (int e = 0; e < N; ++e) {
    std::copy(reset_rec, reset_rec + m.rec_dim, state + (size_t)e * m.rec_dim);
    if (obs) std::copy(reset_rec, reset_rec + m.obs_pad, obs + (size_t)e * obs_ld);
    if (met && met->episode_returns) {
      met->episode_returns[e] = 0.f; met->episode_lengths[e] = 0; met->returned_episode_returns[e] = 0.f; met->returned_episode_lengths[e] = 0;
      met->timestep[e] = 0; met->returned_episode[e] = 0;
    }
  }
}

// HumanoidEnv.step for N environments (env.py:148-196), one environment per thread
void twin_env_step(const void* h, int32_t N, int32_t n_frames, const RewardCfg* rcp, float* state, const float* reset_rec, const float* action, int32_t act_ld,
                   float* obs, int32_t obs_ld, float* reward, uint8_t* done, const twin_metrics* met) {
  const Model& m = *static_cast<const Model*>(h);
  const RewardCfg rc = *rcp;
  const int nq = m.nq, nv = m.nv, nb = m.nb, OP = m.obs_pad, R = m.rec_dim;
  const float hstep = m.timestep;
#pragma omp parallel
  {
    Work w(m);
    std::vector<float> qpos(nq), qvel(nv), warm(nv), rhs(nv), qe(nv);
#pragma omp for schedule(static)
    for (int e = 0; e < N; ++e) {
      float* rec = state + (size_t)e * R;
      const float* act = action + (size_t)e * act_ld;
      std::copy(rec, rec + OP, obs + (size_t)e * obs_ld);  // get_obs of the PRE-step state (env.py:163)
      std::copy(rec, rec + nq, qpos.begin()); std::copy(rec + nq, rec + nq + nv, qvel.begin()); std::copy(rec + OP, rec + OP + nv, warm.begin());
      float p0 = 0.f;
      for (int i = 0; i < nq; ++i) { const float d = m.F(BF_qpos0)[i] - qpos[i]; p0 += d * d; }
      p0 = std::sqrt(p0);
      const float pre_z = qpos[2], pre_comx = rec[OP + nv], time_in = rec[OP + nv + 1];
      float comx = 0.f;
      bool bad = false;
      for (int f = 0; f < n_frames; ++f) {
        comx = forward(m, w, qpos.data(), qvel.data(), act, warm.data());
        bad = bad || any_nan(w.cinert.data(), 10 * nb) || any_nan(w.cvel.data(), 6 * nb) || any_nan(w.qact.data(), nv) || any_nan(w.xpos.data(), 3 * nb) ||
              any_nan(w.xquat.data(), 4 * nb) || any_nan(w.qfc.data(), nv);
        // euler: implicit joint damping, semi-implicit integration
        for (int i = 0; i < nv; ++i) rhs[i] = w.qfs[i] + w.qfc[i];
        std::copy(w.M.begin(), w.M.end(), w.Le.begin());
        for (int i = 0; i < nv; ++i) w.Le[i * nv + i] += hstep * m.F(BF_dof_damping)[i];
        cholesky(w.Le.data(), w.Le.data(), nv);
        chol_solve(w.Le.data(), nv, rhs.data(), qe.data());
        for (int i = 0; i < nv; ++i) { qvel[i] += hstep * qe[i]; warm[i] = w.qacc[i]; }
        for (int j = 0; j < m.njnt; ++j) {
          const int qa = m.I(BI_jnt_qposadr)[j], da = m.I(BI_jnt_dofadr)[j];
          if (m.I(BI_jnt_type)[j] == JNT_FREE) {
            for (int k = 0; k < 3; ++k) qpos[qa + k] += hstep * qvel[da + k];
            const V3 wv = ld3(&qvel[da + 3]);
            const float n = std::sqrt(dot3(wv, wv));
            const V3 ax = n > 0.f ? mul3(wv, 1.f / n) : wv;
            st4(&qpos[qa + 3], qnormalize(qmul(ld4(&qpos[qa + 3]), axis_angle(ax, n * hstep))));
          } else {
            qpos[qa] += hstep * qvel[da];
          }
        }
      }
      // reward (env.py:199-235: pose and height of the PRE-step state), done (post-step height, NaN guard)
      float asq = 0.f;
      for (int i = 0; i < m.nu; ++i) asq += act[i] * act[i];
      const float pos_r = std::exp(-rc.exp_coefficient * p0) - rc.subtraction_factor * std::min(std::max(p0, 0.f), rc.max_diff_norm);
      float healthy = pre_z < rc.height_min_z ? 0.f : 1.f;
      healthy = pre_z > rc.height_max_z ? 0.f : healthy;
      const float dt_env = hstep * (float)n_frames;
      const float rew = rc.w_ctrl_cost * (-asq) + rc.w_original_pos * pos_r + rc.w_velocity * ((comx - pre_comx) / dt_env) + rc.w_is_healthy * healthy;
      const float z = qpos[2];
      bad = bad || any_nan(qpos.data(), nq) || any_nan(qvel.data(), nv) || any_nan(warm.data(), nv) || std::isnan(comx);
      const bool dn = !(rc.height_min_z < z && z < rc.height_max_z) || bad;
      if (dn) {
        std::copy(reset_rec, reset_rec + R, rec);
        std::copy(reset_rec, reset_rec + OP, obs + (size_t)e * obs_ld);
      } else {
        std::fill(rec, rec + R, 0.f);
        std::copy(qpos.begin(), qpos.end(), rec); std::copy(qvel.begin(), qvel.end(), rec + nq);
        int o = nq + nv;
        if (m.include_c) {
          std::copy(w.cinert.begin() + 10, w.cinert.begin() + 10 * nb, rec + o); o += 10 * (nb - 1);
          std::copy(w.cvel.begin() + 6, w.cvel.begin() + 6 * nb, rec + o); o += 6 * (nb - 1);
        }
        std::copy(w.qact.begin(), w.qact.end(), rec + o);
        std::copy(warm.begin(), warm.end(), rec + OP);
        rec[OP + nv] = comx; rec[OP + nv + 1] = time_in + dt_env;
      }
      reward[e] = rew; done[e] = dn ? 1 : 0;
      if (met && met->episode_returns) {
        const float nd = dn ? 0.f : 1.f;
        const int ndi = dn ? 0 : 1;
        const float new_ret = met->episode_returns[e] + rew;
        const int new_len = met->episode_lengths[e] + 1;
        met->episode_returns[e] = new_ret * nd; met->episode_lengths[e] = new_len * ndi;
        met->returned_episode_returns[e] = met->returned_episode_returns[e] * nd + new_ret * (dn ? 1.f : 0.f);
        met->returned_episode_lengths[e] = met->returned_episode_lengths[e] * ndi + new_len * (dn ? 1 : 0);
        met->timestep[e] += 1; met->returned_episode[e] = dn ? 1 : 0;
      }
    }
  }
}

}  // extern "C"
