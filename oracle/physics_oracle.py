"""CPU oracle for the rigid-body step behind `pipeline_step` / `pipeline_init`.   TEST INFRASTRUCTURE.

PARITY UNPINNED.  The reference calls `brax.envs.base.PipelineEnv.pipeline_step`
(`minppo/env.py:162`) and `pipeline_init` (`env.py:120`), i.e. `mujoco.mjx.step`
/ `mjx.forward` with solver=CG, 6 iterations, 6 line-search iterations
(`env.py:95-97`).  Brax, MuJoCo and MJX are third-party, un-pinned
(`requirements.txt:8-13`; likely brax 0.10-0.11 / mujoco 3.2.x on 2024-10-16),
absent from /root/reference and not installable here.  This file restates the
*published* MuJoCo/MJX algorithm for the model class the engine supports
(free / hinge / slide joints, sphere & capsule geoms against one ground plane,
joint limits, pyramidal friction cones, affine-bias actuators on joints):

  fwd_position : kinematics, com_pos (subtree_com, cinert, cdof), crb (dense M),
                 collision (plane-sphere, plane-capsule), make_constraint
  fwd_velocity : com_vel (cvel, cdof_dot), passive, rne (qfrc_bias)
  fwd_actuation, fwd_acceleration
  solve        : CG with Polak-Ribiere, M^-1 preconditioner, exact 1-D Newton
                 line search with bracketing (MJX solver.py structure)
  euler        : implicit joint damping, semi-implicit integration

It is validated by physical invariants and analytic cases
(tests/test_oracle_physics.py) and pins the HIP kernel through the committed
fixtures in tests/golden/.  Agreement with MJX itself was never measured.

All arrays are batched over environments: shape [N, ...].  dtype-generic
(float64 for the reference answer, float32 to mimic working precision).
"""

from __future__ import annotations

import math
from typing import Dict, NamedTuple

import numpy as np

JNT_FREE, JNT_HINGE, JNT_SLIDE = 0, 2, 3
MJ_MINVAL = 1e-15
MJ_MINIMP = 0.0001
MJ_MAXIMP = 0.9999


# ---------------------------------------------------------------------------
# batched quaternion / vector helpers
# ---------------------------------------------------------------------------


def qmul(a, b):
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bw, bx, by, bz = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack(
        [
            aw * bw - ax * bx - ay * by - az * bz,
            aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw,
        ],
        -1,
    )


def qmat(q):
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    r = np.stack(
        [
            w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z,
        ],
        -1,
    )
    return r.reshape(q.shape[:-1] + (3, 3))


def qrot(q, v):
    return np.einsum("...ij,...j->...i", qmat(q), v)


def safe_normalize(v):
    n = np.linalg.norm(v, axis=-1, keepdims=True)
    return v / np.where(n > 0, n, 1.0)


def axis_angle_quat(axis, angle):
    s = np.sin(angle * 0.5)[..., None]
    c = np.cos(angle * 0.5)[..., None]
    return np.concatenate([c, axis * s], -1)


def quat_integrate(q, w, dt):
    """MuJoCo mju_quatIntegrate: q <- normalize(q * exp(w*dt)), w in the local frame."""
    n = np.linalg.norm(w, axis=-1)
    ax = w / np.where(n > 0, n, 1.0)[..., None]
    dq = axis_angle_quat(ax, n * dt)
    return safe_normalize(qmul(q, dq))


def inert_mul(i, v):
    """cinert (10) times spatial motion (6) -> spatial force (6)   (mju_mulInertVec)."""
    return np.stack(
        [
            i[..., 0] * v[..., 0] + i[..., 3] * v[..., 1] + i[..., 4] * v[..., 2] - i[..., 8] * v[..., 4] + i[..., 7] * v[..., 5],
            i[..., 3] * v[..., 0] + i[..., 1] * v[..., 1] + i[..., 5] * v[..., 2] + i[..., 8] * v[..., 3] - i[..., 6] * v[..., 5],
            i[..., 4] * v[..., 0] + i[..., 5] * v[..., 1] + i[..., 2] * v[..., 2] - i[..., 7] * v[..., 3] + i[..., 6] * v[..., 4],
            i[..., 8] * v[..., 1] - i[..., 7] * v[..., 2] + i[..., 9] * v[..., 3],
            i[..., 6] * v[..., 2] - i[..., 8] * v[..., 0] + i[..., 9] * v[..., 4],
            i[..., 7] * v[..., 0] - i[..., 6] * v[..., 1] + i[..., 9] * v[..., 5],
        ],
        -1,
    )


def cross_motion(vel, v):
    """mju_crossMotion: spatial motion cross product vel x v."""
    w, l = vel[..., :3], vel[..., 3:]
    return np.concatenate([np.cross(w, v[..., :3]), np.cross(w, v[..., 3:]) + np.cross(l, v[..., :3])], -1)


def cross_force(vel, f):
    """mju_crossForce: spatial force cross product vel x* f."""
    w, l = vel[..., :3], vel[..., 3:]
    return np.concatenate([np.cross(w, f[..., :3]) + np.cross(l, f[..., 3:]), np.cross(w, f[..., 3:])], -1)


def make_frame(a):
    """MJX math.make_frame: orthonormal frame whose first row is `a`."""
    a = safe_normalize(a)
    y = np.zeros_like(a); y[..., 1] = 1
    z = np.zeros_like(a); z[..., 2] = 1
    b = np.where(((-0.5 < a[..., 1]) & (a[..., 1] < 0.5))[..., None], y, z)
    b = b - a * np.sum(a * b, -1, keepdims=True)
    b = safe_normalize(b)
    return np.stack([a, b, np.cross(a, b)], -2)


def _normalize_with_norm(x):
    """MJX math.normalize_with_norm: x / (|x| + 1e-6 [|x| == 0]), |x|."""
    n = np.linalg.norm(x, axis=-1)
    return x / (n + x.dtype.type(1e-6) * (n == 0).astype(x.dtype))[..., None], n


def closest_segment_point(a, b, pt):
    """MJX math.closest_segment_point: the point of segment a-b nearest to pt."""
    ab = b - a
    tt = np.sum((pt - a) * ab, -1) / (np.sum(ab * ab, -1) + 1e-6)
    return a + np.clip(tt, 0.0, 1.0)[..., None] * ab


def closest_segment_to_segment_points(a0, a1, b0, b1):
    """MJX math.closest_segment_to_segment_points (the routine behind sphere_capsule / capsule_capsule; a sphere is a segment
    of length 0): minimise over the two infinite lines, clip to the half lengths, then repair the clipping by projecting each
    clipped point on the other segment and keeping the closer of the two repairs."""
    dir_a, len_a = _normalize_with_norm(a1 - a0)
    dir_b, len_b = _normalize_with_norm(b1 - b0)
    half_a, half_b = 0.5 * len_a, 0.5 * len_b
    a_mid = a0 + dir_a * half_a[..., None]
    b_mid = b0 + dir_b * half_b[..., None]
    trans = a_mid - b_mid
    dab = np.sum(dir_a * dir_b, -1)
    dat = np.sum(dir_a * trans, -1)
    dbt = np.sum(dir_b * trans, -1)
    denom = 1 - dab * dab
    ta = (-dat + dab * dbt) / (denom + 1e-6)
    tb = dbt + ta * dab
    ta = np.clip(ta, -half_a, half_a)
    tb = np.clip(tb, -half_b, half_b)
    best_a = a_mid + dir_a * ta[..., None]
    best_b = b_mid + dir_b * tb[..., None]
    new_a = closest_segment_point(a0, a1, best_b)
    new_b = closest_segment_point(b0, b1, best_a)
    d1 = np.sum((new_a - best_b) ** 2, -1)
    d2 = np.sum((new_b - best_a) ** 2, -1)
    pick = (d1 < d2)[..., None]
    return np.where(pick, new_a, best_a), np.where(pick, best_b, new_b)


def manifold_points(poly, mask, n):
    """MJX collision_convex._manifold_points, batched over environments ([3P-recall]: MuJoCo 3.1 / 3.2; not present under
    /root/reference): four vertices of `poly` [V, 3] that span the contact patch of the candidates `mask` [N, V] seen along the
    normal `n` [N, 3] - a: the first candidate, b: the candidate farthest from a, c: the one farthest from the line a-b (in the
    plane orthogonal to n), d: the one farthest from the edges a-c / b-c.  Non-candidates carry -1e6; argmax takes the first maximum."""
    dist_mask = np.where(mask, 0.0, -1e6)
    a_idx = np.argmax(dist_mask, axis=1)
    a = poly[a_idx]                                                     # [N, 3]
    b_idx = np.argmax(((a[:, None, :] - poly[None]) ** 2).sum(-1) + dist_mask, axis=1)
    b = poly[b_idx]
    ab = np.cross(n, a - b)
    ap = a[:, None, :] - poly[None]                                     # [N, V, 3]
    c_idx = np.argmax(np.abs(np.einsum("nvk,nk->nv", ap, ab)) + dist_mask, axis=1)
    c = poly[c_idx]
    ac, bc = np.cross(n, a - c), np.cross(n, b - c)
    bp = b[:, None, :] - poly[None]
    dist_bp = np.abs(np.einsum("nvk,nk->nv", bp, bc)) + dist_mask
    dist_ap = np.abs(np.einsum("nvk,nk->nv", ap, ac)) + dist_mask
    d_idx = np.argmax(np.concatenate([dist_bp, dist_ap], axis=1), axis=1) % poly.shape[0]
    return np.stack([a_idx, b_idx, c_idx, d_idx], axis=1)


def sphere_sphere(pos1, r1, pos2, r2):
    """MJX collision_primitive._sphere_sphere -> dist, contact point, normal (from geom 1 to geom 2; +x when the centres coincide)."""
    n, dist = _normalize_with_norm(pos2 - pos1)
    n = np.where((dist == 0)[..., None], np.array([1.0, 0.0, 0.0], n.dtype), n)
    dist = dist - (r1 + r2)
    return dist, pos1 + n * (r1 + 0.5 * dist)[..., None], n


def plane_cylinder(height, axis, xaxis, r, half):
    """MJX collision_primitive.plane_cylinder ([3P-recall]: MuJoCo 3.1 / 3.2, the three-contact form of the C engine's mjc_PlaneCylinder;
    not present under /root/reference) for the plane z = const with normal +z, batched: `height` [N] of the cylinder's centre above the
    plane, unit `axis` and the geom's `xaxis` [N, 3] in the world.  Slot 0: the point of the lower rim nearest to the plane; slots 1, 2: two
    more points of that rim, 120 degrees to either side; when the cylinder lies on its side (|axis . n| half < 1e-3) slot 1 is the
    nearest point of the OTHER rim instead.  -> dist [N, 3], pos - centre [N, 3, 3]."""
    dt = axis.dtype
    n = np.array([0.0, 0.0, 1.0], dt)
    prjaxis = axis @ n
    sign = -np.where(prjaxis < 0, dt.type(-1), dt.type(1))  # (the axis is turned towards the plane)
    axis, prjaxis = axis * sign[:, None], prjaxis * sign
    vec = axis * prjaxis[:, None] - n
    ln = np.linalg.norm(vec, axis=-1)
    vec = np.where((ln < 1e-12)[:, None], xaxis * r, vec / (ln + dt.type(1e-15) * (ln == 0))[:, None] * r)
    prjvec = vec @ n
    axis, prjaxis = axis * half, prjaxis * half
    prjvec1 = -prjvec * dt.type(0.5)
    vec1 = _normalize_with_norm(np.cross(vec, axis))[0] * (r * dt.type(math.sqrt(3.0)) * dt.type(0.5))
    d1 = height + prjaxis + prjvec
    d2 = height + prjaxis + prjvec1
    d3 = height - prjaxis + prjvec
    side = np.abs(prjaxis) < 1e-3
    dist = np.stack([d1, np.where(side, d3, d2), d2], axis=1)
    half_v = vec * dt.type(0.5)
    p0 = axis + vec - n * (d1 * dt.type(0.5))[:, None]
    p1 = np.where(side[:, None], vec - axis - n * (d3 * dt.type(0.5))[:, None], axis + vec1 - half_v - n * (d2 * dt.type(0.5))[:, None])
    p2 = axis - vec1 - half_v - n * (d2 * dt.type(0.5))[:, None]
    return dist, np.stack([p0, p1, p2], axis=1)


class Hull:
    """One convex geom as sphere_convex / capsule_convex see it (MJX ConvexInfo): vertices, polygon faces (vertex lists,
    counter-clockwise seen from outside), outward face normals, edges and the two face normals beside each edge - all in the frame
    of the BODY the geom is fixed to (MJX works in the geom's frame; the geom's pose in its body is folded into the tables)."""

    def __init__(self, t, h, dtype):
        v0 = int(t["hull_vadr"][h])
        self.vert = np.asarray(t["hull_vert"][v0:int(t["hull_vadr"][h + 1])], dtype)
        f0, f1 = int(t["hull_fadr"][h]), int(t["hull_fadr"][h + 1])
        self.faces = [np.asarray(t["hull_fidx"][int(t["hull_face_adr"][f]):int(t["hull_face_adr"][f + 1])]) - v0 for f in range(f0, f1)]
        self.fnormal = np.asarray(t["hull_fnormal"][f0:f1], dtype)
        e0, e1 = int(t["hull_eadr"][h]), int(t["hull_eadr"][h + 1])
        self.edge = np.asarray(t["hull_edge"][e0:e1]) - v0
        self.enormal = np.asarray(t["hull_enormal"][e0:e1], dtype).reshape(-1, 2, 3)
        if "hull_udadr" in t:  # the hull's edge directions with the parallel ones dropped (the axes of convex_convex's edge-edge tests)
            self.udir = np.asarray(t["hull_udir"][int(t["hull_udadr"][h]):int(t["hull_udadr"][h + 1])], dtype).reshape(-1, 3)


def _face_support(hull, pts, r):
    """min over the points `pts` [N, P, 3] of (p - n r - face[0]) . n for every face -> [N, F] (MJX get_support)."""
    v0 = np.stack([hull.vert[f[0]] for f in hull.faces])                              # [F, 3]
    return np.min(np.einsum("npfk,fk->npf", pts[:, :, None, :] - v0[None, None], hull.fnormal) - r, axis=1)


def sphere_convex(sp, r, hull):
    """MJX collision_convex._sphere_convex ([3P-recall]: MuJoCo 3.1 / 3.2; not present under /root/reference), batched: sphere
    centre `sp` [N, 3] in the hull's frame.  The face with the least penetration among those the sphere's lowest point is behind
    (argmax of the support, faces with support >= 0 masked to -1e12); the sphere centre projected on that face's plane and, if it
    falls outside the polygon, on the nearest edge it is in front of.  -> dist [N], pos [N, 3], normal [N, 3] (sphere -> hull)."""
    dt = sp.dtype
    N = sp.shape[0]
    support = _face_support(hull, sp[:, None, :], r)
    support = np.where(support >= 0, dt.type(-1e12), support)
    best = np.argmax(support, axis=1)
    pt_out = np.zeros((N, 3), dt)
    for f in np.unique(best):
        sel = best == f
        poly, n = hull.vert[hull.faces[f]], hull.fnormal[f]
        c = sp[sel]
        pt = c - ((c - poly[0]) @ n)[:, None] * n
        p0, p1 = np.roll(poly, 1, axis=0), poly
        en = np.cross(p1 - p0, n)                                                      # [m, 3] outward in the face's plane
        ed = np.einsum("nmk,mk->nm", pt[:, None, :] - p0[None], en)
        inside = np.all(ed <= 0, axis=1)
        degenerate = np.all(en == 0, axis=1)
        ed = np.where(degenerate[None] | (ed < 0), dt.type(1e12), ed)
        idx = np.argmin(ed, axis=1)
        edge_pt = closest_segment_point(p0[idx], p1[idx], pt)
        pt_out[sel] = np.where(inside[:, None], pt, edge_pt)
    n, d = _normalize_with_norm(pt_out - sp)
    spt = sp + n * r
    return d - r, (pt_out + spt) * dt.type(0.5), n


def _clip_edge_to_planes(e0, e1, plane_pts, plane_normals):
    """MJX / Brax clip_edge_to_planes, batched over the edge ([N, 3] ends) for one set of planes [m, 3]: an end in front of a
    plane moves to the line's intersection with it (the candidate most along the edge wins); an edge with both ends in front of one
    plane is kept as it is and masked out, and so is an edge whose ends crossed."""
    dt = e0.dtype
    in0 = np.einsum("nmk,mk->nm", e0[:, None, :] - plane_pts[None], plane_normals) > 1e-6
    in1 = np.einsum("nmk,mk->nm", e1[:, None, :] - plane_pts[None], plane_normals) > 1e-6
    dirn = e1 - e0
    denom = dirn @ plane_normals.T                                                       # [N, m]
    tt = np.einsum("nmk,mk->nm", plane_pts[None] - e0[:, None, :], plane_normals) / (denom + dt.type(1e-6) * (denom == 0))
    cand = e0[:, None, :] + tt[..., None] * dirn[:, None, :]                             # [N, m, 3]
    rows = np.arange(e0.shape[0])

    def clip_point(p0, p1, in_front):
        new = np.where(in_front[..., None], cand, p0[:, None, :])
        dots = np.einsum("nmk,nk->nm", new - p0[:, None, :], p1 - p0)
        return new[rows, np.argmax(dots, axis=1)]

    n0, n1 = clip_point(e0, e1, in0), clip_point(e1, e0, in1)
    mask = ~np.any(in0 & in1, axis=1)
    n0 = np.where(mask[:, None], n0, e0)
    n1 = np.where(mask[:, None], n1, e1)
    mask = np.where(np.sum((e0 - e1) * (n0 - n1), -1) < 0, False, mask)
    return n0, n1, mask


def _clip_polygons(clip_poly, subj_poly, clip_n, subj_n):
    """MJX collision_convex._clip ([3P-recall]): the subject polygon's edges clipped against the side planes of the clipping polygon, and
    the clipping polygon's edges - projected onto the subject's plane along the clipping normal - clipped against the subject's side
    planes.  Polygons [m, 3] counter-clockwise seen from outside (side plane normals (p1 - p0) x n point away from the centre).
    -> candidate points [2 ms + 2 mc, 3] on the subject's plane and their mask."""
    dt = clip_poly.dtype
    c0, c1 = np.roll(clip_poly, 1, axis=0), clip_poly
    cn = np.cross(c1 - c0, clip_n)
    s0, s1 = np.roll(subj_poly, 1, axis=0), subj_poly
    sn = np.cross(s1 - s0, subj_n)
    e0, e1, m = _clip_edge_to_planes(s0, s1, c0, cn)

    def onto_subject_plane(poly):
        denom = clip_n @ subj_n
        tt = (subj_poly[0] @ subj_n - poly @ subj_n) / (denom + dt.type(1e-6) * (denom == 0))
        return poly + tt[:, None] * clip_n

    f0, f1, ms = _clip_edge_to_planes(onto_subject_plane(c0), onto_subject_plane(c1), s0, sn)
    return np.concatenate([e0, e1, f0, f1]), np.concatenate([m, m, ms, ms])


def convex_convex(A, B, R, tr):
    """A box or a mesh hull against a box or a mesh hull of another body, the form of MJX collision_convex._box_box ([3P-recall]; MJX runs
    this form for two boxes and - since 3.1.3 - a Gauss-map variant of it for meshes, which prunes the edge pairs and places an edge
    contact at the two edges' closest points: here every hull pair takes the box form).  Everything in B's frame; R [N, 3, 3] / tr [N, 3]
    take A's frame into it.  Separating-axis test over A's face normals, B's face normals and the cross products of the hulls' edge
    directions (parallel pairs ignored): per axis the smaller of the two overlaps of the projections, the axis of the least overlap wins
    (argmin: the first among equals - face axes before edge axes).  The reference face is the face most aligned with the axis on the hull
    that is better aligned, the other hull's most anti-aligned face is clipped against its side planes (_clip), the candidates behind the
    reference plane are projected onto it and four of them chosen (_manifold_points).  An edge-axis winner keeps the deepest point only.
    -> dist [N, 4] (1 = slot unused), pos [N, 4, 3] on the reference face, normal [N, 3] from A to B."""
    dt = B.vert.dtype
    N = R.shape[0]
    dist, pos, normal = np.ones((N, 4), dt), np.zeros((N, 4, 3), dt), np.zeros((N, 3), dt)
    nfa, nfb = len(A.faces), len(B.faces)
    for e in range(N):
        va = A.vert @ R[e].T + tr[e]
        na = A.fnormal @ R[e].T
        ea = A.udir @ R[e].T
        vb, nb, eb = B.vert, B.fnormal, B.udir
        cr = np.cross(np.tile(ea, (len(eb), 1)), np.repeat(eb, len(ea), axis=0))      # index j * nEa + i
        degenerate = (cr ** 2).sum(1) < 1e-6
        nrm = np.linalg.norm(cr, axis=1)
        cr = cr / (nrm + dt.type(1e-6) * (nrm == 0))[:, None]
        axes = np.concatenate([na, nb, cr])
        pa, pb = va @ axes.T, vb @ axes.T                                               # [V, naxes]
        d1, d2 = pa.max(0) - pb.min(0), pb.max(0) - pa.min(0)
        sign = np.where(d1 > d2, -1.0, 1.0).astype(dt)
        sup = np.minimum(d1, d2)
        sup[nfa + nfb:] = np.where(degenerate, dt.type(1e6), sup[nfa + nfb:])
        best = int(np.argmin(sup))
        axis, sg = axes[best], sign[best]
        da, db = na @ axis, nb @ axis
        fa, fb = int(np.argmax(da * sg)), int(np.argmax(db * -sg))
        poly_a, poly_b = va[A.faces[fa]], vb[B.faces[fb]]
        if abs(da[fa]) > abs(db[fb]):
            ref, ref_n, inc, inc_n = poly_a, na[fa], poly_b, nb[fb]
        else:
            ref, ref_n, inc, inc_n = poly_b, nb[fb], poly_a, na[fa]
        cand, mask = _clip_polygons(ref, inc, ref_n, inc_n)
        on_ref = cand - ((cand - ref[0]) @ ref_n)[:, None] * ref_n
        mask = mask & (((cand - ref[0]) @ -ref_n) > 1e-6)
        idx = manifold_points(on_ref, mask[None], ref_n[None])[0]
        pen = (cand[idx] - on_ref[idx]) @ -ref_n
        dd = np.where(mask[idx], -pen, dt.type(1.0))
        pp = on_ref[idx]
        if best >= nfa + nfb:  # an edge-edge axis: the deepest point of the manifold alone
            k = int(np.argmin(dd))
            dd = np.array([dd[k], 1.0, 1.0, 1.0], dt)
            pp = np.tile(pp[k], (4, 1))
        dist[e], pos[e], normal[e] = dd, pp, sg * axis
    return dist, pos, normal


def capsule_convex(cp, half, r, hull):
    """MJX collision_convex._capsule_convex ([3P-recall], as above), batched: capsule centre `cp` and half-axis vector `half`
    [N, 3] in the hull's frame.  Two slots: the capsule's segment clipped to the side planes of the best face (as in sphere_convex,
    support = the lower of the two ends), each clipped end a contact against the face; slot 0 is replaced by an edge contact when the
    hull edge nearest to the segment is in the segment's Voronoi region, penetrates by less than both face contacts and is not
    parallel to the face normal.  -> dist [N, 2], pos [N, 2, 3], normal [N, 2, 3] (capsule -> hull); an unused slot has dist = 1."""
    dt = cp.dtype
    N = cp.shape[0]
    c0, c1 = cp - half, cp + half
    support = _face_support(hull, np.stack([c0, c1], axis=1), r)
    has_support = np.all(support < 0, axis=1)
    support = np.where(support >= 0, dt.type(-1e12), support)
    best = np.argmax(support, axis=1)
    pos = np.zeros((N, 2, 3), dt)
    fpen = np.zeros((N, 2), dt)
    normal = hull.fnormal[best]                                                          # [N, 3]
    for f in np.unique(best):
        sel = best == f
        poly, n = hull.vert[hull.faces[f]], hull.fnormal[f]
        p0, p1 = np.roll(poly, 1, axis=0), poly
        en = np.cross(p1 - p0, n)
        q0, q1, mask = _clip_edge_to_planes(c0[sel], c1[sel], p0, en)
        for j, q in enumerate((q0, q1)):
            q = q - n * r
            fp = q - ((q - poly[0]) @ n)[:, None] * n
            pos[sel, j] = (q + fp) * dt.type(0.5)
            fpen[sel, j] = np.where(mask & has_support[sel], (fp - q) @ n, dt.type(-1))
    # the shallow edge contact
    ea, eb = hull.vert[hull.edge[:, 0]], hull.vert[hull.edge[:, 1]]                        # [E, 3]
    E = ea.shape[0]
    ep, cq = closest_segment_to_segment_points(np.broadcast_to(ea[None], (N, E, 3)), np.broadcast_to(eb[None], (N, E, 3)),
                                                np.broadcast_to(c0[:, None, :], (N, E, 3)), np.broadcast_to(c1[:, None, :], (N, E, 3)))
    edir = ep - cq
    eaxis, edist = _normalize_with_norm(edir)
    k = np.argmin(np.abs(edist), axis=1)
    rows = np.arange(N)
    eaxis, edist, ep, cq = eaxis[rows, k], edist[rows, k], ep[rows, k], cq[rows, k]
    degenerate = np.sum(edir[rows, k] ** 2, -1) < 1e-6
    front = np.all(np.einsum("njk,nk->nj", hull.enormal[k], eaxis) < 0, axis=1)
    epen = np.where(~degenerate & front, r - edist, dt.type(-1))
    epos = (ep + cq + eaxis * r) * dt.type(0.5)
    parallel = (np.abs(np.sum(eaxis * normal, -1)) > 0.99) & ~degenerate
    minf = fpen.min(axis=1)
    has_edge = (epen > 0) & np.where(minf > 0, epen < minf, True) & ~parallel
    pos[:, 0] = np.where(has_edge[:, None], epos, pos[:, 0])
    nrm = np.stack([np.where(has_edge[:, None], eaxis, -normal), -normal], axis=1)
    pen = np.stack([np.where(has_edge, epen, fpen[:, 0]), np.where(has_edge, dt.type(-1), fpen[:, 1])], axis=1)
    return -pen, pos, nrm


# ---------------------------------------------------------------------------
# data container
# ---------------------------------------------------------------------------


class PhysState(dict):
    """Batched MJX-`Data`-like record; keys are MuJoCo field names. dict for easy select()."""

    __getattr__ = dict.__getitem__

    def copy(self):  # shallow per-array copy
        return PhysState({k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in self.items()})


STATE_FIELDS = ("qpos", "qvel", "qacc_warmstart", "time",
                "cinert", "cvel", "qfrc_actuator", "subtree_com", "qacc", "xpos", "xquat")


class Physics:
    """Batched pipeline for one compiled model (tables from minppo_amd.model.CompiledModel.t)."""

    def __init__(self, tables: Dict[str, np.ndarray], dtype=np.float64, n_frames: int = 1):
        self.t = t = {k: (np.asarray(v, dtype) if np.asarray(v).dtype.kind == "f" else np.asarray(v)) for k, v in tables.items()}
        self.dtype = np.dtype(dtype)
        self.nq, self.nv, self.nu = int(t["nq"]), int(t["nv"]), int(t["nu"])
        self.nbody, self.njnt = int(t["nbody"]), int(t["njnt"])
        self.ncon, self.nlimit = int(t["ncon"]), int(t["nlimit"])
        self.npair = int(t.get("npair", 0))
        self.nefc = self.nlimit + 4 * self.ncon
        self.n_frames = n_frames
        self.timestep = float(t["timestep"])
        self.dt = self.timestep * n_frames
        # ancestor-dof lists per body (last dof chain)
        self.body_lastdof = np.full(self.nbody, -1, np.int64)
        for b in range(1, self.nbody):
            p = t["body_parent"][b]
            self.body_lastdof[b] = (t["body_dofadr"][b] + t["body_dofnum"][b] - 1) if t["body_dofnum"][b] > 0 else self.body_lastdof[p]
        # qpos index of each hinge/slide dof, -1 for free-joint dofs (for passive springs / limits)
        self.dof_qposadr = np.full(self.nv, -1, np.int64)
        for j in range(self.njnt):
            if t["jnt_type"][j] != JNT_FREE:
                self.dof_qposadr[t["jnt_dofadr"][j]] = t["jnt_qposadr"][j]

    # -- fwd_position ---------------------------------------------------------
    def kinematics(self, d: PhysState) -> None:
        t, nb = self.t, self.nbody
        N = d.qpos.shape[0]
        dt = self.dtype
        xpos = np.zeros((N, nb, 3), dt)
        xquat = np.zeros((N, nb, 4), dt); xquat[..., 0] = 1
        xanchor = np.zeros((N, self.njnt, 3), dt)
        xaxis = np.zeros((N, self.njnt, 3), dt)
        for b in range(1, nb):
            p = t["body_parent"][b]
            pos = xpos[:, p] + qrot(xquat[:, p], t["body_pos"][b])
            quat = qmul(xquat[:, p], np.broadcast_to(t["body_quat"][b], (N, 4)))
            for j in range(t["body_jntadr"][b], t["body_jntadr"][b] + t["body_jntnum"][b]):
                qa = t["jnt_qposadr"][j]
                jt = t["jnt_type"][j]
                if jt == JNT_FREE:
                    pos = d.qpos[:, qa:qa + 3].copy()
                    quat = safe_normalize(d.qpos[:, qa + 3:qa + 7])
                    xanchor[:, j] = pos
                    xaxis[:, j] = qrot(quat, t["jnt_axis"][j])
                else:
                    anchor = pos + qrot(quat, t["jnt_pos"][j])
                    axis = qrot(quat, t["jnt_axis"][j])
                    xanchor[:, j], xaxis[:, j] = anchor, axis
                    disp = d.qpos[:, qa] - t["qpos0"][qa]
                    if jt == JNT_HINGE:
                        qloc = axis_angle_quat(np.broadcast_to(t["jnt_axis"][j], (N, 3)), disp)
                        quat = qmul(quat, qloc)
                        pos = anchor - qrot(quat, t["jnt_pos"][j])  # re-anchor
                    else:  # slide
                        pos = pos + axis * disp[:, None]
            xpos[:, b], xquat[:, b] = pos, safe_normalize(quat)
        d["xpos"], d["xquat"], d["xanchor"], d["xaxis"] = xpos, xquat, xanchor, xaxis
        d["xmat"] = qmat(xquat)
        d["xipos"] = xpos + qrot(xquat, t["body_ipos"][None])
        d["ximat"] = qmat(qmul(xquat, np.broadcast_to(t["body_iquat"][None], xquat.shape)))

    def com_pos(self, d: PhysState) -> None:
        t, nb, nv = self.t, self.nbody, self.nv
        N = d.qpos.shape[0]
        dt = self.dtype
        mass = t["body_mass"]
        mpos = d.xipos * mass[None, :, None]
        msum = np.broadcast_to(mass, (N, nb)).copy()
        for b in range(nb - 1, 0, -1):
            p = t["body_parent"][b]
            mpos[:, p] += mpos[:, b]
            msum[:, p] += msum[:, b]
        d["subtree_com"] = mpos / np.maximum(msum, MJ_MINVAL)[..., None]
        root_com = d.subtree_com[:, t["body_rootid"]]
        off = d.xipos - root_com
        I = np.einsum("nbij,bj,nbkj->nbik", d.ximat, t["body_inertia"], d.ximat)
        oo = np.sum(off * off, -1)
        I = I + mass[None, :, None, None] * (oo[..., None, None] * np.eye(3, dtype=dt) - off[..., :, None] * off[..., None, :])
        cinert = np.zeros((N, nb, 10), dt)
        cinert[..., 0], cinert[..., 1], cinert[..., 2] = I[..., 0, 0], I[..., 1, 1], I[..., 2, 2]
        cinert[..., 3], cinert[..., 4], cinert[..., 5] = I[..., 0, 1], I[..., 0, 2], I[..., 1, 2]
        cinert[..., 6:9] = off * mass[None, :, None]
        cinert[..., 9] = mass[None]
        cinert[:, 0] = 0
        d["cinert"] = cinert
        cdof = np.zeros((N, nv, 6), dt)
        for j in range(self.njnt):
            b = t["jnt_bodyid"][j]
            da = t["jnt_dofadr"][j]
            offj = d.subtree_com[:, t["body_rootid"][b]] - d.xanchor[:, j]
            jt = t["jnt_type"][j]
            if jt == JNT_FREE:
                for k in range(3):
                    cdof[:, da + k, 3 + k] = 1
                    ax = d.xmat[:, b, :, k]
                    cdof[:, da + 3 + k, :3] = ax
                    cdof[:, da + 3 + k, 3:] = np.cross(ax, offj)
            elif jt == JNT_HINGE:
                cdof[:, da, :3] = d.xaxis[:, j]
                cdof[:, da, 3:] = np.cross(d.xaxis[:, j], offj)
            else:
                cdof[:, da, 3:] = d.xaxis[:, j]
        d["cdof"] = cdof

    def crb(self, d: PhysState) -> None:
        t, nb, nv = self.t, self.nbody, self.nv
        N = d.qpos.shape[0]
        crb = d.cinert.copy()
        for b in range(nb - 1, 0, -1):
            crb[:, t["body_parent"][b]] += crb[:, b]
        M = np.zeros((N, nv, nv), self.dtype)
        for i in range(nv):
            buf = inert_mul(crb[:, t["dof_bodyid"][i]], d.cdof[:, i])
            j = i
            while j >= 0:
                v = np.sum(d.cdof[:, j] * buf, -1)
                M[:, i, j] = v
                M[:, j, i] = v
                j = t["dof_parentid"][j]
            M[:, i, i] += t["dof_armature"][i]
        d["qM"] = M
        d["qLD"] = np.linalg.cholesky(M)

    def solve_m(self, d: PhysState, x):
        L = d.qLD
        y = np.linalg.solve(L, x[..., None])
        return np.linalg.solve(np.swapaxes(L, -1, -2), y)[..., 0]

    def jacp(self, d: PhysState, point, body: int):
        """Translational Jacobian [N,3,nv] of a world point attached to `body` (mj_jac)."""
        t = self.t
        N = point.shape[0]
        J = np.zeros((N, 3, self.nv), self.dtype)
        off = point - d.subtree_com[:, t["body_rootid"][body]]
        dof = self.body_lastdof[body]
        while dof >= 0:
            J[:, :, dof] = d.cdof[:, dof, 3:] + np.cross(d.cdof[:, dof, :3], off)
            dof = t["dof_parentid"][dof]
        return J

    def collision(self, d: PhysState) -> None:
        """Every candidate keeps a slot (MJX static shapes).  The first ncon - npair slots are ground contacts (MJX
        collision_primitive.plane_sphere / plane_capsule, plane_cylinder, collision_convex.plane_convex for boxes and meshes), the last npair are geom-geom
        pairs (sphere_sphere / sphere_capsule / capsule_capsule: one contact each)."""
        t = self.t
        N = d.qpos.shape[0]
        nc, npair = self.ncon, self.npair
        dt = self.dtype
        dist = np.zeros((N, nc), dt)
        cpos = np.zeros((N, nc, 3), dt)
        frame = np.zeros((N, nc, 3, 3), dt)
        n = np.array([0.0, 0.0, 1.0], dt)
        # convex (mesh) geoms: which hull vertices fill the geom's four slots this step (MJX collision_convex.plane_convex)
        ncvx = int(t["ncvx"]) if "ncvx" in t else 0
        sel_pos = np.zeros((N, 4 * ncvx, 3), dt)
        sel_ok = np.ones((N, 4 * ncvx), bool)
        for k in range(ncvx):
            b = t["cvx_body"][k]
            vert = np.asarray(t["cvx_vert"][t["cvx_vadr"][k]:t["cvx_vadr"][k + 1]], dt)  # [V, 3], body frame
            R = qmat(d.xquat[:, b])                                                        # [N, 3, 3]: world = R . local
            nl = R[:, 2, :]                                                                # the plane's normal (+z) in the body frame
            support = (t["plane_z"] - d.xpos[:, b, 2])[:, None] - nl @ vert.T               # [N, V]: depth below the plane
            idx = manifold_points(vert, support > np.maximum(0.0, support.max(1) - 1e-3)[:, None], nl)  # [N, 4]
            for j in range(4):
                sel_pos[:, 4 * k + j] = vert[idx[:, j]]
                sel_ok[:, 4 * k + j] = ~np.any(idx[:, :j] == idx[:, j:j + 1], axis=1)      # unique = first occurrence of the vertex
        for c in range(nc - npair):
            b = t["con_bodyid"][c]
            kind = int(t["con_cvx"][c]) if "con_cvx" in t else -1
            if kind <= -2:
                # a cylinder's three slots (MJX plane_cylinder), computed at the first
                if kind == -2:
                    centre = d.xpos[:, b] + qrot(d.xquat[:, b], np.broadcast_to(np.asarray(t["con_lpos"][c], dt), (N, 3)))
                    hv = np.asarray(t["con_axis"][c], dt)
                    half = dt.type(np.linalg.norm(t["con_axis"][c]))
                    axis = qrot(d.xquat[:, b], np.broadcast_to(hv / half, (N, 3)))
                    xaxis = qrot(d.xquat[:, b], np.broadcast_to(np.asarray(t["con_axis"][c + 1], dt), (N, 3)))
                    dd, pp = plane_cylinder(centre[:, 2] - dt.type(t["plane_z"]), axis, xaxis, dt.type(t["con_radius"][c]), half)
                    for j in range(3):
                        dist[:, c + j] = dd[:, j]
                        cpos[:, c + j] = centre + pp[:, j]
                        frame[:, c + j] = make_frame(np.broadcast_to(n, (N, 3)))
                continue
            cs = kind if ncvx else -1
            lpos = sel_pos[:, cs] if cs >= 0 else np.broadcast_to(np.asarray(t["con_lpos"][c], dt), (N, 3))
            centre = d.xpos[:, b] + qrot(d.xquat[:, b], lpos)
            r = t["con_radius"][c]
            dist[:, c] = centre[:, 2] - t["plane_z"] - r
            if cs >= 0:
                dist[:, c] = np.where(sel_ok[:, cs], dist[:, c], 1.0)  # a vertex chosen twice: the later slot is switched off (dist = 1)
            cpos[:, c] = centre - n * (r + 0.5 * dist[:, c])[:, None]
            # plane_capsule aligns the first tangent with the capsule axis projected on the plane, unless that projection is
            # shorter than 0.5 (then, and for spheres, make_frame's choice: +y for a z normal)
            axis = qrot(d.xquat[:, b], np.broadcast_to(t["con_axis"][c], (N, 3)).astype(dt))
            bvec = axis - n * (axis @ n)[:, None]
            bn = np.linalg.norm(bvec, axis=-1)
            t1 = np.where((bn < 0.5)[:, None], np.array([0.0, 1.0, 0.0], dt), bvec / np.maximum(bn, dt.type(1e-30))[:, None])
            frame[:, c, 0], frame[:, c, 1], frame[:, c, 2] = n, t1, np.cross(np.broadcast_to(n, (N, 3)), t1)
        for k in range(npair):
            c = nc - npair + k
            b1, b2 = t["pair_body"][k]
            g = t["pair_geom"][k]
            p1, h1, r1 = g[0:3], g[3:6], g[6]
            p2, h2, r2 = g[8:11], g[11:14], g[14]
            c1 = d.xpos[:, b1] + qrot(d.xquat[:, b1], p1)
            c2 = d.xpos[:, b2] + qrot(d.xquat[:, b2], p2)
            a1 = qrot(d.xquat[:, b1], np.broadcast_to(h1, (N, 3)).astype(dt))
            a2 = qrot(d.xquat[:, b2], np.broadcast_to(h2, (N, 3)).astype(dt))
            hid, slot = int(g[7]) - 1, int(g[15])
            hid1 = int(g[14]) - 1  # geom 1 is a convex hull too (box / mesh against box / mesh): four slots, computed at the first
            if hid >= 0 and hid1 >= 0:
                if slot == 0:
                    R1, R2 = qmat(d.xquat[:, b1]), qmat(d.xquat[:, b2])
                    Rr = np.einsum("nji,njk->nik", R2, R1)                               # A's (body 1) frame -> B's (body 2) frame
                    trr = np.einsum("nji,nj->ni", R2, d.xpos[:, b1] - d.xpos[:, b2])
                    dd, pp, nn = convex_convex(Hull(t, hid1, dt), Hull(t, hid, dt), Rr, trr)
                    fr = make_frame(np.einsum("nij,nj->ni", R2, nn))
                    for j in range(4):
                        dist[:, c + j] = dd[:, j]
                        cpos[:, c + j] = d.xpos[:, b2] + np.einsum("nij,nj->ni", R2, pp[:, j])
                        frame[:, c + j] = fr
                continue
            if hid >= 0:
                # geom 2 is a convex hull (box / mesh) fixed to b2: work in b2's frame, come back to the world (MJX sphere_convex /
                # capsule_convex); a capsule's pair owns two consecutive slots and is computed at the first
                if slot == 1:
                    continue
                R2 = qmat(d.xquat[:, b2])
                to_hull = lambda v: np.einsum("nji,nj->ni", R2, v)
                hull = Hull(t, hid, dt)
                if np.any(h1 != 0):
                    dd, pp, nn = capsule_convex(to_hull(c1 - d.xpos[:, b2]), to_hull(a1), dt.type(r1), hull)
                else:
                    dd, pp, nn = sphere_convex(to_hull(c1 - d.xpos[:, b2]), dt.type(r1), hull)
                    dd, pp, nn = dd[:, None], pp[:, None], nn[:, None]
                for j in range(dd.shape[1]):
                    dist[:, c + j] = dd[:, j]
                    cpos[:, c + j] = d.xpos[:, b2] + np.einsum("nij,nj->ni", R2, pp[:, j])
                    frame[:, c + j] = make_frame(np.einsum("nij,nj->ni", R2, nn[:, j]))
                continue
            q1, q2 = closest_segment_to_segment_points(c1 - a1, c1 + a1, c2 - a2, c2 + a2)
            dist[:, c], cpos[:, c], nn = sphere_sphere(q1, r1, q2, r2)
            frame[:, c] = make_frame(nn)
        d["con_dist"], d["con_pos"], d["con_frame"] = dist, cpos, frame

    def _kbi(self, solref, solimp, pos):
        dt = self.dtype
        timeconst = max(float(solref[0]), 2 * self.timestep)  # refsafe
        dampratio = float(solref[1])
        dmin, dmax, width, mid, power = [float(x) for x in solimp]
        dmin = min(max(dmin, MJ_MINIMP), MJ_MAXIMP)
        dmax = min(max(dmax, MJ_MINIMP), MJ_MAXIMP)
        width = max(MJ_MINVAL, width)
        mid = min(max(mid, MJ_MINIMP), MJ_MAXIMP)
        power = max(1.0, power)
        k = 1.0 / (dmax * dmax * timeconst * timeconst * dampratio * dampratio)
        b = 2.0 / (dmax * timeconst)
        if solref[0] <= 0:
            k = -float(solref[0]) / (dmax * dmax)
        if solref[1] <= 0:
            b = -float(solref[1]) / dmax
        imp_x = np.abs(pos) / dt.type(width)
        imp_a = dt.type(1.0 / mid ** (power - 1)) * imp_x ** power
        imp_b = 1 - dt.type(1.0 / (1 - mid) ** (power - 1)) * np.abs(1 - imp_x) ** power
        imp_y = np.where(imp_x < mid, imp_a, imp_b)
        imp = dmin + imp_y * (dmax - dmin)
        imp = np.clip(imp, dmin, dmax)
        imp = np.where(imp_x > 1.0, dmax, imp).astype(dt)
        return dt.type(k), dt.type(b), imp

    def make_constraint(self, d: PhysState) -> None:
        """Rows: joint limits (1 each) then contacts (4 pyramid edges each). Inactive rows are zeroed."""
        t, nv = self.t, self.nv
        N = d.qpos.shape[0]
        dt = self.dtype
        J = np.zeros((N, self.nefc, nv), dt)
        pos = np.zeros((N, self.nefc), dt)
        invw = np.zeros((N, self.nefc), dt)
        act = np.zeros((N, self.nefc), bool)
        row = 0
        for jid in t["lim_jntid"]:
            qa, da = t["jnt_qposadr"][jid], t["jnt_dofadr"][jid]
            dmin = d.qpos[:, qa] - t["jnt_range"][jid, 0]
            dmax = t["jnt_range"][jid, 1] - d.qpos[:, qa]
            p = np.minimum(dmin, dmax)
            a = p < 0
            J[:, row, da] = np.where(a, np.where(dmin < dmax, 1.0, -1.0), 0.0)
            pos[:, row] = np.where(a, p, 0.0)
            invw[:, row] = np.where(a, t["dof_invweight0"][da], 0.0)
            act[:, row] = a
            row += 1
        k_l, b_l, imp_l = self._kbi(t["limit_solref"], t["limit_solimp"], pos[:, :self.nlimit])
        for c in range(self.ncon):
            b = t["con_bodyid"][c]
            a = d.con_dist[:, c] < 0
            jp = self.jacp(d, d.con_pos[:, c], b)  # [N,3,nv]   (ground contacts: body1 = world -> zero)
            tw = t["body_invweight0"][b, 0]
            if c >= self.ncon - self.npair:  # geom-geom: relative motion of body2 against body1, both translational weights
                b1 = t["pair_body"][c - (self.ncon - self.npair)][0]
                jp = jp - self.jacp(d, d.con_pos[:, c], b1)
                tw = tw + t["body_invweight0"][b1, 0]
            jc = np.einsum("nij,njv->niv", d.con_frame[:, c], jp)  # rows: normal, t1, t2
            fri = t["con_friction"][c]
            iw = (tw + fri[0] * fri[0] * tw) * 2 * fri[0] * fri[0] / t["impratio"]
            r = row
            for k in (1, 2):
                for s in (1.0, -1.0):
                    J[:, r] = np.where(a[:, None], jc[:, 0] + jc[:, k] * (s * fri[0]), 0.0)  # condim 3: both tangents use the sliding coefficient
                    pos[:, r] = np.where(a, d.con_dist[:, c], 0.0)
                    invw[:, r] = np.where(a, iw, 0.0)
                    act[:, r] = a
                    r += 1
            row += 4
        k_c, b_c, imp_c = self._kbi(t["contact_solref"], t["contact_solimp"], pos[:, self.nlimit:])
        imp = np.concatenate([imp_l, imp_c], 1)
        kk = np.concatenate([np.full((N, self.nlimit), k_l, dt), np.full((N, 4 * self.ncon), k_c, dt)], 1)
        bb = np.concatenate([np.full((N, self.nlimit), b_l, dt), np.full((N, 4 * self.ncon), b_c, dt)], 1)
        R = np.maximum(invw * (1 - imp) / imp, MJ_MINVAL)
        jv = np.einsum("nrv,nv->nr", J, d.qvel)
        d["efc_J"] = J
        d["efc_D"] = np.where(act, 1.0 / R, 0.0).astype(dt)  # inactive rows are inert (J=0, aref=0)
        d["efc_aref"] = (-bb * jv - kk * imp * pos).astype(dt)
        d["efc_active_row"] = act

    # -- fwd_velocity -----------------------------------------------------------
    def com_vel(self, d: PhysState) -> None:
        t, nb, nv = self.t, self.nbody, self.nv
        N = d.qpos.shape[0]
        cvel = np.zeros((N, nb, 6), self.dtype)
        cdof_dot = np.zeros((N, nv, 6), self.dtype)
        for b in range(1, nb):
            v = cvel[:, t["body_parent"][b]].copy()
            for j in range(t["body_jntadr"][b], t["body_jntadr"][b] + t["body_jntnum"][b]):
                da = t["jnt_dofadr"][j]
                if t["jnt_type"][j] == JNT_FREE:
                    for k in range(3):  # translational dofs: cdof_dot = 0
                        v = v + d.cdof[:, da + k] * d.qvel[:, da + k, None]
                    for k in range(3, 6):
                        cdof_dot[:, da + k] = cross_motion(v, d.cdof[:, da + k])
                    for k in range(3, 6):
                        v = v + d.cdof[:, da + k] * d.qvel[:, da + k, None]
                else:
                    cdof_dot[:, da] = cross_motion(v, d.cdof[:, da])
                    v = v + d.cdof[:, da] * d.qvel[:, da, None]
            cvel[:, b] = v
        d["cvel"], d["cdof_dot"] = cvel, cdof_dot

    def passive(self, d: PhysState) -> None:
        t = self.t
        f = -t["dof_damping"][None] * d.qvel
        for dof in range(self.nv):
            qa = self.dof_qposadr[dof]
            if qa >= 0:
                stiff = t["jnt_stiffness"][t["dof_jntid"][dof]]
                if stiff != 0:
                    f[:, dof] -= stiff * (d.qpos[:, qa] - t["qpos_spring"][qa])
        d["qfrc_passive"] = f

    def rne(self, d: PhysState) -> None:
        t, nb, nv = self.t, self.nbody, self.nv
        N = d.qpos.shape[0]
        cacc = np.zeros((N, nb, 6), self.dtype)
        cacc[:, 0, 3:] = -t["gravity"]
        for b in range(1, nb):
            a = cacc[:, t["body_parent"][b]].copy()
            for k in range(t["body_dofadr"][b], t["body_dofadr"][b] + t["body_dofnum"][b]):
                a = a + d.cdof_dot[:, k] * d.qvel[:, k, None]
            cacc[:, b] = a
        cfrc = inert_mul(d.cinert, cacc) + cross_force(d.cvel, inert_mul(d.cinert, d.cvel))
        cfrc[:, 0] = 0
        for b in range(nb - 1, 0, -1):
            cfrc[:, t["body_parent"][b]] += cfrc[:, b]
        d["qfrc_bias"] = np.sum(d.cdof * cfrc[:, t["dof_bodyid"]], -1)

    # -- actuation / acceleration -------------------------------------------------
    def fwd_actuation(self, d: PhysState) -> None:
        t = self.t
        N = d.qpos.shape[0]
        qfrc = np.zeros((N, self.nv), self.dtype)
        if self.nu:
            ctrl = d.ctrl
            lim = t["act_ctrllimited"].astype(bool)
            ctrl = np.where(lim[None], np.clip(ctrl, t["act_ctrlrange"][:, 0], t["act_ctrlrange"][:, 1]), ctrl)
            length = t["act_gear"][None] * d.qpos[:, t["act_qposadr"]]
            velocity = t["act_gear"][None] * d.qvel[:, t["act_dofid"]]
            force = t["act_gain"][None] * ctrl + t["act_bias"][None, :, 0] + t["act_bias"][None, :, 1] * length + t["act_bias"][None, :, 2] * velocity
            flim = t["act_forcelimited"].astype(bool)
            force = np.where(flim[None], np.clip(force, t["act_forcerange"][:, 0], t["act_forcerange"][:, 1]), force)
            np.add.at(qfrc, (slice(None), t["act_dofid"]), force * t["act_gear"][None])
            d["actuator_force"] = force
        # MJX fwd_actuation: qfrc_actuator clipped to the joint's actuatorfrcrange where it has one (the table holds -FLT_MAX / FLT_MAX elsewhere)
        rng_ = np.asarray(t["dof_actfrcrange"], self.dtype)
        qfrc = np.where(qfrc < rng_[None, :, 0], rng_[None, :, 0], np.where(qfrc > rng_[None, :, 1], rng_[None, :, 1], qfrc))
        d["qfrc_actuator"] = qfrc.astype(self.dtype)

    def fwd_acceleration(self, d: PhysState) -> None:
        d["qfrc_smooth"] = d.qfrc_passive - d.qfrc_bias + d.qfrc_actuator
        d["qacc_smooth"] = self.solve_m(d, d.qfrc_smooth)

    # -- constraint solver (MJX solver.py structure) -------------------------------
    def _ctx_update_constraint(self, d, c):
        active = c["Jaref"] < 0
        c["active"] = active
        c["efc_force"] = d.efc_D * -c["Jaref"] * active
        c["qfrc_constraint"] = np.einsum("nrv,nr->nv", d.efc_J, c["efc_force"])
        c["gauss"] = 0.5 * np.sum((c["Ma"] - d.qfrc_smooth) * (c["qacc"] - d.qacc_smooth), -1)
        c["prev_cost"] = c["cost"]
        c["cost"] = 0.5 * np.sum(d.efc_D * c["Jaref"] * c["Jaref"] * active, -1) + c["gauss"]

    def _ctx_update_gradient(self, d, c):
        c["grad"] = c["Ma"] - d.qfrc_smooth - c["qfrc_constraint"]
        c["Mgrad"] = self.solve_m(d, c["grad"])

    def _ctx_create(self, d, qacc, grad=True):
        N = qacc.shape[0]
        c = {
            "qacc": qacc.copy(),
            "Jaref": np.einsum("nrv,nv->nr", d.efc_J, qacc) - d.efc_aref,
            "Ma": np.einsum("nij,nj->ni", d.qM, qacc),
            "cost": np.full(N, np.inf, self.dtype),
        }
        self._ctx_update_constraint(d, c)
        if grad:
            self._ctx_update_gradient(d, c)
            c["search"] = -c["Mgrad"]
        return c

    def _ls_point(self, alpha, jaref, jv, quad, quad_gauss):
        x = jaref + alpha[:, None] * jv
        active = x < 0
        q = np.sum(quad * active[:, None, :], -1) + quad_gauss  # [N,3]
        cost = alpha * alpha * q[:, 2] + alpha * q[:, 1] + q[:, 0]
        d0 = 2 * alpha * q[:, 2] + q[:, 1]
        d1 = 2 * q[:, 2] + (q[:, 2] == 0) * MJ_MINVAL
        return {"alpha": alpha, "cost": cost, "d0": d0, "d1": d1}

    def _linesearch(self, d, c, run):
        t = self.t
        dt = self.dtype
        scale = float(t["meaninertia"]) * max(1, self.nv)
        smag = np.linalg.norm(c["search"], axis=-1) * scale
        gtol = float(t["tolerance"]) * float(t["ls_tolerance"]) * smag
        mv = np.einsum("nij,nj->ni", d.qM, c["search"])
        jv = np.einsum("nrv,nv->nr", d.efc_J, c["search"])
        quad_gauss = np.stack([c["gauss"], np.sum(c["search"] * c["Ma"], -1) - np.sum(c["search"] * d.qfrc_smooth, -1),
                               0.5 * np.sum(c["search"] * mv, -1)], -1)
        quad = np.stack([0.5 * c["Jaref"] * c["Jaref"], jv * c["Jaref"], 0.5 * jv * jv], 1) * d.efc_D[:, None, :]
        pf = lambda a: self._ls_point(a, c["Jaref"], jv, quad, quad_gauss)
        N = smag.shape[0]
        p0 = pf(np.zeros(N, dt))
        lo = pf(p0["alpha"] - p0["d0"] / p0["d1"])
        lesser = lo["d0"] < p0["d0"]
        sel = lambda m, a, b: {k: np.where(m, a[k], b[k]) for k in a}
        hi = sel(lesser, p0, lo)
        lo = sel(lesser, lo, p0)
        swap = np.ones(N, bool)
        ls_iter = np.zeros(N, np.int64)
        in_br = lambda x, y: ((x["d0"] < y["d0"]) & (y["d0"] < 0)) | ((x["d0"] > y["d0"]) & (y["d0"] > 0))
        while True:
            done = (ls_iter >= int(t["ls_iterations"])) | ~swap | ((lo["d0"] < 0) & (lo["d0"] > -gtol)) | ((hi["d0"] > 0) & (hi["d0"] < gtol))
            go = ~done & run
            if not go.any():
                break
            with np.errstate(all="ignore"):
                lo_next = pf(lo["alpha"] - lo["d0"] / lo["d1"])
                hi_next = pf(hi["alpha"] - hi["d0"] / hi["d1"])
                mid = pf(0.5 * (lo["alpha"] + hi["alpha"]))
            nlo, nhi = lo, hi
            s1 = in_br(nlo, lo_next); nlo = sel(s1, lo_next, nlo)
            s2 = in_br(nlo, mid); nlo = sel(s2, mid, nlo)
            s3 = in_br(nlo, hi_next); nlo = sel(s3, hi_next, nlo)
            s4 = in_br(nhi, hi_next); nhi = sel(s4, hi_next, nhi)
            s5 = in_br(nhi, mid); nhi = sel(s5, mid, nhi)
            s6 = in_br(nhi, lo_next); nhi = sel(s6, lo_next, nhi)
            lo = sel(go, nlo, lo)
            hi = sel(go, nhi, hi)
            swap = np.where(go, s1 | s2 | s3 | s4 | s5 | s6, swap)
            ls_iter = ls_iter + go
        improved = (lo["cost"] < p0["cost"]) | (hi["cost"] < p0["cost"])
        alpha = np.where(lo["cost"] < hi["cost"], lo["alpha"], hi["alpha"])
        step = (improved & run) * alpha
        c["qacc"] = c["qacc"] + step[:, None] * c["search"]
        c["Ma"] = c["Ma"] + step[:, None] * mv
        c["Jaref"] = c["Jaref"] + step[:, None] * jv

    def solve(self, d: PhysState) -> None:
        t = self.t
        N = d.qpos.shape[0]
        if self.nefc == 0:
            d["qacc"] = d.qacc_smooth.copy()
            d["qfrc_constraint"] = np.zeros_like(d.qacc)
            d["qacc_warmstart"] = d.qacc.copy()
            d["solver_niter"] = np.zeros(N, np.int64)
            return
        warm = self._ctx_create(d, d.qacc_warmstart, grad=False)
        smth = self._ctx_create(d, d.qacc_smooth, grad=False)
        qacc = np.where((warm["cost"] < smth["cost"])[:, None], d.qacc_warmstart, d.qacc_smooth)
        c = self._ctx_create(d, qacc)
        scale = float(t["meaninertia"]) * max(1, self.nv)
        tol = float(t["tolerance"])
        niter = np.zeros(N, np.int64)
        while True:
            improvement = (c["prev_cost"] - c["cost"]) / scale
            gradient = np.linalg.norm(c["grad"], axis=-1) / scale
            done = (niter >= int(t["iterations"])) | (improvement < tol) | (gradient < tol)
            run = ~done
            if not run.any():
                break
            old = {k: c[k].copy() for k in ("qacc", "Ma", "Jaref", "grad", "Mgrad", "search", "cost", "prev_cost",
                                            "gauss", "efc_force", "qfrc_constraint", "active")}
            self._linesearch(d, c, run)
            prev_grad, prev_Mgrad = c["grad"], c["Mgrad"]
            self._ctx_update_constraint(d, c)
            self._ctx_update_gradient(d, c)
            beta = np.sum(c["grad"] * (c["Mgrad"] - prev_Mgrad), -1) / np.maximum(MJ_MINVAL, np.sum(prev_grad * prev_Mgrad, -1))
            beta = np.maximum(0, beta)
            c["search"] = -c["Mgrad"] + beta[:, None] * c["search"]
            for k, v in old.items():  # envs that already terminated keep their state (vmapped while_loop)
                m = run.reshape((N,) + (1,) * (v.ndim - 1))
                c[k] = np.where(m, c[k], v)
            niter = niter + run
        d["qacc"] = c["qacc"]
        d["qacc_warmstart"] = c["qacc"].copy()
        d["qfrc_constraint"] = c["qfrc_constraint"]
        d["efc_force"] = c["efc_force"]
        d["solver_niter"] = niter

    # -- integrator --------------------------------------------------------------
    def euler(self, d: PhysState) -> None:
        t = self.t
        h = self.dtype.type(self.timestep)
        # MJX forward.euler integrates (M + h D)^-1 (qfrc_smooth + qfrc_constraint) ALWAYS - it has no "is any dof damped" test (the C engine's
        # mj_Euler skips the solve for an undamped model and integrates the solver's qacc; MJX cannot branch on an array).  The two differ
        # wherever the solver has not converged (six CG iterations), damped or not: round 5 found the oracle on the C engine's side of
        # this for undamped models while kernel and twin were on MJX's (synth_pile: 24 contact slots at rest).
        dh = d.qM + h * np.eye(self.nv, dtype=self.dtype)[None] * t["dof_damping"][None, :, None]
        qacc = np.linalg.solve(dh, (d.qfrc_smooth + d.qfrc_constraint)[..., None])[..., 0]
        qvel = d.qvel + qacc * h
        qpos = d.qpos.copy()
        for j in range(self.njnt):
            qa, da = t["jnt_qposadr"][j], t["jnt_dofadr"][j]
            if t["jnt_type"][j] == JNT_FREE:
                qpos[:, qa:qa + 3] = d.qpos[:, qa:qa + 3] + h * qvel[:, da:da + 3]
                qpos[:, qa + 3:qa + 7] = quat_integrate(d.qpos[:, qa + 3:qa + 7], qvel[:, da + 3:da + 6], h)
            else:
                qpos[:, qa] = d.qpos[:, qa] + h * qvel[:, da]
        d["qpos"], d["qvel"] = qpos.astype(self.dtype), qvel.astype(self.dtype)
        d["time"] = d.time + h

    # -- public: mjx.forward / mjx.step / brax pipeline_init / pipeline_step ---------
    def forward(self, d: PhysState) -> None:
        self.kinematics(d); self.com_pos(d); self.crb(d); self.collision(d); self.make_constraint(d)
        self.com_vel(d); self.passive(d); self.rne(d)
        self.fwd_actuation(d); self.fwd_acceleration(d)
        self.solve(d)

    def make_data(self, N: int) -> PhysState:
        dt = self.dtype
        return PhysState(qpos=np.tile(self.t["qpos0"], (N, 1)).astype(dt), qvel=np.zeros((N, self.nv), dt),
                         ctrl=np.zeros((N, self.nu), dt), qacc_warmstart=np.zeros((N, self.nv), dt),
                         time=np.zeros(N, dt))

    def pipeline_init(self, qpos, qvel) -> PhysState:
        """brax.mjx.pipeline.init: make_data, set qpos/qvel (ctrl = 0), mjx.forward."""
        d = self.make_data(qpos.shape[0])
        d["qpos"], d["qvel"] = qpos.astype(self.dtype).copy(), qvel.astype(self.dtype).copy()
        self.forward(d)
        return d

    def pipeline_step(self, d: PhysState, action) -> PhysState:
        """brax.mjx.pipeline.step x n_frames: ctrl <- action; mjx.step = forward + euler."""
        d = d.copy()
        d["ctrl"] = action.astype(self.dtype)
        for _ in range(self.n_frames):
            self.forward(d)
            self.euler(d)
        return d
