"""a sphere or a capsule of one body against a box or a mesh (convex hull) of another - MJX collision_convex
sphere_convex (one contact) / capsule_convex (two).  Known answers on a box for the oracle's restatement, the hull topology the model
compiler derives (faces, normals, edges), the blob's hull section, and the environment kernel (emulator build here, HIP with -m gpu)
and the C++ twin against the oracle on a scene of free bodies placed around the hulls."""
import math

import numpy as np
import pytest

from minppo_amd import mjcf
from minppo_amd.model import (GEOM_BOX, GEOM_CAPSULE, GEOM_MESH, GEOM_SPHERE, JNT_FREE, BodySpec, GeomSpec, JointSpec, ModelSpec, compile_model,
                              hull_topology)
from oracle.physics_oracle import Hull, Physics, PhysState, capsule_convex, convex_convex, sphere_convex

f32, f64 = np.float32, np.float64


def _free(name):
    return [JointSpec(name, JNT_FREE)]


def scene(extra_box_mesh_pair: bool = False) -> ModelSpec:
    """A box (off-centre, rotated in its body) and a 20-vertex mesh hull, a sphere and a capsule, all free, no gravity, no ground.  The
    round geoms carry contype 1 / conaffinity 0 (capsule: 1 / 1), the hulls 0 / 1: sphere-capsule, sphere-box, sphere-mesh, capsule-box
    and capsule-mesh are candidates, box-mesh is not."""
    rng = np.random.default_rng(5)
    v = rng.normal(size=(20, 3))
    v = v / np.linalg.norm(v, axis=1, keepdims=True) * [0.3, 0.25, 0.2]
    hull = dict(contype=1 if extra_box_mesh_pair else 0, conaffinity=1)
    return ModelSpec("cvx_scene", [
        BodySpec("box", "world", pos=(0, 0, 0), mass=2.0, inertia=(0.02, 0.03, 0.04), joints=_free("jb"),
                 geoms=[GeomSpec(GEOM_BOX, (0.3, 0.2, 0.1), pos=(0.02, 0.0, 0.01), quat=(0.9, 0.1, 0.2, 0.3), **hull)]),
        BodySpec("rock", "world", pos=(2, 0, 0), mass=1.5, inertia=(0.02, 0.02, 0.02), joints=_free("jr"), geoms=[GeomSpec(GEOM_MESH, (0,), vertices=v, **hull)]),
        BodySpec("ball", "world", pos=(0, 2, 0), mass=0.5, inertia=(0.002,) * 3, joints=_free("js"),
                 geoms=[GeomSpec(GEOM_SPHERE, (0.08,), pos=(0.01, 0, 0), contype=1, conaffinity=0)]),
        BodySpec("rod", "world", pos=(0, 4, 0), mass=0.7, inertia=(0.004, 0.004, 0.001), joints=_free("jc"),
                 geoms=[GeomSpec(GEOM_CAPSULE, (0.05, 0.15), quat=(0.8, 0.6, 0, 0), contype=1, conaffinity=1)]),
    ], [], gravity=(0, 0, 0), has_plane=False, free_root_z=0.0)


def scene_states(cm, N, rng):
    """Random orientations; the ball and the rod each 0.15 .. 0.42 from the centre of one of the two hulls (alternating): separated,
    touching and overlapping poses against faces, edges and corners."""
    q = np.tile(cm.t["qpos0"], (N, 1))
    for b in range(4):
        x = rng.normal(size=(N, 4))
        q[:, 7 * b + 3:7 * b + 7] = x / np.linalg.norm(x, axis=1, keepdims=True)
    q[:, 0:3] = 0
    q[:, 7:10] = [1.5, 0, 0]
    for b, base in ((2, 14), (3, 21)):
        d = rng.normal(size=(N, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        tgt = np.where((np.arange(N) % 2 == (b % 2))[:, None], q[:, 0:3], q[:, 7:10])
        q[:, base:base + 3] = tgt + d * rng.uniform(0.15, 0.42, size=(N, 1))
    return q


# ---------------------------------------------------------------------------
# the model compiler's side
# ---------------------------------------------------------------------------


def test_hull_topology_of_a_box_and_of_a_random_hull():
    c = np.array([[sx, sy, sz] for sx in (-1.0, 1.0) for sy in (-1.0, 1.0) for sz in (-1.0, 1.0)]) * [0.1, 0.2, 0.3]
    faces, normals, edges, enormals = hull_topology(c)
    assert sorted(len(f) for f in faces) == [4] * 6 and len(edges) == 12                     # the twelve triangles merged into six quads
    assert sorted(map(tuple, np.round(normals).astype(int).tolist())) == sorted([(0, 0, -1), (0, 0, 1), (0, -1, 0), (0, 1, 0), (-1, 0, 0), (1, 0, 0)])
    # a hexagonal prism: the four triangles of each hexagon merge into ONE six-sided face
    ang = np.arange(6) * np.pi / 3
    prism = np.array([[np.cos(a), np.sin(a), z] for z in (-0.3, 0.4) for a in ang]) * [0.2, 0.25, 1.0]
    faces, normals, edges, enormals = hull_topology(prism)
    assert sorted(len(f) for f in faces) == [4] * 6 + [6, 6] and len(edges) == 18
    rng = np.random.default_rng(0)
    v = rng.normal(size=(30, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    for verts in (c, v, prism):
        faces, normals, edges, enormals = hull_topology(verts)
        assert len(verts) - len(edges) + len(faces) == 2                                       # Euler
        for f, n in zip(faces, normals):
            p = verts[f]
            assert abs(np.linalg.norm(n) - 1) < 1e-12 and np.all((verts - p[0]) @ n < 1e-9)  # outward: the hull is behind every face
            for i in range(len(f)):                                                            # counter-clockwise seen from outside, convex
                assert np.dot(np.cross(p[i] - p[i - 1], p[(i + 1) % len(f)] - p[i]), n) > 0
        for (a, b), n2 in zip(edges, enormals):
            assert a < b and abs(np.dot(n2[0], verts[a] - verts[b])) < 1e-9 and abs(np.dot(n2[1], verts[a] - verts[b])) < 1e-9  # both faces contain the edge


def test_pair_rows_hull_tables_and_the_blob_section():
    cm = compile_model(scene())
    t = cm.t
    # MJX's function-table order: (sphere, capsule), (sphere, box), (sphere, mesh), (capsule, box) x 2 slots, (capsule, mesh) x 2 slots
    assert cm.npair == 7 and cm.ncon == 7 and int(t["nhull"]) == 2
    assert t["pair_body"].tolist() == [[3, 4], [3, 1], [3, 2], [4, 1], [4, 1], [4, 2], [4, 2]]
    assert t["pair_geom"][:, 7].tolist() == [0, 1, 2, 1, 1, 2, 2] and t["pair_geom"][:, 15].tolist() == [0, 0, 0, 0, 1, 0, 1]
    assert t["hull_vadr"].tolist() == [0, 8, 28] and t["hull_fadr"][1] == 6 and t["hull_eadr"][1] == 12
    # the box's hull is in the BODY frame: its geom's offset and rotation are folded in
    np.testing.assert_allclose(t["hull_vert"][:8].mean(0), [0.02, 0.0, 0.01], atol=1e-12)
    blob = cm.to_blob(True)
    words = np.frombuffer(blob, "<i4")
    total, hull_words = int(words[2]), int(words[35])
    assert hull_words > 0 and 4 * (total + hull_words) == len(blob)
    assert words[total:total + 5].tolist() == [2, 28, len(t["hull_fnormal"]), len(t["hull_fidx"]), len(t["hull_edge"])]
    # a model without such pairs has no section, and its table part is what it was
    from minppo_amd.model import load_model
    plain = load_model("synth_stompy_pro_sc").to_blob(True)
    assert np.frombuffer(plain, "<i4")[35] == 0 and 4 * int(np.frombuffer(plain, "<i4")[2]) == len(plain)
    # round 6: a box against a mesh hull of another body is a pair too (MJX convex_convex): four more slots, both hulls named in the rows
    t2 = compile_model(scene(extra_box_mesh_pair=True)).t
    assert t2["pair_body"].tolist()[-4:] == [[1, 2]] * 4 and t2["pair_geom"][-4:, 15].tolist() == [0, 1, 2, 3]
    assert t2["pair_geom"][-4:, 7].tolist() == [2] * 4 and t2["pair_geom"][-4:, 14].tolist() == [1] * 4
    assert t2["hull_udadr"].tolist()[:2] == [0, 3] and np.allclose(np.abs(t2["hull_udir"][:3] @ t2["hull_udir"][:3].T), np.eye(3), atol=1e-9)  # a box: three edge directions


MJCF = """
<mujoco model="cvx">
  <option timestep="0.002" gravity="0 0 0"/>
  <asset><mesh name="rock" vertex="{verts}"/></asset>
  <worldbody>
    <body name="box" pos="0 0 0"><freejoint/><inertial pos="0 0 0" mass="2" diaginertia="0.02 0.03 0.04"/>
      <geom type="box" size="0.3 0.2 0.1" pos="0.02 0 0.01" quat="0.9 0.1 0.2 0.3" contype="0" conaffinity="1"/></body>
    <body name="rock" pos="2 0 0"><freejoint/><inertial pos="0 0 0" mass="1.5" diaginertia="0.02 0.02 0.02"/>
      <geom type="mesh" mesh="rock" contype="0" conaffinity="1"/></body>
    <body name="ball" pos="0 2 0"><freejoint/><inertial pos="0 0 0" mass="0.5" diaginertia="0.002 0.002 0.002"/>
      <geom type="sphere" size="0.08" pos="0.01 0 0" contype="1" conaffinity="0"/></body>
    <body name="rod" pos="0 4 0"><freejoint/><inertial pos="0 0 0" mass="0.7" diaginertia="0.004 0.004 0.001"/>
      <geom type="capsule" size="0.05 0.15" quat="0.8 0.6 0 0" contype="1" conaffinity="1"/></body>
  </worldbody>
</mujoco>
"""


def test_the_same_scene_from_mjcf_text():
    spec = scene()
    verts = " ".join(repr(float(x)) for x in np.asarray(spec.bodies[1].geoms[0].vertices).reshape(-1))
    parsed = mjcf.parse_mjcf(MJCF.format(verts=verts), "cvx")
    parsed.has_plane, parsed.free_root_z = False, 0.0
    a, b = compile_model(parsed), compile_model(spec)
    for k in ("pair_body", "pair_geom", "hull_vadr", "hull_fadr", "hull_eadr", "hull_face_adr", "hull_fidx", "hull_edge"):
        np.testing.assert_array_equal(a.t[k], b.t[k], err_msg=k)
    for k in ("hull_vert", "hull_fnormal", "hull_enormal"):
        np.testing.assert_allclose(a.t[k], b.t[k], atol=1e-12, err_msg=k)


# ---------------------------------------------------------------------------
# the oracle's restatement: known answers on a cube of half-size 0.5
# ---------------------------------------------------------------------------


def _cube():
    spec = ModelSpec("cube", [
        BodySpec("cube", "world", joints=_free("a"), geoms=[GeomSpec(GEOM_BOX, (0.5, 0.5, 0.5), contype=0, conaffinity=1)]),
        BodySpec("ball", "world", pos=(0, 0, 2), joints=_free("b"), geoms=[GeomSpec(GEOM_SPHERE, (0.1,), contype=1, conaffinity=0)])],
        [], has_plane=False, free_root_z=0.0)
    return Hull(compile_model(spec).t, 0, np.dtype(f64))


def test_sphere_convex_known_answers():
    h = _cube()
    S = lambda p, r: [x[0] for x in sphere_convex(np.array([p], f64), f64(r), h)]
    s2, s3 = math.sqrt(2), math.sqrt(3)
    d, pos, n = S((0.1, -0.2, 0.6), 0.2)                       # over the top face
    assert d == pytest.approx(-0.1) and pos == pytest.approx([0.1, -0.2, 0.45]) and n == pytest.approx([0, 0, -1])
    d, pos, n = S((0.7, 0, 0.7), 0.3)                          # off the edge x = z = 0.5
    assert d == pytest.approx(0.2 * s2 - 0.3, abs=1e-6) and n == pytest.approx([-1 / s2, 0, -1 / s2], abs=1e-5)
    assert pos == pytest.approx(np.array([0.5, 0, 0.5]) - 0.5 * d * np.array(n), abs=1e-5)   # halfway between the two surfaces
    d, pos, n = S((0.7, 0.7, 0.7), 0.4)                        # off the corner
    assert d == pytest.approx(0.2 * s3 - 0.4, abs=1e-6) and n == pytest.approx([-1 / s3] * 3, abs=1e-5)
    d, pos, n = S((0.1, 0.2, 2.0), 0.2)                        # far away: some positive distance, never a contact
    assert d > 1.0


def test_capsule_convex_known_answers():
    h = _cube()
    C = lambda p, hv, r: [x[0] for x in capsule_convex(np.array([p], f64), np.array([hv], f64), f64(r), h)]
    d, pos, n = C((0, 0.1, 0.55), (0.3, 0, 0), 0.1)           # lying on the top face: both ends touch
    assert d == pytest.approx([-0.05, -0.05]) and n == pytest.approx(np.array([[0, 0, -1]] * 2))
    assert pos == pytest.approx(np.array([[-0.3, 0.1, 0.475], [0.3, 0.1, 0.475]]))
    d, pos, n = C((0.4, 0, 0.55), (0.3, 0, 0), 0.1)           # overhanging: the outer end is clipped to the face's side plane x = 0.5
    assert d == pytest.approx([-0.05, -0.05]) and pos == pytest.approx(np.array([[0.1, 0, 0.475], [0.5, 0, 0.475]]))
    a = 0.2 / math.sqrt(2)
    d, pos, n = C((0.65, 0, 0.65), (a, 0, -a), 0.25)          # across the edge x = z = 0.5, tangent to it: ONE edge contact, slot 1 off
    assert d[0] == pytest.approx(0.15 * math.sqrt(2) - 0.25, abs=1e-6) and d[1] == 1.0
    assert n[0] == pytest.approx([-1 / math.sqrt(2), 0, -1 / math.sqrt(2)], abs=1e-5)
    d, pos, n = C((0, 0, 1.5), (0.3, 0, 0), 0.1)              # far above: no support on the top face -> both slots off
    assert (d == 1.0).all()


def _two_cubes():
    spec = ModelSpec("cubes", [
        BodySpec("a", "world", pos=(0, 0, 2), joints=_free("a"), geoms=[GeomSpec(GEOM_BOX, (0.5, 0.5, 0.5), contype=1, conaffinity=1)]),
        BodySpec("b", "world", joints=_free("b"), geoms=[GeomSpec(GEOM_BOX, (0.5, 0.5, 0.5), contype=1, conaffinity=1)])], [], has_plane=False, free_root_z=0.0)
    t = compile_model(spec).t
    return Hull(t, 0, np.dtype(f64)), Hull(t, 1, np.dtype(f64))


def _rot(axis, angle):
    axis = np.asarray(axis, f64) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + math.sin(angle) * K + (1 - math.cos(angle)) * K @ K


def test_convex_convex_known_answers():
    """Round 6: two unit cubes, cube A placed against cube B (B at the origin of its own frame) - face on face, face on face turned by 45
    degrees, edge across edge, a corner pushed into a face, and apart."""
    A, B = _two_cubes()
    CC = lambda R, t: [x[0] for x in convex_convex(A, B, np.asarray([R], f64), np.asarray([t], f64))]
    # face on face, A on top of B, shifted: the overlap is the square [-0.3, 0.5] x [-0.4, 0.5]; 5 cm deep; the normal points from A down to B
    d, pos, n = CC(np.eye(3), (0.2, 0.1, 0.95))
    assert n == pytest.approx([0, 0, -1]) and d == pytest.approx([-0.05] * 4)
    corners = {(-0.3, -0.4), (-0.3, 0.5), (0.5, -0.4), (0.5, 0.5)}
    picked = set(map(tuple, np.round(pos[:, :2], 6).tolist()))
    # (_manifold_points' fourth pick ties between the remaining corner and the first one; argmax takes whichever candidate comes first - the
    # duplicate MJX's plane_convex switches off stays on here, as in _create_contact_manifold)
    assert picked <= corners and len(picked) >= 3
    assert np.allclose(np.abs(pos[:, 2]), 0.5, atol=1e-9)  # on the reference face (one of the two touching faces)
    # the same with A turned by 45 degrees about z: the overlap is an octagon; four of its corners, all 5 cm deep, spanning it
    d, pos, n = CC(_rot((0, 0, 1), math.pi / 4), (0, 0, 0.95))
    assert n == pytest.approx([0, 0, -1]) and d == pytest.approx([-0.05] * 4) and len({tuple(np.round(p, 6)) for p in pos}) >= 3
    assert np.all(np.abs(pos[:, :2]).max(1) <= 0.5 + 1e-9) and np.ptp(pos[:, 0]) > 0.7 and np.ptp(pos[:, 1]) > 0.7
    # edge across edge: A turned by 45 degrees about x and by 90 about z, so that its lowest EDGE (along y in B's frame ... along x after the turn) crosses
    # B's top edge at right angles; the least overlap is on the cross product of the two edges
    Rx = _rot((1, 0, 0), math.pi / 4)                      # A's edge (along x) is its lowest line, sqrt(0.5) below its centre
    Ry = _rot((0, 1, 0), math.pi / 4)                      # B's highest line would be an edge along y: tilt the whole frame instead - put A over B's edge
    d, pos, n = CC(Ry.T @ Rx, Ry.T @ np.array([0.0, 0.0, 2 * math.sqrt(0.5) - 0.04]))
    # (ONE contact; its depth is the clipped face point's depth behind the reference FACE - 45 degrees off the axis here, hence sqrt(2) x the
    # overlap along the axis: "for edge contacts we use the clipped face point ... for small penetration roughly the edge contact point")
    assert (d[1:] == 1.0).all() and d[0] == pytest.approx(-0.04 * math.sqrt(2), abs=1e-6)
    assert n == pytest.approx(Ry.T @ np.array([0, 0, -1.0]), abs=1e-6)
    # a corner of A pushed 3 cm into B's top face: ONE candidate lies behind the face, and _manifold_points' four picks all land on it (the
    # manifold of MJX's convex routines has no "first occurrence" mask like plane_convex's: four coincident contacts)
    Rc = _rot((1, -1, 0), math.atan(math.sqrt(2)))        # A's diagonal (1, 1, 1) turned onto -z: the corner points down
    d, pos, n = CC(Rc, (0.05, -0.1, 0.5 + math.sqrt(0.75) - 0.03))
    assert n == pytest.approx([0, 0, -1], abs=1e-9) and d == pytest.approx([-0.03] * 4, abs=1e-9)
    assert pos == pytest.approx(np.array([[0.05, -0.1, 0.5]] * 4), abs=1e-9)
    # apart by 10 cm: nothing
    d, pos, n = CC(np.eye(3), (0.0, 0.0, 1.1))
    assert (d > 0).all()


def hull_scene() -> ModelSpec:
    """Two boxes and a 20-vertex mesh hull, all free, no gravity, no ground, every pair a candidate: (box, box), (box, mesh) twice - twelve slots."""
    rng = np.random.default_rng(5)
    v = rng.normal(size=(20, 3))
    v = v / np.linalg.norm(v, axis=1, keepdims=True) * [0.3, 0.25, 0.2]
    return ModelSpec("hull_scene", [
        BodySpec("boxa", "world", pos=(0, 0, 0), mass=2.0, inertia=(0.02, 0.03, 0.04), joints=_free("ja"),
                 geoms=[GeomSpec(GEOM_BOX, (0.3, 0.2, 0.1), pos=(0.02, 0.0, 0.01), quat=(0.9, 0.1, 0.2, 0.3), contype=1, conaffinity=1)]),
        BodySpec("boxb", "world", pos=(1, 0, 0), mass=1.0, inertia=(0.01, 0.02, 0.02), joints=_free("jb"), geoms=[GeomSpec(GEOM_BOX, (0.15, 0.25, 0.2), contype=1, conaffinity=1)]),
        BodySpec("rock", "world", pos=(2, 0, 0), mass=1.5, inertia=(0.02, 0.02, 0.02), joints=_free("jr"), geoms=[GeomSpec(GEOM_MESH, (0,), vertices=v, contype=1, conaffinity=1)]),
    ], [], gravity=(0, 0, 0), has_plane=False, free_root_z=0.0)


def hull_scene_states(cm, N, rng, far=0.0):
    q = np.tile(cm.t["qpos0"], (N, 1))
    for b in range(3):
        x = rng.normal(size=(N, 4))
        q[:, 7 * b + 3:7 * b + 7] = x / np.linalg.norm(x, axis=1, keepdims=True)
    q[:, 0:3] = 0
    for base, lo, hi in ((7, 0.18 + far, 0.5), (14, 0.25 + far, 0.6)):
        d = rng.normal(size=(N, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        q[:, base:base + 3] = d * rng.uniform(lo, hi, size=(N, 1))
    return q


def test_hull_pair_contacts_match_the_oracle(be):
    """Round 6, MJX convex_convex (box / mesh against box / mesh, four slots a pair): the kernel's rows of all twelve slots against the oracle
    on 96 poses, as for the round pairs above - active sets equal, reference accelerations and Jacobian rows at 5e-4 on the poses where the
    float32 and float64 oracles agree (a tie between two axes or faces, or a clipped point on a side plane, is ill-conditioned)."""
    from test_kernels_physics import _probe

    cm = compile_model(hull_scene())
    assert cm.npair == 12 and cm.t["pair_body"].tolist() == [[1, 2]] * 4 + [[1, 3]] * 4 + [[2, 3]] * 4
    h, dims, _keep = be.model(cm)
    N = 96
    rng = np.random.default_rng(2)
    qpos, qvel = hull_scene_states(cm, N, rng), 0.2 * rng.standard_normal((N, cm.nv))
    q32 = [x.astype(f32) for x in (qpos, qvel, np.zeros((N, 1)), np.zeros((N, cm.nv)))]
    ref, ref32 = _reference(cm, q32), _reference(cm, q32, f32)
    got = _probe(be, h, cm, *q32)
    scale = lambda k: np.abs(ref[k]).max() + 1e-9
    good = (np.abs(ref32.efc_J - ref.efc_J).reshape(N, -1).max(1) <= 2e-4 * scale("efc_J")) & \
           (np.abs(ref32.efc_aref - ref.efc_aref).max(1) <= 2e-4 * scale("efc_aref")) & ((ref32.efc_D > 0) == (ref.efc_D > 0)).all(1)
    active = (ref.efc_D > 0).reshape(N, cm.ncon, 4)[:, :, 0]
    per_pair = active[good].reshape(-1, 3, 4).any(2).sum(0)
    # (the manifold is tie-prone: a clipped point appears in the candidate list several times, and which of the equal values argmax meets first
    # under float32 rounding decides the pick - about two thirds of the poses are well-conditioned)
    assert good.mean() >= 0.6 and per_pair.min() >= 8, (good.mean(), per_pair)
    assert ((got["efc_D"].reshape(N, -1) > 0) == (ref.efc_D > 0))[good].all()
    for k, tol in dict(qM=1e-5, xpos=1e-5).items():
        r, g = ref[k], got[k].reshape(ref[k].shape)
        assert np.abs(g[good] - r[good]).max() <= tol * scale(k), (k, np.abs(g[good] - r[good]).max() / scale(k))
    # The manifold's first three points (a: the first candidate, b: the farthest from a, c: the farthest from the line a-b) slot by slot.  The
    # FOURTH is the first maximum of the list [|(b - p) . bc| ..., |(a - p) . ac| ...]: with three distinct corners a, b, c the two halves tie
    # exactly (twice the triangle's area, computed two ways) and rounding decides between a duplicate of a and a duplicate of b - harmless,
    # and not a property the float32 / float64 oracles' agreement can vouch for (NumPy rounds both the same way): the kernel's fourth row
    # must be one of the oracle's four rows of that pair.
    rows = lambda x: x.reshape(N, cm.ncon, 4, -1)
    for k, tol in dict(efc_D=5e-4, efc_aref=5e-4, efc_J=5e-4).items():
        r, g = rows(ref[k])[good], rows(got[k].reshape(ref[k].shape))[good]
        r, g = r.reshape(-1, 3, 4, *r.shape[2:]), g.reshape(-1, 3, 4, *g.shape[2:])      # [good, pair, slot, pyramid row, ...]
        assert np.abs(g[:, :, :3] - r[:, :, :3]).max() <= tol * scale(k), (k, np.abs(g[:, :, :3] - r[:, :, :3]).max() / scale(k))
        d4 = np.abs(g[:, :, 3:4] - r).reshape(*r.shape[:3], -1).max(-1).min(-1)           # the fourth slot against each of the oracle's four
        assert d4.max() <= tol * scale(k), (k, "fourth slot", d4.max() / scale(k))
    be.lib.model_close(h)


# ---------------------------------------------------------------------------
# the kernel (emulator build / HIP) and the C++ twin against the oracle
# ---------------------------------------------------------------------------


def _reference(cm, q32, dtype=f64):
    N = q32[0].shape[0]
    ref = PhysState(qpos=q32[0].astype(f64), qvel=q32[1].astype(f64), ctrl=np.zeros((N, 0)), qacc_warmstart=np.zeros((N, cm.nv)), time=np.zeros(N))
    if dtype is not f64:
        ref = PhysState({k: v.astype(dtype) for k, v in ref.items()})
    Physics(cm.t, dtype).forward(ref)
    return ref


def test_convex_pair_contacts_match_the_oracle(be):
    """Constraint rows of all seven slots from 96 poses: which are active (efc_D), the reference acceleration (a function of the
    distance) and the Jacobian rows (normal and contact point).  A pose in which the float32 oracle itself departs from the float64 one
    (a tie between two faces or edges, a contact point on top of the sphere's centre) is ill-conditioned and left out of the Jacobian /
    solver comparison; at least 85 % of the poses must remain, and every hull slot must fire among them."""
    from test_kernels_physics import _cost, _probe

    cm = compile_model(scene())
    h, dims, _keep = be.model(cm)
    N = 96
    rng = np.random.default_rng(1)
    qpos, qvel = scene_states(cm, N, rng), 0.2 * rng.standard_normal((N, cm.nv))
    q32 = [x.astype(f32) for x in (qpos, qvel, np.zeros((N, 1)), np.zeros((N, cm.nv)))]
    ref, ref32 = _reference(cm, q32), _reference(cm, q32, f32)
    got = _probe(be, h, cm, *q32)
    scale = lambda k: np.abs(ref[k]).max() + 1e-9
    good = (np.abs(ref32.efc_J - ref.efc_J).reshape(N, -1).max(1) <= 2e-4 * scale("efc_J")) & \
           (np.abs(ref32.efc_aref - ref.efc_aref).max(1) <= 2e-4 * scale("efc_aref"))
    active = (ref.efc_D > 0).reshape(N, cm.ncon, 4)[:, :, 0]
    assert good.mean() >= 0.85 and active[good][:, 1:].sum(0).min() >= 5, (good.mean(), active[good].sum(0))  # (slot 0, ball against rod, is the old kind of pair)
    assert ((got["efc_D"].reshape(N, -1) > 0) == (ref.efc_D > 0))[good].all()
    for k, tol in dict(efc_D=5e-4, efc_aref=5e-4, efc_J=5e-4, qM=1e-5, xpos=1e-5).items():
        r, g = ref[k], got[k].reshape(ref[k].shape)
        assert np.abs(g[good] - r[good]).max() <= tol * scale(k), (k, np.abs(g[good] - r[good]).max() / scale(k))
    c_got, c_ref = _cost(ref, got["qacc"]), _cost(ref, ref.qacc)
    # (six CG iterations from a cold start on overlaps of up to 0.18 m do not converge; float32 and float64 stop at different points of
    # the same descent - the solver's envelope, DESIGN.md section 5 - while the rows they descend on agree to 5e-4)
    crel = np.abs(c_got - c_ref)[good] / (c_ref[good] + 1e-3)
    assert np.median(crel) <= 1e-3 and crel.max() <= 0.25 and np.all(c_got[good] <= _cost(ref, ref.qacc_smooth)[good] * (1 + 1e-5) + 1e-6), (np.median(crel), crel.max())
    shallow = good & (ref.con_dist.min(1) > -0.05)  # (the solver's result itself: where no overlap is deeper than 5 cm)
    rel = (np.abs(got["qacc"] - ref.qacc).max(1) / (np.abs(ref.qacc).max(1) + 1e-9))[shallow]
    assert shallow.mean() >= 0.4 and active[shallow].any(1).sum() >= 15 and np.median(rel) <= 5e-3 and rel.max() <= 0.3, (shallow.mean(), np.median(rel), rel.max())
    be.lib.model_close(h)


def test_twin_steps_the_convex_scene_like_the_oracle():
    """One environment step of the C++ twin from the same poses: state after the step against the oracle's (contacts that fire change the
    velocities of both bodies; a wrong normal or contact point shows up in the angular part)."""
    from oracle.cpu_twin import RewardCfg as TwinReward, Twin

    cm = compile_model(scene())
    N = 64
    rng = np.random.default_rng(2)
    qpos, qvel = scene_states(cm, N, rng).astype(f32), (0.2 * rng.standard_normal((N, cm.nv))).astype(f32)
    ph = Physics(cm.t)
    d = ph.pipeline_init(qpos.astype(f64), qvel.astype(f64))
    assert ((d["con_dist"] < 0).sum(0)[1:] >= 4).all()
    shallow = d["con_dist"].min(1) > -0.05                      # (deeper overlaps: six CG iterations stop far from the optimum, in either precision)
    warm = d["qacc_warmstart"].astype(f32)                    # (pipeline_init ends with a forward pass: its qacc is the step's warm start)
    d = ph.pipeline_step(d, np.zeros((N, 0)))
    tw = Twin(cm, reward=TwinReward(-100.0, 100.0, 2.0, 0.2, 0.5, 0.1, 4.0, 1.0, 1.25))  # (no height band: nothing is reset by the wrapper)
    tw.reset(N)
    tw.state[:, :cm.nq] = qpos
    tw.state[:, cm.nq:cm.nq + cm.nv] = qvel
    tw.state[:, tw.obs_pad:tw.obs_pad + cm.nv] = warm
    tw.step(np.zeros((N, 0), f32))
    dv = np.abs(tw.state[:, cm.nq:cm.nq + cm.nv] - d["qvel"]).max(1)
    moved = np.abs(d["qvel"] - qvel).max(1)
    assert shallow.sum() >= N // 3 and (moved[shallow] > 0.05).sum() >= 10   # contacts did act in the poses that are compared
    rel = dv[shallow] / (moved[shallow] + 0.05)
    assert np.median(rel) <= 2e-3 and np.quantile(rel, 0.9) <= 0.05 and rel.max() <= 0.5, (np.median(rel), np.quantile(rel, 0.9), rel.max())
    assert np.abs(tw.state[:, :cm.nq] - d["qpos"])[shallow].max() <= 2e-3
    tw.close()


def test_twin_steps_the_hull_scene_like_the_oracle():
    """The same for the hull pairs of round 6 (box / mesh against box / mesh, MJX convex_convex): one step of the C++ twin from 64 poses of the
    two boxes and the rock against the oracle's, on the poses whose overlaps are shallow."""
    from oracle.cpu_twin import RewardCfg as TwinReward, Twin

    cm = compile_model(hull_scene())
    N = 160
    rng = np.random.default_rng(3)
    qpos, qvel = hull_scene_states(cm, N, rng, far=0.12).astype(f32), (0.2 * rng.standard_normal((N, cm.nv))).astype(f32)
    ph = Physics(cm.t)
    d = ph.pipeline_init(qpos.astype(f64), qvel.astype(f64))
    shallow = (d["con_dist"].min(1) > -0.05) & (d["con_dist"].min(1) < 0)
    warm = d["qacc_warmstart"].astype(f32)
    d = ph.pipeline_step(d, np.zeros((N, 0)))
    tw = Twin(cm, reward=TwinReward(-100.0, 100.0, 2.0, 0.2, 0.5, 0.1, 4.0, 1.0, 1.25))
    tw.reset(N)
    tw.state[:, :cm.nq] = qpos
    tw.state[:, cm.nq:cm.nq + cm.nv] = qvel
    tw.state[:, tw.obs_pad:tw.obs_pad + cm.nv] = warm
    tw.step(np.zeros((N, 0), f32))
    dv = np.abs(tw.state[:, cm.nq:cm.nq + cm.nv] - d["qvel"]).max(1)
    moved = np.abs(d["qvel"] - qvel).max(1)
    assert shallow.sum() >= 8 and (moved[shallow] > 0.05).sum() >= 5, (shallow.sum(), (moved[shallow] > 0.05).sum())
    rel = dv[shallow] / (moved[shallow] + 0.05)
    assert np.median(rel) <= 5e-3 and np.quantile(rel, 0.75) <= 0.1, (np.median(rel), np.quantile(rel, 0.75), rel.max())
    tw.close()
