"""Robot model description -> flat model tables -> device blob.

The reference obtains its robot from the K-Scale API at run time
(`minppo/env.py:27-50`, id `5eb3cb7f23232298`), hands the MJCF to MuJoCo,
overrides the solver (`env.py:95-97`: CG, 6 iterations, 6 line-search
iterations) and converts it with `brax.io.mjcf.load_model` (`env.py:100`).
None of that is reachable on the target (no network, no MuJoCo), so the engine
has its own model pipeline:

  ModelSpec (bodies / joints / geoms / actuators, MJCF-like)   [this file]
      -> compile_model(): flat tables + constants MuJoCo's compiler would
         derive (`dof_invweight0`, `body_invweight0`, `stat.meaninertia`,
         tree bookkeeping)                                    [this file]
      -> to_blob(): the int32/float32 blob `mppo_model_open` consumes
         (layout documented in include/minppo_hip.h)

Built-in robots are *stand-ins* and are labelled as such everywhere:
  synth_stompy_pro  : free root + 2 legs x 5 hinges   (nq=17 nv=16 nu=10, O=225)
  synth_stompy_full : + 2 arms x 5 hinges             (nq=27 nv=26 nu=20, O=415)
`kscale_id: 5eb3cb7f23232298` resolves to synth_stompy_pro.

Conventions follow MuJoCo: quaternions are (w,x,y,z); body 0 is the world;
spatial vectors are [rotational(3), translational(3)].
"""

from __future__ import annotations

import math
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

JNT_FREE, JNT_HINGE, JNT_SLIDE = 0, 2, 3  # (ball=1 unsupported)
GEOM_SPHERE, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = 2, 3, 5, 6, 7  # MuJoCo's mjtGeom numbering

MJ_MINVAL = 1e-15
MAX_CONVEX_VERTS = 64  # hull vertices of one mesh collider (the kernel scans them five times per step)
MAX_BODIES = 128  # subtree sets are two 64-bit words per body (one up to 64 bodies); dofs: one word, 64

BLOB_MAGIC = 0x4D50504F  # "MPPO"
BLOB_VERSION = 7  # 7: hull section carries the hulls' edge directions (convex_convex: box / mesh against box / mesh); 6: dof_actfrcrange (joint actuatorfrcrange); 2: header word include_c_vals; 3: geom-geom pairs (npair, pair_body, pair_geom) and con_axis; 4: convex (mesh) geoms against the plane; 5: hull section (sphere / capsule against box / mesh)


# ---------------------------------------------------------------------------
# specification objects
# ---------------------------------------------------------------------------


@dataclass
class JointSpec:
    name: str
    type: int = JNT_HINGE
    pos: Sequence[float] = (0.0, 0.0, 0.0)
    axis: Sequence[float] = (0.0, 0.0, 1.0)
    range: Optional[Tuple[float, float]] = None
    damping: float = 0.0
    armature: float = 0.0
    stiffness: float = 0.0
    ref: float = 0.0  # qpos0 for hinge/slide
    actuatorfrcrange: Optional[Tuple[float, float]] = None  # MJCF actuatorfrcrange: the total actuator force on this (hinge / slide) joint is clamped to it
    springref: Optional[float] = None  # position at which the joint spring (stiffness) is at rest: MuJoCo's qpos_spring; None = at `ref` (the built-in robots)


@dataclass
class GeomSpec:
    type: int
    size: Sequence[float]  # sphere: (r,), capsule / cylinder: (r, half_length) along local z, box: the three half sizes
    pos: Sequence[float] = (0.0, 0.0, 0.0)
    quat: Sequence[float] = (1.0, 0.0, 0.0, 0.0)
    friction: Sequence[float] = (1.0, 0.005, 0.0001)
    # MuJoCo's collision masks: two geoms are candidates iff (contype1 & conaffinity2) | (contype2 & conaffinity1).  The defaults
    # here (0 / 1, against the plane's 1 / 1) make a geom collide with the ground only; the MJCF loader passes the file's values
    # (MuJoCo's own default is 1 / 1: every pair that is not parent-child).
    contype: int = 0
    conaffinity: int = 1
    # GEOM_MESH: the vertices of the mesh's CONVEX HULL in the geom's own frame ([V, 3]; `size` is unused).  A mesh collides as its
    # convex hull, like in MuJoCo / MJX: with the ground plane, and with the spheres and capsules of other bodies (not with boxes / meshes).
    vertices: Optional[Sequence[Sequence[float]]] = None
    name: str = ""  # (what an explicit contact pair - ModelSpec.contact_pairs - refers to)


@dataclass
class BodySpec:
    name: str
    parent: str  # "world" or a body name
    pos: Sequence[float] = (0.0, 0.0, 0.0)
    quat: Sequence[float] = (1.0, 0.0, 0.0, 0.0)
    mass: float = 1.0
    inertia: Sequence[float] = (0.01, 0.01, 0.01)  # diagonal, in the inertial frame
    ipos: Sequence[float] = (0.0, 0.0, 0.0)
    iquat: Sequence[float] = (1.0, 0.0, 0.0, 0.0)
    joints: List[JointSpec] = field(default_factory=list)
    geoms: List[GeomSpec] = field(default_factory=list)


@dataclass
class ActuatorSpec:
    joint: str
    gear: float = 1.0
    kp: float = 0.0  # position servo: force = kp*(ctrl - length) - kv*velocity ; kp=0 -> motor (gain 1)
    kv: float = 0.0
    # MJCF <general gaintype="fixed" biastype="affine"> / <velocity>: force = gain * ctrl + bias[0] + bias[1] * length + bias[2] * velocity; when
    # `gain` is given it replaces what kp / kv would produce
    gain: Optional[float] = None
    bias: Sequence[float] = (0.0, 0.0, 0.0)
    ctrlrange: Optional[Tuple[float, float]] = None
    forcerange: Optional[Tuple[float, float]] = None


@dataclass
class ModelSpec:
    name: str
    bodies: List[BodySpec]
    actuators: List[ActuatorSpec]
    timestep: float = 0.002
    gravity: Sequence[float] = (0.0, 0.0, -9.81)
    # solver settings the reference forces (env.py:95-97); tolerance / ls_tolerance are MuJoCo defaults
    iterations: int = 6
    ls_iterations: int = 6
    tolerance: float = 1e-8
    ls_tolerance: float = 0.01
    impratio: float = 1.0
    contact_solref: Sequence[float] = (0.02, 1.0)
    contact_solimp: Sequence[float] = (0.9, 0.95, 0.001, 0.5, 2.0)
    limit_solref: Sequence[float] = (0.02, 1.0)
    limit_solimp: Sequence[float] = (0.9, 0.95, 0.001, 0.5, 2.0)
    plane_friction: Sequence[float] = (1.0, 0.005, 0.0001)
    plane_z: float = 0.0
    plane_contype: int = 1
    plane_conaffinity: int = 1
    has_plane: bool = True
    free_root_z: float = 1.0  # qpos0[2] of the (first) free joint
    meaninertia: Optional[float] = None  # MJCF <statistic meaninertia>: overrides the value derived from M(qpos0) (it scales the solver's tolerance)
    contact_excludes: List[Tuple[str, str]] = field(default_factory=list)  # MJCF <contact><exclude body1 body2/>: no geom pairs between these bodies
    # MJCF <contact><pair geom1 geom2 friction/>: geom pairs that collide whatever their masks, their bodies' kinship or an <exclude> say, with
    # a sliding friction of their own (MuJoCo: an explicit pair's parameters are its own, the geoms' are not used - the MJCF loader passes the pair's value or MuJoCo's
    # pair default, 1; None here: the larger of the two geoms' values, as for a generated pair).
    # One of the two may be the ground plane (`plane_name`): the geom then gets its ground contact slots.
    contact_pairs: List[Tuple[str, str, Optional[float]]] = field(default_factory=list)
    plane_name: str = ""


# ---------------------------------------------------------------------------
# small quaternion helpers (host side, float64)
# ---------------------------------------------------------------------------


def _qmul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array(
        [
            aw * bw - ax * bx - ay * by - az * bz,
            aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw,
        ]
    )


def _qmat(q):
    w, x, y, z = q
    return np.array(
        [
            [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
            [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
            [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z],
        ]
    )


def _qrot(q, v):
    return _qmat(q) @ np.asarray(v, dtype=np.float64)


def _normalize(v):
    v = np.asarray(v, dtype=np.float64)
    n = np.linalg.norm(v)
    return v / n if n > 0 else v


# ---------------------------------------------------------------------------
# compiled model
# ---------------------------------------------------------------------------


@dataclass
class CompiledModel:
    name: str
    t: Dict[str, np.ndarray]  # tables (float64 / int32); scalars stored as 0-d arrays
    body_names: List[str]
    joint_names: List[str]
    meaninertia_override: Optional[float] = None

    # -- dims -------------------------------------------------------------
    @property
    def nq(self) -> int:
        return int(self.t["nq"])

    @property
    def nv(self) -> int:
        return int(self.t["nv"])

    @property
    def nu(self) -> int:
        return int(self.t["nu"])

    @property
    def nbody(self) -> int:
        return int(self.t["nbody"])

    @property
    def njnt(self) -> int:
        return int(self.t["njnt"])

    @property
    def ncon(self) -> int:
        return int(self.t["ncon"])

    @property
    def npair(self) -> int:
        """geom-geom candidates: the last `npair` of the `ncon` contact slots (the first ncon - npair are ground contacts)."""
        return int(self.t["npair"])

    @property
    def nlimit(self) -> int:
        return int(self.t["nlimit"])

    @property
    def ncvx(self) -> int:
        """convex (mesh) geoms that can meet the ground plane: four contact slots each."""
        return int(self.t["ncvx"])

    @property
    def nefc(self) -> int:
        return self.nlimit + 4 * self.ncon

    def obs_size(self, include_c_vals: bool = True) -> int:
        """`get_obs` width (reference `minppo/env.py:245-261`)."""
        if include_c_vals:
            return self.nq + self.nv + 16 * (self.nbody - 1) + self.nv
        return self.nq + self.nv + self.nv

    @property
    def dt(self) -> float:
        return float(self.t["timestep"])  # x n_frames at the env level

    def to_blob(self, include_c_vals: bool = True) -> bytes:
        """The model as the engine reads it.  `include_c_vals` (reference `environment.include_c_vals`, `env.py:246-259`) selects
        what an observation is: qpos, qvel, cinert[1:], cvel[1:], qfrc_actuator - or qpos, qvel, qfrc_actuator only."""
        return _to_blob(self, include_c_vals)


def hull_topology(verts: np.ndarray) -> Tuple[List[List[int]], np.ndarray, np.ndarray, np.ndarray]:
    """Faces, outward unit normals, edges and the two face normals beside every edge of the convex hull of `verts` ([V, 3], every one
    of them a hull vertex) - what MJX keeps of a convex geom for sphere_convex / capsule_convex (its mesh.py: the hull's triangles,
    coplanar ones merged into polygons; a box is the six quads this gives for its corners).  A face lists its vertices counter-clockwise
    seen from outside.  Returns (faces, normals [F, 3], edges [E, 2] with i < j, edge_normals [E, 2, 3])."""
    from scipy.spatial import ConvexHull

    v = np.asarray(verts, np.float64).reshape(-1, 3)
    hull = ConvexHull(v)
    tris = [tuple(int(i) for i in tri) for tri in hull.simplices]
    nrm = hull.equations[:, :3] / np.linalg.norm(hull.equations[:, :3], axis=1, keepdims=True)
    parent = list(range(len(tris)))

    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i

    by_edge: Dict[Tuple[int, int], List[int]] = {}
    for ti, tri in enumerate(tris):
        for a, b in ((tri[0], tri[1]), (tri[1], tri[2]), (tri[2], tri[0])):
            by_edge.setdefault((min(a, b), max(a, b)), []).append(ti)
    for ts in by_edge.values():
        for other in ts[1:]:
            if np.dot(nrm[ts[0]], nrm[other]) > 1.0 - 1e-9:  # coplanar neighbours: one polygon
                parent[find(other)] = find(ts[0])
    groups: Dict[int, List[int]] = {}
    for ti in range(len(tris)):
        groups.setdefault(find(ti), []).append(ti)
    faces: List[List[int]] = []
    normals: List[np.ndarray] = []
    for members in sorted(groups.values(), key=lambda m: min(min(tris[ti]) for ti in m) * len(v) * len(v) + min(m)):
        n = nrm[members].mean(0)
        n /= np.linalg.norm(n)
        idx = sorted({i for ti in members for i in tris[ti]})
        c = v[idx].mean(0)
        u = v[idx[0]] - c
        u -= n * np.dot(u, n)
        u /= np.linalg.norm(u)
        w = np.cross(n, u)  # (u, w, n) right-handed: increasing angle = counter-clockwise seen from outside
        ang = [math.atan2(np.dot(v[i] - c, w), np.dot(v[i] - c, u)) for i in idx]
        order = [i for _, i in sorted(zip(ang, idx))]
        k0 = order.index(min(order))
        faces.append(order[k0:] + order[:k0])
        normals.append(n)
    edge_faces: Dict[Tuple[int, int], List[int]] = {}
    for fi, f in enumerate(faces):
        for a, b in zip(f, f[1:] + f[:1]):
            edge_faces.setdefault((min(a, b), max(a, b)), []).append(fi)
    edges = sorted(edge_faces)
    if any(len(edge_faces[e]) != 2 for e in edges):
        raise ValueError("convex hull: an edge that does not separate exactly two faces (degenerate mesh)")
    return faces, np.asarray(normals), np.asarray(edges, np.int32).reshape(-1, 2), np.asarray([[normals[edge_faces[e][0]], normals[edge_faces[e][1]]] for e in edges]).reshape(-1, 2, 3)


def compile_model(spec: ModelSpec) -> CompiledModel:
    """Flattens a ModelSpec and derives the constants MuJoCo's compiler would
    (`mj_setConst`: `dof_invweight0`, `body_invweight0`, `stat.meaninertia`)."""
    names = ["world"] + [b.name for b in spec.bodies]
    if len(set(names)) != len(names):
        raise ValueError("duplicate body names")
    nbody = len(names)
    body_parent = np.zeros(nbody, np.int32)
    body_pos = np.zeros((nbody, 3))
    body_quat = np.tile(np.array([1.0, 0, 0, 0]), (nbody, 1))
    body_ipos = np.zeros((nbody, 3))
    body_iquat = np.tile(np.array([1.0, 0, 0, 0]), (nbody, 1))
    body_mass = np.zeros(nbody)
    body_inertia = np.zeros((nbody, 3))
    body_jntadr = np.full(nbody, -1, np.int32)
    body_jntnum = np.zeros(nbody, np.int32)
    body_dofadr = np.full(nbody, -1, np.int32)
    body_dofnum = np.zeros(nbody, np.int32)

    jnt_type, jnt_qposadr, jnt_dofadr, jnt_bodyid = [], [], [], []
    jnt_pos, jnt_axis, jnt_range, jnt_limited, jnt_stiffness = [], [], [], [], []
    joint_names: List[str] = []
    dof_bodyid, dof_jntid, dof_armature, dof_damping = [], [], [], []
    dof_actfrcrange: List[List[float]] = []
    F32_MAX = float(np.finfo(np.float32).max)
    qpos0: List[float] = []
    qpos_spring: List[float] = []
    geoms = []  # (type, bodyid, pos, quat, size, friction, contype, conaffinity)

    nq = nv = 0
    seen_free = False
    for bi, b in enumerate(spec.bodies, start=1):
        if b.parent not in names[:bi]:
            raise ValueError(f"body {b.name}: parent {b.parent!r} must be defined earlier")
        body_parent[bi] = names.index(b.parent)
        body_pos[bi] = b.pos
        body_quat[bi] = _normalize(b.quat)
        body_ipos[bi] = b.ipos
        body_iquat[bi] = _normalize(b.iquat)
        body_mass[bi] = b.mass
        body_inertia[bi] = b.inertia
        if b.joints:
            body_jntadr[bi] = len(jnt_type)
            body_dofadr[bi] = nv
        body_jntnum[bi] = len(b.joints)
        for j in b.joints:
            if j.name in joint_names:
                raise ValueError(f"duplicate joint name {j.name}")
            joint_names.append(j.name)
            jnt_type.append(j.type)
            jnt_qposadr.append(nq)
            jnt_dofadr.append(nv)
            jnt_bodyid.append(bi)
            jnt_pos.append(list(j.pos))
            jnt_axis.append(list(_normalize(j.axis)))
            jnt_range.append(list(j.range) if j.range is not None else [0.0, 0.0])
            jnt_limited.append(1 if j.range is not None else 0)
            jnt_stiffness.append(j.stiffness)
            jid = len(jnt_type) - 1
            if j.type == JNT_FREE:
                if body_parent[bi] != 0 or len(b.joints) != 1:
                    raise ValueError("free joint only on a top-level body, alone")
                # the first free body stands at free_root_z (the robot's root); further free bodies keep their own height
                qpos0 += [b.pos[0], b.pos[1], spec.free_root_z if not seen_free else b.pos[2], *body_quat[bi]]
                qpos_spring += qpos0[-7:]  # (a free joint has no spring)
                seen_free = True
                nq += 7
                for _ in range(6):
                    dof_bodyid.append(bi)
                    dof_jntid.append(jid)
                    dof_armature.append(j.armature)
                    dof_damping.append(j.damping)
                    dof_actfrcrange.append([-F32_MAX, F32_MAX])
                nv += 6
            elif j.type in (JNT_HINGE, JNT_SLIDE):
                qpos0.append(j.ref)
                qpos_spring.append(j.ref if j.springref is None else j.springref)
                nq += 1
                dof_bodyid.append(bi)
                dof_jntid.append(jid)
                dof_armature.append(j.armature)
                dof_damping.append(j.damping)
                if j.actuatorfrcrange is not None and not (j.actuatorfrcrange[0] < j.actuatorfrcrange[1]):
                    raise ValueError(f"joint {j.name}: actuatorfrcrange {tuple(j.actuatorfrcrange)} is empty")
                dof_actfrcrange.append([-F32_MAX, F32_MAX] if j.actuatorfrcrange is None else [float(j.actuatorfrcrange[0]), float(j.actuatorfrcrange[1])])
                nv += 1
            else:
                raise ValueError(f"unsupported joint type {j.type}")
        body_dofnum[bi] = nv - (body_dofadr[bi] if body_dofadr[bi] >= 0 else nv)
        for g in b.geoms:
            size = list(g.size) + [0.0] * (3 - len(g.size))
            if g.type == GEOM_MESH:
                if g.vertices is None or len(g.vertices) < 4:
                    raise ValueError(f"body {b.name}: a mesh geom needs the (at least four) vertices of its convex hull")
                if len(g.vertices) > MAX_CONVEX_VERTS:
                    raise ValueError(f"body {b.name}: a mesh collider with {len(g.vertices)} hull vertices (limit {MAX_CONVEX_VERTS}: decimate the collision mesh)")
            geoms.append((g.type, bi, list(g.pos), list(_normalize(g.quat)), size, list(g.friction), int(g.contype), int(g.conaffinity),
                          None if g.vertices is None else np.asarray(g.vertices, np.float64).reshape(-1, 3), g.name))

    njnt = len(jnt_type)
    # dof_parentid: previous dof in the same body, else last dof of the nearest ancestor with dofs
    dof_parentid = np.full(nv, -1, np.int32)
    body_lastdof = np.full(nbody, -1, np.int32)
    for bi in range(1, nbody):
        last = body_lastdof[body_parent[bi]]
        if body_dofnum[bi] > 0:
            for d in range(body_dofadr[bi], body_dofadr[bi] + body_dofnum[bi]):
                dof_parentid[d] = last
                last = d
        body_lastdof[bi] = last
    body_rootid = np.zeros(nbody, np.int32)
    for bi in range(1, nbody):
        body_rootid[bi] = bi if body_parent[bi] == 0 else body_rootid[body_parent[bi]]
    depth = np.zeros(nbody, np.int32)
    for bi in range(1, nbody):
        depth[bi] = depth[body_parent[bi]] + 1

    if nbody > MAX_BODIES or nv > 64:
        raise ValueError(f"the physics kernel supports at most {MAX_BODIES} bodies (the world included) and 64 dofs (bitmask topology tables); "
                         f"this model has {nbody} bodies and {nv} dofs")
    # bodies grouped by depth (level-synchronous kinematics)
    nlevel = int(depth.max()) if nbody > 1 else 0
    level_adr = np.zeros(nlevel + 1, np.int32)
    level_body: List[int] = []
    for lv in range(1, nlevel + 1):
        level_adr[lv - 1] = len(level_body)
        level_body += [b for b in range(1, nbody) if depth[b] == lv]
    level_adr[nlevel] = len(level_body)
    root_body = [b for b in range(1, nbody) if body_parent[b] == 0]
    # bitmask topology: subtree(b) as a set of bodies; ancestor-or-own dofs of body b;
    # dofs whose motion precedes dof j in mj_comVel (used for cdof_dot)
    subtree = [0] * nbody  # (Python integers: up to 128 bits)
    for b in range(nbody - 1, 0, -1):
        subtree[b] |= 1 << b
        subtree[body_parent[b]] |= subtree[b]
    ancdof = np.zeros(nbody, np.uint64)
    for b in range(1, nbody):
        ancdof[b] = ancdof[body_parent[b]]
        for d in range(body_dofadr[b], body_dofadr[b] + body_dofnum[b]) if body_dofnum[b] > 0 else []:
            ancdof[b] |= np.uint64(1) << np.uint64(d)
    velmask = np.zeros(nv, np.uint64)
    for j in range(njnt):
        b, da = jnt_bodyid[j], jnt_dofadr[j]
        before = ancdof[body_parent[b]]
        for jj in range(body_jntadr[b], j):  # earlier joints of the same body
            nd = 6 if jnt_type[jj] == JNT_FREE else 1
            for d in range(jnt_dofadr[jj], jnt_dofadr[jj] + nd):
                before |= np.uint64(1) << np.uint64(d)
        if jnt_type[j] == JNT_FREE:
            for k in range(3):
                velmask[da + k] = before  # value unused: translational cdof_dot is 0
            trans = before
            for k in range(3):
                trans |= np.uint64(1) << np.uint64(da + k)
            for k in range(3, 6):
                velmask[da + k] = trans  # all three rotational cdof_dot use the velocity after translation only
        else:
            velmask[da] = before
    dof_qposadr = np.full(nv, -1, np.int32)
    for j in range(njnt):
        if jnt_type[j] != JNT_FREE:
            dof_qposadr[jnt_dofadr[j]] = jnt_qposadr[j]

    def _m64(a):
        a = np.asarray(a, np.uint64)
        return np.stack([(a & np.uint64(0xFFFFFFFF)).astype(np.uint32), (a >> np.uint64(32)).astype(np.uint32)], -1).astype(np.uint32).view(np.int32).reshape(-1)

    # actuators (joint transmission on hinge/slide only)
    nu = len(spec.actuators)
    act_dofid = np.zeros(nu, np.int32)
    act_qposadr = np.zeros(nu, np.int32)
    act_gear = np.zeros(nu)
    act_gain = np.zeros(nu)
    act_bias = np.zeros((nu, 3))
    act_ctrlrange = np.zeros((nu, 2))
    act_ctrllimited = np.zeros(nu, np.int32)
    act_forcerange = np.zeros((nu, 2))
    act_forcelimited = np.zeros(nu, np.int32)
    for ai, a in enumerate(spec.actuators):
        jid = joint_names.index(a.joint)
        if jnt_type[jid] == JNT_FREE:
            raise ValueError("actuator on a free joint is not supported")
        act_dofid[ai] = jnt_dofadr[jid]
        act_qposadr[ai] = jnt_qposadr[jid]
        act_gear[ai] = a.gear
        if a.gain is not None:
            act_gain[ai] = a.gain
            act_bias[ai] = list(a.bias)
        elif a.kp > 0:
            act_gain[ai] = a.kp
            act_bias[ai] = [0.0, -a.kp, -a.kv]
        else:
            act_gain[ai] = 1.0
        if a.ctrlrange is not None:
            act_ctrlrange[ai] = a.ctrlrange
            act_ctrllimited[ai] = 1
        if a.forcerange is not None:
            act_forcerange[ai] = a.forcerange
            act_forcelimited[ai] = 1

    # collision candidates, MJX-style static slots.  First every sphere / capsule end / hull (box, mesh) / cylinder against the ground plane ...
    def _masks_match(ct1, ca1, ct2, ca2):
        return bool((ct1 & ca2) | (ct2 & ca1))

    con_bodyid, con_lpos, con_radius, con_friction, con_axis, con_cvx = [], [], [], [], [], []
    cvx_body, cvx_vadr, cvx_vert = [], [0], []
    # explicit pairs (MJCF <contact><pair>): by geom name; with the ground plane or between two geoms of different bodies
    geom_index = {g[9]: k for k, g in enumerate(geoms) if g[9]}
    pair_with_plane: Dict[int, Optional[float]] = {}
    pair_of_geoms: Dict[frozenset, Optional[float]] = {}
    for n1, n2, mu in spec.contact_pairs:
        for n in (n1, n2):
            if n not in geom_index and not (spec.has_plane and spec.plane_name and n == spec.plane_name):
                raise ValueError(f"contact pair ({n1!r}, {n2!r}): unknown geom {n!r} (a pair names collision geoms of bodies, or the ground plane)")
        if mu is not None and not (mu >= 0):
            raise ValueError(f"contact pair ({n1!r}, {n2!r}): friction {mu}")
        on_plane = [n for n in (n1, n2) if n not in geom_index]
        if len(on_plane) == 2 or n1 == n2:
            raise ValueError(f"contact pair ({n1!r}, {n2!r}): a pair needs two different geoms, at most one of them the ground plane")
        if on_plane:
            pair_with_plane[geom_index[n2 if on_plane[0] == n1 else n1]] = mu
        else:
            i_, j_ = geom_index[n1], geom_index[n2]
            if geoms[i_][1] == geoms[j_][1]:
                raise ValueError(f"contact pair ({n1!r}, {n2!r}): both geoms belong to the same body")
            pair_of_geoms[frozenset((i_, j_))] = mu
    for gk, (gt, bi, gpos, gquat, gsize, gfri, gct, gca, gverts, _gname) in enumerate(geoms):
        if gt not in (GEOM_SPHERE, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH):
            raise ValueError(f"unsupported geom type {gt}")
        if not spec.has_plane or not (_masks_match(spec.plane_contype, spec.plane_conaffinity, gct, gca) or gk in pair_with_plane):
            continue
        fri = np.maximum(np.asarray(gfri), np.asarray(spec.plane_friction))
        if pair_with_plane.get(gk) is not None:
            fri = np.asarray([pair_with_plane[gk], fri[1], fri[2]])  # (an explicit pair's sliding friction is its own)
        axis_l = np.zeros(3)
        if gt in (GEOM_MESH, GEOM_BOX):
            # a convex hull against the plane, MJX collision_convex.plane_convex: FOUR slots per geom; which hull vertices fill them
            # is decided every step (the deepest ones, spread out: _manifold_points), duplicates are switched off.  A BOX takes the same
            # route in MJX (its collision table sends (plane, box) to plane_convex: the box is a convex mesh of its eight corners, in the
            # order of mesh.box - x outermost, z innermost) - NOT the C engine's mjc_PlaneBox: only corners within 1 mm of the deepest one
            # are candidates.  Until the end of round 5 a box was eight independent corner contacts here (the C engine's behaviour).
            if gt == GEOM_BOX:
                gverts = np.asarray([[sx * gsize[0], sy * gsize[1], sz * gsize[2]] for sx in (-1.0, 1.0) for sy in (-1.0, 1.0) for sz in (-1.0, 1.0)])
            k = len(cvx_body)
            cvx_body.append(bi)
            for v in gverts:
                cvx_vert.append(list(np.asarray(gpos) + _qrot(gquat, v)))  # body frame
            cvx_vadr.append(len(cvx_vert))
            for j in range(4):
                con_bodyid.append(bi); con_lpos.append([0.0, 0.0, 0.0]); con_radius.append(0.0); con_friction.append(list(fri)); con_axis.append([0.0, 0.0, 0.0])
                con_cvx.append(4 * k + j)
            continue
        if gt == GEOM_CYLINDER:
            # a cylinder against the plane, MJX collision_primitive.plane_cylinder: THREE slots per geom, placed every step - the rim point
            # nearest to the plane and two more on the same rim (a triangle), or, lying on its side, the nearest points of both rims.
            # con_cvx = -2 marks the first slot (it computes all three), -3 / -4 the others; con_axis carries the half-axis vector (first
            # slot) and the geom's x axis (second slot: the direction MJX picks when the disk is parallel to the plane), body frame
            for j in range(3):
                con_bodyid.append(bi); con_lpos.append(list(gpos)); con_radius.append(gsize[0]); con_friction.append(list(fri))
                con_axis.append(list(_qrot(gquat, [0, 0, 1.0]) * gsize[1]) if j == 0 else list(_qrot(gquat, [1.0, 0, 0])) if j == 1 else [0.0, 0.0, 0.0])
                con_cvx.append(-2 - j)
            continue
        if gt == GEOM_SPHERE:
            ends = [np.asarray(gpos, dtype=np.float64)]
        else:
            axis_l = _qrot(gquat, [0, 0, 1.0])  # the two end contacts share a frame whose first tangent follows this axis
            axis = axis_l * gsize[1]
            ends = [np.asarray(gpos) + axis, np.asarray(gpos) - axis]
        for e in ends:
            con_bodyid.append(bi)
            con_lpos.append(list(e))
            con_radius.append(gsize[0])
            con_friction.append(list(fri))
            con_axis.append(list(axis_l))
            con_cvx.append(-1)
    nplane = len(con_bodyid)
    # ... then geom-geom pairs between different bodies, filtered as MuJoCo filters them: same weld group and parent-child weld
    # groups are skipped (mj_filterBodyPair), then the contype / conaffinity masks.  geom1 is the one with the smaller type id
    # (sphere < capsule < box < mesh); groups are ordered (sphere, sphere), (sphere, capsule), (sphere, box), (sphere, mesh),
    # (capsule, capsule), (capsule, box), (capsule, mesh) like MJX's collision-function table.  A box or a mesh meets a sphere / capsule
    # as a convex hull (MJX sphere_convex: one contact; capsule_convex: two); box / mesh against box / mesh: convex_convex, four (round 6).
    weld = np.arange(nbody)
    for b in range(1, nbody):
        if body_jntnum[b] == 0:
            weld[b] = weld[body_parent[b]]
    pair_rows = []
    excluded = set()
    for n1, n2 in spec.contact_excludes:
        if n1 not in names or n2 not in names:
            raise ValueError(f"contact exclude ({n1!r}, {n2!r}): unknown body")
        excluded.add(frozenset((names.index(n1), names.index(n2))))
    for i in range(len(geoms)):
        for j in range(i + 1, len(geoms)):
            gi, gj = geoms[i], geoms[j]
            if gi[0] > gj[0]:
                gi, gj = gj, gi
            w1, w2 = weld[gi[1]], weld[gj[1]]
            key = frozenset((i, j))
            if key not in pair_of_geoms:  # (an explicit pair passes every filter - MuJoCo adds it to the dynamically generated ones, in their place if it is among them)
                if w1 == w2 or frozenset((gi[1], gj[1])) in excluded:
                    continue
                if w1 != 0 and w2 != 0 and (w1 == weld[body_parent[w2]] or w2 == weld[body_parent[w1]]):
                    continue
                if not _masks_match(gi[6], gi[7], gj[6], gj[7]):
                    continue
            mu_pair = pair_of_geoms.get(key)
            if GEOM_CYLINDER in (gi[0], gj[0]):
                raise ValueError("a cylinder geom can only collide with the ground plane (MJX pairs it through signed-distance functions: not built): "
                                 "exclude it from geom-geom pairs with contype / conaffinity")
            pair_rows.append(((gi[0], gj[0]), gi, gj, 0, mu_pair))
            if gj[0] in (GEOM_BOX, GEOM_MESH) and gi[0] == GEOM_CAPSULE:
                pair_rows.append(((gi[0], gj[0]), gi, gj, 1, mu_pair))  # capsule_convex fills two contact slots
            if gi[0] in (GEOM_BOX, GEOM_MESH):  # box / mesh against box / mesh (round 6; MJX convex_convex): a manifold of four contact slots
                for slot_ in (1, 2, 3):
                    pair_rows.append(((gi[0], gj[0]), gi, gj, slot_, mu_pair))
    pair_rows.sort(key=lambda r: r[0])  # stable: geom order inside a group, a pair's two slots next to each other
    pair_body, pair_geom = [], []
    # hull section: the convex geoms that take part in a pair, vertices / normals in the BODY frame
    hull_of: Dict[int, int] = {}
    hull_body: List[int] = []
    hull_vadr, hull_fadr, hull_eadr, hull_face_adr = [0], [0], [0], [0]
    hull_vert: List[List[float]] = []
    hull_fidx: List[int] = []
    hull_fnormal: List[List[float]] = []
    hull_edge: List[List[int]] = []
    hull_enormal: List[List[float]] = []
    hull_udadr: List[int] = [0]      # per hull: its edge DIRECTIONS with parallel ones dropped (unit vectors, body frame): the edge axes of convex_convex
    hull_udir: List[List[float]] = []

    def hull_id(gx) -> int:
        if id(gx) not in hull_of:
            if gx[0] == GEOM_BOX:
                local = np.asarray([[sx * gx[4][0], sy * gx[4][1], sz * gx[4][2]] for sx in (-1.0, 1.0) for sy in (-1.0, 1.0) for sz in (-1.0, 1.0)])
            else:
                local = gx[8]
            faces_, fn_, edges_, en_ = hull_topology(local)
            v0 = len(hull_vert)
            hull_of[id(gx)] = len(hull_body)
            hull_body.append(gx[1])
            hull_vert.extend([list(np.asarray(gx[2]) + _qrot(gx[3], v)) for v in local])
            for f_, n_ in zip(faces_, fn_):
                hull_fidx.extend([v0 + i for i in f_])
                hull_face_adr.append(len(hull_fidx))
                hull_fnormal.append(list(_qrot(gx[3], n_)))
            hull_edge.extend([[v0 + int(a), v0 + int(b)] for a, b in edges_])
            hull_enormal.extend([[*_qrot(gx[3], n2[0]), *_qrot(gx[3], n2[1])] for n2 in en_])
            kept: List[np.ndarray] = []
            for a, b in edges_:
                dvec = np.asarray(local[int(b)], np.float64) - np.asarray(local[int(a)], np.float64)
                dvec = dvec / np.linalg.norm(dvec)
                if all(np.sum(np.cross(dvec, k_) ** 2) >= 1e-6 for k_ in kept):
                    kept.append(dvec)
            hull_udir.extend([list(_qrot(gx[3], k_)) for k_ in kept])
            hull_udadr.append(len(hull_udir))
            hull_vadr.append(len(hull_vert)); hull_fadr.append(len(hull_fnormal)); hull_eadr.append(len(hull_edge))
        return hull_of[id(gx)]

    for _, gi, gj, slot, mu_pair in pair_rows:
        pair_body += [gi[1], gj[1]]
        hid = hull_id(gj) if gj[0] in (GEOM_BOX, GEOM_MESH) else -1
        hid1 = hull_id(gi) if gi[0] in (GEOM_BOX, GEOM_MESH) else -1
        for g_, r_, tag in ((gi, 0.0, float(hid + 1)), (gj, float(hid1 + 1), float(slot))):
            half = _qrot(g_[3], [0, 0, 1.0]) * g_[4][1] if g_[0] == GEOM_CAPSULE else np.zeros(3)
            # [7]: geom 2's hull + 1 (0: none), [14]: geom 1's hull + 1 (a hull pair; a round geom 2 carries its radius there), [15]: slot of a pair with several contacts
            pair_geom += [*g_[2], *half, g_[4][0] if g_[0] in (GEOM_SPHERE, GEOM_CAPSULE) else r_, tag]
        con_bodyid.append(gj[1])
        con_lpos.append([0.0, 0.0, 0.0])
        con_radius.append(0.0)
        fri_pair = np.maximum(np.asarray(gi[5]), np.asarray(gj[5]))
        if mu_pair is not None:
            fri_pair = np.asarray([mu_pair, fri_pair[1], fri_pair[2]])
        con_friction.append(list(fri_pair))
        con_axis.append([0.0, 0.0, 0.0])
        con_cvx.append(-1)
    npair = len(pair_rows)
    ncvx, ncvxvert = len(cvx_body), len(cvx_vert)
    ncyl = sum(1 for k in con_cvx if k == -2)
    ncon = len(con_bodyid)

    lim_jnt = [j for j in range(njnt) if jnt_limited[j] and jnt_type[j] != JNT_FREE]
    nlimit = len(lim_jnt)

    t: Dict[str, np.ndarray] = {}

    def put(k, v, dt=np.float64):
        t[k] = np.asarray(v, dtype=dt)

    for k, v in dict(nq=nq, nv=nv, nu=nu, nbody=nbody, njnt=njnt, ncon=ncon, nlimit=nlimit, npair=npair, ncvx=ncvx, ncvxvert=ncvxvert,
                     iterations=spec.iterations, ls_iterations=spec.ls_iterations).items():
        put(k, v, np.int32)
    for k, v in dict(timestep=spec.timestep, tolerance=spec.tolerance, ls_tolerance=spec.ls_tolerance,
                     impratio=spec.impratio, plane_z=spec.plane_z).items():
        put(k, v)
    put("gravity", spec.gravity)
    put("body_parent", body_parent, np.int32)
    put("body_rootid", body_rootid, np.int32)
    put("body_depth", depth, np.int32)
    put("nlevel", nlevel, np.int32)
    put("nroot", len(root_body), np.int32)
    put("level_adr", level_adr, np.int32)
    put("level_body", level_body, np.int32)
    put("root_body", root_body, np.int32)
    # bodies 0 .. 63 of every body's subtree, then (models with more than 64 bodies only) bodies 64 .. 127 of every body's subtree
    lo64 = (1 << 64) - 1
    put("body_subtree_mask", np.concatenate([_m64([m & lo64 for m in subtree])] + ([_m64([m >> 64 for m in subtree])] if nbody > 64 else [])), np.int32)
    put("body_ancdof_mask", _m64(ancdof), np.int32)
    put("dof_velmask", _m64(velmask), np.int32)
    put("dof_qposadr", dof_qposadr, np.int32)
    put("body_pos", body_pos)
    put("body_quat", body_quat)
    put("body_ipos", body_ipos)
    put("body_iquat", body_iquat)
    put("body_mass", body_mass)
    put("body_inertia", body_inertia)
    put("body_jntadr", body_jntadr, np.int32)
    put("body_jntnum", body_jntnum, np.int32)
    put("body_dofadr", body_dofadr, np.int32)
    put("body_dofnum", body_dofnum, np.int32)
    put("jnt_type", jnt_type, np.int32)
    put("jnt_qposadr", jnt_qposadr, np.int32)
    put("jnt_dofadr", jnt_dofadr, np.int32)
    put("jnt_bodyid", jnt_bodyid, np.int32)
    put("jnt_pos", np.reshape(jnt_pos, (njnt, 3)))
    put("jnt_axis", np.reshape(jnt_axis, (njnt, 3)))
    put("jnt_range", np.reshape(jnt_range, (njnt, 2)))
    put("jnt_limited", jnt_limited, np.int32)
    put("jnt_stiffness", jnt_stiffness)
    put("dof_bodyid", dof_bodyid, np.int32)
    put("dof_jntid", dof_jntid, np.int32)
    put("dof_parentid", dof_parentid, np.int32)
    put("dof_armature", dof_armature)
    put("dof_damping", dof_damping)
    put("qpos0", qpos0)
    put("qpos_spring", qpos_spring)
    put("dof_actfrcrange", np.reshape(dof_actfrcrange, (nv, 2)))
    put("act_dofid", act_dofid, np.int32)
    put("act_qposadr", act_qposadr, np.int32)
    put("act_gear", act_gear)
    put("act_gain", act_gain)
    put("act_bias", act_bias)
    put("act_ctrlrange", act_ctrlrange)
    put("act_ctrllimited", act_ctrllimited, np.int32)
    put("act_forcerange", act_forcerange)
    put("act_forcelimited", act_forcelimited, np.int32)
    put("con_bodyid", con_bodyid, np.int32)
    put("con_lpos", np.reshape(con_lpos, (ncon, 3)))
    put("con_radius", con_radius)
    put("con_friction", np.reshape(con_friction, (ncon, 3)))
    put("con_axis", np.reshape(con_axis, (ncon, 3)))
    put("con_cvx", con_cvx, np.int32)
    put("cvx_body", cvx_body, np.int32)
    put("cvx_vadr", cvx_vadr, np.int32)
    put("cvx_vert", np.reshape(cvx_vert, (ncvxvert, 3)))
    put("pair_body", np.reshape(pair_body, (npair, 2)), np.int32)
    put("pair_geom", np.reshape(pair_geom, (npair, 16)))
    put("ncyl", ncyl, np.int32)
    put("nhull", len(hull_body), np.int32)
    put("hull_body", hull_body, np.int32)
    put("hull_vadr", hull_vadr, np.int32)
    put("hull_fadr", hull_fadr, np.int32)
    put("hull_eadr", hull_eadr, np.int32)
    put("hull_face_adr", hull_face_adr, np.int32)
    put("hull_fidx", hull_fidx, np.int32)
    put("hull_edge", np.reshape(hull_edge, (len(hull_edge), 2)), np.int32)
    put("hull_vert", np.reshape(hull_vert, (len(hull_vert), 3)))
    put("hull_fnormal", np.reshape(hull_fnormal, (len(hull_fnormal), 3)))
    put("hull_enormal", np.reshape(hull_enormal, (len(hull_enormal), 6)))
    put("hull_udadr", hull_udadr, np.int32)
    put("hull_udir", np.reshape(hull_udir, (len(hull_udir), 3)))
    put("lim_jntid", lim_jnt, np.int32)
    put("contact_solref", spec.contact_solref)
    put("contact_solimp", spec.contact_solimp)
    put("limit_solref", spec.limit_solref)
    put("limit_solimp", spec.limit_solimp)

    cm = CompiledModel(spec.name, t, names, joint_names, spec.meaninertia)
    _set_const(cm)
    return cm


# ---------------------------------------------------------------------------
# mj_setConst equivalent: M(qpos0) -> invweights, meaninertia      (float64)
# ---------------------------------------------------------------------------


def _forward_position0(cm: CompiledModel):
    """Kinematics + com + CRB at qpos0 for one instance (host, float64)."""
    t = cm.t
    nb, nv = cm.nbody, cm.nv
    q = t["qpos0"]
    xpos = np.zeros((nb, 3))
    xquat = np.tile(np.array([1.0, 0, 0, 0]), (nb, 1))
    xanchor = np.zeros((cm.njnt, 3))
    xaxis = np.zeros((cm.njnt, 3))
    for b in range(1, nb):
        p = t["body_parent"][b]
        pos = xpos[p] + _qrot(xquat[p], t["body_pos"][b])
        quat = _qmul(xquat[p], t["body_quat"][b])
        for j in range(t["body_jntadr"][b], t["body_jntadr"][b] + t["body_jntnum"][b]):
            qa = t["jnt_qposadr"][j]
            if t["jnt_type"][j] == JNT_FREE:
                pos = q[qa:qa + 3].copy()
                quat = _normalize(q[qa + 3:qa + 7])
                xanchor[j] = pos
                xaxis[j] = _qrot(quat, t["jnt_axis"][j])
            else:
                xanchor[j] = pos + _qrot(quat, t["jnt_pos"][j])
                xaxis[j] = _qrot(quat, t["jnt_axis"][j])
                # qpos == qpos0 -> zero joint displacement
        xpos[b], xquat[b] = pos, quat
    xipos = np.array([xpos[b] + _qrot(xquat[b], t["body_ipos"][b]) for b in range(nb)])
    ximat = np.array([_qmat(_qmul(xquat[b], t["body_iquat"][b])) for b in range(nb)])
    # subtree com
    mpos = xipos * t["body_mass"][:, None]
    msum = t["body_mass"].copy()
    for b in range(nb - 1, 0, -1):
        p = t["body_parent"][b]
        mpos[p] += mpos[b]
        msum[p] += msum[b]
    subtree_com = mpos / np.maximum(msum, MJ_MINVAL)[:, None]
    cinert = np.zeros((nb, 10))
    for b in range(1, nb):
        off = xipos[b] - subtree_com[t["body_rootid"][b]]
        m = t["body_mass"][b]
        I = ximat[b] @ np.diag(t["body_inertia"][b]) @ ximat[b].T + m * (off @ off * np.eye(3) - np.outer(off, off))
        cinert[b] = [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2], *(m * off), m]
    cdof = np.zeros((nv, 6))
    for j in range(cm.njnt):
        b = t["jnt_bodyid"][j]
        da = t["jnt_dofadr"][j]
        off = subtree_com[t["body_rootid"][b]] - xanchor[j]
        if t["jnt_type"][j] == JNT_FREE:
            R = _qmat(xquat[b])
            for k in range(3):
                cdof[da + k, 3 + k] = 1.0
                ax = R[:, k]
                cdof[da + 3 + k, :3] = ax
                cdof[da + 3 + k, 3:] = np.cross(ax, off)
        elif t["jnt_type"][j] == JNT_HINGE:
            cdof[da, :3] = xaxis[j]
            cdof[da, 3:] = np.cross(xaxis[j], off)
        else:
            cdof[da, 3:] = xaxis[j]
    crb = cinert.copy()
    for b in range(nb - 1, 0, -1):
        crb[t["body_parent"][b]] += crb[b]
    M = np.zeros((nv, nv))
    for i in range(nv):
        buf = _inert_mul(crb[t["dof_bodyid"][i]], cdof[i])
        j = i
        while j >= 0:
            M[i, j] = M[j, i] = cdof[j] @ buf
            j = t["dof_parentid"][j]
        M[i, i] += t["dof_armature"][i]
    return dict(xpos=xpos, xquat=xquat, xipos=xipos, subtree_com=subtree_com, cdof=cdof, M=M)


def _inert_mul(i, v):
    return np.array(
        [
            i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5],
            i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5],
            i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4],
            i[8] * v[1] - i[7] * v[2] + i[9] * v[3],
            i[6] * v[2] - i[8] * v[0] + i[9] * v[4],
            i[7] * v[0] - i[6] * v[1] + i[9] * v[5],
        ]
    )


def _set_const(cm: CompiledModel) -> None:
    t = cm.t
    nv, nb = cm.nv, cm.nbody
    f = _forward_position0(cm)
    M = f["M"]
    Minv = np.linalg.inv(M)
    t["meaninertia"] = np.asarray(np.trace(M) / max(nv, 1)) if cm.meaninertia_override is None else np.asarray(float(cm.meaninertia_override))
    dof_inv = np.diag(Minv).copy()
    for j in range(cm.njnt):
        if t["jnt_type"][j] == JNT_FREE:
            da = t["jnt_dofadr"][j]
            dof_inv[da:da + 3] = dof_inv[da:da + 3].mean()
            dof_inv[da + 3:da + 6] = dof_inv[da + 3:da + 6].mean()
    t["dof_invweight0"] = dof_inv
    body_inv = np.zeros((nb, 2))
    for b in range(1, nb):
        jp = np.zeros((3, nv))
        jr = np.zeros((3, nv))
        off = f["xipos"][b] - f["subtree_com"][t["body_rootid"][b]]
        d = t["body_dofadr"][b] + t["body_dofnum"][b] - 1 if t["body_dofnum"][b] > 0 else -1
        if d < 0:  # walk up to the nearest ancestor with dofs
            a = t["body_parent"][b]
            while a > 0 and t["body_dofnum"][a] == 0:
                a = t["body_parent"][a]
            d = t["body_dofadr"][a] + t["body_dofnum"][a] - 1 if a > 0 else -1
        while d >= 0:
            jr[:, d] = f["cdof"][d, :3]
            jp[:, d] = f["cdof"][d, 3:] + np.cross(f["cdof"][d, :3], off)
            d = t["dof_parentid"][d]
        body_inv[b, 0] = np.trace(jp @ Minv @ jp.T) / 3.0
        body_inv[b, 1] = np.trace(jr @ Minv @ jr.T) / 3.0
    t["body_invweight0"] = body_inv
    t["M0"] = M


# ---------------------------------------------------------------------------
# device blob
# ---------------------------------------------------------------------------

# Order of the arrays in the blob.  (name, dtype) ; shapes follow from the header dims.
_BLOB_INT = [
    "body_parent", "body_rootid", "body_depth", "body_jntadr", "body_jntnum", "body_dofadr", "body_dofnum",
    "jnt_type", "jnt_qposadr", "jnt_dofadr", "jnt_bodyid", "jnt_limited",
    "dof_bodyid", "dof_jntid", "dof_parentid",
    "act_dofid", "act_qposadr", "act_ctrllimited", "act_forcelimited",
    "con_bodyid", "lim_jntid", "pair_body",
    "level_adr", "level_body", "root_body", "body_subtree_mask", "body_ancdof_mask", "dof_velmask", "dof_qposadr",
    "con_cvx", "cvx_body", "cvx_vadr",
]
_BLOB_F32 = [
    "gravity", "body_pos", "body_quat", "body_ipos", "body_iquat", "body_mass", "body_inertia",
    "jnt_pos", "jnt_axis", "jnt_range", "jnt_stiffness",
    "dof_armature", "dof_damping", "dof_invweight0", "body_invweight0",
    "qpos0", "qpos_spring",
    "act_gear", "act_gain", "act_bias", "act_ctrlrange", "act_forcerange",
    "con_lpos", "con_radius", "con_friction", "con_axis", "pair_geom",
    "contact_solref", "contact_solimp", "limit_solref", "limit_solimp",
    "cvx_vert", "dof_actfrcrange",
]
_HDR_INT = ["nq", "nv", "nu", "nbody", "njnt", "ncon", "nlimit", "iterations", "ls_iterations", "nlevel", "nroot", "include_c_vals", "npair"]
_HDR_F32 = ["timestep", "tolerance", "ls_tolerance", "impratio", "plane_z", "meaninertia"]
_HDR_INT2 = ["ncvx", "ncvxvert", "hull_words", "ncyl"]  # words 33..: dims that arrived after the first header block was full (hull_words: length of the hull section behind the table part)
BLOB_HEADER_WORDS = 64  # fixed-size header; array directory follows


def _to_blob(cm: CompiledModel, include_c_vals: bool = True) -> bytes:
    """Packs the model into one little-endian blob of 4-byte words.

    word 0: magic, 1: version, 2: total words of the TABLE part, 3..: header ints (see _HDR_INT),
    words 16..: header floats (see _HDR_F32), word 32: number of arrays, 33..: _HDR_INT2, the rest reserved.
    Then a directory of (offset_words, count) pairs for every array in
    _BLOB_INT + _BLOB_F32 order, then the arrays (each padded to 4 words) - the table part: what the environment kernel copies
    into LDS.  Behind it, if the model has convex geoms in geom-geom pairs, the HULL SECTION (header word `hull_words`; read from
    global memory, a few lines per step): eight words (nhull, nvert, nface, nfidx, nedge, 0, 0, 0), then the arrays of
    _HULL_ARRAYS in that order, each padded to 4 words.  The C side (csrc/model_view.h) mirrors this layout.
    """
    t = cm.t
    names = _BLOB_INT + _BLOB_F32
    ndir = len(names)
    words: List[bytes] = []
    dir_entries: List[Tuple[int, int]] = []
    cursor = BLOB_HEADER_WORDS + 2 * ndir
    cursor = (cursor + 3) // 4 * 4
    base = cursor
    payload = bytearray()
    for k in names:
        a = np.ascontiguousarray(t[k]).reshape(-1)
        if k in _BLOB_INT:
            raw = a.astype("<i4").tobytes()
        else:
            raw = a.astype("<f4").tobytes()
        n = a.size
        dir_entries.append((cursor, n))
        pad = (-n) % 4
        payload += raw + b"\0" * (4 * pad)
        cursor += n + pad
    total = cursor
    hdr = bytearray(4 * BLOB_HEADER_WORDS)
    struct.pack_into("<3I", hdr, 0, BLOB_MAGIC, BLOB_VERSION, total)
    for i, k in enumerate(_HDR_INT):
        struct.pack_into("<i", hdr, 4 * (3 + i), int(include_c_vals) if k == "include_c_vals" else int(t[k]))
    for i, k in enumerate(_HDR_F32):
        struct.pack_into("<f", hdr, 4 * (16 + i), float(t[k]))
    struct.pack_into("<i", hdr, 4 * 32, ndir)
    hull = _hull_section(t)
    for i, k in enumerate(_HDR_INT2):
        struct.pack_into("<i", hdr, 4 * (33 + i), len(hull) // 4 if k == "hull_words" else int(t[k]))
    d = bytearray()
    for off, n in dir_entries:
        d += struct.pack("<2i", off, n)
    d += b"\0" * (4 * (base - BLOB_HEADER_WORDS - 2 * ndir))
    blob = bytes(hdr) + bytes(d) + bytes(payload)
    assert len(blob) == 4 * total, (len(blob), total)
    return blob + hull


_HULL_ARRAYS = [("hull_vadr", "<i4"), ("hull_fadr", "<i4"), ("hull_eadr", "<i4"), ("hull_face_adr", "<i4"), ("hull_fidx", "<i4"), ("hull_edge", "<i4"),
                ("hull_vert", "<f4"), ("hull_fnormal", "<f4"), ("hull_enormal", "<f4"), ("hull_udadr", "<i4"), ("hull_udir", "<f4")]


def _hull_section(t: Dict[str, np.ndarray]) -> bytes:
    nh = int(t["nhull"]) if "nhull" in t else 0
    if nh == 0:
        return b""
    out = bytearray(struct.pack("<8i", nh, len(t["hull_vert"]), len(t["hull_fnormal"]), len(t["hull_fidx"]), len(t["hull_edge"]), len(t["hull_udir"]), 0, 0))
    for k, dt in _HULL_ARRAYS:
        raw = np.ascontiguousarray(t[k]).reshape(-1).astype(dt).tobytes()
        out += raw + b"\0" * ((-len(raw)) % 16)
    return bytes(out)


# ---------------------------------------------------------------------------
# built-in stand-in robots
# ---------------------------------------------------------------------------


def _leg(side: str, y: float) -> List[BodySpec]:
    s = side
    return [
        BodySpec(f"{s}_hip_yaw", "torso", pos=(0.0, y, -0.10), mass=0.8, inertia=(0.002, 0.002, 0.002),
                 joints=[JointSpec(f"{s}_hip_yaw", JNT_HINGE, axis=(0, 0, 1), range=(-0.8, 0.8), damping=2.0, armature=0.1)]),
        BodySpec(f"{s}_hip_roll", f"{s}_hip_yaw", pos=(0.0, 0.0, -0.05), mass=0.8, inertia=(0.002, 0.002, 0.002),
                 joints=[JointSpec(f"{s}_hip_roll", JNT_HINGE, axis=(1, 0, 0), range=(-0.6, 0.6), damping=2.0, armature=0.1)]),
        BodySpec(f"{s}_thigh", f"{s}_hip_roll", pos=(0.0, 0.0, -0.05), mass=3.0, inertia=(0.045, 0.045, 0.006),
                 ipos=(0.0, 0.0, -0.18),
                 joints=[JointSpec(f"{s}_hip_pitch", JNT_HINGE, axis=(0, 1, 0), range=(-1.6, 1.0), damping=2.0, armature=0.1)]),
        BodySpec(f"{s}_shin", f"{s}_thigh", pos=(0.0, 0.0, -0.38), mass=2.0, inertia=(0.028, 0.028, 0.003),
                 ipos=(0.0, 0.0, -0.17),
                 joints=[JointSpec(f"{s}_knee", JNT_HINGE, axis=(0, 1, 0), range=(-0.1, 2.2), damping=2.0, armature=0.1)],
                 geoms=[GeomSpec(GEOM_SPHERE, (0.05,), pos=(0.0, 0.0, 0.0))]),
        BodySpec(f"{s}_foot", f"{s}_shin", pos=(0.0, 0.0, -0.36), mass=1.2, inertia=(0.004, 0.008, 0.008),
                 ipos=(0.03, 0.0, -0.03),
                 joints=[JointSpec(f"{s}_ankle", JNT_HINGE, axis=(0, 1, 0), range=(-0.9, 0.9), damping=2.0, armature=0.1)],
                 geoms=[GeomSpec(GEOM_CAPSULE, (0.03, 0.08), pos=(0.03, 0.0, -0.04), quat=(0.70710678, 0.0, 0.70710678, 0.0))]),
    ]


def _arm(side: str, y: float) -> List[BodySpec]:
    s = side
    return [
        BodySpec(f"{s}_shoulder_pitch", "torso", pos=(0.0, y, 0.32), mass=0.6, inertia=(0.001, 0.001, 0.001),
                 joints=[JointSpec(f"{s}_shoulder_pitch", JNT_HINGE, axis=(0, 1, 0), range=(-2.0, 2.0), damping=1.0, armature=0.05)]),
        BodySpec(f"{s}_shoulder_roll", f"{s}_shoulder_pitch", pos=(0.0, 0.0, -0.03), mass=0.6, inertia=(0.001, 0.001, 0.001),
                 joints=[JointSpec(f"{s}_shoulder_roll", JNT_HINGE, axis=(1, 0, 0), range=(-1.5, 1.5), damping=1.0, armature=0.05)]),
        BodySpec(f"{s}_upper_arm", f"{s}_shoulder_roll", pos=(0.0, 0.0, -0.03), mass=1.2, inertia=(0.008, 0.008, 0.001),
                 ipos=(0.0, 0.0, -0.12),
                 joints=[JointSpec(f"{s}_shoulder_yaw", JNT_HINGE, axis=(0, 0, 1), range=(-1.5, 1.5), damping=1.0, armature=0.05)]),
        BodySpec(f"{s}_forearm", f"{s}_upper_arm", pos=(0.0, 0.0, -0.26), mass=0.9, inertia=(0.005, 0.005, 0.0008),
                 ipos=(0.0, 0.0, -0.11),
                 joints=[JointSpec(f"{s}_elbow", JNT_HINGE, axis=(0, 1, 0), range=(-2.2, 0.1), damping=1.0, armature=0.05)]),
        BodySpec(f"{s}_hand", f"{s}_forearm", pos=(0.0, 0.0, -0.24), mass=0.4, inertia=(0.0005, 0.0005, 0.0003),
                 ipos=(0.0, 0.0, -0.04),
                 joints=[JointSpec(f"{s}_wrist", JNT_HINGE, axis=(0, 0, 1), range=(-1.5, 1.5), damping=0.5, armature=0.02)],
                 geoms=[GeomSpec(GEOM_SPHERE, (0.04,), pos=(0.0, 0.0, -0.05))]),
    ]


def _humanoid(name: str, arms: bool, self_collide: bool = False) -> ModelSpec:
    torso = BodySpec(
        "torso", "world", pos=(0.0, 0.0, 0.0), mass=10.0 if not arms else 9.0, inertia=(0.12, 0.10, 0.06),
        ipos=(0.0, 0.0, 0.12),
        joints=[JointSpec("root", JNT_FREE)],
        geoms=[GeomSpec(GEOM_SPHERE, (0.11,), pos=(0.0, 0.0, 0.05))],
    )
    bodies = [torso] + _leg("left", 0.09) + _leg("right", -0.09)
    joints = [j.name for b in bodies[1:] for j in b.joints]
    if arms:
        arm_bodies = _arm("left", 0.19) + _arm("right", -0.19)
        bodies += arm_bodies
        joints += [j.name for b in arm_bodies for j in b.joints]
    acts = []
    for jn in joints:
        leg = any(k in jn for k in ("hip", "knee", "ankle"))
        acts.append(ActuatorSpec(jn, kp=40.0 if leg else 15.0, kv=0.0, ctrlrange=(-1.5, 1.5),
                                 forcerange=(-60.0, 60.0) if leg else (-20.0, 20.0)))
    # standing height: hip chain 0.10+0.05+0.05, thigh 0.38, shin 0.36, foot capsule centre 0.04 below ankle, r=0.03
    root_z = 0.10 + 0.05 + 0.05 + 0.38 + 0.36 + 0.04 + 0.03 - 0.0005
    if self_collide:  # MuJoCo's default masks: every geom pair that is not parent-child is a candidate
        for b in bodies:
            for g_ in b.geoms:
                g_.contype = 1
    return ModelSpec(name=name, bodies=bodies, actuators=acts, free_root_z=root_z)


def synth_stompy_pro() -> ModelSpec:
    """Stand-in for the stompy_pro MJCF (unobtainable on the target): free root + 2 legs x 5 hinges."""
    return _humanoid("synth_stompy_pro", arms=False)


def synth_stompy_full() -> ModelSpec:
    """Stand-in with a larger action dimension (BASELINE config 5): + 2 arms x 5 hinges."""
    return _humanoid("synth_stompy_full", arms=True)


def synth_stompy_pro_sc() -> ModelSpec:
    """synth_stompy_pro with MuJoCo's default collision masks: + 8 geom-geom candidates (torso / shins / feet across the legs)."""
    return _humanoid("synth_stompy_pro_sc", arms=False, self_collide=True)


def synth_stompy_frames() -> ModelSpec:
    """synth_stompy_full the way an export leaves a robot: every link carries three jointless frames (sensor / fastener / cover bodies with a
    little mass, which MuJoCo keeps as bodies of their own), and each hand two two-link fingers behind them - 93 bodies with the world, 34
    dofs: beyond the 64 bodies one mask word covers (the run-time-sized kernel's two-word subtree sets; bodies and dofs past index 64)."""
    spec = _humanoid("synth_stompy_frames", arms=True)
    links = list(spec.bodies)
    frames = []
    for bi, b in enumerate(links):
        for k in range(3):
            off = (0.02 * (k - 1), 0.015 * ((bi + k) % 3 - 1), -0.03 * k)
            frames.append(BodySpec(f"{b.name}_frame{k}", b.name, pos=off, quat=_normalize((1.0, 0.1 * k, 0.05 * bi % 0.3, 0.0)), mass=0.02 + 0.01 * k,
                                   inertia=(2e-5, 3e-5, 4e-5), ipos=(0.0, 0.005, 0.0)))
    fingers = []
    for side in ("left", "right"):
        for f in range(2):
            y = 0.015 * (2 * f - 1)
            fingers.append(BodySpec(f"{side}_finger{f}_a", f"{side}_hand", pos=(0.0, y, -0.09), mass=0.05, inertia=(2e-5, 2e-5, 1e-5), ipos=(0.0, 0.0, -0.02),
                                    joints=[JointSpec(f"{side}_finger{f}_a", JNT_HINGE, axis=(0, 1, 0), range=(-0.2, 1.2), damping=0.05, armature=0.002)]))
            fingers.append(BodySpec(f"{side}_finger{f}_b", f"{side}_finger{f}_a", pos=(0.0, 0.0, -0.04), mass=0.03, inertia=(1e-5, 1e-5, 5e-6), ipos=(0.0, 0.0, -0.015),
                                    joints=[JointSpec(f"{side}_finger{f}_b", JNT_HINGE, axis=(0, 1, 0), range=(-0.2, 1.2), damping=0.05, armature=0.002)],
                                    geoms=[GeomSpec(GEOM_SPHERE, (0.012,), pos=(0.0, 0.0, -0.03))]))
    # in document order (depth first), as MuJoCo numbers the bodies of an MJCF: a link, its frames, then the next link of the chain
    everything = links + frames + fingers
    ordered: List[BodySpec] = []

    def visit(parent: str) -> None:
        for b in everything:
            if b.parent == parent:
                ordered.append(b)
                visit(b.name)

    visit("world")
    spec.bodies = ordered
    spec.actuators = spec.actuators + [ActuatorSpec(b.joints[0].name, kp=2.0, ctrlrange=(-1.0, 1.0), forcerange=(-2.0, 2.0)) for b in fingers]
    return spec


def synth_tumblers() -> ModelSpec:
    """Three free bodies (sphere, two capsules) that collide with each other and with the ground: all three pair routines,
    several kinematic trees."""
    cap = (0.70710678, 0.0, 0.70710678, 0.0)  # capsule axis along x
    bodies = [
        BodySpec("ball", "world", pos=(0.0, 0.0, 0.0), mass=1.0, inertia=(0.004, 0.004, 0.004),
                 joints=[JointSpec("ball_root", JNT_FREE)], geoms=[GeomSpec(GEOM_SPHERE, (0.1,), contype=1)]),
        BodySpec("rod_a", "world", pos=(0.25, 0.0, 0.4), mass=0.8, inertia=(0.002, 0.012, 0.012),
                 joints=[JointSpec("rod_a_root", JNT_FREE)], geoms=[GeomSpec(GEOM_CAPSULE, (0.05, 0.2), quat=cap, contype=1)]),
        BodySpec("rod_b", "world", pos=(0.0, 0.3, 0.45), mass=0.6, inertia=(0.010, 0.010, 0.001),
                 joints=[JointSpec("rod_b_root", JNT_FREE)], geoms=[GeomSpec(GEOM_CAPSULE, (0.04, 0.15), contype=1)]),
    ]
    return ModelSpec(name="synth_tumblers", bodies=bodies, actuators=[], free_root_z=0.4)


def synth_pendulum() -> ModelSpec:
    """Two-link pendulum on a free-floating base-less world hinge; used by analytic physics tests."""
    bodies = [
        BodySpec("link1", "world", pos=(0.0, 0.0, 2.0), mass=1.0, inertia=(0.01, 0.01, 0.001), ipos=(0.0, 0.0, -0.25),
                 joints=[JointSpec("j1", JNT_HINGE, axis=(0, 1, 0))]),
        BodySpec("link2", "link1", pos=(0.0, 0.0, -0.5), mass=0.5, inertia=(0.004, 0.004, 0.0005), ipos=(0.0, 0.0, -0.2),
                 joints=[JointSpec("j2", JNT_HINGE, axis=(0, 1, 0))]),
    ]
    return ModelSpec(name="synth_pendulum", bodies=bodies, actuators=[ActuatorSpec("j1"), ActuatorSpec("j2")])


def synth_ball() -> ModelSpec:
    """A single free sphere above the plane: free-fall and resting-contact analytic tests."""
    bodies = [
        BodySpec("ball", "world", mass=1.0, inertia=(0.004, 0.004, 0.004),
                 joints=[JointSpec("root", JNT_FREE)], geoms=[GeomSpec(GEOM_SPHERE, (0.1,))]),
    ]
    return ModelSpec(name="synth_ball", bodies=bodies, actuators=[], free_root_z=0.5)


def synth_brick() -> ModelSpec:
    """A free box above the plane, tilted: the box collider (plane_convex on its eight corners; landing on one corner, then an edge, then a face)."""
    bodies = [
        BodySpec("brick", "world", quat=(0.9659258, 0.1830127, 0.1830127, 0.0), mass=1.2, inertia=(0.0013, 0.0044, 0.0055),
                 joints=[JointSpec("root", JNT_FREE)], geoms=[GeomSpec(GEOM_BOX, (0.10, 0.05, 0.02))]),
    ]
    return ModelSpec(name="synth_brick", bodies=bodies, actuators=[], free_root_z=0.12)


def synth_can() -> ModelSpec:
    """A free cylinder above the plane, tilted, with a small hinged cylinder (a lid that swings about the can's x axis) on top: the cylinder
    collider of SURVEY 8(f1) - MJX collision_primitive.plane_cylinder, three slots per geom (what a URDF-derived export uses for wheels and
    feet; URDF has no capsule)."""
    bodies = [
        BodySpec("can", "world", quat=(0.9537, 0.2132, 0.2132, 0.0), mass=0.8, inertia=(0.0029, 0.0029, 0.0014), joints=[JointSpec("root", JNT_FREE)],
                 geoms=[GeomSpec(GEOM_CYLINDER, (0.06, 0.09), pos=(0.0, 0.005, 0.01), friction=(0.9, 0.005, 0.0001))]),
        BodySpec("lid", "can", pos=(0.0, 0.0, 0.13), mass=0.1, inertia=(0.00004, 0.00004, 0.00006),
                 joints=[JointSpec("lid", JNT_HINGE, axis=(1, 0, 0), range=(-0.8, 0.8), damping=0.02, armature=0.001)],
                 geoms=[GeomSpec(GEOM_CYLINDER, (0.035, 0.012), quat=(0.9961947, 0.0, 0.0871557, 0.0))]),
    ]
    return ModelSpec(name="synth_can", bodies=bodies, actuators=[ActuatorSpec("lid", kp=1.0, ctrlrange=(-1.0, 1.0))], free_root_z=0.14)


def synth_pile() -> ModelSpec:
    """Every contact routine of SURVEY 8(f1) in one scene, at rest under gravity: a box and a 20-vertex mesh rock on the ground (corner
    contacts, plane_convex), a ball on the box (sphere_convex), a rod across the rock (capsule_convex), a can standing beside them
    (plane_cylinder) - each starting 2 - 3 mm inside what it rests on, so that the contacts act from the first step (what an external MJX
    fixture of ten steps records: tools/make_mjx_fixtures.py).  Masks: the plane types 1 | 4 and is affine to nothing; box and rock are
    affine to 1, ball and rod type 1 and are affine to 4, the can is affine to 4 only - no box-rock and no cylinder pair arises."""
    rng = np.random.default_rng(5)
    v = rng.normal(size=(20, 3))
    v = v / np.linalg.norm(v, axis=1, keepdims=True) * [0.3, 0.25, 0.2]
    top = v[np.argmax(v[:, 2])]
    rock_z = -float(v[:, 2].min()) - 0.003
    free = lambda n: [JointSpec(n, JNT_FREE)]
    bodies = [
        BodySpec("box", "world", pos=(0.0, 0.0, 0.0), mass=2.0, inertia=(0.033, 0.067, 0.087), joints=free("box"),
                 geoms=[GeomSpec(GEOM_BOX, (0.3, 0.2, 0.1), contype=0, conaffinity=1, friction=(0.9, 0.005, 0.0001))]),
        BodySpec("rock", "world", pos=(1.0, 0.0, rock_z), mass=1.5, inertia=(0.02, 0.025, 0.03), joints=free("rock"),
                 geoms=[GeomSpec(GEOM_MESH, (), vertices=tuple(map(tuple, v)), contype=0, conaffinity=1)]),
        BodySpec("ball", "world", pos=(0.05, 0.03, 0.2 + 0.08 - 0.003), mass=0.5, inertia=(0.0013, 0.0013, 0.0013), joints=free("ball"),
                 geoms=[GeomSpec(GEOM_SPHERE, (0.08,), contype=1, conaffinity=4)]),
        BodySpec("rod", "world", pos=(1.0 + float(top[0]), float(top[1]), rock_z + float(top[2]) + 0.05 - 0.002), quat=(0.70710678, 0.0, 0.70710678, 0.0),
                 mass=0.4, inertia=(0.0032, 0.0032, 0.0005), joints=free("rod"), geoms=[GeomSpec(GEOM_CAPSULE, (0.05, 0.15), contype=1, conaffinity=4)]),
        BodySpec("can", "world", pos=(0.0, 1.0, 0.09 - 0.002), mass=0.8, inertia=(0.0029, 0.0029, 0.0014), joints=free("can"),
                 geoms=[GeomSpec(GEOM_CYLINDER, (0.06, 0.09), contype=0, conaffinity=4)]),
    ]
    return ModelSpec(name="synth_pile", bodies=bodies, actuators=[], free_root_z=0.1 - 0.002, plane_contype=5, plane_conaffinity=0)


# hull vertices of an irregular "foot" (no two edges parallel, no symmetric pairs: the argmax steps of plane_convex have no exact ties)
WEDGE_VERTS = ((0.11, 0.045, -0.021), (0.12, -0.04, -0.019), (-0.09, -0.052, -0.02), (-0.10, 0.038, -0.018), (0.07, 0.03, 0.028), (0.06, -0.025, 0.03),
               (-0.05, -0.03, 0.035), (-0.06, 0.02, 0.04), (0.0, 0.0, 0.055))


def synth_wedge() -> ModelSpec:
    """A free body whose collider is a convex MESH (nine hull vertices) with a second, hinged mesh body hanging off it: the
    plane-convex contact of SURVEY 8(f1) (MJX collision_convex.plane_convex: four slots per mesh geom)."""
    bodies = [
        BodySpec("wedge", "world", quat=(0.9848078, 0.1227878, 0.1227878, 0.0), mass=1.5, inertia=(0.0021, 0.0052, 0.0061),
                 joints=[JointSpec("root", JNT_FREE)], geoms=[GeomSpec(GEOM_MESH, (), vertices=WEDGE_VERTS, friction=(0.9, 0.005, 0.0001))]),
        BodySpec("flap", "wedge", pos=(0.12, 0.0, 0.03), mass=0.3, inertia=(0.0004, 0.0006, 0.0007),
                 joints=[JointSpec("flap_hinge", JNT_HINGE, axis=(0.0, 1.0, 0.0), range=(-0.8, 0.8), damping=0.05, armature=0.002)],
                 geoms=[GeomSpec(GEOM_MESH, (), pos=(0.06, 0.0, 0.0), quat=(0.9914449, 0.0, 0.0, 0.1305262),
                                 vertices=tuple((0.6 * x, 0.7 * y, 0.8 * z) for x, y, z in WEDGE_VERTS))]),
    ]
    return ModelSpec(name="synth_wedge", bodies=bodies, actuators=[ActuatorSpec("flap_hinge", kp=6.0, kv=0.3, ctrlrange=(-0.8, 0.8))], free_root_z=0.14)


KSCALE_ID_TABLE = {
    "5eb3cb7f23232298": "synth_stompy_pro",  # reference configs/stompy_pro.yaml:1
}

BUILTIN_MODELS = {
    "synth_stompy_pro": synth_stompy_pro,
    "synth_stompy_full": synth_stompy_full,
    "synth_stompy_pro_sc": synth_stompy_pro_sc,
    "synth_stompy_frames": synth_stompy_frames,
    "synth_tumblers": synth_tumblers,
    "synth_pendulum": synth_pendulum,
    "synth_ball": synth_ball,
    "synth_brick": synth_brick,
    "synth_can": synth_can,
    "synth_pile": synth_pile,
    "synth_wedge": synth_wedge,
}

_CACHE: Dict[str, CompiledModel] = {}


def load_model(name_or_id: str) -> CompiledModel:
    """Resolves a `kscale_id`, a built-in model name or an MJCF path to a CompiledModel.

    Stands where the reference's `load_mjcf_model` + `mjcf.load_model` stand
    (`minppo/env.py:27-50,94-100`)."""
    key = KSCALE_ID_TABLE.get(name_or_id, name_or_id)
    if key in _CACHE:
        return _CACHE[key]
    if key in BUILTIN_MODELS:
        cm = compile_model(BUILTIN_MODELS[key]())
    elif key.endswith(".xml"):
        from minppo_amd.mjcf import load_mjcf  # MJCF subset compiler

        cm = compile_model(load_mjcf(key))
    else:
        raise ValueError(
            f"Unknown robot '{name_or_id}': not a built-in model ({sorted(BUILTIN_MODELS)}), "
            f"not a known kscale_id ({sorted(KSCALE_ID_TABLE)}) and not an .xml path"
        )
    _CACHE[key] = cm
    return cm
