"""MJCF subset -> ModelSpec (and back): the on-disk robot format of the engine.

Stands where the reference's `load_mjcf_model` stands (`minppo/env.py:27-50`): it fetches an MJCF through the K-Scale API,
strips `frictionloss` from `<default><joint>` "as Brax does not support it" (`env.py:31-45`), and hands the file to MuJoCo;
`HumanoidEnv.__init__` then forces the CG solver with 6 iterations / 6 line-search iterations (`env.py:95-97`).  Neither the
API, the file nor MuJoCo exist on the target, so a user who HAS the robot's MJCF passes its path as `kscale_id` and this
module turns the subset below into the engine's `ModelSpec` (`minppo_amd/model.py`), which `compile_model` lowers to the
blob the physics kernel consumes.  PARITY UNPINNED: MuJoCo's XML semantics are restated from its documentation; no MuJoCo
is available here to compare compiled models with.  Whatever is outside the subset raises `ValueError` naming the element
or attribute - nothing is silently dropped except purely visual / bookkeeping content (`<asset>`, `<visual>`, `<sensor>`,
`<keyframe>`, `<statistic>`, `<size>`, `<custom>`, materials, rgba, names of geoms ...).

Subset
  <compiler angle="degree|radian" eulerseq autolimits inertiafromgeom="auto|true|false" meshdir strippath boundmass boundinertia settotalmass
            balanceinertia> (fusestatic="true" is refused when the model has a jointless body: fusing would renumber the bodies)
  <option timestep gravity impratio>            solver / iterations / ls_iterations are OVERRIDDEN (CG, 6, 6: env.py:95-97)
  <default> with nested <default class="...">: <joint>, <geom>, <position>, <motor>, <general> attribute inheritance;
            `childclass` on <body>, `class` on elements
  <worldbody>: one <geom type="plane"> (the ground), <body name pos quat|euler|axisangle|xyaxes|zaxis childclass>
      <inertial pos quat|euler mass diaginertia|fullinertia>
      <freejoint/> | <joint type="free|hinge|slide" name pos axis range limited ref springref damping armature stiffness actuatorfrcrange>
      <geom type="sphere|capsule|cylinder|box|mesh" size pos quat|euler fromto friction mass density contype conaffinity mesh>
            (a box - the convex mesh of its eight corners - and a mesh collide with the ground as a CONVEX HULL, as in
             MuJoCo / MJX - up to four contacts per step, MJX's plane_convex; a cylinder meets the ground with three contacts per step,
             MJX's plane_cylinder, and nothing else; type="ellipsoid" geoms are accepted ONLY with contype="0" conaffinity="0", i.e.
             visual or inertia-only, and then still contribute to inertiafromgeom; a mesh never contributes an inertia: its body needs
             an <inertial>)
  <asset><mesh name vertex="x y z ..." | file="*.obj|*.stl" scale maxhullvert class>: the collision geometry of mesh geoms (files relative
            to the MJCF, honouring <compiler meshdir>; <default><mesh scale maxhullvert> applies - where exports put the millimetre scale;
            maxhullvert caps the hull as MuJoCo does, so a mesh of thousands of vertices loads with maxhullvert="64" instead of being
            decimated by hand); everything else under <asset> is visual and ignored
  <actuator>: <position joint kp kv gear ctrlrange forcerange>, <motor joint gear ctrlrange forcerange>, <velocity joint kv ...>,
            <general joint gainprm biastype="none|affine" biasprm ...> (dyntype none, gaintype fixed)
  <contact><exclude body1 body2/>: no contacts between the geoms of these two bodies; <pair geom1 geom2 friction solref solimp/>: an explicit
            geom pair (or a geom with the ground plane) with a sliding friction of its own, whatever masks / kinship / excludes say
Contacts: geom-vs-ground-plane, and the geom pairs between bodies that MuJoCo would test (contype / conaffinity masks, same-body and
parent-child pairs filtered, <exclude>d body pairs dropped): sphere / capsule among themselves, and a sphere or capsule against a box
or a mesh hull of another body (MJX sphere_convex / capsule_convex); a box or mesh that the masks pair with another box or mesh is an
error (`compile_model`).
"""

from __future__ import annotations

import logging
import math
import xml.etree.ElementTree as ET
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from minppo_amd.model import (GEOM_BOX, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_MESH, GEOM_SPHERE, JNT_FREE, JNT_HINGE, JNT_SLIDE, MAX_CONVEX_VERTS, ActuatorSpec, BodySpec, GeomSpec, JointSpec,
                              ModelSpec, _normalize, _qmat, _qmul)

logger = logging.getLogger(__name__)

_IGNORED_TOP = {"asset", "visual", "sensor", "keyframe", "statistic", "size", "custom", "extension"}
_UNSUPPORTED_TOP = {"equality", "tendon", "deformable", "flexcomp", "composite"}
_DEFAULT_DENSITY = 1000.0
_MJ_SOLREF = (0.02, 1.0)
_MJ_SOLIMP = (0.9, 0.95, 0.001, 0.5, 2.0)


# ---------------------------------------------------------------------------
# small helpers
# ---------------------------------------------------------------------------


def _floats(text: str, n: Optional[int] = None, what: str = "") -> List[float]:
    vals = [float(x) for x in text.replace(",", " ").split()]
    if n is not None and len(vals) != n:
        raise ValueError(f"{what}: expected {n} numbers, got {len(vals)} ({text!r})")
    return vals


def _mat_to_quat(R: np.ndarray) -> np.ndarray:
    """Rotation matrix -> unit quaternion (w,x,y,z), w >= 0."""
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = [0.0] * 4
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    q = _normalize(q)
    return q if q[0] >= 0 else -q


def _axis_angle_quat(axis, angle: float) -> np.ndarray:
    a = _normalize(axis)
    return np.array([math.cos(angle / 2), *(math.sin(angle / 2) * a)])


def _z_to(direction) -> np.ndarray:
    """Shortest rotation taking the local z axis onto `direction` (MuJoCo's fromto / zaxis convention)."""
    d = _normalize(direction)
    z = np.array([0.0, 0.0, 1.0])
    c = float(np.dot(z, d))
    if c > 1 - 1e-12:
        return np.array([1.0, 0.0, 0.0, 0.0])
    if c < -1 + 1e-12:
        return np.array([0.0, 1.0, 0.0, 0.0])
    return _axis_angle_quat(np.cross(z, d), math.acos(c))


class _Compiler:
    def __init__(self, el: Optional[ET.Element]):
        a = el.attrib if el is not None else {}
        angle = a.get("angle", "degree")
        if angle not in ("degree", "radian"):
            raise ValueError(f"<compiler angle={angle!r}>: degree or radian")
        self.deg = angle == "degree"
        self.eulerseq = a.get("eulerseq", "xyz")
        if len(self.eulerseq) != 3 or any(c not in "xyzXYZ" for c in self.eulerseq):
            raise ValueError(f"<compiler eulerseq={self.eulerseq!r}>")
        self.autolimits = a.get("autolimits", "true") == "true"
        self.inertiafromgeom = a.get("inertiafromgeom", "auto")
        for k in a:
            if k not in ("angle", "eulerseq", "autolimits", "inertiafromgeom", "meshdir", "texturedir", "assetdir", "discardvisual", "strippath", "balanceinertia",
                         "boundmass", "boundinertia", "settotalmass", "coordinate", "fusestatic"):
                raise ValueError(f"<compiler {k}=...> is outside the supported MJCF subset")
        if a.get("coordinate", "local") != "local":
            raise ValueError("<compiler coordinate='global'> is not supported")
        # mass / inertia post-processing of MuJoCo's compiler (until round 5 these five were accepted and ignored)
        self.boundmass, self.boundinertia = float(a.get("boundmass", "0")), float(a.get("boundinertia", "0"))
        self.settotalmass = float(a.get("settotalmass", "-1"))
        self.balanceinertia = a.get("balanceinertia", "false") == "true"
        self.strippath = a.get("strippath", "false") == "true"
        self.fusestatic = a.get("fusestatic", "false") == "true"

    def ang(self, x: float) -> float:
        return math.radians(x) if self.deg else x

    def orientation(self, a: Dict[str, str], what: str) -> np.ndarray:
        given = [k for k in ("quat", "euler", "axisangle", "xyaxes", "zaxis") if k in a]
        if len(given) > 1:
            raise ValueError(f"{what}: more than one orientation attribute {given}")
        if not given:
            return np.array([1.0, 0.0, 0.0, 0.0])
        k = given[0]
        if k == "quat":
            return _normalize(_floats(a[k], 4, what))
        if k == "euler":
            e = _floats(a[k], 3, what)
            q = np.array([1.0, 0.0, 0.0, 0.0])
            for c, v in zip(self.eulerseq, e):
                ax = {"x": (1, 0, 0), "y": (0, 1, 0), "z": (0, 0, 1)}[c.lower()]
                r = _axis_angle_quat(ax, self.ang(v))
                q = _qmul(q, r) if c.islower() else _qmul(r, q)  # lower case: rotating frame; upper case: fixed frame
            return _normalize(q)
        if k == "axisangle":
            v = _floats(a[k], 4, what)
            return _axis_angle_quat(v[:3], self.ang(v[3]))
        if k == "zaxis":
            return _z_to(_floats(a[k], 3, what))
        v = _floats(a[k], 6, what)  # xyaxes
        x = _normalize(v[:3])
        y = np.asarray(v[3:]) - np.dot(v[3:], x) * x
        y = _normalize(y)
        return _mat_to_quat(np.stack([x, y, np.cross(x, y)], axis=1))


class _Defaults:
    """`<default>` tree: class name -> {element tag -> attributes}, children inherit from their parents."""

    TAGS = ("joint", "geom", "position", "motor", "general", "velocity", "mesh")

    def __init__(self, root: ET.Element):
        self.classes: Dict[str, Dict[str, Dict[str, str]]] = {"main": {t: {} for t in self.TAGS}}
        for top in root.findall("default"):
            self._walk(top, "main", is_top=True)

    def _walk(self, el: ET.Element, parent: str, is_top: bool = False) -> None:
        name = el.get("class", "main" if is_top else None)
        if name is None:
            raise ValueError("nested <default> needs a class name")
        base = self.classes[parent]
        cur = self.classes.setdefault(name, {t: dict(base[t]) for t in self.TAGS}) if name != "main" else self.classes["main"]
        for ch in el:
            if ch.tag == "default":
                continue
            if ch.tag in self.TAGS:
                attrs = dict(ch.attrib)
                if ch.tag == "joint" and "frictionloss" in attrs:
                    # the reference deletes exactly this attribute before MuJoCo sees the file (env.py:41-45)
                    del attrs["frictionloss"]
                cur[ch.tag].update(attrs)
            elif ch.tag in ("material", "site", "camera", "light", "pair", "equality", "tendon"):
                if ch.tag in ("equality", "tendon", "pair"):
                    raise ValueError(f"<default><{ch.tag}> is outside the supported MJCF subset")
            else:
                raise ValueError(f"<default><{ch.tag}> is outside the supported MJCF subset")
        for ch in el.findall("default"):
            self._walk(ch, name)

    def resolve(self, tag: str, el: ET.Element, childclass: Optional[str]) -> Dict[str, str]:
        cls = el.get("class", childclass or "main")
        if cls not in self.classes:
            raise ValueError(f"<{tag} class={cls!r}>: unknown default class")
        out = dict(self.classes[cls][tag])
        out.update({k: v for k, v in el.attrib.items() if k != "class"})
        return out


# ---------------------------------------------------------------------------
# inertia of primitive geoms (inertiafromgeom)
# ---------------------------------------------------------------------------


def _geom_mass_inertia(gtype: str, size: Sequence[float], density: float, mass: Optional[float]) -> Tuple[float, np.ndarray]:
    """Mass and principal inertia (about the geom's own centre, geom frame) of a solid primitive."""
    if gtype == "sphere":
        r = size[0]
        vol = 4.0 / 3.0 * math.pi * r ** 3
        m = mass if mass is not None else density * vol
        return m, np.full(3, 0.4 * m * r * r)
    if gtype == "capsule":
        r, h = size[0], 2 * size[1]  # h = cylinder length
        vc, vs = math.pi * r * r * h, 4.0 / 3.0 * math.pi * r ** 3
        m = mass if mass is not None else density * (vc + vs)
        mc, ms = m * vc / (vc + vs), m * vs / (vc + vs)
        izz = 0.5 * mc * r * r + 0.4 * ms * r * r
        ixx = mc * (r * r / 4 + h * h / 12) + ms * (0.4 * r * r + 0.375 * r * h + 0.25 * h * h)
        return m, np.array([ixx, ixx, izz])
    if gtype == "cylinder":
        r, h = size[0], 2 * size[1]
        m = mass if mass is not None else density * math.pi * r * r * h
        ixx = m * (3 * r * r + h * h) / 12
        return m, np.array([ixx, ixx, 0.5 * m * r * r])
    if gtype == "box":
        a, b, c = (2 * s for s in size[:3])
        m = mass if mass is not None else density * a * b * c
        return m, m / 12 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
    if gtype == "ellipsoid":
        a, b, c = size[:3]
        m = mass if mass is not None else density * 4.0 / 3.0 * math.pi * a * b * c
        return m, m / 5 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
    raise ValueError(f"cannot derive an inertia from a geom of type {gtype!r}: give the body an <inertial>")


def _combine_inertia(parts: List[Tuple[float, np.ndarray, np.ndarray, np.ndarray]]):
    """[(mass, principal inertia, pos, quat)] in the body frame -> (mass, ipos, iquat, diagonal inertia)."""
    M = sum(p[0] for p in parts)
    if M <= 0:
        raise ValueError("body has geoms but zero total mass")
    com = sum(p[0] * np.asarray(p[2]) for p in parts) / M
    I = np.zeros((3, 3))
    for m, diag, pos, quat in parts:
        R = _qmat(quat)
        d = np.asarray(pos) - com
        I += R @ np.diag(diag) @ R.T + m * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
    w, V = np.linalg.eigh(I)
    order = np.argsort(-w)  # MuJoCo orders principal moments descending
    w, V = w[order], V[:, order]
    if np.linalg.det(V) < 0:
        V[:, 2] = -V[:, 2]
    return M, com, _mat_to_quat(V), w


# ---------------------------------------------------------------------------
# meshes: vertices -> convex hull
# ---------------------------------------------------------------------------


def _read_mesh_file(path: Path) -> np.ndarray:
    """Vertex positions of a Wavefront OBJ (`v x y z` lines) or an STL (binary or ASCII) file, [V, 3]."""
    raw = path.read_bytes()
    suffix = path.suffix.lower()
    if suffix == ".obj":
        v = [[float(x) for x in line.split()[1:4]] for line in raw.decode("utf-8", "replace").splitlines() if line.startswith("v ")]
        return np.asarray(v, np.float64).reshape(-1, 3)
    if suffix == ".stl":
        if raw[:5].lower() == b"solid" and b"facet" in raw[:1000]:
            v = [[float(x) for x in line.split()[1:4]] for line in raw.decode("utf-8", "replace").splitlines() if line.strip().startswith("vertex")]
            return np.asarray(v, np.float64).reshape(-1, 3)
        n = int(np.frombuffer(raw[80:84], "<u4")[0])
        if len(raw) < 84 + 50 * n:
            raise ValueError(f"{path}: truncated binary STL ({n} triangles announced)")
        tri = np.frombuffer(raw[84:84 + 50 * n], np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]))
        return tri["v"].reshape(-1, 3).astype(np.float64)
    raise ValueError(f"{path}: mesh files must be .obj or .stl (or give the vertices inline: <mesh vertex=...>)")


def convex_hull_vertices(points: np.ndarray, what: str = "mesh", maxhullvert: int = -1) -> np.ndarray:
    """The vertices of the convex hull of `points`, in the order they appear in `points` (what MuJoCo keeps of a mesh for
    collision).  More than MAX_CONVEX_VERTS hull vertices is an error: collision meshes are expected to be decimated - or to say
    `maxhullvert` (MuJoCo >= 3.1.4, the setting its documentation recommends for MJX): the hull is then the one qhull has after
    `maxhullvert - 4` points were added to its initial simplex (MuJoCo hands qhull the option `TA<maxhullvert - 4>`; [3P-recall] of
    user_mesh.cc, like every MuJoCo detail here - scipy's ConvexHull is the same qhull)."""
    from scipy.spatial import ConvexHull, QhullError

    pts = np.asarray(points, np.float64).reshape(-1, 3)
    _, first = np.unique(pts, axis=0, return_index=True)
    pts = pts[np.sort(first)]  # duplicates dropped, the file's vertex order kept (the order decides ties in plane_convex)
    if len(pts) < 4:
        raise ValueError(f"{what}: a mesh needs at least four distinct vertices")
    if maxhullvert != -1 and maxhullvert < 4:
        raise ValueError(f"{what}: maxhullvert must be larger than 3")
    try:
        hull = ConvexHull(pts) if maxhullvert == -1 else ConvexHull(pts, qhull_options=f"Qt TA{maxhullvert - 4}")
    except QhullError as exc:
        raise ValueError(f"{what}: degenerate mesh (flat or collinear): {str(exc).splitlines()[0]}") from exc
    keep = np.sort(hull.vertices)
    if len(keep) > MAX_CONVEX_VERTS:
        raise ValueError(f"{what}: the convex hull has {len(keep)} vertices (limit {MAX_CONVEX_VERTS}); decimate the collision mesh or give the <mesh> maxhullvert=\"{MAX_CONVEX_VERTS}\"")
    return pts[keep]


# ---------------------------------------------------------------------------
# parser
# ---------------------------------------------------------------------------


def _expand_includes(el: ET.Element, cur_dir: Optional[Path], main_dir: Optional[Path], seen: List[Path]) -> None:
    """Replaces every `<include file=...>` below `el`, at any depth, by the children of the included file's root (`<mujoco>` or
    `<mujocoinclude>`), recursively.  MuJoCo resolves the file name against the directory of the MAIN model file; exports that keep
    their parts in sub-directories and include from there are accepted as well (the including file's own directory is tried second).
    The reference copies the robot's whole directory next to the model for exactly this reason (env.py:35-37).  A file may be included
    once (MuJoCo's rule): a second inclusion - a cycle among them - is an error."""
    i = 0
    while i < len(el):
        ch = el[i]
        if ch.tag != "include":
            _expand_includes(ch, cur_dir, main_dir, seen)
            i += 1
            continue
        fname = ch.get("file")
        if not fname:
            raise ValueError("<include> needs a file attribute")
        if main_dir is None:
            raise ValueError(f"<include file={fname!r}> needs the MJCF's directory (load it with load_mjcf / pass base_dir)")
        cands = [Path(main_dir) / fname] + ([Path(cur_dir) / fname] if cur_dir is not None and Path(cur_dir) != Path(main_dir) else [])
        path = next((c for c in cands if c.is_file()), None)
        if path is None:
            raise ValueError(f"<include file={fname!r}>: not found (tried {', '.join(str(c) for c in cands)})")
        path = path.resolve()
        if path in seen:
            raise ValueError(f"<include file={fname!r}>: {path.name} is included more than once")
        seen.append(path)
        inc = ET.fromstring(path.read_text())
        if inc.tag not in ("mujoco", "mujocoinclude"):
            raise ValueError(f"<include file={fname!r}>: root element <{inc.tag}> (mujoco or mujocoinclude expected)")
        _expand_includes(inc, path.parent, main_dir, seen)
        el.remove(ch)
        for k, sub in enumerate(list(inc)):
            el.insert(i + k, sub)
        i += len(inc)


def _merged(root: ET.Element, tag: str) -> Optional[ET.Element]:
    """MJCF sections may repeat (an included file usually brings its own <compiler>, <option>, <actuator> ...): MuJoCo merges them, later
    attributes overriding earlier ones and children accumulating.  Returns one element carrying the merge (None if the section is absent)."""
    els = root.findall(tag)
    if not els:
        return None
    if len(els) == 1:
        return els[0]
    out = ET.Element(tag)
    for e in els:
        out.attrib.update(e.attrib)
        for ch in e:
            if tag == "option" and ch.tag == "flag" and out.find("flag") is not None:
                out.find("flag").attrib.update(ch.attrib)
            else:
                out.append(ch)
    return out


def parse_mjcf(xml: str, name: str = "mjcf", base_dir: Optional[Path] = None) -> ModelSpec:
    root = ET.fromstring(xml)
    _expand_includes(root, base_dir, base_dir, [])
    if root.tag != "mujoco":
        raise ValueError(f"not an MJCF document: root element <{root.tag}>")
    for ch in root:
        if ch.tag in _UNSUPPORTED_TOP:
            raise ValueError(f"<{ch.tag}> is outside the supported MJCF subset")
        if ch.tag not in _IGNORED_TOP | {"compiler", "option", "default", "worldbody", "actuator", "contact", "include"}:
            raise ValueError(f"<{ch.tag}> is outside the supported MJCF subset")
    comp_el = _merged(root, "compiler")
    comp = _Compiler(comp_el)
    dfl = _Defaults(root)
    spec_kw: Dict[str, object] = {}
    opt = _merged(root, "option")
    if opt is not None:
        for k, v in opt.attrib.items():
            if k == "timestep":
                spec_kw["timestep"] = float(v)
            elif k == "gravity":
                spec_kw["gravity"] = tuple(_floats(v, 3, "option gravity"))
            elif k == "impratio":
                spec_kw["impratio"] = float(v)
            elif k in ("solver", "iterations", "ls_iterations"):
                logger.info("<option %s=%s> is overridden by the environment (CG, 6 iterations, 6 line-search iterations; reference env.py:95-97)", k, v)
            elif k in ("integrator",):
                if v != "Euler":
                    raise ValueError(f"<option integrator={v!r}>: only Euler (what MJX + the reference use by default)")
            elif k in ("cone",):
                if v != "pyramidal":
                    raise ValueError("<option cone='elliptic'> is not supported (pyramidal friction cones only)")
            elif k in ("tolerance", "ls_tolerance"):
                spec_kw[k] = float(v)
            elif k in ("jacobian", "noslip_iterations", "mpr_iterations", "o_margin", "density", "viscosity", "wind", "magnetic"):
                if k in ("density", "viscosity") and float(v) != 0:
                    raise ValueError(f"<option {k}> (fluid forces) is not supported")
            else:
                raise ValueError(f"<option {k}=...> is outside the supported MJCF subset")
        if opt.find("flag") is not None:
            for k, v in opt.find("flag").attrib.items():
                if v != {"contact": "enable", "gravity": "enable", "constraint": "enable", "limit": "enable", "actuation": "enable", "clampctrl": "enable",
                         "warmstart": "enable", "frictionloss": "enable", "eulerdamp": "enable", "filterparent": "enable"}.get(k, v) and k not in ("energy", "fwdinv", "multiccd", "island"):
                    raise ValueError(f"<option><flag {k}={v!r}> changes the physics pipeline and is not supported")

    # collision meshes (everything else under <asset> is visual)
    meshes: Dict[str, np.ndarray] = {}
    mesh_maxhull: Dict[str, int] = {}
    meshdir = ""
    if comp_el is not None:
        meshdir = comp_el.get("meshdir", comp_el.get("assetdir", ""))
    for asset in root.findall("asset"):
        for me in asset.findall("mesh"):
            mname = me.get("name") or (Path(me.get("file", "")).stem if me.get("file") else None)
            if not mname:
                raise ValueError("<asset><mesh> needs a name (or a file to take it from)")
            ma = dfl.resolve("mesh", me, None)  # (<default><mesh scale=... maxhullvert=...>: millimetre STL files are scaled there as a rule)
            scale = np.array(_floats(ma.get("scale", "1 1 1"), 3, f"mesh {mname} scale"))
            mesh_maxhull[mname] = int(ma.get("maxhullvert", "-1"))
            if "vertex" in me.attrib:
                pts = np.asarray(_floats(me.get("vertex")), np.float64)
                if pts.size % 3 or pts.size < 12:
                    raise ValueError(f"mesh {mname}: vertex needs 3 numbers per vertex, at least four vertices")
                pts = pts.reshape(-1, 3)
            elif "file" in me.attrib:
                if base_dir is None:
                    raise ValueError(f"mesh {mname}: file={me.get('file')!r} needs the MJCF's directory (load it with load_mjcf / pass base_dir)")
                pts = _read_mesh_file(Path(base_dir) / meshdir / (Path(me.get("file")).name if comp.strippath else me.get("file")))
            else:
                raise ValueError(f"mesh {mname}: neither vertex nor file")
            if any(k in ma for k in ("refpos", "refquat")):
                raise ValueError(f"mesh {mname}: refpos / refquat are not supported")
            meshes[mname] = pts * scale[None, :]

    world = _merged(root, "worldbody")
    if world is None:
        raise ValueError("MJCF has no <worldbody>")
    bodies: List[BodySpec] = []
    plane: Optional[Dict[str, object]] = None
    solrefs = {"limit": set(), "contact": set()}
    paired_names = {el.get(k) for sec in root.findall("contact") for el in sec.findall("pair") for k in ("geom1", "geom2")}
    free_z: List[float] = []

    def geom_spec(a: Dict[str, str], what: str):
        """-> (kind, GeomSpec or None, inertia part or None); kind in {'plane', 'collide', 'inert'}"""
        gtype = a.get("type", "sphere")
        contype, conaff = int(a.get("contype", "1")), int(a.get("conaffinity", "1"))
        collides = (contype | conaff) != 0 or (a.get("name") is not None and a.get("name") in paired_names)  # (an explicit pair needs no masks)
        if collides and int(a.get("priority", "0")) != 0:
            raise ValueError(f"{what}: geom priority is not supported (a pair's friction is the larger of the two geoms', MuJoCo's rule for equal priorities)")
        quat = comp.orientation(a, what)
        pos = np.array(_floats(a.get("pos", "0 0 0"), 3, what))
        size = _floats(a["size"]) if "size" in a else []
        if gtype == "plane":
            return "plane", dict(contype=contype, conaffinity=conaff, z=float(pos[2]), friction=tuple((_floats(a.get("friction", "1 0.005 0.0001")) + [0.005, 0.0001])[:3]) if "friction" in a else (1.0, 0.005, 0.0001),
                                 quat=quat, a=a), None
        if "fromto" in a:
            if gtype not in ("capsule", "cylinder", "box", "ellipsoid"):
                raise ValueError(f"{what}: fromto on a {gtype}")
            ft = np.array(_floats(a["fromto"], 6, what))
            pos = 0.5 * (ft[:3] + ft[3:])
            quat = _z_to(ft[3:] - ft[:3])
            half = 0.5 * float(np.linalg.norm(ft[3:] - ft[:3]))
            size = [size[0], half] if gtype in ("capsule", "cylinder") else size
        if gtype == "mesh" or ("mesh" in a and "type" not in a):
            if not collides:
                return "inert", None, None  # visual only; a mesh contributes no inertia here (the body needs an <inertial>)
            mname = a.get("mesh")
            if mname not in meshes:
                raise ValueError(f"{what}: mesh {mname!r} is not defined under <asset> (with vertex=... or an .obj / .stl file)")
            if "margin" in a and float(a["margin"]) != 0 or "gap" in a and float(a["gap"]) != 0:
                raise ValueError(f"{what}: contact margin / gap are not supported")
            if int(a.get("condim", "3")) != 3:
                raise ValueError(f"{what}: condim {a['condim']} (only 3: pyramidal sliding friction)")
            for k in ("solref", "solimp"):
                if k in a:
                    solrefs["contact"].add((k, tuple(_floats(a[k]))))
            fr = tuple((_floats(a["friction"]) + [0.005, 0.0001])[:3]) if "friction" in a else (1.0, 0.005, 0.0001)
            # MuJoCo re-centres a mesh on its centre of mass and principal axes and compensates in the geom's pose: the shape in the
            # body frame is unchanged, so the hull is kept in the file's own mesh frame under the geom's pos / quat as written
            gs = GeomSpec(GEOM_MESH, (), pos=tuple(pos), quat=tuple(quat), friction=fr, contype=contype, conaffinity=conaff, name=a.get("name", ""),
                          vertices=tuple(map(tuple, convex_hull_vertices(meshes[mname], f"mesh {mname}", mesh_maxhull.get(mname, -1)))))
            return "collide", gs, None
        need = {"sphere": 1, "capsule": 2, "cylinder": 2, "box": 3, "ellipsoid": 3}.get(gtype)
        if need is None:
            raise ValueError(f"{what}: geom type {gtype!r} is outside the supported MJCF subset")
        if len(size) < need:
            raise ValueError(f"{what}: a {gtype} needs {need} size value(s)")
        mass = float(a["mass"]) if "mass" in a else None
        part = None
        m, diag = _geom_mass_inertia(gtype, size, float(a.get("density", _DEFAULT_DENSITY)), mass)
        part = (m, diag, pos, quat)
        if not collides:
            return "inert", None, part
        if gtype not in ("sphere", "capsule", "cylinder", "box"):
            raise ValueError(f"{what}: only sphere, capsule, cylinder and box geoms can collide (a colliding {gtype} needs contype='0' conaffinity='0' or a capsule approximation)")
        for k in ("solref", "solimp"):
            if k in a:
                solrefs["contact"].add((k, tuple(_floats(a[k]))))
        if "margin" in a and float(a["margin"]) != 0 or "gap" in a and float(a["gap"]) != 0:
            raise ValueError(f"{what}: contact margin / gap are not supported")
        if int(a.get("condim", "3")) != 3:
            raise ValueError(f"{what}: condim {a['condim']} (only 3: pyramidal sliding friction)")
        fr = tuple((_floats(a["friction"]) + [0.005, 0.0001])[:3]) if "friction" in a else (1.0, 0.005, 0.0001)
        gs = GeomSpec({"sphere": GEOM_SPHERE, "capsule": GEOM_CAPSULE, "cylinder": GEOM_CYLINDER, "box": GEOM_BOX}[gtype], tuple(size[:need]), pos=tuple(pos), quat=tuple(quat), friction=fr,
                      contype=contype, conaffinity=conaff, name=a.get("name", ""))
        return "collide", gs, part

    def walk(el: ET.Element, parent: str, childclass: Optional[str]) -> None:
        nonlocal plane
        bname = el.get("name") or f"body{len(bodies) + 1}"
        cc = el.get("childclass", childclass)
        for k in el.attrib:
            if k not in ("name", "pos", "quat", "euler", "axisangle", "xyaxes", "zaxis", "childclass", "mocap", "gravcomp", "user"):
                raise ValueError(f"<body {bname}> attribute {k!r} is outside the supported MJCF subset")
        if el.get("mocap", "false") == "true" or float(el.get("gravcomp", "0")) != 0:
            raise ValueError(f"<body {bname}>: mocap / gravcomp are not supported")
        pos = tuple(_floats(el.get("pos", "0 0 0"), 3, f"body {bname} pos"))
        quat = comp.orientation(el.attrib, f"body {bname}")
        joints: List[JointSpec] = []
        geoms: List[GeomSpec] = []
        parts = []
        inertial = None
        for ch in el:
            what = f"<{ch.tag}> in body {bname}"
            if ch.tag == "inertial":
                a = ch.attrib
                ipos = tuple(_floats(a.get("pos", "0 0 0"), 3, what))
                iquat = comp.orientation(a, what)
                mass = float(a["mass"])
                if "diaginertia" in a:
                    diag = np.array(_floats(a["diaginertia"], 3, what))
                elif "fullinertia" in a:
                    f = _floats(a["fullinertia"], 6, what)  # xx yy zz xy xz yz
                    I = np.array([[f[0], f[3], f[4]], [f[3], f[1], f[5]], [f[4], f[5], f[2]]])
                    w, V = np.linalg.eigh(I)
                    order = np.argsort(-w)
                    w, V = w[order], V[:, order]
                    if np.linalg.det(V) < 0:
                        V[:, 2] = -V[:, 2]
                    diag, iquat = w, _qmul(iquat, _mat_to_quat(V))
                else:
                    raise ValueError(f"{what}: diaginertia or fullinertia required")
                inertial = (mass, ipos, tuple(_normalize(iquat)), tuple(diag))
            elif ch.tag in ("joint", "freejoint"):
                if ch.tag == "freejoint":
                    joints.append(JointSpec(ch.get("name", f"{bname}_free"), JNT_FREE))
                    continue
                a = dfl.resolve("joint", ch, cc)
                if "frictionloss" in ch.attrib and float(ch.attrib["frictionloss"]) != 0:
                    # The reference deletes `frictionloss` from <default><joint> before MuJoCo sees the file, because Brax / MJX reject a model
                    # with dry joint friction (env.py:41-45); an export that carries it on the joints themselves would stop it at
                    # `mjcf.load_model`.  The same treatment here - stripped, said once per joint - so that such a file trains at all: the
                    # physics then has NO dry friction in that joint (MuJoCo would add a friction-loss constraint row).
                    logger.warning("%s: frictionloss=%s is dropped (the engine, like MJX behind the reference, models no dry joint friction; "
                                   "the reference strips the attribute from <default><joint>, env.py:41-45)", what, ch.attrib["frictionloss"])
                a.pop("frictionloss", None)
                jt = {"free": JNT_FREE, "hinge": JNT_HINGE, "slide": JNT_SLIDE}.get(a.get("type", "hinge"))
                if jt is None:
                    raise ValueError(f"{what}: joint type {a.get('type')!r} is outside the supported MJCF subset (free, hinge, slide)")
                jn = a.get("name", f"{bname}_joint{len(joints)}")
                rng = None
                if "range" in a:
                    lim = a.get("limited", "auto")
                    if lim == "true" or (lim == "auto" and comp.autolimits):
                        r = _floats(a["range"], 2, what)
                        rng = (comp.ang(r[0]), comp.ang(r[1])) if jt == JNT_HINGE else (r[0], r[1])
                elif a.get("limited") == "true":
                    raise ValueError(f"{what}: limited='true' without range")
                for k in ("solreflimit", "solimplimit"):
                    if k in a:
                        solrefs["limit"].add((k, tuple(_floats(a[k]))))
                if float(a.get("margin", "0")) != 0:
                    raise ValueError(f"{what}: joint margin is not supported")
                for k in a:
                    if k not in ("name", "type", "pos", "axis", "range", "limited", "ref", "damping", "armature", "stiffness", "solreflimit", "solimplimit",
                                 "springref", "margin", "group", "user", "actuatorfrcrange", "actuatorfrclimited"):
                        raise ValueError(f"{what}: attribute {k!r} is outside the supported MJCF subset")
                ref, sref = float(a.get("ref", "0")), float(a.get("springref", "0"))  # MuJoCo: qpos0 = ref, qpos_spring = springref (both default 0)
                # actuatorfrcrange: the joint's total actuator force is clamped (what URDF-derived exports make of an <effort> limit)
                frc = None
                if "actuatorfrcrange" in a:
                    lim = a.get("actuatorfrclimited", "auto")
                    if lim == "true" or (lim == "auto" and comp.autolimits):
                        if jt == JNT_FREE:
                            raise ValueError(f"{what}: actuatorfrcrange on a free joint")
                        frc = tuple(_floats(a["actuatorfrcrange"], 2, what))
                elif a.get("actuatorfrclimited") == "true":
                    raise ValueError(f"{what}: actuatorfrclimited='true' without actuatorfrcrange")
                joints.append(JointSpec(jn, jt, pos=tuple(_floats(a.get("pos", "0 0 0"), 3, what)), axis=tuple(_floats(a.get("axis", "0 0 1"), 3, what)), range=rng,
                                        damping=float(a.get("damping", "0")), armature=float(a.get("armature", "0")), stiffness=float(a.get("stiffness", "0")),
                                        ref=comp.ang(ref) if jt == JNT_HINGE else ref, springref=comp.ang(sref) if jt == JNT_HINGE else sref, actuatorfrcrange=frc))
            elif ch.tag == "geom":
                kind, gs, part = geom_spec(dfl.resolve("geom", ch, cc), what)
                if kind == "plane":
                    raise ValueError(f"{what}: a plane belongs to the worldbody")
                if gs is not None:
                    geoms.append(gs)
                if part is not None:
                    parts.append(part)
            elif ch.tag == "body":
                pass
            elif ch.tag in ("site", "camera", "light"):
                pass
            else:
                raise ValueError(f"{what} is outside the supported MJCF subset")
        use_geoms = comp.inertiafromgeom == "true" or (comp.inertiafromgeom == "auto" and inertial is None)
        if use_geoms:
            if not parts:
                raise ValueError(f"body {bname}: no <inertial> and no geom to derive one from")
            m, ipos, iquat, diag = _combine_inertia(parts)
            inertial = (m, tuple(ipos), tuple(iquat), tuple(diag))
        elif inertial is None:
            raise ValueError(f"body {bname}: no <inertial> (inertiafromgeom='false')")
        if any(j.type == JNT_FREE for j in joints):
            free_z.append(pos[2])
            if len(free_z) == 1:  # ModelSpec keeps the height of the first free root separately (qpos0[2] = free_root_z)
                pos = (pos[0], pos[1], 0.0)
        if comp.fusestatic and not joints:
            raise ValueError(f"body {bname} has no joint and <compiler fusestatic='true'> would merge it into its parent (renumbering the bodies the observation "
                             "lists): not supported - set fusestatic='false' (MJCF's default) or merge the body by hand")
        mass_, inertia_ = float(inertial[0]), [float(x) for x in inertial[3]]
        if comp.balanceinertia and (inertia_[0] + inertia_[1] < inertia_[2] or inertia_[0] + inertia_[2] < inertia_[1] or inertia_[1] + inertia_[2] < inertia_[0]):
            inertia_ = [sum(inertia_) / 3.0] * 3   # MuJoCo: an inertia that violates A + B >= C gets three equal moments
        mass_ = max(mass_, comp.boundmass)
        inertia_ = [max(x, comp.boundinertia) for x in inertia_]
        inertial = (mass_, inertial[1], inertial[2], tuple(inertia_))
        bodies.append(BodySpec(bname, parent, pos=pos, quat=tuple(quat), mass=inertial[0], inertia=inertial[3], ipos=inertial[1], iquat=inertial[2],
                               joints=joints, geoms=geoms))
        for ch in el.findall("body"):
            walk(ch, bname, cc)

    for ch in world:
        if ch.tag == "geom":
            kind, info, _ = geom_spec(dfl.resolve("geom", ch, None), "<geom> in worldbody")
            if kind != "plane":
                if (int(ch.get("contype", "1")) | int(ch.get("conaffinity", "1"))) != 0:
                    raise ValueError("worldbody geoms other than one ground plane must be non-colliding")
                continue
            if plane is not None:
                raise ValueError("more than one ground plane")
            if abs(abs(float(info["quat"][0])) - 1.0) > 1e-9:
                raise ValueError("the ground plane must be horizontal (normal +z)")
            plane = info
        elif ch.tag == "body":
            walk(ch, "world", None)
        elif ch.tag in ("light", "camera", "site"):
            continue
        else:
            raise ValueError(f"<{ch.tag}> in worldbody is outside the supported MJCF subset")
    if not bodies:
        raise ValueError("MJCF has no bodies")

    free_root_z = float(free_z[0]) if free_z else 1.0
    for st in root.findall("statistic"):   # (extent, center, meansize, meanmass only scale the visualisation; meaninertia scales the solver's tolerance)
        if "meaninertia" in st.attrib:
            spec_kw["meaninertia"] = float(st.get("meaninertia"))
    if comp.settotalmass > 0:   # every mass and inertia scaled so that the robot weighs this much
        scale = comp.settotalmass / sum(b.mass for b in bodies)
        for b in bodies:
            b.mass, b.inertia = b.mass * scale, tuple(x * scale for x in b.inertia)

    joint_names = {j.name for b in bodies for j in b.joints}
    acts: List[ActuatorSpec] = []
    act_el = _merged(root, "actuator")
    for ch in (act_el if act_el is not None else []):
        what = f"<actuator><{ch.tag}>"
        if ch.tag not in ("position", "motor", "velocity", "general"):
            raise ValueError(f"{what} is outside the supported MJCF subset (position, motor, velocity, general)")
        a = dfl.resolve(ch.tag, ch, None)
        if "joint" not in a:
            raise ValueError(f"{what}: only joint transmissions are supported")
        if a["joint"] not in joint_names:
            raise ValueError(f"{what}: unknown joint {a['joint']!r}")
        gear = _floats(a.get("gear", "1"))[0]

        def rng(key: str, flag: str):
            if key not in a:
                if a.get(flag) == "true":
                    raise ValueError(f"{what}: {flag}='true' without {key}")
                return None
            lim = a.get(flag, "auto")
            if lim == "true" or (lim == "auto" and comp.autolimits):
                r = _floats(a[key], 2, what)
                return (r[0], r[1])
            return None

        own = {"position": ("kp", "kv"), "motor": (), "velocity": ("kv",), "general": ("gaintype", "biastype", "dyntype", "gainprm", "biasprm")}[ch.tag]
        for k in a:
            if k not in ("name", "joint", "gear", "ctrlrange", "ctrllimited", "forcerange", "forcelimited", "group", "user") + own:
                raise ValueError(f"{what}: attribute {k!r} is outside the supported MJCF subset")
        kw = dict(gear=gear, ctrlrange=rng("ctrlrange", "ctrllimited"), forcerange=rng("forcerange", "forcelimited"))
        if ch.tag == "position":
            kw.update(kp=float(a.get("kp", "1")), kv=float(a.get("kv", "0")))
        elif ch.tag == "velocity":      # force = kv * (ctrl - velocity)
            kv = float(a.get("kv", "1"))
            kw.update(gain=kv, bias=(0.0, 0.0, -kv))
        elif ch.tag == "general":       # force = gainprm[0] * ctrl + biasprm[0] + biasprm[1] * length + biasprm[2] * velocity
            if a.get("dyntype", "none") != "none" or a.get("gaintype", "fixed") != "fixed" or a.get("biastype", "none") not in ("none", "affine"):
                raise ValueError(f"{what}: only dyntype='none', gaintype='fixed', biastype='none' | 'affine' are supported")
            gp = (_floats(a.get("gainprm", "1")) + [0.0])[0]
            bp = (_floats(a.get("biasprm", "0 0 0")) + [0.0, 0.0, 0.0])[:3] if a.get("biastype", "none") == "affine" else [0.0, 0.0, 0.0]
            kw.update(gain=gp, bias=tuple(bp))
        acts.append(ActuatorSpec(a["joint"], **kw))

    for kind, key_ref, key_imp, field_ref, field_imp in (("limit", "solreflimit", "solimplimit", "limit_solref", "limit_solimp"),
                                                         ("contact", "solref", "solimp", "contact_solref", "contact_solimp")):
        refs = {v for k, v in solrefs[kind] if k == key_ref}
        imps = {v for k, v in solrefs[kind] if k == key_imp}
        if len(refs) > 1 or len(imps) > 1:
            raise ValueError(f"per-element {key_ref} / {key_imp} values differ; the engine keeps one {kind} solref / solimp per model")
        if refs:
            spec_kw[field_ref] = tuple(refs.pop())
        if imps:
            v = list(imps.pop())
            spec_kw[field_imp] = tuple(v + list(_MJ_SOLIMP[len(v):]))
    if plane is not None:
        spec_kw["plane_z"] = plane["z"]
        spec_kw["plane_friction"] = plane["friction"]
        spec_kw["plane_contype"], spec_kw["plane_conaffinity"] = plane["contype"], plane["conaffinity"]
        spec_kw["plane_name"] = plane["a"].get("name", "")
        for k in ("solref", "solimp"):
            if k in plane["a"] and (k, tuple(_floats(plane["a"][k]))) not in solrefs["contact"]:
                logger.warning("ground plane %s is ignored: contact parameters are taken from the robot's geoms (MuJoCo mixes both by solmix)", k)
    else:
        spec_kw["has_plane"] = False
        logger.warning("MJCF has no ground plane: only geom-geom contacts will be generated")
    # <contact>: <exclude body1 body2/> removes every geom pair between two bodies (what exports use where neighbouring collision shapes overlap
    # at rest); <pair geom1 geom2 friction .../> adds a geom pair whatever masks, kinship or excludes say, with contact parameters of its OWN
    # (MuJoCo does not look at the geoms' for an explicit pair): condim 3, no margin / gap, one sliding friction for both tangents, and the
    # model's one contact solref / solimp - anything else is a loud error
    known = {b.name for b in bodies}
    excludes: List[Tuple[str, str]] = []
    pairs: List[Tuple[str, str, Optional[float]]] = []
    for sec in root.findall("contact"):
        for el in sec:
            if el.tag == "pair":
                g1, g2 = el.get("geom1"), el.get("geom2")
                what = f"<contact><pair geom1={g1!r} geom2={g2!r}>"
                if not g1 or not g2:
                    raise ValueError("<contact><pair> needs geom1 and geom2")
                for k in el.attrib:
                    if k not in ("name", "class", "geom1", "geom2", "condim", "friction", "solref", "solimp", "margin", "gap", "solreffriction"):
                        raise ValueError(f"{what}: attribute {k!r} is outside the supported MJCF subset")
                if "class" in el.attrib or "solreffriction" in el.attrib:
                    raise ValueError(f"{what}: class / solreffriction are not supported")
                if int(el.get("condim", "3")) != 3:
                    raise ValueError(f"{what}: condim {el.get('condim')} (only 3: pyramidal sliding friction)")
                if float(el.get("margin", "0")) != 0 or float(el.get("gap", "0")) != 0:
                    raise ValueError(f"{what}: contact margin / gap are not supported")
                fr5 = (_floats(el.get("friction", "")) + [1.0, 1.0, 0.005, 0.0001, 0.0001][len(_floats(el.get("friction", ""))):])[:5] if el.get("friction") else [1.0, 1.0, 0.005, 0.0001, 0.0001]
                if fr5[0] != fr5[1]:
                    raise ValueError(f"{what}: friction {fr5[0]} / {fr5[1]} - the two tangent directions of a contact share one coefficient here")
                want_ref = tuple(spec_kw.get("contact_solref", _MJ_SOLREF))
                want_imp = tuple(spec_kw.get("contact_solimp", _MJ_SOLIMP))
                got_ref = tuple(_floats(el.get("solref"))) if el.get("solref") else _MJ_SOLREF
                got_imp = tuple(_floats(el.get("solimp")) + list(_MJ_SOLIMP[len(_floats(el.get("solimp"))):])) if el.get("solimp") else _MJ_SOLIMP
                if tuple(got_ref) != want_ref or tuple(got_imp) != want_imp:
                    raise ValueError(f"{what}: solref / solimp {got_ref} / {got_imp} (MuJoCo's defaults where the pair gives none: a pair does not take the geoms') differ from "
                                     f"the model's contact values {want_ref} / {want_imp}; the engine keeps one contact solref / solimp per model")
                pairs.append((g1, g2, float(fr5[0])))
                continue
            if el.tag != "exclude":
                raise ValueError(f"<contact><{el.tag}> is outside the supported MJCF subset (only <exclude body1 body2/> and <pair geom1 geom2/>)")
            b1, b2 = el.get("body1"), el.get("body2")
            if b1 not in known or b2 not in known:
                raise ValueError(f"<contact><exclude body1={b1!r} body2={b2!r}>: unknown body")
            excludes.append((b1, b2))
    if excludes:
        spec_kw["contact_excludes"] = excludes
    if pairs:
        spec_kw["contact_pairs"] = pairs
    return ModelSpec(name=name, bodies=bodies, actuators=acts, free_root_z=free_root_z, **spec_kw)


def load_mjcf(path: str) -> ModelSpec:
    p = Path(path)
    return parse_mjcf(p.read_text(), name=p.stem, base_dir=p.parent)


# ---------------------------------------------------------------------------
# writer: ModelSpec -> MJCF (round-trip tests; lets a user run the stand-in robots in real MuJoCo / MJX)
# ---------------------------------------------------------------------------


def _fmt(v) -> str:
    return " ".join(repr(float(x)) for x in np.atleast_1d(v))


def to_mjcf(spec: ModelSpec) -> str:
    root = ET.Element("mujoco", model=spec.name)
    ET.SubElement(root, "compiler", angle="radian", autolimits="true", inertiafromgeom="false")
    ET.SubElement(root, "option", timestep=repr(float(spec.timestep)), gravity=_fmt(spec.gravity), impratio=repr(float(spec.impratio)), solver="CG",
                  iterations=str(spec.iterations), ls_iterations=str(spec.ls_iterations), tolerance=repr(float(spec.tolerance)), ls_tolerance=repr(float(spec.ls_tolerance)))
    if spec.meaninertia is not None:
        ET.SubElement(root, "statistic", meaninertia=repr(float(spec.meaninertia)))
    d = ET.SubElement(ET.SubElement(root, "default"), "joint", solreflimit=_fmt(spec.limit_solref), solimplimit=_fmt(spec.limit_solimp))
    del d
    ET.SubElement(root.find("default"), "geom", solref=_fmt(spec.contact_solref), solimp=_fmt(spec.contact_solimp), condim="3")
    mesh_geoms = [(b.name, gi, g) for b in spec.bodies for gi, g in enumerate(b.geoms) if g.type == GEOM_MESH]
    if mesh_geoms:
        asset = ET.SubElement(root, "asset")
        for bname, gi, g in mesh_geoms:
            ET.SubElement(asset, "mesh", name=f"{bname}_mesh{gi}", vertex=" ".join(_fmt(v) for v in g.vertices))
    world = ET.SubElement(root, "worldbody")
    if spec.has_plane:
        ET.SubElement(world, "geom", name="floor", type="plane", size="0 0 1", pos=f"0 0 {float(spec.plane_z)!r}", friction=_fmt(spec.plane_friction),
                      contype=str(spec.plane_contype), conaffinity=str(spec.plane_conaffinity))
    els = {"world": world}
    first_free = True
    for b in spec.bodies:
        pos = list(b.pos)
        if any(j.type == JNT_FREE for j in b.joints) and first_free:
            pos[2] = spec.free_root_z
            first_free = False
        e = ET.SubElement(els[b.parent], "body", name=b.name, pos=_fmt(pos), quat=_fmt(b.quat))
        els[b.name] = e
        ET.SubElement(e, "inertial", pos=_fmt(b.ipos), quat=_fmt(b.iquat), mass=repr(float(b.mass)), diaginertia=_fmt(b.inertia))
        for j in b.joints:
            if j.type == JNT_FREE:
                ET.SubElement(e, "freejoint", name=j.name)
                continue
            a = dict(name=j.name, type="hinge" if j.type == JNT_HINGE else "slide", pos=_fmt(j.pos), axis=_fmt(j.axis), damping=repr(float(j.damping)),
                     armature=repr(float(j.armature)), stiffness=repr(float(j.stiffness)), ref=repr(float(j.ref)),
                     springref=repr(float(j.ref if j.springref is None else j.springref)))
            if j.actuatorfrcrange is not None:
                a.update(actuatorfrcrange=_fmt(j.actuatorfrcrange), actuatorfrclimited="true")
            if j.range is not None:
                a["range"] = _fmt(j.range)
            ET.SubElement(e, "joint", **a)
        for gi, g in enumerate(b.geoms):
            if g.type == GEOM_MESH:
                ET.SubElement(e, "geom", type="mesh", mesh=f"{b.name}_mesh{gi}", pos=_fmt(g.pos), quat=_fmt(g.quat), friction=_fmt(g.friction), contype=str(g.contype),
                              conaffinity=str(g.conaffinity), **({"name": g.name} if g.name else {}))
                continue
            ET.SubElement(e, "geom", type={GEOM_SPHERE: "sphere", GEOM_CAPSULE: "capsule", GEOM_CYLINDER: "cylinder", GEOM_BOX: "box"}[g.type], size=_fmt(g.size), pos=_fmt(g.pos), quat=_fmt(g.quat), friction=_fmt(g.friction),
                          contype=str(g.contype), conaffinity=str(g.conaffinity), **({"name": g.name} if g.name else {}))
    act = ET.SubElement(root, "actuator")
    for a in spec.actuators:
        kw = dict(joint=a.joint, gear=repr(float(a.gear)))
        if a.ctrlrange is not None:
            kw["ctrlrange"] = _fmt(a.ctrlrange)
        if a.forcerange is not None:
            kw["forcerange"] = _fmt(a.forcerange)
        if a.gain is not None:
            ET.SubElement(act, "general", gainprm=repr(float(a.gain)), biastype="affine", biasprm=_fmt(a.bias), **kw)
        elif a.kp != 0:
            ET.SubElement(act, "position", kp=repr(float(a.kp)), kv=repr(float(a.kv)), **kw)
        else:
            ET.SubElement(act, "motor", **kw)
    if spec.contact_excludes or spec.contact_pairs:
        con = ET.SubElement(root, "contact")
        for b1, b2 in spec.contact_excludes:
            ET.SubElement(con, "exclude", body1=b1, body2=b2)
        for g1, g2, mu in spec.contact_pairs:
            ET.SubElement(con, "pair", geom1=g1, geom2=g2, solref=_fmt(spec.contact_solref), solimp=_fmt(spec.contact_solimp),
                          **({} if mu is None else {"friction": _fmt([mu, mu, 0.005, 0.0001, 0.0001])}))
    ET.indent(root)
    return ET.tostring(root, encoding="unicode")
