"""MJCF subset compiler (`minppo_amd/mjcf.py`; stands where reference env.py:27-50 loads the robot file)."""
import math

from pathlib import Path

import numpy as np
import pytest

from minppo_amd import mjcf
from minppo_amd.model import BUILTIN_MODELS, GEOM_CAPSULE, GEOM_SPHERE, JNT_FREE, JNT_HINGE, compile_model, load_model


@pytest.mark.parametrize("name", sorted(BUILTIN_MODELS))
def test_round_trip_through_mjcf_reproduces_the_compiled_tables(name):
    spec = BUILTIN_MODELS[name]()
    xml = mjcf.to_mjcf(spec)
    back = mjcf.parse_mjcf(xml, name=spec.name)
    a, b = compile_model(spec), compile_model(back)
    assert a.t.keys() == b.t.keys()
    for k in a.t:
        np.testing.assert_allclose(np.asarray(b.t[k], dtype=np.float64), np.asarray(a.t[k], dtype=np.float64), rtol=1e-12, atol=1e-12, err_msg=k)
    assert a.to_blob() == b.to_blob()
    assert a.joint_names == b.joint_names and a.body_names == b.body_names


HAND = """
<mujoco model="hand">
  <compiler angle="degree" eulerseq="xyz"/>
  <option timestep="0.001" gravity="0 0 -9.81" solver="Newton" iterations="50"/>
  <default>
    <joint damping="0.3" armature="0.05" frictionloss="0.1"/>
    <geom friction="0.8" density="500"/>
    <position kp="25" ctrlrange="-1 1"/>
    <default class="knee">
      <joint range="-90 0" damping="0.7"/>
    </default>
  </default>
  <asset><texture name="t" type="2d" builtin="checker" width="8" height="8"/></asset>
  <worldbody>
    <light pos="0 0 3"/>
    <geom type="plane" size="5 5 0.1" pos="0 0 0" friction="1.2"/>
    <body name="base" pos="0.1 -0.2 0.75" euler="0 0 90">
      <freejoint name="root"/>
      <geom type="sphere" size="0.1"/>
      <body name="thigh" pos="0 0 -0.1" childclass="knee">
        <inertial pos="0 0 -0.15" mass="2.0" fullinertia="0.02 0.03 0.01 0.001 0 0"/>
        <joint name="hip" axis="0 1 0" range="-45 45" class="main"/>
        <geom type="capsule" size="0.04" fromto="0 0 0 0 0 -0.3"/>
        <body name="shin" pos="0 0 -0.3">
          <joint name="knee" axis="0 1 0"/>
          <geom type="capsule" size="0.03 0.1" pos="0 0 -0.1"/>
          <geom type="box" size="0.05 0.02 0.01" pos="0.02 0 -0.22" contype="0" conaffinity="0"/>
        </body>
      </body>
    </body>
  </worldbody>
  <actuator>
    <position joint="hip"/>
    <position joint="knee" kp="40" forcerange="-30 30"/>
  </actuator>
</mujoco>
"""


def test_hand_written_mjcf_defaults_units_and_inertia():
    s = mjcf.parse_mjcf(HAND, "hand")
    assert s.timestep == 0.001 and s.iterations == 6 and s.ls_iterations == 6   # solver settings are forced (env.py:95-97)
    assert s.plane_friction[0] == 1.2 and s.plane_z == 0.0 and s.free_root_z == 0.75
    base, thigh, shin = s.bodies
    # euler 0 0 90 deg about z
    np.testing.assert_allclose(base.quat, [math.cos(math.pi / 4), 0, 0, math.sin(math.pi / 4)], atol=1e-12)
    assert base.joints[0].type == JNT_FREE
    # sphere r=0.1 at density 500: m = 4/3 pi r^3 rho, I = 2/5 m r^2
    m = 4 / 3 * math.pi * 1e-3 * 500
    assert base.mass == pytest.approx(m) and base.inertia == pytest.approx([0.4 * m * 0.01] * 3)
    assert base.geoms[0].friction[0] == 0.8
    # thigh: explicit inertial wins over its geom; fullinertia diagonalised (principal moments descending)
    assert thigh.mass == 2.0
    w = np.linalg.eigvalsh(np.array([[0.02, 0.001, 0], [0.001, 0.03, 0], [0, 0, 0.01]]))[::-1]
    np.testing.assert_allclose(thigh.inertia, w, rtol=1e-12)
    hip = thigh.joints[0]
    assert hip.type == JNT_HINGE and hip.damping == 0.3 and hip.armature == 0.05           # class="main" beats childclass
    np.testing.assert_allclose(hip.range, np.radians([-45, 45]))
    g = thigh.geoms[0]
    assert g.type == GEOM_CAPSULE and g.size == pytest.approx((0.04, 0.15))
    np.testing.assert_allclose(g.pos, [0, 0, -0.15], atol=1e-12)
    np.testing.assert_allclose(np.abs(g.quat), [0, 1, 0, 0], atol=1e-12)                # local z -> -z
    knee = shin.joints[0]
    assert knee.damping == 0.7 and knee.armature == 0.05                                # childclass "knee" inherits armature
    np.testing.assert_allclose(knee.range, np.radians([-90, 0]))
    # shin inertia from its capsule AND the non-colliding box; only the capsule collides
    assert len(shin.geoms) == 1 and shin.geoms[0].type == GEOM_CAPSULE
    r, h = 0.03, 0.2
    mc = 500 * (math.pi * r * r * h + 4 / 3 * math.pi * r ** 3)
    mb = 500 * 0.1 * 0.04 * 0.02
    assert shin.mass == pytest.approx(mc + mb)
    com_z = (mc * -0.1 + mb * -0.22) / (mc + mb)
    assert shin.ipos[2] == pytest.approx(com_z) and shin.ipos[0] == pytest.approx(mb * 0.02 / (mc + mb))
    a0, a1 = s.actuators
    assert (a0.joint, a0.kp, a0.ctrlrange, a0.forcerange) == ("hip", 25.0, (-1.0, 1.0), None)
    assert (a1.joint, a1.kp, a1.ctrlrange, a1.forcerange) == ("knee", 40.0, (-1.0, 1.0), (-30.0, 30.0))
    cm = compile_model(s)
    assert (cm.nq, cm.nv, cm.nu) == (9, 8, 2)


def test_frictionloss_is_stripped_like_the_reference_strips_it(caplog):
    """<default><joint frictionloss> is deleted silently (reference env.py:41-45); on a joint itself - where the reference would let Brax
    refuse the model - it is dropped with a warning that names the joint: same compiled model either way."""
    import logging

    base = mjcf.parse_mjcf(HAND)
    with caplog.at_level(logging.WARNING, logger="minppo_amd.mjcf"):
        caplog.clear()
        mjcf.parse_mjcf(HAND)
        assert not [r for r in caplog.records if "frictionloss" in r.getMessage()]
        worn = mjcf.parse_mjcf(HAND.replace('<joint name="knee" axis="0 1 0"/>', '<joint name="knee" axis="0 1 0" frictionloss="0.2"/>'))
        msgs = [r.getMessage() for r in caplog.records if "frictionloss" in r.getMessage()]
    assert len(msgs) == 1 and "shin" in msgs[0] and "0.2" in msgs[0]
    a, b = compile_model(base), compile_model(worn)
    assert a.to_blob(True) == b.to_blob(True)


def test_joint_force_limits_and_the_other_actuator_kinds():
    """<joint actuatorfrcrange> (what URDF-derived exports make of an effort limit) clamps the joint's TOTAL actuator force - MJX fwd_actuation;
    <velocity> and <general biastype="affine"> actuators are gain / bias rows like <position>.  Known answers through the oracle."""
    from oracle.physics_oracle import Physics

    xml = HAND.replace('<joint name="knee" axis="0 1 0"/>', '<joint name="knee" axis="0 1 0" actuatorfrcrange="-3 2"/>') \
              .replace('<position joint="knee" kp="40" forcerange="-30 30"/>', '<position joint="knee" kp="40" forcerange="-30 30"/><velocity joint="knee" kv="0.5" ctrlrange="-10 10"/>'
                       '<general joint="hip" gainprm="2" biastype="affine" biasprm="0.25 -1 -0.1" ctrlrange="-3 3" gear="1.5"/>')
    assert xml != HAND
    spec = mjcf.parse_mjcf(xml)
    assert [b for b in spec.bodies if b.name == "shin"][0].joints[0].actuatorfrcrange == (-3.0, 2.0)
    cm = compile_model(spec)
    nv, t = cm.nv, cm.t
    assert t["dof_actfrcrange"][nv - 1].tolist() == [-3.0, 2.0] and t["dof_actfrcrange"][0, 1] == np.finfo(np.float32).max
    assert t["act_gain"].tolist() == [25.0, 40.0, 0.5, 2.0] and t["act_bias"][2].tolist() == [0.0, 0.0, -0.5] and t["act_bias"][3].tolist() == [0.25, -1.0, -0.1]
    # the writer keeps all of it
    assert compile_model(mjcf.parse_mjcf(mjcf.to_mjcf(spec))).to_blob(True) == cm.to_blob(True)
    ph = Physics(t, np.float64)
    q = np.asarray(t["qpos0"], np.float64)[None].copy()
    v = np.zeros((1, nv))
    q[0, -2], q[0, -1], v[0, -2], v[0, -1] = 0.1, -0.2, 0.3, -0.4      # hip, knee
    d = ph.pipeline_init(q, v)
    d["ctrl"] = np.array([[0.0, 0.0, 1.0, 0.5]])
    ph.fwd_actuation(d)
    hip_pos = 25.0 * (0.0 - 0.1)
    hip_gen = 2.0 * 0.5 + 0.25 - 1.0 * (1.5 * 0.1) - 0.1 * (1.5 * 0.3)      # length and velocity carry the gear
    assert d["qfrc_actuator"][0, -2] == pytest.approx(hip_pos + 1.5 * hip_gen)
    knee = 40.0 * (0.0 + 0.2) + 0.5 * (1.0 + 0.4)                          # position servo 8.0 (inside its forcerange) + velocity servo 0.7
    assert knee == pytest.approx(8.7) and d["qfrc_actuator"][0, -1] == 2.0   # ... clamped by the joint's actuatorfrcrange
    with pytest.raises(ValueError, match="actuatorfrclimited='true' without"):
        mjcf.parse_mjcf(HAND.replace('<joint name="knee" axis="0 1 0"/>', '<joint name="knee" axis="0 1 0" actuatorfrclimited="true"/>'))
    with pytest.raises(ValueError, match="only dyntype='none'"):
        mjcf.parse_mjcf(HAND.replace('<position joint="hip"/>', '<general joint="hip" dyntype="integrator"/>'))


def test_springref_is_the_springs_rest_position_not_ref():
    """MuJoCo: qpos0 = ref, qpos_spring = springref (both default 0) - the reference's own robot sets neither, an export may set both.  The
    spring term of the passive force (oracle passive(), the twin and the kernel read the same table) pulls towards springref."""
    from oracle.physics_oracle import Physics

    knee = '<joint name="knee" axis="0 1 0"/>'
    plain = compile_model(mjcf.parse_mjcf(HAND.replace(knee, '<joint name="knee" axis="0 1 0" stiffness="5" ref="-10"/>')))
    sprung = mjcf.parse_mjcf(HAND.replace(knee, '<joint name="knee" axis="0 1 0" stiffness="5" ref="-10" springref="-40"/>'))
    cm = compile_model(sprung)
    qa = cm.nq - 1
    assert cm.t["qpos0"][qa] == pytest.approx(math.radians(-10)) and plain.t["qpos0"][qa] == pytest.approx(math.radians(-10))
    assert cm.t["qpos_spring"][qa] == pytest.approx(math.radians(-40)) and plain.t["qpos_spring"][qa] == 0.0   # (springref's default is 0, not ref)
    np.testing.assert_array_equal(np.delete(cm.t["qpos_spring"], qa), np.delete(cm.t["qpos0"], qa))
    # the writer keeps it: text -> spec -> text -> spec compiles to the same bytes
    again = compile_model(mjcf.parse_mjcf(mjcf.to_mjcf(sprung)))
    assert again.to_blob(True) == cm.to_blob(True)
    ph = Physics(cm.t, np.float64)
    d = ph.pipeline_init(np.asarray(cm.t["qpos0"], np.float64)[None], np.zeros((1, cm.nv)))
    ph.passive(d)
    assert d["qfrc_passive"][0, cm.nv - 1] == pytest.approx(-5 * math.radians(30)) and not d["qfrc_passive"][0, :-1].any()
    # a programmatic spec without springref keeps its spring at ref (the built-in robots' tables, and the golden fixtures made from them)
    from minppo_amd.model import BUILTIN_MODELS
    for name in BUILTIN_MODELS:
        t = load_model(name).t
        np.testing.assert_array_equal(t["qpos_spring"], t["qpos0"])


@pytest.mark.parametrize("old,new,msg", [
    ('<actuator>', '<equality/><actuator>', "equality"),
    ('type="sphere" size="0.1"', 'type="ellipsoid" size="0.1 0.1 0.2"', "only sphere, capsule, cylinder and box geoms can collide"),
    ('<joint name="knee" axis="0 1 0"/>', '<joint name="knee" type="ball"/>', "joint type"),
    ('<position joint="hip"/>', '<position tendon="t"/>', "joint transmissions"),
    ('<position joint="hip"/>', '<position joint="nope"/>', "unknown joint"),
    ('<option timestep="0.001"', '<option cone="elliptic" timestep="0.001"', "elliptic"),
    ('<body name="shin" pos="0 0 -0.3">', '<body name="shin" pos="0 0 -0.3" mocap="true">', "mocap"),
])
def test_outside_the_subset_is_an_error_not_a_silent_drop(old, new, msg):
    assert old in HAND
    with pytest.raises(ValueError, match=msg):
        mjcf.parse_mjcf(HAND.replace(old, new))


def test_load_model_accepts_an_xml_path_and_the_oracle_can_step_it(tmp_path):
    from oracle.physics_oracle import Physics

    p = tmp_path / "hand.xml"
    p.write_text(HAND)
    cm = load_model(str(p))
    assert cm.name == "hand" and cm.nu == 2
    ph = Physics(cm.t, np.float64)
    q0 = np.tile(np.asarray(cm.t["qpos0"], np.float64), (3, 1))
    d = ph.pipeline_init(q0, np.zeros((3, cm.nv)))
    for _ in range(5):
        d = ph.pipeline_step(d, np.zeros((3, cm.nu)))
    assert np.isfinite(d["qpos"]).all() and np.isfinite(d["qvel"]).all()
    assert (d["qpos"][:, 2] < 0.75).all()  # it falls


def test_box_geom_collides_through_its_eight_corners(tmp_path):
    """box geoms.  A colliding <geom type="box"> meets the ground as MJX has it - plane_convex on the convex mesh of its eight
    corners (in the body frame, geom pose applied; mesh.box's vertex order), four contact slots; it survives the writer / parser round trip."""
    import numpy as np

    from minppo_amd import mjcf
    from minppo_amd.model import compile_model

    xml = """<mujoco model="brick"><option timestep="0.002"/><worldbody><geom type="plane" size="0 0 1"/>
      <body name="brick" pos="0 0 0.3"><freejoint name="root"/><inertial pos="0 0 0" mass="1.2" diaginertia="0.0013 0.0044 0.0055"/>
        <geom type="box" size="0.10 0.05 0.02" pos="0.01 0 0" friction="0.8 0.005 0.0001"/></body></worldbody></mujoco>"""
    p = tmp_path / "brick.xml"
    p.write_text(xml)
    spec = mjcf.load_mjcf(str(p))
    cm = compile_model(spec)
    assert cm.ncon == 4 and cm.nefc == 16 and (np.asarray(cm.t["con_radius"]) == 0).all() and cm.t["con_cvx"].tolist() == [0, 1, 2, 3]
    corners = np.asarray(cm.t["cvx_vert"]).reshape(8, 3)
    want = [(0.01 + sx * 0.10, sy * 0.05, sz * 0.02) for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]
    np.testing.assert_allclose(corners, want, atol=1e-12)
    p2 = tmp_path / "brick2.xml"
    p2.write_text(mjcf.to_mjcf(spec))
    cm2 = compile_model(mjcf.load_mjcf(str(p2)))
    assert cm2.to_blob() == cm.to_blob()


def test_collision_masks_generate_the_pairs_mujoco_would_test(tmp_path):
    """SURVEY 8(f1) body-body pairs.  With MuJoCo's default masks (contype = conaffinity = 1) every sphere / capsule pair on
    different bodies is a candidate except parent-child pairs and bodies welded together; contype / conaffinity remove pairs;
    geom 1 of a mixed pair is the sphere.  A box meets the round geoms of other bodies as a convex hull."""
    import numpy as np

    from minppo_amd import mjcf
    from minppo_amd.model import compile_model

    def xml(extra_a="", extra_d="", dtype="sphere", dsize="0.04"):
        return f"""<mujoco model="chain"><worldbody><geom type="plane" size="0 0 1"/>
      <body name="a" pos="0 0 1"><freejoint name="root"/><inertial pos="0 0 0" mass="1" diaginertia="0.01 0.01 0.01"/>
        <geom name="ga" type="capsule" size="0.05 0.1" {extra_a}/>
        <body name="b" pos="0.2 0 0"><joint name="j1" type="hinge" axis="0 1 0"/><inertial pos="0 0 0" mass="1" diaginertia="0.01 0.01 0.01"/>
          <geom name="gb" type="sphere" size="0.05"/>
          <body name="w" pos="0.1 0 0"><inertial pos="0 0 0" mass="0.1" diaginertia="0.001 0.001 0.001"/>
            <geom name="gw" type="sphere" size="0.03"/>
            <body name="c" pos="0.2 0 0"><joint name="j2" type="hinge" axis="0 1 0"/><inertial pos="0 0 0" mass="1" diaginertia="0.01 0.01 0.01"/>
              <geom name="gc" type="capsule" size="0.04 0.1"/></body></body></body>
        <body name="d" pos="-0.2 0 0"><joint name="j3" type="hinge" axis="0 1 0"/><inertial pos="0 0 0" mass="1" diaginertia="0.01 0.01 0.01"/>
          <geom name="gd" type="{dtype}" size="{dsize}" {extra_d}/></body>
      </body></worldbody></mujoco>"""

    def pairs(text):
        p = tmp_path / "m.xml"
        p.write_text(text)
        cm = compile_model(mjcf.load_mjcf(str(p)))
        return cm, [tuple(int(v) for v in r) for r in np.asarray(cm.t["pair_body"]).reshape(-1, 2)]

    # bodies: a=1, b=2, w=3 (welded to b: no joint), c=4, d=5.  Parent-child: a-b, a-d, and (w welded to b) b-c / w-c, a-w; same weld: b-w.
    cm, pb = pairs(xml())
    assert set(pb) == {(2, 5), (3, 5), (5, 4), (1, 4)}, pb
    assert cm.npair == 4 and cm.ncon == (2 + 1 + 1 + 2 + 1) + 4
    # (sphere, sphere) group first, then (sphere, capsule) with the sphere as geom 1, then (capsule, capsule)
    kinds = [(float(np.linalg.norm(g[3:6])) > 0, float(np.linalg.norm(g[11:14])) > 0) for g in np.asarray(cm.t["pair_geom"]).reshape(-1, 16)]
    assert kinds == sorted(kinds) and kinds[0] == (False, False) and kinds[-1] == (True, True) and (True, False) not in kinds
    # masks: d only affine to nothing and typed 2 -> it meets the plane (plane conaffinity 1 & ... no: contype 2 & plane conaffinity 1 = 0,
    # plane contype 1 & d conaffinity 0 = 0) neither the plane nor anybody
    cm2, pb2 = pairs(xml(extra_d='contype="2" conaffinity="0"'))
    assert set(pb2) == {(1, 4)} and cm2.ncon == (2 + 1 + 1 + 2) + 1
    # a box meets spheres and capsules of other bodies as a convex hull (round 5; tests/test_convex_pairs.py): one slot per sphere, two per
    # capsule, after the groups of the round geoms among themselves - (sphere, box) < (capsule, capsule) < (capsule, box) by MuJoCo's type ids
    cmb, pbb = pairs(xml(dtype="box", dsize="0.04 0.04 0.04"))
    assert pbb == [(2, 5), (3, 5), (1, 4), (4, 5), (4, 5)] and int(cmb.t["nhull"]) == 1 and cmb.ncon == (2 + 1 + 1 + 2 + 4) + 5
    assert np.asarray(cmb.t["pair_geom"])[:, 7].tolist() == [1, 1, 0, 1, 1] and np.asarray(cmb.t["pair_geom"])[:, 15].tolist() == [0, 0, 0, 0, 1]
    cm3, pb3 = pairs(xml(dtype="box", dsize="0.04 0.04 0.04", extra_d='contype="0" conaffinity="1"').replace('name="ga" type="capsule"', 'name="ga" contype="0" type="capsule"')
                     .replace('name="gb" type="sphere"', 'name="gb" contype="0" type="sphere"').replace('name="gw" type="sphere"', 'name="gw" contype="0" type="sphere"')
                     .replace('name="gc" type="capsule"', 'name="gc" contype="0" type="capsule"'))
    assert pb3 == [] and cm3.ncon == 2 + 1 + 1 + 2 + 4  # everything still meets the ground (plane contype 1 & conaffinity 1)


MESH_XML = """<mujoco model="meshy"><compiler angle="radian" meshdir="assets"/><option timestep="0.002"/>
  <asset>{asset}</asset>
  <worldbody><geom type="plane" size="0 0 1"/>
    <body name="rock" pos="0 0 0.3"><freejoint name="root"/><inertial pos="0 0 0" mass="1.0" diaginertia="0.002 0.003 0.004"/>
      <geom type="mesh" mesh="rock" pos="0.01 0 0" friction="0.7 0.005 0.0001" {gattr}/>{extra}</body></worldbody></mujoco>"""
CUBE_PLUS = "0 0 0  0.1 0 0  0 0.1 0  0.1 0.1 0  0 0 0.1  0.1 0 0.1  0 0.1 0.1  0.1 0.1 0.1  0.05 0.05 0.05  0.1 0 0"  # 8 corners, the centre, a duplicate


def test_mesh_geom_collides_as_its_convex_hull(tmp_path):
    """mesh geoms.  <asset><mesh vertex=...> (or an .obj / .stl file next to the MJCF) becomes the vertices of its convex
    hull in the file's order - interior points and duplicates dropped - under the geom's pose, in the body frame; the geom gets four
    contact slots against the ground (MJX plane_convex).  Written back by to_mjcf, it compiles to the same tables."""
    cm = compile_model(mjcf.parse_mjcf(MESH_XML.format(asset=f'<mesh name="rock" vertex="{CUBE_PLUS}" scale="2 1 1"/>', gattr='contype="0"', extra="")))
    assert (cm.ncvx, cm.ncon, cm.nefc) == (1, 4, 16)
    v = np.asarray(cm.t["cvx_vert"])
    want = np.array([[0, 0, 0], [0.2, 0, 0], [0, 0.1, 0], [0.2, 0.1, 0], [0, 0, 0.1], [0.2, 0, 0.1], [0, 0.1, 0.1], [0.2, 0.1, 0.1]]) + [0.01, 0, 0]
    np.testing.assert_allclose(v, want, atol=1e-12)  # scale applied, centre + duplicate gone, file order kept, geom pos added
    assert list(cm.t["con_cvx"]) == [0, 1, 2, 3] and list(cm.t["cvx_vadr"]) == [0, 8]
    np.testing.assert_allclose(np.asarray(cm.t["con_friction"])[:, 0], 1.0)  # max(geom 0.7, plane 1.0), like every other ground contact
    # files: Wavefront OBJ and binary STL with the same corner set
    (tmp_path / "assets").mkdir()
    pts = np.array([float(x) for x in CUBE_PLUS.split()]).reshape(-1, 3)
    (tmp_path / "assets" / "rock.obj").write_text("# cube\n" + "".join(f"v {x} {y} {z}\n" for x, y, z in pts) + "f 1 2 3\n")
    tri = np.zeros(4, np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]))
    tri["v"] = pts[[0, 1, 2, 3, 4, 5, 6, 7, 0, 3, 5, 6]].reshape(4, 3, 3)
    (tmp_path / "assets" / "rock.stl").write_bytes(b"\0" * 80 + np.uint32(4).tobytes() + tri.tobytes())
    for fname in ("rock.obj", "rock.stl"):
        p = tmp_path / "m.xml"
        p.write_text(MESH_XML.format(asset=f'<mesh file="{fname}"/>', gattr='contype="0"', extra=""))
        cmf = compile_model(mjcf.load_mjcf(str(p)))
        assert cmf.ncvx == 1 and {tuple(np.round(r, 6)) for r in np.asarray(cmf.t["cvx_vert"])} == {tuple(np.round(r + [0.01, 0, 0], 6)) for r in pts[:8]}  # (STL keeps float32)
    # round trip through the writer
    spec = mjcf.parse_mjcf(MESH_XML.format(asset=f'<mesh name="rock" vertex="{CUBE_PLUS}"/>', gattr='contype="0"', extra=""))
    cm1, cm2 = compile_model(spec), compile_model(mjcf.parse_mjcf(mjcf.to_mjcf(spec)))
    for k in ("cvx_vert", "cvx_vadr", "con_cvx", "cvx_body", "con_friction"):
        np.testing.assert_allclose(cm1.t[k], cm2.t[k], atol=1e-12, err_msg=k)


def test_mesh_defaults_and_maxhullvert(tmp_path):
    """What a real export's collision meshes need: `<default><mesh scale=...>` (millimetre STL files are scaled there) and
    `maxhullvert` - MuJoCo's cap on a mesh's hull, the setting its documentation recommends for MJX: a mesh whose hull has hundreds of
    vertices is refused without it (the kernel scans 64 at most) and loads with it, as the hull qhull holds after maxhullvert - 4 added points."""
    rng = np.random.default_rng(3)
    pts = rng.normal(size=(400, 3))
    pts = np.round(0.05 * pts / np.linalg.norm(pts, axis=1, keepdims=True) * [1.0, 0.7, 0.5], 7)  # an ellipsoid's surface: every point is a hull vertex
    vtx = " ".join(f"{x:.7f}" for x in pts.ravel())
    with pytest.raises(ValueError, match="maxhullvert"):
        compile_model(mjcf.parse_mjcf(MESH_XML.format(asset=f'<mesh name="rock" vertex="{vtx}"/>', gattr='contype="0"', extra="")))
    cm = compile_model(mjcf.parse_mjcf(MESH_XML.format(asset=f'<mesh name="rock" vertex="{vtx}" maxhullvert="40"/>', gattr='contype="0"', extra="")))
    v = np.asarray(cm.t["cvx_vert"]) - [0.01, 0, 0]
    assert cm.ncvx == 1 and len(v) == 40
    def given(r):
        return np.abs(pts - r).max(axis=1).min() < 1e-9

    assert all(given(r) for r in v)  # vertices of the mesh, not new points
    assert np.ptp(v, axis=0).min() > 0.03  # the capped hull still spans the ellipsoid (qhull adds the furthest points first)
    with pytest.raises(ValueError, match="larger than 3"):
        mjcf.parse_mjcf(MESH_XML.format(asset=f'<mesh name="rock" vertex="{vtx}" maxhullvert="3"/>', gattr='contype="0"', extra=""))
    # the same through <default><mesh>, with the scale there too (class defaults included)
    xml = MESH_XML.format(asset=f'<mesh name="rock" vertex="{vtx}"/>', gattr='contype="0"', extra="").replace(
        "<asset>", '<default><mesh scale="2 2 2" maxhullvert="24"/><default class="big"><mesh maxhullvert="48"/></default></default><asset>')
    cm2 = compile_model(mjcf.parse_mjcf(xml))
    v2 = np.asarray(cm2.t["cvx_vert"]) - [0.01, 0, 0]
    assert len(v2) == 24 and all(given(r / 2) for r in v2)
    cm3 = compile_model(mjcf.parse_mjcf(xml.replace('<mesh name="rock"', '<mesh class="big" name="rock"')))
    assert len(np.asarray(cm3.t["cvx_vert"])) == 48
    assert cm.ncon == 4  # (four ground slots, like every hull)


@pytest.mark.parametrize("asset,gattr,extra,msg", [
    ('<mesh name="other" vertex="0 0 0 1 0 0 0 1 0 0 0 1"/>', 'contype="0"', "", "is not defined under <asset>"),
    ('<mesh name="rock" vertex="0 0 0 1 0 0 0 1 0"/>', 'contype="0"', "", "at least four vertices"),
    ('<mesh name="rock" vertex="0 0 0 1 0 0 0 1 0 1 1 0 0.5 0.5 0"/>', 'contype="0"', "", "degenerate mesh"),
    ('<mesh name="rock" vertex="' + " ".join(f"{np.cos(a):.6f} {np.sin(a):.6f} {0.3 * np.cos(5 * a):.6f}" for a in np.linspace(0, 6.2, 80)) + '"/>', 'contype="0"', "", "decimate the collision mesh"),
])
def test_mesh_geoms_outside_the_subset_are_loud_errors(asset, gattr, extra, msg):
    with pytest.raises(ValueError, match=msg):
        compile_model(mjcf.parse_mjcf(MESH_XML.format(asset=asset, gattr=gattr, extra=extra)))


def test_a_mesh_against_a_box_of_another_body_is_a_pair_of_four_slots():
    """Round 6 (it was a loud error until round 5): a mesh hull and a box on bodies that are neither welded nor parent and child collide as MJX's
    convex_convex pairs them - four contact slots, both hulls in the blob's hull section."""
    extra = ('<body name="b2" pos="0 0 0.2"><joint name="j" axis="0 1 0"/><inertial pos="0 0 0" mass="1" diaginertia="1 1 1"/>'
             '<body name="b3" pos="0 0 0.2"><joint name="j3" axis="0 1 0"/><inertial pos="0 0 0" mass="1" diaginertia="1 1 1"/><geom type="box" size="0.05 0.05 0.05"/></body></body>')
    cm = compile_model(mjcf.parse_mjcf(MESH_XML.format(asset=f'<mesh name="rock" vertex="{CUBE_PLUS}"/>', gattr="", extra=extra)))
    assert cm.npair == 4 and int(cm.t["nhull"]) == 2 and cm.t["pair_geom"][:, 15].tolist() == [0, 1, 2, 3]
    assert (cm.t["pair_geom"][:, 7] > 0).all() and (cm.t["pair_geom"][:, 14] > 0).all() and (cm.t["pair_geom"][:, 7] != cm.t["pair_geom"][:, 14]).all()


EXPORT = Path(__file__).parent / "golden" / "export_biped"


def test_export_style_layout_includes_defaults_meshdir(caplog):
    """The way an onshape / URDF export lays a robot out (tests/golden/export_biped, written by tests/golden/make_export_biped.py): the model
    split over <include> files - resolved against the MAIN file's directory, some inside a <body>, some in sub-directories - repeated
    top-level sections that merge, nested default classes + childclass, meshdir with an .obj collision mesh and a binary .stl visual mesh."""
    import logging

    with caplog.at_level(logging.WARNING, logger="minppo_amd.mjcf"):
        s = mjcf.load_mjcf(str(EXPORT / "robot.xml"))
    names = [b.name for b in s.bodies]
    assert len(names) == 28 and names[0] == "pelvis" and {"l_toe", "r_toe", "l_hand", "r_hand", "head", "torso"} <= set(names)
    by = {b.name: b for b in s.bodies}
    assert by["l_hip_yaw"].parent == "pelvis" and by["l_shoulder"].parent == "torso" and by["r_toe"].parent == "r_foot"  # includes spliced in place
    # nested defaults: leg -> knee / ankle; childclass="leg" on the hip reaches the toe's joint; class on an element beats childclass
    knee, ankle, toe, waist = by["l_shin"].joints[0], by["l_ankle"].joints[0], by["l_toe"].joints[0], by["torso"].joints[0]
    assert (knee.damping, knee.armature, knee.range) == (1.2, 0.03, (0.0, 2.2))
    assert (ankle.damping, ankle.armature, ankle.range) == (0.3, 0.01, (-0.6, 0.6))
    assert (toe.damping, toe.armature, toe.stiffness) == (0.05, 0.03, 2.0)
    assert (waist.damping, waist.armature) == (2.0, 0.05)
    assert by["l_forearm"].joints[0].damping == 0.2 and by["l_hand"].joints[0].range == (-0.7, 0.7)
    # both <actuator> sections arrived, in file order, with the <default><position> attributes
    assert [a.joint for a in s.actuators][:3] == ["l_hip_yaw", "l_hip_roll", "l_hip_pitch"] and len(s.actuators) == 20
    assert s.actuators[0].ctrlrange == (-1.0, 1.0) and s.actuators[0].forcerange == (-80.0, 80.0) and s.actuators[-1].kp == 4.0
    # colliders: two mesh feet (hull of foot.obj: the two interior points are gone), toes, hands, shins; the visual geoms collide with nothing
    feet = [g for n in ("l_foot", "r_foot") for g in by[n].geoms]
    assert len(feet) == 2 and all(g.type == mjcf.GEOM_MESH and len(g.vertices) == 12 for g in feet)
    assert not by["pelvis"].geoms and not by["l_thigh"].geoms and len(by["l_shin"].geoms) == 1 and by["l_shin"].geoms[0].contype == 2
    # dry joint friction: silently gone from the defaults, dropped with a warning from the two joints that carry it themselves
    msgs = [r.getMessage() for r in caplog.records if "frictionloss" in r.getMessage()]
    assert len(msgs) == 2 and any("l_shin" in m for m in msgs) and any("r_forearm" in m for m in msgs)
    # fullinertia -> principal moments (descending) + a rotation
    w = np.linalg.eigvalsh(np.array([[0.045, 0.0008, -0.0004], [0.0008, 0.038, 0.0011], [-0.0004, 0.0011, 0.052]]))[::-1]
    np.testing.assert_allclose(by["pelvis"].inertia, w, rtol=1e-12)
    cm = compile_model(s)
    # pairs: the two shins; each shin capsule against the OTHER leg's foot mesh (capsule_convex: two slots each).  A leg's own shin / foot pair
    # is a candidate by the masks too (the ankle body sits between them) and is taken out by contacts.xml's <exclude>; toes / hands: ground only
    assert s.contact_excludes == [("l_shin", "l_foot"), ("r_shin", "r_foot")]
    assert (cm.nq, cm.nv, cm.nu) == (34, 33, 20) and int(cm.t["npair"]) == 5 and int(cm.t["nhull"]) == 2
    assert cm.t["pair_body"].tolist() == [[5, 12], [5, 14], [5, 14], [12, 7], [12, 7]]
    s.contact_excludes = []
    assert int(compile_model(s).t["npair"]) == 9
    # the writer keeps the excludes; anything but <exclude> and <pair> under <contact> is an error
    s.contact_excludes = [("l_shin", "l_foot"), ("r_shin", "r_foot")]
    assert compile_model(mjcf.parse_mjcf(mjcf.to_mjcf(s))).t["pair_body"].tolist() == cm.t["pair_body"].tolist()
    with pytest.raises(ValueError, match="only <exclude"):
        mjcf.parse_mjcf(HAND.replace("<actuator>", '<contact><tendonpair a="b"/></contact><actuator>'))
    with pytest.raises(ValueError, match="unknown geom"):  # (explicit pairs: test_explicit_contact_pairs)
        compile_model(mjcf.parse_mjcf(HAND.replace("<actuator>", '<contact><pair geom1="a" geom2="b"/></contact><actuator>')))
    with pytest.raises(ValueError, match="unknown body"):
        mjcf.parse_mjcf(HAND.replace("<actuator>", '<contact><exclude body1="base" body2="nobody"/></contact><actuator>'))


def test_include_errors_are_loud(tmp_path):
    main = tmp_path / "m.xml"
    main.write_text('<mujoco><include file="a.xml"/><worldbody><body name="b"><freejoint/><inertial pos="0 0 0" mass="1" diaginertia="1 1 1"/></body></worldbody></mujoco>')
    with pytest.raises(ValueError, match="not found"):
        mjcf.load_mjcf(str(main))
    (tmp_path / "a.xml").write_text('<mujocoinclude><include file="b.xml"/></mujocoinclude>')
    (tmp_path / "b.xml").write_text('<mujocoinclude><include file="a.xml"/></mujocoinclude>')
    with pytest.raises(ValueError, match="more than once"):
        mjcf.load_mjcf(str(main))
    (tmp_path / "a.xml").write_text('<robot/>')
    with pytest.raises(ValueError, match="mujocoinclude"):
        mjcf.load_mjcf(str(main))
    with pytest.raises(ValueError, match="needs the MJCF's directory"):
        mjcf.parse_mjcf(main.read_text())
    (tmp_path / "a.xml").write_text('<mujocoinclude><option timestep="0.004"/><compiler angle="radian"/></mujocoinclude>')
    assert mjcf.load_mjcf(str(main)).timestep == 0.004  # a section brought in by an include counts like one written in place


def test_compiler_mass_post_processing_and_fusestatic():
    """<compiler boundmass boundinertia settotalmass balanceinertia> change masses and inertias (MuJoCo's compiler applies them in this order:
    balance, bounds, total mass); fusestatic='true' would renumber the bodies - an error when the model has a jointless body, nothing otherwise."""
    base = mjcf.parse_mjcf(HAND)
    total = sum(b.mass for b in base.bodies)
    s = mjcf.parse_mjcf(HAND.replace('<compiler angle="degree" eulerseq="xyz"/>', '<compiler angle="degree" eulerseq="xyz" settotalmass="10" fusestatic="true"/>'))
    assert sum(b.mass for b in s.bodies) == pytest.approx(10.0)
    for b0, b1 in zip(base.bodies, s.bodies):
        assert b1.mass == pytest.approx(b0.mass * 10.0 / total) and b1.inertia == pytest.approx(tuple(x * 10.0 / total for x in b0.inertia))
    s = mjcf.parse_mjcf(HAND.replace('<compiler angle="degree" eulerseq="xyz"/>', '<compiler angle="degree" eulerseq="xyz" boundmass="1.0" boundinertia="0.025"/>'))
    assert [b.mass for b in s.bodies] == pytest.approx([max(b.mass, 1.0) for b in base.bodies]) and min(min(b.inertia) for b in s.bodies) == pytest.approx(0.025)
    bad = '<inertial pos="0 0 -0.15" mass="2.0" fullinertia="0.02 0.03 0.01 0.001 0 0"/>'
    assert bad in HAND
    lop = HAND.replace(bad, '<inertial pos="0 0 -0.15" mass="2.0" diaginertia="0.05 0.01 0.01"/>')
    assert mjcf.parse_mjcf(lop).bodies[1].inertia == (0.05, 0.01, 0.01)
    s = mjcf.parse_mjcf(lop.replace('<compiler angle="degree" eulerseq="xyz"/>', '<compiler angle="degree" eulerseq="xyz" balanceinertia="true"/>'))
    assert s.bodies[1].inertia == pytest.approx((0.07 / 3,) * 3)
    with pytest.raises(ValueError, match="fusestatic"):
        mjcf.parse_mjcf(HAND.replace('<compiler angle="degree" eulerseq="xyz"/>', '<compiler angle="degree" eulerseq="xyz" fusestatic="true"/>')
                        .replace('<body name="shin" pos="0 0 -0.3">', '<body name="bracket" pos="0 0 -0.1"><inertial pos="0 0 0" mass="0.1" diaginertia="1e-4 1e-4 1e-4"/></body><body name="shin" pos="0 0 -0.3">'))


def test_statistic_meaninertia_overrides_the_derived_value():
    """<statistic meaninertia> (it scales the solver's tolerance: MJX solver.solve) is honoured and written back; the other <statistic> fields are visual."""
    derived = compile_model(mjcf.parse_mjcf(HAND))
    s = mjcf.parse_mjcf(HAND.replace("<default>", '<statistic meaninertia="0.37" extent="2"/><default>', 1))
    cm = compile_model(s)
    assert float(cm.t["meaninertia"]) == 0.37 != float(derived.t["meaninertia"])
    assert float(compile_model(mjcf.parse_mjcf(mjcf.to_mjcf(s))).t["meaninertia"]) == 0.37


PAIR_XML = """<mujoco model="pairs"><compiler angle="radian"/><option timestep="0.002"/>
  <default><geom solref="0.02 1" solimp="0.9 0.95 0.001 0.5 2"/></default>
  <worldbody><geom name="floor" type="plane" size="0 0 1" friction="0.6 0.005 0.0001"/>
    <body name="a" pos="0 0 0.5"><freejoint name="ra"/><inertial pos="0 0 0" mass="1" diaginertia="0.01 0.01 0.01"/>
      <geom name="ga" type="sphere" size="0.1" contype="0" conaffinity="0" friction="0.3 0.005 0.0001"/>
      <body name="a2" pos="0 0 0.3"><joint name="ja" type="hinge" axis="0 1 0"/><inertial pos="0 0 0" mass="0.5" diaginertia="0.004 0.004 0.004"/>
        <geom name="ga2" type="capsule" size="0.05 0.1" contype="0" conaffinity="0"/></body></body>
    <body name="b" pos="0.5 0 0.5"><freejoint name="rb"/><inertial pos="0 0 0" mass="1" diaginertia="0.01 0.01 0.01"/>
      <geom name="gb" type="sphere" size="0.12" contype="0" conaffinity="0" friction="0.4 0.005 0.0001"/></body></worldbody>
  <contact>{contact}</contact></mujoco>"""


def test_explicit_contact_pairs():
    """SURVEY 8 f1, `<contact><pair>`: geom pairs that collide whatever masks, kinship or excludes say, with a sliding friction of their own
    (MuJoCo does not use the geoms' parameters for an explicit pair; its pair default is 1); one of the two may be the ground plane."""
    none = compile_model(mjcf.parse_mjcf(PAIR_XML.format(contact="")))
    assert none.ncon == 0  # every mask is zero: nothing collides
    cm = compile_model(mjcf.parse_mjcf(PAIR_XML.format(contact='<pair geom1="ga" geom2="gb" friction="0.8 0.8 0.005 0.0001 0.0001"/><pair geom1="floor" geom2="gb"/>'
                                                               '<pair geom1="ga2" geom2="ga"/><exclude body1="a" body2="b"/>')))
    # gb on the ground (one slot, the pair's default friction 1 - not max(0.4, 0.6)); ga-gb in spite of the exclude (0.8); ga2-ga in spite of parent and child (1.0)
    assert (cm.ncon, int(cm.t["npair"])) == (3, 2)
    fr = np.asarray(cm.t["con_friction"])[:, 0]
    np.testing.assert_allclose(sorted(fr), [0.8, 1.0, 1.0])
    np.testing.assert_allclose(fr[0], 1.0)  # (ground slots come first)
    pb = np.asarray(cm.t["pair_body"]).reshape(-1, 2)
    names = ["world", "a", "a2", "b"]
    assert {frozenset((names[i], names[j])) for i, j in pb} == {frozenset(("a", "b")), frozenset(("a", "a2"))}
    # a pair that the masks generate anyway is there once, with the pair's friction
    xml = PAIR_XML.format(contact='<pair geom1="gb" geom2="ga" friction="0.25 0.25"/>')
    xml = xml.replace('name="ga" type="sphere" size="0.1" contype="0" conaffinity="0"', 'name="ga" type="sphere" size="0.1" contype="2" conaffinity="0"')
    xml = xml.replace('name="gb" type="sphere" size="0.12" contype="0" conaffinity="0"', 'name="gb" type="sphere" size="0.12" contype="0" conaffinity="2"')
    cm2 = compile_model(mjcf.parse_mjcf(xml))
    assert int(cm2.t["npair"]) == 1 and cm2.ncon == 1
    np.testing.assert_allclose(np.asarray(cm2.t["con_friction"])[0, 0], 0.25)
    # round trip through the writer
    spec = mjcf.parse_mjcf(PAIR_XML.format(contact='<pair geom1="ga" geom2="gb" friction="0.8 0.8"/><pair geom1="floor" geom2="gb"/>'))
    cm3, cm4 = compile_model(spec), compile_model(mjcf.parse_mjcf(mjcf.to_mjcf(spec)))
    for k in ("con_friction", "pair_body", "pair_geom", "con_bodyid"):
        np.testing.assert_allclose(cm3.t[k], cm4.t[k], atol=1e-12, err_msg=k)


@pytest.mark.parametrize("contact,msg", [
    ('<pair geom1="ga" geom2="nope"/>', "unknown geom"),
    ('<pair geom1="ga" geom2="gb" condim="4"/>', "condim 4"),
    ('<pair geom1="ga" geom2="gb" margin="0.01"/>', "margin / gap"),
    ('<pair geom1="ga" geom2="gb" friction="0.5 0.7"/>', "share one coefficient"),
    ('<pair geom1="ga" geom2="gb" solref="0.01 1"/>', "one contact solref"),
    ('<pair geom1="floor" geom2="floor"/>', "two different geoms"),
    ('<pair geom1="ga" geom2="ga"/>', "two different geoms"),
    ('<pair geom1="ga"/>', "needs geom1 and geom2"),
])
def test_explicit_contact_pairs_outside_the_subset_are_loud_errors(contact, msg):
    with pytest.raises(ValueError, match=msg):
        compile_model(mjcf.parse_mjcf(PAIR_XML.format(contact=contact)))


def test_kernel_follows_the_oracle_on_explicit_pairs(be):
    """The scene of test_explicit_contact_pairs through the environment kernel (emulator / MI355X) against the float64 oracle: two spheres pressed together
    by an explicit pair while every mask is zero, one of them on the ground through its pair with the plane."""
    from test_kernels_physics import _probe
    from test_model_fuzz import Physics, PhysState, f32, f64

    cm = compile_model(mjcf.parse_mjcf(PAIR_XML.format(contact='<pair geom1="ga" geom2="gb" friction="0.8 0.8"/><pair geom1="floor" geom2="gb"/><pair geom1="ga2" geom2="ga"/>')))
    h, dims, _keep = be.model(cm)
    N = 6
    rng = np.random.default_rng(5)
    qpos = np.tile(np.asarray(cm.t["qpos0"], f64), (N, 1))
    qpos[:, 0:3] = [0.0, 0.0, 0.14]    # body a (free): sphere of 0.1
    qpos[:, 8:11] = [0.2, 0.0, 0.10]   # body b (free; a: 7 + hinge 1): sphere of 0.12, 0.02 into the ground, 0.02 into ga
    qpos[:, 0:3] += 0.004 * rng.standard_normal((N, 3))
    qpos[:, 7] = 0.2 * rng.standard_normal(N)
    qvel = 0.1 * rng.standard_normal((N, cm.nv))
    q32 = [x.astype(f32) for x in (qpos, qvel, np.zeros((N, 1)), np.zeros((N, cm.nv)))]
    d = PhysState(qpos=q32[0].astype(f64), qvel=q32[1].astype(f64), ctrl=np.zeros((N, 0)), qacc_warmstart=np.zeros((N, cm.nv)), time=np.zeros(N))
    Physics(cm.t, f64).forward(d)
    got = _probe(be, h, cm, *q32)
    act = d.efc_D > 0
    assert act[:, :4].all() and act[:, 4:8].all() and not act[:, 8:12].any()  # ground slot of gb, the ga-gb pair; the capsule stays clear of its parent's sphere
    assert ((got["efc_D"].reshape(N, -1) > 0) == act).all()
    for k, tol in dict(efc_D=1e-3, efc_aref=1e-3, efc_J=5e-4, qacc=0.3).items():  # (qacc: six CG iterations of an unconverged solver - the envelope of tests/test_kernels_physics.py)
        r, g = np.asarray(d[k]), got[k].reshape(np.asarray(d[k]).shape)
        assert np.abs(g - r).max() <= tol * (np.abs(r).max() + 1e-6), (k, np.abs(g - r).max() / (np.abs(r).max() + 1e-6))
    # the pair's friction is the pyramid's: rows of the ga-gb contact carry mu = 0.8 (the geoms' own 0.3 / 0.4 are not used)
    np.testing.assert_allclose(np.asarray(cm.t["con_friction"])[:, 0], [1.0, 0.8, 1.0])
    be.lib.model_close(h)
