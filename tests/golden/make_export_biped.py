"""Writes tests/golden/export_biped/: a 28-body biped laid out the way an onshape-to-robot / URDF export is.

Authored for this repository's tests (NOT a reference file; the reference ships no robot at all - `env.py:27-50` downloads one).  What it
exercises in minppo_amd/mjcf.py: the model split over `<include file=...>` files (resolved against the main file's directory, nested in
sub-directories, one of them inside a `<body>`), repeated top-level sections that must merge (`<actuator>`, `<asset>`, `<default>`),
nested `<default class>` trees with `childclass`, `meshdir` (an .obj collision mesh and a binary .stl visual mesh in millimetres),
per-body `<inertial>` (some as `fullinertia`), dry joint friction on the defaults AND on two joints (stripped, env.py:41-45), collision
filtering by contype / conaffinity (toes and hands meet the ground only; the two shins meet each other; a FOOT MESH meets the ground and
the shin capsule of the other leg - MJX capsule_convex, two slots per pair) and a `<contact><exclude>` file that takes out each leg's own
shin / foot pair, whose shapes overlap at the ankle.

    python tests/golden/make_export_biped.py
"""
import struct
from pathlib import Path

import numpy as np

OUT = Path(__file__).resolve().parent / "export_biped"

ROBOT = """<!-- authored for this repository's tests (not a reference file): see tests/golden/make_export_biped.py -->
<mujoco model="export_biped">
  <compiler angle="radian" meshdir="meshes" autolimits="true" inertiafromgeom="false"/>
  <option timestep="0.002" gravity="0 0 -9.81" iterations="50" solver="Newton"/>
  <include file="defaults.xml"/>
  <include file="assets.xml"/>
  <worldbody>
    <light name="sun" pos="0 0 3"/>
    <geom name="floor" type="plane" size="10 10 0.1" friction="1.0 0.005 0.0001"/>
    <body name="pelvis" pos="0 0 0.92" childclass="robot">
      <freejoint name="root"/>
      <inertial pos="0 0 0.02" mass="6.5" fullinertia="0.045 0.038 0.052 0.0008 -0.0004 0.0011"/>
      <geom type="mesh" mesh="pelvis_visual" class="visual"/>
      <include file="parts/leg_left.xml"/>
      <include file="parts/leg_right.xml"/>
      <body name="torso" pos="0 0 0.12">
        <inertial pos="0 0 0.14" mass="9.0" diaginertia="0.16 0.14 0.07"/>
        <joint name="waist_yaw" axis="0 0 1" range="-0.8 0.8" class="trunk"/>
        <geom type="capsule" size="0.09" fromto="0 -0.08 0.16 0 0.08 0.16" class="visual"/>
        <include file="parts/upper_body.xml"/>
      </body>
    </body>
  </worldbody>
  <include file="contacts.xml"/>
  <include file="actuators_legs.xml"/>
  <include file="actuators_arms.xml"/>
</mujoco>
"""

DEFAULTS = """<mujocoinclude>
  <default>
    <joint damping="0.5" armature="0.01" frictionloss="0.1"/>
    <geom friction="0.8 0.02 0.001" solref="0.02 1" solimp="0.9 0.95 0.001"/>
    <position kp="40" ctrlrange="-1 1"/>
    <default class="robot">
      <default class="visual"><geom contype="0" conaffinity="0" group="2" density="0"/></default>
      <default class="collision"><geom contype="1" conaffinity="0" group="3"/></default>
      <default class="trunk"><joint damping="2.0" armature="0.05"/></default>
      <default class="leg">
        <joint damping="1.0" armature="0.03"/>
        <default class="knee"><joint range="0 2.2" damping="1.2"/></default>
        <default class="ankle"><joint range="-0.6 0.6" damping="0.3" armature="0.01"/></default>
        <default class="shin_collision"><geom contype="2" conaffinity="2" group="3"/></default>
        <default class="foot_collision"><geom contype="1" conaffinity="2" group="3"/></default>
      </default>
      <default class="arm">
        <joint damping="0.2" range="-1.5 1.5"/>
        <default class="wrist"><joint damping="0.05" range="-0.7 0.7"/></default>
      </default>
    </default>
  </default>
</mujocoinclude>
"""

ASSETS = """<mujocoinclude>
  <asset>
    <mesh name="foot" file="foot.obj"/>
    <mesh name="pelvis_visual" file="pelvis_visual.stl" scale="0.001 0.001 0.001"/>
    <texture name="grid" type="2d" builtin="checker" width="8" height="8"/>
    <material name="grid" texture="grid"/>
  </asset>
</mujocoinclude>
"""


def leg(side: str, sy: int) -> str:
    knee_fl = ' frictionloss="0.3"' if side == "l" else ""
    return f"""<mujocoinclude>
  <body name="{side}_hip_yaw" pos="0 {sy * 0.09} -0.04" childclass="leg">
    <inertial pos="0 0 -0.02" mass="0.9" diaginertia="0.0012 0.0012 0.0009"/>
    <joint name="{side}_hip_yaw" axis="0 0 1" range="-0.5 0.5"/>
    <body name="{side}_hip_roll" pos="0 0 -0.05">
      <inertial pos="0 0 -0.01" mass="0.8" diaginertia="0.001 0.001 0.0008"/>
      <joint name="{side}_hip_roll" axis="1 0 0" range="-0.4 0.4"/>
      <body name="{side}_thigh" pos="0 0 -0.03">
        <inertial pos="0 0 -0.17" mass="3.2" fullinertia="0.045 0.046 0.006 0.0002 {sy * 0.0003} -0.0005"/>
        <joint name="{side}_hip_pitch" axis="0 1 0" range="-1.6 0.8"/>
        <geom type="capsule" size="0.055" fromto="0 0 -0.03 0 0 -0.33" class="visual"/>
        <body name="{side}_shin" pos="0 0 -0.37">
          <inertial pos="0 0 -0.16" mass="2.1" diaginertia="0.026 0.026 0.002"/>
          <joint name="{side}_knee" axis="0 1 0" class="knee"{knee_fl}/>
          <geom type="capsule" size="0.04" fromto="0 0 -0.02 0 0 -0.32" class="shin_collision"/>
          <body name="{side}_ankle" pos="0 0 -0.36">
            <inertial pos="0 0 0" mass="0.25" diaginertia="0.0002 0.0002 0.0002"/>
            <joint name="{side}_ankle_pitch" axis="0 1 0" class="ankle"/>
            <body name="{side}_foot" pos="0 0 -0.03">
              <inertial pos="0.03 0 -0.02" mass="0.7" diaginertia="0.0008 0.0024 0.0028"/>
              <joint name="{side}_ankle_roll" axis="1 0 0" class="ankle" range="-0.35 0.35"/>
              <geom type="mesh" mesh="foot" pos="0.02 0 -0.03" class="foot_collision"/>
              <body name="{side}_toe" pos="0.12 0 -0.035">
                <inertial pos="0.02 0 0" mass="0.12" diaginertia="0.00004 0.00008 0.0001"/>
                <joint name="{side}_toe" axis="0 1 0" range="-0.3 0.6" damping="0.05" stiffness="2.0"/>
                <geom type="capsule" size="0.018" fromto="0 -0.035 0 0 0.035 0" class="collision"/>
              </body>
            </body>
          </body>
        </body>
      </body>
    </body>
  </body>
</mujocoinclude>
"""


def arm(side: str, sy: int) -> str:
    elbow_fl = ' frictionloss="0.05"' if side == "r" else ""
    lo, hi = (-0.3, 1.6) if sy > 0 else (-1.6, 0.3)
    return f"""  <body name="{side}_shoulder" pos="0 {sy * 0.19} 0.27" childclass="arm">
    <inertial pos="0 0 0" mass="0.6" diaginertia="0.0006 0.0006 0.0005"/>
    <joint name="{side}_shoulder_pitch" axis="0 1 0" range="-2.5 1.2"/>
    <body name="{side}_upper_arm_roll" pos="0 {sy * 0.03} 0">
      <inertial pos="0 0 -0.01" mass="0.5" diaginertia="0.0004 0.0004 0.0004"/>
      <joint name="{side}_shoulder_roll" axis="1 0 0" range="{lo} {hi}"/>
      <body name="{side}_upper_arm" pos="0 0 -0.03">
        <inertial pos="0 0 -0.11" mass="1.3" diaginertia="0.006 0.006 0.0009"/>
        <joint name="{side}_shoulder_yaw" axis="0 0 1"/>
        <geom type="capsule" size="0.035" fromto="0 0 0 0 0 -0.22" class="visual"/>
        <body name="{side}_forearm" pos="0 0 -0.25">
          <inertial pos="0 0 -0.09" mass="0.8" diaginertia="0.003 0.003 0.0004"/>
          <joint name="{side}_elbow" axis="0 1 0" range="-2.2 0"{elbow_fl}/>
          <geom type="capsule" size="0.03" fromto="0 0 0 0 0 -0.2" class="visual"/>
          <body name="{side}_hand" pos="0 0 -0.23">
            <inertial pos="0 0 -0.03" mass="0.35" diaginertia="0.0003 0.0003 0.0002"/>
            <joint name="{side}_wrist" axis="1 0 0" class="wrist"/>
            <geom type="sphere" size="0.04" pos="0 0 -0.03" class="collision"/>
          </body>
        </body>
      </body>
    </body>
  </body>
"""


HEAD = """  <body name="neck" pos="0 0 0.33">
    <inertial pos="0 0 0.02" mass="0.4" diaginertia="0.0003 0.0003 0.0002"/>
    <joint name="neck_yaw" axis="0 0 1" range="-1.2 1.2" class="trunk" damping="0.2" armature="0.005"/>
    <body name="head" pos="0 0 0.05">
      <inertial pos="0.01 0 0.06" mass="1.6" diaginertia="0.006 0.006 0.005"/>
      <joint name="neck_pitch" axis="0 1 0" range="-0.5 0.6" class="trunk" damping="0.2" armature="0.005"/>
      <geom type="sphere" size="0.09" pos="0.01 0 0.07" class="visual"/>
      <camera name="eyes" pos="0.1 0 0.08"/>
    </body>
  </body>
"""


def actuators(joints) -> str:
    rows = []
    for side in "lr":
        for j, kp, fr in joints:
            rows.append(f'    <position name="{side}_{j}" joint="{side}_{j}" kp="{kp}"' + (f' forcerange="-{fr} {fr}"' if fr else "") + "/>")
    return "<mujocoinclude>\n  <actuator>\n" + "\n".join(rows) + "\n  </actuator>\n</mujocoinclude>\n"


def main() -> None:
    (OUT / "parts").mkdir(parents=True, exist_ok=True)
    (OUT / "meshes").mkdir(parents=True, exist_ok=True)
    (OUT / "robot.xml").write_text(ROBOT)
    (OUT / "defaults.xml").write_text(DEFAULTS)
    (OUT / "assets.xml").write_text(ASSETS)
    (OUT / "contacts.xml").write_text("<mujocoinclude>\n  <contact>\n" + "".join(f'    <exclude body1="{s}_shin" body2="{s}_foot"/>\n' for s in "lr") + "  </contact>\n</mujocoinclude>\n")
    (OUT / "parts" / "leg_left.xml").write_text(leg("l", 1))
    (OUT / "parts" / "leg_right.xml").write_text(leg("r", -1))
    (OUT / "parts" / "upper_body.xml").write_text("<mujocoinclude>\n" + arm("l", 1) + arm("r", -1) + HEAD + "</mujocoinclude>\n")
    (OUT / "actuators_legs.xml").write_text(actuators([("hip_yaw", 30, 80), ("hip_roll", 40, 80), ("hip_pitch", 60, 80), ("knee", 60, 80), ("ankle_pitch", 25, 80), ("ankle_roll", 15, 80)]))
    (OUT / "actuators_arms.xml").write_text(actuators([("shoulder_pitch", 20, 0), ("shoulder_roll", 20, 0), ("elbow", 15, 0), ("wrist", 4, 0)]))
    # foot.obj: a sole-like solid - eight corners, four more hull points, two interior points (dropped by the hull)
    v = [(-0.07, -0.045, -0.02), (0.11, -0.04, -0.02), (0.11, 0.04, -0.02), (-0.07, 0.045, -0.02), (-0.06, -0.04, 0.03), (0.09, -0.03, 0.015), (0.09, 0.03, 0.015), (-0.06, 0.04, 0.03),
         (0.125, 0.0, -0.018), (-0.085, 0.0, -0.016), (0.02, 0.0, 0.045), (0.02, -0.05, -0.019), (0.0, 0.0, 0.0), (0.03, 0.01, -0.005)]
    f = [(1, 2, 3), (1, 3, 4), (5, 6, 7), (5, 7, 8), (1, 2, 6), (1, 6, 5), (2, 3, 7), (2, 7, 6), (3, 4, 8), (3, 8, 7), (4, 1, 5), (4, 5, 8)]
    (OUT / "meshes" / "foot.obj").write_text("# authored for this repository's tests\no foot\n" + "".join("v %g %g %g\n" % p for p in v) + "".join("f %d %d %d\n" % t for t in f))
    # pelvis_visual.stl: binary, millimetres (scaled by 0.001 in assets.xml), a box of 12 triangles; never collides
    c = np.array([[x, y, z] for x in (-110, 110) for y in (-90, 90) for z in (-60, 80)], np.float32)
    tri = [(0, 1, 3), (0, 3, 2), (4, 6, 7), (4, 7, 5), (0, 4, 5), (0, 5, 1), (2, 3, 7), (2, 7, 6), (0, 2, 6), (0, 6, 4), (1, 5, 7), (1, 7, 3)]
    with open(OUT / "meshes" / "pelvis_visual.stl", "wb") as fh:
        fh.write(b"authored for this repository's tests: binary STL, millimetres".ljust(80, b" "))
        fh.write(struct.pack("<I", len(tri)))
        for a, b, d in tri:
            n = np.cross(c[b] - c[a], c[d] - c[a])
            n = n / np.linalg.norm(n)
            fh.write(struct.pack("<12fH", *n, *c[a], *c[b], *c[d], 0))
    print("wrote", OUT)


if __name__ == "__main__":
    main()
