"""cylinder colliders against the ground - MJX collision_primitive.plane_cylinder, three contact slots per geom.
Known answers for the oracle's restatement, then the kernel (emulator build here, HIP with -m gpu) and the C++ twin against the oracle from
poses that cover the routine's three regimes: standing on a rim (general), flat on a disk (the degenerate direction), lying on its side."""
import math

import numpy as np
import pytest

from minppo_amd import mjcf
from minppo_amd.model import GEOM_CYLINDER, compile_model, load_model
from oracle.physics_oracle import Physics, PhysState, plane_cylinder

f32, f64 = np.float32, np.float64


def test_plane_cylinder_known_answers():
    r, half = 0.1, 0.25
    one = lambda h, axis, x=(1.0, 0.0, 0.0): [v[0] for v in plane_cylinder(np.array([h], f64), np.array([axis], f64), np.array([x], f64), f64(r), f64(half))]
    # upright, the bottom disk 5 cm under the plane: three rim points of that disk, 120 degrees apart, starting along the geom's x axis
    d, p = one(0.2, (0.0, 0.0, 1.0))
    assert d == pytest.approx([-0.05] * 3)
    s = r * math.sqrt(3) / 2
    assert p == pytest.approx(np.array([[r, 0, -half + 0.025], [-r / 2, s, -half + 0.025], [-r / 2, -s, -half + 0.025]]), abs=1e-12)
    assert np.linalg.norm(p[:, :2], axis=1) == pytest.approx([r] * 3)
    # upside down is the same cylinder
    d2, p2 = one(0.2, (0.0, 0.0, -1.0))
    assert d2 == pytest.approx(d) and p2[0] == pytest.approx(p[0])
    # tilted by 30 degrees about y: the lowest rim point is at height h - half cos30 - r sin30; the two others are higher by 1.5 r sin30
    a = math.radians(30)
    d, p = one(0.3, (math.sin(a), 0.0, math.cos(a)))
    low = 0.3 - half * math.cos(a) - r * math.sin(a)
    assert d == pytest.approx([low, low + 1.5 * r * math.sin(a), low + 1.5 * r * math.sin(a)])
    assert p[0] == pytest.approx([-half * math.sin(a) + r * math.cos(a), 0.0, low - 0.3 - 0.5 * low])   # on the rim, halfway into the overlap
    # lying on its side: the line of contact's two ends in slots 0 and 1 (both rims), slot 2 a rim point further up
    d, p = one(0.08, (1.0, 0.0, 0.0))
    assert d[:2] == pytest.approx([0.08 - r] * 2) and d[2] == pytest.approx(0.08 + 0.5 * r)
    assert sorted([p[0][0], p[1][0]]) == pytest.approx([-half, half]) and p[0][2] == pytest.approx(-r - 0.5 * (0.08 - r))


def test_cylinder_compiles_to_three_slots_and_round_trips(tmp_path):
    cm = load_model("synth_can")
    t = cm.t
    assert cm.ncon == 6 and int(t["ncyl"]) == 2 and t["con_cvx"].tolist() == [-2, -3, -4, -2, -3, -4] and cm.nefc == cm.nlimit + 24
    assert np.linalg.norm(t["con_axis"][0]) == pytest.approx(0.09) and np.linalg.norm(t["con_axis"][1]) == pytest.approx(1.0) and not t["con_axis"][2].any()
    assert (t["con_radius"][:3] == 0.06).all() and (t["con_radius"][3:] == 0.035).all()
    spec = mjcf.parse_mjcf(mjcf.to_mjcf(__import__("minppo_amd.model", fromlist=["synth_can"]).synth_can()), "synth_can")
    assert [g.type for b in spec.bodies for g in b.geoms] == [GEOM_CYLINDER, GEOM_CYLINDER]
    assert compile_model(spec).to_blob(True) == cm.to_blob(True)
    # from MJCF text with fromto; a cylinder that the masks pair with another geom is an error
    xml = """<mujoco model="roller"><worldbody><geom type="plane" size="0 0 1"/>
      <body name="a" pos="0 0 0.2"><freejoint/><inertial pos="0 0 0" mass="1" diaginertia="0.01 0.01 0.01"/>
        <geom type="cylinder" size="0.05" fromto="0 -0.1 0 0 0.1 0" {mask}/></body>
      <body name="b" pos="1 0 0.2"><freejoint/><inertial pos="0 0 0" mass="1" diaginertia="0.01 0.01 0.01"/><geom type="sphere" size="0.05" {mask}/></body>
    </worldbody></mujoco>"""
    ok = compile_model(mjcf.parse_mjcf(xml.format(mask='contype="0"')))
    assert ok.ncon == 4 and np.linalg.norm(ok.t["con_axis"][0]) == pytest.approx(0.1) and abs(ok.t["con_axis"][0][1]) == pytest.approx(0.1)
    with pytest.raises(ValueError, match="cylinder geom can only collide with the ground"):
        compile_model(mjcf.parse_mjcf(xml.format(mask="")))


def _poses(cm, N, rng):
    """Random orientations at heights where a rim is near the ground; pose 0 flat on its disk, pose 1 upside down, poses 2 / 3 on the side."""
    q = np.tile(cm.t["qpos0"], (N, 1))
    x = rng.normal(size=(N, 4))
    q[:, 3:7] = x / np.linalg.norm(x, axis=1, keepdims=True)
    q[:, 2] = rng.uniform(0.04, 0.13, N)
    q[:, 7] = rng.uniform(-0.7, 0.7, N)
    s = math.sqrt(0.5)
    q[0, 3:7], q[1, 3:7], q[2, 3:7], q[3, 3:7] = (1, 0, 0, 0), (0, 1, 0, 0), (s, 0, s, 0), (s, s, 0, 0)
    q[0, 2], q[1, 2], q[2, 2], q[3, 2] = 0.07, 0.09, 0.055, 0.05
    q[0:4, 7] = 0.0
    return q


def test_cylinder_contacts_match_the_oracle(be):
    from test_kernels_physics import _cost, _probe

    cm = load_model("synth_can")
    h, dims, _keep = be.model(cm)
    N = 48
    rng = np.random.default_rng(3)
    qpos, qvel = _poses(cm, N, rng), 0.3 * rng.standard_normal((N, cm.nv))
    ctrl = 0.3 * rng.standard_normal((N, cm.nu))
    q32 = [x.astype(f32) for x in (qpos, qvel, ctrl, np.zeros((N, cm.nv)))]
    ref = PhysState(qpos=q32[0].astype(f64), qvel=q32[1].astype(f64), ctrl=q32[2].astype(f64), qacc_warmstart=np.zeros((N, cm.nv)), time=np.zeros(N))
    Physics(cm.t).forward(ref)
    active = (ref.con_dist < 0)
    assert active[:, :3].any(1).sum() >= 20 and active[:, 3:].any(1).sum() >= 4 and active[0, :3].all() and active[2, :2].all()   # all three regimes, both geoms
    got = _probe(be, h, cm, *q32)
    for k, tol in dict(efc_D=5e-4, efc_aref=5e-4, efc_J=2e-5, qM=1e-5, xpos=1e-5).items():
        r, g = ref[k], got[k].reshape(ref[k].shape)
        assert np.abs(g - r).max() <= tol * (np.abs(r).max() + 1e-9), (k, np.abs(g - r).max() / (np.abs(r).max() + 1e-9))
    assert ((got["efc_D"].reshape(N, -1) > 0) == (ref.efc_D > 0)).all()
    c_got, c_ref = _cost(ref, got["qacc"]), _cost(ref, ref.qacc)
    crel = np.abs(c_got - c_ref) / (c_ref + 1e-3)
    assert np.median(crel) <= 1e-3 and crel.max() <= 0.25, (np.median(crel), crel.max())
    be.lib.model_close(h)


def test_can_settles_on_the_ground(be):
    """Dropped tilted from 14 cm, 400 steps (the kernel runs the first 80 and the last 100 of them): finite, ends on the ground (on its bottom disk or on its
    side - either way the centre between the radius and the half height above the plane), and the same through the C++ twin."""
    import ctypes as C

    from minppo_amd import _native as nat
    from oracle.cpu_twin import RewardCfg as TwinReward, Twin

    cm = load_model("synth_can")
    h, dims, _keep = be.model(cm)
    N, OP, R, nu = 4, dims.obs_pad, dims.rec_dim, cm.nu
    state, reset_rec, obs = be.zeros((N, R)), be.zeros((R,)), be.zeros((N, OP))
    rew, done = be.zeros((N,)), be.zeros((N,), np.uint8)
    be.lib.env_reset(h, N, be.ptr(state), be.ptr(reset_rec), be.ptr(obs), OP, be.ptr(rew), be.ptr(done), None, be.stream)
    rc = nat.RewardCfg(-100.0, 100.0, 2.0, 0.2, 0.5, 0.1, 4.0, 1.0, 1.25)
    tw = Twin(cm, reward=TwinReward(-100.0, 100.0, 2.0, 0.2, 0.5, 0.1, 4.0, 1.0, 1.25))
    tw.reset(N)
    a = np.zeros((N, nu), f32)
    for t in range(80):   # the fall and the first impact: both
        be.lib.env_step(h, N, 1, C.byref(rc), be.ptr(state), be.ptr(reset_rec), be.ptr(be.arr(a)), nu, be.ptr(obs), OP, be.ptr(rew), be.ptr(done), None, be.stream)
        tw.step(a)
    assert np.abs(be.host(state)[:, :cm.nq] - tw.state[:, :cm.nq]).max() < 5e-3
    for t in range(220):  # toppling over: the twin alone (the emulator build is slow), then the kernel takes over from its state
        tw.step(a)
    be.put(state, tw.state.copy())
    for t in range(100):
        be.lib.env_step(h, N, 1, C.byref(rc), be.ptr(state), be.ptr(reset_rec), be.ptr(be.arr(a)), nu, be.ptr(obs), OP, be.ptr(rew), be.ptr(done), None, be.stream)
        tw.step(a)
    st = be.host(state)
    for name, s in (("kernel", st), ("twin", tw.state)):
        assert np.isfinite(s).all(), name
        assert (s[:, 2] > 0.045).all() and (s[:, 2] < 0.105).all(), (name, s[:, 2])                  # resting: neither sunk nor hovering (the geom sits 5 - 10 mm off the body's origin)
        assert np.abs(s[:, cm.nq + 2]).max() < 0.05, (name, s[:, cm.nq + 2])  # ... and no longer falling or bouncing (on its side it may still roll)
    tw.close()
    be.lib.model_close(h)
