"""Bullet behaviour that needs no PyBullet to restate (VERDICT r01 "missing" 1, 2, 4): contact response of the non-foot links, the
calf self-collision rule (quadruped.py:237-241), the payload block on the ground, the mass-to-inertia rule.  Oracle known answers,
then the kernel arithmetic (host lane emulation) against the oracle."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot

from emu.emu import Emu
from oracle.qso import Oracle
from qs_amd.config import build_config, unit_inertia_table, URDF_LINKS

TOTAL_MASS = 12.01301
KW = dict(task_env="NO_TASK", observation_space_mode="ENCODER", enable_springs=True, enable_action_filter=False, env_randomizer_mode="NONE",
          isRLGymInterface=False, motor_control_mode="TORQUE", noise=False)


def make(n=1, **kw):
    cfg, _ = build_config(n_envs=n, **dict(KW, **kw))
    return cfg


def fallen_state(o, roll=0.0, pitch=0.0, z=0.12, q=(0.0, 1.2, -2.4)):
    s = o.get_state()
    s[:, :3] = [0, 0, z]
    s[:, 3:7] = Rot.from_euler("xyz", [roll, pitch, 0]).as_quat()
    s[:, 7:] = 0
    s[:, 13:25] = np.tile(q, 4)
    return s


# ------------------------------------------------------------------------------------------------ oracle known answers
@pytest.mark.parametrize("roll,pitch", [(0.0, 0.0), (1.45, 0.0), (0.0, 0.5)])
def test_fallen_robot_rests_on_the_floor(roll, pitch):
    """NO_TASK, no torques: the robot is dropped from a few centimetres in a folded pose (on its belly, on its side, nose down) and
    must come to rest ON the floor, carried by whatever touches it: the sum of all normal forces is m g and nothing sinks in."""
    cfg = make(solver_residual_threshold=0.0)
    o = Oracle(cfg)
    o.reset()
    o.set_state(fallen_state(o, roll, pitch, z=0.16))
    tau = np.zeros(12)
    for _ in range(1500):
        o.phys_step(0, tau)
    s = o.get_state()[0]
    assert np.abs(s[7:13]).max() < 2e-2 and np.abs(s[25:]).max() < 0.2, "not at rest"
    cs = o.contacts()
    ground = [c for c in cs if c[1] == 0]
    assert any(c[2] not in (5, 9, 13, 17) for c in ground), "no non-foot link on the ground"
    total = sum(c[5] for c in ground)
    assert total == pytest.approx(TOTAL_MASS * 9.8, rel=2e-2)
    assert min(c[4] for c in ground) > -2e-3, "a link sank into the floor"
    assert o.get_info(5)[0, 0] >= 1                                    # and they still count as invalid contacts


def test_without_body_contacts_the_fallen_robot_sinks():
    """The switch: body_contacts=False is round 1's behaviour (non-foot links only flag)."""
    cfg = make(body_contacts=False)
    o = Oracle(cfg)
    o.reset()
    o.set_state(fallen_state(o, 1.45, 0.0, z=0.16))
    for _ in range(600):
        o.phys_step(0, np.zeros(12))
    assert min(c[4] for c in o.contacts() if c[1] == 0) < -2e-2


def crossed_calves_state(o):
    """Front legs: hips rolled inwards and knees bent so that the two front calves cross under the trunk."""
    s = o.get_state()
    s[:, :3] = [0, 0, 0.6]
    s[:, 3:7] = [0, 0, 0, 1]
    s[:, 7:] = 0
    q = np.tile([0.0, 0.8, -1.6], 4).astype(float)
    q[0], q[3] = 0.55, -0.55        # FR hip +, FL hip -: both front legs swing towards the centre plane
    s[:, 13:25] = q
    return s


def test_crossed_calves_are_an_invalid_contact():
    cfg = make(task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", isRLGymInterface=True, motor_control_mode="PD")
    o = Oracle(cfg)
    o.reset()
    o.set_state(crossed_calves_state(o))
    o.phys_step(0, np.zeros(12))
    cs = o.contacts()
    assert any(c[0] == 1 and c[1] == 1 and c[2] == 4 and c[3] == 8 for c in cs), cs       # FR calf (4) x FL calf (8)
    assert o.get_info(5)[0, 0] >= 1
    a = np.zeros((1, 6), np.float32)
    _, _, done, trunc = o.step(a)
    assert done[0] and not trunc[0]                                                        # task_base.py:137-147 -> terminated
    cfg2 = make(task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", isRLGymInterface=True, motor_control_mode="PD", self_collision=False)
    o2 = Oracle(cfg2)
    o2.reset()
    o2.set_state(crossed_calves_state(o2))
    o2.phys_step(0, np.zeros(12))
    assert o2.get_info(5)[0, 0] == 0


def test_self_contacts_without_a_calf_do_not_count():
    """quadruped.py:237-241: only self-contacts that involve a calf are invalid.  Thighs pressed together (hips rolled inwards,
    legs straight enough that the calves stay apart) leave the count at zero."""
    cfg = make()
    o = Oracle(cfg)
    o.reset()
    s = o.get_state()
    s[:, :3] = [0, 0, 0.8]; s[:, 3:7] = [0, 0, 0, 1]; s[:, 7:] = 0
    q = np.tile([0.0, 0.0, -0.9], 4).astype(float)
    q[0], q[3] = 1.0, -1.0
    s[:, 13:25] = q
    o.set_state(s)
    o.phys_step(0, np.zeros(12))
    assert all(not (c[0] == 1 and c[1] == 1) for c in o.contacts()) or all(c[2] in (4, 8, 12, 16) for c in o.contacts() if c[0] == c[1] == 1)


def test_payload_block_on_the_ground_is_invalid():
    cfg = make(env_randomizer_mode="MASS_RANDOMIZER", seed=3)
    o = Oracle(cfg)
    o.reset()
    p = o.get_info(6)
    assert p[0, 20] > 0                                       # a payload was drawn
    p[0, 21:24] = [0.0, 0.0, -0.1]                            # hang it 10 cm below the base origin
    o.set_params(5, p)
    s = o.get_state()
    s[:, 2] = 0.148; s[:, 7:] = 0; s[:, 13:25] = np.tile([0.0, 1.2, -2.4], 4)
    o.set_state(s)
    o.phys_step(0, np.zeros(12))
    assert any(c[0] == 2 and c[1] == 0 for c in o.contacts())
    assert o.get_info(5)[0, 0] >= 1


# ------------------------------------------------------------------------------------------------ geometry of the link-link tests
def brute_overlap(ca, Ra, ha, cb, Rb, hb, n=14):
    """Sampling answer: a lattice of points of A (incl. its surface) tested for membership of B and vice versa."""
    g = np.linspace(-1, 1, n)
    P = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    def inside(c1, R1, h1, c2, R2, h2):
        pw = c1 + (P * h1) @ R1.T
        pl = (pw - c2) @ R2
        return bool(np.any(np.all(np.abs(pl) <= h2 + 1e-12, axis=1)))
    return inside(ca, Ra, ha, cb, Rb, hb) or inside(cb, Rb, hb, ca, Ra, ha)


def test_box_overlap_sat_equals_edge_clipping():
    """The kernel decides box / box contact by the separating-axis test, the oracle by clipping edges: both are exact, so they must agree
    on random rod-like boxes; where a sampling lattice finds a common point both must say overlap."""
    cfg = make()
    o, e = Oracle(cfg), Emu(cfg)
    rng = np.random.default_rng(0)
    ha, hb = np.array([0.008, 0.008, 0.1065]), np.array([0.017, 0.01225, 0.1065])
    n_yes = 0
    for k in range(600):
        Ra, Rb = Rot.random(random_state=rng.integers(1 << 30)).as_matrix(), Rot.random(random_state=rng.integers(1 << 30)).as_matrix()
        ca = np.zeros(3)
        cb = rng.normal(size=3) * [0.03, 0.03, 0.08]
        a, b = o.boxes_overlap(ca, Ra, ha, cb, Rb, hb), e.obb_overlap(ca, Ra, ha, cb, Rb, hb)
        # float32 SAT against float64 clipping: skip configurations within a hair of touching
        shrink, grow = o.boxes_overlap(ca, Ra, ha * 0.995, cb, Rb, hb * 0.995), o.boxes_overlap(ca, Ra, ha * 1.005, cb, Rb, hb * 1.005)
        if shrink == grow:
            assert a == b, (k, a, b)
        if brute_overlap(ca, Ra, ha, cb, Rb, hb):
            assert a
        n_yes += a
    assert 100 < n_yes < 500


# ------------------------------------------------------------------------------------------------ mass -> inertia
def test_unit_inertia_tables():
    sc, cs = unit_inertia_table("scale"), unit_inertia_table("collision_shape")
    m, i6, _ = URDF_LINKS["trunk"]
    np.testing.assert_allclose(sc[3] * m, i6, rtol=1e-6)
    # trunk: the principal frame is within 0.6 degrees of the link frame, so Bullet's rule is close to the solid box 0.3762 x 0.0935 x 0.114
    # -- from above: the AABB of the 0.38 m long box seen from the slightly rotated frame is a few millimetres wider
    lx, ly, lz = 0.3762, 0.0935, 0.114
    box = np.array([ly**2 + lz**2, lx**2 + lz**2, lx**2 + ly**2]) / 12
    assert np.all(cs[3][[0, 3, 5]] >= box * (1 - 1e-6)) and np.all(cs[3][[0, 3, 5]] <= box * 1.07)
    # the calf's box is a 16 mm rod: the shape rule makes its long-axis inertia tiny, the URDF's is 3e-5 / 0.131
    assert cs[2][5] < 0.5 * abs(sc[2][5]) or cs[2][5] > 0
    for k in range(4):                                                     # both are positive definite tensors
        for t in (sc[k], cs[k]):
            T = np.array([[t[0], t[1], t[2]], [t[1], t[3], t[4]], [t[2], t[4], t[5]]])
            assert np.all(np.linalg.eigvalsh(T) > 0)


def test_mass_randomizer_uses_the_shape_inertia_and_the_switch_changes_the_dynamics():
    """changeDynamics(mass=...) is called by the mass randomizer only: without it both rules give the URDF tensors."""
    base = dict(task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", isRLGymInterface=True, motor_control_mode="PD")
    c0, c1 = make(**base, mass_inertia_rule="scale"), make(**base, mass_inertia_rule="collision_shape")
    assert np.array_equal(np.array(c0.unit_inertia), np.array(c1.unit_inertia))
    m0, m1 = make(**base, env_randomizer_mode="MASS_RANDOMIZER", mass_inertia_rule="scale"), make(**base, env_randomizer_mode="MASS_RANDOMIZER")
    assert not np.array_equal(np.array(m0.unit_inertia), np.array(m1.unit_inertia))
    accs = []
    for c in (m0, m1):
        o = Oracle(c)
        o.reset()
        s = o.get_state(); s[:, 2] = 1.0; s[:, 10:13] = [1.0, 2.0, -1.0]
        o.set_state(s)
        accs.append(o.aba(0, np.tile([1.0, -2.0, 0.5], 4)))
    assert np.abs(accs[0] - accs[1]).max() > 1e-2


# ------------------------------------------------------------------------------------------------ kernel arithmetic vs oracle
@pytest.mark.parametrize("model", ["cone", "pyramid"])
def test_emu_matches_oracle_on_fallen_robots(model):
    """Robots lying on their side / belly / nose, flailing under random torques: support points of trunk, hips, thighs and calves,
    joint limits and feet all at once (the 12-rows-per-leg rare path).  Float32 oracle on the same float32 state, re-seated every step."""
    n = 6
    cfg = make(n, friction_model=model, solver_residual_threshold=0.0)
    o, e = Oracle(cfg, "f32"), Emu(cfg)
    o.reset(); e.reset()
    s = fallen_state(o)
    for i, (r, p) in enumerate([(1.45, 0.0), (0.0, 0.0), (0.0, 0.5), (-1.45, 0.2), (3.0, 0.0), (0.7, -0.4)]):
        s[i, 3:7] = Rot.from_euler("xyz", [r, p, 0]).as_quat()
    s[:, 2] = 0.2
    o.set_state(s); e.set_state(s)
    rng = np.random.default_rng(5)
    extra = 0
    for i in range(80):
        tau = (4.0 * rng.normal(size=(n, 12))).astype(np.float32) if i > 20 else np.zeros((n, 12), np.float32)
        st = o.get_state()
        o.set_state(st); e.set_state(st)
        o.step(tau); e.step(tau)
        so, se = o.get_state(), e.get_state()
        np.testing.assert_allclose(se[:, :7], so[:, :7], atol=5e-5, err_msg=f"pose step {i}")
        np.testing.assert_allclose(se[:, 7:13], so[:, 7:13], atol=2e-2, err_msg=f"base velocity step {i}")
        np.testing.assert_allclose(se[:, 13:25], so[:, 13:25], atol=2e-4, err_msg=f"q step {i}")
        np.testing.assert_allclose(se[:, 25:], so[:, 25:], atol=1e-1, err_msg=f"qd step {i}")
        np.testing.assert_array_equal(e.get("R_N_INVALID", 1)[:, 0] > 0, o.get_info(5)[:, 0] > 0)
        extra += sum(1 for k in range(n) for c in o.contacts(k) if c[1] == 0 and c[2] not in (5, 9, 13, 17) and c[5] > 1.0)
    assert extra > 200, extra                                   # the scenario does load non-foot links
    assert so[:, 2].min() > 0.03                                # and nobody fell through the floor


def test_emu_matches_oracle_on_fallen_robots_with_the_soft_payload():
    """All kinds of rows in one solve: 12 contact / limit rows per leg and the six rows of the payload block's fixed constraint."""
    n = 6
    cfg = make(n, solver_residual_threshold=0.0, payload="soft", env_randomizer_mode="MASS_RANDOMIZER", seed=3, settle_steps=300)
    o, e = Oracle(cfg, "f32"), Emu(cfg)
    o.reset(); e.reset()
    s = fallen_state(o)
    for i, (r, p) in enumerate([(1.45, 0.0), (0.0, 0.0), (0.0, 0.5), (-1.45, 0.2), (3.0, 0.0), (0.7, -0.4)]):
        s[i, 3:7] = Rot.from_euler("xyz", [r, p, 0]).as_quat()
    s[:, 2] = 0.2
    o.set_state(s); e.set_state(s)
    rng = np.random.default_rng(5)
    lam = 0.0
    for i in range(40):
        tau = (4.0 * rng.normal(size=(n, 12))).astype(np.float32) if i > 10 else np.zeros((n, 12), np.float32)
        st = o.get_state()
        o.set_state(st); e.set_state(st)
        o.step(tau); e.step(tau)
        so, se = o.get_state(), e.get_state()
        np.testing.assert_allclose(se[:, :7], so[:, :7], atol=5e-5, err_msg=f"pose step {i}")
        np.testing.assert_allclose(se[:, 13:25], so[:, 13:25], atol=2e-4, err_msg=f"q step {i}")
        b0, b1 = o.block(), e.block()
        np.testing.assert_allclose(b1[:, :3], b0["pos"], atol=5e-5, err_msg=f"block position step {i}")
        np.testing.assert_allclose(b1[:, 13:19], b0["lam"], atol=1e-3, err_msg=f"constraint impulses step {i}")
        lam = max(lam, np.abs(b0["lam"]).max())
    assert so[:, 2].min() > 0.03 and lam > 1e-3


def test_emu_counts_the_same_self_contacts():
    cfg = make(4)
    o, e = Oracle(cfg), Emu(cfg)
    o.reset(); e.reset()
    rng = np.random.default_rng(2)
    hits = 0
    for i in range(300):
        s = o.get_state()
        s[:, :3] = [0, 0, 1.0]; s[:, 3:7] = Rot.random(4, random_state=i).as_quat(); s[:, 7:] = 0
        lo, hi = np.tile([-1.04, -0.66, -2.72], 4), np.tile([1.04, 2.96, -0.84], 4)
        s[:, 13:25] = rng.uniform(lo, hi, size=(4, 12))
        if i % 3 == 0:
            s[:, 13:25] = crossed_calves_state(o)[:, 13:25] + 0.15 * rng.normal(size=(4, 12))
        o.set_state(s); e.set_state(s)
        for k in range(4):
            o.phys_step(k, np.zeros(12)); e.phys_step(k, np.zeros(12))
        no, ne = o.get_info(5)[:, 0], e.get("R_N_INVALID", 1)[:, 0]
        hits += int((no > 0).sum())
        np.testing.assert_array_equal(ne > 0, no > 0, err_msg=f"config {i}: {no} {ne}")
    assert hits > 100


# ------------------------------------------------------------------------------------------------ the reference's classification rule
def test_contacts_are_classified_as_the_reference_does():
    """tests/golden/contacts.npz: 160 states (standing, lying in random orientations, legs tangled in the air, payload block hanging low)
    with the contact list one substep produces (PyBullet's body / link numbering) and what the reference's own Quadruped.GetContactInfo
    (quadruped.py:224-258) made of that list.  From the recorded state the oracle must reproduce list and verdict, and the kernel
    arithmetic (lane emulation) the verdict: invalid or not, which feet stand, with what force."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "contacts.npz"))
    kinds = set()
    for k in range(len(g["states"])):
        cfg, _ = build_config(n_envs=1, **dict(KW, env_randomizer_mode="MASS_RANDOMIZER" if k % 4 == 3 else "NONE", seed=k))
        cfg.randomizer_flags |= 8          # keep the parameters given below
        o, e = Oracle(cfg), Emu(cfg)
        o.set_params(5, g["params"][k][None])
        e.records()[0, e.field("R_PARAMS"):e.field("R_PARAMS") + 24] = g["params"][k]
        o.set_state(g["states"][k][None]); e.set_state(g["states"][k][None])
        o.phys_step(0, np.zeros(12)); e.phys_step(0, np.zeros(12))
        ref = g["reference"][k]
        rows = g["contacts"][k]
        rows = rows[~np.isnan(rows[:, 0])]
        got = o.contacts(0)
        assert [c[:4] for c in got] == [tuple(int(x) for x in r[:4]) for r in rows], k
        for ba, bb, la, lb, dist, force in rows:
            kinds.add((int(ba), int(bb), "foot" if la in (5, 9, 13, 17) else "calf" if la in (4, 8, 12, 16) else "thigh" if la in (3, 7, 11, 15) else "other"))
        assert o.get_info(5)[0, 0] == ref[1], (k, o.get_info(5), ref[1])
        np.testing.assert_array_equal(o.get_info(1)[0], ref[6:10])
        np.testing.assert_allclose(o.get_info(0)[0], ref[2:6], rtol=0.25, atol=2.0)     # the recorded substep had a warm start, this one has none
        assert (e.get("R_N_INVALID", 1)[0, 0] > 0) == (ref[1] > 0), (k, e.get("R_N_INVALID", 1), ref[1])
        np.testing.assert_array_equal(e.get("R_FOOT_CONTACT", 4)[0], ref[6:10])
        np.testing.assert_allclose(e.get("R_FOOT_FORCE", 4)[0], o.get_info(0)[0], rtol=3e-2, atol=0.5)
    assert {(1, 0, "foot"), (1, 0, "calf"), (1, 0, "thigh"), (1, 0, "other"), (2, 0, "other"), (1, 1, "calf")} <= kinds, kinds


# ------------------------------------------------------------------------------------------------ what the payload weld leaves out
def test_payload_weld_against_the_soft_fixed_constraint():
    """The product welds the mass randomizer's payload block to the trunk; the reference holds it as a second body on a six-row fixed
    constraint of the same PGS (quadruped.py:796-819).  Both exist (payload="weld" | "soft"; the weld is the default): over jump episodes with
    random actions the constraint keeps the pivots within a tenth of a millimetre with 7 % of its impulse budget, and one env step from
    the same state moves the base by micrometres differently -- the weld is the soft constraint up to that."""
    kw = dict(task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True, enable_action_filter=True,
              env_randomizer_mode="TEST_RANDOMIZER", noise=False, seed=5, isRLGymInterface=True, motor_control_mode="PD")
    n = 8
    cw, _ = build_config(n_envs=n, payload="weld", **kw)
    cs, _ = build_config(n_envs=n, payload="soft", **kw)
    ow, osf = Oracle(cw), Oracle(cs)
    a0, a1 = ow.reset(), osf.reset()
    assert ow.get_info(6)[:, 20].min() > 0.05                      # every environment drew a payload
    np.testing.assert_allclose(a1, a0, atol=2e-4)                   # the settled poses agree
    assert osf.block()["gap"].max() < 1e-4
    rng = np.random.default_rng(0)
    worst_pose = worst_vel = gap = lam = 0.0
    for t in range(150):
        a = rng.uniform(-1, 1, size=(n, 6)).astype(np.float32)
        if t % 50 > 35:
            a[:] = [0, -1, 1, 0, -1, 1]
        sw = ow.get_state()
        if t % 10 == 0:                                             # free-running in between: the block keeps its own history
            osf.set_state(sw)
        else:
            ow.set_state(osf.get_state())
        s_in = ow.get_state()
        dw, ds = ow.step(a), osf.step(a)
        s0, s1 = ow.get_state(), osf.get_state()
        worst_pose = max(worst_pose, np.abs(s0[:, :7] - s1[:, :7]).max())
        worst_vel = max(worst_vel, np.abs(s0[:, 7:13] - s1[:, 7:13]).max())
        b = osf.block()
        gap, lam = max(gap, b["gap"].max()), max(lam, np.abs(b["lam"]).max())
        done = dw[2] | ds[2]
        if done.any():
            ow.reset(done.astype(np.uint8)); osf.reset(done.astype(np.uint8))
            osf.set_state(ow.get_state())
    # impulse bound of the constraint: 500 N x dt = 0.5 N s.  (Jumps load it to 7 %; since round 5 every link pushes back by default, and a
    # crash landing on the trunk -- the last env step of an episode that ends in a fall -- loads it to 23 %: 0.115.)
    assert gap < 5e-4 and lam < 0.15, (gap, lam)
    assert worst_pose < 5e-5 and worst_vel < 2e-2, (worst_pose, worst_vel)


@pytest.mark.parametrize("resid", [0.0, 1e-7])
@pytest.mark.parametrize("model", ["cone", "pyramid"])
def test_emu_matches_oracle_with_the_soft_payload(model, resid):
    """payload="soft" in the kernel arithmetic (the six replicated rows of the fixed constraint -- with every robot on its feet: next to
    the twelve foot rows of the common-path solver, round 3; with a joint at its stop or a link on the floor: in the many-rows solver --,
    the block's state in the record): free-running against the float32 oracle from the same reset, jump episodes under random actions
    with the mass randomizer's payload draws -- robot state, block state, constraint impulses and pivot gap."""
    n = 4
    kw = dict(task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True, enable_action_filter=True,
              env_randomizer_mode="TEST_RANDOMIZER", noise=False, seed=5, isRLGymInterface=True, motor_control_mode="PD", settle_steps=400)
    cfg, _ = build_config(n_envs=n, payload="soft", friction_model=model, solver_residual_threshold=resid, **kw)
    o, e = Oracle(cfg, "f32"), Emu(cfg)
    oo, oe = o.reset(), e.reset()
    np.testing.assert_allclose(oe, oo, atol=1e-3)
    assert o.get_info(6)[:, 20].min() > 0.05                       # every environment drew a payload
    b0, b1 = o.block(), e.block()
    np.testing.assert_allclose(b1[:, :3], b0["pos"], atol=1e-5)
    assert b1[:, 19].max() < 1e-4                                   # the settle left the pivots together
    rng = np.random.default_rng(0)
    lam = 0.0
    for t in range(60):
        a = rng.uniform(-1, 1, size=(n, 6)).astype(np.float32)
        if t % 30 > 20:
            a[:] = [0, -1, 1, 0, -1, 1]
        st = o.get_state()
        if t % 10 == 9:                                             # re-seat now and then: both place the block anew
            o.set_state(st); e.set_state(st)
        ro, re_ = o.step(a), e.step(a)
        so, se = o.get_state(), e.get_state()
        np.testing.assert_allclose(se[:, :7], so[:, :7], atol=5e-5, err_msg=f"pose step {t}")
        np.testing.assert_allclose(se[:, 7:13], so[:, 7:13], atol=5e-3, err_msg=f"base velocity step {t}")
        b0, b1 = o.block(), e.block()
        np.testing.assert_allclose(b1[:, :3], b0["pos"], atol=5e-5, err_msg=f"block position step {t}")
        np.testing.assert_allclose(b1[:, 3:7], b0["quat"], atol=5e-5)
        np.testing.assert_allclose(b1[:, 7:10], b0["v"], atol=5e-3)
        np.testing.assert_allclose(b1[:, 13:19], b0["lam"], atol=2e-4, err_msg=f"constraint impulses step {t}")
        np.testing.assert_allclose(b1[:, 19], b0["gap"], atol=2e-5)
        lam = max(lam, np.abs(b0["lam"]).max())
        np.testing.assert_array_equal(re_[2], ro[2])
        done = ro[2]
        if done.any():
            o.reset(done.astype(np.uint8)); e.reset(done.astype(np.uint8))
            e.set_state(o.get_state().astype(np.float32)); o.set_state(o.get_state().astype(np.float32))
    assert 1e-3 < lam < 0.1, lam                                    # the constraint carried the block (bound 0.5 N s)


def test_soft_payload_block_hits_the_ground_with_its_own_pose():
    """The invalid-contact rule for the payload block (quadruped.py:248-249) reads the block's own pose under payload="soft": a block
    hanging 0.1 under a crouched trunk is the only thing on the floor between z = 0.14 and 0.15."""
    cfg, _ = build_config(n_envs=1, payload="soft", task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", env_randomizer_mode="MASS_RANDOMIZER",
                          settle_steps=50)
    cfg.randomizer_flags |= 8          # keep the parameters given below
    o, e = Oracle(cfg), Emu(cfg)
    o.reset(); e.reset()
    p = o.get_info(6).copy()
    p[0, 20:24] = [0.6, 0.0, 0.0, -0.1]
    o.set_params(5, p)
    e.records()[0, e.field("R_PARAMS"):e.field("R_PARAMS") + 24] = p[0]
    s = o.get_state()
    s[:, 7:] = 0; s[:, 13:25] = np.tile([0.0, 1.3, -2.6], 4)
    for z, only_block in ((0.145, True), (0.155, False)):
        s[:, 2] = z
        o.set_state(s); e.set_state(s.astype(np.float32))
        np.testing.assert_allclose(e.block()[0, :3], o.block()["pos"][0], atol=1e-6)
        o.phys_step(0, np.zeros(12)); e.phys_step(0, np.zeros(12))
        assert [c[:3] for c in o.contacts(0)] == ([(2, 0, -1)] if only_block else [])
        assert (e.get("R_N_INVALID", 1)[0, 0] > 0) == only_block
    with pytest.raises(KeyError):
        build_config(n_envs=1, payload="glued")


@pytest.mark.parametrize("kw", [dict(friction_model="cone"), dict(friction_model="pyramid"),
                                dict(friction_model="cone", payload="soft", env_randomizer_mode="MASS_RANDOMIZER", seed=3, settle_steps=300)],
                         ids=["cone", "pyramid", "soft_payload"])
def test_support_margin_rule_leaves_out_only_rows_that_end_at_zero_impulse(kw):
    """qs_config::support_margin: the kernels build a non-foot support point's rows only once its normal row comes within
    0.5 m/s of acting; Bullet and the oracle build the rows of every point in range.  Robots thrown at the floor at 0.5 - 4 m/s in random
    attitudes with spinning joints -- approaching support points, impacts, under both friction models and with the payload block as its own
    body: the arithmetic with the rule (margin 0.5) must land where the arithmetic without it (margin inf) and the float32 oracle land,
    env step after env step from the same state.  A row left out that would have carried an impulse shows as a velocity error of the
    impact's size (decimetres per second), far above the bounds.  (Measured on this scenario: margin 0.5 and 1.0 differ from "every row" by
    rounding, 2e-6; margins 0 and 0.25 leave out a row that acts, 3e-4 rad/s in one env step of 320.)"""
    n = 8
    cfg_rule = make(n, solver_residual_threshold=0.0, **kw)
    cfg_all = make(n, solver_residual_threshold=0.0, support_margin=float("inf"), **kw)
    assert cfg_rule.support_margin == pytest.approx(0.5) and cfg_all.support_margin > 1e30 and cfg_rule.body_contacts == 1
    o, e_rule, e_all = Oracle(cfg_all, "f32"), Emu(cfg_rule), Emu(cfg_all)
    o.reset(); e_rule.reset(); e_all.reset()
    rng = np.random.default_rng(11)
    s = fallen_state(o)
    for i in range(n):
        s[i, 3:7] = Rot.from_euler("xyz", [rng.uniform(-3.1, 3.1), rng.uniform(-1.2, 1.2), rng.uniform(-3, 3)]).as_quat()
        s[i, 2] = rng.uniform(0.25, 0.45)
        s[i, 7:10] = [rng.uniform(-1, 1), rng.uniform(-1, 1), -rng.uniform(0.5, 4.0)]
        s[i, 10:13] = rng.uniform(-3, 3, 3)
        s[i, 13:25] = np.tile([0.0, 1.0, -2.0], 4) + rng.uniform(-0.3, 0.3, 12)
        s[i, 25:] = rng.uniform(-5, 5, 12)
    for em in (o, e_rule, e_all):
        em.set_state(s.astype(np.float32))
    loaded = impacts = off_oracle = 0
    for i in range(40):
        tau = (3.0 * rng.normal(size=(n, 12))).astype(np.float32)
        st = e_all.get_state()
        blk = e_all.block().copy() if "payload" in kw else None
        for em in (o, e_rule, e_all):     # (all three are re-seated: set_state also clears the contact warm start)
            em.set_state(st)
        if blk is not None:     # the block's own state travels with the re-seat (set_state re-places it at the constraint's rest otherwise)
            assert np.isfinite(blk).all()
        v_in = st[:, 9].copy()
        o.step(tau); e_rule.step(tau); e_all.step(tau)
        sa, sr, so = e_all.get_state(), e_rule.get_state(), o.get_state()
        # the SAME arithmetic with every row: strict
        np.testing.assert_allclose(sr[:, :7], sa[:, :7], atol=5e-6, err_msg=f"pose vs every row, step {i}")
        np.testing.assert_allclose(sr[:, 13:25], sa[:, 13:25], atol=2e-5, err_msg=f"q vs every row, step {i}")
        np.testing.assert_allclose(sr[:, 7:13], sa[:, 7:13], atol=1e-4, err_msg=f"base velocity vs every row, step {i}")
        np.testing.assert_allclose(sr[:, 25:], sa[:, 25:], atol=1e-3, err_msg=f"joint velocity vs every row, step {i}")
        # the float32 oracle (velocity-space sweep, every row): robots tumbling at 10 - 30 rad/s with joints running into their stops sit ON
        # discontinuities of the step map now and then (this scenario's one under the soft payload was run down: from states 1e-6
        # away both the oracle and the emulation jump between the same three outcomes, 1.4 rad/s apart) -- counted, not asserted one by one
        dev = (np.abs(sr[:, :7] - so[:, :7]).max(1) > 5e-5) | (np.abs(sr[:, 13:25] - so[:, 13:25]).max(1) > 2e-4) | \
              (np.abs(sr[:, 7:13] - so[:, 7:13]) > 5e-2 + 5e-3 * np.abs(so[:, 7:13])).any(1) | (np.abs(sr[:, 25:] - so[:, 25:]) > 5e-1 + 5e-3 * np.abs(so[:, 25:])).any(1)
        off_oracle += int(dev.sum())
        loaded += sum(1 for k in range(n) for c in o.contacts(k) if c[1] == 0 and c[2] not in (5, 9, 13, 17) and c[5] > 1.0)
        impacts += int(((sa[:, 9] - v_in) > 0.3).sum())
    assert loaded > 40 and impacts >= 4, (loaded, impacts)       # links did hit the floor and stop falling inside the run
    assert off_oracle <= 3, off_oracle                             # of 320 env steps


def rest_pose(engine, roll, pitch, steps=3000, mode=None):
    """a robot dropped in a folded pose and left alone for `steps` substeps: (state, ground contacts or None)"""
    if mode is not None:
        engine.set_manifold(mode)
    engine.set_state(fallen_state(engine, roll, pitch, z=0.16))
    for _ in range(steps):
        engine.phys_step(0, np.zeros(12))
    return engine.get_state()[0].copy()


@pytest.mark.parametrize("roll,pitch,pose_tol", [(1.45, 0.0, 1e-5), (0.0, 0.0, 1e-5), (3.0, 0.0, 1e-5), (2.8, 0.3, 1e-4)], ids=["side", "belly", "back", "back_tilted"])
def test_support_point_cap_against_four_points_per_primitive(roll, pitch, pose_tol):
    """Sensitivity of the cap on a leg's support points.  A btPersistentManifold can hold four
    points per collision primitive; oracle and kernels give a leg three contact points: the foot and two support points, or -- round 6 --
    three support points when the foot is off the ground.  The oracle's experiment mode 1 (up to four vertices / rim points per primitive) against
    the default on a robot dropped in a folded pose and left alone for 3 s: both at rest, carried by m g, and the resting poses agree to
    micrometres on its side, on its belly AND on its back.  Rounds 2-5 capped at two support points whatever the foot did (mode 3 keeps
    that): on its back -- trunk corner + both ends of each thigh box in range, the choice of two flipping from substep to substep -- the
    robot lay 1 mm off and kept creeping at 7 mm/s."""
    res = {}
    for mode in (0, 1, 3):
        o = Oracle(make(solver_residual_threshold=0.0))
        o.reset()
        res[mode] = rest_pose(o, roll, pitch, mode=mode)
        ground = [c for c in o.contacts() if c[1] == 0]
        assert sum(c[5] for c in ground) == pytest.approx(TOTAL_MASS * 9.8, rel=5e-3) and min(c[4] for c in ground) > -2e-3
    assert np.abs(res[0][:3] - res[1][:3]).max() < pose_tol
    assert np.abs(res[0][7:13]).max() < 2e-3 and np.abs(res[1][7:13]).max() < 2e-3        # at rest in every attitude
    if roll == 3.0:       # what the old cap did there
        assert np.abs(res[3][:3] - res[1][:3]).max() > 5e-4 and np.abs(res[3][7:13]).max() > 2e-3


@pytest.mark.parametrize("roll,pitch", [(1.45, 0.0), (0.0, 0.0), (3.0, 0.0)], ids=["side", "belly", "back"])
def test_kernel_arithmetic_rests_where_four_points_per_primitive_rest(roll, pitch):
    """The same known answer on the kernels' own arithmetic (host lane emulation): dropped on its side / belly / back the robot comes to rest
    within 1e-4 m of where the float64 oracle with four points per primitive does (the third support point of a leg whose foot is in the
    air rides in the foot's row slot, qs_core.h)."""
    cfg = make(solver_residual_threshold=0.0)
    o, e = Oracle(cfg), Emu(cfg)
    o.reset(); e.reset()
    ref, got = rest_pose(o, roll, pitch, mode=1), rest_pose(e, roll, pitch)
    # height and attitude (the plane's normal in trunk coordinates) are what the support points decide; where the robot has slid to on the
    # floor while it fell is the friction's stick / slip history (float32 against float64: a millimetre), held loosely
    up = lambda st: Rot.from_quat(st[3:7]).as_matrix()[2]
    assert abs(got[2] - ref[2]) < 1e-4 and np.abs(up(got) - up(ref)).max() < 1e-3, (got[2] - ref[2], up(got) - up(ref))
    assert np.abs(got[:2] - ref[:2]).max() < 5e-3
    assert np.abs(got[7:13]).max() < 5e-3


@pytest.mark.parametrize("roll,pitch", [(1.45, 0.0), (0.0, 0.0), (3.0, 0.0)], ids=["side", "belly", "back"])
def test_warm_starting_the_support_points_is_a_small_effect(roll, pitch):
    """Another stated deviation, measured: Bullet warm-starts the normal row of every manifold point that persists (0.1 x its last
    impulse); oracle and kernels do so for the feet only.  The oracle's experiment mode 2 carries a support point's impulse over while the same
    candidate stays selected: a robot dropped in a folded pose lands within 4e-5 m and 1.2e-3 rad of where it lands without, after 3 s."""
    res = []
    for mode in (0, 2):
        o = Oracle(make())
        o.reset()
        o.set_manifold(mode)
        o.set_state(fallen_state(o, roll, pitch, z=0.16))
        for _ in range(3000):
            o.phys_step(0, np.zeros(12))
        res.append(o.get_state()[0].copy())
    assert np.abs(res[0][:3] - res[1][:3]).max() < 1e-4 and np.abs(res[0][13:25] - res[1][13:25]).max() < 5e-3
    assert 0.0 < np.abs(res[0] - res[1]).max()                 # (the switch does something)
