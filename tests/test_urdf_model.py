"""The oracle's Go1 model (oracle/qso_model.c, restated by hand from SURVEY App. A) against the REFERENCE's robot description.

tests/golden/urdf_tables.npz holds the numbers of go1/go1_description/urdf/go1.urdf (extracted by tests/golden/gen_golden.py:
link masses, centres of mass, inertia tensors, joint origins / axes / limits, collision primitives).  From those tables alone
this file builds a first-principles model -- homogeneous transforms, link kinetic energies, potential energy -- and derives the
joint-space mass matrix and the gravity forces by finite differences.  Nothing of the oracle's spatial algebra is used on this
side, so agreement pins both the restated data and the oracle's CRBA / RNEA to the reference's URDF."""
import numpy as np
import pytest

from oracle.qso import Oracle
from qs_amd.config import build_config

MOTORS = [f"{leg}_{j}_joint" for leg in ("FR", "FL", "RR", "RL") for j in ("hip", "thigh", "calf")]


def rpy_R(rpy):
    r, p, y = rpy
    Rx = np.array([[1, 0, 0], [0, np.cos(r), -np.sin(r)], [0, np.sin(r), np.cos(r)]])
    Ry = np.array([[np.cos(p), 0, np.sin(p)], [0, 1, 0], [-np.sin(p), 0, np.cos(p)]])
    Rz = np.array([[np.cos(y), -np.sin(y), 0], [np.sin(y), np.cos(y), 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def axis_R(axis, q):
    a = axis / np.linalg.norm(axis)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(q) * K + (1 - np.cos(q)) * K @ K


def hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])


def expm_so3(w):
    th = np.linalg.norm(w)
    return np.eye(3) if th < 1e-300 else axis_R(w / th, th)


class UrdfModel:
    def __init__(self, g):
        self.g = g
        self.links = list(g["link_name"])
        self.children = {}
        for j, p in enumerate(g["parent"]):
            self.children.setdefault(str(p), []).append(j)
        self.root = [l for l in self.links if l not in set(g["child"])][0]

    def link_frames(self, p_base, R_base, q12):
        """-> {link: (R, p)} in the world frame."""
        g, out = self.g, {self.root: (R_base, p_base)}
        stack = [self.root]
        while stack:
            l = stack.pop()
            R, p = out[l]
            for j in self.children.get(l, []):
                Rj = R @ rpy_R(g["joint_rpy"][j])
                pj = p + R @ g["joint_xyz"][j]
                name = str(g["joint_name"][j])
                if g["joint_type"][j] == "revolute":
                    Rj = Rj @ axis_R(g["axis"][j], q12[MOTORS.index(name)])
                out[str(g["child"][j])] = (Rj, pj)
                stack.append(str(g["child"][j]))
        return out

    def bodies(self, p_base, R_base, q12):
        """-> list of (mass, com_world, R_inertial_world, I_local) for every link."""
        g, fr, out = self.g, self.link_frames(p_base, R_base, q12), []
        for i, l in enumerate(self.links):
            R, p = fr[str(l)]
            ixx, ixy, ixz, iyy, iyz, izz = g["inertia"][i]
            I = np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]])
            out.append((g["mass"][i], p + R @ g["com"][i], R @ rpy_R(g["com_rpy"][i]), I))
        return out

    def displaced(self, p, R, q, v, eps):
        """configuration reached from (p, R, q) along the generalized velocity v = [w_b, v_b (base frame), qd] for a time eps"""
        return p + R @ v[3:6] * eps, R @ expm_so3(v[0:3] * eps), q + v[6:] * eps

    def kinetic(self, p, R, q, v, eps=1e-6):
        a, b = self.bodies(*self.displaced(p, R, q, v, eps)), self.bodies(*self.displaced(p, R, q, v, -eps))
        T = 0.0
        for (m, ca, Ra, I), (_, cb, Rb, _) in zip(a, b):
            vc = (ca - cb) / (2 * eps)
            W = (Ra @ Rb.T - Rb @ Ra.T) / (4 * eps)            # hat(w_world) to second order
            w = np.array([W[2, 1], W[0, 2], W[1, 0]])
            Rm = self.bodies_mid_R(Ra, Rb)
            T += 0.5 * m * vc @ vc + 0.5 * w @ (Rm @ I @ Rm.T) @ w
        return T

    @staticmethod
    def bodies_mid_R(Ra, Rb):
        U, _, Vt = np.linalg.svd(Ra + Rb)                       # mean rotation
        return U @ Vt

    def mass_matrix(self, p, R, q):
        n = 18
        E = np.eye(n)
        Td = np.array([self.kinetic(p, R, q, E[i]) for i in range(n)])
        M = np.zeros((n, n))
        for i in range(n):
            M[i, i] = 2 * Td[i]
            for j in range(i):
                M[i, j] = M[j, i] = self.kinetic(p, R, q, E[i] + E[j]) - Td[i] - Td[j]
        return M

    def potential(self, p, R, q, grav):
        return sum(m * grav * c[2] for m, c, _, _ in self.bodies(p, R, q))

    def gravity_force(self, p, R, q, grav, eps=1e-6):
        G = np.zeros(18)
        for i in range(18):
            e = np.zeros(18); e[i] = 1.0
            G[i] = (self.potential(*self.displaced(p, R, q, e, eps), grav) - self.potential(*self.displaced(p, R, q, e, -eps), grav)) / (2 * eps)
        return G


def quat_xyzw(R):
    from scipy.spatial.transform import Rotation
    return Rotation.from_matrix(R).as_quat()


@pytest.fixture(scope="module")
def model(golden):
    return UrdfModel(golden("urdf_tables.npz"))


def test_urdf_totals_and_limits(model, golden):
    g = model.g
    assert abs(g["mass"].sum() - 12.01301) < 1e-9 and model.root == "base"
    cfg, meta = build_config(n_envs=1, task_env="NO_TASK", observation_space_mode="ENCODER", enable_springs=True)
    rc = meta["robot_config"]
    for k, name in enumerate(MOTORS):
        j = list(g["joint_name"]).index(name)
        lo, hi, effort, vel = g["limit"][j]
        assert effort >= rc.TORQUE_LIMITS[k] - 1e-9      # configs_go1_*.py clips at or below the URDF's effort limits (calf: 33.55 vs 35.55)
        assert lo <= rc.RL_LOWER_ANGLE_JOINT[k] + 1e-9 and rc.RL_UPPER_ANGLE_JOINT[k] <= hi + 1e-9   # the RL box sits inside the joint range
    # collision primitives of App. A: foot spheres r = 0.02, trunk box, hip cylinders, thigh / calf boxes
    for leg in ("FR", "FL", "RR", "RL"):
        i = list(g["link_name"]).index(f"{leg}_foot")
        assert g["col_type"][i] == "sphere" and abs(g["col_size"][i][0] - 0.02) < 1e-12
    assert g["col_type"][list(g["link_name"]).index("trunk")] == "box"


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_mass_matrix_and_gravity_from_first_principles(model, seed):
    rng = np.random.default_rng(seed)
    cfg, _ = build_config(n_envs=1, task_env="NO_TASK", observation_space_mode="ENCODER", enable_springs=True, env_randomizer_mode="NONE")
    o = Oracle(cfg)
    s = o.get_state()
    from scipy.spatial.transform import Rotation
    R = Rotation.random(random_state=seed).as_matrix()
    p = rng.normal(size=3) + np.array([0, 0, 1.0])
    q = np.array([0.0, 0.8, -1.6] * 4) + 0.5 * rng.normal(size=12)
    s[0, :3], s[0, 3:7], s[0, 7:13], s[0, 13:25], s[0, 25:] = p, quat_xyzw(R), 0.0, q, 0.0
    o.set_state(s)
    H, Cb = o.crba_rnea(0)
    M = model.mass_matrix(p, R, q)
    np.testing.assert_allclose(H, M, atol=2e-6, rtol=1e-6)
    # at rest H a + Cb = tau, so Cb is the gravity force: dU/d(generalized coordinate)
    G = model.gravity_force(p, R, q, 9.8)
    np.testing.assert_allclose(Cb, G, atol=2e-6, rtol=1e-6)
    assert abs(np.linalg.norm(Cb[3:6]) - 12.01301 * 9.8) < 1e-4   # the base-force part is the robot's weight (g crosses the ABI as float32)


def lowest_points(model, p, R, q):
    """height of the lowest point of every link's collision primitive above the plane z = 0, from the URDF tables"""
    g, fr, out = model.g, model.link_frames(p, R, q), {}
    for i, l in enumerate(model.links):
        t = str(g["col_type"][i])
        if t == "none":
            continue
        Rl, pl = fr[str(l)]
        Rc, c = Rl @ rpy_R(g["col_rpy"][i]), pl + Rl @ g["col_xyz"][i]
        size = g["col_size"][i]
        if t == "sphere":
            low = c[2] - size[0]
        elif t == "box":
            low = c[2] - 0.5 * np.abs(Rc[2, :] * size).sum()
        else:  # cylinder along its local z
            az = Rc[2, 2]
            low = c[2] - 0.5 * size[1] * abs(az) - size[0] * np.sqrt(max(1 - az * az, 0.0))
        out[str(l)] = low
    return out


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_collision_geometry_matches_the_urdf(model, seed):
    """Contact flags of the oracle's substep vs distances computed from the URDF's collision primitives: feet (sphere r = 0.02
    at the foot_fixed joint) flagged exactly when closer than the 0.727 mm breaking threshold; trunk / hip / thigh / calf
    primitives counted as invalid contacts when they clearly penetrate and not when they clearly do not."""
    rng = np.random.default_rng(seed)
    from scipy.spatial.transform import Rotation
    # self_collision off: this test is about the primitives against the plane (a random hip angle of 1 rad does cross two calves)
    cfg, _ = build_config(n_envs=1, task_env="NO_TASK", observation_space_mode="ENCODER", enable_springs=False, env_randomizer_mode="NONE", self_collision=False)
    o = Oracle(cfg)
    feet = ["FR_foot", "FL_foot", "RR_foot", "RL_foot"]
    for trial in range(6):
        R = Rotation.from_euler("xyz", rng.uniform(-0.5, 0.5, size=3)).as_matrix()
        q = np.array([0.0, 0.8, -1.6] * 4) + 0.3 * rng.normal(size=12)
        low = lowest_points(model, np.zeros(3), R, q)
        foot_low = np.array([low[f] for f in feet])
        for gap in (+0.0004, -0.0010, +0.0020):        # lowest foot 0.4 mm above, 1 mm inside, 2 mm above the plane
            p = np.array([0.3, -0.2, -foot_low.min() + gap])
            s = o.get_state()
            s[0, :3], s[0, 3:7], s[0, 7:13], s[0, 13:25], s[0, 25:] = p, quat_xyzw(R), 0.0, q, 0.0
            o.set_state(s)
            o.phys_step(0, np.zeros(12))
            expect = (foot_low - foot_low.min() + gap) < 7.27e-4
            np.testing.assert_array_equal(o.get_info(1)[0] > 0.5, expect, err_msg=f"trial {trial} gap {gap}")
            assert o.get_info(5)[0, 0] == 0 or min(v for k, v in low.items() if k not in feet) - foot_low.min() + gap < 2e-3
    # a robot lying on its belly: trunk box and hips well inside the ground, feet in the air
    q = np.array([0.0, 1.3, -2.6] * 4)
    low = lowest_points(model, np.zeros(3), np.eye(3), q)
    p = np.array([0.0, 0.0, -low["trunk"] - 0.005])
    s = o.get_state()
    s[0, :3], s[0, 3:7], s[0, 7:13], s[0, 13:25], s[0, 25:] = p, [0, 0, 0, 1], 0.0, q, 0.0
    o.set_state(s)
    o.phys_step(0, np.zeros(12))
    n_pen = sum(1 for k, v in low.items() if k not in feet and v + p[2] < -1e-3)
    assert n_pen >= 1 and o.get_info(5)[0, 0] >= 1


@pytest.mark.parametrize("seed,friction,engine", [(0, "pyramid", "oracle"), (1, "pyramid", "oracle"), (2, "pyramid", "oracle"), (3, "pyramid", "oracle"),
                                                  (0, "cone", "oracle"), (3, "cone", "oracle"), (1, "pyramid", "kernel"), (3, "cone", "kernel")])
def test_contact_step_balances_impulse_and_momentum(model, seed, friction, engine):
    """(engine "kernel": the same check on the step kernel's own arithmetic -- csrc/qs_core.h in float32 through the host lane emulation --
    with float32 tolerances, so that the kernel code is tied to the URDF by first principles as well, not only through the oracle.)
    One oracle substep from rest on the ground under random joint torques, checked against first principles only: with the mass
    matrix M, the gravity force G and the foot-point Jacobians J all derived from the URDF tables by finite differences,
    M nu+ = dt (tau - G) + J^T p must hold for SOME contact impulses p (18 equations, 12 unknowns), whose normal parts are the foot
    forces the oracle reports, which push (p_n >= 0), stay inside the friction pyramid, and leave no foot approaching the ground
    faster than its gap allows.  Pins the constrained half of the step (contact Jacobians incl. the 0.08 m hip offset of the URDF,
    H^-1 J^T lambda, force read-out) the way test_mass_matrix_and_gravity_from_first_principles pins the unconstrained half."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(seed)
    mu, dt, r_foot = 0.7, 1e-3, 0.02
    cfg, _ = build_config(n_envs=1, isRLGymInterface=False, motor_control_mode="TORQUE", task_env="NO_TASK", observation_space_mode="ENCODER",
                          enable_springs=False, env_randomizer_mode="NONE", enable_action_filter=False, friction_model=friction,
                          solver_residual_threshold=0.0)     # the converged solve: all sweeps
    slop = cfg.contact_slop
    o = Oracle(cfg)
    o.reset()
    hold = o.get_info(2)[0].copy()                         # the joint torques that carry the settled stance
    o.set_params(0, np.array([[mu]]))
    s = o.get_state()
    s[0, 13:25] += 0.0007 * rng.normal(size=12)            # the feet end up at slightly different heights (fractions of a millimetre)
    s[0, 3:7] = quat_xyzw(Rotation.from_quat(s[0, 3:7]).as_matrix() @ Rotation.from_rotvec(0.0004 * rng.normal(size=3)).as_matrix())
    s[0, 7:13], s[0, 25:] = 0.0, 0.0                       # at rest: the bias force is gravity alone
    p, R, q = s[0, :3].copy(), Rotation.from_quat(s[0, 3:7]).as_matrix(), s[0, 13:25].copy()
    feet = [model.link_frames(p, R, q)[f"{leg}_foot"] for leg in ("FR", "FL", "RR", "RL")]
    p[2] -= min(pf[2] for _, pf in feet) - r_foot + 2e-4   # lowest foot 0.2 mm into the ground
    s[0, :3] = p
    o.set_state(s)
    tau = hold + 2.0 * rng.normal(size=12)
    tau[0::3] += rng.choice([-6.0, 6.0])                   # all hips pushed the same way: some feet reach the edge of the friction pyramid
    if engine == "kernel":
        from emu.emu import Emu
        k = Emu(cfg)
        k.set_mu(mu)
        k.set_state(s); k.phys_step(0, tau)
        s1, f_n = k.get_state()[0].astype(np.float64), k.get("R_FOOT_FORCE", 4)[0].astype(np.float64)
        tight, tol_f = 2e-4, 2e-3
    else:
        o.phys_step(0, tau)
        s1, f_n = o.get_state()[0], o.get_info(0)[0]
        tight, tol_f = 2e-6, 1e-4
    nu = np.concatenate([R.T @ s1[10:13], R.T @ s1[7:10], s1[25:]])        # [w_b, v_b (base frame), qd], the coordinates of H
    M, G = model.mass_matrix(p, R, q), model.gravity_force(p, R, q, 9.8)

    def contact_points(cfgn):                                                # material points of the four feet that touch down first
        fr = model.link_frames(*cfgn)
        return np.array([fr[f"{leg}_foot"][1] + fr[f"{leg}_foot"][0] @ (Rf0.T @ np.array([0, 0, -r_foot]))
                         for leg, (Rf0, _) in zip(("FR", "FL", "RR", "RL"), feet)])

    eps, J = 1e-6, np.zeros((12, 18))
    for i in range(18):
        e = np.zeros(18); e[i] = 1.0
        d = (contact_points(model.displaced(p, R, q, e, eps)) - contact_points(model.displaced(p, R, q, e, -eps))) / (2 * eps)   # [4, 3] world
        for k in range(4):
            J[3 * k: 3 * k + 3, i] = d[k]                                    # rows: x, y, z velocity of foot k's contact point
    rhs = M @ nu - dt * (np.concatenate([np.zeros(6), tau]) - G)
    imp, res, *_ = np.linalg.lstsq(J.T, rhs, rcond=None)
    assert np.linalg.norm(J.T @ imp - rhs) < tight * max(1.0, np.linalg.norm(rhs)) + 2e-8, "the velocity change is not a sum of foot impulses"
    imp = imp.reshape(4, 3)
    gap = contact_points((p, R, q))[:, 2]
    np.testing.assert_allclose(imp[:, 2] / dt, f_n, rtol=tol_f, atol=1e-3 if engine == "oracle" else 0.05)    # getContactPoints()[9] = normal impulse / dt
    slack = 1e-6 if engine == "oracle" else 2e-3
    assert np.all(imp[:, 2] >= -1e-9 - slack * dt) and f_n.sum() > 30.0 and (f_n > 0).sum() >= 2   # they push, several feet at once, a good part of the 118 N
    touching = f_n > 0
    if friction == "pyramid":                                                                 # x and y separately
        assert np.all(np.abs(imp[touching, 0]) <= mu * imp[touching, 2] * (1 + slack) + 1e-9 + slack * dt)
        assert np.all(np.abs(imp[touching, 1]) <= mu * imp[touching, 2] * (1 + slack) + 1e-9 + slack * dt)
    else:                                                                                     # the disc of PyBullet's implicit cone
        assert np.all(np.hypot(imp[touching, 0], imp[touching, 1]) <= mu * imp[touching, 2] * (1 + slack) + 1e-9 + slack * dt)
        assert np.any(np.hypot(imp[touching, 0], imp[touching, 1]) >= mu * imp[touching, 2] * (1 - slack) - slack * dt)   # and some foot is on its rim
    assert np.all(np.abs(imp[~touching]) < 1e-9 + slack * dt)
    vz = (J @ nu).reshape(4, 3)[:, 2]
    # a contact row lets a foot that is still `gap` (+ Bullet's linear slop) above the plane close exactly that distance in this step
    assert np.all(vz[touching] >= -np.maximum(gap[touching] + slop, 0) / dt - 2e-3), "a touching foot still moves into the ground"


def _foot_gaps(model, s, r_foot=0.02):
    from scipy.spatial.transform import Rotation
    fr = model.link_frames(s[:3], Rotation.from_quat(s[3:7]).as_matrix(), s[13:25])
    return np.array([fr[f"{leg}_foot"][1][2] - r_foot for leg in ("FR", "FL", "RR", "RL")])


def _engine(name, cfg, mu=0.8):
    """(substep(tau), get_state() -> float64 [37], set_state, foot normal forces of the last substep)"""
    if name == "kernel":
        from emu.emu import Emu
        k = Emu(cfg)
        k.set_mu(mu)
        return (lambda tau: k.phys_step(0, tau)), (lambda: k.get_state()[0].astype(np.float64)), (lambda s: k.set_state(s[None].astype(np.float32))), \
               (lambda: k.get("R_FOOT_FORCE", 4)[0].astype(np.float64))
    o = Oracle(cfg)
    o.set_params(0, np.array([[mu]]))
    return (lambda tau: o.phys_step(0, tau)), (lambda: o.get_state()[0]), (lambda s: o.set_state(s[None])), (lambda: o.get_info(0)[0])


@pytest.mark.parametrize("engine", ["oracle", "kernel"])
def test_penetration_recovers_at_the_erp_rate(model, engine):
    """Bullet's multibody contact row (btMultiBodyConstraintSolver::setupMultiBodyContactConstraint): a penetrating contact is pushed out at
    erp2 x penetration / dt, penetration = distance + linearSlop -- a foot 2 mm inside the ground leaves at 0.08 x 1.99 mm per substep, and the
    penetration decays geometrically at (1 - erp) per substep while the contact pushes.  Foot heights from the URDF tables, not from the
    engine.  (physics pins that need no PyBullet.)"""
    cfg, _ = build_config(n_envs=1, isRLGymInterface=False, motor_control_mode="TORQUE", task_env="NO_TASK", observation_space_mode="ENCODER",
                          enable_springs=False, env_randomizer_mode="NONE", enable_action_filter=False, solver_residual_threshold=0.0)
    o = Oracle(cfg)
    o.reset()
    hold, s = o.get_info(2)[0].copy(), o.get_state()[0].copy()
    s[7:13], s[25:] = 0.0, 0.0
    s[2] -= _foot_gaps(model, s).min() + 2e-3                      # lowest foot 2 mm inside
    step, get, put, _ = _engine(engine, cfg)
    put(s)
    erp, slop, tol = cfg.contact_erp, cfg.contact_slop, (1e-3 if engine == "oracle" else 5e-3)
    pen = -(_foot_gaps(model, s) + slop)
    assert np.all(pen > 1.5e-3)                                    # the settled stance is level: all four feet are inside
    for i in range(12):
        step(hold)
        pen1 = -(_foot_gaps(model, get()) + slop)
        np.testing.assert_allclose(pen1 / pen, 1.0 - erp, atol=tol, err_msg=f"substep {i}")
        pen = pen1
    assert np.all(pen < 2e-3 * (1 - erp) ** 12 * 1.05)


@pytest.mark.parametrize("engine", ["oracle", "kernel"])
def test_inelastic_touchdown_after_a_five_centimetre_drop(model, engine):
    """The robot is released at rest with its feet 5 cm above the ground, joints limp (no torque: in free fall it then moves as one rigid
    body), and lands at 0.99 m/s.  Known answers, all from first principles and the URDF tables: (1) until the first foot comes within the 0.727 mm contact
    range the centre of mass falls at g; (2) the contact is INELASTIC (Bullet's default restitution 0): in the substep in which a foot
    touches down its normal velocity goes from -1 m/s to what the row allows -- closing the remaining gap, or leaving at erp x penetration
    / dt -- and never to a rebound; (3) in every substep the vertical momentum of the whole robot changes by the feet's normal impulses
    minus m g dt (momentum = sum of link masses x the velocities of their centres of mass, differentiated from the URDF tables; impulses = reported foot forces x dt); (4) no foot goes
    deeper than the distance it covers in the one substep that carries it across the contact range."""
    from scipy.spatial.transform import Rotation
    cfg, _ = build_config(n_envs=1, isRLGymInterface=False, motor_control_mode="TORQUE", task_env="NO_TASK", observation_space_mode="ENCODER",
                          enable_springs=False, env_randomizer_mode="NONE", enable_action_filter=False, solver_residual_threshold=0.0)
    o = Oracle(cfg)
    o.reset()
    hold, s = np.zeros(12), o.get_state()[0].copy()
    s[7:13], s[25:] = 0.0, 0.0
    s[2] += 0.05 - _foot_gaps(model, s).min()
    step, get, put, forces = _engine(engine, cfg)
    put(s)
    dt, slop, thr, g, mass = cfg.dt, cfg.contact_slop, 7.27e-4, 9.8, 12.01301
    tol_p = 2e-6 if engine == "oracle" else 2e-4

    def pz(st, at=None, eps=1e-6):
        """vertical momentum of the whole robot from the URDF tables: sum of m_i x (vertical velocity of link i's centre of mass), the links'
        velocities by moving the configuration `at` (default: st's own) along st's generalized velocity"""
        at = st if at is None else at
        R = Rotation.from_quat(at[3:7]).as_matrix()
        nu = np.concatenate([R.T @ st[10:13], R.T @ st[7:10], st[25:]])
        a, b = model.bodies(*model.displaced(at[:3], R, at[13:25], nu, eps)), model.bodies(*model.displaced(at[:3], R, at[13:25], nu, -eps))
        return sum(m * (ca[2] - cb[2]) / (2 * eps) for (m, ca, _, _), (_, cb, _, _) in zip(a, b))

    touched, deepest, v_land, landed_at = False, 0.0, None, None
    s0 = get()
    for i in range(140):
        gap0 = _foot_gaps(model, s0)
        step(hold)
        s1, f = get(), forces()
        gap1 = _foot_gaps(model, s1)
        vz = (gap1 - gap0) / dt                                   # feet's vertical velocity over the substep (semi-implicit Euler: the NEW velocity)
        if not touched and not (f > 0).any():
            assert abs((s1[9] - s0[9]) / dt + g) < 1e-3 and np.abs(s1[25:]).max() < 1e-3   # (1) still one rigid body falling at g
        if (f > 0).any() and not touched:
            # (3) the touchdown substep: the robot comes in as one translating rigid body (no velocity-product forces), so the velocity step
            # M(q0) (nu1 - nu0) = dt (tau - G) + J^T p holds without further terms: the jump of the vertical momentum is the feet's impulse
            assert abs(pz(s1, at=s0) - pz(s0) - (f.sum() * dt - mass * g * dt)) < tol_p * max(1.0, f.sum() * dt), f"substep {i}"
            assert f.sum() * dt > 0.2                              # (limp legs: the first impulse stops the feet and shanks, 0.5 of the 11.9 kg m/s; the rest follows)
        elif (f > 0).any() and i % 4 == 0:
            # later substeps, legs folding: momentum at the new configuration against the impulses, to the integrator's O(dt)
            assert abs(pz(s1) - pz(s0) - (f.sum() * dt - mass * g * dt)) < 2e-2 * max(0.05, f.sum() * dt), f"substep {i}"
        for k in np.flatnonzero(f > 0):                           # (2) a foot that the ground pushes neither rebounds nor keeps falling
            allowed = -(gap0[k] + slop) / dt if gap0[k] + slop > 0 else cfg.contact_erp * -(gap0[k] + slop) / dt
            assert abs(vz[k] - allowed) < (2e-3 if engine == "oracle" else 1e-2), (i, k, vz[k], allowed)
        if (f > 0).any() and not touched:
            touched, landed_at, v_land = True, i, -s0[9]
        deepest = min(deepest, gap1.min())
        s0 = s1
    assert touched and 0.9 < v_land < 1.1 and 95 <= landed_at <= 105   # sqrt(2 g h) = 0.99 m/s after 0.101 s
    assert deepest > -(v_land * dt - thr) - 1e-4                       # (4) 0.3 mm: one substep's travel minus the contact range
