#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference (this container only: /root/reference never travels).

    python tests/golden/gen_golden.py            # rewrites tests/golden/*.npz

The reference's numpy half is importable with four tiny shims: a `collections.Sequence`
alias, fake `pybullet` / `pybullet_data` / `pybullet_utils.bullet_client`, fake `gym`, fake `absl.logging`.
Nothing of the reference is copied into the repository: the fixtures hold inputs and the outputs the
reference's own functions produced for them.

Two kinds of fixtures:
  g1..g12  stateless functions of the numpy half, called directly on the reference's classes;
  g15      the reference's own QuadrupedGymEnv (reset/step, tasks, rewards, sensors, action filter) driven through a
           FakeBulletClient whose rigid-body step is the build's CPU oracle (oracle/qso_phys.c).  This pins the
           *caller* semantics around the absent PyBullet engine; it does not pin the engine itself.
"""
import collections
import collections.abc
import logging
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.environ.get("QS_GOLDEN_OUT") or os.path.join(REPO, "tests", "golden")   # QS_GOLDEN_OUT: regenerate elsewhere, e.g. to check reproducibility
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "quadruped-springs_amd"))
sys.path.insert(0, "/root/reference")


# --------------------------------------------------------------------------------------------- shims
def install_shims():
    collections.Sequence = collections.abc.Sequence
    pb = types.ModuleType("pybullet")
    pb.invertTransform = lambda position, orientation: (
        [-p for p in position], (-orientation[0], -orientation[1], -orientation[2], orientation[3]))
    pb.GUI = 1
    sys.modules["pybullet"] = pb
    pd_ = types.ModuleType("pybullet_data")
    pd_.getDataPath = lambda: "/nonexistent"
    sys.modules["pybullet_data"] = pd_
    pu = types.ModuleType("pybullet_utils")
    bc = types.ModuleType("pybullet_utils.bullet_client")
    bc.BulletClient = lambda *a, **k: FakeBulletClient()
    pu.bullet_client = bc
    sys.modules["pybullet_utils"] = pu
    sys.modules["pybullet_utils.bullet_client"] = bc

    gym = types.ModuleType("gym")

    class Env:
        pass

    class Wrapper(Env):
        def __init__(self, env):
            self.env = env

        def __getattr__(self, name):
            return getattr(self.env, name)

        @property
        def unwrapped(self):
            return self.env.unwrapped if hasattr(self.env, "unwrapped") else self.env

        def step(self, a):
            return self.env.step(a)

        def reset(self):
            return self.env.reset()

    class Box:
        def __init__(self, low, high, dtype=np.float32):
            self.low, self.high, self.dtype, self.shape = np.asarray(low, dtype), np.asarray(high, dtype), dtype, np.shape(low)

    gym.Env, gym.Wrapper = Env, Wrapper
    spaces = types.ModuleType("gym.spaces")
    spaces.Box = Box
    gym.spaces = spaces
    envs = types.ModuleType("gym.envs")
    reg = types.ModuleType("gym.envs.registration")
    reg.register = lambda **k: None
    envs.registration = reg
    gym.envs = envs
    for n, m in (("gym", gym), ("gym.spaces", spaces), ("gym.envs", envs), ("gym.envs.registration", reg)):
        sys.modules[n] = m
    absl = types.ModuleType("absl")
    absl.logging = logging
    sys.modules["absl"] = absl
    sys.modules["absl.logging"] = logging
    cv2 = types.ModuleType("cv2")
    sys.modules["cv2"] = cv2
    sb3, sb3c, sb3e = (types.ModuleType(n) for n in ("stable_baselines3", "stable_baselines3.common", "stable_baselines3.common.env_util"))

    def is_wrapped(env, cls):
        while env is not None:
            if isinstance(env, cls):
                return True
            env = env.__dict__.get("env")
        return False

    sb3e.is_wrapped = is_wrapped
    sb3.common, sb3c.env_util = sb3c, sb3e
    for m in (sb3, sb3c, sb3e):
        sys.modules[m.__name__] = m
    import matplotlib
    matplotlib.use = lambda *a, **k: None


# --------------------------------------------------------------------------------------------- fake Bullet on oracle physics
_JOINT_NAMES = ["floating_base", "imu_joint"]
for leg in ("FR", "FL", "RR", "RL"):
    _JOINT_NAMES += [f"{leg}_hip_joint", f"{leg}_thigh_joint", f"{leg}_calf_joint", f"{leg}_foot_fixed"]
_MOTOR_IDS = [2, 3, 4, 6, 7, 8, 10, 11, 12, 14, 15, 16]
_FOOT_IDS = [5, 9, 13, 17]
_LINK_MASS = {-1: 1e-5, 0: 5.204, 1: 0.001}
for k, base in enumerate((2, 6, 10, 14)):
    _LINK_MASS.update({base: 0.591, base + 1: 0.92, base + 2: 0.131, base + 3: 0.06})


class FakeBulletClient:
    """The 30-odd pybullet methods the reference calls; dynamics = oracle/qso_phys.c."""

    COV_ENABLE_PLANAR_REFLECTION = 0
    COV_ENABLE_RGB_BUFFER_PREVIEW = COV_ENABLE_DEPTH_BUFFER_PREVIEW = COV_ENABLE_SEGMENTATION_MARK_PREVIEW = COV_ENABLE_GUI = 0
    TORQUE_CONTROL, VELOCITY_CONTROL, POSITION_CONTROL = 2, 0, 1
    URDF_USE_SELF_COLLISION = 8
    JOINT_FIXED = 4
    LINK_FRAME = 1
    GUI = 1
    oracle_factory = None  # set by the generator: callable(dt, solver_iters) -> Oracle
    log = None

    def __init__(self):
        self.o = None
        self.dt = 0.001
        self.iters = 30
        self.mu = 1.0
        self.tau = np.zeros(12)
        self.mu_log = []

    # world
    def resetSimulation(self):
        self.mu = 1.0
        self.tau[:] = 0

    def setPhysicsEngineParameter(self, numSolverIterations=None, **k):
        if numSolverIterations is not None:
            self.iters = int(numSolverIterations)

    def setTimeStep(self, dt):
        self.dt = dt

    def setGravity(self, x, y, z):
        self.g = -z

    def loadURDF(self, path, basePosition=None, baseOrientation=None, flags=0, **k):
        if path.endswith("plane.urdf"):
            return 0
        self.o = FakeBulletClient.oracle_factory(self.dt, self.iters)
        s = self.o.get_state()
        s[0, :3] = basePosition
        s[0, 3:7] = baseOrientation
        s[0, 7:] = 0
        self.o.set_state(s)
        self.o.set_gravity(self.g)
        return 1

    def changeVisualShape(self, *a, **k):
        pass

    def configureDebugVisualizer(self, *a, **k):
        pass

    def disconnect(self):
        pass

    def resetDebugVisualizerCamera(self, *a, **k):
        pass

    # model queries
    def getNumJoints(self, body):
        return 18

    def getJointInfo(self, body, i):
        return (i, _JOINT_NAMES[i].encode("UTF-8"))

    def getDynamicsInfo(self, body, link):
        return (_LINK_MASS[link], 1.0, (0.0, 0.0, 0.0))

    GEOM_BOX = 3
    rand_log = None   # set by gen_randomizers: list collecting what the randomizers write through the client

    def createCollisionShape(self, *a, **k):
        return 7

    def createMultiBody(self, baseMass=0, baseCollisionShapeIndex=-1, basePosition=None, baseOrientation=None, **k):
        if FakeBulletClient.rand_log is not None:
            FakeBulletClient.rand_log.append(("payload", float(baseMass), np.array(basePosition, float) - self._state()[:3]))
        return 2

    def createConstraint(self, *a, **k):
        if FakeBulletClient.rand_log is not None:
            FakeBulletClient.rand_log.append(("constraint", a[4], np.array(a[7], float)))
        return 3

    def setCollisionFilterPair(self, *a, **k):
        pass

    def changeDynamics(self, body, link, lateralFriction=None, mass=None, **k):
        if mass is not None and FakeBulletClient.rand_log is not None:
            FakeBulletClient.rand_log.append(("mass", int(link), float(mass)))
        if body == 0 and link == -1 and lateralFriction is not None:
            self.mu = float(lateralFriction)
            self.mu_log.append(self.mu)
            if self.o is not None:
                self.o.set_params(0, np.array([self.mu]))

    # state
    def _state(self):
        return self.o.get_state()[0]

    def resetJointState(self, body, jid, angle, targetVelocity=0):
        s = self.o.get_state()
        k = _MOTOR_IDS.index(jid)
        s[0, 13 + k] = angle
        s[0, 25 + k] = targetVelocity
        self.o.set_state(s)

    def resetBasePositionAndOrientation(self, body, pos, orn):
        s = self.o.get_state()
        s[0, :3] = pos
        s[0, 3:7] = orn
        self.o.set_state(s)

    def resetBaseVelocity(self, body, lin, ang):
        s = self.o.get_state()
        s[0, 7:10] = lin
        s[0, 10:13] = ang
        self.o.set_state(s)

    def setJointMotorControl2(self, bodyIndex=None, jointIndex=None, controlMode=None, force=0, **k):
        if controlMode == self.TORQUE_CONTROL and jointIndex in _MOTOR_IDS:
            self.tau[_MOTOR_IDS.index(jointIndex)] += force  # App. D-1: torques accumulate within a step

    def stepSimulation(self):
        self.o.set_params(0, np.array([self.mu]))
        self.o.phys_step(0, self.tau)
        self.tau[:] = 0

    def getJointState(self, body, jid):
        s = self._state()
        k = _MOTOR_IDS.index(jid)
        return (s[13 + k], s[25 + k], (0,) * 6, 0.0)

    def getBasePositionAndOrientation(self, body):
        s = self._state()
        return tuple(s[:3]), tuple(s[3:7])

    def getBaseVelocity(self, body):
        s = self._state()
        return tuple(s[7:10]), tuple(s[10:13])

    def getContactPoints(self, *a, **k):
        # every contact point of the last substep with PyBullet's body / link numbering (oracle/qso.h qso_get_contacts), so that the
        # reference's own GetContactInfo (quadruped.py:224-258) does the classification: feet valid; thighs, other links on the plane,
        # the payload block, and self-contacts that involve a calf invalid
        return [(0, ba, bb, la, lb, (0, 0, 0), (0, 0, 0), (0, 0, 1), float(dist), float(force))
                for (ba, bb, la, lb, dist, force) in self.o.contacts(0)]

    # maths helpers (independent of the oracle's C versions: scipy)
    def getEulerFromQuaternion(self, q):
        from scipy.spatial.transform import Rotation as R
        return tuple(R.from_quat(q).as_euler("xyz"))

    def getQuaternionFromEuler(self, e):
        from scipy.spatial.transform import Rotation as R
        return tuple(R.from_euler("xyz", e).as_quat())

    def getMatrixFromQuaternion(self, q):
        from scipy.spatial.transform import Rotation as R
        return tuple(R.from_quat(q).as_matrix().flatten())

    def multiplyTransforms(self, positionA, orientationA, positionB, orientationB):
        from scipy.spatial.transform import Rotation as R
        ra, rb = R.from_quat(orientationA), R.from_quat(orientationB)
        return tuple(np.asarray(positionA) + ra.apply(positionB)), tuple((ra * rb).as_quat())

    def invertTransform(self, position, orientation):
        from scipy.spatial.transform import Rotation as R
        r = R.from_quat(orientation).inv()
        return tuple(-r.apply(position)), tuple(r.as_quat())


# --------------------------------------------------------------------------------------------- stateless goldens
def stub_env(cfg_mod, task_env="NO_TASK"):
    e = types.SimpleNamespace()
    e._robot_config = cfg_mod
    e.task_env = task_env
    return e


def stub_robot(cfg_mod):
    from quadruped_spring.env.quadruped import Quadruped
    r = object.__new__(Quadruped)
    r._robot_config = cfg_mod
    return r


def gen_stateless():
    import importlib
    from quadruped_spring.env.control_interface import action_interface as ai
    from quadruped_spring.env.control_interface import motor_interface as mi
    from quadruped_spring.env.quadruped_motor import QuadrupedMotorModel
    from quadruped_spring.utils.action_filter import ActionFilterButter
    rng = np.random.default_rng(0)
    out = {}
    for springs in (True, False):
        tag = "s1" if springs else "s0"
        mod = importlib.import_module(
            "quadruped_spring.go1.configs_go1_with_springs" if springs else "quadruped_spring.go1.configs_go1_without_springs")
        robot = stub_robot(mod)
        # G1/G2 action map + inverse
        for mname, mcls in (("PD", mi.MotorInterfacePD), ("CARTESIAN_PD", mi.MotorInterfaceCARTESIAN_PD)):
            for aname, acls in (("DEFAULT", ai.DefaultActionWrapper), ("SYMMETRIC", ai.SymmetricActionWrapper),
                                ("SYMMETRIC_NO_HIP", ai.SymmetricNoHipActionWrapper)):
                iface = acls(mcls(stub_env(mod)))
                iface._reset(robot)
                d = iface.get_action_space_dim()
                a = rng.uniform(-1.5, 1.5, size=(64, d))
                a[0] = 0
                cmd = np.array([iface._transform_action_to_motor_command(x) for x in a])
                ref = rng.uniform(-3, 3, size=(64, 12)) if mname == "PD" else rng.uniform(-0.5, 0.5, size=(64, 12))
                inv = np.array([iface._transform_motor_command_to_action(x) for x in ref])
                key = f"g1_{tag}_{mname}_{aname}"
                out[key + "_a"], out[key + "_cmd"], out[key + "_ref"], out[key + "_inv"] = a, cmd, ref, inv
                out[key + "_init_action"] = iface.get_init_action()
                out[key + "_landing_action"] = iface.get_landing_action()
                out[key + "_settle_cmd"] = iface._convert_reference_to_command(iface.get_init_pose())
        # G4 PD torques, G5 spring torques
        mm = QuadrupedMotorModel(robot_config=mod, enable_springs=springs, kp=mod.MOTOR_KP, kd=mod.MOTOR_KD,
                                 torque_limits=mod.RL_TORQUE_LIMITS, motor_control_mode="PD")
        q = rng.uniform(mod.REAL_LOWER_ANGLE_JOINT, mod.REAL_UPPER_ANGLE_JOINT, size=(128, 12))
        qd = rng.uniform(-30, 30, size=(128, 12))
        cmd = rng.uniform(mod.RL_LOWER_ANGLE_JOINT, mod.RL_UPPER_ANGLE_JOINT, size=(128, 12))
        out[f"g4_{tag}_q"], out[f"g4_{tag}_qd"], out[f"g4_{tag}_cmd"] = q, qd, cmd
        out[f"g4_{tag}_tau"] = np.array([mm.convert_to_torque(c, a, b)[0] for c, a, b in zip(cmd, q, qd)])
        out[f"g4_{tag}_tau_torque_mode"] = np.array([mm.convert_to_torque(c * 20, a, b, "TORQUE")[0] for c, a, b in zip(cmd, q, qd)])
        out[f"g4_{tag}_kp"], out[f"g4_{tag}_kd"] = np.array(mod.MOTOR_KP[:3], float), np.array(mod.MOTOR_KD[:3], float)
        if springs:
            qs = q.copy()
            qs[:16] = np.array(mod.SPRINGS_REST_ANGLE * 4) + rng.uniform(-1e-3, 1e-3, size=(16, 12))  # around the gates
            out["g5_q"], out["g5_qd"] = qs, qd
            out["g5_tau"] = np.array([mm.compute_spring_torques(a, b) for a, b in zip(qs, qd)])
            k2, b2 = [22.0, 18.5, 31.0], [0.33, 0.27, 0.31]
            mm._setSpringStiffness(k2); mm._setSpringDumping(b2)
            out["g5_k2"], out["g5_b2"] = np.array(k2), np.array(b2)
            out["g5_tau2"] = np.array([mm.compute_spring_torques(a, b) for a, b in zip(qs, qd)])
            out["g5_k"], out["g5_b"], out["g5_rest"] = (np.array(mod.SPRINGS_STIFFNESS, float), np.array(mod.SPRINGS_DAMPING, float),
                                                        np.array(mod.SPRINGS_REST_ANGLE, float))
        # G6 FK/J, G7 IK
        qq = rng.uniform(mod.REAL_LOWER_ANGLE_JOINT, mod.REAL_UPPER_ANGLE_JOINT, size=(128, 12))
        J = np.zeros((128, 4, 3, 3)); P = np.zeros((128, 4, 3))
        for i, x in enumerate(qq):
            for leg in range(4):
                J[i, leg], P[i, leg] = robot._compute_jacobian_and_position(x, leg)
        out[f"g6_{tag}_q"], out[f"g6_{tag}_J"], out[f"g6_{tag}_p"] = qq, J, P
        xyz = rng.uniform([-0.3, -0.25, -0.45], [0.3, 0.25, -0.02], size=(128, 4, 3))
        xyz[:8] *= 3.0  # out of reach -> clamps
        out[f"g7_{tag}_xyz"] = xyz
        out[f"g7_{tag}_q"] = np.array([[robot.ComputeInverseKinematics(leg, x[leg]) for leg in range(4)] for x in xyz])
        # G10 sensor bundles
        from quadruped_spring.env.sensors.sensor import SensorList
        from quadruped_spring.env.sensors.sensor_collection import SensorCollection
        for mode in SensorCollection()._dict:
            sl = SensorList(list(SensorCollection().get_el(mode)), stub_env(mod))
            sl._init(mod)
            out[f"g10_{tag}_{mode}_high"] = sl._get_high_limits()
            out[f"g10_{tag}_{mode}_low"] = sl._get_low_limits()
            out[f"g10_{tag}_{mode}_std"] = np.concatenate([np.atleast_1d(np.asarray(s._noise_std, float)).flatten() for s in sl._sensor_list])
            out[f"g10_{tag}_{mode}_names"] = np.array([s._name for s in sl._sensor_list])
        out[f"g10_{tag}_fallen_height"] = np.array(mod.IS_FALLEN_HEIGHT)
    # G3 Butterworth filter
    for fs in (100, 250, 500):
        f = ActionFilterButter(sampling_rate=fs, num_joints=6)
        out[f"g3_{fs}_b"], out[f"g3_{fs}_a"] = f.b[0].copy(), f.a[0].copy()
        f.reset()
        x0 = rng.uniform(-1, 1, size=6)
        f.init_history(x0)
        xs = rng.uniform(-1, 1, size=(200, 6))
        out[f"g3_{fs}_x0"], out[f"g3_{fs}_x"] = x0, xs
        out[f"g3_{fs}_y"] = np.array([f.filter(x) for x in xs])
    f = ActionFilterButter(sampling_rate=100, num_joints=1)
    f.reset()
    out["g3_step_response"] = np.array([f.filter(np.ones(1)) for _ in range(5)]).flatten()
    # G12 euler / backflip pitch on a quaternion sweep (reference uses scipy + Bullet's getEulerFromQuaternion)
    from scipy.spatial.transform import Rotation as R
    from quadruped_spring.env.sensors.robot_sensors import PitchBackFlip
    quats = np.concatenate([R.from_euler("y", np.linspace(-3.1, 3.1, 63)).as_quat(), R.random(96, random_state=1).as_quat()])
    pbf = []
    for sw in (False, True):
        for qv in quats:
            env = types.SimpleNamespace(robot=types.SimpleNamespace(GetBaseOrientation=lambda qv=qv: qv),
                                        task=types.SimpleNamespace(_switched_controller=sw))
            pbf.append(PitchBackFlip._get_pitch(env))
    out["g12_quat"] = quats
    out["g12_rpy_scipy_xyz"] = R.from_quat(quats).as_euler("xyz")
    out["g12_pitch_backflip"] = np.array(pbf).reshape(2, -1)
    out["g12_matrix"] = R.from_quat(quats).as_matrix()
    np.savez_compressed(os.path.join(OUT, "stateless.npz"), **out)
    print("stateless.npz:", len(out), "arrays")


# --------------------------------------------------------------------------------------------- G9 rewards as pure functions
def gen_rewards():
    import importlib
    from quadruped_spring.env.tasks import robot_tasks as rt
    rng = np.random.default_rng(9)
    out = {}
    classes = {"JUMPING_IN_PLACE": rt.JumpingInPlace, "JUMPING_FORWARD": rt.JumpingForward,
               "CONTINUOUS_JUMPING_FORWARD": rt.JumpingForwardContinuous, "CONTINUOUS_JUMPING_FORWARD2": rt.JumpingForwardContinuous2,
               "JUMPING_IN_PLACE_PPO": rt.JumpingInPlacePPO, "JUMPING_FORWARD_PPO": rt.JumpingForwardPPO,
               "JUMPING_IN_PLACE_PPO_HP": rt.JumpingInPlacePPOHP, "JUMPING_FORWARD_PPO_HP": rt.JumpingForwardPPOHP,
               "BACKFLIP": rt.BackFlip, "BACKFLIP_PPO": rt.BackflipPPO,
               "CONTINUOUS_JUMPING_FORWARD3": rt.JumpingForwardContinuous3, "CONTINUOUS_JUMPING_FORWARD_PPO": rt.ContinuousJumpingForwardPPO}
    mod = importlib.import_module("quadruped_spring.go1.configs_go1_with_springs")
    n = 96
    for name, cls in classes.items():
        rows, rew_step, rew_end, old_tau, new_tau, quats = [], [], [], [], [], []
        for i in range(n):
            term = bool(i % 2)
            env = types.SimpleNamespace()
            env._robot_config = mod
            env._MAX_EP_LEN = 10
            sim_step = int(rng.integers(0, 10000))
            env.get_sim_time = lambda s=sim_step: s * 0.001
            env.get_ac_interface = lambda: None
            ff = [float(rng.uniform(0, 600)), 0, 0, 0] if rng.uniform() < 0.5 else [float(rng.uniform(700, 1500)), 0, 0, 0]
            env.robot = types.SimpleNamespace(GetContactInfo=lambda ff=ff: (0, 0, ff, [1, 0, 0, 0]))
            t = cls(env)
            t._terminated = lambda term=term: term
            t._switched_controller = bool(rng.integers(0, 2))
            t._relative_max_height = float(rng.uniform(0, 1.2)) if i % 3 else float(rng.uniform(0, 0.25))
            t._max_pitch = float(rng.uniform(0, 0.6))
            t._max_delta_x = float(rng.uniform(0, 0.5))
            t._max_forward_distance = float(rng.uniform(0, 1.6))
            t._max_flight_time = float(rng.uniform(0, 0.6))
            t._max_height = float(rng.uniform(0.2, 1.0))
            t._pos_abs = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.1, 0.1), rng.uniform(0.1, 1.3)])
            t._orient_rpy = np.array([0.0, rng.uniform(-0.5, 0.5), 0.0])
            t._old_torque, t._new_torque = rng.uniform(-30, 30, 12), rng.uniform(-30, 30, 12)
            t.cumulative_fwd, t.cumulative_flight_time = float(rng.uniform(0, 2)), float(rng.uniform(0, 1))
            t.old_fwd = float(rng.uniform(0, 1.5))
            t.actual_fwd = t.old_fwd if i % 5 == 0 else float(rng.uniform(0, 1.5))
            t.max_pitch = float(rng.uniform(0, 6.5))
            # TaskContinuousJumping2 per-jump arrays (task_base.py:295-297): 0..6 recorded jumps
            nj = int(rng.integers(0, 7)) if i % 4 else 0
            jl = getattr(t, "jump_limit", 0.5); hl = getattr(t, "height_limit", 0.5)
            t.fwd_array = np.minimum(rng.uniform(0, 0.8, nj), jl) * (rng.uniform(size=nj) > 0.15)
            t.height_array = np.minimum(rng.uniform(0.2, 0.7, nj), hl)
            t.performance_array = getattr(t, "fwd_weight", 0.7) * t.fwd_array / jl + getattr(t, "height_weight", 0.3) * t.height_array / hl
            t.jump_counter = nj
            t.good_jump_counter = int(np.sum(t.performance_array >= getattr(t, "performance_bound", 0.85)))
            quat = np.array([0.0, np.sin(0.5 * t._orient_rpy[1]), 0.0, np.cos(0.5 * t._orient_rpy[1])])
            env.robot.GetBaseOrientation = lambda quat=quat: quat
            env.task = t
            row = np.zeros(48)
            row[0] = t._switched_controller; row[9] = t._max_flight_time; row[10] = t._max_forward_distance
            row[11] = t._max_pitch; row[12] = t._relative_max_height; row[13] = t._max_delta_x; row[14] = t._max_height
            row[15] = t.cumulative_fwd; row[16] = t.cumulative_flight_time; row[17] = t.old_fwd; row[18] = t.actual_fwd
            row[19] = t.max_pitch
            f = t.fwd_array
            row[20] = nj; row[21] = t.good_jump_counter; row[22] = f.sum(); row[23] = np.sum(f[f > 0] * np.log2(f[f > 0]))
            row[24] = t.height_array.sum(); row[25] = t.performance_array.sum()
            row[26] = t.performance_array.max() if nj else 0.0; row[27] = t.performance_array[-1] if nj else 0.0
            row[32:35] = t._pos_abs; row[38:41] = t._orient_rpy
            row[41] = 1 if term else 0; row[42] = sum(ff); row[43] = sim_step
            if not term:
                row[34] = max(row[34], 0.2)  # keep "not fallen" consistent with the forced _terminated()
                t._pos_abs[2] = row[34]
            quats.append(quat)
            rows.append(row); old_tau.append(t._old_torque); new_tau.append(t._new_torque)
            rew_step.append(float(t._reward())); rew_end.append(float(t._reward_end_episode()))
        out[f"g9_{name}_task"], out[f"g9_{name}_old_tau"], out[f"g9_{name}_new_tau"] = np.array(rows), np.array(old_tau), np.array(new_tau)
        out[f"g9_{name}_rew_step"], out[f"g9_{name}_rew_end"] = np.array(rew_step), np.array(rew_end)
        out[f"g9_{name}_quat"] = np.array(quats)
    np.savez_compressed(os.path.join(OUT, "rewards.npz"), **out)
    print("rewards.npz:", len(out), "arrays")


# --------------------------------------------------------------------------------------------- G15 differential traces
def scripted_actions(rng, n, d, jump_at, land=None, ext=(-0.8, 1.0)):
    """crouch, explosive extension, then smooth random actions (or a noisy landing pose): produces flight phases and landings."""
    a = np.zeros((n, d))
    for t in range(n):
        ph = t % jump_at
        if ph < jump_at * 0.5:
            base = np.array([0.0, 0.9, -0.9])
        elif ph < jump_at * 0.62:
            base = np.array([0.0, ext[0], ext[1]])
        else:
            base = np.array([0.0, 0.1, 0.2])
        full = np.tile(base, d // 3) if d % 3 == 0 else np.tile(base[1:], d // 2)
        a[t] = full + 0.15 * rng.standard_normal(d)
        if land is not None and ph >= jump_at * 0.62:
            a[t] = np.asarray(land) + 0.05 * rng.standard_normal(d)
    return a


def gen_traces():
    import importlib
    from qs_amd.config import build_config
    from oracle.qso import Oracle
    from quadruped_spring.env.quadruped_gym_env import QuadrupedGymEnv

    cases = [
        dict(name="jip_s1", task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True,
             enable_action_filter=True, action_space_mode="SYMMETRIC", motor_control_mode="PD", steps=260, jump_at=90),
        dict(name="jip_s0", task_env="JUMPING_IN_PLACE", observation_space_mode="ARS_BASIC", enable_springs=False,
             enable_action_filter=False, action_space_mode="SYMMETRIC", motor_control_mode="PD", steps=200, jump_at=70),
        dict(name="jf_s1", task_env="JUMPING_FORWARD", observation_space_mode="PPO_BASIC_CONTACT", enable_springs=True,
             enable_action_filter=True, action_space_mode="DEFAULT", motor_control_mode="PD", steps=220, jump_at=80),
        dict(name="cjf_s1", task_env="CONTINUOUS_JUMPING_FORWARD", observation_space_mode="PPO_CONTINUOUS_JUMPING_FORWARD",
             enable_springs=True, enable_action_filter=True, action_space_mode="SYMMETRIC", motor_control_mode="PD", steps=260, jump_at=60),
        dict(name="cjf2_s1", task_env="CONTINUOUS_JUMPING_FORWARD2", observation_space_mode="PPO_CONTINUOUS_JUMPING_FORWARD",
             enable_springs=True, enable_action_filter=True, action_space_mode="SYMMETRIC_NO_HIP", motor_control_mode="PD", steps=200, jump_at=60),
        dict(name="jipppo_s1", task_env="JUMPING_IN_PLACE_PPO", observation_space_mode="PPO_BASIC_X", enable_springs=True,
             enable_action_filter=True, action_space_mode="SYMMETRIC", motor_control_mode="PD", steps=200, jump_at=80),
        dict(name="jfppo_s1", task_env="JUMPING_FORWARD_PPO", observation_space_mode="LANDING_SENSOR", enable_springs=True,
             enable_action_filter=True, action_space_mode="SYMMETRIC", motor_control_mode="PD", steps=200, jump_at=80),
        dict(name="bf_s1", task_env="BACKFLIP", observation_space_mode="PPO_BACKFLIP", enable_springs=True,
             enable_action_filter=True, action_space_mode="SYMMETRIC", motor_control_mode="PD", steps=200, jump_at=80),
        dict(name="bfppo_s1", task_env="BACKFLIP_PPO", observation_space_mode="PPO_BACKFLIP", enable_springs=True,
             enable_action_filter=True, action_space_mode="SYMMETRIC", motor_control_mode="PD", steps=200, jump_at=80),
        dict(name="cjf3_s1", task_env="CONTINUOUS_JUMPING_FORWARD3", observation_space_mode="PPO_CONTINUOUS_JUMPING_FORWARD",
             enable_springs=True, enable_action_filter=True, action_space_mode="SYMMETRIC", motor_control_mode="PD", steps=420, jump_at=70),
        dict(name="cjfppo_s1", task_env="CONTINUOUS_JUMPING_FORWARD_PPO", observation_space_mode="PPO_CONTINUOUS_JUMPING_FORWARD",
             enable_springs=True, enable_action_filter=True, action_space_mode="SYMMETRIC", motor_control_mode="PD", steps=420, jump_at=70),
        dict(name="cart_s1", task_env="JUMPING_IN_PLACE", observation_space_mode="CARTESIAN_NO_IMU", enable_springs=True,
             enable_action_filter=True, action_space_mode="SYMMETRIC", motor_control_mode="CARTESIAN_PD", steps=150, jump_at=70),
        # gym_env.py:187-205 _interpolate_actions, with and without the filter
        dict(name="interp_f1", task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True, enable_action_interpolation=True,
             enable_action_filter=True, action_space_mode="SYMMETRIC", motor_control_mode="PD", steps=120, jump_at=60),
        # the two high-performance PPO tasks, the two remaining sensor bundles, BASELINE.json configs[1] (dt = 2 ms x 5, 60 sweeps)
        dict(name="jipppohp_s1", task_env="JUMPING_IN_PLACE_PPO_HP", observation_space_mode="ARS_SENSOR", enable_springs=True,
             enable_action_filter=True, action_space_mode="SYMMETRIC", motor_control_mode="PD", steps=200, jump_at=80),
        dict(name="jfppohp_s0", task_env="JUMPING_FORWARD_PPO_HP", observation_space_mode="ARS_BACKFLIP", enable_springs=False,
             enable_action_filter=True, action_space_mode="SYMMETRIC_NO_HIP", motor_control_mode="PD", steps=200, jump_at=80),
        dict(name="dt2_s1", task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True, time_step=0.002, action_repeat=5,
             enable_action_filter=True, action_space_mode="SYMMETRIC", motor_control_mode="PD", steps=200, jump_at=80),
        # hopf_network.py:183-190 style: raw joint torques, no RL interface; reset settles by PD for 1500 steps (control_interface/utils.py:22-31)
        dict(name="raw_tau", isRLGymInterface=False, task_env="NO_TASK", observation_space_mode="ENCODER", enable_springs=True,
             enable_action_filter=False, action_space_mode="DEFAULT", motor_control_mode="TORQUE", steps=150, jump_at=60, raw_torque=True),
        dict(name="raw_tau_s0", isRLGymInterface=False, task_env="NO_TASK", observation_space_mode="ENCODER_2", enable_springs=False, action_repeat=1,
             enable_action_filter=False, action_space_mode="DEFAULT", motor_control_mode="TORQUE", steps=400, jump_at=60, raw_torque=True),
        dict(name="interp_f0", task_env="JUMPING_FORWARD", observation_space_mode="PPO_BASIC", enable_springs=False, enable_action_interpolation=True,
             enable_action_filter=False, action_space_mode="DEFAULT", motor_control_mode="PD", steps=120, jump_at=60),
        # BASELINE.json configs[0] as SURVEY.md 8d spells it: one environment, no springs, PD, jump in place, SYMMETRIC actions, the PPO_BASIC bundle
        dict(name="cfg0_s0", task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=False,
             enable_action_filter=False, action_space_mode="SYMMETRIC", motor_control_mode="PD", steps=240, jump_at=70),
    ]
    out = {}
    for case in cases:
        name = case["name"]
        kw = {k: v for k, v in case.items() if k not in ("name", "steps", "jump_at", "raw_torque")}
        mod = importlib.import_module("quadruped_spring.go1.configs_go1_with_springs" if kw["enable_springs"]
                                      else "quadruped_spring.go1.configs_go1_without_springs")
        saved = {}
        for attr in dir(mod):  # deterministic traces: switch the i.i.d. sensor noise off in the reference's config
            if attr.endswith("_NOISE"):
                saved[attr] = getattr(mod, attr)
                setattr(mod, attr, np.zeros_like(np.asarray(saved[attr], float)))
        saved_upper = mod.RL_UPPER_ANGLE_JOINT.copy()

        def factory(dt, iters, kw=kw):
            cfg, _ = build_config(n_envs=1, noise=False, env_randomizer_mode="NONE", **dict(kw, time_step=dt))
            cfg.solver_iters = iters
            cfg.randomizer_flags = 8
            return Oracle(cfg)

        FakeBulletClient.oracle_factory = factory
        np.random.seed(1234)
        rng = np.random.default_rng(5)
        env = QuadrupedGymEnv(env_randomizer_mode="GROUND_RANDOMIZER", **kw)
        client = env._pybullet_client
        d = env.action_dim
        acts = scripted_actions(rng, case["steps"], d, case["jump_at"])
        raw = bool(case.get("raw_torque"))   # torques of a joint PD toward a slowly moving crouch, evaluated in closed loop on the reference env
        keys = None
        obs_l, rew_l, done_l, trunc_l, reset_obs, reset_at, mu_at_reset, state_l = [], [], [], [], [], [], [], []
        o = env.reset()
        keys = list(o.keys())
        flat = lambda ob: np.concatenate([np.atleast_1d(np.asarray(ob[k], float)).flatten() for k in keys])
        reset_obs.append(flat(o)); reset_at.append(0); mu_at_reset.append(client.mu)
        for t in range(case["steps"]):
            if raw:
                q_des = np.array([0.0, 0.8, -1.6] * 4) + 0.25 * np.sin(2 * np.pi * t * env.env_time_step / 0.5) * np.array([0.0, 1.0, -2.0] * 4)
                acts[t] = np.clip(60.0 * (q_des - env.robot.GetMotorAngles()) - 1.5 * env.robot.GetMotorVelocities() + 0.5 * rng.standard_normal(12), -20, 20)
            ob, r, dn, info = env.step(acts[t])
            obs_l.append(flat(ob)); rew_l.append(r); done_l.append(dn)
            trunc_l.append(bool(info.get("TimeLimit.truncated", False)))
            state_l.append(client.o.get_state()[0].copy())
            if dn:
                o = env.reset()
                client = env._pybullet_client
                reset_obs.append(flat(o)); reset_at.append(t + 1); mu_at_reset.append(client.mu)
        out[f"{name}_actions"], out[f"{name}_obs"], out[f"{name}_rew"] = acts, np.array(obs_l), np.array(rew_l, float)
        out[f"{name}_done"], out[f"{name}_trunc"] = np.array(done_l), np.array(trunc_l)
        out[f"{name}_reset_obs"], out[f"{name}_reset_at"], out[f"{name}_mu"] = np.array(reset_obs), np.array(reset_at), np.array(mu_at_reset)
        out[f"{name}_state"] = np.array(state_l)
        out[f"{name}_keys"] = np.array(keys)
        out[f"{name}_kwargs"] = np.array(repr(kw))
        out[f"{name}_landing_action"] = np.asarray(env.get_landing_action(), float)
        print(f"trace {name}: steps={case['steps']} episodes={len(reset_at)} dones={int(np.sum(done_l))} obs_dim={len(obs_l[0])} "
              f"max_h={np.max(np.array(state_l)[:, 2]):.3f}")
        for attr, v in saved.items():
            setattr(mod, attr, v)
        mod.RL_UPPER_ANGLE_JOINT[:] = saved_upper
    np.savez_compressed(os.path.join(OUT, "traces.npz"), **out)
    print("traces.npz:", len(out), "arrays")


# --------------------------------------------------------------------------------------------- reference-state initialisation
def gen_rsi():
    """reset() with a desired robot state (set_robot_desired_state -> quadruped.py:521-525, gym_env.py:289-290: no settle, _last_action
    stays zero), as ReferenceStateInitializationWrapper uses it, followed by a few steps."""
    import importlib
    from qs_amd.config import build_config
    from oracle.qso import Oracle
    from quadruped_spring.env.quadruped_gym_env import QuadrupedGymEnv
    out = {}
    tr = np.load(os.path.join(OUT, "traces.npz"))
    for name, src, at, kw in (
            ("rsi_s1", "jip_s1", 118, dict(task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True, enable_action_filter=True,
                                            action_space_mode="SYMMETRIC", motor_control_mode="PD")),
            ("rsi_s0", "jf_s1", 40, dict(task_env="JUMPING_FORWARD_PPO", observation_space_mode="PPO_BASIC_CONTACT", enable_springs=False, enable_action_filter=False,
                                         action_space_mode="DEFAULT", motor_control_mode="PD"))):
        mod = importlib.import_module("quadruped_spring.go1.configs_go1_with_springs" if kw["enable_springs"]
                                      else "quadruped_spring.go1.configs_go1_without_springs")
        saved = {}
        for attr in dir(mod):
            if attr.endswith("_NOISE"):
                saved[attr] = getattr(mod, attr)
                setattr(mod, attr, np.zeros_like(np.asarray(saved[attr], float)))

        def factory(dt, iters, kw=kw):
            cfg, _ = build_config(n_envs=1, noise=False, env_randomizer_mode="NONE", **dict(kw, time_step=dt))
            cfg.solver_iters = iters
            cfg.randomizer_flags = 8
            return Oracle(cfg)

        FakeBulletClient.oracle_factory = factory
        np.random.seed(4321)
        rng = np.random.default_rng(17)
        env = QuadrupedGymEnv(env_randomizer_mode="GROUND_RANDOMIZER", **kw)
        st = tr[f"{src}_state"][at].copy()      # a state in the middle of a jump
        d = env.action_dim
        desired = (np.zeros(d), st[13:25], st[25:37], st[0:3], st[3:7], st[7:10], st[10:13], [0.0])   # the demo-row tuple of read_demo
        env.set_robot_desired_state(desired)
        o = env.reset()
        client = env._pybullet_client
        keys = list(o.keys())
        flat = lambda ob: np.concatenate([np.atleast_1d(np.asarray(ob[k], float)).flatten() for k in keys])
        acts = scripted_actions(rng, 40, d, 200)
        obs_l, rew_l, done_l, state_l = [], [], [], []
        for t in range(40):
            ob, r, dn, info = env.step(acts[t])
            obs_l.append(flat(ob)); rew_l.append(r); done_l.append(dn); state_l.append(client.o.get_state()[0].copy())
            if dn:
                break
        out[f"{name}_desired"], out[f"{name}_reset_obs"], out[f"{name}_mu"] = st, flat(o), np.array(client.mu)
        out[f"{name}_actions"], out[f"{name}_obs"], out[f"{name}_rew"] = acts[:len(obs_l)], np.array(obs_l), np.array(rew_l, float)
        out[f"{name}_done"], out[f"{name}_state"], out[f"{name}_kwargs"] = np.array(done_l), np.array(state_l), np.array(repr(kw))
        print(f"rsi {name}: start z={st[2]:.3f} vz={st[9]:.3f}, {len(obs_l)} steps, dones={int(np.sum(done_l))}")
        for attr, v in saved.items():
            setattr(mod, attr, v)
    np.savez_compressed(os.path.join(OUT, "rsi.npz"), **out)


# --------------------------------------------------------------------------------------------- G22 Go1 model data
def gen_urdf():
    """Numeric tables of the reference's robot description (go1/go1_description/urdf/go1.urdf): per link mass,
    centre of mass, inertia tensor and collision primitives; per joint type, parent, child, origin, axis and limits."""
    import xml.etree.ElementTree as ET
    root = ET.parse("/root/reference/quadruped_spring/go1/go1_description/urdf/go1.urdf").getroot()
    f3 = lambda t, d="0 0 0": np.array([float(x) for x in (t if t is not None else d).split()])
    links, joints = root.findall("link"), root.findall("joint")
    lname = [l.get("name") for l in links]
    mass, com, com_rpy, inertia, col_type, col_size, col_xyz, col_rpy = [], [], [], [], [], [], [], []
    for l in links:
        i = l.find("inertial")
        mass.append(float(i.find("mass").get("value")))
        o = i.find("origin")
        com.append(f3(o.get("xyz") if o is not None else None)); com_rpy.append(f3(o.get("rpy") if o is not None else None))
        t = i.find("inertia")
        inertia.append([float(t.get(k)) for k in ("ixx", "ixy", "ixz", "iyy", "iyz", "izz")])
        c = l.find("collision")
        if c is None:
            col_type.append("none"); col_size.append(np.zeros(3)); col_xyz.append(np.zeros(3)); col_rpy.append(np.zeros(3))
            continue
        g = c.find("geometry")[0]
        size = {"box": lambda: f3(g.get("size")), "sphere": lambda: np.array([float(g.get("radius")), 0, 0]),
                "cylinder": lambda: np.array([float(g.get("radius")), float(g.get("length")), 0])}[g.tag]()
        o = c.find("origin")
        col_type.append(g.tag); col_size.append(size)
        col_xyz.append(f3(o.get("xyz") if o is not None else None)); col_rpy.append(f3(o.get("rpy") if o is not None else None))
    jname, jtype, parent, child, jxyz, jrpy, axis, lim = [], [], [], [], [], [], [], []
    for j in joints:
        jname.append(j.get("name")); jtype.append(j.get("type"))
        parent.append(j.find("parent").get("link")); child.append(j.find("child").get("link"))
        o = j.find("origin")
        jxyz.append(f3(o.get("xyz") if o is not None else None)); jrpy.append(f3(o.get("rpy") if o is not None else None))
        a = j.find("axis")
        axis.append(f3(a.get("xyz")) if a is not None else np.zeros(3))
        L = j.find("limit")
        lim.append([float(L.get(k, 0)) for k in ("lower", "upper", "effort", "velocity")] if L is not None else [0, 0, 0, 0])
    out = dict(link_name=np.array(lname), mass=np.array(mass), com=np.array(com), com_rpy=np.array(com_rpy), inertia=np.array(inertia),
               col_type=np.array(col_type), col_size=np.array(col_size), col_xyz=np.array(col_xyz), col_rpy=np.array(col_rpy),
               joint_name=np.array(jname), joint_type=np.array(jtype), parent=np.array(parent), child=np.array(child),
               joint_xyz=np.array(jxyz), joint_rpy=np.array(jrpy), axis=np.array(axis), limit=np.array(lim))
    np.savez_compressed(os.path.join(OUT, "urdf_tables.npz"), **out)
    print(f"urdf_tables.npz: {len(lname)} links (total mass {sum(mass):.5f} kg), {len(jname)} joints "
          f"({sum(t == 'revolute' for t in jtype)} revolute), collision primitives {sorted(set(col_type))}")


# --------------------------------------------------------------------------------------------- G11 randomizers
def gen_randomizers():
    """What the reference's randomizer stack (env_randomizer.py) writes into Bullet and into the motor model, draw by draw:
    ground friction, the three leg-link masses (same for every leg), payload mass + position, trunk mass, spring k and b."""
    from qs_amd.config import build_config
    from oracle.qso import Oracle
    from quadruped_spring.env.quadruped_gym_env import QuadrupedGymEnv

    def factory(dt, iters):
        cfg, _ = build_config(n_envs=1, time_step=dt, noise=False, env_randomizer_mode="NONE", task_env="JUMPING_IN_PLACE",
                              observation_space_mode="PPO_BASIC", enable_springs=True)
        cfg.solver_iters = iters
        cfg.randomizer_flags = 8
        return Oracle(cfg)

    FakeBulletClient.oracle_factory = factory
    out = {}
    np.random.seed(77)
    env = QuadrupedGymEnv(env_randomizer_mode="TEST_RANDOMIZER", task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC",
                          enable_springs=True, action_space_mode="SYMMETRIC", motor_control_mode="PD")
    client = env._pybullet_client
    n = 4000
    rows = np.zeros((n, 24))     # layout of QS_INFO_PARAMS: mu, k3, b3, rest3, kp3, kd3, m_trunk, m_leg3, m_pay, r_pay3
    leg_spread = 0.0
    for i in range(n):
        FakeBulletClient.rand_log = log = []
        client.mu_log.clear()
        env._env_randomizers.randomize_env()
        masses = {l: m for tag, l, m in [x for x in log if x[0] == "mass"]}
        pay = [x for x in log if x[0] == "payload"][0]
        con = [x for x in log if x[0] == "constraint"][0]
        assert con[1] == client.JOINT_FIXED and np.allclose(con[2], -pay[2])
        legs = np.array([[masses[b + j] for j in range(3)] for b in (2, 6, 10, 14)])
        leg_spread = max(leg_spread, np.abs(legs - legs[0]).max())
        k, b, _ = env.robot._motor_model._springs.get_spring_nominal_params()
        rows[i, 0] = client.mu_log[-1]
        rows[i, 1:4], rows[i, 4:7] = np.asarray(k, float)[:3], np.asarray(b, float)[:3]
        rows[i, 16], rows[i, 17:20], rows[i, 20], rows[i, 21:24] = masses[0], legs[0], pay[1], pay[2]
    FakeBulletClient.rand_log = None
    out["g11_params"] = rows
    out["g11_leg_spread"] = np.array(leg_spread)
    out["g11_total_mass"] = np.array(sum(_LINK_MASS.values()))
    print("g11: mu [%.3f, %.3f]  k_thigh [%.2f, %.2f]  m_pay [%.3f, %.3f]  m_trunk [%.3f, %.3f]  leg spread %.1e" % (
        rows[:, 0].min(), rows[:, 0].max(), rows[:, 2].min(), rows[:, 2].max(), rows[:, 20].min(), rows[:, 20].max(),
        rows[:, 16].min(), rows[:, 16].max(), leg_spread))
    np.savez_compressed(os.path.join(OUT, "randomizers.npz"), **out)


# --------------------------------------------------------------------------------------------- landing / go-to-rest wrappers
def gen_wrappers():
    """The reference's LandingWrapper / GoToRestWrapper around its own QuadrupedGymEnv (fake Bullet on oracle physics): every
    inner env.step the wrappers issue is logged, so the on-device phase machine can be checked step by step."""
    import importlib
    import gym
    from qs_amd.config import build_config
    from oracle.qso import Oracle
    from quadruped_spring.env.quadruped_gym_env import QuadrupedGymEnv
    from quadruped_spring.env.wrappers.landing_wrapper import LandingWrapper
    from quadruped_spring.env.wrappers.go_to_rest_wrapper import GoToRestWrapper
    from quadruped_spring.env.wrappers.landing_wrapper_2 import LandingWrapper2
    from quadruped_spring.env.wrappers.landing_wrapper_backflip import LandingWrapperBackflip
    from quadruped_spring.env.wrappers.landing_wrapper_backflip2 import LandingWrapperBackflip2
    from quadruped_spring.env.wrappers.landing_wrapper_continuous import LandingWrapperContinuous
    from quadruped_spring.env.wrappers.landing_wrapper_continuous2 import LandingWrapperContinuous2
    classes = dict(LANDING=LandingWrapper, GO_TO_REST=GoToRestWrapper, LANDING2=LandingWrapper2, LANDING_BACKFLIP=LandingWrapperBackflip,
                   LANDING_BACKFLIP2=LandingWrapperBackflip2, LANDING_CONTINUOUS=LandingWrapperContinuous,
                   LANDING_CONTINUOUS2=LandingWrapperContinuous2)

    class InnerLog(gym.Wrapper):
        def __init__(self, env):
            super().__init__(env)
            self.rows = []

        def step(self, a):
            kp = float(np.atleast_1d(self.env.robot._motor_model._kp)[0])
            ob, r, dn, info = self.env.step(a)
            self.rows.append((np.array(a, float), ob, r, dn, bool(info.get("TimeLimit.truncated", False)),
                              self.env._pybullet_client.o.get_state()[0].copy(), kp))
            return ob, r, dn, info

    cases = [
        dict(name="land_s1", wrapper="LANDING", task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True,
             enable_action_filter=True, action_space_mode="SYMMETRIC", motor_control_mode="PD", steps=130, jump_at=90),
        dict(name="land_s0", wrapper="LANDING", task_env="JUMPING_FORWARD", observation_space_mode="ARS_BASIC", enable_springs=False,
             enable_action_filter=True, action_space_mode="DEFAULT", motor_control_mode="PD", steps=110, jump_at=70),
        dict(name="rest_s1", wrapper="GO_TO_REST", task_env="JUMPING_IN_PLACE_PPO", observation_space_mode="PPO_BASIC_X", enable_springs=True,
             enable_action_filter=True, action_space_mode="SYMMETRIC", motor_control_mode="PD", steps=150, jump_at=80, ext=(-0.5, 0.6)),
        dict(name="rest_s0", wrapper="GO_TO_REST", task_env="JUMPING_FORWARD", observation_space_mode="PPO_BASIC", enable_springs=False,
             enable_action_filter=False, action_space_mode="SYMMETRIC_NO_HIP", motor_control_mode="PD", steps=260, jump_at=70),
        dict(name="land2_s1", wrapper="LANDING2", task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True,
             enable_action_filter=True, action_space_mode="SYMMETRIC", motor_control_mode="PD", steps=200, jump_at=80, ext=(-0.5, 0.6), land=True),
        dict(name="landbf_s1", wrapper="LANDING_BACKFLIP", task_env="BACKFLIP", observation_space_mode="PPO_BACKFLIP", enable_springs=True,
             enable_action_filter=True, action_space_mode="SYMMETRIC", motor_control_mode="PD", steps=170, jump_at=80),
        dict(name="landbf2_s1", wrapper="LANDING_BACKFLIP2", task_env="BACKFLIP", observation_space_mode="PPO_BACKFLIP", enable_springs=True,
             enable_action_filter=False, action_space_mode="SYMMETRIC", motor_control_mode="PD", steps=170, jump_at=80),
        dict(name="landc_s1", wrapper="LANDING_CONTINUOUS", task_env="CONTINUOUS_JUMPING_FORWARD", observation_space_mode="PPO_CONTINUOUS_JUMPING_FORWARD",
             enable_springs=True, enable_action_filter=True, action_space_mode="SYMMETRIC", motor_control_mode="PD", steps=260, jump_at=60,
             ext=(-0.5, 0.6), land=True),
        dict(name="landc2_s1", wrapper="LANDING_CONTINUOUS2", task_env="CONTINUOUS_JUMPING_FORWARD", observation_space_mode="PPO_CONTINUOUS_JUMPING_FORWARD",
             enable_springs=True, enable_action_filter=True, action_space_mode="SYMMETRIC", motor_control_mode="PD", steps=130, jump_at=60,
             ext=(-0.5, 0.6), land=True),
    ]
    out = {}
    for case in cases:
        name = case["name"]
        kw = {k: v for k, v in case.items() if k not in ("name", "steps", "jump_at", "wrapper", "ext", "land")}
        mod = importlib.import_module("quadruped_spring.go1.configs_go1_with_springs" if kw["enable_springs"]
                                      else "quadruped_spring.go1.configs_go1_without_springs")
        saved = {}
        for attr in dir(mod):
            if attr.endswith("_NOISE"):
                saved[attr] = getattr(mod, attr)
                setattr(mod, attr, np.zeros_like(np.asarray(saved[attr], float)))
        saved_upper = mod.RL_UPPER_ANGLE_JOINT.copy()   # the BACKFLIP tasks mutate the module-level limits in place

        def factory(dt, iters, kw=kw):
            cfg, _ = build_config(n_envs=1, time_step=dt, noise=False, env_randomizer_mode="NONE", **kw)
            cfg.solver_iters = iters
            cfg.randomizer_flags = 8
            return Oracle(cfg)

        FakeBulletClient.oracle_factory = factory
        np.random.seed(1234)
        rng = np.random.default_rng(11)
        log = InnerLog(QuadrupedGymEnv(env_randomizer_mode="GROUND_RANDOMIZER", **kw))
        env = classes[case["wrapper"]](log)
        acts = scripted_actions(rng, case["steps"], log.env.action_dim, case["jump_at"],
                                land=log.env.get_landing_action() if case["wrapper"] == "GO_TO_REST" or case.get("land") else None,
                                ext=case.get("ext", (-0.8, 1.0)))
        o = env.reset()
        keys = list(o.keys())
        flat = lambda ob: np.concatenate([np.atleast_1d(np.asarray(ob[k], float)).flatten() for k in keys])
        reset_obs, reset_at, outer_of_inner, outer_obs, outer_rew, outer_done = [flat(o)], [0], [], [], [], []
        mus = [log.env._pybullet_client.mu]
        for t in range(case["steps"]):
            n0 = len(log.rows)
            ob, r, dn, info = env.step(acts[t])
            outer_of_inner += [t] * (len(log.rows) - n0)
            outer_obs.append(flat(ob)); outer_rew.append(r); outer_done.append(dn)
            if dn:
                o = env.reset()
                reset_obs.append(flat(o)); reset_at.append(len(log.rows)); mus.append(log.env._pybullet_client.mu)
        rows = log.rows
        out[f"{name}_mu"] = np.array(mus)
        out[f"{name}_actions"] = acts
        out[f"{name}_outer_of_inner"] = np.array(outer_of_inner)
        out[f"{name}_inner_action"] = np.array([r[0] for r in rows])
        out[f"{name}_obs"] = np.array([flat(r[1]) for r in rows])
        out[f"{name}_rew"] = np.array([r[2] for r in rows], float)
        out[f"{name}_done"] = np.array([r[3] for r in rows])
        out[f"{name}_trunc"] = np.array([r[4] for r in rows])
        out[f"{name}_state"] = np.array([r[5] for r in rows])
        out[f"{name}_kp"] = np.array([r[6] for r in rows])
        out[f"{name}_outer_obs"], out[f"{name}_outer_rew"], out[f"{name}_outer_done"] = np.array(outer_obs), np.array(outer_rew, float), np.array(outer_done)
        out[f"{name}_reset_obs"], out[f"{name}_reset_at"] = np.array(reset_obs), np.array(reset_at)
        out[f"{name}_kwargs"] = np.array(repr(dict(kw, wrapper=case["wrapper"])))
        scripted = np.array([i > 0 and outer_of_inner[i] == outer_of_inner[i - 1] for i in range(len(rows))])
        print(f"wrapper {name}: outer={case['steps']} inner={len(rows)} scripted={int(scripted.sum())} episodes={len(reset_at)} "
              f"dones={int(np.sum(out[name + '_done']))} kp_set={sorted(set(out[name + '_kp']))}")
        for attr, v in saved.items():
            setattr(mod, attr, v)
        mod.RL_UPPER_ANGLE_JOINT[:] = saved_upper
    np.savez_compressed(os.path.join(OUT, "wrappers.npz"), **out)
    print("wrappers.npz:", len(out), "arrays")


# --------------------------------------------------------------------------------------------- DEMO tasks + reference-state initialisation
def gen_demo():
    """The imitation tasks (TaskJumpingDemo / TaskJumpingDemo2, task_base.py:169-220, 402-453; robot_tasks.py:222-247) and
    ReferenceStateInitializationWrapper on top of them.  The reference's repository does not hold the demonstrations its DEMO tasks
    load, so one is RECORDED here with the reference's own GetDemonstrationWrapper (a scripted jump under the matching plain task),
    saved under the file name the DEMO task expects in a scratch directory, and robot_tasks' package handle is pointed there."""
    import importlib
    import tempfile
    import types
    from qs_amd.config import build_config
    from oracle.qso import Oracle
    from quadruped_spring.env.quadruped_gym_env import QuadrupedGymEnv
    from quadruped_spring.env.wrappers.get_demonstration_wrapper import GetDemonstrationWrapper
    from quadruped_spring.env.wrappers.reference_state_initialization_wrapper import ReferenceStateInitializationWrapper
    import quadruped_spring.env.tasks.robot_tasks as rt

    scratch = tempfile.mkdtemp(prefix="qs_demo_")
    os.makedirs(os.path.join(scratch, "demonstrations"))
    rt.qs = types.SimpleNamespace(__file__=os.path.join(scratch, "__init__.py"))
    cases = [
        dict(name="demo_jip", task_env="JUMPING_IN_PLACE_DEMO", record_task="JUMPING_IN_PLACE", file="demo_list_jip_0", rsi=5, plain=1,
             kw=dict(observation_space_mode="PPO_BASIC", enable_springs=True, enable_action_filter=True, action_space_mode="SYMMETRIC", motor_control_mode="PD")),
        dict(name="demo_bf", task_env="BACKFLIP_DEMO", record_task="JUMPING_FORWARD", file="backflip-1", rsi=4, plain=1,
             kw=dict(observation_space_mode="PPO_BACKFLIP", enable_springs=False, enable_action_filter=False, action_space_mode="SYMMETRIC", motor_control_mode="PD")),
        dict(name="demo_jf12", task_env="JUMPING_FORWARD_DEMO", record_task="JUMPING_FORWARD", file="demo_list_jf_0", rsi=0, plain=2,
             kw=dict(observation_space_mode="PPO_BASIC_CONTACT", enable_springs=True, enable_action_filter=True, action_space_mode="DEFAULT", motor_control_mode="PD")),
        dict(name="demo_cjf", task_env="CONTINUOUS_JUMPING_FORWARD_DEMO", record_task="CONTINUOUS_JUMPING_FORWARD", file="continuous-jf-1", rsi=0, plain=2,
             kw=dict(observation_space_mode="PPO_CONTINUOUS_JUMPING_FORWARD", enable_springs=True, enable_action_filter=True, action_space_mode="SYMMETRIC_NO_HIP",
                     motor_control_mode="PD")),
    ]
    out = {}
    for case in cases:
        name, kw = case["name"], case["kw"]
        mod = importlib.import_module("quadruped_spring.go1.configs_go1_with_springs" if kw["enable_springs"]
                                      else "quadruped_spring.go1.configs_go1_without_springs")
        saved = {}
        for attr in dir(mod):
            if attr.endswith("_NOISE"):
                saved[attr] = getattr(mod, attr)
                setattr(mod, attr, np.zeros_like(np.asarray(saved[attr], float)))

        def factory(dt, iters, kw=kw):
            # physics only (the task lives in the reference's env on top); every link pushes back (the default)
            cfg, _ = build_config(n_envs=1, noise=False, env_randomizer_mode="NONE", task_env="NO_TASK", **dict(kw, time_step=dt))
            cfg.solver_iters = iters
            cfg.randomizer_flags = 8
            return Oracle(cfg)

        FakeBulletClient.oracle_factory = factory
        np.random.seed(99)
        rng = np.random.default_rng(23)
        # 1. record a demonstration: crouch, jump, land (get_demonstration_wrapper.py)
        rec = GetDemonstrationWrapper(QuadrupedGymEnv(env_randomizer_mode="GROUND_RANDOMIZER", task_env=case["record_task"], **kw),
                                      path=os.path.join(scratch, "demonstrations"), name=case["file"])
        rec.reset()
        d = rec.env.action_dim
        acts = scripted_actions(rng, 110, d, 70, ext=(-0.5, 0.6))
        for t in range(110):
            _, _, dn, _ = rec.step(acts[t])
            if dn:
                break
        with contextlib_redirect():
            rec.save_demo()
        demo = np.load(os.path.join(scratch, "demonstrations", case["file"] + ".npy"))
        L = demo.shape[0]
        # 2. the DEMO task on it: plain episodes (settle, counter 0) first, then reference-state initialisation
        env = QuadrupedGymEnv(env_randomizer_mode="GROUND_RANDOMIZER", task_env=case["task_env"], **kw)
        assert env.task.demo_length == L
        wrapped = ReferenceStateInitializationWrapper(env)
        assert wrapped.enable_wrapper == (case["task_env"] != "CONTINUOUS_JUMPING_FORWARD_DEMO")
        if wrapped.enable_wrapper:
            wrapped._rng.seed(7)
        keys = None
        A, O, R, D, TR, S, C = [], [], [], [], [], [], []
        reset_at, reset_el, reset_obs, mus = [], [], [], []
        for ep in range(case["plain"] + case["rsi"]):
            if ep < case["plain"]:
                o = env.reset(); el = -1
            else:
                o = wrapped.reset(); el = wrapped.random_el
            keys = keys or list(o.keys())
            flat = lambda ob: np.concatenate([np.atleast_1d(np.asarray(ob[k], float)).flatten() for k in keys])
            client = env._pybullet_client
            reset_at.append(len(A)); reset_el.append(el); reset_obs.append(flat(o)); mus.append(client.mu)
            assert env.task.demo_counter == max(el, 0)
            while True:
                a = demo[env.task.demo_counter, :d] + (0.02 + 0.1 * (ep % 3)) * rng.standard_normal(d)
                ob, r, dn, info = env.step(a)
                A.append(a); O.append(flat(ob)); R.append(r); D.append(dn); TR.append(bool(info.get("TimeLimit.truncated", False)))
                S.append(client.o.get_state()[0].copy()); C.append(env.task.demo_counter)
                if dn:
                    break
        out[f"{name}_demo"], out[f"{name}_kwargs"] = demo, np.array(repr(dict(kw, task_env=case["task_env"])))
        out[f"{name}_actions"], out[f"{name}_obs"], out[f"{name}_rew"] = np.array(A), np.array(O), np.array(R, float)
        out[f"{name}_done"], out[f"{name}_trunc"], out[f"{name}_state"], out[f"{name}_counter"] = np.array(D), np.array(TR), np.array(S), np.array(C)
        out[f"{name}_reset_at"], out[f"{name}_reset_el"], out[f"{name}_reset_obs"], out[f"{name}_mu"] = np.array(reset_at), np.array(reset_el), np.array(reset_obs), np.array(mus)
        ends = [int(C[i]) for i in range(len(D)) if D[i]]
        print(f"demo {name}: L={L} d={d} episodes={len(reset_at)} steps={len(A)} reset_el={reset_el} counters at done={ends} "
              f"demo ends={sum(c == L for c in ends)} rew sum={np.sum(R):.3f}")
        for attr, v in saved.items():
            setattr(mod, attr, v)
    np.savez_compressed(os.path.join(OUT, "demo.npz"), **out)
    print("demo.npz:", len(out), "arrays")


def contextlib_redirect():
    import contextlib
    import io
    return contextlib.redirect_stdout(io.StringIO())


# --------------------------------------------------------------------------------------------- G13 Hopf CPG
def gen_cpg():
    from quadruped_spring.hopf_network import HopfNetwork
    out = {}
    import contextlib
    import io
    for gait, (ws, wst) in {"TROT": (16.0, 4.0), "WALK": (24.0, 25.0), "PACE": (20.0, 20.0), "BOUND": (10.0, 40.0)}.items():
        np.random.seed(7)
        with contextlib.redirect_stdout(io.StringIO()):
            cpg = HopfNetwork(gait=gait, omega_swing=ws * np.pi, omega_stance=wst * np.pi, time_step=0.001)
        X0 = cpg.X.copy()
        xs, zs, Xs = [], [], []
        for _ in range(2000):
            x, z = cpg.update()
            xs.append(x); zs.append(z); Xs.append(cpg.X.copy())
        out[f"g13_{gait}_X0"], out[f"g13_{gait}_X"] = X0, np.array(Xs)
        out[f"g13_{gait}_x"], out[f"g13_{gait}_z"] = np.array(xs), np.array(zs)
        out[f"g13_{gait}_params"] = np.array([ws * np.pi, wst * np.pi, cpg._mu, cpg._des_step_len, cpg._robot_height])
        out[f"g13_{gait}_phi"] = cpg.PHI.copy()
        out[f"g13_{gait}_shape"] = np.array([cpg._ground_clearance, cpg._ground_penetration, cpg._coupling_strength, 50.0])
    np.savez_compressed(os.path.join(OUT, "cpg.npz"), **out)
    print("cpg.npz:", len(out), "arrays")


# --------------------------------------------------------------------------------------------- contact classification rule
def gen_contacts():
    """The reference's GetContactInfo (quadruped.py:224-258) applied to contact lists that the oracle produces in scenarios with feet,
    non-foot links, the payload block and link-link contacts; the fixture holds the lists and what the reference made of them."""
    from qs_amd.config import build_config
    from oracle.qso import Oracle
    from quadruped_spring.env.quadruped import Quadruped
    from scipy.spatial.transform import Rotation as Rot
    rob = object.__new__(Quadruped)
    rob._calf_ids, rob._thigh_ids, rob._foot_link_ids = [4, 8, 12, 16], [3, 7, 11, 15], [5, 9, 13, 17]
    kw = dict(task_env="NO_TASK", observation_space_mode="ENCODER", enable_springs=True, enable_action_filter=False, isRLGymInterface=False,
              motor_control_mode="TORQUE", noise=False)
    rng = np.random.default_rng(11)
    lists, outs, states, params, warms = [], [], [], [], []
    for k in range(160):
        cfg, _ = build_config(n_envs=1, env_randomizer_mode="MASS_RANDOMIZER" if k % 4 == 3 else "NONE", seed=k, **kw)
        o = Oracle(cfg)
        o.reset()
        s = o.get_state()
        s[0, 7:] = 0
        if k % 4 == 0:      # standing / crouching on the feet
            s[0, 2] = rng.uniform(0.12, 0.32); s[0, 13:25] = np.tile([0.0, rng.uniform(0.6, 1.3), rng.uniform(-2.5, -1.3)], 4)
        elif k % 4 == 1:    # lying in random orientations
            s[0, 2] = rng.uniform(0.05, 0.15); s[0, 3:7] = Rot.random(random_state=k).as_quat()
            s[0, 13:25] = rng.uniform(np.tile([-0.5, -0.6, -2.7], 4), np.tile([0.5, 2.0, -0.9], 4))
        elif k % 4 == 2:    # legs tangled in the air
            s[0, 2] = 1.0
            q = np.tile([0.0, 0.8, -1.6], 4) + 0.25 * rng.normal(size=12)
            q[0], q[3] = 0.55, -0.55
            s[0, 13:25] = q
        else:               # payload hanging low
            p = o.get_info(6); p[0, 21:24] = [rng.uniform(-0.1, 0.1), 0.0, -0.1]; o.set_params(5, p)
            s[0, 2] = rng.uniform(0.13, 0.2); s[0, 13:25] = np.tile([0.0, 1.2, -2.4], 4)
        o.set_state(s)
        for _ in range(2):
            o.phys_step(0, np.zeros(12))
        state_in, warm_in = o.get_state()[0].copy(), o.get_info(0)[0].copy() * cfg.dt     # state (and contact warm start) before the recorded substep
        o.phys_step(0, np.zeros(12))
        cl = o.contacts(0)
        rob._pybullet_client = types.SimpleNamespace(getContactPoints=lambda cl=cl: [(0, a, b, c, d, (0, 0, 0), (0, 0, 0), (0, 0, 1), x, f) for (a, b, c, d, x, f) in cl])
        nv, ni, ff, fb = rob.GetContactInfo()
        row = np.full((64, 6), np.nan); row[:len(cl)] = np.array(cl, float).reshape(-1, 6)
        lists.append(row); outs.append([nv, ni] + list(ff) + list(fb))
        states.append(state_in); params.append(o.get_info(6)[0]); warms.append(warm_in)
        assert int(o.get_info(5)[0, 0]) == ni, (k, ni, o.get_info(5))       # the oracle's own count follows the same rule
    out = dict(contacts=np.array(lists), reference=np.array(outs, float), states=np.array(states), params=np.array(params), warm=np.array(warms))
    np.savez_compressed(os.path.join(OUT, "contacts.npz"), **out)
    kinds = np.array(outs)
    print("contacts.npz:", len(lists), "scenarios;", int((kinds[:, 1] > 0).sum()), "with invalid contacts,", int((kinds[:, 0] > 0).sum()), "with feet on the ground")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    install_shims()
    logging.disable(logging.CRITICAL)
    which = sys.argv[1:] or ["stateless", "rewards", "traces", "cpg", "wrappers", "randomizers", "urdf", "rsi", "demo", "contacts"]
    for w in which:
        {"stateless": gen_stateless, "rewards": gen_rewards, "traces": gen_traces, "cpg": gen_cpg, "wrappers": gen_wrappers,
         "randomizers": gen_randomizers, "urdf": gen_urdf, "rsi": gen_rsi, "demo": gen_demo, "contacts": gen_contacts}[w]()
