"""Plain-data configuration handed across the C ABI (include/qs_amd.h `qs_config`).

`build_config` turns the reference's constructor keywords (quadruped_spring/env/quadruped_gym_env.py:52-70)
and string registries (task_collection.py:19-37, sensor_collection.py:92-105, control_interface/collection.py:33,49,
env_randomizer_collection.py:15-21) into numbers.
"""
import ctypes as C
import math

import numpy as np

from . import go1_config

MAX_SENSORS = 16
MAX_OBS = 64

ACTION_SPACE_MODES = {"DEFAULT": (0, 12), "SYMMETRIC": (1, 6), "SYMMETRIC_NO_HIP": (2, 4),
                      "CPG": (3, 5)}  # build extension for BASELINE.json configs[4]: the policy drives a Hopf CPG

# hopf_network.py:74-115
_PI = math.pi
CPG_GAITS = {
    "TROT": [[0, -_PI, -_PI, 0], [_PI, 0, 0, _PI], [_PI, 0, 0, _PI], [0, -_PI, -_PI, 0]],
    "WALK": [[0, -_PI, -_PI / 2, _PI / 2], [_PI, 0, _PI / 2, 3 * _PI / 2], [_PI / 2, -_PI / 2, 0, _PI], [-_PI / 2, -3 * _PI / 2, -_PI, 0]],
    "BOUND": [[0, 0, -_PI, -_PI], [0, 0, -_PI, -_PI], [_PI, _PI, 0, 0], [_PI, _PI, 0, 0]],
    "PACE": [[0, -_PI, 0, -_PI], [_PI, 0, _PI, 0], [0, -_PI, 0, -_PI], [_PI, 0, _PI, 0]],
}
# action in [-1, 1]^5 -> (omega_swing, omega_stance, mu, des_step_len, robot_height); ranges chosen around the values the
# reference's CPG driver uses (hopf_network.py:36-45, 195-206)
CPG_LO = [2 * _PI, 2 * _PI, 0.5, 0.0, 0.15]
CPG_HI = [40 * _PI, 40 * _PI, 2.5, 0.10, 0.30]
MOTOR_CONTROL_MODES = {"PD": 0, "CARTESIAN_PD": 1, "TORQUE": 2}
TASKS = {
    "NO_TASK": 0,
    "JUMPING_IN_PLACE": 1,
    "JUMPING_FORWARD": 2,
    "CONTINUOUS_JUMPING_FORWARD": 3,
    "CONTINUOUS_JUMPING_FORWARD2": 4,
    "JUMPING_IN_PLACE_PPO": 5,
    "JUMPING_FORWARD_PPO": 6,
    "BACKFLIP": 7,
    "JUMPING_IN_PLACE_PPO_HP": 8,
    "JUMPING_FORWARD_PPO_HP": 9,
    "BACKFLIP_PPO": 10,
    "CONTINUOUS_JUMPING_FORWARD3": 11,
    "CONTINUOUS_JUMPING_FORWARD_PPO": 12,
    # imitation tasks (task_base.py:169-220, 402-453): the demonstration comes in through the `demo` keyword
    "JUMPING_IN_PLACE_DEMO": 13,
    "JUMPING_FORWARD_DEMO": 14,
    "BACKFLIP_DEMO": 15,
    "CONTINUOUS_JUMPING_FORWARD_DEMO": 16,
}
# the file each DEMO task of the reference loads from <package>/demonstrations/ (robot_tasks.py:224, 230, 236, 246); the
# reference's repository does not hold them, so here the rows are given with `demo=` (array, or the path of such a .npy)
DEMO_FILES = {"JUMPING_IN_PLACE_DEMO": "demo_list_jip_0.npy", "JUMPING_FORWARD_DEMO": "demo_list_jf_0.npy",
              "BACKFLIP_DEMO": "backflip-1.npy", "CONTINUOUS_JUMPING_FORWARD_DEMO": "continuous-jf-1.npy"}

# sensor ids (include/qs_amd.h QS_SENS_*): name, dim, (high, low, noise) attribute names, reference obs-dict key
SENSORS = {
    "JointPosition": (0, 12, "JOINT_ANGLES", "Encoder"),
    "JointVelocity": (1, 12, "JOINT_VELOCITIES", "JointVelocity"),
    "Pitch": (2, 1, "PITCH", "Pitch"),
    "Height": (3, 1, "HEIGHT", "Height"),
    "BaseHeightVelocity": (4, 1, "VEL_LIN[2]", "Base Linear Velocity z direction"),
    "Landing": (5, 1, None, "is landing"),
    "Jumping": (6, 1, None, "is jumping"),
    "PitchRate": (7, 1, "PITCH_RATE", "Pitch rate"),
    "VelocityX": (8, 1, "VEL_LIN[0]", "Base Height Velocity X"),
    "BooleanContact": (9, 4, "CONTACT_BOOL", "BoolContatc"),
    "LinearVelocity": (10, 3, "VEL_LIN", "Base Linear Velocity"),
    "AngularVelocity": (11, 3, "VEL_ANG", "Base Angular Velocity"),
    "FeetPostion": (12, 12, "FEET_POS", "FeetPosition"),
    "FeetVelocity": (13, 12, "FEET_VEL", "FeetVelocity"),
    "PitchBackFlip": (14, 1, "PITCH", "Pitch-BackFlip"),
    "OrientationRPY": (15, 3, "ORIENT_RPY", "Orientation Roll Pitch Yaw"),
    "Quaternion": (16, 4, "QUATERNION", "Quaternion"),
}
# sensor_collection.py:18-90
SENSOR_BUNDLES = {
    "ENCODER": ["JointPosition", "JointVelocity"],
    "ENCODER_2": ["LinearVelocity", "AngularVelocity", "JointPosition", "JointVelocity"],
    "CARTESIAN_NO_IMU": ["FeetPostion", "FeetVelocity"],
    "ARS_BASIC": ["JointPosition", "JointVelocity", "Pitch", "Height", "BaseHeightVelocity"],
    "ARS_SENSOR": ["JointPosition", "JointVelocity", "Pitch", "PitchRate", "Height", "BaseHeightVelocity"],
    "LANDING_SENSOR": ["JointPosition", "JointVelocity", "Pitch", "PitchRate", "Height", "BaseHeightVelocity", "Landing"],
    "PPO_BASIC": ["JointPosition", "JointVelocity", "Pitch", "Height", "BaseHeightVelocity", "Landing"],
    "PPO_BASIC_X": ["JointPosition", "JointVelocity", "Pitch", "Height", "BaseHeightVelocity", "VelocityX", "Landing"],
    "PPO_BASIC_CONTACT": ["JointPosition", "JointVelocity", "Pitch", "Height", "BaseHeightVelocity", "Landing", "BooleanContact"],
    "ARS_BACKFLIP": ["JointPosition", "JointVelocity", "Height", "BaseHeightVelocity", "PitchBackFlip"],
    "PPO_BACKFLIP": ["JointPosition", "JointVelocity", "Height", "BaseHeightVelocity", "PitchBackFlip", "Landing"],
    "PPO_CONTINUOUS_JUMPING_FORWARD": ["JointPosition", "JointVelocity", "Height", "BaseHeightVelocity", "Pitch", "Landing", "Jumping"],
}
# env_randomizer_collection.py:15-21 (curriculum variant is dead code in the reference)
RAND_GROUND, RAND_MASSES, RAND_SPRINGS = 1, 2, 4
RANDOMIZERS = {
    "GROUND_RANDOMIZER": RAND_GROUND,
    "MASS_RANDOMIZER": RAND_GROUND | RAND_MASSES,
    "SPRING_RANDOMIZER": RAND_GROUND | RAND_SPRINGS,
    "TEST_RANDOMIZER": RAND_GROUND | RAND_MASSES | RAND_SPRINGS,
    "TEST_RANDOMIZER_CURRICULUM": RAND_GROUND | RAND_MASSES | RAND_SPRINGS,
    "NONE": 0,  # build extension: BASELINE.json config 2 ("flat ground, mu = 1")
}

EPISODE_LENGTH = 10  # gym_env.py:35
OBSERVATION_EPS = 0.01  # gym_env.py:34


class QsConfig(C.Structure):
    """Mirror of `qs_config` (include/qs_amd.h) and `qso_config` (oracle/qso.h)."""

    _fields_ = [
        ("n_envs", C.c_int32), ("action_dim", C.c_int32), ("action_space_mode", C.c_int32),
        ("motor_control_mode", C.c_int32), ("symm_idx", C.c_int32), ("rl_interface", C.c_int32),
        ("task", C.c_int32), ("n_sensors", C.c_int32), ("sensors", C.c_int32 * MAX_SENSORS),
        ("obs_dim", C.c_int32), ("enable_springs", C.c_int32), ("enable_filter", C.c_int32),
        ("enable_interp", C.c_int32), ("action_repeat", C.c_int32), ("solver_iters", C.c_int32),
        ("settle_steps", C.c_int32), ("max_sim_steps", C.c_int32), ("randomizer_flags", C.c_int32),
        ("noise_enabled", C.c_int32), ("auto_reset", C.c_int32), ("reset_lookahead", C.c_int32), ("env_id_offset", C.c_int32), ("wrapper_mode", C.c_int32),
        ("seed", C.c_uint64), ("dt", C.c_double), ("filt_b", C.c_double * 3), ("filt_a", C.c_double * 3), ("gravity", C.c_float),
        ("kp", C.c_float * 3), ("kd", C.c_float * 3), ("tau_max", C.c_float * 3),
        ("cmd_lo", C.c_float * 12), ("cmd_hi", C.c_float * 12),
        ("settle_cmd", C.c_float * 12), ("settle_action", C.c_float * 12),
        ("spring_k", C.c_float * 3), ("spring_b", C.c_float * 3), ("spring_rest", C.c_float * 3),
        ("fallen_height", C.c_float), ("leg_len", C.c_float * 3),
        ("contact_erp", C.c_float), ("joint_erp", C.c_float), ("warmstart", C.c_float), ("vel_cap", C.c_float),
        ("obs_noise_std", C.c_float * MAX_OBS), ("task_p", C.c_float * 16), ("contact_slop", C.c_float), ("body_contacts", C.c_int32), ("self_collision", C.c_int32), ("info_fields", C.c_int32), ("payload_soft", C.c_int32), ("support_margin", C.c_float), ("reserved_f", C.c_float * 2),
        ("cpg_phi", C.c_float * 16), ("cpg_lo", C.c_float * 5), ("cpg_hi", C.c_float * 5),
        ("cpg_clearance", C.c_float), ("cpg_penetration", C.c_float), ("cpg_coupling", C.c_float), ("cpg_alpha", C.c_float),
        ("solver_residual_threshold", C.c_float), ("friction_cone", C.c_int32),
        ("landing_action", C.c_float * 12), ("landing_kp", C.c_float), ("landing_kd", C.c_float),
        ("rest_kp", C.c_float), ("rest_kd", C.c_float), ("rest_time", C.c_float), ("reserved_h", C.c_float * 3), ("unit_inertia", (C.c_float * 6) * 4),
    ]


def decode_flags(flags):
    """The last column of a fused result row is done + 2 * truncated (include/qs_amd.h, qs_step_fused): 0 running, 1 terminated,
    3 truncated (gym_env.py:245-246: truncation is a done without termination).  Returns (done, truncated) as booleans; works on
    torch tensors and numpy arrays alike.  The one decoder for QuadrupedVecEnv and ShardedVecEnv."""
    return flags > 0.5, flags > 2.5


def butter2_lowpass(fc, fs):
    """Coefficients of scipy.signal.butter(2, [fc/(fs/2)]) (utils/action_filter.py:191-213) in closed form
    (bilinear transform of the 2nd-order Butterworth prototype with pre-warping)."""
    k = math.tan(math.pi * fc / fs)
    q = math.sqrt(2.0)
    norm = 1.0 / (1.0 + q * k + k * k)
    b0 = k * k * norm
    return np.array([b0, 2 * b0, b0]), np.array([1.0, 2 * (k * k - 1) * norm, (1 - q * k + k * k) * norm])


def _lookup(table, key, what):
    if key not in table:
        raise KeyError(f"the {what} {key} is not implemented yet.")  # base_collection.py:8-12 prints the same text
    return table[key]


def _attr(cfg, spec, suffix):
    if "[" in spec:
        base, idx = spec[:-1].split("[")
        return np.atleast_1d(np.asarray(getattr(cfg, f"{base}_{suffix}"))[int(idx)])
    return np.atleast_1d(np.asarray(getattr(cfg, f"{spec}_{suffix}"), dtype=np.float64))


def sensor_layout(robot_config, observation_space_mode):
    """ids, names, per-element (high, low, noise std) of a bundle (sensor.py:71-136, robot_sensors.py)."""
    names = _lookup(SENSOR_BUNDLES, observation_space_mode, "sensor package")
    ids, keys, dims, high, low, std = [], [], [], [], [], []
    for n in names:
        sid, dim, spec, key = SENSORS[n]
        ids.append(sid); keys.append(key); dims.append(dim)
        if spec is None:  # Landing / Jumping: [0, 1], no noise (robot_sensors.py:141-188)
            high.append(np.ones(1)); low.append(np.zeros(1)); std.append(np.zeros(1))
        else:
            high.append(_attr(robot_config, spec, "HIGH")); low.append(_attr(robot_config, spec, "LOW"))
            s = _attr(robot_config, spec, "NOISE")
            std.append(s if np.all(s > 0) else np.zeros_like(s))  # sensor.py:25-32: all-or-nothing
    return dict(ids=ids, keys=keys, dims=dims, high=np.concatenate(high), low=np.concatenate(low), std=np.concatenate(std))


def scale_action_to_command(a12, lo, hi):
    """interface_base.py:84-90."""
    a = np.clip(a12, -1, 1)
    return np.clip(lo + 0.5 * (a + 1) * (hi - lo), lo, hi)


def scale_command_to_action(cmd, lo, hi):
    """interface_base.py:92-100."""
    c = np.clip(cmd, lo, hi)
    return np.clip(-1 + 2 * (c - lo) / (hi - lo), -1, 1)


def to_default_action_space(a, mode, symm_idx):
    """action_interface.py:14-15, :29-39, :58-65."""
    a = np.asarray(a, dtype=np.float64)
    if mode == "DEFAULT":
        return a.copy()
    if mode == "SYMMETRIC":
        fr, rr = a[0:3].copy(), a[3:6].copy()
        fl, rl = fr.copy(), rr.copy()
        fl[symm_idx] = -fr[symm_idx]
        rl[symm_idx] = -rr[symm_idx]
        return np.concatenate((fr, fl, rr, rl))
    fr = np.insert(a[0:2], symm_idx, 0)
    rr = np.insert(a[2:4], symm_idx, 0)
    return np.concatenate((fr, fr, rr, rr))


def to_actual_action_space(a12, mode, symm_idx):
    """action_interface.py:17-18, :41-44, :67-74."""
    if mode == "DEFAULT":
        return a12.copy()
    fr, rr = a12[0:3], a12[6:9]
    if mode == "SYMMETRIC":
        return np.concatenate((fr, rr))
    return np.concatenate((np.delete(fr, symm_idx), np.delete(rr, symm_idx)))


# Link inertials and collision primitives of go1.urdf that the mass-to-inertia rule needs (FR leg; mirror signs are applied on the
# device).  mass, inertia (ixx, ixy, ixz, iyy, iyz, izz) about the centre of mass, collision box extents in link axes (the hip's
# cylinder enters Bullet's compound AABB as its bounding box: btCylinderShape::getAabb).  go1.urdf lines: trunk :74-85, hip :121-136,
# thigh :173-188, calf :200-216.  Pinned against tests/golden/urdf_tables.npz by tests/test_host_cpu.py.
URDF_LINKS = {
    "hip": (0.591, (0.000374268192, -3.6844422e-05, -9.86754e-07, 0.000635923669, 1.172894e-06, 0.000457647394), (0.092, 0.04, 0.092)),
    "thigh": (0.92, (0.005851561134, -1.783284e-06, 0.000328291374, 0.005596155105, -2.1430713e-05, 0.00107157026), (0.034, 0.0245, 0.213)),
    "calf": (0.131, (0.002939186297, 1.440899e-06, -0.00010535955, 0.00295576935, -2.4397752e-05, 3.0273372e-05), (0.016, 0.016, 0.213)),
    "trunk": (5.204, (0.0168352186, 0.0004636141, 0.0002367952, 0.0656071082, 3.6671e-05, 0.0742720659), (0.3762, 0.0935, 0.114)),
}
# sign of each FR tensor entry relative to the magnitudes the device mirrors per leg (csrc/qs_core.h build_model): the device
# tables hold |FR| for the hip and the FR value itself elsewhere
_DEVICE_SIGN = {"hip": (1, -1, -1, 1, 1, 1), "thigh": (1, -1, 1, 1, -1, 1), "calf": (1, 1, 1, 1, 1, 1), "trunk": (1, 1, 1, 1, 1, 1)}


def unit_inertia_table(rule):
    """[4][6] inertia per unit mass of hip, thigh, calf, trunk (device sign convention) under a mass-to-inertia rule.

    "scale": I_urdf / m_urdf, i.e. the tensor scales with the mass.
    "collision_shape": Bullet's changeDynamics(mass=m) on a multibody link (quadruped.py:761, 776) replaces the link's inertia by
    collisionShape->calculateLocalInertia(m); the shape is the btCompoundShape the URDF importer wraps every link's collision in, whose
    calculateLocalInertia is the solid-box formula on the compound's AABB, m/12 (ly^2 + lz^2, lx^2 + lz^2, lx^2 + ly^2), taken in the
    link's inertial frame = the principal axes of the URDF tensor (the importer diagonalises it).  Rotated back to link axes here."""
    out = np.zeros((4, 6))
    for k, name in enumerate(("hip", "thigh", "calf", "trunk")):
        m0, i6, ext = URDF_LINKS[name]
        I = np.array([[i6[0], i6[1], i6[2]], [i6[1], i6[3], i6[4]], [i6[2], i6[4], i6[5]]])
        if rule == "scale":
            T = I / m0
        elif rule == "collision_shape":
            _, R = np.linalg.eigh(I)                       # columns: principal axes in link coordinates
            half = 0.5 * np.asarray(ext)
            l = 2.0 * (np.abs(R.T) @ half)                 # AABB extents of the (link-axis aligned) box seen from the principal frame
            d = np.array([l[1] ** 2 + l[2] ** 2, l[0] ** 2 + l[2] ** 2, l[0] ** 2 + l[1] ** 2]) / 12.0
            T = R @ np.diag(d) @ R.T
        else:
            raise KeyError(f"the mass inertia rule {rule} is not implemented yet.")
        t6 = np.array([T[0, 0], T[0, 1], T[0, 2], T[1, 1], T[1, 2], T[2, 2]])
        out[k] = t6 * np.array(_DEVICE_SIGN[name])
    return out


def build_config(
    n_envs=1,
    isRLGymInterface=True,
    time_step=0.001,
    action_repeat=10,
    motor_control_mode="PD",
    task_env="NO_TASK",
    observation_space_mode="ENCODER",
    action_space_mode="SYMMETRIC",
    enable_springs=False,
    enable_action_interpolation=False,
    enable_action_filter=False,
    env_randomizer_mode="GROUND_RANDOMIZER",
    seed=0,
    noise=True,
    auto_reset=False,
    settle_steps=2500,
    env_id_offset=0,
    cpg_gait="BOUND",
    solver_residual_threshold=1e-7,
    wrapper=None,
    robot_config=None,
    demo=None,
    contact_erp=0.08,
    contact_slop=1e-5,
    joint_erp=0.2,
    warmstart=0.1,
    friction_model="cone",
    body_contacts=True,
    support_margin=0.5,
    self_collision=True,
    mass_inertia_rule="collision_shape",
    info_fields=True,
    payload="weld",
    on_rack=False,
    render=False,
    camera_mode="CLASSIC",
    curriculum_level=0.0,
    verbose=0,
):
    """Returns (QsConfig, meta). `meta` keeps the python-side view (names, limits, robot config).

    The keywords are the reference constructor's (quadruped_gym_env.py:52-70) plus this build's own; anything else is a TypeError, as it
    is there (a misspelt `enable_spring=True` must not silently simulate without springs).  Of the reference's, `camera_mode`,
    `curriculum_level` and `verbose` are accepted and unused (rendering / dead curriculum code), `on_rack` and `render` must be False."""
    if on_rack or render:
        raise NotImplementedError("on_rack / render need the PyBullet GUI path, which this build does not provide")
    if motor_control_mode == "TORQUE" and isRLGymInterface:
        # gym_env.py:167-168
        raise ValueError(f"the motor control mode {motor_control_mode} not" "implemented yet for RL Gym interface.")
    rc = robot_config if robot_config is not None else go1_config.make_config(enable_springs)
    mode_id, action_dim = _lookup(ACTION_SPACE_MODES, action_space_mode, "action space mode")
    motor_id = _lookup(MOTOR_CONTROL_MODES, motor_control_mode, "motor control mode")
    task_id = _lookup(TASKS, task_env, "task")
    rand = _lookup(RANDOMIZERS, env_randomizer_mode, "env randomizer")
    if motor_control_mode == "PD" and task_env == "BACKFLIP":
        # motor_interface.py:17-22 mutates the config arrays in place BEFORE the sensors read their limits
        # (gym_env.py:119 precedes :127), so the JointPosition limits widen as well
        for i in (7, 10):
            rc.RL_UPPER_ANGLE_JOINT[i] = math.pi / 2
    lay = sensor_layout(rc, observation_space_mode)

    cfg = QsConfig()
    cfg.n_envs, cfg.action_dim, cfg.action_space_mode, cfg.motor_control_mode = n_envs, action_dim, mode_id, motor_id
    cfg.rl_interface, cfg.task = int(bool(isRLGymInterface)), task_id
    if not isRLGymInterface:
        cfg.action_dim = action_dim = 12  # raw motor commands (gym_env.py:212-214)
    # limits and poses (motor_interface.py:9-32, :50-63, :94-100)
    if action_space_mode == "CPG" and motor_control_mode != "PD":
        raise ValueError("the CPG action layer drives joint PD targets: motor_control_mode must be PD")
    if motor_control_mode == "PD":
        lo, hi, symm = rc.RL_LOWER_ANGLE_JOINT, rc.RL_UPPER_ANGLE_JOINT, 0
        init_pose, landing_pose = rc.INIT_MOTOR_ANGLES, rc.ANGLE_LANDING_POSE
    elif motor_control_mode == "CARTESIAN_PD":
        lo, hi, symm = rc.RL_LOWER_CARTESIAN_POS, rc.RL_UPPER_CARTESIAN_POS, 1
        init_pose, landing_pose = rc.NOMINAL_FOOT_POS_LEG_FRAME, rc.CARTESIAN_LANDING_POSE
    else:
        lo, hi, symm = -rc.TORQUE_LIMITS, rc.TORQUE_LIMITS, 0
        init_pose, landing_pose = np.zeros(12), np.zeros(12)
    cfg.symm_idx = symm
    cfg.n_sensors = len(lay["ids"])
    for i, s in enumerate(lay["ids"]):
        cfg.sensors[i] = s
    cfg.obs_dim = int(sum(lay["dims"]))
    cfg.enable_springs, cfg.enable_filter = int(bool(enable_springs)), int(bool(enable_action_filter))
    cfg.enable_interp = int(bool(enable_action_interpolation))
    cfg.action_repeat = int(action_repeat)
    cfg.solver_iters = int(300 / action_repeat)  # gym_env.py:113
    cfg.settle_steps = int(settle_steps) if isRLGymInterface else 1500  # gym_env.py:115 / control_interface/utils.py:31
    # gym_env.py:245: done when sim_step_counter * dt > 10 (evaluated in python float arithmetic, like the reference)
    n = max(0, int(EPISODE_LENGTH / time_step) - 2)
    while n * time_step <= EPISODE_LENGTH:
        n += 1
    cfg.max_sim_steps = n - 1
    cfg.randomizer_flags, cfg.noise_enabled, cfg.auto_reset = rand, int(bool(noise)), int(bool(auto_reset))
    cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    cfg.env_id_offset = int(env_id_offset)
    cfg.dt, cfg.gravity = time_step, 9.8  # gym_env.py:309
    for i in range(3):
        cfg.kp[i], cfg.kd[i], cfg.tau_max[i] = rc.MOTOR_KP[i], rc.MOTOR_KD[i], rc.RL_TORQUE_LIMITS[i]
        cfg.spring_k[i], cfg.spring_b[i], cfg.spring_rest[i] = rc.SPRINGS_STIFFNESS[i], rc.SPRINGS_DAMPING[i], rc.SPRINGS_REST_ANGLE[i]
    cfg.leg_len[0], cfg.leg_len[1], cfg.leg_len[2] = rc.HIP_LINK_LENGTH, rc.THIGH_LINK_LENGTH, rc.CALF_LINK_LENGTH
    for i in range(12):
        cfg.cmd_lo[i], cfg.cmd_hi[i] = lo[i], hi[i]
    # settle reference -> action -> command round trip (interface_base.py:68-73, 182-200)
    for i, row in enumerate(CPG_GAITS[cpg_gait]):
        for j, v in enumerate(row):
            cfg.cpg_phi[4 * i + j] = v
    for i in range(5):
        cfg.cpg_lo[i], cfg.cpg_hi[i] = CPG_LO[i], CPG_HI[i]
    cfg.cpg_clearance, cfg.cpg_penetration, cfg.cpg_coupling, cfg.cpg_alpha = 0.05, 0.01, 1.0, 50.0  # hopf_network.py:42-43, 35, 142
    if isRLGymInterface and action_space_mode == "CPG":
        settle_cmd = scale_action_to_command(scale_command_to_action(np.asarray(init_pose, float), lo, hi), lo, hi)
        settle_action = np.zeros(5)
    elif isRLGymInterface:
        aspace = action_space_mode
        ref_action = to_actual_action_space(scale_command_to_action(np.asarray(init_pose, float), lo, hi), aspace, symm)
        settle_scaled = scale_action_to_command(to_default_action_space(ref_action, aspace, symm), lo, hi)
        if motor_control_mode == "CARTESIAN_PD":
            from .kinematics import leg_ik
            settle_cmd = np.concatenate([leg_ik(rc, L, settle_scaled[3 * L:3 * L + 3]) for L in range(4)])
        else:
            settle_cmd = settle_scaled
        # interface_base.py:194: the returned "settling action" is the inverse map applied to the *motor command*;
        # for CARTESIAN_PD that command is joint angles pushed through the Cartesian scaling (reference quirk, kept).
        settle_action = to_actual_action_space(scale_command_to_action(settle_cmd, lo, hi), aspace, symm)
    else:  # settle_robot_by_pd: DEFAULT/PD interface, 1500 steps toward INIT_MOTOR_ANGLES (control_interface/utils.py:24-32)
        plo, phi = rc.RL_LOWER_ANGLE_JOINT, rc.RL_UPPER_ANGLE_JOINT
        settle_cmd = scale_action_to_command(scale_command_to_action(rc.INIT_MOTOR_ANGLES, plo, phi), plo, phi)
        settle_action = np.zeros(12)
    for i in range(12):
        cfg.settle_cmd[i] = settle_cmd[i]
        cfg.settle_action[i] = settle_action[i] if i < len(settle_action) else 0.0
    fb, fa = butter2_lowpass(3.0, 1.0 / (action_repeat * time_step))  # action_filter.py:41-43, gym_env.py:262
    for i in range(3):
        cfg.filt_b[i], cfg.filt_a[i] = fb[i], fa[i]
    cfg.fallen_height = rc.IS_FALLEN_HEIGHT
    # engine constants assumed for PyBullet defaults (DESIGN.md "contact model")
    # Bullet solver constants assumed for PyBullet's defaults: keywords so that they can follow
    # what tools/pin_against_pybullet.py finds on a machine that has PyBullet
    cfg.contact_erp, cfg.joint_erp, cfg.warmstart, cfg.vel_cap = float(contact_erp), float(joint_erp), float(warmstart), rc.VELOCITY_LIMITS[0]
    cfg.contact_slop = float(contact_slop)
    cfg.info_fields = int(bool(info_fields))
    cfg.payload_soft = {"weld": 0, "soft": 1}[payload]   # "soft": the block as a second body on a fixed constraint (many-rows solver in every substep)
    # True (the default since round 5): every collision primitive of the URDF pushes back on the plane, as in PyBullet (quadruped.py:533-539).
    # "auto": only where the episode goes on after such a contact, i.e. under NO_TASK (the reference's CPG driver, hopf_network.py:183-190);
    # every other task ends the episode at the end of the env step in which a non-foot link touches (task_base.py:137-147), so the response
    # shapes the last <= action_repeat substeps of an episode that is over.  Measured over 105 909 episodes of the benchmark
    # (tools/body_contacts_delta.py, profiles/r05_a_body_contacts_delta.md): "auto" changes no episode's `done` step and the terminal
    # reward by < 6e-5, but the TERMINAL OBSERVATION of 30 % of the fall-ended episodes beyond the parity tolerances (joint velocities by
    # rad/s) -- so it is an opt-in for learners that never read a terminated episode's last observation, at about 1.7x the rate.
    if body_contacts == "auto":
        body_contacts = task_env == "NO_TASK"
    cfg.body_contacts, cfg.self_collision = int(bool(body_contacts)), int(bool(self_collision))
    # m/s: a support point's rows are built once its normal row comes this close to acting (include/qs_amd.h); inf = every point in range
    # gets its rows, Bullet's (and the oracle's) row set
    if not float(support_margin) >= 0.0:
        raise ValueError(f"support_margin must be >= 0 (m/s) or inf, got {support_margin!r}")
    cfg.support_margin = min(float(support_margin), 3.0e38)
    # changeDynamics(mass=...) is only ever called by the mass randomizer (env_randomizer.py:56-83 -> quadruped.py:761, 776), at every
    # reset and for every randomised link, also when the drawn mass equals the URDF's: without it the URDF tensors stay
    rule = mass_inertia_rule if (rand & RAND_MASSES) else "scale"
    tab = unit_inertia_table(rule)
    for k in range(4):
        for i in range(6):
            cfg.unit_inertia[k][i] = tab[k][i]
    cfg.friction_cone = {"pyramid": 0, "cone": 1}[friction_model]   # PyBullet's enableConeFriction off / on (see include/qs_amd.h)
    cfg.solver_residual_threshold = float(solver_residual_threshold)
    for i, s in enumerate(lay["std"]):
        cfg.obs_noise_std[i] = s
    if not isRLGymInterface:
        landing_action = np.zeros(12)
    elif action_space_mode == "CPG":
        landing_action = np.zeros(5)
    else:
        landing_action = to_actual_action_space(scale_command_to_action(np.asarray(landing_pose, float), lo, hi), action_space_mode, symm)
    # env/wrappers/: LandingWrapper, GoToRestWrapper, LandingWrapper2, LandingWrapperBackflip, LandingWrapperBackflip2,
    # LandingWrapperContinuous.  LandingWrapperContinuous2 can never trigger in the reference (landing_wrapper_continuous2.py:66
    # tests the bound method `self.robot._is_flying`, which is always truthy), so it is the identity.
    WRAPPERS = {None: 0, "NONE": 0, "LANDING": 1, "GO_TO_REST": 2, "LANDING2": 3, "LANDING_BACKFLIP": 4, "LANDING_BACKFLIP2": 5,
                "LANDING_CONTINUOUS": 6, "LANDING_CONTINUOUS2": 0}
    cfg.wrapper_mode = _lookup(WRAPPERS, wrapper, "wrapper")
    for i in range(12):
        cfg.landing_action[i] = landing_action[i] if i < len(landing_action) else 0.0
    cfg.landing_kp, cfg.landing_kd = 60.0, 1.5                                        # landing_wrapper.py:22-27
    cfg.rest_kp, cfg.rest_kd = 60.0, (0.8 if enable_springs else 1.5)                 # go_to_rest_wrapper.py:26-32
    cfg.rest_time = 1.0 if enable_springs else 0.3                                    # go_to_rest_wrapper.py:16-19
    meta = dict(robot_config=rc, layout=lay, lower=np.array(lo, float), upper=np.array(hi, float), symm_idx=symm,
                init_pose=np.array(init_pose, float), landing_pose=np.array(landing_pose, float),
                settle_action=np.array(settle_action, float), landing_action=np.array(landing_action, float),
                action_space_mode=action_space_mode, motor_control_mode=motor_control_mode, task_env=task_env,
                observation_space_mode=observation_space_mode, env_randomizer_mode=env_randomizer_mode, demo=None)
    if task_env in DEMO_FILES:
        if action_space_mode == "CPG" or not isRLGymInterface:
            raise ValueError("the DEMO tasks compare the policy's action with a recorded one: they need an RL action space")
        if demo is not None:
            rows = np.load(demo) if isinstance(demo, (str, bytes)) or hasattr(demo, "__fspath__") else np.asarray(demo)
            rows = np.ascontiguousarray(rows, dtype=np.float32)
            if rows.ndim != 2 or rows.shape[1] != action_dim + 38 or rows.shape[0] < 1:
                raise ValueError(f"a demonstration for this action space has rows of {action_dim + 38} numbers "
                                 f"(get_demonstration_wrapper.py:35-58), got an array of shape {rows.shape}")
            meta["demo"] = rows
    return cfg, meta
