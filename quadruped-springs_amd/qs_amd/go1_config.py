"""Go1 robot constants for the batched simulation step.

Restates the numeric content of the reference's two config modules as plain data
(quadruped_spring/go1/configs_go1_with_springs.py and configs_go1_without_springs.py;
line numbers in comments refer to the with-springs file unless marked [w/o]).
Pinned by tests/golden/g10_sensor_limits.npz and g1_action_map.npz.
"""
import math
from types import SimpleNamespace

import numpy as np

NUM_MOTORS = 12
NUM_LEGS = 4


def _tile(v):
    return np.array(list(v) * NUM_LEGS, dtype=np.float64)


def make_config(enable_springs: bool) -> SimpleNamespace:
    """Return a fresh namespace (the BACKFLIP task edits joint limits in place, motor_interface.py:17-22)."""
    c = SimpleNamespace()
    c.NUM_MOTORS, c.NUM_LEGS, c.MOTORS_PER_LEG = 12, 4, 3
    c.INIT_POSITION = [0, 0, 0.32]                                   # :23
    c.IS_FALLEN_HEIGHT = 0.10 if enable_springs else 0.12            # :24 / [w/o] :24
    c.INIT_ORIENTATION = (0, 0, 0, 1)                                # :26
    thigh0, calf0 = math.pi / 4, -math.pi / 2                        # :31-33
    c.INIT_MOTOR_ANGLES = _tile([0.0, thigh0, calf0])                # :35-36
    c.ANGLE_LANDING_POSE = c.INIT_MOTOR_ANGLES                       # :38
    c.ANGLE_SETTLING_POSE = _tile([0.0, 1.14, -2.5 if enable_springs else -2.19])  # :39 / [w/o] :38
    c.JOINT_DIRECTIONS = np.ones(12)
    c.JOINT_OFFSETS = np.zeros(12)
    c.HIP_LINK_LENGTH, c.THIGH_LINK_LENGTH, c.CALF_LINK_LENGTH = 0.0847, 0.213, 0.213  # :56-58
    c.X_OFFSET, c.Y_OFFSET = 0.1881, 0.04675                         # :60-61
    sign = [-1, 1, -1, 1]
    c.NOMINAL_FOOT_POS_LEG_FRAME = np.array([[0, s * 0.0847, -0.32] for s in sign]).flatten()      # :64-71
    c.CARTESIAN_LANDING_POSE = np.array([[0, s * 0.0847, -0.29] for s in sign]).flatten()          # :72
    c.CARTESIAN_SETTLING_POSE = np.array([[-0.02, s * 0.0847, -0.15] for s in sign]).flatten()     # :73
    c.INIT_HEIGHT = 0.35
    c.REAL_UPPER_ANGLE_JOINT = _tile([1.0471975512, 2.96705972839, -0.837758040957])   # :80
    c.REAL_LOWER_ANGLE_JOINT = _tile([-1.0471975512, -0.663225115758, -2.72271363311])  # :81
    c.RL_UPPER_ANGLE_JOINT = _tile([0.2, thigh0 + 0.5, -0.95])       # :84
    c.RL_LOWER_ANGLE_JOINT = _tile([-0.2, thigh0 - 0.5, -2.5 if enable_springs else -2.12])  # :85-87 / [w/o] :82
    up_z = 0.18 if enable_springs else 0.11                          # :90-92 / [w/o] :87
    c.RL_UPPER_CARTESIAN_POS = c.NOMINAL_FOOT_POS_LEG_FRAME + _tile([0.2, 0.05, up_z])
    c.RL_LOWER_CARTESIAN_POS = c.NOMINAL_FOOT_POS_LEG_FRAME - _tile([0.2, 0.05, 0.07])   # :94-96
    c.TORQUE_LIMITS = _tile([23.7, 23.7, 33.55])                     # :100
    c.RL_TORQUE_LIMITS = c.TORQUE_LIMITS.copy()
    c.VELOCITY_LIMITS = _tile([30.1, 30.1, 30.1])                    # :102
    c.RL_VELOCITY_LIMITS = _tile([10, 10, 10])                       # :103
    if enable_springs:
        c.MOTOR_KP, c.MOTOR_KD = [75.0] * 12, [0.8, 1.0, 1.0] * 4    # :106-107
        c.kpCartesian, c.kdCartesian = np.diag([1200, 2000, 2000]), np.diag([13, 15, 15])  # :113-114
        c.SPRINGS_STIFFNESS = [20, 20, 30]                           # :150-158
        c.SPRINGS_DAMPING = [0.3, 0.3, 0.3]
        c.SPRINGS_REST_ANGLE = [0, thigh0, calf0 + 0.3]              # :160
    else:
        c.MOTOR_KP, c.MOTOR_KD = [55, 60, 60] * 4, [0.8, 1.0, 1.0] * 4  # [w/o] :108-109 (last assignment wins)
        c.kpCartesian, c.kdCartesian = np.diag([500, 500, 500]), np.diag([10, 10, 10])  # [w/o] :112-113
        c.SPRINGS_STIFFNESS = [0, 0, 0]
        c.SPRINGS_DAMPING = [0, 0, 0]
        c.SPRINGS_REST_ANGLE = [0, thigh0, calf0 + 0.3]
    # sensor limits :176-209
    c.HEIGHT_HIGH, c.HEIGHT_LOW = np.array([0.4]), np.array([0.1])
    c.VEL_LIN_HIGH = np.array([5.0] * 3); c.VEL_LIN_LOW = -c.VEL_LIN_HIGH
    c.VEL_ANG_HIGH = np.array([3.0] * 3); c.VEL_ANG_LOW = -c.VEL_ANG_HIGH
    c.ORIENT_RPY_HIGH = np.array([math.pi] * 3); c.ORIENT_RPY_LOW = -c.ORIENT_RPY_HIGH
    c.ORIENT_RATE_HIGH = np.array([5.0] * 3)
    c.JOINT_ANGLES_HIGH = c.RL_UPPER_ANGLE_JOINT      # same array object, like the reference (:182)
    c.JOINT_ANGLES_LOW = c.RL_LOWER_ANGLE_JOINT
    c.JOINT_VELOCITIES_HIGH = c.RL_VELOCITY_LIMITS; c.JOINT_VELOCITIES_LOW = -c.JOINT_VELOCITIES_HIGH
    c.CONTACT_FORCE_HIGH = np.array([5.0] * 4); c.CONTACT_FORCE_LOW = -c.CONTACT_FORCE_HIGH
    c.CONTACT_BOOL_HIGH = np.array([1.0] * 4); c.CONTACT_BOOL_LOW = np.array([0.0] * 4)
    c.FEET_POS_HIGH = c.RL_UPPER_CARTESIAN_POS; c.FEET_POS_LOW = c.RL_LOWER_CARTESIAN_POS
    c.FEET_VEL_HIGH = np.array([10.0] * 12); c.FEET_VEL_LOW = -c.FEET_POS_HIGH   # typo kept from :206
    c.QUATERNION_HIGH = np.ones(4); c.QUATERNION_LOW = np.zeros(4)
    c.PITCH_HIGH = np.array([math.pi]); c.PITCH_LOW = -c.PITCH_HIGH
    c.PITCH_RATE_HIGH = np.array([5.0]); c.PITCH_RATE_LOW = -c.PITCH_RATE_HIGH
    # noise :215-230
    k = 0.01
    c.STD_COEFF = k
    c.HEIGHT_NOISE = c.HEIGHT_HIGH * k * 0.8
    c.VEL_LIN_NOISE = c.VEL_LIN_HIGH * k * 0.8
    c.VEL_ANG_NOISE = c.VEL_ANG_HIGH * k
    c.ORIENT_RPY_NOISE = c.ORIENT_RPY_HIGH * k
    c.ORIENT_RATE_NOISE = c.ORIENT_RATE_HIGH * k
    c.JOINT_ANGLES_NOISE = np.maximum(abs(c.JOINT_ANGLES_HIGH), abs(c.JOINT_ANGLES_LOW)) * k * 0.1
    c.JOINT_VELOCITIES_NOISE = c.JOINT_VELOCITIES_HIGH * k * 0.6
    c.CONTACT_FORCE_NOISE = c.CONTACT_FORCE_HIGH * k
    c.CONTACT_BOOL_NOISE = np.zeros(4)
    c.FEET_POS_NOISE = _tile([0.1, 0.05, 0.1]) * k
    c.FEET_VEL_NOISE = c.FEET_VEL_HIGH * k
    c.QUATERNION_NOISE = c.QUATERNION_HIGH * k
    c.PITCH_NOISE = c.PITCH_HIGH * k * 0.9
    c.PITCH_RATE_NOISE = c.PITCH_RATE_HIGH * k
    return c
