"""Analytic Go1 leg kinematics on the host (numpy), for callers that need them outside the device step
(settle command of CARTESIAN_PD, wrappers).  Follows quadruped_spring/env/quadruped.py:348-438."""
import numpy as np


def _side(leg):
    return -1 if leg in (0, 2) else 1


def leg_fk_jacobian(robot_config, q, leg):
    """quadruped.py:348-392 -> (J[3,3], pos[3]) in the hip frame."""
    q = np.asarray(q)[leg * 3: leg * 3 + 3] if len(q) == 12 else np.asarray(q)
    l1, l2, l3 = robot_config.HIP_LINK_LENGTH, robot_config.THIGH_LINK_LENGTH, robot_config.CALF_LINK_LENGTH
    sg = _side(leg)
    s1, s2, s3 = np.sin(q)
    c1, c2, c3 = np.cos(q)
    c23 = c2 * c3 - s2 * s3
    s23 = s2 * c3 + c2 * s3
    J = np.zeros((3, 3))
    J[1, 0] = -sg * l1 * s1 + l2 * c2 * c1 + l3 * c23 * c1
    J[2, 0] = sg * l1 * c1 + l2 * c2 * s1 + l3 * c23 * s1
    J[0, 1] = -l3 * c23 - l2 * c2
    J[1, 1] = -l2 * s2 * s1 - l3 * s23 * s1
    J[2, 1] = l2 * s2 * c1 + l3 * s23 * c1
    J[0, 2] = -l3 * c23
    J[1, 2] = -l3 * s23 * s1
    J[2, 2] = l3 * s23 * c1
    pos = np.array([
        -l3 * s23 - l2 * s2,
        l1 * sg * c1 + l3 * (s1 * c23) + l2 * c2 * s1,
        l1 * sg * s1 - l3 * (c1 * c23) - l2 * c1 * c2,
    ])
    return J, pos


def leg_ik(robot_config, leg, xyz):
    """quadruped.py:399-438."""
    sh, el, wr = robot_config.HIP_LINK_LENGTH, robot_config.THIGH_LINK_LENGTH, robot_config.CALF_LINK_LENGTH
    x, y, z = xyz
    D = (y ** 2 + z ** 2 - sh ** 2 + x ** 2 - el ** 2 - wr ** 2) / (2 * wr * el)
    D = np.clip(D, -1.0, 1.0)
    sg = _side(leg)
    wrist = np.arctan2(-np.sqrt(1 - D ** 2), D)
    sc = max(y ** 2 + z ** 2 - sh ** 2, 0.0)
    shoulder = -np.arctan2(z, y) - np.arctan2(np.sqrt(sc), sg * sh)
    elbow = np.arctan2(-x, np.sqrt(sc)) - np.arctan2(wr * np.sin(wrist), el + wr * np.cos(wrist))
    return np.array([-shoulder, elbow, wrist])
