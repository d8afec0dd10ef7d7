#!/usr/bin/env python3
"""Pin the oracle's rigid-body substep against real PyBullet -- on a machine that has PyBullet.

STATUS: NOT RUN in the build image or on the GPU box: `pybullet` is not installed on either and there is no network
("parity unpinned" for stepSimulation / getContactPoints).  The script is the recipe a maintainer
of the reference runs once to close that gap; without PyBullet it prints why it cannot and exits 0.

What it does, following the reference's call sites (no reference module is imported; only its URDF, a data file, is read
from the path given with --urdf):

  world    resetSimulation, numSolverIterations = int(300 / action_repeat), setTimeStep, plane.urdf at x = 80,
           gravity (0, 0, -9.8)                                                        quadruped_gym_env.py:299-309
  robot    loadURDF(go1.urdf, (0, 0, 0.32), identity, URDF_USE_SELF_COLLISION)         quadruped.py:533-546
           linear / angular damping 0, lateralFriction 1 on every link, maxJointVelocity 30.1 on the motor links
                                                                                      quadruped.py:663-683
           default joint motors off (VELOCITY_CONTROL, force 0), resetJointState       quadruped.py:496-511
  substep  joint torque = PD (clipped) + PEA spring, two TORQUE_CONTROL writes per joint, stepSimulation
                                                                                      quadruped.py:288-320, quadruped_gym_env.py:218-225

Every substep the oracle (oracle/libqso_f64.so) is put INTO PyBullet's state, given the same joint torques, stepped once, and
compared with PyBullet's next state (positions, velocities, per-foot normal force): a resynchronised comparison, the same
protocol tests/test_gpu_parity.py uses between the kernel and the oracle.  A stand / crouch / push-off / flight / landing
script covers sticking contact, lift-off, free flight and impact.  With --write the (state, torque, next state, foot force)
rows are saved as a fixture (tests/golden/pybullet_steps.npz) that tests/test_oracle_physics.py would then check on every
run; until such a file exists the physics half of the oracle is pinned only by first-principles tests.
"""
import argparse
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "quadruped-springs_amd"))

MOTOR_IDS = [2, 3, 4, 6, 7, 8, 10, 11, 12, 14, 15, 16]     # FR, FL, RR, RL x hip, thigh, calf
FOOT_IDS = [5, 9, 13, 17]
INIT_Q = np.tile([0.0, np.pi / 4, -np.pi / 2], 4)
KP, KD = 75.0, np.tile([0.8, 1.0, 1.0], 4)                  # configs_go1_with_springs.py:106-107
TAU_MAX = np.tile([23.7, 23.7, 33.55], 4)                   # :100-101
SPRING_K, SPRING_B = np.tile([20.0, 20.0, 30.0], 4), np.tile([0.3, 0.3, 0.3], 4)   # :150-160
SPRING_REST = np.tile([0.0, np.pi / 4, -np.pi / 2 + 0.3], 4)


def spring_torque(q, qd):
    """springs.py:28-74: every spring pushes one way only."""
    k, b = SPRING_K.copy(), SPRING_B.copy()
    for leg in range(4):
        h, t, c = 3 * leg, 3 * leg + 1, 3 * leg + 2
        right = leg in (0, 2)
        if (q[h] > SPRING_REST[h]) if right else (q[h] < SPRING_REST[h]):
            k[h] = b[h] = 0.0
        if q[t] < SPRING_REST[t]:
            k[t] = b[t] = 0.0
        if q[c] > SPRING_REST[c]:
            k[c] = b[c] = 0.0
    return -k * (q - SPRING_REST) - b * qd


def script(i):
    """Joint targets for substep i: stand, crouch, push off, tuck in flight, land."""
    crouch, extend = np.tile([0.0, 1.0, -2.0], 4), np.tile([0.0, 0.7, -1.5], 4)
    if i < 1000:
        return INIT_Q
    if i < 1400:
        return INIT_Q + (crouch - INIT_Q) * (i - 1000) / 400.0
    if i < 1430:
        return extend      # on the oracle alone: a 10 cm hop, 257 substeps of flight, landing on four feet by substep ~1700
    return INIT_Q


def bullet_state(p, robot):
    pos, quat = p.getBasePositionAndOrientation(robot)
    v, w = p.getBaseVelocity(robot)
    js = p.getJointStates(robot, MOTOR_IDS)
    return np.concatenate([pos, quat, v, w, [j[0] for j in js], [j[1] for j in js]])


def foot_forces(p, robot, plane):
    f = np.zeros(4)
    for c in p.getContactPoints(bodyA=robot, bodyB=plane):
        if c[3] in FOOT_IDS:
            f[FOOT_IDS.index(c[3])] += c[9]
    return f


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--urdf", required=True, help="the reference's go1.urdf (quadruped_spring/go1/go1_description/urdf/go1.urdf)")
    ap.add_argument("--substeps", type=int, default=2600)
    ap.add_argument("--springs", type=int, default=1)
    ap.add_argument("--mu", type=float, default=1.0, help="ground lateralFriction (env_randomizer.py:287-289 draws 0.5 .. 1)")
    ap.add_argument("--grid", nargs="+", default=["friction_model=cone,pyramid", "contact_erp=0.08,0.2"],
                    help="hypotheses about the engine to try, key=v1,v2,... each (keywords of qs_amd.config.build_config: friction_model, "
                         "contact_erp, contact_slop, joint_erp, warmstart, solver_residual_threshold, mass_inertia_rule, body_contacts, self_collision); "
                         "every combination gets its own table, the build's defaults first.  Defaults: the implicit cone (btMultiBodyDynamicsWorld sets "
                         "SOLVER_USE_2_FRICTION_DIRECTIONS, implicit cone friction is on unless disabled) vs the pyramid with Bullet's skip rule, and "
                         "PyBullet's erp2 = 0.08 vs Bullet's 0.2.  mass_inertia_rule=collision_shape,scale only shows with --remass")
    ap.add_argument("--remass", action="store_true",
                    help="call changeDynamics(mass = URDF mass) on trunk and leg links first, as the reference's mass randomizer does at every reset "
                         "(quadruped.py:761, 776): PyBullet then replaces their inertia by its collision-shape rule; the oracle is built with the "
                         "mass randomizer's rule (mass_inertia_rule of the grid) and nominal masses")
    ap.add_argument("--payload", nargs=3, type=float, default=None, metavar=("MASS", "X", "Z"),
                    help="attach the mass randomizer's payload block as the reference does (quadruped.py:778-819: createMultiBody with a box of half "
                         "extent 0.05 at base + (X, 0, Z), createConstraint(JOINT_FIXED), collisions with the robot's links off); the oracle gets the same "
                         "block, welded or on its own fixed constraint: --grid payload=weld,soft")
    ap.add_argument("--write", default="", help="save the compared rows as an .npz fixture")
    args = ap.parse_args()
    try:
        import pybullet
        import pybullet_data
        from pybullet_utils import bullet_client
    except ImportError as e:
        print(f"reference PyBullet path unavailable ({e}): nothing compared, nothing written")
        return 0

    from oracle.qso import Oracle
    from qs_amd.config import build_config
    import itertools
    axes = []
    for item in args.grid:
        key, _, vals = item.partition("=")
        conv = (lambda v: v) if key in ("friction_model", "mass_inertia_rule", "payload") else (lambda v: bool(int(v))) if key in ("body_contacts", "self_collision") else float
        axes.append([(key, conv(v)) for v in vals.split(",")])
    oracles, labels = [], []
    for combo in itertools.product(*axes):
        cfg, _ = build_config(n_envs=1, isRLGymInterface=False, motor_control_mode="TORQUE", task_env="NO_TASK", observation_space_mode="ENCODER",
                              enable_springs=bool(args.springs), env_randomizer_mode="MASS_RANDOMIZER" if args.remass else "NONE",
                              enable_action_filter=False, **dict(combo))
        cfg.randomizer_flags = 8                     # keep the parameters set below: nominal masses, no payload
        oracles.append(Oracle(cfg))
        par = oracles[-1].get_info(6)
        par[0, 0], par[0, 16:24] = args.mu, [5.204, 0.591, 0.92, 0.131, 0.0, 0.0, 0.0, 0.0]
        if args.payload:
            par[0, 20], par[0, 21], par[0, 23] = args.payload
        oracles[-1].set_params(5, par)
        labels.append(", ".join(f"{k} = {v}" for k, v in combo))

    p = bullet_client.BulletClient(connection_mode=pybullet.DIRECT)
    p.resetSimulation()
    p.setPhysicsEngineParameter(numSolverIterations=30)
    p.setTimeStep(0.001)
    plane = p.loadURDF(os.path.join(pybullet_data.getDataPath(), "plane.urdf"), basePosition=[80, 0, 0])
    p.setGravity(0, 0, -9.8)
    robot = p.loadURDF(args.urdf, [0, 0, 0.32], [0, 0, 0, 1], flags=p.URDF_USE_SELF_COLLISION)
    nj = p.getNumJoints(robot)
    assert nj == 18 and [p.getJointInfo(robot, j)[2] == p.JOINT_REVOLUTE for j in range(nj)].count(True) == 12
    p.changeDynamics(robot, -1, linearDamping=0, angularDamping=0)       # net effect of quadruped.py:663-668
    for j in range(-1, nj):
        p.changeDynamics(robot, j, lateralFriction=1.0)
    p.changeDynamics(plane, -1, lateralFriction=args.mu)
    for j in MOTOR_IDS:
        p.changeDynamics(robot, j, maxJointVelocity=30.1)
    if args.payload:
        mass, px, pz = args.payload
        shape = p.createCollisionShape(p.GEOM_BOX, halfExtents=[0.05] * 3, collisionFramePosition=[0, 0, 0])
        block = p.createMultiBody(baseMass=mass, baseCollisionShapeIndex=shape, basePosition=[px, 0, 0.32 + pz], baseOrientation=[0, 0, 0, 1])
        p.createConstraint(robot, -1, block, -1, p.JOINT_FIXED, [0, 0, 0], [0, 0, 0], [-px, 0, -pz])
        for j in range(-1, nj):
            p.setCollisionFilterPair(robot, block, j, -1, 0)
    if args.remass:
        p.changeDynamics(robot, 0, mass=5.204)
        for j, m in zip(MOTOR_IDS, [0.591, 0.92, 0.131] * 4):
            p.changeDynamics(robot, j, mass=m)
    for j in range(nj):
        p.setJointMotorControl2(robot, j, p.VELOCITY_CONTROL, targetVelocity=0, force=0)
    for j, a in zip(MOTOR_IDS, INIT_Q):
        p.resetJointState(robot, j, a, targetVelocity=0)

    rows = dict(state=[], tau=[], next_state=[], foot_force=[])
    keys = ("pos", "quat", "v", "w", "q", "qd", "force_rel")
    worst = [dict.fromkeys(keys, 0.0) for _ in oracles]
    phase_of = lambda i: "stand" if i < 1000 else "crouch" if i < 1400 else "push" if i < 1430 else "flight+landing"
    by_phase = [dict() for _ in oracles]
    noted_other = False
    for i in range(args.substeps):
        s = bullet_state(p, robot)
        q, qd = s[13:25], s[25:37]
        tau_m = np.clip(-KP * (q - script(i)) - KD * qd, -TAU_MAX, TAU_MAX)      # quadruped_motor.py:45-99
        tau_s = spring_torque(q, qd) if args.springs else np.zeros(12)
        for j, tm, ts in zip(MOTOR_IDS, tau_m, tau_s):                           # two writes per joint: they add (App. D-1)
            p.setJointMotorControl2(robot, j, p.TORQUE_CONTROL, force=tm)
            if args.springs:
                p.setJointMotorControl2(robot, j, p.TORQUE_CONTROL, force=ts)
        for o in oracles:
            o.set_state(s[None])
            o.phys_step(0, tau_m + tau_s)
        p.stepSimulation()
        other = [c[3] for c in p.getContactPoints(bodyA=robot) if c[3] not in FOOT_IDS]
        if other and not noted_other:
            # the oracle (body_contacts) lets these links push back too, with at most two support points per leg; only the feet's forces
            # are compared below
            print(f"substep {i}: PyBullet reports contact on non-foot link(s) {sorted(set(other))}")
            noted_other = True
        sb, fb = bullet_state(p, robot), foot_forces(p, robot, plane)
        for k_o, o in enumerate(oracles):
            so, fo = o.get_state()[0], o.get_info(0)[0]
            d = dict(pos=np.abs(sb[:3] - so[:3]).max(), quat=min(np.abs(sb[3:7] - so[3:7]).max(), np.abs(sb[3:7] + so[3:7]).max()),
                     v=np.abs(sb[7:10] - so[7:10]).max(), w=np.abs(sb[10:13] - so[10:13]).max(),
                     q=np.abs(sb[13:25] - so[13:25]).max(), qd=np.abs(sb[25:] - so[25:]).max(),
                     force_rel=np.abs(fb - fo).max() / max(1.0, fb.max()))
            ph = by_phase[k_o].setdefault(phase_of(i), dict.fromkeys(d, 0.0))
            for k, x in d.items():
                worst[k_o][k] = max(worst[k_o][k], x)
                ph[k] = max(ph[k], x)
        rows["state"].append(s); rows["tau"].append(tau_m + tau_s); rows["next_state"].append(sb); rows["foot_force"].append(fb)

    for label, phases, w in zip(labels, by_phase, worst):
        print(f"one-substep deviation oracle ({label}) vs PyBullet (max over the script; oracle re-seated in PyBullet's state every substep)")
        for name, ph in phases.items():
            print(f"  {name:16s} " + "  ".join(f"{k} {x:.3e}" for k, x in ph.items()))
        print("  overall          " + "  ".join(f"{k} {x:.3e}" for k, x in w.items()))
    if args.write:
        np.savez_compressed(args.write, mu=args.mu, springs=args.springs, pybullet_api_version=pybullet.getAPIVersion(),
                            **{k: np.asarray(v) for k, v in rows.items()})
        print("wrote", args.write)
    return 0


if __name__ == "__main__":
    sys.exit(main())
