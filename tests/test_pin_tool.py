"""tools/pin_against_pybullet.py (the recipe that pins the oracle's substep against real PyBullet) cannot meet PyBullet in this image.
Here its whole control flow runs against a stand-in client that answers the PyBullet calls the tool makes from the oracle itself
(the build's defaults: implicit cone, contact_erp = 0.08): the matching hypothesis must come out with zero deviation, a different one must not, and the fixture it
writes must hold the compared rows."""
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pin_recipe_runs_against_a_stand_in_client(tmp_path, monkeypatch, capsys):
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import pin_against_pybullet as P
    from oracle.qso import Oracle
    from qs_amd.config import build_config
    cfg, _ = build_config(n_envs=1, isRLGymInterface=False, motor_control_mode="TORQUE", task_env="NO_TASK", observation_space_mode="ENCODER",
                          enable_springs=True, env_randomizer_mode="NONE", enable_action_filter=False)

    class StandIn:
        JOINT_REVOLUTE, URDF_USE_SELF_COLLISION, VELOCITY_CONTROL, TORQUE_CONTROL = 0, 8, 0, 2

        def __init__(self):
            self.o = Oracle(cfg)
            self.o.set_params(0, np.array([[1.0]]))
            self.st = np.zeros(37); self.st[2] = 0.32; self.st[6] = 1.0
            self.tau, self.f = np.zeros(12), np.zeros(4)

        def loadURDF(self, path, *a, **k):
            return 0 if "plane" in path else 1

        def getNumJoints(self, b):
            return 18

        def getJointInfo(self, b, j):
            return (j, b"joint", 0 if j in P.MOTOR_IDS else 4)

        def setJointMotorControl2(self, b, j, mode, targetVelocity=0, force=0):
            if mode == self.TORQUE_CONTROL:
                self.tau[P.MOTOR_IDS.index(j)] += force          # two writes per joint add up

        def resetJointState(self, b, j, a, targetVelocity=0):
            self.st[13 + P.MOTOR_IDS.index(j)] = a

        def getBasePositionAndOrientation(self, b):
            return self.st[0:3], self.st[3:7]

        def getBaseVelocity(self, b):
            return self.st[7:10], self.st[10:13]

        def getJointStates(self, b, ids):
            return [(self.st[13 + i], self.st[25 + i], 0, 0) for i in range(12)]

        def stepSimulation(self):
            self.o.set_state(self.st[None]); self.o.phys_step(0, self.tau)
            self.st, self.f = self.o.get_state()[0].copy(), self.o.get_info(0)[0]
            self.tau[:] = 0

        def getContactPoints(self, bodyA=None, bodyB=None):
            return [(0, 1, 0, P.FOOT_IDS[k], -1, 0, 0, 0, 0, self.f[k]) for k in range(4) if self.f[k] > 0]

        GEOM_BOX, JOINT_FIXED = 3, 4

        def __getattr__(self, name):     # resetSimulation, setGravity, changeDynamics, ...: nothing to do
            return lambda *a, **k: None

    pb = types.ModuleType("pybullet"); pb.DIRECT = 2; pb.getAPIVersion = lambda: 0
    pd = types.ModuleType("pybullet_data"); pd.getDataPath = lambda: "/nowhere"
    pu, bc = types.ModuleType("pybullet_utils"), types.ModuleType("pybullet_utils.bullet_client")
    bc.BulletClient = lambda connection_mode=None: StandIn()
    pu.bullet_client = bc
    for name, mod in (("pybullet", pb), ("pybullet_data", pd), ("pybullet_utils", pu), ("pybullet_utils.bullet_client", bc)):
        monkeypatch.setitem(sys.modules, name, mod)
    out = tmp_path / "rows.npz"
    monkeypatch.setattr(sys, "argv", ["pin", "--urdf", "/nowhere/go1.urdf", "--substeps", "1750", "--write", str(out)])
    assert P.main() == 0
    text = capsys.readouterr().out
    blocks = text.split("one-substep deviation oracle")
    assert len(blocks) == 5                                   # 2 friction models x 2 contact ERPs
    assert "friction_model = cone, contact_erp = 0.08" in blocks[1] and "overall          pos 0.000e+00" in blocks[1] and "force_rel 0.000e+00" in blocks[1].splitlines()[-1]
    assert "friction_model = cone, contact_erp = 0.2" in blocks[2] and "force_rel 0.000e+00" not in blocks[2].split("overall")[1]
    assert "friction_model = pyramid, contact_erp = 0.08" in blocks[3] and "overall          pos 0.000e+00" not in blocks[3]      # the hop slides a little: the models differ
    z = np.load(out)
    assert z["state"].shape == (1750, 37) and z["tau"].shape == (1750, 12) and z["foot_force"].shape == (1750, 4)
    assert z["state"][:, 2].max() > 0.38 and (z["foot_force"].sum(axis=1) == 0).sum() > 100      # the script hops: there is a flight phase
