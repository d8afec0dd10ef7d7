"""QuadrupedGymEnv: the reference's gym.Env surface (quadruped_spring/env/quadruped_gym_env.py:41-436) as an N = 1
view over the batched device step.  Same constructor keywords, same 4-tuple step / old-gym reset, same getters."""
import numpy as np

from ..config import EPISODE_LENGTH, scale_command_to_action, to_actual_action_space
from ..kinematics import leg_fk_jacobian, leg_ik
from ..spaces import GymEnv
from ..vec_env import QuadrupedVecEnv

ACTION_EPS = 0.01
OBSERVATION_EPS = 0.01


class _RobotView:
    """State getters of quadruped.py:107-262, 348-449 on the single environment."""

    def __init__(self, env):
        self._env = env
        self._robot_config = env._robot_config

    def _state(self):
        row = self._env._replay_row   # inside a sub-step callback: the state after that physics substep (trace tap)
        if row is not None:
            return row[1:38]
        return self._env._vec.get_state()[0].cpu().numpy().astype(np.float64)

    def GetBasePosition(self):
        return tuple(self._state()[0:3])

    def GetBaseOrientation(self):
        return tuple(self._state()[3:7])

    def GetBaseLinearVelocity(self):
        return self._state()[7:10]

    def GetBaseAngularVelocity(self):
        return self._state()[10:13]

    def GetMotorAngles(self):
        return self._state()[13:25]

    def GetMotorVelocities(self):
        return self._state()[25:37]

    def GetMotorTorques(self):
        row = self._env._replay_row
        if row is not None:
            return row[38:50]
        return self._env._vec.get_info("torque")[0].cpu().numpy().astype(np.float64)

    def GetBaseOrientationRollPitchYaw(self):
        return self._env._vec.get_info("task")[0, 38:41].cpu().numpy().astype(np.float64)

    def GetContactInfo(self):
        row = self._env._replay_row
        if row is not None:   # invalid contacts are not part of the trace row: the end-of-step count is reported
            n_invalid = int(self._env._vec.get_info("n_invalid")[0, 0].item())
            return int(row[66:70].sum()), n_invalid, list(row[62:66]), [int(f) for f in row[66:70]]
        force = self._env._vec.get_info("foot_force")[0].cpu().numpy()
        flag = self._env._vec.get_info("foot_contact")[0].cpu().numpy()
        n_invalid = int(self._env._vec.get_info("n_invalid")[0, 0].item())
        return int(flag.sum()), n_invalid, list(force.astype(float)), [int(f) for f in flag]

    def _is_flying(self):
        return not any(self.GetContactInfo()[3])

    def ComputeJacobianAndPosition(self, legID):
        return leg_fk_jacobian(self._robot_config, self.GetMotorAngles(), legID)

    def ComputeInverseKinematics(self, legID, xyz_coord):
        return leg_ik(self._robot_config, legID, xyz_coord)

    def ComputeFeetPosAndVel(self):
        q, dq = self.GetMotorAngles(), self.GetMotorVelocities()
        pos, vel = np.zeros(12), np.zeros(12)
        for i in range(4):
            J, xyz = leg_fk_jacobian(self._robot_config, q, i)
            pos[3 * i:3 * i + 3] = xyz
            vel[3 * i:3 * i + 3] = J @ dq[3 * i:3 * i + 3]
        return pos, vel

    def get_spring_nominal_params(self):
        c = self._robot_config
        return c.SPRINGS_STIFFNESS, c.SPRINGS_DAMPING, c.SPRINGS_REST_ANGLE

    # ---- further getters of quadruped.py that evaluation / analysis code of the reference calls
    def getHeight(self):                                   # quadruped.py:82-84
        return self.GetBasePosition()[-1]

    def GetBaseOrientationMatrix(self):                    # quadruped.py:172-175 (getMatrixFromQuaternion: base -> world, row major)
        x, y, z, w = self.GetBaseOrientation()
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                         [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])

    def TransformAngularVelocityToLocalFrame(self, angular_velocity, orientation):   # quadruped.py:151-170: R(orientation)^T w
        x, y, z, w = orientation
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        return R.T @ np.asarray(angular_velocity, dtype=np.float64)

    def GetTrueBaseRollPitchYawRate(self):                 # quadruped.py:141-149
        return self.TransformAngularVelocityToLocalFrame(self.GetBaseAngularVelocity(), self.GetBaseOrientation())

    def _params(self):
        return self._env._vec.get_info("params")[0].cpu().numpy().astype(np.float64)

    def get_spring_real_stiffness_and_damping(self):       # quadruped.py:727-730 -> springs.py:28-74: (k, b, rest angles), unilateral gating
        p, q = self._params(), self.GetMotorAngles()
        k, b, rest = np.tile(p[1:4], 4), np.tile(p[4:7], 4), np.tile(p[7:10], 4)
        if not self._env._enable_springs:
            return np.zeros(12), np.zeros(12), rest
        for leg in range(4):
            hip, thigh, calf = q[3 * leg:3 * leg + 3]
            right = leg % 2 == 0                           # side_map = ["right", "left"] * 2
            off = [hip > rest[0] if right else hip < rest[0], thigh < rest[1], calf > rest[2]]
            for j in range(3):
                if off[j]:
                    k[3 * leg + j] = b[3 * leg + j] = 0.0
        return k, b, rest

    # masses as _RecordMassAndInertiaInfoFromURDF keeps them (quadruped.py:605-645; go1.urdf, PyBullet link order: SURVEY.md App. A)
    def GetBaseMassFromURDF(self):
        return [5.204]

    def GetLegMassesFromURDF(self):
        return [0.591, 0.92, 0.131] * 4

    def GetFootMassesFromURDF(self):
        return [0.06] * 4

    def GetTotalMassFromURDF(self):
        return [1e-5, 5.204, 0.001] + [0.591, 0.92, 0.131, 0.06] * 4

    def get_offset_mass_value(self):                       # quadruped.py:701-707: the payload block of the mass randomizer (0 without one)
        return float(self._params()[20])

    def get_offset_mass_position(self):                    # quadruped.py:709-719: block centre - base position, world frame
        if self._env._vec.cfg.payload_soft:                # the block is a body of its own: its own centre
            blk = self._env._vec.get_info("payload_block")[0, :3].cpu().numpy().astype(np.float64)
            return blk - np.asarray(self.GetBasePosition())
        return self.GetBaseOrientationMatrix() @ self._params()[21:24]

    def set_spring_stiffness(self, k):
        self._env._vec.set_params("spring_k", np.asarray(k, np.float32)[None])

    def set_spring_damping(self, b):
        self._env._vec.set_params("spring_b", np.asarray(b, np.float32)[None])


class _MotorModelView:
    """`robot._motor_model._kp / _kd`: the gains the reference's wrappers swap temporarily (landing_wrapper.py:18-37)."""

    def __init__(self, env):
        self._env = env

    def _get(self, a, b):
        return self._env._vec.get_info("params")[0, a:b].cpu().numpy().astype(np.float64)

    @property
    def _kp(self):
        return np.tile(self._get(10, 13), 4)

    @_kp.setter
    def _kp(self, v):
        self._env._vec.set_params("kp", np.broadcast_to(np.asarray(v, np.float32).ravel(), (12,))[:3].reshape(1, 3).copy()
                                  if np.size(v) > 1 else np.full((1, 3), float(np.asarray(v).ravel()[0]), np.float32))

    @property
    def _kd(self):
        return np.tile(self._get(13, 16), 4)

    @_kd.setter
    def _kd(self, v):
        self._env._vec.set_params("kd", np.broadcast_to(np.asarray(v, np.float32).ravel(), (12,))[:3].reshape(1, 3).copy()
                                  if np.size(v) > 1 else np.full((1, 3), float(np.asarray(v).ravel()[0]), np.float32))


class _ActionInterfaceView:
    """The part of env.get_ac_interface() the wrappers call (interface_base.py:75-82, 92-100, 111-119)."""

    def __init__(self, env):
        self._env = env

    def get_init_action(self):
        return np.asarray(self._env._settling_action, float)

    def get_landing_action(self):
        return np.asarray(self._env.get_landing_action(), float)

    def _transform_motor_command_to_action(self, command):
        m = self._env._vec.meta
        a12 = scale_command_to_action(np.asarray(command, float), m["lower"], m["upper"])
        return to_actual_action_space(a12, self._env._action_space_mode, m["symm_idx"])

    @staticmethod
    def generate_ramp(i, i_min, i_max, u_min, u_max):
        if i < i_min:
            return u_min
        if i > i_max:
            return u_max
        return u_min + (u_max - u_min) * (i - i_min) / (i_max - i_min)


class _TaskView:
    def __init__(self, env):
        self._env = env

    def compute_time_for_peak_heihgt(self):   # task_base.py:157-160 (sic)
        return float(self._env.robot.GetBaseLinearVelocity()[2]) / 9.81

    def get_jumping(self):
        return bool(self._scalars()[2] > 0.5)

    @property
    def is_jumping(self):                     # what the "is jumping" sensor reads (robot_sensors.py:182)
        return self.get_jumping()

    def compute_jumping_distance(self):       # task_base.py:108-116 (evaluation_wrapper.py:38): forward distance in the take-off frame
        sc = self._scalars()                  # [4:7] pose at take-off, [7] yaw at take-off, [32:35] task._pos_abs (the info block's pose cache)
        dx, dy, yaw = float(sc[32] - sc[4]), float(sc[33] - sc[5]), float(sc[7])
        return max(np.cos(yaw) * dx - np.sin(yaw) * dy, 0.0)

    def enable_rest_mode(self):   # robot_tasks.py:346-347: sets a flag the reference never reads
        pass

    # DEMO tasks (task_base.py:169-220): what ReferenceStateInitializationWrapper reads and sets
    @property
    def demo_list(self):
        return self._env._vec.demo_list

    @property
    def demo_length(self):
        return self._env._vec.demo_length

    @property
    def demo_counter(self):
        return int(self._scalars()[44])

    def set_demo_counter(self, value):
        self._env._vec.set_demo_counter(int(value))

    def _scalars(self):
        return self._env._vec.get_info("task")[0].cpu().numpy()

    @property
    def _switched_controller(self):
        return bool(self._scalars()[0] > 0.5)

    def is_switched_controller(self):
        return self._switched_controller

    @property
    def _max_height(self):
        return float(self._scalars()[14])

    @property
    def _relative_max_height(self):
        return float(self._scalars()[12])

    @property
    def _max_forward_distance(self):
        return float(self._scalars()[10])


class QuadrupedGymEnv(GymEnv):
    metadata = {"render.modes": ["rgb_array"]}

    def __init__(
        self,
        isRLGymInterface=True,
        time_step=0.001,
        action_repeat=10,
        motor_control_mode="PD",
        task_env="NO_TASK",
        observation_space_mode="ENCODER",
        action_space_mode="SYMMETRIC",
        on_rack=False,
        render=False,
        enable_springs=False,
        enable_action_interpolation=False,
        enable_action_filter=False,
        env_randomizer_mode="GROUND_RANDOMIZER",
        camera_mode="CLASSIC",
        curriculum_level=0.0,
        verbose=0,
        device=0,
        seed=0,
        noise=True,   # extensions (not in the reference's signature): device, seed of the counter-based RNG, sensor noise on / off
        demo=None,    # DEMO tasks: the demonstration rows (array or .npy path) the reference would np.load (task_base.py:173)
        **solver_settings,   # the engine settings of qs_amd.config.build_config (friction_model, contact_erp, body_contacts, payload, ...)
    ):
        if on_rack or render:
            raise NotImplementedError("on_rack / render need the PyBullet GUI path, which this build does not provide")
        unknown = set(solver_settings) - {"friction_model", "contact_erp", "contact_slop", "joint_erp", "warmstart", "solver_residual_threshold",
                                          "body_contacts", "self_collision", "payload", "mass_inertia_rule"}
        if unknown:
            raise TypeError(f"unexpected keyword argument(s) {sorted(unknown)}")
        # ONE environment, stepped by a caller who may go on after `done` (no auto-reset here): every collision primitive pushes back, as in
        # the reference (quadruped.py:533-539).  The vectorised environment's "auto" leaves that response off under a task because a
        # launch of thousands waits for the one wave whose robot has just fallen; with one environment nobody waits.
        solver_settings.setdefault("body_contacts", True)
        self.verbose = verbose
        self._vec = QuadrupedVecEnv(
            num_envs=1, device=device, auto_reset=False, isRLGymInterface=isRLGymInterface, time_step=time_step,
            action_repeat=action_repeat, motor_control_mode=motor_control_mode, task_env=task_env,
            observation_space_mode=observation_space_mode, action_space_mode=action_space_mode, enable_springs=enable_springs,
            enable_action_interpolation=enable_action_interpolation, enable_action_filter=enable_action_filter,
            env_randomizer_mode=env_randomizer_mode, seed=seed, noise=noise, demo=demo, **solver_settings)
        meta = self._vec.meta
        self._robot_config = meta["robot_config"]
        self._enable_springs = enable_springs
        self._isRLGymInterface = isRLGymInterface
        self.sim_time_step = time_step
        self._action_repeat = action_repeat
        self.env_time_step = action_repeat * time_step
        self._enable_action_filter = enable_action_filter
        self._enable_action_interpolation = enable_action_interpolation
        self._num_bullet_solver_iterations = int(300 / action_repeat)
        self._MAX_EP_LEN = EPISODE_LENGTH
        self._settling_steps = 2500
        self.task_env = task_env
        self._motor_control_mode = motor_control_mode
        self._action_space_mode = action_space_mode
        self._observation_space_mode = observation_space_mode
        self._env_randomizer_mode = env_randomizer_mode
        self.curriculum_level = curriculum_level
        self.action_dim = self._vec.action_dim
        self.action_space = self._vec.action_space
        self.observation_space = self._vec.observation_space
        self._replay_row = None
        self.robot_desired_state = None
        self.robot = _RobotView(self)
        self.robot._motor_model = _MotorModelView(self)
        self.task = _TaskView(self)
        self._ac_interface = _ActionInterfaceView(self)
        self._keys, self._dims = meta["layout"]["keys"], meta["layout"]["dims"]
        self._last_action = np.zeros(self.action_dim)
        self._settling_action = meta["settle_action"]
        self.sub_step_callback = None
        if self.verbose > 0:
            self.print_info()

    def _as_dict(self, flat):
        obs, n = {}, 0
        for k, d in zip(self._keys, self._dims):  # sensor.py:107-111
            obs[k] = flat[n:n + d].astype(np.float64)
            n += d
        return obs

    def reset(self):
        if self.robot_desired_state is not None:   # gym_env.py:289-290, quadruped.py:521-525: no settle, _last_action stays zero
            _, q, qd, pos, quat, lin, ang, _ = self.robot_desired_state
            st = np.concatenate([pos, quat, lin, ang, q, qd]).astype(np.float32)[None]
            keep = self.task.demo_counter if self._vec.demo_list is not None else None   # task_base.py:177-179: the counter survives this reset
            flat = self._vec.reset_tensor(states=st)[0].cpu().numpy()
            if keep is not None:
                self._vec.set_demo_counter(keep)
            self._last_action = np.zeros(self.action_dim)
            return self._as_dict(flat)
        flat = self._vec.reset()[0]
        self._last_action = np.asarray(self._settling_action, float)
        return self._as_dict(flat)

    def get_reward_end_episode(self):
        """gym_env.py:363-365: the task's end-of-episode bonus / malus for the current state."""
        return float(self._vec.get_info("reward_end")[0, 0])

    def set_robot_desired_state(self, state):
        """gym_env.py:400-402: the 8-tuple of GetDemonstrationWrapper.read_demo (action, q, qd, base position, base quaternion, linear
        velocity, angular velocity, landing flag), or None; the next reset() places the robot there instead of settling it."""
        self.robot_desired_state = state

    def step(self, action):
        a = np.asarray(action, dtype=np.float32).reshape(1, self.action_dim)
        self._last_action = a[0].astype(np.float64).copy()
        obs, rew, done, infos = self._vec.step(a)
        if self.sub_step_callback is not None:   # gym_env.py:207-216 fires it after every physics substep
            try:
                for row in self._vec.get_trace(as_dict=False):
                    self._replay_row = row
                    self.sub_step_callback()
            finally:
                self._replay_row = None
        info = {}
        if done[0]:
            info["TimeLimit.truncated"] = infos[0]["TimeLimit.truncated"]
        return self._as_dict(obs[0]), float(rew[0]), bool(done[0]), info

    def render(self, mode="rgb_array"):
        return None

    def close(self):
        self._vec.close()

    # ---- getters used by the reference's wrappers (gym_env.py:343-426)
    def get_observation(self):
        import ctypes as C
        from .. import lib as _lib
        out = self._vec.torch.empty((1, self._vec.obs_dim), dtype=self._vec.torch.float32, device=self._vec.device)
        _lib.check(self._vec.lib.qs_get_obs(self._vec.h, C.c_void_p(out.data_ptr())))
        return self._as_dict(out[0].cpu().numpy())

    def get_sim_time(self):
        if self._replay_row is not None:
            return float(self._replay_row[0])
        return float(self._vec.get_info("counters")[0, 0].item()) * self.sim_time_step

    def get_ac_interface(self):
        return self._ac_interface

    def get_last_filtered_action(self):
        return self._vec.get_info("filtered_action")[0, : self.action_dim].cpu().numpy().astype(np.float64) if self._enable_action_filter \
            else self._last_action

    def get_motor_control_mode(self):
        return self._motor_control_mode

    def get_robot_config(self):
        return self._robot_config

    def are_springs_enabled(self):
        return self._enable_springs

    def get_init_pose(self):
        return self._vec.meta["init_pose"]

    def get_settling_action(self):
        return self._settling_action

    def get_landing_action(self):
        return self._vec.meta["landing_action"]

    def get_last_action(self):
        return self._last_action

    def get_observation_space_mode(self):
        return self._observation_space_mode

    def get_curriculum_level(self):
        return self.curriculum_level

    def get_randomizer_mode(self):
        return self._env_randomizer_mode

    def get_quadruped_config(self):
        """gym_env.py:395-397: the keyword set the reference builds its Quadruped with (no Bullet client here)."""
        return dict(pybullet_client=None, robot_config=self._robot_config, motor_control_mode="PD" if self._motor_control_mode != "TORQUE" else "TORQUE",
                    on_rack=False, render=False, enable_springs=self._enable_springs, desired_state=self.robot_desired_state)

    def reinit_randomizers(self, env):
        """gym_env.py:411-413 re-points the Python randomizers at a wrapping env; the randomizers live in the reset kernel here."""

    def reinit_sensors(self, env):
        """gym_env.py:415-417, same for the sensors (they are part of the step kernel's epilogue)."""

    def increase_curriculum_level(self, value):
        """gym_env.py:423-426; only the *_CURRICULUM randomizers react to it in the reference, and those are not selectable
        (env_randomizer_collection.py:15-21 never hands them a working env: SURVEY App. C-4), so the level is just recorded."""
        assert 0 <= value <= 1, "curriculum level change should be in [0,1]."
        self.curriculum_level = min(1.0, self.curriculum_level + value)

    def set_sub_step_callback(self, callback):
        """evaluation_wrapper.py:14: the callback reads the robot after every physics substep.  The substeps are fused on the
        device, so the per-substep trace tap records them and step() replays the rows: while the callback runs, get_sim_time()
        and the robot's state getters answer for that substep."""
        self.sub_step_callback = callback
        self._vec.set_trace(0 if callback is not None else None)

    def print_info(self):
        print("\n*** Environment Info ***")
        print(f"task environment -> {self.task_env}")
        print(f"spring enabled -> {self._enable_springs}")
        print(f"low-pass action filter > {self._enable_action_filter}")
        print(f"sensors -> {self._observation_space_mode}")
        print(f"env randomizer -> {self._env_randomizer_mode}")
        print("")
