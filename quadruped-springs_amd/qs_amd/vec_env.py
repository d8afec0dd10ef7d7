"""QuadrupedVecEnv: N Go1 environments advanced by one HIP kernel launch per step.

Duck-types stable_baselines3.common.vec_env.VecEnv as the reference's consumer uses it (load_model.py:109-137:
num_envs, observation_space, action_space, reset, step_async/step_wait/step, auto-reset with
infos[i]["terminal_observation"] and infos[i]["TimeLimit.truncated"], get_attr/set_attr/env_method/env_is_wrapped, seed).
State lives on the GPU; `step_tensor` keeps actions / observations there for on-device learners."""
import ctypes as C

import numpy as np

from . import lib as _lib
from .config import DEMO_FILES, OBSERVATION_EPS, build_config
from .spaces import Box, SB3VecEnv

INFO = dict(foot_force=0, foot_contact=1, torque=2, spring_torque=3, task=4, n_invalid=5, params=6, counters=7,
            last_action=8, terminal_obs=9, wrapper=10, filtered_action=11, reward_end=12, payload_block=13)
PHASE = ("policy", "take_off", "landing", "rest")
PARAM = dict(mu=0, spring_k=1, spring_b=2, kp=3, kd=4, all=5)


class QuadrupedVecEnv(SB3VecEnv):
    def __init__(self, num_envs=1, device=0, auto_reset=True, reset_lookahead=None, copy_outputs=True, **env_kwargs):
        """reset_lookahead = K: every environment keeps the settled reset states of its next K episodes ready (computed by extra workgroups
        of the step kernel while the environments step), so a reset is a copy; results are bitwise those of K = 0, where every reset runs
        the reference's 2500-substep settle in place (gym_env.py:278-297, 323-329).  Default: 16 with auto_reset, 0 without.  What K = 16
        costs: N x 16 x 1152 bytes of slots plus five staging slices of min(2 N, 131072) records (151 + 94 MB at N = 8192, 1.2 + 0.75 GB at
        N = 65536) and N x 16 settles inside the constructor (0.1 s at N = 8192, 0.8 s at 65536); every launch also carries the settle
        lanes' workgroups of five full cohorts, of which those beyond the cohort's jobs leave at once (5120 of them at N = 8192: 0.4 % of the
        launch; 40960 next to 4096 working ones at N = 65536: under 1 %).  A handle whose episodes never end early can do with K = 2.
        copy_outputs=False: step() / step_wait() return views of the page-locked result block instead of copies (valid until the end of
        the next step: two blocks alternate)."""
        cfg, meta = build_config(n_envs=num_envs, auto_reset=auto_reset, **env_kwargs)
        cfg.reset_lookahead = int((16 if auto_reset else 0) if reset_lookahead is None else reset_lookahead)
        self._setup(cfg, meta, device, copy_outputs)

    @classmethod
    def from_config(cls, cfg, meta, device=0, copy_outputs=True, load_demo=True):
        """A handle for a qs_config built (and possibly edited) by the caller: build_config(...) -> (cfg, meta).  load_demo=False leaves
        the demonstration of a DEMO task to a later set_demo()."""
        self = cls.__new__(cls)
        self._setup(cfg, meta, device, copy_outputs, load_demo)
        return self

    def _setup(self, cfg, meta, device, copy_outputs, load_demo=True):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("QuadrupedVecEnv needs a HIP device (torch.cuda.is_available() is False); there is no CPU path")
        self.torch = torch
        self.lib = _lib.load()
        self.cfg, self.meta = cfg, meta
        self.num_envs = int(cfg.n_envs)
        self.device = torch.device("cuda", device)
        lay = self.meta["layout"]
        self.observation_space = Box(lay["low"] - OBSERVATION_EPS, lay["high"] + OBSERVATION_EPS, dtype=np.float32)  # gym_env.py:160-164
        d = self.cfg.action_dim
        self.action_space = Box(-np.ones(d), np.ones(d), dtype=np.float32)                                       # gym_env.py:179-182
        self.action_dim, self.obs_dim = d, self.cfg.obs_dim
        self._init_vec_env_base()
        self.h = C.c_void_p()
        self._closed = False
        _lib.check(self.lib.qs_create(C.byref(self.cfg), device, C.byref(self.h)))
        n, o = self.num_envs, self.obs_dim
        with torch.cuda.device(self.device):
            self._obs = torch.zeros((n, o), dtype=torch.float32, device=self.device)
            self._rew = torch.zeros(n, dtype=torch.float32, device=self.device)
            self._done = torch.zeros(n, dtype=torch.uint8, device=self.device)
            self._trunc = torch.zeros(n, dtype=torch.uint8, device=self.device)
            self._act = torch.zeros((n, d), dtype=torch.float32, device=self.device)
        self._views, self._infos, self._dirty = {}, [{} for _ in range(self.num_envs)], []
        self._terminal_hook = None     # DeviceVecNormalize.step_async: normalises the per-environment terminal observations (overflow of the compact list)
        self.copy_outputs = bool(copy_outputs)
        self._trace = None
        self.render_mode = None
        self.demo_list, self.demo_length = None, 0
        if load_demo and self.meta["task_env"] in DEMO_FILES:
            if self.meta["demo"] is None:
                self.close()
                raise ValueError(f"task {self.meta['task_env']} imitates a demonstration: pass demo=<array [L, action_dim + 38] or path of the "
                                 f".npy> (the reference loads demonstrations/{DEMO_FILES[self.meta['task_env']]}, task_base.py:173)")
            self.set_demo(self.meta["demo"])

    def _init_vec_env_base(self):
        """stable_baselines3.common.vec_env.VecEnv.__init__(num_envs, observation_space, action_space) when SB3 is importable and this
        class therefore IS a VecEnv (qs_amd/spaces.py): SB3's wrappers (VecNormalize.load(stats, env), load_model.py:120) read what the base
        class sets up, and isinstance(env, VecEnv) holds.  Without SB3 the base is `object` and there is nothing to call."""
        if SB3VecEnv is not object:
            SB3VecEnv.__init__(self, self.num_envs, self.observation_space, self.action_space)

    # ---- plumbing
    def _stream(self):
        s = self.torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self.lib.qs_set_stream(self.h, C.c_void_p(s)))

    def _ptr(self, t):
        return C.c_void_p(t.data_ptr())

    def close(self):
        if not getattr(self, "_closed", True) and self.h:
            h, self.h = self.h, C.c_void_p()     # later calls reach the library with a null handle and fail with its error text
            self.lib.qs_destroy(h)
            self._closed = True

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # ---- device-resident API
    def reset_tensor(self, mask=None, states=None):
        """reset() of the masked environments (all when mask is None).  With `states` ([N,37], layout of get_state) the robots are
        placed there instead of being spawned and settled: reference-state initialisation (set_robot_desired_state)."""
        self._stream()
        m = None
        if mask is not None:
            m = self.torch.as_tensor(mask, device=self.device).to(self.torch.uint8).contiguous()
        if states is not None:
            st = self.torch.as_tensor(states, dtype=self.torch.float32, device=self.device).reshape(self.num_envs, 37).contiguous()
            _lib.check(self.lib.qs_reset_to(self.h, None if m is None else self._ptr(m), self._ptr(st)))
        else:
            _lib.check(self.lib.qs_reset(self.h, None if m is None else self._ptr(m)))
        _lib.check(self.lib.qs_get_obs(self.h, self._ptr(self._obs)))
        return self._obs

    def step_tensor(self, actions):
        """actions: float32 CUDA tensor [N, action_dim] -> (obs, rew, done, truncated) CUDA tensors (reused buffers)."""
        a = actions
        if a.dtype != self.torch.float32 or not a.is_contiguous() or a.device != self.device:
            a = a.to(device=self.device, dtype=self.torch.float32).contiguous()
        if a.shape != (self.num_envs, self.action_dim):
            raise ValueError(f"actions must have shape {(self.num_envs, self.action_dim)}, got {tuple(a.shape)}")
        self._stream()
        _lib.check(self.lib.qs_step(self.h, self._ptr(a), self._ptr(self._obs), self._ptr(self._rew), self._ptr(self._done), self._ptr(self._trunc)))
        return self._obs, self._rew, self._done, self._trunc

    def step_fused(self, actions, out):
        """One step with a single output: `out` [N, obs_dim + 2] float32 CUDA tensor = observation | reward | done + 2 * truncated
        (the row a sharded run all-gathers, qs_amd/sharded.py)."""
        t = self.torch
        for name, x, shape in (("actions", actions, (self.num_envs, self.action_dim)), ("out", out, (self.num_envs, self.obs_dim + 2))):
            if tuple(x.shape) != shape or x.dtype != t.float32 or x.device != self.device or not x.is_contiguous():
                raise ValueError(f"{name} must be a contiguous float32 tensor of shape {shape} on {self.device}")
        self._stream()
        _lib.check(self.lib.qs_step_fused(self.h, self._ptr(actions), self._ptr(out)))
        return out

    def get_state(self):
        out = self.torch.empty((self.num_envs, 37), dtype=self.torch.float32, device=self.device)
        self._stream()
        _lib.check(self.lib.qs_get_state(self.h, self._ptr(out)))
        return out

    def set_state(self, state):
        s = self.torch.as_tensor(state, dtype=self.torch.float32, device=self.device).contiguous().reshape(self.num_envs, 37)
        self._stream()
        _lib.check(self.lib.qs_set_state(self.h, self._ptr(s)))

    def get_info(self, which):
        k = INFO[which] if isinstance(which, str) else int(which)
        dim = self.lib.qs_info_dim(self.h, k)
        out = self.torch.empty((self.num_envs, dim), dtype=self.torch.float32, device=self.device)
        self._stream()
        _lib.check(self.lib.qs_get_info(self.h, k, self._ptr(out)))
        return out

    def set_params(self, which, values):
        k = PARAM[which] if isinstance(which, str) else int(which)
        v = self.torch.as_tensor(values, dtype=self.torch.float32, device=self.device).contiguous()
        self._stream()
        _lib.check(self.lib.qs_set_params(self.h, k, self._ptr(v)))
        self.torch.cuda.current_stream(self.device).synchronize()  # `v` may be a temporary

    def stats(self):
        a, b = C.c_uint64(), C.c_uint64()
        _lib.check(self.lib.qs_stats(self.h, C.byref(a), C.byref(b)))
        return dict(settle_substeps=a.value, resets=b.value)

    COUNTERS = dict(settle_substeps=0, resets=1, lookahead_served=2, lookahead_settled=3, limit_path_substeps=4, self_narrow_substeps=5,
                    reset_stalls=6, lookahead_backlog=7)

    def counter(self, which):
        v = C.c_uint64(0)
        _lib.check(self.lib.qs_counter(self.h, self.COUNTERS[which] if isinstance(which, str) else int(which), C.byref(v)))
        return int(v.value)

    def counters_snapshot(self):
        """The counters as they stand at this point of the stream, in a device tensor [8] (int64; index = COUNTERS, entry 7 unused) -- no
        synchronisation: read it (.cpu()) whenever convenient."""
        out = self.torch.zeros(8, dtype=self.torch.int64, device=self.device)
        self._stream()
        _lib.check(self.lib.qs_counters_async(self.h, self._ptr(out)))
        return out

    def enable_timing(self, on=True):
        """True / False: arm / disarm the batch timing of the step kernel; 2: close the batch now (the closing event goes onto the stream,
        last_step_kernel_ms() waits for it whenever it is called)."""
        self._stream()
        _lib.check(self.lib.qs_enable_timing(self.h, int(on)))

    def last_step_kernel_ms(self):
        ms = C.c_float()
        _lib.check(self.lib.qs_last_step_kernel_ms(self.h, C.byref(ms)))
        return ms.value

    TRACE_FIELDS = dict(time=(0, 1), base_position=(1, 4), base_quaternion=(4, 8), base_linear_velocity=(8, 11), base_angular_velocity=(11, 14),
                        joint_angles=(14, 26), joint_velocities=(26, 38), torques=(38, 50), spring_tau=(50, 62), feet_normal_forces=(62, 66),
                        feet_in_contact=(66, 70))

    def set_trace(self, env=0):
        """Per-substep tap on one environment (the reference's set_sub_step_callback users: evaluation_wrapper.py:14,
        monitor_state.py:66-85).  env=None switches it off.  Read the rows of the last step with get_trace()."""
        if env is None:
            self._trace = None
            _lib.check(self.lib.qs_set_trace(self.h, -1, None))
            return
        self._trace = self.torch.zeros((self.cfg.action_repeat, 70), dtype=self.torch.float32, device=self.device)
        _lib.check(self.lib.qs_set_trace(self.h, int(env), self._ptr(self._trace)))

    def get_trace(self, as_dict=True):
        """Rows [action_repeat, 70] of the traced environment for the most recent step; with as_dict the monitor_state.py
        quantities, incl. the spring energy 0.5 k (q - q_rest)^2 with the gated stiffness of springs.py:34-61."""
        rows = self._trace.cpu().numpy().astype(np.float64)
        if not as_dict:
            return rows
        out = {k: rows[:, a:b] for k, (a, b) in self.TRACE_FIELDS.items()}
        out["time"] = out["time"][:, 0]
        return out

    # ---- DEMO tasks (task_base.py:169-220)
    def set_demo(self, rows):
        """The demonstration the DEMO task imitates: [L, action_dim + 38] rows as demo_rows() / GetDemonstrationWrapper record them."""
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if rows.ndim != 2 or rows.shape[1] != self.action_dim + 38:
            raise ValueError(f"demonstration rows have {self.action_dim + 38} entries, got shape {rows.shape}")
        t = self.torch.from_numpy(rows).to(self.device)
        self._stream()
        _lib.check(self.lib.qs_set_demo(self.h, self._ptr(t), int(rows.shape[0])))   # synchronises: `t` may go
        self.demo_list, self.demo_length = rows, int(rows.shape[0])

    def set_demo_counter(self, values, mask=None):
        """task.set_demo_counter (task_base.py:219-220) of the masked environments; after reset_tensor(mask, states=...)."""
        t = self.torch
        v = t.as_tensor(values, device=self.device).to(t.int32).expand(self.num_envs).contiguous()
        m = None if mask is None else t.as_tensor(mask, device=self.device).to(t.uint8).contiguous()
        self._stream()
        _lib.check(self.lib.qs_set_demo_counter(self.h, None if m is None else self._ptr(m), self._ptr(v)))

    def demo_counter(self):
        return self.get_info("task")[:, 44].to(self.torch.int64)

    @staticmethod
    def demo_states(rows, action_dim):
        """[K, 37] rigid-body states (layout of get_state) of demonstration rows (read_demo order: a, q, qd, pos, quat, vlin, vang, flag)."""
        r = np.atleast_2d(np.asarray(rows, dtype=np.float32))
        d = action_dim
        return np.concatenate([r[:, d + 24:d + 37], r[:, d:d + 24]], axis=1)

    # ---- demonstration rows (get_demonstration_wrapper.py:35-70)
    def demo_rows(self, done=None):
        """[N, d + 38] on the device: the row GetDemonstrationWrapper._get_demo records after a step -- filtered action (d), joint
        angles 12, joint velocities 12, base position 3, quaternion 4, linear velocity 3, angular velocity 3, landing_started 1
        (latched once the controller has switched and vz <= 0; cleared when the episode ends).  Call it once per step with that
        step's `done` (tensor or array; default: the buffer step_tensor filled)."""
        t = self.torch
        st = self.get_state()
        if getattr(self, "_landing_started", None) is None:
            self._landing_started = t.zeros(self.num_envs, dtype=t.bool, device=self.device)
        switched = self.get_info("task")[:, 0] > 0.5
        self._landing_started |= switched & (st[:, 9] <= 0.0)
        act = (self.get_info("filtered_action") if self.cfg.enable_filter else self.get_info("last_action"))[:, : self.action_dim]
        rows = t.cat([act, st[:, 13:37], st[:, 0:13], self._landing_started.to(t.float32)[:, None]], dim=1)
        d = self._done.bool() if done is None else t.as_tensor(np.asarray(done) if not t.is_tensor(done) else done, device=self.device).bool()
        self._landing_started &= ~d
        return rows

    @staticmethod
    def read_demo(demo, action_dim=6, num_joints=12):
        """get_demonstration_wrapper.py:61-70: split one row into (action, q, qd, base_pos, base_quat, lin_vel, ang_vel, landing)."""
        cuts = np.cumsum([action_dim, num_joints, num_joints, 3, 4, 3, 3, 1])
        return [demo[a:b] for a, b in zip(np.concatenate(([0], cuts[:-1])), cuts)]

    def settle_lanes(self, on=True):
        """Switch the settle lanes of the look-ahead resets on / off (qs_settle_lanes; on from construction).  Off: resets use up the
        states that are ready, then settle in place."""
        self._stream()
        _lib.check(self.lib.qs_settle_lanes(self.h, int(bool(on))))

    # ---- SB3 VecEnv surface (numpy)
    def reset(self):
        return self.reset_tensor().cpu().numpy().copy()

    def step_async(self, actions):
        """VecEnv.step_async: the actions go to the device and the step and the copy of its results back are enqueued (qs_host_step_begin);
        nothing is waited for."""
        a = np.ascontiguousarray(actions, dtype=np.float32)
        if a.size != self.num_envs * self.action_dim:
            raise ValueError(f"actions must have shape {(self.num_envs, self.action_dim)}, got {a.shape}")
        self._stream()
        rc = self.lib.qs_host_step_begin(self.h, a.ctypes.data_as(C.c_void_p))
        if rc != 0:
            msg = self.lib.qs_last_error().decode()
            # a failure BEHIND the step's launch leaves the step pending (include/qs_amd.h): close it, so that the next step_async starts clean
            self.lib.qs_host_step_end(self.h, C.byref(_lib.HostResult()))
            raise RuntimeError("qs_amd: " + msg)

    def _host_views(self, res):
        """numpy views of the result block qs_host_step_end points at (one set per host block: two alternate)."""
        key = res.obs
        views = self._views.get(key)
        if views is None:
            n, o = self.num_envs, self.obs_dim
            f32, u8 = C.POINTER(C.c_float), C.POINTER(C.c_uint8)
            views = (np.ctypeslib.as_array(C.cast(res.obs, f32), shape=(n, o)), np.ctypeslib.as_array(C.cast(res.rew, f32), shape=(n,)),
                     np.ctypeslib.as_array(C.cast(res.done, u8), shape=(n,)).view(np.bool_),
                     np.ctypeslib.as_array(C.cast(res.truncated, u8), shape=(n,)).view(np.bool_),
                     np.ctypeslib.as_array(C.cast(res.terminal_rows, f32), shape=(res.terminal_cap, o + 1)))
            self._views[key] = views
        return views

    def _entry(self, old, truncated, terminal_observation):
        """the info dict of an environment whose episode ended in this step: a new one under copy_outputs, the old one refilled otherwise"""
        if self.copy_outputs:
            return {"TimeLimit.truncated": truncated, "terminal_observation": terminal_observation}
        old["TimeLimit.truncated"] = truncated; old["terminal_observation"] = terminal_observation
        return old

    def step_wait(self):
        """VecEnv.step_wait: (obs [N, o] float32, rewards [N] float32, dones [N] bool, infos) -- the SB3 convention of load_model.py:113-133:
        finished environments are already reset, their last observation is infos[i]["terminal_observation"], infos[i]["TimeLimit.truncated"]
        says whether the time limit ended the episode (gym_env.py:246).  One H2D copy, the step, ONE D2H copy into page-locked host memory;
        the arrays returned are copies of it (copy_outputs=False: views, valid until the end of the NEXT step).  `infos` is ONE list
        of N dicts kept by the environment, of which a step touches only the entries of environments that have (or had) something to
        say.  With copy_outputs=True (the default) such an entry gets a NEW dict: a dict handed out by an earlier step is never changed,
        so a consumer may keep the ones it cares about (a replay buffer that stores terminal observations); the LIST is the same object
        every step -- copy it (list(infos)) to keep a whole step's worth.  copy_outputs=False reuses the dicts in place as well."""
        res = _lib.HostResult()
        _lib.check(self.lib.qs_host_step_end(self.h, C.byref(res)))
        obs, rew, done, trunc, term = self._host_views(res)
        if self.copy_outputs:
            obs, rew, done = obs.copy(), rew.copy(), done.copy()
        infos = self._infos
        if self.copy_outputs:
            for i in self._dirty:
                infos[i] = {}
        else:
            for i in self._dirty:
                infos[i].clear()
        dirty = self._dirty = []
        idx = np.flatnonzero(done)
        if idx.size:
            if not self.cfg.auto_reset:
                for i in idx.tolist():
                    infos[i] = self._entry(infos[i], bool(trunc[i]), obs[i].copy())
            else:
                rows = term[: idx.size] if idx.size <= term.shape[0] else None
                if rows is None:    # more episode ends than the compact list holds: the per-environment array
                    full = self.get_info("terminal_obs").cpu().numpy()
                    if self._terminal_hook is not None:      # DeviceVecNormalize: the per-environment array holds raw observations
                        full = self._terminal_hook(full)
                    for i in idx.tolist():
                        infos[i] = self._entry(infos[i], bool(trunc[i]), full[i])
                else:
                    rows = rows.copy()
                    envs = rows[:, 0].view(np.int32)
                    for k, i in enumerate(envs.tolist()):
                        infos[i] = self._entry(infos[i], bool(trunc[i]), rows[k, 1:])
            dirty.extend(idx.tolist())
        if self.cfg.wrapper_mode:
            # the reference's LandingWrapper / GoToRestWrapper loop over env.step inside one wrapper.step; here every inner
            # step is one launch and the ones whose action was scripted are flagged, so a learner can mask them out.  Only the
            # environments in a scripted phase get the keys (read them with infos[i].get("scripted", False)).
            w = self.get_info("wrapper").cpu().numpy()
            for i in np.nonzero((w[:, 0] > 0.5) | (w[:, 1] > 0.5))[0].tolist():
                if self.copy_outputs and not infos[i]:
                    infos[i] = {}            # (an empty dict may have been handed out by an earlier step: it stays empty)
                infos[i]["scripted"] = bool(w[i, 1])
                infos[i]["phase"] = PHASE[int(w[i, 0])]
                dirty.append(i)
        return obs, rew, done, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def seed(self, seed=None):
        """VecEnv.seed(seed) (SB3: each sub-environment gets seed + index; load_model.py's make_vec_env(seed=...) calls it right after the
        constructor).  Randomness here is counter based -- Philox streams keyed by (seed, global environment id, episode / step), which
        also key the reset states settled ahead of time -- so a new seed means a new device handle: the old one is destroyed and one
        with cfg.seed = seed created in its place.  EVERY ENVIRONMENT IS UNRESET AFTERWARDS, as after the constructor: call reset()
        next (what SB3's own flow does).  seed=None leaves everything as it is.  Returns SB3's list: seed + i for environment i (None s
        when nothing was changed)."""
        if seed is None:
            return [None] * self.num_envs
        seed = int(seed)
        if seed < 0:
            raise ValueError(f"seed must be a non-negative integer, got {seed}")
        h, self.h = self.h, C.c_void_p()
        if h:
            self.lib.qs_destroy(h)
        self.cfg.seed = seed
        self._views, self._dirty = {}, []
        self._infos = [{} for _ in range(self.num_envs)]
        _lib.check(self.lib.qs_create(C.byref(self.cfg), self.device.index or 0, C.byref(self.h)))
        if self.demo_list is not None:
            self.set_demo(self.demo_list)
        self._trace = None
        return [seed + i for i in range(self.num_envs)]

    def get_attr(self, attr_name, indices=None):
        return [getattr(self, attr_name)] * len(self._indices(indices))

    def set_attr(self, attr_name, value, indices=None):
        setattr(self, attr_name, value)

    def env_method(self, method_name, *args, indices=None, **kwargs):
        return [getattr(self, method_name)(*args, **kwargs)] * len(self._indices(indices))

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * len(self._indices(indices))

    def get_images(self):
        return [None] * self.num_envs

    def render(self, mode="rgb_array"):
        return None

    def _indices(self, indices):
        if indices is None:
            return list(range(self.num_envs))
        return [indices] if isinstance(indices, int) else list(indices)
