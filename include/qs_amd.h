/*
 * qs_amd.h -- C ABI of the MI355X-native batched Go1 + PEA simulation step.
 *
 * This is the drop-in boundary for the hot path of francescovezzi/quadruped-springs.  The reference has no FFI:
 * the path sits behind the Python class QuadrupedGymEnv, so every entry point names the Python-level interface
 * it replaces (paths relative to the reference's quadruped_spring/ package):
 *
 *   qs_create      QuadrupedGymEnv.__init__                      env/quadruped_gym_env.py:52-155
 *   qs_reset       QuadrupedGymEnv.reset (+ randomizers, settle)  env/quadruped_gym_env.py:278-329,
 *                                                                 env/env_randomizers/env_randomizer.py:19-122,279-291
 *   qs_step        QuadrupedGymEnv.step for N environments        env/quadruped_gym_env.py:227-256
 *                  (filter, action map, ApplyAction = PD + PEA,   utils/action_filter.py:110-121,
 *                   stepSimulation x action_repeat, task,          env/quadruped.py:288-320, env/quadruped_motor.py:45-104,
 *                   reward, termination, sensors)                  env/springs.py:28-74, env/tasks/, env/sensors/
 *   qs_get_obs     QuadrupedGymEnv.get_observation                env/quadruped_gym_env.py:343-345
 *   qs_get_state / qs_set_state    Quadruped state getters / reset_desired_state   env/quadruped.py:107-207, 521-525
 *   qs_get_info    GetContactInfo, GetMotorTorques, task scalars  env/quadruped.py:209-258, env/tasks/task_base.py:44-59
 *   qs_set_params  set_spring_stiffness/damping, changeDynamics(lateralFriction), kp/kd swaps of the landing wrappers
 *                                                                 env/quadruped.py:732-742, env/wrappers/landing_wrapper.py:22-30
 *
 * Conventions
 *   - plain pointers and sizes only; all array arguments are DEVICE pointers (HIP) owned by the caller, row-major,
 *     float32 unless stated; the handle owns the persistent per-environment records.
 *   - returns 0 on success, a negative code on failure with text in qs_last_error(); never throws.
 *   - a handle is driven by one host thread; work is enqueued on the stream given to qs_set_stream (default: the
 *     null stream) and is asynchronous with respect to the host.  Handles are independent (one per GPU rank).
 *   - there is no CPU fallback: qs_create fails if no HIP device is usable.
 */
#ifndef QS_AMD_H
#define QS_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { QS_ACT_DEFAULT = 0, QS_ACT_SYMMETRIC = 1, QS_ACT_SYMMETRIC_NO_HIP = 2, QS_ACT_CPG = 3 /* build extension */ };  /* control_interface/collection.py:49 */
enum { QS_MOTOR_PD = 0, QS_MOTOR_CARTESIAN_PD = 1, QS_MOTOR_TORQUE = 2 };       /* control_interface/collection.py:33 */
enum {                                                                          /* tasks/task_collection.py:19-37 */
    QS_TASK_NO_TASK = 0, QS_TASK_JUMPING_IN_PLACE = 1, QS_TASK_JUMPING_FORWARD = 2,
    QS_TASK_CONT_JUMPING_FORWARD = 3, QS_TASK_CONT_JUMPING_FORWARD2 = 4,
    QS_TASK_JUMPING_IN_PLACE_PPO = 5, QS_TASK_JUMPING_FORWARD_PPO = 6, QS_TASK_BACKFLIP = 7,
    QS_TASK_JUMPING_IN_PLACE_PPO_HP = 8, QS_TASK_JUMPING_FORWARD_PPO_HP = 9, QS_TASK_BACKFLIP_PPO = 10,
    QS_TASK_CONT_JUMPING_FORWARD3 = 11, QS_TASK_CONT_JUMPING_FORWARD_PPO = 12,
    /* imitation tasks (TaskJumpingDemo task_base.py:169-220, TaskJumpingDemo2 :402-453; robot_tasks.py:222-247): the rows come
     * through qs_set_demo, not from the demonstrations/<name>.npy files the reference loads (its repository does not hold them) */
    QS_TASK_JUMPING_IN_PLACE_DEMO = 13, QS_TASK_JUMPING_FORWARD_DEMO = 14, QS_TASK_BACKFLIP_DEMO = 15,
    QS_TASK_CONT_JUMPING_FORWARD_DEMO = 16,
};
enum {                                                                          /* sensors/robot_sensors.py */
    QS_SENS_JOINT_POS = 0, QS_SENS_JOINT_VEL = 1, QS_SENS_PITCH = 2, QS_SENS_HEIGHT = 3, QS_SENS_VEL_Z = 4,
    QS_SENS_LANDING = 5, QS_SENS_JUMPING = 6, QS_SENS_PITCH_RATE = 7, QS_SENS_VEL_X = 8, QS_SENS_BOOL_CONTACT = 9,
    QS_SENS_LIN_VEL = 10, QS_SENS_ANG_VEL = 11, QS_SENS_FEET_POS = 12, QS_SENS_FEET_VEL = 13,
    QS_SENS_PITCH_BACKFLIP = 14, QS_SENS_RPY = 15, QS_SENS_QUAT = 16,
};
#define QS_RAND_GROUND 1   /* env_randomizer.py:279-291 */
#define QS_RAND_MASSES 2   /* env_randomizer.py:19-83 */
#define QS_RAND_SPRINGS 4  /* env_randomizer.py:86-122 */
#define QS_RAND_KEEP 8     /* reset keeps the parameters last written by qs_set_params */
enum { QS_WRAP_NONE = 0, QS_WRAP_LANDING = 1, QS_WRAP_GO_TO_REST = 2, QS_WRAP_LANDING2 = 3, QS_WRAP_LANDING_BACKFLIP = 4,
       QS_WRAP_LANDING_BACKFLIP2 = 5, QS_WRAP_LANDING_CONTINUOUS = 6 };
enum { QS_PHASE_POLICY = 0, QS_PHASE_TAKEOFF = 1, QS_PHASE_LANDING = 2, QS_PHASE_REST = 3 };

#define QS_MAX_SENSORS 16
#define QS_MAX_OBS 64
#define QS_STATE_DIM 37    /* pos3 quat4(xyzw) vlin3 vang3 q12 qd12 */
#define QS_PARAM_DIM 24    /* mu, k3, b3, rest3, kp3, kd3, m_trunk, m_leg3, m_payload, r_payload3 */
#define QS_TASK_DIM 48
#define QS_TRACE_DIM 70 /* floats per substep row of the trace tap, see qs_set_trace */    /* 32 task scalars, pose cache 9 (pos, vel, rpy), n_invalid, foot-force sum, sim_step, 4 spare */

/* Keyword arguments of QuadrupedGymEnv.__init__ (gym_env.py:52-70) resolved to numbers by the host
 * (quadruped-springs_amd/qs_amd/config.py); constants come from go1/configs_go1_*.py. */
typedef struct qs_config {
    int32_t n_envs;
    int32_t action_dim;          /* 12 / 6 / 4 (action_interface.py:12,27,56); 5 for the CPG layer */
    int32_t action_space_mode;
    int32_t motor_control_mode;
    int32_t symm_idx;            /* motor_interface.py:15,56 */
    int32_t rl_interface;        /* isRLGymInterface */
    int32_t task;
    int32_t n_sensors;
    int32_t sensors[QS_MAX_SENSORS];
    int32_t obs_dim;
    int32_t enable_springs;
    int32_t enable_filter;
    int32_t enable_interp;
    int32_t action_repeat;
    int32_t solver_iters;        /* int(300 / action_repeat), gym_env.py:113 */
    int32_t settle_steps;        /* 2500, gym_env.py:115 */
    int32_t max_sim_steps;       /* truncation when sim_step_counter > max_sim_steps (sim time > 10 s, gym_env.py:245) */
    int32_t randomizer_flags;
    int32_t noise_enabled;
    int32_t auto_reset;          /* SB3 VecEnv convention: finished environments are reset inside qs_step */
    int32_t reset_lookahead;     /* K: reset states kept ready per environment.  A reset's settled state depends on (seed, global environment
                                  * id, episode number) only, so the states of an environment's next K episodes are computed ahead of
                                  * time -- at qs_create for episodes 0 .. K-1, then one per reset by extra workgroups of the step kernel
                                  * (qs_settle_lanes) -- and a reset copies its own.  0: every reset runs the 2500-substep settle in place.
                                  * Either way the results are bitwise the same; K only decides when the settle work is done.  A reset
                                  * whose state is not ready (K consecutive episodes shorter than one settle) settles in place and is
                                  * counted (QS_COUNTER_RESET_STALLS).  Ignored under QS_RAND_KEEP.  N x K x 1152 bytes. */
    int32_t env_id_offset;       /* global id of environment 0 (sharded runs): RNG streams are keyed by the global id */
    int32_t wrapper_mode;        /* QS_WRAP_*: none, or one of the reference's six landing / go-to-rest wrappers as a per-environment phase machine */
    uint64_t seed;
    double dt;
    double filt_b[3], filt_a[3]; /* scipy.signal.butter(2, 3 Hz) at 1/env_dt, action_filter.py:191-213 */
    float gravity;
    float kp[3], kd[3], tau_max[3];
    float cmd_lo[12], cmd_hi[12];
    float settle_cmd[12];
    float settle_action[12];
    float spring_k[3], spring_b[3], spring_rest[3];
    float fallen_height;
    float leg_len[3];
    float contact_erp, joint_erp, warmstart, vel_cap;
    float obs_noise_std[QS_MAX_OBS];
    float task_p[16];
    float contact_slop;        /* btContactSolverInfo::m_linearSlop (PhysicsServerCommandProcessor sets 1e-5): added to a contact's distance
                                * before the positional / speculative error terms of its normal row */
    int32_t body_contacts;     /* 1: trunk / hip / thigh / calf primitives that touch the plane push back (normal + friction rows in the
                                * many-rows solve, at most two support points per leg); 0: they only count as invalid contacts (quadruped.py:243-249) */
    int32_t self_collision;    /* 1: link-link contacts that involve a calf are detected and counted as invalid contacts
                                * (URDF_USE_SELF_COLLISION quadruped.py:533-539, rule :237-241); 0: no link-link test */
    int32_t info_fields;       /* 1: every step also writes the info block of the records (foot forces and flags, motor and spring torque, the
                                * task's pose cache: what QS_INFO_FOOT_FORCE / _FOOT_CONTACT / _TORQUE / _SPRING_TORQUE and the cache slots of
                                * QS_INFO_TASK return); 0: a learner that reads observations, rewards and done flags only skips those stores,
                                * and the getters fail */
    int32_t payload_soft;      /* 0: the payload block of the mass randomizer is welded to the trunk (the default: same motion to micrometres,
                                * tests/test_body_contacts.py, and the common-path kernel); 1: a second body held by a six-row fixed constraint
                                * in the same PGS, as the reference builds it (quadruped.py:796-819): under the implicit cone its rows ride
                                * with the foot rows in the common-path solver (about 3x the step time: the constraint's rows need all the
                                * sweeps), under the friction pyramid every substep takes the many-rows solve; its state: QS_INFO_PAYLOAD_BLOCK */
    float support_margin;      /* m/s.  body_contacts: a non-foot support point inside its contact range gets its rows once its normal row comes this
                                * close to acting on the substep's predicted velocities (a point still approaching has speculative rows that end
                                * every sweep at zero impulse: the same solve at the many-rows price); >= 1e30: EVERY point in range
                                * gets its rows, as Bullet and the oracle build them (parity / debug runs).  Default 0.5. */
    float reserved_f[2];
    /* Hopf-oscillator CPG action layer (hopf_network.py:26-173); BASELINE.json configs[4] */
    float cpg_phi[16];         /* coupling phase matrix PHI[i][j] of the gait (hopf_network.py:74-115) */
    float cpg_lo[5], cpg_hi[5];/* action -> (omega_swing, omega_stance, mu, des_step_len, robot_height) */
    float cpg_clearance, cpg_penetration, cpg_coupling, cpg_alpha;  /* :42-43, :35, :142 */
    float solver_residual_threshold; /* PyBullet setPhysicsEngineParameter(solverResidualThreshold): a sweep whose largest
                                      * squared velocity change is <= this ends the solve; 0 = always `solver_iters` sweeps */
    int32_t friction_cone;     /* 0: friction pyramid, each direction clamped on its own and left alone while its normal impulse is zero
                                * (btMultiBodyConstraintSolver::solveSingleIteration without implicit cone friction); 1: the two friction
                                * rows of a contact are updated together and projected onto the disc of radius mu x normal impulse
                                * (resolveConeFrictionConstraintRows, PyBullet's enableConeFriction) */
    /* scripted phases of env/wrappers/landing_wrapper.py:18-69 and go_to_rest_wrapper.py:22-95 */
    float landing_action[12];  /* get_landing_action(), gym_env.py:375-379 */
    float landing_kp, landing_kd; /* landing_wrapper.py:22-27 */
    float rest_kp, rest_kd, rest_time; /* go_to_rest_wrapper.py:16-19, 26-32 */
    float reserved_h[3];
    /* Inertia tensors about the centre of mass PER UNIT MASS (xx, xy, xz, yy, yz, zz; link axes, sign convention of the FR leg) of
     * hip, thigh, calf and trunk: a link of mass m has the tensor m * unit_inertia.  mass_inertia_rule = "scale": the URDF tensor over the
     * URDF mass.  "collision_shape": what Bullet's changeDynamics(mass=...) (quadruped.py:761,776) leaves behind -- the box inertia of the
     * collision compound's AABB in the link's principal frame.  The host fills the table (qs_amd/config.py). */
    float unit_inertia[4][6];
} qs_config;

typedef struct qs_handle qs_handle;

int qs_create(const qs_config* cfg, int device, qs_handle** out);
void qs_destroy(qs_handle* h);
int qs_set_stream(qs_handle* h, void* hip_stream);
/* mask: device pointer to n_envs bytes, or NULL for all environments */
int qs_reset(qs_handle* h, const uint8_t* mask);
/* Reference-state initialisation (reference_state_initialization_wrapper.py:25-43 -> set_robot_desired_state, quadruped.py:521-525,
 * gym_env.py:289-290): reset of the masked environments that runs the randomizers, places the robot at states[env] ([N,37], layout of
 * qs_get_state) instead of spawning and settling it, then resets task, sensors and filter as every reset does. */
int qs_reset_to(qs_handle* h, const uint8_t* mask, const float* states);
int qs_get_obs(qs_handle* h, float* obs /*[N,obs_dim]*/);
/* The demonstration of the DEMO tasks: rows[length][action_dim + 38] (device memory; copied) in the layout
 * GetDemonstrationWrapper._get_demo records (get_demonstration_wrapper.py:35-58: filtered action, q 12, qd 12, base position 3,
 * quaternion 4, linear velocity 3, angular velocity 3, landing flag).  Replaces `np.load(self.demo_path)` (task_base.py:173);
 * qs_step fails for a DEMO task until it was called.  Row `demo counter` is compared with the action of each step; the counter
 * and its value at the start of the episode are slots 44 and 45 of QS_INFO_TASK. */
int qs_set_demo(qs_handle* h, const float* rows, int length);
/* task.set_demo_counter(value) (task_base.py:219-220) for the masked environments (mask NULL = all), as
 * ReferenceStateInitializationWrapper.reset does after set_robot_desired_state (reference_state_initialization_wrapper.py:25-33):
 * call it after qs_reset_to.  values[N] int32, device memory. */
int qs_set_demo_counter(qs_handle* h, const uint8_t* mask, const int32_t* values);
int qs_step(qs_handle* h, const float* actions /*[N,action_dim]*/, float* obs /*[N,obs_dim]*/, float* rew /*[N]*/,
            uint8_t* done /*[N]*/, uint8_t* truncated /*[N]*/);
int qs_get_state(qs_handle* h, float* state /*[N,37]*/);
int qs_set_state(qs_handle* h, const float* state /*[N,37]*/);
enum {
    QS_INFO_FOOT_FORCE = 0, QS_INFO_FOOT_CONTACT = 1, QS_INFO_TORQUE = 2, QS_INFO_SPRING_TORQUE = 3, QS_INFO_TASK = 4,
    QS_INFO_N_INVALID = 5, QS_INFO_PARAMS = 6, QS_INFO_COUNTERS = 7, QS_INFO_LAST_ACTION = 8, QS_INFO_TERMINAL_OBS = 9,
    QS_INFO_FILTERED_ACTION = 11,  /* [N,12]: output of the action filter at the last step (get_last_filtered_action, gym_env.py:385-387) */
    QS_INFO_REWARD_END = 12,  /* [N,1]: get_reward_end_episode() (gym_env.py:363-365): the end-of-episode bonus / malus the task would add
                               * if the episode ended in the current state */
    QS_INFO_PAYLOAD_BLOCK = 13,  /* [N,20], cfg.payload_soft only: the block's centre 3, quaternion 4, linear 3 and angular 3 velocity (world), the
                                  * six impulses of its fixed constraint at the last substep, the distance between the two pivots */
    QS_INFO_WRAPPER = 10,  /* [N,4]: phase after the step (0 policy, 1 take-off hold, 2 landing, 3 rest), scripted (the step just
                            * made ignored the caller's action), timer, end time */
};
int qs_info_dim(const qs_handle* h, int which);
int qs_get_info(qs_handle* h, int which, float* out /*[N, qs_info_dim]*/);
enum { QS_PARAM_MU = 0, QS_PARAM_SPRING_K = 1, QS_PARAM_SPRING_B = 2, QS_PARAM_KP = 3, QS_PARAM_KD = 4, QS_PARAM_ALL = 5 };
int qs_set_params(qs_handle* h, int which, const float* vals);
/* number of env-steps' worth of settle substeps executed so far (reset cost accounting) */
int qs_stats(qs_handle* h, uint64_t* settle_substeps, uint64_t* resets);
/* qs_step with ONE output array: fused[N][obs_dim + 2] = observation | reward | done + 2 * truncated (floats).  This is the
 * buffer a sharded run all-gathers to the learner rank (one collective per step, qs_amd/sharded.py), written by the step
 * kernel itself instead of being packed from four arrays afterwards. */
int qs_step_fused(qs_handle* h, const float* actions, float* fused);
/* ---- Host (numpy) path: the SB3 consumer of load_model.py:113-133 hands over HOST arrays.  qs_host_step_begin = VecEnv.step_async:
 * `actions` is a host array [N, action_dim] (any memory; copied into page-locked staging before the call returns); the step is enqueued on
 * the handle's stream with its action and result pointers IN that page-locked host memory, mapped into the device's address space: the
 * kernel reads its 24 B of actions per environment and writes its result rows over PCIe itself, the waves that finish first while the
 * others still compute (QS_HOST_PATH=copy in the environment: an H2D and a D2H copy around the step instead; 47 against 55 M env-steps/s
 * at N = 8192).  qs_host_step_end = VecEnv.step_wait: waits for the step and points `out` at the results in page-locked host memory
 * owned by the handle: two blocks alternate, so the arrays of a step stay valid until the end of the
 * NEXT step.  terminal_rows: the observations of the environments that ended their episode in this step BEFORE their auto-reset
 * (SB3: infos[i]["terminal_observation"]) as a compact list -- row r = [environment index (int32 bits), observation], in no particular
 * order; the number of rows is the number of set `done` flags, of which the list holds the first terminal_cap (256, or N if smaller):
 * a step that ends more episodes than that reads the rest with qs_get_info(QS_INFO_TERMINAL_OBS). */
typedef struct qs_host_result {
    const float* obs;            /* [N, obs_dim] */
    const float* rew;            /* [N] */
    const uint8_t* done;         /* [N] 0 / 1 */
    const uint8_t* truncated;    /* [N] 0 / 1: done by the time limit, not by the task (gym_env.py:245-246) */
    const float* terminal_rows;  /* [terminal_cap, 1 + obs_dim] */
    int32_t terminal_cap;
} qs_host_result;
/* Failures: one BEFORE the step's launch (bad arguments, a DEMO task without its demonstration, the previous step not collected) leaves the
 * handle as it was.  One BEHIND the launch (the attached normalisation or the copy of the block failed) is returned by qs_host_step_begin
 * and the step STAYS pending -- the simulation has advanced --: qs_host_step_end waits for it, reports the same failure once more and
 * closes the step; the host block then does not hold that step's results. */
int qs_host_step_begin(qs_handle* h, const float* actions);
int qs_host_step_end(qs_handle* h, qs_host_result* out);

/* Telemetry counters (synchronises the stream). */
enum { QS_COUNTER_SETTLE_SUBSTEPS = 0,      /* settle substeps executed (k_reset, in-step settles, settle lanes) */
       QS_COUNTER_RESETS = 1,               /* environment resets */
       QS_COUNTER_LOOKAHEAD_SERVED = 2,     /* resets that took a look-ahead state */
       QS_COUNTER_LOOKAHEAD_SETTLED = 3,    /* look-ahead states the settle lanes have delivered */
       QS_COUNTER_LIMIT_PATH_SUBSTEPS = 4,  /* wave-substeps in which some joint of the wave's 16 environments sat at a stop or (body_contacts)
                                               a non-foot link touched the plane: those run the many-rows solve (of this handle; per
                                               process before round 4) */
       QS_COUNTER_SELF_NARROW_SUBSTEPS = 5, /* wave-substeps (the last of an env step) whose self-collision broad phase found a calf close
                                               enough to another leg or the trunk to run the link-link tests (of this handle) */
       QS_COUNTER_RESET_STALLS = 6,         /* resets of a handle with reset_lookahead > 0 whose state was not ready: settled in place */
       QS_COUNTER_LOOKAHEAD_BACKLOG = 7     /* reset states the environments' look-ahead windows lack and no settle lane has taken yet */ };
int qs_counter(qs_handle* h, int which, uint64_t* value);
/* The counters 0 .. 6 as they stand at this point of the handle's stream, copied to `dev_out[8]` (DEVICE memory; entry 7 is left alone) by
 * copies enqueued on the stream: nothing is waited for.  For a reader that wants the counters of a region without idling the device in
 * front of it (bench.py: a synchronising read right before a short timed region made its first launch ~100 us late). */
int qs_counters_async(qs_handle* h, uint64_t* dev_out);
/* Average duration of the step-kernel launches of a BATCH of qs_step calls, from two HIP events on the handle's stream: one recorded in
 * front of the first step launched after qs_enable_timing(h, 1), one recorded by qs_last_step_kernel_ms, which waits for it, returns
 * elapsed milliseconds / launches since the first event and starts the next batch.  For bench.py's roofline leg: the figure covers the
 * timed region itself and includes the gaps between back-to-back launches, so it can never be shorter than the kernel's own duration nor
 * longer than the wall time per step (a pair of events around every single 0.07-ms launch read 5 % high).  qs_enable_timing(h, 2) closes
 * the batch without waiting: the closing event is recorded on the stream there and qs_last_step_kernel_ms, whenever it is called, waits for
 * that one (a caller that times the region on its own clock keeps the wait out of it). */
int qs_enable_timing(qs_handle* h, int on);
int qs_last_step_kernel_ms(qs_handle* h, float* ms);
/* The settle lanes of a handle with cfg.reset_lookahead > 0 (on from qs_create).  While on, every qs_step launch carries extra
 * workgroups that advance resets (gym_env.py:278-297, 325-327: randomizer draws, spawn, 2500 substeps under the settling command) by
 * action_repeat substeps each, through the same substep loop as the environments; a finished one goes to its environment's look-ahead
 * slot.  An epoch = the settle_steps / action_repeat launches one settle takes; the lanes work in five cohorts that start a fifth of an
 * epoch apart, each taking at its start what the environments' windows lack (environment e in episode X wants X + 1 .. X + K).  Off:
 * nothing is settled ahead (the settles in progress start over when the lanes come back), resets use up the states that are ready and
 * then settle in place -- results do not change, only when the work is done. */
int qs_settle_lanes(qs_handle* h, int on);
/* Per-substep trace tap for ONE environment (evaluation_wrapper.py:14,36-41 set_sub_step_callback; monitor_state.py:66-85):
 * while set, every qs_step writes action_repeat rows of QS_TRACE_DIM floats into `rows` (device memory, caller owned), one per
 * physics substep of environment `env`: sim time, base position 3, quaternion xyzw 4, linear velocity 3, angular velocity 3,
 * joint angles 12, joint velocities 12, applied motor torque 12, spring torque 12, foot normal force 4, foot contact 4.
 * env < 0 or rows == NULL switches it off. */
int qs_set_trace(qs_handle* h, int env, float* rows);

/* ---- SB3 VecNormalize on device (stable_baselines3 1.5.1a7: common/vec_env/vec_normalize.py, common/running_mean_std.py),
 * as applied by the reference at load_model.py:109-137 and get_demonstrations.py:71.  Operates in place on the device arrays
 * a step / reset produced; statistics are float64.  norm handles are independent of simulation handles. */
typedef struct qs_norm qs_norm;
int qs_norm_create(int n_envs, int obs_dim, double clip_obs, double clip_reward, double gamma, double epsilon, int device, qs_norm** out);
void qs_norm_destroy(qs_norm* h);
int qs_norm_set_stream(qs_norm* h, void* hip_stream);
/* host arrays [obs_dim]; RunningMeanStd.mean / .var / .count of obs_rms and ret_rms (VecNormalize.load / save) */
int qs_norm_set_stats(qs_norm* h, const double* obs_mean, const double* obs_var, double obs_count, double ret_mean, double ret_var, double ret_count);
/* what the handle was created for (any pointer may be NULL) */
int qs_norm_dims(const qs_norm* h, int* n_envs, int* obs_dim, int* device);
int qs_norm_get_stats(qs_norm* h, double* obs_mean, double* obs_var, double* obs_count, double* ret_mean, double* ret_var, double* ret_count);
/* VecNormalize.reset: returns <- 0, obs_rms.update(obs) if training && norm_obs, obs <- normalize_obs(obs) */
int qs_norm_reset(qs_norm* h, float* obs /* [N,o] */, int training, int norm_obs);
/* VecNormalize.step_wait: obs_rms.update(obs); obs <- normalize_obs(obs); returns <- returns*gamma + rew; ret_rms.update(returns);
 * rew <- normalize_reward(rew); term_obs (may be NULL) <- normalize_obs(term_obs); returns[done] <- 0.
 * raw_obs [N,o] / raw_rew [N] (either may be NULL) receive the values before normalisation: VecNormalize.old_obs / old_reward
 * (get_original_obs / get_original_reward), written by the same pass */
int qs_norm_step(qs_norm* h, float* obs /* [N,o] */, float* rew /* [N] */, const uint8_t* done /* [N] */, float* term_obs /* [N,o] or NULL */,
                 int training, int norm_obs, int norm_reward, float* raw_obs, float* raw_rew);

/* qs_norm_step with everything a host-path step carries.  The arrays of the step (device memory): obs [N,o], rew [N], done [N], trunc [N]
 * (may be NULL), term_obs [N,o] (may be NULL), tail_rows = the compact list of the step's terminal observations, [tail_cap][1 + o] (may
 * be NULL; qs_host_result::terminal_rows while still on the device).  out_* all NULL: normalised in place.  Otherwise out_obs / out_rew
 * receive the normalised arrays, out_done / out_trunc / out_tail copies of the flags and the (normalised) list -- e.g. mapped host memory,
 * which the kernel then writes itself; the inputs stay raw (term_obs is normalised in place either way).  raw_obs / raw_rew as in
 * qs_norm_step. */
typedef struct qs_norm_io {
    float* obs; float* rew; const uint8_t* done; const uint8_t* trunc; float* term_obs; float* tail_rows; int32_t tail_cap;
    float* out_obs; float* out_rew; uint8_t* out_done; uint8_t* out_trunc; float* out_tail;
    float* raw_obs; float* raw_rew;
    const uint64_t* tail_count;   /* device memory, may be NULL: how many rows of tail_rows this step filled (only those are normalised / copied;
                                   * NULL: all tail_cap rows) */
} qs_norm_io;
int qs_norm_step_io(qs_norm* h, const qs_norm_io* io, int training, int norm_obs, int norm_reward);
/* VecNormalize around the HOST path (load_model.py:109-137: VecNormalize.load(stats, env), then env.step(numpy actions)): from the next
 * qs_host_step_begin on, the step's results pass through `norm` (qs_norm_step_io: statistics update if training, observations, rewards
 * and the terminal observations of the compact list normalised) before they reach the host block, so that qs_host_step_end hands out
 * what VecNormalize.step_wait returns.  raw_obs [N,o] / raw_rew [N] (device memory, may be NULL): the values before normalisation
 * (get_original_obs / get_original_reward).  norm == NULL switches it off (at any time); it is switched on between two steps, not between
 * a begin and its end.  The handle keeps the pointer, not the object: switch it off (or destroy the simulation handle) before
 * qs_norm_destroy(norm).  Fails -- before anything is launched -- if `norm` was created for another number of environments, another
 * observation width or another device than `h`. */
int qs_host_set_norm(qs_handle* h, qs_norm* norm, int training, int norm_obs, int norm_reward, float* raw_obs, float* raw_rew);

const char* qs_last_error(void);
const char* qs_version(void);
/* Bumped whenever the meaning or type of an existing entry point's argument or of a struct field changes (a caller built against an older
 * header would pass garbage without any loader error): 5 = round 5 (qs_norm_create takes its four float arguments as double since round 4;
 * qs_config::reserved_f[0] became support_margin).  A binding compares it with the QS_ABI_VERSION of the header it was written against. */
#define QS_ABI_VERSION 5
int qs_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
