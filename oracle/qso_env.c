/* qso_env.c -- ORACLE (test infrastructure, not product code).
 * Restatement of the numpy half of the reference's hot path; every function cites the reference lines it follows
 * (paths relative to /root/reference/quadruped_spring/). Pinned by tests/golden/ (tests/golden/gen_golden.py). */
#include "qso_internal.h"
#ifdef _OPENMP
#include <omp.h>
#endif

static char g_err[256] = "";
const char* qso_last_error(void) { return g_err; }
#define FAIL(...) do { snprintf(g_err, sizeof(g_err), __VA_ARGS__); return -1; } while (0)

#define PI ((real)3.14159265358979323846)

/* ------------------------------------------------------------------ counter-based RNG (Philox4x32-10) */
void qso_philox(uint64_t seed, uint32_t env, uint32_t stream, uint32_t ctr, uint32_t blk, uint32_t out[4]) {
    uint32_t c0 = env, c1 = stream, c2 = ctr, c3 = blk;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
float qso_u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

static void normal4(uint64_t seed, uint32_t env, uint32_t stream, uint32_t ctr, uint32_t blk, real z[4]) {
    uint32_t r[4]; qso_philox(seed, env, stream, ctr, blk, r);
    for (int h = 0; h < 2; h++) {
        real u1 = qso_u01(r[2 * h]), u2 = qso_u01(r[2 * h + 1]);
        real rad = sqrt(-2 * log(u1)), th = 2 * PI * u2;
        z[2 * h] = rad * cos(th); z[2 * h + 1] = rad * sin(th);
    }
}

/* ------------------------------------------------------------------ action -> motor command */
static real clampr(real x, real lo, real hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* env/control_interface/action_interface.py:14-15 (Default), :29-39 (Symmetric), :58-65 (NoHip) */
static void expand_action(const qso_config* cfg, const real* a, real* a12) {
    int si = cfg->symm_idx;
    if (cfg->action_space_mode == QSO_ACT_DEFAULT) {
        for (int i = 0; i < 12; i++) a12[i] = a[i];
    } else if (cfg->action_space_mode == QSO_ACT_SYMMETRIC) {
        for (int i = 0; i < 3; i++) { a12[i] = a[i]; a12[3 + i] = a[i]; a12[6 + i] = a[3 + i]; a12[9 + i] = a[3 + i]; }
        a12[3 + si] = -a[si]; a12[9 + si] = -a[3 + si];
    } else { /* np.insert(leg, symm_idx, 0): FL = FR, RL = RR (no mirroring, :61-62) */
        for (int leg = 0; leg < 2; leg++) {
            real t[3]; int k = 0;
            for (int i = 0; i < 3; i++) t[i] = (i == si) ? 0 : a[2 * leg + k++];
            for (int i = 0; i < 3; i++) { a12[6 * leg + i] = t[i]; a12[6 * leg + 3 + i] = t[i]; }
        }
    }
}

/* env/control_interface/interface_base.py:84-90 + motor_interface.py:34-36 (PD), :70-80 (CARTESIAN_PD) */
void qso_action_to_command(const qso_config* cfg, const real* action, real* cmd12) {
    real a12[12]; expand_action(cfg, action, a12);
    real s[12];
    for (int i = 0; i < 12; i++) {
        real a = clampr(a12[i], -1, 1), lo = cfg->cmd_lo[i], hi = cfg->cmd_hi[i];
        s[i] = clampr(lo + (real)0.5 * (a + 1) * (hi - lo), lo, hi);
    }
    if (cfg->motor_control_mode == QSO_MOTOR_CARTESIAN_PD) {
        for (int L = 0; L < 4; L++) qso_leg_ik(cfg->leg_len, L, s + 3 * L, cmd12 + 3 * L);
    } else {
        for (int i = 0; i < 12; i++) cmd12[i] = s[i];
    }
}

/* interface_base.py:92-100 then action_interface.py:41-44 / :67-74 (inverse of the scaling only; no FK for cartesian) */
void qso_command_to_action(const qso_config* cfg, const real* cmd12, real* action) {
    real a12[12];
    for (int i = 0; i < 12; i++) {
        real lo = cfg->cmd_lo[i], hi = cfg->cmd_hi[i];
        real c = clampr(cmd12[i], lo, hi);
        a12[i] = clampr(-1 + 2 * (c - lo) / (hi - lo), -1, 1);
    }
    if (cfg->action_space_mode == QSO_ACT_DEFAULT) {
        for (int i = 0; i < 12; i++) action[i] = a12[i];
    } else if (cfg->action_space_mode == QSO_ACT_SYMMETRIC) {
        for (int i = 0; i < 3; i++) { action[i] = a12[i]; action[3 + i] = a12[6 + i]; }
    } else {
        int k = 0;
        for (int i = 0; i < 3; i++) if (i != cfg->symm_idx) action[k++] = a12[i];
        for (int i = 0; i < 3; i++) if (i != cfg->symm_idx) action[k++] = a12[6 + i];
    }
}

/* env/quadruped_motor.py:45-99 */
void qso_pd_torque(const qso_config* cfg, const real* kp3, const real* kd3, const real* cmd12, const real* q, const real* qd, real* tau) {
    for (int i = 0; i < 12; i++) {
        real lim = cfg->tau_max[i % 3], t;
        if (cfg->motor_control_mode == QSO_MOTOR_TORQUE) t = cmd12[i];
        else t = -1 * (kp3[i % 3] * (q[i] - cmd12[i])) - kd3[i % 3] * (qd[i] - 0);
        tau[i] = clampr(t, -lim, lim);
    }
}

/* env/quadruped_motor.py:101-104 + env/springs.py:34-74 (legs 0,2 are "right", 1,3 "left") */
void qso_spring_torque(const real* k3, const real* b3, const real* rest3, const real* q, const real* qd, real* tau) {
    for (int L = 0; L < 4; L++) {
        int right = (L % 2) == 0;
        for (int j = 0; j < 3; j++) {
            real a = q[3 * L + j], k = k3[j], b = b3[j];
            int off;
            if (j == 0) off = right ? (a > rest3[0]) : (a < rest3[0]);
            else if (j == 1) off = a < rest3[1];
            else off = a > rest3[2];
            if (off) { k = 0; b = 0; }
            tau[3 * L + j] = -k * (a - rest3[j]) - b * qd[3 * L + j];
        }
    }
}

/* env/quadruped.py:348-392 */
void qso_leg_fk_jac(const float* len, int leg, const real* q, real* J, real* pos) {
    real l1 = len[0], l2 = len[1], l3 = len[2];
    real side = (leg == 0 || leg == 2) ? -1 : 1;
    real s1 = sin(q[0]), s2 = sin(q[1]), s3 = sin(q[2]), c1 = cos(q[0]), c2 = cos(q[1]), c3 = cos(q[2]);
    real c23 = c2 * c3 - s2 * s3, s23 = s2 * c3 + c2 * s3;
    for (int i = 0; i < 9; i++) J[i] = 0;
    J[3] = -side * l1 * s1 + l2 * c2 * c1 + l3 * c23 * c1;
    J[6] = side * l1 * c1 + l2 * c2 * s1 + l3 * c23 * s1;
    J[1] = -l3 * c23 - l2 * c2;
    J[4] = -l2 * s2 * s1 - l3 * s23 * s1;
    J[7] = l2 * s2 * c1 + l3 * s23 * c1;
    J[2] = -l3 * c23;
    J[5] = -l3 * s23 * s1;
    J[8] = l3 * s23 * c1;
    pos[0] = -l3 * s23 - l2 * s2;
    pos[1] = l1 * side * c1 + l3 * (s1 * c23) + l2 * c2 * s1;
    pos[2] = l1 * side * s1 - l3 * (c1 * c23) - l2 * c1 * c2;
}

/* env/quadruped.py:399-438 */
void qso_leg_ik(const float* len, int leg, const real* xyz, real* q) {
    real sh = len[0], el = len[1], wr = len[2];
    real x = xyz[0], y = xyz[1], z = xyz[2];
    real D = (y * y + z * z - sh * sh + x * x - el * el - wr * wr) / (2 * wr * el);
    D = clampr(D, -1, 1);
    real side = (leg == 0 || leg == 2) ? -1 : 1;
    real wrist = atan2(-sqrt(1 - D * D), D);
    real sc = y * y + z * z - sh * sh;
    if (sc < 0) sc = 0;
    real shoulder = -atan2(z, y) - atan2(sqrt(sc), side * sh);
    real elbow = atan2(-x, sqrt(sc)) - atan2(wr * sin(wrist), el + wr * cos(wrist));
    q[0] = -shoulder; q[1] = elbow; q[2] = wrist;
}

/* utils/action_filter.py:110-121; xhist/yhist[0..d) newest, [d..2d) older */
void qso_filter_step(const double* b, const double* a, int d, const real* x, real* xh, real* yh, real* y) {
    for (int i = 0; i < d; i++) {
        real v = x[i] * (real)b[0] + (xh[i] * (real)b[1] + xh[d + i] * (real)b[2]) - (yh[i] * (real)a[1] + yh[d + i] * (real)a[2]);
        xh[d + i] = xh[i]; xh[i] = x[i];
        yh[d + i] = yh[i]; yh[i] = v;
        y[i] = v;
    }
}

/* pybullet getEulerFromQuaternion (btQuaternion::getEulerZYX), used at env/quadruped.py:131-139 */
void qso_quat_to_rpy(const real* q, real* rpy) {
    real x = q[0], y = q[1], z = q[2], w = q[3];
    real sqx = x * x, sqy = y * y, sqz = z * z, sqw = w * w;
    real sarg = -2 * (x * z - w * y);
    if (sarg <= (real)-0.99999) { rpy[1] = -(real)0.5 * PI; rpy[0] = 0; rpy[2] = 2 * atan2(x, -y); }
    else if (sarg >= (real)0.99999) { rpy[1] = (real)0.5 * PI; rpy[0] = 0; rpy[2] = 2 * atan2(-x, y); }
    else {
        rpy[1] = asin(sarg);
        rpy[0] = atan2(2 * (y * z + w * x), sqw - sqx - sqy + sqz);
        rpy[2] = atan2(2 * (x * y + w * z), sqw + sqx - sqy - sqz);
    }
}

/* env/sensors/robot_sensors.py:333-340: -Rotation.from_quat(q).as_euler("yxz")[0], +2pi after take-off */
real qso_pitch_backflip(const real* q, int switched) {
    real R[3][3]; qso_quat_to_mat(q, R);
    real pitch = -atan2(-R[2][0], R[2][2]);
    if (pitch < 0 && switched) pitch = 2 * PI + pitch;
    return pitch;
}

/* ------------------------------------------------------------------ Hopf CPG (hopf_network.py:117-173) */
void qso_cpg_update(const qso_config* cfg, const real* p, real dt, real* X, real* x, real* z) {
    real w_swing = p[0], w_stance = p[1], mu = p[2], d_step = p[3], h = p[4];
    real rd[4], td[4];
    for (int i = 0; i < 4; i++) {
        real r = X[i], th = X[4 + i];
        rd[i] = (real)cfg->cpg_alpha * (mu - r * r) * r;                                    /* :149 */
        td[i] = sin(th) > 0 ? w_swing : w_stance;                                           /* :152-156 */
        for (int j = 0; j < 4; j++)                                                         /* :159-162 */
            if (j != i) td[i] += X[j] * (real)cfg->cpg_coupling * sin(X[4 + j] - th - (real)cfg->cpg_phi[4 * i + j]);
    }
    for (int i = 0; i < 4; i++) {
        X[i] += dt * rd[i];
        real th = X[4 + i] + dt * td[i];
        X[4 + i] = th - 2 * PI * floor(th / (2 * PI));                                      /* :170 python % */
    }
    for (int i = 0; i < 4; i++) {                                                           /* :124-133 */
        real r = X[i], th = X[4 + i], s = sin(th);
        x[i] = -d_step * r * cos(th);
        z[i] = s > 0 ? -h + (real)cfg->cpg_clearance * s : -h + (real)cfg->cpg_penetration * s;
    }
}

/* CPG action layer (BASELINE.json configs[4]): action in [-1,1]^5 -> CPG parameters; feet at (x, side * hip_len, z) -> IK */
static void cpg_params(const qso_config* cfg, const real* act, real* p) {
    for (int i = 0; i < 5; i++) {
        real a = clampr(act[i], -1, 1);
        p[i] = (real)cfg->cpg_lo[i] + (real)0.5 * (a + 1) * ((real)cfg->cpg_hi[i] - (real)cfg->cpg_lo[i]);
    }
}
static void cpg_command(const qso_config* cfg, qso_env* e, const real* p, real* cmd) {
    real x[4], z[4];
    qso_cpg_update(cfg, p, (real)cfg->dt, e->cpg, x, z);
    for (int L = 0; L < 4; L++) {
        real xyz[3] = {x[L], ((L & 1) ? 1 : -1) * (real)cfg->leg_len[0], z[L]};
        qso_leg_ik(cfg->leg_len, L, xyz, cmd + 3 * L);
    }
}

/* ------------------------------------------------------------------ task state machine */
static real sim_time(const qso_config* cfg, const qso_env* e) { return (real)e->sim_step * (real)cfg->dt; }
static int is_flying(const qso_env* e) { /* quadruped.py:260-262 */
    return !(e->foot_contact[0] || e->foot_contact[1] || e->foot_contact[2] || e->foot_contact[3]);
}
static real jump_distance(const qso_task* t) { /* task_base.py:108-116: (pos - pose_to) @ Rz(-yaw) */
    real dx = t->pos[0] - t->pose_to[0], dy = t->pos[1] - t->pose_to[1];
    real d = cos(t->yaw_to) * dx - sin(t->yaw_to) * dy;
    return d > 0 ? d : 0;
}
static void max_fwd_update(qso_task* t) { real d = jump_distance(t); if (d > t->max_fwd) t->max_fwd = d; }
static int task_family_continuous(int task) { return task == QSO_TASK_CONT_JUMPING_FORWARD || task == QSO_TASK_CONT_JUMPING_FORWARD2; }
static int task_family_continuous2(int task) { return task == QSO_TASK_CONT_JUMPING_FORWARD3 || task == QSO_TASK_CONT_JUMPING_FORWARD_PPO || task == QSO_TASK_CONT_JUMPING_FORWARD_DEMO; }
static int task_family_demo(int task) { return task >= QSO_TASK_JUMPING_IN_PLACE_DEMO && task <= QSO_TASK_CONT_JUMPING_FORWARD_DEMO; }
/* TaskContinuousJumping2 constants (task_base.py:286-290; robot_tasks.py:171-175, 559-561) */
static void cj2_constants(int task, real* jump_limit, real* height_limit, real* bound) {
    if (task == QSO_TASK_CONT_JUMPING_FORWARD3) { *jump_limit = (real)0.6; *height_limit = (real)0.45; *bound = (real)0.7; }
    else if (task == QSO_TASK_CONT_JUMPING_FORWARD_DEMO) { *jump_limit = (real)0.5; *height_limit = (real)0.5; *bound = (real)0.85; } /* the base class's */
    else { *jump_limit = (real)0.6; *height_limit = (real)0.5; *bound = (real)0.85; }
}
/* get_entropy_fwd (task_base.py:376-383) from the sums: -sum p log2 p = log2 S - (sum f log2 f) / S, over max(n, 3) entries */
static real cj2_entropy(const qso_task* t) {
    if (t->jump_count == 0 || t->sum_fwd < (real)0.05) return 0;
    real n = t->jump_count < 3 ? 3 : t->jump_count;
    return (log2(t->sum_fwd) - t->sum_flogf / t->sum_fwd) / log2(n);
}

static void task_on_step(const qso_config* cfg, qso_env* e) {
    qso_task* t = &e->task;
    if (cfg->task == QSO_TASK_NO_TASK) return;
    int flying = is_flying(e);
    real vz = e->s.vlin[2];
    /* task_base.py:152-160 (note g = 9.81 here) */
    if (!t->switched && flying && vz / (real)9.81 > (real)0.06) t->switched = 1;
    /* :68-70 */
    memcpy(t->old_tau, t->new_tau, sizeof(t->old_tau));
    memcpy(t->new_tau, e->tau_pd, sizeof(t->new_tau));
    /* :72-75 */
    memcpy(t->pos, e->s.pos, sizeof(t->pos)); memcpy(t->vel, e->s.vlin, sizeof(t->vel));
    qso_quat_to_rpy(e->s.quat, t->rpy);
    /* :81-90 */
    real z = t->pos[2], dh = z - t->init_h; if (dh < 0) dh = 0;
    if (dh > t->rel_max_h) t->rel_max_h = dh;
    if (fabs(z) > t->max_h) t->max_h = fabs(z);
    if (fabs(t->pos[0]) > t->max_dx) t->max_dx = fabs(t->pos[0]);
    if (fabs(t->rpy[1]) > t->max_pitch) t->max_pitch = fabs(t->rpy[1]);
    real now = sim_time(cfg, e);
    if (task_family_continuous2(cfg->task)) { /* task_base.py:321-355 */
        real jump_limit, height_limit, bound; cj2_constants(cfg->task, &jump_limit, &height_limit, &bound);
        t->end_jump = 0;
        if (flying) {
            if (!t->all_air) {
                t->all_air = 1; t->t_takeoff = now; memcpy(t->pose_to, t->pos, sizeof(t->pos)); t->yaw_to = t->rpy[2];
                t->is_jumping = flying && vz / (real)9.81 > (real)0.06;
                t->max_jump_h = 0; /* restart_jump_performance_variables() right after max_jump_height = z (:327-330) */
            } else if (t->pos[2] > t->max_jump_h) t->max_jump_h = t->pos[2];
        } else if (t->all_air) {
            if (now - t->t_takeoff > t->max_flight) t->max_flight = now - t->t_takeoff;
            if (t->first_jump == 0) { /* :342-353; the first jump is ignored */
                real d = jump_distance(t);
                real fwd = d < jump_limit ? d : jump_limit, hgt = t->max_jump_h < height_limit ? t->max_jump_h : height_limit;
                real perf = (real)0.7 * fwd / jump_limit + (real)0.3 * hgt / height_limit;
                t->jump_count += 1; t->sum_fwd += fwd; t->sum_flogf += fwd > 0 ? fwd * log2(fwd) : 0; t->sum_height += hgt;
                t->sum_perf += perf; if (perf > t->max_perf) t->max_perf = perf; t->last_perf = perf;
                if (perf >= bound) t->good_jumps += 1;
                t->end_jump = 1;
            } else t->first_jump = 0;
            t->all_air = 0; t->is_jumping = 0;
        }
    } else if (!task_family_continuous(cfg->task)) { /* :92-106 */
        if (flying) {
            if (!t->all_air) { t->all_air = 1; t->t_takeoff = now; memcpy(t->pose_to, t->pos, sizeof(t->pos)); t->yaw_to = t->rpy[2]; }
            else max_fwd_update(t);
        } else {
            if (t->all_air) { if (now - t->t_takeoff > t->max_flight) t->max_flight = now - t->t_takeoff; max_fwd_update(t); t->all_air = 0; }
            else t->max_fwd = 0;
        }
    } else { /* task_base.py:244-280 */
        real jump_limit = 0.5, time_limit = cfg->task == QSO_TASK_CONT_JUMPING_FORWARD ? (real)0.15 : (real)0.35;
        if (flying) {
            if (!t->all_air) {
                t->all_air = 1; t->t_takeoff = now; memcpy(t->pose_to, t->pos, sizeof(t->pos)); t->yaw_to = t->rpy[2];
                t->is_jumping = flying && vz / (real)9.81 > (real)0.06;
            }
        } else if (t->all_air) {
            if (now - t->t_takeoff > t->max_flight) t->max_flight = now - t->t_takeoff;
            max_fwd_update(t);
            t->cum_fwd += t->max_fwd < jump_limit ? t->max_fwd : jump_limit;
            t->cum_ft += t->max_flight < time_limit ? t->max_flight : time_limit;
            t->all_air = 0; t->is_jumping = 0;
        }
    }
    if (cfg->task == QSO_TASK_JUMPING_FORWARD_PPO || cfg->task == QSO_TASK_JUMPING_FORWARD_PPO_HP) { /* robot_tasks.py:418-425 */
        t->old_fwd = t->actual_fwd; t->actual_fwd = t->max_fwd;
    }
    if (cfg->task == QSO_TASK_BACKFLIP) { /* robot_tasks.py:527-530 */
        real p = qso_pitch_backflip(e->s.quat, t->switched);
        if (p > t->bf_max_pitch) t->bf_max_pitch = p;
    }
    if (cfg->task == QSO_TASK_BACKFLIP_PPO) { /* robot_tasks.py:752-754: reuses TaskJumping._max_pitch */
        real p = qso_pitch_backflip(e->s.quat, t->switched);
        if (p > t->max_pitch) t->max_pitch = p;
    }
}

static void task_reset(const qso_config* cfg, qso_env* e) { /* task_base.py:40-59 */
    qso_task* t = &e->task;
    real keep_bf = t->bf_max_pitch; /* BackFlip.max_pitch is only initialised in __init__ (robot_tasks.py:524) */
    memset(t, 0, sizeof(*t));
    t->bf_max_pitch = keep_bf;
    t->first_jump = 1;
    if (cfg->task == QSO_TASK_NO_TASK) return;
    t->t_takeoff = sim_time(cfg, e);
    memcpy(t->pose_to, e->s.pos, sizeof(t->pose_to));
    t->init_h = e->s.pos[2];
    real rpy[3]; qso_quat_to_rpy(e->s.quat, rpy); t->yaw_to = rpy[2];
    memcpy(t->old_tau, e->tau_pd, sizeof(t->old_tau)); memcpy(t->new_tau, e->tau_pd, sizeof(t->new_tau));
    task_on_step(cfg, e);
}

static int task_terminated(const qso_config* cfg, const qso_env* e) {
    const qso_task* t = &e->task;
    if (cfg->task == QSO_TASK_NO_TASK) return 0;
    int low = t->pos[2] < cfg->fallen_height;
    /* the demonstration is used up: task_base.py:213-214, 446-447; robot_tasks.py:239-241 */
    int demo_end = task_family_demo(cfg->task) && t->demo_counter >= e->demo_len;
    if (cfg->task == QSO_TASK_BACKFLIP || cfg->task == QSO_TASK_BACKFLIP_DEMO) return low || e->n_invalid > 0 || demo_end; /* robot_tasks.py:532-533 */
    real R[3][3]; qso_quat_to_mat(e->s.quat, R);
    int tilted = R[2][2] < (real)0.85; /* task_base.py:126-130 */
    return (tilted && low) || e->n_invalid > 0 || demo_end; /* :132-147 */
}
/* TaskJumpingDemo._reward (task_base.py:194-211): distance between the demonstration's action (the FILTERED action the
   demonstration recorded) and the action env.step was just given (get_last_action: the unfiltered one), shared out over
   the rows that were left when the episode began; consumes one row */
static real demo_reward(const qso_handle* h, qso_env* e) {
    qso_task* t = &e->task;
    const int d = h->cfg.action_dim, row = t->demo_counter < h->demo_len ? t->demo_counter : h->demo_len - 1;
    const float* a = h->demo + (size_t)row * (d + 38);
    real n2 = 0;
    for (int k = 0; k < d; k++) { real x = (real)a[k] - e->last_action[k]; n2 += x * x; }
    t->demo_counter++;
    return exp(-(real)0.35 * sqrt(n2)) / (real)(h->demo_len - t->demo_start);
}

static real clipped_height(real z, real lo, real hi) { return (z < lo || z > hi) ? 0 : z; }

static real task_reward(const qso_config* cfg, const qso_env* e) {
    const qso_task* t = &e->task;
    int ppo_ip = cfg->task == QSO_TASK_JUMPING_IN_PLACE_PPO || cfg->task == QSO_TASK_JUMPING_IN_PLACE_PPO_HP;
    int ppo_fw = cfg->task == QSO_TASK_JUMPING_FORWARD_PPO || cfg->task == QSO_TASK_JUMPING_FORWARD_PPO_HP;
    if (cfg->task == QSO_TASK_BACKFLIP_PPO) { /* robot_tasks.py:709-800 */
        real rew_h = (real)0.026 * clipped_height(t->pos[2], (real)0.29, (real)0.7);
        real nd = 0; for (int i = 0; i < 12; i++) { real d = t->old_tau[i] - t->new_tau[i]; nd += d * d; }
        real rew_smooth = (real)0.015 * exp(-(real)0.1 * sqrt(nd));
        real cf = e->foot_force[0] + e->foot_force[1] + e->foot_force[2] + e->foot_force[3];
        real rew_contact = -(real)3e-4 * (cf > 800 ? cf : 0);
        real rew_pitch = (real)0.014 * (t->pos[2] > (real)0.5 ? qso_pitch_backflip(e->s.quat, t->switched) : 0);
        return (real)0.4 * rew_contact + (real)0.2 * rew_smooth + (real)0.25 * rew_h + (real)0.3 * rew_pitch;
    }
    /* ContinuousJumpingForwardPPO._reward tests a bound method (`if not self.is_switched_controller:`, robot_tasks.py:669),
       which is always truthy: its step reward is the constant 0 */
    if (!ppo_ip && !ppo_fw) return 0;
    /* robot_tasks.py:258-344 / :369-472 */
    real max_h = ppo_ip ? (cfg->task == QSO_TASK_JUMPING_IN_PLACE_PPO ? (real)1.0 : (real)1.25)
                        : (cfg->task == QSO_TASK_JUMPING_FORWARD_PPO ? (real)0.9 : (real)1.1);
    real k_h = ppo_ip ? (real)0.023 : (real)0.026;
    real rew_h = k_h * clipped_height(t->pos[2], (real)0.29, max_h);
    real nd = 0; for (int i = 0; i < 12; i++) { real d = t->old_tau[i] - t->new_tau[i]; nd += d * d; }
    real rew_smooth = (real)0.015 * exp(-(real)0.1 * sqrt(nd));
    real cf = e->foot_force[0] + e->foot_force[1] + e->foot_force[2] + e->foot_force[3];
    real rew_contact = -(real)3e-4 * (cf > 800 ? cf : 0);
    real rew_pitch = (real)0.014 * exp(-26 * fabs(t->rpy[1]));
    if (ppo_ip) {
        real rew_pos = (real)0.013 * exp(-40 * fabs(t->pos[0]));
        return (real)0.05 * rew_pos + (real)0.5 * rew_contact + (real)0.2 * rew_smooth + (real)0.45 * rew_h + (real)0.3 * rew_pitch;
    }
    real max_fwd = cfg->task == QSO_TASK_JUMPING_FORWARD_PPO ? (real)1.3 : (real)1.4;
    real fwd = t->actual_fwd;
    if (fwd > max_fwd || fwd == t->old_fwd) fwd = 0;
    real rew_fwd = (real)0.038 * fwd;
    return (real)0.4 * rew_contact + (real)0.2 * rew_smooth + (real)0.25 * rew_h + (real)0.3 * rew_pitch + (real)0.4 * rew_fwd;
}

static real task_reward_end(const qso_config* cfg, const qso_env* e) {
    const qso_task* t = &e->task;
    int term = task_terminated(cfg, e);
    real r = 0;
    switch (cfg->task) {
    case QSO_TASK_JUMPING_IN_PLACE: { /* robot_tasks.py:31-57 */
        real hn = t->rel_max_h > (real)0.9 ? 1 : t->rel_max_h / (real)0.9;
        r += (real)0.7 * hn;
        r += hn * (real)0.3 * exp(-t->max_pitch * t->max_pitch / ((real)0.15 * (real)0.15));
        r += hn * (real)0.05 * exp(-t->max_dx * t->max_dx / (real)0.05);
        if (!term) r += (real)0.1 * hn; else r -= (real)0.08 * (1 + (real)0.8 * hn);
        break; }
    case QSO_TASK_JUMPING_FORWARD: { /* :70-99 */
        real hn = t->rel_max_h > (real)0.3 ? 1 : t->rel_max_h / (real)0.3;
        real fn = t->max_fwd > (real)1.3 ? 1 : t->max_fwd / (real)1.3;
        real bm = (hn + fn) / 2;
        r += (real)0.25 * hn; r += (real)0.5 * fn * hn;
        r += hn * (real)0.25 * exp(-t->max_pitch * t->max_pitch / ((real)0.15 * (real)0.15));
        if (!term) r += (real)0.1 * bm; else r -= (real)0.08 * (1 + (real)1.2 * bm);
        break; }
    case QSO_TASK_CONT_JUMPING_FORWARD: { /* :112-131 */
        real tn = t->cum_ft / (real)0.15, dn = t->cum_fwd / (real)0.5, bm = (tn + dn) / 2;
        r += (real)0.25 * tn; r += (real)0.5 * dn;
        r += tn * (real)0.25 * exp(-t->max_pitch * t->max_pitch / ((real)0.15 * (real)0.15));
        if (!term) r += (real)0.1 * bm;
        break; }
    case QSO_TASK_CONT_JUMPING_FORWARD2: { /* :144-165 */
        real tl = 0.35, jl = 0.5;
        real tn = (t->max_flight < tl ? t->max_flight : tl) / tl, dn = (t->max_fwd < jl ? t->max_fwd : jl) / jl, bm = (tn + dn) / 2;
        r += (real)0.25 * tn; r += (real)0.5 * dn;
        r += dn * (real)0.15 * exp(-(t->max_pitch * t->max_pitch / ((real)0.15 * (real)0.15)));
        r += (real)0.4 * (sim_time(cfg, e) / 10) * bm;
        if (!term) r += (real)0.2 * bm;
        break; }
    case QSO_TASK_JUMPING_IN_PLACE_PPO: case QSO_TASK_JUMPING_IN_PLACE_PPO_HP: /* :348-358 */
        if (term) r -= (real)0.25 * t->max_h;
        break;
    case QSO_TASK_JUMPING_FORWARD_PPO: case QSO_TASK_JUMPING_FORWARD_PPO_HP: /* :475-485 */
        if (!term) r += (real)0.05 * (t->max_fwd + t->max_h) / 2;
        break;
    case QSO_TASK_BACKFLIP: { /* :535-550 */
        real h = t->max_h - (real)0.3; if (h < 0) h = 0; if (h > (real)0.4) h = (real)0.4; h /= (real)0.4;
        real pm = t->bf_max_pitch / (2 * PI);
        r += pm * (real)0.4; r += h * (real)0.4; r += h * pm;
        if (t->switched && !term) r += (real)0.2;
        break; }
    case QSO_TASK_BACKFLIP_PPO: /* robot_tasks.py:802-809 */
        if (!term) r += (real)0.2 * ((real)0.7 * t->max_pitch / 5 + (real)0.3 * t->max_h) / 2;
        break;
    case QSO_TASK_CONT_JUMPING_FORWARD3: { /* robot_tasks.py:181-212 */
        real n = t->jump_count < 3 ? 3 : t->jump_count;
        real avg = t->sum_perf / n, mx = t->max_perf > 0 ? t->max_perf : 0;
        real rew_entropy = exp((cj2_entropy(t) - 1) / (real)0.3);
        real ra = avg * (real)0.15 * exp(-t->max_pitch * t->max_pitch / ((real)0.15 * (real)0.15));
        ra += avg * (real)0.4 * (sim_time(cfg, e) / 10);
        ra += avg * rew_entropy * (real)0.2;
        ra += avg * (real)0.25;
        r = (real)0.8 * ra + (real)0.2 * mx + (real)0.1 * t->good_jumps;
        if (!term) r += (real)0.2 * avg;
        break; }
    case QSO_TASK_CONT_JUMPING_FORWARD_PPO: { /* robot_tasks.py:686-698 */
        real n = t->jump_count < 3 ? 3 : t->jump_count;
        r = (t->sum_perf / n) * exp((cj2_entropy(t) - 1) / (real)0.3);
        if (term) r -= 1;
        break; }
    default: break;
    }
    return r;
}

/* ------------------------------------------------------------------ sensors (env/sensors/robot_sensors.py, sensor.py:46-60) */
static int sensor_dim(int s) {
    switch (s) {
    case QSO_SENS_JOINT_POS: case QSO_SENS_JOINT_VEL: case QSO_SENS_FEET_POS: case QSO_SENS_FEET_VEL: return 12;
    case QSO_SENS_BOOL_CONTACT: case QSO_SENS_QUAT: return 4;
    case QSO_SENS_LIN_VEL: case QSO_SENS_ANG_VEL: case QSO_SENS_RPY: return 3;
    default: return 1;
    }
}

static void read_sensors(const qso_config* cfg, qso_env* e, float* obs) {
    real o[QSO_MAX_OBS]; int n = 0;
    real rpy[3]; qso_quat_to_rpy(e->s.quat, rpy);
    for (int si = 0; si < cfg->n_sensors; si++) {
        switch (cfg->sensors[si]) {
        case QSO_SENS_JOINT_POS: for (int i = 0; i < 12; i++) o[n++] = e->s.q[i]; break;
        case QSO_SENS_JOINT_VEL: for (int i = 0; i < 12; i++) o[n++] = e->s.qd[i]; break;
        case QSO_SENS_PITCH: o[n++] = rpy[1]; break;
        case QSO_SENS_HEIGHT: o[n++] = e->s.pos[2]; break;
        case QSO_SENS_VEL_Z: o[n++] = e->s.vlin[2]; break;
        case QSO_SENS_VEL_X: o[n++] = e->s.vlin[0]; break;
        case QSO_SENS_LANDING: o[n++] = e->task.switched; break;
        case QSO_SENS_JUMPING: o[n++] = e->task.is_jumping; break;
        case QSO_SENS_PITCH_RATE: { /* quadruped.py:141-170: w_body = R^T w_world */
            real R[3][3], wb[3]; qso_quat_to_mat(e->s.quat, R); m3tv(R, e->s.vang, wb); o[n++] = wb[1]; break; }
        case QSO_SENS_BOOL_CONTACT: for (int i = 0; i < 4; i++) o[n++] = e->foot_contact[i]; break;
        case QSO_SENS_LIN_VEL: for (int i = 0; i < 3; i++) o[n++] = e->s.vlin[i]; break;
        case QSO_SENS_ANG_VEL: for (int i = 0; i < 3; i++) o[n++] = e->s.vang[i]; break;
        case QSO_SENS_RPY: for (int i = 0; i < 3; i++) o[n++] = rpy[i]; break;
        case QSO_SENS_QUAT: for (int i = 0; i < 4; i++) o[n++] = e->s.quat[i]; break;
        case QSO_SENS_FEET_POS: case QSO_SENS_FEET_VEL: /* quadruped.py:440-449 */
            for (int L = 0; L < 4; L++) {
                real J[9], p[3]; qso_leg_fk_jac(cfg->leg_len, L, e->s.q + 3 * L, J, p);
                for (int i = 0; i < 3; i++) {
                    if (cfg->sensors[si] == QSO_SENS_FEET_POS) o[n++] = p[i];
                    else o[n++] = J[3 * i] * e->s.qd[3 * L] + J[3 * i + 1] * e->s.qd[3 * L + 1] + J[3 * i + 2] * e->s.qd[3 * L + 2];
                }
            }
            break;
        case QSO_SENS_PITCH_BACKFLIP: o[n++] = qso_pitch_backflip(e->s.quat, e->task.switched); break;
        default: break;
        }
    }
    for (int i = 0; i < n; i++) obs[i] = (float)o[i];
}

/* i.i.d. Gaussian noise resampled at every read (sensor.py:25-32, 46-60); the reference draws from the global
 * unseeded np.random, so only the distribution is reproducible: here Philox(seed; env, stream 0, total_steps, block). */
static void add_noise(const qso_config* cfg, qso_env* e, int env_id, float* obs) {
    if (!cfg->noise_enabled) return;
    for (int blk = 0; blk * 4 < cfg->obs_dim; blk++) {
        real z[4]; normal4(cfg->seed, (uint32_t)(env_id + cfg->env_id_offset), 0, e->total_steps, (uint32_t)blk, z);
        for (int k = 0; k < 4 && blk * 4 + k < cfg->obs_dim; k++) {
            float sd = cfg->obs_noise_std[blk * 4 + k];
            if (sd > 0) obs[blk * 4 + k] = (float)((real)obs[blk * 4 + k] + (real)sd * z[k]);
        }
    }
}

/* ------------------------------------------------------------------ reset / step */
static void apply_and_step_mode(const qso_config* cfg, qso_env* e, const real* cmd, real g, int settling) {
    /* quadruped.py:288-320 then gym_env.py:218-219.  settling: control_interface/utils.py:7-31 switches the motor model to "PD"
       for the settle of a reset, also when the environment itself is driven by raw torques */
    real tau[12];
    if (settling && cfg->motor_control_mode == QSO_MOTOR_TORQUE) {
        qso_config pd = *cfg; pd.motor_control_mode = QSO_MOTOR_PD;
        qso_pd_torque(&pd, e->kp, e->kd, cmd, e->s.q, e->s.qd, e->tau_pd);
    } else qso_pd_torque(cfg, e->kp, e->kd, cmd, e->s.q, e->s.qd, e->tau_pd);
    if (cfg->enable_springs) qso_spring_torque(e->k, e->b, e->rest, e->s.q, e->s.qd, e->tau_spring);
    else memset(e->tau_spring, 0, sizeof(e->tau_spring));
    for (int i = 0; i < 12; i++) tau[i] = e->tau_pd[i] + e->tau_spring[i];
    qso_physics_substep(cfg, e, tau, g);
}
static void apply_and_step(const qso_config* cfg, qso_env* e, const real* cmd, real g) { apply_and_step_mode(cfg, e, cmd, g, 0); }

static void randomize(const qso_config* cfg, qso_env* e, int env_id) {
    if ((cfg->randomizer_flags & QSO_RAND_KEEP) && e->episode >= 0) return;
    /* nominal */
    e->mu = 1;
    for (int i = 0; i < 3; i++) { e->k[i] = cfg->spring_k[i]; e->b[i] = cfg->spring_b[i]; e->rest[i] = cfg->spring_rest[i]; e->kp[i] = cfg->kp[i]; e->kd[i] = cfg->kd[i]; e->m_leg[i] = QSO_M_LEG[i]; e->r_pay[i] = 0; }
    e->m_trunk = QSO_M_TRUNK; e->m_pay = 0;
    uint32_t r[16];
    for (int b = 0; b < 4; b++) qso_philox(cfg->seed, (uint32_t)(env_id + cfg->env_id_offset), 1, (uint32_t)e->episode, (uint32_t)b, r + 4 * b);
    if (cfg->randomizer_flags & QSO_RAND_GROUND) /* env_randomizer.py:287-289 */
        e->mu = (real)0.5 + (real)0.5 * qso_u01(r[0]);
    if ((cfg->randomizer_flags & QSO_RAND_SPRINGS) && cfg->enable_springs) { /* :100-122 */
        for (int i = 0; i < 3; i++) {
            real lo = (real)cfg->spring_k[i] * (real)0.9, hi = (real)cfg->spring_k[i] * (real)1.1;
            e->k[i] = lo + (hi - lo) * qso_u01(r[1 + i]);
            lo = (real)cfg->spring_b[i] * (real)0.9; hi = (real)cfg->spring_b[i] * (real)1.1;
            e->b[i] = lo + (hi - lo) * qso_u01(r[4 + i]);
        }
    }
    if (cfg->randomizer_flags & QSO_RAND_MASSES) { /* :56-83: leg links +-10 %, payload U(0,1) kg at U(+-[0.1,0,0.1]) m, total mass kept */
        real legs = 0, legs0 = 0;
        for (int i = 0; i < 3; i++) {
            real lo = QSO_M_LEG[i] * (real)0.9, hi = QSO_M_LEG[i] * (real)1.1;
            e->m_leg[i] = lo + (hi - lo) * qso_u01(r[8 + i]);
            legs += 4 * e->m_leg[i]; legs0 += 4 * QSO_M_LEG[i];
        }
        e->m_pay = qso_u01(r[7]);
        e->r_pay[0] = (real)-0.1 + (real)0.2 * qso_u01(r[11]);
        e->r_pay[1] = 0;
        e->r_pay[2] = (real)-0.1 + (real)0.2 * qso_u01(r[13]);
        /* :43-47,61-65: total_mass sums EVERY URDF link, so the imu (0.001) and floating-base (1e-5) masses end up in the trunk too */
        e->m_trunk = QSO_M_TRUNK + (real)0.00101 + legs0 - legs - e->m_pay;
    }
    qso_model_build(&e->model, cfg->unit_inertia, e->m_trunk, e->m_leg, cfg->payload_soft ? 0 : e->m_pay, e->r_pay);
}

static void reset_env_to(qso_handle* h, int i, const real* st37);
static void reset_env(qso_handle* h, int i) { reset_env_to(h, i, NULL); }
/* st37 != NULL: reference-state initialisation (reference_state_initialization_wrapper.py:25-43, quadruped.py:521-525,
   gym_env.py:289-290): the robot is placed at the given state, the settle is skipped and _last_action stays zero (:284) */
static void reset_env_to(qso_handle* h, int i, const real* st37) {
    const qso_config* cfg = &h->cfg; qso_env* e = &h->env[i];
    /* gym_env.py:278-297 */
    e->episode++;
    e->sim_step = 0; e->env_step = 0;
    memset(e->last_action, 0, sizeof(e->last_action)); memset(e->last_filtered, 0, sizeof(e->last_filtered));
    randomize(cfg, e, i);
    /* quadruped.py:487-519, configs:23,26,31-36 */
    memset(&e->s, 0, sizeof(e->s));
    e->s.pos[2] = (real)0.32; e->s.quat[3] = 1;
    for (int L = 0; L < 4; L++) { e->s.q[3 * L] = 0; e->s.q[3 * L + 1] = PI / 4; e->s.q[3 * L + 2] = -PI / 2; }
    memset(e->warm, 0, sizeof(e->warm)); memset(e->foot_force, 0, sizeof(e->foot_force));
    memset(e->foot_contact, 0, sizeof(e->foot_contact)); e->n_invalid = 0;
    memset(e->tau_pd, 0, sizeof(e->tau_pd)); memset(e->tau_spring, 0, sizeof(e->tau_spring));
    qso_block_place(e);   /* _add_base_mass_offset (quadruped.py:778-819): the block appears next to the freshly spawned robot */
    /* interface_base.py:182-200: settle, sim counter frozen */
    if (st37) {
        memcpy(e->s.pos, st37, 3 * sizeof(real)); memcpy(e->s.quat, st37 + 3, 4 * sizeof(real)); memcpy(e->s.vlin, st37 + 7, 3 * sizeof(real));
        memcpy(e->s.vang, st37 + 10, 3 * sizeof(real)); memcpy(e->s.q, st37 + 13, 12 * sizeof(real)); memcpy(e->s.qd, st37 + 25, 12 * sizeof(real));
        qso_block_place(e);   /* (the reference leaves the block at the spawn pose and lets the constraint drag it; here it moves with the robot) */
    } else {
        real cmd[12]; for (int k = 0; k < 12; k++) cmd[k] = cfg->settle_cmd[k];
        for (int n = 0; n < cfg->settle_steps; n++) apply_and_step_mode(cfg, e, cmd, h->gravity, 1);
        for (int k = 0; k < 12; k++) e->last_action[k] = k < cfg->action_dim ? (real)cfg->settle_action[k] : 0;
    }
    if (cfg->action_space_mode == QSO_ACT_CPG) { /* hopf_network.py:62-63: r ~ 0.1 U(0,1), theta = PHI[0,:] */
        uint32_t rr[4]; qso_philox(cfg->seed, (uint32_t)(i + cfg->env_id_offset), 3, (uint32_t)e->episode, 0, rr);
        for (int L = 0; L < 4; L++) { e->cpg[L] = (real)0.1 * qso_u01(rr[L]); e->cpg[4 + L] = cfg->cpg_phi[L]; }
    }
    memset(&e->wrap, 0, sizeof(e->wrap));
    e->wrap.h_old = e->wrap.h_act = e->s.pos[2];   /* go_to_rest_wrapper.py:86-90 */
    e->wrap.armed = 1;                             /* _enable_landing, landing_wrapper_2.py:73-76 */
    task_reset(cfg, e);
    read_sensors(cfg, e, e->obs);
    add_noise(cfg, e, i, e->obs);
    /* gym_env.py:267-269 */
    int d = cfg->action_dim;
    for (int k = 0; k < d; k++) { e->xhist[k] = e->xhist[d + k] = e->last_action[k]; e->yhist[k] = e->yhist[d + k] = e->last_action[k]; }
}

int qso_create(const qso_config* cfg, qso_handle** out) {
    if (!cfg || !out) FAIL("null argument");
    if (cfg->n_envs <= 0) FAIL("n_envs must be positive");
    if (cfg->obs_dim > QSO_MAX_OBS || cfg->n_sensors > QSO_MAX_SENSORS) FAIL("observation too large");
    if (cfg->motor_control_mode == QSO_MOTOR_TORQUE && cfg->rl_interface) FAIL("TORQUE mode not implemented for the RL interface");
    int od = 0; for (int i = 0; i < cfg->n_sensors; i++) od += sensor_dim(cfg->sensors[i]);
    if (od != cfg->obs_dim) FAIL("obs_dim %d does not match sensor bundle (%d)", cfg->obs_dim, od);
    qso_handle* h = (qso_handle*)calloc(1, sizeof(*h));
    h->cfg = *cfg; h->gravity = cfg->gravity;
    h->env = (qso_env*)calloc((size_t)cfg->n_envs, sizeof(qso_env));
    for (int i = 0; i < cfg->n_envs; i++) {
        qso_env* e = &h->env[i];
        e->episode = -1;
        randomize(cfg, e, i); /* nominal-ish params so that physics-only calls work before the first reset */
        e->s.pos[2] = (real)0.32; e->s.quat[3] = 1;
        for (int L = 0; L < 4; L++) { e->s.q[3 * L + 1] = PI / 4; e->s.q[3 * L + 2] = -PI / 2; }
    }
    *out = h;
    return 0;
}
void qso_destroy(qso_handle* h) { if (h) { free(h->demo); free(h->env); free(h); } }

int qso_set_demo(qso_handle* h, const float* rows, int length) {
    if (!rows || length <= 0) FAIL("a demonstration needs at least one row");
    const size_t n = (size_t)length * (h->cfg.action_dim + 38);
    float* copy = (float*)malloc(n * sizeof(float));
    if (!copy) FAIL("out of memory");
    memcpy(copy, rows, n * sizeof(float));
    free(h->demo); h->demo = copy; h->demo_len = length;
    for (int i = 0; i < h->cfg.n_envs; i++) h->env[i].demo_len = length;
    return 0;
}
int qso_set_demo_counter(qso_handle* h, const uint8_t* mask, const int32_t* values) {
    for (int i = 0; i < h->cfg.n_envs; i++) if (!mask || mask[i]) {
        if (values[i] < 0 || values[i] >= (h->demo_len > 0 ? h->demo_len : 1)) FAIL("demo counter %d of environment %d outside the demonstration (%d rows)", values[i], i, h->demo_len);
        h->env[i].task.demo_counter = h->env[i].task.demo_start = values[i];
    }
    return 0;
}

int qso_reset(qso_handle* h, const uint8_t* mask) {
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int i = 0; i < h->cfg.n_envs; i++) if (!mask || mask[i]) reset_env(h, i);
    return 0;
}
int qso_reset_to(qso_handle* h, const uint8_t* mask, const real* states) {
    for (int i = 0; i < h->cfg.n_envs; i++) if (!mask || mask[i]) reset_env_to(h, i, states + (size_t)i * 37);
    return 0;
}
int qso_get_obs(qso_handle* h, float* obs) {
    for (int i = 0; i < h->cfg.n_envs; i++) memcpy(obs + (size_t)i * h->cfg.obs_dim, h->env[i].obs, h->cfg.obs_dim * sizeof(float));
    return 0;
}

/* what distinguishes the reference's six landing wrappers (env/wrappers/landing_wrapper*.py) */
typedef struct { int landing_family, trigger_jumping, pitch_takeoff, gains, exit_kind /* 0 done, 1 while flying, 2 while jumping */, one_shot; } wrap_traits;
static wrap_traits wrap_traits_of(int mode) {
    switch (mode) {
    case QSO_WRAP_LANDING:            return (wrap_traits){1, 0, 0, 1, 0, 0};
    case QSO_WRAP_LANDING2:           return (wrap_traits){1, 0, 0, 0, 1, 1};
    case QSO_WRAP_LANDING_BACKFLIP:   return (wrap_traits){1, 0, 1, 0, 0, 0};
    case QSO_WRAP_LANDING_BACKFLIP2:  return (wrap_traits){1, 0, 1, 0, 1, 1};
    case QSO_WRAP_LANDING_CONTINUOUS: return (wrap_traits){1, 1, 0, 0, 2, 0};
    default:                          return (wrap_traits){0, 0, 0, 0, 0, 0};
    }
}

/* one QuadrupedGymEnv.step of environment i (gym_env.py:227-256); returns 1 when the episode ended */
static int step_env(qso_handle* h, int i, const float* action_row, float* obs_row, float* rew, uint8_t* done, uint8_t* trunc) {
    const qso_config* cfg = &h->cfg; int d = cfg->action_dim;
    {
        qso_env* e = &h->env[i];
        real act[12], act_in[12];
        for (int k = 0; k < d; k++) act_in[k] = act[k] = action_row[k];
        /* scripted phases of the wrappers, one inner env.step per call (landing_wrapper*.py, go_to_rest_wrapper.py:43-80) */
        real kp_save[3], kd_save[3]; int swapped = 0;
        const real env_dt = (real)cfg->action_repeat * (real)cfg->dt;
        const wrap_traits wt = wrap_traits_of(cfg->wrapper_mode);
        if (wt.landing_family) {
            if (e->wrap.phase == QSO_PHASE_TAKEOFF) {
                if (wt.pitch_takeoff) { /* landing_wrapper_backflip.py:22,55-62: fixed take-off action until the pitch trigger */
                    static const real TAKE_OFF_ACTION[6] = {0, 1, -1, 0, 1, -1};
                    for (int k = 0; k < d; k++) act[k] = TAKE_OFF_ACTION[k % 6];
                } else if (e->wrap.timer > e->wrap.end) e->wrap.phase = QSO_PHASE_LANDING; /* landing_wrapper.py:47-54 + utils/timer.py */
                else { e->wrap.timer += env_dt; for (int k = 0; k < d; k++) act[k] = e->wrap.action[k]; }
            }
            if (e->wrap.phase == QSO_PHASE_LANDING) {
                for (int k = 0; k < d; k++) act[k] = cfg->landing_action[k];
                if (wt.gains) { /* only landing_wrapper.py:39 keeps the decorator: kp = 60, kd = 1.5 */
                    for (int k = 0; k < 3; k++) { kp_save[k] = e->kp[k]; kd_save[k] = e->kd[k]; e->kp[k] = cfg->landing_kp; e->kd[k] = cfg->landing_kd; }
                    swapped = 1;
                }
            }
        } else if (cfg->wrapper_mode == QSO_WRAP_GO_TO_REST && e->wrap.phase == QSO_PHASE_REST) { /* go_to_rest: ramp to the init action */
            real t = sim_time(cfg, e), t0 = e->wrap.t_start, t1 = t0 + (real)cfg->rest_time;
            for (int k = 0; k < d; k++) { /* interface_base.py:111-119 generate_ramp */
                real u0 = e->wrap.action[k], u1 = cfg->settle_action[k];
                act[k] = t < t0 ? u0 : (t > t1 ? u1 : u0 + (u1 - u0) * (t - t0) / (t1 - t0));
            }
            for (int k = 0; k < 3; k++) { kp_save[k] = e->kp[k]; kd_save[k] = e->kd[k]; e->kp[k] = cfg->rest_kp; e->kd[k] = cfg->rest_kd; }
            swapped = 1;
        }
        e->wrap.scripted = e->wrap.phase != QSO_PHASE_POLICY;
        for (int k = 0; k < d; k++) e->last_action[k] = act[k];
        if (cfg->enable_filter) { qso_filter_step(cfg->filt_b, cfg->filt_a, d, act, e->xhist, e->yhist, act); memcpy(e->last_filtered, act, d * sizeof(real)); }
        /* _interpolate_actions (:187-205) is an identity in the reference: _last_action/_last_filtered_action
           were overwritten with the current action at :230/:234 before the substeps run. */
        real cmd[12], cpgp[5];
        const int cpg = cfg->action_space_mode == QSO_ACT_CPG;
        if (cpg) cpg_params(cfg, act, cpgp);
        else if (cfg->rl_interface) qso_action_to_command(cfg, act, cmd);
        else for (int k = 0; k < 12; k++) cmd[k] = act[k];
        for (int s = 0; s < cfg->action_repeat; s++) {
            if (cpg) cpg_command(cfg, e, cpgp, cmd);   /* the oscillators tick at the physics rate (hopf_network.py:241-289) */
            apply_and_step(cfg, e, cmd, h->gravity); e->sim_step++;
            if (h->trace && i == h->trace_env) { /* monitor_state.py:66-85 */
                real* r = h->trace + (size_t)s * QSO_TRACE_DIM;
                r[0] = sim_time(cfg, e);
                memcpy(r + 1, e->s.pos, 3 * sizeof(real)); memcpy(r + 4, e->s.quat, 4 * sizeof(real)); memcpy(r + 8, e->s.vlin, 3 * sizeof(real));
                memcpy(r + 11, e->s.vang, 3 * sizeof(real)); memcpy(r + 14, e->s.q, 12 * sizeof(real)); memcpy(r + 26, e->s.qd, 12 * sizeof(real));
                memcpy(r + 38, e->tau_pd, 12 * sizeof(real)); memcpy(r + 50, e->tau_spring, 12 * sizeof(real));
                for (int f = 0; f < 4; f++) { r[62 + f] = e->foot_force[f]; r[66 + f] = (real)e->foot_contact[f]; }
            }
        }
        e->env_step++; e->total_steps++;
        task_on_step(cfg, e);
        real r = task_family_demo(cfg->task) ? demo_reward(h, e) : task_reward(cfg, e);
        int term = task_terminated(cfg, e);
        int dn = term || e->sim_step > cfg->max_sim_steps;
        if (dn) r += task_reward_end(cfg, e);
        if (swapped) for (int k = 0; k < 3; k++) { e->kp[k] = kp_save[k]; e->kd[k] = kd_save[k]; }
        if (cfg->wrapper_mode == QSO_WRAP_GO_TO_REST) { e->wrap.h_old = e->wrap.h_act; e->wrap.h_act = e->s.pos[2]; }
        if (!dn && wt.landing_family) {
            const int flying = !(e->foot_contact[0] || e->foot_contact[1] || e->foot_contact[2] || e->foot_contact[3]); /* quadruped.py:260-262 */
            const int trigger = wt.trigger_jumping ? e->task.is_jumping : e->task.switched;
            if (e->wrap.phase == QSO_PHASE_POLICY) {
                if (trigger && e->wrap.armed) { /* landing_wrapper.py:54-66 */
                    e->wrap.phase = QSO_PHASE_TAKEOFF;
                    e->wrap.timer = sim_time(cfg, e); e->wrap.end = e->wrap.timer + e->s.vlin[2] / (real)9.81;
                    for (int k = 0; k < d; k++) e->wrap.action[k] = act_in[k];
                }
            } else if (e->wrap.phase == QSO_PHASE_TAKEOFF) {
                if (wt.pitch_takeoff && qso_pitch_backflip(e->s.quat, e->task.switched) >= (real)(5.0 * PI / 8.0)) /* landing_wrapper_backflip.py:23-24 */
                    e->wrap.phase = QSO_PHASE_LANDING;
            } else if (e->wrap.phase == QSO_PHASE_LANDING) {
                /* landing_wrapper_2.py:41-45 (while flying), landing_wrapper_continuous.py:41-45 (while the task says jumping) */
                if ((wt.exit_kind == 1 && !flying) || (wt.exit_kind == 2 && !e->task.is_jumping)) {
                    e->wrap.phase = QSO_PHASE_POLICY;
                    if (wt.one_shot) e->wrap.armed = 0;   /* _enable_landing = False, landing_wrapper_2.py:68 */
                }
            }
        } else if (!dn && cfg->wrapper_mode == QSO_WRAP_GO_TO_REST && e->wrap.phase == QSO_PHASE_POLICY && e->task.switched &&
                   /* go_to_rest_wrapper.py:88-95 */
                   e->foot_contact[0] && e->foot_contact[1] && e->foot_contact[2] && e->foot_contact[3] && e->wrap.h_act - e->wrap.h_old > 0) {
            e->wrap.phase = QSO_PHASE_REST; e->wrap.t_start = sim_time(cfg, e);
            qso_command_to_action(cfg, e->s.q, e->wrap.action);   /* get_start_action, :55-57 */
        }
        read_sensors(cfg, e, e->obs);
        add_noise(cfg, e, i, e->obs);
        *rew = (float)r; *done = (uint8_t)dn; *trunc = (uint8_t)(dn && !term);
        if (dn && cfg->auto_reset) { memcpy(e->term_obs, e->obs, sizeof(e->obs)); reset_env(h, i); }
        if (obs_row) memcpy(obs_row, e->obs, cfg->obs_dim * sizeof(float));
        return dn;
    }
}

int qso_step(qso_handle* h, const float* actions, float* obs, float* rew, uint8_t* done, uint8_t* trunc) {
    const qso_config* cfg = &h->cfg; const int d = cfg->action_dim;
    if (task_family_demo(cfg->task) && !h->demo) FAIL("the DEMO tasks need a demonstration: qso_set_demo first");
    /* environments are independent (gym_env.py:132-137): with -fopenmp they spread over OMP_NUM_THREADS host threads (default: one).
       dynamic, 1: an environment that ends its episode settles 2500 substeps inside its step (250 steps' worth of work) */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int i = 0; i < cfg->n_envs; i++)
        step_env(h, i, actions + (size_t)i * d, obs + (size_t)i * cfg->obs_dim, rew + i, done + i, trunc + i);
    return 0;
}

/* Throughput harness of bench.py's cpu_baseline leg.  Every host thread owns a contiguous share of the environments and steps them
   `steps` times with the actions ring[t % n_ring][env][:] WITHOUT meeting the other threads between steps: the reference scaled over the
   host's cores as independent workers is the CPU's best case -- a reset's in-place settle (2500 substeps, gym_env.py:323-329) then delays
   its own thread only, not a whole lock-stepped batch.  resets (may be NULL) receives the number of episodes that ended. */
int qso_rollout(qso_handle* h, const float* ring, int n_ring, int steps, unsigned long long* resets) {
    const qso_config* cfg = &h->cfg; const int d = cfg->action_dim, n = cfg->n_envs;
    if (task_family_demo(cfg->task) && !h->demo) FAIL("the DEMO tasks need a demonstration: qso_set_demo first");
    if (n_ring <= 0 || steps < 0) FAIL("qso_rollout: bad ring / step count");
    unsigned long long total = 0;
#ifdef _OPENMP
#pragma omp parallel reduction(+ : total)
#endif
    {
        int nt = 1, me = 0;
#ifdef _OPENMP
        nt = omp_get_num_threads(); me = omp_get_thread_num();
#endif
        const int lo = (int)((long long)n * me / nt), hi = (int)((long long)n * (me + 1) / nt);
        float r; uint8_t dn, tr;
        for (int t = 0; t < steps; t++) {
            const float* a = ring + (size_t)(t % n_ring) * n * d;
            for (int i = lo; i < hi; i++) total += (unsigned long long)step_env(h, i, a + (size_t)i * d, NULL, &r, &dn, &tr);
        }
    }
    if (resets) *resets = total;
    return 0;
}

/* threads the environment loops of THIS library spread over (OpenMP); 1 unless a test or bench.py's cpu_baseline asks for more */
int qso_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n < 1 ? 1 : n);
#else
    (void)n;
#endif
    return 0;
}
int qso_set_trace(qso_handle* h, int env, real* rows) { h->trace_env = env; h->trace = env >= 0 ? rows : NULL; return 0; }

int qso_get_state(qso_handle* h, real* st) {
    for (int i = 0; i < h->cfg.n_envs; i++) {
        const qso_dyn* s = &h->env[i].s; real* o = st + (size_t)i * 37;
        memcpy(o, s->pos, 3 * sizeof(real)); memcpy(o + 3, s->quat, 4 * sizeof(real)); memcpy(o + 7, s->vlin, 3 * sizeof(real));
        memcpy(o + 10, s->vang, 3 * sizeof(real)); memcpy(o + 13, s->q, 12 * sizeof(real)); memcpy(o + 25, s->qd, 12 * sizeof(real));
    }
    return 0;
}
int qso_set_state(qso_handle* h, const real* st) {
    for (int i = 0; i < h->cfg.n_envs; i++) {
        qso_dyn* s = &h->env[i].s; const real* o = st + (size_t)i * 37;
        memcpy(s->pos, o, 3 * sizeof(real)); memcpy(s->quat, o + 3, 4 * sizeof(real)); memcpy(s->vlin, o + 7, 3 * sizeof(real));
        memcpy(s->vang, o + 10, 3 * sizeof(real)); memcpy(s->q, o + 13, 12 * sizeof(real)); memcpy(s->qd, o + 25, 12 * sizeof(real));
        memset(h->env[i].warm, 0, sizeof(h->env[i].warm));
        memset(h->env[i].warm_sup, 0, sizeof(h->env[i].warm_sup));
        qso_block_place(&h->env[i]);
    }
    return 0;
}

/* contact warm start of the feet (the normal impulses of the last substep): what qso_set_state zeroes; lets a test re-seat the oracle in a
 * device state that carries its warm start (tests/test_gpu_parity.py::test_full_size_oracle_sampled) */
int qso_set_warm(qso_handle* h, const real* w) {
    for (int i = 0; i < h->cfg.n_envs; i++)
        for (int L = 0; L < 4; L++) h->env[i].warm[L] = w[(size_t)i * 4 + L];
    return 0;
}

size_t qso_snapshot_size(qso_handle* h) { return (size_t)h->cfg.n_envs * sizeof(qso_env); }
int qso_snapshot(qso_handle* h, void* buf) { if (!buf) FAIL("null buffer"); memcpy(buf, h->env, qso_snapshot_size(h)); return 0; }
int qso_restore(qso_handle* h, const void* buf) { if (!buf) FAIL("null buffer"); memcpy(h->env, buf, qso_snapshot_size(h)); return 0; }

static void pack_params(const qso_env* e, real* o) {
    o[0] = e->mu;
    for (int k = 0; k < 3; k++) { o[1 + k] = e->k[k]; o[4 + k] = e->b[k]; o[7 + k] = e->rest[k]; o[10 + k] = e->kp[k]; o[13 + k] = e->kd[k]; o[17 + k] = e->m_leg[k]; o[21 + k] = e->r_pay[k]; }
    o[16] = e->m_trunk; o[20] = e->m_pay;
}

int qso_get_info(qso_handle* h, int which, real* out) {
    for (int i = 0; i < h->cfg.n_envs; i++) {
        const qso_env* e = &h->env[i]; const qso_task* t = &e->task;
        switch (which) {
        case QSO_INFO_FOOT_FORCE: for (int k = 0; k < 4; k++) out[4 * i + k] = e->foot_force[k]; break;
        case QSO_INFO_FOOT_CONTACT: for (int k = 0; k < 4; k++) out[4 * i + k] = e->foot_contact[k]; break;
        case QSO_INFO_TORQUE: for (int k = 0; k < 12; k++) out[12 * i + k] = e->tau_pd[k]; break;
        case QSO_INFO_SPRING_TORQUE: for (int k = 0; k < 12; k++) out[12 * i + k] = e->tau_spring[k]; break;
        case QSO_INFO_N_INVALID: out[i] = e->n_invalid; break;
        case QSO_INFO_PARAMS: pack_params(e, out + 24 * i); break;
        case QSO_INFO_COUNTERS: out[4 * i] = e->sim_step; out[4 * i + 1] = e->env_step; out[4 * i + 2] = e->episode; out[4 * i + 3] = e->total_steps; break;
        case QSO_INFO_LAST_ACTION: for (int k = 0; k < 12; k++) out[12 * i + k] = e->last_action[k]; break;
        case QSO_INFO_TERMINAL_OBS: for (int k = 0; k < h->cfg.obs_dim; k++) out[(size_t)i * h->cfg.obs_dim + k] = e->term_obs[k]; break;
        case QSO_INFO_WRAPPER: out[4 * i] = e->wrap.phase; out[4 * i + 1] = e->wrap.scripted; out[4 * i + 2] = e->wrap.timer; out[4 * i + 3] = e->wrap.end; break;
        case QSO_INFO_TASK: {
            real* o = out + 48 * i; memset(o, 0, 48 * sizeof(real));
            o[0] = t->switched; o[1] = t->all_air; o[2] = t->is_jumping; o[3] = t->t_takeoff;
            o[4] = t->pose_to[0]; o[5] = t->pose_to[1]; o[6] = t->pose_to[2]; o[7] = t->yaw_to; o[8] = t->init_h;
            o[9] = t->max_flight; o[10] = t->max_fwd; o[11] = t->max_pitch; o[12] = t->rel_max_h; o[13] = t->max_dx; o[14] = t->max_h;
            o[15] = t->cum_fwd; o[16] = t->cum_ft; o[17] = t->old_fwd; o[18] = t->actual_fwd; o[19] = t->bf_max_pitch;
            o[20] = t->jump_count; o[21] = t->good_jumps; o[22] = t->sum_fwd; o[23] = t->sum_flogf; o[24] = t->sum_height; o[25] = t->sum_perf;
            o[26] = t->max_perf; o[27] = t->last_perf; o[28] = t->max_jump_h; o[29] = t->first_jump; o[30] = t->end_jump;
            for (int k = 0; k < 3; k++) { o[32 + k] = t->pos[k]; o[35 + k] = t->vel[k]; o[38 + k] = t->rpy[k]; }
            o[41] = e->n_invalid; o[42] = e->foot_force[0] + e->foot_force[1] + e->foot_force[2] + e->foot_force[3]; o[43] = e->sim_step;
            o[44] = t->demo_counter; o[45] = t->demo_start;
            break; }
        default: FAIL("unknown info id %d", which);
        }
    }
    return 0;
}

int qso_set_params(qso_handle* h, int which, const real* v) {
    for (int i = 0; i < h->cfg.n_envs; i++) {
        qso_env* e = &h->env[i];
        switch (which) {
        case QSO_PARAM_MU: e->mu = v[i]; break;
        case QSO_PARAM_SPRING_K: for (int k = 0; k < 3; k++) e->k[k] = v[3 * i + k]; break;
        case QSO_PARAM_SPRING_B: for (int k = 0; k < 3; k++) e->b[k] = v[3 * i + k]; break;
        case QSO_PARAM_KP: for (int k = 0; k < 3; k++) e->kp[k] = v[3 * i + k]; break;
        case QSO_PARAM_KD: for (int k = 0; k < 3; k++) e->kd[k] = v[3 * i + k]; break;
        case QSO_PARAM_ALL: {
            const real* o = v + 24 * i;
            e->mu = o[0];
            for (int k = 0; k < 3; k++) { e->k[k] = o[1 + k]; e->b[k] = o[4 + k]; e->rest[k] = o[7 + k]; e->kp[k] = o[10 + k]; e->kd[k] = o[13 + k]; e->m_leg[k] = o[17 + k]; e->r_pay[k] = o[21 + k]; }
            e->m_trunk = o[16]; e->m_pay = o[20];
            qso_model_build(&e->model, h->cfg.unit_inertia, e->m_trunk, e->m_leg, h->cfg.payload_soft ? 0 : e->m_pay, e->r_pay);
            qso_block_place(e);
            break; }
        default: FAIL("unknown param id %d", which);
        }
    }
    return 0;
}

int qso_set_task(qso_handle* h, const real* in) {
    for (int i = 0; i < h->cfg.n_envs; i++) {
        qso_env* e = &h->env[i]; qso_task* t = &e->task; const real* o = in + 48 * i;
        t->switched = o[0] != 0; t->all_air = o[1] != 0; t->is_jumping = o[2] != 0; t->t_takeoff = o[3];
        t->pose_to[0] = o[4]; t->pose_to[1] = o[5]; t->pose_to[2] = o[6]; t->yaw_to = o[7]; t->init_h = o[8];
        t->max_flight = o[9]; t->max_fwd = o[10]; t->max_pitch = o[11]; t->rel_max_h = o[12]; t->max_dx = o[13]; t->max_h = o[14];
        t->cum_fwd = o[15]; t->cum_ft = o[16]; t->old_fwd = o[17]; t->actual_fwd = o[18]; t->bf_max_pitch = o[19];
        t->jump_count = o[20]; t->good_jumps = o[21]; t->sum_fwd = o[22]; t->sum_flogf = o[23]; t->sum_height = o[24]; t->sum_perf = o[25];
        t->max_perf = o[26]; t->last_perf = o[27]; t->max_jump_h = o[28]; t->first_jump = o[29]; t->end_jump = o[30];
        for (int k = 0; k < 3; k++) { t->pos[k] = o[32 + k]; t->vel[k] = o[35 + k]; t->rpy[k] = o[38 + k]; }
        e->n_invalid = (int)o[41];
        e->foot_force[0] = o[42]; e->foot_force[1] = e->foot_force[2] = e->foot_force[3] = 0;
        e->sim_step = (int)o[43];
    }
    return 0;
}
int qso_eval_reward(qso_handle* h, int which, real* out) {
    for (int i = 0; i < h->cfg.n_envs; i++) {
        const qso_env* e = &h->env[i];
        out[i] = which == 0 ? task_reward(&h->cfg, e) : which == 1 ? task_reward_end(&h->cfg, e) : (real)task_terminated(&h->cfg, e);
    }
    return 0;
}
