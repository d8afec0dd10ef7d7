/* qso_internal.h -- ORACLE internals (test infrastructure, not product code). */
#ifndef QSO_INTERNAL_H
#define QSO_INTERNAL_H
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include "qso.h"

#define NB 13   /* rigid bodies after merging fixed joints: 0 = base+trunk+imu(+payload), 1+3L+j = leg L link j */
#define NJ 12
#define NV 18
#define NVX 24  /* + the six velocities of the payload block when it is a body of its own */
#define QSO_MAX_CONTACTS 64   /* generalized velocity: [w_b(3) v_b(3)] in base coordinates, then qd(12) */

/* Go1 rigid-body model, restated from go1/go1_description/urdf/go1.urdf (see qso_model.c). */
typedef struct {
    real mass[NB];
    real com[NB][3];       /* in link frame */
    real Ic[NB][3][3];     /* about COM, link axes */
    real jpos[NB][3];      /* joint origin in parent frame */
    int jaxis[NB];         /* 0 = x, 1 = y */
    int parent[NB];
    real I6[NB][6][6];     /* spatial inertia at link origin, [ang;lin] ordering */
    real total_mass;
} qso_model;

typedef struct {
    real pos[3], quat[4], vlin[3], vang[3], q[NJ], qd[NJ];
} qso_dyn;

typedef struct {
    int switched, all_air, is_jumping;
    real t_takeoff, pose_to[3], yaw_to, init_h;
    real max_flight, max_fwd, max_pitch, rel_max_h, max_dx, max_h;
    real old_tau[NJ], new_tau[NJ];
    real pos[3], vel[3], rpy[3];
    real cum_fwd, cum_ft;
    real old_fwd, actual_fwd;
    real bf_max_pitch;
    /* TaskContinuousJumping2: sufficient statistics of fwd_array / height_array / performance_array */
    real jump_count, good_jumps, sum_fwd, sum_flogf, sum_height, sum_perf, max_perf, last_perf, max_jump_h, first_jump, end_jump;
    /* TaskJumpingDemo: row of the demonstration the next step is compared with, and its value when the episode began */
    int demo_counter, demo_start;
} qso_task;

typedef struct {
    qso_model model;
    qso_dyn s;
    /* per-env parameters */
    real mu, k[3], b[3], rest[3], kp[3], kd[3];
    real m_trunk, m_leg[3], m_pay, r_pay[3];
    /* results of the last physics substep */
    real foot_force[4];
    int foot_contact[4];
    int n_invalid;
    /* every contact point of the last substep as getContactPoints() would list it (PyBullet ids: body 0 plane, 1 robot, 2 payload block;
       link -1 base / plane, 0 trunk, 2/6/10/14 hips, 3/7/11/15 thighs, 4/8/12/16 calves, 5/9/13/17 feet) */
    struct { int body_a, body_b, link_a, link_b; real dist, force; } contacts[QSO_MAX_CONTACTS];
    int n_contacts;
    int manifold_mode;   /* qso_phys_set_manifold: 0 = foot + two support points per leg, three without the foot (what the kernels build); 3 = two whatever the foot does (rounds 2 - 5); 1 = experiment: up to four
                          * points per collision primitive, as Bullet's persistent manifolds can hold; 2 = experiment: mode 0 with
                          * the support points' normal rows warm-started like the feet's (x cfg.warmstart) while the same candidate stays selected */
    real warm_sup[4][5]; /* mode 2: last normal impulse of leg L's candidate i (0 once it is not selected) */
    real warm[4];
    /* payload block as its own body (cfg->payload_soft): position of its centre, orientation, velocities (world); constraint impulses of
       the last substep and the distance between the two pivots */
    struct { real pos[3], quat[4], v[3], w[3], lam[6], gap; } blk;
    real tau_pd[NJ], tau_spring[NJ];
    /* env-level */
    real last_action[12], last_filtered[12], xhist[24], yhist[24];
    int sim_step, env_step, episode;
    uint32_t total_steps;
    qso_task task;
    real cpg[8];            /* Hopf oscillators: r[4], theta[4] */
    struct { int phase, scripted, armed; real timer, end, t_start, h_old, h_act, action[12]; } wrap; /* landing / go-to-rest machine */
    float obs[QSO_MAX_OBS], term_obs[QSO_MAX_OBS];
    int demo_len;           /* rows of the handle's demonstration (copy, for the task functions) */
} qso_env;

struct qso_handle {
    qso_config cfg;
    qso_env* env;
    real gravity;
    real* trace; int trace_env;
    float* demo; int demo_len;   /* qso_set_demo */
};

/* qso_model.c */
void qso_model_build(qso_model* M, const float (*unit)[6], real m_trunk, const real* m_leg3, real m_pay, const real* r_pay);
void qso_block_place(qso_env* e);   /* puts the payload block where the fixed constraint wants it, at the base's velocity */
extern const real QSO_TRUNK_I[6], QSO_HIP_I[6], QSO_THIGH_I[6], QSO_CALF_I[6];   /* URDF tensors (FR-leg magnitudes) */
extern const real QSO_M_TRUNK, QSO_M_LEG[3];
extern const real QSO_JOINT_LO[3], QSO_JOINT_HI[3];

/* qso_phys.c */
void qso_quat_to_mat(const real* q, real R[3][3]);
void qso_physics_substep(const qso_config* cfg, qso_env* e, const real* tau, real g);

/* small vector helpers */
static inline void v3cross(const real* a, const real* b, real* c) {
    real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    c[0] = x; c[1] = y; c[2] = z;
}
static inline real v3dot(const real* a, const real* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void m3v(const real R[3][3], const real* v, real* o) {
    real x = R[0][0] * v[0] + R[0][1] * v[1] + R[0][2] * v[2];
    real y = R[1][0] * v[0] + R[1][1] * v[1] + R[1][2] * v[2];
    real z = R[2][0] * v[0] + R[2][1] * v[1] + R[2][2] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline void m3tv(const real R[3][3], const real* v, real* o) {
    real x = R[0][0] * v[0] + R[1][0] * v[1] + R[2][0] * v[2];
    real y = R[0][1] * v[0] + R[1][1] * v[1] + R[2][1] * v[2];
    real z = R[0][2] * v[0] + R[1][2] * v[1] + R[2][2] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
#endif
