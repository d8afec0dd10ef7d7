// qs_env.h -- one QuadrupedGymEnv.step() / reset() per environment record, on top of Sim<T>::substep.
// Reference lines are relative to quadruped_spring/ ; the same restatement in float64 lives in oracle/qso_env.c.
#pragma once
#include "qs_core.h"

namespace qs {

QS_FN float i2f(int v) { union { int i; float f; } u; u.i = v; return u.f; }
QS_FN int f2i(float v) { union { int i; float f; } u; u.f = v; return u.i; }

template <class T, bool CONE = false, bool HOT = false, bool SOFT = false> struct Env {
    using V = typename T::V;
    using M = typename T::M;
    using S = Sim<T, CONE, HOT, SOFT>;
    using V3v = V3<V>;
    static constexpr float PI = 3.14159265358979323846f;

    // ---- analytic leg IK (quadruped.py:399-438), side sign -1 for right legs = T::sy()
    static QS_FN void leg_ik(const qs_config& cfg, V x, V y, V z, V* q) {
        const float sh = cfg.leg_len[0], el = cfg.leg_len[1], wr = cfg.leg_len[2];
        V D = (y * y + z * z - sh * sh + x * x - el * el - wr * wr) * (1.0f / (2.0f * wr * el));
        D = clampv<V>(D, V(-1.0f), V(1.0f));
        V sw = -qsqrt(V(1.0f) - D * D);          // sin(wrist); cos(wrist) = D: the point (D, sw) is on the unit circle
        V wrist = qatan2(sw, D);
        V sc = qmax(y * y + z * z - sh * sh, V(0.0f));
        V rt = qsqrt(sc);
        V shoulder = -qatan2(z, y) - qatan2(rt, T::sy() * sh);
        V elbow = qatan2(-x, rt) - qatan2(sw * wr, D * wr + el);
        q[0] = -shoulder; q[1] = elbow; q[2] = wrist;
    }
    // ---- analytic leg FK + Jacobian (quadruped.py:348-392)
    static QS_FN void leg_fk(const qs_config& cfg, const V* q, V* J, V* p) {
        const float l1 = cfg.leg_len[0], l2 = cfg.leg_len[1], l3 = cfg.leg_len[2];
        V sg = T::sy();
        V s1, s2, s3, c1, c2, c3; qsincos(q[0], s1, c1); qsincos(q[1], s2, c2); qsincos(q[2], s3, c3);
        V c23 = c2 * c3 - s2 * s3, s23 = s2 * c3 + c2 * s3;
        V zero = V(0.0f);
        J[0] = zero; J[3] = -sg * l1 * s1 + c2 * c1 * l2 + c23 * c1 * l3; J[6] = sg * l1 * c1 + c2 * s1 * l2 + c23 * s1 * l3;
        J[1] = -c23 * l3 - c2 * l2; J[4] = -s2 * s1 * l2 - s23 * s1 * l3; J[7] = s2 * c1 * l2 + s23 * c1 * l3;
        J[2] = -c23 * l3; J[5] = -s23 * s1 * l3; J[8] = s23 * c1 * l3;
        p[0] = -s23 * l3 - s2 * l2;
        p[1] = sg * c1 * l1 + (s1 * c23) * l3 + c2 * s1 * l2;
        p[2] = sg * s1 * l1 - (c1 * c23) * l3 - c1 * c2 * l2;
    }
    // ---- pybullet getEulerFromQuaternion (quadruped.py:131-139)
    static QS_FN void quat_to_rpy(V x, V y, V z, V w, V& roll, V& pitch, V& yaw) {
        V sarg = V(-2.0f) * (x * z - w * y);
        M lo = qle(sarg, V(-0.99999f)), hi = qge(sarg, V(0.99999f));
        V p = qasin(clampv<V>(sarg, V(-1.0f), V(1.0f)));
        V r = qatan2(V(2.0f) * (y * z + w * x), w * w - x * x - y * y + z * z);
        V yw = qatan2(V(2.0f) * (x * y + w * z), w * w + x * x - y * y - z * z);
        pitch = p; roll = r; yaw = yw;
        if (T::any(qor(lo, hi))) {   // gimbal lock (pitch = +-90 degrees): pybullet's special case
            pitch = qsel(lo, V(-0.5f * PI), qsel(hi, V(0.5f * PI), p));
            roll = qsel(qor(lo, hi), V(0.0f), r);
            yaw = qsel(lo, V(2.0f) * qatan2(x, -y), qsel(hi, V(2.0f) * qatan2(-x, y), yw));
        }
    }
    // ---- PitchBackFlip sensor (robot_sensors.py:333-340)
    static QS_FN V pitch_backflip(V x, V y, V z, V w, V switched) {
        V d = x * x + y * y + z * z + w * w, sc = V(2.0f) * qrcp(d);
        V r20 = (x * z - w * y) * sc, r22 = V(1.0f) - (x * x + y * y) * sc;
        V pitch = -qatan2(-r20, r22);
        return qsel(qand(qlt(pitch, V(0.0f)), qgt(switched, V(0.5f))), pitch + 2.0f * PI, pitch);
    }

    // ---- action -> own-leg motor command (action_interface.py:14-65, interface_base.py:84-90, motor_interface.py:70-80)
    static QS_FN void action_to_command(const qs_config& cfg, const V* act, V* cmd) {
        V a[3];
        if (!cfg.rl_interface) {  // raw motor commands (gym_env.py:212-214)
#pragma unroll
            for (int j = 0; j < 3; j++) cmd[j] = act[12 + j];
            return;
        }
        M front = qgt(T::fx(), V(0.0f));
        if (cfg.action_space_mode == QS_ACT_DEFAULT) {
#pragma unroll
            for (int j = 0; j < 3; j++) a[j] = act[12 + j];
        } else if (cfg.action_space_mode == QS_ACT_SYMMETRIC) {
#pragma unroll
            for (int j = 0; j < 3; j++) {
                V t = qsel(front, act[j], act[3 + j]);
                a[j] = j == cfg.symm_idx ? t * (-T::sy()) : t;  // left legs mirror index symm_idx
            }
        } else {   // the dropped joint takes 0, the two others the pairs (front, rear) in order; constant indices: act[] stays in registers
            const int sx = cfg.symm_idx;
            a[0] = sx == 0 ? V(0.0f) : qsel(front, act[0], act[2]);
            a[1] = sx == 1 ? V(0.0f) : (sx < 1 ? qsel(front, act[0], act[2]) : qsel(front, act[1], act[3]));
            a[2] = sx == 2 ? V(0.0f) : qsel(front, act[1], act[3]);
        }
        V s[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            V lo = T::ld_leg(cfg.cmd_lo, j, 3), hi = T::ld_leg(cfg.cmd_hi, j, 3);
            V c = clampv<V>(a[j], V(-1.0f), V(1.0f));
            s[j] = clampv<V>(lo + V(0.5f) * (c + 1.0f) * (hi - lo), lo, hi);
        }
        if (cfg.motor_control_mode == QS_MOTOR_CARTESIAN_PD) leg_ik(cfg, s[0], s[1], s[2], cmd);
        else { cmd[0] = s[0]; cmd[1] = s[1]; cmd[2] = s[2]; }
    }

    // ---- Hopf CPG action layer (hopf_network.py:117-173, 241-289): one oscillator per lane
    static QS_FN void cpg_command(const qs_config& cfg, const V* p, V& r, V& th, V* cmd) {
        const float dt = (float)cfg.dt;
        const V zero = V(0.0f);
        V s_th, c_th; qsincos(th, s_th, c_th);
        V rd = (p[2] - r * r) * r * cfg.cpg_alpha;                                                   // :149
        V td = qsel(qgt(s_th, zero), p[0], p[1]);                                                   // :152-156
#define QS_CPG_COUPLE(J)                                                                             \
    {                                                                                                \
        V rj = T::template bcast<J>(r), tj = T::template bcast<J>(th);                               \
        V sj, cj; qsincos(tj - th - T::ld_leg(cfg.cpg_phi, J, 4), sj, cj);                           \
        td = td + qsel(T::is_leg(J), zero, rj * cfg.cpg_coupling * sj);                              \
    }
        QS_CPG_COUPLE(0) QS_CPG_COUPLE(1) QS_CPG_COUPLE(2) QS_CPG_COUPLE(3)                          // :159-162
#undef QS_CPG_COUPLE
        r = r + rd * dt;
        V t2 = th + td * dt;
        th = t2 - qfloor(t2 * (1.0f / (2.0f * PI))) * (2.0f * PI);                                   // :170
        qsincos(th, s_th, c_th);
        V x = -p[3] * r * c_th;                                                                // :128
        V z = qsel(qgt(s_th, zero), s_th * cfg.cpg_clearance, s_th * cfg.cpg_penetration) - p[4];    // :129-132
        leg_ik(cfg, x, T::sy() * cfg.leg_len[0], z, cmd);
    }

    // ---- record <-> registers
    static QS_FN void load_state(const float* rec, typename S::State& s) {
        s.pos = mk3<V>(T::ld(rec, R_POS), T::ld(rec, R_POS + 1), T::ld(rec, R_POS + 2));
        s.qx = T::ld(rec, R_QUAT); s.qy = T::ld(rec, R_QUAT + 1); s.qz = T::ld(rec, R_QUAT + 2); s.qw = T::ld(rec, R_QUAT + 3);
        s.vlin = mk3<V>(T::ld(rec, R_VLIN), T::ld(rec, R_VLIN + 1), T::ld(rec, R_VLIN + 2));
        s.vang = mk3<V>(T::ld(rec, R_VANG), T::ld(rec, R_VANG + 1), T::ld(rec, R_VANG + 2));
#pragma unroll
        for (int j = 0; j < 3; j++) { s.q[j] = T::ld_leg(rec, R_Q + j, 3); s.qd[j] = T::ld_leg(rec, R_QD + j, 3); }
        s.warm = T::ld_leg(rec, R_WARM, 1);
    }
    static QS_FN void store_state(float* rec, const typename S::State& s, const typename S::Out& o) {
        T::st(rec, R_POS, s.pos.x); T::st(rec, R_POS + 1, s.pos.y); T::st(rec, R_POS + 2, s.pos.z);
        T::st(rec, R_QUAT, s.qx); T::st(rec, R_QUAT + 1, s.qy); T::st(rec, R_QUAT + 2, s.qz); T::st(rec, R_QUAT + 3, s.qw);
        T::st(rec, R_VLIN, s.vlin.x); T::st(rec, R_VLIN + 1, s.vlin.y); T::st(rec, R_VLIN + 2, s.vlin.z);
        T::st(rec, R_VANG, s.vang.x); T::st(rec, R_VANG + 1, s.vang.y); T::st(rec, R_VANG + 2, s.vang.z);
#pragma unroll
        for (int j = 0; j < 3; j++) {
            T::st_leg(rec, R_Q + j, 3, s.q[j]); T::st_leg(rec, R_QD + j, 3, s.qd[j]);
            T::st_leg(rec, R_TAU_PD + j, 3, o.tau_pd[j]); T::st_leg(rec, R_TAU_SPRING + j, 3, o.tau_spring[j]);
        }
        T::st_leg(rec, R_WARM, 1, s.warm);
        T::st_leg(rec, R_FOOT_FORCE, 1, o.foot_force); T::st_leg(rec, R_FOOT_CONTACT, 1, o.foot_contact);
        T::st(rec, R_N_INVALID, o.n_invalid);
    }
    static QS_FN void load_par(const qs_config& cfg, const float* rec, typename S::Par& P) {
        const float* p = rec + R_PARAMS;
        P.mu = T::ld(p, P_MU);
#pragma unroll
        for (int j = 0; j < 3; j++) {
            P.k[j] = T::ld(p, P_K + j); P.b[j] = T::ld(p, P_B + j); P.rest[j] = T::ld(p, P_REST + j);
            P.kp[j] = T::ld(p, P_KP + j); P.kd[j] = T::ld(p, P_KD + j); P.m_leg[j] = T::ld(p, P_M_LEG + j);
        }
        S::build_base(cfg, P, T::ld(p, P_M_TRUNK), T::ld(p, P_M_PAY), mk3<V>(T::ld(p, P_R_PAY), T::ld(p, P_R_PAY + 1), T::ld(p, P_R_PAY + 2)));
    }

    // ---- task state machine (tasks/task_base.py:61-166, 222-280) on replicated values
    struct Task {
        V switched, all_air, is_jumping, t_takeoff, pose_to[3], yaw_to, init_h, max_flight, max_fwd, max_pitch, rel_max_h, max_dx,
            max_h, cum_fwd, cum_ft, old_fwd, actual_fwd, bf_max_pitch;
        V jump_count, good_jumps, sum_fwd, sum_flogf, sum_height, sum_perf, max_perf, last_perf, max_jump_h, first_jump, end_jump;
        V pos[3], vel[3], rpy[3];
        V dtau2;  // |old_torque - new_torque|^2
    };
    static QS_FN void load_task(const float* rec, Task& t) {
        const float* p = rec + R_TASK;
        t.switched = T::ld(p, T_SWITCHED); t.all_air = T::ld(p, T_ALL_AIR); t.is_jumping = T::ld(p, T_IS_JUMPING); t.t_takeoff = T::ld(p, T_TAKEOFF);
#pragma unroll
        for (int k = 0; k < 3; k++) { t.pose_to[k] = T::ld(p, T_POSE_TO + k); t.pos[k] = V(0.0f); t.vel[k] = V(0.0f); t.rpy[k] = V(0.0f); }   // the pose cache is a result (info block): task_on_step fills it before any use
        t.yaw_to = T::ld(p, T_YAW_TO); t.init_h = T::ld(p, T_INIT_H); t.max_flight = T::ld(p, T_MAX_FLIGHT); t.max_fwd = T::ld(p, T_MAX_FWD);
        t.max_pitch = T::ld(p, T_MAX_PITCH); t.rel_max_h = T::ld(p, T_REL_MAX_H); t.max_dx = T::ld(p, T_MAX_DX); t.max_h = T::ld(p, T_MAX_H);
        t.cum_fwd = T::ld(p, T_CUM_FWD); t.cum_ft = T::ld(p, T_CUM_FT); t.old_fwd = T::ld(p, T_OLD_FWD); t.actual_fwd = T::ld(p, T_ACTUAL_FWD);
        t.bf_max_pitch = T::ld(p, T_BF_MAX_PITCH);
        t.jump_count = T::ld(p, T_JUMP_COUNT); t.good_jumps = T::ld(p, T_GOOD_JUMPS); t.sum_fwd = T::ld(p, T_SUM_FWD); t.sum_flogf = T::ld(p, T_SUM_FLOGF);
        t.sum_height = T::ld(p, T_SUM_HEIGHT); t.sum_perf = T::ld(p, T_SUM_PERF); t.max_perf = T::ld(p, T_MAX_PERF); t.last_perf = T::ld(p, T_LAST_PERF);
        t.max_jump_h = T::ld(p, T_MAX_JUMP_H); t.first_jump = T::ld(p, T_FIRST_JUMP); t.end_jump = T::ld(p, T_END_JUMP);
        t.dtau2 = V(0.0f);
    }
    static QS_FN void store_task(float* rec, const Task& t) {
        float* p = rec + R_TASK;
        T::st(p, T_SWITCHED, t.switched); T::st(p, T_ALL_AIR, t.all_air); T::st(p, T_IS_JUMPING, t.is_jumping); T::st(p, T_TAKEOFF, t.t_takeoff);
#pragma unroll
        for (int k = 0; k < 3; k++) { T::st(p, T_POSE_TO + k, t.pose_to[k]); T::st(rec, R_POSE_CACHE + k, t.pos[k]); T::st(rec, R_POSE_CACHE + 3 + k, t.vel[k]); T::st(rec, R_POSE_CACHE + 6 + k, t.rpy[k]); }
        T::st(p, T_YAW_TO, t.yaw_to); T::st(p, T_INIT_H, t.init_h); T::st(p, T_MAX_FLIGHT, t.max_flight); T::st(p, T_MAX_FWD, t.max_fwd);
        T::st(p, T_MAX_PITCH, t.max_pitch); T::st(p, T_REL_MAX_H, t.rel_max_h); T::st(p, T_MAX_DX, t.max_dx); T::st(p, T_MAX_H, t.max_h);
        T::st(p, T_CUM_FWD, t.cum_fwd); T::st(p, T_CUM_FT, t.cum_ft); T::st(p, T_OLD_FWD, t.old_fwd); T::st(p, T_ACTUAL_FWD, t.actual_fwd);
        T::st(p, T_BF_MAX_PITCH, t.bf_max_pitch);
        T::st(p, T_JUMP_COUNT, t.jump_count); T::st(p, T_GOOD_JUMPS, t.good_jumps); T::st(p, T_SUM_FWD, t.sum_fwd); T::st(p, T_SUM_FLOGF, t.sum_flogf);
        T::st(p, T_SUM_HEIGHT, t.sum_height); T::st(p, T_SUM_PERF, t.sum_perf); T::st(p, T_MAX_PERF, t.max_perf); T::st(p, T_LAST_PERF, t.last_perf);
        T::st(p, T_MAX_JUMP_H, t.max_jump_h); T::st(p, T_FIRST_JUMP, t.first_jump); T::st(p, T_END_JUMP, t.end_jump);
    }
    static QS_FN bool continuous(int task) { return task == QS_TASK_CONT_JUMPING_FORWARD || task == QS_TASK_CONT_JUMPING_FORWARD2; }
    static QS_FN bool continuous2(int task) { return task == QS_TASK_CONT_JUMPING_FORWARD3 || task == QS_TASK_CONT_JUMPING_FORWARD_PPO || task == QS_TASK_CONT_JUMPING_FORWARD_DEMO; }
    static QS_FN bool demo_task(int task) { return task >= QS_TASK_JUMPING_IN_PLACE_DEMO && task <= QS_TASK_CONT_JUMPING_FORWARD_DEMO; }
    // get_entropy_fwd (task_base.py:376-383) from the sums: -sum p log2 p = log2 S - (sum f log2 f) / S over max(n, 3) entries
    static QS_FN V cj2_entropy(const Task& t) {
        const float il2 = 1.4426950408889634f;
        V n = qmax(t.jump_count, V(3.0f));
        V S = qmax(t.sum_fwd, V(1e-30f));
        V h = (qlog(S) * il2 - t.sum_flogf * qrcp(S)) * qrcp(qlog(n) * il2);
        return qsel(qor(qlt(t.jump_count, V(0.5f)), qlt(t.sum_fwd, V(0.05f))), V(0.0f), h);
    }
    static QS_FN V jump_distance(const Task& t) {  // task_base.py:108-116
        V dx = t.pos[0] - t.pose_to[0], dy = t.pos[1] - t.pose_to[1];
        V sn, cs; qsincos(t.yaw_to, sn, cs);
        return qmax(cs * dx - sn * dy, V(0.0f));
    }
    // `old_tau` is the record's R_NEW_TAU (own leg), `o.tau_pd` the torque of the last substep
    static QS_FN void task_on_step(const qs_config& cfg, Task& t, const typename S::State& s, const typename S::Out& o, const V* old_tau, V now) {
        if (cfg.task == QS_TASK_NO_TASK) return;
        const V zero = V(0.0f), one = V(1.0f);
        M flying = qlt(T::quad_sum(o.foot_contact), V(0.5f));                     // quadruped.py:260-262
        M takeoff_v = qgt(s.vlin.z * (1.0f / 9.81f), V(0.06f));                   // task_base.py:157-160 (g = 9.81 here)
        t.switched = qsel(qand(qand(qlt(t.switched, V(0.5f)), flying), takeoff_v), one, t.switched);  // :152-155
        V d2 = zero;
#pragma unroll
        for (int j = 0; j < 3; j++) { V d = old_tau[j] - o.tau_pd[j]; d2 = d2 + d * d; }
        t.dtau2 = T::quad_sum(d2);                                                 // :68-70, :149-150
        t.pos[0] = s.pos.x; t.pos[1] = s.pos.y; t.pos[2] = s.pos.z; t.vel[0] = s.vlin.x; t.vel[1] = s.vlin.y; t.vel[2] = s.vlin.z;
        quat_to_rpy(s.qx, s.qy, s.qz, s.qw, t.rpy[0], t.rpy[1], t.rpy[2]);        // :72-75
        V z = t.pos[2];
        t.rel_max_h = qmax(t.rel_max_h, qmax(z - t.init_h, zero));                 // :81-86
        t.max_h = qmax(qabs(z), t.max_h);
        t.max_dx = qmax(qabs(t.pos[0]), t.max_dx);
        t.max_pitch = qmax(qabs(t.rpy[1]), t.max_pitch);                           // :88-90
        M air = qgt(t.all_air, V(0.5f));
        M take = qand(flying, qnot(air)), land = qand(qnot(flying), air);
        // at take-off: remember time / pose / yaw
        V t_takeoff0 = t.t_takeoff;
        t.t_takeoff = qsel(take, now, t.t_takeoff);
#pragma unroll
        for (int k = 0; k < 3; k++) t.pose_to[k] = qsel(take, t.pos[k], t.pose_to[k]);
        t.yaw_to = qsel(take, t.rpy[2], t.yaw_to);
        V dist = jump_distance(t);
        V fwd_upd = qmax(dist, t.max_fwd);
        V flight_upd = qmax(now - t_takeoff0, t.max_flight);
        if (continuous2(cfg.task)) {  // TaskContinuousJumping2, task_base.py:321-355
            const float jump_limit = cfg.task == QS_TASK_CONT_JUMPING_FORWARD_DEMO ? 0.5f : 0.6f;   // the base class's (task_base.py:286-290)
            const float height_limit = cfg.task == QS_TASK_CONT_JUMPING_FORWARD3 ? 0.45f : 0.5f;
            const float bound = cfg.task == QS_TASK_CONT_JUMPING_FORWARD3 ? 0.7f : 0.85f;
            M in_flight = qand(flying, air);
            t.is_jumping = qsel(take, qflag(takeoff_v), qsel(land, zero, t.is_jumping));
            t.max_jump_h = qsel(take, zero, qsel(in_flight, qmax(t.max_jump_h, z), t.max_jump_h));  // :327-332
            t.max_flight = qsel(land, flight_upd, t.max_flight);
            M count = qand(land, qlt(t.first_jump, V(0.5f)));                                          // :342, first jump ignored
            V fwd = qmin(dist, V(jump_limit)), hgt = qmin(t.max_jump_h, V(height_limit));
            V perf = fwd * (0.7f / jump_limit) + hgt * (0.3f / height_limit);
            V flogf = qsel(qgt(fwd, zero), fwd * qlog(qmax(fwd, V(1e-30f))) * 1.4426950408889634f, zero);
            t.jump_count = qsel(count, t.jump_count + 1.0f, t.jump_count);
            t.sum_fwd = qsel(count, t.sum_fwd + fwd, t.sum_fwd);
            t.sum_flogf = qsel(count, t.sum_flogf + flogf, t.sum_flogf);
            t.sum_height = qsel(count, t.sum_height + hgt, t.sum_height);
            t.sum_perf = qsel(count, t.sum_perf + perf, t.sum_perf);
            t.max_perf = qsel(count, qmax(t.max_perf, perf), t.max_perf);
            t.last_perf = qsel(count, perf, t.last_perf);
            t.good_jumps = qsel(qand(count, qge(perf, V(bound))), t.good_jumps + 1.0f, t.good_jumps);
            t.end_jump = qflag(count);
            t.first_jump = qsel(land, zero, t.first_jump);
        } else if (!continuous(cfg.task)) {  // task_base.py:92-106
            M in_flight = qand(flying, air);
            t.max_flight = qsel(land, flight_upd, t.max_flight);
            t.max_fwd = qsel(qor(in_flight, land), fwd_upd, qsel(qand(qnot(flying), qnot(air)), zero, t.max_fwd));
        } else {                      // task_base.py:244-280
            const float jump_limit = 0.5f, time_limit = cfg.task == QS_TASK_CONT_JUMPING_FORWARD ? 0.15f : 0.35f;
            t.is_jumping = qsel(take, qflag(takeoff_v), qsel(land, zero, t.is_jumping));
            t.max_flight = qsel(land, flight_upd, t.max_flight);
            t.max_fwd = qsel(land, fwd_upd, t.max_fwd);
            t.cum_fwd = qsel(land, t.cum_fwd + qmin(t.max_fwd, V(jump_limit)), t.cum_fwd);
            t.cum_ft = qsel(land, t.cum_ft + qmin(t.max_flight, V(time_limit)), t.cum_ft);
        }
        t.all_air = qsel(take, one, qsel(land, zero, t.all_air));
        if (cfg.task == QS_TASK_JUMPING_FORWARD_PPO || cfg.task == QS_TASK_JUMPING_FORWARD_PPO_HP) { t.old_fwd = t.actual_fwd; t.actual_fwd = t.max_fwd; }  // robot_tasks.py:418-425
        if (cfg.task == QS_TASK_BACKFLIP) t.bf_max_pitch = qmax(t.bf_max_pitch, pitch_backflip(s.qx, s.qy, s.qz, s.qw, t.switched));   // :527-530
        if (cfg.task == QS_TASK_BACKFLIP_PPO) t.max_pitch = qmax(t.max_pitch, pitch_backflip(s.qx, s.qy, s.qz, s.qw, t.switched));  // :752-754
    }
    static QS_FN V task_terminated(const qs_config& cfg, const Task& t, const typename S::State& s, V n_invalid, M demo_end) {
        if (cfg.task == QS_TASK_NO_TASK) return V(0.0f);
        M low = qlt(t.pos[2], V(cfg.fallen_height));                              // task_base.py:123-124
        M bad = qor(qgt(n_invalid, V(0.5f)), demo_end);                           // :137-147; demonstration used up: :213-214, 446-447
        if (cfg.task == QS_TASK_BACKFLIP || cfg.task == QS_TASK_BACKFLIP_DEMO) return qflag(qor(low, bad));   // robot_tasks.py:532-533, 239-241
        V d = s.qx * s.qx + s.qy * s.qy + s.qz * s.qz + s.qw * s.qw;
        V r22 = V(1.0f) - (s.qx * s.qx + s.qy * s.qy) * (V(2.0f) * qrcp(d));
        return qflag(qor(qand(qlt(r22, V(0.85f)), low), bad));                    // task_base.py:126-135
    }
    static QS_FN V task_reward(const qs_config& cfg, const Task& t, V contact_force, V bf_pitch) {
        if (cfg.task == QS_TASK_BACKFLIP_PPO) {  // robot_tasks.py:709-800
            V z = t.pos[2];
            V rew_h = qsel(qor(qlt(z, V(0.29f)), qgt(z, V(0.7f))), V(0.0f), z) * 0.026f;
            V rew_smooth = qexp(qsqrt(t.dtau2) * (-0.1f)) * 0.015f;
            V rew_contact = qsel(qgt(contact_force, V(800.0f)), contact_force, V(0.0f)) * (-3e-4f);
            V rew_pitch = qsel(qgt(z, V(0.5f)), bf_pitch, V(0.0f)) * 0.014f;
            return rew_contact * 0.4f + rew_smooth * 0.2f + rew_h * 0.25f + rew_pitch * 0.3f;
        }
        // ContinuousJumpingForwardPPO._reward is the constant 0 in the reference (bound-method test at robot_tasks.py:669)
        const bool ip = cfg.task == QS_TASK_JUMPING_IN_PLACE_PPO || cfg.task == QS_TASK_JUMPING_IN_PLACE_PPO_HP;
        const bool fw = cfg.task == QS_TASK_JUMPING_FORWARD_PPO || cfg.task == QS_TASK_JUMPING_FORWARD_PPO_HP;
        if (!ip && !fw) return V(0.0f);
        // robot_tasks.py:258-344 / :369-472
        const float max_h = ip ? (cfg.task == QS_TASK_JUMPING_IN_PLACE_PPO ? 1.0f : 1.25f) : (cfg.task == QS_TASK_JUMPING_FORWARD_PPO ? 0.9f : 1.1f);
        const float k_h = ip ? 0.023f : 0.026f;
        V z = t.pos[2];
        V rew_h = qsel(qor(qlt(z, V(0.29f)), qgt(z, V(max_h))), V(0.0f), z) * k_h;
        V rew_smooth = qexp(qsqrt(t.dtau2) * (-0.1f)) * 0.015f;
        V rew_contact = qsel(qgt(contact_force, V(800.0f)), contact_force, V(0.0f)) * (-3e-4f);
        V rew_pitch = qexp(qabs(t.rpy[1]) * (-26.0f)) * 0.014f;
        if (ip) {
            V rew_pos = qexp(qabs(t.pos[0]) * (-40.0f)) * 0.013f;
            return rew_pos * 0.05f + rew_contact * 0.5f + rew_smooth * 0.2f + rew_h * 0.45f + rew_pitch * 0.3f;
        }
        const float max_fwd = cfg.task == QS_TASK_JUMPING_FORWARD_PPO ? 1.3f : 1.4f;
        V fwd = t.actual_fwd;
        M drop = qor(qgt(fwd, V(max_fwd)), qand(qle(fwd, t.old_fwd), qge(fwd, t.old_fwd)));
        V rew_fwd = qsel(drop, V(0.0f), fwd) * 0.038f;
        return rew_contact * 0.4f + rew_smooth * 0.2f + rew_h * 0.25f + rew_pitch * 0.3f + rew_fwd * 0.4f;
    }
    static QS_FN V task_reward_end(const qs_config& cfg, const Task& t, V term, V now) {
        const V zero = V(0.0f), one = V(1.0f);
        M alive = qlt(term, V(0.5f));
        V g = qexp(-t.max_pitch * t.max_pitch * (1.0f / (0.15f * 0.15f)));
        switch (cfg.task) {
        case QS_TASK_JUMPING_IN_PLACE: {  // robot_tasks.py:31-57
            V hn = qsel(qgt(t.rel_max_h, V(0.9f)), one, t.rel_max_h * (1.0f / 0.9f));
            V r = hn * 0.7f + hn * 0.3f * g + hn * 0.05f * qexp(-t.max_dx * t.max_dx * (1.0f / 0.05f));
            return r + qsel(alive, hn * 0.1f, -(one + hn * 0.8f) * 0.08f);
        }
        case QS_TASK_JUMPING_FORWARD: {   // :70-99
            V hn = qsel(qgt(t.rel_max_h, V(0.3f)), one, t.rel_max_h * (1.0f / 0.3f));
            V fn = qsel(qgt(t.max_fwd, V(1.3f)), one, t.max_fwd * (1.0f / 1.3f));
            V bm = (hn + fn) * 0.5f;
            V r = hn * 0.25f + fn * hn * 0.5f + hn * 0.25f * g;
            return r + qsel(alive, bm * 0.1f, -(one + bm * 1.2f) * 0.08f);
        }
        case QS_TASK_CONT_JUMPING_FORWARD: {  // :112-131
            V tn = t.cum_ft * (1.0f / 0.15f), dn = t.cum_fwd * (1.0f / 0.5f), bm = (tn + dn) * 0.5f;
            return tn * 0.25f + dn * 0.5f + tn * 0.25f * g + qsel(alive, bm * 0.1f, zero);
        }
        case QS_TASK_CONT_JUMPING_FORWARD2: { // :144-165
            V tn = qmin(t.max_flight, V(0.35f)) * (1.0f / 0.35f), dn = qmin(t.max_fwd, V(0.5f)) * (1.0f / 0.5f), bm = (tn + dn) * 0.5f;
            return tn * 0.25f + dn * 0.5f + dn * 0.15f * g + (now * 0.1f) * bm * 0.4f + qsel(alive, bm * 0.2f, zero);
        }
        case QS_TASK_JUMPING_IN_PLACE_PPO: case QS_TASK_JUMPING_IN_PLACE_PPO_HP:  // :348-358
            return qsel(alive, zero, -t.max_h * 0.25f);
        case QS_TASK_JUMPING_FORWARD_PPO: case QS_TASK_JUMPING_FORWARD_PPO_HP:    // :475-485
            return qsel(alive, (t.max_fwd + t.max_h) * 0.025f, zero);
        case QS_TASK_BACKFLIP: {              // :535-550
            V h = clampv<V>(t.max_h - 0.3f, zero, V(0.4f)) * (1.0f / 0.4f);
            V pm = t.bf_max_pitch * (1.0f / (2.0f * PI));
            return pm * 0.4f + h * 0.4f + h * pm + qsel(qand(qgt(t.switched, V(0.5f)), alive), V(0.2f), zero);
        }
        case QS_TASK_BACKFLIP_PPO:            // :802-809
            return qsel(alive, (t.max_pitch * (0.7f / 5.0f) + t.max_h * 0.3f) * 0.1f, zero);
        case QS_TASK_CONT_JUMPING_FORWARD3: { // :181-212
            V n = qmax(t.jump_count, V(3.0f));
            V avg = t.sum_perf * qrcp(n), mx = qmax(t.max_perf, zero);
            V rew_entropy = qexp((cj2_entropy(t) - 1.0f) * (1.0f / 0.3f));
            V ra = avg * 0.15f * g + avg * 0.4f * (now * 0.1f) + avg * rew_entropy * 0.2f + avg * 0.25f;
            return ra * 0.8f + mx * 0.2f + t.good_jumps * 0.1f + qsel(alive, avg * 0.2f, zero);
        }
        case QS_TASK_CONT_JUMPING_FORWARD_PPO: {  // :686-698
            V n = qmax(t.jump_count, V(3.0f));
            V r = t.sum_perf * qrcp(n) * qexp((cj2_entropy(t) - 1.0f) * (1.0f / 0.3f));
            return qsel(alive, r, r - 1.0f);
        }
        default: return zero;
        }
    }
    static QS_FN void task_reset(const qs_config& cfg, Task& t, const typename S::State& s, const typename S::Out& o, V now) {  // task_base.py:40-59
        const V zero = V(0.0f);
        V keep = t.bf_max_pitch;  // BackFlip.max_pitch is initialised in __init__ only (robot_tasks.py:524)
        t.switched = zero; t.all_air = zero; t.is_jumping = zero; t.t_takeoff = now;
        t.pose_to[0] = s.pos.x; t.pose_to[1] = s.pos.y; t.pose_to[2] = s.pos.z; t.init_h = s.pos.z;
        V r, p, y; quat_to_rpy(s.qx, s.qy, s.qz, s.qw, r, p, y); t.yaw_to = y;
        t.max_flight = zero; t.max_fwd = zero; t.max_pitch = zero; t.rel_max_h = zero; t.max_dx = zero; t.max_h = zero;
        t.cum_fwd = zero; t.cum_ft = zero; t.old_fwd = zero; t.actual_fwd = zero; t.bf_max_pitch = keep;
        t.jump_count = zero; t.good_jumps = zero; t.sum_fwd = zero; t.sum_flogf = zero; t.sum_height = zero; t.sum_perf = zero;
        t.max_perf = zero; t.last_perf = zero; t.max_jump_h = zero; t.first_jump = V(1.0f); t.end_jump = zero;
        task_on_step(cfg, t, s, o, o.tau_pd, now);  // old == new torque at reset
    }

    // ---- sensors (sensors/robot_sensors.py, sensor.py:46-60) into obs[QS_MAX_OBS] of this environment
    // `rpy`: roll, pitch, yaw of s if the caller has them already (task_on_step computes them for every task but NO_TASK)
    static QS_FN void write_obs(const qs_config& cfg, float* obs, const typename S::State& s, const typename S::Out& o, const Task& t,
                                uint32_t env_id, uint32_t total_steps, const V* rpy = nullptr) {
        int n = 0;
        V roll, pitch, yaw;
        if (rpy) { roll = rpy[0]; pitch = rpy[1]; yaw = rpy[2]; }
        else quat_to_rpy(s.qx, s.qy, s.qz, s.qw, roll, pitch, yaw);
        for (int si = 0; si < cfg.n_sensors; si++) {
            switch (cfg.sensors[si]) {
            case QS_SENS_JOINT_POS: for (int j = 0; j < 3; j++) T::st_leg(obs, n + j, 3, s.q[j]); n += 12; break;
            case QS_SENS_JOINT_VEL: for (int j = 0; j < 3; j++) T::st_leg(obs, n + j, 3, s.qd[j]); n += 12; break;
            case QS_SENS_PITCH: T::st(obs, n++, pitch); break;
            case QS_SENS_HEIGHT: T::st(obs, n++, s.pos.z); break;
            case QS_SENS_VEL_Z: T::st(obs, n++, s.vlin.z); break;
            case QS_SENS_VEL_X: T::st(obs, n++, s.vlin.x); break;
            case QS_SENS_LANDING: T::st(obs, n++, t.switched); break;
            case QS_SENS_JUMPING: T::st(obs, n++, t.is_jumping); break;
            case QS_SENS_PITCH_RATE: {  // quadruped.py:141-170: (R^T w_world)[1]
                V d = s.qx * s.qx + s.qy * s.qy + s.qz * s.qz + s.qw * s.qw, sc = V(2.0f) * qrcp(d);
                V r01 = (s.qx * s.qy - s.qw * s.qz) * sc, r11 = V(1.0f) - (s.qx * s.qx + s.qz * s.qz) * sc, r21 = (s.qy * s.qz + s.qw * s.qx) * sc;
                T::st(obs, n++, r01 * s.vang.x + r11 * s.vang.y + r21 * s.vang.z);
                break;
            }
            case QS_SENS_BOOL_CONTACT: T::st_leg(obs, n, 1, o.foot_contact); n += 4; break;
            case QS_SENS_LIN_VEL: T::st(obs, n, s.vlin.x); T::st(obs, n + 1, s.vlin.y); T::st(obs, n + 2, s.vlin.z); n += 3; break;
            case QS_SENS_ANG_VEL: T::st(obs, n, s.vang.x); T::st(obs, n + 1, s.vang.y); T::st(obs, n + 2, s.vang.z); n += 3; break;
            case QS_SENS_RPY: T::st(obs, n, roll); T::st(obs, n + 1, pitch); T::st(obs, n + 2, yaw); n += 3; break;
            case QS_SENS_QUAT: T::st(obs, n, s.qx); T::st(obs, n + 1, s.qy); T::st(obs, n + 2, s.qz); T::st(obs, n + 3, s.qw); n += 4; break;
            case QS_SENS_FEET_POS: case QS_SENS_FEET_VEL: {  // quadruped.py:440-449
                V J[9], p[3]; leg_fk(cfg, s.q, J, p);
                for (int i = 0; i < 3; i++) {
                    V v = cfg.sensors[si] == QS_SENS_FEET_POS ? p[i] : J[3 * i] * s.qd[0] + J[3 * i + 1] * s.qd[1] + J[3 * i + 2] * s.qd[2];
                    T::st_leg(obs, n + i, 3, v);
                }
                n += 12; break;
            }
            case QS_SENS_PITCH_BACKFLIP: T::st(obs, n++, pitch_backflip(s.qx, s.qy, s.qz, s.qw, t.switched)); break;
            default: break;
            }
        }
        T::sync();  // the noise pass reads observation entries written by other lanes of the quad
        if (cfg.noise_enabled) {  // sensor.py:25-32, 46-52: i.i.d. N(0, sigma) resampled at every read
            for (int blk0 = 0; blk0 * 4 < cfg.obs_dim; blk0 += 4) {
                V z[4]; Rng<T>::normal4(cfg.seed, env_id, 0u, total_steps, (uint32_t)blk0, z);
#pragma unroll
                for (int k = 0; k < 4; k++) {  // element index = 4 * (blk0 + leg) + k ; sigma is zero-padded beyond obs_dim
                    V sd = T::ld_leg(cfg.obs_noise_std, 4 * blk0 + k, 4);
                    V cur = T::ld_leg(obs, 4 * blk0 + k, 4);
                    T::st_leg(obs, 4 * blk0 + k, 4, qsel(qgt(sd, V(0.0f)), cur + sd * z[k], cur));
                }
            }
        }
    }

    // ---- Butterworth action filter (action_filter.py:110-121) in a form whose DC gain is exactly 1 in float32:
    //      y = y1 + a2 (y1 - y2) + b0 (x - y1) + b1 (x1 - y1) + b2 (x2 - y1)     [uses 1 + a1 + a2 = b0 + b1 + b2]
    static QS_FN V filter(const qs_config& cfg, V x, V x1, V x2, V y1, V y2) {
        const float a2 = (float)cfg.filt_a[2], b0 = (float)cfg.filt_b[0], b1 = (float)cfg.filt_b[1], b2 = (float)cfg.filt_b[2];
        return y1 + (y1 - y2) * a2 + (x - y1) * b0 + (x1 - y1) * b1 + (x2 - y1) * b2;
    }

    // what distinguishes the reference's landing wrappers (env/wrappers/landing_wrapper*.py)
    struct WrapTraits { bool landing_family, trigger_jumping, pitch_takeoff, gains; int exit_kind; bool one_shot; };
    static QS_FN WrapTraits wrap_traits(int mode) {
        switch (mode) {
        case QS_WRAP_LANDING: return {true, false, false, true, 0, false};
        case QS_WRAP_LANDING2: return {true, false, false, false, 1, true};
        case QS_WRAP_LANDING_BACKFLIP: return {true, false, true, false, 0, false};
        case QS_WRAP_LANDING_BACKFLIP2: return {true, false, true, false, 1, true};
        case QS_WRAP_LANDING_CONTINUOUS: return {true, true, false, false, 2, false};
        default: return {false, false, false, false, 0, false};
        }
    }
    struct StepOut { V reward, done, trunc; int resume; };   // resume >= 0 (HOT builds, wave-uniform): go on in the full build at that substep (see step)

    // One env.step(action) (gym_env.py:227-256).  `rec` = this environment's record, `act` = its action row,
    // `obs` = QS_MAX_OBS floats of staging for its observation.
    // `settle_n` > 0 turns the call into a slice of a reset's settle (gym_env.py:325-327) for a record of the streaming reset
    // pool: that many substeps under the settling command and nothing else.  It goes through the SAME substep loop as a
    // regular step, so a wave of settling records costs what a wave of environments costs and shares its instructions.
    // `trace` (null except in the lanes of a traced environment; `any_trace` is the wave-uniform "some lane has one") receives
    // one row per substep (qs_set_trace).
    static QS_FN void write_trace(float* row, float time, const typename S::State& s, const typename S::Out& o) {
        T::st(row, TR_TIME, V(time));
        T::st(row, TR_POS, s.pos.x); T::st(row, TR_POS + 1, s.pos.y); T::st(row, TR_POS + 2, s.pos.z);
        T::st(row, TR_QUAT, s.qx); T::st(row, TR_QUAT + 1, s.qy); T::st(row, TR_QUAT + 2, s.qz); T::st(row, TR_QUAT + 3, s.qw);
        T::st(row, TR_VLIN, s.vlin.x); T::st(row, TR_VLIN + 1, s.vlin.y); T::st(row, TR_VLIN + 2, s.vlin.z);
        T::st(row, TR_VANG, s.vang.x); T::st(row, TR_VANG + 1, s.vang.y); T::st(row, TR_VANG + 2, s.vang.z);
#pragma unroll
        for (int j = 0; j < 3; j++) {
            T::st_leg(row, TR_Q + j, 3, s.q[j]); T::st_leg(row, TR_QD + j, 3, s.qd[j]);
            T::st_leg(row, TR_TAU + j, 3, o.tau_pd[j]); T::st_leg(row, TR_TAU_SPRING + j, 3, o.tau_spring[j]);
        }
        T::st_leg(row, TR_FOOT_FORCE, 1, o.foot_force); T::st_leg(row, TR_FOOT_CONTACT, 1, o.foot_contact);
    }
    // A HOT build's step hands over where a wave needs a rare path: substep k of the common-path build gives up, the step leaves the
    // environment's state of that moment in the record, what the prologue produced in the (still unused) observation row -- ST_* below --
    // and returns resume = k; the kernel then calls the FULL build's step<true>(..., k), which skips the prologue (the filter history and
    // the scripted phases have moved on already), picks those values up and runs substeps k .. n - 1 and the epilogue.  The environments
    // of the wave that have no rare row of their own get from the full build, bit for bit, what the common-path build gives them
    // (qs_core.h), so nothing depends on where the switch happens.  (Rounds 2-3 fetched the records again and repeated the WHOLE env step in
    // the full build: the wave that every launch waits for paid up to two steps' time.  Continuing inside the same function -- the full
    // build's substep loop behind the hot one -- was tried first: the values live across both loops put ~30 scratch instructions into the
    // hot loop, 112.1 -> 102.5 M env-steps/s.)
    enum { RESUME_AT_BOUNDARY = 0x100 };
    enum { ST_CMD = 0, ST_KP = 12, ST_KD = 15, ST_W = 18, ST_CPGP = 22, ST_CPGR = 28, ST_CPGTH = 32 };
    static QS_FN void stash_state(float* rec, const typename S::State& s) {
        T::st(rec, R_POS, s.pos.x); T::st(rec, R_POS + 1, s.pos.y); T::st(rec, R_POS + 2, s.pos.z);
        T::st(rec, R_QUAT, s.qx); T::st(rec, R_QUAT + 1, s.qy); T::st(rec, R_QUAT + 2, s.qz); T::st(rec, R_QUAT + 3, s.qw);
        T::st(rec, R_VLIN, s.vlin.x); T::st(rec, R_VLIN + 1, s.vlin.y); T::st(rec, R_VLIN + 2, s.vlin.z);
        T::st(rec, R_VANG, s.vang.x); T::st(rec, R_VANG + 1, s.vang.y); T::st(rec, R_VANG + 2, s.vang.z);
#pragma unroll
        for (int j = 0; j < 3; j++) { T::st_leg(rec, R_Q + j, 3, s.q[j]); T::st_leg(rec, R_QD + j, 3, s.qd[j]); }
        T::st_leg(rec, R_WARM, 1, s.warm);
    }
    // LEAN (HOT builds under a 256-register budget: k_step_dense): the substep loop keeps the per-environment parameters in LDS instead
    // of 36 registers and fetches them at the top of every substep -- the raw parameters from the record, the scripted gains from the
    // stash, the base's inertia (what build_base derives from the masses) from 11 more floats of the observation row.  Same values, same
    // results; nine ds_read_b128 per substep against ~65 scratch instructions in the loop.
    enum { ST_I0 = 40 };
    static QS_FN void load_par_lean(const float* rec, const float* obs, typename S::Par& P) {
        const float* p = rec + R_PARAMS;
        P.mu = T::ld(p, P_MU);
#pragma unroll
        for (int j = 0; j < 3; j++) {
            P.k[j] = T::ld(p, P_K + j); P.b[j] = T::ld(p, P_B + j); P.rest[j] = T::ld(p, P_REST + j);
            P.kp[j] = T::ld(obs, ST_KP + j); P.kd[j] = T::ld(obs, ST_KD + j); P.m_leg[j] = T::ld(p, P_M_LEG + j);
        }
        P.m_pay = T::ld(p, P_M_PAY); P.r_pay = mk3<V>(T::ld(p, P_R_PAY), T::ld(p, P_R_PAY + 1), T::ld(p, P_R_PAY + 2));
        P.I0.m = T::ld(obs, ST_I0); P.I0.h = mk3<V>(T::ld(obs, ST_I0 + 1), T::ld(obs, ST_I0 + 2), T::ld(obs, ST_I0 + 3));
        P.I0.I.xx = T::ld(obs, ST_I0 + 4); P.I0.I.xy = T::ld(obs, ST_I0 + 5); P.I0.I.xz = T::ld(obs, ST_I0 + 6);
        P.I0.I.yy = T::ld(obs, ST_I0 + 7); P.I0.I.yz = T::ld(obs, ST_I0 + 8); P.I0.I.zz = T::ld(obs, ST_I0 + 9);
        P.mtot = T::ld(obs, ST_I0 + 10);
    }
    template <bool RESUME = false, bool LEAN = false>
    static QS_FN StepOut step(const qs_config& cfg, float* rec, const float* act_row, float* obs, uint32_t env_id, int settle_n = 0,
                              float* trace = nullptr, bool any_trace = false, const float* demo_rows = nullptr, int demo_len = 0, int resume_k = 0) {
        static_assert(!(RESUME && HOT), "an env step resumes in the full build");
        typename S::State s; typename S::Par P; typename S::Out o;
        if (RESUME) T::sync();   // (the stash was written by other lanes of the quad)
        load_state(rec, s); load_par(cfg, rec, P);
        const int d = settle_n > 0 ? 0 : cfg.action_dim;
        // action: copy, filter (gym_env.py:229-234); every lane keeps the d replicated values plus its own-leg slice
        // raw action: d == 12 -> every lane holds the 3 entries of its own leg in act[12..14];
        //             d  < 12 -> the d values are replicated over the quad in act[0..d)
        // (every loop over these arrays runs to a constant bound under `#pragma unroll` with the runtime d as a guard: a runtime trip count
        // would put the arrays into scratch memory -- ~120 dword stores per environment and step, measured as WRITE_SIZE)
        // (a RESUMED step fetches what only its epilogue needs -- the action as given, _last_action, the previous torque, the counters --
        // behind the substep loop: the full build's loop is short of registers as it is)
        V act[15], act_in[15];
#pragma unroll
        for (int k = 0; k < 15; k++) act[k] = V(0.0f);
#define QS_LOAD_ACTION_ROW                                                                                             \
        if (d == 12) {                                                                                                 \
            _Pragma("unroll") for (int j = 0; j < 3; j++) act[12 + j] = T::ld_leg(act_row, j, 3);                      \
        } else {                                                                                                       \
            _Pragma("unroll") for (int k = 0; k < 12; k++)                                                             \
                if (k < d) act[k] = T::ld(act_row, k);                                                                 \
        }                                                                                                              \
        _Pragma("unroll") for (int k = 0; k < 15; k++) act_in[k] = act[k];
        if (!RESUME) { QS_LOAD_ACTION_ROW }
        // scripted phases of the landing / go-to-rest wrappers (one inner env.step per call)
        V w_phase = V(0.0f), w_timer = V(0.0f), w_end = V(0.0f), w_tstart = V(0.0f);
        if (!RESUME && cfg.wrapper_mode != QS_WRAP_NONE && settle_n == 0) {
            const float* w = rec + R_WRAP;
            w_phase = T::ld(w, W_PHASE); w_timer = T::ld(w, W_TIMER); w_end = T::ld(w, W_END); w_tstart = T::ld(w, W_TSTART);
            const float env_dt = (float)((double)cfg.action_repeat * cfg.dt);
            V now0 = V((float)((double)f2i(rec[R_SIM_STEP]) * cfg.dt));
            M scripted_gain = qlt(V(1.0f), V(0.0f));
            V gkp = V(0.0f), gkd = V(0.0f);
            const WrapTraits wt = wrap_traits(cfg.wrapper_mode);
            if (wt.landing_family) {                        // landing_wrapper*.py
                M in_to = qand(qgt(w_phase, V(0.5f)), qlt(w_phase, V(1.5f)));
                M hold = in_to;
                if (!wt.pitch_takeoff) {                    // landing_wrapper.py:47-54 + utils/timer.py: hold the last action until the timer is up
                    M up = qand(in_to, qgt(w_timer, w_end));
                    w_phase = qsel(up, V(2.0f), w_phase);
                    hold = qand(in_to, qnot(up));
                    w_timer = qsel(hold, w_timer + env_dt, w_timer);
                }
                M land = qgt(w_phase, V(1.5f));
#pragma unroll
                for (int k = 0; k < 15; k++) {
                    if ((d == 12) != (k >= 12)) continue;
                    V held = d == 12 ? T::ld_leg(w, W_ACTION + (k - 12), 3) : T::ld(w, W_ACTION + k);
                    if (wt.pitch_takeoff) held = V(k % 3 == 0 ? 0.0f : (k % 3 == 1 ? 1.0f : -1.0f));   // landing_wrapper_backflip.py:22
                    V la = d == 12 ? T::ld_leg(cfg.landing_action, k - 12, 3) : V(cfg.landing_action[k < 12 ? k : 0]);
                    act[k] = qsel(land, la, qsel(hold, held, act[k]));
                }
                if (wt.gains) { scripted_gain = land; gkp = V(cfg.landing_kp); gkd = V(cfg.landing_kd); }   // only landing_wrapper.py:39
            } else {                                          // go_to_rest_wrapper.py:59-80, interface_base.py:111-119
                M rest = qgt(w_phase, V(2.5f));
                V t1 = w_tstart + cfg.rest_time;
                V frac = clampv<V>((now0 - w_tstart) * qrcp(V(cfg.rest_time)), V(0.0f), V(1.0f));
#pragma unroll
                for (int k = 0; k < 15; k++) {
                    if ((d == 12) != (k >= 12)) continue;
                    V u0 = d == 12 ? T::ld_leg(w, W_ACTION + (k - 12), 3) : T::ld(w, W_ACTION + k);
                    V u1 = d == 12 ? T::ld_leg(cfg.settle_action, k - 12, 3) : V(cfg.settle_action[k < 12 ? k : 0]);
                    V u = qsel(qlt(now0, w_tstart), u0, qsel(qgt(now0, t1), u1, u0 + (u1 - u0) * frac));
                    act[k] = qsel(rest, u, act[k]);
                }
                scripted_gain = rest; gkp = V(cfg.rest_kp); gkd = V(cfg.rest_kd);
            }
#pragma unroll
            for (int j = 0; j < 3; j++) { P.kp[j] = qsel(scripted_gain, gkp, P.kp[j]); P.kd[j] = qsel(scripted_gain, gkd, P.kd[j]); }
            T::st(rec, R_WRAP + W_SCRIPTED, qflag(qgt(w_phase, V(0.5f))));
        }
        V act_last[15];   // what env.step was given (after the scripted phases): _last_action, gym_env.py:230
#pragma unroll
        for (int k = 0; k < 15; k++) act_last[k] = act[k];
        if (RESUME) {   // what env.step was given is in the record already (R_LAST_ACTION), the filter has run: fetched behind the loop
        } else if (d == 12) {  // DEFAULT space / raw commands: every lane filters the 3 entries of its own leg
#pragma unroll
            for (int j = 0; j < 3; j++) {
                V a = act[12 + j];
                T::st_leg(rec, R_LAST_ACTION + j, 3, a);
                if (cfg.enable_filter) {
                    V x1 = T::ld_leg(rec, R_XHIST + j, 3), x2 = T::ld_leg(rec, R_XHIST + 12 + j, 3), y1 = T::ld_leg(rec, R_YHIST + j, 3), y2 = T::ld_leg(rec, R_YHIST + 12 + j, 3);
                    V y = filter(cfg, a, x1, x2, y1, y2);
                    T::st_leg(rec, R_XHIST + 12 + j, 3, x1); T::st_leg(rec, R_XHIST + j, 3, a); T::st_leg(rec, R_YHIST + 12 + j, 3, y1); T::st_leg(rec, R_YHIST + j, 3, y);
                    a = y;
                }
                act[12 + j] = a;
            }
        } else {        // SYMMETRIC (6) / SYMMETRIC_NO_HIP (4) / CPG (5): the few values are replicated over the quad
#pragma unroll
            for (int k = 0; k < 12; k++) {
                if (k >= d) continue;
                V a = act[k];
                T::st(rec, R_LAST_ACTION + k, a);
                if (cfg.enable_filter) {
                    V x1 = T::ld(rec, R_XHIST + k), x2 = T::ld(rec, R_XHIST + 12 + k), y1 = T::ld(rec, R_YHIST + k), y2 = T::ld(rec, R_YHIST + 12 + k);
                    V y = filter(cfg, a, x1, x2, y1, y2);
                    T::st(rec, R_XHIST + 12 + k, x1); T::st(rec, R_XHIST + k, a); T::st(rec, R_YHIST + 12 + k, y1); T::st(rec, R_YHIST + k, y);
                    a = y;
                }
                act[k] = a;
            }
        }
        // _interpolate_actions (gym_env.py:187-205) is the identity in the reference (both "last" actions are overwritten
        // with the current one at :230/:234 before the substeps run), so there is nothing to do for enable_interp.
        V cmd[3];
        const bool cpg = cfg.action_space_mode == QS_ACT_CPG && settle_n == 0;
        V cpg_p[5], cpg_r = V(0.0f), cpg_th = V(0.0f);
        if (RESUME) {
#pragma unroll
            for (int j = 0; j < 3; j++) { cmd[j] = T::ld_leg(obs, ST_CMD + j, 3); P.kp[j] = T::ld(obs, ST_KP + j); P.kd[j] = T::ld(obs, ST_KD + j); }
            w_phase = T::ld(obs, ST_W); w_timer = T::ld(obs, ST_W + 1); w_end = T::ld(obs, ST_W + 2); w_tstart = T::ld(obs, ST_W + 3);
#pragma unroll
            for (int k = 0; k < 5; k++) cpg_p[k] = T::ld(obs, ST_CPGP + k);
            cpg_r = T::ld_leg(obs, ST_CPGR, 1); cpg_th = T::ld_leg(obs, ST_CPGTH, 1);
            T::sync();   // (the row is the wave's scratch from here on, until the epilogue writes the observation into it)
        } else if (settle_n > 0) {
#pragma unroll
            for (int k = 0; k < 5; k++) cpg_p[k] = V(0.0f);
#pragma unroll
            for (int j = 0; j < 3; j++) cmd[j] = T::ld_leg(cfg.settle_cmd, j, 3);
        } else if (cpg) {
#pragma unroll
            for (int k = 0; k < 5; k++) cpg_p[k] = clampv<V>(act[k], V(-1.0f), V(1.0f)) * (0.5f * (cfg.cpg_hi[k] - cfg.cpg_lo[k])) + 0.5f * (cfg.cpg_hi[k] + cfg.cpg_lo[k]);
            cpg_r = T::ld_leg(rec, R_CPG, 1); cpg_th = T::ld_leg(rec, R_CPG + 4, 1);
            cmd[0] = V(0.0f); cmd[1] = V(0.0f); cmd[2] = V(0.0f);   // (cpg_command sets them in every substep)
        } else {
#pragma unroll
            for (int k = 0; k < 5; k++) cpg_p[k] = V(0.0f);
            action_to_command(cfg, act, cmd);
        }
        int sim_step = 0, env_step = 0, total = 0;
        V old_tau[3];
#define QS_LOAD_COUNTERS                                                                                               \
        sim_step = f2i(rec[R_SIM_STEP]); env_step = f2i(rec[R_ENV_STEP]); total = f2i(rec[R_TOTAL_STEPS]);             \
        _Pragma("unroll") for (int j = 0; j < 3; j++) old_tau[j] = T::ld_leg(rec, R_NEW_TAU + j, 3);
        if (!RESUME) { QS_LOAD_COUNTERS }
        const int n_sub = settle_n > 0 ? settle_n : cfg.action_repeat;
        float* const blk = cfg.payload_soft ? rec + R_BLOCK : nullptr;
        if (HOT) {
            // what a resumed step needs of the prologue goes into the (still unused) observation row here, once per step and whether or not
            // the step is ever handed over: 30 LDS writes -- stored only at the hand-over, these values stayed live for that cold block all
            // through the hot loop (0.8 % of the headline)
#pragma unroll
            for (int j = 0; j < 3; j++) { T::st_leg(obs, ST_CMD + j, 3, cmd[j]); T::st(obs, ST_KP + j, P.kp[j]); T::st(obs, ST_KD + j, P.kd[j]); }
            T::st(obs, ST_W, w_phase); T::st(obs, ST_W + 1, w_timer); T::st(obs, ST_W + 2, w_end); T::st(obs, ST_W + 3, w_tstart);
#pragma unroll
            for (int i = 0; i < 5; i++) T::st(obs, ST_CPGP + i, cpg_p[i]);
            if (LEAN) {
                T::st(obs, ST_I0, P.I0.m); T::st(obs, ST_I0 + 1, P.I0.h.x); T::st(obs, ST_I0 + 2, P.I0.h.y); T::st(obs, ST_I0 + 3, P.I0.h.z);
                T::st(obs, ST_I0 + 4, P.I0.I.xx); T::st(obs, ST_I0 + 5, P.I0.I.xy); T::st(obs, ST_I0 + 6, P.I0.I.xz);
                T::st(obs, ST_I0 + 7, P.I0.I.yy); T::st(obs, ST_I0 + 8, P.I0.I.yz); T::st(obs, ST_I0 + 9, P.I0.I.zz); T::st(obs, ST_I0 + 10, P.mtot);
                T::sync();
            }
        }
        static_assert(!LEAN || HOT, "the lean loop keeps its parameters in the observation row, which the full build's many-rows solve borrows");
        // resume_k: the substep a handed-over step goes on at; + RESUME_AT_BOUNDARY when the common-path build finished the substep before
        // and handed over between two substeps (the Hopf oscillators have not ticked for this one yet)
        const bool resume_ticked = !(resume_k & RESUME_AT_BOUNDARY);
        resume_k &= RESUME_AT_BOUNDARY - 1;
        int k = RESUME ? resume_k : 0;
        bool gave_up = false;   // (HOT builds, wave-uniform) substep k needs a rare path; `s` is as that substep found it
        bool at_boundary = false;   // ... and that was seen coming one substep ahead: `s` is as substep k - 1 left it, nothing of substep k has run
        for (; k < n_sub; k++) {  // gym_env.py:236-237, 207-216
            V tau[3];
            if (cpg && !(RESUME && k == resume_k && resume_ticked)) cpg_command(cfg, cpg_p, cpg_r, cpg_th, cmd);  // the oscillators tick at the physics rate (the tick of a substep that gave up half-way has happened)
            QS_PHASE_SUB_BEGIN
            if (LEAN) load_par_lean(rec, obs, P);
            S::actuate(cfg, P, s, cmd, o, tau, settle_n > 0);
            const int rc = S::substep(cfg, P, s, tau, o, k == n_sub - 1 || cfg.body_contacts, blk, obs, k == n_sub - 1);
            if (__builtin_expect(rc == 1, 0)) { gave_up = true; break; }
            QS_PHASE_SUB(k)
            if (__builtin_expect(any_trace, 0)) {
                if (trace) write_trace(trace + k * QS_TRACE_DIM, (float)((double)(f2i(rec[R_SIM_STEP]) + k + 1) * cfg.dt), s, o);
            }
            // (round 6) the next substep is going to need the rare path: hand over HERE, at the substep boundary, instead of running its
            // dynamics up to the vote for nothing
            if (HOT && __builtin_expect(rc == 2, 0) && k + 1 < n_sub) { k++; gave_up = true; at_boundary = true; break; }
        }
        // (Tried instead, round 4: the state checkpointed into the LDS record after every substep -- eleven LDS stores -- so that the hand-over
        // needs nothing kept: ~70 register moves fewer per substep in the ISA and 0.9 % SLOWER on the GPU, 109.5 against 110.5 M; the state
        // stored inside the substep, at each of the three votes that give up: 109.9 against 110.7 M.)
        if (HOT && __builtin_expect(gave_up, 0)) {   // hand over to the full build: the state of this moment (the rest of the stash is in place)
            stash_state(rec, s);
            if (cpg) { T::st_leg(obs, ST_CMD, 3, cmd[0]); T::st_leg(obs, ST_CMD + 1, 3, cmd[1]); T::st_leg(obs, ST_CMD + 2, 3, cmd[2]); T::st_leg(obs, ST_CPGR, 1, cpg_r); T::st_leg(obs, ST_CPGTH, 1, cpg_th); }
            StepOut z; z.reward = V(0.0f); z.done = V(0.0f); z.trunc = V(0.0f); z.resume = k | (at_boundary ? (int)RESUME_AT_BOUNDARY : 0);
            return z;
        }
        if (settle_n > 0) { store_state(rec, s, o); StepOut z; z.reward = V(0.0f); z.done = V(0.0f); z.trunc = V(0.0f); z.resume = -1; return z; }
        if (RESUME) {
            T::sync();
            QS_LOAD_ACTION_ROW
            QS_LOAD_COUNTERS
#pragma unroll
            for (int k = 0; k < 15; k++) act_last[k] = V(0.0f);
            if (d == 12) {
#pragma unroll
                for (int j = 0; j < 3; j++) act_last[12 + j] = T::ld_leg(rec, R_LAST_ACTION + j, 3);
            } else {
#pragma unroll
                for (int k = 0; k < 12; k++)
                    if (k < d) act_last[k] = T::ld(rec, R_LAST_ACTION + k);
            }
        }
#undef QS_LOAD_ACTION_ROW
#undef QS_LOAD_COUNTERS
        QS_PHASE(32)
        if (cpg) { T::st_leg(rec, R_CPG, 1, cpg_r); T::st_leg(rec, R_CPG + 4, 1, cpg_th); }
        sim_step += cfg.action_repeat; env_step += 1; total += 1;
        Task t;
        load_task(rec, t);
        V now = V((float)((double)sim_step * cfg.dt));
        QS_PHASE(33)
        task_on_step(cfg, t, s, o, old_tau, now);
        QS_PHASE(34)
        V force = T::quad_sum(o.foot_force);
        V reward;
        M demo_end = qlt(V(1.0f), V(0.0f));
        if (demo_task(cfg.task)) {
            // TaskJumpingDemo._reward (task_base.py:194-211): distance between the demonstration's (filtered) action of row `counter`
            // and the action this step was given (get_last_action: before the filter), over the rows left when the episode began
            V cnt = T::ld(rec, R_DEMO), start = T::ld(rec, R_DEMO + 1);
            int row = (int)T::first(cnt);
            row = row < demo_len ? row : demo_len - 1;
            const float* a = demo_rows + (size_t)row * (size_t)(d + 38);
            V n2 = V(0.0f);
            if (d == 12) {
#pragma unroll
                for (int j = 0; j < 3; j++) { V e = T::ld_leg(a, j, 3) - act_last[12 + j]; n2 = n2 + e * e; }
                n2 = T::quad_sum(n2);
            } else {
                V av[12];   // (a row is d + 38 floats: entries 0..11 exist for every d; all loads go out together, not one trip to memory per term)
#pragma unroll
                for (int k = 0; k < 12; k++) av[k] = T::ld(a, k);
#pragma unroll
                for (int k = 0; k < 12; k++)
                    if (k < d) { V e = av[k] - act_last[k]; n2 = n2 + e * e; }
            }
            reward = qexp(qsqrt(n2) * (-0.35f)) / (V((float)demo_len) - start);
            cnt = cnt + 1.0f;
            T::st(rec, R_DEMO, cnt);
            demo_end = qge(cnt, V((float)demo_len));
        } else {
            reward = task_reward(cfg, t, force, cfg.task == QS_TASK_BACKFLIP_PPO ? pitch_backflip(s.qx, s.qy, s.qz, s.qw, t.switched) : V(0.0f));
        }
        V term = task_terminated(cfg, t, s, o.n_invalid, demo_end);
        bool timeout = sim_step > cfg.max_sim_steps;   // gym_env.py:245
        V done = timeout ? V(1.0f) : term;
        reward = reward + qsel(qgt(done, V(0.5f)), task_reward_end(cfg, t, term, now), V(0.0f));  // :250-251
        QS_PHASE(35)
        StepOut r; r.reward = reward; r.done = done; r.trunc = qsel(qgt(term, V(0.5f)), V(0.0f), done);  // :246
        r.resume = -1;
        if (cfg.wrapper_mode != QS_WRAP_NONE) {
            float* w = rec + R_WRAP;
            M running = qlt(done, V(0.5f));
            M go = qand(qand(running, qlt(w_phase, V(0.5f))), qgt(t.switched, V(0.5f)));
            const WrapTraits wt = wrap_traits(cfg.wrapper_mode);
            if (wt.landing_family) {
                V disarmed = T::ld(w, W_DISARMED);
                M pol = qlt(w_phase, V(0.5f)), tko = qand(qgt(w_phase, V(0.5f)), qlt(w_phase, V(1.5f))), lnd = qgt(w_phase, V(1.5f));
                go = qand(qand(running, pol), qand(qgt(wt.trigger_jumping ? t.is_jumping : t.switched, V(0.5f)), qlt(disarmed, V(0.5f))));   // landing_wrapper.py:54-66
                w_timer = qsel(go, now, w_timer);
                w_end = qsel(go, now + s.vlin.z * (1.0f / 9.81f), w_end);
#pragma unroll
                for (int k = 0; k < 15; k++) {
                    if ((d == 12) != (k >= 12)) continue;
                    if (d == 12) T::st_leg(w, W_ACTION + (k - 12), 3, qsel(go, act_in[k], T::ld_leg(w, W_ACTION + (k - 12), 3)));
                    else if (k < d) T::st(w, W_ACTION + k, qsel(go, act_in[k], T::ld(w, W_ACTION + k)));
                }
                M to_land = qand(running, tko);
                if (wt.pitch_takeoff) to_land = qand(to_land, qge(pitch_backflip(s.qx, s.qy, s.qz, s.qw, t.switched), V(5.0f * 3.14159265358979f / 8.0f)));  // landing_wrapper_backflip.py:23-24
                else to_land = qlt(V(1.0f), V(0.0f));
                M leave = qlt(V(1.0f), V(0.0f));
                if (wt.exit_kind == 1) leave = qand(qand(running, lnd), qgt(T::quad_sum(o.foot_contact), V(0.5f)));   // landing_wrapper_2.py:41-45
                if (wt.exit_kind == 2) leave = qand(qand(running, lnd), qlt(t.is_jumping, V(0.5f)));               // landing_wrapper_continuous.py:41-45
                w_phase = qsel(go, V(1.0f), qsel(to_land, V(2.0f), qsel(leave, V(0.0f), w_phase)));
                if (wt.one_shot) T::st(w, W_DISARMED, qsel(leave, V(1.0f), disarmed));                                  // landing_wrapper_2.py:68
            } else {                                              // go_to_rest_wrapper.py:43-57, 88-95
                V h_old = T::ld(w, W_HACT), h_act = s.pos.z;
                T::st(w, W_HOLD, h_old); T::st(w, W_HACT, h_act);
                go = qand(go, qand(qgt(T::quad_sum(o.foot_contact), V(3.5f)), qgt(h_act - h_old, V(0.0f))));
                w_phase = qsel(go, V(3.0f), w_phase);
                w_tstart = qsel(go, now, w_tstart);
                V a_own[3];                                       // interface_base.py:92-100 on the current joint angles
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    V lo = T::ld_leg(cfg.cmd_lo, j, 3), hi = T::ld_leg(cfg.cmd_hi, j, 3);
                    a_own[j] = clampv<V>((clampv<V>(s.q[j], lo, hi) - lo) * qrcp(hi - lo) * 2.0f - 1.0f, V(-1.0f), V(1.0f));
                }
                if (d == 12) {
#pragma unroll
                    for (int j = 0; j < 3; j++) T::st_leg(w, W_ACTION + j, 3, qsel(go, a_own[j], T::ld_leg(w, W_ACTION + j, 3)));
                } else {                                          // action_interface.py:41-44, 67-74: FR and RR legs
                    int kk = 0;
                    for (int half = 0; half < 2; half++)
                        for (int j = 0; j < 3; j++) {
                            if (cfg.action_space_mode == QS_ACT_SYMMETRIC_NO_HIP && j == cfg.symm_idx) continue;
                            V v = half == 0 ? T::template bcast<0>(a_own[j]) : T::template bcast<2>(a_own[j]);
                            T::st(w, W_ACTION + kk, qsel(go, v, T::ld(w, W_ACTION + kk)));
                            kk++;
                        }
                }
            }
            T::st(w, W_PHASE, w_phase); T::st(w, W_TIMER, w_timer); T::st(w, W_END, w_end); T::st(w, W_TSTART, w_tstart);
        }
        QS_PHASE(36)
        store_state(rec, s, o); store_task(rec, t);
#pragma unroll
        for (int j = 0; j < 3; j++) T::st_leg(rec, R_NEW_TAU + j, 3, o.tau_pd[j]);
        T::st(rec, R_SIM_STEP, V(i2f(sim_step))); T::st(rec, R_ENV_STEP, V(i2f(env_step))); T::st(rec, R_TOTAL_STEPS, V(i2f(total)));
        QS_PHASE(37)
        write_obs(cfg, obs, s, o, t, env_id, (uint32_t)total, cfg.task != QS_TASK_NO_TASK ? t.rpy : nullptr);
        QS_PHASE(38)
        return r;
    }

    // Per-reset parameter draws (env_randomizers/env_randomizer.py), Philox(seed; env, stream 1, episode, block)
    static QS_FN void randomize(const qs_config& cfg, float* rec, uint32_t env_id, int episode, bool force) {
        float* p = rec + R_PARAMS;
        if ((cfg.randomizer_flags & QS_RAND_KEEP) && !force) return;
        float mu = 1.0f, k[3], b[3], ml[3] = {go1::HIP_M, go1::THIGH_M, go1::CALF_M}, mt = go1::TRUNK_M, mp = 0.0f, rp[3] = {0, 0, 0};
        for (int j = 0; j < 3; j++) { k[j] = cfg.spring_k[j]; b[j] = cfg.spring_b[j]; }
        uint32_t r[16];
        for (int blk = 0; blk < 4; blk++) philox4x32(cfg.seed, env_id, 1u, (uint32_t)episode, (uint32_t)blk, r + 4 * blk);
        if (cfg.randomizer_flags & QS_RAND_GROUND) mu = 0.5f + 0.5f * u01(r[0]);                       // :287-289
        if ((cfg.randomizer_flags & QS_RAND_SPRINGS) && cfg.enable_springs)                           // :100-122
            for (int j = 0; j < 3; j++) {
                float lo = cfg.spring_k[j] * 0.9f, hi = cfg.spring_k[j] * 1.1f; k[j] = lo + (hi - lo) * u01(r[1 + j]);
                lo = cfg.spring_b[j] * 0.9f; hi = cfg.spring_b[j] * 1.1f; b[j] = lo + (hi - lo) * u01(r[4 + j]);
            }
        if (cfg.randomizer_flags & QS_RAND_MASSES) {                                                  // :56-83
            float legs = 0, legs0 = 0;
            for (int j = 0; j < 3; j++) { float m0 = ml[j]; ml[j] = m0 * 0.9f + (m0 * 1.1f - m0 * 0.9f) * u01(r[8 + j]); legs += 4 * ml[j]; legs0 += 4 * m0; }
            mp = u01(r[7]); rp[0] = -0.1f + 0.2f * u01(r[11]); rp[2] = -0.1f + 0.2f * u01(r[13]);
            mt = go1::TRUNK_M + 0.00101f + legs0 - legs - mp;   // env_randomizer.py:43-47,61-65: total_mass also counts the imu and floating-base links
        }
        T::st(p, P_MU, V(mu));
        for (int j = 0; j < 3; j++) {
            T::st(p, P_K + j, V(k[j])); T::st(p, P_B + j, V(b[j])); T::st(p, P_REST + j, V(cfg.spring_rest[j]));
            T::st(p, P_KP + j, V(cfg.kp[j])); T::st(p, P_KD + j, V(cfg.kd[j])); T::st(p, P_M_LEG + j, V(ml[j])); T::st(p, P_R_PAY + j, V(rp[j]));
        }
        T::st(p, P_M_TRUNK, V(mt)); T::st(p, P_M_PAY, V(mp));
    }

    // env.reset() (gym_env.py:278-297): randomize, spawn (quadruped.py:487-519), settle 2500 substeps toward the init pose
    // with the sim counter frozen (interface_base.py:182-200), task / sensor / filter reset.
    // First half of reset(): randomizer draws + the spawn state (quadruped.py:454-519), written into the record.  Used by the
    // streaming reset pool, whose settle then proceeds in slices through step(..., settle_n).
    static QS_FN void settle_spawn(const qs_config& cfg, float* rec, uint32_t env_id, int episode) {
        randomize(cfg, rec, env_id, episode, false);
        T::sync();
        typename S::State s; typename S::Out o;
        const V zero = V(0.0f);
        s.pos = mk3<V>(zero, zero, V(0.32f)); s.qx = zero; s.qy = zero; s.qz = zero; s.qw = V(1.0f);
        s.vlin = mk3<V>(zero, zero, zero); s.vang = mk3<V>(zero, zero, zero);
        s.q[0] = zero; s.q[1] = V(0.25f * PI); s.q[2] = V(-0.5f * PI);
        s.qd[0] = zero; s.qd[1] = zero; s.qd[2] = zero; s.warm = zero;
        o.foot_force = zero; o.foot_contact = zero; o.n_invalid = zero;
        for (int j = 0; j < 3; j++) { o.tau_pd[j] = zero; o.tau_spring[j] = zero; }
        store_state(rec, s, o);
        if (cfg.payload_soft) { typename S::Par P; load_par(cfg, rec, P); S::block_place(rec + R_BLOCK, s, P); }
    }
    // cfg.payload_soft: the block goes where the fixed constraint wants it for the state and parameters in the record (qs_set_state,
    // qs_set_params, reference-state initialisation; the reference leaves it at the spawn pose and lets the constraint drag it)
    static QS_FN void place_block(const qs_config& cfg, float* rec) {
        typename S::State s; typename S::Par P;
        load_state(rec, s); load_par(cfg, rec, P);
        S::block_place(rec + R_BLOCK, s, P);
    }
    static QS_FN void reset(const qs_config& cfg, float* rec, float* obs, uint32_t env_id, bool settle) {
        int episode = f2i(rec[R_EPISODE]) + 1;
        int total = f2i(rec[R_TOTAL_STEPS]);
        typename S::State s; typename S::Out o; Task t;
        const V zero = V(0.0f);
        if (settle) {
            randomize(cfg, rec, env_id, episode, false);
            T::sync();  // parameters were written by lane 0 of the quad
            s.pos = mk3<V>(zero, zero, V(0.32f)); s.qx = zero; s.qy = zero; s.qz = zero; s.qw = V(1.0f);
            s.vlin = mk3<V>(zero, zero, zero); s.vang = mk3<V>(zero, zero, zero);
            s.q[0] = zero; s.q[1] = V(0.25f * PI); s.q[2] = V(-0.5f * PI);
            s.qd[0] = zero; s.qd[1] = zero; s.qd[2] = zero; s.warm = zero;
            typename S::Par P;
            o.foot_force = zero; o.foot_contact = zero; o.n_invalid = zero;
            for (int j = 0; j < 3; j++) { o.tau_pd[j] = zero; o.tau_spring[j] = zero; }
            load_par(cfg, rec, P);
            if (cfg.payload_soft) S::block_place(rec + R_BLOCK, s, P);   // _add_base_mass_offset: the block appears next to the freshly spawned robot
            V cmd[3];
#pragma unroll
            for (int j = 0; j < 3; j++) cmd[j] = T::ld_leg(cfg.settle_cmd, j, 3);
            for (int n = 0; n < cfg.settle_steps; n++) {
                V tau[3]; S::actuate(cfg, P, s, cmd, o, tau, true);
                S::substep(cfg, P, s, tau, o, n == cfg.settle_steps - 1 || cfg.body_contacts, cfg.payload_soft ? rec + R_BLOCK : nullptr, obs, n == cfg.settle_steps - 1);
            }
            store_state(rec, s, o);
        } else {  // the record already holds a settled state (copied from the pre-settled pool)
            load_state(rec, s);
            o.foot_force = T::ld_leg(rec, R_FOOT_FORCE, 1); o.foot_contact = T::ld_leg(rec, R_FOOT_CONTACT, 1); o.n_invalid = T::ld(rec, R_N_INVALID);
#pragma unroll
            for (int j = 0; j < 3; j++) { o.tau_pd[j] = T::ld_leg(rec, R_TAU_PD + j, 3); o.tau_spring[j] = T::ld_leg(rec, R_TAU_SPRING + j, 3); }
        }
        load_task(rec, t);
        T::st(rec, R_SIM_STEP, V(i2f(0))); T::st(rec, R_ENV_STEP, V(i2f(0))); T::st(rec, R_EPISODE, V(i2f(episode)));
        for (int k = 0; k < 12; k++) {  // gym_env.py:325-327, 267-269
            V a = V(k < cfg.action_dim ? cfg.settle_action[k] : 0.0f);
            T::st(rec, R_LAST_ACTION + k, a);
            T::st(rec, R_XHIST + k, a); T::st(rec, R_XHIST + 12 + k, a); T::st(rec, R_YHIST + k, a); T::st(rec, R_YHIST + 12 + k, a);
        }
        if (cfg.action_space_mode == QS_ACT_CPG) {  // hopf_network.py:62-63: r ~ 0.1 U(0,1), theta = PHI[0,:]
            uint32_t rr[4]; philox4x32(cfg.seed, env_id, 3u, (uint32_t)episode, 0u, rr);
            for (int L = 0; L < 4; L++) { T::st(rec, R_CPG + L, V(0.1f * u01(rr[L]))); T::st(rec, R_CPG + 4 + L, V(cfg.cpg_phi[L])); }
        }
        if (cfg.wrapper_mode != QS_WRAP_NONE) {
            for (int k = 0; k < 20; k++) T::st(rec, R_WRAP + k, V(0.0f));
            T::st(rec, R_WRAP + W_HOLD, s.pos.z); T::st(rec, R_WRAP + W_HACT, s.pos.z);   // go_to_rest_wrapper.py:86-90
        }
        if (demo_task(cfg.task)) { T::st(rec, R_DEMO, V(0.0f)); T::st(rec, R_DEMO + 1, V(0.0f)); }   // task_base.py:177-183 (set_demo_counter comes after)
        task_reset(cfg, t, s, o, V(0.0f));
        store_task(rec, t);
#pragma unroll
        for (int j = 0; j < 3; j++) T::st_leg(rec, R_NEW_TAU + j, 3, o.tau_pd[j]);
        write_obs(cfg, obs, s, o, t, env_id, (uint32_t)total);
    }
};

}  // namespace qs
