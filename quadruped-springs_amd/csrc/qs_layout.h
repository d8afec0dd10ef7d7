// qs_layout.h -- persistent per-environment record (env-major: records[N][QS_REC] float32 in HBM, QS_REC_END of them used).
//
// A wavefront owns 16 consecutive environments = 16 records, of which it moves the leading part a step needs (704 B in, 608 B out per
// record by default, 272 B in and 176 B out for a slice of a reset's settle; more only for handles that use the optional layers below) HBM <-> LDS with coalesced 16-byte-per-lane loads / stores
// at kernel entry / exit; the lanes then pick their fields out of LDS.
// Fields mirror the state the reference carries between env.step() calls.
#pragma once

enum {
    // ---- read by every step, written by resets and setters only
    R_PARAMS = 0,         // 24: mu, k3, b3, rest3, kp3, kd3, m_trunk, m_leg3, m_pay, r_pay3
    // ---- read-write block [24, 174): what every step reads AND writes
    // rigid-body state, same order as qs_get_state rows (quadruped.py:107-207); with the parameters in front of it, all a settle needs
    R_POS = 24, R_QUAT = 27, R_VLIN = 31, R_VANG = 34, R_Q = 37, R_QD = 49,
    R_WARM = 61,          // 4: normal impulse of each foot at the previous substep (contact warm start)
    R_LAST_ACTION = 65,   // 12: gym_env.py:230,284
    R_XHIST = 77,         // 2 x 12: action_filter.py:98-108 (row 0 newest)
    R_YHIST = 101,        // 2 x 12
    R_SIM_STEP = 125, R_ENV_STEP = 126, R_EPISODE = 127, R_TOTAL_STEPS = 128,  // integers stored as float bit patterns
    // task scalars (task_base.py:44-59, 228-233; robot_tasks.py:418-425, 524-530), order of QS_INFO_TASK
    R_TASK = 129,         // 32 slots, see T_* below
    R_NEW_TAU = 161,      // 12: task._new_torque
    R_N_INVALID = 173,    // invalid contacts of the last substep (a result; kept here: get_reward_end_episode reads it)
    // ---- state of optional layers: moved only by handles that use them
    R_WRAP = 174,         // 20: scripted-phase machine of the landing / go-to-rest wrappers: phase, timer, end, t_start, h_old,
                          //     h_actual, held or ramp-start action [12], scripted, disarmed
    R_CPG = 194,          // 8: Hopf oscillator amplitudes r[4] and phases theta[4] (hopf_network.py:50)
    R_DEMO = 194,         // 2: demo counter and its value at the start of the episode (task_base.py:177-183); shares the CPG slots,
                          //    the DEMO tasks do not take the CPG action layer (qs_create refuses the combination)
    // ---- info block: results of the last substep that no step reads back (the reference keeps them as Python attributes; getters,
    // the pooled reset and the trace consumers use them).  Written by every step unless cfg.info_fields == 0.
    R_POSE_CACHE = 202,   // 9: task._pos_abs, _vel_abs, _orient_rpy (task_base.py:72-75)
    R_FOOT_FORCE = 211,   // 4
    R_FOOT_CONTACT = 215, // 4
    R_TAU_PD = 219,       // 12: observed motor torque of the last substep (quadruped.py:299)
    R_TAU_SPRING = 231,   // 12
    // ---- payload block as a body of its own (cfg.payload_soft; quadruped.py:778-819): position of its centre 3, quaternion 4, linear 3 and
    // angular 3 velocity (world), the six impulses of the fixed constraint at the last substep, the pivot gap (the QS_INFO_PAYLOAD_BLOCK
    // row).  Moved by the tile load / store only under cfg.payload_soft.
    R_BLOCK = 244,        // 20
    QS_REC_END = 264,     // end of the used floats (the LDS stride of a record under cfg.payload_soft)
    QS_REC = 288,         // HBM stride: 1152 B = 9 x 128 B, so that the 704-B range a step fetches starts on a 128-byte line (6 lines instead of
                          // 6 or 7: the fetch counters read 8 % less)
    // ---- ranges [begin, end) that the tile load / store move (multiples of 4 floats = 16-byte vector moves; a range may end inside the
    // next field: what is stored was loaded).  Loads start at 0, stores at QS_RW_BEGIN unless the parameters were rewritten.
    QS_RW_BEGIN = 24,
    QS_SETTLE_END = 68,   // parameters + rigid-body state + warm start: all that a slice of a reset's settle loads, and stores from QS_RW_BEGIN
    QS_HOT = 176,         // what a step loads (704 B) and, from QS_RW_BEGIN, stores (608 B) when nothing else is asked for
    QS_HOT_WRAP = 196,    // + the wrapper machine
    QS_HOT_ALL = 204,     // + CPG / DEMO slots
    QS_INFO_END = 244,    // + the info block; also the LDS stride of a record unless cfg.payload_soft (then QS_REC): 16 records + observation
                          //   and action rows = 20480 B per workgroup, eight workgroups per CU for the two-waves-per-SIMD kernel
};
enum { B_POS = 0, B_QUAT = 3, B_V = 7, B_W = 10, B_LAM = 13, B_GAP = 19, QS_BLOCK_DIM = 20 };
enum { P_MU = 0, P_K = 1, P_B = 4, P_REST = 7, P_KP = 10, P_KD = 13, P_M_TRUNK = 16, P_M_LEG = 17, P_M_PAY = 20, P_R_PAY = 21 };
enum { T_SWITCHED = 0, T_ALL_AIR = 1, T_IS_JUMPING = 2, T_TAKEOFF = 3, T_POSE_TO = 4, T_YAW_TO = 7, T_INIT_H = 8, T_MAX_FLIGHT = 9,
       T_MAX_FWD = 10, T_MAX_PITCH = 11, T_REL_MAX_H = 12, T_MAX_DX = 13, T_MAX_H = 14, T_CUM_FWD = 15, T_CUM_FT = 16,
       T_OLD_FWD = 17, T_ACTUAL_FWD = 18, T_BF_MAX_PITCH = 19,
       // TaskContinuousJumping2 (task_base.py:283-400): the unbounded per-jump arrays are only ever reduced to these sums
       T_JUMP_COUNT = 20, T_GOOD_JUMPS = 21, T_SUM_FWD = 22, T_SUM_FLOGF = 23 /* sum f log2 f */, T_SUM_HEIGHT = 24, T_SUM_PERF = 25,
       T_MAX_PERF = 26, T_LAST_PERF = 27, T_MAX_JUMP_H = 28, T_FIRST_JUMP = 29, T_END_JUMP = 30, T_N = 32 };

// one row of the per-substep trace tap (qs_set_trace), monitor_state.py:66-85
enum { TR_TIME = 0, TR_POS = 1, TR_QUAT = 4, TR_VLIN = 8, TR_VANG = 11, TR_Q = 14, TR_QD = 26, TR_TAU = 38, TR_TAU_SPRING = 50, TR_FOOT_FORCE = 62,
       TR_FOOT_CONTACT = 66 };
enum { W_PHASE = 0, W_TIMER = 1, W_END = 2, W_TSTART = 3, W_HOLD = 4, W_HACT = 5, W_ACTION = 6, W_SCRIPTED = 18, W_DISARMED = 19 };

#define QS_ENVS_PER_WAVE 16
#define QS_WAVE 64
