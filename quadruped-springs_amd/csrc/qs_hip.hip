// qs_hip.hip -- gfx950 kernels and the C ABI (include/qs_amd.h) of the batched Go1 + PEA simulation step.
//
// Launch geometry: one 64-lane wavefront per workgroup = 16 environments (a quad of lanes per environment, one lane
// per leg).  N = 8192 environments -> 512 single-wave workgroups spread over the 256 CUs; the kernel is a long chain of
// dependent fp32 VALU work per lane, so the design goal is the shortest per-lane instruction stream, registers instead of
// memory (everything between the tile load and the tile store lives in VGPRs), and DPP for the 4-lane reductions.
// HBM traffic: each wave moves the leading range of its 16 records that a step needs (704 B in, 608 B out per record by default,
// qs_layout.h) HBM -> LDS -> HBM with 16-byte-per-lane coalesced accesses, plus the action / observation / reward rows.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include "qs_env.h"
#include "qs_host.h"

using qs::Env;
using E = Env<LaneDev>;

#define QS_TILE_FLOATS (QS_ENVS_PER_WAVE * QS_REC_END)
#ifndef QS_LEAN_WAVES
#define QS_LEAN_WAVES(W) ((W) == 2)     // which step kernels keep the per-environment parameters in LDS (Env::step, LEAN): the two-waves-per-SIMD one
#endif

// ------------------------------------------------------------------ tile movement (coalesced, 16 B per lane)
// Range of a record that a step moves (qs_layout.h).  Loads [0, end): the parameters, the read-write block; the wrapper / CPG / DEMO
// slots behind them only for handles that use those layers (or that store the info block, which lies behind them); everything under
// cfg.payload_soft (the block's own state ends the record).  Stores [QS_RW_BEGIN, end) -- from 0 when a look-ahead reset rewrote the
// parameters --: through the read-write block, the optional layers' slots when in use, the info block under cfg.info_fields.
enum { TILE_INFO = QS_INFO_END, TILE_ALL = QS_REC_END };
__device__ __forceinline__ int tile_extent(const qs_config& cfg, bool store) {
    if (cfg.payload_soft) return TILE_ALL;
    if (cfg.info_fields) return store ? TILE_INFO : QS_HOT_ALL;
    if (cfg.action_space_mode == QS_ACT_CPG || E::demo_task(cfg.task)) return QS_HOT_ALL;
    if (cfg.wrapper_mode != QS_WRAP_NONE) return QS_HOT_WRAP;
    return QS_HOT;
}
// `stride`: floats between two records in LDS (QS_REC_END, or QS_INFO_END in the step kernels of handles without the payload block's state)
//
// All moves are BATCHED: every round's load is issued before the first result is used (global -> registers -> LDS on the way in, LDS ->
// registers -> global on the way out).  Written as a rolled loop each round was `global_load_dwordx4; s_waitcnt vmcnt(0); ds_write_b128`:
// eleven trips to memory one after the other, 5.7 k cycles = 2.3 us at the entry of every launch whatever N (256 environments as well as
// 8192: a latency chain, not a bandwidth burst -- tools/phase_profile.py, profiles/r03_e_phase_cycles.md); batched it is one trip.
// PER > 0: float4s per record known at compile time (the index split is a multiply-shift and the rounds need no predicate when they divide).
// (keeps a batch of loaded values in front of the predicated stores that consume them: without a use of its own the compiler sinks each
// load into its store's branch, one memory trip per round again)
__device__ __forceinline__ void keep_loaded(const float4& v) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }
__device__ __forceinline__ void keep_loaded(float v) { asm volatile("" ::"v"(v)); }
template <int PER> __device__ __forceinline__ void tile_load_rounds(float4* __restrict__ dst, const float4* __restrict__ src, int nenv, int stride4, int per_rt) {
    constexpr int MAX_PER = QS_REC_END / 4;
    constexpr int ROUNDS = PER > 0 ? (QS_ENVS_PER_WAVE * PER + QS_WAVE - 1) / QS_WAVE : (QS_ENVS_PER_WAVE * MAX_PER + QS_WAVE - 1) / QS_WAVE;
    const int per = PER > 0 ? PER : per_rt, total = QS_ENVS_PER_WAVE * per;
    const unsigned inv = ((1u << 20) + (unsigned)per - 1u) / (unsigned)per;      // i / per = (i * inv) >> 20, exact for i * per < 2^20
    float4 v[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        const int i = (int)threadIdx.x + r * QS_WAVE;
        // (every round loads -- beyond the range, the tile's first float4 again -- and only the LDS write is predicated: an array element
        // assigned under a condition would send the whole array to scratch)
        const bool in = (PER > 0 && (QS_ENVS_PER_WAVE * PER) % QS_WAVE == 0) || i < total;
        const int e = PER > 0 ? i / PER : (int)(((unsigned)i * inv) >> 20), o = i - e * per;
        v[r] = src[in ? (e < nenv ? e : 0) * (QS_REC / 4) + o : 0];   /* tail quads replay the tile's first record (never stored) */
    }
    if (!(PER > 0 && (QS_ENVS_PER_WAVE * PER) % QS_WAVE == 0)) {
#pragma unroll
        for (int r = 0; r < ROUNDS; r++) keep_loaded(v[r]);
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        const int i = (int)threadIdx.x + r * QS_WAVE;
        if ((PER > 0 && (QS_ENVS_PER_WAVE * PER) % QS_WAVE == 0) || i < total) {
            const int e = PER > 0 ? i / PER : (int)(((unsigned)i * inv) >> 20), o = i - e * per;
            dst[e * stride4 + o] = v[r];
        }
    }
}
__device__ __forceinline__ void tile_load(float* lds, const float* __restrict__ g, int first_env, int n_envs, int extent, int stride = QS_REC_END) {
    // Only the leading range [0, extent) of each record is fetched (QS_HOT by default: the info block behind it is written, never read
    // back, by a step).
    const float4* src = reinterpret_cast<const float4*>(g + (size_t)first_env * QS_REC);
    float4* dst = reinterpret_cast<float4*>(lds);
    const int nenv = min(QS_ENVS_PER_WAVE, n_envs - first_env);
    if (extent == QS_HOT) tile_load_rounds<QS_HOT / 4>(dst, src, nenv, stride / 4, 0);
    else if (extent == QS_SETTLE_END) tile_load_rounds<QS_SETTLE_END / 4>(dst, src, nenv, stride / 4, 0);
    else tile_load_rounds<0>(dst, src, nenv, stride / 4, extent / 4);
}
template <int PER> __device__ __forceinline__ void tile_store_rounds(float4* __restrict__ dst, const float4* __restrict__ src, int nenv, int stride4, int b4, int per_rt) {
    constexpr int MAX_PER = QS_REC_END / 4;
    constexpr int ROUNDS = PER > 0 ? (QS_ENVS_PER_WAVE * PER + QS_WAVE - 1) / QS_WAVE : (QS_ENVS_PER_WAVE * MAX_PER + QS_WAVE - 1) / QS_WAVE;
    const int per = PER > 0 ? PER : per_rt, total = nenv * per;
    const unsigned inv = ((1u << 20) + (unsigned)per - 1u) / (unsigned)per;
    float4 v[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        const int i = (int)threadIdx.x + r * QS_WAVE;
        const int e = PER > 0 ? i / PER : (int)(((unsigned)i * inv) >> 20), o = i - e * per + b4;
        v[r] = src[i < total ? e * stride4 + o : 0];
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) keep_loaded(v[r]);
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        const int i = (int)threadIdx.x + r * QS_WAVE;
        if (i < total) { const int e = PER > 0 ? i / PER : (int)(((unsigned)i * inv) >> 20), o = i - e * per + b4; dst[e * (QS_REC / 4) + o] = v[r]; }
    }
}
__device__ __forceinline__ void tile_store(const float* lds, float* __restrict__ g, int first_env, int n_envs, int begin, int end, int stride = QS_REC_END) {
    float4* dst = reinterpret_cast<float4*>(g + (size_t)first_env * QS_REC);
    const float4* src = reinterpret_cast<const float4*>(lds);
    const int nenv = min(QS_ENVS_PER_WAVE, n_envs - first_env);
    if (begin == QS_RW_BEGIN && end == QS_HOT) tile_store_rounds<(QS_HOT - QS_RW_BEGIN) / 4>(dst, src, nenv, stride / 4, QS_RW_BEGIN / 4, 0);
    else if (begin == QS_RW_BEGIN && end == QS_SETTLE_END) tile_store_rounds<(QS_SETTLE_END - QS_RW_BEGIN) / 4>(dst, src, nenv, stride / 4, QS_RW_BEGIN / 4, 0);
    else tile_store_rounds<0>(dst, src, nenv, stride / 4, begin / 4, (end - begin) / 4);
}

// the observation rows of the tile (LDS, QS_MAX_OBS apart) to the caller's rows ([N, od], or [N, od + 2] in the fused layout) and the
// handle's own copy.  i / od by multiply-shift: exact for i < 16 * 64, od <= 64 with inv = ceil(2^16 / od) (one division per wave
// instead of two per element).  Batched like the tile moves: the LDS reads of all rounds first, then the stores.
__device__ __forceinline__ void obs_store(const float* s_obs, int nrow, int od, int first, float* __restrict__ obs_out, bool fused, float* __restrict__ obs_keep) {
    const unsigned inv = (65536u + (unsigned)od - 1u) / (unsigned)od;
    const int row = fused ? od + 2 : od, total = nrow * od;
    constexpr int ROUNDS = QS_ENVS_PER_WAVE * QS_MAX_OBS / QS_WAVE;
    float v[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        const int i = (int)threadIdx.x + r * QS_WAVE;
        const int e = (int)(((unsigned)i * inv) >> 16), o = i - e * od;
        v[r] = s_obs[i < total ? e * QS_MAX_OBS + o : 0];
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) keep_loaded(v[r]);
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        const int i = (int)threadIdx.x + r * QS_WAVE;
        if (i < total) {
            const int e = (int)(((unsigned)i * inv) >> 16), o = i - e * od;
            obs_out[(size_t)(first + e) * row + o] = v[r];
            obs_keep[(size_t)first * od + i] = v[r];
        }
    }
}

// ------------------------------------------------------------------ look-ahead resets
// reset() = randomizer draws + spawn + settle_steps (2500) substeps (gym_env.py:278-297, 323-329): 250 env-steps' worth of physics whose
// result depends on (seed, global environment id, episode number) only, not on the trajectory.  So every environment's NEXT K reset states
// are computed ahead of time: slot (env, X mod K) holds the settled state of episode X once its R_EPISODE field says X.  A reset to
// episode X copies that slot and queues the settle of episode X + K; the queue is served by extra workgroups of k_step ("settle lanes"),
// action_repeat substeps per launch through the step's own substep loop, so the settle work of a run's resets is executed inside the run,
// next to the stepping, on SIMDs the environments leave idle.  A reset that finds its slot not ready (K consecutive episodes shorter than
// one settle: counted as a stall) settles inside the step as with K = 0 -- every output is bitwise what reset_lookahead = 0 produces.
struct LookAhead {
    float* slots;           // [N][K] records
    int* cur;               // [N]: the episode each environment is in (written by every reset; what the planning scan reads)
    int* handed;            // [N]: last episode of each environment whose settle a lane has taken (or that was settled at create)
    int K;
};
// Settle lanes: workgroups beyond the environments' ones advance records of a staging area through a reset's settle, one slice of substeps
// per launch; ctl = the counters in qs_handle::d_stats.  The staging area is split into QS_COHORTS slices whose settles start QS_COHORTS-th
// of an epoch (= the launches one settle takes) apart, so that a wanted state waits at most epoch / QS_COHORTS launches for a lane.
#define QS_COHORTS 5
#define QS_MAX_SLICE 2048
struct SettleLanes { float* staging; const int2* stage_jobs; int n_env_waves, slice; int wave0[QS_COHORTS + 1] /* first settle wave of each cohort */,
                     spawn[QS_COHORTS], last[QS_COHORTS], settle_n[QS_COHORTS]; };
struct TraceTap { float* rows; int env; };
struct DemoTab { const float* rows; int length; };   // qs_set_demo: the demonstration of the DEMO tasks
// Host path (qs_host_step_*): the terminal observations of the step as a compact list [cap][1 + obs_dim] (environment index as int bits,
// observation) behind the result block, so that ONE device-to-host copy brings everything a VecEnv.step_wait returns.  The list's
// fill count alternates between two counters: a step counts in cnt[parity] and clears the other one for the next step.
struct TermTail { float* rows; int cap, parity; };
enum { CTL_SETTLE_SUBSTEPS = 0, CTL_RESETS = 1, CTL_SERVED = 2, CTL_SETTLED = 3, CTL_BACKLOG = 4, CTL_STALLS = 6,
       CTL_R = 8 /* one per cohort */, CTL_TERM_CNT = 8 + QS_COHORTS /* two */, CTL_DEV = 10 + QS_COHORTS /* QS_DEVCTR_*: the rare paths' telemetry */,
#if defined(QS_PROBE_LAZY) || defined(QS_PROBE_SWEEPS) || defined(QS_PROBE_WARM)
       CTL_N = 12 + QS_COHORTS + 10 /* the counting builds' counters (qs_core.h, tools/probe_*.py) */ };
#else
       CTL_N = 12 + QS_COHORTS };
#endif

// settled-state fields a look-ahead reset copies into the record (everything the 2500-substep settle determines), and the slot's tag.
// Every load is issued before the first value is used: as four rolled loops (`rec[i] = src[i]`) the copy was a load, a wait and an LDS write
// per element -- about twenty trips to memory, one after the other, on the path of the wave that every launch waits for (round 3: found in
// the ISA; the launch got 2 us shorter).
__device__ __forceinline__ float copy_settled(float* rec, const float* __restrict__ src, bool block) {
    static_assert(R_PARAMS == 0 && R_PARAMS + QS_PARAM_DIM == R_POS, "parameters and rigid-body state are one range");
    const int lane = threadIdx.x & 3;
    constexpr int END_A = R_WARM + 4, NA = (END_A + 3) / 4;                   // [0, 65): parameters, rigid-body state, warm start
    constexpr int NB = (R_TAU_SPRING + 12 - R_FOOT_FORCE) / 4;               // [211, 243): contact results, torques
    constexpr int NC = QS_BLOCK_DIM / 4;                                      // the payload block settled with the robot
    static_assert((R_TAU_SPRING + 12 - R_FOOT_FORCE) % 4 == 0 && QS_BLOCK_DIM % 4 == 0, "whole rounds");
    float a[NA], b[NB], c[NC];
#pragma unroll
    for (int k = 0; k < NA; k++) { const int i = lane + 4 * k; a[k] = src[i < END_A ? i : 0]; }
#pragma unroll
    for (int k = 0; k < NB; k++) b[k] = src[R_FOOT_FORCE + lane + 4 * k];
    const float ninv = src[R_N_INVALID], tag = src[R_EPISODE];
    if (block) {
#pragma unroll
        for (int k = 0; k < NC; k++) c[k] = src[R_BLOCK + lane + 4 * k];
    }
#pragma unroll
    for (int k = 0; k < NA; k++) { const int i = lane + 4 * k; if (i < END_A) rec[i] = a[k]; }
#pragma unroll
    for (int k = 0; k < NB; k++) rec[R_FOOT_FORCE + lane + 4 * k] = b[k];
    if (lane == 0) rec[R_N_INVALID] = ninv;
    if (block) {
#pragma unroll
        for (int k = 0; k < NC; k++) rec[R_BLOCK + lane + 4 * k] = c[k];
    }
    return tag;
}
// rows of at most QS_MAX_OBS floats from LDS to global memory by the four lanes of a quad, the LDS reads of all rounds first
__device__ __forceinline__ void quad_row_store(float* __restrict__ dst, const float* row, int n) {
    float v[QS_MAX_OBS / 4];
#pragma unroll
    for (int k = 0; k < QS_MAX_OBS / 4; k++) { const int i = (int)(threadIdx.x & 3) + 4 * k; v[k] = row[i < n ? i : 0]; }
#pragma unroll
    for (int k = 0; k < QS_MAX_OBS / 4; k++) keep_loaded(v[k]);
#pragma unroll
    for (int k = 0; k < QS_MAX_OBS / 4; k++) { const int i = (int)(threadIdx.x & 3) + 4 * k; if (i < n) dst[i] = v[k]; }
}
// A reset of `env` to episode X (called by the four lanes of its quad): copies the slot of that episode into the record -- all of its loads
// go out together with the one of the slot's tag, ONE trip to memory on the path of a wave that every launch waits for -- and says
// whether the slot held that episode (if not, the in-step settle overwrites what was copied).  Nothing is queued here: the planning
// scan of the settle lanes reads `cur`.
__device__ __forceinline__ bool lookahead_take(const LookAhead& la, unsigned long long* __restrict__ ctl, float* rec, int env, int X, bool block, bool count = true) {
    if (la.K == 0) return false;
    const float* slot = la.slots + ((size_t)env * la.K + (size_t)(X % la.K)) * QS_REC;
    const float tag = copy_settled(rec, slot, block);
    const bool ready = qs::f2i(tag) == X;
    if ((threadIdx.x & 3) == 0) {
        la.cur[env] = X;
        if (count) atomicAdd(&ctl[ready ? CTL_SERVED : CTL_STALLS], 1ull);
    }
    return ready;
}
__device__ __forceinline__ void zero_tile_tail(float* lds, int from, int stride) {   // floats [from, stride) of the 16 records in LDS
    const int per = stride - from;
    for (int i = threadIdx.x; i < QS_ENVS_PER_WAVE * per; i += QS_WAVE) { const int e = i / per; lds[e * stride + from + (i - e * per)] = 0.0f; }
}

// ------------------------------------------------------------------ kernels
__global__ __launch_bounds__(QS_WAVE, 1) void k_init(const qs_config* __restrict__ cfgp, float* __restrict__ recs) {
    const qs_config& cfg = *cfgp;
    int env = blockIdx.x * QS_ENVS_PER_WAVE + (threadIdx.x >> 2);
    if (env >= cfg.n_envs) return;
    float* r = recs + (size_t)env * QS_REC;
    for (int i = threadIdx.x & 3; i < QS_REC_END; i += 4) r[i] = 0.0f;
    LaneDev::sync();
    if ((threadIdx.x & 3) == 0) {
        r[R_EPISODE] = qs::i2f(-1);
        r[R_QUAT + 3] = 1.0f; r[R_POS + 2] = 0.32f; r[R_TASK + T_FIRST_JUMP] = 1.0f;
        for (int L = 0; L < 4; L++) { r[R_Q + 3 * L + 1] = 0.78539816339f; r[R_Q + 3 * L + 2] = -1.57079632679f; }
    }
    E::randomize(cfg, r, (uint32_t)(env + cfg.env_id_offset), -1, true);
}

// QuadrupedGymEnv.step for 16 environments per wave (gym_env.py:227-256); auto-reset per the SB3 VecEnv convention.
// The body is compiled twice (k_step / k_step_dense below) under different register budgets.
// SOFT: the build for handles with cfg.payload_soft whose common-path part holds the payload block's rows (qs_core.h; implicit cone only)
template <bool CONE, int WAVES, bool SOFT> static __device__ __forceinline__ void step_body(const qs_config* __restrict__ cfgp, float* __restrict__ recs,
                                                 const float* __restrict__ actions, float* __restrict__ obs_out,
                                                 float* __restrict__ rew_out, uint8_t* __restrict__ done_out,
                                                 uint8_t* __restrict__ trunc_out, float* __restrict__ obs_keep,
                                                 float* __restrict__ term_obs, LookAhead la,
                                                 unsigned long long* __restrict__ stats, SettleLanes lanes, TraceTap tap, DemoTab demo, TermTail tail) {
    // the friction model is compiled in (qs_config::friction_cone picks the kernel at launch).  Everything of the full build is inlined in
    // both kernels: as real functions (round 2) the many-rows solvers took State / Out by reference, which kept them in memory around the
    // call, and expressions that then span a store and a load are no longer contracted into the FMAs the common-path build forms -- a wave
    // on a rare path gave its other 15 environments different last bits (tests/test_gpu_round2.py::test_results_do_not_depend_on_wave_mates).
    using E = Env<LaneDev, CONE, false>;                // the full build: resets, in-step settle, the rest of a handed-over step
    using EH = Env<LaneDev, CONE, true, SOFT>;          // the env step: common-path substeps, then (rare) the full build's (qs_env.h, Env::step)
    // LDS (sized at launch, step_lds_bytes): the 16 records at stride `ls`, the observation rows, the action rows
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    const qs_config& cfg = *cfgp;
    const int ls = cfg.payload_soft ? (int)QS_REC_END : (int)QS_INFO_END;
    float* const s_rec = s_dyn;
    float* const s_obs = s_dyn + QS_ENVS_PER_WAVE * ls;
    float* const s_act = s_obs + QS_ENVS_PER_WAVE * QS_MAX_OBS;
    QS_PHASE_BEGIN
    if (tail.rows && blockIdx.x == 0 && threadIdx.x == 0) stats[CTL_TERM_CNT + (tail.parity ^ 1)] = 0ull;
    const bool settling = (int)blockIdx.x >= lanes.n_env_waves;          // wave-uniform: this workgroup settles staging records
    const int sw = (int)blockIdx.x - lanes.n_env_waves;                  // settle lanes: which of them
    int cohort = 0;
#pragma unroll
    for (int c = 1; c < QS_COHORTS; c++) cohort += settling && sw >= lanes.wave0[c] ? 1 : 0;
    const int first = settling ? cohort * lanes.slice + (sw - lanes.wave0[cohort]) * QS_ENVS_PER_WAVE : (int)blockIdx.x * QS_ENVS_PER_WAVE;
    // CTL_R[cohort] is only written between launches (k_lookahead_plan)
    const int limit = settling ? cohort * lanes.slice + (int)stats[CTL_R + cohort] : cfg.n_envs;
    if (first >= limit) return;
    const int settle_n = settling ? lanes.settle_n[cohort] : 0;
    if (settling && settle_n == 0) return;                               // cohort not started yet
    float* const base = settling ? lanes.staging : recs;
    const int slot = threadIdx.x >> 2;
    const int env = first + slot;
    const bool valid = env < limit;
    const int d = cfg.action_dim, od = cfg.obs_dim;
    QS_PHASE(26)
    // the quad of an environment fetches its action row (lane l takes entries l, l + 4, l + 8) -- issued before the tile loads, whose
    // latency then covers it
    float a_pre[3] = {0.0f, 0.0f, 0.0f};
    int2 job = make_int2(0, 0);                                          // settle lanes: whose reset this record is (environment, episode)
    if (!settling) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int k = (int)(threadIdx.x & 3u) + 4 * j;
            if (k < d && valid) a_pre[j] = actions[(size_t)env * d + k];
        }
    } else job = lanes.stage_jobs[valid ? env : first];
    const bool spawn = settling && lanes.spawn[cohort], last = settling && lanes.last[cohort];
    const int load_extent = settling ? (spawn ? 0 : (cfg.payload_soft ? (int)TILE_ALL : (int)QS_SETTLE_END)) : tile_extent(cfg, false);
    // (a settle's first slice writes parameters and spawn state itself and reads nothing; its last slice leaves a whole record behind,
    // of which only the settled fields mean anything: the rest is zero rather than whatever the LDS held)
    if (load_extent > 0) tile_load(s_rec, base, first, limit, load_extent, ls);
    if (spawn || last) zero_tile_tail(s_rec, load_extent, ls);
    QS_PHASE(27)
    if (!settling) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int k = (int)(threadIdx.x & 3u) + 4 * j;
            if (k < d) s_act[slot * 12 + k] = a_pre[j];
        }
    }
    QS_PHASE(28)
    __syncthreads();
    QS_PHASE(29)
    float* rec = s_rec + slot * ls;
    float* ob = s_obs + slot * QS_MAX_OBS;
    if (cfg.info_fields && !settling && (threadIdx.x & 3) == 0) rec[QS_INFO_END - 1] = 0.0f;   // the pad float behind the info block, stored with it
    const uint32_t gid = (uint32_t)((settling ? job.x : env) + cfg.env_id_offset);
    if (spawn) { E::settle_spawn(cfg, rec, gid, job.y); LaneDev::sync(); }
    const bool any_trace = tap.rows != nullptr && !settling && tap.env >= first && tap.env < first + QS_ENVS_PER_WAVE;   // wave-uniform
    QS_PHASE(13)
    float* const trow = any_trace && env == tap.env ? tap.rows : nullptr;
    // The env step starts in the common-path build; a wave in which some environment needs a rare path goes on in the full build from the
    // substep where that shows (Env::step: the state of that moment is in the LDS record, the prologue's results in the observation row).
    // A payload_soft handle under the friction pyramid has its block's rows in no common-path build: its first substep already hands over.
    typename E::StepOut r;
    {
        const typename EH::StepOut rh = EH::template step<false, QS_LEAN_WAVES(WAVES)>(cfg, rec, s_act + slot * 12, ob, gid, settle_n, trow, any_trace, demo.rows, demo.length);
        r.reward = rh.reward; r.done = rh.done; r.trunc = rh.trunc; r.resume = rh.resume;
    }
    if (__builtin_expect(r.resume >= 0, 0)) r = E::template step<true>(cfg, rec, s_act + slot * 12, ob, gid, settle_n, trow, any_trace, demo.rows, demo.length, r.resume);
    QS_PHASE(14)
    if (settling) {
        if (valid && (threadIdx.x & 3) == 0) atomicAdd(&stats[CTL_SETTLE_SUBSTEPS], (unsigned long long)settle_n);
        __syncthreads();
        // a slice of a settle changes the rigid-body state and the warm start; its first slice also drew the parameters, its last one
        // leaves the info block's results (and n_invalid) that copy_settled hands to a reset
        tile_store(s_rec, base, first, limit, spawn ? 0 : (int)QS_RW_BEGIN,
                   cfg.payload_soft ? (int)TILE_ALL : (last ? (int)TILE_INFO : (int)QS_SETTLE_END), ls);
        return;
    }
    const bool dn = r.done > 0.5f;
    if (valid && (threadIdx.x & 3) == 0) {
        if (rew_out) { rew_out[env] = r.reward; done_out[env] = dn ? 1 : 0; trunc_out[env] = r.trunc > 0.5f ? 1 : 0; }
        else {   // fused layout (qs_step_fused): one row [obs | reward | done + 2 * truncated] per environment
            float* row = obs_out + (size_t)env * (od + 2);
            row[od] = r.reward; row[od + 1] = (dn ? 1.0f : 0.0f) + (r.trunc > 0.5f ? 2.0f : 0.0f);
        }
    }
    bool any_reset = false;   // wave-uniform: a reset rewrote the parameters of some record of the tile
    if (cfg.auto_reset) {   // a finished environment takes its look-ahead state, or settles here when that is not ready / reset_lookahead = 0
        const bool do_reset = dn && valid;
        if (__builtin_expect(__any(do_reset), 0)) {
            any_reset = true;
            LaneDev::sync();
            bool ahead = false;   // this environment's settled reset state was ready
            if (do_reset) {  // keep the terminal observation (SB3: infos[i]["terminal_observation"])
                quad_row_store(term_obs + (size_t)env * od, ob, od);
                if ((threadIdx.x & 3) == 0) atomicAdd(&stats[CTL_RESETS], 1ull);
                if (tail.rows) {   // host path: also as a row of the compact list
                    int at = 0;
                    if ((threadIdx.x & 3) == 0) at = (int)atomicAdd(&stats[CTL_TERM_CNT + tail.parity], 1ull);
                    at = __shfl(at, (int)(threadIdx.x & ~3u));
                    if (at < tail.cap) {
                        float* row = tail.rows + (size_t)at * (od + 1);
                        if ((threadIdx.x & 3) == 0) row[0] = qs::i2f(env);
                        quad_row_store(row + 1, ob, od);
                    }
                }
                ahead = lookahead_take(la, stats, rec, env, qs::f2i(rec[R_EPISODE]) + 1, cfg.payload_soft != 0);
            }
            LaneDev::sync();
            if (__builtin_expect(__any(do_reset && !ahead), 0)) {
                // no settled state ahead (reset_lookahead = 0, or all K states used up faster than the lanes settle): the settle happens here,
                // in the kernel's full build (cold code behind the step).  Its solver's v_mfma_f32_4x4x1 ignores EXEC, so it must not run
                // under a divergent branch: what the step produced is published first, then EVERY quad runs the reset on its LDS copy (as
                // k_reset does) and only the resetting environments keep the result -- one whose state was ready gets from its own settle the
                // very bits it had copied from its slot.
                __syncthreads();
                tile_store(s_rec, recs, first, cfg.n_envs, QS_RW_BEGIN, tile_extent(cfg, true), ls);
                obs_store(s_obs, min(QS_ENVS_PER_WAVE, cfg.n_envs - first), od, first, obs_out, rew_out == nullptr, obs_keep);
                __syncthreads();
                E::reset(cfg, rec, ob, gid, true);
                if (do_reset && !ahead && (threadIdx.x & 3) == 0) atomicAdd(&stats[CTL_SETTLE_SUBSTEPS], (unsigned long long)cfg.settle_steps);
                LaneDev::sync();
                if (do_reset) {
                    float* g = recs + (size_t)env * QS_REC;
                    const int end = tile_extent(cfg, true);
                    for (int i = threadIdx.x & 3; i < end; i += 4) g[i] = rec[i];
                    for (int i = threadIdx.x & 3; i < od; i += 4) {
                        if (rew_out) obs_out[(size_t)env * od + i] = ob[i];
                        else obs_out[(size_t)env * (od + 2) + i] = ob[i];
                        obs_keep[(size_t)env * od + i] = ob[i];
                    }
                }
                return;
            }
            if (do_reset) E::reset(cfg, rec, ob, gid, false);
        }
    }
    __syncthreads();
    tile_store(s_rec, recs, first, cfg.n_envs, any_reset ? 0 : (int)QS_RW_BEGIN, tile_extent(cfg, true), ls);
    obs_store(s_obs, min(QS_ENVS_PER_WAVE, cfg.n_envs - first), od, first, obs_out, rew_out == nullptr, obs_keep);
    QS_PHASE(15)
}

#define QS_STEP_ARGS const qs_config* __restrict__ cfgp, float* __restrict__ recs, const float* __restrict__ actions, float* __restrict__ obs_out,    \
                     float* __restrict__ rew_out, uint8_t* __restrict__ done_out, uint8_t* __restrict__ trunc_out, float* __restrict__ obs_keep, \
                     float* __restrict__ term_obs, LookAhead la, unsigned long long* __restrict__ stats, SettleLanes lanes, TraceTap tap, DemoTab demo, TermTail tail
#define QS_STEP_PASS cfgp, recs, actions, obs_out, rew_out, done_out, trunc_out, obs_keep, term_obs, la, stats, lanes, tap, demo, tail
// One wave per SIMD: the whole 512-entry register file (256 VGPR + 256 AGPR) for one wave.  The common-path substep loop holds no scratch
// instruction (tools/isa_headline.py); the cold code behind it -- the full build's substeps of a handed-over step, the in-step settle --
// spills (~110 values, ~470 B of scratch per lane).  The launch time is one wave's instruction stream, so this is the variant while the
// grid does not oversubscribe the chip's SIMDs.
template <bool CONE, bool SOFT> __global__ __launch_bounds__(QS_WAVE, 1) void k_step(QS_STEP_ARGS) { step_body<CONE, 1, SOFT>(QS_STEP_PASS); }
// Two waves per SIMD: 256 registers per wave (all of them VGPRs: the compiler takes no AGPRs under this budget), a second wave to issue from
// while the first waits on a dependent result.  Slower per wave, faster per chip once every SIMD has work queued (N = 16384: 146 M
// env-steps/s against 110 M at 8192; 65536: 222 M).  Its common-path loop keeps the per-environment parameters in LDS (Env::step, LEAN):
// 23 scratch instructions per substep where round 3 had 65 (~1900 values of the whole kernel are spilled, nearly all of them in the cold
// code) -- which measured as NO change of throughput (146.3 against 146.5 M at 16384, round 4): with two waves per SIMD the loop is bound by
// VALU issue, and the scratch traffic hides behind the other wave.
template <bool CONE, bool SOFT> __global__ __launch_bounds__(QS_WAVE, 2) void k_step_dense(QS_STEP_ARGS) { step_body<CONE, 2, SOFT>(QS_STEP_PASS); }

// Settle lanes, between two settles of a cohort (an epoch = the launches one settle takes): the staging records that finished settling go
// to the look-ahead slots of their environments (R_EPISODE marks the slot as holding that episode) ...
__global__ void k_lookahead_publish(unsigned long long* __restrict__ ctl, const float* __restrict__ staging, const int2* __restrict__ stage_jobs,
                                    LookAhead la, int cohort, int slice) {
    const int n = (int)ctl[CTL_R + cohort];
    const float* src = staging + (size_t)cohort * slice * QS_REC;
    const int2* jobs = stage_jobs + (size_t)cohort * slice;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)n * QS_REC_END; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i / QS_REC_END), f = (int)(i % QS_REC_END);
        const int2 job = jobs[e];
        // a state whose episode the environment has reached or passed meanwhile (it settled in place: a stall) is of no use -- and must not
        // land in a slot that a later episode's state is going to as well.  The live episodes of an environment (the K after its current
        // one) have K different slots.
        if (job.y <= la.cur[job.x]) continue;
        float* dst = la.slots + ((size_t)job.x * la.K + (size_t)(job.y % la.K)) * QS_REC;
        dst[f] = f == R_EPISODE ? qs::i2f(job.y) : src[(size_t)e * QS_REC + f];
        if (f == 0) atomicAdd(&ctl[CTL_SETTLED], 1ull);      // states DELIVERED (a stale one, dropped above, does not count)
    }
}
// ... then the cohort takes what the environments' windows lack: environment e in episode X wants the states of X + 1 .. X + K; those up to
// handed[e] are settled or being settled.  One block scans all environments, starting at a rotating offset, in two passes:
//   1. the URGENT ones -- more than a quarter of the window missing (K = 16: fewer than 12 states ready or on the way) -- get all they lack,
//      whatever that takes (up to the `slice` records the launch carries lanes for).  That is the start of a training run: every robot
//      falls within a few dozen steps, the settle work is a multiple of the stepping work, the launch runs in several rounds -- which
//      costs what the work weighs, whereas a reset that finds no state ready settles inside a step at 170 steps' time;
//   2. the others share what is left of `base_cap` = the SIMDs the environments' waves leave idle: after a burst (everybody at the 10-s
//      limit in the same step -- the usual picture when all environments of a learner start together) nobody is short of states, and
//      working the burst off over a few cohorts costs the launches nothing, where taking it at once would double their time for an epoch.
// Decided HERE, on the device: the host launches ahead of the GPU by hundreds of steps and knows nothing current (round 3's first version
// trimmed the lanes to the idle SIMDs from the host: 0.6 M env-steps/s under a policy that throws every robot down every 38 steps, against
// 33.6 M; tools/falling_policy_rate.py).  take = 0: count only (the backlog counter).
__global__ __launch_bounds__(1024) void k_lookahead_plan(unsigned long long* __restrict__ ctl, LookAhead la, int n_envs, int2* __restrict__ stage_jobs, int cohort,
                                                          int slice, int base_cap, int take, int offset) {
    __shared__ int s_count, s_want, s_urgent;
    if (threadIdx.x == 0) { s_count = 0; s_want = 0; s_urgent = 0; }
    __syncthreads();
    int2* dst = stage_jobs + (size_t)cohort * slice;
    for (int pass = 0; pass < 2; pass++) {
        const int cap = take ? (pass == 0 ? slice : (s_count > base_cap ? 0 : base_cap)) : 0;     // (s_count: what the urgent pass took; read after the barrier)
        for (int i = threadIdx.x; i < n_envs; i += blockDim.x) {
            const int e = (i + offset) % n_envs;
            const int X = la.cur[e];
            int h = la.handed[e];
            if (h < X) h = X;
            const int want = X + la.K - h;
            if (want <= 0 || (4 * want > la.K) != (pass == 0)) continue;
            atomicAdd(&s_want, want);
            if (cap <= 0) continue;
            const int at = atomicAdd(&s_count, want);
            int took = 0;
            for (int j = 0; j < want && at + j < cap; j++) { dst[at + j] = make_int2(e, h + 1 + j); took++; }
            if (took) la.handed[e] = h + took;
        }
        __syncthreads();
        if (threadIdx.x == 0 && pass == 0) { if (s_count > slice) s_count = slice; s_urgent = s_count; }    // (requests beyond the slice were not served)
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int n = s_count;
        const int lim = s_urgent > base_cap ? s_urgent : base_cap;       // the second pass stopped at base_cap, unless the first had gone beyond it already
        if (n > lim) n = lim;
        if (take) ctl[CTL_R + cohort] = (unsigned long long)n;
        ctl[CTL_BACKLOG] = (unsigned long long)(s_want - (take ? n : 0));
    }
}
// Settle lanes switched off (qs_settle_lanes(h, 0)): the settles in progress are dropped; their states count as not handed out again
__global__ void k_lookahead_requeue(unsigned long long* __restrict__ ctl, LookAhead la, const int2* __restrict__ stage_jobs, int slice) {
    for (int c = 0; c < QS_COHORTS; c++) {
        const int n = (int)ctl[CTL_R + c];
        for (int i = threadIdx.x; i < n; i += blockDim.x) { const int2 job = stage_jobs[(size_t)c * slice + i]; atomicMin(&la.handed[job.x], job.y - 1); }
    }
    __syncthreads();
    if (threadIdx.x == 0) for (int c = 0; c < QS_COHORTS; c++) ctl[CTL_R + c] = 0;
}

// QuadrupedGymEnv.reset for the masked environments (gym_env.py:278-297).  An environment whose look-ahead slot holds the coming episode
// takes it; the others settle side by side.
template <bool CONE> __global__ __launch_bounds__(QS_WAVE, 1) void k_reset(const qs_config* __restrict__ cfgp, float* __restrict__ recs,
                                                      const uint8_t* __restrict__ mask, float* __restrict__ obs_keep,
                                                      unsigned long long* __restrict__ stats, const float* __restrict__ states, LookAhead la) {
    using E = Env<LaneDev, CONE>;
    __shared__ __attribute__((aligned(16))) float s_rec[QS_TILE_FLOATS];
    __shared__ __attribute__((aligned(16))) float s_obs[QS_ENVS_PER_WAVE * QS_MAX_OBS];
    const qs_config& cfg = *cfgp;
    const int first = blockIdx.x * QS_ENVS_PER_WAVE;
    const int slot = threadIdx.x >> 2, env = first + slot;
    const bool valid = env < cfg.n_envs;
    const bool sel = valid && (mask == nullptr || mask[env] != 0);
    if (!__any(sel)) return;
    tile_load(s_rec, recs, first, cfg.n_envs, QS_REC_END);   // the whole record: what comes back is written as a whole
    const int od = cfg.obs_dim;
    __syncthreads();
    float* rec = s_rec + slot * QS_REC_END;
    float* ob = s_obs + slot * QS_MAX_OBS;
    const uint32_t gid = (uint32_t)((valid ? env : 0) + cfg.env_id_offset);
    bool ahead = false;
    if (sel) {
        if (states == nullptr) ahead = lookahead_take(la, stats, rec, env, qs::f2i(rec[R_EPISODE]) + 1, cfg.payload_soft != 0);
        else if (la.K > 0 && (threadIdx.x & 3) == 0) la.cur[env] = qs::f2i(rec[R_EPISODE]) + 1;   // (the look-ahead window moves on all the same)
        if ((threadIdx.x & 3) == 0) atomicAdd(&stats[CTL_RESETS], 1ull);
    }
    bool write_back = false;
    if (states) {   // reference-state initialisation (gym_env.py:278-297 with robot_desired_state set): randomizers, then the given
                    // rigid-body state instead of spawn + settle, then the task / sensor / filter reset of every reset
        if (sel) {
            E::randomize(cfg, rec, gid, qs::f2i(rec[R_EPISODE]) + 1, false);
            for (int i = threadIdx.x & 3; i < 37; i += 4) rec[R_POS + i] = states[(size_t)env * 37 + i];
            for (int i = threadIdx.x & 3; i < 4; i += 4) rec[R_WARM + i] = 0.0f;
            if ((threadIdx.x & 3) == 0) {
                for (int i = 0; i < 4; i++) { rec[R_FOOT_FORCE + i] = 0.0f; rec[R_FOOT_CONTACT + i] = 0.0f; }
                rec[R_N_INVALID] = 0.0f;
                for (int i = 0; i < 24; i++) rec[R_TAU_PD + i] = 0.0f;
            }
        }
        LaneDev::sync();
        if (cfg.payload_soft) { E::place_block(cfg, rec); LaneDev::sync(); }
        E::reset(cfg, rec, ob, gid, false);
        LaneDev::sync();
        if (sel)   // no settle ran, so _last_action and with it the filter history stay zero (gym_env.py:284, 267-269)
            for (int i = threadIdx.x & 3; i < 12 + 24 + 24; i += 4) rec[R_LAST_ACTION + i] = 0.0f;
        write_back = sel;
    } else {
        LaneDev::sync();
        if (ahead) {
            E::reset(cfg, rec, ob, gid, false);
            float* g = recs + (size_t)env * QS_REC;
            for (int i = threadIdx.x & 3; i < QS_REC_END; i += 4) g[i] = rec[i];
            for (int i = threadIdx.x & 3; i < od; i += 4) obs_keep[(size_t)env * od + i] = ob[i];
        }
        const bool settle = sel && !ahead;
        if (__any(settle)) {
            // every quad of the wave runs the settle (identical control flow keeps the wave votes of the solver valid);
            // quads that do not need it work on their LDS copy and simply do not write it back
            __syncthreads();
            E::reset(cfg, rec, ob, gid, true);
            if (settle && (threadIdx.x & 3) == 0) atomicAdd(&stats[CTL_SETTLE_SUBSTEPS], (unsigned long long)cfg.settle_steps);
        }
        write_back = settle;
    }
    __syncthreads();
    if (write_back) {
        float* g = recs + (size_t)env * QS_REC;
        for (int i = threadIdx.x & 3; i < QS_REC_END; i += 4) g[i] = rec[i];
        for (int i = threadIdx.x & 3; i < od; i += 4) obs_keep[(size_t)env * od + i] = ob[i];
    }
}

// The look-ahead slots at qs_create: entry j = the reset of environment j % N to episode j / N (episodes 0 .. K - 1), all side by side.
template <bool CONE> __global__ __launch_bounds__(QS_WAVE, 1) void k_lookahead_fill(const qs_config* __restrict__ cfgp, LookAhead la) {
    using E = Env<LaneDev, CONE>;
    __shared__ __attribute__((aligned(16))) float s_rec[QS_TILE_FLOATS];
    __shared__ __attribute__((aligned(16))) float s_obs[QS_ENVS_PER_WAVE * QS_MAX_OBS];
    const qs_config& cfg = *cfgp;
    const int slot = threadIdx.x >> 2;
    const long long j = (long long)blockIdx.x * QS_ENVS_PER_WAVE + slot, total = (long long)cfg.n_envs * la.K;
    const int env = (int)((j < total ? j : 0) % cfg.n_envs), episode = (int)((j < total ? j : 0) / cfg.n_envs);
    float* rec = s_rec + slot * QS_REC_END;
    for (int i = threadIdx.x & 3; i < QS_REC_END; i += 4) rec[i] = 0.0f;
    LaneDev::sync();
    if ((threadIdx.x & 3) == 0) rec[R_EPISODE] = qs::i2f(episode - 1);
    LaneDev::sync();
    E::reset(cfg, rec, s_obs + slot * QS_MAX_OBS, (uint32_t)(env + cfg.env_id_offset), true);
    LaneDev::sync();
    if (j < total) {
        float* g = la.slots + ((size_t)env * la.K + (size_t)(episode % la.K)) * QS_REC;
        for (int i = threadIdx.x & 3; i < QS_REC_END; i += 4) g[i] = rec[i];     // (R_EPISODE = episode: the slot's tag)
        if (episode == la.K - 1 && (threadIdx.x & 3) == 0) { la.handed[env] = la.K - 1; la.cur[env] = -1; }
    }
}

__global__ void k_gather(const float* __restrict__ recs, int n, int off, int dim, float* __restrict__ out, int as_int) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * dim) return;
    float v = recs[(size_t)(i / dim) * QS_REC + off + i % dim];
    out[i] = as_int ? (float)qs::f2i(v) : v;
}
__global__ void k_scatter(float* __restrict__ recs, int n, int off, int dim, const float* __restrict__ in, int zero_warm) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * dim) return;
    recs[(size_t)(i / dim) * QS_REC + off + i % dim] = in[i];
    if (zero_warm && i % dim < 4) recs[(size_t)(i / dim) * QS_REC + R_WARM + i % dim] = 0.0f;
}
__global__ void k_task_info(const float* __restrict__ recs, int n, float* __restrict__ out, int demo, int info) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float* r = recs + (size_t)e * QS_REC;
    float* o = out + (size_t)e * QS_TASK_DIM;
    for (int k = 0; k < T_N; k++) o[k] = r[R_TASK + k];
    for (int k = 0; k < 9; k++) o[T_N + k] = info ? r[R_POSE_CACHE + k] : 0.0f;   // the pose cache and the force sum come from the info block
    o[41] = r[R_N_INVALID];
    o[42] = info ? r[R_FOOT_FORCE] + r[R_FOOT_FORCE + 1] + r[R_FOOT_FORCE + 2] + r[R_FOOT_FORCE + 3] : 0.0f;
    o[43] = (float)qs::f2i(r[R_SIM_STEP]);
    o[44] = demo ? r[R_DEMO] : 0.0f; o[45] = demo ? r[R_DEMO + 1] : 0.0f;   // demo counter, and at the start of the episode
    for (int k = 46; k < QS_TASK_DIM; k++) o[k] = 0.0f;
}

#ifdef QS_ISA_ONLY
// tools/isa_headline.py: device code of the named step kernels only (seconds instead of minutes), for reading the ISA
#if defined(QS_ISA_PYRAMID)
template __global__ void k_step<false, false>(QS_STEP_ARGS);
#elif !defined(QS_ISA_DENSE)
template __global__ void k_step<true, false>(QS_STEP_ARGS);
#else
template __global__ void k_step_dense<true, false>(QS_STEP_ARGS);
#endif
#else
// ------------------------------------------------------------------ host side of the C ABI
thread_local char qs_g_err[512] = "";   // shared with qs_norm.hip
#define g_err qs_g_err
#define QS_FAIL(code, ...) do { snprintf(g_err, sizeof(g_err), __VA_ARGS__); return (code); } while (0)
#define QS_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) QS_FAIL(-2, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)

struct HostPath;
struct qs_handle {
    qs_config cfg;
    int device;
    hipStream_t stream;
    qs_config* d_cfg;
    float* d_rec;
    float* d_obs;       // last observation of every environment [N, obs_dim]
    float* d_term_obs;  // [N, obs_dim]
    LookAhead la;       // look-ahead reset states (cfg.reset_lookahead)
    float* d_staging;   // records being settled by the extra workgroups of k_step (settle lanes), QS_COHORTS slices
    int2* d_stage_jobs; // whose reset each staging record is
    int slice;          // staging records per cohort
    int lanes_on;
    long long tick;         // qs_step launches since the settle lanes were switched on
    float* trace_rows; int trace_env;
    float* d_demo; int demo_len;   // qs_set_demo
    int n_simd, step_variant;   // SIMDs of the device; 0 = pick k_step / k_step_dense by grid size, 1 / 2 = forced (QS_STEP_VARIANT)
    unsigned long long* d_stats;
    TermTail tail;          // set for the launch of a host-path step
    struct HostPath* host;  // qs_host_step_*: pinned buffers and the device result block (allocated at first use)
    hipEvent_t ev0, ev1;
    int timing;             // qs_enable_timing: 1 = armed (the next launch records ev0), 2 = ev0 recorded
    long long timed_launches;
};

static int sensor_dim(int s) {
    switch (s) {
    case QS_SENS_JOINT_POS: case QS_SENS_JOINT_VEL: case QS_SENS_FEET_POS: case QS_SENS_FEET_VEL: return 12;
    case QS_SENS_BOOL_CONTACT: case QS_SENS_QUAT: return 4;
    case QS_SENS_LIN_VEL: case QS_SENS_ANG_VEL: case QS_SENS_RPY: return 3;
    default: return 1;
    }
}
static int n_waves(int n) { return (n + QS_ENVS_PER_WAVE - 1) / QS_ENVS_PER_WAVE; }

#define QS_STR2(x) #x
#define QS_STR(x) QS_STR2(x)

extern "C" {

const char* qs_last_error(void) { return g_err; }
// build.py passes the fingerprint of the source tree the library is compiled from (every file under csrc/, the public header, the build
// script with its flags): a binary says itself which sources it is, and the parity gate (tools/gate.sh) records what it validated
#ifndef QS_SOURCE_SHA
#define QS_SOURCE_SHA "unknown"
#endif
const char* qs_version(void) { return "qs_amd 0.6 (gfx950, quad-per-env; ABI " QS_STR(QS_ABI_VERSION) "; source " QS_SOURCE_SHA ")"; }
int qs_abi_version(void) { return QS_ABI_VERSION; }

static int create_impl(const qs_config* cfg, int device, qs_handle* h);

int qs_create(const qs_config* cfg, int device, qs_handle** out) {
    if (!cfg || !out) QS_FAIL(-1, "null argument");
    if (cfg->n_envs <= 0) QS_FAIL(-1, "n_envs must be positive");
    if (cfg->n_envs > (1 << 24)) QS_FAIL(-1, "n_envs %d exceeds %d environments per handle (16 GB of records); shard over more handles", cfg->n_envs, 1 << 24);
    if (cfg->obs_dim <= 0 || cfg->obs_dim > QS_MAX_OBS || cfg->n_sensors > QS_MAX_SENSORS) QS_FAIL(-1, "observation bundle too large");
    if (cfg->action_dim != 12 && cfg->action_dim != 6 && cfg->action_dim != 4 && cfg->action_dim != 5) QS_FAIL(-1, "action_dim must be 12, 6, 4 or 5 (CPG)");
    if (cfg->wrapper_mode != QS_WRAP_NONE && (cfg->action_space_mode == QS_ACT_CPG || !cfg->rl_interface))
        QS_FAIL(-1, "the landing / go-to-rest phase machine needs an RL action space (not CPG, not raw commands)");
    if (cfg->task >= QS_TASK_JUMPING_IN_PLACE_DEMO && cfg->task <= QS_TASK_CONT_JUMPING_FORWARD_DEMO && (cfg->action_space_mode == QS_ACT_CPG || !cfg->rl_interface))
        QS_FAIL(-1, "the DEMO tasks compare the policy's action with a recorded one: they need an RL action space (not CPG, not raw commands)");
    if (cfg->task < 0 || cfg->task > QS_TASK_CONT_JUMPING_FORWARD_DEMO) QS_FAIL(-1, "unknown task id %d", cfg->task);
    if (cfg->friction_cone != 0 && cfg->friction_cone != 1) QS_FAIL(-1, "friction_cone must be 0 (pyramid) or 1 (implicit cone), got %d", cfg->friction_cone);
    if (cfg->reset_lookahead < 0 || cfg->reset_lookahead > 64) QS_FAIL(-1, "reset_lookahead must be between 0 (settle inside the step) and 64 reset states per environment, got %d", cfg->reset_lookahead);
    if (cfg->motor_control_mode == QS_MOTOR_TORQUE && cfg->rl_interface)  // gym_env.py:167-168
        QS_FAIL(-1, "the motor control mode TORQUE not implemented yet for RL Gym interface.");
    int od = 0;
    for (int i = 0; i < cfg->n_sensors; i++) od += sensor_dim(cfg->sensors[i]);
    if (od != cfg->obs_dim) QS_FAIL(-1, "obs_dim %d does not match the sensor bundle (%d)", cfg->obs_dim, od);
    int ndev = 0;
    hipError_t derr = hipGetDeviceCount(&ndev);
    if (derr != hipSuccess || ndev <= 0)
        QS_FAIL(-3, "no HIP device available (hipGetDeviceCount: %s, %d devices): this library has no CPU path", hipGetErrorString(derr), ndev);
    if (device < 0 || device >= ndev) QS_FAIL(-3, "HIP device %d out of range (%d visible)", device, ndev);
    DeviceGuard guard(device);
    qs_handle* h = new (std::nothrow) qs_handle();
    if (!h) QS_FAIL(-4, "out of host memory");
    memset(h, 0, sizeof(*h));
    int rc = create_impl(cfg, device, h);
    if (rc != 0) { qs_destroy(h); return rc; }   // frees whatever was allocated before the failure (the error text is kept)
    *out = h;
    return 0;
}

static int create_impl(const qs_config* cfg, int device, qs_handle* h) {
    h->cfg = *cfg; h->device = device; h->stream = nullptr;
    {
        hipDeviceProp_t prop;
        QS_HIP(hipGetDeviceProperties(&prop, device));
        h->n_simd = 4 * prop.multiProcessorCount;
        const char* v = getenv("QS_STEP_VARIANT");
        h->step_variant = v ? atoi(v) : 0;
    }
    const size_t n = (size_t)cfg->n_envs;
    QS_HIP(hipMalloc(&h->d_cfg, sizeof(QsDevCfg)));   // the configuration, and behind it the address of the rare paths' counters (qs_lane.h)
    QS_HIP(hipMalloc(&h->d_rec, n * QS_REC * sizeof(float)));
    QS_HIP(hipMalloc(&h->d_obs, n * cfg->obs_dim * sizeof(float)));
    QS_HIP(hipMalloc(&h->d_term_obs, n * cfg->obs_dim * sizeof(float)));
    QS_HIP(hipMalloc(&h->d_stats, CTL_N * sizeof(unsigned long long)));
    {
        QsDevCfg dc;
        dc.cfg = h->cfg; dc.counters = h->d_stats + CTL_DEV;
        QS_HIP(hipMemcpy(h->d_cfg, &dc, sizeof(dc), hipMemcpyHostToDevice));
    }
    QS_HIP(hipMemset(h->d_obs, 0, n * cfg->obs_dim * sizeof(float)));
    QS_HIP(hipMemset(h->d_term_obs, 0, n * cfg->obs_dim * sizeof(float)));
    QS_HIP(hipMemset(h->d_stats, 0, CTL_N * sizeof(unsigned long long)));
    QS_HIP(hipEventCreate(&h->ev0));
    QS_HIP(hipEventCreate(&h->ev1));
    hipLaunchKernelGGL(k_init, dim3(n_waves(cfg->n_envs)), dim3(QS_WAVE), 0, h->stream, h->d_cfg, h->d_rec);
    QS_HIP(hipGetLastError());
    // look-ahead reset states: K slots per environment, filled for episodes 0 .. K - 1 before anything steps; not under QS_RAND_KEEP, where
    // a reset's parameters are whatever qs_set_params wrote last (the settle cannot be computed ahead of that)
    memset(&h->la, 0, sizeof(h->la));
    if (cfg->reset_lookahead > 0 && !(cfg->randomizer_flags & QS_RAND_KEEP)) {
        const int K = cfg->reset_lookahead;
        h->la.K = K;
        // a cohort settles at most `slice` records at a time: 2 N -- with five cohorts per epoch of 250 launches that is room for every
        // environment to fall every 25 steps (the start of a training run); a launch only carries the lanes the backlog asks for
        h->slice = (int)((2 * n + QS_ENVS_PER_WAVE - 1) / QS_ENVS_PER_WAVE * QS_ENVS_PER_WAVE);
        if (h->slice < QS_MAX_SLICE) h->slice = QS_MAX_SLICE;
        if (h->slice > 131072) h->slice = 131072;
        QS_HIP(hipMalloc(&h->la.slots, n * K * QS_REC * sizeof(float)));
        QS_HIP(hipMalloc(&h->la.cur, n * sizeof(int)));
        QS_HIP(hipMalloc(&h->la.handed, n * sizeof(int)));
        QS_HIP(hipMalloc(&h->d_staging, (size_t)QS_COHORTS * h->slice * QS_REC * sizeof(float)));
        QS_HIP(hipMalloc(&h->d_stage_jobs, (size_t)QS_COHORTS * h->slice * sizeof(int2)));
        QS_HIP(hipMemsetAsync(h->d_staging, 0, (size_t)QS_COHORTS * h->slice * QS_REC * sizeof(float), h->stream));
        QS_HIP(hipMemsetAsync(h->d_stage_jobs, 0, (size_t)QS_COHORTS * h->slice * sizeof(int2), h->stream));

        const unsigned fill_grid = (unsigned)((n * K + QS_ENVS_PER_WAVE - 1) / QS_ENVS_PER_WAVE);
        if (cfg->friction_cone) hipLaunchKernelGGL((k_lookahead_fill<true>), dim3(fill_grid), dim3(QS_WAVE), 0, h->stream, h->d_cfg, h->la);
        else hipLaunchKernelGGL((k_lookahead_fill<false>), dim3(fill_grid), dim3(QS_WAVE), 0, h->stream, h->d_cfg, h->la);
        QS_HIP(hipGetLastError());
        h->lanes_on = 1;
    }
    QS_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

static void host_path_free(qs_handle* h);
void qs_destroy(qs_handle* h) {   // also used on a partially built handle (null members are skipped)
    if (!h) return;
    QS_ON_DEVICE(h);
    hipStreamSynchronize(h->stream);
    hipFree(h->d_cfg); hipFree(h->d_rec); hipFree(h->d_obs); hipFree(h->d_term_obs); hipFree(h->d_stats);
    if (h->la.slots) hipFree(h->la.slots);
    if (h->la.cur) hipFree(h->la.cur);
    if (h->la.handed) hipFree(h->la.handed);
    if (h->d_staging) hipFree(h->d_staging);
    if (h->d_stage_jobs) hipFree(h->d_stage_jobs);
    if (h->d_demo) hipFree(h->d_demo);
    host_path_free(h);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    delete h;
}

#if defined(QS_PROBE_LAZY) || defined(QS_PROBE_SWEEPS) || defined(QS_PROBE_WARM)
extern "C" int qs_probe_counters(qs_handle* h, unsigned long long* out10) {   // counting builds only (tools/probe_*.py)
    QS_ON_DEVICE(h);
    QS_HIP(hipStreamSynchronize(h->stream));
    QS_HIP(hipMemcpy(out10, h->d_stats + CTL_DEV + 2, 10 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}
#endif
int qs_set_stream(qs_handle* h, void* s) { if (!h) QS_FAIL(-1, "null handle"); h->stream = (hipStream_t)s; return 0; }
int qs_enable_timing(qs_handle* h, int on) {
    if (!h) QS_FAIL(-1, "null handle");
    if (on == 2) {   // close the batch: the closing event goes onto the stream now, qs_last_step_kernel_ms waits for it later
        if (h->timing != 2 || h->timed_launches <= 0) QS_FAIL(-1, "no step has been launched since qs_enable_timing(h, 1)");
        QS_ON_DEVICE(h);
        QS_HIP(hipEventRecord(h->ev1, h->stream));
        h->timing = 3;
        return 0;
    }
    h->timing = on ? 1 : 0; h->timed_launches = 0;
    return 0;
}

int qs_reset(qs_handle* h, const uint8_t* mask) {
    if (!h) QS_FAIL(-1, "null handle");
    QS_ON_DEVICE(h);
    if (h->cfg.friction_cone) hipLaunchKernelGGL((k_reset<true>), dim3(n_waves(h->cfg.n_envs)), dim3(QS_WAVE), 0, h->stream, h->d_cfg, h->d_rec, mask, h->d_obs, h->d_stats, (const float*)nullptr, h->la);
    else hipLaunchKernelGGL((k_reset<false>), dim3(n_waves(h->cfg.n_envs)), dim3(QS_WAVE), 0, h->stream, h->d_cfg, h->d_rec, mask, h->d_obs, h->d_stats, (const float*)nullptr, h->la);
    QS_HIP(hipGetLastError());
    return 0;
}

int qs_reset_to(qs_handle* h, const uint8_t* mask, const float* states) {
    if (!h || !states) QS_FAIL(-1, "null argument");
    QS_ON_DEVICE(h);
    if (h->cfg.friction_cone) hipLaunchKernelGGL((k_reset<true>), dim3(n_waves(h->cfg.n_envs)), dim3(QS_WAVE), 0, h->stream, h->d_cfg, h->d_rec, mask, h->d_obs, h->d_stats, states, h->la);
    else hipLaunchKernelGGL((k_reset<false>), dim3(n_waves(h->cfg.n_envs)), dim3(QS_WAVE), 0, h->stream, h->d_cfg, h->d_rec, mask, h->d_obs, h->d_stats, states, h->la);
    QS_HIP(hipGetLastError());
    return 0;
}

int qs_get_obs(qs_handle* h, float* obs) {
    if (!h || !obs) QS_FAIL(-1, "null argument");
    QS_ON_DEVICE(h);
    QS_HIP(hipMemcpyAsync(obs, h->d_obs, (size_t)h->cfg.n_envs * h->cfg.obs_dim * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    return 0;
}

int qs_settle_lanes(qs_handle* h, int on) {
    if (!h) QS_FAIL(-1, "null handle");
    QS_ON_DEVICE(h);
    if (h->la.K == 0) { if (on) QS_FAIL(-1, "this handle keeps no look-ahead reset states (cfg.reset_lookahead = 0, or QS_RAND_KEEP)"); return 0; }
    if (!on && h->lanes_on) {   // settles in progress are dropped, their jobs go back to the front of the queue
        hipLaunchKernelGGL(k_lookahead_requeue, dim3(1), dim3(256), 0, h->stream, h->d_stats, h->la, h->d_stage_jobs, h->slice);
        QS_HIP(hipGetLastError());
    }
    if (on && !h->lanes_on) h->tick = 0;
    h->lanes_on = on ? 1 : 0;
    return 0;
}

static int launch_step(qs_handle* h, const float* actions, float* obs, float* rew, uint8_t* done, uint8_t* trunc);

int qs_step(qs_handle* h, const float* actions, float* obs, float* rew, uint8_t* done, uint8_t* trunc) {
    if (!h || !actions || !obs || !rew || !done || !trunc) QS_FAIL(-1, "null argument");
    return launch_step(h, actions, obs, rew, done, trunc);
}

int qs_step_fused(qs_handle* h, const float* actions, float* fused) {
    if (!h || !actions || !fused) QS_FAIL(-1, "null argument");
    return launch_step(h, actions, fused, nullptr, nullptr, nullptr);   // a null reward pointer selects the fused row layout in the kernel
}

static int launch_step(qs_handle* h, const float* actions, float* obs, float* rew, uint8_t* done, uint8_t* trunc) {
    QS_ON_DEVICE(h);
    SettleLanes lanes;
    memset(&lanes, 0, sizeof(lanes));
    lanes.n_env_waves = n_waves(h->cfg.n_envs); lanes.slice = 0;
    TraceTap tap; tap.rows = h->trace_rows; tap.env = h->trace_env;
    DemoTab demo; demo.rows = h->d_demo; demo.length = h->demo_len;
    if (E::demo_task(h->cfg.task) && !h->d_demo) QS_FAIL(-1, "the DEMO tasks need a demonstration: qs_set_demo first");
    TermTail tail = h->tail;
    memset(&h->tail, 0, sizeof(h->tail));       // (set by qs_host_step_begin for its own launch only)
    int grid = lanes.n_env_waves;
    bool lanes_fit = true;      // the environments' waves and the lanes' usual share fit the SIMDs
    if (h->la.K > 0 && h->lanes_on) {
        // one settle = settle_steps substeps = `epoch` launches of action_repeat substeps (the last one takes the remainder);
        // cohort c runs the same schedule c * epoch / QS_COHORTS launches later
        const int rep = h->cfg.action_repeat, epoch = (h->cfg.settle_steps + rep - 1) / rep;
        const int slice = h->slice;
        lanes.staging = h->d_staging; lanes.stage_jobs = h->d_stage_jobs; lanes.slice = slice;
        // Every launch carries the lanes of five full cohorts (a workgroup beyond its cohort's jobs leaves at once: 5120 of them at
        // N = 8192 cost the launch 0.4 %); how many of them work is decided on the device (k_lookahead_plan): normally what fits the SIMDs
        // the environments leave idle (base_waves per cohort: the one-wave-per-SIMD kernel then runs the launch in one round), more
        // when environments run short of states.
        const int free_simd = h->n_simd - lanes.n_env_waves;
        const int cohort_waves = slice / QS_ENVS_PER_WAVE;
        int base_waves = cohort_waves;
        if (h->step_variant != 2 && free_simd / QS_COHORTS >= 64 && base_waves > free_simd / QS_COHORTS) base_waves = free_simd / QS_COHORTS;
        lanes_fit = n_waves(h->cfg.n_envs) + QS_COHORTS * base_waves <= h->n_simd;
        for (int c = 0; c < QS_COHORTS; c++) {
            const long long t = h->tick - (long long)c * epoch / QS_COHORTS;
            lanes.wave0[c] = c * cohort_waves;
            if (t < 0) continue;                       // not started yet: settle_n stays 0
            const int phase = (int)(t % epoch);
            if (phase == 0) {
                hipLaunchKernelGGL(k_lookahead_publish, dim3(128), dim3(256), 0, h->stream, h->d_stats, h->d_staging, h->d_stage_jobs, h->la, c, slice);
                hipLaunchKernelGGL(k_lookahead_plan, dim3(1), dim3(1024), 0, h->stream, h->d_stats, h->la, h->cfg.n_envs, h->d_stage_jobs, c, slice, base_waves * QS_ENVS_PER_WAVE, 1,
                                   (int)((h->tick / (epoch / QS_COHORTS > 0 ? epoch / QS_COHORTS : 1)) * 4099 % h->cfg.n_envs));
            }
            lanes.spawn[c] = phase == 0; lanes.last[c] = phase == epoch - 1;
            lanes.settle_n[c] = phase == epoch - 1 ? h->cfg.settle_steps - rep * (epoch - 1) : rep;
        }
        lanes.wave0[QS_COHORTS] = QS_COHORTS * cohort_waves;
        grid += lanes.wave0[QS_COHORTS];
        h->tick++;
    }
    if (h->timing == 1) { hipEventRecord(h->ev0, h->stream); h->timing = 2; }   // (after this step's publish / plan launches, if any)
    if (h->timing == 1 || h->timing == 2) h->timed_launches++;
    // more waves than SIMDs: the two-waves-per-SIMD build of the same body (see k_step_dense) instead of a second round of one-wave-per-
    // SIMD workgroups (N = 12288 with its settle lanes: 0.109 ms in two rounds)
    const bool dense = h->step_variant == 2 || (h->step_variant == 0 && (lanes.n_env_waves > h->n_simd || !lanes_fit));
    const size_t lds = (size_t)QS_ENVS_PER_WAVE * ((h->cfg.payload_soft ? QS_REC_END : QS_INFO_END) + QS_MAX_OBS + 12) * sizeof(float);
#define QS_LAUNCH_STEP(KERNEL) hipLaunchKernelGGL((KERNEL), dim3(grid), dim3(QS_WAVE), lds, h->stream, h->d_cfg, h->d_rec, actions, obs, rew, done, trunc, \
                                                 h->d_obs, h->d_term_obs, h->la, h->d_stats, lanes, tap, demo, tail)
#define QS_PICK(C, S) { if (dense) QS_LAUNCH_STEP((k_step_dense<C, S>)); else QS_LAUNCH_STEP((k_step<C, S>)); }
    if (h->cfg.friction_cone && h->cfg.payload_soft) QS_PICK(true, true)
    else if (h->cfg.friction_cone) QS_PICK(true, false)
    else QS_PICK(false, false)
#undef QS_PICK
#undef QS_LAUNCH_STEP
    QS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ host (numpy) path: VecEnv.step_async / step_wait on HOST arrays
// load_model.py:113-133 drives the environment with numpy arrays.  One step of that path = copy of the actions into pinned staging +
// H2D, the step kernel writing observations / rewards / flags into ONE device block (with the step's terminal observations as a
// compact list behind them), ONE D2H copy of that block into page-locked ordinary host memory (hipHostRegister: the DMA engine writes it,
// the host reads it at cache speed -- hipHostMalloc'ed memory read three times slower on the test box, tools/host_copy_rate.py).  Two
// host blocks alternate, so the arrays of one step stay valid during the next.
struct HostPath {
    size_t bytes, off_rew, off_done, off_trunc, off_tail;
    int cap;                 // rows of the compact terminal list
    int zero_copy;           // the step kernel reads the actions from / writes the results to the page-locked host memory itself
    uint8_t* d_block;        // (copy mode) the device result block
    uint8_t* h_block[2];
    uint8_t* hd_block[2];    // (zero-copy mode) device addresses of the host blocks
    float* h_act; float* d_act; float* hd_act;
    int cur, parity, pending;
    int late_rc;            // a failure behind the launch of the pending step (qs_host_step_begin), reported again by qs_host_step_end
    int registered;          // bits 0, 1: h_block[k], bit 2: h_act are page-locked (hipHostRegister succeeded)
    hipEvent_t ev;
    // qs_host_set_norm: VecNormalize.step_wait between the step and the results' way to the host (the step then writes the DEVICE block,
    // qs_norm_step_io reads it and writes the normalised arrays, the flags and the list into the mapped host block -- or, under
    // QS_HOST_PATH=copy, works in place and one copy brings the block over)
    qs_norm* norm; int norm_training, norm_obs, norm_rew; float* raw_obs; float* raw_rew;
};
// (`registered`: which of the page-locked buffers hipHostRegister accepted -- only those are unregistered)
static void host_path_release(HostPath* p) {
    if (!p) return;
    for (int k = 0; k < 2; k++) if (p->h_block[k]) { if (p->registered & (1 << k)) hipHostUnregister(p->h_block[k]); free(p->h_block[k]); }
    if (p->h_act) { if (p->registered & 4) hipHostUnregister(p->h_act); free(p->h_act); }
    if (p->d_block) hipFree(p->d_block);
    if (p->d_act) hipFree(p->d_act);
    if (p->ev) hipEventDestroy(p->ev);
    delete p;
}
static void host_path_free(qs_handle* h) { host_path_release(h->host); h->host = nullptr; }
static int host_path_build(qs_handle* h, HostPath* p);
static int host_path_init(qs_handle* h) {
    if (h->host) return 0;
    HostPath* p = new (std::nothrow) HostPath();
    if (!p) QS_FAIL(-4, "out of host memory");
    memset(p, 0, sizeof(*p));
    // the handle gets the path only once every buffer stands: a failed allocation / registration leaves no half-built one behind
    if (int rc = host_path_build(h, p)) { host_path_release(p); return rc; }
    h->host = p;
    return 0;
}
static int host_path_build(qs_handle* h, HostPath* p) {
    const size_t n = (size_t)h->cfg.n_envs, o = (size_t)h->cfg.obs_dim, d = (size_t)h->cfg.action_dim;
    p->cap = (int)(n < 256 ? n : 256);
    p->off_rew = n * o * 4; p->off_done = p->off_rew + n * 4; p->off_trunc = p->off_done + n;
    p->off_tail = (p->off_trunc + n + 15) / 16 * 16;
    p->bytes = p->off_tail + (size_t)p->cap * (o + 1) * 4;
    // QS_HOST_PATH=copy: H2D copy of the actions and D2H copy of a device result block around the step (two DMA operations on the stream);
    // default: zero copy -- the page-locked host buffers are mapped into the device's address space and the kernel reads its 24 B of
    // actions per environment and writes its rows over PCIe itself, the waves that finish first while the others still compute
    const char* mode = getenv("QS_HOST_PATH");
    p->zero_copy = !(mode && strcmp(mode, "copy") == 0);
    const unsigned flags = p->zero_copy ? hipHostRegisterMapped : hipHostRegisterDefault;
    const size_t page = 4096, hb = (p->bytes + page - 1) / page * page, ab = (n * d * 4 + page - 1) / page * page;
    for (int k = 0; k < 2; k++) {
        void* m = nullptr;
        if (posix_memalign(&m, page, hb) != 0) QS_FAIL(-4, "out of host memory");
        memset(m, 0, hb);
        p->h_block[k] = (uint8_t*)m;
        QS_HIP(hipHostRegister(m, hb, flags));
        p->registered |= 1 << k;
        if (p->zero_copy) QS_HIP(hipHostGetDevicePointer((void**)&p->hd_block[k], m, 0));
    }
    void* m = nullptr;
    if (posix_memalign(&m, page, ab) != 0) QS_FAIL(-4, "out of host memory");
    memset(m, 0, ab);
    p->h_act = (float*)m;
    QS_HIP(hipHostRegister(m, ab, flags));
    p->registered |= 4;
    if (p->zero_copy) QS_HIP(hipHostGetDevicePointer((void**)&p->hd_act, m, 0));
    else {
        QS_HIP(hipMalloc(&p->d_block, p->bytes));
        QS_HIP(hipMemset(p->d_block, 0, p->bytes));
        QS_HIP(hipMalloc(&p->d_act, n * d * 4));
    }
    QS_HIP(hipEventCreateWithFlags(&p->ev, hipEventDisableTiming));
    return 0;
}

// what qs_host_step_begin enqueues behind the step's launch: the normalisation attached by qs_host_set_norm, the copy of the result block
static int host_step_after_launch(qs_handle* h, HostPath* p, uint8_t* blk, bool via_device) {
    if (p->norm) {
        if (int rc = qs_norm_set_stream(p->norm, (void*)h->stream)) return rc;
        qs_norm_io io;
        memset(&io, 0, sizeof(io));
        io.obs = (float*)blk; io.rew = (float*)(blk + p->off_rew); io.done = blk + p->off_done; io.trunc = blk + p->off_trunc;
        io.tail_rows = (float*)(blk + p->off_tail); io.tail_cap = p->cap; io.raw_obs = p->raw_obs; io.raw_rew = p->raw_rew;
        io.tail_count = (const uint64_t*)&h->d_stats[CTL_TERM_CNT + p->parity];      // the rows this step's launch claimed
        if (p->zero_copy) {     // the normalising kernel writes the host block itself
            uint8_t* to = p->hd_block[p->cur];
            io.out_obs = (float*)to; io.out_rew = (float*)(to + p->off_rew); io.out_done = to + p->off_done; io.out_trunc = to + p->off_trunc;
            io.out_tail = (float*)(to + p->off_tail);
        }
        if (int rc = qs_norm_step_io(p->norm, &io, p->norm_training, p->norm_obs, p->norm_rew)) return rc;
    }
    if (via_device && !(p->norm && p->zero_copy)) QS_HIP(hipMemcpyAsync(p->h_block[p->cur], p->d_block, p->bytes, hipMemcpyDeviceToHost, h->stream));
    return 0;
}

int qs_host_step_begin(qs_handle* h, const float* actions_host) {
    if (!h || !actions_host) QS_FAIL(-1, "null argument");
    QS_ON_DEVICE(h);
    if (int rc = host_path_init(h)) return rc;
    HostPath* p = h->host;
    if (p->pending) QS_FAIL(-1, "qs_host_step_begin: the previous step has not been collected (qs_host_step_end)");
    const size_t n = (size_t)h->cfg.n_envs, d = (size_t)h->cfg.action_dim;
    memcpy(p->h_act, actions_host, n * d * 4);
    // the block and the terminal-list counter this step uses are the OTHER ones; the switch is made once the step is on the stream (a
    // launch that fails -- a DEMO task without its demonstration -- must not leave a counter behind that no step has cleared)
    const int cur = p->cur ^ 1, parity = p->parity ^ 1;
    const bool via_device = !p->zero_copy || p->norm != nullptr;     // the results pass through the device block
    if (via_device && !p->d_block) {
        QS_HIP(hipMalloc(&p->d_block, p->bytes));
        QS_HIP(hipMemsetAsync(p->d_block, 0, p->bytes, h->stream));
    }
    uint8_t* blk = via_device ? p->d_block : p->hd_block[cur];
    const float* act = p->zero_copy ? p->hd_act : p->d_act;
    if (!p->zero_copy) QS_HIP(hipMemcpyAsync(p->d_act, p->h_act, n * d * 4, hipMemcpyHostToDevice, h->stream));
    h->tail.rows = (float*)(blk + p->off_tail); h->tail.cap = p->cap; h->tail.parity = parity;
    if (int rc = launch_step(h, act, (float*)blk, (float*)(blk + p->off_rew), blk + p->off_done, blk + p->off_trunc)) { memset(&h->tail, 0, sizeof(h->tail)); return rc; }
    p->cur = cur; p->parity = parity;
    // From here on the simulation HAS advanced one step.  Whatever fails behind the launch -- the normalisation, the copy -- must not leave the
    // handle looking as if no step were under way (round 4 returned with pending = 0: the step's results could not be collected and
    // qs_host_step_end said "without begin"): the step stays pending, the failure is reported here AND by the qs_host_step_end that collects it.
    const int late = host_step_after_launch(h, p, blk, via_device);
    hipEventRecord(p->ev, h->stream);
    p->pending = 1; p->late_rc = late;
    return late;
}

int qs_host_set_norm(qs_handle* h, qs_norm* norm, int training, int norm_obs, int norm_reward, float* raw_obs, float* raw_rew) {
    if (!h) QS_FAIL(-1, "null handle");
    QS_ON_DEVICE(h);
    if (!norm && !h->host) return 0;     // nothing to switch off
    if (int rc = host_path_init(h)) return rc;
    HostPath* p = h->host;
    // (switching it off is always possible: a step already on the stream keeps using the statistics' buffers, and qs_norm_destroy waits
    // for its stream before it frees them)
    if (p->pending && norm) QS_FAIL(-1, "qs_host_set_norm between qs_host_step_begin and qs_host_step_end");
    if (norm) {   // everything that can be checked is checked HERE: a failure behind a step's launch costs that step's results (qs_host_step_begin)
        int nn = 0, no = 0, nd = -1;
        if (int rc = qs_norm_dims(norm, &nn, &no, &nd)) return rc;
        if (nn != h->cfg.n_envs || no != h->cfg.obs_dim || nd != h->device)
            QS_FAIL(-1, "qs_host_set_norm: the normalisation handle is for %d environments x %d observations on device %d, the simulation handle for %d x %d on device %d",
                    nn, no, nd, h->cfg.n_envs, h->cfg.obs_dim, h->device);
    }
    p->norm = norm; p->norm_training = training; p->norm_obs = norm_obs; p->norm_rew = norm_reward; p->raw_obs = raw_obs; p->raw_rew = raw_rew;
    return 0;
}

int qs_host_step_end(qs_handle* h, qs_host_result* out) {
    if (!h || !out) QS_FAIL(-1, "null argument");
    HostPath* p = h->host;
    if (!p || !p->pending) QS_FAIL(-1, "qs_host_step_end without qs_host_step_begin");
    QS_ON_DEVICE(h);
    // (polling the event with hipEventQuery before blocking on it was measured in round 4: the wait is 99.7 us instead of 100.7, for a host
    // core that spins -- not kept)
    QS_HIP(hipEventSynchronize(p->ev));
    p->pending = 0;
    if (p->late_rc) {
        const int rc = p->late_rc;
        p->late_rc = 0;
        QS_FAIL(rc, "qs_host_step_end: the step ran, but what qs_host_step_begin enqueued behind it (normalisation / copy) had failed with %d: the host block does not hold its results", rc);
    }
    const uint8_t* b = p->h_block[p->cur];
    out->obs = (const float*)b; out->rew = (const float*)(b + p->off_rew); out->done = b + p->off_done; out->truncated = b + p->off_trunc;
    out->terminal_rows = (const float*)(b + p->off_tail); out->terminal_cap = p->cap;
    return 0;
}

__global__ void k_set_demo_counter(float* __restrict__ recs, int n, const uint8_t* __restrict__ mask, const int32_t* __restrict__ values, int length) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n || (mask && !mask[e])) return;
    int v = values[e];
    v = v < 0 ? 0 : (v >= length ? length - 1 : v);
    float* r = recs + (size_t)e * QS_REC + R_DEMO;
    r[0] = (float)v; r[1] = (float)v;
}

int qs_set_demo(qs_handle* h, const float* rows, int length) {
    if (!h || !rows) QS_FAIL(-1, "null argument");
    if (!E::demo_task(h->cfg.task)) QS_FAIL(-1, "task %d is not a DEMO task", h->cfg.task);
    if (length <= 0 || length >= (1 << 24)) QS_FAIL(-1, "a demonstration has between 1 and 2^24 rows (got %d)", length);
    QS_ON_DEVICE(h);
    const size_t bytes = (size_t)length * (size_t)(h->cfg.action_dim + 38) * sizeof(float);
    float* copy = nullptr;
    QS_HIP(hipMalloc(&copy, bytes));
    hipError_t e = hipMemcpyAsync(copy, rows, bytes, hipMemcpyDeviceToDevice, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);   // also: no step still reads the table that is about to go
    if (e != hipSuccess) { hipFree(copy); QS_FAIL(-2, "copying the demonstration failed: %s", hipGetErrorString(e)); }
    if (h->d_demo) hipFree(h->d_demo);
    h->d_demo = copy; h->demo_len = length;
    return 0;
}

int qs_set_demo_counter(qs_handle* h, const uint8_t* mask, const int32_t* values) {
    if (!h || !values) QS_FAIL(-1, "null argument");
    if (!h->d_demo) QS_FAIL(-1, "no demonstration (qs_set_demo)");
    QS_ON_DEVICE(h);
    hipLaunchKernelGGL(k_set_demo_counter, dim3((h->cfg.n_envs + 255) / 256), dim3(256), 0, h->stream, h->d_rec, h->cfg.n_envs, mask, values, h->demo_len);
    QS_HIP(hipGetLastError());
    return 0;
}

int qs_set_trace(qs_handle* h, int env, float* rows) {
    if (!h) QS_FAIL(-1, "null handle");
    if (env >= h->cfg.n_envs) QS_FAIL(-1, "trace environment %d out of range (%d environments)", env, h->cfg.n_envs);
    h->trace_env = env; h->trace_rows = env >= 0 ? rows : nullptr;
    return 0;
}

int qs_counters_async(qs_handle* h, uint64_t* dev_out) {
    if (!h || !dev_out) QS_FAIL(-1, "null argument");
    QS_ON_DEVICE(h);
    static_assert(CTL_SETTLE_SUBSTEPS == QS_COUNTER_SETTLE_SUBSTEPS && CTL_RESETS == QS_COUNTER_RESETS && CTL_SERVED == QS_COUNTER_LOOKAHEAD_SERVED &&
                  CTL_SETTLED == QS_COUNTER_LOOKAHEAD_SETTLED && CTL_STALLS == QS_COUNTER_RESET_STALLS, "the handle's counters sit at their public indices");
    unsigned long long* out = reinterpret_cast<unsigned long long*>(dev_out);
    QS_HIP(hipMemcpyAsync(out, h->d_stats, 4 * sizeof(unsigned long long), hipMemcpyDeviceToDevice, h->stream));
    QS_HIP(hipMemcpyAsync(out + QS_COUNTER_RESET_STALLS, h->d_stats + CTL_STALLS, sizeof(unsigned long long), hipMemcpyDeviceToDevice, h->stream));
    QS_HIP(hipMemcpyAsync(out + QS_COUNTER_LIMIT_PATH_SUBSTEPS, h->d_stats + CTL_DEV + QS_DEVCTR_RARE_PATH, sizeof(unsigned long long), hipMemcpyDeviceToDevice, h->stream));
    QS_HIP(hipMemcpyAsync(out + QS_COUNTER_SELF_NARROW_SUBSTEPS, h->d_stats + CTL_DEV + QS_DEVCTR_SELF_NARROW, sizeof(unsigned long long), hipMemcpyDeviceToDevice, h->stream));
    return 0;
}

int qs_counter(qs_handle* h, int which, uint64_t* value) {
    if (!h || !value) QS_FAIL(-1, "null argument");
    QS_ON_DEVICE(h);
    QS_HIP(hipStreamSynchronize(h->stream));
    unsigned long long v = 0;
    switch (which) {
    case QS_COUNTER_SETTLE_SUBSTEPS: QS_HIP(hipMemcpy(&v, &h->d_stats[CTL_SETTLE_SUBSTEPS], sizeof(v), hipMemcpyDeviceToHost)); break;
    case QS_COUNTER_RESETS: QS_HIP(hipMemcpy(&v, &h->d_stats[CTL_RESETS], sizeof(v), hipMemcpyDeviceToHost)); break;
    case QS_COUNTER_LOOKAHEAD_SERVED: QS_HIP(hipMemcpy(&v, &h->d_stats[CTL_SERVED], sizeof(v), hipMemcpyDeviceToHost)); break;
    case QS_COUNTER_LOOKAHEAD_SETTLED: QS_HIP(hipMemcpy(&v, &h->d_stats[CTL_SETTLED], sizeof(v), hipMemcpyDeviceToHost)); break;
    case QS_COUNTER_RESET_STALLS: QS_HIP(hipMemcpy(&v, &h->d_stats[CTL_STALLS], sizeof(v), hipMemcpyDeviceToHost)); break;
    case QS_COUNTER_LOOKAHEAD_BACKLOG:
        if (h->la.K > 0) {   // a counting pass of the planning scan
            hipLaunchKernelGGL(k_lookahead_plan, dim3(1), dim3(1024), 0, h->stream, h->d_stats, h->la, h->cfg.n_envs, h->d_stage_jobs, 0, h->slice, 0, 0, 0);
            QS_HIP(hipGetLastError());
            QS_HIP(hipStreamSynchronize(h->stream));
            QS_HIP(hipMemcpy(&v, &h->d_stats[CTL_BACKLOG], sizeof(v), hipMemcpyDeviceToHost));
        }
        break;
    case QS_COUNTER_LIMIT_PATH_SUBSTEPS: QS_HIP(hipMemcpy(&v, &h->d_stats[CTL_DEV + QS_DEVCTR_RARE_PATH], sizeof(v), hipMemcpyDeviceToHost)); break;
    case QS_COUNTER_SELF_NARROW_SUBSTEPS: QS_HIP(hipMemcpy(&v, &h->d_stats[CTL_DEV + QS_DEVCTR_SELF_NARROW], sizeof(v), hipMemcpyDeviceToHost)); break;
    default: QS_FAIL(-1, "unknown counter %d", which);
    }
    *value = v;
    return 0;
}

int qs_last_step_kernel_ms(qs_handle* h, float* ms) {
    if (!h || !ms) QS_FAIL(-1, "null argument");
    if ((h->timing != 2 && h->timing != 3) || h->timed_launches <= 0) QS_FAIL(-1, "no step has been launched since qs_enable_timing(h, 1)");
    QS_ON_DEVICE(h);
    if (h->timing == 2) QS_HIP(hipEventRecord(h->ev1, h->stream));     // (3: qs_enable_timing(h, 2) recorded it already)
    QS_HIP(hipEventSynchronize(h->ev1));
    float total = 0.0f;
    QS_HIP(hipEventElapsedTime(&total, h->ev0, h->ev1));
    *ms = total / (float)h->timed_launches;
    h->timing = 1; h->timed_launches = 0;     // armed again: the next launch starts a new batch
    return 0;
}

// cfg.payload_soft: after qs_set_state / qs_set_params the payload block goes where its fixed constraint wants it
__global__ __launch_bounds__(QS_WAVE) void k_block_place(const qs_config* __restrict__ cfgp, float* __restrict__ recs) {
    const qs_config& cfg = *cfgp;
    const int env = blockIdx.x * QS_ENVS_PER_WAVE + (threadIdx.x >> 2);
    if (env >= cfg.n_envs) return;
    E::place_block(cfg, recs + (size_t)env * QS_REC);
}
static int place_blocks(qs_handle* h) {
    if (!h->cfg.payload_soft) return 0;
    hipLaunchKernelGGL(k_block_place, dim3(n_waves(h->cfg.n_envs)), dim3(QS_WAVE), 0, h->stream, h->d_cfg, h->d_rec);
    QS_HIP(hipGetLastError());
    return 0;
}
static int gather(qs_handle* h, int off, int dim, float* out, int as_int) {
    int total = h->cfg.n_envs * dim;
    hipLaunchKernelGGL(k_gather, dim3((total + 255) / 256), dim3(256), 0, h->stream, h->d_rec, h->cfg.n_envs, off, dim, out, as_int);
    QS_HIP(hipGetLastError());
    return 0;
}
static int scatter(qs_handle* h, int off, int dim, const float* in, int zero_warm) {
    int total = h->cfg.n_envs * dim;
    hipLaunchKernelGGL(k_scatter, dim3((total + 255) / 256), dim3(256), 0, h->stream, h->d_rec, h->cfg.n_envs, off, dim, in, zero_warm);
    QS_HIP(hipGetLastError());
    return 0;
}

// get_reward_end_episode (gym_env.py:363-365 -> task._reward_end_episode()) on the task state the records hold
__global__ __launch_bounds__(QS_WAVE) void k_reward_end(const qs_config* __restrict__ cfgp, const float* __restrict__ recs, float* __restrict__ out) {
    const qs_config& cfg = *cfgp;
    const int env = blockIdx.x * QS_ENVS_PER_WAVE + (threadIdx.x >> 2);
    const float* rec = recs + (size_t)(env < cfg.n_envs ? env : 0) * QS_REC;
    typename E::S::State s; E::Task t;
    E::load_state(rec, s); E::load_task(rec, t);
    t.pos[0] = s.pos.x; t.pos[1] = s.pos.y; t.pos[2] = s.pos.z;   // the task's pose cache after a step IS the state (task_base.py:72-75)
    float term = E::task_terminated(cfg, t, s, rec[R_N_INVALID], false);
    float r = E::task_reward_end(cfg, t, term, (float)((double)qs::f2i(rec[R_SIM_STEP]) * cfg.dt));
    if (env < cfg.n_envs && (threadIdx.x & 3) == 0) out[env] = r;
}
__global__ void k_wrapper_info(const float* __restrict__ recs, int n, float* __restrict__ out) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float* w = recs + (size_t)e * QS_REC + R_WRAP;
    out[4 * e] = w[W_PHASE]; out[4 * e + 1] = w[W_SCRIPTED]; out[4 * e + 2] = w[W_TIMER]; out[4 * e + 3] = w[W_END];
}
static int gather_wrapper(qs_handle* h, float* out) {
    hipLaunchKernelGGL(k_wrapper_info, dim3((h->cfg.n_envs + 255) / 256), dim3(256), 0, h->stream, h->d_rec, h->cfg.n_envs, out);
    QS_HIP(hipGetLastError());
    return 0;
}

int qs_get_state(qs_handle* h, float* st) { if (!h || !st) QS_FAIL(-1, "null argument"); QS_ON_DEVICE(h); return gather(h, R_POS, QS_STATE_DIM, st, 0); }
int qs_set_state(qs_handle* h, const float* st) {
    if (!h || !st) QS_FAIL(-1, "null argument");
    QS_ON_DEVICE(h);
    if (int rc = scatter(h, R_POS, QS_STATE_DIM, st, 1)) return rc;
    return place_blocks(h);
}

int qs_info_dim(const qs_handle* h, int which) {
    switch (which) {
    case QS_INFO_FOOT_FORCE: case QS_INFO_FOOT_CONTACT: case QS_INFO_COUNTERS: return 4;
    case QS_INFO_TORQUE: case QS_INFO_SPRING_TORQUE: case QS_INFO_LAST_ACTION: return 12;
    case QS_INFO_TASK: return QS_TASK_DIM;
    case QS_INFO_N_INVALID: return 1;
    case QS_INFO_PARAMS: return QS_PARAM_DIM;
    case QS_INFO_TERMINAL_OBS: return h ? h->cfg.obs_dim : -1;
    case QS_INFO_WRAPPER: return 4;
    case QS_INFO_FILTERED_ACTION: return 12;
    case QS_INFO_REWARD_END: return 1;
    case QS_INFO_PAYLOAD_BLOCK: return QS_BLOCK_DIM;
    default: return -1;
    }
}

int qs_get_info(qs_handle* h, int which, float* out) {
    if (!h || !out) QS_FAIL(-1, "null argument");
    QS_ON_DEVICE(h);
    if (!h->cfg.info_fields && (which == QS_INFO_FOOT_FORCE || which == QS_INFO_FOOT_CONTACT || which == QS_INFO_TORQUE || which == QS_INFO_SPRING_TORQUE))
        QS_FAIL(-1, "info id %d lives in the records' info block, which this handle does not write (cfg.info_fields = 0)", which);
    switch (which) {
    case QS_INFO_FOOT_FORCE: return gather(h, R_FOOT_FORCE, 4, out, 0);
    case QS_INFO_FOOT_CONTACT: return gather(h, R_FOOT_CONTACT, 4, out, 0);
    case QS_INFO_TORQUE: return gather(h, R_TAU_PD, 12, out, 0);
    case QS_INFO_SPRING_TORQUE: return gather(h, R_TAU_SPRING, 12, out, 0);
    case QS_INFO_N_INVALID: return gather(h, R_N_INVALID, 1, out, 0);
    case QS_INFO_PARAMS: return gather(h, R_PARAMS, QS_PARAM_DIM, out, 0);
    case QS_INFO_COUNTERS: return gather(h, R_SIM_STEP, 4, out, 1);
    case QS_INFO_LAST_ACTION: return gather(h, R_LAST_ACTION, 12, out, 0);
    case QS_INFO_WRAPPER: return gather_wrapper(h, out);
    case QS_INFO_FILTERED_ACTION: return gather(h, R_YHIST, 12, out, 0);
    case QS_INFO_PAYLOAD_BLOCK:
        if (!h->cfg.payload_soft) QS_FAIL(-1, "the payload block is a body of its own only under cfg.payload_soft");
        return gather(h, R_BLOCK, QS_BLOCK_DIM, out, 0);
    case QS_INFO_REWARD_END:
        hipLaunchKernelGGL(k_reward_end, dim3(n_waves(h->cfg.n_envs)), dim3(QS_WAVE), 0, h->stream, h->d_cfg, h->d_rec, out);
        QS_HIP(hipGetLastError());
        return 0;
    case QS_INFO_TASK:
        hipLaunchKernelGGL(k_task_info, dim3((h->cfg.n_envs + 255) / 256), dim3(256), 0, h->stream, h->d_rec, h->cfg.n_envs, out, (int)E::demo_task(h->cfg.task), h->cfg.info_fields);
        QS_HIP(hipGetLastError());
        return 0;
    case QS_INFO_TERMINAL_OBS:
        QS_HIP(hipMemcpyAsync(out, h->d_term_obs, (size_t)h->cfg.n_envs * h->cfg.obs_dim * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
        return 0;
    default: QS_FAIL(-1, "unknown info id %d", which);
    }
}

int qs_set_params(qs_handle* h, int which, const float* v) {
    if (!h || !v) QS_FAIL(-1, "null argument");
    QS_ON_DEVICE(h);
    switch (which) {
    case QS_PARAM_MU: return scatter(h, R_PARAMS + P_MU, 1, v, 0);
    case QS_PARAM_SPRING_K: return scatter(h, R_PARAMS + P_K, 3, v, 0);
    case QS_PARAM_SPRING_B: return scatter(h, R_PARAMS + P_B, 3, v, 0);
    case QS_PARAM_KP: return scatter(h, R_PARAMS + P_KP, 3, v, 0);
    case QS_PARAM_KD: return scatter(h, R_PARAMS + P_KD, 3, v, 0);
    case QS_PARAM_ALL: if (int rc = scatter(h, R_PARAMS, QS_PARAM_DIM, v, 0)) return rc; return place_blocks(h);
    default: QS_FAIL(-1, "unknown param id %d", which);
    }
}

int qs_stats(qs_handle* h, uint64_t* settle_substeps, uint64_t* resets) {
    if (!h) QS_FAIL(-1, "null handle");
    QS_ON_DEVICE(h);
    unsigned long long v[2];
    QS_HIP(hipStreamSynchronize(h->stream));
    QS_HIP(hipMemcpy(v, h->d_stats, sizeof(v), hipMemcpyDeviceToHost));
    if (settle_substeps) *settle_substeps = v[0];
    if (resets) *resets = v[1];
    return 0;
}

#ifdef QS_PROFILE_PHASES
// debug builds only (tools/phase_profile.py): cycles of workgroup 0 per substep phase, summed over all substeps so far
int qs_debug_phases(unsigned long long* out32, int reset) {
    QS_HIP(hipDeviceSynchronize());
    QS_HIP(hipMemcpyFromSymbol(out32, HIP_SYMBOL(qs_phase_cycles), 48 * sizeof(unsigned long long)));
    if (reset) { unsigned long long z[48] = {0}; QS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(qs_phase_cycles), z, sizeof(z))); }
    return 0;
}
#endif

}  // extern "C"
#endif  // QS_ISA_ONLY
