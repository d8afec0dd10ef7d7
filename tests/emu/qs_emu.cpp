// qs_emu.cpp -- TEST-ONLY host emulation of the quad-per-environment kernels.
// Instantiates the kernel arithmetic of quadruped-springs_amd/csrc/qs_env.h with the 4-wide LaneEmu type so that the
// CPU test-suite (no GPU in the build container) can compare it with the oracle.  Never linked into the product.
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../quadruped-springs_amd/csrc/qs_env.h"

using E = qs::Env<LaneEmu>;            // friction pyramid
using EC = qs::Env<LaneEmu, true>;     // implicit cone (cfg.friction_cone): the kernels are built for both, so is this harness

struct Emu {
    qs_config cfg;
    std::vector<float> rec, obs, term_obs;
    float* trace = nullptr; int trace_env = -1;
    std::vector<float> demo; int demo_len = 0;
};

static void init_record(const qs_config& cfg, float* r, int env) {
    memset(r, 0, QS_REC * sizeof(float));
    r[R_EPISODE] = qs::i2f(-1);
    r[R_QUAT + 3] = 1.0f; r[R_POS + 2] = 0.32f; r[R_TASK + T_FIRST_JUMP] = 1.0f;
    for (int L = 0; L < 4; L++) { r[R_Q + 3 * L + 1] = 0.78539816339f; r[R_Q + 3 * L + 2] = -1.57079632679f; }
    E::randomize(cfg, r, (uint32_t)(env + cfg.env_id_offset), -1, true);
}

template <class EV> static void phys_step_impl(Emu* e, int i, const float* tau12) {
    float* rec = &e->rec[(size_t)i * QS_REC];
    typename EV::S::State s; typename EV::S::Par P; typename EV::S::Out o;
    EV::load_state(rec, s); EV::load_par(e->cfg, rec, P);
    V4 tau[3];
    for (int j = 0; j < 3; j++) { tau[j] = LaneEmu::ld_leg(tau12, j, 3); o.tau_pd[j] = V4(0.0f); o.tau_spring[j] = V4(0.0f); }
    EV::S::substep(e->cfg, P, s, tau, o, true, e->cfg.payload_soft ? rec + R_BLOCK : nullptr);
    EV::store_state(rec, s, o);
}

extern "C" {
void* qse_create(const qs_config* cfg) {
    Emu* e = new Emu();
    e->cfg = *cfg;
    e->rec.assign((size_t)cfg->n_envs * QS_REC, 0.0f);
    e->obs.assign((size_t)cfg->n_envs * QS_MAX_OBS, 0.0f);
    e->term_obs.assign((size_t)cfg->n_envs * QS_MAX_OBS, 0.0f);
    for (int i = 0; i < cfg->n_envs; i++) init_record(e->cfg, &e->rec[(size_t)i * QS_REC], i);
    return e;
}
void qse_destroy(void* h) { delete (Emu*)h; }
int qse_set_trace(void* h, int env, float* rows) { Emu* e = (Emu*)h; e->trace_env = env; e->trace = env >= 0 ? rows : nullptr; return 0; }
int qse_reset(void* h, const uint8_t* mask) {
    Emu* e = (Emu*)h;
    for (int i = 0; i < e->cfg.n_envs; i++)
        if (!mask || mask[i]) {
            if (e->cfg.friction_cone) EC::reset(e->cfg, &e->rec[(size_t)i * QS_REC], &e->obs[(size_t)i * QS_MAX_OBS], (uint32_t)(i + e->cfg.env_id_offset), true);
            else E::reset(e->cfg, &e->rec[(size_t)i * QS_REC], &e->obs[(size_t)i * QS_MAX_OBS], (uint32_t)(i + e->cfg.env_id_offset), true);
        }
    return 0;
}
// qs_reset_to (k_reset with states): randomizers, the given rigid-body state, task / sensor / filter reset, zero action history
int qse_reset_to(void* h, const uint8_t* mask, const float* states) {
    Emu* e = (Emu*)h;
    for (int i = 0; i < e->cfg.n_envs; i++) {
        if (mask && !mask[i]) continue;
        float* rec = &e->rec[(size_t)i * QS_REC];
        const uint32_t gid = (uint32_t)(i + e->cfg.env_id_offset);
        E::randomize(e->cfg, rec, gid, qs::f2i(rec[R_EPISODE]) + 1, false);
        memcpy(rec + R_POS, states + (size_t)i * 37, 37 * sizeof(float));
        for (int k = 0; k < 4; k++) { rec[R_WARM + k] = 0.0f; rec[R_FOOT_FORCE + k] = 0.0f; rec[R_FOOT_CONTACT + k] = 0.0f; }
        rec[R_N_INVALID] = 0.0f;
        for (int k = 0; k < 24; k++) rec[R_TAU_PD + k] = 0.0f;
        if (e->cfg.payload_soft) E::place_block(e->cfg, rec);
        E::reset(e->cfg, rec, &e->obs[(size_t)i * QS_MAX_OBS], gid, false);
        for (int k = 0; k < 12 + 24 + 24; k++) rec[R_LAST_ACTION + k] = 0.0f;
    }
    return 0;
}
int qse_set_demo(void* h, const float* rows, int length) {
    Emu* e = (Emu*)h;
    e->demo.assign(rows, rows + (size_t)length * (e->cfg.action_dim + 38));
    e->demo_len = length;
    return 0;
}
int qse_set_demo_counter(void* h, const uint8_t* mask, const int32_t* values) {
    Emu* e = (Emu*)h;
    for (int i = 0; i < e->cfg.n_envs; i++)
        if (!mask || mask[i]) { float* r = &e->rec[(size_t)i * QS_REC + R_DEMO]; r[0] = r[1] = (float)values[i]; }
    return 0;
}
int qse_get_obs(void* h, float* obs) {
    Emu* e = (Emu*)h;
    for (int i = 0; i < e->cfg.n_envs; i++) memcpy(obs + (size_t)i * e->cfg.obs_dim, &e->obs[(size_t)i * QS_MAX_OBS], e->cfg.obs_dim * sizeof(float));
    return 0;
}
// infos[i]["terminal_observation"] of the last episode each environment finished (auto_reset)
int qse_get_term_obs(void* h, float* obs) {
    Emu* e = (Emu*)h;
    for (int i = 0; i < e->cfg.n_envs; i++) memcpy(obs + (size_t)i * e->cfg.obs_dim, &e->term_obs[(size_t)i * QS_MAX_OBS], e->cfg.obs_dim * sizeof(float));
    return 0;
}
int qse_step(void* h, const float* actions, float* obs, float* rew, uint8_t* done, uint8_t* trunc) {
    Emu* e = (Emu*)h;
    const int d = e->cfg.action_dim;
    for (int i = 0; i < e->cfg.n_envs; i++) {
        float* rec = &e->rec[(size_t)i * QS_REC];
        float* ob = &e->obs[(size_t)i * QS_MAX_OBS];
        float* tr = (e->trace && i == e->trace_env) ? e->trace : nullptr;
        float rw, dn, tc;
        if (e->cfg.friction_cone) { EC::StepOut r = EC::step(e->cfg, rec, actions + (size_t)i * d, ob, (uint32_t)(i + e->cfg.env_id_offset), 0, tr, tr != nullptr, e->demo.data(), e->demo_len); rw = r.reward.v[0]; dn = r.done.v[0]; tc = r.trunc.v[0]; }
        else { E::StepOut r = E::step(e->cfg, rec, actions + (size_t)i * d, ob, (uint32_t)(i + e->cfg.env_id_offset), 0, tr, tr != nullptr, e->demo.data(), e->demo_len); rw = r.reward.v[0]; dn = r.done.v[0]; tc = r.trunc.v[0]; }
        rew[i] = rw; done[i] = dn > 0.5f; trunc[i] = tc > 0.5f;
        if (done[i] && e->cfg.auto_reset) {
            memcpy(&e->term_obs[(size_t)i * QS_MAX_OBS], ob, QS_MAX_OBS * sizeof(float));
            if (e->cfg.friction_cone) EC::reset(e->cfg, rec, ob, (uint32_t)(i + e->cfg.env_id_offset), true);
            else E::reset(e->cfg, rec, ob, (uint32_t)(i + e->cfg.env_id_offset), true);
        }
        memcpy(obs + (size_t)i * e->cfg.obs_dim, ob, e->cfg.obs_dim * sizeof(float));
    }
    return 0;
}
int qse_get_state(void* h, float* st) {
    Emu* e = (Emu*)h;
    for (int i = 0; i < e->cfg.n_envs; i++) memcpy(st + (size_t)i * 37, &e->rec[(size_t)i * QS_REC + R_POS], 37 * sizeof(float));
    return 0;
}
int qse_set_state(void* h, const float* st) {
    Emu* e = (Emu*)h;
    for (int i = 0; i < e->cfg.n_envs; i++) {
        float* r = &e->rec[(size_t)i * QS_REC];
        memcpy(r + R_POS, st + (size_t)i * 37, 37 * sizeof(float));
        for (int k = 0; k < 4; k++) r[R_WARM + k] = 0.0f;
        if (e->cfg.payload_soft) E::place_block(e->cfg, r);
    }
    return 0;
}
// the payload block as its own body (cfg.payload_soft): [N, 20], the oracle's qso_get_block row
int qse_get_block(void* h, float* out) {
    Emu* e = (Emu*)h;
    for (int i = 0; i < e->cfg.n_envs; i++) memcpy(out + (size_t)i * QS_BLOCK_DIM, &e->rec[(size_t)i * QS_REC + R_BLOCK], QS_BLOCK_DIM * sizeof(float));
    return 0;
}
float* qse_records(void* h) { return ((Emu*)h)->rec.data(); }
int qse_rec_size(void) { return QS_REC; }
int qse_field(const char* name) {
#define F(n) if (!strcmp(name, #n)) return n;
    F(R_POS) F(R_QUAT) F(R_VLIN) F(R_VANG) F(R_Q) F(R_QD) F(R_WARM) F(R_LAST_ACTION) F(R_XHIST) F(R_YHIST) F(R_SIM_STEP) F(R_ENV_STEP)
    F(R_EPISODE) F(R_TOTAL_STEPS) F(R_TASK) F(R_NEW_TAU) F(R_PARAMS) F(R_FOOT_FORCE) F(R_FOOT_CONTACT) F(R_N_INVALID) F(R_TAU_PD)
    F(R_TAU_SPRING) F(R_POSE_CACHE) F(R_CPG) F(R_DEMO) F(R_WRAP) F(R_BLOCK)
#undef F
    return -1;
}
// the separating-axis box / box test of the self-collision rule (qs_core.h obb_overlap); R row-major with the axes as COLUMNS
int qse_obb_overlap(const float* ca, const float* Ra, const float* ha, const float* cb, const float* Rb, const float* hb) {
    using S = E::S;
    auto col = [](const float* R, int j) { return qs::mk3<V4>(V4(R[j]), V4(R[3 + j]), V4(R[6 + j])); };
    S::H3 HA = {{ha[0], ha[1], ha[2]}}, HB = {{hb[0], hb[1], hb[2]}};
    M4 m = S::obb_overlap(qs::mk3<V4>(V4(ca[0]), V4(ca[1]), V4(ca[2])), col(Ra, 0), col(Ra, 1), col(Ra, 2), HA,
                          qs::mk3<V4>(V4(cb[0]), V4(cb[1]), V4(cb[2])), col(Rb, 0), col(Rb, 1), col(Rb, 2), HB);
    return m.v[0] ? 1 : 0;
}
// one physics substep of env `i` under given joint torques (KATs on the kernel arithmetic)
int qse_phys_step(void* h, int i, const float* tau12) {
    Emu* e = (Emu*)h;
    if (e->cfg.friction_cone) phys_step_impl<EC>(e, i, tau12);
    else phys_step_impl<E>(e, i, tau12);
    return 0;
}
}
