// qs_rare.h -- the many-rows contact solve: one environment at a time, ONE ROW PER LANE.
//
// Rows beyond the three foot-contact rows of a leg -- a row per violated joint limit, normal + friction rows of up to two more support
// points per leg (trunk corner, hip housing, thigh ends, knee end of the calf: a fallen robot rests on them), the six rows of the payload
// block's fixed constraint -- make up to 54 rows per environment.  The quad layout of the common path (one leg per lane, 16 environments
// per wave) has no room for them: rounds 2-3 swept them in VELOCITY space, twelve row slots per lane, every slot's candidate recomputed by
// nine FMAs in all four lanes of the quad for the one that owned it -- ~38 instructions per row and sweep, ~216 k cycles per substep of
// a wave with ONE fallen robot (profiles/r03_g_rare_path.md), and the launch ends with its slowest wave.
//
// Here the wave turns to one environment with rare rows at a time (the others wait; they get the common-path solver's result anyway):
//   * the environment's quad writes its rows to LDS, each at the record of the lane that is to take it: the rows are packed, in Bullet's
//     sweep order, into three regions of the wave (limit + payload rows, normals, friction pairs -- RarePos below), and a row's lane is
//     worked out by the quad lane that holds it (round 5: a prefix sum over the quad; round 4 had every lane search a 64-bit mask);
//   * every lane builds ITS column of the Delassus matrix, A[j][p] = w_j . w_p + [same body part] a_j . b_p, pre-scaled by -1 / A_pp,
//     into registers, the other rows' data fetched with v_readlane: up to 54 values;
//   * projected Gauss-Seidel in impulse space on lane-private candidates, as in the common path: a row update is one v_med3 (every lane
//     clamps its own candidate), a subtract, one v_readlane of the owner's delta into an SGPR and one v_fmac of it into every lane's
//     candidate -- six instructions per unilateral row, eighteen per friction pair, only for rows that exist (they are packed into
//     consecutive lanes);
//   * the environment leaves the sweeps when ITS residual is under PyBullet's threshold (not when the wave's sixteen are);
//   * the impulses go back to the quad through LDS, which turns them into velocities (Sim::substep).
// Same rows, order, clamps and early exit as before (and as oracle/qso_phys.c); a result depends on the environment's own rows only.
//
// LDS: 54 rows x 16 floats + 64 impulses + a dummy record = 3.9 KB of the wave's observation rows, which nobody uses between two epilogues.
// (included by qs_core.h inside namespace qs, after SimTypes)
#pragma once

// positions of an environment's rows in sweep order: 0..11 joint limits (leg K, joint j: 3 K + j), 12..17 payload rows, 18..29 normals
// (leg K, contact point c, 0 = the foot: 18 + 3 K + c), 30..53 friction rows (30 + 6 K + 2 c + t).  `row`: index into the quad lane's
// twelve rows (contact point c at 3 c, 3 c + 1, 3 c + 2; joint limits at 9 + j); `rec`: index of the row's record in LDS.
struct RarePos {
    static constexpr int N = 54, LIM0 = 0, PAY0 = 12, NRM0 = 18, FRI0 = 30, REC_FLOATS = 16, LAM_OFF = N * REC_FLOATS, SCRATCH_FLOATS = LAM_OFF + 64;
    static constexpr int leg(int p) { return p < PAY0 ? p / 3 : p < NRM0 ? 4 : p < FRI0 ? (p - NRM0) / 3 : (p - FRI0) / 6; }   // 4: the payload block
    static constexpr int row(int p) { return p < PAY0 ? 9 + p % 3 : p < NRM0 ? p - PAY0 : p < FRI0 ? 3 * ((p - NRM0) % 3) : 3 * (((p - FRI0) % 6) / 2) + 1 + (p - FRI0) % 2; }
    static constexpr int rec(int p) { return p >= PAY0 && p < NRM0 ? 48 + (p - PAY0) : 12 * leg(p) + row(p); }
    static constexpr int normal_of(int p) { return NRM0 + 3 * ((p - FRI0) / 6) + ((p - FRI0) % 6) / 2; }   // friction position -> its contact's normal
    static constexpr int of_row(int K, int r) { return r >= 9 ? 3 * K + (r - 9) : (r % 3 == 0 ? NRM0 + 3 * K + r / 3 : FRI0 + 6 * K + 2 * (r / 3) + (r % 3 - 1)); }
};

template <class T, bool CONE> struct RareSolver;
// M(0) .. M(n - 1) for the first `count` of them, as nested ifs: once j reaches the (wave-uniform) count ONE branch leaves the lot, and the
// code stays structured (a goto per row, or a switch that falls through the rows, comes out of LLVM's CFG structurizer as flag variables
// and four branches per row: measured 1.2 k cycles per sweep of ten rows)
#define QS_NEST12(count, M) if (0 < (count)) { M(0) if (1 < (count)) { M(1) if (2 < (count)) { M(2) if (3 < (count)) { M(3) if (4 < (count)) { M(4) if (5 < (count)) { M(5) if (6 < (count)) { M(6) if (7 < (count)) { M(7) if (8 < (count)) { M(8) if (9 < (count)) { M(9) if (10 < (count)) { M(10) if (11 < (count)) { M(11) }}}}}}}}}}}}
#define QS_NEST4(count, M) if (0 < (count)) { M(0) if (1 < (count)) { M(1) if (2 < (count)) { M(2) if (3 < (count)) { M(3) }}}}
#define QS_NEST6(count, M) if (0 < (count)) { M(0) if (1 < (count)) { M(1) if (2 < (count)) { M(2) if (3 < (count)) { M(3) if (4 < (count)) { M(4) if (5 < (count)) { M(5) }}}}}}
#define QS_NEST18(count, M) if (0 < (count)) { M(0) if (1 < (count)) { M(1) if (2 < (count)) { M(2) if (3 < (count)) { M(3) if (4 < (count)) { M(4) if (5 < (count)) { M(5) if (6 < (count)) { M(6) if (7 < (count)) { M(7) if (8 < (count)) { M(8) if (9 < (count)) { M(9) if (10 < (count)) { M(10) if (11 < (count)) { M(11) if (12 < (count)) { M(12) if (13 < (count)) { M(13) if (14 < (count)) { M(14) if (15 < (count)) { M(15) if (16 < (count)) { M(16) if (17 < (count)) { M(17) }}}}}}}}}}}}}}}}}}

#if defined(__HIPCC__)
// ------------------------------------------------------------------ the wave-wide solver (device)
// Only rows that exist take a lane: the limit and payload rows go to lanes 0.. (region A, in sweep order), the normals to lanes 18..
// (region B) and the friction pairs to lanes 30.. (region C; pair k belongs to the normal in lane 18 + k), each region packed to its
// start.  The unrolled sweep code then walks a region up to its fill count and meets no empty row -- the first version kept every row
// at its fixed position and skipped the empty ones with a scalar branch each: 44 taken branches per sweep of a typical fallen robot
// (10 rows), 1.7 k cycles per sweep of which the rows themselves were 0.4 k (profiles/r04_d_rare_phases.md).
template <bool CONE> struct RareSolver<LaneDev, CONE> {
    using Ty = SimTypes<LaneDev>;
    using Row = typename Ty::Row;
    using PayRows = typename Ty::PayRows;
    static QS_DEV float rl(float x, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane)); }
    static QS_DEV int rli(int x, int lane) { return __builtin_amdgcn_readlane(x, lane); }

    // Delassus column, candidates and sweeps of one environment's solve (the lanes hold their rows: w, a, b, rhs, dinv, diag, lam, lo, hi, grp).
    // NAX / NBX: how many rows of region A / contact points the coefficient arrays hold -- <18, 12> takes everything, <0, 6> what the benchmark's
    // falling robots need (no joint at its stop, six contact points at most) with a quarter of the registers, <0, 4> the four-point solves among
    // them with a sixth (round 5: 66.2 -> 68.1 -> 70.3 M on the benchmark; a finer ladder -- <0, 2>, <0, 3> -- added nothing).
    template <int NAX, int NBX>
    static QS_DEV void core(const qs_config& cfg, const float (&w)[6], const float (&a)[3], const float (&b)[3], float rhs, float dinv, float diag, float& lam,
                            float lo, float hi, int grp, int mA, int mB, float mu_e, bool track, float thr, int& n_sweeps) {
        using P = RarePos;
        constexpr int A0 = 0, B0 = P::NRM0, C0 = P::FRI0, IB = NAX, IC = NAX + NBX;
        const int lane = (int)threadIdx.x;
            // ---- this lane's column of the Delassus matrix, x (-1 / A_pp): Ap[J] = what a unit impulse of the row in lane J does to this
        // lane's candidate.  The row data of lane J come over by v_readlane (SGPRs); the self entry is zero.
        float Ap[NAX + 3 * NBX], ApR[NAX > 0 ? NAX : 1];
        {
            const float nd = -dinv;
            float ws[6], bs[3];
#pragma unroll
            for (int i = 0; i < 6; i++) ws[i] = w[i] * nd;
#pragma unroll
            for (int i = 0; i < 3; i++) bs[i] = b[i] * nd;
#define QS_W_COL(DST, J)                                                                                               \
    {                                                                                                                  \
        const int l_ = (J);                                                                                            \
        float t_ = rl(w[0], l_) * ws[0];                                                                               \
        t_ = fmaf(rl(w[1], l_), ws[1], t_); t_ = fmaf(rl(w[2], l_), ws[2], t_); t_ = fmaf(rl(w[3], l_), ws[3], t_);     \
        t_ = fmaf(rl(w[4], l_), ws[4], t_); t_ = fmaf(rl(w[5], l_), ws[5], t_);                                         \
        float u_ = rl(a[0], l_) * bs[0];                                                                               \
        u_ = fmaf(rl(a[1], l_), bs[1], u_); u_ = fmaf(rl(a[2], l_), bs[2], u_);                                         \
        t_ = grp == rli(grp, l_) ? t_ + u_ : t_;                                                                       \
        DST = lane == l_ ? 0.0f : t_;                                                                                  \
    }
            // the rows of region A are swept forwards and backwards in turn: ApR[j] is the entry of the row that the BACKWARD sweep meets
            // j-th (lane mA - 1 - j), so that both directions index their coefficients with constants
#define QS_W_COL_A(j) QS_W_COL(Ap[(j)], A0 + (j))
#define QS_W_COL_R(j) QS_W_COL(ApR[(j)], mA - 1 - (j))
#define QS_W_COL_B(j) QS_W_COL(Ap[IB + (j)], B0 + (j)) QS_W_COL(Ap[IC + 2 * (j)], C0 + 2 * (j)) QS_W_COL(Ap[IC + 2 * (j) + 1], C0 + 2 * (j) + 1)
            if constexpr (NAX > 0) { QS_NEST18(mA, QS_W_COL_A) QS_NEST18(mA, QS_W_COL_R) }
            if constexpr (NBX > 6) { QS_NEST12(mB, QS_W_COL_B) } else if constexpr (NBX > 4) { QS_NEST6(mB, QS_W_COL_B) } else { QS_NEST4(mB, QS_W_COL_B) }
#undef QS_W_COL_A
#undef QS_W_COL_R
#undef QS_W_COL_B
#undef QS_W_COL
        }
        QS_PHASE_G(42)
        // ---- candidates: rhs - dinv sum_{j != p} A_pj lambda_j; the warm start of the feet's normal rows is in already
        float res = rhs;
#define QS_W_WARM(j) res = fmaf(Ap[IB + (j)], rl(lam, B0 + (j)), res);
        if constexpr (NBX > 6) { QS_NEST12(mB, QS_W_WARM) } else if constexpr (NBX > 4) { QS_NEST6(mB, QS_W_WARM) } else { QS_NEST4(mB, QS_W_WARM) }
#undef QS_W_WARM
        // one row: every lane clamps its own candidate, the owner's change goes round
#define QS_W_ROW(COEF, J)                                                                                              \
    {                                                                                                                  \
        const int l_ = (J);                                                                                            \
        const float cand = qmed3(res, lo, hi);                                                                         \
        const float d_ = rl(cand - lam, l_);                                                                           \
        lam = lane == l_ ? cand : lam;                                                                                 \
        res = fmaf((COEF), d_, res);                                                                                   \
    }
        // the friction rows of contact point KK (lanes C0 + 2 KK, + 1; its normal sits in lane B0 + KK): implicit cone -- both from the
        // same candidates, their sum scaled back onto the disc of radius mu x the normal impulse --, or the pyramid: one by one, each
        // bounded by mu x the normal impulse and left alone while that is not positive (Bullet's rule)
#define QS_W_FRICTION(KK)                                                                                              \
    {                                                                                                                  \
        constexpr int FP = C0 + 2 * (KK);                                                                              \
        if (CONE) {                                                                                                    \
        const float lim_ = rl(lam * mu_e, B0 + (KK));          /* (every lane's product; the normal's counts) */   \
        const float ca_ = rl(res, FP), cb_ = rl(res, FP + 1);                                                      \
        const float r2_ = fmaf(cb_, cb_, fmaf(ca_, ca_, 1e-30f));   /* (+ 1e-30: no candidate, no division by zero) */ \
        const float sc_ = qmin(lim_ * qrsqrt(r2_), 1.0f);                                                          \
        const float cand = res * sc_;                                                                              \
        const float dl_ = cand - lam;                                                                              \
        const float da_ = rl(dl_, FP), db_ = rl(dl_, FP + 1);                                                      \
        lam = (lane >> 1) == (FP >> 1) ? cand : lam;                                                               \
        res = fmaf(Ap[IC + 2 * (KK)], da_, res); res = fmaf(Ap[IC + 2 * (KK) + 1], db_, res);                                            \
        } else {                                                                                                       \
        const float ln_ = rl(lam, B0 + (KK)), lim_ = mu_e * ln_;                                                   \
        _Pragma("unroll") for (int t_ = 0; t_ < 2; t_++) {                                                         \
            const float cl_ = qmed3(res, -lim_, lim_);                                                             \
            const float d_ = ln_ > 0.0f ? rl(cl_ - lam, FP + t_) : 0.0f;                                           \
            lam = lane == FP + t_ && ln_ > 0.0f ? cl_ : lam;                                                       \
            res = fmaf(Ap[IC + 2 * (KK) + t_], d_, res);                                                                      \
        }                                                                                                          \
        }                                                                                                              \
    }
        const int mA_fix = mA, mB_fix = mB;
        for (int it = 0; it < cfg.solver_iters; it++) {
            n_sweeps++;
            const float lam_in = lam;
            // (opaque copies: otherwise the compiler hoists the ~100 loop-invariant comparisons `j < count` and `lane == j` out of the
            // sweep loop as lane masks, spills them into VGPR lanes and fetches each back with two v_readlane per row)
            int mA = mA_fix, mB = mB_fix, lane = (int)threadIdx.x;
            asm volatile("" : "+s"(mA), "+s"(mB), "+v"(lane));
#define QS_W_FWD_A(j) QS_W_ROW(Ap[(j)], A0 + (j))
#define QS_W_BWD_A(j) QS_W_ROW(ApR[(j)], mA - 1 - (j))
#define QS_W_FWD_B(j) QS_W_ROW(Ap[IB + (j)], B0 + (j))
#define QS_W_FWD_C(j) QS_W_FRICTION(j)
            if constexpr (NAX > 0) {
                if (it & 1) { QS_NEST18(mA, QS_W_FWD_A) }   // limit rows, then the payload rows, forwards; on even sweeps the same backwards
                else { QS_NEST18(mA, QS_W_BWD_A) }
            }
            if constexpr (NBX > 6) { QS_NEST12(mB, QS_W_FWD_B) QS_NEST12(mB, QS_W_FWD_C) } else if constexpr (NBX > 4) { QS_NEST6(mB, QS_W_FWD_B) QS_NEST6(mB, QS_W_FWD_C) } else { QS_NEST4(mB, QS_W_FWD_B) QS_NEST4(mB, QS_W_FWD_C) }
#undef QS_W_FWD_A
#undef QS_W_BWD_A
#undef QS_W_FWD_B
#undef QS_W_FWD_C
            // PyBullet's solverResidualThreshold: every row moved once in this sweep, by lam - lam_in
            if (track && !__any(fabsf((lam - lam_in) * diag) > thr)) break;
        }
#undef QS_W_FRICTION
#undef QS_W_ROW
    }

    // xr: the twelve rows of this lane's leg; pay: the block's rows (replicated over the quad) or nullptr; mine: this lane's environment has
    // rare rows; warm: the foot's warm-start impulse (already x cfg.warmstart x act); scr: RarePos::SCRATCH_FLOATS floats of LDS shared by
    // the wave.  Results (for the lanes with `mine`, zero elsewhere): lam12 = the impulses of the lane's twelve rows, plam = the payload rows'.
    static QS_DEV void solve(const qs_config& cfg, float mu, const Row* xr, const PayRows* pay, bool mine, float warm, float* scr, float* lam12, float* plam) {
        using P = RarePos;
        constexpr int A0 = 0, B0 = P::NRM0, C0 = P::FRI0, NA = P::NRM0, NB = P::FRI0 - P::NRM0;
        const int lane = (int)threadIdx.x, slot = lane >> 2, K = lane & 3;
#pragma unroll
        for (int r = 0; r < 12; r++) lam12[r] = 0.0f;
#pragma unroll
        for (int k = 0; k < 6; k++) plam[k] = 0.0f;
        const bool track = cfg.solver_residual_threshold > 0.0f;
        const float thr = sqrtf(cfg.solver_residual_threshold);
        const float big = 1e10f, bound = 500.0f * (float)cfg.dt;
        const unsigned long long maskA = (1ull << P::NRM0) - 1ull, maskB = ((1ull << P::FRI0) - 1ull) & ~maskA, maskC = ((1ull << P::N) - 1ull) & ~(maskA | maskB);
        // Which lane a row goes to is worked out by the lanes that HOLD the rows (round 5; before: a 64-bit mask of the canonical positions from
        // twelve ballots, and three loops over its set bits in which every lane looked for its rank): a leg's limit rows, normals and friction
        // pairs sit leg by leg in their regions, so a row's lane is the region's start + the rows of the legs in front of it (an exclusive
        // prefix over the quad: three DPP broadcasts) + the leg's own rows in front of it.
        bool lrow[3], nrow[3];
        int n_lim = 0, n_nrm = 0;
#pragma unroll
        for (int j = 0; j < 3; j++) { lrow[j] = xr[9 + j].act > 0.5f; n_lim += lrow[j] ? 1 : 0; nrow[j] = xr[3 * j].act > 0.5f; n_nrm += nrow[j] ? 1 : 0; }
        const int packed = n_lim | (n_nrm << 8);     // (one value through the DPP moves)
        auto qb = [](int x, int k) {
            return k == 0 ? __builtin_amdgcn_mov_dpp(x, 0x00, 0xF, 0xF, true) : k == 1 ? __builtin_amdgcn_mov_dpp(x, 0x55, 0xF, 0xF, true)
                 : k == 2 ? __builtin_amdgcn_mov_dpp(x, 0xAA, 0xF, 0xF, true) : __builtin_amdgcn_mov_dpp(x, 0xFF, 0xF, 0xF, true);
        };
        const int q0 = qb(packed, 0), q1 = qb(packed, 1), q2 = qb(packed, 2), q3 = qb(packed, 3);
        const int pre = (K > 0 ? q0 : 0) + (K > 1 ? q1 : 0) + (K > 2 ? q2 : 0), tot = q0 + q1 + q2 + q3;
        const int preA = pre & 0xFF, preB = pre >> 8;
        constexpr int DUMMY = 60;          // a record nobody reads: where the rows that do not exist are written (no EXEC juggling per row)
        static_assert((DUMMY + 1) * P::REC_FLOATS <= P::LAM_OFF + 64 + 64 && DUMMY >= P::N, "the dummy record lies behind the rows'");
        unsigned long long todo = __ballot(mine);
        while (todo) {
            const int e = (__ffsll((long long)todo) - 1) >> 2;     // wave-uniform: the quad this pass works for
            todo &= ~(0xFull << (4 * e));
            const float mu_e = rl(mu, 4 * e);
            const int tot_e = rli(tot, 4 * e);
            const int mLim = tot_e & 0xFF, mB = tot_e >> 8;        // limit rows and contact points of the environment
            const bool pay_live = pay != nullptr && rl(pay->act, 4 * e) > 0.5f;
            const int mA = mLim + (pay_live ? 6 : 0);              // region A: the limit rows, then the payload rows
            // ---- the quad's rows -> LDS, each at the record of ITS lane (16 floats: w 6, a 3, leg, b 3, rhs, dinv, diag); the normals' warm
            // start goes where the impulses come back from
            unsigned rows_used = 0u;     // bit r: row r exists in some leg (wave-uniform: a row slot nobody fills is not written at all)
#pragma unroll
            for (int r = 0; r < 12; r++) if (__ballot(slot == e && xr[r].act > 0.5f) != 0ull) rows_used |= 1u << r;
            if (slot == e) {
                const float legf = __builtin_bit_cast(float, K);
                auto put = [&](const Row& q, int at, float grp_bits) {
                    float4* d = reinterpret_cast<float4*>(scr + at * P::REC_FLOATS);
                    d[0] = make_float4(q.w[0], q.w[1], q.w[2], q.w[3]);
                    d[1] = make_float4(q.w[4], q.w[5], q.jq[0], q.jq[1]);
                    d[2] = make_float4(q.jq[2], grp_bits, q.u[0], q.u[1]);
                    d[3] = make_float4(q.u[2], q.rhs, q.dinv, q.diag);
                };
                int dl = A0 + preA, dn = B0 + preB;
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    if (!((rows_used >> (9 + j)) & 1u)) continue;
                    const int at = lrow[j] ? dl : DUMMY;
                    put(xr[9 + j], at, legf);
                    scr[P::LAM_OFF + at] = 0.0f;
                    dl += lrow[j] ? 1 : 0;
                }
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    if (!((rows_used >> (3 * c)) & 1u)) continue;
                    const int at = nrow[c] ? dn : DUMMY, af = nrow[c] ? C0 + 2 * (dn - B0) : DUMMY;
                    put(xr[3 * c], at, legf);
                    put(xr[3 * c + 1], af, legf);
                    put(xr[3 * c + 2], nrow[c] ? af + 1 : DUMMY, legf);
                    scr[P::LAM_OFF + at] = c == 0 ? warm : 0.0f;
                    scr[P::LAM_OFF + af] = 0.0f; scr[P::LAM_OFF + (nrow[c] ? af + 1 : DUMMY)] = 0.0f;
                    dn += nrow[c] ? 1 : 0;
                }
                // (payload rows: a = -(rB x e_k) resp. -e_k, the block's angular Jacobian; b = a / inertia.  The linear part only meets itself.)
                if (pay_live) {
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        if ((k & 3) != K) continue;           // lane K of the quad writes rows K and K + 4
                        const int at = A0 + mLim + k;
                        float4* d = reinterpret_cast<float4*>(scr + at * P::REC_FLOATS);
                        const PayRows& q = *pay;
                        const float ax = k == 0 ? 0.0f : k == 1 ? q.rB.z : k == 2 ? -q.rB.y : k == 3 ? -1.0f : 0.0f;
                        const float ay = k == 0 ? -q.rB.z : k == 1 ? 0.0f : k == 2 ? q.rB.x : k == 4 ? -1.0f : 0.0f;
                        const float az = k == 0 ? q.rB.y : k == 1 ? -q.rB.x : k == 5 ? -1.0f : 0.0f;
                        d[0] = make_float4(q.w[k][0], q.w[k][1], q.w[k][2], q.w[k][3]);
                        d[1] = make_float4(q.w[k][4], q.w[k][5], ax, ay);
                        d[2] = make_float4(az, __builtin_bit_cast(float, 4), ax * q.mI, ay * q.mI);
                        d[3] = make_float4(az * q.mI, q.rhs[k], q.dinv[k], q.diag[k]);
                        scr[P::LAM_OFF + at] = 0.0f;
                    }
                }
            }
            // ---- a lane inside the filled part of a region has a row; its position in Bullet's sweep order is the lane itself
            const bool inA = lane < A0 + mA, inB = lane >= B0 && lane < B0 + mB, inC = lane >= C0 && lane < C0 + 2 * mB;
            const bool alive = inA || inB || inC;
            LaneDev::sync();
            // ---- this lane's row; the feet's warm start
            float w[6], a[3], b[3], rhs, dinv, diag, lam, lo, hi;
            int grp;
            {
                const int me = alive ? lane : DUMMY;
                const float4* s4 = reinterpret_cast<const float4*>(scr + me * P::REC_FLOATS);
                const float4 r0 = s4[0], r1 = s4[1], r2 = s4[2], r3 = s4[3];
                lam = scr[P::LAM_OFF + me];
                w[0] = r0.x; w[1] = r0.y; w[2] = r0.z; w[3] = r0.w; w[4] = r1.x; w[5] = r1.y;
                a[0] = r1.z; a[1] = r1.w; a[2] = r2.x;
                grp = __builtin_bit_cast(int, r2.y);
                b[0] = r2.z; b[1] = r2.w; b[2] = r3.x;
                rhs = r3.y; dinv = r3.z; diag = r3.w;
                const bool payrow = inA && lane >= A0 + mLim;
                lo = payrow ? -bound : 0.0f; hi = payrow ? bound : big;
                if (!alive) { grp = -1; rhs = 0.0f; diag = 0.0f; lam = 0.0f; }
            }
            QS_PHASE_G(41)
            // ---- Delassus column, candidates, sweeps (core<>, above).  A solve without limit / payload rows and with six contact points at most --
            // every one of the benchmark's -- takes an instantiation with 18 (12: four contact points) coefficient registers instead of 72: the
            // sweep loop of the large one re-read ~ 30 coefficients from AGPRs and SGPR-spill lanes at the head of every sweep
            int n_sweeps = 0;
#ifdef QS_DBG_CORE4   // diagnostic: both small instantiations on the same rows, bitwise mismatches into the self-narrow counter (tools/diag/core4_differential.py)
            if (mA == 0 && mB <= 4) {
                float lam6 = lam; int ns6 = 0;
                core<0, 6>(cfg, w, a, b, rhs, dinv, diag, lam6, lo, hi, grp, mA, mB, mu_e, track, thr, ns6);
                core<0, 4>(cfg, w, a, b, rhs, dinv, diag, lam, lo, hi, grp, mA, mB, mu_e, track, thr, n_sweeps);
                const bool bad = alive && __builtin_bit_cast(int, lam) != __builtin_bit_cast(int, lam6);
                const unsigned long long bm = __ballot(bad);
                if ((bm != 0ull || ns6 != n_sweeps) && threadIdx.x == 0)
                    atomicAdd(&reinterpret_cast<const QsDevCfg&>(cfg).counters[QS_DEVCTR_SELF_NARROW], 1ull + (ns6 != n_sweeps ? 1ull << 20 : 0ull) + ((unsigned long long)mB << 40));
            } else if (mA == 0 && mB <= 6) core<0, 6>(cfg, w, a, b, rhs, dinv, diag, lam, lo, hi, grp, mA, mB, mu_e, track, thr, n_sweeps);
            else core<NA, NB>(cfg, w, a, b, rhs, dinv, diag, lam, lo, hi, grp, mA, mB, mu_e, track, thr, n_sweeps);
#else
            if (mA == 0 && mB <= 4) core<0, 4>(cfg, w, a, b, rhs, dinv, diag, lam, lo, hi, grp, mA, mB, mu_e, track, thr, n_sweeps);
            else if (mA == 0 && mB <= 6) core<0, 6>(cfg, w, a, b, rhs, dinv, diag, lam, lo, hi, grp, mA, mB, mu_e, track, thr, n_sweeps);
            else core<NA, NB>(cfg, w, a, b, rhs, dinv, diag, lam, lo, hi, grp, mA, mB, mu_e, track, thr, n_sweeps);
#endif
            QS_PHASE_G(43)
#if defined(QS_PROFILE_PHASES) && defined(__HIP_DEVICE_COMPILE__)
            if (threadIdx.x == 0) {   // all workgroups: solves, sweeps, live rows, live contact points (normals)
                atomicAdd(&qs_phase_cycles[0], 1ull); atomicAdd(&qs_phase_cycles[46], (unsigned long long)n_sweeps);
                atomicAdd(&qs_phase_cycles[47], (unsigned long long)(mA + 3 * mB));
                atomicAdd(&qs_phase_cycles[30], (unsigned long long)mB);
            }
#endif
            (void)n_sweeps;
            // ---- the impulses back to the quad: every lane leaves its row's where the quad's lane -- which knows where it sent the row -- finds it
            if (alive) scr[P::LAM_OFF + lane] = lam;
            LaneDev::sync();
            if (slot == e) {
                int dl = A0 + preA, dn = B0 + preB;
#pragma unroll
                for (int j = 0; j < 3; j++) { lam12[9 + j] = lrow[j] ? scr[P::LAM_OFF + dl] : 0.0f; dl += lrow[j] ? 1 : 0; }
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const int af = C0 + 2 * (dn - B0);
                    lam12[3 * c] = nrow[c] ? scr[P::LAM_OFF + dn] : 0.0f;
                    lam12[3 * c + 1] = nrow[c] ? scr[P::LAM_OFF + af] : 0.0f;
                    lam12[3 * c + 2] = nrow[c] ? scr[P::LAM_OFF + af + 1] : 0.0f;
                    dn += nrow[c] ? 1 : 0;
                }
#pragma unroll
                for (int k = 0; k < 6; k++) plam[k] = pay_live ? scr[P::LAM_OFF + A0 + mLim + k] : 0.0f;
            }
            LaneDev::sync();
            QS_PHASE_G(44)
        }
    }
};
#endif

#if !defined(__HIP_DEVICE_COMPILE__)
// ------------------------------------------------------------------ the same solve for the 4-wide host emulation (tests only): one
// environment, plain loops over its rows in the same order
template <bool CONE> struct RareSolver<LaneEmu, CONE> {
    using Ty = SimTypes<LaneEmu>;
    using Row = typename Ty::Row;
    using PayRows = typename Ty::PayRows;
    static void solve(const qs_config& cfg, V4 mu, const Row* xr, const PayRows* pay, M4 mine, V4 warm, float*, V4* lam12, V4* plam) {
        using P = RarePos;
        for (int r = 0; r < 12; r++) lam12[r] = V4(0.0f);
        for (int k = 0; k < 6; k++) plam[k] = V4(0.0f);
        if (!mine.v[0]) return;
        float w[P::N][6], a[P::N][3], b[P::N][3], rhs[P::N], dinv[P::N], diag[P::N], lam[P::N], res[P::N];
        bool live[P::N];
        for (int p = 0; p < P::N; p++) {
            const int r = P::row(p);
            if (p >= P::PAY0 && p < P::NRM0) {
                const int k = p - P::PAY0;
                live[p] = pay != nullptr && pay->act.v[0] > 0.5f;
                for (int i = 0; i < 6; i++) w[p][i] = pay ? pay->w[k][i].v[0] : 0.0f;
                const float rx = pay ? pay->rB.x.v[0] : 0.0f, ry = pay ? pay->rB.y.v[0] : 0.0f, rz = pay ? pay->rB.z.v[0] : 0.0f, mI = pay ? pay->mI.v[0] : 0.0f;
                const float ja[6][3] = {{0, -rz, ry}, {rz, 0, -rx}, {-ry, rx, 0}, {-1, 0, 0}, {0, -1, 0}, {0, 0, -1}};
                for (int i = 0; i < 3; i++) { a[p][i] = ja[k][i]; b[p][i] = ja[k][i] * mI; }
                rhs[p] = pay ? pay->rhs[k].v[0] : 0.0f; dinv[p] = pay ? pay->dinv[k].v[0] : 0.0f; diag[p] = pay ? pay->diag[k].v[0] : 0.0f;
            } else {
                const int L = P::leg(p);
                const Row& q = xr[r];
                live[p] = q.act.v[L] > 0.5f;
                for (int i = 0; i < 6; i++) w[p][i] = q.w[i].v[L];
                for (int i = 0; i < 3; i++) { a[p][i] = q.jq[i].v[L]; b[p][i] = q.u[i].v[L]; }
                rhs[p] = q.rhs.v[L]; dinv[p] = q.dinv.v[L]; diag[p] = q.diag.v[L];
            }
            lam[p] = 0.0f;
        }
        for (int L = 0; L < 4; L++) if (live[P::NRM0 + 3 * L]) lam[P::NRM0 + 3 * L] = warm.v[L];
        float A[P::N][P::N];          // A[j][p] x (-1 / A_pp), self entries zero (11.6 KB on the stack: the emulation must be re-entrant -- threaded tests)
        for (int p = 0; p < P::N; p++)
            for (int j = 0; j < P::N; j++) {
                float t = 0.0f;
                if (live[p] && live[j] && j != p) {
                    const float nd = -dinv[p];
                    t = w[j][0] * (w[p][0] * nd);
                    for (int i = 1; i < 6; i++) t = fmaf(w[j][i], w[p][i] * nd, t);
                    if (P::leg(j) == P::leg(p)) {
                        float u = a[j][0] * (b[p][0] * nd);
                        for (int i = 1; i < 3; i++) u = fmaf(a[j][i], b[p][i] * nd, u);
                        t = t + u;
                    }
                }
                A[j][p] = t;
            }
        for (int p = 0; p < P::N; p++) {
            res[p] = rhs[p];
            for (int L = 0; L < 4; L++) if (live[P::NRM0 + 3 * L]) res[p] = fmaf(A[P::NRM0 + 3 * L][p], lam[P::NRM0 + 3 * L], res[p]);
        }
        const float mu_e = mu.v[0], big = 1e10f, bound = 500.0f * (float)cfg.dt;
        const bool track = cfg.solver_residual_threshold > 0.0f;
        const float thr = sqrtf(cfg.solver_residual_threshold);
        auto apply = [&](int p, float cand) {
            const float d = cand - lam[p];
            lam[p] = cand;
            for (int q = 0; q < P::N; q++) res[q] = fmaf(A[p][q], d, res[q]);
        };
        auto row = [&](int p, float lo, float hi) { if (live[p]) apply(p, fminf(fmaxf(res[p], lo), hi)); };
        for (int it = 0; it < cfg.solver_iters; it++) {
            float lam_in[P::N];
            for (int p = 0; p < P::N; p++) lam_in[p] = lam[p];
            if (it & 1) { for (int p = 0; p < P::PAY0; p++) row(p, 0.0f, big); for (int p = P::PAY0; p < P::NRM0; p++) row(p, -bound, bound); }
            else { for (int p = P::NRM0 - 1; p >= P::PAY0; p--) row(p, -bound, bound); for (int p = P::PAY0 - 1; p >= 0; p--) row(p, 0.0f, big); }
            for (int p = P::NRM0; p < P::FRI0; p++) row(p, 0.0f, big);
            for (int f = P::FRI0; f < P::N; f += 2) {
                if (!live[f]) continue;
                const float ln = lam[P::normal_of(f)], lim = mu_e * ln;
                if (CONE) {
                    const float ca = res[f], cb = res[f + 1];
                    const float r2 = fmaf(cb, cb, fmaf(ca, ca, 1e-30f));
                    const float sc = fminf(lim * (1.0f / sqrtf(r2)), 1.0f);
                    const float na = ca * sc, nb = cb * sc, da = na - lam[f], db = nb - lam[f + 1];
                    lam[f] = na; lam[f + 1] = nb;
                    for (int q = 0; q < P::N; q++) { res[q] = fmaf(A[f][q], da, res[q]); res[q] = fmaf(A[f + 1][q], db, res[q]); }
                } else {
                    for (int t = 0; t < 2; t++) apply(f + t, ln > 0.0f ? fminf(fmaxf(res[f + t], -lim), lim) : lam[f + t]);
                }
            }
            if (track) {
                bool moving = false;
                for (int p = 0; p < P::N; p++) moving = moving || (live[p] && fabsf((lam[p] - lam_in[p]) * diag[p]) > thr);
                if (!moving) break;
            }
        }
        for (int L = 0; L < 4; L++)
            for (int r = 0; r < 12; r++) lam12[r].v[L] = lam[P::of_row(L, r)];
        for (int k = 0; k < 6; k++) plam[k] = V4(lam[P::PAY0 + k]);
    }
};
#endif
