// split_probe.hip -- measurement behind DESIGN.md 10: would splitting one physics substep over TWO waves of a workgroup shorten the
// launch at N = 8192?  ("measure the two-wave split, do not estimate it".)
//
// The substep of csrc/qs_core.h re-assembled from the product's own building blocks (qs_core.h algebra, chol6 / lsolve6, the 4x4x1 MFMA
// Delassus block, the impulse-space cone sweeps of solve_and_integrate) in two mappings over the same 16-environments-per-wave quad layout:
//   mono  : one wave per 16 environments runs everything (the product's mapping; 512 workgroups x 64 lanes at N = 8192)
//   split : two waves per 16 environments (512 workgroups x 128 lanes = 1024 waves, one per SIMD).  Both run the prerequisites (base
//           rotation, leg kinematics, link inertias).  Wave A: RNEA bias + collision (foot distance, heights of the non-foot primitives,
//           joint-limit tests).  Wave B: CRBA, K = D^-1, B K, Schur complement + Cholesky, Jacobian part of the contact rows (u, w, diag)
//           and the Delassus block.  A hands its 9 bias values and the contact flags to B through LDS (s_barrier), B computes the
//           accelerations, the rows' right-hand sides, the sweeps, delta v and the integration, and hands the new state (19 values per
//           lane) back through LDS for A's next substep (second s_barrier).
// Both kernels claim the whole register file of a SIMD (256 VGPR + 256 AGPR, like k_step) so that the dispatcher can put no two waves on
// one SIMD: without that the probe's small kernels are packed two to a SIMD and measure something else.
// Every wave runs ITER dependent substeps (the state feeds back) and wave 0 of workgroup 0 reports cycles per substep; results of the
// two mappings are compared.  `--extra W` adds W more workgroups of the same kind to the launch: the settle lanes of the look-ahead
// resets (185 waves' worth at the benchmark's reset rate), which in the mono mapping run on SIMDs the environments leave idle and in the
// split mapping find every SIMD taken.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -ffinite-math-only -fno-signed-zeros -fno-trapping-math \
//         -mllvm -amdgpu-sched-strategy=iterative-ilp -mllvm -amdgpu-mfma-vgpr-form -I include -o split_probe tools/split_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../quadruped-springs_amd/csrc/qs_core.h"

using namespace qs;
using T = LaneDev;
using V = float;
typedef V3<V> V3f;
typedef Sp<V> Spf;
typedef SI<V> SIf;

#define ITER 40
#define SWEEPS 8          // the benchmark's mix leaves the sweeps after ~7-8 of 30 (solverResidualThreshold 1e-7)
#define NR 3

struct St { V3f pos; V qx, qy, qz, qw; V3f vlin, vang; V q[3], qd[3], warm; };   // 19 + 1 values
struct Pre {                    // what both roles need first
    V R[9]; V3f Rx, Ry, Rz; Spf v0;
    V3f p1, p2, p3, rf, ax1, Y, Z1, X2, Z2, X3, Z3;
    SIf I1, I2, I3; Spf S1, S2, S3;
};
struct BiasOut { V C[3], Cb[6], dist, active, n_invalid; };      // wave A's results
struct Row { V jq[3], u[3], w[6], rhs, dinv, diag; V3f ja, d; };
struct MassOut {                // wave B's results before the join
    V K[6], Bm[3][6], BK[3][6], Sm[21], Ld[6];
    Row rows[NR];
    V Ap[4 * NR][NR];
};

__device__ __forceinline__ void prereq(const St& s, Pre& P) {
    using namespace go1;
    const V one = 1.0f, zero = 0.0f;
    V fx = T::fx(), sy = T::sy();
    V x = s.qx, y = s.qy, z = s.qz, w = s.qw;
    V sc = 2.0f * qrcp(x * x + y * y + z * z + w * w);
    V xs = x * sc, ys = y * sc, zs = z * sc;
    V wx = w * xs, wy = w * ys, wz = w * zs, xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
    V* R = P.R;
    R[0] = one - (yy + zz); R[1] = xy - wz; R[2] = xz + wy; R[3] = xy + wz; R[4] = one - (xx + zz); R[5] = yz - wx; R[6] = xz - wy; R[7] = yz + wx; R[8] = one - (xx + yy);
    P.Rx = mk3<V>(R[0], R[1], R[2]); P.Ry = mk3<V>(R[3], R[4], R[5]); P.Rz = mk3<V>(R[6], R[7], R[8]);
    P.v0.a = mk3<V>(R[0] * s.vang.x + R[3] * s.vang.y + R[6] * s.vang.z, R[1] * s.vang.x + R[4] * s.vang.y + R[7] * s.vang.z, R[2] * s.vang.x + R[5] * s.vang.y + R[8] * s.vang.z);
    P.v0.l = mk3<V>(R[0] * s.vlin.x + R[3] * s.vlin.y + R[6] * s.vlin.z, R[1] * s.vlin.x + R[4] * s.vlin.y + R[7] * s.vlin.z, R[2] * s.vlin.x + R[5] * s.vlin.y + R[8] * s.vlin.z);
    V s1, c1, s2, c2, s23, c23;
    qsincos(s.q[0], s1, c1); qsincos(s.q[1], s2, c2); qsincos(s.q[1] + s.q[2], s23, c23);
    P.p1 = mk3<V>(fx * HIP_X, sy * HIP_Y, zero); P.ax1 = mk3<V>(one, zero, zero);
    P.Y = mk3<V>(zero, c1, s1); P.Z1 = mk3<V>(zero, -s1, c1);
    P.p2 = P.p1 + P.Y * (sy * THIGH_Y);
    P.X2 = mk3<V>(c2, s1 * s2, -c1 * s2); P.Z2 = mk3<V>(s2, -s1 * c2, c1 * c2);
    P.p3 = P.p2 + P.Z2 * V(LEG_Z);
    P.X3 = mk3<V>(c23, s1 * s23, -c1 * s23); P.Z3 = mk3<V>(s23, -s1 * c23, c1 * c23);
    P.rf = P.p3 + P.Z3 * V(LEG_Z);
    S3<V> Ih, It, Ic;
    Ih.xx = HIP_M * HIP_I[0] / HIP_M; Ih.xy = fx * sy * HIP_I[1]; Ih.xz = -fx * HIP_I[2]; Ih.yy = HIP_I[3]; Ih.yz = -sy * HIP_I[4]; Ih.zz = HIP_I[5];
    It.xx = THIGH_I[0]; It.xy = sy * THIGH_I[1]; It.xz = THIGH_I[2]; It.yy = THIGH_I[3]; It.yz = sy * THIGH_I[4]; It.zz = THIGH_I[5];
    Ic.xx = CALF_I[0]; Ic.xy = CALF_I[1]; Ic.xz = CALF_I[2]; Ic.yy = CALF_I[3]; Ic.yz = CALF_I[4]; Ic.zz = CALF_I[5];
    P.I1 = part_inertia<V>(HIP_M, mk3<V>(fx * (-HIP_C[0]), sy * (-HIP_C[1]), HIP_C[2]), Ih, P.p1, P.ax1, P.Y, P.Z1);
    P.I2 = part_inertia<V>(THIGH_M, mk3<V>(THIGH_C[0], sy * (-THIGH_C[1]), THIGH_C[2]), It, P.p2, P.X2, P.Y, P.Z2);
    P.I3 = part_inertia<V>(CALF_M, mk3<V>(CALF_C[0], CALF_C[1], CALF_C[2]), Ic, P.p3, P.X3, P.Y, P.Z3) + point_inertia<V>(FOOT_M, FOOT_I, P.rf);
    P.S1.a = P.ax1; P.S1.l = cross(P.p1, P.ax1); P.S2.a = P.Y; P.S2.l = cross(P.p2, P.Y); P.S3.a = P.Y; P.S3.l = cross(P.p3, P.Y);
}

// wave A: RNEA bias (qdd = 0, gravity as the fictitious base acceleration) + collision
__device__ __forceinline__ void role_bias(const St& s, const Pre& P, const SIf& I0, BiasOut& o) {
    using namespace go1;
    const V zero = 0.0f, one = 1.0f;
    Spf a0; a0.a = mk3<V>(zero, zero, zero); a0.l = P.Rz * V(9.8f);
    Spf vj1, vj2, vj3;
    vj1.a = P.S1.a * s.qd[0]; vj1.l = P.S1.l * s.qd[0]; vj2.a = P.S2.a * s.qd[1]; vj2.l = P.S2.l * s.qd[1]; vj3.a = P.S3.a * s.qd[2]; vj3.l = P.S3.l * s.qd[2];
    Spf v1 = P.v0 + vj1, v2 = v1 + vj2, v3 = v2 + vj3;
    Spf a1 = crm_add(a0, P.v0, vj1), a2 = crm_add(a1, v1, vj2), a3 = crm_add(a2, v2, vj3);
    Spf f1 = crf_add(apply(P.I1, a1), v1, apply(P.I1, v1));
    Spf f2 = crf_add(apply(P.I2, a2), v2, apply(P.I2, v2));
    Spf f3 = crf_add(apply(P.I3, a3), v3, apply(P.I3, v3));
    Spf fs2 = f2 + f3, fs1 = f1 + fs2;
    o.C[0] = dot(P.S1, fs1); o.C[1] = dot(P.S2, fs2); o.C[2] = dot(P.S3, f3);
    Spf f0 = crf_add(apply(I0, a0), P.v0, apply(I0, P.v0));
    o.Cb[0] = T::quad_sum(fs1.a.x) + f0.a.x; o.Cb[1] = T::quad_sum(fs1.a.y) + f0.a.y; o.Cb[2] = T::quad_sum(fs1.a.z) + f0.a.z;
    o.Cb[3] = T::quad_sum(fs1.l.x) + f0.l.x; o.Cb[4] = T::quad_sum(fs1.l.y) + f0.l.y; o.Cb[5] = T::quad_sum(fs1.l.z) + f0.l.z;
    // collision: foot sphere vs plane; lowest vertices of trunk corner / hip housing / thigh ends / calf knee end; joint limits
    V fx = T::fx(), sy = T::sy();
    const V zc = s.pos.z;
    o.dist = zc + dot(P.Rz, P.rf) - FOOT_R;
    o.active = qflag(o.dist < THR_FOOT);
    V az = dot(P.Rz, P.Y), gx2 = dot(P.Rz, P.X2), gx3 = dot(P.Rz, P.X3), gp2 = dot(P.Rz, P.p2), gp3 = dot(P.Rz, P.p3);
    V h_trunk = zc + P.Rz.x * (fx * TRUNK_HALF[0]) + P.Rz.y * (sy * TRUNK_HALF[1]) - qabs(P.Rz.z) * TRUNK_HALF[2];
    V h_hip = zc + dot(P.Rz, P.p1) - HIP_CYL_HALF_LEN * qabs(az) - HIP_CYL_R * qsqrt(qmax(one - az * az, zero));
    V th_off = qabs(gx2) * THIGH_HALF[0] + qabs(az) * THIGH_HALF[1], cf_off = (qabs(gx3) + qabs(az)) * CALF_HALF[0];
    V h_th = qmin(zc + gp2 - th_off, zc + gp3 - th_off), h_cf = qmin(zc + gp3 - cf_off, zc + dot(P.Rz, P.rf) - cf_off);
    V n = qflag(h_hip < THR_HIP) + qflag(h_th < THR_THIGH) + qflag(h_cf < THR_CALF);
    o.n_invalid = T::quad_sum(n) + qflag(T::quad_sum(qflag(h_trunk < THR_TRUNK)) > zero);
    V lim = zero;
#pragma unroll
    for (int j = 0; j < 3; j++) lim = lim + qflag(s.q[j] - JLO[j] <= zero) + qflag(V(JHI[j]) - s.q[j] <= zero);
    o.n_invalid = o.n_invalid + lim;
}

// wave B before the join: CRBA, K, B K, Schur + Cholesky, Jacobian part of the three contact rows, Delassus block
__device__ __forceinline__ void role_mass(const St& s, const Pre& P, const SIf& I0, V mtot, MassOut& m) {
    using namespace go1;
    const V zero = 0.0f;
    SIf Ic2 = P.I2 + P.I3, Ic1 = P.I1 + Ic2;
    Spf F1 = apply(Ic1, P.S1), F2 = apply(Ic2, P.S2), F3 = apply(P.I3, P.S3);
    V D11 = dot(P.S1, F1), D12 = dot(P.S1, F2), D13 = dot(P.S1, F3), D22 = dot(P.S2, F2), D23 = dot(P.S2, F3), D33 = dot(P.S3, F3);
    V c11 = D22 * D33 - D23 * D23, c12 = D13 * D23 - D12 * D33, c13 = D12 * D23 - D13 * D22;
    V id = qrcp(D11 * c11 + D12 * c12 + D13 * c13);
    V K11 = c11 * id, K12 = c12 * id, K13 = c13 * id, K22 = (D11 * D33 - D13 * D13) * id, K23 = (D12 * D13 - D11 * D23) * id, K33 = (D11 * D22 - D12 * D12) * id;
    m.K[0] = K11; m.K[1] = K12; m.K[2] = K13; m.K[3] = K22; m.K[4] = K23; m.K[5] = K33;
    V Bm[3][6] = {{F1.a.x, F1.a.y, F1.a.z, F1.l.x, F1.l.y, F1.l.z}, {F2.a.x, F2.a.y, F2.a.z, F2.l.x, F2.l.y, F2.l.z}, {F3.a.x, F3.a.y, F3.a.z, F3.l.x, F3.l.y, F3.l.z}};
#pragma unroll
    for (int i = 0; i < 6; i++) {
        m.Bm[0][i] = Bm[0][i]; m.Bm[1][i] = Bm[1][i]; m.Bm[2][i] = Bm[2][i];
        m.BK[0][i] = Bm[0][i] * K11 + Bm[1][i] * K12 + Bm[2][i] * K13;
        m.BK[1][i] = Bm[0][i] * K12 + Bm[1][i] * K22 + Bm[2][i] * K23;
        m.BK[2][i] = Bm[0][i] * K13 + Bm[1][i] * K23 + Bm[2][i] * K33;
    }
    SIf It; It.m = mtot;
    It.h = mk3<V>(T::quad_sum(Ic1.h.x), T::quad_sum(Ic1.h.y), T::quad_sum(Ic1.h.z)) + I0.h;
    It.I.xx = T::quad_sum(Ic1.I.xx) + I0.I.xx; It.I.xy = T::quad_sum(Ic1.I.xy) + I0.I.xy; It.I.xz = T::quad_sum(Ic1.I.xz) + I0.I.xz;
    It.I.yy = T::quad_sum(Ic1.I.yy) + I0.I.yy; It.I.yz = T::quad_sum(Ic1.I.yz) + I0.I.yz; It.I.zz = T::quad_sum(Ic1.I.zz) + I0.I.zz;
    V* Sm = m.Sm;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) Sm[tri(i, j)] = T::quad_sum(m.BK[0][i] * Bm[0][j] + m.BK[1][i] * Bm[1][j] + m.BK[2][i] * Bm[2][j]);
    V H[21];
    H[tri(0, 0)] = It.I.xx; H[tri(1, 0)] = It.I.xy; H[tri(1, 1)] = It.I.yy; H[tri(2, 0)] = It.I.xz; H[tri(2, 1)] = It.I.yz; H[tri(2, 2)] = It.I.zz;
    H[tri(3, 0)] = zero; H[tri(3, 1)] = It.h.z; H[tri(3, 2)] = -It.h.y; H[tri(4, 0)] = -It.h.z; H[tri(4, 1)] = zero; H[tri(4, 2)] = It.h.x;
    H[tri(5, 0)] = It.h.y; H[tri(5, 1)] = -It.h.x; H[tri(5, 2)] = zero;
    H[tri(3, 3)] = It.m; H[tri(4, 3)] = zero; H[tri(4, 4)] = It.m; H[tri(5, 3)] = zero; H[tri(5, 4)] = zero; H[tri(5, 5)] = It.m;
#pragma unroll
    for (int i = 0; i < 21; i++) Sm[i] = H[i] - Sm[i];
    chol6<V>(Sm, m.Ld);
    // contact rows (every foot treated as a contact candidate: the direction is scaled by the flag after the join)
    V3f rc = P.rf - P.Rz * V(FOOT_R);
    V3f d1 = rc - P.p1, d2 = rc - P.p2, d3 = rc - P.p3;
    V3f g1 = cross(P.ax1, d1), g2 = cross(P.Y, d2), g3 = cross(P.Y, d3);
    V3f dirs[3] = {P.Rz, mk3<V>(-P.Ry.x, -P.Ry.y, -P.Ry.z), P.Rx};
#pragma unroll
    for (int r = 0; r < NR; r++) {
        Row& R_ = m.rows[r];
        V3f d_ = dirs[r];
        V3f ja = cross(rc, d_);
        R_.ja = ja; R_.d = d_;
        R_.jq[0] = dot(d_, g1); R_.jq[1] = dot(d_, g2); R_.jq[2] = dot(d_, g3);
        R_.u[0] = K11 * R_.jq[0] + K12 * R_.jq[1] + K13 * R_.jq[2];
        R_.u[1] = K12 * R_.jq[0] + K22 * R_.jq[1] + K23 * R_.jq[2];
        R_.u[2] = K13 * R_.jq[0] + K23 * R_.jq[1] + K33 * R_.jq[2];
        V jb[6] = {ja.x, ja.y, ja.z, d_.x, d_.y, d_.z};
#pragma unroll
        for (int i = 0; i < 6; i++) { V t_ = jb[i] - Bm[0][i] * R_.u[0]; t_ = t_ - Bm[1][i] * R_.u[1]; R_.w[i] = t_ - Bm[2][i] * R_.u[2]; }
        lsolve6<V>(Sm, m.Ld, R_.w);
        V diag = R_.jq[0] * R_.u[0] + R_.jq[1] * R_.u[1] + R_.jq[2] * R_.u[2];
#pragma unroll
        for (int i = 0; i < 6; i++) diag = diag + R_.w[i] * R_.w[i];
        R_.dinv = qrcp(qmax(diag, V(1e-30f))); R_.diag = diag;
    }
    // Delassus columns of the own rows (54 MFMAs), as in Sim::solve_and_integrate
    V wc[NR][6], locs[NR][NR];
#pragma unroll
    for (int c = 0; c < NR; c++) {
        V nd = -m.rows[c].dinv;
#pragma unroll
        for (int i = 0; i < 6; i++) wc[c][i] = m.rows[c].w[i] * nd;
#pragma unroll
        for (int r = 0; r < NR; r++) locs[r][c] = (m.rows[r].jq[0] * m.rows[c].u[0] + m.rows[r].jq[1] * m.rows[c].u[1] + m.rows[r].jq[2] * m.rows[c].u[2]) * nd;
    }
    T::Acc4 acc[NR][NR];
#pragma unroll
    for (int r = 0; r < NR; r++)
#pragma unroll
        for (int c = 0; c < NR; c++) acc[r][c] = T::acc4_zero();
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int r = 0; r < NR; r++)
#pragma unroll
            for (int c = 0; c < NR; c++) T::outer_fma(m.rows[r].w[i], wc[c][i], acc[r][c]);
#define SCATTER(Kq)                                                                                   \
    {                                                                                                 \
        bool own = T::is_leg(Kq); V ownf = qflag(own);                                                \
        _Pragma("unroll") for (int r = 0; r < NR; r++) _Pragma("unroll") for (int c = 0; c < NR; c++) { \
            V a_ = T::template acc4_get<Kq>(acc[r][c]) + ownf * locs[r][c];                            \
            m.Ap[NR * Kq + r][c] = (r == c) ? qsel(own, V(0.0f), a_) : a_;                            \
        }                                                                                             \
    }
    SCATTER(0) SCATTER(1) SCATTER(2) SCATTER(3)
#undef SCATTER
}

// after the join: accelerations, v*, right-hand sides, SWEEPS cone sweeps, delta v, integration
__device__ __forceinline__ void tail(St& s, const Pre& P, const BiasOut& b, const MassOut& m, const V* tau, V mu, float dt) {
    const V zero = 0.0f, one = 1.0f, big = 1e10f, cap = 30.1f;
    const V K11 = m.K[0], K12 = m.K[1], K13 = m.K[2], K22 = m.K[3], K23 = m.K[4], K33 = m.K[5];
    V t1 = tau[0] - b.C[0], t2 = tau[1] - b.C[1], t3 = tau[2] - b.C[2];
    V y1 = K11 * t1 + K12 * t2 + K13 * t3, y2 = K12 * t1 + K22 * t2 + K23 * t3, y3 = K13 * t1 + K23 * t2 + K33 * t3;
    V ab[6];
#pragma unroll
    for (int i = 0; i < 6; i++) ab[i] = -b.Cb[i] - T::quad_sum(m.Bm[0][i] * y1 + m.Bm[1][i] * y2 + m.Bm[2][i] * y3);
    lsolve6<V>(m.Sm, m.Ld, ab);
    ltsolve6<V>(m.Sm, m.Ld, ab);
    V qdd[3] = {y1, y2, y3};
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 6; i++) qdd[j] = qdd[j] - m.BK[j][i] * ab[i];
    const V* R = P.R;
    V3f wxv = cross(P.v0.a, P.v0.l);
    V3f al = mk3<V>(ab[3] + wxv.x, ab[4] + wxv.y, ab[5] + wxv.z);
    s.vang.x = clampv<V>(s.vang.x + dt * (R[0] * ab[0] + R[1] * ab[1] + R[2] * ab[2]), -cap, cap);
    s.vang.y = clampv<V>(s.vang.y + dt * (R[3] * ab[0] + R[4] * ab[1] + R[5] * ab[2]), -cap, cap);
    s.vang.z = clampv<V>(s.vang.z + dt * (R[6] * ab[0] + R[7] * ab[1] + R[8] * ab[2]), -cap, cap);
    s.vlin.x = clampv<V>(s.vlin.x + dt * (R[0] * al.x + R[1] * al.y + R[2] * al.z), -cap, cap);
    s.vlin.y = clampv<V>(s.vlin.y + dt * (R[3] * al.x + R[4] * al.y + R[5] * al.z), -cap, cap);
    s.vlin.z = clampv<V>(s.vlin.z + dt * (R[6] * al.x + R[7] * al.y + R[8] * al.z), -cap, cap);
#pragma unroll
    for (int j = 0; j < 3; j++) s.qd[j] = clampv<V>(s.qd[j] + dt * qdd[j], -cap, cap);
    Spf vs;
    vs.a = mk3<V>(R[0] * s.vang.x + R[3] * s.vang.y + R[6] * s.vang.z, R[1] * s.vang.x + R[4] * s.vang.y + R[7] * s.vang.z, R[2] * s.vang.x + R[5] * s.vang.y + R[8] * s.vang.z);
    vs.l = mk3<V>(R[0] * s.vlin.x + R[3] * s.vlin.y + R[6] * s.vlin.z, R[1] * s.vlin.x + R[4] * s.vlin.y + R[7] * s.vlin.z, R[2] * s.vlin.x + R[5] * s.vlin.y + R[8] * s.vlin.z);
    const V inv_dt = qrcp(V(dt));
    V res[NR], lam_all[4 * NR];
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const Row& R_ = m.rows[r];
        V rel = R_.ja.x * vs.a.x + R_.ja.y * vs.a.y + R_.ja.z * vs.a.z + R_.d.x * vs.l.x + R_.d.y * vs.l.y + R_.d.z * vs.l.z + R_.jq[0] * s.qd[0] + R_.jq[1] * s.qd[1] + R_.jq[2] * s.qd[2];
        if (r == 0) {
            V pen = b.dist + 1e-5f;
            V pos_err = qsel(pen > zero, zero, (-pen) * (0.08f * inv_dt));
            V vel_err = (-rel) - qsel(pen > zero, pen * inv_dt, zero);
            res[r] = (pos_err + vel_err) * R_.dinv * b.active;
        } else res[r] = (-rel) * R_.dinv * b.active;
    }
#pragma unroll
    for (int i = 0; i < 4 * NR; i++) lam_all[i] = zero;
    {
        V l_own = s.warm * 0.1f * b.active;
        lam_all[0] = T::bcast<0>(l_own); lam_all[NR] = T::bcast<1>(l_own); lam_all[2 * NR] = T::bcast<2>(l_own); lam_all[3 * NR] = T::bcast<3>(l_own);
#pragma unroll
        for (int c = 0; c < NR; c++) res[c] = res[c] + m.Ap[0][c] * lam_all[0] + m.Ap[NR][c] * lam_all[NR] + m.Ap[2 * NR][c] * lam_all[2 * NR] + m.Ap[3 * NR][c] * lam_all[3 * NR];
    }
    for (int it = 0; it < SWEEPS; it++) {
#define ROWN(Kq)                                                                                      \
    {                                                                                                 \
        constexpr int i_ = NR * (Kq);                                                                 \
        V cand = qmed3(res[0], zero, big);                                                            \
        V dk = T::template bcast<Kq>(cand) - lam_all[i_];                                             \
        lam_all[i_] = lam_all[i_] + dk;                                                               \
        T::fma2(m.Ap[i_][0], m.Ap[i_][1], dk, res[0], res[1]); res[2] = res[2] + m.Ap[i_][2] * dk;    \
    }
#define PAIR(Kq)                                                                                      \
    {                                                                                                 \
        constexpr int ia_ = NR * (Kq) + 1, ib_ = NR * (Kq) + 2;                                       \
        V lim = mu * lam_all[NR * (Kq)];                                                              \
        V r2 = res[1] * res[1] + res[2] * res[2];                                                     \
        V sc = qmin(lim * qrsqrt(qmax(r2, V(1e-30f))), one);                                          \
        V da = T::template bcast<Kq>(res[1] * sc) - lam_all[ia_], db = T::template bcast<Kq>(res[2] * sc) - lam_all[ib_]; \
        lam_all[ia_] = lam_all[ia_] + da; lam_all[ib_] = lam_all[ib_] + db;                           \
        _Pragma("unroll") for (int c = 0; c < NR; c++) { res[c] = res[c] + m.Ap[ia_][c] * da; res[c] = res[c] + m.Ap[ib_][c] * db; } \
    }
        ROWN(0) ROWN(1) ROWN(2) ROWN(3) PAIR(0) PAIR(1) PAIR(2) PAIR(3)
#undef ROWN
#undef PAIR
    }
    V lam_own[NR];
#pragma unroll
    for (int c = 0; c < NR; c++)
        lam_own[c] = qsel(T::is_leg(0), lam_all[c], qsel(T::is_leg(1), lam_all[NR + c], qsel(T::is_leg(2), lam_all[2 * NR + c], lam_all[3 * NR + c])));
    s.warm = lam_own[0];
    V z[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        V t = m.rows[0].w[i] * lam_own[0];
#pragma unroll
        for (int r = 1; r < NR; r++) t = t + m.rows[r].w[i] * lam_own[r];
        z[i] = T::quad_sum(t);
    }
    ltsolve6<V>(m.Sm, m.Ld, z);
#pragma unroll
    for (int j = 0; j < 3; j++) {
        V t = m.rows[0].u[j] * lam_own[0];
#pragma unroll
        for (int r = 1; r < NR; r++) t = t + m.rows[r].u[j] * lam_own[r];
#pragma unroll
        for (int i = 0; i < 6; i++) t = t - m.BK[j][i] * z[i];
        s.qd[j] = clampv<V>(s.qd[j] + t, -cap, cap);
    }
    s.vang.x = clampv<V>(s.vang.x + R[0] * z[0] + R[1] * z[1] + R[2] * z[2], -cap, cap);
    s.vang.y = clampv<V>(s.vang.y + R[3] * z[0] + R[4] * z[1] + R[5] * z[2], -cap, cap);
    s.vang.z = clampv<V>(s.vang.z + R[6] * z[0] + R[7] * z[1] + R[8] * z[2], -cap, cap);
    s.vlin.x = clampv<V>(s.vlin.x + R[0] * z[3] + R[1] * z[4] + R[2] * z[5], -cap, cap);
    s.vlin.y = clampv<V>(s.vlin.y + R[3] * z[3] + R[4] * z[4] + R[5] * z[5], -cap, cap);
    s.vlin.z = clampv<V>(s.vlin.z + R[6] * z[3] + R[7] * z[4] + R[8] * z[5], -cap, cap);
    s.pos.x = s.pos.x + dt * s.vlin.x; s.pos.y = s.pos.y + dt * s.vlin.y; s.pos.z = s.pos.z + dt * s.vlin.z;
    V th2 = dot(s.vang, s.vang) * (dt * dt);
    V sc = V(0.5f * dt) * (one - th2 * (1.0f / 24.0f) * (one - th2 * (1.0f / 80.0f)));
    V dw = one - th2 * 0.125f * (one - th2 * (1.0f / 48.0f));
    V dx = s.vang.x * sc, dy = s.vang.y * sc, dz = s.vang.z * sc;
    V nx = dw * s.qx + dx * s.qw + dy * s.qz - dz * s.qy, ny = dw * s.qy - dx * s.qz + dy * s.qw + dz * s.qx;
    V nz = dw * s.qz + dx * s.qy - dy * s.qx + dz * s.qw, nw = dw * s.qw - dx * s.qx - dy * s.qy - dz * s.qz;
    V inv = qrsqrt(nx * nx + ny * ny + nz * nz + nw * nw);
    s.qx = nx * inv; s.qy = ny * inv; s.qz = nz * inv; s.qw = nw * inv;
#pragma unroll
    for (int j = 0; j < 3; j++) s.q[j] = s.q[j] + dt * s.qd[j];
}

__device__ __forceinline__ void load_state(const float* x, St& s) {
    s.pos = mk3<V>(x[0], x[1], x[2]); s.qx = x[3]; s.qy = x[4]; s.qz = x[5]; s.qw = x[6];
    s.vlin = mk3<V>(x[7], x[8], x[9]); s.vang = mk3<V>(x[10], x[11], x[12]);
    const int L = threadIdx.x & 3;
    for (int j = 0; j < 3; j++) { s.q[j] = x[13 + 3 * L + j]; s.qd[j] = x[25 + 3 * L + j]; }
    s.warm = 0.0f;
}
__device__ __forceinline__ void pd_torque(const St& s, V* tau) {
    const float q0[3] = {0.0f, 0.78539816f, -1.5707963f};
    for (int j = 0; j < 3; j++) tau[j] = clampv<V>(-60.0f * (s.q[j] - q0[j]) - 1.5f * s.qd[j], -35.0f, 35.0f);
}
__device__ __forceinline__ SIf base_inertia(V& mtot) {
    using namespace go1;
    SIf I0 = point_inertia<V>(TRUNK_M, 0.05f, mk3<V>(TRUNK_CX, 0.0f, TRUNK_CZ));
    mtot = I0.m + 4.0f * (HIP_M + THIGH_M + CALF_M + FOOT_M);
    return I0;
}
struct Result { float st[20]; };
__device__ __forceinline__ void store_result(Result& r, const St& s) {
    float* o = r.st;
    o[0] = s.pos.x; o[1] = s.pos.y; o[2] = s.pos.z; o[3] = s.qx; o[4] = s.qy; o[5] = s.qz; o[6] = s.qw; o[7] = s.vlin.x; o[8] = s.vlin.y; o[9] = s.vlin.z;
    o[10] = s.vang.x; o[11] = s.vang.y; o[12] = s.vang.z; for (int j = 0; j < 3; j++) { o[13 + j] = s.q[j]; o[16 + j] = s.qd[j]; } o[19] = s.warm;
}

// ------------------------------------------------------------------ mono: one wave does everything
__global__ __launch_bounds__(64, 1) void k_mono(const float* __restrict__ in, Result* __restrict__ out, unsigned long long* cycles, int n_wg) {
    asm volatile("" ::: "v255", "a255");               // one wave per SIMD, as the product kernel
    const int wg = blockIdx.x % n_wg;                  // extra workgroups (settle lanes) repeat the environments' inputs
    const int env = wg * 16 + (threadIdx.x >> 2);
    St s; load_state(in + (size_t)env * 40, s);
    V mtot; const SIf I0 = base_inertia(mtot);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; it++) {
        V tau[3]; pd_torque(s, tau);
        Pre P; prereq(s, P);
        BiasOut b; role_bias(s, P, I0, b);
        MassOut m; role_mass(s, P, I0, mtot, m);
        tail(s, P, b, m, tau, V(0.8f), 1e-3f);
        T::opaque(s.q[0]);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((int)blockIdx.x < n_wg) store_result(out[blockIdx.x * 64 + threadIdx.x], s);
    if (blockIdx.x == 0 && threadIdx.x == 0) cycles[0] = (t1 - t0) / ITER;
}

// ------------------------------------------------------------------ split: wave 0 = A (bias + collision), wave 1 = B (mass matrix side, then everything after the join)
__global__ __launch_bounds__(128, 1) void k_split(const float* __restrict__ in, Result* __restrict__ out, unsigned long long* cycles, int n_wg) {
    asm volatile("" ::: "v255", "a255");
    __shared__ float x_bias[64 * 12];      // A -> B: C[3], Cb[6], dist, active, n_invalid
    __shared__ float x_state[64 * 20];     // B -> A: the new state
    const int wg = blockIdx.x % n_wg, lane = threadIdx.x & 63;
    const bool roleB = threadIdx.x >= 64;
    const int env = wg * 16 + (lane >> 2);
    St s; load_state(in + (size_t)env * 40, s);
    V mtot; const SIf I0 = base_inertia(mtot);
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long acc_own = 0, acc_wait = 0, acc_tail = 0;      // wave's own work up to the join, wait at the join, B: after the join
    for (int it = 0; it < ITER; it++) {
        const unsigned long long ta = __builtin_readcyclecounter();
        Pre P; prereq(s, P);
        if (!roleB) {
            BiasOut b; role_bias(s, P, I0, b);
            float* xb = x_bias + lane;          // [value][lane]: conflict-free
#pragma unroll
            for (int k = 0; k < 3; k++) xb[64 * k] = b.C[k];
#pragma unroll
            for (int k = 0; k < 6; k++) xb[64 * (3 + k)] = b.Cb[k];
            xb[64 * 9] = b.dist; xb[64 * 10] = b.active; xb[64 * 11] = b.n_invalid;
            const unsigned long long tb = __builtin_readcyclecounter();
            __syncthreads();                    // join: B picks the bias up
            __syncthreads();                    // B has finished the substep
            acc_own += tb - ta; acc_wait += __builtin_readcyclecounter() - tb;
            const float* xs = x_state + lane;
            s.pos = mk3<V>(xs[0], xs[64], xs[128]); s.qx = xs[192]; s.qy = xs[256]; s.qz = xs[320]; s.qw = xs[384];
            s.vlin = mk3<V>(xs[448], xs[512], xs[576]); s.vang = mk3<V>(xs[640], xs[704], xs[768]);
#pragma unroll
            for (int j = 0; j < 3; j++) { s.q[j] = xs[64 * (13 + j)]; s.qd[j] = xs[64 * (16 + j)]; }
            s.warm = xs[64 * 19];
        } else {
            V tau[3]; pd_torque(s, tau);
            MassOut m; role_mass(s, P, I0, mtot, m);
            T::opaque(m.Ap[0][0]);
            const unsigned long long tb = __builtin_readcyclecounter();
            __syncthreads();                    // join
            const unsigned long long tc = __builtin_readcyclecounter();
            BiasOut b; const float* xb = x_bias + lane;
#pragma unroll
            for (int k = 0; k < 3; k++) b.C[k] = xb[64 * k];
#pragma unroll
            for (int k = 0; k < 6; k++) b.Cb[k] = xb[64 * (3 + k)];
            b.dist = xb[64 * 9]; b.active = xb[64 * 10]; b.n_invalid = xb[64 * 11];
            tail(s, P, b, m, tau, V(0.8f), 1e-3f);
            float* xs = x_state + lane;
            xs[0] = s.pos.x; xs[64] = s.pos.y; xs[128] = s.pos.z; xs[192] = s.qx; xs[256] = s.qy; xs[320] = s.qz; xs[384] = s.qw;
            xs[448] = s.vlin.x; xs[512] = s.vlin.y; xs[576] = s.vlin.z; xs[640] = s.vang.x; xs[704] = s.vang.y; xs[768] = s.vang.z;
#pragma unroll
            for (int j = 0; j < 3; j++) { xs[64 * (13 + j)] = s.q[j]; xs[64 * (16 + j)] = s.qd[j]; }
            xs[64 * 19] = s.warm;
            __syncthreads();
            acc_own += tb - ta; acc_wait += tc - tb; acc_tail += __builtin_readcyclecounter() - tc;
        }
        T::opaque(s.q[0]);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (roleB && (int)blockIdx.x < n_wg) store_result(out[blockIdx.x * 64 + lane], s);
    if (blockIdx.x == 0 && threadIdx.x == 64) { cycles[0] = (t1 - t0) / ITER; cycles[1] = acc_own / ITER; cycles[2] = acc_wait / ITER; cycles[3] = acc_tail / ITER; }
    if (blockIdx.x == 0 && threadIdx.x == 0) { cycles[4] = acc_own / ITER; cycles[5] = acc_wait / ITER; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    int n_env = 8192, extra = 0;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--extra") && i + 1 < argc) extra = atoi(argv[++i]);
        if (!strcmp(argv[i], "--envs") && i + 1 < argc) n_env = atoi(argv[++i]);
    }
    const int n_wg = n_env / 16;
    std::vector<float> in((size_t)n_env * 40, 0.0f);
    srand(1);
    auto u = [] { return (float)rand() / RAND_MAX * 2.0f - 1.0f; };
    for (int e = 0; e < n_env; e++) {       // standing robots, feet on or just above the ground, small velocities
        float* x = &in[(size_t)e * 40];
        x[2] = 0.30f + 0.02f * u(); x[3] = 0.02f * u(); x[4] = 0.02f * u(); x[5] = 0.0f; x[6] = 1.0f;
        for (int k = 0; k < 6; k++) x[7 + k] = 0.1f * u();
        for (int L = 0; L < 4; L++) {
            x[13 + 3 * L] = 0.05f * u(); x[14 + 3 * L] = 0.785f + 0.1f * u(); x[15 + 3 * L] = -1.57f + 0.1f * u();
            for (int j = 0; j < 3; j++) x[25 + 3 * L + j] = 0.5f * u();
        }
    }
    float* d_in; Result *d_m, *d_s; unsigned long long* d_c;
    CK(hipMalloc(&d_in, in.size() * 4)); CK(hipMalloc(&d_m, (size_t)n_env * 4 * sizeof(Result))); CK(hipMalloc(&d_s, (size_t)n_env * 4 * sizeof(Result)));
    CK(hipMalloc(&d_c, 64));
    CK(hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long cm = 0, cs = 0, det[8] = {0}; float msm = 0, mss = 0;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_mono, dim3(n_wg + extra), dim3(64), 0, 0, d_in, d_m, d_c, n_wg); hipEventRecord(e1);
        CK(hipDeviceSynchronize()); hipEventElapsedTime(&msm, e0, e1); CK(hipMemcpy(&cm, d_c, 8, hipMemcpyDeviceToHost));
        hipEventRecord(e0); hipLaunchKernelGGL(k_split, dim3(n_wg + extra), dim3(128), 0, 0, d_in, d_s, d_c, n_wg); hipEventRecord(e1);
        CK(hipDeviceSynchronize()); hipEventElapsedTime(&mss, e0, e1); CK(hipMemcpy(det, d_c, 48, hipMemcpyDeviceToHost)); cs = det[0];
    }
    std::vector<Result> rm((size_t)n_env * 4), rs((size_t)n_env * 4);
    CK(hipMemcpy(rm.data(), d_m, rm.size() * sizeof(Result), hipMemcpyDeviceToHost));
    CK(hipMemcpy(rs.data(), d_s, rs.size() * sizeof(Result), hipMemcpyDeviceToHost));
    double worst = 0, scale = 0; int nan = 0;
    for (size_t i = 0; i < rm.size(); i++)
        for (int k = 0; k < 20; k++) {
            if (rm[i].st[k] != rm[i].st[k]) nan++;
            worst = fmax(worst, fabs((double)rm[i].st[k] - rs[i].st[k])); scale = fmax(scale, fabs((double)rm[i].st[k]));
        }
    printf("{\"n_env\": %d, \"substeps\": %d, \"sweeps\": %d, \"extra_workgroups\": %d, \"mono\": {\"waves\": %d, \"cycles_per_substep\": %llu, \"kernel_ms\": %.4f, \"us_per_substep\": %.3f}, "
           "\"split\": {\"waves\": %d, \"cycles_per_substep\": %llu, \"kernel_ms\": %.4f, \"us_per_substep\": %.3f, \"B_before_join\": %llu, \"B_waits_for_A\": %llu, \"B_after_join\": %llu, \"A_before_join\": %llu, \"A_waits_for_B\": %llu}, \"max_abs_difference\": %.3e, \"max_abs_value\": %.3e, \"nan\": %d}\n",
           n_env, ITER, SWEEPS, extra, n_wg + extra, cm, msm, msm * 1e3 / ITER, 2 * (n_wg + extra), cs, mss, mss * 1e3 / ITER, det[1], det[2], det[3], det[4], det[5], worst, scale, nan);
    return 0;
}
