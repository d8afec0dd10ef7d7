// qs_core.h -- the batched Go1 + PEA simulation step, written once against the lane abstraction of qs_lane.h.
//
// What the reference obtains from 49 PyBullet C-API calls per 1 ms substep (quadruped.py:288-320 + stepSimulation,
// gym_env.py:207-225) is one call of Sim<T>::substep here; one QuadrupedGymEnv.step() (gym_env.py:227-256) is
// Env<T>::step.  Formulation (MI355X-first, not Bullet's):
//   * one lane per leg, everything expressed in BASE-frame coordinates, so composite inertias are plain sums and the
//     only cross-lane traffic is 4-lane sums (DPP): the floating-base mass matrix has the arrow structure
//         H = [ Hbb  B1 B2 B3 B4 ]      Hbb 6x6 (replicated), B_L 6x3 and D_L 3x3 private to leg L
//             [ B_L^T     D_L    ]
//     and is eliminated leg-locally:  S = Hbb - sum_L B_L D_L^-1 B_L^T  (one 21-value quad sum), S = L L^T.
//   * forward dynamics = CRBA + RNEA + that block solve (mathematically the ABA the oracle uses; tests compare them)
//   * contact: foot sphere vs plane, rows (normal, t1, t2) per foot; Delassus columns A[:, own rows] stay in registers;
//     projected Gauss-Seidel in impulse space in Bullet's row order (limits, normals, frictions), `solver_iters` sweeps
//   * semi-implicit Euler, quaternion exponential map.
// Numerical contract: float32; compared with oracle/ (float64 ABA) in tests/ within the tolerances stated there.
#pragma once
#include "qs_lane.h"
#include "qs_layout.h"
#include "../../include/qs_amd.h"

// Optional cycle accounting per phase of the substep (build with -DQS_PROFILE_PHASES; tools/phase_profile.py): s_memtime at
// the phase boundaries of workgroup 0, accumulated in a __device__ array.  Compiled out otherwise.
#if defined(QS_PROFILE_PHASES) && defined(__HIPCC__)
__device__ unsigned long long qs_phase_cycles[48];
__device__ unsigned long long qs_phase_t0, qs_phase_sub0;   // written by one lane of workgroup 0 only
#endif
#if defined(QS_PROFILE_PHASES) && defined(__HIP_DEVICE_COMPILE__)
#define QS_PHASE_BEGIN if (blockIdx.x == 0 && threadIdx.x == 0) qs_phase_t0 = __builtin_readcyclecounter();
#define QS_PHASE_G(k) if (blockIdx.x == 0 && threadIdx.x == 0) { unsigned long long n_ = __builtin_readcyclecounter(); qs_phase_cycles[k] += n_ - qs_phase_t0; qs_phase_t0 = n_; }
#define QS_PHASE(k) QS_PHASE_G(k)
#define QS_PHASE_SUB(k) if (blockIdx.x == 0 && threadIdx.x == 0) { qs_phase_cycles[16 + ((k) < 15 ? (k) : 15)] += __builtin_readcyclecounter() - qs_phase_sub0; }
#define QS_PHASE_SUB_BEGIN if (blockIdx.x == 0 && threadIdx.x == 0) qs_phase_sub0 = __builtin_readcyclecounter();
#define QS_PHASE_END
#elif defined(QS_COUNT_PHASES) && defined(__HIP_DEVICE_COMPILE__)
// static instruction counts per phase: scheduling barriers + assembly comments at the phase boundaries (tools/phase_count.py)
#define QS_PHASE_BEGIN { __builtin_amdgcn_sched_barrier(0); asm volatile("; QS_PHASE_MARK 0"); __builtin_amdgcn_sched_barrier(0); }
#define QS_PHASE_G(k) { __builtin_amdgcn_sched_barrier(0); asm volatile("; QS_PHASE_MARK " #k); __builtin_amdgcn_sched_barrier(0); }
#define QS_PHASE(k) QS_PHASE_G(k)
#define QS_PHASE_SUB(k)
#define QS_PHASE_SUB_BEGIN
#define QS_PHASE_END
#else
#define QS_PHASE_BEGIN
#define QS_PHASE_G(k)
#define QS_PHASE(k)
#define QS_PHASE_SUB(k)
#define QS_PHASE_SUB_BEGIN
#define QS_PHASE_END
#endif

#ifndef QS_PGS_PACKED
#define QS_PGS_PACKED 1   // candidate updates of the sweep as one packed + one scalar FMA per row (0: three scalar FMAs)
#endif

namespace qs {

// ------------------------------------------------------------------ Go1 model constants (go1.urdf)
namespace go1 {
constexpr float HIP_X = 0.1881f, HIP_Y = 0.04675f, THIGH_Y = 0.08f, LEG_Z = -0.213f, FOOT_R = 0.02f;
constexpr float BASE_M = 0.00001f, BASE_I = 1e-5f;                                        // :55-59
constexpr float TRUNK_M = 5.204f, TRUNK_CX = 0.0223f, TRUNK_CZ = -0.0005f;                // :80-85
constexpr float TRUNK_I[6] = {0.0168352186f, 0.0004636141f, 0.0002367952f, 0.0656071082f, 3.6671e-05f, 0.0742720659f};
constexpr float IMU_M = 0.001f, IMU_I = 0.0001f, IMU_X = -0.01592f, IMU_Y = -0.06659f, IMU_Z = -0.00617f;  // :87-97
constexpr float HIP_M = 0.591f, HIP_C[3] = {0.00541f, 0.00074f, 6e-06f};                   // :134-136 and mirrors
constexpr float HIP_I[6] = {0.000374268192f, 3.6844422e-05f, 9.86754e-07f, 0.000635923669f, 1.172894e-06f, 0.000457647394f};
constexpr float THIGH_M = 0.92f, THIGH_C[3] = {-0.003468f, 0.018947f, -0.032736f};         // :186-188
constexpr float THIGH_I[6] = {0.005851561134f, 1.783284e-06f, 0.000328291374f, 0.005596155105f, 2.1430713e-05f, 0.00107157026f};
constexpr float CALF_M = 0.131f, CALF_C[3] = {0.006286f, 0.001307f, -0.122269f};           // :212-216
constexpr float CALF_I[6] = {0.002939186297f, 1.440899e-06f, -0.00010535955f, 0.00295576935f, -2.4397752e-05f, 3.0273372e-05f};
constexpr float FOOT_M = 0.06f, FOOT_I = 9.6e-06f;                                         // :237-240
constexpr float JLO[3] = {-1.0471975512f, -0.663225115758f, -2.72271363311f};              // :117 :169 :196
constexpr float JHI[3] = {1.0471975512f, 2.96705972839f, -0.837758040957f};
// contact breaking thresholds = 0.02 x angular-motion-disc of each link's collision compound (DESIGN.md "contact model")
constexpr float THR_FOOT = 0.000727f, THR_TRUNK = 0.00407f, THR_HIP = 0.00139f, THR_THIGH = 0.00433f, THR_CALF = 0.00429f;
constexpr float TRUNK_HALF[3] = {0.1881f, 0.04675f, 0.057f};                               // box :74-79
constexpr float HIP_CYL_HALF_LEN = 0.02f, HIP_CYL_R = 0.046f;                              // cylinder :128-131
constexpr float LINK_BOX_Z = -0.1065f, THIGH_HALF[3] = {0.017f, 0.01225f, 0.1065f}, CALF_HALF[3] = {0.008f, 0.008f, 0.1065f};
constexpr float PAYLOAD_I = 0.1f * 0.1f / 6.0f;                                            // cube of half extent 0.05, quadruped.py:793
constexpr float PAYLOAD_HALF = 0.05f, THR_PAYLOAD = 0.00173f;                              // its box against the plane (0.02 x |half extents|)
constexpr float HIP_SELF_R = 0.046f;   // link-link tests treat the hip's motor housing (cylinder r 0.046, half length 0.02) as a sphere
}  // namespace go1

// ------------------------------------------------------------------ small linear algebra on lane values
template <class V> struct V3 { V x, y, z; };
template <class V> QS_FN V3<V> mk3(V x, V y, V z) { V3<V> r; r.x = x; r.y = y; r.z = z; return r; }
template <class V> QS_FN V3<V> operator+(V3<V> a, V3<V> b) { return mk3<V>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <class V> QS_FN V3<V> operator-(V3<V> a, V3<V> b) { return mk3<V>(a.x - b.x, a.y - b.y, a.z - b.z); }
// a * s is kept as the pair (a, s) until it is used: `p + a * s`, `p - a * s`, `a * s + b * t` then are ONE expression per component
// and the front end contracts each into an FMA (-ffp-contract=on fuses within an expression only; through two operator functions the
// product and the sum were a v_mul and a v_add -- 88 of them per substep).  Anywhere else the pair converts to the plain product.
template <class V> struct V3s {
    V3<V> a; V s;
    QS_FN operator V3<V>() const { return mk3<V>(a.x * s, a.y * s, a.z * s); }
    QS_FN V3<V> v() const { return mk3<V>(a.x * s, a.y * s, a.z * s); }
};
template <class V> QS_FN V3s<V> operator*(V3<V> a, V s) { V3s<V> r; r.a = a; r.s = s; return r; }
template <class V> QS_FN V3s<V> operator*(V3s<V> a, V t) { V3s<V> r; r.a = a.v(); r.s = t; return r; }
template <class V> QS_FN V3<V> operator+(V3<V> p, V3s<V> q) { return mk3<V>(p.x + q.a.x * q.s, p.y + q.a.y * q.s, p.z + q.a.z * q.s); }
template <class V> QS_FN V3<V> operator+(V3s<V> q, V3<V> p) { return mk3<V>(q.a.x * q.s + p.x, q.a.y * q.s + p.y, q.a.z * q.s + p.z); }
template <class V> QS_FN V3<V> operator-(V3<V> p, V3s<V> q) { return mk3<V>(p.x - q.a.x * q.s, p.y - q.a.y * q.s, p.z - q.a.z * q.s); }
template <class V> QS_FN V3<V> operator-(V3s<V> q, V3<V> p) { return mk3<V>(q.a.x * q.s - p.x, q.a.y * q.s - p.y, q.a.z * q.s - p.z); }
template <class V> QS_FN V3<V> operator+(V3s<V> p, V3s<V> q) { return mk3<V>(p.a.x * p.s + q.a.x * q.s, p.a.y * p.s + q.a.y * q.s, p.a.z * p.s + q.a.z * q.s); }
template <class V> QS_FN V3<V> operator-(V3s<V> p, V3s<V> q) { return mk3<V>(p.a.x * p.s - q.a.x * q.s, p.a.y * p.s - q.a.y * q.s, p.a.z * p.s - q.a.z * q.s); }
template <class V> QS_FN V dot(V3<V> a, V3<V> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class V> QS_FN V3<V> cross(V3<V> a, V3<V> b) { return mk3<V>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
template <class V> struct S3 { V xx, xy, xz, yy, yz, zz; };
template <class V> QS_FN V3<V> mul(S3<V> I, V3<V> w) {
    return mk3<V>(I.xx * w.x + I.xy * w.y + I.xz * w.z, I.xy * w.x + I.yy * w.y + I.yz * w.z, I.xz * w.x + I.yz * w.y + I.zz * w.z);
}
template <class V> struct Sp { V3<V> a, l; };  // spatial vector: (angular; linear) for motion, (moment; force) for force
template <class V> QS_FN Sp<V> operator+(Sp<V> p, Sp<V> q) { Sp<V> r; r.a = p.a + q.a; r.l = p.l + q.l; return r; }
// (Sums of products are written as ONE left-associated chain: without reassociation the compiler turns exactly that into a multiply and
// a chain of FMAs; `cross(..) + cross(..)` or `dot(..) + dot(..)` cost an extra add per component and a longer dependency chain.)
template <class V> QS_FN V dot(Sp<V> p, Sp<V> q) { return p.a.x * q.a.x + p.a.y * q.a.y + p.a.z * q.a.z + p.l.x * q.l.x + p.l.y * q.l.y + p.l.z * q.l.z; }
template <class V> QS_FN Sp<V> crm(Sp<V> v, Sp<V> u) {
    Sp<V> r; r.a = cross(v.a, u.a);
    r.l = mk3<V>(v.a.y * u.l.z - v.a.z * u.l.y + v.l.y * u.a.z - v.l.z * u.a.y, v.a.z * u.l.x - v.a.x * u.l.z + v.l.z * u.a.x - v.l.x * u.a.z,
                 v.a.x * u.l.y - v.a.y * u.l.x + v.l.x * u.a.y - v.l.y * u.a.x);
    return r;
}
template <class V> QS_FN Sp<V> crf(Sp<V> v, Sp<V> f) {
    Sp<V> r; r.l = cross(v.a, f.l);
    r.a = mk3<V>(v.a.y * f.a.z - v.a.z * f.a.y + v.l.y * f.l.z - v.l.z * f.l.y, v.a.z * f.a.x - v.a.x * f.a.z + v.l.z * f.l.x - v.l.x * f.l.z,
                 v.a.x * f.a.y - v.a.y * f.a.x + v.l.x * f.l.y - v.l.y * f.l.x);
    return r;
}
// acc + crm(v, u), acc + crf(v, f): the accumulator starts each component's chain (no separate vector add)
template <class V> QS_FN Sp<V> crm_add(Sp<V> c, Sp<V> v, Sp<V> u) {
    Sp<V> r;
    r.a = mk3<V>(c.a.x + v.a.y * u.a.z - v.a.z * u.a.y, c.a.y + v.a.z * u.a.x - v.a.x * u.a.z, c.a.z + v.a.x * u.a.y - v.a.y * u.a.x);
    r.l = mk3<V>(c.l.x + v.a.y * u.l.z - v.a.z * u.l.y + v.l.y * u.a.z - v.l.z * u.a.y, c.l.y + v.a.z * u.l.x - v.a.x * u.l.z + v.l.z * u.a.x - v.l.x * u.a.z,
                 c.l.z + v.a.x * u.l.y - v.a.y * u.l.x + v.l.x * u.a.y - v.l.y * u.a.x);
    return r;
}
template <class V> QS_FN Sp<V> crf_add(Sp<V> c, Sp<V> v, Sp<V> f) {
    Sp<V> r;
    r.l = mk3<V>(c.l.x + v.a.y * f.l.z - v.a.z * f.l.y, c.l.y + v.a.z * f.l.x - v.a.x * f.l.z, c.l.z + v.a.x * f.l.y - v.a.y * f.l.x);
    r.a = mk3<V>(c.a.x + v.a.y * f.a.z - v.a.z * f.a.y + v.l.y * f.l.z - v.l.z * f.l.y, c.a.y + v.a.z * f.a.x - v.a.x * f.a.z + v.l.z * f.l.x - v.l.x * f.l.z,
                 c.a.z + v.a.x * f.a.y - v.a.y * f.a.x + v.l.x * f.l.y - v.l.y * f.l.x);
    return r;
}
// spatial inertia about the base origin, base coordinates
template <class V> struct SI { V m; V3<V> h; S3<V> I; };
template <class V> QS_FN SI<V> operator+(SI<V> p, SI<V> q) {
    SI<V> r; r.m = p.m + q.m; r.h = p.h + q.h;
    r.I.xx = p.I.xx + q.I.xx; r.I.xy = p.I.xy + q.I.xy; r.I.xz = p.I.xz + q.I.xz; r.I.yy = p.I.yy + q.I.yy; r.I.yz = p.I.yz + q.I.yz; r.I.zz = p.I.zz + q.I.zz;
    return r;
}
template <class V> QS_FN Sp<V> apply(SI<V> I, Sp<V> v) {   // (I w + h x v ; m v - h x w)
    Sp<V> f;
    f.a = mk3<V>(I.I.xx * v.a.x + I.I.xy * v.a.y + I.I.xz * v.a.z + I.h.y * v.l.z - I.h.z * v.l.y,
                 I.I.xy * v.a.x + I.I.yy * v.a.y + I.I.yz * v.a.z + I.h.z * v.l.x - I.h.x * v.l.z,
                 I.I.xz * v.a.x + I.I.yz * v.a.y + I.I.zz * v.a.z + I.h.x * v.l.y - I.h.y * v.l.x);
    f.l = mk3<V>(v.l.x * I.m - I.h.y * v.a.z + I.h.z * v.a.y, v.l.y * I.m - I.h.z * v.a.x + I.h.x * v.a.z, v.l.z * I.m - I.h.x * v.a.y + I.h.y * v.a.x);
    return f;
}
// rigid part with mass m, COM cl and inertia Il (6 unique, about the COM) given in a link frame whose origin is p and whose
// axes are X, Y, Z (all in base coordinates)
template <class V> QS_FN SI<V> part_inertia(V m, V3<V> cl, S3<V> Il, V3<V> p, V3<V> X, V3<V> Y, V3<V> Z) {
    V3<V> c = p + X * cl.x + Y * cl.y + Z * cl.z;
    V3<V> Tx = X * Il.xx + Y * Il.xy + Z * Il.xz;
    V3<V> Ty = X * Il.xy + Y * Il.yy + Z * Il.yz;
    V3<V> Tz = X * Il.xz + Y * Il.yz + Z * Il.zz;
    SI<V> r; r.m = m; r.h = c * m;
    V cc = dot(c, c);
    r.I.xx = Tx.x * X.x + Ty.x * Y.x + Tz.x * Z.x + m * (cc - c.x * c.x);
    r.I.xy = Tx.x * X.y + Ty.x * Y.y + Tz.x * Z.y - m * c.x * c.y;
    r.I.xz = Tx.x * X.z + Ty.x * Y.z + Tz.x * Z.z - m * c.x * c.z;
    r.I.yy = Tx.y * X.y + Ty.y * Y.y + Tz.y * Z.y + m * (cc - c.y * c.y);
    r.I.yz = Tx.y * X.z + Ty.y * Y.z + Tz.y * Z.z - m * c.y * c.z;
    r.I.zz = Tx.z * X.z + Ty.z * Y.z + Tz.z * Z.z + m * (cc - c.z * c.z);
    return r;
}
template <class V> QS_FN SI<V> point_inertia(V m, V iso, V3<V> c) {  // isotropic inertia `iso` about its COM c
    SI<V> r; r.m = m; r.h = c * m;
    V cc = dot(c, c);
    r.I.xx = iso + m * (cc - c.x * c.x); r.I.xy = -(m * c.x * c.y); r.I.xz = -(m * c.x * c.z);
    r.I.yy = iso + m * (cc - c.y * c.y); r.I.yz = -(m * c.y * c.z); r.I.zz = iso + m * (cc - c.z * c.z);
    return r;
}
// symmetric 6x6 in packed lower-triangular storage, idx(i,j) = i(i+1)/2 + j for j <= i
QS_FN constexpr int tri(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }
template <class V> QS_FN void chol6(V* s, V* dinv) {  // in place: s <- L with S = L L^T ; dinv[j] = 1 / L_jj
#pragma unroll
    for (int j = 0; j < 6; j++) {
        V d = s[tri(j, j)];
#pragma unroll
        for (int k = 0; k < j; k++) d = d - s[tri(j, k)] * s[tri(j, k)];
        V inv = qrsqrt(d);
        s[tri(j, j)] = d * inv;
        dinv[j] = inv;
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            V t = s[tri(i, j)];
#pragma unroll
            for (int k = 0; k < j; k++) t = t - s[tri(i, k)] * s[tri(j, k)];
            s[tri(i, j)] = t * inv;
        }
    }
}
template <class V> QS_FN void lsolve6(const V* L, const V* dinv, V* b) {  // b <- L^-1 b   (dinv[i] = 1/L_ii)
#pragma unroll
    for (int i = 0; i < 6; i++) {
        V t = b[i];
#pragma unroll
        for (int k = 0; k < i; k++) t = t - L[tri(i, k)] * b[k];
        b[i] = t * dinv[i];
    }
}
template <class V> QS_FN void ltsolve6(const V* L, const V* dinv, V* b) {  // b <- L^-T b
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        V t = b[i];
#pragma unroll
        for (int k = i + 1; k < 6; k++) t = t - L[tri(k, i)] * b[k];
        b[i] = t * dinv[i];
    }
}
template <class V> QS_FN V clampv(V x, V lo, V hi) { return qmed3(x, lo, hi); }   // one v_med3_f32 on the device (no NaN canonicalisation of the three operands)

// ------------------------------------------------------------------ Philox4x32-10 (same stream definition as oracle/qso_env.c)
QS_FN void philox4x32(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
QS_FN float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }
QS_FN void normal4_scalar(uint64_t seed, uint32_t env, uint32_t stream, uint32_t ctr, uint32_t blk, float z[4]) {
    uint32_t r[4]; philox4x32(seed, env, stream, ctr, blk, r);
#pragma unroll
    for (int h = 0; h < 2; h++) {
        float u1 = u01(r[2 * h]), u2 = u01(r[2 * h + 1]);
#if defined(__HIP_DEVICE_COMPILE__)
        // Box-Muller on the hardware transcendentals: v_log_f32 is log2, v_sin_f32 / v_cos_f32 take their argument in revolutions,
        // which is what u2 is (libm's sinf / cosf / logf cost ~3 us per step at N = 8192 for noise of sigma ~ 1e-2)
        float rad = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
        z[2 * h] = rad * __builtin_amdgcn_cosf(u2); z[2 * h + 1] = rad * __builtin_amdgcn_sinf(u2);
#else
        float rad = sqrtf(-2.0f * logf(u1)), th = 6.283185307179586f * u2;
        z[2 * h] = rad * cosf(th); z[2 * h + 1] = rad * sinf(th);
#endif
    }
}
template <class T> struct Rng;
#if defined(__HIPCC__)
template <> struct Rng<LaneDev> {  // lane `leg` draws block blk0 + leg
    static QS_DEV void normal4(uint64_t seed, uint32_t env, uint32_t stream, uint32_t ctr, uint32_t blk0, float z[4]) {
        normal4_scalar(seed, env, stream, ctr, blk0 + (uint32_t)(threadIdx.x & 3u), z);
    }
};
#endif
#if !defined(__HIP_DEVICE_COMPILE__)
template <> struct Rng<LaneEmu> {
    static void normal4(uint64_t seed, uint32_t env, uint32_t stream, uint32_t ctr, uint32_t blk0, V4 z[4]) {
        for (int l = 0; l < 4; l++) { float t[4]; normal4_scalar(seed, env, stream, ctr, blk0 + l, t); for (int k = 0; k < 4; k++) z[k].v[l] = t[k]; }
    }
};
#endif

// ------------------------------------------------------------------ rigid-body substep
// CONE: friction model of the contact rows, a compile-time choice so that neither variant costs the other registers or a branch:
// false = pyramid with Bullet's skip rule, true = implicit cone (qs_config::friction_cone; the kernels are built for both).
// HOT: the build of the substep that holds the common path ONLY.  Where a wave would need one of the rare paths (a joint at its stop,
// a non-foot link on the plane, the link-link tests of the self-collision rule) substep() gives up, leaves the state as it found it and
// returns true: the env step of that wave then goes on FROM THAT SUBSTEP in the full build (HOT = false; Env::step), whose rare code thus
// sits behind the hot loop and costs it neither registers nor schedule (measured: inlined into the loop the rare code took 20 % off the
// headline).  (Rounds 2-3 repeated the whole env step with the full build: a rare-path wave paid up to two steps' time.)
// SOFT (HOT builds): the common-path build ALSO holds the payload block's six rows (cfg.payload_soft), next to the twelve foot rows in
// solve_and_integrate<.., PAY>; without it a common-path build gives up on every substep of such a handle and the full build does the work.
// the values a substep works on: one set of types for every build of Sim (an env step may start in the common-path build and go on in
// the full one, qs_env.h)
template <class T> struct SimTypes {
    using V = typename T::V;
    using M = typename T::M;
    using V3v = V3<V>;
    using Spv = Sp<V>;
    struct State {          // registers carried through the substeps of one env step
        V3v pos; V qx, qy, qz, qw; V3v vlin, vang;  // replicated over the quad
        V q[3], qd[3];                              // own leg
        V warm;                                     // own foot
    };
    struct Par {            // per-environment parameters (replicated over the quad)
        V mu, k[3], b[3], rest[3], kp[3], kd[3];
        V m_leg[3];
        SI<V> I0;                                   // base + trunk + imu (+ payload) about the base origin
        V mtot;
        V m_pay; V3v r_pay;                         // payload block (quadruped.py:778-819): its box can touch the ground
    };
    struct Model {          // link constants of the own leg (mirror signs, mass scaling), rebuilt inside every substep
        V m_hip, m_thigh, m_calf;
        V3v c_hip, c_thigh; S3<V> I_hip, I_thigh, I_calf;
    };
    struct Out {            // results of the last substep
        V tau_pd[3], tau_spring[3], foot_force, foot_contact, n_invalid;
    };
    struct Row { V jq[3], u[3], w[6], rhs, dinv, act, diag; };
    // the six rows of the payload block's fixed constraint (cfg.payload_soft, see integrate_rare)
    struct PayRows {
        V w[6][6], rhs[6], dinv[6], diag[6];   // base side (whitened), right-hand side x dinv, 1 / A_kk, A_kk
        V3v rB; V mI, mM, act;                  // block centre -> pivot (world), 1 / inertia, 1 / mass, 1 while the block exists
        V lam[6]; V3v dw, dv;                   // results: impulses, the block's velocity change (world)
    };
};

#include "qs_rare.h"

template <class T, bool CONE = false, bool HOT = false, bool SOFT = false> struct Sim : SimTypes<T> {
    using V = typename T::V;
    using M = typename T::M;
    using V3v = V3<V>;
    using Spv = Sp<V>;
    using State = typename SimTypes<T>::State;
    using Par = typename SimTypes<T>::Par;
    using Model = typename SimTypes<T>::Model;
    using Out = typename SimTypes<T>::Out;
    using Row = typename SimTypes<T>::Row;
    using PayRows = typename SimTypes<T>::PayRows;

    // per-env model from the randomizable masses (env_randomizer.py:56-83); a link of mass m has the inertia m x cfg.unit_inertia
    // (the host picks the rule: URDF tensor scaled with the mass, or Bullet's collision-shape inertia; qs_amd/config.py)
    static QS_FN void build_base(const qs_config& cfg, Par& P, V m_trunk, V m_pay, V3v r_pay) {
        using namespace go1;
        const float* U = cfg.unit_inertia[3];
        SI<V> I0 = point_inertia<V>(V(BASE_M), V(BASE_I), mk3<V>(V(0.0f), V(0.0f), V(0.0f)));
        S3<V> It; It.xx = m_trunk * U[0]; It.xy = m_trunk * U[1]; It.xz = m_trunk * U[2]; It.yy = m_trunk * U[3]; It.yz = m_trunk * U[4]; It.zz = m_trunk * U[5];
        V3v ex = mk3<V>(V(1.0f), V(0.0f), V(0.0f)), ey = mk3<V>(V(0.0f), V(1.0f), V(0.0f)), ez = mk3<V>(V(0.0f), V(0.0f), V(1.0f));
        V3v zero = mk3<V>(V(0.0f), V(0.0f), V(0.0f));
        I0 = I0 + part_inertia<V>(m_trunk, mk3<V>(V(TRUNK_CX), V(0.0f), V(TRUNK_CZ)), It, zero, ex, ey, ez);
        I0 = I0 + point_inertia<V>(V(IMU_M), V(IMU_I), mk3<V>(V(IMU_X), V(IMU_Y), V(IMU_Z)));
        if (!cfg.payload_soft) I0 = I0 + point_inertia<V>(m_pay, m_pay * PAYLOAD_I, r_pay);   // welded to the trunk; "soft": a body of its own
        P.I0 = I0;
        P.mtot = I0.m + 4.0f * (P.m_leg[0] + P.m_leg[1] + P.m_leg[2] + FOOT_M);
        P.m_pay = m_pay; P.r_pay = r_pay;
    }
    static QS_FN void build_model(const qs_config& cfg, Model& P, const V* m_leg) {
        using namespace go1;
        V fx = T::fx(), sy = T::sy();
        const float* H = cfg.unit_inertia[0]; const float* Th = cfg.unit_inertia[1]; const float* Cf = cfg.unit_inertia[2];
        V s1 = m_leg[0], s2 = m_leg[1], s3 = m_leg[2];
        P.m_hip = m_leg[0]; P.m_thigh = m_leg[1]; P.m_calf = m_leg[2];
        P.c_hip = mk3<V>(fx * (-HIP_C[0]), sy * (-HIP_C[1]), V(HIP_C[2]));
        P.I_hip.xx = s1 * H[0]; P.I_hip.xy = s1 * (fx * sy) * H[1]; P.I_hip.xz = s1 * fx * (-H[2]);
        P.I_hip.yy = s1 * H[3]; P.I_hip.yz = s1 * sy * (-H[4]); P.I_hip.zz = s1 * H[5];
        P.c_thigh = mk3<V>(V(THIGH_C[0]), sy * (-THIGH_C[1]), V(THIGH_C[2]));
        P.I_thigh.xx = s2 * Th[0]; P.I_thigh.xy = s2 * sy * Th[1]; P.I_thigh.xz = s2 * Th[2];
        P.I_thigh.yy = s2 * Th[3]; P.I_thigh.yz = s2 * sy * Th[4]; P.I_thigh.zz = s2 * Th[5];
        P.I_calf.xx = s3 * Cf[0]; P.I_calf.xy = s3 * Cf[1]; P.I_calf.xz = s3 * Cf[2];
        P.I_calf.yy = s3 * Cf[3]; P.I_calf.yz = s3 * Cf[4]; P.I_calf.zz = s3 * Cf[5];
    }

    // PD law + torque clip (quadruped_motor.py:45-99) and unilateral PEA (quadruped_motor.py:101-104, springs.py:34-74)
    // `settling`: a reset's settle always runs the PD law (control_interface/utils.py:7-31 switches the motor model to "PD" for it,
    // also when the environment itself is driven by raw torques)
    static QS_FN void actuate(const qs_config& cfg, const Par& P, const State& s, const V* cmd, Out& o, V* tau, bool settling = false) {
        V sy = T::sy();
#pragma unroll
        for (int j = 0; j < 3; j++) {
            V lim = V(cfg.tau_max[j]);
            V t = (cfg.motor_control_mode == QS_MOTOR_TORQUE && !settling) ? cmd[j] : (-(P.kp[j] * (s.q[j] - cmd[j])) - P.kd[j] * s.qd[j]);
            o.tau_pd[j] = clampv<V>(t, -lim, lim);
            V ts = V(0.0f);
            if (cfg.enable_springs) {
                V dq = s.q[j] - P.rest[j];
                M off = j == 0 ? qlt(sy * dq, V(0.0f)) : (j == 1 ? qlt(dq, V(0.0f)) : qgt(dq, V(0.0f)));
                ts = qsel(off, V(0.0f), -P.k[j] * dq - P.b[j] * s.qd[j]);
            }
            o.tau_spring[j] = ts;
            tau[j] = o.tau_pd[j] + ts;
        }
    }

    // One stepSimulation() (gym_env.py:218-219) under joint torques tau[3] per leg.
    // TRACK: follow PyBullet's solverResidualThreshold -- an environment whose sweep changed no row velocity by more than
    // sqrt(threshold) is frozen (its residuals are zeroed, so later sweeps leave it untouched) and the wave leaves the
    // loop when all of its 16 environments are frozen.  Without TRACK every sweep is executed.
    // PAY (cfg.payload_soft with every robot of interest on its feet): the six rows of the payload block's fixed constraint ride along with
    // the twelve foot rows, in impulse space like them (replicated over the quad: the rows' relative velocities prel[6], moved by a payload
    // delta through the 6 x 6 block App and by the foot rows' deltas of a sweep, together, at its end), swept where Bullet sorts them (in
    // front of the normals; forwards on odd sweeps, backwards on even ones); a payload delta moves the lane's own foot candidates through
    // the precomputed couplings Apc[k][c] = -dinv_c (w_c . pw_k).  Same rows, clamps and order as the many-rows solver (which keeps them
    // in velocity space), at a quarter of its instructions per sweep.
    template <int NR, bool TRACK, bool PAY = false> static QS_FN void solve_and_integrate(const qs_config& cfg, V mu, State& s, Out& o, const Row* rows,
                                                            const V* Lc, const V* Ld, const V (*BK)[6], const V* R, PayRows* pq = nullptr) {
        constexpr int NT = 4 * NR;
        const float dt = (float)cfg.dt;
        // Delassus columns of the own rows, pre-scaled by the own row's 1/diag:
        //   Ap[(k,r)][c] = -(w_kr . w_c + [k == own] jq_r . u_c) * dinv_c      (negated: the update is a v_fmac_f32_dpp)
        // Projected Gauss-Seidel then runs on the lane-private UNCLAMPED candidates
        //   cand_c = lam_c + rhs_c - dinv_c * sum_j A_cj lam_j        (the Gauss-Seidel update of row c is lam_c <- clamp(cand_c))
        // A row's own update leaves its candidate unchanged (A_cc * dinv_c = 1), so per row the loop needs: clamp, delta =
        // clamp(cand) - lam, one DPP broadcast of delta over the quad, and cand_c -= Ap[i][c] * delta for the other rows
        // (the self entry of Ap is zeroed).  Rows of inactive contacts / limits have w = jq = u = rhs = 0: cand = lam = 0.
        V Ap[NT][NR];
        V lam_own[NR], res[NR];
        V loc[NR][NR];
        V diag_all[TRACK ? NT : 1];
#pragma unroll
        for (int r = 0; r < NR; r++)
#pragma unroll
            for (int c = 0; c < NR; c++) loc[r][c] = rows[r].jq[0] * rows[c].u[0] + rows[r].jq[1] * rows[c].u[1] + rows[r].jq[2] * rows[c].u[2];
        // The 12 x 12 Delassus matrix restricted to the own columns is, per pair (r, c) of row kinds, a 4 x 4 matrix over
        // (leg K, own leg): sum_t w[K,r][t] * w[own,c][t] -- six rank-1 updates that ONE 4x4x1 MFMA each performs for all
        // sixteen environments of the wave.  The columns are pre-scaled by -1/diag_c, the leg-local term jq_r . u_c is added on
        // the diagonal block, and the self entries are zeroed.
        {
            V wc[NR][6], locs[NR][NR];
#pragma unroll
            for (int c = 0; c < NR; c++) {
                V nd = -rows[c].dinv;
#pragma unroll
                for (int i = 0; i < 6; i++) wc[c][i] = rows[c].w[i] * nd;
#pragma unroll
                for (int r = 0; r < NR; r++) locs[r][c] = loc[r][c] * nd;
            }
            typename T::Acc4 acc[NR][NR];
#pragma unroll
            for (int r = 0; r < NR; r++)
#pragma unroll
                for (int c = 0; c < NR; c++) acc[r][c] = T::acc4_zero();
            // (MFMA in the full build as well since round 5: round 4 ran these products on the vector ALU there -- bitwise the same -- because hipcc
            // crashed on spilled MFMA accumulators; it does not on this source, and build.py still retries without -amdgpu-mfma-vgpr-form if it
            // does: +1.5 % with the links' response on)
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int r = 0; r < NR; r++)
#pragma unroll
                    for (int c = 0; c < NR; c++) { T::outer_fma(rows[r].w[i], wc[c][i], acc[r][c]); }
#define QS_SCATTER(K)                                                                                                  \
    {                                                                                                                  \
        M own = T::is_leg(K);                                                                                          \
        V ownf = qflag(own);                                                                                           \
        _Pragma("unroll") for (int r = 0; r < NR; r++) {                                                               \
            _Pragma("unroll") for (int c = 0; c < NR; c++) {                                                           \
                V a_ = T::template acc4_get<K>(acc[r][c]) + ownf * locs[r][c];                                         \
                Ap[NR * K + r][c] = (r == c) ? qsel(own, V(0.0f), a_) : a_;                                            \
            }                                                                                                          \
            if (TRACK) diag_all[TRACK ? NR * K + r : 0] = T::template bcast<K>(rows[r].diag);                          \
        }                                                                                                              \
    }
            QS_SCATTER(0) QS_SCATTER(1) QS_SCATTER(2) QS_SCATTER(3)
#undef QS_SCATTER
        }
        // warm start: normal rows only, factor cfg.warmstart (btMultiBodyConstraintSolver, SOLVER_USE_WARMSTARTING)
#pragma unroll
        for (int r = 0; r < NR; r++) res[r] = rows[r].rhs;   // res[] holds the candidates: rhs_c - dinv_c sum_{j != c} A_cj lam_j
        V lam_all[NT];                                         // every impulse of the environment, replicated over the quad
#pragma unroll
        for (int i = 0; i < NT; i++) lam_all[i] = V(0.0f);
        {
            V l_own = s.warm * cfg.warmstart * rows[0].act;
            lam_all[NR * 0] = T::template bcast<0>(l_own); lam_all[NR * 1] = T::template bcast<1>(l_own);
            lam_all[NR * 2] = T::template bcast<2>(l_own); lam_all[NR * 3] = T::template bcast<3>(l_own);
#pragma unroll
            for (int c = 0; c < NR; c++)
                res[c] = res[c] + Ap[NR * 0][c] * lam_all[NR * 0] + Ap[NR * 1][c] * lam_all[NR * 1] + Ap[NR * 2][c] * lam_all[NR * 2] + Ap[NR * 3][c] * lam_all[NR * 3];
        }
        QS_PHASE_G(9)
        const V big = V(1e10f), zero = V(0.0f);
        // payload rows (PAY): state of the velocity-space part
        // prel[P]: the velocity along payload row P under the impulses so far (what the row's clamp looks at); a delta of row Q moves it by
        // App[Q][P] = pw_Q . pw_P + the block's own response (1 / mass on the pivot rows, 1 / inertia on the rotation rows and through the
        // lever rB on the pivot rows), a delta of foot row c of this lane by Cpc[P][c] = w_c . pw_P; Apc = -dinv_c Cpc moves the foot
        // candidates.  J^T lambda's base part and the block's velocity change are put together from the impulses after the sweeps.
        V prel[PAY ? 6 : 1], plam[PAY ? 6 : 1], Apc[PAY ? 6 : 1][NR], Cpc[PAY ? 6 : 1][NR], App[PAY ? 6 : 1][PAY ? 6 : 1];
        V3v pja[PAY ? 3 : 1];
        V plive = V(1.0f);
        const V pbound = V(500.0f * (float)cfg.dt);
        if (PAY) {
            const PayRows& q = *pq;
            V py0[6];
#pragma unroll
            for (int i = 0; i < 6; i++) { py0[i] = T::quad_sum(rows[0].w[i] * (s.warm * cfg.warmstart * rows[0].act)); plam[PAY ? i : 0] = zero; }
#pragma unroll
            for (int k = 0; k < 6; k++) {
#pragma unroll
                for (int c = 0; c < NR; c++) {
                    V t = rows[c].w[0] * q.w[k][0];
#pragma unroll
                    for (int i = 1; i < 6; i++) t = t + rows[c].w[i] * q.w[k][i];
                    Cpc[PAY ? k : 0][c] = t;
                    Apc[PAY ? k : 0][c] = -(t * rows[c].dinv);
                }
                V r0 = q.w[k][0] * py0[0];
#pragma unroll
                for (int i = 1; i < 6; i++) r0 = r0 + q.w[k][i] * py0[i];
                prel[PAY ? k : 0] = r0;                                   // the warm-started normal impulses' share
            }
            pja[0] = mk3<V>(zero, -q.rB.z, q.rB.y); pja[PAY ? 1 : 0] = mk3<V>(q.rB.z, zero, -q.rB.x); pja[PAY ? 2 : 0] = mk3<V>(-q.rB.y, q.rB.x, zero);   // -(rB x e_k)
#pragma unroll
            for (int P = 0; P < 6; P++)
#pragma unroll
                for (int Q = P; Q < 6; Q++) {
                    V t = q.w[P][0] * q.w[Q][0];
#pragma unroll
                    for (int i = 1; i < 6; i++) t = t + q.w[P][i] * q.w[Q][i];
                    if (Q < 3) { t = t + q.mI * dot(pja[PAY ? P : 0], pja[PAY ? Q : 0]); if (P == Q) t = t + q.mM; }             // pivot x pivot
                    else if (P < 3) { const V3v jp = pja[PAY ? P : 0]; t = t - q.mI * (Q == 3 ? jp.x : Q == 4 ? jp.y : jp.z); }   // pivot x rotation
                    else if (P == Q) t = t + q.mI;                                                                              // rotation x rotation
                    App[PAY ? P : 0][PAY ? Q : 0] = t; App[PAY ? Q : 0][PAY ? P : 0] = t;
                }
        }
        const V thr = V(sqrtf(cfg.solver_residual_threshold));
        // (the implicit cone's projection is not idempotent in floating point: once an environment is frozen its friction bound is lifted,
        // so that the sweeps the rest of the wave still needs leave it exactly alone and no result depends on the wave's other environments)
        V mu_c = mu;
#if defined(QS_PROBE_SWEEPS) && defined(__HIP_DEVICE_COMPILE__)
        float probe_wave = 0.0f, probe_own = 0.0f, probe_frozen = 0.0f;   // counting build (tools/probe_sweeps.py): sweeps the wave ran / this environment needed
#endif
        for (int it = 0; it < cfg.solver_iters; it++) {
#if defined(QS_PROBE_SWEEPS) && defined(__HIP_DEVICE_COMPILE__)
            probe_wave += 1.0f; probe_own += 1.0f - probe_frozen;
#endif
            V dvmax = zero;   // largest |row velocity change| of this sweep (replicated over the quad)
            // (PAY) the impulse changes of the lane's OWN foot rows in this sweep.  The payload rows read J^T lambda's base part (py) at the
            // start of a sweep only, the foot rows do not read it at all (their candidates move with Ap / Apc directly): so the foot rows'
            // share is added once per sweep, each lane summing its own three rows and one quad_sum per component -- 48 instructions where
            // 144 broadcast-multiply-adds (every row's delta times lane K's row vector, fetched by DPP) used to keep py current row by row.
            V own_d[PAY ? NR : 1];
            if (PAY) { _Pragma("unroll") for (int r = 0; r < NR; r++) own_d[PAY ? r : 0] = zero; }
            // Row (K, RR): every lane clamps the candidate of ITS row RR, lane K's result is the real one; one DPP-fused
            // subtract fetches it (delta = bcast_K(cand) - lam), the replicated impulse is advanced and the candidates of the
            // lane's own rows move by Ap * delta.
#define QS_ROW_UPDATE(K, RR, KIND)                                                                                    \
    {                                                                                                                  \
        constexpr int i_ = NR * (K) + (RR);                                                                            \
        V cand;                                                                                                        \
        if (KIND == 0) { /* unilateral row: [0, 1e10] */                                                               \
            cand = qmed3(res[RR], zero, big);                                                                          \
        } else { /* friction row bounded by mu * current normal impulse; skipped while that impulse is not positive */ \
            V tot = lam_all[NR * (K)];                                                                                 \
            V lim = mu * tot;                                                                                          \
            cand = qsel(qgt(tot, zero), qmed3(res[RR], -lim, lim), lam_all[i_]);                                       \
        }                                                                                                              \
        V dk = T::template bcast<K>(cand) - lam_all[i_];                                                               \
        lam_all[i_] = lam_all[i_] + dk;                                                                                \
        if (NR == 3 && QS_PGS_PACKED) { T::fma2(Ap[i_][0], Ap[i_][1], dk, res[0], res[1]); res[2] = res[2] + Ap[i_][2] * dk; }      \
        else { _Pragma("unroll") for (int c = 0; c < NR; c++) res[c] = res[c] + Ap[i_][c] * dk; }                     \
        if (PAY) own_d[PAY ? (RR) : 0] = qsel(T::is_leg(K), dk, own_d[PAY ? (RR) : 0]);                                 \
        if (TRACK) dvmax = qmax(dvmax, qabs(dk * diag_all[TRACK ? i_ : 0]));                                           \
    }
            // Implicit cone friction (CONE; resolveConeFrictionConstraintRows): the two friction rows of foot K from the same
            // state, their summed impulse scaled back onto the disc of radius mu x normal impulse, both deltas applied together.
#define QS_PAIR_UPDATE(K)                                                                                              \
    {                                                                                                                  \
        constexpr int ia_ = NR * (K) + 1, ib_ = NR * (K) + 2;                                                          \
        V lim = mu_c * lam_all[NR * (K)];                                                                              \
        V r2 = res[1] * res[1] + res[2] * res[2];                                                                      \
        V sc = qmin(lim * qrsqrt(qmax(r2, V(1e-30f))), V(1.0f));   /* min(1, mu lambda_n / |c|) */                      \
        V da = T::template bcast<K>(res[1] * sc) - lam_all[ia_], db = T::template bcast<K>(res[2] * sc) - lam_all[ib_]; \
        lam_all[ia_] = lam_all[ia_] + da; lam_all[ib_] = lam_all[ib_] + db;                                             \
        /* two chained FMAs per candidate (not res + (a da + b db): one instruction and one level of the dependency chain less) */ \
        _Pragma("unroll") for (int c = 0; c < NR; c++) { res[c] = res[c] + Ap[ia_][c] * da; res[c] = res[c] + Ap[ib_][c] * db; } \
        if (PAY) { own_d[PAY ? 1 : 0] = qsel(T::is_leg(K), da, own_d[PAY ? 1 : 0]); own_d[PAY ? 2 : 0] = qsel(T::is_leg(K), db, own_d[PAY ? 2 : 0]); } \
        if (TRACK) dvmax = T::absmax_mul2(dvmax, da, diag_all[TRACK ? ia_ : 0], db, diag_all[TRACK ? ib_ : 0]);           \
    }
            // payload row P (0..2: the pivot along world axis P, 3..5: the frames' relative rotation), replicated over the quad
#define QS_PAYROW(P)                                                                                                   \
    {                                                                                                                  \
        const PayRows& q_ = *pq;                                                                                        \
        V cand = clampv<V>(plam[PAY ? (P) : 0] + (q_.rhs[P] - q_.dinv[P] * prel[PAY ? (P) : 0]), -pbound, pbound);     \
        V dl = (cand - plam[PAY ? (P) : 0]) * pla;                                                                     \
        plam[PAY ? (P) : 0] = plam[PAY ? (P) : 0] + dl;                                                                \
        _Pragma("unroll") for (int k = 0; k < 6; k++) prel[PAY ? k : 0] = prel[PAY ? k : 0] + App[PAY ? (P) : 0][PAY ? k : 0] * dl; \
        _Pragma("unroll") for (int c = 0; c < NR; c++) res[c] = res[c] + Apc[PAY ? (P) : 0][c] * dl;                   \
        if (TRACK) dvmax = qmax(dvmax, qabs(dl * q_.diag[P]));                                                         \
    }
            const V pla = PAY ? plive * pq->act : zero;   // (a frozen environment's payload deltas are dropped)
            if (PAY) {
                if (it & 1) { QS_PAYROW(0) QS_PAYROW(1) QS_PAYROW(2) QS_PAYROW(3) QS_PAYROW(4) QS_PAYROW(5) }
                else { QS_PAYROW(5) QS_PAYROW(4) QS_PAYROW(3) QS_PAYROW(2) QS_PAYROW(1) QS_PAYROW(0) }
            }
#undef QS_PAYROW
            QS_ROW_UPDATE(0, 0, 0) QS_ROW_UPDATE(1, 0, 0) QS_ROW_UPDATE(2, 0, 0) QS_ROW_UPDATE(3, 0, 0)
            if (NR == 3 && CONE) {
                QS_PAIR_UPDATE(0) QS_PAIR_UPDATE(1) QS_PAIR_UPDATE(2) QS_PAIR_UPDATE(3)
            } else {
                QS_ROW_UPDATE(0, 1, 1) QS_ROW_UPDATE(0, 2, 1) QS_ROW_UPDATE(1, 1, 1) QS_ROW_UPDATE(1, 2, 1)
                QS_ROW_UPDATE(2, 1, 1) QS_ROW_UPDATE(2, 2, 1) QS_ROW_UPDATE(3, 1, 1) QS_ROW_UPDATE(3, 2, 1)
            }
#undef QS_PAIR_UPDATE
#undef QS_ROW_UPDATE
            if (PAY) {
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    V t = Cpc[PAY ? k : 0][0] * own_d[0];
#pragma unroll
                    for (int r = 1; r < NR; r++) t = t + Cpc[PAY ? k : 0][r] * own_d[PAY ? r : 0];
                    prel[PAY ? k : 0] = prel[PAY ? k : 0] + T::quad_sum(t);
                }
            }
            if (TRACK) {
                M conv = qle(dvmax, thr);
                if (PAY) plive = qsel(conv, zero, plive);     // a frozen environment's payload deltas are dropped
                if (CONE) mu_c = qsel(conv, V(1e30f), mu_c);
                // frozen environment: make clamp(cand) == lam for its rows from now on
#pragma unroll
                for (int c = 0; c < NR; c++) {
                    V mine = qsel(T::is_leg(0), lam_all[NR * 0 + c], qsel(T::is_leg(1), lam_all[NR * 1 + c], qsel(T::is_leg(2), lam_all[NR * 2 + c], lam_all[NR * 3 + c])));
                    res[c] = qsel(conv, mine, res[c]);
                }
#if defined(QS_PROBE_SWEEPS) && defined(__HIP_DEVICE_COMPILE__)
                probe_frozen = conv ? 1.0f : probe_frozen;
#endif
                if (!T::any(qnot(conv))) break;
            }
        }
#if defined(QS_PROBE_SWEEPS) && defined(__HIP_DEVICE_COMPILE__)
        {   // probe[0] solves (wave level), [1] sweeps the waves ran, [2] environment-solves, [3] sweeps the environments needed (until frozen),
            // [4] solves in which the wave ran every sweep, [5] environment-solves that never froze
            unsigned long long* pc = reinterpret_cast<const QsDevCfg&>(cfg).counters + 2;
            const float own_sum = T::quad_sum(probe_own) * 0.25f;     // (replicated over the quad)
            float tot = own_sum;                                       // sum over the wave's 16 environments: one lane per quad
            const unsigned long long never = __ballot(probe_frozen < 0.5f);
            for (int off = 4; off < 64; off <<= 1) tot += __shfl_xor(tot, off);
            if (threadIdx.x == 0) {
                atomicAdd(&pc[0], 1ull); atomicAdd(&pc[1], (unsigned long long)probe_wave);
                atomicAdd(&pc[2], 16ull); atomicAdd(&pc[3], (unsigned long long)(tot + 0.5f));
                if (probe_wave >= (float)cfg.solver_iters) atomicAdd(&pc[4], 1ull);
                atomicAdd(&pc[5], (unsigned long long)(__popcll(never) / 4));
            }
        }
#endif
#pragma unroll
        for (int c = 0; c < NR; c++)
            lam_own[c] = qsel(T::is_leg(0), lam_all[NR * 0 + c], qsel(T::is_leg(1), lam_all[NR * 1 + c], qsel(T::is_leg(2), lam_all[NR * 2 + c], lam_all[NR * 3 + c])));
        QS_PHASE_G(10)
        o.foot_force = lam_own[0] * qrcp(dt);   // getContactPoints()[9] = normal impulse / dt
        s.warm = lam_own[0];

        // delta v = H^-1 J^T lambda :  dv_b = L^-T sum_i w_i lam_i ;  dqd = sum_own u_r lam_r - (B K)^T dv_b
        V z[6];
        if (PAY) {   // the base part of J^T lambda: the foot rows' as without the block, plus the payload rows'; the block's velocity change
            const PayRows& q = *pq;
#pragma unroll
            for (int i = 0; i < 6; i++) {
                V t = rows[0].w[i] * lam_own[0];
#pragma unroll
                for (int r = 1; r < NR; r++) t = t + rows[r].w[i] * lam_own[r];
                V zp = q.w[0][i] * plam[0];
#pragma unroll
                for (int k = 1; k < 6; k++) zp = zp + q.w[k][i] * plam[PAY ? k : 0];
                z[i] = T::quad_sum(t) + zp;
            }
#pragma unroll
            for (int k = 0; k < 6; k++) pq->lam[k] = plam[PAY ? k : 0];
            V3v dwb = pja[0] * (q.mI * plam[0]);
            dwb = dwb + pja[PAY ? 1 : 0] * (q.mI * plam[PAY ? 1 : 0]);
            dwb = dwb + pja[PAY ? 2 : 0] * (q.mI * plam[PAY ? 2 : 0]);
            dwb = dwb - mk3<V>(plam[PAY ? 3 : 0], plam[PAY ? 4 : 0], plam[PAY ? 5 : 0]) * q.mI;
            pq->dw = dwb; pq->dv = mk3<V>(-(q.mM * plam[0]), -(q.mM * plam[PAY ? 1 : 0]), -(q.mM * plam[PAY ? 2 : 0]));
        } else {
#pragma unroll
            for (int i = 0; i < 6; i++) {
                V t = rows[0].w[i] * lam_own[0];
#pragma unroll
                for (int r = 1; r < NR; r++) t = t + rows[r].w[i] * lam_own[r];
                z[i] = T::quad_sum(t);
            }
        }
        ltsolve6<V>(Lc, Ld, z);
#pragma unroll
        for (int j = 0; j < 3; j++) {
            V t = rows[0].u[j] * lam_own[0];
#pragma unroll
            for (int r = 1; r < NR; r++) t = t + rows[r].u[j] * lam_own[r];
#pragma unroll
            for (int i = 0; i < 6; i++) t = t - BK[j][i] * z[i];
            s.qd[j] = clampv<V>(s.qd[j] + t, V(-cfg.vel_cap), V(cfg.vel_cap));
        }
        const V cap = V(cfg.vel_cap);
        s.vang.x = clampv<V>(s.vang.x + R[0] * z[0] + R[1] * z[1] + R[2] * z[2], -cap, cap);
        s.vang.y = clampv<V>(s.vang.y + R[3] * z[0] + R[4] * z[1] + R[5] * z[2], -cap, cap);
        s.vang.z = clampv<V>(s.vang.z + R[6] * z[0] + R[7] * z[1] + R[8] * z[2], -cap, cap);
        s.vlin.x = clampv<V>(s.vlin.x + R[0] * z[3] + R[1] * z[4] + R[2] * z[5], -cap, cap);
        s.vlin.y = clampv<V>(s.vlin.y + R[3] * z[3] + R[4] * z[4] + R[5] * z[5], -cap, cap);
        s.vlin.z = clampv<V>(s.vlin.z + R[6] * z[3] + R[7] * z[4] + R[8] * z[5], -cap, cap);
    }

    // The rare path: rows beyond the three foot-contact rows of a leg -- one row per violated joint limit (falls), normal + friction rows
    // of up to two more support points of the leg (trunk corner, hip housing, thigh ends, knee end of the calf: a fallen robot rests on
    // them), the six rows of the payload block's fixed constraint (cfg.payload_soft, quadruped.py:778-819: three hold the block's pivot on
    // the base origin, three hold the frames parallel; impulse bound 500 N x dt, ERP cfg.joint_erp; swept with the joint-limit rows as
    // Bullet sorts them).  Row layout of a leg: contact point c = 0 .. 2 (0 = the foot) at rows 3c (normal), 3c + 1, 3c + 2 (friction), joint
    // limits at rows 9 + j.  The solve itself is RareSolver (qs_rare.h): one environment at a time, one row per lane of the wave; this is
    // what comes after it, in the quad layout again: delta v = H^-1 J^T lambda from the impulses lam[12] of the lane's rows (and the block's
    // plam[6]):  dv_b = L^-T sum_i w_i lam_i ;  dqd = sum_own u_r lam_r - (B K)^T dv_b.
    static QS_FN void integrate_rare(const qs_config& cfg, State& s, Out& o, const Row* xr, const V* lam, const V* Sm, const V* Ld, const V (*BK)[6], const V* R,
                                     PayRows* pay, const V* plam, V foot_act) {
        const float dt = (float)cfg.dt;
        // (foot_act: slot 0 holds the FOOT's rows -- with the foot off the ground it may hold the leg's lowest support point, round 6)
        o.foot_force = lam[0] * qrcp(dt) * foot_act;
        s.warm = lam[0] * foot_act;
        V z[6], x[3];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            V t = xr[0].w[i] * lam[0];
#pragma unroll
            for (int r = 1; r < 12; r++) t = t + xr[r].w[i] * lam[r];
            z[i] = T::quad_sum(t);
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
            V t = xr[0].u[j] * lam[0];
#pragma unroll
            for (int r = 1; r < 12; r++) t = t + xr[r].u[j] * lam[r];
            x[j] = t;
        }
        if (pay) {   // the block's rows: their base part of J^T lambda (they act on every lane's copy alike), the block's velocity change
            PayRows& q = *pay;
#pragma unroll
            for (int i = 0; i < 6; i++) {
                V zp = q.w[0][i] * plam[0];
#pragma unroll
                for (int k = 1; k < 6; k++) zp = zp + q.w[k][i] * plam[k];
                z[i] = z[i] + zp;
            }
#pragma unroll
            for (int k = 0; k < 6; k++) q.lam[k] = plam[k];
            const V zero = V(0.0f);
            const V3v ja0 = mk3<V>(zero, -q.rB.z, q.rB.y), ja1 = mk3<V>(q.rB.z, zero, -q.rB.x), ja2 = mk3<V>(-q.rB.y, q.rB.x, zero);   // -(rB x e_k)
            V3v dwb = ja0 * (q.mI * plam[0]);
            dwb = dwb + ja1 * (q.mI * plam[1]);
            dwb = dwb + ja2 * (q.mI * plam[2]);
            dwb = dwb - mk3<V>(plam[3], plam[4], plam[5]) * q.mI;
            q.dw = dwb; q.dv = mk3<V>(-(q.mM * plam[0]), -(q.mM * plam[1]), -(q.mM * plam[2]));
        }
        ltsolve6<V>(Sm, Ld, z);
        const V cap = V(cfg.vel_cap);
#pragma unroll
        for (int j = 0; j < 3; j++) {
            V t = x[j];
#pragma unroll
            for (int i = 0; i < 6; i++) t = t - BK[j][i] * z[i];
            s.qd[j] = clampv<V>(s.qd[j] + t, -cap, cap);
        }
        s.vang.x = clampv<V>(s.vang.x + R[0] * z[0] + R[1] * z[1] + R[2] * z[2], -cap, cap);
        s.vang.y = clampv<V>(s.vang.y + R[3] * z[0] + R[4] * z[1] + R[5] * z[2], -cap, cap);
        s.vang.z = clampv<V>(s.vang.z + R[6] * z[0] + R[7] * z[1] + R[8] * z[2], -cap, cap);
        s.vlin.x = clampv<V>(s.vlin.x + R[0] * z[3] + R[1] * z[4] + R[2] * z[5], -cap, cap);
        s.vlin.y = clampv<V>(s.vlin.y + R[3] * z[3] + R[4] * z[4] + R[5] * z[5], -cap, cap);
        s.vlin.z = clampv<V>(s.vlin.z + R[6] * z[3] + R[7] * z[4] + R[8] * z[5], -cap, cap);
    }

    // ---- link-link tests of the self-collision rule (rare path)
    // Do two boxes (centre, orthonormal axes, half extents) overlap?  Separating-axis test over the 15 candidate axes -- what Bullet's
    // btBoxBoxDetector (dBoxBox2) decides the contact of a box / box pair by (no contact while any axis separates).
    struct H3 { float v[3]; QS_FN float operator[](int i) const { return v[i]; } };
    static QS_FN M obb_overlap(V3v ca, V3v a0, V3v a1, V3v a2, H3 ha, V3v cb, V3v b0, V3v b1, V3v b2, H3 hb) {
        V3v t0 = cb - ca;
        V3v A[3] = {a0, a1, a2}, B[3] = {b0, b1, b2};
        V t[3] = {dot(t0, a0), dot(t0, a1), dot(t0, a2)};
        V R[3][3], AR[3][3];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) { R[i][j] = dot(A[i], B[j]); AR[i][j] = qabs(R[i][j]) + 1e-6f; }
        M sep = qlt(V(1.0f), V(0.0f));
#pragma unroll
        for (int i = 0; i < 3; i++) sep = qor(sep, qgt(qabs(t[i]), V(ha[i]) + (AR[i][0] * hb[0] + AR[i][1] * hb[1] + AR[i][2] * hb[2])));
#pragma unroll
        for (int j = 0; j < 3; j++)
            sep = qor(sep, qgt(qabs(t[0] * R[0][j] + t[1] * R[1][j] + t[2] * R[2][j]), (AR[0][j] * ha[0] + AR[1][j] * ha[1] + AR[2][j] * ha[2]) + hb[j]));
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
                V ra = AR[i2][j] * ha[i1] + AR[i1][j] * ha[i2], rb = AR[i][j2] * hb[j1] + AR[i][j1] * hb[j2];
                sep = qor(sep, qgt(qabs(t[i2] * R[i1][j] - t[i1] * R[i2][j]), ra + rb));
            }
        return qnot(sep);
    }
    // distance from a sphere's surface to a box (negative inside)
    static QS_FN V sphere_box_dist(V3v c, float r, V3v cb, V3v b0, V3v b1, V3v b2, H3 hb) {
        V3v d = c - cb;
        V l0 = dot(d, b0), l1 = dot(d, b1), l2 = dot(d, b2);
        V e0 = l0 - clampv<V>(l0, V(-hb[0]), V(hb[0])), e1 = l1 - clampv<V>(l1, V(-hb[1]), V(hb[1])), e2 = l2 - clampv<V>(l2, V(-hb[2]), V(hb[2]));
        return qsqrt(e0 * e0 + e1 * e1 + e2 * e2) - r;
    }
    struct LegGeom { V3v p1, ct, X2, Y, Z2, cc, X3, Z3, rf; };   // hip centre; thigh box; calf box; foot centre (base coordinates)
    template <int K> static QS_FN LegGeom partner(const LegGeom& g) {
        LegGeom r;
#define QS_X3(f) r.f = mk3<V>(T::template xorl<K>(g.f.x), T::template xorl<K>(g.f.y), T::template xorl<K>(g.f.z));
        QS_X3(p1) QS_X3(ct) QS_X3(X2) QS_X3(Y) QS_X3(Z2) QS_X3(cc) QS_X3(X3) QS_X3(Z3) QS_X3(rf)
#undef QS_X3
        return r;
    }
    // contacts of the own calf with the links of leg (own ^ K): thigh and calf boxes, hip housing and foot as spheres; a calf / calf pair
    // is seen from both of its lanes and counts one half in each
    template <int K> static QS_FN V calf_vs_leg(const LegGeom& g) {
        using namespace go1;
        const LegGeom p = partner<K>(g);
        const H3 CALF_H = {{CALF_HALF[0], CALF_HALF[1], CALF_HALF[2]}}, THIGH_H = {{THIGH_HALF[0], THIGH_HALF[1], THIGH_HALF[2]}};
        V n = qflag(obb_overlap(g.cc, g.X3, g.Y, g.Z3, CALF_H, p.ct, p.X2, p.Y, p.Z2, THIGH_H));
        n = n + qflag(obb_overlap(g.cc, g.X3, g.Y, g.Z3, CALF_H, p.cc, p.X3, p.Y, p.Z3, CALF_H)) * 0.5f;
        n = n + qflag(qlt(sphere_box_dist(p.p1, HIP_SELF_R, g.cc, g.X3, g.Y, g.Z3, CALF_H), V(THR_HIP)));
        n = n + qflag(qlt(sphere_box_dist(p.rf, FOOT_R, g.cc, g.X3, g.Y, g.Z3, CALF_H), V(THR_FOOT)));
        return n;
    }
    // number of link-link contacts that involve a calf (quadruped.py:237-241), own-lane share; c1..c3, ct: broad-phase flags
    static QS_FN V self_contacts(const LegGeom& g, V c1, V c2, V c3, V ctr) {
        using namespace go1;
        V n = calf_vs_leg<1>(g) * c1 + calf_vs_leg<2>(g) * c2 + calf_vs_leg<3>(g) * c3;
        V3v o = mk3<V>(V(0.0f), V(0.0f), V(0.0f)), ex = mk3<V>(V(1.0f), V(0.0f), V(0.0f)), ey = mk3<V>(V(0.0f), V(1.0f), V(0.0f)), ez = mk3<V>(V(0.0f), V(0.0f), V(1.0f));
        const H3 CALF_H = {{CALF_HALF[0], CALF_HALF[1], CALF_HALF[2]}}, TRUNK_H = {{TRUNK_HALF[0], TRUNK_HALF[1], TRUNK_HALF[2]}};
        n = n + qflag(obb_overlap(g.cc, g.X3, g.Y, g.Z3, CALF_H, o, ex, ey, ez, TRUNK_H)) * ctr;
        return n;
    }

    // returns true iff HOT and the wave needs a rare path: `s` (and the payload block's state) is then as it came in, and the caller repeats
    // this substep with the full build.
    // ---- cfg.payload_soft: the payload block as a body of its own (quadruped.py:778-819: createMultiBody(box of half extent 0.05) +
    // createConstraint(base, block, JOINT_FIXED, child pivot -delta) = a btMultiBodyFixedConstraint in the same PGS as the contacts)
    // puts the block where the constraint wants it, at the base's velocity (a reset's spawn, qs_set_state, qs_set_params)
    static QS_FN void block_place(float* blk, const State& s, const Par& Pr) {
        V x = s.qx, y = s.qy, z = s.qz, w = s.qw;
        V sc = V(2.0f) * qrcp(x * x + y * y + z * z + w * w);
        V xs = x * sc, ys = y * sc, zs = z * sc;
        V wx = w * xs, wy = w * ys, wz = w * zs, xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
        const V3v r = Pr.r_pay;
        V3v d = mk3<V>((V(1.0f) - (yy + zz)) * r.x + (xy - wz) * r.y + (xz + wy) * r.z, (xy + wz) * r.x + (V(1.0f) - (xx + zz)) * r.y + (yz - wx) * r.z,
                       (xz - wy) * r.x + (yz + wx) * r.y + (V(1.0f) - (xx + yy)) * r.z);
        V3v v = s.vlin + cross(s.vang, d);
        T::st(blk, B_POS, s.pos.x + d.x); T::st(blk, B_POS + 1, s.pos.y + d.y); T::st(blk, B_POS + 2, s.pos.z + d.z);
        T::st(blk, B_QUAT, s.qx); T::st(blk, B_QUAT + 1, s.qy); T::st(blk, B_QUAT + 2, s.qz); T::st(blk, B_QUAT + 3, s.qw);
        T::st(blk, B_V, v.x); T::st(blk, B_V + 1, v.y); T::st(blk, B_V + 2, v.z);
        T::st(blk, B_W, s.vang.x); T::st(blk, B_W + 1, s.vang.y); T::st(blk, B_W + 2, s.vang.z);
        for (int k = 0; k < 7; k++) T::st(blk, B_LAM + k, V(0.0f));
    }
    // the six rows of the fixed constraint from the predicted velocities; applies gravity to the block (its only other force: the
    // inertia is isotropic, so there is no gyroscopic term) and leaves the result in the record
    static QS_FN void payload_rows(const qs_config& cfg, const Par& Pr, const State& s, const Spv& vs, V3v Rx, V3v Ry, V3v Rz, const V* Sm, const V* Ld,
                                   float* blk, PayRows& q) {
        const float dt = (float)cfg.dt;
        const V zero = V(0.0f), one = V(1.0f);
        V3v bp = mk3<V>(T::ld(blk, B_POS), T::ld(blk, B_POS + 1), T::ld(blk, B_POS + 2));
        V bx = T::ld(blk, B_QUAT), by = T::ld(blk, B_QUAT + 1), bz = T::ld(blk, B_QUAT + 2), bw = T::ld(blk, B_QUAT + 3);
        V3v bv = mk3<V>(T::ld(blk, B_V), T::ld(blk, B_V + 1), T::ld(blk, B_V + 2) - dt * cfg.gravity);
        V3v bo = mk3<V>(T::ld(blk, B_W), T::ld(blk, B_W + 1), T::ld(blk, B_W + 2));
        T::st(blk, B_V + 2, bv.z);
        V3v rB;
        {
            V sc = V(2.0f) * qrcp(bx * bx + by * by + bz * bz + bw * bw);
            V xs = bx * sc, ys = by * sc, zs = bz * sc;
            V wx = bw * xs, wy = bw * ys, wz = bw * zs, xx = bx * xs, xy = bx * ys, xz = bx * zs, yy = by * ys, yz = by * zs, zz = bz * zs;
            V3v n = mk3<V>(-Pr.r_pay.x, -Pr.r_pay.y, -Pr.r_pay.z);
            rB = mk3<V>((one - (yy + zz)) * n.x + (xy - wz) * n.y + (xz + wy) * n.z, (xy + wz) * n.x + (one - (xx + zz)) * n.y + (yz - wx) * n.z,
                        (xz - wy) * n.x + (yz + wx) * n.y + (one - (xx + yy)) * n.z);
        }
        V perr[3] = {s.pos.x - (bp.x + rB.x), s.pos.y - (bp.y + rB.y), s.pos.z - (bp.z + rB.z)};   // pivot on the base (its origin) - pivot on the block
        T::st(blk, B_GAP, qsqrt(perr[0] * perr[0] + perr[1] * perr[1] + perr[2] * perr[2]));
        // orientation error: rotation vector of q_base q_block^-1 (world), small angle
        V ex = s.qw * (-bx) + s.qx * bw + s.qy * (-bz) - s.qz * (-by);
        V ey = s.qw * (-by) - s.qx * (-bz) + s.qy * bw + s.qz * (-bx);
        V ez = s.qw * (-bz) + s.qx * (-by) - s.qy * (-bx) + s.qz * bw;
        V ew = s.qw * bw - s.qx * (-bx) - s.qy * (-by) - s.qz * (-bz);
        V sg = qsel(qlt(ew, zero), V(-2.0f), V(2.0f));
        V aerr[3] = {sg * ex, sg * ey, sg * ez};
        M has = qgt(Pr.m_pay, zero);
        V mp = qsel(has, Pr.m_pay, one);
        q.act = qflag(has);
        q.mM = qrcp(mp); q.mI = qrcp(mp * go1::PAYLOAD_I);
        q.rB = rB;
        V3v ja[3] = {mk3<V>(zero, -rB.z, rB.y), mk3<V>(rB.z, zero, -rB.x), mk3<V>(-rB.y, rB.x, zero)};   // -(rB x e_k)
        V3v ab[3] = {Rx, Ry, Rz};                                                                         // R^T e_k
        V bvk[3] = {bv.x, bv.y, bv.z}, bok[3] = {bo.x, bo.y, bo.z};
        const V erp = V(cfg.joint_erp * (1.0f / dt));
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int c = k % 3;
            if (k < 3) { q.w[k][0] = zero; q.w[k][1] = zero; q.w[k][2] = zero; q.w[k][3] = ab[c].x * q.act; q.w[k][4] = ab[c].y * q.act; q.w[k][5] = ab[c].z * q.act; }
            else { q.w[k][0] = ab[c].x * q.act; q.w[k][1] = ab[c].y * q.act; q.w[k][2] = ab[c].z * q.act; q.w[k][3] = zero; q.w[k][4] = zero; q.w[k][5] = zero; }
            lsolve6<V>(Sm, Ld, q.w[k]);
            V diag = k < 3 ? q.mI * dot(ja[c], ja[c]) + q.mM : q.mI;
#pragma unroll
            for (int i = 0; i < 6; i++) diag = diag + q.w[k][i] * q.w[k][i];
            q.diag[k] = diag; q.dinv[k] = qrcp(diag);
            V rel = k < 3 ? dot(ab[c], vs.l) + dot(ja[c], bo) - bvk[c] : dot(ab[c], vs.a) - bok[c];
            V err = k < 3 ? perr[c] : aerr[c];
            q.rhs[k] = ((-err) * erp - rel) * q.dinv[k] * q.act;
        }
    }
    // after the solve: the block's velocities, then its pose (semi-implicit Euler, exponential map like the base)
    static QS_FN void payload_integrate(const qs_config& cfg, float* blk, const PayRows& q) {
        const float dt = (float)cfg.dt;
        const V one = V(1.0f);
        V3v bv = mk3<V>(T::ld(blk, B_V) + q.dv.x, T::ld(blk, B_V + 1) + q.dv.y, T::ld(blk, B_V + 2) + q.dv.z);
        V3v bo = mk3<V>(T::ld(blk, B_W) + q.dw.x, T::ld(blk, B_W + 1) + q.dw.y, T::ld(blk, B_W + 2) + q.dw.z);
        V bx = T::ld(blk, B_QUAT), by = T::ld(blk, B_QUAT + 1), bz = T::ld(blk, B_QUAT + 2), bw = T::ld(blk, B_QUAT + 3);
        T::st(blk, B_V, bv.x); T::st(blk, B_V + 1, bv.y); T::st(blk, B_V + 2, bv.z);
        T::st(blk, B_W, bo.x); T::st(blk, B_W + 1, bo.y); T::st(blk, B_W + 2, bo.z);
        T::st(blk, B_POS, T::ld(blk, B_POS) + dt * bv.x); T::st(blk, B_POS + 1, T::ld(blk, B_POS + 1) + dt * bv.y); T::st(blk, B_POS + 2, T::ld(blk, B_POS + 2) + dt * bv.z);
        V th2 = dot(bo, bo) * (dt * dt);
        V sc = V(0.5f * dt) * (one - th2 * (1.0f / 24.0f) * (one - th2 * (1.0f / 80.0f)));
        V dw = one - th2 * 0.125f * (one - th2 * (1.0f / 48.0f));
        V dx = bo.x * sc, dy = bo.y * sc, dz = bo.z * sc;
        V nx = dw * bx + dx * bw + dy * bz - dz * by;
        V ny = dw * by - dx * bz + dy * bw + dz * bx;
        V nz = dw * bz + dx * by - dy * bx + dz * bw;
        V nw = dw * bw - dx * bx - dy * by - dz * bz;
        V inv = qrsqrt(nx * nx + ny * ny + nz * nz + nw * nw);
        T::st(blk, B_QUAT, nx * inv); T::st(blk, B_QUAT + 1, ny * inv); T::st(blk, B_QUAT + 2, nz * inv); T::st(blk, B_QUAT + 3, nw * inv);
#pragma unroll
        for (int k = 0; k < 6; k++) T::st(blk, B_LAM + k, q.lam[k]);
    }

    // results of the block's rows: the many-rows solver's for an environment with rare rows of its own, the common-path solver's otherwise
    static QS_FN void keep_rare_payload(PayRows& c, const PayRows& r, M rare_mine) {
#pragma unroll
        for (int k = 0; k < 6; k++) c.lam[k] = qsel(rare_mine, r.lam[k], c.lam[k]);
        c.dw = mk3<V>(qsel(rare_mine, r.dw.x, c.dw.x), qsel(rare_mine, r.dw.y, c.dw.y), qsel(rare_mine, r.dw.z, c.dw.z));
        c.dv = mk3<V>(qsel(rare_mine, r.dv.x, c.dv.x), qsel(rare_mine, r.dv.y, c.dv.y), qsel(rare_mine, r.dv.z, c.dv.z));
    }

    // `detect`: classify the contacts of the non-foot links, the payload block and the link-link pairs (o.n_invalid).  The reference reads
    // GetContactInfo after the LAST stepSimulation of an env step (task_base.py:137-147 via gym_env.py:241-245), so the callers ask for it
    // there only -- unless cfg.body_contacts, where those links' heights decide in every substep whether they push back.
    // `blk`: the payload block's state in the record (R_BLOCK) under cfg.payload_soft, nullptr otherwise
    // `scratch_row`: this environment's observation row in the wave's staging area (unused between two epilogues): the many-rows solve
    // borrows the sixteen rows of the wave (RareSolver)
    // `last`: the contact classification is READ after this substep (the last one of an env step: gym_env.py:241-245); the link-link tests of
    // the self-collision rule only run then -- they decide nothing about the motion (round 4: with cfg.body_contacts they ran in every
    // substep, 11 k cycles each for a robot lying on folded legs)
    // Returns (HOT builds; the full build always 0): 0 = done; 1 = this wave needs a rare path IN this substep: nothing has been written,
    // the env step goes on in the full build from this substep; 2 = done, and the wave will need the rare path in the NEXT substep (a link in
    // its contact range whose rows cannot act yet but will by then, predicted from this substep's closing speeds): the caller hands over at
    // the substep boundary instead of letting the next substep run into its vote (round 6: that vote sits behind the substep's dynamics --
    // 10 k cycles of a 16 k substep, thrown away on the path of the wave every launch waits for).  A wrong guess costs time only: results
    // do not depend on where the switch happens (Env::step).
    static QS_FN int substep(const qs_config& cfg, const Par& Pr, State& s, const V* tau, Out& o, bool detect = true, float* blk = nullptr, float* scratch_row = nullptr,
                              bool last = true) {
        using namespace go1;
        if (HOT && !SOFT && cfg.payload_soft) return 1;   // this build holds no payload rows
        bool hand_over_next = false;
        const bool soft = (!HOT || SOFT) && cfg.payload_soft && blk != nullptr;
        Model P;
        {   // opaque copies keep the compiler from hoisting the 24 leg constants out of the substep loop (where they would
            // occupy registers for the whole env step); rebuilding them is ~40 multiplications per substep
            V ml[3] = {Pr.m_leg[0], Pr.m_leg[1], Pr.m_leg[2]};
            T::opaque(ml[0]); T::opaque(ml[1]); T::opaque(ml[2]);
            build_model(cfg, P, ml);
        }
        QS_PHASE_BEGIN
        const float dt = (float)cfg.dt;
        const V zero = V(0.0f), one = V(1.0f);
        V fx = T::fx(), sy = T::sy();
        // ---- base rotation (row-major, base -> world), velocities in base coordinates
        V R[9];
        {
            V x = s.qx, y = s.qy, z = s.qz, w = s.qw;
            V d = x * x + y * y + z * z + w * w;
            V sc = V(2.0f) * qrcp(d);
            V xs = x * sc, ys = y * sc, zs = z * sc;
            V wx = w * xs, wy = w * ys, wz = w * zs, xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
            R[0] = one - (yy + zz); R[1] = xy - wz; R[2] = xz + wy;
            R[3] = xy + wz; R[4] = one - (xx + zz); R[5] = yz - wx;
            R[6] = xz - wy; R[7] = yz + wx; R[8] = one - (xx + yy);
        }
        V3v Rx = mk3<V>(R[0], R[1], R[2]), Ry = mk3<V>(R[3], R[4], R[5]), Rz = mk3<V>(R[6], R[7], R[8]);  // R^T e_x, R^T e_y, R^T e_z
        Spv v0;
        v0.a = mk3<V>(R[0] * s.vang.x + R[3] * s.vang.y + R[6] * s.vang.z, R[1] * s.vang.x + R[4] * s.vang.y + R[7] * s.vang.z, R[2] * s.vang.x + R[5] * s.vang.y + R[8] * s.vang.z);
        v0.l = mk3<V>(R[0] * s.vlin.x + R[3] * s.vlin.y + R[6] * s.vlin.z, R[1] * s.vlin.x + R[4] * s.vlin.y + R[7] * s.vlin.z, R[2] * s.vlin.x + R[5] * s.vlin.y + R[8] * s.vlin.z);

        QS_PHASE(1)
        // ---- leg kinematics in base coordinates
        V s1, c1, s2, c2, s23, c23;
        qsincos(s.q[0], s1, c1); qsincos(s.q[1], s2, c2); qsincos(s.q[1] + s.q[2], s23, c23);
        V3v p1 = mk3<V>(fx * HIP_X, sy * HIP_Y, zero);
        V3v ax1 = mk3<V>(one, zero, zero);
        V3v Y = mk3<V>(zero, c1, s1);                       // joint axis of thigh and calf
        V3v Z1 = mk3<V>(zero, -s1, c1);
        V3v p2 = p1 + Y * (sy * THIGH_Y);
        V3v X2 = mk3<V>(c2, s1 * s2, -c1 * s2), Z2 = mk3<V>(s2, -s1 * c2, c1 * c2);
        V3v p3 = p2 + Z2 * V(LEG_Z);
        V3v X3 = mk3<V>(c23, s1 * s23, -c1 * s23), Z3 = mk3<V>(s23, -s1 * c23, c1 * c23);
        V3v rf = p3 + Z3 * V(LEG_Z);                        // foot centre

        QS_PHASE(2)
        // ---- link inertias about the base origin
        SI<V> I1 = part_inertia<V>(P.m_hip, P.c_hip, P.I_hip, p1, ax1, Y, Z1);
        SI<V> I2 = part_inertia<V>(P.m_thigh, P.c_thigh, P.I_thigh, p2, X2, Y, Z2);
        SI<V> I3 = part_inertia<V>(P.m_calf, mk3<V>(V(CALF_C[0]), V(CALF_C[1]), V(CALF_C[2])), P.I_calf, p3, X3, Y, Z3) +
                   point_inertia<V>(V(FOOT_M), V(FOOT_I), rf);
        // motion subspaces S_j = (a_j ; p_j x a_j)
        Spv S1, S2, S3j;
        S1.a = ax1; S1.l = cross(p1, ax1);
        S2.a = Y; S2.l = cross(p2, Y);
        S3j.a = Y; S3j.l = cross(p3, Y);
        QS_PHASE(3)
        // ---- RNEA bias with qdd = 0, base acceleration 0, gravity as the fictitious base acceleration -g
        Spv a0; a0.a = mk3<V>(zero, zero, zero); a0.l = Rz * V(cfg.gravity);
        Spv vj1, vj2, vj3;
        vj1.a = S1.a * s.qd[0]; vj1.l = S1.l * s.qd[0];
        vj2.a = S2.a * s.qd[1]; vj2.l = S2.l * s.qd[1];
        vj3.a = S3j.a * s.qd[2]; vj3.l = S3j.l * s.qd[2];
        Spv v1 = v0 + vj1, v2 = v1 + vj2, v3 = v2 + vj3;
        Spv a1 = crm_add(a0, v0, vj1), a2 = crm_add(a1, v1, vj2), a3 = crm_add(a2, v2, vj3);
        Spv f1 = crf_add(apply(I1, a1), v1, apply(I1, v1));
        Spv f2 = crf_add(apply(I2, a2), v2, apply(I2, v2));
        Spv f3 = crf_add(apply(I3, a3), v3, apply(I3, v3));
        Spv fs2 = f2 + f3, fs1 = f1 + fs2;
        V C1 = dot(S1, fs1), C2 = dot(S2, fs2), C3 = dot(S3j, f3);
        Spv f0 = crf_add(apply(Pr.I0, a0), v0, apply(Pr.I0, v0));
        V Cb[6] = {T::quad_sum(fs1.a.x) + f0.a.x, T::quad_sum(fs1.a.y) + f0.a.y, T::quad_sum(fs1.a.z) + f0.a.z,
                   T::quad_sum(fs1.l.x) + f0.l.x, T::quad_sum(fs1.l.y) + f0.l.y, T::quad_sum(fs1.l.z) + f0.l.z};
        QS_PHASE(4)
        // ---- CRBA: B = [F1 F2 F3] (6x3), D (3x3 symmetric)
        SI<V> Ic2 = I2 + I3, Ic1 = I1 + Ic2;
        Spv F1 = apply(Ic1, S1), F2 = apply(Ic2, S2), F3 = apply(I3, S3j);
        V D11 = dot(S1, F1), D12 = dot(S1, F2), D13 = dot(S1, F3), D22 = dot(S2, F2), D23 = dot(S2, F3), D33 = dot(S3j, F3);
        // K = D^-1 (symmetric cofactor inverse)
        V K11, K12, K13, K22, K23, K33;
        {
            V c11 = D22 * D33 - D23 * D23, c12 = D13 * D23 - D12 * D33, c13 = D12 * D23 - D13 * D22;
            V det = D11 * c11 + D12 * c12 + D13 * c13;
            V id = qrcp(det);
            K11 = c11 * id; K12 = c12 * id; K13 = c13 * id;
            K22 = (D11 * D33 - D13 * D13) * id; K23 = (D12 * D13 - D11 * D23) * id; K33 = (D11 * D22 - D12 * D12) * id;
        }
        V Bm[3][6] = {{F1.a.x, F1.a.y, F1.a.z, F1.l.x, F1.l.y, F1.l.z}, {F2.a.x, F2.a.y, F2.a.z, F2.l.x, F2.l.y, F2.l.z}, {F3.a.x, F3.a.y, F3.a.z, F3.l.x, F3.l.y, F3.l.z}};
        V BK[3][6];   // BK[j] = column j of B K
#pragma unroll
        for (int i = 0; i < 6; i++) {
            BK[0][i] = Bm[0][i] * K11 + Bm[1][i] * K12 + Bm[2][i] * K13;
            BK[1][i] = Bm[0][i] * K12 + Bm[1][i] * K22 + Bm[2][i] * K23;
            BK[2][i] = Bm[0][i] * K13 + Bm[1][i] * K23 + Bm[2][i] * K33;
        }
        QS_PHASE(5)
        // ---- S = Hbb - sum_legs B K B^T, then Cholesky (replicated)
        SI<V> Itot;
        Itot.m = Pr.mtot;
        {
            SI<V> Il = Ic1;
            Itot.h = mk3<V>(T::quad_sum(Il.h.x), T::quad_sum(Il.h.y), T::quad_sum(Il.h.z)) + Pr.I0.h;
            Itot.I.xx = T::quad_sum(Il.I.xx) + Pr.I0.I.xx; Itot.I.xy = T::quad_sum(Il.I.xy) + Pr.I0.I.xy; Itot.I.xz = T::quad_sum(Il.I.xz) + Pr.I0.I.xz;
            Itot.I.yy = T::quad_sum(Il.I.yy) + Pr.I0.I.yy; Itot.I.yz = T::quad_sum(Il.I.yz) + Pr.I0.I.yz; Itot.I.zz = T::quad_sum(Il.I.zz) + Pr.I0.I.zz;
        }
        V Sm[21];
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j <= i; j++) Sm[tri(i, j)] = T::quad_sum(BK[0][i] * Bm[0][j] + BK[1][i] * Bm[1][j] + BK[2][i] * Bm[2][j]);
        {
            V H[21];
            H[tri(0, 0)] = Itot.I.xx; H[tri(1, 0)] = Itot.I.xy; H[tri(1, 1)] = Itot.I.yy; H[tri(2, 0)] = Itot.I.xz; H[tri(2, 1)] = Itot.I.yz; H[tri(2, 2)] = Itot.I.zz;
            // rows 3..5 (linear) x cols 0..2 (angular): -[h]x  ; diagonal block m
            H[tri(3, 0)] = zero; H[tri(3, 1)] = Itot.h.z; H[tri(3, 2)] = -Itot.h.y;
            H[tri(4, 0)] = -Itot.h.z; H[tri(4, 1)] = zero; H[tri(4, 2)] = Itot.h.x;
            H[tri(5, 0)] = Itot.h.y; H[tri(5, 1)] = -Itot.h.x; H[tri(5, 2)] = zero;
            H[tri(3, 3)] = Itot.m; H[tri(4, 3)] = zero; H[tri(4, 4)] = Itot.m; H[tri(5, 3)] = zero; H[tri(5, 4)] = zero; H[tri(5, 5)] = Itot.m;
#pragma unroll
            for (int i = 0; i < 21; i++) Sm[i] = H[i] - Sm[i];
        }
        V Ld[6];
        chol6<V>(Sm, Ld);

        QS_PHASE(6)
        // ---- unconstrained accelerations
        V t1 = tau[0] - C1, t2 = tau[1] - C2, t3 = tau[2] - C3;
        V y1 = K11 * t1 + K12 * t2 + K13 * t3, y2 = K12 * t1 + K22 * t2 + K23 * t3, y3 = K13 * t1 + K23 * t2 + K33 * t3;
        V ab[6];
#pragma unroll
        for (int i = 0; i < 6; i++) ab[i] = -Cb[i] - T::quad_sum(Bm[0][i] * y1 + Bm[1][i] * y2 + Bm[2][i] * y3);
        lsolve6<V>(Sm, Ld, ab);
        ltsolve6<V>(Sm, Ld, ab);
        V qdd[3] = {y1, y2, y3};
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int i = 0; i < 6; i++) qdd[j] = qdd[j] - BK[j][i] * ab[i];
        QS_PHASE(7)
        // ---- collision: foot sphere vs plane z = 0 ; the other link primitives (trunk box, hip cylinder, thigh and calf boxes, payload
        // block) count as invalid contacts (quadruped.py:243-249) and, with cfg.body_contacts, push back on the rare path below
        const V zc = s.pos.z;
        const V grf = dot(Rz, rf);
        V dist = zc + grf - FOOT_R;
        M act_m = qlt(dist, V(THR_FOOT));
        V active = qflag(act_m);
        o.foot_contact = active;
        // height of the lowest vertex of each primitive (this lane: the trunk corner on its own side, its hip, the two ends of its
        // thigh box, the two ends of its calf box)
        V az = zero, gx2 = zero, gx3 = zero, h_trunk = zero, h_hip = zero, h_th_hi = zero, h_th_lo = zero, h_cf_hi = zero;
        M any_extra = qlt(one, zero);
        if (detect) {
            az = dot(Rz, Y); gx2 = dot(Rz, X2); gx3 = dot(Rz, X3);
            const V gp2 = dot(Rz, p2), gp3 = dot(Rz, p3);
            h_trunk = zc + Rz.x * (fx * TRUNK_HALF[0]) + Rz.y * (sy * TRUNK_HALF[1]) - qabs(Rz.z) * TRUNK_HALF[2];
            h_hip = zc + dot(Rz, p1) - HIP_CYL_HALF_LEN * qabs(az) - HIP_CYL_R * qsqrt(qmax(one - az * az, zero));
            const V th_off = qabs(gx2) * THIGH_HALF[0] + qabs(az) * THIGH_HALF[1], cf_off = (qabs(gx3) + qabs(az)) * CALF_HALF[0];
            h_th_hi = zc + gp2 - th_off; h_th_lo = zc + gp3 - th_off; h_cf_hi = zc + gp3 - cf_off;
            const V h_cf_lo = zc + grf - cf_off;
            M m_trunk = qlt(h_trunk, V(THR_TRUNK)), m_hip = qlt(h_hip, V(THR_HIP));
            M m_thigh = qlt(qmin(h_th_hi, h_th_lo), V(THR_THIGH)), m_calf = qlt(qmin(h_cf_hi, h_cf_lo), V(THR_CALF));
            V trunk = qflag(qgt(T::quad_sum(qflag(m_trunk)), zero));
            V n = qflag(m_hip) + qflag(m_thigh) + qflag(m_calf);
            // payload block (a second body in the reference, quadruped.py:778-819: a plane / block contact is an invalid one, :248-249)
            V h_pay = zc + dot(Rz, Pr.r_pay) - (qabs(Rz.x) + qabs(Rz.y) + qabs(Rz.z)) * PAYLOAD_HALF;
            if (soft) {   // the block's own pose: third row of its rotation matrix
                V bx = T::ld(blk, B_QUAT), by = T::ld(blk, B_QUAT + 1), bz = T::ld(blk, B_QUAT + 2), bw = T::ld(blk, B_QUAT + 3);
                V sc = V(2.0f) * qrcp(bx * bx + by * by + bz * bz + bw * bw);
                h_pay = T::ld(blk, B_POS + 2) - (qabs((bx * bz - bw * by) * sc) + qabs((by * bz + bw * bx) * sc) + qabs(one - (bx * bx + by * by) * sc)) * PAYLOAD_HALF;
            }
            V pay = qflag(qand(qgt(Pr.m_pay, zero), qlt(h_pay, V(THR_PAYLOAD))));
            o.n_invalid = T::quad_sum(n) + trunk + pay;
            if (cfg.body_contacts) any_extra = qor(qor(m_trunk, m_hip), qor(m_thigh, qlt(h_cf_hi, V(THR_CALF))));
        }
        if (detect && last && cfg.self_collision) {
            // Link-link contacts count only when a calf is involved (quadruped.py:237-241).  Broad phase, every substep: extents of
            // the own calf and of the own whole leg towards the robot's centre planes, in mirrored coordinates (sy y: towards the own
            // side, fx x: towards the own end); thigh within 0.021 of its axis, calf and foot within 0.02, hip housing within 0.046.
            const float MARGIN = 0.005f;
            V y2 = sy * p2.y, y3 = sy * p3.y, yf = sy * rf.y, x3 = fx * p3.x, xf = fx * rf.x;
            V mc = qmin(y3, yf) - 0.02f, nc = qmin(x3, xf) - 0.02f;
            V ml = qmin(qmin(qmin(y2, y3) - 0.021f, yf - 0.02f), V(HIP_Y - HIP_SELF_R));
            V nl = qmin(qmin(x3 - 0.021f, xf - 0.02f), V(HIP_X - HIP_SELF_R));
            M cy1 = qlt(mc + T::template xorl<1>(ml), V(MARGIN));                                   // other side, same end
            M cx2 = qlt(nc + T::template xorl<2>(nl), V(MARGIN));                                   // same side, other end
            M c3 = qand(qlt(mc + T::template xorl<3>(ml), V(MARGIN)), qlt(nc + T::template xorl<3>(nl), V(MARGIN)));   // diagonal
            M ctr = qand(qand(qlt(mc, V(TRUNK_HALF[1] + MARGIN)), qgt(qmax(p3.z, rf.z) + 0.02f, V(-TRUNK_HALF[2] - MARGIN))), qlt(nc, V(TRUNK_HALF[0] + MARGIN)));
            if (__builtin_expect(T::any(qor(qor(cy1, cx2), qor(c3, ctr))), 0)) {
                if (HOT) return 1;
                T::count_self_narrow(cfg);
                LegGeom lg;
                lg.p1 = p1; lg.ct = p2 + Z2 * V(LINK_BOX_Z); lg.X2 = X2; lg.Y = Y; lg.Z2 = Z2; lg.cc = p3 + Z3 * V(LINK_BOX_Z); lg.X3 = X3; lg.Z3 = Z3; lg.rf = rf;
                o.n_invalid = o.n_invalid + T::quad_sum(self_contacts(lg, qflag(cy1), qflag(cx2), qflag(c3), qflag(ctr)));
            }
        }

        // joint limits: a row exists only while the limit is violated (btMultiBodyJointLimitConstraint)
        V lim_pen[3], lim_sgn[3], lim_act[3];
        M any_lim = qlt(one, zero);
#pragma unroll
        for (int j = 0; j < 3; j++) {
            V plo = s.q[j] - JLO[j], phi = V(JHI[j]) - s.q[j];
            M vlo = qle(plo, zero), vhi = qle(phi, zero);
            lim_pen[j] = qsel(vlo, plo, phi);
            lim_sgn[j] = qsel(vlo, one, -one);
            M v = qor(vlo, vhi);
            lim_act[j] = qflag(v);
            any_lim = qor(any_lim, v);
        }
        // Nothing to solve when no foot of the wave's 16 environments is within contact range and no joint sits at a stop:
        // every row would be inactive (rhs = lambda = 0), i.e. delta v = 0 exactly.  Flight phases of a whole wave skip the
        // rows, the Delassus columns and the sweeps.
        if (HOT && __builtin_expect(T::any(any_lim), 0)) return 1;
        if (HOT && __builtin_expect(T::any(any_extra), 0)) {
            // A non-foot link is inside its contact range.  The full build gives a support point rows only once its normal row can act on the
            // predicted velocities (cfg.support_margin, below); while every point in range is still approaching, its result IS the common-path
            // result -- so the wave stays here.  Same predicate, same expressions, evaluated for all five candidates of the leg (the full
            // build looks at the two lowest: a subset).  Cold code: it runs in the three or four substeps a falling link spends in the
            // 4-mm range, which the full build took at 25 k cycles each instead of 16 k (round 5: ranges scaled to a quarter, as a
            // diagnostic, read 67.7 against 63.6 M).
            const V cap_ = V(cfg.vel_cap), inv_dt_ = V(qrcp(dt));
            V3v wxv_ = cross(v0.a, v0.l);
            V3v al_ = mk3<V>(ab[3] + wxv_.x, ab[4] + wxv_.y, ab[5] + wxv_.z);
            V3v wa_, wl_;
            wa_.x = clampv<V>(s.vang.x + dt * (R[0] * ab[0] + R[1] * ab[1] + R[2] * ab[2]), -cap_, cap_);
            wa_.y = clampv<V>(s.vang.y + dt * (R[3] * ab[0] + R[4] * ab[1] + R[5] * ab[2]), -cap_, cap_);
            wa_.z = clampv<V>(s.vang.z + dt * (R[6] * ab[0] + R[7] * ab[1] + R[8] * ab[2]), -cap_, cap_);
            wl_.x = clampv<V>(s.vlin.x + dt * (R[0] * al_.x + R[1] * al_.y + R[2] * al_.z), -cap_, cap_);
            wl_.y = clampv<V>(s.vlin.y + dt * (R[3] * al_.x + R[4] * al_.y + R[5] * al_.z), -cap_, cap_);
            wl_.z = clampv<V>(s.vlin.z + dt * (R[6] * al_.x + R[7] * al_.y + R[8] * al_.z), -cap_, cap_);
            V qp_[3];
#pragma unroll
            for (int j = 0; j < 3; j++) qp_[j] = clampv<V>(s.qd[j] + dt * qdd[j], -cap_, cap_);
            Spv vp_;
            vp_.a = mk3<V>(R[0] * wa_.x + R[3] * wa_.y + R[6] * wa_.z, R[1] * wa_.x + R[4] * wa_.y + R[7] * wa_.z, R[2] * wa_.x + R[5] * wa_.y + R[8] * wa_.z);
            vp_.l = mk3<V>(R[0] * wl_.x + R[3] * wl_.y + R[6] * wl_.z, R[1] * wl_.x + R[4] * wl_.y + R[7] * wl_.z, R[2] * wl_.x + R[5] * wl_.y + R[8] * wl_.z);
            V sgz_ = qsel(qlt(Rz.z, zero), -one, one), sga_ = qsel(qlt(az, zero), -one, one);
            V sg2_ = qsel(qlt(gx2, zero), -one, one), sg3_ = qsel(qlt(gx3, zero), -one, one);
            V rad_ = HIP_CYL_R * qrsqrt(qmax(one - az * az, V(1e-12f)));
            V3v pc_[5];
            pc_[0] = mk3<V>(fx * TRUNK_HALF[0], sy * TRUNK_HALF[1], sgz_ * (-TRUNK_HALF[2]));
            pc_[1] = p1 - Y * (sga_ * HIP_CYL_HALF_LEN) - (Rz - Y * az) * rad_;
            V3v toff_ = X2 * (sg2_ * THIGH_HALF[0]) + Y * (sga_ * THIGH_HALF[1]);
            pc_[2] = p2 - toff_; pc_[3] = p3 - toff_;
            pc_[4] = p3 - X3 * (sg3_ * CALF_HALF[0]) - Y * (sga_ * CALF_HALF[1]);
            const V hc_[5] = {h_trunk, h_hip, h_th_hi, h_th_lo, h_cf_hi};
            const float thr_[5] = {THR_TRUNK, THR_HIP, THR_THIGH, THR_THIGH, THR_CALF};
            const int dep_[5] = {0, 1, 2, 2, 3};   // joints of the leg that move the point
            M can_act = qlt(one, zero), can_next = qlt(one, zero);
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const V3v pt = pc_[i];
                const V f1 = V(dep_[i] > 0 ? 1.0f : 0.0f), f2 = V(dep_[i] > 1 ? 1.0f : 0.0f), f3 = V(dep_[i] > 2 ? 1.0f : 0.0f);
                V3v e1 = cross(ax1, pt - p1) * f1, e2 = cross(Y, pt - p2) * f2, e3 = cross(Y, pt - p3) * f3;
                V3v jan = cross(pt, Rz);
                V reln = jan.x * vp_.a.x + jan.y * vp_.a.y + jan.z * vp_.a.z + Rz.x * vp_.l.x + Rz.y * vp_.l.y + Rz.z * vp_.l.z +
                         dot(Rz, e1) * qp_[0] + dot(Rz, e2) * qp_[1] + dot(Rz, e3) * qp_[2];
                V pen_x = hc_[i] + cfg.contact_slop;
                can_act = qor(can_act, qand(qlt(hc_[i], V(thr_[i])), qor(qle(pen_x, zero), qgt((-reln) - pen_x * inv_dt_, V(-cfg.support_margin)))));
                // the same predicate one substep on, at this substep's closing speed: the gap has shrunk by reln dt
                can_next = qor(can_next, qand(qlt(hc_[i] + reln * dt, V(thr_[i])), qgt((-2.0f) * reln - pen_x * inv_dt_, V(-cfg.support_margin))));
            }
            if (T::any(can_act)) return 1;
            hand_over_next = T::any(can_next);
        }
        // ---- v* = v + dt a (world frame for the base; classical acceleration of the origin = a_lin + w x v).  The first write to `s` of the
        // substep: a HOT build's wave that gives up (above) hands the state back as it came, and the env step goes on in the full build from
        // THIS substep (qs_env.h).  (The first version updated the velocities before the collision phase and kept the old ones for that
        // case: 23 more AGPR copies in the hot loop, -2.5 % on the headline.)
        const V cap = V(cfg.vel_cap);
        {
            V3v wxv = cross(v0.a, v0.l);
            V3v al = mk3<V>(ab[3] + wxv.x, ab[4] + wxv.y, ab[5] + wxv.z);
            s.vang.x = clampv<V>(s.vang.x + dt * (R[0] * ab[0] + R[1] * ab[1] + R[2] * ab[2]), -cap, cap);
            s.vang.y = clampv<V>(s.vang.y + dt * (R[3] * ab[0] + R[4] * ab[1] + R[5] * ab[2]), -cap, cap);
            s.vang.z = clampv<V>(s.vang.z + dt * (R[6] * ab[0] + R[7] * ab[1] + R[8] * ab[2]), -cap, cap);
            s.vlin.x = clampv<V>(s.vlin.x + dt * (R[0] * al.x + R[1] * al.y + R[2] * al.z), -cap, cap);
            s.vlin.y = clampv<V>(s.vlin.y + dt * (R[3] * al.x + R[4] * al.y + R[5] * al.z), -cap, cap);
            s.vlin.z = clampv<V>(s.vlin.z + dt * (R[6] * al.x + R[7] * al.y + R[8] * al.z), -cap, cap);
#pragma unroll
            for (int j = 0; j < 3; j++) s.qd[j] = clampv<V>(s.qd[j] + dt * qdd[j], -cap, cap);
        }
        Spv vs;  // predicted base velocity in base coordinates
        vs.a = mk3<V>(R[0] * s.vang.x + R[3] * s.vang.y + R[6] * s.vang.z, R[1] * s.vang.x + R[4] * s.vang.y + R[7] * s.vang.z, R[2] * s.vang.x + R[5] * s.vang.y + R[8] * s.vang.z);
        vs.l = mk3<V>(R[0] * s.vlin.x + R[3] * s.vlin.y + R[6] * s.vlin.z, R[1] * s.vlin.x + R[4] * s.vlin.y + R[7] * s.vlin.z, R[2] * s.vlin.x + R[5] * s.vlin.y + R[8] * s.vlin.z);
        if (soft || T::any(qor(qor(act_m, any_lim), any_extra))) {
        QS_PHASE(8)
        // ---- constraint rows of this leg: 0 normal, 1 t1 = -y_world, 2 t2 = +x_world ; 3..5 joint limits (rare path)
        Row rows[6];
#pragma unroll
        for (int j = 0; j < 3; j++) rows[3 + j].act = lim_act[j];
        V3v rc = rf - Rz * V(FOOT_R);                   // contact point on the sphere, base coordinates
        V3v d1 = rc - p1, d2 = rc - p2, d3 = rc - p3;
        V3v g1 = cross(ax1, d1), g2 = cross(Y, d2), g3 = cross(Y, d3);  // d(point)/dq_j
        const V inv_dt = V(qrcp(dt));   // v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division
        // one row of a contact at the point RC (base coordinates) of a link whose point moves by G1, G2, G3 per unit joint velocity,
        // DIST above the plane; btMultiBodyConstraintSolver::setupMultiBodyContactConstraint: penetration = distance + linear slop
#define QS_CONTACT_ROW_AT(ROW, DIR, NORMAL, RC, G1, G2, G3, DIST, ACT)                                                 \
    {                                                                                                                  \
        Row& r_ = ROW;                                                                                                 \
        /* the row direction of an inactive contact is the zero vector: jq, u, w and the relative velocity vanish with it (no   \
           masking multiplications further down); its 1 / diag is a finite 1e30 */                                     \
        V3v d_ = (DIR) * (ACT);                                                                                        \
        V3v ja = cross(RC, d_);                                                                                        \
        r_.jq[0] = dot(d_, G1); r_.jq[1] = dot(d_, G2); r_.jq[2] = dot(d_, G3);                                        \
        r_.u[0] = K11 * r_.jq[0] + K12 * r_.jq[1] + K13 * r_.jq[2];                                                    \
        r_.u[1] = K12 * r_.jq[0] + K22 * r_.jq[1] + K23 * r_.jq[2];                                                    \
        r_.u[2] = K13 * r_.jq[0] + K23 * r_.jq[1] + K33 * r_.jq[2];                                                    \
        V jb[6] = {ja.x, ja.y, ja.z, d_.x, d_.y, d_.z};                                                                \
        /* jb - B u as three chained FMAs per component */                                                            \
        _Pragma("unroll") for (int i = 0; i < 6; i++) {                                                                \
            V t_ = jb[i] - Bm[0][i] * r_.u[0];                                                                         \
            t_ = t_ - Bm[1][i] * r_.u[1];                                                                              \
            r_.w[i] = t_ - Bm[2][i] * r_.u[2];                                                                         \
        }                                                                                                              \
        lsolve6<V>(Sm, Ld, r_.w);                                                                                      \
        V diag = r_.jq[0] * r_.u[0] + r_.jq[1] * r_.u[1] + r_.jq[2] * r_.u[2];                                         \
        _Pragma("unroll") for (int i = 0; i < 6; i++) diag = diag + r_.w[i] * r_.w[i];                                 \
        r_.dinv = qrcp(qmax(diag, V(1e-30f))); r_.diag = diag;                                                         \
        V rel = ja.x * vs.a.x + ja.y * vs.a.y + ja.z * vs.a.z + d_.x * vs.l.x + d_.y * vs.l.y + d_.z * vs.l.z +       \
                r_.jq[0] * s.qd[0] + r_.jq[1] * s.qd[1] + r_.jq[2] * s.qd[2];                                          \
        if (NORMAL) {                                                                                                  \
            V pen_ = (DIST) + cfg.contact_slop;                                                                        \
            V pos_err = qsel(qgt(pen_, zero), zero, (-pen_) * (cfg.contact_erp * inv_dt));                             \
            V vel_err = (-rel) - qsel(qgt(pen_, zero), pen_ * inv_dt, zero);                                           \
            r_.rhs = (pos_err + vel_err) * r_.dinv * (ACT);                                                            \
        } else {                                                                                                       \
            r_.rhs = (-rel) * r_.dinv;                                                                                 \
        }                                                                                                              \
        r_.act = (ACT);                                                                                                \
    }
#define QS_CONTACT_ROW(IDX, DIR, NORMAL) QS_CONTACT_ROW_AT(rows[IDX], DIR, NORMAL, rc, g1, g2, g3, dist, active)
        QS_CONTACT_ROW(0, Rz, true)
        QS_CONTACT_ROW(1, (mk3<V>(-Ry.x, -Ry.y, -Ry.z)), false)
        QS_CONTACT_ROW(2, Rx, false)
#undef QS_CONTACT_ROW
        V Kc[3][3] = {{K11, K12, K13}, {K12, K22, K23}, {K13, K23, K33}};
        // joint-limit rows of this leg into dst[0..2]
#define QS_LIMIT_ROWS(DST)                                                                                             \
    _Pragma("unroll") for (int j = 0; j < 3; j++) {                                                                    \
        Row& r_ = (DST)[j];                                                                                            \
        r_.act = lim_act[j];                                                                                           \
        V a_ = r_.act * lim_sgn[j];                                                                                    \
        _Pragma("unroll") for (int c = 0; c < 3; c++) { r_.jq[c] = c == j ? a_ : zero; r_.u[c] = Kc[c][j] * a_; }      \
        _Pragma("unroll") for (int i = 0; i < 6; i++) r_.w[i] = -BK[j][i] * a_;                                        \
        lsolve6<V>(Sm, Ld, r_.w);                                                                                      \
        V diag = Kc[j][j];                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 6; i++) diag = diag + r_.w[i] * r_.w[i];                                 \
        r_.dinv = qrcp(diag); r_.diag = diag;                                                                          \
        V rel = lim_sgn[j] * s.qd[j];                                                                                  \
        r_.rhs = ((-lim_pen[j]) * (cfg.joint_erp * inv_dt) - rel) * r_.dinv * r_.act;                                  \
    }
        // this environment has rare rows of its own and takes the many-rows solver's result: a joint at its stop, or a support point whose
        // rows are built (below)
        M rare_mine = qgt(T::quad_sum(qflag(any_lim)), V(0.5f));
        // The many-rows solve comes FIRST in the full build (on a copy; only its few results stay live while the common-path solver runs:
        // the other way round, that one's copies of the state sat on the registers the many-rows part is short of).
        State s_r = s; Out o_r = o;
        PayRows pay_c, pay_r;   // cfg.payload_soft: the block's six rows (built once: payload_rows also applies gravity to the block) and the two solvers' results for them
        if (!HOT && soft) payload_rows(cfg, Pr, s, vs, Rx, Ry, Rz, Sm, Ld, blk, pay_c);
        QS_PHASE_G(39)
        if (!HOT && T::any(qor(any_lim, any_extra))) {
            Row xr[12];   // this leg's rows: contact point c at 3c .. 3c + 2 (0 = the foot), joint limits at 9 + j
#pragma unroll
            for (int r = 0; r < 3; r++) xr[r] = rows[r];
#pragma unroll
            for (int r = 3; r < 12; r++) {
                Row& e_ = xr[r];
                e_.rhs = zero; e_.dinv = zero; e_.act = zero; e_.diag = zero;
#pragma unroll
                for (int i = 0; i < 6; i++) e_.w[i] = zero;
#pragma unroll
                for (int i = 0; i < 3; i++) { e_.jq[i] = zero; e_.u[i] = zero; }
            }
            M extra_live = qlt(one, zero);
#if defined(QS_PROBE_WARM) && defined(__HIP_DEVICE_COMPILE__)
            M probe_live[2] = {qlt(one, zero), qlt(one, zero)};
#endif
#if defined(QS_PROBE_LAZY) && defined(__HIP_DEVICE_COMPILE__)
            M probe_rule[2][4];
#pragma unroll
            for (int a_ = 0; a_ < 2; a_++)
#pragma unroll
                for (int m_ = 0; m_ < 4; m_++) probe_rule[a_][m_] = qlt(one, zero);
#endif
            const M lim_env = qgt(T::quad_sum(qflag(any_lim)), V(0.5f));   // a joint of this environment sits at its stop
            if (T::any(any_extra)) {
                // A non-foot primitive of some environment of the wave is within its contact range: up to two support points per leg
                // besides the foot, the lowest of {own trunk corner, hip housing, the two ends of the thigh box, knee end of the calf box}.
                const V bigh = V(1e9f);
                V hh[5] = {qsel(qlt(h_trunk, V(THR_TRUNK)), h_trunk, bigh), qsel(qlt(h_hip, V(THR_HIP)), h_hip, bigh), qsel(qlt(h_th_hi, V(THR_THIGH)), h_th_hi, bigh),
                           qsel(qlt(h_th_lo, V(THR_THIGH)), h_th_lo, bigh), qsel(qlt(h_cf_hi, V(THR_CALF)), h_cf_hi, bigh)};
                // candidate points (base coordinates): lowest vertex of each primitive
                V sgz = qsel(qlt(Rz.z, zero), -one, one), sga = qsel(qlt(az, zero), -one, one);
                V sg2 = qsel(qlt(gx2, zero), -one, one), sg3 = qsel(qlt(gx3, zero), -one, one);
                V rad = HIP_CYL_R * qrsqrt(qmax(one - az * az, V(1e-12f)));
                V3v pc[5];
                pc[0] = mk3<V>(fx * TRUNK_HALF[0], sy * TRUNK_HALF[1], sgz * (-TRUNK_HALF[2]));
                pc[1] = p1 - Y * (sga * HIP_CYL_HALF_LEN) - (Rz - Y * az) * rad;
                V3v toff = X2 * (sg2 * THIGH_HALF[0]) + Y * (sga * THIGH_HALF[1]);
                pc[2] = p2 - toff; pc[3] = p3 - toff;
                pc[4] = p3 - X3 * (sg3 * CALF_HALF[0]) - Y * (sga * CALF_HALF[1]);
                const float depth[5] = {0.0f, 1.0f, 2.0f, 2.0f, 3.0f};   // joints of the leg that move the point
                // A leg carries at most three contact points: its foot and two support points -- or, with the foot off the ground, THREE support
                // points (round 6; rounds 2-5: two whatever the foot did.  A robot on its back rests on trunk corner + both ends of each thigh
                // box: with two of the three the choice flipped from substep to substep, the robot crept at 7 mm/s and lay 1-2 mm off where four
                // points per primitive put it; with three it lies within 3e-7 m of that, at rest).  Pass 0 gives the lowest
                // candidate of such a leg the FOOT's row slot, which is empty there (an inactive contact's rows are all zero): the leg's points
                // stay lowest-first in Bullet's sweep order, the row layout, the lanes of the many-rows solve and its instantiations stay as they
                // are.  Legs whose foot touches sit pass 0 out.
                const M foot_off = qnot(act_m);
#pragma unroll
                for (int pass = 0; pass < 3; pass++) {
                    const int slot = pass - 1;   // -1: the foot's slot (rows 0 .. 2 of xr), 0 / 1: rows 3 .. 5 / 6 .. 8
                    const M elig = pass == 0 ? foot_off : qlt(zero, one);
                    V best = qsel(elig, hh[0], bigh), bi = zero;
#pragma unroll
                    for (int i = 1; i < 5; i++) { M m = qand(elig, qlt(hh[i], best)); best = qsel(m, hh[i], best); bi = qsel(m, V((float)i), bi); }
                    if (!T::any(qlt(best, V(1e8f)))) { if (pass == 0) continue; else break; }   // nobody in the wave has a point for this slot: its rows stay empty
                    bi = qsel(qlt(best, V(1e8f)), bi, V(4.0f));   // empty slot: any point that the joints move (an all-zero Jacobian has no 1 / diag)
                    V3v pt = pc[0]; V dep = V(depth[0]);
#pragma unroll
                    for (int i = 1; i < 5; i++) {
                        M m = qgt(bi, V(i - 0.5f));   // bi >= i: later candidates overwrite
                        pt = mk3<V>(qsel(m, pc[i].x, pt.x), qsel(m, pc[i].y, pt.y), qsel(m, pc[i].z, pt.z));
                        dep = qsel(m, V(depth[i]), dep);
                    }
#pragma unroll
                    for (int i = 0; i < 5; i++) hh[i] = qsel(qand(qlt(best, V(1e8f)), qand(qgt(bi, V(i - 0.5f)), qlt(bi, V(i + 0.5f)))), bigh, hh[i]);   // taken
                    V dist_x = qsel(qlt(best, V(1e8f)), best, zero);
                    V f1 = qflag(qgt(dep, V(0.5f))), f2 = qflag(qgt(dep, V(1.5f))), f3 = qflag(qgt(dep, V(2.5f)));
                    V3v e1 = cross(ax1, pt - p1) * f1, e2 = cross(Y, pt - p2) * f2, e3 = cross(Y, pt - p3) * f3;
                    // The point's rows are built once its normal row can act in this substep.  A contact inside the range but still
                    // APPROACHING is speculative (btMultiBodyConstraintSolver: a positive distance allows a closing speed of distance / dt):
                    // its rows end every sweep at zero impulse and the environment has the common-path solver's result -- 81 % of the
                    // environment-substeps that reach this code in the benchmark with body_contacts=True (4 mm range, robots falling at
                    // ~1 m/s).  Left out: a point whose predicted closing speed (v* of this substep) stays cfg.support_margin (0.5 m/s) short of what
                    // the gap allows.  Measured with a counting build over 8868 such environment-substeps: at a margin of 0 m/s 0.76 % of
                    // the rows left out would have ended with an impulse (the feet's impulses tilt the trunk), at 0.25 m/s none; the
                    // rule's margin is twice that.  An environment with a joint AT ITS STOP keeps all its points: the
                    // impact at the stop changes the leg's velocities by metres per second inside the solve, which no margin on v*
                    // covers (tests/test_emu_vs_oracle.py::test_joint_limits_together_with_sliding_contacts found it).
                    V3v jan = cross(pt, Rz);
                    V reln = jan.x * vs.a.x + jan.y * vs.a.y + jan.z * vs.a.z + Rz.x * vs.l.x + Rz.y * vs.l.y + Rz.z * vs.l.z +
                             dot(Rz, e1) * s.qd[0] + dot(Rz, e2) * s.qd[1] + dot(Rz, e3) * s.qd[2];
                    V pen_x = dist_x + cfg.contact_slop;
                    M live = qand(qlt(best, V(1e8f)), qor(qor(lim_env, qle(pen_x, zero)), qgt((-reln) - pen_x * inv_dt, V(-cfg.support_margin))));
#if defined(QS_PROBE_LAZY) && defined(__HIP_DEVICE_COMPILE__)
                    // counting build (tools/probe_lazy_rows.py): every point in range gets its rows, as before the rule; what the rule
                    // would have said at four margins is kept next to it
#pragma unroll
                    for (int m_ = 0; m_ < 4; m_++) probe_rule[slot][m_] = qand(qlt(best, V(1e8f)), qor(qor(lim_env, qle(pen_x, zero)), qgt((-reln) - pen_x * inv_dt, V(-0.25f * (float)((1 << m_) >> 1)))));
                    live = qlt(best, V(1e8f));
#endif
                    extra_live = qor(extra_live, live);
#if defined(QS_PROBE_WARM) && defined(__HIP_DEVICE_COMPILE__)
                    probe_live[pass == 0 ? 0 : slot] = pass == 0 ? probe_live[0] : live;
#endif
                    if (!T::any(live)) continue;               // nobody's point in this slot can act: no rows
                    V act_x = qflag(live);
                    if (pass == 0) {
                        // into the foot's slot of the legs that have such a point (their foot is off the ground: its rows there are all zero); the
                        // other legs keep what the slot holds
                        Row t0_, t1_, t2_;
                        QS_CONTACT_ROW_AT(t0_, Rz, true, pt, e1, e2, e3, dist_x, act_x)
                        QS_CONTACT_ROW_AT(t1_, (mk3<V>(-Ry.x, -Ry.y, -Ry.z)), false, pt, e1, e2, e3, dist_x, act_x)
                        QS_CONTACT_ROW_AT(t2_, Rx, false, pt, e1, e2, e3, dist_x, act_x)
                        const Row* tt_[3] = {&t0_, &t1_, &t2_};
#pragma unroll
                        for (int r = 0; r < 3; r++) {
                            Row& d_ = xr[r]; const Row& a_ = *tt_[r];
#pragma unroll
                            for (int i = 0; i < 6; i++) d_.w[i] = qsel(live, a_.w[i], d_.w[i]);
#pragma unroll
                            for (int i = 0; i < 3; i++) { d_.jq[i] = qsel(live, a_.jq[i], d_.jq[i]); d_.u[i] = qsel(live, a_.u[i], d_.u[i]); }
                            d_.rhs = qsel(live, a_.rhs, d_.rhs); d_.dinv = qsel(live, a_.dinv, d_.dinv); d_.diag = qsel(live, a_.diag, d_.diag); d_.act = qsel(live, a_.act, d_.act);
                        }
                    } else {
                        QS_CONTACT_ROW_AT(xr[3 + 3 * slot], Rz, true, pt, e1, e2, e3, dist_x, act_x)
                        QS_CONTACT_ROW_AT(xr[4 + 3 * slot], (mk3<V>(-Ry.x, -Ry.y, -Ry.z)), false, pt, e1, e2, e3, dist_x, act_x)
                        QS_CONTACT_ROW_AT(xr[5 + 3 * slot], Rx, false, pt, e1, e2, e3, dist_x, act_x)
                    }
                }
            }
            rare_mine = qgt(T::quad_sum(qflag(qor(any_lim, extra_live))), V(0.5f));
#if defined(QS_PROBE_WARM) && defined(__HIP_DEVICE_COMPILE__)
            {   // counting build (tools/probe_warm.py): what the environments that reach the many-rows solve look like.  [0] environment-substeps
                // with rows of their own, [1] of them with a joint at its stop, [2] no stop but a leg with two live support points, [3..5] no stop,
                // at most one point per leg: 1 / 2 / 3+ points in all; [6] wave-substeps in the full build, [7] of them with a many-rows solve,
                // [8] of them in which every environment with rows of its own is "simple" (no stop, one point per leg at most)
                unsigned long long* pc = reinterpret_cast<const QsDevCfg&>(cfg).counters + 2;
                const float pts_leg = qflag(probe_live[0]) + qflag(probe_live[1]);
                const float pts = T::quad_sum(pts_leg), two = T::quad_sum(pts_leg > 1.5f ? 1.0f : 0.0f);
                const bool mine = rare_mine, lim = lim_env;
                const bool simple = mine && !lim && two < 0.5f;
                const unsigned long long b0 = __ballot(mine), b1 = __ballot(mine && lim), b2 = __ballot(mine && !lim && two > 0.5f),
                                         b3 = __ballot(simple && pts < 1.5f), b4 = __ballot(simple && pts > 1.5f && pts < 2.5f), b5 = __ballot(simple && pts > 2.5f);
                if (threadIdx.x == 0) {
                    atomicAdd(&pc[0], (unsigned long long)(__popcll(b0) / 4)); atomicAdd(&pc[1], (unsigned long long)(__popcll(b1) / 4));
                    atomicAdd(&pc[2], (unsigned long long)(__popcll(b2) / 4)); atomicAdd(&pc[3], (unsigned long long)(__popcll(b3) / 4));
                    atomicAdd(&pc[4], (unsigned long long)(__popcll(b4) / 4)); atomicAdd(&pc[5], (unsigned long long)(__popcll(b5) / 4));
                    atomicAdd(&pc[6], 1ull);
                    if (b0) atomicAdd(&pc[7], 1ull);
                    if (b0 && b0 == (b3 | b4 | b5)) atomicAdd(&pc[8], 1ull);
                }
            }
#endif
            if (T::any(rare_mine)) {
                T::count_rare_path(cfg);
                if (T::any(any_lim)) { QS_LIMIT_ROWS(xr + 9) }
                QS_PHASE_G(40)
                V lam12[12], plam[6];
                if (soft) pay_r = pay_c;   // (the rows; the results in it are overwritten)
                RareSolver<T, CONE>::solve(cfg, Pr.mu, xr, soft ? &pay_r : nullptr, rare_mine, s.warm * cfg.warmstart * rows[0].act, T::wave_scratch(scratch_row), lam12, plam);
                integrate_rare(cfg, s_r, o_r, xr, lam12, Sm, Ld, BK, R, soft ? &pay_r : nullptr, plam, active);
#if defined(QS_PROBE_LAZY) && defined(__HIP_DEVICE_COMPILE__)
                {   // probe[0] environment-substeps with a support point in range, [1] of them with every such row at zero impulse and no joint
                    // at its stop, [2 + 2 m] environments the rule at margin m would have sent to the many-rows solve, [3 + 2 m] environments in
                    // which it would have left out a point whose normal row ended with an impulse; margins 0, 0.25, 0.5, 1.0 m/s
                    unsigned long long* pc = reinterpret_cast<const QsDevCfg&>(cfg).counters + 2;
                    const bool ext_q = T::quad_sum(qflag(any_extra)) > 0.5f;
                    const bool zero_q = !lim_env && T::quad_sum(qabs(lam12[3]) + qabs(lam12[6])) == 0.0f;
                    const unsigned long long b0 = __ballot(ext_q), b1 = __ballot(ext_q && zero_q);
                    if (threadIdx.x == 0) { atomicAdd(&pc[0], (unsigned long long)(__popcll(b0) / 4)); atomicAdd(&pc[1], (unsigned long long)(__popcll(b1) / 4)); }
#pragma unroll
                    for (int m_ = 0; m_ < 4; m_++) {
                        const bool keep_q = lim_env || T::quad_sum(qflag(probe_rule[0][m_]) + qflag(probe_rule[1][m_])) > 0.5f;
                        const float w0 = !probe_rule[0][m_] && qabs(lam12[3]) > 0.0f ? 1.0f : 0.0f, w1 = !probe_rule[1][m_] && qabs(lam12[6]) > 0.0f ? 1.0f : 0.0f;
                        const bool wrong_q = T::quad_sum(w0 + w1) > 0.5f;
                        const unsigned long long bk = __ballot(ext_q && keep_q), bw = __ballot(ext_q && wrong_q);
                        if (threadIdx.x == 0) { atomicAdd(&pc[2 + 2 * m_], (unsigned long long)(__popcll(bk) / 4)); atomicAdd(&pc[3 + 2 * m_], (unsigned long long)(__popcll(bw) / 4)); }
                    }
                }
#endif
                QS_PHASE_G(45)
            }
        }
        // The common-path solve.  In the full build's wave on a rare path the environments that have NO rare row of their own (no joint at a
        // stop, no link on the floor) keep ITS result -- bit for bit what the common-path build gives them in a wave without such a
        // neighbour; only the environments with rare rows take the many-rows solver's.  (Both solvers run for the whole wave: wave votes
        // must not sit under a divergent branch.)
        if (SOFT || (!HOT && soft)) {
            if (HOT) payload_rows(cfg, Pr, s, vs, Rx, Ry, Rz, Sm, Ld, blk, pay_c);      // (a handle without the block does not launch this build)
            if (cfg.solver_residual_threshold > 0.0f) solve_and_integrate<3, true, true>(cfg, Pr.mu, s, o, rows, Sm, Ld, BK, R, &pay_c);
            else solve_and_integrate<3, false, true>(cfg, Pr.mu, s, o, rows, Sm, Ld, BK, R, &pay_c);
            if (HOT) payload_integrate(cfg, blk, pay_c);
        } else {
            if (cfg.solver_residual_threshold > 0.0f) solve_and_integrate<3, true>(cfg, Pr.mu, s, o, rows, Sm, Ld, BK, R);
            else solve_and_integrate<3, false>(cfg, Pr.mu, s, o, rows, Sm, Ld, BK, R);
        }
        if (!HOT) {   // (in a wave without any rare row: the common-path result for everybody)
#pragma unroll
            for (int j = 0; j < 3; j++) s.qd[j] = qsel(rare_mine, s_r.qd[j], s.qd[j]);
            s.vang = mk3<V>(qsel(rare_mine, s_r.vang.x, s.vang.x), qsel(rare_mine, s_r.vang.y, s.vang.y), qsel(rare_mine, s_r.vang.z, s.vang.z));
            s.vlin = mk3<V>(qsel(rare_mine, s_r.vlin.x, s.vlin.x), qsel(rare_mine, s_r.vlin.y, s.vlin.y), qsel(rare_mine, s_r.vlin.z, s.vlin.z));
            s.warm = qsel(rare_mine, s_r.warm, s.warm); o.foot_force = qsel(rare_mine, o_r.foot_force, o.foot_force);
            if (soft) { keep_rare_payload(pay_c, pay_r, rare_mine); payload_integrate(cfg, blk, pay_c); }
        }
#undef QS_LIMIT_ROWS
#undef QS_CONTACT_ROW_AT
        } else {
            o.foot_force = zero; s.warm = zero;
        }
        QS_PHASE(11)
        // ---- positions: semi-implicit Euler, quaternion by the exponential map of w_world * dt
        s.pos.x = s.pos.x + dt * s.vlin.x; s.pos.y = s.pos.y + dt * s.vlin.y; s.pos.z = s.pos.z + dt * s.vlin.z;
        {
            // exp map of w*dt: (w * sin(th/2)/|w|, cos(th/2)), th = |w| dt <= vel_cap*sqrt(3)*dt: series in th^2
            V th2 = dot(s.vang, s.vang) * (dt * dt);
            V sc = V(0.5f * dt) * (one - th2 * (1.0f / 24.0f) * (one - th2 * (1.0f / 80.0f)));
            V dw = one - th2 * 0.125f * (one - th2 * (1.0f / 48.0f));
            V dx = s.vang.x * sc, dy = s.vang.y * sc, dz = s.vang.z * sc;
            V nx = dw * s.qx + dx * s.qw + dy * s.qz - dz * s.qy;
            V ny = dw * s.qy - dx * s.qz + dy * s.qw + dz * s.qx;
            V nz = dw * s.qz + dx * s.qy - dy * s.qx + dz * s.qw;
            V nw = dw * s.qw - dx * s.qx - dy * s.qy - dz * s.qz;
            V inv = qrsqrt(nx * nx + ny * ny + nz * nz + nw * nw);
            s.qx = nx * inv; s.qy = ny * inv; s.qz = nz * inv; s.qw = nw * inv;
        }
#pragma unroll
        for (int j = 0; j < 3; j++) s.q[j] = s.q[j] + dt * s.qd[j];
        QS_PHASE(12)
        QS_PHASE_END
        return HOT && hand_over_next ? 2 : 0;
    }
};

}  // namespace qs
