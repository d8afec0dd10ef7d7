// qs_lane.h -- lane abstraction for the quad-per-environment kernels.
//
// One environment is simulated by a QUAD of four adjacent lanes of a wavefront, one lane per leg
// (FR, FL, RR, RL = PyBullet motor order, quadruped.py:586-596); a 64-wide wavefront carries 16 environments.
// Per-leg quantities are lane-private, base quantities are replicated over the quad, and the only cross-lane
// traffic is a 4-lane sum / broadcast, which maps onto DPP quad_perm moves (no LDS, no ds_bpermute).
//
// The arithmetic in qs_core.h is written once against this abstraction:
//   LaneDev  : V = float, the real thing (HIP device code)
//   LaneEmu  : V = V4, a 4-wide value type evaluated on the host in lockstep; used ONLY by tests/ to exercise the
//              kernel arithmetic without a GPU.  The product library never instantiates it.
#pragma once
#include <math.h>
#include <stdint.h>
#include "../../include/qs_amd.h"   // QS_MAX_OBS

#if defined(__HIPCC__)
#define QS_FN __host__ __device__ __forceinline__
#define QS_DEV __device__ __forceinline__
#else
#define QS_FN inline
#endif

// ------------------------------------------------------------------ scalar (device) flavour
QS_FN float qabs(float x) { return fabsf(x); }
QS_FN float qfloor(float x) { return floorf(x); }
QS_FN float qmin(float a, float b) { return fminf(a, b); }
QS_FN float qmax(float a, float b) { return fmaxf(a, b); }
QS_FN float qsin(float x) { return sinf(x); }
QS_FN float qcos(float x) { return cosf(x); }
QS_FN float qexp(float x) { return expf(x); }
QS_FN float qlog(float x) { return logf(x); }
#if defined(__HIP_DEVICE_COMPILE__)
QS_FN float qsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }   // v_sqrt_f32, 1 ulp
QS_FN float qrcp(float x) { return __builtin_amdgcn_rcpf(x); }      // v_rcp_f32, 1 ulp
QS_FN float qrsqrt(float x) { return __builtin_amdgcn_rsqf(x); }    // v_rsq_f32, 1 ulp
#else
QS_FN float qsqrt(float x) { return sqrtf(x); }
QS_FN float qrcp(float x) { return 1.0f / x; }
QS_FN float qrsqrt(float x) { return 1.0f / sqrtf(x); }
#endif
// atan2 / asin for the Euler angles of the task and sensor epilogue: Cephes' single-precision arctangent polynomial on
// min / max of the magnitudes (|error| < 2e-7 rad), branch-free; libm's versions cost ~5x the instructions
QS_FN float qatan2(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float a = mx > 0.0f ? mn * qrcp(mx) : 0.0f;                 // in [0, 1]
    const bool mid = a > 0.4142135623730950f;                   // tan(pi/8): atan(a) = pi/4 + atan((a - 1) / (a + 1))
    const float t = mid ? (a - 1.0f) * qrcp(a + 1.0f) : a;
    const float z = t * t;
    float r = ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z) * t + t;
    r = mid ? r + 0.7853981633974483f : r;
    r = ay > ax ? 1.5707963267948966f - r : r;
    r = x < 0.0f ? 3.141592653589793f - r : r;
    return y < 0.0f ? -r : r;
}
QS_FN float qasin(float x) { return qatan2(x, qsqrt(fmaxf(1.0f - x * x, 0.0f))); }
// sin and cos of a joint angle (|x| < ~10): 2-term Cody-Waite reduction by pi/2, degree-7/8 minimax polynomials on
// [-pi/4, pi/4] (abs. error < 2e-7), a third of the instruction count of the libm pair
QS_FN void qsincos(float x, float& s, float& c) {
    float k = rintf(x * 0.636619772367581343f);
    float r = fmaf(k, -1.57079601287841796875f, x);
    r = fmaf(k, -3.139164786504813217e-7f, r);
    float r2 = r * r;
    float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f), r2 * r, r);
    float cp = fmaf(fmaf(fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f), r2, -0.5f), r2, 1.0f);
    int q = (int)k;
    float ss = (q & 1) ? cp : sp, cc = (q & 1) ? sp : cp;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
}
#if defined(__HIP_DEVICE_COMPILE__)
QS_FN float qmed3(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }   // clamp in one instruction
#else
QS_FN float qmed3(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
#endif
QS_FN float qsel(bool m, float a, float b) { return m ? a : b; }
QS_FN bool qlt(float a, float b) { return a < b; }
QS_FN bool qgt(float a, float b) { return a > b; }
QS_FN bool qle(float a, float b) { return a <= b; }
QS_FN bool qge(float a, float b) { return a >= b; }
QS_FN bool qand(bool a, bool b) { return a && b; }
QS_FN bool qor(bool a, bool b) { return a || b; }
QS_FN bool qnot(bool a) { return !a; }
QS_FN float qflag(bool m) { return m ? 1.0f : 0.0f; }

#if defined(__HIPCC__)
// The configuration as the kernels see it: the public struct, and behind it the handle's counter block -- telemetry of the rare paths
// (per HANDLE since round 4; a process-wide pair of __device__ variables before: two handles on one GPU mixed their counts).
struct QsDevCfg { qs_config cfg; unsigned long long* counters; };
enum { QS_DEVCTR_RARE_PATH = 0, QS_DEVCTR_SELF_NARROW = 1 };   // wave-substeps through the many-rows solve / whose broad phase asked for the link-link tests
struct LaneDev {
    using V = float;
    using M = bool;
    // quad_perm DPP controls: xor1 = [1,0,3,2] = 0xB1, xor2 = [2,3,0,1] = 0x4E, broadcast k = k*0x55
    template <int CTRL> static QS_DEV float dpp(float x) {
        return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
    }
    static QS_DEV float quad_sum(float x) {
        x += dpp<0xB1>(x);
        x += dpp<0x4E>(x);
        return x;
    }
    template <int K> static QS_DEV float bcast(float x) { return dpp<K * 0x55>(x); }
    // value of lane (l ^ K) of the quad, K = 1..3: quad_perm [1,0,3,2] / [2,3,0,1] / [3,2,1,0]
    template <int K> static QS_DEV float xorl(float x) { return dpp<K == 1 ? 0xB1 : (K == 2 ? 0x4E : 0x1B)>(x); }
    // acc + bcast<K>(d) * a.  (gfx950 can encode this as one v_fmac_f32_dpp, but hipcc 7.2's DPP combiner folds a v_mov_dpp into adds,
    // subtracts and max only, not into the tied-accumulator FMA: this is a v_mov_b32_dpp and a v_fmac_f32, whatever `old` value the move is
    // given -- checked in the ISA in round 3.  Writing the instruction in inline asm would need its own s_nop against the DPP read hazard,
    // which the hazard recognizer does not see inside asm.  The common-path solver therefore avoids per-row broadcast FMAs: qs_core.h, own_d.)
    template <int K> static QS_DEV float fma_bcast(float d, float a, float acc) {
        float b = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, acc), __builtin_bit_cast(int, d), K * 0x55, 0xF, 0xF, true));
        return fmaf(b, a, acc);
    }
    // (c0, c1) += (a0, a1) * b as one packed v_pk_fma_f32 (two fp32 FMAs per lane per issue slot, full rate on CDNA3/4)
    typedef float F2 __attribute__((ext_vector_type(2)));
    static QS_DEV void fma2(float a0, float a1, float b, float& c0, float& c1) {
        F2 a = {a0, a1}, c = {c0, c1}, bb = {b, b};
        c = __builtin_elementwise_fma(a, bb, c);
        c0 = c.x; c1 = c.y;
    }
    // max(m, |x0 d0|, |x1 d1|): one v_pk_mul_f32 and one v_max3_f32 with |.| source modifiers (the residual tracking of the solver sweeps)
    static QS_DEV float absmax_mul2(float m, float x0, float d0, float x1, float d1) {
        F2 x = {x0, x1}, d = {d0, d1};
        F2 p = x * d;
        return __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(p.x), __builtin_fabsf(p.y)));
    }
    // acc[K] += (a of lane K of the quad) * (b of this lane), K = 0..3: one v_mfma_f32_4x4x1_16b_f32 does it for the sixteen
    // quads of the wave (16 independent 4x4 outer products, block = quad; measured layout: tools/mfma_layout.hip).  The
    // instruction ignores EXEC, so it may only be used where the whole wave runs the same path.
    typedef float Acc4 __attribute__((ext_vector_type(4)));
    static QS_DEV Acc4 acc4_zero() { Acc4 z = {0.0f, 0.0f, 0.0f, 0.0f}; return z; }
    static QS_DEV void outer_fma(float a, float b, Acc4& acc) { acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 0); }
    // the same four products on the vector ALU (DPP broadcast + v_fma_f32): what the FULL build uses -- its register pressure spills MFMA
    // accumulators, and ROCm 7.2's "Rewrite AGPR-Copy-MFMA" pass (at work under -amdgpu-mfma-vgpr-form) segfaults on spilled ones.
    // Bitwise the MFMA's result (both are one fused multiply-add per element; tests/test_gpu_round2.py::test_results_do_not_depend_on_wave_mates
    // holds the two builds to each other).
    static QS_DEV void outer_fma_valu(float a, float b, Acc4& acc) {
        acc[0] = fmaf(bcast<0>(a), b, acc[0]); acc[1] = fmaf(bcast<1>(a), b, acc[1]);
        acc[2] = fmaf(bcast<2>(a), b, acc[2]); acc[3] = fmaf(bcast<3>(a), b, acc[3]);
    }
    template <int K> static QS_DEV float acc4_get(const Acc4& acc) { return acc[K]; }
    static QS_DEV float bcast_dyn(float x, int k) { return __shfl(x, (int)((threadIdx.x & 60u) | (unsigned)k), 64); }  // rare path only
    static QS_DEV int leg() { return (int)(threadIdx.x & 3u); }
    static QS_DEV float fx() { return (threadIdx.x & 2u) ? -1.0f : 1.0f; }  // front +, rear -
    static QS_DEV float sy() { return (threadIdx.x & 1u) ? 1.0f : -1.0f; }  // right -, left +
    static QS_DEV bool is_leg(int k) { return (int)(threadIdx.x & 3u) == k; }
    static QS_DEV bool any(bool m) { return __any(m); }                     // wave-uniform vote (16 environments)
    static QS_DEV float ld_leg(const float* rec, int base, int stride) { return rec[base + stride * (int)(threadIdx.x & 3u)]; }
    static QS_DEV void st_leg(float* rec, int base, int stride, float v) { rec[base + stride * (int)(threadIdx.x & 3u)] = v; }
    static QS_DEV float ld(const float* rec, int i) { return rec[i]; }
    static QS_DEV void st(float* rec, int i, float v) { if ((threadIdx.x & 3u) == 0) rec[i] = v; }
    static QS_DEV float first(float x) { return x; }
    static QS_DEV void opaque(float& x) { asm volatile("" : "+v"(x)); }
    // the wave's sixteen observation rows (LDS) from the row of this lane's environment
    static QS_DEV float* wave_scratch(float* row) { return row - (size_t)(threadIdx.x >> 2) * QS_MAX_OBS; }
    // (every qs_config the device code gets is the first member of a QsDevCfg in device memory: qs_hip.hip, create_impl)
    static QS_DEV void count_rare_path(const qs_config& cfg) { if (threadIdx.x == 0) atomicAdd(&reinterpret_cast<const QsDevCfg&>(cfg).counters[QS_DEVCTR_RARE_PATH], 1ull); }
    static QS_DEV void count_self_narrow(const qs_config& cfg) { if (threadIdx.x == 0) atomicAdd(&reinterpret_cast<const QsDevCfg&>(cfg).counters[QS_DEVCTR_SELF_NARROW], 1ull); }
    // orders LDS traffic between the lanes of a wave (in-order LDS queue per wave; this only pins the compiler)
    static QS_DEV void sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
};
#endif

// ------------------------------------------------------------------ 4-wide host emulation (tests only)
#if !defined(__HIP_DEVICE_COMPILE__)
struct V4 {
    float v[4];
    V4() : v{0, 0, 0, 0} {}
    V4(float a) : v{a, a, a, a} {}
    V4(float a, float b, float c, float d) : v{a, b, c, d} {}
};
struct M4 { bool v[4]; };
#define QS_V4_BIN(op)                                                                                          \
    inline V4 operator op(V4 a, V4 b) { V4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] op b.v[i]; return r; } \
    inline V4 operator op(V4 a, float b) { V4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] op b; return r; }   \
    inline V4 operator op(float a, V4 b) { V4 r; for (int i = 0; i < 4; i++) r.v[i] = a op b.v[i]; return r; }
QS_V4_BIN(+) QS_V4_BIN(-) QS_V4_BIN(*) QS_V4_BIN(/)
inline V4 operator-(V4 a) { V4 r; for (int i = 0; i < 4; i++) r.v[i] = -a.v[i]; return r; }
inline V4& operator+=(V4& a, V4 b) { a = a + b; return a; }
inline V4& operator-=(V4& a, V4 b) { a = a - b; return a; }
inline V4& operator*=(V4& a, V4 b) { a = a * b; return a; }
#define QS_V4_FN1(name, f) inline V4 name(V4 a) { V4 r; for (int i = 0; i < 4; i++) r.v[i] = f(a.v[i]); return r; }
QS_V4_FN1(qsqrt, sqrtf) QS_V4_FN1(qabs, fabsf) QS_V4_FN1(qfloor, floorf) QS_V4_FN1(qsin, sinf) QS_V4_FN1(qcos, cosf) QS_V4_FN1(qasin, asinf)
QS_V4_FN1(qexp, expf) QS_V4_FN1(qlog, logf) QS_V4_FN1(qrcp, qrcp) QS_V4_FN1(qrsqrt, qrsqrt)
inline void qsincos(V4 x, V4& s, V4& c) { for (int i = 0; i < 4; i++) qsincos(x.v[i], s.v[i], c.v[i]); }
inline V4 qmin(V4 a, V4 b) { V4 r; for (int i = 0; i < 4; i++) r.v[i] = fminf(a.v[i], b.v[i]); return r; }
inline V4 qmax(V4 a, V4 b) { V4 r; for (int i = 0; i < 4; i++) r.v[i] = fmaxf(a.v[i], b.v[i]); return r; }
inline V4 qatan2(V4 a, V4 b) { V4 r; for (int i = 0; i < 4; i++) r.v[i] = atan2f(a.v[i], b.v[i]); return r; }
inline V4 qmed3(V4 x, V4 lo, V4 hi) { return qmin(qmax(x, lo), hi); }
inline V4 qsel(M4 m, V4 a, V4 b) { V4 r; for (int i = 0; i < 4; i++) r.v[i] = m.v[i] ? a.v[i] : b.v[i]; return r; }
#define QS_V4_CMP(name, op) inline M4 name(V4 a, V4 b) { M4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] op b.v[i]; return r; }
QS_V4_CMP(qlt, <) QS_V4_CMP(qgt, >) QS_V4_CMP(qle, <=) QS_V4_CMP(qge, >=)
inline M4 qand(M4 a, M4 b) { M4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] && b.v[i]; return r; }
inline M4 qor(M4 a, M4 b) { M4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] || b.v[i]; return r; }
inline M4 qnot(M4 a) { M4 r; for (int i = 0; i < 4; i++) r.v[i] = !a.v[i]; return r; }
inline V4 qflag(M4 m) { V4 r; for (int i = 0; i < 4; i++) r.v[i] = m.v[i] ? 1.0f : 0.0f; return r; }

struct LaneEmu {
    using V = V4;
    using M = M4;
    static V4 quad_sum(V4 x) {  // same association as the DPP butterfly: (l ^ 1) first, then (l ^ 2)
        V4 t, r;
        for (int i = 0; i < 4; i++) t.v[i] = x.v[i] + x.v[i ^ 1];
        for (int i = 0; i < 4; i++) r.v[i] = t.v[i] + t.v[i ^ 2];
        return r;
    }
    template <int K> static V4 bcast(V4 x) { return V4(x.v[K]); }
    template <int K> static V4 xorl(V4 x) { V4 r; for (int i = 0; i < 4; i++) r.v[i] = x.v[i ^ K]; return r; }
    static V4 bcast_dyn(V4 x, int k) { return V4(x.v[k]); }
    template <int K> static V4 fma_bcast(V4 d, V4 a, V4 acc) { V4 r; for (int i = 0; i < 4; i++) r.v[i] = acc.v[i] + d.v[K] * a.v[i]; return r; }
    static V4 fx() { return V4(1, 1, -1, -1); }
    static V4 sy() { return V4(-1, 1, -1, 1); }
    static M4 is_leg(int k) { M4 m; for (int i = 0; i < 4; i++) m.v[i] = (i == k); return m; }
    static V4 absmax_mul2(V4 m, V4 x0, V4 d0, V4 x1, V4 d1) {
        V4 r;
        for (int l = 0; l < 4; l++) r.v[l] = fmaxf(m.v[l], fmaxf(fabsf(x0.v[l] * d0.v[l]), fabsf(x1.v[l] * d1.v[l])));
        return r;
    }
    static void fma2(V4 a0, V4 a1, V4 b, V4& c0, V4& c1) { for (int l = 0; l < 4; l++) { c0.v[l] = fmaf(a0.v[l], b.v[l], c0.v[l]); c1.v[l] = fmaf(a1.v[l], b.v[l], c1.v[l]); } }
    static float* wave_scratch(float* row) { return row; }
    static void count_rare_path(const qs_config&) {}
    static void count_self_narrow(const qs_config&) {}
    struct Acc4 { V4 k[4]; };
    static Acc4 acc4_zero() { Acc4 z; for (int i = 0; i < 4; i++) z.k[i] = V4(0.0f); return z; }
    static void outer_fma(V4 a, V4 b, Acc4& acc) { for (int K = 0; K < 4; K++) for (int l = 0; l < 4; l++) acc.k[K].v[l] = fmaf(a.v[K], b.v[l], acc.k[K].v[l]); }
    static void outer_fma_valu(V4 a, V4 b, Acc4& acc) { outer_fma(a, b, acc); }
    template <int K> static V4 acc4_get(const Acc4& acc) { return acc.k[K]; }
    static bool any(M4 m) { return m.v[0] || m.v[1] || m.v[2] || m.v[3]; }
    static V4 ld_leg(const float* rec, int base, int stride) { V4 r; for (int i = 0; i < 4; i++) r.v[i] = rec[base + stride * i]; return r; }
    static void st_leg(float* rec, int base, int stride, V4 v) { for (int i = 0; i < 4; i++) rec[base + stride * i] = v.v[i]; }
    static V4 ld(const float* rec, int i) { return V4(rec[i]); }
    static void st(float* rec, int i, V4 v) { rec[i] = v.v[0]; }
    static float first(V4 x) { return x.v[0]; }
    static void opaque(V4&) {}
    static void sync() {}
};
#endif
