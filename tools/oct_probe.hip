// oct_probe.hip -- measurement behind DESIGN.md 10a: what would two lanes per leg (8 lanes per environment, 1024 waves at N = 8192)
// buy?  The per-leg dynamics phases of the substep that do not depend on the solver -- link inertias about the base origin, the RNEA
// bias, the CRBA columns F_j = Ic_j S_j with their projections -- written twice against the same small algebra as csrc/qs_core.h:
//   quad : one lane per leg (the product's mapping), the three links one after the other, the base term replicated over the quad
//   oct  : two lanes per leg.  Lane h = 0 carries hip + calf/foot, lane h = 1 thigh + the base term (whole on one lane of the eight,
//          zero on the others); both run the SAME instruction stream over their own two bodies and exchange partial sums with the pair
//          partner (quad_perm xor 1), leg sums run over the eight lanes of the environment (xor 2, then row_half_mirror).
// Both kernels run ITER dependent iterations per wave (the joint angles of the next iteration depend on the previous result, so
// nothing is hoisted) and report s_memtime cycles per iteration of wave 0; results are compared lane for lane.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -ffinite-math-only -fno-signed-zeros -fno-trapping-math \
//         -mllvm -amdgpu-sched-strategy=iterative-ilp -I include -o oct_probe tools/oct_probe.hip && ./oct_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../quadruped-springs_amd/csrc/qs_core.h"

using namespace qs;
using V = float;
typedef V3<V> V3f;
typedef Sp<V> Spf;
typedef SI<V> SIf;

#define ITER 64

template <int CTRL> __device__ __forceinline__ float dppq(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float half_mirror(float x) {   // lane i <-> 7 - i of each group of eight (DPP row_half_mirror = 0x141)
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
}

struct LinkConst { float m; float c[3]; float I[6]; };     // mass, centre of mass, inertia about it (link axes)
__device__ __forceinline__ SIf link_inertia(const LinkConst& L, V3f p, V3f X, V3f Y, V3f Z) {
    S3<V> Il; Il.xx = L.I[0]; Il.xy = L.I[1]; Il.xz = L.I[2]; Il.yy = L.I[3]; Il.yz = L.I[4]; Il.zz = L.I[5];
    return part_inertia<V>(L.m, mk3<V>(L.c[0], L.c[1], L.c[2]), Il, p, X, Y, Z);
}
__device__ __forceinline__ Spf body_force(const SIf& I, const Spf& a, const Spf& v) { return apply(I, a) + crf(v, apply(I, v)); }

struct Frames { V3f p1, p2, p3, ax1, Y, Z1, X2, Z2, X3, Z3, rf; };
__device__ __forceinline__ Frames kinematics(float fx, float sy, const float* q) {
    using namespace go1;
    Frames f;
    float s1, c1, s2, c2, s23, c23;
    qsincos(q[0], s1, c1); qsincos(q[1], s2, c2); qsincos(q[1] + q[2], s23, c23);
    f.p1 = mk3<V>(fx * HIP_X, sy * HIP_Y, 0.0f);
    f.ax1 = mk3<V>(1.0f, 0.0f, 0.0f);
    f.Y = mk3<V>(0.0f, c1, s1);
    f.Z1 = mk3<V>(0.0f, -s1, c1);
    f.p2 = f.p1 + f.Y * (sy * THIGH_Y);
    f.X2 = mk3<V>(c2, s1 * s2, -c1 * s2); f.Z2 = mk3<V>(s2, -s1 * c2, c1 * c2);
    f.p3 = f.p2 + f.Z2 * LEG_Z;
    f.X3 = mk3<V>(c23, s1 * s23, -c1 * s23); f.Z3 = mk3<V>(s23, -s1 * c23, c1 * c23);
    f.rf = f.p3 + f.Z3 * LEG_Z;
    return f;
}

struct Result { float C[3], Cb[6], D[6]; };   // joint bias, base bias (summed over the legs), D = S^T Ic S (6 unique)

__device__ __forceinline__ LinkConst hip_c(float fx, float sy) {
    using namespace go1;
    LinkConst L; L.m = HIP_M; L.c[0] = fx * (-HIP_C[0]); L.c[1] = sy * (-HIP_C[1]); L.c[2] = HIP_C[2];
    L.I[0] = HIP_I[0]; L.I[1] = fx * sy * HIP_I[1]; L.I[2] = -fx * HIP_I[2]; L.I[3] = HIP_I[3]; L.I[4] = -sy * HIP_I[4]; L.I[5] = HIP_I[5];
    return L;
}
__device__ __forceinline__ LinkConst thigh_c(float sy) {
    using namespace go1;
    LinkConst L; L.m = THIGH_M; L.c[0] = THIGH_C[0]; L.c[1] = -sy * THIGH_C[1]; L.c[2] = THIGH_C[2];
    L.I[0] = THIGH_I[0]; L.I[1] = sy * THIGH_I[1]; L.I[2] = THIGH_I[2]; L.I[3] = THIGH_I[3]; L.I[4] = sy * THIGH_I[4]; L.I[5] = THIGH_I[5];
    return L;
}
__device__ __forceinline__ LinkConst calf_c() {
    using namespace go1;
    LinkConst L; L.m = CALF_M; for (int i = 0; i < 3; i++) L.c[i] = CALF_C[i]; for (int i = 0; i < 6; i++) L.I[i] = CALF_I[i];
    return L;
}

// ------------------------------------------------------------------ quad: one lane per leg
__global__ __launch_bounds__(64, 1) void k_quad(const float* __restrict__ in, Result* __restrict__ out, unsigned long long* cycles) {
    using namespace go1;
    const int lane = threadIdx.x, leg = lane & 3;
    const float fx = (leg & 2) ? -1.0f : 1.0f, sy = (leg & 1) ? 1.0f : -1.0f;
    const float* x = in + (size_t)(blockIdx.x * 64 + lane) * 16;
    float q[3] = {x[0], x[1], x[2]}, qd[3] = {x[3], x[4], x[5]};
    Spf v0; v0.a = mk3<V>(x[6], x[7], x[8]); v0.l = mk3<V>(x[9], x[10], x[11]);
    V3f g = mk3<V>(x[12], x[13], x[14]);
    const LinkConst Lh = hip_c(fx, sy), Lt = thigh_c(sy), Lc = calf_c();
    SIf I0 = point_inertia<V>(TRUNK_M, 0.05f, mk3<V>(TRUNK_CX, 0.0f, TRUNK_CZ));
    Result r;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; it++) {
        Frames f = kinematics(fx, sy, q);
        SIf I1 = link_inertia(Lh, f.p1, f.ax1, f.Y, f.Z1), I2 = link_inertia(Lt, f.p2, f.X2, f.Y, f.Z2);
        SIf I3 = link_inertia(Lc, f.p3, f.X3, f.Y, f.Z3) + point_inertia<V>(FOOT_M, FOOT_I, f.rf);
        Spf S1, S2, S3; S1.a = f.ax1; S1.l = cross(f.p1, f.ax1); S2.a = f.Y; S2.l = cross(f.p2, f.Y); S3.a = f.Y; S3.l = cross(f.p3, f.Y);
        Spf a0; a0.a = mk3<V>(0.0f, 0.0f, 0.0f); a0.l = g * 9.8f;
        Spf vj1, vj2, vj3; vj1.a = S1.a * qd[0]; vj1.l = S1.l * qd[0]; vj2.a = S2.a * qd[1]; vj2.l = S2.l * qd[1]; vj3.a = S3.a * qd[2]; vj3.l = S3.l * qd[2];
        Spf v1 = v0 + vj1, v2 = v1 + vj2, v3 = v2 + vj3;
        Spf a1 = a0 + crm(v0, vj1), a2 = a1 + crm(v1, vj2), a3 = a2 + crm(v2, vj3);
        Spf f1 = body_force(I1, a1, v1), f2 = body_force(I2, a2, v2), f3 = body_force(I3, a3, v3);
        Spf fs2 = f2 + f3, fs1 = f1 + fs2;
        r.C[0] = dot(S1, fs1); r.C[1] = dot(S2, fs2); r.C[2] = dot(S3, f3);
        Spf f0 = body_force(I0, a0, v0);
        r.Cb[0] = LaneDev::quad_sum(fs1.a.x) + f0.a.x; r.Cb[1] = LaneDev::quad_sum(fs1.a.y) + f0.a.y; r.Cb[2] = LaneDev::quad_sum(fs1.a.z) + f0.a.z;
        r.Cb[3] = LaneDev::quad_sum(fs1.l.x) + f0.l.x; r.Cb[4] = LaneDev::quad_sum(fs1.l.y) + f0.l.y; r.Cb[5] = LaneDev::quad_sum(fs1.l.z) + f0.l.z;
        SIf Ic2 = I2 + I3, Ic1 = I1 + Ic2;
        Spf F1 = apply(Ic1, S1), F2 = apply(Ic2, S2), F3 = apply(I3, S3);
        r.D[0] = dot(S1, F1); r.D[1] = dot(S1, F2); r.D[2] = dot(S1, F3); r.D[3] = dot(S2, F2); r.D[4] = dot(S2, F3); r.D[5] = dot(S3, F3);
        // feed back: the next iteration's state depends on this one's result
        for (int j = 0; j < 3; j++) { qd[j] += 1e-4f * (r.C[j] + r.Cb[j]) * (1.0f / (1.0f + r.D[3])); q[j] += 1e-3f * qd[j]; }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + lane] = r;
    if (blockIdx.x == 0 && lane == 0) cycles[0] = (t1 - t0) / ITER;
}

// ------------------------------------------------------------------ oct: two lanes per leg
// lane = 8 env + 2 leg + h ; pair partner = lane ^ 1 ; legs 0, 1 in the first quad, 2, 3 in the second
__device__ __forceinline__ float pair_sum(float x) { return x + dppq<0xB1>(x); }
__device__ __forceinline__ float leg_sum8(float x) {     // sum over the four legs of a pair-replicated value
    x += dppq<0x4E>(x);          // legs 0 + 1 (or 2 + 3)
    return x + half_mirror(x);   // + the other quad's (lane i <-> 7 - i: same sum, mirrored)
}
__device__ __forceinline__ Spf pair_sum(Spf f) {
    f.a.x = pair_sum(f.a.x); f.a.y = pair_sum(f.a.y); f.a.z = pair_sum(f.a.z); f.l.x = pair_sum(f.l.x); f.l.y = pair_sum(f.l.y); f.l.z = pair_sum(f.l.z);
    return f;
}
__device__ __forceinline__ float sel(bool h, float a, float b) { return h ? a : b; }
__device__ __forceinline__ V3f sel(bool h, V3f a, V3f b) { return mk3<V>(h ? a.x : b.x, h ? a.y : b.y, h ? a.z : b.z); }
__device__ __forceinline__ Spf sel(bool h, Spf a, Spf b) { Spf r; r.a = sel(h, a.a, b.a); r.l = sel(h, a.l, b.l); return r; }
__device__ __forceinline__ Spf keep(Spf f, bool k) { Spf z; z.a = mk3<V>(0.0f, 0.0f, 0.0f); z.l = z.a; return sel(k, f, z); }

__global__ __launch_bounds__(64, 1) void k_oct(const float* __restrict__ in, Result* __restrict__ out, unsigned long long* cycles) {
    using namespace go1;
    const int lane = threadIdx.x, leg = (lane >> 1) & 3;
    const bool h1 = lane & 1;                                   // h = 1: thigh + the base share ; h = 0: hip + calf/foot
    const float fx = (leg & 2) ? -1.0f : 1.0f, sy = (leg & 1) ? 1.0f : -1.0f;
    // the same inputs as the quad kernel's lane (env, leg): element index = (block * 64 + lane) / 2 within a twice as large grid
    const int env = (blockIdx.x * 64 + lane) >> 3;
    const float* x = in + (size_t)(env * 4 + leg) * 16;
    float q[3] = {x[0], x[1], x[2]}, qd[3] = {x[3], x[4], x[5]};
    Spf v0; v0.a = mk3<V>(x[6], x[7], x[8]); v0.l = mk3<V>(x[9], x[10], x[11]);
    V3f g = mk3<V>(x[12], x[13], x[14]);
    // body A of this lane: hip (h = 0) or thigh (h = 1); body B: calf (+ foot) or the base term (one lane of the eight carries it whole:
    // leg 0's h = 1 lane; the other h = 1 lanes carry a zero inertia through the same instructions)
    const LinkConst La = h1 ? thigh_c(sy) : hip_c(fx, sy);
    const LinkConst Lb = calf_c();
    const bool base_lane = h1 && leg == 0;
    SIf I0 = point_inertia<V>(base_lane ? TRUNK_M : 0.0f, base_lane ? 0.05f : 0.0f, mk3<V>(TRUNK_CX, 0.0f, TRUNK_CZ));
    Result r;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; it++) {
        Frames f = kinematics(fx, sy, q);                       // replicated over the pair
        Spf S1, S2, S3; S1.a = f.ax1; S1.l = cross(f.p1, f.ax1); S2.a = f.Y; S2.l = cross(f.p2, f.Y); S3.a = f.Y; S3.l = cross(f.p3, f.Y);
        Spf a0; a0.a = mk3<V>(0.0f, 0.0f, 0.0f); a0.l = g * 9.8f;
        Spf vj1, vj2, vj3; vj1.a = S1.a * qd[0]; vj1.l = S1.l * qd[0]; vj2.a = S2.a * qd[1]; vj2.l = S2.l * qd[1]; vj3.a = S3.a * qd[2]; vj3.l = S3.l * qd[2];
        Spf v1 = v0 + vj1, v2 = v1 + vj2, v3 = v2 + vj3;          // replicated (cheap: the propagation is 100 of the 480 instructions)
        Spf a1 = a0 + crm(v0, vj1), a2 = a1 + crm(v1, vj2), a3 = a2 + crm(v2, vj3);
        // own bodies: A = hip | thigh, B = calf + foot | base
        SIf IA = link_inertia(La, sel(h1, f.p2, f.p1), sel(h1, f.X2, f.ax1), f.Y, sel(h1, f.Z2, f.Z1));
        SIf IB = h1 ? I0 : link_inertia(Lb, f.p3, f.X3, f.Y, f.Z3) + point_inertia<V>(FOOT_M, FOOT_I, f.rf);
        Spf fA = body_force(IA, sel(h1, a2, a1), sel(h1, v2, v1));
        Spf fB = body_force(IB, sel(h1, a0, a3), sel(h1, v0, v3));
        // h = 0 holds f1 (A) and f3 (B) ; h = 1 holds f2 (A) and the base term (B, non-zero on one lane of the eight)
        Spf f1p = keep(fA, !h1), f2p = keep(fA, h1), f3p = keep(fB, !h1), f0p = keep(fB, h1);
        Spf f3 = pair_sum(f3p), fs2 = pair_sum(f2p + f3p), fs1 = pair_sum(f1p + f2p + f3p);
        r.C[0] = dot(S1, fs1); r.C[1] = dot(S2, fs2); r.C[2] = dot(S3, f3);
        // base bias = sum over the eight lanes of (fs1 counted once per leg: on h = 0) + (the base term)
        Spf t = sel(h1, f0p, fs1);
        r.Cb[0] = leg_sum8(pair_sum(t.a.x)); r.Cb[1] = leg_sum8(pair_sum(t.a.y)); r.Cb[2] = leg_sum8(pair_sum(t.a.z));
        r.Cb[3] = leg_sum8(pair_sum(t.l.x)); r.Cb[4] = leg_sum8(pair_sum(t.l.y)); r.Cb[5] = leg_sum8(pair_sum(t.l.z));
        // CRBA columns: F1 = (I1 + I2 + I3) S1, F2 = (I2 + I3) S2, F3 = I3 S3 from the pair's partial inertias
        SIf IBl = IB; if (h1) { IBl.m = 0; IBl.h = mk3<V>(0, 0, 0); IBl.I.xx = IBl.I.xy = IBl.I.xz = IBl.I.yy = IBl.I.yz = IBl.I.zz = 0; }   // the base is no leg link
        SIf Ip1 = IA + IBl;                     // h0: I1 + I3 ; h1: I2
        SIf Ip2 = h1 ? IA : IBl;                // h0: I3      ; h1: I2
        Spf F1 = pair_sum(apply(Ip1, S1)), F2 = pair_sum(apply(Ip2, S2)), F3 = pair_sum(apply(IBl, S3));
        r.D[0] = dot(S1, F1); r.D[1] = dot(S1, F2); r.D[2] = dot(S1, F3); r.D[3] = dot(S2, F2); r.D[4] = dot(S2, F3); r.D[5] = dot(S3, F3);
        for (int j = 0; j < 3; j++) { qd[j] += 1e-4f * (r.C[j] + r.Cb[j]) * (1.0f / (1.0f + r.D[3])); q[j] += 1e-3f * qd[j]; }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + lane] = r;
    if (blockIdx.x == 0 && lane == 0) cycles[0] = (t1 - t0) / ITER;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const int n_env = 8192, n_leg = n_env * 4;
    std::vector<float> in((size_t)n_leg * 16);
    srand(1);
    auto u = [] { return (float)rand() / RAND_MAX * 2.0f - 1.0f; };
    for (int e = 0; e < n_env; e++) {
        float base[9]; for (int k = 0; k < 9; k++) base[k] = u();
        float gn = sqrtf(base[6] * base[6] + base[7] * base[7] + base[8] * base[8]);
        for (int L = 0; L < 4; L++) {
            float* x = &in[(size_t)(e * 4 + L) * 16];
            x[0] = 0.3f * u(); x[1] = 0.8f + 0.4f * u(); x[2] = -1.6f + 0.5f * u();
            for (int k = 0; k < 3; k++) x[3 + k] = 3.0f * u();
            for (int k = 0; k < 6; k++) x[6 + k] = base[k];               // base twist: the same for the four legs of an environment
            for (int k = 0; k < 3; k++) x[12 + k] = base[6 + k] / gn;
        }
    }
    float* d_in; Result *d_q, *d_o; unsigned long long* d_c;
    CK(hipMalloc(&d_in, in.size() * 4)); CK(hipMalloc(&d_q, (size_t)n_leg * sizeof(Result))); CK(hipMalloc(&d_o, (size_t)n_leg * 2 * sizeof(Result)));
    CK(hipMalloc(&d_c, 16));
    CK(hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long cq = 0, co = 0; float msq = 0, mso = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_quad, dim3(n_leg / 64), dim3(64), 0, 0, d_in, d_q, d_c); hipEventRecord(e1);
        CK(hipDeviceSynchronize()); hipEventElapsedTime(&msq, e0, e1); CK(hipMemcpy(&cq, d_c, 8, hipMemcpyDeviceToHost));
        hipEventRecord(e0); hipLaunchKernelGGL(k_oct, dim3(n_leg * 2 / 64), dim3(64), 0, 0, d_in, d_o, d_c); hipEventRecord(e1);
        CK(hipDeviceSynchronize()); hipEventElapsedTime(&mso, e0, e1); CK(hipMemcpy(&co, d_c, 8, hipMemcpyDeviceToHost));
    }
    std::vector<Result> rq(n_leg), ro((size_t)n_leg * 2);
    CK(hipMemcpy(rq.data(), d_q, rq.size() * sizeof(Result), hipMemcpyDeviceToHost));
    CK(hipMemcpy(ro.data(), d_o, ro.size() * sizeof(Result), hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    for (int e = 0; e < n_env; e++)
        for (int L = 0; L < 4; L++)
            for (int h = 0; h < 2; h++) {
                const Result& a = rq[e * 4 + L]; const Result& b = ro[(size_t)e * 8 + 2 * L + h];
                const float* pa = (const float*)&a; const float* pb = (const float*)&b;
                for (int k = 0; k < 15; k++) { worst = fmax(worst, fabs((double)pa[k] - pb[k])); scale = fmax(scale, fabs((double)pa[k])); }
            }
    printf("{\"n_env\": %d, \"iterations\": %d, \"quad\": {\"waves\": %d, \"cycles_per_iteration\": %llu, \"kernel_ms\": %.4f}, "
           "\"oct\": {\"waves\": %d, \"cycles_per_iteration\": %llu, \"kernel_ms\": %.4f}, \"max_abs_difference\": %.3e, \"max_abs_value\": %.3e}\n",
           n_env, ITER, n_leg / 64, cq, msq, n_leg * 2 / 64, co, mso, worst, scale);
    return worst < 1e-2 * fmax(1.0, scale) ? 0 : 2;
}
