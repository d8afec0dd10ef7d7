// Issue cost of v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 against v_fma_f32 for ONE wave per SIMD (the regime of k_step).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float F2 __attribute__((ext_vector_type(2)));
#define REP 512
__global__ void k(float* out, unsigned long long* cyc, float s) {
    F2 a[8]; float b[16];
    for (int i = 0; i < 8; i++) { a[i].x = out[threadIdx.x + i]; a[i].y = out[threadIdx.x + 8 + i]; }
    for (int i = 0; i < 16; i++) b[i] = out[threadIdx.x + 16 + i];
    F2 m = {s, s * 0.5f}, c = {0.25f, 0.125f};
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = __builtin_elementwise_fma(a[i], m, c);   // 8 independent packed FMAs
    }
    unsigned long long t1 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) b[i] = fmaf(b[i], s, 0.25f);                  // 16 independent scalar FMAs = the same flops
    }
    unsigned long long t2 = __builtin_readcyclecounter();
    float acc = 0; for (int i = 0; i < 8; i++) acc += a[i].x + a[i].y; for (int i = 0; i < 16; i++) acc += b[i];
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}
int main() {
    float* o; unsigned long long* c; unsigned long long h[2];
    hipMalloc(&o, 4096); hipMemset(o, 0, 4096); hipMalloc(&c, 16);
    for (int it = 0; it < 2; it++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, c, 0.999f); hipMemcpy(h, c, 16, hipMemcpyDeviceToHost); }
    printf("8 v_pk_fma_f32 per iteration: %.2f cycles each;  16 v_fma_f32 per iteration: %.2f cycles each\n", h[0] / (8.0 * REP), h[1] / (16.0 * REP));
    return 0;
}
