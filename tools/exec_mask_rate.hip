// exec_mask_rate.hip -- two questions about one wave alone on a SIMD (the situation of k_step at N = 8192):
//  (1) does a wave64 VALU instruction get cheaper when part of the wave is masked off?  (It would make waves of 4 or 8 environments
//      attractive, half of the SIMDs idle at N = 8192.)   Answer on MI355X: no.
//  (2) what does an instruction cost in a dependent chain, and with 2 / 4 / 8 independent chains to interleave (scalar and packed fp32)?
// One wave per launch, lanes >= `active` leave the kernel first; s_memtime around the loop.
//   hipcc --offload-arch=gfx950 -O3 tools/exec_mask_rate.hip -o tools/bin/exec_mask_rate && tools/bin/exec_mask_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int CH> __global__ void chains(float* out, long long* cycles, int active, int iters) {
    if ((int)threadIdx.x >= active) return;
    float a[CH];
    for (int c = 0; c < CH; c++) a[c] = threadIdx.x * 1e-3f + c;
    float b = 1.0001f, d = 1e-7f;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 32 / CH; k++)
#pragma unroll
            for (int c = 0; c < CH; c++) a[c] = __builtin_fmaf(a[c], b, d);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int c = 0; c < CH; c++) s += a[c];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) *cycles = t1 - t0;
}
template <int CH> __global__ void chains_pk(float* out, long long* cycles, int active, int iters) {
    if ((int)threadIdx.x >= active) return;
    float2v a[CH];
    for (int c = 0; c < CH; c++) { a[c].x = threadIdx.x * 1e-3f + c; a[c].y = a[c].x + 0.5f; }
    float2v b = {1.0001f, 1.0002f}, d = {1e-7f, 2e-7f};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 32 / CH; k++)
#pragma unroll
            for (int c = 0; c < CH; c++) a[c] = __builtin_elementwise_fma(a[c], b, d);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int c = 0; c < CH; c++) s += a[c].x + a[c].y;
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) *cycles = t1 - t0;
}
template <class K> static double run(K kernel, float* out, long long* cyc, int active, int iters) {
    long long c = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(kernel, dim3(1), dim3(64), 0, 0, out, cyc, active, iters);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    }
    return (double)c / ((double)iters * 32);
}
int main() {
    float* out; long long* cyc; (void)hipMalloc(&out, 256); (void)hipMalloc(&cyc, 8);
    const int iters = 20000;
    for (int active : {64, 32, 16, 8, 4}) printf("one dependent chain, %2d active lanes: %.2f s_memtime counts per v_fma_f32\n", active, run(chains<1>, out, cyc, active, iters));
    printf("independent chains (64 lanes), counts per instruction: v_fma_f32  1: %.2f  2: %.2f  4: %.2f  8: %.2f\n", run(chains<1>, out, cyc, 64, iters),
           run(chains<2>, out, cyc, 64, iters), run(chains<4>, out, cyc, 64, iters), run(chains<8>, out, cyc, 64, iters));
    printf("independent chains (64 lanes), counts per instruction: v_pk_fma_f32 1: %.2f  2: %.2f  4: %.2f  8: %.2f\n", run(chains_pk<1>, out, cyc, 64, iters),
           run(chains_pk<2>, out, cyc, 64, iters), run(chains_pk<4>, out, cyc, 64, iters), run(chains_pk<8>, out, cyc, 64, iters));
    return 0;
}
