#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* o) {
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
    for (int i = 0; i < 4; i++) o[threadIdx.x * 4 + i] = c[i];
}
int main() {
    float ha[64], hb[64], ho[256];
    for (int l = 0; l < 64; l++) { ha[l] = 1 + (l % 4) + 10 * (l / 4); hb[l] = 100 * (1 + l % 4) + 1000 * (l / 4); }
    float *a, *b, *o;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&o, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, o);
    hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
    // hypothesis: lane l (block l/4, j = l%4) register i holds a[block, i] * b[block, j]
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int i = 0; i < 4; i++) {
        float exp = ha[(l / 4) * 4 + i] * hb[l];
        if (ho[l * 4 + i] != exp) { if (bad < 8) printf("lane %d reg %d: got %g expected %g\n", l, i, ho[l * 4 + i], exp); bad++; }
    }
    printf("mismatches under hypothesis D[i][j]: lane=4*blk+j, reg=i : %d\n", bad);
    for (int l = 0; l < 8; l++) printf("lane %d: %g %g %g %g\n", l, ho[l*4], ho[l*4+1], ho[l*4+2], ho[l*4+3]);
    return 0;
}
