// What the first touch of a wave's records costs at kernel entry: 512 (or more) one-wave workgroups, each loading 16 records' leading 704 B
// (11 x global_load_dwordx4 per lane, all issued together) from a 9 MB array of 1152-B records that the PREVIOUS kernel wrote (as k_step's
// tile_store does) or that nobody wrote since the last read; cycles from the first issue to the last arrival, workgroup 0 and the average.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/entry_latency tools/entry_latency.hip && tools/bin/entry_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REC 288
__global__ __launch_bounds__(64) void k_write(float* recs, float v) {
    float4* dst = reinterpret_cast<float4*>(recs + (size_t)blockIdx.x * 16 * REC);
    for (int i = threadIdx.x; i < 16 * 38; i += 64) { int e = i / 38, o = i - e * 38 + 6; dst[e * (REC / 4) + o] = make_float4(v, v, v, v); }
}
__global__ __launch_bounds__(64) void k_read(const float* __restrict__ recs, unsigned long long* cycles, float* sink, int serial) {
    const float4* src = reinterpret_cast<const float4*>(recs + (size_t)blockIdx.x * 16 * REC);
    unsigned long long t0 = __builtin_readcyclecounter();
    float4 v[11];
    float acc = 0.0f;
    if (!serial) {
#pragma unroll
        for (int r = 0; r < 11; r++) { int i = threadIdx.x + 64 * r, e = i / 44, o = i - e * 44; v[r] = src[e * (REC / 4) + o]; }
#pragma unroll
        for (int r = 0; r < 11; r++) acc += v[r].x + v[r].y + v[r].z + v[r].w;
    } else {
        for (int r = 0; r < 11; r++) {
            int i = threadIdx.x + 64 * r, e = i / 44, o = i - e * 44;
            float4 w = src[e * (REC / 4) + o];
            acc += w.x + w.y + w.z + w.w;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc == 12345.678f) sink[0] = acc;
}
int main(int argc, char** argv) {
    const int waves = argc > 1 ? atoi(argv[1]) : 512, reps = 200;
    float* recs; unsigned long long* cyc; float* sink;
    hipMalloc(&recs, (size_t)waves * 16 * REC * 4); hipMalloc(&cyc, waves * 8); hipMalloc(&sink, 4);
    hipMemset(recs, 0, (size_t)waves * 16 * REC * 4);
    std::vector<unsigned long long> h(waves);
    for (int mode = 0; mode < 4; mode++) {   // bit 0: a writer kernel in front of every read; bit 1: one load at a time
        double s0 = 0, sall = 0, smax = 0;
        for (int rep = 0; rep < reps; rep++) {
            if (mode & 1) hipLaunchKernelGGL(k_write, dim3(waves), dim3(64), 0, 0, recs, (float)rep);
            hipLaunchKernelGGL(k_read, dim3(waves), dim3(64), 0, 0, recs, cyc, sink, mode >> 1);
            hipMemcpy(h.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
            double a = 0, m = 0; for (auto c : h) { a += (double)c; if ((double)c > m) m = (double)c; }
            s0 += (double)h[0]; sall += a / waves; smax += m;
        }
        printf("%s, %s: cycles from first issue to last arrival: workgroup 0 %.0f, average over %d workgroups %.0f, slowest %.0f\n",
               (mode & 1) ? "records written by the kernel before" : "records untouched since the last read", (mode >> 1) ? "one load at a time" : "11 loads in flight",
               s0 / reps, waves, sall / reps, smax / reps);
    }
    return 0;
}
