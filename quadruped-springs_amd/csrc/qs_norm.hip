// SB3 VecNormalize on device: running mean / variance of observations and discounted returns, observation and reward
// normalisation with clipping, terminal observations, returns reset at episode ends.
//
// Follows stable_baselines3 == 1.5.1a7 (pinned by the reference's setup.py:11; absent from /root/reference and from this image):
// common/running_mean_std.py (RunningMeanStd.update / update_from_moments, parallel-variance merge) and
// common/vec_env/vec_normalize.py (VecNormalize.step_wait, _update_reward, normalize_obs, normalize_reward, reset).  The
// reference applies it at load_model.py:109-137 and get_demonstrations.py:71 (VecNormalize.load(stats, env); training = False;
// norm_reward = False).  Statistics are float64 like SB3's; the batch moments are accumulated in float64 (numpy's float32
// pairwise mean differs from that at the 1e-7 level).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include "qs_amd.h"
#include "qs_host.h"

extern thread_local char qs_g_err[512];
#define QN_FAIL(code, ...) do { snprintf(qs_g_err, sizeof(qs_g_err), __VA_ARGS__); return (code); } while (0)
#define QN_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) QN_FAIL(-2, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)

struct qs_norm {
    int n, o, device;
    float clip_obs, clip_rew, gamma, eps;
    hipStream_t stream;
    // [0, C) mean, [C, 2C) var, [2C, 2C+2) count (obs, returns), [2C+2, 3C+2) batch mean, [3C+2, 4C+2) batch var,
    // [4C+2, 5C+2) 1 / sqrt(var + eps); C = o + 1, column o = returns
    double* d_stat;
    void* d_part;    // per-block moments of the batch [n_parts][C] (k_norm_partial -> k_norm_update)
    int n_parts, rows_per_block;
    double* d_ret;   // discounted return of every environment (VecNormalize.returns)
    unsigned* d_sync;   // k_norm_fused: [0] blocks that have delivered their moments (ever), [1] launches whose statistics are ready
    unsigned gen;       // training launches of k_norm_fused so far
    int fused;          // all n_parts blocks of k_norm_fused are resident at once (checked at create): one launch per step
};

namespace {
struct Moments { double n, mean, m2; };
__device__ inline Moments merge(Moments a, Moments b) {   // Chan et al. pairwise merge, the same formula RunningMeanStd uses
    if (b.n == 0.0) return a;
    if (a.n == 0.0) return b;
    double tot = a.n + b.n, delta = b.mean - a.mean;
    Moments r; r.n = tot; r.mean = a.mean + delta * b.n / tot; r.m2 = a.m2 + b.m2 + delta * delta * a.n * b.n / tot;
    return r;
}

// Batch moments, stage 1: a block reduces `rows_per_block` consecutive rows, every column at once -- thread (cx, ry) takes column
// c0 + cx of rows ry, ry + 8, ...: a wavefront reads two whole rows (coalesced), not one column with a stride of a row.  The
// returns column (index o) also advances VecNormalize.returns (vec_normalize.py:_update_reward).
#define QN_MAX_PARTS 1024
__global__ __launch_bounds__(256) void k_norm_partial(const float* __restrict__ obs, const float* __restrict__ rew, double* __restrict__ ret, int n, int o,
                                                      double gamma, int with_obs, int with_ret, int rows_per_block, Moments* __restrict__ part) {
    const int C = o + 1, cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
    __shared__ Moments sh[8][32];
    for (int c0 = 0; c0 < C; c0 += 32) {
        const int c = c0 + cx;
        const bool act = c < C && ((c < o && with_obs) || (c == o && with_ret));
        double pivot = 0.0, s1 = 0.0, s2 = 0.0, cnt = 0.0;       // sums about the first value: no cancellation in s2 - s1^2 / cnt
        if (act)
            for (int i = r0 + ry; i < r1; i += 8) {
                double x;
                if (c < o) x = (double)obs[(size_t)i * o + c];
                else { x = ret[i] * gamma + (double)rew[i]; ret[i] = x; }
                if (cnt == 0.0) pivot = x;
                const double d = x - pivot;
                s1 += d; s2 += d * d; cnt += 1.0;
            }
        Moments m; m.n = cnt; m.mean = cnt > 0.0 ? pivot + s1 / cnt : 0.0; m.m2 = cnt > 0.0 ? s2 - s1 * s1 / cnt : 0.0;
        sh[ry][cx] = m;
        __syncthreads();
        if (ry == 0 && c < C) {
            for (int k = 1; k < 8; k++) m = merge(m, sh[k][cx]);
            part[(size_t)blockIdx.x * C + c] = m;
        }
        __syncthreads();
    }
}

// Stage 2 + running_mean_std.py update_from_moments: one block merges the per-block moments of every column (Chan's pairwise
// merge, the formula RunningMeanStd itself uses), folds the batch into the running statistics and refreshes 1 / sqrt(var + eps).
__global__ __launch_bounds__(1024) void k_norm_update(double* __restrict__ stat, int o, double batch_count, int with_obs, int with_ret,
                                                      const Moments* __restrict__ part, int n_parts, double eps) {
    const int C = o + 1, cx = threadIdx.x & 31, gy = threadIdx.x >> 5;
    __shared__ Moments sh[32][33];
    __shared__ double old_count[2];
    if (threadIdx.x < 2) old_count[threadIdx.x] = stat[2 * C + threadIdx.x];
    __syncthreads();
    for (int c0 = 0; c0 < C; c0 += 32) {
        const int c = c0 + cx;
        Moments m; m.n = 0.0; m.mean = 0.0; m.m2 = 0.0;
        if (c < C)
            for (int g = gy; g < n_parts; g += 32) m = merge(m, part[(size_t)g * C + c]);
        sh[gy][cx] = m;
        __syncthreads();
        for (int s = 16; s > 0; s >>= 1) {
            if (gy < s) sh[gy][cx] = merge(sh[gy][cx], sh[gy + s][cx]);
            __syncthreads();
        }
        if (gy == 0 && c < C && ((c < o && with_obs) || (c == o && with_ret))) {
            const Moments b = sh[0][cx];
            const double bm = b.mean, bv = b.m2 / b.n;
            const double count = old_count[c == o ? 1 : 0], mean = stat[c], var = stat[C + c];
            const double delta = bm - mean, tot = count + batch_count;
            const double new_var = (var * count + bv * batch_count + delta * delta * count * batch_count / (count + batch_count)) / (count + batch_count);
            stat[c] = mean + delta * batch_count / tot;
            stat[C + c] = new_var;
            stat[2 * C + 2 + c] = bm; stat[3 * C + 2 + c] = bv;
            stat[4 * C + 2 + c] = 1.0 / sqrt(new_var + eps);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && with_obs) stat[2 * C] = old_count[0] + batch_count;
    if (threadIdx.x == 0 && with_ret) stat[2 * C + 1] = old_count[1] + batch_count;
}

// vec_normalize.py: normalize_obs (incl. terminal observations), normalize_reward, returns[dones] = 0
__global__ void k_norm_apply(float* __restrict__ obs, float* __restrict__ term_obs, float* __restrict__ rew, const uint8_t* __restrict__ done,
                             double* __restrict__ ret, int n, int o, const double* __restrict__ stat, double eps, double clip_obs, double clip_rew,
                             int norm_obs, int norm_rew, float* __restrict__ raw_obs, float* __restrict__ raw_rew) {
    const int C = o + 1;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (raw_obs && i < (size_t)n * o) raw_obs[i] = obs[i];
    if (raw_rew && i < (size_t)n) raw_rew[i] = rew[i];
    if (i < (size_t)n * o && norm_obs) {
        const int c = (int)(i % o);
        const double mean = stat[c], inv = stat[4 * C + 2 + c];   // 1 / sqrt(var + eps), refreshed by k_norm_update / qs_norm_set_stats
        obs[i] = (float)fmin(fmax(((double)obs[i] - mean) * inv, -clip_obs), clip_obs);
        if (term_obs) term_obs[i] = (float)fmin(fmax(((double)term_obs[i] - mean) * inv, -clip_obs), clip_obs);
    }
    if (i < (size_t)n && rew) {
        if (norm_rew) rew[i] = (float)fmin(fmax((double)rew[i] / sqrt(stat[C + o] + eps), -clip_rew), clip_rew);
        if (done && done[i]) ret[i] = 0.0;
    }
}
// (the three-launch path's share of the host path's compact terminal list, see k_norm_fused)
__global__ void k_norm_tail(float* __restrict__ tail_rows, int tail_cap, int o, const double* __restrict__ stat, double clip_obs) {
    const int C = o + 1, e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= tail_cap * o) return;
    const int r = e / o, c = e - r * o;
    float* p = tail_rows + (size_t)r * (o + 1) + 1 + c;
    *p = (float)fmin(fmax(((double)*p - stat[c]) * stat[4 * C + 2 + c], -clip_obs), clip_obs);
}

// The three kernels above as ONE launch (training mode: a step of a learner pays one launch latency instead of three).
// Block b owns rows [b * rows_per_block, ...): it delivers their moments, waits until the LAST block to deliver has folded all of them
// into the running statistics (the code of k_norm_update on 8 x 32 threads), then normalises its own rows.  Two device-wide
// hand-overs: a counter that every block bumps behind its moments (`sync[0]`: launch g is complete at (g + 1) * gridDim.x) and a word
// the folding block sets behind the statistics (`sync[1]` = g + 1).  Waiting inside a kernel needs every block resident: qs_norm_create
// checks that (occupancy x CUs >= n_parts) and keeps the three launches otherwise.  training = 0: no hand-over, k_norm_apply's work only.
__global__ __launch_bounds__(256) void k_norm_fused(float* __restrict__ obs, float* __restrict__ term_obs, float* __restrict__ rew, const uint8_t* __restrict__ done,
                                                    double* __restrict__ ret, int n, int o, double gamma, double* __restrict__ stat, Moments* __restrict__ part,
                                                    int rows_per_block, unsigned* __restrict__ sync, unsigned gen, double batch_count, double eps,
                                                    double clip_obs, double clip_rew, int training, int with_obs, int with_ret, int norm_obs, int norm_rew,
                                                    float* __restrict__ raw_obs, float* __restrict__ raw_rew, float* __restrict__ tail_rows, int tail_cap) {
    const int C = o + 1, cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
    __shared__ Moments sh[8][33];
    __shared__ double old_count[2];
    __shared__ int s_last;
    if (training) {
        // ---- this block's moments (k_norm_partial)
        for (int c0 = 0; c0 < C; c0 += 32) {
            const int c = c0 + cx;
            const bool act = c < C && ((c < o && with_obs) || (c == o && with_ret));
            double pivot = 0.0, s1 = 0.0, s2 = 0.0, cnt = 0.0;
            if (act)
                for (int i = r0 + ry; i < r1; i += 8) {
                    double x;
                    if (c < o) x = (double)obs[(size_t)i * o + c];
                    else { x = ret[i] * gamma + (double)rew[i]; ret[i] = x; }
                    if (cnt == 0.0) pivot = x;
                    const double d = x - pivot;
                    s1 += d; s2 += d * d; cnt += 1.0;
                }
            Moments m; m.n = cnt; m.mean = cnt > 0.0 ? pivot + s1 / cnt : 0.0; m.m2 = cnt > 0.0 ? s2 - s1 * s1 / cnt : 0.0;
            sh[ry][cx] = m;
            __syncthreads();
            if (ry == 0 && c < C) {
                for (int k = 1; k < 8; k++) m = merge(m, sh[k][cx]);
                part[(size_t)blockIdx.x * C + c] = m;
            }
            __syncthreads();
        }
        // ---- delivered: the last block to say so folds the batch into the running statistics (k_norm_update)
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) s_last = atomicAdd(&sync[0], 1u) + 1u == (gen + 1u) * gridDim.x;
        __syncthreads();
        if (s_last) {
            __threadfence();
            if (threadIdx.x < 2) old_count[threadIdx.x] = stat[2 * C + threadIdx.x];
            __syncthreads();
            for (int c0 = 0; c0 < C; c0 += 32) {
                const int c = c0 + cx;
                Moments m; m.n = 0.0; m.mean = 0.0; m.m2 = 0.0;
                if (c < C)
                    for (int g = ry; g < (int)gridDim.x; g += 8) m = merge(m, part[(size_t)g * C + c]);
                sh[ry][cx] = m;
                __syncthreads();
                for (int s = 4; s > 0; s >>= 1) {
                    if (ry < s) sh[ry][cx] = merge(sh[ry][cx], sh[ry + s][cx]);
                    __syncthreads();
                }
                if (ry == 0 && c < C && ((c < o && with_obs) || (c == o && with_ret))) {
                    const Moments b = sh[0][cx];
                    const double bm = b.mean, bv = b.m2 / b.n;
                    const double count = old_count[c == o ? 1 : 0], mean = stat[c], var = stat[C + c];
                    const double delta = bm - mean, tot = count + batch_count;
                    const double new_var = (var * count + bv * batch_count + delta * delta * count * batch_count / (count + batch_count)) / (count + batch_count);
                    stat[c] = mean + delta * batch_count / tot;
                    stat[C + c] = new_var;
                    stat[2 * C + 2 + c] = bm; stat[3 * C + 2 + c] = bv;
                    stat[4 * C + 2 + c] = 1.0 / sqrt(new_var + eps);
                }
                __syncthreads();
            }
            if (threadIdx.x == 0 && with_obs) stat[2 * C] = old_count[0] + batch_count;
            if (threadIdx.x == 0 && with_ret) stat[2 * C + 1] = old_count[1] + batch_count;
            __threadfence();
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(&sync[1], gen + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (threadIdx.x == 0)
                while (__hip_atomic_load(&sync[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != gen + 1u) __builtin_amdgcn_s_sleep(2);
            __syncthreads();
            __threadfence();
        }
    }
    // ---- normalise this block's rows (k_norm_apply); the statistics were written by another block of this launch: no cached copy
    const volatile double* vstat = stat;
    const int rows = r1 - r0;
    if (rows <= 0) return;
    // the host path's compact list of terminal observations [tail_cap][1 + o] (environment index, observation): the last block's share
    if (tail_rows && norm_obs && blockIdx.x == gridDim.x - 1)
        for (int e = threadIdx.x; e < tail_cap * o; e += 256) {
            const int r = e / o, c = e - r * o;
            float* p = tail_rows + (size_t)r * (o + 1) + 1 + c;
            *p = (float)fmin(fmax(((double)*p - vstat[c]) * vstat[4 * C + 2 + c], -clip_obs), clip_obs);
        }
    for (int e = threadIdx.x; e < rows * o; e += 256) {
        const size_t i = (size_t)r0 * o + e;
        const int c = (int)(i % o);
        const float x = obs[i];
        if (raw_obs) raw_obs[i] = x;
        if (norm_obs) {
            const double mean = vstat[c], inv = vstat[4 * C + 2 + c];
            obs[i] = (float)fmin(fmax(((double)x - mean) * inv, -clip_obs), clip_obs);
            if (term_obs) term_obs[i] = (float)fmin(fmax(((double)term_obs[i] - mean) * inv, -clip_obs), clip_obs);
        }
    }
    if (rew)
        for (int e = threadIdx.x; e < rows; e += 256) {
            const int i = r0 + e;
            const float x = rew[i];
            if (raw_rew) raw_rew[i] = x;
            if (norm_rew) rew[i] = (float)fmin(fmax((double)x / sqrt(vstat[C + o] + eps), -clip_rew), clip_rew);
            if (done && done[i]) ret[i] = 0.0;
        }
}
}  // namespace

extern "C" {

int qs_norm_create(int n_envs, int obs_dim, float clip_obs, float clip_reward, float gamma, float epsilon, int device, qs_norm** out) {
    if (!out || n_envs <= 0 || obs_dim <= 0 || obs_dim > 255) QN_FAIL(-1, "bad argument (n_envs %d, obs_dim %d)", n_envs, obs_dim);
    int ndev = 0;
    hipError_t derr = hipGetDeviceCount(&ndev);
    if (derr != hipSuccess || ndev <= 0) QN_FAIL(-3, "no HIP device available: this library has no CPU path");
    if (device < 0 || device >= ndev) QN_FAIL(-3, "HIP device %d out of range (%d visible)", device, ndev);
    DeviceGuard guard(device);
    qs_norm* h = new (std::nothrow) qs_norm();
    if (!h) QN_FAIL(-4, "out of host memory");
    memset(h, 0, sizeof(*h));
    h->n = n_envs; h->o = obs_dim; h->device = device; h->clip_obs = clip_obs; h->clip_rew = clip_reward; h->gamma = gamma; h->eps = epsilon;
    const int C = obs_dim + 1;
#define QN_HIP_H(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { snprintf(qs_g_err, sizeof(qs_g_err), "%s failed: %s", #call, hipGetErrorString(e_)); qs_norm_destroy(h); return -2; } } while (0)
    QN_HIP_H(hipMalloc(&h->d_stat, (size_t)(5 * C + 2) * sizeof(double)));
    // rows per block: 128 from N = 8192 on (64 blocks there: the folding block of k_norm_fused merges 8 partial moments per thread), 64 below,
    // never more than QN_MAX_PARTS blocks
    h->rows_per_block = ((n_envs + QN_MAX_PARTS - 1) / QN_MAX_PARTS + 7) / 8 * 8;
    const int want = n_envs >= 8192 ? 128 : 64;
    if (h->rows_per_block < want) h->rows_per_block = want;
    h->n_parts = (n_envs + h->rows_per_block - 1) / h->rows_per_block;
    QN_HIP_H(hipMalloc(&h->d_sync, 2 * sizeof(unsigned)));
    QN_HIP_H(hipMemset(h->d_sync, 0, 2 * sizeof(unsigned)));
    {   // k_norm_fused waits inside the kernel for its other blocks: only if all of them fit the device at once (QS_NORM_FUSED=0: never)
        int per_cu = 0; hipDeviceProp_t prop;
        const char* sw = getenv("QS_NORM_FUSED");
        if (!(sw && sw[0] == '0') && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_norm_fused, 256, 0) == hipSuccess &&
            hipGetDeviceProperties(&prop, device) == hipSuccess)
            h->fused = (long long)per_cu * prop.multiProcessorCount >= 2LL * h->n_parts;   // (twice: room for whatever else runs on the device)
    }
    QN_HIP_H(hipMalloc(&h->d_part, (size_t)h->n_parts * C * 3 * sizeof(double)));
    QN_HIP_H(hipMalloc(&h->d_ret, (size_t)n_envs * sizeof(double)));
    QN_HIP_H(hipMemset(h->d_ret, 0, (size_t)n_envs * sizeof(double)));
    double init[5 * 256 + 2];
    for (int c = 0; c < C; c++) { init[c] = 0.0; init[C + c] = 1.0; init[2 * C + 2 + c] = 0.0; init[3 * C + 2 + c] = 0.0; init[4 * C + 2 + c] = 1.0 / sqrt(1.0 + (double)epsilon); }
    init[2 * C] = init[2 * C + 1] = 1e-4;   // RunningMeanStd(epsilon=1e-4)
    QN_HIP_H(hipMemcpy(h->d_stat, init, (size_t)(5 * C + 2) * sizeof(double), hipMemcpyHostToDevice));
#undef QN_HIP_H
    *out = h;
    return 0;
}

void qs_norm_destroy(qs_norm* h) {
    if (!h) return;
    QS_ON_DEVICE(h);
    hipStreamSynchronize(h->stream);
    hipFree(h->d_stat); hipFree(h->d_ret); hipFree(h->d_part); hipFree(h->d_sync);
    delete h;
}

int qs_norm_set_stream(qs_norm* h, void* s) { if (!h) QN_FAIL(-1, "null handle"); h->stream = (hipStream_t)s; return 0; }

int qs_norm_set_stats(qs_norm* h, const double* obs_mean, const double* obs_var, double obs_count, double ret_mean, double ret_var, double ret_count) {
    if (!h || !obs_mean || !obs_var) QN_FAIL(-1, "null argument");
    QS_ON_DEVICE(h);
    const int C = h->o + 1;
    double buf[2 * 256 + 2];
    for (int c = 0; c < h->o; c++) { buf[c] = obs_mean[c]; buf[C + c] = obs_var[c]; }
    buf[h->o] = ret_mean; buf[C + h->o] = ret_var; buf[2 * C] = obs_count; buf[2 * C + 1] = ret_count;
    QN_HIP(hipStreamSynchronize(h->stream));
    QN_HIP(hipMemcpy(h->d_stat, buf, (size_t)(2 * C + 2) * sizeof(double), hipMemcpyHostToDevice));
    double inv[256];
    for (int c = 0; c < C; c++) inv[c] = 1.0 / sqrt(buf[C + c] + (double)h->eps);
    QN_HIP(hipMemcpy(h->d_stat + 4 * C + 2, inv, (size_t)C * sizeof(double), hipMemcpyHostToDevice));
    return 0;
}

int qs_norm_get_stats(qs_norm* h, double* obs_mean, double* obs_var, double* obs_count, double* ret_mean, double* ret_var, double* ret_count) {
    if (!h) QN_FAIL(-1, "null handle");
    QS_ON_DEVICE(h);
    const int C = h->o + 1;
    double buf[2 * 256 + 2];
    QN_HIP(hipStreamSynchronize(h->stream));
    QN_HIP(hipMemcpy(buf, h->d_stat, (size_t)(2 * C + 2) * sizeof(double), hipMemcpyDeviceToHost));
    for (int c = 0; c < h->o; c++) { if (obs_mean) obs_mean[c] = buf[c]; if (obs_var) obs_var[c] = buf[C + c]; }
    if (ret_mean) *ret_mean = buf[h->o];
    if (ret_var) *ret_var = buf[C + h->o];
    if (obs_count) *obs_count = buf[2 * C];
    if (ret_count) *ret_count = buf[2 * C + 1];
    return 0;
}

// VecNormalize.reset (vec_normalize.py): returns = 0; obs_rms.update(obs) when training; normalize
int qs_norm_reset(qs_norm* h, float* obs, int training, int norm_obs) {
    if (!h || !obs) QN_FAIL(-1, "null argument");
    QS_ON_DEVICE(h);
    QN_HIP(hipMemsetAsync(h->d_ret, 0, (size_t)h->n * sizeof(double), h->stream));
    const int upd = training && norm_obs;
    if (h->fused) {
        hipLaunchKernelGGL(k_norm_fused, dim3(h->n_parts), dim3(256), 0, h->stream, obs, (float*)nullptr, (float*)nullptr, (const uint8_t*)nullptr, h->d_ret, h->n, h->o,
                           (double)h->gamma, h->d_stat, (Moments*)h->d_part, h->rows_per_block, h->d_sync, h->gen, (double)h->n, (double)h->eps, (double)h->clip_obs,
                           (double)h->clip_rew, upd, 1, 0, norm_obs, 0, (float*)nullptr, (float*)nullptr, (float*)nullptr, 0);
        QN_HIP(hipGetLastError());
        if (upd) h->gen++;
        return 0;
    }
    if (upd) {
        hipLaunchKernelGGL(k_norm_partial, dim3(h->n_parts), dim3(256), 0, h->stream, obs, (const float*)nullptr, h->d_ret, h->n, h->o, (double)h->gamma, 1, 0,
                           h->rows_per_block, (Moments*)h->d_part);
        hipLaunchKernelGGL(k_norm_update, dim3(1), dim3(1024), 0, h->stream, h->d_stat, h->o, (double)h->n, 1, 0, (const Moments*)h->d_part, h->n_parts, (double)h->eps);
    }
    const size_t total = (size_t)h->n * h->o;
    hipLaunchKernelGGL(k_norm_apply, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, obs, (float*)nullptr, (float*)nullptr, (const uint8_t*)nullptr,
                       h->d_ret, h->n, h->o, h->d_stat, (double)h->eps, (double)h->clip_obs, (double)h->clip_rew, norm_obs, 0, (float*)nullptr, (float*)nullptr);
    QN_HIP(hipGetLastError());
    return 0;
}

// VecNormalize.step_wait on the arrays a step produced (all in place, device memory; term_obs may be NULL).  tail_rows (may be NULL): the host
// path's compact list of the step's terminal observations, [tail_cap][1 + obs_dim], normalised like term_obs
int qs_norm_step_rows(qs_norm* h, float* obs, float* rew, const uint8_t* done, float* term_obs, int training, int norm_obs, int norm_reward,
                      float* raw_obs, float* raw_rew, float* tail_rows, int tail_cap) {
    if (!h || !obs || !rew || !done) QN_FAIL(-1, "null argument");
    QS_ON_DEVICE(h);
    if (h->fused) {
        hipLaunchKernelGGL(k_norm_fused, dim3(h->n_parts), dim3(256), 0, h->stream, obs, term_obs, rew, done, h->d_ret, h->n, h->o, (double)h->gamma, h->d_stat,
                           (Moments*)h->d_part, h->rows_per_block, h->d_sync, h->gen, (double)h->n, (double)h->eps, (double)h->clip_obs, (double)h->clip_rew,
                           training ? 1 : 0, norm_obs, 1, norm_obs, norm_reward, raw_obs, raw_rew, tail_rows, tail_cap);
        QN_HIP(hipGetLastError());
        if (training) h->gen++;
        return 0;
    }
    if (training) {
        hipLaunchKernelGGL(k_norm_partial, dim3(h->n_parts), dim3(256), 0, h->stream, obs, rew, h->d_ret, h->n, h->o, (double)h->gamma, norm_obs, 1,
                           h->rows_per_block, (Moments*)h->d_part);
        hipLaunchKernelGGL(k_norm_update, dim3(1), dim3(1024), 0, h->stream, h->d_stat, h->o, (double)h->n, norm_obs, 1, (const Moments*)h->d_part, h->n_parts, (double)h->eps);
    }
    const size_t total = (size_t)h->n * h->o;
    hipLaunchKernelGGL(k_norm_apply, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, obs, term_obs, rew, done, h->d_ret, h->n, h->o,
                       h->d_stat, (double)h->eps, (double)h->clip_obs, (double)h->clip_rew, norm_obs, norm_reward, raw_obs, raw_rew);
    if (tail_rows && norm_obs && tail_cap > 0)
        hipLaunchKernelGGL(k_norm_tail, dim3((unsigned)((tail_cap * h->o + 255) / 256)), dim3(256), 0, h->stream, tail_rows, tail_cap, h->o, h->d_stat, (double)h->clip_obs);
    QN_HIP(hipGetLastError());
    return 0;
}

int qs_norm_step(qs_norm* h, float* obs, float* rew, const uint8_t* done, float* term_obs, int training, int norm_obs, int norm_reward,
                 float* raw_obs, float* raw_rew) {
    return qs_norm_step_rows(h, obs, rew, done, term_obs, training, norm_obs, norm_reward, raw_obs, raw_rew, nullptr, 0);
}

}  // extern "C"
