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
    // [0, C) mean, [C, 2C) var, [2C, 2C+2) count (obs, returns), [2C+2, 3C+2) batch mean, [3C+2, 4C+2) batch var; C = o + 1, column o = returns
    double* d_stat;
    double* d_ret;   // discounted return of every environment (VecNormalize.returns)
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

// one block per column (obs_dim observation columns + the returns column): batch mean and population variance over the N rows
__global__ void k_norm_moments(const float* __restrict__ obs, const float* __restrict__ rew, double* __restrict__ ret, int n, int o, double gamma,
                               int with_obs, int with_ret, double* __restrict__ stat) {
    const int c = blockIdx.x, C = o + 1;
    if ((c < o && !with_obs) || (c == o && !with_ret)) return;
    Moments m; m.n = 0.0; m.mean = 0.0; m.m2 = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        double x;
        if (c < o) x = (double)obs[(size_t)i * o + c];
        else { x = ret[i] * gamma + (double)rew[i]; ret[i] = x; }   // vec_normalize.py:_update_reward
        m.n += 1.0;
        double d = x - m.mean;
        m.mean += d / m.n;
        m.m2 += d * (x - m.mean);
    }
    __shared__ Moments sh[256];
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = merge(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) { stat[2 * C + 2 + c] = sh[0].mean; stat[3 * C + 2 + c] = sh[0].m2 / sh[0].n; }
}

// running_mean_std.py: update_from_moments
__global__ void k_norm_update(double* __restrict__ stat, int o, double batch_count, int with_obs, int with_ret) {
    const int c = threadIdx.x, C = o + 1;
    const bool active = c < C && ((c < o && with_obs) || (c == o && with_ret));
    double new_mean = 0.0, new_var = 0.0, tot = 0.0;
    if (active) {
        const double count = stat[2 * C + (c == o ? 1 : 0)];
        const double mean = stat[c], var = stat[C + c], bm = stat[2 * C + 2 + c], bv = stat[3 * C + 2 + c];
        const double delta = bm - mean;
        tot = count + batch_count;
        new_mean = mean + delta * batch_count / tot;
        new_var = (var * count + bv * batch_count + delta * delta * count * batch_count / (count + batch_count)) / (count + batch_count);
    }
    __syncthreads();   // every column has read the old count
    if (active) {
        stat[c] = new_mean; stat[C + c] = new_var;
        if (c == 0) stat[2 * C] = tot;
        if (c == o) stat[2 * C + 1] = tot;
    }
}

// vec_normalize.py: normalize_obs (incl. terminal observations), normalize_reward, returns[dones] = 0
__global__ void k_norm_apply(float* __restrict__ obs, float* __restrict__ term_obs, float* __restrict__ rew, const uint8_t* __restrict__ done,
                             double* __restrict__ ret, int n, int o, const double* __restrict__ stat, double eps, double clip_obs, double clip_rew,
                             int norm_obs, int norm_rew) {
    const int C = o + 1;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)n * o && norm_obs) {
        const int c = (int)(i % o);
        const double mean = stat[c], inv = 1.0 / sqrt(stat[C + c] + eps);
        obs[i] = (float)fmin(fmax(((double)obs[i] - mean) * inv, -clip_obs), clip_obs);
        if (term_obs) term_obs[i] = (float)fmin(fmax(((double)term_obs[i] - mean) * inv, -clip_obs), clip_obs);
    }
    if (i < (size_t)n && rew) {
        if (norm_rew) rew[i] = (float)fmin(fmax((double)rew[i] / sqrt(stat[C + o] + eps), -clip_rew), clip_rew);
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
    QN_HIP_H(hipMalloc(&h->d_stat, (size_t)(4 * C + 2) * sizeof(double)));
    QN_HIP_H(hipMalloc(&h->d_ret, (size_t)n_envs * sizeof(double)));
    QN_HIP_H(hipMemset(h->d_ret, 0, (size_t)n_envs * sizeof(double)));
    double init[4 * 256 + 2];
    for (int c = 0; c < C; c++) { init[c] = 0.0; init[C + c] = 1.0; init[2 * C + 2 + c] = 0.0; init[3 * C + 2 + c] = 0.0; }
    init[2 * C] = init[2 * C + 1] = 1e-4;   // RunningMeanStd(epsilon=1e-4)
    QN_HIP_H(hipMemcpy(h->d_stat, init, (size_t)(4 * C + 2) * sizeof(double), hipMemcpyHostToDevice));
#undef QN_HIP_H
    *out = h;
    return 0;
}

void qs_norm_destroy(qs_norm* h) {
    if (!h) return;
    QS_ON_DEVICE(h);
    hipStreamSynchronize(h->stream);
    hipFree(h->d_stat); hipFree(h->d_ret);
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
    if (upd) {
        hipLaunchKernelGGL(k_norm_moments, dim3(h->o + 1), dim3(256), 0, h->stream, obs, (const float*)nullptr, h->d_ret, h->n, h->o, (double)h->gamma, 1, 0, h->d_stat);
        hipLaunchKernelGGL(k_norm_update, dim3(1), dim3(256), 0, h->stream, h->d_stat, h->o, (double)h->n, 1, 0);
    }
    const size_t total = (size_t)h->n * h->o;
    hipLaunchKernelGGL(k_norm_apply, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, obs, (float*)nullptr, (float*)nullptr, (const uint8_t*)nullptr,
                       h->d_ret, h->n, h->o, h->d_stat, (double)h->eps, (double)h->clip_obs, (double)h->clip_rew, norm_obs, 0);
    QN_HIP(hipGetLastError());
    return 0;
}

// VecNormalize.step_wait on the arrays a step produced (all in place, device memory; term_obs may be NULL)
int qs_norm_step(qs_norm* h, float* obs, float* rew, const uint8_t* done, float* term_obs, int training, int norm_obs, int norm_reward) {
    if (!h || !obs || !rew || !done) QN_FAIL(-1, "null argument");
    QS_ON_DEVICE(h);
    if (training) {
        hipLaunchKernelGGL(k_norm_moments, dim3(h->o + 1), dim3(256), 0, h->stream, obs, rew, h->d_ret, h->n, h->o, (double)h->gamma, norm_obs, 1, h->d_stat);
        hipLaunchKernelGGL(k_norm_update, dim3(1), dim3(256), 0, h->stream, h->d_stat, h->o, (double)h->n, norm_obs, 1);
    }
    const size_t total = (size_t)h->n * h->o;
    hipLaunchKernelGGL(k_norm_apply, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, obs, term_obs, rew, done, h->d_ret, h->n, h->o,
                       h->d_stat, (double)h->eps, (double)h->clip_obs, (double)h->clip_rew, norm_obs, norm_reward);
    QN_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
