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
    double clip_obs, clip_rew, gamma, eps;      // (double like SB3's Python floats: gamma = 0.99 as a float32 moved the returns' statistics by 7e-9)
    hipStream_t stream;
    // two copies of [0, C) mean, [C, 2C) var, [2C, 2C+2) count (obs, returns), [2C+2, 3C+2) batch mean, [3C+2, 4C+2) batch var,
    // [4C+2, 5C+2) 1 / sqrt(var + eps); C = o + 1, column o = returns.  A training step reads copy `cur` and writes the other one (every
    // block of k_norm_finish works out the new statistics from the old ones; block 0 stores them where no block reads), then `cur` flips.
    double* d_stat;
    int cur;
    void* d_part;    // per-block moments of the batch [n_parts][C] (k_norm_moments -> k_norm_finish)
    int n_parts, rows_per_block;
    double* d_ret;   // discounted return of every environment (VecNormalize.returns)
};
#define QN_STAT(h, which) ((h)->d_stat + (size_t)(which) * (5 * ((h)->o + 1) + 2))

namespace {
struct Moments { double n, mean, m2; };

// Both kernels are chains of memory round trips, not of arithmetic (written with one load per loop trip the three kernels of rounds 2-3
// took +55 us per step at N = 8192, `tools/time_vecnormalize.py`; a single kernel whose blocks wait for each other's moments was built and
// measured in round 4: the device-scope fences of the hand-over write back and invalidate a whole L2 per XCD -- 120 us).  Every phase
// issues all of a thread's loads before the first use (QN_CH at a time); a block's moments are plain sums about one of its own values (the
// first row's: no cancellation, and adding needs no division); the blocks' moments are added up as sums about block 0's mean -- Chan's
// pairwise merge (RunningMeanStd's own formula) written without its two divisions per pair.
#define QN_MAX_PARTS 512
#ifndef QN_CH
#define QN_CH 4
#endif
#ifndef QN_MCH
#define QN_MCH 8
#endif

// Batch moments: block b reduces rows [b * rows_per_block, ...), every column at once -- thread (cx, ry) takes column c0 + cx of rows ry,
// ry + 8, ...: a wavefront reads two whole rows (coalesced).  The returns column (index o) also advances VecNormalize.returns
// (vec_normalize.py:_update_reward).
__global__ __launch_bounds__(256) void k_norm_moments(const float* __restrict__ obs, const float* __restrict__ rew, double* __restrict__ ret, int n, int o,
                                                      double gamma, int with_obs, int with_ret, int rows_per_block, Moments* __restrict__ part) {
    const int C = o + 1, cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
    __shared__ double sh[3][8][33];
    __shared__ double piv[32];
    for (int c0 = 0; c0 < C; c0 += 32) {
        const int c = c0 + cx;
        const bool act = c < C && ((c < o && with_obs) || (c == o && with_ret));
        double s1 = 0.0, s2 = 0.0, cnt = 0.0, p = 0.0;
        for (int k0 = 0; k0 * 8 < rows_per_block; k0 += QN_CH) {
            double x[QN_CH];
#pragma unroll
            for (int k = 0; k < QN_CH; k++) x[k] = 0.0;
            if (act && c < o) {
                float v[QN_CH];
#pragma unroll
                for (int k = 0; k < QN_CH; k++) { const int i = r0 + ry + 8 * (k0 + k); v[k] = obs[(size_t)(i < r1 ? i : r0) * o + c]; }
#pragma unroll
                for (int k = 0; k < QN_CH; k++) x[k] = (double)v[k];
            } else if (act) {
                double q[QN_CH]; float v[QN_CH];
#pragma unroll
                for (int k = 0; k < QN_CH; k++) { const int i = r0 + ry + 8 * (k0 + k); q[k] = i < r1 ? ret[i] : 0.0; v[k] = i < r1 ? rew[i] : 0.0f; }
#pragma unroll
                for (int k = 0; k < QN_CH; k++) { x[k] = q[k] * gamma + (double)v[k]; if (r0 + ry + 8 * (k0 + k) < r1) ret[r0 + ry + 8 * (k0 + k)] = x[k]; }
            }
            if (k0 == 0) {   // the pivot: row r0 of the column (thread ry = 0 holds it in x[0])
                if (ry == 0) piv[cx] = x[0];
                __syncthreads();
                p = piv[cx];
            }
            if (act) {
#pragma unroll
                for (int k = 0; k < QN_CH; k++)
                    if (r0 + ry + 8 * (k0 + k) < r1) { const double d = x[k] - p; s1 += d; s2 += d * d; cnt += 1.0; }
            }
        }
        sh[0][ry][cx] = cnt; sh[1][ry][cx] = s1; sh[2][ry][cx] = s2;
        __syncthreads();
        if (ry == 0 && c < C) {
#pragma unroll
            for (int k = 1; k < 8; k++) { cnt += sh[0][k][cx]; s1 += sh[1][k][cx]; s2 += sh[2][k][cx]; }
            Moments m; m.n = cnt; m.mean = cnt > 0.0 ? p + s1 / cnt : 0.0; m.m2 = cnt > 0.0 ? s2 - s1 * s1 / cnt : 0.0;
            part[(size_t)blockIdx.x * C + c] = m;
        }
        __syncthreads();
    }
}

// running_mean_std.py update_from_moments + vec_normalize.py normalize_obs (incl. terminal observations), normalize_reward,
// returns[dones] = 0.  training: EVERY block adds up the blocks' moments (the same sums in the same order: the same bits) and folds the
// batch into the statistics it reads from `stat`; block 0 stores the result in `stat_next`.  Then each block normalises its own rows.
__global__ __launch_bounds__(256) void k_norm_finish(qs_norm_io io, double* __restrict__ ret, int n, int o, const double* __restrict__ stat, double* __restrict__ stat_next,
                                                     const Moments* __restrict__ part, int n_parts, int rows_per_block, double batch_count, double eps,
                                                     double clip_obs, double clip_rew, int training, int with_obs, int with_ret, int norm_obs, int norm_rew) {
    float* const obs = io.obs; float* const term_obs = io.term_obs; float* const rew = io.rew; const uint8_t* const done = io.done;
    float* const raw_obs = io.raw_obs; float* const raw_rew = io.raw_rew;
    float* const obs_to = io.out_obs ? io.out_obs : io.obs; float* const rew_to = io.out_rew ? io.out_rew : io.rew;     // (in place unless told otherwise)
    const int C = o + 1, cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
    __shared__ double sh[3][8][33];
    __shared__ double s_mean[256], s_inv[256];
    __shared__ double s_rvar;      // variance of the returns (column o) after this step's update
    for (int c0 = 0; c0 < C; c0 += 32) {
        const int c = c0 + cx, cc = c < C ? c : 0;
        const bool upd = training && c < C && ((c < o && with_obs) || (c == o && with_ret));
        if (training) {
            const double P = part[cc].mean;        // sums about block 0's mean
            double S0 = 0.0, S1 = 0.0, S2 = 0.0;
            for (int g0 = ry; g0 < n_parts; g0 += 8 * QN_MCH) {
                Moments m[QN_MCH];
#pragma unroll
                for (int k = 0; k < QN_MCH; k++) { const int g = g0 + 8 * k; m[k] = part[(size_t)(g < n_parts ? g : 0) * C + cc]; }
#pragma unroll
                for (int k = 0; k < QN_MCH; k++)
                    if (g0 + 8 * k < n_parts) { const double d = m[k].mean - P; S0 += m[k].n; S1 += m[k].n * d; S2 += m[k].m2 + m[k].n * d * d; }
            }
            sh[0][ry][cx] = S0; sh[1][ry][cx] = S1; sh[2][ry][cx] = S2;
            __syncthreads();
            if (ry == 0 && c < C) {
#pragma unroll
                for (int k = 1; k < 8; k++) { S0 += sh[0][k][cx]; S1 += sh[1][k][cx]; S2 += sh[2][k][cx]; }
                double mean = stat[c], var = stat[C + c], bm = stat[2 * C + 2 + c], bv = stat[3 * C + 2 + c], inv = stat[4 * C + 2 + c];
                if (upd) {
                    bm = P + S1 / S0; bv = (S2 - S1 * S1 / S0) / S0;
                    const double count = stat[2 * C + (c == o ? 1 : 0)];
                    const double delta = bm - mean, tot = count + batch_count;
                    const double new_var = (var * count + bv * batch_count + delta * delta * count * batch_count / (count + batch_count)) / (count + batch_count);
                    mean = mean + delta * batch_count / tot;
                    var = new_var;
                    inv = 1.0 / sqrt(new_var + eps);
                }
                s_mean[c] = mean; s_inv[c] = inv;
                if (c == o) s_rvar = var;
                if (blockIdx.x == 0) { stat_next[c] = mean; stat_next[C + c] = var; stat_next[2 * C + 2 + c] = bm; stat_next[3 * C + 2 + c] = bv; stat_next[4 * C + 2 + c] = inv; }
            }
            __syncthreads();
        } else if (ry == 0 && c < C) { s_mean[c] = stat[c]; s_inv[c] = stat[4 * C + 2 + c]; if (c == o) s_rvar = stat[C + c]; }
    }
    if (training && blockIdx.x == 0 && threadIdx.x < 2)
        stat_next[2 * C + threadIdx.x] = stat[2 * C + threadIdx.x] + ((threadIdx.x == 0 ? with_obs : with_ret) ? batch_count : 0.0);
    __syncthreads();
    const int rows = r1 - r0;
    if (rows <= 0) return;
    // the host path's compact list of terminal observations [tail_cap][1 + o] (environment index, observation): the last block's share
    if (io.tail_rows && blockIdx.x == gridDim.x - 1) {
        float* const tail_to = io.out_tail ? io.out_tail : io.tail_rows;
        // only the rows THIS step filled (io.tail_count; round 4 normalised all tail_cap rows, stale ones of earlier steps again and again)
        int filled = io.tail_cap;
        if (io.tail_count) { const unsigned long long c = *io.tail_count; filled = c < (unsigned long long)io.tail_cap ? (int)c : io.tail_cap; }
        for (int e = threadIdx.x; e < filled * (o + 1); e += 256) {
            const int r = e / (o + 1), c = e - r * (o + 1) - 1;
            const float x = io.tail_rows[e];
            if (c >= 0 && norm_obs) tail_to[e] = (float)fmin(fmax(((double)x - s_mean[c]) * s_inv[c], -clip_obs), clip_obs);
            else if (io.out_tail) tail_to[e] = x;
        }
    }
    // the flags travel with the arrays (io.out_done: the host block of qs_host_step_*)
    if (io.out_done)
        for (int e = threadIdx.x; e < rows; e += 256) { io.out_done[r0 + e] = done[r0 + e]; if (io.trunc) io.out_trunc[r0 + e] = io.trunc[r0 + e]; }
    const int total = rows * o;
    const size_t base = (size_t)r0 * o;
    for (int e0 = threadIdx.x; e0 < total; e0 += 256 * QN_CH) {
        float x[QN_CH], t[QN_CH];
#pragma unroll
        for (int k = 0; k < QN_CH; k++) {
            const int e = e0 + 256 * k, ee = e < total ? e : 0;
            x[k] = obs[base + ee];
            t[k] = term_obs && norm_obs ? term_obs[base + ee] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < QN_CH; k++) {
            const int e = e0 + 256 * k;
            if (e >= total) continue;
            if (raw_obs) raw_obs[base + e] = x[k];
            if (norm_obs) {
                const int c = (int)((base + e) % o);
                const double mean = s_mean[c], inv = s_inv[c];
                obs_to[base + e] = (float)fmin(fmax(((double)x[k] - mean) * inv, -clip_obs), clip_obs);
                if (term_obs) term_obs[base + e] = (float)fmin(fmax(((double)t[k] - mean) * inv, -clip_obs), clip_obs);
            } else if (io.out_obs) obs_to[base + e] = x[k];
        }
    }
    if (rew) {
        const double rstd = sqrt(s_rvar + eps);
        for (int e = threadIdx.x; e < rows; e += 256) {
            const int i = r0 + e;
            const float x = rew[i];
            if (raw_rew) raw_rew[i] = x;
            if (norm_rew) rew_to[i] = (float)fmin(fmax((double)x / rstd, -clip_rew), clip_rew);
            else if (io.out_rew) rew_to[i] = x;
            if (done && done[i]) ret[i] = 0.0;
        }
    }
}
}  // namespace

extern "C" {

int qs_norm_create(int n_envs, int obs_dim, double clip_obs, double clip_reward, double gamma, double epsilon, int device, qs_norm** out) {
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
    const int C = obs_dim + 1, S = 5 * C + 2;
#define QN_HIP_H(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { snprintf(qs_g_err, sizeof(qs_g_err), "%s failed: %s", #call, hipGetErrorString(e_)); qs_norm_destroy(h); return -2; } } while (0)
    QN_HIP_H(hipMalloc(&h->d_stat, (size_t)2 * S * sizeof(double)));
    // rows per block: 64 (128 blocks at N = 8192: eight rows per thread and column, sixteen blocks' moments per thread in k_norm_finish),
    // never more than QN_MAX_PARTS blocks
    h->rows_per_block = ((n_envs + QN_MAX_PARTS - 1) / QN_MAX_PARTS + 7) / 8 * 8;
    if (h->rows_per_block < 64) h->rows_per_block = 64;
    h->n_parts = (n_envs + h->rows_per_block - 1) / h->rows_per_block;
    QN_HIP_H(hipMalloc(&h->d_part, (size_t)h->n_parts * C * 3 * sizeof(double)));
    QN_HIP_H(hipMalloc(&h->d_ret, (size_t)n_envs * sizeof(double)));
    QN_HIP_H(hipMemset(h->d_ret, 0, (size_t)n_envs * sizeof(double)));
    double init[5 * 256 + 2];
    for (int c = 0; c < C; c++) { init[c] = 0.0; init[C + c] = 1.0; init[2 * C + 2 + c] = 0.0; init[3 * C + 2 + c] = 0.0; init[4 * C + 2 + c] = 1.0 / sqrt(1.0 + (double)epsilon); }
    init[2 * C] = init[2 * C + 1] = 1e-4;   // RunningMeanStd(epsilon=1e-4)
    for (int k = 0; k < 2; k++) QN_HIP_H(hipMemcpy(QN_STAT(h, k), init, (size_t)S * sizeof(double), hipMemcpyHostToDevice));
#undef QN_HIP_H
    *out = h;
    return 0;
}

void qs_norm_destroy(qs_norm* h) {
    if (!h) return;
    QS_ON_DEVICE(h);
    hipStreamSynchronize(h->stream);
    hipFree(h->d_stat); hipFree(h->d_ret); hipFree(h->d_part);
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
    QN_HIP(hipMemcpy(QN_STAT(h, h->cur), buf, (size_t)(2 * C + 2) * sizeof(double), hipMemcpyHostToDevice));
    double inv[256];
    for (int c = 0; c < C; c++) inv[c] = 1.0 / sqrt(buf[C + c] + (double)h->eps);
    QN_HIP(hipMemcpy(QN_STAT(h, h->cur) + 4 * C + 2, inv, (size_t)C * sizeof(double), hipMemcpyHostToDevice));
    return 0;
}

int qs_norm_get_stats(qs_norm* h, double* obs_mean, double* obs_var, double* obs_count, double* ret_mean, double* ret_var, double* ret_count) {
    if (!h) QN_FAIL(-1, "null handle");
    QS_ON_DEVICE(h);
    const int C = h->o + 1;
    double buf[2 * 256 + 2];
    QN_HIP(hipStreamSynchronize(h->stream));
    QN_HIP(hipMemcpy(buf, QN_STAT(h, h->cur), (size_t)(2 * C + 2) * sizeof(double), hipMemcpyDeviceToHost));
    for (int c = 0; c < h->o; c++) { if (obs_mean) obs_mean[c] = buf[c]; if (obs_var) obs_var[c] = buf[C + c]; }
    if (ret_mean) *ret_mean = buf[h->o];
    if (ret_var) *ret_var = buf[C + h->o];
    if (obs_count) *obs_count = buf[2 * C];
    if (ret_count) *ret_count = buf[2 * C + 1];
    return 0;
}

// the two launches of a step / reset: batch moments (training only), then fold + normalise
static int norm_launch(qs_norm* h, const qs_norm_io& io, int training, int with_obs, int with_ret, int norm_obs, int norm_reward) {
    if (training)
        hipLaunchKernelGGL(k_norm_moments, dim3(h->n_parts), dim3(256), 0, h->stream, io.obs, io.rew, h->d_ret, h->n, h->o, h->gamma, with_obs, with_ret,
                           h->rows_per_block, (Moments*)h->d_part);
    hipLaunchKernelGGL(k_norm_finish, dim3(h->n_parts), dim3(256), 0, h->stream, io, h->d_ret, h->n, h->o, QN_STAT(h, h->cur), QN_STAT(h, h->cur ^ 1),
                       (const Moments*)h->d_part, h->n_parts, h->rows_per_block, (double)h->n, h->eps, h->clip_obs, h->clip_rew,
                       training, with_obs, with_ret, norm_obs, norm_reward);
    QN_HIP(hipGetLastError());
    if (training) h->cur ^= 1;
    return 0;
}

int qs_norm_dims(const qs_norm* h, int* n_envs, int* obs_dim, int* device) {
    if (!h) QN_FAIL(-1, "null handle");
    if (n_envs) *n_envs = h->n;
    if (obs_dim) *obs_dim = h->o;
    if (device) *device = h->device;
    return 0;
}

// VecNormalize.reset (vec_normalize.py): returns = 0; obs_rms.update(obs) when training; normalize
int qs_norm_reset(qs_norm* h, float* obs, int training, int norm_obs) {
    if (!h || !obs) QN_FAIL(-1, "null argument");
    QS_ON_DEVICE(h);
    QN_HIP(hipMemsetAsync(h->d_ret, 0, (size_t)h->n * sizeof(double), h->stream));
    qs_norm_io io;
    memset(&io, 0, sizeof(io));
    io.obs = obs;
    return norm_launch(h, io, training && norm_obs, 1, 0, norm_obs, 0);
}

// VecNormalize.step_wait on the arrays a step produced (device memory; in place unless io->out_* say where the results go)
int qs_norm_step_io(qs_norm* h, const qs_norm_io* io, int training, int norm_obs, int norm_reward) {
    if (!h || !io || !io->obs || !io->rew || !io->done) QN_FAIL(-1, "null argument");
    if (io->out_done && (!io->out_obs || !io->out_rew || (io->trunc && !io->out_trunc) || (io->tail_rows && !io->out_tail)))
        QN_FAIL(-1, "qs_norm_io: out_done set, but not every array that is given has its out_ counterpart");
    QS_ON_DEVICE(h);
    return norm_launch(h, *io, training ? 1 : 0, norm_obs, 1, norm_obs, norm_reward);
}

int qs_norm_step(qs_norm* h, float* obs, float* rew, const uint8_t* done, float* term_obs, int training, int norm_obs, int norm_reward,
                 float* raw_obs, float* raw_rew) {
    qs_norm_io io;
    memset(&io, 0, sizeof(io));
    io.obs = obs; io.rew = rew; io.done = done; io.term_obs = term_obs; io.raw_obs = raw_obs; io.raw_rew = raw_rew;
    return qs_norm_step_io(h, &io, training, norm_obs, norm_reward);
}

}  // extern "C"
