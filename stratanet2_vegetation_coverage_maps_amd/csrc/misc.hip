// misc.hip -- BatchNorm finalisation, per-plot max pool, flat Adam step.
#include "mlp.h"

// ------------------------------------------------------------------------------------------------------------
// BatchNorm1d finalisation (torch defaults: eps 1e-5, momentum 0.1, biased variance to normalise, unbiased variance
// into running_var) -- model/point_net2.py:45-53.  The sums arrive as per-workgroup fp32 slots.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void bn_finalize_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ a, float* __restrict__ c, float* __restrict__ mean_out,
                                   float* __restrict__ invstd_out, const float* __restrict__ slots, int nslots,
                                   const unsigned long long* __restrict__ count_dev, long count_imm, int training) {
    // thread = (slot group g, column col of the 2C-wide slot row): consecutive threads read consecutive floats of one slot
    // row (coalesced; one lane per (channel, 16 slots) with a 2C-float stride took 13 us for 1024 slots), every thread adds
    // its slots g, g+G, ... in fp64 in a fixed order, then the G partials of a column are added in a fixed order: the
    // statistics are identical from run to run.
    __shared__ double s_part[1024];
    const int W2 = 2 * C, G = 1024 / W2;
    const int col = threadIdx.x % W2, g = threadIdx.x / W2;
    double acc = 0.0;
    if (training && g < G) {
        int k = g;
        for (; k + 3 * G < nslots; k += 4 * G) {      // four independent loads in flight
            const float v0 = slots[(size_t)k * W2 + col], v1 = slots[(size_t)(k + G) * W2 + col],
                        v2 = slots[(size_t)(k + 2 * G) * W2 + col], v3 = slots[(size_t)(k + 3 * G) * W2 + col];
            acc += (double)v0;
            acc += (double)v1;
            acc += (double)v2;
            acc += (double)v3;
        }
        for (; k < nslots; k += G) acc += (double)slots[(size_t)k * W2 + col];
    }
    s_part[threadIdx.x] = acc;
    __syncthreads();
    const int o = threadIdx.x;
    const bool live = o < C;
    double s1 = 0.0, s2 = 0.0;
    if (training && live) {
        for (int gg = 0; gg < G; ++gg) {
            s1 += s_part[gg * W2 + o];
            s2 += s_part[gg * W2 + C + o];
        }
    }
    const int j = 0;
    const float eps = 1e-5f, mom = 0.1f;
    if (!live || j != 0) return;
    float mean, invstd;
    if (training) {
        double n = count_dev ? (double)(*count_dev) : (double)count_imm;
        if (n < 1.0) n = 1.0;
        const double m = s1 / n;
        double var = s2 / n - m * m;
        if (var < 0.0) var = 0.0;
        mean = (float)m;
        invstd = 1.0f / sqrtf((float)var + eps);
        const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
        running_mean[o] = (1.f - mom) * running_mean[o] + mom * mean;
        running_var[o] = (1.f - mom) * running_var[o] + mom * (float)unbiased;
    } else {
        mean = running_mean[o];
        invstd = 1.0f / sqrtf(running_var[o] + eps);
    }
    const float aa = gamma[o] * invstd;
    a[o] = aa;
    c[o] = beta[o] - mean * aa;
    mean_out[o] = mean;
    invstd_out[o] = invstd;
}

int sn2_bn_finalize(const sn2_block* blk, int nslots, const unsigned long long* count_dev, long count_imm, int training,
                    hipStream_t st) {
    if (!blk || blk->cout <= 0 || blk->cout > 64 || nslots < 0 || nslots > SN2_STAT_SLOTS) return SN2_EINVAL;
    if (training && (!blk->stat_slots || nslots < 1)) return SN2_EINVAL;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(1024), 0, st, blk->cout, blk->gamma, blk->beta, blk->running_mean,
                       blk->running_var, blk->a, blk->c, blk->mean, blk->invstd, (const float*)blk->stat_slots, nslots,
                       count_dev, count_imm, training);
    SN2_RETURN_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// per-plot max of a*h + c over R rows -- global_max_pool, model/point_net2.py:39.  One workgroup per plot, thread =
// (row group, channel); first row wins ties (torch_scatter CPU updates on strict '>').
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void plot_max_kernel(const float* __restrict__ h, int hs, const float* __restrict__ a,
                                                       const float* __restrict__ c, int R, int C, float* __restrict__ out,
                                                       int* __restrict__ arg) {
    __shared__ float s_v[256];
    __shared__ int s_i[256];
    const int b = blockIdx.x, ch = threadIdx.x % C, g = threadIdx.x / C, G = 256 / C;
    float best = -INFINITY;
    int bi = 0x7FFFFFFF;
    if (g < G) {
        const float aa = a[ch], cc = c[ch];
        for (int r = g; r < R; r += G) {
            const float v = fmaf(aa, h[((size_t)b * R + r) * hs + ch], cc);
            if (v > best) {
                best = v;
                bi = r;
            }
        }
    }
    s_v[threadIdx.x] = best;
    s_i[threadIdx.x] = bi;
    __syncthreads();
    if (threadIdx.x < C) {
        for (int k = 1; k < G; ++k) {
            const float v = s_v[k * C + ch];
            const int i = s_i[k * C + ch];
            if (v > best || (v == best && i < bi)) {
                best = v;
                bi = i;
            }
        }
        out[(size_t)b * C + ch] = best;
        arg[(size_t)b * C + ch] = bi;
    }
}

extern "C" int sn2_plot_max_forward(const float* h, const float* a, const float* c, int B, int R_per_plot, int C,
                                    float* out, int* arg, void* stream) {
    if (!h || !a || !c || !out || !arg || B <= 0 || R_per_plot <= 0 || C <= 0 || C > 256) return SN2_EINVAL;
    const int hs = (C + 3) & ~3;
    hipLaunchKernelGGL(plot_max_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, h, hs, a, c, R_per_plot, C, out, arg);
    SN2_RETURN_LAUNCH();
}

__global__ void plot_max_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ arg, int B, int R, int C,
                                    int hs, float* __restrict__ dy) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, ch = i - b * C;
    const int r = arg[i];
    if (r >= 0 && r < R) dy[((size_t)b * R + r) * hs + ch] = dout[i];
}

extern "C" int sn2_plot_max_backward(const float* dout, const int* arg, int B, int R_per_plot, int C, float* dy,
                                     void* stream) {
    if (!dout || !arg || !dy || B <= 0 || R_per_plot <= 0 || C <= 0) return SN2_EINVAL;
    const int hs = (C + 3) & ~3;
    hipLaunchKernelGGL(plot_max_bwd_kernel, dim3(sn2_cdiv((long)B * C, 256)), dim3(256), 0, (hipStream_t)stream, dout, arg,
                       B, R_per_plot, C, hs, dy);
    SN2_RETURN_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// torch.optim.Adam (amsgrad=False, L2 weight decay folded into the gradient) on flat fp32 buffers.
// ------------------------------------------------------------------------------------------------------------
// The step counter lives on the device (adam_tick_kernel increments it, adam_kernel reads it), so a captured hipGraph
// of the whole training step replays with the right bias corrections.
__global__ void adam_tick_kernel(int* __restrict__ step) { *step += 1; }

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int n, float lr, float b1, float b2, float eps, float wd,
                            const int* __restrict__ step, float gscale) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float t = (float)(*step);
    const float bc1 = 1.f - powf(b1, t);
    const float bc2_sqrt = sqrtf(1.f - powf(b2, t));
    float grad = g[i] * gscale;
    const float pi = p[i];
    grad = fmaf(wd, pi, grad);
    const float mi = m[i] + (1.f - b1) * (grad - m[i]);       // lerp, as torch
    const float vi = b2 * v[i] + (1.f - b2) * grad * grad;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
}

extern "C" int sn2_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int* step_dev, float grad_scale,
                             void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !step_dev || n <= 0) return SN2_EINVAL;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
    hipLaunchKernelGGL(adam_kernel, dim3(sn2_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, (const int*)step_dev, grad_scale);
    SN2_RETURN_LAUNCH();
}

extern "C" int sn2_version(void) { return SN2_VERSION; }
