// misc.hip -- BatchNorm finalisation, per-plot max pool, flat Adam step.
#include "mlp.h"

// ------------------------------------------------------------------------------------------------------------
// BatchNorm1d finalisation (torch defaults: eps 1e-5, momentum 0.1, biased variance to normalise, unbiased variance
// into running_var) -- model/point_net2.py:45-53.  The sums arrive as per-workgroup fp32 slots.
// ------------------------------------------------------------------------------------------------------------
// APPLY (round 5): the set-abstraction levels' last finalisation also writes the level's output, out = a ext + c on the (rows, C)
// extremum (0 for a centroid that received no message) -- sa_finalize_kernel's job, a launch of its own until now.  The grid
// is then several workgroups: EVERY one finalises (the same slots in the same order: the same a, c; ~100 KB of L2 reads each),
// workgroup 0 alone writes the statistics and the running-statistics update, and each applies its share of the rows.
template <bool APPLY>
__global__ __launch_bounds__(1024) void bn_finalize_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ a, float* __restrict__ c, float* __restrict__ mean_out,
                                   float* __restrict__ invstd_out, const float* __restrict__ slots, int nslots,
                                   const unsigned long long* __restrict__ count_dev, long count_imm, int training,
                                   long long* __restrict__ num_batches_tracked, const float* __restrict__ ext = nullptr,
                                   const int* __restrict__ arg = nullptr, float* __restrict__ out = nullptr, long n_out = 0) {
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
        // all of a thread's loads in flight at once, added in slot order (1024 slots of 32 floats: 32 loads per thread; eight at
        // a time were four dependent round trips to L2 / HBM, the kernel's whole duration)
        for (; k + 31 * G < nslots; k += 32 * G) {
            float v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = slots[(size_t)(k + u * G) * W2 + col];
#pragma unroll
            for (int u = 0; u < 32; ++u) acc += (double)v[u];
        }
        for (; k + 7 * G < nslots; k += 8 * G) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slots[(size_t)(k + u * G) * W2 + col];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += (double)v[u];
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
    const bool lead = !APPLY || blockIdx.x == 0;
    __shared__ float s_ac[2][64];
    if (live) {
        if (lead && training && o == 0 && num_batches_tracked) *num_batches_tracked += 1;   // BatchNorm1d's counter: once per training forward
        float mean, invstd, aa, cc;
        if (training) {
            const double n = count_dev ? (double)(*count_dev) : (double)count_imm;
            sn2_bn_from_sums(s1, s2, n, gamma[o], beta[o], lead ? &running_mean[o] : nullptr, lead ? &running_var[o] : nullptr, aa, cc,
                             mean, invstd);
        } else {
            mean = running_mean[o];
            invstd = 1.0f / sqrtf(running_var[o] + 1e-5f);
            aa = gamma[o] * invstd;
            cc = beta[o] - mean * aa;
        }
        if (lead) {
            a[o] = aa;
            c[o] = cc;
            mean_out[o] = mean;
            invstd_out[o] = invstd;
        }
        if constexpr (APPLY) s_ac[0][o] = aa, s_ac[1][o] = cc;
    }
    if constexpr (APPLY) {
        __syncthreads();
        // out = a ext + c on this workgroup's share of the (rows, C) elements, four at a time (C is a multiple of 4)
        const long n4 = n_out >> 2;
        for (long i4 = (long)blockIdx.x * 1024 + threadIdx.x; i4 < n4; i4 += (long)gridDim.x * 1024) {
            const float4 e = reinterpret_cast<const float4*>(ext)[i4];
            const int4 g4 = reinterpret_cast<const int4*>(arg)[i4];
            const int o0 = (int)((i4 << 2) % C);
            float4 r;
            r.x = g4.x >= 0 ? fmaf(s_ac[0][o0], e.x, s_ac[1][o0]) : 0.f;
            r.y = g4.y >= 0 ? fmaf(s_ac[0][o0 + 1], e.y, s_ac[1][o0 + 1]) : 0.f;
            r.z = g4.z >= 0 ? fmaf(s_ac[0][o0 + 2], e.z, s_ac[1][o0 + 2]) : 0.f;
            r.w = g4.w >= 0 ? fmaf(s_ac[0][o0 + 3], e.w, s_ac[1][o0 + 3]) : 0.f;
            reinterpret_cast<float4*>(out)[i4] = r;
        }
    }
}

int sn2_bn_finalize(const sn2_block* blk, int nslots, const unsigned long long* count_dev, long count_imm, int training,
                    hipStream_t st) {
    if (!blk || blk->cout <= 0 || blk->cout > 64 || nslots < 0 || nslots > SN2_STAT_SLOTS) return SN2_EINVAL;
    if (training && (!blk->stat_slots || nslots < 1)) return SN2_EINVAL;
    // an EVAL pass sums nothing: 64 threads (one per channel).  A 1024-thread workgroup needs a CU with sixteen free wave slots at
    // once, and beside the full-chip kernels of other passes in flight it waited for them: 88 us on average, seven times per launch
    // of the parcel loop, on the feature stream's chain (profiles/r05_kernel_stats_inference.csv; round 5)
    hipLaunchKernelGGL(bn_finalize_kernel<false>, dim3(1), dim3(training ? 1024 : 64), 0, st, blk->cout, blk->gamma, blk->beta, blk->running_mean,
                       blk->running_var, blk->a, blk->c, blk->mean, blk->invstd, (const float*)blk->stat_slots, nslots,
                       count_dev, count_imm, training, blk->num_batches_tracked, (const float*)nullptr, (const int*)nullptr,
                       (float*)nullptr, 0L);
    SN2_RETURN_LAUNCH();
}

// ... and the level's output out = a ext + c (0 where arg < 0) in the same launch; ext, arg, out: (rows, cout), cout % 4 == 0
int sn2_bn_finalize_apply(const sn2_block* blk, int nslots, const unsigned long long* count_dev, long count_imm, int training,
                          const float* ext, const int* arg, float* out, long rows, hipStream_t st) {
    if (!blk || blk->cout <= 0 || blk->cout > 64 || (blk->cout & 3) || nslots < 0 || nslots > SN2_STAT_SLOTS || !ext || !arg || !out ||
        rows <= 0)
        return SN2_EINVAL;
    if (training && (!blk->stat_slots || nslots < 1)) return SN2_EINVAL;
    const long n_out = rows * blk->cout;
    // ~8 elements per thread; every workgroup of a TRAINING pass reads all the slots again (nslots x 2 cout floats: keep that
    // to a few MB of L2 reads in total), an eval pass reads none: the grid follows the rows (parcel inference: 20 M elements)
    int grid = sn2_cdiv(n_out, 1024 * 8);
    long cap = 4096;
    if (training) {
        const long per_wg = (long)nslots * 2 * blk->cout * 4;
        cap = (8L << 20) / (per_wg > 0 ? per_wg : 1);
        cap = cap < 32 ? 32 : (cap > 256 ? 256 : cap);
    }
    grid = grid < 1 ? 1 : (grid > cap ? (int)cap : grid);
    hipLaunchKernelGGL(bn_finalize_kernel<true>, dim3(grid), dim3(1024), 0, st, blk->cout, blk->gamma, blk->beta, blk->running_mean,
                       blk->running_var, blk->a, blk->c, blk->mean, blk->invstd, (const float*)blk->stat_slots, nslots,
                       count_dev, count_imm, training, blk->num_batches_tracked, ext, (const int*)arg, out, n_out);
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
        const float* hb = h + (size_t)b * R * hs + ch;
        int r = g;
        for (; r + 7 * G < R; r += 8 * G) {          // eight independent row loads in flight (one at a time: 18 us of latency)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = hb[(size_t)(r + u * G) * hs];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float y = fmaf(aa, v[u], cc);
                if (y > best) {
                    best = y;
                    bi = r + u * G;
                }
            }
        }
        for (; r < R; r += G) {
            const float v = fmaf(aa, hb[(size_t)r * hs], cc);
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
// The backward of the global level's pool in ONE launch (round 5; it was interp_gather_long_kernel + plot_max_bwd_kernel +
// fp_bwd_bn_small_kernel<64>: 15 us of three launches for 1 MB):
//   dx[b][c] += sum_r du[b R + r][c]      the transpose of knn_interpolate with k = 1 from the plot's ONE source (all weights 1,
//                                         model/point_net2.py:137 with :41's single position), rows added in a fixed order;
//   dy[b R + arg[b][c]][c] = dx[b][c]     the backward of global_max_pool (:39); dy zero elsewhere (the caller's zero fill);
//   dbeta[c] += sum_b dx[b][c],  dgamma[c] += sum_b dx[b][c] (h[b R + arg[b][c]][c] - mean[c]) invstd[c]
//                                         the BatchNorm sums of the block whose pre-BatchNorm rows h are: dy has B C non-zeros,
//                                         so the sums over its B R rows are sums over B terms per channel.
// One workgroup of 16 waves per plot: thread = (row group, channel), eight row loads in flight.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void global_pool_bwd_kernel(const float* __restrict__ du, int du_stride,
                                                               const int* __restrict__ arg, const float* __restrict__ h,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               int R, float* __restrict__ dx, float* __restrict__ dy,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float s_part[16][64];
    const int b = blockIdx.x, ch = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const float* db = du + (size_t)b * R * du_stride + ch;
    float a0 = 0.f, a1 = 0.f;
    int r = rg;
    for (; r + 7 * 16 < R; r += 8 * 16) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = db[(size_t)(r + u * 16) * du_stride];
        a0 += (v[0] + v[1]) + (v[2] + v[3]);
        a1 += (v[4] + v[5]) + (v[6] + v[7]);
    }
    for (; r < R; r += 16) a0 += db[(size_t)r * du_stride];
    s_part[rg][ch] = a0 + a1;
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += s_part[k][ch];
        const size_t i = (size_t)b * 64 + ch;
        const float g = dx[i] + t;
        dx[i] = g;
        const int ra = arg[i];
        if (ra >= 0 && ra < R) {
            const size_t row = (size_t)b * R + ra;
            dy[row * 64 + ch] = g;
            if (g != 0.f) {
                atomicAdd(&dbeta[ch], g);
                atomicAdd(&dgamma[ch], g * ((h[row * 64 + ch] - mean[ch]) * invstd[ch]));
            }
        }
    }
}

extern "C" int sn2_global_pool_backward(const float* du, int du_stride, const int* arg, const float* h, const float* mean,
                                        const float* invstd, int B, int R_per_plot, int C, float* dx, float* dy, float* dgamma,
                                        float* dbeta, void* stream) {
    if (!du || !arg || !h || !mean || !invstd || !dx || !dy || !dgamma || !dbeta || B <= 0 || R_per_plot <= 0 || du_stride < C)
        return SN2_EINVAL;
    if (C != 64) return SN2_ELIMIT;
    hipLaunchKernelGGL(global_pool_bwd_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, du, du_stride, arg, h, mean, invstd,
                       R_per_plot, dx, dy, dgamma, dbeta);
    SN2_RETURN_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// torch.optim.Adam (amsgrad=False, L2 weight decay folded into the gradient) on flat fp32 buffers.
// ------------------------------------------------------------------------------------------------------------
// The step counter lives on the device, so a captured hipGraph of the whole training step replays with the right bias
// corrections.  step[0] = steps taken so far, step[1] = arrival ticket (zero between launches): every workgroup reads
// step[0] before it takes a ticket, and the workgroup that takes the last one stores the incremented count -- one launch
// (a one-thread "tick" kernel in front of this one was 4 us of every step).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int n, float lr, float b1, float b2, float eps,
                                                   float wd, int* __restrict__ step, float gscale) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int now = __hip_atomic_load(&step[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    if (i < n) {
        const float t = (float)now;
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
    __syncthreads();                                               // every wave of this workgroup has its count
    if (threadIdx.x == 0) {
        const int ticket = atomicAdd(&step[1], 1);
        if (ticket == (int)gridDim.x - 1) {                        // all workgroups have read step[0]
            __hip_atomic_store(&step[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&step[0], now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// The same step with the gradient still spread over `replicas` images (sn2_block.grad_replicas): folds them exactly as
// grad_reduce_kernel does (same order of additions), writes the folded gradient back to image 0 -- the parameters' .grad views
// are right after the step -- and updates.  One launch instead of sn2_grad_reduce + sn2_adam_step where nothing (no exchange
// between ranks) needs the folded gradient in between.
__global__ __launch_bounds__(256) void adam_images_kernel(float* __restrict__ p, float* __restrict__ flat, int replicas, int stride,
                                                          float* __restrict__ m, float* __restrict__ v, int n, float lr, float b1,
                                                          float b2, float eps, float wd, int* __restrict__ step, float gscale) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int now = __hip_atomic_load(&step[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    if (i < n) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int r = 1;
        for (; r + 4 <= replicas; r += 4) {
            s0 += flat[(size_t)r * stride + i];
            s1 += flat[(size_t)(r + 1) * stride + i];
            s2 += flat[(size_t)(r + 2) * stride + i];
            s3 += flat[(size_t)(r + 3) * stride + i];
        }
        for (; r < replicas; ++r) s0 += flat[(size_t)r * stride + i];
        const float gsum = flat[i] + ((s0 + s1) + (s2 + s3));
        flat[i] = gsum;
        const float t = (float)now;
        const float bc1 = 1.f - powf(b1, t);
        const float bc2_sqrt = sqrtf(1.f - powf(b2, t));
        float grad = gsum * gscale;
        const float pi = p[i];
        grad = fmaf(wd, pi, grad);
        const float mi = m[i] + (1.f - b1) * (grad - m[i]);
        const float vi = b2 * v[i] + (1.f - b2) * grad * grad;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int ticket = atomicAdd(&step[1], 1);
        if (ticket == (int)gridDim.x - 1) {
            __hip_atomic_store(&step[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&step[0], now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

extern "C" int sn2_adam_step_images(float* param, float* grad_images, int replicas, int stride, float* exp_avg, float* exp_avg_sq,
                                    int n, float lr, float beta1, float beta2, float eps, float weight_decay, int* step_dev,
                                    float grad_scale, void* stream) {
    if (!param || !grad_images || !exp_avg || !exp_avg_sq || !step_dev || n <= 0 || replicas < 1 || (replicas > 1 && stride < n))
        return SN2_EINVAL;
    hipLaunchKernelGGL(adam_images_kernel, dim3(sn2_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, param, grad_images, replicas,
                       stride, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step_dev, grad_scale);
    SN2_RETURN_LAUNCH();
}

extern "C" int sn2_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int* step_dev, float grad_scale,
                             void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !step_dev || n <= 0) return SN2_EINVAL;
    hipLaunchKernelGGL(adam_kernel, dim3(sn2_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step_dev, grad_scale);
    SN2_RETURN_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// Input pipeline of a batch: load_cloud of the reference DataLoader (/root/reference/data_loader/loader.py:73-87) --
// center_cloud :127-132, add_fake_empty_ground_points :90-105, augment :161-214, rescale_cloud :135-158 and the gather of
// sample_cloud_data :233-255 -- for all plots of a batch in one kernel, one thread per OUTPUT point.  The random draws
// (angle, flips, subsample indices, noise) are inputs.  Arithmetic follows numpy 1.21 (the reference's pin): a float32
// array combined with a python/np.float64 scalar stays float32; the rotation is a float64 product cast back to float32.
// Row order of a raw point: x, y, z, red, green, blue, near_infrared, intensity, return_num, num_returns (config.py:66).
// ------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void prepare_plots_kernel(const float* __restrict__ raw, long T, const int* __restrict__ offs,
                                                            const float* __restrict__ centers, const float* __restrict__ fake_xy,
                                                            int n_fake, const int* __restrict__ idx, int B, int N, int train,
                                                            const double* __restrict__ rot, const int* __restrict__ flips,
                                                            const float* __restrict__ noise, const long* __restrict__ noise_offs,
                                                            long noise_T, float z_max, float* __restrict__ cloud,
                                                            float* __restrict__ xyz) {
#pragma clang fp contract(off)
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    if (o >= (long)B * N) return;
    const int b = (int)(o / N), n = (int)(o - (long)b * N);
    const int lo = offs[b], n_raw = offs[b + 1] - lo;
    const int src = idx[o];
    float v[10];
    if (src < n_raw) {
#pragma unroll
        for (int c = 0; c < 10; ++c) v[c] = raw[(size_t)c * T + lo + src];
        v[0] = v[0] - centers[2 * b];                     // center_cloud: before the fake points are appended
        v[1] = v[1] - centers[2 * b + 1];
    } else {
        const int k = src - n_raw;                        // fake ground point k: position only, features 0
        v[0] = k < n_fake ? fake_xy[2 * k] : 0.f;
        v[1] = k < n_fake ? fake_xy[2 * k + 1] : 0.f;
#pragma unroll
        for (int c = 2; c < 10; ++c) v[c] = 0.f;
    }
    float px = v[0], py = v[1];
    const float pz = v[2];                                // xyz = cloud[:3].copy(): metres, never rescaled
    if (train) {
        const double cs = rot[2 * b], sn = rot[2 * b + 1];    // np.dot(cloud[:2].T, [[c,-s],[s,c]]).T in float64
        const double rx = (double)v[0] * cs + (double)v[1] * sn, ry = (double)v[0] * (-sn) + (double)v[1] * cs;
        v[0] = px = (float)rx;
        v[1] = py = (float)ry;
        if (flips[2 * b]) { v[0] = -v[0]; px = -px; }
        if (flips[2 * b + 1]) { v[1] = -v[1]; py = -py; }
        if (noise) {                                      // noise rows: x, y, red, green, blue, nir; columns: the plot's points
            const long q = noise_offs[b] + src;           // BEFORE subsampling (duplicates of a point share their noise)
            v[0] += noise[q];
            v[1] += noise[noise_T + q];
#pragma unroll
            for (int c = 0; c < 4; ++c) v[3 + c] += noise[(2 + c) * noise_T + q];
        }
    }
    v[0] = v[0] / 10.f;                                   // rescale_cloud
    v[1] = v[1] / 10.f;
    v[2] = v[2] / z_max;
#pragma unroll
    for (int c = 3; c < 7; ++c) v[c] = v[c] / 65536.f;
    v[7] = v[7] / 32768.f;
    v[8] = (v[8] - 1.f) / 6.f;
    v[9] = (v[9] - 1.f) / 6.f;
#pragma unroll
    for (int c = 0; c < 10; ++c) cloud[((size_t)b * 10 + c) * N + n] = v[c];
    xyz[((size_t)b * 3 + 0) * N + n] = px;
    xyz[((size_t)b * 3 + 1) * N + n] = py;
    xyz[((size_t)b * 3 + 2) * N + n] = pz;
}
}  // namespace

extern "C" int sn2_prepare_plots(const float* raw, long T, const int* offsets, const float* centers, const float* fake_xy,
                                 int n_fake, const int* idx, int B, int N, int train, const double* rot, const int* flips,
                                 const float* noise, const long* noise_offsets, long noise_T, float z_max, float* cloud,
                                 float* xyz, void* stream) {
    if (!raw || !offsets || !centers || !idx || !cloud || !xyz || B <= 0 || N <= 0 || T <= 0 || !(z_max > 0.f)) return SN2_EINVAL;
    if (n_fake > 0 && !fake_xy) return SN2_EINVAL;
    if (train && (!rot || !flips)) return SN2_EINVAL;
    if (noise && !noise_offsets) return SN2_EINVAL;
    hipLaunchKernelGGL(prepare_plots_kernel, dim3(sn2_cdiv((long)B * N, 256)), dim3(256), 0, (hipStream_t)stream, raw, T, offsets,
                       centers, fake_xy, n_fake, idx, B, N, train, rot, flips, noise, noise_offsets, noise_T, z_max, cloud, xyz);
    SN2_RETURN_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// Regression guard for mlp.h's rule "one MFMA shape per accumulation chain".  D (16 x 16) = A (16 x 48) B (48 x 16) with
// bfloat16 operands, K = 32 through v_mfma_f32_16x16x32_bf16 and the tail K = 16 through v_mfma_f32_16x16x16_bf16, the second
// instruction taking the first one's result as its accumulator:
//   mode 0  the two builtins as the compiler schedules them HERE (it happens to put seven wait states between them: right);
//   mode 1  the same with 16 wait states forced between them: right;
//   mode 2  contract<true, 12> (mlp.h: K = 48 as two K = 32 instructions, the tail padded with zeros): what the library does;
//   mode 3  the distance the compiler chose inside sa_mfma_bwd_kernel (below): three vector instructions, no s_nop, the K = 16
//           instruction reading the K = 32 result as SrcC into another vDst.  The 8-pass K = 32 form (new on gfx950) has not
//           written all four result registers by then: rows 4q, 4q+1 of the tile lack the first product;
//   mode 4  mode 3 with 16 wait states: right.
// tests/test_gpu_bf16.py holds modes 0, 1, 2 and 4 to the product of the rounded operands and reports mode 3.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void debug_mfma_chain_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              float* __restrict__ d, int mode) {
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (mode == 2) {
        acc = contract<true, 12>(acc, [&](int kb) { return a[r * 48 + 4 * kb + q]; }, [&](int kb) { return b[(4 * kb + q) * 16 + r]; });
    } else {
        bf16x8 av, bv;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            av[t] = (__bf16)a[r * 48 + q * 8 + t];
            bv[t] = (__bf16)b[(q * 8 + t) * 16 + r];
        }
        bf16x4 at, bt;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            at[t] = (__bf16)a[r * 48 + 32 + q * 4 + t];
            bt[t] = (__bf16)b[(32 + q * 4 + t) * 16 + r];
        }
        if (mode >= 3) {
            // the sequence hipcc emitted inside sa_mfma_bwd_kernel when contract<> still mixed the shapes (ROCm 7.2, -O3):
            //     v_mfma_f32_16x16x32_bf16 a[32:35], v[246:249], v[238:241], a[0:3]
            //     v_and_b32_e32 v38, 0xffff, v55 ; v_ashrrev_i32_e32 v57, 31, v56 ; v_lshl_add_u64 v[56:57], s[94:95], 0, v[56:57]
            //     v_mfma_f32_16x16x16_bf16 a[36:39], v[40:41], v[38:39], a[32:35]
            // i.e. the K = 16 instruction reads the K = 32 result as SrcC (another vDst) THREE vector instructions later and
            // no s_nop.  Mode 3 replays exactly that distance, mode 4 the same with 16 wait states in between.
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const u32x4 a8 = __builtin_bit_cast(u32x4, av), b8 = __builtin_bit_cast(u32x4, bv);
            const u32x2 a4 = __builtin_bit_cast(u32x2, at), b4 = __builtin_bit_cast(u32x2, bt);
            f32x4 t, o;
            unsigned dummy = lane;
            if (mode == 3)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %3, %4, 0\n\t"
                             "v_and_b32 %2, 0xffff, %2\n\tv_ashrrev_i32 %2, 1, %2\n\tv_add_u32 %2, 1, %2\n\t"
                             "v_mfma_f32_16x16x16_bf16 %1, %5, %6, %0\n\t"
                             "s_nop 7\n\ts_nop 7"
                             : "=&v"(t), "=&v"(o), "+v"(dummy) : "v"(a8), "v"(b8), "v"(a4), "v"(b4));
            else
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %3, %4, 0\n\t"
                             "v_and_b32 %2, 0xffff, %2\n\tv_ashrrev_i32 %2, 1, %2\n\tv_add_u32 %2, 1, %2\n\t"
                             "s_nop 7\n\ts_nop 7\n\t"
                             "v_mfma_f32_16x16x16_bf16 %1, %5, %6, %0\n\t"
                             "s_nop 7\n\ts_nop 7"
                             : "=&v"(t), "=&v"(o), "+v"(dummy) : "v"(a8), "v"(b8), "v"(a4), "v"(b4));
            acc = o;
            if (dummy == 0xFFFFFFFFu) acc[0] = 0.f;    // (keeps the filler instructions' register alive)
        } else {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc, 0, 0, 0);
            if (mode == 1) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc));
            acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, at), __builtin_bit_cast(s16x4, bt), acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) d[(4 * q + j) * 16 + r] = acc[j];
}

extern "C" int sn2_debug_mfma_chain(const float* a, const float* b, float* d, int mode, void* stream) {
    if (!a || !b || !d || mode < 0 || mode > 4) return SN2_EINVAL;
    hipLaunchKernelGGL(debug_mfma_chain_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, d, mode);
    SN2_RETURN_LAUNCH();
}

// ---- diagnostic: `blocks` one-wave workgroups that do nothing but wait for `clocks` shader clocks (bounded: every wave leaves
// when its own clock count is up).  scripts/debug_contention.py uses it to tell what a concurrent kernel costs the feature
// pass by merely BEING there from what it costs by the CUs, caches and memory it uses.
namespace {
__global__ __launch_bounds__(64) void debug_spin_kernel(long long clocks, int* __restrict__ out) {
    const long long t0 = __builtin_readcyclecounter();
    int n = 0;
    while ((long long)__builtin_readcyclecounter() - t0 < clocks) {
        __builtin_amdgcn_s_sleep(8);
        ++n;
    }
    if (out && threadIdx.x == 0 && blockIdx.x == 0) *out = n;
}
}  // namespace

extern "C" int sn2_debug_spin(int blocks, long long clocks, int* out, void* stream) {
    if (blocks <= 0 || blocks > 4096 || clocks < 0 || clocks > (1LL << 34)) return SN2_EINVAL;
    hipLaunchKernelGGL(debug_spin_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, clocks, out);
    SN2_RETURN_LAUNCH();
}

extern "C" int sn2_version(void) { return SN2_VERSION; }

// ---- measured peaks (bench.py: "peaks measured in the same run", SURVEY.md 8d / BASELINE.md 3.5b) ------------------------
// mode 0: stream COPY dst[i] = src[i] (bytes moved = 2 n), mode 1: stream READ (sum into one word per workgroup; bytes = n):
// 16 bytes per lane and load, four loads in flight per lane, grid = 8 workgroups per CU, grid-stride.
namespace {
__global__ __launch_bounds__(256) void stream_probe_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4,
                                                           int mode, float* __restrict__ sink) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        if (mode == 0) {
            dst[i] = a, dst[i + stride] = b, dst[i + 2 * stride] = c, dst[i + 3 * stride] = d;
        } else {
            acc += (a.x + b.y) + (c.z + d.w);
        }
    }
    for (; i < n4; i += stride) {
        const float4 a = src[i];
        if (mode == 0) dst[i] = a;
        else acc += a.x;
    }
    if (mode != 0 && acc == 123456.789f) sink[blockIdx.x] = acc;        // (keeps the loads alive)
}

// MODE 0: v_mfma_f32_16x16x4_f32 (2048 flops per wave-instruction), 1: v_mfma_f32_32x32x2_f32 (4096), 2: v_mfma_f32_16x16x32_bf16
// (16 384), 3: v_mfma_f32_32x32x16_bf16 (32 768).  Eight independent accumulator chains per wave, `iters` rounds.
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256) void mfma_probe_kernel(int iters, float* __restrict__ sink) {
    const float av = 1.0f + (threadIdx.x & 7) * 0.125f, bv = 0.5f + (threadIdx.x & 3) * 0.25f;
    float total = 0.f;
    if constexpr (MODE == 0 || MODE == 2) {
        f32x4 acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 a8, b8;
#pragma unroll
        for (int j = 0; j < 8; ++j) a8[j] = (__bf16)av, b8[j] = (__bf16)bv;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if constexpr (MODE == 0) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[k], 0, 0, 0);
                else acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[k], 0, 0, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) total += acc[k][0] + acc[k][3];
    } else {
        f32x16 acc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[k][j] = 0.f;
        bf16x8 a8, b8;
#pragma unroll
        for (int j = 0; j < 8; ++j) a8[j] = (__bf16)av, b8[j] = (__bf16)bv;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if constexpr (MODE == 1) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[k], 0, 0, 0);
                else acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[k], 0, 0, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) total += acc[k][0] + acc[k][15];
    }
    if (total == 123456.789f) sink[blockIdx.x] = total;                 // (keeps the chains alive)
}
}  // namespace

extern "C" int sn2_debug_stream_probe(const float* src, float* dst, size_t n_floats, int mode, float* sink, void* stream) {
    if (!src || !sink || n_floats < 4 || (n_floats & 3) || (mode != 0 && mode != 1) || (mode == 0 && !dst)) return SN2_EINVAL;
    hipLaunchKernelGGL(stream_probe_kernel, dim3(8 * sn2_cu_count()), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), n_floats / 4, mode, sink);
    SN2_RETURN_LAUNCH();
}

// -> *flops (host, may be NULL) = the flops the launch performs (8 workgroups of 4 waves per CU)
extern "C" int sn2_debug_mfma_probe(int mode, int iters, float* sink, double* flops, void* stream) {
    if (!sink || iters <= 0 || iters > (1 << 22) || mode < 0 || mode > 3) return SN2_EINVAL;
    const int grid = 8 * sn2_cu_count();
    hipStream_t st = (hipStream_t)stream;
    const double per_instr[4] = {2048.0, 4096.0, 16384.0, 32768.0};
    const int chains[4] = {8, 4, 8, 4};
    if (flops) *flops = (double)grid * 4.0 * iters * chains[mode] * per_instr[mode];
    if (mode == 0) hipLaunchKernelGGL(mfma_probe_kernel<0>, dim3(grid), dim3(256), 0, st, iters, sink);
    else if (mode == 1) hipLaunchKernelGGL(mfma_probe_kernel<1>, dim3(grid), dim3(256), 0, st, iters, sink);
    else if (mode == 2) hipLaunchKernelGGL(mfma_probe_kernel<2>, dim3(grid), dim3(256), 0, st, iters, sink);
    else hipLaunchKernelGGL(mfma_probe_kernel<3>, dim3(grid), dim3(256), 0, st, iters, sink);
    SN2_RETURN_LAUNCH();
}


// ---- images of a flat gradient vector -> image 0 (sn2_block.grad_replicas) ---------------------------------------
namespace {
__global__ __launch_bounds__(256) void grad_reduce_kernel(float* __restrict__ flat, int n, int replicas, int stride) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = 1;
    for (; r + 4 <= replicas; r += 4) {                   // four independent loads in flight
        s0 += flat[(size_t)r * stride + i];
        s1 += flat[(size_t)(r + 1) * stride + i];
        s2 += flat[(size_t)(r + 2) * stride + i];
        s3 += flat[(size_t)(r + 3) * stride + i];
    }
    for (; r < replicas; ++r) s0 += flat[(size_t)r * stride + i];
    flat[i] += (s0 + s1) + (s2 + s3);
}
}  // namespace

extern "C" int sn2_grad_reduce(float* flat, int n, int replicas, int stride, void* stream) {
    if (!flat || n <= 0 || replicas < 1 || (replicas > 1 && stride < n)) return SN2_EINVAL;
    if (replicas == 1) return 0;
    hipLaunchKernelGGL(grad_reduce_kernel, dim3(sn2_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, flat, n, replicas, stride);
    SN2_RETURN_LAUNCH();
}
