// project.hip -- scatter-max projections of pointwise coverages onto 2D rasters.
// Replaces /root/reference/model/project_to_2d.py: project_to_plotwise_coverages (:7-55, torch.unique + torch_scatter
// scatter_max/scatter_mean with per-plot device<->CPU bounces) and project_to_2d_rasters (:58-113, a python loop over
// pixels).  Both become: pixel id per point (index arithmetic operation-for-operation in fp32 -- the ids must be
// bit-exact), a 64-bit max per (pixel, channel) over the key (ordered(value) << 32 | ~point), i.e. largest value, FIRST
// point on ties as torch_scatter's CPU loop -- LDS atomics per workgroup slice, then one global atomic per touched
// pixel -- and a tiny per-plot finalisation.  HBM-bound: 8 B (xy) + 16 B (coverages) read per point.
#include "common.h"

namespace {

constexpr int MAX_CELLS = 2025;  // diam_pix <= 45: the D*D*3 keys of a workgroup fit the default 48 KiB dynamic LDS

// project_to_2d.py:16-22   floor((xy - min) / (max - min + 0.0001) * diam_pix).int()
__device__ __forceinline__ int p2_pix(float v, float mn, float mx, int D) {
#pragma clang fp contract(off)
    const float num = v - mn;
    const float den = (mx - mn) + 0.0001f;
    const float q = num / den;
    const float s = q * (float)D;
    int i = (int)floorf(s);
    return i < 0 ? 0 : (i > D - 1 ? D - 1 : i);
}

// project_to_2d.py:68-78   clip(floor((xy + 0.0001) * scaling_factor + diam_meters // 2).int(), 0, diam_pix - 1)
__device__ __forceinline__ int p1_pix(float v, float sf, float off, int D) {
#pragma clang fp contract(off)
    const float t = v + 0.0001f;
    const float m = t * sf;
    const float s = m + off;
    int i = (int)floorf(s);
    return i < 0 ? 0 : (i > D - 1 ? D - 1 : i);
}

// also clears the plot's (pixel, channel) key table for the scatter kernel that follows (see sn2_fill_words in common.h
// for why this is not a hipMemsetAsync)
__global__ __launch_bounds__(1024) void plot_minmax_kernel(const float* __restrict__ xy, long plot_stride, int N,
                                                           float* __restrict__ mm, unsigned long long* __restrict__ keys,
                                                           int ncell3) {
    __shared__ float s[4][16];
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < ncell3; i += 1024) keys[(size_t)b * ncell3 + i] = 0ull;
    const float* x = xy + (size_t)b * plot_stride;
    const float* y = x + N;
    float xmn = INFINITY, xmx = -INFINITY, ymn = INFINITY, ymx = -INFINITY;
    int i = threadIdx.x;
    for (; i + 7 * 1024 < N; i += 8 * 1024) {        // sixteen independent loads in flight
        float a[8], c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = x[i + u * 1024], c[u] = y[i + u * 1024];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            xmn = fminf(xmn, a[u]);
            xmx = fmaxf(xmx, a[u]);
            ymn = fminf(ymn, c[u]);
            ymx = fmaxf(ymx, c[u]);
        }
    }
    for (; i < N; i += 1024) {
        const float a = x[i], c = y[i];
        xmn = fminf(xmn, a);
        xmx = fmaxf(xmx, a);
        ymn = fminf(ymn, c);
        ymx = fmaxf(ymx, c);
    }
    xmn = wave_min(xmn);
    xmx = wave_max(xmx);
    ymn = wave_min(ymn);
    ymx = wave_max(ymx);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s[0][w] = xmn;
        s[1][w] = xmx;
        s[2][w] = ymn;
        s[3][w] = ymx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 16; ++k) {
            xmn = fminf(xmn, s[0][k]);
            xmx = fmaxf(xmx, s[1][k]);
            ymn = fminf(ymn, s[2][k]);
            ymx = fmaxf(ymx, s[3][k]);
        }
        mm[b * 4 + 0] = xmn;
        mm[b * 4 + 1] = xmx;
        mm[b * 4 + 2] = ymn;
        mm[b * 4 + 3] = ymx;
    }
}

// MODE 0: P2 grid (bbox-normalised, cell = x_pix*D + y_pix).  MODE 1: P1 grid (fixed, cell = y_pix*D + x_pix).
template <int MODE>
__global__ __launch_bounds__(1024) void scatter_max_kernel(const float* __restrict__ vals, const float* __restrict__ xy,
                                                           long plot_stride, int N, int D, const float* __restrict__ mm,
                                                           float sf, float off, unsigned long long* __restrict__ keys,
                                                           int* __restrict__ pix) {
    extern __shared__ unsigned long long s_keys[];  // D*D*3
    const int b = blockIdx.y;
    const int ncell3 = D * D * 3;
    for (int i = threadIdx.x; i < ncell3; i += 1024) s_keys[i] = 0ull;
    __syncthreads();
    const float* x = xy + (size_t)b * plot_stride;
    const float* y = x + N;
    float xmn = 0.f, xmx = 0.f, ymn = 0.f, ymx = 0.f;
    if (MODE == 0) {
        xmn = mm[b * 4 + 0];
        xmx = mm[b * 4 + 1];
        ymn = mm[b * 4 + 2];
        ymx = mm[b * 4 + 3];
    }
    const int per = (N + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = min(N, lo + per);
    for (int n = lo + threadIdx.x; n < hi; n += 1024) {
        int cell;
        if (MODE == 0) {
            cell = p2_pix(x[n], xmn, xmx, D) * D + p2_pix(y[n], ymn, ymx, D);
        } else {
            cell = p1_pix(y[n], sf, off, D) * D + p1_pix(x[n], sf, off, D);
        }
        pix[(size_t)b * N + n] = cell;
        const float4 v = reinterpret_cast<const float4*>(vals)[(size_t)b * N + n];
        const unsigned long long tag = (unsigned long long)(0xFFFFFFFFu - (unsigned)n);
        atomicMax(&s_keys[cell * 3 + 0], ((unsigned long long)f2ord(v.x) << 32) | tag);
        atomicMax(&s_keys[cell * 3 + 1], ((unsigned long long)f2ord(v.z) << 32) | tag);
        atomicMax(&s_keys[cell * 3 + 2], ((unsigned long long)f2ord(v.w) << 32) | tag);
    }
    __syncthreads();
    unsigned long long* g = keys + (size_t)b * ncell3;
    for (int i = threadIdx.x; i < ncell3; i += 1024) {
        const unsigned long long k = s_keys[i];
        if (k) atomicMax(&g[i], k);
    }
}

// The pixel id of every point depends on the plot's x, y only (project_to_2d.py:16-22): a training loop that runs its
// position-only kernels ahead of the feature pass computes them there (sn2_plot_pixels) and the feature pass keeps two
// launches: this scatter from the stored ids and the finalisation.  No key table to clear and no global atomics: the
// `gridDim.x` slices of a plot each write their own table, the finalisation takes the maximum over the slices (same keys,
// same winner: the maximum of maxima).
__global__ __launch_bounds__(1024) void plot_pixels_kernel(const float* __restrict__ xy, long plot_stride, int N, int D,
                                                           const float* __restrict__ mm, int* __restrict__ pix) {
    const int b = blockIdx.y;
    const float* x = xy + (size_t)b * plot_stride;
    const float* y = x + N;
    const float xmn = mm[b * 4 + 0], xmx = mm[b * 4 + 1], ymn = mm[b * 4 + 2], ymx = mm[b * 4 + 3];
    const int per = (N + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = min(N, lo + per);
    for (int n = lo + threadIdx.x; n < hi; n += 1024) pix[(size_t)b * N + n] = p2_pix(x[n], xmn, xmx, D) * D + p2_pix(y[n], ymn, ymx, D);
}

__global__ __launch_bounds__(1024) void scatter_max_pix_kernel(const float* __restrict__ vals, const int* __restrict__ pix, int N,
                                                               int D, unsigned long long* __restrict__ keys_part) {
    extern __shared__ unsigned long long s_keys[];  // D*D*3
    const int b = blockIdx.y;
    const int ncell3 = D * D * 3;
    for (int i = threadIdx.x; i < ncell3; i += 1024) s_keys[i] = 0ull;
    __syncthreads();
    const int per = (N + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = min(N, lo + per);
    for (int n = lo + threadIdx.x; n < hi; n += 1024) {
        const int cell = pix[(size_t)b * N + n];
        const float4 v = reinterpret_cast<const float4*>(vals)[(size_t)b * N + n];
        const unsigned long long tag = (unsigned long long)(0xFFFFFFFFu - (unsigned)n);
        atomicMax(&s_keys[cell * 3 + 0], ((unsigned long long)f2ord(v.x) << 32) | tag);
        atomicMax(&s_keys[cell * 3 + 1], ((unsigned long long)f2ord(v.z) << 32) | tag);
        atomicMax(&s_keys[cell * 3 + 2], ((unsigned long long)f2ord(v.w) << 32) | tag);
    }
    __syncthreads();
    unsigned long long* g = keys_part + ((size_t)b * gridDim.x + blockIdx.x) * ncell3;
    for (int i = threadIdx.x; i < ncell3; i += 1024) g[i] = s_keys[i];
}

// P2: pred[b] = mean over occupied pixels of [low, 1-low, med, high]   (project_to_2d.py:41-53)
// parts: key tables per plot (1 = the table the global atomics of scatter_max_kernel<0> filled; more: the slices of
// scatter_max_pix_kernel, of which the maximum is taken here)
__global__ __launch_bounds__(256) void p2_finalize_kernel(const unsigned long long* __restrict__ keys, int D, int parts,
                                                          int* __restrict__ arg, int* __restrict__ nocc,
                                                          float* __restrict__ pred) {
    __shared__ float s[5][4];
    const int b = blockIdx.x;
    const int ncell = D * D;
    float lo = 0.f, so = 0.f, me = 0.f, hi = 0.f, cnt = 0.f;
    for (int c = threadIdx.x; c < ncell; c += 256) {
        const size_t base = ((size_t)b * ncell + c) * 3;
        const unsigned long long* kp = keys + ((size_t)b * parts * ncell + c) * 3;
        unsigned long long k0 = kp[0], k1 = kp[1], k2 = kp[2];
        for (int q = 1; q < parts; ++q) {
            const unsigned long long* kq = kp + (size_t)q * ncell * 3;
            const unsigned long long a0 = kq[0], a1 = kq[1], a2 = kq[2];
            k0 = a0 > k0 ? a0 : k0;
            k1 = a1 > k1 ? a1 : k1;
            k2 = a2 > k2 ? a2 : k2;
        }
        if (k0) {
            const float l = ord2f((uint32_t)(k0 >> 32));
            lo += l;
            so += 1.0f - l;
            me += ord2f((uint32_t)(k1 >> 32));
            hi += ord2f((uint32_t)(k2 >> 32));
            cnt += 1.f;
            arg[base + 0] = (int)(0xFFFFFFFFu - (uint32_t)(k0 & 0xFFFFFFFFull));
            arg[base + 1] = (int)(0xFFFFFFFFu - (uint32_t)(k1 & 0xFFFFFFFFull));
            arg[base + 2] = (int)(0xFFFFFFFFu - (uint32_t)(k2 & 0xFFFFFFFFull));
        } else {
            arg[base + 0] = arg[base + 1] = arg[base + 2] = -1;
        }
    }
    lo = wave_sum(lo);
    so = wave_sum(so);
    me = wave_sum(me);
    hi = wave_sum(hi);
    cnt = wave_sum(cnt);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s[0][w] = lo;
        s[1][w] = so;
        s[2][w] = me;
        s[3][w] = hi;
        s[4][w] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t[5];
        for (int q = 0; q < 5; ++q) t[q] = (s[q][0] + s[q][1]) + (s[q][2] + s[q][3]);
        const float n = fmaxf(t[4], 1.f);
        pred[b * 4 + 0] = t[0] / n;
        pred[b * 4 + 1] = t[1] / n;
        pred[b * 4 + 2] = t[2] / n;
        pred[b * 4 + 3] = t[3] / n;
        nocc[b] = (int)t[4];
    }
}

__global__ void p2_backward_kernel(const float* __restrict__ dpred, const int* __restrict__ arg,
                                   const int* __restrict__ nocc, const int* __restrict__ pix, int B, int N, int D,
                                   float4* __restrict__ dpw) {
    // gather form: a point receives a pixel's gradient iff it is that pixel's arg-max, so every row of dpointwise is
    // written exactly once (no zero fill in front, no scatter): one launch
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * N) return;
    const int b = (int)(i / N), n = (int)(i - (size_t)b * N);
    const int* a = arg + ((size_t)b * D * D + pix[i]) * 3;
    const float inv = 1.0f / fmaxf((float)nocc[b], 1.f);
    const float4 g = reinterpret_cast<const float4*>(dpred)[b];
    float4 o;
    o.x = a[0] == n ? (g.x - g.y) * inv : 0.f;   // low vegetation also feeds bare soil = 1 - low
    o.y = 0.f;
    o.z = a[1] == n ? g.z * inv : 0.f;
    o.w = a[2] == n ? g.w * inv : 0.f;
    dpw[i] = o;
}

// P1: rasters (B,3,D,D) [low,med,high], image[y][x], NaN where empty, rows flipped (project_to_2d.py:80-113)
__global__ void p1_finalize_kernel(const unsigned long long* __restrict__ keys, int B, int D, float* __restrict__ rasters) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int ncell = D * D;
    if (i >= B * ncell * 3) return;
    const int b = i / (ncell * 3), rem = i - b * ncell * 3;
    const int cell = rem / 3, slot = rem - cell * 3;
    const int y = cell / D, x = cell - y * D;
    const unsigned long long k = keys[i];
    const float v = k ? ord2f((uint32_t)(k >> 32)) : __uint_as_float(0x7FC00000u);
    rasters[(((size_t)b * 3 + slot) * D + (D - 1 - y)) * D + x] = v;
}

int grid_slices(int N) {
    int s = sn2_cdiv(N, 4096);
    return s < 1 ? 1 : (s > 64 ? 64 : s);
}

}  // namespace

extern "C" int sn2_plot_project_forward(const float* pred_pointwise, const float* cloud_xy, long plot_stride, int B, int N,
                                        int D, unsigned long long* keys, int* pix, int* arg, int* nocc, float* pred,
                                        void* stream) {
    if (!pred_pointwise || !cloud_xy || !keys || !pix || !arg || !nocc || !pred || B <= 0 || N <= 0 || D <= 0)
        return SN2_EINVAL;
    if (D * D > MAX_CELLS || plot_stride < 2L * N) return SN2_ELIMIT;
    hipStream_t st = (hipStream_t)stream;
    // the first 4*B floats of `pred` double as the per-plot (xmin,xmax,ymin,ymax) scratch until the finalisation
    float* mm = pred;
    hipLaunchKernelGGL(plot_minmax_kernel, dim3(B), dim3(1024), 0, st, cloud_xy, plot_stride, N, mm, keys, D * D * 3);
    hipLaunchKernelGGL((scatter_max_kernel<0>), dim3(grid_slices(N), B), dim3(1024), (size_t)D * D * 3 * 8, st,
                       pred_pointwise, cloud_xy, plot_stride, N, D, (const float*)mm, 0.f, 0.f, keys, pix);
    hipLaunchKernelGGL(p2_finalize_kernel, dim3(B), dim3(256), 0, st, (const unsigned long long*)keys, D, 1, arg, nocc, pred);
    SN2_RETURN_LAUNCH();
}

extern "C" int sn2_plot_pixels(const float* cloud_xy, long plot_stride, int B, int N, int D, float* mm, int* pix, void* stream) {
    if (!cloud_xy || !mm || !pix || B <= 0 || N <= 0 || D <= 0) return SN2_EINVAL;
    if (D * D > MAX_CELLS || plot_stride < 2L * N) return SN2_ELIMIT;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(plot_minmax_kernel, dim3(B), dim3(1024), 0, st, cloud_xy, plot_stride, N, mm,
                       (unsigned long long*)nullptr, 0);
    hipLaunchKernelGGL(plot_pixels_kernel, dim3(grid_slices(N), B), dim3(1024), 0, st, cloud_xy, plot_stride, N, D,
                       (const float*)mm, pix);
    SN2_RETURN_LAUNCH();
}

extern "C" int sn2_plot_project_forward_pix(const float* pred_pointwise, const int* pix, int B, int N, int D,
                                            unsigned long long* keys, int* arg, int* nocc, float* pred, void* stream) {
    if (!pred_pointwise || !pix || !keys || !arg || !nocc || !pred || B <= 0 || N <= 0 || D <= 0) return SN2_EINVAL;
    if (D * D > MAX_CELLS) return SN2_ELIMIT;
    hipStream_t st = (hipStream_t)stream;
    const int parts = grid_slices(N);
    hipLaunchKernelGGL(scatter_max_pix_kernel, dim3(parts, B), dim3(1024), (size_t)D * D * 3 * 8, st, pred_pointwise, pix, N, D,
                       keys);
    hipLaunchKernelGGL(p2_finalize_kernel, dim3(B), dim3(256), 0, st, (const unsigned long long*)keys, D, parts, arg, nocc, pred);
    SN2_RETURN_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// Projection + loss of the training step in THREE launches instead of seven (round 5): learning/train.py:54-62,
//   pred = project_to_plotwise_coverages(coverages_pointwise, clouds, args)                       model/project_to_2d.py:7-55
//   loss = get_absolute_loss(pred, gt) + m get_NLL_loss(proba, pdf) + e get_entropy_loss(proba)   learning/loss_functions.py:9-57
// and their gradients.  The chip's floor for ANY launch is ~4.5 us (an empty kernel), the seven kernels were 4-7 us each:
//   launch 1  two independent jobs side by side: workgroups [0, parts B) scatter the coverages into per-slice key tables
//             (scatter_max_pix_kernel's body), the rest sum the pointwise loss terms (loss_point_kernel's body);
//   launch 2  per plot: maximum over the slices, arg-max points, mean over occupied pixels (p2_finalize_kernel's body) -- and
//             the workgroup that leaves last adds the loss up (loss_final_kernel's body); the B x 5 words it needs from its
//             peers travel as agent-scope atomics, complete before their ticket;
//   backward  one pass over the points: d loss / d proba (loss_bwd_kernel's body) and d loss / d coverages (p2_backward_kernel's,
//             with the B x 3 values of d loss / d pred it needs recomputed from pred and gt by every thread).
// Same arithmetic per element as the separate kernels: the same pred, arg, dcov, dproba bits; the loss sums are grouped by
// 1024 instead of 256 rows per workgroup (fp64: 1e-16).
// ------------------------------------------------------------------------------------------------------------
namespace {
constexpr int PL_LOSS_BLOCKS = 512;
constexpr float PL_EPS_F = 0.0001f;
constexpr double PL_EPS_D = 0.0001;

__device__ __forceinline__ double pl_wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <bool NLL, bool ENT>
__global__ __launch_bounds__(1024) void projected_loss_scatter_kernel(const float* __restrict__ vals, const int* __restrict__ pix,
                                                                     int B, int N, int D, int parts,
                                                                     unsigned long long* __restrict__ keys_part,
                                                                     const float4* __restrict__ proba, const double* __restrict__ pdf,
                                                                     int R, double* __restrict__ partials, int n_loss_blocks,
                                                                     unsigned* __restrict__ ticket) {
    extern __shared__ unsigned long long s_keys[];  // D*D*3 (the scatter workgroups)
    const int n_scatter = parts * B;
    if ((int)blockIdx.x < n_scatter) {
        const int b = blockIdx.x / parts, part = blockIdx.x - b * parts;
        const int ncell3 = D * D * 3;
        for (int i = threadIdx.x; i < ncell3; i += 1024) s_keys[i] = 0ull;
        __syncthreads();
        const int per = (N + parts - 1) / parts;
        const int lo = part * per, hi = min(N, lo + per);
        for (int n = lo + threadIdx.x; n < hi; n += 1024) {
            const int cell = pix[(size_t)b * N + n];
            const float4 v = reinterpret_cast<const float4*>(vals)[(size_t)b * N + n];
            const unsigned long long tag = (unsigned long long)(0xFFFFFFFFu - (unsigned)n);
            atomicMax(&s_keys[cell * 3 + 0], ((unsigned long long)f2ord(v.x) << 32) | tag);
            atomicMax(&s_keys[cell * 3 + 1], ((unsigned long long)f2ord(v.z) << 32) | tag);
            atomicMax(&s_keys[cell * 3 + 2], ((unsigned long long)f2ord(v.w) << 32) | tag);
        }
        __syncthreads();
        unsigned long long* g = keys_part + ((size_t)b * parts + part) * ncell3;
        for (int i = threadIdx.x; i < ncell3; i += 1024) g[i] = s_keys[i];
        return;
    }
    // ---- the pointwise loss terms: one row per lane, 16 B of probabilities + 24 B of densities
    const int lb = blockIdx.x - n_scatter;
    if (lb == 0 && threadIdx.x == 0) *ticket = 0u;                  // the finalisation's arrival counter (next launch)
    __shared__ double s_part[2][16];
    double nll = 0.0, ent = 0.0;
    for (int i = lb * 1024 + threadIdx.x; i < R; i += n_loss_blocks * 1024) {
        const float4 p = proba[i];
        if constexpr (NLL) {
            const double f0 = pdf[3 * (size_t)i], f1 = pdf[3 * (size_t)i + 1], f2 = pdf[3 * (size_t)i + 2];
            const float pg = p.x + p.y;
            const double lik = ((double)pg * f0 + (double)p.z * f1) + (double)p.w * f2;
            nll -= log(lik);
        }
        if constexpr (ENT) {
            const float e2 = p.z * logf(p.z + PL_EPS_F) + (1.f - p.z) * logf(1.f - p.z + PL_EPS_F);
            const float e3 = p.w * logf(p.w + PL_EPS_F) + (1.f - p.w) * logf(1.f - p.w + PL_EPS_F);
            ent -= (double)e2 + (double)e3;
        }
    }
    nll = pl_wave_sum_f64(nll);
    ent = pl_wave_sum_f64(ent);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) s_part[0][w] = nll, s_part[1][w] = ent;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, c = 0.0;
        for (int k = 0; k < 16; ++k) a += s_part[0][k], c += s_part[1][k];
        partials[2 * lb] = a;
        partials[2 * lb + 1] = c;
    }
}

__global__ __launch_bounds__(256) void projected_loss_finalize_kernel(const unsigned long long* __restrict__ keys, int B, int D,
                                                                      int parts, int* __restrict__ arg, int* __restrict__ nocc,
                                                                      float* __restrict__ pred, const double* __restrict__ gt,
                                                                      const double* __restrict__ partials, int n_loss_blocks, int R,
                                                                      double m, double e, unsigned* __restrict__ ticket,
                                                                      double* __restrict__ out) {
    __shared__ float s[5][4];
    __shared__ int s_last;
    __shared__ double s3[3][256];
    const int b = blockIdx.x;
    const int ncell = D * D;
    float lo = 0.f, so = 0.f, me = 0.f, hi = 0.f, cnt = 0.f;
    for (int c = threadIdx.x; c < ncell; c += 256) {
        const size_t base = ((size_t)b * ncell + c) * 3;
        const unsigned long long* kp = keys + ((size_t)b * parts * ncell + c) * 3;
        unsigned long long k0 = kp[0], k1 = kp[1], k2 = kp[2];
        for (int q = 1; q < parts; ++q) {
            const unsigned long long* kq = kp + (size_t)q * ncell * 3;
            const unsigned long long a0 = kq[0], a1 = kq[1], a2 = kq[2];
            k0 = a0 > k0 ? a0 : k0;
            k1 = a1 > k1 ? a1 : k1;
            k2 = a2 > k2 ? a2 : k2;
        }
        if (k0) {
            const float l = ord2f((uint32_t)(k0 >> 32));
            lo += l;
            so += 1.0f - l;
            me += ord2f((uint32_t)(k1 >> 32));
            hi += ord2f((uint32_t)(k2 >> 32));
            cnt += 1.f;
            arg[base + 0] = (int)(0xFFFFFFFFu - (uint32_t)(k0 & 0xFFFFFFFFull));
            arg[base + 1] = (int)(0xFFFFFFFFu - (uint32_t)(k1 & 0xFFFFFFFFull));
            arg[base + 2] = (int)(0xFFFFFFFFu - (uint32_t)(k2 & 0xFFFFFFFFull));
        } else {
            arg[base + 0] = arg[base + 1] = arg[base + 2] = -1;
        }
    }
    lo = wave_sum(lo);
    so = wave_sum(so);
    me = wave_sum(me);
    hi = wave_sum(hi);
    cnt = wave_sum(cnt);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) s[0][w] = lo, s[1][w] = so, s[2][w] = me, s[3][w] = hi, s[4][w] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t[5];
        for (int q = 0; q < 5; ++q) t[q] = (s[q][0] + s[q][1]) + (s[q][2] + s[q][3]);
        const float n = fmaxf(t[4], 1.f);
        // (agent-scope stores: the workgroup that adds the loss up reads them from another CU, maybe another XCD)
        for (int q = 0; q < 4; ++q)
            __hip_atomic_store(reinterpret_cast<unsigned*>(pred) + b * 4 + q, __float_as_uint(t[q] / n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(nocc + b, (int)t[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (atomicAdd(ticket, 1u) == (unsigned)B - 1u) ? 1 : 0;      // (the stores above are this thread's: program order)
    }
    __syncthreads();
    if (!s_last) return;
    // ---- the last workgroup out: the loss value (loss_final_kernel)
    double nll = 0.0, ent = 0.0, ab = 0.0;
    for (int i = threadIdx.x; i < n_loss_blocks; i += 256) nll += partials[2 * i], ent += partials[2 * i + 1];
    for (int i = threadIdx.x; i < 3 * B; i += 256) {
        const int bb = i / 3, c = i - 3 * bb, col = c == 0 ? 0 : c + 1;
        const float pv = __uint_as_float(__hip_atomic_load(reinterpret_cast<unsigned*>(pred) + 4 * bb + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const double d = (double)pv - gt[4 * bb + col];
        ab += sqrt(d * d + PL_EPS_D);
    }
    s3[0][threadIdx.x] = nll, s3[1][threadIdx.x] = ent, s3[2][threadIdx.x] = ab;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            s3[0][threadIdx.x] += s3[0][threadIdx.x + o];
            s3[1][threadIdx.x] += s3[1][threadIdx.x + o];
            s3[2][threadIdx.x] += s3[2][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double l_abs = s3[2][0] / (3.0 * B), l_nll = n_loss_blocks > 0 ? s3[0][0] / R : 0.0;
        const double l_ent = n_loss_blocks > 0 ? (double)(float)(s3[1][0] / (2.0 * R)) : 0.0;
        out[0] = l_abs + m * l_nll + e * l_ent;
        out[1] = l_abs;
        out[2] = l_nll;
        out[3] = l_ent;
    }
}

template <bool NLL, bool ENT>
__global__ __launch_bounds__(256) void projected_loss_bwd_kernel(const float* __restrict__ pred, const double* __restrict__ gt, int B,
                                                                 const float4* __restrict__ proba, const double* __restrict__ pdf,
                                                                 int N, int D, double m, double e, const double* __restrict__ gout,
                                                                 const int* __restrict__ arg, const int* __restrict__ nocc,
                                                                 const int* __restrict__ pix, float4* __restrict__ dcov,
                                                                 float4* __restrict__ dproba) {
    const size_t R = (size_t)B * N;
    const double g = gout[0];
    const double cn = g * m / (double)R, ce = g * e / (2.0 * (double)R);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < R; i += (size_t)gridDim.x * 256) {
        const float4 p = proba[i];
        double d0 = 0.0, d2 = 0.0, d3 = 0.0;
        if constexpr (NLL) {
            const double f0 = pdf[3 * i], f1 = pdf[3 * i + 1], f2 = pdf[3 * i + 2];
            const float pg = p.x + p.y;
            const double lik = ((double)pg * f0 + (double)p.z * f1) + (double)p.w * f2;
            const double il = -cn / lik;
            d0 = il * f0, d2 = il * f1, d3 = il * f2;
        }
        if constexpr (ENT) {
            const float h2 = -(logf(p.z + PL_EPS_F) + p.z / (p.z + PL_EPS_F) - logf(1.f - p.z + PL_EPS_F) - (1.f - p.z) / (1.f - p.z + PL_EPS_F));
            const float h3 = -(logf(p.w + PL_EPS_F) + p.w / (p.w + PL_EPS_F) - logf(1.f - p.w + PL_EPS_F) - (1.f - p.w) / (1.f - p.w + PL_EPS_F));
            d2 += ce * (double)h2, d3 += ce * (double)h3;
        }
        float4 d;
        d.x = d.y = (float)d0;
        d.z = (float)d2;
        d.w = (float)d3;
        dproba[i] = d;
        // d loss / d coverages: the point receives its pixel's gradient iff it is the pixel's arg-max (p2_backward_kernel), with
        // d loss / d pred of its plot recomputed here (loss_bwd_kernel's formula: column 1, bare soil, has no target)
        const int b = (int)(i / N), n = (int)(i - (size_t)b * N);
        const float4 pr = reinterpret_cast<const float4*>(pred)[b];
        const double e0 = (double)pr.x - gt[4 * b + 0], e2 = (double)pr.z - gt[4 * b + 2], e3 = (double)pr.w - gt[4 * b + 3];
        const float gx = (float)(g * e0 / sqrt(e0 * e0 + PL_EPS_D) / (3.0 * B));
        const float gz = (float)(g * e2 / sqrt(e2 * e2 + PL_EPS_D) / (3.0 * B));
        const float gw = (float)(g * e3 / sqrt(e3 * e3 + PL_EPS_D) / (3.0 * B));
        const int* a = arg + ((size_t)b * D * D + pix[i]) * 3;
        const float inv = 1.0f / fmaxf((float)nocc[b], 1.f);
        float4 o;
        o.x = a[0] == n ? (gx - 0.f) * inv : 0.f;      // (g.x - g.y) with g.y = 0: bare soil carries no gradient of its own
        o.y = 0.f;
        o.z = a[1] == n ? gz * inv : 0.f;
        o.w = a[2] == n ? gw * inv : 0.f;
        dcov[i] = o;
    }
}
}  // namespace

extern "C" int sn2_projected_loss_forward(const float* coverages, const int* pix, const float* proba, const double* pdf,
                                          const double* gt, int B, int N, int D, double m, double e, unsigned long long* keys,
                                          int* arg, int* nocc, float* pred, double* partials, double* out, void* stream) {
    if (!coverages || !pix || !proba || !gt || !keys || !arg || !nocc || !pred || !partials || !out || B <= 0 || N <= 0 || D <= 0)
        return SN2_EINVAL;
    if (D * D > MAX_CELLS || (long)B * N >= (1L << 31)) return SN2_ELIMIT;
    const bool nll = m != 0.0, ent = e != 0.0;
    if (nll && !pdf) return SN2_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int parts = grid_slices(N), R = B * N;
    int nlb = (nll || ent) ? sn2_cdiv(R, 1024) : 0;
    if (nlb > PL_LOSS_BLOCKS) nlb = PL_LOSS_BLOCKS;
    unsigned* ticket = reinterpret_cast<unsigned*>(partials + 2 * PL_LOSS_BLOCKS);
    const int grid = parts * B + (nlb > 0 ? nlb : 1);              // (at least one loss workgroup: it clears the ticket)
    const size_t lds = (size_t)D * D * 3 * 8;
    const float4* p4 = reinterpret_cast<const float4*>(proba);
    if (nll && ent) hipLaunchKernelGGL((projected_loss_scatter_kernel<true, true>), dim3(grid), dim3(1024), lds, st, coverages, pix, B, N, D, parts, keys, p4, pdf, nlb > 0 ? R : 0, partials, nlb > 0 ? nlb : 1, ticket);
    else if (nll) hipLaunchKernelGGL((projected_loss_scatter_kernel<true, false>), dim3(grid), dim3(1024), lds, st, coverages, pix, B, N, D, parts, keys, p4, pdf, nlb > 0 ? R : 0, partials, nlb > 0 ? nlb : 1, ticket);
    else if (ent) hipLaunchKernelGGL((projected_loss_scatter_kernel<false, true>), dim3(grid), dim3(1024), lds, st, coverages, pix, B, N, D, parts, keys, p4, pdf, nlb > 0 ? R : 0, partials, nlb > 0 ? nlb : 1, ticket);
    else hipLaunchKernelGGL((projected_loss_scatter_kernel<false, false>), dim3(grid), dim3(1024), lds, st, coverages, pix, B, N, D, parts, keys, p4, pdf, 0, partials, 1, ticket);
    hipLaunchKernelGGL(projected_loss_finalize_kernel, dim3(B), dim3(256), 0, st, (const unsigned long long*)keys, B, D, parts, arg, nocc,
                       pred, gt, (const double*)partials, nlb, R, m, e, ticket, out);
    SN2_RETURN_LAUNCH();
}

extern "C" int sn2_projected_loss_backward(const float* pred, const double* gt, int B, const float* proba, const double* pdf, int N,
                                           int D, double m, double e, const double* grad_total, const int* arg, const int* nocc,
                                           const int* pix, float* dcoverages, float* dproba, void* stream) {
    if (!pred || !gt || !proba || !grad_total || !arg || !nocc || !pix || !dcoverages || !dproba || B <= 0 || N <= 0 || D <= 0)
        return SN2_EINVAL;
    const bool nll = m != 0.0, ent = e != 0.0;
    if (nll && !pdf) return SN2_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    int grid = sn2_cdiv((long)B * N, 256);
    if (grid > 2048) grid = 2048;
    const float4* p4 = reinterpret_cast<const float4*>(proba);
    float4 *dc = reinterpret_cast<float4*>(dcoverages), *dp = reinterpret_cast<float4*>(dproba);
    if (nll && ent) hipLaunchKernelGGL((projected_loss_bwd_kernel<true, true>), dim3(grid), dim3(256), 0, st, pred, gt, B, p4, pdf, N, D, m, e, grad_total, arg, nocc, pix, dc, dp);
    else if (nll) hipLaunchKernelGGL((projected_loss_bwd_kernel<true, false>), dim3(grid), dim3(256), 0, st, pred, gt, B, p4, pdf, N, D, m, e, grad_total, arg, nocc, pix, dc, dp);
    else if (ent) hipLaunchKernelGGL((projected_loss_bwd_kernel<false, true>), dim3(grid), dim3(256), 0, st, pred, gt, B, p4, pdf, N, D, m, e, grad_total, arg, nocc, pix, dc, dp);
    else hipLaunchKernelGGL((projected_loss_bwd_kernel<false, false>), dim3(grid), dim3(256), 0, st, pred, gt, B, p4, pdf, N, D, m, e, grad_total, arg, nocc, pix, dc, dp);
    SN2_RETURN_LAUNCH();
}

extern "C" int sn2_plot_project_backward(const float* dpred, const int* arg, const int* nocc, const int* pix, int B, int N,
                                         int D, float* dpointwise, void* stream) {
    if (!dpred || !arg || !nocc || !pix || !dpointwise || B <= 0 || N <= 0 || D <= 0) return SN2_EINVAL;
    hipLaunchKernelGGL(p2_backward_kernel, dim3(sn2_cdiv((long)B * N, 256)), dim3(256), 0, (hipStream_t)stream, dpred, arg,
                       nocc, pix, B, N, D, reinterpret_cast<float4*>(dpointwise));
    SN2_RETURN_LAUNCH();
}

extern "C" int sn2_raster_project(const float* coverages, const float* cloud_xy, long plot_stride, int B, int N, int D,
                                  int diam_meters, unsigned long long* keys, int* pix, float* rasters, void* stream) {
    if (!coverages || !cloud_xy || !keys || !pix || !rasters || B <= 0 || N <= 0 || D <= 0 || diam_meters <= 0)
        return SN2_EINVAL;
    if (D * D > MAX_CELLS || plot_stride < 2L * N) return SN2_ELIMIT;
    hipStream_t st = (hipStream_t)stream;
    const float sf = (float)(10.0 * ((double)D / (double)diam_meters));  // python: 10 * (diam_pix / diam_meters)
    const float off = (float)(diam_meters / 2);                           // diam_meters // 2
    sn2_fill_words(keys, 0u, (size_t)B * D * D * 3 * 2, st);
    hipLaunchKernelGGL((scatter_max_kernel<1>), dim3(grid_slices(N), B), dim3(1024), (size_t)D * D * 3 * 8, st, coverages,
                       cloud_xy, plot_stride, N, D, (const float*)nullptr, sf, off, keys, pix);
    hipLaunchKernelGGL(p1_finalize_kernel, dim3(sn2_cdiv((long)B * D * D * 3, 256)), dim3(256), 0, st,
                       (const unsigned long long*)keys, B, D, rasters);
    SN2_RETURN_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// Parcel mosaic: fold the plots of a batch, in order, into the running (mean, weight-sum) rasters with the rule of the
// rasterio.merge callback (inference/geotiff_raster.py:294-347).  One thread per (band, parcel pixel) of the window; the
// plot offsets are wave-uniform (scalar loads), the raster reads are contiguous along x.
// ------------------------------------------------------------------------------------------------------------
namespace {
__global__ void mosaic_merge_kernel(const float* __restrict__ rasters, const float* __restrict__ weights,
                                    const int* __restrict__ offsets, int B, int D, int H, int W,
                                    float* __restrict__ mean, float* __restrict__ wsum, int y0, int x0, int wh, int ww) {
    // each product, sum and quotient rounded to fp32 on its own, as numpy does on the reference's float32 canvas (bit-exact
    // against tests/golden/f_mosaic.npz): no fused multiply-add
#pragma clang fp contract(off)
    const int wx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int wy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int band = blockIdx.z;
    if (wx >= ww || wy >= wh) return;
    const int gy = y0 + wy, gx = x0 + wx;
    if (gy < 0 || gy >= H || gx < 0 || gx >= W) return;
    const size_t o = ((size_t)band * H + gy) * W + gx;
    float V = mean[o], Wt = wsum[o];
    const float nan = __int_as_float(0x7fc00000);
    for (int b = 0; b < B; ++b) {
        const int y = gy - offsets[2 * b], x = gx - offsets[2 * b + 1];
        if (y < 0 || y >= D || x < 0 || x >= D) continue;     // pixel outside this plot's raster: the callback is not
                                                              // called on it
        const float v = rasters[(((size_t)b * 3 + band) * D + y) * D + x];
        const float w = weights[y * D + x];
        const bool on = V != V, nn = v != v, own = Wt != Wt, nwn = w != w;
        if (on && nn) {
            V = nan;
        } else {
            float a_old = V * Wt, a_new = v * w;               // NaN (no data) contributes nothing: np.nansum
            a_old = (on || a_old != a_old) ? 0.f : a_old;
            a_new = (nn || a_new != a_new) ? 0.f : a_new;
            const float w_old = (on || own) ? 0.f : Wt, w_new = (nn || nwn) ? 0.f : w;
            V = (a_old + a_new) / (w_old + w_new);
        }
        Wt = (own && nwn) ? nan : (own ? 0.f : Wt) + (nwn ? 0.f : w);
    }
    mean[o] = V;
    wsum[o] = Wt;
}
}  // namespace

extern "C" int sn2_mosaic_merge(const float* rasters, const float* weights, const int* offsets, int B, int D, int H, int W,
                                float* mean, float* wsum, int win_y0, int win_x0, int win_h, int win_w, void* stream) {
    if (!rasters || !weights || !offsets || !mean || !wsum || B <= 0 || D <= 0 || H <= 0 || W <= 0) return SN2_EINVAL;
    if (win_h <= 0 || win_w <= 0) return 0;
    hipLaunchKernelGGL(mosaic_merge_kernel, dim3(sn2_cdiv(win_w, 64), sn2_cdiv(win_h, 4), 3), dim3(256), 0,
                       (hipStream_t)stream, rasters, weights, offsets, B, D, H, W, mean, wsum, win_y0, win_x0, win_h, win_w);
    SN2_RETURN_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// Mosaic finalisation: finalize_merged_raster without the GIS step (inference/geotiff_raster.py:262-285 + :119-144).
// The reference finds the hard medium-vegetation threshold by evaluating 10 001 thresholds, each over the whole image
// (O(10^4 x pixels) in numpy).  Here every valid pixel is binned once by the number k(v) of thresholds below its value;
// the number of pixels above threshold i is then a suffix sum of that histogram, the 10 001 deltas come from it, and the
// first minimum is taken as numpy's argmin does.  Thresholds lin[i] = i * (1/10000) in fp64, lin[10000] = 1 (np.linspace);
// the comparison v > lin[i] is made in fp64.
// ------------------------------------------------------------------------------------------------------------
namespace {
constexpr int HARD_STEPS = 10001;

__device__ __forceinline__ double hard_lin(int i) { return i >= HARD_STEPS - 1 ? 1.0 : (double)i * (1.0 / 10000.0); }

// ws: [0..HARD_STEPS] histogram of k, [HARD_STEPS+1] number of valid pixels; sum_ws: fp64 sum of the valid values
__global__ __launch_bounds__(256) void hard_hist_kernel(const float* __restrict__ med, long P, int* __restrict__ ws,
                                                        double* __restrict__ sum_ws) {
    __shared__ double s_sum[4];
    __shared__ int s_cnt[4];
    double acc = 0.0;
    int nv = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < P; i += (long)gridDim.x * 256) {
        const float vf = med[i];
        if (vf != vf) continue;
        const double v = (double)vf;
        int k = (int)floor(v * 10000.0);
        k = k < 0 ? 0 : (k > HARD_STEPS - 1 ? HARD_STEPS - 1 : k);
        while (k <= HARD_STEPS - 1 && hard_lin(k) < v) ++k;            // k = #{ i : lin[i] < v }
        while (k > 0 && !(hard_lin(k - 1) < v)) --k;
        atomicAdd(&ws[k], 1);
        acc += v;
        ++nv;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        acc += __shfl_xor(acc, o);
        nv += __shfl_xor(nv, o);
    }
    if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = acc; s_cnt[threadIdx.x >> 6] = nv; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(sum_ws, (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]));
        atomicAdd(&ws[HARD_STEPS + 1], (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]));
    }
}

// one workgroup: suffix sums, deltas, first minimum -> thr_out[0] = threshold (fp64 value as float), thr_out[1] = index
__global__ __launch_bounds__(1024) void hard_threshold_kernel(const int* __restrict__ ws, const double* __restrict__ sum_ws,
                                                              float* __restrict__ thr_out) {
    __shared__ long long s_above[HARD_STEPS + 1];     // s_above[i] = #{pixels with k > i}
    __shared__ double s_best[1024];
    __shared__ int s_idx[1024];
    const int tid = threadIdx.x;
    // chunked suffix sum over k = HARD_STEPS .. 0: 1024 threads x 10 bins
    constexpr int PER = (HARD_STEPS + 1 + 1023) / 1024;
    long long loc = 0;
    const int hi = HARD_STEPS - tid * PER;            // this thread's bins: hi, hi-1, ... (descending)
    for (int j = 0; j < PER; ++j) {
        const int k = hi - j;
        if (k >= 0) loc += ws[k];
    }
    __shared__ long long s_chunk[1024];
    s_chunk[tid] = loc;
    __syncthreads();
    if (tid == 0) {
        long long run = 0;
        for (int t = 0; t < 1024; ++t) { const long long c = s_chunk[t]; s_chunk[t] = run; run += c; }
    }
    __syncthreads();
    long long run = s_chunk[tid];                     // pixels with k above this thread's highest bin
    for (int j = 0; j < PER; ++j) {
        const int k = hi - j;
        if (k >= 0) {
            if (k <= HARD_STEPS) s_above[k] = run;    // #{k' > k}
            run += ws[k];
        }
    }
    __syncthreads();
    const int nvalid = ws[HARD_STEPS + 1];
    // np.nanmean of a float32 image is a float32; the hard images are fp64 (1.0 * bool)
    const double target = nvalid > 0 ? (double)(float)(sum_ws[0] / (double)nvalid) : __longlong_as_double(0x7ff8000000000000LL);
    double best = INFINITY;
    int bidx = 0x7FFFFFFF;
    for (int i = tid; i < HARD_STEPS; i += 1024) {
        // pixels with v > lin[i]  <=>  k(v) > i
        const double hard_mean = nvalid > 0 ? (double)s_above[i] / (double)nvalid : __longlong_as_double(0x7ff8000000000000LL);
        const double d = fabs(target - hard_mean);
        if (d < best) { best = d; bidx = i; }         // ascending i per thread: first minimum kept
    }
    s_best[tid] = best;
    s_idx[tid] = bidx;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) {
            const double ob = s_best[tid + o];
            const int oi = s_idx[tid + o];
            if (ob < s_best[tid] || (ob == s_best[tid] && oi < s_idx[tid])) { s_best[tid] = ob; s_idx[tid] = oi; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const int i = s_idx[0] == 0x7FFFFFFF ? 0 : s_idx[0];   // all-NaN deltas: np.argmin returns 0
        thr_out[0] = (float)hard_lin(i);
        thr_out[1] = (float)i;
    }
}

// out (5,H,W) = [Vb, Vm_soft, Vh, Vm_hard, weights] with the reference's NaN rule: NaN -> 0 wherever at least one of the
// three scores is a number, all bands NaN elsewhere
__global__ __launch_bounds__(256) void mosaic_finalize_kernel(const float* __restrict__ mean, const float* __restrict__ wsum,
                                                              long P, const float* __restrict__ thr, float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float nanv = __int_as_float(0x7fc00000);
    const float b0 = mean[i], b1 = mean[P + i], b2 = mean[2 * P + i], w = wsum[i];
    const bool none = (b0 != b0) && (b1 != b1) && (b2 != b2);
    const double t = hard_lin((int)thr[1]);
    float hard = (b1 != b1) ? nanv : (((double)b1 > t) ? 1.f : 0.f);
    auto fix = [&](float v) { return none ? nanv : (v != v ? 0.f : v); };
    out[i] = fix(b0);
    out[P + i] = fix(b1);
    out[2 * P + i] = fix(b2);
    out[3 * P + i] = fix(hard);
    out[4 * P + i] = fix(w);
}
}  // namespace

extern "C" int sn2_mosaic_finalize(const float* mean, const float* wsum, int H, int W, int* hist_ws, double* sum_ws,
                                   float* thr_out, float* out, void* stream) {
    if (!mean || !wsum || !hist_ws || !sum_ws || !thr_out || !out || H <= 0 || W <= 0) return SN2_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const long P = (long)H * W;
    sn2_fill_words(hist_ws, 0u, (size_t)SN2_MOSAIC_HIST_WORDS, st);
    sn2_fill_words(sum_ws, 0u, 2, st);
    int grid = sn2_cdiv(P, 256);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(hard_hist_kernel, dim3(grid), dim3(256), 0, st, mean + P, P, hist_ws, sum_ws);
    hipLaunchKernelGGL(hard_threshold_kernel, dim3(1), dim3(1024), 0, st, (const int*)hist_ws, (const double*)sum_ws, thr_out);
    hipLaunchKernelGGL(mosaic_finalize_kernel, dim3(sn2_cdiv(P, 256)), dim3(256), 0, st, mean, wsum, P, (const float*)thr_out, out);
    SN2_RETURN_LAUNCH();
}
