// loss.hip -- the loss block of the training step as three kernels (gfx950).
//
// Replaces the ~75 elementwise / reduction launches torch needs for
//   loss = get_absolute_loss(pred, gt) + m * get_NLL_loss(proba, pdf_all) + e * get_entropy_loss(proba)
// (/root/reference/learning/loss_functions.py:9-57 as combined by learning/train.py:58-62) and for its backward.
// Arithmetic types follow what torch's type promotion gives the reference: the absolute loss and the NLL in fp64
// (gt and the KDE densities are fp64 numpy arrays), the entropy terms in fp32.  Sums are per-workgroup partials added in
// a fixed order: deterministic.
#include "common.h"

namespace {

constexpr int LOSS_BLOCKS = SN2_LOSS_BLOCKS;
constexpr float EPS_F = 0.0001f;
constexpr double EPS_D = 0.0001;

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// one row per lane: 16 B of probabilities + 24 B of densities, both contiguous across the wave.  NLL / ENT: which of the two
// pointwise terms are wanted (a term whose weight is zero is SKIPPED, not multiplied by zero: a caller that asks for the entropy
// of rows that are no probability vectors -- the reference's docstring says "coverage raster" -- must not get 0 * log(<= 0) =
// NaN from a likelihood nobody asked for, and the densities need not exist)
template <bool NLL, bool ENT>
__global__ __launch_bounds__(256) void loss_point_kernel(const float4* __restrict__ proba, const double* __restrict__ pdf,
                                                         int R, double* __restrict__ partials) {
    __shared__ double s_part[2][4];
    double nll = 0.0, ent = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < R; i += gridDim.x * 256) {
        const float4 p = proba[i];
        if constexpr (NLL) {
            const double f0 = pdf[3 * (size_t)i], f1 = pdf[3 * (size_t)i + 1], f2 = pdf[3 * (size_t)i + 2];
            const float pg = p.x + p.y;                               // pred[:, :2].sum(1) in fp32 (:44)
            const double lik = ((double)pg * f0 + (double)p.z * f1) + (double)p.w * f2;
            nll -= log(lik);
        }
        if constexpr (ENT) {
            const float e2 = p.z * logf(p.z + EPS_F) + (1.f - p.z) * logf(1.f - p.z + EPS_F);
            const float e3 = p.w * logf(p.w + EPS_F) + (1.f - p.w) * logf(1.f - p.w + EPS_F);
            ent -= (double)e2 + (double)e3;
        }
    }
    nll = wave_sum_f64(nll);
    ent = wave_sum_f64(ent);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_part[0][w] = nll; s_part[1][w] = ent; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = (s_part[0][0] + s_part[0][1]) + (s_part[0][2] + s_part[0][3]);
        partials[2 * blockIdx.x + 1] = (s_part[1][0] + s_part[1][1]) + (s_part[1][2] + s_part[1][3]);
    }
}

// out[0..3] = total, absolute, NLL, entropy
__global__ __launch_bounds__(256) void loss_final_kernel(const float* __restrict__ pred, const double* __restrict__ gt, int B,
                                                         const double* __restrict__ partials, int nblocks, int R, double m,
                                                         double e, double* __restrict__ out) {
    __shared__ double s[3][256];
    double nll = 0.0, ent = 0.0, ab = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) { nll += partials[2 * i]; ent += partials[2 * i + 1]; }
    for (int i = threadIdx.x; i < 3 * B; i += 256) {                   // strata low, medium, high = columns 0, 2, 3 (:12)
        const int b = i / 3, c = i - 3 * b, col = c == 0 ? 0 : c + 1;
        const double d = (double)pred[4 * b + col] - gt[4 * b + col];
        ab += sqrt(d * d + EPS_D);
    }
    s[0][threadIdx.x] = nll; s[1][threadIdx.x] = ent; s[2][threadIdx.x] = ab;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            s[0][threadIdx.x] += s[0][threadIdx.x + o];
            s[1][threadIdx.x] += s[1][threadIdx.x + o];
            s[2][threadIdx.x] += s[2][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double l_abs = B > 0 ? s[2][0] / (3.0 * B) : 0.0, l_nll = nblocks > 0 ? s[0][0] / R : 0.0;
        const double l_ent = nblocks > 0 ? (double)(float)(s[1][0] / (2.0 * R)) : 0.0;      // the reference's entropy is an fp32 tensor
        out[0] = l_abs + m * l_nll + e * l_ent;
        out[1] = l_abs;
        out[2] = l_nll;
        out[3] = l_ent;
    }
}

// d loss / d proba, d loss / d pred for the upstream gradient g of the TOTAL loss (device scalar)
template <bool NLL, bool ENT>
__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ pred, const double* __restrict__ gt, int B,
                                                       const float4* __restrict__ proba, const double* __restrict__ pdf, int R,
                                                       double m, double e, const double* __restrict__ gout,
                                                       float* __restrict__ dpred, float4* __restrict__ dproba) {
    const double g = gout[0];
    const double cn = R > 0 ? g * m / R : 0.0, ce = R > 0 ? g * e / (2.0 * R) : 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < R; i += gridDim.x * 256) {
        const float4 p = proba[i];
        double d0 = 0.0, d2 = 0.0, d3 = 0.0;
        if constexpr (NLL) {
            const double f0 = pdf[3 * (size_t)i], f1 = pdf[3 * (size_t)i + 1], f2 = pdf[3 * (size_t)i + 2];
            const float pg = p.x + p.y;
            const double lik = ((double)pg * f0 + (double)p.z * f1) + (double)p.w * f2;
            const double il = -cn / lik;
            d0 = il * f0, d2 = il * f1, d3 = il * f2;
        }
        if constexpr (ENT) {
            const float h2 = -(logf(p.z + EPS_F) + p.z / (p.z + EPS_F) - logf(1.f - p.z + EPS_F) - (1.f - p.z) / (1.f - p.z + EPS_F));
            const float h3 = -(logf(p.w + EPS_F) + p.w / (p.w + EPS_F) - logf(1.f - p.w + EPS_F) - (1.f - p.w) / (1.f - p.w + EPS_F));
            d2 += ce * (double)h2, d3 += ce * (double)h3;
        }
        float4 d;
        d.x = d.y = (float)d0;
        d.z = (float)d2;
        d.w = (float)d3;
        dproba[i] = d;
    }
    if (blockIdx.x == 0 && dpred) {
        for (int i = threadIdx.x; i < 4 * B; i += 256) {
            const int col = i & 3;
            float r = 0.f;
            if (col != 1) {
                const double d = (double)pred[i] - gt[i];
                r = (float)(g * d / sqrt(d * d + EPS_D) / (3.0 * B));
            }
            dpred[i] = r;
        }
    }
}

// KDE-mixture densities at the points' heights: KdeMixture.predict (learning/kde_mixture.py:65-70) = three scipy
// interp1d(kind="linear") over one knot vector, evaluated by get_NLL_loss on the CPU for all B*N points every step
// (learning/loss_functions.py:30-42).  Same arithmetic as scipy's _call_linear, in fp64 without contraction:
//   i = clip(searchsorted(X, z, side="left"), 1, K-1);  slope = (y[i] - y[i-1]) / (X[i] - X[i-1]);  y = slope*(z - X[i-1]) + y[i-1]
// z = fp64( fp32(cloud[b][zc][n]) * fp32(z_max) ) as the reference forms it (a float32 tensor times a python float).
// Heights outside [X[0], X[K-1]] (scipy raises ValueError there) give NaN.
__global__ __launch_bounds__(256) void kde_lookup_kernel(const float* __restrict__ cloud, int C, int N, int zc, float z_max,
                                                         const double* __restrict__ X, const double* __restrict__ Y, int K,
                                                         long R, double* __restrict__ pdf) {
#pragma clang fp contract(off)
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const long b = r / N, n = r - b * N;
    const double z = (double)(cloud[(b * C + zc) * N + n] * z_max);
    int lo = 0, hi = K;                         // first index with X[i] >= z
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (X[mid] < z) lo = mid + 1;
        else hi = mid;
    }
    const bool inside = z >= X[0] && z <= X[K - 1];
    int i = lo < 1 ? 1 : (lo > K - 1 ? K - 1 : lo);
    const double x_lo = X[i - 1], x_hi = X[i];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double y_lo = Y[(size_t)k * K + i - 1], y_hi = Y[(size_t)k * K + i];
        const double slope = (y_hi - y_lo) / (x_hi - x_lo);
        const double v = slope * (z - x_lo) + y_lo;
        pdf[3 * r + k] = inside ? v : __longlong_as_double(0x7ff8000000000000LL);
    }
}

inline int loss_grid(int R) {
    const int n = sn2_cdiv(R, 256);
    return n < LOSS_BLOCKS ? n : LOSS_BLOCKS;
}

}  // namespace

extern "C" int sn2_loss_forward(const float* pred, const double* gt, int B, const float* proba, const double* pdf, int R,
                                double m, double e, double* partials, double* out, void* stream) {
    // a term that is switched off is skipped and its inputs may be absent: B = 0 (no absolute term: pred, gt unused),
    // m == 0 (no NLL: pdf unused), e == 0 (no entropy); m == e == 0 or R == 0: no pass over the points at all
    if (!out || B < 0 || R < 0 || (B > 0 && (!pred || !gt))) return SN2_EINVAL;
    const bool nll = m != 0.0 && R > 0, ent = e != 0.0 && R > 0;
    if ((nll || ent) && (!proba || !partials)) return SN2_EINVAL;
    if (nll && !pdf) return SN2_EINVAL;
    const int nb = (nll || ent) ? loss_grid(R) : 0;
    hipStream_t st = (hipStream_t)stream;
    const float4* p4 = reinterpret_cast<const float4*>(proba);
    if (nll && ent) hipLaunchKernelGGL((loss_point_kernel<true, true>), dim3(nb), dim3(256), 0, st, p4, pdf, R, partials);
    else if (nll) hipLaunchKernelGGL((loss_point_kernel<true, false>), dim3(nb), dim3(256), 0, st, p4, pdf, R, partials);
    else if (ent) hipLaunchKernelGGL((loss_point_kernel<false, true>), dim3(nb), dim3(256), 0, st, p4, pdf, R, partials);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, st, pred, gt, B, (const double*)partials, nb, R > 0 ? R : 1, m, e, out);
    SN2_RETURN_LAUNCH();
}

extern "C" int sn2_loss_backward(const float* pred, const double* gt, int B, const float* proba, const double* pdf, int R,
                                 double m, double e, const double* grad_total, float* dpred, float* dproba, void* stream) {
    if (!grad_total || B < 0 || R < 0 || (B > 0 && (!pred || !gt || !dpred)) || (R > 0 && (!proba || !dproba))) return SN2_EINVAL;
    const bool nll = m != 0.0 && R > 0, ent = e != 0.0 && R > 0;
    if (nll && !pdf) return SN2_EINVAL;
    if (B == 0 && R == 0) return 0;
    const int grid = R > 0 ? loss_grid(R) : 1;
    hipStream_t st = (hipStream_t)stream;
    const float4* p4 = reinterpret_cast<const float4*>(proba);
    float4* d4 = reinterpret_cast<float4*>(dproba);
    float* dp = B > 0 ? dpred : nullptr;
    if (nll && ent) hipLaunchKernelGGL((loss_bwd_kernel<true, true>), dim3(grid), dim3(256), 0, st, pred, gt, B, p4, pdf, R, m, e, grad_total, dp, d4);
    else if (nll) hipLaunchKernelGGL((loss_bwd_kernel<true, false>), dim3(grid), dim3(256), 0, st, pred, gt, B, p4, pdf, R, m, e, grad_total, dp, d4);
    else if (ent) hipLaunchKernelGGL((loss_bwd_kernel<false, true>), dim3(grid), dim3(256), 0, st, pred, gt, B, p4, pdf, R, m, e, grad_total, dp, d4);
    else hipLaunchKernelGGL((loss_bwd_kernel<false, false>), dim3(grid), dim3(256), 0, st, pred, gt, B, p4, pdf, R, m, e, grad_total, dp, d4);
    SN2_RETURN_LAUNCH();
}

extern "C" int sn2_kde_lookup(const float* cloud, int B, int C, int N, int z_channel, float z_max, const double* X,
                              const double* Y, int K, double* pdf, void* stream) {
    if (!cloud || !X || !Y || !pdf || B <= 0 || C <= 0 || N <= 0 || z_channel < 0 || z_channel >= C || K < 2) return SN2_EINVAL;
    const long R = (long)B * N;
    hipLaunchKernelGGL(kde_lookup_kernel, dim3(sn2_cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, cloud, C, N, z_channel,
                       z_max, X, Y, K, R, pdf);
    SN2_RETURN_LAUNCH();
}
