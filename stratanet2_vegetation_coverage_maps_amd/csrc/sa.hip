// sa.hip -- set abstraction (PointConv gather + shared MLP + BatchNorm + max), host side and the small per-centroid
// kernels.  Replaces torch_geometric PointConv(local_nn, aggr='max') + torch_scatter as called from SAModule.forward,
// /root/reference/model/point_net2.py:19,21-29.  The message passes themselves run on the matrix cores: sa_mfma.hip.
//
// Training-mode BatchNorm normalises over ALL E messages of the batch, so a block's statistics must be complete before
// the next block can run:
//   forward  nl=2:  [pass 0: block-0 statistics] -> bn_finalize(0) -> [pass 1: block 0, block 1, statistics, signed
//                   extremum + slot] -> bn_finalize(1);   nl=1: [pass 1] -> bn_finalize(0)
//   then out = a*ext + c on the (B*M, C) extremum only: BN is a per-channel affine, so
//   max_e (a*h_e + c) = a * (a >= 0 ? max_e h_e : min_e h_e) + c, and sign(a) = sign(gamma) is known beforehand.
//   backward: prep (dgamma/dbeta of the last BN from the B*M extremum rows) -> pass C (nl=2: last block's dW/db, first
//             BN's dgamma/dbeta) -> pass D (first block's dW/db, input-feature gradient).
// The E x C message tensors of the reference are never materialised: every pass re-gathers the 48/80-byte source rows
// (L2-resident: a plot's rows are 1.5 MB) and recomputes the MLP in registers.
#include "mlp.h"

// forward passes on the matrix cores (sa_mfma.hip)
template <int CF, int NL, int C1, int C2, int PASS>
int sa_mfma_launch_fwd(const sn2_sa* p, int training, hipStream_t st, int* nblocks_out);
template <int CF, int NL, int C1, int C2, int PASS>
int sa_mfma_launch_bwd(const sn2_sa* p, hipStream_t st);

namespace {

// out = a*ext + c  (0 for a centroid that received no message, as PointConv's scatter-max)
__global__ void sa_finalize_kernel(const float* __restrict__ ext, const int* __restrict__ arg, const float* __restrict__ a,
                                   const float* __restrict__ c, int rows, int C, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * C) return;
    const int o = i % C;
    out[i] = arg[i] >= 0 ? fmaf(a[o], ext[i], c[o]) : 0.f;
}

// dbeta[o] = sum_i dout[i][o],  dgamma[o] = sum_i dout[i][o] * xhat(ext[i][o])  over the B*M extremum rows
__global__ __launch_bounds__(256) void sa_bwd_prep_kernel(const float* __restrict__ dout, const float* __restrict__ ext,
                                                          const int* __restrict__ arg, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, int rows, int C,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int o = threadIdx.x % C, g = threadIdx.x / C, G = 256 / C;
    float sb = 0.f, sg = 0.f;
    const float mu = mean[o], is = invstd[o];
    // four rows per turn, all twelve loads issued before the first use (one dependent arg -> dout, ext chain per row was
    // 8 round trips = 9.6 us for 16 384 rows)
    const int step = gridDim.x * G;
    for (int r0 = blockIdx.x * G + g; g < G && r0 < rows; r0 += 4 * step) {
        int ar[4];
        float dv[4], ev[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + u * step;
            const size_t i = (size_t)(r < rows ? r : r0) * C + o;
            ar[u] = r < rows ? arg[i] : -1;
            dv[u] = dout[i];
            ev[u] = ext[i];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (ar[u] >= 0) {
                sb += dv[u];
                sg = fmaf(dv[u], (ev[u] - mu) * is, sg);
            }
    }
    // workgroup-level reduction first (256 threads -> 2*C values), then one global atomic per value
    __shared__ float s_red[128];
    for (int i = threadIdx.x; i < 2 * C; i += 256) s_red[i] = 0.f;
    __syncthreads();
    if (g < G) {
        atomicAdd(&s_red[o], sb);
        atomicAdd(&s_red[C + o], sg);
    }
    __syncthreads();
    if (threadIdx.x < 2 * C) {
        const float v = s_red[threadIdx.x];
        if (v != 0.f) atomicAdd(threadIdx.x < C ? &dbeta[threadIdx.x] : &dgamma[threadIdx.x - C], v);
    }
}

int check(const sn2_sa* p) {
    if (!p || p->B <= 0 || p->Nsrc <= 0 || p->M <= 0 || p->cap <= 0) return SN2_EINVAL;
    if (!p->feat || !p->spos || !p->cpos || !p->nbr || !p->cnt || !p->total || !p->ext || !p->arg || !p->out)
        return SN2_EINVAL;
    if (p->feat_stride < p->cf || (p->feat_stride & 3) || p->spos_stride < 4 || (p->spos_stride & 3)) return SN2_EINVAL;
    const bool sa1 = p->cf == 8 && p->nl == 2 && p->blk[0].cin == 11 && p->blk[0].cout == 16 && p->blk[1].cin == 16 &&
                     p->blk[1].cout == 16;
    const bool sa2 = p->cf == 16 && p->nl == 1 && p->blk[0].cin == 19 && p->blk[0].cout == 32;
    // a third ball-query level, MLP[35,64]: not in the reference (BASELINE config 2 names it: the "3sa-arch" variant)
    const bool sa3 = p->cf == 32 && p->nl == 1 && p->blk[0].cin == 35 && p->blk[0].cout == 64;
    return (sa1 || sa2 || sa3) ? 0 : SN2_ELIMIT;
}

template <int CF, int NL, int C1, int C2>
int forward_t(const sn2_sa* p, int training, hipStream_t st) {
    const sn2_block* last = &p->blk[NL - 1];
    int nb = 0;
    if constexpr (NL == 2) {
        if (training) SN2_TRY((sa_mfma_launch_fwd<CF, NL, C1, C2, 0>(p, training, st, &nb)));
        SN2_TRY(sn2_bn_finalize(&p->blk[0], nb, p->total, 0, training, st));
    }
    SN2_TRY((sa_mfma_launch_fwd<CF, NL, C1, C2, 1>(p, training, st, &nb)));
    SN2_TRY(sn2_bn_finalize(last, nb, p->total, 0, training, st));
    const int rows = p->B * p->M, C = last->cout;
    hipLaunchKernelGGL(sa_finalize_kernel, dim3(sn2_cdiv((long)rows * C, 256)), dim3(256), 0, st, p->ext, p->arg, last->a,
                       last->c, rows, C, p->out);
    SN2_RETURN_LAUNCH();
}

template <int CF, int NL, int C1, int C2>
int backward_t(const sn2_sa* p, hipStream_t st) {
    const sn2_block* last = &p->blk[NL - 1];
    const int rows = p->B * p->M, C = last->cout;
    int pb = sn2_cdiv(rows, 256 / C);
    if (pb > 256) pb = 256;
    hipLaunchKernelGGL(sa_bwd_prep_kernel, dim3(pb), dim3(256), 0, st, p->dout, p->ext, p->arg, last->mean, last->invstd,
                       rows, C, last->dgamma, last->dbeta);
    if constexpr (NL == 2) SN2_TRY((sa_mfma_launch_bwd<CF, NL, C1, C2, 2>(p, st)));
    SN2_TRY((sa_mfma_launch_bwd<CF, NL, C1, C2, 3>(p, st)));
    return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------
// Work items of the SA passes.  A wave step is four 16-message tiles.  A centroid with more than SN2_SA_SOLO_MIN
// neighbours takes all four tiles itself (a SOLO item: 64 messages per step); the others are ranked by descending
// neighbour count (ties: ascending id) and ranks 4k..4k+3 share a step as a QUAD, one tile each, so that the four lists
// end together.  (Ball sizes at C2: median 5, mean 25, maximum 261 -- four tiles of one centroid ran 32 % full; quads
// alone made the longest item 17 steps.)  Ranking = comparison counting against the plot's counts in LDS: deterministic,
// no atomics.  Items of all plots are interleaved heaviest first -- item k of plot b at position k*B + b -- and the SA
// kernels deal them to their waves in that order (in a snake).  order: 4 ints per position (solo: the id | SN2_SA_SOLO_FLAG
// four times; quad: four ids; -1 = none), 4*B*M ints + a 4-int trailer whose first word is max_b(items of plot b).
// ------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void sa_order_kernel(const int* __restrict__ cnt, int B, int M, int* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) int s_cnt[];
    const int b = blockIdx.y;
    const int* cb = cnt + (size_t)b * M;
    for (int i = threadIdx.x; i < M; i += 256) s_cnt[i] = cb[i];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const int mine = s_cnt[i];
    int rank = 0, nsolo = 0;
    int j = 0;
    for (; j + 4 <= M; j += 4) {                        // four counts per (broadcast) LDS read
        const int4 o = *reinterpret_cast<const int4*>(&s_cnt[j]);
        rank += (o.x > mine || (o.x == mine && j < i)) ? 1 : 0;
        rank += (o.y > mine || (o.y == mine && j + 1 < i)) ? 1 : 0;
        rank += (o.z > mine || (o.z == mine && j + 2 < i)) ? 1 : 0;
        rank += (o.w > mine || (o.w == mine && j + 3 < i)) ? 1 : 0;
        nsolo += (o.x > SN2_SA_SOLO_MIN) + (o.y > SN2_SA_SOLO_MIN) + (o.z > SN2_SA_SOLO_MIN) + (o.w > SN2_SA_SOLO_MIN);
    }
    for (; j < M; ++j) {
        const int o = s_cnt[j];
        rank += (o > mine || (o == mine && j < i)) ? 1 : 0;
        nsolo += o > SN2_SA_SOLO_MIN ? 1 : 0;
    }
    const int id = b * M + i;
    if (mine > SN2_SA_SOLO_MIN) {                       // the solos are exactly the ranks 0..nsolo-1
        int* dst = order + 4 * ((size_t)rank * B + b);
        dst[0] = dst[1] = dst[2] = dst[3] = id | SN2_SA_SOLO_FLAG;
    } else {
        const int rl = rank - nsolo;
        order[4 * ((size_t)(nsolo + (rl >> 2)) * B + b) + (rl & 3)] = id;
    }
    if (i == 0) atomicMax(&order[(size_t)4 * B * M], nsolo + ((M - nsolo + 3) >> 2));
}

// The same order from a SORT: one workgroup per plot sorts the keys ((cap - count) << 14 | index) -- all different, so the
// order is that of sa_order_kernel (count descending, index ascending) by construction -- with a bitonic network in LDS:
// log2(P2) (log2(P2) + 1) / 2 compare-exchange stages instead of M comparisons per centroid (M = 1024: 55 stages against 1024
// compares; the parcel loop's M = 2500: 78 against 2500: 0.17 -> 0.02 ms per call).
__global__ __launch_bounds__(1024) void sa_order_sort_kernel(const int* __restrict__ cnt, int B, int M, int P2,
                                                             int* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) unsigned s_key[];       // [P2]
    __shared__ int s_nsolo;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int* cb = cnt + (size_t)b * M;
    if (tid == 0) s_nsolo = 0;
    __syncthreads();
    int solo_here = 0;
    for (int i = tid; i < P2; i += 1024) {
        unsigned key = 0xFFFFFFFFu;                                        // padding sorts last
        if (i < M) {
            const int c = cb[i];
            key = ((unsigned)(0x3FFFF - c) << 14) | (unsigned)i;           // counts <= 2000 < 2^18, indices < 2^14
            solo_here += c > SN2_SA_SOLO_MIN ? 1 : 0;
        }
        s_key[i] = key;
    }
    if (solo_here) atomicAdd(&s_nsolo, solo_here);
    __syncthreads();
    for (int k = 2; k <= P2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P2 >> 1); t += 1024) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;      // the pair (lo, lo + j)
                const unsigned a = s_key[lo], c = s_key[hi];
                const bool up = (lo & k) == 0;                              // ascending block
                if ((a > c) == up) {
                    s_key[lo] = c;
                    s_key[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    const int nsolo = s_nsolo;
    for (int r = tid; r < M; r += 1024) {
        const int id = b * M + (int)(s_key[r] & 0x3FFFu);
        if (r < nsolo) {                                                    // the solos are exactly the ranks 0..nsolo-1
            int* dst = order + 4 * ((size_t)r * B + b);
            dst[0] = dst[1] = dst[2] = dst[3] = id | SN2_SA_SOLO_FLAG;
        } else {
            const int rl = r - nsolo;
            order[4 * ((size_t)(nsolo + (rl >> 2)) * B + b) + (rl & 3)] = id;
        }
    }
    if (tid == 0) atomicMax(&order[(size_t)4 * B * M], nsolo + ((M - nsolo + 3) >> 2));
}
}  // namespace

extern "C" int sn2_sa_order(const int* cnt, int B, int M, int* order, void* stream) {
    if (!cnt || !order || B <= 0 || M <= 0) return SN2_EINVAL;
    if (M > 16384) return SN2_ELIMIT;                     // the plot's counts must fit LDS
    hipStream_t st = (hipStream_t)stream;
    sn2_fill_words(order, 0xFFFFFFFFu, (size_t)SN2_SA_ORDER_WORDS(B, M), st);                                      // all -1
    static const bool by_counting = getenv("SN2_SA_ORDER_COUNTING") != nullptr;      // (cross-check switch: the O(M^2) form)
    if (!by_counting) {
        int P2 = 2;
        while (P2 < M) P2 <<= 1;
        if ((size_t)P2 * 4 > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sa_order_sort_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, P2 * 4);
        hipLaunchKernelGGL(sa_order_sort_kernel, dim3(B), dim3(1024), (size_t)P2 * 4, st, cnt, B, M, P2, order);
        SN2_RETURN_LAUNCH();
    }
    hipLaunchKernelGGL(sa_order_kernel, dim3(sn2_cdiv(M, 256), B), dim3(256), (size_t)M * 4, st, cnt, B, M, order);
    SN2_RETURN_LAUNCH();
}

extern "C" int sn2_sa_forward(const sn2_sa* p, int training, void* stream) {
    SN2_TRY(check(p));
    if (p->nl == 2) return forward_t<8, 2, 16, 16>(p, training, (hipStream_t)stream);
    if (p->cf == 32) return forward_t<32, 1, 64, 64>(p, training, (hipStream_t)stream);
    return forward_t<16, 1, 32, 32>(p, training, (hipStream_t)stream);
}

extern "C" int sn2_sa_backward(const sn2_sa* p, void* stream) {
    SN2_TRY(check(p));
    if (!p->dout) return SN2_EINVAL;
    if (p->nl == 2) return backward_t<8, 2, 16, 16>(p, (hipStream_t)stream);
    if (p->cf == 32) return backward_t<32, 1, 64, 64>(p, (hipStream_t)stream);
    return backward_t<16, 1, 32, 32>(p, (hipStream_t)stream);
}
