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

// (out = a*ext + c, 0 for a centroid that received no message as PointConv's scatter-max: misc.hip, bn_finalize_kernel<true>)

// dbeta[o] = sum_i dout[i][o],  dgamma[o] = sum_i dout[i][o] * xhat(ext[i][o])  over the B*M extremum rows
__global__ __launch_bounds__(256) void sa_bwd_prep_kernel(const float* __restrict__ dout, const float* __restrict__ ext,
                                                          const int* __restrict__ arg, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, int rows, int C,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int o = threadIdx.x % C, g = threadIdx.x / C, G = 256 / C;
    float sb = 0.f, sg = 0.f;
    const float mu = mean[o], is = invstd[o];
    // eight rows per turn, all 24 loads issued before the first use (one dependent arg -> dout, ext chain per row was
    // 8 round trips = 9.6 us for 16 384 rows)
    const int step = gridDim.x * G;
    for (int r0 = blockIdx.x * G + g; g < G && r0 < rows; r0 += 8 * step) {
        int ar[8];
        float dv[8], ev[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + u * step;
            const size_t i = (size_t)(r < rows ? r : r0) * C + o;
            ar[u] = r < rows ? arg[i] : -1;
            dv[u] = dout[i];
            ev[u] = ext[i];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
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
    // training == SN2_BN_FROZEN_KEEP: the kernels of a training pass (ext, arg kept for the backward), the finalisations of an eval
    // pass (running statistics, nothing updated); block 0's statistics pass has nothing to measure then
    const int batch_stats = training == 1;
    if constexpr (NL == 2) {
        if (batch_stats) SN2_TRY((sa_mfma_launch_fwd<CF, NL, C1, C2, 0>(p, training, st, &nb)));
        SN2_TRY(sn2_bn_finalize(&p->blk[0], nb, p->total, 0, batch_stats, st));
    }
    if (!training) {
        // EVAL: every block's (a, c) comes from its running statistics, known before the pass: the kernel writes the level's
        // output a ext + c itself (p->ext, p->arg stay untouched)
        SN2_TRY(sn2_bn_finalize(last, 0, p->total, 0, 0, st));
        return sa_mfma_launch_fwd<CF, NL, C1, C2, 1>(p, 0, st, &nb);
    }
    SN2_TRY((sa_mfma_launch_fwd<CF, NL, C1, C2, 1>(p, training, st, &nb)));
    // the last block's statistics -> (a, c), and out = a ext + c, in ONE launch (round 5: bn_finalize + sa_finalize_kernel were two)
    return sn2_bn_finalize_apply(last, batch_stats ? nb : 0, p->total, 0, batch_stats, p->ext, p->arg, p->out, (long)p->B * p->M, st);
}

template <int CF, int NL, int C1, int C2>
int backward_t(const sn2_sa* p, hipStream_t st) {
    const sn2_block* last = &p->blk[NL - 1];
    const int rows = p->B * p->M, C = last->cout;
    int pb = sn2_cdiv(rows, 256 / C);
    if (pb > 256) pb = 256;      // (fewer workgroups = fewer same-address atomics, but slower: 8.7 / 9.8 / 16 us at 256 / 64 / 32)
    hipLaunchKernelGGL(sa_bwd_prep_kernel, dim3(pb), dim3(256), 0, st, p->dout, p->ext, p->arg, last->mean, last->invstd,
                       rows, C, last->dgamma, last->dbeta);
    if constexpr (NL == 2) SN2_TRY((sa_mfma_launch_bwd<CF, NL, C1, C2, 2>(p, st)));
    SN2_TRY((sa_mfma_launch_bwd<CF, NL, C1, C2, 3>(p, st)));
    return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------
// Work items of the SA passes.  A wave step is four 16-message tiles, and the plot's centroids are ranked by descending
// neighbour count (ties: ascending id):
//   SOLO  a centroid with more than SN2_SA_SOLO_MIN neighbours takes all four tiles itself (64 messages per step);
//   QUAD  SN2_SA_QUAD_MIN < n <= SOLO_MIN: consecutive ranks 4k..4k+3 share the steps, one tile each (the lists end together);
//   OCT   SN2_SA_OCT_MIN < n <= QUAD_MIN: eight consecutive ranks, half a tile each, ONE step;
//   HEX   n <= OCT_MIN: sixteen, a quarter tile each, one step.
// (Ball sizes at C2: median 5, mean 25, maximum 261 -- four tiles of one centroid ran 32 % full, quads alone 65 %: 40 % of
// the centroids have at most four neighbours and another 38 % at most eight, each of them a whole 16-row tile of its own;
// the parcel loop's 10 000-point plots: 52 %.)  Ranking = a bitonic sort of (count, index) keys in LDS: deterministic, no
// atomics on the order.  Items of all plots are interleaved heaviest first -- item k of plot b at position k*B + b -- and
// the SA kernels deal the positions to their waves in that order (in a snake).
// order (SN2_SA_ORDER_WORDS ints, -1 = none):
//   [0, 4 B M)               SOLO / QUAD positions, 4 ints each (solo: id | SN2_SA_SOLO_FLAG four times; quad: four ids);
//   then 16 B SN2_SA_PACKED_ITEMS(M) ints: OCT / HEX positions, 16 ints each = the centroid of every QUARTER tile (an OCT's
//                            ids twice each, | SN2_SA_OCT_FLAG);
//   then the trailer: [0] = max_b(SOLO + QUAD items of plot b), [1] = max_b(OCT + HEX items of plot b).
// ------------------------------------------------------------------------------------------------------------
namespace {
// (grouped launches, round 5: plot bg of the launch is plot bg % B of batch bg / B, whose table starts `stride` ints behind the
// previous batch's -- sn2_sa_order_group)
__global__ __launch_bounds__(1024) void sa_order_sort_kernel(const int* __restrict__ cnt, int B, int M, int P2,
                                                             int* __restrict__ order_all, size_t stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned s_key[];       // [P2]
    __shared__ int s_n[3];                                                  // centroids of the classes SOLO, QUAD, OCT
    const int bg = blockIdx.x, tid = threadIdx.x, NT = (int)blockDim.x;    // 1024 threads, or 256 when there are many plots
    const int hb = bg / B, b = bg - hb * B;
    int* __restrict__ order = order_all + (size_t)hb * stride;
    const int* cb = cnt + (size_t)bg * M;
    if (tid < 3) s_n[tid] = 0;
    __syncthreads();
    int n_here[3] = {0, 0, 0};
    for (int i = tid; i < P2; i += NT) {
        unsigned key = 0xFFFFFFFFu;                                        // padding sorts last
        if (i < M) {
            const int c = cb[i];
            key = ((unsigned)(0x3FFFF - c) << 14) | (unsigned)i;           // counts <= 2000 < 2^18, indices < 2^14
            n_here[0] += c > SN2_SA_SOLO_MIN ? 1 : 0;
            n_here[1] += (c > SN2_SA_QUAD_MIN && c <= SN2_SA_SOLO_MIN) ? 1 : 0;
            n_here[2] += (c > SN2_SA_OCT_MIN && c <= SN2_SA_QUAD_MIN) ? 1 : 0;
        }
        s_key[i] = key;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (n_here[k]) atomicAdd(&s_n[k], n_here[k]);
    __syncthreads();
    for (int k = 2; k <= P2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P2 >> 1); t += NT) {
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
    // the classes are the rank ranges [0, nsolo), [nsolo, r_oct), [r_oct, r_hex), [r_hex, M)
    const int nsolo = s_n[0], r_oct = nsolo + s_n[1], r_hex = r_oct + s_n[2];
    const int items_a = nsolo + ((s_n[1] + 3) >> 2), n_oct_items = (s_n[2] + 7) >> 3, n_hex_items = (M - r_hex + 15) >> 4;
    int* packed = order + (size_t)4 * B * M;
    for (int r = tid; r < M; r += NT) {
        const int id = b * M + (int)(s_key[r] & 0x3FFFu);
        if (r < nsolo) {
            int* dst = order + 4 * ((size_t)r * B + b);
            dst[0] = dst[1] = dst[2] = dst[3] = id | SN2_SA_SOLO_FLAG;
        } else if (r < r_oct) {
            const int rl = r - nsolo;
            order[4 * ((size_t)(nsolo + (rl >> 2)) * B + b) + (rl & 3)] = id;
        } else if (r < r_hex) {
            const int rl = r - r_oct;                                       // half-tile (rl & 7) of OCT item rl >> 3
            int* dst = packed + 16 * ((size_t)(rl >> 3) * B + b) + 2 * (rl & 7);
            dst[0] = dst[1] = id | SN2_SA_OCT_FLAG;
        } else {
            const int rl = r - r_hex;                                       // quarter-tile (rl & 15) of HEX item rl >> 4
            packed[16 * ((size_t)(n_oct_items + (rl >> 4)) * B + b) + (rl & 15)] = id;
        }
    }
    if (tid == 0) {
        int* trailer = packed + (size_t)16 * B * SN2_SA_PACKED_ITEMS(M);
        atomicMax(&trailer[0], items_a);
        atomicMax(&trailer[1], n_oct_items + n_hex_items);
    }
}
}  // namespace

static int sa_order_impl(const int* cnt, int G, int B, int M, int* order, size_t stride, hipStream_t st);
extern "C" int sn2_sa_order(const int* cnt, int B, int M, int* order, void* stream) {
    if (!cnt || !order || B <= 0 || M <= 0) return SN2_EINVAL;
    return sa_order_impl(cnt, 1, B, M, order, SN2_SA_ORDER_WORDS(B, M), (hipStream_t)stream);
}
// G consecutive batches of B plots each in one launch pair: cnt (G*B*M), batch h's table at order + h * stride_words
extern "C" int sn2_sa_order_group(const int* cnt, int G, int B, int M, int* order, size_t stride_words, void* stream) {
    if (!cnt || !order || G <= 0 || B <= 0 || M <= 0 || stride_words < SN2_SA_ORDER_WORDS(B, M)) return SN2_EINVAL;
    return sa_order_impl(cnt, G, B, M, order, stride_words, (hipStream_t)stream);
}
static int sa_order_impl(const int* cnt, int G, int B, int M, int* order, size_t stride, hipStream_t st) {
    if (M > 16384) return SN2_ELIMIT;                     // the plot's keys must fit LDS (and 14 bits)
    sn2_fill_words(order, 0xFFFFFFFFu, (size_t)(G - 1) * stride + (size_t)SN2_SA_ORDER_WORDS(B, M), st);         // all -1
    int P2 = 2;
    while (P2 < M) P2 <<= 1;
    if ((size_t)P2 * 4 > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sa_order_sort_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, P2 * 4);
    // one workgroup per plot.  With hundreds of plots (the parcel loop) a 1024-thread workgroup waits for a CU with sixteen free
    // wave slots while the four-wave workgroups of concurrent kernels keep taking every slot that frees up (0.27 ms instead of
    // 0.03 for 256 plots of 2500 centroids): many plots -> 256 threads each
    hipLaunchKernelGGL(sa_order_sort_kernel, dim3(G * B), dim3(sn2_small_sort_wg(B) ? 256 : 1024), (size_t)P2 * 4, st, cnt, B, M, P2,
                       order, stride);
    SN2_RETURN_LAUNCH();
}

extern "C" int sn2_sa_forward(const sn2_sa* p, int training, void* stream) {
    SN2_TRY(check(p));
    if (training < 0 || training > SN2_BN_FROZEN_KEEP) return SN2_EINVAL;
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
