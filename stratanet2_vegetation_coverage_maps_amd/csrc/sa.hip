// sa.hip -- set abstraction: neighbour gather + shared MLP (Linear->ReLU->BN blocks) + max aggregation, forward and
// backward.  Replaces torch_geometric PointConv(local_nn, aggr='max') + torch_scatter as called from
// SAModule.forward, /root/reference/model/point_net2.py:19,21-29.
//
// Work decomposition: ONE WAVE PER CENTROID (grid-stride), one neighbour ("message", "edge") per lane, 64 messages per
// step.  The E x C message tensors of the reference are never materialised: every pass re-gathers the 48/80-byte
// source rows (L2-resident: a plot's rows are 1.5 MB) and recomputes the MLP in registers.
//
// Training-mode BatchNorm normalises over ALL E messages of the batch, so a block's statistics must be complete
// before the next block can run:
//   forward  nl=2:  [stats0 pass] -> bn_finalize(0) -> [main pass: block0, block1, stats1, extremum] -> bn_finalize(1)
//            nl=1:  [main pass: block0, stats0, extremum] -> bn_finalize(0)
//   then out = a*ext + c on the (B*M, C) extremum only: BN is a per-channel affine, so
//   max_e (a*h_e + c) = a * (a >= 0 ? max_e h_e : min_e h_e) + c, and sign(a) = sign(gamma) is known beforehand.
//   backward: prep (dgamma/dbeta of the last BN from the B*M extremum rows) -> pass C (nl=2: last block's dW/db,
//             first BN's dgamma/dbeta) -> pass D (first block's dW/db, input-feature gradient).
#include "mlp.h"

namespace {

// Read-only, wave-uniform arrays of one block.  They reach the kernel as individual `const float* __restrict__`
// kernel arguments (noalias + readonly): only then may the compiler keep them in SGPRs via s_load and hoist them
// across the kernel's own stores; as members of a by-value struct they would be re-read with vector loads after
// every store (256 VGPRs + scratch in a first version of this kernel).
struct BlkRO {
    cfp W, b, gamma, a, c, mean, invstd, dgamma, dbeta;
};
// fresh provenance for one loop iteration (see common.h: opaque)
__device__ __forceinline__ BlkRO launder(const BlkRO& k) {
    return BlkRO{opaque(k.W), opaque(k.b), opaque(k.gamma), opaque(k.a), opaque(k.c), opaque(k.mean), opaque(k.invstd),
                 opaque(k.dgamma), opaque(k.dbeta)};
}
// writable side (atomics)
struct BlkDev {
    float* slots;  // per-workgroup batch-statistics partials or nullptr
    float *dW, *db, *dgamma, *dbeta;
};
#define BLK_RO_PARAMS(n)                                                                                              \
    const float *__restrict__ W##n, const float *__restrict__ b##n, const float *__restrict__ g##n,                  \
        const float *__restrict__ a##n, const float *__restrict__ c##n, const float *__restrict__ mean##n,           \
        const float *__restrict__ invstd##n, const float *__restrict__ dgam##n, const float *__restrict__ dbet##n
#define BLK_RO_MAKE(n)                                                                                               \
    BlkRO{as_const(W##n), as_const(b##n), as_const(g##n), as_const(a##n), as_const(c##n), as_const(mean##n),         \
          as_const(invstd##n), as_const(dgam##n), as_const(dbet##n)}

struct SaDev {
    int B, Nsrc, M, cap, feat_stride, spos_stride;
    BlkDev k0, k1;
    float* ext;
    int* arg;
    float* dfeat;
};

// The single by-value kernel argument.  The kernel does NOT keep its ~40 pointers live in SGPRs (that alone is 80 of the
// ~100 SGPRs and forced thousands of v_readlane spill reloads per iteration in the backward passes): it re-reads what it
// needs from the kernarg segment -- constant memory -- inside the loops, behind `opaque()` (common.h).
struct SaK {
    SaDev p;
    const float *feat, *spos, *cpos;
    const int *nbr, *cnt;
    const unsigned long long* total;
    const float* ro0[9];   // W, b, gamma, a, c, mean, invstd, dgamma, dbeta of block 0 (read-only in this pass)
    const float* ro1[9];   // ... of block 1 (= block 0 when nl == 1)
    const float* dout;
    const int* arg_in;
};
typedef const SaK __attribute__((address_space(4)))* SaKp;
__device__ __forceinline__ SaKp opaque_k(SaKp k) {
    uint64_t v = (uint64_t)k;
    asm volatile("" : "+s"(v));
    return (SaKp)v;
}
__device__ __forceinline__ BlkRO make_ro(const float* const __attribute__((address_space(4)))* t) {
    return BlkRO{as_const(t[0]), as_const(t[1]), as_const(t[2]), as_const(t[3]), as_const(t[4]), as_const(t[5]),
                 as_const(t[6]), as_const(t[7]), as_const(t[8])};
}

enum { PASS_STATS0 = 0, PASS_MAIN = 1, PASS_BWD_C = 2, PASS_BWD_D = 3 };

template <int CF>
__device__ __forceinline__ void load_msg(const float* __restrict__ feat, int feat_stride, const float* __restrict__ spos,
                                         int spos_stride, size_t src_row, float cx, float cy, float cz,
                                         float (&u)[CF + 3]) {
    const float4* f = reinterpret_cast<const float4*>(feat + src_row * feat_stride);
#pragma unroll
    for (int q = 0; q < CF / 4; ++q) {
        const float4 v = f[q];
        u[4 * q + 0] = v.x;
        u[4 * q + 1] = v.y;
        u[4 * q + 2] = v.z;
        u[4 * q + 3] = v.w;
    }
    const float4 sp = *reinterpret_cast<const float4*>(spos + src_row * spos_stride);
    u[CF + 0] = sp.x - cx;  // pos_j - pos_i  (PointConv.message)
    u[CF + 1] = sp.y - cy;
    u[CF + 2] = sp.z - cz;
}

template <int C>
__device__ __forceinline__ void affine(cfp a, cfp c, const float (&h)[C], float (&y)[C]) {
#pragma unroll
    for (int o = 0; o < C; ++o) y[o] = fmaf(a[o], h[o], c[o]);
}

// d loss / d pre-activation of a (Linear->ReLU->BN) block from d loss / d BN-output, training-mode BN backward:
//   dh = gamma*invstd * (dy - dbeta/E - xhat*dgamma/E),  dpre = dh * [h > 0]
template <int C>
__device__ __forceinline__ void bn_relu_bwd(const BlkRO& k, const float (&h)[C], const float (&dy)[C], float invE,
                                            bool valid, float (&dp)[C]) {
#pragma unroll
    for (int o = 0; o < C; ++o) {
        const float is = k.invstd[o];
        const float xh = (h[o] - k.mean[o]) * is;
        const float dh = k.gamma[o] * is * (dy[o] - k.dbeta[o] * invE - xh * k.dgamma[o] * invE);
        dp[o] = (valid && h[o] > 0.f) ? dh : 0.f;
    }
}

template <int CF, int NL, int C1, int C2, int PASS>
__global__ __launch_bounds__(256) void sa_pass_kernel(const SaK karg) {
    (void)karg;
    const SaKp kbase = (SaKp)__builtin_amdgcn_kernarg_segment_ptr();
    struct { int B, Nsrc, M, cap, feat_stride, spos_stride; } p;   // the sizes; pointers are fetched where they are used
    p.B = kbase->p.B; p.Nsrc = kbase->p.Nsrc; p.M = kbase->p.M; p.cap = kbase->p.cap;
    p.feat_stride = kbase->p.feat_stride; p.spos_stride = kbase->p.spos_stride;
    constexpr int CIN = CF + 3;

    constexpr int CL = NL == 2 ? C2 : C1;  // width of the last block
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float s_red[128];  // block-level reduction of the batch statistics (2 x up to 64 channels)
    const int lane = threadIdx.x & 63;
    const int wave_in_blk = threadIdx.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wave_in_blk));
    const int nwaves = gridDim.x * 4;
    const int ncent = p.B * p.M;

    float invE = 0.f;
    if constexpr (PASS == PASS_BWD_C || PASS == PASS_BWD_D) {
        const unsigned long long e = *kbase->total;
        invE = e > 0 ? (float)(1.0 / (double)e) : 0.f;
    }

    // per-wave accumulators
    float ssum[PASS == PASS_STATS0 ? C1 : CL], ssq[PASS == PASS_STATS0 ? C1 : CL];
#pragma unroll
    for (int o = 0; o < (PASS == PASS_STATS0 ? C1 : CL); ++o) ssum[o] = ssq[o] = 0.f;

    // backward accumulators
    constexpr bool BWD_LAST = (PASS == PASS_BWD_C && NL == 2) || (PASS == PASS_BWD_D && NL == 1);
    constexpr bool BWD_FIRST2 = (PASS == PASS_BWD_D && NL == 2);
    using AccLast = OuterAcc<CL, NL == 2 ? C1 : CIN>;
    using AccFirst = OuterAcc<C1, CIN>;
    constexpr int LDS_PER_WAVE = BWD_LAST ? AccLast::LDS_FLOATS : (BWD_FIRST2 ? AccFirst::LDS_FLOATS : 0);
    float* lds = smem + wave_in_blk * LDS_PER_WAVE;
    AccLast accL;
    AccFirst accF;
    float dbias[BWD_LAST ? CL : (BWD_FIRST2 ? C1 : 1)];
    float dbeta0[(PASS == PASS_BWD_C && NL == 2) ? C1 : 1], dgamma0[(PASS == PASS_BWD_C && NL == 2) ? C1 : 1];
    if constexpr (BWD_LAST) accL.init(lds);
    if constexpr (BWD_FIRST2) accF.init(lds);
#pragma unroll
    for (int o = 0; o < (BWD_LAST ? CL : (BWD_FIRST2 ? C1 : 1)); ++o) dbias[o] = 0.f;
#pragma unroll
    for (int o = 0; o < ((PASS == PASS_BWD_C && NL == 2) ? C1 : 1); ++o) dbeta0[o] = dgamma0[o] = 0.f;

    for (int ci = wave; ci < ncent; ci += nwaves) {
        const SaKp kc = opaque_k(kbase);
        const int b = ci / p.M;
        const int n = kc->cnt[ci];
        const float4 cp = reinterpret_cast<const float4*>(kc->cpos)[ci];
        const int* nl = kc->nbr + (size_t)ci * p.cap;
        float best[CL];
        int barg[CL];
#pragma unroll
        for (int o = 0; o < CL; ++o) {
            best[o] = -INFINITY;
            barg[o] = -1;
        }
        for (int e0 = 0; e0 < n; e0 += 64) {
            const SaKp k = opaque_k(kbase);                     // pointers and weights are re-read inside the loop
            const BlkRO r0 = make_ro(k->ro0), r1 = make_ro(k->ro1);
            const BlkRO& rl = NL == 2 ? r1 : r0;                // last block, read-only side
            const float* feat = k->feat;
            const float* spos = k->spos;
            const float* dout = k->dout;
            const int* arg_in = k->arg_in;
            const int e = e0 + lane;
            const bool valid = e < n;
            const int j = nl[valid ? e : 0];
            const size_t src_row = (size_t)b * p.Nsrc + j;
            float u[CIN];
            load_msg<CF>(feat, p.feat_stride, spos, p.spos_stride, src_row, cp.x, cp.y, cp.z, u);
            float h1[C1];
            dense<CIN, C1, true>(r0.W, r0.b, u, h1);

            if constexpr (PASS == PASS_STATS0) {
                if (valid) {
#pragma unroll
                    for (int o = 0; o < C1; ++o) {
                        ssum[o] += h1[o];
                        ssq[o] = fmaf(h1[o], h1[o], ssq[o]);
                    }
                }
            } else {
            // ---- activations of the last block
            float y1[NL == 2 ? C1 : 1];
            float hl[CL];
            if constexpr (NL == 2) {
                affine<C1>(r0.a, r0.c, h1, y1);
                dense<C1, C2, true>(r1.W, r1.b, y1, hl);
            } else {
#pragma unroll
                for (int o = 0; o < CL; ++o) hl[o] = h1[o];
            }

            if constexpr (PASS == PASS_MAIN) {
                if (valid) {
#pragma unroll
                    for (int o = 0; o < CL; ++o) {
                        ssum[o] += hl[o];
                        ssq[o] = fmaf(hl[o], hl[o], ssq[o]);
                        const float s = rl.gamma[o] < 0.f ? -hl[o] : hl[o];
                        if (s > best[o]) {
                            best[o] = s;
                            barg[o] = e;
                        }
                    }
                }
            } else {
            // ---- backward passes: d loss / d (BN output of the last block) is non-zero only on the extremum slot
            float dyl[CL], dpl[CL];
#pragma unroll
            for (int o = 0; o < CL; ++o)
                dyl[o] = (valid && arg_in[(size_t)ci * CL + o] == e) ? dout[(size_t)ci * CL + o] : 0.f;
            bn_relu_bwd<CL>(rl, hl, dyl, invE, valid, dpl);

            if constexpr (NL == 2) {
                float dy1[C1];
                dense_t<C1, C2, C1>(r1.W, dpl, dy1);
                if constexpr (PASS == PASS_BWD_C) {
                    accL.add(lds, dpl, y1);
#pragma unroll
                    for (int o = 0; o < CL; ++o) dbias[o] += dpl[o];
#pragma unroll
                    for (int k = 0; k < C1; ++k) {
                        dbeta0[k] += dy1[k];
                        dgamma0[k] = fmaf(dy1[k], (h1[k] - r0.mean[k]) * r0.invstd[k], dgamma0[k]);
                    }
                } else {  // PASS_BWD_D
                    float dp1[C1];
                    bn_relu_bwd<C1>(r0, h1, dy1, invE, valid, dp1);
                    accF.add(lds, dp1, u);
#pragma unroll
                    for (int k = 0; k < C1; ++k) dbias[k] += dp1[k];
                    float* dfeat = k->p.dfeat;
                    if (dfeat) {
                        float du[CF];
                        dense_t<CIN, C1, CF>(r0.W, dp1, du);
                        if (valid) {
#pragma unroll
                            for (int q = 0; q < CF; ++q) atomicAdd(&dfeat[src_row * CF + q], du[q]);
                        }
                    }
                }
            } else {  // NL == 1, PASS_BWD_D: the single block
                accL.add(lds, dpl, u);
#pragma unroll
                for (int o = 0; o < CL; ++o) dbias[o] += dpl[o];
                float* dfeat = k->p.dfeat;
                if (dfeat) {
                    float du[CF];
                    dense_t<CIN, C1, CF>(r0.W, dpl, du);
                    if (valid) {
#pragma unroll
                        for (int q = 0; q < CF; ++q) atomicAdd(&dfeat[src_row * CF + q], du[q]);
                    }
                }
            }
            }  // backward passes
            }  // not STATS0
        }

        if constexpr (PASS == PASS_MAIN) {
            // cross-lane extremum: value by wave max, slot = that of the lowest lane attaining it
            const BlkRO rl = make_ro(NL == 2 ? kc->ro1 : kc->ro0);
            float my_ext = 0.f;
            int my_arg = -1;
#pragma unroll
            for (int o = 0; o < CL; ++o) {
                const float m = wave_max(best[o]);
                const unsigned long long bal = __ballot(best[o] == m);
                const int src = __ffsll((long long)bal) - 1;
                const int a = __shfl(barg[o], src);
                if (lane == o) {
                    my_ext = (rl.gamma[o] < 0.f) ? -m : m;
                    my_arg = a;
                }
            }
            if (lane < CL) {
                kc->p.ext[(size_t)ci * CL + lane] = n > 0 ? my_ext : 0.f;
                kc->p.arg[(size_t)ci * CL + lane] = n > 0 ? my_arg : -1;
            }
        }
    }

    // ---- everything that leaves the kernel is first reduced over the workgroup (see mlp.h)
    const SaKp ke = opaque_k(kbase);   // the write-side pointers are only needed now
    if constexpr (PASS == PASS_STATS0) stats_to_slot<C1>(ssum, ssq, s_red, ke->p.k0.slots);
    if constexpr (PASS == PASS_MAIN) {
        float* slots = NL == 2 ? ke->p.k1.slots : ke->p.k0.slots;
        if (slots) stats_to_slot<CL>(ssum, ssq, s_red, slots);
    }
    if constexpr (PASS == PASS_BWD_C || PASS == PASS_BWD_D) {
        constexpr int CKL = NL == 2 ? C1 : CIN;                  // columns of the last block's dW
        constexpr int NW = BWD_LAST ? CL * CKL : C1 * CIN;       // dW image
        constexpr int NB = BWD_LAST ? CL : C1;                   // bias gradient
        constexpr int NG = (PASS == PASS_BWD_C && NL == 2) ? 2 * C1 : 0;  // first BN's dbeta | dgamma
        float* red = smem;                                       // the wave-private staging regions are free now
        __syncthreads();
        for (int i = threadIdx.x; i < NW + NB + NG; i += 256) red[i] = 0.f;
        __syncthreads();
        if constexpr (BWD_LAST) accL.flush_lds(red);
        if constexpr (BWD_FIRST2) accF.flush_lds(red);
        sums_to_lds<NB>(dbias, red + NW);
        if constexpr (NG > 0) {
            sums_to_lds<C1>(dbeta0, red + NW + NB);
            sums_to_lds<C1>(dgamma0, red + NW + NB + C1);
        }
        __syncthreads();
        const bool last1 = BWD_LAST && NL == 2;   // which block's dW/db this pass produces
        float* dWp = last1 ? ke->p.k1.dW : ke->p.k0.dW;
        float* dbp = last1 ? ke->p.k1.db : ke->p.k0.db;
        float* dbeta0p = ke->p.k0.dbeta;
        float* dgamma0p = ke->p.k0.dgamma;
        for (int i = threadIdx.x; i < NW + NB + NG; i += 256) {
            const float v = red[i];
            if (v == 0.f) continue;
            if (i < NW) atomicAdd(&dWp[i], v);
            else if (i < NW + NB) atomicAdd(&dbp[i - NW], v);
            else if (i < NW + NB + C1) atomicAdd(&dbeta0p[i - NW - NB], v);
            else atomicAdd(&dgamma0p[i - NW - NB - C1], v);
        }
    }
}

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
    for (int r = blockIdx.x * G + g; g < G && r < rows; r += gridDim.x * G) {
        const size_t i = (size_t)r * C + o;
        if (arg[i] >= 0) {
            const float gch = dout[i];
            sb += gch;
            sg = fmaf(gch, (ext[i] - mu) * is, sg);
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

BlkDev to_dev(const sn2_block& k) {
    BlkDev d;
    d.slots = k.stat_slots; d.dW = k.dW; d.db = k.db; d.dgamma = k.dgamma; d.dbeta = k.dbeta;
    return d;
}

SaDev to_dev(const sn2_sa* p) {
    SaDev d;
    d.B = p->B; d.Nsrc = p->Nsrc; d.M = p->M; d.cap = p->cap;
    d.feat_stride = p->feat_stride; d.spos_stride = p->spos_stride;
    d.k0 = to_dev(p->blk[0]);
    d.k1 = to_dev(p->blk[p->nl == 2 ? 1 : 0]);
    d.ext = p->ext; d.arg = p->arg; d.dfeat = p->dfeat;
    return d;
}

int check(const sn2_sa* p) {
    if (!p || p->B <= 0 || p->Nsrc <= 0 || p->M <= 0 || p->cap <= 0) return SN2_EINVAL;
    if (!p->feat || !p->spos || !p->cpos || !p->nbr || !p->cnt || !p->total || !p->ext || !p->arg || !p->out)
        return SN2_EINVAL;
    if (p->feat_stride < p->cf || (p->feat_stride & 3) || p->spos_stride < 4 || (p->spos_stride & 3)) return SN2_EINVAL;
    const bool sa1 = p->cf == 8 && p->nl == 2 && p->blk[0].cin == 11 && p->blk[0].cout == 16 && p->blk[1].cin == 16 &&
                     p->blk[1].cout == 16;
    const bool sa2 = p->cf == 16 && p->nl == 1 && p->blk[0].cin == 19 && p->blk[0].cout == 32;
    return (sa1 || sa2) ? 0 : SN2_ELIMIT;
}

template <int CF, int NL, int C1, int C2, int PASS>
int launch_pass(const sn2_sa* p, int training, hipStream_t st, int* nblocks_out = nullptr) {
    constexpr int CIN = CF + 3, CL = NL == 2 ? C2 : C1;
    constexpr bool BWD = PASS == PASS_BWD_C || PASS == PASS_BWD_D;
    constexpr bool BWD_LAST = (PASS == PASS_BWD_C && NL == 2) || (PASS == PASS_BWD_D && NL == 1);
    constexpr bool BWD_FIRST2 = (PASS == PASS_BWD_D && NL == 2);
    constexpr int LDS_PER_WAVE = BWD_LAST ? OuterAcc<CL, NL == 2 ? C1 : CIN>::LDS_FLOATS
                                          : (BWD_FIRST2 ? OuterAcc<C1, CIN>::LDS_FLOATS : 0);
    SaDev d = to_dev(p);
    if (PASS == PASS_MAIN && !training) (NL == 2 ? d.k1 : d.k0).slots = nullptr;  // eval: no batch statistics
    const long ncent = (long)d.B * d.M;
    // forward passes: as many waves as centroids (up to 8 waves per SIMD worth); backward passes keep MFMA accumulators
    // per wave for the whole kernel, so use a fixed persistent grid and let each wave walk many centroids.
    // forward: one statistics slot per workgroup (<= SN2_STAT_SLOTS); backward: one workgroup per CU, each wave walks
    // many centroids and keeps its MFMA accumulators in registers for the whole kernel
    int blocks = sn2_cdiv(ncent, 4);
    const int cap_blocks = SN2_STAT_SLOTS;
    if (blocks > cap_blocks) blocks = cap_blocks;
    if (nblocks_out) *nblocks_out = blocks;
    constexpr size_t lds_bytes = (size_t)LDS_PER_WAVE * 4 * sizeof(float);
    if (lds_bytes > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sa_pass_kernel<CF, NL, C1, C2, PASS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    const sn2_block& k0 = p->blk[0];
    const sn2_block& k1 = p->blk[NL == 2 ? 1 : 0];
    // read-only views.  A gradient array that THIS pass accumulates into (block 0's dgamma/dbeta in pass C) is not
    // handed in as a read-only table.
    const bool rd0 = (PASS == PASS_BWD_D);
    const bool rd1 = BWD;
    SaK k;
    k.p = d;
    k.feat = p->feat; k.spos = p->spos; k.cpos = p->cpos; k.nbr = p->nbr; k.cnt = p->cnt; k.total = p->total;
    const float* t0[9] = {k0.W, k0.b, k0.gamma, k0.a, k0.c, k0.mean, k0.invstd, rd0 ? k0.dgamma : nullptr,
                          rd0 ? k0.dbeta : nullptr};
    const float* t1[9] = {k1.W, k1.b, k1.gamma, k1.a, k1.c, k1.mean, k1.invstd, rd1 ? k1.dgamma : nullptr,
                          rd1 ? k1.dbeta : nullptr};
    for (int i = 0; i < 9; ++i) { k.ro0[i] = t0[i]; k.ro1[i] = t1[i]; }
    k.dout = BWD ? p->dout : nullptr;
    k.arg_in = BWD ? p->arg : nullptr;
    hipLaunchKernelGGL((sa_pass_kernel<CF, NL, C1, C2, PASS>), dim3(blocks), dim3(256), lds_bytes, st, k);
    SN2_RETURN_LAUNCH();
}

template <int CF, int NL, int C1, int C2>
int forward_t(const sn2_sa* p, int training, hipStream_t st) {
    const sn2_block* last = &p->blk[NL - 1];
    int nb = 0;
    if (NL == 2) {
        if (training) SN2_TRY((launch_pass<CF, NL, C1, C2, PASS_STATS0>(p, training, st, &nb)));
        SN2_TRY(sn2_bn_finalize(&p->blk[0], nb, p->total, 0, training, st));
    }
    SN2_TRY((launch_pass<CF, NL, C1, C2, PASS_MAIN>(p, training, st, &nb)));
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
    if constexpr (NL == 2) SN2_TRY((launch_pass<CF, NL, C1, C2, PASS_BWD_C>(p, 1, st)));
    SN2_TRY((launch_pass<CF, NL, C1, C2, PASS_BWD_D>(p, 1, st)));
    return 0;
}

}  // namespace

extern "C" int sn2_sa_forward(const sn2_sa* p, int training, void* stream) {
    SN2_TRY(check(p));
    if (p->nl == 2) return forward_t<8, 2, 16, 16>(p, training, (hipStream_t)stream);
    return forward_t<16, 1, 32, 32>(p, training, (hipStream_t)stream);
}

extern "C" int sn2_sa_backward(const sn2_sa* p, void* stream) {
    SN2_TRY(check(p));
    if (!p->dout) return SN2_EINVAL;
    if (p->nl == 2) return backward_t<8, 2, 16, 16>(p, (hipStream_t)stream);
    return backward_t<16, 1, 32, 32>(p, (hipStream_t)stream);
}
