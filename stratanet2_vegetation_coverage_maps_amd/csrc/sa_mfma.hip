// sa_mfma.hip -- set-abstraction FORWARD passes with the shared MLP on the matrix cores.
// (PointConv(local_nn) message + max aggregation, /root/reference/model/point_net2.py:19,21-29.)
//
// Layout: one wave per work item (sn2_sa_order), 64 messages per step = 4 tiles of 16.  A SOLO item is one centroid with a
// long list: tile t takes messages e0 + 16 t + c, 64 per step.  A QUAD item is four centroids with short lists of similar
// length: tile t takes the next 16 messages of the quad's t-th centroid.  (Ball sizes are very uneven -- median 5, mean
// 25, maximum 261 at C2: four tiles of one centroid ran 32 % full.)  lane = (q = lane>>4, c = lane&15).
// v_mfma_f32_16x16x4_f32 computes D[i][j] += sum_k A[i][k] B[k][j] with
//     A: lane holds A[i = c][k = q],   B: lane holds B[k = q][j = c],   D: lane holds D[i = 4q + r][j = c], r = 0..3.
// Here i = output channel, j = message, k = input channel, i.e. D = W . U:
//   * the weights are the A operands: W[16 io + c][4 kb + q] -- a handful of VGPRs loaded ONCE per kernel, so no weight
//     ever streams through SGPRs (the VALU version spent 2 v_readlane per FMA on spilled SGPRs);
//   * the B operand of layer 1 is the message input gathered straight into MFMA layout: lane (q,c) of tile t loads
//     element 4 kb + q of the source row of message 16 t + c (16 rows x 16 B per load instruction); the last k-block is
//     [dx, dy, dz, 1] so the bias rides in the weight matrix;
//   * a layer's output tile is directly the next layer's B operand: contraction block (is, kb) takes register kb of
//     every lane, i.e. channels {16 is + 4 q + kb}; the A registers of the next layer are pre-permuted accordingly, so
//     nothing moves between lanes;
//   * ReLU, the BatchNorm affine, the batch statistics and the signed running extremum are elementwise on D registers
//     with per-lane constants (4 channels per lane);
//   * the max over a centroid's messages = elementwise over tiles and steps, then a 16-lane DPP row reduction (the 16
//     lanes of a DPP row are exactly the 16 messages of a tile): value first, then the lowest slot among the lanes that
//     attain it -- "first maximum wins" in ascending source index, as torch_scatter.
#include "mlp.h"

namespace {

struct SaFwdArgs {
    int B, Nsrc, M, cap, feat_stride, spos_stride, group;
    const float *feat, *spos, *cpos;
    const int *nbr, *cnt, *order;
    const float *W0, *b0, *a0, *c0, *gamma0;
    const float *W1, *b1, *gamma1;
    float *slots;  // statistics slots of the block this pass measures, or nullptr
    float* ext;
    int* arg;
    // EVAL (STATS = false): the last block's BatchNorm on its running statistics, (a, c) finalised BEFORE the launch, and the
    // level's output a ext + c written by this kernel (no ext / arg arrays, no pass over them afterwards)
    const float *al, *cl;
    float* out;
};

// every lane ends up with the reduction over its DPP row (16 lanes)
__device__ __forceinline__ float row_max(float v) {
    v = fmaxf(v, __int_as_float(SN2_DPP(__float_as_int(v), 0xB1, 0xF)));
    v = fmaxf(v, __int_as_float(SN2_DPP(__float_as_int(v), 0x4E, 0xF)));
    v = fmaxf(v, __int_as_float(SN2_DPP(__float_as_int(v), 0x141, 0xF)));
    v = fmaxf(v, __int_as_float(SN2_DPP(__float_as_int(v), 0x140, 0xF)));
    return v;
}
__device__ __forceinline__ unsigned row_min_u32(unsigned v) {
    v = min(v, (unsigned)SN2_DPP((int)v, 0xB1, 0xF));
    v = min(v, (unsigned)SN2_DPP((int)v, 0x4E, 0xF));
    v = min(v, (unsigned)SN2_DPP((int)v, 0x141, 0xF));
    v = min(v, (unsigned)SN2_DPP((int)v, 0x140, 0xF));
    return v;
}
__device__ __forceinline__ float row_sum(float v) {
    v += __int_as_float(SN2_DPP(__float_as_int(v), 0xB1, 0xF));
    v += __int_as_float(SN2_DPP(__float_as_int(v), 0x4E, 0xF));
    v += __int_as_float(SN2_DPP(__float_as_int(v), 0x141, 0xF));
    v += __int_as_float(SN2_DPP(__float_as_int(v), 0x140, 0xF));
    return v;
}


// Reductions over a SEGMENT of a DPP row: W = 16 (a tile is one centroid's), 8 or 4 lanes (two / four short lists share a
// tile: OCT and HEX items).  W is wave-uniform; every lane ends up with its segment's result.
__device__ __forceinline__ float seg_max(float v, int W) {
    v = fmaxf(v, __int_as_float(SN2_DPP(__float_as_int(v), 0xB1, 0xF)));
    v = fmaxf(v, __int_as_float(SN2_DPP(__float_as_int(v), 0x4E, 0xF)));
    if (W > 4) v = fmaxf(v, __int_as_float(SN2_DPP(__float_as_int(v), 0x141, 0xF)));
    if (W > 8) v = fmaxf(v, __int_as_float(SN2_DPP(__float_as_int(v), 0x140, 0xF)));
    return v;
}
__device__ __forceinline__ unsigned seg_min_u32(unsigned v, int W) {
    v = min(v, (unsigned)SN2_DPP((int)v, 0xB1, 0xF));
    v = min(v, (unsigned)SN2_DPP((int)v, 0x4E, 0xF));
    if (W > 4) v = min(v, (unsigned)SN2_DPP((int)v, 0x141, 0xF));
    if (W > 8) v = min(v, (unsigned)SN2_DPP((int)v, 0x140, 0xF));
    return v;
}

// The first links of a wave's chain -- trailer (how many positions there are) -> the position's record -> counts, first indices ->
// rows -- asked for at the very top of the kernels, in front of the ~50 loads of the weights and BatchNorm constants: the record of
// the wave's FIRST position is read speculatively as a SOLO / QUAD record (order + 4 qi: inside the table for every wave), which it is
// for the heavy items that decide how long the kernel runs; a packed (light) position reads its own record later.  With barely more
// positions than waves (2 560 against 3 072 at config 2) a kernel is one such chain per wave: two round trips fewer (round 5).
struct SaFirst {
    int nA, nB;
    int rec[4];
};
__device__ __forceinline__ SaFirst sa_first(const int* __restrict__ order, const int* __restrict__ fallback, int B, int M, int wave) {
    SaFirst f;
    const int ncent = B * M;
    const int* trailer = order ? order + (size_t)4 * ncent + (size_t)16 * B * SN2_SA_PACKED_ITEMS(M) : nullptr;
    const int* base = order ? order : fallback;                // (unconditional loads: a valid address either way)
    const int* tr = trailer ? trailer : fallback;
    const int t0 = tr[0], t1 = tr[trailer ? 1 : 0];
    const int w4 = 4 * (wave < ncent ? wave : 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) f.rec[t] = base[order ? w4 + t : 0];      // (without a table: word 0 of the fallback, never used)
    f.nA = order ? t0 : 0, f.nB = order ? t1 : 0;
    return f;
}

// One work item of sn2_sa_order, as every lane sees it.  lane = (q, c), tile t: the message slot (t, c) belongs to centroid
// lc[t], is entry le0 + eoff * t + (steps done) * estep of its neighbour list, which has ln[t] entries (-1: no centroid).
//   SOLO  one centroid, 64 entries per step (tile t: entries 16 t + c)        QUAD  four centroids, 16 entries per step each
//   OCT   eight centroids with 5..8 neighbours, half a tile each, one step     HEX   sixteen with 0..4, a quarter tile each
// W = lanes per centroid in a tile (16, 8, 4); bt[t] = the plot of tile t's centroids (wave-uniform, as estep, eoff, W, nmax).
// Returns false for an empty position.
struct SaItem {
    int lc[4], ln[4], le0;
    int bt[4], estep, eoff, W, nmax;
    bool solo;
};
// (`pre`: the four ints at order + 4 qi, loaded ahead by the caller -- the record of position qi if it is a SOLO / QUAD position
// of a table whose plots are taken all at once; nullptr: none)
__device__ __forceinline__ bool sa_item(SaItem& it, const int* __restrict__ order, const int* __restrict__ cnt, int B, int M,
                                        int nA, int nB, int G, int qi, int c, const int* pre = nullptr) {
    const int ncent = B * M;
    if (!order) {                                            // no order table: quads in index order
        it.solo = false, it.estep = 16, it.eoff = 0, it.W = 16, it.nmax = 0, it.le0 = c;
        // (the four counts are loaded UNCONDITIONALLY -- slot 0 for a missing centroid, masked afterwards -- and together: as
        // `ci >= 0 ? cnt[ci] : -1` each was a load under a branch, waited for before the next was issued: four round trips)
        int cn[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ci = 4 * qi + t < ncent ? 4 * qi + t : -1;
            it.lc[t] = ci >= 0 ? ci : 0;
            cn[t] = cnt[it.lc[t]];
            it.ln[t] = ci;
            it.bt[t] = it.lc[t] / M;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            it.ln[t] = it.ln[t] >= 0 ? cn[t] : -1;
            it.nmax = max(it.nmax, __builtin_amdgcn_readfirstlane(it.ln[t]));
        }
        return it.nmax > 0;
    }
    // turn qi -> (plot b, item k): the plots are taken G at a time (all of them when they are few), the items of a group
    // interleaved heaviest first; the SOLO / QUAD items of all groups come before the packed ones.  (The table itself keeps
    // item k of plot b at position k B + b.)  With hundreds of plots -- the parcel loop's 256 -- the waves of a round then
    // gather from G plots' feature rows, which stay in L2, instead of from all of them.
    const int ngrp = (B + G - 1) / G;
    const bool packed = qi >= ngrp * nA * G;
    const int qq = packed ? qi - ngrp * nA * G : qi, nk = packed ? nB : nA;
    const int g = qq / (nk * G), rem = qq - g * (nk * G), k = rem / G, pb = g * G + (rem - k * G);
    if (pb >= B) return false;
    const int qp = k * B + pb;
    const int* rec = packed ? order + (size_t)4 * ncent + (size_t)16 * qp : order + (size_t)4 * qp;
    int ent[4];
    if (pre && !packed && G == B) {
#pragma unroll
        for (int t = 0; t < 4; ++t) ent[t] = pre[t];                         // (qp == qi there)
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) ent[t] = rec[packed ? 4 * t + (c >> 2) : t];
    }
    const int e0 = __builtin_amdgcn_readfirstlane(ent[0]);
    if (e0 < 0) return false;
    it.solo = !packed && (e0 & SN2_SA_SOLO_FLAG) != 0;
    it.W = !packed ? 16 : ((e0 & SN2_SA_OCT_FLAG) ? 8 : 4);
    it.estep = it.solo ? 64 : 16;
    it.eoff = it.solo ? 16 : 0;
    it.le0 = c & (it.W - 1);
    it.nmax = 0;
    int cn[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const bool some = ent[t] >= 0;
        it.lc[t] = some ? (ent[t] & ~(SN2_SA_SOLO_FLAG | SN2_SA_OCT_FLAG)) : 0;
        cn[t] = cnt[it.lc[t]];                               // (unconditional, all four in flight together: see above)
        it.bt[t] = pb;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        it.ln[t] = ent[t] >= 0 ? cn[t] : -1;
        if (!packed) it.nmax = max(it.nmax, __builtin_amdgcn_readfirstlane(it.ln[t]));
    }
    if (packed) it.nmax = it.W;                              // one step (lists of at most W entries)
    return it.nmax > 0;
}

// PASS 0: statistics of block 0 only (nl == 2);  PASS 1: full forward, statistics of the last block, extremum
// STATS = false (an EVAL pass: BatchNorm on its running statistics, nothing kept for a backward): no statistic sums and no
// arg-max slots -- a quarter of the vector instructions of a step, which is what bounds the parcel loop's SA1 pass (64 M
// messages per launch) -- and (round 5) the level's OUTPUT a ext + c straight from the kernel: (a, c) of the last block
// are known before the launch, so the ext / arg arrays and the pass over them (parcel loop: 410 MB per launch for SA1
// where 82 MB of output do) fall away
template <int CF, int NL, int C1, int C2, int PASS, bool BF16, bool STATS = true>
__global__ __launch_bounds__(256, (PASS == 1 && CF == 8) ? 3 : 1) void sa_mfma_fwd_kernel(const SaFwdArgs a) {
    static_assert(STATS || PASS == 1, "pass 0 IS the statistics pass");
    constexpr int CIN = CF + 3, KB1 = CF / 4 + 1, TO1 = C1 / 16, TO2 = C2 / 16;
    constexpr int CL = NL == 2 ? C2 : C1, TOL = CL / 16, CS = PASS == 0 ? C1 : CL, TOS = CS / 16;
    static_assert(CF % 4 == 0 && C1 % 16 == 0 && C2 % 16 == 0, "tile shapes");
    __shared__ float s_red[4][2 * CS];          // per wave: added in wave order => the statistics are the same on every run
    const int lane = threadIdx.x & 63, q = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int nwaves = gridDim.x * 4;
    const int ncent = a.B * a.M;
    const SaFirst first = sa_first(a.order, a.cnt, a.B, a.M, wave);

    // ---- operand registers, loaded once
    // the bias of the first layer rides in the weight matrix against the constant-1 column of the gathered input -- in fp32.
    // With bf16 operands it stays OUT of the contraction (it would be rounded to bfloat16) and starts the accumulator.
    float A1[TO1][KB1], bias1[TO1][4];
#pragma unroll
    for (int io = 0; io < TO1; ++io) {
#pragma unroll
        for (int kb = 0; kb < KB1; ++kb) {
            const int k = 4 * kb + q, o = 16 * io + c;
            A1[io][kb] = k < CIN ? a.W0[o * CIN + k] : ((k == CIN && !BF16) ? a.b0[o] : 0.f);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) bias1[io][r] = BF16 ? a.b0[16 * io + 4 * q + r] : 0.f;
    }
    float A2[NL == 2 ? TO2 : 1][TO1][4], bias2[NL == 2 ? TO2 : 1][4], a1v[TO1][4], c1v[TO1][4];
    if constexpr (NL == 2 && PASS == 1) {
#pragma unroll
        for (int io = 0; io < TO2; ++io) {
#pragma unroll
            for (int is = 0; is < TO1; ++is)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) A2[io][is][kb] = a.W1[(16 * io + c) * C1 + 16 * is + 4 * q + kb];
#pragma unroll
            for (int r = 0; r < 4; ++r) bias2[io][r] = a.b1[16 * io + 4 * q + r];
        }
#pragma unroll
        for (int is = 0; is < TO1; ++is)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a1v[is][r] = a.a0[16 * is + 4 * q + r];
                c1v[is][r] = a.c0[16 * is + 4 * q + r];
            }
    }
    float sgn[TOL][4];
    if constexpr (PASS == 1) {
        const float* gl = NL == 2 ? a.gamma1 : a.gamma0;
#pragma unroll
        for (int io = 0; io < TOL; ++io)
#pragma unroll
            for (int r = 0; r < 4; ++r) sgn[io][r] = gl[16 * io + 4 * q + r] < 0.f ? -1.f : 1.f;
    }
    float alv[STATS ? 1 : TOL][4], clv[STATS ? 1 : TOL][4];
    if constexpr (!STATS) {
#pragma unroll
        for (int io = 0; io < TOL; ++io)
#pragma unroll
            for (int r = 0; r < 4; ++r) alv[io][r] = a.al[16 * io + 4 * q + r], clv[io][r] = a.cl[16 * io + 4 * q + r];
    }
    float ssum[TOS][4], ssq[TOS][4];
#pragma unroll
    for (int io = 0; io < TOS; ++io)
#pragma unroll
        for (int r = 0; r < 4; ++r) ssum[io][r] = ssq[io][r] = 0.f;

    // positions of the order table: nA per plot for SOLO / QUAD items, then nB per plot for the packed ones (OCT / HEX)
    const int nA = first.nA, nB = first.nB;
    const int G = a.group > 0 && a.group < a.B ? a.group : a.B;
    const int nitems = a.order ? (nA + nB) * G * ((a.B + G - 1) / G) : (ncent + 3) >> 2;
    // items are ordered heaviest first: the waves take them in a snake (round 0: item w, round 1: item 2W-1-w, ...), so the
    // wave with the longest item of a round gets the shortest of the next.  (A shared work counter was tried: 8000 atomics
    // on one address cost more than the imbalance they removed.  Round 5: dealing the positions so that an XCD's waves take
    // the items of TWO of the sixteen plots -- 3 MB of feature rows per L2 instead of four plots' 6 MB on two XCDs each --
    // made all six SA kernels 1-4 % slower: whole plots per XCD are not equal work.)
    for (int round = 0;; ++round) {
        const int qi = round * nwaves + ((round & 1) ? nwaves - 1 - wave : wave);
        if (round * nwaves >= nitems) break;
        if (qi >= nitems) continue;
        SaItem it;
        if (!sa_item(it, a.order, a.cnt, a.B, a.M, nA, nB, G, qi, c, round == 0 ? first.rec : nullptr)) continue;      // an empty position
        const int nmax = it.nmax, estep = it.estep;
        float cpq[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) cpq[t] = a.cpos[(size_t)it.lc[t] * 4 + (q < 3 ? q : 0)];
        float best[4][TOL][4];
        int barg[4][TOL][4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int io = 0; io < TOL; ++io)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    best[t][io][r] = -INFINITY;
                    barg[t][io][r] = 0x7FFFFFFF;
                }
        // the neighbour indices of a step are fetched one step ahead: index load -> row gather -> MFMA chain is what a
        // step waits for, and the first link does not depend on the step before.  The FIRST step's indices are requested
        // together with the counts (the address needs the centroid only; entries past the count are masked afterwards)
        int jn[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = it.le0 + it.eoff * t;
            jn[t] = a.nbr[(size_t)it.lc[t] * a.cap + (e < a.cap ? e : 0)];     // (unconditional: masked below)
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) jn[t] = it.le0 + it.eoff * t < it.ln[t] ? jn[t] : 0;
        for (int e0 = 0; e0 < nmax; e0 += estep) {
            // ---- layer 1: gather straight into the B-operand layout, 4 message tiles
            f32x4 D1[TO1][4];
            bool val[4];
            int jc[4];
            // every load of a step is issued HERE, unconditional, before any of them is waited for: the four tiles' rows and the
            // next step's four indices (from a clamped slot, masked afterwards).  As `en < n ? nbr[..] : 0` the index prefetch was a
            // load under a branch -- the compiler then cannot count what is outstanding and drained everything behind each tile's
            // gathers: four memory round trips per step, one after the other, where one is needed (scripts/isa_scan.py; round 5).
            // (Gathering a step's ROWS one step ahead as well -- scripts/sa_stamps.py has the five-step items that decide the
            // kernels' duration at 3.5 us per step -- made all four SA1 kernels slower, 15.1 -> 18.3 us the statistics pass: most
            // items are ONE step, and each then gathers a second set of rows nobody reads.  In the EVAL variant alone -- the parcel
            // loop: three or four steps per item -- it was slower too: 2139 -> 2241 us per launch for SA1, 701 -> 796 for SA2.)
            float bkt[4][KB1];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                jc[t] = jn[t];
                const int en = e0 + estep + it.le0 + it.eoff * t;
                const bool more = en < it.ln[t];
                const int jr = a.nbr[(size_t)it.lc[t] * a.cap + (more ? en : 0)];
                jn[t] = more ? jr : 0;
                const size_t row = (size_t)it.bt[t] * a.Nsrc + jc[t];
#pragma unroll
                for (int kb = 0; kb < KB1 - 1; ++kb) bkt[t][kb] = a.feat[row * a.feat_stride + 4 * kb + q];
                bkt[t][KB1 - 1] = a.spos[row * a.spos_stride + (q < 3 ? q : 0)];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int e = e0 + it.le0 + it.eoff * t;                            // slot in the list of this lane's centroid
                val[t] = e < it.ln[t];
                float bk[KB1];
#pragma unroll
                for (int kb = 0; kb < KB1 - 1; ++kb) bk[kb] = bkt[t][kb];
                bk[KB1 - 1] = q < 3 ? bkt[t][KB1 - 1] - cpq[t] : 1.0f;   // pos_j - pos_i | bias column
#pragma unroll
                for (int io = 0; io < TO1; ++io) {
                    f32x4 acc = {bias1[io][0], bias1[io][1], bias1[io][2], bias1[io][3]};
                    acc = contract<BF16, KB1>(acc, [&](int kb) { return A1[io][kb]; }, [&](int kb) { return bk[kb]; });
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = fmaxf(acc[r], 0.f);
                    D1[io][t] = acc;
                }
            }
            if constexpr (PASS == 0) {
#pragma unroll
                for (int io = 0; io < TO1; ++io)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float h = val[t] ? D1[io][t][r] : 0.f;
                            ssum[io][r] += h;
                            ssq[io][r] = fmaf(h, h, ssq[io][r]);
                        }
            } else {
                // ---- layer 2 (nl == 2): BN affine of block 0, then the next contraction on the same registers
                f32x4 DL[TOL][4];
                if constexpr (NL == 2) {
#pragma unroll
                    for (int is = 0; is < TO1; ++is)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) D1[is][t][r] = fmaf(a1v[is][r], D1[is][t][r], c1v[is][r]);
#pragma unroll
                    for (int io = 0; io < TO2; ++io)
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            f32x4 acc = {bias2[io][0], bias2[io][1], bias2[io][2], bias2[io][3]};
                            acc = contract<BF16, 4 * TO1>(acc, [&](int kk) { return A2[io][kk >> 2][kk & 3]; },
                                                          [&](int kk) { return D1[kk >> 2][t][kk & 3]; });
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[r] = fmaxf(acc[r], 0.f);
                            DL[io][t] = acc;
                        }
                } else {
#pragma unroll
                    for (int io = 0; io < TOL; ++io)
#pragma unroll
                        for (int t = 0; t < 4; ++t) DL[io][t] = D1[io][t];
                }
                // ---- statistics of the last block + signed running extremum (slot = position in the neighbour list)
#pragma unroll
                for (int io = 0; io < TOL; ++io)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int e = e0 + it.le0 + it.eoff * t;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if constexpr (STATS) {
                                const float h = val[t] ? DL[io][t][r] : 0.f;
                                ssum[io][r] += h;
                                ssq[io][r] = fmaf(h, h, ssq[io][r]);
                            }
                            const float s = sgn[io][r] * DL[io][t][r];
                            if constexpr (STATS) {
                                if (val[t] && s > best[t][io][r]) {
                                    best[t][io][r] = s;
                                    barg[t][io][r] = e;
                                }
                            } else {                     // eval: nobody routes a gradient, the slot of the extremum is not kept
                                best[t][io][r] = fmaxf(best[t][io][r], val[t] ? s : -INFINITY);
                            }
                        }
                    }
            }
        }
        if constexpr (PASS == 1) {
            if (it.solo) {   // the four tiles belong to one centroid: fold them (greater value, then lower slot), tile 0 writes
#pragma unroll
                for (int t = 1; t < 4; ++t)
#pragma unroll
                    for (int io = 0; io < TOL; ++io)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if constexpr (STATS) {
                                const bool take = best[t][io][r] > best[0][io][r] ||
                                                  (best[t][io][r] == best[0][io][r] && barg[t][io][r] < barg[0][io][r]);
                                best[0][io][r] = take ? best[t][io][r] : best[0][io][r];
                                barg[0][io][r] = take ? barg[t][io][r] : barg[0][io][r];
                            } else {
                                best[0][io][r] = fmaxf(best[0][io][r], best[t][io][r]);
                            }
                        }
            }
            const int W = it.W;
            const bool writer = (c & (W - 1)) == 0;          // first lane of a centroid's segment of the tile
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (it.solo && t > 0) continue;
#pragma unroll
                for (int io = 0; io < TOL; ++io) {
                    float ev[4];
                    int av[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float m = seg_max(best[t][io][r], W);
                        ev[r] = it.ln[t] > 0 ? sgn[io][r] * m : 0.f;
                        if constexpr (STATS) {
                            const unsigned am = seg_min_u32(best[t][io][r] == m ? (unsigned)barg[t][io][r] : 0xFFFFFFFFu, W);
                            av[r] = it.ln[t] > 0 ? (int)am : -1;
                        } else {
                            // eval: the level's output itself, as bn_finalize_kernel<true> forms it (0 without neighbours)
                            av[r] = 0;
                            ev[r] = it.ln[t] > 0 ? fmaf(alv[io][r], ev[r], clv[io][r]) : 0.f;
                        }
                    }
                    if (writer && it.ln[t] >= 0) {
                        const size_t o = (size_t)it.lc[t] * CL + 16 * io + 4 * q;
                        if constexpr (STATS) {
                            *reinterpret_cast<float4*>(a.ext + o) = make_float4(ev[0], ev[1], ev[2], ev[3]);
                            *reinterpret_cast<int4*>(a.arg + o) = make_int4(av[0], av[1], av[2], av[3]);
                        } else {
                            *reinterpret_cast<float4*>(a.out + o) = make_float4(ev[0], ev[1], ev[2], ev[3]);
                        }
                    }
                }
            }
        }
    }

    // ---- batch statistics: row sums (over the 16 message lanes), then workgroup slot
    if (STATS && a.slots) {
        // no float atomics here: their order varies from run to run, the BatchNorm statistics with it (1e-7), and now and
        // then a pre-activation next to zero changes sign -- one ReLU mask flip moved a weight gradient by 1 %
        const int wv = threadIdx.x >> 6;
        for (int i = threadIdx.x; i < 4 * 2 * CS; i += 256) (&s_red[0][0])[i] = 0.f;
        __syncthreads();
#pragma unroll
        for (int io = 0; io < TOS; ++io)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s1 = row_sum(ssum[io][r]), s2 = row_sum(ssq[io][r]);
                if (c == 0) {
                    s_red[wv][16 * io + 4 * q + r] = s1;
                    s_red[wv][CS + 16 * io + 4 * q + r] = s2;
                }
            }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * CS; i += 256)
            a.slots[(size_t)blockIdx.x * 2 * CS + i] = (s_red[0][i] + s_red[1][i]) + (s_red[2][i] + s_red[3][i]);
    }
}


// ---------------------------------------------------------------------------------------------- backward passes
// Same layout.  The forward is recomputed on the matrix cores; what is new:
//   * BatchNorm(train)+ReLU backward per element with three per-lane constants per channel,
//         dpre = [h > 0] * (A*dy - C*h + D),  A = gamma*is,  C = gamma*is^2*dgamma/E,  D = gamma*is*(mean*is*dgamma - dbeta)/E
//     (dy is non-zero only on the slot that attained the extremum);
//   * the input gradient of a layer, dy1 = W2^T dp2, is again an MFMA whose B operand is the register tile dp2 as it
//     stands (contraction block (io, kb) = register kb), with pre-permuted A registers W2[16 io + 4 q + kb][16 is + c];
//   * weight gradients contract over MESSAGES: both factors go through a wave-private LDS image [message][channel]
//     (one ds_write_b128 per tile for a register tile) and come back in A/B layout with the message index as K.
struct SaBwdArgs {
    int B, Nsrc, M, cap, feat_stride, spos_stride, group, frozen;
    const float *feat, *spos, *cpos;
    const int *nbr, *cnt, *order;
    const unsigned long long* total;
    const float *W0, *b0, *a0, *c0, *gamma0, *mean0, *invstd0, *dgamma0, *dbeta0;   // dgamma0/dbeta0: read in pass D only
    const float *W1, *b1, *gamma1, *mean1, *invstd1, *dgamma1, *dbeta1;
    const float* dout;
    const int* arg;
    float *dW0, *db0, *dW1, *db1, *dgamma0_out, *dbeta0_out;   // accumulated (atomics)
    int rep_k, rep_stride;                                     // images of (dW, db): sn2_block.grad_replicas
    float* dfeat;
};

#ifdef SN2_SA_STAMPS
// diagnostic build only (never shipped; scripts/sa_stamps.py): per-wave phase stamps of sa_mfma_bwd_kernel<8, 2, 16, 16, 3>
__device__ unsigned long long g_sa_dbg[4096 * 10];
extern "C" int sn2_debug_sa_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sa_dbg), sizeof(g_sa_dbg));
}
#define SASTAMP(i)                                                                                  \
    if (CF == 8 && PASS == 3 && (threadIdx.x & 63) == 0 && wave < 4096) {                              \
        unsigned long long t_;                                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                   \
        g_sa_dbg[wave * 10 + (i)] = t_;                                                              \
    }
#define SACOUNT(i, v) if (CF == 8 && PASS == 3 && (threadIdx.x & 63) == 0 && wave < 4096) g_sa_dbg[wave * 10 + (i)] = (v);
#else
#define SASTAMP(i)
#define SACOUNT(i, v)
#endif

// PASS 2 = "C" (nl == 2): dW/db of block 1, dgamma/dbeta of block 0.   PASS 3 = "D": dW/db of block 0 (+ dfeat).
template <int CF, int NL, int C1, int C2, int PASS, bool BF16>
__global__ __launch_bounds__(256) void sa_mfma_bwd_kernel(const SaBwdArgs a) {
    constexpr int CIN = CF + 3, KB1 = CF / 4 + 1, TO1 = C1 / 16, TO2 = C2 / 16;
    constexpr bool LAST2 = NL == 2 && PASS == 2;             // this pass produces block 1's weight gradient
    constexpr int PO = LAST2 ? C2 : C1;                      // rows of the dW image
    constexpr int QK = LAST2 ? C1 : CIN + 1;                 // columns (block 0: inputs | bias)
    using Acc = OuterAcc<PO, QK>;
    constexpr int PS = Acc::PS, QS = Acc::QS, TP = Acc::TO, TQ = Acc::TK;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, q = lane >> 4, c = lane & 15;
    const int wib = threadIdx.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wib));
    const int nwaves = gridDim.x * 4;
    const int ncent = a.B * a.M;
    const SaFirst first = sa_first(a.order, a.cnt, a.B, a.M, wave);
    float* lds_p = smem + wib * Acc::LDS_FLOATS;
    float* lds_q = lds_p + 64 * PS;
    SASTAMP(0);
    for (int i = lane; i < Acc::LDS_FLOATS; i += 64) lds_p[i] = 0.f;

    // (frozen: the forward ran BatchNorm on its running statistics -- no batch-mean / batch-variance terms: 1 / E := 0)
    const unsigned long long etot = *a.total;
    const float invE = (etot > 0 && !a.frozen) ? (float)(1.0 / (double)etot) : 0.f;

    // ---- operand registers
    float A1[TO1][KB1], bias1[TO1][4];      // bias of the first layer: inside the contraction in fp32, outside with bf16 operands
#pragma unroll
    for (int io = 0; io < TO1; ++io) {
#pragma unroll
        for (int kb = 0; kb < KB1; ++kb) {
            const int k = 4 * kb + q, o = 16 * io + c;
            A1[io][kb] = k < CIN ? a.W0[o * CIN + k] : ((k == CIN && !BF16) ? a.b0[o] : 0.f);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) bias1[io][r] = BF16 ? a.b0[16 * io + 4 * q + r] : 0.f;
    }
    float A2[NL == 2 ? TO2 : 1][TO1][4], A2T[TO1][NL == 2 ? TO2 : 1][4], bias2[NL == 2 ? TO2 : 1][4], a1v[TO1][4], c1v[TO1][4];
    // BN-backward constants: block 0 (index 0) and, for nl == 2, block 1 (index 1)
    float cA0[TO1][4], cC0[TO1][4], cD0[TO1][4], mu0[TO1][4], is0[TO1][4];
    float cA1[NL == 2 ? TO2 : 1][4], cC1[NL == 2 ? TO2 : 1][4], cD1[NL == 2 ? TO2 : 1][4];
#pragma unroll
    for (int is = 0; is < TO1; ++is)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ch = 16 * is + 4 * q + r;
            mu0[is][r] = a.mean0[ch];
            is0[is][r] = a.invstd0[ch];
            a1v[is][r] = NL == 2 ? a.a0[ch] : 0.f;
            c1v[is][r] = NL == 2 ? a.c0[ch] : 0.f;
            if (NL == 1 || PASS == 3) {
                const float g = a.gamma0[ch], dg = a.dgamma0[ch], db = a.dbeta0[ch], m = mu0[is][r], s = is0[is][r];
                cA0[is][r] = g * s;
                cC0[is][r] = g * s * s * dg * invE;
                cD0[is][r] = g * s * (m * s * dg - db) * invE;
            } else {
                cA0[is][r] = cC0[is][r] = cD0[is][r] = 0.f;
            }
        }
    if constexpr (NL == 2) {
#pragma unroll
        for (int io = 0; io < TO2; ++io) {
#pragma unroll
            for (int is = 0; is < TO1; ++is)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    A2[io][is][kb] = a.W1[(16 * io + c) * C1 + 16 * is + 4 * q + kb];
                    A2T[is][io][kb] = a.W1[(16 * io + 4 * q + kb) * C1 + 16 * is + c];
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ch = 16 * io + 4 * q + r;
                bias2[io][r] = a.b1[ch];
                const float g = a.gamma1[ch], dg = a.dgamma1[ch], db = a.dbeta1[ch], m = a.mean1[ch], s = a.invstd1[ch];
                cA1[io][r] = g * s;
                cC1[io][r] = g * s * s * dg * invE;
                cD1[io][r] = g * s * (m * s * dg - db) * invE;
            }
        }
    }
    // input-gradient operand of block 0 (pass D with dfeat): W0^T rows k < CF, one 16-wide tile per 16 feature channels
    constexpr int TF = CF / 16 > 0 ? CF / 16 : 1;
    float A0T[TF][TO1][4];
    if (PASS == 3 && a.dfeat) {
#pragma unroll
        for (int kt = 0; kt < TF; ++kt)
#pragma unroll
            for (int io = 0; io < TO1; ++io)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const int k = 16 * kt + c;
                    A0T[kt][io][kb] = k < CF ? a.W0[(16 * io + 4 * q + kb) * CIN + k] : 0.f;
                }
    }

    f32x4 acc[TP][TQ];
#pragma unroll
    for (int i = 0; i < TP; ++i)
#pragma unroll
        for (int j = 0; j < TQ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbias[LAST2 ? TO2 : 1][4], dbe0[TO1][4], dga0[TO1][4];   // block 1 bias gradient; block 0 BN gradients (pass C)
#pragma unroll
    for (int i = 0; i < (LAST2 ? TO2 : 1); ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) dbias[i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < TO1; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) dbe0[i][r] = dga0[i][r] = 0.f;

    constexpr int CL = NL == 2 ? C2 : C1, TOL = CL / 16;
    SASTAMP(1);
    unsigned long long n_items_dbg = 0, n_steps_dbg = 0;
    // positions of the order table: nA per plot for SOLO / QUAD items, then nB per plot for the packed ones (OCT / HEX)
    const int nA = first.nA, nB = first.nB;
    const int G = a.group > 0 && a.group < a.B ? a.group : a.B;
    const int nitems = a.order ? (nA + nB) * G * ((a.B + G - 1) / G) : (ncent + 3) >> 2;
    // items are ordered heaviest first: the waves take them in a snake (round 0: item w, round 1: item 2W-1-w, ...), so the
    // wave with the longest item of a round gets the shortest of the next.  (A shared work counter was tried: 8000 atomics
    // on one address cost more than the imbalance they removed.)
    for (int round = 0;; ++round) {
        const int qi = round * nwaves + ((round & 1) ? nwaves - 1 - wave : wave);
        if (round * nwaves >= nitems) break;
        if (qi >= nitems) continue;
        SaItem it;
        if (!sa_item(it, a.order, a.cnt, a.B, a.M, nA, nB, G, qi, c, round == 0 ? first.rec : nullptr)) continue;      // an empty position
        if (n_items_dbg == 0) { SASTAMP(2); }
        ++n_items_dbg;
        n_steps_dbg += (it.nmax + it.estep - 1) / it.estep;
        const int nmax = it.nmax, estep = it.estep;
        float cpq[4];
        // d loss / d output and the winning slot of this lane's 4 channels, per tile (of this lane's centroid)
        float dov[4][TOL][4];
        int arv[4][TOL][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            cpq[t] = a.cpos[(size_t)it.lc[t] * 4 + (q < 3 ? q : 0)];
#pragma unroll
            for (int io = 0; io < TOL; ++io) {
                const size_t o = (size_t)it.lc[t] * CL + 16 * io + 4 * q;
                const float4 dv = *reinterpret_cast<const float4*>(a.dout + o);
                const int4 av = *reinterpret_cast<const int4*>(a.arg + o);
                dov[t][io][0] = dv.x; dov[t][io][1] = dv.y; dov[t][io][2] = dv.z; dov[t][io][3] = dv.w;
                arv[t][io][0] = av.x; arv[t][io][1] = av.y; arv[t][io][2] = av.z; arv[t][io][3] = av.w;
            }
        }
        int jn[4];                                 // neighbour indices, fetched one step ahead (the first step's: with the counts)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = it.le0 + it.eoff * t;
            jn[t] = a.nbr[(size_t)it.lc[t] * a.cap + (e < a.cap ? e : 0)];     // (unconditional: masked below)
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) jn[t] = it.le0 + it.eoff * t < it.ln[t] ? jn[t] : 0;
        for (int e0 = 0; e0 < nmax; e0 += estep) {
            f32x4 D1[TO1][4];
            float bks[4][KB1];
            bool val[4];
            size_t rows[4];
            int jc[4];
            // (every load of a step issued before any is waited for, the index prefetch unconditional from a clamped slot: see the
            // forward kernel)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                jc[t] = jn[t];
                const int en = e0 + estep + it.le0 + it.eoff * t;
                const bool more = en < it.ln[t];
                const int jr = a.nbr[(size_t)it.lc[t] * a.cap + (more ? en : 0)];
                jn[t] = more ? jr : 0;
                rows[t] = (size_t)it.bt[t] * a.Nsrc + jc[t];
#pragma unroll
                for (int kb = 0; kb < KB1 - 1; ++kb) bks[t][kb] = a.feat[rows[t] * a.feat_stride + 4 * kb + q];
                bks[t][KB1 - 1] = a.spos[rows[t] * a.spos_stride + (q < 3 ? q : 0)];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int e = e0 + it.le0 + it.eoff * t;                            // slot in the list of this lane's centroid
                val[t] = e < it.ln[t];
                bks[t][KB1 - 1] = q < 3 ? bks[t][KB1 - 1] - cpq[t] : 1.0f;
#pragma unroll
                for (int io = 0; io < TO1; ++io) {
                    f32x4 v = {bias1[io][0], bias1[io][1], bias1[io][2], bias1[io][3]};
                    v = contract<BF16, KB1>(v, [&](int kb) { return A1[io][kb]; }, [&](int kb) { return bks[t][kb]; });
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                    D1[io][t] = v;
                }
            }
            f32x4 dy1[TO1][4];   // d loss / d (BN output of block 0)
            if constexpr (NL == 2) {
                f32x4 Y1[TO1][4], dp2[TO2][4];
#pragma unroll
                for (int is = 0; is < TO1; ++is)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) Y1[is][t][r] = fmaf(a1v[is][r], D1[is][t][r], c1v[is][r]);
#pragma unroll
                for (int io = 0; io < TO2; ++io)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        f32x4 v = {bias2[io][0], bias2[io][1], bias2[io][2], bias2[io][3]};
                        v = contract<BF16, 4 * TO1>(v, [&](int kk) { return A2[io][kk >> 2][kk & 3]; },
                                                    [&](int kk) { return Y1[kk >> 2][t][kk & 3]; });
                        const int e = e0 + it.le0 + it.eoff * t;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float h = fmaxf(v[r], 0.f);
                            const float dy = (val[t] && arv[t][io][r] == e) ? dov[t][io][r] : 0.f;
                            dp2[io][t][r] = (val[t] && h > 0.f) ? fmaf(cA1[io][r], dy, fmaf(-cC1[io][r], h, cD1[io][r])) : 0.f;
                        }
                    }
                // d loss / d y1 = W2^T dp2
#pragma unroll
                for (int is = 0; is < TO1; ++is)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        f32x4 v = {0.f, 0.f, 0.f, 0.f};
                        v = contract<BF16, 4 * TO2>(v, [&](int kk) { return A2T[is][kk >> 2][kk & 3]; },
                                                    [&](int kk) { return dp2[kk >> 2][t][kk & 3]; });
                        dy1[is][t] = v;
                    }
                if constexpr (PASS == 2) {
                    // images [message][channel]: dp2 and y1
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
#pragma unroll
                        for (int io = 0; io < TO2; ++io)
                            *reinterpret_cast<float4*>(lds_p + (16 * t + c) * PS + 16 * io + 4 * q) =
                                make_float4(dp2[io][t][0], dp2[io][t][1], dp2[io][t][2], dp2[io][t][3]);
#pragma unroll
                        for (int is = 0; is < TO1; ++is)
                            *reinterpret_cast<float4*>(lds_q + (16 * t + c) * QS + 16 * is + 4 * q) =
                                make_float4(Y1[is][t][0], Y1[is][t][1], Y1[is][t][2], Y1[is][t][3]);
                    }
#pragma unroll
                    for (int io = 0; io < TO2; ++io)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) dbias[io][r] += dp2[io][t][r];
#pragma unroll
                    for (int is = 0; is < TO1; ++is)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                dbe0[is][r] += dy1[is][t][r];
                                dga0[is][r] = fmaf(dy1[is][t][r], (D1[is][t][r] - mu0[is][r]) * is0[is][r], dga0[is][r]);
                            }
                }
            } else {
                // nl == 1: the only block is the last one, dy comes from the extremum slot
#pragma unroll
                for (int is = 0; is < TO1; ++is)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int e = e0 + it.le0 + it.eoff * t;
#pragma unroll
                        for (int r = 0; r < 4; ++r) dy1[is][t][r] = (val[t] && arv[t][is][r] == e) ? dov[t][is][r] : 0.f;
                    }
            }
            if constexpr (PASS == 3) {
                f32x4 dp1[TO1][4];
#pragma unroll
                for (int is = 0; is < TO1; ++is)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float h = D1[is][t][r];
                            dp1[is][t][r] = (val[t] && h > 0.f)
                                                ? fmaf(cA0[is][r], dy1[is][t][r], fmaf(-cC0[is][r], h, cD0[is][r])) : 0.f;
                        }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int is = 0; is < TO1; ++is)
                        *reinterpret_cast<float4*>(lds_p + (16 * t + c) * PS + 16 * is + 4 * q) =
                            make_float4(dp1[is][t][0], dp1[is][t][1], dp1[is][t][2], dp1[is][t][3]);
#pragma unroll
                    for (int kb = 0; kb < KB1; ++kb) lds_q[(16 * t + c) * QS + 4 * kb + q] = bks[t][kb];
                }
                if (a.dfeat) {
                    // d loss / d source features = W0^T dp1, rows k < CF; added onto the source rows of the messages
#pragma unroll
                    for (int kt = 0; kt < TF; ++kt)
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            f32x4 v = {0.f, 0.f, 0.f, 0.f};
                            v = contract<BF16, 4 * TO1>(v, [&](int kk) { return A0T[kt][kk >> 2][kk & 3]; },
                                                        [&](int kk) { return dp1[kk >> 2][t][kk & 3]; });
                            if (val[t]) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int k = 16 * kt + 4 * q + r;
                                    if (k < CF) atomicAdd(&a.dfeat[rows[t] * CF + k], v[r]);
                                }
                            }
                        }
                }
            }
            // ---- weight gradient of this pass: contraction over the 64 messages of the step (16 k-steps of 4)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            {
                const float* rp = lds_p + q * PS + c;
                const float* rq = lds_q + q * QS + c;
                if constexpr (!BF16) {
#pragma unroll 4
                    for (int st = 0; st < 16; ++st) {
                        float av[TP], bv[TQ];
#pragma unroll
                        for (int i = 0; i < TP; ++i) av[i] = rp[st * 4 * PS + 16 * i];
#pragma unroll
                        for (int j = 0; j < TQ; ++j) bv[j] = rq[st * 4 * QS + 16 * j];
#pragma unroll
                        for (int i = 0; i < TP; ++i)
#pragma unroll
                            for (int j = 0; j < TQ; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
                    }
                } else {
                    // the 64 messages of the step are the K of two v_mfma_f32_16x16x32_bf16 per tile pair
#pragma unroll
                    for (int s8 = 0; s8 < 2; ++s8) {
                        float av[TP][8], bv[TQ][8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
#pragma unroll
                            for (int i = 0; i < TP; ++i) av[i][u] = rp[(8 * s8 + u) * 4 * PS + 16 * i];
#pragma unroll
                            for (int j = 0; j < TQ; ++j) bv[j][u] = rq[(8 * s8 + u) * 4 * QS + 16 * j];
                        }
#pragma unroll
                        for (int i = 0; i < TP; ++i)
#pragma unroll
                            for (int j = 0; j < TQ; ++j)
                                acc[i][j] = contract<true, 8>(acc[i][j], [&](int u) { return av[i][u]; }, [&](int u) { return bv[j][u]; });
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (n_items_dbg == 1) { SASTAMP(3); }
    }
    SASTAMP(4);
    SACOUNT(8, n_items_dbg);
    SACOUNT(9, n_steps_dbg);

    // ---- workgroup-level reduction (every wave's image in its own staging region, plain stores, added in wave order: LDS
    // float atomics run at ~0.4 lane-adds per clock), then one global atomic per element
    constexpr int NW = PO * QK, NB = LAST2 ? C2 : 0, NG = LAST2 ? 2 * C1 : 0;
    static_assert(NW + NB + NG <= Acc::LDS_FLOATS, "a wave's image fits its staging region");
    float* slab = lds_p;
#pragma unroll
    for (int i = 0; i < TP; ++i)
#pragma unroll
        for (int j = 0; j < TQ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = 16 * i + 4 * q + r, k = 16 * j + c;
                if (o < PO && k < QK) slab[o * QK + k] = acc[i][j][r];
            }
    if constexpr (LAST2) {
#pragma unroll
        for (int io = 0; io < TO2; ++io)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = row_sum(dbias[io][r]);
                if (c == 0) slab[NW + 16 * io + 4 * q + r] = v;
            }
#pragma unroll
        for (int is = 0; is < TO1; ++is)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v1 = row_sum(dbe0[is][r]), v2 = row_sum(dga0[is][r]);
                if (c == 0) {
                    slab[NW + NB + 16 * is + 4 * q + r] = v1;
                    slab[NW + NB + C1 + 16 * is + 4 * q + r] = v2;
                }
            }
    }
    SASTAMP(5);
    __syncthreads();
    SASTAMP(6);
    const int img = sn2_grad_image(a.rep_k, a.rep_stride);
    for (int i = threadIdx.x; i < NW + NB + NG; i += 256) {
        const float v = (smem[i] + smem[Acc::LDS_FLOATS + i]) + (smem[2 * Acc::LDS_FLOATS + i] + smem[3 * Acc::LDS_FLOATS + i]);
        if (v == 0.f) continue;
        if (i < NW) {
            if constexpr (LAST2) {
                SN2_FLUSH_ADD(&a.dW1[img + i], v);
            } else {
                const int o = i / QK, k = i - o * QK;   // block 0: columns = inputs | bias
                if (k < CIN) SN2_FLUSH_ADD(&a.dW0[img + o * CIN + k], v);
                else SN2_FLUSH_ADD(&a.db0[img + o], v);
            }
        } else if (i < NW + NB) {
            SN2_FLUSH_ADD(&a.db1[img + i - NW], v);
        } else if (i < NW + NB + C1) {
            SN2_FLUSH_ADD(&a.dbeta0_out[i - NW - NB], v);
        } else {
            SN2_FLUSH_ADD(&a.dgamma0_out[i - NW - NB - C1], v);
        }
    }
    SASTAMP(7);
}

}  // namespace

// plots whose items a round of waves works on together (sa_item): all of a training batch, 8 of a parcel launch's hundreds
static int sa_plot_group(int B) {
    static const int g_env = getenv("SN2_SA_GROUP") ? atoi(getenv("SN2_SA_GROUP")) : 0;
    if (g_env > 0) return g_env;
    return B <= 32 ? B : 8;
}

// launched from sa.hip
template <int CF, int NL, int C1, int C2, int PASS>
int sa_mfma_launch_fwd(const sn2_sa* p, int training, hipStream_t st, int* nblocks_out) {
    const bool bf16 = p->blk[0].mma_bf16 != 0;
    SaFwdArgs a;
    a.B = p->B; a.Nsrc = p->Nsrc; a.M = p->M; a.cap = p->cap; a.feat_stride = p->feat_stride; a.spos_stride = p->spos_stride;
    a.feat = p->feat; a.spos = p->spos; a.cpos = p->cpos; a.nbr = p->nbr; a.cnt = p->cnt; a.order = p->order;
    a.group = sa_plot_group(p->B);
    const sn2_block& k0 = p->blk[0];
    const sn2_block& k1 = p->blk[NL == 2 ? 1 : 0];
    a.W0 = k0.W; a.b0 = k0.b; a.a0 = k0.a; a.c0 = k0.c; a.gamma0 = k0.gamma;
    a.W1 = k1.W; a.b1 = k1.b; a.gamma1 = k1.gamma;
    a.slots = training == 1 ? (PASS == 0 ? k0.stat_slots : k1.stat_slots) : nullptr;     // (SN2_BN_FROZEN_KEEP: no sums wanted)
    a.ext = p->ext; a.arg = p->arg;
    a.al = k1.a; a.cl = k1.c; a.out = p->out;
    int blocks = sn2_cdiv((long)p->B * p->M, 16);          // one wave per quad of centroids, four waves per workgroup
    if (blocks > SN2_STAT_SLOTS) blocks = SN2_STAT_SLOTS;
    // .. and no more workgroups than the chip holds at once (waves per SIMD by the kernels' register counts): the items are
    // dealt heaviest first, a second wave of workgroups would start on the light tail when the first is done with the heads
    static const int occ_env = getenv("SN2_SA_FWD_OCC") ? atoi(getenv("SN2_SA_FWD_OCC")) : 0;
    const int occ = occ_env > 0 ? occ_env : (CF == 8 ? (PASS == 0 ? 4 : 3) : (CF == 16 ? 2 : 1));
    if (blocks > sn2_cu_count() * occ) blocks = sn2_cu_count() * occ;
    if (nblocks_out) *nblocks_out = blocks;
    if constexpr (PASS == 1) {
        if (!training) {           // eval: the variant without the statistic sums and the arg-max slots
            if (bf16) hipLaunchKernelGGL((sa_mfma_fwd_kernel<CF, NL, C1, C2, PASS, true, false>), dim3(blocks), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((sa_mfma_fwd_kernel<CF, NL, C1, C2, PASS, false, false>), dim3(blocks), dim3(256), 0, st, a);
            SN2_RETURN_LAUNCH();
        }
    }
    if (bf16) hipLaunchKernelGGL((sa_mfma_fwd_kernel<CF, NL, C1, C2, PASS, true>), dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((sa_mfma_fwd_kernel<CF, NL, C1, C2, PASS, false>), dim3(blocks), dim3(256), 0, st, a);
    SN2_RETURN_LAUNCH();
}

template int sa_mfma_launch_fwd<8, 2, 16, 16, 0>(const sn2_sa*, int, hipStream_t, int*);
template int sa_mfma_launch_fwd<8, 2, 16, 16, 1>(const sn2_sa*, int, hipStream_t, int*);
template int sa_mfma_launch_fwd<16, 1, 32, 32, 1>(const sn2_sa*, int, hipStream_t, int*);
template int sa_mfma_launch_fwd<32, 1, 64, 64, 1>(const sn2_sa*, int, hipStream_t, int*);   // third ball-query level (3sa-arch)

template <int CF, int NL, int C1, int C2, int PASS>
int sa_mfma_launch_bwd(const sn2_sa* p, hipStream_t st) {
    constexpr int CIN = CF + 3;
    constexpr bool LAST2 = NL == 2 && PASS == 2;
    using Acc = OuterAcc<(LAST2 ? C2 : C1), (LAST2 ? C1 : CIN + 1)>;
    SaBwdArgs a;
    a.B = p->B; a.Nsrc = p->Nsrc; a.M = p->M; a.cap = p->cap; a.feat_stride = p->feat_stride; a.spos_stride = p->spos_stride;
    a.feat = p->feat; a.spos = p->spos; a.cpos = p->cpos; a.nbr = p->nbr; a.cnt = p->cnt; a.order = p->order; a.total = p->total;
    a.group = sa_plot_group(p->B);
    const sn2_block& k0 = p->blk[0];
    const sn2_block& k1 = p->blk[NL == 2 ? 1 : 0];
    a.frozen = (k0.frozen_stats || k1.frozen_stats) ? 1 : 0;      // (one forward pass = one mode for the module's blocks)
    a.W0 = k0.W; a.b0 = k0.b; a.a0 = k0.a; a.c0 = k0.c; a.gamma0 = k0.gamma; a.mean0 = k0.mean; a.invstd0 = k0.invstd;
    a.dgamma0 = k0.dgamma; a.dbeta0 = k0.dbeta;
    a.W1 = k1.W; a.b1 = k1.b; a.gamma1 = k1.gamma; a.mean1 = k1.mean; a.invstd1 = k1.invstd; a.dgamma1 = k1.dgamma;
    a.dbeta1 = k1.dbeta;
    a.dout = p->dout; a.arg = p->arg;
    a.dW0 = k0.dW; a.db0 = k0.db; a.dW1 = k1.dW; a.db1 = k1.db; a.dgamma0_out = k0.dgamma; a.dbeta0_out = k0.dbeta;
    a.rep_k = k0.grad_replicas; a.rep_stride = k0.grad_replica_stride;
    a.dfeat = p->dfeat;
    int blocks = sn2_cdiv((long)p->B * p->M, 16);
    static const int occ_env = getenv("SN2_SA_BWD_OCC") ? atoi(getenv("SN2_SA_BWD_OCC")) : 0;
    const int occ = occ_env > 0 ? occ_env : (CF == 8 ? 2 : 1);            // workgroups per CU that are resident together
    if (blocks > sn2_cu_count() * occ) blocks = sn2_cu_count() * occ;
    const size_t lds = (size_t)Acc::LDS_FLOATS * 4 * sizeof(float);
    auto kern = p->blk[0].mma_bf16 ? &sa_mfma_bwd_kernel<CF, NL, C1, C2, PASS, true> : &sa_mfma_bwd_kernel<CF, NL, C1, C2, PASS, false>;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, st, a);
    SN2_RETURN_LAUNCH();
}

template int sa_mfma_launch_bwd<8, 2, 16, 16, 2>(const sn2_sa*, hipStream_t);
template int sa_mfma_launch_bwd<8, 2, 16, 16, 3>(const sn2_sa*, hipStream_t);
template int sa_mfma_launch_bwd<16, 1, 32, 32, 3>(const sn2_sa*, hipStream_t);
template int sa_mfma_launch_bwd<32, 1, 64, 64, 3>(const sn2_sa*, hipStream_t);
