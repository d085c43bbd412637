// mlp.h -- per-lane dense layers (weights wave-uniform -> scalar loads / SGPR operands) and the MFMA outer-product
// accumulator used for every weight gradient (dW = sum over rows of dpre (x) input: the row index is the MFMA K).
#pragma once
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- one contraction on the matrix cores, in either operand precision ---------------------------------------------
// Every dense contraction of the library has the same shape: a 16 x 16 output tile per wave, K walked in slots of four
// (lane (q, c) supplies element k = 4 kb + q of its row / column for slot kb).  `contract<BF16, KBN>(acc, a, b)` adds
// sum_{kb < KBN} A(kb) * B(kb) to the tile, a(kb) / b(kb) returning this lane's two operand values of slot kb:
//   BF16 = false:  one v_mfma_f32_16x16x4_f32 per slot, in slot order (exact fp32 products, fp32 accumulate);
//   BF16 = true:   the operands are rounded to bfloat16 (v_cvt_pk_bf16_f32, round to nearest even) and EIGHT slots go
//                  into one v_mfma_f32_16x16x32_bf16 (a contraction of no more than four slots uses one
//                  v_mfma_f32_16x16x16_bf16), fp32 accumulate -- BASELINE.json configs[4], "bf16 MLP weights on MFMA".  Which k a register
//                  slot stands for does not matter as long as A and B agree, so the fp32 operand registers are reused as
//                  they are: lane (q, c) holds 8 (4) consecutive entries of the instruction's K = 32 (16).
// Loop-invariant operands (the weights) are converted once: the conversions are hoisted out of the loops by the compiler.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool BF16, int KBN, class FA, class FB>
__device__ __forceinline__ f32x4 contract(f32x4 acc, FA a, FB b) {
    if constexpr (!BF16) {
#pragma unroll
        for (int kb = 0; kb < KBN; ++kb) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a(kb), b(kb), acc, 0, 0, 0);
    } else {
        // One instruction shape per accumulation chain: a v_mfma_f32_16x16x16_bf16 that takes the result of a
        // v_mfma_f32_16x16x32_bf16 as its accumulator three instructions later read two of its four registers too early
        // on this toolchain (ROCm 7.2; rows 4q+0, 4q+1 of the tile wrong, scripts/debug_bf16_sa3.py) -- so more than four
        // slots use the K = 32 form throughout (a short tail is padded with zeros), four or fewer the K = 16 form.
#pragma unroll
        for (int k0 = 0; k0 < KBN; k0 += 8) {
            if (KBN > 4) {
                bf16x8 av, bv;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    av[t] = (__bf16)(k0 + t < KBN ? a(k0 + t < KBN ? k0 + t : 0) : 0.f);
                    bv[t] = (__bf16)(k0 + t < KBN ? b(k0 + t < KBN ? k0 + t : 0) : 0.f);
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc, 0, 0, 0);
            } else {
                bf16x4 av, bv;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    av[t] = (__bf16)(k0 + t < KBN ? a(k0 + t < KBN ? k0 + t : 0) : 0.f);
                    bv[t] = (__bf16)(k0 + t < KBN ? b(k0 + t < KBN ? k0 + t : 0) : 0.f);
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, av), __builtin_bit_cast(s16x4, bv), acc,
                                                                0, 0, 0);
            }
        }
    }
    return acc;
}

// out[o] = (relu) ( b[o] + sum_k W[o*CI+k] * in[k] ).  W, b are wave-uniform constant-address-space pointers (common.h):
// the loads are s_load_dwordx* and the FMAs take SGPR operands.
// ---- BatchNorm1d from the sums of a batch (torch defaults: eps 1e-5, momentum 0.1, biased variance to normalise, unbiased
// variance into running_var -- model/point_net2.py:45-53): one channel.  Used by bn_finalize_kernel (misc.hip) and by the
// kernels that finalise their own statistics (fp.hip: global_level_fwd_kernel).
__device__ __forceinline__ void sn2_bn_from_sums(double s1, double s2, double n, float gamma, float beta, float* running_mean,
                                                 float* running_var, float& a, float& c, float& mean, float& invstd) {
    const float eps = 1e-5f, mom = 0.1f;
    if (n < 1.0) n = 1.0;
    const double m = s1 / n;
    double var = s2 / n - m * m;
    if (var < 0.0) var = 0.0;
    mean = (float)m;
    invstd = 1.0f / sqrtf((float)var + eps);
    const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
    if (running_mean) {
        *running_mean = (1.f - mom) * *running_mean + mom * mean;
        *running_var = (1.f - mom) * *running_var + mom * (float)unbiased;
    }
    a = gamma * invstd;
    c = beta - mean * a;
}

template <int CI, int CO, bool RELU>
__device__ __forceinline__ void dense(cfp W, cfp b, const float (&in)[CI], float (&out)[CO]) {
#pragma unroll
    for (int o = 0; o < CO; ++o) {
        float acc = b[o];
#pragma unroll
        for (int k = 0; k < CI; ++k) acc = fmaf(W[o * CI + k], in[k], acc);
        out[o] = RELU ? fmaxf(acc, 0.f) : acc;
    }
}

// out[k] = sum_o W[o*CI+k] * g[o]   (transpose product: input gradient)
template <int CI, int CO, int CI_USED>
__device__ __forceinline__ void dense_t(cfp W, const float (&g)[CO], float (&out)[CI_USED]) {
#pragma unroll
    for (int k = 0; k < CI_USED; ++k) {
        float acc = 0.f;
#pragma unroll
        for (int o = 0; o < CO; ++o) acc = fmaf(W[o * CI + k], g[o], acc);
        out[k] = acc;
    }
}

// D[o][k] += sum over the wave's 64 rows r of P[r][o] * Q[r][k], lane r holding row r of P (CO values) and Q (CK values).
// The rows are staged in a wave-private LDS region ([ROWS][PS] and [ROWS][QS] floats) and contracted with
// v_mfma_f32_16x16x4_f32 (exact fp32): A[o][r] = P[r][o] is read as lds_p[(4s + lane>>4)*PS + 16*to + (lane&15)],
// B[r][k] likewise; ROWS/4 k-steps cover ROWS rows, 64/ROWS phases cover the wave (phase h stages the rows of lanes
// [h*ROWS, (h+1)*ROWS)).  ROWS = 32 or 16 trades a few more (partially masked) ds_write for a 2-4x smaller region: the
// staging area, not registers, is what limits the waves per CU of the row kernels (head_bwd ran 1 wave per SIMD with
// ROWS = 64).  Every add writes its full tile columns (values, then zeros up to the tile edge), so the region needs no
// initialisation and several accumulators can take turns in one region.  Row strides are == 16 (mod 32) floats so the
// two 32-lane halves of a ds_read_b32 hit disjoint banks.  Accumulators live in registers for the whole kernel and are
// flushed once (D layout: row = 4*(lane>>4) + reg, col = lane & 15).
template <int CO, int CK, int ROWS = 64>
struct OuterAcc {
    static_assert(ROWS == 64 || ROWS == 32 || ROWS == 16, "ROWS");
    static constexpr int TO = (CO + 15) / 16, TK = (CK + 15) / 16;
    static constexpr int PS = ((TO * 16) % 32 == 0) ? TO * 16 + 16 : TO * 16;
    static constexpr int QS = ((TK * 16) % 32 == 0) ? TK * 16 + 16 : TK * 16;
    static constexpr int LDS_FLOATS = ROWS * (PS + QS);
    static constexpr int STAGED_ROWS = ROWS;
    f32x4 acc[TO][TK];

    __device__ __forceinline__ void init(float*) {
#pragma unroll
        for (int a = 0; a < TO; ++a)
#pragma unroll
            for (int c = 0; c < TK; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    __device__ __forceinline__ void add(float* lds, const float (&p)[CO], const float (&q)[CK]) {
        add_then(lds, p, q, [](int) {});
    }

    // as add(); `after(h)` runs while phase h's rows (lanes [h*ROWS, (h+1)*ROWS) of the wave) are still staged: the P rows
    // are at lds[row * PS + o], row = lane % ROWS -- a second contraction over them (the input gradient) needs no restaging
    template <class F>
    __device__ __forceinline__ void add_then(float* lds, const float (&p)[CO], const float (&q)[CK], F after) {
        const int lane = threadIdx.x & 63;
        const int row = lane & (ROWS - 1);
        float* lp = lds + row * PS;
        float* lq = lds + ROWS * PS + row * QS;
        const int r4 = lane >> 4, c = lane & 15;
        const float* rp = lds + r4 * PS + c;
        const float* rq = lds + ROWS * PS + r4 * QS + c;
#pragma unroll
        for (int h = 0; h < 64 / ROWS; ++h) {
            if (ROWS == 64 || (lane / ROWS) == h) {
#pragma unroll
                for (int o = 0; o < TO * 16; o += 4) {
                    float v[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (o + t < CO) v[t] = p[o + t < CO ? o + t : 0];
                        else v[t] = 0.f;
                    }
                    *reinterpret_cast<float4*>(lp + o) = make_float4(v[0], v[1], v[2], v[3]);
                }
#pragma unroll
                for (int k = 0; k < TK * 16; k += 4) {
                    float v[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (k + t < CK) v[t] = q[k + t < CK ? k + t : 0];
                        else v[t] = 0.f;
                    }
                    *reinterpret_cast<float4*>(lq + k) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll 4
            for (int s = 0; s < ROWS / 4; ++s) {
                float av[TO], bv[TK];
#pragma unroll
                for (int a = 0; a < TO; ++a) av[a] = rp[s * 4 * PS + a * 16];
#pragma unroll
                for (int b = 0; b < TK; ++b) bv[b] = rq[s * 4 * QS + b * 16];
#pragma unroll
                for (int a = 0; a < TO; ++a)
#pragma unroll
                    for (int b = 0; b < TK; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc[a][b], 0, 0, 0);
            }
            after(h);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }

    // this wave's tile sums as an image slab[CO][CK] (plain stores, every element written once): the workgroup then adds
    // its waves' slabs in wave order.  The wave's own staging region is free for it once its last add() has returned.
    // (flush_lds below does the same with LDS float atomics: ~0.4 lane-adds per clock -- 5-10 us at the end of a kernel.)
    __device__ __forceinline__ void store_slab(float* slab) {
        const int lane = threadIdx.x & 63;
        const int r4 = lane >> 4, c = lane & 15;
#pragma unroll
        for (int a = 0; a < TO; ++a)
#pragma unroll
            for (int b = 0; b < TK; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = a * 16 + r4 * 4 + r, k = b * 16 + c;
                    if (o < CO && k < CK) slab[o * CK + k] = acc[a][b][r];
                }
    }

    // add this wave's tile sums into a block-shared LDS image red[CO][CK] (LDS float atomics; the image must be zeroed
    // and the workgroup synchronised before, and synchronised again before it is read)
    __device__ __forceinline__ void flush_lds(float* red) {
        const int lane = threadIdx.x & 63;
        const int r4 = lane >> 4, c = lane & 15;
#pragma unroll
        for (int a = 0; a < TO; ++a)
#pragma unroll
            for (int b = 0; b < TK; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = a * 16 + r4 * 4 + r, k = b * 16 + c;
                    if (o < CO && k < CK) atomicAdd(&red[o * CK + k], acc[a][b][r]);
                }
    }
};

// Every reduction that leaves a kernel goes through the workgroup first: thousands of waves adding to the same few
// addresses serialise in the memory-side atomic units (a first version that let every wave issue its own global
// atomics spent 1.5 ms per pass there).  Per-lane partials -> wave sum -> LDS float atomic on a block-shared image.
template <int C>
__device__ __forceinline__ void sums_to_lds(const float (&v)[C], float* red) {
#pragma unroll
    for (int o = 0; o < C; ++o) {
        const float s = wave_sum(v[o]);
        if ((threadIdx.x & 63) == 0) atomicAdd(&red[o], s);
    }
}

// BatchNorm batch statistics leave a kernel as one slot per workgroup: slot[blockIdx.x] = [sum(C) | sumsq(C)] (fp32
// partials, every workgroup writes its slot, so no zeroing and no atomics; bn_finalize adds the slots in fp64 in a
// fixed order).  Inside the workgroup the waves' sums are added in wave order, not with LDS atomics: the statistics are
// identical from run to run, bit for bit (an order that varies moves them by 1e-7, and now and then a pre-activation
// next to zero changes sign with it: one ReLU mask flip was seen to move a weight gradient by 1 %).
// `red` = block-shared scratch of (blockDim.x / 64) * 2 * C floats.
template <int C>
__device__ __forceinline__ void stats_to_slot(const float (&ssum)[C], const float (&ssq)[C], float* red,
                                              float* __restrict__ slots) {
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int o = 0; o < C; ++o) {
        const float s1 = wave_sum(ssum[o]), s2 = wave_sum(ssq[o]);
        if ((threadIdx.x & 63) == 0) {
            red[w * 2 * C + o] = s1;
            red[w * 2 * C + C + o] = s2;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
        float acc = 0.f;
        for (int k = 0; k < nw; ++k) acc += red[k * 2 * C + i];
        slots[(size_t)blockIdx.x * 2 * C + i] = acc;
    }
}

// launcher of the BatchNorm finalisation kernel (misc.hip): from (sum, sumsq, count) or the running statistics to the
// affine (a, c), the saved (mean, invstd) and the running-statistics update.
int sn2_bn_finalize(const sn2_block* blk, int nslots, const unsigned long long* count_dev, long count_imm,
                    int training, hipStream_t st);
// the same + a set-abstraction level's output out = a ext + c (0 where arg < 0) over (rows, cout) in the same launch (misc.hip)
int sn2_bn_finalize_apply(const sn2_block* blk, int nslots, const unsigned long long* count_dev, long count_imm, int training,
                          const float* ext, const int* arg, float* out, long rows, hipStream_t st);
