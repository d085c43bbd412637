// mlp.h -- per-lane dense layers (weights wave-uniform -> scalar loads / SGPR operands) and the MFMA outer-product
// accumulator used for every weight gradient (dW = sum over rows of dpre (x) input: the row index is the MFMA K).
#pragma once
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// out[o] = (relu) ( b[o] + sum_k W[o*CI+k] * in[k] ).  W, b are wave-uniform constant-address-space pointers (common.h):
// the loads are s_load_dwordx* and the FMAs take SGPR operands.
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
// The rows are staged in a wave-private LDS region ([64][PS] and [64][QS] floats, pad columns stay zero) and contracted
// with v_mfma_f32_16x16x4_f32 (exact fp32): A[o][r] = P[r][o] is read as lds_p[(4s + lane>>4)*PS + 16*to + (lane&15)],
// B[r][k] likewise; 16 k-steps cover the 64 rows.  Row strides are == 16 (mod 32) floats so the two 32-lane halves of
// a ds_read_b32 hit disjoint banks.  Accumulators live in registers for the whole kernel and are flushed once with
// float atomics (D layout: row = 4*(lane>>4) + reg, col = lane & 15).
template <int CO, int CK>
struct OuterAcc {
    static constexpr int TO = (CO + 15) / 16, TK = (CK + 15) / 16;
    static constexpr int PS = ((TO * 16) % 32 == 0) ? TO * 16 + 16 : TO * 16;
    static constexpr int QS = ((TK * 16) % 32 == 0) ? TK * 16 + 16 : TK * 16;
    static constexpr int LDS_FLOATS = 64 * (PS + QS);
    f32x4 acc[TO][TK];

    __device__ __forceinline__ void init(float* lds) {
#pragma unroll
        for (int a = 0; a < TO; ++a)
#pragma unroll
            for (int c = 0; c < TK; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int lane = threadIdx.x & 63;
        for (int i = lane; i < LDS_FLOATS; i += 64) lds[i] = 0.f;
        __builtin_amdgcn_wave_barrier();
    }

    __device__ __forceinline__ void add(float* lds, const float (&p)[CO], const float (&q)[CK]) {
        const int lane = threadIdx.x & 63;
        float* lp = lds + lane * PS;
        float* lq = lds + 64 * PS + lane * QS;
#pragma unroll
        for (int o = 0; o + 3 < CO; o += 4) *reinterpret_cast<float4*>(lp + o) = make_float4(p[o], p[o + 1], p[o + 2], p[o + 3]);
#pragma unroll
        for (int o = CO & ~3; o < CO; ++o) lp[o] = p[o];
#pragma unroll
        for (int k = 0; k + 3 < CK; k += 4) *reinterpret_cast<float4*>(lq + k) = make_float4(q[k], q[k + 1], q[k + 2], q[k + 3]);
#pragma unroll
        for (int k = CK & ~3; k < CK; ++k) lq[k] = q[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int r4 = lane >> 4, c = lane & 15;
        const float* rp = lds + r4 * PS + c;
        const float* rq = lds + 64 * PS + r4 * QS + c;
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
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
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
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
// fixed order => run-to-run identical statistics).  `red` = block-shared scratch of 2*C floats.
template <int C>
__device__ __forceinline__ void stats_to_slot(const float (&ssum)[C], const float (&ssq)[C], float* red,
                                              float* __restrict__ slots) {
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) red[i] = 0.f;
    __syncthreads();
    sums_to_lds<C>(ssum, red);
    sums_to_lds<C>(ssq, red + C);
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) slots[(size_t)blockIdx.x * 2 * C + i] = red[i];
}

// launcher of the BatchNorm finalisation kernel (misc.hip): from (sum, sumsq, count) or the running statistics to the
// affine (a, c), the saved (mean, invstd) and the running-statistics update.
int sn2_bn_finalize(const sn2_block* blk, int nslots, const unsigned long long* count_dev, long count_imm,
                    int training, hipStream_t st);
