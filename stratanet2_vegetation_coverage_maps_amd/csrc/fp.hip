// fp.hip -- dense-row (Linear->ReLU->BN) blocks with interpolated + skip inputs, forward and backward, and the pointwise
// head.  Replaces FPModule.forward (knn_interpolate's differentiable half + cat + MLP), GlobalSAModule's MLP and the
// head of PointNet2.forward: /root/reference/model/point_net2.py:37-42, 62-67, 141-151.
//
// One row per lane.  A row's input u = [ interp (ca) | skip (cb) ] is rebuilt in registers from the 1 or 3 source rows
// (16-byte gathers out of an L2-resident table), the Linear layer runs against wave-uniform weights (SGPR operands),
// the pre-BN activation h is stored once (row stride padded to 16 B) and its batch statistics are reduced per wave and
// added as fp64 atomics.  Consumers apply the BN affine (a, c) when they read h, so no BN-apply pass exists.
// Backward: (1) dgamma/dbeta reduction over rows, (2) main pass: dpre, dW|db through the MFMA outer-product accumulator
// (rows = MFMA K), input gradient du; (3) the interpolation's transpose as a gather through an inverted index (no
// floating-point atomics: see "backward (3)").
#include "mlp.h"

namespace {

template <int CA, int CB, bool KNN>
__device__ __forceinline__ void build_input(const float* __restrict__ src, int src_stride, cfp src_a, cfp src_c,
                                            const int* __restrict__ knn_idx,
                                            const float* __restrict__ knn_w, const float* __restrict__ skip,
                                            int skip_stride, size_t r, size_t src_plot_base, float (&u)[CA + CB + 1]) {
    constexpr int Q = (CA + 3) / 4;
    if constexpr (KNN) {
        const int i0 = knn_idx[r * 3 + 0], i1 = knn_idx[r * 3 + 1], i2 = knn_idx[r * 3 + 2];
        const float w0 = knn_w[r * 3 + 0], w1 = knn_w[r * 3 + 1], w2 = knn_w[r * 3 + 2];
        const float inv = 1.0f / ((w0 + w1) + w2);
        const float4* s0 = reinterpret_cast<const float4*>(src + (src_plot_base + i0) * src_stride);
        const float4* s1 = reinterpret_cast<const float4*>(src + (src_plot_base + i1) * src_stride);
        const float4* s2 = reinterpret_cast<const float4*>(src + (src_plot_base + i2) * src_stride);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const float4 a = s0[q], b = s1[q], c = s2[q];
            const float v[4] = {(a.x * w0 + b.x * w1) + c.x * w2, (a.y * w0 + b.y * w1) + c.y * w2,
                                (a.z * w0 + b.z * w1) + c.z * w2, (a.w * w0 + b.w * w1) + c.w * w2};
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (4 * q + t < CA) u[4 * q + t] = v[t] * inv;
        }
    } else {
        const float4* s0 = reinterpret_cast<const float4*>(src + r * src_stride);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const float4 a = s0[q];
            const float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (4 * q + t < CA) u[4 * q + t] = v[t];
        }
    }
    if (src_a) {
#pragma unroll
        for (int k = 0; k < CA; ++k) u[k] = fmaf(src_a[k], u[k], src_c[k]);
    }
    if constexpr (CB > 0) {
        const float* sk = skip + r * skip_stride;
        if constexpr (CB % 4 == 0) {
#pragma unroll
            for (int q = 0; q < CB / 4; ++q) {
                const float4 a = reinterpret_cast<const float4*>(sk)[q];
                u[CA + 4 * q + 0] = a.x;
                u[CA + 4 * q + 1] = a.y;
                u[CA + 4 * q + 2] = a.z;
                u[CA + 4 * q + 3] = a.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < CB; ++k) u[CA + k] = sk[k];
        }
    }
    u[CA + CB] = 1.0f;  // the bias column of the outer-product accumulator
}

// ---------------------------------------------------------------------------------------------- forward
template <int CA, int CB, int CO, bool KNN>
__global__ __launch_bounds__(256) void fp_fwd_kernel(int R, int R_per_plot, int S_per_plot, int src_stride, int skip_stride,
                                                     int h_stride, const float* __restrict__ src,
                                                     const float* __restrict__ src_a, const float* __restrict__ src_c,
                                                     const int* __restrict__ knn_idx, const float* __restrict__ knn_w,
                                                     const float* __restrict__ skip, const float* __restrict__ Wg,
                                                     const float* __restrict__ biasg, float* __restrict__ h,
                                                     float* __restrict__ slots) {
    constexpr int CI = CA + CB;
    __shared__ float s_red[4 * 2 * CO];
    float ssum[CO], ssq[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) ssum[o] = ssq[o] = 0.f;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < R; r += (long)gridDim.x * 256) {
        const cfp W = opaque(as_const(Wg)), bias = opaque(as_const(biasg));   // stream the weights inside the loop
        float u[CI + 1];
        const size_t plot = (size_t)(r / R_per_plot);
        build_input<CA, CB, KNN>(src, src_stride, opaque(as_const(src_a)), opaque(as_const(src_c)), knn_idx, knn_w, skip,
                                 skip_stride, (size_t)r, plot * S_per_plot, u);
        float* hr = h + (size_t)r * h_stride;
#pragma unroll
        for (int o4 = 0; o4 < CO; o4 += 4) {
            float v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int o = o4 + t;
                float acc = 0.f;
                if (o < CO) {
                    acc = bias[o];
#pragma unroll
                    for (int k = 0; k < CI; ++k) acc = fmaf(W[o * CI + k], u[k], acc);
                    acc = fmaxf(acc, 0.f);
                    ssum[o] += acc;
                    ssq[o] = fmaf(acc, acc, ssq[o]);
                }
                v[t] = acc;
            }
            *reinterpret_cast<float4*>(hr + o4) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    if (slots) stats_to_slot<CO>(ssum, ssq, s_red, slots);
}

// ---------------------------------------------------------------------------------------------- backward (1)
// dbeta[o] = sum_r dy[r][o];  dgamma[o] = sum_r dy[r][o] * (h[r][o] - mean[o]) * invstd[o]
template <int CO>
__global__ __launch_bounds__(256) void fp_bwd_bn_kernel(int R, int h_stride, const float* __restrict__ h,
                                                        const float* __restrict__ dy, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta, const int* __restrict__ done) {
    if (done && *done == 1) return;                   // the sums already came from the consumer's gradients
    float sb[CO], sg[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) sb[o] = sg[o] = 0.f;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < R; r += (long)gridDim.x * 256) {
        const float4* hr = reinterpret_cast<const float4*>(h + (size_t)r * h_stride);
        const float4* dr = reinterpret_cast<const float4*>(dy + (size_t)r * h_stride);
#pragma unroll
        for (int q = 0; q < (CO + 3) / 4; ++q) {
            const float4 hv = hr[q], dv = dr[q];
            const float hh[4] = {hv.x, hv.y, hv.z, hv.w}, dd[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int o = 4 * q + t;
                if (o < CO) {
                    sb[o] += dd[t];
                    sg[o] = fmaf(dd[t], (hh[t] - mean[o]) * invstd[o], sg[o]);
                }
            }
        }
    }
    __shared__ float s_red[2 * CO];
    for (int i = threadIdx.x; i < 2 * CO; i += 256) s_red[i] = 0.f;
    __syncthreads();
    sums_to_lds<CO>(sb, s_red);
    sums_to_lds<CO>(sg, s_red + CO);
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * CO; i += 256) {
        const float v = s_red[i];
        if (v != 0.f) atomicAdd(i < CO ? &dbeta[i] : &dgamma[i - CO], v);
    }
}

// the same sums for the small layers (4k-16k rows): lane = channel, wave = row subgroup, 64 rows per workgroup -- coalesced
// row reads, no cross-lane reduction (the row-per-lane form spent ~20 us per launch on 128 wave reductions for 1 MB)
template <int CO>
__global__ __launch_bounds__(256) void fp_bwd_bn_small_kernel(int R, int h_stride, const float* __restrict__ h,
                                                              const float* __restrict__ dy, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, const int* __restrict__ done) {
    static_assert(CO <= 64, "one lane per channel");
    if (done && *done == 1) return;
    __shared__ float s_part[2][4][64];
    const int o = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool on = o < CO;
    const float mu = on ? mean[o] : 0.f, is = on ? invstd[o] : 0.f;
    float sb = 0.f, sg = 0.f;
    const long r0 = (long)blockIdx.x * 64;
    float hv[16], dv[16];                               // all 32 loads in flight (four at a time: 10.7 us for 4096 rows)
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const long r = r0 + w + 4 * u;
        const bool ok = r < R && on;
        hv[u] = ok ? h[(size_t)r * h_stride + o] : mu;
        dv[u] = ok ? dy[(size_t)r * h_stride + o] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        sb += dv[u];
        sg = fmaf(dv[u], (hv[u] - mu) * is, sg);
    }
    s_part[0][w][o] = sb;
    s_part[1][w][o] = sg;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6;
        if (on) {
            const float v = (s_part[which][0][o] + s_part[which][1][o]) + (s_part[which][2][o] + s_part[which][3][o]);
            if (v != 0.f) atomicAdd(which == 0 ? &dbeta[o] : &dgamma[o], v);
        }
    }
}

// ---------------------------------------------------------------------------------------------- backward (2)
template <int CA, int CB, int CO, bool KNN, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void fp_bwd_main_kernel(
    int R, int R_per_plot, int S_per_plot, int src_stride, int skip_stride, int h_stride, int dskip_stride,
    int du_stride, float invR,
    const float* __restrict__ src, const float* __restrict__ src_a, const float* __restrict__ src_c,
    const int* __restrict__ knn_idx, const float* __restrict__ knn_w, const float* __restrict__ skip,
    const float* __restrict__ Wg, const float* __restrict__ gammag, const float* __restrict__ meang,
    const float* __restrict__ invstdg, const float* __restrict__ dgammag, const float* __restrict__ dbetag,
    const float* __restrict__ h, const float* __restrict__ dy, float* __restrict__ dW, float* __restrict__ db,
    float* __restrict__ du_out /* KNN: (R,CA) scratch; else ACCUMULATED rows (R,du_stride) */,
    float* __restrict__ dskip, int rep_k, int rep_stride) {
    constexpr int CI = CA + CB;
    using Acc = OuterAcc<CO, CI + 1, 32>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* lds = smem + (threadIdx.x >> 6) * Acc::LDS_FLOATS;
    Acc acc;
    acc.init(lds);
    // W as the B operand of the input-gradient product: lane (qq, cc) keeps W[o = 4 kb + qq][col = 16 jt + cc]
    constexpr int KBO = (CO + 3) / 4, TJ = (CI + 15) / 16;
    const int qq = (threadIdx.x & 63) >> 4, cc = threadIdx.x & 15;
    float Wb[KBO][TJ];
#pragma unroll
    for (int kb = 0; kb < KBO; ++kb)
#pragma unroll
        for (int jt = 0; jt < TJ; ++jt) {
            const int o = 4 * kb + qq, col = 16 * jt + cc;
            Wb[kb][jt] = (o < CO && col < CI) ? Wg[o * CI + col] : 0.f;
        }
    const long nthreads = (long)gridDim.x * WAVES * 64;
    const long rounds = (R + nthreads - 1) / nthreads;  // every lane of a wave runs the same number of rounds
    for (long it = 0; it < rounds; ++it) {
        const long r = it * nthreads + (long)blockIdx.x * WAVES * 64 + threadIdx.x;
        const bool valid = r < R;
        const size_t rr = valid ? (size_t)r : 0;
        const cfp W = opaque(as_const(Wg)), gamma = opaque(as_const(gammag)), mean = opaque(as_const(meang)),
                  invstd = opaque(as_const(invstdg)), dgamma = opaque(as_const(dgammag)), dbeta = opaque(as_const(dbetag));
        float u[CI + 1];
        build_input<CA, CB, KNN>(src, src_stride, opaque(as_const(src_a)), opaque(as_const(src_c)), knn_idx, knn_w, skip,
                                 skip_stride, rr, (rr / R_per_plot) * S_per_plot, u);
        float dp[CO];
        const float4* hr = reinterpret_cast<const float4*>(h + rr * h_stride);
        const float4* dr = reinterpret_cast<const float4*>(dy + rr * h_stride);
#pragma unroll
        for (int q = 0; q < (CO + 3) / 4; ++q) {
            const float4 hv = hr[q], dv = dr[q];
            const float hh[4] = {hv.x, hv.y, hv.z, hv.w}, dd[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int o = 4 * q + t;
                if (o < CO) {
                    const float is = invstd[o];
                    const float xh = (hh[t] - mean[o]) * is;
                    const float dh = gamma[o] * is * (dd[t] - dbeta[o] * invR - xh * dgamma[o] * invR);
                    dp[o] = (valid && hh[t] > 0.f) ? dh : 0.f;
                }
            }
        }
        // weight gradient: dW += dp^T [u | 1] over the wave's rows.  Input gradient: d[u] = dp W over the same staged dp
        // rows, also on the matrix cores -- A[row][o] read back from the staging region, B[o][col] = W in registers for the
        // whole kernel (as per-lane FMA chains this product was ~1400 FMAs per row fed by ~300 scalar loads of 16 dwords,
        // and the waves spent most of their time waiting for those loads).
        const long wave_row0 = it * nthreads + (long)blockIdx.x * WAVES * 64 + (threadIdx.x & ~63);
        acc.add_then(lds, dp, u, [&](int hph) __attribute__((always_inline)) {
            if (!du_out && !(CB > 0 && dskip)) return;
#pragma unroll
            for (int t = 0; t < Acc::STAGED_ROWS / 16; ++t) {
                f32x4 D[TJ];
#pragma unroll
                for (int jt = 0; jt < TJ; ++jt) D[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < KBO; ++kb) {
                    const float av = lds[(16 * t + cc) * Acc::PS + 4 * kb + qq];
#pragma unroll
                    for (int jt = 0; jt < TJ; ++jt) D[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Wb[kb][jt], D[jt], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long row = wave_row0 + hph * Acc::STAGED_ROWS + 16 * t + 4 * qq + r;
                    if (row >= R) continue;
#pragma unroll
                    for (int jt = 0; jt < TJ; ++jt) {
                        const int col = 16 * jt + cc;
                        if (col < CA) {
                            if (du_out) {
                                float* dst = du_out + (size_t)row * du_stride + col;
                                if constexpr (KNN) *dst = D[jt][r];
                                else *dst += D[jt][r];
                            }
                        } else if (col < CI) {
                            if constexpr (CB > 0) {
                                if (dskip) dskip[(size_t)row * dskip_stride + (col - CA)] += D[jt][r];
                            }
                        }
                    }
                }
            }
        });
    }
    // workgroup-level reduction of the [dW | db] image (every wave's slab in its own staging region, added in wave order;
    // LDS float atomics where a slab does not fit the region), then one global atomic per element and workgroup
    constexpr bool SLAB = CO * (CI + 1) <= Acc::LDS_FLOATS;
    float* red = smem;
    if constexpr (SLAB) {
        acc.store_slab(lds);
        __syncthreads();
    } else {
        __syncthreads();
        for (int i = threadIdx.x; i < CO * (CI + 1); i += WAVES * 64) red[i] = 0.f;
        __syncthreads();
        acc.flush_lds(red);
        __syncthreads();
    }
    for (int i = threadIdx.x; i < CO * (CI + 1); i += WAVES * 64) {
        float v = 0.f;
        if constexpr (SLAB) {
#pragma unroll
            for (int w = 0; w < WAVES; ++w) v += smem[w * Acc::LDS_FLOATS + i];
        } else {
            v = red[i];
        }
        if (v == 0.f) continue;
        const int o = i / (CI + 1), k = i - o * (CI + 1), img = sn2_grad_image(rep_k, rep_stride);
        if (k < CI) SN2_FLUSH_ADD(&dW[img + o * CI + k], v);
        else SN2_FLUSH_ADD(&db[img + o], v);
    }
}

// ---------------------------------------------------------------------------------------------- backward (3)
// Transpose of the interpolation:  dsrc[s][k] += sum over (target r, slot j) with idx_j(r) = s of (w_j / sum w) * du[r][k].
// Done as a GATHER through an inverted index (source -> list of (row, weight)), built per call from the saved 3-NN table:
//   A  per (plot, row slice): histogram of the slice's source ids in LDS (integer atomics on 4 KB) -> H[plot][slice][s]
//   B  per plot: exclusive prefix over the slices of every source and over the sources -> list offsets
//   C  per (plot, row slice): LDS cursors -> (row, normalised weight) entries at their final positions
//   D  one wave per source row, lane = channel: coalesced du rows, accumulation in registers, one plain store.
// No floating-point atomics anywhere (LDS float atomics ran at ~0.4 lane-ops/clk/CU here: 275 us for FP1; global float
// atomics onto random rows are worse), and dsrc is written exactly once per row.
constexpr int INV_SLICE_ROWS = 2048;

// The workspace of ONE batch's inverted index (SN2_INTERP_WS_WORDS(B, Rp, S) 32-bit words), carved the same way on the host
// (carve_interp_index below) and inside the kernels that build it:
//   H [B*SL*S] | off [B*S] | cnt [B*S] | inv_row [3*B*Rp] | inv_w [3*B*Rp] | (16-byte aligned) items [B*S] int4 | chunks [B*CM] int4
// GROUPED builds (round 5: sn2_interp_index_group): one launch covers G consecutive batches of B plots each -- plot bg of the
// launch is plot bg % B of batch bg / B, whose workspace starts ws_stride words behind the previous batch's -- so that the
// position-only pass of a pipelined loop, which samples eight batches in one FPS launch, builds their inverted indices in 12
// launches instead of 96.  Every batch's workspace is an ordinary B-plot workspace: its consumers do not change.
struct InvWs {
    int *H, *off, *cnt, *inv_row;
    float* inv_w;
    int4 *items, *chunks;
};
__host__ __device__ __forceinline__ InvWs inv_ws_of(float* ws, int B, int Rp, int S) {
    const int SL = (Rp + INV_SLICE_ROWS - 1) / INV_SLICE_ROWS;
    InvWs x;
    x.H = reinterpret_cast<int*>(ws);
    x.off = x.H + (size_t)B * SL * S;
    x.cnt = x.off + (size_t)B * S;
    x.inv_row = x.cnt + (size_t)B * S;
    x.inv_w = reinterpret_cast<float*>(x.inv_row + (size_t)3 * B * Rp);
    x.items = reinterpret_cast<int4*>((reinterpret_cast<uintptr_t>(x.inv_w + (size_t)3 * B * Rp) + 15) & ~(uintptr_t)15);
    x.chunks = x.items + (size_t)B * S;
    return x;
}

__global__ __launch_bounds__(1024) void inv_hist_kernel(int R_per_plot, int S, const int* __restrict__ knn_idx,
                                                        const float* __restrict__ knn_w, float* __restrict__ ws, int Bb,
                                                        size_t ws_stride) {
    extern __shared__ int s_hist[];
    const int bg = blockIdx.y, sl = blockIdx.x, SL = gridDim.x;
    const int hb = bg / Bb, b = bg - hb * Bb;                    // batch of the group, plot of the batch
    int* H = inv_ws_of(ws + (size_t)hb * ws_stride, Bb, R_per_plot, S).H;
    for (int i = threadIdx.x; i < S; i += 1024) s_hist[i] = 0;
    __syncthreads();
    const int r_lo = sl * INV_SLICE_ROWS, r_hi = min(R_per_plot, r_lo + INV_SLICE_ROWS);
    for (int rl = r_lo + threadIdx.x; rl < r_hi; rl += 1024) {
        const size_t r = (size_t)bg * R_per_plot + rl;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j == 0 || knn_w[r * 3 + j] != 0.f) atomicAdd(&s_hist[knn_idx[r * 3 + j]], 1);
    }
    __syncthreads();
    int* out = H + ((size_t)b * SL + sl) * S;
    for (int i = threadIdx.x; i < S; i += 1024) out[i] = s_hist[i];
}

// H[plot][slice][s] -> exclusive prefix over slices (in place);  off[plot*S + s] = plot*3*R + exclusive scan of the totals;
// cnt[plot*S + s] = total
__global__ __launch_bounds__(1024) void inv_scan_kernel(int R_per_plot, int S, int SL, float* __restrict__ ws, int Bb,
                                                        size_t ws_stride) {
    __shared__ int s_w[16];
    __shared__ int s_carry;
    const int bg = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hb = bg / Bb, b = bg - hb * Bb;
    const InvWs x = inv_ws_of(ws + (size_t)hb * ws_stride, Bb, R_per_plot, S);
    int* __restrict__ H = x.H;
    int* __restrict__ off = x.off;
    int* __restrict__ cnt = x.cnt;
    if (threadIdx.x == 0) s_carry = b * 3 * R_per_plot;
    __syncthreads();
    for (int s0 = 0; s0 < S; s0 += 1024) {
        const int s = s0 + threadIdx.x;
        int tot = 0;
        if (s < S) {
            for (int sl = 0; sl < SL; ++sl) {
                int* h = H + ((size_t)b * SL + sl) * S + s;
                const int t = *h;
                *h = tot;
                tot += t;
            }
            cnt[(size_t)b * S + s] = tot;
        }
        int incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        int base = s_carry;
        for (int k = 0; k < wave; ++k) base += s_w[k];
        if (s < S) off[(size_t)b * S + s] = base + incl - tot;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = base + incl;
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void inv_fill_kernel(int R_per_plot, int S, const int* __restrict__ knn_idx,
                                                        const float* __restrict__ knn_w, float* __restrict__ ws, int Bb,
                                                        size_t ws_stride, const int* __restrict__ row_perm) {
    extern __shared__ int s_cur[];
    const int bg = blockIdx.y, sl = blockIdx.x, SL = gridDim.x;
    const int hb = bg / Bb, b = bg - hb * Bb;
    const InvWs x = inv_ws_of(ws + (size_t)hb * ws_stride, Bb, R_per_plot, S);
    const int* __restrict__ H = x.H;
    const int* __restrict__ off = x.off;
    int* __restrict__ inv_row = x.inv_row;
    float* __restrict__ inv_w = x.inv_w;
    const int* hp = H + ((size_t)b * SL + sl) * S;
    for (int i = threadIdx.x; i < S; i += 1024) s_cur[i] = off[(size_t)b * S + i] + hp[i];
    __syncthreads();
    const int r_lo = sl * INV_SLICE_ROWS, r_hi = min(R_per_plot, r_lo + INV_SLICE_ROWS);
    for (int rl = r_lo + threadIdx.x; rl < r_hi; rl += 1024) {
        const size_t r = (size_t)bg * R_per_plot + rl;
        const float w0 = knn_w[r * 3 + 0], w1 = knn_w[r * 3 + 1], w2 = knn_w[r * 3 + 2];
        const float inv = 1.0f / ((w0 + w1) + w2);
        const float w[3] = {w0, w1, w2};
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j == 0 || w[j] != 0.f) {
                const int p = atomicAdd(&s_cur[knn_idx[r * 3 + j]], 1);
                inv_row[p] = row_perm ? row_perm[r] : rl;          // where the row's d pre-activation is kept (sn2_fp.row_perm)
                inv_w[p] = w[j] * inv;
            }
    }
}

// E  the plot's sources along a Morton curve (identity without positions): items[plot*S + rank] = {source id, list offset,
//    list length, -}: one load tells a wave of the source-side kernel all about its source.  The source-side kernels walk
//    the sources in this order, one contiguous stretch of it per XCD, so that the target rows shared by neighbouring
//    sources (every target row is on the lists of its three nearest sources) are fetched into that XCD's L2 once.
//    Keys: 10 bits per axis inside the plot's bounding box; ranks by comparison counting in LDS (ties by id).
__device__ __forceinline__ unsigned spread10(unsigned v) {
    v &= 1023u;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
//    F  the lists cut into CHUNKS of at most INV_CHUNK entries, in the same order: chunks[plot*CM + j] = {source id, offset of
//    the chunk's first entry, its length, plot}, CM = inv_chunks_per_plot (unused slots: length 0), and items[].w = the plot-local
//    number of the source's first chunk.  The lists are anything but even (C2: median 25 entries, a tenth of the sources
//    500-800: the synthetic stands are clumped like real ones), and a wave per SOURCE left the source pass waiting for a few
//    waves that walk 13 chunks one after the other; a wave per CHUNK has one short chain for everybody.
constexpr int INV_CHUNK = 63;
__host__ __device__ constexpr int inv_chunks_per_plot(int Rp, int S) { return (3 * Rp + INV_CHUNK - 1) / INV_CHUNK + S; }

__global__ __launch_bounds__(1024) void inv_order_kernel(const float4* __restrict__ pos, int S, int CM, int R_per_plot,
                                                         float* __restrict__ ws, int Bb, size_t ws_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned s_key[];   // [S rounded up to 4] keys | [S] source of every rank
    __shared__ float s_lo[3][16], s_hi[3][16];
    __shared__ int s_w[16];
    __shared__ int s_carry;
    const int bg = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hb = bg / Bb, b = bg - hb * Bb;
    const InvWs x = inv_ws_of(ws + (size_t)hb * ws_stride, Bb, R_per_plot, S);
    const int* __restrict__ off = x.off;
    const int* __restrict__ cnt = x.cnt;
    int4* __restrict__ items = x.items;
    int4* __restrict__ chunks = x.chunks;
    const float4* pb = pos + (size_t)bg * S;
    const int S4 = (S + 3) & ~3;
    int* s_ord = reinterpret_cast<int*>(s_key + S4);
    if (pos) {
        float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        for (int i = threadIdx.x; i < S; i += 1024) {
            const float4 p = pb[i];
            lo[0] = fminf(lo[0], p.x), lo[1] = fminf(lo[1], p.y), lo[2] = fminf(lo[2], p.z);
            hi[0] = fmaxf(hi[0], p.x), hi[1] = fmaxf(hi[1], p.y), hi[2] = fmaxf(hi[2], p.z);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                lo[a] = fminf(lo[a], __shfl_xor(lo[a], o));
                hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o));
            }
            if (lane == 0) s_lo[a][wave] = lo[a], s_hi[a][wave] = hi[a];
        }
        __syncthreads();
        float sc[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float l = s_lo[a][0], h = s_hi[a][0];
            for (int w = 1; w < 16; ++w) l = fminf(l, s_lo[a][w]), h = fmaxf(h, s_hi[a][w]);
            lo[a] = l;
            sc[a] = h > l ? 1023.999f / (h - l) : 0.f;
        }
        for (int i = threadIdx.x; i < S4; i += 1024) {
            unsigned key = 0xFFFFFFFFu;                                    // padding sorts last
            if (i < S) {
                const float4 p = pb[i];
                const unsigned qx = (unsigned)((p.x - lo[0]) * sc[0]), qy = (unsigned)((p.y - lo[1]) * sc[1]),
                               qz = (unsigned)((p.z - lo[2]) * sc[2]);
                key = spread10(qx) | (spread10(qy) << 1) | (spread10(qz) << 2);
            }
            s_key[i] = key;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < S; i += 1024) {
            const unsigned mine = s_key[i];
            int rank = 0;
            for (int j = 0; j < S4; j += 4) {
                const uint4 o = *reinterpret_cast<const uint4*>(&s_key[j]);
                rank += (o.x < mine || (o.x == mine && j < i)) ? 1 : 0;
                rank += (o.y < mine || (o.y == mine && j + 1 < i)) ? 1 : 0;
                rank += (o.z < mine || (o.z == mine && j + 2 < i)) ? 1 : 0;
                rank += (o.w < mine || (o.w == mine && j + 3 < i)) ? 1 : 0;
            }
            s_ord[rank] = i;
        }
    } else {                                                           // no positions: identity
        for (int i = threadIdx.x; i < S; i += 1024) s_ord[i] = i;
    }
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int k0 = 0; k0 < S; k0 += 1024) {
        const int k = k0 + threadIdx.x;
        int id = 0, o = 0, n = 0, nch = 0;
        if (k < S) {
            id = b * S + s_ord[k];
            o = off[id], n = cnt[id];
            nch = (n + INV_CHUNK - 1) / INV_CHUNK;
        }
        int incl = nch;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        int base = s_carry;
        for (int w = 0; w < wave; ++w) base += s_w[w];
        const int first = base + incl - nch;
        if (k < S) {
            items[(size_t)b * S + k] = make_int4(id, o, n, first);
            for (int c = 0; c < nch; ++c)
                chunks[(size_t)b * CM + first + c] = make_int4(id, o + c * INV_CHUNK, min(INV_CHUNK, n - c * INV_CHUNK), b);
        }
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = base + incl;
        __syncthreads();
    }
    for (int j = s_carry + threadIdx.x; j < CM; j += 1024) chunks[(size_t)b * CM + j] = make_int4(0, 0, 0, 0);
}

template <int CA>
__global__ __launch_bounds__(256) void interp_gather_kernel(int n_src, int R_per_plot, int S, int dsrc_stride,
                                                            const int* __restrict__ off, const int* __restrict__ cnt,
                                                            const int* __restrict__ inv_row, const float* __restrict__ inv_w,
                                                            const float* __restrict__ du, float* __restrict__ dsrc) {
    const int lane = threadIdx.x & 63;
    const int s = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (s >= n_src) return;
    const int b = s / S;
    const int n = cnt[s], st = off[s];
    const float* dub = du + (size_t)b * R_per_plot * CA;
    const bool on = lane < CA;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
    // The list is ~96 entries long and every entry is one 136-byte row somewhere in a 70 MB array: pure latency.  64 list
    // entries are fetched with one coalesced load (lane j keeps entry j, broadcast by v_readlane), then eight independent
    // row loads are in flight at a time (four gave 1.7 TB/s).
    for (int base = 0; base < n; base += 64) {
        const int m = (n - base) < 64 ? (n - base) : 64;
        const int rj = lane < m ? inv_row[st + base + lane] : 0;          // entries past the end: row 0 with weight 0
        const float wj = lane < m ? inv_w[st + base + lane] : 0.f;
        for (int i = 0; i < m; i += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = __builtin_amdgcn_readlane(rj, i + u);
                v[u] = on ? dub[(size_t)r * CA + lane] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; u += 4) {
                g0 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(wj), i + u)), v[u], g0);
                g1 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(wj), i + u + 1)), v[u + 1], g1);
                g2 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(wj), i + u + 2)), v[u + 2], g2);
                g3 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(wj), i + u + 3)), v[u + 3], g3);
            }
        }
    }
    if (on) dsrc[(size_t)s * dsrc_stride + lane] += (g0 + g1) + (g2 + g3);
}

// The same for LONG lists (the global level: one source per plot, every target row on its list): one workgroup of WAVES
// waves per source, wave w takes the entries w, w + WAVES, ... in blocks of 64, the partial sums are added in wave order
// (one wave walking 256 entries, eight in flight, took 15 us).
template <int CA, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void interp_gather_long_kernel(int n_src, int R_per_plot, int S, int dsrc_stride,
                                                                        const int* __restrict__ off, const int* __restrict__ cnt,
                                                                        const int* __restrict__ inv_row,
                                                                        const float* __restrict__ inv_w,
                                                                        const float* __restrict__ du, float* __restrict__ dsrc) {
    __shared__ float s_part[WAVES][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int s = blockIdx.x;
    const int b = s / S;
    const int n = cnt[s], st = off[s];
    const float* dub = du + (size_t)b * R_per_plot * CA;
    const bool on = lane < CA;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
    for (int base = wave * 8; base < n; base += WAVES * 8) {           // eight entries per wave and turn, all in flight
        const int m = (n - base) < 8 ? (n - base) : 8;
        const int rj = lane < m ? inv_row[st + base + lane] : 0;        // entries past the end: row 0 with weight 0
        const float wj = lane < m ? inv_w[st + base + lane] : 0.f;
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = __builtin_amdgcn_readlane(rj, u);
            v[u] = on ? dub[(size_t)r * CA + lane] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; u += 4) {
            g0 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(wj), u)), v[u], g0);
            g1 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(wj), u + 1)), v[u + 1], g1);
            g2 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(wj), u + 2)), v[u + 2], g2);
            g3 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(wj), u + 3)), v[u + 3], g3);
        }
    }
    s_part[wave][lane] = (g0 + g1) + (g2 + g3);
    __syncthreads();
    if (wave == 0 && on) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) t += s_part[w][lane];
        dsrc[(size_t)s * dsrc_stride + lane] += t;
    }
}

// ---------------------------------------------------------------------------------------------- source-side form
// The per-point layer (FP1: 524 288 rows interpolating 16 384 source rows) spent most of its time on work that is linear
// in the interpolation and can be done once per SOURCE row instead of once per target row:
//   W_A u_r = W_A (a o sum_j w_rj src[i_rj] + c) = sum_j w_rj T[i_rj],   T[s] = W_A (a o src[s] + c)      (sum_j w_rj = 1)
//   dsrc[s] = sum_{r,j: i_rj = s} w_rj dp_r W_A = G[s] W_A,              G[s] = sum_{r,j: i_rj = s} w_rj dp_r
//   dW_A    = sum_r dp_r^T u_r = sum_s G[s]^T (a o src[s] + c)
// so the row side keeps only the skip columns (CB = 8 of 42 inputs: 5x fewer multiply-adds) and becomes a streaming pass:
//   forward : T (n_src x CO, one small kernel), then per row 3 gathers of T rows + W_B skip + b -> relu -> h, statistics
//   backward: rows: dp = BN/ReLU backward of dy (stored once), dW_B | db;  sources: G (gather of dp rows through the
//             inverted index), dsrc += G W_A;  dW_A += G^T (a o src + c) on the matrix cores over the n_src rows.
// The row kernels use QH = ceil(CO/4) consecutive lanes per row, one float4 quad each: a load or store instruction covers
// 64/QH whole rows = ~1 KB of consecutive bytes, and nothing but the lane's own quad constants lives in registers.
// Results differ from the row-per-lane form by fp32 re-association only.
// ---- rows of per-point activations in either storage precision (sn2_fp.act_bf16 / sn2_head.act_bf16).  A row has `stride`
// ELEMENTS either way; quad q = elements 4q .. 4q+3: one 16-byte (fp32) or one 8-byte (bfloat16) access.  bfloat16 rows are
// written with v_cvt_pk_bf16_f32 (round to nearest even) and read back exactly (a bfloat16 IS the upper half of an fp32).
template <bool BF>
__device__ __forceinline__ float4 row_quad_ld(const float* __restrict__ base, size_t row, int stride, int q) {
    if constexpr (BF) {
        const uint2 u = reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + row * stride)[q];
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                           __uint_as_float(u.y & 0xFFFF0000u));
    } else {
        return reinterpret_cast<const float4*>(base + row * stride)[q];
    }
}
__device__ __forceinline__ uint2 pack_bf16x4(float a, float b, float c, float d) {
    bf16x4 v;
    v[0] = (__bf16)a; v[1] = (__bf16)b; v[2] = (__bf16)c; v[3] = (__bf16)d;
    return __builtin_bit_cast(uint2, v);
}
template <bool BF>
__device__ __forceinline__ void row_quad_st(float* __restrict__ base, size_t row, int stride, int q, float a, float b, float c, float d) {
    if constexpr (BF) reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + row * stride)[q] = pack_bf16x4(a, b, c, d);
    else reinterpret_cast<float4*>(base + row * stride)[q] = make_float4(a, b, c, d);
}
// the value a bfloat16 store keeps (for sums that must describe the STORED rows)
__device__ __forceinline__ float bf16_round(float x) { return (float)(__bf16)x; }

// which kernel builds the source table: 1 (default) fp_src_table_mfma_kernel, 0 fp_src_table_kernel (test hook:
// sn2_debug_fp_table_form)
static int g_fp_table_form = (getenv("SN2_FP_TABLE_MFMA") && atoi(getenv("SN2_FP_TABLE_MFMA")) == 0) ? 0 : 1;
template <int CA, int CB, int CO>
__global__ __launch_bounds__(256) void fp_src_table_kernel(int n_src, int src_stride, const float* __restrict__ src,
                                                           const float* __restrict__ src_a, const float* __restrict__ src_c,
                                                           const float* __restrict__ Wg, float* __restrict__ T) {
    // 64 source rows per workgroup, wave g = output channels [g*QH, (g+1)*QH): 4x the waves, 4x shorter FMA chains
    constexpr int CI = CA + CB, QH = (CO + 3) / 4, HS = 4 * QH;
    const int s = blockIdx.x * 64 + (threadIdx.x & 63), grp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const size_t ss = s < n_src ? (size_t)s : 0;
    const cfp W = opaque(as_const(Wg));
    float x[CA];
    const float4* sr = reinterpret_cast<const float4*>(src + ss * src_stride);
#pragma unroll
    for (int q4 = 0; q4 < (CA + 3) / 4; ++q4) {
        const float4 a = sr[q4];
        const float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (4 * q4 + t < CA) x[4 * q4 + t] = v[t];
    }
    if (src_a) {
        const cfp sa = opaque(as_const(src_a)), sc = opaque(as_const(src_c));
#pragma unroll
        for (int k = 0; k < CA; ++k) x[k] = fmaf(sa[k], x[k], sc[k]);
    }
    float* out = T + ss * HS + grp * QH;
#pragma unroll
    for (int j = 0; j < QH; ++j) {
        const int o = grp * QH + j;                     // wave-uniform
        float acc = 0.f;
        if (o < CO) {
#pragma unroll
            for (int k = 0; k < CA; ++k) acc = fmaf(W[o * CI + k], x[k], acc);
        }
        if (s < n_src) out[j] = acc;
    }
}

// a wave re-reads LDS words other lanes of the SAME wave wrote: the LDS executes a wave's instructions in order, the compiler
// must not move the accesses across this point
#define WAVE_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// The same table on the matrix cores (round 5): a wave takes 64 consecutive source rows -- fetched with coalesced float4 loads into
// an LDS tile, the BatchNorm affine of the layer in front applied on the way out of it --, contracts them with W_A held in
// registers (`v_mfma_f32_16x16x4_f32`, k ascending: the exact fp32 products and the accumulation order of the scalar kernel's
// fmaf chain), and writes the 64 table rows back through the tile as coalesced float4.  fp_src_table_kernel gives every lane a
// row and takes its weights through scalar loads, ~1200 FMA instructions per row-lane behind 144-byte strided loads: 243 us for
// the parcel loop's 1.28 M sources (1.5 TB/s); this form streams.
template <int CA, int CB, int CO>
__global__ __launch_bounds__(256) void fp_src_table_mfma_kernel(int n_src, int src_stride, const float* __restrict__ src,
                                                                const float* __restrict__ src_a, const float* __restrict__ src_c,
                                                                const float* __restrict__ Wg, float* __restrict__ T) {
    constexpr int CI = CA + CB, QH = (CO + 3) / 4, HS = 4 * QH, KS = (CA + 3) / 4, TJ = (HS + 15) / 16;
    // tile row stride: the source row, padded so that the sixteen rows of an A-operand read sit in different banks
    constexpr int LS = (4 * KS) % 32 == 0 ? 4 * KS + 4 : 4 * KS;
    static_assert(HS <= LS, "the table rows go back through the tile");
    extern __shared__ __attribute__((aligned(16))) float s_tile[];            // [4 waves][64][LS]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* st = s_tile + (size_t)wave * 64 * LS;
    const int n = lane & 15, kq = lane >> 4;
    float wb[TJ][KS], ak[KS], ck[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k = 4 * ks + kq;
        ak[ks] = k < CA ? (src_a ? src_a[k] : 1.f) : 0.f;
        ck[ks] = (k < CA && src_a) ? src_c[k] : 0.f;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int o = 16 * j + n;
            wb[j][ks] = (o < CO && k < CA) ? Wg[o * CI + k] : 0.f;
        }
    }
    const long n_turns = ((long)n_src + 63) / 64;
    for (long turn = (long)blockIdx.x * 4 + wave; turn < n_turns; turn += (long)gridDim.x * 4) {
        const long s0 = turn * 64;
        // ---- 64 rows x KS quads, coalesced; rows past the end: the last row (never written back)
        const int rows_here = n_src - s0 < 64 ? (int)(n_src - s0) : 64;
        const float* base = src + (size_t)s0 * src_stride;
        const int QR = src_stride / 4;                                      // quads of a source row in memory
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            const int e = lane + 64 * i;                                     // quad e of the 64 x KS quads this wave wants
            const int r = e / KS, qk = e - r * KS;
            const int rc = r < rows_here ? r : rows_here - 1;
            const float4 v = reinterpret_cast<const float4*>(base + (size_t)rc * src_stride)[qk < QR ? qk : QR - 1];
            *reinterpret_cast<float4*>(&st[r * LS + 4 * qk]) = v;
        }
        WAVE_LDS_SYNC();
        f32x4 acc[4][TJ];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const float x = st[(16 * t + n) * LS + 4 * ks + kq];
                const float a = (4 * ks + kq < CA) ? (src_a ? fmaf(ak[ks], x, ck[ks]) : x) : 0.f;
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[j][ks], a, acc[t][j], 0, 0, 0);
            }
        WAVE_LDS_SYNC();
        // acc[t][j][r]: output channel 16 j + 4 kq + r of source row 16 t + n  (A = weights: rows of D are channels)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int o = 16 * j + 4 * kq;
                if (o < HS) *reinterpret_cast<float4*>(&st[(16 * t + n) * LS + o]) = make_float4(acc[t][j][0], acc[t][j][1], acc[t][j][2], acc[t][j][3]);
            }
        WAVE_LDS_SYNC();
        float* out = T + (size_t)s0 * HS;
#pragma unroll
        for (int i = 0; i < QH; ++i) {
            const int e = lane + 64 * i;
            const int r = e / QH, qo = e - r * QH;
            if (r < rows_here) reinterpret_cast<float4*>(out + (size_t)r * HS)[qo] = *reinterpret_cast<const float4*>(&st[r * LS + 4 * qo]);
        }
        WAVE_LDS_SYNC();
    }
}
template <int CA, int CB, int CO>
int launch_src_table(int n_src, int src_stride, const float* src, const float* src_a, const float* src_c, const float* W, float* T,
                     hipStream_t st) {
    // where it pays: many sources of the 34-channel layer (the parcel loop's 1.28 M: 243 -> 122 us).  Not the 64-channel layer
    // (68-word tile rows, two workgroups per CU: 85 -> 99 us) and not a training batch's 16 384 sources (64 workgroups, each one
    // long chain: 6.9 -> 9.1 us).  Both kernels give the same bits (tests), so the choice is free.
    if (g_fp_table_form != 0 && CA <= 36 && n_src >= 65536 && src_stride >= 4 * ((CA + 3) / 4)) {
        constexpr int KS = (CA + 3) / 4, LS = (4 * KS) % 32 == 0 ? 4 * KS + 4 : 4 * KS;
        const size_t lds = (size_t)4 * 64 * LS * sizeof(float);
        auto k = &fp_src_table_mfma_kernel<CA, CB, CO>;
        if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        long grid = ((long)n_src + 255) / 256;
        const long cap = 4L * sn2_cu_count();
        if (grid > cap) grid = cap;
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), lds, st, n_src, src_stride, src, src_a, src_c, W, T);
    } else {
        hipLaunchKernelGGL((fp_src_table_kernel<CA, CB, CO>), dim3(sn2_cdiv(n_src, 64)), dim3(256), 0, st, n_src, src_stride, src, src_a,
                           src_c, W, T);
    }
    SN2_RETURN_LAUNCH();
}

// The interpolated part of a pre-activation, in ONE spelled-out order of operations -- fma(fma(fma(c, w2, fma(b, w1, a w0)) ...:
//   s = a w0;  s = fma(b, w1, s);  s = fma(c, w2, s);  acc = fma(s, 1 / sum w, bias)
// Left to the compiler's contraction, the two row passes below (same source text) fused different products and differed in the
// last bit of every fourth column.  fp_fwd_rows2_kernel issues the same operations two channels at a time (v_pk_mul_f32 /
// v_pk_fma_f32: IEEE per component, the same bits).
__device__ __forceinline__ float interp_bias(float a, float b, float c, float w0, float w1, float w2, float inv, float bias) {
    float s2;
    {
#pragma clang fp contract(off)
        s2 = a * w0;
    }
    s2 = fmaf(b, w1, s2);
    s2 = fmaf(c, w2, s2);
    return fmaf(s2, inv, bias);
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 interp_bias2(f32x2 a, f32x2 b, f32x2 c, float w0, float w1, float w2, float inv, f32x2 bias) {
    f32x2 s2;
    {
#pragma clang fp contract(off)
        s2 = a * (f32x2){w0, w0};
    }
    s2 = __builtin_elementwise_fma(b, (f32x2){w1, w1}, s2);
    s2 = __builtin_elementwise_fma(c, (f32x2){w2, w2}, s2);
    return __builtin_elementwise_fma(s2, (f32x2){inv, inv}, bias);
}
// Which iterations (FP_ROWS_PER_IT consecutive rows each) a wave of the row kernels works on: it0, it0 + stride, ... < it_hi.
// Workgroups go to the XCDs round-robin (blockIdx % 8), and the iterations are dealt the same way when they go wave after wave --
// every XCD's L2 then holds the table rows of ALL plots (2.4 MB of a 4 MB L2 at config 2) beside the rows streaming through it.
// With the grid a multiple of 8, XCD x takes the x-th EIGHTH of the rows instead (whole plots where the batch is a multiple of
// eight plots): its L2 holds an eighth of the table.
struct RowIters {
    int it0, stride, it_hi;
};
__device__ __forceinline__ RowIters row_iters(int n_it, int wave, int wpw = 4) {           // wpw: waves per workgroup
    RowIters r;
    if ((gridDim.x & 7) == 0) {
        const int xcd = blockIdx.x & 7, wg_x = blockIdx.x >> 3, n_wg_x = gridDim.x >> 3;
        const int lo = (int)((long)n_it * xcd / 8);
        r.it_hi = (int)((long)n_it * (xcd + 1) / 8);
        r.stride = n_wg_x * wpw;
        r.it0 = lo + wg_x * wpw + wave;
    } else {
        r.it_hi = n_it, r.stride = (int)gridDim.x * wpw, r.it0 = (int)blockIdx.x * wpw + wave;
    }
    return r;
}
// what a (row, quad) lane of the row kernels reads ahead of its gathers: the row's 3-NN entry and skip columns
template <int QB>
struct FpRowIn {
    unsigned rr;
    bool valid;
    int i0, i1, i2;
    float w0, w1, w2;
    float4 sk[QB];
};
template <int QB>
__device__ __forceinline__ FpRowIn<QB> fp_row_in(long row, bool on, int R, const int* __restrict__ knn_idx,
                                                 const float* __restrict__ knn_w, const float* __restrict__ skip,
                                                 int skip_stride) {
    FpRowIn<QB> in;
    in.valid = on && row < R;
    in.rr = in.valid ? (unsigned)row : 0u;
    in.i0 = knn_idx[in.rr * 3 + 0], in.i1 = knn_idx[in.rr * 3 + 1], in.i2 = knn_idx[in.rr * 3 + 2];
    in.w0 = knn_w[in.rr * 3 + 0], in.w1 = knn_w[in.rr * 3 + 1], in.w2 = knn_w[in.rr * 3 + 2];
#pragma unroll
    for (int b = 0; b < QB; ++b) in.sk[b] = reinterpret_cast<const float4*>(skip + (size_t)in.rr * skip_stride)[b];
    return in;
}

template <int CA, int CB, int CO, bool BF>
__global__ __launch_bounds__(256) void fp_fwd_rows_kernel(int R, int R_per_plot, int S_per_plot, int skip_stride,
                                                          const float* __restrict__ T, const int* __restrict__ knn_idx,
                                                          const float* __restrict__ knn_w, const float* __restrict__ skip,
                                                          const float* __restrict__ Wg, const float* __restrict__ biasg,
                                                          float* __restrict__ h, float* __restrict__ slots) {
    constexpr int CI = CA + CB, QH = (CO + 3) / 4, HS = 4 * QH, G = 64 / QH, QB = CB / 4, U = 2;
    static_assert(CB > 0 && CB % 4 == 0, "skip quads");
    __shared__ float s_part[8][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane % QH, g = lane / QH;
    const bool on = lane < G * QH;
    float wB[4][CB], b4[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int o = 4 * q + t;
        b4[t] = o < CO ? biasg[o] : 0.f;
#pragma unroll
        for (int k = 0; k < CB; ++k) wB[t][k] = o < CO ? Wg[o * CI + CA + k] : 0.f;
    }
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
    // (a wave's turn = U groups of G rows = one "iteration" of row_iters)
    const RowIters ri = row_iters((int)(((long)R + G * U - 1) / (G * U)), wave);
    const long n_grp_all = ((long)R + G - 1) / G;
    const long n_grp = n_grp_all < (long)ri.it_hi * U ? n_grp_all : (long)ri.it_hi * U;
    const long n_waves = ri.stride;
    long grp0 = (long)ri.it0 * U;
    // the 3-NN entries and skip columns run one iteration ahead of the gathers that depend on them
    FpRowIn<QB> nx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) nx[u] = fp_row_in<QB>((grp0 + u) * G + g, on && grp0 < n_grp, R, knn_idx, knn_w, skip, skip_stride);
    for (; grp0 < n_grp; grp0 += n_waves * U) {
        FpRowIn<QB> in[U];
        float4 ta[U][3];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            in[u] = nx[u];
            const unsigned base = (in[u].rr / (unsigned)R_per_plot) * (unsigned)S_per_plot;
#if defined(SN2_FR_DIAG) && (SN2_FR_DIAG & 2)
            // (diagnostic: no table gathers -- values made from the indices)
            ta[u][0] = make_float4((float)(base + in[u].i0), 1.f, 2.f, 3.f);
            ta[u][1] = make_float4((float)in[u].i1, 1.f, 2.f, 3.f);
            ta[u][2] = make_float4((float)in[u].i2, 1.f, 2.f, 3.f);
#else
            ta[u][0] = reinterpret_cast<const float4*>(T + (size_t)(base + in[u].i0) * HS)[q];
            ta[u][1] = reinterpret_cast<const float4*>(T + (size_t)(base + in[u].i1) * HS)[q];
            ta[u][2] = reinterpret_cast<const float4*>(T + (size_t)(base + in[u].i2) * HS)[q];
#endif
        }
        const long grp1 = grp0 + n_waves * U;
#pragma unroll
        for (int u = 0; u < U; ++u) nx[u] = fp_row_in<QB>((grp1 + u) * G + g, on && grp1 < n_grp, R, knn_idx, knn_w, skip, skip_stride);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float w0 = in[u].w0, w1 = in[u].w1, w2 = in[u].w2;
            const float inv = 1.0f / ((w0 + w1) + w2);
            const float4 a = ta[u][0], b = ta[u][1], c = ta[u][2];
            float v[4] = {interp_bias(a.x, b.x, c.x, w0, w1, w2, inv, b4[0]), interp_bias(a.y, b.y, c.y, w0, w1, w2, inv, b4[1]),
                          interp_bias(a.z, b.z, c.z, w0, w1, w2, inv, b4[2]), interp_bias(a.w, b.w, c.w, w0, w1, w2, inv, b4[3])};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float acc = v[t];
#if defined(SN2_FR_DIAG) && (SN2_FR_DIAG & 4)
                acc += in[u].sk[0].x + in[u].sk[QB - 1].w;          // (diagnostic: no skip contraction)
#else
#pragma unroll
                for (int b2 = 0; b2 < QB; ++b2) {
                    acc = fmaf(wB[t][4 * b2 + 0], in[u].sk[b2].x, acc);
                    acc = fmaf(wB[t][4 * b2 + 1], in[u].sk[b2].y, acc);
                    acc = fmaf(wB[t][4 * b2 + 2], in[u].sk[b2].z, acc);
                    acc = fmaf(wB[t][4 * b2 + 3], in[u].sk[b2].w, acc);
                }
#endif
                acc = (in[u].valid && 4 * q + t < CO) ? fmaxf(acc, 0.f) : 0.f;
                if constexpr (BF) acc = bf16_round(acc);     // the batch statistics describe the rows as they are stored
                ssum[t] += acc;
                ssq[t] = fmaf(acc, acc, ssq[t]);
                v[t] = acc;
            }
#if defined(SN2_FR_DIAG) && (SN2_FR_DIAG & 1)
            if (in[u].valid && v[0] == 12345.678f) row_quad_st<BF>(h, in[u].rr, HS, q, v[0], v[1], v[2], v[3]);   // (diagnostic: no stores)
#else
            if (in[u].valid) row_quad_st<BF>(h, in[u].rr, HS, q, v[0], v[1], v[2], v[3]);
#endif
        }
    }
    if (!slots) return;
    // batch statistics: per-lane partials -> LDS -> one slot per workgroup ([sum(C) | sumsq(C)], as stats_to_slot)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        s_part[t][threadIdx.x] = on ? ssum[t] : 0.f;
        s_part[4 + t][threadIdx.x] = on ? ssq[t] : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * CO; e += 256) {
        const int which = e / CO, c = e - which * CO, cq = c >> 2, ct = c & 3;
        float acc = 0.f;
        for (int w = 0; w < 4; ++w)
            for (int gg = 0; gg < G; ++gg) acc += s_part[which * 4 + ct][w * 64 + cq + QH * gg];
        slots[(size_t)blockIdx.x * 2 * CO + e] = acc;
    }
}

// The same row pass with its INPUT STREAM decoupled from the lanes that consume it (round 5).  fp_fwd_rows_kernel's nine lanes of a
// row each load the row's 3-NN entry and skip columns themselves, one iteration ahead: 16 registers per row and stage, so one
// stage is all that fits, and a wave's iteration (14 rows, ~0.3 us of arithmetic) then waits out a memory round trip (~1 us):
// with every load and store but these switched off the kernel still took 18 us for 29 MB (scripts/time_fp1.py; 2048 waves x
// 784 B in flight = 1.6 MB: Little's law).  Here a wave fetches an iteration's 42 indices, 42 weights and 14 x QB skip quads
// with ONE element per lane (three load instructions, six registers per stage), FP_ROWS_PD iterations ahead, hands them to the
// (row, quad) lanes through a wave-private LDS region, and asks for the NEXT iteration's table rows before it computes the
// current one.  Same rows per wave, same lane mapping, same arithmetic in the same order as fp_fwd_rows_kernel: same bits,
// statistics slots included.
#ifndef SN2_FR_PD
#define SN2_FR_PD 2       // (2 / 4 / 6 stages: FP1's forward entry 41.1 / 42.3 / 44.5 us at config 2 -- more in flight is not faster here)
#endif
#ifndef SN2_FR_OCC
#define SN2_FR_OCC 2
#endif
constexpr int FP_ROWS_PD = SN2_FR_PD;
template <class F, int... I>
__device__ __forceinline__ void for_each_stage(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int CA, int CB, int CO, bool BF>
__global__ __launch_bounds__(256, SN2_FR_OCC) void fp_fwd_rows2_kernel(int R, int R_per_plot, int S_per_plot, int skip_stride,
                                                              const float* __restrict__ T, const int* __restrict__ knn_idx,
                                                              const float* __restrict__ knn_w, const float* __restrict__ skip,
                                                              const float* __restrict__ Wg, const float* __restrict__ biasg,
                                                              float* __restrict__ h, float* __restrict__ slots) {
    constexpr int CI = CA + CB, QH = (CO + 3) / 4, HS = 4 * QH, G = 64 / QH, QB = CB / 4, U = 2, RPI = G * U, PD = FP_ROWS_PD;
    static_assert(CB > 0 && CB % 4 == 0, "skip quads");
    static_assert(3 * RPI <= 64 && RPI * QB <= 64, "an iteration's inputs are one element per lane");
    static_assert(PD % 2 == 0, "the two exchange regions alternate over the stages");
    constexpr int LW = 3 * RPI + 3 * RPI + 4 * RPI * QB;                  // words of an exchange region: idx | w | skip quads
    constexpr int LWP = (LW + 3) / 4 * 4;
    static_assert((6 * RPI) % 4 == 0, "the skip quads start 16-byte aligned");
    __shared__ float s_part[8][256];
    __shared__ __attribute__((aligned(16))) float s_x[4][2][LWP];         // per wave: two regions (this iteration's, the next one's)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane % QH, g = lane / QH;
    const bool on = lane < G * QH;
    // channel PAIRS (4 q + 2 pr, + 1): the skip weights, the bias and the statistics, for the packed fp32 instructions
    f32x2 wB[2][CB], b4[2];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int o = 4 * q + 2 * pr + e;
            b4[pr][e] = o < CO ? biasg[o] : 0.f;
#pragma unroll
            for (int k = 0; k < CB; ++k) wB[pr][k][e] = o < CO ? Wg[o * CI + CA + k] : 0.f;
        }
    f32x2 ssum[2] = {{0.f, 0.f}, {0.f, 0.f}}, ssq[2] = {{0.f, 0.f}, {0.f, 0.f}};
    const int n_it_all = (R + RPI - 1) / RPI;                              // iterations: RPI consecutive rows each (3 R < 2^31)
    const RowIters ri = row_iters(n_it_all, wave);
    const int n_it = ri.it_hi, n_waves = ri.stride, it0 = ri.it0;          // this wave's iterations: it0 + k n_waves < n_it
    // ---- the input stream: one element per lane and stage
    typedef float f32x4 __attribute__((ext_vector_type(4)));              // (an array of HIP's float4 STRUCT stayed in scratch memory)
    int p_idx[PD];
    float p_w[PD];
    f32x4 p_sk[PD];
    // (every load of the steady state is UNCONDITIONAL, its address clamped into the arrays: a branch around a load makes the
    // compiler's count of outstanding memory operations unknown and it answers with s_waitcnt vmcnt(0) -- the first version of this
    // kernel drained its whole pipeline 24 times per unrolled body and ran 2 us faster than the kernel it replaces, not 10)
    const int last_e = 3 * R - 1, last_it = n_it - 1;
    const int lane_e = lane < 3 * RPI ? lane : 3 * RPI - 1, lane_s = lane < RPI * QB ? lane : RPI * QB - 1;
    auto fetch = [&](auto S, int it) {
        constexpr int s = decltype(S)::value;            // (a run-time stage index left the stage registers in scratch memory)
        const int itc = it < last_it ? it : last_it;
        const int e0 = itc * (3 * RPI) + lane_e, e = e0 < last_e ? e0 : last_e;
        p_idx[s] = knn_idx[e];
        p_w[s] = knn_w[e];
        const int rs0 = itc * RPI + lane_s / QB, rs = rs0 < R ? rs0 : R - 1;
        p_sk[s] = reinterpret_cast<const f32x4*>(skip + (size_t)rs * skip_stride)[lane_s % QB];
    };
    // stage s -> exchange region `buf`; the (row, quad) lanes read their rows' indices back and ask for the table rows
    auto hand_over = [&](auto S, int buf, int it, float4 (&ta)[U][3]) {
        constexpr int s = decltype(S)::value;
        float* xw = s_x[wave][buf];
        xw[lane_e] = __int_as_float(p_idx[s]);                             // (the surplus lanes hold a copy of the last element)
        xw[3 * RPI + lane_e] = p_w[s];
        reinterpret_cast<f32x4*>(xw + 6 * RPI)[lane_s] = p_sk[s];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rl = on ? G * u + g : 0;                              // row of the iteration
            const int row = it * RPI + rl;
            const bool valid = on && row < R && it < n_it;
            const unsigned rr = valid ? (unsigned)row : 0u;
            const int i0 = valid ? __float_as_int(xw[3 * rl + 0]) : 0, i1 = valid ? __float_as_int(xw[3 * rl + 1]) : 0,
                      i2 = valid ? __float_as_int(xw[3 * rl + 2]) : 0;
            const unsigned base = (rr / (unsigned)R_per_plot) * (unsigned)S_per_plot;
#if defined(SN2_FR_DIAG) && (SN2_FR_DIAG & 2)
            ta[u][0] = make_float4((float)(base + i0), 1.f, 2.f, 3.f);      // (diagnostic: no table gathers)
            ta[u][1] = make_float4((float)i1, 1.f, 2.f, 3.f);
            ta[u][2] = make_float4((float)i2, 1.f, 2.f, 3.f);
#else
            ta[u][0] = reinterpret_cast<const float4*>(T + (size_t)(base + i0) * HS)[q];
            ta[u][1] = reinterpret_cast<const float4*>(T + (size_t)(base + i1) * HS)[q];
            ta[u][2] = reinterpret_cast<const float4*>(T + (size_t)(base + i2) * HS)[q];
#endif
        }
    };
    auto compute = [&](int buf, int it, const float4 (&ta)[U][3]) {
        const float* xw = s_x[wave][buf];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rl = on ? G * u + g : 0;
            const int row = it * RPI + rl;
            const bool valid = on && row < R && it < n_it;
            const float w0 = xw[3 * RPI + 3 * rl + 0], w1 = xw[3 * RPI + 3 * rl + 1], w2 = xw[3 * RPI + 3 * rl + 2];
            float4 sk[QB];
#pragma unroll
            for (int b = 0; b < QB; ++b) sk[b] = reinterpret_cast<const float4*>(xw + 6 * RPI)[rl * QB + b];
            const float inv = 1.0f / ((w0 + w1) + w2);
            const float4 a = ta[u][0], b = ta[u][1], c = ta[u][2];
            const f32x2 a2[2] = {{a.x, a.y}, {a.z, a.w}}, b2[2] = {{b.x, b.y}, {b.z, b.w}}, c2[2] = {{c.x, c.y}, {c.z, c.w}};
            float v[4];
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                f32x2 acc = interp_bias2(a2[pr], b2[pr], c2[pr], w0, w1, w2, inv, b4[pr]);
#if defined(SN2_FR_DIAG) && (SN2_FR_DIAG & 4)
                acc += (f32x2){sk[0].x + sk[QB - 1].w, sk[0].x};             // (diagnostic: no skip contraction)
#else
#pragma unroll
                for (int k4 = 0; k4 < QB; ++k4) {
                    acc = __builtin_elementwise_fma(wB[pr][4 * k4 + 0], (f32x2){sk[k4].x, sk[k4].x}, acc);
                    acc = __builtin_elementwise_fma(wB[pr][4 * k4 + 1], (f32x2){sk[k4].y, sk[k4].y}, acc);
                    acc = __builtin_elementwise_fma(wB[pr][4 * k4 + 2], (f32x2){sk[k4].z, sk[k4].z}, acc);
                    acc = __builtin_elementwise_fma(wB[pr][4 * k4 + 3], (f32x2){sk[k4].w, sk[k4].w}, acc);
                }
#endif
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float x = (valid && 4 * q + 2 * pr + e < CO) ? fmaxf(acc[e], 0.f) : 0.f;
                    if constexpr (BF) x = bf16_round(x);     // the batch statistics describe the rows as they are stored
                    acc[e] = x;
                    v[2 * pr + e] = x;
                }
                ssum[pr] += acc;
                ssq[pr] = __builtin_elementwise_fma(acc, acc, ssq[pr]);
            }
#if defined(SN2_FR_DIAG) && (SN2_FR_DIAG & 1)
            if (valid && v[0] == 12345.678f) row_quad_st<BF>(h, (size_t)row, HS, q, v[0], v[1], v[2], v[3]);   // (diagnostic: no stores)
#else
            if (valid) row_quad_st<BF>(h, (size_t)row, HS, q, v[0], v[1], v[2], v[3]);
#endif
        }
    };
    // this wave's iterations, rounded up to whole blocks of PD: the surplus ones have no valid row (nothing stored, zeros added to
    // the statistics) and keep the loop body free of branches
    const int n_mine = it0 < n_it ? (n_it - it0 + n_waves - 1) / n_waves : 0;
    const int n_blocks = (n_mine + PD - 1) / PD;
    if (n_blocks > 0) {
        using S0 = std::integral_constant<int, 0>;
        using Stages = std::make_integer_sequence<int, PD>;
        for_each_stage([&](auto S) { fetch(S, it0 + decltype(S)::value * n_waves); }, Stages{});
        float4 ta_a[U][3], ta_b[U][3];
        hand_over(S0{}, 0, it0, ta_a);
        fetch(S0{}, it0 + PD * n_waves);
        // the stage of an iteration is its count modulo PD; exchange region and table-row set = its parity.  One step: hand over the
        // NEXT iteration's inputs (its table rows are then on their way), refill that stage, compute THIS iteration
        auto step = [&](auto S, int kb) {
            constexpr int st = decltype(S)::value;
            using SN = std::integral_constant<int, (st + 1) % PD>;
            const int it = it0 + (kb * PD + st) * n_waves, itn = it + n_waves;
            if constexpr (st % 2 == 0) {
                hand_over(SN{}, 1, itn, ta_b);
                fetch(SN{}, itn + PD * n_waves);
                compute(0, it, ta_a);
            } else {
                hand_over(SN{}, 0, itn, ta_a);
                fetch(SN{}, itn + PD * n_waves);
                compute(1, it, ta_b);
            }
            // (this iteration's reads of its region lie in front of the hand-over that refills it, two iterations on)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        for (int kb = 0; kb < n_blocks; ++kb) for_each_stage([&](auto S) { step(S, kb); }, Stages{});
    }
    if (!slots) return;
    // batch statistics: per-lane partials -> LDS -> one slot per workgroup ([sum(C) | sumsq(C)], as stats_to_slot)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        s_part[t][threadIdx.x] = on ? ssum[t >> 1][t & 1] : 0.f;
        s_part[4 + t][threadIdx.x] = on ? ssq[t >> 1][t & 1] : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * CO; e += 256) {
        const int which = e / CO, c = e - which * CO, cq = c >> 2, ct = c & 3;
        float acc = 0.f;
        for (int w = 0; w < 4; ++w)
            for (int gg = 0; gg < G; ++gg) acc += s_part[which * 4 + ct][w * 64 + cq + QH * gg];
        slots[(size_t)blockIdx.x * 2 * CO + e] = acc;
    }
}

// Where the d pre-activation rows of the source-side backward live (du_scratch).  With 33 or 34 channels a 36-float row
// straddles two 128-byte lines nearly always, and the source pass -- which gathers single rows from all over a 75 MB array --
// fetched both: 135 MB over the fabric for 75 MB of rows (PMC, round 3).  SPLIT layout for those widths: channels 0..31 of
// row r as ONE aligned line at main[r * 32] (bfloat16: 64 bytes), channels 32, 33 as a pair at side[r * 2] behind the
// R main rows (4.2 MB for the metric's batch: it stays in L2).  Other widths: plain rows of HS.
template <int CO>
constexpr bool dp_split() { return CO > 32 && CO <= 34; }
template <bool BF>
__device__ __forceinline__ void dp_side_st(float* __restrict__ dp, size_t R, size_t row, float a, float b) {
    if constexpr (BF) {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        bf16x2 v;
        v[0] = (__bf16)a; v[1] = (__bf16)b;
        reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(dp) + R * 32)[row] = __builtin_bit_cast(unsigned, v);
    } else {
        reinterpret_cast<float2*>(dp + R * 32)[row] = make_float2(a, b);
    }
}
template <bool BF>
__device__ __forceinline__ float2 dp_side_ld(const float* __restrict__ dp, size_t R, size_t row) {
    if constexpr (BF) {
        const unsigned u = reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned short*>(dp) + R * 32)[row];
        return make_float2(__uint_as_float(u << 16), __uint_as_float(u & 0xFFFF0000u));
    } else {
        return reinterpret_cast<const float2*>(dp + R * 32)[row];
    }
}

// rows: dp = relu'/BN backward of dy (stored, row stride HS, pad channels 0), dW_B | db
template <int CA, int CB, int CO, int NT, bool BF>
__global__ __launch_bounds__(NT) void fp_bwd_rows_kernel(int R, int skip_stride, float invR, const float* __restrict__ skip,
                                                          const float* __restrict__ gammag, const float* __restrict__ meang,
                                                          const float* __restrict__ invstdg, const float* __restrict__ dgammag,
                                                          const float* __restrict__ dbetag, const float* __restrict__ h,
                                                          const float* __restrict__ dy, float* __restrict__ dp_out,
                                                          float* __restrict__ dW, float* __restrict__ db, int rep_k,
                                                          int rep_stride, const int* __restrict__ row_perm, int R_per_plot) {
    constexpr int CI = CA + CB, QH = (CO + 3) / 4, HS = 4 * QH, G = 64 / QH, QB = CB / 4, U = 2, NV = 4 * (CB + 1);
    static_assert(CB > 0 && CB % 4 == 0, "skip quads");
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [NT / 64][HS][CB + 1]: the waves' sums
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane % QH, g = lane / QH;
    const bool on = lane < G * QH;
    float c_is[4], c_mu[4], c_gis[4], c_dbR[4], c_dg[4];
    bool ch[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int o = 4 * q + t;
        ch[t] = o < CO;
        const int oo = ch[t] ? o : 0;
        c_is[t] = invstdg[oo];
        c_mu[t] = meang[oo];
        c_gis[t] = gammag[oo] * c_is[t];
        c_dbR[t] = dbetag[oo] * invR;
        c_dg[t] = dgammag[oo];
    }
    float aW[4][CB], ab[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        ab[t] = 0.f;
#pragma unroll
        for (int k = 0; k < CB; ++k) aW[t][k] = 0.f;
    }
    // (round 5: dealing these rows XCD-aware as in the forward row pass -- XCD x then writes the d pre-activation rows of the plots whose
    // sources fp_bwd_src_chunk_kernel gathers from XCD x -- made the source pass 2 us faster (some of the rows are still in that L2)
    // and this pass 1.7 us slower: nothing in sum, scripts/time_fp1_bwd.py)
    const long n_grp = ((long)R + G - 1) / G;
    const long n_waves = (long)gridDim.x * (NT / 64);
    for (long grp0 = ((long)blockIdx.x * (NT / 64) + wave) * U; grp0 < n_grp; grp0 += n_waves * U) {
        float4 hv[U], dv[U], sk[U][QB];
        bool valid[U];
        unsigned rr[U], ro[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long row = (grp0 + u) * G + g;
            valid[u] = on && row < R;
            rr[u] = valid[u] ? (unsigned)row : 0u;
            // where the row's d pre-activation goes: its own place, or its plot's row_perm[] place (the source pass gathers
            // rows that are neighbours in space: along a space-filling curve they are neighbours in memory too)
            ro[u] = row_perm ? (rr[u] / (unsigned)R_per_plot) * (unsigned)R_per_plot + (unsigned)row_perm[rr[u]] : rr[u];
            hv[u] = row_quad_ld<BF>(h, rr[u], HS, q);
            dv[u] = row_quad_ld<BF>(dy, rr[u], HS, q);
#pragma unroll
            for (int b = 0; b < QB; ++b)
                sk[u][b] = reinterpret_cast<const float4*>(skip + (size_t)rr[u] * skip_stride)[b];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float hh[4] = {hv[u].x, hv[u].y, hv[u].z, hv[u].w}, dd[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
            float d4[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float xh = (hh[t] - c_mu[t]) * c_is[t];
                const float dh = c_gis[t] * (dd[t] - c_dbR[t] - xh * c_dg[t] * invR);
                d4[t] = (valid[u] && ch[t] && hh[t] > 0.f) ? dh : 0.f;
                ab[t] += d4[t];
#pragma unroll
                for (int b = 0; b < QB; ++b) {
                    aW[t][4 * b + 0] = fmaf(d4[t], sk[u][b].x, aW[t][4 * b + 0]);
                    aW[t][4 * b + 1] = fmaf(d4[t], sk[u][b].y, aW[t][4 * b + 1]);
                    aW[t][4 * b + 2] = fmaf(d4[t], sk[u][b].z, aW[t][4 * b + 2]);
                    aW[t][4 * b + 3] = fmaf(d4[t], sk[u][b].w, aW[t][4 * b + 3]);
                }
            }
            if constexpr (dp_split<CO>()) {
                if (valid[u] && q < 8) row_quad_st<BF>(dp_out, ro[u], 32, q, d4[0], d4[1], d4[2], d4[3]);
                if (valid[u] && q == 8) dp_side_st<BF>(dp_out, (size_t)R, ro[u], d4[0], d4[1]);
            } else {
                if (valid[u]) row_quad_st<BF>(dp_out, ro[u], HS, q, d4[0], d4[1], d4[2], d4[3]);
            }
        }
    }
    // per-lane partials -> the wave's sums over its G row groups by lane shuffles (lanes q, q + QH, q + 2 QH, ... hold the same
    // channel quad) -> a [wave][channel][CB + 1] image in LDS (10 KB; all lanes' partials side by side were 74 KB per
    // workgroup: one workgroup per CU wherever an FPS workgroup holds its share of the LDS) -> one sum per element and
    // workgroup -> global atomics
    constexpr int NWV = NT / 64;
    static_assert(G <= 8, "three shuffle steps add up to eight row groups");
    float* red = smem;                                   // [NWV][HS][CB + 1]
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int k = 0; k <= CB; ++k) {
            float v = on ? (k < CB ? aW[t][k] : ab[t]) : 0.f;
#pragma unroll
            for (int step = 4; step > 0; step >>= 1) {
                const float o2 = __shfl(v, (lane + QH * step) & 63);
                if (g < step && g + step < G) v += o2;
            }
            if (g == 0 && on) red[(wave * HS + 4 * q + t) * (CB + 1) + k] = v;
        }
    __syncthreads();
    for (int e = threadIdx.x; e < CO * (CB + 1); e += NT) {
        const int o = e / (CB + 1), k = e - o * (CB + 1);
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) acc += red[(w * HS + o) * (CB + 1) + k];
        const int img = sn2_grad_image(rep_k, rep_stride);
        if (acc != 0.f) SN2_FLUSH_ADD(k < CB ? &dW[img + o * CI + CA + k] : &db[img + o], acc);
    }
    (void)NV;
}

// Source pass of the source-side backward: G[s] = sum over the inverted list of s of w * d pre-activation row.
// The unit of work is a CHUNK of a list (inv_order_kernel's chunk table: at most INV_CHUNK entries = one load of row numbers,
// one of weights, STEPS row gathers in flight per lane), and a wave owns L CONSECUTIVE slots of the chunk table, wave w the
// slots [w L, (w + 1) L): it loads their headers with one instruction, and the row numbers of the next chunk travel together
// with the gathers of this one -- one memory round trip per chunk instead of three, the same short chain for
// every wave whatever the lists look like (a wave per SOURCE waited for the few sources whose lists are 13 chunks long:
// C2's lists have a median of 25 entries and a tenth of them 500-800).  Sums run on across the chunks of one source and are
// written out as a partial row Gpart[slot] at the source's last chunk and at the wave's last slot;
// fp_bwd_src_merge_dw_kernel adds a source's partial rows in slot order (and applies G to dsrc and dW_A).
template <int CA, int CB, int CO, bool BF>
__global__ __launch_bounds__(256) void fp_bwd_src_chunk_kernel(int n_chunks, int L, int R_total, int R_per_plot, int S,
                                                               const int4* __restrict__ chunks, const int* __restrict__ inv_row,
                                                               const float* __restrict__ inv_w, const float* __restrict__ dp,
                                                               float* __restrict__ Gpart) {
    constexpr bool SPLIT = dp_split<CO>();
    constexpr int QH = (CO + 3) / 4, HS = 4 * QH, QM = SPLIT ? 8 : QH, RS = SPLIT ? 32 : HS, G = 64 / QM, STEPS = 64 / G;
    constexpr unsigned EB = BF ? 2u : 4u, ROWB = RS * EB;                   // bytes per element / per row
    static_assert(G * STEPS >= INV_CHUNK, "a chunk is one round of gathers");
    static_assert(STEPS % 2 == 0, "two halves");
    __shared__ __attribute__((aligned(16))) float s_part[SPLIT ? 1 : 4][G][HS];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (uniform: SGPRs)
    const int q = lane % QM, g = lane / QM;
    const bool on = lane < G * QM;
    const unsigned qoff = (unsigned)q * 4u * EB;
    // workgroups go to the XCDs round-robin; the waves of one XCD own consecutive stretches of the table (plots one after the
    // other, each along its sources' Morton order): neighbouring sources share target rows, and so an L2 (gridDim.x % 8 == 0)
    const int xcd = blockIdx.x & 7, wg_x = blockIdx.x >> 3, n_wg_x = gridDim.x >> 3;
    const long r0 = ((long)(xcd * n_wg_x + wg_x) * 4 + wave) * L;
    if (r0 >= n_chunks) return;                              // (no workgroup barriers in this kernel)
    const int nk = (n_chunks - r0) < L ? (int)(n_chunks - r0) : L;
    const int4 it = lane < nk ? chunks[r0 + lane] : make_int4(0, 0, 0, 0);       // L <= 64
    // (the headers are awaited HERE, once: left to the loop, the wait at its top also sits out every partial row's store)
    // padding only: NO slot of the wave holds a chunk (first and last slot empty is not enough: with plots of fewer than 64
    // chunks a wave's range may start in one plot's padding, cover the next plot's chunks and end in that plot's padding)
    if (__ballot(lane < nk && it.z != 0) == 0ull) return;
    int id = 0, pb = 0, m = 0, rj = 0;                               // round k = -1 only fetches the entries of chunk 0
    float wj = 0.f;
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, sa0 = 0.f, sa1 = 0.f;
    // one round of gathers: NS steps of G entries (entries past the chunk's end: row 0 with weight 0)
    auto gather = [&](auto ns, const char* __restrict__ base) {
        constexpr int NS = decltype(ns)::value;
        float4 v[NS];
        float wv[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int e = G * j + g;
            const unsigned r = (unsigned)__shfl(rj, e);
            wv[j] = __shfl(wj, e);
            if constexpr (!SPLIT) wv[j] = on ? wv[j] : 0.f;
            const unsigned off = r * ROWB + qoff;
            if constexpr (BF) {
                const uint2 u = *reinterpret_cast<const uint2*>(base + off);
                v[j] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                                   __uint_as_float(u.y & 0xFFFF0000u));
            } else {
                v[j] = *reinterpret_cast<const float4*>(base + off);
            }
        }
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            acc[0] = fmaf(wv[j], v[j].x, acc[0]);
            acc[1] = fmaf(wv[j], v[j].y, acc[1]);
            acc[2] = fmaf(wv[j], v[j].z, acc[2]);
            acc[3] = fmaf(wv[j], v[j].w, acc[3]);
        }
    };
    for (int k = -1; k < nk; ++k) {
        int id_n = 0, st_n = 0, m_n = 0, pb_n = 0;
        if (k + 1 < nk) {
            id_n = __builtin_amdgcn_readlane(it.x, k + 1);
            pb_n = __builtin_amdgcn_readlane(it.w, k + 1);
            st_n = __builtin_amdgcn_readlane(it.y, k + 1);
            m_n = __builtin_amdgcn_readlane(it.z, k + 1);
        }
        // the next chunk's entries first: in flight together with this chunk's gathers (loads return in order)
        const int rj_n = lane < m_n ? inv_row[st_n + lane] : 0;          // entries past the end: row 0 with weight 0
        const float wj_n = lane < m_n ? inv_w[st_n + lane] : 0.f;
        if (m > 0) {
            const size_t row0 = (size_t)pb * R_per_plot;
            const char* base = reinterpret_cast<const char*>(dp) + row0 * ROWB;
            float2 sv = make_float2(0.f, 0.f);
            if constexpr (SPLIT) sv = dp_side_ld<BF>(dp, (size_t)R_total, row0 + rj);   // this lane's own entry
            if (m > G * STEPS / 2) gather(std::integral_constant<int, STEPS>{}, base);
            else gather(std::integral_constant<int, STEPS / 2>{}, base);
            sa0 = fmaf(wj, sv.x, sa0);
            sa1 = fmaf(wj, sv.y, sa1);
            if (m_n == 0 || id_n != id) {                    // the source's last chunk, or this wave's last: a partial row
                float* out = Gpart + (size_t)(r0 + k) * HS;
                if constexpr (SPLIT) {
                    // eight lanes per entry: the entries of a step differ in lane bits 3..5
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        acc[t] += __int_as_float(SN2_DPP(__float_as_int(acc[t]), 0x128, 0xF));     // row_ror:8 = lane ^ 8
                        acc[t] += __shfl_xor(acc[t], 16);
                        acc[t] += __shfl_xor(acc[t], 32);
                    }
                    const float s0 = wave_sum(sa0), s1 = wave_sum(sa1);
                    if (lane < 8) reinterpret_cast<float4*>(out)[lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
                    if (lane == 8) reinterpret_cast<float4*>(out)[8] = make_float4(s0, s1, 0.f, 0.f);
                } else {
                    if (on) *reinterpret_cast<float4*>(&s_part[wave][g][4 * q]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (lane < HS) {
                        float gk = 0.f;
#pragma unroll
                        for (int gg = 0; gg < G; ++gg) gk += s_part[wave][gg][lane];
                        out[lane] = gk;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                acc[0] = acc[1] = acc[2] = acc[3] = 0.f;
                sa0 = sa1 = 0.f;
            }
        }
        rj = rj_n, wj = wj_n, id = id_n, pb = pb_n, m = m_n;
    }
}

// What G[s] is for, 64 sources per workgroup in the order of the item table, one per lane of EVERY wave: the four
// waves add up the sources' partial rows (in slot order, a quarter of the channels each) and then each does a quarter of the
// rest: dsrc[s][k] += sum_o G[s][o] W_A[o][k] for its quarter of the channels k (the weights through scalar loads), and
// dW_A[:, 16 w .. 16 w + 15] += sum_s G[s]^T (a o src[s] + c) for its 16 columns (rows = MFMA K).  (One wave doing all of it
// for its 64 sources was one long chain per CU: 18 us.)
template <int CA, int CB, int CO>
__global__ __launch_bounds__(256) void fp_bwd_src_merge_dw_kernel(int n_src, int S, int CM, int src_stride, int dsrc_stride,
                                                                  const int4* __restrict__ items, const float* __restrict__ src,
                                                                  const float* __restrict__ src_a, const float* __restrict__ src_c,
                                                                  const float* __restrict__ Gpart, const float* __restrict__ Wg,
                                                                  float* __restrict__ dsrc, float* __restrict__ dW, int rep_k,
                                                                  int rep_stride, int L) {
    constexpr int CI = CA + CB, QH = (CO + 3) / 4, HS = 4 * QH, TK = (CA + 15) / 16, KQ = (CA + 3) / 4;
    static_assert(TK <= 4, "a wave per 16 columns of dW_A");
    static_assert(KQ <= CA, "a window of KQ columns fits a row");
    using Acc = OuterAcc<CO, 16, 32>;
    static_assert(CO * 16 <= Acc::LDS_FLOATS, "a wave's image fits its staging region");
    __shared__ __attribute__((aligned(16))) float smem[4 * Acc::LDS_FLOATS];
    __shared__ float s_p[HS * 64];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* lds = smem + w * Acc::LDS_FLOATS;
    const int pp = blockIdx.x * 64 + lane;
    const bool valid = pp < n_src;
    const int4 item = valid ? items[pp] : make_int4(0, 0, 0, 0);
    const size_t ss = (size_t)item.x;
    // what comes from memory at the END of the chain is asked for now: the old values of this wave's quarter of dsrc[s] ..
    float* dr = dsrc + ss * dsrc_stride;
    float d_old[KQ];
    const int k0 = w * KQ + KQ <= CA ? w * KQ : CA - KQ;     // (the last wave's window of KQ columns is pulled back inside the row)
#pragma unroll
    for (int i = 0; i < KQ; ++i) d_old[i] = valid ? dr[k0 + i] : 0.f;
    // .. and this wave's columns of (a o src + c)
    float x[16];
    if (w < TK) {
        const float4* sr = reinterpret_cast<const float4*>(src + ss * src_stride);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (4 * (4 * w + q4) < CA) a = sr[4 * w + q4];
            x[4 * q4] = a.x, x[4 * q4 + 1] = a.y, x[4 * q4 + 2] = a.z, x[4 * q4 + 3] = a.w;
        }
    }
    // the source's chunks are the slots [f, f + nch) of the chunk table; its partial rows: at its last slot and at every
    // slot that is the last of a wave of fp_bwd_src_chunk_kernel (slot % L == L - 1).  Wave w adds up the quads w, w + 4, ..
    // of the rows, four rows in flight, and the waves swap their sums through LDS.
    const int nch = (item.z + INV_CHUNK - 1) / INV_CHUNK;
    const int f = (item.x / S) * CM + item.w, last = f + nch - 1;        // (slots: B * CM < 2^31, checked by the host)
    constexpr int NQ = (QH + 3) / 4;
    float4 pq[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) pq[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = f; __builtin_amdgcn_ballot_w64(r <= last) != 0;) {
        float4 a[4][NQ];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool more = r <= last;
            const int e = more ? min(last, (int)((unsigned)r / (unsigned)L) * L + L - 1) : 0;
            const float4* gr = reinterpret_cast<const float4*>(Gpart + (size_t)e * HS);
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                a[j][i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (more && w + 4 * i < QH) a[j][i] = gr[w + 4 * i];
            }
            if (more) r = e + 1;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < NQ; ++i)
                pq[i].x += a[j][i].x, pq[i].y += a[j][i].y, pq[i].z += a[j][i].z, pq[i].w += a[j][i].w;
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i)
        if (w + 4 * i < QH) {
            const int o = 4 * (w + 4 * i);
            s_p[(o + 0) * 64 + lane] = pq[i].x, s_p[(o + 1) * 64 + lane] = pq[i].y;
            s_p[(o + 2) * 64 + lane] = pq[i].z, s_p[(o + 3) * 64 + lane] = pq[i].w;
        }
    __syncthreads();
    float p[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) p[o] = s_p[o * 64 + lane];
    if (w < TK) {
        const cfp sa = opaque(as_const(src_a)), sc = opaque(as_const(src_c));
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int k = 16 * w + i;
            if (k >= CA || !valid) x[i] = 0.f;
            else if (src_a) x[i] = fmaf(sa[k], x[i], sc[k]);
        }
        if (!valid) {
#pragma unroll
            for (int o = 0; o < CO; ++o) p[o] = 0.f;
        }
        Acc acc;
        acc.init(lds);
        acc.add(lds, p, x);
        acc.store_slab(lds);                                 // slab[o * 16 + i]
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int img = sn2_grad_image(rep_k, rep_stride);
        for (int i = lane; i < CO * 16; i += 64) {
            const int o = i >> 4, k = 16 * w + (i & 15);
            const float v = lds[i];
            if (k < CA && v != 0.f) SN2_FLUSH_ADD(&dW[img + o * CI + k], v);
        }
    }
    // dsrc last: its arithmetic runs while the float atomics above are on their way
    if (valid) {
        // (KQ consecutive weights per o: wide scalar loads; the last wave's window is pulled back inside the row, its
        // first columns computed twice and not stored)
        const cfp W = opaque(as_const(Wg)) + k0;
        float d[KQ];
#pragma unroll
        for (int i = 0; i < KQ; ++i) d[i] = 0.f;
#pragma unroll 2
        for (int o = 0; o < CO; ++o) {                       // (not unrolled in full: 306 weights do not fit the SGPRs)
            const float po = s_p[o * 64 + lane];
#pragma unroll
            for (int i = 0; i < KQ; ++i) d[i] = fmaf(po, W[o * CI + i], d[i]);
        }
        const int sh = w * KQ - k0;                          // columns of the window that belong to the wave before
#pragma unroll
        for (int i = 0; i < KQ; ++i)
            if (i >= sh) dr[k0 + i] = d_old[i] + d[i];
    }
}

// The four waves of a 64-row workgroup build the rows' inputs [u | 1] together in LDS, s_q[64][QS]; wave g builds the rows
// 16 g .. 16 g + 15.  A row's interpolated part is CA / 4 float4 quads: that many consecutive lanes share a row, so one load
// instruction covers 64 / (CA / 4) whole source rows -- with one row per lane every instruction touched 64 different cache
// lines and the four waves queued behind the CU's one address unit (17 000 clocks of a 46 000-clock kernel).  The same
// arithmetic, element by element, as build_input.  Columns past CA + CB are left alone.
template <int CA, int CB, bool KNN>
__device__ __forceinline__ void stage_inputs(float* __restrict__ s_q, int QS, int g, int lane, long r0, int R, int R_per_plot,
                                             int S_per_plot, const float* __restrict__ src, int src_stride,
                                             const float* __restrict__ src_a, const float* __restrict__ src_c,
                                             const int* __restrict__ knn_idx, const float* __restrict__ knn_w,
                                             const float* __restrict__ skip, int skip_stride) {
    constexpr int CI = CA + CB, QA = (CA + 3) / 4;
    constexpr int LPR = QA <= 8 ? 8 : (QA <= 16 ? 16 : (QA <= 32 ? 32 : 64));     // lanes per row (a power of two >= QA)
    static_assert(QA <= 64, "at most 256 interpolated channels");
    constexpr int RPI = 64 / LPR;                     // rows per load instruction
    const int q = lane & (LPR - 1);
    const bool qon = q < QA;
    const int qc = qon ? q : 0;
    // the skip columns of the few-channel form (CB no multiple of four: the positions of the global SA block) are asked for HERE,
    // together with the rows of the interpolated part: behind them they were a memory round trip of their own
    constexpr bool SKIP_SCALAR = CB > 0 && !(CB % 4 == 0 && ((CB / 4) & (CB / 4 - 1)) == 0 && CB <= 64);
    float skr[SKIP_SCALAR ? CB : 1];
    if constexpr (SKIP_SCALAR) {
        const long r = r0 + 16 * g + (lane & 15);
        const size_t rr = r < R ? (size_t)r : (size_t)(R - 1);
#pragma unroll
        for (int k = 0; k < CB; ++k) skr[k] = skip[rr * skip_stride + k];
    }
    float a4[4] = {1.f, 1.f, 1.f, 1.f}, c4[4] = {0.f, 0.f, 0.f, 0.f};
    if (src_a) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (4 * qc + t < CA) a4[t] = src_a[4 * qc + t], c4[t] = src_c[4 * qc + t];
    }
#pragma unroll
    for (int st = 0; st < 16 / RPI; ++st) {
        const int row = 16 * g + st * RPI + lane / LPR;
        const long r = r0 + row;
        const size_t rr = r < R ? (size_t)r : (size_t)(R - 1);
        float v[4];
        if constexpr (KNN) {
            const size_t base = (rr / R_per_plot) * S_per_plot;
            const int i0 = knn_idx[rr * 3 + 0], i1 = knn_idx[rr * 3 + 1], i2 = knn_idx[rr * 3 + 2];
            const float w0 = knn_w[rr * 3 + 0], w1 = knn_w[rr * 3 + 1], w2 = knn_w[rr * 3 + 2];
            const float inv = 1.0f / ((w0 + w1) + w2);
            const float4 a = reinterpret_cast<const float4*>(src + (base + i0) * src_stride)[qc];
            const float4 b = reinterpret_cast<const float4*>(src + (base + i1) * src_stride)[qc];
            const float4 c = reinterpret_cast<const float4*>(src + (base + i2) * src_stride)[qc];
            v[0] = ((a.x * w0 + b.x * w1) + c.x * w2) * inv, v[1] = ((a.y * w0 + b.y * w1) + c.y * w2) * inv;
            v[2] = ((a.z * w0 + b.z * w1) + c.z * w2) * inv, v[3] = ((a.w * w0 + b.w * w1) + c.w * w2) * inv;
        } else {
            const float4 a = reinterpret_cast<const float4*>(src + rr * src_stride)[qc];
            v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w;
        }
        if (src_a) {
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = fmaf(a4[t], v[t], c4[t]);
        }
        if (qon) {
            if (CA % 4 == 0 || q < QA - 1) {
                *reinterpret_cast<float4*>(&s_q[row * QS + 4 * q]) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (4 * q + t < CA) s_q[row * QS + 4 * q + t] = v[t];
            }
        }
    }
    if constexpr (CB > 0 && CB % 4 == 0 && ((CB / 4) & (CB / 4 - 1)) == 0 && CB <= 64) {
        constexpr int QB = CB / 4, RPB = 64 / QB;     // the skip part the same way
#pragma unroll
        for (int st = 0; st < (16 + RPB - 1) / RPB; ++st) {
            const int rl = st * RPB + lane / QB;      // 0..15 within the wave's rows
            const int row = 16 * g + rl;
            const long r = r0 + row;
            const size_t rr = r < R ? (size_t)r : (size_t)(R - 1);
            if (rl < 16) {
                const float4 a = reinterpret_cast<const float4*>(skip + rr * skip_stride)[lane & (QB - 1)];
                float* d = &s_q[row * QS + CA + 4 * (lane & (QB - 1))];
                d[0] = a.x, d[1] = a.y, d[2] = a.z, d[3] = a.w;
            }
        }
    } else if constexpr (CB > 0) {
        if (lane < 16) {
            const int row = 16 * g + lane;
#pragma unroll
            for (int k = 0; k < CB; ++k) s_q[row * QS + CA + k] = skr[k];
        }
    }
    if (lane < 16) s_q[(16 * g + lane) * QS + CI] = 1.0f;       // the bias column
}

// ---------------------------------------------------------------------------------------------- small layers
// Layers with few rows (SA3, FP3, FP2: 4k-16k rows, up to 96 -> 64 channels) gain nothing from one long FMA stream per
// lane: 4096 rows are only 64 waves on a 1024-SIMD chip and each wave would issue >6000 dependent FMAs (the first
// version ran 0.3-0.7 ms per kernel at ~1 % of the chip).  Here one workgroup = 64 rows x 4 waves and wave g owns the
// output channels [g*COG, (g+1)*COG): 4x the waves, 4x shorter streams, the same weights-in-SGPR inner loops.
template <int CA, int CB, int CO, bool KNN, bool BF16>
__global__ __launch_bounds__(256) void fp_fwd_split_kernel(int R, int R_per_plot, int S_per_plot, int src_stride,
                                                           int skip_stride, int h_stride, const float* __restrict__ src,
                                                           const float* __restrict__ src_a, const float* __restrict__ src_c,
                                                           const int* __restrict__ knn_idx, const float* __restrict__ knn_w,
                                                           const float* __restrict__ skip, const float* __restrict__ W,
                                                           const float* __restrict__ bias, float* __restrict__ h,
                                                           float* __restrict__ slots) {
    // 64 rows per workgroup.  The four waves first build the rows' inputs [u | 1] together in LDS (wave g: the channel
    // quads q = g mod 4 of the interpolated part and of the skip part), then wave g computes the 16 output channels
    // [16 g, 16 g + 16) of all 64 rows on the matrix cores: A[row][k] from the staged inputs, B[k][o] = W^T | bias in
    // registers ((CI + 1)/4 values per lane).  As per-lane FMA chains with scalar-loaded weights (16 x 96 per lane for FP3,
    // every wave rebuilding the whole input) these layers took 15-30 us each for 4-16k rows.
    constexpr int CI = CA + CB, CK = CI + 1, KB = (CK + 3) / 4, QA = (CA + 3) / 4, NG = (CO + 15) / 16;
    constexpr int QS = OuterAcc<16, CK>::QS;
    static_assert(NG <= 4 && 4 * KB <= QS, "fp_fwd_split_kernel shapes");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_q = smem;                       // [64][QS]
    float* s_red = smem + 64 * QS;           // [2 * 16 * NG]
    const int lane = threadIdx.x & 63;
    const int g = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long r = (long)blockIdx.x * 64 + lane;
    const bool valid = r < R;
    const size_t rr = valid ? (size_t)r : (size_t)(R - 1);
    stage_inputs<CA, CB, KNN>(s_q, QS, g, lane, (long)blockIdx.x * 64, R, R_per_plot, S_per_plot, src, src_stride, src_a, src_c,
                              knn_idx, knn_w, skip, skip_stride);
    if (g == 2) {                            // K padding of the contraction: finite (0 x garbage would be NaN)
#pragma unroll
        for (int k = CK; k < 4 * KB; ++k) s_q[lane * QS + k] = 0.f;
    }
    // B operand: lane (qq, cc) keeps [W | bias][o = 16 g + cc][k = 4 kb + qq]
    const int qq = lane >> 4, cc = lane & 15;
    const int o = 16 * g + cc;
    float Wb[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const int k = 4 * kb + qq;
        // bf16 operands: the bias stays out of the contraction (it would be rounded) and starts the accumulator instead
        Wb[kb] = (g < NG && o < CO) ? (k < CI ? W[o * CI + k] : ((k == CI && !BF16) ? bias[o] : 0.f)) : 0.f;
    }
    const float bias_o = (BF16 && g < NG && o < CO) ? bias[o] : 0.f;
    __syncthreads();
    float ssum = 0.f, ssq = 0.f;
    if (g < NG) {
        f32x4 D[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) D[t] = f32x4{bias_o, bias_o, bias_o, bias_o};
        if constexpr (!BF16) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    D[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(s_q[(16 * t + cc) * QS + 4 * kb + qq], Wb[kb], D[t], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                D[t] = contract<true, KB>(D[t], [&](int kb) { return s_q[(16 * t + cc) * QS + 4 * kb + qq]; },
                                          [&](int kb) { return Wb[kb]; });
        }
        // D[t][j]: row 16 t + 4 qq + j, channel o
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long row = (long)blockIdx.x * 64 + 16 * t + 4 * qq + j;
                const float v = (row < R && o < CO) ? fmaxf(D[t][j], 0.f) : 0.f;
                ssum += v;
                ssq = fmaf(v, v, ssq);
                if (row < R && o < h_stride) h[(size_t)row * h_stride + o] = v;      // pad channels: 0
            }
        // the four row groups (qq) of a channel
        ssum += __shfl_xor(ssum, 16);
        ssq += __shfl_xor(ssq, 16);
        ssum += __shfl_xor(ssum, 32);
        ssq += __shfl_xor(ssq, 32);
        if (qq == 0) {
            s_red[o] = ssum;
            s_red[16 * NG + o] = ssq;
        }
    }
    if (slots) {
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * CO; i += 256)
            slots[(size_t)blockIdx.x * 2 * CO + i] = i < CO ? s_red[i] : s_red[16 * NG + (i - CO)];
    }
}

// ---------------------------------------------------------------------------------------------- the global level, forward
// SA3 (MLP[35,64] on cat[x2, pos2]) -> its BatchNorm -> the plot's max -> FP3 (MLP[96,64] on cat[plot feature, x2]) -> its
// BatchNorm, TRAINING mode, in ONE launch (round 4).  As separate launches these are fp_fwd_split_kernel<32,3,64>, bn_finalize,
// plot_max, fp_fwd_split_kernel<64,32,64>, bn_finalize: 30 us for 4096 rows of 64 channels, each launch a dependent round trip.
// Here one workgroup of 16 waves owns a PLOT: four groups of four waves run the 64-row blocks of the split kernel side by
// side (same staging, same tiles, same per-block statistics), the max of the plot and the plot feature never leave the
// workgroup, and only the two BatchNorm statistics cross workgroups -- through 8-byte {tag, value} granules as the
// multi-workgroup FPS exchanges its records (agent-scope relaxed stores and loads, the data is its own flag):
//   every group publishes its 2 x 64 sums; after SA3 every workgroup sweeps all B x 4 x 128 granules until the tags match and
//   finalises the statistics ITSELF (fp64, fixed order: the same a, c in every workgroup); after FP3 only workgroup 0 waits,
//   finalises and writes the block's a, c, mean, invstd and running statistics (workgroup 0 does that for SA3 too).
// The tag is (launch epoch, phase); the epoch lives in ctl[0] and is advanced by workgroup 0 at the very end (every workgroup
// has read it before anyone can pass the first exchange).  Residency: B workgroups of 1024 threads; a wait is bounded
// (spin_limit sweeps), a workgroup whose wait runs out counts it in ctl[1] and carries on with whatever it has (wrong
// statistics, no hang): the host reads ctl[1] where it synchronises anyway (hip_ops.global_level_gave_up) and raises.
#ifdef SN2_GL_STAMPS
// diagnostic build only (never shipped): phase stamps of thread 0 of workgroup 0 of global_level_fwd_kernel
__device__ unsigned long long g_gl_dbg[16];
extern "C" int sn2_debug_gl_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gl_dbg), sizeof(g_gl_dbg));
}
#define GSTAMP(i)                                                                                   \
    if (blockIdx.x == 0 && threadIdx.x == 0) {                                                      \
        unsigned long long t_;                                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                   \
        g_gl_dbg[i] = t_;                                                                           \
    }
#else
#define GSTAMP(i)
#endif
typedef unsigned long long gl_u64;
constexpr int GL_GROUPS = 4, GL_QS = OuterAcc<16, 36>::QS;       // 48: the staged rows [x2 (32) | pos2 (3) | 1] of SA3; FP3 reads the first 32
constexpr int GL_W3 = 64 * 35 + 64, GL_WF = 64 * 96 + 64;         // the two layers' [W | bias], copied into LDS once per workgroup
constexpr int GL_FIXED_FLOATS = GL_GROUPS * 64 * GL_QS + GL_GROUPS * 128 + 128 + 64 + 2 * 1024 + 2 * 8 * 128 + GL_W3 + GL_WF;
constexpr int GL_MAX_PLOTS = 28;                                  // + B * 4 * 128 floats of collected granules: 155 KB at 28 plots
static_assert((GL_FIXED_FLOATS + GL_MAX_PLOTS * GL_GROUPS * 128) * 4 <= 160 * 1024, "LDS");

// one 64-row block of a split layer on a group of four waves: fp_fwd_split_kernel's tiles (wave g: output channels
// [16 g, 16 g + 16)), accumulators started at `init`, rows past R_lim masked; adds the block's statistics of channel
// 16 g + cc to (ssum, ssq); V: the block's outputs (row 16 t + 4 qq + j of the block, channel 16 g + cc) stay with the caller
template <int KB>
__device__ __forceinline__ void gl_block_tiles(const float* s_q, int QS, int lane, int g, long row0, long R_lim,
                                               const float (&Wb)[KB], float init, float* __restrict__ h, int h_stride,
                                               float& ssum, float& ssq, f32x4 (&V)[4]) {
    const int qq = lane >> 4, cc = lane & 15, o = 16 * g + cc;
#pragma unroll
    for (int t = 0; t < 4; ++t) V[t] = f32x4{init, init, init, init};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            V[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(s_q[(16 * t + cc) * QS + 4 * kb + qq], Wb[kb], V[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long row = row0 + 16 * t + 4 * qq + j;
            const float v = row < R_lim ? fmaxf(V[t][j], 0.f) : 0.f;
            V[t][j] = v;
            ssum += v;
            ssq = fmaf(v, v, ssq);
            if (row < R_lim) h[(size_t)row * h_stride + o] = v;
        }
}

struct GlLayer {
    const float *W, *bias, *gamma, *beta;
    float *running_mean, *running_var, *a, *c, *mean, *invstd;
    long long* nbt;
    float* h;
};
struct GlArgs {
    int B, M2;
    const float* x2;        // (B*M2, 32)
    const float* pos2;      // (B*M2, 4)
    const int* knn_idx;     // FP3's table (B*M2, 3): every entry names the plot's one source
    const float* knn_w;
    float* x3;              // (B, 64)
    int* arg3;
    GlLayer sa3, fp3;
    gl_u64* xchg;           // [2 phases][B * 4 groups][128]
    unsigned* ctl;          // [0] epoch of the last finished launch, [1] workgroups that gave up (sticky), [2] ... that a repair
                            // launch has handled, [3] running statistics workgroup 0 updated in the last launch (bit 0 SA3, 1 FP3)
    unsigned spin_limit;
};

// all granules of a phase -> s_x (floats), every thread its share, eight loads in flight, swept until every tag matches (or
// the limit runs out, or a publisher says that it gave up: the POISON tag = tag with the top bit flipped)
constexpr unsigned GL_POISON = 0x80000000u;
__device__ __forceinline__ bool gl_collect(const gl_u64* gx, int n, unsigned tag, float* s_x, unsigned spin_limit) {
    bool ok = true;
    for (int i0 = threadIdx.x; i0 < n; i0 += 8 * 1024) {
        gl_u64 v[8];
        unsigned spins = 0;
        bool all, poisoned;
        do {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 1024;
                v[u] = __hip_atomic_load(gx + (i < n ? i : i0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            all = true, poisoned = false;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                all = all && (unsigned)(v[u] >> 32) == tag;
                poisoned = poisoned || (unsigned)(v[u] >> 32) == (tag ^ GL_POISON);
            }
            if (!all && !poisoned) __builtin_amdgcn_s_sleep(2);
        } while (!all && !poisoned && ++spins < spin_limit);
        if (!all) ok = false;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + u * 1024 < n) s_x[i0 + u * 1024] = __uint_as_float((unsigned)v[u]);
    }
    return ok;
}

// REPAIR = false: the launch proper, one workgroup per plot.  A workgroup whose wait for its peers' statistics runs out (HIP does
// not promise that the B workgroups of a launch are resident together) or that finds a peer's POISON gives up: it counts itself
// in ctl[1], publishes POISON instead of its FP3 sums and leaves; workgroup 0 updates a layer's running statistics only when
// its own collection of that layer's sums was complete, and says which it updated in ctl[3].
// REPAIR = true: run by the workgroup that leaves the launch LAST (global_level_fwd_kernel's exit protocol) when ctl[1] != ctl[2]
// -- a wait gave up since the last repair: it computes the WHOLE level alone, plot after plot, with the same tiles, the same
// per-group sums published to and collected from the same exchange area and the same fixed-order finalisation -- the bits of
// an undisturbed launch --, applies the running-statistics updates workgroup 0 did not (ctl[3]), and moves the epoch past
// every tag a late workgroup of the failed launch may have written.
template <bool REPAIR>
__device__ __forceinline__ void gl_level_body(const GlArgs& A, float* gl_smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 8), g = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);
    float* s_q = gl_smem + grp * 64 * GL_QS;                     // the group's staged rows

    float* s_red = gl_smem + GL_GROUPS * 64 * GL_QS;             // [4 groups][sum(64) | sumsq(64)]
    float* s_ac = s_red + GL_GROUPS * 128;                       // a[64] | c[64] of SA3's BatchNorm
    float* s_x3 = s_ac + 128;
    float* s_mv = s_x3 + 64;                                     // [1024] the max's partial values ...
    int* s_mi = reinterpret_cast<int*>(s_mv + 1024);             // ... and rows
    double* s_d = reinterpret_cast<double*>(s_mi + 1024);        // [8][128] partial sums of a finalisation
    float* s_w3 = reinterpret_cast<float*>(s_d + 8 * 128);       // SA3's [W (64 x 35) | bias (64)]
    float* s_wf = s_w3 + GL_W3;                                  // FP3's [W (64 x 96) | bias (64)]
    float* s_x = s_wf + GL_WF;                                   // [B * 4 * 128] the collected granules of an exchange
    __shared__ unsigned s_epoch;
    __shared__ int s_fail;
    const int B = A.B, M2 = A.M2;
    const int b_lo = REPAIR ? 0 : (int)blockIdx.x, b_hi = REPAIR ? B : b_lo + 1;       // the plots of this workgroup
    const bool lead = REPAIR || blockIdx.x == 0;                                       // writes the shared results
    // Both layers' weights come in once per workgroup, coalesced, and the lanes take their tile operands from LDS: sixteen
    // waves each fetching their own (output, k) elements straight from memory were ~400 cache lines per wave through the CU's
    // one address unit -- half of the kernel's first phase.  Round 5: every thread's seven loads are ISSUED here, unconditional
    // (clamped indices), in front of the epoch read and its barrier, and land in LDS behind it: as copy loops `s_w[i] = W[i]`
    // they were five memory round trips one after the other (each iteration's load waited for by its own store), cold, at the
    // head of a kernel that is one latency chain (scripts/isa_scan.py).
    float w3r[3], b3r, bfr;
    f32x4 wfr[2];                                    // (the native vector type: an array of HIP's float4 struct stays in scratch memory)
#pragma unroll
    for (int j = 0; j < 3; ++j) w3r[j] = A.sa3.W[min(tid + 1024 * j, 64 * 35 - 1)];
#pragma unroll
    for (int j = 0; j < 2; ++j) wfr[j] = reinterpret_cast<const f32x4*>(A.fp3.W)[min(tid + 1024 * j, 64 * 96 / 4 - 1)];
    b3r = A.sa3.bias[tid & 63], bfr = A.fp3.bias[tid & 63];
    unsigned epoch_now = 0;
    if (tid == 0) epoch_now = __hip_atomic_load(&A.ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int j = 0; j < 3; ++j)
        if (tid + 1024 * j < 64 * 35) s_w3[tid + 1024 * j] = w3r[j];
#pragma unroll
    for (int j = 0; j < 2; ++j)
        if (tid + 1024 * j < 64 * 96 / 4) reinterpret_cast<f32x4*>(s_wf)[tid + 1024 * j] = wfr[j];
    if (tid < 64) s_w3[64 * 35 + tid] = b3r, s_wf[64 * 96 + tid] = bfr;
    if (tid == 0) {
        s_epoch = epoch_now + 1u;
        s_fail = 0;
    }
    const int nblk = (M2 + 63) >> 6, trips = (nblk + GL_GROUPS - 1) / GL_GROUPS;
    const int qq = lane >> 4, cc = lane & 15, o = 16 * g + cc;
    const int n_gran = B * GL_GROUPS * 128;
    const double n_rows = (double)B * (double)M2;
    __syncthreads();
    const unsigned epoch = s_epoch;
    // which running statistics workgroup 0 of the launch in front already updated (bit 0: SA3's, bit 1: FP3's)
    unsigned applied = 0;
    if constexpr (REPAIR) applied = __hip_atomic_load(&A.ctl[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    f32x4 V3[4];                                                 // SA3's outputs of the group's (last) block: the max reads them
#pragma unroll
    for (int t = 0; t < 4; ++t) V3[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    GSTAMP(0)
    // ================================================================ SA3: [x2 (32) | pos2 (3) | 1] -> 64
    for (int b = b_lo; b < b_hi; ++b) {
        constexpr int CA = 32, CB = 3, CI = CA + CB, CK = CI + 1, KB = (CK + 3) / 4, QS = OuterAcc<16, CK>::QS;
        const long row_lo = (long)b * M2, R_lim = row_lo + M2;
        float Wb[KB];
        float ssum = 0.f, ssq = 0.f;
        for (int it = 0; it < trips; ++it) {
            const int blk = it * GL_GROUPS + grp;
            const long r0 = row_lo + (long)blk * 64;
            if (blk < nblk) {
                stage_inputs<CA, CB, false>(s_q, QS, g, lane, r0, (int)R_lim, M2, M2, A.x2, 32, nullptr, nullptr, nullptr, nullptr,
                                            A.pos2, 4);
                if (g == 2) {
#pragma unroll
                    for (int k = CK; k < 4 * KB; ++k) s_q[lane * QS + k] = 0.f;
                }
            }
            __syncthreads();
            if (it == 0) {                       // (the weights' copy is complete behind the same barrier)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    const int k = 4 * kb + qq;
                    Wb[kb] = k < CI ? s_w3[o * CI + k] : (k == CI ? s_w3[64 * CI + o] : 0.f);
                }
            }
            if (blk < nblk) gl_block_tiles<KB>(s_q, QS, lane, g, r0, R_lim, Wb, 0.f, A.sa3.h, 64, ssum, ssq, V3);
            __syncthreads();
        }
        ssum += __shfl_xor(ssum, 16);
        ssq += __shfl_xor(ssq, 16);
        ssum += __shfl_xor(ssum, 32);
        ssq += __shfl_xor(ssq, 32);
        if (qq == 0) {
            s_red[grp * 128 + o] = ssum;
            s_red[grp * 128 + 64 + o] = ssq;
        }
        __syncthreads();
        if (b == b_lo) { GSTAMP(1) }
        // ---- publish the four groups' sums of this plot
        if (tid < GL_GROUPS * 128)
            __hip_atomic_store(A.xchg + (size_t)b * GL_GROUPS * 128 + tid,
                               ((gl_u64)(epoch * 2u + 0u) << 32) | (gl_u64)__float_as_uint(s_red[tid]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        if constexpr (REPAIR) __syncthreads();   // (s_red is the next plot's)
    }
    // ---- collect everybody's, finalise SA3's BatchNorm (every workgroup for itself)
    {
        const unsigned tag = epoch * 2u + 0u;
        if (!gl_collect(A.xchg, n_gran, tag, s_x, A.spin_limit)) s_fail = 1;
        __syncthreads();
        GSTAMP(2)
        if (s_fail) {
            // this workgroup's wait ran out: what it would compute from here on is wrong.  Tell workgroup 0 (POISON in place of
            // the FP3 sums), count, leave the repair launch behind this one to redo the level.
            if constexpr (!REPAIR) {
                if (tid < GL_GROUPS * 128)
                    __hip_atomic_store(A.xchg + n_gran + (size_t)b_lo * GL_GROUPS * 128 + tid, (gl_u64)((epoch * 2u + 1u) ^ GL_POISON) << 32,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (tid == 0) {
                    atomicAdd(&A.ctl[1], 1u);
                    if (lead) {
                        __hip_atomic_store(&A.ctl[3], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&A.ctl[0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                return;
            }
        }
        {
            // column col of the 128, sixteen threads each: partial sums over every sixteenth publisher, then the sixteen in order
            const int col = tid & 127, part = tid >> 7;          // 8 parts x 128 columns
            double acc = 0.0;
            for (int w = part; w < B * GL_GROUPS; w += 8) acc += (double)s_x[w * 128 + col];
            s_d[part * 128 + col] = acc;
        }
        __syncthreads();
        if (tid < 64) {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int part = 0; part < 8; ++part) s1 += s_d[part * 128 + tid], s2 += s_d[part * 128 + 64 + tid];
            float aa, cc2, mean, invstd;
            const bool upd = lead && !(applied & 1u);
            sn2_bn_from_sums(s1, s2, n_rows, A.sa3.gamma[tid], A.sa3.beta[tid], upd ? &A.sa3.running_mean[tid] : nullptr,
                             upd ? &A.sa3.running_var[tid] : nullptr, aa, cc2, mean, invstd);
            s_ac[tid] = aa;
            s_ac[64 + tid] = cc2;
            if (lead) {
                A.sa3.a[tid] = aa, A.sa3.c[tid] = cc2, A.sa3.mean[tid] = mean, A.sa3.invstd[tid] = invstd;
                if (tid == 0 && upd && A.sa3.nbt) *A.sa3.nbt += 1;
            }
        }
        __syncthreads();
    }
    GSTAMP(3)
    for (int b = b_lo; b < b_hi; ++b) {
    const long row_lo = (long)b * M2, R_lim = row_lo + M2;
    // ================================================================ the plot's max of a h + c (first row wins ties)
    if (!REPAIR && trips == 1) {
        // the group's block is still in registers (V3[t][j]: row 16 t + 4 qq + j of block grp, channel o): rows in ascending
        // order per lane, then the four row quarters (qq) of the channel, then the four groups -- ties to the lower row
        const float aa = s_ac[o], cc2 = s_ac[64 + o];
        float best = -INFINITY;
        int bi = 0x7FFFFFFF;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = grp * 64 + 16 * t + 4 * qq + j;
                const float y = fmaf(aa, V3[t][j], cc2);
                if (r < M2 && y > best) best = y, bi = r;
            }
#pragma unroll
        for (int m = 16; m <= 32; m <<= 1) {
            const float v = __shfl_xor(best, m);
            const int i = __shfl_xor(bi, m);
            if (v > best || (v == best && i < bi)) best = v, bi = i;
        }
        if (qq == 0) s_mv[grp * 64 + o] = best, s_mi[grp * 64 + o] = bi;
        __syncthreads();
        if (tid < 64) {
            best = s_mv[tid], bi = s_mi[tid];
            for (int k2 = 1; k2 < GL_GROUPS; ++k2) {
                const float v = s_mv[k2 * 64 + tid];
                const int i = s_mi[k2 * 64 + tid];
                if (v > best || (v == best && i < bi)) best = v, bi = i;
            }
            A.x3[(size_t)b * 64 + tid] = best;
            A.arg3[(size_t)b * 64 + tid] = bi;
            s_x3[tid] = best;
        }
        __syncthreads();
    } else {
        const int ch = tid & 63, rg = tid >> 6;                  // 16 row groups
        const float aa = s_ac[ch], cc2 = s_ac[64 + ch];
        float best = -INFINITY;
        int bi = 0x7FFFFFFF;
        const float* hb = A.sa3.h + (size_t)row_lo * 64 + ch;
        int r = rg;
        for (; r + 7 * 16 < M2; r += 8 * 16) {                   // eight row loads in flight
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = hb[(size_t)(r + u * 16) * 64];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float y = fmaf(aa, v[u], cc2);
                if (y > best) best = y, bi = r + u * 16;
            }
        }
        for (; r < M2; r += 16) {
            const float y = fmaf(aa, hb[(size_t)r * 64], cc2);
            if (y > best) best = y, bi = r;
        }
        s_mv[tid] = best;
        s_mi[tid] = bi;
        __syncthreads();
        if (tid < 64) {
            for (int k2 = 1; k2 < 16; ++k2) {
                const float v = s_mv[k2 * 64 + ch];
                const int i = s_mi[k2 * 64 + ch];
                if (v > best || (v == best && i < bi)) best = v, bi = i;
            }
            A.x3[(size_t)b * 64 + ch] = best;
            A.arg3[(size_t)b * 64 + ch] = bi;
            s_x3[ch] = best;
        }
        __syncthreads();
    }
    if (b == b_lo) { GSTAMP(4) }
    // ================================================================ FP3: [plot feature (64) | x2 (32) | 1] -> 64
    {
        // Every row of the plot interpolates the plot's ONE source: the 64 interpolated inputs are the plot feature x3[b] for
        // all of them (knn_interpolate with k = 1: x w / w), so their part of the layer is one vector per plot,
        // pv = b + W[:, 0:64] x3[b], and the rows contract their 32 skip channels only -- 8 k-steps instead of 25 and no
        // dependent gather in the staging.  (The separate kernel rebuilds x w / w per row: equal to x3 to an ulp.)
        constexpr int CA = 64, CB = 32, CI = CA + CB, KB = CB / 4, QS = OuterAcc<16, 36>::QS;
        float* s_pv = s_mv;                                      // [64]
        {
            const int oo = tid >> 4, part = tid & 15;            // output oo, inputs 4 part .. 4 part + 3
            const float4 w4 = *reinterpret_cast<const float4*>(s_wf + oo * CI + 4 * part);
            const float4 u4 = *reinterpret_cast<const float4*>(s_x3 + 4 * part);
            float acc = ((w4.x * u4.x + w4.y * u4.y) + w4.z * u4.z) + w4.w * u4.w;
#pragma unroll
            for (int m = 8; m > 0; m >>= 1) acc += __shfl_xor(acc, m);
            if (part == 0) s_pv[oo] = acc + s_wf[64 * CI + oo];
        }
        float Wb[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) Wb[kb] = s_wf[o * CI + CA + 4 * kb + qq];
        float ssum = 0.f, ssq = 0.f;
        f32x4 Vf[4];
        for (int it = 0; it < trips; ++it) {
            const int blk = it * GL_GROUPS + grp;
            const long r0 = row_lo + (long)blk * 64;
            if (blk < nblk && (REPAIR || trips > 1)) {
                // the skip rows: eight lanes per row, one float4 each; 32 rows per pass of the group's 256 threads
                // (one trip, one plot: the group's tile still holds them -- SA3 staged x2 into the same columns, row stride QS)
                const int t256 = tid & 255;
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    const int row = pass * 32 + (t256 >> 3), q8 = t256 & 7;
                    const long r = r0 + row;
                    const size_t rr = r < R_lim ? (size_t)r : (size_t)(R_lim - 1);
                    *reinterpret_cast<float4*>(&s_q[row * QS + 4 * q8]) = reinterpret_cast<const float4*>(A.x2 + rr * 32)[q8];
                }
            }
            __syncthreads();
            if (blk < nblk) gl_block_tiles<KB>(s_q, QS, lane, g, r0, R_lim, Wb, s_pv[o], A.fp3.h, 64, ssum, ssq, Vf);
            __syncthreads();
        }
        ssum += __shfl_xor(ssum, 16);
        ssq += __shfl_xor(ssq, 16);
        ssum += __shfl_xor(ssum, 32);
        ssq += __shfl_xor(ssq, 32);
        if (qq == 0) {
            s_red[grp * 128 + o] = ssum;
            s_red[grp * 128 + 64 + o] = ssq;
        }
    }
    __syncthreads();
    if (b == b_lo) { GSTAMP(5) }
    if (tid < GL_GROUPS * 128)
        __hip_atomic_store(A.xchg + n_gran + (size_t)b * GL_GROUPS * 128 + tid,
                           ((gl_u64)(epoch * 2u + 1u) << 32) | (gl_u64)__float_as_uint(s_red[tid]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    if constexpr (REPAIR) __syncthreads();       // (s_red, s_mv, s_x3 are the next plot's)
    }
    {
        gl_u64* gx = A.xchg + n_gran;
        const unsigned tag = epoch * 2u + 1u;
        if (!lead) return;
        if (!gl_collect(gx, n_gran, tag, s_x, A.spin_limit)) s_fail = 1;
        __syncthreads();
        GSTAMP(6)
        if (s_fail) {
            // (not in a repair launch: every granule it waits for is its own) a peer's FP3 sums did not arrive or are POISON:
            // SA3's running statistics are updated, FP3's are not -- the repair launch finishes the level
            if (tid == 0) {
                atomicAdd(&A.ctl[1], 1u);
                __hip_atomic_store(&A.ctl[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&A.ctl[0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        {
            const int col = tid & 127, part = tid >> 7;
            double acc = 0.0;
            for (int w = part; w < B * GL_GROUPS; w += 8) acc += (double)s_x[w * 128 + col];
            s_d[part * 128 + col] = acc;
        }
        __syncthreads();
        if (tid < 64) {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int part = 0; part < 8; ++part) s1 += s_d[part * 128 + tid], s2 += s_d[part * 128 + 64 + tid];
            float aa, cc2, mean, invstd;
            const bool upd = !(applied & 2u);
            sn2_bn_from_sums(s1, s2, n_rows, A.fp3.gamma[tid], A.fp3.beta[tid], upd ? &A.fp3.running_mean[tid] : nullptr,
                             upd ? &A.fp3.running_var[tid] : nullptr, aa, cc2, mean, invstd);
            A.fp3.a[tid] = aa, A.fp3.c[tid] = cc2, A.fp3.mean[tid] = mean, A.fp3.invstd[tid] = invstd;
            if (tid == 0 && upd && A.fp3.nbt) *A.fp3.nbt += 1;
        }
        __syncthreads();
        GSTAMP(7)
        if (tid == 0) {
            if constexpr (REPAIR) {
                // the level is whole again: mark the give-ups as handled, and move the epoch past every tag a late workgroup of
                // the failed launch may have written (it read ctl[0] after workgroup 0 advanced it: epoch + 1)
                __hip_atomic_store(&A.ctl[2], __hip_atomic_load(&A.ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&A.ctl[0], epoch + 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_store(&A.ctl[3], 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&A.ctl[0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// The launch: one workgroup per plot runs the level (gl_level_body<false>); then the exit protocol -- every workgroup takes a
// ticket on its way out, and the one that takes the LAST ticket (all its peers have left) looks at the give-up count: unchanged
// since the last repair (the usual case: one barrier and one atomic per workgroup) and it leaves too; otherwise it runs the
// whole level again alone (gl_level_body<true>).  (Round 5's first version repaired with a launch of its own behind this one:
// 4.8 us for a kernel that reads two words -- more than the fused launch saves.)
// No device-scope fence in that protocol (one per workgroup cost 3.5 us of the kernel's 27), and none is needed:
//   * the words the repair DECIDES by (ctl[], the granules) are agent-scope atomics issued in front of the workgroup's barrier,
//     hence complete before its ticket;
//   * everything else a workgroup of the failed launch wrote with plain stores is either the value the repair writes itself --
//     SA3's rows do not depend on the exchange; FP3's rows, x3, a / c / mean / invstd were only written by workgroups whose
//     collection of the statistics was COMPLETE, i.e. from the same sums in the same order -- so a late store of it changes
//     nothing, or it is guarded: the running statistics and counters of a layer are touched by workgroup 0 OR by the repair,
//     never both (ctl[3], an atomic, says which).
__global__ __launch_bounds__(1024) void global_level_fwd_kernel(GlArgs A) {
    extern __shared__ __attribute__((aligned(16))) float gl_smem[];
    __shared__ int s_repair;
    gl_level_body<false>(A, gl_smem);
    __syncthreads();                             // (every wave's atomics of the body are complete)
    if (threadIdx.x == 0) {
        const unsigned t = atomicAdd(&A.ctl[4], 1u);
        int repair = 0;
        if (t == gridDim.x - 1) {                // every other workgroup of the launch has left its body
            __hip_atomic_store(&A.ctl[4], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            repair = __hip_atomic_load(&A.ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) !=
                     __hip_atomic_load(&A.ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_repair = repair;
    }
    __syncthreads();
    if (!s_repair) return;
    gl_level_body<true>(A, gl_smem);
}

#ifdef SN2_SPLIT_STAMPS
// diagnostic build only (never shipped): phase stamps of thread 0 of one workgroup of fp_bwd_split_kernel<64, 32, 64>
__device__ unsigned long long g_split_dbg[16];
extern "C" int sn2_debug_split_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_split_dbg), sizeof(g_split_dbg));
}
#define FSTAMP(i)                                                                                   \
    if (CA == 64 && CB == 32 && blockIdx.x == 7 && threadIdx.x == 0) {                              \
        unsigned long long t_;                                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                   \
        g_split_dbg[i] = t_;                                                                        \
    }
#else
#define FSTAMP(i)
#endif

// LDS of fp_bwd_split_kernel: inputs [64][QS] + per-wave dp [4][64][16] + the input-gradient slab(s) [1 or 4][64][CI|1]
// + the BatchNorm vectors [5][64]
template <int CI>
constexpr size_t fp_split_lds_bytes(int slabs) {
    return (size_t)(64 * OuterAcc<16, CI + 1>::QS + 4 * 64 * 16 + slabs * 64 * (CI | 1) + 5 * 64) * 4;
}
template <int CI>
constexpr bool fp_split_du_seq() { return fp_split_lds_bytes<CI>(4) > 150 * 1024; }   // one slab, the waves take turns

template <int CA, int CB, int CO, bool KNN, bool BF16>
__global__ __launch_bounds__(256) void fp_bwd_split_kernel(
    int R, int R_per_plot, int S_per_plot, int src_stride, int skip_stride, int h_stride, int dskip_stride, int du_stride,
    float invR, const float* __restrict__ src, const float* __restrict__ src_a, const float* __restrict__ src_c,
    const int* __restrict__ knn_idx, const float* __restrict__ knn_w, const float* __restrict__ skip,
    const float* __restrict__ W, const float* __restrict__ gamma, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ dgamma, const float* __restrict__ dbeta,
    const float* __restrict__ h, const float* __restrict__ dy, float* __restrict__ dW, float* __restrict__ db,
    float* __restrict__ du_out, float* __restrict__ dskip, int rep_k, int rep_stride) {
    constexpr int CI = CA + CB, COG = (CO + 3) / 4;
    using Acc = OuterAcc<16, CI + 1>;
    constexpr int QS = Acc::QS, TK = Acc::TK, CIP = CI | 1;
    constexpr bool DU_SEQ = fp_split_du_seq<CI>();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_q = smem;                       // [64][QS]  the rows' inputs | 1, shared by the four waves
    float* s_p = s_q + 64 * QS;              // [4][64][16] per-wave d pre-activation of its channel group
    float* s_du = s_p + 4 * 64 * 16;         // [4][64][CIP] per-wave partial input gradient (summed at write-out: LDS
                                             // float atomics are slow on this chip, plain stores + 4 reads are not)
    const int lane = threadIdx.x & 63;
    const int g = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long r = (long)blockIdx.x * 64 + lane;
    const bool valid = r < R;
    const size_t rr = valid ? (size_t)r : (size_t)(R - 1);
    FSTAMP(0);
    // no clearing pass over the staging regions (it cost 4000 clocks + a barrier): the input columns past CI only feed
    // output columns nobody reads, the channel pads of s_p are written below
    FSTAMP(1);
    stage_inputs<CA, CB, KNN>(s_q, QS, g, lane, (long)blockIdx.x * 64, R, R_per_plot, S_per_plot, src, src_stride, src_a, src_c,
                              knn_idx, knn_w, skip, skip_stride);
    // d pre-activation: the block's 64 x CO tile of h and dy read thread-linearly as float4 quads (whole rows per load
    // instruction, as above), the five per-channel vectors of the BatchNorm backward staged in LDS
    constexpr int HS4 = (CO + 3) / 4, NU = (64 * HS4 + 255) / 256;
    float* s_par = s_du + (DU_SEQ ? 1 : 4) * 64 * CIP;       // [5][64]: mean, invstd, gamma, dbeta, dgamma
    if (threadIdx.x < CO) {
        const int o = threadIdx.x;
        s_par[0 * 64 + o] = mean[o], s_par[1 * 64 + o] = invstd[o], s_par[2 * 64 + o] = gamma[o];
        s_par[3 * 64 + o] = dbeta[o], s_par[4 * 64 + o] = dgamma[o];
    }
    float4 hv[NU], dv[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int i4 = threadIdx.x + 256 * u, row = i4 / HS4, qd = i4 - row * HS4;
        const long rw = (long)blockIdx.x * 64 + row;
        const size_t ra = (i4 < 64 * HS4 && rw < R) ? (size_t)rw : (size_t)(R - 1);
        hv[u] = reinterpret_cast<const float4*>(h + ra * h_stride)[i4 < 64 * HS4 ? qd : 0];
        dv[u] = reinterpret_cast<const float4*>(dy + ra * h_stride)[i4 < 64 * HS4 ? qd : 0];
    }
#pragma unroll
    for (int t = 0; t < 16; ++t)
        if (t >= COG || g * COG + t >= CO) s_p[(g * 64 + lane) * 16 + t] = 0.f;   // K padding of dp . W: finite
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int i4 = threadIdx.x + 256 * u, row = i4 / HS4, qd = i4 - row * HS4;
        const bool live = i4 < 64 * HS4 && (long)blockIdx.x * 64 + row < R;
        const float hq[4] = {hv[u].x, hv[u].y, hv[u].z, hv[u].w}, dq[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
        if (i4 < 64 * HS4) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int o = 4 * qd + t;
                if (o < CO) {
                    const float hh = hq[t], dd = dq[t];
                    const float is = s_par[1 * 64 + o];
                    const float xh = (hh - s_par[0 * 64 + o]) * is;
                    const float dh = s_par[2 * 64 + o] * is * (dd - s_par[3 * 64 + o] * invR - xh * s_par[4 * 64 + o] * invR);
                    s_p[((o / COG) * 64 + row) * 16 + (o % COG)] = (live && hh > 0.f) ? dh : 0.f;
                }
            }
        }
    }
    FSTAMP(2);
    __syncthreads();
    FSTAMP(3);
    // dW|db rows of this channel group: rows of the block are the MFMA K dimension
    f32x4 acc[TK];
#pragma unroll
    for (int c = 0; c < TK; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        const int r4 = lane >> 4, c16 = lane & 15;
        const float* rp = s_p + (g * 64 + r4) * 16 + c16;
        const float* rq = s_q + r4 * QS + c16;
        if constexpr (!BF16) {
#pragma unroll 4
            for (int st = 0; st < 16; ++st) {
                const float av = rp[st * 4 * 16];
#pragma unroll
                for (int c = 0; c < TK; ++c)
                    acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, rq[st * 4 * QS + c * 16], acc[c], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s8 = 0; s8 < 2; ++s8) {          // the block's 64 rows = the K of two v_mfma_f32_16x16x32_bf16
                float av[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) av[u] = rp[(8 * s8 + u) * 4 * 16];
#pragma unroll
                for (int c = 0; c < TK; ++c)
                    acc[c] = contract<true, 8>(acc[c], [&](int u) { return av[u]; },
                                               [&](int u) { return rq[(8 * s8 + u) * 4 * QS + c * 16]; });
            }
        }
    }
    FSTAMP(4);
    // partial input gradient of this channel group, dp_g (64 x 16) . W_g (16 x CI), on the matrix cores: A[row][o] read
    // back from this wave's staged dp rows, B[o][col] = the group's weight rows in registers (as per-lane FMA chains with
    // scalar-loaded weights this was the longest part of the kernel: 16 x 96 FMAs per lane for FP3)
    {
        constexpr int TJ = (CI + 15) / 16;
        const int qq = lane >> 4, cc = lane & 15;
        float Wb[4][TJ];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int jt = 0; jt < TJ; ++jt) {
                const int t = 4 * kb + qq, o = g * COG + t, col = 16 * jt + cc;
                Wb[kb][jt] = (t < COG && o < CO && col < CI) ? W[o * CI + col] : 0.f;
            }
        // DU_SEQ (wide inputs: CI = 128): one slab for the workgroup, the four waves add their partial products in turn
        // (four slabs would not fit LDS); otherwise one slab per wave, summed at write-out
        for (int turn = 0; turn < (DU_SEQ ? 4 : 1); ++turn) {
            if (!DU_SEQ || g == turn) {
#pragma unroll
                for (int tile = 0; tile < 4; ++tile) {
                    f32x4 D[TJ];
#pragma unroll
                    for (int jt = 0; jt < TJ; ++jt) D[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if constexpr (!BF16) {
#pragma unroll
                        for (int kb = 0; kb < 4; ++kb) {
                            const float av = s_p[(g * 64 + 16 * tile + cc) * 16 + 4 * kb + qq];
#pragma unroll
                            for (int jt = 0; jt < TJ; ++jt)
                                D[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Wb[kb][jt], D[jt], 0, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int jt = 0; jt < TJ; ++jt)
                            D[jt] = contract<true, 4>(D[jt], [&](int kb) { return s_p[(g * 64 + 16 * tile + cc) * 16 + 4 * kb + qq]; },
                                                      [&](int kb) { return Wb[kb][jt]; });
                    }
#pragma unroll
                    for (int jt = 0; jt < TJ; ++jt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int col = 16 * jt + cc;
                            if (col < CI) {
                                float* dst = &s_du[((DU_SEQ ? 0 : g) * 64 + 16 * tile + 4 * qq + r) * CIP + col];
                                *dst = (DU_SEQ && turn > 0) ? *dst + D[jt][r] : D[jt][r];
                            }
                        }
                }
            }
            if (DU_SEQ) __syncthreads();
        }
    }
    FSTAMP(5);
    __syncthreads();
    FSTAMP(6);
    auto du_sum = [&](int row, int k) {
        if (DU_SEQ) return s_du[row * CIP + k];
        return (s_du[(0 * 64 + row) * CIP + k] + s_du[(1 * 64 + row) * CIP + k]) +
               (s_du[(2 * 64 + row) * CIP + k] + s_du[(3 * 64 + row) * CIP + k]);
    };
    const long r0 = (long)blockIdx.x * 64;
    // accumulated outputs: all the old values are loaded before the first is used (one dependent load per loop turn made
    // this write-out 12 000 clocks, a quarter of the kernel)
    if (du_out) {
        constexpr int NI = (64 * CA + 255) / 256;
        float old[NI];
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int i = threadIdx.x + 256 * u, row = i / CA, k = i - row * CA;
            old[u] = (!KNN && i < 64 * CA && r0 + row < R) ? du_out[(size_t)(r0 + row) * du_stride + k] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int i = threadIdx.x + 256 * u, row = i / CA, k = i - row * CA;
            if (i < 64 * CA && r0 + row < R) {
                float* dst = du_out + (size_t)(r0 + row) * du_stride + k;
                if (KNN) *dst = du_sum(row, k);
                else *dst = old[u] + du_sum(row, k);
            }
        }
    }
    if (CB > 0 && dskip) {
        constexpr int NI = (64 * CB + 255) / 256;
        float old[NI > 0 ? NI : 1];
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int i = threadIdx.x + 256 * u, row = i / (CB > 0 ? CB : 1), k = i - row * CB;
            old[u] = (i < 64 * CB && r0 + row < R) ? dskip[(size_t)(r0 + row) * dskip_stride + k] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int i = threadIdx.x + 256 * u, row = i / (CB > 0 ? CB : 1), k = i - row * CB;
            if (i < 64 * CB && r0 + row < R) dskip[(size_t)(r0 + row) * dskip_stride + k] = old[u] + du_sum(row, CA + k);
        }
    }
    FSTAMP(7);
    {
        const int r4 = lane >> 4, c16 = lane & 15, img = sn2_grad_image(rep_k, rep_stride);
#pragma unroll
        for (int c = 0; c < TK; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int t = r4 * 4 + q, o = g * COG + t, k = c * 16 + c16;
                if (t < COG && o < CO) {
                    if (k < CI) SN2_FLUSH_ADD(&dW[img + o * CI + k], acc[c][q]);
                    else if (k == CI) SN2_FLUSH_ADD(&db[img + o], acc[c][q]);
                }
            }
    }
    FSTAMP(8);
}

// which row pass the source-side forward launches: 1 (default) fp_fwd_rows2_kernel, 0 fp_fwd_rows_kernel (same bits)
static int g_fp_rows_form = (getenv("SN2_FP_ROWS2") && atoi(getenv("SN2_FP_ROWS2")) == 0) ? 0 : 1;
// (experiment switch) SN2_GRID_MULT: the row kernels' grids times this factor (more, shorter workgroups than the chip holds at
// once: the hardware then hands the later ones to whichever CU frees up first -- dynamic balancing beside concurrent kernels)
static const int grid_mult = getenv("SN2_GRID_MULT") ? atoi(getenv("SN2_GRID_MULT")) : 1;

int pick_grid(long R, int threads, int rows_per_lane) {
    long g = (R + (long)threads * rows_per_lane - 1) / ((long)threads * rows_per_lane);
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    return (int)g;
}

// the source-side form applies: workspace given, quads aligned, 32-bit row offsets
template <int CA, int CO>
bool fp_source_side_ok(const sn2_fp* p) {
    const long R = (long)p->B * p->R_per_plot;
    return p->src_ws && p->knn_idx && (p->skip_stride & 3) == 0 && p->src_stride >= 4 * ((CA + 3) / 4) &&
           p->h_stride == 4 * ((CO + 3) / 4) && R * 3 < (1L << 31);
}

template <int CA, int CB, int CO, bool KNN>
int fp_forward_t(const sn2_fp* p, int mode, hipStream_t st) {
    // mode == SN2_BN_FROZEN_KEEP: the kernel a TRAINING pass of this shape takes (the backward pass that follows chooses its
    // kernels by the same rules), without statistic sums; BatchNorm finalised from the running statistics, nothing updated
    const int training = mode == 1, keep = mode != 0;
    const int R = p->B * p->R_per_plot;
    // the matrix-core kernel (64 rows x 4 channel groups per workgroup): its per-workgroup statistic slots bound the rows of a
    // TRAINING pass; an eval pass writes no statistics, so any row count takes it (parcel inference: FP3 on 80 000 rows ran
    // the scalar-weight fallback below at 0.31 ms, 15 x what the contraction needs)
    const bool small = sn2_cdiv(R, 64) <= SN2_STAT_SLOTS;
    bool src_side = false;                      // (the per-point layer keeps its source-side form in both modes)
    if constexpr (KNN && CB > 0 && CB % 4 == 0 && CB <= 16) src_side = !small && fp_source_side_ok<CA, CO>(p);
    if (small || (!keep && !src_side)) {
        if (p->act_bf16) return SN2_ELIMIT;
        const int grid = sn2_cdiv(R, 64);
        constexpr size_t lf = (size_t)(64 * OuterAcc<16, CA + CB + 1>::QS + 2 * 16 * ((CO + 15) / 16)) * sizeof(float);
        static_assert(lf <= 48 * 1024, "fp_fwd_split_kernel staging");
        auto kf = p->blk.mma_bf16 ? &fp_fwd_split_kernel<CA, CB, CO, KNN, true> : &fp_fwd_split_kernel<CA, CB, CO, KNN, false>;
        hipLaunchKernelGGL(kf, dim3(grid), dim3(256), lf, st, R, p->R_per_plot,
                           p->S_per_plot, p->src_stride, p->skip_stride, p->h_stride, p->src, p->src_a, p->src_c, p->knn_idx,
                           p->knn_w, p->skip, p->blk.W, p->blk.b, p->h, training ? p->blk.stat_slots : (float*)nullptr);
        hipError_t e0 = hipGetLastError();
        if (e0 != hipSuccess) return (int)e0;
        return sn2_bn_finalize(&p->blk, training ? grid : 0, nullptr, R, training, st);
    }
    if (p->blk.mma_bf16) return SN2_ELIMIT;   // bf16 operands exist on the matrix-core kernel of the small layers only
    if constexpr (KNN && CB > 0 && CB % 4 == 0 && CB <= 16) {
        if (fp_source_side_ok<CA, CO>(p)) {               // the per-point layer: source-side form
            const int n_src = p->B * p->S_per_plot;
            SN2_TRY((launch_src_table<CA, CB, CO>(n_src, p->src_stride, p->src, p->src_a, p->src_c, p->blk.W, p->src_ws, st)));
            const long n_grp = sn2_cdiv(R, 64 / ((CO + 3) / 4));
            int grid = sn2_cdiv(n_grp, 8);
            // two workgroups per CU: at 143 VGPRs three waves fit a SIMD, so 1024 workgroups ran as one full round and a
            // third of a second one (0.057 ms; 768: 0.056; 512: 0.052; 384: 0.058)
            static const int wgs_per_cu = getenv("SN2_FR_WGS_PER_CU") ? atoi(getenv("SN2_FR_WGS_PER_CU")) : 2;      // (experiment switch)
            const int cap_fwd_rows = wgs_per_cu * grid_mult * sn2_cu_count() < SN2_STAT_SLOTS ? wgs_per_cu * grid_mult * sn2_cu_count() : SN2_STAT_SLOTS;
            if (grid > cap_fwd_rows) grid = cap_fwd_rows;
            // (round 4: a variant with the plot's whole table in LDS -- 144 KB, one 16-wave workgroup per CU, the 226 MB of L2
            // gathers replaced by ds_read_b128 -- ran in 38.1 us against this kernel's 36.5: the gathers are not its bound; at
            // ~14 instructions per row and wave-instruction it is instruction issue, like the head kernels; the skip part's FMA
            // chains as v_pk_fma_f32 pairs -- 32 instructions fewer per two rows -- ran in 38.0 us as well)
            // (round 5: the row pass with its input stream fetched one element per lane, four iterations ahead;
            // sn2_debug_fp_rows_form(0) / SN2_FP_ROWS2=0: the first form, kept for cross-checks -- same bits)
            const bool rows2 = g_fp_rows_form != 0;
            auto kr = rows2 ? (p->act_bf16 ? &fp_fwd_rows2_kernel<CA, CB, CO, true> : &fp_fwd_rows2_kernel<CA, CB, CO, false>)
                            : (p->act_bf16 ? &fp_fwd_rows_kernel<CA, CB, CO, true> : &fp_fwd_rows_kernel<CA, CB, CO, false>);
            hipLaunchKernelGGL(kr, dim3(grid), dim3(256), 0, st, R, p->R_per_plot, p->S_per_plot,
                               p->skip_stride, (const float*)p->src_ws, p->knn_idx, p->knn_w, p->skip, p->blk.W, p->blk.b, p->h,
                               training ? p->blk.stat_slots : (float*)nullptr);
            hipError_t e1 = hipGetLastError();
            if (e1 != hipSuccess) return (int)e1;
            return sn2_bn_finalize(&p->blk, grid, nullptr, R, training, st);
        }
    }
    if (p->act_bf16) return SN2_ELIMIT;       // bfloat16 rows: the source-side form of the per-point layer only
    int grid = pick_grid(R, 256, 4);
    if (grid > SN2_STAT_SLOTS) grid = SN2_STAT_SLOTS;
    hipLaunchKernelGGL((fp_fwd_kernel<CA, CB, CO, KNN>), dim3(grid), dim3(256), 0, st, R, p->R_per_plot,
                       p->S_per_plot, p->src_stride, p->skip_stride, p->h_stride, p->src, p->src_a, p->src_c, p->knn_idx,
                       p->knn_w, p->skip, p->blk.W, p->blk.b, p->h, training ? p->blk.stat_slots : (float*)nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    return sn2_bn_finalize(&p->blk, grid, nullptr, R, training, st);
}

// the inverted index of a 3-NN table (kernels A-C, E above); workspace carve (32-bit words):
// H [B*SL*S] | off [B*S] | cnt [B*S] | inv_row [3*B*Rp] | inv_w [3*B*Rp] | (16-byte aligned) items [B*S] int4 |
// chunks [B*CM] int4
struct InterpIndex {
    int *H, *off, *cnt, *inv_row;
    int4 *items, *chunks;
    float* inv_w;
    int CM;
};
InterpIndex carve_interp_index(float* ws, int B, int Rp, int S) {
    const int SL = sn2_cdiv(Rp, INV_SLICE_ROWS);
    InterpIndex x;
    x.H = reinterpret_cast<int*>(ws);
    x.off = x.H + (size_t)B * SL * S;
    x.cnt = x.off + (size_t)B * S;
    x.inv_row = x.cnt + (size_t)B * S;
    x.inv_w = reinterpret_cast<float*>(x.inv_row + (size_t)3 * B * Rp);
    x.items = reinterpret_cast<int4*>((reinterpret_cast<uintptr_t>(x.inv_w + (size_t)3 * B * Rp) + 15) & ~(uintptr_t)15);
    x.chunks = x.items + (size_t)B * S;
    x.CM = inv_chunks_per_plot(Rp, S);
    return x;
}
// G batches of B plots each in one set of launches (G = 1: one batch); batch h's workspace at ws + h * ws_stride words
int build_interp_index(const int* knn_idx, const float* knn_w, const float* src_pos, int B, int Rp, int S, float* ws,
                       hipStream_t st, const int* row_perm = nullptr, int G = 1, size_t ws_stride = 0) {
    if (S > 8192) return SN2_ELIMIT;
    if (G < 1 || (G > 1 && (ws_stride & 3))) return SN2_EINVAL;           // (every batch's workspace 16-byte aligned)
    const int SL = sn2_cdiv(Rp, INV_SLICE_ROWS);
    const int CM = inv_chunks_per_plot(Rp, S);
    if ((long)G * B >= 65535) return SN2_ELIMIT;                           // grid.y
    hipLaunchKernelGGL(inv_hist_kernel, dim3(SL, G * B), dim3(1024), (size_t)S * 4, st, Rp, S, knn_idx, knn_w, ws, B, ws_stride);
    hipLaunchKernelGGL(inv_scan_kernel, dim3(G * B), dim3(1024), 0, st, Rp, S, SL, ws, B, ws_stride);
    hipLaunchKernelGGL(inv_fill_kernel, dim3(SL, G * B), dim3(1024), (size_t)S * 4, st, Rp, S, knn_idx, knn_w, ws, B, ws_stride,
                       row_perm);
    // keys + the source of every rank: 64 KB of dynamic LDS at the S = 8192 limit (+ ~450 B static): above the 48 KB a kernel
    // gets without asking
    const size_t order_lds = (size_t)(((S + 3) & ~3) + S) * 4;
    if (order_lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&inv_order_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)order_lds);
    hipLaunchKernelGGL(inv_order_kernel, dim3(G * B), dim3(1024), order_lds, st, reinterpret_cast<const float4*>(src_pos), S, CM, Rp,
                       ws, B, ws_stride);
    SN2_RETURN_LAUNCH();
}

// diagnostic (bench.py): which of the three kernels of the per-point layer's source-side backward run -- bit 0 the row pass, 1 the
// source pass over the chunk table, 2 the merge (7 = all, the only setting that computes the gradients)
extern "C" int sn2_debug_fp_table_form(int form) {
    g_fp_table_form = form ? 1 : 0;
    return 0;
}
extern "C" int sn2_debug_fp_rows_form(int form) {
    g_fp_rows_form = form ? 1 : 0;
    return 0;
}
static int g_fp1_bwd_parts = 7;
extern "C" int sn2_debug_fp1_backward_parts(int mask) {
    g_fp1_bwd_parts = mask & 7 ? mask & 7 : 7;
    return 0;
}

template <int CA, int CB, int CO, bool KNN>
int fp_backward_t(const sn2_fp* p, hipStream_t st) {
    constexpr int CI = CA + CB;
    using Acc = OuterAcc<CO, CI + 1, 32>;
    // waves per workgroup (one workgroup per CU): as many as the staging regions and 256 VGPRs per lane allow
    constexpr int WAVES = (Acc::LDS_FLOATS * 4 * 8 <= 150 * 1024) ? 8 : ((Acc::LDS_FLOATS * 4 * 4 <= 150 * 1024) ? 4 : 2);
    const int R = p->B * p->R_per_plot;
    // 1 / (rows of the batch statistics); frozen_stats: the forward's statistics were constants (running statistics) -- the
    // batch-mean and batch-variance terms of the BatchNorm backward vanish
    const float invR = p->blk.frozen_stats ? 0.f : 1.0f / (float)R;
    if (p->act_bf16 && !(p->bn_sums_done && p->src_ws && p->du_scratch && p->scatter_ws && p->dsrc && !p->dskip &&
                         sn2_cdiv(R, 64) > SN2_STAT_SLOTS))
        return SN2_ELIMIT;                    // bfloat16 rows: the source-side form of the per-point layer only
    // bn_sums_done (non-NULL): dgamma / dbeta of this block's BatchNorm already came from sn2_head_bn_sums / sn2_fp_bn_sums
    if (p->bn_sums_done) {
        // sn2_head_bn_sums / sn2_fp_bn_sums already completed dgamma / dbeta of this block (by the identity or by their
        // own pass over the rows)
    } else if (R <= (1 << 16)) {
        hipLaunchKernelGGL((fp_bwd_bn_small_kernel<CO>), dim3(sn2_cdiv(R, 64)), dim3(256), 0, st, R, p->h_stride, p->h, p->dy,
                           p->blk.mean, p->blk.invstd, p->blk.dgamma, p->blk.dbeta, p->bn_sums_done);
    } else {
        int g1 = pick_grid(R, 256, R >= (1 << 18) ? 8 : 1);
        if (g1 > 256) g1 = 256;
        hipLaunchKernelGGL((fp_bwd_bn_kernel<CO>), dim3(g1), dim3(256), 0, st, R, p->h_stride, p->h, p->dy,
                           p->blk.mean, p->blk.invstd, p->blk.dgamma, p->blk.dbeta, p->bn_sums_done);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    float* du_out0 = KNN ? p->du_scratch : p->dsrc;
    if (KNN && p->dsrc && !p->du_scratch) return SN2_EINVAL;
    if (KNN && !p->dsrc) du_out0 = nullptr;
    constexpr size_t lb = fp_split_lds_bytes<CI>(fp_split_du_seq<CI>() ? 1 : 4);
    static_assert(lb <= 150 * 1024, "fp_bwd_split_kernel staging must fit LDS");
    const bool small = sn2_cdiv(R, 64) <= SN2_STAT_SLOTS;   // the 64-row x 4-channel-group kernel for small layers
    if (!small && p->blk.mma_bf16) return SN2_ELIMIT;       // bf16 operands: that kernel only
    if (small) {
        auto ks = p->blk.mma_bf16 ? &fp_bwd_split_kernel<CA, CB, CO, KNN, true> : &fp_bwd_split_kernel<CA, CB, CO, KNN, false>;
        if (lb > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
        hipLaunchKernelGGL(ks, dim3(sn2_cdiv(R, 64)), dim3(256), lb, st, R, p->R_per_plot, p->S_per_plot, p->src_stride,
                           p->skip_stride, p->h_stride, p->dskip_stride, KNN ? CA : p->dsrc_stride, invR, p->src,
                           p->src_a, p->src_c, p->knn_idx, p->knn_w, p->skip, p->blk.W, p->blk.gamma,
                           (const float*)p->blk.mean, (const float*)p->blk.invstd, (const float*)p->blk.dgamma,
                           (const float*)p->blk.dbeta, (const float*)p->h, p->dy, p->blk.dW, p->blk.db, du_out0, p->dskip,
                           p->blk.grad_replicas, p->blk.grad_replica_stride);
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    if constexpr (KNN && CB > 0 && CB % 4 == 0 && CB <= 16) {
        if (!small && fp_source_side_ok<CA, CO>(p) && p->dsrc && !p->dskip && p->du_scratch && p->scatter_ws) {
            const int S = p->S_per_plot, Rp = p->R_per_plot, B = p->B, n_src = B * S;
            if (S > 8192) return SN2_ELIMIT;
            constexpr int NT = 512;                       // 2 workgroups x 8 waves per CU
            constexpr size_t lb1 = (size_t)(NT / 64) * (4 * ((CO + 3) / 4)) * (CB + 1) * sizeof(float);
            auto k1 = p->act_bf16 ? &fp_bwd_rows_kernel<CA, CB, CO, NT, true> : &fp_bwd_rows_kernel<CA, CB, CO, NT, false>;
            if (lb1 > 48 * 1024)
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb1);
            const int parts = g_fp1_bwd_parts;
            if (parts & 1)
            hipLaunchKernelGGL(k1, dim3(2 * grid_mult * sn2_cu_count()), dim3(NT), lb1, st, R, p->skip_stride, invR, p->skip,
                               p->blk.gamma, (const float*)p->blk.mean, (const float*)p->blk.invstd,
                               (const float*)p->blk.dgamma, (const float*)p->blk.dbeta, (const float*)p->h, p->dy,
                               p->du_scratch, p->blk.dW, p->blk.db, p->blk.grad_replicas, p->blk.grad_replica_stride,
                               p->row_perm, Rp);
            if (!p->scatter_ready) SN2_TRY(build_interp_index(p->knn_idx, p->knn_w, nullptr, B, Rp, S, p->scatter_ws, st, p->row_perm));
            const InterpIndex x = carve_interp_index(p->scatter_ws, B, Rp, S);
            // waves over the chunk table (as many as the chip holds at once, L <= 64 slots each), then a lane per source for
            // what G[s] is for
            if ((long)B * x.CM >= (1L << 31) / 64) return SN2_ELIMIT;
            const int n_chunks = B * x.CM;
            const int waves_resident = sn2_cu_count() * 32;
            int L = sn2_cdiv(n_chunks, waves_resident);
            if (L < 4) L = 4;
            static const int L_env = getenv("SN2_FP_SRC_L") ? atoi(getenv("SN2_FP_SRC_L")) : 0;
            if (L_env > 0) L = L_env;
            if (L > 64) L = 64;
            const int gc = (sn2_cdiv(sn2_cdiv(n_chunks, L), 4) + 7) & ~7;
            auto kc = p->act_bf16 ? &fp_bwd_src_chunk_kernel<CA, CB, CO, true> : &fp_bwd_src_chunk_kernel<CA, CB, CO, false>;
            if (parts & 2)
            hipLaunchKernelGGL(kc, dim3(gc), dim3(256), 0, st, n_chunks, L, R, Rp, S, (const int4*)x.chunks, (const int*)x.inv_row,
                               (const float*)x.inv_w, (const float*)p->du_scratch, p->src_ws);
            if (parts & 4)
            hipLaunchKernelGGL((fp_bwd_src_merge_dw_kernel<CA, CB, CO>), dim3(sn2_cdiv(n_src, 64)), dim3(256), 0, st, n_src, S, x.CM,
                               p->src_stride, p->dsrc_stride, (const int4*)x.items, p->src, p->src_a, p->src_c,
                               (const float*)p->src_ws, p->blk.W, p->dsrc,
                               p->blk.dW, p->blk.grad_replicas, p->blk.grad_replica_stride, L);
            SN2_RETURN_LAUNCH();
        }
    }
    if (p->act_bf16 || p->row_perm) return SN2_ELIMIT;    // (both belong to the source-side form above)
    constexpr size_t lds_bytes = (size_t)Acc::LDS_FLOATS * 4 * WAVES;
    auto kern = &fp_bwd_main_kernel<CA, CB, CO, KNN, WAVES>;
    if (lds_bytes > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds_bytes);
    float* du_out = KNN ? p->du_scratch : p->dsrc;
    if (KNN && p->dsrc && !p->du_scratch) return SN2_EINVAL;
    if (KNN && !p->dsrc) du_out = nullptr;
    int grid = pick_grid(R, WAVES * 64, 4);
    if (grid > 256) grid = 256;
    if (!small) hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds_bytes, st, R, p->R_per_plot, p->S_per_plot, p->src_stride,
                       p->skip_stride, p->h_stride, p->dskip_stride, KNN ? CA : p->dsrc_stride, invR, p->src, p->src_a, p->src_c,
                       p->knn_idx, p->knn_w, p->skip, p->blk.W, p->blk.gamma, (const float*)p->blk.mean,
                       (const float*)p->blk.invstd, (const float*)p->blk.dgamma, (const float*)p->blk.dbeta,
                       (const float*)p->h, p->dy, p->blk.dW, p->blk.db, du_out, p->dskip, p->blk.grad_replicas,
                       p->blk.grad_replica_stride);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (KNN && p->dsrc && p->scatter_ready >= 0) {       // (scatter_ready < 0: the caller transposes the interpolation itself)
        if (!p->scatter_ws) return SN2_EINVAL;
        const int S = p->S_per_plot, Rp = p->R_per_plot, B = p->B;
        if (S > 8192) return SN2_ELIMIT;
        if (!p->scatter_ready) SN2_TRY(build_interp_index(p->knn_idx, p->knn_w, nullptr, B, Rp, S, p->scatter_ws, st));
        const InterpIndex x = carve_interp_index(p->scatter_ws, B, Rp, S);
        const int* off = x.off;
        const int* cnt = x.cnt;
        const int* inv_row = x.inv_row;
        const float* inv_w = x.inv_w;
        const int n_src = B * S;
        if ((long)Rp >= 128L * S && CA <= 64)     // >= 128 entries per list on average: a workgroup per source
            hipLaunchKernelGGL((interp_gather_long_kernel<CA, 16>), dim3(n_src), dim3(1024), 0, st, n_src, Rp, S,
                               p->dsrc_stride, (const int*)off, (const int*)cnt, (const int*)inv_row, (const float*)inv_w,
                               (const float*)p->du_scratch, p->dsrc);
        else
            hipLaunchKernelGGL((interp_gather_kernel<CA>), dim3(sn2_cdiv(n_src, 4)), dim3(256), 0, st, n_src, Rp, S,
                               p->dsrc_stride, (const int*)off, (const int*)cnt, (const int*)inv_row, (const float*)inv_w,
                               (const float*)p->du_scratch, p->dsrc);
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

int check_fp(const sn2_fp* p) {
    if (!p || p->B <= 0 || p->R_per_plot <= 0 || p->S_per_plot <= 0 || !p->src || !p->h || !p->blk.W || !p->blk.b)
        return SN2_EINVAL;
    if ((p->src_stride & 3) || p->src_stride < p->ca || (p->h_stride & 3) || p->h_stride < p->blk.cout) return SN2_EINVAL;
    if (p->cb > 0 && (!p->skip || p->skip_stride < p->cb)) return SN2_EINVAL;
    if (p->cb % 4 == 0 && p->cb > 0 && (p->skip_stride & 3)) return SN2_EINVAL;
    if ((p->knn_idx == nullptr) != (p->knn_w == nullptr)) return SN2_EINVAL;
    if (p->dsrc && p->dsrc_stride < p->ca) return SN2_EINVAL;
    if (p->blk.cin != p->ca + p->cb) return SN2_EINVAL;
    return 0;
}

// the four dense-row blocks of the reference architecture (model/point_net2.py:83,88-93)
#define FP_DISPATCH(FN, ...)                                                                                      \
    do {                                                                                                          \
        const bool knn = p->knn_idx != nullptr;                                                                   \
        if (!knn && p->ca == 32 && p->cb == 3 && p->blk.cout == 64) return FN<32, 3, 64, false>(__VA_ARGS__);     \
        if (knn && p->ca == 64 && p->cb == 32 && p->blk.cout == 64) return FN<64, 32, 64, true>(__VA_ARGS__);     \
        if (knn && p->ca == 64 && p->cb == 16 && p->blk.cout == 34) return FN<64, 16, 34, true>(__VA_ARGS__);     \
        if (knn && p->ca == 34 && p->cb == 8 && p->blk.cout == 34) return FN<34, 8, 34, true>(__VA_ARGS__);       \
        /* the two extra blocks of the 3sa-arch variant: global SA on [x3 | pos3], FP4 on [global | x3] */       \
        if (!knn && p->ca == 64 && p->cb == 3 && p->blk.cout == 64) return FN<64, 3, 64, false>(__VA_ARGS__);     \
        if (knn && p->ca == 64 && p->cb == 64 && p->blk.cout == 64) return FN<64, 64, 64, true>(__VA_ARGS__);     \
        return SN2_ELIMIT;                                                                                        \
    } while (0)

}  // namespace

extern "C" int sn2_interp_index(const int* knn_idx, const float* knn_w, const float* src_pos, int B, int R_per_plot,
                                int S_per_plot, float* ws, void* stream) {
    if (!knn_idx || !knn_w || !ws || B <= 0 || R_per_plot <= 0 || S_per_plot <= 0) return SN2_EINVAL;
    return build_interp_index(knn_idx, knn_w, src_pos, B, R_per_plot, S_per_plot, ws, (hipStream_t)stream);
}

extern "C" int sn2_interp_index_perm(const int* knn_idx, const float* knn_w, const float* src_pos, const int* row_perm, int B,
                                     int R_per_plot, int S_per_plot, float* ws, void* stream) {
    if (!knn_idx || !knn_w || !ws || B <= 0 || R_per_plot <= 0 || S_per_plot <= 0) return SN2_EINVAL;
    return build_interp_index(knn_idx, knn_w, src_pos, B, R_per_plot, S_per_plot, ws, (hipStream_t)stream, row_perm);
}

extern "C" int sn2_interp_index_group(const int* knn_idx, const float* knn_w, const float* src_pos, const int* row_perm, int G, int B,
                                      int R_per_plot, int S_per_plot, float* ws, size_t ws_stride_words, void* stream) {
    if (!knn_idx || !knn_w || !ws || G <= 0 || B <= 0 || R_per_plot <= 0 || S_per_plot <= 0) return SN2_EINVAL;
    if (G > 1 && ws_stride_words < SN2_INTERP_WS_WORDS(B, R_per_plot, S_per_plot)) return SN2_EINVAL;
    return build_interp_index(knn_idx, knn_w, src_pos, B, R_per_plot, S_per_plot, ws, (hipStream_t)stream, row_perm, G, ws_stride_words);
}

extern "C" int sn2_fp_forward(const sn2_fp* p, int training, void* stream) {
    SN2_TRY(check_fp(p));
    if (training < 0 || training > SN2_BN_FROZEN_KEEP) return SN2_EINVAL;
    FP_DISPATCH(fp_forward_t, p, training, (hipStream_t)stream);
}

extern "C" int sn2_fp_backward(const sn2_fp* p, void* stream) {
    SN2_TRY(check_fp(p));
    if (!p->dy || !p->blk.dW || !p->blk.db || !p->blk.dgamma || !p->blk.dbeta) return SN2_EINVAL;
    FP_DISPATCH(fp_backward_t, p, (hipStream_t)stream);
}

// =============================================================================================== head
namespace {

struct HeadOut {
    float y[35];   // fa*f+fc | 1
    float z1[17];  // relu(lin1) | 1
    float p[4];
    float dens;
};

// drop_mask / drop_scale: F.dropout(relu(lin1), p) of model/point_net2.py:142 -- bit j of the row's word set = channel j kept
// and scaled by 1/(1-p); drop_mask == nullptr: no dropout.  z1 holds the values lin2 reads (after the dropout).
// fv: the row's nine float4 quads (36 floats, 34 used)
template <class WP>
__device__ __forceinline__ void head_row_v(const float4 (&fv)[9], WP fa, WP fc, WP W1, WP b1, WP W2, WP b2, size_t r,
                                           HeadOut& o, const int* __restrict__ drop_mask = nullptr, float drop_scale = 1.f) {
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const float4 v = fv[q];
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (4 * q + t < 34) o.y[4 * q + t] = fmaf(fa[4 * q + t], vv[t], fc[4 * q + t]);
    }
    o.y[34] = 1.f;
    // two accumulators per output (even / odd inputs): pairs of consecutive weights and inputs are packed FMAs
    // (v_pk_fma_f32: 2 x the rate of the scalar-operand FMA; one chain per output left 544 of the kernel's 1250 unpacked)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        f32x2 acc = {b1[j], 0.f};
#pragma unroll
        for (int k = 0; k < 34; k += 2) {
            const f32x2 w = {W1[j * 34 + k], W1[j * 34 + k + 1]}, y2 = {o.y[k], o.y[k + 1]};
            acc = __builtin_elementwise_fma(w, y2, acc);
        }
        o.z1[j] = fmaxf(acc[0] + acc[1], 0.f);
    }
    if (drop_mask) {
        const int keep = drop_mask[r];
#pragma unroll
        for (int j = 0; j < 16; ++j) o.z1[j] = ((keep >> j) & 1) ? o.z1[j] * drop_scale : 0.f;
    }
    o.z1[16] = 1.f;
    float s[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        float acc = b2[i];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = fmaf(W2[i * 16 + j], o.z1[j], acc);
        s[i] = acc;
    }
    const float m = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
    float e[4], den = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        e[i] = expf(s[i] - m);
        den += e[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) o.p[i] = e[i] / den;
    o.dens = 1.0f / (1.0f + expf(-s[4]));
}

template <bool BF = false, class WP = cfp>
__device__ __forceinline__ void head_row(const float* __restrict__ f, int f_stride, WP fa, WP fc, WP W1, WP b1, WP W2,
                                         WP b2, size_t r, HeadOut& o, const int* __restrict__ drop_mask = nullptr,
                                         float drop_scale = 1.f) {
    float4 fv[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) fv[q] = row_quad_ld<BF>(f, r, f_stride, q);
    head_row_v(fv, fa, fc, W1, b1, W2, b2, r, o, drop_mask, drop_scale);
}

// a wave re-reads LDS words other lanes of the SAME wave wrote: the LDS executes a wave's instructions in order, the compiler
// must not move the accesses across this point (the regions are reused under different element types)
// (WAVE_LDS_SYNC: defined with fp_src_table_mfma_kernel above)
constexpr int HEAD_T_QUADS = 64 * 9;   // a wave's 64 consecutive rows of 36 floats: 9216 contiguous bytes, nine quads per lane

// The head forward on the matrix cores (rows of exactly 36 floats).  Per wave and turn 64 consecutive rows:
//   global -> LDS (nine fully coalesced float4 loads; one row per lane, 144-byte stride, touched 64 lines per load) -> lin1: A[row][k] = fa_k f + fc_k read back from LDS
//   in the MFMA operand layout (stride 36: conflict-free), B = W1^T in nine registers per lane, bias = accumulator start
//   -> ReLU (+ dropout) in the result layout -> z1 to LDS [64][20] -> lin2 the same way (four k-steps, five live outputs)
//   -> scores to LDS [64][8] -> one row per lane: softmax, sigmoid, two coalesced float4 stores.
// 52 MFMAs per 64 rows instead of 624 FMA instructions per row-lane fed by scalar weight loads.  It is NOT faster than that
// form (24-28 us for 92 MB either way: the kernel streams at 3.3-3.8 TB/s and fp32 MFMA has the packed-VALU rate, 2 x the
// scalar-operand FMA rate); it frees the VALU and scalar cache for whatever runs beside it.  The backward
// (head_bwd_mfma_kernel, round 4) is built the same way.
// workgroups per CU the kernel is compiled for = its register budget.  At 4 (128 VGPRs, one spilled) the compiler issued the nine
// row loads of a turn ONE BY ONE, each behind an s_waitcnt vmcnt(0) of its own (every load into the same four registers): 23.4 us
// at config 2; at 3 / 2 the loads are in flight together: 22.3 / 22.1 us (scripts/time_head_fwd.py)
#ifndef SN2_HF_OCC
#define SN2_HF_OCC 3
#endif
template <bool BF>
__global__ __launch_bounds__(256, SN2_HF_OCC) void head_fwd_mfma_kernel(int R, const float* __restrict__ f, const float* __restrict__ fa,
                                                            const float* __restrict__ fc, const float* __restrict__ W1,
                                                            const float* __restrict__ b1, const float* __restrict__ W2,
                                                            const float* __restrict__ b2, float* __restrict__ cov,
                                                            float* __restrict__ proba, const int* __restrict__ drop_mask,
                                                            float drop_scale, float4* __restrict__ zero4, long nzero4) {
    __shared__ float4 s_t[4 * HEAD_T_QUADS];
    // sn2_head.zero_fill: the backward pass's accumulate-into arena, cleared here -- 5 MB of stores beside 92 MB of rows --
    // instead of by a launch of its own in front of the backward pass (4.9 us: the floor of any launch on this chip)
    if (zero4)
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nzero4; i += (long)gridDim.x * 256) zero4[i] = float4{0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4* st4 = s_t + wave * HEAD_T_QUADS;
    float* st = reinterpret_cast<float*>(st4);
    float* zt = st;                    // [64][20] after lin1 has read the rows
    float* sc = st + 64 * 20;          // [64][8]
    const int n = lane & 15, kq = lane >> 4;
    float w1[9], ak[9], ck[9], w2[4];
#pragma unroll
    for (int ks = 0; ks < 9; ++ks) {
        const int k = 4 * ks + kq;
        w1[ks] = k < 34 ? W1[n * 34 + k] : 0.f;
        ak[ks] = k < 34 ? fa[k] : 0.f;
        ck[ks] = k < 34 ? fc[k] : 0.f;
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) w2[ks] = n < 5 ? W2[n * 16 + 4 * ks + kq] : 0.f;
    const float bias1 = b1[n], bias2 = n < 5 ? b2[n] : 0.f;
    for (long r0 = ((long)blockIdx.x * 4 + wave) * 64; r0 < R; r0 += (long)gridDim.x * 256) {
        {
            const long lim = (R - r0) * 9;
            float4 t[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int e = lane + 64 * k;
                // quad e of the wave's 64 consecutive rows: 16 bytes of fp32 or 8 bytes of bfloat16, contiguous either way
                // (unconditional loads from a clamped address: a load under a divergent branch is waited for at the join)
                const float4 v = row_quad_ld<BF>(f, (size_t)r0, 36, e < lim ? e : 0);
                t[k] = e < lim ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) st4[lane + 64 * k] = t[k];
        }
        WAVE_LDS_SYNC();
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[t] = f32x4{bias1, bias1, bias1, bias1};
#pragma unroll
            for (int ks = 0; ks < 9; ++ks) {
                const float v = st[(16 * t + n) * 36 + 4 * ks + kq];
                const float a = (4 * ks + kq < 34) ? fmaf(ak[ks], v, ck[ks]) : 0.f;
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w1[ks], acc[t], 0, 0, 0);
            }
        }
        WAVE_LDS_SYNC();
        // acc[t][j]: row 16 t + 4 kq + j, hidden channel n
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = 16 * t + 4 * kq + j;
                float z = fmaxf(acc[t][j], 0.f);
                if (drop_mask) {
                    const long r = r0 + row;
                    const int keep = drop_mask[r < R ? r : R - 1];
                    z = ((keep >> n) & 1) ? z * drop_scale : 0.f;
                }
                zt[row * 20 + n] = z;
            }
        WAVE_LDS_SYNC();
        f32x4 s2[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            s2[t] = f32x4{bias2, bias2, bias2, bias2};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                s2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(zt[(16 * t + n) * 20 + 4 * ks + kq], w2[ks], s2[t], 0, 0, 0);
        }
        if (n < 8) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) sc[(16 * t + 4 * kq + j) * 8 + n] = s2[t][j];
        }
        WAVE_LDS_SYNC();
        const long r = r0 + lane;
        const float4 s03 = *reinterpret_cast<const float4*>(&sc[lane * 8]);
        const float s4 = sc[lane * 8 + 4];
        WAVE_LDS_SYNC();
        const float sv[4] = {s03.x, s03.y, s03.z, s03.w};
        const float m = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
        float e[4], den = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            e[i] = expf(sv[i] - m);
            den += e[i];
        }
        float pr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pr[i] = e[i] / den;
        const float dens = 1.0f / (1.0f + expf(-s4));
        if (r < R) {
            reinterpret_cast<float4*>(proba)[r] = make_float4(pr[0], pr[1], pr[2], pr[3]);
            reinterpret_cast<float4*>(cov)[r] = make_float4(pr[0] * dens, pr[1] * dens, pr[2] * dens, pr[3] * dens);
        }
    }
}

// EVAL: the per-point layer FP1 (source-side form: fp_fwd_rows_kernel's row side) and the head in ONE kernel.  An eval pass keeps
// nothing for a backward, so the 144-byte rows of h1 need not exist: a wave computes 63 consecutive rows (nine groups of seven,
// nine lanes per row as in fp_fwd_rows_kernel) straight into the LDS tile head_fwd_mfma_kernel reads its rows from, and runs
// that kernel's turn on it (row 63 of the tile is padding).  Same operations in the same order as the two kernels: the same
// bits.  Saves the write and the read of h1 (parcel inference: 740 MB per launch of 256 plots) and a launch.
template <int CA, int CB, int CO>
__global__ __launch_bounds__(256, 2) void fp_head_eval_kernel(int R, int R_per_plot, int S_per_plot, int skip_stride,
                                                              const float* __restrict__ T, const int* __restrict__ knn_idx,
                                                              const float* __restrict__ knn_w, const float* __restrict__ skip,
                                                              const float* __restrict__ Wg, const float* __restrict__ biasg,
                                                              const float* __restrict__ fa, const float* __restrict__ fc,
                                                              const float* __restrict__ W1, const float* __restrict__ b1,
                                                              const float* __restrict__ W2, const float* __restrict__ b2,
                                                              float* __restrict__ cov, float* __restrict__ proba) {
    constexpr int CI = CA + CB, QH = (CO + 3) / 4, HS = 4 * QH, G = 64 / QH, QB = CB / 4, U = 3, ROWS = G * 9;
    static_assert(CO == 34 && HS == 36 && G == 7 && ROWS == 63, "the head reads rows of 36 floats, 63 per turn");
    static_assert(CB > 0 && CB % 4 == 0, "skip quads");
    __shared__ float4 s_t[4 * HEAD_T_QUADS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4* st4 = s_t + wave * HEAD_T_QUADS;
    float* st = reinterpret_cast<float*>(st4);
    float* zt = st;                    // [64][20] after lin1 has read the rows
    float* sc = st + 64 * 20;          // [64][8]
    const int q = lane % QH, g = lane / QH;
    const bool on = lane < G * QH;
    const int n = lane & 15, kq = lane >> 4;
    // the row side's weights (fp_fwd_rows_kernel) and the head's (head_fwd_mfma_kernel), in registers for the whole kernel
    float wB[4][CB], b4[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int o = 4 * q + t;
        b4[t] = o < CO ? biasg[o] : 0.f;
#pragma unroll
        for (int k = 0; k < CB; ++k) wB[t][k] = o < CO ? Wg[o * CI + CA + k] : 0.f;
    }
    float w1[9], ak[9], ck[9], w2[4];
#pragma unroll
    for (int ks = 0; ks < 9; ++ks) {
        const int k = 4 * ks + kq;
        w1[ks] = k < 34 ? W1[n * 34 + k] : 0.f;
        ak[ks] = k < 34 ? fa[k] : 0.f;
        ck[ks] = k < 34 ? fc[k] : 0.f;
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) w2[ks] = n < 5 ? W2[n * 16 + 4 * ks + kq] : 0.f;
    const float bias1 = b1[n], bias2 = n < 5 ? b2[n] : 0.f;
    const long n_turns = ((long)R + ROWS - 1) / ROWS;
    for (long turn = (long)blockIdx.x * 4 + wave; turn < n_turns; turn += (long)gridDim.x * 4) {
        const long r0 = turn * ROWS;
        // ---- FP1, rows r0 .. r0 + 62 -> the tile (a group of seven rows per step, U groups' loads in flight)
#pragma unroll 1
        for (int g0 = 0; g0 < 9; g0 += U) {
            FpRowIn<QB> in[U];
            float4 ta[U][3];
#pragma unroll
            for (int u = 0; u < U; ++u) in[u] = fp_row_in<QB>(r0 + (long)(g0 + u) * G + g, on, R, knn_idx, knn_w, skip, skip_stride);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned base = (in[u].rr / (unsigned)R_per_plot) * (unsigned)S_per_plot;
                ta[u][0] = reinterpret_cast<const float4*>(T + (size_t)(base + in[u].i0) * HS)[q];
                ta[u][1] = reinterpret_cast<const float4*>(T + (size_t)(base + in[u].i1) * HS)[q];
                ta[u][2] = reinterpret_cast<const float4*>(T + (size_t)(base + in[u].i2) * HS)[q];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float w0 = in[u].w0, wa = in[u].w1, wb = in[u].w2;
                const float inv = 1.0f / ((w0 + wa) + wb);
                const float4 a = ta[u][0], b = ta[u][1], c = ta[u][2];
                // (interp_bias: the one spelled-out order of operations of the row kernels -- the same bits as the separate pass)
                float v[4] = {interp_bias(a.x, b.x, c.x, w0, wa, wb, inv, b4[0]), interp_bias(a.y, b.y, c.y, w0, wa, wb, inv, b4[1]),
                              interp_bias(a.z, b.z, c.z, w0, wa, wb, inv, b4[2]), interp_bias(a.w, b.w, c.w, w0, wa, wb, inv, b4[3])};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float acc = v[t];
#pragma unroll
                    for (int b2q = 0; b2q < QB; ++b2q) {
                        acc = fmaf(wB[t][4 * b2q + 0], in[u].sk[b2q].x, acc);
                        acc = fmaf(wB[t][4 * b2q + 1], in[u].sk[b2q].y, acc);
                        acc = fmaf(wB[t][4 * b2q + 2], in[u].sk[b2q].z, acc);
                        acc = fmaf(wB[t][4 * b2q + 3], in[u].sk[b2q].w, acc);
                    }
                    v[t] = (in[u].valid && 4 * q + t < CO) ? fmaxf(acc, 0.f) : 0.f;
                }
                if (on) st4[((g0 + u) * G + g) * QH + q] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        if (lane < QH) st4[ROWS * QH + lane] = make_float4(0.f, 0.f, 0.f, 0.f);      // row 63: padding
        WAVE_LDS_SYNC();
        // ---- the head on the tile: head_fwd_mfma_kernel's turn
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[t] = f32x4{bias1, bias1, bias1, bias1};
#pragma unroll
            for (int ks = 0; ks < 9; ++ks) {
                const float v = st[(16 * t + n) * 36 + 4 * ks + kq];
                const float a = (4 * ks + kq < 34) ? fmaf(ak[ks], v, ck[ks]) : 0.f;
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w1[ks], acc[t], 0, 0, 0);
            }
        }
        WAVE_LDS_SYNC();
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) zt[(16 * t + 4 * kq + j) * 20 + n] = fmaxf(acc[t][j], 0.f);
        WAVE_LDS_SYNC();
        f32x4 s2[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            s2[t] = f32x4{bias2, bias2, bias2, bias2};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                s2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(zt[(16 * t + n) * 20 + 4 * ks + kq], w2[ks], s2[t], 0, 0, 0);
        }
        if (n < 8) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) sc[(16 * t + 4 * kq + j) * 8 + n] = s2[t][j];
        }
        WAVE_LDS_SYNC();
        const long r = r0 + lane;
        const float4 s03 = *reinterpret_cast<const float4*>(&sc[lane * 8]);
        const float s4 = sc[lane * 8 + 4];
        WAVE_LDS_SYNC();
        const float sv[4] = {s03.x, s03.y, s03.z, s03.w};
        const float m = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
        float e[4], den = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            e[i] = expf(sv[i] - m);
            den += e[i];
        }
        float pr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pr[i] = e[i] / den;
        const float dens = 1.0f / (1.0f + expf(-s4));
        if (lane < ROWS && r < R) {
            reinterpret_cast<float4*>(proba)[r] = make_float4(pr[0], pr[1], pr[2], pr[3]);
            reinterpret_cast<float4*>(cov)[r] = make_float4(pr[0] * dens, pr[1] * dens, pr[2] * dens, pr[3] * dens);
        }
    }
}

// The same kernel with the row side's INPUT STREAM decoupled from the lanes that consume it, as fp_fwd_rows2_kernel (round 5).  The
// first form's nine lanes of a row load the row's 3-NN entry and skip columns themselves, then gather, then compute, three groups
// of seven rows at a time: six memory round trips per 63-row turn, one after the other, in front of the head's arithmetic.  Here
// a wave fetches a whole turn's 189 indices, 189 weights and 63 x QB skip quads with one element per lane and load (eight loads),
// ONE TURN AHEAD (the loads are issued in front of the head phase of the turn before), hands them to the (row, quad) lanes
// through LDS, and asks for a batch's table rows before it computes the batch before: one round trip per turn is left in the
// open.  Turns dealt XCD-aware (row_iters: an XCD's L2 holds its own plots' table rows).  Same operations in the same order:
// the same bits as the first form (sn2_debug_fp_rows_form(0)) and as the two separate kernels.
template <int CA, int CB, int CO>
__global__ __launch_bounds__(256, 2) void fp_head_eval2_kernel(int R, int R_per_plot, int S_per_plot, int skip_stride,
                                                               const float* __restrict__ T, const int* __restrict__ knn_idx,
                                                               const float* __restrict__ knn_w, const float* __restrict__ skip,
                                                               const float* __restrict__ Wg, const float* __restrict__ biasg,
                                                               const float* __restrict__ fa, const float* __restrict__ fc,
                                                               const float* __restrict__ W1, const float* __restrict__ b1,
                                                               const float* __restrict__ W2, const float* __restrict__ b2,
                                                               float* __restrict__ cov, float* __restrict__ proba) {
    constexpr int CI = CA + CB, QH = (CO + 3) / 4, HS = 4 * QH, G = 64 / QH, QB = CB / 4, U = 3, ROWS = G * 9;
    static_assert(CO == 34 && HS == 36 && G == 7 && ROWS == 63, "the head reads rows of 36 floats, 63 per turn");
    static_assert(CB > 0 && CB % 4 == 0 && QB == 2, "two skip quads per row: 126 quads per turn = two per lane");
    constexpr int XW = 3 * ROWS + 3 * ROWS + 4 * ROWS * QB + 2;                // idx | w | skip quads (16-byte aligned: 378 % 4 = 2 -> +2)
    constexpr int XO_W = 3 * ROWS, XO_S = 6 * ROWS + 2;
    static_assert(XO_S % 4 == 0, "the skip quads start 16-byte aligned");
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    __shared__ float4 s_t[4 * HEAD_T_QUADS];
    __shared__ __attribute__((aligned(16))) float s_x[4][(XW + 3) / 4 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4* st4 = s_t + wave * HEAD_T_QUADS;
    float* st = reinterpret_cast<float*>(st4);
    float* zt = st;                    // [64][20] after lin1 has read the rows
    float* sc = st + 64 * 20;          // [64][8]
    float* xw = s_x[wave];
    const int q = lane % QH, g = lane / QH;
    const bool on = lane < G * QH;
    const int n = lane & 15, kq = lane >> 4;
    float wB[4][CB], b4[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int o = 4 * q + t;
        b4[t] = o < CO ? biasg[o] : 0.f;
#pragma unroll
        for (int k = 0; k < CB; ++k) wB[t][k] = o < CO ? Wg[o * CI + CA + k] : 0.f;
    }
    float w1[9], ak[9], ck[9], w2[4];
#pragma unroll
    for (int ks = 0; ks < 9; ++ks) {
        const int k = 4 * ks + kq;
        w1[ks] = k < 34 ? W1[n * 34 + k] : 0.f;
        ak[ks] = k < 34 ? fa[k] : 0.f;
        ck[ks] = k < 34 ? fc[k] : 0.f;
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) w2[ks] = n < 5 ? W2[n * 16 + 4 * ks + kq] : 0.f;
    const float bias1 = b1[n], bias2 = n < 5 ? b2[n] : 0.f;
    const int n_turns = (R + ROWS - 1) / ROWS;
    const RowIters ri = row_iters(n_turns, wave);
    // ---- a turn's inputs, one element per lane and load, unconditional from clamped addresses
    int p_idx[3];
    float p_w[3];
    f32x4v p_sk[2];
    const int last_e = 3 * R - 1;
    auto fetch = [&](int turn) {
        const int tc = turn < ri.it_hi ? turn : ri.it_hi - 1;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int el = lane + 64 * k < 3 * ROWS ? lane + 64 * k : 3 * ROWS - 1;
            const int e0 = tc * (3 * ROWS) + el, e = e0 < last_e ? e0 : last_e;
            p_idx[k] = knn_idx[e];
            p_w[k] = knn_w[e];
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int el = lane + 64 * k < ROWS * QB ? lane + 64 * k : ROWS * QB - 1;
            const int r0 = tc * ROWS + el / QB, r = r0 < R ? r0 : R - 1;
            p_sk[k] = reinterpret_cast<const f32x4v*>(skip + (size_t)r * skip_stride)[el % QB];
        }
    };
    auto hand_over = [&]() {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int el = lane + 64 * k < 3 * ROWS ? lane + 64 * k : 3 * ROWS - 1;     // (the surplus lanes rewrite the last element)
            xw[el] = __int_as_float(p_idx[k]);
            xw[XO_W + el] = p_w[k];
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int el = lane + 64 * k < ROWS * QB ? lane + 64 * k : ROWS * QB - 1;
            reinterpret_cast<f32x4v*>(xw + XO_S)[el] = p_sk[k];
        }
        WAVE_LDS_SYNC();
    };
    // the table rows of batch `bt` (groups 3 bt .. 3 bt + 2) of the turn
    auto gathers = [&](int turn, int bt, float4 (&ta)[U][3]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rl = on ? (3 * bt + u) * G + g : 0;                     // row of the turn
            const int row = turn * ROWS + rl;
            const bool valid = on && row < R;
            const unsigned rr = valid ? (unsigned)row : 0u;
            const int i0 = valid ? __float_as_int(xw[3 * rl + 0]) : 0, i1 = valid ? __float_as_int(xw[3 * rl + 1]) : 0,
                      i2 = valid ? __float_as_int(xw[3 * rl + 2]) : 0;
            const unsigned base = (rr / (unsigned)R_per_plot) * (unsigned)S_per_plot;
            ta[u][0] = reinterpret_cast<const float4*>(T + (size_t)(base + i0) * HS)[q];
            ta[u][1] = reinterpret_cast<const float4*>(T + (size_t)(base + i1) * HS)[q];
            ta[u][2] = reinterpret_cast<const float4*>(T + (size_t)(base + i2) * HS)[q];
        }
    };
    auto rows_of = [&](int turn, int bt, const float4 (&ta)[U][3]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rl = on ? (3 * bt + u) * G + g : 0;
            const int row = turn * ROWS + rl;
            const bool valid = on && row < R;
            const float w0 = xw[XO_W + 3 * rl + 0], wa = xw[XO_W + 3 * rl + 1], wb = xw[XO_W + 3 * rl + 2];
            float4 sk[QB];
#pragma unroll
            for (int b = 0; b < QB; ++b) sk[b] = reinterpret_cast<const float4*>(xw + XO_S)[rl * QB + b];
            const float inv = 1.0f / ((w0 + wa) + wb);
            const float4 a = ta[u][0], b = ta[u][1], c = ta[u][2];
            float v[4] = {interp_bias(a.x, b.x, c.x, w0, wa, wb, inv, b4[0]), interp_bias(a.y, b.y, c.y, w0, wa, wb, inv, b4[1]),
                          interp_bias(a.z, b.z, c.z, w0, wa, wb, inv, b4[2]), interp_bias(a.w, b.w, c.w, w0, wa, wb, inv, b4[3])};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float acc = v[t];
#pragma unroll
                for (int b2q = 0; b2q < QB; ++b2q) {
                    acc = fmaf(wB[t][4 * b2q + 0], sk[b2q].x, acc);
                    acc = fmaf(wB[t][4 * b2q + 1], sk[b2q].y, acc);
                    acc = fmaf(wB[t][4 * b2q + 2], sk[b2q].z, acc);
                    acc = fmaf(wB[t][4 * b2q + 3], sk[b2q].w, acc);
                }
                v[t] = (valid && 4 * q + t < CO) ? fmaxf(acc, 0.f) : 0.f;
            }
            if (on) st4[((3 * bt + u) * G + g) * QH + q] = make_float4(v[0], v[1], v[2], v[3]);
        }
    };
    if (ri.it0 < ri.it_hi) fetch(ri.it0);
    for (int turn = ri.it0; turn < ri.it_hi; turn += ri.stride) {
        const long r0 = (long)turn * ROWS;
        // ---- FP1, rows r0 .. r0 + 62 -> the tile
        hand_over();
        float4 ta_a[U][3], ta_b[U][3];
        gathers(turn, 0, ta_a);
        gathers(turn, 1, ta_b);
        rows_of(turn, 0, ta_a);
        gathers(turn, 2, ta_a);
        rows_of(turn, 1, ta_b);
        rows_of(turn, 2, ta_a);
        if (lane < QH) st4[ROWS * QH + lane] = make_float4(0.f, 0.f, 0.f, 0.f);      // row 63: padding
        WAVE_LDS_SYNC();
        fetch(turn + ri.stride);                              // the NEXT turn's inputs travel while the head runs (clamped past the end)
        // ---- the head on the tile: head_fwd_mfma_kernel's turn
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[t] = f32x4{bias1, bias1, bias1, bias1};
#pragma unroll
            for (int ks = 0; ks < 9; ++ks) {
                const float v = st[(16 * t + n) * 36 + 4 * ks + kq];
                const float a = (4 * ks + kq < 34) ? fmaf(ak[ks], v, ck[ks]) : 0.f;
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w1[ks], acc[t], 0, 0, 0);
            }
        }
        WAVE_LDS_SYNC();
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) zt[(16 * t + 4 * kq + j) * 20 + n] = fmaxf(acc[t][j], 0.f);
        WAVE_LDS_SYNC();
        f32x4 s2[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            s2[t] = f32x4{bias2, bias2, bias2, bias2};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                s2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(zt[(16 * t + n) * 20 + 4 * ks + kq], w2[ks], s2[t], 0, 0, 0);
        }
        if (n < 8) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) sc[(16 * t + 4 * kq + j) * 8 + n] = s2[t][j];
        }
        WAVE_LDS_SYNC();
        const long r = r0 + lane;
        const float4 s03 = *reinterpret_cast<const float4*>(&sc[lane * 8]);
        const float s4 = sc[lane * 8 + 4];
        WAVE_LDS_SYNC();
        const float sv[4] = {s03.x, s03.y, s03.z, s03.w};
        const float m = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
        float e[4], den = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            e[i] = expf(sv[i] - m);
            den += e[i];
        }
        float pr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pr[i] = e[i] / den;
        const float dens = 1.0f / (1.0f + expf(-s4));
        if (lane < ROWS && r < R) {
            reinterpret_cast<float4*>(proba)[r] = make_float4(pr[0], pr[1], pr[2], pr[3]);
            reinterpret_cast<float4*>(cov)[r] = make_float4(pr[0] * dens, pr[1] * dens, pr[2] * dens, pr[3] * dens);
        }
    }
}

// The head backward on the matrix cores (round 4; rows of exactly 36 floats).  Its predecessor gave every lane one row: ~1900
// FMA instructions per row, a third of them on the two weight-gradient outer products through an LDS transposition, behind
// 144-byte strided row loads and stores (64 lines per instruction): 57 us for 168 MB.  Here a wave takes 64 consecutive rows
// per turn, as head_fwd_mfma_kernel does, and every contraction is a chain of 16x16x4 tiles fed from three LDS regions:
//   st [64][36]  y = fa f + fc (applied once, on the way in from the coalesced float4 loads; column 34 := 1 -- the bias
//                column of [y | 1] --, column 35 := 0); at the end of the turn the d rows, stored coalesced the same way
//   zt [64][20]  z1 = dropout(relu(lin1)) for lin2 and dW2, then d pre-activation of lin1 for dW1 and the d rows
//   sc [64][12]  the five scores, then their gradients (softmax / sigmoid backward with one row per lane), columns 5..7 zero
//   lin1 36 + lin2 16 + dW2 16 + d pre 8 + dW1 48 + d rows 48 = 172 tiles per 64 rows; dW1 | db1 and dW2 stay in
//   accumulators for the whole kernel, db2 is a per-lane sum; the next turn's rows and gradients are requested before the
//   current turn's arithmetic.
// LDS traffic, not the matrix cores, is what the layout is about (ds_read_b32 / ds_write_b32: 32 banks, lanes 0..31 and
// 32..63 apart; ds_read_b64: 64 banks):
//   * a tile's k index is free as long as both operands agree.  Where a lane's operand runs along a ROW (lin1, lin2, d pre,
//     d rows: lane (n, kq) = row n of the tile) steps 2p and 2p+1 take columns 8p + 2kq and 8p + 2kq + 1: ONE ds_read_b64
//     per two tiles, conflict-free at the even strides 36 / 20 / 12 (the plain 4 ks + kq columns are 2-way, 4-way at 12);
//   * where it runs along a COLUMN (dW2, dW1: lane (n, kq) = column n) step s takes rows 16 (s / 4) + s % 4 + 4 kq: lanes
//     kq and kq + 1 are four rows = 16 banks apart at every stride.
//   * the per-lane weight operands (30 floats) live in a table of float4 per (quad, lane), read phase by phase: held in
//     registers for the whole loop they left no room beside the prefetched rows (spills, and a spill's reload waits for the
//     prefetch with it).
// Operands are read in batches in front of their tiles (sched_barrier: left alone the scheduler puts every read right in front
// of its tile and pays the LDS latency 170 times per turn).
// Same sums as the row-per-lane form up to fp32 re-association.
#ifdef SN2_HB_STAMPS
// diagnostic build only (never shipped): phase stamps of wave 0 of one workgroup of head_bwd_mfma_kernel, second turn
__device__ unsigned long long g_hb_dbg[16];
extern "C" int sn2_debug_hb_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hb_dbg), sizeof(g_hb_dbg));
}
#define HSTAMP(i)                                                                                   \
    if (blockIdx.x == 37 && threadIdx.x == 0 && turn_no == 1) {                                     \
        unsigned long long t_;                                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                   \
        g_hb_dbg[i] = t_;                                                                           \
    }
#else
#define HSTAMP(i)
#endif
// one ds_read_b64 (left to the compiler two of them at nearby offsets become a ds_read2_b64: banked like ds_read_b32, 4 x the
// cycles).  The compiler does not count this read: the caller waits (HB_WAIT_B64) before the first use.
typedef float hb_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ hb_f32x2 lds_read_b64(const float* p) {
    hb_f32x2 v;
    const unsigned a = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)p;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a) : "memory");
    return v;
}
constexpr int HB_ST = 64 * 36, HB_ZT = 64 * 20, HB_SC = 64 * 12;
constexpr int HB_WAVE_FLOATS = HB_ST + HB_ZT + HB_SC + 64;      // + the rows' dropout words
#ifndef SN2_HB_DIAG
#define SN2_HB_DIAG 0      /* timing experiments (scripts/time_head_bwd.py): 1 = no d-row stores, 2 = no arithmetic (rows in, rows out) */
#endif
constexpr int HB_DIAG = SN2_HB_DIAG;
constexpr int HB_CQ = 8;                                        // float4 quads of per-lane weight operands (one table per workgroup)
constexpr int HB_TAB_FLOATS = HB_CQ * 64 * 4 + 18 * 4;          // + fa, fc as nine quads each
constexpr int HB_RED = 16 * 35 + 5 * 16 + 5;                    // a wave's weight-gradient image: dW1 | db1, dW2, db2
template <bool BF>
__global__ __launch_bounds__(256, 2) void head_bwd_mfma_kernel(int R, const float* __restrict__ f, const float* __restrict__ fa,
                                                            const float* __restrict__ fc, const float* __restrict__ W1,
                                                            const float* __restrict__ b1, const float* __restrict__ W2,
                                                            const float* __restrict__ b2, const float* __restrict__ dcov,
                                                            const float* __restrict__ dproba, float* __restrict__ dy,
                                                            float* __restrict__ dW1, float* __restrict__ db1,
                                                            float* __restrict__ dW2, float* __restrict__ db2, int rep_k,
                                                            int rep_stride, const int* __restrict__ drop_mask, float drop_scale) {
    typedef hb_f32x2 f32x2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* st = smem + wave * HB_WAVE_FLOATS;
    float4* st4 = reinterpret_cast<float4*>(st);
    float* zt = st + HB_ST;
    float* sc = zt + HB_ZT;
    int* mk = reinterpret_cast<int*>(sc + HB_SC);
    const int n = lane & 15, kq = lane >> 4;
    float* tab = smem + 4 * HB_WAVE_FLOATS;
    // ---- the weight operands of lane (n, kq) -> the table
    if (wave == 0) {
        float c[4 * HB_CQ];
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {                        // lin1: B[k][n] = W1[n][k], k = 8 pp + 2 kq (+ 1)
            c[2 * pp] = W1[n * 34 + 8 * pp + 2 * kq];
            c[2 * pp + 1] = W1[n * 34 + 8 * pp + 2 * kq + 1];
        }
        c[8] = kq < 2 ? W1[n * 34 + 32 + kq] : 0.f;             // ... and the ninth step: k = 32 + kq
        c[9] = b1[n];
        c[10] = n < 5 ? b2[n] : 0.f;
        c[11] = 0.f;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {                        // lin2: B[k][n] = W2[n][k]
            c[12 + 2 * pp] = n < 5 ? W2[n * 16 + 8 * pp + 2 * kq] : 0.f;
            c[13 + 2 * pp] = n < 5 ? W2[n * 16 + 8 * pp + 2 * kq + 1] : 0.f;
        }
        c[16] = 2 * kq < 5 ? W2[(2 * kq) * 16 + n] : 0.f;       // d pre: B[i][j] = W2[i][j], i = 2 kq (+ 1)
        c[17] = 2 * kq + 1 < 5 ? W2[(2 * kq + 1) * 16 + n] : 0.f;
        c[18] = c[19] = 0.f;
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)                          // d rows: B[j][col] = W1[j][col], j = 8 pp + 2 kq (+ 1)
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int col = 16 * ct + n;
                c[20 + 4 * ct + 2 * pp] = col < 34 ? W1[(8 * pp + 2 * kq) * 34 + col] : 0.f;
                c[21 + 4 * ct + 2 * pp] = col < 34 ? W1[(8 * pp + 2 * kq + 1) * 34 + col] : 0.f;
            }
        float4* ct4 = reinterpret_cast<float4*>(tab);
#pragma unroll
        for (int q = 0; q < HB_CQ; ++q) ct4[q * 64 + lane] = make_float4(c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]);
        if (lane < 36) {                                        // [y | 1 | 0]: fa = 0 and fc = 1 in column 34, both 0 in column 35
            tab[HB_CQ * 256 + lane] = lane < 34 ? fa[lane] : 0.f;
            tab[HB_CQ * 256 + 36 + lane] = lane < 34 ? fc[lane] : (lane == 34 ? 1.f : 0.f);
        }
    }
    __syncthreads();
    const float4* ctab = reinterpret_cast<const float4*>(tab) + lane;
    const float4* fa4 = reinterpret_cast<const float4*>(tab + HB_CQ * 256);
    const float4* fc4 = fa4 + 9;
    const int lane9 = lane % 9;
    const int nn = n < 8 ? n : 5;                               // lanes n >= 8 of dW2's A operand read a zero column
    f32x4 dw1acc[3], dw2acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) dw1acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dsum[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const long stride = (long)gridDim.x * 256;
    long r0 = ((long)blockIdx.x * 4 + wave) * 64;
    float4 t[9], gc = make_float4(0.f, 0.f, 0.f, 0.f), gp = gc;
    int keep = 0xFFFF;
    // (a gradient or mask that is absent is loaded from the rows instead -- 16 valid bytes per row -- and dropped at its use:
    // a load under a branch, even a uniform one, is waited for where the branch joins, which would end the prefetch; for the
    // same reason the rows past the end are loaded from a clamped address: finite values whose d scores are zero)
    const float4* gcp = reinterpret_cast<const float4*>(dcov ? dcov : f);
    const float4* gpp = reinterpret_cast<const float4*>(dproba ? dproba : f);
    const int* kp = drop_mask ? drop_mask : reinterpret_cast<const int*>(f);
    auto request = [&](long q0) {               // the rows of a turn, one row's incoming gradients and dropout word per lane
        const long lim = (R - q0) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int e = lane + 64 * k;
            t[k] = row_quad_ld<BF>(f, (size_t)q0, 36, e < lim ? e : 0);
        }
        const long r = q0 + lane;
        const size_t rr = r < R ? (size_t)r : 0;
        gc = gcp[rr];
        gp = gpp[rr];
        keep = kp[rr];
    };
    if (r0 < R) request(r0);
    int turn_no = -1;
    for (; r0 < R; r0 += stride) {
        ++turn_no;
        HSTAMP(0)
        // ---- quad e = lane + 64 k of the tile is quad (lane + k) % 9 of its row
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int q9 = lane9 + k < 9 ? lane9 + k : lane9 + k - 9;
            const float4 a4 = fa4[q9], c4 = fc4[q9], v = t[k];
            float4 y = make_float4(fmaf(a4.x, v.x, c4.x), fmaf(a4.y, v.y, c4.y), fmaf(a4.z, v.z, c4.z), fmaf(a4.w, v.w, c4.w));
            if (q9 == 8) y.z = 1.f, y.w = 0.f;                  // (whatever the rows' padding holds)
            st4[lane + 64 * k] = y;
        }
        if (drop_mask) mk[lane] = keep;
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 gcv = dcov ? gc : zero4, gpv = dproba ? gp : zero4;
        const bool valid = r0 + lane < R;
        WAVE_LDS_SYNC();
        HSTAMP(1)
        if (!(HB_DIAG & 2)) {
        // ---- lin1, ReLU, dropout (z[tt][j]: row 16 tt + 4 kq + j, hidden channel n)
#pragma unroll
        for (int tp = 0; tp < 4; tp += 2) {
            f32x2 av[2][4];
            float as[2];
            f32x4 z[2];
            const float4 q0 = ctab[0], q1 = ctab[64], q2 = ctab[128];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
                for (int pp = 0; pp < 4; ++pp)
                    av[tt][pp] = lds_read_b64(&st[(16 * (tp + tt) + n) * 36 + 8 * pp + 2 * kq]);
                as[tt] = st[(16 * (tp + tt) + n) * 36 + 32 + kq];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av[0][0]), "+v"(av[0][1]), "+v"(av[0][2]), "+v"(av[0][3]), "+v"(av[1][0]), "+v"(av[1][1]), "+v"(av[1][2]), "+v"(av[1][3]) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            const float w1p[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) z[tt] = f32x4{q2.y, q2.y, q2.y, q2.y};
#pragma unroll
            for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
                        z[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tt][pp][h], w1p[2 * pp + h], z[tt], 0, 0, 0);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) z[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[tt], q2.x, z[tt], 0, 0, 0);
            int kw[2][4];
            if (drop_mask) {
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) kw[tt][j] = mk[16 * (tp + tt) + 4 * kq + j];
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float zz = fmaxf(z[tt][j], 0.f);
                    if (drop_mask) zz = ((kw[tt][j] >> n) & 1) ? zz * drop_scale : 0.f;
                    zt[(16 * (tp + tt) + 4 * kq + j) * 20 + n] = zz;
                }
        }
        WAVE_LDS_SYNC();
        HSTAMP(2)
        // ---- lin2 -> scores
        {
            f32x2 zv[4][2];
            f32x4 s2[4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) zv[tt][pp] = lds_read_b64(&zt[(16 * tt + n) * 20 + 8 * pp + 2 * kq]);
            const float4 w2q = ctab[3 * 64];
            const float bias2 = ctab[2 * 64].z;
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(zv[0][0]), "+v"(zv[0][1]), "+v"(zv[1][0]), "+v"(zv[1][1]), "+v"(zv[2][0]), "+v"(zv[2][1]), "+v"(zv[3][0]), "+v"(zv[3][1]) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            const float w2p[4] = {w2q.x, w2q.y, w2q.z, w2q.w};
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) s2[tt] = f32x4{bias2, bias2, bias2, bias2};
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
                        s2[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(zv[tt][pp][h], w2p[2 * pp + h], s2[tt], 0, 0, 0);
            if (n < 8) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) sc[(16 * tt + 4 * kq + j) * 12 + n] = s2[tt][j];
            }
        }
        WAVE_LDS_SYNC();
        HSTAMP(3)
        // ---- one row per lane: softmax, sigmoid and their backward -> d scores
        {
            const float4 s03 = *reinterpret_cast<const float4*>(&sc[lane * 12]);
            const float s4 = sc[lane * 12 + 4];
            const float sv[4] = {s03.x, s03.y, s03.z, s03.w};
            const float m = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
            float e[4], den = 0.f, pr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                e[i] = expf(sv[i] - m);
                den += e[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) pr[i] = e[i] / den;
            const float dens = 1.0f / (1.0f + expf(-s4));
            const float gcs[4] = {gcv.x, gcv.y, gcv.z, gcv.w}, gps[4] = {gpv.x, gpv.y, gpv.z, gpv.w};
            float dp[4], dot = 0.f, ddens = 0.f, ds[5];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dp[i] = fmaf(gcs[i], dens, gps[i]);
                ddens = fmaf(gcs[i], pr[i], ddens);
                dot = fmaf(dp[i], pr[i], dot);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) ds[i] = valid ? pr[i] * (dp[i] - dot) : 0.f;
            ds[4] = valid ? ddens * dens * (1.f - dens) : 0.f;
#pragma unroll
            for (int i = 0; i < 5; ++i) dsum[i] += ds[i];
            WAVE_LDS_SYNC();
            *reinterpret_cast<float4*>(&sc[lane * 12]) = make_float4(ds[0], ds[1], ds[2], ds[3]);
            *reinterpret_cast<float4*>(&sc[lane * 12 + 4]) = make_float4(ds[4], 0.f, 0.f, 0.f);
        }
        WAVE_LDS_SYNC();
        HSTAMP(4)
        // ---- dW2[i][j] += sum_rows d score[row][i] z1[row][j]   (step s: rows 16 (s / 4) + s % 4 + 4 kq)
        f32x4 dpre[4];
        {
            f32x4 odd = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float da[8], zb[8];
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) {
                    const int sidx = 8 * half + s8, row0 = 16 * (sidx >> 2) + (sidx & 3);
                    da[s8] = sc[(row0 + 4 * kq) * 12 + nn];
                    zb[s8] = zt[(row0 + 4 * kq) * 20 + n];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s8 = 0; s8 < 8; s8 += 2) {
                    dw2acc = __builtin_amdgcn_mfma_f32_16x16x4f32(da[s8], zb[s8], dw2acc, 0, 0, 0);
                    odd = __builtin_amdgcn_mfma_f32_16x16x4f32(da[s8 + 1], zb[s8 + 1], odd, 0, 0, 0);
                }
            }
            HSTAMP(5)
            // ---- d pre-activation of lin1 = (d scores W2) through the dropout and the ReLU (z1 > 0: kept AND active)
            f32x2 dv[4];
            float zm[4][4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                dv[tt] = lds_read_b64(&sc[(16 * tt + n) * 12 + 2 * kq]);
#pragma unroll
                for (int j = 0; j < 4; ++j) zm[tt][j] = zt[(16 * tt + 4 * kq + j) * 20 + n];
            }
            const float4 c4 = ctab[4 * 64];
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dv[0]), "+v"(dv[1]), "+v"(dv[2]), "+v"(dv[3]) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) dpre[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) dpre[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[tt][h], h ? c4.y : c4.x, dpre[tt], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) dw2acc[j] += odd[j];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int j = 0; j < 4; ++j) dpre[tt][j] = zm[tt][j] > 0.f ? dpre[tt][j] * drop_scale : 0.f;
        }
        WAVE_LDS_SYNC();                        // dW2 has read z1: its region takes d pre
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int j = 0; j < 4; ++j) zt[(16 * tt + 4 * kq + j) * 20 + n] = dpre[tt][j];
        WAVE_LDS_SYNC();
        HSTAMP(6)
        // (the next turn's rows are requested here, in front of the two longest tile chains, not at the top of the turn: a dozen
        // vector-memory instructions in front of lin1's LDS reads held those back -- 50.4 -> 48.8 us)
        if (r0 + stride < R) request(r0 + stride);
        // ---- dW1 | db1 += d pre^T [y | 1]   (the tile's columns 35.. feed accumulator columns nobody reads)
#pragma unroll
        for (int quarter = 0; quarter < 4; ++quarter) {
            float pa[4], yv[4][3];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int row = 16 * quarter + s4 + 4 * kq;
                pa[s4] = zt[row * 20 + n];
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) yv[s4][ct] = st[row * 36 + 16 * ct + n];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) dw1acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[s4], yv[s4][ct], dw1acc[ct], 0, 0, 0);
        }
        WAVE_LDS_SYNC();                        // dW1 has read the rows: their region takes the d rows
        HSTAMP(7)
        // ---- d rows = d pre W1
        {
            f32x2 pv[4][2];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) pv[tt][pp] = lds_read_b64(&zt[(16 * tt + n) * 20 + 8 * pp + 2 * kq]);
            float w1b[3][4];
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) {
                const float4 v = ctab[(5 + ct) * 64];
                w1b[ct][0] = v.x, w1b[ct][1] = v.y, w1b[ct][2] = v.z, w1b[ct][3] = v.w;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pv[0][0]), "+v"(pv[0][1]), "+v"(pv[1][0]), "+v"(pv[1][1]), "+v"(pv[2][0]), "+v"(pv[2][1]), "+v"(pv[3][0]), "+v"(pv[3][1]) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                f32x4 o[3];
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) o[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int ct = 0; ct < 3; ++ct)
                            o[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[tt][pp][h], w1b[ct][2 * pp + h], o[ct], 0, 0, 0);
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) {
                    if (16 * ct + n < 36) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) st[(16 * tt + 4 * kq + j) * 36 + 16 * ct + n] = o[ct][j];
                    }
                }
            }
        }
        }
        WAVE_LDS_SYNC();
        HSTAMP(8)
        if (!(HB_DIAG & 1)) {
            const long lim = (R - r0) * 9;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int e = lane + 64 * k;
                const float4 v = st4[e];
                if (e < lim) row_quad_st<BF>(dy, (size_t)r0, 36, e, v.x, v.y, v.z, v.w);
            }
        }
        WAVE_LDS_SYNC();
        HSTAMP(9)
    }
    // ---- the wave's weight-gradient image -> LDS, summed over the workgroup's waves, one atomic per element and workgroup
    __syncthreads();
    float* img_w = smem + wave * HB_WAVE_FLOATS;
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) {
        const int col = 16 * ct + n;
        if (col < 35) {
#pragma unroll
            for (int j = 0; j < 4; ++j) img_w[(4 * kq + j) * 35 + col] = dw1acc[ct][j];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (4 * kq + j < 5) img_w[16 * 35 + (4 * kq + j) * 16 + n] = dw2acc[j];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        float v = dsum[i];
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
        if (lane == 0) img_w[16 * 35 + 5 * 16 + i] = v;
    }
    __syncthreads();
    const int img = sn2_grad_image(rep_k, rep_stride);
    for (int i = threadIdx.x; i < HB_RED; i += 256) {
        const float v = (smem[i] + smem[HB_WAVE_FLOATS + i]) + (smem[2 * HB_WAVE_FLOATS + i] + smem[3 * HB_WAVE_FLOATS + i]);
        if (v == 0.f) continue;
        if (i < 16 * 35) {
            const int oo = i / 35, k = i - oo * 35;
            SN2_FLUSH_ADD(k < 34 ? &dW1[img + oo * 34 + k] : &db1[img + oo], v);
        } else if (i < 16 * 35 + 5 * 16) {
            SN2_FLUSH_ADD(&dW2[img + (i - 16 * 35)], v);
        } else {
            SN2_FLUSH_ADD(&db2[img + (i - 16 * 35 - 5 * 16)], v);
        }
    }
}

// dgamma / dbeta of a BatchNorm from the gradients of the Linear layer that consumes its output.  Let y = gamma*xhat +
// beta be the BatchNorm's output rows and let the consumer see u[r] = sum_k w_rk * y[idx_rk] with sum_k w_rk = 1 (the head:
// u = y; an FP block: the inverse-distance interpolation of knn_interpolate) in columns col0.. of its input.  With
// dy = (transposed interpolation of) W^T dpre:
//   dbeta[o]  = sum_rows dy[.][o]          = sum_j W[j][col0+o] * db[j]
//   dgamma[o] = sum_rows dy[.][o]*xhat[.][o] = sum_j W[j][col0+o] * G[j][o],  G[j][o] = sum_r dpre[r][j] * sum_k w_rk xhat[idx_rk][o]
// and dW[j][col0+o] = sum_r dpre[r][j]*u[r][o] = gamma[o]*G[j][o] + beta[o]*db[j], so G = (dW - beta*db) / gamma.
// C dot products of length cout instead of a pass over all rows (FP1's BatchNorm: 0.03 ms and 150 MB at C2).  Needs
// |gamma| > 1e-4 on every channel; accumulated in fp64.  When some |gamma| is too small for the division the same
// workgroups make the ordinary pass themselves, one channel each over all rows (h = the BatchNorm's input rows, dyv = the
// gradient of its output that the consumer's backward left: slow -- a strided column per workgroup -- and rare), so the
// sums are complete either way and sn2_fp_backward launches no kernel of its own for them (three launches per step that
// did nothing but read a flag).  ok (device int): 1 = the identity was used, 0 = the pass over the rows.
__global__ __launch_bounds__(256) void bn_sums_from_consumer_kernel(
    int C, int cout, int cin, int col0, const float* __restrict__ W, const float* __restrict__ dW,
    const float* __restrict__ db, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ dgamma, float* __restrict__ dbeta, int* __restrict__ ok, int rep_k, int rep_stride,
    const float* __restrict__ h, int h_stride, const float* __restrict__ dyv, int dy_stride, long R,
    const float* __restrict__ mean, const float* __restrict__ invstd, int rows_bf16) {
    // one workgroup per channel o of the BatchNorm; its threads share the (consumer row j, gradient image r) pairs
    __shared__ int s_ok;
    __shared__ double s_red[2][4];
    const int o = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_ok = 1;
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256)
        if (!(fabsf(gamma[c]) > 1e-4f)) s_ok = 0;               // also catches NaN
    __syncthreads();
    if (o == 0 && threadIdx.x == 0) *ok = s_ok;
    if (!s_ok) {
        const float mu = mean[o], is = invstd[o];
        double sb = 0.0, sg = 0.0;
        for (long r = threadIdx.x; r < R; r += 256) {
            float dd, hh;
            if (rows_bf16) {
                dd = __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(dyv)[(size_t)r * dy_stride + o] << 16);
                hh = __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(h)[(size_t)r * h_stride + o] << 16);
            } else {
                dd = dyv[(size_t)r * dy_stride + o], hh = h[(size_t)r * h_stride + o];
            }
            sb += (double)dd;
            sg += (double)(dd * ((hh - mu) * is));
        }
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) {
            sb += __shfl_xor(sb, m);
            sg += __shfl_xor(sg, m);
        }
        if (lane == 0) s_red[0][wave] = sb, s_red[1][wave] = sg;
        __syncthreads();
        if (threadIdx.x == 0) {
            dbeta[o] += (float)((s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]));
            dgamma[o] += (float)((s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]));
        }
        return;
    }
    const int images = rep_k > 1 ? rep_k : 1;                   // the consumer's (dW, db) images are summed on the fly
    const double g = (double)gamma[o], b = (double)beta[o];
    double sb = 0.0, sg = 0.0;
    for (int idx = threadIdx.x; idx < cout * images; idx += 256) {
        const int r = idx / cout, j = idx - r * cout;
        const double w = (double)W[j * cin + col0 + o];
        const double dbj = (double)db[(size_t)r * rep_stride + j];
        const double dwj = (double)dW[(size_t)r * rep_stride + j * cin + col0 + o];
        sb += w * dbj;
        sg += w * (dwj - b * dbj);
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        sb += __shfl_xor(sb, m);
        sg += __shfl_xor(sg, m);
    }
    if (lane == 0) s_red[0][wave] = sb, s_red[1][wave] = sg;
    __syncthreads();
    if (threadIdx.x == 0) {
        sb = (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]);
        sg = (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]);
        dbeta[o] += (float)sb;
        dgamma[o] += (float)(sg / g);
    }
}

int check_head(const sn2_head* p) {
    if (!p || p->R <= 0 || p->cin != 34 || p->f_stride != 36) return p && p->R > 0 ? SN2_ELIMIT : SN2_EINVAL;
    if (!p->f || !p->fa || !p->fc || !p->W1 || !p->b1 || !p->W2 || !p->b2) return SN2_EINVAL;
    return 0;
}

}  // namespace

// sweeps (~1 us each) before an exchange wait of global_level_fwd_kernel gives up; tests shorten it to provoke a give-up
static unsigned g_gl_spin_limit = 1u << 18;
extern "C" int sn2_debug_global_spin_limit(unsigned sweeps) {       // (0 = back to the default)
    g_gl_spin_limit = sweeps ? sweeps : (1u << 18);
    return 0;
}

extern "C" int sn2_global_level_forward(const sn2_fp* sa3, const sn2_fp* fp3, float* x3, int* arg3, unsigned long long* xchg,
                                        unsigned* ctl, void* stream) {
    if (!sa3 || !fp3 || !x3 || !arg3 || !xchg || !ctl) return SN2_EINVAL;
    const int B = sa3->B, M2 = sa3->R_per_plot;
    // the two layers of the reference architecture's global level, fp32, training mode (both BatchNorms take batch statistics)
    if (!(B > 0 && M2 > 0 && fp3->B == B && fp3->R_per_plot == M2 && sa3->S_per_plot == M2 && fp3->S_per_plot == 1)) return SN2_EINVAL;
    if (!(sa3->ca == 32 && sa3->cb == 3 && sa3->blk.cin == 35 && sa3->blk.cout == 64 && fp3->ca == 64 && fp3->cb == 32 &&
          fp3->blk.cin == 96 && fp3->blk.cout == 64))
        return SN2_ELIMIT;
    if (sa3->blk.mma_bf16 || fp3->blk.mma_bf16 || sa3->act_bf16 || fp3->act_bf16) return SN2_ELIMIT;
    if (sa3->knn_idx || sa3->src_a || !fp3->knn_idx || !fp3->knn_w || fp3->src_a) return SN2_EINVAL;
    if (!sa3->src || sa3->src_stride != 32 || !sa3->skip || sa3->skip_stride != 4 || !sa3->h || sa3->h_stride != 64) return SN2_EINVAL;
    if (fp3->src != x3 || fp3->src_stride != 64 || fp3->skip != sa3->src || fp3->skip_stride != 32 || !fp3->h || fp3->h_stride != 64)
        return SN2_EINVAL;
    if (B > GL_MAX_PLOTS || (long)B * M2 >= (1L << 31) / 64) return SN2_ELIMIT;
    for (const sn2_block* k : {&sa3->blk, &fp3->blk})
        if (!k->W || !k->b || !k->gamma || !k->beta || !k->running_mean || !k->running_var || !k->a || !k->c || !k->mean || !k->invstd)
            return SN2_EINVAL;
    GlArgs A;
    A.B = B, A.M2 = M2;
    A.x2 = sa3->src, A.pos2 = sa3->skip, A.knn_idx = fp3->knn_idx, A.knn_w = fp3->knn_w, A.x3 = x3, A.arg3 = arg3;
    auto layer = [](const sn2_fp* p) {
        GlLayer l;
        l.W = p->blk.W, l.bias = p->blk.b, l.gamma = p->blk.gamma, l.beta = p->blk.beta;
        l.running_mean = p->blk.running_mean, l.running_var = p->blk.running_var;
        l.a = p->blk.a, l.c = p->blk.c, l.mean = p->blk.mean, l.invstd = p->blk.invstd;
        l.nbt = p->blk.num_batches_tracked, l.h = p->h;
        return l;
    };
    A.sa3 = layer(sa3), A.fp3 = layer(fp3);
    A.xchg = xchg, A.ctl = ctl;
    A.spin_limit = g_gl_spin_limit;
    const size_t lds = ((size_t)GL_FIXED_FLOATS + (size_t)B * GL_GROUPS * 128) * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&global_level_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // (the repair of a launch whose waits gave up -- nothing guarantees that its B workgroups are resident together -- is done
    // inside the launch, by the workgroup that leaves last)
    hipLaunchKernelGGL(global_level_fwd_kernel, dim3(B), dim3(1024), lds, (hipStream_t)stream, A);
    SN2_RETURN_LAUNCH();
}

extern "C" int sn2_head_forward(const sn2_head* p, void* stream) {
    SN2_TRY(check_head(p));
    if (!p->coverages || !p->proba) return SN2_EINVAL;
    if (p->zero_fill && ((p->zero_fill_words & 3) || p->zero_fill_words <= 0 || ((uintptr_t)p->zero_fill & 15))) return SN2_EINVAL;
    // check_head: rows of exactly 36 floats (34 channels)
    auto kf = p->act_bf16 ? &head_fwd_mfma_kernel<true> : &head_fwd_mfma_kernel<false>;
    // no more workgroups than are resident together (SN2_HF_OCC per CU; each wave loops over its turns): with 1024 workgroups at
    // three per CU a quarter of them ran as a second round at a third of the occupancy (round 5; SN2_HF_WGS_PER_CU: experiment switch)
    static const int hf_wgs = getenv("SN2_HF_WGS_PER_CU") ? atoi(getenv("SN2_HF_WGS_PER_CU")) : SN2_HF_OCC;
    int hf_grid = pick_grid(p->R * grid_mult, 256, 2);
    if (hf_wgs > 0 && hf_grid > hf_wgs * sn2_cu_count()) hf_grid = hf_wgs * sn2_cu_count();
    hipLaunchKernelGGL(kf, dim3(hf_grid), dim3(256), 0, (hipStream_t)stream, p->R, p->f, p->fa,
                       p->fc, p->W1, p->b1, p->W2, p->b2, p->coverages, p->proba, p->drop_mask,
                       p->drop_mask ? p->drop_scale : 1.f, reinterpret_cast<float4*>(p->zero_fill),
                       p->zero_fill ? p->zero_fill_words / 4 : 0L);
    SN2_RETURN_LAUNCH();
}

extern "C" int sn2_fp_head_eval(const sn2_fp* p, const sn2_head* hd, void* stream) {
    // (neither p->h nor hd->f is read: the rows stay in LDS)
    if (!p || !hd || p->B <= 0 || p->R_per_plot <= 0 || p->S_per_plot <= 0 || !p->src || !p->blk.W || !p->blk.b || !p->skip ||
        !p->blk.a || !p->blk.c || (p->src_stride & 3) || p->src_stride < p->ca || p->blk.cin != p->ca + p->cb)
        return SN2_EINVAL;
    if (hd->R <= 0 || hd->cin != 34 || !hd->fa || !hd->fc || !hd->W1 || !hd->b1 || !hd->W2 || !hd->b2) return SN2_EINVAL;
    if (!hd->coverages || !hd->proba || hd->drop_mask || hd->act_bf16 || p->act_bf16) return SN2_EINVAL;
    if (!(p->knn_idx && p->ca == 34 && p->cb == 8 && p->blk.cout == 34 && hd->R == p->B * p->R_per_plot)) return SN2_ELIMIT;
    if (!(p->src_ws && (p->skip_stride & 3) == 0 && p->src_stride >= 36 && (long)hd->R * 3 < (1L << 31))) return SN2_ELIMIT;
    hipStream_t st = (hipStream_t)stream;
    const int R = hd->R, n_src = p->B * p->S_per_plot;
    // the layer's BatchNorm on its running statistics -> (a, c) = what the head applies to the rows (hd->fa, hd->fc name the
    // same two vectors: p->blk.a, p->blk.c)
    SN2_TRY(sn2_bn_finalize(&p->blk, 0, nullptr, R, 0, st));
    SN2_TRY((launch_src_table<34, 8, 34>(n_src, p->src_stride, p->src, p->src_a, p->src_c, p->blk.W, p->src_ws, st)));
    const long turns = ((long)R + 62) / 63;
    int grid = sn2_cdiv(turns, 4);
    if (g_fp_rows_form != 0 && grid >= 2 * sn2_cu_count()) {
        // (round 5) the pipelined form: as many workgroups as are resident together (two per CU by its registers), each wave
        // several turns with the next turn's inputs in flight
        grid = 2 * sn2_cu_count();
        hipLaunchKernelGGL((fp_head_eval2_kernel<34, 8, 34>), dim3(grid), dim3(256), 0, st, R, p->R_per_plot, p->S_per_plot,
                           p->skip_stride, (const float*)p->src_ws, p->knn_idx, p->knn_w, p->skip, p->blk.W, p->blk.b, hd->fa, hd->fc,
                           hd->W1, hd->b1, hd->W2, hd->b2, hd->coverages, hd->proba);
        SN2_RETURN_LAUNCH();
    }
    const int cap = 4 * sn2_cu_count();
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL((fp_head_eval_kernel<34, 8, 34>), dim3(grid), dim3(256), 0, st, R, p->R_per_plot, p->S_per_plot,
                       p->skip_stride, (const float*)p->src_ws, p->knn_idx, p->knn_w, p->skip, p->blk.W, p->blk.b, hd->fa, hd->fc,
                       hd->W1, hd->b1, hd->W2, hd->b2, hd->coverages, hd->proba);
    SN2_RETURN_LAUNCH();
}

extern "C" int sn2_head_bn_sums(const sn2_head* p, const float* gamma, const float* beta, const float* mean,
                                const float* invstd, float* dgamma, float* dbeta, int* ok, void* stream) {
    SN2_TRY(check_head(p));
    if (!p->dW1 || !p->db1 || !p->dy || !gamma || !beta || !mean || !invstd || !dgamma || !dbeta || !ok) return SN2_EINVAL;
    hipLaunchKernelGGL(bn_sums_from_consumer_kernel, dim3(p->cin), dim3(256), 0, (hipStream_t)stream, p->cin, 16, p->cin, 0, p->W1,
                       (const float*)p->dW1, (const float*)p->db1, gamma, beta, dgamma, dbeta, ok, p->grad_replicas,
                       p->grad_replica_stride, p->f, p->f_stride, (const float*)p->dy, p->f_stride, (long)p->R, mean, invstd,
                       p->act_bf16);
    SN2_RETURN_LAUNCH();
}

extern "C" int sn2_fp_bn_sums(const sn2_fp* p, const float* gamma, const float* beta, const float* mean, const float* invstd,
                              float* dgamma, float* dbeta, int* ok, void* stream) {
    SN2_TRY(check_fp(p));
    if (!p->knn_idx || !p->blk.dW || !p->blk.db || !p->dsrc || !gamma || !beta || !mean || !invstd || !dgamma || !dbeta || !ok ||
        p->ca > 64)
        return SN2_EINVAL;
    hipLaunchKernelGGL(bn_sums_from_consumer_kernel, dim3(p->ca), dim3(256), 0, (hipStream_t)stream, p->ca, p->blk.cout, p->blk.cin, 0,
                       (const float*)p->blk.W, (const float*)p->blk.dW, (const float*)p->blk.db, gamma, beta, dgamma, dbeta, ok,
                       p->blk.grad_replicas, p->blk.grad_replica_stride, p->src, p->src_stride, (const float*)p->dsrc,
                       p->dsrc_stride, (long)p->B * p->S_per_plot, mean, invstd, 0);
    SN2_RETURN_LAUNCH();
}

extern "C" int sn2_head_backward(const sn2_head* p, void* stream) {
    SN2_TRY(check_head(p));
    if (!p->dy || !p->dW1 || !p->db1 || !p->dW2 || !p->db2) return SN2_EINVAL;
    constexpr size_t lds_m = ((size_t)HB_WAVE_FLOATS * 4 + HB_TAB_FLOATS) * 4;      // 79 136 bytes: two workgroups per CU
    auto km = p->act_bf16 ? &head_bwd_mfma_kernel<true> : &head_bwd_mfma_kernel<false>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(km), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m);
    int gm = sn2_cdiv(sn2_cdiv(p->R, 64), 4);
    const int cap = 2 * sn2_cu_count() * grid_mult;
    if (gm > cap) gm = cap;
    hipLaunchKernelGGL(km, dim3(gm), dim3(256), lds_m, (hipStream_t)stream, p->R, p->f, p->fa, p->fc, p->W1, p->b1, p->W2,
                       p->b2, p->dcoverages, p->dproba, p->dy, p->dW1, p->db1, p->dW2, p->db2, p->grad_replicas,
                       p->grad_replica_stride, p->drop_mask, p->drop_mask ? p->drop_scale : 1.f);
    SN2_RETURN_LAUNCH();
}
