// geometry.hip -- position-only kernels: row packing, farthest point sampling, radius ball query, 3-NN.
// All discrete decisions use the canonical fp32 squared distance sn2_d2 (common.h) so that the index structures
// are bit-identical to the oracle's.
#include "common.h"

// ------------------------------------------------------------------------------------------------------------
// pack_rows: (cloud (B,C,N), xyz (B,3,N)) -> rows0 (B*N,12) = [cloud rows 2..9 | x y z 0]
// HBM-bound: reads 44 B/point coalesced (11 channel rows), writes 48 B/point as 3 x 16 B per lane.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ cloud, const float* __restrict__ xyz,
                                                        int C, int N, float* __restrict__ rows0) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float* cl = cloud + (size_t)b * C * N + n;
    const float* xp = xyz + (size_t)b * 3 * N + n;
    float4 v0 = make_float4(cl[2 * (size_t)N], cl[3 * (size_t)N], cl[4 * (size_t)N], cl[5 * (size_t)N]);
    float4 v1 = make_float4(cl[6 * (size_t)N], cl[7 * (size_t)N], cl[8 * (size_t)N], cl[9 * (size_t)N]);
    float4 v2 = make_float4(xp[0], xp[(size_t)N], xp[2 * (size_t)N], 0.f);
    float4* o = reinterpret_cast<float4*>(rows0 + ((size_t)b * N + n) * 12);
    o[0] = v0;
    o[1] = v1;
    o[2] = v2;
}

extern "C" int sn2_pack_rows(const float* cloud, const float* xyz, int B, int C, int N, float* rows0, void* stream) {
    if (!cloud || !xyz || !rows0 || B <= 0 || N <= 0) return SN2_EINVAL;
    if (C != 10) return SN2_ELIMIT;
    dim3 grid(sn2_cdiv(N, 256), B);
    hipLaunchKernelGGL(pack_rows_kernel, grid, dim3(256), 0, (hipStream_t)stream, cloud, xyz, C, N, rows0);
    SN2_RETURN_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// fps: one workgroup per plot; the plot's points and their running min-distance stay in VGPRs (PPT per lane);
// per round: PPT canonical distance updates per lane, a per-lane running (max, slot), a 64-bit wave max over
// (dist_bits << 32 | ~index) -- max distance, lowest index on ties, as torch.argmax -- one LDS exchange between the
// waves and ONE barrier (LDS slots double-buffered by round parity).
// Latency/VALU-bound by construction: M strictly sequential rounds; HBM traffic is 12 B/point once.
// ------------------------------------------------------------------------------------------------------------
template <int PPT, int T, bool ZLDS>
__global__ __launch_bounds__(T) void fps_kernel(const float* __restrict__ pos, int N, int M,
                                                const int* __restrict__ start, int* __restrict__ idx_out,
                                                float* __restrict__ cpos_soa, float* __restrict__ cpos_aos) {
    constexpr int NW = T / 64;
    __shared__ unsigned long long s_key[2][NW];
    // ZLDS (PPT = 32): x, y and the running distance fill the 128-VGPR budget of a 1024-lane workgroup, so the z
    // row lives in LDS (128 KiB, each lane re-reads only its own slots: conflict-free ds_read_b32).
    __shared__ float s_z[ZLDS ? PPT * T : 1];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* px = pos + (size_t)b * 3 * N;
    const float* py = px + N;
    const float* pz = py + N;
    float x[PPT], y[PPT], z[ZLDS ? 1 : PPT], d[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int j = k * T + tid;
        const bool v = j < N;
        x[k] = v ? px[j] : 0.f;
        y[k] = v ? py[j] : 0.f;
        if constexpr (ZLDS) s_z[j] = v ? pz[j] : 0.f; else z[k] = v ? pz[j] : 0.f;
        d[k] = v ? INFINITY : 0.f;  // padding lanes: distance 0 and an index above every real point -> never win
    }
    int cur = start ? start[b] : 0;
    cur = cur < 0 ? 0 : (cur >= N ? N - 1 : cur);
    for (int i = 0; i < M; ++i) {
        cur = __builtin_amdgcn_readfirstlane(cur);
        const float lx = px[cur], ly = py[cur], lz = pz[cur];
        if (tid == 0) {
            idx_out[(size_t)b * M + i] = cur;
            cpos_soa[((size_t)b * 3 + 0) * M + i] = lx;
            cpos_soa[((size_t)b * 3 + 1) * M + i] = ly;
            cpos_soa[((size_t)b * 3 + 2) * M + i] = lz;
            reinterpret_cast<float4*>(cpos_aos)[(size_t)b * M + i] = make_float4(lx, ly, lz, 0.f);
        }
        if (i == M - 1) break;
        float best = -1.f;
        int bk = 0;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            float zk;
            if constexpr (ZLDS) zk = s_z[k * T + tid]; else zk = z[k];
            const float dd = sn2_d2(x[k], y[k], zk, lx, ly, lz);
            const float nd = fminf(d[k], dd);
            d[k] = nd;
            if (nd > best) {  // strict: the lowest slot (= lowest index of this lane) wins ties
                best = nd;
                bk = k;
            }
            // keep the scheduler from hoisting all 32 LDS reads at once (that spills at the 128-VGPR budget)
            if constexpr (ZLDS) { if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0); }
        }
        unsigned long long key = ((unsigned long long)__float_as_uint(best) << 32) |
                                 (unsigned long long)(0xFFFFFFFFu - (unsigned)(bk * T + tid));
        key = wave_max_u64(key);
        if (lane == 0) s_key[i & 1][wave] = key;
        __syncthreads();
        unsigned long long k2 = s_key[i & 1][lane % NW];
#pragma unroll
        for (int o = NW / 2; o > 0; o >>= 1) {
            unsigned long long t = __shfl_xor(k2, o);
            k2 = t > k2 ? t : k2;
        }
        cur = (int)(0xFFFFFFFFu - (unsigned)(k2 & 0xFFFFFFFFull));
    }
}

template <int PPT, int T, bool ZLDS = false>
static int launch_fps(const float* pos, int B, int N, int M, const int* start, int* idx, float* cs, float* ca,
                      hipStream_t st) {
    hipLaunchKernelGGL((fps_kernel<PPT, T, ZLDS>), dim3(B), dim3(T), 0, st, pos, N, M, start, idx, cs, ca);
    SN2_RETURN_LAUNCH();
}

extern "C" int sn2_fps(const float* pos_soa, int B, int N, int M, const int* start, int* idx, float* cpos_soa,
                       float* cpos_aos, void* stream) {
    if (!pos_soa || !idx || !cpos_soa || !cpos_aos || B <= 0 || N <= 0 || M <= 0 || M > N) return SN2_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (N <= 256) return launch_fps<1, 256>(pos_soa, B, N, M, start, idx, cpos_soa, cpos_aos, st);
    if (N <= 512) return launch_fps<2, 256>(pos_soa, B, N, M, start, idx, cpos_soa, cpos_aos, st);
    if (N <= 1024) return launch_fps<4, 256>(pos_soa, B, N, M, start, idx, cpos_soa, cpos_aos, st);
    if (N <= 2048) return launch_fps<2, 1024>(pos_soa, B, N, M, start, idx, cpos_soa, cpos_aos, st);
    if (N <= 4096) return launch_fps<4, 1024>(pos_soa, B, N, M, start, idx, cpos_soa, cpos_aos, st);
    if (N <= 8192) return launch_fps<8, 1024>(pos_soa, B, N, M, start, idx, cpos_soa, cpos_aos, st);
    if (N <= 16384) return launch_fps<16, 1024>(pos_soa, B, N, M, start, idx, cpos_soa, cpos_aos, st);
    if (N <= 32768) return launch_fps<32, 1024, true>(pos_soa, B, N, M, start, idx, cpos_soa, cpos_aos, st);
    return SN2_ELIMIT;
}

// ------------------------------------------------------------------------------------------------------------
// ball_query: one wave owns TC centroids of one plot (coordinates and running counts wave-uniform -> SGPRs) and
// streams the plot's points 64 at a time (coalesced SoA loads, L2-resident after the first wave); per centroid a
// ballot + prefix popcount compacts the hits, so each list comes out in ascending source index and the stores of
// one wave-instruction are contiguous.  Algorithmic HBM bytes: 12*(N+M) read + 4*E + 4*M written per plot.
// ------------------------------------------------------------------------------------------------------------
template <int TC>
__global__ __launch_bounds__(256) void ball_query_kernel(const float* __restrict__ src, int B, int N,
                                                         const float* __restrict__ cpos, int M, float r2, int cap,
                                                         int* __restrict__ nbr, int* __restrict__ cnt,
                                                         unsigned long long* __restrict__ total, int tiles_per_plot) {
    const int lane = threadIdx.x & 63;
    const int wg = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));
    const int b = wg / tiles_per_plot;
    if (b >= B) return;
    const int c0 = (wg - b * tiles_per_plot) * TC;
    const float* px = src + (size_t)b * 3 * N;
    const float* py = px + N;
    const float* pz = py + N;
    float cx[TC], cy[TC], cz[TC];
    int n[TC];
#pragma unroll
    for (int t = 0; t < TC; ++t) {
        const int ci = (c0 + t < M) ? c0 + t : M - 1;
        cx[t] = cpos[((size_t)b * 3 + 0) * M + ci];
        cy[t] = cpos[((size_t)b * 3 + 1) * M + ci];
        cz[t] = cpos[((size_t)b * 3 + 2) * M + ci];
        n[t] = 0;
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int j0 = 0; j0 < N; j0 += 64) {
        const int j = j0 + lane;
        const bool valid = j < N;
        const float x = valid ? px[j] : 0.f, y = valid ? py[j] : 0.f, z = valid ? pz[j] : 0.f;
#pragma unroll
        for (int t = 0; t < TC; ++t) {
            const bool hit = valid && (sn2_d2(x, y, z, cx[t], cy[t], cz[t]) < r2);
            const unsigned long long mask = __ballot(hit);
            if (mask) {
                const int p = n[t] + __popcll(mask & below);
                if (hit && p < cap && c0 + t < M) nbr[((size_t)b * M + c0 + t) * cap + p] = j;
                n[t] += __popcll(mask);
            }
        }
    }
    unsigned long long sum = 0;
#pragma unroll
    for (int t = 0; t < TC; ++t) {
        const int c = n[t] < cap ? n[t] : cap;
        if (c0 + t < M) {
            if (lane == 0) cnt[(size_t)b * M + c0 + t] = c;
            sum += (unsigned long long)c;
        }
    }
    if (lane == 0 && total) atomicAdd(total, sum);
}

extern "C" int sn2_ball_query(const float* src_soa, int B, int N, const float* cpos_soa, int M, float r2, int cap,
                              int* nbr, int* cnt, unsigned long long* total, void* stream) {
    if (!src_soa || !cpos_soa || !nbr || !cnt || B <= 0 || N <= 0 || M <= 0 || cap <= 0) return SN2_EINVAL;
    constexpr int TC = 8;
    const int tiles = sn2_cdiv(M, TC);
    const long waves = (long)B * tiles;
    hipLaunchKernelGGL((ball_query_kernel<TC>), dim3(sn2_cdiv(waves, 4)), dim3(256), 0, (hipStream_t)stream, src_soa, B,
                       N, cpos_soa, M, r2, cap, nbr, cnt, total, tiles);
    SN2_RETURN_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// three_nn: one lane per target point, the plot's source positions staged through LDS in tiles of 1024 (AoS4, every
// lane reads the same address -> broadcast), ascending source index with strict '<' insertion so ties keep the
// lowest index (oracle: stable sort).  Writes idx (B*T,3), w (B*T,3) = 1/max(d2,1e-16); unused slots w = 0.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void three_nn_kernel(const float* __restrict__ src, int S, const float* __restrict__ dst,
                                                       int T, int k, int* __restrict__ idx, float* __restrict__ w) {
    __shared__ float4 s_src[1024];
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const bool valid = t < T;
    const float* sx = src + (size_t)b * 3 * S;
    const float* dx = dst + (size_t)b * 3 * T;
    const float qx = valid ? dx[t] : 0.f, qy = valid ? dx[(size_t)T + t] : 0.f, qz = valid ? dx[2 * (size_t)T + t] : 0.f;
    float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
    int i0 = -1, i1 = -1, i2 = -1;
    for (int s0 = 0; s0 < S; s0 += 1024) {
        const int tn = (S - s0) < 1024 ? (S - s0) : 1024;
        __syncthreads();
        for (int i = threadIdx.x; i < tn; i += 256)
            s_src[i] = make_float4(sx[s0 + i], sx[(size_t)S + s0 + i], sx[2 * (size_t)S + s0 + i], 0.f);
        __syncthreads();
        for (int i = 0; i < tn; ++i) {
            const float4 p = s_src[i];
            const float dd = sn2_d2(p.x, p.y, p.z, qx, qy, qz);
            if (dd < d2) {
                if (dd < d1) {
                    d2 = d1;
                    i2 = i1;
                    if (dd < d0) {
                        d1 = d0;
                        i1 = i0;
                        d0 = dd;
                        i0 = s0 + i;
                    } else {
                        d1 = dd;
                        i1 = s0 + i;
                    }
                } else {
                    d2 = dd;
                    i2 = s0 + i;
                }
            }
        }
    }
    if (!valid) return;
    const size_t o = ((size_t)b * T + t) * 3;
    const bool u1 = (k >= 2) && (i1 >= 0), u2 = (k >= 3) && (i2 >= 0);
    idx[o + 0] = i0;
    idx[o + 1] = u1 ? i1 : i0;
    idx[o + 2] = u2 ? i2 : i0;
    w[o + 0] = 1.0f / fmaxf(d0, 1e-16f);
    w[o + 1] = u1 ? 1.0f / fmaxf(d1, 1e-16f) : 0.f;
    w[o + 2] = u2 ? 1.0f / fmaxf(d2, 1e-16f) : 0.f;
}

extern "C" int sn2_three_nn(const float* src_soa, int B, int S, const float* dst_soa, int T, int k, int* idx, float* w,
                            void* stream) {
    if (!src_soa || !dst_soa || !idx || !w || B <= 0 || S <= 0 || T <= 0 || k < 1 || k > 3) return SN2_EINVAL;
    dim3 grid(sn2_cdiv(T, 256), B);
    hipLaunchKernelGGL(three_nn_kernel, grid, dim3(256), 0, (hipStream_t)stream, src_soa, S, dst_soa, T, k, idx, w);
    SN2_RETURN_LAUNCH();
}
