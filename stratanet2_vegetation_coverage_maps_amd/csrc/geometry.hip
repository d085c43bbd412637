// geometry.hip -- position-only kernels: row packing, farthest point sampling, radius ball query, 3-NN.
// All discrete decisions use the canonical fp32 squared distance sn2_d2 (common.h) so that the index structures
// are bit-identical to the oracle's.
#include "common.h"
#include <stdlib.h>

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
    // PLDS: a copy of the positions in LDS (16 B per point) where it is small: the winner's coordinates are then one LDS read
    // per round instead of three dependent global loads (~500 clocks of the ~1250 a round of the 1024 -> 256 level took)
    constexpr bool PLDS = !ZLDS && PPT * T * 16 <= 32 * 1024;
    __shared__ float4 s_pos[PLDS ? PPT * T : 1];
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
        if constexpr (PLDS) s_pos[j] = make_float4(x[k], y[k], z[k], 0.f);
        d[k] = v ? INFINITY : 0.f;  // padding lanes: distance 0 and an index above every real point -> never win
    }
    int cur = start ? start[b] : 0;
    cur = cur < 0 ? 0 : (cur >= N ? N - 1 : cur);
    if constexpr (PLDS) __syncthreads();
    for (int i = 0; i < M; ++i) {
        cur = __builtin_amdgcn_readfirstlane(cur);
        float lx, ly, lz;
        if constexpr (PLDS) {
            const float4 c = s_pos[cur];
            lx = c.x; ly = c.y; lz = c.z;
        } else {
            lx = px[cur]; ly = py[cur]; lz = pz[cur];
        }
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
        // wave level: maximal distance by DPP, then the lowest index among the lanes attaining it (two 6-step DPP
        // reductions instead of a 64-bit ds_bpermute butterfly)
        const float wm = wave_max_dpp(best);
        const unsigned wi = wave_min_u32_dpp(best == wm ? (unsigned)(bk * T + tid) : 0xFFFFFFFFu);
        if (lane == 0) s_key[i & 1][wave] = ((unsigned long long)__float_as_uint(wm) << 32) | (unsigned long long)wi;
        __syncthreads();
        const unsigned long long k2 = s_key[i & 1][lane % NW];
        const float gv = wave_max_dpp(__uint_as_float((unsigned)(k2 >> 32)));
        const unsigned gi = wave_min_u32_dpp(__uint_as_float((unsigned)(k2 >> 32)) == gv ? (unsigned)(k2 & 0xFFFFFFFFull) : 0xFFFFFFFFu);
        cur = (int)gi;
    }
}

template <int PPT, int T, bool ZLDS = false>
static int launch_fps(const float* pos, int B, int N, int M, const int* start, int* idx, float* cs, float* ca,
                      hipStream_t st) {
    hipLaunchKernelGGL((fps_kernel<PPT, T, ZLDS>), dim3(B), dim3(T), 0, st, pos, N, M, start, idx, cs, ca);
    SN2_RETURN_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// Bucketed FPS (exact).  The brute-force kernel above touches all N points in each of the M sequential rounds
// (4.1 ms for 16 x 32768 -> 1024 on MI355X: 40 % of a training step).  But a new sample s only lowers the running
// distance of points closer to s than their current distance, i.e. points near s.  So:
//   * spatial_order_kernel sorts a plot's points by the Morton code of a 16x16x16 cell grid (LDS counting sort);
//   * consecutive runs of 64 sorted points form a BUCKET = one VGPR slot of one wave (bucket b -> wave b % 16, slot
//     b / 16, so spatially adjacent buckets sit in different waves and a round's work spreads over the 16 waves);
//   * per bucket LDS keeps the bounding box and the maximum running distance; per round each wave tests its <= 32
//     buckets (one lane each): a bucket can only change if  boxdist2(s) < bucket_max.  The test is EXACT, not a
//     heuristic: boxdist2 is evaluated with the same canonical fp32 operations as sn2_d2, and fp32 rounding is
//     monotone, so boxdist2(s) <= d2(p, s) for every point p of the box, bit for bit; if boxdist2(s) >= bucket_max
//     then min(dist_p, d2(p,s)) == dist_p for all its points and skipping the bucket changes nothing.
//   * dirty buckets are updated by their wave (x, y, z in VGPRs, running distance in LDS) and get a new maximum;
//   * the next sample is the point of maximal running distance; exact ties are resolved towards the lowest ORIGINAL
//     index, as torch.argmax does on the unsorted array (candidates = lanes whose distance equals the global maximum).
// Result: bit-identical indices to the brute-force kernel, with ~20x fewer distance evaluations after the first rounds.
// ------------------------------------------------------------------------------------------------------------
// 16 x 16 x 16 cells over the plot's bounding box, full 3-D Morton order.  (Measured on the synthetic plots: 19.6 of 512
// buckets change per FPS round on average, against 27.7 for a 32 x 32 x 8 grid whose two top levels split x,y only.)
constexpr int ORDER_GX = 16, ORDER_GZ = 16, ORDER_CELLS = ORDER_GX * ORDER_GX * ORDER_GZ;

__device__ __forceinline__ unsigned morton_cell(unsigned cx, unsigned cy, unsigned cz) {
    unsigned c = 0;
#pragma unroll
    for (int bit = 3; bit >= 0; --bit) c = (c << 3) | (((cx >> bit) & 1u) << 2) | (((cy >> bit) & 1u) << 1) | ((cz >> bit) & 1u);
    return c;
}

// grid header per plot (32-bit words): [0..4096] first sorted position of every Morton cell (+ end), then lo.xyz, scale.xyz
constexpr int GRID_WORDS = ORDER_CELLS + 1 + 6 + 1;   // padded to an even count
// exchange area of the multi-workgroup FPS (fps_cluster_kernel): per plot FPS_XCHG_WORDS words of tagged granules, then
// FPS_CTL_WORDS control words per launch (ticket counter, timeout count).  Zeroed HERE, by the kernel in front of every FPS
// launch (a kernel boundary: visible to every workgroup behind it; the tags count super-rounds from 1, so 0 = nothing yet).
constexpr int FPS_XCHG_WORDS = 4096, FPS_CTL_WORDS = 32;
// One workgroup of 1024 threads per plot walks the points three times (bounding box, histogram, scatter); a trip fetches U = 16
// points per thread before it touches the first (two trips per pass at N = 32 768; one DEPENDENT load per trip took 71 us,
// all 32 points of a thread in registers at once spill at 1024 threads).  What is left (61 us) are the LDS atomics: 55 % of a
// plot's points are ground points in a few cell layers, and a wave's adds onto one counter serialise.
// rank (or NULL): rank[plot * N + i] = sorted position of point i -- the inverse of `order` (sn2_fp.row_perm)
template <int U, int NT>
__global__ __launch_bounds__(NT) void spatial_order_chunk_kernel(const float* __restrict__ pos, int N, int* __restrict__ order,
                                                                 float4* __restrict__ sorted, int* __restrict__ grid,
                                                                 unsigned* __restrict__ xchg, unsigned* __restrict__ ctl,
                                                                 int* __restrict__ rank) {
    __shared__ int s_hist[ORDER_CELLS];
    __shared__ float s_mm[6][NT / 64];
    __shared__ int s_wsum[NT / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* px = pos + (size_t)b * 3 * N;
    const float* py = px + N;
    const float* pz = py + N;
    // fn(i, x, y, z) for every point i of the plot, U points per thread and trip
    auto for_points = [&](auto fn) {
        for (int i0 = 0; i0 < N; i0 += NT * U) {
            float vx[U], vy[U], vz[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * NT + tid;
                const int ii = i < N ? i : N - 1;
                vx[u] = px[ii]; vy[u] = py[ii]; vz[u] = pz[ii];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * NT + tid;
                if (i < N) fn(i, vx[u], vy[u], vz[u]);
            }
        }
    };
    if (xchg) {
        for (int i = tid; i < FPS_XCHG_WORDS; i += NT) xchg[(size_t)b * FPS_XCHG_WORDS + i] = 0u;
        if (b == 0 && tid < FPS_CTL_WORDS) ctl[tid] = 0u;
    }
    for (int i = tid; i < ORDER_CELLS; i += NT) s_hist[i] = 0;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for_points([&](int, float vx, float vy, float vz) {
        mn[0] = fminf(mn[0], vx); mx[0] = fmaxf(mx[0], vx);
        mn[1] = fminf(mn[1], vy); mx[1] = fmaxf(mx[1], vy);
        mn[2] = fminf(mn[2], vz); mx[2] = fmaxf(mx[2], vz);
    });
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        mn[a] = wave_min(mn[a]);
        mx[a] = wave_max(mx[a]);
        if (lane == 0) {
            s_mm[a][wave] = mn[a];
            s_mm[3 + a][wave] = mx[a];
        }
    }
    __syncthreads();
    float lo[3], sc[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = s_mm[a][0], h = s_mm[3 + a][0];
        for (int k = 1; k < NT / 64; ++k) {
            l = fminf(l, s_mm[a][k]);
            h = fmaxf(h, s_mm[3 + a][k]);
        }
        lo[a] = l;
        const float ext = fmaxf(h - l, 1e-6f);
        sc[a] = (a < 2 ? (float)ORDER_GX : (float)ORDER_GZ) / ext;
    }
    auto cell_of = [&](float vx, float vy, float vz) -> unsigned {     // the arithmetic of spatial_order_kernel, bit for bit
        int cx = (int)((vx - lo[0]) * sc[0]), cy = (int)((vy - lo[1]) * sc[1]), cz = (int)((vz - lo[2]) * sc[2]);
        cx = cx < 0 ? 0 : (cx > ORDER_GX - 1 ? ORDER_GX - 1 : cx);
        cy = cy < 0 ? 0 : (cy > ORDER_GX - 1 ? ORDER_GX - 1 : cy);
        cz = cz < 0 ? 0 : (cz > ORDER_GZ - 1 ? ORDER_GZ - 1 : cz);
        return morton_cell((unsigned)cx, (unsigned)cy, (unsigned)cz);
    };
    for_points([&](int, float vx, float vy, float vz) { atomicAdd(&s_hist[cell_of(vx, vy, vz)], 1); });
    __syncthreads();
    // exclusive scan over the cells: PER consecutive cells per thread, wave scan, 16 wave totals
    constexpr int PER = ORDER_CELLS / NT;
    int loc[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        loc[k] = s_hist[tid * PER + k];
        sum += loc[k];
    }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < wave; ++k) base += s_wsum[k];
    int run = base + incl - sum;
    int* gb = grid + (size_t)b * GRID_WORDS;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        s_hist[tid * PER + k] = run;
        gb[tid * PER + k] = run;                        // cell -> first sorted position (the ball query walks these cell lists)
        run += loc[k];
    }
    if (tid == 0) {
        gb[ORDER_CELLS] = N;
        float* gf = reinterpret_cast<float*>(gb + ORDER_CELLS + 1);
        gf[0] = lo[0]; gf[1] = lo[1]; gf[2] = lo[2]; gf[3] = sc[0]; gf[4] = sc[1]; gf[5] = sc[2];
        gb[GRID_WORDS - 1] = 0;
    }
    __syncthreads();
    int* ob = order + (size_t)b * N;
    float4* sb = sorted + (size_t)b * N;
    for_points([&](int i, float vx, float vy, float vz) {
        const int p = atomicAdd(&s_hist[cell_of(vx, vy, vz)], 1);
        ob[p] = i;
        sb[p] = make_float4(vx, vy, vz, INFINITY);      // .w = running FPS distance
        if (rank) rank[(size_t)b * N + i] = p;
    });
}

// the plot's Morton order, sorted table and cell starts (+ the zeroed exchange area)
static void launch_spatial_order(const float* pos, int B, int N, int* order, float4* sorted, int* grid, unsigned* xchg,
                                 unsigned* ctl, hipStream_t st) {
    int* rank = reinterpret_cast<int*>(ctl + FPS_CTL_WORDS);          // the last B*N words of the workspace
    if (sn2_small_sort_wg(B) && N <= 16 * 1024)          // many plots: 256 threads per plot (common.h)
        hipLaunchKernelGGL((spatial_order_chunk_kernel<16, 256>), dim3(B), dim3(256), 0, st, pos, N, order, sorted, grid, xchg, ctl, rank);
    else if (N <= 8 * 1024)
        hipLaunchKernelGGL((spatial_order_chunk_kernel<8, 1024>), dim3(B), dim3(1024), 0, st, pos, N, order, sorted, grid, xchg, ctl, rank);
    else
        hipLaunchKernelGGL((spatial_order_chunk_kernel<16, 1024>), dim3(B), dim3(1024), 0, st, pos, N, order, sorted, grid, xchg, ctl, rank);
}

// canonical, monotone lower bound of sn2_d2(p, c) over all p inside the box [lo, hi]
__device__ __forceinline__ float sn2_box_d2(float lx, float ly, float lz, float hx, float hy, float hz, float cx, float cy,
                                            float cz) {
#pragma clang fp contract(off)
    const float dx = fmaxf(fmaxf(lx - cx, cx - hx), 0.f);
    const float dy = fmaxf(fmaxf(ly - cy, cy - hy), 0.f);
    const float dz = fmaxf(fmaxf(lz - cz, cz - hz), 0.f);
    const float xx = dx * dx;
    const float yy = dy * dy;
    const float zz = dz * dz;
    const float s = xx + yy;
    return s + zz;
}

// SPW = bucket slots per wave (bucket b lives in wave b % 16, slot b / 16); 16 waves.  The sorted points stay in
// global memory (an L2-resident 16 B x N table, one dwordx4 per lane fetches a whole bucket) and the table's .w lane is
// the point's RUNNING DISTANCE: it arrives with the coordinates in the same dwordx4 and is written back when it shrinks.
// Plain loads and stores are enough: a bucket is only ever touched by its own wave, and a CU's vector L1 is write-through
// and coherent with that CU's own stores (agent-scope `sc1` accesses were tried first: they bypass the XCD's L2 too and
// made the kernel 27 % slower).
// LDS holds only the bucket boxes and the per-wave exchange records (13 KB): with the distances in LDS (128 KB at
// N = 32768) a workgroup needed a CU with no other LDS user, and next to the feature kernels of the pipelined training
// step such a CU only turned up when another FPS workgroup retired -- two FPS passes in flight ran back to back.
// Registers hold no per-point state, so the round loop is a short body executed only for the buckets that can change
// (no 32-way unrolled branch chain: that version was instruction-fetch bound).
__device__ __forceinline__ void fps_st_dist(float4* t, int p, float v) { reinterpret_cast<float*>(t + p)[3] = v; }
#ifdef SN2_FPS_STAMPS
// diagnostic build only (never shipped): per-phase cycle totals of wave 0 of workgroup 0
__device__ unsigned long long g_fps_dbg[8];
__device__ unsigned long long g_fps_dbg2[32];
extern "C" int sn2_debug_fps_stamps2(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fps_dbg2), sizeof(g_fps_dbg2));
}
#define STAMP(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
extern "C" int sn2_debug_fps_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fps_dbg), sizeof(g_fps_dbg));
}
#else
#define STAMP(var)
#endif

template <int SPW, int NW>
__global__ __launch_bounds__(NW * 64) void fps_bucket_kernel(const float* __restrict__ pos, int N, int M,
                                                          const int* __restrict__ start, const int* __restrict__ order,
                                                          float4* sorted, int* __restrict__ idx_out,
                                                          float* __restrict__ cpos_soa, float* __restrict__ cpos_aos) {
    constexpr int NBK = SPW * NW;
    constexpr int SL = (SPW + 63) / 64;                // bucket slots per lane: slot s = 64 h + lane, h < SL
    static_assert(SPW <= 128 && NW <= 16, "at most two bucket slots per lane");
    __shared__ float s_box[6 * NBK];                   // bucket boxes, [component][wave][slot]: lane j reads word j of its
                                                       // wave's row => conflict-free (an AoS box layout cost 32-way conflicts)
    __shared__ float4 s_xchg[2][2][NW];                // per wave: (max distance, tie flag, -, -) and (x, y, z, sorted position)
    __shared__ unsigned s_win;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* px = pos + (size_t)b * 3 * N;
    const float* py = px + N;
    const float* pz = py + N;
    const int* ord = order + (size_t)b * N;
    float4* pts = sorted + (size_t)b * N;              // (x, y, z, running distance = +inf from spatial_order_kernel)
    float* my_box = s_box + wave * SPW;                // + component * NBK + slot
    for (int k = 0; k < SPW; ++k) {
        const int p = (k * NW + wave) * 64 + lane;     // position in the sorted order (bucket k*NW + wave)
        const bool v = p < N;
        const float4 q = v ? pts[p] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float lx = -wave_max_dpp(v ? -q.x : -INFINITY), ly = -wave_max_dpp(v ? -q.y : -INFINITY),
                    lz = -wave_max_dpp(v ? -q.z : -INFINITY);
        const float hx = wave_max_dpp(v ? q.x : -INFINITY), hy = wave_max_dpp(v ? q.y : -INFINITY),
                    hz = wave_max_dpp(v ? q.z : -INFINITY);
        if (lane == 0) {
            my_box[0 * NBK + k] = lx; my_box[1 * NBK + k] = ly; my_box[2 * NBK + k] = lz;
            my_box[3 * NBK + k] = hx; my_box[4 * NBK + k] = hy; my_box[5 * NBK + k] = hz;
        }
    }
    // lane j < SPW keeps the state of bucket slot j of this wave in registers: its maximal running distance, the point
    // attaining it (lowest lane on ties) and whether several points share the maximum (then the lowest ORIGINAL index must
    // be found the slow way)
    float mine[SL], bx[SL], by[SL], bz[SL];
    int bpos[SL];
    bool tiej[SL];
#pragma unroll
    for (int h = 0; h < SL; ++h) {
        const int sl = 64 * h + lane;
        mine[h] = (sl < SPW && (sl * NW + wave) * 64 < N) ? INFINITY : -1.f;
        bx[h] = by[h] = bz[h] = 0.f;
        bpos[h] = 0;
        tiej[h] = false;
    }
    int cur = start ? start[b] : 0;
    cur = cur < 0 ? 0 : (cur >= N ? N - 1 : cur);
    cur = __builtin_amdgcn_readfirstlane(cur);
    float cx = px[cur], cy = py[cur], cz = pz[cur];
    int cur_pos = -1;                                   // >= 0: the sample is sorted point cur_pos (index = ord[cur_pos])
    __syncthreads();

    // one dirty bucket: new distances, bucket maximum, its point -> the registers of lane k
    auto update = [&](int k, int p, const float4& q) {
        const float d = q.w;
        const float dd = sn2_d2(q.x, q.y, q.z, cx, cy, cz);
        const float nd = p < N ? fminf(d, dd) : -1.f;   // padding lanes of the last bucket never win
        if (p < N && nd < d) fps_st_dist(pts, p, nd);
        const float m = wave_max_dpp(nd);
        const unsigned long long bal = __ballot(nd == m);
        const int first = __ffsll((long long)bal) - 1;
        const float fx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.x), first));
        const float fy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.y), first));
        const float fz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.z), first));
#pragma unroll
        for (int h = 0; h < SL; ++h)
            if (k == 64 * h + lane) {
                mine[h] = m;
                tiej[h] = __popcll(bal) > 1;
                bx[h] = fx;
                by[h] = fy;
                bz[h] = fz;
                bpos[h] = p - lane + first;
            }
    };

    for (int i = 0; i < M; ++i) {
        if (tid == 0) {
            const int oi = cur_pos >= 0 ? ord[cur_pos] : cur;
            idx_out[(size_t)b * M + i] = oi;
            cpos_soa[((size_t)b * 3 + 0) * M + i] = cx;
            cpos_soa[((size_t)b * 3 + 1) * M + i] = cy;
            cpos_soa[((size_t)b * 3 + 2) * M + i] = cz;
            reinterpret_cast<float4*>(cpos_aos)[(size_t)b * M + i] = make_float4(cx, cy, cz, 0.f);
        }
        if (i == M - 1) break;
        STAMP(t0);
        // (a) which of this wave's buckets can change?  lane j tests slots j, j + 64, ...
        unsigned long long masks[SL];
#pragma unroll
        for (int h = 0; h < SL; ++h) {
            const int sl = 64 * h + lane;
            bool dirty = false;
            if (sl < SPW) {
                dirty = sn2_box_d2(my_box[0 * NBK + sl], my_box[1 * NBK + sl], my_box[2 * NBK + sl], my_box[3 * NBK + sl],
                                   my_box[4 * NBK + sl], my_box[5 * NBK + sl], cx, cy, cz) < mine[h];
            }
            masks[h] = __ballot(dirty);
        }
        STAMP(t1);
#ifdef SN2_FPS_STAMPS
        int ndirty = 0;
#pragma unroll
        for (int h = 0; h < SL; ++h) ndirty += __popcll(masks[h]);
#endif
        // (b) update the dirty buckets, four at a time so that their L2 loads overlap
#pragma unroll
        for (int h = 0; h < SL; ++h) {
            unsigned long long mask = masks[h];
            while (mask) {
                int k[4], p[4];
                bool on[4];
                float4 q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    on[u] = mask != 0;
                    k[u] = 64 * h + (on[u] ? __ffsll((long long)mask) - 1 : 0);
                    mask &= mask - 1;   // no-op when mask == 0
                    p[u] = (k[u] * NW + wave) * 64 + lane;
                    q[u] = pts[p[u] < N ? p[u] : 0];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (on[u]) update(k[u], p[u], q[u]);   // the round loop is VALU-issue bound: no duplicated work
            }
        }
        STAMP(t2);
        // (c) this wave's best bucket, then ONE barrier and the best of the 16 waves
        float lm = -1.f;
#pragma unroll
        for (int h = 0; h < SL; ++h) lm = fmaxf(lm, mine[h]);               // invalid slots hold -1
        const float wm = wave_max_dpp(lm);
        int nbest = 0;
        bool published = false;
#pragma unroll
        for (int h = 0; h < SL; ++h) {
            const unsigned long long balw = __ballot(mine[h] == wm);
            nbest += __popcll(balw);
        }
#pragma unroll
        for (int h = 0; h < SL; ++h) {
            const unsigned long long balw = __ballot(mine[h] == wm);
            if (balw && !published) {                                        // wave-uniform: the first slot group that has it
                const int slot = __ffsll((long long)balw) - 1;
                if (lane == slot) {
                    const int wtie = (nbest > 1) || tiej[h];
                    s_xchg[i & 1][0][wave] = make_float4(wm, __int_as_float(wtie), 0.f, 0.f);
                    s_xchg[i & 1][1][wave] = make_float4(bx[h], by[h], bz[h], __int_as_float(bpos[h]));
                }
                published = true;
            }
        }
        STAMP(t3);
        __syncthreads();
        STAMP(t4);
        const float4 e = s_xchg[i & 1][0][lane & (NW - 1)];
        const float4 r = s_xchg[i & 1][1][lane & (NW - 1)];
        const float V = wave_max_dpp(e.x);
        const unsigned balv = (unsigned)__ballot(e.x == V) & ((1u << NW) - 1u);
        const int ww = __ffs(balv) - 1;
        const bool gtie = (__popc(balv) > 1) || (__builtin_amdgcn_readlane(__float_as_int(e.y), ww) != 0);
        if (!gtie) {
            cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.x), ww));
            cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.y), ww));
            cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.z), ww));
            cur_pos = __builtin_amdgcn_readlane(__float_as_int(r.w), ww);
        } else {
            // rare exact tie of the maximal distance (duplicated points): lowest ORIGINAL index among all candidates
            if (tid == 0) s_win = 0xFFFFFFFFu;
            __syncthreads();
#pragma unroll
            for (int h = 0; h < SL; ++h) {
                unsigned long long cand = __ballot(mine[h] == V);
                while (cand) {
                    const int kk = 64 * h + __ffsll((long long)cand) - 1;
                    cand &= cand - 1;
                    const int pp = (kk * NW + wave) * 64 + lane;
                    const float d = pp < N ? pts[pp].w : -1.f;
                    unsigned oi = 0xFFFFFFFFu;
                    if (d == V) oi = (unsigned)ord[pp];
                    oi = wave_min_u32_dpp(oi);
                    if (lane == 0) atomicMin(&s_win, oi);
                }
            }
            __syncthreads();
            cur = __builtin_amdgcn_readfirstlane((int)s_win);
            cur_pos = -1;
            cx = px[cur];
            cy = py[cur];
            cz = pz[cur];
        }
#ifdef SN2_FPS_STAMPS
        STAMP(t5);
        if (b == 0 && tid == 0) {
            g_fps_dbg[0] += t1 - t0;
            g_fps_dbg[1] += t2 - t1;
            g_fps_dbg[2] += t3 - t2;
            g_fps_dbg[3] += t4 - t3;
            g_fps_dbg[4] += t5 - t4;
            g_fps_dbg[5] += (unsigned long long)ndirty;
            g_fps_dbg[6] += 1;
            g_fps_dbg[7] += gtie ? 1 : 0;
        }
#endif
    }
}


// ------------------------------------------------------------------------------------------------------------
// Speculative bucketed FPS (exact; round 2).  fps_bucket_kernel above pays, for EVERY sample, a sequential chain of
// test -> L2 load -> update -> wave reduction -> LDS exchange -> barrier -> arg-max (1.75 us per round at N = 32768, of
// which the 16-wave barrier + exchange + selection are more than half).  But consecutive FPS samples are far apart:
// sample t only lowers distances inside the ball of radius sqrt(D_t(max)) around it, and the next maxima sit in other
// holes of the sampling.  So one SUPER-ROUND accepts up to K samples from one arg-max pass:
//   let m_0 > m_1 > ... be the bucket maxima in strictly decreasing order, c_e the point attaining m_e.  c_0 is the next
//   sample.  c_e (e >= 1) is the sample after c_0..c_{e-1}  iff  (1) m_e is strictly larger than every other bucket
//   maximum left (m_e > m_{e+1}) and than the second-largest distance of its own bucket and of the buckets of
//   c_0..c_{e-1} (their other points only ever get smaller), and (2) c_e itself is untouched by c_0..c_{e-1}:
//   sn2_d2(c_e, c_j) >= m_e for all j < e (then fminf(D(c_e), d2) == D(c_e) bit for bit).  Under (1)+(2) c_e is the
//   unique arg-max of the running distances after the first e samples were applied -- without applying them.
//   The accepted prefix c_0..c_{j-1} is then applied to the dirty buckets in ONE pass (a bucket is loaded once and
//   min-ed with the samples whose box test it fails), and the barriers, the exchange and the arg-max are paid once per
//   super-round instead of once per sample.  Exact ties anywhere (equal maxima, duplicated points) end the prefix; a tie
//   for m_0 takes the lowest-ORIGINAL-index search of the old kernel.  Indices are bit-identical to the brute-force
//   kernel and to the oracle (tests/test_gpu_geometry.py), whatever K is.
// Per super-round: all waves test + update their dirty buckets and refresh (max, second max, arg-max point) of those
// buckets in LDS; barrier; wave 0 alone extracts the K+1 largest maxima (three sorted heads per lane, DPP wave max),
// runs the acceptance tests (lane e = candidate e), emits the samples; barrier.
// ------------------------------------------------------------------------------------------------------------
// wave64 max with the DPP modifier fused into v_max_f32 (the update_dpp builtin above costs a v_mov, a v_mov_dpp, an
// s_nop and the v_max per step: 24 instructions per reduction; this is 12).  All 64 lanes must be active.
__device__ __forceinline__ float wave_max_fused(float v) {
    asm("s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

template <int SPW, int NW, int K>
__global__ __launch_bounds__(NW * 64) void fps_spec_kernel(const float* __restrict__ pos, int N, int M,
                                                        const int* __restrict__ start, const int* __restrict__ order,
                                                        float4* sorted, int* __restrict__ idx_out,
                                                        float* __restrict__ cpos_soa, float* __restrict__ cpos_aos,
                                                        const unsigned* gate, unsigned* status) {
    constexpr int NBK = SPW * NW;
    constexpr int SL = (SPW + 63) / 64;                // bucket slots per lane of the owning wave
    constexpr int T = 4;                               // maxima every wave hands to the final selection
    static_assert(SPW <= 128 && NW <= 16 && K >= 2 && K <= 16 && NW * T <= 64 && SPW >= T, "limits");
    extern __shared__ __attribute__((aligned(16))) unsigned char fps_smem[];
    float4* s_pt = reinterpret_cast<float4*>(fps_smem);            // [NBK] arg-max point of the bucket: x, y, z, sorted position
    float4* s_acc = s_pt + NBK;                                    // [K] accepted samples of this super-round
    float* s_box = reinterpret_cast<float*>(s_acc + K);            // [6][NBK] bucket boxes, [component][wave][slot]
    float* s_val = s_box + 6 * NBK;                                // [NBK] largest running distance of the bucket (-1: empty)
    float* s_max2 = s_val + NBK;                                   // [NBK] >= the second largest (== s_val: treated as a tie)
    unsigned* s_q = reinterpret_cast<unsigned*>(s_max2 + NBK);     // [NBK] work queue of a super-round: bucket | samples << 11
    float2* s_top = reinterpret_cast<float2*>(s_q + NBK);          // [NW*T] per-wave maxima: (value, bucket | place << 16)
    float2* s_cand = s_top + NW * T;                               // [K + 2] the selection's candidates in order (+ the next one)
    int* s_keys = reinterpret_cast<int*>(s_cand + K + 2);          // [64] the selection's integer keys
    int* s_ctl = s_keys + 64;           // [0] accepted (0: tie search), [1] done, [2] tie value, [3] queue length
    unsigned* s_win = reinterpret_cast<unsigned*>(s_ctl + 4);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* px = pos + (size_t)b * 3 * N;
    const float* py = px + N;
    const float* pz = py + N;
    const int* ord = order + (size_t)b * N;
    float4* pts = sorted + (size_t)b * N;              // (x, y, z, running distance = +inf from spatial_order_kernel)
    if (gate) {
        // REPAIR launch behind fps_cluster_kernel (gate = that launch's control words): nothing to do unless one of its waits
        // gave up (gate[1] = the count); then every plot is sampled again by this kernel, which waits for nobody.  The running
        // distances the abandoned pass left in the sorted table go back to +inf first.
        const unsigned gave_up = __builtin_amdgcn_readfirstlane(gate[1]);
        if (gave_up == 0u) return;
        for (int p = tid; p < N; p += NW * 64) fps_st_dist(pts, p, INFINITY);
        if (b == 0 && tid == 0 && status) atomicAdd(status, gave_up);      // sticky: the host reads it where it synchronises anyway
        __threadfence_block();
        __syncthreads();
    }
    float* my_box = s_box + wave * SPW;                // + component * NBK + slot; bucket of (wave, slot) = slot * NW + wave
    for (int k = 0; k < SPW; ++k) {
        const int p = (k * NW + wave) * 64 + lane;     // position in the sorted order (bucket k*NW + wave)
        const bool v = p < N;
        const float4 q = v ? pts[p] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float lx = -wave_max_fused(v ? -q.x : -INFINITY), ly = -wave_max_fused(v ? -q.y : -INFINITY),
                    lz = -wave_max_fused(v ? -q.z : -INFINITY);
        const float hx = wave_max_fused(v ? q.x : -INFINITY), hy = wave_max_fused(v ? q.y : -INFINITY),
                    hz = wave_max_fused(v ? q.z : -INFINITY);
        if (lane == 0) {
            my_box[0 * NBK + k] = lx; my_box[1 * NBK + k] = ly; my_box[2 * NBK + k] = lz;
            my_box[3 * NBK + k] = hx; my_box[4 * NBK + k] = hy; my_box[5 * NBK + k] = hz;
        }
    }
    for (int g = tid; g < NBK; g += NW * 64) {
        s_val[g] = g * 64 < N ? INFINITY : -1.f;       // every real bucket is dirty for the first sample
        s_max2[g] = -1.f;
    }
    int cur = start ? start[b] : 0;
    cur = cur < 0 ? 0 : (cur >= N ? N - 1 : cur);
    cur = __builtin_amdgcn_readfirstlane(cur);
    // the accepted samples of the current super-round live in lanes 0..j-1 of (ax, ay, az) in EVERY wave
    float ax = px[cur], ay = py[cur], az = pz[cur];
    int j = 1, cnt = 1;
    int* out_idx = idx_out + (size_t)b * M;            // sorted positions are stored as -1 - position and translated at the end
    if (tid == 0) {
        out_idx[0] = cur;
        cpos_soa[((size_t)b * 3 + 0) * M] = ax;
        cpos_soa[((size_t)b * 3 + 1) * M] = ay;
        cpos_soa[((size_t)b * 3 + 2) * M] = az;
        reinterpret_cast<float4*>(cpos_aos)[(size_t)b * M] = make_float4(ax, ay, az, 0.f);
        s_ctl[3] = 0;
    }
    if (M <= 1) return;
    __syncthreads();

    while (true) {
        STAMP(t0);
        // (A) which of this wave's buckets can change, and through which of the j samples?  lane = bucket slot.  Dirty
        // buckets go into ONE queue of the workgroup, so that (B) can deal them out evenly: with the buckets tied to their
        // waves the slowest wave had twice the average load and the others waited for it.
#pragma unroll
        for (int h = 0; h < SL; ++h) {
            const int sl = 64 * h + lane;
            unsigned sm = 0u;
            if (sl < SPW) {
                const float b0 = my_box[0 * NBK + sl], b1 = my_box[1 * NBK + sl], b2 = my_box[2 * NBK + sl];
                const float b3 = my_box[3 * NBK + sl], b4 = my_box[4 * NBK + sl], b5 = my_box[5 * NBK + sl];
                const float mx = s_val[sl * NW + wave];
                for (int k = 0; k < j; ++k) {
                    const float sx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ax), k));
                    const float sy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ay), k));
                    const float sz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(az), k));
                    if (sn2_box_d2(b0, b1, b2, b3, b4, b5, sx, sy, sz) < mx) sm |= 1u << k;
                }
            }
            const unsigned long long bal = __ballot(sm != 0u);
            if (bal) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_ctl[3], __popcll(bal));
                base = __builtin_amdgcn_readfirstlane(base);
                if (sm != 0u) s_q[base + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u))] =
                    (unsigned)(sl * NW + wave) | (sm << 11);
            }
        }
        STAMP(t1);
        __syncthreads();
        // (B) the queue, four entries per wave in flight: new distances; if the bucket's maximum (or its holder) moved, the
        // new (max, second max, arg-max point) go to LDS
        const int qn = s_ctl[3];
        constexpr int QD = 4;                            // queue entries per wave in flight (8 measured no faster: the phase is VALU-bound)
        for (int i0 = wave; i0 < qn; i0 += QD * NW) {
            unsigned ent[QD];
            int p[QD];
            bool on[QD];
            float4 q[QD];
            float U[QD];
#pragma unroll
            for (int u = 0; u < QD; ++u) {
                const int i = i0 + u * NW;
                on[u] = i < qn;
                ent[u] = s_q[on[u] ? i : 0];
                const int g = (int)(ent[u] & 2047u);
                p[u] = g * 64 + lane;
                q[u] = pts[p[u] < N ? p[u] : 0];
                U[u] = s_val[g];
            }
#pragma unroll
            for (int u = 0; u < QD; ++u) {
                if (!on[u]) continue;
                const int g = (int)(ent[u] & 2047u);
                unsigned sm = ent[u] >> 11;
                const float d0 = q[u].w;
                float nd = d0;
                while (sm) {
                    const int k = __ffs(sm) - 1;
                    sm &= sm - 1;
                    const float sx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ax), k));
                    const float sy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ay), k));
                    const float sz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(az), k));
                    nd = fminf(nd, sn2_d2(q[u].x, q[u].y, q[u].z, sx, sy, sz));
                }
                const bool real = p[u] < N;
                const bool changed = real && nd < d0;
                if (changed) fps_st_dist(pts, p[u], nd);
#ifdef SN2_FPS_STAMPS
                if (b == 0 && wave == 0 && lane == 0) {
                    g_fps_dbg2[25] += 1;                                                     // entries
                    g_fps_dbg2[26] += __ballot(changed) != 0ull ? 1 : 0;                     // ... with a changed point
                    g_fps_dbg2[27] += (__ballot(changed && d0 == U[u]) == 0ull && U[u] != INFINITY) ? 0 : 1;   // ... slow path
                    g_fps_dbg2[28] += __popc(ent[u] >> 11);                                  // samples per entry
                    g_fps_dbg2[29] += __popcll(__ballot(changed));                           // changed points
                }
#endif
                // five of six queue entries change no point at all (the box test is necessary, not sufficient: fps_stamps.py)
                if (__ballot(changed) == 0ull) continue;
                // nothing of the bucket's maximum moved (no changed point held it): its (max, point) stand, and its
                // second-max entry stays an UPPER bound of the true one, which is all the acceptance tests need
                if (__ballot(changed && d0 == U[u]) == 0ull && U[u] != INFINITY) continue;
                if (!real) nd = -1.f;                               // padding lanes of the last bucket never win
                const float m1 = wave_max_fused(nd);
                const unsigned long long bal = __ballot(nd == m1);
                const int first = __ffsll((long long)bal) - 1;
                float m2 = m1;
                if (__popcll(bal) == 1) m2 = wave_max_fused(nd == m1 ? -1.f : nd);
                const float fx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q[u].x), first));
                const float fy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q[u].y), first));
                const float fz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q[u].z), first));
                if (lane == 0) {
                    s_val[g] = m1;
                    s_max2[g] = m2;
                    s_pt[g] = make_float4(fx, fy, fz, __int_as_float(g * 64 + first));
                }
            }
        }
        STAMP(t2);
        __syncthreads();
        STAMP(t3);
        // (C) every wave: the T largest maxima of its own buckets, in order -> s_top
        {
            float v[SL];
            int gid[SL];
#pragma unroll
            for (int h = 0; h < SL; ++h) {
                const int sl = 64 * h + lane;
                gid[h] = sl * NW + wave;
                v[h] = sl < SPW ? s_val[gid[h]] : -2.f;
            }
#pragma unroll
            for (int t = 0; t < T; ++t) {
                float lm = v[0];
#pragma unroll
                for (int h = 1; h < SL; ++h) lm = fmaxf(lm, v[h]);
                const float m = wave_max_fused(lm);
                const int owner = __ffsll((long long)__ballot(lm == m)) - 1;
                int hh = SL - 1;
#pragma unroll
                for (int h = SL - 2; h >= 0; --h) hh = v[h] == m ? h : hh;
                const int gsel = __builtin_amdgcn_readlane(gid[0] + hh * 64 * NW, owner);
                if (lane == owner) {
#pragma unroll
                    for (int h = 0; h < SL; ++h) if (h == hh) v[h] = -3.f;
                }
                if (lane == 0) s_top[wave * T + t] = make_float2(m, __int_as_float(gsel | (t << 16)));
            }
        }
        STAMP(t3b);
        __syncthreads();
        STAMP(t4);
        // (D) wave 0: the K+1 largest of the NW*T values in order, acceptance tests, emission -- written for latency: this
        // wave runs alone while the other fifteen wait (reduction chains, one per candidate, took 5300 clocks here).
        //   ranks: every entry counts the entries above it (integer keys = value bits with the low 6 bits replaced by the
        //   entry's place, so all keys differ; 16 broadcast LDS reads + 64 compare-and-adds, no chains) and drops itself
        //   into s_cand[rank];   tests: lane 8 e + j looks at the pair (candidate e, earlier candidate j);
        //   prefix length: scalar bit arithmetic on two ballots.
        if (wave == 0) {
            static_assert(K == 8, "the acceptance tests use an 8 x 8 lane grid");
            const float2 tp = s_top[lane < NW * T ? lane : 0];
            const float val = lane < NW * T ? tp.x : -3.f;
            const int key = (__float_as_int(val) & ~63) | (63 - lane);
            s_keys[lane] = key;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            int rank = 0;
#pragma unroll
            for (int q4 = 0; q4 < 16; ++q4) {
                const int4 k4 = reinterpret_cast<const int4*>(s_keys)[q4];
                rank += (k4.x > key ? 1 : 0) + (k4.y > key ? 1 : 0) + (k4.z > key ? 1 : 0) + (k4.w > key ? 1 : 0);
            }
            if (rank <= K) s_cand[rank] = make_float2(val, tp.y);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int e = lane >> 3, jj = lane & 7;
            const float2 ce = s_cand[e], cj = s_cand[jj], cn = s_cand[e + 1];
            const float me = ce.x;
            const int ie = __float_as_int(ce.y), ij = __float_as_int(cj.y);
            const float4 pte = s_pt[ie & 0xFFFF], ptj = s_pt[ij & 0xFFFF];
            const float m2e = s_max2[ie & 0xFFFF], m2j = s_max2[ij & 0xFFFF];
            // strictly above the next maximum even at the keys' 64-ulp resolution (then above every other entry too), own
            // bucket's second maximum below it, and its wave still has a known next value (a wave's fifth largest is unknown)
            const bool self_ok = (ie >> 16) < T - 1 && me > 0.f && (__float_as_int(me) >> 6) > (__float_as_int(cn.x) >> 6) && m2e < me;
            const bool pair_bad = jj < e && (sn2_d2(pte.x, pte.y, pte.z, ptj.x, ptj.y, ptj.z) < me || m2j >= me);
            const unsigned long long okm = __ballot(self_ok), badm = __ballot(pair_bad);
            int nj = 0;
#pragma unroll
            for (int c = 0; c < K; ++c) {
                const bool okc = ((okm >> (8 * c)) & 1ull) != 0 && ((badm >> (8 * c)) & 0xFFull) == 0;
                if (okc && nj == c) nj = c + 1;
            }
#ifdef SN2_FPS_STAMPS
            if (b == 0 && lane == 0) g_fps_dbg2[nj] += 1;
#endif
            const int rem = M - cnt;
            nj = nj > rem ? rem : nj;
            if (jj == 0 && e < nj) {
                s_acc[e] = make_float4(pte.x, pte.y, pte.z, 0.f);
                out_idx[cnt + e] = -1 - __float_as_int(pte.w);
                cpos_soa[((size_t)b * 3 + 0) * M + cnt + e] = pte.x;
                cpos_soa[((size_t)b * 3 + 1) * M + cnt + e] = pte.y;
                cpos_soa[((size_t)b * 3 + 2) * M + cnt + e] = pte.z;
                reinterpret_cast<float4*>(cpos_aos)[(size_t)b * M + cnt + e] = make_float4(pte.x, pte.y, pte.z, 0.f);
            }
            // no candidate accepted = the maximum is not unique at the keys' resolution: the tie search needs the TRUE maximum
            // (candidate 0 is only the first of the entries that share the top key)
            float vtop = 0.f;
            if (nj == 0) vtop = wave_max_fused(val);
            if (lane == 0) {
                s_ctl[0] = nj;
                s_ctl[1] = (cnt + (nj > 0 ? nj : 1) >= M) ? 1 : 0;
                s_ctl[2] = __float_as_int(vtop);
                s_ctl[3] = 0;
                *s_win = 0xFFFFFFFFu;
            }
        }
        STAMP(t5);
        __syncthreads();
        j = s_ctl[0];
        const int done = s_ctl[1];
#ifdef SN2_FPS_STAMPS
        STAMP(t6);
        if (b == 0 && tid == 0) {
            g_fps_dbg[0] += t1 - t0;
            g_fps_dbg[1] += t2 - t1;
            g_fps_dbg[2] += t5 - t4;
            g_fps_dbg[3] += (t3 - t2) + (t6 - t5) + (t4 - t3b);
            g_fps_dbg2[24] += t3b - t3;
            g_fps_dbg[4] += (unsigned long long)(j > 0 ? j : 1);
            g_fps_dbg[5] += (unsigned long long)qn;
            g_fps_dbg[6] += 1;
            g_fps_dbg[7] += j == 0 ? 1 : 0;
        }
#endif
        if (j > 0) {
            const float4 a = s_acc[lane < j ? lane : 0];
            ax = a.x; ay = a.y; az = a.z;
        } else {
            // exact tie of the maximal distance (duplicated points, ...): lowest ORIGINAL index among all points attaining it
            const float V = __int_as_float(s_ctl[2]);
#pragma unroll
            for (int h = 0; h < SL; ++h) {
                const int sl = 64 * h + lane;
                unsigned long long cand = __ballot(sl < SPW && s_val[sl * NW + wave] == V);
                while (cand) {
                    const int kk = 64 * h + __ffsll((long long)cand) - 1;
                    cand &= cand - 1;
                    const int pp = (kk * NW + wave) * 64 + lane;
                    const float d = pp < N ? pts[pp].w : -1.f;
                    unsigned oi = 0xFFFFFFFFu;
                    if (d == V) oi = (unsigned)ord[pp];
                    oi = wave_min_u32_dpp(oi);
                    if (lane == 0) atomicMin(s_win, oi);
                }
            }
            __syncthreads();
            cur = __builtin_amdgcn_readfirstlane((int)*s_win);
            ax = px[cur]; ay = py[cur]; az = pz[cur];
            if (tid == 0) {
                out_idx[cnt] = cur;
                cpos_soa[((size_t)b * 3 + 0) * M + cnt] = ax;
                cpos_soa[((size_t)b * 3 + 1) * M + cnt] = ay;
                cpos_soa[((size_t)b * 3 + 2) * M + cnt] = az;
                reinterpret_cast<float4*>(cpos_aos)[(size_t)b * M + cnt] = make_float4(ax, ay, az, 0.f);
            }
            j = 1;
        }
        cnt += j;
        if (done) break;
    }
    // sorted positions -> original indices (kept out of the loop: the load would sit on wave 0's critical path)
    __syncthreads();
    for (int i = tid; i < M; i += NW * 64) {
        const int v = out_idx[i];
        if (v < 0) out_idx[i] = ord[-1 - v];
    }
}

template <int SPW, int NW, int K>
static size_t fps_spec_lds_bytes() {
    return (size_t)(SPW * NW) * 16 + (size_t)K * 16 + (size_t)(SPW * NW) * 4 * 9 + (size_t)NW * 4 * 8 + (size_t)(K + 2) * 8 + 64 * 4 + 32;
}

#ifndef SN2_FPS_K
#define SN2_FPS_K 8       // samples a super-round may accept (measured at 16 x 32768 -> 1024: 4.9 accepted on average with 8, 5.5
                          // with 12, 5.6 with 16 -- the prefix mostly ends at a candidate touched by an earlier one -- while
                          // the selection's cost grows with K)
#endif
template <int SPW, int NW = 16>
static int launch_fps_bucket(const float* pos, int B, int N, int M, const int* start, int* ws, int* idx, float* cs,
                             float* ca, hipStream_t st, bool speculate = true, bool repair = false, unsigned* status = nullptr) {
    int* order = ws;                                               // B*N ints
    float4* sorted = reinterpret_cast<float4*>(ws + (size_t)B * N);   // B*N float4 (16-byte aligned: B*N*4 bytes offset
                                                                   // from a 16-byte aligned base with B*N % 4 == 0)
    int* grid = ws + (size_t)5 * B * N;                              // B*GRID_WORDS ints
    unsigned* xchg = reinterpret_cast<unsigned*>(grid + (size_t)B * GRID_WORDS);      // B*FPS_XCHG_WORDS + FPS_CTL_WORDS
    // repair = this launch follows fps_cluster_kernel on the same workspace: the tables are there, and the kernel returns at once
    // unless the control words say that the multi-workgroup pass gave up
    const unsigned* gate = repair ? xchg + (size_t)B * FPS_XCHG_WORDS : nullptr;
    if (!repair) launch_spatial_order(pos, B, N, order, sorted, grid, xchg, xchg + (size_t)B * FPS_XCHG_WORDS, st);
    if (speculate) {
        constexpr int K = SN2_FPS_K;
        const size_t lds = fps_spec_lds_bytes<SPW, NW, K>();
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fps_spec_kernel<SPW, NW, K>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        // (Round 4 tried the FPS waves at the highest wave priority, `s_setprio 3`: beside the feature kernels a super-round takes
        // three times as long as alone, and what the pass costs the step follows the time it is resident -- no effect, 0.7652
        // against 0.762 ms per step; round 3 had tried the opposite, the feature kernels raised: none either.)
        hipLaunchKernelGGL((fps_spec_kernel<SPW, NW, K>), dim3(B), dim3(NW * 64), lds, st, pos, N, M, start, (const int*)order,
                           sorted, idx, cs, ca, gate, status);
        SN2_RETURN_LAUNCH();
    }
    hipLaunchKernelGGL((fps_bucket_kernel<SPW, NW>), dim3(B), dim3(NW * 64), 0, st, pos, N, M, start, (const int*)order,
                       sorted, idx, cs, ca);
    SN2_RETURN_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// Multi-workgroup speculative FPS (exact; round 3).  fps_spec_kernel keeps a whole plot in ONE workgroup: 16 plots use 16
// of 256 CUs, and two thirds of a super-round are work that only scales with the buckets a workgroup owns -- the box tests
// (A) and the dirty-bucket updates (B: VALU-issue bound on one CU's four SIMDs).  Here P workgroups share a plot:
//   * bucket g (64 consecutive points of the Morton order) belongs to workgroup g % P -- interleaved, so the ~17 buckets a
//     new sample dirties spread evenly over the P workgroups; each workgroup keeps (max, second max, arg-max point) of
//     its NBL = ceil(buckets / P) buckets in LDS and runs phases (A) and (B) of fps_spec_kernel on them;
//   * per super-round ONE exchange through L2: wave 0 of every workgroup publishes its TP = 8 largest bucket maxima (value,
//     second max, point) and a bound on everything it did not publish as 8-byte {tag = super-round, value} granules
//     (agent-scope relaxed stores = `global_store_dwordx2 sc1`, the data is its own flag: cdna_hip_programming.md
//     Guideline 16, form R2), sweeps the granules of all P workgroups until every tag matches, and then runs the SAME
//     selection on the SAME 8 P records as its peers: every workgroup derives the same accepted samples, no broadcast;
//   * the local top-8 come from a THRESHOLD, not from reduction chains: tau = the value of the (accepted + 16)-th merged
//     candidate of the previous super-round (running distances only ever shrink, so at most that many buckets of the whole
//     plot can still reach tau); the buckets >= tau are compacted with two ballots and ranked among themselves (a dozen
//     integer compares per lane).  Whatever tau is, the result is exact: a workgroup states `bound` = tau (all its other
//     buckets are below) or its 8-th value (when more than 8 reach tau), a candidate is accepted only above every
//     workgroup's bound, and a super-round that finds nothing above the bounds lowers tau and tries again (more than 64
//     survivors: eight wave-max rounds instead of the ranking);
//   * acceptance tests, exact-tie search (lowest ORIGINAL index, one more granule per workgroup) and emitted samples are
//     those of fps_spec_kernel, bit for bit (tests/test_gpu_geometry.py holds all kernels to each other and to the oracle).
// Residency: a workgroup waits only for the P-1 peers of its plot.  Plots are handed out by a ticket counter in arrival
// order (not by blockIdx), so the peers of every resident workgroup are resident too or are the very next workgroups to
// start -- no dispatch-order assumption; every spin is bounded (`spin_limit` sweeps, then the kernel gives up, counts the
// timeout in control word 1 and exits WITHOUT writing its samples: never a hang).  Nothing guarantees that the B * P workgroups
// are resident together (another stream's or another process's kernels may hold the CUs a peer needs), so every launch of
// this kernel is followed, on the same stream, by the single-workgroup kernel as a REPAIR launch (fps_spec_kernel with
// `gate`): it returns at once when control word 1 is zero and samples every plot again when it is not -- the results are
// right either way, and the count reaches the host through `status` (sn2_fps_status).
// ------------------------------------------------------------------------------------------------------------
typedef unsigned long long fc_u64;
constexpr int FC_TP = 8;                     // records a workgroup publishes per super-round
constexpr int FC_PARITY_U64 = 512;           // granules per parity (6 * 8P record fields + P bounds + P tie words <= 400)
constexpr unsigned FC_SPIN_LIMIT = 1u << 18; // sweeps (~1 us each) before a wait gives up (a resident peer answers within a few)
#ifndef SN2_FC_FLAG_SLEEP
#define SN2_FC_FLAG_SLEEP 3
#endif
#ifndef SN2_FC_TAU_KEEP
#define SN2_FC_TAU_KEEP 15                   // candidates beyond the accepted ones that stay above the next threshold
#endif
static_assert(2 * 2 * FC_PARITY_U64 * 2 == FPS_XCHG_WORDS, "exchange area: two copies x two parities");
constexpr int FC_COPY_U64 = 2 * FC_PARITY_U64;      // copy 0: plain stores (same-XCD peers see them in their L2), copy 1: write-through

#ifndef SN2_FC_DUAL
#define SN2_FC_DUAL 0
#endif
constexpr bool FC_DUAL = SN2_FC_DUAL != 0;
// (FC_DUAL, measured and switched off: see below) one granule goes out TWICE: a plain store into copy 0 (it stays in this XCD's L2, where a peer on the SAME XCD finds it with an
// L1-bypassing load a few hundred clocks later) and a write-through `sc1` store into copy 1 (what a peer on ANOTHER XCD sees,
// ~1000 clocks later).  A reader takes whichever copy carries the tag first: which one it is depends on placement, the value
// does not -- both hold the same 8 bytes, each written by one store.
__device__ __forceinline__ void fc_store(fc_u64* g, unsigned epoch, unsigned v) {
    const fc_u64 x = ((fc_u64)epoch << 32) | (fc_u64)v;
    if (FC_DUAL) __hip_atomic_store(g, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(g + FC_COPY_U64, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// both copies of a granule -> the one that carries `epoch` (or one that does not: the caller checks the tag)
__device__ __forceinline__ fc_u64 fc_pick(fc_u64 a, fc_u64 b, unsigned epoch) { return (FC_DUAL && (unsigned)(a >> 32) == epoch) ? a : b; }
__device__ __forceinline__ fc_u64 fc_load(const fc_u64* g) {
    return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// max over lanes 0..7 (three DPP steps inside the first row), broadcast to the wave
__device__ __forceinline__ float max_of_first8(float v) {
    asm("s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
}

template <int SPW, int NW, int P, bool LP>
__global__ __launch_bounds__(NW * 64) void fps_cluster_kernel(const float* __restrict__ pos, int B, int N, int M,
                                                           const int* __restrict__ start, const int* __restrict__ order,
                                                           float4* sorted, int* __restrict__ idx_out,
                                                           float* __restrict__ cpos_soa, float* __restrict__ cpos_aos,
                                                           fc_u64* xchg_all, unsigned* ctl, int log_cap, int tau_keep,
                                                           unsigned spin_limit) {
    constexpr int NBL = SPW * NW;                      // buckets of this workgroup: local bucket lb = slot * NW + wave
    constexpr int SL = (SPW + 63) / 64;                // bucket slots per lane of the owning wave
    constexpr int VL = (NBL + 63) / 64;                // bucket values per lane of wave 0 in the selection
    constexpr int K = 8, TP = FC_TP, NE = TP * P;      // accepted per super-round, records per workgroup, merged records
    constexpr int SPWP = SPW < 64 ? SPW : 64;          // phase A: lane = (slot, sample): SPWP slots x KPL samples per pass
    constexpr int KPL = 64 / SPWP < 8 ? 64 / SPWP : 8;
    static_assert(NE <= 64 && P <= 8 && SPW <= 128 && NBL <= 2048 && NW <= 16 && 6 * NE + 2 * P <= FC_PARITY_U64, "limits");
    static_assert((SPWP & (SPWP - 1)) == 0 && TP <= 15, "slots per wave: a power of two; the record count rides in 4 bits");
    extern __shared__ __attribute__((aligned(16))) unsigned char fc_smem[];
    float4* s_pt = reinterpret_cast<float4*>(fc_smem);             // [NBL] arg-max point of the bucket: x, y, z, sorted position
    float4* s_acc = s_pt + NBL;                                    // [K] accepted samples of this super-round
    float4* s_crA = s_acc + K;                                     // [NE + 16] merged records in order: (value, second max, position, -)
    float4* s_crB = s_crA + NE + 16;                               // [NE + 16] ... (x, y, z, -)
    float4* s_ownA = s_crB + NE + 16;                              // [TP] this workgroup's published records (not re-read from memory)
    float4* s_ownB = s_ownA + TP;                                  // [TP]
    float* s_box = reinterpret_cast<float*>(s_ownB + TP);          // [6][NBL] bucket boxes, [component][wave][slot]
    float* s_val = s_box + 6 * NBL;                                // [NBL] largest running distance of the bucket (-1: empty)
    float* s_max2 = s_val + NBL;                                   // [NBL] >= the second largest (== s_val: treated as a tie)
    unsigned* s_q = reinterpret_cast<unsigned*>(s_max2 + NBL);     // [NBL] work queue: local bucket | samples << 11
    int* s_keys = reinterpret_cast<int*>(s_q + NBL);               // [64 + 8] integer keys of a ranking (+ the pad of its last quad)
    int* s_ctl = s_keys + 72;                                      // [0] accepted, [1] done, [2] tie value, [3] queue length,
                                                                   // [4] mode (0 accept, 1 tie search, 2 lower tau, 3 gave up),
                                                                   // [5] tie winner, [7] ticket
    unsigned* s_win = reinterpret_cast<unsigned*>(s_ctl + 8);
    float4* s_log = reinterpret_cast<float4*>(s_win + 4);          // [log_cap] every sample of the plot (x, y, z, -1 - position | index):
                                                                   // written out once at the end (log_cap = 0: emitted as they come)
    float4* s_pts = s_log + log_cap;                               // LP: [NBL * 64] this workgroup's points (x, y, z, running distance):
                                                                   // the updates of phase B never leave the CU
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // which plot, which part: by XCD.  Plot c belongs to XCD c % 8; a workgroup takes the next free part of its XCD's plots (a
    // per-XCD arrival counter), so the P workgroups of a plot share an L2 and exchange through it.  HIP promises nothing about
    // placement: a workgroup whose XCD is full waits until all B * P have registered and takes the first hole left on another
    // XCD (its exchange then runs through the write-through copy: slower, same result).  ctl: [0] overflow tickets,
    // [1] timeouts, [16 + x] arrivals on XCD x.
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        const unsigned t = atomicAdd(&ctl[16 + xcc], 1u);
        const unsigned mine = xcc < (unsigned)B ? ((unsigned)B - 1u - xcc) / 8u + 1u : 0u;     // plots of this XCD
        int slot = -1;
        if (t < mine * P) {
            slot = (int)((xcc + 8u * (t / P)) * P + t % P);
        } else {
            unsigned o = atomicAdd(&ctl[0], 1u);
            unsigned n[8];
            for (unsigned spins = 0; spins < spin_limit; ++spins) {
                unsigned sum = 0;
                for (int x = 0; x < 8; ++x) {
                    n[x] = __hip_atomic_load(&ctl[16 + x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sum += n[x];
                }
                if (sum == (unsigned)(B * P)) {
                    for (int c = 0; c < B && slot < 0; ++c) {
                        const unsigned before = (unsigned)(c / 8) * P, nx = n[c & 7];
                        const unsigned have = nx <= before ? 0u : (nx - before < (unsigned)P ? nx - before : (unsigned)P);
                        if (o < (unsigned)P - have) slot = c * P + (int)(have + o);
                        else o -= (unsigned)P - have;
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
            if (slot < 0) atomicAdd(&ctl[1], 1u);
        }
        s_ctl[7] = slot;
    }
    __syncthreads();
    const int ticket = s_ctl[7];
    if (ticket < 0) return;
    const int b = ticket / P, part = ticket - b * P;
    const bool use_log = log_cap >= M;
    const float* px = pos + (size_t)b * 3 * N;
    const float* py = px + N;
    const float* pz = py + N;
    const int* ord = order + (size_t)b * N;
    float4* pts = sorted + (size_t)b * N;              // (x, y, z, running distance = +inf from spatial_order_kernel)
    fc_u64* xchg = xchg_all + (size_t)b * (2 * FC_COPY_U64);
    float* my_box = s_box + wave * SPW;                // + component * NBL + slot
    for (int k = 0; k < SPW; ++k) {
        const int p = ((k * NW + wave) * P + part) * 64 + lane;       // sorted position: global bucket (k*NW + wave)*P + part
        const bool v = p < N;
        const float4 q = v ? pts[p] : make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (LP) s_pts[(k * NW + wave) * 64 + lane] = q;
        const float lx = -wave_max_fused(v ? -q.x : -INFINITY), ly = -wave_max_fused(v ? -q.y : -INFINITY),
                    lz = -wave_max_fused(v ? -q.z : -INFINITY);
        const float hx = wave_max_fused(v ? q.x : -INFINITY), hy = wave_max_fused(v ? q.y : -INFINITY),
                    hz = wave_max_fused(v ? q.z : -INFINITY);
        if (lane == 0) {
            my_box[0 * NBL + k] = lx; my_box[1 * NBL + k] = ly; my_box[2 * NBL + k] = lz;
            my_box[3 * NBL + k] = hx; my_box[4 * NBL + k] = hy; my_box[5 * NBL + k] = hz;
        }
    }
    for (int lb = tid; lb < NBL; lb += NW * 64) {
        s_val[lb] = (lb * P + part) * 64 < N ? INFINITY : -1.f;     // every real bucket is dirty for the first sample
        s_max2[lb] = -1.f;
    }
    int cur = start ? start[b] : 0;
    cur = cur < 0 ? 0 : (cur >= N ? N - 1 : cur);
    cur = __builtin_amdgcn_readfirstlane(cur);
    // the accepted samples of the current super-round: s_acc[0..j-1], and lanes 0..j-1 of (ax, ay, az) in EVERY wave
    float ax = px[cur], ay = py[cur], az = pz[cur];
    int j = 1, cnt = 1;
    unsigned epoch = 0;
    float tau = -1.f;                                  // wave 0's threshold for the next selection
    bool gave_up = false;                              // a wait ran out: leave without writing (the repair launch samples the plot)
    int* out_idx = idx_out + (size_t)b * M;
    // one sample of the plot: (x, y, z) and its code = -1 - sorted position (translated at the end) or the original index
    auto emit = [&](int at, float x, float y, float z, int code) {
        if (use_log) {
            s_log[at] = make_float4(x, y, z, __int_as_float(code));
        } else {
            out_idx[at] = code;
            cpos_soa[((size_t)b * 3 + 0) * M + at] = x;
            cpos_soa[((size_t)b * 3 + 1) * M + at] = y;
            cpos_soa[((size_t)b * 3 + 2) * M + at] = z;
            reinterpret_cast<float4*>(cpos_aos)[(size_t)b * M + at] = make_float4(x, y, z, 0.f);
        }
    };
    if (tid == 0) {
        if (part == 0) emit(0, ax, ay, az, cur);
        s_acc[0] = make_float4(ax, ay, az, 0.f);
        s_ctl[3] = 0;
    }
    if (M <= 1 && !use_log) return;
    __syncthreads();
    // phase A: lane = (slot asl, sample group akk), slot-major so that the hits of one slot are contiguous ballot bits
    const int asl = lane / KPL, akk = lane & (KPL - 1);

#ifdef SN2_FC_STAMPS
    unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned n_rounds = 0, n_acc = 0, n_fb = 0, n_m1 = 0, n_m2 = 0, n_surv = 0, n_spins = 0;
#define FCSTAMP(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#else
#define FCSTAMP(var)
#endif
    while (M > 1) {
        ++epoch;
        fc_u64* xg = xchg + (epoch & 1u) * FC_PARITY_U64;
        FCSTAMP(t0);
        // (A) which of this wave's buckets can change, and through which of the j samples?  lane = (slot, sample): every lane
        // runs ONE box test per pass of KPL samples (a lane per slot ran j of them in sequence: 1100 clocks whatever SPW was)
#pragma unroll
        for (int h = 0; h < SL; ++h) {
            const int sl = 64 * h + asl;
            const bool slot_ok = asl < SPWP && sl < SPW;
            float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f, b4 = 0.f, b5 = 0.f, mx = -1.f;
            if (slot_ok) {
                b0 = my_box[0 * NBL + sl]; b1 = my_box[1 * NBL + sl]; b2 = my_box[2 * NBL + sl];
                b3 = my_box[3 * NBL + sl]; b4 = my_box[4 * NBL + sl]; b5 = my_box[5 * NBL + sl];
                mx = s_val[sl * NW + wave];
            }
            unsigned sm = 0u;
            for (int k0 = 0; k0 < j; k0 += KPL) {
                const int k = k0 + akk;
                bool hit = false;
                if (slot_ok && k < j) {
                    const float4 sp = s_acc[k];
                    hit = sn2_box_d2(b0, b1, b2, b3, b4, b5, sp.x, sp.y, sp.z) < mx;
                }
                const unsigned long long hb = __ballot(hit);
                sm |= ((unsigned)(hb >> ((asl * KPL) & 63)) & ((1u << KPL) - 1u)) << k0;     // the KPL hits of this lane's slot
            }
            if (akk != 0 || !slot_ok) sm = 0u;           // one lane per slot pushes
            const unsigned long long bal = __ballot(sm != 0u);
            if (bal) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_ctl[3], __popcll(bal));
                base = __builtin_amdgcn_readfirstlane(base);
                if (sm != 0u) s_q[base + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u))] =
                    (unsigned)(sl * NW + wave) | (sm << 11);
            }
        }
        FCSTAMP(t1);
        __syncthreads();
        FCSTAMP(t2);
        // (B) the queue, four entries per wave in flight (fps_spec_kernel's phase B on this workgroup's buckets)
        const int qn = s_ctl[3];
        constexpr int QD = 4;
        for (int i0 = wave; i0 < qn; i0 += QD * NW) {
            unsigned ent[QD];
            int p[QD];
            bool on[QD];
            float4 q[QD];
            float U[QD];
#pragma unroll
            for (int u = 0; u < QD; ++u) {
                const int i = i0 + u * NW;
                on[u] = i < qn;
                ent[u] = s_q[on[u] ? i : 0];
                const int lb = (int)(ent[u] & 2047u);
                p[u] = (lb * P + part) * 64 + lane;
                if constexpr (LP) q[u] = s_pts[lb * 64 + lane]; else q[u] = pts[p[u] < N ? p[u] : 0];
                U[u] = s_val[lb];
            }
#pragma unroll
            for (int u = 0; u < QD; ++u) {
                if (!on[u]) continue;
                const int lb = (int)(ent[u] & 2047u);
                unsigned sm = ent[u] >> 11;
                const float d0 = q[u].w;
                float nd = d0;
                while (sm) {
                    const int k = __ffs(sm) - 1;
                    sm &= sm - 1;
                    const float sx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ax), k));
                    const float sy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ay), k));
                    const float sz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(az), k));
                    nd = fminf(nd, sn2_d2(q[u].x, q[u].y, q[u].z, sx, sy, sz));
                }
                const bool real = p[u] < N;
                const bool changed = real && nd < d0;
                if (changed) {
                    if constexpr (LP) reinterpret_cast<float*>(s_pts + lb * 64 + lane)[3] = nd; else fps_st_dist(pts, p[u], nd);
                }
                if (__ballot(changed) == 0ull) continue;
                if (__ballot(changed && d0 == U[u]) == 0ull && U[u] != INFINITY) continue;
                if (!real) nd = -1.f;                               // padding lanes of the last bucket never win
                const float m1 = wave_max_fused(nd);
                const unsigned long long bal = __ballot(nd == m1);
                const int first = __ffsll((long long)bal) - 1;
                float m2 = m1;
                if (__popcll(bal) == 1) m2 = wave_max_fused(nd == m1 ? -1.f : nd);
                const float fx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q[u].x), first));
                const float fy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q[u].y), first));
                const float fz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q[u].z), first));
                if (lane == 0) {
                    s_val[lb] = m1;
                    s_max2[lb] = m2;
                    s_pt[lb] = make_float4(fx, fy, fz, __int_as_float((lb * P + part) * 64 + first));
                }
            }
        }
        FCSTAMP(t3);
        __syncthreads();
        FCSTAMP(t4);
        // (C + D) wave 0: this workgroup's records, the exchange, the selection (identical in all P workgroups).  ONE wave issues
        // an instruction every 4-5 clocks whatever its kind: this block is written for instruction count.
        if (wave == 0) {
            // one record leaves the workgroup: as granules for the peers and into LDS for ourselves
            auto publish = [&](int r, float val, float m2, const float4& pt) {
                fc_u64* rg = xg + part * TP + r;       // record (part, r), field f at xg[f * NE + part * TP + r]
                fc_store(rg + 0 * NE, epoch, __float_as_uint(val));
                fc_store(rg + 1 * NE, epoch, __float_as_uint(m2));
                fc_store(rg + 2 * NE, epoch, __float_as_uint(pt.w));
                fc_store(rg + 3 * NE, epoch, __float_as_uint(pt.x));
                fc_store(rg + 4 * NE, epoch, __float_as_uint(pt.y));
                fc_store(rg + 5 * NE, epoch, __float_as_uint(pt.z));
                s_ownA[r] = make_float4(val, m2, pt.w, 0.f);
                s_ownB[r] = make_float4(pt.x, pt.y, pt.z, 0.f);
            };
            float v[VL], m2v[VL];
            float4 ptv[VL];
            bool surv[VL];
            unsigned long long bal[VL];
            int n = 0;
#pragma unroll
            for (int h = 0; h < VL; ++h) {
                const int lb = 64 * h + lane;
                v[h] = lb < NBL ? s_val[lb] : -1.f;
            }
#pragma unroll
            for (int h = 0; h < VL; ++h) {
                surv[h] = v[h] >= tau && v[h] >= 0.f;
                bal[h] = __ballot(surv[h]);
                n += __popcll(bal[h]);
                m2v[h] = -1.f;
                ptv[h] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (surv[h] && n <= 64) {              // what a published record carries (read now: off the ranking's path)
                    m2v[h] = s_max2[64 * h + lane];
                    ptv[h] = s_pt[64 * h + lane];
                }
            }
            float bound = tau;                         // every bucket of this workgroup that is not published is below `bound`
            int npub = n < TP ? n : TP;                // records this workgroup publishes (the peers ignore the rest of its area)
            if (n <= 64) {
                // the survivors ranked among themselves: integer keys = value bits with the low 6 bits replaced by the
                // survivor's place (all keys differ), every survivor counts the keys above its own
                int key[VL], rank[VL];
                if (n > TP && lane < 4) s_keys[n + lane] = INT_MIN;
                int base = 0;
#pragma unroll
                for (int h = 0; h < VL; ++h) {
                    key[h] = INT_MAX;
                    rank[h] = 0;
                    if (surv[h]) {
                        const int si = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal[h] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal[h], 0u));
                        key[h] = (__float_as_int(v[h]) & ~63) | (63 - si);
                        rank[h] = si;                  // n <= TP: every survivor is published, in any order (the merge ranks them)
                        if (n > TP) s_keys[si] = key[h];
                    }
                    base += __popcll(bal[h]);
                }
                if (n > TP) {
#pragma unroll
                    for (int h = 0; h < VL; ++h) rank[h] = 0;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
                    for (int q4 = 0; q4 < (n + 3) / 4; ++q4) {
                        const int4 k4 = reinterpret_cast<const int4*>(s_keys)[q4];
#pragma unroll
                        for (int h = 0; h < VL; ++h)
                            rank[h] += (k4.x > key[h] ? 1 : 0) + (k4.y > key[h] ? 1 : 0) + (k4.z > key[h] ? 1 : 0) + (k4.w > key[h] ? 1 : 0);
                    }
                }
                float b7 = -1.f;
#pragma unroll
                for (int h = 0; h < VL; ++h) {
                    if (surv[h] && rank[h] < TP) publish(rank[h], v[h], m2v[h], ptv[h]);
                    if (surv[h] && rank[h] == TP - 1) b7 = v[h];
                }
                if (n > TP) {                          // the unpublished survivors are <= the TP-th value
                    const int l7 = __ffsll((long long)__ballot(b7 >= 0.f)) - 1;
                    bound = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b7), l7));
                }
            } else {
                // more than 64 buckets reach tau (the first super-rounds, or after tau was lowered): TP wave-max rounds
                float pval = -1.f;
                int plb = 0;
                npub = 0;
#pragma unroll 1
                for (int t = 0; t < TP; ++t) {
                    float lm = v[0];
#pragma unroll
                    for (int h = 1; h < VL; ++h) lm = fmaxf(lm, v[h]);
                    const float m = wave_max_fused(lm);
                    const int owner = __ffsll((long long)__ballot(lm == m)) - 1;
                    int hh = VL - 1;
#pragma unroll
                    for (int h = VL - 2; h >= 0; --h) hh = v[h] == m ? h : hh;
                    const int lbsel = __builtin_amdgcn_readlane(64 * hh + lane, owner);
                    if (lane == owner) {
#pragma unroll
                        for (int h = 0; h < VL; ++h) if (h == hh) v[h] = -3.f;
                    }
                    if (lane == t) {
                        pval = m;
                        plb = lbsel;
                    }
                    if (m >= 0.f) {
                        npub = t + 1;
                        bound = m;
                    }
                }
                if (lane < npub) publish(lane, pval, s_max2[plb], s_pt[plb]);
            }
            // the header: the bound (its low 4 bits give way to the record count: only bits 6 and up are ever compared)
            if (lane == 0) fc_store(xg + 6 * NE + part, epoch, (__float_as_uint(bound) & ~15u) | (unsigned)npub);
            // sweep: lane l < NE takes record l of the merged set and its workgroup's header, lane l < P also header l (for the
            // bounds); our own records come from LDS.  A record counts once its header and -- if the header lists it -- its
            // six fields carry this super-round's tag.
            fc_u64 x[6] = {0, 0, 0, 0, 0, 0}, hw = 0, hb = 0;
            const int rw = lane / TP, rr = lane - rw * TP;
            const bool peer_rec = lane < NE && rw != part, peer_bnd = lane < P && lane != part;
            bool failed = false;
            FCSTAMP(t5);
#ifdef SN2_FC_STAMPS
            n_surv += n; n_fb += n > 64 ? 1 : 0;
#endif
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
                if (peer_rec) {
                    const fc_u64 h0 = FC_DUAL ? fc_load(xg + 6 * NE + rw) : 0ull, h1 = fc_load(xg + FC_COPY_U64 + 6 * NE + rw);
                    fc_u64 y[6];
#pragma unroll
                    for (int f = 0; f < 6; ++f) {
                        x[f] = FC_DUAL ? fc_load(xg + f * NE + lane) : 0ull;
                        y[f] = fc_load(xg + FC_COPY_U64 + f * NE + lane);
                    }
                    hw = fc_pick(h0, h1, epoch);
#pragma unroll
                    for (int f = 0; f < 6; ++f) x[f] = fc_pick(x[f], y[f], epoch);
                    ok = (unsigned)(hw >> 32) == epoch;
                    if (rr < (int)((unsigned)hw & 15u)) {
#pragma unroll
                        for (int f = 0; f < 6; ++f) ok &= (unsigned)(x[f] >> 32) == epoch;
                    }
                }
                if (peer_bnd) {
                    hb = fc_pick(FC_DUAL ? fc_load(xg + 6 * NE + lane) : 0ull, fc_load(xg + FC_COPY_U64 + 6 * NE + lane), epoch);
                    ok &= (unsigned)(hb >> 32) == epoch;
                }
                if (__ballot(!ok) == 0ull) break;
                if (++spins > spin_limit) {
                    failed = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            FCSTAMP(t6);
#ifdef SN2_FC_STAMPS
            n_spins += spins;
#endif
            float4 rA = make_float4(__uint_as_float((unsigned)x[0]), __uint_as_float((unsigned)x[1]), __uint_as_float((unsigned)x[2]), 0.f);
            float4 rB = make_float4(__uint_as_float((unsigned)x[3]), __uint_as_float((unsigned)x[4]), __uint_as_float((unsigned)x[5]), 0.f);
            bool valid = peer_rec && rr < (int)((unsigned)hw & 15u);
            if (lane < NE && !peer_rec && rr < npub) {
                rA = s_ownA[rr];
                rB = s_ownB[rr];
                valid = true;
            }
            const float val = rA.x;
            const unsigned long long vb = __ballot(valid);
            const int nvalid = __popcll(vb);
            // the valid records ranked by value (keys as above): rank = place in the merged order
            const int key = (__float_as_int(val) & ~63) | (63 - lane);
            if (lane < 4) s_keys[nvalid + lane] = INT_MIN;
            if (valid) s_keys[__builtin_amdgcn_mbcnt_hi((unsigned)(vb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)vb, 0u))] = key;
            if (lane < 9) s_crA[nvalid + lane] = make_float4(-1.f, -1.f, 0.f, 0.f);       // sentinels behind the valid records
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            const float Ub = max_of_first8(lane < P ? __uint_as_float((lane == part ? __float_as_uint(bound) : (unsigned)hb) & ~15u) : -1.f);
            __builtin_amdgcn_wave_barrier();
            int rank = 0;
#pragma unroll 2
            for (int q4 = 0; q4 < (nvalid + 3) / 4; ++q4) {
                const int4 k4 = reinterpret_cast<const int4*>(s_keys)[q4];
                rank += (k4.x > key ? 1 : 0) + (k4.y > key ? 1 : 0) + (k4.z > key ? 1 : 0) + (k4.w > key ? 1 : 0);
            }
            if (valid) {
                s_crA[rank] = rA;
                s_crB[rank] = rB;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // acceptance tests: lane 8 e + jj looks at the pair (candidate e, earlier candidate jj)
            const int e = lane >> 3, jj = lane & 7;
            const float4 Ae = s_crA[e], Be = s_crB[e], Aj = s_crA[jj], Bj = s_crB[jj], An = s_crA[e + 1];
            const float me = Ae.x;
            const float nxt = fmaxf(An.x, Ub);          // the next known maximum, or what an unpublished bucket may still hold
            const bool self_ok = me > 0.f && (__float_as_int(me) >> 6) > (__float_as_int(nxt) >> 6) && Ae.y < me;
            const bool pair_bad = jj < e && (sn2_d2(Be.x, Be.y, Be.z, Bj.x, Bj.y, Bj.z) < me || Aj.y >= me);
            // candidate c stands iff bit 8c of okm is set and byte c of badm is empty; accepted = the leading run of those
            const unsigned long long okm = __ballot(self_ok);
            unsigned long long badm = __ballot(pair_bad);
            badm |= badm >> 4;
            badm |= badm >> 2;
            badm |= badm >> 1;
            const unsigned long long stand = okm & ~badm & 0x0101010101010101ull;
            const unsigned long long fell = ~stand & 0x0101010101010101ull;
            int nj = fell ? __builtin_ctzll(fell) >> 3 : K;
            const int rem = M - cnt;
            nj = nj > rem ? rem : nj;
            if (jj == 0 && e < nj) {
                s_acc[e] = make_float4(Be.x, Be.y, Be.z, 0.f);
                if (part == 0) emit(cnt + e, Be.x, Be.y, Be.z, -1 - __float_as_int(Ae.z));
            }
            int mode = 0;
            float vtop = 0.f;
            if (nj == 0) {
                // nothing accepted: either the maximum is not unique at the keys' resolution (exact-tie search for the TRUE
                // maximum) or nothing reached the bounds (lower the threshold and select again)
                if (nvalid > 0) {
                    vtop = wave_max_fused(valid ? val : -1.f);
                    mode = 1;
                } else {
                    mode = 2;
                }
            }
            // the next threshold: the candidate SN2_FC_TAU_KEEP places behind the accepted ones stays above it.  Only buckets
            // that reached THIS threshold are known, so a list that is too short is extended downwards by its own average
            // spacing (any value is exact -- see the header comment --, a good one keeps a dozen survivors per super-round:
            // few enough to rank in a few compares, enough that a super-round never runs out of candidates)
            tau = -1.f;
            if (mode != 2 && nvalid >= 2) {
                const int want = nj + tau_keep;
                if (want <= nvalid - 1) {
                    tau = s_crA[want].x;
                } else {
                    const float v0 = s_crA[0].x, vl = s_crA[nvalid - 1].x;
                    tau = vl - (v0 - vl) / (float)(nvalid - 1) * (float)(want - (nvalid - 1));
                }
            }
            if (failed) mode = 3;
            if (lane == 0) {
                if (failed) atomicAdd(&ctl[1], 1u);
                s_ctl[0] = nj;
                s_ctl[1] = (mode == 3 || cnt + (mode == 0 ? nj : (mode == 1 ? 1 : 0)) >= M) ? 1 : 0;
                s_ctl[2] = __float_as_int(vtop);
                s_ctl[3] = 0;
                s_ctl[4] = mode;
                *s_win = 0xFFFFFFFFu;
            }
#ifdef SN2_FC_STAMPS
            FCSTAMP(t7);
            acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2; acc[3] += t4 - t3; acc[4] += t5 - t4; acc[5] += t6 - t5;
            acc[6] += t7 - t6;
            n_rounds += 1; n_acc += nj; n_m1 += mode == 1; n_m2 += mode == 2;
#endif
        }
        FCSTAMP(t8);
        __syncthreads();
#ifdef SN2_FC_STAMPS
        { FCSTAMP(t9); acc[7] += t9 - t8; }
#endif
        j = s_ctl[0];
        int done = s_ctl[1];
        const int mode = s_ctl[4];
        if (mode == 3) {
            gave_up = true;
            break;
        }
        if (mode == 0) {
            const float4 a = s_acc[lane < j ? lane : 0];
            ax = a.x; ay = a.y; az = a.z;
        } else if (mode == 1) {
            // exact tie of the maximal distance: lowest ORIGINAL index among all points of the plot attaining it
            const float V = __int_as_float(s_ctl[2]);
#pragma unroll
            for (int h = 0; h < SL; ++h) {
                const int sl = 64 * h + lane;
                unsigned long long cand = __ballot(sl < SPW && s_val[sl * NW + wave] == V);
                while (cand) {
                    const int kk = 64 * h + __ffsll((long long)cand) - 1;
                    cand &= cand - 1;
                    const int pp = ((kk * NW + wave) * P + part) * 64 + lane;
                    float d = -1.f;
                    if (pp < N) {
                        if constexpr (LP) d = s_pts[(kk * NW + wave) * 64 + lane].w; else d = pts[pp].w;
                    }
                    unsigned oi = 0xFFFFFFFFu;
                    if (d == V) oi = (unsigned)ord[pp];
                    oi = wave_min_u32_dpp(oi);
                    if (lane == 0) atomicMin(s_win, oi);
                }
            }
            __syncthreads();
            if (wave == 0) {
                fc_u64* tg = xg + 6 * NE + P;           // one tie word per workgroup
                if (lane == 0) fc_store(tg + part, epoch, *s_win);
                fc_u64 tw = 0;
                bool failed = false;
                for (unsigned spins = 0;;) {
                    bool ok = true;
                    if (lane < P) {
                        tw = fc_pick(FC_DUAL ? fc_load(tg + lane) : 0ull, fc_load(tg + FC_COPY_U64 + lane), epoch);
                        ok = (unsigned)(tw >> 32) == epoch;
                    }
                    if (__ballot(!ok) == 0ull) break;
                    if (++spins > spin_limit) {
                        failed = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                const unsigned w = wave_min_u32_dpp(lane < P ? (unsigned)tw : 0xFFFFFFFFu);
                if (lane == 0) {
                    s_ctl[5] = (int)w;
                    if (failed || w == 0xFFFFFFFFu) {
                        atomicAdd(&ctl[1], 1u);
                        s_ctl[4] = 3;
                    } else {
                        s_acc[0] = make_float4(px[w], py[w], pz[w], 0.f);
                    }
                }
            }
            __syncthreads();
            if (s_ctl[4] == 3) {
                gave_up = true;
                break;
            }
            cur = __builtin_amdgcn_readfirstlane(s_ctl[5]);
            ax = px[cur]; ay = py[cur]; az = pz[cur];
            if (tid == 0 && part == 0) emit(cnt, ax, ay, az, cur);
            j = 1;
        } else {
            j = 0;                                      // mode 2: no sample this super-round, the threshold was lowered
        }
        cnt += j;
        if (done) break;
    }
#ifdef SN2_FC_STAMPS
    if (ticket == 0 && tid == 0) {
        for (int i = 0; i < 8; ++i) ctl[2 + i] = (unsigned)(acc[i] >> 4);       // 16-clock units
        ctl[10] = n_rounds; ctl[11] = n_acc; ctl[12] = n_fb; ctl[13] = n_m1 | (n_m2 << 16); ctl[14] = n_surv; ctl[15] = n_spins;
    }
#endif
    // the samples leave: sorted positions -> original indices (kept out of the loop: the loads would sit on wave 0's critical path)
    // (after a give-up the log holds `cnt` samples and uninitialised LDS behind them: nothing is translated, nothing written)
    if (part != 0 || gave_up) return;
    __syncthreads();
    for (int i = tid; i < M; i += NW * 64) {
        if (use_log) {
            const float4 e = s_log[i];
            const int code = __float_as_int(e.w);
            out_idx[i] = code < 0 ? ord[-1 - code] : code;
            cpos_soa[((size_t)b * 3 + 0) * M + i] = e.x;
            cpos_soa[((size_t)b * 3 + 1) * M + i] = e.y;
            cpos_soa[((size_t)b * 3 + 2) * M + i] = e.z;
            reinterpret_cast<float4*>(cpos_aos)[(size_t)b * M + i] = make_float4(e.x, e.y, e.z, 0.f);
        } else {
            const int v = out_idx[i];
            if (v < 0) out_idx[i] = ord[-1 - v];
        }
    }
}

static bool fps_cluster_lds_points = getenv("SN2_FC_NO_LDS_POINTS") == nullptr;     // (diagnostic switches)
static int fps_cluster_tau_keep = getenv("SN2_FC_TAU_KEEP") ? atoi(getenv("SN2_FC_TAU_KEEP")) : SN2_FC_TAU_KEEP;
static unsigned fps_cluster_spin_limit = getenv("SN2_FC_SPIN_LIMIT") ? (unsigned)strtoul(getenv("SN2_FC_SPIN_LIMIT"), nullptr, 0) : FC_SPIN_LIMIT;
// tests only: how many sweeps a wait of fps_cluster_kernel makes before it gives up (0 = back to the default); a tiny limit
// makes every exchange give up, which is how tests/test_gpu_geometry.py exercises the repair launch
extern "C" int sn2_debug_fps_spin_limit(unsigned sweeps) {
    fps_cluster_spin_limit = sweeps ? sweeps : FC_SPIN_LIMIT;
    return 0;
}
template <int SPW, int NW, int P>
static int launch_fps_cluster(const float* pos, int B, int N, int M, const int* start, int* ws, int* idx, float* cs,
                              float* ca, hipStream_t st) {
    constexpr int NBL = SPW * NW, NE = FC_TP * P;
    int* order = ws;
    float4* sorted = reinterpret_cast<float4*>(ws + (size_t)B * N);
    int* grid = ws + (size_t)5 * B * N;
    unsigned* xchg = reinterpret_cast<unsigned*>(grid + (size_t)B * GRID_WORDS);
    unsigned* ctl = xchg + (size_t)B * FPS_XCHG_WORDS;
    launch_spatial_order(pos, B, N, order, sorted, grid, xchg, ctl, st);
    const int log_cap = M <= 4096 ? M : 0;           // the samples of a plot stay in LDS until the end (16 B each) when they fit
    const size_t lds0 = (size_t)NBL * 16 + 8 * 16 + 2 * (size_t)(NE + 16) * 16 + 2 * FC_TP * 16 + (size_t)NBL * 4 * 9 + 72 * 4 + 8 * 4 + 16 +
                        (size_t)log_cap * 16;
    // the workgroup's points in LDS (64 KB at 8 x 8 x 8) when they fit beside the rest: phase B then never waits for L2
    constexpr bool can_lp = (size_t)NBL * 64 * 16 <= 96 * 1024;
    if (can_lp && lds0 + (size_t)NBL * 64 * 16 <= 150 * 1024 && fps_cluster_lds_points) {
        const size_t lds = lds0 + (size_t)NBL * 64 * 16;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fps_cluster_kernel<SPW, NW, P, can_lp>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((fps_cluster_kernel<SPW, NW, P, can_lp>), dim3(B * P), dim3(NW * 64), lds, st, pos, B, N, M, start,
                           (const int*)order, sorted, idx, cs, ca, reinterpret_cast<fc_u64*>(xchg), ctl, log_cap, fps_cluster_tau_keep,
                           fps_cluster_spin_limit);
        SN2_RETURN_LAUNCH();
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fps_cluster_kernel<SPW, NW, P, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds0);
    hipLaunchKernelGGL((fps_cluster_kernel<SPW, NW, P, false>), dim3(B * P), dim3(NW * 64), lds0, st, pos, B, N, M, start,
                       (const int*)order, sorted, idx, cs, ca, reinterpret_cast<fc_u64*>(xchg), ctl, log_cap, fps_cluster_tau_keep,
                       fps_cluster_spin_limit);
    SN2_RETURN_LAUNCH();
}

// P workgroups of NW waves per plot; the slots per wave follow from the plot size
template <int NW, int P>
static int dispatch_fps_cluster(const float* pos, int B, int N, int M, const int* start, int* ws, int* idx, float* cs,
                                float* ca, hipStream_t st) {
    const int nbl = sn2_cdiv(sn2_cdiv(N, 64), P);          // buckets per workgroup
    const int spw = sn2_cdiv(nbl, NW);
    if (spw <= 2) return launch_fps_cluster<2, NW, P>(pos, B, N, M, start, ws, idx, cs, ca, st);
    if (spw <= 4) return launch_fps_cluster<4, NW, P>(pos, B, N, M, start, ws, idx, cs, ca, st);
    if (spw <= 8) return launch_fps_cluster<8, NW, P>(pos, B, N, M, start, ws, idx, cs, ca, st);
    if (spw <= 16) return launch_fps_cluster<16, NW, P>(pos, B, N, M, start, ws, idx, cs, ca, st);
    if (spw <= 32) return launch_fps_cluster<32, NW, P>(pos, B, N, M, start, ws, idx, cs, ca, st);
    if constexpr (NW <= 8) {     // (64 slots x 16 waves = 16 bucket values per lane of the selecting wave: spills at 128 VGPRs)
        if (spw <= 64) return launch_fps_cluster<64, NW, P>(pos, B, N, M, start, ws, idx, cs, ca, st);
    }
    return SN2_ELIMIT;
}

// the single-workgroup kernel for a plot of N points, 16 waves (what `waves = 16` runs); repair = behind fps_cluster_kernel
static int dispatch_fps_bucket16(const float* pos_soa, int B, int N, int M, const int* start, int* order_ws, int* idx,
                                 float* cpos_soa, float* cpos_aos, hipStream_t st, bool spec, bool repair, unsigned* status) {
    if (N <= 4096) return launch_fps_bucket<4>(pos_soa, B, N, M, start, order_ws, idx, cpos_soa, cpos_aos, st, spec, repair, status);
    if (N <= 8192) return launch_fps_bucket<8>(pos_soa, B, N, M, start, order_ws, idx, cpos_soa, cpos_aos, st, spec, repair, status);
    if (N <= 16384) return launch_fps_bucket<16>(pos_soa, B, N, M, start, order_ws, idx, cpos_soa, cpos_aos, st, spec, repair, status);
    if (N <= 32768) return launch_fps_bucket<32>(pos_soa, B, N, M, start, order_ws, idx, cpos_soa, cpos_aos, st, spec, repair, status);
    if (N <= 65536) return launch_fps_bucket<64>(pos_soa, B, N, M, start, order_ws, idx, cpos_soa, cpos_aos, st, spec, repair, status);
    if (N <= 131072) return launch_fps_bucket<128>(pos_soa, B, N, M, start, order_ws, idx, cpos_soa, cpos_aos, st, spec, repair, status);
    return SN2_ELIMIT;
}

extern "C" int sn2_fps_waves(const float* pos_soa, int B, int N, int M, const int* start, int* idx, float* cpos_soa,
                             float* cpos_aos, int* order_ws, int waves, void* stream) {
    return sn2_fps_status(pos_soa, B, N, M, start, idx, cpos_soa, cpos_aos, order_ws, waves, nullptr, stream);
}

extern "C" int sn2_fps_status(const float* pos_soa, int B, int N, int M, const int* start, int* idx, float* cpos_soa,
                              float* cpos_aos, int* order_ws, int waves, unsigned* status, void* stream) {
    if (!pos_soa || !idx || !cpos_soa || !cpos_aos || B <= 0 || N <= 0 || M <= 0 || M > N) return SN2_EINVAL;
    bool cluster = waves == 34 || waves == 36 || waves == 40 || waves == 66 || waves == 68 || waves == 72;
    if (waves != 0 && waves != 16 && waves != 8 && waves != 4 && waves != 1 && !cluster) return SN2_EINVAL;
    if (waves == 0) {
        // the shortest pass: several workgroups per plot wherever that fits (measured at 16 x 32 768 -> 1024: 8 workgroups of 8
        // waves 0.87 ms, 4 of 8 0.99, the single workgroup 1.24; scripts/time_fps_cluster.py)
        const int nbk = sn2_cdiv(N, 64);
        for (int Pc = 8; Pc >= 2 && !cluster; Pc >>= 1) {
            if ((long)B * Pc <= sn2_cu_count() && nbk >= 2 * Pc * 8 && nbk <= 512 * Pc) {
                waves = 64 + Pc;
                cluster = true;
            }
        }
    }
    const bool spec = waves != 1;      // 1: the one-sample-per-round kernel (round 1's; cross-checks and timing comparisons)
    hipStream_t st = (hipStream_t)stream;
    // Many small plots (the parcel loop's level 2: 256 .. 512 plots of 2 500 points -> 625): the bucketed kernel has 40 buckets to
    // prune among and takes 2 us per sample with 16 waves per plot (1.1 ms at half of every CU's wave slots); the brute-force
    // kernel with the plot's points in the registers of FOUR waves takes the whole plot per round and is shorter.  Mirrored by
    // hip_ops.fps_fills_ws (no workspace is filled then: the ball query behind it scans the plot).
    const bool many_small = N <= 4096 && B > 32;
    if (order_ws && N > 2048 && !many_small && M > 16 && (((size_t)B * N) % 4 == 0) && (((size_t)order_ws) % 16 == 0)) {
        // 32 + P / 64 + P: P = 2, 4 or 8 workgroups of 16 / 8 waves per plot (fps_cluster_kernel); needs at least two buckets
        // per wave and all B * P workgroups resident at once, else the single-workgroup kernel below runs
        if (cluster) {
            const int P = waves & 15, NWc = (waves & 64) ? 8 : 16;
            const bool fits = (long)B * P <= sn2_cu_count() && sn2_cdiv(N, 64) >= 2 * P * NWc && sn2_cdiv(N, 64) <= 512 * P;
            if (fits) {
                int rc;
                if (waves == 34) rc = dispatch_fps_cluster<16, 2>(pos_soa, B, N, M, start, order_ws, idx, cpos_soa, cpos_aos, st);
                else if (waves == 36) rc = dispatch_fps_cluster<16, 4>(pos_soa, B, N, M, start, order_ws, idx, cpos_soa, cpos_aos, st);
                else if (waves == 40) rc = dispatch_fps_cluster<16, 8>(pos_soa, B, N, M, start, order_ws, idx, cpos_soa, cpos_aos, st);
                else if (waves == 66) rc = dispatch_fps_cluster<8, 2>(pos_soa, B, N, M, start, order_ws, idx, cpos_soa, cpos_aos, st);
                else if (waves == 68) rc = dispatch_fps_cluster<8, 4>(pos_soa, B, N, M, start, order_ws, idx, cpos_soa, cpos_aos, st);
                else rc = dispatch_fps_cluster<8, 8>(pos_soa, B, N, M, start, order_ws, idx, cpos_soa, cpos_aos, st);
                if (rc != 0) return rc;
                // the repair launch: B workgroups that read control word 1 and leave -- unless a wait of the pass above gave
                // up (its workgroups are not guaranteed to be resident together), in which case they sample every plot again
                return dispatch_fps_bucket16(pos_soa, B, N, M, start, order_ws, idx, cpos_soa, cpos_aos, st, true, true, status);
            }
            waves = 16;
        }
        // bucketed path (exact, see above): up to 128 bucket slots per wave = 131 072 points per plot.
        // waves = 8: half the waves per plot.  Alone the pass is 9 % slower (1.37 vs 1.26 ms at 32 x 32 768: fewer loads in
        // flight), but it leaves half of its CU's wave slots to whatever else runs: beside the feature pass of a pipelined
        // training loop the STEP is 2.4 % shorter (0.918 vs 0.940 ms; 4 waves: 2.07 ms alone, 0.922 ms per step).
        // waves = 4 (plots of at most 16 384 points): half the wave slots again, for passes that share the chip with MANY other
        // workgroups (the parcel loop: 512 plots per launch = two FPS workgroups per CU); larger plots take the 8-wave kernel
        if (waves == 4 && N <= 16384) {
            if (N <= 8192) return launch_fps_bucket<32, 4>(pos_soa, B, N, M, start, order_ws, idx, cpos_soa, cpos_aos, st);
            return launch_fps_bucket<64, 4>(pos_soa, B, N, M, start, order_ws, idx, cpos_soa, cpos_aos, st);
        }
        // (four waves at 32 768 points -- launch_fps_bucket<128, 4> -- were measured in the training loop: 0.773 against 0.769 ms
        // per step with eight; not instantiated)
        if (waves == 4) waves = 8;
        if (waves == 8 && N <= 32768) {
            if (N <= 4096) return launch_fps_bucket<8, 8>(pos_soa, B, N, M, start, order_ws, idx, cpos_soa, cpos_aos, st);
            if (N <= 8192) return launch_fps_bucket<16, 8>(pos_soa, B, N, M, start, order_ws, idx, cpos_soa, cpos_aos, st);
            if (N <= 16384) return launch_fps_bucket<32, 8>(pos_soa, B, N, M, start, order_ws, idx, cpos_soa, cpos_aos, st);
            return launch_fps_bucket<64, 8>(pos_soa, B, N, M, start, order_ws, idx, cpos_soa, cpos_aos, st);
        }
        return dispatch_fps_bucket16(pos_soa, B, N, M, start, order_ws, idx, cpos_soa, cpos_aos, st, spec, false, nullptr);
    }
    if (N <= 256) return launch_fps<1, 256>(pos_soa, B, N, M, start, idx, cpos_soa, cpos_aos, st);
    if (N <= 512) return launch_fps<2, 256>(pos_soa, B, N, M, start, idx, cpos_soa, cpos_aos, st);
    if (N <= 1024) return launch_fps<4, 256>(pos_soa, B, N, M, start, idx, cpos_soa, cpos_aos, st);
    if (N <= 2048) return launch_fps<2, 1024>(pos_soa, B, N, M, start, idx, cpos_soa, cpos_aos, st);
    if (N <= 4096 && many_small) return launch_fps<16, 256>(pos_soa, B, N, M, start, idx, cpos_soa, cpos_aos, st);
    if (N <= 4096) return launch_fps<4, 1024>(pos_soa, B, N, M, start, idx, cpos_soa, cpos_aos, st);
    if (N <= 8192) return launch_fps<8, 1024>(pos_soa, B, N, M, start, idx, cpos_soa, cpos_aos, st);
    if (N <= 16384) return launch_fps<16, 1024>(pos_soa, B, N, M, start, idx, cpos_soa, cpos_aos, st);
    if (N <= 32768) return launch_fps<32, 1024, true>(pos_soa, B, N, M, start, idx, cpos_soa, cpos_aos, st);
    return SN2_ELIMIT;
}

extern "C" int sn2_fps(const float* pos_soa, int B, int N, int M, const int* start, int* idx, float* cpos_soa,
                       float* cpos_aos, int* order_ws, void* stream) {
    return sn2_fps_waves(pos_soa, B, N, M, start, idx, cpos_soa, cpos_aos, order_ws, 0, stream);
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
    (void)sum;
    (void)total;
}

// ------------------------------------------------------------------------------------------------------------
// Grid ball query (exact): the sources were already sorted into 16^3 Morton cells by spatial_order_kernel (FPS level 1
// runs on the same points).  One wave per centroid walks only the cells that can intersect the ball -- the cell range of
// [c - r, c + r] per axis under the same monotone fp32 cell function, so no point with d2 < r2 is missed --, tests the
// cells' points with the canonical sn2_d2, compacts the hits into a wave-private LDS list, rank-sorts them by ORIGINAL
// index (the contract: ascending source index, first `cap` kept) and writes the list.  ~200 candidates per centroid
// instead of 32 768.  A ball with more hits than the LDS list (GQ_LIST) falls back to the full scan for that centroid.
// ------------------------------------------------------------------------------------------------------------
constexpr int GQ_LIST = 512;
#ifndef SN2_GQ_DENSE
#define SN2_GQ_DENSE 512
#endif
constexpr int GQ_DENSE = SN2_GQ_DENSE;   // hits beyond which the bitmap path takes over from the rank sort (measured: switching at 192
                                         // instead of 512 made the query slower at both sizes, 0.30 vs 0.26 ms at 8 x 131 072, 0.075 vs 0.054 at 16 x 32 768)
static_assert(GQ_DENSE <= GQ_LIST, "the list holds the sparse balls");
#ifndef SN2_GQ_BM_CAND
#define SN2_GQ_BM_CAND 160
#endif
constexpr int GQ_BM_WORDS = 512, GQ_BM_CAND = SN2_GQ_BM_CAND;   // bitmap first: plots of <= 16 384 points, more candidates than this

__global__ __launch_bounds__(256) void ball_query_grid_kernel(const float* __restrict__ src, int B, int N,
                                                              const float* __restrict__ cpos, int M, float r, float r2,
                                                              int cap, const int* __restrict__ order,
                                                              const float4* __restrict__ sorted, const int* __restrict__ grid,
                                                              int* __restrict__ nbr, int* __restrict__ cnt,
                                                              unsigned long long* __restrict__ total, int xcd_aware) {
    __shared__ int s_list[4][GQ_LIST];
    extern __shared__ unsigned gq_bits[];              // [4][(N + 31) / 32]: one bit per source point, per wave (dense balls)
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    // XCD-aware placement (round 4): workgroups go to the eight XCDs round-robin, and every XCD has its own L2 -- with the
    // centroids dealt out in index order all eight L2s fetched every plot's sorted table (PMC: 5.2 x the compulsory bytes at
    // 16 x 32 768).  Plot b is worked on by the workgroups of XCD b % 8 only: workgroup w = (XCD w & 7, turn w >> 3), the turns
    // of an XCD walk its plots one after the other, (M + 3) / 4 workgroups of four centroids per plot.
    // (xcd_aware = 0: fewer plots than that balances -- a single plot would run on one XCD --: workgroup w takes plot w / bpp)
    const int bpp = (M + 3) >> 2;
    const int turn = xcd_aware ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int pb = xcd_aware ? (int)(blockIdx.x & 7u) + 8 * (turn / bpp) : turn / bpp;
    const int pm = (turn % bpp) * 4 + wib;
    if (pb >= B || pm >= M) return;
    const int ci = __builtin_amdgcn_readfirstlane(pb * M + pm);
    const int nwords = (N + 31) >> 5;
    unsigned* bits = gq_bits + (size_t)wib * nwords;
    // (Round 4 tried working the centroids in the order of their Morton cells instead of FPS order -- consecutive waves then
    // walk the same cells while those are in L1 / L2 --: no gain at 16 x 32 768 (step 0.7661 against 0.7656 ms) and a small loss
    // in the parcel loop (51.2 against 52.2 k plots/s): the kernel is bound by its instructions per candidate, not by where the
    // candidates come from.)
    const int b = ci / M, m = ci - b * M;
    const float cx = cpos[((size_t)b * 3 + 0) * M + m], cy = cpos[((size_t)b * 3 + 1) * M + m],
                cz = cpos[((size_t)b * 3 + 2) * M + m];
    const int* gb = grid + (size_t)b * GRID_WORDS;
    const float* gf = reinterpret_cast<const float*>(gb + ORDER_CELLS + 1);
    const int* ord = order + (size_t)b * N;
    const float4* pts = sorted + (size_t)b * N;
    int* list = s_list[wib];
    // conservative cell range: slightly enlarged radius, same cell function as the sort (monotone in the coordinate)
    const float rr = r * 1.0001f + 1e-6f;
    int lo3[3], hi3[3];
    {
        const float cc[3] = {cx, cy, cz};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int G = a < 2 ? ORDER_GX : ORDER_GZ;
            int l = (int)(((cc[a] - rr) - gf[a]) * gf[3 + a]), h = (int)(((cc[a] + rr) - gf[a]) * gf[3 + a]);
            // (int) truncates toward zero: a negative argument must still map to cell 0 -> clamp covers it
            lo3[a] = l < 0 ? 0 : (l > G - 1 ? G - 1 : l);
            hi3[a] = h < 0 ? 0 : (h > G - 1 ? G - 1 : h);
            if ((cc[a] - rr) - gf[a] < 0.f) lo3[a] = 0;
        }
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    int h = 0;
    // one lane per candidate cell: all cell bounds arrive with ONE round trip, then the candidates of all cells form one
    // flat sequence that the wave walks 64 at a time (a cell-by-cell walk serialised ~50 dependent loads per centroid)
    const int ncx = hi3[0] - lo3[0] + 1, ncy = hi3[1] - lo3[1] + 1, ncz = hi3[2] - lo3[2] + 1;
    const int ncell = ncx * ncy * ncz;
    bool overflow = ncell > 64;
    bool dense = false;                                // more hits than the list holds: the bitmap path below
    if (!overflow) {
        __shared__ int s_pre[4][65], s_p0[4][64];
        int p0 = 0, len = 0;
        if (lane < ncell) {
            const int ix = lo3[0] + lane % ncx, iy = lo3[1] + (lane / ncx) % ncy, iz = lo3[2] + lane / (ncx * ncy);
            const unsigned cell = morton_cell((unsigned)ix, (unsigned)iy, (unsigned)iz);
            p0 = gb[cell];
            len = gb[cell + 1] - p0;
        }
        int incl = len;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        const int T = __shfl(incl, 63);
        s_pre[wib][lane + 1] = incl;   // exclusive prefix of cell i = s_pre[i]
        if (lane == 0) s_pre[wib][0] = 0;
        s_p0[wib][lane] = p0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // four blocks of 64 candidates per turn, their positions AND original indices all requested before the first test:
        // one round trip per 256 candidates (block by block, with the index fetched for the hits only, a centroid with 250
        // candidates waited for eight dependent loads: 54 us per launch at 16 x 32 768)
#ifndef SN2_GQ_INFLIGHT
#define SN2_GQ_INFLIGHT 4
#endif
        constexpr int GQ_INFLIGHT = SN2_GQ_INFLIGHT;
        // (a ball with few candidates -- the parcel loop's 10 000-point plots put ~70 into the 27 cells -- takes the same turn
        // with one or two blocks: the cell search and the tests of empty blocks were a fifth of the kernel's instructions)
        auto turn = [&](auto nb, int t0) {
            constexpr int NB = decltype(nb)::value;
            float4 qv[NB];
            int oi[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                const int t = t0 + 64 * c + lane;
                // the cell of candidate t: largest i with s_pre[i] <= t (binary search over <= 64 entries)
                int lo_i = 0;
#pragma unroll
                for (int step = 32; step > 0; step >>= 1) {
                    const int mid = lo_i + step;
                    if (mid < ncell && s_pre[wib][mid] <= t) lo_i = mid;
                }
                const int p = t < T ? s_p0[wib][lo_i] + (t - s_pre[wib][lo_i]) : 0;
                qv[c] = pts[p];
                oi[c] = ord[p];
            }
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                const bool in = t0 + 64 * c + lane < T;
                const bool hit = in && (sn2_d2(qv[c].x, qv[c].y, qv[c].z, cx, cy, cz) < r2);
                const unsigned long long mask = __ballot(hit);
                if (mask && !dense) {
                    const int nh = __popcll(mask);
                    if (h + nh > GQ_DENSE) {
                        dense = true;
                    } else {
                        if (hit) list[h + __popcll(mask & below)] = oi[c];
                        h += nh;
                    }
                }
            }
        };
        // Small plots with many candidates (the parcel loop: 10 000 points, r = sqrt 2: ~500 candidates and ~150 hits per centroid)
        // go to the bitmap at once: it is N / 32 words (5 trips of the wave to clear, 5 to read) whatever the count, while
        // rank-sorting h hits is h^2 / 64 steps -- 1.3 ms of the 5.5 ms a launch of 256 plots takes went into those sorts --, and
        // deciding it before the first walk saves the second one.  Same lists: ascending original index either way.
        if (nwords <= GQ_BM_WORDS && T > GQ_BM_CAND) dense = true;
        else if (T <= 64) turn(std::integral_constant<int, 1>{}, 0);
        else if (T <= 128) turn(std::integral_constant<int, 2>{}, 0);
        else
            for (int t0 = 0; t0 < T && !dense; t0 += 64 * GQ_INFLIGHT) turn(std::integral_constant<int, GQ_INFLIGHT>{}, t0);
        if (dense) {
            // A dense ball (the ground layer of a 131 072-point plot puts ~700 points into a 1 m ball): rank-sorting h hits
            // costs h^2 / 64 steps and the old fallback re-scanned the whole plot (2048 steps per centroid: 0.78 ms at
            // BASELINE config 5).  Instead: one bit per source point in LDS, the candidates walked once more to set the
            // hits' bits, then the bitmap read in order -- ascending original index by construction, no sort, any count.
            for (int i = lane; i < nwords; i += 64) bits[i] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int t0 = 0; t0 < T; t0 += 64 * GQ_INFLIGHT) {
                float4 qv[GQ_INFLIGHT];
                int oi[GQ_INFLIGHT];
#pragma unroll
                for (int c = 0; c < GQ_INFLIGHT; ++c) {
                    const int t = t0 + 64 * c + lane;
                    int lo_i = 0;
#pragma unroll
                    for (int step = 32; step > 0; step >>= 1) {
                        const int mid = lo_i + step;
                        if (mid < ncell && s_pre[wib][mid] <= t) lo_i = mid;
                    }
                    const int p = t < T ? s_p0[wib][lo_i] + (t - s_pre[wib][lo_i]) : 0;
                    qv[c] = pts[p];
                    oi[c] = ord[p];
                }
#pragma unroll
                for (int c = 0; c < GQ_INFLIGHT; ++c)
                    if (t0 + 64 * c + lane < T && (sn2_d2(qv[c].x, qv[c].y, qv[c].z, cx, cy, cz) < r2))
                        atomicOr(&bits[oi[c] >> 5], 1u << (oi[c] & 31));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            int* outd = nbr + (size_t)ci * cap;
            int n = 0;
            for (int w0 = 0; w0 < nwords; w0 += 64) {
                unsigned word = (w0 + lane < nwords) ? bits[w0 + lane] : 0u;
                const int c = __popc(word);
                if (__ballot(c != 0) == 0ull) continue;
                int incl = c;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int tt = __shfl_up(incl, o);
                    if (lane >= o) incl += tt;
                }
                int pos = n + incl - c;
                while (word) {
                    const int bit = __ffs(word) - 1;
                    word &= word - 1;
                    if (pos < cap) outd[pos] = (w0 + lane) * 32 + bit;
                    ++pos;
                }
                n += __shfl(incl, 63);
            }
            if (lane == 0) cnt[ci] = n < cap ? n : cap;
            return;
        }
    }
    int* out = nbr + (size_t)ci * cap;
    if (!overflow) {
        // rank sort by original index (indices are distinct), keep the first `cap`
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int i0 = 0; i0 < h; i0 += 64) {
            const int i = i0 + lane;
            const int mine = i < h ? list[i] : 0x7FFFFFFF;
            int rank = 0;
            for (int j = 0; j < h; ++j) rank += list[j] < mine ? 1 : 0;
            if (i < h && rank < cap) out[rank] = mine;
        }
        const int c = h < cap ? h : cap;
        if (lane == 0) cnt[ci] = c;
    } else {
        // dense ball: full scan of the plot in index order (identical to ball_query_kernel for one centroid)
        const float* px = src + (size_t)b * 3 * N;
        const float* py = px + N;
        const float* pz = py + N;
        int n = 0;
        for (int j0 = 0; j0 < N; j0 += 64) {
            const int j = j0 + lane;
            const bool valid = j < N;
            const float x = valid ? px[j] : 0.f, y = valid ? py[j] : 0.f, z = valid ? pz[j] : 0.f;
            const bool hit = valid && (sn2_d2(x, y, z, cx, cy, cz) < r2);
            const unsigned long long mask = __ballot(hit);
            if (mask) {
                const int p = n + __popcll(mask & below);
                if (hit && p < cap) out[p] = j;
                n += __popcll(mask);
            }
        }
        const int c = n < cap ? n : cap;
        if (lane == 0) cnt[ci] = c;
    }
}

// *total = sum of cnt (one workgroup; thousands of same-address atomics from the query waves cost 0.2 ms)
// (grouped: workgroup h sums the n counts of batch h into total[h * tstride] -- sn2_count_sum_group)
__global__ __launch_bounds__(1024) void count_sum_kernel(const int* __restrict__ cnt_all, int n, unsigned long long* __restrict__ total_all,
                                                         size_t tstride = 0) {
    __shared__ unsigned long long s_tot;
    const int* __restrict__ cnt = cnt_all + (size_t)blockIdx.x * n;
    unsigned long long* __restrict__ total = total_all + (size_t)blockIdx.x * tstride;
    unsigned long long acc = 0;
    // sixteen counts per lane in flight (one dependent load per trip: 0.19 ms for the parcel loop's 640 000 counts)
    const int n16 = (reinterpret_cast<uintptr_t>(cnt) & 15) == 0 ? n / 16 : 0;
    const int4* c4 = reinterpret_cast<const int4*>(cnt);
    for (int i = threadIdx.x; i < n16; i += 1024) {
        int4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = c4[(size_t)i * 4 + u];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += (unsigned long long)((long)v[u].x + v[u].y + v[u].z + v[u].w);
    }
    for (int i = n16 * 16 + threadIdx.x; i < n; i += 1024) acc += (unsigned long long)cnt[i];
    if (threadIdx.x == 0) s_tot = 0ull;
    __syncthreads();
    atomicAdd(&s_tot, acc);
    __syncthreads();
    if (threadIdx.x == 0) *total = s_tot;
}

extern "C" int sn2_ball_query(const float* src_soa, int B, int N, const float* cpos_soa, int M, float r2, int cap,
                              int* nbr, int* cnt, unsigned long long* total, const int* fps_ws, void* stream) {
    if (!src_soa || !cpos_soa || !nbr || !cnt || B <= 0 || N <= 0 || M <= 0 || cap <= 0 || !(r2 > 0.f)) return SN2_EINVAL;
    const float r = sqrtf(r2);   // only used (enlarged) to bound the cell range; the hit test is d2 < r2
    if (fps_ws && N > 2048 && (((size_t)B * N) % 4 == 0)) {
        // the sources are the point set FPS just sorted: walk its cell lists instead of the whole plot
        const int* order = fps_ws;
        const float4* sorted = reinterpret_cast<const float4*>(fps_ws + (size_t)B * N);
        const int* grid = fps_ws + (size_t)5 * B * N;
        const size_t lds = (size_t)4 * ((N + 31) / 32) * sizeof(unsigned);       // the dense-ball bitmaps: 64 KB at N = 131 072
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ball_query_grid_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        // grid: eight XCDs x the plots of the fullest XCD x the workgroups of a plot (ball_query_grid_kernel: placement)
        // (plots per XCD must balance: a multiple of 8 plots, or so many that the remainder does not matter)
        static const bool gq_no_xcd = getenv("SN2_GQ_NO_XCD") != nullptr;        // (diagnostic switch)
        const int xcd_aware = (!gq_no_xcd && (B % 8 == 0 || B >= 64)) ? 1 : 0;
        const long gq_blocks = xcd_aware ? 8L * sn2_cdiv(B, 8) * sn2_cdiv(M, 4) : (long)B * sn2_cdiv(M, 4);
        if (gq_blocks >= (1L << 31)) return SN2_ELIMIT;
        hipLaunchKernelGGL(ball_query_grid_kernel, dim3((unsigned)gq_blocks), dim3(256), lds, (hipStream_t)stream, src_soa,
                           B, N, cpos_soa, M, r, r2, cap, order, sorted, grid, nbr, cnt, total, xcd_aware);
        if (total) hipLaunchKernelGGL(count_sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const int*)cnt, B * M, total);
        SN2_RETURN_LAUNCH();
    }
    constexpr int TC = 8;
    const int tiles = sn2_cdiv(M, TC);
    const long waves = (long)B * tiles;
    hipLaunchKernelGGL((ball_query_kernel<TC>), dim3(sn2_cdiv(waves, 4)), dim3(256), 0, (hipStream_t)stream, src_soa, B,
                       N, cpos_soa, M, r2, cap, nbr, cnt, total, tiles);
    if (total) hipLaunchKernelGGL(count_sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const int*)cnt, B * M, total);
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

// ------------------------------------------------------------------------------------------------------------
// Grid 3-NN (exact), for targets that FPS already put into Morton order (FP1: the targets are the plot's points).
// The plot's S sources are binned once into a G x G grid over their x,y bounding box (nn_grid_build_kernel: LDS counting
// sort -> a table of (x, y, z, source index) in cell order + the cell starts).  One wave takes 64 consecutive SORTED
// targets -- neighbours in space -- and treats them as one query box: it visits the cells under the box, then square
// rings of growing radius rho around it, every lane testing every visited source (wave-uniform loops, LDS broadcast
// reads; a per-lane walk of each lane's own cells was tried first and lost to the brute-force scan: a wave runs as long
// as its worst lane at every cell).  It stops when the largest k-th best squared distance of its lanes is below what any
// unvisited source can reach: such a source lies in a cell more than rho columns or rows away from the cells of ALL the
// wave's targets, so its x or y gap -- hence its distance -- to each of them exceeds rho cell widths (taken 0.1 % short
// and minus 1e-4 m, which covers the fp32 rounding of the cell assignment and of sn2_d2).  Candidates arrive in arbitrary
// index order, so the running best three are kept in (d2, index) lexicographic order: exactly the brute-force scan's
// "ascending index, strict <" result, bit for bit.  ~50-150 distance evaluations per target instead of S = 1024.
// ------------------------------------------------------------------------------------------------------------
constexpr int NN_GMAX = 32;
constexpr int NN_HDR = NN_GMAX * NN_GMAX + 1 + 7;   // cell starts (G*G+1), then x0, y0, inv_x, inv_y, min cell width, G (as int), pad

__global__ __launch_bounds__(256) void nn_grid_build_kernel(const float* __restrict__ src, int S, int G,
                                                            float4* __restrict__ tbl, int* __restrict__ hdr) {
    __shared__ int s_hist[NN_GMAX * NN_GMAX + 1];
    __shared__ float s_mm[4][4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* sx = src + (size_t)b * 3 * S;
    const float* sy = sx + S;
    const float* sz = sy + S;
    float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    for (int i = tid; i < S; i += 256) {
        mnx = fminf(mnx, sx[i]); mxx = fmaxf(mxx, sx[i]);
        mny = fminf(mny, sy[i]); mxy = fmaxf(mxy, sy[i]);
    }
    mnx = wave_min(mnx); mny = wave_min(mny); mxx = wave_max(mxx); mxy = wave_max(mxy);
    if (lane == 0) { s_mm[0][wave] = mnx; s_mm[1][wave] = mny; s_mm[2][wave] = mxx; s_mm[3][wave] = mxy; }
    for (int i = tid; i <= G * G; i += 256) s_hist[i] = 0;
    __syncthreads();
    const float x0 = fminf(fminf(s_mm[0][0], s_mm[0][1]), fminf(s_mm[0][2], s_mm[0][3]));
    const float y0 = fminf(fminf(s_mm[1][0], s_mm[1][1]), fminf(s_mm[1][2], s_mm[1][3]));
    const float x1 = fmaxf(fmaxf(s_mm[2][0], s_mm[2][1]), fmaxf(s_mm[2][2], s_mm[2][3]));
    const float y1 = fmaxf(fmaxf(s_mm[3][0], s_mm[3][1]), fmaxf(s_mm[3][2], s_mm[3][3]));
    const float ex = fmaxf(x1 - x0, 1e-6f), ey = fmaxf(y1 - y0, 1e-6f);
    const float ix = (float)G / ex, iy = (float)G / ey;
    auto cell_of = [&](int i) {
        int cx = (int)((sx[i] - x0) * ix), cy = (int)((sy[i] - y0) * iy);
        cx = cx < 0 ? 0 : (cx > G - 1 ? G - 1 : cx);
        cy = cy < 0 ? 0 : (cy > G - 1 ? G - 1 : cy);
        return cy * G + cx;
    };
    for (int i = tid; i < S; i += 256) atomicAdd(&s_hist[cell_of(i)], 1);
    __syncthreads();
    {   // exclusive scan over the <= 1024 cells: 4 consecutive cells per thread, wave scan, 4 wave totals
        __shared__ int s_wsum[4];
        int loc[4], sum = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            loc[c] = (tid * 4 + c < G * G) ? s_hist[tid * 4 + c] : 0;
            sum += loc[c];
        }
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        int run = incl - sum;
        for (int k = 0; k < wave; ++k) run += s_wsum[k];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (tid * 4 + c < G * G) s_hist[tid * 4 + c] = run;
            run += loc[c];
        }
        if (tid == 255) s_hist[G * G] = run;     // = S: thread 255 ends the scan whatever G is (empty tail cells add 0)
    }
    __syncthreads();
    int* hb = hdr + (size_t)b * NN_HDR;
    for (int i = tid; i <= G * G; i += 256) hb[i] = s_hist[i];
    if (tid == 0) {
        float* hf = reinterpret_cast<float*>(hb + NN_GMAX * NN_GMAX + 1);
        hf[0] = x0; hf[1] = y0; hf[2] = ix; hf[3] = iy; hf[4] = fminf(ex, ey) / (float)G;
        hb[NN_GMAX * NN_GMAX + 1 + 5] = G;
    }
    __syncthreads();
    float4* tb = tbl + (size_t)b * S;
    for (int i = tid; i < S; i += 256) {
        const int p = atomicAdd(&s_hist[cell_of(i)], 1);
        tb[p] = make_float4(sx[i], sy[i], sz[i], __int_as_float(i));
    }
}

#ifdef SN2_NN_STAMPS
// diagnostic build only (never shipped): per wave {cycles, candidates tested, final ring, query box cells}
__device__ unsigned long long g_nn_dbg[4 * 16384];
extern "C" int sn2_debug_nn_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nn_dbg), sizeof(unsigned long long) * n);
}
#endif
__global__ __launch_bounds__(256) void three_nn_grid_kernel(const float4* __restrict__ tbl, const int* __restrict__ hdr, int S,
                                                            int T, int k, const int* __restrict__ dst_order,
                                                            const float4* __restrict__ dst_sorted,
                                                            int* __restrict__ idx, float* __restrict__ w, int B, int gx) {
    extern __shared__ __attribute__((aligned(16))) float4 s_tbl[];      // S entries, then the cell starts
    // XCD-aware placement (as ball_query_grid_kernel): with B = 0 the grid is (gx, plots) as before; with B > 0 it is one-
    // dimensional, workgroup w = (XCD w & 7, turn w >> 3), and plot b is worked on by XCD b % 8 only -- the results go to the
    // targets' ORIGINAL rows, 24 bytes each into lines that ten waves share, and only waves behind the same L2 merge them
    // there before they are written back
    int b = blockIdx.y, bx = blockIdx.x;
    if (B > 0) {
        const int turn = (int)(blockIdx.x >> 3);
        b = (int)(blockIdx.x & 7u) + 8 * (turn / gx);
        bx = turn % gx;
        if (b >= B) return;
    }
    const int lane = threadIdx.x & 63;
    const int* hb = hdr + (size_t)b * NN_HDR;
    const int G = __builtin_amdgcn_readfirstlane(hb[NN_GMAX * NN_GMAX + 1 + 5]);
    int* s_cell = reinterpret_cast<int*>(s_tbl + S);
    for (int i = threadIdx.x; i < S; i += 256) s_tbl[i] = tbl[(size_t)b * S + i];
    for (int i = threadIdx.x; i <= G * G; i += 256) s_cell[i] = hb[i];
    __syncthreads();
    const int p = bx * 256 + threadIdx.x;                                // sorted position
    if (p - lane >= T) return;                                           // whole wave past the end
    const bool valid = p < T;
    const float* hf = reinterpret_cast<const float*>(hb + NN_GMAX * NN_GMAX + 1);
    const float x0 = hf[0], y0 = hf[1], ix = hf[2], iy = hf[3], cw = hf[4];
    const float4 q = dst_sorted[(size_t)b * T + (valid ? p : T - 1)];
    const float qx = q.x, qy = q.y, qz = q.z;
    // the cells under the wave's targets (clamped like the sources' cells; the cell function is monotone)
    auto cell1 = [&](float v, float o, float inv) {
        int c = (int)((v - o) * inv);
        return (v - o < 0.f || c < 0) ? 0 : (c > G - 1 ? G - 1 : c);
    };
    const int bx0 = __builtin_amdgcn_readfirstlane(cell1(wave_min(qx), x0, ix)),
              bx1 = __builtin_amdgcn_readfirstlane(cell1(wave_max(qx), x0, ix));
    const int by0 = __builtin_amdgcn_readfirstlane(cell1(wave_min(qy), y0, iy)),
              by1 = __builtin_amdgcn_readfirstlane(cell1(wave_max(qy), y0, iy));
    constexpr unsigned long long KINF = (0x7F800000ull << 32) | 0x7FFFFFFFull;     // (+inf, no index)
    unsigned long long k0 = KINF, k1 = KINF, k2 = KINF;
    const int kk = k < S ? k : S;          // slots that will be filled
#ifdef SN2_NN_STAMPS
    unsigned long long t_begin;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_begin)::"memory");
    unsigned n_cand = 0;
    int rho_last = 0;
#endif
    for (int rho = 0;; ++rho) {
#ifdef SN2_NN_STAMPS
        rho_last = rho;
#endif
        const int xl = bx0 - rho, xh = bx1 + rho, yl = by0 - rho, yh = by1 + rho;
        const int cxl = xl < 0 ? 0 : xl, cxh = xh > G - 1 ? G - 1 : xh;
        for (int cy = (yl < 0 ? 0 : yl); cy <= (yh > G - 1 ? G - 1 : yh); ++cy) {
            // a new row of the rectangle is one segment of contiguous cells; an old row contributes its two new end cells
            const bool whole = rho == 0 || cy == yl || cy == yh;
            for (int sg = 0; sg < (whole ? 1 : 2); ++sg) {
                int c_lo, c_hi;
                if (whole) {
                    c_lo = cy * G + cxl; c_hi = cy * G + cxh;
                } else if (sg == 0) {
                    if (xl < 0) continue;
                    c_lo = c_hi = cy * G + xl;
                } else {
                    if (xh > G - 1) continue;
                    c_lo = c_hi = cy * G + xh;
                }
                // wave-uniform bounds in SGPRs: scalar loop control and one broadcast LDS read per candidate; the common
                // case (the candidate beats nobody's third best) is 8 VALU ops, one compare and one branch.  (Written
                // inline, not as a lambda: called through a closure the best-three state ended up in scratch memory.)
                const int p_lo = __builtin_amdgcn_readfirstlane(s_cell[c_lo]), p_hi = __builtin_amdgcn_readfirstlane(s_cell[c_hi + 1]);
#ifdef SN2_NN_STAMPS
                n_cand += p_hi - p_lo;
#endif
// the running best three as 64-bit keys (d2 bits << 32 | source index): d2 >= 0, so the unsigned order of the keys IS the
// lexicographic (d2, index) order, and an insertion is three branch-free min/max steps.  (With 64 lanes per query box
// nearly every candidate is a hit for SOME lane, so the nested-if insertion ran, divergent, for most candidates.)
#define NN_INSERT(DD, CW)                                                                                  \
    {                                                                                                      \
        const unsigned long long kn_ = ((unsigned long long)__float_as_uint(DD) << 32) | (unsigned)__float_as_int(CW); \
        k2 = kn_ < k2 ? kn_ : k2;                                                                          \
        const unsigned long long a_ = k1 < k2 ? k1 : k2, b_ = k1 < k2 ? k2 : k1;                           \
        k1 = a_; k2 = b_;                                                                                  \
        const unsigned long long c_ = k0 < k1 ? k0 : k1, e_ = k0 < k1 ? k1 : k0;                           \
        k0 = c_; k1 = e_;                                                                                  \
    }
                int sp = p_lo;
                for (; sp + 4 <= p_hi; sp += 4) {
                    const float4 c0 = s_tbl[sp], c1 = s_tbl[sp + 1], c2 = s_tbl[sp + 2], c3 = s_tbl[sp + 3];
                    const float e0 = sn2_d2(c0.x, c0.y, c0.z, qx, qy, qz), e1 = sn2_d2(c1.x, c1.y, c1.z, qx, qy, qz);
                    const float e2 = sn2_d2(c2.x, c2.y, c2.z, qx, qy, qz), e3 = sn2_d2(c3.x, c3.y, c3.z, qx, qy, qz);
                    if (fminf(fminf(e0, e1), fminf(e2, e3)) <= __uint_as_float((unsigned)(k2 >> 32))) {
                        NN_INSERT(e0, c0.w)
                        NN_INSERT(e1, c1.w)
                        NN_INSERT(e2, c2.w)
                        NN_INSERT(e3, c3.w)
                    }
                }
                for (; sp < p_hi; ++sp) {
                    const float4 c = s_tbl[sp];
                    const float dd = sn2_d2(c.x, c.y, c.z, qx, qy, qz);
                    NN_INSERT(dd, c.w)
                }
#undef NN_INSERT
            }
        }
        if (xl <= 0 && yl <= 0 && xh >= G - 1 && yh >= G - 1) break;     // the whole grid has been visited
        const float reach = (float)rho * cw * 0.999f - 1e-4f;           // nothing unvisited is closer than this
        const float dk = __uint_as_float((unsigned)((kk >= 3 ? k2 : (kk == 2 ? k1 : k0)) >> 32));
        const float worst = wave_max(valid ? dk : 0.f);
        if (reach > 0.f && worst < reach * reach) break;
    }
    const float d0 = __uint_as_float((unsigned)(k0 >> 32)), d1 = __uint_as_float((unsigned)(k1 >> 32)),
                d2 = __uint_as_float((unsigned)(k2 >> 32));
    const int i0 = (int)(unsigned)k0, i1 = (int)(unsigned)k1, i2 = (int)(unsigned)k2;
#ifdef SN2_NN_STAMPS
    {
        unsigned long long t_end;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_end)::"memory");
        const int wid = (b * gx + bx) * 4 + (threadIdx.x >> 6);
        if (lane == 0 && wid < 16384) {
            g_nn_dbg[4 * wid + 0] = t_end - t_begin;
            g_nn_dbg[4 * wid + 1] = n_cand;
            g_nn_dbg[4 * wid + 2] = rho_last;
            g_nn_dbg[4 * wid + 3] = (unsigned long long)((bx1 - bx0 + 1) * (by1 - by0 + 1));
        }
    }
#endif
    if (!valid) return;
    const int t = dst_order[(size_t)b * T + p];
    const size_t o = ((size_t)b * T + t) * 3;
    const bool u1 = (k >= 2) && (i1 != 0x7FFFFFFF), u2 = (k >= 3) && (i2 != 0x7FFFFFFF);
    idx[o + 0] = i0;
    idx[o + 1] = u1 ? i1 : i0;
    idx[o + 2] = u2 ? i2 : i0;
    w[o + 0] = 1.0f / fmaxf(d0, 1e-16f);
    w[o + 1] = u1 ? 1.0f / fmaxf(d1, 1e-16f) : 0.f;
    w[o + 2] = u2 ? 1.0f / fmaxf(d2, 1e-16f) : 0.f;
}

extern "C" int sn2_count_sum_group(const int* cnt, int G, int n, unsigned long long* total, size_t total_stride, void* stream) {
    if (!cnt || !total || G <= 0 || n <= 0 || (G > 1 && total_stride < 1)) return SN2_EINVAL;
    hipLaunchKernelGGL(count_sum_kernel, dim3(G), dim3(1024), 0, (hipStream_t)stream, cnt, n, total, total_stride);
    SN2_RETURN_LAUNCH();
}

extern "C" int sn2_count_sum(const int* cnt, int n, unsigned long long* total, void* stream) {
    if (!cnt || !total || n <= 0) return SN2_EINVAL;
    hipLaunchKernelGGL(count_sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, cnt, n, total);
    SN2_RETURN_LAUNCH();
}

// The targets of a plot in the order of the SOURCE grid's cells (rows of cells walked in a snake, so that consecutive cells
// are always neighbours): counting sort in LDS, one workgroup per plot.  A wave of three_nn_grid_kernel then holds 64
// targets of one or two adjacent cells -- a 1-2 cell query box for every wave.  (In the order sn2_fps leaves, a 3-D
// Morton order of a 16^3 grid, the boxes averaged 7 cells and reached 72: 162 candidates per wave on average, 726 in
// the slowest, and the slowest wave is the kernel.)  The order inside a cell is whatever the cursor atomics give: the
// search is exact, so its results do not depend on it.
__global__ __launch_bounds__(1024) void nn_target_sort_kernel(const float* __restrict__ dst, int T, const int* __restrict__ hdr,
                                                              int* __restrict__ order, float4* __restrict__ sorted) {
    __shared__ int s_hist[NN_GMAX * NN_GMAX + 1];
    __shared__ int s_wsum[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int* hb = hdr + (size_t)b * NN_HDR;
    const int G = hb[NN_GMAX * NN_GMAX + 1 + 5];
    const float* hf = reinterpret_cast<const float*>(hb + NN_GMAX * NN_GMAX + 1);
    const float x0 = hf[0], y0 = hf[1], ix = hf[2], iy = hf[3];
    const float* dx = dst + (size_t)b * 3 * T;
    const float* dy = dx + T;
    const float* dz = dy + T;
    auto key_of = [&](int i) {
        const float vx = dx[i] - x0, vy = dy[i] - y0;
        int cx = (int)(vx * ix), cy = (int)(vy * iy);
        cx = (vx < 0.f || cx < 0) ? 0 : (cx > G - 1 ? G - 1 : cx);
        cy = (vy < 0.f || cy < 0) ? 0 : (cy > G - 1 ? G - 1 : cy);
        return cy * G + ((cy & 1) ? G - 1 - cx : cx);
    };
    for (int i = tid; i <= G * G; i += 1024) s_hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < T; i += 1024) atomicAdd(&s_hist[key_of(i)], 1);
    __syncthreads();
    {   // exclusive scan over the <= 1024 cells: one cell per thread, wave scan, 16 wave totals
        const int v = tid < G * G ? s_hist[tid] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        int run = incl - v;
        for (int k = 0; k < wave; ++k) run += s_wsum[k];
        __syncthreads();
        if (tid < G * G) s_hist[tid] = run;
    }
    __syncthreads();
    int* ob = order + (size_t)b * T;
    float4* sb = sorted + (size_t)b * T;
    for (int i = tid; i < T; i += 1024) {
        const int p = atomicAdd(&s_hist[key_of(i)], 1);
        ob[p] = i;
        sb[p] = make_float4(dx[i], dy[i], dz[i], 0.f);
    }
}

// nn_target_sort_kernel with U loads per coordinate in flight (the loop form spends a trip of dependent loads per point and
// pass: 54 us at T = 32 768); same keys, same tables
template <int U, int NT>
__global__ __launch_bounds__(NT) void nn_target_sort_chunk_kernel(const float* __restrict__ dst, int T, const int* __restrict__ hdr,
                                                                    int* __restrict__ order, float4* __restrict__ sorted) {
    __shared__ int s_hist[NN_GMAX * NN_GMAX + 1];
    __shared__ int s_wsum[NT / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* dx = dst + (size_t)b * 3 * T;
    const float* dy = dx + T;
    const float* dz = dy + T;
    const int* hb = hdr + (size_t)b * NN_HDR;
    const int G = hb[NN_GMAX * NN_GMAX + 1 + 5];
    const float* hf = reinterpret_cast<const float*>(hb + NN_GMAX * NN_GMAX + 1);
    const float x0 = hf[0], y0 = hf[1], ix = hf[2], iy = hf[3];
    auto key_of = [&](float px_, float py_) {                       // nn_target_sort_kernel's key, bit for bit
        const float vx = px_ - x0, vy = py_ - y0;
        int cx = (int)(vx * ix), cy = (int)(vy * iy);
        cx = (vx < 0.f || cx < 0) ? 0 : (cx > G - 1 ? G - 1 : cx);
        cy = (vy < 0.f || cy < 0) ? 0 : (cy > G - 1 ? G - 1 : cy);
        return cy * G + ((cy & 1) ? G - 1 - cx : cx);
    };
    for (int i = tid; i <= G * G; i += NT) s_hist[i] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < T; i0 += NT * U) {
        float vx[U], vy[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * NT + tid, ii = i < T ? i : T - 1;
            vx[u] = dx[ii]; vy[u] = dy[ii];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i0 + u * NT + tid < T) atomicAdd(&s_hist[key_of(vx[u], vy[u])], 1);
    }
    __syncthreads();
    {   // exclusive scan over the <= 1024 cells: CPT consecutive cells per thread, wave scan, NT / 64 wave totals
        constexpr int CPT = (NN_GMAX * NN_GMAX + NT - 1) / NT;
        int v[CPT], sum = 0;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            v[c] = tid * CPT + c < G * G ? s_hist[tid * CPT + c] : 0;
            sum += v[c];
        }
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        int run = incl - sum;
        for (int k = 0; k < wave; ++k) run += s_wsum[k];
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            if (tid * CPT + c < G * G) s_hist[tid * CPT + c] = run;
            run += v[c];
        }
    }
    __syncthreads();
    int* ob = order + (size_t)b * T;
    float4* sb = sorted + (size_t)b * T;
    for (int i0 = 0; i0 < T; i0 += NT * U) {
        float vx[U], vy[U], vz[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * NT + tid, ii = i < T ? i : T - 1;
            vx[u] = dx[ii]; vy[u] = dy[ii]; vz[u] = dz[ii];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * NT + tid;
            if (i < T) {
                const int p = atomicAdd(&s_hist[key_of(vx[u], vy[u])], 1);
                ob[p] = i;
                sb[p] = make_float4(vx[u], vy[u], vz[u], 0.f);
            }
        }
    }
}

extern "C" int sn2_three_nn_xy(const float* src_soa, int B, int S, const float* dst_soa, int T, int k, int* idx, float* w,
                               void* ws, void* stream) {
    if (!src_soa || !dst_soa || !idx || !w || !ws || B <= 0 || S <= 0 || T <= 0 || k < 1 || k > 3) return SN2_EINVAL;
    if (S < 128 || S > 8192 || (((size_t)ws) % 16) != 0) return SN2_ELIMIT;
    hipStream_t st = (hipStream_t)stream;
    int G = (int)sqrtf((float)S / 4.f);   // about 4 sources per cell
    G = G < 2 ? 2 : (G > NN_GMAX ? NN_GMAX : G);
    float4* tbl = reinterpret_cast<float4*>(ws);
    float4* sorted = tbl + (size_t)B * S;
    int* hdr = reinterpret_cast<int*>(sorted + (size_t)B * T);
    int* order = hdr + (size_t)B * NN_HDR;
    hipLaunchKernelGGL(nn_grid_build_kernel, dim3(B), dim3(256), 0, st, src_soa, S, G, tbl, hdr);
    if (sn2_small_sort_wg(B) && T <= 16 * 1024)        // many plots: 256 threads per plot (common.h)
        hipLaunchKernelGGL((nn_target_sort_chunk_kernel<16, 256>), dim3(B), dim3(256), 0, st, dst_soa, T, (const int*)hdr, order, sorted);
    else if (T <= 8 * 1024)
        hipLaunchKernelGGL((nn_target_sort_chunk_kernel<8, 1024>), dim3(B), dim3(1024), 0, st, dst_soa, T, (const int*)hdr, order, sorted);
    else
        hipLaunchKernelGGL((nn_target_sort_chunk_kernel<16, 1024>), dim3(B), dim3(1024), 0, st, dst_soa, T, (const int*)hdr, order, sorted);
    const size_t lds = (size_t)S * 16 + (size_t)(G * G + 1) * 4;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&three_nn_grid_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    {
        const int gx = sn2_cdiv(T, 256);
        static const bool no_xcd = getenv("SN2_NN_NO_XCD") != nullptr;      // (diagnostic switch)
        if (!no_xcd && (B % 8 == 0 || B >= 64))      // plots balance over the XCDs: XCD-aware placement
            hipLaunchKernelGGL(three_nn_grid_kernel, dim3(8 * sn2_cdiv(B, 8) * gx), dim3(256), lds, st, (const float4*)tbl,
                               (const int*)hdr, S, T, k, (const int*)order, (const float4*)sorted, idx, w, B, gx);
        else
            hipLaunchKernelGGL(three_nn_grid_kernel, dim3(gx, B), dim3(256), lds, st, (const float4*)tbl,
                               (const int*)hdr, S, T, k, (const int*)order, (const float4*)sorted, idx, w, 0, gx);
    }
    SN2_RETURN_LAUNCH();
}

extern "C" int sn2_three_nn(const float* src_soa, int B, int S, const float* dst_soa, int T, int k, int* idx, float* w,
                            void* ws, const int* dst_fps_ws, void* stream) {
    if (!src_soa || !dst_soa || !idx || !w || B <= 0 || S <= 0 || T <= 0 || k < 1 || k > 3) return SN2_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // grid search: needs the targets in the spatial order sn2_fps built for them (workspace laid out as there)
    if (ws && dst_fps_ws && S >= 128 && S <= 8192 && T > 2048 && (((size_t)B * T) % 4 == 0) && (((size_t)ws) % 16 == 0)) {
        int G = (int)sqrtf((float)S / 4.f);   // about 4 sources per cell (2, 3, 8 and 12 per cell measured slower)
        G = G < 2 ? 2 : (G > NN_GMAX ? NN_GMAX : G);
        float4* tbl = reinterpret_cast<float4*>(ws);
        int* hdr = reinterpret_cast<int*>(tbl + (size_t)B * S);
        hipLaunchKernelGGL(nn_grid_build_kernel, dim3(B), dim3(256), 0, st, src_soa, S, G, tbl, hdr);
        const size_t lds = (size_t)S * 16 + (size_t)(G * G + 1) * 4;
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&three_nn_grid_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(three_nn_grid_kernel, dim3(sn2_cdiv(T, 256), B), dim3(256), lds, st, (const float4*)tbl,
                           (const int*)hdr, S, T, k, dst_fps_ws, reinterpret_cast<const float4*>(dst_fps_ws + (size_t)B * T),
                           idx, w, 0, sn2_cdiv(T, 256));
        SN2_RETURN_LAUNCH();
    }
    dim3 grid(sn2_cdiv(T, 256), B);
    hipLaunchKernelGGL(three_nn_kernel, grid, dim3(256), 0, st, src_soa, S, dst_soa, T, k, idx, w);
    SN2_RETURN_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// z-normalisation of a raw plot: z_i - min{ z_j : |xy_i - xy_j| <= r }  -- normalize_z_with_minz_in_a_radius,
// /root/reference/utils/load_data.py:237-249 (an sklearn kd-tree radius query over x,y, then a python loop over all points).
// sklearn works on the float64 copies of the coordinates and keeps neighbours with reduced distance
// (dx*dx + dy*dy, summed in that order) <= r*r, inclusive: the same test is made here in fp64 without contraction.
// Points are binned into square cells of side r (counting sort through global memory: a raw plot is one variable-length
// cloud), every point then scans the 3 x 3 cells around its own.
// ------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void znorm_cell_kernel(const float* __restrict__ x, const float* __restrict__ y, int n,
                                                         float x0, float y0, float inv, int GX, int GY,
                                                         int* __restrict__ cell, int* __restrict__ hist) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int cx = (int)((x[i] - x0) * inv), cy = (int)((y[i] - y0) * inv);
    cx = cx < 0 ? 0 : (cx > GX - 1 ? GX - 1 : cx);
    cy = cy < 0 ? 0 : (cy > GY - 1 ? GY - 1 : cy);
    const int c = cy * GX + cx;
    cell[i] = c;
    atomicAdd(&hist[c], 1);
}

// exclusive scan of hist (ncell <= 2^20) by one workgroup; start[c], and cursor[c] = start[c] for the scatter
__global__ __launch_bounds__(1024) void znorm_scan_kernel(const int* __restrict__ hist, int ncell, int* __restrict__ start,
                                                          int* __restrict__ cursor) {
    __shared__ int s_tot[1024];
    const int tid = threadIdx.x;
    const int per = (ncell + 1023) / 1024;
    int sum = 0;
    for (int k = 0; k < per; ++k) {
        const int c = tid * per + k;
        if (c < ncell) sum += hist[c];
    }
    s_tot[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int t = 0; t < 1024; ++t) { const int v = s_tot[t]; s_tot[t] = run; run += v; }
    }
    __syncthreads();
    int run = s_tot[tid];
    for (int k = 0; k < per; ++k) {
        const int c = tid * per + k;
        if (c < ncell) {
            start[c] = run;
            cursor[c] = run;
            run += hist[c];
        }
    }
    if (tid == 1023) start[ncell] = run;
}

__global__ __launch_bounds__(256) void znorm_scatter_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                            const float* __restrict__ z, int n, const int* __restrict__ cell,
                                                            int* __restrict__ cursor, float4* __restrict__ sorted) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int p = atomicAdd(&cursor[cell[i]], 1);
    sorted[p] = make_float4(x[i], y[i], z[i], 0.f);
}

__global__ __launch_bounds__(256) void znorm_query_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          const float* __restrict__ z, int n, const int* __restrict__ cell,
                                                          const int* __restrict__ start, const float4* __restrict__ sorted,
                                                          int GX, int GY, double r2, float* __restrict__ zmin,
                                                          float* __restrict__ z_out) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double qx = (double)x[i], qy = (double)y[i];
    const int c = cell[i], cx = c % GX, cy = c / GX;
    float best = z[i];                                    // the point is its own neighbour (distance 0)
    for (int yy = (cy > 0 ? cy - 1 : 0); yy <= (cy < GY - 1 ? cy + 1 : GY - 1); ++yy) {
        const int c_lo = yy * GX + (cx > 0 ? cx - 1 : 0), c_hi = yy * GX + (cx < GX - 1 ? cx + 1 : GX - 1);
        for (int p = start[c_lo]; p < start[c_hi + 1]; ++p) {     // three adjacent cells of a row are contiguous
            const float4 s = sorted[p];
            const double dx = qx - (double)s.x, dy = qy - (double)s.y;
            const double d2 = dx * dx + dy * dy;
            if (d2 <= r2) best = fminf(best, s.z);
        }
    }
    if (zmin) zmin[i] = best;
    if (z_out) z_out[i] = (float)((double)z[i] - (double)best);   // float32 array minus a python list of minima: fp64, then cast
}
}  // namespace

extern "C" int sn2_znorm(const float* x, const float* y, const float* z, int n, float radius, float x_min, float y_min,
                         float x_max, float y_max, int* ws, float* zmin, float* z_out, void* stream) {
    if (!x || !y || !z || !ws || (!zmin && !z_out) || n <= 0 || !(radius > 0.f) || !(x_max >= x_min) || !(y_max >= y_min))
        return SN2_EINVAL;
    // cells of side >= radius (a point's neighbours are then within the 3 x 3 block around its cell)
    const float inv = 1.0f / (radius * 1.0001f);
    const long gx = (long)((x_max - x_min) * inv) + 1, gy = (long)((y_max - y_min) * inv) + 1;
    if (gx * gy > (1L << 20)) return SN2_ELIMIT;
    const int GX = (int)gx, GY = (int)gy, ncell = GX * GY;
    hipStream_t st = (hipStream_t)stream;
    int* cell = ws;                                  // n
    int* hist = cell + n;                            // ncell
    int* start = hist + ncell;                       // ncell + 1
    int* cursor = start + ncell + 1;                 // ncell
    float4* sorted = reinterpret_cast<float4*>(ws + (((size_t)n + 3 * (size_t)ncell + 1 + 3) & ~(size_t)3));   // 16-byte aligned
    sn2_fill_words(hist, 0u, (size_t)ncell, st);
    const int blocks = sn2_cdiv(n, 256);
    hipLaunchKernelGGL(znorm_cell_kernel, dim3(blocks), dim3(256), 0, st, x, y, n, x_min, y_min, inv, GX, GY, cell, hist);
    hipLaunchKernelGGL(znorm_scan_kernel, dim3(1), dim3(1024), 0, st, (const int*)hist, ncell, start, cursor);
    hipLaunchKernelGGL(znorm_scatter_kernel, dim3(blocks), dim3(256), 0, st, x, y, z, n, (const int*)cell, cursor, sorted);
    const double r2 = (double)radius * (double)radius;
    hipLaunchKernelGGL(znorm_query_kernel, dim3(blocks), dim3(256), 0, st, x, y, z, n, (const int*)cell, (const int*)start,
                       (const float4*)sorted, GX, GY, r2, zmin, z_out);
    SN2_RETURN_LAUNCH();
}
