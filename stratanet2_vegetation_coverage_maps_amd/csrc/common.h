// common.h -- device helpers shared by the gfx950 kernels of libstrata_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/strata_hip.h"

#define SN2_WAVE 64

#define SN2_RETURN_LAUNCH()                                   \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        return e__ == hipSuccess ? 0 : (int)e__;              \
    } while (0)

#define SN2_TRY(expr)                \
    do {                             \
        int r__ = (expr);            \
        if (r__ != 0) return r__;    \
    } while (0)

// every gradient sum that leaves a kernel through a global float atomic goes through this macro (one place to count them)
#ifdef SN2_NO_FLUSH   /* timing experiments only: the adds are kept alive but never executed */
#define SN2_FLUSH_ADD(ptr, v) do { if ((v) == 12345.678f) atomicAdd((ptr), (v)); } while (0)
#else
#define SN2_FLUSH_ADD(ptr, v) atomicAdd((ptr), (v))
#endif

// sn2_block.grad_replicas images of (dW, db), grad_replica_stride floats apart: a workgroup adds into image
// blockIdx.x % replicas.  Thousands of workgroups adding the same few KB serialise at the memory-side atomic units (the
// float atomics run at full rate only when spread over many channels): 100 us per training step went into these tails.
#ifdef __HIPCC__
__device__ __forceinline__ int sn2_grad_image(int replicas, int stride) {
    return replicas > 1 ? (int)(blockIdx.x % (unsigned)replicas) * stride : 0;
}
#endif

static inline int sn2_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- clearing a buffer.  Always a KERNEL, never hipMemsetAsync: every entry point may be captured into a hipGraph, and on
// this runtime (ROCm 7.2) a captured memset NODE is only right on the FIRST replay -- from the second replay on it fills
// the buffer with whatever its recycled argument block holds (scripts/debug_graph_d2h.py::memset_node: buf := 0; buf += 1
// reads 1 once, then 97736275787777 = a pointer-looking word + 1).  That was round 1's NaN loss under `bench.py
// --host-inputs`: the key table of the plot-wise projection came back "cleared" to garbage, empty pixels decoded to NaN;
// it stayed hidden in the resident mode only because the stale block still held zeros until a device-to-host copy
// reused it.  A kernel node carries its arguments with it and is ordered like any other node.
#ifdef __HIPCC__
static __global__ __launch_bounds__(256) void sn2_fill_words_kernel(uint32_t* __restrict__ p, uint32_t v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
static inline void sn2_fill_words(void* p, uint32_t v, size_t nwords, hipStream_t st) {
    if (nwords == 0) return;
    int grid = sn2_cdiv((long)nwords, 256 * 4);
    grid = grid < 1 ? 1 : (grid > 2048 ? 2048 : grid);
    hipLaunchKernelGGL(sn2_fill_words_kernel, dim3(grid), dim3(256), 0, st, (uint32_t*)p, v, nwords);
}
#endif

// One-workgroup-per-plot kernels (the sorts): 1024 threads per plot are the shortest pass for a few plots (a training batch),
// but a 1024-thread workgroup needs a CU with sixteen free wave slots AT ONCE, and beside full-chip kernels of four-wave
// workgroups (which take every slot the moment it frees up) it starves: in the parcel loop (256 plots per launch, four passes
// in flight) the target sort of the 3-NN search took 0.82 ms instead of 0.03, the spatial sort 0.46 instead of 0.06.  With
// many plots the chip is full anyway: 256 threads per plot.
static inline bool sn2_small_sort_wg(int B) { return B > 32; }

// compute units of the current device (256 on MI355X), for the grids of the persistent kernels
static inline int sn2_cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            n = v;
        else
            n = 256;
    }
    return n;
}

// ---- wave-uniform read-only tables (weights, BN constants) -------------------------------------------------------
// `cfp` = the same global memory viewed through the CONSTANT address space: uniform loads from it always become
// s_load (the memory is not written while the kernel runs).  `opaque()` hides a pointer's provenance from the
// optimiser for one loop iteration: without it LICM hoists the 400-1400 weight loads of an MLP out of the row loop into
// SGPRs, of which there are ~100 -- the rest is spilled to VGPR lanes and every FMA then pays two v_readlane
// (head_fwd: 6355 v_readlane for 698 FMAs; the SA passes 2300-2800).  With it the weights stream through s_load_dwordx16
// inside the loop (scalar-cache hits) and no SGPR is spilled.
typedef const float __attribute__((address_space(4)))* cfp;
__device__ __forceinline__ cfp as_const(const float* p) { return (cfp)(uintptr_t)p; }
__device__ __forceinline__ cfp opaque(cfp p) {
    uint64_t v = (uint64_t)p;
    asm volatile("" : "+s"(v));
    return (cfp)v;
}

// ---- canonical squared distance: (dx*dx + dy*dy) + dz*dz, every operation rounded to fp32 on its
// own.  The pragma removes the `contract` flag from these operations so the backend cannot fuse them into FMAs even
// after inlining into a kernel compiled with the default -ffp-contract=fast.
__device__ __forceinline__ float sn2_d2(float ax, float ay, float az, float bx, float by, float bz) {
#pragma clang fp contract(off)
    float dx = ax - bx;
    float dy = ay - by;
    float dz = az - bz;
    float xx = dx * dx;
    float yy = dy * dy;
    float zz = dz * dz;
    float s = xx + yy;
    return s + zz;
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long t = __shfl_xor(v, o);
        v = t > v ? t : v;
    }
    return v;
}

// ---- wave64 reductions on the DPP network (6 VALU ops, no LDS round trip; __shfl_xor lowers to ds_bpermute, whose
// ~64-cycle latency per step dominated the sequential rounds of the FPS kernel).  Result is wave-uniform.
#define SN2_DPP(x, ctrl, rmask) __builtin_amdgcn_update_dpp((x), (x), (ctrl), (rmask), 0xF, false)
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = fmaxf(v, __int_as_float(SN2_DPP(__float_as_int(v), 0xB1, 0xF)));   // quad_perm [1,0,3,2]
    v = fmaxf(v, __int_as_float(SN2_DPP(__float_as_int(v), 0x4E, 0xF)));   // quad_perm [2,3,0,1]
    v = fmaxf(v, __int_as_float(SN2_DPP(__float_as_int(v), 0x141, 0xF)));  // row_half_mirror
    v = fmaxf(v, __int_as_float(SN2_DPP(__float_as_int(v), 0x140, 0xF)));  // row_mirror: every lane = its row's max
    v = fmaxf(v, __int_as_float(SN2_DPP(__float_as_int(v), 0x142, 0xA)));  // row_bcast15 into rows 1,3
    v = fmaxf(v, __int_as_float(SN2_DPP(__float_as_int(v), 0x143, 0xC)));  // row_bcast31 into rows 2,3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_min_dpp(float v) { return -wave_max_dpp(-v); }
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += __int_as_float(SN2_DPP(__float_as_int(v), 0xB1, 0xF));   // pairs
    v += __int_as_float(SN2_DPP(__float_as_int(v), 0x4E, 0xF));   // quads
    v += __int_as_float(SN2_DPP(__float_as_int(v), 0x141, 0xF));  // halves of a row
    v += __int_as_float(SN2_DPP(__float_as_int(v), 0x140, 0xF));  // rows of 16
    v += __int_as_float(SN2_DPP(__float_as_int(v), 0x142, 0xA));  // row 1 += row 0, row 3 += row 2
    v += __int_as_float(SN2_DPP(__float_as_int(v), 0x143, 0xC));  // rows 2,3 += rows 0+1
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// the names used throughout the kernels
__device__ __forceinline__ float wave_sum(float v) { return wave_sum_dpp(v); }
__device__ __forceinline__ float wave_max(float v) { return wave_max_dpp(v); }
__device__ __forceinline__ float wave_min(float v) { return wave_min_dpp(v); }
__device__ __forceinline__ unsigned wave_min_u32_dpp(unsigned v) {
    v = min(v, (unsigned)SN2_DPP((int)v, 0xB1, 0xF));
    v = min(v, (unsigned)SN2_DPP((int)v, 0x4E, 0xF));
    v = min(v, (unsigned)SN2_DPP((int)v, 0x141, 0xF));
    v = min(v, (unsigned)SN2_DPP((int)v, 0x140, 0xF));
    v = min(v, (unsigned)SN2_DPP((int)v, 0x142, 0xA));
    v = min(v, (unsigned)SN2_DPP((int)v, 0x143, 0xC));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// order-preserving map float -> uint32 (any non-NaN float maps to a value > 0)
__device__ __forceinline__ uint32_t f2ord(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __uint_as_float(u);
}
