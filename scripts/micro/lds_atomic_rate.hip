// Microbenchmark: the rate of ds_add_f32 (no return) on gfx950 for the access shapes a scatter form of FP1's source pass would
// make: 7 rows x 9 quad-lanes per wave, each lane adding 4 consecutive floats of a [1024][STRIDE] table row in LDS.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic_rate lds_atomic_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE, int STRIDE>
__global__ __launch_bounds__(1024) void k(const int* __restrict__ src_of, int iters, float* __restrict__ out) {
    extern __shared__ float tab[];                       // [1024][STRIDE]
    for (int i = threadIdx.x; i < 1024 * STRIDE; i += 1024) tab[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane % 9, g = lane / 9;
    const bool on = lane < 63;
    unsigned h = (blockIdx.x * 16 + wave) * 7919u + g * 104729u;
    for (int it = 0; it < iters; ++it) {
        int s;
        if (MODE == 0) s = lane;                                       // conflict-free dword per lane (address = lane + 64 t)
        else if (MODE == 1) { h = h * 1664525u + 1013904223u; s = (h >> 10) & 1023; }      // random source per row
        else s = src_of[(((blockIdx.x * 16 + wave) * iters + it) * 7 + g) & 0xFFFFF];       // table-driven (spatially coherent)
        const float v = 1.0f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            int ss = MODE == 0 ? s : ((s + j * 37) & 1023);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float* p = MODE == 0 ? &tab[lane + 64 * (4 * j + t)] : &tab[ss * STRIDE + 4 * q + t];
                if (on) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();
    float acc = 0.f;
    for (int i = threadIdx.x; i < 1024 * STRIDE; i += 1024) acc += tab[i];
    if (acc == 12345.f) out[0] = acc;
}

template <int MODE, int STRIDE>
void run(const char* name, const int* d_src, float* d_out, int iters) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    size_t lds = 1024 * STRIDE * 4;
    CK(hipFuncSetAttribute((const void*)k<MODE, STRIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k<MODE, STRIDE>), dim3(256), dim3(1024), lds, 0, d_src, 2, d_out);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((k<MODE, STRIDE>), dim3(256), dim3(1024), lds, 0, d_src, 0, d_out);     // the fixed part
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<MODE, STRIDE>), dim3(256), dim3(1024), lds, 0, d_src, 0, d_out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float t0; CK(hipEventElapsedTime(&t0, a, b));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<MODE, STRIDE>), dim3(256), dim3(1024), lds, 0, d_src, iters, d_out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float t1; CK(hipEventElapsedTime(&t1, a, b));
    const double us = (t1 - t0) * 1e3;
    const double instr_per_cu = 16.0 * iters * 12;                 // wave-instructions per CU
    const double laneops = instr_per_cu * 63;
    printf("%-44s stride %2d: %8.1f us for %d iters (fixed %.1f us): %.1f ns per wave-instruction per CU, %.2f lane-ops/ns/CU\n", name,
           STRIDE, us, iters, t0 * 1e3, us * 1e3 / instr_per_cu, laneops / (us * 1e3));
}

int main() {
    std::vector<int> src(1 << 20);
    // spatially coherent stand-in: consecutive rows hit sources that move slowly (random walk over a 32 x 32 grid)
    unsigned h = 12345; int x = 16, y = 16;
    for (size_t i = 0; i < src.size(); ++i) {
        h = h * 1664525u + 1013904223u;
        x = (x + (int)((h >> 8) % 3) - 1) & 31; y = (y + (int)((h >> 16) % 3) - 1) & 31;
        src[i] = y * 32 + x;
    }
    int* d_src; float* d_out;
    CK(hipMalloc(&d_src, src.size() * 4)); CK(hipMalloc(&d_out, 64));
    CK(hipMemcpy(d_src, src.data(), src.size() * 4, hipMemcpyHostToDevice));
    const int iters = 293;        // 2048 rows per CU x ... : a CU's share of FP1's rows is 2048 rows = 293 groups of 7 over 16 waves -> ~18; x16 for timing
    run<0, 36>("conflict-free (lane + 64 k)", d_src, d_out, iters);
    run<1, 36>("random source per row, 9 lanes x 4 floats", d_src, d_out, iters);
    run<1, 37>("random source per row, 9 lanes x 4 floats", d_src, d_out, iters);
    run<1, 40>("random source per row, 9 lanes x 4 floats", d_src, d_out, iters);
    run<2, 36>("random-walk sources (coherent)", d_src, d_out, iters);
    return 0;
}
