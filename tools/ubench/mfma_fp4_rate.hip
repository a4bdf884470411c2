// What rate does v_mfma_scale_f32_32x32x64_f8f6f4 with E2M1 (FP4) operands reach on this part when nothing else is in the
// way?  CHAINS independent accumulator chains per wave, WAVES waves per SIMD, operands resident in registers, no LDS, no
// epilogue.  The matcher's roofline fraction is quoted against the 10 PF/s of MI355X_MICROARCH.md; this is the figure the
// instruction itself sustains here (clock under load included).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_fp4_rate.hip -o /tmp/mfma_fp4_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i32 __attribute__((ext_vector_type(8)));
typedef float v16f32 __attribute__((ext_vector_type(16)));

// the same with NV integer VALU instructions (a dependent v_min / v_max chain on registers the MFMAs never touch) behind
// every MFMA, the order pinned: does the vector ALU run in the matrix pipe's shadow?
template <int CHAINS, int NV>
__global__ __launch_bounds__(256) void k_rate_valu(float* out, int iters, int seed)
{
    v8i32 a = {seed + (int)threadIdx.x, 0x2A2A2A2A, 0x22222222, (int)0xAAAAAAAA, 0, 0, 0, 0};
    v8i32 b = {0x2222AAAA, seed, 0x2A2A2A2A, (int)0xA2A2A2A2, 0, 0, 0, 0};
    v16f32 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = (float)(c + i);
    int l0 = seed, l1 = seed + 7, x = (int)threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                acc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[c], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    if (v & 1) l1 = max(l1, l0 ^ x);
                    else l0 = min(l0, l1 + x);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    float s = (float)(l0 + l1);
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[c][i];
    if (s == 123.456f) out[0] = s;
}

template <int CHAINS>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters, int seed)
{
    v8i32 a = {seed + (int)threadIdx.x, 0x2A2A2A2A, 0x22222222, (int)0xAAAAAAAA, 0, 0, 0, 0};
    v8i32 b = {0x2222AAAA, seed, 0x2A2A2A2A, (int)0xA2A2A2A2, 0, 0, 0, 0};
    v16f32 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = (float)(c + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c)
                acc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[c], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    }
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[c][i];
    if (s == 123.456f) out[0] = s;      // never true: keeps the chains alive
}

template <int CHAINS>
static void run(int blocks_per_cu, const char* what)
{
    float* d = nullptr;
    hipMalloc(&d, 4);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 4000;
    const dim3 grid(cus * blocks_per_cu), block(256);          // 256 threads = one wave per SIMD and block
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<CHAINS>, grid, block, 0, 0, d, 100, 1);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_rate<CHAINS>, grid, block, 0, 0, d, iters, 1);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double mfmas = (double)grid.x * 4.0 * iters * 8.0 * CHAINS;            // wave-level instructions
    const double ops = mfmas * 32.0 * 32.0 * 64.0 * 2.0;
    printf("%-28s chains %d  waves/SIMD %d  %8.3f ms  %7.1f TOP/s  (%.1f cycles per MFMA and SIMD at 2.4 GHz)\n", what, CHAINS, blocks_per_cu,
           best, ops / (best * 1e-3) / 1e12, best * 1e-3 * 2.4e9 / (mfmas / (cus * 4.0)));
    hipFree(d);
}

template <int CHAINS, int NV>
static void run_valu(int blocks_per_cu)
{
    float* d = nullptr;
    hipMalloc(&d, 4);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 4000;
    const dim3 grid(cus * blocks_per_cu), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_rate_valu<CHAINS, NV>), grid, block, 0, 0, d, 100, 1);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_rate_valu<CHAINS, NV>), grid, block, 0, 0, d, iters, 1);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double mfmas = (double)grid.x * 4.0 * iters * 8.0 * CHAINS;
    const double ops = mfmas * 32.0 * 32.0 * 64.0 * 2.0;
    printf("%d VALU behind every MFMA      chains %d  waves/SIMD %d  %8.3f ms  %7.1f TOP/s  (%.1f cycles per MFMA and SIMD at 2.4 GHz)\n", NV, CHAINS,
           blocks_per_cu, best, ops / (best * 1e-3) / 1e12, best * 1e-3 * 2.4e9 / (mfmas / (cus * 4.0)));
    hipFree(d);
}

int main()
{
    run_valu<2, 2>(2);
    run_valu<2, 4>(2);
    run_valu<2, 6>(2);
    run_valu<2, 8>(2);
    run_valu<2, 4>(1);
    run_valu<2, 4>(4);
    run_valu<1, 4>(4);
    run<1>(1, "one dependent chain");
    run<2>(1, "two chains");
    run<4>(1, "four chains");
    run<1>(2, "one chain, two waves");
    run<2>(2, "two chains, two waves");
    run<1>(4, "one chain, four waves");
    run<2>(4, "two chains, four waves");
    return 0;
}
