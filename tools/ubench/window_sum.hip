// k_orient_describe's window sums in isolation: 109 ordered additions of {Lx, Ly} into 42 lanes (windows), each sample
// added only in the lanes whose window contains it.  Which form of "only in these lanes" is cheapest on this part?
//   A  the sample's 64-bit membership mask -> SGPR pair (v_readlane x 2) -> EXEC, one v_pk_add_f32 (the kernel's form)
//   B  the same mask as the SGPR selector of two v_cndmask_b32, then the packed add (no EXEC write)
//   C  per-lane bit test of the mask words (shift, and, compare, two selects, add)
//   D  A with the masks ALREADY in SGPRs (scalar loads): what the additions alone cost
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/window_sum.hip -o /tmp/window_sum
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256, 7) void k_win(const uint2* __restrict__ masks, const float2* __restrict__ vals, float2* __restrict__ out,
                                                int reps)
{
    __shared__ float2 s_r[4][128];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t base = ((size_t)blockIdx.x * 4 + wv) * 128;
    v2f acc = {0.0f, 0.0f};
    for (int rep = 0; rep < reps; ++rep) {
        s_r[wv][lane] = vals[base + lane];
        s_r[wv][lane + 64] = vals[base + lane + 64];
        const uint2 m0 = masks[base + lane], m1 = masks[base + lane + 64];
        v2f sum = {0.0f, 0.0f};
        if (MODE == 3) {
            const uint2* ms = masks + (size_t)__builtin_amdgcn_readfirstlane((int)base);      // scalar loads
#pragma unroll
            for (int k = 0; k < 108; k += 4) {
                const uint2 a = ms[k], b = ms[k + 1], c = ms[k + 2], d = ms[k + 3];
                const unsigned long long q0 = ((unsigned long long)a.y << 32) | a.x, q1 = ((unsigned long long)b.y << 32) | b.x,
                                         q2 = ((unsigned long long)c.y << 32) | c.x, q3 = ((unsigned long long)d.y << 32) | d.x;
                const float2 r0 = s_r[wv][k], r1 = s_r[wv][k + 1], r2 = s_r[wv][k + 2], r3 = s_r[wv][k + 3];
                const v2f v0 = {r0.x, r0.y}, v1 = {r1.x, r1.y}, v2 = {r2.x, r2.y}, v3 = {r3.x, r3.y};
                unsigned long long saved;
                asm("s_mov_b64 %[sv], exec\n\ts_mov_b64 exec, %[m0]\n\tv_pk_add_f32 %[s], %[s], %[r0]\n\t"
                    "s_mov_b64 exec, %[m1]\n\tv_pk_add_f32 %[s], %[s], %[r1]\n\t"
                    "s_mov_b64 exec, %[m2]\n\tv_pk_add_f32 %[s], %[s], %[r2]\n\t"
                    "s_mov_b64 exec, %[m3]\n\tv_pk_add_f32 %[s], %[s], %[r3]\n\ts_mov_b64 exec, %[sv]"
                    : [s] "+v"(sum), [sv] "=&s"(saved)
                    : [m0] "s"(q0), [m1] "s"(q1), [m2] "s"(q2), [m3] "s"(q3), [r0] "v"(v0), [r1] "v"(v1), [r2] "v"(v2), [r3] "v"(v3));
            }
        } else {
#pragma unroll
            for (int k = 0; k < 108; ++k) {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(k < 64 ? m0.x : m1.x), k & 63);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(k < 64 ? m0.y : m1.y), k & 63);
                const float2 rk = s_r[wv][k];
                v2f rv = {rk.x, rk.y};
                if (MODE == 0) {
                    const unsigned long long q = ((unsigned long long)hi << 32) | lo;
                    unsigned long long saved;
                    asm("s_mov_b64 %[sv], exec\n\ts_mov_b64 exec, %[m]\n\tv_pk_add_f32 %[s], %[s], %[r]\n\ts_mov_b64 exec, %[sv]"
                        : [s] "+v"(sum), [sv] "=&s"(saved)
                        : [m] "s"(q), [r] "v"(rv));
                } else if (MODE == 1) {
                    const unsigned long long q = ((unsigned long long)hi << 32) | lo;
                    float x, y;
                    asm("v_cndmask_b32 %0, 0, %2, %3\n\tv_cndmask_b32 %1, 0, %4, %3" : "=&v"(x), "=&v"(y) : "v"(rv.x), "s"(q), "v"(rv.y));
                    sum += (v2f){x, y};
                } else {
                    const bool in = (((lane >> 5) ? hi : lo) >> (lane & 31)) & 1u;
                    sum += (v2f){in ? rv.x : 0.0f, in ? rv.y : 0.0f};
                }
            }
        }
        acc += sum;
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = make_float2(acc.x, acc.y);
}

template <int MODE>
static void run(const char* what, const uint2* dm, const float2* dv, float2* dout, int blocks)
{
    const int reps = 20;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_win<MODE>, dim3(blocks), dim3(256), 0, 0, dm, dv, dout, 2);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_win<MODE>, dim3(blocks), dim3(256), 0, 0, dm, dv, dout, reps);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    // SIMD cycles per (wave, sample) at 2.4 GHz: every SIMD runs blocks * 4 / 1024 waves' worth of work
    const double waves_per_simd = blocks * 4.0 / 1024.0;
    printf("%-52s %8.3f ms   %6.1f SIMD cycles per sample and wave\n", what, best, best * 1e-3 * 2.4e9 / (waves_per_simd * reps * 108.0));
}

int main()
{
    const int blocks = 256 * 7 * 4;                 // seven waves per SIMD, four rounds
    const size_t n = (size_t)blocks * 4 * 128;
    uint2* hm = (uint2*)malloc(n * sizeof(uint2));
    float2* hv = (float2*)malloc(n * sizeof(float2));
    unsigned s = 12345;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        const int start = (s >> 8) % 42;                                   // a run of 7 windows, wrapping, like the real masks
        unsigned long long m = 0;
        for (int j = 0; j < 7; ++j) m |= 1ull << ((start + j) % 42);
        hm[i] = make_uint2((unsigned)m, (unsigned)(m >> 32));
        hv[i] = make_float2((float)((s >> 4) & 1023) * 1e-3f, (float)((s >> 14) & 1023) * -1e-3f);
    }
    uint2* dm; float2 *dv, *dout;
    hipMalloc(&dm, n * sizeof(uint2)); hipMalloc(&dv, n * sizeof(float2)); hipMalloc(&dout, (size_t)blocks * 256 * sizeof(float2));
    hipMemcpy(dm, hm, n * sizeof(uint2), hipMemcpyHostToDevice);
    hipMemcpy(dv, hv, n * sizeof(float2), hipMemcpyHostToDevice);
    run<0>("A  mask -> SGPR pair -> EXEC, v_pk_add_f32", dm, dv, dout, blocks);
    run<1>("B  mask -> SGPR pair as v_cndmask selector, v_pk_add", dm, dv, dout, blocks);
    run<2>("C  per-lane bit test, two selects, add", dm, dv, dout, blocks);
    run<3>("D  masks by scalar loads -> EXEC, v_pk_add_f32", dm, dv, dout, blocks);
    return 0;
}
