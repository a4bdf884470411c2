// Microbenchmark: unfused f32 multiply+add throughput, scalar (v_mul_f32 + v_add_f32) vs packed
// (v_pk_mul_f32 + v_pk_add_f32), same number of element operations.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
// Result on MI355X: packed 75 T element-ops/s (= the f32 vector peak), scalar (-fno-slp-vectorize) 52.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int PK>
__global__ __launch_bounds__(256) void k(float* out, float k0, float k1, int iters)
{
    float s = threadIdx.x * 1e-3f;
    if (PK) {
        v2f a[8], x[8];
        for (int i = 0; i < 8; ++i) { a[i] = (v2f){0.f, 0.f}; x[i] = (v2f){s + i, s - i}; }
        v2f kk = {k0, k0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v2f p = x[i] * kk;
                a[i] = p + a[i];
                x[i] = x[i] + (v2f){k1, k1};
            }
        }
        v2f r = {0, 0};
        for (int i = 0; i < 8; ++i) r += a[i];
        out[blockIdx.x * 256 + threadIdx.x] = r.x + r.y;
    } else {
        float a[16], x[16];
        for (int i = 0; i < 16; ++i) { a[i] = 0.f; x[i] = s + i; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float p = x[i] * k0;
                a[i] = p + a[i];
                x[i] = x[i] + k1;
            }
        }
        float r = 0;
        for (int i = 0; i < 16; ++i) r += a[i];
        out[blockIdx.x * 256 + threadIdx.x] = r;
    }
}

int main()
{
    float* d;
    hipMalloc(&d, 256 * 4096 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096, blocks = 4096;
    for (int pk = 0; pk < 2; ++pk)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (pk) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, iters);
            else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double elem_ops = 3.0 * 16 * (double)iters * blocks * 256;
            printf("pk=%d  %.3f ms  %.2f T elem-ops/s\n", pk, ms, elem_ops / ms / 1e9);
        }
    return 0;
}
