#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
// A: 32 rows x 64 k (fp4), B: 64 k x 32 cols. lane l: row/col = l & 31, k block = l >> 5 (32 values = 16 bytes)
__global__ void k(const uint4* a, const uint4* b, float* out, float init, int scaled)
{
    const int lane = threadIdx.x;
    uint4 av = a[lane], bv = b[lane];
    v8i A = {(int)av.x, (int)av.y, (int)av.z, (int)av.w, 0, 0, 0, 0};
    v8i B = {(int)bv.x, (int)bv.y, (int)bv.z, (int)bv.w, 0, 0, 0, 0};
    v16f c;
    for (int i = 0; i < 16; ++i) c[i] = init;
    // scaled: A carries the E8M0 scale 2^5 (0x84), so the product sum arrives multiplied by 32
    if (scaled) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 4, 4, 0, 0x84848484, 0, 0x7F7F7F7F);
    else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    for (int i = 0; i < 16; ++i) out[lane * 16 + i] = c[i];
}
int main()
{
    // random +-1 vectors: A[32][64], B[32][64] (B indexed by column)
    std::vector<int> A(32 * 64), B(32 * 64);
    srand(5);
    for (auto& v : A) v = (rand() & 1) ? 1 : -1;
    for (auto& v : B) v = (rand() & 1) ? 1 : -1;
    std::vector<uint32_t> ha(64 * 4), hb(64 * 4);
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 32; ++j) {
            int k = (l >> 5) * 32 + j;
            uint32_t na = A[(l & 31) * 64 + k] > 0 ? 0x2u : 0xAu, nb = B[(l & 31) * 64 + k] > 0 ? 0x2u : 0xAu;
            ha[l * 4 + j / 8] |= na << (4 * (j % 8));
            hb[l * 4 + j / 8] |= nb << (4 * (j % 8));
        }
    uint4 *da, *db; float* dout;
    hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dout, 64 * 16 * 4);
    hipMemcpy(da, ha.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 1024, hipMemcpyHostToDevice);
    for (int scaled = 0; scaled < 2; ++scaled)
    for (float init : {0.0f, 12582912.0f}) {
        k<<<1, 64>>>(da, db, dout, init, scaled);
        std::vector<float> o(1024);
        hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                int dot = 0;
                for (int kk = 0; kk < 64; ++kk) dot += A[row * 64 + kk] * B[col * 64 + kk];
                if (o[l * 16 + r] != init + (float)(scaled ? 32 * dot : dot)) { if (bad < 5) printf("lane %d r %d got %f want %f\n", l, r, o[l*16+r], init + dot); ++bad; }
            }
        printf("scaled %d init %.1f: %d mismatches\n", scaled, init, bad);
    }
    return 0;
}
