// fetch_calib.hip — what rocprofv3's FETCH_SIZE reports for access patterns with a KNOWN byte count, on this chip
// (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern").  The pattern that matters here is
// k_orient_describe's: 8-byte {Lx, Ly} / 4-byte Lt samples that each pull one 32-byte sector of a plane far larger than the
// caches.  Every kernel below touches a fresh region of an 8 GiB buffer exactly once (nothing is re-read, the 256 MiB
// Infinity Cache cannot help), so the bytes HBM must deliver are known at each granularity: 32-byte sectors, 64-byte
// half-lines, 128-byte lines.
//   build: hipcc --offload-arch=gfx950 -O3 tools/ubench/fetch_calib.hip -o /tmp/fetch_calib
//   run:   rocprofv3 --pmc FETCH_SIZE -d out -o calib -- /tmp/fetch_calib     (tools/fetch_calib.sh does both and prints the table)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// every lane reads W bytes; consecutive lanes are STRIDE bytes apart (STRIDE == W: a coalesced stream)
template <typename T, int STRIDE>
__global__ __launch_bounds__(256) void k_strided(const unsigned char* __restrict__ base, size_t n_lanes, unsigned long long* __restrict__ sink)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_lanes) return;
    T v = *reinterpret_cast<const T*>(base + i * STRIDE);
    unsigned first;
    __builtin_memcpy(&first, &v, 4);
    if (first == 0x12345678u) *sink = first;      // never true for the 0x01 fill, but the compiler cannot know: keeps the load
}
// every lane reads 8 bytes from the start of a 32-byte sector chosen by a bijective scramble of its index over the region
// (an odd multiplier modulo a power of two): all sectors of the region are touched exactly once, in an order with no locality
__global__ __launch_bounds__(256) void k_scatter8(const unsigned char* __restrict__ base, size_t n_sectors_pow2, unsigned long long* __restrict__ sink)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_sectors_pow2) return;
    size_t j = (i * 0x9E3779B97F4A7C15ull) & (n_sectors_pow2 - 1);
    unsigned long long v = *reinterpret_cast<const unsigned long long*>(base + j * 32);
    if (v == 0x1234567890ABCDEFull) *sink = v;
}
// the same, but only ONE sector of every 128-byte line is touched (the other three never): what the memory side fetches for
// a lone sector
__global__ __launch_bounds__(256) void k_scatter8_line(const unsigned char* __restrict__ base, size_t n_lines_pow2, unsigned long long* __restrict__ sink)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_lines_pow2) return;
    size_t j = (i * 0x9E3779B97F4A7C15ull) & (n_lines_pow2 - 1);
    unsigned long long v = *reinterpret_cast<const unsigned long long*>(base + j * 128 + 32 * (j & 3));
    if (v == 0x1234567890ABCDEFull) *sink = v;
}

int main()
{
    const size_t total = (size_t)8 << 30, region = (size_t)1 << 30;   // 8 regions of 1 GiB, one per kernel
    unsigned char* buf = nullptr;
    unsigned long long* sink = nullptr;
    CHECK(hipMalloc(&buf, total));
    CHECK(hipMalloc(&sink, 8));
    CHECK(hipMemset(buf, 1, total));
    CHECK(hipDeviceSynchronize());
    size_t r = 0;
    auto grid = [](size_t n) { return dim3((unsigned)((n + 255) / 256)); };
    // name                 lanes                 useful B   sectors(32)  half-lines(64)  lines(128)
    printf("# kernel lanes useful_bytes bytes_at_32 bytes_at_64 bytes_at_128\n");
    {   size_t n = region / 16; hipLaunchKernelGGL((k_strided<uint4, 16>), grid(n), dim3(256), 0, 0, buf + r, n, sink); r += region;
        printf("k_strided<uint4,16> %zu %zu %zu %zu %zu\n", n, n * 16, region, region, region); }
    {   size_t n = region / 8; hipLaunchKernelGGL((k_strided<uint2, 8>), grid(n), dim3(256), 0, 0, buf + r, n, sink); r += region;
        printf("k_strided<uint2,8> %zu %zu %zu %zu %zu\n", n, n * 8, region, region, region); }
    {   size_t n = region / 4; hipLaunchKernelGGL((k_strided<unsigned, 4>), grid(n), dim3(256), 0, 0, buf + r, n, sink); r += region;
        printf("k_strided<unsigned,4> %zu %zu %zu %zu %zu\n", n, n * 4, region, region, region); }
    {   size_t n = region / 32; hipLaunchKernelGGL((k_strided<uint2, 32>), grid(n), dim3(256), 0, 0, buf + r, n, sink); r += region;
        printf("k_strided<uint2,32> %zu %zu %zu %zu %zu\n", n, n * 8, region, region, region); }
    {   size_t n = region / 128; hipLaunchKernelGGL((k_strided<uint2, 128>), grid(n), dim3(256), 0, 0, buf + r, n, sink); r += region;
        printf("k_strided<uint2,128> %zu %zu %zu %zu %zu\n", n, n * 8, n * 32, n * 64, n * 128); }
    {   size_t n = region / 32; hipLaunchKernelGGL(k_scatter8, grid(n), dim3(256), 0, 0, buf + r, n, sink); r += region;
        printf("k_scatter8 %zu %zu %zu %zu %zu\n", n, n * 8, region, region, region); }
    {   size_t n = region / 128; hipLaunchKernelGGL(k_scatter8_line, grid(n), dim3(256), 0, 0, buf + r, n, sink); r += region;
        printf("k_scatter8_line %zu %zu %zu %zu %zu\n", n, n * 8, n * 32, n * 64, n * 128); }
    CHECK(hipDeviceSynchronize());
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
