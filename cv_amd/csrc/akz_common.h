// akz_common.h — internal declarations shared by the HIP translation units of libakz.
// Nothing here is part of the ABI (that is include/akz.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "../../include/akz.h"

// ---- error plumbing -------------------------------------------------------------------------
extern thread_local int g_akz_last_hip;
#define AKZ_HIP(call)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            g_akz_last_hip = (int)e_;                                                            \
            return (e_ == hipErrorOutOfMemory) ? AKZ_E_OOM : AKZ_E_HIP;                          \
        }                                                                                        \
    } while (0)
#define AKZ_TRY(expr)                                                                            \
    do {                                                                                         \
        int32_t s_ = (expr);                                                                     \
        if (s_ != AKZ_OK) return s_;                                                             \
    } while (0)
#define AKZ_LAUNCH_CHECK() AKZ_HIP(hipGetLastError())

// `stream_to_wait` arguments of the ABI: NULL = nothing to wait for; AKZ_STREAM_LEGACY (include/akz.h, the value of
// hipStreamLegacy) = the legacy default stream, whose own handle is NULL too and could not be told from "none".
inline hipStream_t akz_wait_stream(void* h) { return h == (void*)1 ? (hipStream_t)nullptr : (hipStream_t)h; }

// Slots per frame in every per-(frame, level) table (candidate counts and lists, the keypoint kernels' level
// table).  A configuration whose pyramid has more levels is refused at akz_create (AKZ_E_INVALID).
constexpr int kAkzMaxLevels = 32;
// A stream restricted to compute units [first, first + count) of EVERY XCD (MI355X: 8 XCDs x 32 CUs; bit n of the mask is
// CU n / 8 of XCD n % 8 — the driver deals the mask's bits round-robin over the XCDs, so workgroup b still lands on XCD b % 8).
// hipExtStreamCreateWithCUMask takes neither flags nor a priority.
inline hipError_t akz_stream_on_cus(hipStream_t* s, int first, int count)
{
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = first; c < first + count && c < 32; ++c)
        for (int x = 0; x < 8; ++x) mask[(c * 8 + x) >> 5] |= 1u << ((c * 8 + x) & 31);
    return hipExtStreamCreateWithCUMask(s, 8, mask);
}
// Largest frame a context accepts, in pixels: the diffusion / determinant kernels address a frame's planes with 32-bit BYTE
// offsets (at_bytes in akz_scale_space.hip), the widest being the 8-byte {Lx, Ly} plane: pixels * 8 must stay below 2^32.
// 2^28 pixels (16384 x 16384) leaves a factor of two; akz_create_ex refuses more with AKZ_E_TOO_LARGE.
constexpr size_t kAkzMaxPixels = (size_t)1 << 28;
static_assert(kAkzMaxPixels * 8 <= ((size_t)1 << 32) / 2, "32-bit byte offsets into the {Lx, Ly} plane");
// Largest per-frame keypoint list / per-(frame, level) candidate list a context can be created for.
constexpr uint32_t kAkzMaxKeypoints = 262144u;   // (Akaze::dense() on 1080p noise: 82 000 keypoints in one frame)
// Keys a per-frame / per-level sort keeps in LDS (128 KB of the CU's 160 KB); longer lists sort through global memory.
constexpr uint32_t kAkzLdsSortKeys = 16384u;
// Longest Gaussian kernel of the generic blur path (base_scale_offset up to 255.5)
// scratch of the parallel suppression (akz_keypoints.hip: SupFrame), in 32-bit words
constexpr int kAkzSupDeg = 24;         // neighbours kept per candidate (each direction)
// chunk flags per frame (even, so that the uint2 array behind stays 8-byte aligned)
__host__ __device__ __forceinline__ uint32_t sup_done_words(uint32_t cap) { return ((cap + 1023u) / 1024u + 2u) & ~1u; }
// words of scratch per call of `nframes` frames; the first sup_zero_words() of them must be zero when k_sup_adj starts
__host__ __device__ __forceinline__ size_t sup_scratch_words(uint32_t cap, uint32_t nframes)
{
    return (size_t)nframes * (sup_done_words(cap) + (size_t)cap * (7 + 2 * kAkzSupDeg));
}
__host__ __device__ __forceinline__ size_t sup_zero_words(uint32_t cap, uint32_t nframes)
{
    return (size_t)nframes * (sup_done_words(cap) + 2 * (size_t)cap);
}
// calls of at most this many frames are treated as latency-bound chains of launches (longer FED blocks, ...)
constexpr int kLatencyFrames = 4;
constexpr int kAkzMaxTaps = 1023;

// ---- host-side plan: what Akaze::allocate_evolutions computes (akaze/src/evolution.rs:80-126) ---
struct AkzLevel {
    int w, h;
    uint32_t octave, sublevel;
    double esigma, etime;
    uint32_t deriv_sigma;       // round(esigma*derivative_factor/2^octave), detector_response.rs:11-13
    float sigma_quat;           // (deriv_sigma^4) as f32, detector_response.rs:38
    float kp_size;              // (esigma*derivative_factor) as f32, scale_space_extrema.rs:63
    std::vector<double> tau;    // fed_tau_steps (f64); each step uses tau as f32 (lib.rs:254)
    bool new_octave;            // octave > previous level's octave (lib.rs:219)
    float cand_border;          // smax * sigma_size of the border test (scale_space_extrema.rs:97-100)
    int cand_x_lo, cand_x_hi, cand_y_lo, cand_y_hi;   // candidates may sit at these pixels only (akz_plan.cpp)
    size_t pixels() const { return (size_t)w * (size_t)h; }
};
struct AkzPlan {
    int w = 0, h = 0;
    std::vector<AkzLevel> levels;
    int n_octaves = 0;
    size_t sum_pixels = 0;
};
// Host scalar math (no device work): evolution.rs:46-126 + fed_tau.rs:26-93.
void akz_build_plan(const akz_config& cfg, int w, int h, AkzPlan* plan);
// gaussian_kernel — image.rs:360-374 (host libm expf, as the reference).
void akz_host_gaussian_kernel(float r, int ksize, float* out);
// kernel radius of gaussian_blur(sigma) — image.rs:385.
int akz_gaussian_radius(float r);

// ---- small POD passed by value to kernels ---------------------------------------------------
struct GaussTaps {
    float k[12];  // up to 9 taps used on the AKAZE path (sigma 1.6 -> 9, sigma 1.0 -> 5)
    int n;
};

struct ScharrW {   // computer_scharr_kernel, derivatives.rs:57-79
    float norm;    // side weight
    float middle;  // centre weight
    int sigma;
};
ScharrW akz_scharr_weights(uint32_t sigma);

// C++ exceptions must not cross the C ABI: every int32_t entry point runs its body through this.
template <typename F>
static inline int32_t akz_guard(F&& f) noexcept
{
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return AKZ_E_OOM;
    } catch (...) {
        return AKZ_E_INTERNAL;
    }
}

static inline int akz_div_up(int a, int b) { return (a + b - 1) / b; }
static inline size_t akz_align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// Stable LSD radix sort of element ids by 32-bit keys, in LDS, for a 1024-thread block (16 waves): `rk[e]` is the key
// of element e, `ia` holds the ids 0..n-1 in their initial order and `ib` is scratch of the same size; four 8-bit
// digits (or only the `npass` lowest).  Returns the buffer that holds the sorted ids.  `wh` is [16][256] words, `tot`
// [256].  A wave owns a contiguous range of positions, so "wave-major, then position" is the list order: per pass the
// waves count their digits, a scan turns the counts into (digit, wave) offsets, and every wave scatters its range in
// order — 64 elements at a time, rank inside the step by bit-sliced ballots.  Equal keys keep their order.
constexpr uint32_t kRadixSortMax = 8192;   // elements per list the callers' LDS layouts hold (3 x 4 B each + 16 KB of counters)
constexpr size_t kRadixSortLdsBytes = sizeof(uint32_t) * (3 * (size_t)kRadixSortMax + 16 * 256);
__device__ __forceinline__ uint32_t* lds_radix_sort_ids(const uint32_t* rk, uint32_t* ia, uint32_t* ib, uint32_t* wh,
                                                        uint32_t* tot, uint32_t n, int npass)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t per = (n + 15u) / 16u, r0 = min(n, wv * per), r1 = min(n, r0 + per);   // this wave's range
    for (int pass = 0; pass < npass; ++pass) {
        const int sh = 8 * pass;
        for (uint32_t b = tid; b < 16 * 256; b += 1024) wh[b] = 0u;
        __syncthreads();
        for (uint32_t p0 = r0; p0 < r1; p0 += 64) {
            const uint32_t p = p0 + lane;
            if (p < r1) atomicAdd(&wh[wv * 256 + ((rk[ia[p]] >> sh) & 255u)], 1u);
        }
        __syncthreads();
        if (tid < 256) {
            uint32_t t = 0;
            for (int w = 0; w < 16; ++w) t += wh[w * 256 + tid];
            tot[tid] = t;
        }
        __syncthreads();
        if (tid < 256) {
            uint32_t off = 0;
            for (uint32_t b = 0; b < tid; ++b) off += tot[b];
            for (int w = 0; w < 16; ++w) {
                const uint32_t cnt = wh[w * 256 + tid];
                wh[w * 256 + tid] = off;
                off += cnt;
            }
        }
        __syncthreads();
        for (uint32_t p0 = r0; p0 < r1; p0 += 64) {
            const uint32_t p = p0 + lane;
            const bool on = p < r1;
            const uint32_t el = on ? ia[p] : 0u;
            const uint32_t dig = on ? ((rk[el] >> sh) & 255u) : 0u;
            unsigned long long same = __ballot(on);
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const unsigned long long bal = __ballot((dig >> bit) & 1u);
                same &= ((dig >> bit) & 1u) ? bal : ~bal;
            }
            if (on) {
                const uint32_t rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
                const uint32_t base = wh[wv * 256 + dig];
                ib[base + rank] = el;
                if (rank == 0) wh[wv * 256 + dig] = base + (uint32_t)__popcll(same);   // the step's first lane of the digit
            }
        }
        __syncthreads();
        uint32_t* t = ia;
        ia = ib;
        ib = t;
    }
    return ia;
}

// ---- device helpers ---------------------------------------------------------------------------
#if defined(__HIPCC__)
// Ascending bitonic sort of np2 (a power of two) 64-bit keys in LDS by one block of NT threads (NT a multiple of
// 64, all threads call it; the keys must be complete and a barrier passed before).  Work item t owns the pair
// (i, i | j) with i = t with a zero inserted at bit log2(j), so every thread compares one pair per item.  The
// elements a wave touches at strides j <= 64 form 128-element blocks that no other wave touches before the next
// stride >= 128, and a wave's LDS operations execute in order: those stages need no block barrier (28 of the 91
// stages of an 8192-key sort keep one).  Ends with a barrier.
template <int NT>
__device__ __forceinline__ void bitonic_sort_lds_u64(unsigned long long* key, uint32_t np2)
{
    const uint32_t half = np2 >> 1;
    for (uint32_t k2 = 2; k2 <= np2; k2 <<= 1) {
        for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < half; t += NT) {
                const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), ixj = i | j;
                const unsigned long long a = key[i], b = key[ixj];
                const bool up = (i & k2) == 0;
                if ((a > b) == up) {
                    key[i] = b;
                    key[ixj] = a;
                }
            }
            if (j > 64u || j == 1u) __syncthreads();          // next stride (k2 of the next phase) may cross waves
            else __builtin_amdgcn_wave_barrier();              // same wave owns the same 128-element blocks
        }
    }
    __syncthreads();
}
// Ascending sort of np2 (a power of two) 64-bit keys that live in GLOBAL memory, by one block of NT threads with an
// LDS buffer of lds_n keys (a power of two): lists of up to lds_n keys never leave LDS (bitonic_sort_lds_u64 on a
// copy); longer ones run the same bitonic network with the strides >= lds_n as passes over global memory and the
// strides below it chunk by chunk in LDS (one load / store of every chunk per merge level).  The rare path of the
// keypoint sorts: frames with more than 16384 keypoints or a level with more than 16384 extrema.
template <int NT>
__device__ __forceinline__ void bitonic_sort_big_u64(unsigned long long* g, uint32_t np2, unsigned long long* lds, uint32_t lds_n)
{
    const uint32_t tid = threadIdx.x;
    auto local_passes = [&](uint32_t gbase, uint32_t k2, uint32_t jstart) {   // passes j = jstart .. 1 of merge level k2
        for (uint32_t j = jstart; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < (lds_n >> 1); t += NT) {
                const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), ixj = i | j;
                const unsigned long long a = lds[i], b = lds[ixj];
                const bool up = ((gbase + i) & k2) == 0;
                if ((a > b) == up) {
                    lds[i] = b;
                    lds[ixj] = a;
                }
            }
            __syncthreads();
        }
    };
    if (np2 <= lds_n) {
        for (uint32_t i = tid; i < np2; i += NT) lds[i] = g[i];
        __syncthreads();
        bitonic_sort_lds_u64<NT>(lds, np2);
        for (uint32_t i = tid; i < np2; i += NT) g[i] = lds[i];
        __syncthreads();
        return;
    }
    // merge levels up to lds_n inside each chunk (direction by the chunk's global position)
    for (uint32_t base = 0; base < np2; base += lds_n) {
        for (uint32_t i = tid; i < lds_n; i += NT) lds[i] = g[base + i];
        __syncthreads();
        for (uint32_t k2 = 2; k2 <= lds_n; k2 <<= 1) local_passes(base, k2, k2 >> 1);
        for (uint32_t i = tid; i < lds_n; i += NT) g[base + i] = lds[i];
        __syncthreads();
    }
    for (uint32_t k2 = lds_n << 1; k2 <= np2; k2 <<= 1) {
        for (uint32_t j = k2 >> 1; j >= lds_n; j >>= 1) {                  // strides that cross chunks: global passes
            for (uint32_t t = tid; t < (np2 >> 1); t += NT) {
                const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), ixj = i | j;
                const unsigned long long a = g[i], b = g[ixj];
                const bool up = (i & k2) == 0;
                if ((a > b) == up) {
                    g[i] = b;
                    g[ixj] = a;
                }
            }
            __threadfence_block();
            __syncthreads();
        }
        for (uint32_t base = 0; base < np2; base += lds_n) {              // the remaining strides, chunk by chunk
            for (uint32_t i = tid; i < lds_n; i += NT) lds[i] = g[base + i];
            __syncthreads();
            local_passes(base, k2, lds_n >> 1);
            for (uint32_t i = tid; i < lds_n; i += NT) g[base + i] = lds[i];
            __threadfence_block();
            __syncthreads();
        }
    }
}
#endif

