// akz_scale_space.hip — gfx950 kernels for the nonlinear scale space and the Hessian response
// (SURVEY.md §8a rows A1..A11).  Compiled with -ffp-contract=off: every f32 expression below is
// evaluated in exactly the order the reference evaluates it, one rounding per operation.
//
// Reference functions implemented here (paths relative to the rust-cv/cv checkout):
//   GrayFloatImage::from_dynamic (Luma8 arm)      akaze/src/image.rs:47-56          k_blur_tile<.., uint8_t, ..>
//   horizontal_filter / vertical_filter           akaze/src/image.rs:202-331        lane4_dot(), k_filter1d
//   gaussian_blur                                 akaze/src/image.rs:383-389        k_blur_tile
//   GrayFloatImage::half_size                     akaze/src/image.rs:154-199        k_half_size
//   simple_scharr_horizontal / _vertical          akaze/src/derivatives.rs:3-11     k_blur_tile epilogues
//   compute_contrast_factor                       akaze/src/contrast_factor.rs:16-64  EPI_CMAX / EPI_CHIST / k_contrast_finish
//   pm_g2                                         akaze/src/nonlinear_diffusion.rs:70-83  EPI_FLOW
//   calculate_step                                akaze/src/nonlinear_diffusion.rs:14-58  k_fed_step*
//   scharr_horizontal / scharr_vertical           akaze/src/derivatives.rs:23-79    k_deriv_first / k_deriv_second
//   Akaze::detector_response                      akaze/src/detector_response.rs:33-57   k_deriv_second
//   Akaze::create_nonlinear_scale_space           akaze/src/lib.rs:193-258          akz_run_scale_space
//
// Layout: every pyramid buffer is a dense row-major f32 plane per frame, frames back to back
// (frame stride = level pixels), so consecutive lanes touch consecutive addresses of one row and
// blockIdx.z selects the frame: one launch covers the whole batch.
//
// Clamp-border rule used by every fused stage: an on-chip tile position holds the value of the
// image at the CLAMPED coordinate of that position, and a stage evaluated at a position outside
// the image is evaluated at its clamped coordinate.  That is exactly what the reference's
// edge-replicated scratch rows/columns produce stage by stage.
#include "akz_ctx.h"

// ---- the three arithmetic orders that live in un-vendored crates of the reference ([3P-unverified]) ----
// This translation unit is compiled once per combination (-DAKZ_ARITH=0..7, cv_amd/build.py); akz_options.arith selects
// the copy a context runs (akz_arith.hip).  AKZ_ARITH = 0 is what the crate sources imply for a default x86-64 build and
// what every measured number is quoted on; in that copy every `if constexpr` below folds to the code of rounds 1-3.
//   bit 0  AKZ_ARITH_REDUCE_PAIRWISE   wide::f32x4::reduce_add = (a0 + a1) + (a2 + a3)   [0: ((a0 + a1) + a2) + a3]
//   bit 1  AKZ_ARITH_FMA               wide::f32x4::mul_add is one fused operation        [0: multiply, then add]
//   bit 2  AKZ_ARITH_HALF_SEQUENTIAL   ndarray sum() of the 2 x 2 window = ((a + b) + c) + d   [0: (a + b) + (c + d)]
// (akaze/src/image.rs:242-247, :320-325 and :160-195; the oracle has the same switches: oracle/akaze_oracle.c ORC_OPT_*.)
#ifndef AKZ_ARITH
#define AKZ_ARITH 0
#endif
#define AKZ_SS_CAT2(a, b) a##b
#define AKZ_SS_CAT(a, b) AKZ_SS_CAT2(a, b)
#define AKZ_SS(name) AKZ_SS_CAT(name##_arith, AKZ_ARITH)

namespace {

constexpr bool kArithPairwise = (AKZ_ARITH & 1) != 0;
constexpr bool kArithFma = (AKZ_ARITH & 2) != 0;
constexpr bool kArithHalfSeq = (AKZ_ARITH & 4) != 0;

constexpr int kTW = 64;  // output tile width  (one wave wide: a wave reads/writes one contiguous row segment)
constexpr int kTH = 32;  // output tile height
constexpr int kFTH = 32; // row tile of the contrast passes: 32 rows x 512 threads, every thread a 4-pixel patch in the gradient phase
                         // (24 rows — three blocks of 53 KB per CU instead of two of 62 — left a quarter of the threads idle there:
                         // 1 543 vs 1 328 us per 256 frames once the fine histogram had joined the block's LDS)
constexpr int kDTH = 12; // output tile height of the two-frame determinant kernel
constexpr int kCTiles = 17; // row tiles per block of the contrast passes in a batch (17 x 32 rows: two blocks per 1080p column;
                           // 5 / 9 / 12 / 17 tiles: 1 452 / 1 377 / 1 340 / 1 328 us); a few-frame call takes kCTilesFew
constexpr int kCTilesFew = 4; // ... so that one frame still fills the chip (1080p: 30 x 9 blocks)
constexpr int kFNT = 512; // its block size: 2 blocks x 8 waves per CU

enum { EPI_BLUR = 0, EPI_FLOW = 1, EPI_CMAX = 2, EPI_CHIST = 3 };

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// XCD-aware tile order.  The dispatcher places workgroup b on XCD b % 8 (observed, not contractual: this is a
// speed choice only), and each XCD has its own L2.  Tiles that share halo rows / columns should therefore have
// ids that are congruent mod 8: the bijective remap below hands XCD x the x-th contiguous eighth of the tile
// sequence (x fastest, then y, then frame pair), so a tile's neighbours hit the L2 that already holds the halo.
__device__ __forceinline__ uint3 xcd_tile_linear(uint32_t orig, uint3 grid);
__device__ __forceinline__ uint3 xcd_tile(uint3 bid, uint3 grid)
{
    return xcd_tile_linear(bid.x + grid.x * (bid.y + grid.y * bid.z), grid);
}
// the same for a linear workgroup number (kernels that walk a virtual grid: the number's low three bits must be the XCD
// the workgroup runs on, i.e. the walk's stride a multiple of 8)
__device__ __forceinline__ uint3 xcd_tile_linear(uint32_t orig, uint3 grid)
{
    const uint32_t nwg = grid.x * grid.y * grid.z;
    const uint32_t xcd = orig & 7u, q = nwg >> 3, r = nwg & 7u;
    const uint32_t t = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + (orig >> 3);
    const uint32_t z = t / (grid.x * grid.y), rem = t - z * (grid.x * grid.y);
    const uint32_t y = rem / grid.x;
    return make_uint3(rem - y * grid.x, y, z);
}

// f32::from(v) / 255f32 and / 65535f32 (image.rs:54, :57-66) are true IEEE divisions per pixel.  For the integers
// 0..255 over 255 and 0..65535 over 65535 the quotient is reproduced bit for bit by one multiply by the rounded
// reciprocal and one Newton correction — q0 = v * r, q = fma(fma(-d, q0, v), r, q0) — checked exhaustively over both
// ranges (tests/test_abi.py: test_pixel_normalisation_shortcut_is_exact); the compiler's division is ~11 instructions.
__device__ __forceinline__ float px_over_255(float v)
{
    const float r = 1.0f / 255.0f, q0 = v * r;
    return __builtin_fmaf(__builtin_fmaf(-255.0f, q0, v), r, q0);
}
__device__ __forceinline__ float px_over_65535(float v)
{
    const float r = 1.0f / 65535.0f, q0 = v * r;
    return __builtin_fmaf(__builtin_fmaf(-65535.0f, q0, v), r, q0);
}
__device__ __forceinline__ float load_px(const float* p, size_t i) { return p[i]; }
// image.rs:54 — f32::from(v) / 255f32: a true IEEE division per pixel.
__device__ __forceinline__ float load_px(const uint8_t* p, size_t i) { return px_over_255((float)p[i]); }
// image.rs:57-66 — Luma16: f32::from(v) / 65535f32
__device__ __forceinline__ float load_px(const uint16_t* p, size_t i) { return px_over_65535((float)p[i]); }

// wide::f32x4 accumulate + reduce_add as used by horizontal_filter/vertical_filter
// (image.rs:242-247, :320-325): tap i goes to lane i&3, lanes accumulate in chunk order with an
// unfused multiply-then-add starting from +0, lanes are summed ((a0+a1)+a2)+a3.
// The fold starts every lane from +0.  Only lane 0's start is observable: x + (+0) differs from x for x = -0 alone,
// a0 = ... + (p0 + 0) can therefore never be -0, and a sum whose left operand is not -0 does not depend on the sign of
// a zero on its right — so lanes 1..3 start from their first product (three adds fewer per filter tap group, same bits).
// (AKZ_ARITH: a fused mul_add changes every accumulation after a lane's first product — fma(s, k, +0) is s * k up to the
// sign of a zero, which the argument above makes unobservable; a pairwise reduce_add changes the last two additions, and
// a0 + a1 is not -0 either, so the argument covers it as well.)
__device__ __forceinline__ float arith_mad(float s, float k, float acc)
{
    if constexpr (kArithFma) return __builtin_fmaf(s, k, acc);
    else return s * k + acc;
}
__device__ __forceinline__ float arith_reduce(float a0, float a1, float a2, float a3)
{
    if constexpr (kArithPairwise) return (a0 + a1) + (a2 + a3);
    else return ((a0 + a1) + a2) + a3;
}
template <int N>
__device__ __forceinline__ float lane4_dot(const float* s, int stride, const float* k)
{
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float v = s[i * stride];
        if ((i & 3) == 0) a0 = arith_mad(v, k[i], a0);
        else if ((i & 3) == 1) a1 = i == 1 ? v * k[i] : arith_mad(v, k[i], a1);
        else if ((i & 3) == 2) a2 = i == 2 ? v * k[i] : arith_mad(v, k[i], a2);
        else a3 = i == 3 ? v * k[i] : arith_mad(v, k[i], a3);
    }
    return arith_reduce(a0, a1, a2, a3);
}

// The multiscale Scharr "off" kernel [n, 0 .. 0, m, 0 .. 0, n] (taps 0, sigma, 2 sigma of 2 sigma + 1) through the
// four-lane accumulation: tap i goes to lane i & 3, so with a = v(-s), b = v(0), c = v(+s) and pa = a n (+0: lane 0's
// start), pb = b m, pc = c n the result is one of
//   mode 0  (pa + pc) + pb      a and c share lane 0 (sigma % 4 == 2), or lanes 0, 3, 2 summed in lane order (sigma % 4 == 3)
//   mode 1  (pa + pb) + pc      all three in lane 0 (sigma % 4 == 0), or lanes 0, 1, 2 (sigma % 4 == 1; the simple Scharr [3, 10, 3])
//   mode 2  pa + (pc + pb)      lanes 0, 3, 2 under a pairwise reduce_add: (l0 + l1) + (l2 + l3)                [AKZ_ARITH bit 0]
//   mode 3  fma(c, n, pa) + pb            a and c share lane 0, fused mul_add                                      [AKZ_ARITH bit 1]
//   mode 4  fma(c, n, fma(b, m, pa))      all three in lane 0, fused mul_add                                       [AKZ_ARITH bit 1]
// (a lane that holds a single product is the same fused or not; every other lane is +0.)
struct OffK {
    float n, m;
    int mode;
};
constexpr int offk_mode(uint32_t sigma)
{
    if (sigma == 1) return 1;
    const int lb = (int)(sigma & 3u), lc = (int)((2u * sigma) & 3u);
    if (lc == 0 && lb != 0) return kArithFma ? 3 : 0;
    if (lb == 0) return kArithFma ? 4 : 1;
    if (lb < lc) return 1;
    return kArithPairwise ? 2 : 0;
}
__device__ __forceinline__ float off_combine(const OffK k, float a, float b, float c)
{
    // the first tap of lane 0 is accumulated onto the +0 the reference's fold starts from (a product that
    // underflows to -0 becomes +0); after that no partial sum can be -0, so no other tap needs it
    const float pa = a * k.n + 0.0f;
    if constexpr (!kArithFma && !kArithPairwise) {
        const float pb = b * k.m, pc = c * k.n;
        return k.mode == 0 ? (pa + pc) + pb : (pa + pb) + pc;
    } else {
        switch (k.mode) {
        case 0: return (pa + c * k.n) + b * k.m;
        case 1: return (pa + b * k.m) + c * k.n;
        case 2: return pa + (c * k.n + b * k.m);
        case 3: return __builtin_fmaf(c, k.n, pa) + b * k.m;
        default: return __builtin_fmaf(c, k.n, __builtin_fmaf(b, k.m, pa));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Two-frame packed arithmetic.  The reference's filters are unfused multiply-then-add chains, so the VALU
// count is what bounds the fused tile kernels (rocprof: 88 % VALU-busy).  gfx950 issues v_pk_mul_f32 /
// v_pk_add_f32 at twice the element rate of the scalar forms, and they round each half exactly like the
// scalar instruction, so a block processes the SAME tile of two consecutive frames with every on-chip
// value stored as a {frame a, frame b} pair: same operation order per frame, half the instructions.
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f splat(float a) { return (v2f){a, a}; }

// Element `byte_off` bytes behind a frame's base pointer.  With the base uniform (a frame index from blockIdx) and the
// offset 32 bits wide the access compiles to `global_load/store v, v_off, s[base]` — without it every access builds its own
// 64-bit address (a shift and two 64-bit adds: three to four VALU instructions per access, 4-5 % of the diffusion kernels).
// A context never holds a frame above kAkzMaxPixels (akz_common.h; akz_create_ex refuses it): pixels * 8 bytes < 2^31.
template <typename T>
__device__ __forceinline__ T* at_bytes(T* base, uint32_t byte_off)
{
    return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off);
}
template <typename T>
__device__ __forceinline__ const T* at_bytes(const T* base, uint32_t byte_off)
{
    return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}

// 1.0f / d for both halves, correctly rounded, for d that is finite and >= 1 (pm_g2's denominator 1 + k (lx^2 + ly^2),
// nonlinear_diffusion.rs:80): the compiler's IEEE f32 division is v_div_scale x2, v_rcp, a chain of six FMAs,
// v_div_fmas and v_div_fixup per lane half — 22 scalar instructions for a {frame a, frame b} pair.  In that range
// the scaling and the fix-up are identities, so the same chain on the raw operands gives the same bits: two v_rcp_f32
// and six v_pk_fma_f32.  Callers route anything else (NaN, infinity) to the plain division.
__device__ __forceinline__ v2f rcp_pair_finite(v2f d)
{
    const v2f one = splat(1.0f);
    const v2f r0 = (v2f){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const v2f e0 = __builtin_elementwise_fma(-d, r0, one);
    const v2f r1 = __builtin_elementwise_fma(e0, r0, r0);
    const v2f q0 = r1;                                   // numerator 1
    const v2f e1 = __builtin_elementwise_fma(-d, q0, one);
    const v2f q1 = __builtin_elementwise_fma(e1, r1, q0);
    const v2f e2 = __builtin_elementwise_fma(-d, q1, one);
    return __builtin_elementwise_fma(e2, r1, q1);
}
// v_cmp_class_f32: signalling NaN 0x1, quiet NaN 0x2, -infinity 0x4, +infinity 0x200
__device__ __forceinline__ bool not_finite(float v) { return __builtin_amdgcn_classf(v, 0x207); }

// On-chip tile layout for the two-frame kernels.  A row of C columns (C % 4 == 0) is stored as two planes:
// plane 0 holds the column pairs {4c, 4c+1}, plane 1 the pairs {4c+2, 4c+3}, each pair being 16 bytes
// ({col, frame a}, {col, frame b}, {col+1, a}, {col+1, b}).  A thread works on a 4-column strip c and reads
// it as 16-byte chunks plane0[c + j], plane1[c + j]: consecutive lanes touch consecutive 16-byte chunks, so
// a ds_read_b128 covers all 64 banks once (with the strip stored contiguously, lanes L and L+8 of a
// 16-lane group collide and the read takes twice the LDS cycles).
template <int C>
__device__ __forceinline__ int tile_col(int col) { return ((col & 2) ? C / 2 : 0) + ((col >> 2) << 1) + (col & 1); }

// columns [4c + LO, 4c + HI] of 12 consecutive columns starting at strip c; v[j] = column 4c + j
typedef float f4v __attribute__((ext_vector_type(4)));
// A 16-byte LDS read.  (Forcing the full width with an empty asm when a caller uses only one 8-byte half was
// measured slower: 231 vs 211 us on the full-resolution front kernel; the compiler's narrowed reads stay.)
__device__ __forceinline__ f4v lds_chunk(const float4* p) { return *reinterpret_cast<const f4v*>(p); }

template <int C, int LO, int HI>
__device__ __forceinline__ void lds_read12(const float4* __restrict__ tile, int row, int c, v2f (&v)[12])
{
    const float4* p0 = tile + (row * (C / 2) + c);       // C / 2 chunks of 16 bytes per row
    const float4* p1 = p0 + C / 4;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (4 * j + 1 >= LO && 4 * j <= HI) {
            f4v t = lds_chunk(p0 + j);
            v[4 * j] = (v2f){t.x, t.y};
            v[4 * j + 1] = (v2f){t.z, t.w};
        }
        if (4 * j + 3 >= LO && 4 * j + 2 <= HI) {
            f4v t = lds_chunk(p1 + j);
            v[4 * j + 2] = (v2f){t.x, t.y};
            v[4 * j + 3] = (v2f){t.z, t.w};
        }
    }
}
// Work item -> (row q, 4-column strip c) of a pass over `rows` tile rows of NS strips (16 <= NS <= 20).
// A ds_read_b128 is served in four 16-lane groups drawn from one 32-lane half each, one LDS cycle per group when
// the group's 16 chunks fall on 16 different 16-byte slots of the 256-byte bank row (MI355X_MICROARCH.md, LDS).
// Items in plain raster order put a row break inside most groups and the two rows then overlap on some slots
// (2-way, the reads run at half rate).  Here lanes 16k .. 16k+15 take strips 0..15 of ONE row, two rows per
// 32-lane half, and the odd row's strips are rotated by ROT = (row stride in chunks) mod 16, so the 32 lanes of
// a half read 32 chunks whose slots are lane-consecutive; the NS - 16 strips left over per row go to the items
// after rows * 16 (a few lanes with conflicts instead of every wave).  A pure permutation of the work items.
// Used by the determinant kernel (bank-conflict cycles 21-27 % -> 6-14 %, 1.5 % faster); the blur / front-end
// passes measured no different with it (they wait on barriers and loads, not on the LDS) and keep raster order.
template <int NS, int ROT>
__device__ __forceinline__ void strip_item(int idx, int rows, int& q, int& c)
{
    const int main_items = rows * 16;
    if (idx < main_items) {
        q = idx >> 4;
        c = ((idx & 15) - ROT * (q & 1)) & 15;
    } else {
        const int j = idx - main_items;
        q = j / (NS > 16 ? NS - 16 : 1);
        c = 16 + (j - q * (NS > 16 ? NS - 16 : 1));
    }
}

template <int C>
__device__ __forceinline__ void lds_write4(v2f* __restrict__ row, int c, v2f a, v2f b, v2f cc, v2f d)
{
    reinterpret_cast<float4*>(row)[c] = make_float4(a.x, a.y, b.x, b.y);
    reinterpret_cast<float4*>(row + C / 2)[c] = make_float4(cc.x, cc.y, d.x, d.y);
}

// lane4_dot on register operands (see lane4_dot): taps v[0..N-1]
__device__ __forceinline__ v2f arith_mad_v(v2f s, float k, v2f acc)
{
    if constexpr (kArithFma) return __builtin_elementwise_fma(s, splat(k), acc);
    else return s * splat(k) + acc;
}
template <int N>
__device__ __forceinline__ v2f lane4_dot_v(const v2f* v, const float* k)
{
    v2f a0 = splat(0.0f), a1 = splat(0.0f), a2 = splat(0.0f), a3 = splat(0.0f);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if ((i & 3) == 0) a0 = arith_mad_v(v[i], k[i], a0);
        else if ((i & 3) == 1) a1 = i == 1 ? v[i] * splat(k[i]) : arith_mad_v(v[i], k[i], a1);   // (lanes 1..3 start from their first product: see lane4_dot)
        else if ((i & 3) == 2) a2 = i == 2 ? v[i] * splat(k[i]) : arith_mad_v(v[i], k[i], a2);
        else a3 = i == 3 ? v[i] * splat(k[i]) : arith_mad_v(v[i], k[i], a3);
    }
    if constexpr (kArithPairwise) return (a0 + a1) + (a2 + a3);
    else return ((a0 + a1) + a2) + a3;
}

// off_combine with the accumulation order fixed at compile time (offk_mode(SG)): the derivative scales the two-frame
// kernels serve are 2, 3 and 4.
template <int SG>
__device__ __forceinline__ v2f off_combine_sg(const OffK k, v2f a, v2f b, v2f c)
{
    constexpr int mode = offk_mode(SG);
    const v2f pa = a * splat(k.n) + splat(0.0f);
    if constexpr (mode == 0) return (pa + c * splat(k.n)) + b * splat(k.m);
    else if constexpr (mode == 1) return (pa + b * splat(k.m)) + c * splat(k.n);
    else if constexpr (mode == 2) return pa + (c * splat(k.n) + b * splat(k.m));
    else if constexpr (mode == 3) return __builtin_elementwise_fma(c, splat(k.n), pa) + b * splat(k.m);
    else return __builtin_elementwise_fma(c, splat(k.n), __builtin_elementwise_fma(b, splat(k.m), pa));
}

__device__ __forceinline__ float4 load4_px(const float* p, size_t i) { return *reinterpret_cast<const float4*>(p + i); }
__device__ __forceinline__ float4 load4_px(const uint8_t* p, size_t i)
{
    uint32_t u = *reinterpret_cast<const uint32_t*>(p + i);
    return make_float4(px_over_255((float)(u & 0xFFu)), px_over_255((float)((u >> 8) & 0xFFu)),
                       px_over_255((float)((u >> 16) & 0xFFu)), px_over_255((float)(u >> 24)));
}

// k_level_front for two frames per block (requires w % 4 == 0 so that every 4-column strip is 16-byte
// aligned in HBM and either fully inside or fully outside the image).  Column bookkeeping: s_in column ix
// is image x = tx0 - 8 + ix (80 columns), s_h / s_g column p is image x = tx0 - 4 + p (72 columns), so an
// output strip, the blur windows and the +-SG derivative taps all start on 4-column boundaries.
// Every thread computes strips of 4 consecutive columns from registers: no per-pixel index arithmetic, one
// LDS read per ~1.3 multiply-adds instead of one per multiply.  The H and V passes run with plain
// (unclamped) windows over positions that hold values at clamped coordinates; positions of the blurred tile
// that lie outside the image are then overwritten with the value at their clamped coordinate (border tiles
// only), which is what the reference's edge replication produces stage by stage.
// Raw (unconverted) 4-pixel loads, so a prefetched u8 tile costs one VGPR per item and frame
__device__ __forceinline__ float4 load4_raw(const float* p, size_t i) { return *reinterpret_cast<const float4*>(p + i); }
__device__ __forceinline__ uint32_t load4_raw(const uint8_t* p, size_t i) { return *reinterpret_cast<const uint32_t*>(p + i); }
__device__ __forceinline__ uint2 load4_raw(const uint16_t* p, size_t i) { return *reinterpret_cast<const uint2*>(p + i); }
__device__ __forceinline__ float4 raw_px(float4 r) { return r; }
__device__ __forceinline__ float4 raw_px(uint2 u)     // image.rs:57-66 — f32::from(v) / 65535f32
{
    return make_float4(px_over_65535((float)(u.x & 0xFFFFu)), px_over_65535((float)(u.x >> 16)),
                       px_over_65535((float)(u.y & 0xFFFFu)), px_over_65535((float)(u.y >> 16)));
}
__device__ __forceinline__ float4 raw_px(uint32_t u)   // image.rs:54 — f32::from(v) / 255f32
{
    return make_float4(px_over_255((float)(u & 0xFFu)), px_over_255((float)((u >> 8) & 0xFFu)),
                       px_over_255((float)((u >> 16) & 0xFFu)), px_over_255((float)(u >> 24)));
}
template <typename InT> struct RawOf { typedef float4 type; };
template <> struct RawOf<uint8_t> { typedef uint32_t type; };
template <> struct RawOf<uint16_t> { typedef uint2 type; };

// Register-held input tile of the NEXT row tile of a block (interior tile columns only): fetched right after
// the current tile went to LDS, so the HBM/L2 latency of the loads hides behind the current tile's passes.
template <int R, int SG, int TH, int NT, typename InT>
struct PairTileRegs {
    static constexpr int CI = kTW + 16, IH = TH + 2 * SG + 2 * R;
    static constexpr int ITEMS = (IH * (CI / 4) + NT - 1) / NT;
    typename RawOf<InT>::type ra[ITEMS], rb[ITEMS];
    __device__ __forceinline__ void fetch(const InT* __restrict__ in, int w, int h, size_t fs, int fa, int fb, int tx0,
                                          int ty0)
    {
        const InT* srca = in + (size_t)fa * fs;
        const InT* srcb = in + (size_t)fb * fs;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            int idx = (int)threadIdx.x + i * NT;
            if (idx < IH * (CI / 4)) {
                int iy = idx / (CI / 4), c4 = idx - iy * (CI / 4);
                int cy = clampi(ty0 - SG - R + iy, 0, h - 1);
                size_t o = (size_t)cy * w + (tx0 - 8 + 4 * c4);
                ra[i] = load4_raw(srca, o);
                rb[i] = load4_raw(srcb, o);
            }
        }
    }
    __device__ __forceinline__ void commit(v2f* __restrict__ s_in) const
    {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            int idx = (int)threadIdx.x + i * NT;
            if (idx < IH * (CI / 4)) {
                int iy = idx / (CI / 4), c4 = idx - iy * (CI / 4);
                float4 a = raw_px(ra[i]), b = raw_px(rb[i]);
                lds_write4<CI>(&s_in[iy * CI], c4, (v2f){a.x, b.x}, (v2f){a.y, b.y}, (v2f){a.z, b.z}, (v2f){a.w, b.w});
            }
        }
    }
};

// Blurred two-frame tile: loads the input tile of frames fa / fb, runs the separable blur and leaves the
// blurred tile (GH x CG, split-plane layout) in s_a; s_h is scratch.  Shared by the level front-end and the
// contrast-factor passes.
template <int R, int SG, int TH, int NT, typename InT>
__device__ __forceinline__ void pair_blur_tile(const InT* __restrict__ in, int w, int h, size_t fs, int fa, int fb,
                                               int tx0, int ty0, int next_ty0, const GaussTaps& taps,
                                               PairTileRegs<R, SG, TH, NT, InT>& regs, v2f* __restrict__ s_a,
                                               v2f* __restrict__ s_h)
{
    constexpr int N = 2 * R + 1;
    constexpr int CI = kTW + 16, CG = kTW + 8;
    constexpr int GH = TH + 2 * SG, IH = GH + 2 * R;
    v2f* s_in = s_a;
    v2f* s_g = s_a;
    const int tid = threadIdx.x;
    const bool x_inside = tx0 >= 8 && tx0 + kTW + 8 <= w;   // same for every row tile of the block
    if (x_inside) {
        if (next_ty0 == -2) regs.fetch(in, w, h, fs, fa, fb, tx0, ty0);   // no prefetch: load this tile now
        regs.commit(s_in);                                  // fetched by the caller / the previous tile
        if (next_ty0 >= 0) regs.fetch(in, w, h, fs, fa, fb, tx0, next_ty0);
    } else {
        // tile columns on the image border: per-element clamped loads, eight per thread in flight (one load per round
        // trip made these tiles several times as slow as the others; on the small octaves they are a quarter to half of all tiles)
        const InT* srca = in + (size_t)fa * fs;
        const InT* srcb = in + (size_t)fb * fs;
        constexpr int BATCH = 8;
        for (int base = 0; base < IH * CI; base += NT * BATCH) {
            InT ea[BATCH], eb[BATCH];
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int idx = base + tid + NT * i;
                ea[i] = eb[i] = InT(0);
                if (idx < IH * CI) {
                    int iy = idx / CI, ix = idx - iy * CI;
                    int cx = clampi(tx0 - 8 + ix, 0, w - 1);
                    int cy = clampi(ty0 - SG - R + iy, 0, h - 1);
                    size_t o = (size_t)cy * w + cx;
                    ea[i] = srca[o];
                    eb[i] = srcb[o];
                }
            }
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int idx = base + tid + NT * i;
                if (idx < IH * CI) {
                    int iy = idx / CI, ix = idx - iy * CI;
                    s_in[iy * CI + tile_col<CI>(ix)] = (v2f){load_px(&ea[i], 0), load_px(&eb[i], 0)};
                }
            }
        }
    }
    __syncthreads();
    // horizontal pass: s_h[iy][p] = sum_i s_in[iy][p + 4 - R + i] * k[i]
    for (int idx = tid; idx < IH * (CG / 4); idx += NT) {
        int iy = idx / (CG / 4), c = idx - iy * (CG / 4);
        v2f v[12];
        lds_read12<CI, 4 - R, 7 - R + N - 1>(reinterpret_cast<const float4*>(s_in), iy, c, v);
        lds_write4<CG>(&s_h[iy * CG], c, lane4_dot_v<N>(v + 4 - R, taps.k), lane4_dot_v<N>(v + 5 - R, taps.k),
                       lane4_dot_v<N>(v + 6 - R, taps.k), lane4_dot_v<N>(v + 7 - R, taps.k));
    }
    __syncthreads();  // s_in is dead from here on: s_g overwrites it
    // vertical pass: s_g[q][p] = sum_i s_h[q + i][p] * k[i]
    for (int idx = tid; idx < GH * (CG / 4); idx += NT) {
        int q = idx / (CG / 4), c = idx - q * (CG / 4);
        v2f col[4][N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float4* rowp = reinterpret_cast<const float4*>(s_h) + ((q + i) * (CG / 2) + c);
            f4v t0 = lds_chunk(rowp);
            f4v t1 = lds_chunk(rowp + CG / 4);
            col[0][i] = (v2f){t0.x, t0.y};
            col[1][i] = (v2f){t0.z, t0.w};
            col[2][i] = (v2f){t1.x, t1.y};
            col[3][i] = (v2f){t1.z, t1.w};
        }
        lds_write4<CG>(&s_g[q * CG], c, lane4_dot_v<N>(col[0], taps.k), lane4_dot_v<N>(col[1], taps.k),
                       lane4_dot_v<N>(col[2], taps.k), lane4_dot_v<N>(col[3], taps.k));
    }
    __syncthreads();
    // positions outside the image take the value at their clamped coordinate (reads touch in-image
    // positions only, writes out-of-image positions only)
    if (tx0 < SG || tx0 + kTW + SG > w || ty0 < SG || ty0 + TH + SG > h) {
        for (int idx = tid; idx < GH * CG; idx += NT) {
            int q = idx / CG, p = idx - q * CG;
            int x = tx0 - 4 + p, y = ty0 - SG + q;
            int xc = clampi(x, 0, w - 1), yc = clampi(y, 0, h - 1);
            if (xc != x || yc != y)
                s_g[q * CG + tile_col<CG>(p)] = s_g[(yc - (ty0 - SG)) * CG + tile_col<CG>(xc - (tx0 - 4))];
        }
        __syncthreads();
    }
}

template <int R, int SG, int TH, int NT, typename InT, bool FLOW, int TPB>
__global__ __launch_bounds__(NT, NT == 512 ? 4 : 3) void k_level_front2(const InT* __restrict__ in, int w, int h, size_t fs, int n,
                                                       GaussTaps taps, OffK k, float* __restrict__ out_g,
                                                       float* __restrict__ out_flow, float2* __restrict__ out_xy,
                                                       const float* __restrict__ invk, int invk_off)
{
    constexpr int CI = kTW + 16, CG = kTW + 8;
    constexpr int GH = TH + 2 * SG, IH = GH + 2 * R;
    __shared__ __attribute__((aligned(16))) v2f s_a[IH * CI];   // input tile, later the blurred tile (GH x CG)
    __shared__ __attribute__((aligned(16))) v2f s_h[IH * CG];
    v2f* s_g = s_a;
    const uint3 tile = xcd_tile(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z));
    const int fa = 2 * (int)tile.z;
    const bool has_b = fa + 1 < n;
    const int fb = has_b ? fa + 1 : fa;
    const int tx0 = (int)tile.x * kTW;
    const int tid = threadIdx.x;
    v2f inverse_k = splat(0.0f);
    if (FLOW) inverse_k = (v2f){invk[(size_t)fa * 8 + invk_off], invk[(size_t)fb * 8 + invk_off]};
    // TPB vertically adjacent tiles per block, the next tile's input prefetched into registers
    PairTileRegs<R, SG, TH, NT, InT> regs;
    if (TPB > 1 && tx0 >= 8 && tx0 + kTW + 8 <= w) regs.fetch(in, w, h, fs, fa, fb, tx0, (int)tile.y * TPB * TH);
    for (int it = 0; it < TPB; ++it) {
    const int ty0 = ((int)tile.y * TPB + it) * TH;
    if (ty0 >= h) break;
    if (it) __syncthreads();                     // the previous tile's readers are done with s_a
    const int ty1 = TPB == 1 ? -2 : (it + 1 < TPB && ty0 + TH < h) ? ty0 + TH : -1;
    pair_blur_tile<R, SG, TH, NT, InT>(in, w, h, fs, fa, fb, tx0, ty0, ty1, taps, regs, s_a, s_h);
    for (int idx = tid; idx < TH * (kTW / 4); idx += NT) {
        const int q = idx / (kTW / 4), c = idx - q * (kTW / 4);
        const int x0 = tx0 + 4 * c, y = ty0 + q;
        if (x0 >= w || y >= h) continue;
        const float4* g4 = reinterpret_cast<const float4*>(s_g);
        const int r0 = q + SG;                  // v[4 + o] is output pixel o of the strip
        v2f z[12], res_x[4], res_y[4], res_f[4];
        lds_read12<CG, (SG > 1 ? 4 - SG : 3), (SG > 1 ? 7 + SG : 8)>(g4, r0, c, z);
        if (FLOW) {
            // simple Scharr (derivatives.rs:3-11) + pm_g2 (nonlinear_diffusion.rs:80)
            v2f m[12], pz[12];
            lds_read12<CG, 3, 8>(g4, r0 - 1, c, m);
            lds_read12<CG, 3, 8>(g4, r0 + 1, c, pz);
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                v2f hx_m = m[5 + o] - m[3 + o];
                v2f hx_0 = z[5 + o] - z[3 + o];
                v2f hx_p = pz[5 + o] - pz[3 + o];
                v2f lx = (splat(3.0f) * hx_m + splat(10.0f) * hx_0) + splat(3.0f) * hx_p;
                v2f hy_m = (splat(3.0f) * m[3 + o] + splat(10.0f) * m[4 + o]) + splat(3.0f) * m[5 + o];
                v2f hy_p = (splat(3.0f) * pz[3 + o] + splat(10.0f) * pz[4 + o]) + splat(3.0f) * pz[5 + o];
                v2f ly = hy_p - hy_m;
                res_f[o] = splat(1.0f) + inverse_k * (lx * lx + ly * ly);   // the denominator; inverted below
            }
            // a wave whose denominators are all finite (always, unless a frame has no gradient at all: then
            // inverse_k is infinite and the reference's quotient is NaN) takes the packed reciprocal
            // (one test on the sum of the four: it is not finite whenever one of them is not; a sum of finite values that
            // overflows only sends the wave through the plain division, which is exact for every input)
            const v2f dsum = (res_f[0] + res_f[1]) + (res_f[2] + res_f[3]);
            const bool odd = not_finite(dsum.x) || not_finite(dsum.y);
            if (__any(odd)) {
#pragma unroll
                for (int o = 0; o < 4; ++o) res_f[o] = splat(1.0f) / res_f[o];
            } else {
#pragma unroll
                for (int o = 0; o < 4; ++o) res_f[o] = rcp_pair_finite(res_f[o]);
            }
        }
        {
            // multiscale Scharr first derivatives (derivatives.rs:23-49), taps at -SG, 0, +SG
            v2f m[12], pz[12];
            lds_read12<CG, 4 - SG, 7 + SG>(g4, r0 - SG, c, m);
            lds_read12<CG, 4 - SG, 7 + SG>(g4, r0 + SG, c, pz);
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                v2f mm = m[4 + o - SG], m0 = m[4 + o], mp = m[4 + o + SG];
                v2f zm = z[4 + o - SG], zp = z[4 + o + SG];
                v2f pm = pz[4 + o - SG], p0 = pz[4 + o], pp = pz[4 + o + SG];
                res_x[o] = off_combine_sg<SG>(k, mp - mm, zp - zm, pp - pm);
                res_y[o] = off_combine_sg<SG>(k, pm, p0, pp) - off_combine_sg<SG>(k, mm, m0, mp);
            }
        }
        const size_t pix = (size_t)y * w + x0;
        {
            const size_t o = (size_t)fa * fs + pix;
            if (out_g) *reinterpret_cast<float4*>(out_g + o) = make_float4(z[4].x, z[5].x, z[6].x, z[7].x);
            if (FLOW)
                *reinterpret_cast<float4*>(out_flow + o) = make_float4(res_f[0].x, res_f[1].x, res_f[2].x, res_f[3].x);
            float4* xy = reinterpret_cast<float4*>(out_xy + o);
            xy[0] = make_float4(res_x[0].x, res_y[0].x, res_x[1].x, res_y[1].x);
            xy[1] = make_float4(res_x[2].x, res_y[2].x, res_x[3].x, res_y[3].x);
        }
        if (has_b) {
            const size_t o = (size_t)fb * fs + pix;
            if (out_g) *reinterpret_cast<float4*>(out_g + o) = make_float4(z[4].y, z[5].y, z[6].y, z[7].y);
            if (FLOW)
                *reinterpret_cast<float4*>(out_flow + o) = make_float4(res_f[0].y, res_f[1].y, res_f[2].y, res_f[3].y);
            float4* xy = reinterpret_cast<float4*>(out_xy + o);
            xy[0] = make_float4(res_x[0].y, res_y[0].y, res_x[1].y, res_y[1].y);
            xy[1] = make_float4(res_x[2].y, res_y[2].y, res_x[3].y, res_y[3].y);
        }
    }
    }
}

// ---- contrast factor without the histogram pass -------------------------------------------------------------
// contrast_factor.rs:41-64 walks the histogram until the running count reaches threshold = floor(npoints *
// percentile): the bin it stops after is the bin of the threshold-th smallest non-zero magnitude, so the result is
// an ORDER STATISTIC of v = lx^2 + ly^2, k = bin(v_(threshold)) + 1.  The max pass therefore also files every
// non-zero v by its f64 exponent and top 6 mantissa bits (kFineBins keys over [2^-22, 2^10), clamped outside);
// k_contrast_resolve finds the key that holds rank `threshold`, and when the reference bin of the key's smallest
// and largest possible value agree (the bin is monotone in v) that bin is the answer and the exact histogram
// pass is skipped for the frame; otherwise the frame is flagged and k_contrast_pair<EPI_CHIST> +
// k_contrast_finish settle it as before.
constexpr int kFineBins = 2048;
constexpr int kFineBase = (1023 - 22) << 6;
__device__ __forceinline__ int fine_key(double v)
{
    const int k = (__double2hiint(v) >> 14) - kFineBase;
    return k < 0 ? 0 : (k > kFineBins - 1 ? kFineBins - 1 : k);
}
// smallest value of key j (j >= 1) as a bit pattern
__device__ __forceinline__ unsigned long long fine_lo_bits(int j) { return (unsigned long long)(unsigned)((j + kFineBase) << 14) << 32; }

__device__ __forceinline__ void contrast_write(int f, double hmax, unsigned long long k, bool reached, int nbins,
                                               int n_octaves, double* __restrict__ contrast, float* __restrict__ invk)
{
    double cf = reached ? hmax * (double)k / (double)nbins : 0.03;
    contrast[f] = cf;
    for (int o = 0; o < 8; ++o) {
        if (o > 0) cf *= 0.75;
        invk[(size_t)f * 8 + o] = (o < n_octaves) ? (float)(1.0 / (cf * cf)) : 0.0f;
    }
}

__global__ __launch_bounds__(256) void k_contrast_resolve(const unsigned long long* __restrict__ cmax,
                                                          const uint32_t* __restrict__ fine,
                                                          const uint32_t* __restrict__ npoints,
                                                          const double* __restrict__ thr, int nbins, double percentile,
                                                          int n_octaves, double* __restrict__ contrast,
                                                          float* __restrict__ invk, uint32_t* __restrict__ flag,
                                                          int force_odd)
{
    __shared__ uint32_t s_sum[256];
    __shared__ int s_key;
    const int f = blockIdx.x, tid = threadIdx.x;
    if (force_odd && (f & 1)) {                    // test knob: odd frames always take the exact pass
        if (tid == 0) flag[f] = 1u;
        return;
    }
    const double hmax = sqrt(__longlong_as_double((long long)cmax[f]));
    const double t = (double)npoints[f] * percentile;
    const unsigned long long threshold = t > 0.0 ? (unsigned long long)t : 0ull;
    if (threshold == 0ull) {                       // the reference's loop does not run: k = 0
        if (tid == 0) {
            contrast_write(f, hmax, 0ull, true, nbins, n_octaves, contrast, invk);
            flag[f] = 0u;
        }
        return;
    }
    // key that holds rank `threshold`: 8 keys per thread, block scan of the partial sums
    const uint32_t* F = fine + (size_t)f * kFineBins;
    uint32_t loc[8], part = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        loc[i] = F[tid * 8 + i];
        part += loc[i];
    }
    s_sum[tid] = part;
    if (tid == 0) s_key = -1;
    __syncthreads();
    unsigned long long before = 0;
    for (int i = 0; i < tid; ++i) before += s_sum[i];
    if (before < threshold && before + part >= threshold) {   // exactly one thread
        unsigned long long run = before;
        int key = tid * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (run < threshold && run + loc[i] >= threshold) key = tid * 8 + i;
            run += loc[i];
        }
        s_key = key;
    }
    __syncthreads();
    if (tid != 0) return;
    const int key = s_key;
    if (key < 0) {                                  // counts and points disagree: let the exact pass decide
        flag[f] = 1u;
        return;
    }
    const double* T = thr + (size_t)f * 512;
    auto bin_of = [&](double x) {                   // max k in [0, nbins-1] with T[k] <= x (T ascending, T[0] = 0)
        int lo = 0, hi = nbins - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (T[mid] <= x) lo = mid;
            else hi = mid - 1;
        }
        return lo;
    };
    const double vlo = key == 0 ? __longlong_as_double(1ll) : __longlong_as_double((long long)fine_lo_bits(key));
    const double vhi = key == kFineBins - 1 ? __longlong_as_double(0x7FEFFFFFFFFFFFFFll)
                                            : __longlong_as_double((long long)(fine_lo_bits(key + 1) - 1ull));
    const int b0 = bin_of(vlo), b1 = bin_of(vhi);
    if (b0 == b1) {
        contrast_write(f, hmax, (unsigned long long)(b0 + 1), true, nbins, n_octaves, contrast, invk);
        flag[f] = 0u;
    } else {
        flag[f] = 1u;
    }
}

// ---------------------------------------------------------------------------------------------
// Contrast factor on two frames per block (contrast_factor.rs:16-64): the same blurred two-frame tile as the
// level front-end (sigma 1.0, one-pixel ring), then the simple Scharr gradient and
//   pass CMAX : the per-frame maximum of v = f64(lx*lx) + f64(ly*ly) over interior pixels,
//   pass CHIST: the histogram of floor(nbins * (sqrt(v) / hmax)).
// The histogram pass does not evaluate the f64 square root and division per pixel.  The bin index is a
// non-decreasing function of v (sqrt, division by a positive constant, multiplication and floor are all
// monotone under round-to-nearest), so bin k is exactly the v-interval [T[k], T[k+1]) where T[k] is the
// smallest f64 with floor(nbins * (sqrt(T[k]) / hmax)) >= k.  k_contrast_thresholds finds the T[k] by
// bisection over the f64 bit patterns with the reference's own expression; the pass then takes an f32
// estimate of the bin and moves it to the interval that contains v (two f64 compares).
__global__ __launch_bounds__(512) void k_contrast_thresholds(const unsigned long long* __restrict__ cmax, int nbins,
                                                             double* __restrict__ thr)
{
    const int f = blockIdx.x, k = threadIdx.x;
    if (k > nbins) return;
    double* T = thr + (size_t)f * 512;
    const double vmax = __longlong_as_double((long long)cmax[f]);
    const double hmax = sqrt(vmax);
    if (k == 0) { T[0] = 0.0; return; }
    if (k == nbins || !(hmax > 0.0)) { T[k] = __longlong_as_double(0x7FF0000000000000ll); return; }   // +inf: bin nbins-1 is open-ended
    // smallest bit pattern in [0, bits(vmax)] whose bin is >= k; if even vmax falls short, nothing reaches bin k
    unsigned long long lo = 0, hi = cmax[f];
    auto bin_of = [&](unsigned long long bits) {
        double modg = sqrt(__longlong_as_double((long long)bits));
        return floor((double)nbins * (modg / hmax));
    };
    if (bin_of(hi) < (double)k) { T[k] = __longlong_as_double(0x7FF0000000000000ll); return; }
    while (lo < hi) {
        unsigned long long mid = lo + ((hi - lo) >> 1);
        if (bin_of(mid) >= (double)k) hi = mid;
        else lo = mid + 1;
    }
    T[k] = __longlong_as_double((long long)lo);
}

template <typename InT, int EPI>
__global__ __launch_bounds__(kFNT) void k_contrast_pair(const InT* __restrict__ in, int w, int h, size_t fs, int n,
                                                        GaussTaps taps, unsigned long long* __restrict__ cmax,
                                                        const double* __restrict__ thr, uint32_t* __restrict__ hist,
                                                        uint32_t* __restrict__ npoints, int nbins,
                                                        uint32_t* __restrict__ fine, const uint32_t* __restrict__ flag, int ctiles)
{
    constexpr int R = 2, SG = 1, TH = kFTH, NT = kFNT;
    constexpr int CI = kTW + 16, CG = kTW + 8;
    constexpr int GH = TH + 2 * SG, IH = GH + 2 * R;
    __shared__ __attribute__((aligned(16))) v2f s_a[IH * CI];
    __shared__ __attribute__((aligned(16))) v2f s_h[IH * CG];
    __shared__ uint32_t s_hist[(EPI == EPI_CHIST) ? 2 * 512 : 1];
    __shared__ double s_thr[(EPI == EPI_CHIST) ? 2 * 512 : 1];
    __shared__ double s_red[(EPI == EPI_CMAX) ? 2 * (NT / 64) : 1];
    __shared__ uint32_t s_fine[(EPI == EPI_CMAX) ? 2 * kFineBins : 1];
    const uint3 tile = xcd_tile(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z));
    const int fa = 2 * (int)tile.z;
    const bool has_b = fa + 1 < n;
    const int fb = has_b ? fa + 1 : fa;
    const int tx0 = (int)tile.x * kTW;
    const int tid = threadIdx.x, lane = tid & 63;
    // the exact histogram is only needed for the frames k_contrast_resolve could not settle
    if (EPI == EPI_CHIST && flag && !flag[fa] && !flag[fb]) return;
    if (EPI == EPI_CMAX && fine) {
        for (int i = tid; i < 2 * kFineBins; i += NT) s_fine[i] = 0;   // ordered before use by the tile barriers
    }
    if (EPI == EPI_CHIST) {
        for (int i = tid; i < 2 * 512; i += NT) {
            s_hist[i] = 0;
            const int f = i >> 9, kk = i & 511;
            s_thr[i] = kk <= nbins ? thr[(size_t)(f ? fb : fa) * 512 + kk] : 0.0;
        }
    }
    const float4* g4 = reinterpret_cast<const float4*>(s_a);
    double lmax[2] = {-1.0, -1.0};
    float inv_hmax[2] = {0.0f, 0.0f};
    uint32_t npts[2] = {0u, 0u};
    if (EPI == EPI_CHIST) {
        inv_hmax[0] = 1.0f / sqrtf((float)__longlong_as_double((long long)cmax[fa]));
        inv_hmax[1] = 1.0f / sqrtf((float)__longlong_as_double((long long)cmax[fb]));
    }
    // ctiles vertically adjacent tiles per block: the per-frame maximum / histogram is flushed to HBM once
    // per block, and with one flush per tile the device-scope atomics on the 300 bins of a frame (1350 blocks
    // each) took as long as the arithmetic (rocprof: 850 us vs 405 us for the max pass)
    PairTileRegs<R, SG, TH, NT, InT> regs;
    if (tx0 >= 8 && tx0 + kTW + 8 <= w) regs.fetch(in, w, h, fs, fa, fb, tx0, (int)tile.y * ctiles * TH);
    for (int it = 0; it < ctiles; ++it) {
    const int ty0 = ((int)tile.y * ctiles + it) * TH;
    if (ty0 >= h) break;
    if (it) __syncthreads();                     // the previous tile's readers are done with s_a
    const int ty1 = (it + 1 < ctiles && ty0 + TH < h) ? ty0 + TH : -1;
    pair_blur_tile<R, SG, TH, NT, InT>(in, w, h, fs, fa, fb, tx0, ty0, ty1, taps, regs, s_a, s_h);   // ends with a barrier
    for (int idx = tid; idx < TH * (kTW / 4); idx += NT) {
        const int q = idx / (kTW / 4), c = idx - q * (kTW / 4);
        const int x0 = tx0 + 4 * c, y = ty0 + q;
        if (x0 >= w || y < 1 || y > h - 2) continue;   // contrast_factor.rs:27-37: interior pixels only
        const int r0 = q + SG;
        v2f m[12], z[12], pz[12];
        lds_read12<CG, 3, 8>(g4, r0 - 1, c, m);
        lds_read12<CG, 3, 8>(g4, r0, c, z);
        lds_read12<CG, 3, 8>(g4, r0 + 1, c, pz);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            // simple Scharr (derivatives.rs:3-11) in the reference's lane order
            v2f hx_m = m[5 + o] - m[3 + o];
            v2f hx_0 = z[5 + o] - z[3 + o];
            v2f hx_p = pz[5 + o] - pz[3 + o];
            v2f lx = (splat(3.0f) * hx_m + splat(10.0f) * hx_0) + splat(3.0f) * hx_p;
            v2f hy_m = (splat(3.0f) * m[3 + o] + splat(10.0f) * m[4 + o]) + splat(3.0f) * m[5 + o];
            v2f hy_p = (splat(3.0f) * pz[3 + o] + splat(10.0f) * pz[4 + o]) + splat(3.0f) * pz[5 + o];
            v2f ly = hy_p - hy_m;
            v2f lx2 = lx * lx, ly2 = ly * ly;           // squares in f32, sum in f64
            const int x = x0 + o;
            if (x < 1 || x > w - 2) continue;
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                if (f == 1 && !has_b) continue;
                const double v = (double)lx2[f] + (double)ly2[f];
                if (EPI == EPI_CMAX) {
                    lmax[f] = v > lmax[f] ? v : lmax[f];
                    if (fine && v != 0.0) {            // modg != 0: counted, and filed by magnitude for the order statistic
                        atomicAdd(&s_fine[f * kFineBins + fine_key(v)], 1u);
                        npts[f] += 1u;
                    }
                } else if (v != 0.0) {                 // modg != 0
                    const double* T = s_thr + f * 512;
                    int b = (int)((float)nbins * (sqrtf((float)v) * inv_hmax[f]));
                    b = b < 0 ? 0 : (b > nbins - 1 ? nbins - 1 : b);
                    b += (v >= T[b + 1] ? 1 : 0) - (v < T[b] ? 1 : 0);   // the f32 estimate is off by at most one bin
                    if (v < T[b] || v >= T[b + 1]) {                      // (kept exact regardless)
                        while (b > 0 && v < T[b]) --b;
                        while (b < nbins - 1 && v >= T[b + 1]) ++b;
                    }
                    atomicAdd(&s_hist[f * 512 + b], 1u);
                    npts[f] += 1u;
                }
            }
        }
    }
    }
    if (EPI == EPI_CMAX) {
        // non-negative doubles order like their bit patterns: wave max by shuffles, one atomic per block and frame
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            double v = lmax[f];
            for (int off = 32; off > 0; off >>= 1) {
                double o = __shfl_down(v, off);
                v = o > v ? o : v;
            }
            if ((tid & 63) == 0) s_red[f * (NT / 64) + (tid >> 6)] = v;
        }
        __syncthreads();
        if (tid < 2) {
            double mx = s_red[tid * (NT / 64)];
            for (int i = 1; i < NT / 64; ++i) mx = s_red[tid * (NT / 64) + i] > mx ? s_red[tid * (NT / 64) + i] : mx;
            if (mx >= 0.0 && (tid == 0 || has_b))
                atomicMax(&cmax[tid ? fb : fa], (unsigned long long)__double_as_longlong(mx));
        }
        if (fine) {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                uint32_t v = npts[f];
                for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
                if (lane == 0 && v && (f == 0 || has_b)) atomicAdd(&npoints[f ? fb : fa], v);
            }
            for (int i = tid; i < 2 * kFineBins; i += NT) {   // s_fine is complete: the barrier above
                const int f = i / kFineBins;
                if (s_fine[i] && (f == 0 || has_b)) atomicAdd(&fine[(size_t)(f ? fb : fa) * kFineBins + (i - f * kFineBins)], s_fine[i]);
            }
        }
    }
    if (EPI == EPI_CHIST) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            uint32_t v = flag ? 0u : npts[f];         // with the fine pass on, the points were counted there
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane == 0 && v) atomicAdd(&s_hist[f * 512 + 511], v);
        }
        __syncthreads();
        for (int i = tid; i < 2 * 512; i += NT) {
            const int f = i >> 9, kk = i & 511;
            if (f == 1 && !has_b) continue;
            if (!s_hist[i]) continue;
            if (kk < nbins) atomicAdd(&hist[(size_t)(f ? fb : fa) * nbins + kk], s_hist[i]);
            else if (kk == 511) atomicAdd(&npoints[f ? fb : fa], s_hist[i]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Level front-end: everything a level needs before its diffusion, in one pass over the input tile.
//   levels >= 1 (R = 2, FLOW):  Lsmooth = blur(Lt, 1.0) -> simple Scharr -> Lflow = g2   (lib.rs:232-248)
//                               and the multiscale first derivatives {Lx, Ly} of Lsmooth (detector_response.rs:63-64)
//   level 0     (R = 4, !FLOW): Lt[0] = Lsmooth[0] = blur(image, 1.6) (lib.rs:199-201) and its {Lx, Ly}
// Lsmooth itself never goes to HBM (unless the parity taps ask for it): the blurred tile plus an SG-pixel
// ring lives in LDS and both derivative families read it there.  Same exact-arithmetic rules as
// k_blur_tile: every intermediate is a rounded f32 in LDS, positions hold values at clamped coordinates.
template <int R, int SG, typename InT, bool FLOW>
__global__ __launch_bounds__(256) void k_level_front(const InT* __restrict__ in, int w, int h, size_t fs,
                                                     GaussTaps taps, OffK k, float* __restrict__ out_g,
                                                     float* __restrict__ out_flow, float2* __restrict__ out_xy,
                                                     const float* __restrict__ invk, int invk_off)
{
    constexpr int N = 2 * R + 1;
    constexpr int GW = kTW + 2 * SG, GH = kTH + 2 * SG;
    constexpr int IW = GW + 2 * R, IH = GH + 2 * R;
    __shared__ float s_a[IH * IW];   // input tile, later reused for the blurred tile
    __shared__ float s_h[IH * GW];
    float* s_in = s_a;
    float* s_g = s_a;
    const int frame = blockIdx.z;
    const int tx0 = blockIdx.x * kTW, ty0 = blockIdx.y * kTH;
    const int tid = threadIdx.x;
    const InT* src = in + (size_t)frame * fs;
    for (int idx = tid; idx < IH * IW; idx += 256) {
        int iy = idx / IW, ix = idx - iy * IW;
        int cx = clampi(tx0 - SG - R + ix, 0, w - 1);
        int cy = clampi(ty0 - SG - R + iy, 0, h - 1);
        s_in[idx] = load_px(src, (size_t)cy * w + cx);
    }
    __syncthreads();
    for (int idx = tid; idx < IH * GW; idx += 256) {
        int j = idx / GW, p = idx - j * GW;
        int cc = clampi(tx0 - SG + p, 0, w - 1);
        s_h[idx] = lane4_dot<N>(&s_in[j * IW + (cc - tx0 + SG)], 1, taps.k);
    }
    __syncthreads();  // s_in is dead from here on: s_g overwrites it
    for (int idx = tid; idx < GH * GW; idx += 256) {
        int q = idx / GW, p = idx - q * GW;
        int rc = clampi(ty0 - SG + q, 0, h - 1);
        s_g[idx] = lane4_dot<N>(&s_h[(rc - ty0 + SG) * GW + p], GW, taps.k);
    }
    __syncthreads();
    float inverse_k = 0.0f;
    if (FLOW) inverse_k = invk[(size_t)frame * 8 + invk_off];
    for (int idx = tid; idx < kTH * kTW; idx += 256) {
        int q = idx / kTW, p = idx - q * kTW;
        int x = tx0 + p, y = ty0 + q;
        if (x >= w || y >= h) continue;
        const float* g = &s_g[(q + SG) * GW + (p + SG)];
        const size_t o = (size_t)frame * fs + (size_t)y * w + x;
        if (out_g) out_g[o] = g[0];
        if (FLOW) {
            // simple Scharr (derivatives.rs:3-11) + pm_g2 (nonlinear_diffusion.rs:80)
            float hx_m = g[-GW + 1] - g[-GW - 1];
            float hx_0 = g[1] - g[-1];
            float hx_p = g[GW + 1] - g[GW - 1];
            float lx = (3.0f * hx_m + 10.0f * hx_0) + 3.0f * hx_p;
            float hy_m = (3.0f * g[-GW - 1] + 10.0f * g[-GW]) + 3.0f * g[-GW + 1];
            float hy_p = (3.0f * g[GW - 1] + 10.0f * g[GW]) + 3.0f * g[GW + 1];
            float ly = hy_p - hy_m;
            out_flow[o] = 1.0f / (1.0f + inverse_k * (lx * lx + ly * ly));
        }
        // multiscale Scharr first derivatives (derivatives.rs:23-49), taps at -SG, 0, +SG
        float mm = g[-SG * GW - SG], m0 = g[-SG * GW], mp = g[-SG * GW + SG];
        float zm = g[-SG], zp = g[SG];
        float pm = g[SG * GW - SG], p0 = g[SG * GW], pp = g[SG * GW + SG];
        float mlx = off_combine(k, mp - mm, zp - zm, pp - pm);
        float mly = off_combine(k, pm, p0, pp) - off_combine(k, mm, m0, mp);
        out_xy[o] = make_float2(mlx, mly);
    }
}

// ---------------------------------------------------------------------------------------------
// Fused Gaussian tile kernel: [u8->f32] -> H pass -> V pass -> epilogue, all intermediates
// materialised as rounded f32 in LDS exactly like the reference's intermediate images.
//   R = Gaussian radius (4 for sigma 1.6, 2 for sigma 1.0), E = extra halo the epilogue needs.
template <int R, int E, typename InT, int EPI>
__global__ __launch_bounds__(256) void k_blur_tile(const InT* __restrict__ in, int w, int h, size_t in_fs,
                                                   GaussTaps taps, float* __restrict__ out_g,
                                                   float* __restrict__ out_flow, size_t out_fs,
                                                   const float* __restrict__ invk, int invk_off,
                                                   unsigned long long* __restrict__ cmax,
                                                   uint32_t* __restrict__ hist, uint32_t* __restrict__ npoints,
                                                   int nbins)
{
    constexpr int N = 2 * R + 1;
    constexpr int GW = kTW + 2 * E, GH = kTH + 2 * E;  // blurred region
    constexpr int IW = GW + 2 * R, IH = GH + 2 * R;    // input region
    __shared__ float s_in[IH * IW];
    __shared__ float s_h[IH * GW];
    __shared__ float s_g[GH * GW];
    __shared__ uint32_t s_hist[(EPI == EPI_CHIST) ? 512 : 1];
    __shared__ double s_red[(EPI == EPI_CMAX) ? 4 : 1];

    const int frame = blockIdx.z;
    const int tx0 = blockIdx.x * kTW, ty0 = blockIdx.y * kTH;
    const int tid = threadIdx.x;
    const InT* src = in + (size_t)frame * in_fs;

    for (int idx = tid; idx < IH * IW; idx += 256) {
        int iy = idx / IW, ix = idx - iy * IW;
        int cx = clampi(tx0 - E - R + ix, 0, w - 1);
        int cy = clampi(ty0 - E - R + iy, 0, h - 1);
        s_in[idx] = load_px(src, (size_t)cy * w + cx);
    }
    if (EPI == EPI_CHIST)
        for (int i = tid; i < 512; i += 256) s_hist[i] = 0;
    __syncthreads();
    // horizontal pass at the clamped column of each position
    for (int idx = tid; idx < IH * GW; idx += 256) {
        int j = idx / GW, p = idx - j * GW;
        int cc = clampi(tx0 - E + p, 0, w - 1);
        s_h[idx] = lane4_dot<N>(&s_in[j * IW + (cc - tx0 + E)], 1, taps.k);
    }
    __syncthreads();
    // vertical pass at the clamped row of each position
    for (int idx = tid; idx < GH * GW; idx += 256) {
        int q = idx / GW, p = idx - q * GW;
        int rc = clampi(ty0 - E + q, 0, h - 1);
        s_g[idx] = lane4_dot<N>(&s_h[(rc - ty0 + E) * GW + p], GW, taps.k);
    }
    __syncthreads();

    if (EPI == EPI_BLUR) {
        float* dst = out_g + (size_t)frame * out_fs;
        for (int idx = tid; idx < kTH * kTW; idx += 256) {
            int q = idx / kTW, p = idx - q * kTW;
            int x = tx0 + p, y = ty0 + q;
            if (x < w && y < h) dst[(size_t)y * w + x] = s_g[(q + E) * GW + (p + E)];
        }
        return;
    }

    // simple Scharr on the blurred tile (derivatives.rs:3-11) in the reference's lane order:
    //   Lx = V[3,10,3](H[-1,0,1] g):  hx = g(x+1) - g(x-1);  Lx = (3*hx(y-1) + 10*hx(y)) + 3*hx(y+1)
    //   Ly = V[-1,0,1](H[3,10,3] g):  hy = (3*g(x-1) + 10*g(x)) + 3*g(x+1);  Ly = hy(y+1) - hy(y-1)
    double lmax = -1.0;
    float inverse_k = 0.0f;
    double hmax = 0.0;
    if (EPI == EPI_FLOW) inverse_k = invk[(size_t)frame * 8 + invk_off];
    if (EPI == EPI_CHIST) hmax = sqrt(__longlong_as_double((long long)cmax[frame]));
    for (int idx = tid; idx < kTH * kTW; idx += 256) {
        int q = idx / kTW, p = idx - q * kTW;
        int x = tx0 + p, y = ty0 + q;
        if (x >= w || y >= h) continue;
        const float* g = &s_g[(q + E) * GW + (p + E)];
        float hx_m = g[-GW + 1] - g[-GW - 1];
        float hx_0 = g[1] - g[-1];
        float hx_p = g[GW + 1] - g[GW - 1];
        float lx = (3.0f * hx_m + 10.0f * hx_0) + 3.0f * hx_p;
        float hy_m = (3.0f * g[-GW - 1] + 10.0f * g[-GW]) + 3.0f * g[-GW + 1];
        float hy_p = (3.0f * g[GW - 1] + 10.0f * g[GW]) + 3.0f * g[GW + 1];
        float ly = hy_p - hy_m;
        if (EPI == EPI_FLOW) {
            out_g[(size_t)frame * out_fs + (size_t)y * w + x] = g[0];
            // pm_g2, nonlinear_diffusion.rs:80
            out_flow[(size_t)frame * out_fs + (size_t)y * w + x] =
                1.0f / (1.0f + inverse_k * (lx * lx + ly * ly));
        } else {
            // contrast_factor.rs:27-37: interior pixels only; squares in f32, sum in f64
            if (x >= 1 && x <= w - 2 && y >= 1 && y <= h - 2) {
                double v = (double)(lx * lx) + (double)(ly * ly);
                if (EPI == EPI_CMAX) {
                    lmax = v > lmax ? v : lmax;
                } else {
                    double modg = sqrt(v);
                    if (modg != 0.0) {
                        double b = floor((double)nbins * (modg / hmax));
                        int bin = b >= (double)nbins ? nbins - 1 : (int)b;
                        atomicAdd(&s_hist[bin], 1u);
                        atomicAdd(&s_hist[511], 1u);  // num_points
                    }
                }
            }
        }
    }
    if (EPI == EPI_CMAX) {
        // non-negative doubles order like their bit patterns: wave max by shuffles, one atomic per block
        for (int off = 32; off > 0; off >>= 1) {
            double o = __shfl_down(lmax, off);
            lmax = o > lmax ? o : lmax;
        }
        if ((tid & 63) == 0) s_red[tid >> 6] = lmax;
        __syncthreads();
        if (tid == 0) {
            double m = s_red[0];
            for (int i = 1; i < 4; ++i) m = s_red[i] > m ? s_red[i] : m;
            if (m >= 0.0) atomicMax(&cmax[frame], (unsigned long long)__double_as_longlong(m));
        }
    }
    if (EPI == EPI_CHIST) {
        __syncthreads();
        for (int i = tid; i < nbins; i += 256)
            if (s_hist[i]) atomicAdd(&hist[(size_t)frame * nbins + i], s_hist[i]);
        if (tid == 0 && s_hist[511]) atomicAdd(&npoints[frame], s_hist[511]);
    }
}

// contrast_factor.rs:48-63 + the per-octave `contrast_factor *= 0.75` of lib.rs:222 and the
// inverse_k of nonlinear_diffusion.rs:73.  One thread per frame (scalar f64 work).
__global__ void k_contrast_finish(const unsigned long long* __restrict__ cmax, const uint32_t* __restrict__ hist,
                                  const uint32_t* __restrict__ npoints, int nbins, double percentile, int n,
                                  int n_octaves, double* __restrict__ contrast, float* __restrict__ invk,
                                  const uint32_t* __restrict__ flag)
{
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    if (flag && !flag[f]) return;                  // settled by k_contrast_resolve
    double hmax = sqrt(__longlong_as_double((long long)cmax[f]));
    double num_points = (double)npoints[f];
    double t = num_points * percentile;
    unsigned long long threshold = t > 0.0 ? (unsigned long long)t : 0ull;
    unsigned long long num_elements = 0;
    int k = 0;
    while (num_elements < threshold && k < nbins) {
        num_elements += hist[(size_t)f * nbins + k];
        k += 1;
    }
    contrast_write(f, hmax, (unsigned long long)k, num_elements >= threshold, nbins, n_octaves, contrast, invk);
}

// ---------------------------------------------------------------------------------------------
// half_size — image.rs:154-199.  2x2 window sum is (a+b)+(c+d) (ndarray row-wise fold), odd edges
// take the 1x2 / 2x1 / 1x1 rule from the LAST input row/column.
__global__ __launch_bounds__(256) void k_half_size(const float* __restrict__ in, float* __restrict__ out, int w,
                                                   int h, size_t in_fs, size_t out_fs)
{
    int ow = w >> 1, oh = h >> 1;
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= ow || y >= oh) return;
    const float* src = in + (size_t)blockIdx.z * in_fs;
    bool last_row = (oh * 2 != h) && (y == oh - 1);
    bool last_col = (ow * 2 != w) && (x == ow - 1);
    float v;
    if (last_row && last_col) {
        v = src[(size_t)(h - 1) * w + (w - 1)];
    } else if (last_col) {
        v = (src[(size_t)(2 * y) * w + (w - 1)] + src[(size_t)(2 * y + 1) * w + (w - 1)]) * 0.5f;
    } else if (last_row) {
        v = (src[(size_t)(h - 1) * w + 2 * x] + src[(size_t)(h - 1) * w + 2 * x + 1]) * 0.5f;
    } else {
        const float* p = src + (size_t)(2 * y) * w + 2 * x;
        if constexpr (kArithHalfSeq) v = (((p[0] + p[1]) + p[w]) + p[w + 1]) * 0.25f;   // ndarray's fold over the window in memory order
        else v = ((p[0] + p[1]) + (p[w] + p[w + 1])) * 0.25f;
    }
    out[(size_t)blockIdx.z * out_fs + (size_t)y * ow + x] = v;
}

// ---------------------------------------------------------------------------------------------
// calculate_step — nonlinear_diffusion.rs:14-58.  Jacobi: flows from the pre-update image `src`,
//   hf(x) = ((0.5*tau)*(c(x)+c(x+1)))*(L(x+1)-L(x)),  vf likewise in y,
//   dst = (((L + hf(x)) - hf(x-1)) + vf(y)) - vf(y-1), terms that cross the border are skipped.
// HBM traffic per pixel-step: 4 B (L) + 4 B (c) read, 4 B write = 12 B.
__device__ __forceinline__ float fed_flow(float ht, float ca, float cb, float a, float b)
{
    return (ht * (ca + cb)) * (b - a);
}

// scalar variant (any width)
__global__ __launch_bounds__(256) void k_fed_step(const float* __restrict__ src, const float* __restrict__ c,
                                                  float* __restrict__ dst, int w, int h, size_t fs, float half_tau)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    size_t base = (size_t)blockIdx.z * fs + (size_t)y * w + x;
    const float* L = src + base;
    const float* C = c + base;
    float l0 = L[0], c0 = C[0];
    float v = l0;
    if (x < w - 1) v = v + fed_flow(half_tau, c0, C[1], l0, L[1]);
    if (x > 0) v = v - fed_flow(half_tau, C[-1], c0, L[-1], l0);
    if (y < h - 1) v = v + fed_flow(half_tau, c0, C[w], l0, L[w]);
    if (y > 0) v = v - fed_flow(half_tau, C[-w], c0, L[-w], l0);
    dst[base] = v;
}

// the half-tau values of one temporally blocked launch (up to 16 steps)
constexpr int kFedMaxBlock = 16;
struct FedTaus {
    float half_tau[kFedMaxBlock];
};

// ---------------------------------------------------------------------------------------------
// calculate_step for two frames per block with the image held in REGISTERS across the T steps of a launch.
// A thread owns a 4x4 pixel patch of both frames ({a, b} pairs, packed arithmetic); a block is 16x16 patches
// = a 64x64 window whose outer patch ring(s) are the halo: the valid region shrinks by one pixel per step, so one
// ring (56x56 tile) serves T <= 4 and two rings (48x48) T <= 8 — on the smaller octaves, which run 4 to 29 steps
// per level, halving the launches and the passes over memory outweighs the smaller useful tile.  Per step a thread needs only the neighbouring patches' facing edges:
// left/right columns come from the adjacent lanes with DPP row shifts (a 16-lane DPP row is one patch row),
// top/bottom rows go through a 16 KB LDS exchange (two barriers per step).  Every flow is
// evaluated once (the reference's Jacobi update needs each twice, as +flow for one pixel and -flow for the
// other).  A flow across the image border is replaced by +0: L is never -0 (it starts from sums of
// non-negative products and x + (-0) = x, (+0) - (+0) = +0), so adding or subtracting +0 leaves every value
// bit-identical to skipping the term as nonlinear_diffusion.rs:31-52 does.
// Output tile edge of k_fed_pair<T>: the 64 x 64 window minus a halo of whole 4-pixel patches that covers the T
// pixels a launch invalidates on every side (one patch up to 4 steps, two up to 8).
constexpr int fed_halo_patches(int T) { return (T + 3) / 4; }
constexpr int fed_tile_edge(int T) { return 64 - 8 * fed_halo_patches(T); }

__device__ __forceinline__ v2f fed_flow2(v2f ht, v2f ca, v2f cb, v2f a, v2f b) { return (ht * (ca + cb)) * (b - a); }

template <int CTRL>
__device__ __forceinline__ v2f dpp_row(v2f v)   // value of the lane CTRL selects within the 16-lane row; 0 outside it
{
    int x = __builtin_amdgcn_update_dpp(0, __float_as_int(v.x), CTRL, 0xF, 0xF, true);
    int y = __builtin_amdgcn_update_dpp(0, __float_as_int(v.y), CTRL, 0xF, 0xF, true);
    return (v2f){__int_as_float(x), __int_as_float(y)};
}

// The FED steps of a launch on the register-resident patches (shared by k_fed_pair and k_front_fed).  Every flow is
// evaluated by ONE thread and handed to the other pixel it belongs to: the flow through a patch's right edge goes to the
// right-hand neighbour lane by a DPP row shift (it is that patch's left-edge flow: same operands, same expression),
// the flow through its bottom edge goes to the patch below through LDS (s_vd).  Per step a thread exchanges its top
// image row (s_top: the patch above needs it for its bottom-edge flow) and its four bottom-edge flows; the top
// conductivity rows are static (s_ct).  Rows are updated bottom-up, so row r - 1 is still the pre-update image when the
// flow between rows r - 1 and r is taken, and the flow from the patch above is needed last (the second barrier of the
// step sits in front of row 0's update only).
// Flows across the image border: nonlinear_diffusion.rs:31-52 skips those terms.  Here the thread that EVALUATES such a
// flow takes a zero step size for it, (0 * (ca + cb)) * (b - a) = +-0 — a zero of either sign leaves every value it
// is added to or subtracted from unchanged, because no term of the update is ever -0 (k_fed_pair's header; the data
// are finite: pixels are finite and c is in [0, 1]).  Which flows those are is fixed per thread: the one through the
// patch's right edge (z_right: that edge is an image border; patches are wholly inside or outside horizontally) and
// the ones below its rows (z_below(r): row y0 + r is the image's last row or lies above its first) — five selects on
// the step size per step instead of one on every flow (56 of the 312 vector instructions of a step).
__device__ __forceinline__ void fed_steps(v2f (&L)[4][4], v2f (&C)[4][4], float4* __restrict__ s_top,
                                          float4* __restrict__ s_ct, float4* __restrict__ s_vd, const FedTaus& taus,
                                          int nsteps, int tid, int up, int dn, int x0, int y0, int w, int h)
{
    s_ct[tid * 2] = make_float4(C[0][0].x, C[0][0].y, C[0][1].x, C[0][1].y);
    s_ct[tid * 2 + 1] = make_float4(C[0][2].x, C[0][2].y, C[0][3].x, C[0][3].y);
    const bool z_right = x0 + 4 <= 0 || x0 + 4 >= w;
    bool z_below[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) z_below[r] = y0 + r < 0 || y0 + r >= h - 1;
#pragma unroll 1
    for (int t = 0; t < nsteps; ++t) {
        const float htf = taus.half_tau[t];
        const v2f ht = splat(htf);
        // an unrolled step loop lets the compiler keep every c(x) + c(x+1) sum and every neighbour's c across the
        // steps (190 VGPRs); the empty asm makes C opaque per step so they are recomputed instead
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(C[r][c]));
        s_top[tid * 2] = make_float4(L[0][0].x, L[0][0].y, L[0][1].x, L[0][1].y);
        s_top[tid * 2 + 1] = make_float4(L[0][2].x, L[0][2].y, L[0][3].x, L[0][3].y);
        __syncthreads();   // (also: every reader of the previous step's s_vd is done before this step's writes below)
        v2f vd[4];         // flow through the bottom edge of the row being updated
        {
            const float4 a = s_top[dn * 2], b = s_top[dn * 2 + 1], c = s_ct[dn * 2], d = s_ct[dn * 2 + 1];
            const v2f Lb[4] = {(v2f){a.x, a.y}, (v2f){a.z, a.w}, (v2f){b.x, b.y}, (v2f){b.z, b.w}};
            const v2f Cb[4] = {(v2f){c.x, c.y}, (v2f){c.z, c.w}, (v2f){d.x, d.y}, (v2f){d.z, d.w}};
            const v2f ht3 = splat(z_below[3] ? 0.0f : htf);
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) vd[cc] = fed_flow2(ht3, C[3][cc], Cb[cc], L[3][cc], Lb[cc]);
            s_vd[tid * 2] = make_float4(vd[0].x, vd[0].y, vd[1].x, vd[1].y);
            s_vd[tid * 2 + 1] = make_float4(vd[2].x, vd[2].y, vd[3].x, vd[3].y);
        }
        const v2f htr = splat(z_right ? 0.0f : htf);
#pragma unroll
        for (int r = 3; r >= 0; --r) {
            const v2f Lr = dpp_row<0x101>(L[r][0]), Cr = dpp_row<0x101>(C[r][0]);   // the right-hand neighbour's first column
            v2f hf[5];
#pragma unroll
            for (int c = 1; c < 4; ++c) hf[c] = fed_flow2(ht, C[r][c - 1], C[r][c], L[r][c - 1], L[r][c]);
            hf[4] = fed_flow2(htr, C[r][3], Cr, L[r][3], Lr);
            hf[0] = dpp_row<0x111>(hf[4]);                                            // the left-hand neighbour's hf[4]
            v2f vu[4];
            if (r > 0) {
                const v2f htu = splat(z_below[r > 0 ? r - 1 : 0] ? 0.0f : htf);
#pragma unroll
                for (int c = 0; c < 4; ++c) vu[c] = fed_flow2(htu, C[r - (r > 0)][c], C[r][c], L[r - (r > 0)][c], L[r][c]);
            } else {
                __syncthreads();   // the bottom-edge flows of the patch above are in s_vd
                const float4 a = s_vd[up * 2], b = s_vd[up * 2 + 1];
                vu[0] = (v2f){a.x, a.y}; vu[1] = (v2f){a.z, a.w}; vu[2] = (v2f){b.x, b.y}; vu[3] = (v2f){b.z, b.w};
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                L[r][c] = (((L[r][c] + hf[c + 1]) - hf[c]) + vd[c]) - vu[c];   // nonlinear_diffusion.rs:31-52 order
                vd[c] = vu[c];
            }
        }
    }
}

// The same steps with the conductivity sums c(x) + c(x+1), c(y) + c(y+1) of every flow held in registers for the whole
// launch (64 VGPRs instead of the 32 of the conductivities themselves): a step is 32 additions, the eight row shifts of
// the neighbours' conductivities and two LDS reads shorter.  For kernels that run at three waves per SIMD anyway
// (k_front_fed); k_fed_pair keeps four waves with the form above.
__device__ __forceinline__ void fed_steps_sums(v2f (&L)[4][4], const v2f (&C)[4][4], float4* __restrict__ s_top,
                                               float4* __restrict__ s_ct, float4* __restrict__ s_vd, const FedTaus& taus,
                                               int nsteps, int tid, int up, int dn, int x0, int y0, int w, int h)
{
    s_ct[tid * 2] = make_float4(C[0][0].x, C[0][0].y, C[0][1].x, C[0][1].y);
    s_ct[tid * 2 + 1] = make_float4(C[0][2].x, C[0][2].y, C[0][3].x, C[0][3].y);
    v2f SH[4][4], SV[4][4];   // SH[r][c]: columns c, c + 1 of row r (c = 3: with the right-hand neighbour); SV[r][c]: rows r, r + 1
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const v2f Cr = dpp_row<0x101>(C[r][0]);
#pragma unroll
        for (int c = 0; c < 3; ++c) SH[r][c] = C[r][c] + C[r][c + 1];
        SH[r][3] = C[r][3] + Cr;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) SV[r][c] = C[r][c] + C[r + 1][c];
    __syncthreads();
    {
        const float4 c = s_ct[dn * 2], d = s_ct[dn * 2 + 1];
        SV[3][0] = C[3][0] + (v2f){c.x, c.y};
        SV[3][1] = C[3][1] + (v2f){c.z, c.w};
        SV[3][2] = C[3][2] + (v2f){d.x, d.y};
        SV[3][3] = C[3][3] + (v2f){d.z, d.w};
    }
    const bool z_right = x0 + 4 <= 0 || x0 + 4 >= w;
    bool z_below[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) z_below[r] = y0 + r < 0 || y0 + r >= h - 1;
#pragma unroll 1
    for (int t = 0; t < nsteps; ++t) {
        const float htf = taus.half_tau[t];
        const v2f ht = splat(htf);
        s_top[tid * 2] = make_float4(L[0][0].x, L[0][0].y, L[0][1].x, L[0][1].y);
        s_top[tid * 2 + 1] = make_float4(L[0][2].x, L[0][2].y, L[0][3].x, L[0][3].y);
        __syncthreads();
        v2f vd[4];
        {
            const float4 a = s_top[dn * 2], b = s_top[dn * 2 + 1];
            const v2f Lb[4] = {(v2f){a.x, a.y}, (v2f){a.z, a.w}, (v2f){b.x, b.y}, (v2f){b.z, b.w}};
            const v2f ht3 = splat(z_below[3] ? 0.0f : htf);
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) vd[cc] = (ht3 * SV[3][cc]) * (Lb[cc] - L[3][cc]);
            s_vd[tid * 2] = make_float4(vd[0].x, vd[0].y, vd[1].x, vd[1].y);
            s_vd[tid * 2 + 1] = make_float4(vd[2].x, vd[2].y, vd[3].x, vd[3].y);
        }
        const v2f htr = splat(z_right ? 0.0f : htf);
#pragma unroll
        for (int r = 3; r >= 0; --r) {
            const v2f Lr = dpp_row<0x101>(L[r][0]);
            v2f hf[5];
#pragma unroll
            for (int c = 1; c < 4; ++c) hf[c] = (ht * SH[r][c - 1]) * (L[r][c] - L[r][c - 1]);
            hf[4] = (htr * SH[r][3]) * (Lr - L[r][3]);
            hf[0] = dpp_row<0x111>(hf[4]);
            v2f vu[4];
            if (r > 0) {
                const v2f htu = splat(z_below[r > 0 ? r - 1 : 0] ? 0.0f : htf);
#pragma unroll
                for (int c = 0; c < 4; ++c) vu[c] = (htu * SV[r - (r > 0)][c]) * (L[r][c] - L[r - (r > 0)][c]);
            } else {
                __syncthreads();
                const float4 a = s_vd[up * 2], b = s_vd[up * 2 + 1];
                vu[0] = (v2f){a.x, a.y}; vu[1] = (v2f){a.z, a.w}; vu[2] = (v2f){b.x, b.y}; vu[3] = (v2f){b.z, b.w};
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                L[r][c] = (((L[r][c] + hf[c + 1]) - hf[c]) + vd[c]) - vu[c];   // nonlinear_diffusion.rs:31-52 order
                vd[c] = vu[c];
            }
        }
    }
}

__device__ __forceinline__ void fed_store_patch(const v2f (&L)[4][4], float* __restrict__ dst, int fa, int fb, bool has_b,
                                                size_t fs, int w, int h, int x0, int y0)
{
    float* da = dst + (size_t)fa * fs;
    float* db = dst + (size_t)fb * fs;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int y = y0 + r;
        if (y >= h) break;
        const uint32_t o = (uint32_t)(y * w + x0) << 2;
        *reinterpret_cast<float4*>(at_bytes(da, o)) = make_float4(L[r][0].x, L[r][1].x, L[r][2].x, L[r][3].x);
        if (has_b) *reinterpret_cast<float4*>(at_bytes(db, o)) = make_float4(L[r][0].y, L[r][1].y, L[r][2].y, L[r][3].y);
    }
}

// half_size (image.rs:154-199, as k_half_size) of the level an octave ends with, from the patch the level's LAST FED launch
// holds in registers: the next octave's first image without reading the level back (w % 4 == 0, patches 4-aligned, so
// every 2 x 2 window lies inside one patch).  An odd height keeps the reference's rule: the last output row is the 1 x 2
// mean of the LAST input row alone.
__device__ __forceinline__ void fed_store_half(const v2f (&L)[4][4], float* __restrict__ half, int fa, int fb, bool has_b,
                                               size_t hfs, int w, int h, int x0, int y0)
{
    const int ow = w >> 1, oh = h >> 1;
    const bool odd_h = (h & 1) != 0;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int y = y0 + 2 * rr;
        if (y >= h) break;
        const int oy = y >> 1;
        v2f v0, v1;
        bool put = false;
        int orow = oy;
        if (oy < oh && !(odd_h && oy == oh - 1)) {
            if constexpr (kArithHalfSeq) {   // ndarray's fold over the window in memory order
                v0 = (((L[2 * rr][0] + L[2 * rr][1]) + L[2 * rr + 1][0]) + L[2 * rr + 1][1]) * splat(0.25f);
                v1 = (((L[2 * rr][2] + L[2 * rr][3]) + L[2 * rr + 1][2]) + L[2 * rr + 1][3]) * splat(0.25f);
            } else {
                v0 = ((L[2 * rr][0] + L[2 * rr][1]) + (L[2 * rr + 1][0] + L[2 * rr + 1][1])) * splat(0.25f);
                v1 = ((L[2 * rr][2] + L[2 * rr][3]) + (L[2 * rr + 1][2] + L[2 * rr + 1][3])) * splat(0.25f);
            }
            put = true;
        } else if (odd_h && y == h - 1) {
            v0 = (L[2 * rr][0] + L[2 * rr][1]) * splat(0.5f);
            v1 = (L[2 * rr][2] + L[2 * rr][3]) * splat(0.5f);
            orow = oh - 1;
            put = oh > 0;
        }
        if (put) {
            const uint32_t o = (uint32_t)(orow * ow + (x0 >> 1)) << 2;
            *reinterpret_cast<float2*>(at_bytes(half + (size_t)fa * hfs, o)) = make_float2(v0.x, v1.x);
            if (has_b) *reinterpret_cast<float2*>(at_bytes(half + (size_t)fb * hfs, o)) = make_float2(v0.y, v1.y);
        }
    }
}

template <int T>
__global__ __launch_bounds__(256, 4) void k_fed_pair(const float* __restrict__ src, const float* __restrict__ cnd,
                                                  float* __restrict__ dst, int w, int h, size_t fs, int n, FedTaus taus,
                                                  float* __restrict__ half_out, size_t half_fs)
{
    __shared__ __attribute__((aligned(16))) float4 s_top[256 * 2];   // [patch][4 px x 2 frames]: top image rows
    __shared__ __attribute__((aligned(16))) float4 s_vd[256 * 2];    // bottom-edge flows
    __shared__ __attribute__((aligned(16))) float4 s_ct[256 * 2];    // top rows of C (fixed for the whole launch)
    const uint3 tile = xcd_tile(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z));
    const int fa = 2 * (int)tile.z;
    const bool has_b = fa + 1 < n;
    const int fb = has_b ? fa + 1 : fa;
    const int tid = threadIdx.x, pc = tid & 15, pr = tid >> 4;
    constexpr int HP = fed_halo_patches(T), U = fed_tile_edge(T);
    const int x0 = (int)tile.x * U - 4 * HP + 4 * pc, y0 = (int)tile.y * U - 4 * HP + 4 * pr;
    const bool col_in = x0 >= 0 && x0 < w;   // w % 4 == 0: a patch column is entirely inside or outside
    v2f L[4][4], C[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int y = y0 + r;
        float4 la = make_float4(0.f, 0.f, 0.f, 0.f), lb = la, ca = la, cb = la;
        if (col_in && y >= 0 && y < h) {
            const uint32_t o = (uint32_t)(y * w + x0) << 2;
            la = *reinterpret_cast<const float4*>(at_bytes(src + (size_t)fa * fs, o));
            lb = *reinterpret_cast<const float4*>(at_bytes(src + (size_t)fb * fs, o));
            ca = *reinterpret_cast<const float4*>(at_bytes(cnd + (size_t)fa * fs, o));
            cb = *reinterpret_cast<const float4*>(at_bytes(cnd + (size_t)fb * fs, o));
        }
        L[r][0] = (v2f){la.x, lb.x}; L[r][1] = (v2f){la.y, lb.y}; L[r][2] = (v2f){la.z, lb.z}; L[r][3] = (v2f){la.w, lb.w};
        C[r][0] = (v2f){ca.x, cb.x}; C[r][1] = (v2f){ca.y, cb.y}; C[r][2] = (v2f){ca.z, cb.z}; C[r][3] = (v2f){ca.w, cb.w};
    }
    const int up = pr > 0 ? tid - 16 : tid, dn = pr < 15 ? tid + 16 : tid;   // block-edge patches are halo
    const bool useful = pc >= HP && pc <= 15 - HP && pr >= HP && pr <= 15 - HP && col_in;
    fed_steps(L, C, s_top, s_ct, s_vd, taus, T, tid, up, dn, x0, y0, w, h);
    if (useful) {
        fed_store_patch(L, dst, fa, fb, has_b, fs, w, h, x0, y0);
        if (half_out) fed_store_half(L, half_out, fa, fb, has_b, half_fs, w, h, x0, y0);
    }
}


// ---------------------------------------------------------------------------------------------
// Level front-end AND the first FED launch of the level in one kernel (two frames per block, w % 4 == 0).
// lib.rs:230-256 does, for level i > 0:  Lt_i <- copy of Lt_{i-1};  Lsmooth = blur(Lt_i, 1.0);  Lflow = pm_g2(simple
// Scharr(Lsmooth));  {Lx, Ly} = multiscale Scharr(Lsmooth);  then the FED steps on Lt_i with Lflow.  The front kernel
// and the FED kernel above split that at Lflow: 16 B/pixel + 12 B/pixel per launch.  Here a block keeps the FED
// kernel's organisation — a 64 x 64 window of 16 x 16 threads, each owning a 4 x 4 patch of both frames in registers —
// and builds the conductivity of its patch itself, so Lflow never leaves the chip:
//   1. the input window (+2 px: the blur radius) goes to LDS; every thread blurs ITS patch (horizontal then vertical,
//      taps in the reference's lane order), the 64 x 64 blurred window replaces the input in LDS;
//   2. from it every thread takes the simple Scharr gradient and pm_g2 of its patch (-> C[4][4], registers), and the
//      threads of the useful region the multiscale Scharr {Lx, Ly} of theirs (-> HBM);
//   3. the FED steps run exactly as in k_fed_pair (same exchange through LDS, same border rules).
// HBM traffic: 4 B in (Lt), 4 B (Lt') + 8 B ({Lx, Ly}) out = 16 B/pixel for what took 28.
// Validity: the conductivity of the window's outermost pixels needs the blurred values one pixel outside it, so the
// blur also covers that ring (260 pixels, one or two per thread) when RING is set; a launch of T steps is then right T
// pixels inside the window, as in k_fed_pair: HP halo patches serve T <= 4 HP steps — without the ring (cheaper by
// 6 %) the conductivity is right on window pixels 1..62 only and HP patches serve T <= 4 HP - 1.
// Every value is computed by the same expression, in the same order, as in k_level_front2 / k_fed_pair.
constexpr int kFFWaves = 3;                    // k_front_fed: waves per SIMD the register budget is held to
constexpr int kFFW = 64;                       // window edge
constexpr int kFFIn = kFFW + 6;                // input rows: the window, its one-pixel ring, the blur radius 2
constexpr int kFFInC = kFFW + 8;               // input columns held (4 either side: whole 16-byte chunks)
constexpr int kFFGS = kFFW + 4;                // row stride of the blurred window in LDS (2 apron columns either side)
constexpr int front_fed_tile(int HP) { return kFFW - 8 * HP; }

// LDS rows of k_front_fed are stored like the two-frame tiles above: the 16-byte chunks ({col, a}, {col, b}, {col+1, a},
// {col+1, b}) of a row in two planes, even chunks first, so that the sixteen threads of a patch row, which read chunks
// two apart, touch consecutive 16-byte words.  RS = chunks per row (even).
template <int RS>
__device__ __forceinline__ int ff_chunk(int row, int ci) { return row * RS + ((ci & 1) ? RS / 2 : 0) + (ci >> 1); }
template <int RS>
__device__ __forceinline__ int ff_elem(int row, int col) { return 2 * ff_chunk<RS>(row, col >> 1) + (col & 1); }

// the interior input window of k_front_fed (rows wy0 - 3 .., columns wx0 - 4 .., all inside the image) as ITEMS float4 per
// thread and frame; every load is issued before any is consumed
template <int ITEMS>
__device__ __forceinline__ void front_fed_fetch(float4 (&ra)[ITEMS], float4 (&rb)[ITEMS], const float* __restrict__ srca,
                                                const float* __restrict__ srcb, int w, int wx0, int wy0, int tid)
{
    constexpr int NCH = kFFIn * (kFFInC / 4);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int idx = tid + 256 * i;
        if (idx < NCH) {
            const int iy = idx / (kFFInC / 4), c4 = idx - iy * (kFFInC / 4);
            // (the window lies inside the image here: a 32-bit element offset from the frame's uniform base — an SGPR base and
            // a VGPR offset in the load, instead of a 64-bit address built per item)
            const uint32_t o = (uint32_t)((wy0 - 3 + iy) * w + (wx0 - 4 + 4 * c4)) << 2;      // bytes (a frame is far below 4 GB)
            ra[i] = *reinterpret_cast<const float4*>(at_bytes(srca, o));
            rb[i] = *reinterpret_cast<const float4*>(at_bytes(srcb, o));
        }
    }
}

template <int SG, int HP, bool RING, bool WRITE_FLOW>
__global__ __launch_bounds__(256, kFFWaves) void k_front_fed(const float* __restrict__ in, int w, int h, size_t fs, int n,
                                                      GaussTaps taps, OffK k, FedTaus taus, int nsteps,
                                                      float* __restrict__ out_lt, float* __restrict__ out_flow,
                                                      float2* __restrict__ out_xy, const float* __restrict__ invk,
                                                      int invk_off, float* __restrict__ half_out, size_t half_fs)
{
    // one block of LDS, three lives: input window [kFFIn][kFFInC], blurred window [kFFW + 2][kFFGS] (one apron row
    // above and below), FED exchange buffers
    __shared__ __attribute__((aligned(16))) v2f s_buf[kFFIn * kFFInC];
    static_assert((kFFW + 2) * kFFGS <= kFFIn * kFFInC, "blurred window fits in the input window's space");
    static_assert(3 * 256 * 2 * 2 <= kFFIn * kFFInC, "FED exchange buffers fit");
    constexpr int U = front_fed_tile(HP);
    const uint3 tile = xcd_tile(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z));
    const int tid = threadIdx.x, pc = tid & 15, pr = tid >> 4;
    const int fa = 2 * (int)tile.z;
    const bool has_b = fa + 1 < n;
    const int fb = has_b ? fa + 1 : fa;
    const int wx0 = (int)tile.x * U - 4 * HP, wy0 = (int)tile.y * U - 4 * HP;   // window origin in the image
    const float* srca = in + (size_t)fa * fs;
    const float* srcb = in + (size_t)fb * fs;
    // ---- 1a. input window: rows wy0 - 3 .., columns wx0 - 4 .., clamped coordinates outside the image ----
    {
        const bool inside = wx0 >= 4 && wx0 + kFFW + 4 <= w && wy0 >= 3 && wy0 + kFFW + 3 <= h;
        if (inside) {
            // every load of the thread is issued before the first one is consumed (a loop that loads, waits and stores per
            // item is five dependent round trips to HBM: 8.5 us of a 21 us block, measured)
            constexpr int NCH = kFFIn * (kFFInC / 4), ITEMS = (NCH + 255) / 256;
            float4 ra[ITEMS], rb[ITEMS];
            front_fed_fetch<ITEMS>(ra, rb, srca, srcb, w, wx0, wy0, tid);
            float4* d = reinterpret_cast<float4*>(s_buf);
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const int idx = tid + 256 * i;
                if (idx < NCH) {
                    const int iy = idx / (kFFInC / 4), c4 = idx - iy * (kFFInC / 4);
                    d[ff_chunk<kFFInC / 2>(iy, 2 * c4)] = make_float4(ra[i].x, rb[i].x, ra[i].y, rb[i].y);
                    d[ff_chunk<kFFInC / 2>(iy, 2 * c4 + 1)] = make_float4(ra[i].z, rb[i].z, ra[i].w, rb[i].w);
                }
            }
        } else {
            // windows on the image border (15 % of them at 1080p): per-element clamped loads, issued in batches of eight
            // per thread before they are consumed (one load per round trip made these windows three times as slow)
            constexpr int NE = kFFIn * kFFInC, BATCH = 8;
            for (int base = 0; base < NE; base += 256 * BATCH) {
                float ea[BATCH], eb[BATCH];
#pragma unroll
                for (int i = 0; i < BATCH; ++i) {
                    const int idx = base + tid + 256 * i;
                    ea[i] = eb[i] = 0.0f;
                    if (idx < NE) {
                        const int iy = idx / kFFInC, ix = idx - iy * kFFInC;
                        const int cx = clampi(wx0 - 4 + ix, 0, w - 1), cy = clampi(wy0 - 3 + iy, 0, h - 1);
                        const size_t o = (size_t)cy * w + cx;
                        ea[i] = srca[o];
                        eb[i] = srcb[o];
                    }
                }
#pragma unroll
                for (int i = 0; i < BATCH; ++i) {
                    const int idx = base + tid + 256 * i;
                    if (idx < NE) {
                        const int iy = idx / kFFInC, ix = idx - iy * kFFInC;
                        s_buf[ff_elem<kFFInC / 2>(iy, ix)] = (v2f){ea[i], eb[i]};
                    }
                }
            }
        }
    }
    __syncthreads();
    // ---- 1b. Gaussian blur (sigma 1.0, 5 taps) of the thread's own patch ----
    // Horizontal pass: every window row once — a thread takes the four rows of its own patch, and the four rows the
    // vertical pass needs above and below the window (rows -2, -1, 64, 65) are dealt one pixel per thread; the rows go
    // through LDS to the threads that need them (a thread blurring rows -2 .. +5 of its patch by itself evaluates every
    // row twice: 320 packed operations instead of 170).
    v2f g[4][4];
    v2f L[4][4];   // the patch itself: the FED steps start from it (taken here, the input window is about to be replaced)
    v2f hown[4][4], hhalo;
    const int halo_row = (tid >> 6) < 2 ? (tid >> 6) - 2 : kFFW - 2 + (tid >> 6), halo_col = tid & 63;   // window coordinates
    {
        const float4* tile = reinterpret_cast<const float4*>(s_buf);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // window row 4 pr + j is input row 4 pr + j + 3; window column X sits at input column X + 4, the patch
            // needs columns 4 pc - 2 .. 4 pc + 5 of the window
            v2f v[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f4v t = lds_chunk(tile + ff_chunk<kFFInC / 2>(4 * pr + j + 3, 2 * pc + 1 + q));
                v[2 * q] = (v2f){t.x, t.y};
                v[2 * q + 1] = (v2f){t.z, t.w};
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                hown[j][o] = lane4_dot_v<5>(v + o, taps.k);
                L[j][o] = v[o + 2];
            }
        }
        v2f v[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) v[i] = s_buf[ff_elem<kFFInC / 2>(halo_row + 3, halo_col + 2 + i)];
        hhalo = lane4_dot_v<5>(v, taps.k);
    }
    // the ring around the window: pixel t of (top row, bottom row, left column, right column), same two passes
    v2f ring[2];
    int ring_x[2], ring_y[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int t = tid + 256 * q;
        int X = 0, Y = 0;
        if (t < kFFW + 2) { X = t - 1; Y = -1; }
        else if (t < 2 * (kFFW + 2)) { X = t - (kFFW + 2) - 1; Y = kFFW; }
        else if (t < 2 * (kFFW + 2) + kFFW) { X = -1; Y = t - 2 * (kFFW + 2); }
        else { X = kFFW; Y = t - 2 * (kFFW + 2) - kFFW; }
        ring_x[q] = X;
        ring_y[q] = Y;
        ring[q] = splat(0.0f);
        if (RING && t < 4 * kFFW + 4) {
            v2f hr[5];
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                v2f v[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) v[i] = s_buf[ff_elem<kFFInC / 2>(Y + 1 + r, X + 2 + i)];   // row Y - 2 + r, column X - 2 + i
                hr[r] = lane4_dot_v<5>(v, taps.k);
            }
            ring[q] = lane4_dot_v<5>(hr, taps.k);
        }
    }
    __syncthreads();   // every thread has read its input: the horizontally blurred rows take the space
    {
        // row Y of the window (-2 .. 65) is row Y + 2 of s_h, 64 columns = 32 chunks in the two-plane layout
        constexpr int HR = kFFW / 2;
        static_assert((kFFW + 4) * kFFW <= kFFIn * kFFInC, "horizontally blurred rows fit in the input window's space");
        float4* h4 = reinterpret_cast<float4*>(s_buf);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h4[ff_chunk<HR>(4 * pr + j + 2, 2 * pc)] = make_float4(hown[j][0].x, hown[j][0].y, hown[j][1].x, hown[j][1].y);
            h4[ff_chunk<HR>(4 * pr + j + 2, 2 * pc + 1)] = make_float4(hown[j][2].x, hown[j][2].y, hown[j][3].x, hown[j][3].y);
        }
        s_buf[ff_elem<HR>(halo_row + 2, halo_col)] = hhalo;
        __syncthreads();
        // vertical pass: rows 4 pr - 2 .. 4 pr + 5 of the patch's own columns; the middle four are the thread's own
        v2f hb[8][4];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (r >= 2 && r < 6) {
#pragma unroll
                for (int o = 0; o < 4; ++o) hb[r][o] = hown[r - 2][o];
            } else {
                const f4v t0 = lds_chunk(h4 + ff_chunk<HR>(4 * pr + r, 2 * pc));
                const f4v t1 = lds_chunk(h4 + ff_chunk<HR>(4 * pr + r, 2 * pc + 1));
                hb[r][0] = (v2f){t0.x, t0.y}; hb[r][1] = (v2f){t0.z, t0.w};
                hb[r][2] = (v2f){t1.x, t1.y}; hb[r][3] = (v2f){t1.z, t1.w};
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const v2f col[5] = {hb[j][o], hb[j + 1][o], hb[j + 2][o], hb[j + 3][o], hb[j + 4][o]};
                g[j][o] = lane4_dot_v<5>(col, taps.k);
            }
    }
    __syncthreads();   // every thread has read its rows: the blurred window takes the space
    // blurred window: pixel (X, Y) of the window is element (row Y + 1, column X + 2) of s_g
    v2f* s_g = s_buf;
    float4* g4 = reinterpret_cast<float4*>(s_buf);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        g4[ff_chunk<kFFGS / 2>(4 * pr + j + 1, 2 * pc + 1)] = make_float4(g[j][0].x, g[j][0].y, g[j][1].x, g[j][1].y);
        g4[ff_chunk<kFFGS / 2>(4 * pr + j + 1, 2 * pc + 2)] = make_float4(g[j][2].x, g[j][2].y, g[j][3].x, g[j][3].y);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if (RING && tid + 256 * q < 4 * kFFW + 4) s_g[ff_elem<kFFGS / 2>(ring_y[q] + 1, ring_x[q] + 2)] = ring[q];
    __syncthreads();
    // positions outside the image take the blurred value at their clamped coordinate (the derivative filters clamp
    // Lsmooth, image.rs:230-236 / :287-300): reads touch in-image positions only, writes out-of-image positions only
    if (wx0 < 1 || wx0 + kFFW + 1 > w || wy0 < 1 || wy0 + kFFW + 1 > h) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const int x = wx0 + 4 * pc + o, y = wy0 + 4 * pr + j;
                const int xc = clampi(x, 0, w - 1), yc = clampi(y, 0, h - 1);
                if (xc != x || yc != y) {
                    const int X = clampi(xc - wx0, -1, kFFW), Y = clampi(yc - wy0, -1, kFFW);
                    s_g[ff_elem<kFFGS / 2>(4 * pr + j + 1, 4 * pc + o + 2)] = s_g[ff_elem<kFFGS / 2>(Y + 1, X + 2)];
                }
            }
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (RING && tid + 256 * q < 4 * kFFW + 4) {
                const int x = wx0 + ring_x[q], y = wy0 + ring_y[q];
                const int xc = clampi(x, 0, w - 1), yc = clampi(y, 0, h - 1);
                if (xc != x || yc != y) {
                    const int X = clampi(xc - wx0, -1, kFFW), Y = clampi(yc - wy0, -1, kFFW);
                    s_g[ff_elem<kFFGS / 2>(ring_y[q] + 1, ring_x[q] + 2)] = s_g[ff_elem<kFFGS / 2>(Y + 1, X + 2)];
                }
            }
        __syncthreads();
    }
    const int x0 = wx0 + 4 * pc, y0 = wy0 + 4 * pr;
    const bool col_in = x0 >= 0 && x0 < w;   // w % 4 == 0: a patch column is entirely inside or outside
    const bool useful = pc >= HP && pc <= 15 - HP && pr >= HP && pr <= 15 - HP && col_in;
    // ---- 2a. conductivity of the patch: simple Scharr (derivatives.rs:3-11) + pm_g2 (nonlinear_diffusion.rs:80) ----
    v2f C[4][4];
    {
        const v2f inverse_k = (v2f){invk[(size_t)fa * 8 + invk_off], invk[(size_t)fb * 8 + invk_off]};
        // per source row: hx = v(+1) - v(-1), hy = (3 v(-1) + 10 v(0)) + 3 v(+1); a three-row window walks down
        v2f hx[3][4], hy[3][4];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            // source row 4 pr - 1 + r, window columns 4 pc - 2 .. 4 pc + 5 (two 16-byte chunks either side of the patch)
            v2f v[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f4v t = lds_chunk(g4 + ff_chunk<kFFGS / 2>(4 * pr + r, 2 * pc + q));
                v[2 * q] = (v2f){t.x, t.y};
                v[2 * q + 1] = (v2f){t.z, t.w};
            }
            const int s = r % 3;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                hx[s][o] = v[o + 3] - v[o + 1];
                hy[s][o] = (splat(3.0f) * v[o + 1] + splat(10.0f) * v[o + 2]) + splat(3.0f) * v[o + 3];
            }
            if (r >= 2) {
                const int j = r - 2, sm = (r - 2) % 3, s0 = (r - 1) % 3, sp = r % 3;
                v2f den[4];
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const v2f lx = (splat(3.0f) * hx[sm][o] + splat(10.0f) * hx[s0][o]) + splat(3.0f) * hx[sp][o];
                    const v2f ly = hy[sp][o] - hy[sm][o];
                    den[o] = splat(1.0f) + inverse_k * (lx * lx + ly * ly);   // the denominator; inverted below
                }
                // (as in k_level_front2: the packed reciprocal unless some denominator of the wave is not finite)
                const v2f dsum = (den[0] + den[1]) + (den[2] + den[3]);   // (as in k_level_front2: one test for the four)
                const bool odd = not_finite(dsum.x) || not_finite(dsum.y);
                if (__any(odd)) {
#pragma unroll
                    for (int o = 0; o < 4; ++o) C[j][o] = splat(1.0f) / den[o];
                } else {
#pragma unroll
                    for (int o = 0; o < 4; ++o) C[j][o] = rcp_pair_finite(den[o]);
                }
            }
        }
    }
    if (WRITE_FLOW && useful) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int y = y0 + j;
            if (y >= h) break;
            const uint32_t o = (uint32_t)(y * w + x0) << 2;
            *reinterpret_cast<float4*>(at_bytes(out_flow + (size_t)fa * fs, o)) = make_float4(C[j][0].x, C[j][1].x, C[j][2].x, C[j][3].x);
            if (has_b)
                *reinterpret_cast<float4*>(at_bytes(out_flow + (size_t)fb * fs, o)) = make_float4(C[j][0].y, C[j][1].y, C[j][2].y, C[j][3].y);
        }
    }
    // ---- 2b. multiscale Scharr first derivatives of the useful patches (derivatives.rs:23-49), taps at -SG, 0, +SG ----
    if (useful) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int y = y0 + j;
            if (y >= h) break;
            // rows Y - SG, Y, Y + SG of the window, columns 4 pc - 4 .. 4 pc + 7
            v2f m[12], z[12], p[12];
            const int Y = 4 * pr + j;
            constexpr int Q0 = (4 - SG) / 2, Q1 = (7 + SG) / 2;   // 16-byte chunks that hold columns 4 - SG .. 7 + SG of the 12
#pragma unroll
            for (int q = Q0; q <= Q1; ++q) {
                const f4v a = lds_chunk(g4 + ff_chunk<kFFGS / 2>(Y - SG + 1, 2 * pc - 1 + q));
                const f4v b = lds_chunk(g4 + ff_chunk<kFFGS / 2>(Y + 1, 2 * pc - 1 + q));
                const f4v c = lds_chunk(g4 + ff_chunk<kFFGS / 2>(Y + SG + 1, 2 * pc - 1 + q));
                m[2 * q] = (v2f){a.x, a.y}; m[2 * q + 1] = (v2f){a.z, a.w};
                z[2 * q] = (v2f){b.x, b.y}; z[2 * q + 1] = (v2f){b.z, b.w};
                p[2 * q] = (v2f){c.x, c.y}; p[2 * q + 1] = (v2f){c.z, c.w};
            }
            v2f rx[4], ry[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const v2f mm = m[4 + o - SG], m0 = m[4 + o], mp = m[4 + o + SG];
                const v2f zm = z[4 + o - SG], zp = z[4 + o + SG];
                const v2f pm = p[4 + o - SG], p0 = p[4 + o], pp = p[4 + o + SG];
                rx[o] = off_combine_sg<SG>(k, mp - mm, zp - zm, pp - pm);
                ry[o] = off_combine_sg<SG>(k, pm, p0, pp) - off_combine_sg<SG>(k, mm, m0, mp);
            }
            const uint32_t o = (uint32_t)(y * w + x0) << 3;
            {
                float4* xy = reinterpret_cast<float4*>(at_bytes(out_xy + (size_t)fa * fs, o));
                xy[0] = make_float4(rx[0].x, ry[0].x, rx[1].x, ry[1].x);
                xy[1] = make_float4(rx[2].x, ry[2].x, rx[3].x, ry[3].x);
            }
            if (has_b) {
                float4* xy = reinterpret_cast<float4*>(at_bytes(out_xy + (size_t)fb * fs, o));
                xy[0] = make_float4(rx[0].y, ry[0].y, rx[1].y, ry[1].y);
                xy[1] = make_float4(rx[2].y, ry[2].y, rx[3].y, ry[3].y);
            }
        }
    }
    // ---- 3. the FED steps, as k_fed_pair ----
    // (k_fed_pair loads zeros outside the image: same here, for the image and the conductivity)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int y = y0 + r;
        const bool in_img = col_in && y >= 0 && y < h;
        if (!in_img) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                L[r][c] = splat(0.0f);
                C[r][c] = splat(0.0f);
            }
        }
    }
    __syncthreads();   // the blurred window is dead: the exchange buffers take the space
    float4* s_top = reinterpret_cast<float4*>(s_buf);            // [256 * 2]  [patch][4 px x 2 frames]: top image rows
    float4* s_vd = s_top + 256 * 2;                              // bottom-edge flows
    float4* s_ct = s_vd + 256 * 2;                               // top rows of C
    const int up = pr > 0 ? tid - 16 : tid, dn = pr < 15 ? tid + 16 : tid;   // block-edge patches are halo
    fed_steps_sums(L, C, s_top, s_ct, s_vd, taus, nsteps, tid, up, dn, x0, y0, w, h);
    if (useful) {
        fed_store_patch(L, out_lt, fa, fb, has_b, fs, w, h, x0, y0);
        if (half_out) fed_store_half(L, half_out, fa, fb, has_b, half_fs, w, h, x0, y0);
    }
}

// ---------------------------------------------------------------------------------------------
// A whole level of a SMALL octave on ONE compute unit: front end + EVERY FED step of the level in one launch, the image in
// registers for the whole diffusion, no halo, no Lflow, one pass over HBM (4 B in, 4 B Lt + 8 B {Lx, Ly} out per pixel).
// lib.rs:230-256 for level i:  Lsmooth = blur(Lt, 1.0);  Lflow = pm_g2(simple Scharr(Lsmooth));  {Lx, Ly} = multiscale
// Scharr(Lsmooth);  then every tau of the level's FED cycle (nonlinear_diffusion.rs:14-58).
//
// The tile kernels above spend the deeper octaves on halos and launches: octave 3 of a 1080p pyramid (240 x 135, 17-29 steps
// per level) takes 13 launches of 64 x 64 windows of which 48 x 48 are useful, re-fetched and re-stored every 8 steps.  Here
// one workgroup holds one frame: a wave is a row of 4-column patches (lane = patch column, 256 pixels a wave), the waves
// stack vertically, and the frame is cut into an UPPER half (image rows [0, S)) and a LOWER half (rows [S, h)), S = waves x R:
// a thread owns an R x 4 patch at the same position of both halves and holds them as {upper, lower} pairs — the packed
// arithmetic of the two-frame kernels (v_pk_mul_f32 / v_pk_add_f32 round each half like the scalar instruction), with the
// two "frames" being the two halves of one image.  Every expression is the two-frame kernels' (k_front_fed's phases, k_fed_pair's
// step); what is new is the seam and the borders:
//   * the halves meet at rows S - 1 | S: the last wave's patch continues in the first wave's LOWER patch.  In the FED steps that
//     is two re-addressed exchange reads (the last wave takes the first wave's top row's LOWER half as the row below its
//     UPPER half; the first wave takes the last wave's UPPER bottom-edge flow as the flow above its LOWER half) — every other
//     edge of the two halves is an image border (a zero step size, as in fed_steps);
//   * the stencil phases (vertical blur pass, simple Scharr, multiscale Scharr) read a plane of {upper, lower} pairs in LDS
//     whose apron (4 rows / columns around the S x w body) and out-of-image body rows are filled by ONE rule, res_fill_apron:
//     every position holds the value at its CLAMPED IMAGE coordinate — across the seam that is the other half's data,
//     beyond the image the reference's edge replication (image.rs:230-236, :287-300);
//   * the horizontal blur pass takes its neighbours' columns from the adjacent lanes (DPP wave shifts), clamped at lanes 0 and
//     w / 4 - 1.
// LDS: the plane ((w + 8) x (S + 8) pairs: 151 KB at 240 x 135) and, once it is dead, the FED exchange buffers in its place.
// Used for levels below the first octave with w % 4 == 0, w <= 256, h <= 2 * 16 * R and a plane that fits (akz_resident_fits).
constexpr int kResMaxSteps = 64;                 // FED steps of one level (level 15 of the default pyramid: 29)
struct FedTausN {
    float half_tau[kResMaxSteps];
};
constexpr int kResLdsBytes = 160 * 1024 - 2048;   // the kernel's static LDS block (a CU has 160 KB)
#ifndef AKZ_RES_R
#define AKZ_RES_R 6        // 12 waves x 6 rows x 2 halves: 144 rows (168 registers a thread at three waves per SIMD)
#define AKZ_RES_WAVES 12
#endif

// plane element (row, col), both apron-inclusive (image row j of a half is row j + 4, column c is col c + 4): 16-byte chunks of
// two columns, even chunks first (ff_chunk's order: the lanes of a wave, which read chunks two apart, touch consecutive words)
__device__ __forceinline__ int res_chunk(int rs, int row, int ci) { return row * rs + ((ci & 1) ? (rs >> 1) : 0) + (ci >> 1); }
__device__ __forceinline__ int res_elem(int rs, int row, int col) { return 2 * res_chunk(rs, row, col >> 1) + (col & 1); }

// Every plane position that is not an in-image body position takes the image's value at its clamped coordinate.
// Upper half (.x) position (j, c) is image pixel (j, c), lower half (.y) position (j, c) is image pixel (S + j, c); an image
// pixel (y, x), clamped to the image, lives in the body at (y, x).x for y < S and at (y - S, x).y otherwise.  Reads touch
// in-image components only, writes out-of-image components only (the caller's barriers sit around the call).
__device__ __forceinline__ void res_fill_apron(v2f* __restrict__ plane, int rs, int w, int h, int S, int tid, int nthreads)
{
    const int PW = w + 8, hb = h - S;
    auto src = [&](int y_img, int c) -> float {
        y_img = clampi(y_img, 0, h - 1);
        const int cc = clampi(c, 0, w - 1);
        const bool lower = y_img >= S;
        const v2f e = plane[res_elem(rs, (lower ? y_img - S : y_img) + 4, cc + 4)];
        return lower ? e.y : e.x;
    };
    // 1: the four apron rows above and below, full width; 2: the four apron columns either side of the body rows;
    // 3: the body rows of the lower half that lie below the image (lower component only)
    const int n1 = 8 * PW, n2 = 8 * S, n3 = (S - hb) * w;
    for (int idx = tid; idx < n1 + n2 + n3; idx += nthreads) {
        int j, c;
        bool only_lower = false;
        if (idx < n1) {
            const int q = idx / PW;
            c = idx - q * PW - 4;
            j = q < 4 ? q - 4 : S + (q - 4);
        } else if (idx < n1 + n2) {
            const int t = idx - n1, e = t & 7;
            j = t >> 3;
            c = e < 4 ? e - 4 : w + (e - 4);
        } else {
            const int t = idx - n1 - n2, q = t / w;
            c = t - q * w;
            j = hb + q;
            only_lower = true;
        }
        const float lo = src(S + j, c);
        if (only_lower) reinterpret_cast<float*>(plane)[2 * res_elem(rs, j + 4, c + 4) + 1] = lo;
        else plane[res_elem(rs, j + 4, c + 4)] = (v2f){src(j, c), lo};
    }
}

// The FED steps of k_fed_pair / fed_steps on R x 4 patches of {upper, lower} pairs (see above).  zb[r]: row r of the LOWER
// patch is the image's last row or lies below it (its downward flow takes a zero step); the upper half has no such row —
// below its last row the lower half begins.  Exchange slots are per thread; `dn` / `up` name the thread below / above, for
// the last / first wave the first / last wave's thread of the same lane (the seam).
template <int R>
__device__ __forceinline__ void fed_steps_halves(v2f (&L)[R][4], v2f (&C)[R][4], float4* __restrict__ s_top,
                                                 float4* __restrict__ s_ct, float4* __restrict__ s_vd, const FedTausN& taus,
                                                 int nsteps, int tid, int up, int dn, bool first_wave, bool last_wave,
                                                 bool z_right, const bool (&zb)[R])
{
    s_ct[tid * 2] = make_float4(C[0][0].x, C[0][0].y, C[0][1].x, C[0][1].y);
    s_ct[tid * 2 + 1] = make_float4(C[0][2].x, C[0][2].y, C[0][3].x, C[0][3].y);
#pragma unroll 1
    for (int t = 0; t < nsteps; ++t) {
        const float htf = taus.half_tau[t];
        const v2f ht = splat(htf);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(C[r][c]));   // (as fed_steps: no sums kept across the steps)
        s_top[tid * 2] = make_float4(L[0][0].x, L[0][0].y, L[0][1].x, L[0][1].y);
        s_top[tid * 2 + 1] = make_float4(L[0][2].x, L[0][2].y, L[0][3].x, L[0][3].y);
        __syncthreads();
        v2f vd[4];
        {
            const float4 a = s_top[dn * 2], b = s_top[dn * 2 + 1], c = s_ct[dn * 2], d = s_ct[dn * 2 + 1];
            v2f Lb[4] = {(v2f){a.x, a.y}, (v2f){a.z, a.w}, (v2f){b.x, b.y}, (v2f){b.z, b.w}};
            v2f Cb[4] = {(v2f){c.x, c.y}, (v2f){c.z, c.w}, (v2f){d.x, d.y}, (v2f){d.z, d.w}};
            if (last_wave) {   // the row below the upper half is the lower half's first row (first wave, lower component);
#pragma unroll                 // below the lower half there is nothing (zero step: any finite value)
                for (int cc = 0; cc < 4; ++cc) {
                    Lb[cc] = splat(Lb[cc].y);
                    Cb[cc] = splat(Cb[cc].y);
                }
            }
            const v2f ht3 = (v2f){htf, zb[R - 1] ? 0.0f : htf};
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) vd[cc] = fed_flow2(ht3, C[R - 1][cc], Cb[cc], L[R - 1][cc], Lb[cc]);
            s_vd[tid * 2] = make_float4(vd[0].x, vd[0].y, vd[1].x, vd[1].y);
            s_vd[tid * 2 + 1] = make_float4(vd[2].x, vd[2].y, vd[3].x, vd[3].y);
        }
        const v2f htr = splat(z_right ? 0.0f : htf);
#pragma unroll
        for (int r = R - 1; r >= 0; --r) {
            const v2f Lr = dpp_row<0x130>(L[r][0]), Cr = dpp_row<0x130>(C[r][0]);   // lane + 1 (wave_shl:1; 0 past lane 63)
            v2f hf[5];
#pragma unroll
            for (int c = 1; c < 4; ++c) hf[c] = fed_flow2(ht, C[r][c - 1], C[r][c], L[r][c - 1], L[r][c]);
            hf[4] = fed_flow2(htr, C[r][3], Cr, L[r][3], Lr);
            hf[0] = dpp_row<0x138>(hf[4]);                                            // lane - 1 (wave_shr:1; +0 into lane 0)
            v2f vu[4];
            if (r > 0) {
                const v2f htu = (v2f){htf, zb[r > 0 ? r - 1 : 0] ? 0.0f : htf};
#pragma unroll
                for (int c = 0; c < 4; ++c) vu[c] = fed_flow2(htu, C[r - (r > 0)][c], C[r][c], L[r - (r > 0)][c], L[r][c]);
            } else {
                __syncthreads();
                const float4 a = s_vd[up * 2], b = s_vd[up * 2 + 1];
                vu[0] = (v2f){a.x, a.y}; vu[1] = (v2f){a.z, a.w}; vu[2] = (v2f){b.x, b.y}; vu[3] = (v2f){b.z, b.w};
                if (first_wave) {   // nothing above the image's first row (+0: the skipped term); above the lower half the
#pragma unroll                      // upper half's last row (last wave, upper component)
                    for (int c = 0; c < 4; ++c) vu[c] = (v2f){0.0f, vu[c].x};
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                L[r][c] = (((L[r][c] + hf[c + 1]) - hf[c]) + vd[c]) - vu[c];   // nonlinear_diffusion.rs:31-52 order
                vd[c] = vu[c];
            }
        }
    }
}

template <int SG, int R, int NWMAX>
__global__ __launch_bounds__(NWMAX * 64) void k_level_resident(const float* __restrict__ in, int w, int h, int S, size_t fs,
                                                               GaussTaps taps, OffK k, FedTausN taus, int nsteps,
                                                               float* __restrict__ out_lt, float2* __restrict__ out_xy,
                                                               const float* __restrict__ invk, int invk_off)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[kResLdsBytes];
    v2f* plane = reinterpret_cast<v2f*>(s_raw);
    float4* p4 = reinterpret_cast<float4*>(s_raw);
    const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), NW = (int)(blockDim.x >> 6);
    const int frame = blockIdx.x;
    const int nl = w >> 2, rs = (w + 8) >> 1;
    const int x0 = 4 * lane, r0 = wv * R;
    const bool act = lane < nl, first_lane = lane == 0, last_lane = lane == nl - 1;
    const float* src = in + (size_t)frame * fs;
    bool vb[R];            // row r of the lower patch lies inside the image (the upper patch always does: S <= h - 1)
#pragma unroll
    for (int r = 0; r < R; ++r) vb[r] = S + r0 + r < h;
    auto load_patch = [&](v2f (&L)[R][4]) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float4 la = make_float4(0.f, 0.f, 0.f, 0.f), lb = la;
            if (act) {
                la = *reinterpret_cast<const float4*>(at_bytes(src, (uint32_t)((r0 + r) * w + x0) << 2));
                if (vb[r]) lb = *reinterpret_cast<const float4*>(at_bytes(src, (uint32_t)((S + r0 + r) * w + x0) << 2));
            }
            L[r][0] = (v2f){la.x, lb.x}; L[r][1] = (v2f){la.y, lb.y}; L[r][2] = (v2f){la.z, lb.z}; L[r][3] = (v2f){la.w, lb.w};
        }
    };
    auto store_plane = [&](const v2f (&V)[R][4]) {   // the patch's rows into the plane body (both halves)
        if (act) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                p4[res_chunk(rs, r0 + r + 4, 2 * lane + 2)] = make_float4(V[r][0].x, V[r][0].y, V[r][1].x, V[r][1].y);
                p4[res_chunk(rs, r0 + r + 4, 2 * lane + 3)] = make_float4(V[r][2].x, V[r][2].y, V[r][3].x, V[r][3].y);
            }
        }
    };
    // ---- 1. Gaussian blur (sigma 1.0, 5 taps): horizontal pass from the neighbouring lanes' registers ----
    v2f G[R][4];
    {
        v2f L[R][4];
        load_patch(L);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            v2f lm2 = dpp_row<0x138>(L[r][2]), lm1 = dpp_row<0x138>(L[r][3]);   // columns x0 - 2, x0 - 1 (lane - 1)
            v2f rp0 = dpp_row<0x130>(L[r][0]), rp1 = dpp_row<0x130>(L[r][1]);   // columns x0 + 4, x0 + 5 (lane + 1)
            if (first_lane) lm2 = lm1 = L[r][0];                                 // clamped columns (image.rs:230-236)
            if (last_lane) rp0 = rp1 = L[r][3];
            const v2f v[8] = {lm2, lm1, L[r][0], L[r][1], L[r][2], L[r][3], rp0, rp1};
#pragma unroll
            for (int o = 0; o < 4; ++o) G[r][o] = lane4_dot_v<5>(v + o, taps.k);
        }
    }
    store_plane(G);
    __syncthreads();
    res_fill_apron(plane, rs, w, h, S, tid, nthreads);
    __syncthreads();
    // vertical pass: rows r0 - 2 .. r0 + R + 1 of the patch's own columns
    if (act) {
        v2f win[5][4];
#pragma unroll
        for (int j = 0; j < R + 4; ++j) {
            const f4v t0 = lds_chunk(p4 + res_chunk(rs, r0 + j + 2, 2 * lane + 2));
            const f4v t1 = lds_chunk(p4 + res_chunk(rs, r0 + j + 2, 2 * lane + 3));
            const int sl = j % 5;
            win[sl][0] = (v2f){t0.x, t0.y}; win[sl][1] = (v2f){t0.z, t0.w};
            win[sl][2] = (v2f){t1.x, t1.y}; win[sl][3] = (v2f){t1.z, t1.w};
            if (j >= 4) {
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const v2f col[5] = {win[(j - 4) % 5][o], win[(j - 3) % 5][o], win[(j - 2) % 5][o], win[(j - 1) % 5][o], win[j % 5][o]};
                    G[j - 4][o] = lane4_dot_v<5>(col, taps.k);
                }
            }
        }
    }
    __syncthreads();   // every thread has read its rows: the blurred image takes the plane
    store_plane(G);
    __syncthreads();
    res_fill_apron(plane, rs, w, h, S, tid, nthreads);
    __syncthreads();
    // ---- 2a. multiscale Scharr first derivatives (derivatives.rs:23-49), taps at -SG, 0, +SG -> HBM ----
    if (act) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            // plane rows Y - SG, Y, Y + SG, columns x0 - 4 .. x0 + 7
            v2f m[12], z[12], p[12];
            const int Y = r0 + j + 4;
            constexpr int Q0 = (4 - SG) / 2, Q1 = (7 + SG) / 2;   // chunks that hold columns 4 - SG .. 7 + SG of the 12
#pragma unroll
            for (int q = Q0; q <= Q1; ++q) {
                const f4v a = lds_chunk(p4 + res_chunk(rs, Y - SG, 2 * lane + q));
                const f4v b = lds_chunk(p4 + res_chunk(rs, Y, 2 * lane + q));
                const f4v c = lds_chunk(p4 + res_chunk(rs, Y + SG, 2 * lane + q));
                m[2 * q] = (v2f){a.x, a.y}; m[2 * q + 1] = (v2f){a.z, a.w};
                z[2 * q] = (v2f){b.x, b.y}; z[2 * q + 1] = (v2f){b.z, b.w};
                p[2 * q] = (v2f){c.x, c.y}; p[2 * q + 1] = (v2f){c.z, c.w};
            }
            v2f rx[4], ry[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const v2f mm = m[4 + o - SG], m0 = m[4 + o], mp = m[4 + o + SG];
                const v2f zm = z[4 + o - SG], zp = z[4 + o + SG];
                const v2f pm = p[4 + o - SG], p0 = p[4 + o], pp = p[4 + o + SG];
                rx[o] = off_combine_sg<SG>(k, mp - mm, zp - zm, pp - pm);
                ry[o] = off_combine_sg<SG>(k, pm, p0, pp) - off_combine_sg<SG>(k, mm, m0, mp);
            }
            {
                float4* xy = reinterpret_cast<float4*>(at_bytes(out_xy + (size_t)frame * fs, (uint32_t)((r0 + j) * w + x0) << 3));
                xy[0] = make_float4(rx[0].x, ry[0].x, rx[1].x, ry[1].x);
                xy[1] = make_float4(rx[2].x, ry[2].x, rx[3].x, ry[3].x);
            }
            if (vb[j]) {
                float4* xy = reinterpret_cast<float4*>(at_bytes(out_xy + (size_t)frame * fs, (uint32_t)((S + r0 + j) * w + x0) << 3));
                xy[0] = make_float4(rx[0].y, ry[0].y, rx[1].y, ry[1].y);
                xy[1] = make_float4(rx[2].y, ry[2].y, rx[3].y, ry[3].y);
            }
        }
    }
    // ---- 2b. conductivity of the patch: simple Scharr (derivatives.rs:3-11) + pm_g2 (nonlinear_diffusion.rs:80) ----
    v2f L[R][4], C[R][4];
    load_patch(L);   // the image again (L2): the blur's copy did not stay in registers through the stencil phases
    {
        const v2f inverse_k = splat(invk[(size_t)frame * 8 + invk_off]);
        v2f hx[3][4], hy[3][4];
#pragma unroll
        for (int r = 0; r < R + 2; ++r) {
            // source row r0 - 1 + r, columns x0 - 2 .. x0 + 5
            v2f v[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f4v t = (f4v){0.f, 0.f, 0.f, 0.f};
                if (act) t = lds_chunk(p4 + res_chunk(rs, r0 + r + 3, 2 * lane + 1 + q));
                v[2 * q] = (v2f){t.x, t.y};
                v[2 * q + 1] = (v2f){t.z, t.w};
            }
            const int s = r % 3;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                hx[s][o] = v[o + 3] - v[o + 1];
                hy[s][o] = (splat(3.0f) * v[o + 1] + splat(10.0f) * v[o + 2]) + splat(3.0f) * v[o + 3];
            }
            if (r >= 2) {
                const int j = r - 2, sm = (r - 2) % 3, s0 = (r - 1) % 3, sp = r % 3;
                v2f den[4];
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const v2f lx = (splat(3.0f) * hx[sm][o] + splat(10.0f) * hx[s0][o]) + splat(3.0f) * hx[sp][o];
                    const v2f ly = hy[sp][o] - hy[sm][o];
                    den[o] = splat(1.0f) + inverse_k * (lx * lx + ly * ly);
                }
                // (as in k_level_front2: the packed reciprocal unless some denominator of the wave is not finite)
                const v2f dsum = (den[0] + den[1]) + (den[2] + den[3]);
                const bool odd = not_finite(dsum.x) || not_finite(dsum.y);
                if (__any(odd)) {
#pragma unroll
                    for (int o = 0; o < 4; ++o) C[j][o] = splat(1.0f) / den[o];
                } else {
#pragma unroll
                    for (int o = 0; o < 4; ++o) C[j][o] = rcp_pair_finite(den[o]);
                }
            }
        }
    }
    // (k_fed_pair loads zeros outside the image: the lower patch's rows below the image, and the lanes right of it)
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (!act) C[r][c] = splat(0.0f);
            else if (!vb[r]) C[r][c] = (v2f){C[r][c].x, 0.0f};
        }
    // ---- 3. every FED step of the level ----
    __syncthreads();   // the plane is dead: the exchange buffers take its place
    float4* s_top = p4;                           // [threads * 2]: top image rows
    float4* s_vd = s_top + NWMAX * 64 * 2;        // bottom-edge flows
    float4* s_ct = s_vd + NWMAX * 64 * 2;         // top rows of C
    static_assert(3 * NWMAX * 64 * 2 * sizeof(float4) <= kResLdsBytes, "FED exchange buffers fit");
    const bool first_wave = wv == 0, last_wave = wv == NW - 1;
    const int up = first_wave ? (NW - 1) * 64 + lane : tid - 64, dn = last_wave ? lane : tid + 64;
    bool zb[R];
#pragma unroll
    for (int r = 0; r < R; ++r) zb[r] = S + r0 + r >= h - 1;
    fed_steps_halves<R>(L, C, s_top, s_ct, s_vd, taus, nsteps, tid, up, dn, first_wave, last_wave, x0 + 4 >= w, zb);
    if (act) {
        float* dst = out_lt + (size_t)frame * fs;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            *reinterpret_cast<float4*>(at_bytes(dst, (uint32_t)((r0 + r) * w + x0) << 2)) = make_float4(L[r][0].x, L[r][1].x, L[r][2].x, L[r][3].x);
            if (vb[r])
                *reinterpret_cast<float4*>(at_bytes(dst, (uint32_t)((S + r0 + r) * w + x0) << 2)) = make_float4(L[r][0].y, L[r][1].y, L[r][2].y, L[r][3].y);
        }
    }
}

// Multiscale Scharr (derivatives.rs:23-79) evaluated sparsely: of the 2*sigma+1 taps only
// {0, sigma, 2*sigma} are non-zero, and the reference's 4-lane summation puts them in lanes
// {0, sigma&3, (2*sigma)&3}.  With the sequential lane reduce that collapses to
//   main kernel [-1,0..0,1]:      v(+s) - v(-s)                          (every sigma)
//   off  kernel [n,0..,m,..0,n]:  (v(-s)*n + v(+s)*n) + v(0)*m           (sigma with 2*sigma % 4 != 0 ... see below)
//                                 (v(-s)*n + v(0)*m) + v(+s)*n           (sigma % 4 == 0: all three taps share lane 0)
// sigma == 2: taps 0,2,4 -> lanes 0,2,0: lane0 = v(+s)n + v(-s)n, lane2 = v(0)m -> (lane0 + lane2)
// sigma == 3: taps 0,3,6 -> lanes 0,3,2: ((v(-s)n + 0) + v(+s)n) + v(0)m
// sigma == 1 is the unnormalised simple Scharr [3,10,3]: (3 v(-1) + 10 v(0)) + 3 v(+1).
// Lx = V_off(H_main(Lsmooth)), Ly = V_main(H_off(Lsmooth)) — detector_response.rs:63-64.
__global__ __launch_bounds__(256) void k_deriv_first(const float* __restrict__ sm, float2* __restrict__ Lxy, int w,
                                                     int h, size_t fs, int s, OffK k)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const float* S = sm + (size_t)blockIdx.z * fs;
    int xm = clampi(x - s, 0, w - 1), xp = clampi(x + s, 0, w - 1);
    size_t rm = (size_t)clampi(y - s, 0, h - 1) * w, r0 = (size_t)y * w, rp = (size_t)clampi(y + s, 0, h - 1) * w;
    float mm = S[rm + xm], m0 = S[rm + x], mp = S[rm + xp];
    float zm = S[r0 + xm], zp = S[r0 + xp];
    float pm = S[rp + xm], p0 = S[rp + x], pp = S[rp + xp];
    // H_main rows
    float hm_m = mp - mm, hm_0 = zp - zm, hm_p = pp - pm;
    float lx = off_combine(k, hm_m, hm_0, hm_p);
    // H_off rows y-s and y+s
    float ho_m = off_combine(k, mm, m0, mp);
    float ho_p = off_combine(k, pm, p0, pp);
    float ly = ho_p - ho_m;
    Lxy[(size_t)blockIdx.z * fs + r0 + x] = make_float2(lx, ly);
}

// Lxx = scharr_h(Lx) = V_off(H_main Lx); Lyy = scharr_v(Ly) = V_main(H_off Ly); Lxy = scharr_v(Lx) =
// V_main(H_off Lx) — detector_response.rs:65-67; Ldet = (Lxx*Lyy - Lxy*Lxy) * sigma^4 — :46.
__global__ __launch_bounds__(256) void k_deriv_second(const float2* __restrict__ Lxy, float* __restrict__ Ldet, int w,
                                                      int h, size_t fs, int s, OffK k, float sigma_quat)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const float2* D = Lxy + (size_t)blockIdx.z * fs;
    int xm = clampi(x - s, 0, w - 1), xp = clampi(x + s, 0, w - 1);
    size_t rm = (size_t)clampi(y - s, 0, h - 1) * w, r0 = (size_t)y * w, rp = (size_t)clampi(y + s, 0, h - 1) * w;
    // the 3x3 stencil (spacing s) minus its centre; .x = Lx, .y = Ly
    float2 mm = D[rm + xm], m0 = D[rm + x], mp = D[rm + xp];
    float2 zm = D[r0 + xm], zp = D[r0 + xp];
    float2 pm = D[rp + xm], p0 = D[rp + x], pp = D[rp + xp];
    float lxx = off_combine(k, mp.x - mm.x, zp.x - zm.x, pp.x - pm.x);
    float lxy = off_combine(k, pm.x, p0.x, pp.x) - off_combine(k, mm.x, m0.x, mp.x);
    float lyy = off_combine(k, pm.y, p0.y, pp.y) - off_combine(k, mm.y, m0.y, mp.y);
    Ldet[(size_t)blockIdx.z * fs + r0 + x] = (lxx * lyy - lxy * lxy) * sigma_quat;
}

// Fused second-order pass + extrema candidates (detector_response.rs:65-67,:46 and
// scale_space_extrema.rs:34-60,96-104).  A block produces a 64x32 tile of Ldet plus a one-pixel ring
// (kept in LDS only), writes the tile, and tests every interior pixel against the threshold and its 8
// neighbours straight from LDS — Ldet is never re-read from HBM for detection.  For the derivative scales the
// AKAZE path uses (sigma 2..4) the {Lx,Ly} tile and its halo are staged in LDS with coalesced 8-byte row reads
// first, so each input pixel crosses the L1 once instead of eight times.  Candidates that also pass
// the border test are appended to the frame's per-level list in arbitrary order; k_cand_sort restores
// the reference's raster order afterwards.
// A candidate as the determinant kernels append it (arbitrary order): position, response and the eight
// determinant values around it, which is all do_subpixel_refinement (scale_space_extrema.rs:297-347) ever reads
// of Ldet.  With them in the list the Ldet planes need not go to HBM at all (10 % of the scale space's traffic);
// they are still written when the parity taps are on (AKZ_KEEP_ALL=1).
struct CandU {
    uint32_t xy;     // x | y << 16
    float v;         // Ldet at the pixel
    float nb[8];     // (x-1,y-1) (x,y-1) (x+1,y-1) (x-1,y) (x+1,y) (x-1,y+1) (x,y+1) (x+1,y+1)
};

struct CandParams {
    float thr;         // detector_threshold as f32
    float border;      // smax * sigma_size  (scale_space_extrema.rs:97-100)
    uint32_t level;
    uint32_t cap;      // per-level capacity of the candidate list
    // the pixels that are interior (:50) AND pass the border test (:96-104), as inclusive integer ranges: the test is
    // monotone in x and in y, so the host evaluates the reference's f32 expressions once per column / row (akz_plan.cpp)
    int x_lo, x_hi, y_lo, y_hi;
};


template <int SG>  // SG = deriv_sigma (2, 3 or 4); 0 = generic (direct global gathers)
__global__ __launch_bounds__(256) void k_deriv_second_cand(const float2* __restrict__ Lxy, float* __restrict__ Ldet,
                                                           int w, int h, size_t fs, int s, OffK k, float sigma_quat,
                                                           CandParams cp, CandU* __restrict__ cand,
                                                           uint32_t* __restrict__ ncand, uint32_t* __restrict__ err)
{
    constexpr int TW = 64, TH = 32, GW = TW + 2, GH = TH + 2;
    constexpr int HL = (SG > 0 ? SG : 1) + 1;            // halo of the staged {Lx,Ly} tile
    constexpr int SW = TW + 2 * HL, SH = TH + 2 * HL;
    __shared__ float s_d[GH * GW];
    __shared__ float2 s_xy[SG > 0 ? SH * SW : 1];
    const int frame = blockIdx.z;
    const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
    const float2* D = Lxy + (size_t)frame * fs;
    if (SG > 0) {
        // stage {Lx,Ly} at CLAMPED coordinates: consecutive lanes read consecutive 8-byte pixels of a row
        for (int idx = threadIdx.x; idx < SH * SW; idx += 256) {
            int q = idx / SW, p = idx - q * SW;
            int cx = clampi(tx0 - HL + p, 0, w - 1), cy = clampi(ty0 - HL + q, 0, h - 1);
            s_xy[idx] = D[(size_t)cy * w + cx];
        }
        __syncthreads();
    }
    for (int idx = threadIdx.x; idx < GH * GW; idx += 256) {
        int q = idx / GW, p = idx - q * GW;
        int x = tx0 - 1 + p, y = ty0 - 1 + q;
        float v = 0.0f;
        if (x >= 0 && x < w && y >= 0 && y < h) {
            float2 mm, m0, mp, zm, zp, pm, p0, pp;
            if (SG > 0) {
                // the +-s taps at clamped coordinates; a staged position holds the value of its clamped coordinate
                int lxm = clampi(x - SG, 0, w - 1) - (tx0 - HL), lx0 = x - (tx0 - HL), lxp = clampi(x + SG, 0, w - 1) - (tx0 - HL);
                int lym = (clampi(y - SG, 0, h - 1) - (ty0 - HL)) * SW, ly0 = (y - (ty0 - HL)) * SW,
                    lyp = (clampi(y + SG, 0, h - 1) - (ty0 - HL)) * SW;
                mm = s_xy[lym + lxm]; m0 = s_xy[lym + lx0]; mp = s_xy[lym + lxp];
                zm = s_xy[ly0 + lxm]; zp = s_xy[ly0 + lxp];
                pm = s_xy[lyp + lxm]; p0 = s_xy[lyp + lx0]; pp = s_xy[lyp + lxp];
            } else {
                int xm = clampi(x - s, 0, w - 1), xp = clampi(x + s, 0, w - 1);
                size_t rm = (size_t)clampi(y - s, 0, h - 1) * w, r0 = (size_t)y * w, rp = (size_t)clampi(y + s, 0, h - 1) * w;
                mm = D[rm + xm]; m0 = D[rm + x]; mp = D[rm + xp];
                zm = D[r0 + xm]; zp = D[r0 + xp];
                pm = D[rp + xm]; p0 = D[rp + x]; pp = D[rp + xp];
            }
            float lxx = off_combine(k, mp.x - mm.x, zp.x - zm.x, pp.x - pm.x);
            float lxy = off_combine(k, pm.x, p0.x, pp.x) - off_combine(k, mm.x, m0.x, mp.x);
            float lyy = off_combine(k, pm.y, p0.y, pp.y) - off_combine(k, mm.y, m0.y, mp.y);
            v = (lxx * lyy - lxy * lxy) * sigma_quat;
            if (Ldet && p >= 1 && p <= TW && q >= 1 && q <= TH) Ldet[(size_t)frame * fs + (size_t)y * w + x] = v;
        }
        s_d[idx] = v;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < TH * TW; idx += 256) {
        int q = idx / TW, p = idx - q * TW;
        int x = tx0 + p, y = ty0 + q;
        if (x < 1 || x > w - 2 || y < 1 || y > h - 2) continue;  // interior pixels only (:50)
        const float* c = &s_d[(q + 1) * GW + (p + 1)];
        float v = c[0];
        bool is_cand = v > cp.thr && v > c[-GW - 1] && v > c[-GW] && v > c[-GW + 1] && v > c[-1] && v > c[1] &&
                       v > c[GW - 1] && v > c[GW] && v > c[GW + 1];
        if (!is_cand) continue;
        // border test (:96-104); a candidate failing it can neither push nor replace (:105)
        const float px = (float)x, py = (float)y;
        float left_x = roundf(px - cp.border) - 1.0f;
        float right_x = roundf(px + cp.border) + 1.0f;
        float up_y = roundf(py - cp.border) - 1.0f;
        float down_y = roundf(py + cp.border) + 1.0f;
        bool is_out = left_x < 0.0f || right_x >= (float)w || up_y < 0.0f || down_y >= (float)h;
        if (is_out) continue;
        uint32_t slot = atomicAdd(&ncand[(size_t)frame * kAkzMaxLevels + cp.level], 1u);
        if (slot < cp.cap) {
            CandU cu = {(uint32_t)x | ((uint32_t)y << 16), v,
                        {c[-GW - 1], c[-GW], c[-GW + 1], c[-1], c[1], c[GW - 1], c[GW], c[GW + 1]}};
            cand[((size_t)frame * kAkzMaxLevels + cp.level) * cp.cap + slot] = cu;
        } else {
            *err = 1u;
        }
    }
}

// k_deriv_second_cand for two frames per block (w % 4 == 0), same organisation as k_level_front2: the
// {Lx, Ly} tiles of both frames are staged as two split-plane v2f tiles (X and Y), every thread produces
// 4-column strips of the determinant from registers with packed arithmetic, and the candidate test reads the
// determinant tile (plus its one-pixel ring) from LDS.  Staged positions hold the value at their clamped
// coordinate, so the +-SG taps need no clamping of their own.  12 output rows per block: 18 strips x 14 ring
// rows = 252 strips for 256 threads.
template <int SG, int TH, int NT>
__global__ __launch_bounds__(NT) void k_deriv_second_cand2(const float2* __restrict__ Lxy, float* __restrict__ Ldet,
                                                            int w, int h, size_t fs, int n, OffK k, float sigma_quat,
                                                            CandParams cp, CandU* __restrict__ cand,
                                                            uint32_t* __restrict__ ncand, uint32_t* __restrict__ err)
{
    constexpr int TW = 64;
    constexpr int CS = TW + 16, RS = TH + 2 + 2 * SG;    // staged tiles: x = tx0 - 8 + col, y = ty0 - 1 - SG + row
    constexpr int CG = TW + 8, RG = TH + 2;              // determinant tile: x = tx0 - 4 + col, y = ty0 - 1 + row
    __shared__ __attribute__((aligned(16))) v2f s_x[RS * CS];
    __shared__ __attribute__((aligned(16))) v2f s_y[RS * CS];
    __shared__ __attribute__((aligned(16))) v2f s_d[RG * CG];
    const uint3 tile = xcd_tile(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z));
    const int fa = 2 * (int)tile.z;
    const bool has_b = fa + 1 < n;
    const int fb = has_b ? fa + 1 : fa;
    const int tx0 = (int)tile.x * TW, ty0 = (int)tile.y * TH;
    const int tid = threadIdx.x;
    const float2* Da = Lxy + (size_t)fa * fs;
    const float2* Db = Lxy + (size_t)fb * fs;
    if (tx0 >= 8 && tx0 + TW + 8 <= w) {
        // item = (row, 4-pixel group g): two 16-byte loads per frame, one 16-byte chunk per plane and tile —
        // consecutive lanes store consecutive chunks of a plane (a pixel-pair item alternates between the planes
        // and its stores collide on the banks)
        for (int idx = tid; idx < RS * (CS / 4); idx += NT) {
            int r, g;
            strip_item<CS / 4, 0>(idx, RS, r, g);
            int cy = clampi(ty0 - 1 - SG + r, 0, h - 1);
            size_t o = (size_t)cy * w + (tx0 - 8 + 4 * g);
            float4 a0 = *reinterpret_cast<const float4*>(Da + o), a1 = *reinterpret_cast<const float4*>(Da + o + 2);
            float4 b0 = *reinterpret_cast<const float4*>(Db + o), b1 = *reinterpret_cast<const float4*>(Db + o + 2);
            const int chunk = r * (CS / 2) + g;
            reinterpret_cast<float4*>(s_x)[chunk] = make_float4(a0.x, b0.x, a0.z, b0.z);
            reinterpret_cast<float4*>(s_x)[chunk + CS / 4] = make_float4(a1.x, b1.x, a1.z, b1.z);
            reinterpret_cast<float4*>(s_y)[chunk] = make_float4(a0.y, b0.y, a0.w, b0.w);
            reinterpret_cast<float4*>(s_y)[chunk + CS / 4] = make_float4(a1.y, b1.y, a1.w, b1.w);
        }
    } else {
        for (int idx = tid; idx < RS * CS; idx += NT) {
            int r = idx / CS, p = idx - r * CS;
            int cx = clampi(tx0 - 8 + p, 0, w - 1), cy = clampi(ty0 - 1 - SG + r, 0, h - 1);
            size_t o = (size_t)cy * w + cx;
            float2 a = Da[o], b = Db[o];
            s_x[r * CS + tile_col<CS>(p)] = (v2f){a.x, b.x};
            s_y[r * CS + tile_col<CS>(p)] = (v2f){a.y, b.y};
        }
    }
    __syncthreads();
    const float4* x4 = reinterpret_cast<const float4*>(s_x);
    const float4* y4 = reinterpret_cast<const float4*>(s_y);
    for (int idx = tid; idx < RG * (CG / 4); idx += NT) {
        int q, c;
        strip_item<CG / 4, (CS / 2) & 15>(idx, RG, q, c);
        v2f xm[12], xz[12], xp[12], ym[12], yp[12], det[4];
        lds_read12<CS, 4 - SG, 7 + SG>(x4, q, c, xm);
        lds_read12<CS, 4 - SG, 7 + SG>(x4, q + SG, c, xz);
        lds_read12<CS, 4 - SG, 7 + SG>(x4, q + 2 * SG, c, xp);
        lds_read12<CS, 4 - SG, 7 + SG>(y4, q, c, ym);
        lds_read12<CS, 4 - SG, 7 + SG>(y4, q + 2 * SG, c, yp);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int jm = 4 + o - SG, j0 = 4 + o, jp = 4 + o + SG;
            v2f lxx = off_combine_sg<SG>(k, xm[jp] - xm[jm], xz[jp] - xz[jm], xp[jp] - xp[jm]);
            v2f lxy = off_combine_sg<SG>(k, xp[jm], xp[j0], xp[jp]) - off_combine_sg<SG>(k, xm[jm], xm[j0], xm[jp]);
            v2f lyy = off_combine_sg<SG>(k, yp[jm], yp[j0], yp[jp]) - off_combine_sg<SG>(k, ym[jm], ym[j0], ym[jp]);
            det[o] = (lxx * lyy - lxy * lxy) * splat(sigma_quat);
        }
        lds_write4<CG>(&s_d[q * CG], c, det[0], det[1], det[2], det[3]);
        const int x0 = tx0 - 4 + 4 * c, y = ty0 - 1 + q;
        if (Ldet && c >= 1 && c <= TW / 4 && q >= 1 && q <= TH && x0 < w && y < h) {
            const size_t pix = (size_t)y * w + x0;
            *reinterpret_cast<float4*>(Ldet + (size_t)fa * fs + pix) = make_float4(det[0].x, det[1].x, det[2].x, det[3].x);
            if (has_b)
                *reinterpret_cast<float4*>(Ldet + (size_t)fb * fs + pix) =
                    make_float4(det[0].y, det[1].y, det[2].y, det[3].y);
        }
    }
    __syncthreads();
    const float4* d4 = reinterpret_cast<const float4*>(s_d);
    for (int idx = tid; idx < TH * (TW / 4); idx += NT) {
        int q, c;
        strip_item<TW / 4, (CG / 2) & 15>(idx, TH, q, c);
        const int x0 = tx0 + 4 * c, y = ty0 + q;
        if (x0 >= w || y < 1 || y > h - 2) continue;          // interior pixels only (:50)
        v2f m[12], z[12], p[12];
        lds_read12<CG, 4, 7>(d4, q + 1, c, z);                // v[4 + o] is pixel o of the strip (ring tile x = tx0 - 4 + col)
        // most of an image is below the detector threshold: the 3x3 comparison runs only for waves that hold
        // at least one pixel above it
        bool above = false;
#pragma unroll
        for (int o = 0; o < 4; ++o) above |= z[4 + o].x > cp.thr || z[4 + o].y > cp.thr;
        if (!__any(above)) continue;
        lds_read12<CG, 3, 8>(d4, q, c, m);
        lds_read12<CG, 3, 8>(d4, q + 1, c, z);
        lds_read12<CG, 3, 8>(d4, q + 2, c, p);
        // branch-free extremum test of the strip's 8 pixel-frames; candidates are rare, so the (divergent) append
        // path below runs for few strips
        uint32_t hits = 0u;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const float v = z[4 + o][f];
                // three-operand groups: four v_max3_f32 per pixel-frame
                const float n0 = fmaxf(fmaxf(m[3 + o][f], m[4 + o][f]), m[5 + o][f]);
                const float n1 = fmaxf(fmaxf(p[3 + o][f], p[4 + o][f]), p[5 + o][f]);
                const float nb = fmaxf(fmaxf(n0, n1), fmaxf(z[3 + o][f], z[5 + o][f]));
                hits |= (v > cp.thr && v > nb ? 1u : 0u) << (2 * o + f);
            }
        }
        if (!has_b) hits &= 0x55u;
        while (hits) {
            const int bit = __ffs((int)hits) - 1;
            hits &= hits - 1u;
            const int o = bit >> 1, f = bit & 1;
            const int x = x0 + o;
            if (x < 1 || x > w - 2) continue;
            // rows of the 3x3 neighbourhood, columns 3 + o .. 5 + o, without dynamic register indexing
            float r3[3][3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float a = m[3 + j][f], b = z[3 + j][f], d = p[3 + j][f];
                a = o == 1 ? m[4 + j][f] : a; b = o == 1 ? z[4 + j][f] : b; d = o == 1 ? p[4 + j][f] : d;
                a = o == 2 ? m[5 + j][f] : a; b = o == 2 ? z[5 + j][f] : b; d = o == 2 ? p[5 + j][f] : d;
                a = o == 3 ? m[6 + j][f] : a; b = o == 3 ? z[6 + j][f] : b; d = o == 3 ? p[6 + j][f] : d;
                r3[0][j] = a;
                r3[1][j] = b;
                r3[2][j] = d;
            }
            const float v = r3[1][1];
            // border test (:96-104); a candidate failing it can neither push nor replace (:105)
            const float px = (float)x, py = (float)y;
            float left_x = roundf(px - cp.border) - 1.0f;
            float right_x = roundf(px + cp.border) + 1.0f;
            float up_y = roundf(py - cp.border) - 1.0f;
            float down_y = roundf(py + cp.border) + 1.0f;
            bool is_out = left_x < 0.0f || right_x >= (float)w || up_y < 0.0f || down_y >= (float)h;
            if (is_out) continue;
            const size_t list = (size_t)(f ? fb : fa) * kAkzMaxLevels + cp.level;
            uint32_t slot = atomicAdd(&ncand[list], 1u);
            if (slot < cp.cap) {
                CandU cu = {(uint32_t)x | ((uint32_t)y << 16), v,
                            {r3[0][0], r3[0][1], r3[0][2], r3[1][0], r3[1][2], r3[2][0], r3[2][1], r3[2][2]}};
                cand[list * cp.cap + slot] = cu;
            } else {
                *err = 1u;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Streaming form of the second-order pass (detector_response.rs:65-67,:46 + scale_space_extrema.rs:34-60,96-104).
// The tile kernels above re-stage {Lx,Ly} rows for every 12-row tile (a 1.5-1.8x halo), evaluate each horizontal
// partial once per OUTPUT row that uses it (three times for H_main(Lx), twice for the H_off rows), park at block
// barriers and append every candidate with its own atomic (thousands of returns on one counter per launch).
// Here a WAVE walks down the rows of a band of 128 columns of one frame (two columns per lane) with NO shared memory
// in its loop:
//   * every {Lx,Ly} row is read once (per pixel, at clamped coordinates: any width, borders included — a position
//     outside the image holds the value at its clamped coordinate, which is the reference's edge replication);
//   * the neighbouring columns come from the adjacent lanes by DPP wave shifts (the lanes at the ends of the wave
//     receive zeros: they are the band's halo and produce nothing);
//   * the horizontal partials of the row
//         hm = Lx(x+s) - Lx(x-s)                       (H_main of scharr_horizontal(Lx),     -> Lxx)
//         ho = off(P(x-s), P(x), P(x+s)), P = {Lx,Ly}  (H_off  of scharr_vertical(Lx / Ly), -> Lxy, Lyy; one packed chain)
//     are computed ONCE and kept in a register ring of the last 2s+1 rows; the loop is unrolled by the ring length,
//     so every ring index is a compile-time constant;
//   * determinant row r comes out of ring rows r-s, r, r+s in exactly the reference's operation order:
//         Lxx = off(hm[r-s], hm[r], hm[r+s]),  {Lxy, Lyy} = ho[r+s] - ho[r-s],  Ldet = (Lxx*Lyy - Lxy*Lxy) * s^4;
//   * the last three determinant rows stay in registers with their horizontal 3-maxima, so the 3x3 extremum test of
//     row r-1 is three max operations and two compares per pixel (strictly greater than all eight neighbours ==
//     strictly greater than their maximum);
//   * candidates (rare) collect in a small LDS buffer private to the wave and reach the frame's list with one atomic
//     per flush.
template <int SG>
__device__ __forceinline__ float off_combine_s(const OffK k, float a, float b, float c)
{
    constexpr int mode = offk_mode(SG);
    const float pa = a * k.n + 0.0f;
    if constexpr (mode == 0) return (pa + c * k.n) + b * k.m;
    else if constexpr (mode == 1) return (pa + b * k.m) + c * k.n;
    else if constexpr (mode == 2) return pa + (c * k.n + b * k.m);
    else if constexpr (mode == 3) return __builtin_fmaf(c, k.n, pa) + b * k.m;
    else return __builtin_fmaf(c, k.n, __builtin_fmaf(b, k.m, pa));
}

constexpr int det_stream_halo(int SG) { return (SG + 2) & ~1; }                    // columns each side: even, >= SG + 1
constexpr int det_stream_band(int SG) { return 128 - 2 * det_stream_halo(SG); }     // useful columns per wave

// value of lane (lane - N) [N > 0] or (lane + |N|) [N < 0] of the WAVE, 0 beyond its ends: DPP wave_shr:1 /
// wave_shl:1 (GFX9 encodings 0x138 / 0x130, one lane per instruction; two hops for |N| = 2)
template <int N>
__device__ __forceinline__ float dpp_shift(float v)
{
    static_assert(N >= -2 && N <= 2 && N != 0, "one or two lanes");
    constexpr int ctrl = N > 0 ? 0x138 : 0x130;
    int x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, true);
    if (N == 2 || N == -2) x = __builtin_amdgcn_update_dpp(0, x, ctrl, 0xF, 0xF, true);
    return __int_as_float(x);
}

constexpr int kCandBuf = 64;   // candidates a wave collects before it appends them with one atomic

// append the n buffered candidates of a wave to the (frame, level) list (n is wave-uniform)
__device__ __forceinline__ void cand_flush(const CandU* buf, int n, int lane, CandU* __restrict__ cand,
                                           uint32_t* __restrict__ ncand, size_t list, uint32_t cap, uint32_t* __restrict__ err)
{
    if (n == 0) return;
    __builtin_amdgcn_wave_barrier();
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&ncand[list], (uint32_t)n);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (base + (uint32_t)n > cap) *err = 1u;
    // 10 dwords per record, copied dword by dword: consecutive lanes write consecutive addresses
    const uint32_t* src = reinterpret_cast<const uint32_t*>(buf);
    uint32_t* dst = reinterpret_cast<uint32_t*>(cand + list * cap + base);
    const int room = base < cap ? (int)min((uint32_t)n, cap - base) : 0;
    for (int i = lane; i < room * 10; i += 64) dst[i] = src[i];
    __builtin_amdgcn_wave_barrier();
}

template <int SG, bool WRITE_DET>
__global__ __launch_bounds__(256) void k_det_stream(const float2* __restrict__ Lxy, float* __restrict__ Ldet, int w, int h,
                                                    size_t fs, OffK k, float sigma_quat, CandParams cp,
                                                    CandU* __restrict__ cand, uint32_t* __restrict__ ncand,
                                                    uint32_t* __restrict__ err, int nbands, int seg_rows)
{
    constexpr int NR = 2 * SG + 1;                 // ring length = unroll factor
    constexpr int KC = (SG + 1) / 2;               // neighbour lanes (2 pixels each) needed on either side
    constexpr int HB = det_stream_halo(SG), BW = det_stream_band(SG);
    constexpr int NP = 2 * (2 * KC + 1);           // pixels in a lane's window: offsets -2 KC .. 2 KC + 1
    constexpr int PF = 2;                          // rows in flight (deeper prefetch measured no different)
    __shared__ __attribute__((aligned(16))) CandU s_cb[4][kCandBuf];        // the wave's pending candidates
    int nbuf = 0;                                                            // wave-uniform
    // the wave index is wave-uniform: tell the compiler, so the whole row walk (row numbers, clamps, row base
    // addresses, loop control) lives in SGPRs and on the scalar unit
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    // frames in DESCENDING order: the kernel runs right behind the front end that wrote this level's {Lx,Ly} frame by frame in
    // ascending order, so the last frames written are what the 256 MB memory-side cache still holds — they are read first
    // (the 4.2 GB of a full-size level do not fit; 9 117-9 147 vs 9 072-9 082 frames/s, three pairs on one box)
    const int frame = (int)(gridDim.y - 1u - blockIdx.y);
    const int item = (int)blockIdx.x * 4 + wv;
    const int band = item % nbands, seg = item / nbands;
    const int ys = seg * seg_rows;
    if (ys >= h) return;                            // whole wave (no block barrier anywhere)
    const int ye = min(ys + seg_rows, h);
    // the lane's columns x0, x0 + 1 and the useful columns [bx, ux_hi) of the band
    const int bx = band * BW;
    const int x0 = bx - HB + 2 * lane;
    const int ux_hi = min(bx + BW, w);
    const float2* D = Lxy + (size_t)frame * fs;
    const int cx0 = clampi(x0, 0, w - 1), cx1 = clampi(x0 + 1, 0, w - 1);
    const size_t list = (size_t)frame * kAkzMaxLevels + cp.level;
    auto load_row = [&](int y) {
        const float2* r = D + (size_t)clampi(y, 0, h - 1) * w;     // scalar base + per-lane 32-bit offsets
        const float2 a = r[cx0], b = r[cx1];
        return make_float4(a.x, a.y, b.x, b.y);
    };
    float hm[NR][2];
    v2f ho[NR][2];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        hm[i][0] = hm[i][1] = 0.0f;
        ho[i][0] = ho[i][1] = splat(0.0f);
    }
    // determinant rows ys-1 .. ye are needed (the candidate rows and their 3x3 ring): input rows ys-1-SG .. ye+SG
    const int y_first = ys - 1 - SG, y_last = ye + SG;
    // candidate rows [rc_lo, rc_hi) and columns: interior (:50) and inside the border test (:96-104; a candidate
    // failing it can neither push nor replace, :105) — integer ranges from the host (CandParams)
    const int rc_lo = max(ys, cp.y_lo), rc_hi = min(ye, cp.y_hi + 1);
    const bool in0 = x0 >= bx && x0 < ux_hi, in1 = x0 + 1 >= bx && x0 + 1 < ux_hi;   // columns this lane answers for
    const bool useful0 = in0 && x0 >= cp.x_lo && x0 <= cp.x_hi, useful1 = in1 && x0 + 1 >= cp.x_lo && x0 + 1 <= cp.x_hi;
    // determinant rows r-2 (A), r-1 (B) of the lane's two pixels, the 3-maxima of row A and of row B, and the
    // side maximum (left, right) of row B
    float dA[2] = {0.f, 0.f}, dB[2] = {0.f, 0.f}, h3A[2] = {0.f, 0.f}, h3B[2] = {0.f, 0.f}, s2B[2] = {0.f, 0.f};
    float4 nx[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) nx[i] = load_row(y_first + i);
    for (int base = y_first; base <= y_last; base += NR) {
#pragma unroll
        for (int kk = 0; kk < NR; ++kk) {
            const int yi = base + kk;
            const float4 cur = nx[0];
#pragma unroll
            for (int i = 0; i + 1 < PF; ++i) nx[i] = nx[i + 1];
            nx[PF - 1] = load_row(yi + PF);
            // the row's window: pixels -2 KC .. 2 KC + 1 relative to x0, from the neighbouring lanes
            v2f P[NP];
            {
                float4 t[2 * KC + 1];
                t[KC] = cur;
                t[KC - 1] = make_float4(dpp_shift<1>(cur.x), dpp_shift<1>(cur.y), dpp_shift<1>(cur.z), dpp_shift<1>(cur.w));
                t[KC + 1] = make_float4(dpp_shift<-1>(cur.x), dpp_shift<-1>(cur.y), dpp_shift<-1>(cur.z), dpp_shift<-1>(cur.w));
                if (KC > 1) {
                    t[KC - 2] = make_float4(dpp_shift<2>(cur.x), dpp_shift<2>(cur.y), dpp_shift<2>(cur.z), dpp_shift<2>(cur.w));
                    t[KC + 2] = make_float4(dpp_shift<-2>(cur.x), dpp_shift<-2>(cur.y), dpp_shift<-2>(cur.z), dpp_shift<-2>(cur.w));
                }
#pragma unroll
                for (int j = 0; j <= 2 * KC; ++j) {
                    P[2 * j] = (v2f){t[j].x, t[j].y};
                    P[2 * j + 1] = (v2f){t[j].z, t[j].w};
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                constexpr int C0 = 2 * KC;
                hm[kk][j] = P[C0 + j + SG].x - P[C0 + j - SG].x;
                ho[kk][j] = off_combine_sg<SG>(k, P[C0 + j - SG], P[C0 + j], P[C0 + j + SG]);
            }
            // determinant row r = yi - SG: ring rows r - SG (oldest), r, r + SG (the one just made)
            const int so = (kk + 1) % NR, sm = (kk + 1 + SG) % NR;
            float dC[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float lxx = off_combine_s<SG>(k, hm[so][j], hm[sm][j], hm[kk][j]);
                const v2f d2 = ho[kk][j] - ho[so][j];                      // {Lxy, Lyy}
                dC[j] = (lxx * d2.y - d2.x * d2.x) * sigma_quat;
            }
            const int r = yi - SG;
            if (WRITE_DET && r >= ys && r < ye) {              // the Ldet planes exist for the parity taps only
                float* dst = Ldet + (size_t)frame * fs + (size_t)r * w;
                if (in0) dst[x0] = dC[0];
                if (in1) dst[x0 + 1] = dC[1];
            }
            // horizontal maxima of the new row: pixel 0 has neighbours (left lane's pixel 1, own pixel 1), pixel 1
            // has (own pixel 0, right lane's pixel 0)
            const float lC = dpp_shift<1>(dC[1]), rC = dpp_shift<-1>(dC[0]);
            const float s2C[2] = {fmaxf(lC, dC[1]), fmaxf(dC[0], rC)};
            const float h3C[2] = {fmaxf(s2C[0], dC[0]), fmaxf(s2C[1], dC[1])};
            // extremum test of row rc = r - 1 (rows A, B, C = rc - 1, rc, rc + 1)
            const int rc = r - 1;
            if (rc >= rc_lo && rc < rc_hi) {                    // wave-uniform
                bool isc[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float nbm = fmaxf(fmaxf(h3A[j], h3C[j]), s2B[j]);
                    isc[j] = (j == 0 ? useful0 : useful1) && dB[j] > cp.thr && dB[j] > nbm;
                }
                const unsigned long long m0 = __ballot(isc[0]), m1 = __ballot(isc[1]);
                if ((m0 | m1) != 0ull) {                        // wave-uniform and rare: build the records
                    const float lA = dpp_shift<1>(dA[1]), rA = dpp_shift<-1>(dA[0]);
                    const float lB = dpp_shift<1>(dB[1]), rB = dpp_shift<-1>(dB[0]);
                    // columns x0-1 .. x0+2 of the three rows
                    const float rm[4] = {lA, dA[0], dA[1], rA};
                    const float rz[4] = {lB, dB[0], dB[1], rB};
                    const float rp[4] = {lC, dC[0], dC[1], rC};
                    const int cnt = __popcll(m0) + __popcll(m1);
                    if (nbuf + cnt > kCandBuf) {
                        cand_flush(s_cb[wv], nbuf, lane, cand, ncand, list, cp.cap, err);
                        nbuf = 0;
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const unsigned long long m = j == 0 ? m0 : m1;
                        if (isc[j]) {
                            const int slot = nbuf + (j == 1 ? __popcll(m0) : 0) +
                                             (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                            CandU cu = {(uint32_t)(x0 + j) | ((uint32_t)rc << 16), rz[1 + j],
                                        {rm[j], rm[j + 1], rm[j + 2], rz[j], rz[j + 2], rp[j], rp[j + 1], rp[j + 2]}};
                            s_cb[wv][slot] = cu;
                        }
                    }
                    nbuf += cnt;
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                dA[j] = dB[j]; dB[j] = dC[j];
                h3A[j] = h3B[j]; h3B[j] = h3C[j];
                s2B[j] = s2C[j];
            }
        }
    }
    cand_flush(s_cb[wv], nbuf, lane, cand, ncand, list, cp.cap, err);
}

// Raster order (y, then x) of each (frame, level) candidate list: bitonic sort of 64-bit keys
// (y << 48 | x << 32 | slot) in LDS ((x, y) is unique, so the order is total), then a gather of the unsorted
// records into the sorted position / response list and the sorted neighbourhood list.
constexpr uint32_t kCandRadixMax = 4096;
constexpr size_t kCandRadixLdsBytes = sizeof(uint32_t) * (3 * (size_t)kCandRadixMax + 16 * 256);
struct CandLevelRows {
    uint32_t h[kAkzMaxLevels];   // rows of every level's plane
};
constexpr uint32_t kCandCountRows = 2048;   // the counting path's row histogram (one thread owns two rows) ...
constexpr uint32_t kCandCountMax = 4096;    // ... and its list length (four candidates per thread)
__global__ __launch_bounds__(1024) void k_cand_sort(const CandU* __restrict__ cand_u, const uint32_t* __restrict__ ncand,
                                                    uint32_t cap, uint2* __restrict__ cand, float* __restrict__ cand_nb,
                                                    unsigned long long* __restrict__ gkeys, uint32_t gstride, uint32_t lds_keys,
                                                    CandLevelRows rows)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* lds = reinterpret_cast<unsigned long long*>(smem);
    const uint32_t level = blockIdx.x, frame = blockIdx.y;
    const uint32_t n = min(ncand[(size_t)frame * kAkzMaxLevels + level], cap);
    if (n == 0) return;
    const size_t list = ((size_t)frame * kAkzMaxLevels + level) * cap;
    const CandU* seg = cand_u + list;
    if (n <= 256) {
        // a short list (the last octaves): positions are unique, so a candidate's place is the number of keys below its
        // own — one thread per candidate, the keys read back four at a time as LDS broadcasts.  (Quadratic: at 1 000
        // candidates the block took 13 us, LDS-bound; the radix passes further down cost ~21 us whatever the length.)
        uint32_t* rk = reinterpret_cast<uint32_t*>(smem);
        const uint32_t tid = threadIdx.x, n4 = (n + 3u) & ~3u;
        uint32_t mine = 0xFFFFFFFFu;
        CandU cu;
        if (tid < n) {
            cu = seg[tid];
            mine = ((cu.xy >> 16) << 16) | (cu.xy & 0xFFFFu);   // y in the high half: raster order
        }
        if (tid < n4) rk[tid] = mine;
        __syncthreads();
        if (tid < n) {
            const uint4* r4 = reinterpret_cast<const uint4*>(rk);
            uint32_t rank = 0;
            for (uint32_t j = 0; j < n4 / 4u; ++j) {
                const uint4 k = r4[j];
                rank += (k.x < mine ? 1u : 0u) + (k.y < mine ? 1u : 0u) + (k.z < mine ? 1u : 0u) + (k.w < mine ? 1u : 0u);
            }
            cand[list + rank] = make_uint2(cu.xy, __float_as_uint(cu.v));
            float4* nb = reinterpret_cast<float4*>(cand_nb + (list + rank) * 8);
            nb[0] = make_float4(cu.nb[0], cu.nb[1], cu.nb[2], cu.nb[3]);
            nb[1] = make_float4(cu.nb[4], cu.nb[5], cu.nb[6], cu.nb[7]);
        }
        return;
    }
    const uint32_t h = rows.h[level];
    if (n <= kCandCountMax && h < kCandCountRows) {
        // Counting sort by row, then by column inside the row.  A 1080p level holds ~1 300 candidates on 1 080 rows: a row
        // group is one or two entries, so the place of a candidate is its row's start (a histogram of the rows and its
        // exclusive scan) plus the number of entries of the row's group left of it — three barriers and no pass structure
        // (~6 us a list against the ~21 us of four radix passes).
        uint32_t* hist = reinterpret_cast<uint32_t*>(smem);                    // [h + 1] counts, then row starts
        uint2* grp = reinterpret_cast<uint2*>(hist + kCandCountRows + 4);       // [n] {x, id} grouped by row, arrival order
        __shared__ uint32_t s_wsum[16];
        const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
        for (uint32_t b = tid; b <= h; b += 1024) hist[b] = 0u;
        __syncthreads();
        uint32_t xy[4], slot[4];
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t i = tid + 1024u * q;
            xy[q] = 0u;
            slot[q] = 0u;
            if (i < n) {
                xy[q] = seg[i].xy;
                slot[q] = atomicAdd(&hist[min(xy[q] >> 16, h)], 1u);
            }
        }
        __syncthreads();
        // exclusive scan of the row counts: thread t owns rows 2t and 2t + 1
        const uint32_t c0 = 2u * tid <= h ? hist[2u * tid] : 0u, c1 = 2u * tid + 1u <= h ? hist[2u * tid + 1u] : 0u;
        uint32_t inc = c0 + c1;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off);
            if ((int)lane >= off) inc += o;
        }
        if (lane == 63u) s_wsum[wv] = inc;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t w = 0; w < wv; ++w) before += s_wsum[w];
        const uint32_t excl = before + inc - (c0 + c1);
        if (2u * tid <= h) hist[2u * tid] = excl;
        if (2u * tid + 1u <= h) hist[2u * tid + 1u] = excl + c0;
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t i = tid + 1024u * q;
            if (i < n) grp[hist[min(xy[q] >> 16, h)] + slot[q]] = make_uint2(xy[q] & 0xFFFFu, i);
        }
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t i = tid + 1024u * q;
            if (i < n) {
                const uint32_t y = min(xy[q] >> 16, h), x = xy[q] & 0xFFFFu;
                const uint32_t g0 = hist[y], g1 = y < h ? hist[y + 1u] : n;
                uint32_t pos = g0;
                for (uint32_t k = g0; k < g1; ++k) pos += grp[k].x < x ? 1u : 0u;
                const CandU cu = seg[i];
                cand[list + pos] = make_uint2(cu.xy, __float_as_uint(cu.v));
                float4* nb = reinterpret_cast<float4*>(cand_nb + (list + pos) * 8);
                nb[0] = make_float4(cu.nb[0], cu.nb[1], cu.nb[2], cu.nb[3]);
                nb[1] = make_float4(cu.nb[4], cu.nb[5], cu.nb[6], cu.nb[7]);
            }
        }
        return;
    }
    if (n <= kCandRadixMax) {
        // positions inside a level are unique 32-bit keys (y << 16 | x): LSD radix sort of the ids in LDS (akz_common.h)
        // instead of the bitonic network over padded 64-bit keys; lists of up to kCandRadixMax candidates (64 KB of LDS:
        // two blocks per CU, as before)
        uint32_t* rk = reinterpret_cast<uint32_t*>(smem);
        uint32_t* ia = rk + kCandRadixMax;
        uint32_t* ib = ia + kCandRadixMax;
        uint32_t* wh = ib + kCandRadixMax;
        __shared__ uint32_t s_tot[256];
        for (uint32_t i = threadIdx.x; i < n; i += 1024) {
            const uint32_t xy = seg[i].xy;
            rk[i] = ((xy >> 16) << 16) | (xy & 0xFFFFu);      // y in the high half: raster order
            ia[i] = i;
        }
        const uint32_t* sorted = lds_radix_sort_ids(rk, ia, ib, wh, s_tot, n, 4);
        for (uint32_t i = threadIdx.x; i < n; i += 1024) {
            const CandU cu = seg[sorted[i]];
            cand[list + i] = make_uint2(cu.xy, __float_as_uint(cu.v));
            float4* nb = reinterpret_cast<float4*>(cand_nb + (list + i) * 8);
            nb[0] = make_float4(cu.nb[0], cu.nb[1], cu.nb[2], cu.nb[3]);
            nb[1] = make_float4(cu.nb[4], cu.nb[5], cu.nb[6], cu.nb[7]);
        }
        return;
    }
    uint32_t np2 = 1;
    while (np2 < n) np2 <<= 1;
    // lists longer than the LDS buffer sort through the list's global key scratch (akz_common.h)
    const bool big = np2 > lds_keys;
    unsigned long long* key = big ? gkeys + ((size_t)frame * kAkzMaxLevels + level) * gstride : lds;
    for (uint32_t i = threadIdx.x; i < np2; i += 1024) {
        unsigned long long kk = ~0ull;
        if (i < n) {
            const uint32_t xy = seg[i].xy;
            const uint32_t yx = ((xy >> 16) << 16) | (xy & 0xFFFFu);  // y in the high half: raster order
            kk = ((unsigned long long)yx << 32) | (unsigned long long)i;
        }
        key[i] = kk;
    }
    __threadfence_block();
    __syncthreads();
    if (big) bitonic_sort_big_u64<1024>(key, np2, lds, lds_keys);
    else bitonic_sort_lds_u64<1024>(key, np2);
    for (uint32_t i = threadIdx.x; i < n; i += 1024) {
        const CandU cu = seg[(uint32_t)key[i]];
        cand[list + i] = make_uint2(cu.xy, __float_as_uint(cu.v));
        float4* nb = reinterpret_cast<float4*>(cand_nb + (list + i) * 8);
        nb[0] = make_float4(cu.nb[0], cu.nb[1], cu.nb[2], cu.nb[3]);
        nb[1] = make_float4(cu.nb[4], cu.nb[5], cu.nb[6], cu.nb[7]);
    }
}

// The same raster order for calls of a few frames, as a RANK sort: positions inside a level are unique, so a
// candidate's place is the number of candidates with a smaller (y, x).  A block ranks 32 candidates of one
// (frame, level) list, its 256 threads split every 1024-key tile 8 ways: the whole chip instead of one block per list
// (single 1080p frame: 34 -> ~10 us on the critical path; a batch keeps the bitonic blocks, which cost less in total).
__global__ __launch_bounds__(256) void k_cand_rank(const CandU* __restrict__ cand_u, const uint32_t* __restrict__ ncand,
                                                   uint32_t cap, uint2* __restrict__ cand, float* __restrict__ cand_nb)
{
    constexpr int kI = 32, kJ = 8, kSlice = 1024 / kJ;
    __shared__ uint32_t s_key[1024];
    __shared__ uint32_t s_part[kJ][kI];
    const uint32_t level = blockIdx.y, frame = blockIdx.z;
    const uint32_t n = min(ncand[(size_t)frame * kAkzMaxLevels + level], cap);
    if (blockIdx.x * (uint32_t)kI >= n) return;    // whole block
    const size_t list = ((size_t)frame * kAkzMaxLevels + level) * cap;
    const CandU* seg = cand_u + list;
    const uint32_t ii = threadIdx.x & (kI - 1), js = threadIdx.x / kI;
    const uint32_t i = blockIdx.x * (uint32_t)kI + ii;
    uint32_t mine = 0xFFFFFFFFu;
    if (i < n) {
        const uint32_t xy = seg[i].xy;
        mine = ((xy >> 16) << 16) | (xy & 0xFFFFu);   // y in the high half: raster order
    }
    uint32_t rank = 0;
    for (uint32_t t0 = 0; t0 < n; t0 += 1024) {
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < 1024; j += 256) {
            uint32_t k = 0xFFFFFFFFu;
            if (t0 + j < n) {
                const uint32_t xy = seg[t0 + j].xy;
                k = ((xy >> 16) << 16) | (xy & 0xFFFFu);
            }
            s_key[j] = k;
        }
        __syncthreads();
        const uint32_t m = min(1024u, n - t0);
        const uint32_t j0 = js * kSlice, j1 = min(m, j0 + (uint32_t)kSlice);
#pragma unroll 8
        for (uint32_t j = j0; j < j1; ++j) rank += s_key[j] < mine ? 1u : 0u;
    }
    s_part[js][ii] = rank;
    __syncthreads();
    if (js == 0 && i < n) {
        rank = 0;
#pragma unroll
        for (int k = 0; k < kJ; ++k) rank += s_part[k][ii];
        const CandU cu = seg[i];
        cand[list + rank] = make_uint2(cu.xy, __float_as_uint(cu.v));
        float4* nb = reinterpret_cast<float4*>(cand_nb + (list + rank) * 8);
        nb[0] = make_float4(cu.nb[0], cu.nb[1], cu.nb[2], cu.nb[3]);
        nb[1] = make_float4(cu.nb[4], cu.nb[5], cu.nb[6], cu.nb[7]);
    }
}

// GrayFloatImage::from_dynamic (image.rs:45-109) on its own: pixels -> f32 (the generic-radius path of level 0)
template <typename InT>
__global__ __launch_bounds__(256) void k_to_f32(const InT* __restrict__ in, float* __restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = load_px(in, i);
}

__global__ __launch_bounds__(256) void k_deinterleave(const float2* __restrict__ in, float* __restrict__ out, size_t n,
                                                      int component)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = component ? in[i].y : in[i].x;
}

// ---------------------------------------------------------------------------------------------
// Generic dense 1-D filter for the stand-alone akaze::image API (image.rs:202-331), any odd ksize.
__global__ __launch_bounds__(256) void k_filter1d(const float* __restrict__ in, float* __restrict__ out, int w, int h,
                                                  const float* __restrict__ kern, int ksize, int vertical)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    int half = ksize / 2;
    float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int i = 0; i < ksize; ++i) {
        float s = vertical ? in[(size_t)clampi(y + i - half, 0, h - 1) * w + x]
                           : in[(size_t)y * w + clampi(x + i - half, 0, w - 1)];
        int l = i & 3;
        // (branch-free lane select keeps a[] in registers)
        a[0] = l == 0 ? arith_mad(s, kern[i], a[0]) : a[0];
        a[1] = l == 1 ? arith_mad(s, kern[i], a[1]) : a[1];
        a[2] = l == 2 ? arith_mad(s, kern[i], a[2]) : a[2];
        a[3] = l == 3 ? arith_mad(s, kern[i], a[3]) : a[3];
    }
    out[(size_t)y * w + x] = arith_reduce(a[0], a[1], a[2], a[3]);
}

OffK make_offk(uint32_t sigma)
{
    ScharrW sw = akz_scharr_weights(sigma);
    OffK k;
    if (sigma == 1) {  // simple_scharr: [3,10,3], taps in lanes 0,1,2 -> (a*3 + b*10) + c*3
        k.n = 3.0f;
        k.m = 10.0f;
    } else {
        k.n = sw.norm;
        k.m = sw.middle;
    }
    k.mode = offk_mode(sigma);
    return k;
}

inline dim3 grid_px(int w, int h, int n) { return dim3(akz_div_up(w, 64), akz_div_up(h, 4), n); }

// k_level_resident's instantiation: rows per patch half and the most waves of a workgroup (kResWaves x 64 threads hold
// kResWaves x kResR x 2 rows of up to 256 columns), and whether a w x h level fits: S = rows of the upper half, nw = waves.
constexpr int kResR = AKZ_RES_R, kResWaves = AKZ_RES_WAVES;
inline bool resident_geometry(int w, int h, int R, int nwmax, int* S_out, int* nw_out)
{
    if ((w & 3) != 0 || w < 8 || w > 256 || h < 4 * R) return false;
    const int nw = akz_div_up((h + 1) / 2, R);
    if (nw > nwmax) return false;
    const int S = nw * R;
    if (S > h - 1) return false;                                  // the lower half holds at least one image row
    if ((size_t)((w + 8) / 2) * 16 * (size_t)(S + 8) > (size_t)kResLdsBytes) return false;   // the plane of pairs fits
    *S_out = S;
    *nw_out = nw;
    return true;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
int32_t AKZ_SS(akz_dev_filter1d)(hipStream_t s, const float* in, float* out, int w, int h, const float* d_kernel, int ksize,
                                int vertical)
{
    AKZ_LAUNCH(k_filter1d, grid_px(w, h, 1), dim3(256), 0, s, in, out, w, h, d_kernel, ksize, vertical);
    AKZ_LAUNCH_CHECK();
    return AKZ_OK;
}

#if AKZ_ARITH == 0   // (no arithmetic in it: one copy serves every context)
int32_t akz_dev_deinterleave(hipStream_t s, const float2* in, float* out, size_t n, int component)
{
    AKZ_LAUNCH(k_deinterleave, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n, component);
    AKZ_LAUNCH_CHECK();
    return AKZ_OK;
}
#endif

int32_t AKZ_SS(akz_dev_half_size)(hipStream_t s, const float* in, float* out, int w, int h, int n, size_t in_fs,
                                 size_t out_fs)
{
    if (w / 2 <= 0 || h / 2 <= 0) return AKZ_E_INVALID;
    AKZ_LAUNCH(k_half_size, grid_px(w / 2, h / 2, n), dim3(256), 0, s, in, out, w, h, in_fs, out_fs);
    AKZ_LAUNCH_CHECK();
    return AKZ_OK;
}

template <int R, int E, typename InT, int EPI>
static int32_t launch_blur(akz_ctx* c, const InT* in, int w, int h, size_t in_fs, const GaussTaps& taps, float* out_g,
                           float* out_flow, size_t out_fs, int invk_off, int n)
{
    AkzSet& S = c->S();
    dim3 grid(akz_div_up(w, kTW), akz_div_up(h, kTH), n);
    AKZ_LAUNCH((k_blur_tile<R, E, InT, EPI>), grid, dim3(256), 0, c->stream, in, w, h, in_fs, taps, out_g,
                       out_flow, out_fs, S.d_invk, invk_off, S.d_cmax, S.d_hist, S.d_npoints,
                       (int)c->cfg.contrast_factor_num_bins);
    AKZ_LAUNCH_CHECK();
    return AKZ_OK;
}

static GaussTaps make_taps(float sigma)
{
    GaussTaps t;
    int r = akz_gaussian_radius(sigma);
    t.n = 2 * r + 1;
    for (int i = 0; i < 12; ++i) t.k[i] = 0.0f;
    akz_host_gaussian_kernel(sigma, t.n, t.k);
    return t;
}

template <typename InT>
static int32_t scale_space_impl(akz_ctx* c, const InT* d_imgs, int n)
{
    const AkzPlan& P = c->plan;
    AkzSet& S = c->S();
    const int w = P.w, h = P.h;
    const size_t P0 = (size_t)w * h;
    const int nlev = (int)P.levels.size();
    if (nlev == 0) return AKZ_OK;
    hipStream_t s = c->stream;
    const int nbins = (int)c->cfg.contrast_factor_num_bins;

    // base_scale_offset and the gradient-histogram scale are config values; the fused tile kernel is
    // instantiated for the radii the AKAZE path uses (sigma 1.6 -> 4, sigma 1.0 -> 2).
    // the fused tile kernels are instantiated for the radii of the default configuration (base_scale_offset in
    // (1.5, 2.0] -> radius 4, and the fixed sigma 1.0 -> radius 2); any other base_scale_offset takes the dense
    // stand-alone filter (k_filter1d) for the one blur that depends on it
    const float sigma0 = (float)c->cfg.base_scale_offset;
    const int r0 = akz_gaussian_radius(sigma0);
    const bool tile0 = r0 == 4;
    GaussTaps t0 = make_taps(tile0 ? sigma0 : 1.6f);
    GaussTaps t1 = make_taps(1.0f);
    if (t1.n != 5 || (tile0 && t0.n != 9)) return AKZ_E_INTERNAL;

    akz_timer_begin(c, AKZ_T_SCALE_SPACE, s);
    // lib.rs:199-201 — Lt[0] = gaussian_blur(image, base_scale_offset); Lsmooth[0] = Lt[0]
    const bool fused0 = tile0 && P.levels[0].deriv_sigma == 2;  // fused blur + first derivatives (default config)
    akz_timer_begin(c, AKZ_T_FRONT0, s);
    if (fused0 && (w & 3) == 0 && c->front_pair) {
#define AKZ_FRONT0(THV, NTV, TPBV)                                                                                  \
    AKZ_LAUNCH((k_level_front2<4, 2, THV, NTV, InT, false, TPBV>),                                              \
                       dim3(akz_div_up(w, kTW), akz_div_up(akz_div_up(h, THV), TPBV), (n + 1) / 2), dim3(NTV), 0, s,    \
                       d_imgs, w, h, P0, n, t0, make_offk(2), S.Lt[0], (float*)nullptr, S.Lxy[0], (const float*)nullptr, 0)
        // 32-row tiles (measured against 24 x 512 and against 5 row tiles per block with register prefetch: 10.18
        // vs 10.50 / 10.83 ms of scale space per 64 frames); 512 threads here, where the u8 input leaves registers
        // for eight waves per block (441 vs 490 us per 64 frames), 256 on the f32 levels (178 vs 184 us)
        AKZ_FRONT0(32, 512, 1);
#undef AKZ_FRONT0
        AKZ_LAUNCH_CHECK();
    } else if (fused0) {
        AKZ_LAUNCH((k_level_front<4, 2, InT, false>), dim3(akz_div_up(w, kTW), akz_div_up(h, kTH), n), dim3(256), 0,
                           s, d_imgs, w, h, P0, t0, make_offk(2), S.Lt[0], (float*)nullptr, S.Lxy[0],
                           (const float*)nullptr, 0);
        AKZ_LAUNCH_CHECK();
    } else if (tile0) {
        AKZ_TRY((launch_blur<4, 0, InT, EPI_BLUR>(c, d_imgs, w, h, P0, t0, S.Lt[0], nullptr, P0, 0, n)));
    } else {
        // generic radius: pixels -> f32 (scratch plane), horizontal pass -> Ldet[0] (free until the determinant
        // kernel of level 0 runs), vertical pass -> Lt[0], frame by frame
        const int ks = 2 * r0 + 1;
        if (ks > kAkzMaxTaps) return AKZ_E_INVALID;
        c->h_taps.assign((size_t)ks, 0.0f);
        akz_host_gaussian_kernel(sigma0, ks, c->h_taps.data());
        AKZ_HIP(hipMemcpyAsync(c->d_taps, c->h_taps.data(), sizeof(float) * ks, hipMemcpyHostToDevice, s));
        AKZ_LAUNCH((k_to_f32<InT>), dim3((unsigned)((P0 * n + 255) / 256)), dim3(256), 0, s, d_imgs, S.tmp, P0 * n);
        AKZ_LAUNCH_CHECK();
        for (int f = 0; f < n; ++f) {
            AKZ_LAUNCH(k_filter1d, grid_px(w, h, 1), dim3(256), 0, s, (const float*)(S.tmp + (size_t)f * P0),
                               S.Ldet[0] + (size_t)f * P0, w, h, (const float*)c->d_taps, ks, 0);
            AKZ_LAUNCH(k_filter1d, grid_px(w, h, 1), dim3(256), 0, s, (const float*)(S.Ldet[0] + (size_t)f * P0),
                               S.Lt[0] + (size_t)f * P0, w, h, (const float*)c->d_taps, ks, 1);
            AKZ_LAUNCH_CHECK();
        }
    }
    akz_timer_end(c, AKZ_T_FRONT0, s, 1, (uint64_t)P0 * n);
    // lib.rs:206-211 — contrast factor on the ORIGINAL image
    // d_cmax, d_npoints, d_ncand, d_hist, d_fine: contiguous (carve_set)
    AKZ_HIP(hipMemsetAsync(S.d_cmax, 0, S.zero_bytes, s));
    const bool pairc = (w & 3) == 0 && c->front_pair && nbins <= 510;
    akz_timer_begin(c, AKZ_T_CONTRAST, s);
    const bool fine = pairc && c->contrast_fine;
    if (pairc) {
        const int ctiles = n <= kLatencyFrames ? kCTilesFew : kCTiles;
        dim3 gridc(akz_div_up(w, kTW), akz_div_up(akz_div_up(h, kFTH), ctiles), (n + 1) / 2);
        AKZ_LAUNCH((k_contrast_pair<InT, EPI_CMAX>), gridc, dim3(kFNT), 0, s, d_imgs, w, h, P0, n, t1, S.d_cmax,
                           (const double*)S.d_cthr, S.d_hist, S.d_npoints, nbins, fine ? S.d_fine : (uint32_t*)nullptr,
                           (const uint32_t*)nullptr, ctiles);
        AKZ_LAUNCH_CHECK();
        AKZ_LAUNCH(k_contrast_thresholds, dim3(n), dim3(512), 0, s, S.d_cmax, nbins, S.d_cthr);
        AKZ_LAUNCH_CHECK();
        if (fine) {
            AKZ_LAUNCH(k_contrast_resolve, dim3(n), dim3(256), 0, s, S.d_cmax, S.d_fine, S.d_npoints,
                               (const double*)S.d_cthr, nbins, c->cfg.contrast_percentile, P.n_octaves, S.d_contrast,
                               S.d_invk, S.d_cflag, c->contrast_force_odd ? 1 : 0);
            AKZ_LAUNCH_CHECK();
        }
        AKZ_LAUNCH((k_contrast_pair<InT, EPI_CHIST>), gridc, dim3(kFNT), 0, s, d_imgs, w, h, P0, n, t1, S.d_cmax,
                           (const double*)S.d_cthr, S.d_hist, S.d_npoints, nbins, (uint32_t*)nullptr,
                           fine ? (const uint32_t*)S.d_cflag : (const uint32_t*)nullptr, ctiles);
        AKZ_LAUNCH_CHECK();
    } else {
        AKZ_TRY((launch_blur<2, 1, InT, EPI_CMAX>(c, d_imgs, w, h, P0, t1, nullptr, nullptr, 0, 0, n)));
        AKZ_TRY((launch_blur<2, 1, InT, EPI_CHIST>(c, d_imgs, w, h, P0, t1, nullptr, nullptr, 0, 0, n)));
    }
    AKZ_LAUNCH(k_contrast_finish, dim3(akz_div_up(n, 64)), dim3(64), 0, s, S.d_cmax, S.d_hist,
                       S.d_npoints, nbins, c->cfg.contrast_percentile, n, P.n_octaves, S.d_contrast, S.d_invk,
                       fine ? (const uint32_t*)S.d_cflag : (const uint32_t*)nullptr);
    AKZ_LAUNCH_CHECK();
    akz_timer_end(c, AKZ_T_CONTRAST, s, 2, (uint64_t)P0 * n);

    uint64_t fed_launches = 0, fed_units = 0;
    // the side stream shortens the dependency chain of a few-frame call; a batch fills the chip either way (measured:
    // 6342 vs 6310 frames/s) and its kernels are easier to read in a profile when they do not overlap each other
    const bool det_side = c->det_side_stream && c->stream_det != nullptr && n <= kLatencyFrames;
    int half_ready = -1;        // the level whose half-sized start image its predecessor's last FED launch has written
    for (int i = 0; i < nlev; ++i) {
        const AkzLevel& L = P.levels[i];
        const size_t fs = L.pixels();
        const float* smooth = S.Lt[0];
        bool fused_front = false;
        if (i > 0) {
            const int nsteps = (int)L.tau.size();
            // Ping-pong so the last FED step lands in Lt[i]; `init` is where the un-diffused Lt[i] lives.
            float* bufA = S.Lt[i];
            float* bufB = S.tmp;
            const bool blocked = (L.w & 3) == 0 && c->fed_block > 1 && c->front_pair;   // k_fed_pair
            // steps per launch: the first octave's levels have 3-4 steps; deeper octaves run more steps on smaller
            // images, where fewer, longer launches win over the smaller useful tile
            // (measured, 256 x 1080p: 4 / 5 / 6 / 8 steps per launch below the first octave -> 5186 / 5216 / 5259 /
            // 5299 frames/s; a three-patch halo for up to 12 steps loses again)
            int fed_block = L.octave == 0 ? (c->fed_block < 4 ? c->fed_block : 4) : c->fed_block;
            // a call of a few frames is a chain of dependent launches (~5 us each before any work): below the first
            // octave it runs up to 16 steps per launch (three- and four-patch halos, 40 x 40 / 32 x 32 useful tiles),
            // which a batch would lose on (see above) and a single frame wins on
            if (n <= kLatencyFrames && L.octave > 0 && c->fed_block == 8) fed_block = kFedMaxBlock;
            const int nwrites = blocked ? (nsteps + fed_block - 1) / fed_block : nsteps;  // FED launches
            // The level an octave ends with hands the next octave its half-sized start image straight from the registers of
            // its last FED launch (fed_store_half) instead of being read back by k_half_size.  The image goes to the plane of
            // the level AFTER the next one, which nobody touches until that level's own diffusion.
            // k_level_resident: the whole level (front end + every FED step) in one launch, one workgroup per frame, when
            // the level is small enough to live on one compute unit (octave 3 of a 1080p pyramid; octaves >= 2 of smaller frames)
            int res_S = 0, res_nw = 0;
            // One workgroup per frame: a call of few frames leaves most compute units idle under it, and the tile kernels'
            // many small launches win (1080p, octave 3, serial phase times: 64 frames 5.85 vs 5.78 ms of scale space, 96 frames
            // a tie, 128 frames 10.74 vs 10.84, 256 frames 20.30 vs 20.82)
            const int res_min = c->resident_min_frames > 0 ? c->resident_min_frames : (3 * c->n_cu + 7) / 8;
            const bool resident = c->resident_levels && n >= res_min && blocked && L.octave > 0 && L.deriv_sigma >= 2 && L.deriv_sigma <= 4 &&
                                  !c->keep_all && nsteps >= 1 && nsteps <= kResMaxSteps &&
                                  resident_geometry(L.w, L.h, kResR, kResWaves, &res_S, &res_nw);
            float* half_next = nullptr;
            size_t half_next_fs = 0;
            if (!resident && blocked && nsteps > 0 && i + 2 < nlev && P.levels[i + 1].new_octave && P.levels[i + 1].w == (L.w >> 1) &&
                P.levels[i + 1].h == (L.h >> 1) && !P.levels[i + 2].new_octave) {
                half_next = S.Lt[i + 2];
                half_next_fs = P.levels[i + 1].pixels();
            }
            const float* init;
            if (L.new_octave && half_ready == i) {
                init = S.Lt[i + 1];      // written by the previous level's last FED launch (fed_store_half)
            } else if (L.new_octave) {
                const AkzLevel& Lp = P.levels[i - 1];
                // the first FED launch must not write where it reads: with an odd number of launches the
                // first one writes Lt[i], so the half-sized image goes to the scratch plane, and vice versa
                float* half_dst = (nwrites % 2 == 0) ? bufA : bufB;
                if (nwrites == 0) half_dst = bufA;
                if (resident) half_dst = bufB;   // (one launch that writes Lt[i]: the start image goes to the scratch plane)
                AKZ_TRY(AKZ_SS(akz_dev_half_size)(s, S.Lt[i - 1], half_dst, Lp.w, Lp.h, n, Lp.pixels(), fs));
                init = half_dst;
            } else {
                init = S.Lt[i - 1];  // lib.rs:230 clone(): read in place, never modified again
            }
            // lib.rs:232-248 — Lsmooth = blur(Lt, 1.0); Lx,Ly = simple Scharr; Lflow = pm_g2
            fused_front = L.deriv_sigma >= 2 && L.deriv_sigma <= 4;
            if (resident) {
                FedTausN ft;
                for (int q = 0; q < kResMaxSteps; ++q) ft.half_tau[q] = q < nsteps ? 0.5f * (float)L.tau[q] : 0.0f;
                OffK kk = make_offk(L.deriv_sigma);
                akz_timer_begin(c, AKZ_T_LEVEL_RESIDENT, s);
#define AKZ_RES(SGV)                                                                                                 \
    AKZ_LAUNCH((k_level_resident<SGV, kResR, kResWaves>), dim3(n), dim3(res_nw * 64), 0, s, init, L.w, L.h, res_S, fs, t1, kk, ft,   \
               nsteps, S.Lt[i], S.Lxy[i], (const float*)S.d_invk, (int)L.octave)
                switch (L.deriv_sigma) {
                case 2: AKZ_RES(2); break;
                case 3: AKZ_RES(3); break;
                default: AKZ_RES(4); break;
                }
#undef AKZ_RES
                AKZ_LAUNCH_CHECK();
                akz_timer_end(c, AKZ_T_LEVEL_RESIDENT, s, 1, (uint64_t)fs * n);
                fed_launches += nsteps;
                fed_units += (uint64_t)nsteps * fs * n;
            } else {
            const int t_front = AKZ_T_FRONT_SG2 + (int)L.deriv_sigma - 2;
            // the FED launches of the level: groups of up to fed_block steps, balanced sizes (5 -> 3 + 2)
            std::vector<int> groups;
            if (blocked)
                for (int left = nsteps; left > 0;) {
                    int ngr = (left + fed_block - 1) / fed_block;  // groups still to emit
                    int g = (left + ngr - 1) / ngr;
                    groups.push_back(g);
                    left -= g;
                }
            // k_front_fed: front end + the first of those launches in one kernel (Lflow stays on chip unless a later
            // launch of the level needs it: WRITE_FLOW).  The first octave's levels are one or two launches of at most 4
            // steps (one halo patch, HP = 1); below it the first launch runs up to 8 steps behind a two-patch halo (HP = 2:
            // rejected in rounds 2-3 at 9636 vs 9781 frames/s, adopted in round 5 at -4.9 % of the scale-space kernel time
            // once the kernel had lost 17 % of its instructions).  A few-frame call's 9..16-step launches keep the split path.
            const bool front_fed = c->fuse_front_fed && blocked && fused_front && !c->keep_all && !groups.empty() &&
                                   groups[0] <= (L.octave == 0 ? 4 : 8);
            if (front_fed) {
                const int ng = (int)groups.size();
                float* dst0 = ((ng - 1) % 2 == 0) ? bufA : bufB;
                FedTaus ft;
                for (int q = 0; q < kFedMaxBlock; ++q) ft.half_tau[q] = q < groups[0] ? 0.5f * (float)L.tau[q] : 0.0f;
                OffK kk = make_offk(L.deriv_sigma);
                const int t_ff = (L.octave == 0 ? AKZ_T_FRONT_FED_SG2 : AKZ_T_FRONT_FED_DEEP_SG2) + (int)L.deriv_sigma - 2;
                akz_timer_begin(c, t_ff, s);
#define AKZ_FF3(SGV, HPV, RGV, WFV)                                                                                  \
    AKZ_LAUNCH((k_front_fed<SGV, HPV, RGV, WFV>),                                                             \
                       dim3(akz_div_up(L.w, front_fed_tile(HPV)), akz_div_up(L.h, front_fed_tile(HPV)), (n + 1) / 2), \
                       dim3(256), 0, s, init, L.w, L.h, fs, n, t1, kk, ft, groups[0], dst0, S.Lflow[i], S.Lxy[i],      \
                       (const float*)S.d_invk, (int)L.octave, ng == 1 ? half_next : (float*)nullptr, half_next_fs)
#define AKZ_FF2(SGV, HPV, RGV) if (ng > 1) { AKZ_FF3(SGV, HPV, RGV, true); } else { AKZ_FF3(SGV, HPV, RGV, false); }
#define AKZ_FF(SGV)                                                                                                  \
    if (groups[0] <= 3) { AKZ_FF2(SGV, 1, false) } else if (groups[0] == 4) { AKZ_FF2(SGV, 1, true) }               \
    else if (groups[0] <= 7) { AKZ_FF2(SGV, 2, false) } else { AKZ_FF2(SGV, 2, true) }
                switch (L.deriv_sigma) {
                case 2: AKZ_FF(2) break;
                case 3: AKZ_FF(3) break;
                default: AKZ_FF(4) break;
                }
#undef AKZ_FF
#undef AKZ_FF2
#undef AKZ_FF3
                AKZ_LAUNCH_CHECK();
                akz_timer_end(c, t_ff, s, 1, (uint64_t)fs * n);
            } else if (fused_front) {
                akz_timer_begin(c, t_front, s);
                float* lsm_out = c->keep_all ? S.Lsm[i] : nullptr;  // Lsmooth stays on chip unless the taps want it
                dim3 gridf(akz_div_up(L.w, kTW), akz_div_up(L.h, kTH), n);
                OffK kk = make_offk(L.deriv_sigma);
#define AKZ_FRONT(SGV)                                                                                               \
    AKZ_LAUNCH((k_level_front<2, SGV, float, true>), gridf, dim3(256), 0, s, init, L.w, L.h, fs, t1, kk,      \
                       lsm_out, S.Lflow[i], S.Lxy[i], (const float*)S.d_invk, (int)L.octave)
#define AKZ_FRONT2X(SGV, THV, NTV, TPBV)                                                                             \
    AKZ_LAUNCH((k_level_front2<2, SGV, THV, NTV, float, true, TPBV>),                                            \
                       dim3(akz_div_up(L.w, kTW), akz_div_up(akz_div_up(L.h, THV), TPBV), (n + 1) / 2), dim3(NTV), 0, s, \
                       init, L.w, L.h, fs, n, t1, kk, lsm_out, S.Lflow[i], S.Lxy[i], (const float*)S.d_invk,            \
                       (int)L.octave)
#define AKZ_FRONT2(SGV)                                                                                              \
    AKZ_FRONT2X(SGV, 32, 256, 1)
                const bool pair = (L.w & 3) == 0 && c->front_pair;
                switch (L.deriv_sigma) {
                case 2: if (pair) { AKZ_FRONT2(2); } else AKZ_FRONT(2); break;
                case 3: if (pair) { AKZ_FRONT2(3); } else AKZ_FRONT(3); break;
                default: if (pair) { AKZ_FRONT2(4); } else AKZ_FRONT(4); break;
                }
#undef AKZ_FRONT
#undef AKZ_FRONT2
#undef AKZ_FRONT2X
                AKZ_LAUNCH_CHECK();
                akz_timer_end(c, t_front, s, 1, (uint64_t)fs * n);
            } else {
                AKZ_TRY((launch_blur<2, 1, float, EPI_FLOW>(c, init, L.w, L.h, fs, t1, S.Lsm[i], S.Lflow[i], fs,
                                                           (int)L.octave, n)));
            }
            // lib.rs:251-256 — FED cycle
            akz_timer_begin(c, AKZ_T_FED, s);
            const float* src = init;
            if (blocked) {
                // temporally blocked: the ping-pong parity is chosen per GROUP so that the last group lands in Lt[i]
                const int ng = (int)groups.size();  // == nwrites
                int j = 0;
                for (int gi = 0; gi < ng; ++gi) {
                    float* dstb = ((ng - 1 - gi) % 2 == 0) ? bufA : bufB;
                    if (gi == 0 && front_fed) {                    // k_front_fed did this launch
                        j += groups[0];
                        src = dstb;
                        continue;
                    }
                    FedTaus ft;
                    for (int q = 0; q < kFedMaxBlock; ++q) ft.half_tau[q] = q < groups[gi] ? 0.5f * (float)L.tau[j + q] : 0.0f;
#define AKZ_FED_CASE(TT)                                                                                              \
    case TT: {                                                                                                        \
        dim3 gridp(akz_div_up(L.w, fed_tile_edge(TT)), akz_div_up(L.h, fed_tile_edge(TT)), (n + 1) / 2);              \
        AKZ_LAUNCH((k_fed_pair<TT>), gridp, dim3(256), 0, s, src, S.Lflow[i], dstb, L.w, L.h, fs, n, ft,      \
                   gi == ng - 1 ? half_next : (float*)nullptr, half_next_fs);                                        \
    } break;
                    const int t_fed = AKZ_T_FED_T1 + (groups[gi] <= 8 ? groups[gi] : 8) - 1;   // (9..16 steps: single-frame calls only)
                    akz_timer_begin(c, t_fed, s);
                    switch (groups[gi]) {
                        AKZ_FED_CASE(1) AKZ_FED_CASE(2) AKZ_FED_CASE(3) AKZ_FED_CASE(4)
                        AKZ_FED_CASE(5) AKZ_FED_CASE(6) AKZ_FED_CASE(7) AKZ_FED_CASE(8)
                        AKZ_FED_CASE(9) AKZ_FED_CASE(10) AKZ_FED_CASE(11) AKZ_FED_CASE(12)
                        AKZ_FED_CASE(13) AKZ_FED_CASE(14) AKZ_FED_CASE(15) AKZ_FED_CASE(16)
                    }
#undef AKZ_FED_CASE
                    AKZ_LAUNCH_CHECK();
                    akz_timer_end(c, t_fed, s, 1, (uint64_t)fs * n);
                    j += groups[gi];
                    src = dstb;
                }
            } else
            for (int j = 0; j < nsteps; ++j) {
                float* dst = ((nsteps - 1 - j) % 2 == 0) ? bufA : bufB;
                float half_tau = 0.5f * (float)L.tau[j];
                AKZ_LAUNCH(k_fed_step, grid_px(L.w, L.h, n), dim3(256), 0, s, src, S.Lflow[i], dst, L.w, L.h, fs,
                                   half_tau);
                AKZ_LAUNCH_CHECK();
                src = dst;
            }
            akz_timer_end(c, AKZ_T_FED, s, (uint64_t)nwrites, (uint64_t)nsteps * fs * n, (uint64_t)nwrites * fs * n);
            if (half_next) half_ready = i + 1;
            fed_launches += nsteps;
            fed_units += (uint64_t)nsteps * fs * n;
            if (nsteps == 0 && init != bufA)
                AKZ_HIP(hipMemcpyAsync(bufA, init, sizeof(float) * fs * n, hipMemcpyDeviceToDevice, s));
            }   // !resident
            smooth = S.Lsm[i];
        }
        // detector_response.rs:60-67 + :33-57
        OffK k = make_offk(L.deriv_sigma);
        if (!(i == 0 ? fused0 : fused_front)) {
            AKZ_LAUNCH(k_deriv_first, grid_px(L.w, L.h, n), dim3(256), 0, s, smooth, S.Lxy[i], L.w, L.h, fs,
                               (int)L.deriv_sigma, k);
            AKZ_LAUNCH_CHECK();
        }
        {
            // {Lx, Ly} of this level are complete on `s_main`: the determinant / candidate kernel is a side branch
            hipStream_t s_main = s;
            if (det_side) {
                AKZ_HIP(hipEventRecord(c->ev_level[i], s_main));
                AKZ_HIP(hipStreamWaitEvent(c->stream_det, c->ev_level[i], 0));
            }
            hipStream_t s = det_side ? c->stream_det : s_main;
            CandParams cp;
            cp.thr = (float)c->cfg.detector_threshold;
            cp.border = L.cand_border;
            cp.level = (uint32_t)i;
            cp.cap = c->max_cand;
            cp.x_lo = L.cand_x_lo; cp.x_hi = L.cand_x_hi; cp.y_lo = L.cand_y_lo; cp.y_hi = L.cand_y_hi;
            dim3 grid2(akz_div_up(L.w, 64), akz_div_up(L.h, 32), n);
            float* ldet_out = c->keep_all ? S.Ldet[i] : nullptr;   // refinement reads the candidates' own 3x3 values
#define AKZ_D2(SGV)                                                                                                  \
    AKZ_LAUNCH((k_deriv_second_cand<SGV>), grid2, dim3(256), 0, s, S.Lxy[i], ldet_out, L.w, L.h, fs,          \
                       (int)L.deriv_sigma, k, L.sigma_quat, cp, (CandU*)S.d_cand_u, S.d_ncand, c->d_err)
#define AKZ_D2P(SGV)                                                                                                 \
    AKZ_LAUNCH((k_deriv_second_cand2<SGV, kDTH, 256>),                                                       \
                       dim3(akz_div_up(L.w, 64), akz_div_up(L.h, kDTH), (n + 1) / 2), dim3(256), 0, s, S.Lxy[i],     \
                       ldet_out, L.w, L.h, fs, n, k, L.sigma_quat, cp, (CandU*)S.d_cand_u, S.d_ncand, c->d_err)
            const bool pair2 = (L.w & 3) == 0 && c->front_pair;
            const bool t_det_on = L.deriv_sigma >= 2 && L.deriv_sigma <= 4;
            if (t_det_on) akz_timer_begin(c, AKZ_T_DET_SG2 + (int)L.deriv_sigma - 2, s);
            // streaming kernel: a wave per (band of columns, segment of rows, frame); segments sized so that the
            // launch holds a few waves per SIMD of the chip, never shorter than 32 rows (each segment re-reads
            // 2 sigma + 2 rows of warm-up)
#define AKZ_DS(SGV)                                                                                                  \
    {                                                                                                                \
        const int nb = akz_div_up(L.w, det_stream_band(SGV));                                                        \
        int nseg = akz_div_up(c->det_stream_waves, nb * n);                                                          \
        const int max_seg = akz_div_up(L.h, 32);                                                                     \
        nseg = nseg < 1 ? 1 : (nseg > max_seg ? max_seg : nseg);                                                     \
        const int seg_rows = akz_div_up(L.h, nseg);                                                                  \
        nseg = akz_div_up(L.h, seg_rows);                                                                            \
        if (ldet_out)                                                                                                \
            AKZ_LAUNCH((k_det_stream<SGV, true>), dim3(akz_div_up(nb * nseg, 4), n), dim3(256), 0, s, S.Lxy[i],  \
                               ldet_out, L.w, L.h, fs, k, L.sigma_quat, cp, (CandU*)S.d_cand_u, S.d_ncand, c->d_err, nb, \
                               seg_rows);                                                                            \
        else                                                                                                         \
            AKZ_LAUNCH((k_det_stream<SGV, false>), dim3(akz_div_up(nb * nseg, 4), n), dim3(256), 0, s, S.Lxy[i], \
                               ldet_out, L.w, L.h, fs, k, L.sigma_quat, cp, (CandU*)S.d_cand_u, S.d_ncand, c->d_err, nb, \
                               seg_rows);                                                                            \
    }
            // the streaming kernel wants a few thousand waves (band x 32-row segment x frame); a single frame or the
            // smallest levels of a small batch keep the tile kernel (single 1080p frame: 1.85 vs 2.02 ms)
            const int sbw = det_stream_band(L.deriv_sigma >= 2 && L.deriv_sigma <= 4 ? (int)L.deriv_sigma : 2);
            const size_t stream_waves = (size_t)akz_div_up(L.w, sbw) * (size_t)akz_div_up(L.h, 32) * (size_t)n;
            const bool stream = c->stream_kernels && t_det_on && L.w >= 8 && stream_waves >= c->stream_min_waves;
            switch (L.deriv_sigma) {
            case 2: if (stream) { AKZ_DS(2); } else if (pair2) { AKZ_D2P(2); } else AKZ_D2(2); break;
            case 3: if (stream) { AKZ_DS(3); } else if (pair2) { AKZ_D2P(3); } else AKZ_D2(3); break;
            case 4: if (stream) { AKZ_DS(4); } else if (pair2) { AKZ_D2P(4); } else AKZ_D2(4); break;
            default: AKZ_D2(0); break;
            }
#undef AKZ_DS
#undef AKZ_D2
#undef AKZ_D2P
            AKZ_LAUNCH_CHECK();
            if (t_det_on) akz_timer_end(c, AKZ_T_DET_SG2 + (int)L.deriv_sigma - 2, s, 1, (uint64_t)fs * n);
        }
    }
    if (det_side) {
        AKZ_HIP(hipEventRecord(c->ev_det_done, c->stream_det));
        AKZ_HIP(hipStreamWaitEvent(s, c->ev_det_done, 0));
    }
    {
        uint32_t np2 = 1;
        while (np2 < c->max_cand) np2 <<= 1;
        const uint32_t lds_keys = np2 < kAkzLdsSortKeys ? np2 : kAkzLdsSortKeys;
        CandLevelRows cand_rows;
        for (int i = 0; i < kAkzMaxLevels; ++i) cand_rows.h[i] = i < nlev ? (uint32_t)P.levels[i].h : 0u;
        if (n <= kLatencyFrames)
            AKZ_LAUNCH(k_cand_rank, dim3(akz_div_up((int)c->max_cand, 32), nlev, n), dim3(256), 0, s, (const CandU*)S.d_cand_u,
                               S.d_ncand, c->max_cand, S.d_cand, S.d_cand_nb);
        else
            AKZ_LAUNCH(k_cand_sort, dim3(nlev, n), dim3(1024),
                               std::max<size_t>(sizeof(unsigned long long) * lds_keys, kCandRadixLdsBytes), s, (const CandU*)S.d_cand_u,
                               S.d_ncand, c->max_cand, S.d_cand, S.d_cand_nb, S.d_keys_cand, np2, lds_keys, cand_rows);
        AKZ_LAUNCH_CHECK();
    }
    akz_timer_end(c, AKZ_T_SCALE_SPACE, s, 0, (uint64_t)n);
    (void)fed_launches;
    (void)fed_units;
    return AKZ_OK;
}

int32_t AKZ_SS(akz_run_scale_space)(akz_ctx* c, const void* d_imgs, int fmt, int n)
{
    AkzTimerScope timer_scope(c);
    if (fmt == AKZ_FMT_U8) return scale_space_impl<uint8_t>(c, (const uint8_t*)d_imgs, n);
    if (fmt == AKZ_FMT_U16) return scale_space_impl<uint16_t>(c, (const uint16_t*)d_imgs, n);
    return scale_space_impl<float>(c, (const float*)d_imgs, n);
}
