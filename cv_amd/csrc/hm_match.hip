// hm_match.hip — brute-force Hamming 2-NN matcher for 512-bit descriptors on gfx950
// (SURVEY.md §8a rows M1, M2).
//
// Replaces, for BitArray<64> descriptors (paths relative to the rust-cv/cv checkout):
//   space::LinearKnn{metric: Hamming, iter}.knn(q, 2)   call sites akaze/tests/estimate_pose.rs:82-88,
//                                                        tutorial-code/chapter5-…/src/main.rs:155-161
//   matching / symmetric_matching                       tutorial ch5 main.rs:154-200, cv-sfm/src/lib.rs:3097-3133
//   match_descriptors (Lowe ratio)                      akaze/tests/estimate_pose.rs:78-97
//
// Kernel shape: integer VALU work, not a GEMM.  A lane owns kQPT query descriptors in VGPRs
// (16 dwords each); a workgroup stages a tile of targets in LDS and every lane walks the tile with
// broadcast ds_read_b128, 16 x (v_xor_b32 + v_bcnt_u32_b32 accumulate) per distance.  (Fetching the
// wave-uniform target through scalar loads instead was measured 24 % slower: rocprof r01, k_knn2 6.18 vs 5.0 ms
// per 64 frame pairs — s_load latency is not hidden at two targets in flight.)  The running
// (nearest, second) pair per query is two packed keys  distance << 22 | target_index  updated with
// v_min_u32 / v_med3_u32: smaller key == smaller (distance, index), which is exactly LinearKnn's
// "lowest index wins ties" order (space 0.17: partition_point(d <= new) insertion).
#include <algorithm>
#include <atomic>

#include <hip/hip_ext.h>

#include <unordered_map>

#include "akz_common.h"

namespace {

constexpr int kQPT = 2;          // queries per lane
constexpr int kBlock = 256;
constexpr int kQPB = kQPT * kBlock;  // queries per workgroup
constexpr int kTile = 256;       // targets per LDS tile (16 KB)
constexpr uint32_t kIdxBits = 22;

struct HmProb {            // one (queries -> targets) problem
    const uint4* q;        // query descriptors (4 x uint4 each)
    const uint32_t* nq;    // device count
    uint32_t q_cap;
    const uint4* t;
    const uint32_t* nt;
    uint32_t t_cap;
    akz_neighbor* out;     // [q_cap][2]
};

__global__ __launch_bounds__(kBlock) void k_knn2(const HmProb* __restrict__ probs)
{
    __shared__ uint4 s_t[kTile * 4];
    const HmProb P = probs[blockIdx.y];
    uint32_t nq = min(*P.nq, P.q_cap), nt = min(*P.nt, P.t_cap);
    const uint32_t q0 = blockIdx.x * kQPB;
    if (q0 >= nq) return;
    uint32_t qv[kQPT][16];
    uint32_t k0[kQPT], k1[kQPT];
#pragma unroll
    for (int r = 0; r < kQPT; ++r) {
        uint32_t qi = q0 + r * kBlock + threadIdx.x;
        uint32_t qc = qi < nq ? qi : nq - 1;  // clamp: idle lanes redo the last query, results discarded
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            uint4 x = P.q[(size_t)qc * 4 + v];
            qv[r][4 * v + 0] = x.x;
            qv[r][4 * v + 1] = x.y;
            qv[r][4 * v + 2] = x.z;
            qv[r][4 * v + 3] = x.w;
        }
        k0[r] = 0xFFFFFFFFu;
        k1[r] = 0xFFFFFFFFu;
    }
    for (uint32_t t0 = 0; t0 < nt; t0 += kTile) {
        uint32_t cnt = min((uint32_t)kTile, nt - t0);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < cnt * 4; i += kBlock) s_t[i] = P.t[(size_t)t0 * 4 + i];
        __syncthreads();
        for (uint32_t j = 0; j < cnt; ++j) {
            uint4 a = s_t[j * 4 + 0], b = s_t[j * 4 + 1], c = s_t[j * 4 + 2], d = s_t[j * 4 + 3];
            uint32_t tv[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
#pragma unroll
            for (int r = 0; r < kQPT; ++r) {
                uint32_t dist = 0;
#pragma unroll
                for (int v = 0; v < 16; ++v) dist += __popc(qv[r][v] ^ tv[v]);
                uint32_t key = (dist << kIdxBits) | (t0 + j);
                // sorted pair (k0 <= k1): new k1 = median(k0, k1, key), new k0 = min(k0, key)
                uint32_t lo = min(k0[r], key);
                uint32_t hi = max(k0[r], key);
                k1[r] = min(k1[r], hi);
                k0[r] = lo;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < kQPT; ++r) {
        uint32_t qi = q0 + r * kBlock + threadIdx.x;
        if (qi < nq) {
            akz_neighbor n0 = {k0[r] & ((1u << kIdxBits) - 1u), k0[r] >> kIdxBits};
            akz_neighbor n1 = {k1[r] & ((1u << kIdxBits) - 1u), k1[r] >> kIdxBits};
            P.out[(size_t)qi * 2 + 0] = n0;
            P.out[(size_t)qi * 2 + 1] = n1;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// MFMA formulation of the same 2-NN search.  With the descriptor bits recoded as int8 +1 / -1,
//   sum_k q_k * t_k = 512 - 2 * hamming(q, t),
// so all distances between 32 queries and 32 targets are one 32x32x512 int8 contraction: sixteen
// v_mfma_i32_32x32x32_i8 (exact integer accumulate).  Queries are the B operand (output COLUMNS): lane l
// then holds, for query (l & 31), sixteen different targets per tile, and the running (nearest, second)
// keys are reduced inside the lane; the two half-waves are merged once at the end.  Per distance the VALU
// does ~6 operations (key build + min/max) instead of 36, the xor/popcount work moves to the matrix pipe.
// k_expand writes the +-1 recoding (512 B per descriptor) once per descriptor block.
// Measured (rocprof, 128 problems of ~5070 x 5070, profiles/): VALU kernel k_knn2 5.0 ms; this kernel with
// fragment-shaped global loads 6.9 ms (TA-bound); with LDS-staged 32-row tiles and 8 waves per block 2.7 ms;
// 64-row tiles with two accumulator chains 3.4 ms (slower: LDS footprint halves the resident blocks).
typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef int v16i32 __attribute__((ext_vector_type(16)));

struct HmExpandJob {
    const uint32_t* src;   // [cap][16] descriptor words
    const uint32_t* count; // device count
    uint32_t cap;
    uint32_t* dst;         // [cap][128] words of +-1 bytes
};

__global__ __launch_bounds__(256) void k_expand(const HmExpandJob* __restrict__ jobs)
{
    const HmExpandJob J = jobs[blockIdx.y];
    const uint32_t n = min(*J.count, J.cap);
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;  // one thread per (descriptor, 32-bit word)
    const uint32_t d = t >> 4, wd = t & 15u;
    if (d >= n) return;
    const uint32_t bits = J.src[(size_t)d * 16 + wd];
    uint4 o[2];
    uint32_t* ow = reinterpret_cast<uint32_t*>(o);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        uint32_t nib = (bits >> (4 * g)) & 0xFu;
        // spread 4 bits to 4 bytes (bit i -> byte i), then map 0 -> 0xFF (-1), 1 -> 0x01 (+1)
        uint32_t sp = (nib * 0x00204081u) & 0x01010101u;
        ow[g] = ((sp ^ 0x01010101u) * 0xFFu) | sp;
    }
    uint4* dst = reinterpret_cast<uint4*>(J.dst + ((size_t)d * 128 + (size_t)wd * 8));
    dst[0] = o[0];
    dst[1] = o[1];
}

struct HmProbX {           // like HmProb, on the +-1 recoded descriptors
    const uint32_t* q;     // [q_cap][128] words
    const uint32_t* nq;
    uint32_t q_cap;
    const uint32_t* t;
    const uint32_t* nt;
    uint32_t t_cap;
    akz_neighbor* out;
};

constexpr int kMfmaBlock = 512;   // 8 waves x 32 queries

// insert v into the ascending list l[0..K): K-1 min/max pairs and one min
template <int K>
__device__ __forceinline__ void topk_insert(int (&l)[K], int v)
{
    if constexpr (K == 2) {
        // l[0] <= l[1]: the new second is the median of (l0, l1, v) — one v_med3_i32 and one v_min_i32
        int m;
        asm("v_med3_i32 %0, %1, %2, %3" : "=v"(m) : "v"(l[0]), "v"(l[1]), "v"(v));
        l[0] = min(l[0], v);
        l[1] = m;
    } else if constexpr (K == 3) {
        // l0 <= l1 <= l2: the new second is the median of (l0, l1, v), the new third the median of (l1, l2, v) — three
        // instructions instead of the five of the min / max ladder
        int m1, m2;
        asm("v_med3_i32 %0, %1, %2, %3" : "=v"(m1) : "v"(l[0]), "v"(l[1]), "v"(v));
        asm("v_med3_i32 %0, %1, %2, %3" : "=v"(m2) : "v"(l[1]), "v"(l[2]), "v"(v));
        l[0] = min(l[0], v);
        l[1] = m1;
        l[2] = m2;
    } else {
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int lo = min(l[i], v);
            v = max(l[i], v);
            l[i] = lo;
        }
    }
}

// Two values at once into the ascending pair l[0] <= l[1] in three instructions instead of four: the new smallest is
// min3(l0, a, b); the new second is min(l1, median(l0, a, b)) — if l1 is below the median of {l0, a, b} then l0 is that
// triple's smallest (l0 <= l1) and l1 the second of all four, otherwise the median is.  On this part an integer VALU
// instruction behind an FP4 MFMA costs the SIMD ~5.6 cycles of matrix time (they do not overlap at two waves per SIMD,
// tools/ubench/mfma_fp4_rate.hip), so the key epilogue is priced per instruction.
template <int K>
__device__ __forceinline__ void topk_insert2(int (&l)[K], int a, int b)
{
    if constexpr (K == 2) {
        int m, lo;
        asm("v_med3_i32 %0, %1, %2, %3" : "=v"(m) : "v"(l[0]), "v"(a), "v"(b));
        asm("v_min3_i32 %0, %1, %2, %3" : "=v"(lo) : "v"(l[0]), "v"(a), "v"(b));
        l[0] = lo;
        l[1] = min(l[1], m);
    } else {
        topk_insert<K>(l, a);
        topk_insert<K>(l, b);
    }
}
// an ascending pair g0 <= g1 into the ascending pair l: as above, one instruction less (min3 = min: g0 <= g1)
template <int K>
__device__ __forceinline__ void topk_merge2(int (&l)[K], const int (&g)[K])
{
    if constexpr (K == 2) {
        int m;
        asm("v_med3_i32 %0, %1, %2, %3" : "=v"(m) : "v"(l[0]), "v"(g[0]), "v"(g[1]));
        l[0] = min(l[0], g[0]);
        l[1] = min(l[1], m);
    } else {
#pragma unroll
        for (int i = 0; i < K; ++i) topk_insert<K>(l, g[i]);
    }
}

// One 32-target x 32-query tile: 16 chained MFMAs (K = 512 = 16 x 32) with the LDS fragment reads kept
// four deep in flight, then the (distance, row) keys and the tile's KNN smallest, merged into k[].
// D layout: column = lane & 31 (query), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (target).
// nacc = -dot (the resident query fragments are negated).  The unsigned key (512 - dot) << 21 | row is
// carried as the signed value key - (512 << 21) = (nacc << 21) | row: one v_lshl_or_b32 per accumulator
// register with inline constants only; the lane's row offset (t0 + 4 * half) is added to the tile's
// smallest afterwards (row bits never carry: row < 2^21).  TAIL (the last, partial tile) masks rows >= nt.
template <bool TAIL, int KNN>
__device__ __forceinline__ void knn_keys(const v16i32& nacc, uint32_t t0, uint32_t half, uint32_t nt, int (&k)[KNN])
{
    const int row0 = (int)(t0 + 4u * half);
    int l[KNN];
#pragma unroll
    for (int i = 0; i < KNN; ++i) l[i] = 0x7FFFFFFF;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int off = (r & 3) + 8 * (r >> 2);
        int key = (int)(((uint32_t)nacc[r] << 21) | (uint32_t)off);
        if (TAIL) key = (uint32_t)(row0 + off) < nt ? key : 0x7FFFFFFF - row0;
        topk_insert<KNN>(l, key);
    }
    // 16 keys went in, so every l[i] is a key (masked ones become INT_MAX again): no overflow below
#pragma unroll
    for (int i = 0; i < KNN; ++i) topk_insert<KNN>(k, l[i] + row0);
}

// knn_keys for accumulators that already ARE tile-local keys (the FP4 kernel): with the targets' block scale at
// 2^5 and the bias block at 1.5 * 2^23 + off(r), register r holds the f32 1.5 * 2^23 + 32 nacc + off(r), whose bit
// pattern 0x4B400000 + 32 nacc + off orders like (nacc, off) as a signed integer — no key has to be built.  The
// tile's KNN smallest become global keys (nacc << 21) | (row0 + off): `<< 16` moves 32 nacc to bit 21 and drops
// the constant (its lowest set bit is bit 22), the low five bits are off, and off never uses bit 2, so
// `| row0` (row0 = t0 + 4 half) adds it.  Three instructions per survivor instead of one per register.
// tile-local key -> global key: three instructions (clear the row tag, shift the rest to bit 21 and OR the row in)
__device__ __forceinline__ int knn_global_key(int l, int row0)
{
    const int low = (l & 31) | row0;                                  // v_and_or_b32
    return (int)((uint32_t)(l & ~31) << 16) | low;                    // v_and_b32, v_lshl_or_b32: bits 16..20 stay clear
}

template <bool TAIL, int KNN>
__device__ __forceinline__ void knn_keys_tagged(const v16i32& raw, uint32_t t0, uint32_t half, uint32_t nt, int (&k)[KNN])
{
    const int row0 = (int)(t0 + 4u * half);
    int l[KNN];
#pragma unroll
    for (int i = 0; i < KNN; ++i) l[i] = 0x7FFFFFFF;
    int key[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int off = (r & 3) + 8 * (r >> 2);
        key[r] = raw[r];
        if (TAIL) key[r] = (uint32_t)(row0 + off) < nt ? key[r] : 0x7FFFFFFF;
    }
#pragma unroll
    for (int r = 0; r < 16; r += 2) topk_insert2<KNN>(l, key[r], key[r + 1]);
    int g[KNN];
#pragma unroll
    for (int i = 0; i < KNN; ++i) {
        g[i] = knn_global_key(l[i], row0);
        if (TAIL) g[i] = l[i] == 0x7FFFFFFF ? 0x7FFFFFFF : g[i];
    }
    topk_merge2<KNN>(k, g);          // (g ascending: the rekeying is monotone, and INT_MAX stays last)
}

template <int KNN>
__device__ __forceinline__ void knn_tile(const uint4* __restrict__ tp, const v4i32 (&qb)[16], uint32_t t0,
                                         uint32_t half, uint32_t nt, int (&k)[KNN])
{
    v16i32 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint4 f[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = tp[2 * i];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        uint4 av = f[i & 3];
        v4i32 a = {(int)av.x, (int)av.y, (int)av.z, (int)av.w};
        if (i + 4 < 16) f[i & 3] = tp[2 * (i + 4)];
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, qb[i], acc, 0, 0, 0);
    }
    // keep the fragment reads four deep ahead of the dependent MFMA chain
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    if (t0 + 32u <= nt) knn_keys<false, KNN>(acc, t0, half, nt, k);
    else knn_keys<true, KNN>(acc, t0, half, nt, k);
}

// LinearKnn::knn(q, KNN) for every query of every problem: out[q][KNN], ascending (distance, index).
// Neighbours that do not exist (nt < KNN) come back as {index 2^22 - 1, distance 1023}.
template <int KNN>
__global__ __launch_bounds__(kMfmaBlock, 4) void k_knn_mfma(const HmProbX* __restrict__ probs)
{
    // target tile: 32 descriptors x 512 B, rows padded to 33 x 16 B so the 16-byte fragment reads of the
    // 32 rows fall on distinct bank groups; double-buffered, filled with full 512-B-row coalesced loads
    constexpr int RS = 33;
    __shared__ uint4 s_t[2][32 * RS];
    // XCD-aware order (workgroup b runs on XCD b % 8, observed): the query blocks of one problem stream the same
    // target set, so they get ids that are congruent mod 8 and share one L2 instead of filling all eight
    uint32_t bx, by;
    {
        const uint32_t nwg = gridDim.x * gridDim.y, orig = blockIdx.x + gridDim.x * blockIdx.y;
        const uint32_t xcd = orig & 7u, q = nwg >> 3, r = nwg & 7u;
        const uint32_t t = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + (orig >> 3);
        by = t / gridDim.x;
        bx = t - by * gridDim.x;
    }
    const HmProbX P = probs[by];
    const uint32_t nq = min(*P.nq, P.q_cap), nt = min(*P.nt, P.t_cap);
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t qblk = bx * 256u;
    if (qblk >= nq) return;                             // whole block
    const uint32_t q0 = qblk + wv * 32u;                // 32 queries (one column block) per wave
    const bool wave_on = q0 < nq;                       // idle waves still help staging and hit the barriers
    const uint32_t col = lane & 31u, half = lane >> 5;
    int k[KNN];                                         // signed keys, see knn_keys
#pragma unroll
    for (int i = 0; i < KNN; ++i) k[i] = 0x7FFFFFFF;
    if (nt > 0) {
        // B fragments: query (q0 + col), bytes [32k + 16*half, +16) of its 512 -> 16 x v4i32, resident
        v4i32 qb[16];
        {
            uint32_t qi = min(q0 + col, nq - 1u);
            const v4i32* qp = reinterpret_cast<const v4i32*>(P.q + (size_t)qi * 128) + half;
#pragma unroll
            for (int i = 0; i < 16; ++i) qb[i] = qp[2 * i] ^ (int)0xFEFEFEFE;   // +-1 bytes negated (see knn_keys)
        }
        const uint4* tg = reinterpret_cast<const uint4*>(P.t);  // 32 uint4 per descriptor
        // staging map: thread -> 2 x (row, 16-byte column): idx = tid + 512 i, row = idx >> 5, c = idx & 31
        const uint32_t srow0 = threadIdx.x >> 5, srow1 = (threadIdx.x + 512u) >> 5, scol = threadIdx.x & 31u;
        uint4 st0, st1;
        st0 = tg[(size_t)min(srow0, nt - 1u) * 32 + scol];
        st1 = tg[(size_t)min(srow1, nt - 1u) * 32 + scol];
        s_t[0][srow0 * RS + scol] = st0;
        s_t[0][srow1 * RS + scol] = st1;
        int buf = 0;
        for (uint32_t t0 = 0; t0 < nt; t0 += 32u) {
            __syncthreads();                                 // tile `buf` is complete; tile `buf^1` is free
            const bool more = t0 + 32u < nt;
            if (more) {                                      // global loads in flight during the MFMAs
                st0 = tg[(size_t)min(t0 + 32u + srow0, nt - 1u) * 32 + scol];
                st1 = tg[(size_t)min(t0 + 32u + srow1, nt - 1u) * 32 + scol];
            }
            if (wave_on) knn_tile<KNN>(&s_t[buf][col * RS + half], qb, t0, half, nt, k);
            if (more) {
                s_t[buf ^ 1][srow0 * RS + scol] = st0;
                s_t[buf ^ 1][srow1 * RS + scol] = st1;
            }
            buf ^= 1;
        }
    }
    if (!wave_on) return;
    // merge the two half-waves' sorted lists, back to the unsigned key (INT_MAX = no neighbour -> all ones)
    int o[KNN];
#pragma unroll
    for (int i = 0; i < KNN; ++i) o[i] = __shfl_xor(k[i], 32);
#pragma unroll
    for (int i = 0; i < KNN; ++i) topk_insert<KNN>(k, o[i]);
    const uint32_t qi = q0 + col;
    if (half == 0 && qi < nq) {
#pragma unroll
        for (int i = 0; i < KNN; ++i) {
            const uint32_t m = k[i] == 0x7FFFFFFF ? 0xFFFFFFFFu : (uint32_t)k[i] + (512u << 21);
            akz_neighbor nb = {m & ((1u << kIdxBits) - 1u), m >> kIdxBits};
            P.out[(size_t)qi * KNN + i] = nb;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// FP4 form of the same contraction.  +1 / -1 are exact E2M1 values (0x2 / 0xA), so the block-scaled
// v_mfma_scale_f32_32x32x64_f8f6f4 with unit scales (E8M0 0x7F) computes the same integer dot products in f32
// (|sum| <= 512, exact), at twice the K per instruction and half the bytes per descriptor: 8 MFMAs and 8 KB of
// LDS fragments per 32 x 32 tile instead of 16 and 16 KB.  The accumulators start at 1.5 * 2^23, where one ulp is
// 1: the f32 bit pattern is 0x4B400000 + nacc, and the key's `<< 21` drops the constant (its lowest set bit is
// bit 22), so knn_keys takes the raw bits unchanged.  (tools/ubench/fp4_probe.hip checks both facts on the GPU.)
typedef int v8i32 __attribute__((ext_vector_type(8)));
typedef float v16f32 __attribute__((ext_vector_type(16)));

// descriptor bits -> E2M1 nibbles (bit b -> nibble b: 1 -> 0x2 = +1.0, 0 -> 0xA = -1.0), 256 B per descriptor
__global__ __launch_bounds__(256) void k_expand4(const HmExpandJob* __restrict__ jobs)
{
    const HmExpandJob J = jobs[blockIdx.y];
    const uint32_t n = min(*J.count, J.cap);
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;  // one thread per (descriptor, 32-bit word)
    const uint32_t d = t >> 4, wd = t & 15u;
    if (d >= n) return;
    const uint32_t bits = J.src[(size_t)d * 16 + wd];
    uint4 o;
    uint32_t* ow = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint32_t b = (bits >> (8 * g)) & 0xFFu;
        // spread 8 bits to 8 nibbles (bit i -> bit 4 i): halves, pairs, single bits
        uint32_t sp = (b | (b << 12)) & 0x000F000Fu;
        sp = (sp | (sp << 6)) & 0x03030303u;
        sp = (sp | (sp << 3)) & 0x11111111u;
        ow[g] = 0xAAAAAAAAu ^ (sp << 3);                 // set bit -> clear the sign: 0xA -> 0x2
    }
    reinterpret_cast<uint4*>(J.dst + (size_t)d * 64)[wd] = o;
}

// The 8 chained MFMAs of tile `tp` into `out`, interleaved with the key epilogue of the PREVIOUS tile's accumulators
// (`prev`, always a full tile): a dependent MFMA issues every ~32 cycles, the seven or so VALU instructions of the
// epilogue that fit in between are independent of it, so the matrix pipe and the VALU of one wave overlap.
template <int KNN, bool EPI, int STRIDE = 2>    // STRIDE: uint4 between a lane's consecutive fragments in the LDS image
__device__ __forceinline__ void knn_chain4(const uint4* __restrict__ tp, const v4i32 (&qb)[8], const v16f32& bias,
                                           v16f32& out, const v16f32& prev, uint32_t tprev, uint32_t half, int (&k)[KNN])
{
    uint4 f[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = tp[STRIDE * i];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint4 av = f[i & 3];
        const v8i32 a = {(int)av.x, (int)av.y, (int)av.z, (int)av.w, 0, 0, 0, 0};
        const v8i32 b = {qb[i][0], qb[i][1], qb[i][2], qb[i][3], 0, 0, 0, 0};
        if (i + 4 < 8) f[i & 3] = tp[STRIDE * (i + 4)];
        // the chain starts from the resident bias block (D != C): no per-tile accumulator initialisation
        // block scales: targets 2^5 (E8M0 0x84), queries 1 (0x7F): the accumulators are tile-local keys, see
        // knn_keys_tagged
        if (i == 0) out = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, bias, 4, 4, 0, 0x84848484, 0, 0x7F7F7F7F);
        else out = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, out, 4, 4, 0, 0x84848484, 0, 0x7F7F7F7F);
    }
    if (EPI) knn_keys_tagged<false, KNN>(__builtin_bit_cast(v16i32, prev), tprev, half, 0u, k);
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (EPI) __builtin_amdgcn_sched_group_barrier(0x002, KNN == 3 ? 10 : 6, 0);
        if (i < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
}

// k_knn_mfma on the E2M1 recoding: same block shape (8 waves x 32 queries), same staging scheme (one 16-byte
// load per thread and tile), same key arithmetic; the tile loop is software-pipelined by one tile (two
// accumulator sets, loop unrolled by two so they never move).
template <int KNN>
__global__ __launch_bounds__(kMfmaBlock, 4) void k_knn_mfma4(const HmProbX* __restrict__ probs)
{
    constexpr int RS = 17;                               // 16 chunks per descriptor, padded
    __shared__ uint4 s_t[2][32 * RS];
    uint32_t bx, by;
    {
        const uint32_t nwg = gridDim.x * gridDim.y, orig = blockIdx.x + gridDim.x * blockIdx.y;
        const uint32_t xcd = orig & 7u, q = nwg >> 3, r = nwg & 7u;
        const uint32_t t = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + (orig >> 3);
        by = t / gridDim.x;
        bx = t - by * gridDim.x;
    }
    const HmProbX P = probs[by];
    const uint32_t nq = min(*P.nq, P.q_cap), nt = min(*P.nt, P.t_cap);
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t qblk = bx * 256u;
    if (qblk >= nq) return;
    const uint32_t q0 = qblk + wv * 32u;
    const bool wave_on = q0 < nq;
    const uint32_t col = lane & 31u, half = lane >> 5;
    int k[KNN];
#pragma unroll
    for (int i = 0; i < KNN; ++i) k[i] = 0x7FFFFFFF;
    if (nt > 0) {
        // B fragments: query (q0 + col), nibbles [64 j + 32 half, +32) of its 512 -> 8 x 16 bytes, resident, negated
        v4i32 qb[8];
        {
            const uint32_t qi = min(q0 + col, nq - 1u);
            const v4i32* qp = reinterpret_cast<const v4i32*>(P.q + (size_t)qi * 64) + half;
#pragma unroll
            for (int i = 0; i < 8; ++i) qb[i] = qp[2 * i] ^ (int)0x88888888;
        }
        const uint4* tg = reinterpret_cast<const uint4*>(P.t);  // 16 uint4 per descriptor
        const uint32_t srow = threadIdx.x >> 4, scol = threadIdx.x & 15u;
        const uint4* tlane = &s_t[0][col * RS + half];
        uint4 st = tg[(size_t)min(srow, nt - 1u) * 16 + scol];
        s_t[0][srow * RS + scol] = st;
        // 1.5 * 2^23 in sixteen registers that stay put (opaque to the compiler, or it re-creates them per tile)
        v16f32 bias;
#pragma unroll
        for (int i = 0; i < 16; ++i) bias[i] = 12582912.0f + (float)((i & 3) + 8 * (i >> 2));   // + off(r): the row tag
        asm volatile("" : "+v"(bias));
        v16f32 acc0 = bias, acc1 = bias;
        // one pipeline step: tile at t0 (in buffer `buf`) -> `out`, epilogue of `prev` (tile t0 - 32)
        uint32_t t0 = 0;
        int buf = 0;
        bool more;
#define HM_STEP(EPI, out, prev)                                                                                  \
        __syncthreads();                                                                                         \
        more = t0 + 32u < nt;                                                                                    \
        if (more) st = tg[(size_t)min(t0 + 32u + srow, nt - 1u) * 16 + scol];                                    \
        if (wave_on) knn_chain4<KNN, EPI>(tlane + buf * (32 * RS), qb, bias, out, prev, t0 - 32u, half, k);      \
        if (more) s_t[buf ^ 1][srow * RS + scol] = st;                                                           \
        buf ^= 1;
        // the last tile's keys (the only tile that can be partial) are taken where its accumulator set is known
#define HM_LAST(acc)                                                                                             \
        if (wave_on) {                                                                                           \
            const v16i32 nacc = __builtin_bit_cast(v16i32, acc);                                                 \
            if (t0 + 32u <= nt) knn_keys_tagged<false, KNN>(nacc, t0, half, nt, k);                              \
            else knn_keys_tagged<true, KNN>(nacc, t0, half, nt, k);                                              \
        }
        HM_STEP(false, acc0, acc1)
        if (!more) {
            HM_LAST(acc0)
        } else {
            for (;;) {
                t0 += 32u;
                HM_STEP(true, acc1, acc0)
                if (!more) {
                    HM_LAST(acc1)
                    break;
                }
                t0 += 32u;
                HM_STEP(true, acc0, acc1)
                if (!more) {
                    HM_LAST(acc0)
                    break;
                }
            }
        }
#undef HM_LAST
#undef HM_STEP
    }
    if (!wave_on) return;
    int o[KNN];
#pragma unroll
    for (int i = 0; i < KNN; ++i) o[i] = __shfl_xor(k[i], 32);
#pragma unroll
    for (int i = 0; i < KNN; ++i) topk_insert<KNN>(k, o[i]);
    const uint32_t qi = q0 + col;
    if (half == 0 && qi < nq) {
#pragma unroll
        for (int i = 0; i < KNN; ++i) {
            const uint32_t m = k[i] == 0x7FFFFFFF ? 0xFFFFFFFFu : (uint32_t)k[i] + (512u << 21);
            akz_neighbor nb = {m & ((1u << kIdxBits) - 1u), m >> kIdxBits};
            P.out[(size_t)qi * KNN + i] = nb;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The wide FP4 kernel: 64 resident queries per wave and the target tiles brought in by LDS-DMA.
//
// k_knn_mfma4 keeps the matrix pipe busy 61 % of the time (SQ_VALU_MFMA_BUSY_CYCLES against SQ_WAVE_CYCLES,
// profiles/r04_pmc_matcher.txt) and no pipelining of its staging changes that (an LDS-DMA ring with four tiles of
// prefetch and half the barriers measured the same 3.05 ms per 256 frame pairs): its limit is LDS READ BANDWIDTH.  A
// wave multiplies a 32-target x 64-nibble fragment (1 KB from LDS) by 32 resident queries per MFMA; sixteen waves per
// CU fetch 8 KB per tile each, 128 KB per 1 024 MFMA cycles = the LDS's 128 B per clock, all of it.  Here a wave keeps
// 64 queries (two column blocks) in registers and every fragment feeds two MFMAs — half the LDS bytes per MAC — at two
// waves per SIMD (256 registers): the two column blocks' dependent MFMA chains interleave, so one wave alone keeps
// the pipe issuing, and the key epilogue of the previous tile rides between them as before.
//
// The target tiles come by global_load_lds_dwordx4 (gfx950): no staging registers, no ds_write, three STAGES of two
// tiles in a ring, the DMA of stage s + 2 issued when stage s starts, one barrier per stage, a counted vmcnt for the
// stage about to be read.  The DMA writes lane-linearly (M0 base + 16 lane), so the image cannot be padded per row;
// bank conflicts are avoided by WHAT each lane fetches: a 1 KB group holds four rows, lane l the chunk (l >> 2) of
// row (l & 3) — the slot of (row, chunk) is 4 chunk + (row & 3) — and the groups are 64 B apart.  A fragment read
// (lane <-> row, all lanes the same chunk) then touches, per eight lanes, eight different 16-byte columns of the
// 128-byte LDS row, and a lane's eight fragments are 128 B apart: immediate offsets.
constexpr int kGSeg = 1024 + 64;               // bytes of a 4-row group
constexpr int kGTile = 8 * kGSeg;              // 8 704 B per 32-target tile
constexpr int kGStages = 3;                    // ring of stages (kGStages - 1 of them requested ahead) ...
constexpr int kGTps = 2;                       // ... of this many tiles each (even): 3 x 2: 52 224 B per block, two blocks per CU
constexpr int kWideBlock = 256;                // 4 waves x 64 queries

__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst)
{
    // (M0 is the compiler's: saved and restored; the statement is opaque to its wait-count bookkeeping, the waits are ours)
    uint32_t keep;
    lds_dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_dst);    // wave-uniform by construction; an SGPR for the compiler too
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// one tile for both column blocks: 8 fragments, 16 MFMAs (two interleaved dependent chains), and the key epilogue of
// the PREVIOUS tile's two accumulator blocks between them.  The order is written out and pinned (sched_barrier): two
// insertions of the previous tile's keys behind every MFMA — four to five VALU instructions in the ~32 cycles before the
// next MFMA of the other chain can issue.  (sched_group_barrier patterns, which k_knn_mfma4 uses, collapse here: the
// scheduler put the sixteen MFMAs back to back, one chain after the other, and the epilogue behind them.)
// The fragment stream does not stop at the tile's end: f[] arrives holding this tile's fragments 0..3 and leaves
// holding the NEXT tile's (tn, read behind steps 4..7), so no tile starts by waiting for LDS.  When the next tile
// belongs to the next stage, `hook` — the wait for that stage, the block's barrier and the next DMA — runs between
// steps 3 and 4, when this wave's last read of the current stage has been issued.
template <int KNN, bool EPI_, typename Hook>
__device__ __forceinline__ void knn_chain4w(const uint4* __restrict__ tp, const uint4* __restrict__ tn, uint4 (&f)[4], const v4i32 (&qb)[2][8], const v16f32& bias, v16f32& o0, v16f32& o1,
                                            const v16f32& p0, const v16f32& p1, uint32_t tprev, uint32_t half, int (&k0)[KNN],
                                            int (&k1)[KNN], Hook hook)
{
    constexpr int FS = 128 / 16;           // a lane's fragments are 128 B apart
    constexpr bool EPI = EPI_;
    const v16i32 r0 = __builtin_bit_cast(v16i32, p0), r1 = __builtin_bit_cast(v16i32, p1);
    int l0[KNN], l1[KNN];
#pragma unroll
    for (int i = 0; i < KNN; ++i) l0[i] = l1[i] = 0x7FFFFFFF;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i == 4) {
            hook();
            __builtin_amdgcn_sched_barrier(0);
        }
        const uint4 av = f[i & 3];
        const v8i32 a = {(int)av.x, (int)av.y, (int)av.z, (int)av.w, 0, 0, 0, 0};
        const v8i32 b0 = {qb[0][i][0], qb[0][i][1], qb[0][i][2], qb[0][i][3], 0, 0, 0, 0};
        const v8i32 b1 = {qb[1][i][0], qb[1][i][1], qb[1][i][2], qb[1][i][3], 0, 0, 0, 0};
        // block scales: targets 2^5 (E8M0 0x84), queries 1 (0x7F): the accumulators are tile-local keys (knn_keys_tagged)
        if (i == 0) o0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b0, bias, 4, 4, 0, 0x84848484, 0, 0x7F7F7F7F);
        else o0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b0, o0, 4, 4, 0, 0x84848484, 0, 0x7F7F7F7F);
        if (EPI) topk_insert2<KNN>(l0, r0[2 * i], r0[2 * i + 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (i == 0) o1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b1, bias, 4, 4, 0, 0x84848484, 0, 0x7F7F7F7F);
        else o1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b1, o1, 4, 4, 0, 0x84848484, 0, 0x7F7F7F7F);
        // four fragments (eight MFMAs) ahead: this tile's 4..7, then the next tile's 0..3
        // (no run-time branch in here: with control flow inside the chain the MFMAs sink below it and the pinned order is
        // gone — the last tile re-reads itself)
        if (i < 4) f[i & 3] = tp[FS * (i + 4)];
        else f[i & 3] = tn[FS * (i - 4)];
        if (EPI) topk_insert2<KNN>(l1, r1[2 * i], r1[2 * i + 1]);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (EPI) {
        // the tile's KNN smallest become global keys (knn_keys_tagged's second half)
        const int row0 = (int)(tprev + 4u * half);
        int g0[KNN], g1[KNN];
#pragma unroll
        for (int i = 0; i < KNN; ++i) {
            g0[i] = knn_global_key(l0[i], row0);
            g1[i] = knn_global_key(l1[i], row0);
        }
        topk_merge2<KNN>(k0, g0);
        topk_merge2<KNN>(k1, g1);
    }
}

// Two accumulator sets: the previous tile's keys are taken inside the next tile's chain (two waves per SIMD).  (One set with the
// keys taken right behind the chain, three waves per SIMD at 168 registers, measured 3.32 against 2.82 ms per 256 frame pairs.)
template <int KNN>
__global__ __launch_bounds__(kWideBlock, 2) void k_knn_mfma4w(const HmProbX* __restrict__ probs)
{
    static_assert(kGStages >= 3 && kGTps == 2, "ring shape");
    __shared__ __attribute__((aligned(16))) unsigned char s_t[kGStages * kGTps * kGTile];
    uint32_t bx, by;
    {
        const uint32_t nwg = gridDim.x * gridDim.y, orig = blockIdx.x + gridDim.x * blockIdx.y;
        const uint32_t xcd = orig & 7u, q = nwg >> 3, r = nwg & 7u;
        const uint32_t t = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + (orig >> 3);
        by = t / gridDim.x;
        bx = t - by * gridDim.x;
    }
    const HmProbX P = probs[by];
    const uint32_t nq = min(*P.nq, P.q_cap), nt = min(*P.nt, P.t_cap);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t qblk = bx * 256u;
    if (qblk >= nq) return;                              // whole block
    const uint32_t q0 = qblk + wv * 64u;                 // 64 queries (two column blocks) per wave
    const bool wave_on = q0 < nq;                        // idle waves still carry their share of the tiles and hit the barriers
    const uint32_t col = lane & 31u, half = lane >> 5;
    int k0[KNN], k1[KNN];
#pragma unroll
    for (int i = 0; i < KNN; ++i) k0[i] = k1[i] = 0x7FFFFFFF;
    if (nt > 0) {
        // B fragments: queries (q0 + col) and (q0 + 32 + col), nibbles [64 j + 32 half, +32) of their 512, resident, negated
        v4i32 qb[2][8];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const uint32_t qi = min(q0 + 32u * cb + col, nq - 1u);
            const v4i32* qp = reinterpret_cast<const v4i32*>(P.q + (size_t)qi * 64) + half;
#pragma unroll
            for (int i = 0; i < 8; ++i) qb[cb][i] = qp[2 * i] ^ (int)0x88888888;
        }
        const unsigned char* tg = reinterpret_cast<const unsigned char*>(P.t);   // 256 B per descriptor
        const uint32_t lds0 = (uint32_t)(uintptr_t)s_t;                          // LDS byte address of the ring (the low half of its flat address)
        const uint32_t drow = 8u * wv + (lane & 3u), dchunk = lane >> 2;          // this lane's share: rows drow and drow + 4
        // request the tiles of stage `st` (rows past the end repeat the last descriptor: the counts stay uniform)
        auto issue = [&](uint32_t st) {
            const uint32_t slot = lds0 + (st % kGStages) * (uint32_t)(kGTps * kGTile) + 2u * wv * kGSeg;
#pragma unroll
            for (uint32_t h = 0; h < (uint32_t)kGTps; ++h)
#pragma unroll
                for (uint32_t g = 0; g < 2; ++g) {
                    const uint32_t row = min((st * kGTps + h) * 32u + drow + 4u * g, nt - 1u);
                    glds16(tg + (size_t)row * 256 + dchunk * 16, slot + h * kGTile + g * kGSeg);
                }
        };
        // the wave's query fragments must have arrived before the first DMA: the compiler's own vmcnt bookkeeping does not
        // see the DMAs, so none of its loads may be outstanding when they start
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (uint32_t st = 0; st + 1 < (uint32_t)kGStages; ++st) issue(st);
        v16f32 bias;
#pragma unroll
        for (int i = 0; i < 16; ++i) bias[i] = 12582912.0f + (float)((i & 3) + 8 * (i >> 2));   // + off(r): the row tag
        asm volatile("" : "+v"(bias));
        v16f32 a00 = bias, a01 = bias, a10 = bias, a11 = bias;      // [ping-pong][column block]
        const uint4* tlane = reinterpret_cast<const uint4*>(s_t + (col >> 2) * kGSeg + (col & 3u) * 16 + half * 64);
        uint32_t t0 = 0;
#define HM_LAST(x0, x1)                                                                                          \
        if (wave_on) {                                                                                           \
            const v16i32 n0 = __builtin_bit_cast(v16i32, x0), n1 = __builtin_bit_cast(v16i32, x1);               \
            if (t0 + 32u <= nt) {                                                                                \
                knn_keys_tagged<false, KNN>(n0, t0, half, nt, k0);                                               \
                knn_keys_tagged<false, KNN>(n1, t0, half, nt, k1);                                               \
            } else {                                                                                             \
                knn_keys_tagged<true, KNN>(n0, t0, half, nt, k0);                                                \
                knn_keys_tagged<true, KNN>(n1, t0, half, nt, k1);                                                \
            }                                                                                                    \
        }
        // The stage boundary, run by every wave between steps 3 and 4 of a stage's last tile: this wave's reads of the stage
        // have returned; the next stage has landed (its requests are the oldest; those of the kGStages - 2 stages behind it
        // may stay in flight) — for every wave of the block once the barrier is passed, and by then everybody is done reading
        // the current stage, whose slot the next request takes.
        uint32_t st = 0;
        auto boundary = [&]() {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((kGStages - 2) * kGTps * 2) : "memory");
            asm volatile("s_barrier" ::: "memory");
            issue(st + (uint32_t)kGStages);
        };
        auto nothing = [&]() {};
        // stage 0 has landed; the last slot of the ring takes its first request
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kGStages - 2) * kGTps * 2) : "memory");
        asm volatile("s_barrier" ::: "memory");
        issue((uint32_t)kGStages - 1u);
        uint4 f[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = tlane[(128 / 16) * i];
        for (;; ++st) {
            const uint4* ta = tlane + (size_t)((st % kGStages) * (uint32_t)(kGTps * kGTile)) / 16;
            const uint4* tb = ta + kGTile / 16;
            const uint4* tn = tlane + (size_t)(((st + 1u) % kGStages) * (uint32_t)(kGTps * kGTile)) / 16;
            const bool last_a = t0 + 32u >= nt, last_b = t0 + 64u >= nt;
            // (idle waves skip the arithmetic, not the boundary)
            if (wave_on) {
                if (st == 0) knn_chain4w<KNN, false>(ta, last_a ? ta : tb, f, qb, bias, a00, a01, a10, a11, 0u, half, k0, k1, nothing);
                else knn_chain4w<KNN, true>(ta, last_a ? ta : tb, f, qb, bias, a00, a01, a10, a11, t0 - 32u, half, k0, k1, nothing);
            }
            if (last_a) {
                HM_LAST(a00, a01)
                break;
            }
            t0 += 32u;
            if (last_b) {
                if (wave_on) knn_chain4w<KNN, true>(tb, tb, f, qb, bias, a10, a11, a00, a01, t0 - 32u, half, k0, k1, nothing);
                HM_LAST(a10, a11)
                break;
            }
            if (wave_on) knn_chain4w<KNN, true>(tb, tn, f, qb, bias, a10, a11, a00, a01, t0 - 32u, half, k0, k1, boundary);
            else boundary();
            t0 += 32u;
        }
#undef HM_LAST
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the requests past the end
    }
    if (!wave_on) return;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        int (&k)[KNN] = cb ? k1 : k0;
        int o[KNN];
#pragma unroll
        for (int i = 0; i < KNN; ++i) o[i] = __shfl_xor(k[i], 32);
#pragma unroll
        for (int i = 0; i < KNN; ++i) topk_insert<KNN>(k, o[i]);
        const uint32_t qi = q0 + 32u * cb + col;
        if (half == 0 && qi < nq) {
#pragma unroll
            for (int i = 0; i < KNN; ++i) {
                const uint32_t m = k[i] == 0x7FFFFFFF ? 0xFFFFFFFFu : (uint32_t)k[i] + (512u << 21);
                akz_neighbor nb = {m & ((1u << kIdxBits) - 1u), m >> kIdxBits};
                P.out[(size_t)qi * KNN + i] = nb;
            }
        }
    }
}

struct HmPairProb {
    const akz_neighbor* fwd;  // [na][2]  a -> b
    const akz_neighbor* rev;  // [nb][2]  b -> a (symmetric only)
    const uint32_t* na;
    const uint32_t* nb;
    uint32_t a_cap, b_cap;
    uint32_t* pairs;          // [cap][2]
    uint32_t cap;
    uint32_t* n_out;
};

__device__ __forceinline__ bool accept(int rule, uint32_t d0, uint32_t d1, uint32_t pu, float pf)
{
    if (rule == HM_RULE_BETTER_BY_STRICT) return d0 + pu < d1;   // ch5 main.rs:162
    if (rule == HM_RULE_BETTER_BY) return d0 + pu <= d1;         // cv-sfm/src/lib.rs:3107
    return (float)d0 < (float)d1 * pf;                           // akaze/tests/estimate_pose.rs:92
}

// matching() + symmetric_matching(): accept rule, reverse check, ordered (ascending a) compaction.
__global__ __launch_bounds__(1024) void k_pairs(const HmPairProb* __restrict__ probs, int rule, uint32_t pu, float pf,
                                                int symmetric)
{
    __shared__ uint32_t s_wave[16];
    const HmPairProb P = probs[blockIdx.x];
    uint32_t na = min(*P.na, P.a_cap), nb = min(*P.nb, P.b_cap);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t base = 0;
    if (na < 2 || nb < 2) na = 0;  // cv-sfm/src/lib.rs:3099-3101 (the other call sites would panic)
    for (uint32_t a0 = 0; a0 < na; a0 += 1024) {
        uint32_t a = a0 + threadIdx.x;
        bool keep = false;
        uint32_t bidx = 0;
        if (a < na) {
            akz_neighbor n0 = P.fwd[(size_t)a * 2], n1 = P.fwd[(size_t)a * 2 + 1];
            if (accept(rule, n0.distance, n1.distance, pu, pf)) {
                bidx = n0.index;
                keep = true;
                if (symmetric) {
                    akz_neighbor r0 = P.rev[(size_t)bidx * 2], r1 = P.rev[(size_t)bidx * 2 + 1];
                    keep = accept(rule, r0.distance, r1.distance, pu, pf) && r0.index == a;
                }
            }
        }
        unsigned long long bal = __ballot(keep);
        if (lane == 0) s_wave[wv] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t woff = 0, tot = 0;
        for (int q = 0; q < 16; ++q) {
            if (q < wv) woff += s_wave[q];
            tot += s_wave[q];
        }
        if (keep) {
            uint32_t o = base + woff + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            if (o < P.cap) {
                P.pairs[(size_t)o * 2] = a;
                P.pairs[(size_t)o * 2 + 1] = bidx;
            }
        }
        base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *P.n_out = base;
}

}  // namespace

struct hm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t max_q = 0, max_t = 0;
    // staging for the host-buffer API
    uint4* d_a = nullptr;
    uint4* d_b = nullptr;
    uint32_t* d_na = nullptr;  // [2]: na, nb
    uint32_t resident_nt = 0;  // hm_set_targets: d_b holds this many target descriptors (0: nothing resident)
    bool resident = false;
    uint64_t generation = 0;   // of the resident set: hm_set_targets hands out a new one per upload (hm_targets_generation)
    akz_neighbor* d_fwd = nullptr;
    akz_neighbor* d_rev = nullptr;
    uint32_t* d_pairs = nullptr;
    uint32_t* d_npairs = nullptr;
    // problem descriptor ring (device) for batched calls
    // (a ring of kStageRing slots: a call waits only for the copies of the call that used its slot four calls
    // ago, so the host can run several micro-batches ahead of the GPU; with a single slot every call waited
    // for the previous call's copies, i.e. for the previous micro-batch's whole extraction)
    struct StageSlot {
        void* d = nullptr;
        void* h = nullptr;
        size_t bytes = 0;
        hipEvent_t ev = nullptr;
        bool pending = false;
    };
    static constexpr int kStageRing = 4;
    StageSlot ring[kStageRing];
    uint64_t ring_pos = 0;
    StageSlot* slot = nullptr;     // the current call's slot
    void* d_probs = nullptr;       // = slot->d
    void* h_probs = nullptr;       // = slot->h, pinned staging twin
    // scratch knn results for batched device calls
    // MFMA path: +-1 recoded descriptor blocks
    uint32_t* d_exp = nullptr;
    size_t exp_words = 0;
    bool use_mfma = true;
    bool use_fp4 = true;           // E2M1 recoding + v_mfma_scale_f32_32x32x64_f8f6f4 (AKZ_MATCH_FP4=0: int8 MFMA)
    bool use_glds = true;          // the wide FP4 kernel, tiles by LDS-DMA (HM_OPT_NO_LDS_DMA: k_knn_mfma4, register-staged)
    // optional timing of the k-NN launches (HIP events on the matcher stream), for bench.py's MFMA roofline
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> t_pending;
    std::vector<hipEvent_t> t_pool;
    double t_ms = 0.0;
    uint64_t t_launches = 0, t_macs = 0;
    akz_neighbor* d_bfwd = nullptr;
    akz_neighbor* d_brev = nullptr;
    size_t bscratch_elems = 0;
    hipEvent_t ev = nullptr;
    // scratch of the host-buffer place-recognition calls (hm_hash_bag, hm_hash_knn)
    void* d_lsh = nullptr;
    size_t lsh_bytes = 0;
};

// Problem descriptors are built in pinned host memory and copied stream-ordered; the staging
// buffer is reused only after the previous copy has completed.
static int32_t hm_ensure_probs(hm_ctx* c, size_t bytes)
{
    hm_ctx::StageSlot& S = c->ring[c->ring_pos++ % hm_ctx::kStageRing];
    c->slot = &S;
    if (S.pending) {
        AKZ_HIP(hipEventSynchronize(S.ev));
        S.pending = false;
    }
    if (!S.ev) AKZ_HIP(hipEventCreateWithFlags(&S.ev, hipEventDisableTiming));
    if (bytes > S.bytes) {
        AKZ_HIP(hipStreamSynchronize(c->stream));   // kernels still reading the old device copy
        if (S.d) AKZ_HIP(hipFree(S.d));
        if (S.h) AKZ_HIP(hipHostFree(S.h));
        S.d = S.h = nullptr;
        S.bytes = 0;
        size_t nb = akz_align_up(bytes * 2, 4096);
        AKZ_HIP(hipMalloc(&S.d, nb));
        AKZ_HIP(hipHostMalloc(&S.h, nb, hipHostMallocDefault));
        S.bytes = nb;
    }
    c->d_probs = S.d;
    c->h_probs = S.h;
    return AKZ_OK;
}
static int32_t hm_push_probs(hm_ctx* c, size_t off, const void* src, size_t bytes)
{
    memcpy((char*)c->h_probs + off, src, bytes);
    AKZ_HIP(hipMemcpyAsync((char*)c->d_probs + off, (char*)c->h_probs + off, bytes, hipMemcpyHostToDevice, c->stream));
    AKZ_HIP(hipEventRecord(c->slot->ev, c->stream));
    c->slot->pending = true;
    return AKZ_OK;
}

extern "C" int32_t hm_create_ex(int32_t device, uint32_t max_queries, uint32_t max_targets, uint32_t flags, hm_ctx** out)
{
    return akz_guard([&]() -> int32_t {
        // the MFMA kernel carries (2 * hamming) << 21 | row: rows need 21 bits
        if (!out || max_queries == 0 || max_targets == 0 || max_targets >= (1u << (kIdxBits - 1))) return AKZ_E_INVALID;
        if (flags & ~(HM_OPT_NO_FP4 | HM_OPT_NO_MFMA | HM_OPT_STREAM_PRIORITY | HM_OPT_NO_LDS_DMA | HM_OPT_CU_MASK)) return AKZ_E_INVALID;   // unknown switches
        const int cus = (int)((flags & HM_OPT_CU_MASK) >> HM_OPT_CU_SHIFT);
        if (cus > 32) return AKZ_E_INVALID;            // (before anything is allocated)
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return AKZ_E_NO_DEVICE;
        AKZ_HIP(hipSetDevice(device));
        hm_ctx* c = new hm_ctx();
        c->device = device;
        c->max_q = max_queries;
        c->max_t = max_targets;
        uint32_t m = max_queries > max_targets ? max_queries : max_targets;
        c->max_q = c->max_t = m;  // symmetric matching swaps the roles
        {
            int prio_lo = 0, prio_hi = 0;
            hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
            // HM_OPT_STREAM_PRIORITY: the matcher as least-urgent filler work that yields to the scale-space stream
            if (cus) AKZ_HIP(akz_stream_on_cus(&c->stream, 32 - cus, cus));
            else AKZ_HIP(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, (flags & HM_OPT_STREAM_PRIORITY) ? prio_lo : 0));
        }
        AKZ_HIP(hipEventCreateWithFlags(&c->ev, hipEventDisableTiming));
        c->use_mfma = !(flags & HM_OPT_NO_MFMA);
        c->use_fp4 = !(flags & HM_OPT_NO_FP4);
        c->use_glds = !(flags & HM_OPT_NO_LDS_DMA);
        AKZ_HIP(hipMalloc(&c->d_a, (size_t)m * 64));
        AKZ_HIP(hipMalloc(&c->d_b, (size_t)m * 64));
        AKZ_HIP(hipMalloc(&c->d_na, sizeof(uint32_t) * 4));
        AKZ_HIP(hipMalloc(&c->d_fwd, sizeof(akz_neighbor) * 2 * (size_t)m));
        AKZ_HIP(hipMalloc(&c->d_rev, sizeof(akz_neighbor) * 2 * (size_t)m));
        AKZ_HIP(hipMalloc(&c->d_pairs, sizeof(uint32_t) * 2 * (size_t)m));
        AKZ_HIP(hipMalloc(&c->d_npairs, sizeof(uint32_t) * 4));
        *out = c;
        return AKZ_OK;
    });
}

extern "C" int32_t hm_create(int32_t device, uint32_t max_queries, uint32_t max_targets, hm_ctx** out)
{
    return hm_create_ex(device, max_queries, max_targets, 0u, out);
}

extern "C" int32_t hm_destroy(hm_ctx* c)
{
    return akz_guard([&]() -> int32_t {
        if (!c) return AKZ_OK;
        hipSetDevice(c->device);
        if (c->stream) hipStreamSynchronize(c->stream);
        hipFree(c->d_a);
        hipFree(c->d_b);
        hipFree(c->d_na);
        hipFree(c->d_fwd);
        hipFree(c->d_rev);
        hipFree(c->d_pairs);
        hipFree(c->d_npairs);
        for (auto& S : c->ring) {
            if (S.d) hipFree(S.d);
            if (S.h) hipHostFree(S.h);
            if (S.ev) hipEventDestroy(S.ev);
        }
        for (auto& pr : c->t_pending) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
        for (auto e : c->t_pool) hipEventDestroy(e);
        hipFree(c->d_exp);
        hipFree(c->d_bfwd);
        hipFree(c->d_brev);
        hipFree(c->d_lsh);
        if (c->ev) hipEventDestroy(c->ev);
        if (c->stream) hipStreamDestroy(c->stream);
        delete c;
        return AKZ_OK;
    });
}

extern "C" void* hm_stream(hm_ctx* c) { return c ? (void*)c->stream : nullptr; }
extern "C" int32_t hm_sync(hm_ctx* c)
{
    return akz_guard([&]() -> int32_t {
        if (!c) return AKZ_E_INVALID;
        AKZ_HIP(hipStreamSynchronize(c->stream));
        return AKZ_OK;
    });
}

static int32_t launch_knn2(hm_ctx* c, const HmProb* h_probs, uint32_t n_probs, uint32_t max_nq, size_t probs_off,
                           int knn = 2)
{
    if (n_probs == 0) return AKZ_OK;
    if (!c->use_mfma && knn == 2) {   // the VALU kernel exists for k = 2 only
        HmProb* dp = reinterpret_cast<HmProb*>((char*)c->d_probs + probs_off);
        AKZ_TRY(hm_push_probs(c, probs_off, h_probs, sizeof(HmProb) * n_probs));
        dim3 grid((max_nq + kQPB - 1) / kQPB, n_probs);
        hipLaunchKernelGGL(k_knn2, grid, dim3(kBlock), 0, c->stream, dp);
        AKZ_LAUNCH_CHECK();
        return AKZ_OK;
    }
    // unique descriptor blocks referenced by the problems -> one +-1 recoding each
    std::vector<HmExpandJob> jobs;
    std::vector<size_t> job_off;   // word offset of each job's output inside d_exp
    std::vector<HmProbX> px(n_probs);
    size_t words = 0;
    const size_t ew = c->use_fp4 ? 64 : 128;   // words of recoded descriptor
    std::unordered_map<const void*, size_t> seen;   // block -> its job (a window call has thousands of problems over few blocks)
    auto expanded = [&](const uint4* src, const uint32_t* cnt, uint32_t cap) -> size_t {
        auto it = seen.find((const void*)src);
        if (it != seen.end()) return job_off[it->second];
        seen.emplace((const void*)src, jobs.size());
        jobs.push_back(HmExpandJob{(const uint32_t*)src, cnt, cap, nullptr});
        job_off.push_back(words);
        words += (size_t)cap * ew;
        return job_off.back();
    };
    std::vector<size_t> qoff(n_probs), toff(n_probs);
    uint32_t max_cap = 0;
    for (uint32_t i = 0; i < n_probs; ++i) {
        qoff[i] = expanded(h_probs[i].q, h_probs[i].nq, h_probs[i].q_cap);
        toff[i] = expanded(h_probs[i].t, h_probs[i].nt, h_probs[i].t_cap);
        max_cap = std::max(max_cap, std::max(h_probs[i].q_cap, h_probs[i].t_cap));
    }
    if (words > c->exp_words) {
        AKZ_HIP(hipStreamSynchronize(c->stream));
        if (c->d_exp) AKZ_HIP(hipFree(c->d_exp));
        c->d_exp = nullptr;
        AKZ_HIP(hipMalloc(&c->d_exp, sizeof(uint32_t) * words));
        c->exp_words = words;
    }
    for (size_t i = 0; i < jobs.size(); ++i) jobs[i].dst = c->d_exp + job_off[i];
    for (uint32_t i = 0; i < n_probs; ++i)
        px[i] = HmProbX{c->d_exp + qoff[i], h_probs[i].nq, h_probs[i].q_cap, c->d_exp + toff[i], h_probs[i].nt,
                        h_probs[i].t_cap, h_probs[i].out};
    // descriptors for this call: [HmProbX x n_probs][HmExpandJob x jobs] inside the knn region of the staging buffer
    const size_t jobs_off = probs_off + akz_align_up(sizeof(HmProbX) * n_probs, 64);
    AKZ_TRY(hm_push_probs(c, probs_off, px.data(), sizeof(HmProbX) * n_probs));
    AKZ_TRY(hm_push_probs(c, jobs_off, jobs.data(), sizeof(HmExpandJob) * jobs.size()));
    if (c->use_fp4)
        hipLaunchKernelGGL(k_expand4, dim3((max_cap * 16 + 255) / 256, (uint32_t)jobs.size()), dim3(256), 0, c->stream,
                           reinterpret_cast<const HmExpandJob*>((char*)c->d_probs + jobs_off));
    else
        hipLaunchKernelGGL(k_expand, dim3((max_cap * 16 + 255) / 256, (uint32_t)jobs.size()), dim3(256), 0, c->stream,
                           reinterpret_cast<const HmExpandJob*>((char*)c->d_probs + jobs_off));
    AKZ_LAUNCH_CHECK();
    // timing: the launch carries its own start / stop events (hipExtLaunchKernel: the dispatch's begin and end timestamps,
    // i.e. the duration rocprofv3's kernel trace reports) — an event bracket on the stream would also count the time the
    // launch waits for the other streams' kernels
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (c->timing) {
        auto take = [&]() {
            hipEvent_t e = nullptr;
            if (!c->t_pool.empty()) { e = c->t_pool.back(); c->t_pool.pop_back(); }
            else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
            return e;
        };
        ev0 = take();
        ev1 = take();
        if (!ev0 || !ev1) ev0 = ev1 = nullptr;
    }
    {
        dim3 grid((max_nq + 255) / 256, n_probs);
        const HmProbX* dp = reinterpret_cast<const HmProbX*>((char*)c->d_probs + probs_off);
        if (c->use_fp4 && c->use_glds) {
            switch (knn) {
            case 1: hipExtLaunchKernelGGL(k_knn_mfma4w<1>, grid, dim3(kWideBlock), 0, c->stream, ev0, ev1, 0, dp); break;
            case 3: hipExtLaunchKernelGGL(k_knn_mfma4w<3>, grid, dim3(kWideBlock), 0, c->stream, ev0, ev1, 0, dp); break;
            default: hipExtLaunchKernelGGL(k_knn_mfma4w<2>, grid, dim3(kWideBlock), 0, c->stream, ev0, ev1, 0, dp); break;
            }
        } else if (c->use_fp4) {
            switch (knn) {
            case 1: hipExtLaunchKernelGGL(k_knn_mfma4<1>, grid, dim3(kMfmaBlock), 0, c->stream, ev0, ev1, 0, dp); break;
            case 3: hipExtLaunchKernelGGL(k_knn_mfma4<3>, grid, dim3(kMfmaBlock), 0, c->stream, ev0, ev1, 0, dp); break;
            default: hipExtLaunchKernelGGL(k_knn_mfma4<2>, grid, dim3(kMfmaBlock), 0, c->stream, ev0, ev1, 0, dp); break;
            }
        } else {
            switch (knn) {
            case 1: hipExtLaunchKernelGGL(k_knn_mfma<1>, grid, dim3(kMfmaBlock), 0, c->stream, ev0, ev1, 0, dp); break;
            case 3: hipExtLaunchKernelGGL(k_knn_mfma<3>, grid, dim3(kMfmaBlock), 0, c->stream, ev0, ev1, 0, dp); break;
            default: hipExtLaunchKernelGGL(k_knn_mfma<2>, grid, dim3(kMfmaBlock), 0, c->stream, ev0, ev1, 0, dp); break;
            }
        }
    }
    if (ev0 && ev1) {
        c->t_pending.emplace_back(ev0, ev1);
        c->t_launches += 1;
    }
    AKZ_LAUNCH_CHECK();
    return AKZ_OK;
}

// bytes of staging the knn stage of a call needs
static size_t knn_stage_bytes(uint32_t n_probs)
{
    return akz_align_up(sizeof(HmProbX) * n_probs, 64) + akz_align_up(sizeof(HmExpandJob) * 2 * n_probs, 64) + 256;
}

extern "C" int32_t hm_knn2(hm_ctx* c, const akz_descriptor* q, uint32_t nq, const akz_descriptor* t, uint32_t nt,
                           akz_neighbor* out)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !q || !t || !out) return AKZ_E_INVALID;
        if (nt < 2) return AKZ_E_INVALID;  // the reference asserts two neighbours (estimate_pose.rs:89)
        if (nq > c->max_q || nt > c->max_t) return AKZ_E_TOO_LARGE;
        if (nq == 0) return AKZ_OK;
        AKZ_HIP(hipSetDevice(c->device));
        AKZ_TRY(hm_ensure_probs(c, knn_stage_bytes(1)));
        uint32_t cnt[2] = {nq, nt};
        AKZ_HIP(hipMemcpyAsync(c->d_na, cnt, sizeof(cnt), hipMemcpyHostToDevice, c->stream));
        AKZ_HIP(hipMemcpyAsync(c->d_a, q, (size_t)nq * 64, hipMemcpyHostToDevice, c->stream));
        c->resident = false;   // (the staging buffer of the targets is overwritten)
        AKZ_HIP(hipMemcpyAsync(c->d_b, t, (size_t)nt * 64, hipMemcpyHostToDevice, c->stream));
        HmProb p = {c->d_a, c->d_na, nq, c->d_b, c->d_na + 1, nt, c->d_fwd};
        AKZ_TRY(launch_knn2(c, &p, 1, nq, 0));
        AKZ_HIP(hipMemcpyAsync(out, c->d_fwd, sizeof(akz_neighbor) * 2 * (size_t)nq, hipMemcpyDeviceToHost, c->stream));
        AKZ_HIP(hipStreamSynchronize(c->stream));
        return AKZ_OK;
    });
}

// LinearKnn{metric: Hamming, iter: t}.knn(q, k) for k = 1, 2, 3 (cv-sfm/src/lib.rs:1474 uses 3).
extern "C" int32_t hm_knn(hm_ctx* c, const akz_descriptor* q, uint32_t nq, const akz_descriptor* t, uint32_t nt,
                          uint32_t k, akz_neighbor* out)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !q || !t || !out || k < 1 || k > 3) return AKZ_E_INVALID;
        if (nq > c->max_q || nt > c->max_t) return AKZ_E_TOO_LARGE;
        if (nq == 0) return AKZ_OK;
        AKZ_HIP(hipSetDevice(c->device));
        AKZ_TRY(hm_ensure_probs(c, knn_stage_bytes(1)));
        size_t need = (size_t)nq * 3;
        if (need > c->bscratch_elems) {
            AKZ_HIP(hipStreamSynchronize(c->stream));
            if (c->d_bfwd) AKZ_HIP(hipFree(c->d_bfwd));
            if (c->d_brev) AKZ_HIP(hipFree(c->d_brev));
            c->d_bfwd = c->d_brev = nullptr;
            AKZ_HIP(hipMalloc(&c->d_bfwd, sizeof(akz_neighbor) * need));
            AKZ_HIP(hipMalloc(&c->d_brev, sizeof(akz_neighbor) * need));
            c->bscratch_elems = need;
        }
        uint32_t cnt[2] = {nq, nt};
        AKZ_HIP(hipMemcpyAsync(c->d_na, cnt, sizeof(cnt), hipMemcpyHostToDevice, c->stream));
        AKZ_HIP(hipMemcpyAsync(c->d_a, q, (size_t)nq * 64, hipMemcpyHostToDevice, c->stream));
        c->resident = false;   // (the staging buffer of the targets is overwritten)
        if (nt) AKZ_HIP(hipMemcpyAsync(c->d_b, t, (size_t)nt * 64, hipMemcpyHostToDevice, c->stream));
        HmProb p = {c->d_a, c->d_na, nq, c->d_b, c->d_na + 1, nt ? nt : 1u, c->d_bfwd};
        AKZ_TRY(launch_knn2(c, &p, 1, nq, 0, (int)k));
        AKZ_HIP(hipMemcpyAsync(out, c->d_bfwd, sizeof(akz_neighbor) * k * (size_t)nq, hipMemcpyDeviceToHost, c->stream));
        AKZ_HIP(hipStreamSynchronize(c->stream));
        return AKZ_OK;
    });
}

// space::LinearKnn keeps its target set (`iter`) and is asked once per query descriptor (akaze/tests/estimate_pose.rs:82-88:
// `.map(|d1| knn.knn(d1, 2))`).  A literal port of that loop through hm_knn uploads the whole target set for every query;
// hm_set_targets uploads it ONCE and hm_knn_targets answers any number of queries (one, or a batch) against the resident
// copy.  Any other host-buffer call on the context (hm_knn, hm_knn2, hm_match, hm_hash_bag) reuses the staging buffer and
// ends the residency: hm_knn_targets then answers AKZ_E_INVALID instead of searching stale data.
extern "C" int32_t hm_set_targets(hm_ctx* c, const akz_descriptor* t, uint32_t nt)
{
    return akz_guard([&]() -> int32_t {
        if (!c || (nt && !t)) return AKZ_E_INVALID;
        if (nt > c->max_t) return AKZ_E_TOO_LARGE;
        AKZ_HIP(hipSetDevice(c->device));
        c->resident = false;
        if (nt) AKZ_HIP(hipMemcpyAsync(c->d_b, t, (size_t)nt * 64, hipMemcpyHostToDevice, c->stream));
        AKZ_HIP(hipStreamSynchronize(c->stream));   // the caller's array may go away
        c->resident_nt = nt;
        c->resident = true;
        // process-wide, never reused: a context destroyed and re-created at the same address (a binding that grows its
        // matcher) must not hand out a number an older upload already carries
        static std::atomic<uint64_t> g_generation{0};
        c->generation = ++g_generation;
        return AKZ_OK;
    });
}
// Which upload the context holds: the number of the hm_set_targets call whose targets are resident, 0 when none is (never
// uploaded, or another host-buffer call took the staging buffer since).  A binding remembers the value its own upload got
// and asks again before every hm_knn_targets: two target sets alternating on one context, a set dropped and another
// allocated at the same address — every case a pointer comparison misses — shows up as a different number.
extern "C" uint64_t hm_targets_generation(hm_ctx* c)
{
    return (c && c->resident) ? c->generation : 0;
}
extern "C" int32_t hm_knn_targets(hm_ctx* c, const akz_descriptor* q, uint32_t nq, uint32_t k, akz_neighbor* out)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !q || !out || k < 1 || k > 3 || !c->resident) return AKZ_E_INVALID;
        if (nq > c->max_q) return AKZ_E_TOO_LARGE;
        if (nq == 0) return AKZ_OK;
        AKZ_HIP(hipSetDevice(c->device));
        AKZ_TRY(hm_ensure_probs(c, knn_stage_bytes(1)));
        const size_t need = (size_t)nq * 3;
        if (need > c->bscratch_elems) {
            AKZ_HIP(hipStreamSynchronize(c->stream));
            if (c->d_bfwd) AKZ_HIP(hipFree(c->d_bfwd));
            if (c->d_brev) AKZ_HIP(hipFree(c->d_brev));
            c->d_bfwd = c->d_brev = nullptr;
            AKZ_HIP(hipMalloc(&c->d_bfwd, sizeof(akz_neighbor) * need));
            AKZ_HIP(hipMalloc(&c->d_brev, sizeof(akz_neighbor) * need));
            c->bscratch_elems = need;
        }
        const uint32_t nt = c->resident_nt;
        uint32_t cnt[2] = {nq, nt};
        AKZ_HIP(hipMemcpyAsync(c->d_na, cnt, sizeof(cnt), hipMemcpyHostToDevice, c->stream));
        AKZ_HIP(hipMemcpyAsync(c->d_a, q, (size_t)nq * 64, hipMemcpyHostToDevice, c->stream));
        HmProb p = {c->d_a, c->d_na, nq, c->d_b, c->d_na + 1, nt ? nt : 1u, c->d_bfwd};
        AKZ_TRY(launch_knn2(c, &p, 1, nq, 0, (int)k));
        AKZ_HIP(hipMemcpyAsync(out, c->d_bfwd, sizeof(akz_neighbor) * k * (size_t)nq, hipMemcpyDeviceToHost, c->stream));
        AKZ_HIP(hipStreamSynchronize(c->stream));
        return AKZ_OK;
    });
}

// One query frame against n_views stored views (cv-sfm/src/lib.rs:1468-1486: every feature of the new frame is
// matched with knn(., 3) against each of the recent views).  Everything device-resident: d_q [cap][64] with
// its count d_nq, d_views [>= max(view_idx)+1][cap][64] with counts d_nviews; view v of this call is block
// view_idx[v].  d_out [n_views][cap][k].  Stream-ordered after `stream_to_wait`; results are ready on hm_stream().
extern "C" int32_t hm_knn_views_device(hm_ctx* c, const void* d_q, const void* d_nq, const void* d_views,
                                       const void* d_nviews, uint32_t cap_per_img, const uint32_t* view_idx,
                                       uint32_t n_views, uint32_t k, void* d_out, void* stream_to_wait)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !d_q || !d_nq || !d_views || !d_nviews || !view_idx || !d_out || k < 1 || k > 3) return AKZ_E_INVALID;
        if (cap_per_img == 0 || cap_per_img >= (1u << (kIdxBits - 1))) return AKZ_E_INVALID;
        if (n_views == 0) return AKZ_OK;
        AKZ_HIP(hipSetDevice(c->device));
        if (stream_to_wait) {
            AKZ_HIP(hipEventRecord(c->ev, akz_wait_stream(stream_to_wait)));
            AKZ_HIP(hipStreamWaitEvent(c->stream, c->ev, 0));
        }
        AKZ_TRY(hm_ensure_probs(c, knn_stage_bytes(n_views)));
        std::vector<HmProb> hp(n_views);
        for (uint32_t v = 0; v < n_views; ++v)
            hp[v] = HmProb{(const uint4*)d_q, (const uint32_t*)d_nq, cap_per_img,
                           (const uint4*)d_views + (size_t)view_idx[v] * cap_per_img * 4,
                           (const uint32_t*)d_nviews + view_idx[v], cap_per_img,
                           (akz_neighbor*)d_out + (size_t)v * cap_per_img * k};
        return launch_knn2(c, hp.data(), n_views, cap_per_img, 0, (int)k);
    });
}

// The batched form of the same search: problem p = every feature of query block iq[p] (of d_q) against target block
// it[p] (of d_t), k neighbours each — a whole step's window of recent views (cv-sfm/src/lib.rs:1462-1486 for every
// frame of a micro-batch: n_frames x K problems) in ONE call; every distinct block is recoded once.  d_out [n_probs][cap][k].
extern "C" int32_t hm_knn_batch_device(hm_ctx* c, const void* d_q, const void* d_nq, const void* d_t, const void* d_nt,
                                       uint32_t cap_per_img, const uint32_t* iq, const uint32_t* it, uint32_t n_probs,
                                       uint32_t k, void* d_out, void* stream_to_wait)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !d_q || !d_nq || !d_t || !d_nt || !iq || !it || !d_out || k < 1 || k > 3) return AKZ_E_INVALID;
        if (cap_per_img == 0 || cap_per_img >= (1u << (kIdxBits - 1)) || n_probs > 65535u) return AKZ_E_INVALID;
        if (n_probs == 0) return AKZ_OK;
        AKZ_HIP(hipSetDevice(c->device));
        if (stream_to_wait) {
            AKZ_HIP(hipEventRecord(c->ev, akz_wait_stream(stream_to_wait)));
            AKZ_HIP(hipStreamWaitEvent(c->stream, c->ev, 0));
        }
        AKZ_TRY(hm_ensure_probs(c, knn_stage_bytes(n_probs)));
        std::vector<HmProb> hp(n_probs);
        for (uint32_t p = 0; p < n_probs; ++p)
            hp[p] = HmProb{(const uint4*)d_q + (size_t)iq[p] * cap_per_img * 4, (const uint32_t*)d_nq + iq[p], cap_per_img,
                           (const uint4*)d_t + (size_t)it[p] * cap_per_img * 4, (const uint32_t*)d_nt + it[p], cap_per_img,
                           (akz_neighbor*)d_out + (size_t)p * cap_per_img * k};
        return launch_knn2(c, hp.data(), n_probs, cap_per_img, 0, (int)k);
    });
}

// Best-of-views landmark selection of the registration path (cv-sfm/src/lib.rs:1489-1542): every feature of the new
// frame got k neighbours in each of n_views views (hm_knn_views_device); a neighbour is an observation of a
// landmark (landmarks[view block][feature index]).  Per feature: keep the best (smallest) distance of every
// distinct landmark, take the three best landmarks, and classify them with the reference's two rules:
//   best[0].d + better_by <= best[1].d                   -> 1: unique match to best[0]
//   else best[1].d + better_by <= best[2].d              -> 2: best[0] and best[1] are merge candidates (the caller
//                                                              still checks are_landmarks_sharing_view, :1528)
//   else                                                 -> 0
// The reference collects the landmarks in a HashMap, so its order among equal distances is unspecified; here ties
// go to the lower landmark key (parity unpinned for ties).  One thread per feature: n_views * k <= 96 entries.
__global__ __launch_bounds__(256) void k_best_of_views(const akz_neighbor* __restrict__ knn, const uint32_t* __restrict__ nq,
                                                       uint32_t cap, uint32_t n_views, uint32_t k,
                                                       const uint32_t* __restrict__ landmarks, const uint32_t* __restrict__ view_idx,
                                                       const uint32_t* __restrict__ nviews, uint32_t better_by,
                                                       uint2* __restrict__ best, uint32_t* __restrict__ decision,
                                                       const uint32_t* __restrict__ q_block)
{
    // blockIdx.y = frame of a batched call: its slab of the k-NN output, its row of view indices, its count
    const uint32_t f = blockIdx.y;
    knn += (size_t)f * n_views * cap * k;
    view_idx += (size_t)f * n_views;
    best += (size_t)f * cap * 3;
    decision += (size_t)f * cap;
    if (q_block) nq += q_block[f];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= min(*nq, cap)) return;
    const uint32_t NONE = 0xFFFFFFFFu;
    uint32_t bl[3] = {NONE, NONE, NONE}, bd[3] = {NONE, NONE, NONE};   // ascending (distance, landmark)
    for (uint32_t v = 0; v < n_views; ++v) {
        const uint32_t blk = view_idx[v], nt = min(nviews[blk], cap);
        for (uint32_t j = 0; j < k && j < nt; ++j) {
            const akz_neighbor nb = knn[((size_t)v * cap + i) * k + j];
            const uint32_t l = landmarks[(size_t)blk * cap + nb.index], d = nb.distance;
            // already among the best three: keep the smaller distance
            int at = -1;
            for (int q = 0; q < 3; ++q)
                if (bl[q] == l && bd[q] != NONE) at = q;
            if (at >= 0) {
                if (d >= bd[at]) continue;
                for (int q = at; q < 2; ++q) { bl[q] = bl[q + 1]; bd[q] = bd[q + 1]; }   // take it out, re-insert below
                bl[2] = NONE; bd[2] = NONE;
            }
            int pos = 3;
            for (int q = 2; q >= 0; --q)
                if (d < bd[q] || (d == bd[q] && l < bl[q])) pos = q;
            if (pos == 3) continue;
            for (int q = 2; q > pos; --q) { bl[q] = bl[q - 1]; bd[q] = bd[q - 1]; }
            bl[pos] = l; bd[pos] = d;
        }
    }
    for (int q = 0; q < 3; ++q) best[(size_t)i * 3 + q] = make_uint2(bl[q], bd[q]);
    uint32_t dec = 0;
    // the reference unwraps three landmarks (:1509-1513): with fewer than three there is no decision
    if (bd[2] != NONE) {
        if (bd[0] + better_by <= bd[1]) dec = 1;
        else if (bd[1] + better_by <= bd[2]) dec = 2;
    }
    decision[i] = dec;
}

extern "C" int32_t hm_best_of_views_device(hm_ctx* c, const void* d_knn, const void* d_nq, uint32_t cap_per_img,
                                           const uint32_t* view_idx, uint32_t n_views, uint32_t k, const void* d_landmarks,
                                           const void* d_nviews, uint32_t better_by, void* d_best, void* d_decision,
                                           void* stream_to_wait)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !d_knn || !d_nq || !view_idx || !d_landmarks || !d_nviews || !d_best || !d_decision) return AKZ_E_INVALID;
        if (k < 1 || k > 3 || n_views == 0 || n_views > 64 || cap_per_img == 0) return AKZ_E_INVALID;
        AKZ_HIP(hipSetDevice(c->device));
        if (stream_to_wait) {
            AKZ_HIP(hipEventRecord(c->ev, akz_wait_stream(stream_to_wait)));
            AKZ_HIP(hipStreamWaitEvent(c->stream, c->ev, 0));
        }
        AKZ_TRY(hm_ensure_probs(c, 256));
        AKZ_TRY(hm_push_probs(c, 0, view_idx, sizeof(uint32_t) * n_views));
        hipLaunchKernelGGL(k_best_of_views, dim3((cap_per_img + 255) / 256), dim3(256), 0, c->stream, (const akz_neighbor*)d_knn,
                           (const uint32_t*)d_nq, cap_per_img, n_views, k, (const uint32_t*)d_landmarks,
                           (const uint32_t*)c->d_probs, (const uint32_t*)d_nviews, better_by, (uint2*)d_best,
                           (uint32_t*)d_decision, (const uint32_t*)nullptr);
        AKZ_LAUNCH_CHECK();
        return AKZ_OK;
    });
}

// The same for every frame of a micro-batch in one launch: frame f's neighbours are slab f of hm_knn_batch_device's output
// laid out [n_frames][n_views][cap][k] (problem f * n_views + v = frame f against its v-th view), its count d_nq[iq[f]], its
// views view_idx[f][0..n_views); d_best [n_frames][cap][3], d_decision [n_frames][cap].
extern "C" int32_t hm_best_of_views_batch_device(hm_ctx* c, const void* d_knn, const void* d_nq, const uint32_t* iq, uint32_t cap_per_img,
                                                 const uint32_t* view_idx, uint32_t n_frames, uint32_t n_views, uint32_t k,
                                                 const void* d_landmarks, const void* d_nviews, uint32_t better_by, void* d_best,
                                                 void* d_decision, void* stream_to_wait)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !d_knn || !d_nq || !iq || !view_idx || !d_landmarks || !d_nviews || !d_best || !d_decision) return AKZ_E_INVALID;
        if (k < 1 || k > 3 || n_views == 0 || n_views > 64 || cap_per_img == 0 || n_frames > 65535u) return AKZ_E_INVALID;
        if (n_frames == 0) return AKZ_OK;
        AKZ_HIP(hipSetDevice(c->device));
        if (stream_to_wait) {
            AKZ_HIP(hipEventRecord(c->ev, akz_wait_stream(stream_to_wait)));
            AKZ_HIP(hipStreamWaitEvent(c->stream, c->ev, 0));
        }
        const size_t vi_bytes = akz_align_up(sizeof(uint32_t) * (size_t)n_frames * n_views, 64);
        AKZ_TRY(hm_ensure_probs(c, vi_bytes + sizeof(uint32_t) * n_frames + 64));
        AKZ_TRY(hm_push_probs(c, 0, view_idx, sizeof(uint32_t) * (size_t)n_frames * n_views));
        AKZ_TRY(hm_push_probs(c, vi_bytes, iq, sizeof(uint32_t) * n_frames));
        hipLaunchKernelGGL(k_best_of_views, dim3((cap_per_img + 255) / 256, n_frames), dim3(256), 0, c->stream,
                           (const akz_neighbor*)d_knn, (const uint32_t*)d_nq, cap_per_img, n_views, k, (const uint32_t*)d_landmarks,
                           (const uint32_t*)c->d_probs, (const uint32_t*)d_nviews, better_by, (uint2*)d_best, (uint32_t*)d_decision,
                           (const uint32_t*)((const char*)c->d_probs + vi_bytes));
        AKZ_LAUNCH_CHECK();
        return AKZ_OK;
    });
}

// ---- from the landmark decisions to the consensus' input (cv-sfm/src/lib.rs:1516-1532, 1549-1563, 1583-1604) ----
// register_frame_subset, between the best-of-views decision and single_view_consensus.model_inliers.  original_matches holds
// ([best0], feature) for a uniquely good best landmark (decision 1, :1516-1520) and ([best0, best1], feature) for a merge
// candidate (decision 2) whose two landmarks share no view (:1521-1531 — are_landmarks_sharing_view is the caller's graph
// test: its verdict arrives as the per-feature mask merge_ok).  landmark_counts then counts EVERY landmark of EVERY
// original match, both landmarks of a merge included (:1549-1552), and a match survives only if all of its landmarks were
// counted once (:1555-1559: "two separate features match to the landmark, that is always 100 % incorrect").  What is left
// becomes FeatureWorldMatch(bearing(feature), triangulate_landmark_robust(landmark)) or, for a merge,
// triangulate_merged_landmark_robust([a, b]) — dropped when the triangulation is None (:1583-1604, filter_map).
// Here: one workgroup per frame.  The landmarks of all original matches go into an open-addressing hash set in LDS (32 768
// slots for at most 2 x 8 192 keys; a key met a second time sets its slot's bit in a duplicate bitmap — set semantics, so
// the outcome does not depend on the order the lanes arrive in); the survivors leave in ascending feature order as
// {feature, row of the world table} — exactly the pair-list form rs_p3p_arrsac_batch_device takes.  The world table is the
// caller's: rows [0, n_world) by landmark key and, when a merge mask is given, rows n_world + f * cap + j = the merged
// triangulation for feature j of frame f; a row with w < 0 (impossible for a Projective point) says "None".  The
// reference's stable sort by observation count (:1561-1574) fixes the order the consensus sees — and ARRSAC's sampling depends
// on data order: given the per-landmark observation counts (hm_landmark_matches_ordered_batch_device) the kernel applies it,
// a stable LSD radix sort in the same workgroup.
constexpr uint32_t kLmSlots = 32768u;
constexpr size_t kLmLdsBytes = sizeof(uint32_t) * (kLmSlots + kLmSlots / 32);
__device__ __forceinline__ uint32_t lm_slot(uint32_t key) { return (key * 2654435761u) >> 17; }
__device__ __forceinline__ void lm_insert(uint32_t* tab, uint32_t* dup, uint32_t key)
{
    uint32_t s = lm_slot(key);
    for (;;) {
        const uint32_t prev = atomicCAS(&tab[s], 0xFFFFFFFFu, key);
        if (prev == 0xFFFFFFFFu) return;
        if (prev == key) { atomicOr(&dup[s >> 5], 1u << (s & 31u)); return; }
        s = (s + 1u) & (kLmSlots - 1u);
    }
}
__device__ __forceinline__ bool lm_claimed_once(const uint32_t* tab, const uint32_t* dup, uint32_t key)
{
    uint32_t s = lm_slot(key);
    while (tab[s] != key) s = (s + 1u) & (kLmSlots - 1u);
    return ((dup[s >> 5] >> (s & 31u)) & 1u) == 0u;
}
__global__ __launch_bounds__(1024) void k_landmark_pairs(const uint2* __restrict__ best, const uint32_t* __restrict__ decision,
                                                         const uint8_t* __restrict__ merge_ok,
                                                         const uint32_t* __restrict__ nq, const uint32_t* __restrict__ iq,
                                                         uint32_t cap, const double* __restrict__ world, uint32_t n_world,
                                                         uint2* __restrict__ pairs, uint32_t* __restrict__ npairs,
                                                         const uint32_t* __restrict__ obs)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* tab = reinterpret_cast<uint32_t*>(smem);         // [kLmSlots] landmark keys (0xFFFFFFFF: empty)
    uint32_t* dup = tab + kLmSlots;                            // [kLmSlots / 32] bit s: slot s' key was inserted more than once
    __shared__ uint32_t s_wave[16], s_base;
    const uint32_t f = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    uint32_t n = nq[iq[f]];
    n = n < cap ? n : cap;
    const uint2* bf = best + (size_t)f * cap * 3;
    const uint32_t* df = decision + (size_t)f * cap;
    const uint8_t* mf = merge_ok ? merge_ok + (size_t)f * cap : nullptr;
    for (uint32_t s = tid; s < kLmSlots; s += 1024) tab[s] = 0xFFFFFFFFu;
    for (uint32_t s = tid; s < kLmSlots / 32; s += 1024) dup[s] = 0u;
    if (tid == 0) s_base = 0;
    __syncthreads();
    // kind of feature j's original match: 0 none, 1 ([best0], j), 2 ([best0, best1], j)
    auto kind_of = [&](uint32_t j, uint32_t& l0, uint32_t& l1) -> uint32_t {
        const uint32_t d = df[j];
        l0 = bf[(size_t)j * 3].x;
        l1 = bf[(size_t)j * 3 + 1].x;
        if (l0 == 0xFFFFFFFFu) return 0u;
        if (d == 1u) return 1u;
        if (d == 2u && mf && mf[j] && l1 != 0xFFFFFFFFu) return 2u;
        return 0u;
    };
    // 1. landmark_counts: every landmark of every original match
    for (uint32_t j = tid; j < n; j += 1024) {
        uint32_t l0, l1;
        const uint32_t kind = kind_of(j, l0, l1);
        if (kind >= 1u) lm_insert(tab, dup, l0);
        if (kind == 2u) lm_insert(tab, dup, l1);
    }
    __syncthreads();
    // 2. the survivors in feature order, with a robust world point
    uint2* out = pairs + (size_t)f * cap;
    for (uint32_t j0 = 0; j0 < n; j0 += 1024) {
        const uint32_t j = j0 + tid;
        bool on = false;
        uint32_t row = 0;
        if (j < n) {
            uint32_t l0, l1;
            const uint32_t kind = kind_of(j, l0, l1);
            if (kind == 1u) {
                on = lm_claimed_once(tab, dup, l0) && l0 < n_world;
                row = l0;
            } else if (kind == 2u) {
                on = lm_claimed_once(tab, dup, l0) && lm_claimed_once(tab, dup, l1);
                row = n_world + f * cap + j;
            }
            if (on) on = world[(size_t)4 * row + 3] >= 0.0;
        }
        const unsigned long long bal = __ballot(on);
        if (lane == 0) s_wave[wv] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t off = s_base;
        for (uint32_t q = 0; q < wv; ++q) off += s_wave[q];
        if (on) out[off + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = make_uint2(j, row);
        __syncthreads();
        if (tid == 0) {
            uint32_t t = 0;
            for (int q = 0; q < 16; ++q) t += s_wave[q];
            s_base += t;
        }
        __syncthreads();
    }
    if (tid == 0) npairs[f] = s_base;
    if (!obs) return;
    // 3. the order the reference's consensus sees (cv-sfm/src/lib.rs:1561-1574): a STABLE sort of the matches by the summed
    // observation count of their landmarks, largest first — equal sums keep the feature order.  (The reference sorts before
    // it drops the matches without a robust triangulation; a stable sort and an order-preserving filter commute.)  The hash
    // set is dead: its LDS holds the keys and the two id buffers of the radix sort.
    static_assert(3 * kRadixSortMax + 16 * 256 <= kLmSlots + kLmSlots / 32, "sort buffers fit in the hash set's LDS");
    __shared__ uint32_t s_tot[256];
    const uint32_t m = s_base;                       // (written before the last barrier of the loop above)
    __syncthreads();                                  // every reader of the hash set is done
    uint32_t* rk = tab;
    uint32_t* ia = rk + kRadixSortMax;
    uint32_t* ib = ia + kRadixSortMax;
    uint32_t* wh = ib + kRadixSortMax;
    for (uint32_t i = tid; i < m; i += 1024) {
        const uint2 p = out[i];
        const uint32_t l0 = bf[(size_t)p.x * 3].x;
        uint32_t sum = l0 < n_world ? obs[l0] : 0u;
        if (p.y >= n_world) {                         // a merged match: both landmarks count
            const uint32_t l1 = bf[(size_t)p.x * 3 + 1].x;
            const uint32_t o1 = l1 < n_world ? obs[l1] : 0u;
            sum = sum + o1 < sum ? 0xFFFFFFFFu : sum + o1;
        }
        rk[i] = ~sum;                                 // ascending keys = descending sums
        ia[i] = i;
    }
    __syncthreads();
    const uint32_t* sorted = lds_radix_sort_ids(rk, ia, ib, wh, s_tot, m, 4);
    uint2 mine[kRadixSortMax / 1024];
#pragma unroll
    for (uint32_t q = 0; q < kRadixSortMax / 1024; ++q) {
        const uint32_t i = tid + 1024 * q;
        mine[q] = i < m ? out[sorted[i]] : make_uint2(0u, 0u);
    }
    __syncthreads();                                  // every entry has been read before any is overwritten
#pragma unroll
    for (uint32_t q = 0; q < kRadixSortMax / 1024; ++q) {
        const uint32_t i = tid + 1024 * q;
        if (i < m) out[i] = mine[q];
    }
}

extern "C" int32_t hm_landmark_matches_ordered_batch_device(hm_ctx* c, const void* d_best, const void* d_decision, const void* d_merge_ok,
                                                            const void* d_obs_counts, const void* d_nq, const uint32_t* iq,
                                                            uint32_t cap_per_img, uint32_t n_frames, const void* d_world,
                                                            uint32_t n_world, void* d_pairs, void* d_npairs, void* stream_to_wait)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !d_best || !d_decision || !d_nq || !iq || !d_world || !d_pairs || !d_npairs) return AKZ_E_INVALID;
        if (cap_per_img == 0 || n_world == 0 || n_frames > 65535u) return AKZ_E_INVALID;
        if (cap_per_img > kLmSlots / 4) return AKZ_E_TOO_LARGE;           // 2 keys per feature at a load factor <= 1/2
        if (d_merge_ok && (uint64_t)n_world + (uint64_t)n_frames * cap_per_img > 0xFFFFFFFFull) return AKZ_E_TOO_LARGE;
        if (n_frames == 0) return AKZ_OK;
        AKZ_HIP(hipSetDevice(c->device));
        if (stream_to_wait) {
            AKZ_HIP(hipEventRecord(c->ev, akz_wait_stream(stream_to_wait)));
            AKZ_HIP(hipStreamWaitEvent(c->stream, c->ev, 0));
        }
        AKZ_TRY(hm_ensure_probs(c, sizeof(uint32_t) * n_frames + 64));
        AKZ_TRY(hm_push_probs(c, 0, iq, sizeof(uint32_t) * n_frames));
        AKZ_HIP(hipFuncSetAttribute((const void*)k_landmark_pairs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLmLdsBytes));
        hipLaunchKernelGGL(k_landmark_pairs, dim3(n_frames), dim3(1024), kLmLdsBytes, c->stream, (const uint2*)d_best,
                           (const uint32_t*)d_decision, (const uint8_t*)d_merge_ok, (const uint32_t*)d_nq, (const uint32_t*)c->d_probs,
                           cap_per_img, (const double*)d_world, n_world, (uint2*)d_pairs, (uint32_t*)d_npairs,
                           (const uint32_t*)d_obs_counts);
        AKZ_LAUNCH_CHECK();
        return AKZ_OK;
    });
}

// the same in ascending feature order (no observation counts: the caller orders, or shuffles, the matches itself)
extern "C" int32_t hm_landmark_matches_batch_device(hm_ctx* c, const void* d_best, const void* d_decision, const void* d_merge_ok,
                                                    const void* d_nq, const uint32_t* iq, uint32_t cap_per_img, uint32_t n_frames,
                                                    const void* d_world, uint32_t n_world, void* d_pairs, void* d_npairs,
                                                    void* stream_to_wait)
{
    return hm_landmark_matches_ordered_batch_device(c, d_best, d_decision, d_merge_ok, nullptr, d_nq, iq, cap_per_img, n_frames, d_world,
                                                    n_world, d_pairs, d_npairs, stream_to_wait);
}

// hm_landmark_matches_batch_device with no merge mask: no decision-2 match passes the caller's graph test.
extern "C" int32_t hm_landmark_pairs_batch_device(hm_ctx* c, const void* d_best, const void* d_decision, const void* d_nq, const uint32_t* iq,
                                                  uint32_t cap_per_img, uint32_t n_frames, const void* d_world, uint32_t n_world,
                                                  void* d_pairs, void* d_npairs, void* stream_to_wait)
{
    return hm_landmark_matches_batch_device(c, d_best, d_decision, nullptr, d_nq, iq, cap_per_img, n_frames, d_world, n_world, d_pairs,
                                            d_npairs, stream_to_wait);
}

// Timing of the k-NN kernel launches (HIP events on hm_stream()): enable, run, then read the accumulated
// milliseconds and launch count.  hm_timing_get synchronises the pending events.
extern "C" int32_t hm_timing_enable(hm_ctx* c, int32_t on)
{
    return akz_guard([&]() -> int32_t {
        if (!c) return AKZ_E_INVALID;
        c->timing = on != 0;
        return AKZ_OK;
    });
}
extern "C" int32_t hm_timing_get(hm_ctx* c, double* ms, uint64_t* launches, int32_t reset)
{
    return akz_guard([&]() -> int32_t {
        if (!c) return AKZ_E_INVALID;
        for (auto& pr : c->t_pending) {
            float t = 0.0f;
            hipEventSynchronize(pr.second);
            if (hipEventElapsedTime(&t, pr.first, pr.second) == hipSuccess) c->t_ms += (double)t;
            c->t_pool.push_back(pr.first);
            c->t_pool.push_back(pr.second);
        }
        c->t_pending.clear();
        if (ms) *ms = c->t_ms;
        if (launches) *launches = c->t_launches;
        if (reset) {
            c->t_ms = 0.0;
            c->t_launches = 0;
        }
        return AKZ_OK;
    });
}

extern "C" int32_t hm_match(hm_ctx* c, const akz_descriptor* a, uint32_t na, const akz_descriptor* b, uint32_t nb,
                            int32_t rule, uint32_t param_u, float param_f, int32_t symmetric, uint32_t* pairs,
                            uint32_t cap, uint32_t* n_out)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !n_out || (na && !a) || (nb && !b) || (cap && !pairs)) return AKZ_E_INVALID;
        if (rule < 0 || rule > 2) return AKZ_E_INVALID;
        if (na > c->max_q || nb > c->max_t) return AKZ_E_TOO_LARGE;
        *n_out = 0;
        if (na < 2 || nb < 2) {
            // cv-sfm returns no matches (cv-sfm/src/lib.rs:3099-3101); the tutorial/test call sites would
            // panic indexing knn[1] — reported as an invalid argument instead of aborting.
            return rule == HM_RULE_BETTER_BY ? AKZ_OK : AKZ_E_INVALID;
        }
        AKZ_HIP(hipSetDevice(c->device));
        const size_t pair_off = akz_align_up(knn_stage_bytes(2), 256);
        AKZ_TRY(hm_ensure_probs(c, pair_off + sizeof(HmPairProb) + 256));
        uint32_t cnt[2] = {na, nb};
        AKZ_HIP(hipMemcpyAsync(c->d_na, cnt, sizeof(cnt), hipMemcpyHostToDevice, c->stream));
        AKZ_HIP(hipMemcpyAsync(c->d_a, a, (size_t)na * 64, hipMemcpyHostToDevice, c->stream));
        c->resident = false;   // (the staging buffer of the targets is overwritten)
        AKZ_HIP(hipMemcpyAsync(c->d_b, b, (size_t)nb * 64, hipMemcpyHostToDevice, c->stream));
        HmProb p[2] = {{c->d_a, c->d_na, na, c->d_b, c->d_na + 1, nb, c->d_fwd},
                       {c->d_b, c->d_na + 1, nb, c->d_a, c->d_na, na, c->d_rev}};
        AKZ_TRY(launch_knn2(c, p, symmetric ? 2 : 1, na > nb ? na : nb, 0));
        uint32_t kcap = cap < na ? cap : na;
        HmPairProb pp = {c->d_fwd, c->d_rev, c->d_na, c->d_na + 1, na, nb, c->d_pairs, kcap, c->d_npairs};
        HmPairProb* dpp = reinterpret_cast<HmPairProb*>((char*)c->d_probs + pair_off);
        AKZ_TRY(hm_push_probs(c, pair_off, &pp, sizeof(pp)));
        hipLaunchKernelGGL(k_pairs, dim3(1), dim3(1024), 0, c->stream, dpp, (int)rule, param_u, param_f, (int)symmetric);
        AKZ_LAUNCH_CHECK();
        uint32_t n = 0;
        AKZ_HIP(hipMemcpyAsync(&n, c->d_npairs, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        AKZ_HIP(hipStreamSynchronize(c->stream));
        *n_out = n;
        uint32_t ncopy = n < kcap ? n : kcap;
        if (ncopy) AKZ_HIP(hipMemcpy(pairs, c->d_pairs, sizeof(uint32_t) * 2 * (size_t)ncopy, hipMemcpyDeviceToHost));
        return n > cap ? AKZ_E_CAPACITY : AKZ_OK;
    });
}

extern "C" int32_t hm_match_batch_device(hm_ctx* c, const void* d_a, const void* d_na, const void* d_b,
                                         const void* d_nb, uint32_t cap_per_img, const uint32_t* ia,
                                         const uint32_t* ib, uint32_t n_pairs, int32_t rule, uint32_t param_u,
                                         float param_f, int32_t symmetric, void* d_pairs, void* d_n_out,
                                         void* stream_to_wait)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !d_a || !d_na || !d_b || !d_nb || !ia || !ib || !d_pairs || !d_n_out) return AKZ_E_INVALID;
        if (rule < 0 || rule > 2 || cap_per_img == 0 || cap_per_img >= (1u << (kIdxBits - 1))) return AKZ_E_INVALID;
        if (n_pairs == 0) return AKZ_OK;
        AKZ_HIP(hipSetDevice(c->device));
        if (stream_to_wait) {
            AKZ_HIP(hipEventRecord(c->ev, akz_wait_stream(stream_to_wait)));
            AKZ_HIP(hipStreamWaitEvent(c->stream, c->ev, 0));
        }
        const uint32_t ndir = symmetric ? 2u : 1u;
        size_t need = (size_t)n_pairs * cap_per_img * 2;
        if (need > c->bscratch_elems) {
            AKZ_HIP(hipStreamSynchronize(c->stream));
            if (c->d_bfwd) AKZ_HIP(hipFree(c->d_bfwd));
            if (c->d_brev) AKZ_HIP(hipFree(c->d_brev));
            c->d_bfwd = c->d_brev = nullptr;
            AKZ_HIP(hipMalloc(&c->d_bfwd, sizeof(akz_neighbor) * need));
            AKZ_HIP(hipMalloc(&c->d_brev, sizeof(akz_neighbor) * need));
            c->bscratch_elems = need;
        }
        size_t knn_bytes = akz_align_up(knn_stage_bytes(n_pairs * ndir), 256);
        AKZ_TRY(hm_ensure_probs(c, knn_bytes + sizeof(HmPairProb) * n_pairs));
        std::vector<HmProb> hp(n_pairs * ndir);
        std::vector<HmPairProb> hpp(n_pairs);
        const uint4* A = (const uint4*)d_a;
        const uint4* B = (const uint4*)d_b;
        const uint32_t* NA = (const uint32_t*)d_na;
        const uint32_t* NB = (const uint32_t*)d_nb;
        for (uint32_t p = 0; p < n_pairs; ++p) {
            const uint4* qa = A + (size_t)ia[p] * cap_per_img * 4;
            const uint4* tb = B + (size_t)ib[p] * cap_per_img * 4;
            akz_neighbor* fwd = c->d_bfwd + (size_t)p * cap_per_img * 2;
            akz_neighbor* rev = c->d_brev + (size_t)p * cap_per_img * 2;
            hp[p] = HmProb{qa, NA + ia[p], cap_per_img, tb, NB + ib[p], cap_per_img, fwd};
            if (symmetric) hp[n_pairs + p] = HmProb{tb, NB + ib[p], cap_per_img, qa, NA + ia[p], cap_per_img, rev};
            hpp[p] = HmPairProb{fwd, rev, NA + ia[p], NB + ib[p], cap_per_img, cap_per_img,
                                (uint32_t*)d_pairs + (size_t)p * cap_per_img * 2, cap_per_img,
                                (uint32_t*)d_n_out + p};
        }
        AKZ_TRY(launch_knn2(c, hp.data(), n_pairs * ndir, cap_per_img, 0));
        HmPairProb* dpp = reinterpret_cast<HmPairProb*>((char*)c->d_probs + knn_bytes);
        AKZ_TRY(hm_push_probs(c, knn_bytes, hpp.data(), sizeof(HmPairProb) * n_pairs));
        hipLaunchKernelGGL(k_pairs, dim3(n_pairs), dim3(1024), 0, c->stream, dpp, (int)rule, param_u, param_f,
                           (int)symmetric);
        AKZ_LAUNCH_CHECK();
        return AKZ_OK;
    });
}

// ---------------------------------------------------------------------------------------------
// Frame-level place recognition.  Replaces
//   hasher.hash_bag(features)                      cv-sfm/src/lib.rs:672   (HammingHasher<64, 512>, :205,216)
//   lsh_to_frame.knn_values(&lsh, search_num)      cv-sfm/src/lib.rs:622-624
// hash_bag is the matcher's own workload with the codebook as the target set: k_knn_mfma<1> gives every feature
// its nearest codeword (lowest index among equals), k_bag_bits ORs the word bits into the frame's hash.  The
// hashing crate (hamming-lsh 0.3.2) is not vendored in the reference: parity unpinned, see oracle/lsh_oracle.c.

// hash bit w of frame f <- some feature of f has nearest codeword w (bit w at byte w >> 3, position w & 7)
__global__ __launch_bounds__(256) void k_bag_bits(const akz_neighbor* __restrict__ words, const uint32_t* __restrict__ counts,
                                                  uint32_t cap, uint32_t hash_words, uint32_t* __restrict__ hash)
{
    const uint32_t f = blockIdx.y;
    const uint32_t n = min(counts[f], cap);
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t w = words[(size_t)f * cap + i].index;
    atomicOr(&hash[(size_t)f * hash_words + (w >> 5)], 1u << (w & 31u));
}

static int32_t hash_bag_launch(hm_ctx* c, const uint4* d_descs, const uint32_t* d_counts, uint32_t cap, uint32_t n_frames,
                               const uint4* d_codewords, uint32_t n_codewords, uint32_t* d_hash, akz_neighbor* d_words)
{
    // the codeword count has to be readable on the device like every other count: it rides in the staging slot
    const size_t cnt_off = akz_align_up(knn_stage_bytes(n_frames), 256);
    AKZ_TRY(hm_ensure_probs(c, cnt_off + 64));
    AKZ_TRY(hm_push_probs(c, cnt_off, &n_codewords, sizeof(uint32_t)));
    const uint32_t* d_ncw = reinterpret_cast<const uint32_t*>((char*)c->d_probs + cnt_off);
    std::vector<HmProb> hp(n_frames);
    for (uint32_t f = 0; f < n_frames; ++f)
        hp[f] = HmProb{d_descs + (size_t)f * cap * 4, d_counts + f, cap, d_codewords, d_ncw, n_codewords,
                       d_words + (size_t)f * cap};
    AKZ_TRY(launch_knn2(c, hp.data(), n_frames, cap, 0, 1));
    const uint32_t hash_words = n_codewords / 32u;
    AKZ_HIP(hipMemsetAsync(d_hash, 0, sizeof(uint32_t) * (size_t)n_frames * hash_words, c->stream));
    hipLaunchKernelGGL(k_bag_bits, dim3((cap + 255u) / 256u, n_frames), dim3(256), 0, c->stream, d_words, d_counts, cap,
                       hash_words, d_hash);
    AKZ_LAUNCH_CHECK();
    return AKZ_OK;
}

extern "C" int32_t hm_hash_bag_device(hm_ctx* c, const void* d_descs, const void* d_counts, uint32_t cap_per_img,
                                      uint32_t n_frames, const void* d_codewords, uint32_t n_codewords, void* d_hash,
                                      void* d_words, void* stream_to_wait)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !d_descs || !d_counts || !d_codewords || !d_hash || !d_words) return AKZ_E_INVALID;
        if (n_codewords == 0 || (n_codewords & 31u) || n_codewords >= (1u << (kIdxBits - 1))) return AKZ_E_INVALID;
        if (cap_per_img == 0 || cap_per_img >= (1u << (kIdxBits - 1)) || n_frames > 65535u) return AKZ_E_INVALID;
        if (n_frames == 0) return AKZ_OK;
        AKZ_HIP(hipSetDevice(c->device));
        if (stream_to_wait) {
            AKZ_HIP(hipEventRecord(c->ev, akz_wait_stream(stream_to_wait)));
            AKZ_HIP(hipStreamWaitEvent(c->stream, c->ev, 0));
        }
        return hash_bag_launch(c, (const uint4*)d_descs, (const uint32_t*)d_counts, cap_per_img, n_frames,
                               (const uint4*)d_codewords, n_codewords, (uint32_t*)d_hash, (akz_neighbor*)d_words);
    });
}

// grow-only device scratch of the host-buffer place-recognition calls
static int32_t hm_lsh_scratch(hm_ctx* c, size_t bytes)
{
    if (bytes <= c->lsh_bytes) return AKZ_OK;
    AKZ_HIP(hipStreamSynchronize(c->stream));
    if (c->d_lsh) AKZ_HIP(hipFree(c->d_lsh));
    c->d_lsh = nullptr;
    c->lsh_bytes = 0;
    AKZ_HIP(hipMalloc(&c->d_lsh, bytes));
    c->lsh_bytes = bytes;
    return AKZ_OK;
}

extern "C" int32_t hm_hash_bag(hm_ctx* c, const akz_descriptor* feats, uint32_t n, const akz_descriptor* codewords,
                               uint32_t n_codewords, uint8_t* hash, akz_neighbor* words)
{
    return akz_guard([&]() -> int32_t {
        if (!c || (n && !feats) || !codewords || !hash) return AKZ_E_INVALID;
        if (n_codewords == 0 || (n_codewords & 31u)) return AKZ_E_INVALID;
        if (n > c->max_q || n_codewords > c->max_t) return AKZ_E_TOO_LARGE;
        AKZ_HIP(hipSetDevice(c->device));
        const uint32_t cap = n ? n : 1u;
        const size_t hash_off = akz_align_up(sizeof(akz_neighbor) * (size_t)cap, 256);
        AKZ_TRY(hm_lsh_scratch(c, hash_off + n_codewords / 8));
        akz_neighbor* d_words = reinterpret_cast<akz_neighbor*>(c->d_lsh);
        uint32_t* d_hash = reinterpret_cast<uint32_t*>((char*)c->d_lsh + hash_off);
        AKZ_HIP(hipMemcpyAsync(c->d_na, &n, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        if (n) AKZ_HIP(hipMemcpyAsync(c->d_a, feats, (size_t)n * 64, hipMemcpyHostToDevice, c->stream));
        c->resident = false;   // (the staging buffer of the targets is overwritten)
        AKZ_HIP(hipMemcpyAsync(c->d_b, codewords, (size_t)n_codewords * 64, hipMemcpyHostToDevice, c->stream));
        AKZ_TRY(hash_bag_launch(c, c->d_a, c->d_na, cap, 1, c->d_b, n_codewords, d_hash, d_words));
        AKZ_HIP(hipMemcpyAsync(hash, d_hash, n_codewords / 8, hipMemcpyDeviceToHost, c->stream));
        if (words && n)
            AKZ_HIP(hipMemcpyAsync(words, d_words, sizeof(akz_neighbor) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
        AKZ_HIP(hipStreamSynchronize(c->stream));
        return AKZ_OK;
    });
}

// Hamming distance of one query hash to every stored hash: one wave per stored hash (512-byte hashes are one
// 8-byte load per lane), wave reduction.
__global__ __launch_bounds__(256) void k_hash_dist(const uint32_t* __restrict__ q, const uint32_t* __restrict__ hashes,
                                                   uint32_t n, uint32_t words, uint32_t* __restrict__ dist)
{
    const uint32_t i = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (i >= n) return;
    const uint32_t* h = hashes + (size_t)i * words;
    uint32_t acc = 0;
    for (uint32_t w = lane; w < words; w += 64u) acc += (uint32_t)__popc(q[w] ^ h[w]);
#pragma unroll
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) dist[i] = acc;
}

// The k stored hashes nearest to `query` in (distance, index) order — the exact answer where the reference
// asks an approximate index (HggLite::knn_values).  Distances on the device, the k-selection over n (one entry
// per stored frame) on the host.
extern "C" int32_t hm_hash_knn(hm_ctx* c, const uint8_t* query, const uint8_t* hashes, uint32_t n, uint32_t hash_bytes,
                               uint32_t k, akz_neighbor* out, uint32_t* n_out)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !query || (n && !hashes) || !n_out || (k && !out)) return AKZ_E_INVALID;
        if (hash_bytes == 0 || (hash_bytes & 3u)) return AKZ_E_INVALID;
        *n_out = 0;
        if (n == 0 || k == 0) return AKZ_OK;
        AKZ_HIP(hipSetDevice(c->device));
        const uint32_t words = hash_bytes / 4u;
        const size_t q_off = akz_align_up((size_t)n * hash_bytes, 256), d_off = q_off + akz_align_up(hash_bytes, 256);
        AKZ_TRY(hm_lsh_scratch(c, d_off + sizeof(uint32_t) * (size_t)n));
        uint32_t* d_h = reinterpret_cast<uint32_t*>(c->d_lsh);
        uint32_t* d_q = reinterpret_cast<uint32_t*>((char*)c->d_lsh + q_off);
        uint32_t* d_d = reinterpret_cast<uint32_t*>((char*)c->d_lsh + d_off);
        AKZ_HIP(hipMemcpyAsync(d_h, hashes, (size_t)n * hash_bytes, hipMemcpyHostToDevice, c->stream));
        AKZ_HIP(hipMemcpyAsync(d_q, query, hash_bytes, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_hash_dist, dim3((n + 3u) / 4u), dim3(256), 0, c->stream, d_q, d_h, n, words, d_d);
        AKZ_LAUNCH_CHECK();
        std::vector<uint32_t> dist(n);
        AKZ_HIP(hipMemcpyAsync(dist.data(), d_d, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
        AKZ_HIP(hipStreamSynchronize(c->stream));
        std::vector<uint64_t> keys(n);
        for (uint32_t i = 0; i < n; ++i) keys[i] = ((uint64_t)dist[i] << 32) | i;
        const uint32_t m = k < n ? k : n;
        std::partial_sort(keys.begin(), keys.begin() + m, keys.end());
        for (uint32_t i = 0; i < m; ++i) out[i] = akz_neighbor{(uint32_t)keys[i], (uint32_t)(keys[i] >> 32)};
        *n_out = m;
        return AKZ_OK;
    });
}
