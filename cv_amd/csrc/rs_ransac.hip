// rs_ransac.hip — batched two-view geometric verification on gfx950 (SURVEY.md §8a rows R1-R4,
// BASELINE.json configs[3]): eight-point essential matrices for a batch of minimal samples, the four
// candidate poses of each, the triangulation residual of every (pose, match) pair, consensus by inlier
// count.  All f64 VALU work, -ffp-contract=off, the same operation sequence as oracle/ransac_oracle.c.
//
// Reference code implemented here (paths relative to rust-cv/cv):
//   CameraIntrinsics::calibrate (+K1)                     cv-pinhole/src/lib.rs:108-117,191-202   rs_calibrate (host)
//   encode_epipolar_equation, EightPoint::from_matches    eight-point/src/lib.rs:11-58            k_rs_hypotheses
//   EssentialMatrix::possible_unscaled_poses              cv-pinhole/src/essential.rs:114-162,217-231  k_rs_hypotheses
//   CameraToCamera::residual                              cv-core/src/pose.rs:249-295             k_rs_score, k_rs_inliers
//   Consensus::model_inliers                              call sites akaze/tests/estimate_pose.rs:63-67,
//                                                         tutorial ch5 main.rs:70-72, cv-sfm/src/lib.rs:1394-1406
// nalgebra's eigen/SVD and the arrsac sampler are un-vendored: the eigen-solver is the shared cyclic Jacobi
// of include/akz_ransac_math.h, the minimal samples are supplied by the caller, and every hypothesis is
// scored against every match (parity: oracle == HIP, bit for bit).
#include <math.h>
#include <vector>

#include "akz_common.h"
#include "../../include/akz_ransac_math.h"
#include "../../include/akz_p3p_math.h"

namespace {

constexpr double kEpsHyp = 1e-12;   // EightPoint::default epsilon (eight-point/src/lib.rs:60-67)
constexpr int kItersHyp = 1000;     // EightPoint::default iterations
constexpr double kEpsRes = 1e-12;   // try_symmetric_eigen(1e-12, 1024), cv-core/src/pose.rs:272
constexpr int kItersRes = 1024;

__device__ __forceinline__ bool finite_d(double v) { return v == v && fabs(v) != INFINITY; }

// one lane per hypothesis; the 9x9 normal matrix and its eigenvectors live in LDS, interleaved across the
// 64 lanes (element e of lane t at [e*64 + t]) so every access is conflict-free.
__global__ __launch_bounds__(64) void k_rs_hypotheses(const double* __restrict__ ba, const double* __restrict__ bb,
                                                      const uint32_t* __restrict__ sample_idx, uint32_t n_hyp,
                                                      double* __restrict__ poses, uint32_t* __restrict__ ok)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* sM = reinterpret_cast<double*>(smem) + threadIdx.x;        // [81][64]
    double* sV = reinterpret_cast<double*>(smem) + 81 * 64 + threadIdx.x;
    const uint32_t hh = blockIdx.x * 64 + threadIdx.x;
    if (hh >= n_hyp) return;
    double A[8][9];
    for (int i = 0; i < 8; ++i) {
        uint32_t m = sample_idx[(size_t)hh * 8 + i];
        const double* a = ba + (size_t)3 * m;
        const double* b = bb + (size_t)3 * m;
        double az = a[2];
        double ap[3] = {a[0] / az, a[1] / az, a[2] / az};
        double bp[3] = {b[0] / az, b[1] / az, b[2] / az};  // sic: divided by a.z (eight-point/src/lib.rs:16)
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k) A[i][3 * j + k] = ap[j] * bp[k];
    }
    for (int r = 0; r < 9; ++r)
        for (int c = 0; c < 9; ++c) {
            double s = 0.0;
            for (int i = 0; i < 8; ++i) s += A[i][r] * A[i][c];
            sM[(r * 9 + c) * 64] = s;
        }
    akz_rm_jacobi9(sM, sV, 64, kEpsHyp, kItersHyp);
    int best = 0;
    for (int i = 1; i < 9; ++i)
        if (sM[(i * 9 + i) * 64] < sM[(best * 9 + best) * 64]) best = i;
    double E[9];
    bool good = true;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            E[r * 3 + c] = sV[((c * 3 + r) * 9 + best) * 64];  // Matrix3::from_iterator is column-major
            good = good && finite_d(E[r * 3 + c]);
        }
    // possible_unscaled_poses: SVD of E through the eigen-decomposition of E^T E
    double M[9], V[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += E[k * 3 + r] * E[k * 3 + c];
            M[r * 3 + c] = s;
        }
    akz_rm_jacobi3(M, V, 1, kEpsHyp, kItersHyp);
    int ord[3] = {0, 1, 2};
    for (int i = 0; i < 3; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (M[ord[j] * 3 + ord[j]] > M[ord[i] * 3 + ord[i]]) {
                int t = ord[i];
                ord[i] = ord[j];
                ord[j] = t;
            }
    double Vs[9], U[9];
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) Vs[r * 3 + c] = V[r * 3 + ord[c]];
    for (int c = 0; c < 2; ++c) {
        double lam = M[ord[c] * 3 + ord[c]];
        double s = AKZ_RM_SQRT(lam > 0.0 ? lam : 0.0);
        if (!(s > 0.0)) good = false;
        for (int r = 0; r < 3; ++r) {
            double acc = 0.0;
            for (int k = 0; k < 3; ++k) acc += E[r * 3 + k] * Vs[k * 3 + c];
            U[r * 3 + c] = acc / s;
        }
    }
    U[0 * 3 + 2] = U[1 * 3 + 0] * U[2 * 3 + 1] - U[2 * 3 + 0] * U[1 * 3 + 1];
    U[1 * 3 + 2] = U[2 * 3 + 0] * U[0 * 3 + 1] - U[0 * 3 + 0] * U[2 * 3 + 1];
    U[2 * 3 + 2] = U[0 * 3 + 0] * U[1 * 3 + 1] - U[1 * 3 + 0] * U[0 * 3 + 1];
    double detV = Vs[0] * (Vs[4] * Vs[8] - Vs[5] * Vs[7]) - Vs[1] * (Vs[3] * Vs[8] - Vs[5] * Vs[6]) +
                  Vs[2] * (Vs[3] * Vs[7] - Vs[4] * Vs[6]);
    if (detV < 0.0)
        for (int r = 0; r < 3; ++r) Vs[r * 3 + 2] = -Vs[r * 3 + 2];
    // R1 = U W V^T, R2 = U W^T V^T with W = [[0,-1,0],[1,0,0],[0,0,1]]; the products are written out with
    // the same term order as the oracle's generic 3x3 multiply (k = 0,1,2, zeros included)
    const double W[9] = {0.0, -1.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0};
    const double Wt[9] = {0.0, 1.0, 0.0, -1.0, 0.0, 0.0, 0.0, 0.0, 1.0};
    double R[2][9];
    for (int which = 0; which < 2; ++which) {
        const double* Wm = which ? Wt : W;
        double UW[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                double s = 0.0;
                for (int k = 0; k < 3; ++k) s += U[r * 3 + k] * Wm[k * 3 + c];
                UW[r * 3 + c] = s;
            }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                double s = 0.0;
                for (int k = 0; k < 3; ++k) s += UW[r * 3 + k] * Vs[c * 3 + k];  // V^T(k,c) = Vs(c,k)
                R[which][r * 3 + c] = s;
            }
    }
    const double t[3] = {U[2], U[5], U[8]};
    double* out = poses + (size_t)hh * 48;
    for (int p = 0; p < 4; ++p) {
        const double* Rm = R[p & 1];
        double sg = (p & 2) ? -1.0 : 1.0;
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) {
                out[p * 12 + r * 4 + c] = Rm[r * 3 + c];
                good = good && finite_d(Rm[r * 3 + c]);
            }
            out[p * 12 + r * 4 + 3] = sg * t[r];
            good = good && finite_d(t[r]);
        }
    }
    for (int p = 0; p < 4; ++p) ok[(size_t)hh * 4 + p] = good ? 1u : 0u;  // validity per pose
}

// CameraToCamera::residual for one (pose, match) — cv-core/src/pose.rs:249-295.
__device__ double rs_residual(const double* __restrict__ pose, const double* a, const double* b)
{
    double design[16], V[16];
    for (int i = 0; i < 16; ++i) design[i] = 0.0;
    for (int view = 0; view < 2; ++view) {
        double P[12];
        for (int i = 0; i < 12; ++i) P[i] = view == 0 ? ((i == 0 || i == 5 || i == 10) ? 1.0 : 0.0) : pose[i];
        const double* br = view == 0 ? a : b;
        double term[12];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) {
                double s = 0.0;
                for (int k = 0; k < 3; ++k) s += (br[r] * br[k]) * P[k * 4 + c];
                term[r * 4 + c] = P[r * 4 + c] - s;
            }
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) {
                double s = 0.0;
                for (int k = 0; k < 3; ++k) s += term[k * 4 + r] * term[k * 4 + c];
                design[r * 4 + c] += s;
            }
    }
    akz_rm_jacobi4_sym(design, V, kEpsRes, kItersRes);
    // eigenvector of the eigenvalue with the smallest magnitude (first one on ties), selected without a runtime
    // index into V: a dynamic index would move both matrices from registers to scratch memory
    double bestv = fabs(design[0]);
    double p[4] = {V[0], V[4], V[8], V[12]};
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        const double d = fabs(design[i * 4 + i]);
        const bool take = d < bestv;
        bestv = take ? d : bestv;
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = take ? V[r * 4 + i] : p[r];
    }
    if (__builtin_signbit(p[3]))
        for (int i = 0; i < 4; ++i) p[i] = -p[i];
    double nrm = AKZ_RM_SQRT(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    for (int i = 0; i < 4; ++i) p[i] = p[i] / nrm;
    for (int i = 0; i < 4; ++i)
        if (!finite_d(p[i])) return 2.0;
    double q[4];
    for (int r = 0; r < 3; ++r)
        q[r] = ((pose[r * 4 + 0] * p[0] + pose[r * 4 + 1] * p[1]) + pose[r * 4 + 2] * p[2]) + pose[r * 4 + 3] * p[3];
    q[3] = p[3];
    if (__builtin_signbit(q[3]))
        for (int i = 0; i < 4; ++i) q[i] = -q[i];
    double qn = AKZ_RM_SQRT(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    for (int i = 0; i < 4; ++i) q[i] = q[i] / qn;
    double ad = (a[0] * p[0] + a[1] * p[1]) + a[2] * p[2];
    double bd = (b[0] * q[0] + b[1] * q[1]) + b[2] * q[2];
    double res = 0.5 * (1.0 - ad + 1.0 - bd);
    return res == res ? res : 2.0;
}

// grid: (match blocks, pose id = hyp*4 + p).  Inlier count per pose by ballot + one atomic per wave.
__global__ __launch_bounds__(256) void k_rs_score(const double* __restrict__ ba, const double* __restrict__ bb,
                                                  uint32_t n, const double* __restrict__ poses,
                                                  const uint32_t* __restrict__ ok, double thresh,
                                                  uint32_t* __restrict__ counts)
{
    const uint32_t pid = blockIdx.y;
    if (!ok[pid]) return;
    const uint32_t m = blockIdx.x * 256 + threadIdx.x;
    bool inl = false;
    if (m < n) {
        double pose[12];
        for (int i = 0; i < 12; ++i) pose[i] = poses[(size_t)pid * 12 + i];
        double a[3] = {ba[3 * (size_t)m], ba[3 * (size_t)m + 1], ba[3 * (size_t)m + 2]};
        double b[3] = {bb[3 * (size_t)m], bb[3 * (size_t)m + 1], bb[3 * (size_t)m + 2]};
        inl = rs_residual(pose, a, b) < thresh;
    }
    unsigned long long bal = __ballot(inl);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&counts[pid], (uint32_t)__popcll(bal));
}

// argmax of (count, -id): the pose with the most inliers, lowest id on ties (= the oracle's first-maximum scan)
__global__ __launch_bounds__(1024) void k_rs_best(const uint32_t* __restrict__ counts, const uint32_t* __restrict__ ok,
                                                  uint32_t n_pose, uint32_t* __restrict__ best)
{
    __shared__ unsigned long long s_key[16];
    unsigned long long key = 0ull;  // 0 = nothing valid
    for (uint32_t i = threadIdx.x; i < n_pose; i += 1024)
        if (ok[i]) {
            unsigned long long k = ((unsigned long long)(counts[i] + 1u) << 32) | (unsigned long long)(0xFFFFFFFFu - i);
            key = k > key ? k : key;
        }
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long o = __shfl_down(key, off);
        key = o > key ? o : key;
    }
    if ((threadIdx.x & 63) == 0) s_key[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) key = s_key[i] > key ? s_key[i] : key;
        best[0] = key ? (0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull)) : 0xFFFFFFFFu;
        best[1] = key ? (uint32_t)(key >> 32) - 1u : 0u;
    }
}

// inlier indices of the best pose in ascending match order (ordered compaction, one block)
__global__ __launch_bounds__(1024) void k_rs_inliers(const double* __restrict__ ba, const double* __restrict__ bb,
                                                     uint32_t n, const double* __restrict__ poses,
                                                     const uint32_t* __restrict__ best, double thresh,
                                                     uint32_t* __restrict__ inlier_idx, uint32_t cap,
                                                     uint32_t* __restrict__ n_inliers, double* __restrict__ best_pose,
                                                     unsigned long long* __restrict__ n_eval)
{
    __shared__ uint32_t s_wave[16];
    const uint32_t pid = best[0];
    if (pid == 0xFFFFFFFFu) {
        if (threadIdx.x == 0) *n_inliers = 0;
        return;
    }
    if (threadIdx.x == 0 && n_eval) atomicAdd(n_eval, (unsigned long long)n);
    double pose[12];
    for (int i = 0; i < 12; ++i) pose[i] = poses[(size_t)pid * 12 + i];
    if (threadIdx.x < 12) best_pose[threadIdx.x] = pose[threadIdx.x];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t base = 0;
    for (uint32_t m0 = 0; m0 < n; m0 += 1024) {
        uint32_t m = m0 + threadIdx.x;
        bool inl = false;
        if (m < n) {
            double a[3] = {ba[3 * (size_t)m], ba[3 * (size_t)m + 1], ba[3 * (size_t)m + 2]};
            double b[3] = {bb[3 * (size_t)m], bb[3 * (size_t)m + 1], bb[3 * (size_t)m + 2]};
            inl = rs_residual(pose, a, b) < thresh;
        }
        unsigned long long bal = __ballot(inl);
        if (lane == 0) s_wave[wv] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t woff = 0, tot = 0;
        for (int q = 0; q < 16; ++q) {
            if (q < wv) woff += s_wave[q];
            tot += s_wave[q];
        }
        if (inl) {
            uint32_t o = base + woff + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            if (o < cap) inlier_idx[o] = m;
        }
        base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_inliers = base;
}

// ---- ARRSAC-shaped consensus (row R4) ---------------------------------------------------
// arrsac::Arrsac::model_inliers (un-vendored crate, arrsac 0.10; call sites vslam-sandbox/src/main.rs:105-117,
// cv-sfm/src/lib.rs:1394-1412) scores its hypotheses breadth-first, block of matches by block of matches, and drops
// the ones that can no longer win.  The same shape on the device:
//   k_rs_sample        xoshiro256++ minimal samples drawn on the device (the caller need not ship n_hyp x 8 indices)
//   k_rs_score_block   every live pose against the next `block` matches (one wave per pose and 64 matches)
//   k_rs_prune         after each block: the best count so far B, then a pose is retired when
//                        (bound)  count + matches_left < B            — it cannot reach the best: exact, always on
//                        (cap)    it is not among the max_candidates best after the initialisation blocks
//                        (SPRT)   its likelihood ratio (delta/eps)^c ((1-delta)/(1-eps))^(seen-c) exceeds the threshold,
//                                 eps = B / seen (Wald's test as in SPRT-RANSAC; arrsac's likelihood_ratio_threshold)
//                      and the survivors are compacted in ascending pose order (device-side count, no host round trip).
// With the bound alone the winner, its count and its inlier set are those of exhaustive scoring.
struct Xo256 {
    unsigned long long s[4];
};
__host__ __device__ __forceinline__ unsigned long long xo_rotl(unsigned long long x, int k) { return (x << k) | (x >> (64 - k)); }
__host__ __device__ __forceinline__ unsigned long long xo_splitmix(unsigned long long* x)
{
    unsigned long long z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ unsigned long long xo_next(Xo256* g)   // xoshiro256++ (Blackman & Vigna)
{
    const unsigned long long r = xo_rotl(g->s[0] + g->s[3], 23) + g->s[0];
    const unsigned long long t = g->s[1] << 17;
    g->s[2] ^= g->s[0];
    g->s[3] ^= g->s[1];
    g->s[1] ^= g->s[2];
    g->s[0] ^= g->s[3];
    g->s[2] ^= t;
    g->s[3] = xo_rotl(g->s[3], 45);
    return r;
}
// hypothesis h draws K distinct match indices from its own stream: state = splitmix64 chain of seed + h (the
// seed_from_u64 construction), index = high 32 bits x n >> 32, duplicates redrawn
template <int K>
__host__ __device__ __forceinline__ void rs_draw_sample(unsigned long long seed, uint32_t h, uint32_t n, uint32_t* out)
{
    unsigned long long x = seed + 0xD1B54A32D192ED03ull * (unsigned long long)(h + 1u);
    Xo256 g;
    for (int i = 0; i < 4; ++i) g.s[i] = xo_splitmix(&x);
    for (int i = 0; i < K; ++i) {
        uint32_t v;
        bool dup;
        do {
            v = (uint32_t)(((xo_next(&g) >> 32) * (unsigned long long)n) >> 32);
            dup = false;
            for (int j = 0; j < i; ++j) dup = dup || out[j] == v;
        } while (dup);
        out[i] = v;
    }
}

template <int K>
__global__ __launch_bounds__(256) void k_rs_sample(unsigned long long seed, uint32_t n, uint32_t n_hyp, uint32_t* __restrict__ sample_idx)
{
    const uint32_t h = blockIdx.x * 256 + threadIdx.x;
    if (h >= n_hyp) return;
    uint32_t s[K];
    rs_draw_sample<K>(seed, h, n, s);
    for (int i = 0; i < K; ++i) sample_idx[(size_t)h * K + i] = s[i];
}

// alive list initialisation: valid poses in ascending order (one block)
__global__ __launch_bounds__(1024) void k_rs_alive_init(const uint32_t* __restrict__ ok, uint32_t n_pose, uint32_t* __restrict__ alive,
                                                        uint32_t* __restrict__ n_alive)
{
    __shared__ uint32_t s_wave[16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t base = 0;
    for (uint32_t i0 = 0; i0 < n_pose; i0 += 1024) {
        const uint32_t i = i0 + threadIdx.x;
        const bool keep = i < n_pose && ok[i] != 0;
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_wave[wv] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t woff = 0, tot = 0;
        for (int q = 0; q < 16; ++q) {
            if (q < wv) woff += s_wave[q];
            tot += s_wave[q];
        }
        if (keep) alive[base + woff + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = i;
        base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_alive = base;
}

// one wave per (live pose, 64 matches of the block); grid.x covers the worst case, waves beyond the live count exit
template <bool P3P>   // P3P: ba = bearings [n][3], bb = world points [n][4], WorldToCamera::residual (cv-core/src/pose.rs:194-201)
__global__ __launch_bounds__(256) void k_rs_score_block(const double* __restrict__ ba, const double* __restrict__ bb, uint32_t m_lo,
                                                        uint32_t m_hi, const double* __restrict__ poses,
                                                        const uint32_t* __restrict__ alive, const uint32_t* __restrict__ n_alive,
                                                        const uint32_t* __restrict__ first, double thresh,
                                                        uint32_t* __restrict__ counts, unsigned long long* __restrict__ n_eval)
{
    // `first` (optional): score only the list entries from *first on (the poses a re-sampling round appended)
    const uint32_t slot = (first ? *first : 0u) + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= *n_alive) return;
    const uint32_t pid = alive[slot];
    const uint32_t lane = threadIdx.x & 63;
    double pose[12];
    for (int i = 0; i < 12; ++i) pose[i] = poses[(size_t)pid * 12 + i];
    uint32_t cnt = 0;
    for (uint32_t m0 = m_lo + blockIdx.y * 64; m0 < m_hi; m0 += gridDim.y * 64) {
        const uint32_t m = m0 + lane;
        bool inl = false;
        if (m < m_hi) {
            double a[3] = {ba[3 * (size_t)m], ba[3 * (size_t)m + 1], ba[3 * (size_t)m + 2]};
            if (P3P) {
                double wp[4] = {bb[4 * (size_t)m], bb[4 * (size_t)m + 1], bb[4 * (size_t)m + 2], bb[4 * (size_t)m + 3]};
                inl = akz_w2c_residual(pose, a, wp) < thresh;
            } else {
                double b[3] = {bb[3 * (size_t)m], bb[3 * (size_t)m + 1], bb[3 * (size_t)m + 2]};
                inl = rs_residual(pose, a, b) < thresh;
            }
        }
        cnt += (uint32_t)__popcll(__ballot(inl));
    }
    if (lane == 0) {
        if (cnt) atomicAdd(&counts[pid], cnt);
        if (blockIdx.y == 0) atomicAdd(n_eval, (unsigned long long)(m_hi - m_lo));
    }
}

struct RsPrune {
    uint32_t seen, n_total;       // matches scored so far / in all
    uint32_t cap;                 // keep at most this many poses from now on (0 = no cap)
    uint32_t use_sprt;
    double log_delta, log_1m_delta, log_ratio;   // ln(delta), ln(1 - delta), ln(likelihood ratio threshold)
    const double* log_table;      // ln(k), k = 0 .. n_total, filled by the host's libm (no device transcendental decides)
};

__global__ __launch_bounds__(1024) void k_rs_prune(RsPrune P, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ alive_in,
                                                   const uint32_t* __restrict__ n_in, uint32_t* __restrict__ alive_out,
                                                   uint32_t* __restrict__ n_out)
{
    __shared__ uint32_t s_wave[16], s_wave_t[16];
    __shared__ uint32_t s_hist[2048];
    __shared__ uint32_t s_best, s_T, s_budget;
    const uint32_t n = *n_in;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_best = 0;
    for (int i = threadIdx.x; i < 2048; i += 1024) s_hist[i] = 0;
    __syncthreads();
    uint32_t lmax = 0;
    for (uint32_t i = threadIdx.x; i < n; i += 1024) {
        const uint32_t c = counts[alive_in[i]];
        lmax = c > lmax ? c : lmax;
        if (P.cap) atomicAdd(&s_hist[c < 2047u ? c : 2047u], 1u);
    }
    atomicMax(&s_best, lmax);
    __syncthreads();
    const uint32_t best = s_best, left = P.n_total - P.seen;
    if (threadIdx.x == 0) {
        // count threshold of the cap: poses with count > T all stay, those with count == T in pose order up to the budget
        uint32_t T = 0, budget = 0xFFFFFFFFu;
        if (P.cap && n > P.cap) {
            uint32_t acc = 0;
            int t = 2047;
            for (; t >= 0; --t) {
                if (acc + s_hist[t] >= P.cap) break;
                acc += s_hist[t];
            }
            T = (uint32_t)(t < 0 ? 0 : t);
            budget = P.cap - acc;
        }
        s_T = T;
        s_budget = budget;
    }
    __syncthreads();
    const uint32_t T = s_T;
    // SPRT constants: eps = best / seen
    double l_in = 0.0, l_out = 0.0;
    if (P.use_sprt && best > 0 && best < P.seen) {
        // eps = best / seen: ln(eps) = ln(best) - ln(seen), ln(1 - eps) = ln(seen - best) - ln(seen)
        l_in = P.log_delta - (P.log_table[best] - P.log_table[P.seen]);                  // per inlier  (negative)
        l_out = P.log_1m_delta - (P.log_table[P.seen - best] - P.log_table[P.seen]);     // per outlier (positive)
    }
    uint32_t base = 0, ties_before = 0;
    for (uint32_t i0 = 0; i0 < n; i0 += 1024) {
        const uint32_t i = i0 + threadIdx.x;
        bool keep = false, tie = false;
        uint32_t pid = 0;
        if (i < n) {
            pid = alive_in[i];
            const uint32_t c = counts[pid];
            keep = c + left >= best;                                                   // bound (exact)
            if (keep && P.use_sprt && l_out > 0.0)
                keep = (double)c * l_in + (double)(P.seen - c) * l_out <= P.log_ratio || c == best;
            const uint32_t cc = c < 2047u ? c : 2047u;
            if (keep && P.cap && n > P.cap) {
                if (cc < T) keep = false;
                tie = keep && cc == T;
            }
        }
        // ties at the cap threshold are admitted in pose order while the budget lasts
        const unsigned long long tb = __ballot(tie);
        if (lane == 0) s_wave_t[wv] = (uint32_t)__popcll(tb);
        __syncthreads();
        uint32_t toff = 0, ttot = 0;
        for (int q = 0; q < 16; ++q) {
            if (q < wv) toff += s_wave_t[q];
            ttot += s_wave_t[q];
        }
        if (tie) {
            const uint32_t rank = ties_before + toff + (uint32_t)__popcll(tb & ((1ull << lane) - 1ull));
            if (rank >= s_budget) keep = false;
        }
        ties_before += ttot;
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_wave[wv] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t woff = 0, tot = 0;
        for (int q = 0; q < 16; ++q) {
            if (q < wv) woff += s_wave[q];
            tot += s_wave[q];
        }
        if (keep) alive_out[base + woff + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = pid;
        base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_out = base;
}

// inlier-guided re-sampling (arrsac's estimations_per_block): E new minimal samples drawn among the inliers (over the
// matches seen so far, list L of length *nL) of the best pose; disabled (flag 0) when there are fewer than K inliers
template <int K>
__global__ __launch_bounds__(256) void k_rs_resample(unsigned long long seed, uint32_t next_h, uint32_t E,
                                                     const uint32_t* __restrict__ L, const uint32_t* __restrict__ nL,
                                                     uint32_t* __restrict__ sample_idx, uint32_t* __restrict__ enable)
{
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    const uint32_t n = *nL;
    if (e == 0) *enable = n >= (uint32_t)K ? 1u : 0u;
    if (e >= E) return;
    const uint32_t h = next_h + e;
    uint32_t s[K];
    if (n >= (uint32_t)K) {
        rs_draw_sample<K>(seed ^ 0xA5A5A5A55A5A5A5Aull, h, n, s);
        for (int i = 0; i < K; ++i) s[i] = L[s[i]];
    } else {
        for (int i = 0; i < K; ++i) s[i] = (uint32_t)i;    // placeholder: the hypotheses are gated off below
    }
    for (int i = 0; i < K; ++i) sample_idx[(size_t)h * K + i] = s[i];
}

// valid new poses join the live list in (hypothesis, pose) order — their ids exceed every id already in it — with
// zeroed counters; *first receives the list length before the append
__global__ __launch_bounds__(1024) void k_rs_alive_append(const uint32_t* __restrict__ ok, uint32_t base_pid, uint32_t n_new,
                                                          const uint32_t* __restrict__ enable, uint32_t* __restrict__ alive,
                                                          uint32_t* __restrict__ n_alive, uint32_t* __restrict__ first,
                                                          uint32_t* __restrict__ counts)
{
    __shared__ uint32_t s_wave[16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t old = *n_alive, on = *enable;
    uint32_t base = old;
    for (uint32_t i0 = 0; i0 < n_new; i0 += 1024) {
        const uint32_t i = i0 + threadIdx.x;
        const bool keep = on && i < n_new && ok[base_pid + i] != 0;
        if (i < n_new) counts[base_pid + i] = 0u;
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_wave[wv] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t woff = 0, tot = 0;
        for (int q = 0; q < 16; ++q) {
            if (q < wv) woff += s_wave[q];
            tot += s_wave[q];
        }
        if (keep) alive[base + woff + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = base_pid + i;
        base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *first = old;
        *n_alive = base;
    }
}

// argmax of (count, -id) over the survivors (all of them have seen every match)
__global__ __launch_bounds__(1024) void k_rs_best_alive(const uint32_t* __restrict__ counts, const uint32_t* __restrict__ alive,
                                                        const uint32_t* __restrict__ n_alive, uint32_t* __restrict__ best)
{
    __shared__ unsigned long long s_key[16];
    unsigned long long key = 0ull;
    const uint32_t n = *n_alive;
    for (uint32_t i = threadIdx.x; i < n; i += 1024) {
        const uint32_t pid = alive[i];
        const unsigned long long k = ((unsigned long long)(counts[pid] + 1u) << 32) | (unsigned long long)(0xFFFFFFFFu - pid);
        key = k > key ? k : key;
    }
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long o = __shfl_down(key, off);
        key = o > key ? o : key;
    }
    if ((threadIdx.x & 63) == 0) s_key[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) key = s_key[i] > key ? s_key[i] : key;
        best[0] = key ? (0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull)) : 0xFFFFFFFFu;
        best[1] = key ? (uint32_t)(key >> 32) - 1u : 0u;
        best[2] = n;
    }
}

// ---- PnP: Lambda Twist hypotheses + WorldToCamera residual (row R5) ----------------------------------
__global__ __launch_bounds__(64) void k_p3p_hypotheses(const double* __restrict__ bearings, const double* __restrict__ world,
                                                       const uint32_t* __restrict__ sample_idx, uint32_t n_hyp,
                                                       double* __restrict__ poses, uint32_t* __restrict__ ok)
{
    const uint32_t hh = blockIdx.x * 64 + threadIdx.x;
    if (hh >= n_hyp) return;
    double b3[9], w3[12], P[48];
    for (int i = 0; i < 3; ++i) {
        uint32_t m = sample_idx[(size_t)hh * 3 + i];
        for (int k = 0; k < 3; ++k) b3[3 * i + k] = bearings[(size_t)3 * m + k];
        for (int k = 0; k < 4; ++k) w3[4 * i + k] = world[(size_t)4 * m + k];
    }
    int np = akz_p3p_poses(b3, w3, 5, P);  // LambdaTwist::default(): 5 Gauss-Newton iterations
    for (int p = 0; p < 4; ++p) {
        ok[(size_t)hh * 4 + p] = p < np ? 1u : 0u;
        if (p < np)
            for (int i = 0; i < 12; ++i) poses[(size_t)hh * 48 + p * 12 + i] = P[p * 12 + i];
    }
}

__global__ __launch_bounds__(256) void k_p3p_score(const double* __restrict__ bearings, const double* __restrict__ world,
                                                   uint32_t n, const double* __restrict__ poses,
                                                   const uint32_t* __restrict__ ok, double thresh,
                                                   uint32_t* __restrict__ counts)
{
    const uint32_t pid = blockIdx.y;
    if (!ok[pid]) return;
    const uint32_t m = blockIdx.x * 256 + threadIdx.x;
    bool inl = false;
    if (m < n) {
        double pose[12];
        for (int i = 0; i < 12; ++i) pose[i] = poses[(size_t)pid * 12 + i];
        double b[3] = {bearings[3 * (size_t)m], bearings[3 * (size_t)m + 1], bearings[3 * (size_t)m + 2]};
        double w[4] = {world[4 * (size_t)m], world[4 * (size_t)m + 1], world[4 * (size_t)m + 2], world[4 * (size_t)m + 3]};
        inl = akz_w2c_residual(pose, b, w) < thresh;
    }
    unsigned long long bal = __ballot(inl);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&counts[pid], (uint32_t)__popcll(bal));
}

__global__ __launch_bounds__(1024) void k_p3p_inliers(const double* __restrict__ bearings, const double* __restrict__ world,
                                                      uint32_t n, const double* __restrict__ poses,
                                                      const uint32_t* __restrict__ best, double thresh,
                                                      uint32_t* __restrict__ inlier_idx, uint32_t cap,
                                                      uint32_t* __restrict__ n_inliers, double* __restrict__ best_pose,
                                                      unsigned long long* __restrict__ n_eval)
{
    __shared__ uint32_t s_wave[16];
    const uint32_t pid = best[0];
    if (pid == 0xFFFFFFFFu) {
        if (threadIdx.x == 0) *n_inliers = 0;
        return;
    }
    if (threadIdx.x == 0 && n_eval) atomicAdd(n_eval, (unsigned long long)n);
    double pose[12];
    for (int i = 0; i < 12; ++i) pose[i] = poses[(size_t)pid * 12 + i];
    if (threadIdx.x < 12) best_pose[threadIdx.x] = pose[threadIdx.x];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t base = 0;
    for (uint32_t m0 = 0; m0 < n; m0 += 1024) {
        uint32_t m = m0 + threadIdx.x;
        bool inl = false;
        if (m < n) {
            double b[3] = {bearings[3 * (size_t)m], bearings[3 * (size_t)m + 1], bearings[3 * (size_t)m + 2]};
            double w[4] = {world[4 * (size_t)m], world[4 * (size_t)m + 1], world[4 * (size_t)m + 2], world[4 * (size_t)m + 3]};
            inl = akz_w2c_residual(pose, b, w) < thresh;
        }
        unsigned long long bal = __ballot(inl);
        if (lane == 0) s_wave[wv] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t woff = 0, tot = 0;
        for (int q = 0; q < 16; ++q) {
            if (q < wv) woff += s_wave[q];
            tot += s_wave[q];
        }
        if (inl) {
            uint32_t o = base + woff + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            if (o < cap) inlier_idx[o] = m;
        }
        base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_inliers = base;
}

}  // namespace

struct rs_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t max_matches = 0, max_hyp = 0;
    double *d_a = nullptr, *d_b = nullptr, *d_w = nullptr, *d_poses = nullptr, *d_best_pose = nullptr;
    uint32_t *d_samples = nullptr, *d_ok = nullptr, *d_counts = nullptr, *d_best = nullptr, *d_inl = nullptr,
             *d_ninl = nullptr;
    uint32_t *d_alive[2] = {nullptr, nullptr}, *d_nalive = nullptr;   // ARRSAC: live pose lists (ping-pong) + counts [2]
    unsigned long long* d_neval = nullptr;                             // residuals evaluated
    double* d_logtab = nullptr;                                        // ln(k), k = 0 .. max_matches (host libm values)
    uint32_t *d_first = nullptr, *d_enable = nullptr;                  // re-sampling: first appended slot, enable flag
    uint32_t last_hyp = 0;
};

extern "C" int32_t rs_create(int32_t device, uint32_t max_matches, uint32_t max_hyp, rs_ctx** out)
{
    return akz_guard([&]() -> int32_t {
        if (!out || max_matches < 8 || max_hyp == 0 || max_hyp > (1u << 28)) return AKZ_E_INVALID;
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return AKZ_E_NO_DEVICE;
        AKZ_HIP(hipSetDevice(device));
        rs_ctx* c = new rs_ctx();
        c->device = device;
        c->max_matches = max_matches;
        c->max_hyp = max_hyp;
        AKZ_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        AKZ_HIP(hipMalloc(&c->d_a, sizeof(double) * 3 * (size_t)max_matches));
        AKZ_HIP(hipMalloc(&c->d_b, sizeof(double) * 3 * (size_t)max_matches));
        AKZ_HIP(hipMalloc(&c->d_poses, sizeof(double) * 48 * (size_t)max_hyp));
        AKZ_HIP(hipMalloc(&c->d_best_pose, sizeof(double) * 12));
        AKZ_HIP(hipMalloc(&c->d_samples, sizeof(uint32_t) * 8 * (size_t)max_hyp));
        AKZ_HIP(hipMalloc(&c->d_ok, sizeof(uint32_t) * 4 * (size_t)max_hyp));
        AKZ_HIP(hipMalloc(&c->d_w, sizeof(double) * 4 * (size_t)max_matches));
        AKZ_HIP(hipMalloc(&c->d_counts, sizeof(uint32_t) * 4 * (size_t)max_hyp));
        AKZ_HIP(hipMalloc(&c->d_best, sizeof(uint32_t) * 4));
        AKZ_HIP(hipMalloc(&c->d_inl, sizeof(uint32_t) * (size_t)max_matches));
        AKZ_HIP(hipMalloc(&c->d_ninl, sizeof(uint32_t) * 4));
        AKZ_HIP(hipMalloc(&c->d_alive[0], sizeof(uint32_t) * 4 * (size_t)max_hyp));
        AKZ_HIP(hipMalloc(&c->d_alive[1], sizeof(uint32_t) * 4 * (size_t)max_hyp));
        AKZ_HIP(hipMalloc(&c->d_nalive, sizeof(uint32_t) * 4));
        AKZ_HIP(hipMalloc(&c->d_neval, sizeof(unsigned long long) * 2));
        AKZ_HIP(hipMalloc(&c->d_first, sizeof(uint32_t) * 2));
        c->d_enable = c->d_first + 1;
        {
            // the SPRT's logarithms come from the host's libm, tabulated once: the retirement decisions then do not
            // depend on the device's log() (oracle/arrsac_oracle.c builds the same table)
            std::vector<double> tab((size_t)max_matches + 1);
            for (uint32_t k = 0; k <= max_matches; ++k) tab[k] = log((double)k);
            AKZ_HIP(hipMalloc(&c->d_logtab, sizeof(double) * tab.size()));
            AKZ_HIP(hipMemcpy(c->d_logtab, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice));
        }
        *out = c;
        return AKZ_OK;
    });
}

extern "C" int32_t rs_destroy(rs_ctx* c)
{
    return akz_guard([&]() -> int32_t {
        if (!c) return AKZ_OK;
        hipSetDevice(c->device);
        if (c->stream) hipStreamSynchronize(c->stream);
        hipFree(c->d_a); hipFree(c->d_b); hipFree(c->d_w); hipFree(c->d_poses); hipFree(c->d_best_pose); hipFree(c->d_samples);
        hipFree(c->d_ok); hipFree(c->d_counts); hipFree(c->d_best); hipFree(c->d_inl); hipFree(c->d_ninl);
        hipFree(c->d_alive[0]); hipFree(c->d_alive[1]); hipFree(c->d_nalive); hipFree(c->d_neval);
        hipFree(c->d_logtab); hipFree(c->d_first);
        if (c->stream) hipStreamDestroy(c->stream);
        delete c;
        return AKZ_OK;
    });
}

// cv_pinhole::CameraIntrinsics::calibrate / CameraIntrinsicsK1Distortion::calibrate: host scalar math.
extern "C" int32_t rs_calibrate(const double* intr, int32_t use_k1, double k1, const akz_keypoint* kps, uint32_t n,
                                double* out)
{
    return akz_guard([&]() -> int32_t {
        if (!intr || (n && (!kps || !out))) return AKZ_E_INVALID;
        for (uint32_t i = 0; i < n; ++i) {
            double cx = (double)kps[i].x - intr[2], cy = (double)kps[i].y - intr[3];
            double y = cy / intr[1];
            double x = (cx - intr[4] * y) / intr[0];
            if (use_k1) {
                double r2 = x * x + y * y;
                double d = 1.0 + k1 * r2;
                x = x / d;
                y = y / d;
            }
            double nrm = sqrt(x * x + y * y + 1.0 * 1.0);
            out[3 * i + 0] = x / nrm;
            out[3 * i + 1] = y / nrm;
            out[3 * i + 2] = 1.0 / nrm;
        }
        return AKZ_OK;
    });
}

extern "C" int32_t rs_essential_batch(rs_ctx* c, const double* bearings_a, const double* bearings_b, uint32_t n,
                                      const uint32_t* sample_idx, uint32_t n_hyp, double thresh, double* best_pose,
                                      uint32_t* best_id, uint32_t* inlier_idx, uint32_t cap, uint32_t* n_inliers)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !bearings_a || !bearings_b || !sample_idx || !best_pose || !best_id || !n_inliers || (cap && !inlier_idx))
            return AKZ_E_INVALID;
        if (n < 8 || n_hyp == 0) return AKZ_E_INVALID;  // EightPoint::MIN_SAMPLES (eight-point/src/lib.rs:73)
        if (n > c->max_matches || n_hyp > c->max_hyp) return AKZ_E_TOO_LARGE;
        for (size_t i = 0; i < (size_t)n_hyp * 8; ++i)
            if (sample_idx[i] >= n) return AKZ_E_INVALID;
        AKZ_HIP(hipSetDevice(c->device));
        hipStream_t s = c->stream;
        AKZ_HIP(hipMemcpyAsync(c->d_a, bearings_a, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, s));
        AKZ_HIP(hipMemcpyAsync(c->d_b, bearings_b, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, s));
        AKZ_HIP(hipMemcpyAsync(c->d_samples, sample_idx, sizeof(uint32_t) * 8 * (size_t)n_hyp, hipMemcpyHostToDevice, s));
        AKZ_HIP(hipMemsetAsync(c->d_counts, 0, sizeof(uint32_t) * 4 * (size_t)n_hyp, s));
        hipLaunchKernelGGL(k_rs_hypotheses, dim3((n_hyp + 63) / 64), dim3(64), sizeof(double) * 162 * 64, s, c->d_a, c->d_b,
                           c->d_samples, n_hyp, c->d_poses, c->d_ok);
        AKZ_LAUNCH_CHECK();
        // grid.y is limited to 65535: score the poses in slabs
        const uint32_t n_pose = n_hyp * 4;
        for (uint32_t p0 = 0; p0 < n_pose; p0 += 65532) {
            uint32_t np = n_pose - p0 < 65532 ? n_pose - p0 : 65532;
            hipLaunchKernelGGL(k_rs_score, dim3((n + 255) / 256, np), dim3(256), 0, s, c->d_a, c->d_b, n,
                               c->d_poses + (size_t)p0 * 12, c->d_ok + p0, thresh, c->d_counts + p0);
            AKZ_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(k_rs_best, dim3(1), dim3(1024), 0, s, c->d_counts, c->d_ok, n_pose, c->d_best);
        AKZ_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_rs_inliers, dim3(1), dim3(1024), 0, s, c->d_a, c->d_b, n, c->d_poses, c->d_best, thresh,
                           c->d_inl, n, c->d_ninl, c->d_best_pose, (unsigned long long*)nullptr);
        AKZ_LAUNCH_CHECK();
        uint32_t best[2] = {0, 0}, ninl = 0;
        AKZ_HIP(hipMemcpyAsync(best, c->d_best, sizeof(best), hipMemcpyDeviceToHost, s));
        AKZ_HIP(hipMemcpyAsync(&ninl, c->d_ninl, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        AKZ_HIP(hipMemcpyAsync(best_pose, c->d_best_pose, sizeof(double) * 12, hipMemcpyDeviceToHost, s));
        AKZ_HIP(hipStreamSynchronize(s));
        c->last_hyp = n_hyp;
        *best_id = best[0];
        *n_inliers = ninl;
        if (best[0] == 0xFFFFFFFFu) {
            *n_inliers = 0;
            return AKZ_OK;  // Consensus::model_inliers returned None: no hypothesis produced a model
        }
        uint32_t ncopy = ninl < cap ? ninl : cap;
        if (ncopy) AKZ_HIP(hipMemcpy(inlier_idx, c->d_inl, sizeof(uint32_t) * ncopy, hipMemcpyDeviceToHost));
        return ninl > cap ? AKZ_E_CAPACITY : AKZ_OK;
    });
}

// Consensus::model_inliers in ARRSAC's shape: breadth-first block scoring with retirement (see the kernels above), for
// EightPoint (two bearing sets, 8-match samples) and for LambdaTwist (bearings + world points, 3-match samples).
// sample_idx == NULL draws the minimal samples on the device.
template <bool P3P>
static int32_t arrsac_run(rs_ctx* c, const double* in_a, const double* in_b, uint32_t n, const uint32_t* sample_idx,
                          const rs_arrsac_params* prm, double* best_pose, uint32_t* best_id, uint32_t* inlier_idx,
                          uint32_t cap, uint32_t* n_inliers, rs_arrsac_stats* stats)
{
    constexpr uint32_t K = P3P ? 3u : 8u, BW = P3P ? 4u : 3u;   // sample size; doubles per element of the second input
    if (!c || !in_a || !in_b || !prm || !best_pose || !best_id || !n_inliers || (cap && !inlier_idx)) return AKZ_E_INVALID;
    if (prm->struct_size != sizeof(rs_arrsac_params)) return AKZ_E_INVALID;
    const uint32_t n_hyp = prm->n_hypotheses;
    if (n < K || n_hyp == 0 || prm->block_size == 0) return AKZ_E_INVALID;   // MIN_SAMPLES (eight-point/src/lib.rs:73, lambda-twist/src/lib.rs:333)
    if (prm->reserved != 0 || (prm->flags & ~(RS_PRUNE_BOUND | RS_PRUNE_SPRT | RS_PRUNE_HALVE))) return AKZ_E_INVALID;
    const uint32_t E = prm->estimations_per_block;
    // every block but the last may add E hypotheses: they need room in the context's pose arrays
    const uint64_t n_blocks_max = ((uint64_t)n + prm->block_size - 1) / prm->block_size;
    if (n > c->max_matches || n_hyp > c->max_hyp || (uint64_t)n_hyp + (uint64_t)E * n_blocks_max > c->max_hyp)
        return AKZ_E_TOO_LARGE;
    if ((prm->flags & RS_PRUNE_SPRT) && !(prm->sprt_delta > 0.0 && prm->sprt_delta < 1.0 && prm->sprt_ratio > 1.0))
        return AKZ_E_INVALID;
    if (sample_idx)
        for (size_t i = 0; i < (size_t)n_hyp * K; ++i)
            if (sample_idx[i] >= n) return AKZ_E_INVALID;
    AKZ_HIP(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    double* d_second = P3P ? c->d_w : c->d_b;
    AKZ_HIP(hipMemcpyAsync(c->d_a, in_a, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, s));
    AKZ_HIP(hipMemcpyAsync(d_second, in_b, sizeof(double) * BW * (size_t)n, hipMemcpyHostToDevice, s));
    if (sample_idx) {
        AKZ_HIP(hipMemcpyAsync(c->d_samples, sample_idx, sizeof(uint32_t) * K * (size_t)n_hyp, hipMemcpyHostToDevice, s));
    } else {
        hipLaunchKernelGGL((k_rs_sample<(int)K>), dim3((n_hyp + 255) / 256), dim3(256), 0, s, (unsigned long long)prm->seed, n,
                           n_hyp, c->d_samples);
        AKZ_LAUNCH_CHECK();
    }
    AKZ_HIP(hipMemsetAsync(c->d_counts, 0, sizeof(uint32_t) * 4 * (size_t)n_hyp, s));
    AKZ_HIP(hipMemsetAsync(c->d_neval, 0, sizeof(unsigned long long) * 2, s));
    if (P3P)
        hipLaunchKernelGGL(k_p3p_hypotheses, dim3((n_hyp + 63) / 64), dim3(64), 0, s, c->d_a, c->d_w, c->d_samples, n_hyp,
                           c->d_poses, c->d_ok);
    else
        hipLaunchKernelGGL(k_rs_hypotheses, dim3((n_hyp + 63) / 64), dim3(64), sizeof(double) * 162 * 64, s, c->d_a, c->d_b,
                           c->d_samples, n_hyp, c->d_poses, c->d_ok);
    AKZ_LAUNCH_CHECK();
    const uint32_t n_pose = n_hyp * 4;
    hipLaunchKernelGGL(k_rs_alive_init, dim3(1), dim3(1024), 0, s, c->d_ok, n_pose, c->d_alive[0], c->d_nalive);
    AKZ_LAUNCH_CHECK();
    // (the cap ranks poses through a 2048-bin histogram of their counts: counts above 2046 share the top bin)
    int cur = 0;
    uint32_t blocks = 0;
    // the live count is known to the host only as an upper bound: n_pose before the cap applies, the cap after
    uint32_t live_bound = n_pose;
    const bool prune = (prm->flags & (RS_PRUNE_BOUND | RS_PRUNE_SPRT)) != 0 || prm->max_candidates != 0 || E != 0;
    uint32_t next_h = n_hyp;                                  // first hypothesis slot of the next re-sampling round
    for (uint32_t m_lo = 0; m_lo < n;) {
        // without pruning there is nothing to decide between blocks: one block = all matches
        const uint32_t bs = prune ? prm->block_size : n;
        const uint32_t m_hi = m_lo + bs < n ? m_lo + bs : n;
        const uint32_t chunks = (m_hi - m_lo + 63) / 64;
        const uint32_t gy = chunks < 16 ? chunks : 16;
        hipLaunchKernelGGL((k_rs_score_block<P3P>), dim3((live_bound + 3) / 4, gy), dim3(256), 0, s, c->d_a, d_second, m_lo, m_hi,
                           c->d_poses, c->d_alive[cur], c->d_nalive + cur, (const uint32_t*)nullptr, prm->threshold, c->d_counts,
                           c->d_neval);
        AKZ_LAUNCH_CHECK();
        ++blocks;
        m_lo = m_hi;
        if (prune && m_lo < n) {
            RsPrune P;
            P.seen = m_lo;
            P.n_total = n;
            P.cap = 0u;
            if (prm->max_candidates && blocks >= prm->init_blocks) {
                P.cap = prm->max_candidates;
                if (prm->flags & RS_PRUNE_HALVE) {             // ARRSAC's shrinking candidate set: half per block, never empty
                    const uint32_t sh = blocks - prm->init_blocks;
                    P.cap = sh >= 31 ? 0u : P.cap >> sh;
                    if (P.cap == 0) P.cap = 1;
                }
            }
            P.log_table = c->d_logtab;
            P.use_sprt = (prm->flags & RS_PRUNE_SPRT) ? 1u : 0u;
            P.log_delta = P.use_sprt ? log(prm->sprt_delta) : 0.0;
            P.log_1m_delta = P.use_sprt ? log(1.0 - prm->sprt_delta) : 0.0;
            P.log_ratio = P.use_sprt ? log(prm->sprt_ratio) : 0.0;
            hipLaunchKernelGGL(k_rs_prune, dim3(1), dim3(1024), 0, s, P, c->d_counts, c->d_alive[cur], c->d_nalive + cur,
                               c->d_alive[cur ^ 1], c->d_nalive + (cur ^ 1));
            AKZ_LAUNCH_CHECK();
            cur ^= 1;
            if (P.cap && P.cap < live_bound) live_bound = P.cap;
            if (E && blocks >= prm->init_blocks) {
                // inlier-guided re-sampling: list the inliers (matches seen so far) of the best survivor, draw E minimal
                // samples among them, estimate, and let the valid poses join the live list after catching up on [0, seen)
                hipLaunchKernelGGL(k_rs_best_alive, dim3(1), dim3(1024), 0, s, c->d_counts, c->d_alive[cur], c->d_nalive + cur,
                                   c->d_best);
                AKZ_LAUNCH_CHECK();
                if (P3P)
                    hipLaunchKernelGGL(k_p3p_inliers, dim3(1), dim3(1024), 0, s, c->d_a, c->d_w, m_lo, c->d_poses, c->d_best,
                                       prm->threshold, c->d_inl, m_lo, c->d_ninl, c->d_best_pose, c->d_neval);
                else
                    hipLaunchKernelGGL(k_rs_inliers, dim3(1), dim3(1024), 0, s, c->d_a, c->d_b, m_lo, c->d_poses, c->d_best,
                                       prm->threshold, c->d_inl, m_lo, c->d_ninl, c->d_best_pose, c->d_neval);
                AKZ_LAUNCH_CHECK();
                hipLaunchKernelGGL((k_rs_resample<(int)K>), dim3((E + 255) / 256), dim3(256), 0, s, (unsigned long long)prm->seed,
                                   next_h, E, c->d_inl, c->d_ninl, c->d_samples, c->d_enable);
                AKZ_LAUNCH_CHECK();
                if (P3P)
                    hipLaunchKernelGGL(k_p3p_hypotheses, dim3((E + 63) / 64), dim3(64), 0, s, c->d_a, c->d_w,
                                       c->d_samples + (size_t)next_h * K, E, c->d_poses + (size_t)next_h * 48, c->d_ok + (size_t)next_h * 4);
                else
                    hipLaunchKernelGGL(k_rs_hypotheses, dim3((E + 63) / 64), dim3(64), sizeof(double) * 162 * 64, s, c->d_a, c->d_b,
                                       c->d_samples + (size_t)next_h * K, E, c->d_poses + (size_t)next_h * 48, c->d_ok + (size_t)next_h * 4);
                AKZ_LAUNCH_CHECK();
                hipLaunchKernelGGL(k_rs_alive_append, dim3(1), dim3(1024), 0, s, c->d_ok, next_h * 4, E * 4, c->d_enable,
                                   c->d_alive[cur], c->d_nalive + cur, c->d_first, c->d_counts);
                AKZ_LAUNCH_CHECK();
                const uint32_t cchunks = (m_lo + 63) / 64;
                hipLaunchKernelGGL((k_rs_score_block<P3P>), dim3(E, cchunks < 16 ? cchunks : 16), dim3(256), 0, s, c->d_a, d_second,
                                   0u, m_lo, c->d_poses, c->d_alive[cur], c->d_nalive + cur, (const uint32_t*)c->d_first,
                                   prm->threshold, c->d_counts, c->d_neval);
                AKZ_LAUNCH_CHECK();
                next_h += E;
                live_bound += 4 * E;
            }
        }
    }
    hipLaunchKernelGGL(k_rs_best_alive, dim3(1), dim3(1024), 0, s, c->d_counts, c->d_alive[cur], c->d_nalive + cur, c->d_best);
    AKZ_LAUNCH_CHECK();
    if (P3P)
        hipLaunchKernelGGL(k_p3p_inliers, dim3(1), dim3(1024), 0, s, c->d_a, c->d_w, n, c->d_poses, c->d_best, prm->threshold,
                           c->d_inl, n, c->d_ninl, c->d_best_pose, (unsigned long long*)nullptr);
    else
        hipLaunchKernelGGL(k_rs_inliers, dim3(1), dim3(1024), 0, s, c->d_a, c->d_b, n, c->d_poses, c->d_best, prm->threshold,
                           c->d_inl, n, c->d_ninl, c->d_best_pose, (unsigned long long*)nullptr);
    AKZ_LAUNCH_CHECK();
    uint32_t best[3] = {0, 0, 0}, ninl = 0;
    unsigned long long neval = 0;
    AKZ_HIP(hipMemcpyAsync(best, c->d_best, sizeof(best), hipMemcpyDeviceToHost, s));
    AKZ_HIP(hipMemcpyAsync(&ninl, c->d_ninl, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    AKZ_HIP(hipMemcpyAsync(&neval, c->d_neval, sizeof(neval), hipMemcpyDeviceToHost, s));
    AKZ_HIP(hipMemcpyAsync(best_pose, c->d_best_pose, sizeof(double) * 12, hipMemcpyDeviceToHost, s));
    AKZ_HIP(hipStreamSynchronize(s));
    c->last_hyp = next_h;
    if (stats) {
        stats->poses = next_h * 4;
        stats->survivors = best[2];
        stats->blocks = blocks;
        stats->reserved = 0;
        stats->residuals_evaluated = neval;
        stats->residuals_exhaustive = (uint64_t)next_h * 4 * n;
    }
    *best_id = best[0];
    *n_inliers = ninl;
    if (best[0] == 0xFFFFFFFFu) {
        *n_inliers = 0;
        return AKZ_OK;
    }
    uint32_t ncopy = ninl < cap ? ninl : cap;
    if (ncopy) AKZ_HIP(hipMemcpy(inlier_idx, c->d_inl, sizeof(uint32_t) * ncopy, hipMemcpyDeviceToHost));
    return ninl > cap ? AKZ_E_CAPACITY : AKZ_OK;
}

extern "C" int32_t rs_essential_arrsac(rs_ctx* c, const double* bearings_a, const double* bearings_b, uint32_t n,
                                       const uint32_t* sample_idx, const rs_arrsac_params* prm, double* best_pose,
                                       uint32_t* best_id, uint32_t* inlier_idx, uint32_t cap, uint32_t* n_inliers,
                                       rs_arrsac_stats* stats)
{
    return akz_guard([&]() -> int32_t {
        return arrsac_run<false>(c, bearings_a, bearings_b, n, sample_idx, prm, best_pose, best_id, inlier_idx, cap, n_inliers, stats);
    });
}

// the registration path's consensus (cv-sfm/src/lib.rs:1619-1622; vslam-sandbox/src/main.rs:105-111: Arrsac with 16384
// initialisation hypotheses, 1024 candidates) in the same shape: Lambda Twist hypotheses from 3-match samples
extern "C" int32_t rs_p3p_arrsac(rs_ctx* c, const double* bearings, const double* world, uint32_t n,
                                 const uint32_t* sample_idx, const rs_arrsac_params* prm, double* best_pose,
                                 uint32_t* best_id, uint32_t* inlier_idx, uint32_t cap, uint32_t* n_inliers,
                                 rs_arrsac_stats* stats)
{
    return akz_guard([&]() -> int32_t {
        return arrsac_run<true>(c, bearings, world, n, sample_idx, prm, best_pose, best_id, inlier_idx, cap, n_inliers, stats);
    });
}

// the minimal samples rs_essential_arrsac draws on the device for (seed, n): host restatement for callers and tests
extern "C" int32_t rs_arrsac_samples(uint64_t seed, uint32_t n, uint32_t n_hyp, uint32_t sample_size, uint32_t* sample_idx)
{
    return akz_guard([&]() -> int32_t {
        if (!sample_idx || (sample_size != 8 && sample_size != 3) || n < sample_size) return AKZ_E_INVALID;
        for (uint32_t h = 0; h < n_hyp; ++h) {
            if (sample_size == 8) rs_draw_sample<8>((unsigned long long)seed, h, n, sample_idx + (size_t)h * 8);
            else rs_draw_sample<3>((unsigned long long)seed, h, n, sample_idx + (size_t)h * 3);
        }
        return AKZ_OK;
    });
}

// parity tap: inlier count of every (hypothesis, pose) of the last rs_essential_batch call
extern "C" int32_t rs_debug_counts(rs_ctx* c, uint32_t* counts, uint32_t cap)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !counts) return AKZ_E_INVALID;
        if (cap < c->last_hyp * 4) return AKZ_E_CAPACITY;
        AKZ_HIP(hipSetDevice(c->device));
        AKZ_HIP(hipMemcpy(counts, c->d_counts, sizeof(uint32_t) * 4 * (size_t)c->last_hyp, hipMemcpyDeviceToHost));
        return AKZ_OK;
    });
}

// Consensus::model_inliers(&LambdaTwist::new(), world_matches) with the sampler factored out
// (cv-sfm/src/lib.rs:1619-1622; lambda-twist/tests/consensus.rs:59-61): n_hyp sample triples.
extern "C" int32_t rs_p3p_batch(rs_ctx* c, const double* bearings, const double* world, uint32_t n,
                                const uint32_t* sample_idx, uint32_t n_hyp, double thresh, double* best_pose,
                                uint32_t* best_id, uint32_t* inlier_idx, uint32_t cap, uint32_t* n_inliers)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !bearings || !world || !sample_idx || !best_pose || !best_id || !n_inliers || (cap && !inlier_idx))
            return AKZ_E_INVALID;
        if (n < 3 || n_hyp == 0) return AKZ_E_INVALID;  // LambdaTwist::MIN_SAMPLES (lambda-twist/src/lib.rs:333)
        if (n > c->max_matches || n_hyp > c->max_hyp) return AKZ_E_TOO_LARGE;
        for (size_t i = 0; i < (size_t)n_hyp * 3; ++i)
            if (sample_idx[i] >= n) return AKZ_E_INVALID;
        AKZ_HIP(hipSetDevice(c->device));
        hipStream_t s = c->stream;
        AKZ_HIP(hipMemcpyAsync(c->d_a, bearings, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, s));
        AKZ_HIP(hipMemcpyAsync(c->d_w, world, sizeof(double) * 4 * (size_t)n, hipMemcpyHostToDevice, s));
        AKZ_HIP(hipMemcpyAsync(c->d_samples, sample_idx, sizeof(uint32_t) * 3 * (size_t)n_hyp, hipMemcpyHostToDevice, s));
        AKZ_HIP(hipMemsetAsync(c->d_counts, 0, sizeof(uint32_t) * 4 * (size_t)n_hyp, s));
        hipLaunchKernelGGL(k_p3p_hypotheses, dim3((n_hyp + 63) / 64), dim3(64), 0, s, c->d_a, c->d_w, c->d_samples, n_hyp,
                           c->d_poses, c->d_ok);
        AKZ_LAUNCH_CHECK();
        const uint32_t n_pose = n_hyp * 4;
        for (uint32_t p0 = 0; p0 < n_pose; p0 += 65532) {
            uint32_t np = n_pose - p0 < 65532 ? n_pose - p0 : 65532;
            hipLaunchKernelGGL(k_p3p_score, dim3((n + 255) / 256, np), dim3(256), 0, s, c->d_a, c->d_w, n,
                               c->d_poses + (size_t)p0 * 12, c->d_ok + p0, thresh, c->d_counts + p0);
            AKZ_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(k_rs_best, dim3(1), dim3(1024), 0, s, c->d_counts, c->d_ok, n_pose, c->d_best);
        AKZ_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_p3p_inliers, dim3(1), dim3(1024), 0, s, c->d_a, c->d_w, n, c->d_poses, c->d_best, thresh,
                           c->d_inl, n, c->d_ninl, c->d_best_pose, (unsigned long long*)nullptr);
        AKZ_LAUNCH_CHECK();
        uint32_t best[2] = {0, 0}, ninl = 0;
        AKZ_HIP(hipMemcpyAsync(best, c->d_best, sizeof(best), hipMemcpyDeviceToHost, s));
        AKZ_HIP(hipMemcpyAsync(&ninl, c->d_ninl, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        AKZ_HIP(hipMemcpyAsync(best_pose, c->d_best_pose, sizeof(double) * 12, hipMemcpyDeviceToHost, s));
        AKZ_HIP(hipStreamSynchronize(s));
        c->last_hyp = n_hyp;
        *best_id = best[0];
        *n_inliers = ninl;
        if (best[0] == 0xFFFFFFFFu) {
            *n_inliers = 0;
            return AKZ_OK;
        }
        uint32_t ncopy = ninl < cap ? ninl : cap;
        if (ncopy) AKZ_HIP(hipMemcpy(inlier_idx, c->d_inl, sizeof(uint32_t) * ncopy, hipMemcpyDeviceToHost));
        return ninl > cap ? AKZ_E_CAPACITY : AKZ_OK;
    });
}
