/* arrsac_oracle.c — CPU statement of the ARRSAC-shaped consensus of include/akz.h (rs_essential_arrsac /
 * rs_p3p_arrsac): sampler, breadth-first block scoring, retirement rules, inlier-guided re-sampling.
 *
 * TEST INFRASTRUCTURE ONLY (see akaze_oracle.c header).
 *
 * What the reference runs here is arrsac::Arrsac::model_inliers (arrsac 0.10, a crates.io dependency that is NOT
 * vendored in the reference checkout; call sites vslam-sandbox/src/main.rs:105-117, cv-sfm/src/lib.rs:1394-1412,
 * 1619-1622, akaze/tests/estimate_pose.rs:63-67).  Its published algorithm (Raguram, Frahm, Pollefeys: "A Comparative
 * Analysis of RANSAC Techniques Leading to Adaptive Real-Time Random Sample Consensus", ECCV 2008) scores an initial
 * hypothesis set breadth-first over blocks of data, keeps at most M candidates and halves that number block by
 * block (the preemption function f(i) = floor(M 2^-floor(i/B))), rejects hypotheses by Wald's SPRT, and generates new
 * hypotheses from the inliers of the current best one.  The crate's exact random stream, block bookkeeping and tie
 * handling cannot be read here, so this file is the SPECIFICATION of this repository's version of that shape and
 * the device path is held to it bit for bit; against the reference itself parity is unpinned beyond the count pin
 * inliers.len() == 11 (akaze/tests/estimate_pose.rs:75).  The per-hypothesis geometry (eight-point, poses,
 * residuals, Lambda Twist) is the restatement of ransac_oracle.c / p3p_oracle.c.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/akz.h"

int orc_eight_point(const double* a8, const double* b8, double eps, int iters, double* E);
int orc_essential_poses(const double* E, double eps, int iters, double* poses);
double orc_residual(const double* pose, const double* a, const double* b, double eps, int iters);
int orc_p3p_poses(const double* bearings3, const double* world3, double* poses);
double orc_w2c_residual(const double* pose, const double* bearing, const double* world);

/* ---- sampler: one xoshiro256++ stream per hypothesis, seeded by a splitmix64 chain ---- */
typedef struct { uint64_t s[4]; } xo256;
static uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static uint64_t splitmix(uint64_t* x)
{
    uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint64_t xo_next(xo256* g)
{
    const uint64_t r = rotl64(g->s[0] + g->s[3], 23) + g->s[0];
    const uint64_t t = g->s[1] << 17;
    g->s[2] ^= g->s[0];
    g->s[3] ^= g->s[1];
    g->s[1] ^= g->s[2];
    g->s[0] ^= g->s[3];
    g->s[2] ^= t;
    g->s[3] = rotl64(g->s[3], 45);
    return r;
}
/* K distinct indices below n for hypothesis h: high 32 bits x n >> 32, duplicates redrawn */
void orc_arrsac_draw(uint64_t seed, uint32_t h, uint32_t n, uint32_t K, uint32_t* out)
{
    uint64_t x = seed + 0xD1B54A32D192ED03ull * (uint64_t)(h + 1u);
    xo256 g;
    for (int i = 0; i < 4; ++i) g.s[i] = splitmix(&x);
    for (uint32_t i = 0; i < K; ++i) {
        uint32_t v;
        int dup;
        do {
            v = (uint32_t)(((xo_next(&g) >> 32) * (uint64_t)n) >> 32);
            dup = 0;
            for (uint32_t j = 0; j < i; ++j) dup = dup || out[j] == v;
        } while (dup);
        out[i] = v;
    }
}

typedef struct {
    int p3p;
    const double *a, *b; /* bearings a + (bearings b | world points [4]) */
    double thresh;
} scene_t;

static int inlier(const scene_t* sc, const double* pose, uint32_t m)
{
    if (sc->p3p) return orc_w2c_residual(pose, sc->a + 3 * (size_t)m, sc->b + 4 * (size_t)m) < sc->thresh;
    return orc_residual(pose, sc->a + 3 * (size_t)m, sc->b + 3 * (size_t)m, 1e-12, 1024) < sc->thresh;
}

/* poses of one minimal sample -> out[48]; returns a validity mask over the four pose slots */
static unsigned make_poses(const scene_t* sc, const uint32_t* sample, double* out)
{
    if (sc->p3p) {
        double b3[9], w3[12];
        for (int i = 0; i < 3; ++i) {
            memcpy(b3 + 3 * i, sc->a + 3 * (size_t)sample[i], 24);
            memcpy(w3 + 4 * i, sc->b + 4 * (size_t)sample[i], 32);
        }
        int np = orc_p3p_poses(b3, w3, out);
        return np >= 4 ? 15u : ((1u << np) - 1u);
    }
    double a8[24], b8[24], E[9];
    for (int i = 0; i < 8; ++i) {
        memcpy(a8 + 3 * i, sc->a + 3 * (size_t)sample[i], 24);
        memcpy(b8 + 3 * i, sc->b + 3 * (size_t)sample[i], 24);
    }
    int ok = orc_eight_point(a8, b8, 1e-12, 1000, E) == 0 && orc_essential_poses(E, 1e-12, 1000, out) == 0;
    return ok ? 15u : 0u;
}

/* The whole procedure.  The SPRT decisions use log(k) of integer counts from a table filled with the host libm,
 * exactly as the library fills the table it uploads (no second transcendental implementation is involved).
 * Returns 0, or -1 when no hypothesis produced a model.  stats[0] = residuals evaluated (lo), [1] = hi,
 * [2] = survivors, [3] = blocks, [4] = hypotheses generated in all. */
int orc_arrsac_ordered(int p3p, const double* a, const double* b, uint32_t n, const uint32_t* sample_idx /* may be NULL */,
                       const uint32_t* order /* scoring order: position -> match; NULL = identity */,
                       const rs_arrsac_params* prm, double* best_pose, uint32_t* best_id,
                       uint32_t* inlier_idx, uint32_t* n_inliers, uint32_t* stats)
{
    const uint32_t K = p3p ? 3u : 8u;
    double* log_table = (double*)malloc(sizeof(double) * ((size_t)n + 1));
    for (uint32_t i = 0; i <= n; ++i) log_table[i] = log((double)i);
    scene_t sc = {p3p, a, b, prm->threshold};
    const uint32_t E = prm->estimations_per_block;
    const uint32_t n_blocks_max = (n + prm->block_size - 1) / prm->block_size;
    const uint32_t max_h = prm->n_hypotheses + E * n_blocks_max;
    double* poses = (double*)malloc(sizeof(double) * 48 * (size_t)max_h);
    uint32_t* counts = (uint32_t*)calloc((size_t)max_h * 4, sizeof(uint32_t));
    uint32_t* alive = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)max_h * 4);
    uint32_t* keep = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)max_h * 4);
    uint32_t* L = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n ? n : 1));
    uint32_t n_alive = 0, next_h = prm->n_hypotheses;
    uint64_t neval = 0;
    for (uint32_t h = 0; h < prm->n_hypotheses; ++h) {
        uint32_t s[8];
        if (sample_idx) memcpy(s, sample_idx + (size_t)h * K, sizeof(uint32_t) * K);
        else orc_arrsac_draw(prm->seed, h, n, K, s);
        unsigned ok = make_poses(&sc, s, poses + 48 * (size_t)h);
        for (uint32_t p = 0; p < 4; ++p)
            if (ok >> p & 1u) alive[n_alive++] = h * 4 + p;
    }
    const int prune = (prm->flags & (RS_PRUNE_BOUND | RS_PRUNE_SPRT)) != 0 || prm->max_candidates != 0 || E != 0;
    uint32_t seen = 0, blocks = 0;
    while (seen < n) {
        const uint32_t bs = prune ? prm->block_size : n;
        const uint32_t m_hi = seen + bs < n ? seen + bs : n;
        for (uint32_t i = 0; i < n_alive; ++i) {
            const double* pose = poses + 12 * (size_t)alive[i];
            for (uint32_t j = seen; j < m_hi; ++j) counts[alive[i]] += (uint32_t)inlier(&sc, pose, order ? order[j] : j);
            neval += m_hi - seen;
        }
        seen = m_hi;
        ++blocks;
        if (!prune || seen >= n) continue;
        /* ---- retirement: best count so far, bound, SPRT, candidate cap (halved block by block when asked) ---- */
        uint32_t best = 0;
        for (uint32_t i = 0; i < n_alive; ++i) best = counts[alive[i]] > best ? counts[alive[i]] : best;
        const uint32_t left = n - seen;
        uint32_t cap = 0;
        if (prm->max_candidates && blocks >= prm->init_blocks) {
            cap = prm->max_candidates;
            if (prm->flags & RS_PRUNE_HALVE) {
                uint32_t sh = blocks - prm->init_blocks;
                cap = sh >= 31 ? 0u : cap >> sh;
                if (cap == 0) cap = 1;
            }
        }
        double l_in = 0.0, l_out = 0.0;
        const int sprt = (prm->flags & RS_PRUNE_SPRT) && best > 0 && best < seen;
        if (sprt) {
            l_in = log(prm->sprt_delta) - (log_table[best] - log_table[seen]);
            l_out = log(1.0 - prm->sprt_delta) - (log_table[seen - best] - log_table[seen]);
        }
        const double log_ratio = (prm->flags & RS_PRUNE_SPRT) ? log(prm->sprt_ratio) : 0.0;
        /* The cap ranks the poses by their distance to the best count, best - count, through a 2048-bin histogram over the
         * population BEFORE this block's retirements (distances of 2047 and more share the last bin): poses closer than T
         * all stay, those at T are admitted in ascending pose id while the budget lasts. */
        uint32_t T = 0xFFFFFFFFu, budget = 0xFFFFFFFFu;
        const int capped = cap && n_alive > cap;
        if (capped) {
            static uint32_t hist[2048];
            memset(hist, 0, sizeof(hist));
            for (uint32_t i = 0; i < n_alive; ++i) {
                uint32_t d = best - counts[alive[i]];
                hist[d < 2047u ? d : 2047u]++;
            }
            uint32_t acc = 0, t = 0;
            for (; t < 2048u; ++t) {
                if (acc + hist[t] >= cap) break;
                acc += hist[t];
            }
            T = t < 2048u ? t : 2047u;
            budget = cap - acc;
        }
        uint32_t nk = 0, ties = 0;
        for (uint32_t i = 0; i < n_alive; ++i) {
            const uint32_t pid = alive[i], c = counts[pid];
            int k = c + left >= best;
            if (k && sprt && l_out > 0.0) k = (double)c * l_in + (double)(seen - c) * l_out <= log_ratio || c == best;
            const uint32_t d = best - c, dd = d < 2047u ? d : 2047u;
            if (k && capped) {
                if (dd > T) k = 0;
                else if (dd == T) {
                    if (ties >= budget) k = 0;
                    ties++;
                }
            }
            if (k) keep[nk++] = pid;
        }
        memcpy(alive, keep, sizeof(uint32_t) * nk);
        n_alive = nk;
        /* a single survivor cannot be overtaken when nothing is re-sampled: the halving scheme ends here */
        if (cap == 1 && E == 0 && (prm->flags & RS_PRUNE_HALVE)) break;
        /* ---- inlier-guided re-sampling: E new hypotheses from the inliers (among the matches seen) of the best pose ---- */
        if (E && blocks >= prm->init_blocks && !n_alive) next_h += E;   /* (the round's slots stay unused) */
        if (E && blocks >= prm->init_blocks && n_alive) {
            uint32_t bpid = alive[0], bc = counts[alive[0]];
            for (uint32_t i = 1; i < n_alive; ++i)
                if (counts[alive[i]] > bc) {
                    bc = counts[alive[i]];
                    bpid = alive[i];
                }
            uint32_t nL = 0;
            for (uint32_t j = 0; j < seen; ++j) {
                const uint32_t m = order ? order[j] : j;
                if (inlier(&sc, poses + 12 * (size_t)bpid, m)) L[nL++] = m;
            }
            neval += seen;
            if (nL >= K) {
                for (uint32_t e = 0; e < E; ++e) {
                    const uint32_t h = next_h + e;
                    uint32_t s[8];
                    orc_arrsac_draw(prm->seed ^ 0xA5A5A5A55A5A5A5Aull, h, nL, K, s);
                    for (uint32_t i = 0; i < K; ++i) s[i] = L[s[i]];
                    unsigned ok = make_poses(&sc, s, poses + 48 * (size_t)h);
                    for (uint32_t p = 0; p < 4; ++p) {
                        if (!(ok >> p & 1u)) continue;
                        const uint32_t pid = h * 4 + p;
                        for (uint32_t j = 0; j < seen; ++j)
                            counts[pid] += (uint32_t)inlier(&sc, poses + 12 * (size_t)pid, order ? order[j] : j);
                        neval += seen;
                        alive[n_alive++] = pid;
                    }
                }
            }
            next_h += E;
        }
    }
    int rc = -1;
    if (n_alive) {
        uint32_t bpid = alive[0], bc = counts[alive[0]];
        for (uint32_t i = 1; i < n_alive; ++i)
            if (counts[alive[i]] > bc) {
                bc = counts[alive[i]];
                bpid = alive[i];
            }
        memcpy(best_pose, poses + 12 * (size_t)bpid, sizeof(double) * 12);
        *best_id = bpid;
        uint32_t k = 0;
        for (uint32_t m = 0; m < n; ++m)
            if (inlier(&sc, best_pose, m)) inlier_idx[k++] = m;
        *n_inliers = k;
        rc = 0;
    }
    if (stats) {
        stats[0] = (uint32_t)neval;
        stats[1] = (uint32_t)(neval >> 32);
        stats[2] = n_alive;
        stats[3] = blocks;
        stats[4] = next_h;
    }
    free(log_table);
    free(poses);
    free(counts);
    free(alive);
    free(keep);
    free(L);
    return rc;
}

int orc_arrsac(int p3p, const double* a, const double* b, uint32_t n, const uint32_t* sample_idx /* may be NULL */,
               const rs_arrsac_params* prm, double* best_pose, uint32_t* best_id,
               uint32_t* inlier_idx, uint32_t* n_inliers, uint32_t* stats)
{
    return orc_arrsac_ordered(p3p, a, b, n, sample_idx, NULL, prm, best_pose, best_id, inlier_idx, n_inliers, stats);
}

/* ---- the micro-batch entry (rs_essential_arrsac_batch_device), one scene of it ----
 * What cv-sfm does with a frame pair's matches before and around the consensus (cv-sfm/src/lib.rs:1385-1412): shuffle,
 * map every [a, b] to the calibrated bearing pair (match_ix_kps; cv-pinhole/src/lib.rs:108-117), run model_inliers.
 * The shuffle is the reference's caller-side rng (unpinned); here: position j holds the match with the j-th smallest
 * 32-bit key splitmix64((seed_s ^ 0x5851F42D4C957F2D) + 0xD1342543DE82EF95 j) >> 32, equal keys in index order.
 * Scene s of a call draws from seed_s = seed + 0x9E3779B97F4A7C15 s. */
void orc_calibrate(const double* intr, int use_k1, double k1, const akz_keypoint* kps, uint32_t n, double* out);

uint64_t orc_scene_seed(uint64_t seed, uint32_t scene) { return seed + 0x9E3779B97F4A7C15ull * (uint64_t)scene; }

uint32_t orc_shuffle_key(uint64_t scene_seed, uint32_t j)
{
    uint64_t x = (scene_seed ^ 0x5851F42D4C957F2Dull) + 0xD1342543DE82EF95ull * (uint64_t)j;
    return (uint32_t)(splitmix(&x) >> 32);
}

void orc_shuffle_order(uint64_t scene_seed, uint32_t n, uint32_t* order)
{
    /* stable ascending sort by key: insertion into a merge would do; n <= 8192, so a plain stable merge sort */
    uint32_t* key = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint32_t* tmp = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
    for (uint32_t j = 0; j < n; ++j) {
        key[j] = orc_shuffle_key(scene_seed, j);
        order[j] = j;
    }
    for (uint32_t w = 1; w < n; w *= 2) {
        for (uint32_t lo = 0; lo < n; lo += 2 * w) {
            uint32_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            uint32_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) tmp[k++] = key[order[j]] < key[order[i]] ? order[j++] : order[i++];
            while (i < mid) tmp[k++] = order[i++];
            while (j < hi) tmp[k++] = order[j++];
        }
        memcpy(order, tmp, sizeof(uint32_t) * n);
    }
    free(key);
    free(tmp);
}

/* cam = {fx, fy, cx, cy, skew, k1}, use_k1 flag separately.  pairs [n][2] index kps_a / kps_b.  bearings_a / _b
 * (optional, [n][3]) and order (optional, [n]) receive what the consensus ran on.  Returns 0, -1 (no model). */
int orc_arrsac_pairs(const akz_keypoint* kps_a, const akz_keypoint* kps_b, const uint32_t* pairs, uint32_t n,
                     const double* cam_a, int use_k1_a, const double* cam_b, int use_k1_b, uint32_t scene, int shuffle,
                     const rs_arrsac_params* prm, double* best_pose, uint32_t* best_id, uint32_t* inlier_idx,
                     uint32_t* n_inliers, uint32_t* stats, double* bearings_a, double* bearings_b, uint32_t* order_out)
{
    *n_inliers = 0;
    *best_id = 0xFFFFFFFFu;
    double* a = (double*)malloc(sizeof(double) * 3 * (n ? n : 1));
    double* b = (double*)malloc(sizeof(double) * 3 * (n ? n : 1));
    uint32_t* order = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
    for (uint32_t j = 0; j < n; ++j) {
        orc_calibrate(cam_a, use_k1_a, cam_a[5], kps_a + pairs[2 * j], 1, a + 3 * j);
        orc_calibrate(cam_b, use_k1_b, cam_b[5], kps_b + pairs[2 * j + 1], 1, b + 3 * j);
    }
    rs_arrsac_params p = *prm;
    p.seed = orc_scene_seed(prm->seed, scene);
    if (shuffle) orc_shuffle_order(p.seed, n, order);
    /* fewer matches than a minimal sample: Consensus::model_inliers has nothing to estimate from (None) */
    int rc = n < 8 ? -1
                   : orc_arrsac_ordered(0, a, b, n, NULL, shuffle ? order : NULL, &p, best_pose, best_id, inlier_idx, n_inliers, stats);
    if (bearings_a) memcpy(bearings_a, a, sizeof(double) * 3 * n);
    if (bearings_b) memcpy(bearings_b, b, sizeof(double) * 3 * n);
    if (order_out && shuffle) memcpy(order_out, order, sizeof(uint32_t) * n);
    free(a);
    free(b);
    free(order);
    return rc;
}

/* One scene of rs_p3p_arrsac_batch_device (cv-sfm/src/lib.rs:1571-1622): pairs [n][2] = {feature index into kps, index
 * into world ([..][4] homogeneous points)}; bearing = calibrate(keypoint).  bearings ([n][3]), world_out ([n][4]) and
 * order ([n]) optionally receive what the consensus ran on.  Returns 0, -1 (no model). */
int orc_p3p_arrsac_pairs(const akz_keypoint* kps, const uint32_t* pairs, uint32_t n, const double* cam, int use_k1,
                         const double* world, uint32_t scene, int shuffle, const rs_arrsac_params* prm, double* best_pose,
                         uint32_t* best_id, uint32_t* inlier_idx, uint32_t* n_inliers, uint32_t* stats, double* bearings,
                         double* world_out, uint32_t* order_out)
{
    *n_inliers = 0;
    *best_id = 0xFFFFFFFFu;
    double* a = (double*)malloc(sizeof(double) * 3 * (n ? n : 1));
    double* b = (double*)malloc(sizeof(double) * 4 * (n ? n : 1));
    uint32_t* order = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
    for (uint32_t j = 0; j < n; ++j) {
        orc_calibrate(cam, use_k1, cam[5], kps + pairs[2 * j], 1, a + 3 * j);
        memcpy(b + 4 * j, world + (size_t)4 * pairs[2 * j + 1], sizeof(double) * 4);
    }
    rs_arrsac_params p = *prm;
    p.seed = orc_scene_seed(prm->seed, scene);
    if (shuffle) orc_shuffle_order(p.seed, n, order);
    /* fewer matches than a minimal sample (LambdaTwist::MIN_SAMPLES = 3): None */
    int rc = n < 3 ? -1
                   : orc_arrsac_ordered(1, a, b, n, NULL, shuffle ? order : NULL, &p, best_pose, best_id, inlier_idx, n_inliers, stats);
    if (bearings) memcpy(bearings, a, sizeof(double) * 3 * n);
    if (world_out) memcpy(world_out, b, sizeof(double) * 4 * n);
    if (order_out && shuffle) memcpy(order_out, order, sizeof(uint32_t) * n);
    free(a);
    free(b);
    free(order);
    return rc;
}

/* The refusal rule of the batched device entry points (k_rsb_prepare): a scene whose pair list names a feature outside its
 * keypoint block (>= limit_a) or a second index outside the other block / the world table (>= limit_b) ends with "no model";
 * returns 1 when every entry of the list is in range (the scene is then one of the two functions above). */
int orc_pairs_in_range(const uint32_t* pairs, uint32_t n, uint32_t limit_a, uint32_t limit_b)
{
    for (uint32_t j = 0; j < n; ++j)
        if (pairs[2 * j] >= limit_a || pairs[2 * j + 1] >= limit_b) return 0;
    return 1;
}
