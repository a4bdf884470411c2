/* match_oracle.c — CPU restatement of the reference's brute-force Hamming matcher.
 *
 * TEST INFRASTRUCTURE ONLY (see akaze_oracle.c header): the checker for cv_amd's HIP matcher.
 *
 * The arithmetic lives in two un-vendored crates (no Cargo.lock in the reference; versions are the
 * semver ranges of akaze/Cargo.toml:26-27): `bitarray` 0.9 (Hamming distance of BitArray<64> =
 * popcount of the XOR over all 64 bytes) and `space` 0.17 (LinearKnn::knn: take the first `num`
 * items, sort them by distance, then insert every later item at partition_point(d <= new) and pop
 * the tail — so among equal distances the LOWEST index wins, for the 1st and the 2nd neighbour).
 * Parity is anchored on the reference's call sites:
 *   akaze/tests/estimate_pose.rs:78-97  match_descriptors (Lowe ratio 0.5, f32)   -> pin: 11 matches
 *   tutorial-code/chapter5-geometric-verification/src/main.rs:154-200  matching / symmetric_matching
 *   cv-sfm/src/lib.rs:3097-3133  matching / symmetric_matching with better_by (<=) and the <2 guard
 * Tie-break parity beyond the match COUNT is unpinned by the reference.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/akz.h"

static inline uint32_t hamming64(const uint8_t* a, const uint8_t* b)
{
    uint32_t d = 0;
    for (int i = 0; i < 64; i += 8) {
        uint64_t x, y;
        memcpy(&x, a + i, 8);
        memcpy(&y, b + i, 8);
        d += (uint32_t)__builtin_popcountll(x ^ y);
    }
    return d;
}

uint32_t orc_hamming(const akz_descriptor* a, const akz_descriptor* b) { return hamming64(a->bytes, b->bytes); }

/* LinearKnn{metric: Hamming, iter: t}.knn(q, 2) for each query. out[2*i], out[2*i+1]. nt >= 2. */
int orc_knn2(const akz_descriptor* q, uint32_t nq, const akz_descriptor* t, uint32_t nt, akz_neighbor* out)
{
    if (nt < 2) return -1;
    for (uint32_t i = 0; i < nq; ++i) {
        akz_neighbor n0 = {0, hamming64(q[i].bytes, t[0].bytes)};
        akz_neighbor n1 = {1, hamming64(q[i].bytes, t[1].bytes)};
        if (n1.distance < n0.distance) { /* sort_unstable_by_key on two items */
            akz_neighbor tmp = n0;
            n0 = n1;
            n1 = tmp;
        }
        for (uint32_t j = 2; j < nt; ++j) {
            uint32_t d = hamming64(q[i].bytes, t[j].bytes);
            /* partition_point(|n| n.distance <= d): insert after every neighbour with distance <= d */
            if (n0.distance <= d) {
                if (n1.distance <= d) continue; /* position == num: not inserted */
                n1.index = j;
                n1.distance = d;
            } else {
                n1 = n0;
                n0.index = j;
                n0.distance = d;
            }
        }
        out[2 * i] = n0;
        out[2 * i + 1] = n1;
    }
    return 0;
}

/* LinearKnn::knn(q, k), any k >= 1 (space 0.17): the first min(k, nt) items sorted by distance (stable),
 * then every later item inserted at partition_point(d <= new) and the tail popped.  out[nq][k]; slots
 * past min(k, nt) are filled with {UINT32_MAX, UINT32_MAX} (the reference's Vec is simply shorter). */
int orc_knn(const akz_descriptor* q, uint32_t nq, const akz_descriptor* t, uint32_t nt, uint32_t k, akz_neighbor* out)
{
    if (k == 0) return -1;
    for (uint32_t i = 0; i < nq; ++i) {
        akz_neighbor* o = out + (size_t)i * k;
        uint32_t n = 0;
        for (uint32_t j = 0; j < nt; ++j) {
            uint32_t d = hamming64(q[i].bytes, t[j].bytes);
            /* position after every kept neighbour with distance <= d (for the first k items this is the
             * stable insertion sort the initial sort amounts to) */
            uint32_t pos = 0;
            while (pos < n && o[pos].distance <= d) ++pos;
            if (pos >= k) continue;
            uint32_t last = n < k ? n : k - 1;
            for (uint32_t m = last; m > pos; --m) o[m] = o[m - 1];
            o[pos].index = j;
            o[pos].distance = d;
            if (n < k) ++n;
        }
        for (uint32_t m = n; m < k; ++m) o[m].index = o[m].distance = UINT32_MAX;
    }
    return 0;
}

static int accept(int rule, uint32_t d0, uint32_t d1, uint32_t pu, float pf)
{
    switch (rule) {
    case HM_RULE_BETTER_BY_STRICT: return d0 + pu < d1;            /* ch5 main.rs:162 */
    case HM_RULE_BETTER_BY: return d0 + pu <= d1;                  /* cv-sfm lib.rs:3107 */
    default: return (float)d0 < (float)d1 * pf;                    /* estimate_pose.rs:92 */
    }
}

/* one-directional matching(): fwd[i] = index in b or UINT32_MAX */
static int matching(const akz_descriptor* a, uint32_t na, const akz_descriptor* b, uint32_t nb, int rule,
                    uint32_t pu, float pf, uint32_t* fwd)
{
    akz_neighbor* nn = (akz_neighbor*)malloc(sizeof(akz_neighbor) * 2 * (na ? na : 1));
    if (orc_knn2(a, na, b, nb, nn) != 0) {
        free(nn);
        return -1;
    }
    for (uint32_t i = 0; i < na; ++i)
        fwd[i] = accept(rule, nn[2 * i].distance, nn[2 * i + 1].distance, pu, pf) ? nn[2 * i].index : UINT32_MAX;
    free(nn);
    return 0;
}

/* returns number of pairs, or -1 when a side has < 2 descriptors under a rule whose reference
 * implementation would panic (LinearKnn on < 2 items then indexing knn[1]). */
int orc_match(const akz_descriptor* a, uint32_t na, const akz_descriptor* b, uint32_t nb, int rule,
              uint32_t pu, float pf, int symmetric, uint32_t* pairs, uint32_t cap)
{
    if (na < 2 || nb < 2) {
        if (rule == HM_RULE_BETTER_BY) return 0; /* cv-sfm/src/lib.rs:3099-3101 */
        if (nb < 2 || (symmetric && na < 2)) return -1;
    }
    uint32_t* fwd = (uint32_t*)malloc(sizeof(uint32_t) * (na ? na : 1));
    uint32_t* rev = (uint32_t*)malloc(sizeof(uint32_t) * (nb ? nb : 1));
    int n = 0;
    if (matching(a, na, b, nb, rule, pu, pf, fwd) != 0) n = -1;
    if (n == 0 && symmetric && matching(b, nb, a, na, rule, pu, pf, rev) != 0) n = -1;
    if (n == 0) {
        for (uint32_t i = 0; i < na; ++i) {
            if (fwd[i] == UINT32_MAX) continue;
            if (symmetric && rev[fwd[i]] != i) continue;
            if ((uint32_t)n < cap) {
                pairs[2 * n] = i;
                pairs[2 * n + 1] = fwd[i];
            }
            n++;
        }
    }
    free(fwd);
    free(rev);
    return n;
}

/* Best-of-views landmark selection, cv-sfm/src/lib.rs:1489-1532: per feature the minimum distance of every distinct
 * landmark over the n_views x k neighbours (the HashMap of :1489-1501), the three best by (distance, landmark key)
 * — the reference's order among equal distances is its HashMap's iteration order, i.e. unspecified: parity unpinned
 * for ties — and the decision of :1516-1532.  knn [n_views][cap][k]; landmarks [blocks][cap]; best [nq][3][2]. */
void orc_best_of_views(const akz_neighbor* knn, uint32_t nq, uint32_t cap, uint32_t n_views, uint32_t k,
                       const uint32_t* landmarks, const uint32_t* view_idx, const uint32_t* nviews, uint32_t better_by,
                       uint32_t* best, uint32_t* decision)
{
    for (uint32_t i = 0; i < nq; ++i) {
        /* the dedup map as a small array (at most n_views * k entries) */
        uint32_t ml[192], md[192], nm = 0;
        for (uint32_t v = 0; v < n_views; ++v) {
            uint32_t blk = view_idx[v], nt = nviews[blk] < cap ? nviews[blk] : cap;
            for (uint32_t j = 0; j < k && j < nt; ++j) {
                akz_neighbor nb = knn[((size_t)v * cap + i) * k + j];
                uint32_t l = landmarks[(size_t)blk * cap + nb.index], found = nm;
                for (uint32_t q = 0; q < nm; ++q)
                    if (ml[q] == l) found = q;
                if (found == nm) { ml[nm] = l; md[nm] = nb.distance; nm++; }
                else if (md[found] > nb.distance) md[found] = nb.distance;
            }
        }
        uint32_t bl[3] = {UINT32_MAX, UINT32_MAX, UINT32_MAX}, bd[3] = {UINT32_MAX, UINT32_MAX, UINT32_MAX};
        for (uint32_t q = 0; q < nm; ++q) {                      /* selection of the three smallest (distance, key) */
            int pos = 3;
            for (int r = 2; r >= 0; --r)
                if (md[q] < bd[r] || (md[q] == bd[r] && ml[q] < bl[r])) pos = r;
            if (pos == 3) continue;
            for (int r = 2; r > pos; --r) { bl[r] = bl[r - 1]; bd[r] = bd[r - 1]; }
            bl[pos] = ml[q]; bd[pos] = md[q];
        }
        for (int r = 0; r < 3; ++r) { best[((size_t)i * 3 + r) * 2] = bl[r]; best[((size_t)i * 3 + r) * 2 + 1] = bd[r]; }
        uint32_t dec = 0;
        if (nm >= 3) {
            if (bd[0] + better_by <= bd[1]) dec = 1;
            else if (bd[1] + better_by <= bd[2]) dec = 2;
        }
        decision[i] = dec;
    }
}

/* The FeatureWorldMatch list of one frame, cv-sfm/src/lib.rs:1516-1532 (decision 1 -> ([best0], feature); decision 2 with
 * the caller's are_landmarks_sharing_view verdict merge_ok[i] != 0 -> ([best0, best1], feature)), :1549-1563 (landmark_counts
 * over every landmark of every original match; a match survives iff each of its landmarks was counted once) and :1583-1604
 * (triangulation None -> dropped; here: landmark key >= n_world, or a world row with w < 0; the merged triangulation of
 * feature i is row merged_base + i).  best [nq][3][2] {landmark, distance}, decision [nq], merge_ok [nq] or NULL;
 * pairs [..][2] {feature, world row} in ascending feature order; returns their number. */
static uint32_t orc_lm_kind(const uint32_t* best, const uint32_t* decision, const uint8_t* merge_ok, uint32_t i)
{
    if (best[(size_t)i * 6] == 0xFFFFFFFFu) return 0;
    if (decision[i] == 1u) return 1;
    if (decision[i] == 2u && merge_ok && merge_ok[i] && best[(size_t)i * 6 + 2] != 0xFFFFFFFFu) return 2;
    return 0;
}
static uint32_t orc_lm_claims(const uint32_t* best, const uint32_t* decision, const uint8_t* merge_ok, uint32_t nq, uint32_t lm)
{
    uint32_t claims = 0;
    for (uint32_t j = 0; j < nq; ++j) {
        const uint32_t k = orc_lm_kind(best, decision, merge_ok, j);
        if (k >= 1 && best[(size_t)j * 6] == lm) claims++;
        if (k == 2 && best[(size_t)j * 6 + 2] == lm) claims++;
    }
    return claims;
}
uint32_t orc_landmark_matches(const uint32_t* best, const uint32_t* decision, const uint8_t* merge_ok, uint32_t nq,
                              const double* world, uint32_t n_world, uint32_t merged_base, uint32_t* pairs)
{
    uint32_t n = 0;
    for (uint32_t i = 0; i < nq; ++i) {
        const uint32_t k = orc_lm_kind(best, decision, merge_ok, i);
        if (k == 0) continue;
        if (orc_lm_claims(best, decision, merge_ok, nq, best[(size_t)i * 6]) != 1) continue;
        uint32_t row = best[(size_t)i * 6];
        if (k == 2) {
            if (orc_lm_claims(best, decision, merge_ok, nq, best[(size_t)i * 6 + 2]) != 1) continue;
            row = merged_base + i;
        } else if (row >= n_world) continue;
        if (!(world[(size_t)4 * row + 3] >= 0.0)) continue;
        pairs[2 * n] = i;
        pairs[2 * n + 1] = row;
        n++;
    }
    return n;
}
/* The same list in the order the reference hands to the consensus (cv-sfm/src/lib.rs:1549-1604): original_matches in feature
 * order (:1489-1542 walks the features in order), `retain` of the matches whose landmarks were all counted once (:1555-1559),
 * then `sort_by_key(Reverse(sum of landmark(..).observations.len()))` (:1561-1574) — a STABLE sort: equal sums keep the
 * feature order —, then the filter_map that drops matches without a robust triangulation (:1583-1604).  obs[l] = number of
 * observations of landmark key l (l < n_world; a key beyond the table counts 0). */
uint32_t orc_landmark_matches_ordered(const uint32_t* best, const uint32_t* decision, const uint8_t* merge_ok, const uint32_t* obs,
                                      uint32_t nq, const double* world, uint32_t n_world, uint32_t merged_base, uint32_t* pairs)
{
    uint32_t* feat = (uint32_t*)malloc(sizeof(uint32_t) * (nq ? nq : 1));
    uint64_t* key = (uint64_t*)malloc(sizeof(uint64_t) * (nq ? nq : 1));
    uint32_t m = 0;
    for (uint32_t i = 0; i < nq; ++i) {                       /* retain */
        const uint32_t k = orc_lm_kind(best, decision, merge_ok, i);
        if (k == 0) continue;
        const uint32_t l0 = best[(size_t)i * 6], l1 = best[(size_t)i * 6 + 2];
        if (orc_lm_claims(best, decision, merge_ok, nq, l0) != 1) continue;
        if (k == 2 && orc_lm_claims(best, decision, merge_ok, nq, l1) != 1) continue;
        uint64_t sum = l0 < n_world ? obs[l0] : 0;
        if (k == 2) sum += l1 < n_world ? obs[l1] : 0;
        feat[m] = i;
        key[m] = sum;
        m++;
    }
    for (uint32_t a = 1; a < m; ++a) {                        /* stable insertion sort, descending sums */
        const uint32_t f = feat[a];
        const uint64_t kk = key[a];
        uint32_t b = a;
        while (b > 0 && key[b - 1] < kk) {
            feat[b] = feat[b - 1];
            key[b] = key[b - 1];
            b--;
        }
        feat[b] = f;
        key[b] = kk;
    }
    uint32_t n = 0;
    for (uint32_t a = 0; a < m; ++a) {                        /* filter_map: a robust world point */
        const uint32_t i = feat[a];
        const uint32_t k = orc_lm_kind(best, decision, merge_ok, i);
        uint32_t row = best[(size_t)i * 6];
        if (k == 2) row = merged_base + i;
        else if (row >= n_world) continue;
        if (!(world[(size_t)4 * row + 3] >= 0.0)) continue;
        pairs[2 * n] = i;
        pairs[2 * n + 1] = row;
        n++;
    }
    free(feat);
    free(key);
    return n;
}
uint32_t orc_landmark_pairs(const uint32_t* best, const uint32_t* decision, uint32_t nq, const double* world, uint32_t n_world,
                            uint32_t* pairs)
{
    return orc_landmark_matches(best, decision, NULL, nq, world, n_world, 0, pairs);
}
