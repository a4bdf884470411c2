/* lsh_oracle.c — CPU restatement of cv-sfm's frame-level place recognition.
 *
 * TEST INFRASTRUCTURE ONLY (see akaze_oracle.c header): the checker for hm_hash_bag* / hm_hash_knn.
 *
 * PARITY UNPINNED.  The call sites are in the reference,
 *   cv-sfm/src/lib.rs:205,216   hasher: HammingHasher<64, 512> = new_with_codewords(codewords::codewords())
 *                               (4096 = 512 * 8 codewords of 64 bytes, cv-sfm/src/codewords.rs)
 *   cv-sfm/src/lib.rs:672       lsh = hasher.hash_bag(features.iter().map(|(d, _)| d))      -> BitArray<512>
 *   cv-sfm/src/lib.rs:684       lsh_to_frame.insert(lsh, frame)
 *   cv-sfm/src/lib.rs:622-624   lsh_to_frame.knn_values(&frames[frame].lsh, similar_frames_search_num)
 * but the hashing itself lives in the un-vendored crate `hamming-lsh` 0.3.2 (cv-sfm/Cargo.toml:40) and the
 * reference holds no test or golden vector for it.  Restated here as the bag-of-words form the type
 * parameters imply (one hash bit per codeword; SURVEY.md §8f: "descriptor -> nearest codeword"): every feature
 * sets the bit of its nearest codeword, ties to the lowest codeword index (Iterator::min_by_key keeps the first
 * minimum), bit i at byte i >> 3, position i & 7 (the BitArray order of akaze/src/descriptors.rs:197).  The word
 * table (nearest codeword and its distance per feature) is returned as well, so a count-based variant of the
 * hash can be derived from the same pass.  The frame search is HGG's approximate knn in the reference; here it
 * is the exact answer in (distance, insertion index) order.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/akz.h"

static inline uint32_t hamming_bytes(const uint8_t* a, const uint8_t* b, uint32_t n)
{
    uint32_t d = 0;
    for (uint32_t i = 0; i < n; ++i) d += (uint32_t)__builtin_popcount((unsigned)(a[i] ^ b[i]));
    return d;
}

/* hash_bag: hash[n_codewords / 8] (zeroed here), words[n] = {nearest codeword, distance} (may be NULL). */
int orc_hash_bag(const akz_descriptor* feats, uint32_t n, const akz_descriptor* codewords, uint32_t n_codewords,
                 uint8_t* hash, akz_neighbor* words)
{
    if (n_codewords == 0 || (n_codewords & 31u)) return -1;
    memset(hash, 0, n_codewords / 8);
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t best = 0, bd = hamming_bytes(feats[i].bytes, codewords[0].bytes, 64);
        for (uint32_t j = 1; j < n_codewords; ++j) {
            uint32_t d = hamming_bytes(feats[i].bytes, codewords[j].bytes, 64);
            if (d < bd) {
                bd = d;
                best = j;
            }
        }
        hash[best >> 3] |= (uint8_t)(1u << (best & 7u));
        if (words) {
            words[i].index = best;
            words[i].distance = bd;
        }
    }
    return 0;
}

/* The k stored hashes nearest to `query`, ascending (distance, index); returns how many were written. */
uint32_t orc_hash_knn(const uint8_t* query, const uint8_t* hashes, uint32_t n, uint32_t hash_bytes, uint32_t k,
                      akz_neighbor* out)
{
    uint32_t m = 0;
    for (uint32_t i = 0; i < n; ++i) {
        akz_neighbor nb = {i, hamming_bytes(query, hashes + (size_t)i * hash_bytes, hash_bytes)};
        /* insertion after every entry with distance <= d: equal distances keep insertion order */
        uint32_t pos = m;
        while (pos > 0 && out[pos - 1].distance > nb.distance) --pos;
        if (pos >= k) continue;
        uint32_t last = m < k ? m : k - 1;
        for (uint32_t j = last; j > pos; --j) out[j] = out[j - 1];
        out[pos] = nb;
        if (m < k) ++m;
    }
    return m;
}
