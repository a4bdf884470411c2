/* batch_oracle.c — the oracle over a batch of frames on all host cores: what bench.py's cpu_baseline legs time.
 *
 * TEST INFRASTRUCTURE ONLY (see akaze_oracle.c header).
 *
 * SURVEY.md §8d "CPU baseline timing": the reference binary cannot be built here (no Rust toolchain), so the timed CPU
 * path is this C restatement, built a second time as liboracle_fast.so with -O3 -march=native -fopenmp (still
 * -ffp-contract=off, no fast-math: bench.py asserts its outputs bit-identical to the -O2 checker before timing it) and
 * parallel over frames, the way a caller looping Akaze::extract over frames with rayon would be
 * (cv-sfm/src/lib.rs:2200-2204 extracts per frame; akaze's own rayon points are inside a frame: lib.rs:243,
 * detector_response.rs:21,54,71-83, scale_space_extrema.rs:352, descriptors.rs:35).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/akz.h"

typedef struct orc_ctx orc_ctx;
orc_ctx* orc_create(const akz_config* cfg, int width, int height);
void orc_destroy(orc_ctx* c);
int orc_extract_u8(orc_ctx* c, const uint8_t* image, int stride);
uint32_t orc_keypoints(const orc_ctx* c, int stage, const akz_keypoint** out);
const akz_descriptor* orc_descriptors(const orc_ctx* c);
int orc_match(const akz_descriptor* a, uint32_t na, const akz_descriptor* b, uint32_t nb, int rule, uint32_t pu, float pf,
              int symmetric, uint32_t* pairs, uint32_t cap);

void orc_set_option(int which, int value);

int orc_threads_available(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* n frames (w x h Luma8, back to back) -> keypoints / descriptors [n][cap], counts [n]; then, with match != 0, the
 * symmetric better-by-`param_u` (strict) match of frame i against frame i-1 -> pairs [n][cap][2], npairs [n] (frame 0:
 * none).  threads <= 0: all the runtime gives.  Returns 0, or -1 when a frame has more than cap keypoints. */
int orc_extract_match_many_u8(const akz_config* cfg, int w, int h, const uint8_t* imgs, int n, int threads, uint32_t cap,
                              akz_keypoint* kps, akz_descriptor* descs, uint32_t* counts, int match, uint32_t param_u,
                              uint32_t* pairs, uint32_t* npairs)
{
    int bad = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        orc_ctx* c = orc_create(cfg, w, h);           /* one pyramid per thread, reused over its frames */
#pragma omp for schedule(dynamic, 1)
        for (int i = 0; i < n; ++i) {
            orc_extract_u8(c, imgs + (size_t)i * w * h, w);
            const akz_keypoint* k = NULL;
            uint32_t m = orc_keypoints(c, 3, &k);
            if (m > cap) {
#pragma omp atomic write
                bad = 1;
                m = cap;
            }
            counts[i] = m;
            memcpy(kps + (size_t)i * cap, k, sizeof(akz_keypoint) * m);
            memcpy(descs + (size_t)i * cap, orc_descriptors(c), sizeof(akz_descriptor) * m);
        }
        orc_destroy(c);
    }
    if (match) {
#pragma omp parallel for schedule(dynamic, 1)
        for (int i = 1; i < n; ++i) {
            int r = orc_match(descs + (size_t)i * cap, counts[i], descs + (size_t)(i - 1) * cap, counts[i - 1], 0 /* d0 + N < d1 */,
                              param_u, 0.0f, 1, pairs + (size_t)i * cap * 2, cap);
            npairs[i] = r < 0 ? 0u : (uint32_t)r;
        }
        if (n > 0) npairs[0] = 0;
    }
    return bad ? -1 : 0;
}

/* The reference as its own driver runs it: vslam-sandbox hands frames to VSlam::add_frame ONE AT A TIME
 * (vslam-sandbox/src/main.rs:124-160), and a frame's extraction is parallel only at the akaze crate's `rayon` points
 * (ORC_OPT_INTRA, akaze_oracle.c header).  n frames in sequence, each on `threads` threads; outputs as above (no matching).
 * Returns 0, or -1 when a frame has more than cap keypoints. */
int orc_extract_many_intra_u8(const akz_config* cfg, int w, int h, const uint8_t* imgs, int n, int threads, uint32_t cap,
                              akz_keypoint* kps, akz_descriptor* descs, uint32_t* counts)
{
    int bad = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    orc_set_option(4 /* ORC_OPT_INTRA */, 1);
    orc_ctx* c = orc_create(cfg, w, h);
    for (int i = 0; i < n; ++i) {
        orc_extract_u8(c, imgs + (size_t)i * w * h, w);
        const akz_keypoint* k = NULL;
        uint32_t m = orc_keypoints(c, 3, &k);
        if (m > cap) {
            bad = 1;
            m = cap;
        }
        counts[i] = m;
        memcpy(kps + (size_t)i * cap, k, sizeof(akz_keypoint) * m);
        memcpy(descs + (size_t)i * cap, orc_descriptors(c), sizeof(akz_descriptor) * m);
    }
    orc_destroy(c);
    orc_set_option(4, 0);
    return bad ? -1 : 0;
}
