/* p3p_oracle.c — CPU side of the PnP (registration) consensus: Lambda Twist P3P hypotheses from
 * caller-provided sample triples, WorldToCamera::residual scoring, best pose by inlier count
 * (reference call site cv-sfm/src/lib.rs:1619-1622,
 * lambda-twist/tests/consensus.rs:59-61).
 *
 * TEST INFRASTRUCTURE ONLY.  The solver arithmetic is include/akz_p3p_math.h (a restatement of
 * lambda-twist/src/lib.rs shared with the kernels, see its header for what is un-vendored); this file is
 * the plain-C driver the HIP path is compared against.  Pinned by lambda-twist/tests/consensus.rs
 * (tests/test_oracle_ransac.py): pose recovered within 1e-6; the 9-sample regression case terminates.
 */
#include <stdint.h>
#include <string.h>

#include "../include/akz_p3p_math.h"

int orc_p3p_poses(const double* bearings3, const double* world3, double* poses)
{
    return akz_p3p_poses(bearings3, world3, 5, poses); /* LambdaTwist::default: 5 Gauss-Newton iterations */
}
double orc_w2c_residual(const double* pose, const double* bearing, const double* world)
{
    return akz_w2c_residual(pose, bearing, world);
}

/* counts: [n_hyp][4] (0 for absent poses). returns 0 / -1 (no model). */
int orc_p3p_batch(const double* bearings, const double* world, uint32_t n, const uint32_t* sample_idx, uint32_t n_hyp,
                  double thresh, double* best_pose, uint32_t* best_id, uint32_t* inlier_idx, uint32_t* n_inliers,
                  uint32_t* counts)
{
    uint32_t best_count = 0, bid = 0xFFFFFFFFu;
    double bestp[12];
    for (uint32_t hh = 0; hh < n_hyp; ++hh) {
        double b3[9], w3[12], poses[48];
        for (int i = 0; i < 3; ++i) {
            uint32_t m = sample_idx[hh * 3 + i];
            memcpy(b3 + 3 * i, bearings + 3 * m, 24);
            memcpy(w3 + 4 * i, world + 4 * m, 32);
        }
        int np = akz_p3p_poses(b3, w3, 5, poses);
        for (int p = 0; p < 4; ++p) {
            uint32_t cnt = 0;
            if (p < np)
                for (uint32_t m = 0; m < n; ++m)
                    if (akz_w2c_residual(poses + 12 * p, bearings + 3 * m, world + 4 * m) < thresh) cnt++;
            if (counts) counts[hh * 4 + p] = cnt;
            if (p < np && (bid == 0xFFFFFFFFu || cnt > best_count)) {
                best_count = cnt;
                bid = hh * 4 + (uint32_t)p;
                memcpy(bestp, poses + 12 * p, sizeof(bestp));
            }
        }
    }
    if (bid == 0xFFFFFFFFu) return -1;
    memcpy(best_pose, bestp, sizeof(bestp));
    *best_id = bid;
    uint32_t k = 0;
    for (uint32_t m = 0; m < n; ++m)
        if (akz_w2c_residual(bestp, bearings + 3 * m, world + 4 * m) < thresh) inlier_idx[k++] = m;
    *n_inliers = k;
    return 0;
}
