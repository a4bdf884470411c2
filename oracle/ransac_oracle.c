/* ransac_oracle.c — CPU restatement of the reference's two-view geometric verification (SURVEY.md §8a
 * rows R1-R4): calibrate -> eight-point essential -> four candidate poses -> triangulation residual ->
 * consensus by exhaustive scoring of caller-provided minimal samples.
 *
 * TEST INFRASTRUCTURE ONLY (see akaze_oracle.c header).
 *
 * In-tree reference code restated here (paths relative to rust-cv/cv):
 *   CameraIntrinsics::calibrate, K1 variant          cv-pinhole/src/lib.rs:108-117, 191-202
 *   encode_epipolar_equation, EightPoint::from_matches   eight-point/src/lib.rs:11-58
 *   possible_rotations_unscaled_translation / _poses     cv-pinhole/src/essential.rs:114-162, 217-231
 *   CameraToCamera::residual                              cv-core/src/pose.rs:249-295
 *   Projective::from_homogeneous, bearing, transform      cv-core/src/point.rs:20-49, pose.rs:125-133
 * Un-vendored pieces: nalgebra's symmetric eigen / SVD (replaced on BOTH sides by the cyclic Jacobi of
 * include/akz_ransac_math.h) and the `arrsac` crate (sampling + SPRT): here the caller provides the
 * minimal-sample indices and every hypothesis is scored against every match, the winner being the pose
 * with the most inliers (ties: lowest hypothesis, then lowest pose index).  **Parity unpinned** against
 * the reference beyond its count pin inliers.len() == 11 (akaze/tests/estimate_pose.rs:75); parity is
 * oracle == HIP, bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/akz.h"
#include "../include/akz_ransac_math.h"

/* calibrate: pixel -> unit bearing.  k1 == 0 reproduces the plain CameraIntrinsics arm exactly
 * (division by 1.0 + 0*r2 == 1.0 is the identity). use_k1 selects the arm. */
void orc_calibrate(const double* intr /* fx, fy, cx, cy, skew */, int use_k1, double k1, const akz_keypoint* kps,
                   uint32_t n, double* out)
{
    for (uint32_t i = 0; i < n; ++i) {
        double px = (double)kps[i].x, py = (double)kps[i].y; /* ImagePoint::image_point: f32 -> f64 */
        double cx = px - intr[2], cy = py - intr[3];
        double y = cy / intr[1];
        double x = (cx - intr[4] * y) / intr[0];
        if (use_k1) {
            double r2 = x * x + y * y;
            double d = 1.0 + k1 * r2;
            x = x / d;
            y = y / d;
        }
        double nrm = sqrt(x * x + y * y + 1.0 * 1.0); /* UnitVector3::new_normalize([x, y, 1]) */
        out[3 * i + 0] = x / nrm;
        out[3 * i + 1] = y / nrm;
        out[3 * i + 2] = 1.0 / nrm;
    }
}

/* EightPoint::from_matches — eight-point/src/lib.rs:11-58. E is returned row-major (E[r*3+c]). */
int orc_eight_point(const double* a8 /*[8][3]*/, const double* b8, double eps, int iters, double* E)
{
    double A[8][9];
    for (int i = 0; i < 8; ++i) {
        const double* a = a8 + 3 * i;
        const double* b = b8 + 3 * i;
        double ap[3] = {a[0] / a[2], a[1] / a[2], a[2] / a[2]};
        double bp[3] = {b[0] / a[2], b[1] / a[2], b[2] / a[2]}; /* sic: divided by a.z (:16) */
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k) A[i][3 * j + k] = ap[j] * bp[k];
    }
    double M[81], V[81];
    for (int r = 0; r < 9; ++r)
        for (int c = 0; c < 9; ++c) {
            double s = 0.0;
            for (int i = 0; i < 8; ++i) s += A[i][r] * A[i][c];
            M[r * 9 + c] = s;      /* symmetric bit for bit (the products commute); the solver reads r <= c only */
        }
    akz_rm_jacobi9_sym(M, V, eps, iters);
    int best = 0;
    for (int i = 1; i < 9; ++i)
        if (M[i * 9 + i] < M[best * 9 + best]) best = i; /* min_by_key(FloatOrd): first minimum */
    /* Matrix3::from_iterator(eigenvector) is column-major: E(r,c) = v[c*3 + r] */
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) E[r * 3 + c] = V[(c * 3 + r) * 9 + best];
    for (int i = 0; i < 9; ++i)
        if (!isfinite(E[i])) return -1;
    return 0;
}

static void mat3_mul(const double* a, const double* b, double* o)
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += a[r * 3 + k] * b[k * 3 + c];
            o[r * 3 + c] = s;
        }
}

/* possible_unscaled_poses — cv-pinhole/src/essential.rs:114-162,217-231.  poses[4][12], each a row-major
 * 3x4 [R | t], in the reference's order (t,R1), (t,R2), (-t,R1), (-t,R2). */
int orc_essential_poses(const double* E, double eps, int iters, double* poses)
{
    double M[9], V[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += E[k * 3 + r] * E[k * 3 + c];
            M[r * 3 + c] = s;
        }
    akz_rm_jacobi3(M, V, 1, eps, iters);
    int ord[3] = {0, 1, 2}; /* singular values descending, stable */
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
        double s = sqrt(lam > 0.0 ? lam : 0.0);
        if (!(s > 0.0)) return -1;
        for (int r = 0; r < 3; ++r) {
            double acc = 0.0;
            for (int k = 0; k < 3; ++k) acc += E[r * 3 + k] * Vs[k * 3 + c];
            U[r * 3 + c] = acc / s;
        }
    }
    /* third column: with d = (a a 0) it is undetermined up to sign and the reference forces det(U) > 0 */
    U[0 * 3 + 2] = U[1 * 3 + 0] * U[2 * 3 + 1] - U[2 * 3 + 0] * U[1 * 3 + 1];
    U[1 * 3 + 2] = U[2 * 3 + 0] * U[0 * 3 + 1] - U[0 * 3 + 0] * U[2 * 3 + 1];
    U[2 * 3 + 2] = U[0 * 3 + 0] * U[1 * 3 + 1] - U[1 * 3 + 0] * U[0 * 3 + 1];
    /* force det(V^T) > 0 by flipping its last row (= last column of V) */
    double detV = Vs[0] * (Vs[4] * Vs[8] - Vs[5] * Vs[7]) - Vs[1] * (Vs[3] * Vs[8] - Vs[5] * Vs[6]) +
                  Vs[2] * (Vs[3] * Vs[7] - Vs[4] * Vs[6]);
    if (detV < 0.0)
        for (int r = 0; r < 3; ++r) Vs[r * 3 + 2] = -Vs[r * 3 + 2];
    double Vt[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Vt[r * 3 + c] = Vs[c * 3 + r];
    const double W[9] = {0.0, -1.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0};
    const double Wt[9] = {0.0, 1.0, 0.0, -1.0, 0.0, 0.0, 0.0, 0.0, 1.0};
    double UW[9], R1[9], R2[9];
    mat3_mul(U, W, UW);
    mat3_mul(UW, Vt, R1);
    mat3_mul(U, Wt, UW);
    mat3_mul(UW, Vt, R2);
    double t[3] = {U[2], U[5], U[8]};
    for (int p = 0; p < 4; ++p) {
        const double* R = (p & 1) ? R2 : R1;
        double sg = (p & 2) ? -1.0 : 1.0;
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) poses[p * 12 + r * 4 + c] = R[r * 3 + c];
            poses[p * 12 + r * 4 + 3] = sg * t[r];
        }
    }
    for (int i = 0; i < 48; ++i)
        if (!isfinite(poses[i])) return -1;
    return 0;
}

/* CameraToCamera::residual — cv-core/src/pose.rs:249-295. pose = row-major 3x4 [R | t]. */
double orc_residual(const double* pose, const double* a, const double* b, double eps, int iters)
{
    static const double ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    double design[16];
    for (int i = 0; i < 16; ++i) design[i] = 0.0;
    for (int view = 0; view < 2; ++view) {
        const double* P = view == 0 ? ident : pose;
        const double* br = view == 0 ? a : b;
        double bbt[9], bP[12], term[12];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) bbt[r * 3 + c] = br[r] * br[c];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) {
                double s = 0.0;
                for (int k = 0; k < 3; ++k) s += bbt[r * 3 + k] * P[k * 4 + c];
                bP[r * 4 + c] = s;
            }
        for (int i = 0; i < 12; ++i) term[i] = P[i] - bP[i];
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) {
                double s = 0.0;
                for (int k = 0; k < 3; ++k) s += term[k * 4 + r] * term[k * 4 + c];
                design[r * 4 + c] += s;
            }
    }
    double V[16];
    akz_rm_jacobi4_sym(design, V, eps, iters);
    int best = 0;
    for (int i = 1; i < 4; ++i)
        if (fabs(design[i * 4 + i]) < fabs(design[best * 4 + best])) best = i; /* min_by_key(|l|.to_bits()) */
    double p[4] = {V[0 * 4 + best], V[1 * 4 + best], V[2 * 4 + best], V[3 * 4 + best]};
    /* Projective::from_homogeneous — point.rs:20-25 */
    if (signbit(p[3]))
        for (int i = 0; i < 4; ++i) p[i] = -p[i];
    double nrm = sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    for (int i = 0; i < 4; ++i) p[i] = p[i] / nrm;
    for (int i = 0; i < 4; ++i)
        if (!isfinite(p[i])) return 2.0;
    /* transform: isometry.to_homogeneous() * p, then from_homogeneous again (pose.rs:125-133) */
    double q[4];
    for (int r = 0; r < 3; ++r)
        q[r] = ((pose[r * 4 + 0] * p[0] + pose[r * 4 + 1] * p[1]) + pose[r * 4 + 2] * p[2]) + pose[r * 4 + 3] * p[3];
    q[3] = p[3];
    if (signbit(q[3]))
        for (int i = 0; i < 4; ++i) q[i] = -q[i];
    double qn = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    for (int i = 0; i < 4; ++i) q[i] = q[i] / qn;
    double ad = (a[0] * p[0] + a[1] * p[1]) + a[2] * p[2];
    double bd = (b[0] * q[0] + b[1] * q[1]) + b[2] * q[2];
    double res = 0.5 * (1.0 - ad + 1.0 - bd);
    return res == res ? res : 2.0;
}

/* Exhaustive consensus over caller-provided minimal samples (the role arrsac::Arrsac::model_inliers plays
 * at akaze/tests/estimate_pose.rs:63-67, tutorial ch5 main.rs:70-72, cv-sfm/src/lib.rs:1394-1406).
 * counts (optional) receives n_hyp*4 inlier counts (0 for failed hypotheses). Returns 0, or -1 if no
 * hypothesis produced a model. */
int orc_essential_batch(const double* ba, const double* bb, uint32_t n, const uint32_t* sample_idx, uint32_t n_hyp,
                        double thresh, double eps, int iters, double* best_pose, uint32_t* best_hyp_pose,
                        uint32_t* inlier_idx, uint32_t* n_inliers, uint32_t* counts)
{
    uint32_t best_count = 0, best_id = 0xFFFFFFFFu;
    double bestp[12];
    for (uint32_t hh = 0; hh < n_hyp; ++hh) {
        double a8[24], b8[24], E[9], poses[48];
        for (int i = 0; i < 8; ++i) {
            uint32_t m = sample_idx[hh * 8 + i];
            memcpy(a8 + 3 * i, ba + 3 * m, 24);
            memcpy(b8 + 3 * i, bb + 3 * m, 24);
        }
        int ok = orc_eight_point(a8, b8, eps, iters, E) == 0 && orc_essential_poses(E, eps, iters, poses) == 0;
        for (int p = 0; p < 4; ++p) {
            uint32_t cnt = 0;
            if (ok)
                for (uint32_t m = 0; m < n; ++m)
                    if (orc_residual(poses + 12 * p, ba + 3 * m, bb + 3 * m, eps, iters) < thresh) cnt++;
            if (counts) counts[hh * 4 + p] = cnt;
            if (ok && (best_id == 0xFFFFFFFFu || cnt > best_count)) {
                best_count = cnt;
                best_id = hh * 4 + (uint32_t)p;
                memcpy(bestp, poses + 12 * p, sizeof(bestp));
            }
        }
    }
    if (best_id == 0xFFFFFFFFu) return -1;
    memcpy(best_pose, bestp, sizeof(bestp));
    *best_hyp_pose = best_id;
    uint32_t k = 0;
    for (uint32_t m = 0; m < n; ++m)
        if (orc_residual(bestp, ba + 3 * m, bb + 3 * m, eps, iters) < thresh) inlier_idx[k++] = m;
    *n_inliers = k;
    return 0;
}
