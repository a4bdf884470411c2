// rs_ransac.hip — batched two-view geometric verification on gfx950 (SURVEY.md §8a rows R1-R5, §8f rank 1,
// BASELINE.json configs[3]): eight-point essential matrices for batches of minimal samples, the four candidate poses
// of each, the triangulation residual of every (pose, match) pair, consensus by inlier count — for one scene with
// host buffers, or for a whole micro-batch of frame pairs straight from the matcher's device-resident pair lists.
// All f64 VALU work, -ffp-contract=off, the same operation sequence as oracle/ransac_oracle.c.
//
// Reference code implemented here (paths relative to rust-cv/cv):
//   CameraIntrinsics::calibrate (+K1)                     cv-pinhole/src/lib.rs:108-117,191-202   rs_calibrate, k_rsb_prepare
//   encode_epipolar_equation, EightPoint::from_matches    eight-point/src/lib.rs:11-58            k_rsb_hypotheses
//   EssentialMatrix::possible_unscaled_poses              cv-pinhole/src/essential.rs:114-162,217-231  k_rsb_hypotheses
//   CameraToCamera::residual                              cv-core/src/pose.rs:249-295             k_rsb_score_first, k_rsb_score
//   Consensus::model_inliers                              call sites akaze/tests/estimate_pose.rs:63-67,
//                                                         tutorial ch5 main.rs:70-72, cv-sfm/src/lib.rs:1394-1412
// nalgebra's eigen/SVD and the arrsac sampler are un-vendored: the eigen-solver is the shared cyclic Jacobi of
// include/akz_ransac_math.h, the consensus loop is specified by oracle/arrsac_oracle.c (parity: oracle == HIP, bit for
// bit).
//
// Layout: everything a call needs lives in one per-context arena of S scene slots (rs_batch_reserve); a scene is one
// frame pair: its matches (bearings a / b, or bearings / world points), its hypotheses, poses, counters and live list.
// Every kernel of the ARRSAC-shaped loop takes the scene from its block index, so ONE launch chain serves all scenes
// of a micro-batch (grid = scenes x hypotheses): the launches a single scene is bound by are shared by 256 of them.
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

// cv_pinhole::CameraIntrinsics::calibrate / CameraIntrinsicsK1Distortion::calibrate for one keypoint
// (cv-pinhole/src/lib.rs:108-117,191-202): the host entry point and the batch kernel share this statement.
__host__ __device__ __forceinline__ void rs_calibrate_one(const double* intr /* fx, fy, cx, cy, skew */, int use_k1, double k1,
                                                          float kx, float ky, double* out)
{
    double cx = (double)kx - intr[2], cy = (double)ky - intr[3];
    double y = cy / intr[1];
    double x = (cx - intr[4] * y) / intr[0];
    if (use_k1) {
        double r2 = x * x + y * y;
        double d = 1.0 + k1 * r2;
        x = x / d;
        y = y / d;
    }
    double nrm = AKZ_RM_SQRT(x * x + y * y + 1.0 * 1.0);
    out[0] = x / nrm;
    out[1] = y / nrm;
    out[2] = 1.0 / nrm;
}

// `WorldToCamera::residual(pose, match) < thresh` (cv-core/src/pose.rs:194-201) — the boolean only.  The exact statement
// (akz_w2c_residual) spends most of its instructions on three f64 divisions and a square root; almost every (pose, match)
// pair is far from the threshold, so the test is first made without either:
//     1 - (a . q) / |q| < thresh   <=>   a . q > 0  and  (a . q)^2 > (1 - thresh)^2 (q . q)        (0 < thresh < 1/2)
// on q = [R | t] w evaluated with fused multiply-adds — 24 instructions instead of the 55 of a reciprocal square root with
// two Newton steps (round 3), which was the registration consensus' whole cost (k_rsb_score_p3p: 58 % of a pipeline+register
// step).  The pre-test's value differs from the exact statement's by parts in 1e15; it decides only outside a band of
// +-2e-9 around the threshold (the two constants below), the exact statement inside it, and whenever q . q is not an
// ordinary number or the threshold is outside (0, 1/2).  The result is the exact statement's for every input
// (tests/test_gpu_parity.py::test_p3p_inlier_test_is_exact_at_the_threshold).
__device__ __forceinline__ bool rs_w2c_inlier(const double* __restrict__ pose, const double* __restrict__ a,
                                              const double* __restrict__ w, double thresh)
{
    double q[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
        // q = [R | t] w in the exact statement's own unfused order ((a + b) + c) + d (akz_w2c_residual): when R w + t cancels —
        // a world point near the hypothesis' camera centre — a fused q would differ from the exact one by far more than
        // the band below; s2 and c have no cancellation in the band (|c| ~ |q| there) and keep their FMAs
        q[r] = ((pose[r * 4 + 0] * w[0] + pose[r * 4 + 1] * w[1]) + pose[r * 4 + 2] * w[2]) + pose[r * 4 + 3] * w[3];
    const double s2 = __builtin_fma(q[2], q[2], __builtin_fma(q[1], q[1], q[0] * q[0]));
    double c = __builtin_fma(a[2], q[2], __builtin_fma(a[1], q[1], a[0] * q[0]));
    if (__builtin_signbit(w[3])) c = -c;
    const double u = 1.0 - thresh, u2 = u * u;                       // (loop-invariant in every caller)
    const double k_in = u2 * (1.0 + 4e-9), k_out = u2 * (1.0 - 4e-9);
    if (thresh > 0.0 && thresh < 0.5 && s2 > 1e-280 && s2 < 1e280) {
        const double c2 = c * c;
        if (c > 0.0 && c2 > k_in * s2) return true;
        if (c <= 0.0 || c2 < k_out * s2) return false;               // (a NaN fails every comparison: the exact statement decides)
    }
    return akz_w2c_residual(pose, a, w) < thresh;
}

// EightPoint::from_matches + possible_unscaled_poses for one minimal sample, ONE LANE, registers only: the 9 x 9
// normal matrix (upper triangle) and its 81 eigenvector components never leave the register file (akz_rm_jacobi9_sym;
// the kernel is compiled for one wave per SIMD, 512 VGPRs).  a / b: the scene's bearings; smp: 8 match indices.
// Returns validity; poses[4][12] row-major [R | t] in the reference's order (t,R1), (t,R2), (-t,R1), (-t,R2).
__device__ __forceinline__ bool rs_eight_point_poses(const double* __restrict__ ba, const double* __restrict__ bb,
                                                     const uint32_t* __restrict__ smp, double* __restrict__ out)
{
    double M[81], V[81];
#pragma unroll
    for (int i = 0; i < 81; ++i) M[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t m = smp[i];
        const double* a = ba + (size_t)3 * m;
        const double* b = bb + (size_t)3 * m;
        const double az = a[2];
        const double ap[3] = {a[0] / az, a[1] / az, a[2] / az};
        const double bp[3] = {b[0] / az, b[1] / az, b[2] / az};  // sic: divided by a.z (eight-point/src/lib.rs:16)
        double A[9];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) A[3 * j + k] = ap[j] * bp[k];
        // M(r,c) = sum_i A_i[r] A_i[c], i ascending, starting from +0 — entry by entry the oracle's sum
#pragma unroll
        for (int r = 0; r < 9; ++r)
#pragma unroll
            for (int c = r; c < 9; ++c) M[r * 9 + c] += A[r] * A[c];
    }
    akz_rm_jacobi9_sym(M, V, kEpsHyp, kItersHyp);
    // eigenvector of the smallest eigenvalue (first minimum), selected without a runtime index
    double bestv = M[0];
    double ev[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) ev[e] = V[e * 9];
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        const bool take = M[i * 9 + i] < bestv;
        bestv = take ? M[i * 9 + i] : bestv;
#pragma unroll
        for (int e = 0; e < 9; ++e) ev[e] = take ? V[e * 9 + i] : ev[e];
    }
    double E[9];
    bool good = true;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            E[r * 3 + c] = ev[c * 3 + r];  // Matrix3::from_iterator is column-major
            good = good && finite_d(E[r * 3 + c]);
        }
    // possible_unscaled_poses: SVD of E through the eigen-decomposition of E^T E
    double M3[9], V3[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) s += E[k * 3 + r] * E[k * 3 + c];
            M3[r * 3 + c] = s;
        }
    akz_rm_jacobi3(M3, V3, 1, kEpsHyp, kItersHyp);
    // singular values descending, stable: the oracle's exchange sort of an index triple, here on (value, column)
    // triples with selects — (0,1), (0,2), (1,2), exchanging when the later one is strictly larger
    double lam[3] = {M3[0], M3[4], M3[8]};
    double col[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) col[c][r] = V3[r * 3 + c];
#define RS_CSWAP(I, J)                                           \
    {                                                            \
        const bool sw = lam[J] > lam[I];                         \
        const double tl = lam[I];                                \
        lam[I] = sw ? lam[J] : lam[I];                           \
        lam[J] = sw ? tl : lam[J];                               \
        for (int r = 0; r < 3; ++r) {                            \
            const double tc = col[I][r];                         \
            col[I][r] = sw ? col[J][r] : col[I][r];              \
            col[J][r] = sw ? tc : col[J][r];                     \
        }                                                        \
    }
    RS_CSWAP(0, 1)
    RS_CSWAP(0, 2)
    RS_CSWAP(1, 2)
#undef RS_CSWAP
    double Vs[9], U[9];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) Vs[r * 3 + c] = col[c][r];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const double l = lam[c];
        const double s = AKZ_RM_SQRT(l > 0.0 ? l : 0.0);
        if (!(s > 0.0)) good = false;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) acc += E[r * 3 + k] * Vs[k * 3 + c];
            U[r * 3 + c] = acc / s;
        }
    }
    U[0 * 3 + 2] = U[1 * 3 + 0] * U[2 * 3 + 1] - U[2 * 3 + 0] * U[1 * 3 + 1];
    U[1 * 3 + 2] = U[2 * 3 + 0] * U[0 * 3 + 1] - U[0 * 3 + 0] * U[2 * 3 + 1];
    U[2 * 3 + 2] = U[0 * 3 + 0] * U[1 * 3 + 1] - U[1 * 3 + 0] * U[0 * 3 + 1];
    const double detV = Vs[0] * (Vs[4] * Vs[8] - Vs[5] * Vs[7]) - Vs[1] * (Vs[3] * Vs[8] - Vs[5] * Vs[6]) +
                        Vs[2] * (Vs[3] * Vs[7] - Vs[4] * Vs[6]);
    if (detV < 0.0)
#pragma unroll
        for (int r = 0; r < 3; ++r) Vs[r * 3 + 2] = -Vs[r * 3 + 2];
    // R1 = U W V^T, R2 = U W^T V^T with W = [[0,-1,0],[1,0,0],[0,0,1]]; the products are written out with
    // the same term order as the oracle's generic 3x3 multiply (k = 0,1,2, zeros included)
    const double W[9] = {0.0, -1.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0};
    const double Wt[9] = {0.0, 1.0, 0.0, -1.0, 0.0, 0.0, 0.0, 0.0, 1.0};
    double R[2][9];
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        double UW[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < 3; ++k) s += U[r * 3 + k] * (which ? Wt[k * 3 + c] : W[k * 3 + c]);
                UW[r * 3 + c] = s;
            }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < 3; ++k) s += UW[r * 3 + k] * Vs[c * 3 + k];  // V^T(k,c) = Vs(c,k)
                R[which][r * 3 + c] = s;
            }
    }
    const double t[3] = {U[2], U[5], U[8]};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const double sg = (p & 2) ? -1.0 : 1.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                out[p * 12 + r * 4 + c] = R[p & 1][r * 3 + c];
                good = good && finite_d(R[p & 1][r * 3 + c]);
            }
            out[p * 12 + r * 4 + 3] = sg * t[r];
            good = good && finite_d(t[r]);
        }
    }
    return good;
}

// CameraToCamera::residual for one (pose, match) — cv-core/src/pose.rs:249-295.
// The design matrix is sum over the two views of (P - b b^T P)^T (P - b b^T P).  The oracle evaluates both views with
// the generic expression; for the first view P = [I | 0], whose zeros and ones make most of that arithmetic vacuous:
//   term(r,c) = delta_rc - a_r a_c (c < 3), +0 (c = 3)  — (x*1 = x, sums of signed zeros start from +0, and
//   0 - (+-0) = +0 either way), so its contribution is the 3 x 3 block sum_k term(k,r) term(k,c) and +0 elsewhere.
// Dropping the leading "0.0 +" of that block's sums cannot change the final entry: it could only turn a -0 partial
// sum into +0, and the second view's sum — which keeps its own leading +0 and is therefore never -0 — is added on top
// (x + s1 is the same for x = +-0).  Only the upper triangle is built: akz_rm_jacobi4_sym reads nothing else.
//
// The mirrored pose comes for free.  possible_unscaled_poses returns (t,R1), (t,R2), (-t,R1), (-t,R2): poses p and
// p + 2 differ in the sign of t only.  With S = diag(1,1,1,-1) the design matrix of [R | -t] is S D S entry by entry
// (negation commutes with every rounding), the Jacobi iteration on it is the mirror image of the one on D — same
// h, same c, t and s negated in the rotations that involve index 3, same sweeps — and its eigenvectors are S V S.
// The normalised triangulated point of the mirrored pose is therefore (-x, -y, -z, w) / |xyz| where the pose's own
// is (x, y, z, w) / |xyz|, its image in the second camera is the negated one, and
//     residual(-t) = 0.5 (1 + a.p + 1 + b.q)      next to      residual(t) = 0.5 (1 - a.p + 1 - b.q).
// Exactness: every mirrored quantity is the exact negation of its counterpart or a zero of either sign, and a zero's
// sign can reach a result through one place only — signbit(p[3]) when the eigenvector's last component is exactly
// zero.  Those lanes (a point exactly at infinity) evaluate the mirrored pose directly in a second pass of the same
// code; for all others the two residuals of a pair cost one eigen-decomposition.
// (NOT force-inlined: with always_inline the sweep loop of the eigen-solver is optimised after inlining into the kernels'
// own loops and comes out with twice the registers — 249 instead of 122 VGPRs — and a third more instructions)
__device__ void rs_residual_core(const double* __restrict__ pose, double tsign, const double* a, const double* b,
                                                 double* res_out, double* res_mirror, double* w_raw)
{
    double design[16], V[16];
    {
        double t0[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) t0[r * 3 + c] = (r == c ? 1.0 : 0.0) - a[r] * a[c];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = r; c < 3; ++c)
                design[r * 4 + c] = (t0[0 * 3 + r] * t0[0 * 3 + c] + t0[1 * 3 + r] * t0[1 * 3 + c]) + t0[2 * 3 + r] * t0[2 * 3 + c];
#pragma unroll
        for (int r = 0; r < 4; ++r) design[r * 4 + 3] = 0.0;
    }
    double P[12];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) P[r * 4 + c] = pose[r * 4 + c];
        P[r * 4 + 3] = tsign * pose[r * 4 + 3];      // tsign = +-1: exact
    }
    {
        double term[12];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < 3; ++k) s += (b[r] * b[k]) * P[k * 4 + c];
                term[r * 4 + c] = P[r * 4 + c] - s;
            }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = r; c < 4; ++c) {
                double s = 0.0;
#pragma unroll
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
    *w_raw = p[3];
    if (__builtin_signbit(p[3]))
        for (int i = 0; i < 4; ++i) p[i] = -p[i];
    double nrm = AKZ_RM_SQRT(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    for (int i = 0; i < 4; ++i) p[i] = p[i] / nrm;
    bool fin = true;
    for (int i = 0; i < 4; ++i) fin = fin && finite_d(p[i]);
    double q[4];
    for (int r = 0; r < 3; ++r)
        q[r] = ((P[r * 4 + 0] * p[0] + P[r * 4 + 1] * p[1]) + P[r * 4 + 2] * p[2]) + P[r * 4 + 3] * p[3];
    q[3] = p[3];
    if (__builtin_signbit(q[3]))
        for (int i = 0; i < 4; ++i) q[i] = -q[i];
    double qn = AKZ_RM_SQRT(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    for (int i = 0; i < 4; ++i) q[i] = q[i] / qn;
    const double ad = (a[0] * p[0] + a[1] * p[1]) + a[2] * p[2];
    const double bd = (b[0] * q[0] + b[1] * q[1]) + b[2] * q[2];
    const double res = 0.5 * (1.0 - ad + 1.0 - bd);
    const double adm = -ad, bdm = -bd;
    const double resm = 0.5 * (1.0 - adm + 1.0 - bdm);
    *res_out = (fin && res == res) ? res : 2.0;
    *res_mirror = (fin && resm == resm) ? resm : 2.0;
}

// A (pose, match) pair that cannot be an inlier, decided without the eigen-decomposition.  Whatever point p a
// triangulation returns, the rays to p from the two cameras and the baseline t lie in one plane through t, so the angular
// errors alpha (view a) and beta (view b) are at least the angles of f0 = R a and f1 = b to that plane, and over all planes
// through t that sum is smallest at one of the rays' own epipolar planes:
//     alpha + beta >= S,   sin S = |t . (f0 x f1)| / max(|f0 x t|, |f1 x t|)
// (the closed-form L1-optimal two-view triangulation of Lee & Civera, ICCV 2019; DESIGN.md 7 has the three-line proof).
// CameraToCamera::residual = ((1 - cos alpha) + (1 - cos beta)) / 2
// >= 1 - cos(S / 2) >= 0.122 sin^2 S  on [0, pi / 2].  So 0.122 num^2 > thresh' den^2 proves residual >= thresh for [R | t]
// AND its mirror [R | -t] (S does not see the sign of t); thresh' carries a margin of 1e-6 relative + 1e-13 absolute, a
// thousand times the rounding of either side, and the bound is used only for bearings that are unit vectors to 4e-15
// (calibrated ones are) and poses whose R is orthonormal to 1e-9.  A NaN anywhere makes the comparison false: the exact
// statement decides.
__device__ __forceinline__ bool rs_pair_far(const double* __restrict__ pose, const double* a, const double* b, double thresh)
{
    const double na = (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2], nb = (b[0] * b[0] + b[1] * b[1]) + b[2] * b[2];
    if (!(fabs(na - 1.0) <= 4e-15 && fabs(nb - 1.0) <= 4e-15)) return false;
    double f[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) f[r] = (pose[r * 4 + 0] * a[0] + pose[r * 4 + 1] * a[1]) + pose[r * 4 + 2] * a[2];
    const double t[3] = {pose[3], pose[7], pose[11]};
    const double c[3] = {f[1] * b[2] - f[2] * b[1], f[2] * b[0] - f[0] * b[2], f[0] * b[1] - f[1] * b[0]};
    const double num = (t[0] * c[0] + t[1] * c[1]) + t[2] * c[2];
    const double u[3] = {f[1] * t[2] - f[2] * t[1], f[2] * t[0] - f[0] * t[2], f[0] * t[1] - f[1] * t[0]};
    const double v[3] = {b[1] * t[2] - b[2] * t[1], b[2] * t[0] - b[0] * t[2], b[0] * t[1] - b[1] * t[0]};
    const double d0 = (u[0] * u[0] + u[1] * u[1]) + u[2] * u[2], d1 = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
    const double den2 = d0 > d1 ? d0 : d1;
    // (the caller has checked that R preserves angles to 1e-9: rs_rotation_checked, bit 1 of the pose's ok word)
    return 0.122 * (num * num) > (thresh * (1.0 + 1e-6) + 1e-13) * den2;
}

__device__ __forceinline__ bool rs_rotation_checked(const double* pose)
{
    bool fine = true;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) {
            const double d = (pose[i] * pose[j] + pose[4 + i] * pose[4 + j]) + pose[8 + i] * pose[8 + j];   // (R^T R)(i, j)
            fine = fine && fabs(d - (i == j ? 1.0 : 0.0)) <= 1e-9;
        }
    return fine;
}

__device__ double rs_residual(const double* __restrict__ pose, const double* a, const double* b)
{
    double r, rm, w;
    rs_residual_core(pose, 1.0, a, b, &r, &rm, &w);
    return r;
}

// residuals of [R | t] (r0) and [R | -t] (r1) — poses p and p + 2 of a hypothesis — from one eigen-decomposition
__device__ void rs_residual_pair(const double* __restrict__ pose, const double* a, const double* b, double* r0, double* r1)
{
    double tsign = 1.0;
    for (int pass = 0; pass < 2; ++pass) {
        double r, rm, w;
        rs_residual_core(pose, tsign, a, b, &r, &rm, &w);
        if (pass == 0) {
            *r0 = r;
            *r1 = rm;
            if (w != 0.0) break;       // (a NaN component is no zero either: both residuals are 2.0 then)
            tsign = -1.0;              // the eigenvector's last component is exactly zero: the mirrored pose directly
        } else {
            *r1 = r;
        }
    }
}

// ---- the scene arena as the kernels see it ---------------------------------------------------------------------
// Scene s owns slot s of every array: n_cap matches, H hypothesis slots (4 poses each).
struct RsB {
    uint32_t n_cap, H;
    const uint32_t* n;          // [S] matches of the scene
    const double* a;            // [S][n_cap][3] bearings of view a (P3P: the bearings)
    const double* b;            // [S][n_cap][4] slots: bearings of view b packed [n][3], or world points [n][4]
    const uint32_t* order;      // [S][n_cap] scoring order (position -> match), or nullptr = identity
    uint32_t* samples;          // [S][H][K]   (slot stride 8 H)
    double* poses;              // [S][H][4][12]
    uint32_t* ok;               // [S][4H] validity per pose
    uint32_t* counts;           // [S][4H] inliers among the matches scored so far
    uint32_t* alive;            // [S][4H] live poses in ascending id order
    uint32_t* nalive;           // [S]
    unsigned long long* neval;  // [S] residuals evaluated
    uint32_t* best;             // [S][4]: pose id, its count, live poses
    uint32_t* inl;              // [S][n_cap] inlier list (re-sampling rounds / single-scene output)
    uint32_t* ninl;             // [S]
    double* best_pose;          // [S][12]
    uint32_t* first;            // [S] first slot a re-sampling round appended
    uint32_t* enable;           // [S] the round drew samples (enough inliers)
    __device__ __forceinline__ const double* sa(uint32_t s) const { return a + (size_t)s * n_cap * 3; }
    __device__ __forceinline__ const double* sb(uint32_t s) const { return b + (size_t)s * n_cap * 4; }
    __device__ __forceinline__ uint32_t* ssamples(uint32_t s) const { return samples + (size_t)s * H * 8; }
    __device__ __forceinline__ double* sposes(uint32_t s) const { return poses + (size_t)s * H * 48; }
    __device__ __forceinline__ size_t p4(uint32_t s) const { return (size_t)s * H * 4; }
};

// ---- exhaustive scoring of caller-provided samples (rs_p3p_batch: one scene, slot 0; rs_essential_batch runs
// k_rsb_score_first over all matches) --------
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
        inl = rs_w2c_inlier(pose, b, w, thresh);
    }
    unsigned long long bal = __ballot(inl);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&counts[pid], (uint32_t)__popcll(bal));
}

// every valid pose joins the live list (exhaustive scoring keeps them all): the live-list kernels then pick the winner
__global__ __launch_bounds__(1024) void k_rsb_alive_init(RsB B, uint32_t n_pose)
{
    __shared__ uint32_t s_wave[16];
    const uint32_t s = blockIdx.x;
    const uint32_t* ok = B.ok + B.p4(s);
    uint32_t* alive = B.alive + B.p4(s);
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
    if (threadIdx.x == 0) B.nalive[s] = base;
}

// ---- ARRSAC-shaped consensus (row R4) ---------------------------------------------------
// arrsac::Arrsac::model_inliers (un-vendored crate, arrsac 0.10; call sites vslam-sandbox/src/main.rs:105-117,
// cv-sfm/src/lib.rs:1394-1412) scores its hypotheses breadth-first, block of matches by block of matches, and drops
// the ones that can no longer win.  The same shape on the device, every kernel over all scenes of the call:
//   k_rsb_prepare      (micro-batch entry) calibrate the matcher's pairs into bearings, optional seeded shuffle order
//   k_rsb_sample       xoshiro256++ minimal samples drawn on the device (the caller need not ship n_hyp x 8 indices)
//   k_rsb_hypotheses   one lane per minimal sample: essential matrix + four poses, registers only
//   k_rsb_score        every live pose against the next `block` matches (2^lg lanes per pose; k_rsb_score_p3p: a lane per pose)
//   k_rsb_prune        after each block: the best count so far B, then a pose is retired when
//                        (bound)  count + matches_left < B            — it cannot reach the best: exact, always on
//                        (cap)    it is not among the max_candidates best after the initialisation blocks
//                        (SPRT)   its likelihood ratio (delta/eps)^c ((1-delta)/(1-eps))^(seen-c) exceeds the threshold,
//                                 eps = B / seen (Wald's test as in SPRT-RANSAC; arrsac's likelihood_ratio_threshold)
//                      and the survivors are compacted in place, ascending pose order (device-side count, no host hop).
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
// scene s of a batched call draws from its own seed (scene 0 = the caller's seed: a one-scene call is the batch of one)
__host__ __device__ __forceinline__ unsigned long long rs_scene_seed(unsigned long long seed, uint32_t s)
{
    return seed + 0x9E3779B97F4A7C15ull * (unsigned long long)s;
}
// key of match j in the seeded shuffle of a scene's matches (cv-sfm/src/lib.rs:1385 shuffles them with the caller's
// rng before the consensus): the scoring order is the stable ascending sort of these 32-bit keys
__host__ __device__ __forceinline__ uint32_t rs_shuffle_key(unsigned long long scene_seed, uint32_t j)
{
    unsigned long long x = (scene_seed ^ 0x5851F42D4C957F2Dull) + 0xD1342543DE82EF95ull * (unsigned long long)j;
    return (uint32_t)(xo_splitmix(&x) >> 32);
}

template <int K>
__global__ __launch_bounds__(256) void k_rsb_sample(RsB B, unsigned long long seed, uint32_t n_hyp)
{
    const uint32_t s = blockIdx.y;
    const uint32_t h = blockIdx.x * 256 + threadIdx.x;
    const uint32_t n = B.n[s];
    if (h >= n_hyp || n < (uint32_t)K) return;
    uint32_t smp[K];
    rs_draw_sample<K>(rs_scene_seed(seed, s), h, n, smp);
    uint32_t* out = B.ssamples(s) + (size_t)h * K;
    for (int i = 0; i < K; ++i) out[i] = smp[i];
}

// one lane per hypothesis h0 + i of scene blockIdx.y.  gate (optional, per scene): a re-sampling round that drew nothing.
__global__ __launch_bounds__(64) void k_rsb_hypotheses(RsB B, uint32_t h0, uint32_t nh, const uint32_t* __restrict__ gate)
{
    const uint32_t s = blockIdx.y;
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= nh) return;
    const uint32_t hh = h0 + i;
    uint32_t* ok = B.ok + B.p4(s) + (size_t)hh * 4;
    if (B.n[s] < 8u || (gate && !gate[s])) {
        for (int p = 0; p < 4; ++p) ok[p] = 0u;
        return;
    }
    uint32_t smp[8];
    const uint32_t* sp = B.ssamples(s) + (size_t)hh * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) smp[k] = sp[k];
    double P[48];
    const bool good = rs_eight_point_poses(B.sa(s), B.sb(s), smp, P);
    double* out = B.sposes(s) + (size_t)hh * 48;
#pragma unroll
    for (int k = 0; k < 48; ++k) out[k] = P[k];
    // validity per pose; bit 1: the pose's R is orthonormal to 1e-9 (what rs_pair_far's angle argument needs: a product of
    // Jacobi rotations is, but nothing here depends on the SVD's third column having come out that way)
    for (int p = 0; p < 4; ++p) ok[p] = good ? (rs_rotation_checked(P + p * 12) ? 3u : 1u) : 0u;
}

// ---- PnP: Lambda Twist hypotheses (row R5), same indexing; bearings in a, world points [n][4] in b ----
__global__ __launch_bounds__(64) void k_rsb_p3p_hypotheses(RsB B, uint32_t h0, uint32_t nh, const uint32_t* __restrict__ gate)
{
    const uint32_t s = blockIdx.y;
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= nh) return;
    const uint32_t hh = h0 + i;
    uint32_t* ok = B.ok + B.p4(s) + (size_t)hh * 4;
    if (B.n[s] < 3u || (gate && !gate[s])) {
        for (int p = 0; p < 4; ++p) ok[p] = 0u;
        return;
    }
    const double* bearings = B.sa(s);
    const double* world = B.sb(s);
    const uint32_t* sp = B.ssamples(s) + (size_t)hh * 3;
    double b3[9], w3[12], P[48];
    for (int k = 0; k < 3; ++k) {
        const uint32_t m = sp[k];
        for (int q = 0; q < 3; ++q) b3[3 * k + q] = bearings[(size_t)3 * m + q];
        for (int q = 0; q < 4; ++q) w3[4 * k + q] = world[(size_t)4 * m + q];
    }
    const int np = akz_p3p_poses(b3, w3, 5, P);  // LambdaTwist::default(): 5 Gauss-Newton iterations
    double* out = B.sposes(s) + (size_t)hh * 48;
    for (int p = 0; p < 4; ++p) {
        // (bit 1: R^T R = I to 1e-9 — what k_rsb_score_p3p's fused pre-test needs to bound the terms of [R | t] w)
        ok[p] = p < np ? (rs_rotation_checked(P + p * 12) ? 3u : 1u) : 0u;
        if (p < np)
            for (int k = 0; k < 12; ++k) out[p * 12 + k] = P[p * 12 + k];
    }
}

// Block scoring of the two-view consensus.  Scene = blockIdx.z; a pose takes 2^lg lanes (one match each), a wave 64 >> lg poses: a 16-match
// block of the initialisation round keeps every lane busy on four poses instead of a quarter of them on one.
// Positions [m_lo, min(m_hi, n_s)) of the scene's scoring order; blockIdx.y strides the positions of long blocks.
// from_first: only the slots a re-sampling round appended (catching up on the matches seen so far).
__global__ __launch_bounds__(256) void k_rsb_score(RsB B, uint32_t m_lo, uint32_t m_hi, uint32_t lg, uint32_t from_first,
                                                   double thresh)
{
    const uint32_t s = blockIdx.z;
    const uint32_t n = B.n[s];
    const uint32_t hi = m_hi < n ? m_hi : n;
    if (m_lo >= hi) return;
    const uint32_t G = 1u << lg, lane = threadIdx.x & 63u, g = lane >> lg, j = lane & (G - 1u);
    const uint32_t first = from_first ? B.first[s] : 0u;
    const uint32_t nal = B.nalive[s];
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave == 0 && lane == 0 && blockIdx.y == 0 && nal > first)
        atomicAdd(&B.neval[s], (unsigned long long)(nal - first) * (unsigned long long)(hi - m_lo));
    const uint32_t slot0 = first + (wave << (6u - lg));
    if (slot0 >= nal) return;
    const uint32_t slot = slot0 + g;
    const bool live = slot < nal;
    const uint32_t pid = live ? B.alive[B.p4(s) + slot] : 0u;
    const bool rot_ok = live && (B.ok[B.p4(s) + pid] & 2u);
    double pose[12];
    const double* pp = B.sposes(s) + (size_t)pid * 12;
    for (int i = 0; i < 12; ++i) pose[i] = live ? pp[i] : 0.0;
    const double* ba = B.sa(s);
    const double* bb = B.sb(s);
    const uint32_t* order = B.order ? B.order + (size_t)s * B.n_cap : nullptr;
    const unsigned long long gmask = (lg == 6u ? ~0ull : ((1ull << G) - 1ull)) << (g << lg);
    uint32_t cnt = 0;
    for (uint32_t m0 = m_lo + blockIdx.y * G; m0 < hi; m0 += gridDim.y * G) {
        const uint32_t pos = m0 + j;
        bool inl = false;
        if (live && pos < hi) {
            const uint32_t m = order ? order[pos] : pos;
            double a[3] = {ba[3 * (size_t)m], ba[3 * (size_t)m + 1], ba[3 * (size_t)m + 2]};
            double b[3] = {bb[3 * (size_t)m], bb[3 * (size_t)m + 1], bb[3 * (size_t)m + 2]};
            inl = !(rot_ok && rs_pair_far(pose, a, b, thresh)) && rs_residual(pose, a, b) < thresh;
        }
        cnt += (uint32_t)__popcll(__ballot(inl) & gmask);
    }
    if (live && j == 0 && cnt) atomicAdd(&B.counts[B.p4(s) + pid], cnt);
}

// Block scoring of the registration consensus.  WorldToCamera::residual is a dozen multiplications per (pose, match), so
// the match-per-lane organisation above spends its time gathering 56 bytes of match per lane; here a LANE is a live pose
// (its 12 values in registers), the workgroup stages the matches of its part of [m_lo, m_hi) in LDS once and every lane
// walks them (broadcast reads), counting in a register: one atomic per pose and launch.  blockIdx.y splits long ranges.
constexpr uint32_t kP3PTile = 256;   // matches staged per pass (8 doubles each: sign-folded bearing, guard, world point)
// The pre-test of rs_w2c_inlier with a FUSED q = [R | t] w (12 instructions instead of 21; 27 instead of 39 a residual) —
// allowed to decide when
//   * the pose's R is certified (bit 1 of its ok word, rs_rotation_checked): |R_ri| <= 1 + 1e-9, so the terms of row r of
//     [R | t] w sum to at most 2 W3 + |q_r| in magnitude, W3 = |w0| + |w1| + |w2| (because |t_r w3| <= |q_r| + |R_r . w|), and
//     the fused and the unfused q_r — each within 4 ulp of that sum of the true value — differ by at most 8 eps (2 W3 + |q_r|);
//   * there is no deep cancellation: s2 = |q|^2 >= 1.6e-9 W3^2, i.e. |q| >= 4e-5 W3 (the match's `guard`).  Then c^2 / s2 of
//     the fused q is within 56 eps (4 W3 / |q| + 1) < 1e-9 of the unfused one's, and a decision taken outside a band of
//     +-6e-9 around u^2 = (1 - thresh)^2 is the decision rs_w2c_inlier's own pre-test takes outside its +-4e-9 band — which
//     is the exact statement's (its header).  A sign of c that differs between the two means |c| <= 1e-9 |q|: an outlier either way.
// Everything else — an uncertified R, a threshold outside (0, 1/2), s2 below the guard or not an ordinary number, a value
// inside the band — goes to rs_w2c_inlier.  The result is rs_w2c_inlier's, hence the exact statement's, for every input
// (tests/test_gpu_parity.py::test_block_scoring_pre_test_is_exact_at_the_threshold).
// One pose a lane.  Measured on a 256-frame registration step (16 384 hypotheses, 256 estimations per 64-match block, ~4 150
// matches a frame; profiles/r06_register_*): the catch-up launches of the re-sampling rounds — up to 1 024 new poses against
// every match seen so far, 39 of the consensus' 52 ms — run at 0.72 of the f64 issue rate (1.07e9 residuals in 1 030 us at
// the last block).  Two poses a lane halve the LDS reads per residual and make the first block (65 536 poses a scene) 30 %
// faster but leave the second workgroup of a catch-up launch half empty: 339 ms against 301 ms of scoring per six steps.
// The transpose (lane = match, pose by scalar loads into SGPR operands) is bound by the scalar loads' latency: 376-407 ms.
// (The compiler's form of this loop matters as much: the exact statement below must stay a BRANCH — flattened into
// selects, which happened when the same source was written over an array of poses per lane, every residual pays the square
// root and the three divisions, 1 104 ms — and the loop must be unrolled twice to keep the next match's LDS reads in
// flight: a convergent vote or a volatile asm in the branch forbids that, 330-383 ms.  tools/check_isa.py holds both.)
__global__ __launch_bounds__(256) void k_rsb_score_p3p(RsB B, uint32_t m_lo, uint32_t m_hi, uint32_t from_first, double thresh)
{
    __shared__ __attribute__((aligned(16))) double s_m[kP3PTile][8];
    const uint32_t s = blockIdx.z;
    const uint32_t n = B.n[s];
    const uint32_t hi = m_hi < n ? m_hi : n;
    if (m_lo >= hi) return;
    const uint32_t first = from_first ? B.first[s] : 0u;
    const uint32_t nal = B.nalive[s];
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && nal > first)
        atomicAdd(&B.neval[s], (unsigned long long)(nal - first) * (unsigned long long)(hi - m_lo));
    const uint32_t slot0 = first + blockIdx.x * 256u;
    if (slot0 >= nal) return;   // whole workgroup
    const uint32_t part = (hi - m_lo + gridDim.y - 1) / gridDim.y;
    const uint32_t lo = m_lo + blockIdx.y * part;
    const uint32_t hi_p = lo + part < hi ? lo + part : hi;
    if (lo >= hi_p) return;
    const uint32_t slot = slot0 + threadIdx.x;
    const bool live = slot < nal;
    const uint32_t pid = live ? B.alive[B.p4(s) + slot] : 0u;
    double pose[12];
    const double* pp = B.sposes(s) + (size_t)pid * 12;
#pragma unroll
    for (int i = 0; i < 12; ++i) pose[i] = live ? pp[i] : 0.0;
    const double u = 1.0 - thresh, u2 = u * u, k_in6 = u2 * (1.0 + 6e-9), k_out6 = u2 * (1.0 - 6e-9);
    const bool fused_ok = live && thresh > 0.0 && thresh < 0.5 && (B.ok[B.p4(s) + pid] & 2u);
    const double* ba = B.sa(s);
    const double* bb = B.sb(s);
    const uint32_t* order = B.order ? B.order + (size_t)s * B.n_cap : nullptr;
    uint32_t cnt = 0;
    for (uint32_t t0 = lo; t0 < hi_p; t0 += kP3PTile) {
        const uint32_t tn = hi_p - t0 < kP3PTile ? hi_p - t0 : kP3PTile;
        __syncthreads();   // the previous tile has been read
        if (threadIdx.x < tn) {
            const uint32_t m = order ? order[t0 + threadIdx.x] : t0 + threadIdx.x;
            double* d = s_m[threadIdx.x];
            const double w0 = bb[4 * (size_t)m], w1 = bb[4 * (size_t)m + 1], w2 = bb[4 * (size_t)m + 2], w3 = bb[4 * (size_t)m + 3];
            const bool flip = __builtin_signbit(w3);                  // (-a) . q = -(a . q) exactly: the sign goes into the bearing
            const double a0 = ba[3 * (size_t)m], a1 = ba[3 * (size_t)m + 1], a2 = ba[3 * (size_t)m + 2];
            const double W3 = (fabs(w0) + fabs(w1)) + fabs(w2), g0 = 1.6e-9 * W3 * W3;
            d[0] = flip ? -a0 : a0; d[1] = flip ? -a1 : a1; d[2] = flip ? -a2 : a2;
            d[3] = g0 > 1e-280 ? g0 : 1e-280;                         // the guard (and the ordinary-number floor of s2)
            d[4] = w0; d[5] = w1; d[6] = w2; d[7] = w3;
        }
        __syncthreads();
        if (live) {
#pragma unroll 2
            for (uint32_t i = 0; i < tn; ++i) {
                const double2 a01 = *reinterpret_cast<const double2*>(&s_m[i][0]);
                const double2 a2g = *reinterpret_cast<const double2*>(&s_m[i][2]);
                const double2 w01 = *reinterpret_cast<const double2*>(&s_m[i][4]);
                const double2 w23 = *reinterpret_cast<const double2*>(&s_m[i][6]);
                const double w[4] = {w01.x, w01.y, w23.x, w23.y};
                bool inl = false, decided = false;
                if (fused_ok) {
                    double q[3];
#pragma unroll
                    for (int r = 0; r < 3; ++r)
                        q[r] = __builtin_fma(pose[r * 4 + 0], w[0], __builtin_fma(pose[r * 4 + 1], w[1],
                                             __builtin_fma(pose[r * 4 + 2], w[2], pose[r * 4 + 3] * w[3])));
                    const double s2 = __builtin_fma(q[2], q[2], __builtin_fma(q[1], q[1], q[0] * q[0]));
                    const double c = __builtin_fma(a2g.x, q[2], __builtin_fma(a01.y, q[1], a01.x * q[0]));
                    const double c2 = c * c;
                    const bool yes = c > 0.0 && c2 > k_in6 * s2;
                    const bool no = c <= 0.0 || c2 < k_out6 * s2;               // (a NaN fails every comparison)
                    decided = s2 >= a2g.y && s2 < 1e280 && (yes || no);
                    inl = yes;
                }
                if (!decided) {
                    const bool flip = __builtin_signbit(w[3]);
                    const double a[3] = {flip ? -a01.x : a01.x, flip ? -a01.y : a01.y, flip ? -a2g.x : a2g.x};
                    inl = rs_w2c_inlier(pose, a, w, thresh);
                }
                cnt += inl ? 1u : 0u;
            }
        }
    }
    if (live && cnt) atomicAdd(&B.counts[B.p4(s) + pid], cnt);
}

// The first block of a call: nothing has been retired yet, so the live list is every valid pose and the four poses of
// a hypothesis are all in it.  Units of work are (hypothesis, R1 | R2): [R | t] and [R | -t] of a unit are scored from one
// eigen-decomposition per match (rs_residual_pair) — half the arithmetic of scoring the four poses one by one.
// Most units are poses of contaminated samples, for which rs_pair_far discards nearly every match; a lane-per-pair layout
// would leave those lanes idle beside the few that do need the eigen-decomposition.  A workgroup therefore owns 256 units
// and works in two phases per tile of 16 matches (staged in LDS):
//   1. lane <-> unit (pose in registers, matches by broadcast reads): the far test of the unit against the 16 matches
//      -> a 16-bit mask; the surviving (unit, match) pairs are queued in LDS (positions by a workgroup prefix sum);
//   2. lane <-> queued pair: the eigen-decomposition, with every lane busy; inliers counted in LDS per unit.
// One atomic per pose at the end.  With vslam-sandbox's parameters on a 10 %-outlier batch 43 % of the pairs reach phase 2,
// with 30 % outliers 6 %.
constexpr uint32_t kFirstUnits = 256, kFirstTile = 16;
__global__ __launch_bounds__(256) void k_rsb_score_first(RsB B, uint32_t m_hi, uint32_t n_hyp, double thresh)
{
    __shared__ __attribute__((aligned(16))) double s_pose[kFirstUnits][12];
    __shared__ __attribute__((aligned(16))) double s_m[kFirstTile][6];
    __shared__ unsigned short s_pair[kFirstUnits * kFirstTile];
    __shared__ uint32_t s_cnt[kFirstUnits][2];
    __shared__ uint32_t s_wsum[4];
    const uint32_t s = blockIdx.z;
    const uint32_t n = B.n[s];
    const uint32_t hi = m_hi < n ? m_hi : n;
    if (hi == 0) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t nal = B.nalive[s];
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && nal) atomicAdd(&B.neval[s], (unsigned long long)nal * (unsigned long long)hi);
    // blockIdx.y takes a share of the tiles (long match lists of a single scene: rs_essential_batch)
    const uint32_t n_tiles = (hi + kFirstTile - 1) / kFirstTile, part = (n_tiles + gridDim.y - 1) / gridDim.y;
    const uint32_t m_begin = blockIdx.y * part * kFirstTile;
    const uint32_t m_end = m_begin + part * kFirstTile < hi ? m_begin + part * kFirstTile : hi;
    if (m_begin >= m_end) return;
    const uint32_t unit = blockIdx.x * kFirstUnits + tid;
    const uint32_t pid = (unit >> 1) * 4u + (unit & 1u);
    const uint32_t okw = unit < 2u * n_hyp ? B.ok[B.p4(s) + pid] : 0u;
    double pose[12];
    {
        const double* pp = B.sposes(s) + (size_t)pid * 12;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            pose[i] = okw ? pp[i] : 0.0;
            s_pose[tid][i] = pose[i];
        }
    }
    s_cnt[tid][0] = 0u;
    s_cnt[tid][1] = 0u;
    const double* ba = B.sa(s);
    const double* bb = B.sb(s);
    const uint32_t* order = B.order ? B.order + (size_t)s * B.n_cap : nullptr;
    for (uint32_t t0 = m_begin; t0 < m_end; t0 += kFirstTile) {
        const uint32_t tn = m_end - t0 < kFirstTile ? m_end - t0 : kFirstTile;
        __syncthreads();   // the previous tile's pairs have been scored; s_pose / s_cnt initialised
        if (tid < tn) {
            const uint32_t m = order ? order[t0 + tid] : t0 + tid;
            s_m[tid][0] = ba[3 * (size_t)m]; s_m[tid][1] = ba[3 * (size_t)m + 1]; s_m[tid][2] = ba[3 * (size_t)m + 2];
            s_m[tid][3] = bb[3 * (size_t)m]; s_m[tid][4] = bb[3 * (size_t)m + 1]; s_m[tid][5] = bb[3 * (size_t)m + 2];
        }
        __syncthreads();
        // ---- phase 1: which of the tile's matches can be inliers of this lane's unit at all ----
        uint32_t mask = 0u;
        if (okw & 2u) {
            for (uint32_t i = 0; i < tn; ++i) {
                const double a[3] = {s_m[i][0], s_m[i][1], s_m[i][2]};
                const double b[3] = {s_m[i][3], s_m[i][4], s_m[i][5]};
                mask |= rs_pair_far(pose, a, b, thresh) ? 0u : (1u << i);
            }
        } else if (okw) {
            mask = (1u << tn) - 1u;   // a valid pose whose R was not certified: every match goes to the exact statement
        }
        // queue positions: exclusive prefix sum of the lanes' pair counts over the workgroup
        const uint32_t mine = (uint32_t)__popc(mask);
        uint32_t incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off);
            if (lane >= (uint32_t)off) incl += o;
        }
        if (lane == 63u) s_wsum[wv] = incl;
        __syncthreads();
        uint32_t base = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4; ++w) {
            base += w < wv ? s_wsum[w] : 0u;
            total += s_wsum[w];
        }
        uint32_t at = base + incl - mine;
        for (uint32_t mk = mask; mk; mk &= mk - 1u) s_pair[at++] = (unsigned short)((tid << 4) | (uint32_t)(__ffs((int)mk) - 1));
        __syncthreads();
        // ---- phase 2: the eigen-decomposition of every queued pair, one per lane ----
        for (uint32_t q = tid; q < total; q += 256) {
            const uint32_t e = s_pair[q], ul = e >> 4, i = e & 15u;
            double ps[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) ps[k] = s_pose[ul][k];
            const double a[3] = {s_m[i][0], s_m[i][1], s_m[i][2]};
            const double b[3] = {s_m[i][3], s_m[i][4], s_m[i][5]};
            double r0, r1;
            rs_residual_pair(ps, a, b, &r0, &r1);
            if (r0 < thresh) atomicAdd(&s_cnt[ul][0], 1u);
            if (r1 < thresh) atomicAdd(&s_cnt[ul][1], 1u);
        }
    }
    __syncthreads();
    if (okw) {
        if (s_cnt[tid][0]) atomicAdd(&B.counts[B.p4(s) + pid], s_cnt[tid][0]);
        if (s_cnt[tid][1]) atomicAdd(&B.counts[B.p4(s) + pid + 2], s_cnt[tid][1]);
    }
}

struct RsPrune {
    uint32_t seen;                // matches scored so far (the same position for every scene still running)
    uint32_t cap;                 // keep at most this many poses from now on (0 = no cap)
    uint32_t use_sprt, pad;
    double log_delta, log_1m_delta, log_ratio;   // ln(delta), ln(1 - delta), ln(likelihood ratio threshold)
    const double* log_table;      // ln(k), k = 0 .. n_cap, filled by the host's libm (no device transcendental decides)
};

// One workgroup per scene.  The cap ranks the poses by their exact distance to the best count: a 2048-bin histogram of
// best - count (poses 2047 or more inliers behind the best share the last bin), so the best-supported poses are never
// the ones the cap retires; ties at the threshold are admitted in ascending pose order while the budget lasts.
// Threads of the one-workgroup-per-scene kernels that sit in the block loop's chain of launches (prune, append, best inliers).
// Four waves, not sixteen: while the matcher and the extraction fill the chip, a workgroup that needs sixteen free wave slots on
// ONE compute unit waits for them at every launch of the chain (the registration leg of the bench: 122.4-122.8 ms per step at
// 1 024 threads, 120.0-120.2 at 512, 119.9-120.0 at 256; loading more per pass at twice the registers made it 140).
constexpr int kChainNT = 256;
__global__ __launch_bounds__(kChainNT) void k_rsb_prune(RsB B, RsPrune P)
{
    __shared__ uint32_t s_wave[16], s_wave_t[16];
    __shared__ uint32_t s_hist[2048];
    __shared__ uint32_t s_best, s_T, s_budget;
    const uint32_t s = blockIdx.x;
    const uint32_t n_total = B.n[s];
    if (P.seen >= n_total) return;            // no retirement after a scene's last block (nor for scenes that ended earlier)
    const uint32_t* counts = B.counts + B.p4(s);
    uint32_t* alive = B.alive + B.p4(s);
    const uint32_t n = B.nalive[s];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_best = 0;
    for (int i = threadIdx.x; i < 2048; i += kChainNT) s_hist[i] = 0;
    __syncthreads();
    uint32_t lmax = 0;
    for (uint32_t i = threadIdx.x; i < n; i += kChainNT) {
        const uint32_t c = counts[alive[i]];
        lmax = c > lmax ? c : lmax;
    }
    atomicMax(&s_best, lmax);
    __syncthreads();
    const uint32_t best = s_best, left = n_total - P.seen;
    const bool capped = P.cap && n > P.cap;
    if (capped) {
        for (uint32_t i = threadIdx.x; i < n; i += kChainNT) {
            const uint32_t d = best - counts[alive[i]];
            atomicAdd(&s_hist[d < 2047u ? d : 2047u], 1u);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // distance threshold of the cap: poses closer than T to the best all stay, those at T in pose order up to the budget
        uint32_t T = 0xFFFFFFFFu, budget = 0xFFFFFFFFu;
        if (capped) {
            uint32_t acc = 0;
            uint32_t t = 0;
            for (; t < 2048u; ++t) {
                if (acc + s_hist[t] >= P.cap) break;
                acc += s_hist[t];
            }
            T = t < 2048u ? t : 2047u;
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
    for (uint32_t i0 = 0; i0 < n; i0 += kChainNT) {
        const uint32_t i = i0 + threadIdx.x;
        bool keep = false, tie = false;
        uint32_t pid = 0;
        if (i < n) {
            pid = alive[i];
            const uint32_t c = counts[pid];
            keep = c + left >= best;                                                   // bound (exact)
            if (keep && P.use_sprt && l_out > 0.0)
                keep = (double)c * l_in + (double)(P.seen - c) * l_out <= P.log_ratio || c == best;
            const uint32_t d = best - c, dd = d < 2047u ? d : 2047u;
            if (keep && capped) {
                if (dd > T) keep = false;
                tie = keep && dd == T;
            }
        }
        // ties at the cap threshold are admitted in pose order while the budget lasts
        const unsigned long long tb = __ballot(tie);
        if (lane == 0) s_wave_t[wv] = (uint32_t)__popcll(tb);
        __syncthreads();
        uint32_t toff = 0, ttot = 0;
        for (int q = 0; q < kChainNT / 64; ++q) {
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
        for (int q = 0; q < kChainNT / 64; ++q) {
            if (q < wv) woff += s_wave[q];
            tot += s_wave[q];
        }
        // in place: every read of this 1024-slot step happened before the barriers above, and a survivor moves to a
        // slot at or before its own
        if (keep) alive[base + woff + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = pid;
        base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) B.nalive[s] = base;
}

// inlier-guided re-sampling (arrsac's estimations_per_block): E new minimal samples drawn among the inliers (over the
// matches seen so far, list inl[] of length ninl) of the best pose; disabled (enable 0) with fewer than K inliers
template <int K>
__global__ __launch_bounds__(256) void k_rsb_resample(RsB B, unsigned long long seed, uint32_t next_h, uint32_t E)
{
    const uint32_t s = blockIdx.y;
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    const uint32_t n = B.ninl[s];
    if (e == 0) B.enable[s] = n >= (uint32_t)K ? 1u : 0u;
    if (e >= E || n < (uint32_t)K) return;
    const uint32_t h = next_h + e;
    const uint32_t* L = B.inl + (size_t)s * B.n_cap;
    uint32_t smp[K];
    rs_draw_sample<K>(rs_scene_seed(seed, s) ^ 0xA5A5A5A55A5A5A5Aull, h, n, smp);
    uint32_t* out = B.ssamples(s) + (size_t)h * K;
    for (int i = 0; i < K; ++i) out[i] = L[smp[i]];
}

// valid new poses join the live list in (hypothesis, pose) order — their ids exceed every id already in it — with
// zeroed counters; first[s] receives the list length before the append
__global__ __launch_bounds__(kChainNT) void k_rsb_alive_append(RsB B, uint32_t base_pid, uint32_t n_new)
{
    __shared__ uint32_t s_wave[16];
    const uint32_t s = blockIdx.x;
    const uint32_t* ok = B.ok + B.p4(s);
    uint32_t* alive = B.alive + B.p4(s);
    uint32_t* counts = B.counts + B.p4(s);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t old = B.nalive[s], on = B.enable[s];
    uint32_t base = old;
    for (uint32_t i0 = 0; i0 < n_new; i0 += kChainNT) {
        const uint32_t i = i0 + threadIdx.x;
        const bool keep = on && i < n_new && ok[base_pid + i] != 0;
        if (i < n_new) counts[base_pid + i] = 0u;
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_wave[wv] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t woff = 0, tot = 0;
        for (int q = 0; q < kChainNT / 64; ++q) {
            if (q < wv) woff += s_wave[q];
            tot += s_wave[q];
        }
        if (keep) alive[base + woff + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = base_pid + i;
        base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        B.first[s] = old;
        B.nalive[s] = base;
    }
}

// where the results of a call go (device pointers; element s of each array belongs to scene s)
struct RsOut {
    uint32_t* best_id;        // [S]
    double* pose;             // [S][12]
    uint32_t* inl;            // [S][inl_stride]
    uint32_t* ninl;           // [S]
    rs_arrsac_stats* stats;   // [S] or nullptr
    uint32_t inl_stride;
    uint32_t n_hyp, resample, init_blocks;   // initial hypotheses, estimations_per_block, init_blocks of the call
    uint32_t blocks_run, block_size, min_samples;
};

// One workgroup per scene: the best live pose — argmax of (count, -id) — and its inliers, ordered compaction.
//   final = 0  re-sampling round after `limit` matches: inliers among the positions [0, limit) of the scoring order, in
//              that order, into the arena's own list; scenes with no match left take no part (ninl = 0)
//   final = 1  the answer: inliers over all matches in ascending match index, pose, id, stats -> O
template <bool P3P>
__global__ __launch_bounds__(kChainNT) void k_rsb_best_inliers(RsB B, uint32_t limit, uint32_t final_, double thresh, RsOut O)
{
    __shared__ unsigned long long s_key[16];
    __shared__ uint32_t s_wave[16];
    const uint32_t s = blockIdx.x;
    const uint32_t n = B.n[s];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t* out_inl = final_ ? O.inl + (size_t)s * O.inl_stride : B.inl + (size_t)s * B.n_cap;
    uint32_t* out_n = final_ ? O.ninl + s : B.ninl + s;
    if (!final_ && limit >= n) {
        if (threadIdx.x == 0) *out_n = 0;
        return;
    }
    const uint32_t* counts = B.counts + B.p4(s);
    const uint32_t* alive = B.alive + B.p4(s);
    const uint32_t nal = B.nalive[s];
    unsigned long long key = 0ull;  // 0 = nothing alive
    for (uint32_t i = threadIdx.x; i < nal; i += kChainNT) {
        const uint32_t pid = alive[i];
        const unsigned long long k = ((unsigned long long)(counts[pid] + 1u) << 32) | (unsigned long long)(0xFFFFFFFFu - pid);
        key = k > key ? k : key;
    }
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long o = __shfl_down(key, off);
        key = o > key ? o : key;
    }
    if (lane == 0) s_key[wv] = key;
    __syncthreads();
    key = s_key[0];
    for (int i = 1; i < kChainNT / 64; ++i) key = s_key[i] > key ? s_key[i] : key;
    const uint32_t pid = key ? (0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull)) : 0xFFFFFFFFu;
    if (threadIdx.x == 0) {
        B.best[4 * s + 0] = pid;
        B.best[4 * s + 1] = key ? (uint32_t)(key >> 32) - 1u : 0u;
        B.best[4 * s + 2] = nal;
        if (final_) {
            O.best_id[s] = pid;
            if (O.stats) {
                rs_arrsac_stats st;
                // blocks this scene was scored on, and the re-sampling rounds that followed them (one after every block
                // from init_blocks on that was not the scene's last)
                const uint32_t per = O.block_size ? (n + O.block_size - 1u) / O.block_size : 1u;
                const uint32_t blocks = n >= O.min_samples ? (per < O.blocks_run ? per : O.blocks_run) : 0u;
                const uint32_t from = O.init_blocks > 1u ? O.init_blocks : 1u;
                const uint32_t rounds = blocks > from ? blocks - from : 0u;
                st.poses = n >= O.min_samples ? (O.n_hyp + O.resample * rounds) * 4u : 0u;
                st.survivors = nal;
                st.blocks = blocks;
                st.reserved = 0;
                st.residuals_evaluated = B.neval[s];
                st.residuals_exhaustive = (uint64_t)st.poses * n;
                O.stats[s] = st;
            }
        }
    }
    if (pid == 0xFFFFFFFFu) {
        if (threadIdx.x == 0) *out_n = 0;
        return;
    }
    const uint32_t range = final_ ? n : limit;
    if (threadIdx.x == 0 && !final_) atomicAdd(&B.neval[s], (unsigned long long)range);
    double pose[12];
    const double* pp = B.sposes(s) + (size_t)pid * 12;
    for (int i = 0; i < 12; ++i) pose[i] = pp[i];
    if (final_ && threadIdx.x < 12) O.pose[(size_t)s * 12 + threadIdx.x] = pose[threadIdx.x];
    const double* ba = B.sa(s);
    const double* bb = B.sb(s);
    const uint32_t* order = (!final_ && B.order) ? B.order + (size_t)s * B.n_cap : nullptr;
    uint32_t base = 0;
    for (uint32_t m0 = 0; m0 < range; m0 += kChainNT) {
        const uint32_t pos = m0 + threadIdx.x;
        bool inl = false;
        uint32_t m = 0;
        if (pos < range) {
            m = order ? order[pos] : pos;
            double a[3] = {ba[3 * (size_t)m], ba[3 * (size_t)m + 1], ba[3 * (size_t)m + 2]};
            if (P3P) {
                double w[4] = {bb[4 * (size_t)m], bb[4 * (size_t)m + 1], bb[4 * (size_t)m + 2], bb[4 * (size_t)m + 3]};
                inl = rs_w2c_inlier(pose, a, w, thresh);
            } else {
                double b[3] = {bb[3 * (size_t)m], bb[3 * (size_t)m + 1], bb[3 * (size_t)m + 2]};
                inl = rs_residual(pose, a, b) < thresh;
            }
        }
        unsigned long long bal = __ballot(inl);
        if (lane == 0) s_wave[wv] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t woff = 0, tot = 0;
        for (int q = 0; q < kChainNT / 64; ++q) {
            if (q < wv) woff += s_wave[q];
            tot += s_wave[q];
        }
        if (inl) out_inl[base + woff + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = m;
        base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *out_n = base;
}

// ---- micro-batch entry: the matcher's pair lists -> calibrated bearings (+ the seeded scoring order) ----------
struct RsCam {
    double intr[5];   // fx, fy, cx, cy, skew
    double k1;
    int use_k1, pad;
};
// One workgroup per scene.  Scene s = pair list `s` of the matcher's output ([cap][2] indices into keypoint blocks
// fa[s] of d_kps_a and fb[s] of d_kps_b); FeatureMatch(a, b) of cv-sfm/src/lib.rs:1400-1403 (match_ix_kps).
// P3P: scene s = a list of (feature of keypoint block fa[s], world point) index pairs — FeatureWorldMatch(bearing, point) of
// cv-sfm/src/lib.rs:1571-1590; kps_b is the table of homogeneous world points ([..][4] f64) and cam_b / fb are unused.
template <bool P3P>
__global__ __launch_bounds__(1024) void k_rsb_prepare(RsB B, const akz_keypoint* __restrict__ kps_a, const void* __restrict__ kps_b_or_world,
                                                      uint32_t cap_per_img, const uint32_t* __restrict__ fa, const uint32_t* __restrict__ fb,
                                                      const uint32_t* __restrict__ pairs, const uint32_t* __restrict__ npairs,
                                                      RsCam cam_a, RsCam cam_b, uint32_t* __restrict__ n_out, double* __restrict__ a_out,
                                                      double* __restrict__ b_out, uint32_t* __restrict__ order_out, uint32_t shuffle,
                                                      unsigned long long seed, uint32_t limit_b)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t s = blockIdx.x;
    uint32_t n = npairs[s];
    n = n < cap_per_img ? n : cap_per_img;
    n = n < B.n_cap ? n : B.n_cap;
    const akz_keypoint* ka = kps_a + (size_t)fa[s] * cap_per_img;
    const uint32_t* pr = pairs + (size_t)s * cap_per_img * 2;
    double* ao = a_out + (size_t)s * B.n_cap * 3;
    double* bo = b_out + (size_t)s * B.n_cap * 4;
    // A pair list is the caller's device data (normally hm_match_batch_device's output): an entry that points outside
    // the keypoint block (>= cap_per_img) or the world table (>= limit_b) refuses the whole scene — it ends with "no
    // model", nothing is read out of bounds (oracle/arrsac_oracle.c has the same rule).
    bool bad = false;
    for (uint32_t j = threadIdx.x; j < n; j += 1024) bad = bad || pr[2 * j] >= cap_per_img || pr[2 * j + 1] >= limit_b;
    if (__syncthreads_or(bad)) n = 0;
    if (threadIdx.x == 0) n_out[s] = n;
    for (uint32_t j = threadIdx.x; j < n; j += 1024) {
        const uint32_t ia = pr[2 * j], ib = pr[2 * j + 1];
        double o[3];
        rs_calibrate_one(cam_a.intr, cam_a.use_k1, cam_a.k1, ka[ia].x, ka[ia].y, o);
        ao[3 * j] = o[0]; ao[3 * j + 1] = o[1]; ao[3 * j + 2] = o[2];
        if (P3P) {
            const double* wp = (const double*)kps_b_or_world + (size_t)4 * ib;
            bo[4 * j] = wp[0]; bo[4 * j + 1] = wp[1]; bo[4 * j + 2] = wp[2]; bo[4 * j + 3] = wp[3];
        } else {
            const akz_keypoint* kb = (const akz_keypoint*)kps_b_or_world + (size_t)fb[s] * cap_per_img;
            rs_calibrate_one(cam_b.intr, cam_b.use_k1, cam_b.k1, kb[ib].x, kb[ib].y, o);
            bo[3 * j] = o[0]; bo[3 * j + 1] = o[1]; bo[3 * j + 2] = o[2];
        }
    }
    if (!shuffle) return;
    // scoring order = stable ascending sort of the matches' 32-bit shuffle keys (LSD radix sort in LDS)
    uint32_t* rk = reinterpret_cast<uint32_t*>(smem);
    uint32_t* ia_ = rk + kRadixSortMax;
    uint32_t* ib_ = ia_ + kRadixSortMax;
    uint32_t* wh = ib_ + kRadixSortMax;
    __shared__ uint32_t tot[256];
    const unsigned long long ss = rs_scene_seed(seed, s);
    for (uint32_t j = threadIdx.x; j < n; j += 1024) {
        rk[j] = rs_shuffle_key(ss, j);
        ia_[j] = j;
    }
    __syncthreads();
    const uint32_t* sorted = lds_radix_sort_ids(rk, ia_, ib_, wh, tot, n, 4);
    uint32_t* oo = order_out + (size_t)s * B.n_cap;
    for (uint32_t j = threadIdx.x; j < n; j += 1024) oo[j] = sorted[j];
}

// argmax of (count, -id) over ALL poses of one scene (exhaustive scoring): the pose with the most inliers, lowest id on ties
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

// inlier indices of pose best[0] in ascending match order (ordered compaction, one block; exhaustive entry points)
template <bool P3P>
__global__ __launch_bounds__(1024) void k_rs_inliers(const double* __restrict__ ba, const double* __restrict__ bb,
                                                     uint32_t n, const double* __restrict__ poses,
                                                     const uint32_t* __restrict__ best, double thresh,
                                                     uint32_t* __restrict__ inlier_idx, uint32_t cap,
                                                     uint32_t* __restrict__ n_inliers, double* __restrict__ best_pose)
{
    __shared__ uint32_t s_wave[16];
    const uint32_t pid = best[0];
    if (pid == 0xFFFFFFFFu) {
        if (threadIdx.x == 0) *n_inliers = 0;
        return;
    }
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
            if (P3P) {
                double w[4] = {bb[4 * (size_t)m], bb[4 * (size_t)m + 1], bb[4 * (size_t)m + 2], bb[4 * (size_t)m + 3]};
                inl = rs_w2c_inlier(pose, a, w, thresh);
            } else {
                double b[3] = {bb[3 * (size_t)m], bb[3 * (size_t)m + 1], bb[3 * (size_t)m + 2]};
                inl = rs_residual(pose, a, b) < thresh;
            }
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

// parity tap: the residual of every (pose, match), directly or through the (t, -t) pair path
__global__ __launch_bounds__(256) void k_rs_debug_residuals(const double* __restrict__ poses, uint32_t n_pose, const double* __restrict__ ba,
                                                            const double* __restrict__ bb, uint32_t n, int paired, double* __restrict__ out)
{
    const uint32_t m = blockIdx.x * 256 + threadIdx.x, pid = blockIdx.y;
    if (m >= n) return;
    double pose[12];
    for (int i = 0; i < 12; ++i) pose[i] = poses[(size_t)pid * 12 + i];
    double a[3] = {ba[3 * (size_t)m], ba[3 * (size_t)m + 1], ba[3 * (size_t)m + 2]};
    double b[3] = {bb[3 * (size_t)m], bb[3 * (size_t)m + 1], bb[3 * (size_t)m + 2]};
    if (paired) {
        double r0, r1;
        rs_residual_pair(pose, a, b, &r0, &r1);
        out[((size_t)pid * 2 + 0) * n + m] = r0;
        out[((size_t)pid * 2 + 1) * n + m] = r1;
    } else {
        out[(size_t)pid * n + m] = rs_residual(pose, a, b);
    }
}

// parity tap: rs_pair_far of every (pose, match) — 1 where the bound alone decides "not an inlier" (with the pose's rotation check)
__global__ __launch_bounds__(256) void k_rs_debug_far(const double* __restrict__ poses, uint32_t n_pose, const double* __restrict__ ba,
                                                      const double* __restrict__ bb, uint32_t n, double thresh, unsigned char* __restrict__ out)
{
    const uint32_t m = blockIdx.x * 256 + threadIdx.x, pid = blockIdx.y;
    if (m >= n) return;
    double pose[12];
    for (int i = 0; i < 12; ++i) pose[i] = poses[(size_t)pid * 12 + i];
    const double a[3] = {ba[3 * (size_t)m], ba[3 * (size_t)m + 1], ba[3 * (size_t)m + 2]};
    const double b[3] = {bb[3 * (size_t)m], bb[3 * (size_t)m + 1], bb[3 * (size_t)m + 2]};
    out[(size_t)pid * n + m] = (rs_rotation_checked(pose) && rs_pair_far(pose, a, b, thresh)) ? 1 : 0;
}

}  // namespace

// The per-scene work arrays: slot s of every array belongs to scene s (layout: RsB).  An arena is either complete
// (max_scenes > 0, every pointer valid) or empty (max_scenes == 0, every pointer null) — never half-built: it is
// allocated into a temporary and swapped in only when every allocation succeeded.
struct RsArena {
    uint32_t max_scenes = 0;
    uint32_t* d_n = nullptr;
    double *d_a = nullptr, *d_b = nullptr, *d_poses = nullptr, *d_best_pose = nullptr;
    uint32_t *d_order = nullptr, *d_samples = nullptr, *d_ok = nullptr, *d_counts = nullptr, *d_alive = nullptr, *d_nalive = nullptr,
             *d_best = nullptr, *d_inl = nullptr, *d_ninl = nullptr, *d_first = nullptr, *d_enable = nullptr, *d_frames = nullptr;
    unsigned long long* d_neval = nullptr;
    rs_arrsac_stats* d_stats = nullptr;
};
struct rs_ctx : RsArena {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev = nullptr;
    uint32_t max_matches = 0, max_hyp = 0;
    double* d_logtab = nullptr;                                        // ln(k), k = 0 .. max_matches (host libm values)
    uint32_t last_hyp = 0;
};

static void rs_free_arena(RsArena* c)
{
    hipFree(c->d_n); hipFree(c->d_a); hipFree(c->d_b); hipFree(c->d_poses); hipFree(c->d_best_pose); hipFree(c->d_order);
    hipFree(c->d_samples); hipFree(c->d_ok); hipFree(c->d_counts); hipFree(c->d_alive); hipFree(c->d_nalive); hipFree(c->d_best);
    hipFree(c->d_inl); hipFree(c->d_ninl); hipFree(c->d_first); hipFree(c->d_enable); hipFree(c->d_frames); hipFree(c->d_neval);
    hipFree(c->d_stats);
    *c = RsArena();
}

static int32_t rs_alloc_arena_into(RsArena* A, size_t n, size_t H, uint32_t S)
{
    const size_t s = S;
    AKZ_HIP(hipMalloc(&A->d_n, sizeof(uint32_t) * s));
    AKZ_HIP(hipMalloc(&A->d_a, sizeof(double) * 3 * n * s));
    AKZ_HIP(hipMalloc(&A->d_b, sizeof(double) * 4 * n * s));
    AKZ_HIP(hipMalloc(&A->d_order, sizeof(uint32_t) * n * s));
    AKZ_HIP(hipMalloc(&A->d_samples, sizeof(uint32_t) * 8 * H * s));
    AKZ_HIP(hipMalloc(&A->d_poses, sizeof(double) * 48 * H * s));
    AKZ_HIP(hipMalloc(&A->d_ok, sizeof(uint32_t) * 4 * H * s));
    AKZ_HIP(hipMalloc(&A->d_counts, sizeof(uint32_t) * 4 * H * s));
    AKZ_HIP(hipMalloc(&A->d_alive, sizeof(uint32_t) * 4 * H * s));
    AKZ_HIP(hipMalloc(&A->d_nalive, sizeof(uint32_t) * s));
    AKZ_HIP(hipMalloc(&A->d_neval, sizeof(unsigned long long) * s));
    AKZ_HIP(hipMalloc(&A->d_best, sizeof(uint32_t) * 4 * s));
    AKZ_HIP(hipMalloc(&A->d_inl, sizeof(uint32_t) * n * s));
    AKZ_HIP(hipMalloc(&A->d_ninl, sizeof(uint32_t) * s));
    AKZ_HIP(hipMalloc(&A->d_best_pose, sizeof(double) * 12 * s));
    AKZ_HIP(hipMalloc(&A->d_first, sizeof(uint32_t) * s));
    AKZ_HIP(hipMalloc(&A->d_enable, sizeof(uint32_t) * s));
    AKZ_HIP(hipMalloc(&A->d_frames, sizeof(uint32_t) * 2 * s));
    AKZ_HIP(hipMalloc(&A->d_stats, sizeof(rs_arrsac_stats) * s));
    A->max_scenes = S;
    return AKZ_OK;
}
// A complete arena for S scenes in `*out`, or an error and nothing allocated.
static int32_t rs_alloc_arena(const rs_ctx* c, uint32_t S, RsArena* out)
{
    RsArena A;
    const int32_t st = rs_alloc_arena_into(&A, c->max_matches, c->max_hyp, S);
    if (st != AKZ_OK) {
        rs_free_arena(&A);
        return st;
    }
    *out = A;
    return AKZ_OK;
}
// every entry point that touches the arena starts with this: a context whose arena could not be rebuilt after a failed
// rs_batch_reserve answers AKZ_E_OOM instead of launching on null pointers
#define RS_NEED_ARENA(c)                              \
    do {                                              \
        if ((c)->max_scenes == 0) return AKZ_E_OOM;   \
    } while (0)

static RsB rs_view(const rs_ctx* c, bool with_order)
{
    RsB B;
    B.n_cap = c->max_matches;
    B.H = c->max_hyp;
    B.n = c->d_n;
    B.a = c->d_a;
    B.b = c->d_b;
    B.order = with_order ? c->d_order : nullptr;
    B.samples = c->d_samples;
    B.poses = c->d_poses;
    B.ok = c->d_ok;
    B.counts = c->d_counts;
    B.alive = c->d_alive;
    B.nalive = c->d_nalive;
    B.neval = c->d_neval;
    B.best = c->d_best;
    B.inl = c->d_inl;
    B.ninl = c->d_ninl;
    B.best_pose = c->d_best_pose;
    B.first = c->d_first;
    B.enable = c->d_enable;
    return B;
}

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
        {
            // The consensus is a chain of hundreds of small dependent launches: behind bulk kernels of equal priority every
            // one of them waits for compute units.  Most urgent priority: the registration leg of the bench went from 142.6
            // to 126.5 ms per 256 frames (1 795 -> 2 024 frames/s; with the k = 3 matcher least urgent 125.0).
            int prio_lo = 0, prio_hi = 0;
            hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
            AKZ_HIP(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_hi));
        }
        AKZ_HIP(hipEventCreateWithFlags(&c->ev, hipEventDisableTiming));
        {
            RsArena A;
            AKZ_TRY(rs_alloc_arena(c, 1, &A));
            static_cast<RsArena&>(*c) = A;
        }
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
        rs_free_arena(static_cast<RsArena*>(c));
        hipFree(c->d_logtab);
        if (c->ev) hipEventDestroy(c->ev);
        if (c->stream) hipStreamDestroy(c->stream);
        delete c;
        return AKZ_OK;
    });
}

// Room for `max_scenes` scenes per call (rs_create leaves room for one): the arena is re-carved, nothing else changes.
extern "C" int32_t rs_batch_reserve(rs_ctx* c, uint32_t max_scenes)
{
    return akz_guard([&]() -> int32_t {
        if (!c || max_scenes == 0 || max_scenes > 65535u) return AKZ_E_INVALID;
        if (max_scenes <= c->max_scenes) return AKZ_OK;
        AKZ_HIP(hipSetDevice(c->device));
        AKZ_HIP(hipStreamSynchronize(c->stream));
        // the larger arena first, beside the old one: a failure leaves the context exactly as it was
        RsArena fresh;
        int32_t st = rs_alloc_arena(c, max_scenes, &fresh);
        if (st != AKZ_OK) {
            // not beside it: give the old one back first, and rebuild it if the larger one still does not fit
            const uint32_t old_scenes = c->max_scenes;
            rs_free_arena(static_cast<RsArena*>(c));
            st = rs_alloc_arena(c, max_scenes, &fresh);
            if (st != AKZ_OK) {
                RsArena back;
                if (old_scenes && rs_alloc_arena(c, old_scenes, &back) == AKZ_OK) static_cast<RsArena&>(*c) = back;
                return st;        // (if even that failed the arena is empty and every entry point answers AKZ_E_OOM)
            }
        } else {
            rs_free_arena(static_cast<RsArena*>(c));
        }
        static_cast<RsArena&>(*c) = fresh;
        return AKZ_OK;
    });
}

extern "C" int32_t rs_sync(rs_ctx* c)
{
    return akz_guard([&]() -> int32_t {
        if (!c) return AKZ_E_INVALID;
        AKZ_HIP(hipSetDevice(c->device));
        AKZ_HIP(hipStreamSynchronize(c->stream));
        return AKZ_OK;
    });
}

extern "C" void* rs_stream(rs_ctx* c) { return c ? (void*)c->stream : nullptr; }

// cv_pinhole::CameraIntrinsics::calibrate / CameraIntrinsicsK1Distortion::calibrate: host scalar math.
extern "C" int32_t rs_calibrate(const double* intr, int32_t use_k1, double k1, const akz_keypoint* kps, uint32_t n,
                                double* out)
{
    return akz_guard([&]() -> int32_t {
        if (!intr || (n && (!kps || !out))) return AKZ_E_INVALID;
        for (uint32_t i = 0; i < n; ++i) rs_calibrate_one(intr, use_k1, k1, kps[i].x, kps[i].y, out + 3 * (size_t)i);
        return AKZ_OK;
    });
}

// results of a single-scene call: scene 0's slots of the arena -> the caller's host buffers
static int32_t rs_fetch_single(rs_ctx* c, double* best_pose, uint32_t* best_id, uint32_t* inlier_idx, uint32_t cap, uint32_t* n_inliers,
                               uint32_t* survivors, unsigned long long* neval)
{
    hipStream_t s = c->stream;
    uint32_t best[4] = {0, 0, 0, 0}, ninl = 0;
    unsigned long long ne = 0;
    AKZ_HIP(hipMemcpyAsync(best, c->d_best, sizeof(best), hipMemcpyDeviceToHost, s));
    AKZ_HIP(hipMemcpyAsync(&ninl, c->d_ninl, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    AKZ_HIP(hipMemcpyAsync(&ne, c->d_neval, sizeof(ne), hipMemcpyDeviceToHost, s));
    AKZ_HIP(hipMemcpyAsync(best_pose, c->d_best_pose, sizeof(double) * 12, hipMemcpyDeviceToHost, s));
    AKZ_HIP(hipStreamSynchronize(s));
    if (survivors) *survivors = best[2];
    if (neval) *neval = ne;
    *best_id = best[0];
    *n_inliers = ninl;
    if (best[0] == 0xFFFFFFFFu) {
        *n_inliers = 0;
        return AKZ_OK;  // Consensus::model_inliers returned None: no hypothesis produced a model
    }
    uint32_t ncopy = ninl < cap ? ninl : cap;
    if (ncopy) AKZ_HIP(hipMemcpy(inlier_idx, c->d_inl, sizeof(uint32_t) * ncopy, hipMemcpyDeviceToHost));
    return ninl > cap ? AKZ_E_CAPACITY : AKZ_OK;
}

// exhaustive scoring of caller-provided minimal samples, one scene (slot 0 of the arena)
template <bool P3P>
static int32_t exhaustive_run(rs_ctx* c, const double* in_a, const double* in_b, uint32_t n, const uint32_t* sample_idx,
                              uint32_t n_hyp, double thresh, double* best_pose, uint32_t* best_id, uint32_t* inlier_idx,
                              uint32_t cap, uint32_t* n_inliers)
{
    constexpr uint32_t K = P3P ? 3u : 8u, BW = P3P ? 4u : 3u;
    if (!c || !in_a || !in_b || !sample_idx || !best_pose || !best_id || !n_inliers || (cap && !inlier_idx)) return AKZ_E_INVALID;
    RS_NEED_ARENA(c);
    if (n < K || n_hyp == 0) return AKZ_E_INVALID;  // MIN_SAMPLES (eight-point/src/lib.rs:73, lambda-twist/src/lib.rs:333)
    if (n > c->max_matches || n_hyp > c->max_hyp) return AKZ_E_TOO_LARGE;
    for (size_t i = 0; i < (size_t)n_hyp * K; ++i)
        if (sample_idx[i] >= n) return AKZ_E_INVALID;
    AKZ_HIP(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    AKZ_HIP(hipMemcpyAsync(c->d_a, in_a, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, s));
    AKZ_HIP(hipMemcpyAsync(c->d_b, in_b, sizeof(double) * BW * (size_t)n, hipMemcpyHostToDevice, s));
    AKZ_HIP(hipMemcpyAsync(c->d_samples, sample_idx, sizeof(uint32_t) * K * (size_t)n_hyp, hipMemcpyHostToDevice, s));
    AKZ_HIP(hipMemsetD32Async((hipDeviceptr_t)c->d_n, (int)n, 1, s));
    AKZ_HIP(hipMemsetAsync(c->d_counts, 0, sizeof(uint32_t) * 4 * (size_t)n_hyp, s));
    AKZ_HIP(hipMemsetAsync(c->d_neval, 0, sizeof(unsigned long long), s));
    const RsB B = rs_view(c, false);
    if (P3P)
        hipLaunchKernelGGL(k_rsb_p3p_hypotheses, dim3((n_hyp + 63) / 64, 1), dim3(64), 0, s, B, 0u, n_hyp, (const uint32_t*)nullptr);
    else
        hipLaunchKernelGGL(k_rsb_hypotheses, dim3((n_hyp + 63) / 64, 1), dim3(64), 0, s, B, 0u, n_hyp, (const uint32_t*)nullptr);
    AKZ_LAUNCH_CHECK();
    // grid.y is limited to 65535: score the poses in slabs
    const uint32_t n_pose = n_hyp * 4;
    for (uint32_t p0 = 0; P3P && p0 < n_pose; p0 += 65532) {
        uint32_t np = n_pose - p0 < 65532 ? n_pose - p0 : 65532;     // (65532: whole hypotheses per slab)
        if (P3P)
            hipLaunchKernelGGL(k_p3p_score, dim3((n + 255) / 256, np), dim3(256), 0, s, c->d_a, c->d_b, n,
                               c->d_poses + (size_t)p0 * 12, c->d_ok + p0, thresh, c->d_counts + p0);
        AKZ_LAUNCH_CHECK();
    }
    if (!P3P) {
        // the first-block kernel of the ARRSAC-shaped loop over ALL matches: far pairs are discarded by the bound, the rest
        // are dealt to full waves; blockIdx.y splits the match list so that one scene still fills the chip
        AKZ_HIP(hipMemsetAsync(c->d_nalive, 0, sizeof(uint32_t), s));   // (no residual count is kept here)
        const uint32_t gx = (2 * n_hyp + kFirstUnits - 1) / kFirstUnits, n_tiles = (n + kFirstTile - 1) / kFirstTile;
        uint32_t gy = gx >= 1024 ? 1 : 1024 / gx;
        gy = gy > n_tiles ? n_tiles : gy;
        hipLaunchKernelGGL(k_rsb_score_first, dim3(gx, gy, 1), dim3(256), 0, s, B, n, n_hyp, thresh);
        AKZ_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_rs_best, dim3(1), dim3(1024), 0, s, c->d_counts, c->d_ok, n_pose, c->d_best);
    AKZ_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_rs_inliers<P3P>), dim3(1), dim3(1024), 0, s, c->d_a, c->d_b, n, c->d_poses, c->d_best, thresh,
                       c->d_inl, n, c->d_ninl, c->d_best_pose);
    AKZ_LAUNCH_CHECK();
    c->last_hyp = n_hyp;
    return rs_fetch_single(c, best_pose, best_id, inlier_idx, cap, n_inliers, nullptr, nullptr);
}

extern "C" int32_t rs_essential_batch(rs_ctx* c, const double* bearings_a, const double* bearings_b, uint32_t n,
                                      const uint32_t* sample_idx, uint32_t n_hyp, double thresh, double* best_pose,
                                      uint32_t* best_id, uint32_t* inlier_idx, uint32_t cap, uint32_t* n_inliers)
{
    return akz_guard([&]() -> int32_t {
        return exhaustive_run<false>(c, bearings_a, bearings_b, n, sample_idx, n_hyp, thresh, best_pose, best_id, inlier_idx, cap,
                                     n_inliers);
    });
}

// Consensus::model_inliers(&LambdaTwist::new(), world_matches) with the sampler factored out
// (cv-sfm/src/lib.rs:1619-1622; lambda-twist/tests/consensus.rs:59-61): n_hyp sample triples.
extern "C" int32_t rs_p3p_batch(rs_ctx* c, const double* bearings, const double* world, uint32_t n,
                                const uint32_t* sample_idx, uint32_t n_hyp, double thresh, double* best_pose,
                                uint32_t* best_id, uint32_t* inlier_idx, uint32_t cap, uint32_t* n_inliers)
{
    return akz_guard([&]() -> int32_t {
        return exhaustive_run<true>(c, bearings, world, n, sample_idx, n_hyp, thresh, best_pose, best_id, inlier_idx, cap, n_inliers);
    });
}

static int32_t rs_check_params(const rs_ctx* c, const rs_arrsac_params* prm, uint32_t n_max, uint32_t* blocks_max)
{
    if (prm->struct_size != sizeof(rs_arrsac_params)) return AKZ_E_INVALID;
    if (prm->n_hypotheses == 0 || prm->block_size == 0) return AKZ_E_INVALID;
    if (prm->reserved != 0 || (prm->flags & ~(RS_PRUNE_BOUND | RS_PRUNE_SPRT | RS_PRUNE_HALVE))) return AKZ_E_INVALID;
    // every block but the last may add E hypotheses: they need room in the context's pose arrays
    const uint64_t n_blocks_max = ((uint64_t)n_max + prm->block_size - 1) / prm->block_size;
    if (n_max > c->max_matches || prm->n_hypotheses > c->max_hyp ||
        (uint64_t)prm->n_hypotheses + (uint64_t)prm->estimations_per_block * n_blocks_max > c->max_hyp)
        return AKZ_E_TOO_LARGE;
    if ((prm->flags & RS_PRUNE_SPRT) && !(prm->sprt_delta > 0.0 && prm->sprt_delta < 1.0 && prm->sprt_ratio > 1.0))
        return AKZ_E_INVALID;
    *blocks_max = (uint32_t)n_blocks_max;
    return AKZ_OK;
}

// The ARRSAC-shaped loop over S scenes whose matches (and d_n) are already in the arena; enqueues everything on the
// context's stream and returns.  n_max: the host's upper bound of the scenes' match counts (sizes the block loop).
// have_samples: the minimal samples are already in the arena (single-scene calls with caller samples).
template <bool P3P>
static int32_t arrsac_engine(rs_ctx* c, uint32_t S, uint32_t n_max, const rs_arrsac_params* prm, bool have_samples, bool with_order,
                             const RsOut& out_in, uint32_t* blocks_run, uint32_t* hyp_made)
{
    constexpr uint32_t K = P3P ? 3u : 8u;
    hipStream_t s = c->stream;
    const RsB B = rs_view(c, with_order);
    const uint32_t n_hyp = prm->n_hypotheses, E = prm->estimations_per_block;
    if (!have_samples) {
        hipLaunchKernelGGL((k_rsb_sample<(int)K>), dim3((n_hyp + 255) / 256, S), dim3(256), 0, s, B, (unsigned long long)prm->seed, n_hyp);
        AKZ_LAUNCH_CHECK();
    }
    // counters of the initial poses (re-sampled poses zero theirs when they join); slot stride 4 H
    if (S == 1) AKZ_HIP(hipMemsetAsync(c->d_counts, 0, sizeof(uint32_t) * 4 * (size_t)n_hyp, s));
    else AKZ_HIP(hipMemsetAsync(c->d_counts, 0, sizeof(uint32_t) * 4 * (size_t)c->max_hyp * S, s));
    AKZ_HIP(hipMemsetAsync(c->d_neval, 0, sizeof(unsigned long long) * S, s));
    if (P3P)
        hipLaunchKernelGGL(k_rsb_p3p_hypotheses, dim3((n_hyp + 63) / 64, S), dim3(64), 0, s, B, 0u, n_hyp, (const uint32_t*)nullptr);
    else
        hipLaunchKernelGGL(k_rsb_hypotheses, dim3((n_hyp + 63) / 64, S), dim3(64), 0, s, B, 0u, n_hyp, (const uint32_t*)nullptr);
    AKZ_LAUNCH_CHECK();
    const uint32_t n_pose = n_hyp * 4;
    hipLaunchKernelGGL(k_rsb_alive_init, dim3(S), dim3(1024), 0, s, B, n_pose);
    AKZ_LAUNCH_CHECK();
    uint32_t blocks = 0;
    // the live count is known to the host only as an upper bound: n_pose before the cap applies, the cap after
    uint32_t live_bound = n_pose;
    const bool prune = (prm->flags & (RS_PRUNE_BOUND | RS_PRUNE_SPRT)) != 0 || prm->max_candidates != 0 || E != 0;
    uint32_t next_h = n_hyp;                                  // first hypothesis slot of the next re-sampling round
    auto score = [&](uint32_t m_lo, uint32_t m_hi, uint32_t slots, uint32_t from_first) -> int32_t {
        const uint32_t range = m_hi - m_lo;
        if constexpr (P3P) {
            uint32_t gy = range / 64;
            gy = gy < 1 ? 1 : (gy > 16 ? 16 : gy);
            hipLaunchKernelGGL(k_rsb_score_p3p, dim3((slots + 255) / 256, gy, S), dim3(256), 0, s, B, m_lo, m_hi, from_first, prm->threshold);
        } else {
            uint32_t lg = 6;
            if (range < 64) {
                lg = 0;
                while ((1u << lg) < range) ++lg;
            }
            const uint32_t G = 1u << lg, per_wave = 64u >> lg;
            const uint32_t chunks = (range + G - 1) / G, gy = chunks < 16 ? chunks : 16;
            const uint32_t waves = (slots + per_wave - 1) / per_wave;
            hipLaunchKernelGGL(k_rsb_score, dim3((waves + 3) / 4, gy, S), dim3(256), 0, s, B, m_lo, m_hi, lg, from_first, prm->threshold);
        }
        AKZ_LAUNCH_CHECK();
        return AKZ_OK;
    };
    auto score_first = [&](uint32_t m_hi) -> int32_t {     // block 0 of the two-view consensus: poses in (t, -t) pairs
        hipLaunchKernelGGL(k_rsb_score_first, dim3((2 * n_hyp + kFirstUnits - 1) / kFirstUnits, 1, S), dim3(256), 0, s, B, m_hi, n_hyp,
                           prm->threshold);
        AKZ_LAUNCH_CHECK();
        return AKZ_OK;
    };
    for (uint32_t m_lo = 0; m_lo < n_max;) {
        // without pruning there is nothing to decide between blocks: one block = all matches
        const uint32_t bs = prune ? prm->block_size : n_max;
        const uint32_t m_hi = m_lo + bs < n_max ? m_lo + bs : n_max;
        if (!P3P && m_lo == 0) AKZ_TRY(score_first(m_hi));
        else AKZ_TRY(score(m_lo, m_hi, live_bound, 0u));
        ++blocks;
        m_lo = m_hi;
        if (prune && m_lo < n_max) {
            RsPrune P;
            P.seen = m_lo;
            P.cap = 0u;
            P.pad = 0u;
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
            hipLaunchKernelGGL(k_rsb_prune, dim3(S), dim3(kChainNT), 0, s, B, P);
            AKZ_LAUNCH_CHECK();
            if (P.cap && P.cap < live_bound) live_bound = P.cap;
            // a single survivor cannot be overtaken when nothing is re-sampled: the block loop ends (the specification's rule)
            if (P.cap == 1 && E == 0 && (prm->flags & RS_PRUNE_HALVE)) break;
            if (E && blocks >= prm->init_blocks) {
                // inlier-guided re-sampling: list the inliers (matches seen so far) of the best survivor, draw E minimal
                // samples among them, estimate, and let the valid poses join the live list after catching up on [0, seen)
                hipLaunchKernelGGL((k_rsb_best_inliers<P3P>), dim3(S), dim3(kChainNT), 0, s, B, m_lo, 0u, prm->threshold, out_in);
                AKZ_LAUNCH_CHECK();
                hipLaunchKernelGGL((k_rsb_resample<(int)K>), dim3((E + 255) / 256, S), dim3(256), 0, s, B, (unsigned long long)prm->seed,
                                   next_h, E);
                AKZ_LAUNCH_CHECK();
                if (P3P)
                    hipLaunchKernelGGL(k_rsb_p3p_hypotheses, dim3((E + 63) / 64, S), dim3(64), 0, s, B, next_h, E, (const uint32_t*)c->d_enable);
                else
                    hipLaunchKernelGGL(k_rsb_hypotheses, dim3((E + 63) / 64, S), dim3(64), 0, s, B, next_h, E, (const uint32_t*)c->d_enable);
                AKZ_LAUNCH_CHECK();
                hipLaunchKernelGGL(k_rsb_alive_append, dim3(S), dim3(kChainNT), 0, s, B, next_h * 4, E * 4);
                AKZ_LAUNCH_CHECK();
                AKZ_TRY(score(0u, m_lo, 4 * E, 1u));
                next_h += E;
                live_bound += 4 * E;
            }
        }
    }
    RsOut O = out_in;
    O.n_hyp = n_hyp;
    O.resample = E;
    O.init_blocks = prm->init_blocks;
    O.blocks_run = blocks;
    O.block_size = prune ? prm->block_size : 0u;
    O.min_samples = K;
    hipLaunchKernelGGL((k_rsb_best_inliers<P3P>), dim3(S), dim3(kChainNT), 0, s, B, 0u, 1u, prm->threshold, O);
    AKZ_LAUNCH_CHECK();
    *blocks_run = blocks;
    *hyp_made = next_h;
    return AKZ_OK;
}

// Consensus::model_inliers in ARRSAC's shape for one scene with host buffers: EightPoint (two bearing sets, 8-match
// samples) or LambdaTwist (bearings + world points, 3-match samples).  sample_idx == NULL draws the samples on the device.
template <bool P3P>
static int32_t arrsac_run(rs_ctx* c, const double* in_a, const double* in_b, uint32_t n, const uint32_t* sample_idx,
                          const rs_arrsac_params* prm, double* best_pose, uint32_t* best_id, uint32_t* inlier_idx,
                          uint32_t cap, uint32_t* n_inliers, rs_arrsac_stats* stats)
{
    constexpr uint32_t K = P3P ? 3u : 8u, BW = P3P ? 4u : 3u;   // sample size; doubles per element of the second input
    if (!c || !in_a || !in_b || !prm || !best_pose || !best_id || !n_inliers || (cap && !inlier_idx)) return AKZ_E_INVALID;
    RS_NEED_ARENA(c);
    if (n < K) return AKZ_E_INVALID;   // MIN_SAMPLES (eight-point/src/lib.rs:73, lambda-twist/src/lib.rs:333)
    uint32_t blocks_max = 0;
    AKZ_TRY(rs_check_params(c, prm, n, &blocks_max));
    if (sample_idx)
        for (size_t i = 0; i < (size_t)prm->n_hypotheses * K; ++i)
            if (sample_idx[i] >= n) return AKZ_E_INVALID;
    AKZ_HIP(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    AKZ_HIP(hipMemcpyAsync(c->d_a, in_a, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, s));
    AKZ_HIP(hipMemcpyAsync(c->d_b, in_b, sizeof(double) * BW * (size_t)n, hipMemcpyHostToDevice, s));
    AKZ_HIP(hipMemsetD32Async((hipDeviceptr_t)c->d_n, (int)n, 1, s));
    if (sample_idx)
        AKZ_HIP(hipMemcpyAsync(c->d_samples, sample_idx, sizeof(uint32_t) * K * (size_t)prm->n_hypotheses, hipMemcpyHostToDevice, s));
    RsOut O;
    O.best_id = c->d_best + 3;      // (slot 3 of scene 0's best[4]: the final kernel also fills slots 0..2)
    O.pose = c->d_best_pose;
    O.inl = c->d_inl;
    O.ninl = c->d_ninl;
    O.stats = nullptr;
    O.inl_stride = c->max_matches;
    O.n_hyp = O.resample = O.init_blocks = O.blocks_run = O.block_size = O.min_samples = 0;
    uint32_t blocks = 0, made = 0;
    AKZ_TRY((arrsac_engine<P3P>(c, 1, n, prm, sample_idx != nullptr, false, O, &blocks, &made)));
    uint32_t survivors = 0;
    unsigned long long neval = 0;
    const int32_t st = rs_fetch_single(c, best_pose, best_id, inlier_idx, cap, n_inliers, &survivors, &neval);
    c->last_hyp = made;
    if (stats) {
        stats->poses = made * 4;
        stats->survivors = survivors;
        stats->blocks = blocks;
        stats->reserved = 0;
        stats->residuals_evaluated = neval;
        stats->residuals_exhaustive = (uint64_t)made * 4 * n;
    }
    return st;
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

// Two-view verification of a whole micro-batch, device-resident end to end (cv-sfm/src/lib.rs:1385-1412 for every
// frame pair the matcher produced): scene s = pair list s of hm_match_batch_device's output, keypoints of blocks
// ia[s] / ib[s]; calibrate (cv-pinhole/src/lib.rs:108-117), optional seeded shuffle, the ARRSAC-shaped loop over all
// scenes at once.  Enqueued on rs_stream() after stream_to_wait; nothing is copied to the host.
extern "C" int32_t rs_essential_arrsac_batch_device(rs_ctx* c, const void* d_kps_a, const void* d_kps_b, uint32_t cap_per_img,
                                                    const uint32_t* ia, const uint32_t* ib, const void* d_pairs, const void* d_npairs,
                                                    uint32_t n_scenes, const rs_camera* cam_a, const rs_camera* cam_b,
                                                    const rs_arrsac_params* prm, uint32_t flags, void* d_pose, void* d_best_id,
                                                    void* d_inliers, void* d_n_inliers, void* d_stats, void* stream_to_wait)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !d_kps_a || !d_kps_b || !ia || !ib || !d_pairs || !d_npairs || !cam_a || !cam_b || !prm || !d_pose || !d_best_id ||
            !d_inliers || !d_n_inliers)
            return AKZ_E_INVALID;
        if (cap_per_img == 0 || (flags & ~(uint32_t)RS_BATCH_SHUFFLE) || cam_a->reserved != 0 || cam_b->reserved != 0) return AKZ_E_INVALID;
        if (n_scenes == 0) return AKZ_OK;
        RS_NEED_ARENA(c);
        if (n_scenes > c->max_scenes) return AKZ_E_TOO_LARGE;
        const uint32_t n_max = cap_per_img < c->max_matches ? cap_per_img : c->max_matches;
        if (n_max < 8) return AKZ_E_INVALID;
        if ((flags & RS_BATCH_SHUFFLE) && n_max > kRadixSortMax) return AKZ_E_TOO_LARGE;   // the shuffle sorts a scene's keys in LDS
        uint32_t blocks_max = 0;
        AKZ_TRY(rs_check_params(c, prm, n_max, &blocks_max));
        AKZ_HIP(hipSetDevice(c->device));
        hipStream_t s = c->stream;
        if (stream_to_wait) {
            AKZ_HIP(hipEventRecord(c->ev, akz_wait_stream(stream_to_wait)));
            AKZ_HIP(hipStreamWaitEvent(s, c->ev, 0));
        }
        AKZ_HIP(hipMemcpyAsync(c->d_frames, ia, sizeof(uint32_t) * n_scenes, hipMemcpyHostToDevice, s));
        AKZ_HIP(hipMemcpyAsync(c->d_frames + c->max_scenes, ib, sizeof(uint32_t) * n_scenes, hipMemcpyHostToDevice, s));
        auto cam = [](const rs_camera* k) {
            RsCam r;
            r.intr[0] = k->fx; r.intr[1] = k->fy; r.intr[2] = k->cx; r.intr[3] = k->cy; r.intr[4] = k->skew;
            r.k1 = k->k1;
            r.use_k1 = k->use_k1 ? 1 : 0;
            r.pad = 0;
            return r;
        };
        const bool shuffle = (flags & RS_BATCH_SHUFFLE) != 0;
        const RsB B = rs_view(c, shuffle);
        if (shuffle)
            AKZ_HIP(hipFuncSetAttribute((const void*)k_rsb_prepare<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRadixSortLdsBytes));
        hipLaunchKernelGGL(k_rsb_prepare<false>, dim3(n_scenes), dim3(1024), shuffle ? kRadixSortLdsBytes : 0, s, B, (const akz_keypoint*)d_kps_a,
                           d_kps_b, cap_per_img, (const uint32_t*)c->d_frames,
                           (const uint32_t*)(c->d_frames + c->max_scenes), (const uint32_t*)d_pairs, (const uint32_t*)d_npairs, cam(cam_a),
                           cam(cam_b), c->d_n, c->d_a, c->d_b, c->d_order, shuffle ? 1u : 0u, (unsigned long long)prm->seed, cap_per_img);
        AKZ_LAUNCH_CHECK();
        RsOut O;
        O.best_id = (uint32_t*)d_best_id;
        O.pose = (double*)d_pose;
        O.inl = (uint32_t*)d_inliers;
        O.ninl = (uint32_t*)d_n_inliers;
        O.stats = (rs_arrsac_stats*)d_stats;
        O.inl_stride = cap_per_img;
        O.n_hyp = O.resample = O.init_blocks = O.blocks_run = O.block_size = O.min_samples = 0;
        uint32_t blocks = 0, made = 0;
        AKZ_TRY((arrsac_engine<false>(c, n_scenes, n_max, prm, false, shuffle, O, &blocks, &made)));
        c->last_hyp = made;
        return AKZ_OK;
    });
}

// The registration path's consensus for a whole micro-batch (cv-sfm/src/lib.rs:1571-1622 for every new frame): scene s =
// a list of (feature of keypoint block ik[s], world point) index pairs; bearing = calibrate(keypoint), point = d_world[idx]
// (homogeneous, [n_world][4] f64: the control plane's triangulated landmarks; a scene that names a point >= n_world is refused); Lambda Twist hypotheses from 3-match
// samples, WorldToCamera::residual, the same ARRSAC-shaped loop over all scenes at once.
extern "C" int32_t rs_p3p_arrsac_batch_device(rs_ctx* c, const void* d_kps, uint32_t cap_per_img, const uint32_t* ik, const void* d_pairs,
                                              const void* d_npairs, uint32_t n_scenes, const void* d_world, uint32_t n_world,
                                              const rs_camera* cam, const rs_arrsac_params* prm, uint32_t flags, void* d_pose,
                                              void* d_best_id, void* d_inliers, void* d_n_inliers, void* d_stats, void* stream_to_wait)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !d_kps || !ik || !d_pairs || !d_npairs || !d_world || !cam || !prm || !d_pose || !d_best_id || !d_inliers || !d_n_inliers)
            return AKZ_E_INVALID;
        if (n_world == 0) return AKZ_E_INVALID;
        if (cap_per_img == 0 || (flags & ~(uint32_t)RS_BATCH_SHUFFLE) || cam->reserved != 0) return AKZ_E_INVALID;
        if (n_scenes == 0) return AKZ_OK;
        RS_NEED_ARENA(c);
        if (n_scenes > c->max_scenes) return AKZ_E_TOO_LARGE;
        const uint32_t n_max = cap_per_img < c->max_matches ? cap_per_img : c->max_matches;
        if (n_max < 3) return AKZ_E_INVALID;
        if ((flags & RS_BATCH_SHUFFLE) && n_max > kRadixSortMax) return AKZ_E_TOO_LARGE;
        uint32_t blocks_max = 0;
        AKZ_TRY(rs_check_params(c, prm, n_max, &blocks_max));
        AKZ_HIP(hipSetDevice(c->device));
        hipStream_t s = c->stream;
        if (stream_to_wait) {
            AKZ_HIP(hipEventRecord(c->ev, akz_wait_stream(stream_to_wait)));
            AKZ_HIP(hipStreamWaitEvent(s, c->ev, 0));
        }
        AKZ_HIP(hipMemcpyAsync(c->d_frames, ik, sizeof(uint32_t) * n_scenes, hipMemcpyHostToDevice, s));
        RsCam k;
        k.intr[0] = cam->fx; k.intr[1] = cam->fy; k.intr[2] = cam->cx; k.intr[3] = cam->cy; k.intr[4] = cam->skew;
        k.k1 = cam->k1;
        k.use_k1 = cam->use_k1 ? 1 : 0;
        k.pad = 0;
        const bool shuffle = (flags & RS_BATCH_SHUFFLE) != 0;
        const RsB B = rs_view(c, shuffle);
        if (shuffle)
            AKZ_HIP(hipFuncSetAttribute((const void*)k_rsb_prepare<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRadixSortLdsBytes));
        hipLaunchKernelGGL(k_rsb_prepare<true>, dim3(n_scenes), dim3(1024), shuffle ? kRadixSortLdsBytes : 0, s, B, (const akz_keypoint*)d_kps,
                           d_world, cap_per_img, (const uint32_t*)c->d_frames, (const uint32_t*)c->d_frames, (const uint32_t*)d_pairs,
                           (const uint32_t*)d_npairs, k, k, c->d_n, c->d_a, c->d_b, c->d_order, shuffle ? 1u : 0u,
                           (unsigned long long)prm->seed, n_world);
        AKZ_LAUNCH_CHECK();
        RsOut O;
        O.best_id = (uint32_t*)d_best_id;
        O.pose = (double*)d_pose;
        O.inl = (uint32_t*)d_inliers;
        O.ninl = (uint32_t*)d_n_inliers;
        O.stats = (rs_arrsac_stats*)d_stats;
        O.inl_stride = cap_per_img;
        O.n_hyp = O.resample = O.init_blocks = O.blocks_run = O.block_size = O.min_samples = 0;
        uint32_t blocks = 0, made = 0;
        AKZ_TRY((arrsac_engine<true>(c, n_scenes, n_max, prm, false, shuffle, O, &blocks, &made)));
        c->last_hyp = made;
        return AKZ_OK;
    });
}

// debug tap of the batched entry: the calibrated bearings and the scoring order of scene `scene` of the last call
extern "C" int32_t rs_debug_scene(rs_ctx* c, uint32_t scene, uint32_t* n, double* bearings_a, double* bearings_b, uint32_t* order,
                                  uint32_t cap)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !n || scene >= c->max_scenes) return AKZ_E_INVALID;
        AKZ_HIP(hipSetDevice(c->device));
        AKZ_HIP(hipStreamSynchronize(c->stream));
        AKZ_HIP(hipMemcpy(n, c->d_n + scene, sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (*n > cap) return AKZ_E_CAPACITY;
        const size_t N = c->max_matches;
        if (bearings_a) AKZ_HIP(hipMemcpy(bearings_a, c->d_a + scene * N * 3, sizeof(double) * 3 * (size_t)*n, hipMemcpyDeviceToHost));
        if (bearings_b) AKZ_HIP(hipMemcpy(bearings_b, c->d_b + scene * N * 4, sizeof(double) * 3 * (size_t)*n, hipMemcpyDeviceToHost));
        if (order) AKZ_HIP(hipMemcpy(order, c->d_order + scene * N, sizeof(uint32_t) * (size_t)*n, hipMemcpyDeviceToHost));
        return AKZ_OK;
    });
}

// the same tap after rs_p3p_arrsac_batch_device: bearings [n][3], world points [n][4]
extern "C" int32_t rs_debug_scene_world(rs_ctx* c, uint32_t scene, uint32_t* n, double* bearings, double* world, uint32_t* order,
                                        uint32_t cap)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !n || scene >= c->max_scenes) return AKZ_E_INVALID;
        AKZ_HIP(hipSetDevice(c->device));
        AKZ_HIP(hipStreamSynchronize(c->stream));
        AKZ_HIP(hipMemcpy(n, c->d_n + scene, sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (*n > cap) return AKZ_E_CAPACITY;
        const size_t N = c->max_matches;
        if (bearings) AKZ_HIP(hipMemcpy(bearings, c->d_a + scene * N * 3, sizeof(double) * 3 * (size_t)*n, hipMemcpyDeviceToHost));
        if (world) AKZ_HIP(hipMemcpy(world, c->d_b + scene * N * 4, sizeof(double) * 4 * (size_t)*n, hipMemcpyDeviceToHost));
        if (order) AKZ_HIP(hipMemcpy(order, c->d_order + scene * N, sizeof(uint32_t) * (size_t)*n, hipMemcpyDeviceToHost));
        return AKZ_OK;
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

// parity tap: inlier count of every (hypothesis, pose) of the last single-scene call
extern "C" int32_t rs_debug_counts(rs_ctx* c, uint32_t* counts, uint32_t cap)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !counts) return AKZ_E_INVALID;
        RS_NEED_ARENA(c);
        if (cap < c->last_hyp * 4) return AKZ_E_CAPACITY;
        AKZ_HIP(hipSetDevice(c->device));
        AKZ_HIP(hipMemcpy(counts, c->d_counts, sizeof(uint32_t) * 4 * (size_t)c->last_hyp, hipMemcpyDeviceToHost));
        return AKZ_OK;
    });
}

// parity tap: the poses [n_hyp][4][12] and validity flags [n_hyp][4] of the last single-scene call
extern "C" int32_t rs_debug_poses(rs_ctx* c, double* poses, uint32_t* ok, uint32_t n_hyp)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !poses || !ok) return AKZ_E_INVALID;
        RS_NEED_ARENA(c);
        if (n_hyp > c->last_hyp) return AKZ_E_INVALID;
        AKZ_HIP(hipSetDevice(c->device));
        AKZ_HIP(hipStreamSynchronize(c->stream));
        AKZ_HIP(hipMemcpy(poses, c->d_poses, sizeof(double) * 48 * (size_t)n_hyp, hipMemcpyDeviceToHost));
        AKZ_HIP(hipMemcpy(ok, c->d_ok, sizeof(uint32_t) * 4 * (size_t)n_hyp, hipMemcpyDeviceToHost));
        return AKZ_OK;
    });
}

// parity tap: CameraToCamera::residual of every (pose, match) as the device evaluates it (host buffers in and out)
extern "C" int32_t rs_debug_residuals(rs_ctx* c, const double* poses, uint32_t n_pose, const double* bearings_a, const double* bearings_b,
                                      uint32_t n, int32_t paired, double* out)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !poses || !bearings_a || !bearings_b || !out || n_pose == 0 || n == 0 || n_pose > 65535u) return AKZ_E_INVALID;
        RS_NEED_ARENA(c);
        AKZ_HIP(hipSetDevice(c->device));
        double *d_p = nullptr, *d_a = nullptr, *d_b = nullptr, *d_o = nullptr;
        const size_t no = (size_t)n_pose * n * (paired ? 2 : 1);
        AKZ_HIP(hipMalloc(&d_p, sizeof(double) * 12 * n_pose));
        AKZ_HIP(hipMalloc(&d_a, sizeof(double) * 3 * n));
        AKZ_HIP(hipMalloc(&d_b, sizeof(double) * 3 * n));
        AKZ_HIP(hipMalloc(&d_o, sizeof(double) * no));
        AKZ_HIP(hipMemcpy(d_p, poses, sizeof(double) * 12 * n_pose, hipMemcpyHostToDevice));
        AKZ_HIP(hipMemcpy(d_a, bearings_a, sizeof(double) * 3 * n, hipMemcpyHostToDevice));
        AKZ_HIP(hipMemcpy(d_b, bearings_b, sizeof(double) * 3 * n, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_rs_debug_residuals, dim3((n + 255) / 256, n_pose), dim3(256), 0, c->stream, d_p, n_pose, d_a, d_b, n,
                           paired ? 1 : 0, d_o);
        AKZ_LAUNCH_CHECK();
        AKZ_HIP(hipStreamSynchronize(c->stream));
        AKZ_HIP(hipMemcpy(out, d_o, sizeof(double) * no, hipMemcpyDeviceToHost));
        hipFree(d_p); hipFree(d_a); hipFree(d_b); hipFree(d_o);
        return AKZ_OK;
    });
}

// parity tap of the shortcut in front of the eigen-decomposition: out[pose][match] = 1 where rs_pair_far (and the pose's
// rotation check) decides "residual >= thresh" on its own.  Tests hold it to the residuals of rs_debug_residuals.
extern "C" int32_t rs_debug_far(rs_ctx* c, const double* poses, uint32_t n_pose, const double* bearings_a, const double* bearings_b,
                                uint32_t n, double thresh, uint8_t* out)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !poses || !bearings_a || !bearings_b || !out || n_pose == 0 || n == 0 || n_pose > 65535u) return AKZ_E_INVALID;
        RS_NEED_ARENA(c);
        AKZ_HIP(hipSetDevice(c->device));
        double *d_p = nullptr, *d_a = nullptr, *d_b = nullptr;
        unsigned char* d_o = nullptr;
        const size_t no = (size_t)n_pose * n;
        AKZ_HIP(hipMalloc(&d_p, sizeof(double) * 12 * n_pose));
        AKZ_HIP(hipMalloc(&d_a, sizeof(double) * 3 * n));
        AKZ_HIP(hipMalloc(&d_b, sizeof(double) * 3 * n));
        AKZ_HIP(hipMalloc(&d_o, no));
        AKZ_HIP(hipMemcpy(d_p, poses, sizeof(double) * 12 * n_pose, hipMemcpyHostToDevice));
        AKZ_HIP(hipMemcpy(d_a, bearings_a, sizeof(double) * 3 * n, hipMemcpyHostToDevice));
        AKZ_HIP(hipMemcpy(d_b, bearings_b, sizeof(double) * 3 * n, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_rs_debug_far, dim3((n + 255) / 256, n_pose), dim3(256), 0, c->stream, d_p, n_pose, d_a, d_b, n, thresh, d_o);
        AKZ_LAUNCH_CHECK();
        AKZ_HIP(hipStreamSynchronize(c->stream));
        AKZ_HIP(hipMemcpy(out, d_o, no, hipMemcpyDeviceToHost));
        hipFree(d_p); hipFree(d_a); hipFree(d_b); hipFree(d_o);
        return AKZ_OK;
    });
}
