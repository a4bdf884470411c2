/* akz_ransac_math.h — the dense f64 eigen-solver used by the two-view geometric-verification path
 * (SURVEY.md §8a rows R1-R3), written as plain IEEE double arithmetic so that gcc (CPU oracle) and
 * hipcc (gfx950 kernels) execute the same operation sequence (build: -ffp-contract=off, no fast-math).
 *
 * Why shared: the reference takes its eigen-decompositions and its SVD from nalgebra 0.30
 * (`try_symmetric_eigen(eps, max_iter)` at eight-point/src/lib.rs:49 and cv-core/src/pose.rs:272,
 * `SVD::try_new` at cv-pinhole/src/essential.rs:128), which is not vendored in the reference tree and
 * cannot be built here.  Its iteration (tridiagonalisation + implicit QR) is not restated; both sides use
 * this cyclic Jacobi iteration instead, so RANSAC parity is "oracle == HIP", bit for bit, while agreement
 * with the reference is at the level its own tests pin: residual < 1e-4 on exact data
 * (eight-point/tests/random.rs), pose recovery (cv-pinhole doc-tests).
 */
#ifndef AKZ_RANSAC_MATH_H
#define AKZ_RANSAC_MATH_H

#if defined(__HIPCC__) || defined(__HIP__)
#define AKZ_RM_FN __host__ __device__ static inline
#else
#define AKZ_RM_FN static inline
#endif

/* Babylonian-free: sqrt is the one non-arithmetic primitive; it is correctly rounded on both sides. */
#if defined(__HIP_DEVICE_COMPILE__)
#define AKZ_RM_SQRT(x) __dsqrt_rn(x)
#else
#include <math.h>
#define AKZ_RM_SQRT(x) sqrt(x)
#endif

/* Cyclic Jacobi for a symmetric N x N matrix stored row-major in a[N*N] (destroyed: ends diagonal);
 * v[N*N] receives the eigenvectors as COLUMNS (v[r*N + c] = component r of eigenvector c).
 * Stops when the off-diagonal sum of squares is <= eps^2 * (sum of squares of the diagonal) or after
 * max_sweeps.  Returns the number of sweeps used. */
#define AKZ_RM_DEFINE_JACOBI(NAME, N)                                                              \
    AKZ_RM_FN int NAME(double* a, double* v, int S, double eps, int max_sweeps)                           \
    {                                                                                              \
        for (int i = 0; i < (N); ++i)                                                              \
            for (int j = 0; j < (N); ++j) v[(i * (N) + j) * S] = (i == j) ? 1.0 : 0.0;                   \
        int sweep = 0;                                                                             \
        for (; sweep < max_sweeps; ++sweep) {                                                      \
            double off = 0.0, diag = 0.0;                                                          \
            for (int p = 0; p < (N); ++p) {                                                        \
                diag += a[(p * (N) + p) * S] * a[(p * (N) + p) * S];                                           \
                for (int q = p + 1; q < (N); ++q) off += a[(p * (N) + q) * S] * a[(p * (N) + q) * S];          \
            }                                                                                      \
            if (off <= eps * eps * diag || off == 0.0) break;                                      \
            for (int p = 0; p < (N)-1; ++p)                                                        \
                for (int q = p + 1; q < (N); ++q) {                                                \
                    double apq = a[(p * (N) + q) * S];                                                   \
                    if (apq == 0.0) continue;                                                      \
                    double app = a[(p * (N) + p) * S], aqq = a[(q * (N) + q) * S];                             \
                    double theta = (aqq - app) / (2.0 * apq);                                      \
                    double at = theta < 0.0 ? -theta : theta;                                      \
                    double t = 1.0 / (at + AKZ_RM_SQRT(theta * theta + 1.0));                      \
                    if (theta < 0.0) t = -t;                                                       \
                    double c = 1.0 / AKZ_RM_SQRT(t * t + 1.0);                                     \
                    double s = t * c;                                                              \
                    for (int k = 0; k < (N); ++k) { /* columns p and q */                          \
                        double akp = a[(k * (N) + p) * S], akq = a[(k * (N) + q) * S];                         \
                        a[(k * (N) + p) * S] = c * akp - s * akq;                                        \
                        a[(k * (N) + q) * S] = s * akp + c * akq;                                        \
                    }                                                                              \
                    for (int k = 0; k < (N); ++k) { /* rows p and q */                             \
                        double apk = a[(p * (N) + k) * S], aqk = a[(q * (N) + k) * S];                         \
                        a[(p * (N) + k) * S] = c * apk - s * aqk;                                        \
                        a[(q * (N) + k) * S] = s * apk + c * aqk;                                        \
                    }                                                                              \
                    for (int k = 0; k < (N); ++k) {                                                \
                        double vkp = v[(k * (N) + p) * S], vkq = v[(k * (N) + q) * S];                         \
                        v[(k * (N) + p) * S] = c * vkp - s * vkq;                                        \
                        v[(k * (N) + q) * S] = s * vkp + c * vkq;                                        \
                    }                                                                              \
                }                                                                                  \
        }                                                                                          \
        return sweep;                                                                              \
    }

AKZ_RM_DEFINE_JACOBI(akz_rm_jacobi3, 3)   /* 3 x 3: the SVD of E through E^T E (R2), Lambda Twist's eigen step */


/* The per-(pose, match) residual (R3) spends its time in a 4 x 4 eigen-decomposition, 40 M of them per scene of
 * BASELINE config 4, so that size gets its own form of the same cyclic Jacobi iteration: only the upper triangle
 * is rotated (a[i*4 + j], i <= j; the lower triangle is never read or written), the diagonal takes the closed
 * form a_pp - t a_pq / a_qq + t a_pq, and the rotation tangent comes from
 *   t = sgn(h) w / (|h| + sqrt(h^2 + w^2)),  w = 2 a_pq,  h = a_qq - a_pp
 * — the same t as sgn(theta) / (|theta| + sqrt(theta^2 + 1)) with theta = h / w, one division fewer.  Per rotation:
 * 2 divisions, 2 square roots and ~50 multiply-adds instead of 3, 2 and ~90.  Same sweep order (p, q ascending),
 * same stopping rule as AKZ_RM_DEFINE_JACOBI.  v[r*4 + c] = component r of eigenvector c. */
#define AKZ_RM_J4_ROT(P, Q, K1, K2)                                                                  \
    do {                                                                                             \
        const double apq = a[(P) * 4 + (Q)];                                                         \
        if (apq != 0.0) {                                                                            \
            const double h = a[(Q) * 4 + (Q)] - a[(P) * 4 + (P)], w = 2.0 * apq;                     \
            const double ah = h < 0.0 ? -h : h;                                                      \
            double t = w / (ah + AKZ_RM_SQRT(h * h + w * w));                                        \
            if (h < 0.0) t = -t;                                                                     \
            const double c = 1.0 / AKZ_RM_SQRT(t * t + 1.0), s = t * c;                              \
            a[(P) * 4 + (P)] = a[(P) * 4 + (P)] - t * apq;                                           \
            a[(Q) * 4 + (Q)] = a[(Q) * 4 + (Q)] + t * apq;                                           \
            a[(P) * 4 + (Q)] = 0.0;                                                                  \
            {                                                                                        \
                double* xp = &a[(K1) < (P) ? (K1) * 4 + (P) : (P) * 4 + (K1)];                       \
                double* xq = &a[(K1) < (Q) ? (K1) * 4 + (Q) : (Q) * 4 + (K1)];                       \
                const double akp = *xp, akq = *xq;                                                   \
                *xp = c * akp - s * akq;                                                             \
                *xq = s * akp + c * akq;                                                             \
            }                                                                                        \
            {                                                                                        \
                double* xp = &a[(K2) < (P) ? (K2) * 4 + (P) : (P) * 4 + (K2)];                       \
                double* xq = &a[(K2) < (Q) ? (K2) * 4 + (Q) : (Q) * 4 + (K2)];                       \
                const double akp = *xp, akq = *xq;                                                   \
                *xp = c * akp - s * akq;                                                             \
                *xq = s * akp + c * akq;                                                             \
            }                                                                                        \
            for (int k = 0; k < 4; ++k) {                                                            \
                const double vkp = v[k * 4 + (P)], vkq = v[k * 4 + (Q)];                             \
                v[k * 4 + (P)] = c * vkp - s * vkq;                                                  \
                v[k * 4 + (Q)] = s * vkp + c * vkq;                                                  \
            }                                                                                        \
        }                                                                                            \
    } while (0)

AKZ_RM_FN int akz_rm_jacobi4_sym(double* a, double* v, double eps, int max_sweeps)
{
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) v[i * 4 + j] = (i == j) ? 1.0 : 0.0;
    int sweep = 0;
    for (; sweep < max_sweeps; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int p = 0; p < 4; ++p) {
            diag += a[p * 4 + p] * a[p * 4 + p];
            for (int q = p + 1; q < 4; ++q) off += a[p * 4 + q] * a[p * 4 + q];
        }
        if (off <= eps * eps * diag || off == 0.0) break;
        AKZ_RM_J4_ROT(0, 1, 2, 3);
        AKZ_RM_J4_ROT(0, 2, 1, 3);
        AKZ_RM_J4_ROT(0, 3, 1, 2);
        AKZ_RM_J4_ROT(1, 2, 0, 3);
        AKZ_RM_J4_ROT(1, 3, 0, 2);
        AKZ_RM_J4_ROT(2, 3, 0, 1);
    }
    return sweep;
}

/* The eight-point hypothesis (R1) is one 9 x 9 symmetric eigen-decomposition per minimal sample, 2 M of them per
 * micro-batch of 256 frame pairs at vslam-sandbox's 8192 initialisation hypotheses.  Same iteration as
 * akz_rm_jacobi4_sym for N = 9: upper triangle only (a[i*9 + j], i <= j), closed-form diagonal, the rotated
 * element set to zero, t from (h, w) with one division.  Every index below is a compile-time constant once the
 * p / q / k loops are unrolled (AKZ_RM_UNROLL: hipcc only), so on the device the 45 + 81 doubles of a and v live in
 * registers — one hypothesis per lane, no LDS, no scratch.  v[r*9 + c] = component r of eigenvector c. */
#if defined(__HIPCC__) || defined(__HIP__)
#define AKZ_RM_UNROLL _Pragma("unroll")
#else
#define AKZ_RM_UNROLL
#endif
#define AKZ_RM_UT9(a, i, j) a[(i) < (j) ? (i) * 9 + (j) : (j) * 9 + (i)]
AKZ_RM_FN int akz_rm_jacobi9_sym(double* a, double* v, double eps, int max_sweeps)
{
    AKZ_RM_UNROLL
    for (int i = 0; i < 9; ++i) {
        AKZ_RM_UNROLL
        for (int j = 0; j < 9; ++j) v[i * 9 + j] = (i == j) ? 1.0 : 0.0;
    }
    int sweep = 0;
    for (; sweep < max_sweeps; ++sweep) {
        double off = 0.0, diag = 0.0;
        AKZ_RM_UNROLL
        for (int p = 0; p < 9; ++p) {
            diag += a[p * 9 + p] * a[p * 9 + p];
            AKZ_RM_UNROLL
            for (int q = p + 1; q < 9; ++q) off += a[p * 9 + q] * a[p * 9 + q];
        }
        if (off <= eps * eps * diag || off == 0.0) break;
        AKZ_RM_UNROLL
        for (int p = 0; p < 8; ++p) {
            AKZ_RM_UNROLL
            for (int q = p + 1; q < 9; ++q) {
                const double apq = a[p * 9 + q];
                if (apq != 0.0) {
                    const double h = a[q * 9 + q] - a[p * 9 + p], w = 2.0 * apq;
                    const double ah = h < 0.0 ? -h : h;
                    double t = w / (ah + AKZ_RM_SQRT(h * h + w * w));
                    if (h < 0.0) t = -t;
                    const double c = 1.0 / AKZ_RM_SQRT(t * t + 1.0), s = t * c;
                    a[p * 9 + p] = a[p * 9 + p] - t * apq;
                    a[q * 9 + q] = a[q * 9 + q] + t * apq;
                    a[p * 9 + q] = 0.0;
                    AKZ_RM_UNROLL
                    for (int k = 0; k < 9; ++k) {
                        if (k != p && k != q) {
                            const double akp = AKZ_RM_UT9(a, k, p), akq = AKZ_RM_UT9(a, k, q);
                            AKZ_RM_UT9(a, k, p) = c * akp - s * akq;
                            AKZ_RM_UT9(a, k, q) = s * akp + c * akq;
                        }
                    }
                    AKZ_RM_UNROLL
                    for (int k = 0; k < 9; ++k) {
                        const double vkp = v[k * 9 + p], vkq = v[k * 9 + q];
                        v[k * 9 + p] = c * vkp - s * vkq;
                        v[k * 9 + q] = s * vkp + c * vkq;
                    }
                }
            }
        }
    }
    return sweep;
}

#endif /* AKZ_RANSAC_MATH_H */
