/* akz_p3p_math.h — Lambda Twist P3P and the world-to-camera residual, as plain
 * IEEE double arithmetic shared by the CPU oracle (gcc) and the gfx950 kernels (hipcc), both built with
 * -ffp-contract=off.  The solver is a closed-form f64 chain (cubic root by Newton, 2x2 quadratics,
 * Gauss-Newton polish): one source keeps the operation order identical on both sides.
 *
 * Restates, expression by expression (paths relative to rust-cv/cv):
 *   LambdaTwist::compute_poses_nordberg        lambda-twist/src/lib.rs:107-318
 *   gauss_newton_refine_lambda                 lambda-twist/src/lib.rs:361-409
 *   l1_norm, root2real, cube_root              lambda-twist/src/lib.rs:411-497
 *   eigen_decomposition_singular               lambda-twist/src/lib.rs:499-554
 *   WorldToCamera::residual                    cv-core/src/pose.rs:194-201 (+ Pose::transform :125-133,
 *                                              Projective::{from_homogeneous, point, bearing} point.rs:20-49)
 * Un-vendored nalgebra pieces and what replaces them (parity unpinned vs the reference, pinned only by its
 * tolerance tests, lambda-twist/tests/consensus.rs: pose within 1e-6):
 *   Matrix3::try_inverse          -> adjugate / determinant
 *   Rotation3::from_matrix_eps    -> polar projection R = M (M^T M)^(-1/2) through the shared Jacobi solver
 */
#ifndef AKZ_P3P_MATH_H
#define AKZ_P3P_MATH_H

#include "akz_ransac_math.h"

AKZ_RM_FN double akz_p3p_abs(double v) { return v < 0.0 ? -v : v; }
AKZ_RM_FN int akz_p3p_finite(double v) { return v == v && akz_p3p_abs(v) <= 1.7976931348623157e308; }

/* root2real — lib.rs:416-428.  returns real-ness; r1, r2 as the reference */
AKZ_RM_FN int akz_p3p_root2real(double b, double c, double* r1, double* r2)
{
    double discriminant = b * b - 4.0 * c;
    if (discriminant < 0.0) {
        *r1 = *r2 = 0.5 * b;
        return 0;
    } else if (b < 0.0) {
        double y = AKZ_RM_SQRT(discriminant);
        *r1 = 0.5 * (-b + y);
        *r2 = 0.5 * (-b - y);
        return 1;
    } else {
        double y = AKZ_RM_SQRT(discriminant);
        *r1 = 2.0 * c / (-b + y);
        *r2 = 2.0 * c / (-b - y);
        return 1;
    }
}

/* cube_root — lib.rs:451-497 */
AKZ_RM_FN double akz_p3p_cube_root(double b, double c, double d)
{
    double r0;
    if (b * b >= 3.0 * c) {
        double v = AKZ_RM_SQRT(b * b - 3.0 * c);
        double t1 = (-b - v) / 3.0;
        double k = ((t1 + b) * t1 + c) * t1 + d;
        if (k > 0.0) {
            r0 = t1 - AKZ_RM_SQRT(-k / (3.0 * t1 + b));
        } else {
            double t2 = (-b + v) / 3.0;
            k = ((t2 + b) * t2 + c) * t2 + d;
            r0 = t2 + AKZ_RM_SQRT(-k / (3.0 * t2 + b));
        }
    } else {
        r0 = -b / 3.0;
        if (akz_p3p_abs((3.0 * r0 + 2.0 * b) * r0 + c) < 1e-4) r0 += 1.0;
    }
    for (int i = 0; i < 7; ++i) {
        double fx = ((r0 + b) * r0 + c) * r0 + d;
        double fpx = (3.0 * r0 + 2.0 * b) * r0 + c;
        r0 -= fx / fpx;
    }
    for (int i = 0; i < 43; ++i) {
        double fx = ((r0 + b) * r0 + c) * r0 + d;
        if (akz_p3p_abs(fx) > 1e-13) {
            double fpx = (3.0 * r0 + 2.0 * b) * r0 + c;
            r0 -= fx / fpx;
        } else {
            break;
        }
    }
    return r0;
}

/* gauss_newton_refine_lambda — lib.rs:361-409 */
AKZ_RM_FN void akz_p3p_refine(double* l, int iterations, double a12, double a13, double a23, double b12, double b13,
                              double b23)
{
    double l1 = l[0], l2 = l[1], l3 = l[2];
    double r1 = l1 * l1 + l2 * l2 + b12 * l1 * l2 - a12;
    double r2 = l1 * l1 + l3 * l3 + b13 * l1 * l3 - a13;
    double r3 = l2 * l2 + l3 * l3 + b23 * l2 * l3 - a23;
    for (int it = 0; it < iterations; ++it) {
        if (akz_p3p_abs(r1) + akz_p3p_abs(r2) + akz_p3p_abs(r3) < 1e-10) break;
        double dr1dl1 = 2.0 * l1 + b12 * l2;
        double dr1dl2 = 2.0 * l2 + b12 * l1;
        double dr2dl1 = 2.0 * l1 + b13 * l3;
        double dr2dl3 = 2.0 * l3 + b13 * l1;
        double dr3dl2 = 2.0 * l2 + b23 * l3;
        double dr3dl3 = 2.0 * l3 + b23 * l2;
        double det = 1.0 / (-dr1dl1 * dr2dl3 * dr3dl2 - dr1dl2 * dr2dl1 * dr3dl3);
        /* jacobian (row-major rows) times res, nalgebra gemv order: column by column accumulation */
        double j00 = -dr2dl3 * dr3dl2, j01 = -dr1dl2 * dr3dl3, j02 = dr1dl2 * dr2dl3;
        double j10 = -dr2dl1 * dr3dl3, j11 = dr1dl1 * dr3dl3, j12 = -dr1dl1 * dr2dl3;
        double j20 = dr2dl1 * dr3dl2, j21 = -dr1dl1 * dr3dl2, j22 = -dr1dl2 * dr2dl1;
        double v0 = (j00 * r1 + j01 * r2) + j02 * r3;
        double v1 = (j10 * r1 + j11 * r2) + j12 * r3;
        double v2 = (j20 * r1 + j21 * r2) + j22 * r3;
        double n1 = l1 - det * v0, n2 = l2 - det * v1, n3 = l3 - det * v2;
        double q1 = n1 * n1 + n2 * n2 + b12 * n1 * n2 - a12;
        double q2 = n1 * n1 + n3 * n3 + b13 * n1 * n3 - a13;
        double q3 = n2 * n2 + n3 * n3 + b23 * n2 * n3 - a23;
        if (akz_p3p_abs(q1) + akz_p3p_abs(q2) + akz_p3p_abs(q3) > akz_p3p_abs(r1) + akz_p3p_abs(r2) + akz_p3p_abs(r3)) {
            break;
        } else {
            l1 = n1; l2 = n2; l3 = n3;
            r1 = q1; r2 = q2; r3 = q3;
        }
    }
    l[0] = l1; l[1] = l2; l[2] = l3;
}

/* R <- nearest rotation to M (row-major 3x3), replacing Rotation3::from_matrix_eps.  returns 0 on failure */
AKZ_RM_FN int akz_p3p_nearest_rotation(const double* M, double* R)
{
    double A[9], V[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += M[k * 3 + r] * M[k * 3 + c];
            A[r * 3 + c] = s;
        }
    akz_rm_jacobi3(A, V, 1, 1e-15, 100);
    /* (M^T M)^(-1/2) = V diag(1/sqrt(l)) V^T */
    double inv[3];
    for (int i = 0; i < 3; ++i) {
        double lam = A[i * 3 + i];
        if (!(lam > 0.0)) return 0;
        inv[i] = 1.0 / AKZ_RM_SQRT(lam);
    }
    double S[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += (V[r * 3 + k] * inv[k]) * V[c * 3 + k];
            S[r * 3 + c] = s;
        }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += M[r * 3 + k] * S[k * 3 + c];
            R[r * 3 + c] = s;
        }
    return 1;
}

/* LambdaTwist::compute_poses_nordberg — lib.rs:107-318.  bearings[3][3] unit vectors, world[3][4]
 * homogeneous points (Projective representation).  poses[4][12] row-major [R | t].  returns the count. */
AKZ_RM_FN int akz_p3p_poses(const double* bearings, const double* world, int gn_iterations, double* poses)
{
    double wp[3][3];
    for (int i = 0; i < 3; ++i) { /* Projective::point(): xyz / w, None when w == 0 (point at infinity) */
        double w = world[4 * i + 3];
        if (w == 0.0) return 0;
        for (int k = 0; k < 3; ++k) wp[i][k] = world[4 * i + k] / w;
    }
    const double* y1 = bearings;
    const double* y2 = bearings + 3;
    const double* y3 = bearings + 6;
    double d12[3], d13[3], d23[3];
    for (int k = 0; k < 3; ++k) {
        d12[k] = wp[0][k] - wp[1][k];
        d13[k] = wp[0][k] - wp[2][k];
        d23[k] = wp[1][k] - wp[2][k];
    }
    double d12xd13[3] = {d12[1] * d13[2] - d12[2] * d13[1], d12[2] * d13[0] - d12[0] * d13[2],
                         d12[0] * d13[1] - d12[1] * d13[0]};
    double a12 = (d12[0] * d12[0] + d12[1] * d12[1]) + d12[2] * d12[2];
    double a13 = (d13[0] * d13[0] + d13[1] * d13[1]) + d13[2] * d13[2];
    double a23 = (d23[0] * d23[0] + d23[1] * d23[1]) + d23[2] * d23[2];
    double c12 = (y1[0] * y2[0] + y1[1] * y2[1]) + y1[2] * y2[2];
    double c23 = (y2[0] * y3[0] + y2[1] * y3[1]) + y2[2] * y3[2];
    double c31 = (y3[0] * y1[0] + y3[1] * y1[1]) + y3[2] * y1[2];
    double blob = c12 * c23 * c31 - 1.0;
    double s12_sqr = 1.0 - c12 * c12;
    double s23_sqr = 1.0 - c23 * c23;
    double s31_sqr = 1.0 - c31 * c31;
    double b12 = -2.0 * c12;
    double b13 = -2.0 * c31;
    double b23 = -2.0 * c23;
    double p3 = a13 * (a23 * s31_sqr - a13 * s23_sqr);
    double p2 = 2.0 * blob * a23 * a13 + a13 * (2.0 * a12 + a13) * s23_sqr + a23 * (a23 - a12) * s31_sqr;
    double p1 = a23 * (a13 - a23) * s12_sqr - a12 * a12 * s23_sqr - 2.0 * a12 * (blob * a23 + a13 * s23_sqr);
    double p0 = a12 * (a12 * s23_sqr - a23 * s12_sqr);
    double g = akz_p3p_cube_root(p2 / p3, p1 / p3, p0 / p3);
    double m11 = a23 * (1.0 - g);
    double m12 = -(a23 * c12);
    double m13 = a23 * c31 * g;
    double m22 = a23 - a12 + a13 * g;
    double m23 = -c23 * (a13 * g - a12);
    double m33 = g * (a13 - a23) - a12;
    /* eigen_decomposition_singular — lib.rs:499-554 (x is symmetric: m21 = m12, ...) */
    double v3[3] = {m12 * m23 - m13 * m22, m13 * m12 - m23 * m11, m22 * m11 - m12 * m12};
    {
        double nrm = AKZ_RM_SQRT((v3[0] * v3[0] + v3[1] * v3[1]) + v3[2] * v3[2]);
        v3[0] = v3[0] / nrm; v3[1] = v3[1] / nrm; v3[2] = v3[2] / nrm;
    }
    double x12_sqr = m12 * m12;
    double eb = -m11 - m22 - m33;
    double ec = -x12_sqr - m13 * m13 - m23 * m23 + m11 * (m22 + m33) + m22 * m33;
    double e1, e2;
    akz_p3p_root2real(eb, ec, &e1, &e2);
    if (akz_p3p_abs(e1) < akz_p3p_abs(e2)) {
        double t = e1; e1 = e2; e2 = t;
    }
    double mx0011 = -m11 * m22;
    double prec_0 = m12 * m23 - m13 * m22;
    double prec_1 = m12 * m13 - m11 * m23;
    double ev[2][3];
    for (int i = 0; i < 2; ++i) {
        double e = i == 0 ? e1 : e2;
        double tmp = 1.0 / (e * (m11 + m22) + mx0011 - e * e + x12_sqr);
        double a1 = -(e * m13 + prec_0) * tmp;
        double a2 = -(e * m23 + prec_1) * tmp;
        double rnorm = 1.0 / AKZ_RM_SQRT(a1 * a1 + a2 * a2 + 1.0);
        a1 *= rnorm;
        a2 *= rnorm;
        ev[i][0] = a1; ev[i][1] = a2; ev[i][2] = rnorm;
    }
    /* eig_vectors = [v1 v2 v3] as columns: m11 = v1[0], m12 = v2[0], m21 = v1[1], m22 = v2[1], m31 = v1[2], m32 = v2[2] */
    double ratio0 = -e2 / e1;
    double eigen_ratio = AKZ_RM_SQRT(0.0 > ratio0 ? 0.0 : ratio0); /* 0.0_f64.max(x): NaN -> 0 */
    if (!(ratio0 == ratio0)) eigen_ratio = 0.0;
    double lambdas[4][3];
    int nl = 0;
    for (int sgn = 0; sgn < 2; ++sgn) {
        double ratio = sgn == 0 ? eigen_ratio : -eigen_ratio;
        double w2 = 1.0 / (ratio * ev[1][0] - ev[0][0]);
        double w0 = w2 * (ev[0][1] - ratio * ev[1][1]);
        double w1 = w2 * (ev[0][2] - ratio * ev[1][2]);
        double qa = 1.0 / ((a13 - a12) * w1 * w1 - a12 * b13 * w1 - a12);
        double qb = qa * (a13 * b12 * w1 - a12 * b13 * w0 - 2.0 * w0 * w1 * (a12 - a13));
        double qc = qa * ((a13 - a12) * w0 * w0 + a13 * b12 * w0 + a13);
        if (qb * qb - 4.0 * qc >= 0.0) {
            double tau[2];
            akz_p3p_root2real(qb, qc, &tau[0], &tau[1]);
            for (int ti = 0; ti < 2; ++ti) {
                double t = tau[ti];
                if (t > 0.0) {
                    double dd = a23 / (t * (b23 + t) + 1.0);
                    if (dd > 0.0) {
                        double l2 = AKZ_RM_SQRT(dd);
                        double l3 = t * l2;
                        double l1 = w0 * l2 + w1 * l3;
                        if (l1 >= 0.0 && nl < 4) {
                            lambdas[nl][0] = l1; lambdas[nl][1] = l2; lambdas[nl][2] = l3;
                            nl++;
                        }
                    }
                }
            }
        }
    }
    /* x_mat = [d12 d13 d12xd13] (columns), inverted by adjugate / determinant (replaces try_inverse) */
    double X[9] = {d12[0], d13[0], d12xd13[0], d12[1], d13[1], d12xd13[1], d12[2], d13[2], d12xd13[2]};
    double det = X[0] * (X[4] * X[8] - X[5] * X[7]) - X[1] * (X[3] * X[8] - X[5] * X[6]) + X[2] * (X[3] * X[7] - X[4] * X[6]);
    if (det == 0.0 || !akz_p3p_finite(det)) return 0;
    double Xi[9] = {(X[4] * X[8] - X[5] * X[7]) / det, (X[2] * X[7] - X[1] * X[8]) / det, (X[1] * X[5] - X[2] * X[4]) / det,
                    (X[5] * X[6] - X[3] * X[8]) / det, (X[0] * X[8] - X[2] * X[6]) / det, (X[2] * X[3] - X[0] * X[5]) / det,
                    (X[3] * X[7] - X[4] * X[6]) / det, (X[1] * X[6] - X[0] * X[7]) / det, (X[0] * X[4] - X[1] * X[3]) / det};
    int np = 0;
    for (int li = 0; li < nl; ++li) {
        double l[3] = {lambdas[li][0], lambdas[li][1], lambdas[li][2]};
        akz_p3p_refine(l, gn_iterations, a12, a13, a23, b12, b13, b23);
        double ry1[3], ry2[3], ry3[3], yd1[3], yd2[3];
        for (int k = 0; k < 3; ++k) {
            ry1[k] = l[0] * y1[k];
            ry2[k] = l[1] * y2[k];
            ry3[k] = l[2] * y3[k];
            yd1[k] = ry1[k] - ry2[k];
            yd2[k] = ry1[k] - ry3[k];
        }
        double yx[3] = {yd1[1] * yd2[2] - yd1[2] * yd2[1], yd1[2] * yd2[0] - yd1[0] * yd2[2], yd1[0] * yd2[1] - yd1[1] * yd2[0]};
        double Y[9] = {yd1[0], yd2[0], yx[0], yd1[1], yd2[1], yx[1], yd1[2], yd2[2], yx[2]};
        double rot[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                double s = 0.0;
                for (int k = 0; k < 3; ++k) s += Y[r * 3 + k] * Xi[k * 3 + c];
                rot[r * 3 + c] = s;
            }
        double trans[3];
        for (int r = 0; r < 3; ++r)
            trans[r] = ry1[r] - ((rot[r * 3 + 0] * wp[0][0] + rot[r * 3 + 1] * wp[0][1]) + rot[r * 3 + 2] * wp[0][2]);
        double R[9];
        if (!akz_p3p_nearest_rotation(rot, R)) continue;
        int ok = 1;
        for (int i = 0; i < 9; ++i) ok = ok && akz_p3p_finite(R[i]);
        for (int i = 0; i < 3; ++i) ok = ok && akz_p3p_finite(trans[i]);
        if (!ok) continue;
        double* P = poses + 12 * np;
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) P[r * 4 + c] = R[r * 3 + c];
            P[r * 4 + 3] = trans[r];
        }
        np++;
    }
    return np;
}

/* WorldToCamera::residual — cv-core/src/pose.rs:194-201 */
AKZ_RM_FN double akz_w2c_residual(const double* pose, const double* bearing, const double* world)
{
    double q[4];
    for (int r = 0; r < 3; ++r)
        q[r] = ((pose[r * 4 + 0] * world[0] + pose[r * 4 + 1] * world[1]) + pose[r * 4 + 2] * world[2]) + pose[r * 4 + 3] * world[3];
    q[3] = world[3];
    if (__builtin_signbit(q[3]))
        for (int i = 0; i < 4; ++i) q[i] = -q[i];
    double qn = AKZ_RM_SQRT((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]);
    double o0 = q[0] / qn, o1 = q[1] / qn, o2 = q[2] / qn;
    return 1.0 - ((bearing[0] * o0 + bearing[1] * o1) + bearing[2] * o2);
}

#endif /* AKZ_P3P_MATH_H */
