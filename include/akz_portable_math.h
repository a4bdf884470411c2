/* akz_portable_math.h — the three transcendental functions on the AKAZE keypoint path,
 * specified as plain IEEE-754 double arithmetic (+,-,*,/ only, no FMA contraction, no libm).
 *
 * Why this exists: the reference computes keypoint orientation and the descriptor rotation with
 * Rust's f32::atan2 / f32::cos / f32::sin (akaze/src/scale_space_extrema.rs:242,284 and
 * akaze/src/descriptors.rs:70-71), which lower to the host libm's atan2f/cosf/sinf.  Their bits
 * depend on whichever libm the reference was linked against, so "bit-exact angle" is only defined
 * relative to one libm (SURVEY.md §7 hard part 2, §8d "Tolerances").  This header fixes ONE
 * definition that compiles to identical IEEE operations with gcc (the CPU oracle) and hipcc
 * (the gfx950 kernels): evaluate in double to ~1e-16, round once to float.  The result equals
 * the correctly rounded f32 value except when the true value lies within ~1e-16 relative of a
 * rounding midpoint, i.e. it agrees with glibc's sinf/cosf in all but ~1e-8 of the inputs and is
 * within 1 ulp of glibc 2.35's atan2f (tests/test_oracle_math.py measures both).
 *
 * Build contract: every translation unit including this header is compiled with
 * -ffp-contract=off and without -ffast-math.
 */
#ifndef AKZ_PORTABLE_MATH_H
#define AKZ_PORTABLE_MATH_H

#if defined(__HIPCC__) || defined(__HIP__)
#define AKZ_PM_FN __host__ __device__ static inline
#else
#define AKZ_PM_FN static inline
#endif

/* atan(k/8), k = 0..8, correctly rounded doubles. */
#define AKZ_PM_ATAN_TAB(k)                                                                       \
    ((k) == 0   ? 0.0                                                                            \
     : (k) == 1 ? 0x1.fd5ba9aac2f6ep-4                                                           \
     : (k) == 2 ? 0x1.f5b75f92c80ddp-3                                                           \
     : (k) == 3 ? 0x1.6f61941e4def1p-2                                                           \
     : (k) == 4 ? 0x1.dac670561bb4fp-2                                                           \
     : (k) == 5 ? 0x1.1e00babdefeb4p-1                                                           \
     : (k) == 6 ? 0x1.4978fa3269ee1p-1                                                           \
     : (k) == 7 ? 0x1.700a7c5784634p-1                                                           \
                : 0x1.921fb54442d18p-1)

/* atan(t) for 0 <= t <= 1.  t = c + d with c = k/8: atan(t) = atan(c) + atan((t-c)/(1+t*c)),
 * |x| <= 1/16 so the alternating Taylor series to x^15 leaves < 2^-60 relative. */
AKZ_PM_FN double akz_pm_atan01(double t)
{
    int k = (int)(t * 8.0 + 0.5);
    double c = (double)k * 0.125;
    double x = (t - c) / (1.0 + t * c);
    double x2 = x * x;
    double p = 1.0 / 15.0;
    p = 1.0 / 13.0 - x2 * p;
    p = 1.0 / 11.0 - x2 * p;
    p = 1.0 / 9.0 - x2 * p;
    p = 1.0 / 7.0 - x2 * p;
    p = 1.0 / 5.0 - x2 * p;
    p = 1.0 / 3.0 - x2 * p;
    p = 1.0 - x2 * p;
    return AKZ_PM_ATAN_TAB(k) + x * p;
}

/* atan2f with IEEE zero/sign conventions; inputs finite. Result in [-pi, pi], rounded once. */
AKZ_PM_FN float akz_pm_atan2f(float yf, float xf)
{
    const double PI = 0x1.921fb54442d18p+1;
    const double PIO2 = 0x1.921fb54442d18p+0;
    double y = (double)yf, x = (double)xf;
    double ay = y < 0.0 ? -y : y;
    double ax = x < 0.0 ? -x : x;
    int xneg = __builtin_signbit(xf);
    int yneg = __builtin_signbit(yf);
    /* one division: min / max of the magnitudes (the same operands the two branches of the textbook form
     * divide), folded about pi/4 afterwards; a zero numerator gives 0 (also atan2(0, 0), where 0 / 0 is
     * replaced) */
    const int steep = ay > ax;
    const double num = steep ? ax : ay, den = steep ? ay : ax;
    double a = akz_pm_atan01(den == 0.0 ? 0.0 : num / den);
    if (steep) a = PIO2 - a;
    if (ay == 0.0) a = 0.0;
    /* quadrant fold: x < 0 mirrors to pi - a (for |y| > |x| as well, about pi/2) */
    if (xneg) a = PI - a;
    if (yneg) a = -a;
    return (float)a;
}

/* Shared reduction: angle a (any finite f32 of modest size, |a| < 2^20) -> quadrant q, |r|<=pi/4. */
AKZ_PM_FN double akz_pm_reduce(double a, int* q)
{
    const double TWO_OVER_PI = 0x1.45f306dc9c883p-1;
    const double PIO2_HI = 0x1.921fb54400000p+0; /* 33 significant bits: k*HI exact for |k|<2^20 */
    const double PIO2_LO = 0x1.0b4611a626331p-34;
    double kf = a * TWO_OVER_PI;
    int k = (int)(kf < 0.0 ? kf - 0.5 : kf + 0.5);
    double r = (a - (double)k * PIO2_HI) - (double)k * PIO2_LO;
    *q = k & 3;
    return r;
}

AKZ_PM_FN double akz_pm_sin_poly(double r)
{
    double r2 = r * r;
    double p = -1.0 / 1307674368000.0;       /* r^15 */
    p = 1.0 / 6227020800.0 + r2 * p;          /* r^13 */
    p = -1.0 / 39916800.0 + r2 * p;           /* r^11 */
    p = 1.0 / 362880.0 + r2 * p;              /* r^9 */
    p = -1.0 / 5040.0 + r2 * p;               /* r^7 */
    p = 1.0 / 120.0 + r2 * p;                 /* r^5 */
    p = -1.0 / 6.0 + r2 * p;                  /* r^3 */
    return r + r * (r2 * p);
}

AKZ_PM_FN double akz_pm_cos_poly(double r)
{
    double r2 = r * r;
    double p = 1.0 / 20922789888000.0;        /* r^16 */
    p = -1.0 / 87178291200.0 + r2 * p;        /* r^14 */
    p = 1.0 / 479001600.0 + r2 * p;           /* r^12 */
    p = -1.0 / 3628800.0 + r2 * p;            /* r^10 */
    p = 1.0 / 40320.0 + r2 * p;               /* r^8 */
    p = -1.0 / 720.0 + r2 * p;                /* r^6 */
    p = 1.0 / 24.0 + r2 * p;                  /* r^4 */
    p = -0.5 + r2 * p;                        /* r^2 */
    return 1.0 + r2 * p;
}

AKZ_PM_FN float akz_pm_sinf(float af)
{
    int q;
    double r = akz_pm_reduce((double)af, &q);
    double v = (q & 1) ? akz_pm_cos_poly(r) : akz_pm_sin_poly(r);
    if (q & 2) v = -v;
    return (float)v;
}

AKZ_PM_FN float akz_pm_cosf(float af)
{
    int q;
    double r = akz_pm_reduce((double)af, &q);
    double v = (q & 1) ? akz_pm_sin_poly(r) : akz_pm_cos_poly(r);
    if (((q + 1) & 2)) v = -v;
    return (float)v;
}

#endif /* AKZ_PORTABLE_MATH_H */
