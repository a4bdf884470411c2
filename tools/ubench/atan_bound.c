/* The error BOUND behind kOriEps (cv_amd/csrc/akz_keypoints.hip: ori_sample_entry), by exhaustion where exhaustion is
 * possible and by the instructions' specifications elsewhere.
 *
 *   estimate:  t = fl(mn * rcp(mx));  s = fl(t t);  p = Horner in s with fused multiply-adds, 8 coefficients;  a = fl(t p);
 *              then up to three reflections  fl(C1 - a), fl(C2 - a), fl(C3 - a)  with C = fl(pi/2), fl(pi), fl(2 pi).
 *
 * Part 1 (exhaustive): for EVERY f32 t in [2^-13, 1] the value a(t) computed with correctly rounded f32 operations (fmaf is
 * one rounding, as v_fma_f32) against atan(t) in f64: E_poly = max |a(t) - atan t|, 109 051 905 arguments (`--all`: every f32
 * of [0, 1], 1 065 353 217 of them, half a minute).  Below 2^-13: s < 2^-26, so |p - 1| <= |c0 - 1| + 0.34 s + 2^-24 < 2e-7 and
 * |t p - atan t| <= t |p - 1| + t^3 / 3 + ulp(t) / 2 < 1e-10, which the program adds instead of walking those arguments.
 * Part 2 (specification): v_rcp_f32 is accurate to 1 ulp and the product rounds once, so t = (mn / mx)(1 + d), |d| <=
 * 2^-23 + 2^-24 + 2^-47; |atan'| <= 1 and t <= 1 give an angle error <= 1.5 * 2^-23 + 2^-47.  (mx in (1e-30, 1e30): the
 * reciprocal is a normal number; a product that underflows is off by < 2^-126.)
 * Part 3 (reflections): each adds |fl(C) - C| + half an ulp of its result: results <= pi/2, pi, 2 pi.
 * Part 4 (what the estimate is compared WITH): fast_atan2_equiv = fl(fl64->32(atan2 in f64) + fl(2 pi)) [- fl(2 pi), exact]:
 * half an ulp of a value <= pi, half an ulp of a sum <= 3 pi, and |fl(2 pi) - 2 pi| where it is not subtracted again.
 *
 * Prints every part and the total; exit code 1 if the total is not below kOriEps / 4.
 * usage: atan_bound c7 c6 c5 c4 c3 c2 c1 c0 kOriEps   (coefficients in the kernel's order: highest power first) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char** argv)
{
    const int all = argc == 11 && !strcmp(argv[10], "--all");
    if (argc != 10 && !all) return 2;
    float c[8];
    for (int i = 0; i < 8; ++i) c[i] = strtof(argv[1 + i], NULL);
    const double eps = strtod(argv[9], NULL);
    double worst = 0.0;
    uint32_t worst_bits = 0;
    const uint32_t one = 0x3F800000u;
#pragma omp parallel
    {
        double w = 0.0;
        uint32_t wb = 0;
#pragma omp for schedule(static) nowait
        for (int64_t b = all ? 0 : 0x39000000; b <= (int64_t)one; ++b) {
            uint32_t u = (uint32_t)b;
            float t;
            memcpy(&t, &u, 4);
            const float s = t * t;
            float p = c[0];
            for (int i = 1; i < 8; ++i) p = fmaf(p, s, c[i]);
            const float a = t * p;
            const double e = fabs((double)a - atan((double)t));
            if (e > w) { w = e; wb = u; }
        }
#pragma omp critical
        if (w > worst) { worst = w; worst_bits = wb; }
    }
    const double PI = 3.14159265358979323846;
    const double ulp_at = 1.0 / 8388608.0;   /* 2^-23: ulp of [1, 2) */
    const double e_t = 1.5 * ulp_at + ldexp(1.0, -47) + ldexp(1.0, -126);
    const double c1 = fabs((double)(float)(PI / 2) - PI / 2), c2 = fabs((double)(float)PI - PI), c3 = fabs((double)(float)(2 * PI) - 2 * PI);
    /* half ulps: results in [1, 2) -> 2^-24, [2, 4) -> 2^-23, [4, 8) -> 2^-22, [8, 16) -> 2^-21 */
    const double e_refl = (c1 + ldexp(1.0, -24)) + (c2 + ldexp(1.0, -23)) + (c3 + ldexp(1.0, -22));
    const double e_ref = ldexp(1.0, -23) /* f64 -> f32 of a value <= pi */ + 1e-15 /* the f64 evaluation */ + ldexp(1.0, -21) /* sum <= 3 pi */ + c3;
    if (!all) worst += 1e-10;                 /* t < 2^-13, bounded in the header */
    const double total = worst + e_t + e_refl + e_ref;
    printf("E_poly %.4e at t = 0x%08x  |  t %.4e  reflections %.4e  compared-with %.4e  |  TOTAL %.4e  kOriEps %.4e  ratio %.2f\n",
           worst, worst_bits, e_t, e_refl, e_ref, total, eps, eps / total);
    return total < eps / 4 ? 0 : 1;
}
