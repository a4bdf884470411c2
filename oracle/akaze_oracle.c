/* akaze_oracle.c — CPU restatement of the reference's AKAZE feature extractor.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  Nothing under cv_amd/
 * links, imports or calls it.
 *
 * It restates, function by function and in the reference's exact f32 evaluation order, the
 * `akaze` crate of rust-cv/cv (all citations relative to the reference checkout):
 *   image -> f32, half_size, separable filters, Gaussian   akaze/src/image.rs:45-109,154-199,202-389
 *   pyramid schedule, FED step sizes                       akaze/src/evolution.rs:46-126, fed_tau.rs:26-93
 *   contrast factor                                        akaze/src/contrast_factor.rs:16-64
 *   Scharr kernels                                         akaze/src/derivatives.rs:3-79
 *   conductivity + diffusion step                          akaze/src/nonlinear_diffusion.rs:14-83
 *   scale-space driver, extract                            akaze/src/lib.rs:193-258,309-339
 *   Hessian response                                       akaze/src/detector_response.rs:8-85
 *   extrema, sub-pixel refinement, orientation             akaze/src/scale_space_extrema.rs:14-362
 *   M-LDB descriptor                                       akaze/src/descriptors.rs:16-202
 *
 * The reference cannot be built here (no Rust toolchain, no vendored crates), so this restatement
 * is pinned by the reference's own known-answer tests instead (tests/test_oracle_pins.py):
 *   akaze/tests/estimate_pose.rs:41,42,59  — 399 / 343 descriptors, 11 Lowe-ratio matches on the two res/ PNGs
 *   akaze/src/image.rs:396-412             — gaussian_kernel(3.0, 7)
 *   akaze/src/image.rs:414-432             — filters == clamp-border correlation (+-1e-4)
 *
 * Arithmetic that lives in un-vendored third-party crates is selectable at run time
 * (orc_set_option) so the combination that reproduces the pins can be frozen:
 *   ORC_OPT_REDUCE   wide::f32x4::reduce_add order   0: ((a0+a1)+a2)+a3 [default]  1: (a0+a1)+(a2+a3)
 *   ORC_OPT_FMA      wide::f32x4::mul_add            0: unfused mul then add [default]  1: fmaf
 *   ORC_OPT_HALFSUM  ndarray sum() of a 2x2 window   0: (a+b)+(c+d) [default]  1: ((a+b)+c)+d
 *   ORC_OPT_TRIG     atan2f/cosf/sinf                0: include/akz_portable_math.h [default, what
 *                                                       the HIP path implements]  1: host libm (what
 *                                                       a Rust build on this host would call)
 * and one switch that changes the schedule, never a result:
 *   ORC_OPT_INTRA    the akaze crate's `rayon` feature     0: serial [default]  1: the reference's four intra-frame
 *                    parallel points (lib.rs:241-247 join of the two simple Scharr filters; detector_response.rs:21,54
 *                    par_iter over the evolutions with the joins of :71-83 nested; scale_space_extrema.rs:352 and
 *                    descriptors.rs:35 par_iter over the keypoints, collected in order) as OpenMP regions — only in a
 *                    build with -fopenmp (oracle/Makefile `fast`: bench.py's cpu_baseline_intra_frame leg); the
 *                    separable filters, the FED steps and the extrema search are serial in the reference as well.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).  No FMA contraction, no
 * reassociation: rustc never does either.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/akz.h"
#include "../include/akz_portable_math.h"

enum { ORC_OPT_REDUCE = 0, ORC_OPT_FMA = 1, ORC_OPT_HALFSUM = 2, ORC_OPT_TRIG = 3, ORC_OPT_INTRA = 4, ORC_NOPT = 5 };
static int g_opt[ORC_NOPT] = {0, 0, 0, 0, 0};

void orc_set_option(int which, int value)
{
    if (which >= 0 && which < ORC_NOPT) g_opt[which] = value;
}
int orc_get_option(int which) { return (which >= 0 && which < ORC_NOPT) ? g_opt[which] : -1; }

/* ------------------------------------------------------------------------------------------ */
/* images                                                                                      */

typedef struct {
    int w, h;
    float* d;
} img_t;

static img_t img_new(int w, int h)
{
    img_t r;
    r.w = w;
    r.h = h;
    r.d = (float*)calloc((size_t)w * (size_t)h + 1, sizeof(float));
    return r;
}
static void img_free(img_t* a)
{
    free(a->d);
    a->d = NULL;
    a->w = a->h = 0;
}
static img_t img_clone(const img_t* a)
{
    img_t r = img_new(a->w, a->h);
    memcpy(r.d, a->d, sizeof(float) * (size_t)a->w * (size_t)a->h);
    return r;
}

/* Rust `x as usize` from f32: truncate toward zero, saturating, NaN -> 0. */
static size_t sat_usize_f32(float v)
{
    if (!(v > 0.0f)) return 0;
    if (v >= 18446744073709551616.0f) return SIZE_MAX;
    return (size_t)v;
}
static long sat_isize_f32(float v)
{
    if (v != v) return 0;
    if (v >= 9223372036854775808.0f) return INT64_MAX;
    if (v <= -9223372036854775808.0f) return INT64_MIN;
    return (long)v;
}
static size_t sat_usize_f64(double v)
{
    if (!(v > 0.0)) return 0;
    if (v >= 18446744073709551616.0) return SIZE_MAX;
    return (size_t)v;
}

/* GrayFloatImage::from_dynamic, Luma8 arm — image.rs:47-56: f32::from(v) / 255f32. */
void orc_u8_to_f32(const uint8_t* in, int w, int h, int stride, float* out)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) out[(size_t)y * w + x] = (float)in[(size_t)y * stride + x] / 255.0f;
}

/* wide::f32x4 accumulate + reduce_add, image.rs:242-247 / :320-325. `win` has 4*nchunks readable
 * floats, `kpad` is the kernel zero-padded to 4*nchunks. */
static inline float lane4_dot(const float* win, const float* kpad, int nchunks)
{
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    if (g_opt[ORC_OPT_FMA]) {
        for (int c = 0; c < nchunks; ++c) {
            a0 = fmaf(win[4 * c + 0], kpad[4 * c + 0], a0);
            a1 = fmaf(win[4 * c + 1], kpad[4 * c + 1], a1);
            a2 = fmaf(win[4 * c + 2], kpad[4 * c + 2], a2);
            a3 = fmaf(win[4 * c + 3], kpad[4 * c + 3], a3);
        }
    } else {
        for (int c = 0; c < nchunks; ++c) {
            a0 = win[4 * c + 0] * kpad[4 * c + 0] + a0;
            a1 = win[4 * c + 1] * kpad[4 * c + 1] + a1;
            a2 = win[4 * c + 2] * kpad[4 * c + 2] + a2;
            a3 = win[4 * c + 3] * kpad[4 * c + 3] + a3;
        }
    }
    if (g_opt[ORC_OPT_REDUCE]) return (a0 + a1) + (a2 + a3);
    return ((a0 + a1) + a2) + a3;
}

/* horizontal_filter — image.rs:202-251. */
void orc_horizontal_filter(const float* in, int w, int h, const float* kernel, int ksize, float* out)
{
    int half = ksize / 2;
    int nchunks = (ksize + 3) / 4;
    int simd = ksize + 3; /* `4 * (kernel_size + 3) / 4` parses as (4*(k+3))/4 = k+3, image.rs:225 */
    int extra = simd - ksize;
    float kpad[4 * 32];
    float* kp = (nchunks <= 32) ? kpad : (float*)malloc(sizeof(float) * 4 * (size_t)nchunks);
    for (int i = 0; i < 4 * nchunks; ++i) kp[i] = i < ksize ? kernel[i] : 0.0f;
    size_t slen = (size_t)w + 2 * (size_t)half + (size_t)extra;
    float* scratch = (float*)malloc(sizeof(float) * (slen + 4));
    for (int y = 0; y < h; ++y) {
        const float* row = in + (size_t)y * w;
        for (int i = 0; i < half; ++i) scratch[i] = row[0];
        memcpy(scratch + half, row, sizeof(float) * (size_t)w);
        for (int i = 0; i < half; ++i) scratch[half + w + i] = row[w - 1];
        for (size_t i = (size_t)(2 * half + w); i < slen + 4; ++i) scratch[i] = 0.0f;
        float* orow = out + (size_t)y * w;
        for (int x = 0; x < w; ++x) orow[x] = lane4_dot(scratch + x, kp, nchunks);
    }
    free(scratch);
    if (kp != kpad) free(kp);
}

/* vertical_filter — image.rs:253-331 (same lane order along y; clamp top = row 0, bottom = row h-1). */
void orc_vertical_filter(const float* in, int w, int h, const float* kernel, int ksize, float* out)
{
    int half = ksize / 2;
    int nchunks = (ksize + 3) / 4;
    int extra = 3;
    float kpad[4 * 32];
    float* kp = (nchunks <= 32) ? kpad : (float*)malloc(sizeof(float) * 4 * (size_t)nchunks);
    for (int i = 0; i < 4 * nchunks; ++i) kp[i] = i < ksize ? kernel[i] : 0.0f;
    enum { SW = 16 };
    size_t sh = (size_t)h + 2 * (size_t)half + (size_t)extra;
    float* scratch = (float*)malloc(sizeof(float) * (SW * sh + 4));
    for (size_t i = 0; i < SW * sh + 4; ++i) scratch[i] = 0.0f;
    for (int xs = 0; xs < w; xs += SW) {
        int xe = xs + SW < w ? xs + SW : w;
        for (int x = xs; x < xe; ++x) {
            float* col = scratch + (size_t)(x - xs) * sh;
            for (int i = 0; i < half; ++i) col[i] = in[x];
            for (int y = 0; y < h; ++y) col[half + y] = in[(size_t)y * w + x];
            for (int i = 0; i < half; ++i) col[half + h + i] = in[(size_t)(h - 1) * w + x];
            for (int i = 0; i < extra; ++i) col[2 * half + h + i] = 0.0f;
        }
        for (int x = xs; x < xe; ++x) {
            const float* col = scratch + (size_t)(x - xs) * sh;
            for (int y = 0; y < h; ++y) out[(size_t)y * w + x] = lane4_dot(col + y, kp, nchunks);
        }
    }
    free(scratch);
    if (kp != kpad) free(kp);
}

/* separable_filter — image.rs:333-340: horizontal THEN vertical. */
static img_t separable_filter(const img_t* a, const float* hk, int hn, const float* vk, int vn)
{
    img_t t = img_new(a->w, a->h);
    img_t r = img_new(a->w, a->h);
    orc_horizontal_filter(a->d, a->w, a->h, hk, hn, t.d);
    orc_vertical_filter(t.d, a->w, a->h, vk, vn, r.d);
    img_free(&t);
    return r;
}

/* gaussian() + gaussian_kernel() — image.rs:349-374. */
static float gaussian_fn(float x, float r)
{
    const float PI_F = 3.14159274101257324219f;
    float a = 1.0f / (sqrtf(2.0f * PI_F) * r);
    float e = expf(-(x * x) / (2.0f * (r * r)));
    return a * e;
}
void orc_gaussian_kernel(float r, int ksize, float* out)
{
    int halfw = ksize / 2;
    float sum = 0.0f;
    for (int i = -halfw; i <= halfw; ++i) {
        float v = gaussian_fn((float)i, r);
        out[i + halfw] = v;
        sum += v;
    }
    for (int i = 0; i < ksize; ++i) out[i] /= sum;
}
/* gaussian_blur — image.rs:383-389. */
static img_t gaussian_blur(const img_t* a, float r)
{
    int radius = (int)sat_usize_f32(ceilf(2.0f * r));
    int ksize = radius * 2 + 1;
    float* k = (float*)malloc(sizeof(float) * (size_t)ksize);
    orc_gaussian_kernel(r, ksize, k);
    img_t o = separable_filter(a, k, ksize, k, ksize);
    free(k);
    return o;
}
void orc_gaussian_blur(const float* in, int w, int h, float r, float* out)
{
    img_t a = {w, h, (float*)in};
    img_t o = gaussian_blur(&a, r);
    memcpy(out, o.d, sizeof(float) * (size_t)w * h);
    img_free(&o);
}

/* GrayFloatImage::half_size — image.rs:154-199. */
static inline float sum4(float a, float b, float c, float d)
{
    if (g_opt[ORC_OPT_HALFSUM]) return ((a + b) + c) + d;
    return (a + b) + (c + d);
}
void orc_half_size(const float* in, int w, int h, float* out)
{
    int ow = w / 2, oh = h / 2;
    for (int y = 0; y < oh; ++y)
        for (int x = 0; x < ow; ++x) {
            const float* p = in + (size_t)(2 * y) * w + 2 * x;
            out[(size_t)y * ow + x] = sum4(p[0], p[1], p[w], p[w + 1]) * 0.25f;
        }
    if (oh * 2 != h && oh > 0) { /* last OUTPUT row overwritten from the last INPUT row, :168-175 */
        const float* p = in + (size_t)(h - 1) * w;
        for (int x = 0; x < ow; ++x) out[(size_t)(oh - 1) * ow + x] = (p[2 * x] + p[2 * x + 1]) * 0.5f;
    }
    if (ow * 2 != w && ow > 0) { /* last OUTPUT column from the last INPUT column, :178-185 */
        for (int y = 0; y < oh; ++y)
            out[(size_t)y * ow + (ow - 1)] =
                (in[(size_t)(2 * y) * w + (w - 1)] + in[(size_t)(2 * y + 1) * w + (w - 1)]) * 0.5f;
    }
    if (ow * 2 != w && oh * 2 != h && ow > 0 && oh > 0)
        out[(size_t)(oh - 1) * ow + (ow - 1)] = in[(size_t)(h - 1) * w + (w - 1)];
}

/* ------------------------------------------------------------------------------------------ */
/* derivatives.rs                                                                               */

static img_t simple_scharr_horizontal(const img_t* a)
{
    const float hk[3] = {-1.f, 0.f, 1.f}, vk[3] = {3.f, 10.f, 3.f};
    return separable_filter(a, hk, 3, vk, 3);
}
static img_t simple_scharr_vertical(const img_t* a)
{
    const float hk[3] = {3.f, 10.f, 3.f}, vk[3] = {-1.f, 0.f, 1.f};
    return separable_filter(a, hk, 3, vk, 3);
}
/* computer_scharr_kernel — derivatives.rs:57-79. order 0 = Main, 1 = Off. */
void orc_scharr_kernel(uint32_t sigma_size, int order, float* kernel /* 3+2*(sigma-1) */)
{
    double w = 10.0 / 3.0;
    float norm = (float)(1.0 / (2.0 * (double)sigma_size * (w + 2.0)));
    float middle = norm * (float)w;
    int ksize = 3 + 2 * ((int)sigma_size - 1);
    for (int i = 0; i < ksize; ++i) kernel[i] = 0.0f;
    if (order == 0) {
        kernel[0] = -1.0f;
        kernel[ksize - 1] = 1.0f;
    } else {
        kernel[0] = norm;
        kernel[ksize / 2] = middle;
        kernel[ksize - 1] = norm;
    }
}
static img_t scharr_horizontal(const img_t* a, uint32_t sigma)
{
    if (sigma == 1) return simple_scharr_horizontal(a);
    float mk[64], ok[64];
    int ks = 3 + 2 * ((int)sigma - 1);
    orc_scharr_kernel(sigma, 0, mk);
    orc_scharr_kernel(sigma, 1, ok);
    return separable_filter(a, mk, ks, ok, ks);
}
static img_t scharr_vertical(const img_t* a, uint32_t sigma)
{
    if (sigma == 1) return simple_scharr_vertical(a);
    float mk[64], ok[64];
    int ks = 3 + 2 * ((int)sigma - 1);
    orc_scharr_kernel(sigma, 0, mk);
    orc_scharr_kernel(sigma, 1, ok);
    return separable_filter(a, ok, ks, mk, ks);
}
void orc_scharr(const float* in, int w, int h, uint32_t sigma, int vertical, float* out)
{
    img_t a = {w, h, (float*)in};
    img_t o = vertical ? scharr_vertical(&a, sigma) : scharr_horizontal(&a, sigma);
    memcpy(out, o.d, sizeof(float) * (size_t)w * h);
    img_free(&o);
}

/* ------------------------------------------------------------------------------------------ */
/* contrast_factor.rs:16-64                                                                     */

double orc_contrast_factor(const float* image, int w, int h, double percentile, double grad_scale,
                           uint64_t num_bins)
{
    img_t im = {w, h, (float*)image};
    img_t g = gaussian_blur(&im, (float)grad_scale);
    img_t Lx = simple_scharr_horizontal(&g);
    img_t Ly = simple_scharr_vertical(&g);
    uint64_t* hist = (uint64_t*)calloc(num_bins ? num_bins : 1, sizeof(uint64_t));
    double hmax2 = -INFINITY;
    for (int y = 1; y < h - 1; ++y)
        for (int x = 1; x < w - 1; ++x) {
            float lx = Lx.d[(size_t)y * w + x], ly = Ly.d[(size_t)y * w + x];
            double v = (double)(lx * lx) + (double)(ly * ly);
            if (v > hmax2) hmax2 = v;
        }
    double hmax = sqrt(hmax2);
    double num_points = 0.0;
    for (int y = 1; y < h - 1; ++y)
        for (int x = 1; x < w - 1; ++x) {
            float lx = Lx.d[(size_t)y * w + x], ly = Ly.d[(size_t)y * w + x];
            double modg = sqrt((double)(lx * lx) + (double)(ly * ly));
            if (modg != 0.0) {
                size_t bin = sat_usize_f64(floor((double)num_bins * (modg / hmax)));
                if (bin == num_bins) bin -= 1;
                if (bin < num_bins) hist[bin] += 1; /* the reference would panic beyond; unreachable */
                num_points += 1.0;
            }
        }
    size_t threshold = sat_usize_f64(num_points * percentile);
    size_t k = 0, num_elements = 0;
    while (num_elements < threshold && k < num_bins) {
        num_elements += hist[k];
        k += 1;
    }
    double result = (num_elements >= threshold) ? hmax * (double)k / (double)num_bins : 0.03;
    free(hist);
    img_free(&g);
    img_free(&Lx);
    img_free(&Ly);
    return result;
}

/* ------------------------------------------------------------------------------------------ */
/* fed_tau.rs:26-93                                                                             */

static int is_prime_u64(uint64_t n)
{
    if (n < 2) return 0;
    for (uint64_t d = 2; d * d <= n; ++d)
        if (n % d == 0) return 0;
    return 1;
}
/* returns n; tau must hold n doubles (call with tau=NULL to query n). */
int orc_fed_tau_by_process_time(double T, int M, double tau_max, int reordering, double* tau_out, int cap)
{
    const double PI = 3.14159265358979323846;
    double t = T / (double)M;
    size_t n = sat_usize_f64(ceil(sqrt(3.0 * t / tau_max + 0.25) - 0.5 - 1.0e-8) + 0.5);
    double scale = 3.0 * t / (tau_max * (double)(n * (n + 1)));
    if (!tau_out) return (int)n;
    if ((int)n > cap) return -(int)n;
    double* tau = (double*)malloc(sizeof(double) * (n ? n : 1));
    for (size_t k = 0; k < n; ++k) {
        double c = 1.0 / (4.0 * (double)n + 2.0);
        double d = scale * tau_max / 2.0;
        double hh = cos(PI * (2.0 * (double)k + 1.0) * c);
        tau[k] = d / (hh * hh);
    }
    if (reordering && n > 0) {
        size_t kappa = n / 2;
        size_t prime = n + 1;
        while (!is_prime_u64(prime)) prime += 1;
        size_t k = 0;
        for (size_t l = 0; l < n; ++l) {
            size_t index = ((k + 1) * kappa) % prime - 1; /* wraps like release-mode usize */
            while (index >= n) {
                k += 1;
                index = ((k + 1) * kappa) % prime - 1;
            }
            k += 1;
            tau_out[l] = tau[index];
        }
    } else {
        for (size_t k = 0; k < n; ++k) tau_out[k] = tau[k];
    }
    free(tau);
    return (int)n;
}

/* ------------------------------------------------------------------------------------------ */
/* evolution.rs                                                                                 */

typedef struct {
    double etime, esigma;
    uint32_t octave, sublevel, sigma_size;
    img_t Lt, Lsmooth, Lx, Ly, Lxx, Lyy, Lxy, Lflow, Ldet;
    int ntau;
    double* tau;
} evo_t;

typedef struct orc_ctx {
    akz_config cfg;
    int w, h;
    int nlev;
    evo_t* ev;
    double contrast0;
    /* keypoint lists at each stage */
    akz_keypoint* kp_extrema;
    uint32_t n_extrema;
    akz_keypoint* kp_refined;
    uint32_t n_refined;
    akz_keypoint* kp_sorted;
    uint32_t n_sorted;
    akz_keypoint* kp_final;
    akz_descriptor* desc_final;
    uint32_t n_final;
    uint32_t n_candidates; /* raw local maxima above threshold, before suppression */
} orc_ctx;

void orc_config_default(akz_config* c)
{
    c->maximum_features = UINT64_MAX;
    c->num_sublevels = 4;
    c->max_octave_evolution = 4;
    c->base_scale_offset = 1.6;
    c->initial_contrast = 0.001;
    c->contrast_percentile = 0.7;
    c->contrast_factor_num_bins = 300;
    c->derivative_factor = 1.5;
    c->detector_threshold = 0.001;
    c->descriptor_channels = 3;
    c->descriptor_pattern_size = 10;
}

static void evo_free_images(evo_t* e)
{
    img_free(&e->Lt);
    img_free(&e->Lsmooth);
    img_free(&e->Lx);
    img_free(&e->Ly);
    img_free(&e->Lxx);
    img_free(&e->Lyy);
    img_free(&e->Lxy);
    img_free(&e->Lflow);
    img_free(&e->Ldet);
}

/* Akaze::allocate_evolutions — evolution.rs:80-126 (+ EvolutionStep::new :46-70). */
orc_ctx* orc_create(const akz_config* cfg, int width, int height)
{
    orc_ctx* c = (orc_ctx*)calloc(1, sizeof(orc_ctx));
    c->cfg = *cfg;
    c->w = width;
    c->h = height;
    int cap = (int)cfg->max_octave_evolution * (int)(cfg->num_sublevels ? cfg->num_sublevels : 1) + 1;
    c->ev = (evo_t*)calloc((size_t)cap, sizeof(evo_t));
    int n = 0;
    for (uint32_t octave = 0; octave < cfg->max_octave_evolution; ++octave) {
        double rfactor = pow(2.0, -(double)(int)octave); /* exact power of two */
        uint32_t lh = (uint32_t)((double)height * rfactor);
        uint32_t lw = (uint32_t)((double)width * rfactor);
        uint32_t smallest = lw < lh ? lw : lh;
        if (smallest < 40) continue; /* filter_map -> None; later octaves are smaller still */
        uint32_t sublevels = smallest < 80 ? 1 : cfg->num_sublevels;
        for (uint32_t s = 0; s < sublevels; ++s) {
            evo_t* e = &c->ev[n++];
            e->esigma = cfg->base_scale_offset *
                        pow(2.0, (double)s / (double)cfg->num_sublevels + (double)octave);
            e->etime = 0.5 * (e->esigma * e->esigma);
            e->octave = octave;
            e->sublevel = s;
            e->sigma_size = (uint32_t)round(e->esigma);
        }
    }
    c->nlev = n;
    for (int i = 1; i < n; ++i) {
        double ttime = c->ev[i].etime - c->ev[i - 1].etime;
        int nt = orc_fed_tau_by_process_time(ttime, 1, 0.25, 1, NULL, 0);
        c->ev[i].tau = (double*)malloc(sizeof(double) * (size_t)(nt ? nt : 1));
        c->ev[i].ntau = orc_fed_tau_by_process_time(ttime, 1, 0.25, 1, c->ev[i].tau, nt);
    }
    return c;
}

static void free_results(orc_ctx* c)
{
    free(c->kp_extrema);
    free(c->kp_refined);
    free(c->kp_sorted);
    free(c->kp_final);
    free(c->desc_final);
    c->kp_extrema = c->kp_refined = c->kp_sorted = c->kp_final = NULL;
    c->desc_final = NULL;
    c->n_extrema = c->n_refined = c->n_sorted = c->n_final = 0;
}

void orc_destroy(orc_ctx* c)
{
    if (!c) return;
    for (int i = 0; i < c->nlev; ++i) {
        evo_free_images(&c->ev[i]);
        free(c->ev[i].tau);
    }
    free(c->ev);
    free_results(c);
    free(c);
}

int orc_num_levels(const orc_ctx* c) { return c->nlev; }

static uint32_t deriv_sigma(const orc_ctx* c, const evo_t* e)
{
    double ratio = pow(2.0, (double)(int)e->octave);
    return (uint32_t)round(e->esigma * c->cfg.derivative_factor / ratio); /* detector_response.rs:11-13 */
}

int orc_level(const orc_ctx* c, int lvl, akz_level_info* out)
{
    if (lvl < 0 || lvl >= c->nlev) return -1;
    const evo_t* e = &c->ev[lvl];
    int w = c->w, h = c->h;
    for (uint32_t o = 0; o < e->octave; ++o) {
        w /= 2;
        h /= 2;
    }
    out->width = w;
    out->height = h;
    out->octave = e->octave;
    out->sublevel = e->sublevel;
    out->esigma = e->esigma;
    out->etime = e->etime;
    out->n_fed_steps = (uint32_t)e->ntau;
    out->deriv_sigma = deriv_sigma(c, e);
    return 0;
}
int orc_fed_tau(const orc_ctx* c, int lvl, double* tau, int cap)
{
    if (lvl < 0 || lvl >= c->nlev) return -1;
    int n = c->ev[lvl].ntau;
    for (int i = 0; i < n && i < cap; ++i) tau[i] = c->ev[lvl].tau[i];
    return n;
}

/* ------------------------------------------------------------------------------------------ */
/* nonlinear_diffusion.rs                                                                       */

/* pm_g2 — :70-83. */
void orc_pm_g2(const float* Lx, const float* Ly, size_t n, double k, float* out)
{
    float inverse_k = (float)(1.0 / (k * k));
    for (size_t i = 0; i < n; ++i) {
        float x = Lx[i], y = Ly[i];
        out[i] = 1.0f / (1.0f + inverse_k * (x * x + y * y));
    }
}
/* calculate_step — :14-58.  In place on L; flows are computed from the pre-update L (Jacobi),
 * then applied Left(+hf[x]) , Right(-hf[x-1]), Up(+vf[y]), Down(-vf[y-1]) in that order. */
void orc_fed_step(float* L, const float* c, int w, int h, float step_size)
{
    size_t n = (size_t)w * h;
    float* hf = (float*)malloc(sizeof(float) * n);
    float* vf = (float*)malloc(sizeof(float) * n);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w - 1; ++x) {
            size_t i = (size_t)y * w + x;
            hf[i] = 0.5f * step_size * (c[i] + c[i + 1]) * (L[i + 1] - L[i]);
        }
    for (int y = 0; y < h - 1; ++y)
        for (int x = 0; x < w; ++x) {
            size_t i = (size_t)y * w + x;
            vf[i] = 0.5f * step_size * (c[i] + c[i + w]) * (L[i + w] - L[i]);
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            size_t i = (size_t)y * w + x;
            float v = L[i];
            if (x < w - 1) v += hf[i];
            if (x > 0) v -= hf[i - 1];
            if (y < h - 1) v += vf[i];
            if (y > 0) v -= vf[i - w];
            L[i] = v;
        }
    free(hf);
    free(vf);
}

/* ------------------------------------------------------------------------------------------ */
/* lib.rs:193-258 create_nonlinear_scale_space                                                  */

static void create_nonlinear_scale_space(orc_ctx* c, const img_t* image)
{
    evo_t* ev = c->ev;
    for (int i = 0; i < c->nlev; ++i) evo_free_images(&ev[i]);
    ev[0].Lt = gaussian_blur(image, (float)c->cfg.base_scale_offset);
    ev[0].Lsmooth = img_clone(&ev[0].Lt);
    double contrast = orc_contrast_factor(image->d, image->w, image->h, c->cfg.contrast_percentile,
                                          1.0, c->cfg.contrast_factor_num_bins);
    c->contrast0 = contrast;
    for (int i = 1; i < c->nlev; ++i) {
        if (ev[i].octave > ev[i - 1].octave) {
            ev[i].Lt = img_new(ev[i - 1].Lt.w / 2, ev[i - 1].Lt.h / 2);
            orc_half_size(ev[i - 1].Lt.d, ev[i - 1].Lt.w, ev[i - 1].Lt.h, ev[i].Lt.d);
            contrast *= 0.75;
        } else {
            ev[i].Lt = img_clone(&ev[i - 1].Lt);
        }
        ev[i].Lsmooth = gaussian_blur(&ev[i].Lt, 1.0f);
        if (g_opt[ORC_OPT_INTRA]) {                     /* lib.rs:241-247: rayon::join of the two filters */
#pragma omp parallel sections num_threads(2)
            {
#pragma omp section
                ev[i].Lx = simple_scharr_horizontal(&ev[i].Lsmooth);
#pragma omp section
                ev[i].Ly = simple_scharr_vertical(&ev[i].Lsmooth);
            }
        } else {
            ev[i].Lx = simple_scharr_horizontal(&ev[i].Lsmooth);
            ev[i].Ly = simple_scharr_vertical(&ev[i].Lsmooth);
        }
        ev[i].Lflow = img_new(ev[i].Lt.w, ev[i].Lt.h);
        orc_pm_g2(ev[i].Lx.d, ev[i].Ly.d, (size_t)ev[i].Lt.w * ev[i].Lt.h, contrast, ev[i].Lflow.d);
        for (int j = 0; j < ev[i].ntau; ++j)
            orc_fed_step(ev[i].Lt.d, ev[i].Lflow.d, ev[i].Lt.w, ev[i].Lt.h, (float)ev[i].tau[j]);
    }
}

/* detector_response.rs:8-85 */
static void detector_response(orc_ctx* c)
{
    const int intra = g_opt[ORC_OPT_INTRA];
#ifdef _OPENMP
    if (intra) omp_set_max_active_levels(2);            /* the joins of :71-83 nest inside the par_iter of :21 */
#endif
#pragma omp parallel for schedule(dynamic, 1) if (intra)
    for (int i = 0; i < c->nlev; ++i) {
        evo_t* e = &c->ev[i];
        uint32_t sigma = deriv_sigma(c, e);
        img_free(&e->Lx);
        img_free(&e->Ly);
        if (intra) {
#pragma omp parallel sections num_threads(2)
            {
#pragma omp section
                e->Lx = scharr_horizontal(&e->Lsmooth, sigma);
#pragma omp section
                e->Ly = scharr_vertical(&e->Lsmooth, sigma);
            }
#pragma omp parallel sections num_threads(3)
            {
#pragma omp section
                e->Lxx = scharr_horizontal(&e->Lx, sigma);
#pragma omp section
                e->Lyy = scharr_vertical(&e->Ly, sigma);
#pragma omp section
                e->Lxy = scharr_vertical(&e->Lx, sigma);
            }
        } else {
            e->Lx = scharr_horizontal(&e->Lsmooth, sigma);
            e->Ly = scharr_vertical(&e->Lsmooth, sigma);
            e->Lxx = scharr_horizontal(&e->Lx, sigma);
            e->Lyy = scharr_vertical(&e->Ly, sigma);
            e->Lxy = scharr_vertical(&e->Lx, sigma);
        }
    }
#pragma omp parallel for schedule(dynamic, 1) if (intra)
    for (int i = 0; i < c->nlev; ++i) {
        evo_t* e = &c->ev[i];
        double ratio = pow(2.0, (double)(int)e->octave);
        double sigma_size = round(e->esigma * c->cfg.derivative_factor / ratio);
        float quat = (float)(sigma_size * sigma_size * sigma_size * sigma_size);
        e->Ldet = img_new(e->Lxx.w, e->Lxx.h);
        size_t n = (size_t)e->Lxx.w * e->Lxx.h;
        for (size_t p = 0; p < n; ++p) {
            float lxx = e->Lxx.d[p], lyy = e->Lyy.d[p], lxy = e->Lxy.d[p];
            e->Ldet.d[p] = (lxx * lyy - lxy * lxy) * quat;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* scale_space_extrema.rs                                                                       */

typedef struct {
    akz_keypoint* v;
    uint32_t n, cap;
} kpvec;
static void kpvec_push(kpvec* a, akz_keypoint k)
{
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 1024;
        a->v = (akz_keypoint*)realloc(a->v, sizeof(akz_keypoint) * a->cap);
    }
    a->v[a->n++] = k;
}

/* find_scale_space_extrema — :14-143. */
static void find_scale_space_extrema(orc_ctx* c, kpvec* out)
{
    kpvec cache = {0, 0, 0};
    const float smax = 10.0f * sqrtf(2.0f);
    const float thr = (float)c->cfg.detector_threshold;
    c->n_candidates = 0;
    for (int e_id = 0; e_id < c->nlev; ++e_id) {
        const evo_t* ev = &c->ev[e_id];
        int w = ev->Ldet.w, h = ev->Ldet.h;
        const float* D = ev->Ldet.d;
        for (int y = 1; y < h - 1; ++y)
            for (int x = 1; x < w - 1; ++x) {
                const float* p = D + (size_t)y * w + x;
                float v = *p;
                if (!(v > thr && v > p[-w - 1] && v > p[-w] && v > p[-w + 1] && v > p[-1] && v > p[1] &&
                      v > p[w - 1] && v > p[w] && v > p[w + 1]))
                    continue;
                c->n_candidates++;
                akz_keypoint kp;
                kp.response = fabsf(v);
                kp.size = (float)(ev->esigma * c->cfg.derivative_factor);
                kp.octave = ev->octave;
                kp.class_id = (uint32_t)e_id;
                kp.x = (float)x;
                kp.y = (float)y;
                kp.angle = 0.0f;
                float ratio = ldexpf(1.0f, (int)ev->octave); /* f32::powf(2, octave): exact */
                float sigma_size = roundf(kp.size / ratio);
                uint32_t id_repeated = 0;
                int is_repeated = 0, is_extremum = 1;
                for (uint32_t k = 0; k < cache.n; ++k) {
                    const akz_keypoint* q = &cache.v[k];
                    if (kp.class_id == q->class_id || (kp.class_id != 0 && kp.class_id - 1 == q->class_id)) {
                        float dx = kp.x * ratio - q->x;
                        float dy = kp.y * ratio - q->y;
                        float dist = dx * dx + dy * dy;
                        if (dist <= kp.size * kp.size) {
                            if (kp.response > q->response) {
                                id_repeated = k;
                                is_repeated = 1;
                            } else {
                                is_extremum = 0;
                            }
                            break;
                        }
                    }
                }
                if (!is_extremum) continue;
                float left_x = roundf(kp.x - smax * sigma_size) - 1.0f;
                float right_x = roundf(kp.x + smax * sigma_size) + 1.0f;
                float up_y = roundf(kp.y - smax * sigma_size) - 1.0f;
                float down_y = roundf(kp.y + smax * sigma_size) + 1.0f;
                int is_out = left_x < 0.0f || right_x >= (float)w || up_y < 0.0f || down_y >= (float)h;
                if (is_out) continue;
                kp.x = kp.x * ratio + 0.5f * (ratio - 1.0f);
                kp.y = kp.y * ratio + 0.5f * (ratio - 1.0f);
                if (!is_repeated)
                    kpvec_push(&cache, kp);
                else
                    cache.v[id_repeated] = kp;
            }
    }
    /* second pass — :121-140 */
    for (uint32_t i = 0; i < cache.n; ++i) {
        akz_keypoint ki = cache.v[i];
        int rep = 0;
        for (uint32_t j = i + 1; j < cache.n; ++j) {
            const akz_keypoint* kj = &cache.v[j];
            if (ki.class_id + 1 == kj->class_id) {
                float dx = ki.x - kj->x, dy = ki.y - kj->y;
                float dist = dx * dx + dy * dy;
                if (dist <= ki.size * ki.size && ki.response <= kj->response) {
                    rep = 1;
                    break;
                }
            }
        }
        if (!rep) kpvec_push(out, ki);
    }
    free(cache.v);
}

/* GAUSS25 — scale_space_extrema.rs:162-226 (a 7x7 quadrant of the SURF Gaussian table). */
static const float GAUSS25[7][7] = {
    {0.02546481f, 0.02350698f, 0.01849125f, 0.01239505f, 0.00708017f, 0.00344629f, 0.00142946f},
    {0.02350698f, 0.02169968f, 0.01706957f, 0.01144208f, 0.00653582f, 0.00318132f, 0.00131956f},
    {0.01849125f, 0.01706957f, 0.01342740f, 0.00900066f, 0.00514126f, 0.00250252f, 0.00103800f},
    {0.01239505f, 0.01144208f, 0.00900066f, 0.00603332f, 0.00344629f, 0.00167749f, 0.00069579f},
    {0.00708017f, 0.00653582f, 0.00514126f, 0.00344629f, 0.00196855f, 0.00095820f, 0.00039744f},
    {0.00344629f, 0.00318132f, 0.00250252f, 0.00167749f, 0.00095820f, 0.00046640f, 0.00019346f},
    {0.00142946f, 0.00131956f, 0.00103800f, 0.00069579f, 0.00039744f, 0.00019346f, 0.00008024f},
};

static const float PI_F = 3.14159274101257324219f;

static float atan2_sel(float y, float x) { return g_opt[ORC_OPT_TRIG] ? atan2f(y, x) : akz_pm_atan2f(y, x); }
static float cos_sel(float a) { return g_opt[ORC_OPT_TRIG] ? cosf(a) : akz_pm_cosf(a); }
static float sin_sel(float a) { return g_opt[ORC_OPT_TRIG] ? sinf(a) : akz_pm_sinf(a); }

/* cv_fast_atan2_equiv — :242: (atan2(y,x) + 2pi).rem_euclid(2pi), all f32. */
static float fast_atan2_equiv(float y, float x)
{
    float two_pi = 2.0f * PI_F;
    float a = atan2_sel(y, x) + two_pi;
    float r = fmodf(a, two_pi);
    if (r < 0.0f) r += fabsf(two_pi);
    return r;
}

/* compute_main_orientation — :229-288.  Returns 0 on success, -1 if a sample falls outside the
 * level image (the reference would panic in ImageBuffer::get_pixel). */
static int compute_main_orientation(const orc_ctx* c, akz_keypoint* kp)
{
    float res_x[109], res_y[109], angs[109];
    static const int id[13] = {6, 5, 4, 3, 2, 1, 0, 1, 2, 3, 4, 5, 6};
    const evo_t* ev = &c->ev[kp->class_id];
    float ratio = (float)(1u << ev->octave);
    float s = roundf(0.5f * kp->size / ratio);
    float xf = kp->x / ratio;
    float yf = kp->y / ratio;
    int idx = 0;
    for (int j = -6; j <= 6; ++j)
        for (int i = -6; i <= 6; ++i) {
            if (i * i + j * j < 36) {
                size_t iy = sat_usize_f32(roundf(yf + (float)j * s));
                size_t ix = sat_usize_f32(roundf(xf + (float)i * s));
                if (ix >= (size_t)ev->Lx.w || iy >= (size_t)ev->Lx.h) return -1;
                float g = GAUSS25[id[j + 6]][id[i + 6]];
                res_x[idx] = g * ev->Lx.d[iy * (size_t)ev->Lx.w + ix];
                res_y[idx] = g * ev->Ly.d[iy * (size_t)ev->Ly.w + ix];
                angs[idx] = fast_atan2_equiv(res_y[idx], res_x[idx]);
                idx++;
            }
        }
    float ang1 = 0.0f, max = 0.0f;
    while (ang1 < 2.0f * PI_F) {
        float sum_x = 0.0f, sum_y = 0.0f;
        float ang2 = (ang1 + PI_F / 3.0f > 2.0f * PI_F) ? ang1 - 5.0f * PI_F / 3.0f : ang1 + PI_F / 3.0f;
        for (int k = 0; k < 109; ++k) {
            float ang = angs[k];
            if ((ang1 < ang2 && ang1 < ang && ang < ang2) ||
                (ang2 < ang1 && ((ang > 0.0f && ang < ang2) || (ang > ang1 && ang < 2.0f * PI_F)))) {
                sum_x += res_x[k];
                sum_y += res_y[k];
            }
        }
        float val = sum_x * sum_x + sum_y * sum_y;
        if (val > max) {
            max = val;
            kp->angle = fast_atan2_equiv(sum_y, sum_x);
        }
        ang1 += 0.15f;
    }
    return 0;
}

/* do_subpixel_refinement — :297-362. */
/* the closure `process_keypoint` of scale_space_extrema.rs:297-344: 1 and *res when the keypoint survives */
static int refine_one(const orc_ctx* c, const akz_keypoint* kp, akz_keypoint* res)
{
    {
        const evo_t* ev = &c->ev[kp->class_id];
        float ratio = ldexpf(1.0f, (int)kp->octave);
        size_t x = sat_usize_f32(roundf(kp->x / ratio));
        size_t y = sat_usize_f32(roundf(kp->y / ratio));
        size_t w = (size_t)ev->Ldet.w;
        const float* D = ev->Ldet.d;
        float x_i = D[y * w + x];
        float x_p = D[y * w + x + 1];
        float x_m = D[y * w + x - 1];
        float y_p = D[(y + 1) * w + x];
        float y_m = D[(y - 1) * w + x];
        float x_p_y_p = D[(y + 1) * w + x + 1];
        float x_p_y_m = D[(y - 1) * w + x + 1];
        float x_m_y_p = D[(y + 1) * w + x - 1];
        float x_m_y_m = D[(y - 1) * w + x - 1];
        float d_x = 0.5f * (x_p - x_m);
        float d_y = 0.5f * (y_p - y_m);
        float d_xx = x_p + x_m - 2.0f * x_i;
        float d_yy = y_p + y_m - 2.0f * x_i;
        float d_xy = 0.25f * (x_p_y_p + x_m_y_m) - 0.25f * (x_p_y_m + x_m_y_p);
        float inv_det_a = 1.0f / (d_xx * d_yy - d_xy * d_xy);
        float inv_a0 = inv_det_a * d_yy;
        float inv_a1 = inv_det_a * -d_xy;
        float inv_a2 = inv_det_a * -d_xy;
        float inv_a3 = inv_det_a * d_xx;
        float dst0 = -d_x * inv_a0 + -d_y * inv_a1;
        float dst1 = -d_x * inv_a2 + -d_y * inv_a3;
        if (fabsf(dst0) <= 1.0f && fabsf(dst1) <= 1.0f) {
            akz_keypoint k2 = *kp;
            k2.x = (float)x + dst0;
            k2.y = (float)y + dst1;
            float power = ldexpf(1.0f, (int)ev->octave);
            k2.x = k2.x * power + 0.5f * (power - 1.0f);
            k2.y = k2.y * power + 0.5f * (power - 1.0f);
            k2.size *= 2.0f;
            if (compute_main_orientation(c, &k2) != 0) {
                fprintf(stderr, "orc: orientation sample out of bounds (reference would panic)\n");
                return 0;
            }
            *res = k2;
            return 1;
        }
    }
    return 0;
}
static void do_subpixel_refinement(const orc_ctx* c, const kpvec* in, kpvec* out)
{
    if (g_opt[ORC_OPT_INTRA] && in->n) {            /* :345-352: par_iter().filter_map().collect() keeps the order */
        akz_keypoint* tmp = (akz_keypoint*)malloc(sizeof(akz_keypoint) * in->n);
        unsigned char* keep = (unsigned char*)malloc(in->n);
#pragma omp parallel for schedule(static)
        for (long n = 0; n < (long)in->n; ++n) keep[n] = (unsigned char)refine_one(c, &in->v[n], &tmp[n]);
        for (uint32_t n = 0; n < in->n; ++n)
            if (keep[n]) kpvec_push(out, tmp[n]);
        free(tmp);
        free(keep);
        return;
    }
    for (uint32_t n = 0; n < in->n; ++n) {
        akz_keypoint k2;
        if (refine_one(c, &in->v[n], &k2)) kpvec_push(out, k2);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* descriptors.rs                                                                               */

/* mldb_fill_values — :102-177. returns 0 ok, -1 SampleOutOfBounds. */
static int mldb_fill_values(const orc_ctx* c, float* values, size_t sample_step, uint32_t level, float xf,
                            float yf, float co, float si, float scale)
{
    int pattern = (int)c->cfg.descriptor_pattern_size;
    size_t nch = (size_t)c->cfg.descriptor_channels;
    const evo_t* ev = &c->ev[level];
    long W = ev->Lt.w, H = ev->Lt.h;
    size_t valuepos = 0;
    for (int i = -pattern; i < pattern; i += (int)sample_step)
        for (int j = -pattern; j < pattern; j += (int)sample_step) {
            float di = 0.0f, dx = 0.0f, dy = 0.0f;
            size_t nsamples = 0;
            for (int k = i; k < i + (int)sample_step; ++k)
                for (int l = j; l < j + (int)sample_step; ++l) {
                    float lf = (float)l, kf = (float)k;
                    float sample_y = yf + (lf * co * scale + kf * si * scale);
                    float sample_x = xf + (-lf * si * scale + kf * co * scale);
                    long y1 = sat_isize_f32(roundf(sample_y));
                    long x1 = sat_isize_f32(roundf(sample_x));
                    if (x1 < 0 || x1 >= W || y1 < 0 || y1 >= H) return -1;
                    size_t p = (size_t)y1 * (size_t)W + (size_t)x1;
                    float ri = ev->Lt.d[p];
                    di += ri;
                    if (nch > 1) {
                        float rx = ev->Lx.d[p], ry = ev->Ly.d[p];
                        if (nch == 2) {
                            dx += sqrtf(rx * rx + ry * ry);
                        } else {
                            float rry = rx * co + ry * si;
                            float rrx = -rx * si + ry * co;
                            dx += rrx;
                            dy += rry;
                        }
                    }
                    nsamples += 1;
                }
            di /= (float)nsamples;
            dx /= (float)nsamples;
            dy /= (float)nsamples;
            values[valuepos] = di;
            if (nch > 1) values[valuepos + 1] = dx;
            if (nch > 2) values[valuepos + 2] = dy;
            valuepos += nch;
        }
    return 0;
}
/* mldb_binary_comparisons — :181-202. */
static void mldb_binary_comparisons(const float* values, uint8_t* desc, size_t count, size_t* dpos, size_t nch)
{
    for (size_t pos = 0; pos < nch; ++pos)
        for (size_t i = 0; i < count; ++i) {
            float ival = values[nch * i + pos];
            for (size_t j = i + 1; j < count; ++j) {
                uint8_t res = ival > values[nch * j + pos] ? 1 : 0;
                desc[*dpos >> 3] |= (uint8_t)(res << (*dpos & 7));
                *dpos += 1;
            }
        }
}
/* get_mldb_descriptor — :55-98. */
static int get_mldb_descriptor(const orc_ctx* c, const akz_keypoint* kp, akz_descriptor* out)
{
    memset(out->bytes, 0, 64);
    float values[16 * 3];
    for (int i = 0; i < 48; ++i) values[i] = 0.0f;
    const float size_mult[3] = {1.0f, 2.0f / 3.0f, 1.0f / 2.0f};
    float ratio = (float)(1u << kp->octave);
    float scale = roundf(0.5f * kp->size / ratio);
    float xf = kp->x / ratio;
    float yf = kp->y / ratio;
    float co = cos_sel(kp->angle);
    float si = sin_sel(kp->angle);
    float pattern_size = (float)c->cfg.descriptor_pattern_size;
    size_t dpos = 0;
    for (size_t lvl = 0; lvl < 3; ++lvl) {
        size_t val_count = (lvl + 2) * (lvl + 2);
        size_t sample_size = sat_usize_f32(ceilf(pattern_size * size_mult[lvl]));
        if (mldb_fill_values(c, values, sample_size, kp->class_id, xf, yf, co, si, scale) != 0) return -1;
        mldb_binary_comparisons(values, out->bytes, val_count, &dpos, (size_t)c->cfg.descriptor_channels);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* lib.rs:309-339 extract_from_gray_float_image                                                 */

/* sort_unstable_by_key(Reverse(FloatOrd(response))): tie order is unspecified in the reference;
 * we define (response descending, pre-sort index ascending) — SURVEY.md §7 hard part 4. */
typedef struct {
    akz_keypoint k;
    uint32_t idx;
} sort_item;
static int cmp_resp_desc(const void* a, const void* b)
{
    const sort_item* p = (const sort_item*)a;
    const sort_item* q = (const sort_item*)b;
    if (p->k.response > q->k.response) return -1;
    if (p->k.response < q->k.response) return 1;
    return p->idx < q->idx ? -1 : (p->idx > q->idx ? 1 : 0);
}

int orc_extract_f32(orc_ctx* c, const float* image)
{
    free_results(c);
    img_t im = {c->w, c->h, (float*)image};
    create_nonlinear_scale_space(c, &im);
    detector_response(c);
    kpvec ext = {0, 0, 0}, ref = {0, 0, 0};
    find_scale_space_extrema(c, &ext);
    do_subpixel_refinement(c, &ext, &ref);
    c->kp_extrema = ext.v;
    c->n_extrema = ext.n;
    c->kp_refined = ref.v;
    c->n_refined = ref.n;
    /* sort + truncate */
    sort_item* items = (sort_item*)malloc(sizeof(sort_item) * (ref.n ? ref.n : 1));
    for (uint32_t i = 0; i < ref.n; ++i) {
        items[i].k = ref.v[i];
        items[i].idx = i;
    }
    qsort(items, ref.n, sizeof(sort_item), cmp_resp_desc);
    uint32_t ns = ref.n;
    if ((uint64_t)ns > c->cfg.maximum_features) ns = (uint32_t)c->cfg.maximum_features;
    c->kp_sorted = (akz_keypoint*)malloc(sizeof(akz_keypoint) * (ns ? ns : 1));
    for (uint32_t i = 0; i < ns; ++i) c->kp_sorted[i] = items[i].k;
    c->n_sorted = ns;
    free(items);
    /* extract_descriptors — descriptors.rs:16-45: keypoints whose samples leave the image vanish */
    c->kp_final = (akz_keypoint*)malloc(sizeof(akz_keypoint) * (ns ? ns : 1));
    c->desc_final = (akz_descriptor*)malloc(sizeof(akz_descriptor) * (ns ? ns : 1));
    uint32_t nf = 0;
    if (g_opt[ORC_OPT_INTRA] && ns) {               /* descriptors.rs:31-42: par_iter().filter_map().unzip(), in order */
        akz_descriptor* tmp = (akz_descriptor*)malloc(sizeof(akz_descriptor) * ns);
        unsigned char* keep = (unsigned char*)malloc(ns);
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)ns; ++i) keep[i] = (unsigned char)(get_mldb_descriptor(c, &c->kp_sorted[i], &tmp[i]) == 0);
        for (uint32_t i = 0; i < ns; ++i)
            if (keep[i]) {
                c->kp_final[nf] = c->kp_sorted[i];
                c->desc_final[nf] = tmp[i];
                nf++;
            }
        free(tmp);
        free(keep);
    } else {
        for (uint32_t i = 0; i < ns; ++i) {
            akz_descriptor d;
            if (get_mldb_descriptor(c, &c->kp_sorted[i], &d) == 0) {
                c->kp_final[nf] = c->kp_sorted[i];
                c->desc_final[nf] = d;
                nf++;
            }
        }
    }
    c->n_final = nf;
    return (int)nf;
}

int orc_extract_u8(orc_ctx* c, const uint8_t* image, int stride)
{
    float* f = (float*)malloc(sizeof(float) * (size_t)c->w * c->h);
    orc_u8_to_f32(image, c->w, c->h, stride, f);
    int r = orc_extract_f32(c, f);
    free(f);
    return r;
}

/* GrayFloatImage::from_dynamic, ImageLuma16 arm (akaze/src/image.rs:57-66): f32::from(v) / 65535f32 */
void orc_u16_to_f32(const uint16_t* in, int w, int h, int stride, float* out)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) out[(size_t)y * w + x] = (float)in[(size_t)y * stride + x] / 65535.0f;
}

int orc_extract_u16(orc_ctx* c, const uint16_t* image, int stride)
{
    float* f = (float*)malloc(sizeof(float) * (size_t)c->w * c->h);
    orc_u16_to_f32(image, c->w, c->h, stride, f);
    int r = orc_extract_f32(c, f);
    free(f);
    return r;
}

/* Scale space + detector response only (BASELINE configs[1]); used by the cpu_baseline leg. */
int orc_scale_space_u8(orc_ctx* c, const uint8_t* image, int stride)
{
    float* f = (float*)malloc(sizeof(float) * (size_t)c->w * c->h);
    orc_u8_to_f32(image, c->w, c->h, stride, f);
    img_t im = {c->w, c->h, f};
    create_nonlinear_scale_space(c, &im);
    detector_response(c);
    free(f);
    return 0;
}

/* ---- accessors ---- */
const float* orc_level_buffer(const orc_ctx* c, int lvl, int which, int* w, int* h)
{
    if (lvl < 0 || lvl >= c->nlev) return NULL;
    const evo_t* e = &c->ev[lvl];
    const img_t* im = NULL;
    switch (which) {
    case AKZ_BUF_LT: im = &e->Lt; break;
    case AKZ_BUF_LSMOOTH: im = &e->Lsmooth; break;
    case AKZ_BUF_LX: im = &e->Lx; break;
    case AKZ_BUF_LY: im = &e->Ly; break;
    case AKZ_BUF_LDET: im = &e->Ldet; break;
    case AKZ_BUF_LFLOW: im = &e->Lflow; break;
    case 6: im = &e->Lxx; break;
    case 7: im = &e->Lyy; break;
    case 8: im = &e->Lxy; break;
    default: return NULL;
    }
    if (w) *w = im->w;
    if (h) *h = im->h;
    return im->d;
}
double orc_contrast(const orc_ctx* c) { return c->contrast0; }
uint32_t orc_num_candidates(const orc_ctx* c) { return c->n_candidates; }
uint32_t orc_keypoints(const orc_ctx* c, int stage, const akz_keypoint** out)
{
    switch (stage) {
    case 0: *out = c->kp_extrema; return c->n_extrema;
    case 1: *out = c->kp_refined; return c->n_refined;
    case 2: *out = c->kp_sorted; return c->n_sorted;
    default: *out = c->kp_final; return c->n_final;
    }
}
const akz_descriptor* orc_descriptors(const orc_ctx* c) { return c->desc_final; }

/* portable-math probes for tests */
float orc_pm_atan2f(float y, float x) { return akz_pm_atan2f(y, x); }
float orc_pm_sinf(float a) { return akz_pm_sinf(a); }
float orc_pm_cosf(float a) { return akz_pm_cosf(a); }
void orc_pm_atan2f_v(const float* y, const float* x, size_t n, float* out)
{
    for (size_t i = 0; i < n; ++i) out[i] = akz_pm_atan2f(y[i], x[i]);
}
void orc_pm_sincosf_v(const float* a, size_t n, float* s, float* c)
{
    for (size_t i = 0; i < n; ++i) {
        s[i] = akz_pm_sinf(a[i]);
        c[i] = akz_pm_cosf(a[i]);
    }
}
