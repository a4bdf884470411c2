/* color_oracle.c — CPU restatement of the reference's bicubic colour sampling at keypoints.
 *
 * TEST INFRASTRUCTURE ONLY (see akaze_oracle.c header): the checker for akz_sample_colors_rgb8.
 *
 * Follows cv-sfm/src/bicubic.rs:13-68 and its call site cv-sfm/src/lib.rs:2207-2216.  The u8 <- f32
 * conversion is imageproc's `Clamp<f32> for u8` (un-vendored; restated from its published macro:
 * `if x < 255.0 { if x > 0.0 { x as u8 } else { 0 } } else { 255 }`).  PARITY UNPINNED: the reference holds
 * no test or golden vector for this function; the oracle is checked against an independent numpy/float32
 * evaluation of the same expressions (tests/test_oracle_math.py). */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#include "../include/akz.h"

static uint8_t clamp_u8(float x)
{
    if (x < 255.0f) return x > 0.0f ? (uint8_t)x : (uint8_t)0;
    return 255;
}

static uint8_t blend_cubic(float p0, float p1, float p2, float p3, float x)
{
    /* bicubic.rs:27, f32, left to right */
    float in3 = (3.0f * (p1 - p2) + p3) - p0;
    float in2 = (((2.0f * p0 - 5.0f * p1) + 4.0f * p2) - p3) + x * in3;
    float in1 = (p2 - p0) + x * in2;
    float pval = p1 + (0.5f * x) * in1;
    return clamp_u8(pval);
}

int orc_sample_colors_rgb8(const uint8_t* rgb, int w, int h, const akz_keypoint* kps, uint32_t n, uint8_t* colors)
{
    for (uint32_t i = 0; i < n; ++i) {
        float x = kps[i].x, y = kps[i].y;
        float left = floorf(x) - 1.0f, right = left + 4.0f;
        float top = floorf(y) - 1.0f, bottom = top + 4.0f;
        float xw = x - (left + 1.0f), yw = y - (top + 1.0f);
        uint8_t out[3] = {0, 0, 0}; /* default Rgb([0, 0, 0]) */
        if (!(left < 0.0f || right >= (float)w || top < 0.0f || bottom >= (float)h)) {
            uint32_t l = (uint32_t)left, t = (uint32_t)top;
            uint8_t col[4][3];
            for (int r = 0; r < 4; ++r) {
                const uint8_t* p = rgb + ((size_t)(t + r) * w + l) * 3;
                for (int ch = 0; ch < 3; ++ch)
                    col[r][ch] = blend_cubic((float)p[ch], (float)p[3 + ch], (float)p[6 + ch], (float)p[9 + ch], xw);
            }
            for (int ch = 0; ch < 3; ++ch)
                out[ch] = blend_cubic((float)col[0][ch], (float)col[1][ch], (float)col[2][ch], (float)col[3][ch], yw);
        }
        colors[(size_t)i * 3 + 0] = out[0];
        colors[(size_t)i * 3 + 1] = out[1];
        colors[(size_t)i * 3 + 2] = out[2];
    }
    return 0;
}

/* DynamicImage::grayscale() of a colour image, then the sample type's arm of GrayFloatImage::from_dynamic
 * (akaze/src/image.rs:45-109).  The `image` crate (0.24) is not vendored: its published conversion is
 *   rgb_to_luma: l = 2126 R + 7152 G + 722 B in the next larger type (u32 / u64 / f64), l / 10000, narrowed
 * — integer division truncates; f32 pixels go through f64.  PARITY UNPINNED [3P].  fmt: AKZ_FMT_*; channels 3 or 4
 * (alpha ignored); stride in elements; out: w*h samples of the same type. */
void orc_luma(const void* in, int fmt, int channels, int w, int h, int stride, void* out)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            size_t i = (size_t)y * stride + (size_t)x * channels, o = (size_t)y * w + x;
            if (fmt == AKZ_FMT_U8) {
                const uint8_t* p = (const uint8_t*)in + i;
                ((uint8_t*)out)[o] = (uint8_t)((2126u * p[0] + 7152u * p[1] + 722u * p[2]) / 10000u);
            } else if (fmt == AKZ_FMT_U16) {
                const uint16_t* p = (const uint16_t*)in + i;
                ((uint16_t*)out)[o] = (uint16_t)((2126ull * p[0] + 7152ull * p[1] + 722ull * p[2]) / 10000ull);
            } else {
                const float* p = (const float*)in + i;
                ((float*)out)[o] = (float)(((2126.0 * (double)p[0] + 7152.0 * (double)p[1]) + 722.0 * (double)p[2]) / 10000.0);
            }
        }
}
