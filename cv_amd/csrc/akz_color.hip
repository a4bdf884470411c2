// akz_color.hip — bicubic colour sampling at keypoints.
//
// Follows cv-sfm/src/bicubic.rs:13-68 (interpolate_bicubic / blend_cubic, code the reference copied from
// imageproc) as used by VSlam::kps_descriptors (cv-sfm/src/lib.rs:2207-2216): for every keypoint the RGB8
// image is sampled at kp.point; a 4x4 neighbourhood that leaves the image yields the default colour [0,0,0].
// Rows are blended first and ROUNDED TO u8 (the reference's intermediate `col` holds Rgb<u8>), then the four
// row results are blended vertically.  All arithmetic is f32 in the reference's expression order; the final
// conversion is imageproc's Clamp<f32> for u8 [3P, unverified]: x >= 255 -> 255, x <= 0 -> 0, else truncation.
#include "akz_ctx.h"

namespace {

__device__ __forceinline__ uint8_t clamp_u8(float x)
{
    if (x < 255.0f) return x > 0.0f ? (uint8_t)x : (uint8_t)0;   // NaN takes the else branch below, as in Rust
    return (uint8_t)255;
}

// bicubic.rs:27: p1 + 0.5 * x * (p2 - p0 + x * (2.0 * p0 - 5.0 * p1 + 4.0 * p2 - p3 + x * (3.0 * (p1 - p2) + p3 - p0)))
__device__ __forceinline__ uint8_t blend_cubic(float p0, float p1, float p2, float p3, float x)
{
    float in3 = (3.0f * (p1 - p2) + p3) - p0;
    float in2 = (((2.0f * p0 - 5.0f * p1) + 4.0f * p2) - p3) + x * in3;
    float in1 = (p2 - p0) + x * in2;
    float pval = p1 + (0.5f * x) * in1;
    return clamp_u8(pval);
}

__global__ __launch_bounds__(256) void k_bicubic_rgb8(const uint8_t* __restrict__ rgb, int w, int h,
                                                      const akz_keypoint* __restrict__ kps, uint32_t n,
                                                      uint8_t* __restrict__ colors)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = kps[i].x, y = kps[i].y;
    // bicubic.rs:39-46
    const float left = floorf(x) - 1.0f, right = left + 4.0f;
    const float top = floorf(y) - 1.0f, bottom = top + 4.0f;
    const float xw = x - (left + 1.0f), yw = y - (top + 1.0f);
    uint8_t out[3] = {0, 0, 0};
    if (!(left < 0.0f || right >= (float)w || top < 0.0f || bottom >= (float)h)) {
        const uint32_t l = (uint32_t)left, t = (uint32_t)top;
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

}  // namespace

extern "C" int32_t akz_sample_colors_rgb8(akz_ctx* c, const uint8_t* rgb, int32_t w, int32_t h, int32_t stride,
                                          const akz_keypoint* kps, uint32_t n, uint8_t* colors)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !rgb || w <= 0 || h <= 0 || stride < 3 * w || (n && (!kps || !colors))) return AKZ_E_INVALID;
        if (n == 0) return AKZ_OK;
        AKZ_HIP(hipSetDevice(c->device));
        const size_t img_bytes = (size_t)w * h * 3, kp_bytes = sizeof(akz_keypoint) * (size_t)n, col_bytes = 3 * (size_t)n;
        const size_t need = akz_align_up(img_bytes, 256) + akz_align_up(kp_bytes, 256) + akz_align_up(col_bytes, 256);
        if (need > c->color_bytes) {
            AKZ_HIP(hipStreamSynchronize(c->stream_kp));
            if (c->d_color) AKZ_HIP(hipFree(c->d_color));
            c->d_color = nullptr;
            c->color_bytes = 0;
            AKZ_HIP(hipMalloc(&c->d_color, need));
            c->color_bytes = need;
        }
        uint8_t* d_img = (uint8_t*)c->d_color;
        akz_keypoint* d_kp = (akz_keypoint*)(d_img + akz_align_up(img_bytes, 256));
        uint8_t* d_col = (uint8_t*)d_kp + akz_align_up(kp_bytes, 256);
        hipStream_t s = c->stream_kp;
        AKZ_HIP(hipMemcpy2DAsync(d_img, (size_t)w * 3, rgb, (size_t)stride, (size_t)w * 3, (size_t)h, hipMemcpyHostToDevice, s));
        AKZ_HIP(hipMemcpyAsync(d_kp, kps, kp_bytes, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_bicubic_rgb8, dim3((n + 255) / 256), dim3(256), 0, s, d_img, w, h, d_kp, n, d_col);
        AKZ_LAUNCH_CHECK();
        AKZ_HIP(hipMemcpyAsync(colors, d_col, col_bytes, hipMemcpyDeviceToHost, s));
        AKZ_HIP(hipStreamSynchronize(s));
        return AKZ_OK;
    });
}

// ---- colour inputs: the arms of GrayFloatImage::from_dynamic that go through DynamicImage::grayscale() -------------
// akaze/src/image.rs:45-46: `match input_image.grayscale()`.  For 8/16-bit RGB(A) the `image` crate (0.24, un-vendored:
// [3P-unverified]) computes an integer Rec. 709 luma, (2126 R + 7152 G + 722 B) / 10000 in the next larger integer
// type, truncating; the result is the Luma8 / Luma16 arm (image.rs:47-66).  Rgb32F / Rgba32F stay float through
// grayscale() and take `to_luma()` per pixel (image.rs:87-106): the same weights in f64, narrowed to f32.
// The conversion runs on the device into the context's input plane; everything after is the gray path.
namespace {
template <typename T>
__global__ __launch_bounds__(256) void k_to_luma(const T* __restrict__ px, int channels, int w, int h, size_t stride, T* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)w * h) return;
    const int y = (int)(i / (size_t)w), x = (int)(i - (size_t)y * w);
    const T* p = px + (size_t)y * stride + (size_t)x * channels;
    if (sizeof(T) == 4) {
        const double l = ((2126.0 * (double)p[0] + 7152.0 * (double)p[1]) + 722.0 * (double)p[2]) / 10000.0;
        out[i] = (T)l;
    } else {
        const unsigned long long l = (2126ull * (unsigned long long)p[0] + 7152ull * (unsigned long long)p[1] + 722ull * (unsigned long long)p[2]) / 10000ull;
        out[i] = (T)l;
    }
}
}  // namespace

// device side of akz_extract_color: n = 1 interleaved image (host) -> luma plane in the context's input buffer
int32_t akz_color_to_input(akz_ctx* c, const void* pixels, int32_t fmt, int32_t channels, int32_t w, int32_t h, int32_t stride)
{
    const size_t esz = fmt == AKZ_FMT_U8 ? 1 : (fmt == AKZ_FMT_U16 ? 2 : 4);
    const size_t row = (size_t)w * channels * esz, bytes = row * h;
    if (bytes > c->color_bytes) {
        AKZ_HIP(hipStreamSynchronize(c->stream));
        if (c->d_color) AKZ_HIP(hipFree(c->d_color));
        c->d_color = nullptr;
        c->color_bytes = 0;
        AKZ_HIP(hipMalloc(&c->d_color, bytes));
        c->color_bytes = bytes;
    }
    AKZ_HIP(hipMemcpy2DAsync(c->d_color, row, pixels, (size_t)stride * esz, row, (size_t)h, hipMemcpyHostToDevice, c->stream));
    const dim3 grid((unsigned)(((size_t)w * h + 255) / 256));
    const size_t st = (size_t)w * channels;
    if (fmt == AKZ_FMT_U8)
        hipLaunchKernelGGL(k_to_luma<uint8_t>, grid, dim3(256), 0, c->stream, (const uint8_t*)c->d_color, channels, w, h, st, (uint8_t*)c->S().d_in);
    else if (fmt == AKZ_FMT_U16)
        hipLaunchKernelGGL(k_to_luma<uint16_t>, grid, dim3(256), 0, c->stream, (const uint16_t*)c->d_color, channels, w, h, st, (uint16_t*)c->S().d_in);
    else
        hipLaunchKernelGGL(k_to_luma<float>, grid, dim3(256), 0, c->stream, (const float*)c->d_color, channels, w, h, st, (float*)c->S().d_in);
    AKZ_LAUNCH_CHECK();
    return AKZ_OK;
}
