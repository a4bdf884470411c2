// akz_arith.hip — one context, one of eight copies of the scale-space kernels.
//
// Three pieces of the reference's arithmetic live in crates that are not vendored in rust-cv/cv: the order
// of wide::f32x4::reduce_add, whether wide::f32x4::mul_add fuses, and the order of ndarray's sum() over a 2 x 2 window
// (akaze/src/image.rs:242-247, :320-325, :160-195).  The reference's own known answers (399 / 343 / 11) do not tell the
// eight combinations apart, so the library ships all of them: cv_amd/csrc/akz_scale_space.hip is compiled once per
// combination (-DAKZ_ARITH=k gives its entry points the suffix _arithk) and akz_options.arith picks the copy at
// akz_create_ex.  The default, 0, is the combination the crate sources imply for a default x86-64 build; the day somebody
// diffs tools/akaze_dump.py against a cargo build and finds another one, it is a flag, not a rewrite.
#include "akz_ctx.h"

#define AKZ_ARITH_DECL(k)                                                                                              \
    int32_t akz_run_scale_space_arith##k(akz_ctx* c, const void* d_imgs, int fmt, int n);                              \
    int32_t akz_dev_filter1d_arith##k(hipStream_t s, const float* in, float* out, int w, int h, const float* d_kernel, \
                                      int ksize, int vertical);                                                        \
    int32_t akz_dev_half_size_arith##k(hipStream_t s, const float* in, float* out, int w, int h, int n, size_t in_fs,  \
                                       size_t out_fs);
AKZ_ARITH_DECL(0) AKZ_ARITH_DECL(1) AKZ_ARITH_DECL(2) AKZ_ARITH_DECL(3)
AKZ_ARITH_DECL(4) AKZ_ARITH_DECL(5) AKZ_ARITH_DECL(6) AKZ_ARITH_DECL(7)
#undef AKZ_ARITH_DECL

#define AKZ_ARITH_ROUTE(arith, fn, ...)                    \
    switch ((arith) & 7) {                                 \
    case 0: return fn##_arith0(__VA_ARGS__);               \
    case 1: return fn##_arith1(__VA_ARGS__);               \
    case 2: return fn##_arith2(__VA_ARGS__);               \
    case 3: return fn##_arith3(__VA_ARGS__);               \
    case 4: return fn##_arith4(__VA_ARGS__);               \
    case 5: return fn##_arith5(__VA_ARGS__);               \
    case 6: return fn##_arith6(__VA_ARGS__);               \
    default: return fn##_arith7(__VA_ARGS__);              \
    }

int32_t akz_run_scale_space(akz_ctx* c, const void* d_imgs, int fmt, int n)
{
    AKZ_ARITH_ROUTE(c->arith, akz_run_scale_space, c, d_imgs, fmt, n)
}
int32_t akz_dev_filter1d(int arith, hipStream_t s, const float* in, float* out, int w, int h, const float* d_kernel, int ksize,
                         int vertical)
{
    AKZ_ARITH_ROUTE(arith, akz_dev_filter1d, s, in, out, w, h, d_kernel, ksize, vertical)
}
int32_t akz_dev_half_size(int arith, hipStream_t s, const float* in, float* out, int w, int h, int n, size_t in_fs, size_t out_fs)
{
    AKZ_ARITH_ROUTE(arith, akz_dev_half_size, s, in, out, w, h, n, in_fs, out_fs)
}
