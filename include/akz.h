/* akz.h — C ABI of the MI355X-native AKAZE + brute-force Hamming front-end.
 *
 * This is the drop-in boundary for the reference's data-parallel hot path.
 * Every entry point names the reference interface it replaces (paths relative to the rust-cv/cv
 * checkout).  Plain pointers and sizes only: no C++/torch/HIP types appear in a signature
 * (device pointers and streams travel as void*).
 *
 * Conventions
 *   - every function returns an int32_t status: AKZ_OK (0) or a negative akz_status;
 *     nothing throws, nothing aborts, akz_strerror() names the code;
 *   - the library never retains a caller pointer after the call returns;
 *   - a context is bound to one HIP device and one internal stream; use one context per host
 *     thread.  Contexts on different devices are independent;
 *   - "capacity" arguments are in elements.  When an output does not fit, the call returns
 *     AKZ_E_CAPACITY and *n_out holds the required element count;
 *   - there is NO CPU fallback: if no HIP device is usable akz_create() fails with AKZ_E_NO_DEVICE.
 */
#ifndef AKZ_H
#define AKZ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum akz_status {
    AKZ_OK = 0,
    AKZ_E_INVALID = -1,    /* null pointer, bad dimension, unsupported config value */
    AKZ_E_NO_DEVICE = -2,  /* no usable HIP device / device index out of range */
    AKZ_E_OOM = -3,        /* hipMalloc failed */
    AKZ_E_CAPACITY = -4,   /* caller buffer too small; *n_out = required */
    AKZ_E_HIP = -5,        /* a HIP runtime call failed (akz_last_hip_error()) */
    AKZ_E_TOO_LARGE = -6,  /* image or batch exceeds what the context was created for */
    AKZ_E_INTERNAL = -7,   /* device-side overflow of an internal work list */
    AKZ_E_COMM = -8        /* an RCCL call failed, or librccl.so.1 could not be loaded (akz_comm_*) */
} akz_status;

/* akaze::Akaze — akaze/src/lib.rs:109-142 (fields), :169-185 (Default), :147-166 (new/sparse/dense).
 * Field-for-field; usize -> uint64_t. */
typedef struct akz_config {
    uint64_t maximum_features;        /* usize::MAX by default */
    uint32_t num_sublevels;           /* 4 */
    uint32_t max_octave_evolution;    /* 4 */
    double base_scale_offset;         /* 1.6 */
    double initial_contrast;          /* 0.001 (unused by the reference as well) */
    double contrast_percentile;       /* 0.7 */
    uint64_t contrast_factor_num_bins; /* 300 */
    double derivative_factor;         /* 1.5 */
    double detector_threshold;        /* 0.001; sparse 0.01; dense 0.0001 */
    uint64_t descriptor_channels;     /* 3 */
    uint64_t descriptor_pattern_size; /* 10 */
} akz_config;

/* `stream_to_wait` arguments (a hipStream_t passed as void*): the stream that produced the call's device inputs; the
 * library's stream waits for the work enqueued on it so far.  NULL = nothing to wait for.  The legacy default stream's
 * own handle is NULL as well, so a caller that enqueues on it (torch without a stream context, plain `<<<>>>` launches)
 * passes AKZ_STREAM_LEGACY — the value of hipStreamLegacy — instead: the library's streams are non-blocking and do NOT
 * synchronise with the default stream on their own. */
#define AKZ_STREAM_LEGACY ((void*)1)

/* akaze::KeyPoint — akaze/src/lib.rs:69-93. point=(x,y). 28 bytes, no padding. */
typedef struct akz_keypoint {
    float x, y;
    float response;
    float size;
    float angle;
    uint32_t octave;
    uint32_t class_id;
} akz_keypoint;

/* bitarray::BitArray<64> as filled by akaze/src/descriptors.rs:181-202: bit i lives in
 * bytes[i >> 3] at position (i & 7); 486 bits used, bits 486..511 are zero. */
typedef struct akz_descriptor {
    uint8_t bytes[64];
} akz_descriptor;

/* space::Neighbor<u32> — what LinearKnn::knn returns (call sites akaze/tests/estimate_pose.rs:82-88,
 * tutorial-code/chapter5-geometric-verification/src/main.rs:155-161). */
typedef struct akz_neighbor {
    uint32_t index;
    uint32_t distance;
} akz_neighbor;

typedef struct akz_ctx akz_ctx;

/* Akaze::default() (akaze/src/lib.rs:169-185). */
void akz_config_default(akz_config* cfg);

/* Context = the pre-allocated per-device pyramid for up to max_batch frames of up to
 * max_w x max_h pixels (what Akaze::allocate_evolutions, akaze/src/evolution.rs:80-126, allocates
 * per call in the reference).  max_keypoints bounds the per-frame candidate/keypoint lists
 * (0 = default 16384).  max_w, max_h in [3, 65535] and max_w * max_h <= 2^28 pixels (the kernels address a
 * frame's planes with 32-bit byte offsets): a larger frame is refused with AKZ_E_TOO_LARGE. */
int32_t akz_create(const akz_config* cfg, int32_t device, int32_t max_w, int32_t max_h,
                   int32_t max_batch, uint32_t max_keypoints, akz_ctx** out);
int32_t akz_destroy(akz_ctx* ctx);

/* Behaviour switches of ONE context.  The library reads no environment variable: two contexts in one
 * process can differ, and nothing outside the caller's code changes what a context does.  The defaults
 * (all zero) are the measured-best paths; the other values select the fall-back / reference kernels that
 * the parity tests and A/B runs exercise.  Results are bit-identical under every combination (of everything but `arith`, below). */
enum {
    AKZ_OPT_KEEP_ALL = 1u << 0,           /* keep per-level Lsmooth / Lflow and write the Ldet planes (parity taps) */
    AKZ_OPT_NO_FRAME_PAIRS = 1u << 1,     /* one-frame kernels even for widths divisible by 4 */
    AKZ_OPT_SERIAL_SUPPRESSION = 1u << 2, /* one-wave serial walk of scale_space_extrema.rs:61-118 for every frame */
    AKZ_OPT_NO_PIPELINE = 1u << 3,        /* one buffer set: consecutive calls do not overlap */
    AKZ_OPT_STREAM_PRIORITY = 1u << 4,    /* (accepted, no effect: scale-space stream at high and keypoint stream at low
                                           * priority is the default since it measured +1 % on the bench workload) */
    AKZ_OPT_CONTRAST_EXACT = 1u << 5,     /* contrast factor always through the exact histogram pass */
    AKZ_OPT_CONTRAST_FORCE_ODD = 1u << 6, /* test knob: odd frames through the exact pass (mixed pairs) */
    AKZ_OPT_TILE_KERNELS = 1u << 7,       /* the LDS-tile determinant kernels of round 1 instead of the row-streaming one */
    AKZ_OPT_SERIAL_DET = 1u << 8,         /* determinant / candidate kernels on the scale-space stream itself instead of
                                           * the side stream that takes them off the Lt -> Lt dependency chain */
    AKZ_OPT_SPLIT_FRONT_FED = 1u << 9,    /* level front end and first FED launch as two kernels (Lflow through HBM)
                                           * instead of the fused k_front_fed */
    AKZ_OPT_EQUAL_PRIORITY = 1u << 10,    /* all streams of the context at the default priority */
    AKZ_OPT_NO_RESIDENT_LEVELS = 1u << 11 /* the tile kernels (k_front_fed + k_fed_pair launches) also for the levels small enough
                                           * to live on one compute unit, instead of k_level_resident */
};
typedef struct akz_options {
    uint32_t struct_size;     /* sizeof(akz_options) of the caller (lets the struct grow) */
    uint32_t flags;           /* AKZ_OPT_* */
    uint32_t fed_block;       /* most FED steps fused per launch, 1..8; 0 = default (8; the first octave stops at 4) */
    uint32_t sup_capacity;    /* candidates per frame the parallel suppression is sized for; 0 = 4 x max_keypoints */
    uint32_t max_candidates;  /* capacity of each per-(frame, level) extrema candidate list; 0 = max_keypoints */
    uint32_t desc_tile_shift; /* log2 tile edge of the descriptor visiting order, 2..9; 0 = default (5) */
    uint32_t stream_waves;    /* waves a row-streaming launch aims for (sets its row-segment length); 0 = default (8192) */
    uint32_t stream_min_waves; /* launches that cannot field this many streaming waves take the tile kernel; 0 = default (2048) */
    uint32_t arith;           /* AKZ_ARITH_* bits: which of the reference's un-vendored arithmetic orders the filters use; 0 = default */
    uint32_t cu_ss;           /* CU partitioning (hipExtStreamCreateWithCUMask), 0 = none: the scale-space and determinant streams run on the */
    uint32_t cu_kp;           /* FIRST cu_ss compute units of every XCD, the keypoint stream on the LAST cu_kp (1..32 each; measured: DESIGN.md 5).
                               * Experiment-only: a masked stream is created by an API that takes neither a priority nor
                               * hipStreamNonBlocking, so it OVERRIDES AKZ_OPT_EQUAL_PRIORITY (no priorities at all) and is a
                               * blocking stream — it synchronises implicitly with the NULL stream of the process.  The same
                               * holds for HM_OPT_CU_MASK against HM_OPT_STREAM_PRIORITY. */
    uint32_t resident_min_frames; /* fewest frames of a call for which k_level_resident (one workgroup per frame) takes the levels that
                               * fit one compute unit; 0 = default (3/8 of the device's compute units: a call of fewer frames leaves
                               * most of the chip idle under it and keeps the tile kernels); 1 = always (tests) */
    uint32_t reserved[4];     /* must be zero */
} akz_options;
/* akz_options.arith — the only option that CHANGES RESULTS.  Three pieces of the reference's arithmetic live in crates that
 * are not vendored in rust-cv/cv, and its known answers (399 / 343 descriptors, 11 matches) come out the same under all
 * eight combinations, so none can be ruled out from here:
 *   wide::f32x4::reduce_add of the four filter lanes (akaze/src/image.rs:246-247, :324-325):  ((a0 + a1) + a2) + a3  [default]
 *                                                                                     or  (a0 + a1) + (a2 + a3)   [REDUCE_PAIRWISE]
 *   wide::f32x4::mul_add in the filters' accumulation (image.rs:246, :324):  multiply, then add  [default]  or one fused operation [FMA]
 *   ndarray's sum() over a 2 x 2 window in half_size (image.rs:160-166):     (a + b) + (c + d)   [default]  or  ((a + b) + c) + d  [HALF_SEQUENTIAL]
 * The default is what the crate sources imply for a default x86-64 build.  Every combination is a complete copy of the
 * scale-space kernels (no cost in the default one) and is held bit for bit to the oracle under the same ORC_OPT_* switches. */
enum { AKZ_ARITH_REDUCE_PAIRWISE = 1u << 0, AKZ_ARITH_FMA = 1u << 1, AKZ_ARITH_HALF_SEQUENTIAL = 1u << 2 };
/* akz_create with explicit options (NULL = defaults = akz_create). */
int32_t akz_create_ex(const akz_config* cfg, int32_t device, int32_t max_w, int32_t max_h,
                      int32_t max_batch, uint32_t max_keypoints, const akz_options* opts, akz_ctx** out);

/* Akaze::extract on a DynamicImage::ImageLuma8 (akaze/src/lib.rs:295, image.rs:47-56):
 * img is host memory, h rows of w bytes, `stride` bytes apart.  Outputs are index-aligned,
 * ordered by response descending, exactly as the reference returns them. */
int32_t akz_extract_gray_u8(akz_ctx* ctx, const uint8_t* img, int32_t w, int32_t h, int32_t stride,
                            akz_keypoint* kps, akz_descriptor* descs, uint32_t cap, uint32_t* n_out);

/* Akaze::extract on a DynamicImage::ImageLuma16 (akaze/src/image.rs:57-66: f32::from(v) / 65535f32 per
 * pixel); stride in elements. */
int32_t akz_extract_gray_u16(akz_ctx* ctx, const uint16_t* img, int32_t w, int32_t h, int32_t stride,
                             akz_keypoint* kps, akz_descriptor* descs, uint32_t cap, uint32_t* n_out);

/* Akaze::extract_from_gray_float_image (akaze/src/lib.rs:309): f32 pixels in [0,1], stride in
 * elements. */
int32_t akz_extract_gray_f32(akz_ctx* ctx, const float* img, int32_t w, int32_t h, int32_t stride,
                             akz_keypoint* kps, akz_descriptor* descs, uint32_t cap, uint32_t* n_out);

/* Akaze::extract on a COLOUR DynamicImage (akaze/src/image.rs:45-46: `input_image.grayscale()` first): 8- and 16-bit
 * RGB(A) become Luma8 / Luma16 by the `image` crate's integer Rec. 709 luma, (2126 R + 7152 G + 722 B) / 10000
 * truncating, and take the arms of image.rs:47-66; Rgb32F / Rgba32F keep their floats and take `to_luma()` per pixel
 * (image.rs:87-106: the same weights in f64, narrowed to f32).  The `image` crate is not vendored in the reference:
 * colour-input parity is unpinned (oracle/color_oracle.c restates the published formula).  pixels: host memory, h rows
 * of `stride` ELEMENTS, `channels` = 3 or 4 interleaved samples per pixel (alpha ignored); fmt = AKZ_FMT_* of a sample. */
int32_t akz_extract_color(akz_ctx* ctx, const void* pixels, int32_t fmt, int32_t channels, int32_t w, int32_t h, int32_t stride,
                          akz_keypoint* kps, akz_descriptor* descs, uint32_t cap, uint32_t* n_out);

/* Pixel formats of the batched / device entry points: the arms of GrayFloatImage::from_dynamic
 * (akaze/src/image.rs:45-109) that need no colour conversion. */
enum { AKZ_FMT_U8 = 0 /* Luma8: v / 255 */, AKZ_FMT_F32 = 1 /* GrayFloatImage, [0,1] */, AKZ_FMT_U16 = 2 /* Luma16: v / 65535 */ };

/* Batched extract: n host images of identical size (what a caller looping Akaze::extract over
 * frames does, cv-sfm/src/lib.rs:2200-2204).  fmt: AKZ_FMT_*; stride in elements.  kps/descs hold
 * n * cap_per_img entries, frame i at offset i*cap_per_img; n_out[i] = count of frame i. */
int32_t akz_extract_batch(akz_ctx* ctx, const void* const* imgs, int32_t fmt, int32_t n, int32_t w,
                          int32_t h, int32_t stride, akz_keypoint* kps, akz_descriptor* descs,
                          uint32_t cap_per_img, uint32_t* n_out);

/* Zero-copy batched extract for a device-resident pipeline: d_imgs is ONE device buffer of n
 * frames (frame-major, tightly packed w*h elements each), outputs are device buffers laid out
 * as in akz_extract_batch; d_n_out is a device array of n counts.  Work is enqueued on the
 * context's stream; `stream_to_wait` (a hipStream_t passed as void*, may be NULL) is the stream
 * that produced d_imgs.  The call returns after enqueueing; akz_sync() waits. */
int32_t akz_extract_batch_device(akz_ctx* ctx, const void* d_imgs, int32_t fmt, int32_t n, int32_t w,
                                 int32_t h, void* d_kps, void* d_descs, uint32_t cap_per_img,
                                 void* d_n_out, void* stream_to_wait);
int32_t akz_sync(akz_ctx* ctx);
/* The context's hipStream_t, as void*, so callers can order their own work after ours. */
void* akz_stream(akz_ctx* ctx);

/* Scale-space only (BASELINE.json configs[1] "scale-space kernels only"): runs A1..A11 of
 * SURVEY.md §8a — create_nonlinear_scale_space (akaze/src/lib.rs:193-258) + detector_response
 * (akaze/src/detector_response.rs:33-57) — on n device-resident frames and leaves Lt/Lx/Ly/Ldet
 * of every level in the context. */
int32_t akz_scale_space_device(akz_ctx* ctx, const void* d_imgs, int32_t fmt, int32_t n, int32_t w,
                               int32_t h, void* stream_to_wait);

/* After a call answered AKZ_E_INTERNAL: which internal list of which frame of that batch overflowed, and what it
 * would have needed (the reference's Vecs grow without bound, akaze/src/lib.rs:169-171 maximum_features =
 * usize::MAX; here every list has a capacity fixed at akz_create_ex: max_keypoints per frame — at most 262144 — and
 * akz_options.max_candidates raw extrema per (frame, level)).  flags bit 0: a per-level candidate list (needed_candidates
 * = the longest list of the frame); bit 1: the frame's keypoint lists after suppression. */
typedef struct akz_overflow_info {
    uint32_t flags;
    uint32_t needed_candidates;   /* largest per-level extrema count of the frame */
    uint32_t candidate_capacity;
    uint32_t keypoint_capacity;
} akz_overflow_info;
int32_t akz_last_overflow(akz_ctx* ctx, akz_overflow_info* per_frame, uint32_t cap, uint32_t* n_frames);

/* ---- pyramid introspection / parity taps (what Akaze::allocate_evolutions returns) ---- */
typedef struct akz_level_info {
    int32_t width, height;
    uint32_t octave, sublevel;
    double esigma, etime;
    uint32_t n_fed_steps;
    uint32_t deriv_sigma; /* round(esigma*derivative_factor/2^octave), detector_response.rs:11-13 */
} akz_level_info;
int32_t akz_num_levels(akz_ctx* ctx, int32_t w, int32_t h, int32_t* n_levels);
int32_t akz_level(akz_ctx* ctx, int32_t w, int32_t h, int32_t level, akz_level_info* out);
int32_t akz_fed_tau(akz_ctx* ctx, int32_t w, int32_t h, int32_t level, double* tau, uint32_t cap,
                    uint32_t* n_out);
enum { AKZ_BUF_LT = 0, AKZ_BUF_LSMOOTH = 1, AKZ_BUF_LX = 2, AKZ_BUF_LY = 3, AKZ_BUF_LDET = 4,
       AKZ_BUF_LFLOW = 5 };
/* Copy one level buffer of frame `img` of the most recent batch to host `out` (w*h floats). */
int32_t akz_debug_get_level(akz_ctx* ctx, int32_t img, int32_t level, int32_t which, float* out);
/* Contrast factor (compute_contrast_factor, akaze/src/contrast_factor.rs:16-64) of frame img. */
int32_t akz_debug_get_contrast(akz_ctx* ctx, int32_t img, double* out);
/* Keypoint list after a named stage of frame img of the most recent batch:
 * 0 = find_scale_space_extrema output (scale_space_extrema.rs:14-143),
 * 1 = after do_subpixel_refinement + orientation (:297-362), 2 = after sort+truncate (lib.rs:326-327). */
int32_t akz_debug_get_keypoints(akz_ctx* ctx, int32_t img, int32_t stage, akz_keypoint* out,
                                uint32_t cap, uint32_t* n_out);

/* The transcendentals of the orientation and descriptor stages as the DEVICE evaluates them (f32::atan2 at
 * scale_space_extrema.rs:242, f32::cos / sin at descriptors.rs:70-71): which 0: out = atan2f(y, x); 1: sinf(x); 2: cosf(x)
 * (y ignored).  Host buffers.  The tests hold these to the host libm within 1 ulp. */
int32_t akz_debug_portable_math(akz_ctx* ctx, int32_t which, const float* x, const float* y, uint32_t n, float* out);

/* Window membership of orientation samples (weighted gradient (x[i], y[i]); scale_space_extrema.rs:242-287), bit w = the
 * sample's angle lies in window w: fast[i] as the descriptor kernel decides it (an f32 estimate of the angle, the exact
 * expression inside a band around the windows' end points), exact[i] from the exact expression alone, fell_back[i] = 1 where
 * the band applied.  fast == exact for every input is the kernel's contract; the tests drive this with end points +- a few
 * ulp, axes, zeros, denormals and huge operands.  Host buffers. */
int32_t akz_debug_orientation_masks(akz_ctx* ctx, const float* x, const float* y, uint32_t n, uint64_t* fast, uint64_t* exact,
                                    uint32_t* fell_back);
/* The error BOUND of that estimate (tools/ubench/atan_bound.c, tests/test_oracle_math.py: 1.84e-6 in total against the band
 * of 8e-6) takes one thing from the instruction set's specification: v_rcp_f32 is accurate to 1 ulp.  This checks it on the
 * device in use: the instruction against 1 / x in f64 for EVERY f32 whose bit pattern lies in [lo_bits, hi_bits] (positive
 * normal numbers); *max_ulps = the largest error in ulps of the correctly rounded quotient. */
int32_t akz_debug_rcp_error(akz_ctx* ctx, uint32_t lo_bits, uint32_t hi_bits, double* max_ulps);

/* ---- stand-alone image ops (akaze::image public API, akaze/src/image.rs:202-389) ---- */
/* gaussian_kernel(r, kernel_size) — image.rs:360-374.  Host-only scalar math. */
int32_t akz_gaussian_kernel(float r, uint32_t kernel_size, float* out);
/* horizontal_filter / vertical_filter / separable_filter — image.rs:202-340: clamp-border
 * correlation with the reference's 4-lane accumulation order.  Host buffers in and out. */
int32_t akz_horizontal_filter(akz_ctx* ctx, const float* img, int32_t w, int32_t h,
                              const float* kernel, uint32_t ksize, float* out);
int32_t akz_vertical_filter(akz_ctx* ctx, const float* img, int32_t w, int32_t h,
                            const float* kernel, uint32_t ksize, float* out);
/* GrayFloatImage::half_size — image.rs:154-199. out is (w/2)*(h/2). */
int32_t akz_half_size(akz_ctx* ctx, const float* img, int32_t w, int32_t h, float* out);

/* ---- brute-force Hamming matcher (space::LinearKnn{metric: Hamming} + bitarray::BitArray<64>) ---- */
/* Bicubic colour sampling at keypoints: cv-sfm/src/bicubic.rs:34-68 (interpolate_bicubic) as called by
 * VSlam::kps_descriptors (cv-sfm/src/lib.rs:2207-2216) on image.to_rgb8() at kp.point.  rgb: host, h rows of
 * `stride` bytes, 3 bytes per pixel; colors [n][3].  A 4x4 neighbourhood that leaves the image gives the
 * reference's default [0,0,0]. */
int32_t akz_sample_colors_rgb8(akz_ctx* ctx, const uint8_t* rgb, int32_t w, int32_t h, int32_t stride,
                               const akz_keypoint* kps, uint32_t n, uint8_t* colors);

typedef struct hm_ctx hm_ctx;
int32_t hm_create(int32_t device, uint32_t max_queries, uint32_t max_targets, hm_ctx** out);
/* hm_create with kernel-selection flags (0 = defaults): the k-NN kernel is the FP4 MFMA one (64 resident queries per
 * wave, target tiles by LDS-DMA) unless a flag selects its register-staged predecessor (HM_OPT_NO_LDS_DMA), the int8
 * MFMA or the xor/popcount VALU kernel (k = 2 only); all four are bit-identical. */
enum { HM_OPT_NO_FP4 = 1u << 0, HM_OPT_NO_MFMA = 1u << 1, HM_OPT_STREAM_PRIORITY = 1u << 2, HM_OPT_NO_LDS_DMA = 1u << 3,
       HM_OPT_CU_SHIFT = 16, HM_OPT_CU_MASK = 0x3Fu << 16 /* bits 16..21: the matcher's stream on the LAST n compute units of every XCD (0 = all) */ };
int32_t hm_create_ex(int32_t device, uint32_t max_queries, uint32_t max_targets, uint32_t flags, hm_ctx** out);
int32_t hm_destroy(hm_ctx* ctx);
/* LinearKnn::knn(q, 2) for every query (akaze/tests/estimate_pose.rs:82-88): out[2*i+0/1] are the
 * nearest and second-nearest targets of query i under Hamming distance over all 64 bytes, ordered
 * by (distance, index) ascending — the lowest index wins ties.  nt < 2 is AKZ_E_INVALID (the
 * reference asserts two neighbours, estimate_pose.rs:89). Host buffers. */
int32_t hm_knn2(hm_ctx* ctx, const akz_descriptor* q, uint32_t nq, const akz_descriptor* t,
                uint32_t nt, akz_neighbor* out);
/* LinearKnn{metric: Hamming, iter: t}.knn(q, k) for k = 1, 2 or 3 — the registration path asks for 3
 * (cv-sfm/src/lib.rs:1474).  out[nq][k], ascending (distance, index); the reference returns min(k, nt)
 * neighbours, here the slots past nt hold {index 2^22 - 1, distance 1023}. */
int32_t hm_knn(hm_ctx* ctx, const akz_descriptor* q, uint32_t nq, const akz_descriptor* t, uint32_t nt,
               uint32_t k, akz_neighbor* out);
/* LinearKnn keeps its targets (`iter`) and is asked once per query (akaze/tests/estimate_pose.rs:82-88).  hm_set_targets
 * uploads the target set once; hm_knn_targets answers queries — one, or a batch — against the resident copy, results as
 * hm_knn's.  Any other host-buffer call on the context (hm_knn, hm_knn2, hm_match, hm_hash_bag) reuses the staging buffer
 * and ends the residency: hm_knn_targets then returns AKZ_E_INVALID rather than search stale data. */
int32_t hm_set_targets(hm_ctx* ctx, const akz_descriptor* t, uint32_t nt);
/* The number of the hm_set_targets call whose targets the context holds, 0 when nothing is resident.  A LinearKnn mirror keeps
 * the value its own upload produced and compares before every hm_knn_targets: any other upload or residency-ending call on the
 * context in between changes it (a pointer / length comparison of the host array would not notice a re-used allocation). */
uint64_t hm_targets_generation(hm_ctx* ctx);
int32_t hm_knn_targets(hm_ctx* ctx, const akz_descriptor* q, uint32_t nq, uint32_t k, akz_neighbor* out);
/* Device-resident multi-view form of the same call (cv-sfm/src/lib.rs:1468-1486: every feature of the new
 * frame against each of up to 32 recent views): d_q [cap][64] + count d_nq, d_views [..][cap][64] + counts
 * d_nviews, view v of this call = block view_idx[v]; d_out [n_views][cap][k].  Stream-ordered after
 * stream_to_wait; results are complete on hm_stream().  The landmark bookkeeping that follows in the
 * reference (HashMap dedup, merge rules) is control plane and stays with the caller. */
int32_t hm_knn_views_device(hm_ctx* ctx, const void* d_q, const void* d_nq, const void* d_views,
                            const void* d_nviews, uint32_t cap_per_img, const uint32_t* view_idx,
                            uint32_t n_views, uint32_t k, void* d_out, void* stream_to_wait);
/* The batched form: problem p = every feature of query block iq[p] (of d_q, count d_nq[iq[p]]) against target block it[p]
 * (of d_t), k neighbours each — the window of recent views of every frame of a micro-batch (n_frames x K problems) in one
 * call; every distinct block is recoded once.  iq / it host index arrays, n_probs <= 65535; d_out [n_probs][cap][k], slots past
 * a target block's count as in hm_knn. */
int32_t hm_knn_batch_device(hm_ctx* ctx, const void* d_q, const void* d_nq, const void* d_t, const void* d_nt,
                            uint32_t cap_per_img, const uint32_t* iq, const uint32_t* it, uint32_t n_probs, uint32_t k,
                            void* d_out, void* stream_to_wait);
/* What follows hm_knn_views_device in the reference (cv-sfm/src/lib.rs:1489-1542): per feature, the best distance of
 * every distinct landmark among its n_views x k neighbours, the three best landmarks (ascending (distance, landmark
 * key): the reference's HashMap leaves the order of equal distances unspecified), and the decision of :1516-1532 —
 * 1: best[0] is a unique match (best[0].d + better_by <= best[1].d); 2: best[0], best[1] are merge candidates
 * (best[1].d + better_by <= best[2].d; the caller still applies are_landmarks_sharing_view); 0: neither, or fewer
 * than three distinct landmarks.  d_knn = hm_knn_views_device's d_out [n_views][cap][k]; d_landmarks
 * [view blocks][cap] u32 = the landmark key observed by each feature of each stored view (reconstruction.views[v]
 * .landmarks); d_best [cap][3] {landmark, distance} (0xFFFFFFFF when absent), d_decision [cap] u32. */
int32_t hm_best_of_views_device(hm_ctx* ctx, const void* d_knn, const void* d_nq, uint32_t cap_per_img,
                                const uint32_t* view_idx, uint32_t n_views, uint32_t k, const void* d_landmarks,
                                const void* d_nviews, uint32_t better_by, void* d_best, void* d_decision,
                                void* stream_to_wait);
/* The same for every frame of a micro-batch in one launch: d_knn = hm_knn_batch_device's output laid out
 * [n_frames][n_views][cap][k] (problem f * n_views + v = frame f against its v-th view), frame f's count d_nq[iq[f]], its views
 * view_idx[f * n_views + v]; d_best [n_frames][cap][3], d_decision [n_frames][cap].  n_frames <= 65535. */
int32_t hm_best_of_views_batch_device(hm_ctx* ctx, const void* d_knn, const void* d_nq, const uint32_t* iq, uint32_t cap_per_img,
                                      const uint32_t* view_idx, uint32_t n_frames, uint32_t n_views, uint32_t k,
                                      const void* d_landmarks, const void* d_nviews, uint32_t better_by, void* d_best,
                                      void* d_decision, void* stream_to_wait);
/* What follows the decision in register_frame_subset (cv-sfm/src/lib.rs:1516-1532, 1549-1563, 1583-1604), for every frame of
 * a micro-batch.  original_matches: a feature with decision 1 is the match ([best0], feature); a feature with decision 2
 * whose two landmarks share no view — are_landmarks_sharing_view (:1528) is the caller's graph test, its verdict is
 * d_merge_ok [n_frames][cap] u8 (non-zero: the merge may be attempted), NULL when no merge candidate passes — is the match
 * ([best0, best1], feature).  landmark_counts covers every landmark of every original match, both landmarks of a merge
 * included, and a match survives only if each of its landmarks was counted once (:1549-1563).  A survivor whose
 * triangulation is None is dropped (:1583-1604, filter_map): d_world is the caller's table of homogeneous world points,
 * rows [0, n_world) indexed by landmark key (triangulate_landmark_robust) and — only with a merge mask — rows
 * n_world + f * cap + j = triangulate_merged_landmark_robust of frame f's feature j; a row with w < 0 (impossible for a
 * Projective point) or a landmark key >= n_world says "None".  d_best / d_decision / d_nq / iq as
 * hm_best_of_views_batch_device wrote / took them; d_pairs [n_frames][cap][2] u32 {feature, world row} in ascending feature
 * order, d_npairs [n_frames] — the pair lists rs_p3p_arrsac_batch_device takes (with n_world + n_frames * cap rows when a
 * merge mask is given).  cap_per_img <= 8192.  (Ascending feature order: the order for a caller that shuffles the
 * matches itself, RS_BATCH_SHUFFLE; the reference's own order is the next entry point's.) */
int32_t hm_landmark_matches_batch_device(hm_ctx* ctx, const void* d_best, const void* d_decision, const void* d_merge_ok,
                                         const void* d_nq, const uint32_t* iq, uint32_t cap_per_img, uint32_t n_frames,
                                         const void* d_world, uint32_t n_world, void* d_pairs, void* d_npairs, void* stream_to_wait);
/* The same lists IN THE ORDER THE REFERENCE HANDS TO THE CONSENSUS (cv-sfm/src/lib.rs:1561-1574): original_matches is stably
 * sorted by Reverse(sum over the match's landmarks of landmark(..).observations.len()) before model_inliers (:1619-1622) sees
 * it — matches on well-observed landmarks first, equal sums in feature order — and a consensus that samples by position
 * (ARRSAC does) depends on it.  d_obs_counts [n_world] u32: observations of every landmark key (the caller's graph knows them;
 * a key >= n_world counts 0); both landmarks of a merged match count.  The sort runs on the device, in the kernel that
 * builds the list (a stable LSD radix sort in LDS); with d_obs_counts == NULL this is hm_landmark_matches_batch_device.
 * Pass the lists to rs_p3p_arrsac_batch_device WITHOUT RS_BATCH_SHUFFLE to keep the order. */
int32_t hm_landmark_matches_ordered_batch_device(hm_ctx* ctx, const void* d_best, const void* d_decision, const void* d_merge_ok,
                                                 const void* d_obs_counts, const void* d_nq, const uint32_t* iq, uint32_t cap_per_img,
                                                 uint32_t n_frames, const void* d_world, uint32_t n_world, void* d_pairs,
                                                 void* d_npairs, void* stream_to_wait);
/* hm_landmark_matches_batch_device without a merge mask (equals the reference when no decision-2 match passes the graph test). */
int32_t hm_landmark_pairs_batch_device(hm_ctx* ctx, const void* d_best, const void* d_decision, const void* d_nq, const uint32_t* iq,
                                       uint32_t cap_per_img, uint32_t n_frames, const void* d_world, uint32_t n_world,
                                       void* d_pairs, void* d_npairs, void* stream_to_wait);
/* matching()/symmetric_matching() of tutorial ch5 main.rs:154-200 and cv-sfm/src/lib.rs:3097-3133,
 * and match_descriptors() of akaze/tests/estimate_pose.rs:78-97.
 *   rule 0: accept iff d0 + param_u <  d1   (tutorial, param_u = 24)
 *   rule 1: accept iff d0 + param_u <= d1   (cv-sfm better_by = 24); returns no matches when either
 *           side has fewer than 2 descriptors (cv-sfm/src/lib.rs:3099-3101)
 *   rule 2: accept iff (f32)d0 < (f32)d1 * param_f   (Lowe ratio, estimate_pose.rs:91-93)
 * symmetric != 0 keeps [a,b] only when the reverse match of b is a.  pairs = [a0,b0,a1,b1,...],
 * ascending a. */
enum { HM_RULE_BETTER_BY_STRICT = 0, HM_RULE_BETTER_BY = 1, HM_RULE_LOWE = 2 };
int32_t hm_match(hm_ctx* ctx, const akz_descriptor* a, uint32_t na, const akz_descriptor* b,
                 uint32_t nb, int32_t rule, uint32_t param_u, float param_f, int32_t symmetric,
                 uint32_t* pairs, uint32_t cap, uint32_t* n_out);
/* Device-resident batched form: `n_pairs` independent (a,b) problems.  d_a/d_b are frame-major
 * descriptor blocks of cap_per_img entries each; d_na/d_nb the per-frame counts (device).
 * Problem p matches block ia[p] of d_a against block ib[p] of d_b (host index arrays).
 * d_pairs holds n_pairs*cap_per_img [a,b] pairs; d_n_out n_pairs counts.  Enqueued on the hm
 * context's stream after `stream_to_wait`. */
int32_t hm_match_batch_device(hm_ctx* ctx, const void* d_a, const void* d_na, const void* d_b,
                              const void* d_nb, uint32_t cap_per_img, const uint32_t* ia,
                              const uint32_t* ib, uint32_t n_pairs, int32_t rule, uint32_t param_u,
                              float param_f, int32_t symmetric, void* d_pairs, void* d_n_out,
                              void* stream_to_wait);
int32_t hm_sync(hm_ctx* ctx);

/* ---- frame-level place recognition ----
 * HammingHasher<64, 512>::hash_bag (cv-sfm/src/lib.rs:672; hasher built from the 4096-word codebook at :216):
 * every feature sets the hash bit of its nearest codeword (lowest index among equal distances), bit w at byte
 * w >> 3, position w & 7.  hash = n_codewords / 8 bytes (n_codewords a multiple of 32); words[i] (optional) =
 * {nearest codeword, distance} of feature i.  The hashing crate (hamming-lsh 0.3.2) is not vendored in the
 * reference: parity unpinned (oracle/lsh_oracle.c). */
int32_t hm_hash_bag(hm_ctx* ctx, const akz_descriptor* feats, uint32_t n, const akz_descriptor* codewords,
                    uint32_t n_codewords, uint8_t* hash, akz_neighbor* words);
/* Batched device-resident form: frame f = d_descs block f ([cap_per_img][64]) with count d_counts[f];
 * d_codewords [n_codewords][64]; d_hash [n_frames][n_codewords / 8]; d_words [n_frames][cap_per_img] of
 * akz_neighbor, required: it is the pass's intermediate.  Stream-ordered after stream_to_wait on hm_stream(). */
int32_t hm_hash_bag_device(hm_ctx* ctx, const void* d_descs, const void* d_counts, uint32_t cap_per_img,
                           uint32_t n_frames, const void* d_codewords, uint32_t n_codewords, void* d_hash,
                           void* d_words, void* stream_to_wait);
/* lsh_to_frame.knn_values(&lsh, k) (cv-sfm/src/lib.rs:622-624) as an exact search: the min(k, n) stored hashes
 * nearest to `query`, ascending (distance, index); hash_bytes a multiple of 4 (512 in cv-sfm). */
int32_t hm_hash_knn(hm_ctx* ctx, const uint8_t* query, const uint8_t* hashes, uint32_t n, uint32_t hash_bytes,
                    uint32_t k, akz_neighbor* out, uint32_t* n_out);
/* Optional timing of the k-NN kernel launches with HIP events on hm_stream() (bench.py's matcher roofline):
 * hm_timing_get waits for the pending events and returns the accumulated milliseconds / launch count. */
int32_t hm_timing_enable(hm_ctx* ctx, int32_t on);
int32_t hm_timing_get(hm_ctx* ctx, double* ms, uint64_t* launches, int32_t reset);
void* hm_stream(hm_ctx* ctx);

/* ---- two-view geometric verification (second phase) ---- */
typedef struct rs_ctx rs_ctx;
/* The context's stream (rs_stream) is created at the MOST urgent priority of the device: a consensus is a chain of hundreds
 * of small dependent launches, and behind bulk kernels of equal priority (a matcher, an extraction) every one of them queues
 * for compute units — the registration loop of bench.py ran 142.6 -> 126.5 ms per 256 frames on that change alone. */
int32_t rs_create(int32_t device, uint32_t max_matches, uint32_t max_hypotheses, rs_ctx** out);
int32_t rs_destroy(rs_ctx* ctx);
/* cv_pinhole::CameraIntrinsics::calibrate (cv-pinhole/src/lib.rs:108-117) and, with use_k1 != 0,
 * CameraIntrinsicsK1Distortion::calibrate (:191-202): pixel keypoints -> unit bearings [n][3].
 * intrinsics = {focal_x, focal_y, principal_x, principal_y, skew}.  Host-only scalar math. */
int32_t rs_calibrate(const double* intrinsics, int32_t use_k1, double k1, const akz_keypoint* kps, uint32_t n,
                     double* bearings);
/* Consensus::model_inliers(&EightPoint::new(), matches) (call sites akaze/tests/estimate_pose.rs:63-67,
 * tutorial ch5 main.rs:70-72, cv-sfm/src/lib.rs:1394-1406) with the sampler factored out: the caller
 * provides n_hyp minimal samples (8 match indices each, what arrsac draws from its RNG); every sample is
 * turned into an essential matrix (eight-point/src/lib.rs:43-58) and its four poses
 * (cv-pinhole/src/essential.rs:217-231), every pose is scored against all n matches with
 * CameraToCamera::residual (cv-core/src/pose.rs:249-295) < thresh, and the pose with the most inliers wins
 * (ties: lowest hypothesis, then lowest pose index).  best_pose = row-major 3x4 [R | t];
 * best_id = hypothesis*4 + pose, 0xFFFFFFFF if no sample produced a model (the reference returns None);
 * inlier_idx ascending. */
int32_t rs_essential_batch(rs_ctx* ctx, const double* bearings_a, const double* bearings_b, uint32_t n,
                           const uint32_t* sample_idx, uint32_t n_hyp, double thresh, double* best_pose,
                           uint32_t* best_id, uint32_t* inlier_idx, uint32_t cap, uint32_t* n_inliers);
/* The same consensus in arrsac's shape (arrsac::Arrsac::model_inliers; parameters as its builder exposes them at
 * vslam-sandbox/src/main.rs:105-117: initialization_hypotheses, max_candidate_hypotheses, block_size,
 * likelihood_ratio_threshold): the hypotheses are scored breadth-first, `block_size` matches at a time, and after
 * every block a pose is retired when
 *   RS_PRUNE_BOUND  its count plus the matches still to come cannot reach the best count so far (exact: the winner,
 *                   its count and its inlier set equal exhaustive scoring's — rs_essential_batch on the same samples);
 *   max_candidates  it is not among the max_candidates best once init_blocks blocks have been scored (0 = no cap);
 *                   poses are ranked by their exact distance to the best count (distances of 2047 and more rank
 *                   together, after all others), equal ranks in ascending pose id: the best-supported poses are never
 *                   the ones the cap retires;
 *   RS_PRUNE_SPRT   Wald's sequential test rejects it: (delta/eps)^c ((1-delta)/(1-eps))^(seen-c) > sprt_ratio with
 *                   eps = best count / seen and delta = sprt_delta, the inlier rate expected of a wrong model;
 *   RS_PRUNE_HALVE  the candidate cap halves with every block after init_blocks (ARRSAC's preemption function
 *                   f(i) = floor(M 2^-floor(i/B))), down to one survivor; with estimations_per_block == 0 the block
 *                   loop ends when the cap reaches 1 (a single survivor cannot be overtaken; stats count what ran).
 * sample_idx == NULL draws the n_hypotheses minimal samples on the device from xoshiro256++ streams seeded with
 * `seed` (rs_arrsac_samples reproduces them on the host).  The arrsac crate is not vendored in the reference: the
 * sampler, the retirement rules and their order are this library's, specified by oracle/arrsac_oracle.c and held to
 * it bit for bit (parity with the crate unpinned beyond the count pin of akaze/tests/estimate_pose.rs:75).
 * best_id / best_pose / inlier_idx as for rs_essential_batch. */
enum { RS_PRUNE_BOUND = 1u << 0, RS_PRUNE_SPRT = 1u << 1, RS_PRUNE_HALVE = 1u << 2 };
typedef struct rs_arrsac_params {
    uint32_t struct_size;      /* sizeof(rs_arrsac_params) */
    uint32_t n_hypotheses;     /* initialization_hypotheses */
    uint32_t block_size;       /* matches per scoring block */
    uint32_t init_blocks;      /* blocks scored before max_candidates applies */
    uint32_t max_candidates;   /* max_candidate_hypotheses (poses kept); 0 = no cap */
    uint32_t flags;            /* RS_PRUNE_* */
    double threshold;          /* inlier threshold on CameraToCamera::residual */
    double sprt_delta;         /* P(inlier | wrong model), e.g. 0.05 */
    double sprt_ratio;         /* likelihood ratio threshold, arrsac default 1e3 */
    uint64_t seed;             /* sampler seed (Xoshiro256PlusPlus::seed_from_u64 at the call sites) */
    uint32_t estimations_per_block; /* hypotheses generated after every block (from init_blocks on) from minimal samples
                                  * drawn among the inliers of the best pose so far — arrsac's inlier-guided
                                  * re-sampling; each is scored on all matches seen so far and joins the candidates;
                                  * 0 = off.  The context must hold n_hypotheses + this x blocks hypotheses. */
    uint32_t reserved;         /* must be zero */
} rs_arrsac_params;
typedef struct rs_arrsac_stats {
    uint32_t poses, survivors, blocks, reserved;
    uint64_t residuals_evaluated, residuals_exhaustive;
} rs_arrsac_stats;
int32_t rs_essential_arrsac(rs_ctx* ctx, const double* bearings_a, const double* bearings_b, uint32_t n,
                            const uint32_t* sample_idx, const rs_arrsac_params* params, double* best_pose,
                            uint32_t* best_id, uint32_t* inlier_idx, uint32_t cap, uint32_t* n_inliers,
                            rs_arrsac_stats* stats);
/* The same for Consensus::model_inliers(&LambdaTwist::new(), landmark matches) — the registration path's consensus
 * (cv-sfm/src/lib.rs:1619-1622; vslam-sandbox/src/main.rs:105-111: 16384 initialisation hypotheses, 1024 candidates):
 * 3-match samples, WorldToCamera::residual; bearings / world as for rs_p3p_batch. */
int32_t rs_p3p_arrsac(rs_ctx* ctx, const double* bearings, const double* world, uint32_t n,
                      const uint32_t* sample_idx, const rs_arrsac_params* params, double* best_pose,
                      uint32_t* best_id, uint32_t* inlier_idx, uint32_t cap, uint32_t* n_inliers,
                      rs_arrsac_stats* stats);
/* The minimal samples the two functions draw on the device for (seed, n): sample_size 8 (eight-point) or 3 (P3P). */
int32_t rs_arrsac_samples(uint64_t seed, uint32_t n, uint32_t n_hyp, uint32_t sample_size, uint32_t* sample_idx);
/* Consensus::model_inliers(&LambdaTwist::new(), matches) for 3D-2D registration (cv-sfm/src/lib.rs:1619-1622,
 * lambda-twist/tests/consensus.rs:59-61), sampler factored out as above: n_hyp sample triples; each gives up
 * to four WorldToCamera poses (lambda-twist/src/lib.rs:107-318), scored with WorldToCamera::residual
 * (cv-core/src/pose.rs:194-201) < thresh.  bearings [n][3] unit vectors, world [n][4] homogeneous points in the
 * reference's Projective form (xyz normalised, w = 1/distance). */
int32_t rs_p3p_batch(rs_ctx* ctx, const double* bearings, const double* world, uint32_t n,
                     const uint32_t* sample_idx, uint32_t n_hyp, double thresh, double* best_pose,
                     uint32_t* best_id, uint32_t* inlier_idx, uint32_t cap, uint32_t* n_inliers);
/* parity tap: inlier counts [n_hyp][4] of the last rs_essential_batch / rs_p3p_batch call */
int32_t rs_debug_counts(rs_ctx* ctx, uint32_t* counts, uint32_t cap);
/* parity tap: poses [n_hyp][4][12] (row-major [R | t]) and validity flags [n_hyp][4] of the last single-scene call —
 * what EightPoint::estimate / LambdaTwist::estimate returned for each minimal sample (eight-point/src/lib.rs:70-83,
 * lambda-twist/src/lib.rs:330-347) */
int32_t rs_debug_poses(rs_ctx* ctx, double* poses, uint32_t* ok, uint32_t n_hyp);

/* parity tap: CameraToCamera::residual (cv-core/src/pose.rs:249-295) of every (pose, match) as the device evaluates it.
 * poses [n_pose][12] row-major [R | t], bearings [n][3], host buffers.  paired == 0: out [n_pose][n].  paired != 0: out
 * [n_pose][2][n] — the pose and its mirror image [R | -t] (poses p and p + 2 of a hypothesis) through the path that
 * shares one eigen-decomposition between the two. */
int32_t rs_debug_residuals(rs_ctx* ctx, const double* poses, uint32_t n_pose, const double* bearings_a, const double* bearings_b,
                           uint32_t n, int32_t paired, double* out);
/* parity tap of the shortcut in front of the eigen-decomposition (rs_pair_far): out[pose * n + match] = 1 where
 * the epipolar-plane bound alone proves residual >= thresh for [R | t] and [R | -t]; tests hold it to rs_debug_residuals. */
int32_t rs_debug_far(rs_ctx* ctx, const double* poses, uint32_t n_pose, const double* bearings_a, const double* bearings_b,
                     uint32_t n, double thresh, uint8_t* out);

/* ---- two-view verification of a whole micro-batch, device-resident ----
 * What cv-sfm does for every frame pair the matcher produced (cv-sfm/src/lib.rs:1385-1412): shuffle the matches,
 * map them to calibrated bearing pairs (match_ix_kps; CameraIntrinsics::calibrate, cv-pinhole/src/lib.rs:108-117),
 * run the consensus (vslam-sandbox/src/main.rs:112-117: Arrsac, 8192 initialisation hypotheses, 1024 candidates),
 * keep the inliers.  Here all frame pairs of a micro-batch go through ONE chain of launches (grid = scenes x
 * hypotheses), reading the matcher's pair lists and the extractor's keypoints where they lie in HBM.
 *
 * rs_camera: CameraIntrinsics {focals, principal_point, skew} and, with use_k1 != 0, CameraIntrinsicsK1Distortion's k1. */
typedef struct rs_camera {
    double fx, fy, cx, cy, skew;
    double k1;
    int32_t use_k1;
    int32_t reserved;   /* must be zero */
} rs_camera;
/* Room for up to max_scenes frame pairs per call (rs_create leaves room for one); every scene gets the context's
 * max_matches matches and max_hypotheses hypotheses.  1 <= max_scenes <= 65535. */
int32_t rs_batch_reserve(rs_ctx* ctx, uint32_t max_scenes);
enum { RS_BATCH_SHUFFLE = 1u << 0 };   /* score the matches in a seeded shuffled order (the reference shuffles them with
                                        * the caller's rng, cv-sfm/src/lib.rs:1385): position j of scene s holds the
                                        * match with the j-th smallest 32-bit key splitmix64((seed_s ^ 0x5851F42D4C957F2D)
                                        * + 0xD1342543DE82EF95 j) >> 32, ties in index order; needs cap_per_img <= 8192 */
/* Scene s (0 <= s < n_scenes): pair list s of d_pairs ([cap_per_img][2] u32, count d_npairs[s] — hm_match_batch_device's
 * outputs), whose [a, b] index the keypoints of block ia[s] of d_kps_a and block ib[s] of d_kps_b ([..][cap_per_img]
 * akz_keypoint — akz_extract_batch_device's output).  Each scene runs rs_essential_arrsac's procedure on its calibrated
 * bearing pairs with the sampler seed  params->seed + 0x9E3779B97F4A7C15 * s  (scene 0 = the caller's seed; match
 * counts above max_matches are clipped).  Outputs, device arrays indexed by scene: d_pose [12] f64 row-major [R | t],
 * d_best_id u32 (0xFFFFFFFF: no model — fewer than 8 matches, or no sample produced one), d_inliers [cap_per_img] u32
 * (ascending indices into the scene's pair list), d_n_inliers u32, d_stats (optional) rs_arrsac_stats.  Enqueued on
 * rs_stream() after stream_to_wait (may be NULL); returns after enqueueing, rs_sync() waits.  Specified by
 * oracle/arrsac_oracle.c (orc_arrsac_pairs) and held to it bit for bit. */
int32_t rs_essential_arrsac_batch_device(rs_ctx* ctx, const void* d_kps_a, const void* d_kps_b, uint32_t cap_per_img,
                                         const uint32_t* ia, const uint32_t* ib, const void* d_pairs, const void* d_npairs,
                                         uint32_t n_scenes, const rs_camera* cam_a, const rs_camera* cam_b,
                                         const rs_arrsac_params* params, uint32_t flags, void* d_pose, void* d_best_id,
                                         void* d_inliers, void* d_n_inliers, void* d_stats, void* stream_to_wait);
/* The registration path's consensus for a micro-batch of new frames (cv-sfm/src/lib.rs:1571-1622: FeatureWorldMatch(bearing,
 * point) list -> single_view_consensus.model_inliers(&LambdaTwist, ..)).  Scene s: pair list s of d_pairs ([cap_per_img][2]
 * u32, count d_npairs[s]) whose entries are {feature index into keypoint block ik[s] of d_kps, index into d_world};
 * d_world [n_world][4] f64 homogeneous world points (the caller's triangulated landmarks).  A scene whose pair list names a
 * feature >= cap_per_img or a world point >= n_world is refused as a whole (no model; nothing is read out of bounds); the
 * two-view entry point above treats a pair that points outside its keypoint blocks the same way.   bearing = calibrate(keypoint); each
 * scene runs rs_p3p_arrsac's procedure with the scene seed as above.  Outputs as above (pose = WorldToCamera [R | t];
 * no model: fewer than 3 matches or no sample produced a pose).  Specified by oracle/arrsac_oracle.c (orc_p3p_arrsac_pairs). */
int32_t rs_p3p_arrsac_batch_device(rs_ctx* ctx, const void* d_kps, uint32_t cap_per_img, const uint32_t* ik, const void* d_pairs,
                                   const void* d_npairs, uint32_t n_scenes, const void* d_world, uint32_t n_world,
                                   const rs_camera* cam, const rs_arrsac_params* params, uint32_t flags, void* d_pose,
                                   void* d_best_id, void* d_inliers, void* d_n_inliers, void* d_stats, void* stream_to_wait);
int32_t rs_sync(rs_ctx* ctx);
void* rs_stream(rs_ctx* ctx);
/* parity tap: match count, calibrated bearings [n][3] (a, b) and scoring order [n] of scene `scene` of the last batched
 * call (any of the three pointers may be NULL) */
int32_t rs_debug_scene(rs_ctx* ctx, uint32_t scene, uint32_t* n, double* bearings_a, double* bearings_b, uint32_t* order,
                       uint32_t cap);
/* the same after rs_p3p_arrsac_batch_device: bearings [n][3], world points [n][4] */
int32_t rs_debug_scene_world(rs_ctx* ctx, uint32_t scene, uint32_t* n, double* bearings, double* world, uint32_t* order,
                             uint32_t cap);

/* ---- the exchange step of the frame-sharded front-end ----
 * Frames shard over the GPUs of a node as frame g -> rank g mod N (extraction is stateless per frame: akaze::Akaze is
 * Copy, akaze/src/lib.rs:108).  A new frame is matched against its recent predecessors g-1 .. g-k (cv-sfm/src/lib.rs:
 * 1462-1486; tracking_recent_frames = 32, cv-sfm/src/settings.rs:449-450), which live on the other ranks: the
 * descriptor blocks a rank has extracted — fixed capacity, [n_frames][cap_per_img][64] bytes + [n_frames] u32 counts,
 * exactly akz_extract_batch_device's outputs — travel by a ring shift (k = 1: a rank needs its predecessor's block) or
 * by an all-gather (k >= N - 1).  RCCL directly (ncclSend/ncclRecv/ncclAllGather on akz_comm_stream()), loaded with
 * dlopen("librccl.so.1") on first use.  One process per GPU; rank 0 makes the id, the host application hands it to the
 * other ranks by its own means, every rank creates its communicator.  Calls enqueue and return; they wait for
 * stream_to_wait (the producer of the send rows and last reader of the receive rows; may be NULL), consumers take
 * akz_comm_stream() as their stream_to_wait. */
typedef struct akz_comm akz_comm;
int32_t akz_comm_unique_id(uint8_t* id128);                    /* 128 bytes (ncclUniqueId) */
int32_t akz_comm_create(const uint8_t* id128, int32_t rank, int32_t world, int32_t device, akz_comm** out);
int32_t akz_comm_destroy(akz_comm* comm);
/* this rank's blocks -> rank + 1; the blocks of rank - 1 -> d_recv_descs / d_recv_counts (same shapes) */
int32_t akz_comm_shift_blocks(akz_comm* comm, const void* d_descs, const void* d_counts, uint32_t n_frames, uint32_t cap_per_img,
                              void* d_recv_descs, void* d_recv_counts, void* stream_to_wait);
/* every rank's blocks -> every rank: d_all_descs [world][n_frames][cap_per_img][64], d_all_counts [world][n_frames] */
int32_t akz_comm_allgather_blocks(akz_comm* comm, const void* d_descs, const void* d_counts, uint32_t n_frames, uint32_t cap_per_img,
                                  void* d_all_descs, void* d_all_counts, void* stream_to_wait);
int32_t akz_comm_sync(akz_comm* comm);
void* akz_comm_stream(akz_comm* comm);
int32_t akz_comm_rank(akz_comm* comm);
int32_t akz_comm_world(akz_comm* comm);
/* HIP-event time of the transfers on akz_comm_stream() (their exposed time is what a step loses to them), call and byte
 * counts since the last reset; `enable` switches the brackets on or off for the calls that follow. */
int32_t akz_comm_timing(akz_comm* comm, int32_t enable, double* ms, uint64_t* calls, uint64_t* bytes, int32_t reset);
const char* akz_comm_last_error_string(void);

/* ---- misc ---- */
const char* akz_strerror(int32_t status);
/* hipError_t of the most recent failing HIP call on this thread (0 if none) and its text. */
int32_t akz_last_hip_error(void);
const char* akz_last_hip_error_string(void);
const char* akz_version(void);
/* The ABI number: raised whenever a declared signature, struct layout or enum value of this header changes (additions
 * included).  A binding compares akz_abi_version() of the library it loaded with the AKZ_ABI_VERSION it was written against
 * and refuses to run on a mismatch (cv_amd/_lib.py, rust/akaze-mi355x/src/lib.rs, include/akaze.hpp do). */
#define AKZ_ABI_VERSION 8u
uint32_t akz_abi_version(void);

/* HIP-event timing of the kernel families of a batch (bench.py's roofline objects).  Kernel families (every id but the
 * phase ids below): each launch carries its own start / stop events (hipExtLaunchKernel — the dispatch's begin and end
 * timestamps, the duration rocprofv3's kernel trace reports), so a family's time is the sum of its kernels' own
 * durations whatever else shares the GPU.  Phases (AKZ_T_FED, _SCALE_SPACE, _EXTRACT, _DESCRIBE, _REFINE): an event
 * bracket on the stream, i.e. wall time including waits for the other streams.  akz_timing_enable(ctx, 1) turns both
 * kinds on, 2 the kernel families only (no extra packets in the streams: usable inside a timed region), 0 off.
 * `which` is an AKZ_T_* id;
 * the call returns the milliseconds, launch count and processed units accumulated since akz_timing_reset():
 * units = pixel-frames the launches covered (FED: pixel-steps, the contract's unit; AKZ_T_FED_PASS reports the
 * same launches with units = pixel-frames per launch, i.e. passes over memory). */
enum {
    AKZ_T_FED = 0,        /* k_fed_pair / k_fed_step (calculate_step)                          units: pixel-steps */
    AKZ_T_SCALE_SPACE = 1, /* the whole scale space of a call                                  units: frames */
    AKZ_T_EXTRACT = 2,    /* the whole extract call (closes on the keypoint stream)            units: frames */
    AKZ_T_FRONT0 = 3,     /* level-0 front end: pixels -> blur 1.6 -> Lt[0], {Lx,Ly}           units: pixel-frames */
    AKZ_T_FRONT_SG2 = 4,  /* level front end (blur 1.0, Lflow, {Lx,Ly}) at derivative sigma 2  units: pixel-frames */
    AKZ_T_FRONT_SG3 = 5,
    AKZ_T_FRONT_SG4 = 6,
    AKZ_T_DET_SG2 = 7,    /* second derivatives + determinant + extrema candidates, sigma 2    units: pixel-frames */
    AKZ_T_DET_SG3 = 8,
    AKZ_T_DET_SG4 = 9,
    AKZ_T_CONTRAST = 10,  /* contrast-factor passes                                            units: pixel-frames per pass */
    AKZ_T_DESCRIBE = 11,  /* M-LDB descriptors (keypoint stream)                               units: frames */
    AKZ_T_REFINE = 12,    /* sub-pixel refinement + orientation (keypoint stream)              units: frames */
    AKZ_T_FED_PASS = 13,  /* = AKZ_T_FED with units = pixel-frames per launch */
    AKZ_T_FED_T1 = 14,    /* k_fed_pair<T> launches by T = 1..8 (ids 14..21), units = pixel-frames per launch */
    AKZ_T_FRONT_FED_SG2 = 22, /* fused level front end + first FED launch (k_front_fed), sigma 2..4: ids 22..24,
                               * units = pixel-frames */
    AKZ_T_FRONT_FED_SG3 = 23,
    AKZ_T_FRONT_FED_SG4 = 24,
    AKZ_T_ORIENT_DESCRIBE_K = 25, /* the k_orient_describe launch itself (kernel timer inside the AKZ_T_DESCRIBE phase), units: frames */
    AKZ_T_FRONT_FED_DEEP_SG2 = 26, /* k_front_fed on levels BELOW the first octave (two-patch halo for launches of 5..8 steps; Lflow is */
    AKZ_T_FRONT_FED_DEEP_SG3 = 27, /* written when a later launch of the level reads it), sigma 2..4: ids 26..28, units: pixel-frames */
    AKZ_T_FRONT_FED_DEEP_SG4 = 28,
    AKZ_T_LEVEL_RESIDENT = 29, /* k_level_resident: front end + EVERY FED step of a level that fits one compute unit (octave 3 of a 1080p
                                * pyramid), one launch per level, one workgroup per frame; units: pixel-frames */
    AKZ_T_COUNT = 30
};
int32_t akz_timing_enable(akz_ctx* ctx, int32_t on);
int32_t akz_timing_reset(akz_ctx* ctx);
int32_t akz_timing_get(akz_ctx* ctx, int32_t which, double* ms, uint64_t* launches, uint64_t* units);

#ifdef __cplusplus
}
#endif
#endif /* AKZ_H */
