// akz_ctx.h — the context object behind akz_ctx* (internal).
#pragma once
#include <utility>

#include <hip/hip_ext.h>

#include "akz_common.h"

// One keypoint work record on the device; identical to akz_keypoint (28 B).
typedef akz_keypoint DevKp;

struct AkzTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;  // recorded, not yet resolved
    std::vector<hipEvent_t> pool;                            // recycled events
    hipEvent_t cur_start = nullptr;
    double ms = 0.0;
    uint64_t launches = 0, units = 0, units2 = 0;
};

// Every per-batch device buffer (pyramid, work lists, outputs).
struct AkzSet {
    std::vector<float*> Lt, Lsm, Ldet, Lflow;  // [level] -> frame-major f32 planes
    std::vector<float2*> Lxy;                  // [level] -> frame-major {Lx, Ly} planes (interleaved: every consumer
                                               // reads both derivatives at the same pixel)
    float* tmp = nullptr;       // ping-pong partner for FED steps, max_batch * P0
    void* d_in = nullptr;       // staged input frames (u8 or f32), max_batch * P0 * 4 bytes
    // contrast factor (contrast_factor.rs)
    unsigned long long* d_cmax = nullptr;  // [B] bit pattern of max f64 gradient magnitude^2
    uint32_t* d_hist = nullptr;            // [B][nbins]
    uint32_t* d_npoints = nullptr;         // [B]
    double* d_cthr = nullptr;              // [B][512] histogram bin thresholds in magnitude^2 space
    uint32_t* d_fine = nullptr;            // [B][2048] histogram over the f64 exponent + 6 mantissa bits of magnitude^2
    uint32_t* d_cflag = nullptr;           // [B] 1 = the frame needs the exact histogram pass
    double* d_contrast = nullptr;          // [B]
    float* d_invk = nullptr;               // [B][8]  (1/(k_o*k_o)) as f32 per octave (nonlinear_diffusion.rs:73)
    // keypoint stage
    uint32_t* d_ncand = nullptr;           // [B][32] candidates per (frame, level)
    void* d_cand_u = nullptr;              // [B][32][max_cand] CandU records in append order (akz_scale_space.hip)
    float* d_cand_nb = nullptr;            // [B][32][max_cand][8] determinant values around each sorted candidate
    uint2* d_cand = nullptr;               // [B][32][max_cand] {x | y << 16, response bits}, raster-sorted per level
    DevKp* d_cache = nullptr;              // [B][max_kp]  suppression cache (scale_space_extrema.rs:15)
    uint32_t* d_ncache = nullptr;          // [B]
    size_t zero_bytes = 0;                 // d_cmax .. end of d_fine: cleared by one memset at the start of a call
    uint32_t* d_lvl_slot = nullptr;        // [B][kAkzMaxLevels + 1] first cache slot pushed at every level (+ the total)
    uint32_t* d_cand_rows = nullptr;       // [B][sum over levels of (h + 1)] row-start tables of the raster-sorted candidate lists (k_cand_rows)
    void* d_chunk_yr = nullptr;            // [B][ceil(max_kp / 64)] float2 {ymin, ymax} of every 64-slot chunk of the cache (k_chunk_yrange)
    uint32_t* d_sup = nullptr;             // [B][sup_cap * (2 * 24 + 6)] scratch of the parallel suppression (k_sup_*)
    uint32_t* d_sup_flag = nullptr;        // [B] 1 = this frame takes the serial k_suppress
    uint32_t* d_big_flag = nullptr;        // [B] 1 = the serial pass's LDS list overflowed: the frame takes k_suppress_big
    void* d_big_act = nullptr;             // [B][max_kp] 16-byte active-list entries of k_suppress_big (contexts with max_kp > 8192 only)
    DevKp* d_kp_a = nullptr;               // [B][max_kp]  stage-0 list (find_scale_space_extrema output)
    uint32_t* d_n_a = nullptr;
    DevKp* d_kp_b = nullptr;               // [B][max_kp]  stage-1 list (refined + orientation), pre-compaction slots
    uint32_t* d_flag_b = nullptr;          // [B][max_kp]  keep flags
    DevKp* d_kp_c = nullptr;               // [B][max_kp]  stage-1 compacted
    uint32_t* d_n_c = nullptr;
    DevKp* d_kp_d = nullptr;               // [B][max_kp]  stage-2 sorted + truncated
    uint32_t* d_n_d = nullptr;
    akz_descriptor* d_desc_tmp = nullptr;  // [B][max_kp]  descriptors before dropping out-of-bounds keypoints
    uint32_t* d_flag_d = nullptr;          // [B][max_kp]
    uint32_t* d_perm = nullptr;            // [B][max_kp] spatially coherent visiting order for the descriptor stage
    DevKp* d_kp_out = nullptr;             // [B][max_kp]  final (internal copy used by the host-buffer API)
    akz_descriptor* d_desc_out = nullptr;  // [B][max_kp]
    uint32_t* d_n_out = nullptr;           // [B]
    // global key scratch of the bitonic sorts for lists longer than kAkzLdsSortKeys (null when they fit in LDS)
    unsigned long long* d_keys_kp = nullptr;    // [B][np2(max_kp)]
    unsigned long long* d_keys_cand = nullptr;  // [B][kAkzMaxLevels][np2(max_cand)]
};

struct akz_ctx {
    akz_config cfg;
    int device = 0;
    int n_cu = 256;           // compute units of the device (sizes the grids of the tile-walking kernels)
    int arith = 0;            // AKZ_ARITH_* bits (akz_options.arith): which copy of the scale-space kernels the context runs
    hipStream_t stream = nullptr;
    int max_w = 0, max_h = 0, max_batch = 0;
    uint32_t max_kp = 0;      // capacity of every per-frame keypoint list
    uint32_t max_cand = 0;    // capacity of each per-(frame, level) candidate list
    int desc_tile_shift = 5;  // log2 of the tile edge of the descriptor visiting order; env AKZ_DESC_TILE_SHIFT
    bool contrast_force_odd = false; // AKZ_CONTRAST_FINE=2: odd frames are sent to the exact pass regardless (test knob)
    bool contrast_fine = true; // contrast factor from the fine histogram of the max pass where that is unambiguous (AKZ_CONTRAST_FINE=0: always the exact histogram pass)
    int fed_block = 8;        // most FED steps fused per launch (1 = one launch per step; the first octave stops at 4); env AKZ_FED_BLOCK
    bool front_pair = true;   // two-frame packed front kernel (AKZ_FRONT_PAIR=0 selects the one-frame kernel)
    bool keep_all = false;    // keep per-level Lsmooth/Lflow (parity taps) instead of per-octave scratch
    bool stream_kernels = true;   // row-streaming kernels (k_det_stream) instead of the tile kernels (AKZ_OPT_TILE_KERNELS)
    int det_stream_waves = 8192;  // waves a streaming launch aims for (sets the row-segment length)
    size_t stream_min_waves = 2048;  // launches that cannot field this many streaming waves take the tile kernels

    AkzPlan plan;             // for (cur_w, cur_h)
    int cur_w = 0, cur_h = 0, cur_n = 0;

    // ---- device memory (one arena, carved in akz_ctx_prepare) ----
    void* arena = nullptr;
    size_t arena_bytes = 0;
    // Two complete buffer sets: call k works in set k&1, so the scale space of micro-batch k+1 (stream
    // `stream`, HBM-bound) overlaps the keypoint stage of micro-batch k (stream `stream_kp`, latency-bound).
    AkzSet sets[2];
    int nsets = 2;
    int cur = 0;               // set used by the most recent call
    uint64_t calls = 0;
    hipStream_t stream_kp = nullptr;
    // The determinant / candidate kernel of a level reads only that level's {Lx, Ly}: it is a side branch of the
    // Lt -> front -> FED -> Lt chain.  On its own stream it leaves that chain (a third of a single frame's launches)
    // and joins again before the candidate sort.
    hipStream_t stream_det = nullptr;
    bool det_side_stream = true;        // AKZ_OPT_SERIAL_DET clears it
    bool fuse_front_fed = true;         // k_front_fed where it applies (AKZ_OPT_SPLIT_FRONT_FED clears it)
    bool resident_levels = true;        // k_level_resident where a level fits one compute unit (AKZ_OPT_NO_RESIDENT_LEVELS clears it)
    int resident_min_frames = 0;        // ... for calls of at least this many frames (0: 3/8 of the compute units)
    hipEvent_t ev_level[kAkzMaxLevels] = {};   // {Lx, Ly} of level l written (recorded on `stream`)
    hipEvent_t ev_det_done = nullptr;          // every determinant kernel of the call finished (recorded on `stream_det`)
    bool sup_parallel = true;           // AKZ_SUP_PARALLEL=0: serial suppression only
    uint32_t sup_cap = 0;               // candidates per frame the parallel suppression is sized for
    void* d_color = nullptr;            // scratch of akz_sample_colors_rgb8 (image + keypoints + colours), grown on demand
    size_t color_bytes = 0;
    hipEvent_t ev_ss_done[2] = {nullptr, nullptr};  // pyramid + candidates of set b ready
    hipEvent_t ev_kp_done[2] = {nullptr, nullptr};  // keypoint stage of set b finished (pyramid reusable)
    bool kp_pending[2] = {false, false};
    hipEvent_t ev_input = nullptr;                  // orders the caller's producer stream before our scale-space stream
    AkzSet& S() { return sets[cur]; }
    // Host calls (akz_extract_batch & co): pinned, device-visible staging.  The input rows are gathered into h_in and
    // go up as one DMA; the final compaction kernel writes keypoints, descriptors, counts and the overflow flag
    // straight into h_out, so a call ends with ONE stream synchronisation instead of four blocking copies.  Used
    // while the block stays below kAkzHostStageMax bytes; larger batches take the plain copies.
    void* h_in = nullptr;
    size_t h_in_bytes = 0;
    void* h_out = nullptr;
    size_t h_out_bytes = 0;
    uint32_t* d_err = nullptr;             // [1] sticky device-side overflow flag
    void* d_ori = nullptr;                 // OriTables (orientation sample/window tables)
    void* d_desc = nullptr;                // DescTables (M-LDB cell + comparison tables)
    float* d_taps = nullptr;               // [kAkzMaxTaps + 1] Gaussian taps of the generic level-0 blur
    std::vector<float> h_taps;             // their host copy (must outlive the asynchronous upload)

    // timing (akz_timing_*)
    bool timing = false;
    bool timing_phases = false;     // phase timers (event brackets) as well as kernel timers
    int open_kernel_timer = -1;     // AKZ_T_* id of the kernel timer open on the launching thread, -1: none
    AkzTimer timers[AKZ_T_COUNT];   // indexed by the AKZ_T_* ids of include/akz.h
};

// (Re)build plan + carve the arena for images of w x h. Allocates lazily.
int32_t akz_ctx_prepare(akz_ctx* c, int w, int h);

// Stage launchers (enqueue on c->stream).
// akz_scale_space.hip exists once per combination of the three un-vendored arithmetic orders (AKZ_ARITH_* of include/akz.h;
// -DAKZ_ARITH=0..7): akz_arith.hip routes these four calls to the copy c->arith / `arith` names.
int32_t akz_run_scale_space(akz_ctx* c, const void* d_imgs, int fmt, int n);
// h_err_copy (optional, host-visible): receives the sticky overflow flag together with the outputs
int32_t akz_run_keypoints(akz_ctx* c, int n, DevKp* d_kps, akz_descriptor* d_descs, uint32_t cap_per_img,
                          uint32_t* d_n_out, uint32_t* h_err_copy = nullptr);

// stand-alone image ops on device buffers (used by akz_horizontal_filter & co)
int32_t akz_dev_filter1d(int arith, hipStream_t s, const float* in, float* out, int w, int h, const float* d_kernel,
                         int ksize, int vertical);
int32_t akz_dev_deinterleave(hipStream_t s, const float2* in, float* out, size_t n, int component);
int32_t akz_dev_half_size(int arith, hipStream_t s, const float* in, float* out, int w, int h, int n, size_t in_fs,
                          size_t out_fs);

int32_t akz_upload_tables(akz_ctx* c);
size_t akz_ori_table_bytes();
size_t akz_desc_table_bytes();

// Timing of a group of launches on stream `s` (no-ops unless akz_timing_enable is on).  Two kinds of timer:
//   phase timers   (AKZ_T_FED, _SCALE_SPACE, _EXTRACT, _DESCRIBE, _REFINE): a HIP-event bracket on the stream — the wall time
//                  of the phase, waits for the other streams' kernels included;
//   kernel timers  (every other id): each launch made through AKZ_LAUNCH while the timer is open carries its own start /
//                  stop events (hipExtLaunchKernel: the dispatch's own begin and end timestamps — what rocprofv3's
//                  kernel trace reports as the kernel's duration), and the timer accumulates those durations.  A kernel's
//                  roofline is then the same number whether it is read from bench.py's line or from the committed
//                  rocprof kernel statistics of the same command.
void akz_timer_begin(akz_ctx* c, int which, hipStream_t s);
void akz_timer_end(akz_ctx* c, int which, hipStream_t s, uint64_t launches, uint64_t units, uint64_t units2 = 0);
extern thread_local akz_ctx* g_akz_timed_ctx;       // a kernel timer of this context is open on this thread
void akz_timer_launch_events(hipEvent_t* start, hipEvent_t* stop);
// A stage launcher holds one of these: whatever path it returns by (every AKZ_TRY / AKZ_LAUNCH_CHECK between an
// akz_timer_begin and its akz_timer_end is an early return), no kernel timer of the context stays open on the thread —
// an open one would book the next launch of ANY context to this context's timer, or to a freed one.
struct AkzTimerScope {
    akz_ctx* c;
    explicit AkzTimerScope(akz_ctx* ctx) : c(ctx) {}
    ~AkzTimerScope()
    {
        if (g_akz_timed_ctx == c) g_akz_timed_ctx = nullptr;
        c->open_kernel_timer = -1;
    }
    AkzTimerScope(const AkzTimerScope&) = delete;
    AkzTimerScope& operator=(const AkzTimerScope&) = delete;
};
#define AKZ_LAUNCH(kernel, grid, block, shmem, stream, ...)                                              \
    do {                                                                                                 \
        if (g_akz_timed_ctx) {                                                                           \
            hipEvent_t ev0_ = nullptr, ev1_ = nullptr;                                                   \
            akz_timer_launch_events(&ev0_, &ev1_);                                                       \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, ev0_, ev1_, 0, __VA_ARGS__);       \
        } else {                                                                                         \
            hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                         \
        }                                                                                                \
    } while (0)
