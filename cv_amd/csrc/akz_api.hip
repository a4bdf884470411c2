// akz_api.hip — the extern "C" boundary declared in include/akz.h: context lifetime, arena
// carving, host/device entry points, parity taps, timing.  Host-side glue only; the kernels live in
// akz_scale_space.hip / akz_keypoints.hip / hm_match.hip.
#include <math.h>
#include <stdio.h>

#include <new>

#include "akz_ctx.h"

thread_local int g_akz_last_hip = 0;

extern "C" const char* akz_version(void) { return "cv_amd-akz 0.5 (gfx950)"; }
extern "C" uint32_t akz_abi_version(void) { return AKZ_ABI_VERSION; }

extern "C" const char* akz_strerror(int32_t s)
{
    switch (s) {
    case AKZ_OK: return "ok";
    case AKZ_E_INVALID: return "invalid argument or unsupported configuration";
    case AKZ_E_NO_DEVICE: return "no usable HIP device";
    case AKZ_E_OOM: return "out of device memory";
    case AKZ_E_CAPACITY: return "output buffer too small";
    case AKZ_E_HIP: return "HIP runtime error";
    case AKZ_E_TOO_LARGE: return "image or batch larger than the context was created for";
    case AKZ_E_INTERNAL: return "device-side overflow of an internal work list (create the context with a larger max_keypoints / akz_options.max_candidates)";
    case AKZ_E_COMM: return "RCCL call failed or librccl.so.1 could not be loaded (akz_comm_last_error_string())";
    default: return "unknown status";
    }
}
extern "C" int32_t akz_last_hip_error(void) { return g_akz_last_hip; }
extern "C" const char* akz_last_hip_error_string(void) { return hipGetErrorString((hipError_t)g_akz_last_hip); }

extern "C" void akz_config_default(akz_config* c)
{
    if (!c) return;
    // Akaze::default(), akaze/src/lib.rs:169-185
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

// ---- timers ---------------------------------------------------------------------------------
static hipEvent_t timer_event(AkzTimer* t)
{
    if (!t->pool.empty()) {
        hipEvent_t e = t->pool.back();
        t->pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
thread_local akz_ctx* g_akz_timed_ctx = nullptr;
static bool is_phase_timer(int which)
{
    return which == AKZ_T_FED || which == AKZ_T_SCALE_SPACE || which == AKZ_T_EXTRACT || which == AKZ_T_DESCRIBE ||
           which == AKZ_T_REFINE;
}
void akz_timer_begin(akz_ctx* c, int which, hipStream_t s)
{
    if (!c->timing) return;
    AkzTimer* t = &c->timers[which];
    if (!is_phase_timer(which)) {          // kernel timer: the launches bring their own events (AKZ_LAUNCH)
        c->open_kernel_timer = which;
        g_akz_timed_ctx = c;
        return;
    }
    if (!c->timing_phases) return;
    t->cur_start = timer_event(t);
    if (t->cur_start) hipEventRecord(t->cur_start, s);
}
void akz_timer_launch_events(hipEvent_t* start, hipEvent_t* stop)
{
    akz_ctx* c = g_akz_timed_ctx;
    if (!c || c->open_kernel_timer < 0) return;
    AkzTimer* t = &c->timers[c->open_kernel_timer];
    hipEvent_t e0 = timer_event(t), e1 = timer_event(t);
    if (!e0 || !e1) return;
    t->pending.emplace_back(e0, e1);
    *start = e0;
    *stop = e1;
}
void akz_timer_end(akz_ctx* c, int which, hipStream_t s, uint64_t launches, uint64_t units, uint64_t units2)
{
    AkzTimer* t = &c->timers[which];
    if (!c->timing) return;
    if (!is_phase_timer(which)) {
        if (c->open_kernel_timer != which) return;
        c->open_kernel_timer = -1;
        g_akz_timed_ctx = nullptr;
        t->launches += launches;
        t->units += units;
        t->units2 += units2;
        return;
    }
    if (!t->cur_start) return;
    hipEvent_t stop = timer_event(t);
    if (!stop) return;
    hipEventRecord(stop, s);
    t->pending.emplace_back(t->cur_start, stop);
    t->cur_start = nullptr;
    t->launches += launches;
    t->units += units;
    t->units2 += units2;
}
static void timer_resolve(AkzTimer* t)
{
    for (auto& pr : t->pending) {
        float ms = 0.0f;
        hipEventSynchronize(pr.second);
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) t->ms += (double)ms;
        t->pool.push_back(pr.first);
        t->pool.push_back(pr.second);
    }
    t->pending.clear();
}
static void timer_free(AkzTimer* t)
{
    timer_resolve(t);
    for (auto e : t->pool) hipEventDestroy(e);
    t->pool.clear();
}

// ---- context ---------------------------------------------------------------------------------
static int32_t validate_config(const akz_config* cfg)
{
    if (cfg->num_sublevels == 0 || cfg->num_sublevels > 8) return AKZ_E_INVALID;
    if (cfg->max_octave_evolution == 0 || cfg->max_octave_evolution > 8) return AKZ_E_INVALID;
    if (!(cfg->base_scale_offset > 0.0) || !(cfg->derivative_factor > 0.0)) return AKZ_E_INVALID;
    if (cfg->contrast_factor_num_bins == 0 || cfg->contrast_factor_num_bins > 510) return AKZ_E_INVALID;
    if (cfg->descriptor_channels < 1 || cfg->descriptor_channels > 3) return AKZ_E_INVALID;
    if (cfg->descriptor_pattern_size < 1 || cfg->descriptor_pattern_size > 100) return AKZ_E_INVALID;
    // any base_scale_offset the reference accepts (gaussian_blur asserts r > 0, image.rs:384); the fused tile
    // kernel serves radius 4 (sigma in (1.5, 2.0], the default 1.6), other radii the dense filter
    if (2 * akz_gaussian_radius((float)cfg->base_scale_offset) + 1 > kAkzMaxTaps) return AKZ_E_INVALID;
    return AKZ_OK;
}

extern "C" int32_t akz_create_ex(const akz_config* cfg, int32_t device, int32_t max_w, int32_t max_h, int32_t max_batch,
                                 uint32_t max_keypoints, const akz_options* opts, akz_ctx** out)
{
    return akz_guard([&]() -> int32_t {
        if (!cfg || !out || max_w < 3 || max_h < 3 || max_batch < 1 || max_w > 65535 || max_h > 65535) return AKZ_E_INVALID;
        if ((size_t)max_w * (size_t)max_h > kAkzMaxPixels) return AKZ_E_TOO_LARGE;     // 32-bit byte offsets inside a frame (akz_common.h)
        AKZ_TRY(validate_config(cfg));
        akz_options o;
        memset(&o, 0, sizeof(o));
        if (opts) {
            // a caller built against an older (shorter) struct passes its own size; unknown tail = defaults
            if (opts->struct_size < 2 * sizeof(uint32_t) || opts->struct_size > 4096) return AKZ_E_INVALID;
            memcpy(&o, opts, opts->struct_size < sizeof(o) ? opts->struct_size : sizeof(o));
            // a caller built against a NEWER (longer) struct: fields this library does not know must be at their defaults
            // (zero) — anything else would be dropped silently
            for (uint32_t i = (uint32_t)sizeof(o); i < opts->struct_size; ++i)
                if (reinterpret_cast<const unsigned char*>(opts)[i]) return AKZ_E_INVALID;
            if (o.cu_ss > 32 || o.cu_kp > 32) return AKZ_E_INVALID;
            for (uint32_t r : o.reserved)
                if (r) return AKZ_E_INVALID;
            if (o.flags & ~((AKZ_OPT_NO_RESIDENT_LEVELS << 1) - 1u)) return AKZ_E_INVALID;   // unknown switches
            if (o.fed_block > 8 || (o.desc_tile_shift != 0 && (o.desc_tile_shift < 2 || o.desc_tile_shift > 9))) return AKZ_E_INVALID;
            if (o.arith > 7u) return AKZ_E_INVALID;
            // sizes that would only surface as an opaque allocation failure (or wrap in an int) otherwise
            if (o.sup_capacity > (1u << 24) || o.max_candidates > kAkzMaxKeypoints || o.stream_waves > (1u << 24) ||
                o.stream_min_waves > (1u << 24))
                return AKZ_E_INVALID;
        }
        {
            // every per-frame table has kAkzMaxLevels slots per frame: refuse before anything is launched
            AkzPlan probe;
            akz_build_plan(*cfg, max_w, max_h, &probe);
            if ((int)probe.levels.size() > kAkzMaxLevels) return AKZ_E_INVALID;
        }
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
            g_akz_last_hip = (int)e;
            return AKZ_E_NO_DEVICE;  // no CPU fallback, by design
        }
        AKZ_HIP(hipSetDevice(device));
        akz_ctx* c = new (std::nothrow) akz_ctx();
        if (!c) return AKZ_E_OOM;
        c->cfg = *cfg;
        c->device = device;
        {
            int ncu = 0;
            if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) c->n_cu = ncu;
        }
        c->max_w = max_w;
        c->max_h = max_h;
        c->max_batch = max_batch;
        c->max_kp = max_keypoints ? max_keypoints : 16384u;
        if (c->max_kp > kAkzMaxKeypoints) c->max_kp = kAkzMaxKeypoints;
        // per (frame, level) candidate capacity: independent of the keypoint capacity (a level can hold more raw
        // extrema than survive suppression); lists longer than 16384 are sorted through global memory
        c->max_cand = o.max_candidates ? o.max_candidates : c->max_kp;
        if (c->max_cand > kAkzMaxKeypoints) c->max_cand = kAkzMaxKeypoints;
        c->sup_cap = o.sup_capacity ? o.sup_capacity : 4u * c->max_kp;   // all levels together: frames with more take the serial pass
        c->sup_parallel = !(o.flags & AKZ_OPT_SERIAL_SUPPRESSION);
        c->keep_all = (o.flags & AKZ_OPT_KEEP_ALL) != 0;
        c->front_pair = !(o.flags & AKZ_OPT_NO_FRAME_PAIRS);
        c->stream_kernels = !(o.flags & AKZ_OPT_TILE_KERNELS);
        c->det_side_stream = !(o.flags & AKZ_OPT_SERIAL_DET);
        c->fuse_front_fed = !(o.flags & AKZ_OPT_SPLIT_FRONT_FED);
        c->resident_levels = !(o.flags & AKZ_OPT_NO_RESIDENT_LEVELS);
        c->resident_min_frames = (int)o.resident_min_frames;
        c->arith = (int)o.arith;
        if (o.stream_waves) c->det_stream_waves = (int)o.stream_waves;
        if (o.stream_min_waves) c->stream_min_waves = (size_t)o.stream_min_waves;
        c->contrast_fine = !(o.flags & AKZ_OPT_CONTRAST_EXACT);
        c->contrast_force_odd = (o.flags & AKZ_OPT_CONTRAST_FORCE_ODD) != 0;
        c->nsets = (o.flags & AKZ_OPT_NO_PIPELINE) ? 1 : 2;
        if (o.fed_block) c->fed_block = (int)o.fed_block;
        if (o.desc_tile_shift) c->desc_tile_shift = (int)o.desc_tile_shift;
        int32_t st = AKZ_OK;
        // Stream priorities: scale space urgent, keypoint stage filler.  Without gain in round 1 (1982 vs 2029 fps), worth
        // +1 % now that the scale-space kernels are shorter (7 250-7 315 vs 7 205-7 225 frames/s): on unless
        // AKZ_OPT_EQUAL_PRIORITY.
        int prio_lo = 0, prio_hi = 0;
        hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);  // lo = numerically greatest = least urgent
        const bool use_prio = !(o.flags & AKZ_OPT_EQUAL_PRIORITY);
        if (o.cu_ss) {
            if (akz_stream_on_cus(&c->stream, 0, (int)o.cu_ss) != hipSuccess) st = AKZ_E_HIP;
            if (st == AKZ_OK && akz_stream_on_cus(&c->stream_det, 0, (int)o.cu_ss) != hipSuccess) st = AKZ_E_HIP;
        } else {
            if (hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, use_prio ? prio_hi : 0) != hipSuccess) st = AKZ_E_HIP;
            if (st == AKZ_OK && hipStreamCreateWithPriority(&c->stream_det, hipStreamNonBlocking, use_prio ? prio_hi : 0) != hipSuccess)
                st = AKZ_E_HIP;
        }
        if (st == AKZ_OK && o.cu_kp) {
            if (akz_stream_on_cus(&c->stream_kp, 32 - (int)o.cu_kp, (int)o.cu_kp) != hipSuccess) st = AKZ_E_HIP;
        } else if (st == AKZ_OK && hipStreamCreateWithPriority(&c->stream_kp, hipStreamNonBlocking, use_prio ? prio_lo : 0) != hipSuccess)
            st = AKZ_E_HIP;
        if (st == AKZ_OK && hipEventCreateWithFlags(&c->ev_input, hipEventDisableTiming) != hipSuccess) st = AKZ_E_HIP;
        if (st == AKZ_OK && hipEventCreateWithFlags(&c->ev_det_done, hipEventDisableTiming) != hipSuccess) st = AKZ_E_HIP;
        for (int l = 0; l < kAkzMaxLevels && st == AKZ_OK; ++l)
            if (hipEventCreateWithFlags(&c->ev_level[l], hipEventDisableTiming) != hipSuccess) st = AKZ_E_HIP;
        for (int b = 0; b < 2 && st == AKZ_OK; ++b) {
            if (hipEventCreateWithFlags(&c->ev_ss_done[b], hipEventDisableTiming) != hipSuccess) st = AKZ_E_HIP;
            if (st == AKZ_OK && hipEventCreateWithFlags(&c->ev_kp_done[b], hipEventDisableTiming) != hipSuccess) st = AKZ_E_HIP;
        }
        if (st == AKZ_OK) st = akz_ctx_prepare(c, max_w, max_h);
        if (st != AKZ_OK) {
            akz_destroy(c);
            return st;
        }
        *out = c;
        return AKZ_OK;
    });
}

extern "C" int32_t akz_create(const akz_config* cfg, int32_t device, int32_t max_w, int32_t max_h, int32_t max_batch,
                              uint32_t max_keypoints, akz_ctx** out)
{
    return akz_create_ex(cfg, device, max_w, max_h, max_batch, max_keypoints, nullptr, out);
}

extern "C" int32_t akz_destroy(akz_ctx* c)
{
    return akz_guard([&]() -> int32_t {
        if (!c) return AKZ_OK;
        if (g_akz_timed_ctx == c) g_akz_timed_ctx = nullptr;   // (a timer left open by a failed call on this thread)
        hipSetDevice(c->device);
        if (c->stream) hipStreamSynchronize(c->stream);
        if (c->stream_kp) hipStreamSynchronize(c->stream_kp);
        if (c->stream_det) hipStreamSynchronize(c->stream_det);
        for (AkzTimer& t : c->timers) timer_free(&t);
        if (c->arena) hipFree(c->arena);
        if (c->d_color) hipFree(c->d_color);
        if (c->h_in) hipHostFree(c->h_in);
        if (c->h_out) hipHostFree(c->h_out);
        if (c->stream) hipStreamDestroy(c->stream);
        if (c->stream_kp) hipStreamDestroy(c->stream_kp);
        if (c->stream_det) hipStreamDestroy(c->stream_det);
        if (c->ev_input) hipEventDestroy(c->ev_input);
        if (c->ev_det_done) hipEventDestroy(c->ev_det_done);
        for (hipEvent_t e : c->ev_level)
            if (e) hipEventDestroy(e);
        for (int b = 0; b < 2; ++b) {
            if (c->ev_ss_done[b]) hipEventDestroy(c->ev_ss_done[b]);
            if (c->ev_kp_done[b]) hipEventDestroy(c->ev_kp_done[b]);
        }
        delete c;
        return AKZ_OK;
    });
}

namespace {
struct Carver {
    char* base;
    size_t off = 0;
    template <typename T>
    T* take(size_t count)
    {
        off = akz_align_up(off, 256);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += sizeof(T) * count;
        return p;
    }
};
}  // namespace

static void carve_set(akz_ctx* c, AkzSet& S, Carver& cv)
{
    const AkzPlan& P = c->plan;
    const size_t B = (size_t)c->max_batch;
    const size_t P0 = (size_t)c->max_w * c->max_h;  // sized for the largest frame the context accepts
    const int nlev = (int)P.levels.size();
    S.Lt.assign(nlev, nullptr);
    S.Lsm.assign(nlev, nullptr);
    S.Lxy.assign(nlev, nullptr);
    S.Ldet.assign(nlev, nullptr);
    S.Lflow.assign(nlev, nullptr);
    // persistent per-level planes: Lt, Lx, Ly (descriptors), Ldet (extrema, sub-pixel)
    size_t max_level_px = 0;
    for (int i = 0; i < nlev; ++i) {
        size_t px = P.levels[i].pixels() * B;
        S.Lt[i] = cv.take<float>(px);
        S.Lxy[i] = cv.take<float2>(px);
        S.Ldet[i] = cv.take<float>(px);
        if (px > max_level_px) max_level_px = px;
    }
    // transient planes: Lsmooth / Lflow only live while their level is being built
    float* sm_scratch = c->keep_all ? nullptr : cv.take<float>(max_level_px);
    float* fl_scratch = c->keep_all ? nullptr : cv.take<float>(max_level_px);
    for (int i = 0; i < nlev; ++i) {
        size_t px = P.levels[i].pixels() * B;
        if (i == 0) {
            S.Lsm[0] = S.Lt[0];  // lib.rs:201
            S.Lflow[0] = nullptr;
            continue;
        }
        S.Lsm[i] = c->keep_all ? cv.take<float>(px) : sm_scratch;
        S.Lflow[i] = c->keep_all ? cv.take<float>(px) : fl_scratch;
    }
    S.tmp = cv.take<float>(P0 * B);
    S.d_in = cv.take<float>(P0 * B);
    // everything a call zeroes before the contrast pass is one contiguous block: one memset instead of five
    S.d_cmax = cv.take<unsigned long long>(B);
    S.d_npoints = cv.take<uint32_t>(B);
    S.d_ncand = cv.take<uint32_t>(B * kAkzMaxLevels);
    S.d_hist = cv.take<uint32_t>(B * 512);
    S.d_fine = cv.take<uint32_t>(B * 2048);
    S.zero_bytes = (size_t)((char*)(S.d_fine + B * 2048) - (char*)S.d_cmax);
    S.d_cthr = cv.take<double>(B * 512);
    S.d_cflag = cv.take<uint32_t>(B);
    S.d_contrast = cv.take<double>(B);
    S.d_invk = cv.take<float>(B * 8);
    S.d_cand = cv.take<uint2>(B * kAkzMaxLevels * (size_t)c->max_cand);
    S.d_cand_u = cv.take<float>(B * kAkzMaxLevels * (size_t)c->max_cand * 10);   // CandU = 40 bytes
    S.d_cand_nb = cv.take<float>(B * kAkzMaxLevels * (size_t)c->max_cand * 8);
    const size_t K = c->max_kp;
    S.d_cache = cv.take<DevKp>(B * K);
    S.d_ncache = cv.take<uint32_t>(B);
    S.d_lvl_slot = cv.take<uint32_t>((size_t)B * (kAkzMaxLevels + 1));
    S.d_chunk_yr = cv.take<float2>(B * ((K + 63) / 64));
    size_t cand_rows = 0;
    for (const AkzLevel& L : P.levels) cand_rows += (size_t)L.h + 1;
    S.d_cand_rows = cv.take<uint32_t>(B * cand_rows);
    S.d_sup_flag = cv.take<uint32_t>(B);   // directly before d_sup: flags and the reverse-list counters clear in one memset
    S.d_big_flag = cv.take<uint32_t>(B);   // (between the two: the same memset clears it)
    S.d_sup = cv.take<uint32_t>(sup_scratch_words(c->sup_cap, (uint32_t)B));
    S.d_big_act = K > 8192 ? (void*)cv.take<float4>(B * K) : nullptr;   // (kActCap of akz_keypoints.hip)
    S.d_kp_a = cv.take<DevKp>(B * K);
    S.d_n_a = cv.take<uint32_t>(B);
    S.d_kp_b = cv.take<DevKp>(B * K);
    S.d_flag_b = cv.take<uint32_t>(B * K);
    S.d_kp_c = cv.take<DevKp>(B * K);
    S.d_n_c = cv.take<uint32_t>(B);
    S.d_kp_d = cv.take<DevKp>(B * K);
    S.d_n_d = cv.take<uint32_t>(B);
    S.d_desc_tmp = cv.take<akz_descriptor>(B * K);
    S.d_flag_d = cv.take<uint32_t>(B * K);
    S.d_perm = cv.take<uint32_t>(B * K);
    S.d_kp_out = cv.take<DevKp>(B * K);
    S.d_desc_out = cv.take<akz_descriptor>(B * K);
    S.d_n_out = cv.take<uint32_t>(B);
    auto np2_of = [](uint32_t v) {
        size_t p = 1;
        while (p < v) p <<= 1;
        return p;
    };
    S.d_keys_kp = c->max_kp > kAkzLdsSortKeys ? cv.take<unsigned long long>(B * np2_of(c->max_kp)) : nullptr;
    S.d_keys_cand = c->max_cand > kAkzLdsSortKeys ? cv.take<unsigned long long>(B * kAkzMaxLevels * np2_of(c->max_cand)) : nullptr;
}

static void carve(akz_ctx* c, char* base, size_t* total)
{
    Carver cv{base};
    for (int b = 0; b < c->nsets; ++b) carve_set(c, c->sets[b], cv);
    c->d_err = cv.take<uint32_t>(4);
    c->d_ori = cv.take<char>(akz_ori_table_bytes());
    c->d_desc = cv.take<char>(akz_desc_table_bytes());
    c->d_taps = cv.take<float>(kAkzMaxTaps + 1);
    *total = akz_align_up(cv.off, 256);
}

static int32_t sync_all(akz_ctx* c);

int32_t akz_ctx_prepare(akz_ctx* c, int w, int h)
{
    if (w > c->max_w || h > c->max_h) return AKZ_E_TOO_LARGE;
    if (w < 3 || h < 3) return AKZ_E_INVALID;
    if (c->arena && c->cur_w == w && c->cur_h == h) return AKZ_OK;
    AKZ_HIP(hipSetDevice(c->device));
    // The arena is sized once for (max_w, max_h); smaller frames re-carve the same arena.
    if (!c->arena) {
        akz_build_plan(c->cfg, c->max_w, c->max_h, &c->plan);
        size_t total = 0;
        carve(c, nullptr, &total);
        AKZ_HIP(hipMalloc(&c->arena, total));
        c->arena_bytes = total;
    } else {
        AKZ_TRY(sync_all(c));
    }
    akz_build_plan(c->cfg, w, h, &c->plan);
    size_t total = 0;
    carve(c, (char*)c->arena, &total);
    if (total > c->arena_bytes) return AKZ_E_INTERNAL;
    c->cur_w = w;
    c->cur_h = h;
    AKZ_HIP(hipMemsetAsync(c->d_err, 0, sizeof(uint32_t) * 4, c->stream));
    AKZ_TRY(akz_upload_tables(c));
    return AKZ_OK;
}

static int32_t check_device_err(akz_ctx* c)
{
    uint32_t err = 0;
    AKZ_TRY(sync_all(c));
    AKZ_HIP(hipMemcpy(&err, c->d_err, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (err & ~4u) {
        AKZ_HIP(hipMemsetAsync(c->d_err, 0, sizeof(uint32_t), c->stream));
        return AKZ_E_INTERNAL;
    }
    return AKZ_OK;
}

// pinned, device-visible host block of at least `bytes` (grown on demand, never shrunk)
constexpr size_t kAkzHostStageMax = (size_t)96 << 20;
static int32_t host_block(void** p, size_t* have, size_t bytes)
{
    if (*p && *have >= bytes) return AKZ_OK;
    if (*p) {
        AKZ_HIP(hipHostFree(*p));
        *p = nullptr;
        *have = 0;
    }
    const size_t want = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    if (hipHostMalloc(p, want, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();     // not sticky: the caller falls back to plain copies
        *p = nullptr;
        return AKZ_E_OOM;
    }
    *have = want;
    return AKZ_OK;
}

// Start of a batch call: pick the buffer set and make the scale-space stream wait until the keypoint
// stage that last used this set has finished reading its pyramid.
static int32_t begin_call(akz_ctx* c)
{
    c->cur = (int)(c->calls % (uint64_t)c->nsets);
    c->calls++;
    if (c->kp_pending[c->cur]) AKZ_HIP(hipStreamWaitEvent(c->stream, c->ev_kp_done[c->cur], 0));
    return AKZ_OK;
}
static int32_t sync_all(akz_ctx* c)
{
    AKZ_HIP(hipStreamSynchronize(c->stream));
    AKZ_HIP(hipStreamSynchronize(c->stream_kp));
    // the determinant side stream joins the main stream at the end of a call's level loop only: a call that left that
    // loop early (an error return) may still have kernels there that write this buffer set's candidate lists
    if (c->stream_det) AKZ_HIP(hipStreamSynchronize(c->stream_det));
    c->kp_pending[0] = c->kp_pending[1] = false;
    return AKZ_OK;
}

static int32_t wait_for(akz_ctx* c, void* stream_to_wait)
{
    if (!stream_to_wait) return AKZ_OK;
    AKZ_HIP(hipEventRecord(c->ev_input, akz_wait_stream(stream_to_wait)));
    AKZ_HIP(hipStreamWaitEvent(c->stream, c->ev_input, 0));
    return AKZ_OK;
}

// The stream on which a call's outputs complete (the keypoint stream).
extern "C" void* akz_stream(akz_ctx* c) { return c ? (void*)c->stream_kp : nullptr; }
extern "C" int32_t akz_sync(akz_ctx* c)
{
    return akz_guard([&]() -> int32_t {
        if (!c) return AKZ_E_INVALID;
        AKZ_HIP(hipSetDevice(c->device));
        return check_device_err(c);  // synchronises both streams
    });
}

extern "C" int32_t akz_scale_space_device(akz_ctx* c, const void* d_imgs, int32_t fmt, int32_t n, int32_t w, int32_t h,
                                          void* stream_to_wait)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !d_imgs || n < 1 || (fmt < 0 || fmt > 2)) return AKZ_E_INVALID;
        if (n > c->max_batch) return AKZ_E_TOO_LARGE;
        AKZ_HIP(hipSetDevice(c->device));
        AKZ_TRY(akz_ctx_prepare(c, w, h));
        AKZ_TRY(begin_call(c));
        AKZ_TRY(wait_for(c, stream_to_wait));
    c->cur_n = n;
        AKZ_TRY(akz_run_scale_space(c, d_imgs, fmt, n));
        // completion is observable on akz_stream() like every other call
        AKZ_HIP(hipEventRecord(c->ev_ss_done[c->cur], c->stream));
        AKZ_HIP(hipStreamWaitEvent(c->stream_kp, c->ev_ss_done[c->cur], 0));
        AKZ_HIP(hipEventRecord(c->ev_kp_done[c->cur], c->stream_kp));
        c->kp_pending[c->cur] = true;
        return AKZ_OK;
    });
}

extern "C" int32_t akz_extract_batch_device(akz_ctx* c, const void* d_imgs, int32_t fmt, int32_t n, int32_t w,
                                            int32_t h, void* d_kps, void* d_descs, uint32_t cap_per_img,
                                            void* d_n_out, void* stream_to_wait)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !d_imgs || !d_kps || !d_descs || !d_n_out || n < 1 || (fmt < 0 || fmt > 2) || cap_per_img == 0)
            return AKZ_E_INVALID;
        if (n > c->max_batch) return AKZ_E_TOO_LARGE;
        AKZ_HIP(hipSetDevice(c->device));
        AKZ_TRY(akz_ctx_prepare(c, w, h));
        AKZ_TRY(begin_call(c));
        AKZ_TRY(wait_for(c, stream_to_wait));
    c->cur_n = n;
        akz_timer_begin(c, AKZ_T_EXTRACT, c->stream);
        AKZ_TRY(akz_run_scale_space(c, d_imgs, fmt, n));
        AKZ_TRY(akz_run_keypoints(c, n, (DevKp*)d_kps, (akz_descriptor*)d_descs, cap_per_img, (uint32_t*)d_n_out));
        akz_timer_end(c, AKZ_T_EXTRACT, c->stream_kp, 0, (uint64_t)n);   // closes where the outputs complete
        return AKZ_OK;
    });
}

// second half of a host call: the frames are in S().d_in (fmt = their pixel format); run, and hand the outputs back
static int32_t host_extract_finish(akz_ctx* c, int32_t fmt, int32_t n, akz_keypoint* kps, akz_descriptor* descs, uint32_t cap_per_img,
                                   uint32_t* n_out)
{
    c->cur_n = n;
    // ---- outputs: the last kernel writes them where the host can read them ----
    const size_t K = c->max_kp;
    const size_t head = 64 + (((size_t)n * sizeof(uint32_t) + 63) & ~(size_t)63);
    // (the descriptor block starts on a 64-byte boundary: the compaction kernel writes it with 16-byte stores and
    // n * K * sizeof(DevKp) — 28-byte records — is a multiple of 16 only by accident)
    const size_t kp_bytes = akz_align_up((size_t)n * K * sizeof(DevKp), 64);
    const size_t out_bytes = head + kp_bytes + (size_t)n * K * sizeof(akz_descriptor);
    const bool stage_out = out_bytes <= kAkzHostStageMax && host_block(&c->h_out, &c->h_out_bytes, out_bytes) == AKZ_OK;
    akz_timer_begin(c, AKZ_T_EXTRACT, c->stream);
    AKZ_TRY(akz_run_scale_space(c, c->S().d_in, fmt, n));
    int32_t status = AKZ_OK;
    if (stage_out) {
        uint32_t* h_err = (uint32_t*)c->h_out;
        uint32_t* h_n = (uint32_t*)((char*)c->h_out + 64);
        DevKp* h_kp = (DevKp*)((char*)c->h_out + head);
        akz_descriptor* h_desc = (akz_descriptor*)((char*)c->h_out + head + kp_bytes);
        AKZ_TRY(akz_run_keypoints(c, n, h_kp, h_desc, c->max_kp, h_n, h_err));
        akz_timer_end(c, AKZ_T_EXTRACT, c->stream_kp, 0, (uint64_t)n);   // closes where the outputs complete
        // the keypoint stream waited for the scale-space stream: its end is the end of the call
        AKZ_HIP(hipStreamSynchronize(c->stream_kp));
        c->kp_pending[c->cur] = false;
        if (*h_err & ~4u) {
            AKZ_HIP(hipMemsetAsync(c->d_err, 0, sizeof(uint32_t), c->stream));
            return AKZ_E_INTERNAL;
        }
        for (int i = 0; i < n; ++i) {
            const uint32_t cnt = h_n[i];
            n_out[i] = cnt;
            if (cnt > c->max_kp) return AKZ_E_INTERNAL;
            if (cnt > cap_per_img) status = AKZ_E_CAPACITY;
            const uint32_t m = cnt < cap_per_img ? cnt : cap_per_img;
            if (m) {
                memcpy(kps + (size_t)i * cap_per_img, h_kp + (size_t)i * K, sizeof(akz_keypoint) * m);
                memcpy(descs + (size_t)i * cap_per_img, h_desc + (size_t)i * K, sizeof(akz_descriptor) * m);
            }
        }
        return status;
    }
    AKZ_TRY(akz_run_keypoints(c, n, c->S().d_kp_out, c->S().d_desc_out, c->max_kp, c->S().d_n_out));
    akz_timer_end(c, AKZ_T_EXTRACT, c->stream_kp, 0, (uint64_t)n);   // closes where the outputs complete
    std::vector<uint32_t> cnt(n);
    AKZ_TRY(check_device_err(c));  // synchronises both streams
    AKZ_HIP(hipMemcpy(cnt.data(), c->S().d_n_out, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {
        n_out[i] = cnt[i];
        if (cnt[i] > c->max_kp) return AKZ_E_INTERNAL;
        if (cnt[i] > cap_per_img) status = AKZ_E_CAPACITY;
        uint32_t m = cnt[i] < cap_per_img ? cnt[i] : cap_per_img;
        if (m) {
            AKZ_HIP(hipMemcpy(kps + (size_t)i * cap_per_img, c->S().d_kp_out + (size_t)i * c->max_kp, sizeof(akz_keypoint) * m,
                              hipMemcpyDeviceToHost));
            AKZ_HIP(hipMemcpy(descs + (size_t)i * cap_per_img, c->S().d_desc_out + (size_t)i * c->max_kp,
                              sizeof(akz_descriptor) * m, hipMemcpyDeviceToHost));
        }
    }
    return status;
}

extern "C" int32_t akz_extract_batch(akz_ctx* c, const void* const* imgs, int32_t fmt, int32_t n, int32_t w, int32_t h,
                                     int32_t stride, akz_keypoint* kps, akz_descriptor* descs, uint32_t cap_per_img,
                                     uint32_t* n_out)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !imgs || !n_out || n < 1 || (fmt < 0 || fmt > 2) || stride < w) return AKZ_E_INVALID;
        if (cap_per_img && (!kps || !descs)) return AKZ_E_INVALID;
        if (n > c->max_batch) return AKZ_E_TOO_LARGE;
        AKZ_HIP(hipSetDevice(c->device));
        AKZ_TRY(akz_ctx_prepare(c, w, h));
        AKZ_TRY(begin_call(c));
        const size_t esz = fmt == AKZ_FMT_U8 ? 1 : (fmt == AKZ_FMT_U16 ? 2 : 4);
        const size_t P0 = (size_t)w * h;
        for (int i = 0; i < n; ++i)
            if (!imgs[i]) return AKZ_E_INVALID;
        // ---- input: rows gathered into the pinned block, one DMA ----
        const size_t in_bytes = (size_t)n * P0 * esz;
        // (a host that cannot pin the block takes the plain copies: staging is an optimisation, never a requirement)
        const bool stage_in = in_bytes <= kAkzHostStageMax && host_block(&c->h_in, &c->h_in_bytes, in_bytes) == AKZ_OK;
        if (stage_in) {
            // row blocks of ~512 KB: the DMA of a block runs while the next one is gathered
            const int rows_per = (int)std::max<size_t>(1, ((size_t)512 << 10) / ((size_t)w * esz));
            for (int i = 0; i < n; ++i) {
                char* dst = (char*)c->h_in + (size_t)i * P0 * esz;
                const char* src = (const char*)imgs[i];
                for (int y0 = 0; y0 < h; y0 += rows_per) {
                    const int y1 = std::min(h, y0 + rows_per);
                    if (stride == w) memcpy(dst + (size_t)y0 * w * esz, src + (size_t)y0 * w * esz, (size_t)(y1 - y0) * w * esz);
                    else
                        for (int y = y0; y < y1; ++y)
                            memcpy(dst + (size_t)y * w * esz, src + (size_t)y * stride * esz, (size_t)w * esz);
                    AKZ_HIP(hipMemcpyAsync((char*)c->S().d_in + (size_t)i * P0 * esz + (size_t)y0 * w * esz, dst + (size_t)y0 * w * esz,
                                           (size_t)(y1 - y0) * w * esz, hipMemcpyHostToDevice, c->stream));
                }
            }
        } else {
            for (int i = 0; i < n; ++i)
                AKZ_HIP(hipMemcpy2DAsync((char*)c->S().d_in + (size_t)i * P0 * esz, (size_t)w * esz, imgs[i], (size_t)stride * esz,
                                         (size_t)w * esz, (size_t)h, hipMemcpyHostToDevice, c->stream));
        }
        return host_extract_finish(c, fmt, n, kps, descs, cap_per_img, n_out);
    });
}

// Akaze::extract on a colour DynamicImage (ImageRgb8 / Rgba8 / Rgb16 / Rgba16 / Rgb32F / Rgba32F): akaze/src/image.rs:45-46
// grayscale() first, then the matching gray arm.  pixels: h rows of `stride` ELEMENTS, `channels` (3 or 4) interleaved
// samples per pixel; alpha is ignored, as grayscale() drops it.
int32_t akz_color_to_input(akz_ctx* c, const void* pixels, int32_t fmt, int32_t channels, int32_t w, int32_t h, int32_t stride);
extern "C" int32_t akz_extract_color(akz_ctx* c, const void* pixels, int32_t fmt, int32_t channels, int32_t w, int32_t h,
                                     int32_t stride, akz_keypoint* kps, akz_descriptor* descs, uint32_t cap, uint32_t* n_out)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !pixels || !n_out || (fmt < 0 || fmt > 2) || (channels != 3 && channels != 4) || stride < w * channels)
            return AKZ_E_INVALID;
        if (cap && (!kps || !descs)) return AKZ_E_INVALID;
        AKZ_HIP(hipSetDevice(c->device));
        AKZ_TRY(akz_ctx_prepare(c, w, h));
        AKZ_TRY(begin_call(c));
        AKZ_TRY(akz_color_to_input(c, pixels, fmt, channels, w, h, stride));
        return host_extract_finish(c, fmt, 1, kps, descs, cap, n_out);
    });
}

extern "C" int32_t akz_extract_gray_u8(akz_ctx* c, const uint8_t* img, int32_t w, int32_t h, int32_t stride,
                                       akz_keypoint* kps, akz_descriptor* descs, uint32_t cap, uint32_t* n_out)
{
    return akz_guard([&]() -> int32_t {
        const void* p = img;
        return akz_extract_batch(c, &p, 0, 1, w, h, stride, kps, descs, cap, n_out);
    });
}
extern "C" int32_t akz_extract_gray_u16(akz_ctx* c, const uint16_t* img, int32_t w, int32_t h, int32_t stride,
                                        akz_keypoint* kps, akz_descriptor* descs, uint32_t cap, uint32_t* n_out)
{
    return akz_guard([&]() -> int32_t {
        const void* p = img;
        return akz_extract_batch(c, &p, AKZ_FMT_U16, 1, w, h, stride, kps, descs, cap, n_out);
    });
}
extern "C" int32_t akz_extract_gray_f32(akz_ctx* c, const float* img, int32_t w, int32_t h, int32_t stride,
                                        akz_keypoint* kps, akz_descriptor* descs, uint32_t cap, uint32_t* n_out)
{
    return akz_guard([&]() -> int32_t {
        const void* p = img;
        return akz_extract_batch(c, &p, 1, 1, w, h, stride, kps, descs, cap, n_out);
    });
}

// ---- introspection -----------------------------------------------------------------------------
static int32_t plan_for(akz_ctx* c, int w, int h, AkzPlan* tmp, const AkzPlan** out)
{
    if (!c) return AKZ_E_INVALID;
    if (w == c->cur_w && h == c->cur_h) {
        *out = &c->plan;
        return AKZ_OK;
    }
    if (w < 1 || h < 1) return AKZ_E_INVALID;
    akz_build_plan(c->cfg, w, h, tmp);
    *out = tmp;
    return AKZ_OK;
}
extern "C" int32_t akz_num_levels(akz_ctx* c, int32_t w, int32_t h, int32_t* n_levels)
{
    return akz_guard([&]() -> int32_t {
        AkzPlan tmp;
        const AkzPlan* P;
        if (!n_levels) return AKZ_E_INVALID;
        AKZ_TRY(plan_for(c, w, h, &tmp, &P));
        *n_levels = (int32_t)P->levels.size();
        return AKZ_OK;
    });
}
extern "C" int32_t akz_level(akz_ctx* c, int32_t w, int32_t h, int32_t level, akz_level_info* out)
{
    return akz_guard([&]() -> int32_t {
        AkzPlan tmp;
        const AkzPlan* P;
        if (!out) return AKZ_E_INVALID;
        AKZ_TRY(plan_for(c, w, h, &tmp, &P));
        if (level < 0 || level >= (int)P->levels.size()) return AKZ_E_INVALID;
        const AkzLevel& L = P->levels[level];
        out->width = L.w;
        out->height = L.h;
        out->octave = L.octave;
        out->sublevel = L.sublevel;
        out->esigma = L.esigma;
        out->etime = L.etime;
        out->n_fed_steps = (uint32_t)L.tau.size();
        out->deriv_sigma = L.deriv_sigma;
        return AKZ_OK;
    });
}
extern "C" int32_t akz_fed_tau(akz_ctx* c, int32_t w, int32_t h, int32_t level, double* tau, uint32_t cap,
                               uint32_t* n_out)
{
    return akz_guard([&]() -> int32_t {
        AkzPlan tmp;
        const AkzPlan* P;
        if (!n_out) return AKZ_E_INVALID;
        AKZ_TRY(plan_for(c, w, h, &tmp, &P));
        if (level < 0 || level >= (int)P->levels.size()) return AKZ_E_INVALID;
        const auto& t = P->levels[level].tau;
        *n_out = (uint32_t)t.size();
        if (t.size() > cap) return AKZ_E_CAPACITY;
        for (size_t i = 0; i < t.size(); ++i) tau[i] = t[i];
        return AKZ_OK;
    });
}

extern "C" int32_t akz_debug_get_level(akz_ctx* c, int32_t img, int32_t level, int32_t which, float* out)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !out || img < 0 || img >= c->cur_n) return AKZ_E_INVALID;
        if (level < 0 || level >= (int)c->plan.levels.size()) return AKZ_E_INVALID;
        const float* src = nullptr;
        int comp = -1;
        switch (which) {
        case AKZ_BUF_LT: src = c->S().Lt[level]; break;
        case AKZ_BUF_LSMOOTH: src = c->S().Lsm[level]; break;
        case AKZ_BUF_LX: comp = 0; break;
        case AKZ_BUF_LY: comp = 1; break;
        case AKZ_BUF_LDET: src = c->S().Ldet[level]; break;
        case AKZ_BUF_LFLOW: src = c->S().Lflow[level]; break;
        default: return AKZ_E_INVALID;
        }
        if (!src && comp < 0) return AKZ_E_INVALID;
        // Lsmooth / Lflow are transient scratch unless the context was created with AKZ_KEEP_ALL=1
        if (!c->keep_all && level > 0 && (which == AKZ_BUF_LSMOOTH || which == AKZ_BUF_LFLOW)) return AKZ_E_INVALID;
        if (!c->keep_all && which == AKZ_BUF_LDET) return AKZ_E_INVALID;   // Ldet planes exist only for the parity taps
        AKZ_HIP(hipSetDevice(c->device));
        AKZ_TRY(sync_all(c));
        size_t px = c->plan.levels[level].pixels();
        if (comp >= 0) {  // Lx / Ly live interleaved; split one component into the (idle) FED scratch plane
            AKZ_TRY(akz_dev_deinterleave(c->stream, c->S().Lxy[level] + (size_t)img * px, c->S().tmp, px, comp));
            AKZ_HIP(hipStreamSynchronize(c->stream));
            src = c->S().tmp;
            AKZ_HIP(hipMemcpy(out, src, sizeof(float) * px, hipMemcpyDeviceToHost));
            return AKZ_OK;
        }
        AKZ_HIP(hipMemcpy(out, src + (size_t)img * px, sizeof(float) * px, hipMemcpyDeviceToHost));
        return AKZ_OK;
    });
}
extern "C" int32_t akz_debug_get_contrast(akz_ctx* c, int32_t img, double* out)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !out || img < 0 || img >= c->cur_n) return AKZ_E_INVALID;
        AKZ_HIP(hipSetDevice(c->device));
        AKZ_TRY(sync_all(c));
        AKZ_HIP(hipMemcpy(out, c->S().d_contrast + img, sizeof(double), hipMemcpyDeviceToHost));
        return AKZ_OK;
    });
}
extern "C" int32_t akz_debug_get_keypoints(akz_ctx* c, int32_t img, int32_t stage, akz_keypoint* out, uint32_t cap,
                                           uint32_t* n_out)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !n_out || img < 0 || img >= c->cur_n) return AKZ_E_INVALID;
        const DevKp* src;
        const uint32_t* cnt;
        switch (stage) {
        case 0: src = c->S().d_kp_a; cnt = c->S().d_n_a; break;
        case 1: src = c->S().d_kp_c; cnt = c->S().d_n_c; break;
        case 2: src = c->S().d_kp_d; cnt = c->S().d_n_d; break;
        default: return AKZ_E_INVALID;
        }
        AKZ_HIP(hipSetDevice(c->device));
        AKZ_TRY(sync_all(c));
        uint32_t n = 0;
        AKZ_HIP(hipMemcpy(&n, cnt + img, sizeof(uint32_t), hipMemcpyDeviceToHost));
        *n_out = n;
        if (n > c->max_kp) return AKZ_E_INTERNAL;
        uint32_t m = n < cap ? n : cap;
        if (m && !out) return AKZ_E_INVALID;
        if (m) AKZ_HIP(hipMemcpy(out, src + (size_t)img * c->max_kp, sizeof(akz_keypoint) * m, hipMemcpyDeviceToHost));
        // before refinement the angle field carries the candidate index (internal bookkeeping): the reference's
        // keypoints have angle 0 at that stage (scale_space_extrema.rs:111)
        if (stage == 0)
            for (uint32_t i = 0; i < m; ++i) out[i].angle = 0.0f;
        return n > cap ? AKZ_E_CAPACITY : AKZ_OK;
    });
}

extern "C" int32_t akz_last_overflow(akz_ctx* c, akz_overflow_info* per_frame, uint32_t cap, uint32_t* n_frames)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !n_frames) return AKZ_E_INVALID;
        AKZ_HIP(hipSetDevice(c->device));
        AKZ_TRY(sync_all(c));
        const int n = c->cur_n;
        *n_frames = (uint32_t)n;
        if ((uint32_t)n > cap) return AKZ_E_CAPACITY;
        if (n && !per_frame) return AKZ_E_INVALID;
        std::vector<uint32_t> ncand((size_t)n * kAkzMaxLevels), ncache(n);
        if (n) {
            AKZ_HIP(hipMemcpy(ncand.data(), c->S().d_ncand, sizeof(uint32_t) * ncand.size(), hipMemcpyDeviceToHost));
            AKZ_HIP(hipMemcpy(ncache.data(), c->S().d_ncache, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
        }
        for (int f = 0; f < n; ++f) {
            uint32_t mx = 0;
            for (int l = 0; l < kAkzMaxLevels; ++l) mx = ncand[(size_t)f * kAkzMaxLevels + l] > mx ? ncand[(size_t)f * kAkzMaxLevels + l] : mx;
            per_frame[f].needed_candidates = mx;               // the counters keep counting past the capacity
            per_frame[f].candidate_capacity = c->max_cand;
            per_frame[f].keypoint_capacity = c->max_kp;
            per_frame[f].flags = (mx > c->max_cand ? 1u : 0u) | (ncache[f] > c->max_kp ? 2u : 0u);
        }
        return AKZ_OK;
    });
}

// ---- akaze::image stand-alone ops ----------------------------------------------------------------
extern "C" int32_t akz_gaussian_kernel(float r, uint32_t kernel_size, float* out)
{
    return akz_guard([&]() -> int32_t {
        if (!out || kernel_size % 2 != 1 || !(r > 0.0f)) return AKZ_E_INVALID;  // image.rs:361 asserts odd
        akz_host_gaussian_kernel(r, (int)kernel_size, out);
        return AKZ_OK;
    });
}

static int32_t filter_host(akz_ctx* c, const float* img, int w, int h, const float* kernel, uint32_t ksize,
                           float* out, int vertical)
{
    if (!c || !img || !kernel || !out || w < 1 || h < 1 || ksize % 2 != 1 || ksize > 4095) return AKZ_E_INVALID;
    AKZ_HIP(hipSetDevice(c->device));
    float *d_in = nullptr, *d_out = nullptr, *d_k = nullptr;
    size_t px = (size_t)w * h;
    int32_t st = AKZ_OK;
    hipError_t e;
    if ((e = hipMalloc(&d_in, px * 4)) != hipSuccess || (e = hipMalloc(&d_out, px * 4)) != hipSuccess ||
        (e = hipMalloc(&d_k, ksize * 4)) != hipSuccess) {
        g_akz_last_hip = (int)e;
        st = AKZ_E_OOM;
    }
    if (st == AKZ_OK) {
        hipMemcpyAsync(d_in, img, px * 4, hipMemcpyHostToDevice, c->stream);
        hipMemcpyAsync(d_k, kernel, ksize * 4, hipMemcpyHostToDevice, c->stream);
        st = akz_dev_filter1d(c->arith, c->stream, d_in, d_out, w, h, d_k, (int)ksize, vertical);
        if (st == AKZ_OK && hipMemcpyAsync(out, d_out, px * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess) st = AKZ_E_HIP;
        if (hipStreamSynchronize(c->stream) != hipSuccess) st = AKZ_E_HIP;
    }
    hipFree(d_in);
    hipFree(d_out);
    hipFree(d_k);
    return st;
}
extern "C" int32_t akz_horizontal_filter(akz_ctx* c, const float* img, int32_t w, int32_t h, const float* kernel,
                                         uint32_t ksize, float* out)
{
    return akz_guard([&]() -> int32_t {
        return filter_host(c, img, w, h, kernel, ksize, out, 0);
    });
}
extern "C" int32_t akz_vertical_filter(akz_ctx* c, const float* img, int32_t w, int32_t h, const float* kernel,
                                       uint32_t ksize, float* out)
{
    return akz_guard([&]() -> int32_t {
        return filter_host(c, img, w, h, kernel, ksize, out, 1);
    });
}
extern "C" int32_t akz_half_size(akz_ctx* c, const float* img, int32_t w, int32_t h, float* out)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !img || !out || w < 2 || h < 2) return AKZ_E_INVALID;
        AKZ_HIP(hipSetDevice(c->device));
        float *d_in = nullptr, *d_out = nullptr;
        size_t px = (size_t)w * h, opx = (size_t)(w / 2) * (h / 2);
        int32_t st = AKZ_OK;
        if (hipMalloc(&d_in, px * 4) != hipSuccess || hipMalloc(&d_out, opx * 4) != hipSuccess) st = AKZ_E_OOM;
        if (st == AKZ_OK) {
            hipMemcpyAsync(d_in, img, px * 4, hipMemcpyHostToDevice, c->stream);
            st = akz_dev_half_size(c->arith, c->stream, d_in, d_out, w, h, 1, px, opx);
            if (st == AKZ_OK && hipMemcpyAsync(out, d_out, opx * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess) st = AKZ_E_HIP;
            if (hipStreamSynchronize(c->stream) != hipSuccess) st = AKZ_E_HIP;
        }
        hipFree(d_in);
        hipFree(d_out);
        return st;
    });
}

// ---- timing -----------------------------------------------------------------------------------
extern "C" int32_t akz_timing_enable(akz_ctx* c, int32_t on)
{
    return akz_guard([&]() -> int32_t {
        if (!c) return AKZ_E_INVALID;
        c->timing = on != 0;
        c->timing_phases = on == 1;         // 2: kernel timers only (cheap enough to stay on inside a timed region)
        c->open_kernel_timer = -1;
        g_akz_timed_ctx = nullptr;
        return AKZ_OK;
    });
}
extern "C" int32_t akz_timing_reset(akz_ctx* c)
{
    return akz_guard([&]() -> int32_t {
        if (!c) return AKZ_E_INVALID;
        for (AkzTimer& t : c->timers) {
            timer_resolve(&t);
            t.ms = 0.0;
            t.launches = t.units = t.units2 = 0;
        }
        return AKZ_OK;
    });
}
extern "C" int32_t akz_timing_get(akz_ctx* c, int32_t which, double* ms, uint64_t* launches, uint64_t* units)
{
    return akz_guard([&]() -> int32_t {
        if (!c || which < 0 || which >= AKZ_T_COUNT) return AKZ_E_INVALID;
        AkzTimer* t = &c->timers[which == AKZ_T_FED_PASS ? AKZ_T_FED : which];
        AKZ_HIP(hipSetDevice(c->device));
        timer_resolve(t);
        if (ms) *ms = t->ms;
        if (launches) *launches = t->launches;
        if (units) *units = which == AKZ_T_FED_PASS ? t->units2 : t->units;
        return AKZ_OK;
    });
}
