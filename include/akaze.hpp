// akaze.hpp — C++ host-side mirror of the reference's Rust API for the hot path, over the C ABI
// of akz.h.  The reference's own toolchain (cargo/rustc) is not available in the build image, so the
// host side above the C ABI is C++ (and ctypes for the tests); the Rust shim a maintainer would add is
// in INTEGRATION.md / rust/akaze-mi355x/.
//
// Same names, argument meaning and error behaviour as the reference (paths relative to rust-cv/cv):
//   akaze::Akaze            11 public fields, Default, new_/sparse/dense          akaze/src/lib.rs:109-185
//   Akaze::extract / extract_from_gray_float_image                                akaze/src/lib.rs:295, 309
//   akaze::KeyPoint         point, response, size, octave, class_id, angle        akaze/src/lib.rs:69-93
//   bitarray::BitArray<64>  64 descriptor bytes, bit i at bytes[i>>3] bit (i&7)   akaze/src/descriptors.rs:197
//   space::LinearKnn{metric: Hamming, iter}.knn(query, 2) -> [Neighbor; 2]        akaze/tests/estimate_pose.rs:82-88
//   matching / symmetric_matching                      tutorial-code/chapter5-geometric-verification/src/main.rs:154-200
//   match_descriptors (Lowe ratio)                                                akaze/tests/estimate_pose.rs:78-97
//
// Error behaviour: Akaze::extract is infallible in the reference (keypoints whose descriptor samples
// leave the image are silently dropped).  Here device problems (no GPU, out of memory, list overflow)
// additionally throw akaze::Error — there is no CPU fallback to fall back to.
#pragma once

#include <array>
#include <cstdint>
#include <cstring>
#include <limits>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "akz.h"

namespace akaze {

struct Error : std::runtime_error {
    int32_t status;
    Error(int32_t s, const std::string& what) : std::runtime_error(what + ": " + akz_strerror(s)), status(s) {}
};
inline void check(int32_t s, const char* what)
{
    if (s != AKZ_OK) throw Error(s, what);
}
// The library this translation unit is linked to must export the ABI this header was written against (AKZ_ABI_VERSION).
inline void require_abi()
{
    static const bool ok = akz_abi_version() == AKZ_ABI_VERSION;
    if (!ok) throw std::runtime_error("libakz exports ABI " + std::to_string(akz_abi_version()) + ", akaze.hpp was written against " +
                                      std::to_string(AKZ_ABI_VERSION));
}

// akaze::KeyPoint (lib.rs:69-93)
struct KeyPoint {
    std::pair<float, float> point;
    float response;
    float size;
    std::size_t octave;
    std::size_t class_id;
    float angle;
    // cv_core::ImagePoint::image_point (lib.rs:95-99)
    std::pair<double, double> image_point() const { return {(double)point.first, (double)point.second}; }
};

using BitArray64 = std::array<uint8_t, 64>;  // bitarray::BitArray<64>

// A Luma8 image view (what DynamicImage::ImageLuma8 hands to GrayFloatImage::from_dynamic, image.rs:47-56).
struct GrayImageU8 {
    const uint8_t* data;
    int width, height, stride;
};
// A GrayFloatImage view (image.rs:36): f32 pixels in [0,1], row-major.
struct GrayFloatImage {
    const float* data;
    int width, height, stride;
};

// A colour image view (DynamicImage::ImageRgb8 / Rgba8: what image.rs:45-46 sends through grayscale() first).
struct ColorImageU8 {
    const uint8_t* data;
    int width, height, stride;   // stride in bytes
    int channels;                // 3 (RGB) or 4 (RGBA)
};

namespace detail {
// Device contexts are not part of akaze::Akaze (a Copy struct of eleven fields whose methods take &self and keep no
// state, lib.rs:108-142): they live in a small per-thread cache keyed by everything a context is created from, so that
// a loop of extract() calls — cv-sfm/src/lib.rs:2200-2204 — re-uses one pyramid instead of allocating ~0.2 GB per frame.
struct CtxKey {
    akz_config cfg;
    int device, w, h;
    uint32_t max_keypoints;
    uint32_t arith = 0;   // akz_options.arith (tools/pin_arith.py names the value a given Rust build computes)
    bool same(const CtxKey& o) const
    {
        return arith == o.arith && cfg.maximum_features == o.cfg.maximum_features && cfg.num_sublevels == o.cfg.num_sublevels &&
               cfg.max_octave_evolution == o.cfg.max_octave_evolution && cfg.base_scale_offset == o.cfg.base_scale_offset &&
               cfg.initial_contrast == o.cfg.initial_contrast && cfg.contrast_percentile == o.cfg.contrast_percentile &&
               cfg.contrast_factor_num_bins == o.cfg.contrast_factor_num_bins && cfg.derivative_factor == o.cfg.derivative_factor &&
               cfg.detector_threshold == o.cfg.detector_threshold && cfg.descriptor_channels == o.cfg.descriptor_channels &&
               cfg.descriptor_pattern_size == o.cfg.descriptor_pattern_size && device == o.device && max_keypoints == o.max_keypoints;
    }
};
struct CtxCache {
    static constexpr int kSlots = 2;
    struct Slot {
        CtxKey key;
        akz_ctx* ctx = nullptr;
        uint64_t used = 0;
    } slots[kSlots];
    uint64_t tick = 0;
    ~CtxCache()
    {
        for (Slot& s : slots)
            if (s.ctx) akz_destroy(s.ctx);
    }
    akz_ctx* get(const CtxKey& k)
    {
        for (Slot& s : slots)
            if (s.ctx && s.key.same(k) && k.w <= s.key.w && k.h <= s.key.h) {
                s.used = ++tick;
                return s.ctx;
            }
        Slot* victim = &slots[0];
        for (Slot& s : slots) {
            if (!s.ctx) {
                victim = &s;
                break;
            }
            if (s.used < victim->used) victim = &s;
        }
        if (victim->ctx) akz_destroy(victim->ctx);
        victim->ctx = nullptr;
        require_abi();
        akz_options opt = {};
        opt.struct_size = (uint32_t)sizeof(opt);
        opt.arith = k.arith;
        check(akz_create_ex(&k.cfg, k.device, k.w, k.h, 1, k.max_keypoints, k.arith ? &opt : nullptr, &victim->ctx), "akz_create_ex");
        victim->key = k;
        victim->used = ++tick;
        return victim->ctx;
    }
    void drop(akz_ctx* c)
    {
        for (Slot& s : slots)
            if (s.ctx == c) {
                akz_destroy(c);
                s.ctx = nullptr;
            }
    }
};
inline CtxCache& ctx_cache()
{
    static thread_local CtxCache cache;
    return cache;
}
}  // namespace detail

class Akaze {
public:
    // the reference's 11 public fields, same names and defaults (lib.rs:109-185)
    std::size_t maximum_features = std::numeric_limits<std::size_t>::max();
    uint32_t num_sublevels = 4;
    uint32_t max_octave_evolution = 4;
    double base_scale_offset = 1.6;
    double initial_contrast = 0.001;
    double contrast_percentile = 0.7;
    std::size_t contrast_factor_num_bins = 300;
    double derivative_factor = 1.5;
    double detector_threshold = 0.001;
    std::size_t descriptor_channels = 3;
    std::size_t descriptor_pattern_size = 10;
    // placement (not part of the reference struct)
    int device = 0;
    // first capacity of the device's per-frame lists.  The reference's Vecs are unbounded; a call that overflows the
    // capacity is repeated with twice as much (up to the library's 262 144 per frame), so this is a starting point only.
    uint32_t initial_keypoint_capacity = 16384;
    // which of the reference's three un-vendored arithmetic orders the filters and half_size use (akz_options.arith,
    // AKZ_ARITH_* bits; 0 = what the crate sources imply).  `python3 tools/pin_arith.py <rust>_kps.csv <rust>_descs.txt image`
    // names the value that reproduces a given cargo build byte for byte.
    uint32_t arith = 0;

    static Akaze new_(double threshold)  // Akaze::new (lib.rs:147-152); `new` is reserved in C++
    {
        Akaze a;
        a.detector_threshold = threshold;
        return a;
    }
    static Akaze sparse() { return new_(0.01); }    // lib.rs:157-159
    static Akaze dense() { return new_(0.0001); }   // lib.rs:164-166

    using Features = std::pair<std::vector<KeyPoint>, std::vector<BitArray64>>;

    // Akaze::extract on a Luma8 image (lib.rs:295)
    Features extract(const GrayImageU8& img) const
    {
        return run(img.width, img.height, [&](akz_ctx* c, akz_keypoint* k, akz_descriptor* d, uint32_t cap, uint32_t* n) {
            return akz_extract_gray_u8(c, img.data, img.width, img.height, img.stride, k, d, cap, n);
        });
    }
    // Akaze::extract on a colour image: DynamicImage::grayscale() first (image.rs:45-46), on the device
    Features extract(const ColorImageU8& img) const
    {
        return run(img.width, img.height, [&](akz_ctx* c, akz_keypoint* k, akz_descriptor* d, uint32_t cap, uint32_t* n) {
            return akz_extract_color(c, img.data, AKZ_FMT_U8, img.channels, img.width, img.height, img.stride, k, d, cap, n);
        });
    }
    // Akaze::extract_from_gray_float_image (lib.rs:309)
    Features extract_from_gray_float_image(const GrayFloatImage& img) const
    {
        return run(img.width, img.height, [&](akz_ctx* c, akz_keypoint* k, akz_descriptor* d, uint32_t cap, uint32_t* n) {
            return akz_extract_gray_f32(c, img.data, img.width, img.height, img.stride, k, d, cap, n);
        });
    }

    akz_config config() const
    {
        akz_config c;
        c.maximum_features = (uint64_t)maximum_features;
        c.num_sublevels = num_sublevels;
        c.max_octave_evolution = max_octave_evolution;
        c.base_scale_offset = base_scale_offset;
        c.initial_contrast = initial_contrast;
        c.contrast_percentile = contrast_percentile;
        c.contrast_factor_num_bins = (uint64_t)contrast_factor_num_bins;
        c.derivative_factor = derivative_factor;
        c.detector_threshold = detector_threshold;
        c.descriptor_channels = (uint64_t)descriptor_channels;
        c.descriptor_pattern_size = (uint64_t)descriptor_pattern_size;
        return c;
    }

private:
    // one call, with growth: AKZ_E_INTERNAL (a device list overflowed) and AKZ_E_CAPACITY (more keypoints than the output
    // arrays hold) both mean "not enough room" — the reference has no such thing, so the call is repeated with more
    template <typename Call>
    Features run(int w, int h, Call call) const
    {
        constexpr uint32_t kLibraryMax = 262144;
        uint32_t cap = initial_keypoint_capacity < 64 ? 64 : (initial_keypoint_capacity > kLibraryMax ? kLibraryMax : initial_keypoint_capacity);
        if ((uint64_t)maximum_features < cap) cap = (uint32_t)(maximum_features < 64 ? 64 : maximum_features);
        for (;;) {
            detail::CtxKey key{config(), device, w, h, cap, arith};
            akz_ctx* c = detail::ctx_cache().get(key);
            std::vector<akz_keypoint> k(cap);
            std::vector<akz_descriptor> d(cap);
            uint32_t n = 0;
            const int32_t st = call(c, k.data(), d.data(), cap, &n);
            if (st == AKZ_OK) return convert(k, d, n);
            if ((st == AKZ_E_INTERNAL || st == AKZ_E_CAPACITY) && cap < kLibraryMax) {
                detail::ctx_cache().drop(c);
                cap = cap * 2 > kLibraryMax ? kLibraryMax : cap * 2;
                continue;
            }
            check(st, "akz_extract");
        }
    }
    static Features convert(const std::vector<akz_keypoint>& k, const std::vector<akz_descriptor>& d, uint32_t n)
    {
        Features out;
        out.first.reserve(n);
        out.second.reserve(n);
        for (uint32_t i = 0; i < n; ++i) {
            out.first.push_back(KeyPoint{{k[i].x, k[i].y}, k[i].response, k[i].size, k[i].octave, k[i].class_id, k[i].angle});
            BitArray64 b;
            for (int j = 0; j < 64; ++j) b[j] = d[i].bytes[j];
            out.second.push_back(b);
        }
        return out;
    }
};

}  // namespace akaze

namespace space {

struct Hamming {};  // bitarray::Hamming metric marker

// space::Neighbor<u32>
struct Neighbor {
    std::size_t index;
    uint32_t distance;
};

class Matcher {
public:
    explicit Matcher(uint32_t max_descriptors = 16384, int device = 0)
    {
        akaze::require_abi();
        akaze::check(hm_create(device, max_descriptors, max_descriptors, &ctx_), "hm_create");
    }
    ~Matcher()
    {
        if (ctx_) hm_destroy(ctx_);
    }
    Matcher(const Matcher&) = delete;
    Matcher& operator=(const Matcher&) = delete;
    hm_ctx* handle() { return ctx_; }

private:
    hm_ctx* ctx_ = nullptr;
};

// space::LinearKnn { metric: Hamming, iter }: exact k-NN by scanning `iter` (the target descriptors).
// The reference's callers ask once per query descriptor (akaze/tests/estimate_pose.rs:82-88).  `iter` goes to the device
// once (hm_set_targets, on the first question) and stays there; every knn() is then one launch over the resident targets
// — still launch-bound (ask for whole frames with knn_batch() or match_descriptors() where the loop is yours), but no
// longer an upload of the whole target set per query.
class LinearKnn {
public:
    LinearKnn(Hamming, const std::vector<akaze::BitArray64>& iter, Matcher& m) : iter_(iter), m_(m) {}
    // Knn::knn(&self, query, 2): sorted by (distance, index); the lowest index wins ties.
    std::array<Neighbor, 2> knn(const akaze::BitArray64& query, std::size_t num) const
    {
        if (num != 2) throw std::invalid_argument("the MI355X matcher implements knn(query, 2)");
        akz_neighbor out[2];
        ask(&query, 1, 2, out);
        return {Neighbor{out[0].index, out[0].distance}, Neighbor{out[1].index, out[1].distance}};
    }

    // Knn::knn(&self, query, num) -> Vec<Neighbor> for num = 1..3 (cv-sfm/src/lib.rs:1474 asks for 3):
    // min(num, iter.len()) neighbours, as the reference returns.
    std::vector<Neighbor> knn_vec(const akaze::BitArray64& query, std::size_t num) const
    {
        if (num < 1 || num > 3) throw std::invalid_argument("the MI355X matcher implements knn(query, k) for k <= 3");
        akz_neighbor out[3];
        ask(&query, 1, (uint32_t)num, out);
        std::vector<Neighbor> r;
        for (std::size_t i = 0; i < num && i < iter_.size(); ++i) r.push_back(Neighbor{out[i].index, out[i].distance});
        return r;
    }
    // the same for every query in one launch: out[i * num + j] = j-th neighbour of queries[i]
    std::vector<Neighbor> knn_batch(const std::vector<akaze::BitArray64>& queries, std::size_t num) const
    {
        if (num < 1 || num > 3) throw std::invalid_argument("the MI355X matcher implements knn(query, k) for k <= 3");
        std::vector<akz_neighbor> out(queries.size() * num);
        if (!queries.empty()) ask(queries.data(), (uint32_t)queries.size(), (uint32_t)num, out.data());
        std::vector<Neighbor> r(out.size());
        for (std::size_t i = 0; i < out.size(); ++i) r[i] = Neighbor{out[i].index, out[i].distance};
        return r;
    }

private:
    void ask(const akaze::BitArray64* q, uint32_t nq, uint32_t k, akz_neighbor* out) const
    {
        const akz_descriptor* t = reinterpret_cast<const akz_descriptor*>(iter_.data());
        // upload unless the matcher still holds THIS object's upload (its generation number: another LinearKnn on the same
        // matcher, or any host-buffer call that took the staging buffer, changes it)
        if (generation_ == 0 || hm_targets_generation(m_.handle()) != generation_) {
            akaze::check(hm_set_targets(m_.handle(), t, (uint32_t)iter_.size()), "hm_set_targets");
            generation_ = hm_targets_generation(m_.handle());
        }
        akaze::check(hm_knn_targets(m_.handle(), reinterpret_cast<const akz_descriptor*>(q), nq, k, out), "hm_knn_targets");
    }
    const std::vector<akaze::BitArray64>& iter_;
    Matcher& m_;
    mutable uint64_t generation_ = 0;
};

inline std::vector<std::array<std::size_t, 2>> run_match(Matcher& m, const std::vector<akaze::BitArray64>& a,
                                                         const std::vector<akaze::BitArray64>& b, int rule,
                                                         uint32_t pu, float pf, bool symmetric)
{
    std::vector<uint32_t> pairs(2 * (a.size() + 1));
    uint32_t n = 0;
    akaze::check(hm_match(m.handle(), reinterpret_cast<const akz_descriptor*>(a.data()), (uint32_t)a.size(),
                          reinterpret_cast<const akz_descriptor*>(b.data()), (uint32_t)b.size(), rule, pu, pf,
                          symmetric ? 1 : 0, pairs.data(), (uint32_t)a.size() + 1, &n),
                 "hm_match");
    std::vector<std::array<std::size_t, 2>> out(n);
    for (uint32_t i = 0; i < n; ++i) out[i] = {pairs[2 * i], pairs[2 * i + 1]};
    return out;
}
// symmetric_matching of tutorial ch5 (main.rs:183-200): keep [a, b] iff d0 + 24 < d1 both ways.
inline std::vector<std::array<std::size_t, 2>> symmetric_matching(Matcher& m, const std::vector<akaze::BitArray64>& a,
                                                                  const std::vector<akaze::BitArray64>& b)
{
    return run_match(m, a, b, HM_RULE_BETTER_BY_STRICT, 24, 0.f, true);
}
// match_descriptors of akaze/tests/estimate_pose.rs:78-97: a -> b only, Lowe ratio in f32.
inline std::vector<std::array<std::size_t, 2>> match_descriptors(Matcher& m, const std::vector<akaze::BitArray64>& ds1,
                                                                 const std::vector<akaze::BitArray64>& ds2,
                                                                 float lowes_ratio = 0.5f)
{
    return run_match(m, ds1, ds2, HM_RULE_LOWE, 0, lowes_ratio, false);
}

}  // namespace space

// ---- two-view / registration consensus: the sample_consensus::Consensus surface over rs_* -------------------------
//   cv_core::FeatureMatch(a, b) / FeatureWorldMatch(bearing, world)        cv-core/src/matches.rs
//   cv_pinhole::CameraIntrinsics::calibrate                                 cv-pinhole/src/lib.rs:108-117
//   eight_point::EightPoint, lambda_twist::LambdaTwist (Estimator)          eight-point/src/lib.rs:60-83, lambda-twist/src/lib.rs:330-347
//   arrsac::Arrsac::{new, initialization_hypotheses, max_candidate_hypotheses, estimations_per_block, block_size,
//                    likelihood_ratio_threshold} + Consensus::{model, model_inliers}
//                                                                           call sites akaze/tests/estimate_pose.rs:63-75,
//                                                                           vslam-sandbox/src/main.rs:105-117, cv-sfm/src/lib.rs:1394-1412
// The arrsac crate is not vendored in the reference: what runs is this library's ARRSAC-shaped procedure (include/akz.h:
// rs_essential_arrsac / rs_p3p_arrsac, specified by oracle/arrsac_oracle.c); the builder names are the crate's.
namespace cv_core {
struct FeatureMatch {
    std::array<double, 3> a, b;       // unit bearings of the two views
};
struct FeatureWorldMatch {
    std::array<double, 3> bearing;    // unit bearing in the camera
    std::array<double, 4> world;      // homogeneous world point (Projective form: xyz normalised, w = 1 / distance)
};
struct CameraToCamera {
    std::array<double, 12> rt;        // row-major 3 x 4 [R | t]
};
struct WorldToCamera {
    std::array<double, 12> rt;
};
}  // namespace cv_core

namespace cv_pinhole {
struct CameraIntrinsics {
    std::array<double, 2> focals, principal_point;
    double skew = 0.0;
    // CameraModel::calibrate (lib.rs:108-117): pixel keypoint -> unit bearing
    std::array<double, 3> calibrate(const akaze::KeyPoint& kp) const
    {
        const double intr[5] = {focals[0], focals[1], principal_point[0], principal_point[1], skew};
        akz_keypoint k{};
        k.x = kp.point.first;
        k.y = kp.point.second;
        std::array<double, 3> out{};
        akaze::check(rs_calibrate(intr, 0, 0.0, &k, 1, out.data()), "rs_calibrate");
        return out;
    }
};
}  // namespace cv_pinhole

namespace eight_point {
struct EightPoint {
    static constexpr std::size_t MIN_SAMPLES = 8;   // eight-point/src/lib.rs:73
};
}  // namespace eight_point
namespace lambda_twist {
struct LambdaTwist {
    static constexpr std::size_t MIN_SAMPLES = 3;   // lambda-twist/src/lib.rs:333
};
}  // namespace lambda_twist

namespace arrsac {

class Arrsac {
public:
    // Arrsac::new(inlier_threshold, rng): the rng is a seed here (the sampler is the library's xoshiro256++ streams)
    Arrsac(double inlier_threshold, uint64_t seed = 0, int device = 0) : device_(device)
    {
        std::memset(&p_, 0, sizeof(p_));
        p_.struct_size = sizeof(p_);
        // the defaults of arrsac 0.10's Arrsac::new as the crate documents them (un-vendored: not verifiable here) —
        // max_candidate_hypotheses 50, block_size 100, likelihood_ratio_threshold 1e3, initialization_hypotheses 256,
        // initialization_blocks 4, estimations_per_block 64 — so a caller ported with the bare constructor
        // (akaze/tests/estimate_pose.rs:63) runs the shape it ran before, inlier-guided re-sampling included
        p_.n_hypotheses = 256;
        p_.block_size = 100;
        p_.init_blocks = 4;
        p_.max_candidates = 50;
        p_.flags = RS_PRUNE_BOUND | RS_PRUNE_SPRT | RS_PRUNE_HALVE;
        p_.threshold = inlier_threshold;
        p_.sprt_delta = 0.05;
        p_.sprt_ratio = 1e3;
        p_.seed = seed;
        p_.estimations_per_block = 64;
    }
    ~Arrsac()
    {
        if (ctx_) rs_destroy(ctx_);
    }
    Arrsac(const Arrsac&) = delete;
    Arrsac& operator=(const Arrsac&) = delete;
    Arrsac(Arrsac&& o) noexcept : p_(o.p_), device_(o.device_), ctx_(o.ctx_), cap_m_(o.cap_m_), cap_h_(o.cap_h_) { o.ctx_ = nullptr; }

    // the crate's builder methods (vslam-sandbox/src/main.rs:105-117)
    Arrsac&& initialization_hypotheses(uint32_t n) && { p_.n_hypotheses = n; return std::move(*this); }
    Arrsac&& max_candidate_hypotheses(uint32_t n) && { p_.max_candidates = n; return std::move(*this); }
    Arrsac&& estimations_per_block(uint32_t n) && { p_.estimations_per_block = n; return std::move(*this); }
    Arrsac&& block_size(uint32_t n) && { p_.block_size = n; return std::move(*this); }
    Arrsac&& initialization_blocks(uint32_t n) && { p_.init_blocks = n; return std::move(*this); }
    Arrsac&& likelihood_ratio_threshold(double r) && { p_.sprt_ratio = r; return std::move(*this); }

    // Consensus<EightPoint, FeatureMatch>::model_inliers (eight-point/src/lib.rs:70-83 estimates, cv-core/src/pose.rs:249-295 scores)
    std::optional<std::pair<cv_core::CameraToCamera, std::vector<std::size_t>>> model_inliers(const eight_point::EightPoint&,
                                                                                              const std::vector<cv_core::FeatureMatch>& data)
    {
        if (data.size() < eight_point::EightPoint::MIN_SAMPLES) return std::nullopt;
        std::vector<double> a(3 * data.size()), b(3 * data.size());
        for (std::size_t i = 0; i < data.size(); ++i)
            for (int k = 0; k < 3; ++k) {
                a[3 * i + k] = data[i].a[k];
                b[3 * i + k] = data[i].b[k];
            }
        cv_core::CameraToCamera pose{};
        std::vector<std::size_t> inl;
        if (!run(false, a.data(), b.data(), (uint32_t)data.size(), pose.rt.data(), &inl)) return std::nullopt;
        return std::make_pair(pose, std::move(inl));
    }
    std::optional<cv_core::CameraToCamera> model(const eight_point::EightPoint& e, const std::vector<cv_core::FeatureMatch>& data)
    {
        auto r = model_inliers(e, data);
        if (!r) return std::nullopt;
        return r->first;
    }
    // Consensus<LambdaTwist, FeatureWorldMatch>::model_inliers (lambda-twist/src/lib.rs:330-347, cv-core/src/pose.rs:194-201)
    std::optional<std::pair<cv_core::WorldToCamera, std::vector<std::size_t>>> model_inliers(const lambda_twist::LambdaTwist&,
                                                                                             const std::vector<cv_core::FeatureWorldMatch>& data)
    {
        if (data.size() < lambda_twist::LambdaTwist::MIN_SAMPLES) return std::nullopt;
        std::vector<double> a(3 * data.size()), w(4 * data.size());
        for (std::size_t i = 0; i < data.size(); ++i) {
            for (int k = 0; k < 3; ++k) a[3 * i + k] = data[i].bearing[k];
            for (int k = 0; k < 4; ++k) w[4 * i + k] = data[i].world[k];
        }
        cv_core::WorldToCamera pose{};
        std::vector<std::size_t> inl;
        if (!run(true, a.data(), w.data(), (uint32_t)data.size(), pose.rt.data(), &inl)) return std::nullopt;
        return std::make_pair(pose, std::move(inl));
    }

private:
    bool run(bool p3p, const double* a, const double* b, uint32_t n, double* pose, std::vector<std::size_t>* inl)
    {
        const uint32_t blocks = (n + p_.block_size - 1) / p_.block_size;
        const uint32_t need_h = p_.n_hypotheses + p_.estimations_per_block * blocks;
        if (!ctx_ || n > cap_m_ || need_h > cap_h_) {
            if (ctx_) rs_destroy(ctx_);
            ctx_ = nullptr;
            cap_m_ = n < 64 ? 64 : n;
            cap_h_ = need_h;
            akaze::require_abi();
            akaze::check(rs_create(device_, cap_m_, cap_h_, &ctx_), "rs_create");
        }
        std::vector<uint32_t> idx(n);
        uint32_t best = 0, ninl = 0;
        const int32_t st = p3p ? rs_p3p_arrsac(ctx_, a, b, n, nullptr, &p_, pose, &best, idx.data(), n, &ninl, nullptr)
                               : rs_essential_arrsac(ctx_, a, b, n, nullptr, &p_, pose, &best, idx.data(), n, &ninl, nullptr);
        akaze::check(st, p3p ? "rs_p3p_arrsac" : "rs_essential_arrsac");
        if (best == 0xFFFFFFFFu) return false;     // Consensus::model_inliers returned None
        inl->assign(idx.begin(), idx.begin() + ninl);
        return true;
    }
    rs_arrsac_params p_;
    int device_;
    rs_ctx* ctx_ = nullptr;
    uint32_t cap_m_ = 0, cap_h_ = 0;
};

}  // namespace arrsac

// hamming_lsh::HammingHasher<64, H> and the lsh_to_frame map of cv-sfm (cv-sfm/src/lib.rs:205-217, 672, 622-624) over
// hm_hash_bag / hm_hash_knn.  The hashing crate is not vendored in the reference: see oracle/lsh_oracle.c for what
// is restated (nearest-codeword bag hash, one bit per codeword).
namespace hamming_lsh {

class HammingHasher {
public:
    // new_with_codewords: codewords.size() = 8 x hash bytes (4096 for cv-sfm's BitArray<512>)
    HammingHasher(std::vector<akaze::BitArray64> codewords, space::Matcher& m) : cw_(std::move(codewords)), m_(m)
    {
        if (cw_.empty() || cw_.size() % 32) throw std::invalid_argument("codeword count must be a multiple of 32");
    }
    std::size_t hash_bytes() const { return cw_.size() / 8; }
    // hash_bag(features) -> BitArray<H> as bytes
    std::vector<uint8_t> hash_bag(const std::vector<akaze::BitArray64>& features) const
    {
        std::vector<uint8_t> h(hash_bytes());
        akaze::check(hm_hash_bag(m_.handle(), reinterpret_cast<const akz_descriptor*>(features.data()),
                                 (uint32_t)features.size(), reinterpret_cast<const akz_descriptor*>(cw_.data()),
                                 (uint32_t)cw_.size(), h.data(), nullptr),
                     "hm_hash_bag");
        return h;
    }

private:
    std::vector<akaze::BitArray64> cw_;
    space::Matcher& m_;
};

// insert(hash, value) / knn_values(hash, num): exact nearest hashes in (distance, insertion order)
template <typename V>
class HashIndex {
public:
    HashIndex(std::size_t hash_bytes, space::Matcher& m) : hb_(hash_bytes), m_(m) {}
    void insert(const std::vector<uint8_t>& hash, V value)
    {
        if (hash.size() != hb_) throw std::invalid_argument("hash size");
        store_.insert(store_.end(), hash.begin(), hash.end());
        values_.push_back(std::move(value));
    }
    std::vector<std::pair<space::Neighbor, V>> knn_values(const std::vector<uint8_t>& hash, std::size_t num) const
    {
        std::vector<akz_neighbor> out(num ? num : 1);
        uint32_t n = 0;
        akaze::check(hm_hash_knn(m_.handle(), hash.data(), store_.data(), (uint32_t)values_.size(), (uint32_t)hb_,
                                 (uint32_t)num, out.data(), &n),
                     "hm_hash_knn");
        std::vector<std::pair<space::Neighbor, V>> r;
        for (uint32_t i = 0; i < n; ++i) r.push_back({space::Neighbor{out[i].index, out[i].distance}, values_[out[i].index]});
        return r;
    }

private:
    std::size_t hb_;
    space::Matcher& m_;
    std::vector<uint8_t> store_;
    std::vector<V> values_;
};

}  // namespace hamming_lsh
