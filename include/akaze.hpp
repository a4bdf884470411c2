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
#include <limits>
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
    uint32_t max_keypoints = 16384;

    Akaze() = default;
    Akaze(const Akaze& o) { copy_fields(o); }
    Akaze& operator=(const Akaze& o)
    {
        if (this != &o) {
            release();
            copy_fields(o);
        }
        return *this;
    }
    ~Akaze() { release(); }

    static Akaze new_(double threshold)  // Akaze::new (lib.rs:147-152); `new` is reserved in C++
    {
        Akaze a;
        a.detector_threshold = threshold;
        return a;
    }
    static Akaze sparse() { return new_(0.01); }    // lib.rs:157-159
    static Akaze dense() { return new_(0.0001); }   // lib.rs:164-166

    // Akaze::extract on a Luma8 image (lib.rs:295)
    std::pair<std::vector<KeyPoint>, std::vector<BitArray64>> extract(const GrayImageU8& img)
    {
        ensure(img.width, img.height);
        std::vector<akz_keypoint> k(max_keypoints);
        std::vector<akz_descriptor> d(max_keypoints);
        uint32_t n = 0;
        check(akz_extract_gray_u8(ctx_, img.data, img.width, img.height, img.stride, k.data(), d.data(), max_keypoints, &n),
              "akz_extract_gray_u8");
        return convert(k, d, n);
    }
    // Akaze::extract_from_gray_float_image (lib.rs:309)
    std::pair<std::vector<KeyPoint>, std::vector<BitArray64>> extract_from_gray_float_image(const GrayFloatImage& img)
    {
        ensure(img.width, img.height);
        std::vector<akz_keypoint> k(max_keypoints);
        std::vector<akz_descriptor> d(max_keypoints);
        uint32_t n = 0;
        check(akz_extract_gray_f32(ctx_, img.data, img.width, img.height, img.stride, k.data(), d.data(), max_keypoints, &n),
              "akz_extract_gray_f32");
        return convert(k, d, n);
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
    akz_ctx* ctx_ = nullptr;
    int ctx_w_ = 0, ctx_h_ = 0;

    void copy_fields(const Akaze& o)
    {
        maximum_features = o.maximum_features;
        num_sublevels = o.num_sublevels;
        max_octave_evolution = o.max_octave_evolution;
        base_scale_offset = o.base_scale_offset;
        initial_contrast = o.initial_contrast;
        contrast_percentile = o.contrast_percentile;
        contrast_factor_num_bins = o.contrast_factor_num_bins;
        derivative_factor = o.derivative_factor;
        detector_threshold = o.detector_threshold;
        descriptor_channels = o.descriptor_channels;
        descriptor_pattern_size = o.descriptor_pattern_size;
        device = o.device;
        max_keypoints = o.max_keypoints;
    }
    void release()
    {
        if (ctx_) akz_destroy(ctx_);
        ctx_ = nullptr;
    }
    void ensure(int w, int h)
    {
        if (ctx_ && w <= ctx_w_ && h <= ctx_h_) return;
        release();
        akz_config c = config();
        check(akz_create(&c, device, w, h, 1, max_keypoints, &ctx_), "akz_create");
        ctx_w_ = w;
        ctx_h_ = h;
    }
    static std::pair<std::vector<KeyPoint>, std::vector<BitArray64>> convert(const std::vector<akz_keypoint>& k,
                                                                             const std::vector<akz_descriptor>& d,
                                                                             uint32_t n)
    {
        std::pair<std::vector<KeyPoint>, std::vector<BitArray64>> out;
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
class LinearKnn {
public:
    LinearKnn(Hamming, const std::vector<akaze::BitArray64>& iter, Matcher& m) : iter_(iter), m_(m) {}
    // Knn::knn(&self, query, 2): sorted by (distance, index); the lowest index wins ties.
    std::array<Neighbor, 2> knn(const akaze::BitArray64& query, std::size_t num) const
    {
        if (num != 2) throw std::invalid_argument("the MI355X matcher implements knn(query, 2)");
        akz_neighbor out[2];
        akaze::check(hm_knn2(m_.handle(), reinterpret_cast<const akz_descriptor*>(query.data()), 1,
                             reinterpret_cast<const akz_descriptor*>(iter_.data()), (uint32_t)iter_.size(), out),
                     "hm_knn2");
        return {Neighbor{out[0].index, out[0].distance}, Neighbor{out[1].index, out[1].distance}};
    }

    // Knn::knn(&self, query, num) -> Vec<Neighbor> for num = 1..3 (cv-sfm/src/lib.rs:1474 asks for 3):
    // min(num, iter.len()) neighbours, as the reference returns.
    std::vector<Neighbor> knn_vec(const akaze::BitArray64& query, std::size_t num) const
    {
        if (num < 1 || num > 3) throw std::invalid_argument("the MI355X matcher implements knn(query, k) for k <= 3");
        akz_neighbor out[3];
        akaze::check(hm_knn(m_.handle(), reinterpret_cast<const akz_descriptor*>(query.data()), 1,
                            reinterpret_cast<const akz_descriptor*>(iter_.data()), (uint32_t)iter_.size(),
                            (uint32_t)num, out),
                     "hm_knn");
        std::vector<Neighbor> r;
        for (std::size_t i = 0; i < num && i < iter_.size(); ++i) r.push_back(Neighbor{out[i].index, out[i].distance});
        return r;
    }

private:
    const std::vector<akaze::BitArray64>& iter_;
    Matcher& m_;
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
