// estimate_pose.cpp — the reference's integration test akaze/tests/estimate_pose.rs:24-59 restated
// against the C++ host-side mirror (include/akaze.hpp), i.e. through the C ABI of libakz.so:
//   Akaze::sparse().extract x2 on the two KITTI frames -> exactly 399 and 343 descriptors,
//   space::LinearKnn{Hamming}.knn(d, 2) + Lowe ratio 0.5 -> exactly 11 matches.
//   calibrate with K_00 -> Arrsac::new(0.1, rng) + EightPoint::new() -> model_inliers -> exactly 11 inliers (:63-75).
// Also checked here, because only a native caller can: the per-thread context cache (a second extract of the same size
// re-uses the pyramid) and the growth of the device lists (a first capacity far too small still gives 399 descriptors).
// usage: estimate_pose frame0.raw frame14.raw width height
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "akaze.hpp"

static const float LOWES_RATIO = 0.5f;

static std::vector<uint8_t> read_raw(const char* path, size_t n)
{
    std::vector<uint8_t> v(n);
    FILE* f = fopen(path, "rb");
    if (!f || fread(v.data(), 1, n, f) != n) {
        fprintf(stderr, "cannot read %s\n", path);
        exit(2);
    }
    fclose(f);
    return v;
}

static std::pair<std::vector<akaze::KeyPoint>, std::vector<akaze::BitArray64>> image_to_kps(const std::vector<uint8_t>& px,
                                                                                           int w, int h)
{
    akaze::Akaze a = akaze::Akaze::sparse();
    return a.extract(akaze::GrayImageU8{px.data(), w, h, w});
}

// match_descriptors, estimate_pose.rs:78-97, written against the Knn trait surface
static std::vector<std::pair<size_t, size_t>> match_descriptors_knn(space::Matcher& m, const std::vector<akaze::BitArray64>& ds1,
                                                                const std::vector<akaze::BitArray64>& ds2)
{
    std::vector<std::pair<size_t, size_t>> out;
    space::LinearKnn knn(space::Hamming{}, ds2, m);
    for (size_t ix1 = 0; ix1 < ds1.size(); ++ix1) {
        auto neighbors = knn.knn(ds1[ix1], 2);
        if ((float)neighbors[0].distance < (float)neighbors[1].distance * LOWES_RATIO)
            out.push_back({ix1, neighbors[0].index});
    }
    return out;
}

int main(int argc, char** argv)
{
    if (argc != 5) return 2;
    int w = atoi(argv[3]), h = atoi(argv[4]);
    auto f0 = read_raw(argv[1], (size_t)w * h), f1 = read_raw(argv[2], (size_t)w * h);
    try {
        auto r1 = image_to_kps(f0, w, h);
        auto r2 = image_to_kps(f1, w, h);
        printf("descriptors %zu %zu\n", r1.second.size(), r2.second.size());
        if (r1.second.size() != 399 || r2.second.size() != 343) return 1;  // estimate_pose.rs:41-42
        if (r1.first.size() != r1.second.size()) return 1;
        for (size_t i = 1; i < r1.first.size(); ++i)
            if (r1.first[i].response > r1.first[i - 1].response) return 1;  // ordered by response (lib.rs:326)
        space::Matcher m(4096);
        auto matches = match_descriptors_knn(m, r1.second, r2.second);
        auto batched = space::match_descriptors(m, r1.second, r2.second, LOWES_RATIO);
        printf("matches %zu (batched %zu)\n", matches.size(), batched.size());
        if (matches.size() != 11 || batched.size() != 11) return 1;  // estimate_pose.rs:59
        for (size_t i = 0; i < 11; ++i)
            if (matches[i].first != batched[i][0] || matches[i].second != batched[i][1]) return 1;
        auto sym = space::symmetric_matching(m, r1.second, r2.second);
        printf("symmetric better-by-24 matches %zu\n", sym.size());
        // estimate_pose.rs:28-32, 61-75: K_00 of the KITTI sequence, Arrsac::new(0.1, ..) + EightPoint::new()
        cv_pinhole::CameraIntrinsics intrinsics{{9.842439e+02, 9.808141e+02}, {6.900000e+02, 2.331966e+02}, 0.0};
        std::vector<cv_core::FeatureMatch> fm;
        for (auto& mt : matches)
            fm.push_back(cv_core::FeatureMatch{intrinsics.calibrate(r1.first[mt.first]), intrinsics.calibrate(r2.first[mt.second])});
        arrsac::Arrsac consensus(0.1, 1);
        auto res = consensus.model_inliers(eight_point::EightPoint{}, fm);
        if (!res) return 1;
        printf("inliers %zu\n", res->second.size());
        if (res->second.size() != 11) return 1;                              // estimate_pose.rs:75
        for (size_t i = 0; i < 11; ++i)
            if (res->second[i] != i) return 1;
        // R of the recovered pose is a rotation
        const auto& rt = res->first.rt;
        double det = rt[0] * (rt[5] * rt[10] - rt[6] * rt[9]) - rt[1] * (rt[4] * rt[10] - rt[6] * rt[8]) + rt[2] * (rt[4] * rt[9] - rt[5] * rt[8]);
        if (det < 0.999999 || det > 1.000001) return 1;
        // fewer matches than a minimal sample: None
        std::vector<cv_core::FeatureMatch> few(fm.begin(), fm.begin() + 7);
        if (consensus.model_inliers(eight_point::EightPoint{}, few)) return 1;
        // the bare constructor on a scene long enough for its inlier-guided re-sampling to run (defaults: 100-match blocks,
        // 4 initialisation blocks, 64 estimations per block): 900 matches of a known motion, a third of them wrong
        {
            const double ang = 0.07, ca = std::cos(ang), sa = std::sin(ang);
            const double R[9] = {ca, 0, sa, 0, 1, 0, -sa, 0, ca}, t[3] = {0.5, 0.05, 0.1};
            std::vector<cv_core::FeatureMatch> big;
            uint64_t st = 0x9E3779B97F4A7C15ull;
            auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
            auto unit = [](double x, double y, double z) {
                const double n = std::sqrt(x * x + y * y + z * z);
                return std::array<double, 3>{x / n, y / n, z / n};
            };
            size_t good = 0;
            for (int i = 0; i < 900; ++i) {
                const double X = 4 * rnd() - 2, Y = 2.4 * rnd() - 1.2, Z = 3 + 6 * rnd();
                cv_core::FeatureMatch f;
                f.a = unit(X, Y, Z);
                if (i % 3 == 2) f.b = unit(4 * rnd() - 2, 2.4 * rnd() - 1.2, 3 + 6 * rnd());   // an outlier
                else {
                    f.b = unit(R[0] * X + R[1] * Y + R[2] * Z + t[0], R[3] * X + R[4] * Y + R[5] * Z + t[1], R[6] * X + R[7] * Y + R[8] * Z + t[2]);
                    ++good;
                }
                big.push_back(f);
            }
            arrsac::Arrsac bare(1e-7, 3);
            auto rb = bare.model_inliers(eight_point::EightPoint{}, big);
            if (!rb) return 1;
            printf("resampled consensus inliers %zu of %zu true\n", rb->second.size(), good);
            if (rb->second.size() < good * 9 / 10 || rb->second.size() > good + 20) return 1;
            for (size_t i : rb->second)
                if (i >= big.size()) return 1;
            // the recovered rotation is the scene's (to the precision exact synthetic bearings allow)
            const auto& q = rb->first.rt;
            const double err = std::fabs(q[0] - R[0]) + std::fabs(q[2] - R[2]) + std::fabs(q[5] - R[4]) + std::fabs(q[8] - R[6]) + std::fabs(q[10] - R[8]);
            if (err > 1e-6) return 1;
        }
        // growth: a first capacity of 64 keypoints per frame still ends with the reference's 399
        akaze::Akaze tiny = akaze::Akaze::sparse();
        tiny.initial_keypoint_capacity = 64;
        auto rg = tiny.extract(akaze::GrayImageU8{f0.data(), w, h, w});
        printf("grown %zu\n", rg.second.size());
        if (rg.second.size() != 399 || rg.second != r1.second) return 1;
        // a colour view of the same frame (R = G = B = the gray value: integer Rec. 709 luma gives it back exactly)
        std::vector<uint8_t> rgb((size_t)w * h * 3);
        for (size_t i = 0; i < (size_t)w * h; ++i) rgb[3 * i] = rgb[3 * i + 1] = rgb[3 * i + 2] = f0[i];
        auto rc = akaze::Akaze::sparse().extract(akaze::ColorImageU8{rgb.data(), w, h, 3 * w, 3});
        printf("colour %zu\n", rc.second.size());
        if (rc.second != r1.second) return 1;
    } catch (const akaze::Error& e) {
        fprintf(stderr, "akaze error: %s\n", e.what());
        return 3;
    }
    printf("estimate_pose ok\n");
    return 0;
}
