// device_pipeline.cpp — INTEGRATION.md §4's loop from a NATIVE host (no Python, no PyTorch): device buffers from hipMalloc,
//   akz_extract_batch_device (both KITTI frames as one batch)
//   -> hm_match_batch_device (Lowe ratio 0.5, the rule of akaze/tests/estimate_pose.rs:78-97)
//   -> rs_essential_arrsac_batch_device (K_00 of the KITTI sequence, threshold 0.1: estimate_pose.rs:61-75)
// with nothing copied to the host in between, each stage ordered after the previous one by its stream.  Checked here:
// 399 / 343 descriptors, 11 matches, 11 inliers, and keypoints / descriptors byte-equal to the host API's
// (akz_extract_gray_u8) — what a Rust or C++ pipeline that owns device memory would see through the C ABI alone.
// usage: device_pipeline frame0.raw frame14.raw width height
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "akz.h"

#define HIPOK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            return 4;                                                                 \
        }                                                                             \
    } while (0)
#define AKZOK(x)                                                                      \
    do {                                                                              \
        int32_t s_ = (x);                                                             \
        if (s_ != AKZ_OK) {                                                           \
            fprintf(stderr, "%s: %s\n", #x, akz_strerror(s_));                        \
            return 3;                                                                 \
        }                                                                             \
    } while (0)

int main(int argc, char** argv)
{
    if (argc != 5) return 2;
    const int w = atoi(argv[3]), h = atoi(argv[4]);
    const size_t px = (size_t)w * h;
    std::vector<uint8_t> frames(2 * px);
    for (int f = 0; f < 2; ++f) {
        FILE* fp = fopen(argv[1 + f], "rb");
        if (!fp || fread(frames.data() + f * px, 1, px, fp) != px) return 2;
        fclose(fp);
    }
    const uint32_t cap = 2048;
    akz_config cfg;
    akz_config_default(&cfg);
    cfg.detector_threshold = 0.01;   // Akaze::sparse(), akaze/src/lib.rs:157-159
    akz_ctx* ak = nullptr;
    hm_ctx* hm = nullptr;
    rs_ctx* rs = nullptr;
    AKZOK(akz_create(&cfg, 0, w, h, 2, cap, &ak));
    AKZOK(hm_create(0, cap, cap, &hm));
    AKZOK(rs_create(0, cap, 1024, &rs));
    AKZOK(rs_batch_reserve(rs, 1));

    hipStream_t up = nullptr;
    HIPOK(hipStreamCreate(&up));
    uint8_t* d_img = nullptr;
    akz_keypoint* d_kps = nullptr;
    akz_descriptor* d_desc = nullptr;
    uint32_t *d_cnt = nullptr, *d_pairs = nullptr, *d_npairs = nullptr, *d_best = nullptr, *d_inl = nullptr, *d_ninl = nullptr;
    double* d_pose = nullptr;
    rs_arrsac_stats* d_stats = nullptr;
    HIPOK(hipMalloc((void**)&d_img, 2 * px));
    HIPOK(hipMalloc((void**)&d_kps, sizeof(akz_keypoint) * 2 * cap));
    HIPOK(hipMalloc((void**)&d_desc, sizeof(akz_descriptor) * 2 * cap));
    HIPOK(hipMalloc((void**)&d_cnt, sizeof(uint32_t) * 2));
    HIPOK(hipMalloc((void**)&d_pairs, sizeof(uint32_t) * 2 * cap));
    HIPOK(hipMalloc((void**)&d_npairs, sizeof(uint32_t)));
    HIPOK(hipMalloc((void**)&d_best, sizeof(uint32_t)));
    HIPOK(hipMalloc((void**)&d_inl, sizeof(uint32_t) * cap));
    HIPOK(hipMalloc((void**)&d_ninl, sizeof(uint32_t)));
    HIPOK(hipMalloc((void**)&d_pose, sizeof(double) * 12));
    HIPOK(hipMalloc((void**)&d_stats, sizeof(rs_arrsac_stats)));
    HIPOK(hipMemcpyAsync(d_img, frames.data(), 2 * px, hipMemcpyHostToDevice, up));

    // ---- the pipeline: three asynchronous calls, each waiting on the stream of the one before ----
    AKZOK(akz_extract_batch_device(ak, d_img, /*u8*/ 0, 2, w, h, d_kps, d_desc, cap, d_cnt, up));
    const uint32_t ia[1] = {0}, ib[1] = {1};
    AKZOK(hm_match_batch_device(hm, d_desc, d_cnt, d_desc, d_cnt, cap, ia, ib, 1, HM_RULE_LOWE, 0, 0.5f, /*symmetric*/ 0, d_pairs,
                                d_npairs, akz_stream(ak)));
    rs_camera cam;
    memset(&cam, 0, sizeof(cam));
    cam.fx = 9.842439e+02; cam.fy = 9.808141e+02; cam.cx = 6.900000e+02; cam.cy = 2.331966e+02;   // estimate_pose.rs:28-32
    rs_arrsac_params prm;
    memset(&prm, 0, sizeof(prm));
    prm.struct_size = sizeof(prm);
    prm.n_hypotheses = 512;
    prm.block_size = 64;
    prm.init_blocks = 4;
    prm.max_candidates = 1024;
    prm.flags = RS_PRUNE_BOUND | RS_PRUNE_SPRT;
    prm.threshold = 0.1;             // Arrsac::new(0.1, ..), estimate_pose.rs:63
    prm.sprt_delta = 0.05;
    prm.sprt_ratio = 1e3;
    prm.seed = 1;
    AKZOK(rs_essential_arrsac_batch_device(rs, d_kps, d_kps, cap, ia, ib, d_pairs, d_npairs, 1, &cam, &cam, &prm, 0, d_pose, d_best,
                                           d_inl, d_ninl, d_stats, hm_stream(hm)));
    AKZOK(rs_sync(rs));
    AKZOK(akz_sync(ak));
    AKZOK(hm_sync(hm));

    uint32_t cnt[2] = {0, 0}, npairs = 0, ninl = 0, best = 0;
    double pose[12];
    HIPOK(hipMemcpy(cnt, d_cnt, sizeof(cnt), hipMemcpyDeviceToHost));
    HIPOK(hipMemcpy(&npairs, d_npairs, 4, hipMemcpyDeviceToHost));
    HIPOK(hipMemcpy(&ninl, d_ninl, 4, hipMemcpyDeviceToHost));
    HIPOK(hipMemcpy(&best, d_best, 4, hipMemcpyDeviceToHost));
    HIPOK(hipMemcpy(pose, d_pose, sizeof(pose), hipMemcpyDeviceToHost));
    printf("descriptors %u %u\nmatches %u\ninliers %u (winner %u)\n", cnt[0], cnt[1], npairs, ninl, best);
    if (cnt[0] != 399 || cnt[1] != 343 || npairs != 11 || ninl != 11 || best == 0xFFFFFFFFu) return 1;
    const double det = pose[0] * (pose[5] * pose[10] - pose[6] * pose[9]) - pose[1] * (pose[4] * pose[10] - pose[6] * pose[8]) +
                       pose[2] * (pose[4] * pose[9] - pose[5] * pose[8]);
    if (det < 0.999999 || det > 1.000001) return 1;

    // ---- the same frames through the host API: identical bytes ----
    std::vector<akz_keypoint> dk(2 * cap), hk(cap);
    std::vector<akz_descriptor> dd(2 * cap), hd(cap);
    HIPOK(hipMemcpy(dk.data(), d_kps, sizeof(akz_keypoint) * 2 * cap, hipMemcpyDeviceToHost));
    HIPOK(hipMemcpy(dd.data(), d_desc, sizeof(akz_descriptor) * 2 * cap, hipMemcpyDeviceToHost));
    for (int f = 0; f < 2; ++f) {
        uint32_t n = 0;
        AKZOK(akz_extract_gray_u8(ak, frames.data() + f * px, w, h, w, hk.data(), hd.data(), cap, &n));
        if (n != cnt[f]) return 1;
        if (memcmp(hk.data(), dk.data() + (size_t)f * cap, sizeof(akz_keypoint) * n) != 0) return 1;
        if (memcmp(hd.data(), dd.data() + (size_t)f * cap, sizeof(akz_descriptor) * n) != 0) return 1;
    }
    printf("host api == device api\n");
    void* bufs[] = {d_img, d_kps, d_desc, d_cnt, d_pairs, d_npairs, d_best, d_inl, d_ninl, d_pose, d_stats};
    for (void* b : bufs) HIPOK(hipFree(b));
    HIPOK(hipStreamDestroy(up));
    rs_destroy(rs);
    hm_destroy(hm);
    akz_destroy(ak);
    printf("device_pipeline ok\n");
    return 0;
}
