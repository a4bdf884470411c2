"""GPU parity tests proper: the HIP path, called through the C ABI (ctypes), against the CPU oracle
on the same inputs.  Bit-exact for every pyramid buffer, keypoint field, descriptor byte and match
index (SURVEY.md §8d "Tolerances"; the angle is bit-exact too because both sides evaluate
include/akz_portable_math.h)."""
import json
import os

import numpy as np
import pytest

from conftest import synth_frame

pytestmark = pytest.mark.gpu

# Contexts are created with the library's DEFAULT options (the configuration bench.py measures: transient
# Lsmooth/Lflow scratch, no Ldet planes) unless a test asks for something else through akz_options; the tests that
# tap every pyramid buffer pass keep_all=True.  The library reads no environment variable.


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from cv_amd import build
    build.build()
    from cv_amd import akaze, knn
    return akaze, knn


def _opts(**kw):
    from cv_amd import _lib
    return _lib.make_options(**kw)


def _eq(a, b, what):
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.dtype == np.float32:
        same = a.view(np.uint32) == b.view(np.uint32)      # bit-exact, sign of zero and NaN payload included
    elif a.dtype == np.float64:
        same = a.view(np.uint64) == b.view(np.uint64)
    else:
        same = a == b
    if not same.all():
        bad = np.argwhere(~same)
        raise AssertionError(f"{what}: {len(bad)} of {a.size} differ; first at {bad[0]}: "
                             f"{a[tuple(bad[0])]!r} vs {b[tuple(bad[0])]!r}")


def _kp_eq(a, b, what):
    assert len(a) == len(b), (what, len(a), len(b))
    for f in a.dtype.names:
        _eq(a[f], b[f], f"{what}.{f}")


# ---------------------------------------------------------------------------------------------
def test_filters_bit_exact(gpu, oracle, kitti):
    """akaze::image::{horizontal,vertical}_filter (image.rs:202-331) incl. the reference's own test
    kernel gaussian_kernel(3.0, 7), its bench kernels (1.0,7) and (10.0,71), and an asymmetric one."""
    akaze, _ = gpu
    O = oracle
    img = O.u8_to_f32(kitti[0])
    ctx = akaze.Akaze().context(img.shape[1], img.shape[0])
    small = O.u8_to_f32(synth_frame(157, 83, 3))
    kernels = [O.gaussian_kernel(3.0, 7), O.gaussian_kernel(1.0, 7), O.gaussian_kernel(10.0, 71),
               np.array([1.0, 2.0, -0.5, 0.25, 3.0], np.float32), np.array([-1.0, 0.0, 1.0], np.float32),
               O.gaussian_kernel(1.6, 9)]
    for k in kernels:
        _eq(akaze.gaussian_kernel(3.0, 7), O.gaussian_kernel(3.0, 7), "gaussian_kernel")
        for im in (img, small):
            _eq(akaze.horizontal_filter(im, k, ctx), O.horizontal_filter(im, k), f"horizontal k={len(k)}")
            _eq(akaze.vertical_filter(im, k, ctx), O.vertical_filter(im, k), f"vertical k={len(k)}")
    _eq(akaze.gaussian_blur(img, 1.6, ctx), O.gaussian_blur(img, 1.6), "gaussian_blur")


def test_half_size_bit_exact(gpu, oracle):
    akaze, _ = gpu
    rng = np.random.default_rng(5)
    ctx = akaze.Akaze().context(128, 128)
    for (h, w) in ((64, 96), (65, 96), (64, 97), (65, 97), (3, 2), (101, 7)):
        img = rng.random((h, w), dtype=np.float32)
        _eq(akaze.half_size(img, ctx), oracle.half_size(img), f"half_size {w}x{h}")


def _compare_pyramid(akaze, O, img, thr, what, ak=None, ocfg=None):
    """Every pyramid buffer, the contrast factor, every keypoint stage and the final outputs of one frame, HIP
    (keep_all context: all taps) vs oracle; then the final outputs once more from a DEFAULT-options context (the
    benchmarked mode: scratch-aliased Lsmooth/Lflow, no Ldet planes)."""
    h, w = img.shape
    ak = ak or akaze.Akaze.new(thr)
    ctx = akaze.Context(ak, w, h, 1, _opts(keep_all=True))
    (kp, desc), = ctx.extract_batch([img])
    orc = O.Akaze(w, h, ocfg or O.default_config(threshold=thr))
    okp, odesc = orc.extract(img)
    assert ctx.num_levels(w, h) == orc.num_levels
    for lvl in range(orc.num_levels):
        gi, oi = ctx.level(w, h, lvl), orc.level(lvl)
        for f in ("width", "height", "octave", "sublevel", "esigma", "etime", "n_fed_steps", "deriv_sigma"):
            assert getattr(gi, f) == getattr(oi, f), (what, lvl, f)
        _eq(ctx.fed_tau(w, h, lvl), orc.fed_tau(lvl), f"{what} tau[{lvl}]")
    assert ctx.contrast(0) == orc.contrast, (what, ctx.contrast(0), orc.contrast)
    for lvl in range(orc.num_levels):
        for name in ("Lt", "Lsmooth", "Lflow", "Lx", "Ly", "Ldet"):
            if lvl == 0 and name == "Lflow":
                continue
            _eq(ctx.level_buffer(0, lvl, name, w, h), orc.buffer(lvl, name), f"{what} {name}[{lvl}]")
    for stage in (0, 1, 2):
        _kp_eq(ctx.keypoints(0, stage), orc.keypoints(stage), f"{what} stage{stage}")
    _kp_eq(kp, okp, f"{what} final keypoints")
    _eq(desc, odesc, f"{what} descriptors")
    ctx.close()
    ctx = akaze.Context(ak, w, h, 1, _opts(resident_min_frames=1))      # (default but for k_level_resident on a one-frame call)
    (kp2, desc2), = ctx.extract_batch([img])
    _kp_eq(kp2, okp, f"{what} final keypoints (default options)")
    _eq(desc2, odesc, f"{what} descriptors (default options)")
    # the planes that outlive a default-options call (the descriptor stage samples them): the kernels the benchmark runs
    # — k_front_fed, k_fed_pair's fused half-size, k_level_resident — leave the oracle's Lt, Lx, Ly at every level
    for lvl in range(orc.num_levels):
        for name in ("Lt", "Lx", "Ly"):
            _eq(ctx.level_buffer(0, lvl, name, w, h), orc.buffer(lvl, name), f"{what} {name}[{lvl}] (default options)")
    ctx.close()
    return kp, desc


def test_kitti_every_buffer_and_stage(gpu, oracle, kitti, kitti_golden):
    """Config 1 of BASELINE.json through the HIP path: every pyramid buffer, every keypoint stage,
    every descriptor bit equal to the oracle; counts equal the reference's pins
    (akaze/tests/estimate_pose.rs:41-42)."""
    akaze, _ = gpu
    kp0, d0 = _compare_pyramid(akaze, oracle, kitti[0], 0.01, "kitti0 sparse")
    kp1, d1 = _compare_pyramid(akaze, oracle, kitti[1], 0.01, "kitti14 sparse")
    assert len(d0) == 399 and len(d1) == 343
    _eq(d0, kitti_golden["sparse_desc0"], "golden desc0")
    _eq(d1, kitti_golden["sparse_desc14"], "golden desc14")
    kpd, dd = _compare_pyramid(akaze, oracle, kitti[0], 0.001, "kitti0 default")
    _eq(dd, kitti_golden["default_desc0"], "golden default desc0")
    assert kpd.tobytes() == kitti_golden["default_kp0"].tobytes()


@pytest.mark.parametrize("shape", [(333, 251), (640, 480), (97, 83), (258, 130)])
def test_ragged_sizes(gpu, oracle, shape):
    """Odd widths/heights: scalar FED path, odd half_size edges, partial tiles, few octaves."""
    akaze, _ = gpu
    w, h = shape
    img = synth_frame(w, h, seed=w * 1000 + h, n_rect=25, n_disc=25)
    _compare_pyramid(akaze, oracle, img, 0.001, f"synth {w}x{h}")


@pytest.mark.parametrize("name", ["constant", "checkerboard16", "noise_dense", "tiny", "gradient"])
def test_pathological_inputs(gpu, oracle, name):
    """Inputs the reference's tests do not have: no structure at all, a lattice full of exact ties, noise at the
    dense threshold (21 091 candidates, 10 262 keypoints at 640x480 - exercises the capacity paths of the
    parallel suppression), an image smaller than two tiles, a pure ramp.  Same keypoints, same descriptors."""
    akaze, _ = gpu
    rng = np.random.default_rng(1)
    yy, xx = np.indices((480, 640))
    img, thr = {
        "constant": (np.full((240, 320), 128, np.uint8), 0.001),
        "checkerboard16": ((((yy // 16) + (xx // 16)) % 2 * 255).astype(np.uint8), 0.001),
        "noise_dense": (rng.integers(0, 256, (480, 640), dtype=np.uint8), 0.0001),
        "tiny": (rng.integers(0, 256, (48, 64), dtype=np.uint8), 0.001),
        "gradient": (np.tile(np.linspace(0, 255, 640).astype(np.uint8), (480, 1)), 0.0001),
    }[name]
    h, w = img.shape
    ctx = akaze.Context(akaze.Akaze.new(thr), w, h, 1)
    (kp, desc), = ctx.extract_batch([img])
    okp, odesc = oracle.Akaze(w, h, oracle.default_config(threshold=thr)).extract(img)
    _kp_eq(kp, okp, name)
    _eq(desc, odesc, name + " desc")
    ctx.close()


@pytest.mark.parametrize("mode", ["exact", "force_odd"])
def test_contrast_factor_paths(gpu, oracle, mode):
    """The contrast factor normally comes from the order statistic of the max pass's fine histogram; frames whose
    fine key straddles a reference bin take the exact histogram pass.  AKZ_OPT_CONTRAST_EXACT sends every frame
    through the exact pass, AKZ_OPT_CONTRAST_FORCE_ODD every odd frame (mixed pairs): same contrast factor, bit for
    bit, either way."""
    akaze, _ = gpu
    frames = [synth_frame(320, 240, seed=500 + i, n_rect=10 + 7 * i, n_disc=5 + 3 * i) for i in range(5)]
    frames[3] = np.full((240, 320), 77, np.uint8)                     # no gradient at all: zero points
    ctx = akaze.Context(akaze.Akaze.default(), 320, 240, 5, _opts(contrast=mode))
    got = ctx.extract_batch(frames)
    for i, img in enumerate(frames):
        orc = oracle.Akaze(320, 240, oracle.default_config())
        okp, od = orc.extract(img)
        assert ctx.contrast(i) == orc.contrast, f"frame {i} contrast"
        _kp_eq(got[i][0], okp, f"frame {i}")
        _eq(got[i][1], od, f"frame {i} desc")
    ctx.close()


def test_f32_input_path(gpu, oracle):
    akaze, _ = gpu
    img = oracle.u8_to_f32(synth_frame(320, 240, 11))
    _compare_pyramid(akaze, oracle, img, 0.001, "f32 input")


def test_batch_equals_single(gpu, oracle):
    """Frames in a batch are independent (Akaze is Copy and stateless, lib.rs:108): a batched launch
    gives the same outputs as one-by-one calls, and as the oracle."""
    akaze, _ = gpu
    frames = [synth_frame(480, 270, 100 + i) for i in range(5)]
    ak = akaze.Akaze.default()
    res = ak.context(480, 270, 5).extract_batch(frames)
    orc = oracle.Akaze(480, 270, oracle.default_config())
    for i, f in enumerate(frames):
        okp, odesc = orc.extract(f)
        _kp_eq(res[i][0], okp, f"batch frame {i}")
        _eq(res[i][1], odesc, f"batch frame {i} desc")


def test_pipelined_device_calls_match_host_api(gpu):
    """akz_extract_batch_device back to back without any synchronisation in between (two buffer sets, three
    streams inside the library, an odd batch so the last frame pair is half empty): every call's outputs equal
    what the blocking host-buffer API returns for the same frames, and a second pass reproduces them bit for
    bit (size-independent properties at BASELINE's frame size: independence of frames, determinism)."""
    import torch
    akaze, _ = gpu
    from cv_amd import _lib
    L = _lib.lib()
    W, H, B, CAP, NCALL = 1920, 1080, 3, 8192, 4
    dev = torch.device("cuda", 0)
    frames = [synth_frame(W, H, 900 + i, n_rect=150, n_disc=150) for i in range(B * NCALL)]
    d_frames = torch.from_numpy(np.stack(frames)).to(dev)
    ak = akaze.Akaze.default()
    ak.max_keypoints = CAP
    ctx = akaze.Context(ak, W, H, B)
    outs = []
    for rep in range(2):
        kps = torch.zeros((NCALL, B, CAP, 28), dtype=torch.uint8, device=dev)
        descs = torch.zeros((NCALL, B, CAP, 64), dtype=torch.uint8, device=dev)
        cnt = torch.zeros((NCALL, B), dtype=torch.int32, device=dev)
        cur = torch.cuda.current_stream()
        for k in range(NCALL):
            _lib.check(L.akz_extract_batch_device(ctx.handle, d_frames[k * B:(k + 1) * B].data_ptr(), 0, B, W, H,
                                                  kps[k].data_ptr(), descs[k].data_ptr(), CAP, cnt[k].data_ptr(),
                                                  _lib.wait_handle(cur)), "extract")
        _lib.check(L.akz_sync(ctx.handle), "sync")
        outs.append((kps.cpu().numpy(), descs.cpu().numpy(), cnt.cpu().numpy()))
    assert np.array_equal(outs[0][2], outs[1][2]) and np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][0], outs[1][0])
    ref = akaze.Context(ak, W, H, 1)
    kps, descs, cnt = outs[0]
    for k in range(NCALL):
        for j in range(B):
            (rkp, rdesc), = ref.extract_batch([frames[k * B + j]])
            n = int(cnt[k, j])
            assert n == len(rkp) and n > 1000, (k, j, n, len(rkp))
            assert np.array_equal(descs[k, j, :n], rdesc)
            assert kps[k, j, :n].tobytes() == rkp.tobytes()
    ctx.close()
    ref.close()


def test_full_hd_frame(gpu, oracle):
    """One frame at BASELINE's full size (1920x1080, 16 levels, 166 FED steps)."""
    akaze, _ = gpu
    img = synth_frame(1920, 1080, 4242, n_rect=200, n_disc=200)
    ak = akaze.Akaze.default()
    kp, desc = ak.extract_arrays(img)
    orc = oracle.Akaze(1920, 1080, oracle.default_config())
    okp, odesc = orc.extract(img)
    _kp_eq(kp, okp, "1080p keypoints")
    _eq(desc, odesc, "1080p descriptors")
    assert len(kp) > 500


@pytest.mark.parametrize("nch", [1, 2, 3])
def test_descriptor_channels(gpu, oracle, nch):
    """descriptor_channels 1 / 2 / 3 (descriptors.rs:145-156, :188): 1 and 2 run the generic kernel."""
    akaze, _ = gpu
    img = synth_frame(400, 300, 77)
    ak = akaze.Akaze(descriptor_channels=nch)
    kp, desc = ak.extract_arrays(img)
    cfg = oracle.default_config()
    cfg.descriptor_channels = nch
    okp, odesc = oracle.Akaze(400, 300, cfg).extract(img)
    _kp_eq(kp, okp, f"nch={nch}")
    _eq(desc, odesc, f"nch={nch} desc")
    used_bits = nch * 162
    assert not np.unpackbits(desc, axis=1, bitorder="little")[:, used_bits:].any()


def test_serial_suppression_path(gpu, oracle, kitti):
    """AKZ_OPT_SERIAL_SUPPRESSION routes every frame through the one-wave serial pass (the fallback of the parallel
    suppression for frames that overflow its fixed-capacity lists): same keypoints, same descriptors."""
    akaze, _ = gpu
    ctx = akaze.Context(akaze.Akaze.sparse(), 1392, 512, 2, _opts(parallel_suppression=False))
    res = ctx.extract_batch([kitti[0], kitti[1]])
    orc = oracle.Akaze(1392, 512, oracle.default_config(threshold=0.01))
    for i in range(2):
        okp, odesc = orc.extract(kitti[i])
        _kp_eq(res[i][0], okp, f"serial suppression frame {i}")
        _eq(res[i][1], odesc, f"serial suppression frame {i} desc")
    ctx.close()
    # frames with more candidates than the parallel path is sized for are flagged on the device and fall back
    ctx = akaze.Context(akaze.Akaze.sparse(), 1392, 512, 2, _opts(sup_capacity=700))   # frame 0 has 1022 candidates, frame 14 has 844
    res = ctx.extract_batch([kitti[0], kitti[1]])
    for i in range(2):
        okp, odesc = orc.extract(kitti[i])
        _kp_eq(res[i][0], okp, f"fallback frame {i}")
        _eq(res[i][1], odesc, f"fallback frame {i} desc")
    ctx.close()


def test_maximum_features_and_capacity(gpu, oracle, kitti):
    """lib.rs:326-327 truncation; AKZ_E_CAPACITY reports the required count."""
    akaze, _ = gpu
    import ctypes as C
    from cv_amd import _lib
    ak = akaze.Akaze(maximum_features=100)
    kp, desc = ak.extract_arrays(kitti[0])
    orc = oracle.Akaze(kitti[0].shape[1], kitti[0].shape[0], oracle.default_config(maximum_features=100))
    okp, odesc = orc.extract(kitti[0])
    _kp_eq(kp, okp, "truncated")
    _eq(desc, odesc, "truncated desc")
    assert len(kp) <= 100
    ctx = akaze.Akaze.sparse().context(1392, 512)
    kps = np.zeros(10, _lib.KP_DTYPE); descs = np.zeros((10, 64), np.uint8); n = C.c_uint32()
    img = np.ascontiguousarray(kitti[0])
    st = _lib.lib().akz_extract_gray_u8(ctx.handle, img.ctypes.data, 1392, 512, 1392, kps.ctypes.data,
                                        descs.ctypes.data, 10, C.byref(n))
    assert st == -4 and n.value == 399


# ---------------------------------------------------------------------------------------------
def _rand_desc(rng, n):
    d = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    d[:, 61:] = 0
    d[:, 60] &= 0x3F
    return d


def test_knn2_bit_exact_with_ties(gpu, oracle):
    """LinearKnn::knn(q, 2): indices AND distances equal, including tie-breaks (lowest index wins)."""
    _, knn = gpu
    rng = np.random.default_rng(21)
    m = knn.Matcher(8192)
    for nq, nt in ((1, 2), (7, 3), (513, 257), (1000, 777), (300, 5000)):
        q = _rand_desc(rng, nq); t = _rand_desc(rng, nt)
        t[rng.integers(0, nt, nt // 3)] = t[rng.integers(0, nt, nt // 3)]   # exact duplicates -> ties
        if nq > 5:
            q[:5] = t[:5] if nt >= 5 else q[:5]
        got, want = m.knn2(q, t), oracle.knn2(q, t)
        _eq(got["index"], want["index"], f"knn idx {nq}x{nt}")
        _eq(got["distance"], want["distance"], f"knn dist {nq}x{nt}")
    with pytest.raises(Exception):
        m.knn2(_rand_desc(rng, 3), _rand_desc(rng, 1))


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 2, 3])
def test_knn_k_bit_exact(gpu, oracle, k):
    """LinearKnn::knn(q, k) for k = 1..3 (cv-sfm's registration path uses 3): indices and distances equal the
    oracle's, ties included; neighbours that do not exist come back as the documented sentinel."""
    _, knn = gpu
    rng = np.random.default_rng(100 + k)
    m = knn.Matcher(8192)
    for nq, nt in ((5, k), (7, k + 1), (513, 257), (333, 4097)):
        q = _rand_desc(rng, nq); t = _rand_desc(rng, nt)
        if nt > 8:
            t[rng.integers(0, nt, nt // 3)] = t[rng.integers(0, nt, nt // 3)]   # exact duplicates -> ties
            q[:4] = t[:4]
        got, want = m.knn(q, t, k), oracle.knn(q, t, k)
        _eq(got["index"], want["index"], f"knn{k} idx {nq}x{nt}")
        _eq(got["distance"], want["distance"], f"knn{k} dist {nq}x{nt}")
    if k > 1:   # fewer targets than k: the reference's Vec is shorter; the fixed-width output pads with the sentinel
        q = _rand_desc(rng, 9); t = _rand_desc(rng, k - 1)
        got, want = m.knn(q, t, k), oracle.knn(q, t, k)
        _eq(got["index"][:, :k - 1], want["index"][:, :k - 1], "short idx")
        _eq(got["distance"][:, :k - 1], want["distance"][:, :k - 1], "short dist")
        assert (got["index"][:, k - 1:] == (1 << 22) - 1).all() and (got["distance"][:, k - 1:] == 1023).all()
    # LinearKnn mirror
    lk = knn.LinearKnn(knn.Hamming, t)
    nn = lk.knn(q[0], k)
    assert len(nn) == min(k, len(t))


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 2, 3])
def test_knn_at_the_tile_stage_and_block_boundaries(gpu, oracle, k):
    """The wide FP4 kernel streams targets in 32-row tiles, two tiles per LDS-DMA stage, three stages in a ring, with the stage
    boundary inside the second tile's MFMA chain; a wave holds 64 queries, a block 256.  Every target count around a tile, a
    stage and a ring revolution against every query count around a wave and a block, ties included, k = 1..3 — and the same
    through the register-staged kernel."""
    _, knn = gpu
    rng = np.random.default_rng(900 + k)
    nts = [k, 31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 128, 129, 191, 192, 193, 255, 256, 257, 383, 385]
    nqs = [1, 31, 33, 63, 64, 65, 191, 255, 256, 257]
    pool_q = _rand_desc(rng, max(nqs)); pool_t = _rand_desc(rng, max(nts))
    pool_t[rng.integers(0, len(pool_t), 150)] = pool_t[rng.integers(0, len(pool_t), 150)]      # duplicates: ties across tiles
    pool_q[:40] = pool_t[rng.integers(0, len(pool_t), 40)]                                        # distance 0
    for kernel in ("fp4", "fp4_regs"):
        m = knn.Matcher(1024, kernel=kernel)
        for nt in nts:
            want_all = oracle.knn(pool_q, pool_t[:nt], k)
            for nq in nqs:
                got = m.knn(pool_q[:nq], pool_t[:nt], k)
                _eq(got["index"], want_all["index"][:nq], f"{kernel} knn{k} idx {nq}x{nt}")
                _eq(got["distance"], want_all["distance"][:nq], f"{kernel} knn{k} dist {nq}x{nt}")
        m.close()


@pytest.mark.gpu
def test_knn_views_device(gpu, oracle):
    """hm_knn_views_device: one query frame against several stored views, k = 3, device-resident."""
    import ctypes as C
    import torch
    _, knn = gpu
    from cv_amd import _lib
    rng = np.random.default_rng(77)
    cap, nviews = 600, 5
    counts = np.array([600, 123, 2, 511, 37], np.int32)
    views = np.zeros((nviews, cap, 64), np.uint8)
    for v in range(nviews):
        views[v, :counts[v]] = _rand_desc(rng, int(counts[v]))
    q = np.zeros((cap, 64), np.uint8)
    nq = 421
    q[:nq] = _rand_desc(rng, nq)
    q[:3] = views[0, :3]
    dev = torch.device("cuda", 0)
    d_q = torch.from_numpy(q).to(dev)
    d_nq = torch.tensor([nq], dtype=torch.int32, device=dev)
    d_views = torch.from_numpy(views).to(dev)
    d_nv = torch.from_numpy(counts).to(dev)
    out = torch.zeros((3, cap, 3, 2), dtype=torch.int32, device=dev)
    m = knn.Matcher(cap)
    sel = [4, 0, 3]
    idx = (C.c_uint32 * 3)(*sel)
    L = _lib.lib()
    _lib.check(L.hm_knn_views_device(m.handle, d_q.data_ptr(), d_nq.data_ptr(), d_views.data_ptr(), d_nv.data_ptr(), cap,
                                     idx, 3, 3, out.data_ptr(), _lib.wait_handle(torch.cuda.current_stream())), "knn_views")
    _lib.check(L.hm_sync(m.handle), "hm_sync")
    got = out.cpu().numpy()
    for j, v in enumerate(sel):
        want = oracle.knn(q[:nq], views[v, :counts[v]], 3)
        _eq(got[j, :nq, :, 0].astype(np.uint32), want["index"], f"view {v} idx")
        _eq(got[j, :nq, :, 1].astype(np.uint32), want["distance"], f"view {v} dist")
    # best-of-views landmark selection on top of it (cv-sfm/src/lib.rs:1489-1532): few distinct landmarks, so the
    # same landmark shows up in several views and with equal distances
    landmarks = rng.integers(0, 900, (nviews, cap), dtype=np.uint32)
    d_lm = torch.from_numpy(landmarks.view(np.int32)).to(dev)
    d_best = torch.zeros((cap, 3, 2), dtype=torch.int32, device=dev)
    d_dec = torch.full((cap,), 7, dtype=torch.int32, device=dev)
    for better_by in (24, 1):
        _lib.check(L.hm_best_of_views_device(m.handle, out.data_ptr(), d_nq.data_ptr(), cap, idx, 3, 3, d_lm.data_ptr(),
                                             d_nv.data_ptr(), better_by, d_best.data_ptr(), d_dec.data_ptr(), None), "best_of_views")
        _lib.check(L.hm_sync(m.handle), "hm_sync")
        gnb = np.zeros((3, cap, 3), _lib.NB_DTYPE)
        gnb["index"] = got[..., 0]; gnb["distance"] = got[..., 1]
        wbest, wdec = oracle.best_of_views(gnb, nq, landmarks, sel, counts, better_by)
        _eq(d_best.cpu().numpy()[:nq].astype(np.uint32), wbest, f"best of views (better_by {better_by})")
        _eq(d_dec.cpu().numpy()[:nq].astype(np.uint32), wdec, f"decisions (better_by {better_by})")
        assert set(np.unique(wdec)) <= {0, 1, 2} and (wdec == 1).any()


@pytest.mark.gpu
def test_knn_batch_device(gpu, oracle):
    """hm_knn_batch_device: (query block, target block) problems of one call — the windows of several frames at once —
    each equal to the oracle's knn of that pair; blocks repeat as queries and as targets, one block is empty."""
    import ctypes as C
    import torch
    _, knn = gpu
    from cv_amd import _lib
    rng = np.random.default_rng(78)
    cap = 512
    qn = np.array([512, 77, 0, 300], np.int32)
    tn = np.array([1, 512, 200, 0, 45, 333], np.int32)
    qs = np.zeros((len(qn), cap, 64), np.uint8)
    ts = np.zeros((len(tn), cap, 64), np.uint8)
    for b in range(len(qn)):
        qs[b, :qn[b]] = _rand_desc(rng, int(qn[b]))
    for b in range(len(tn)):
        ts[b, :tn[b]] = _rand_desc(rng, int(tn[b]))
    ts[1, :50] = qs[0, :50]                                    # exact hits and ties
    ts[2, :50] = qs[0, :50]
    dev = torch.device("cuda", 0)
    d_q, d_nq = torch.from_numpy(qs).to(dev), torch.from_numpy(qn).to(dev)
    d_t, d_nt = torch.from_numpy(ts).to(dev), torch.from_numpy(tn).to(dev)
    iq = [0, 0, 0, 1, 1, 2, 3, 3, 3, 0]
    it = [1, 2, 5, 1, 0, 1, 3, 4, 1, 1]
    L = _lib.lib()
    m = knn.Matcher(cap)
    for k in (2, 3):
        out = torch.full((len(iq), cap, k, 2), -1, dtype=torch.int32, device=dev)
        _lib.check(L.hm_knn_batch_device(m.handle, d_q.data_ptr(), d_nq.data_ptr(), d_t.data_ptr(), d_nt.data_ptr(), cap,
                                         (C.c_uint32 * len(iq))(*iq), (C.c_uint32 * len(it))(*it), len(iq), k, out.data_ptr(),
                                         _lib.wait_handle(torch.cuda.current_stream())), "knn_batch")
        _lib.check(L.hm_sync(m.handle), "hm_sync")
        got = out.cpu().numpy()
        for p, (a, b) in enumerate(zip(iq, it)):
            if qn[a] == 0:
                continue
            want = oracle.knn(qs[a, :qn[a]], ts[b, :tn[b]], k)
            absent = want["index"] == 0xFFFFFFFF                 # slots past the target count: {2^22 - 1, 1023} (akz.h, hm_knn)
            assert absent[:, min(k, tn[b]):].all() and not absent[:, :min(k, tn[b])].any()
            _eq(got[p, :qn[a], :, 0].astype(np.uint32), np.where(absent, (1 << 22) - 1, want["index"]), f"k {k} problem {p} idx")
            _eq(got[p, :qn[a], :, 1].astype(np.uint32), np.where(absent, 1023, want["distance"]), f"k {k} problem {p} dist")
    # a view call is the batch call with one query block
    o1 = torch.zeros((3, cap, 2, 2), dtype=torch.int32, device=dev)
    o2 = torch.zeros((3, cap, 2, 2), dtype=torch.int32, device=dev)
    sel = (C.c_uint32 * 3)(5, 1, 2)
    _lib.check(L.hm_knn_views_device(m.handle, d_q[3:].data_ptr(), d_nq[3:].data_ptr(), d_t.data_ptr(), d_nt.data_ptr(), cap,
                                     sel, 3, 2, o1.data_ptr(), None), "knn_views")
    _lib.check(L.hm_knn_batch_device(m.handle, d_q.data_ptr(), d_nq.data_ptr(), d_t.data_ptr(), d_nt.data_ptr(), cap,
                                     (C.c_uint32 * 3)(3, 3, 3), sel, 3, 2, o2.data_ptr(), None), "knn_batch")
    _lib.check(L.hm_sync(m.handle), "hm_sync")
    _eq(o1.cpu().numpy()[:, :300], o2.cpu().numpy()[:, :300], "views == batch")
    assert L.hm_knn_batch_device(m.handle, d_q.data_ptr(), d_nq.data_ptr(), d_t.data_ptr(), d_nt.data_ptr(), cap,
                                 sel, sel, 3, 4, o2.data_ptr(), None) == -1                       # AKZ_E_INVALID: k > 3


def test_registration_matching_of_a_micro_batch(gpu, oracle):
    """The registration path's matching for several new frames at once (cv-sfm/src/lib.rs:1462-1532): hm_knn_batch_device
    (k = 3, every frame against its own window of stored views) + hm_best_of_views_batch_device == the oracle's knn and
    best-of-views per frame, and == the single-frame entry points."""
    import ctypes as C
    import torch
    _, knn = gpu
    from cv_amd import _lib
    rng = np.random.default_rng(79)
    cap, F, V, NB, k = 384, 5, 4, 9, 3
    qn = np.array([384, 100, 0, 257, 31], np.int32)
    tn = rng.integers(3, cap + 1, NB).astype(np.int32)
    qs = np.zeros((F, cap, 64), np.uint8); ts = np.zeros((NB, cap, 64), np.uint8)
    for b in range(F):
        qs[b, :qn[b]] = _rand_desc(rng, int(qn[b]))
    for b in range(NB):
        ts[b, :tn[b]] = _rand_desc(rng, int(tn[b]))
    ts[2, :20] = qs[0, :20]; ts[5, :20] = qs[0, :20]                      # equal distances in two views
    landmarks = rng.integers(0, 700, (NB, cap), dtype=np.uint32)           # few distinct landmarks: repeats across views
    views = np.stack([rng.choice(NB, V, replace=False) for _ in range(F)]).astype(np.uint32)
    views[1] = [7, 0, 3, 8]
    ts[7, :10] = qs[1, :10]; tn[7] = max(tn[7], 10)                        # exact hits in one view only: unique matches
    dev = torch.device("cuda", 0)
    d_q, d_nq = torch.from_numpy(qs).to(dev), torch.from_numpy(qn).to(dev)
    d_t, d_nt = torch.from_numpy(ts).to(dev), torch.from_numpy(tn).to(dev)
    d_lm = torch.from_numpy(landmarks.view(np.int32)).to(dev)
    iq = np.repeat(np.arange(F, dtype=np.uint32), V)
    it = views.reshape(-1)
    d_knn = torch.zeros((F, V, cap, k, 2), dtype=torch.int32, device=dev)
    d_best = torch.zeros((F, cap, 3, 2), dtype=torch.int32, device=dev)
    d_dec = torch.full((F, cap), 9, dtype=torch.int32, device=dev)
    L = _lib.lib()
    m = knn.Matcher(cap)
    u32p = lambda a: np.ascontiguousarray(a, np.uint32).ctypes.data_as(C.c_void_p)
    _lib.check(L.hm_knn_batch_device(m.handle, d_q.data_ptr(), d_nq.data_ptr(), d_t.data_ptr(), d_nt.data_ptr(), cap, u32p(iq), u32p(it),
                                     F * V, k, d_knn.data_ptr(), _lib.wait_handle(torch.cuda.current_stream())), "knn_batch")
    fr = np.arange(F, dtype=np.uint32)
    for better_by in (24, 1):
        _lib.check(L.hm_best_of_views_batch_device(m.handle, d_knn.data_ptr(), d_nq.data_ptr(), u32p(fr), cap, u32p(views), F, V, k,
                                                   d_lm.data_ptr(), d_nt.data_ptr(), better_by, d_best.data_ptr(), d_dec.data_ptr(), None),
                   "best_of_views_batch")
        _lib.check(L.hm_sync(m.handle), "hm_sync")
        gk = d_knn.cpu().numpy(); gb = d_best.cpu().numpy(); gd = d_dec.cpu().numpy()
        some = 0
        for f in range(F):
            n = int(qn[f])
            if n == 0:
                assert (gd[f] == 9).all()                                  # an empty frame is left alone
                continue
            gnb = np.zeros((V, cap, k), _lib.NB_DTYPE)
            gnb["index"] = gk[f, ..., 0]; gnb["distance"] = gk[f, ..., 1]
            for v in range(V):
                want = oracle.knn(qs[f, :n], ts[views[f, v], :tn[views[f, v]]], k)
                _eq(gnb["index"][v, :n], want["index"], f"frame {f} view {v} idx")
                _eq(gnb["distance"][v, :n], want["distance"], f"frame {f} view {v} dist")
            wbest, wdec = oracle.best_of_views(gnb, n, landmarks, views[f], tn, better_by)
            _eq(gb[f, :n].astype(np.uint32), wbest, f"frame {f} best of views (better_by {better_by})")
            _eq(gd[f, :n].astype(np.uint32), wdec, f"frame {f} decisions (better_by {better_by})")
            some += int((wdec == 1).sum())
            # the single-frame entry on the same slab
            d_b1 = torch.zeros((cap, 3, 2), dtype=torch.int32, device=dev); d_d1 = torch.zeros((cap,), dtype=torch.int32, device=dev)
            _lib.check(L.hm_best_of_views_device(m.handle, d_knn[f].data_ptr(), d_nq[f:].data_ptr(), cap, u32p(views[f]), V, k,
                                                 d_lm.data_ptr(), d_nt.data_ptr(), better_by, d_b1.data_ptr(), d_d1.data_ptr(), None), "best")
            _lib.check(L.hm_sync(m.handle), "hm_sync")
            _eq(d_b1.cpu().numpy()[:n], gb[f, :n], "single == batch (best)")
            _eq(d_d1.cpu().numpy()[:n], gd[f, :n], "single == batch (decision)")
        assert some >= 10


def test_place_recognition_hash_and_search(gpu, oracle, kitti_golden):
    """hm_hash_bag / hm_hash_bag_device / hm_hash_knn == oracle/lsh_oracle.c (cv-sfm/src/lib.rs:672, :622-624):
    nearest-codeword bag hash over a 4096-word codebook with duplicate words (tie -> lowest index), an empty
    bag, the batched device-resident form on ragged frames, and the exact nearest-hash search with ties."""
    import ctypes as C
    import torch
    _, knn = gpu
    from cv_amd import _lib, lsh
    rng = np.random.default_rng(2024)
    cw = _rand_desc(rng, 4096)
    cw[100] = cw[7]
    cw[4095] = cw[7]
    hasher = lsh.HammingHasher.new_with_codewords(cw)
    assert hasher.hash_bytes == 512
    feats = _rand_desc(rng, 1500)
    feats[:40] = cw[rng.integers(0, 4096, 40)]          # exact hits, some on the duplicated word
    feats[40] = cw[7]
    h, words = hasher.hash_bag(feats, return_words=True)
    want_h, want_w = oracle.hash_bag(feats, cw)
    _eq(words["index"], want_w["index"], "word index")
    _eq(words["distance"], want_w["distance"], "word distance")
    _eq(h, want_h, "hash")
    assert words["index"][40] == 7
    _eq(hasher.hash_bag(np.zeros((0, 64), np.uint8)), np.zeros(512, np.uint8), "empty bag")
    # real descriptors
    for name in ("default_desc0", "default_desc14"):
        _eq(hasher.hash_bag(kitti_golden[name]), oracle.hash_bag(kitti_golden[name], cw)[0], name + " hash")
    # batched, device-resident, ragged
    cap, nf = 700, 5
    counts = np.array([700, 0, 1, 333, 64], np.int32)
    blocks = np.zeros((nf, cap, 64), np.uint8)
    for f in range(nf):
        blocks[f, :counts[f]] = _rand_desc(rng, int(counts[f]))
        blocks[f, counts[f]:] = 0xA5                      # garbage past the count must not vote
    dev = torch.device("cuda", 0)
    d_blocks = torch.from_numpy(blocks).to(dev)
    d_counts = torch.from_numpy(counts).to(dev)
    d_cw = torch.from_numpy(cw).to(dev)
    d_hash = torch.full((nf, 512), 0xFF, dtype=torch.uint8, device=dev)
    d_words = torch.zeros((nf, cap, 2), dtype=torch.int32, device=dev)
    m = knn.Matcher(4096)
    L = _lib.lib()
    for _ in range(2):                                    # the second call reuses the staging ring
        _lib.check(L.hm_hash_bag_device(m.handle, d_blocks.data_ptr(), d_counts.data_ptr(), cap, nf, d_cw.data_ptr(), 4096,
                                        d_hash.data_ptr(), d_words.data_ptr(), _lib.wait_handle(torch.cuda.current_stream())),
                   "hash_bag_device")
    _lib.check(L.hm_sync(m.handle), "hm_sync")
    got_h, got_w = d_hash.cpu().numpy(), d_words.cpu().numpy()
    hashes = []
    for f in range(nf):
        wh, ww = oracle.hash_bag(blocks[f, :counts[f]], cw)
        _eq(got_h[f], wh, f"frame {f} hash")
        _eq(got_w[f, :counts[f], 0].astype(np.uint32), ww["index"], f"frame {f} words")
        hashes.append(wh)
    assert not got_h[1].any()
    # frame search: exact (distance, insertion order)
    index = lsh.HashIndex(512)
    store = [rng.integers(0, 256, 512, dtype=np.uint8) for _ in range(300)] + hashes
    store[17] = store[5].copy()
    for i, hsh in enumerate(store):
        index.insert(hsh, f"frame{i}")
    for qi, k in ((5, 10), (302, 512), (0, 1)):
        got = index.knn_values(store[qi], k)
        want = oracle.hash_knn(store[qi], np.stack(store), k)
        assert len(got) == len(want) == min(k, len(store))
        _eq(np.array([g[0][0] for g in got], np.uint32), want["index"], "hash knn index")
        _eq(np.array([g[0][1] for g in got], np.uint32), want["distance"], "hash knn distance")
        assert got[0][1] == f"frame{want['index'][0]}"
    assert [g[0][0] for g in index.knn_values(store[5], 2)] == [5, 17]


def test_bicubic_colour_sampling(gpu, oracle, kitti):
    """akz_sample_colors_rgb8 == oracle restatement of cv-sfm/src/bicubic.rs on real keypoints plus positions on
    and across the image border (default colour)."""
    akaze, _ = gpu
    img = kitti[0]
    rng = np.random.default_rng(3)
    rgb = np.stack([img, np.roll(img, 7, 1), 255 - img], axis=2)
    rgb = np.ascontiguousarray(rgb ^ rng.integers(0, 32, rgb.shape, dtype=np.uint8))
    ak = akaze.Akaze.sparse()
    ctx = ak.context(img.shape[1], img.shape[0], 1)
    (kp, _), = ctx.extract_batch([img])
    extra = np.zeros(64, kp.dtype)
    extra["x"] = rng.uniform(-4, img.shape[1] + 4, 64).astype(np.float32)
    extra["y"] = rng.uniform(-4, img.shape[0] + 4, 64).astype(np.float32)
    kps = np.concatenate([kp, extra])
    got, want = ctx.sample_colors(rgb, kps), oracle.sample_colors_rgb8(rgb, kps)
    _eq(got, want, "bicubic colours")
    assert len(kp) > 100 and (want[:len(kp)].max() > 0)


def test_matcher_properties_at_full_size(gpu):
    """BASELINE config 3 sizes (two 1080p frames, ~5 000 descriptors each), checked through properties that do
    not need the oracle: every reported distance is the popcount of the XOR with the reported index, the first
    neighbour is a true minimum with the lowest index among ties, and symmetric matching is symmetric."""
    akaze, knn = gpu
    ak = akaze.Akaze.default()
    img = synth_frame(1920, 1080, 31, n_rect=200, n_disc=200)
    a = ak.extract_arrays(img)[1]
    b = ak.extract_arrays(np.ascontiguousarray(np.roll(img, (2, 3), axis=(0, 1))))[1]   # the same scene, shifted
    assert len(a) > 3000 and len(b) > 3000
    m = knn.Matcher(8192)
    nn = m.knn(a, b, 3)
    pop = np.unpackbits(a[:, None, :] ^ b[nn["index"].astype(np.int64)], axis=2).sum(axis=2)
    assert np.array_equal(pop.astype(np.uint32), nn["distance"])
    assert (np.diff(nn["distance"].astype(np.int64), axis=1) >= 0).all()
    rows = np.random.default_rng(0).choice(len(a), 64, replace=False)        # exhaustive check on a sample
    full = np.unpackbits(a[rows, None, :] ^ b[None, :, :], axis=2).sum(axis=2)
    assert np.array_equal(full.min(axis=1).astype(np.uint32), nn["distance"][rows, 0])
    assert np.array_equal(full.argmin(axis=1).astype(np.uint32), nn["index"][rows, 0])   # argmin = lowest index
    ab = knn.symmetric_matching(a, b)
    ba = knn.symmetric_matching(b, a)
    assert len(ab) > 100 and sorted((x, y) for x, y in ab) == sorted((y, x) for x, y in ba)


@pytest.mark.parametrize("switch", ["valu", "int8", "fp4_regs"])
def test_alternative_matcher_kernels(gpu, oracle, switch):
    """HM_OPT_NO_MFMA: the xor/popcount kernel (the reference implementation of the MFMA ones, k = 2);
    HM_OPT_NO_FP4: the int8 MFMA kernel (k = 1..3); HM_OPT_NO_LDS_DMA: the FP4 kernel with its register stage (k = 1..3).
    The default is the FP4 MFMA kernel fed by LDS-DMA, which every other matcher test exercises."""
    _, knn = gpu
    rng = np.random.default_rng(5)
    m = knn.Matcher(4096, kernel=switch)
    q = _rand_desc(rng, 700); t = _rand_desc(rng, 1300)
    t[rng.integers(0, 1300, 300)] = t[rng.integers(0, 1300, 300)]
    got, want = m.knn2(q, t), oracle.knn2(q, t)
    _eq(got["index"], want["index"], switch + " knn idx")
    _eq(got["distance"], want["distance"], switch + " knn dist")
    assert m.match(q, t).tolist() == oracle.match(q, t).tolist()
    if switch in ("int8", "fp4_regs"):
        for k in (1, 3):
            got, want = m.knn(q, t[:1001], k), oracle.knn(q, t[:1001], k)
            _eq(got["index"], want["index"], f"{switch} knn{k} idx")
            _eq(got["distance"], want["distance"], f"{switch} knn{k} dist")
    m.close()


def test_matching_rules(gpu, oracle):
    _, knn = gpu
    rng = np.random.default_rng(22)
    a = _rand_desc(rng, 900)
    b = a[rng.permutation(900)[:700]].copy()
    flips = rng.random((700, 64 * 8)) < 0.04
    b ^= np.packbits(flips, axis=1)
    b[:, 61:] = 0; b[:, 60] &= 0x3F
    b = np.concatenate([b, _rand_desc(rng, 300)])
    m = knn.Matcher(4096)
    for rule, pu, pf in ((0, 24, 0.0), (1, 24, 0.0), (2, 0, 0.5), (0, 0, 0.0), (2, 0, 0.8)):
        for sym in (False, True):
            got = m.match(a, b, rule, pu, pf, sym)
            want = oracle.match(a, b, rule, pu, pf, sym)
            _eq(got, want, f"match rule={rule} sym={sym}")
            assert len(got) > 50


def test_estimate_pose_pipeline(gpu, kitti):
    """akaze/tests/estimate_pose.rs:24-59 restated against the host-side mirror API:
    Akaze::sparse().extract x2 -> 399 / 343 descriptors -> LinearKnn + Lowe 0.5 -> 11 matches."""
    akaze, knn = gpu
    kps1, ds1 = akaze.Akaze.sparse().extract(kitti[0])
    kps2, ds2 = akaze.Akaze.sparse().extract(kitti[1])
    assert len(ds1) == 399 and len(kps1) == 399
    assert len(ds2) == 343 and len(kps2) == 343
    matches = knn.match_descriptors(ds1, ds2, 0.5)
    assert len(matches) == 11
    # the same through the literal Knn trait surface
    lk = knn.LinearKnn(metric=knn.Hamming, iter=ds2)
    two = lk.knn(ds1[0], 2)
    assert len(two) == 2 and two[0].distance <= two[1].distance
    nn = lk.knn_batch(ds1)
    ok = [(i, int(nn[i, 0]["index"])) for i in range(len(ds1))
          if np.float32(nn[i, 0]["distance"]) < np.float32(nn[i, 1]["distance"]) * np.float32(0.5)]
    assert ok == matches


def test_tutorial_ch5_symmetric_matching(gpu, oracle, kitti, kitti_golden):
    """tutorial ch5 main.rs:28-33,154-200: Akaze::default() + symmetric better-by-24 matching."""
    akaze, knn = gpu
    ak = akaze.Akaze.default()
    _, da = ak.extract(kitti[0])
    _, db = ak.extract(kitti[1])
    got = knn.symmetric_matching(da, db)
    assert got == kitti_golden["default_sym24"].tolist()
    fwd = knn.matching(da, db)
    assert len(fwd) == len(da) and sum(x is not None for x in fwd) >= len(got)
    # cv-sfm variant (<=) and its <2 guard
    assert knn.symmetric_matching(da[:1], db, strict=False) == []
    got_le = knn.symmetric_matching(da, db, strict=False)
    assert got_le == oracle.match(da, db, rule=1, param_u=24, symmetric=True).tolist()


def test_cpp_host_mirror_estimate_pose(gpu, kitti, tmp_path):
    """The reference's integration test (akaze/tests/estimate_pose.rs:24-76: extract, match, calibrate, Arrsac + EightPoint
    consensus -> 11 inliers) restated in C++ against include/akaze.hpp — the twin of the Rust shim — and run as a separate
    native process linked to libakz.so; plus the context cache, list growth and the colour arm from a native caller."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "estimate_pose"
    lib_dir = os.path.join(root, "cv_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "estimate_pose.cpp"), "-o", str(exe),
                           "-L", lib_dir, "-lakz", f"-Wl,-rpath,{lib_dir}"])
    f0, f1 = tmp_path / "f0.raw", tmp_path / "f14.raw"
    kitti[0].tofile(f0); kitti[1].tofile(f1)
    h, w = kitti[0].shape
    r = subprocess.run([str(exe), str(f0), str(f1), str(w), str(h)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "descriptors 399 343" in r.stdout and "matches 11" in r.stdout
    assert "inliers 11" in r.stdout and "grown 399" in r.stdout and "colour 399" in r.stdout and "estimate_pose ok" in r.stdout
    assert "resampled consensus inliers" in r.stdout     # the bare Arrsac constructor with the crate's defaults (re-sampling on)


def test_native_host_drives_the_device_pipeline(gpu, kitti, tmp_path):
    """INTEGRATION.md 4 from a native process with no Python and no PyTorch in it (tests/cpp/device_pipeline.cpp): hipMalloc'd
    buffers, akz_extract_batch_device -> hm_match_batch_device -> rs_essential_arrsac_batch_device chained by their streams,
    nothing copied back in between: 399 / 343 descriptors, 11 matches, 11 inliers, and the device API's keypoints and
    descriptors byte-equal to the host API's."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "device_pipeline"
    lib_dir = os.path.join(root, "cv_amd", "lib")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(rocm, "include"),
                           "-I", os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "device_pipeline.cpp"),
                           "-o", str(exe), "-L", lib_dir, "-lakz", "-L", os.path.join(rocm, "lib"), "-lamdhip64",
                           f"-Wl,-rpath,{lib_dir}", f"-Wl,-rpath,{os.path.join(rocm, 'lib')}"])
    f0, f1 = tmp_path / "f0.raw", tmp_path / "f14.raw"
    kitti[0].tofile(f0); kitti[1].tofile(f1)
    h, w = kitti[0].shape
    r = subprocess.run([str(exe), str(f0), str(f1), str(w), str(h)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "descriptors 399 343" in r.stdout and "matches 11" in r.stdout and "inliers 11" in r.stdout
    assert "host api == device api" in r.stdout and "device_pipeline ok" in r.stdout


# ---------------------------------------------------------------------------------------------
def _two_view_scene(rng, n, outlier_frac):
    """the scene of eight-point/tests/random.rs:38-75 with uniformly random outlier
    bearings mixed in."""
    from test_oracle_ransac import _rot
    R = _rot(rng.random(3) * np.pi * 2 * 0.2)
    t = rng.random(3)
    pts = rng.random((n, 3)) * 2.0
    pts[:, 0] -= 1.0; pts[:, 1] -= 1.0; pts[:, 2] += 3.0
    pb = pts @ R.T + t
    a = pts / np.linalg.norm(pts, axis=1, keepdims=True)
    b = pb / np.linalg.norm(pb, axis=1, keepdims=True)
    bad = rng.random(n) < outlier_frac
    rb = rng.standard_normal((n, 3)); rb[:, 2] = np.abs(rb[:, 2]) + 0.5
    b[bad] = (rb / np.linalg.norm(rb, axis=1, keepdims=True))[bad]
    return a, b


def test_ransac_bit_exact(gpu, oracle):
    """R1-R4: per-(hypothesis, pose) inlier counts, the winning pose (all 12 f64 bit patterns), its id and
    its inlier index set equal the oracle's."""
    from cv_amd.ransac import EssentialConsensus
    rng = np.random.default_rng(0x5AC)
    cons = EssentialConsensus(2048, 4096)
    for n, n_hyp, frac, thr in ((200, 300, 0.3, 1e-7), (64, 500, 0.0, 1e-7), (1000, 64, 0.3, 1e-4), (8, 1, 0.0, 0.1)):
        a, b = _two_view_scene(rng, n, frac)
        samples = np.stack([rng.choice(n, 8, replace=False) for _ in range(n_hyp)]).astype(np.uint32)
        got = cons.model_inliers(a, b, samples, thr)
        want = oracle.essential_batch(a, b, samples, thr)
        assert (got is None) == (want is None)
        if want is None:
            continue
        wpose, wbest, winl, wcounts = want
        _eq(cons.counts(n_hyp), wcounts, f"ransac counts n={n} hyp={n_hyp}")
        pose, inl, best = got
        assert best == wbest
        _eq(pose, wpose, "ransac best pose")
        _eq(inl, winl, "ransac inliers")
        if frac > 0 and thr < 1e-5 and n >= 200:
            assert len(inl) > 0.5 * n


def test_far_pair_rejection_never_changes_a_count(gpu, oracle):
    """Before the 4x4 eigen-decomposition the device discards (pose, match) pairs whose residual provably exceeds the
    threshold (rs_pair_far: the rays' angle to each other's epipolar plane bounds the residual from below).  Counts of
    every (hypothesis, pose), winner and inlier list against the oracle, which knows no such shortcut, for thresholds
    from 1e-12 to 0.5 — through the decade where the bound and the threshold meet for most pairs — on a noisy scene, on
    non-unit bearings (the shortcut must stand aside) and on degenerate geometry (rays along the baseline)."""
    from cv_amd.ransac import EssentialConsensus
    rng = np.random.default_rng(0xFA12)
    cons = EssentialConsensus(2048, 4096)
    n, n_hyp = 300, 96
    a, b = _two_view_scene(rng, n, 0.3)
    b = b + rng.standard_normal(b.shape) * 3e-3
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    samples = np.stack([rng.choice(n, 8, replace=False) for _ in range(n_hyp)]).astype(np.uint32)
    scenes = [("noisy", a, b)]
    scenes.append(("non-unit bearings", a * 1.0000001, b * (1.0 + 1e-12)))
    a2 = a.copy(); b2 = b.copy()
    a2[::5] = a2[0]; b2[::5] = b2[0]                     # repeated matches: rank-deficient samples, rays along one direction
    scenes.append(("degenerate", a2, b2))
    checked = 0
    for name, sa, sb in scenes:
        for thr in (1e-12, 1e-9, 1e-7, 3e-6, 1e-4, 2e-3, 0.05, 0.2, 0.5):
            want = oracle.essential_batch(sa, sb, samples, thr)
            got = cons.model_inliers(sa, sb, samples, thr)
            assert (got is None) == (want is None), (name, thr)
            if want is None:
                continue
            _eq(cons.counts(n_hyp), want[3], f"{name}: counts at thr {thr}")
            assert got[2] == want[1], (name, thr)
            _eq(got[0], want[0], f"{name}: pose at thr {thr}")
            _eq(got[1], want[2], f"{name}: inliers at thr {thr}")
            checked += 1
    assert checked >= 20


def test_far_bound_is_a_lower_bound_of_the_device_residual(gpu):
    """rs_pair_far against the residuals the device itself evaluates (rs_debug_residuals, no shortcut): wherever the bound
    says "far" at threshold T, the residual of [R | t] AND of its mirror [R | -t] is >= T — for random rotations and
    baselines, matches from exact to unrelated, rays along the baseline, thresholds over nine decades; the bound must also
    do its job (most unrelated pairs are far at small thresholds) and stand aside for a matrix that is not a rotation and
    for bearings that are not unit vectors."""
    from cv_amd.ransac import EssentialConsensus
    from test_oracle_ransac import _rot
    rng = np.random.default_rng(0xB0D)
    cons = EssentialConsensus(2048, 64)
    n, n_pose = 1500, 24
    poses = np.zeros((n_pose, 3, 4))
    for p in range(n_pose):
        poses[p, :, :3] = _rot(rng.standard_normal(3) * rng.uniform(0.01, 1.5))
        poses[p, :, 3] = rng.standard_normal(3) * rng.uniform(0.01, 2.0)
    a = rng.standard_normal((n, 3)); a[:, 2] = np.abs(a[:, 2]) + 0.3
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = rng.standard_normal((n, 3)); b[:, 2] = np.abs(b[:, 2]) + 0.3
    # a third of the matches agree with pose 0 up to noise of three magnitudes, a few rays point along its baseline
    R0, t0 = poses[0, :, :3], poses[0, :, 3]
    k = n // 3
    X = a[:k] * rng.uniform(2, 9, (k, 1))
    b[:k] = X @ R0.T + t0 + rng.standard_normal((k, 3)) * rng.choice([0.0, 1e-4, 1e-2], (k, 1))
    b[k:k + 20] = t0 + rng.standard_normal((20, 3)) * 1e-3
    a[k:k + 20] = (R0.T @ t0) + rng.standard_normal((20, 3)) * 1e-3
    a /= np.linalg.norm(a, axis=1, keepdims=True); b /= np.linalg.norm(b, axis=1, keepdims=True)
    res = cons.residuals(poses, a, b, paired=True)              # [pose, {t, -t}, match]
    lo = res.min(axis=1)
    for thr in (1e-12, 1e-9, 1e-7, 1e-5, 1e-3, 0.03, 0.1, 0.12, 0.5):
        far = cons.far(poses, a, b, thr)
        assert (lo[far] >= thr).all(), (thr, float(lo[far].min()))
        if thr <= 1e-5:
            assert far[1:].mean() > 0.9                             # unrelated poses: nearly every pair is ruled out ...
            assert not far[0, :k][lo[0, :k] < thr].any()            # ... and no inlier of the true pose ever is
        if thr >= 0.1225:
            assert not far.any()                                    # beyond what the bound can prove
    # not a rotation (scaled, sheared): the bound must not be used
    bad = poses.copy(); bad[:, :, :3] *= 1.0 + 1e-6; bad[3, 0, 1] += 0.2
    assert not cons.far(bad, a, b, 1e-7).any()
    # bearings that are not unit vectors
    assert not cons.far(poses, a * (1 + 1e-9), b, 1e-7).any() and not cons.far(poses, a, b * 0.5, 1e-7).any()


def test_ransac_estimate_pose_pin(gpu, kitti, oracle):
    """akaze/tests/estimate_pose.rs:24-76 end to end on the device: 399/343 descriptors -> 11 matches ->
    calibrate with K_00 -> consensus at 0.1 -> 11 inliers."""
    import itertools
    akaze, knn = gpu
    from cv_amd.ransac import CameraIntrinsics, EssentialConsensus
    kp1, ds1 = akaze.Akaze.sparse().extract_arrays(kitti[0])
    kp2, ds2 = akaze.Akaze.sparse().extract_arrays(kitti[1])
    m = np.array(knn.match_descriptors(ds1, ds2, 0.5))
    assert len(m) == 11
    cam = CameraIntrinsics((9.842439e2, 9.808141e2), (6.9e2, 2.331966e2), 0.0)
    a = cam.calibrate(kp1[m[:, 0]]); b = cam.calibrate(kp2[m[:, 1]])
    _eq(a, oracle.calibrate(kp1[m[:, 0]], 984.2439, 980.8141, 690.0, 233.1966), "calibrate")
    samples = np.array(list(itertools.combinations(range(11), 8)), np.uint32)
    pose, inl, best = EssentialConsensus(64, 256).model_inliers(a, b, samples, 0.1)
    assert len(inl) == 11
    wpose, wbest, winl, _ = oracle.essential_batch(a, b, samples, 0.1)
    assert best == wbest
    _eq(pose, wpose, "kitti pose")


def test_p3p_bit_exact(gpu, oracle):
    """R5: Lambda Twist hypotheses + WorldToCamera residual consensus: counts, winning pose bits, id and inlier
    set equal the oracle's, on the reference's two consensus scenes and on a random scene with outliers."""
    import itertools
    from cv_amd.ransac import EssentialConsensus
    from test_oracle_ransac import arrsac_manual_scene, endless_loop_scene, _projective, _rot
    cons = EssentialConsensus(2048, 4096)
    R, t, b, w = arrsac_manual_scene()
    scenes = [(b, w, np.array(list(itertools.permutations(range(5), 3)), np.uint32), 0.01)]
    b2, w2 = endless_loop_scene()
    scenes.append((b2, w2, np.array(list(itertools.combinations(range(9), 3)), np.uint32), 0.01))
    rng = np.random.default_rng(77)
    n = 500
    Rr = _rot(rng.random(3) * 0.8); tr = rng.random(3)
    pts = rng.random((n, 3)) * 4.0 - 2.0
    pts[:, 2] += 6.0
    cam = pts @ Rr.T + tr
    bb = cam / np.linalg.norm(cam, axis=1, keepdims=True)
    bad = rng.random(n) < 0.3
    rb = rng.standard_normal((n, 3)); rb[:, 2] = np.abs(rb[:, 2]) + 0.5
    bb[bad] = (rb / np.linalg.norm(rb, axis=1, keepdims=True))[bad]
    scenes.append((bb, _projective(pts), np.stack([rng.choice(n, 3, replace=False) for _ in range(1000)]).astype(np.uint32), 1e-6))
    for b, w, samples, thr in scenes:
        got = cons.p3p_model_inliers(b, w, samples, thr)
        want = oracle.p3p_batch(b, w, samples, thr)
        assert (got is None) == (want is None)
        wpose, wbest, winl, wcounts = want
        _eq(cons.counts(len(samples)), wcounts, "p3p counts")
        pose, inl, best = got
        assert best == wbest
        _eq(pose, wpose, "p3p pose")
        _eq(inl, winl, "p3p inliers")
    assert len(inl) > 0.5 * n and np.abs(pose[:, :3] - Rr).max() < 1e-6


def test_p3p_inlier_test_is_exact_at_the_threshold(gpu, oracle):
    """The device decides `WorldToCamera::residual < thresh` on a cheap estimate and falls back to the exact statement
    near the threshold (rs_w2c_inlier).  Thresholds placed EXACTLY on residuals of the winning pose, one ulp above and
    one ulp below, negative homogeneous weights, huge and tiny point scales: the inlier set is always the oracle's."""
    from cv_amd.ransac import EssentialConsensus
    from test_oracle_ransac import _projective, _rot
    rng = np.random.default_rng(0x7E57)
    n = 400
    Rr = _rot(rng.random(3) * 0.8); tr = rng.random(3)
    pts = rng.random((n, 3)) * 4.0 - 2.0
    pts[:, 2] += 6.0
    cam = pts @ Rr.T + tr
    bb = cam / np.linalg.norm(cam, axis=1, keepdims=True)
    bb += rng.standard_normal(bb.shape) * 1e-4
    bb /= np.linalg.norm(bb, axis=1, keepdims=True)
    world = _projective(pts)
    world[::3] *= -1.0                      # signbit(w) arm: the point is negated, not the result
    world[1::7] *= 1e-150                   # q . q far outside the ordinary range: the exact statement decides
    world[2::7] *= 1e120
    samples = np.stack([rng.choice(n, 3, replace=False) for _ in range(64)]).astype(np.uint32)
    cons = EssentialConsensus(1024, 1024)
    wpose, wbest, winl, _ = oracle.p3p_batch(bb, world, samples, 1e-6)
    res = np.array([oracle.w2c_residual(wpose, bb[i], world[i]) for i in range(n)])
    assert np.isfinite(res).all() and (res > 0).sum() > n // 2
    checked = 0
    for k in np.argsort(res)[n // 4::n // 16][:10]:
        for thr in (res[k], np.nextafter(res[k], np.inf), np.nextafter(res[k], -np.inf)):
            want = oracle.p3p_batch(bb, world, samples, float(thr))
            got = cons.p3p_model_inliers(bb, world, samples, float(thr))
            assert (got is None) == (want is None)
            if want is None:
                continue
            _eq(cons.counts(len(samples)), want[3], f"counts at thr {thr!r}")
            assert got[2] == want[1]
            _eq(got[1], want[2], f"inliers at thr {thr!r}")
            checked += 1
    assert checked >= 24


def test_block_scoring_pre_test_is_exact_at_the_threshold(gpu, oracle):
    """k_rsb_score_p3p (the ARRSAC-shaped loop's scoring kernel) decides most residuals on a FUSED estimate of [R | t] w,
    allowed only for certified rotations and away from deep cancellation, and hands everything else to rs_w2c_inlier.  rs_p3p_arrsac with nothing to prune scores every pose against every match in one block: the
    per-pose inlier counts must be the exhaustive oracle's (orc_p3p_batch) with thresholds EXACTLY on residuals of the
    winning pose, one ulp either side; negative homogeneous weights; point scales far outside the ordinary range; and world
    points within 1e-3 .. 1e-12 of the winning pose's camera centre (R w + t cancels: the guard's arm)."""
    import ctypes as C
    from cv_amd import _lib
    from cv_amd.ransac import EssentialConsensus
    from test_oracle_ransac import _projective, _rot
    L = _lib.lib()
    rng = np.random.default_rng(0xF05ED)
    n = 640
    Rr = _rot(rng.random(3) * 0.8); tr = rng.random(3)
    pts = rng.random((n, 3)) * 4.0 - 2.0
    pts[:, 2] += 6.0
    centre = -Rr.T @ tr
    near = np.arange(5, n, 9)
    pts[near] = centre + rng.standard_normal((len(near), 3)) * (10.0 ** -rng.integers(3, 13, len(near)))[:, None]
    cam = pts @ Rr.T + tr
    bb = cam / np.linalg.norm(cam, axis=1, keepdims=True)
    bb[near] = rng.standard_normal((len(near), 3)); bb[near] /= np.linalg.norm(bb[near], axis=1, keepdims=True)
    bb += rng.standard_normal(bb.shape) * 1e-4
    bb /= np.linalg.norm(bb, axis=1, keepdims=True)
    world = _projective(pts)
    world[::3] *= -1.0
    world[1::7] *= 1e-150
    world[2::7] *= 1e120
    good = np.setdiff1d(np.arange(n), near)
    samples = np.stack([rng.choice(good, 3, replace=False) for _ in range(96)]).astype(np.uint32)
    cons = EssentialConsensus(1024, 1024)
    wpose, wbest, winl, _ = oracle.p3p_batch(bb, world, samples, 1e-6)
    res = np.array([oracle.w2c_residual(wpose, bb[i], world[i]) for i in range(n)])
    assert np.isfinite(res[good]).all()
    picks = list(np.argsort(res)[n // 4::n // 16][:8]) + list(near[:4])
    checked = 0
    for k in picks:
        if not (np.isfinite(res[k]) and 0.0 < res[k] < 0.5):
            continue
        for thr in (res[k], np.nextafter(res[k], np.inf), np.nextafter(res[k], -np.inf)):
            want = oracle.p3p_batch(bb, world, samples, float(thr))
            prm = cons.make_params(float(thr), n_hypotheses=len(samples), seed=0, block_size=64, init_blocks=1, max_candidates=0,
                                   bound=False, sprt=False)
            pose = np.zeros((3, 4)); best = C.c_uint32(); inl = np.zeros(n, np.uint32); ninl = C.c_uint32()
            st = L.rs_p3p_arrsac(cons._h, bb.ctypes.data, world.ctypes.data, n, samples.ctypes.data, C.byref(prm), pose.ctypes.data,
                                 C.byref(best), inl.ctypes.data, n, C.byref(ninl), None)
            assert (st == 0) == (want is not None), (st, thr)
            if want is None:
                continue
            _eq(cons.counts(len(samples)), want[3], f"counts at thr {thr!r}")
            assert best.value == want[1]
            _eq(inl[:ninl.value], want[2], f"inliers at thr {thr!r}")
            checked += 1
    assert checked >= 24


def test_two_rank_path_matches_single_rank(gpu, tmp_path):
    """The N>1 path of bench.py end to end on ONE GPU: two ranks (gloo, both on cuda:0) shard 32 global frames
    g -> rank g % 2, pass their descriptor blocks one rank up the ring and match every frame against its
    predecessor; the per-global-frame keypoint counts AND the match pair lists (every [a, b] index pair, byte for
    byte) must equal the single-rank run over the same 32 frames."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    m1, m2 = tmp_path / "m1.npy", tmp_path / "m2.npy"
    common = ["--micro-batch", "8", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras"]
    env = dict(os.environ)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--frames", "32", "--dump-matches", str(m1)] + common,
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(root, "bench.py"),
                        "--gpus", "2", "--frames", "16", "--backend", "gloo", "--share-device",
                        "--dump-matches", str(m2)] + common,
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    a, b = np.load(m1), np.load(m2)
    assert a.shape == b.shape == (32, 2)
    assert np.array_equal(a, b), (a.T, b.T)
    assert a[:, 0].min() > 1000 and a[:, 1].min() > 500
    single = np.load(str(m1) + ".r0.npz")
    ranks = [np.load(str(m2) + f".r{r}.npz") for r in range(2)]
    for g in range(32):
        want, got = single[f"g{g}"], ranks[g % 2][f"g{g}"]
        assert want.shape == got.shape and np.array_equal(want, got), f"pairs of global frame {g} differ"
        assert len(want) == a[g, 1]


# ---------------------------------------------------------------------------------------------
# Round 2: the benchmarked configuration, non-default configurations and input arms under the oracle
def _oracle_frames(frames, cfg_kw=None, procs=8):
    """oracle.extract of every frame (a process pool: a 1080p frame takes the C oracle about a second)."""
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(min(procs, len(frames))) as pool:
        return pool.map(_oracle_one, [(f, cfg_kw or {}) for f in frames], chunksize=1)


def _oracle_one(args):
    img, kw = args
    from oracle import oracle as O
    cfg = O.default_config()
    for k, v in kw.items():
        setattr(cfg, k, v)
    return O.Akaze(img.shape[1], img.shape[0], cfg).extract(img)


@pytest.mark.parametrize("resident", [False, True], ids=["call-size defaults", "octave 3 resident as in a 256-frame call"])
def test_benchmark_mode_full_hd_pipelined_vs_oracle(gpu, resident):
    """The configuration bench.py measures, checked against the ORACLE (not against the library's other API):
    default options (scratch-aliased Lsmooth/Lflow, no Ldet planes, two buffer sets, two streams), 1920x1080,
    an odd micro-batch (the last frame pair is half empty), three akz_extract_batch_device calls back to back with
    no synchronisation in between; keypoints and descriptor bytes of all nine frames equal oracle.extract's.
    A call of 256 frames hands octave 3 (240 x 135) to k_level_resident, a call of three does not by itself:
    `resident` asks for it (resident_min_frames = 1) so that the benchmark's kernel selection is the one under test."""
    import torch
    akaze, _ = gpu
    from cv_amd import _lib
    L = _lib.lib()
    W, H, B, CAP, NCALL = 1920, 1080, 3, 8192, 3
    dev = torch.device("cuda", 0)
    frames = [synth_frame(W, H, 7100 + i, n_rect=150, n_disc=150) for i in range(B * NCALL)]
    want = _oracle_frames(frames)
    d_frames = torch.from_numpy(np.stack(frames)).to(dev)
    ak = akaze.Akaze.default()
    ak.max_keypoints = CAP
    ctx = akaze.Context(ak, W, H, B, _opts(resident_min_frames=1) if resident else None)
    kps = torch.zeros((NCALL, B, CAP, 28), dtype=torch.uint8, device=dev)
    descs = torch.zeros((NCALL, B, CAP, 64), dtype=torch.uint8, device=dev)
    cnt = torch.zeros((NCALL, B), dtype=torch.int32, device=dev)
    cur = torch.cuda.current_stream()
    for k in range(NCALL):
        _lib.check(L.akz_extract_batch_device(ctx.handle, d_frames[k * B:(k + 1) * B].data_ptr(), 0, B, W, H,
                                              kps[k].data_ptr(), descs[k].data_ptr(), CAP, cnt[k].data_ptr(),
                                              _lib.wait_handle(cur)), "extract")
    _lib.check(L.akz_sync(ctx.handle), "sync")
    kps, descs, cnt = kps.cpu().numpy(), descs.cpu().numpy(), cnt.cpu().numpy()
    for k in range(NCALL):
        for j in range(B):
            okp, od = want[k * B + j]
            n = int(cnt[k, j])
            assert n == len(okp) and n > 1000, (k, j, n, len(okp))
            _eq(descs[k, j, :n], od, f"call {k} frame {j} descriptors")
            assert kps[k, j, :n].tobytes() == okp.tobytes(), f"call {k} frame {j} keypoints"
    ctx.close()


def _stress_case(i, seed=2):
    rng = np.random.default_rng(seed * 7919 + i)
    w = int(rng.integers(40, 230)) * 4 if i % 3 else int(rng.integers(120, 900))
    h = int(rng.integers(100, 700))
    thr = float(rng.choice([0.01, 0.003, 0.001, 0.0003]))
    kind = i % 4
    if kind == 0:
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    else:
        img = synth_frame(w, h, seed=int(rng.integers(1 << 30)), n_rect=int(rng.integers(5, 60)), n_disc=int(rng.integers(5, 60)))
    if kind == 2:
        img = (img.astype(np.int16) // 16 * 16).astype(np.uint8)      # plateaus: exact ties
    return thr, img


def test_randomised_parity_slice(gpu):
    """48 cases of tools/stress_parity.py's sweep inside the suite: random sizes (widths divisible by 4 and not),
    contents (noise, shapes, quantised plateaus full of ties) and thresholds, default options, HIP vs oracle."""
    akaze, _ = gpu
    cases = [_stress_case(i) for i in range(48)]
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(8) as pool:
        want = pool.map(_oracle_one, [(img, {"detector_threshold": thr}) for thr, img in cases], chunksize=1)
    total = 0
    for i, (thr, img) in enumerate(cases):
        h, w = img.shape
        c = akaze.Context(akaze.Akaze.new(thr), w, h, 1)
        (kp, d), = c.extract_batch([img])
        c.close()
        okp, od = want[i]
        _kp_eq(kp, okp, f"case {i} ({w}x{h}, thr {thr})")
        _eq(d, od, f"case {i} descriptors")
        total += len(kp)
    assert total > 20000


NON_DEFAULT = [
    ("num_sublevels=2", dict(num_sublevels=2)),
    ("num_sublevels=3", dict(num_sublevels=3)),
    ("max_octave_evolution=2", dict(max_octave_evolution=2)),
    ("max_octave_evolution=3", dict(max_octave_evolution=3)),
    ("sublevels=5,octaves=5", dict(num_sublevels=5, max_octave_evolution=5)),
    ("contrast_percentile=0.5", dict(contrast_percentile=0.5)),
    ("contrast_factor_num_bins=128", dict(contrast_factor_num_bins=128)),
    ("derivative_factor=1.0", dict(derivative_factor=1.0)),
    ("derivative_factor=2.0", dict(derivative_factor=2.0)),          # sigma 5: the generic derivative kernels
    ("descriptor_pattern_size=8", dict(descriptor_pattern_size=8)),
    ("descriptor_pattern_size=12", dict(descriptor_pattern_size=12)),
    ("base_scale_offset=1.2", dict(base_scale_offset=1.2)),          # 7-tap level-0 blur: the dense filter path
    ("base_scale_offset=2.4", dict(base_scale_offset=2.4)),          # 11 taps
    ("base_scale_offset=1.9", dict(base_scale_offset=1.9)),          # still radius 4: the fused tile kernel
    ("maximum_features=50,threshold=0.003", dict(maximum_features=50, detector_threshold=0.003)),
]


@pytest.mark.parametrize("name,kw", NON_DEFAULT, ids=[n for n, _ in NON_DEFAULT])
def test_non_default_configurations(gpu, oracle, name, kw):
    """Every field of akaze::Akaze (akaze/src/lib.rs:109-142) away from its default: pyramid schedule
    (evolution.rs:46-58,80-126), contrast factor (contrast_factor.rs:16-64), derivative scale
    (detector_response.rs:11-13) and descriptor pattern (descriptors.rs:75-98), every buffer and stage vs the
    oracle, on a width divisible by 4 (two-frame kernels) and on a ragged one (one-frame kernels)."""
    akaze, _ = gpu
    for (w, h, seed) in ((640, 400, 5), (333, 251, 6)):
        img = synth_frame(w, h, seed=seed, n_rect=40, n_disc=40)
        ak = akaze.Akaze(**kw)
        cfg = oracle.default_config()
        for k, v in kw.items():
            setattr(cfg, k, v)
        kp, _ = _compare_pyramid(akaze, oracle, img, None, f"{name} {w}x{h}", ak=ak, ocfg=cfg)
        assert len(kp) > 20, (name, len(kp))


def test_configurations_the_library_refuses(gpu):
    """What akz_create answers AKZ_E_INVALID to instead of computing something else than the reference:
    a pyramid of more than 32 levels (the per-frame level tables), more than 510 histogram bins, zero values the
    reference would divide by, a descriptor pattern whose grids do not have (n+2)^2 cells."""
    akaze, _ = gpu
    from cv_amd import _lib
    for kw, wh in ((dict(num_sublevels=8, max_octave_evolution=5), (1920, 1080)),
                   (dict(contrast_factor_num_bins=511), (320, 240)),
                   (dict(num_sublevels=0), (320, 240)), (dict(max_octave_evolution=0), (320, 240)),
                   (dict(descriptor_channels=4), (320, 240)), (dict(descriptor_pattern_size=1), (320, 240)),
                   (dict(base_scale_offset=0.0), (320, 240))):
        with pytest.raises(_lib.AkzError) as ei:
            akaze.Context(akaze.Akaze(**kw), wh[0], wh[1], 1)
        assert ei.value.status == -1, (kw, ei.value.status)
    # 8 sublevels x 4 octaves = 32 levels is the largest pyramid and works
    ctx = akaze.Context(akaze.Akaze(num_sublevels=8), 640, 640, 1)     # every octave keeps min(w, h) >= 80: 8 sublevels each
    assert ctx.num_levels(640, 640) == 32
    ctx.close()


def test_luma16_input_arm(gpu, oracle):
    """GrayFloatImage::from_dynamic, ImageLuma16 arm (image.rs:57-66): v / 65535 on the device == oracle."""
    akaze, _ = gpu
    rng = np.random.default_rng(16)
    for (w, h) in ((320, 240), (333, 251)):
        img8 = synth_frame(w, h, seed=w, n_rect=30, n_disc=30)
        img = (img8.astype(np.uint16) * 257 + rng.integers(-120, 121, img8.shape)).clip(0, 65535).astype(np.uint16)
        kp, desc = akaze.Akaze.default().extract_arrays(img)
        okp, od = oracle.Akaze(w, h, oracle.default_config()).extract(img)
        _kp_eq(kp, okp, f"luma16 {w}x{h}")
        _eq(desc, od, f"luma16 {w}x{h} desc")
        assert len(kp) > 50
    # and through the literal ABI entry point
    import ctypes as C
    from cv_amd import _lib
    ctx = akaze.Akaze.default().context(333, 251)
    kps = np.zeros(8192, _lib.KP_DTYPE); descs = np.zeros((8192, 64), np.uint8); n = C.c_uint32()
    _lib.check(_lib.lib().akz_extract_gray_u16(ctx.handle, img.ctypes.data, 333, 251, 333, kps.ctypes.data,
                                               descs.ctypes.data, 8192, C.byref(n)), "akz_extract_gray_u16")
    assert n.value == len(okp) and kps[:n.value].tobytes() == okp.tobytes()


def test_two_contexts_with_different_options_in_one_process(gpu, oracle):
    """Options are per context (no process-global switches): a keep_all context and a default one side by side,
    used alternately, give the same outputs; only the former can tap Ldet."""
    akaze, _ = gpu
    from cv_amd import _lib
    img = synth_frame(480, 270, 91)
    ak = akaze.Akaze.default()
    a = akaze.Context(ak, 480, 270, 1, _opts(keep_all=True, frame_pairs=False, parallel_suppression=False, pipeline=False))
    b = akaze.Context(ak, 480, 270, 1)
    okp, od = oracle.Akaze(480, 270, oracle.default_config()).extract(img)
    for _ in range(2):
        for c in (a, b):
            (kp, d), = c.extract_batch([img])
            _kp_eq(kp, okp, "ctx")
            _eq(d, od, "ctx desc")
    assert a.level_buffer(0, 1, "Ldet", 480, 270).shape == (270, 480)
    with pytest.raises(_lib.AkzError):
        b.level_buffer(0, 1, "Ldet", 480, 270)
    a.close(); b.close()


def test_arrsac_shaped_consensus(gpu, oracle):
    """rs_essential_arrsac (row R4): (a) with the exact bound alone — block scoring, retirement of poses that cannot
    reach the best count — the winner, its pose bits and its inlier set equal EXHAUSTIVE scoring by the oracle, on
    samples drawn by the device sampler (== rs_arrsac_samples on the host) and on caller samples; (b) with the
    candidate cap and the SPRT test on (arrsac's parameters at vslam-sandbox/src/main.rs:112-117) the same winner
    comes out on the BASELINE configs[3] scene while most residuals are never evaluated."""
    from cv_amd.ransac import EssentialConsensus
    rng = np.random.default_rng(0x5AC)
    n, n_hyp, thr = 1000, 2000, 1e-7
    a, b = _two_view_scene(rng, n, 0.3)
    cons = EssentialConsensus(n, 8192)
    samples = cons.arrsac_samples(0, n, n_hyp)
    assert samples.shape == (n_hyp, 8) and samples.max() < n
    assert all(len(set(r)) == 8 for r in samples[:200].tolist())            # distinct indices per sample
    wpose, wbest, winl, wcounts = oracle.essential_batch(a, b, samples, thr)
    for kw in (dict(sample_idx=None), dict(sample_idx=samples)):
        for bs in (64, 250, 1000):
            pose, inl, best, st = cons.arrsac_model_inliers(a, b, thr, n_hypotheses=n_hyp, seed=0, block_size=bs,
                                                            max_candidates=0, bound=True, sprt=False, **kw)
            assert best == wbest, (bs, best, wbest)
            _eq(pose, wpose, "arrsac pose (bound only)")
            _eq(inl, winl, "arrsac inliers (bound only)")
            assert st["residuals_evaluated"] <= st["residuals_exhaustive"]
    # no pruning at all == rs_essential_batch
    pose, inl, best, st = cons.arrsac_model_inliers(a, b, thr, n_hypotheses=n_hyp, seed=0, max_candidates=0, bound=False, sprt=False)
    assert best == wbest and st["blocks"] == 1 and st["residuals_evaluated"] <= st["residuals_exhaustive"]
    _eq(cons.counts(n_hyp), wcounts, "unpruned counts")
    # arrsac's own shape: cap + SPRT
    pose, inl, best, st = cons.arrsac_model_inliers(a, b, thr, n_hypotheses=n_hyp, seed=0, block_size=64, init_blocks=4,
                                                    max_candidates=1024, bound=True, sprt=True)
    assert best == wbest
    _eq(pose, wpose, "arrsac pose (cap + SPRT)")
    _eq(inl, winl, "arrsac inliers (cap + SPRT)")
    assert st["residuals_evaluated"] < 0.35 * st["residuals_exhaustive"], st
    assert st["survivors"] <= 1024
    # a different seed draws different samples
    assert not np.array_equal(cons.arrsac_samples(1, n, 16), samples[:16])


def test_large_keypoint_lists_and_overflow_report(gpu, oracle):
    """Beyond 16384 keypoints per frame / extrema per level (the reference's lists are unbounded,
    akaze/src/lib.rs:169-171): a noise frame at Akaze::dense() has ~27 000 extrema on level 0 and ~40 000 keypoints.
    With max_keypoints = 65536 every sort takes its global-memory path and the result equals the oracle; with the
    default capacity the call fails with AKZ_E_INTERNAL and akz_last_overflow names the frame and what it needed."""
    akaze, _ = gpu
    from cv_amd import _lib
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (960, 1280), dtype=np.uint8)
    okp, od = oracle.Akaze(1280, 960, oracle.default_config(threshold=0.0001)).extract(img)
    assert len(okp) > 16384
    ak = akaze.Akaze.dense()
    ak.max_keypoints = 65536
    ctx = akaze.Context(ak, 1280, 960, 1)
    (kp, d), = ctx.extract_batch([img])                        # a single frame: the chip-wide rank sorts
    _kp_eq(kp, okp, "dense noise keypoints")
    _eq(d, od, "dense noise descriptors")
    ctx.close()
    ctx = akaze.Context(ak, 1280, 960, 9)
    for j, (kp, d) in enumerate(ctx.extract_batch([img] * 9)):  # a batch: one bitonic block per frame, global key scratch
        _kp_eq(kp, okp, f"dense noise keypoints, batch frame {j}")
        _eq(d, od, f"dense noise descriptors, batch frame {j}")
    ctx.close()
    small = akaze.Akaze.dense()
    small.max_keypoints = 16384
    ctx = akaze.Context(small, 1280, 960, 1)
    with pytest.raises(_lib.AkzError) as ei:
        ctx.extract_batch([img])
    assert ei.value.status == -7 and "frame 0" in str(ei.value), str(ei.value)
    ctx.close()


def test_p3p_arrsac_shaped_consensus(gpu, oracle):
    """rs_p3p_arrsac: the registration path's consensus (cv-sfm/src/lib.rs:1619-1622) in ARRSAC's shape.  With the
    exact bound alone the winner, pose bits and inlier set equal the oracle's exhaustive scoring of the same (device
    drawn) 3-match samples; with cap + SPRT the same winner on this scene at a fraction of the residuals."""
    from cv_amd.ransac import EssentialConsensus
    from test_oracle_ransac import _projective, _rot
    rng = np.random.default_rng(78)
    n, n_hyp, thr = 1000, 4000, 1e-6
    Rr = _rot(rng.random(3) * 0.8); tr = rng.random(3)
    pts = rng.random((n, 3)) * 4.0 - 2.0
    pts[:, 2] += 6.0
    cam = pts @ Rr.T + tr
    bb = cam / np.linalg.norm(cam, axis=1, keepdims=True)
    bad = rng.random(n) < 0.3
    rb = rng.standard_normal((n, 3)); rb[:, 2] = np.abs(rb[:, 2]) + 0.5
    bb[bad] = (rb / np.linalg.norm(rb, axis=1, keepdims=True))[bad]
    world = _projective(pts)
    cons = EssentialConsensus(n, 8192)
    samples = cons.arrsac_samples(3, n, n_hyp, 3)
    wpose, wbest, winl, _ = oracle.p3p_batch(bb, world, samples, thr)
    for kw in (dict(max_candidates=0, sprt=False), dict(max_candidates=1024, sprt=True)):
        pose, inl, best, st = cons.arrsac_model_inliers(bb, world, thr, n_hypotheses=n_hyp, seed=3, p3p=True, **kw)
        assert best == wbest, (kw, best, wbest)
        _eq(pose, wpose, "p3p arrsac pose")
        _eq(inl, winl, "p3p arrsac inliers")
    assert st["residuals_evaluated"] < 0.35 * st["residuals_exhaustive"], st
    assert len(inl) > 0.5 * n and np.abs(pose[:, :3] - Rr).max() < 1e-6


ARRSAC_RULES = [
    ("bound", dict(max_candidates=0, sprt=False)),
    ("cap", dict(max_candidates=96, sprt=False)),
    ("cap+sprt", dict(max_candidates=96, sprt=True)),
    ("cap+sprt+halve", dict(max_candidates=96, sprt=True, halve=True)),
    ("cap+resample", dict(max_candidates=32, sprt=False, estimations_per_block=16)),
    ("everything", dict(max_candidates=64, sprt=True, halve=True, estimations_per_block=24, sprt_delta=0.02, sprt_ratio=200.0)),
    ("resample only", dict(max_candidates=0, sprt=False, bound=False, estimations_per_block=8)),
]


@pytest.mark.parametrize("p3p", [False, True], ids=["eight-point", "p3p"])
@pytest.mark.parametrize("name,kw", ARRSAC_RULES, ids=[n for n, _ in ARRSAC_RULES])
def test_arrsac_equals_its_specification(gpu, oracle, name, kw, p3p):
    """rs_essential_arrsac / rs_p3p_arrsac against oracle/arrsac_oracle.c under every retirement rule and with
    inlier-guided re-sampling, on a noisy scene (where re-sampling changes the winner): the winner's id, the 12 f64 of
    its pose, its inlier list, the survivor count, the block count, the number of poses made and the number of
    residuals evaluated are all equal — i.e. every retirement decision and every re-sampled index was the same."""
    from cv_amd.ransac import EssentialConsensus
    from test_oracle_ransac import _projective, _rot
    rng = np.random.default_rng(0xA22 + int(p3p))
    n, n_hyp = 700, 300
    if p3p:
        Rr = _rot(rng.random(3) * 0.8); tr = rng.random(3)
        pts = rng.random((n, 3)) * 4.0 - 2.0
        pts[:, 2] += 6.0
        cam = pts @ Rr.T + tr
        a = cam / np.linalg.norm(cam, axis=1, keepdims=True)
        a = a + rng.standard_normal(a.shape) * 1e-3
        a /= np.linalg.norm(a, axis=1, keepdims=True)
        bad = rng.random(n) < 0.3
        rb = rng.standard_normal((n, 3)); rb[:, 2] = np.abs(rb[:, 2]) + 0.5
        a[bad] = (rb / np.linalg.norm(rb, axis=1, keepdims=True))[bad]
        b = _projective(pts)
        thr = 2e-3
    else:
        a, b = _two_view_scene(rng, n, 0.3)
        b = b + rng.standard_normal(b.shape) * 2e-3
        b /= np.linalg.norm(b, axis=1, keepdims=True)
        thr = 1e-4
    cons = EssentialConsensus(n, 1024)
    for bs, ib in ((64, 2), (100, 1)):
        want = oracle.arrsac(a, b, thr, n_hyp, seed=9, block_size=bs, init_blocks=ib, p3p=p3p, **kw)
        got = cons.arrsac_model_inliers(a, b, thr, n_hypotheses=n_hyp, seed=9, block_size=bs, init_blocks=ib, p3p=p3p, **kw)
        assert (got is None) == (want is None)
        wpose, winl, wbest, wst = want
        pose, inl, best, st = got
        assert best == wbest, (name, bs, best, wbest, st, wst)
        _eq(pose, wpose, f"arrsac pose ({name})")
        _eq(inl, winl, f"arrsac inliers ({name})")
        for key in ("survivors", "blocks", "poses", "residuals_evaluated"):
            assert st[key] == wst[key], (name, bs, key, st, wst)
        assert len(inl) > 0.4 * n


def _pixel_scene(rng, n_a, n_b, n_pairs, outlier_frac, cam, noise_px=0.3):
    """Keypoints of two frames (pixel coordinates, f32) and a match list between them: a random rigid motion seen
    through `cam` (fx, fy, cx, cy, skew, k1), `outlier_frac` of the pairs joined at random."""
    from test_oracle_ransac import _rot
    R = _rot((rng.random(3) - 0.5) * 0.3)
    t = (rng.random(3) - 0.5) * 0.6
    fx, fy, cx, cy, skew = cam[:5]

    def project(P):
        x, y = P[:, 0] / P[:, 2], P[:, 1] / P[:, 2]
        return np.stack([fx * x + skew * y + cx, fy * y + cy], 1)
    ka = np.zeros(n_a, _lib_kp()); kb = np.zeros(n_b, _lib_kp())
    pts = np.stack([rng.uniform(-2, 2, n_a), rng.uniform(-1.2, 1.2, n_a), rng.uniform(3, 9, n_a)], 1)
    pa = project(pts) + rng.standard_normal((n_a, 2)) * noise_px
    ka["x"], ka["y"] = pa[:, 0], pa[:, 1]
    # frame b: the same points (where they exist) in another order, the rest unrelated
    perm = rng.permutation(max(n_a, n_b))[:n_b] % n_a
    pb = project(pts[perm] @ R.T + t) + rng.standard_normal((n_b, 2)) * noise_px
    kb["x"], kb["y"] = pb[:, 0], pb[:, 1]
    ib = rng.choice(n_b, size=min(n_pairs, n_b), replace=False)
    ia = perm[ib].copy()
    bad = rng.random(len(ib)) < outlier_frac
    ia[bad] = rng.integers(0, n_a, bad.sum())
    o = np.argsort(ia, kind="stable")
    return ka, kb, np.stack([ia[o], ib[o]], 1).astype(np.uint32)


def _lib_kp():
    from cv_amd._lib import KP_DTYPE
    return KP_DTYPE


BATCH_RULES = [
    ("halving, 16-match blocks, shuffled", dict(block_size=16, init_blocks=1, max_candidates=64, halve=True), True),
    ("cap + sprt, 64-match blocks", dict(block_size=64, init_blocks=2, max_candidates=96, sprt=True), False),
    ("re-sampling, 40-match blocks, shuffled", dict(block_size=40, init_blocks=1, max_candidates=48, sprt=True, halve=True,
                                                    estimations_per_block=12), True),
    ("bound only, 100-match blocks", dict(block_size=100, init_blocks=1, max_candidates=0, sprt=False), True),
]


@pytest.mark.parametrize("name,kw,shuffle", BATCH_RULES, ids=[r[0] for r in BATCH_RULES])
def test_batched_two_view_verification_equals_its_specification(gpu, oracle, name, kw, shuffle):
    """rs_essential_arrsac_batch_device (cv-sfm/src/lib.rs:1385-1412 for every frame pair of a
    micro-batch): device-resident keypoints + the matcher's pair lists in, pose / inlier list / id per scene out, all
    scenes through one chain of launches.  Scenes of ragged sizes — empty, fewer than eight matches, exactly eight, a
    full block, the capacity — each equal to oracle/arrsac_oracle.c (orc_arrsac_pairs) in calibrated bearing bits,
    scoring order, winner id, pose bits, inlier list, survivors, blocks and residuals evaluated."""
    import torch
    from cv_amd import _lib
    from cv_amd.ransac import EssentialConsensus
    rng = np.random.default_rng(0xBA7C)
    cap, n_hyp = 512, 192
    cam_a = (984.2439, 980.8141, 690.0, 233.1966, 0.0, None)
    cam_b = (950.0, 955.0, 640.0, 250.0, 0.5, -0.05)       # the K1-distortion arm on the second view
    sizes = [300, 0, 5, 8, 17, 64, 512, 129, 400, 33, 16, 250]
    pairs = np.zeros((len(sizes), cap, 2), np.uint32)
    scenes = []
    for s, n in enumerate(sizes):
        scenes.append(_pixel_scene(rng, cap, cap, n, 0.3, cam_a))
        pairs[s, :n] = scenes[-1][2]
    # keypoint blocks are stored in reverse scene order, so the block index arrays matter
    ia = ib = [len(sizes) - 1 - s for s in range(len(sizes))]
    kps_a = np.stack([sc[0] for sc in scenes[::-1]]); kps_b = np.stack([sc[1] for sc in scenes[::-1]])
    npairs = np.array([len(sc[2]) for sc in scenes], np.uint32)
    dev = torch.device("cuda", 0)
    S = len(sizes)
    d_ka = torch.from_numpy(kps_a.view(np.uint8).reshape(S, cap, 28)).to(dev)
    d_kb = torch.from_numpy(kps_b.view(np.uint8).reshape(S, cap, 28)).to(dev)
    d_pairs = torch.from_numpy(pairs.view(np.int32)).to(dev)
    d_np = torch.from_numpy(npairs.view(np.int32)).to(dev)
    d_pose = torch.zeros((S, 12), dtype=torch.float64, device=dev)
    d_best = torch.zeros((S,), dtype=torch.int32, device=dev)
    d_inl = torch.zeros((S, cap), dtype=torch.int32, device=dev)
    d_ninl = torch.zeros((S,), dtype=torch.int32, device=dev)
    d_stats = torch.zeros((S, 32), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    blocks_max = (cap + kw["block_size"] - 1) // kw["block_size"]
    cons = EssentialConsensus(cap, n_hyp + kw.get("estimations_per_block", 0) * blocks_max)
    cons.reserve(S)
    thr = 2e-7
    prm = cons.make_params(thr, n_hypotheses=n_hyp, seed=77, **kw)
    ca, cb = cons.camera(cam_a), cons.camera(cam_b)
    for rep in range(2):      # twice: the second call runs over the first one's leftovers in the arena
        cons.model_inliers_batch_device(d_ka.data_ptr(), d_kb.data_ptr(), cap, ia, ib, d_pairs.data_ptr(), d_np.data_ptr(), ca, cb,
                                        prm, d_pose.data_ptr(), d_best.data_ptr(), d_inl.data_ptr(), d_ninl.data_ptr(),
                                        d_stats.data_ptr(), shuffle=shuffle)
    cons.sync()
    pose = d_pose.cpu().numpy(); best = d_best.cpu().numpy().view(np.uint32); inl = d_inl.cpu().numpy().view(np.uint32)
    ninl = d_ninl.cpu().numpy().view(np.uint32)
    st = d_stats.cpu().numpy().view(np.dtype([("poses", "<u4"), ("survivors", "<u4"), ("blocks", "<u4"), ("reserved", "<u4"),
                                               ("evaluated", "<u8"), ("exhaustive", "<u8")])).reshape(S)
    some_model = 0
    for s, (ka, kb, pr) in enumerate(scenes):
        want = oracle.arrsac_pairs(ka, kb, pr, cam_a, cam_b, thr, n_hyp, scene=s, shuffle=shuffle, seed=77, **kw)
        ga, gb, go = cons.scene(s, cap)
        _eq(ga, want["bearings_a"], f"scene {s} bearings a")
        _eq(gb, want["bearings_b"], f"scene {s} bearings b")
        if shuffle:
            _eq(go, want["order"], f"scene {s} scoring order")
        assert best[s] == want["best_id"], (name, s, len(pr), best[s], want["best_id"])
        assert ninl[s] == len(want["inliers"]), (name, s, ninl[s], len(want["inliers"]))
        if want["best_id"] == 0xFFFFFFFF:
            assert len(pr) < 8 or st["survivors"][s] == 0
            continue
        some_model += 1
        _eq(pose[s].reshape(3, 4), want["pose"], f"scene {s} pose ({name})")
        _eq(inl[s, :ninl[s]], want["inliers"], f"scene {s} inliers ({name})")
        for key, wkey in (("survivors", "survivors"), ("blocks", "blocks"), ("poses", "poses"), ("evaluated", "residuals_evaluated")):
            assert int(st[key][s]) == want["stats"][wkey], (name, s, key, st[s], want["stats"])
    assert some_model >= 8


@pytest.mark.gpu
def test_pair_lists_that_point_out_of_bounds_refuse_their_scene(gpu, oracle):
    """k_rsb_prepare reads keypoints / world points through the indices of a caller-provided device pair list.  A list
    with an entry outside its keypoint block (>= cap_per_img) or outside the world table (>= n_world) makes THAT scene
    end with "no model" (oracle.pairs_in_range: the specification's rule) — nothing is read out of bounds, and the
    scenes beside it are what they are without it."""
    import torch
    from cv_amd.ransac import EssentialConsensus
    from test_oracle_arrsac import _registration_scene
    rng = np.random.default_rng(0xB0D)
    cap, n_hyp = 256, 96
    cam = (984.2439, 980.8141, 690.0, 233.1966, 0.0, None)
    kw = dict(block_size=32, init_blocks=1, max_candidates=32, sprt=True)
    dev = torch.device("cuda", 0)
    S = 4
    scenes = [_pixel_scene(rng, cap, cap, 120, 0.3, cam) for _ in range(S)]
    pairs = np.zeros((S, cap, 2), np.uint32)
    for s in range(S):
        pairs[s, :120] = scenes[s][2]
    pairs[1, 37, 0] = cap                 # first index one past the block
    pairs[2, 5, 1] = 0xFFFFFFF0           # second index far outside
    kps_a = np.stack([sc[0] for sc in scenes]); kps_b = np.stack([sc[1] for sc in scenes])
    npairs = np.full(S, 120, np.uint32)
    d_ka = torch.from_numpy(kps_a.view(np.uint8).reshape(S, cap, 28)).to(dev)
    d_kb = torch.from_numpy(kps_b.view(np.uint8).reshape(S, cap, 28)).to(dev)
    d_pairs = torch.from_numpy(pairs.view(np.int32)).to(dev)
    d_np = torch.from_numpy(npairs.view(np.int32)).to(dev)
    d_pose = torch.zeros((S, 12), dtype=torch.float64, device=dev)
    d_best = torch.zeros((S,), dtype=torch.int32, device=dev)
    d_inl = torch.zeros((S, cap), dtype=torch.int32, device=dev)
    d_ninl = torch.full((S,), 7, dtype=torch.int32, device=dev)
    cons = EssentialConsensus(cap, n_hyp)
    cons.reserve(S)
    prm = cons.make_params(2e-7, n_hypotheses=n_hyp, seed=5, **kw)
    c = cons.camera(cam)
    ia = list(range(S))
    cons.model_inliers_batch_device(d_ka.data_ptr(), d_kb.data_ptr(), cap, ia, ia, d_pairs.data_ptr(), d_np.data_ptr(), c, c, prm,
                                    d_pose.data_ptr(), d_best.data_ptr(), d_inl.data_ptr(), d_ninl.data_ptr(), None, shuffle=True)
    cons.sync()
    best = d_best.cpu().numpy().view(np.uint32); ninl = d_ninl.cpu().numpy().view(np.uint32)
    for s in range(S):
        ok = oracle.pairs_in_range(pairs[s, :120], cap, cap)
        assert ok == (s in (0, 3))
        if not ok:
            assert best[s] == 0xFFFFFFFF and ninl[s] == 0, (s, best[s], ninl[s])
            continue
        want = oracle.arrsac_pairs(scenes[s][0], scenes[s][1], pairs[s, :120], cam, cam, 2e-7, n_hyp, scene=s, shuffle=True, seed=5, **kw)
        assert best[s] == want["best_id"] and ninl[s] == len(want["inliers"])
        _eq(d_pose.cpu().numpy()[s].reshape(3, 4), want["pose"], f"scene {s} pose")
    # the registration consensus: a world-point index >= n_world
    n_world = 300
    cam_r = (950.0, 955.0, 640.0, 250.0, 0.5, -0.05)
    reg = [_registration_scene(rng, cap, n_world, 90, 0.3, cam_r) for _ in range(2)]
    rp = np.zeros((2, cap, 2), np.uint32)
    for s in range(2):
        rp[s, :90] = reg[s][2]
        rp[s, :90, 1] += s * n_world
    rp[0, 11, 1] = 2 * n_world            # one past the table
    world_all = np.concatenate([reg[0][1], reg[1][1]])
    d_k = torch.from_numpy(np.stack([reg[0][0], reg[1][0]]).view(np.uint8).reshape(2, cap, 28)).to(dev)
    d_rp = torch.from_numpy(rp.view(np.int32)).to(dev)
    d_rn = torch.from_numpy(np.full(2, 90, np.int32)).to(dev)
    d_world = torch.from_numpy(world_all).to(dev)
    prm_r = cons.make_params(1e-6, n_hypotheses=n_hyp, seed=9, **kw)
    cons.p3p_model_inliers_batch_device(d_k.data_ptr(), cap, [0, 1], d_rp.data_ptr(), d_rn.data_ptr(), d_world.data_ptr(),
                                        2 * n_world, cons.camera(cam_r), prm_r, d_pose.data_ptr(), d_best.data_ptr(),
                                        d_inl.data_ptr(), d_ninl.data_ptr(), None, shuffle=False)
    cons.sync()
    best = d_best.cpu().numpy().view(np.uint32); ninl = d_ninl.cpu().numpy().view(np.uint32)
    assert not oracle.pairs_in_range(rp[0, :90], cap, 2 * n_world) and oracle.pairs_in_range(rp[1, :90], cap, 2 * n_world)
    assert best[0] == 0xFFFFFFFF and ninl[0] == 0
    want = oracle.p3p_arrsac_pairs(reg[1][0], rp[1, :90], world_all, cam_r, 1e-6, n_hyp, scene=1, shuffle=False, seed=9, **kw)
    assert best[1] == want["best_id"] and ninl[1] == len(want["inliers"]) and want["best_id"] != 0xFFFFFFFF
    _eq(d_pose.cpu().numpy()[1].reshape(3, 4), want["pose"], "registration scene 1 pose")


REG_RULES = [
    ("halving, 16-match blocks, shuffled", dict(block_size=16, init_blocks=1, max_candidates=64, halve=True), True),
    ("cap + sprt, 64-match blocks", dict(block_size=64, init_blocks=2, max_candidates=96, sprt=True), False),
    ("re-sampling, 40-match blocks, shuffled", dict(block_size=40, init_blocks=1, max_candidates=48, sprt=True, halve=True,
                                                    estimations_per_block=12), True),
    ("bound only, 100-match blocks", dict(block_size=100, init_blocks=1, max_candidates=0, sprt=False), False),
]


@pytest.mark.parametrize("name,kw,shuffle", REG_RULES, ids=[r[0] for r in REG_RULES])
def test_batched_registration_consensus_equals_its_specification(gpu, oracle, name, kw, shuffle):
    """rs_p3p_arrsac_batch_device (cv-sfm/src/lib.rs:1571-1622 for every new frame of a micro-batch):
    device-resident keypoints, (feature, world point) pair lists and the world-point table in, WorldToCamera pose /
    inlier list / id per scene out.  Ragged scenes — empty, two matches, exactly three, a full block, the capacity — each
    equal to oracle/arrsac_oracle.c (orc_p3p_arrsac_pairs) in bearing and point bits, scoring order, winner id, pose
    bits, inlier list, survivors, blocks and residuals evaluated."""
    import torch
    from cv_amd.ransac import EssentialConsensus
    from test_oracle_arrsac import _registration_scene
    rng = np.random.default_rng(0x9E6)
    cap, n_hyp, n_world = 512, 160, 700
    cam = (950.0, 955.0, 640.0, 250.0, 0.5, -0.05)       # skew + the K1-distortion arm
    sizes = [300, 0, 2, 3, 17, 64, 512, 129, 400, 4, 16, 250]
    S = len(sizes)
    pairs = np.zeros((S, cap, 2), np.uint32)
    scenes = []
    worlds = []
    for s, n in enumerate(sizes):
        kps, world, pr, R, t, good = _registration_scene(rng, cap, n_world, n, 0.3, cam)
        pr = pr.copy(); pr[:, 1] += s * n_world            # one table for all scenes
        scenes.append((kps, pr, R, t))
        worlds.append(world)
        pairs[s, :n] = pr
    world_all = np.concatenate(worlds)
    ik = [S - 1 - s for s in range(S)]                      # keypoint blocks stored in reverse scene order
    kps_all = np.stack([sc[0] for sc in scenes[::-1]])
    npairs = np.array(sizes, np.uint32)
    dev = torch.device("cuda", 0)
    d_k = torch.from_numpy(kps_all.view(np.uint8).reshape(S, cap, 28)).to(dev)
    d_pairs = torch.from_numpy(pairs.view(np.int32)).to(dev)
    d_np = torch.from_numpy(npairs.view(np.int32)).to(dev)
    d_world = torch.from_numpy(world_all).to(dev)
    d_pose = torch.zeros((S, 12), dtype=torch.float64, device=dev)
    d_best = torch.zeros((S,), dtype=torch.int32, device=dev)
    d_inl = torch.zeros((S, cap), dtype=torch.int32, device=dev)
    d_ninl = torch.zeros((S,), dtype=torch.int32, device=dev)
    d_stats = torch.zeros((S, 32), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    blocks_max = (cap + kw["block_size"] - 1) // kw["block_size"]
    cons = EssentialConsensus(cap, n_hyp + kw.get("estimations_per_block", 0) * blocks_max)
    cons.reserve(S)
    thr = 1e-6
    prm = cons.make_params(thr, n_hypotheses=n_hyp, seed=41, **kw)
    for rep in range(2):
        cons.p3p_model_inliers_batch_device(d_k.data_ptr(), cap, ik, d_pairs.data_ptr(), d_np.data_ptr(), d_world.data_ptr(),
                                            d_world.shape[0], cons.camera(cam), prm, d_pose.data_ptr(), d_best.data_ptr(), d_inl.data_ptr(),
                                            d_ninl.data_ptr(), d_stats.data_ptr(), shuffle=shuffle)
    cons.sync()
    pose = d_pose.cpu().numpy(); best = d_best.cpu().numpy().view(np.uint32); inl = d_inl.cpu().numpy().view(np.uint32)
    ninl = d_ninl.cpu().numpy().view(np.uint32)
    st = d_stats.cpu().numpy().view(np.dtype([("poses", "<u4"), ("survivors", "<u4"), ("blocks", "<u4"), ("reserved", "<u4"),
                                               ("evaluated", "<u8"), ("exhaustive", "<u8")])).reshape(S)
    some_model = 0
    for s, (kps, pr, R, t) in enumerate(scenes):
        want = oracle.p3p_arrsac_pairs(kps, pr, world_all, cam, thr, n_hyp, scene=s, shuffle=shuffle, seed=41, **kw)
        ga, gw, go = cons.scene_world(s, cap)
        _eq(ga, want["bearings"], f"scene {s} bearings")
        _eq(gw, want["world"], f"scene {s} world points")
        if shuffle:
            _eq(go, want["order"], f"scene {s} scoring order")
        assert best[s] == want["best_id"], (name, s, len(pr), best[s], want["best_id"])
        assert ninl[s] == len(want["inliers"]), (name, s, ninl[s], len(want["inliers"]))
        if want["best_id"] == 0xFFFFFFFF:
            assert len(pr) < 3 or st["survivors"][s] == 0
            continue
        some_model += 1
        _eq(pose[s].reshape(3, 4), want["pose"], f"scene {s} pose ({name})")
        _eq(inl[s, :ninl[s]], want["inliers"], f"scene {s} inliers ({name})")
        for key, wkey in (("survivors", "survivors"), ("blocks", "blocks"), ("poses", "poses"), ("evaluated", "residuals_evaluated")):
            assert int(st[key][s]) == want["stats"][wkey], (name, s, key, st[s], want["stats"])
        if len(pr) >= 64:
            assert np.abs(pose[s].reshape(3, 4)[:, :3] - R).max() < 1e-2, (s, len(pr))
    assert some_model >= 8


def test_arrsac_refusals(gpu):
    """Re-sampling needs room for its hypotheses; reserved fields and unknown flags are refused."""
    from cv_amd import _lib
    from cv_amd.ransac import EssentialConsensus
    rng = np.random.default_rng(5)
    a, b = _two_view_scene(rng, 256, 0.2)
    cons = EssentialConsensus(256, 128)
    with pytest.raises(_lib.AkzError) as e:
        cons.arrsac_model_inliers(a, b, 1e-6, n_hypotheses=120, block_size=64, estimations_per_block=4)   # 120 + 4 x 4 > 128
    assert e.value.status == -6                                # AKZ_E_TOO_LARGE
    assert cons.arrsac_model_inliers(a, b, 1e-6, n_hypotheses=112, block_size=64, estimations_per_block=4) is not None
    # reserved field and unknown flag bits through the raw ABI
    import ctypes as C
    prm = _lib.ArrsacParams()
    prm.struct_size = C.sizeof(_lib.ArrsacParams)
    prm.n_hypotheses, prm.block_size, prm.init_blocks, prm.max_candidates = 64, 64, 1, 16
    prm.flags, prm.threshold, prm.sprt_delta, prm.sprt_ratio, prm.seed = 1, 1e-6, 0.05, 1e3, 0
    pose = np.empty(12); best = C.c_uint32(); ninl = C.c_uint32(); inl = np.empty(256, np.uint32)

    def call():
        return _lib.lib().rs_essential_arrsac(cons._h, a.ctypes.data, b.ctypes.data, 256, None, C.byref(prm), pose.ctypes.data,
                                              C.byref(best), inl.ctypes.data, 256, C.byref(ninl), None)
    assert call() == 0
    prm.reserved = 1
    assert call() == -1
    prm.reserved, prm.flags = 0, 1 << 7
    assert call() == -1
    prm.flags, prm.struct_size = 1, C.sizeof(_lib.ArrsacParams) - 8          # a caller built against the older struct
    assert call() == -1


OPTION_SETS = [
    ("tile determinant kernels", dict(stream_kernels=False)),
    ("streaming kernels everywhere", dict(stream_min_waves=1, stream_waves=64)),
    ("long streaming segments", dict(stream_min_waves=1, stream_waves=1)),
    ("one-frame kernels", dict(frame_pairs=False)),
    ("one FED step per launch", dict(fed_block=1)),
    ("three FED steps per launch", dict(fed_block=3)),
    ("five FED steps per launch (two-patch halo, fused first launch writes Lflow)", dict(fed_block=5)),
    ("seven FED steps per launch", dict(fed_block=7)),
    ("serial suppression, no pipeline", dict(parallel_suppression=False, pipeline=False)),
    ("exact contrast, equal stream priorities", dict(contrast="exact", stream_priority=False)),
    ("small candidate lists", dict(max_candidates=4096, desc_tile_shift=3)),
    ("determinant kernels on the scale-space stream", dict(det_side_stream=False)),
    ("front end and FED as two kernels", dict(fuse_front_fed=False)),
    ("tile kernels for the levels that fit one compute unit", dict(resident_levels=False)),
    ("one workgroup per frame for the levels that fit one compute unit, whatever the call size", dict(resident_min_frames=1)),
]


@pytest.mark.parametrize("name,kw", OPTION_SETS, ids=[n for n, _ in OPTION_SETS])
def test_every_option_gives_the_same_bits(gpu, oracle, name, kw):
    """akz_options select kernels, never results: a batch of five frames (odd: a half-empty frame pair) at 640x400
    and a ragged 333x251 one under each option set equal the oracle byte for byte."""
    akaze, _ = gpu
    for (w, h, nfr) in ((640, 400, 5), (333, 251, 2)):
        frames = [synth_frame(w, h, seed=3000 + i, n_rect=40, n_disc=40) for i in range(nfr)]
        ctx = akaze.Context(akaze.Akaze.default(), w, h, nfr, _opts(**kw))
        got = ctx.extract_batch(frames)
        orc = oracle.Akaze(w, h, oracle.default_config())
        for i, f in enumerate(frames):
            okp, od = orc.extract(f)
            _kp_eq(got[i][0], okp, f"{name} {w}x{h} frame {i}")
            _eq(got[i][1], od, f"{name} {w}x{h} frame {i} desc")
        ctx.close()


@pytest.mark.parametrize("w,h,nfr", [(960, 540, 1), (960, 540, 5), (1024, 512, 2), (1024, 576, 1), (640, 360, 3), (320, 180, 1),
                                     (512, 96, 2), (168, 1100, 1)])
def test_levels_resident_on_one_compute_unit(gpu, oracle, w, h, nfr):
    """k_level_resident (a whole level — front end and every FED step — in one launch, one workgroup per frame, the frame cut
    into an upper and a lower half that share every packed instruction): Lt, Lx, Ly of EVERY level, keypoints and descriptor
    bytes equal the oracle's.  960x540: octave 2 is the 240 x 135 plane of a 1080p pyramid's octave 3 (odd height: the lower
    half ends one row early); 1024x512 / 1024x576: a 256-column octave uses all 64 lanes of a wave (576: its plane no longer
    fits and the tile kernels take that octave); 640x360 / 320x180: 160x90, 80x45, 40x22 levels (few lanes, few waves);
    512x96: an octave of 24 rows, two waves; 168x1100: a tall narrow frame — 42 x 275 is too tall, 21-column levels are not
    divisible by 4 (tile kernels / one-frame kernels).  The same frames with AKZ_OPT_NO_RESIDENT_LEVELS give the same bytes."""
    akaze, _ = gpu
    frames = [synth_frame(w, h, seed=7700 + 13 * i + w, n_rect=50, n_disc=50) for i in range(nfr)]
    orc = oracle.Akaze(w, h, oracle.default_config())
    for kw in (dict(resident_min_frames=1), dict(resident_levels=False)):     # (by default only calls of >= 96 frames take it)
        ctx = akaze.Context(akaze.Akaze.default(), w, h, nfr, _opts(**kw))
        got = ctx.extract_batch(frames)
        for i in reversed(range(nfr)):        # (the oracle's buffers after the loop are frame 0's)
            okp, od = orc.extract(frames[i])
            _kp_eq(got[i][0], okp, f"{w}x{h} frame {i} {kw}")
            _eq(got[i][1], od, f"{w}x{h} frame {i} desc {kw}")
            for lvl in range(orc.num_levels):
                for name in ("Lt", "Lx", "Ly"):
                    _eq(ctx.level_buffer(i, lvl, name, w, h), orc.buffer(lvl, name), f"{w}x{h} frame {i} {name}[{lvl}] {kw}")
        ctx.close()


@pytest.mark.parametrize("nfr", [1, 3, 4, 5, 16, 17])
def test_call_size_selects_kernels_not_results(gpu, oracle, nfr):
    """The library picks kernels by the number of frames of a call — up to 4: determinant kernels on the side stream and
    up to 16 FED steps per launch; up to 8: rank sorts; up to 16: suppression chunks chained across workgroups; above:
    one workgroup per frame, bitonic sorts.  Every frame of every call size equals the oracle byte for byte (832x480:
    three octaves, more than one suppression chunk per frame)."""
    akaze, _ = gpu
    w, h = 832, 480
    frames = [synth_frame(w, h, seed=5100 + (i % 3), n_rect=120, n_disc=120) for i in range(nfr)]
    ctx = akaze.Context(akaze.Akaze.default(), w, h, nfr)
    got = ctx.extract_batch(frames)
    orc = oracle.Akaze(w, h, oracle.default_config())
    want = [orc.extract(frames[i]) for i in range(min(nfr, 3))]
    assert len(want[0][0]) > 1024                     # more than one suppression chunk of candidates
    for i in range(nfr):
        okp, od = want[i % 3]
        _kp_eq(got[i][0], okp, f"call of {nfr} frames, frame {i}")
        _eq(got[i][1], od, f"call of {nfr} frames, frame {i} desc")
    ctx.close()


def test_host_calls_with_and_without_pinned_staging(gpu, oracle):
    """akz_extract_batch stages inputs and outputs through pinned blocks while they stay below 96 MB and falls back to
    plain copies above (72 frames x 16 384 keypoint slots x 92 B = 108 MB): both give the oracle's bytes, and a second
    call on the same context (staging blocks reused) too."""
    akaze, _ = gpu
    w, h = 160, 120
    frames = [synth_frame(w, h, seed=6100 + (i % 4), n_rect=20, n_disc=20) for i in range(72)]
    orc = oracle.Akaze(w, h, oracle.default_config())
    want = [orc.extract(frames[i]) for i in range(4)]
    for nfr in (3, 72):
        ctx = akaze.Context(akaze.Akaze.default(), w, h, nfr)
        for rep in range(2):
            got = ctx.extract_batch(frames[:nfr])
            for i in range(nfr):
                okp, od = want[i % 4]
                _kp_eq(got[i][0], okp, f"host call of {nfr} frames (call {rep}), frame {i}")
                _eq(got[i][1], od, f"host call of {nfr} frames (call {rep}), frame {i} desc")
        ctx.close()


@pytest.mark.parametrize("W,H,h,levels", [(4096, 65528, 2048, (3, 7, 11, 15)), (65528, 4096, 640, (3, 7, 11))], ids=["tall", "wide"])
def test_frame_at_the_pixel_cap_is_addressed_correctly(gpu, oracle, W, H, h, levels):
    """Maximum size.  akz_create accepts frames of up to 2^28 pixels because the diffusion / determinant kernels address a frame's
    planes with 32-bit byte offsets (8 B per pixel for {Lx, Ly}: 2^31 at the cap).  A 4096 x 65528 frame (268.4 M pixels) carries
    its content in the LAST 2048 rows — the highest addresses of every plane — under a constant field, the first rows of the
    content being the same constant, so that clamping (the cropped image the oracle gets) and the constant above (the big
    frame) are the same neighbourhood.  Pixel VALUES do not depend on where a pixel lies: Lt, Lx and Ly of the last level of
    every octave must equal the oracle's planes of the cropped image bit for bit below the rows the different upper boundary
    can have reached (FED steps and filter supports, octave by octave), and the keypoints of the first octaves found there must
    be the oracle's (level, x, response, size; y up to the f32 spacing at 65 000).  "wide" is the same with 65 528-pixel rows
    (512 determinant bands per row, 1 170 diffusion windows per row) and a lower content, checked up to octave 2."""
    akaze, _ = gpu
    y0 = H - h
    assert y0 % 8 == 0 and W * H <= 1 << 28 and W * H > (1 << 28) - (1 << 20)
    small = synth_frame(W, h, seed=0xB16, n_rect=700, n_disc=700)
    small[:16] = 96
    big = np.full((H, W), 96, np.uint8)
    big[y0:] = small
    ak = akaze.Akaze.default()
    ak.max_keypoints = 65536
    ctx = akaze.Context(ak, W, H, 1)
    (kp, _desc), = ctx.extract_batch([big])
    orc = oracle.Akaze(W, h, oracle.default_config())
    okp, _od = orc.extract(small)
    assert ctx.num_levels(W, H) == orc.num_levels == 16
    # rows (in the octave's own pixels) the upper boundary can have influenced: the previous octave's, halved, + this octave's FED
    # steps + a generous 40 for the blur, Scharr and multiscale-derivative supports of its four levels
    reach, steps = [], [10, 22, 44, 90]
    for o in range(4):
        reach.append((reach[-1] + 1) // 2 + steps[o] + 40 if reach else steps[0] + 40)
    checked = 0
    for lvl in levels:
        o = lvl // 4
        m, yl = reach[o] + 8, y0 >> o
        for name in ("Lt", "Lx", "Ly"):
            g = ctx.level_buffer(0, lvl, name, W, H)
            want = orc.buffer(lvl, name)
            assert g.shape == (H >> o, W >> o) and want.shape == (h >> o, W >> o)
            _eq(g[yl + m:], want[m:], f"{name}[{lvl}] rows below the boundary's reach")
            assert np.all(g[:yl - m].view(np.uint32) == g[0, 0].view(np.uint32)), f"{name}[{lvl}]: the constant field is not constant"
            checked += want[m:].size
    assert checked > 4_000_000
    # keypoints of levels 0..6 (they depend on levels <= 7 only: a candidate is matched against entries of its own and the
    # previous level, and the second pass looks one level up) well below the boundary
    sel_g = kp[(kp["class_id"] <= 6) & (kp["y"] > y0 + 400)]
    sel_o = okp[(okp["class_id"] <= 6) & (okp["y"] > 400)]
    assert len(sel_o) > 2000 and len(sel_g) == len(sel_o), (len(sel_g), len(sel_o))
    for f in ("class_id", "octave"):
        assert np.array_equal(sel_g[f], sel_o[f]), f
    for f in ("x", "response", "size"):
        _eq(sel_g[f], sel_o[f], f"large frame keypoint {f}")
    assert np.abs((sel_g["y"].astype(np.float64) - y0) - sel_o["y"]).max() < 0.05
    ctx.close()
    # a little wider is over the cap: refused (AKZ_E_TOO_LARGE), not wrapped
    with pytest.raises(_lib_error()) as e:
        akaze.Context(ak, *((W + 8, H) if W < H else (W, H + 8)), 1)
    assert e.value.status == -6


def _lib_error():
    from cv_amd import _lib
    return _lib.AkzError


def test_unknown_option_bits_are_refused(gpu):
    import ctypes as C
    from cv_amd import _lib
    cfg = _lib.Config()
    _lib.lib().akz_config_default(C.byref(cfg))
    o = _lib.make_options()
    o.flags = 1 << 20
    h = C.c_void_p()
    assert _lib.lib().akz_create_ex(C.byref(cfg), 0, 64, 64, 1, 0, C.byref(o), C.byref(h)) == -1
    assert _lib.lib().hm_create_ex(0, 64, 64, 1 << 24, C.byref(h)) == -1
    assert _lib.lib().hm_create_ex(0, 64, 64, 33 << 16, C.byref(h)) == -1      # (bits 16..21: CUs per XCD for the stream, at most 32)


def test_mirrored_pose_pairs_share_one_eigen_decomposition(gpu, oracle):
    """possible_unscaled_poses returns (t,R1), (t,R2), (-t,R1), (-t,R2) (cv-pinhole/src/essential.rs:217-231); the
    scoring kernels take the residuals of [R | t] and [R | -t] from ONE eigen-decomposition (rs_residual_pair).  Both
    must carry the bits of the direct evaluation — on random scenes, on poses with exactly-zero structure, and on the
    case that path singles out: an eigenvector whose last component is exactly zero (a point at infinity; bearing b
    parallel to t with a = R^T b), which it evaluates directly."""
    from cv_amd.ransac import EssentialConsensus
    from test_oracle_ransac import _rot
    rng = np.random.default_rng(0x314)
    cons = EssentialConsensus(1024, 16)
    a, b = _two_view_scene(rng, 600, 0.3)
    poses = []
    for _ in range(12):
        R = _rot((rng.random(3) - 0.5) * 2.0); t = rng.standard_normal(3); t /= np.linalg.norm(t)
        poses.append(np.concatenate([R, t[:, None]], 1))
    # exact-zero structure: identity rotation, axis-aligned translations, a zero translation
    for t in ([0, 0, 1], [1, 0, 0], [0, -1, 0], [0, 0, 0]):
        poses.append(np.concatenate([np.eye(3), np.array(t, float)[:, None]], 1))
    poses = np.array(poses)
    # matches that hit the exactly-singular cases: b parallel to t, a = b (R = I); axis bearings
    ax = np.array([[0, 0, 1.0], [1.0, 0, 0], [0, -1.0, 0], [0, 1.0, 0], [0.6, 0, 0.8]])
    a = np.concatenate([a, ax, ax]); b = np.concatenate([b, ax, ax[::-1]])
    neg = poses.copy(); neg[:, :, 3] = -neg[:, :, 3]
    direct = cons.residuals(np.concatenate([poses, neg]), a, b)
    paired = cons.residuals(poses, a, b, paired=True)
    _eq(paired[:, 0], direct[:len(poses)], "pair path, pose [R | t]")
    _eq(paired[:, 1], direct[len(poses):], "pair path, mirrored pose [R | -t]")
    want = np.array([[oracle.pose_residual(P, a[m], b[m]) for m in range(len(a))] for P in np.concatenate([poses, neg])])
    _eq(direct, want, "device residual vs oracle")
    # the singled-out case really occurs in this set: identity rotation, t = b = a = e_z gives design = diag(2,2,0,0)
    assert direct[12, 600] == want[12, 600]


def _np_residual(pose, a, b):
    """CameraToCamera::residual restated independently of include/akz_ransac_math.h: numpy, LAPACK's eigh."""
    P0 = np.concatenate([np.eye(3), np.zeros((3, 1))], 1)
    D = np.zeros((4, 4))
    for P, br in ((P0, a), (pose, b)):
        T = P - np.outer(br, br) @ P
        D += T.T @ T
    w, V = np.linalg.eigh(D)
    p = V[:, np.argmin(np.abs(w))]
    if p[3] < 0 or (p[3] == 0 and np.signbit(p[3])):
        p = -p
    p = p / np.linalg.norm(p[:3])
    q = np.concatenate([pose[:, :3] @ p[:3] + pose[:, 3] * p[3], [p[3]]])
    q = q / np.linalg.norm(q[:3])
    return 0.5 * (1.0 - a @ p[:3] + 1.0 - b @ q[:3])


def test_device_geometry_against_independent_f64_statements(gpu):
    """Round-2 verdict, weak spot 1b: for R1-R3 and R5 the oracle and the kernels share their numerical core
    (include/akz_ransac_math.h, akz_p3p_math.h), so HIP == oracle proves gcc == hipcc on one source.  Here DEVICE outputs
    are held to statements that never saw those headers — numpy + LAPACK:
      * eight-point E (recovered from the device's poses) spans the null space numpy.linalg.svd finds for the 8 x 9
        epipolar system (|<E, E_ref>| = 1 to 1e-9) and annihilates the eight matches (eight-point/src/lib.rs:11-58);
      * the four poses are the SVD decomposition of E: R in SO(3), t = +-u3, [t]x R ~ +-E (cv-pinhole/src/essential.rs:114-231);
      * CameraToCamera::residual equals the numpy eigh restatement to 1e-12 (cv-core/src/pose.rs:249-295);
      * Lambda Twist from rs_p3p_batch recovers the literal pose of lambda-twist/tests/consensus.rs:20-57 to 1e-6."""
    from cv_amd.ransac import EssentialConsensus
    from test_oracle_ransac import arrsac_manual_scene, _projective
    rng = np.random.default_rng(0x1B)
    n, n_hyp = 400, 96
    a, b = _two_view_scene(rng, n, 0.0)
    samples = np.stack([rng.choice(n, 8, replace=False) for _ in range(n_hyp)]).astype(np.uint32)
    cons = EssentialConsensus(1024, 256)
    assert cons.model_inliers(a, b, samples, 1e-7) is not None
    P, ok = cons.poses(n_hyp)
    assert ok.all()
    for h in range(n_hyp):
        sa, sb = a[samples[h]], b[samples[h]]
        A = np.stack([np.kron(x / x[2], y / x[2]) for x, y in zip(sa, sb)])      # eight-point/src/lib.rs:11-24 (b over a.z: sic)
        e_ref = np.linalg.svd(A)[2][-1]                                          # null vector, unit norm
        E_ref = e_ref.reshape(3, 3).T                                            # Matrix3::from_iterator: column-major
        for p in range(4):
            R, t = P[h, p, :, :3], P[h, p, :, 3]
            assert abs(np.linalg.det(R) - 1.0) < 1e-9 and np.abs(R @ R.T - np.eye(3)).max() < 1e-9, (h, p)
            assert abs(np.linalg.norm(t) - 1.0) < 1e-9
            tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
            E = tx @ R
            E = E / np.linalg.norm(E)
            assert abs(abs((E * E_ref).sum()) - 1.0) < 1e-9, (h, p, (E * E_ref).sum())
            # b^T E a = 0 on the sample: to 1e-6 — the eigen-solver stops at a relative off-diagonal norm of 1e-12
            # (EightPoint::default, eight-point/src/lib.rs:60-67), which an ill-conditioned sample (second-smallest
            # eigenvalue 1e-5 and below) amplifies into the null vector
            assert np.abs(np.einsum("ni,ij,nj->n", sb, E, sa)).max() < 1e-6
        assert np.allclose(P[h, 0, :, 3], -P[h, 2, :, 3]) and np.array_equal(P[h, 0, :, :3], P[h, 2, :, :3])
        assert np.array_equal(P[h, 1, :, :3], P[h, 3, :, :3]) and not np.allclose(P[h, 0, :, :3], P[h, 1, :, :3])
    # residuals: device vs numpy eigh, on inlier and on random (pose, match) pairs
    poses = P[:6].reshape(-1, 3, 4)
    rr = cons.residuals(poses, a[:64], b[:64])
    for i, pose in enumerate(poses):
        for m in range(64):
            want = _np_residual(pose, a[m], b[m])
            assert abs(rr[i, m] - want) < 1e-12 * max(1.0, abs(want)) + 1e-12, (i, m, rr[i, m], want)
    # Lambda Twist on the device: the reference's own pose-recovery scene (lambda-twist/tests/consensus.rs:20-57)
    R, t, bb, w = arrsac_manual_scene()
    tri = np.array([[0, 1, 2], [1, 2, 3], [0, 2, 4], [2, 3, 4]], np.uint32)
    got = cons.p3p_model_inliers(bb, w, tri, 0.01)         # the reference's threshold (consensus.rs:59)
    assert got is not None
    pose, inl, _ = got
    assert np.abs(pose[:, :3] - R).max() < 1e-6 and np.abs(pose[:, 3] - t).max() < 1e-6
    assert len(inl) == len(bb)


def test_device_transcendentals_against_the_host_libm(gpu, oracle):
    """Round-2 verdict, weak spot 1b (A14): the angle bits equal the oracle's because both sides evaluate
    include/akz_portable_math.h.  Independent evidence on the DEVICE side: atan2f / sinf / cosf as the gfx950 kernels
    compute them, on 10^5 values each, against glibc (what Rust's f32::atan2 / sin / cos call on Linux) within 1 ulp
    and against the correctly rounded f64 value within 1 ulp; and bit for bit against the oracle's build of the header."""
    import ctypes
    from test_oracle_math import _ulp_diff
    akaze, _ = gpu
    from cv_amd import _lib
    ctx = akaze.Akaze.default().context(64, 64, 1)
    rng = np.random.default_rng(0x7A)
    n = 100000
    y = (rng.standard_normal(n) * 10.0 ** rng.uniform(-6, 2, n)).astype(np.float32)
    x = (rng.standard_normal(n) * 10.0 ** rng.uniform(-6, 2, n)).astype(np.float32)
    ang = rng.uniform(0, 2 * np.pi, n).astype(np.float32)
    ang[:8] = np.array([0, np.pi, np.pi / 2, 3 * np.pi / 2, 2 * np.pi, 1e-8, 6.2831855, 3.1415927], np.float32)

    def dev(which, xs, ys=None):
        out = np.empty(len(xs), np.float32)
        _lib.check(_lib.lib().akz_debug_portable_math(ctx.handle, which, xs.ctypes.data, ys.ctypes.data if ys is not None else None,
                                                      len(xs), out.ctypes.data), "akz_debug_portable_math")
        return out
    at, si, co = dev(0, x, y), dev(1, ang), dev(2, ang)
    libm = ctypes.CDLL("libm.so.6")
    libm.atan2f.restype = ctypes.c_float; libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]
    for fn in (libm.sinf, libm.cosf):
        fn.restype = ctypes.c_float; fn.argtypes = [ctypes.c_float]
    g_at = np.array([libm.atan2f(float(a), float(b)) for a, b in zip(y, x)], np.float32)
    g_si = np.array([libm.sinf(float(v)) for v in ang], np.float32)
    g_co = np.array([libm.cosf(float(v)) for v in ang], np.float32)
    assert _ulp_diff(at, g_at).max() <= 1 and _ulp_diff(si, g_si).max() <= 1 and _ulp_diff(co, g_co).max() <= 1
    # (glibc 2.35's atan2f / sinf / cosf are 0.5x-ulp functions, not correctly rounded: ~13 % / ~1 % of inputs are 1 ulp off
    # the correctly rounded value, which is what the device returns in all but a handful of cases)
    assert (si != g_si).mean() < 0.03 and (co != g_co).mean() < 0.03
    cr_at = np.arctan2(y.astype(np.float64), x.astype(np.float64)).astype(np.float32)
    assert _ulp_diff(at, cr_at).max() <= 1 and (at != cr_at).mean() < 1e-4
    assert _ulp_diff(si, np.sin(ang.astype(np.float64)).astype(np.float32)).max() <= 1
    assert _ulp_diff(co, np.cos(ang.astype(np.float64)).astype(np.float32)).max() <= 1
    _eq(at, oracle.pm_atan2f(y, x), "device atan2f vs the oracle's build of akz_portable_math.h")
    os_, oc_ = oracle.pm_sincosf(ang)
    _eq(si, os_, "device sinf"); _eq(co, oc_, "device cosf")


def test_v_rcp_f32_meets_its_specification_on_this_device(gpu):
    """The error bound of the orientation kernel's angle estimate (tests/test_oracle_math.py, tools/ubench/atan_bound.c) takes
    v_rcp_f32's accuracy — 1 ulp — from the instruction set's specification.  Checked here on the device itself, against 1 / x
    in f64, for EVERY f32 of [1e-30, 1e30] (the operand range ori_sample_entry admits to the estimate path): 1.67 G values."""
    import ctypes as C
    akaze, _ = gpu
    from cv_amd import _lib
    ctx = akaze.Akaze.default().context(64, 64, 1)
    lo = int(np.float32(1e-30).view(np.uint32)) - 1
    hi = int(np.float32(1e30).view(np.uint32)) + 1
    worst = C.c_double(-1.0)
    _lib.check(_lib.lib().akz_debug_rcp_error(ctx.handle, lo, hi, C.byref(worst)), "akz_debug_rcp_error")
    assert 0.0 < worst.value <= 1.0, worst.value
    assert _lib.lib().akz_debug_rcp_error(ctx.handle, 0, hi, C.byref(worst)) == -1          # denormals are not part of the claim


def test_orientation_window_membership_estimate_equals_the_exact_expression(gpu, oracle):
    """k_orient_describe places every orientation sample among the windows' end points from an f32 ESTIMATE of the angle and
    evaluates the exact expression (akz_portable_math.h, f64) only when the estimate lies within 8e-6 of an end point
    (ori_sample_masks).  akz_debug_orientation_masks returns both decisions.  They must agree on: 800 000 random gradients;
    gradients whose angle is swept in 1e-7 steps through +-2e-5 around EVERY end point (42 window starts, their ends, 0 and
    2 pi) at several magnitudes; the axes, the diagonals, signed zeros, denormal, huge, infinite and NaN operands.  The
    exact decision itself is re-derived here from the oracle's atan2 and the reference's window predicate
    (scale_space_extrema.rs:242-287), and the band must be rare on random input (it is what the kernel saves)."""
    akaze, _ = gpu
    from cv_amd import _lib
    f32 = np.float32
    ctx = akaze.Akaze.default().context(64, 64, 1)
    rng = np.random.default_rng(0x0A1)
    n_rand = 400000
    mag = 10.0 ** rng.uniform(-8, 3, n_rand)                 # directions uniform on the circle, any magnitude ...
    y = (rng.standard_normal(n_rand) * mag).astype(f32)
    x = (rng.standard_normal(n_rand) * mag).astype(f32)
    wy = (rng.standard_normal(n_rand) * 10.0 ** rng.uniform(-8, 3, n_rand)).astype(f32)   # ... and components decades apart
    wx = (rng.standard_normal(n_rand) * 10.0 ** rng.uniform(-8, 3, n_rand)).astype(f32)   # (angles hugging the axes)
    # the windows of the reference: starts by f32 accumulation of 0.15 below 2 pi, width pi / 3, wrapping
    PI = f32(3.14159274101257324219)
    starts = []
    a1 = f32(0.0)
    while a1 < f32(2.0) * PI:
        starts.append(a1)
        a1 = f32(a1 + f32(0.15))
    assert len(starts) == 42
    ends = [f32(s - f32(f32(5.0) * PI) / f32(3.0)) if f32(s + PI / f32(3.0)) > f32(2.0) * PI else f32(s + PI / f32(3.0)) for s in starts]
    pts = np.array(starts + ends + [0.0, float(f32(2.0) * PI)], np.float64)
    sweep = (pts[:, None] + np.linspace(-2e-5, 2e-5, 401)[None, :]).reshape(-1)
    xs, ys = [x, wx], [y, wy]
    for mag in (1.0, 3.7e-4, 251.0):
        xs.append((np.cos(sweep) * mag).astype(f32)); ys.append((np.sin(sweep) * mag).astype(f32))
    special = np.array([0.0, -0.0, 1.0, -1.0, 1e-40, -1e-40, 1e-30, 3e38, -3e38, np.inf, -np.inf, np.nan, 1e-20, 2.5e-7], f32)
    gx, gy = np.meshgrid(special, special)
    xs.append(gx.reshape(-1)); ys.append(gy.reshape(-1))
    d = (rng.standard_normal(20000) * 10.0 ** rng.uniform(-6, 2, 20000)).astype(f32)
    for sx, sy in ((1, 1), (-1, 1), (1, -1), (-1, -1)):                       # the diagonals, and a hair off them
        xs.append(sx * d); ys.append(sy * d)
        xs.append(sx * d); ys.append((sy * d * f32(1.0000001)).astype(f32))
    x = np.ascontiguousarray(np.concatenate(xs), f32); y = np.ascontiguousarray(np.concatenate(ys), f32)
    n = len(x)
    fast = np.zeros(n, np.uint64); exact = np.zeros(n, np.uint64); fell = np.zeros(n, np.uint32)
    _lib.check(_lib.lib().akz_debug_orientation_masks(ctx.handle, x.ctypes.data, y.ctypes.data, n, fast.ctypes.data, exact.ctypes.data,
                                                      fell.ctypes.data), "akz_debug_orientation_masks")
    bad = np.nonzero(fast != exact)[0]
    assert len(bad) == 0, [(float(x[i]), float(y[i]), hex(int(fast[i])), hex(int(exact[i])), int(fell[i])) for i in bad[:5]]
    assert fell[:n_rand].mean() < 1e-3, fell[:n_rand].mean()                 # the band is rare on directions at random ...
    assert fell[2 * n_rand:2 * n_rand + 3 * len(sweep)].mean() > 0.3          # ... and is what the sweeps exercise
    # the exact decision, independently: (atan2 + 2 pi) mod 2 pi in f32, then the reference's predicate per window
    fin = np.isfinite(x) & np.isfinite(y)
    xa, ya = x[fin], y[fin]
    two_pi = f32(2.0) * PI
    a = (oracle.pm_atan2f(ya, xa) + two_pi).astype(f32)
    ang = np.where(a >= two_pi, (a - two_pi).astype(f32), a)
    want = np.zeros(len(xa), np.uint64)
    for wd, (s, e) in enumerate(zip(starts, ends)):
        inside = ((s < e) & (s < ang) & (ang < e)) | ((e < s) & (((ang > 0) & (ang < e)) | ((ang > s) & (ang < two_pi))))
        want |= inside.astype(np.uint64) << np.uint64(wd)
    assert (want == exact[fin]).all()
    ctx.close()


def test_comm_c_abi_on_one_rank(gpu):
    """akz_comm_* (include/akz.h: the exchange step through the library itself, RCCL loaded with dlopen): a communicator of
    ONE rank on the GPU — the ring shift is a send to oneself, the all-gather a copy — so the RCCL branch of the N > 1
    path has executed on hardware before a multi-GPU run needs it: received rows byte-equal to the sent ones, rows outside
    the transfer untouched, stream ordering through akz_comm_stream(), the timing bracket counts the bytes."""
    import ctypes as C
    import torch
    from cv_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    ident = (C.c_uint8 * 128)()
    _lib.check(L.akz_comm_unique_id(ident), "akz_comm_unique_id")
    h = C.c_void_p()
    _lib.check(L.akz_comm_create(ident, 0, 1, 0, C.byref(h)), "akz_comm_create")
    try:
        assert L.akz_comm_rank(h) == 0 and L.akz_comm_world(h) == 1
        n, cap = 5, 512
        g = torch.Generator(device=dev).manual_seed(3)
        descs = torch.randint(0, 256, (n, cap, 64), generator=g, device=dev, dtype=torch.uint8)
        counts = torch.arange(100, 100 + n, device=dev, dtype=torch.int32)
        rd = torch.full((n + 2, cap, 64), 7, dtype=torch.uint8, device=dev)
        rc = torch.full((n + 2,), -1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        _lib.check(L.akz_comm_timing(h, 1, None, None, None, 1), "timing")
        cur = torch.cuda.current_stream()
        _lib.check(L.akz_comm_shift_blocks(h, descs.data_ptr(), counts.data_ptr(), n, cap, rd[1:].data_ptr(), rc[1:].data_ptr(),
                                           _lib.wait_handle(cur)), "akz_comm_shift_blocks")
        ad = torch.zeros((1, n, cap, 64), dtype=torch.uint8, device=dev)
        ac = torch.zeros((1, n), dtype=torch.int32, device=dev)
        # (the zero fills run on torch's stream, the gather on the communicator's: it waits for them)
        _lib.check(L.akz_comm_allgather_blocks(h, descs.data_ptr(), counts.data_ptr(), n, cap, ad.data_ptr(), ac.data_ptr(),
                                               _lib.wait_handle(cur)), "akz_comm_allgather_blocks")
        _lib.check(L.akz_comm_sync(h), "akz_comm_sync")
        assert torch.equal(rd[1:n + 1], descs) and torch.equal(rc[1:n + 1], counts)
        assert bool((rd[0] == 7).all()) and bool((rd[n + 1] == 7).all()) and int(rc[0]) == -1 and int(rc[n + 1]) == -1
        assert torch.equal(ad[0], descs) and torch.equal(ac[0], counts)
        ms, calls, nbytes = C.c_double(), C.c_uint64(), C.c_uint64()
        _lib.check(L.akz_comm_timing(h, 0, C.byref(ms), C.byref(calls), C.byref(nbytes), 1), "timing")
        assert calls.value == 2 and nbytes.value == 2 * n * (cap * 64 + 4) and ms.value > 0.0
        # the caller's stream here is the LEGACY DEFAULT stream, whose handle is 0 = "nothing to wait for" in the ABI:
        # _lib.wait_handle() names it AKZ_STREAM_LEGACY instead.  (Passed as 0, the fills below raced the gather — one
        # run in five of this test failed on `ac`.)  A long fill directly before every call makes the order visible.
        assert _lib.wait_handle(cur) == _lib.STREAM_LEGACY or cur.cuda_stream != 0
        big = torch.empty((64 << 20,), dtype=torch.uint8, device=dev)
        for rep in range(24):
            big.fill_(rep)                                      # ~10 us of work ahead of the fills on the caller's stream
            ad.fill_(rep & 0xFF)
            ac.fill_(-rep - 1)
            _lib.check(L.akz_comm_allgather_blocks(h, descs.data_ptr(), counts.data_ptr(), n, cap, ad.data_ptr(), ac.data_ptr(),
                                                   _lib.wait_handle(cur)), "akz_comm_allgather_blocks")
            _lib.check(L.akz_comm_sync(h), "akz_comm_sync")
            assert torch.equal(ad[0], descs) and torch.equal(ac[0], counts), rep
        # refusals
        assert L.akz_comm_shift_blocks(h, None, counts.data_ptr(), n, cap, rd.data_ptr(), rc.data_ptr(), None) == -1
        assert L.akz_comm_create(ident, 1, 1, 0, C.byref(C.c_void_p())) == -1
    finally:
        _lib.check(L.akz_comm_destroy(h), "akz_comm_destroy")


@pytest.mark.parametrize("comm,exchange", [("akz", "shift"), ("akz", "allgather"), ("torch", "shift"), ("torch", "allgather")])
def test_exchange_code_paths_on_one_rank(gpu, tmp_path, comm, exchange):
    """bench.py's N > 1 code path with ONE rank on the real backends (--force-exchange: the rank sends to itself): the
    library's own RCCL exchange (akz_comm_*) and torch.distributed's NCCL, ring shift and all-gather — the pair lists of
    every frame must equal the plain single-rank run's.  (Two NCCL ranks cannot share a GPU, so this is as far as the
    RCCL branches can be executed on a one-GPU box.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    m1, m2 = tmp_path / "m1.npy", tmp_path / "m2.npy"
    common = ["--frames", "16", "--micro-batch", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--no-isolated"]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dump-matches", str(m1)] + common,
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    env = dict(os.environ, MASTER_PORT="29541")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dump-matches", str(m2), "--force-exchange", "--comm", comm,
                        "--exchange", exchange] + common, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert "multi_gpu" in line and min(line["multi_gpu"]["per_rank_frames_per_s"]) > 0 and line["multi_gpu"]["ranks"] == 1
    assert np.array_equal(np.load(m1), np.load(m2))
    a, b = np.load(str(m1) + ".r0.npz"), np.load(str(m2) + ".r0.npz")
    for g in range(16):
        assert np.array_equal(a[f"g{g}"], b[f"g{g}"]), f"pairs of frame {g} differ ({comm}, {exchange})"
        assert len(a[f"g{g}"]) > 500


def test_window_of_recent_views_two_ranks_equals_single_rank(gpu, tmp_path):
    """Windowed matching across ranks (cv-sfm/src/lib.rs:1462-1486, tracking_recent_frames): every frame's features against
    each of its K = 4 predecessors.  Two ranks (gloo, both on cuda:0) shard 24 global frames, all-gather their descriptor
    blocks per micro-batch and search the gathered views; the neighbour lists of every global frame — [view][query][2]
    {index, distance} — must equal, byte for byte, the single-rank run over the same frames."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    m1, m2 = tmp_path / "w1.npy", tmp_path / "w2.npy"
    common = ["--micro-batch", "4", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras", "--no-isolated", "--recent", "4"]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--frames", "24", "--dump-matches", str(m1)] + common,
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29537", os.path.join(root, "bench.py"),
                        "--gpus", "2", "--frames", "12", "--backend", "gloo", "--share-device",
                        "--dump-matches", str(m2)] + common,
                       capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    single = np.load(str(m1) + ".r0.npz")
    ranks = [np.load(str(m2) + f".r{r}.npz") for r in range(2)]
    for g in range(24):
        want, got = single[f"g{g}"], ranks[g % 2][f"g{g}"]
        assert want.shape == got.shape and want.shape[0] == 4 and want.shape[1] > 1000, (g, want.shape, got.shape)
        assert np.array_equal(want, got), f"neighbour lists of global frame {g} differ"
    # the window really reaches back: view k of frame g is frame g - k (its first neighbour distances differ per view)
    assert not np.array_equal(single["g10"][0], single["g10"][3])


def test_bench_gpus_2_launches_its_own_ranks(gpu):
    """The driver's command shape with another N — `python3 bench.py --gpus 2 --steps K --warmup W`, no launcher around it —
    must start its own two ranks (here both on cuda:0) and rank 0 must print the one parseable headline as the LAST line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-device", "--steps", "1", "--warmup", "1",
                        "--frames", "16", "--micro-batch", "8", "--no-extras", "--no-cpu-baseline", "--no-isolated"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = r.stdout.strip().splitlines()[-1]
    assert len(line) < 4096
    h = json.loads(line)
    assert h["n_gpus"] == 2 and h["value"] > 0 and h["steps"] == 1 and h["config"]["frames_per_gpu_per_step"] == 16
    assert h["multi_gpu"]["exchange"] == "ring shift" and "roofline" in h


def test_bench_gpus_8_on_one_device_equals_single_rank(gpu, tmp_path):
    """BASELINE configs[4] rehearsed without eight GPUs: the driver's command shape at N = 8 — `python3 bench.py --gpus 8 ...`,
    no launcher around it — starts its own eight ranks (all on cuda:0, gloo), 128 global frames g -> rank g % 8, every rank
    passes its blocks one rank up the ring; the last line parses, its `multi_gpu` object names eight ranks, and the match
    pair lists of all 128 global frames equal the single-rank run's byte for byte.  The day an 8-GPU node runs this the only
    untested thing is the wire."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    m1, m8 = tmp_path / "m1.npy", tmp_path / "m8.npy"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    common = ["--micro-batch", "8", "--steps", "1", "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--no-isolated"]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--share-device", "--frames", "16",
                        "--dump-matches", str(m8)] + common, capture_output=True, text=True, timeout=1800, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = r.stdout.strip().splitlines()[-1]
    assert len(line) < 4096
    h = json.loads(line)
    assert h["n_gpus"] == 8 and h["value"] > 0 and h["config"]["frames_per_gpu_per_step"] == 16
    mg = h["multi_gpu"]
    assert mg["ranks"] == 8 and mg["exchange"] == "ring shift" and len(mg["per_rank_frames_per_s"]) == 8
    assert all(v > 0 for v in mg["per_rank_frames_per_s"])
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--frames", "128", "--dump-matches", str(m1)] + common,
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    a, b = np.load(m1), np.load(m8)
    assert a.shape == b.shape == (128, 2) and np.array_equal(a, b), (a.T, b.T)
    single = np.load(str(m1) + ".r0.npz")
    ranks = [np.load(str(m8) + f".r{q}.npz") for q in range(8)]
    for g in range(128):
        want, got = single[f"g{g}"], ranks[g % 8][f"g{g}"]
        assert want.shape == got.shape and np.array_equal(want, got), f"pairs of global frame {g} differ"
        assert len(want) == a[g, 1] and len(want) > 500


@pytest.mark.parametrize("dtype,channels", [(np.uint8, 3), (np.uint8, 4), (np.uint16, 3), (np.float32, 3), (np.float32, 4)])
def test_colour_input_arms(gpu, oracle, dtype, channels):
    """The colour arms of GrayFloatImage::from_dynamic (akaze/src/image.rs:45-46, 87-106) on the C ABI (akz_extract_color):
    DynamicImage::grayscale() on the device, then the gray path — keypoints and descriptors equal the oracle's on the
    oracle's luma plane (oracle/color_oracle.c: orc_luma), and the luma plane itself is what the numpy restatement of the
    `image` crate's formula gives."""
    akaze, _ = gpu
    rng = np.random.default_rng(0xC0 + channels)
    h, w = 200, 336
    base = synth_frame(w, h, 91).astype(np.float64)
    planes = [np.clip(base * f + rng.uniform(-12, 12, (h, w)), 0, 255) for f in (1.0, 0.8, 1.15)]
    if channels == 4:
        planes.append(rng.uniform(0, 255, (h, w)))
    img = np.stack(planes, 2)
    if dtype == np.uint8:
        img = img.astype(np.uint8)
    elif dtype == np.uint16:
        img = (img * 257.0).astype(np.uint16)
    else:
        img = (img / 255.0).astype(np.float32)
    luma = oracle.luma(img)
    if dtype == np.float32:
        want_l = ((2126.0 * img[..., 0].astype(np.float64) + 7152.0 * img[..., 1].astype(np.float64))
                  + 722.0 * img[..., 2].astype(np.float64)) / 10000.0
        assert np.array_equal(luma, want_l.astype(np.float32))
    else:
        want_l = (2126 * img[..., 0].astype(np.uint64) + 7152 * img[..., 1].astype(np.uint64) + 722 * img[..., 2].astype(np.uint64)) // 10000
        assert np.array_equal(luma, want_l.astype(dtype))
    ak = akaze.Akaze.default()
    kp, d = ak.extract_arrays(img)
    okp, od = oracle.Akaze(w, h, oracle.default_config()).extract(luma)
    assert len(kp) == len(okp) > 50
    _eq(kp.view(np.uint8), okp.view(np.uint8), f"colour arm keypoints ({dtype.__name__} x{channels})")
    _eq(d, od, "colour arm descriptors")
    # the same luma plane through the gray entry gives the same result
    kp2, d2 = ak.extract_arrays(luma)
    assert kp2.tobytes() == kp.tobytes() and np.array_equal(d2, d)


def test_capacity_grows_instead_of_failing(gpu, oracle):
    """The reference's keypoint Vecs are unbounded (maximum_features = usize::MAX, akaze/src/lib.rs:172).  The library's
    lists have a capacity fixed at context creation; the host mirror repeats a call that overflowed it with twice the
    capacity (AKZ_E_INTERNAL + akz_last_overflow -> a larger context), so a caller sees the reference's result."""
    akaze, _ = gpu
    img = synth_frame(640, 360, 17)
    want_kp, want_d = oracle.Akaze(640, 360, oracle.default_config()).extract(img)
    assert len(want_kp) > 500
    ak = akaze.Akaze.default()
    ak.max_keypoints = 128                       # far too small: every list overflows
    kp, d = ak.extract_arrays(img)
    assert ak.max_keypoints >= len(want_kp) and ak.max_keypoints <= 4096
    _eq(kp.view(np.uint8), want_kp.view(np.uint8), "keypoints after growth")
    _eq(d, want_d, "descriptors after growth")
    # the raw context still reports the overflow honestly
    small = akaze.Akaze.default(); small.max_keypoints = 128
    ctx = small.context(640, 360, 1)
    with pytest.raises(akaze.AkzError) as e:
        ctx.extract_batch([img])
    assert e.value.status == -7 and "max_keypoints" in str(e.value)


def test_dense_full_hd_noise_beyond_the_old_limit(gpu, oracle):
    """Round-2 verdict, missing 5: `Akaze::dense()` on 1080p noise holds 82 000 keypoints in one frame — more than the
    65 536 a context could be created for.  The limit is 262 144 now, and the host mirror reaches it by growth from its
    default capacity; keypoints and descriptors equal the oracle's (every sort on its global-memory path)."""
    akaze, _ = gpu
    rng = np.random.default_rng(7)
    rng.integers(0, 256, (960, 1280), dtype=np.uint8)
    img = rng.integers(0, 256, (1080, 1920), dtype=np.uint8)
    okp, od = oracle.Akaze(1920, 1080, oracle.default_config(threshold=0.0001)).extract(img)
    assert len(okp) > 65536
    ak = akaze.Akaze.dense()                    # default first capacity (16 384): three doublings on the way
    kp, d = ak.extract_arrays(img)
    assert ak.max_keypoints == 131072
    _kp_eq(kp, okp, "dense 1080p noise keypoints")
    _eq(d, od, "dense 1080p noise descriptors")


def test_active_list_beyond_the_lds_limit(gpu, oracle):
    """Round-3 verdict, missing 7: the serial suppression pass kept its active list (the cache entries of two adjacent
    classes) in LDS, 8 192 entries; a frame that fell back to it AND outgrew that list ended in AKZ_E_INTERNAL, where the
    reference's Vec just grows (scale_space_extrema.rs:61-140).  Such a frame now starts over in k_suppress_big (list in
    global memory): dense noise, every frame forced through the serial pass, 20 000+ keypoints of which far more than 8 192
    belong to the first two levels — keypoints and descriptors equal the oracle's, and a frame that stays below the limit
    in the same batch is untouched."""
    akaze, _ = gpu
    rng = np.random.default_rng(11)
    w, h = 960, 544
    noise = rng.integers(0, 256, (h, w), dtype=np.uint8)
    quiet = synth_frame(w, h, 5)
    cfg = oracle.default_config(threshold=0.0001)
    orc = oracle.Akaze(w, h, cfg)
    okp, od = orc.extract(noise)
    first_two = int(((okp["class_id"] == 0) | (okp["class_id"] == 1)).sum())
    assert len(okp) > 16384 and first_two > 8192, (len(okp), first_two)
    ak = akaze.Akaze.dense()
    ak.max_keypoints = 65536
    ctx = akaze.Context(ak, w, h, 2, _opts(parallel_suppression=False))
    res = ctx.extract_batch([noise, quiet])
    _kp_eq(res[0][0], okp, "active list beyond the LDS limit: keypoints")
    _eq(res[0][1], od, "active list beyond the LDS limit: descriptors")
    qkp, qd = orc.extract(quiet)
    _kp_eq(res[1][0], qkp, "the frame beside it")
    _eq(res[1][1], qd, "the frame beside it: descriptors")
    ctx.close()
    # the same through the parallel pass's own hand-over: its lists sized far too small, so the device flags the frame,
    # the serial pass takes it, overflows its LDS list and hands it on
    ctx = akaze.Context(ak, w, h, 1, _opts(sup_capacity=1024))
    res = ctx.extract_batch([noise])
    _kp_eq(res[0][0], okp, "flagged -> serial -> big: keypoints")
    _eq(res[0][1], od, "flagged -> serial -> big: descriptors")
    ctx.close()


def test_the_three_unvendored_arithmetic_orders(gpu, oracle, kitti):
    """Round-3 verdict, next 4.  wide::f32x4::reduce_add, wide::f32x4::mul_add and ndarray's 2 x 2 sum() (akaze/src/image.rs:
    160-195, 242-247, 320-325) live in crates the reference does not vendor, and its pins come out the same under all eight
    combinations (tests/test_oracle_pins.py).  The library therefore carries all eight as akz_options.arith: every
    combination == the oracle under the same ORC_OPT_* switches — every pyramid buffer and stage on a ragged frame (odd
    width: the one-frame kernels and the odd-edge rules of half_size), keypoints and descriptors on the KITTI pair (frame
    pairs, the fused kernels), the stand-alone filters and half_size — and the combinations really differ from one another
    (otherwise the test would prove nothing)."""
    akaze, _ = gpu
    O = oracle
    ragged = synth_frame(333, 219, 21)
    taps = [O.gaussian_kernel(1.0, 7), O.gaussian_kernel(10.0, 71), O.gaussian_kernel(1.6, 9)]
    imgf = O.u8_to_f32(synth_frame(200, 150, 3))
    seen = {}
    try:
        for arith in range(8):
            O.set_option(O.OPT_REDUCE, arith & 1)
            O.set_option(O.OPT_FMA, (arith >> 1) & 1)
            O.set_option(O.OPT_HALFSUM, (arith >> 2) & 1)
            # every buffer and stage, one-frame kernels
            ctx = akaze.Context(akaze.Akaze.default(), 333, 219, 1, _opts(keep_all=True, arith=arith))
            res = ctx.extract_batch([ragged])
            orc = O.Akaze(333, 219, O.default_config())
            okp, od = orc.extract(ragged)
            assert ctx.contrast(0) == orc.contrast, (arith, ctx.contrast(0), orc.contrast)
            for lvl in range(orc.num_levels):
                for name in ("Lt", "Lsmooth", "Lflow", "Lx", "Ly", "Ldet"):
                    if lvl == 0 and name == "Lflow":
                        continue
                    _eq(ctx.level_buffer(0, lvl, name, 333, 219), orc.buffer(lvl, name), f"arith {arith} {name}[{lvl}]")
            for stage in (0, 1, 2):
                _kp_eq(ctx.keypoints(0, stage), orc.keypoints(stage), f"arith {arith} stage{stage}")
            _kp_eq(res[0][0], okp, f"arith {arith} ragged keypoints")
            _eq(res[0][1], od, f"arith {arith} ragged descriptors")
            sig = [ctx.level_buffer(0, 5, "Lt", 333, 219).tobytes()]
            # the stand-alone image API of the same context
            for k in taps:
                _eq(akaze.horizontal_filter(imgf, k, ctx), O.horizontal_filter(imgf, k), f"arith {arith} horizontal {len(k)}")
                _eq(akaze.vertical_filter(imgf, k, ctx), O.vertical_filter(imgf, k), f"arith {arith} vertical {len(k)}")
            _eq(akaze.half_size(imgf, ctx), O.half_size(imgf), f"arith {arith} half_size")
            sig.append(akaze.horizontal_filter(imgf, taps[1], ctx).tobytes())
            sig.append(akaze.half_size(imgf, ctx).tobytes())
            ctx.close()
            # the measured configuration's kernels (frame pairs, fused front end + FED, streaming determinant): default options
            ctx = akaze.Context(akaze.Akaze.sparse(), 1392, 512, 2, _opts(arith=arith))
            res = ctx.extract_batch([kitti[0], kitti[1]])
            orc = O.Akaze(1392, 512, O.default_config(threshold=0.01))
            for i in range(2):
                okp, od = orc.extract(kitti[i])
                _kp_eq(res[i][0], okp, f"arith {arith} kitti {i} keypoints")
                _eq(res[i][1], od, f"arith {arith} kitti {i} descriptors")
            assert (len(res[0][1]), len(res[1][1])) == (399, 343)        # the reference's pins hold under every combination
            ctx.close()
            seen[arith] = sig
    finally:
        for o in (O.OPT_REDUCE, O.OPT_FMA, O.OPT_HALFSUM):
            O.set_option(o, 0)
    # each switch changes something: the pyramid under reduce / fma, half_size under the 2 x 2 order
    assert seen[0][0] != seen[1][0] and seen[0][0] != seen[2][0] and seen[0][1] != seen[1][1] and seen[0][1] != seen[2][1]
    assert seen[0][2] != seen[4][2] and seen[0][1] == seen[4][1]


def test_linear_knn_keeps_its_targets_on_the_device(gpu, oracle):
    """space::LinearKnn is built once per frame pair and asked once per query (akaze/tests/estimate_pose.rs:82-88).
    hm_set_targets / hm_knn_targets keep `iter` resident between the questions: per-query answers == the oracle, k = 1..3, a
    batch of questions in one call == the per-query answers, and a call that overwrites the matcher's staging buffer in between
    is noticed (AKZ_E_INVALID from the C ABI, a silent re-upload in the host mirror) instead of searching stale data."""
    _, knn = gpu
    from cv_amd import _lib
    rng = np.random.default_rng(0x71)
    t = _rand_desc(rng, 700); q = _rand_desc(rng, 40)
    t[10] = t[500] = q[3]                                     # ties: the lowest index wins
    lk = knn.LinearKnn(knn.Hamming, t)
    for k in (1, 2, 3):
        want = oracle.knn(q, t, k)
        for i in range(len(q)):
            got = lk.knn(q[i], k)
            assert [(n.index, n.distance) for n in got] == [(int(want[i, j]["index"]), int(want[i, j]["distance"])) for j in range(k)], (k, i)
    m = knn.default_matcher(len(t))
    m.set_targets(t)
    gb = m.knn_targets(q, 3)
    wb = oracle.knn(q, t, 3)
    _eq(gb["index"], wb["index"], "resident batch idx"); _eq(gb["distance"], wb["distance"], "resident batch dist")
    other = _rand_desc(rng, 300)
    m.match(q, other)                                         # takes the staging buffer
    out = np.zeros((1, 2), _lib.NB_DTYPE)
    assert _lib.lib().hm_knn_targets(m.handle, q.ctypes.data, 1, 2, out.ctypes.data) == -1
    got = lk.knn(q[0], 2)                                     # the mirror uploads again
    assert [(n.index, n.distance) for n in got] == [(int(wb[0, j]["index"]), int(wb[0, j]["distance"])) for j in range(2)]
    # fewer targets than neighbours asked for: min(k, len) come back
    small = knn.LinearKnn(knn.Hamming, t[:2])
    assert len(small.knn(q[0], 3)) == 2
    # two LinearKnn objects asked in turn on the SAME matcher (same length, even: a pointer / length comparison cannot tell
    # them apart): each question is answered from its own set — the upload's generation number decides, not a heuristic
    t2 = _rand_desc(rng, 700)
    la, lb = knn.LinearKnn(knn.Hamming, t), knn.LinearKnn(knn.Hamming, t2)
    wa, wb2 = oracle.knn(q, t, 2), oracle.knn(q, t2, 2)
    for i in range(6):
        for lk_, w in ((la, wa), (lb, wb2), (la, wa)):
            got = lk_.knn(q[i], 2)
            assert [(n.index, n.distance) for n in got] == [(int(w[i, j]["index"]), int(w[i, j]["distance"])) for j in range(2)], i
    g0 = m.targets_generation()
    la.knn(q[0], 2); la.knn(q[1], 2)                          # (la asked last: no further upload)
    assert m.targets_generation() == g0 != 0
    # the caller mutating ITS array afterwards does not reach the object (private copy), assigning `iter` forgets the upload
    src = t.copy()
    lc = knn.LinearKnn(knn.Hamming, src)
    first = lc.knn(q[5], 2)
    src[:] = t2
    assert [(n.index, n.distance) for n in lc.knn(q[5], 2)] == [(n.index, n.distance) for n in first]
    lc.iter = t2
    assert [(n.index, n.distance) for n in lc.knn(q[5], 2)] == [(int(wb2[5, j]["index"]), int(wb2[5, j]["distance"])) for j in range(2)]
    # a matcher destroyed and re-created between two LinearKnn values (the Rust shim's with_matcher does exactly that when it
    # grows; the new hm_ctx is a same-size allocation right after a delete and tends to land at the SAME address): the
    # (context pointer, generation) key of an upload must never repeat, so generations are process-wide, not per context
    seen, handles = set(), set()
    for rnd in range(6):
        mm = knn.Matcher(1024)
        a_set, b_set = (t[:16], t2[:16]) if rnd % 2 == 0 else (t2[:16], t[:16])
        ga = mm.set_targets(a_set)
        assert ga not in seen and ga > max(seen | {g0}), (ga, seen)
        seen.add(ga)
        handles.add(mm.handle.value)
        # what a (pointer, generation)-keyed mirror of the PREVIOUS round would do on this context: its key is stale
        want = oracle.knn(q[:4], a_set, 2)
        got = mm.knn_targets(q[:4], 2)
        _eq(got["index"], want["index"], "recreated matcher idx"); _eq(got["distance"], want["distance"], "recreated matcher dist")
        mm.close()
    assert len(seen) == 6          # (handles usually holds ONE address: the case the per-context counter got wrong)


def test_new_entry_points_refuse_what_they_cannot_do(gpu):
    """Round-3 entry points answer with a status, never with a wrong result: the batched consensus (scenes beyond the
    reservation, unknown flags, a shuffle that would not fit its LDS sort, a stale parameter struct), the colour arm
    (two channels, a stride shorter than a row), the exchange (rank outside the world)."""
    import ctypes as C
    import torch
    akaze, _ = gpu
    from cv_amd import _lib
    from cv_amd.ransac import EssentialConsensus
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    cons = EssentialConsensus(512, 64)
    cons.reserve(2)
    cam = cons.camera((1000.0, 1000.0, 320.0, 240.0, 0.0, None))
    prm = cons.make_params(1e-6, n_hypotheses=64, block_size=32)
    kp = torch.zeros((4, 512, 28), dtype=torch.uint8, device=dev)
    pairs = torch.zeros((4, 512, 2), dtype=torch.int32, device=dev)
    npairs = torch.zeros((4,), dtype=torch.int32, device=dev)
    out = [torch.zeros((4, 12), dtype=torch.float64, device=dev), torch.zeros(4, dtype=torch.int32, device=dev),
           torch.zeros((4, 512), dtype=torch.int32, device=dev), torch.zeros(4, dtype=torch.int32, device=dev)]

    def call(n_scenes, flags=0, cap=512, p=prm, camera=cam):
        ia = (C.c_uint32 * max(1, n_scenes))(*range(n_scenes))
        return L.rs_essential_arrsac_batch_device(cons._h, kp.data_ptr(), kp.data_ptr(), cap, ia, ia, pairs.data_ptr(), npairs.data_ptr(),
                                                  n_scenes, C.byref(camera), C.byref(camera), C.byref(p), flags, out[0].data_ptr(),
                                                  out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), None, None)
    assert call(2) == 0 and call(0) == 0
    cons.sync()
    assert out[1].cpu().numpy().view(np.uint32)[0] == 0xFFFFFFFF          # no matches: no model
    assert call(3) == -6                                                   # AKZ_E_TOO_LARGE: beyond rs_batch_reserve
    assert call(2, flags=2) == -1                                          # unknown flag
    bad = cons.make_params(1e-6, n_hypotheses=64, block_size=32); bad.struct_size = 8
    assert call(2, p=bad) == -1
    big = cons.make_params(1e-6, n_hypotheses=65, block_size=32)
    assert call(2, p=big) == -6                                            # more hypotheses than the context holds
    cam2 = cons.camera((1000.0, 1000.0, 320.0, 240.0, 0.0, None)); cam2.reserved = 1
    assert call(2, camera=cam2) == -1
    wide = EssentialConsensus(16384, 16)
    wide.reserve(1)
    kpw = torch.zeros((1, 16384, 28), dtype=torch.uint8, device=dev)
    prw = torch.zeros((1, 16384, 2), dtype=torch.int32, device=dev)
    one = (C.c_uint32 * 1)(0)
    pw = wide.make_params(1e-6, n_hypotheses=16, block_size=64)
    st = L.rs_essential_arrsac_batch_device(wide._h, kpw.data_ptr(), kpw.data_ptr(), 16384, one, one, prw.data_ptr(), npairs.data_ptr(), 1,
                                            C.byref(cam), C.byref(cam), C.byref(pw), _lib.RS_BATCH_SHUFFLE, out[0].data_ptr(),
                                            out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), None, None)
    assert st == -6                                                        # the seeded shuffle sorts at most 8192 keys in LDS
    # colour arm
    ctx = akaze.Akaze.default().context(64, 48, 1)
    img = np.zeros((48, 64, 3), np.uint8)
    k = np.zeros(16, _lib.KP_DTYPE); d = np.zeros((16, 64), np.uint8); n = C.c_uint32()
    assert L.akz_extract_color(ctx.handle, img.ctypes.data, 0, 2, 64, 48, 128, k.ctypes.data, d.ctypes.data, 16, C.byref(n)) == -1
    assert L.akz_extract_color(ctx.handle, img.ctypes.data, 0, 3, 64, 48, 100, k.ctypes.data, d.ctypes.data, 16, C.byref(n)) == -1
    assert L.akz_extract_color(ctx.handle, img.ctypes.data, 5, 3, 64, 48, 192, k.ctypes.data, d.ctypes.data, 16, C.byref(n)) == -1
    assert L.akz_extract_color(ctx.handle, img.ctypes.data, 0, 3, 64, 48, 192, k.ctypes.data, d.ctypes.data, 16, C.byref(n)) == 0 and n.value == 0
    # exchange
    ident = (C.c_uint8 * 128)()
    assert L.akz_comm_create(ident, 2, 2, 0, C.byref(C.c_void_p())) == -1
    assert L.akz_comm_create(ident, 0, 0, 0, C.byref(C.c_void_p())) == -1
    assert L.akz_comm_unique_id(None) == -1
    # the registration consensus: the same refusals, and its own minimum (three matches)
    world = torch.zeros((8, 4), dtype=torch.float64, device=dev)

    def reg(n_scenes, flags=0, cap=512, p=prm, camera=cam, w=world):
        ik = (C.c_uint32 * max(1, n_scenes))(*range(n_scenes))
        return L.rs_p3p_arrsac_batch_device(cons._h, kp.data_ptr(), cap, ik, pairs.data_ptr(), npairs.data_ptr(), n_scenes,
                                            w.data_ptr() if w is not None else None, 8, C.byref(camera), C.byref(p), flags,
                                            out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), None, None)
    assert reg(2) == 0 and reg(0) == 0
    cons.sync()
    assert out[1].cpu().numpy().view(np.uint32)[1] == 0xFFFFFFFF
    assert reg(3) == -6 and reg(2, flags=4) == -1 and reg(2, p=bad) == -1 and reg(2, camera=cam2) == -1 and reg(2, w=None) == -1
    assert reg(2, cap=2) == -1                                             # fewer than a minimal sample can ever hold
    # the batched matcher entries
    from cv_amd import knn as knn_mod
    m = knn_mod.Matcher(512)
    dsc = torch.zeros((2, 512, 64), dtype=torch.uint8, device=dev)
    cnt = torch.zeros((2,), dtype=torch.int32, device=dev)
    nbr = torch.zeros((2, 512, 3, 2), dtype=torch.int32, device=dev)
    two = (C.c_uint32 * 2)(0, 1)
    kb = lambda n_probs, k: L.hm_knn_batch_device(m.handle, dsc.data_ptr(), cnt.data_ptr(), dsc.data_ptr(), cnt.data_ptr(), 512, two, two,
                                                  n_probs, k, nbr.data_ptr(), None)
    assert kb(2, 3) == 0 and kb(0, 2) == 0 and kb(2, 4) == -1 and kb(2, 0) == -1 and kb(70000, 2) == -1
    bov = lambda n_views, k, n_frames=1: L.hm_best_of_views_batch_device(
        m.handle, nbr.data_ptr(), cnt.data_ptr(), two, 512, two, n_frames, n_views, k, pairs.data_ptr(), cnt.data_ptr(), 24,
        nbr.data_ptr(), pairs.data_ptr(), None)
    assert bov(2, 3) == 0 and bov(0, 3) == -1 and bov(65, 3) == -1 and bov(2, 4) == -1 and bov(2, 3, 70000) == -1
    _lib.check(L.hm_sync(m.handle), "hm_sync")
    assert L.rs_debug_far(cons._h, None, 1, None, None, 1, 1e-7, None) == -1
