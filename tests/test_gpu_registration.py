"""The registration path of cv-sfm for a micro-batch of new frames, chained on the device (cv_amd/registration.py):
extract -> hash_bag -> knn(., 3) against the recent views -> best-of-views -> (feature, landmark) pair lists -> Lambda Twist
ARRSAC.  cv-sfm/src/lib.rs:672, 1462-1542, 1549-1604, 1619-1622.  Every stage of the chain is held to the oracle on the
GPU's own intermediate data, and the poses that come out are the camera motion the synthetic frames were made with."""
import numpy as np
import pytest

from conftest import synth_frame

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from cv_amd import build
    build.build()
    return torch


def pan_world(seed, w, h, n, dx, dy):
    """n frames of w x h: a camera panning (dx, dy) pixels per frame over one canvas, +-2 sensor noise."""
    canvas = synth_frame(w + dx * n + 16, h + dy * n + 16, seed, n_rect=140, n_disc=140)
    rng = np.random.default_rng(seed + 1)
    out = np.zeros((n, h, w), np.uint8)
    for g in range(n):
        crop = canvas[dy * g:dy * g + h, dx * g:dx * g + w].astype(np.int16)
        out[g] = np.clip(crop + rng.integers(-2, 3, (h, w)), 0, 255).astype(np.uint8)
    return out


def landmark_keys(kps, g, dx, dy, cell, wc):
    """The synthetic control plane: the landmark a feature observes = the canvas cell its keypoint falls into, per
    evolution level (the same canvas feature seen from two frames lands in the same cell, give or take its edges)."""
    xw = kps["x"].astype(np.float64) + dx * g
    yw = kps["y"].astype(np.float64) + dy * g
    cx = np.clip(np.floor(xw / cell), 0, wc - 1).astype(np.int64)
    cy = np.floor(yw / cell).astype(np.int64)
    return ((cy * wc + cx) * 16 + (kps["class_id"].astype(np.int64) & 15)).astype(np.uint32)


def world_table(wc, hc, cell, f, cx0, cy0, z0):
    """World point of every landmark key: the cell centre on the plane Z = z0 (fronto-parallel canvas seen by a camera
    with focal f translating parallel to it), in the reference's Projective form."""
    keys = np.arange(wc * hc * 16)
    c = keys // 16
    x = ((c % wc) + 0.5) * cell
    y = ((c // wc) + 0.5) * cell
    P = np.stack([(x - cx0) * z0 / f, (y - cy0) * z0 / f, np.full(len(keys), z0), np.ones(len(keys))], 1)
    return P / np.linalg.norm(P[:, :3], axis=1, keepdims=True)


def test_registration_chain_of_a_micro_batch(gpu, oracle):
    torch = gpu
    from cv_amd import _lib
    from cv_amd.akaze import Akaze
    from cv_amd.registration import Registration
    W, H, DX, DY = 640, 360, 4, 2
    V, F = 4, 8
    NB = V + F
    CAP, CELL = 2048, 6
    dev = torch.device("cuda", 0)
    frames = pan_world(0x2E61, W, H, NB, DX, DY)
    ak = Akaze.default()
    ak.max_keypoints = CAP
    ctx = ak.context(W, H, NB)
    L = _lib.lib()
    d_frames = torch.from_numpy(frames).to(dev)
    d_kps = torch.zeros((NB, CAP, 28), dtype=torch.uint8, device=dev)
    d_descs = torch.zeros((NB, CAP, 64), dtype=torch.uint8, device=dev)
    d_counts = torch.zeros((NB,), dtype=torch.int32, device=dev)
    _lib.check(L.akz_extract_batch_device(ctx.handle, d_frames.data_ptr(), 0, NB, W, H, d_kps.data_ptr(), d_descs.data_ptr(), CAP,
                                          d_counts.data_ptr(), _lib.wait_handle(torch.cuda.current_stream())), "extract")
    _lib.check(L.akz_sync(ctx.handle), "akz_sync")
    counts = d_counts.cpu().numpy()
    assert counts.min() > 300, counts
    kps = d_kps.cpu().numpy().view(_lib.KP_DTYPE).reshape(NB, CAP)
    descs = d_descs.cpu().numpy()
    # the caller's bookkeeping: landmarks of every stored feature, the table of triangulated landmarks
    f_cam, z0 = 700.0, 5.0
    cam = (f_cam, f_cam, W / 2.0, H / 2.0, 0.0, None)
    wc, hc = (W + DX * NB) // CELL + 2, (H + DY * NB) // CELL + 2
    landmarks = np.zeros((NB, CAP), np.uint32)
    for g in range(NB):
        landmarks[g, :counts[g]] = landmark_keys(kps[g, :counts[g]], g, DX, DY, CELL, wc)
    world = world_table(wc, hc, CELL, f_cam, W / 2.0, H / 2.0, z0)
    n_world = len(world)
    rng = np.random.default_rng(5)
    world[rng.random(n_world) < 0.05, 3] = -1.0                  # landmarks without a robust triangulation
    d_lm = torch.from_numpy(landmarks.view(np.int32)).to(dev)
    d_world = torch.from_numpy(world).to(dev)
    codewords = rng.integers(0, 256, (256, 64), dtype=np.uint8)
    kw = dict(block_size=32, max_candidates=64, estimations_per_block=16)
    thr, n_hyp = 2e-5, 256
    reg = Registration(torch, CAP, F, V, codewords, cam, threshold=thr, n_hypotheses=n_hyp, seed=11, **kw)
    frame_blocks = [V + f for f in range(F)]
    view_blocks = [[V + f - 1 - v for v in range(V)] for f in range(F)]
    for rep in range(2):        # twice: the second call runs over the first one's leftovers
        reg.enqueue(d_kps, d_descs, d_counts, frame_blocks, view_blocks, d_lm, d_world, n_world,
                    stream_to_wait=_lib.wait_handle(torch.cuda.current_stream()))
    reg.sync()
    g_hash = reg.hash.cpu().numpy(); g_knn = reg.knn.cpu().numpy(); g_best = reg.best.cpu().numpy().view(np.uint32)
    g_dec = reg.decision.cpu().numpy().view(np.uint32); g_pairs = reg.pairs.cpu().numpy().view(np.uint32)
    g_np = reg.npairs.cpu().numpy().view(np.uint32); g_pose = reg.pose.cpu().numpy(); g_id = reg.best_id.cpu().numpy().view(np.uint32)
    g_inl = reg.inliers.cpu().numpy().view(np.uint32); g_ninl = reg.n_inliers.cpu().numpy().view(np.uint32)
    models = 0
    for f in range(F):
        b = frame_blocks[f]
        n = int(counts[b])
        # place-recognition hash of the new frame (cv-sfm/src/lib.rs:672)
        want_hash, _ = oracle.hash_bag(descs[b, :n], codewords)
        assert np.array_equal(g_hash[b - reg.hash_block0], want_hash), f
        # knn(., 3) against every view, best-of-views, decisions (:1462-1532)
        gnb = np.zeros((V, CAP, 3), _lib.NB_DTYPE)
        gnb["index"] = g_knn[f, ..., 0]; gnb["distance"] = g_knn[f, ..., 1]
        for v in (0, V - 1):
            t = view_blocks[f][v]
            want = oracle.knn(descs[b, :n], descs[t, :counts[t]], 3)
            assert np.array_equal(gnb["index"][v, :n], want["index"]) and np.array_equal(gnb["distance"][v, :n], want["distance"]), (f, v)
        wbest, wdec = oracle.best_of_views(gnb, n, landmarks, np.array(view_blocks[f], np.uint32), counts.astype(np.uint32), 24)
        assert np.array_equal(g_best[f, :n], wbest) and np.array_equal(g_dec[f, :n], wdec), f
        # the FeatureWorldMatch list (:1516-1520, 1549-1563, 1583-1604)
        wpairs = oracle.landmark_pairs(wbest, wdec, world)
        assert g_np[f] == len(wpairs) and np.array_equal(g_pairs[f, :g_np[f]], wpairs), (f, g_np[f], len(wpairs))
        assert len(wpairs) > 40, (f, len(wpairs))
        # the consensus (:1619-1622) on exactly that list
        want = oracle.p3p_arrsac_pairs(kps[b], wpairs, world, cam, thr, n_hyp, scene=f, shuffle=True, seed=11, init_blocks=1,
                                       halve=True, sprt=True, **kw)
        assert g_id[f] == want["best_id"] and g_ninl[f] == len(want["inliers"]), (f, g_id[f], want["best_id"])
        if want["best_id"] == 0xFFFFFFFF:
            continue
        models += 1
        assert g_pose[f].tobytes() == want["pose"].tobytes(), f
        assert np.array_equal(g_inl[f, :g_ninl[f]], want["inliers"]), f
        # ... and it is the pose the frame was rendered from: identity rotation, the camera DX * g, DY * g pixels along the canvas
        R, t = g_pose[f].reshape(3, 4)[:, :3], g_pose[f].reshape(3, 4)[:, 3]
        expect_t = -np.array([DX * b * z0 / f_cam, DY * b * z0 / f_cam, 0.0])
        # (cell centres stand in for triangulated landmarks: +-3 px at f = 700 on a plane 5 units away)
        assert np.abs(R - np.eye(3)).max() < 0.08 and np.abs(t - expect_t).max() < 0.4, (f, R, t, expect_t)
        assert g_ninl[f] > 0.3 * len(wpairs)
    assert models >= F - 1
    reg.close()


def test_landmark_pairs_rules(gpu, oracle):
    """hm_landmark_pairs_batch_device on constructed decisions: a landmark claimed twice loses both matches, decision 2 and 0
    never become a match, an untriangulated or out-of-table landmark is dropped, an empty frame gives an empty list, ascending
    feature order — every frame equal to the oracle."""
    import ctypes as C
    torch = gpu
    from cv_amd import _lib
    from cv_amd.knn import Matcher
    rng = np.random.default_rng(0x1A9D)
    cap, F, n_world = 1024, 6, 5000
    nq = np.array([1024, 0, 1, 700, 64, 1024], np.int32)
    best = np.zeros((F, cap, 3, 2), np.uint32)
    best[..., 0] = rng.integers(0, n_world + 50, (F, cap, 3))          # some keys beyond the table
    best[..., 1] = rng.integers(0, 300, (F, cap, 3))
    best[0, :, 0, 0] = rng.integers(0, 600, cap)                        # frame 0: heavy duplication
    best[5, :, 0, 0] = np.arange(cap) * 3 % n_world                     # frame 5: all distinct
    best[3, 10, 0, 0] = 0xFFFFFFFF                                      # an absent landmark
    dec = rng.integers(0, 3, (F, cap)).astype(np.uint32)
    dec[5] = 1
    world = rng.standard_normal((n_world, 4))
    world[:, 3] = np.abs(world[:, 3])
    world[rng.random(n_world) < 0.2, 3] = -1.0
    dev = torch.device("cuda", 0)
    d_best = torch.from_numpy(best.view(np.int32)).to(dev); d_dec = torch.from_numpy(dec.view(np.int32)).to(dev)
    d_nq = torch.from_numpy(nq).to(dev); d_world = torch.from_numpy(world).to(dev)
    d_pairs = torch.full((F, cap, 2), -1, dtype=torch.int32, device=dev); d_np = torch.full((F,), 77, dtype=torch.int32, device=dev)
    m = Matcher(cap)
    iq = np.arange(F, dtype=np.uint32)
    L = _lib.lib()
    _lib.check(L.hm_landmark_pairs_batch_device(m.handle, d_best.data_ptr(), d_dec.data_ptr(), d_nq.data_ptr(), iq.ctypes.data_as(C.c_void_p),
                                                cap, F, d_world.data_ptr(), n_world, d_pairs.data_ptr(), d_np.data_ptr(),
                                                _lib.wait_handle(torch.cuda.current_stream())), "landmark_pairs")
    _lib.check(L.hm_sync(m.handle), "hm_sync")
    gp = d_pairs.cpu().numpy().view(np.uint32); gn = d_np.cpu().numpy()
    total = 0
    for f in range(F):
        n = int(nq[f])
        want = oracle.landmark_pairs(best[f, :n], dec[f, :n], world)
        assert gn[f] == len(want), (f, gn[f], len(want))
        assert np.array_equal(gp[f, :gn[f]], want), f
        assert (gp[f, gn[f]:] == 0xFFFFFFFF).all()                      # nothing written past the list
        total += len(want)
    assert gn[1] == 0 and total > 300
    # refusals
    assert L.hm_landmark_pairs_batch_device(m.handle, d_best.data_ptr(), d_dec.data_ptr(), d_nq.data_ptr(), iq.ctypes.data_as(C.c_void_p),
                                            16384, F, d_world.data_ptr(), n_world, d_pairs.data_ptr(), d_np.data_ptr(), None) == -6
    assert L.hm_landmark_pairs_batch_device(m.handle, None, d_dec.data_ptr(), d_nq.data_ptr(), iq.ctypes.data_as(C.c_void_p),
                                            cap, F, d_world.data_ptr(), n_world, d_pairs.data_ptr(), d_np.data_ptr(), None) == -1
    assert L.hm_landmark_pairs_batch_device(m.handle, d_best.data_ptr(), d_dec.data_ptr(), d_nq.data_ptr(), iq.ctypes.data_as(C.c_void_p),
                                            cap, F, d_world.data_ptr(), 0, d_pairs.data_ptr(), d_np.data_ptr(), None) == -1


def test_landmark_matches_with_merge_candidates(gpu, oracle):
    """hm_landmark_matches_batch_device with the caller's merge verdicts (cv-sfm/src/lib.rs:1521-1531): an accepted decision-2
    feature is the match ([best0, best1], feature); landmark_counts covers BOTH of its landmarks (:1549-1552), so a decision-1
    match whose landmark a merge also claims is dropped and so is the merge, and a surviving merge leaves with its merged
    world row n_world + f * cap + feature — every frame equal to the oracle, and the mask-less call equal to a zero mask."""
    import ctypes as C
    torch = gpu
    from cv_amd import _lib
    from cv_amd.knn import Matcher
    rng = np.random.default_rng(0x3E46E)
    cap, F, n_world = 2048, 5, 6000
    nq = np.array([2048, 1500, 0, 2048, 300], np.int32)
    best = np.zeros((F, cap, 3, 2), np.uint32)
    # mostly distinct landmarks per feature (as after the per-feature dedup), with collisions between features
    for f in range(F):
        for j in range(cap):
            best[f, j, :, 0] = rng.choice(n_world + 40, 3, replace=False)
    best[..., 1] = rng.integers(0, 300, (F, cap, 3))
    best[0, :, 0, 0] = rng.permutation(n_world)[:cap]                  # frame 0: first landmarks all distinct ...
    best[0, :, 1, 0] = (best[0, :, 0, 0] + 1 + rng.integers(0, 3, cap)) % n_world   # ... seconds collide with other firsts
    best[3, 7, 1, 0] = 0xFFFFFFFF
    dec = rng.integers(0, 3, (F, cap)).astype(np.uint32)
    merge_ok = (rng.random((F, cap)) < 0.6).astype(np.uint8)
    world = rng.standard_normal((n_world + F * cap, 4))
    world[:, 3] = np.abs(world[:, 3])
    world[rng.random(len(world)) < 0.2, 3] = -1.0
    dev = torch.device("cuda", 0)
    d_best = torch.from_numpy(best.view(np.int32)).to(dev); d_dec = torch.from_numpy(dec.view(np.int32)).to(dev)
    d_ok = torch.from_numpy(merge_ok).to(dev)
    d_nq = torch.from_numpy(nq).to(dev); d_world = torch.from_numpy(world).to(dev)
    m = Matcher(cap)
    iq = np.arange(F, dtype=np.uint32)
    L = _lib.lib()

    def run(mask):
        d_pairs = torch.full((F, cap, 2), -1, dtype=torch.int32, device=dev); d_np = torch.full((F,), 77, dtype=torch.int32, device=dev)
        _lib.check(L.hm_landmark_matches_batch_device(m.handle, d_best.data_ptr(), d_dec.data_ptr(), None if mask is None else mask.data_ptr(),
                                                      d_nq.data_ptr(), iq.ctypes.data_as(C.c_void_p), cap, F, d_world.data_ptr(), n_world,
                                                      d_pairs.data_ptr(), d_np.data_ptr(), _lib.wait_handle(torch.cuda.current_stream())), "landmark_matches")
        _lib.check(L.hm_sync(m.handle), "hm_sync")
        return d_pairs.cpu().numpy().view(np.uint32), d_np.cpu().numpy()
    gp, gn = run(d_ok)
    merged = dropped_by_merge = 0
    for f in range(F):
        n = int(nq[f])
        want = oracle.landmark_pairs(best[f, :n], dec[f, :n], world, merge_ok=merge_ok[f, :n], n_world=n_world, merged_base=n_world + f * cap)
        plain = oracle.landmark_pairs(best[f, :n], dec[f, :n], world[:n_world])
        assert gn[f] == len(want), (f, gn[f], len(want))
        assert np.array_equal(gp[f, :gn[f]], want), f
        assert (gp[f, gn[f]:] == 0xFFFFFFFF).all()
        merged += int((want[:, 1] >= n_world).sum())
        dropped_by_merge += len(set(map(int, plain[:, 0])) - set(map(int, want[:, 0])))
    assert merged > 100 and dropped_by_merge > 50          # both effects of the merges are really exercised
    g0, n0 = run(torch.zeros_like(d_ok))
    g1, n1 = run(None)
    assert np.array_equal(n0, n1) and np.array_equal(g0, g1)
    # the order the reference's consensus sees (cv-sfm/src/lib.rs:1561-1574): a stable sort by descending summed observation
    # count, applied on the device (hm_landmark_matches_ordered_batch_device).  Few distinct counts: long runs of ties that
    # must keep their feature order; a frame of zero matches; counts whose sum needs more than 16 bits.
    for obs_kind in ("ties", "wide"):
        obs = (rng.integers(1, 5, n_world) if obs_kind == "ties" else rng.integers(0, 1 << 20, n_world)).astype(np.uint32)
        d_obs = torch.from_numpy(obs.view(np.int32)).to(dev)
        d_pairs = torch.full((F, cap, 2), -1, dtype=torch.int32, device=dev); d_np = torch.full((F,), 77, dtype=torch.int32, device=dev)
        _lib.check(L.hm_landmark_matches_ordered_batch_device(m.handle, d_best.data_ptr(), d_dec.data_ptr(), d_ok.data_ptr(), d_obs.data_ptr(),
                                                              d_nq.data_ptr(), iq.ctypes.data_as(C.c_void_p), cap, F, d_world.data_ptr(), n_world,
                                                              d_pairs.data_ptr(), d_np.data_ptr(), _lib.wait_handle(torch.cuda.current_stream())),
                   "landmark_matches_ordered")
        _lib.check(L.hm_sync(m.handle), "hm_sync")
        op, on = d_pairs.cpu().numpy().view(np.uint32), d_np.cpu().numpy()
        moved = 0
        for f in range(F):
            n = int(nq[f])
            want = oracle.landmark_pairs(best[f, :n], dec[f, :n], world, merge_ok=merge_ok[f, :n], n_world=n_world,
                                         merged_base=n_world + f * cap, obs_counts=obs)
            assert on[f] == len(want) == gn[f], (obs_kind, f)
            assert np.array_equal(op[f, :on[f]], want), (obs_kind, f)
            assert sorted(map(tuple, want.tolist())) == sorted(map(tuple, gp[f, :gn[f]].tolist()))     # the same matches, re-ordered
            moved += int((want[:, 0] != gp[f, :gn[f], 0]).sum())
        assert moved > 1000
    m.close()
