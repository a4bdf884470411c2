"""oracle/arrsac_oracle.c — the written specification of rs_essential_arrsac / rs_p3p_arrsac — checked on CPU against
the exhaustive-scoring oracle (oracle/ransac_oracle.c, itself pinned on the reference's eight-point tests) and for the
properties the retirement rules promise.  The `arrsac` crate is not vendored in the reference tree (SURVEY.md §8c:
parity with the crate is unpinned beyond the count pin of akaze/tests/estimate_pose.rs:75); what IS pinned here is
that every rule leaves the exhaustive winner in place where it provably must."""
import numpy as np
import pytest

from test_oracle_ransac import _projective, _rot


def _scene(rng, n, outlier_frac):
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
    return a, b, ~bad


def test_sampler_draws_distinct_indices_per_hypothesis(oracle):
    for k in (3, 8):
        seen = set()
        for h in range(64):
            s = oracle.arrsac_draw(7, h, 50, k)
            assert len(set(s.tolist())) == k and s.max() < 50
            seen.add(tuple(s.tolist()))
        assert len(seen) > 60                                  # streams differ per hypothesis
        assert np.array_equal(oracle.arrsac_draw(7, 5, 50, k), oracle.arrsac_draw(7, 5, 50, k))
        assert not np.array_equal(oracle.arrsac_draw(8, 5, 50, k), oracle.arrsac_draw(7, 5, 50, k))
    s = oracle.arrsac_draw(1, 0, 8, 8)                          # n == K: a permutation
    assert sorted(s.tolist()) == list(range(8))


def test_bound_only_equals_exhaustive_scoring(oracle):
    rng = np.random.default_rng(11)
    n, n_hyp, thr = 300, 200, 1e-7
    a, b, _ = _scene(rng, n, 0.3)
    samples = np.stack([oracle.arrsac_draw(0, h, n, 8) for h in range(n_hyp)])
    wpose, wbest, winl, _ = oracle.essential_batch(a, b, samples, thr)
    for bs in (32, 64, 300):
        pose, inl, best, st = oracle.arrsac(a, b, thr, n_hyp, seed=0, block_size=bs, max_candidates=0, sprt=False)
        assert best == wbest and np.array_equal(inl, winl)
        assert pose.tobytes() == wpose.tobytes()
        assert st["blocks"] == (n + bs - 1) // bs
    # caller samples == sampler samples
    pose2, inl2, best2, _ = oracle.arrsac(a, b, thr, n_hyp, sample_idx=samples, max_candidates=0, sprt=False)
    assert best2 == wbest and np.array_equal(inl2, winl)


def test_cap_halving_and_sprt_keep_the_winner_and_save_work(oracle):
    rng = np.random.default_rng(12)
    n, n_hyp, thr = 400, 400, 1e-7
    a, b, _ = _scene(rng, n, 0.3)
    ex = oracle.arrsac(a, b, thr, n_hyp, seed=2, max_candidates=0, sprt=False, bound=False)
    assert ex[3]["blocks"] == 1 and ex[3]["residuals_evaluated"] <= ex[3]["poses"] * n
    full = ex[3]["residuals_evaluated"]
    prev = full
    for kw in (dict(max_candidates=64, sprt=False), dict(max_candidates=64, sprt=True),
               dict(max_candidates=64, sprt=True, halve=True)):
        pose, inl, best, st = oracle.arrsac(a, b, thr, n_hyp, seed=2, block_size=32, init_blocks=2, **kw)
        assert best == ex[2] and np.array_equal(inl, ex[1])
        assert st["residuals_evaluated"] < prev or kw.get("halve")
        assert st["residuals_evaluated"] < 0.6 * full
        prev = st["residuals_evaluated"]
        if kw.get("halve"):
            assert st["survivors"] <= 2                      # 64 >> (blocks - init_blocks) has long reached 1 (+ ties: none)


def test_inlier_guided_resampling(oracle):
    """On a NOISY scene 64 initial hypotheses rarely hold a good model (a minimal sample of noisy matches); re-sampling
    among the inliers of the best pose so far finds one that explains nearly every true match.  The new hypotheses
    are numbered after the initial ones and their count shows in the statistics."""
    rng = np.random.default_rng(100)
    n, n_hyp, thr, E = 400, 64, 1e-4, 16
    a, b, good = _scene(rng, n, 0.3)
    b = b + rng.standard_normal(b.shape) * 2e-3
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    kw = dict(seed=5, block_size=50, init_blocks=1, max_candidates=16, sprt=False)
    plain = oracle.arrsac(a, b, thr, n_hyp, **kw)
    res = oracle.arrsac(a, b, thr, n_hyp, estimations_per_block=E, **kw)
    assert plain is not None and res is not None
    assert res[3]["poses"] == (n_hyp + E * (res[3]["blocks"] - 1)) * 4
    assert plain[3]["poses"] == n_hyp * 4
    assert len(plain[1]) < 0.5 * good.sum()
    assert len(res[1]) > 0.95 * good.sum() and res[2] >= n_hyp * 4     # the winner is a re-sampled hypothesis
    again = oracle.arrsac(a, b, thr, n_hyp, estimations_per_block=E, **kw)
    assert again[2] == res[2] and again[0].tobytes() == res[0].tobytes() and np.array_equal(again[1], res[1])


def test_p3p_shape(oracle):
    rng = np.random.default_rng(14)
    n, n_hyp, thr = 300, 300, 1e-6
    Rr = _rot(rng.random(3) * 0.8); tr = rng.random(3)
    pts = rng.random((n, 3)) * 4.0 - 2.0
    pts[:, 2] += 6.0
    cam = pts @ Rr.T + tr
    bb = cam / np.linalg.norm(cam, axis=1, keepdims=True)
    bad = rng.random(n) < 0.3
    rb = rng.standard_normal((n, 3)); rb[:, 2] = np.abs(rb[:, 2]) + 0.5
    bb[bad] = (rb / np.linalg.norm(rb, axis=1, keepdims=True))[bad]
    world = _projective(pts)
    samples = np.stack([oracle.arrsac_draw(3, h, n, 3) for h in range(n_hyp)])
    wpose, wbest, winl, _ = oracle.p3p_batch(bb, world, samples, thr)
    for kw in (dict(max_candidates=0, sprt=False), dict(max_candidates=64, sprt=True, halve=True, estimations_per_block=4)):
        pose, inl, best, st = oracle.arrsac(bb, world, thr, n_hyp, seed=3, p3p=True, block_size=32, init_blocks=2, **kw)
        assert np.array_equal(inl, winl) or len(inl) >= len(winl)
        if not kw.get("estimations_per_block"):
            assert best == wbest and pose.tobytes() == wpose.tobytes()
    assert np.abs(pose[:, :3] - Rr).max() < 1e-6


def test_cap_ranks_by_exact_distance_to_the_best(oracle):
    """The candidate cap must never retire the best-supported poses (round-2 advice): with more than 2047 inliers per
    good pose a histogram of raw counts would file them all in one bin and admit ties in pose order, retiring the
    later — here the re-sampled, better — ones.  The cap ranks by best - count, so the winner under a cap equals the
    winner without one."""
    rng = np.random.default_rng(77)
    n, n_hyp, thr = 3000, 24, 1e-7
    a, b, good = _scene(rng, n, 0.1)
    kw = dict(seed=4, block_size=2500, init_blocks=1, sprt=False, bound=False)
    free = oracle.arrsac(a, b, thr, n_hyp, max_candidates=0, **kw)
    capped = oracle.arrsac(a, b, thr, n_hyp, max_candidates=3, **kw)
    assert len(free[1]) > 2047                                    # the regime the old ranking got wrong
    assert capped[2] == free[2] and np.array_equal(capped[1], free[1])
    assert capped[3]["survivors"] <= 3 < free[3]["survivors"]


def test_halving_ends_with_its_last_survivor(oracle):
    """RS_PRUNE_HALVE without re-sampling: once the cap has reached 1 the loop ends — the survivor's pose and inliers
    are those a run to the last block would return, the statistics count only what ran."""
    rng = np.random.default_rng(78)
    n, n_hyp, thr = 600, 200, 1e-7
    a, b, _ = _scene(rng, n, 0.3)
    kw = dict(seed=1, block_size=16, init_blocks=1, max_candidates=8, sprt=False)
    h = oracle.arrsac(a, b, thr, n_hyp, halve=True, **kw)
    assert h[3]["survivors"] == 1 and h[3]["blocks"] == 1 + 3      # caps 8, 4, 2, 1 after blocks 1..4
    assert h[3]["residuals_evaluated"] < n_hyp * 4 * 16 + 15 * 16 + 1
    # the same survivor scored to the end (cap 1 from the fourth block on, no halving flag): same pose, same inliers
    samples = np.stack([oracle.arrsac_draw(1, hh, n, 8) for hh in range(n_hyp)])
    pose, best, inl, _ = oracle.essential_batch(a, b, samples[h[2] // 4:h[2] // 4 + 1], thr)
    assert best == h[2] % 4 or len(inl) >= len(h[1])
    assert h[0].tobytes() == oracle.essential_poses(oracle.eight_point(a[samples[h[2] // 4]], b[samples[h[2] // 4]]))[h[2] % 4].tobytes()


def _pixel_pairs(rng, n, cam, outlier_frac=0.3):
    from test_oracle_ransac import _rot as rot
    from oracle.oracle import KP_DTYPE
    R = rot((rng.random(3) - 0.5) * 0.3); t = (rng.random(3) - 0.5) * 0.6
    fx, fy, cx, cy, skew = cam[:5]
    pts = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.2, 1.2, n), rng.uniform(3, 9, n)], 1)

    def project(P):
        x, y = P[:, 0] / P[:, 2], P[:, 1] / P[:, 2]
        return np.stack([fx * x + skew * y + cx, fy * y + cy], 1)
    ka = np.zeros(n, KP_DTYPE); kb = np.zeros(n, KP_DTYPE)
    pa, pb = project(pts), project(pts @ R.T + t)
    ka["x"], ka["y"], kb["x"], kb["y"] = pa[:, 0], pa[:, 1], pb[:, 0], pb[:, 1]
    ib = np.arange(n)
    bad = rng.random(n) < outlier_frac
    ib[bad] = rng.integers(0, n, bad.sum())
    return ka, kb, np.stack([np.arange(n), ib], 1).astype(np.uint32), R, t, ~bad


def test_micro_batch_scene_entry(oracle):
    """orc_arrsac_pairs, the specification of one scene of rs_essential_arrsac_batch_device: calibrated bearings equal
    orc_calibrate's, the scoring order is a seeded permutation (stable by key), the scene index moves the seed, fewer
    than eight matches give no model, and on exact data the recovered rotation is the scene's (1e-6; f32 pixel
    coordinates are the only noise)."""
    rng = np.random.default_rng(79)
    cam = (984.2439, 980.8141, 690.0, 233.1966, 0.0, None)
    ka, kb, pr, R, t, good = _pixel_pairs(rng, 400, cam)
    kw = dict(seed=3, block_size=16, init_blocks=1, max_candidates=64, halve=True)
    r0 = oracle.arrsac_pairs(ka, kb, pr, cam, cam, 1e-7, 256, scene=0, shuffle=True, **kw)
    assert r0["bearings_a"].tobytes() == oracle.calibrate(ka[pr[:, 0]], *cam[:5]).tobytes()
    assert r0["bearings_b"].tobytes() == oracle.calibrate(kb[pr[:, 1]], *cam[:5]).tobytes()
    assert sorted(r0["order"].tolist()) == list(range(400)) and not np.array_equal(r0["order"], np.arange(400))
    assert np.array_equal(r0["order"], oracle.shuffle_order(oracle.scene_seed(3, 0), 400))
    assert r0["best_id"] != 0xFFFFFFFF and len(r0["inliers"]) > 0.8 * good.sum()
    assert set(r0["inliers"].tolist()) <= set(np.flatnonzero(good).tolist()) | set(r0["inliers"].tolist())
    assert np.abs(r0["pose"][:, :3] - R).max() < 1e-3
    tn = r0["pose"][:, 3] / np.linalg.norm(r0["pose"][:, 3])
    assert np.abs(tn - t / np.linalg.norm(t)).max() < 2e-2
    r1 = oracle.arrsac_pairs(ka, kb, pr, cam, cam, 1e-7, 256, scene=1, shuffle=True, **kw)
    assert not np.array_equal(r1["order"], r0["order"])
    assert np.array_equal(r1["order"], oracle.shuffle_order(oracle.scene_seed(3, 1), 400))
    # the scene entry == the single-scene specification on its bearings with the scene's seed (no shuffle)
    rn = oracle.arrsac_pairs(ka, kb, pr, cam, cam, 1e-7, 256, scene=2, shuffle=False, **kw)
    kw2 = dict(kw); kw2["seed"] = oracle.scene_seed(3, 2)
    ref = oracle.arrsac(rn["bearings_a"], rn["bearings_b"], 1e-7, 256, **kw2)
    assert ref[2] == rn["best_id"] and ref[0].tobytes() == rn["pose"].tobytes() and np.array_equal(ref[1], rn["inliers"])
    few = oracle.arrsac_pairs(ka, kb, pr[:7], cam, cam, 1e-7, 256, scene=0, **kw)
    assert few["best_id"] == 0xFFFFFFFF and len(few["inliers"]) == 0
    # K1 arm: calibrate with distortion on one side
    camk = cam[:5] + (-0.05,)
    rk = oracle.arrsac_pairs(ka, kb, pr, cam, camk, 1e-7, 64, scene=0, **kw)
    assert rk["bearings_b"].tobytes() == oracle.calibrate(kb[pr[:, 1]], *cam[:5], k1=-0.05).tobytes()


def _registration_scene(rng, n_kp, n_world, n_pairs, outlier_frac, cam, noise_px=0.2):
    """One new frame against a landmark table: keypoints (pixel coordinates, f32), homogeneous world points and a
    (feature, world point) pair list, `outlier_frac` of it joined at random.  Returns kps, world, pairs, R, t, good."""
    from test_oracle_ransac import _rot
    from oracle.oracle import KP_DTYPE
    R = _rot((rng.random(3) - 0.5) * 0.6)
    t = (rng.random(3) - 0.5) * 1.0
    fx, fy, cx, cy, skew = cam[:5]
    pts = np.stack([rng.uniform(-2, 2, n_world), rng.uniform(-1.2, 1.2, n_world), rng.uniform(4, 9, n_world)], 1)
    world = np.concatenate([pts, np.ones((n_world, 1))], 1) * rng.uniform(0.5, 2.0, (n_world, 1))   # any projective scale
    camp = pts @ R.T + t
    x, y = camp[:, 0] / camp[:, 2], camp[:, 1] / camp[:, 2]
    kps = np.zeros(n_kp, KP_DTYPE)
    which = rng.integers(0, n_world, n_kp)                       # keypoint i observes world point which[i]
    kps["x"] = fx * x[which] + skew * y[which] + cx + rng.standard_normal(n_kp) * noise_px
    kps["y"] = fy * y[which] + cy + rng.standard_normal(n_kp) * noise_px
    feat = rng.choice(n_kp, size=min(n_pairs, n_kp), replace=False)
    wi = which[feat].copy()
    bad = rng.random(len(feat)) < outlier_frac
    wi[bad] = rng.integers(0, n_world, bad.sum())
    good = wi == which[feat]
    o = np.argsort(feat, kind="stable")
    return kps, world, np.stack([feat[o], wi[o]], 1).astype(np.uint32), R, t, good[o]


def test_registration_scene_entry(oracle):
    """orc_p3p_arrsac_pairs, the specification of one scene of rs_p3p_arrsac_batch_device: bearings = orc_calibrate of the
    paired keypoints, points = the paired rows of the world table, the scene entry equals the single-scene P3P
    specification on them with the scene's seed, fewer than three matches give no model, and the pose of the scene comes
    back (WorldToCamera; 1e-3: f32 pixel coordinates + 0.2 px noise)."""
    rng = np.random.default_rng(80)
    cam = (984.2439, 980.8141, 690.0, 233.1966, 0.0, None)
    kps, world, pr, R, t, good = _registration_scene(rng, 600, 900, 500, 0.3, cam)
    kw = dict(seed=5, block_size=32, init_blocks=1, max_candidates=64, halve=True)
    r0 = oracle.p3p_arrsac_pairs(kps, pr, world, cam, 1e-6, 512, scene=0, shuffle=True, **kw)
    assert r0["bearings"].tobytes() == oracle.calibrate(kps[pr[:, 0]], *cam[:5]).tobytes()
    assert r0["world"].tobytes() == np.ascontiguousarray(world[pr[:, 1]]).tobytes()
    assert np.array_equal(r0["order"], oracle.shuffle_order(oracle.scene_seed(5, 0), len(pr)))
    assert r0["best_id"] != 0xFFFFFFFF and len(r0["inliers"]) > 0.7 * good.sum()
    assert good[r0["inliers"]].mean() > 0.98
    assert np.abs(r0["pose"][:, :3] - R).max() < 1e-3 and np.abs(r0["pose"][:, 3] - t).max() < 1e-2
    rn = oracle.p3p_arrsac_pairs(kps, pr, world, cam, 1e-6, 512, scene=3, shuffle=False, **kw)
    kw2 = dict(kw); kw2["seed"] = oracle.scene_seed(5, 3)
    ref = oracle.arrsac(rn["bearings"], rn["world"], 1e-6, 512, p3p=True, **kw2)
    assert ref[2] == rn["best_id"] and ref[0].tobytes() == rn["pose"].tobytes() and np.array_equal(ref[1], rn["inliers"])
    few = oracle.p3p_arrsac_pairs(kps, pr[:2], world, cam, 1e-6, 64, scene=0, **kw)
    assert few["best_id"] == 0xFFFFFFFF and len(few["inliers"]) == 0
    camk = cam[:5] + (-0.05,)
    rk = oracle.p3p_arrsac_pairs(kps, pr, world, camk, 1e-6, 64, scene=0, **kw)
    assert rk["bearings"].tobytes() == oracle.calibrate(kps[pr[:, 0]], *cam[:5], k1=-0.05).tobytes()


def test_landmark_matches_in_the_reference_order(oracle):
    """orc_landmark_matches_ordered against a literal Python restatement of cv-sfm/src/lib.rs:1549-1604: landmark_counts over
    every landmark of every original match, retain, `sort_by_key(Reverse(sum of observations))` (Python's sort is stable, as
    Rust's sort_by_key), filter_map of the robust world points."""
    from collections import Counter
    rng = np.random.default_rng(0x0B5)
    for trial in range(6):
        nq, n_world = int(rng.integers(50, 700)), 900
        best = np.zeros((nq, 3, 2), np.uint32)
        for j in range(nq):
            best[j, :, 0] = rng.choice(n_world + 20, 3, replace=False)
        best[rng.random(nq) < 0.05, 0, 0] = 0xFFFFFFFF
        dec = rng.integers(0, 3, nq).astype(np.uint32)
        merge_ok = (rng.random(nq) < 0.5).astype(np.uint8)
        world = rng.standard_normal((n_world + nq, 4))
        world[:, 3] = np.abs(world[:, 3])
        world[rng.random(len(world)) < 0.25, 3] = -1.0
        obs = rng.integers(1, 4 if trial % 2 else 1 << 18, n_world).astype(np.uint32)
        original = []                                            # (landmarks, feature) in feature order
        for j in range(nq):
            l0, l1 = int(best[j, 0, 0]), int(best[j, 1, 0])
            if l0 == 0xFFFFFFFF:
                continue
            if dec[j] == 1:
                original.append(([l0], j))
            elif dec[j] == 2 and merge_ok[j] and l1 != 0xFFFFFFFF:
                original.append(([l0, l1], j))
        counts = Counter(l for lms, _ in original for l in lms)
        original = [(lms, j) for lms, j in original if all(counts[l] == 1 for l in lms)]
        original.sort(key=lambda t: -sum(int(obs[l]) if l < n_world else 0 for l in t[0]))
        want = []
        for lms, j in original:
            row = lms[0] if len(lms) == 1 else n_world + j
            if len(lms) == 1 and row >= n_world:
                continue
            if world[row, 3] >= 0.0:
                want.append((j, row))
        got = oracle.landmark_pairs(best, dec, world, merge_ok=merge_ok, n_world=n_world, merged_base=n_world, obs_counts=obs)
        assert [tuple(map(int, r)) for r in got] == want, trial
        plain = oracle.landmark_pairs(best, dec, world, merge_ok=merge_ok, n_world=n_world, merged_base=n_world)
        assert sorted(map(tuple, plain.tolist())) == sorted(want)
