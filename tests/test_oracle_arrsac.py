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
