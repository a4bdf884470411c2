"""Pins the CPU oracle against every known-answer test the reference holds for the hot path
(SURVEY.md §8c).  CPU only.  All file:line citations are relative to the rust-cv/cv checkout."""
import os

import numpy as np
import pytest


def _extract_pair(O, kitti, thr):
    a, b = kitti
    ak = O.Akaze(a.shape[1], a.shape[0], O.default_config(threshold=thr))
    ka, da = ak.extract(a)
    kb, db = ak.extract(b)
    return ka, da, kb, db


@pytest.mark.parametrize("trig", [0, 1])
def test_estimate_pose_counts(oracle, kitti, trig):
    """akaze/tests/estimate_pose.rs:41,42,59 — Akaze::sparse() gives exactly 399 and 343 descriptors
    on the two KITTI frames, and BF 2-NN + Lowe 0.5 (a->b only) gives exactly 11 matches.  Must hold
    with the portable trig (what the HIP path implements) and with the host libm (what a Rust build
    on this machine would call)."""
    O = oracle
    O.set_option(O.OPT_TRIG, trig)
    try:
        ka, da, kb, db = _extract_pair(O, kitti, 0.01)
        assert len(da) == 399 and len(ka) == 399
        assert len(db) == 343 and len(kb) == 343
        m = O.match(da, db, rule=O.RULE_LOWE, param_f=0.5, symmetric=False)
        assert len(m) == 11
    finally:
        O.set_option(O.OPT_TRIG, 0)


def test_pins_hold_for_every_third_party_order(oracle, kitti):
    """The three [3P-unverified] arithmetic orders (wide::f32x4::reduce_add, mul_add fusion, ndarray
    2x2 sum) are switches in the oracle.  The reference's pins do not discriminate between them
    (all 8 combinations give 399/343/11), so the frozen choice is the one the crate sources imply for
    the default x86_64 target (no fma, no sse3): unfused mul+add, ((a0+a1)+a2)+a3, (a+b)+(c+d)."""
    O = oracle
    try:
        for red in (0, 1):
            for fma in (0, 1):
                for hs in (0, 1):
                    O.set_option(O.OPT_REDUCE, red)
                    O.set_option(O.OPT_FMA, fma)
                    O.set_option(O.OPT_HALFSUM, hs)
                    ka, da, kb, db = _extract_pair(O, kitti, 0.01)
                    m = O.match(da, db, rule=O.RULE_LOWE, param_f=0.5, symmetric=False)
                    assert (len(da), len(db), len(m)) == (399, 343, 11), (red, fma, hs)
    finally:
        for o in (O.OPT_REDUCE, O.OPT_FMA, O.OPT_HALFSUM):
            O.set_option(o, 0)


def test_gaussian_kernel_known_answer(oracle):
    """akaze/src/image.rs:396-412."""
    k = oracle.gaussian_kernel(3.0, 7)
    known = [0.10628852, 0.14032133, 0.16577007, 0.17524014, 0.16577007, 0.14032133, 0.10628852]
    assert np.all(np.abs(k - np.array(known, np.float32)) < 1e-4)
    assert k.dtype == np.float32 and abs(float(k.sum()) - 1.0) < 1e-6


def _clamp_correlate(img, kernel, axis):
    """What imageproc::filter::{horizontal,vertical}_filter computes: clamp-border correlation
    (akaze/src/image.rs:414-432 compares against it with 1e-4)."""
    r = len(kernel) // 2
    pad = [(0, 0), (0, 0)]
    pad[axis] = (r, r)
    p = np.pad(img.astype(np.float64), pad, mode="edge")
    out = np.zeros(img.shape, np.float64)
    for i, kv in enumerate(kernel):
        sl = [slice(None), slice(None)]
        sl[axis] = slice(i, i + img.shape[axis])
        out += float(kv) * p[tuple(sl)]
    return out


def test_filters_match_clamp_border_correlation(oracle, kitti):
    """akaze/src/image.rs:414-432 — on the KITTI frame, with gaussian_kernel(3.0, 7)."""
    O = oracle
    img = O.u8_to_f32(kitti[0])
    k = O.gaussian_kernel(3.0, 7)
    assert np.max(np.abs(O.horizontal_filter(img, k) - _clamp_correlate(img, k, 1))) < 1e-4
    assert np.max(np.abs(O.vertical_filter(img, k) - _clamp_correlate(img, k, 0))) < 1e-4
    # asymmetric kernel: orientation must be correlation (no flip)
    ka = np.array([1.0, 2.0, -0.5, 0.25, 3.0], np.float32)
    assert np.max(np.abs(O.horizontal_filter(img, ka) - _clamp_correlate(img, ka, 1))) < 1e-4
    assert np.max(np.abs(O.vertical_filter(img, ka) - _clamp_correlate(img, ka, 0))) < 1e-4


def test_u8_conversion_is_true_division(oracle):
    """image.rs:54 — f32::from(v) / 255f32, not a reciprocal multiply."""
    v = np.arange(256, dtype=np.uint8).reshape(16, 16)
    got = oracle.u8_to_f32(v)
    want = (v.astype(np.float32) / np.float32(255.0)).astype(np.float32)
    assert np.array_equal(got, want)
    assert not np.array_equal(got, v.astype(np.float32) * np.float32(1.0 / 255.0))


def test_pyramid_schedule(oracle):
    """SURVEY.md §8 table (derived from evolution.rs:46-126, fed_tau.rs:44-46,
    detector_response.rs:11-13): 1080p -> 16 levels, 166 FED steps; KITTI -> 13 levels, 93 steps."""
    O = oracle
    ak = O.Akaze(1920, 1080)
    assert ak.num_levels == 16
    steps = [ak.level(i).n_fed_steps for i in range(16)]
    assert steps == [0, 3, 3, 4, 4, 5, 6, 7, 8, 10, 12, 14, 17, 20, 24, 29] and sum(steps) == 166
    assert [ak.level(i).deriv_sigma for i in range(16)] == [2, 3, 3, 4] * 4
    assert [(ak.level(i).width, ak.level(i).height) for i in (0, 4, 8, 12)] == \
        [(1920, 1080), (960, 540), (480, 270), (240, 135)]
    assert abs(ak.level(15).esigma - 21.5269) < 1e-3
    for i in range(1, 16):
        tau = ak.fed_tau(i)
        ttime = ak.level(i).etime - ak.level(i - 1).etime
        assert abs(tau.sum() - ttime) < 1e-9 * max(1.0, ttime)  # a FED cycle sums to the stopping time
    k = O.Akaze(1392, 512)
    assert k.num_levels == 13 and sum(k.level(i).n_fed_steps for i in range(13)) == 93
    # octave dropped below 40 px, single sublevel below 80 px (evolution.rs:89-98)
    assert O.Akaze(100, 100).num_levels == 4 + 1  # 100 -> 4 sublevels, 50 -> 1 sublevel, 25 dropped
    assert O.Akaze(39, 200).num_levels == 0


def test_half_size_odd_edges(oracle):
    """image.rs:154-199: odd height -> the LAST output row is overwritten from the LAST input row
    (1x2 sums * 0.5), likewise columns; corner copied."""
    O = oracle
    rng = np.random.default_rng(1)
    img = rng.random((7, 9), dtype=np.float32)
    out = O.half_size(img)
    assert out.shape == (3, 4)
    want = np.zeros((3, 4), np.float32)
    for y in range(3):
        for x in range(4):
            w = img[2 * y:2 * y + 2, 2 * x:2 * x + 2]
            want[y, x] = ((w[0, 0] + w[0, 1]) + (w[1, 0] + w[1, 1])) * np.float32(0.25)
    for x in range(4):
        want[2, x] = (img[6, 2 * x] + img[6, 2 * x + 1]) * np.float32(0.5)
    for y in range(3):
        want[y, 3] = (img[2 * y, 8] + img[2 * y + 1, 8]) * np.float32(0.5)
    want[2, 3] = img[6, 8]
    assert np.array_equal(out, want)


def test_descriptor_layout(oracle, kitti_golden):
    """descriptors.rs:181-202: 486 bits, LSB-first; bits 486..511 stay zero."""
    d = kitti_golden["default_desc0"]
    assert d.shape[1] == 64
    assert np.all(d[:, 61:] == 0) and np.all((d[:, 60] & 0xC0) == 0)  # 486 = 60*8 + 6
    assert d[:, :60].any()


def test_golden_regression(oracle, kitti, kitti_golden):
    """Oracle output is stable against the committed oracle-generated golden vectors."""
    O = oracle
    ka, da, kb, db = _extract_pair(O, kitti, 0.001)
    g = kitti_golden
    assert np.array_equal(da, g["default_desc0"]) and np.array_equal(db, g["default_desc14"])
    assert ka.tobytes() == g["default_kp0"].tobytes()
    assert np.array_equal(O.match(da, db, rule=O.RULE_STRICT, param_u=24, symmetric=True), g["default_sym24"])
    # keypoints are ordered by response descending (lib.rs:326)
    assert np.all(np.diff(ka["response"]) <= 0)


def test_knn2_tie_rule_and_matching_rules(oracle):
    """space::LinearKnn::knn semantics (lowest index wins ties) and the three acceptance rules."""
    O = oracle
    t = np.zeros((5, 64), np.uint8)
    t[0, 0] = 0b111      # distance 3 from zero
    t[1, 0] = 0b1        # 1
    t[2, 1] = 0b1        # 1 (tie with index 1)
    t[3, 2] = 0b11       # 2
    t[4, 3] = 0b1        # 1 (tie)
    q = np.zeros((1, 64), np.uint8)
    nn = O.knn2(q, t)
    assert (nn[0, 0]["index"], nn[0, 0]["distance"]) == (1, 1)
    assert (nn[0, 1]["index"], nn[0, 1]["distance"]) == (2, 1)
    with pytest.raises(ValueError):
        O.knn2(q, t[:1])
    # rules: d0=0, d1 = 24 exactly
    t2 = np.zeros((2, 64), np.uint8)
    t2[1, :3] = 0xFF
    q2 = np.zeros((2, 64), np.uint8)
    q2[1, 32:] = 0xFF   # second query: d = 256 / 280 -> accepted by no rule except Lowe-false as well
    assert len(O.match(q2, t2, rule=O.RULE_STRICT, param_u=24, symmetric=False)) == 0   # 0+24 < 24 false
    assert O.match(q2, t2, rule=O.RULE_BETTER_BY, param_u=24, symmetric=False).tolist() == [[0, 0], [1, 0]]
    assert O.match(q2, t2, rule=O.RULE_LOWE, param_f=0.5, symmetric=False).tolist() == [[0, 0]]  # 0 < 12
    # cv-sfm guard: fewer than 2 on either side -> no matches (cv-sfm/src/lib.rs:3099-3101)
    assert len(O.match(q, t2[:1], rule=O.RULE_BETTER_BY, symmetric=True)) == 0


def test_knn_general_k_matches_linear_knn_semantics(oracle):
    """orc_knn(k) is LinearKnn::knn restated for any k: it agrees with orc_knn2 for k = 2, with a direct
    (distance, index) sort for k = 1..4, and returns min(k, nt) real neighbours."""
    O = oracle
    rng = np.random.default_rng(5)
    q = rng.integers(0, 256, (40, 64), dtype=np.uint8)
    t = rng.integers(0, 256, (90, 64), dtype=np.uint8)
    t[10:30] = t[40:60]                      # duplicates: ties
    q[:5] = t[:5]
    n2 = O.knn2(q, t)
    g2 = O.knn(q, t, 2)
    assert np.array_equal(n2["index"], g2["index"]) and np.array_equal(n2["distance"], g2["distance"])
    bits = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(axis=2).astype(np.int64)   # [nq, nt]
    for k in (1, 2, 3, 4):
        g = O.knn(q, t, k)
        order = np.lexsort((np.broadcast_to(np.arange(len(t)), bits.shape), bits), axis=1)[:, :k]
        assert np.array_equal(g["index"], order.astype(np.uint32))
        assert np.array_equal(g["distance"], np.take_along_axis(bits, order, 1).astype(np.uint32))
    short = O.knn(q, t[:2], 3)
    assert (short["index"][:, 2] == 0xFFFFFFFF).all() and (short["index"][:, :2] < 2).all()


def test_cpu_baseline_build_is_the_checker_bit_for_bit(oracle):
    """bench.py times a second build of the oracle's sources (-O3 -march=native -fopenmp) as the CPU baseline;
    it must be the same function: keypoints, descriptors and match pairs bit-identical to the -O2 checker, single- and
    multi-threaded (OpenMP over frames only re-orders whole frames)."""
    from conftest import synth_frame
    frames = np.stack([synth_frame(480, 272, 40 + s) for s in range(5)])
    want = oracle.extract_match_many(frames, threads=1, fast=False)
    for threads in (1, 3):
        got = oracle.extract_match_many(frames, threads=threads, fast=True)
        for i, (a, b) in enumerate(zip(want, got)):
            assert a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1]), (threads, i)
            assert (a[2] is None and b[2] is None) or np.array_equal(a[2], b[2]), (threads, i)
    single = oracle.Akaze(480, 272, oracle.default_config())
    kp, d = single.extract(frames[2])
    assert kp.tobytes() == want[2][0].tobytes() and np.array_equal(d, want[2][1])
    assert len(kp) > 100


def test_pin_arith_names_the_combination_a_dump_was_made_with(kitti, tmp_path):
    """tools/pin_arith.py — the one-command route from "parity vs oracle/" to "parity vs rust-cv" (INTEGRATION.md): given the
    two files `cargo run --example akaze` writes (akaze/examples/akaze.rs:11-33) and the image, it names the combination
    of the three un-vendored arithmetic orders that reproduces them byte for byte.  Fed the oracle's own dump at arith 5
    (tools/akaze_dump.py --oracle --arith 5) it must answer 5 and nothing else; a dump whose angles carry another libm's
    last bits is still pinned in its arithmetic."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    img = kitti[0][:256, :640]                      # four octaves, odd sizes on the way down: every switch matters
    np.save(tmp_path / "crop.npy", img)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "akaze_dump.py"), "--oracle", "--arith", "5", "--out-dir", str(tmp_path),
                        str(tmp_path / "crop.npy")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "pin_arith.py"), str(tmp_path / "crop_kps.csv"),
                        str(tmp_path / "crop_descs.txt"), str(tmp_path / "crop.npy")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 17 and sum(ln.startswith("MATCH") for ln in lines) == 1
    assert [ln for ln in lines if ln.startswith("MATCH")][0].startswith("MATCH     arith 5 ") and "trig portable" in lines[5]
    assert lines[-1].startswith("VERDICT: the Rust build computes arith = 5 ")
    # `--arith all` writes the eight variants; variant 5 is the file above, the others differ from it
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "akaze_dump.py"), "--oracle", "--arith", "all", "--out-dir", str(tmp_path / "all"),
                        str(tmp_path / "crop.npy")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    texts = [open(tmp_path / "all" / f"crop_a{k}_kps.csv").read() + open(tmp_path / "all" / f"crop_a{k}_descs.txt").read() for k in range(8)]
    assert texts[5] == open(tmp_path / "crop_kps.csv").read() + open(tmp_path / "crop_descs.txt").read()
    assert len(set(texts)) == 8
    # another libm: nudge one angle by an ulp -> no byte match, the arithmetic is still pinned to 5
    sys.path.insert(0, os.path.join(root, "tools"))
    import akaze_dump as D
    import pin_arith as P
    kps, descs = D.oracle_extract(img, 5, "portable")
    kps = kps.copy()
    kps["angle"][3] = np.nextafter(kps["angle"][3], np.float32(10.0))
    out = []
    matches, _ = P.pin(D.kps_text(kps), D.descs_text(descs), img, out=out.append)
    assert matches == [] and "arith = 5 reproduces every keypoint" in out[-1]
