"""include/akz_portable_math.h (shared by the oracle and the HIP kernels) against the host libm."""
import math

import numpy as np


def _ulp_diff(a, b):
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


def test_portable_atan2_is_correctly_rounded(oracle):
    rng = np.random.default_rng(7)
    n = 400000
    y = (rng.standard_normal(n) * 10.0 ** rng.uniform(-6, 2, n)).astype(np.float32)
    x = (rng.standard_normal(n) * 10.0 ** rng.uniform(-6, 2, n)).astype(np.float32)
    got = oracle.pm_atan2f(y, x)
    want = np.arctan2(y.astype(np.float64), x.astype(np.float64)).astype(np.float32)
    d = _ulp_diff(got, want)
    assert d.max() <= 1
    assert (d != 0).mean() < 1e-5
    # glibc atan2f itself (what Rust's f32::atan2 calls on Linux) stays within 1 ulp of it
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.atan2f.restype = ctypes.c_float
    libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]
    m = 20000
    ref = np.array([libm.atan2f(float(a), float(b)) for a, b in zip(y[:m], x[:m])], np.float32)
    assert _ulp_diff(got[:m], ref).max() <= 1


def test_portable_atan2_special_cases(oracle):
    cases = [(0.0, 1.0), (-0.0, 1.0), (0.0, -1.0), (-0.0, -1.0), (1.0, 0.0), (-1.0, 0.0), (0.0, 0.0),
             (-0.0, 0.0), (0.0, -0.0), (-0.0, -0.0), (1.0, 1.0), (-1.0, -1.0), (1e-30, -1e30), (3.0, -0.0)]
    y = np.array([c[0] for c in cases], np.float32)
    x = np.array([c[1] for c in cases], np.float32)
    got = oracle.pm_atan2f(y, x)
    for (yy, xx), g in zip(cases, got):
        w = np.float32(math.atan2(yy, xx))
        assert g == w and math.copysign(1, g) == math.copysign(1, w), (yy, xx, g, w)


def test_portable_sincos(oracle):
    rng = np.random.default_rng(8)
    a = rng.uniform(0, 2 * math.pi, 400000).astype(np.float32)
    a[:8] = np.array([0, math.pi, math.pi / 2, 3 * math.pi / 2, 2 * math.pi, 1e-8, 6.2831855, 3.1415927], np.float32)
    s, c = oracle.pm_sincosf(a)
    ws = np.sin(a.astype(np.float64)).astype(np.float32)
    wc = np.cos(a.astype(np.float64)).astype(np.float32)
    assert _ulp_diff(s, ws).max() <= 1 and _ulp_diff(c, wc).max() <= 1
    assert (s != ws).mean() < 1e-5 and (c != wc).mean() < 1e-5
    # and it is what glibc sinf/cosf (what Rust's f32::sin/cos call on Linux) give in ~99 % of cases, never off by more than 1 ulp
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    for fn in (libm.sinf, libm.cosf):
        fn.restype = ctypes.c_float
        fn.argtypes = [ctypes.c_float]
    m = 20000
    gs = np.array([libm.sinf(float(v)) for v in a[:m]], np.float32)
    gc = np.array([libm.cosf(float(v)) for v in a[:m]], np.float32)
    assert _ulp_diff(s[:m], gs).max() <= 1 and _ulp_diff(c[:m], gc).max() <= 1
    # glibc 2.35 sinf/cosf are ~0.56-ulp functions, not correctly rounded: ~1.2 % of inputs differ by 1 ulp
    assert (s[:m] != gs).mean() < 0.03 and (c[:m] != gc).mean() < 0.03


def test_bicubic_colour_sampling_matches_float32_restatement(oracle):
    """cv-sfm/src/bicubic.rs: the C restatement agrees with an independent numpy float32 evaluation of the
    same expressions (rows blended and truncated to u8 first, then the column), returns the default colour
    when the 4x4 neighbourhood leaves the image, and reproduces pixels exactly at integer positions."""
    O = oracle
    rng = np.random.default_rng(9)
    h, w = 37, 53
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    kps = np.zeros(300, O.KP_DTYPE)
    kps["x"] = rng.uniform(-3, w + 3, 300).astype(np.float32)
    kps["y"] = rng.uniform(-3, h + 3, 300).astype(np.float32)
    kps["x"][:10] = np.arange(5, 15, dtype=np.float32)      # integer positions: weight 0 -> the pixel itself
    kps["y"][:10] = np.arange(7, 17, dtype=np.float32)
    got = O.sample_colors_rgb8(rgb, kps)

    f = np.float32

    def clamp(v):
        return np.uint8(255) if not (v < f(255)) else (np.uint8(int(v)) if v > f(0) else np.uint8(0))

    def blend(p0, p1, p2, p3, x):
        p0, p1, p2, p3, x = f(p0), f(p1), f(p2), f(p3), f(x)
        in3 = f(f(f(3) * f(p1 - p2)) + p3) - p0
        in2 = f(f(f(f(f(2) * p0) - f(f(5) * p1)) + f(f(4) * p2)) - p3) + f(x * in3)
        in1 = f(p2 - p0) + f(x * in2)
        return clamp(f(p1 + f(f(f(0.5) * x) * in1)))

    for i in range(len(kps)):
        x, y = f(kps["x"][i]), f(kps["y"][i])
        left, top = f(np.floor(x) - f(1)), f(np.floor(y) - f(1))
        if left < 0 or left + f(4) >= w or top < 0 or top + f(4) >= h:
            assert tuple(got[i]) == (0, 0, 0)
            continue
        xw, yw = f(x - f(left + f(1))), f(y - f(top + f(1)))
        l, t = int(left), int(top)
        want = []
        for ch in range(3):
            col = [blend(*[rgb[t + r, l + j, ch] for j in range(4)], xw) for r in range(4)]
            want.append(blend(*col, yw))
        assert tuple(got[i]) == tuple(want), (i, got[i], want)
    for i in range(10):
        assert tuple(got[i]) == tuple(rgb[int(kps["y"][i]), int(kps["x"][i])])


def test_round_half_away_by_one_add_and_floor():
    """k_orient_describe rounds its sample coordinates as floor(v + 0.49999997f) (one add, one convert) instead of roundf's
    compare-and-select sequence (cv_amd/csrc/akz_keypoints.hip: round_flr_i32 / round_sat_u32).  The identity
    floor(v (+) c) == roundf(v), c = 0x3EFFFFFF, holds for every f32 v >= 0: walked here over every f32 of the binades where
    the fraction matters most ([0.25, 16) and [2^22, 2^25)) and a stride of the ones between; below 0.25 the sum stays under
    1, above 2^24 every f32 is an integer and the sum rounds back to it.  For v < 0 the kernel only needs the sign: the sum
    is negative exactly when v <= -0.5."""
    c = np.float32(0.49999997)
    assert c.view(np.uint32) == 0x3EFFFFFF
    for e in range(-4, 26):
        dense = -2 <= e <= 3 or e >= 22
        m = np.arange(0, 1 << 23, 1 if dense else 61, dtype=np.uint32)
        v = (np.uint32((e + 127) << 23) | m).view(np.float32)
        got = np.floor(v + c).astype(np.float64)                       # f32 add (round to nearest even), floor
        want = np.floor(v.astype(np.float64) + 0.5)                    # roundf for v >= 0: the f64 sum is exact
        assert (got == want).all(), e
    neg = -np.concatenate([np.linspace(0, 1, 100001), [0.5, 0.49999997, 0.50000006, 0.25]]).astype(np.float32)
    assert (((neg + c) < 0) == (neg <= np.float32(-0.5))).all()


def test_top2_insertion_of_a_pair_in_three_operations():
    """The matcher keeps the two smallest keys per query (cv_amd/csrc/hm_match.hip: topk_insert2 / topk_merge2).  Two new
    keys a, b go into the ascending pair (l0, l1) as  l0' = min3(l0, a, b),  l1' = min(l1, med3(l0, a, b))  — three
    instructions instead of two insertions of two.  Exhaustive over small values with every tie pattern."""
    v = np.arange(6)
    l0, l1, a, b = np.meshgrid(v, v, v, v, indexing="ij")
    keep = l0 <= l1
    l0, l1, a, b = l0[keep], l1[keep], a[keep], b[keep]
    want = np.sort(np.stack([l0, l1, a, b], 1), axis=1)[:, :2]
    med3 = np.sort(np.stack([l0, a, b], 1), axis=1)[:, 1]
    got0 = np.minimum(np.minimum(l0, a), b)
    got1 = np.minimum(l1, med3)
    assert (got0 == want[:, 0]).all() and (got1 == want[:, 1]).all()


def test_top3_insertion_in_three_operations():
    """k = 3 (cv-sfm's registration path): a new key v goes into the ascending triple (l0, l1, l2) as
    l0' = min(l0, v), l1' = med3(l0, l1, v), l2' = med3(l1, l2, v) (cv_amd/csrc/hm_match.hip: topk_insert<3>).
    Exhaustive over small values with every tie pattern."""
    v = np.arange(6)
    l0, l1, l2, x = np.meshgrid(v, v, v, v, indexing="ij")
    keep = (l0 <= l1) & (l1 <= l2)
    l0, l1, l2, x = l0[keep], l1[keep], l2[keep], x[keep]
    want = np.sort(np.stack([l0, l1, l2, x], 1), axis=1)[:, :3]
    med = lambda a, b, c: np.sort(np.stack([a, b, c], 1), axis=1)[:, 1]
    got = np.stack([np.minimum(l0, x), med(l0, l1, x), med(l1, l2, x)], 1)
    assert (got == want).all()


def test_orientation_angle_estimate_error_bound(tmp_path):
    """The window membership of k_orient_describe comes from an f32 ESTIMATE of the angle wherever the estimate is further
    than kOriEps from every end point (cv_amd/csrc/akz_keypoints.hip: ori_sample_entry).  That is safe iff
    |estimate - exact expression| < kOriEps for EVERY input the estimate path accepts.  tools/ubench/atan_bound.c proves it:
    the polynomial against atan over every f32 argument of [2^-13, 1] with correctly rounded f32 operations (the rest bounded
    analytically), v_rcp_f32's specified 1 ulp, the three reflections' constants and roundings, and the f32 roundings of the
    exact expression it is compared with.  The coefficients and the band are read from the kernel source, so the proof is
    of the code that ships; the total must stay below a quarter of the band."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "cv_amd", "csrc", "akz_keypoints.hip")).read()
    body = src[src.index("__device__ __forceinline__ int ori_sample_entry("):]
    body = body[:body.index("float a = t * p;")]
    first = re.search(r"float p = (-?[0-9.e-]+)f;", body).group(1)
    rest = re.findall(r"p = __builtin_fmaf\(p, s, (-?[0-9.e-]+)f\);", body)
    assert len(rest) == 7
    eps = re.search(r"constexpr float kOriEps = ([0-9.e-]+)f;", src).group(1)
    exe = str(tmp_path / "atan_bound")
    subprocess.check_call(["gcc", "-O2", "-mfma", "-fopenmp", "-ffp-contract=off", os.path.join(root, "tools", "ubench", "atan_bound.c"),
                           "-o", exe, "-lm"])
    r = subprocess.run([exe, first] + rest + [eps], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"TOTAL ([0-9.e+-]+)\s+kOriEps ([0-9.e+-]+)\s+ratio ([0-9.]+)", r.stdout)
    assert m and float(m.group(1)) < 2.0e-6 and float(m.group(3)) >= 4.0, r.stdout
