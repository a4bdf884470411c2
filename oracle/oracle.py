"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg — never by anything under cv_amd/.  See akaze_oracle.c / match_oracle.c for what each function
restates and which reference file:line it follows.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


class Config(C.Structure):
    """akz_config == akaze::Akaze (akaze/src/lib.rs:109-142)."""
    _fields_ = [
        ("maximum_features", C.c_uint64),
        ("num_sublevels", C.c_uint32),
        ("max_octave_evolution", C.c_uint32),
        ("base_scale_offset", C.c_double),
        ("initial_contrast", C.c_double),
        ("contrast_percentile", C.c_double),
        ("contrast_factor_num_bins", C.c_uint64),
        ("derivative_factor", C.c_double),
        ("detector_threshold", C.c_double),
        ("descriptor_channels", C.c_uint64),
        ("descriptor_pattern_size", C.c_uint64),
    ]


class LevelInfo(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("octave", C.c_uint32), ("sublevel", C.c_uint32),
        ("esigma", C.c_double), ("etime", C.c_double),
        ("n_fed_steps", C.c_uint32), ("deriv_sigma", C.c_uint32),
    ]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("response", "<f4"), ("size", "<f4"),
                     ("angle", "<f4"), ("octave", "<u4"), ("class_id", "<u4")])
assert KP_DTYPE.itemsize == 28
NB_DTYPE = np.dtype([("index", "<u4"), ("distance", "<u4")])

OPT_REDUCE, OPT_FMA, OPT_HALFSUM, OPT_TRIG = 0, 1, 2, 3
BUF = {"Lt": 0, "Lsmooth": 1, "Lx": 2, "Ly": 3, "Ldet": 4, "Lflow": 5, "Lxx": 6, "Lyy": 7, "Lxy": 8}


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("akaze_oracle.c", "match_oracle.c", "ransac_oracle.c", "p3p_oracle.c", "color_oracle.c", "lsh_oracle.c", "arrsac_oracle.c",
                                        "batch_oracle.c", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(Config), C.c_int, C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_num_levels.argtypes = [C.c_void_p]
        L.orc_level.argtypes = [C.c_void_p, C.c_int, C.POINTER(LevelInfo)]
        L.orc_fed_tau.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_int]
        L.orc_extract_f32.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_extract_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_extract_u16.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_scale_space_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_level_buffer.restype = fp
        L.orc_level_buffer.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_contrast.restype = C.c_double
        L.orc_contrast.argtypes = [C.c_void_p]
        L.orc_num_candidates.restype = C.c_uint32
        L.orc_num_candidates.argtypes = [C.c_void_p]
        L.orc_keypoints.restype = C.c_uint32
        L.orc_keypoints.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.orc_descriptors.restype = C.c_void_p
        L.orc_descriptors.argtypes = [C.c_void_p]
        L.orc_config_default.argtypes = [C.POINTER(Config)]
        L.orc_gaussian_kernel.argtypes = [C.c_float, C.c_int, C.c_void_p]
        L.orc_horizontal_filter.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_vertical_filter.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_gaussian_blur.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]
        L.orc_half_size.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_scharr.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_void_p]
        L.orc_scharr_kernel.argtypes = [C.c_uint32, C.c_int, C.c_void_p]
        L.orc_contrast_factor.restype = C.c_double
        L.orc_contrast_factor.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_uint64]
        L.orc_pm_g2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_void_p]
        L.orc_fed_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float]
        L.orc_fed_tau_by_process_time.argtypes = [C.c_double, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_int]
        L.orc_u8_to_f32.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_set_option.argtypes = [C.c_int, C.c_int]
        L.orc_get_option.argtypes = [C.c_int]
        L.orc_knn2.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_sample_colors_rgb8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_knn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_arrsac_draw.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_arrsac.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_arrsac_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                       C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_p3p_arrsac_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p]
        L.orc_pairs_in_range.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_shuffle_order.argtypes = [C.c_uint64, C.c_uint32, C.c_void_p]
        L.orc_scene_seed.restype = C.c_uint64
        L.orc_scene_seed.argtypes = [C.c_uint64, C.c_uint32]
        L.orc_luma.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_best_of_views.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_match.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_uint32,
                                C.c_float, C.c_int, C.c_void_p, C.c_uint32]
        L.orc_calibrate.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_eight_point.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p]
        L.orc_essential_poses.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_void_p]
        L.orc_residual.restype = C.c_double
        L.orc_residual.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int]
        L.orc_essential_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_double,
                                          C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p]
        L.orc_p3p_poses.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_w2c_residual.restype = C.c_double
        L.orc_w2c_residual.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_p3p_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_double, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_hash_bag.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_hash_knn.restype = C.c_uint32
        L.orc_hash_knn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_pm_atan2f_v.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_pm_sincosf_v.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def set_option(which, value):
    lib().orc_set_option(which, value)


def default_config(threshold=None, maximum_features=None):
    cfg = Config()
    lib().orc_config_default(C.byref(cfg))
    if threshold is not None:
        cfg.detector_threshold = threshold
    if maximum_features is not None:
        cfg.maximum_features = maximum_features
    return cfg


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def gaussian_kernel(r, ksize):
    out = np.empty(ksize, np.float32)
    lib().orc_gaussian_kernel(r, ksize, out.ctypes.data)
    return out


def horizontal_filter(img, kernel):
    img = _f32(img); k = _f32(kernel); out = np.empty_like(img)
    lib().orc_horizontal_filter(img.ctypes.data, img.shape[1], img.shape[0], k.ctypes.data, len(k), out.ctypes.data)
    return out


def vertical_filter(img, kernel):
    img = _f32(img); k = _f32(kernel); out = np.empty_like(img)
    lib().orc_vertical_filter(img.ctypes.data, img.shape[1], img.shape[0], k.ctypes.data, len(k), out.ctypes.data)
    return out


def gaussian_blur(img, r):
    img = _f32(img); out = np.empty_like(img)
    lib().orc_gaussian_blur(img.ctypes.data, img.shape[1], img.shape[0], r, out.ctypes.data)
    return out


def half_size(img):
    img = _f32(img)
    out = np.empty((img.shape[0] // 2, img.shape[1] // 2), np.float32)
    lib().orc_half_size(img.ctypes.data, img.shape[1], img.shape[0], out.ctypes.data)
    return out


def scharr(img, sigma, vertical):
    img = _f32(img); out = np.empty_like(img)
    lib().orc_scharr(img.ctypes.data, img.shape[1], img.shape[0], sigma, int(vertical), out.ctypes.data)
    return out


def contrast_factor(img, percentile=0.7, scale=1.0, nbins=300):
    img = _f32(img)
    return lib().orc_contrast_factor(img.ctypes.data, img.shape[1], img.shape[0], percentile, scale, nbins)


def pm_g2(lx, ly, k):
    lx = _f32(lx); ly = _f32(ly); out = np.empty_like(lx)
    lib().orc_pm_g2(lx.ctypes.data, ly.ctypes.data, lx.size, k, out.ctypes.data)
    return out


def fed_step(L, c, tau):
    L = _f32(L).copy(); c = _f32(c)
    lib().orc_fed_step(L.ctypes.data, c.ctypes.data, L.shape[1], L.shape[0], np.float32(tau))
    return L


def u8_to_f32(img):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty(img.shape, np.float32)
    lib().orc_u8_to_f32(img.ctypes.data, img.shape[1], img.shape[0], img.shape[1], out.ctypes.data)
    return out


class Akaze:
    """CPU oracle of akaze::Akaze::extract for one image size."""

    def __init__(self, w, h, cfg=None):
        self.cfg = cfg if cfg is not None else default_config()
        self.w, self.h = w, h
        self._c = lib().orc_create(C.byref(self.cfg), w, h)

    def close(self):
        if self._c:
            lib().orc_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_levels(self):
        return lib().orc_num_levels(self._c)

    def level(self, i):
        info = LevelInfo()
        assert lib().orc_level(self._c, i, C.byref(info)) == 0
        return info

    def fed_tau(self, i):
        buf = (C.c_double * 256)()
        n = lib().orc_fed_tau(self._c, i, buf, 256)
        return np.array(buf[:n], np.float64)

    def extract(self, img):
        """img: HxW uint8 (Luma8), uint16 (Luma16) or float32. Returns (keypoints structured array, descriptors [n,64] u8)."""
        img = np.ascontiguousarray(img)
        assert img.shape == (self.h, self.w), (img.shape, self.h, self.w)
        if img.dtype == np.uint8:
            n = lib().orc_extract_u8(self._c, img.ctypes.data, self.w)
        elif img.dtype == np.uint16:
            n = lib().orc_extract_u16(self._c, img.ctypes.data, self.w)
        else:
            img = _f32(img)
            n = lib().orc_extract_f32(self._c, img.ctypes.data)
        return self.keypoints(3), self.descriptors()

    def scale_space(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        lib().orc_scale_space_u8(self._c, img.ctypes.data, self.w)

    def keypoints(self, stage):
        p = C.c_void_p()
        n = lib().orc_keypoints(self._c, stage, C.byref(p))
        if n == 0:
            return np.empty(0, KP_DTYPE)
        buf = (C.c_char * (n * 28)).from_address(p.value)
        return np.frombuffer(buf, KP_DTYPE, n).copy()

    def descriptors(self):
        p = C.c_void_p()
        n = lib().orc_keypoints(self._c, 3, C.byref(p))
        d = lib().orc_descriptors(self._c)
        if n == 0:
            return np.empty((0, 64), np.uint8)
        buf = (C.c_char * (n * 64)).from_address(d)
        return np.frombuffer(buf, np.uint8, n * 64).reshape(n, 64).copy()

    def buffer(self, level, name):
        w = C.c_int(); h = C.c_int()
        p = lib().orc_level_buffer(self._c, level, BUF[name], C.byref(w), C.byref(h))
        if not p or w.value == 0:
            return None
        return np.ctypeslib.as_array(p, (h.value, w.value)).copy()

    @property
    def contrast(self):
        return lib().orc_contrast(self._c)

    @property
    def num_candidates(self):
        return lib().orc_num_candidates(self._c)


def knn2(q, t):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 64)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 64)
    out = np.empty((len(q), 2), NB_DTYPE)
    r = lib().orc_knn2(q.ctypes.data, len(q), t.ctypes.data, len(t), out.ctypes.data)
    if r != 0:
        raise ValueError("knn2 needs at least two targets")
    return out


def sample_colors_rgb8(rgb, kps):
    """interpolate_bicubic at every keypoint of a KP_DTYPE array on an [h, w, 3] uint8 image -> [n, 3] uint8."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    kps = np.ascontiguousarray(kps)
    out = np.zeros((len(kps), 3), np.uint8)
    lib().orc_sample_colors_rgb8(rgb.ctypes.data, rgb.shape[1], rgb.shape[0], kps.ctypes.data, len(kps), out.ctypes.data)
    return out


def knn(q, t, k):
    """LinearKnn.knn(q, k): [nq, k] (index, distance); slots past min(k, len(t)) are (2^32-1, 2^32-1)."""
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 64)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 64)
    out = np.empty((len(q), k), NB_DTYPE)
    if lib().orc_knn(q.ctypes.data, len(q), t.ctypes.data, len(t), k, out.ctypes.data) != 0:
        raise ValueError("k must be >= 1")
    return out


def hash_bag(features, codewords):
    """HammingHasher::hash_bag (lsh_oracle.c): (hash [len(codewords) / 8] u8, words [n] (index, distance))."""
    f = np.ascontiguousarray(features, np.uint8).reshape(-1, 64)
    cw = np.ascontiguousarray(codewords, np.uint8).reshape(-1, 64)
    h = np.empty(len(cw) // 8, np.uint8)
    words = np.empty(len(f), NB_DTYPE)
    if lib().orc_hash_bag(f.ctypes.data, len(f), cw.ctypes.data, len(cw), h.ctypes.data, words.ctypes.data) != 0:
        raise ValueError("the codeword count must be a positive multiple of 32")
    return h, words


def hash_knn(query, hashes, k):
    """The k stored hashes nearest to `query`, ascending (distance, index)."""
    hs = np.ascontiguousarray(hashes, np.uint8)
    q = np.ascontiguousarray(query, np.uint8).reshape(-1)
    hs = hs.reshape(-1, len(q))
    out = np.empty(max(k, 1), NB_DTYPE)
    m = lib().orc_hash_knn(q.ctypes.data, hs.ctypes.data, len(hs), len(q), k, out.ctypes.data)
    return out[:m].copy()


RULE_STRICT, RULE_BETTER_BY, RULE_LOWE = 0, 1, 2


def match(a, b, rule=RULE_STRICT, param_u=24, param_f=0.5, symmetric=True):
    a = np.ascontiguousarray(a, np.uint8).reshape(-1, 64)
    b = np.ascontiguousarray(b, np.uint8).reshape(-1, 64)
    cap = max(len(a), 1)
    pairs = np.empty((cap, 2), np.uint32)
    n = lib().orc_match(a.ctypes.data, len(a), b.ctypes.data, len(b), rule, param_u, param_f,
                        int(symmetric), pairs.ctypes.data, cap)
    if n < 0:
        raise ValueError("match: fewer than two descriptors on a side")
    return pairs[:n].copy()


def pm_atan2f(y, x):
    y = _f32(y); x = _f32(x); out = np.empty_like(y)
    lib().orc_pm_atan2f_v(y.ctypes.data, x.ctypes.data, y.size, out.ctypes.data)
    return out


def pm_sincosf(a):
    a = _f32(a); s = np.empty_like(a); c = np.empty_like(a)
    lib().orc_pm_sincosf_v(a.ctypes.data, a.size, s.ctypes.data, c.ctypes.data)
    return s, c


# ---- two-view geometric verification (ransac_oracle.c) ------------------------------------------------
RS_EPS, RS_ITERS = 1e-12, 1000   # EightPoint::default (eight-point/src/lib.rs:60-67)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def calibrate(kps, fx, fy, cx, cy, skew=0.0, k1=None):
    """CameraIntrinsics::calibrate / CameraIntrinsicsK1Distortion::calibrate -> [n,3] unit bearings."""
    kps = np.ascontiguousarray(kps, dtype=KP_DTYPE)
    intr = _f64([fx, fy, cx, cy, skew])
    out = np.empty((len(kps), 3), np.float64)
    lib().orc_calibrate(intr.ctypes.data, int(k1 is not None), float(k1 or 0.0), kps.ctypes.data, len(kps),
                        out.ctypes.data)
    return out


def eight_point(a8, b8):
    a8 = _f64(a8); b8 = _f64(b8); E = np.empty((3, 3), np.float64)
    if lib().orc_eight_point(a8.ctypes.data, b8.ctypes.data, RS_EPS, RS_ITERS, E.ctypes.data) != 0:
        return None
    return E


def essential_poses(E):
    E = _f64(E); P = np.empty((4, 3, 4), np.float64)
    if lib().orc_essential_poses(E.ctypes.data, RS_EPS, RS_ITERS, P.ctypes.data) != 0:
        return None
    return P


def pose_residual(pose, a, b):
    pose = _f64(pose); a = _f64(a); b = _f64(b)
    return lib().orc_residual(pose.ctypes.data, a.ctypes.data, b.ctypes.data, 1e-12, 1024)


def essential_batch(ba, bb, sample_idx, thresh):
    """Exhaustive consensus. Returns (pose[3,4], best_id (= hyp*4 + pose), inlier indices, counts[n_hyp,4])."""
    ba = _f64(ba); bb = _f64(bb)
    si = np.ascontiguousarray(sample_idx, np.uint32).reshape(-1, 8)
    n = len(ba)
    pose = np.empty((3, 4), np.float64); best = C.c_uint32(); inl = np.empty(max(n, 1), np.uint32); ninl = C.c_uint32()
    counts = np.zeros((len(si), 4), np.uint32)
    r = lib().orc_essential_batch(ba.ctypes.data, bb.ctypes.data, n, si.ctypes.data, len(si), thresh, 1e-12, 1000,
                                  pose.ctypes.data, C.byref(best), inl.ctypes.data, C.byref(ninl), counts.ctypes.data)
    if r != 0:
        return None
    return pose, best.value, inl[:ninl.value].copy(), counts


# ---- PnP: Lambda Twist + WorldToCamera residual (p3p_oracle.c) -----------------------------------------
def p3p_poses(bearings3, world3):
    """LambdaTwist::estimate on 3 (bearing [3], homogeneous world point [4]) samples -> [k,3,4] poses."""
    b = _f64(bearings3).reshape(3, 3); w = _f64(world3).reshape(3, 4)
    out = np.empty((4, 3, 4), np.float64)
    k = lib().orc_p3p_poses(b.ctypes.data, w.ctypes.data, out.ctypes.data)
    return out[:k].copy()


def w2c_residual(pose, bearing, world):
    pose = _f64(pose); bearing = _f64(bearing); world = _f64(world)
    return lib().orc_w2c_residual(pose.ctypes.data, bearing.ctypes.data, world.ctypes.data)


def p3p_batch(bearings, world, sample_idx, thresh):
    b = _f64(bearings); w = _f64(world)
    si = np.ascontiguousarray(sample_idx, np.uint32).reshape(-1, 3)
    n = len(b)
    pose = np.empty((3, 4), np.float64); best = C.c_uint32(); inl = np.empty(max(n, 1), np.uint32); ninl = C.c_uint32()
    counts = np.zeros((len(si), 4), np.uint32)
    r = lib().orc_p3p_batch(b.ctypes.data, w.ctypes.data, n, si.ctypes.data, len(si), thresh, pose.ctypes.data,
                            C.byref(best), inl.ctypes.data, C.byref(ninl), counts.ctypes.data)
    if r != 0:
        return None
    return pose, best.value, inl[:ninl.value].copy(), counts


def best_of_views(knn_out, nq, landmarks, view_idx, nviews, better_by=24):
    """cv-sfm/src/lib.rs:1489-1532 (oracle/match_oracle.c: orc_best_of_views).  knn_out [n_views, cap, k] of
    NB_DTYPE; landmarks [blocks, cap] u32.  Returns (best [nq,3,2] u32, decision [nq] u32)."""
    knn_out = np.ascontiguousarray(knn_out, NB_DTYPE)
    n_views, cap, k = knn_out.shape
    landmarks = np.ascontiguousarray(landmarks, np.uint32)
    vi = np.ascontiguousarray(view_idx, np.uint32); nv = np.ascontiguousarray(nviews, np.uint32)
    best = np.zeros((nq, 3, 2), np.uint32); dec = np.zeros(nq, np.uint32)
    lib().orc_best_of_views(knn_out.ctypes.data, nq, cap, n_views, k, landmarks.ctypes.data, vi.ctypes.data, nv.ctypes.data,
                            better_by, best.ctypes.data, dec.ctypes.data)
    return best, dec


def landmark_pairs(best, decision, world, merge_ok=None, n_world=None, merged_base=0, obs_counts=None):
    """cv-sfm/src/lib.rs:1516-1532, 1549-1563, 1583-1604 (oracle/match_oracle.c: orc_landmark_matches): the (feature, world
    row) list of one frame from the best-of-views output; world [rows, 4] f64 (w < 0: no robust triangulation).  merge_ok
    [nq] (the caller's are_landmarks_sharing_view verdicts) admits decision-2 features as merged matches; their world point is
    row merged_base + feature; n_world = rows addressed by landmark key (default: all of them)."""
    best = np.ascontiguousarray(best, np.uint32).reshape(-1, 3, 2)
    dec = np.ascontiguousarray(decision, np.uint32)
    W = np.ascontiguousarray(world, np.float64).reshape(-1, 4)
    out = np.zeros((max(len(dec), 1), 2), np.uint32)
    mk = None if merge_ok is None else np.ascontiguousarray(merge_ok, np.uint8)
    L = lib()
    L.orc_landmark_matches.restype = C.c_uint32
    L.orc_landmark_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    if obs_counts is not None:
        # ... in the order the reference's consensus sees them: stable sort by descending observation count (lib.rs:1561-1574)
        ob = np.ascontiguousarray(obs_counts, np.uint32)
        L.orc_landmark_matches_ordered.restype = C.c_uint32
        L.orc_landmark_matches_ordered.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                                   C.c_uint32, C.c_void_p]
        n = L.orc_landmark_matches_ordered(best.ctypes.data, dec.ctypes.data, None if mk is None else mk.ctypes.data, ob.ctypes.data,
                                           len(dec), W.ctypes.data, len(W) if n_world is None else n_world, merged_base,
                                           out.ctypes.data)
        return out[:n].copy()
    n = L.orc_landmark_matches(best.ctypes.data, dec.ctypes.data, None if mk is None else mk.ctypes.data, len(dec), W.ctypes.data,
                               len(W) if n_world is None else n_world, merged_base, out.ctypes.data)
    return out[:n].copy()


class ArrsacParams(C.Structure):
    """rs_arrsac_params (include/akz.h), restated here so that tests/ can drive the oracle without the product's
    Python package."""
    _fields_ = [("struct_size", C.c_uint32), ("n_hypotheses", C.c_uint32), ("block_size", C.c_uint32),
                ("init_blocks", C.c_uint32), ("max_candidates", C.c_uint32), ("flags", C.c_uint32),
                ("threshold", C.c_double), ("sprt_delta", C.c_double), ("sprt_ratio", C.c_double), ("seed", C.c_uint64),
                ("estimations_per_block", C.c_uint32), ("reserved", C.c_uint32)]


def arrsac_draw(seed, h, n, k):
    out = np.empty(k, np.uint32)
    lib().orc_arrsac_draw(seed, h, n, k, out.ctypes.data)
    return out


def arrsac(a, b, threshold, n_hypotheses, seed=0, sample_idx=None, block_size=64, init_blocks=4, max_candidates=1024,
           bound=True, sprt=True, sprt_delta=0.05, sprt_ratio=1e3, p3p=False, estimations_per_block=0, halve=False):
    """oracle/arrsac_oracle.c: orc_arrsac — the specification of rs_essential_arrsac / rs_p3p_arrsac.
    Returns (pose, inliers, best_id, stats dict) or None."""
    a = _f64(a); b = _f64(b)
    n = len(a)
    prm = ArrsacParams()
    prm.struct_size = C.sizeof(ArrsacParams)
    prm.n_hypotheses, prm.block_size, prm.init_blocks, prm.max_candidates = n_hypotheses, block_size, init_blocks, max_candidates
    prm.flags = (1 if bound else 0) | (2 if sprt else 0) | (4 if halve else 0)
    prm.threshold, prm.sprt_delta, prm.sprt_ratio, prm.seed = float(threshold), sprt_delta, sprt_ratio, seed
    prm.estimations_per_block = estimations_per_block
    si = None
    if sample_idx is not None:
        si = np.ascontiguousarray(sample_idx, np.uint32).reshape(-1, 3 if p3p else 8)
        prm.n_hypotheses = len(si)
    pose = np.empty((3, 4), np.float64); best = C.c_uint32(); ninl = C.c_uint32()
    inl = np.empty(max(n, 1), np.uint32); st = np.zeros(5, np.uint32)
    r = lib().orc_arrsac(1 if p3p else 0, a.ctypes.data, b.ctypes.data, n, si.ctypes.data if si is not None else None,
                         C.byref(prm), pose.ctypes.data, C.byref(best), inl.ctypes.data, C.byref(ninl), st.ctypes.data)
    if r != 0:
        return None
    stats = {"residuals_evaluated": int(st[0]) | (int(st[1]) << 32), "survivors": int(st[2]), "blocks": int(st[3]),
             "poses": int(st[4]) * 4}
    return pose, inl[:ninl.value].copy(), best.value, stats


def _arrsac_params(threshold, n_hypotheses, seed, block_size, init_blocks, max_candidates, bound, sprt, sprt_delta, sprt_ratio,
                   estimations_per_block, halve):
    prm = ArrsacParams()
    prm.struct_size = C.sizeof(ArrsacParams)
    prm.n_hypotheses, prm.block_size, prm.init_blocks, prm.max_candidates = n_hypotheses, block_size, init_blocks, max_candidates
    prm.flags = (1 if bound else 0) | (2 if sprt else 0) | (4 if halve else 0)
    prm.threshold, prm.sprt_delta, prm.sprt_ratio, prm.seed = float(threshold), sprt_delta, sprt_ratio, seed
    prm.estimations_per_block = estimations_per_block
    return prm


def shuffle_order(scene_seed, n):
    out = np.empty(max(n, 1), np.uint32)
    lib().orc_shuffle_order(scene_seed, n, out.ctypes.data)
    return out[:n]


def scene_seed(seed, scene):
    return int(lib().orc_scene_seed(seed, scene))


def arrsac_pairs(kps_a, kps_b, pairs, cam_a, cam_b, threshold, n_hypotheses, scene=0, shuffle=True, seed=0, block_size=64,
                 init_blocks=4, max_candidates=1024, bound=True, sprt=True, sprt_delta=0.05, sprt_ratio=1e3,
                 estimations_per_block=0, halve=False):
    """oracle/arrsac_oracle.c: orc_arrsac_pairs — one scene of rs_essential_arrsac_batch_device: keypoints + match pairs
    -> calibrated bearings -> (seeded shuffle) -> the ARRSAC-shaped consensus.  cam = (fx, fy, cx, cy, skew, k1 or None).
    Returns dict(pose, inliers, best_id, stats, bearings_a, bearings_b, order) (best_id 0xFFFFFFFF: no model)."""
    ka = np.ascontiguousarray(kps_a, dtype=KP_DTYPE); kb = np.ascontiguousarray(kps_b, dtype=KP_DTYPE)
    pr = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
    n = len(pr)
    prm = _arrsac_params(threshold, n_hypotheses, seed, block_size, init_blocks, max_candidates, bound, sprt, sprt_delta,
                         sprt_ratio, estimations_per_block, halve)

    def cam(c):
        return np.array([c[0], c[1], c[2], c[3], c[4], c[5] if c[5] is not None else 0.0], np.float64), int(c[5] is not None)
    ca, ua = cam(cam_a); cb, ub = cam(cam_b)
    pose = np.zeros((3, 4), np.float64); best = C.c_uint32(); ninl = C.c_uint32()
    inl = np.empty(max(n, 1), np.uint32); st = np.zeros(5, np.uint32)
    ba = np.zeros((max(n, 1), 3), np.float64); bb = np.zeros((max(n, 1), 3), np.float64); order = np.arange(max(n, 1), dtype=np.uint32)
    lib().orc_arrsac_pairs(ka.ctypes.data, kb.ctypes.data, pr.ctypes.data, n, ca.ctypes.data, ua, cb.ctypes.data, ub, scene,
                           int(shuffle), C.byref(prm), pose.ctypes.data, C.byref(best), inl.ctypes.data, C.byref(ninl),
                           st.ctypes.data, ba.ctypes.data, bb.ctypes.data, order.ctypes.data)
    stats = {"residuals_evaluated": int(st[0]) | (int(st[1]) << 32), "survivors": int(st[2]), "blocks": int(st[3]),
             "poses": int(st[4]) * 4}
    return {"pose": pose, "inliers": inl[:ninl.value].copy(), "best_id": best.value, "stats": stats,
            "bearings_a": ba[:n], "bearings_b": bb[:n], "order": order[:n]}


def p3p_arrsac_pairs(kps, pairs, world, cam, threshold, n_hypotheses, scene=0, shuffle=True, seed=0, block_size=64,
                     init_blocks=4, max_candidates=1024, bound=True, sprt=True, sprt_delta=0.05, sprt_ratio=1e3,
                     estimations_per_block=0, halve=False):
    """oracle/arrsac_oracle.c: orc_p3p_arrsac_pairs — one scene of rs_p3p_arrsac_batch_device: keypoints + (feature, world
    point) index pairs + homogeneous world points [..][4] -> bearings / points -> (seeded shuffle) -> the ARRSAC-shaped
    consensus over Lambda Twist hypotheses.  Returns dict(pose, inliers, best_id, stats, bearings, world, order)."""
    k = np.ascontiguousarray(kps, dtype=KP_DTYPE)
    pr = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
    W = np.ascontiguousarray(world, np.float64).reshape(-1, 4)
    n = len(pr)
    prm = _arrsac_params(threshold, n_hypotheses, seed, block_size, init_blocks, max_candidates, bound, sprt, sprt_delta,
                         sprt_ratio, estimations_per_block, halve)
    c = np.array([cam[0], cam[1], cam[2], cam[3], cam[4], cam[5] if cam[5] is not None else 0.0], np.float64)
    pose = np.zeros((3, 4), np.float64); best = C.c_uint32(); ninl = C.c_uint32()
    inl = np.empty(max(n, 1), np.uint32); st = np.zeros(5, np.uint32)
    ba = np.zeros((max(n, 1), 3), np.float64); wo = np.zeros((max(n, 1), 4), np.float64); order = np.arange(max(n, 1), dtype=np.uint32)
    lib().orc_p3p_arrsac_pairs(k.ctypes.data, pr.ctypes.data, n, c.ctypes.data, int(cam[5] is not None), W.ctypes.data, scene,
                               int(shuffle), C.byref(prm), pose.ctypes.data, C.byref(best), inl.ctypes.data, C.byref(ninl),
                               st.ctypes.data, ba.ctypes.data, wo.ctypes.data, order.ctypes.data)
    stats = {"residuals_evaluated": int(st[0]) | (int(st[1]) << 32), "survivors": int(st[2]), "blocks": int(st[3]),
             "poses": int(st[4]) * 4}
    return {"pose": pose, "inliers": inl[:ninl.value].copy(), "best_id": best.value, "stats": stats,
            "bearings": ba[:n], "world": wo[:n], "order": order[:n]}


def pairs_in_range(pairs, limit_a, limit_b):
    """oracle/arrsac_oracle.c: orc_pairs_in_range — the refusal rule of the batched device entry points: False when the
    pair list names an index outside its keypoint block / the world table (the scene then ends with "no model")."""
    pr = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
    return bool(lib().orc_pairs_in_range(pr.ctypes.data, len(pr), int(limit_a), int(limit_b)))


# ---- the CPU baseline build (bench.py's cpu_baseline legs only) ----------------------------------------------------
_FAST_PATH = os.path.join(_HERE, "liboracle_fast.so")
_fast = None


def _host_signature():
    """-march=native code must run on the host it was built on: the build is keyed by the CPU's feature flags."""
    import hashlib
    try:
        with open("/proc/cpuinfo") as f:
            flags = sorted({ln for ln in f if ln.startswith(("flags", "model name"))})
    except OSError:
        flags = []
    return hashlib.sha1("".join(flags).encode()).hexdigest()


def fast_lib():
    """liboracle_fast.so: the same sources at -O3 -march=native -fopenmp (oracle/Makefile, target `fast`), rebuilt when the
    sources or the host CPU changed.  Only its batch entry is bound."""
    global _fast
    if _fast is None:
        sig_path = _FAST_PATH + ".host"
        src = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c") or f == "Makefile"]
        stale = (not os.path.exists(_FAST_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_FAST_PATH) for s in src)
                 or not os.path.exists(sig_path) or open(sig_path).read() != _host_signature())
        if stale:
            subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "fast"])
            with open(sig_path, "w") as f:
                f.write(_host_signature())
        L = C.CDLL(_FAST_PATH)
        _bind_batch(L)
        _fast = L
    return _fast


def _bind_batch(L):
    L.orc_threads_available.restype = C.c_int
    L.orc_extract_match_many_u8.argtypes = [C.POINTER(Config), C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]


def extract_match_many(frames, threads=1, cap=8192, match=True, param_u=24, cfg=None, fast=True):
    """oracle/batch_oracle.c: extract every frame ([n,h,w] u8) and match frame i symmetrically (d0 + param_u < d1) against
    frame i-1, on `threads` OpenMP threads of the fast build (fast=False: the -O2 checker, single thread).  Returns a list
    of (keypoints, descriptors, pairs-or-None) per frame."""
    frames = np.ascontiguousarray(frames, np.uint8)
    n, h, w = frames.shape
    L = fast_lib() if fast else lib()
    if not fast:
        _bind_batch(L)
    cfg = cfg if cfg is not None else default_config()
    kps = np.zeros((n, cap), KP_DTYPE); descs = np.zeros((n, cap, 64), np.uint8); counts = np.zeros(n, np.uint32)
    pairs = np.zeros((n, cap, 2), np.uint32); npairs = np.zeros(n, np.uint32)
    r = L.orc_extract_match_many_u8(C.byref(cfg), w, h, frames.ctypes.data, n, int(threads), cap, kps.ctypes.data, descs.ctypes.data,
                                    counts.ctypes.data, int(match), param_u, pairs.ctypes.data, npairs.ctypes.data)
    assert r == 0, "a frame has more keypoints than cap"
    return [(kps[i, :counts[i]].copy(), descs[i, :counts[i]].copy(), pairs[i, :npairs[i]].copy() if (match and i > 0) else None)
            for i in range(n)]


def extract_many_intra(frames, threads=0, cap=8192, cfg=None):
    """oracle/batch_oracle.c: orc_extract_many_intra_u8 — the frames one after the other, each parallel at the akaze
    crate's own `rayon` points only (ORC_OPT_INTRA), on `threads` OpenMP threads of the fast build.  Returns a list of
    (keypoints, descriptors) per frame."""
    frames = np.ascontiguousarray(frames, np.uint8)
    n, h, w = frames.shape
    L = fast_lib()
    L.orc_extract_many_intra_u8.argtypes = [C.POINTER(Config), C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_uint32,
                                            C.c_void_p, C.c_void_p, C.c_void_p]
    cfg = cfg if cfg is not None else default_config()
    kps = np.zeros((n, cap), KP_DTYPE); descs = np.zeros((n, cap, 64), np.uint8); counts = np.zeros(n, np.uint32)
    r = L.orc_extract_many_intra_u8(C.byref(cfg), w, h, frames.ctypes.data, n, int(threads), cap, kps.ctypes.data,
                                    descs.ctypes.data, counts.ctypes.data)
    assert r == 0, "a frame has more keypoints than cap"
    return [(kps[i, :counts[i]].copy(), descs[i, :counts[i]].copy()) for i in range(n)]


def threads_available():
    return int(fast_lib().orc_threads_available())


def luma(img):
    """DynamicImage::grayscale() restated (oracle/color_oracle.c: orc_luma): HxWx3/4 uint8 / uint16 / float32 -> HxW."""
    a = np.ascontiguousarray(img)
    fmt = {np.dtype(np.uint8): 0, np.dtype(np.float32): 1, np.dtype(np.uint16): 2}[a.dtype]
    h, w, ch = a.shape
    out = np.empty((h, w), a.dtype)
    lib().orc_luma(a.ctypes.data, fmt, ch, w, h, w * ch, out.ctypes.data)
    return out
