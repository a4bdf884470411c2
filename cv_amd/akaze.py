"""Host-side mirror of the reference's `akaze` crate API over the C ABI (include/akz.h).

Same names, argument meaning and error behaviour as akaze/src/lib.rs of rust-cv/cv:
  Akaze (11 public fields, Default, new/sparse/dense)        lib.rs:109-185
  Akaze::extract / extract_from_gray_float_image / extract_path   lib.rs:295, 309, 361
  KeyPoint {point, response, size, octave, class_id, angle}  lib.rs:69-93
  akaze::image::{gaussian_kernel, horizontal_filter, vertical_filter, separable_filter, gaussian_blur}
                                                              image.rs:202-389
`extract` is infallible in the reference: keypoints whose descriptor samples leave the image are
silently dropped (descriptors.rs:23-31).  Here it additionally raises AkzError for device problems
(no GPU, out of memory, internal list overflow) — there is no CPU fallback.
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, AkzError, Config, LevelInfo, Options, check, make_options  # noqa: F401

USIZE_MAX = 2 ** 64 - 1
MAX_KEYPOINTS = 262144   # kAkzMaxKeypoints (cv_amd/csrc/akz_common.h)
BUF = {"Lt": 0, "Lsmooth": 1, "Lx": 2, "Ly": 3, "Ldet": 4, "Lflow": 5}


@dataclass
class KeyPoint:
    """akaze::KeyPoint (lib.rs:69-93)."""
    point: tuple
    response: float
    size: float
    octave: int
    class_id: int
    angle: float

    def image_point(self):
        """cv_core::ImagePoint::image_point (lib.rs:95-99)."""
        return (float(self.point[0]), float(self.point[1]))


def keypoints_from_array(arr):
    return [KeyPoint((float(k["x"]), float(k["y"])), float(k["response"]), float(k["size"]), int(k["octave"]),
                     int(k["class_id"]), float(k["angle"])) for k in arr]


@dataclass
class Akaze:
    """akaze::Akaze (lib.rs:109-185).  Field names and defaults are the reference's."""
    maximum_features: int = USIZE_MAX
    num_sublevels: int = 4
    max_octave_evolution: int = 4
    base_scale_offset: float = 1.6
    initial_contrast: float = 0.001
    contrast_percentile: float = 0.7
    contrast_factor_num_bins: int = 300
    derivative_factor: float = 1.5
    detector_threshold: float = 0.001
    descriptor_channels: int = 3
    descriptor_pattern_size: int = 10
    # not part of the reference struct: placement of the device context
    device: int = field(default=0, compare=False)
    max_keypoints: int = field(default=16384, compare=False)

    @classmethod
    def new(cls, threshold):            # lib.rs:147-152
        return cls(detector_threshold=threshold)

    @classmethod
    def sparse(cls):                    # lib.rs:157-159
        return cls.new(0.01)

    @classmethod
    def dense(cls):                     # lib.rs:164-166
        return cls.new(0.0001)

    @classmethod
    def default(cls):                   # lib.rs:169-185
        return cls()

    # ---- device context management -----------------------------------------------------
    def config(self):
        c = Config()
        c.maximum_features = min(int(self.maximum_features), USIZE_MAX)
        c.num_sublevels = self.num_sublevels
        c.max_octave_evolution = self.max_octave_evolution
        c.base_scale_offset = self.base_scale_offset
        c.initial_contrast = self.initial_contrast
        c.contrast_percentile = self.contrast_percentile
        c.contrast_factor_num_bins = self.contrast_factor_num_bins
        c.derivative_factor = self.derivative_factor
        c.detector_threshold = self.detector_threshold
        c.descriptor_channels = self.descriptor_channels
        c.descriptor_pattern_size = self.descriptor_pattern_size
        return c

    def _key(self):
        return (self.maximum_features, self.num_sublevels, self.max_octave_evolution, self.base_scale_offset,
                self.initial_contrast, self.contrast_percentile, self.contrast_factor_num_bins,
                self.derivative_factor, self.detector_threshold, self.descriptor_channels,
                self.descriptor_pattern_size, self.device, self.max_keypoints)

    def context(self, w, h, batch=1, options=None):
        """A device context able to process `batch` frames of w x h (cached on the instance).  `options`: an
        akz_options (cv_amd._lib.make_options) selecting fall-back kernels / parity taps; None = defaults."""
        ctx = self.__dict__.get("_ctx")
        okey = bytes(options) if options is not None else None
        if ctx is not None and (ctx.key != self._key() or ctx.okey != okey or w > ctx.max_w or h > ctx.max_h
                                or batch > ctx.max_batch):
            ctx.close()
            ctx = None
        if ctx is None:
            ctx = Context(self, w, h, batch, options)
            self.__dict__["_ctx"] = ctx
        return ctx

    def close(self):
        ctx = self.__dict__.pop("_ctx", None)
        if ctx is not None:
            ctx.close()

    # ---- the reference API ---------------------------------------------------------------
    def extract(self, image):
        """Akaze::extract (lib.rs:295).  `image`: HxW uint8 (Luma8), uint16 (Luma16) or float32 in
        [0,1] (a GrayFloatImage).  Returns (keypoints, descriptors): list[KeyPoint] and an [n,64] uint8
        array (BitArray<64> rows), index-aligned, ordered by response descending."""
        arr, descs = self.extract_arrays(image)
        return keypoints_from_array(arr), descs

    def extract_from_gray_float_image(self, float_image):
        """Akaze::extract_from_gray_float_image (lib.rs:309)."""
        return self.extract(np.asarray(float_image, dtype=np.float32))

    def extract_path(self, path):
        """Akaze::extract_path (lib.rs:361): decoding errors propagate (ImageResult in the reference)."""
        from PIL import Image  # decoding only; mirrors image::open
        im = Image.open(path)
        if im.mode in ("L", "I;16"):
            return self.extract(np.asarray(im))
        if im.mode == "LA":
            return self.extract(np.asarray(im)[..., 0])
        return self.extract(np.asarray(im.convert("RGB") if im.mode not in ("RGB", "RGBA") else im))

    def extract_arrays(self, image):
        """extract() returning the raw structured keypoint array instead of KeyPoint objects.  HxW images are the gray
        arms of GrayFloatImage::from_dynamic; HxWx3 / HxWx4 images (uint8, uint16, float32) the colour ones —
        DynamicImage::grayscale() runs on the device (akz_extract_color).

        The reference's lists are unbounded (maximum_features = usize::MAX, lib.rs:172); the library's have a capacity fixed
        at context creation.  A call that overflows it (AKZ_E_INTERNAL + akz_last_overflow) is repeated with a context of
        twice the capacity, up to the library's 262 144 per frame — the caller sees the reference's behaviour, not the cap."""
        img = np.asarray(image)
        if img.ndim == 3 and img.shape[2] in (3, 4):
            if img.dtype not in (np.uint8, np.uint16):
                img = np.ascontiguousarray(img, dtype=np.float32)
            run = lambda ctx: ctx.extract_color(img)
        elif img.ndim == 2:
            if img.dtype not in (np.uint8, np.uint16):   # Luma8 / Luma16 go to the device as they are (image.rs:47-66)
                img = np.ascontiguousarray(img, dtype=np.float32)
            run = lambda ctx: ctx.extract_batch([img])[0]
        else:
            raise ValueError("expected an HxW (gray) or HxWx3/4 (colour) image")
        h, w = img.shape[:2]
        while True:
            try:
                return run(self.context(w, h, 1))
            except AkzError as e:
                if e.status != -7 or self.max_keypoints >= MAX_KEYPOINTS:
                    raise
                self.max_keypoints = min(MAX_KEYPOINTS, 2 * int(self.max_keypoints))   # context() re-creates for the new key


class Context:
    """Owns one akz_ctx (device pyramid for up to `batch` frames of up to w x h)."""

    def __init__(self, akaze, w, h, batch, options=None):
        self.key = akaze._key()
        self.okey = bytes(options) if options is not None else None
        self.max_w, self.max_h, self.max_batch = w, h, batch
        self.max_kp = min(int(akaze.max_keypoints), MAX_KEYPOINTS)
        self._h = C.c_void_p()
        cfg = akaze.config()
        check(_lib.lib().akz_create_ex(C.byref(cfg), akaze.device, w, h, batch, self.max_kp,
                                       C.byref(options) if options is not None else None, C.byref(self._h)),
              "akz_create_ex")

    def close(self):
        if self._h:
            _lib.lib().akz_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def extract_batch(self, images):
        """images: list of same-shaped HxW uint8 or float32 arrays. Returns [(kp_array, desc[n,64])]."""
        n = len(images)
        imgs = [np.ascontiguousarray(im) for im in images]
        h, w = imgs[0].shape
        fmt = {np.dtype(np.uint8): _lib.FMT_U8, np.dtype(np.uint16): _lib.FMT_U16}.get(imgs[0].dtype, _lib.FMT_F32)
        if fmt == _lib.FMT_F32:
            imgs = [np.ascontiguousarray(im, dtype=np.float32) for im in imgs]
        assert all(im.shape == (h, w) and im.dtype == imgs[0].dtype for im in imgs)
        cap = self.max_kp
        # (outputs are not cleared: the library writes cnt[i] entries per frame and only those are returned)
        kps = np.empty((n, cap), KP_DTYPE)
        descs = np.empty((n, cap, 64), np.uint8)
        cnt = np.zeros(n, np.uint32)
        ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
        st = _lib.lib().akz_extract_batch(self._h, ptrs, fmt, n, w, h, w, kps.ctypes.data, descs.ctypes.data, cap,
                                          cnt.ctypes.data)
        if st == -7:
            raise AkzError(st, "akz_extract_batch: " + self.overflow_report())
        check(st, "akz_extract_batch")
        return [(kps[i, :cnt[i]].copy(), descs[i, :cnt[i]].copy()) for i in range(n)]

    def extract_color(self, image):
        """One HxWx3/4 uint8 / uint16 / float32 image through akz_extract_color.  Returns (kp_array, desc[n,64])."""
        img = np.ascontiguousarray(image)
        h, w, ch = img.shape
        fmt = {np.dtype(np.uint8): _lib.FMT_U8, np.dtype(np.uint16): _lib.FMT_U16, np.dtype(np.float32): _lib.FMT_F32}[img.dtype]
        cap = self.max_kp
        kps = np.empty(cap, KP_DTYPE); descs = np.empty((cap, 64), np.uint8); cnt = C.c_uint32()
        st = _lib.lib().akz_extract_color(self._h, img.ctypes.data, fmt, ch, w, h, w * ch, kps.ctypes.data, descs.ctypes.data, cap,
                                          C.byref(cnt))
        if st == -7:
            raise AkzError(st, "akz_extract_color: " + self.overflow_report())
        check(st, "akz_extract_color")
        return kps[:cnt.value].copy(), descs[:cnt.value].copy()

    def overflow_report(self):
        """akz_last_overflow as text: which frame's internal list overflowed and what it needed."""
        info = (_lib.OverflowInfo * self.max_batch)()
        n = C.c_uint32()
        if _lib.lib().akz_last_overflow(self._h, info, self.max_batch, C.byref(n)) != 0:
            return "overflow report unavailable"
        out = []
        for i in range(n.value):
            if info[i].flags & 1:
                out.append(f"frame {i}: a level holds {info[i].needed_candidates} extrema, capacity "
                           f"{info[i].candidate_capacity} (akz_options.max_candidates / max_keypoints)")
            if info[i].flags & 2:
                out.append(f"frame {i}: more than {info[i].keypoint_capacity} keypoints (max_keypoints)")
        return "; ".join(out) or "no list over capacity"

    # ---- parity taps -------------------------------------------------------------------
    def num_levels(self, w, h):
        n = C.c_int32()
        check(_lib.lib().akz_num_levels(self._h, w, h, C.byref(n)))
        return n.value

    def level(self, w, h, i):
        info = LevelInfo()
        check(_lib.lib().akz_level(self._h, w, h, i, C.byref(info)))
        return info

    def fed_tau(self, w, h, i):
        buf = np.zeros(256, np.float64)
        n = C.c_uint32()
        check(_lib.lib().akz_fed_tau(self._h, w, h, i, buf.ctypes.data, 256, C.byref(n)))
        return buf[:n.value].copy()

    def level_buffer(self, img, level, name, w, h):
        info = self.level(w, h, level)
        out = np.empty((info.height, info.width), np.float32)
        check(_lib.lib().akz_debug_get_level(self._h, img, level, BUF[name], out.ctypes.data), "akz_debug_get_level")
        return out

    def contrast(self, img):
        v = C.c_double()
        check(_lib.lib().akz_debug_get_contrast(self._h, img, C.byref(v)))
        return v.value

    def keypoints(self, img, stage):
        out = np.zeros(self.max_kp, KP_DTYPE)
        n = C.c_uint32()
        check(_lib.lib().akz_debug_get_keypoints(self._h, img, stage, out.ctypes.data, self.max_kp, C.byref(n)))
        return out[:n.value].copy()

    def sample_colors(self, rgb, kps):
        """bicubic::interpolate_bicubic(&image.to_rgb8(), kp.point.0, kp.point.1, Rgb([0, 0, 0])) for every
        keypoint (cv-sfm/src/lib.rs:2207-2216): rgb [h, w, 3] uint8, kps a KP_DTYPE array -> [n, 3] uint8."""
        rgb = np.ascontiguousarray(rgb, np.uint8)
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        out = np.zeros((len(kps), 3), np.uint8)
        check(_lib.lib().akz_sample_colors_rgb8(self._h, rgb.ctypes.data, rgb.shape[1], rgb.shape[0],
                                                rgb.strides[0], kps.ctypes.data, len(kps), out.ctypes.data),
              "akz_sample_colors_rgb8")
        return out

    def timing_enable(self, on=True):
        check(_lib.lib().akz_timing_enable(self._h, int(on)))

    def timing_reset(self):
        check(_lib.lib().akz_timing_reset(self._h))

    def timing_get(self, which):
        ms = C.c_double(); la = C.c_uint64(); un = C.c_uint64()
        check(_lib.lib().akz_timing_get(self._h, which, C.byref(ms), C.byref(la), C.byref(un)))
        return ms.value, la.value, un.value


def grayscale(rgb):
    """DynamicImage::grayscale() for RGB(A) 8/16-bit pixels as the `image` 0.24 crate computes it (integer
    Rec. 709: (2126 R + 7152 G + 722 B) / 10000, truncating).  The crate is not vendored in the reference
    checkout (akaze/Cargo.toml:16): colour-input parity is unpinned; gray inputs never come through here."""
    a = np.asarray(rgb)
    if a.ndim != 3 or a.shape[2] < 3 or a.dtype not in (np.uint8, np.uint16):
        raise ValueError("expected an HxWx3/4 uint8 or uint16 image")
    l = (2126 * a[..., 0].astype(np.uint64) + 7152 * a[..., 1].astype(np.uint64) + 722 * a[..., 2].astype(np.uint64)) // 10000
    return l.astype(a.dtype)


# ---- akaze::image ---------------------------------------------------------------------------------
def gaussian_kernel(r, kernel_size):
    """akaze::image::gaussian_kernel (image.rs:360-374); panics (ValueError) on an even size."""
    if kernel_size % 2 != 1:
        raise ValueError("kernel_size must be odd")
    out = np.empty(kernel_size, np.float32)
    check(_lib.lib().akz_gaussian_kernel(r, kernel_size, out.ctypes.data))
    return out


def _ctx_for(img, device=0):
    h, w = img.shape
    return Akaze(device=device).context(w, h, 1)


def horizontal_filter(image, kernel, ctx=None):
    """akaze::image::horizontal_filter (image.rs:202-251)."""
    img = np.ascontiguousarray(image, np.float32); k = np.ascontiguousarray(kernel, np.float32)
    c = ctx or _ctx_for(img); out = np.empty_like(img)
    check(_lib.lib().akz_horizontal_filter(c.handle, img.ctypes.data, img.shape[1], img.shape[0], k.ctypes.data,
                                           len(k), out.ctypes.data))
    return out


def vertical_filter(image, kernel, ctx=None):
    """akaze::image::vertical_filter (image.rs:253-331)."""
    img = np.ascontiguousarray(image, np.float32); k = np.ascontiguousarray(kernel, np.float32)
    c = ctx or _ctx_for(img); out = np.empty_like(img)
    check(_lib.lib().akz_vertical_filter(c.handle, img.ctypes.data, img.shape[1], img.shape[0], k.ctypes.data,
                                         len(k), out.ctypes.data))
    return out


def separable_filter(image, h_kernel, v_kernel, ctx=None):
    """akaze::image::separable_filter (image.rs:333-340): horizontal then vertical."""
    img = np.ascontiguousarray(image, np.float32)
    c = ctx or _ctx_for(img)
    return vertical_filter(horizontal_filter(img, h_kernel, c), v_kernel, c)


def gaussian_blur(image, r, ctx=None):
    """akaze::image::gaussian_blur (image.rs:383-389)."""
    if not r > 0.0:
        raise ValueError("sigma must be > 0.0")
    radius = int(np.ceil(np.float32(2.0) * np.float32(r)))
    k = gaussian_kernel(r, 2 * radius + 1)
    return separable_filter(image, k, k, ctx)


def half_size(image, ctx=None):
    """GrayFloatImage::half_size (image.rs:154-199)."""
    img = np.ascontiguousarray(image, np.float32)
    c = ctx or _ctx_for(img)
    out = np.empty((img.shape[0] // 2, img.shape[1] // 2), np.float32)
    check(_lib.lib().akz_half_size(c.handle, img.ctypes.data, img.shape[1], img.shape[0], out.ctypes.data))
    return out
