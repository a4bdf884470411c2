"""ctypes loader for cv_amd/lib/libakz.so (the C ABI of include/akz.h).

There is no CPU fallback: if the library is missing or no HIP device is usable, every entry point
raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported from here.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libakz.so")


class AkzError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = status
        msg = lib().akz_strerror(status).decode()
        hip = lib().akz_last_hip_error()
        if status == -5 and hip:
            msg += f" (hip {hip}: {lib().akz_last_hip_error_string().decode()})"
        super().__init__(f"{what}: {msg}" if what else msg)


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


class Options(C.Structure):
    """akz_options (include/akz.h): behaviour switches of one context; all-zero = the defaults."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("flags", C.c_uint32), ("fed_block", C.c_uint32),
        ("sup_capacity", C.c_uint32), ("max_candidates", C.c_uint32), ("desc_tile_shift", C.c_uint32),
        ("stream_waves", C.c_uint32), ("stream_min_waves", C.c_uint32), ("arith", C.c_uint32), ("cu_ss", C.c_uint32),
        ("cu_kp", C.c_uint32), ("resident_min_frames", C.c_uint32), ("reserved", C.c_uint32 * 4),
    ]


OPT_KEEP_ALL, OPT_NO_FRAME_PAIRS, OPT_SERIAL_SUPPRESSION, OPT_NO_PIPELINE = 1, 2, 4, 8
OPT_STREAM_PRIORITY, OPT_CONTRAST_EXACT, OPT_CONTRAST_FORCE_ODD, OPT_TILE_KERNELS, OPT_SERIAL_DET, OPT_SPLIT_FRONT_FED, OPT_EQUAL_PRIORITY, OPT_NO_RESIDENT_LEVELS = 16, 32, 64, 128, 256, 512, 1024, 2048
HM_OPT_NO_FP4, HM_OPT_NO_MFMA, HM_OPT_STREAM_PRIORITY, HM_OPT_NO_LDS_DMA = 1, 2, 4, 8
FMT_U8, FMT_F32, FMT_U16 = 0, 1, 2
ARITH_REDUCE_PAIRWISE, ARITH_FMA, ARITH_HALF_SEQUENTIAL = 1, 2, 4


BOOL_OPTIONS = ("keep_all", "frame_pairs", "parallel_suppression", "pipeline", "stream_priority", "stream_kernels",
                "det_side_stream", "fuse_front_fed", "resident_levels")


def make_options(keep_all=False, frame_pairs=True, parallel_suppression=True, pipeline=True, stream_priority=True,
                 contrast="fine", fed_block=0, sup_capacity=0, max_candidates=0, desc_tile_shift=0, stream_kernels=True,
                 stream_waves=0, stream_min_waves=0, det_side_stream=True, fuse_front_fed=True, arith=0, cu_ss=0, cu_kp=0,
                 resident_levels=True, resident_min_frames=0):
    """Options with readable names.  contrast: "fine" (default), "exact", "force_odd".  arith: AKZ_ARITH_* bits (1: pairwise
    reduce_add, 2: fused mul_add, 4: sequential 2 x 2 sum) — the one option that changes results (include/akz.h)."""
    o = Options()
    o.struct_size = C.sizeof(Options)
    o.flags = ((OPT_KEEP_ALL if keep_all else 0) | (0 if frame_pairs else OPT_NO_FRAME_PAIRS)
               | (0 if parallel_suppression else OPT_SERIAL_SUPPRESSION) | (0 if pipeline else OPT_NO_PIPELINE)
               | (0 if stream_priority else OPT_EQUAL_PRIORITY) | (0 if stream_kernels else OPT_TILE_KERNELS)
               | (0 if det_side_stream else OPT_SERIAL_DET) | (0 if fuse_front_fed else OPT_SPLIT_FRONT_FED)
               | (0 if resident_levels else OPT_NO_RESIDENT_LEVELS)
               | {"fine": 0, "exact": OPT_CONTRAST_EXACT, "force_odd": OPT_CONTRAST_FORCE_ODD}[contrast])
    o.fed_block, o.sup_capacity, o.max_candidates, o.desc_tile_shift = fed_block, sup_capacity, max_candidates, desc_tile_shift
    o.stream_waves, o.stream_min_waves = stream_waves, stream_min_waves
    o.arith = arith
    o.cu_ss, o.cu_kp = cu_ss, cu_kp
    o.resident_min_frames = resident_min_frames
    return o


class ArrsacParams(C.Structure):
    """rs_arrsac_params (include/akz.h)."""
    _fields_ = [("struct_size", C.c_uint32), ("n_hypotheses", C.c_uint32), ("block_size", C.c_uint32),
                ("init_blocks", C.c_uint32), ("max_candidates", C.c_uint32), ("flags", C.c_uint32),
                ("threshold", C.c_double), ("sprt_delta", C.c_double), ("sprt_ratio", C.c_double), ("seed", C.c_uint64),
                ("estimations_per_block", C.c_uint32), ("reserved", C.c_uint32)]


class ArrsacStats(C.Structure):
    _fields_ = [("poses", C.c_uint32), ("survivors", C.c_uint32), ("blocks", C.c_uint32), ("reserved", C.c_uint32),
                ("residuals_evaluated", C.c_uint64), ("residuals_exhaustive", C.c_uint64)]


RS_PRUNE_BOUND, RS_PRUNE_SPRT, RS_PRUNE_HALVE = 1, 2, 4
RS_BATCH_SHUFFLE = 1


class Camera(C.Structure):
    """rs_camera (include/akz.h)."""
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("skew", C.c_double),
                ("k1", C.c_double), ("use_k1", C.c_int32), ("reserved", C.c_int32)]


class OverflowInfo(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("needed_candidates", C.c_uint32), ("candidate_capacity", C.c_uint32),
                ("keypoint_capacity", C.c_uint32)]


class LevelInfo(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("octave", C.c_uint32), ("sublevel", C.c_uint32),
        ("esigma", C.c_double), ("etime", C.c_double),
        ("n_fed_steps", C.c_uint32), ("deriv_sigma", C.c_uint32),
    ]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("response", "<f4"), ("size", "<f4"),
                     ("angle", "<f4"), ("octave", "<u4"), ("class_id", "<u4")])
NB_DTYPE = np.dtype([("index", "<u4"), ("distance", "<u4")])
assert KP_DTYPE.itemsize == 28 and NB_DTYPE.itemsize == 8

ABI_VERSION = 8          # include/akz.h AKZ_ABI_VERSION this file's argtypes were written against

# every symbol include/akz.h declares (tests check that the library exports all of them)
ABI_SYMBOLS = [
    "akz_config_default", "akz_create", "akz_create_ex", "akz_destroy", "akz_extract_gray_u8", "akz_extract_gray_u16",
    "akz_extract_gray_f32", "akz_extract_color",
    "akz_extract_batch", "akz_extract_batch_device", "akz_sync", "akz_stream", "akz_scale_space_device",
    "akz_last_overflow", "akz_num_levels", "akz_level", "akz_fed_tau", "akz_debug_get_level", "akz_debug_get_contrast",
    "akz_debug_get_keypoints", "akz_debug_portable_math", "akz_debug_orientation_masks", "akz_debug_rcp_error", "akz_gaussian_kernel", "akz_horizontal_filter", "akz_vertical_filter",
    "akz_half_size", "akz_sample_colors_rgb8", "hm_create", "hm_create_ex", "hm_destroy", "hm_knn2", "hm_knn", "hm_knn_views_device", "hm_knn_batch_device", "hm_best_of_views_device", "hm_best_of_views_batch_device", "hm_match",
    "hm_match_batch_device", "hm_sync", "hm_hash_bag", "hm_hash_bag_device", "hm_hash_knn", "hm_timing_enable",
    "hm_timing_get",
    "hm_stream", "rs_create", "rs_destroy", "rs_calibrate", "rs_essential_batch", "rs_essential_arrsac", "rs_p3p_arrsac", "rs_arrsac_samples",
    "rs_p3p_batch", "rs_debug_counts", "rs_debug_poses", "rs_batch_reserve", "rs_essential_arrsac_batch_device", "rs_sync",
    "rs_stream", "rs_debug_scene", "rs_debug_residuals", "rs_p3p_arrsac_batch_device", "hm_landmark_pairs_batch_device", "hm_landmark_matches_batch_device", "hm_landmark_matches_ordered_batch_device", "hm_set_targets", "hm_targets_generation", "hm_knn_targets", "rs_debug_scene_world", "rs_debug_far",
    "akz_strerror", "akz_last_hip_error", "akz_last_hip_error_string", "akz_version", "akz_abi_version",
    "akz_timing_enable", "akz_timing_reset", "akz_timing_get",
    "akz_comm_unique_id", "akz_comm_create", "akz_comm_destroy", "akz_comm_shift_blocks", "akz_comm_allgather_blocks", "akz_comm_sync",
    "akz_comm_stream", "akz_comm_rank", "akz_comm_world", "akz_comm_timing", "akz_comm_last_error_string",
]

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m cv_amd.build` (hipcc, gfx950). "
            "cv_amd has no CPU fallback.")
    # PyTorch-ROCm bundles its own HIP/HSA runtime.  A process must initialise only one: if libakz pulls in the
    # system runtime first and torch is imported afterwards (or the other way round with the roles swapped), the
    # second copy finds "no ROCm-capable device".  Importing torch first makes libakz resolve against the
    # runtime that is already loaded, whatever order the caller uses.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, u32 = C.c_void_p, C.c_int32, C.c_uint32
    L.akz_strerror.restype = C.c_char_p
    L.akz_strerror.argtypes = [i32]
    L.akz_last_hip_error_string.restype = C.c_char_p
    L.akz_version.restype = C.c_char_p
    if not hasattr(L, "akz_abi_version"):   # a library from before the ABI carried a version: the same remedy
        raise RuntimeError(f"{LIB_PATH} exports no akz_abi_version (a build older than ABI 6), this binding was written "
                           f"against {ABI_VERSION} (include/akz.h AKZ_ABI_VERSION): rebuild with `python -m cv_amd.build`")
    L.akz_abi_version.restype = C.c_uint32
    L.akz_debug_rcp_error.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
    L.hm_targets_generation.restype = C.c_uint64
    L.hm_targets_generation.argtypes = [C.c_void_p]
    if L.akz_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} exports ABI {L.akz_abi_version()}, this binding was written against {ABI_VERSION} "
                           "(include/akz.h AKZ_ABI_VERSION): rebuild with `python -m cv_amd.build`")
    L.akz_config_default.argtypes = [C.POINTER(Config)]
    L.akz_create.argtypes = [C.POINTER(Config), i32, i32, i32, i32, u32, C.POINTER(vp)]
    L.akz_create_ex.argtypes = [C.POINTER(Config), i32, i32, i32, i32, u32, C.POINTER(Options), C.POINTER(vp)]
    L.akz_destroy.argtypes = [vp]
    L.akz_extract_gray_u16.argtypes = [vp, vp, i32, i32, i32, vp, vp, u32, C.POINTER(u32)]
    L.akz_extract_gray_u8.argtypes = [vp, vp, i32, i32, i32, vp, vp, u32, C.POINTER(u32)]
    L.akz_extract_gray_f32.argtypes = [vp, vp, i32, i32, i32, vp, vp, u32, C.POINTER(u32)]
    L.akz_extract_color.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, u32, C.POINTER(u32)]
    L.akz_extract_batch.argtypes = [vp, C.POINTER(vp), i32, i32, i32, i32, i32, vp, vp, u32, vp]
    L.akz_extract_batch_device.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, u32, vp, vp]
    L.akz_scale_space_device.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    L.akz_sync.argtypes = [vp]
    L.akz_stream.restype = vp
    L.akz_stream.argtypes = [vp]
    L.akz_last_overflow.argtypes = [vp, C.POINTER(OverflowInfo), u32, C.POINTER(u32)]
    L.akz_num_levels.argtypes = [vp, i32, i32, C.POINTER(i32)]
    L.akz_level.argtypes = [vp, i32, i32, i32, C.POINTER(LevelInfo)]
    L.akz_fed_tau.argtypes = [vp, i32, i32, i32, vp, u32, C.POINTER(u32)]
    L.akz_debug_get_level.argtypes = [vp, i32, i32, i32, vp]
    L.akz_debug_get_contrast.argtypes = [vp, i32, C.POINTER(C.c_double)]
    L.akz_debug_get_keypoints.argtypes = [vp, i32, i32, vp, u32, C.POINTER(u32)]
    L.akz_debug_portable_math.argtypes = [vp, i32, vp, vp, u32, vp]
    L.akz_debug_orientation_masks.argtypes = [vp, vp, vp, u32, vp, vp, vp]
    L.akz_gaussian_kernel.argtypes = [C.c_float, u32, vp]
    L.akz_horizontal_filter.argtypes = [vp, vp, i32, i32, vp, u32, vp]
    L.akz_vertical_filter.argtypes = [vp, vp, i32, i32, vp, u32, vp]
    L.akz_half_size.argtypes = [vp, vp, i32, i32, vp]
    L.akz_sample_colors_rgb8.argtypes = [vp, vp, i32, i32, i32, vp, u32, vp]
    L.hm_create.argtypes = [i32, u32, u32, C.POINTER(vp)]
    L.hm_create_ex.argtypes = [i32, u32, u32, u32, C.POINTER(vp)]
    L.hm_destroy.argtypes = [vp]
    L.hm_knn2.argtypes = [vp, vp, u32, vp, u32, vp]
    L.hm_knn.argtypes = [vp, vp, u32, vp, u32, u32, vp]
    L.hm_knn_views_device.argtypes = [vp, vp, vp, vp, vp, u32, vp, u32, u32, vp, vp]
    L.hm_knn_batch_device.argtypes = [vp, vp, vp, vp, vp, u32, vp, vp, u32, u32, vp, vp]
    L.hm_best_of_views_batch_device.argtypes = [vp, vp, vp, vp, u32, vp, u32, u32, u32, vp, vp, u32, vp, vp, vp]
    L.hm_best_of_views_device.argtypes = [vp, vp, vp, u32, vp, u32, u32, vp, vp, u32, vp, vp, vp]
    L.hm_set_targets.argtypes = [vp, vp, u32]
    L.hm_knn_targets.argtypes = [vp, vp, u32, u32, vp]
    L.hm_landmark_pairs_batch_device.argtypes = [vp, vp, vp, vp, vp, u32, u32, vp, u32, vp, vp, vp]
    L.hm_landmark_matches_batch_device.argtypes = [vp, vp, vp, vp, vp, vp, u32, u32, vp, u32, vp, vp, vp]
    L.hm_landmark_matches_ordered_batch_device.argtypes = [vp, vp, vp, vp, vp, vp, vp, u32, u32, vp, u32, vp, vp, vp]
    L.hm_match.argtypes = [vp, vp, u32, vp, u32, i32, u32, C.c_float, i32, vp, u32, C.POINTER(u32)]
    L.hm_match_batch_device.argtypes = [vp, vp, vp, vp, vp, u32, vp, vp, u32, i32, u32, C.c_float, i32, vp, vp, vp]
    L.hm_sync.argtypes = [vp]
    L.hm_hash_bag.argtypes = [vp, vp, u32, vp, u32, vp, vp]
    L.hm_hash_bag_device.argtypes = [vp, vp, vp, u32, u32, vp, u32, vp, vp, vp]
    L.hm_hash_knn.argtypes = [vp, vp, vp, u32, u32, u32, vp, C.POINTER(u32)]
    L.hm_timing_enable.argtypes = [vp, i32]
    L.hm_timing_get.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), i32]
    L.hm_stream.restype = vp
    L.hm_stream.argtypes = [vp]
    L.rs_create.argtypes = [i32, u32, u32, C.POINTER(vp)]
    L.rs_destroy.argtypes = [vp]
    L.rs_calibrate.argtypes = [vp, i32, C.c_double, vp, u32, vp]
    L.rs_essential_batch.argtypes = [vp, vp, vp, u32, vp, u32, C.c_double, vp, C.POINTER(u32), vp, u32, C.POINTER(u32)]
    L.rs_essential_arrsac.argtypes = [vp, vp, vp, u32, vp, C.POINTER(ArrsacParams), vp, C.POINTER(u32), vp, u32, C.POINTER(u32),
                                      C.POINTER(ArrsacStats)]
    L.rs_p3p_arrsac.argtypes = L.rs_essential_arrsac.argtypes
    L.rs_arrsac_samples.argtypes = [C.c_uint64, u32, u32, u32, vp]
    L.rs_p3p_batch.argtypes = [vp, vp, vp, u32, vp, u32, C.c_double, vp, C.POINTER(u32), vp, u32, C.POINTER(u32)]
    L.rs_debug_counts.argtypes = [vp, vp, u32]
    L.rs_debug_poses.argtypes = [vp, vp, vp, u32]
    L.rs_batch_reserve.argtypes = [vp, u32]
    L.rs_essential_arrsac_batch_device.argtypes = [vp, vp, vp, u32, vp, vp, vp, vp, u32, C.POINTER(Camera), C.POINTER(Camera),
                                                   C.POINTER(ArrsacParams), u32, vp, vp, vp, vp, vp, vp]
    L.rs_p3p_arrsac_batch_device.argtypes = [vp, vp, u32, vp, vp, vp, u32, vp, u32, C.POINTER(Camera), C.POINTER(ArrsacParams), u32,
                                             vp, vp, vp, vp, vp, vp]
    L.rs_debug_scene_world.argtypes = [vp, u32, C.POINTER(u32), vp, vp, vp, u32]
    L.rs_sync.argtypes = [vp]
    L.rs_stream.restype = vp
    L.rs_stream.argtypes = [vp]
    L.rs_debug_scene.argtypes = [vp, u32, C.POINTER(u32), vp, vp, vp, u32]
    L.rs_debug_residuals.argtypes = [vp, vp, u32, vp, vp, u32, i32, vp]
    L.rs_debug_far.argtypes = [vp, vp, u32, vp, vp, u32, C.c_double, vp]
    L.akz_comm_unique_id.argtypes = [vp]
    L.akz_comm_create.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
    L.akz_comm_destroy.argtypes = [vp]
    L.akz_comm_shift_blocks.argtypes = [vp, vp, vp, u32, u32, vp, vp, vp]
    L.akz_comm_allgather_blocks.argtypes = [vp, vp, vp, u32, u32, vp, vp, vp]
    L.akz_comm_sync.argtypes = [vp]
    L.akz_comm_stream.restype = vp
    L.akz_comm_stream.argtypes = [vp]
    L.akz_comm_rank.argtypes = [vp]
    L.akz_comm_world.argtypes = [vp]
    L.akz_comm_timing.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), i32]
    L.akz_comm_last_error_string.restype = C.c_char_p
    L.akz_timing_enable.argtypes = [vp, i32]
    L.akz_timing_reset.argtypes = [vp]
    L.akz_timing_get.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    _lib = L
    return L


def check(status, what=""):
    if status != 0:
        raise AkzError(status, what)


STREAM_LEGACY = 1   # include/akz.h AKZ_STREAM_LEGACY (= hipStreamLegacy)


def wait_handle(stream):
    """The `stream_to_wait` argument for a torch stream: its handle, or AKZ_STREAM_LEGACY for the legacy default stream —
    whose handle is 0, which the ABI reads as "nothing to wait for" (the library's streams are non-blocking: they do not
    synchronise with the default stream by themselves)."""
    h = stream.cuda_stream
    return h if h else STREAM_LEGACY
