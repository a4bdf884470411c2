"""Host-side mirror of the two-view geometric-verification call sites over rs_* of include/akz.h.

  camera.calibrate(keypoint)                      cv-pinhole/src/lib.rs:108-117 (plain), :191-202 (K1 distortion)
  consensus.model_inliers(&EightPoint::new(), matches)   akaze/tests/estimate_pose.rs:63-67, tutorial ch5 main.rs:70-72
The reference's `arrsac` sampler is an un-vendored crate; here the minimal samples are an explicit argument
(`sample_idx`, 8 match indices per hypothesis) and every hypothesis is scored on the MI355X.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, check


@dataclass
class CameraIntrinsics:
    """cv_pinhole::CameraIntrinsics (focals, principal_point, skew)."""
    focals: tuple
    principal_point: tuple
    skew: float = 0.0
    k1: float = None      # set -> CameraIntrinsicsK1Distortion

    def calibrate(self, keypoints):
        """Pixel keypoints (structured array with x, y) -> [n,3] unit bearings."""
        kps = np.ascontiguousarray(keypoints, dtype=KP_DTYPE)
        intr = np.array([self.focals[0], self.focals[1], self.principal_point[0], self.principal_point[1], self.skew],
                        np.float64)
        out = np.empty((len(kps), 3), np.float64)
        check(_lib.lib().rs_calibrate(intr.ctypes.data, int(self.k1 is not None), float(self.k1 or 0.0),
                                      kps.ctypes.data, len(kps), out.ctypes.data), "rs_calibrate")
        return out


class EssentialConsensus:
    """Owns one rs_ctx.  model_inliers() is the batched Consensus::model_inliers for EightPoint."""

    def __init__(self, max_matches=8192, max_hypotheses=16384, device=0):
        self._h = C.c_void_p()
        self.max_hyp = max_hypotheses
        check(_lib.lib().rs_create(device, max_matches, max_hypotheses, C.byref(self._h)), "rs_create")

    def close(self):
        if self._h:
            _lib.lib().rs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def model_inliers(self, bearings_a, bearings_b, sample_idx, threshold):
        """Returns (pose [3,4] = [R | t], inlier indices, best_id) or None when no sample gave a model."""
        a = np.ascontiguousarray(bearings_a, np.float64); b = np.ascontiguousarray(bearings_b, np.float64)
        si = np.ascontiguousarray(sample_idx, np.uint32).reshape(-1, 8)
        n = len(a)
        pose = np.empty((3, 4), np.float64); best = C.c_uint32(); ninl = C.c_uint32()
        inl = np.empty(max(n, 1), np.uint32)
        check(_lib.lib().rs_essential_batch(self._h, a.ctypes.data, b.ctypes.data, n, si.ctypes.data, len(si),
                                            float(threshold), pose.ctypes.data, C.byref(best), inl.ctypes.data, n,
                                            C.byref(ninl)), "rs_essential_batch")
        if best.value == 0xFFFFFFFF:
            return None
        return pose, inl[:ninl.value].copy(), best.value

    def arrsac_model_inliers(self, bearings_a, bearings_b, threshold, n_hypotheses=8192, seed=0, sample_idx=None,
                             block_size=64, init_blocks=4, max_candidates=1024, bound=True, sprt=True, sprt_delta=0.05,
                             sprt_ratio=1e3, p3p=False, estimations_per_block=0, halve=False):
        """Arrsac::new(threshold, Xoshiro256PlusPlus::seed_from_u64(seed)).initialization_hypotheses(n)
        .max_candidate_hypotheses(k).model_inliers(&EightPoint::new(), matches) in this library's shape (include/akz.h:
        rs_essential_arrsac).  p3p=True: the same for LambdaTwist (bearings_a = bearings [n,3], bearings_b = world
        points [n,4]; 3-match samples; rs_p3p_arrsac).  estimations_per_block: hypotheses re-sampled from the best
        pose's inliers after every block (.estimations_per_block(e)); halve: the candidate cap halves block by block.
        The context needs room for n_hypotheses + estimations_per_block x ceil(n / block_size) hypotheses.
        Returns (pose, inliers, best_id, stats dict) or None."""
        a = np.ascontiguousarray(bearings_a, np.float64); b = np.ascontiguousarray(bearings_b, np.float64)
        n = len(a)
        prm = _lib.ArrsacParams()
        prm.struct_size = C.sizeof(_lib.ArrsacParams)
        prm.n_hypotheses, prm.block_size, prm.init_blocks, prm.max_candidates = n_hypotheses, block_size, init_blocks, max_candidates
        prm.flags = ((_lib.RS_PRUNE_BOUND if bound else 0) | (_lib.RS_PRUNE_SPRT if sprt else 0)
                     | (_lib.RS_PRUNE_HALVE if halve else 0))
        prm.estimations_per_block, prm.reserved = estimations_per_block, 0
        prm.threshold, prm.sprt_delta, prm.sprt_ratio, prm.seed = float(threshold), sprt_delta, sprt_ratio, seed
        si = None
        if sample_idx is not None:
            si = np.ascontiguousarray(sample_idx, np.uint32).reshape(-1, 3 if p3p else 8)
            prm.n_hypotheses = len(si)
        pose = np.empty((3, 4), np.float64); best = C.c_uint32(); ninl = C.c_uint32()
        inl = np.empty(max(n, 1), np.uint32)
        st = _lib.ArrsacStats()
        fn = _lib.lib().rs_p3p_arrsac if p3p else _lib.lib().rs_essential_arrsac
        check(fn(self._h, a.ctypes.data, b.ctypes.data, n, si.ctypes.data if si is not None else None,
                 C.byref(prm), pose.ctypes.data, C.byref(best), inl.ctypes.data, n, C.byref(ninl), C.byref(st)),
              "rs_p3p_arrsac" if p3p else "rs_essential_arrsac")
        if best.value == 0xFFFFFFFF:
            return None
        stats = {"poses": st.poses, "survivors": st.survivors, "blocks": st.blocks,
                 "residuals_evaluated": st.residuals_evaluated, "residuals_exhaustive": st.residuals_exhaustive}
        return pose, inl[:ninl.value].copy(), best.value, stats

    @staticmethod
    def arrsac_samples(seed, n, n_hypotheses, sample_size=8):
        """The minimal samples rs_essential_arrsac (8) / rs_p3p_arrsac (3) draw on the device for (seed, n)."""
        out = np.empty((n_hypotheses, sample_size), np.uint32)
        check(_lib.lib().rs_arrsac_samples(seed, n, n_hypotheses, sample_size, out.ctypes.data), "rs_arrsac_samples")
        return out

    def p3p_model_inliers(self, bearings, world, sample_idx, threshold):
        """Consensus::model_inliers(&LambdaTwist::new(), ...): bearings [n,3], world [n,4] homogeneous
        (Projective form), sample_idx [n_hyp,3].  Returns (pose [3,4], inliers, best_id) or None."""
        b = np.ascontiguousarray(bearings, np.float64); w = np.ascontiguousarray(world, np.float64)
        si = np.ascontiguousarray(sample_idx, np.uint32).reshape(-1, 3)
        n = len(b)
        pose = np.empty((3, 4), np.float64); best = C.c_uint32(); ninl = C.c_uint32()
        inl = np.empty(max(n, 1), np.uint32)
        check(_lib.lib().rs_p3p_batch(self._h, b.ctypes.data, w.ctypes.data, n, si.ctypes.data, len(si), float(threshold),
                                      pose.ctypes.data, C.byref(best), inl.ctypes.data, n, C.byref(ninl)), "rs_p3p_batch")
        if best.value == 0xFFFFFFFF:
            return None
        return pose, inl[:ninl.value].copy(), best.value

    def poses(self, n_hyp):
        """(poses [n_hyp,4,3,4], ok [n_hyp,4]) of the last single-scene call (rs_debug_poses)."""
        P = np.zeros((n_hyp, 4, 3, 4), np.float64); ok = np.zeros((n_hyp, 4), np.uint32)
        check(_lib.lib().rs_debug_poses(self._h, P.ctypes.data, ok.ctypes.data, n_hyp), "rs_debug_poses")
        return P, ok

    # ---- micro-batch entry: every frame pair of the matcher's output in one chain of launches ----
    def reserve(self, max_scenes):
        check(_lib.lib().rs_batch_reserve(self._h, max_scenes), "rs_batch_reserve")

    @staticmethod
    def make_params(threshold, n_hypotheses=8192, seed=0, block_size=64, init_blocks=4, max_candidates=1024, bound=True,
                    sprt=True, sprt_delta=0.05, sprt_ratio=1e3, estimations_per_block=0, halve=False):
        prm = _lib.ArrsacParams()
        prm.struct_size = C.sizeof(_lib.ArrsacParams)
        prm.n_hypotheses, prm.block_size, prm.init_blocks, prm.max_candidates = n_hypotheses, block_size, init_blocks, max_candidates
        prm.flags = ((_lib.RS_PRUNE_BOUND if bound else 0) | (_lib.RS_PRUNE_SPRT if sprt else 0)
                     | (_lib.RS_PRUNE_HALVE if halve else 0))
        prm.estimations_per_block, prm.reserved = estimations_per_block, 0
        prm.threshold, prm.sprt_delta, prm.sprt_ratio, prm.seed = float(threshold), sprt_delta, sprt_ratio, seed
        return prm

    @staticmethod
    def camera(intr):
        """CameraIntrinsics (or (fx, fy, cx, cy, skew, k1-or-None)) -> rs_camera."""
        if isinstance(intr, CameraIntrinsics):
            intr = (intr.focals[0], intr.focals[1], intr.principal_point[0], intr.principal_point[1], intr.skew, intr.k1)
        c = _lib.Camera()
        c.fx, c.fy, c.cx, c.cy, c.skew = (float(v) for v in intr[:5])
        c.k1, c.use_k1, c.reserved = float(intr[5] or 0.0), int(intr[5] is not None), 0
        return c

    def model_inliers_batch_device(self, d_kps_a, d_kps_b, cap_per_img, ia, ib, d_pairs, d_npairs, cam_a, cam_b, params,
                                   d_pose, d_best_id, d_inliers, d_n_inliers, d_stats=None, shuffle=True, stream_to_wait=None):
        """rs_essential_arrsac_batch_device: all arguments named d_* are device pointers (ints); ia / ib host index lists.
        Enqueues and returns; sync() waits."""
        n = len(ia)
        a = (C.c_uint32 * n)(*ia); b = (C.c_uint32 * n)(*ib)
        check(_lib.lib().rs_essential_arrsac_batch_device(
            self._h, d_kps_a, d_kps_b, cap_per_img, a, b, d_pairs, d_npairs, n, C.byref(cam_a), C.byref(cam_b), C.byref(params),
            _lib.RS_BATCH_SHUFFLE if shuffle else 0, d_pose, d_best_id, d_inliers, d_n_inliers, d_stats, stream_to_wait),
            "rs_essential_arrsac_batch_device")

    def p3p_model_inliers_batch_device(self, d_kps, cap_per_img, ik, d_pairs, d_npairs, d_world, n_world, cam, params, d_pose,
                                       d_best_id, d_inliers, d_n_inliers, d_stats=None, shuffle=True, stream_to_wait=None):
        """rs_p3p_arrsac_batch_device: scene s = (feature of keypoint block ik[s], world point index) pairs; d_world
        [n_world][4] f64 (a scene naming a point >= n_world is refused: no model).  Enqueues and returns; sync() waits."""
        n = len(ik)
        k = (C.c_uint32 * n)(*ik)
        check(_lib.lib().rs_p3p_arrsac_batch_device(
            self._h, d_kps, cap_per_img, k, d_pairs, d_npairs, n, d_world, n_world, C.byref(cam), C.byref(params),
            _lib.RS_BATCH_SHUFFLE if shuffle else 0, d_pose, d_best_id, d_inliers, d_n_inliers, d_stats, stream_to_wait),
            "rs_p3p_arrsac_batch_device")

    def sync(self):
        check(_lib.lib().rs_sync(self._h), "rs_sync")

    def stream(self):
        return _lib.lib().rs_stream(self._h)

    def scene(self, scene, cap):
        """(bearings_a, bearings_b, order) of scene `scene` of the last batched call (rs_debug_scene)."""
        n = C.c_uint32()
        a = np.zeros((cap, 3), np.float64); b = np.zeros((cap, 3), np.float64); o = np.zeros(cap, np.uint32)
        check(_lib.lib().rs_debug_scene(self._h, scene, C.byref(n), a.ctypes.data, b.ctypes.data, o.ctypes.data, cap),
              "rs_debug_scene")
        return a[:n.value], b[:n.value], o[:n.value]

    def scene_world(self, scene, cap):
        """(bearings, world points [n][4], order) of scene `scene` of the last rs_p3p_arrsac_batch_device call."""
        n = C.c_uint32()
        a = np.zeros((cap, 3), np.float64); b = np.zeros((cap, 4), np.float64); o = np.zeros(cap, np.uint32)
        check(_lib.lib().rs_debug_scene_world(self._h, scene, C.byref(n), a.ctypes.data, b.ctypes.data, o.ctypes.data, cap),
              "rs_debug_scene_world")
        return a[:n.value], b[:n.value], o[:n.value]

    def far(self, poses, bearings_a, bearings_b, thresh):
        """[n_pose, n] bool: where the device's epipolar-plane bound alone rules a match out (rs_debug_far)."""
        P = np.ascontiguousarray(poses, np.float64).reshape(-1, 12)
        a = np.ascontiguousarray(bearings_a, np.float64).reshape(-1, 3); b = np.ascontiguousarray(bearings_b, np.float64).reshape(-1, 3)
        out = np.zeros((len(P), len(a)), np.uint8)
        check(_lib.lib().rs_debug_far(self._h, P.ctypes.data, len(P), a.ctypes.data, b.ctypes.data, len(a), float(thresh),
                                      out.ctypes.data), "rs_debug_far")
        return out.astype(bool)

    def residuals(self, poses, bearings_a, bearings_b, paired=False):
        """CameraToCamera::residual of every (pose, match) as the device evaluates it (rs_debug_residuals): [n_pose, n],
        or [n_pose, 2, n] (the pose and its mirror [R | -t]) with paired=True."""
        P = np.ascontiguousarray(poses, np.float64).reshape(-1, 12)
        a = np.ascontiguousarray(bearings_a, np.float64); b = np.ascontiguousarray(bearings_b, np.float64)
        out = np.zeros((len(P), 2, len(a)) if paired else (len(P), len(a)), np.float64)
        check(_lib.lib().rs_debug_residuals(self._h, P.ctypes.data, len(P), a.ctypes.data, b.ctypes.data, len(a), int(paired),
                                            out.ctypes.data), "rs_debug_residuals")
        return out

    def counts(self, n_hyp):
        out = np.zeros((n_hyp, 4), np.uint32)
        check(_lib.lib().rs_debug_counts(self._h, out.ctypes.data, out.size), "rs_debug_counts")
        return out
