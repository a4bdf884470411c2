"""Host-side mirror of cv-sfm's frame registration against recent views, for a micro-batch of new frames, device-resident.

  VSlam::add_frame -> hasher.hash_bag(descriptors)                              cv-sfm/src/lib.rs:672
  register_frame_subset: descriptor_features.knn(descriptor, 3) per view match  cv-sfm/src/lib.rs:1462-1486 (up to 32 views,
                                                                                settings.rs:449-450 tracking_recent_frames)
  landmark dedup, three best, unique-match / merge decision                     cv-sfm/src/lib.rs:1489-1532
  duplicate-landmark filter, FeatureWorldMatch list                             cv-sfm/src/lib.rs:1549-1604
  single_view_consensus.model_inliers(&LambdaTwist, matches_3d)                 cv-sfm/src/lib.rs:1619-1622
                                                                                (vslam-sandbox/src/main.rs:105-111: Arrsac,
                                                                                16384 hypotheses, 1024 candidates, 256 per block)

The reference walks this per feature on the CPU; here one call enqueues, for ALL frames of a micro-batch,
hm_hash_bag_device -> hm_knn_batch_device (k = 3, frames x views problems) -> hm_best_of_views_batch_device ->
hm_landmark_matches_batch_device -> rs_p3p_arrsac_batch_device on the blocks where akz_extract_batch_device left them; nothing
returns to the host in between.  What stays with the caller is the reference's control plane: which views a frame is matched
against, which landmark each stored feature observes, the table of triangulated landmarks, and for the merge candidates
(decision 2) the graph test are_landmarks_sharing_view plus the merged triangulation — handed in as a mask and extra world
rows between match_views() and consensus().  There is no CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check
from .knn import Matcher
from .ransac import EssentialConsensus


def _u32(vals):
    a = np.ascontiguousarray(vals, np.uint32).reshape(-1)
    return a, a.ctypes.data_as(C.c_void_p)


class Registration:
    """The chain for micro-batches of up to `max_frames` frames against up to `n_views` views each (feature blocks of
    `cap` entries).  The consensus parameters default to vslam-sandbox's single-view consensus."""

    def __init__(self, torch, cap, max_frames, n_views, codewords, camera, device=0, threshold=1e-5, n_hypotheses=16384,
                 max_candidates=1024, estimations_per_block=256, block_size=64, better_by=24, seed=0, max_matches=None):
        self.torch = torch
        self.dev = torch.device("cuda", device)
        self.cap, self.F, self.V, self.k = cap, max_frames, n_views, 3
        self.better_by = better_by
        cw = np.ascontiguousarray(codewords, np.uint8).reshape(-1, 64)
        if len(cw) == 0 or len(cw) % 32:
            raise ValueError("the codeword count must be a positive multiple of 32")
        self.n_codewords = len(cw)
        self.d_codewords = torch.from_numpy(cw).to(self.dev)
        # (bulk work: least urgent, so that the consensus' chain of small launches — most urgent, rs_create — finds compute units)
        self.matcher = Matcher(max(cap, self.n_codewords), device=device, low_priority=True)
        n_max = max_matches or cap
        blocks = (n_max + block_size - 1) // block_size
        self.cons = EssentialConsensus(n_max, n_hypotheses + estimations_per_block * blocks, device=device)
        self.cons.reserve(max_frames)
        self.prm = self.cons.make_params(threshold, n_hypotheses=n_hypotheses, seed=seed, block_size=block_size, init_blocks=1,
                                         max_candidates=max_candidates, halve=True, sprt=True,
                                         estimations_per_block=estimations_per_block)
        self.cam = self.cons.camera(camera)
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=self.dev)
        F, V, k = max_frames, n_views, self.k
        self.hash = z((F, self.n_codewords // 8), torch.uint8)
        self.words = z((F, cap, 2), torch.int32)
        self.knn = z((F, V, cap, k, 2), torch.int32)
        # What the consensus reads and writes exists twice: call t + 1's matching (matcher stream) then overlaps call t's
        # consensus (its own stream); a set is reused only after the consensus that read it has finished (event).
        self._sets = [dict(best=z((F, cap, 3, 2), torch.int32), decision=z((F, cap), torch.int32), pairs=z((F, cap, 2), torch.int32),
                           npairs=z((F,), torch.int32), pose=z((F, 12), torch.float64), best_id=z((F,), torch.int32),
                           inliers=z((F, cap), torch.int32), n_inliers=z((F,), torch.int32), stats=z((F, 32), torch.uint8),
                           done=torch.cuda.Event(), used=False) for _ in range(2)]
        self._calls = 0
        self._hm_s = torch.cuda.ExternalStream(self.hm_stream(), device=self.dev)
        self._rs_s = torch.cuda.ExternalStream(self.rs_stream(), device=self.dev)
        self._select(0)

    def _select(self, i):
        for key, val in self._sets[i].items():
            if key not in ("done", "used"):
                setattr(self, key, val)

    def hm_stream(self):
        return _lib.lib().hm_stream(self.matcher.handle)

    def rs_stream(self):
        return self.cons.stream()

    def enqueue(self, d_kps, d_descs, d_counts, frame_blocks, view_blocks, d_landmarks, d_world, n_world, stream_to_wait=None,
                shuffle=None, d_merge_ok=None, d_obs_counts=None):
        """match_views() followed by consensus(): the whole chain in one go (see the two for the arguments).  With
        d_obs_counts the consensus sees the matches in register_frame_subset's own order (cv-sfm/src/lib.rs:1561-1574)."""
        self.match_views(d_descs, d_counts, frame_blocks, view_blocks, d_landmarks, stream_to_wait=stream_to_wait)
        self.consensus(d_kps, d_counts, d_world, n_world, shuffle=shuffle, d_merge_ok=d_merge_ok, d_obs_counts=d_obs_counts)

    def match_views(self, d_descs, d_counts, frame_blocks, view_blocks, d_landmarks, stream_to_wait=None):
        """frame_blocks [F]: block index (into d_descs / d_counts / d_kps, [..][cap] each) of every new frame;
        view_blocks [F][V]: the blocks of the views each frame is matched against; d_landmarks [blocks][cap] int32: the
        landmark key observed by every feature of every stored block.  All d_* are torch tensors on the device.  Enqueues
        hash_bag, the k = 3 searches and the best-of-views decision on the matcher's stream and returns.  Outputs:
        self.hash, self.best / decision (slot = position in frame_blocks) — of THIS call until the next match_views() (two
        output sets alternate)."""
        L = _lib.lib()
        cur = self._sets[self._calls & 1]
        self._select(self._calls & 1)
        self._calls += 1
        self._cur = cur
        if cur["used"]:
            self._hm_s.wait_event(cur["done"])           # the consensus of two calls ago has read this set's pair lists
        F = len(frame_blocks)
        V = len(view_blocks[0])
        assert F <= self.F and V <= self.V and all(len(v) == V for v in view_blocks)
        fb, fb_p = _u32(frame_blocks)
        iq, iq_p = _u32(np.repeat(fb, V))
        it, it_p = _u32(view_blocks)
        self._fb = fb
        h = self.matcher.handle
        # hash_bag runs over the frames' blocks where they lie: block b of d_descs -> row b of the hash table of THIS call
        # (the new frames are contiguous in every caller so far: hash the span)
        b0, b1 = int(fb.min()), int(fb.max()) + 1
        assert b1 - b0 <= self.F
        check(L.hm_hash_bag_device(h, d_descs[b0:b1].data_ptr(), d_counts[b0:b1].data_ptr(), self.cap, b1 - b0,
                                   self.d_codewords.data_ptr(), self.n_codewords, self.hash.data_ptr(), self.words.data_ptr(),
                                   stream_to_wait), "hm_hash_bag_device")
        self.hash_block0 = b0
        check(L.hm_knn_batch_device(h, d_descs.data_ptr(), d_counts.data_ptr(), d_descs.data_ptr(), d_counts.data_ptr(), self.cap,
                                    iq_p, it_p, F * V, self.k, self.knn.data_ptr(), None), "hm_knn_batch_device")
        check(L.hm_best_of_views_batch_device(h, self.knn.data_ptr(), d_counts.data_ptr(), fb_p, self.cap, it_p, F, V, self.k,
                                              d_landmarks.data_ptr(), d_counts.data_ptr(), self.better_by, self.best.data_ptr(),
                                              self.decision.data_ptr(), None), "hm_best_of_views_batch_device")

    def consensus(self, d_kps, d_counts, d_world, n_world, shuffle=None, d_merge_ok=None, stream_to_wait=None, d_obs_counts=None):
        """The duplicate-landmark filter, the FeatureWorldMatch lists and single_view_consensus.model_inliers for the frames of
        the last match_views().  d_world [rows][4] f64: rows [0, n_world) indexed by landmark key.  d_merge_ok [F][cap] uint8
        (optional): the caller's are_landmarks_sharing_view verdicts for the merge candidates (decision 2,
        cv-sfm/src/lib.rs:1521-1531, read from self.decision after match_views()) — non-zero admits ([best0, best1], feature)
        as a match, whose world point is row n_world + f * cap + feature of d_world (triangulate_merged_landmark_robust; d_world
        then has n_world + F * cap rows); stream_to_wait = the stream the mask and those rows were written on.  Without a mask no
        merge candidate becomes a match — equal to the reference exactly when none passes its graph test.
        d_obs_counts [n_world] uint32 (optional): observations of every landmark key — the match lists then leave in the
        reference's order (stable sort by descending observation count, cv-sfm/src/lib.rs:1561-1574) and the consensus takes
        them as they are (shuffle defaults to False); without it the lists are in feature order and shuffled with the seed
        (shuffle defaults to True).  Outputs: self.pairs / npairs, self.pose / best_id / inliers / n_inliers / stats."""
        if shuffle is None:
            shuffle = d_obs_counts is None
        L = _lib.lib()
        cur, fb = self._cur, self._fb
        F = len(fb)
        _, fb_p = _u32(fb)
        rows = n_world if d_merge_ok is None else n_world + F * self.cap
        assert d_world.shape[0] >= rows
        check(L.hm_landmark_matches_ordered_batch_device(self.matcher.handle, self.best.data_ptr(), self.decision.data_ptr(),
                                                         None if d_merge_ok is None else d_merge_ok.data_ptr(),
                                                         None if d_obs_counts is None else d_obs_counts.data_ptr(), d_counts.data_ptr(),
                                                         fb_p, self.cap, F, d_world.data_ptr(), n_world, self.pairs.data_ptr(),
                                                         self.npairs.data_ptr(), stream_to_wait), "hm_landmark_matches_ordered_batch_device")
        self.cons.p3p_model_inliers_batch_device(d_kps.data_ptr(), self.cap, [int(b) for b in fb], self.pairs.data_ptr(),
                                                 self.npairs.data_ptr(), d_world.data_ptr(), rows, self.cam, self.prm,
                                                 self.pose.data_ptr(), self.best_id.data_ptr(), self.inliers.data_ptr(),
                                                 self.n_inliers.data_ptr(), self.stats.data_ptr(), shuffle=shuffle,
                                                 stream_to_wait=self.hm_stream())
        cur["done"].record(self._rs_s)
        cur["used"] = True

    def sync(self):
        check(_lib.lib().hm_sync(self.matcher.handle), "hm_sync")
        self.cons.sync()

    def close(self):
        self.cons.close()
        self.matcher.close()
