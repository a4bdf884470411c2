"""Host-side mirror of cv-sfm's frame-level place recognition over hm_hash_* of include/akz.h.

  HammingHasher::<64, 512>::new_with_codewords(codewords::codewords())   cv-sfm/src/lib.rs:205,216
  hasher.hash_bag(features.iter().map(|(d, _)| d))                       cv-sfm/src/lib.rs:672
  lsh_to_frame.insert(lsh, frame) / .knn_values(&lsh, num)               cv-sfm/src/lib.rs:684, :622-624
The codebook is the caller's ([n_codewords, 64] uint8, n_codewords = 8 x hash bytes; cv-sfm ships 4096 words in
cv-sfm/src/codewords.rs).  Everything runs on the MI355X; there is no CPU fallback.  The hashing crate
(hamming-lsh 0.3.2) is not vendored in the reference, see oracle/lsh_oracle.c for what is restated.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import NB_DTYPE, check
from .knn import _desc, default_matcher


class HammingHasher:
    """hamming_lsh::HammingHasher<64, H> with H = len(codewords) / 8."""

    def __init__(self, codewords, device=0):
        self.codewords = _desc(codewords)
        if len(self.codewords) == 0 or len(self.codewords) % 32:
            raise ValueError("the codeword count must be a positive multiple of 32")
        self.device = device

    @classmethod
    def new_with_codewords(cls, codewords, device=0):
        return cls(codewords, device)

    @property
    def hash_bytes(self):
        return len(self.codewords) // 8

    def hash_bag(self, features, return_words=False):
        """BitArray<H> (as [H] uint8) of one frame's descriptors; with return_words also the [n] (index, distance)
        nearest-codeword table of the features."""
        f = _desc(features)
        m = default_matcher(max(len(f), len(self.codewords), 1), self.device)
        h = np.zeros(self.hash_bytes, np.uint8)
        words = np.zeros(len(f), NB_DTYPE)
        check(_lib.lib().hm_hash_bag(m.handle, f.ctypes.data, len(f), self.codewords.ctypes.data, len(self.codewords),
                                     h.ctypes.data, words.ctypes.data), "hm_hash_bag")
        return (h, words) if return_words else h


class HashIndex:
    """The lsh_to_frame map of cv-sfm (an HggLite there): insert(hash, value), knn_values(hash, num) — exact here."""

    def __init__(self, hash_bytes=512, device=0):
        self.hash_bytes = hash_bytes
        self.device = device
        self._hashes = np.zeros((0, hash_bytes), np.uint8)
        self._values = []

    def __len__(self):
        return len(self._values)

    def insert(self, lsh, value):
        lsh = np.ascontiguousarray(lsh, np.uint8).reshape(1, self.hash_bytes)
        self._hashes = np.concatenate([self._hashes, lsh])
        self._values.append(value)

    def knn_values(self, lsh, num):
        """[(Neighbor-like (index, distance), value)], ascending (distance, insertion order), min(num, len) long."""
        q = np.ascontiguousarray(lsh, np.uint8).reshape(self.hash_bytes)
        out = np.zeros(max(num, 1), NB_DTYPE)
        n = C.c_uint32()
        m = default_matcher(1, self.device)
        check(_lib.lib().hm_hash_knn(m.handle, q.ctypes.data, self._hashes.ctypes.data, len(self._values),
                                     self.hash_bytes, num, out.ctypes.data, C.byref(n)), "hm_hash_knn")
        return [((int(o["index"]), int(o["distance"])), self._values[int(o["index"])]) for o in out[:n.value]]
