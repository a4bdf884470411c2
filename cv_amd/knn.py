"""Host-side mirror of the brute-force matcher the reference builds from `space` + `bitarray`.

  space::LinearKnn { metric: Hamming, iter }.knn(query, num)   akaze/tests/estimate_pose.rs:82-88
  matching / symmetric_matching (better-by-24, strict)         tutorial-code/chapter5-…/src/main.rs:154-200
  matching / symmetric_matching (better_by, <=, <2 guard)      cv-sfm/src/lib.rs:3097-3133
  match_descriptors (Lowe ratio 0.5, f32)                      akaze/tests/estimate_pose.rs:78-97
Descriptors are [n,64] uint8 arrays (rows = BitArray<64>).  Everything runs on the MI355X through
hm_* of include/akz.h; there is no CPU fallback.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import NB_DTYPE, check

RULE_STRICT, RULE_BETTER_BY, RULE_LOWE = 0, 1, 2


class Hamming:
    """bitarray::Hamming metric marker."""


@dataclass
class Neighbor:
    """space::Neighbor<u32>."""
    index: int
    distance: int


class Matcher:
    """Owns one hm_ctx."""

    def __init__(self, max_descriptors=16384, device=0, kernel="fp4", low_priority=False, cus=0):
        """kernel: "fp4" (default, FP4 MFMA, target tiles by LDS-DMA), "fp4_regs" (the same with a register stage), "int8"
        (int8 MFMA) or "valu" (xor/popcount, k = 2 only).
        low_priority: the matcher's stream as least-urgent filler work (HM_OPT_STREAM_PRIORITY).
        cus: the matcher's stream on the last `cus` compute units of every XCD (0 = the whole chip)."""
        self._h = C.c_void_p()
        self.cap = max_descriptors
        flags = {"fp4": 0, "fp4_regs": _lib.HM_OPT_NO_LDS_DMA, "int8": _lib.HM_OPT_NO_FP4, "valu": _lib.HM_OPT_NO_MFMA}[kernel]
        flags |= _lib.HM_OPT_STREAM_PRIORITY if low_priority else 0
        if not 0 <= int(cus) <= 32:
            raise ValueError(f"cus = {cus}: the matcher's stream takes 0 (the whole chip) to 32 compute units per XCD")
        flags |= int(cus) << 16
        check(_lib.lib().hm_create_ex(device, max_descriptors, max_descriptors, flags, C.byref(self._h)), "hm_create_ex")

    def close(self):
        if self._h:
            _lib.lib().hm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def knn2(self, q, t):
        q = _desc(q); t = _desc(t)
        out = np.zeros((len(q), 2), NB_DTYPE)
        check(_lib.lib().hm_knn2(self._h, q.ctypes.data, len(q), t.ctypes.data, len(t), out.ctypes.data), "hm_knn2")
        return out

    def knn(self, q, t, k):
        """LinearKnn.knn(q, k) for every row of q, k in {1, 2, 3}: [nq, k] (index, distance); slots past
        len(t) hold the sentinel {2^22 - 1, 1023}."""
        q = _desc(q); t = _desc(t)
        out = np.zeros((len(q), k), NB_DTYPE)
        check(_lib.lib().hm_knn(self._h, q.ctypes.data, len(q), t.ctypes.data, len(t), k, out.ctypes.data), "hm_knn")
        return out

    def set_targets(self, t):
        """hm_set_targets: the target set stays on the device for the knn_targets() calls that follow.  Returns the upload's
        generation number (targets_generation() answers it for as long as this set is the resident one)."""
        t = _desc(t)
        check(_lib.lib().hm_set_targets(self._h, t.ctypes.data, len(t)), "hm_set_targets")
        return self.targets_generation()

    def targets_generation(self):
        return int(_lib.lib().hm_targets_generation(self._h))

    def knn_targets(self, q, k):
        """hm_knn_targets: LinearKnn.knn(q, k) of every row of q against the resident targets."""
        q = _desc(q)
        out = np.zeros((len(q), k), NB_DTYPE)
        check(_lib.lib().hm_knn_targets(self._h, q.ctypes.data, len(q), k, out.ctypes.data), "hm_knn_targets")
        return out

    def match(self, a, b, rule=RULE_STRICT, param_u=24, param_f=0.5, symmetric=True):
        a = _desc(a); b = _desc(b)
        cap = max(len(a), 1)
        pairs = np.zeros((cap, 2), np.uint32)
        n = C.c_uint32()
        check(_lib.lib().hm_match(self._h, a.ctypes.data, len(a), b.ctypes.data, len(b), rule, param_u, param_f,
                                  int(symmetric), pairs.ctypes.data, cap, C.byref(n)), "hm_match")
        return pairs[:n.value].copy()


def _desc(d):
    d = np.ascontiguousarray(d, np.uint8)
    return d.reshape(-1, 64)


_default = {}


def default_matcher(n=16384, device=0):
    m = _default.get(device)
    if m is None or m.cap < n:
        if m is not None:
            m.close()
        m = Matcher(max(n, 16384), device)
        _default[device] = m
    return m


class LinearKnn:
    """space::LinearKnn { metric, iter }: exact k-NN by scanning `iter` (the target descriptors)."""

    def __init__(self, metric=Hamming, iter=None, device=0):
        self.metric = metric
        self.device = device
        self.iter = iter if iter is not None else np.zeros((0, 64), np.uint8)

    @property
    def iter(self):
        return self._iter

    @iter.setter
    def iter(self, value):
        # a private, read-only copy: what is resident on the device cannot drift from what this object scans (a caller
        # mutating its own array in place does not reach it), and assigning a new set forgets the upload
        self._iter = np.array(_desc(value), np.uint8, copy=True)
        self._iter.setflags(write=False)
        self._upload = None              # (matcher, generation) of this object's upload

    def _resident(self):
        m = default_matcher(max(len(self._iter), 1), self.device)
        if self._upload is None or self._upload[0] is not m or m.targets_generation() != self._upload[1]:
            self._upload = (m, m.set_targets(self._iter))
        return m

    def knn(self, query, num):
        """Knn::knn(&self, query, num) -> Vec<Neighbor>, sorted by (distance, index), min(num, len) long.
        The device path implements num <= 3 (the reference asks for 2 when matching frame pairs and 3 when
        registering a frame against recent views, cv-sfm/src/lib.rs:1474).  `iter` is uploaded once and stays on the
        device between calls (hm_set_targets / hm_knn_targets; the upload's generation number is compared before every
        call, so another LinearKnn or a matching() call on the same matcher in between triggers a fresh upload); a call
        still costs one launch per query — match whole frames with knn_batch() or matching() where the caller's loop
        allows it."""
        if not 1 <= num <= 3:
            raise NotImplementedError("the MI355X matcher implements knn(query, k) for k <= 3")
        nn = self._resident().knn_targets(_desc(query)[:1], num)
        return [Neighbor(int(nn[0, i]["index"]), int(nn[0, i]["distance"])) for i in range(min(num, len(self._iter)))]

    def knn_batch(self, queries, num=2):
        """knn(q, num) for every row of `queries` in one launch: [nq,num] structured (index, distance)."""
        m = default_matcher(max(len(self._iter), len(queries), 1), self.device)
        return m.knn2(queries, self._iter) if num == 2 else m.knn(queries, self._iter, num)


def matching(a_descriptors, b_descriptors, better_by=24, strict=True, device=0):
    """matching() of tutorial ch5 (strict: d0 + 24 < d1, main.rs:162) or of cv-sfm (strict=False:
    d0 + better_by <= d1, lib.rs:3107).  Returns a list of Optional[int] like the reference."""
    a = _desc(a_descriptors); b = _desc(b_descriptors)
    if not strict and (len(a) < 2 or len(b) < 2):
        return []                                   # cv-sfm/src/lib.rs:3099-3101
    m = default_matcher(max(len(a), len(b), 1), device)
    pairs = m.match(a, b, RULE_STRICT if strict else RULE_BETTER_BY, better_by, 0.0, symmetric=False)
    out = [None] * len(a)
    for ai, bi in pairs:
        out[int(ai)] = int(bi)
    return out


def symmetric_matching(a, b, better_by=24, strict=True, device=0):
    """symmetric_matching() (ch5 main.rs:183-200 / cv-sfm lib.rs:3116-3133): list of [a, b]."""
    a = _desc(a); b = _desc(b)
    if not strict and (len(a) < 2 or len(b) < 2):
        return []
    m = default_matcher(max(len(a), len(b), 1), device)
    return m.match(a, b, RULE_STRICT if strict else RULE_BETTER_BY, better_by, 0.0, symmetric=True).tolist()


def match_descriptors(ds1, ds2, lowes_ratio=0.5, device=0):
    """match_descriptors() of akaze/tests/estimate_pose.rs:78-97: a->b only, Lowe ratio in f32."""
    m = default_matcher(max(len(ds1), len(ds2), 1), device)
    return [tuple(p) for p in m.match(ds1, ds2, RULE_LOWE, 0, lowes_ratio, symmetric=False).tolist()]
