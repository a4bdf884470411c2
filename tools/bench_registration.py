#!/usr/bin/env python3
"""SURVEY.md §8f rank 2 (the registration path of cv-sfm, cv-sfm/src/lib.rs:1452-1542,1619-1622): one new frame's
descriptors against 32 recent views with knn(., 3), device-resident, then Lambda Twist consensus over 8192 minimal
samples on 1000 landmark matches (30 % outliers).  Prints one JSON line; both stages checked against the oracle
on a sample.  --batch F adds the same matching for F new frames in one call chain (hm_knn_batch_device +
hm_best_of_views_batch_device: frame f against the 32 blocks before it)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from cv_amd import build  # noqa: E402
build.build()
from cv_amd import _lib  # noqa: E402
from cv_amd.knn import Matcher  # noqa: E402
from cv_amd.ransac import EssentialConsensus  # noqa: E402
from oracle import oracle as O  # noqa: E402
from test_oracle_ransac import _projective, _rot  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0x2E6)
cap, nviews, k = 8192, 32, 3
counts = rng.integers(4700, 5300, nviews).astype(np.int32)
views = rng.integers(0, 256, (nviews, cap, 64), dtype=np.uint8)
nq = 5000
q = np.zeros((cap, 64), np.uint8)
q[:nq] = rng.integers(0, 256, (nq, 64), dtype=np.uint8)
d_q, d_views = torch.from_numpy(q).to(dev), torch.from_numpy(views).to(dev)
d_nq = torch.tensor([nq], dtype=torch.int32, device=dev)
d_nv = torch.from_numpy(counts).to(dev)
out = torch.zeros((nviews, cap, k, 2), dtype=torch.int32, device=dev)
m = Matcher(cap)
idx = (C.c_uint32 * nviews)(*range(nviews))


def knn_views():
    _lib.check(L.hm_knn_views_device(m.handle, d_q.data_ptr(), d_nq.data_ptr(), d_views.data_ptr(), d_nv.data_ptr(), cap,
                                     idx, nviews, k, out.data_ptr(), None), "knn_views")


knn_views()
_lib.check(L.hm_sync(m.handle), "sync")
reps = 5
t0 = time.perf_counter()
for _ in range(reps):
    knn_views()
_lib.check(L.hm_sync(m.handle), "sync")
knn_s = (time.perf_counter() - t0) / reps
got = out[3].cpu().numpy()
want = O.knn(q[:200], views[3, :counts[3]], 3)
assert (got[:200, :, 0].astype(np.uint32) == want["index"]).all() and (got[:200, :, 1].astype(np.uint32) == want["distance"]).all()

# P3P consensus
n, n_hyp, thr = 1000, 8192, 1e-6
R, t = _rot(rng.random(3) * 0.8), rng.random(3)
pts = rng.random((n, 3)) * 4.0 - 2.0
pts[:, 2] += 6.0
cam = pts @ R.T + t
b = cam / np.linalg.norm(cam, axis=1, keepdims=True)
bad = rng.random(n) < 0.3
rb = rng.standard_normal((n, 3)); rb[:, 2] = np.abs(rb[:, 2]) + 0.5
b[bad] = (rb / np.linalg.norm(rb, axis=1, keepdims=True))[bad]
w = _projective(pts)
samples = np.stack([rng.choice(n, 3, replace=False) for _ in range(n_hyp)]).astype(np.uint32)
cons = EssentialConsensus(n, n_hyp)
cons.p3p_model_inliers(b, w, samples, thr)
t0 = time.perf_counter()
for _ in range(reps):
    pose, inl, best = cons.p3p_model_inliers(b, w, samples, thr)
p3p_s = (time.perf_counter() - t0) / reps
sub = 200
t0 = time.perf_counter()
wpose, wbest, winl, _ = O.p3p_batch(b, w, samples[:sub], thr)
cpu_s = time.perf_counter() - t0
g = cons.p3p_model_inliers(b, w, samples[:sub], thr)
assert g[2] == wbest and np.array_equal(g[1], winl) and g[0].tobytes() == wpose.tobytes()
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64, help="frames of the batched leg (0: skip)")
args = ap.parse_args()
batch = None
if args.batch:
    F = args.batch
    NB = F + nviews
    bc = rng.integers(4700, 5300, NB).astype(np.int32)
    d_blocks = torch.randint(0, 256, (NB, cap, 64), dtype=torch.uint8, device=dev)
    d_bc = torch.from_numpy(bc).to(dev)
    d_lm = torch.randint(0, 200000, (NB, cap), dtype=torch.int32, device=dev)
    iq = np.repeat(np.arange(nviews, NB, dtype=np.uint32), nviews)                       # frame f = block nviews + f
    it = (np.arange(F, dtype=np.uint32)[:, None] + np.arange(nviews, dtype=np.uint32)[None, :]).reshape(-1)
    fr = np.arange(nviews, NB, dtype=np.uint32)
    d_knn = torch.zeros((F, nviews, cap, k, 2), dtype=torch.int32, device=dev)
    d_best = torch.zeros((F, cap, 3, 2), dtype=torch.int32, device=dev)
    d_dec = torch.zeros((F, cap), dtype=torch.int32, device=dev)
    p = lambda a: np.ascontiguousarray(a, np.uint32).ctypes.data_as(C.c_void_p)

    def run_batch():
        _lib.check(L.hm_knn_batch_device(m.handle, d_blocks.data_ptr(), d_bc.data_ptr(), d_blocks.data_ptr(), d_bc.data_ptr(), cap,
                                         p(iq), p(it), F * nviews, k, d_knn.data_ptr(), None), "knn_batch")
        _lib.check(L.hm_best_of_views_batch_device(m.handle, d_knn.data_ptr(), d_bc.data_ptr(), p(fr), cap, p(it), F, nviews, k,
                                                   d_lm.data_ptr(), d_bc.data_ptr(), 24, d_best.data_ptr(), d_dec.data_ptr(), None), "bov")
    run_batch()
    _lib.check(L.hm_sync(m.handle), "sync")
    t0 = time.perf_counter()
    for _ in range(3):
        run_batch()
    _lib.check(L.hm_sync(m.handle), "sync")
    bs = (time.perf_counter() - t0) / 3
    # one (frame, view) against the oracle
    f, v = F // 2, 5
    blk = d_blocks.cpu().numpy()
    want = O.knn(blk[nviews + f, :100], blk[f + v, :bc[f + v]], 3)
    got = d_knn[f, v].cpu().numpy()
    assert (got[:100, :, 0].astype(np.uint32) == want["index"]).all() and (got[:100, :, 1].astype(np.uint32) == want["distance"]).all()
    dist = float(sum(int(bc[nviews + f_]) * int(bc[f_:f_ + nviews].sum()) for f_ in range(F)))
    batch = {"frames": F, "ms": round(bs * 1e3, 2), "frames_per_s": round(F / bs, 1), "knn_distances_per_s": round(dist / bs, 1),
             "what": f"hm_knn_batch_device (k = 3, {F} x {nviews} problems) + hm_best_of_views_batch_device, one call each"}
print(json.dumps({
    "batched_matching": batch,
    "workload": f"registration: {nq} descriptors x {nviews} views of ~{int(counts.mean())} (knn 3, device-resident) + "
                f"Lambda Twist consensus, {n_hyp} samples x {n} matches (30% outliers), host buffers in/out",
    "knn3_views_ms": round(knn_s * 1e3, 3), "knn_distances_per_s": round(nq * float(counts.sum()) / knn_s, 1),
    "p3p_ms_per_scene": round(p3p_s * 1e3, 3), "p3p_hypotheses_per_s": round(n_hyp / p3p_s, 1), "inliers": int(len(inl)),
    "cpu_oracle_p3p": {"hypotheses_per_s": round(sub / cpu_s, 1), "cores": 1, "sample": f"first {sub} samples"},
    "parity": "knn(.,3) indices/distances on 200 queries of one view and the consensus result on the sampled "
              "hypotheses identical to the oracle"}))
