#!/usr/bin/env python3
"""hash_bag of 128 frames x ~5000 descriptors over a 4096-word codebook, device-resident,
on one MI355X, beside the CPU oracle on a bounded sample of the frames.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cv_amd import build  # noqa: E402
build.build()
import torch  # noqa: E402
from cv_amd import _lib  # noqa: E402
from cv_amd.knn import Matcher  # noqa: E402
from oracle import oracle as O  # noqa: E402

rng = np.random.default_rng(0x15A)
nf, cap, ncw = 128, 8192, 4096
counts = rng.integers(4600, 5400, nf).astype(np.int32)
blocks = rng.integers(0, 256, (nf, cap, 64), dtype=np.uint8)
cw = rng.integers(0, 256, (ncw, 64), dtype=np.uint8)
dev = torch.device("cuda", 0)
d_blocks, d_counts, d_cw = (torch.from_numpy(x).to(dev) for x in (blocks, counts, cw))
d_hash = torch.zeros((nf, ncw // 8), dtype=torch.uint8, device=dev)
d_words = torch.zeros((nf, cap, 2), dtype=torch.int32, device=dev)
m = Matcher(cap)
L = _lib.lib()


def run():
    _lib.check(L.hm_hash_bag_device(m.handle, d_blocks.data_ptr(), d_counts.data_ptr(), cap, nf, d_cw.data_ptr(), ncw,
                                    d_hash.data_ptr(), d_words.data_ptr(), None), "hash_bag_device")


run()
_lib.check(L.hm_sync(m.handle), "sync")
reps = 10
t0 = time.perf_counter()
for _ in range(reps):
    run()
_lib.check(L.hm_sync(m.handle), "sync")
gpu_s = (time.perf_counter() - t0) / reps
got = d_hash.cpu().numpy()
sub = 2
t0 = time.perf_counter()
for f in range(sub):
    want, _ = O.hash_bag(blocks[f, :counts[f]], cw)
    assert (want == got[f]).all()
cpu_s = (time.perf_counter() - t0) / sub
dists = float(counts.sum()) * ncw
print(json.dumps({
    "workload": f"hash_bag: {nf} frames x ~{int(counts.mean())} descriptors x {ncw} codewords, device-resident",
    "gpu_ms_per_batch": round(gpu_s * 1e3, 3), "frames_per_s": round(nf / gpu_s, 1),
    "distances_per_s": round(dists / gpu_s, 1),
    "cpu_oracle": {"frames_per_s": round(1.0 / cpu_s, 2), "cores": 1, "sample": f"first {sub} frames"},
    "parity": "hash bytes identical to the oracle on the sampled frames (hamming-lsh itself: unpinned)"}))
