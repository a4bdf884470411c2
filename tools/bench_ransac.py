#!/usr/bin/env python3
"""BASELINE.json configs[3]: 10k eight-point hypotheses scored on 1000 matches (30 % outliers), threshold 1e-7,
on one MI355X, beside the CPU oracle on a bounded sample of the hypotheses.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cv_amd import build  # noqa: E402
build.build()
from cv_amd.ransac import EssentialConsensus  # noqa: E402
from oracle import oracle as O  # noqa: E402
from test_gpu_parity import _two_view_scene  # noqa: E402

rng = np.random.default_rng(0x5AC)
n, n_hyp, thr = 1000, 10000, 1e-7
a, b = _two_view_scene(rng, n, 0.3)
samples = np.stack([rng.choice(n, 8, replace=False) for _ in range(n_hyp)]).astype(np.uint32)
cons = EssentialConsensus(n, n_hyp)
cons.model_inliers(a, b, samples, thr)  # warm-up
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    pose, inl, best = cons.model_inliers(a, b, samples, thr)
gpu_s = (time.perf_counter() - t0) / reps
sub = 100
t0 = time.perf_counter()
w = O.essential_batch(a, b, samples[:sub], thr)
cpu_s = time.perf_counter() - t0
g2 = cons.model_inliers(a, b, samples[:sub], thr)
assert g2[2] == w[1] and np.array_equal(g2[1], w[2]) and g2[0].tobytes() == w[0].tobytes()
print(json.dumps({
    "workload": "10k eight-point hypotheses x 4 poses x 1000 matches (30% outliers), thr 1e-7, host buffers in/out",
    "gpu_seconds_per_scene": round(gpu_s, 5), "hypotheses_per_s": round(n_hyp / gpu_s, 1),
    "pose_match_residuals_per_s": round(n_hyp * 4 * n / gpu_s, 1), "inliers": int(len(inl)),
    "cpu_oracle": {"hypotheses_per_s": round(sub / cpu_s, 2), "cores": 1, "sample": f"first {sub} hypotheses, {cpu_s:.1f} s"},
    "parity": "best id, pose bits and inlier set identical to the oracle on the sampled hypotheses"}))
