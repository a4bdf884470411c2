#!/usr/bin/env python3
"""BASELINE configs[3] (bench.py: extra_ransac's exhaustive leg) run exactly N times and nothing else: the workload of the
counter / trace passes of tools/pmc_ransac.sh.  usage: ransac_once.py [calls, default 3]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cv_amd import build  # noqa: E402
build.build()
from cv_amd.ransac import EssentialConsensus  # noqa: E402
from test_gpu_parity import _two_view_scene  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rng = np.random.default_rng(0x5AC)
n, n_hyp, thr = 1000, 10000, 1e-7
a, b = _two_view_scene(rng, n, 0.3)
samples = np.stack([rng.choice(n, 8, replace=False) for _ in range(n_hyp)]).astype(np.uint32)
cons = EssentialConsensus(n, n_hyp)
for _ in range(calls):
    pose, inl, best = cons.model_inliers(a, b, samples, thr)
print("calls", calls, "inliers", len(inl), "best", best)
