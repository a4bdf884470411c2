#!/usr/bin/env python3
"""Scale space alone (akz_scale_space_device, 128 synthetic 1080p frames), best of several runs: frames/s.
For quick A/B runs of scale-space changes."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from cv_amd import build  # noqa: E402
build.build()
from cv_amd import _lib  # noqa: E402
from cv_amd.akaze import Akaze  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda", 0)
MB = 128
frames = bench.make_frames(torch, dev, 0, MB, 1)
ak = Akaze.default()
ak.max_keypoints = bench.CAP
kw = {}
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    kw[k] = bool(int(v)) if k in ("det_side_stream", "stream_kernels", "frame_pairs") else int(v)
ctx = ak.context(bench.W, bench.H, MB, options=_lib.make_options(**kw))
best = 1e9
for rep in range(6):
    torch.cuda.synchronize()
    t = time.perf_counter()
    _lib.check(L.akz_scale_space_device(ctx.handle, frames.data_ptr(), 0, MB, bench.W, bench.H, None), "ss")
    _lib.check(L.akz_sync(ctx.handle), "sync")
    dt = time.perf_counter() - t
    if rep:
        best = min(best, dt)
print(f"scale space {kw}: {MB / best:.1f} frames/s ({best * 1e3:.2f} ms per {MB} frames)")
