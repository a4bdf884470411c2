#!/usr/bin/env python3
"""Development aid: run the HIP path and the oracle on one frame and print, for every pyramid buffer and
keypoint stage, how many elements differ (instead of stopping at the first mismatch)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cv_amd import build  # noqa: E402
build.build()
from cv_amd import _lib, akaze, knn  # noqa: E402
from oracle import oracle as O  # noqa: E402
from conftest import synth_frame  # noqa: E402


def diff(a, b):
    a = np.asarray(a); b = np.asarray(b)
    if a.shape != b.shape:
        return f"SHAPE {a.shape} vs {b.shape}"
    bad = a.view(np.uint32) != b.view(np.uint32) if a.dtype == np.float32 else a != b
    n = int(bad.sum())
    if n == 0:
        return "ok"
    idx = np.argwhere(bad)[0]
    return f"{n}/{a.size} differ, first {tuple(idx)}: {a[tuple(idx)]!r} vs {b[tuple(idx)]!r} maxabs={np.nanmax(np.abs(a.astype(np.float64)-b.astype(np.float64))):.3g}"


def run(img, thr, name):
    h, w = img.shape
    print(f"=== {name} {w}x{h} thr={thr}")
    ak = akaze.Akaze.new(thr)
    ctx = ak.context(w, h, 1, options=_lib.make_options(keep_all=True))
    t = time.time()
    (kp, desc), = ctx.extract_batch([img])
    print(f"gpu extract {time.time()-t:.3f}s n={len(kp)}")
    orc = O.Akaze(w, h, O.default_config(threshold=thr))
    t = time.time()
    okp, odesc = orc.extract(img)
    print(f"oracle extract {time.time()-t:.3f}s n={len(okp)} candidates={orc.num_candidates}")
    print("contrast", ctx.contrast(0), orc.contrast)
    for lvl in range(orc.num_levels):
        for nm in ("Lt", "Lsmooth", "Lflow", "Lx", "Ly", "Ldet"):
            if lvl == 0 and nm == "Lflow":
                continue
            r = diff(ctx.level_buffer(0, lvl, nm, w, h), orc.buffer(lvl, nm))
            if r != "ok" or nm == "Ldet":
                print(f"  L{lvl:02d} {nm:8s} {r}")
    for stage in (0, 1, 2):
        g, o = ctx.keypoints(0, stage), orc.keypoints(stage)
        print(f"  stage{stage}: gpu {len(g)} oracle {len(o)}", end="")
        if len(g) == len(o):
            print(" ", {f: diff(g[f], o[f]) for f in g.dtype.names if diff(g[f], o[f]) != "ok"})
        else:
            print()
    print("  final:", len(kp), len(okp), diff(desc, odesc) if len(kp) == len(okp) else "count differs")
    return kp, desc


if __name__ == "__main__":
    z = np.load(os.path.join(ROOT, "tests", "golden", "kitti_pair.npz"))
    kp0, d0 = run(z["frame0"], 0.01, "kitti0")
    kp1, d1 = run(z["frame14"], 0.01, "kitti14")
    print("lowe matches", len(knn.match_descriptors(d0, d1, 0.5)))
    run(synth_frame(333, 251, 7, 25, 25), 0.001, "synth-odd")
    run(synth_frame(1920, 1080, 4242, 200, 200), 0.001, "synth-1080p")
