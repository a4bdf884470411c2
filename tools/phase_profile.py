#!/usr/bin/env python3
"""Serial phase profile: the bench workload's micro-batch run one phase at a time with a device sync in
between (no cross-stream overlap), so `rocprofv3 --kernel-trace --stats` durations are per-kernel costs on an
otherwise idle GPU (the context is created with AKZ_OPT_NO_PIPELINE).
usage: rocprofv3 ... -- python tools/phase_profile.py [--mb 64] [--reps 3]"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=int, default=64)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--opt", action="append", default=[], help="akz_options field for the context, key=value "
                    "(cv_amd._lib.make_options keywords), e.g. --opt stream_prefetch=4 --opt stream_kernels=0")
    a = ap.parse_args()
    from cv_amd import build
    build.build()
    from cv_amd import _lib
    from cv_amd.akaze import Akaze
    from cv_amd.knn import Matcher, RULE_STRICT
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    MB, W, H, CAP = a.mb, bench.W, bench.H, bench.CAP
    frames = bench.make_frames(torch, dev, 0, MB, 1)
    ak = Akaze.default()
    ak.max_keypoints = CAP
    from cv_amd import _lib
    kw = {}
    for kv in a.opt:
        key, val = kv.split("=")
        kw[key] = val if key == "contrast" else int(val)
    for key in _lib.BOOL_OPTIONS:
        if key in kw:
            kw[key] = bool(kw[key])
    ctx = ak.context(W, H, MB, options=_lib.make_options(pipeline=False, **kw))
    matcher = Matcher(CAP, device=0)
    kps = torch.zeros((MB, CAP, 28), dtype=torch.uint8, device=dev)
    descs = torch.zeros((MB, CAP, 64), dtype=torch.uint8, device=dev)
    counts = torch.zeros((MB,), dtype=torch.int32, device=dev)
    pairs = torch.zeros((MB + 2, CAP, 2), dtype=torch.int32, device=dev)
    npairs = torch.zeros((MB + 2,), dtype=torch.int32, device=dev)
    cur = torch.cuda.current_stream()
    js = list(range(MB))
    ia = (C.c_uint32 * MB)(*js)
    ib = (C.c_uint32 * MB)(*[(j - 1) % MB for j in js])

    def sync():
        _lib.check(L.akz_sync(ctx.handle), "akz_sync")
        _lib.check(L.hm_sync(matcher.handle), "hm_sync")
        torch.cuda.synchronize()

    res = {}
    for rep in range(a.reps + 1):
        sync(); t = time.perf_counter()
        _lib.check(L.akz_scale_space_device(ctx.handle, frames.data_ptr(), 0, MB, W, H, None), "ss")
        sync(); t_ss = time.perf_counter() - t
        t = time.perf_counter()
        _lib.check(L.akz_extract_batch_device(ctx.handle, frames.data_ptr(), 0, MB, W, H, kps.data_ptr(),
                                              descs.data_ptr(), CAP, counts.data_ptr(), cur.cuda_stream), "extract")
        sync(); t_ex = time.perf_counter() - t
        t = time.perf_counter()
        _lib.check(L.hm_match_batch_device(matcher.handle, descs.data_ptr(), counts.data_ptr(), descs.data_ptr(),
                                           counts.data_ptr(), CAP, ia, ib, MB, RULE_STRICT, 24, 0.0, 1,
                                           pairs.data_ptr(), npairs.data_ptr(), cur.cuda_stream), "match")
        sync(); t_m = time.perf_counter() - t
        if rep:
            for k, v in (("scale_space", t_ss), ("extract", t_ex), ("match", t_m)):
                res.setdefault(k, []).append(v * 1e3)
    print({k: round(min(v), 3) for k, v in res.items()}, "ms per", MB, "frames; kp/frame",
          counts.float().mean().item())


if __name__ == "__main__":
    main()
