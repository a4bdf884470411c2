#!/usr/bin/env python3
"""Phase timeline of k_orient_describe from an experiment build (-DAKZ_OD_PROF, tools/build_variant.sh): where a wave's
lifetime goes.
usage (GPU box): cp gpurun_variants/odprof/libakz.so cv_amd/lib/libakz.so; python tools/od_prof.py [frames]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from cv_amd import _lib  # noqa: E402
from cv_amd.akaze import Akaze  # noqa: E402

L = _lib.lib()
L.akz_debug_od_prof.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int32]
L.akz_debug_od_prof.restype = C.c_int32
MB = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
frames = bench.make_frames(torch, dev, 0, MB, 1)
ak = Akaze.default()
ak.max_keypoints = bench.CAP
ctx = ak.context(bench.W, bench.H, MB, options=_lib.make_options(pipeline=False))
d_kps = torch.zeros((MB, bench.CAP, 28), dtype=torch.uint8, device=dev)
d_desc = torch.zeros((MB, bench.CAP, 64), dtype=torch.uint8, device=dev)
d_n = torch.zeros((MB,), dtype=torch.int32, device=dev)
NS = 8
import time
for rep in range(3):
    n = C.c_uint32(0)
    L.akz_debug_od_prof(None, 0, C.byref(n), 1)
    _lib.check(L.akz_extract_batch_device(ctx.handle, frames.data_ptr(), 0, MB, bench.W, bench.H, d_kps.data_ptr(), d_desc.data_ptr(),
                                          bench.CAP, d_n.data_ptr(), None), "extract")
    _lib.check(L.akz_sync(ctx.handle), "sync")
torch.cuda.synchronize()
t0 = time.perf_counter()
_lib.check(L.akz_extract_batch_device(ctx.handle, frames.data_ptr(), 0, MB, bench.W, bench.H, d_kps.data_ptr(), d_desc.data_ptr(),
                                      bench.CAP, d_n.data_ptr(), None), "extract")
_lib.check(L.akz_sync(ctx.handle), "sync")
print(f"one extraction of {MB} frames: {(time.perf_counter() - t0) * 1e3:.2f} ms")
n = C.c_uint32(0)
L.akz_debug_od_prof(None, 0, C.byref(n), 1)
_lib.check(L.akz_extract_batch_device(ctx.handle, frames.data_ptr(), 0, MB, bench.W, bench.H, d_kps.data_ptr(), d_desc.data_ptr(),
                                      bench.CAP, d_n.data_ptr(), None), "extract")
_lib.check(L.akz_sync(ctx.handle), "sync")
cap = 1 << 17
buf = np.zeros((cap, NS), dtype=np.uint64)
n = C.c_uint32(0)
L.akz_debug_od_prof(buf.ctypes.data, cap, C.byref(n), 0)
d = buf[: n.value].astype(np.int64)
print("waves recorded:", n.value, "keypoints:", int(d_n.sum().item()))
names = ["kp record + level", "orientation samples + masks", "window sums", "max, angle, cos/sin", "lattice gather + LDS", "cell sums", "bits + store"]
CLK = 100e6   # s_memtime counts the 100 MHz reference clock on gfx9 parts
dur = np.diff(d, axis=1).astype(np.float64)
tot = (d[:, NS - 1] - d[:, 0]).astype(np.float64)
span = (d[:, NS - 1].max() - d[:, 0].min()) / CLK * 1e6
print(f"span of the sampled waves {span:.0f} us; wave lifetime mean {tot.mean() / CLK * 1e6:.2f} us "
      f"(p10 {np.percentile(tot, 10) / CLK * 1e6:.2f}, p90 {np.percentile(tot, 90) / CLK * 1e6:.2f})")
for i, nm in enumerate(names):
    print(f"   {nm:30s} {dur[:, i].mean() / CLK * 1e6:7.2f} us  {100 * dur[:, i].mean() / tot.mean():5.1f} %")
