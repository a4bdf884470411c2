#!/usr/bin/env python3
"""Phase timeline of k_front_fed from an experiment build (tools/variants/experiment_knobs.patch applied, -DAKZ_FF_PROF, tools/build_variant.sh): where a block's
lifetime goes, and how many blocks of a CU are in the same phase at the same time.
usage (GPU box): cp gpurun_variants/prof/libakz.so cv_amd/lib/libakz.so; python tools/ff_prof.py [frames]"""
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
L.akz_debug_ff_prof.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int32]
L.akz_debug_ff_prof.restype = C.c_int32
MB = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
frames = bench.make_frames(torch, dev, 0, MB, 1)
ak = Akaze.default()
ak.max_keypoints = bench.CAP
ctx = ak.context(bench.W, bench.H, MB, options=_lib.make_options(pipeline=False))
NS = 12
for rep in range(2):
    n = C.c_uint32(0)
    L.akz_debug_ff_prof(None, 0, C.byref(n), 1)
    _lib.check(L.akz_scale_space_device(ctx.handle, frames.data_ptr(), 0, MB, bench.W, bench.H, None), "ss")
    _lib.check(L.akz_sync(ctx.handle), "sync")
cap = 1 << 16
buf = np.zeros((cap, NS), dtype=np.uint64)
n = C.c_uint32(0)
L.akz_debug_ff_prof(buf.ctypes.data, cap, C.byref(n), 0)
d = buf[: n.value].astype(np.int64)
print("blocks recorded:", n.value)
# three launches per call (levels 1..3): split by start time gaps
t0 = d[:, 0]
order = np.argsort(t0)
d = d[order]
t0 = d[:, 0]
gaps = np.where(np.diff(t0) > 20000)[0]
bounds = [0] + list(gaps + 1) + [len(d)]
names = ["load+lds", "H pass(+ring)", "Hx+V pass", "g write/fix", "conductivity", "derivs+st", "barrier", "FED", "store"]
CLK = 100e6   # s_memtime counts the 100 MHz reference clock on gfx9 parts
for li in range(len(bounds) - 1):
    seg = d[bounds[li]:bounds[li + 1]]
    if len(seg) < 100:
        continue
    dur = np.diff(seg[:, :10], axis=1).astype(np.float64)
    tot = (seg[:, 9] - seg[:, 0]).astype(np.float64)
    print(f"launch {li}: {len(seg)} blocks sampled, span {(seg[:, 9].max() - seg[:, 0].min()) / CLK * 1e6:.0f} us, "
          f"block lifetime mean {tot.mean() / CLK * 1e6:.2f} us (p10 {np.percentile(tot, 10) / CLK * 1e6:.2f}, p90 {np.percentile(tot, 90) / CLK * 1e6:.2f})")
    for i, nm in enumerate(names):
        print(f"   {nm:16s} {dur[:, i].mean() / CLK * 1e6:7.2f} us  {100 * dur[:, i].mean() / tot.mean():5.1f} %")
    hw = seg[:, 11]
    cu_key = ((hw >> 32) & 0xF) * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 0xF)
    print("   distinct CUs seen:", len(np.unique(cu_key)))
