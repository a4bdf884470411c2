#!/usr/bin/env python3
"""BASELINE.json configs[2] in isolation: consecutive-pair symmetric BF 2-NN of ~5k x 486-bit descriptors per
frame, device-resident, nothing else on the GPU.  Prints the HIP-event time of the k-NN launches.
usage: python tools/bench_match.py [--pairs 128] [--kp 5070] [--reps 5]"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cv_amd import build  # noqa: E402
build.build()
from cv_amd import _lib  # noqa: E402
from cv_amd.knn import Matcher, RULE_STRICT  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=128)
ap.add_argument("--kp", type=int, default=5070)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--kernel", default="fp4", help="fp4 (LDS-DMA staging), fp4_regs, int8, valu")
a = ap.parse_args()
L = _lib.lib()
dev = torch.device("cuda", 0)
cap, n = 8192, a.pairs
g = torch.Generator(device=dev).manual_seed(7)
descs = torch.randint(0, 256, (n, cap, 64), generator=g, device=dev, dtype=torch.uint8)
descs[:, :, 60] &= 0x3F          # 486 bits: the tail of the 512 is zero in every descriptor
descs[:, :, 61:] = 0
counts = torch.randint(a.kp - 200, a.kp + 200, (n,), generator=g, device=dev, dtype=torch.int32)
pairs = torch.zeros((n, cap, 2), dtype=torch.int32, device=dev)
npairs = torch.zeros((n,), dtype=torch.int32, device=dev)
m = Matcher(cap, kernel=a.kernel)
ia = (C.c_uint32 * n)(*range(n))
ib = (C.c_uint32 * n)(*[(j - 1) % n for j in range(n)])


def run():
    _lib.check(L.hm_match_batch_device(m.handle, descs.data_ptr(), counts.data_ptr(), descs.data_ptr(), counts.data_ptr(),
                                       cap, ia, ib, n, RULE_STRICT, 24, 0.0, 1, pairs.data_ptr(), npairs.data_ptr(), None),
               "match")


run()
_lib.check(L.hm_sync(m.handle), "sync")
_lib.check(L.hm_timing_enable(m.handle, 1), "timing")
for _ in range(a.reps):
    run()
    _lib.check(L.hm_sync(m.handle), "sync")
ms, launches = C.c_double(), C.c_uint64()
_lib.check(L.hm_timing_get(m.handle, C.byref(ms), C.byref(launches), 1), "timing")
kp = counts.float().mean().item()
per = ms.value / launches.value
ops = 2.0 * 2.0 * n * kp * kp * 512.0
print(json.dumps({"kernel": a.kernel, "pairs": n, "mean_kp": round(kp, 1), "knn_ms_per_launch": round(per, 4),
                  "tops": round(ops / (per * 1e-3) / 1e12, 1), "checksum": int(npairs.sum().item())}))
