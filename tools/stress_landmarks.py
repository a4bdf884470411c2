#!/usr/bin/env python3
"""Randomised sweep (GPU box) of hm_landmark_matches_batch_device against oracle/match_oracle.c (orc_landmark_matches): random
decisions, merge verdicts, landmark collisions (within and across the first / second landmarks), absent landmarks, keys beyond
the table, untriangulated rows, empty and full frames; with and without a merge mask.
usage: python tools/stress_landmarks.py [--rounds 40] [--seed 1]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cv_amd import build  # noqa: E402
build.build()
from cv_amd import _lib  # noqa: E402
from cv_amd.knn import Matcher  # noqa: E402
from oracle import oracle as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=40)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
L = _lib.lib()
dev = torch.device("cuda", 0)
bad = frames = merged = 0
for r in range(args.rounds):
    rng = np.random.default_rng(args.seed * 104729 + r)
    cap = int(rng.choice([64, 257, 1024, 4096, 8192]))
    F = int(rng.integers(1, 9))
    n_world = int(rng.integers(8, 4 * cap))
    nq = rng.integers(0, cap + 1, F).astype(np.int32)
    nq[rng.integers(0, F)] = cap
    pool = int(rng.integers(4, n_world + 64))                     # small pools: heavy duplication
    best = np.zeros((F, cap, 3, 2), np.uint32)
    best[..., 0] = rng.integers(0, pool, (F, cap, 3))
    same = best[:, :, 0, 0] == best[:, :, 1, 0]                   # a feature's landmarks are distinct after its own dedup
    best[:, :, 1, 0] = np.where(same, best[:, :, 1, 0] + 1, best[:, :, 1, 0])
    best[..., 1] = rng.integers(0, 400, (F, cap, 3))
    best[rng.random((F, cap)) < 0.01, 0, 0] = 0xFFFFFFFF
    best[rng.random((F, cap)) < 0.01, 1, 0] = 0xFFFFFFFF
    dec = rng.integers(0, 3, (F, cap)).astype(np.uint32)
    ok = (rng.random((F, cap)) < rng.random()).astype(np.uint8)
    world = rng.standard_normal((n_world + F * cap, 4))
    world[:, 3] = np.abs(world[:, 3])
    world[rng.random(len(world)) < 0.15, 3] = -1.0
    d_best = torch.from_numpy(best.view(np.int32)).to(dev); d_dec = torch.from_numpy(dec.view(np.int32)).to(dev)
    d_ok = torch.from_numpy(ok).to(dev); d_nq = torch.from_numpy(nq).to(dev); d_world = torch.from_numpy(world).to(dev)
    m = Matcher(cap)
    iq = np.arange(F, dtype=np.uint32)
    for mask in (d_ok, None):
        d_pairs = torch.full((F, cap, 2), -1, dtype=torch.int32, device=dev); d_np = torch.full((F,), 77, dtype=torch.int32, device=dev)
        _lib.check(L.hm_landmark_matches_batch_device(m.handle, d_best.data_ptr(), d_dec.data_ptr(), None if mask is None else mask.data_ptr(),
                                                      d_nq.data_ptr(), iq.ctypes.data_as(C.c_void_p), cap, F, d_world.data_ptr(), n_world,
                                                      d_pairs.data_ptr(), d_np.data_ptr(), None), "landmark_matches")
        _lib.check(L.hm_sync(m.handle), "hm_sync")
        gp = d_pairs.cpu().numpy().view(np.uint32); gn = d_np.cpu().numpy()
        for f in range(F):
            n = int(nq[f])
            want = O.landmark_pairs(best[f, :n], dec[f, :n], world, merge_ok=None if mask is None else ok[f, :n], n_world=n_world,
                                    merged_base=n_world + f * cap)
            frames += 1
            merged += int((want[:, 1] >= n_world).sum()) if len(want) else 0
            if gn[f] != len(want) or not np.array_equal(gp[f, :gn[f]], want) or not (gp[f, gn[f]:] == 0xFFFFFFFF).all():
                bad += 1
                print(f"round {r} frame {f} (cap {cap}, mask {mask is not None}): {gn[f]} vs {len(want)}")
    m.close()
print(f"stress_landmarks seed {args.seed}: {frames} frame lists in {args.rounds} rounds, {merged} merged matches, {bad} mismatches")
sys.exit(1 if bad else 0)
