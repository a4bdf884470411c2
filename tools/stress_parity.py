#!/usr/bin/env python3
"""Randomised parity sweep (GPU box): random frame sizes, contents and thresholds, HIP path vs the CPU oracle —
keypoints and descriptor bytes must be identical.  The oracle runs in a process pool.
usage: python tools/stress_parity.py [--n 48] [--seed 1] [--procs 32]"""
import argparse
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def make_case(i, seed, arms=False):
    from conftest import synth_frame
    rng = np.random.default_rng(seed * 7919 + i)
    w = int(rng.integers(40, 230)) * 4 if i % 3 else int(rng.integers(120, 900))
    h = int(rng.integers(100, 700))
    thr = float(rng.choice([0.01, 0.003, 0.001, 0.0003]))
    kind = i % 4
    if kind == 0:
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    else:
        img = synth_frame(w, h, seed=int(rng.integers(1 << 30)), n_rect=int(rng.integers(5, 60)), n_disc=int(rng.integers(5, 60)))
    if kind == 2:
        img = (img.astype(np.int16) // 16 * 16).astype(np.uint8)      # plateaus: exact ties
    # the input arm (GrayFloatImage::from_dynamic, akaze/src/image.rs:45-109): Luma8 most of the time, else Luma16, f32 gray,
    # RGB8, RGBA16, RGB32F
    arm = int(rng.integers(0, 10)) if arms else 0
    if arm == 5:
        img = (img.astype(np.uint16) * 257 + rng.integers(0, 200, img.shape, dtype=np.uint16) * (img < 255)).astype(np.uint16)
    elif arm == 6:
        img = (img.astype(np.float32) / 255.0 + rng.uniform(-1e-3, 1e-3, img.shape).astype(np.float32)).clip(0, 1).astype(np.float32)
    elif arm == 7:
        img = np.stack([img, np.roll(img, 3, 1), 255 - img], 2)
    elif arm == 8:
        b = img.astype(np.uint16) * 257
        img = np.stack([b, np.roll(b, 5, 0), b // 2, rng.integers(0, 65535, b.shape, dtype=np.uint16)], 2)
    elif arm == 9:
        f = img.astype(np.float32) / 255.0
        img = np.stack([f, f * 0.7, np.roll(f, 2, 1)], 2)
    return w, h, thr, img


def oracle_case(args):
    i, seed, arms, arith = args
    from oracle import oracle as O
    # the three un-vendored arithmetic orders (include/akz.h AKZ_ARITH_*): the oracle's switches follow akz_options.arith
    O.set_option(O.OPT_REDUCE, arith & 1)
    O.set_option(O.OPT_FMA, (arith >> 1) & 1)
    O.set_option(O.OPT_HALFSUM, (arith >> 2) & 1)
    w, h, thr, img = make_case(i, seed, arms)
    if img.ndim == 3:
        img = O.luma(img)                       # DynamicImage::grayscale(), then the gray arm of its type
    kp, d = O.Akaze(w, h, O.default_config(threshold=thr)).extract(img)
    return i, kp.tobytes(), d.tobytes(), len(d)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=48)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--procs", type=int, default=32)
    ap.add_argument("--arms", action="store_true", help="also draw the input arm: Luma16, f32, RGB8, RGBA16, RGB32F")
    ap.add_argument("--opt", action="append", default=[], help="akz_options field, key=value (make_options keywords)")
    ap.add_argument("--copies", type=int, default=1, help="frames per call: the case's frame K times in ONE batch call (call size selects "
                    "kernels: above 4 frames the batch paths run — FED launches of up to 8 steps with the fused first launch, one "
                    "workgroup per frame in the keypoint stage); every copy is compared")
    a = ap.parse_args()
    kw = {k: int(v) for k, v in (kv.split("=") for kv in a.opt)}
    from oracle import oracle as O
    O.build()
    ctx = mp.get_context("spawn")
    with ctx.Pool(a.procs) as pool:
        want = {r[0]: r[1:] for r in pool.map(oracle_case, [(i, a.seed, a.arms, kw.get("arith", 0)) for i in range(a.n)], chunksize=1)}
    from cv_amd import build
    build.build()
    from cv_amd import _lib, akaze
    bad = 0
    total = 0
    for i in range(a.n):
        w, h, thr, img = make_case(i, a.seed, a.arms)
        if img.dtype == np.uint8 and img.ndim == 2:
            ak = akaze.Akaze.new(thr)
            while True:                    # a dense noise frame can pass 16 384 keypoints: more room, as the host mirrors do
                c = akaze.Context(ak, w, h, a.copies, _lib.make_options(**kw) if kw else None)
                try:
                    res = c.extract_batch([img] * a.copies)
                    kp, d = res[0]
                    for q in range(1, a.copies):      # every copy of the call must carry the same bytes
                        if res[q][0].tobytes() != kp.tobytes() or res[q][1].tobytes() != d.tobytes():
                            kp = res[q][0][:0]        # (forces the mismatch report below)
                    break
                except _lib.AkzError as e:
                    if e.status != -7 or ak.max_keypoints >= akaze.MAX_KEYPOINTS:
                        raise
                    ak.max_keypoints = min(akaze.MAX_KEYPOINTS, 2 * int(ak.max_keypoints))
                finally:
                    c.close()
        else:
            kp, d = akaze.Akaze.new(thr).extract_arrays(img)       # the host mirror's dispatch on dtype / channels
        okp, od, n = want[i]
        total += n
        if kp.tobytes() != okp or d.tobytes() != od:
            bad += 1
            print(f"MISMATCH case {i}: {w}x{h} {img.dtype} {img.shape[2:] or ''} thr {thr}: gpu {len(d)} vs oracle {n} keypoints")
    print(f"{a.n} cases, {total} keypoints, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
