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


def make_case(i, seed):
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
    return w, h, thr, img


def oracle_case(args):
    i, seed = args
    from oracle import oracle as O
    w, h, thr, img = make_case(i, seed)
    kp, d = O.Akaze(w, h, O.default_config(threshold=thr)).extract(img)
    return i, kp.tobytes(), d.tobytes(), len(d)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=48)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--procs", type=int, default=32)
    ap.add_argument("--opt", action="append", default=[], help="akz_options field, key=value (make_options keywords)")
    a = ap.parse_args()
    kw = {k: int(v) for k, v in (kv.split("=") for kv in a.opt)}
    from oracle import oracle as O
    O.build()
    ctx = mp.get_context("spawn")
    with ctx.Pool(a.procs) as pool:
        want = {r[0]: r[1:] for r in pool.map(oracle_case, [(i, a.seed) for i in range(a.n)], chunksize=1)}
    from cv_amd import build
    build.build()
    from cv_amd import _lib, akaze
    bad = 0
    total = 0
    for i in range(a.n):
        w, h, thr, img = make_case(i, a.seed)
        c = akaze.Context(akaze.Akaze.new(thr), w, h, 1, _lib.make_options(**kw) if kw else None)
        (kp, d), = c.extract_batch([img])
        c.close()
        okp, od, n = want[i]
        total += n
        if kp.tobytes() != okp or d.tobytes() != od:
            bad += 1
            print(f"MISMATCH case {i}: {w}x{h} thr {thr}: gpu {len(d)} vs oracle {n} keypoints")
    print(f"{a.n} cases, {total} keypoints, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
