#!/usr/bin/env python3
"""Randomised parity sweep of the matcher (GPU box): random set sizes (2 .. cap, ragged, far from tile multiples), descriptor
populations that force ties (duplicates, near-duplicates, few distinct words), the three kernels (FP4 MFMA, int8 MFMA, VALU),
k = 1..3 neighbours, and the three acceptance rules with and without the symmetric check — hm_knn / hm_match vs
oracle/match_oracle.c: indices, distances and pair lists must be identical.  The oracle runs in a process pool.
usage: python tools/stress_match.py [--n 200] [--seed 1] [--procs 32]"""
import argparse
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_case(i, seed):
    rng = np.random.default_rng(seed * 15485863 + i)
    na = int(rng.choice([2, 3, 31, 32, 33, 255, 257, int(rng.integers(2, 3000))]))
    nb = int(rng.choice([2, 3, 31, 32, 33, 255, 257, int(rng.integers(2, 3000))]))
    kind = i % 4
    if kind == 0:
        a = rng.integers(0, 256, (na, 64), dtype=np.uint8); b = rng.integers(0, 256, (nb, 64), dtype=np.uint8)
    elif kind == 1:                                            # b = noisy copies of a's rows: true matches and near ties
        a = rng.integers(0, 256, (na, 64), dtype=np.uint8)
        b = a[rng.integers(0, na, nb)].copy()
        flips = rng.integers(0, 512, (nb, int(rng.integers(0, 40))))
        for r in range(nb):
            for f in flips[r]:
                b[r, f >> 3] ^= np.uint8(1 << (f & 7))
    elif kind == 2:                                            # few distinct words: masses of exact ties
        words = rng.integers(0, 256, (int(rng.integers(2, 9)), 64), dtype=np.uint8)
        a = words[rng.integers(0, len(words), na)]; b = words[rng.integers(0, len(words), nb)]
    else:                                                      # sparse bits (small distances everywhere)
        a = (rng.random((na, 64, 8)) < 0.05); b = (rng.random((nb, 64, 8)) < 0.05)
        a = np.packbits(a, axis=2, bitorder="little").reshape(na, 64); b = np.packbits(b, axis=2, bitorder="little").reshape(nb, 64)
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    k = int(rng.integers(1, 4))
    rule = int(rng.integers(0, 3))
    param_u = int(rng.choice([0, 1, 24, 60]))
    param_f = float(rng.choice([0.5, 0.8, 0.95]))
    symmetric = bool(rng.integers(2))
    kernel = ["fp4", "int8", "valu"][int(rng.integers(0, 3))]
    return a, b, k, rule, param_u, param_f, symmetric, kernel


def oracle_case(args):
    i, seed = args
    from oracle import oracle as O
    a, b, k, rule, param_u, param_f, symmetric, kernel = make_case(i, seed)
    nn = O.knn(a, b, k)
    pairs = O.match(a, b, rule, param_u, param_f, symmetric)
    return i, nn.tobytes(), pairs.tobytes()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--procs", type=int, default=32)
    a_ = ap.parse_args()
    from oracle import oracle as O
    O.build()
    with mp.get_context("spawn").Pool(a_.procs) as pool:
        want = {r[0]: r[1:] for r in pool.map(oracle_case, [(i, a_.seed) for i in range(a_.n)], chunksize=1)}
    from cv_amd import build
    build.build()
    from cv_amd import knn
    ms = {kern: knn.Matcher(4096, kernel=kern) for kern in ("fp4", "int8", "valu")}
    bad = 0
    for i in range(a_.n):
        a, b, k, rule, param_u, param_f, symmetric, kernel = make_case(i, a_.seed)
        m = ms[kernel]
        kk = 2 if kernel == "valu" else k                       # the VALU kernel exists for k = 2 only
        nn = m.knn(a, b, kk)
        wnn = np.frombuffer(want[i][0], O.NB_DTYPE).reshape(len(a), k) if kk == k else O.knn(a, b, kk)
        absent = wnn["index"] == 0xFFFFFFFF                     # slots past the target count: {2^22 - 1, 1023} on the device
        ok = np.array_equal(nn["index"], np.where(absent, (1 << 22) - 1, wnn["index"])) and \
            np.array_equal(nn["distance"], np.where(absent, 1023, wnn["distance"]))
        pairs = m.match(a, b, rule, param_u, param_f, symmetric)
        ok = ok and pairs.tobytes() == want[i][1]
        if not ok:
            bad += 1
            print(f"MISMATCH case {i}: {len(a)} x {len(b)} k {kk} rule {rule} u {param_u} f {param_f} sym {symmetric} kernel {kernel}", flush=True)
    print(f"stress_match seed {a_.seed}: {a_.n} cases, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
