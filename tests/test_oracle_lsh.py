"""oracle/lsh_oracle.c (place recognition) against an independent numpy restatement.

The reference holds no test or golden vector for HammingHasher::hash_bag (the crate is not vendored): these
tests pin the oracle to the stated definition only — parity with hamming-lsh itself is unpinned.
"""
import numpy as np


def _dist(a, b):
    return np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(2)


def test_hash_bag_matches_definition(oracle):
    rng = np.random.default_rng(5)
    cw = rng.integers(0, 256, (4096, 64), dtype=np.uint8)
    cw[9] = cw[3]                                   # equal distances -> the lower codeword index
    f = rng.integers(0, 256, (400, 64), dtype=np.uint8)
    f[0] = cw[9]
    f[1] = cw[4095]
    h, words = oracle.hash_bag(f, cw)
    d = _dist(f, cw)
    idx = d.argmin(1)                               # argmin returns the first minimum
    assert (words["index"] == idx).all() and (words["distance"] == d.min(1)).all()
    assert words["index"][0] == 3 and words["distance"][0] == 0 and words["index"][1] == 4095
    bits = np.zeros(4096, np.uint8)
    bits[idx] = 1
    assert (np.packbits(bits, bitorder="little") == h).all()      # bit w at byte w >> 3, position w & 7
    assert h.shape == (512,) and (h[4095 >> 3] >> 7) & 1 == 1


def test_hash_bag_edge_cases(oracle):
    rng = np.random.default_rng(6)
    cw = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    h, words = oracle.hash_bag(np.zeros((0, 64), np.uint8), cw)
    assert h.shape == (8,) and not h.any() and len(words) == 0
    # a bag is a set: order and repetition of the features do not change the hash
    f = rng.integers(0, 256, (50, 64), dtype=np.uint8)
    h1, _ = oracle.hash_bag(f, cw)
    h2, _ = oracle.hash_bag(np.concatenate([f[::-1], f[:7]]), cw)
    assert (h1 == h2).all()
    try:
        oracle.hash_bag(f, cw[:40])
        assert False, "codeword counts that are not a multiple of 32 are rejected"
    except ValueError:
        pass


def test_hash_knn_order(oracle):
    rng = np.random.default_rng(7)
    hs = rng.integers(0, 256, (200, 512), dtype=np.uint8)
    hs[150] = hs[20]
    hs[60] = hs[20]
    for k in (1, 3, 200, 512):
        r = oracle.hash_knn(hs[20], hs, k)
        d = np.unpackbits(hs ^ hs[20], axis=1).sum(1)
        order = np.lexsort((np.arange(200), d))[:k]
        assert len(r) == min(k, 200)
        assert (r["index"] == order).all() and (r["distance"] == d[order]).all()
    assert list(oracle.hash_knn(hs[20], hs, 3)["index"]) == [20, 60, 150]
    assert len(oracle.hash_knn(hs[0], hs[:0], 4)) == 0
