import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver via gpurun)")


@pytest.fixture(scope="session")
def kitti():
    z = np.load(os.path.join(GOLDEN, "kitti_pair.npz"))
    return z["frame0"], z["frame14"]


@pytest.fixture(scope="session")
def kitti_golden():
    return np.load(os.path.join(GOLDEN, "kitti_oracle_golden.npz"))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


def synth_frame(w, h, seed, n_rect=60, n_disc=60):
    """Small deterministic synthetic frame in the style of SURVEY.md §8d config 2 (value noise +
    rectangles + discs + +-2 noise), uint8."""
    rng = np.random.default_rng(seed)
    img = np.full((h, w), 96.0, np.float32)
    for cell, amp in ((64, 48), (32, 24), (16, 12), (8, 6)):
        gh, gw = h // cell + 2, w // cell + 2
        g = rng.uniform(-amp, amp, (gh, gw)).astype(np.float32)
        ys = np.arange(h, dtype=np.float32) / cell
        xs = np.arange(w, dtype=np.float32) / cell
        y0 = ys.astype(int); x0 = xs.astype(int)
        fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
        img += ((1 - fy) * (1 - fx) * g[y0][:, x0] + (1 - fy) * fx * g[y0][:, x0 + 1]
                + fy * (1 - fx) * g[y0 + 1][:, x0] + fy * fx * g[y0 + 1][:, x0 + 1])
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(n_rect):
        sw, sh = rng.integers(8, 97, 2)
        x, y = rng.integers(0, w), rng.integers(0, h)
        img[y:y + sh, x:x + sw] = rng.integers(0, 256)
    for _ in range(n_disc):
        r = rng.integers(4, 49)
        x, y = rng.integers(0, w), rng.integers(0, h)
        y0, y1, x0, x1 = max(0, y - r), min(h, y + r + 1), max(0, x - r), min(w, x + r + 1)
        m = (yy[y0:y1, x0:x1] - y) ** 2 + (xx[y0:y1, x0:x1] - x) ** 2 <= r * r
        img[y0:y1, x0:x1][m] = rng.integers(0, 256)
    img += rng.integers(-2, 3, (h, w))
    return np.clip(img, 0, 255).astype(np.uint8)
