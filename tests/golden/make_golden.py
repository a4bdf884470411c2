#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.  Run in the build container (needs /root/reference + Pillow).

1. kitti_pair.npz  — the reference's own two fixture frames (res/0000000000.png, res/0000000014.png:
   KITTI 2011_09_26_drive_0035 camera 0, 1392x512 Luma8; provenance in the reference's res/source.txt),
   decoded to raw uint8 so the GPU box (which has no /root/reference) can run the reference's
   known-answer tests (akaze/tests/estimate_pose.rs:41,42,59).  PNG decoding is lossless.
2. kitti_oracle_golden.npz — outputs of OUR CPU oracle on those frames (keypoints, descriptor bytes,
   match indices) for Akaze::sparse() and Akaze::default().  The reference publishes only COUNTS
   (399/343/11); these arrays are oracle-generated regression vectors, not reference outputs.
"""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import oracle as O  # noqa: E402

REF = "/root/reference/res"


def main():
    a = np.array(Image.open(os.path.join(REF, "0000000000.png")))
    b = np.array(Image.open(os.path.join(REF, "0000000014.png")))
    assert a.dtype == np.uint8 and a.shape == (512, 1392) and b.shape == a.shape
    np.savez_compressed(os.path.join(HERE, "kitti_pair.npz"), frame0=a, frame14=b)
    out = {}
    for name, thr in (("sparse", 0.01), ("default", 0.001)):
        ak = O.Akaze(a.shape[1], a.shape[0], O.default_config(threshold=thr))
        ka, da = ak.extract(a)
        kb, db = ak.extract(b)
        out[f"{name}_kp0"], out[f"{name}_desc0"] = ka, da
        out[f"{name}_kp14"], out[f"{name}_desc14"] = kb, db
        out[f"{name}_lowe"] = O.match(da, db, rule=O.RULE_LOWE, param_f=0.5, symmetric=False)
        out[f"{name}_sym24"] = O.match(da, db, rule=O.RULE_STRICT, param_u=24, symmetric=True)
        out[f"{name}_knn2"] = O.knn2(da, db)
        print(name, len(da), len(db), len(out[f"{name}_lowe"]), len(out[f"{name}_sym24"]))
    np.savez_compressed(os.path.join(HERE, "kitti_oracle_golden.npz"), **out)


if __name__ == "__main__":
    main()
