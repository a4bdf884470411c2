#!/usr/bin/env python3
"""Dump keypoints and descriptors in exactly the format of the reference's akaze/examples/akaze.rs:11-33
(`<stem>_kps.csv`: "x, y, angle, size, octave, class_id" with Rust's f32 Display; `<stem>_descs.txt`: the 64
descriptor bytes as 8-bit binary groups joined by '_'), so that anyone with a Rust toolchain can diff our
output against `cargo run --example akaze -- image.png` byte for byte.

usage: akaze_dump.py [--oracle] image.png|image.npy ...     (--oracle: CPU oracle instead of the MI355X path)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rust_f32(x):
    """Rust `{}` for f32: shortest digits that round-trip, never scientific, no trailing '.0'."""
    return np.format_float_positional(np.float32(x), unique=True, trim="-")


def load(path):
    if path.endswith(".npy"):
        return np.load(path)
    from PIL import Image
    im = Image.open(path)
    return np.asarray(im if im.mode in ("L", "I;16") else im.convert("L"))


def main(argv):
    use_oracle = "--oracle" in argv
    paths = [a for a in argv if not a.startswith("--")]
    for path in paths:
        img = load(path)
        if use_oracle:
            from oracle import oracle as O
            kps, descs = O.Akaze(img.shape[1], img.shape[0], O.default_config()).extract(img)
        else:
            from cv_amd.akaze import Akaze
            kps, descs = Akaze.default().extract_arrays(img)
        stem = os.path.splitext(os.path.basename(path))[0]
        with open(stem + "_kps.csv", "w") as f:
            for k in kps:
                f.write(f"{rust_f32(k['x'])}, {rust_f32(k['y'])}, {rust_f32(k['angle'])}, {rust_f32(k['size'])}, "
                        f"{int(k['octave'])}, {int(k['class_id'])}\n")
        with open(stem + "_descs.txt", "w") as f:
            for d in descs:
                f.write("_".join(format(int(b), "08b") for b in d) + "\n")
        print(f"{path}: {len(kps)} keypoints -> {stem}_kps.csv, {stem}_descs.txt")


if __name__ == "__main__":
    main(sys.argv[1:])
