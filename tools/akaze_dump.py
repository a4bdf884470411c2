#!/usr/bin/env python3
"""Dump keypoints and descriptors in exactly the format of the reference's akaze/examples/akaze.rs:11-33
(`<stem>_kps.csv`: "x, y, angle, size, octave, class_id" with Rust's f32 Display; `<stem>_descs.txt`: the 64
descriptor bytes as 8-bit binary groups joined by '_'), so that anyone with a Rust toolchain can diff our
output against `cargo run --example akaze -- image.png` byte for byte.

usage: akaze_dump.py [--oracle] [--arith K|all] [--trig portable|libm] [--out-dir DIR] image.png|image.npy ...

  --oracle       the CPU oracle (oracle/akaze_oracle.c) instead of the MI355X path
  --arith K      which of the reference's three UN-VENDORED arithmetic orders the filters and half_size use
                 (akaze/src/image.rs:160-195, 242-247, 320-325 through `wide` / `ndarray`):
                 bit 0 reduce_add = (a0+a1)+(a2+a3), bit 1 fused mul_add, bit 2 2x2 sum ((a+b)+c)+d; default 0.
                 The GPU path runs akz_options.arith = K, the oracle ORC_OPT_{REDUCE, FMA, HALFSUM}.
                 `all` writes the eight variants as <stem>_a<K>_kps.csv / <stem>_a<K>_descs.txt
  --trig         oracle only: `portable` = include/akz_portable_math.h (what the kernels evaluate), `libm` = the host's
                 atan2f / sinf / cosf (what a Rust build on this machine calls)

tools/pin_arith.py is the comparator: it takes the two files a cargo build wrote and says which combination they are.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ORC_OPT_REDUCE, ORC_OPT_FMA, ORC_OPT_HALFSUM, ORC_OPT_TRIG = 0, 1, 2, 3


def rust_f32(x):
    """Rust `{}` for f32: shortest digits that round-trip, never scientific, no trailing '.0'."""
    return np.format_float_positional(np.float32(x), unique=True, trim="-")


def load(path):
    """The image as GrayFloatImage::from_dynamic would see it (image.rs:45-109): Luma8 / Luma16 as they are; colour images
    through DynamicImage::grayscale() as oracle/color_oracle.c restates it (NOT Pillow's convert('L'), whose weights differ)."""
    if path.endswith(".npy"):
        return np.load(path)
    from PIL import Image
    im = Image.open(path)
    if im.mode in ("L", "I;16"):
        return np.asarray(im)
    if im.mode in ("RGB", "RGBA"):
        from oracle import oracle as O
        return O.luma(np.asarray(im))
    return np.asarray(im.convert("L"))


def kps_text(kps):
    return "".join(f"{rust_f32(k['x'])}, {rust_f32(k['y'])}, {rust_f32(k['angle'])}, {rust_f32(k['size'])}, "
                   f"{int(k['octave'])}, {int(k['class_id'])}\n" for k in kps)


def descs_text(descs):
    return "".join("_".join(format(int(b), "08b") for b in d) + "\n" for d in descs)


def oracle_extract(img, arith=0, trig="portable", cfg=None):
    """Keypoints and descriptors of the CPU oracle under one combination of the un-vendored orders."""
    from oracle import oracle as O
    O.build()
    saved = [O.lib().orc_get_option(i) for i in range(4)]
    try:
        O.set_option(ORC_OPT_REDUCE, arith & 1)
        O.set_option(ORC_OPT_FMA, (arith >> 1) & 1)
        O.set_option(ORC_OPT_HALFSUM, (arith >> 2) & 1)
        O.set_option(ORC_OPT_TRIG, 1 if trig == "libm" else 0)
        return O.Akaze(img.shape[1], img.shape[0], cfg if cfg is not None else O.default_config()).extract(img)
    finally:
        for i, v in enumerate(saved):
            O.set_option(i, v)


def gpu_extract(img, arith=0):
    from cv_amd import _lib
    from cv_amd.akaze import Akaze
    ak = Akaze.default()
    if arith == 0:
        return ak.extract_arrays(img)
    ctx = ak.context(img.shape[1], img.shape[0], 1, options=_lib.make_options(arith=arith))
    return ctx.extract_batch([np.ascontiguousarray(img)])[0]


def main(argv):
    use_oracle = "--oracle" in argv
    arith, trig, out_dir, paths = "0", "portable", ".", []
    it = iter(a for a in argv if a != "--oracle")
    for a in it:
        if a == "--arith":
            arith = next(it)
        elif a == "--trig":
            trig = next(it)
        elif a == "--out-dir":
            out_dir = next(it)
        elif a.startswith("--"):
            raise SystemExit(__doc__)
        else:
            paths.append(a)
    if trig not in ("portable", "libm") or (trig == "libm" and not use_oracle):
        raise SystemExit("--trig libm needs --oracle (the kernels evaluate include/akz_portable_math.h)")
    variants = list(range(8)) if arith == "all" else [int(arith)]
    if not paths or any(k < 0 or k > 7 for k in variants):
        raise SystemExit(__doc__)
    os.makedirs(out_dir, exist_ok=True)
    for path in paths:
        img = load(path)
        stem = os.path.splitext(os.path.basename(path))[0]
        for k in variants:
            kps, descs = oracle_extract(img, k, trig) if use_oracle else gpu_extract(img, k)
            tag = f"_a{k}" if arith == "all" else ""
            kp_path, d_path = os.path.join(out_dir, f"{stem}{tag}_kps.csv"), os.path.join(out_dir, f"{stem}{tag}_descs.txt")
            with open(kp_path, "w") as f:
                f.write(kps_text(kps))
            with open(d_path, "w") as f:
                f.write(descs_text(descs))
            print(f"{path}: arith {k}: {len(kps)} keypoints -> {kp_path}, {d_path}")


if __name__ == "__main__":
    main(sys.argv[1:])
