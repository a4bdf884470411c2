#!/usr/bin/env python3
"""Which of the reference's un-vendored arithmetic orders is a given Rust build's?  — the one command that turns
"parity vs oracle/" into "parity vs rust-cv".

The reference's own known answers (399 / 343 descriptors, 11 matches: akaze/tests/estimate_pose.rs:41-42,59) hold for ALL
eight combinations of the three orders that live in crates absent from /root/reference (`wide::f32x4::reduce_add`,
`wide::f32x4::mul_add`, `ndarray`'s 2 x 2 `sum()`: akaze/src/image.rs:160-195, 242-247, 320-325) and for both trigonometry
sources, so they cannot say which combination a cargo build computes; this library carries all eight
(`akz_options.arith`).  Anyone with a Rust toolchain settles it in two commands:

    cargo run --release --example akaze -- res/0000000000.png          # akaze/examples/akaze.rs:11-33 writes
                                                                       #   0000000000_kps.csv, 0000000000_descs.txt
    python3 tools/pin_arith.py 0000000000_kps.csv 0000000000_descs.txt res/0000000000.png

pin_arith.py runs the CPU oracle on the image under all 8 x 2 combinations (arith 0..7 x {portable, libm} trigonometry),
formats each result exactly as the example does (tools/akaze_dump.py) and compares BYTE FOR BYTE.  It prints one line per
combination — MATCH, or the first differing keypoint / descriptor — then the verdict: the `arith` value to create contexts
with (akz_options.arith, Akaze(arith=...) in the bindings), or, when nothing matches, the closest combination and where
it first departs.  Exit code 0 = at least one combination reproduces the Rust files, 1 = none does.

The image is read the way GrayFloatImage::from_dynamic sees it (tools/akaze_dump.py: load); --config default|sparse|dense
selects Akaze::default() (the example's) / sparse() / dense() (akaze/src/lib.rs:147-166).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

KP_FIELDS = ("x", "y", "angle", "size", "octave", "class_id")


def first_difference(want_kps, want_descs, got_kps, got_descs):
    """None when both texts are equal byte for byte, else a short description of the first departure."""
    if want_kps == got_kps and want_descs == got_descs:
        return None
    wk, gk = want_kps.splitlines(), got_kps.splitlines()
    wd, gd = want_descs.splitlines(), got_descs.splitlines()
    for i in range(min(len(wk), len(gk))):
        if wk[i] != gk[i]:
            a, b = [t.strip() for t in wk[i].split(",")], [t.strip() for t in gk[i].split(",")]
            for name, x, y in zip(KP_FIELDS, a, b):
                if x != y:
                    return f"keypoint {i}: {name} {x} (rust) vs {y}"
            return f"keypoint {i}: line differs"
        if i < min(len(wd), len(gd)) and wd[i] != gd[i]:
            bits = sum(1 for x, y in zip(wd[i], gd[i]) if x != y)
            return f"descriptor {i}: {bits} bit(s) differ (keypoints 0..{i} equal)"
    if len(wk) != len(gk):
        return f"{len(wk)} keypoints (rust) vs {len(gk)}; the first {min(len(wk), len(gk))} are equal"
    for i in range(min(len(wd), len(gd))):
        if wd[i] != gd[i]:
            return f"descriptor {i}: differs"
    return f"{len(wd)} descriptors (rust) vs {len(gd)}"


def up_to_trig(want_kps, want_descs, got_kps, got_descs):
    """The weaker verdict for a build whose libm is neither of the two tried here: every field but `angle` equal for every
    keypoint.  Returns None, or (largest angle difference in f32 ulps, descriptors that differ)."""
    import numpy as np
    wk, gk = want_kps.splitlines(), got_kps.splitlines()
    if len(wk) != len(gk):
        return None
    worst = 0
    for a, b in zip(wk, gk):
        fa, fb = [t.strip() for t in a.split(",")], [t.strip() for t in b.split(",")]
        if len(fa) != 6 or len(fb) != 6 or fa[:2] + fa[3:] != fb[:2] + fb[3:]:
            return None
        xa, xb = np.float32(fa[2]), np.float32(fb[2])
        worst = max(worst, abs(int(xa.view(np.int32)) - int(xb.view(np.int32))))
    return worst, sum(1 for a, b in zip(want_descs.splitlines(), got_descs.splitlines()) if a != b)


def equal_prefix(want_kps, got_kps):
    n = 0
    for a, b in zip(want_kps.splitlines(), got_kps.splitlines()):
        if a != b:
            break
        n += 1
    return n


def pin(rust_kps_text, rust_descs_text, img, config="default", out=print):
    """Returns (matches, report): matches = [(arith, trig)] that reproduce the two texts; report = every combination's line."""
    import akaze_dump as D
    from oracle import oracle as O
    O.build()
    cfg = O.default_config(threshold={"default": None, "sparse": 0.01, "dense": 0.0001}[config])
    matches, report, best, near = [], [], (-1, None, None), []
    for trig in ("portable", "libm"):
        for arith in range(8):
            kps, descs = D.oracle_extract(img, arith, trig, cfg)
            got_k, got_d = D.kps_text(kps), D.descs_text(descs)
            diff = first_difference(rust_kps_text, rust_descs_text, got_k, got_d)
            name = (f"arith {arith} (reduce {'pairwise' if arith & 1 else 'sequential'}, mul_add {'fused' if arith & 2 else 'unfused'}, "
                    f"2x2 sum {'sequential' if arith & 4 else 'pairwise'}), trig {trig}")
            if diff is None:
                matches.append((arith, trig))
                line = f"MATCH     {name}: {len(kps)} keypoints, every byte equal"
            else:
                line = f"differs   {name}: {diff}"
                t = up_to_trig(rust_kps_text, rust_descs_text, got_k, got_d)
                if t is not None:
                    near.append((arith, trig, t))
                    line += f"  [all fields but `angle` equal; angles within {t[0]} ulp, {t[1]} descriptors differ]"
                n = equal_prefix(rust_kps_text, got_k)
                if n > best[0]:
                    best = (n, name, diff)
            report.append(line)
            out(line)
    if matches:
        ar = sorted({a for a, _ in matches})
        out(f"VERDICT: the Rust build computes arith = {ar[0] if len(ar) == 1 else ar} "
            f"(trigonometry: {', '.join(sorted({t for _, t in matches}))}) -> create contexts with akz_options.arith = {ar[0]}"
            + ("" if len(ar) == 1 else "; this image does not separate those — try one with more octaves / odd sizes"))
    elif near:
        ar = sorted({a for a, _, _ in near})
        out(f"VERDICT: no byte-for-byte match, but arith = {ar[0] if len(ar) == 1 else ar} reproduces every keypoint's x, y, size, octave, "
            f"class_id — the remaining differences are the angle's last bits (the Rust build's libm is neither this host's nor "
            f"include/akz_portable_math.h) and the descriptor bits that follow from them: the arithmetic order is pinned, the "
            f"trigonometry is not")
    else:
        out(f"VERDICT: no combination reproduces the Rust files; closest: {best[1]} ({best[0]} leading keypoints equal) — {best[2]}")
    return matches, report


def main(argv):
    config = "default"
    args = []
    it = iter(argv)
    for a in it:
        if a == "--config":
            config = next(it)
        elif a.startswith("--"):
            raise SystemExit(__doc__)
        else:
            args.append(a)
    if len(args) != 3 or config not in ("default", "sparse", "dense"):
        raise SystemExit(__doc__)
    import akaze_dump as D
    with open(args[0]) as f:
        rk = f.read()
    with open(args[1]) as f:
        rd = f.read()
    matches, _ = pin(rk, rd, D.load(args[2]), config)
    return 0 if matches else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
