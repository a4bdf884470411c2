#!/usr/bin/env python3
"""Build profiles/rNN_pmc_traffic.json from the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of
`bench.py --frames MB --micro-batch MB --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-pipeline`).
HBM bytes = WRITE_SIZE + 2 x FETCH_SIZE (KiB units; FETCH_SIZE doubled: the gfx950 correction for 16-byte reads of
MI355X_MICROARCH.md).  bench.py attaches these per-launch figures as roofline.traffic only when its micro-batch
equals the one recorded here.
usage: pmc_traffic.py fetch.db write.db out.json micro_batch [valu.db frames steps]"""
import json
import re
import sqlite3
import sys


def family_key(name):
    """rocprof kernel name -> the family key bench.py's KERNEL_FAMILIES use (template arguments that do not
    change the algorithm are folded)."""
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    n = re.sub(r"^void\s+", "", n)
    n = re.sub(r"\(.*$", "", n)
    m = re.match(r"k_level_front2<(\d+), (\d+), \d+, \d+, ([a-z ]+),", n)
    if m:
        src = {"unsigned char": ",u8", "unsigned short": ",u16"}.get(m.group(3), "") if m.group(1) == "4" else ""
        return f"k_level_front2<{m.group(1)},{m.group(2)},..{src}>"
    m = re.match(r"k_front_fed<(\d+), (\d+),", n)
    if m:
        # <sigma, halo patches, ..>: the first octave runs one-patch halos at sigma 3 / 4; everything with a two-patch halo, and
        # sigma 2 (the first sublevel of every deeper octave), is the "below the first octave" family of bench.py
        deep = m.group(2) != "1" or m.group(1) == "2"
        return f"k_front_fed<{m.group(1)},2,..>" if deep else f"k_front_fed<{m.group(1)},..>"
    m = re.match(r"k_det_stream<(\d+),", n)
    if m:
        return f"k_det_stream<{m.group(1)},..>"
    m = re.match(r"k_deriv_second_cand2<(\d+),", n)
    if m:
        return f"k_deriv_second_cand2<{m.group(1)},..>"
    m = re.match(r"k_fed_pair<(\d+)>", n)
    if m:
        return f"k_fed_pair<{m.group(1)}>"
    if n.startswith("k_contrast_pair<"):
        return "k_contrast_pair"
    return n


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)).fetchall()
    agg = {}
    for name, val in rows:
        a = agg.setdefault(family_key(name), [0, 0.0])
        a[0] += 1
        a[1] += float(val)
    return agg


SCALE_SPACE = ("k_level_front", "k_front_fed", "k_fed_", "k_det_stream", "k_deriv_", "k_contrast", "k_half_size", "k_cand_", "k_blur",
               "k_filter1d", "k_to_f32")
MATCHER = ("k_knn", "k_expand", "k_pairs")


def stage_of(name):
    """which stage of the timed pipeline a kernel belongs to (None: not the library's — frame generation, torch fills)"""
    if not name.startswith("k_"):
        return None
    if name.startswith(SCALE_SPACE):
        return "scale_space"
    if name.startswith(MATCHER):
        return "matcher"
    return "keypoint_stage"


def main(fetch_db, write_db, out, mb, valu_db=None, frames=None, steps=1):
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    v = per_kernel(valu_db, "SQ_INSTS_VALU") if valu_db else {}
    frames = int(frames or mb)
    res = {"micro_batch": int(mb), "frames_per_step": frames, "steps": int(steps),
           "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_INSTS_VALU SQ_WAVES (separate passes) of "
                     f"bench.py --pmc-run --frames {frames} --micro-batch {mb} --steps {steps} --warmup 0: exactly {steps} step(s) "
                     f"of the timed pipeline and nothing else from the library on the GPU; KiB units; hbm bytes = "
                     "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 16-byte-read correction, MI355X_MICROARCH.md)",
           "source_short": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --pmc-run (library kernels only)",
           "kernels": {}}
    tot = {"scale_space": 0.0, "keypoint_stage": 0.0, "matcher": 0.0}
    for k in sorted(f, key=lambda k: -f[k][1]):
        n = f[k][0]
        wk = w.get(k, [0, 0.0])[1]
        res["kernels"][k] = {"launches": n, "fetch_size_kib_x2": round(2 * f[k][1], 1), "write_size_kib": round(wk, 1),
                             "hbm_bytes_per_launch": round((2 * f[k][1] + wk) * 1024 / max(1, n))}
        if k in v and v[k][0]:
            res["kernels"][k]["valu_insts_per_launch"] = round(v[k][1] / v[k][0])
        st = stage_of(k)
        if st:
            tot[st] += (2 * f[k][1] + wk) * 1024
    nfr = float(frames) * float(steps)
    res["per_frame"] = {"hbm_bytes": sum(tot.values()) / nfr, "scale_space_hbm_bytes": tot["scale_space"] / nfr,
                        "keypoint_stage_hbm_bytes": tot["keypoint_stage"] / nfr, "matcher_hbm_bytes": tot["matcher"] / nfr,
                        "note": "sum over the library's kernels (names k_*) of one pipeline step / frames; torch's frame-generation "
                                "and fill kernels are excluded"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v_["hbm_bytes_per_launch"] for k, v_ in list(res["kernels"].items())[:8]}))
    print(json.dumps(res["per_frame"]))


if __name__ == "__main__":
    main(*sys.argv[1:])
