"""bench.py's byte models and roofline arithmetic: the kernel families the library times and their
algorithmic HBM bytes per unit, the gather model of k_orient_describe, the per-family roofline entries, the end-to-end
traffic from the committed counter passes (profiles/), and the consensus' f64 issue roofline.  Imported by bench.py."""
import json
import os
import sys

import numpy as np

from tools.bench_common import W, H, CAP, FRAMES_PER_STEP  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PROFILE_TAG = "r05"     # the committed counter passes bench.py reads (profiles/<tag>_pmc_traffic.json, _pmc_sq_summary.txt)
PROFILE_TAG_RANSAC = "r04"   # profiles/<tag>_pmc_ransac.json (tools/pmc_ransac.sh)
FED_BYTES_PER_PIXEL_STEP = 12.0
CONTRACT_BYTES_PER_FRAME = 1053518400.0   # A1-A11 per 1080p frame, every buffer once per consuming stage
FP64_VALU_PEAK_TFLOPS = 78.6              # MI355X_MICROARCH.md: FP64 vector
# Kernel families the library times (include/akz.h AKZ_T_*): name, timer id, algorithmic HBM bytes per unit.  A unit
# is one pixel of one frame covered by one launch; the bytes are what the kernel must move once given what it fuses
# (DESIGN.md §5): front-end f32 levels 4 in + 4 Lflow + 8 {Lx,Ly} out; level 0: 1 (u8) in + 4 Lt + 8 {Lx,Ly};
# determinant: 8 in ({Lx,Ly}), candidates only out; FED: 4 L + 4 c in, 4 L out per LAUNCH (T steps share the pass);
# contrast: 1 (u8) in per pass; fused front end + first FED launch (k_front_fed): 4 in (Lt), 4 (Lt') + 8 {Lx,Ly} out —
# Lflow stays on chip (the kernel is VALU-bound, its HBM fraction is what is left of the 28 B the split pair moves); the same
# kernel below the first octave (timer ids 26..28) is counted at 16 B as well — 20 B when a later FED launch of the level
# needs Lflow written, so its fraction is understated there, never overstated.
KERNEL_FAMILIES = [
    ("k_level_front2<4,2,..,u8> (level 0: u8->f32, blur 1.6, Lt, {Lx,Ly})", 3, 13.0),
    ("k_level_front2<2,2,..> (blur 1.0, Scharr, pm_g2 -> Lflow, {Lx,Ly}; sigma 2)", 4, 16.0),
    ("k_level_front2<2,3,..> (blur 1.0, Scharr, pm_g2 -> Lflow, {Lx,Ly}; sigma 3)", 5, 16.0),
    ("k_level_front2<2,4,..> (blur 1.0, Scharr, pm_g2 -> Lflow, {Lx,Ly}; sigma 4)", 6, 16.0),
    ("k_front_fed<2,..> (blur 1.0, Scharr, pm_g2, {Lx,Ly}, first FED launch of the level; sigma 2)", 22, 16.0),
    ("k_front_fed<3,..> (blur 1.0, Scharr, pm_g2, {Lx,Ly}, first FED launch of the level; sigma 3)", 23, 16.0),
    ("k_front_fed<4,..> (blur 1.0, Scharr, pm_g2, {Lx,Ly}, first FED launch of the level; sigma 4)", 24, 16.0),
    ("k_front_fed<2,2,..> below the first octave (front end + the level's first FED launch of up to 8 steps; sigma 2)", 26, 16.0),
    ("k_front_fed<3,2,..> below the first octave (front end + the level's first FED launch of up to 8 steps; sigma 3)", 27, 16.0),
    ("k_front_fed<4,2,..> below the first octave (front end + the level's first FED launch of up to 8 steps; sigma 4)", 28, 16.0),
    ("k_level_resident<..> (a level that fits one compute unit: front end + every FED step in one launch, one workgroup per frame)", 29, 16.0),
    ("k_det_stream<2,..> (Lxx,Lyy,Lxy, Ldet, extrema candidates; sigma 2)", 7, 8.0),
    ("k_det_stream<3,..> (Lxx,Lyy,Lxy, Ldet, extrema candidates; sigma 3)", 8, 8.0),
    ("k_det_stream<4,..> (Lxx,Lyy,Lxy, Ldet, extrema candidates; sigma 4)", 9, 8.0),
    ("k_fed_pair<1> (calculate_step, 1 step per launch)", 14, 12.0),
    ("k_fed_pair<2> (calculate_step, 2 steps per launch)", 15, 12.0),
    ("k_fed_pair<3> (calculate_step, 3 steps per launch)", 16, 12.0),
    ("k_fed_pair<4> (calculate_step, 4 steps per launch)", 17, 12.0),
    ("k_fed_pair<5> (calculate_step, 5 steps per launch)", 18, 12.0),
    ("k_fed_pair<6> (calculate_step, 6 steps per launch)", 19, 12.0),
    ("k_fed_pair<7> (calculate_step, 7 steps per launch)", 20, 12.0),
    ("k_fed_pair<8> (calculate_step, 8 steps per launch)", 21, 12.0),
    ("k_contrast_pair (contrast factor passes)", 10, 1.0),
]
# the keypoint-stage kernel with the most GPU time: gathers, no per-pixel byte model — its roofline numerator is the
# distinct 32-byte sectors the frame's keypoints touch, each once (gather_model); the PMC bytes go beside it as `traffic`
ORIENT_DESCRIBE = ("k_orient_describe (main orientation + M-LDB descriptor, one wave per keypoint)", 25)
MFMA_I8_PEAK_TOPS = 3944.0   # dense int8 MFMA, measured ceiling in MI355X_MICROARCH.md (~2x the bf16 rate)
MFMA_FP4_PEAK_TOPS = 10000.0  # dense FP4/FP6 MFMA (MI355X_MICROARCH.md; AMD's 20 PF headline is 2:1 sparse)


def _short_roofline(e):
    """One flat roofline object for the headline: the contract's keys + the kernel's own launch statistics."""
    if not e:
        return None
    r = {k: e.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_launch_us")}
    if e.get("limited_by"):
        r["limited_by"] = e["limited_by"]
    r["kernel"] = str(e.get("kernel", "")).split(" ")[0]
    for k in ("valu_frac", "gpu_ms_per_step"):
        if e.get(k) is not None:
            r[k] = e[k]
    iso = e.get("isolated") or {}
    if iso.get("frac") is not None:
        r["isolated_frac"] = iso["frac"]
        r["isolated_avg_launch_us"] = iso.get("avg_launch_us")
    return r


def read_families(ctx):
    """(name, ms, launches, units, bytes_per_unit) of every timed kernel family since the last timing_reset."""
    fam = []
    for name, tid, bpu in KERNEL_FAMILIES:
        ms, launches, units = ctx.timing_get(tid)
        if launches:
            fam.append((name, ms, launches, units, bpu))
    ms, launches, units = ctx.timing_get(ORIENT_DESCRIBE[1])
    if launches:
        fam.append((ORIENT_DESCRIBE[0], ms, launches, units, None))
    return fam


VALU_ISSUE_PEAK_T = 39.3     # T lane-instructions / s at FOUR cycles per wave64 instruction (256 CUs x 4 SIMDs x 64 lanes / 4 x
                             # 2.4 GHz): the rate of the instructions the big kernels are made of — packed f32 (v_pk_add/mul_f32: two
                             # lane-ops each, i.e. the 78.6 T lane-op/s non-FMA peak) and f64.  MI355X_MICROARCH.md gives a PLAIN
                             # 32-bit VALU instruction two cycles (SIMD-32: 157.3 TFLOP/s of v_fma_f32), so valu_frac computed with
                             # this constant is the fraction of issue time IF every instruction were packed or f64: exact for the
                             # consensus kernels (f64), close for the diffusion kernels (mostly packed), an UPPER bound for kernels
                             # of scalar 32-bit work (k_orient_describe, the sorts).  The cycle-based counters beside it
                             # (issue_counters.valu_busy_pct: SQ_ACTIVE_INST_VALU over busy cycles) do not depend on it.


def gather_model(ctx, kps_frames, counts):
    """What k_orient_describe MUST fetch, from the kernel's own sampling geometry on the GPU's own keypoints: the orientation
    stage reads {Lx, Ly} (8 B) at the 109 lattice points (x + i s, y + j s), i^2 + j^2 < 36 (scale_space_extrema.rs:230-260),
    the descriptor Lt (4 B) and {Lx, Ly} at the 21 x 21 rotated lattice (descriptors.rs:102-177); every sample pulls the
    32-byte sector it lies in.  Per keypoint the DISTINCT sectors of the {Lx, Ly} plane and of the Lt plane are counted (a
    keypoint's samples are gathered once into LDS, so re-use inside a keypoint is the kernel's to have; re-use between
    keypoints is the cache's).  Returns bytes per FRAME at three granularities: 32-byte sectors per keypoint (the algorithmic
    numerator), the 128-byte lines per keypoint (the memory side fetches whole lines: profiles/r04_fetch_calibration.txt)
    and the distinct sectors of the whole frame (the floor a perfect cache would reach)."""
    nl = ctx.num_levels(W, H)
    lw = np.array([ctx.level(W, H, i).width for i in range(nl)], np.int64)
    loct = np.array([ctx.level(W, H, i).octave for i in range(nl)], np.int64)
    ii, jj = np.meshgrid(np.arange(-6, 7), np.arange(-6, 7))
    m = (ii * ii + jj * jj) < 36
    oi, oj = ii[m].astype(np.float32), jj[m].astype(np.float32)
    kk, ll = np.meshgrid(np.arange(-10, 11), np.arange(-10, 11), indexing="ij")
    kk, ll = kk.reshape(-1).astype(np.float32), ll.reshape(-1).astype(np.float32)
    s32 = l128 = fr32 = fr128 = 0.0
    nkp = 0
    tile_hist = np.zeros(6, np.int64)      # 32-px tiles of a level holding 1, 2, 3-4, 5-8, 9-16, > 16 keypoints

    def distinct(a):
        a = np.sort(a, axis=1)
        return 1 + (np.diff(a, axis=1) != 0).sum(1)
    for f, kp in enumerate(kps_frames):
        kp = kp[:int(counts[f])]
        if len(kp) == 0:
            continue
        cls = kp["class_id"].astype(np.int64)
        ratio = (1 << loct[cls]).astype(np.float32)
        sc = np.round(np.float32(0.5) * kp["size"] / ratio)
        xf, yf = kp["x"] / ratio, kp["y"] / ratio
        w = lw[cls][:, None]
        ox = np.round(xf[:, None] + oi[None, :] * sc[:, None]).astype(np.int64)
        oy = np.round(yf[:, None] + oj[None, :] * sc[:, None]).astype(np.int64)
        co, si = np.cos(kp["angle"]), np.sin(kp["angle"])
        dx = np.round(xf[:, None] + (-ll[None, :] * si[:, None] * sc[:, None] + kk[None, :] * co[:, None] * sc[:, None])).astype(np.int64)
        dy = np.round(yf[:, None] + (ll[None, :] * co[:, None] * sc[:, None] + kk[None, :] * si[:, None] * sc[:, None])).astype(np.int64)
        pix_xy = np.concatenate([oy * w + ox, dy * w + dx], 1)          # {Lx, Ly} plane: orientation + descriptor samples
        pix_lt = dy * w + dx                                            # Lt plane: descriptor samples
        s32 += 32.0 * float(distinct(pix_xy // 4).sum() + distinct(pix_lt // 8).sum())
        l128 += 128.0 * float(distinct(pix_xy // 16).sum() + distinct(pix_lt // 32).sum())
        lvl = cls[:, None] * (1 << 40)
        fr32 += 32.0 * float(len(np.unique((pix_xy // 4 + lvl).reshape(-1))) + len(np.unique((pix_lt // 8 + lvl).reshape(-1))))
        fr128 += 128.0 * float(len(np.unique((pix_xy // 16 + lvl).reshape(-1))) + len(np.unique((pix_lt // 32 + lvl).reshape(-1))))
        # how many keypoints share a 32-px tile of their level (what staging a tile's patch in LDS could amortise over)
        tkey = cls * (1 << 40) + (np.round(yf).astype(np.int64) >> 5) * 4096 + (np.round(xf).astype(np.int64) >> 5)
        _, per_tile = np.unique(tkey, return_counts=True)
        tile_hist += np.bincount(np.searchsorted([1, 2, 4, 8, 16], per_tile, side="left"), minlength=6)[:6]
        nkp += len(kp)
    nf = max(1, len(kps_frames))
    return {"sector_bytes_per_frame": s32 / nf, "line_bytes_per_frame": l128 / nf, "frame_distinct_sector_bytes": fr32 / nf,
            "frame_distinct_line_bytes": fr128 / nf,
            "keypoints_per_32px_tile_histogram": {"1": int(tile_hist[0]), "2": int(tile_hist[1]), "3-4": int(tile_hist[2]), "5-8": int(tile_hist[3]),
                                                  "9-16": int(tile_hist[4]), ">16": int(tile_hist[5]), "frames": len(kps_frames)},
            "keypoints_per_frame": nkp / nf, "frames_sampled": len(kps_frames),
            "what": "32-byte sectors of the {Lx,Ly} (8 B/px) and Lt (4 B/px) planes touched by the 109 orientation samples and the "
                    "21 x 21 descriptor lattice, distinct per keypoint, from this run's own keypoints; line_bytes = the same at the "
                    "128-byte granularity the memory side fetches (profiles/r04_fetch_calibration.txt: every read request is 128 B); "
                    "frame_distinct = distinct sectors of the whole frame (perfect re-use between keypoints); "
                    "frame_distinct_line_bytes = the same in 128-byte lines: what HBM must deliver at the granularity the memory "
                    "side fetches (the PMC traffic is to be read against THIS: the gap to the sector figure is line granularity, not "
                    "re-fetching)"}


def roofline_entries(fam_pipe, fam_iso, mb, steps, gather=None, iso_steps=3):
    """Roofline objects of the timed kernel families.  Per family: frac = hbm_frac = algorithmic bytes / kernel time / 8 TB/s
    (always the BYTES fraction); valu_frac = VALU instructions x 64 lanes / kernel time / the VALU issue peak (counters:
    profiles/, taken at this micro-batch); `bound` is "hbm" — the resource `frac` is a fraction of — and `limited_by` names
    whichever of the two fractions is larger.  Ordered by a family's time per step
    with the GPU to itself (isolated pass) — inside the pipeline three streams time-slice the chip and a kernel's duration
    says how the chip was shared, not what the kernel costs."""
    iso = {f[0]: f for f in (fam_iso or [])}
    pmc = pmc_traffic(mb)
    sq = sq_counters()
    out = []
    for name, ms, launches, units, bpu in fam_pipe:
        if ms <= 0:
            continue
        key = name.split(" ")[0]
        model = None
        if bpu is None:        # the gather kernel: units = frames, bytes from its sampling geometry (gather_model)
            if not gather:
                continue
            # compulsory bytes = every sector the frame's keypoints touch, once (what a perfect cache would fetch); the
            # per-keypoint figures (what the L2 is asked for) go beside it as sector_frac / line_frac
            bpu, model = gather["frame_distinct_sector_bytes"], gather
        gbs = units * bpu / (ms * 1e-3) / 1e9
        e = {"bound": "hbm", "kernel": name, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(gbs / HBM_PEAK_GBS, 4), "frac_is": "algorithmic bytes / kernel time / 8 TB/s",
             "traffic": None, "launches": int(launches),
             "avg_launch_us": round(ms * 1e3 / launches, 2), "gpu_ms": round(ms, 2), "gpu_ms_per_step": round(ms / steps, 3),
             "algorithmic_bytes_per_launch": round(units * bpu / launches), "bytes_per_unit": round(bpu, 3),
             "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
             "timed": f"the launches' own start/stop events (hipExtLaunchKernel: the dispatch's begin -> end, rocprofv3's "
                      f"kernel duration) over the {steps} timed steps; the keypoint and matcher streams of neighbouring "
                      f"micro-batches share the GPU"}
        if model:
            e["byte_model"] = {k: (round(v, 1) if isinstance(v, float) else v) for k, v in model.items()}
            e["sector_demand_frac"] = round(units * model["sector_bytes_per_frame"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            e["line_demand_frac"] = round(units * model["line_bytes_per_frame"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            e["distinct_line_bytes_per_launch"] = round(units * model["frame_distinct_line_bytes"] / launches)
            e["frac_is"] = ("distinct 32-byte sectors the frame's keypoints touch (each once) / kernel time / 8 TB/s; sector_demand_frac / "
                            "line_demand_frac = the same with every keypoint's sectors / 128-byte lines counted on their own (what the "
                            "caches are asked for, not what HBM must deliver: they can exceed 1)")
        if pmc and key in pmc["kernels"]:
            k = pmc["kernels"][key]
            e["traffic"] = round(k["hbm_bytes_per_launch"])
            if model:
                e["traffic_over_distinct_lines"] = round(e["traffic"] / max(1, e["distinct_line_bytes_per_launch"]), 3)
            e["traffic_source"] = {"file": pmc["file"], "micro_batch": pmc["micro_batch"], "launches_counted": k["launches"]}
            if k.get("valu_insts_per_launch"):
                lane_ops = k["valu_insts_per_launch"] * 64.0 / (ms * 1e-3 / launches) / 1e12
                e["valu_frac"] = round(lane_ops / VALU_ISSUE_PEAK_T, 4)
                e["valu"] = {"achieved": round(lane_ops, 2), "peak": VALU_ISSUE_PEAK_T, "unit": "T lane-instr/s",
                             "insts_per_launch": round(k["valu_insts_per_launch"]),
                             "note": "SQ_INSTS_VALU (rocprofv3 --pmc, committed pass at this micro-batch) x 64 lanes / this "
                                     "run's kernel time; the frame-pair kernels issue packed f32 (2 lane-ops per "
                                     "instruction), so this is also their fraction of the 78.6 T lane-op/s non-FMA peak"}
                # `bound` names the resource achieved / peak / frac are measured in (bytes: "hbm"); what the kernel is
                # actually limited by goes beside it
                e["limited_by"] = "valu" if e["valu_frac"] > e["hbm_frac"] else "hbm"
        rank_ms = ms / steps
        if name in iso:
            _, ims, il, iu, _ = iso[name]
            igbs = iu * bpu / (ims * 1e-3) / 1e9
            e["isolated"] = {"achieved": round(igbs, 1), "frac": round(igbs / HBM_PEAK_GBS, 4),
                             "avg_launch_us": round(ims * 1e3 / il, 2), "gpu_ms_per_step": round(ims / iso_steps, 3)}
            if pmc and key in pmc["kernels"] and pmc["kernels"][key].get("valu_insts_per_launch"):
                e["isolated"]["valu_frac"] = round(pmc["kernels"][key]["valu_insts_per_launch"] * 64.0 / (ims * 1e-3 / il) / 1e12 / VALU_ISSUE_PEAK_T, 4)
            rank_ms = ims / iso_steps
        e["rank_ms_per_step"] = round(rank_ms, 3)
        if "below the first octave" in name:
            # these launches write Lflow (20 B per pixel) when a later FED launch of the level reads it, 16 B otherwise; the
            # timer does not separate the two: frac is the 16-byte figure (never overstated), this the 20-byte one
            e["frac_at_20_bytes_per_pixel"] = round(gbs * 20.0 / 16.0 / HBM_PEAK_GBS, 4)
        if name.startswith("k_front_fed"):
            # what the same work cost as two kernels (k_level_front2 16 B + k_fed_pair 12 B per pixel): the fused kernel's
            # time expressed against THOSE bytes, for comparison with round 1's front-end / FED fractions only
            e["replaces"] = {"kernels": "k_level_front2<2,sigma,..> + k_fed_pair<T>", "bytes_per_pixel": 28.0,
                             "equivalent_frac_of_peak": round(gbs * 28.0 / 16.0 / HBM_PEAK_GBS, 4)}
        if key in sq:
            e["issue_counters"] = sq[key]
        out.append(e)
    return out


def pipeline_traffic(mb, nf):
    """HBM bytes per frame of the WHOLE timed pipeline (scale space + keypoint stage + matcher; the library's kernels
    only — frame generation and torch fills are not counted) from the committed counter passes of `bench.py --pmc-run`."""
    pmc = pmc_traffic(mb)
    if not pmc or not pmc.get("per_frame") or int(pmc.get("frames_per_step", 0)) != int(nf):
        return None
    pf = pmc["per_frame"]
    valu = sum(k.get("valu_insts_per_launch", 0) * k["launches"] for name, k in pmc["kernels"].items() if name.startswith("k_"))
    return {"valu_insts": round(valu / (float(pmc["frames_per_step"]) * float(pmc.get("steps", 1)))),
            "bytes": round(pf["hbm_bytes"]), "scale_space_bytes": round(pf.get("scale_space_hbm_bytes", 0)),
            "keypoint_stage_bytes": round(pf.get("keypoint_stage_hbm_bytes", 0)), "matcher_bytes": round(pf.get("matcher_hbm_bytes", 0)),
            "file": pmc["file"], "source": pmc.get("source_short", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of bench.py --pmc-run")}


def ransac_roofline(n_hyp, n, dt):
    """configs[3] against the FP64 vector roofline (SURVEY 8d names it as the bound of R1-R4).  Every VALU instruction of
    k_rsb_hypotheses / k_rsb_score_first is f64 arithmetic or its control overhead; SQ_INSTS_VALU per call comes from the
    committed counter pass of exactly this workload (tools/pmc_ransac.sh -> profiles/<tag>_pmc_ransac.json), the time from
    this run.  frac = wave-instructions x 64 lanes / time / the 39.3 T lane-instructions/s the chip can issue (one f64
    instruction per lane and cycle = the 78.6 TFLOP/s FP64 vector peak counted at 2 flops per FMA; the reference's
    arithmetic is unfused, so a lane-instruction is ONE flop here and flops_frac is half of frac)."""
    note = ("exhaustive_equivalent = what evaluating ~2.4 kflop for EVERY pair at this rate would take; it exceeds what the chip "
            "can do because most pairs are decided by the ~100-flop bound (exact: the inlier sets are the oracle's)")
    base = {"bound": "fp64-valu", "kernel": "k_rsb_score_first + k_rsb_hypotheses (CameraToCamera::residual < threshold per (pose, match): a lower "
                                            "bound first, the 4x4 Jacobi where it does not decide; 9x9 Jacobi + SVD per hypothesis)",
            "achieved": None, "peak": VALU_ISSUE_PEAK_T, "unit": "T f64 lane-instr/s", "frac": None, "traffic": None,
            "exhaustive_equivalent_tflops": round(2400.0 * n_hyp * 4 * n / dt / 1e12, 2), "note": note}
    try:
        name = PROFILE_TAG_RANSAC + "_pmc_ransac.json"
        with open(os.path.join(ROOT, "profiles", name)) as f:
            d = json.load(f)
        if "10 000" not in d["workload"] or n_hyp != 10000 or n != 1000:
            return base                      # counters of another workload: nothing is rescaled
        insts = float(d["valu_insts_per_call"])
        whole = insts * 64.0 / dt / 1e12
        base.update({"achieved": round(whole, 2), "frac": round(whole / VALU_ISSUE_PEAK_T, 4),
                     "flops_frac": round(whole / FP64_VALU_PEAK_TFLOPS, 4),
                     "frac_is": "SQ_INSTS_VALU of one call x 64 lanes / this run's wall time per call (host buffers in and out, launches "
                                "included) / 39.3 T lane-instr/s; flops_frac = the same lane-instructions as flops (unfused: one each) / 78.6 TFLOP/s",
                     "valu_insts_per_call": round(insts), "counters": "profiles/" + name, "kernels": {}})
        for k, v in d["kernels"].items():
            if v["kernel_us_per_call"] > 0 and v["valu_insts_per_call"] > 1e6:
                r = v["valu_insts_per_call"] * 64.0 / (v["kernel_us_per_call"] * 1e-6) / 1e12
                base["kernels"][k] = {"valu_insts_per_call": round(v["valu_insts_per_call"]), "kernel_us_per_call": round(v["kernel_us_per_call"], 1),
                                      "frac": round(r / VALU_ISSUE_PEAK_T, 4), "waves_per_call": round(v["waves_per_call"]),
                                      "timed": "rocprofv3 --kernel-trace of the committed pass (the kernel's own duration)"}
    except Exception:
        pass
    return base


def pmc_traffic(mb):
    """HBM bytes per launch of each kernel from the committed rocprofv3 PMC passes (profiles/r02_pmc_traffic.json,
    made by tools/pmc_traffic.py: WRITE_SIZE and doubled FETCH_SIZE per MI355X_MICROARCH.md's gfx950 correction,
    separate --pmc passes).  Only used when the counters were taken at THIS micro-batch: nothing is rescaled."""
    try:
        name = PROFILE_TAG + "_pmc_traffic.json"
        with open(os.path.join(ROOT, "profiles", name)) as f:
            d = json.load(f)
        if int(d["micro_batch"]) != int(mb):
            return None
        d["file"] = "profiles/" + name
        return d
    except Exception:
        return None


def sq_counters():
    """Issue-side counters of each kernel from the committed SQ passes (profiles/r02_pmc_sq_summary.txt, made by
    tools/pmc_sq.sh over the serial phase profile at 64 frames per launch): what a kernel that is not HBM-bound is
    bound by.  Keyed like the kernel families."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from pmc_traffic import family_key
        out = {}
        with open(os.path.join(ROOT, "profiles", PROFILE_TAG + "_pmc_sq_summary.txt")) as f:
            for line in f.read().splitlines()[1:]:
                name, rest = line[:52].strip(), line[52:].split()
                if len(rest) < 8:
                    continue
                key = family_key(name + ">") if name.count("<") > name.count(">") else family_key(name)
                out.setdefault(key, {"file": "profiles/" + PROFILE_TAG + "_pmc_sq_summary.txt", "valu_instructions_per_wave": int(rest[1]),
                                     "valu_busy_pct": int(rest[2]), "lds_busy_pct": int(rest[3]),
                                     "lds_bank_conflict_pct": int(rest[4]), "waves_parked_pct": int(rest[5])})
        return out
    except Exception:
        return {}
