"""The extra legs of bench.py (configs_extra of the detail report): BASELINE configs[2] and configs[3], the pipeline+verify
and pipeline+register chains and the criterion rows.  Each leg times the HIP path and holds a sample of its output to
the oracle (checker use only: nothing here is part of the timed region's product path).  Imported by bench.py."""
import ctypes as C
import os
import sys
import time

import numpy as np

from tools.bench_common import W, H, CAP, FRAMES_PER_STEP, make_frames  # noqa: F401
from tools.roofline import *  # noqa: F401,F403

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def extra_match(torch, dev, L, _lib, n_frames):
    """BASELINE configs[2] as SURVEY 8d defines it: n_frames x 5 000 descriptors; frame 0 = 486 i.i.d. Bernoulli(1/2)
    bits (seed 0xD35C); frame f+1 = 60 % of frame f's descriptors with every bit flipped w.p. 0.05 + 40 % fresh ones,
    shuffled; consecutive pairs matched symmetrically with d0 + 24 < d1.  Device-resident, one call."""
    from cv_amd.knn import Matcher, RULE_STRICT
    from oracle import oracle as O
    ND, cap = 5000, 5000
    g = torch.Generator(device=dev).manual_seed(0xD35C)
    bitmask = torch.zeros(64, dtype=torch.uint8, device=dev)
    bitmask[:60] = 0xFF
    bitmask[60] = 0x3F                      # bits 486..511 stay zero
    w8 = (1 << torch.arange(8, device=dev, dtype=torch.int32)).to(torch.uint8)

    def fresh(n):
        return torch.randint(0, 256, (n, 64), generator=g, device=dev, dtype=torch.uint8) & bitmask

    descs = torch.empty((n_frames, cap, 64), dtype=torch.uint8, device=dev)
    descs[0] = fresh(ND)
    for f in range(1, n_frames):
        keep = torch.randperm(ND, generator=g, device=dev)[:ND * 6 // 10]
        flips = (torch.rand((len(keep), 64, 8), generator=g, device=dev) < 0.05).to(torch.uint8)
        flip_bytes = (flips * w8).sum(dim=2).to(torch.uint8) & bitmask
        nxt = torch.cat([descs[f - 1][keep] ^ flip_bytes, fresh(ND - len(keep))])
        descs[f] = nxt[torch.randperm(ND, generator=g, device=dev)]
    counts = torch.full((n_frames,), ND, dtype=torch.int32, device=dev)
    npr = n_frames - 1
    pairs = torch.zeros((npr, cap, 2), dtype=torch.int32, device=dev)
    npairs = torch.zeros((npr,), dtype=torch.int32, device=dev)
    m = Matcher(cap)
    ia = (C.c_uint32 * npr)(*range(1, n_frames))
    ib = (C.c_uint32 * npr)(*range(0, n_frames - 1))

    def run():
        _lib.check(L.hm_match_batch_device(m.handle, descs.data_ptr(), counts.data_ptr(), descs.data_ptr(),
                                           counts.data_ptr(), cap, ia, ib, npr, RULE_STRICT, 24, 0.0, 1,
                                           pairs.data_ptr(), npairs.data_ptr(), None), "match")
    torch.cuda.synchronize()
    run()
    _lib.check(L.hm_sync(m.handle), "sync")
    _lib.check(L.hm_timing_get(m.handle, None, None, 1), "timing")
    _lib.check(L.hm_timing_enable(m.handle, 1), "timing")
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    _lib.check(L.hm_sync(m.handle), "sync")
    dt = (time.perf_counter() - t0) / reps
    ms, launches = C.c_double(), C.c_uint64()
    _lib.check(L.hm_timing_get(m.handle, C.byref(ms), C.byref(launches), 1), "timing")
    _lib.check(L.hm_timing_enable(m.handle, 0), "timing")
    # oracle on a sample of the pairs
    sample = sorted({int(v) for v in np.linspace(0, npr - 1, min(npr, 64))})
    bad = 0
    hd = descs.cpu().numpy()
    for p_ in sample:
        want = O.match(hd[p_ + 1], hd[p_], rule=O.RULE_STRICT, param_u=24, symmetric=True).astype(np.uint32)
        k = int(npairs[p_].item())
        got = pairs[p_, :k].cpu().numpy().astype(np.uint32)
        bad += int(k != len(want) or not np.array_equal(got, want))
    dist_per_pair = 2.0 * ND * ND                      # both directions
    ops = 2.0 * 512.0 * dist_per_pair * npr * reps     # one MAC = 2 ops per bit of the 512-deep contraction
    tops = ops / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
    out = {"workload": f"{n_frames} frames x {ND} descriptors (Bernoulli(1/2) x 486 bits, 60 % carried over with 5 % "
                       f"bit flips), {npr} consecutive pairs, symmetric d0 + 24 < d1, device-resident",
           "pairs_per_s": round(npr / dt, 1), "distances_per_s": round(dist_per_pair * npr / dt, 1),
           "ms_per_call": round(dt * 1e3, 3), "mean_matches_per_pair": round(npairs.float().mean().item(), 1),
           "roofline": {"bound": "mfma", "kernel": "k_knn_mfma4w<2> (v_mfma_scale_f32_32x32x64_f8f6f4, E2M1 operands)",
                        "achieved": round(tops, 1), "peak": MFMA_FP4_PEAK_TOPS, "unit": "TOP/s",
                        "frac": round(tops / MFMA_FP4_PEAK_TOPS, 4), "traffic": None,
                        "launches": int(launches.value),
                        "avg_launch_us": round(ms.value * 1e3 / max(1, launches.value), 2)},
           "parity": {"pairs_checked": len(sample), "mismatches": bad,
                      "what": "match pair lists of the sampled frame pairs vs oracle/match_oracle.c"}}
    m.close()
    return out


def extra_pipeline_verify(torch, dev, L, _lib, args, step, step_no, barrier, verify, match_done, hm_stream, kps2, pairs2, npairs2,
                          NF, MB):
    """The headline pipeline with the stage that consumes its match lists attached: every frame pair of every
    micro-batch goes from the matcher straight into rs_essential_arrsac_batch_device (calibrate -> seeded shuffle ->
    8192 eight-point hypotheses -> block scoring with a halving candidate set of 1024, SPRT; vslam-sandbox/src/main.rs:
    112-117, cv-sfm/src/lib.rs:1385-1412), nothing leaves the device.  value = verified frame pairs per second of the
    whole pipeline; a sample of scenes from different micro-batch positions is held to oracle/arrsac_oracle.c."""
    from cv_amd.ransac import EssentialConsensus
    from oracle import oracle as O
    cam = (1000.0, 1000.0, W / 2.0, H / 2.0, 0.0, None)     # a pinhole camera for the synthetic frames
    n_hyp, thr = 8192, 1e-7                                  # initialization_hypotheses, two_view_consensus_threshold
    kw = dict(block_size=args.verify_block, init_blocks=1, max_candidates=1024, halve=True, sprt=True)
    cons = EssentialConsensus(CAP, n_hyp)
    cons.reserve(MB + 1)
    prm = cons.make_params(thr, n_hypotheses=n_hyp, seed=0, **kw)
    c = cons.camera(cam)
    rs_stream = torch.cuda.ExternalStream(cons.stream(), device=dev)
    z = lambda shape, dt: [torch.zeros(shape, dtype=dt, device=dev) for _ in range(2)]
    pose2, best2, inl2, ninl2 = z((NF + 2, 12), torch.float64), z((NF + 2,), torch.int32), z((NF + 2, CAP), torch.int32), z((NF + 2,), torch.int32)
    stats2 = z((NF + 2, 32), torch.uint8)
    verify_done = [torch.cuda.Event(), torch.cuda.Event()]
    calls = {}

    def enqueue(p, m0, js, prev_js):
        cons.model_inliers_batch_device(kps2[p].data_ptr(), kps2[p].data_ptr(), CAP, js, prev_js, pairs2[p][m0:].data_ptr(),
                                        npairs2[p][m0:].data_ptr(), c, c, prm, pose2[p][m0:].data_ptr(), best2[p][m0:].data_ptr(),
                                        inl2[p][m0:].data_ptr(), ninl2[p][m0:].data_ptr(), stats2[p][m0:].data_ptr(),
                                        shuffle=True, stream_to_wait=_lib.wait_handle(hm_stream))
        calls[(p, m0)] = (list(js), list(prev_js))
        if m0 + MB >= NF:
            verify_done[p].record(rs_stream)
            if p == 1:
                verify["armed"] = True          # both events have been recorded once
    barrier()
    verify["done"] = verify_done
    verify["on"] = enqueue
    step(); step()                                  # warm-up: both output sets
    cons.sync()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.verify_steps):
        step()
    cons.sync()
    barrier()
    dt = (time.perf_counter() - t0) / args.verify_steps
    verify["on"] = None
    last = (step_no[0] - 1) & 1
    kps, pairs, npairs = kps2[last], pairs2[last], npairs2[last]
    hbest = best2[last].cpu().numpy().view(np.uint32); hninl = ninl2[last].cpu().numpy().view(np.uint32)
    hst = stats2[last].cpu().numpy().view(np.dtype([("poses", "<u4"), ("survivors", "<u4"), ("blocks", "<u4"), ("reserved", "<u4"),
                                                      ("evaluated", "<u8"), ("exhaustive", "<u8")])).reshape(-1)
    hn = npairs.cpu().numpy()
    # parity: scenes spread over the micro-batches of the last step (first, last, odd positions)
    slots = []
    for m0 in range(0, NF, MB):
        js, prev_js = calls[(last, m0)]
        per = max(1, args.verify_check // max(1, NF // MB))
        q = sorted({0, len(js) - 1} | {min(len(js) - 1, (k * len(js) // per) | 1) for k in range(per)})
        slots += [(m0, qq, js[qq], prev_js[qq]) for qq in q if qq < len(js)]
    slots = slots if args.verify_check else []
    bad, detail = 0, []
    t0 = time.perf_counter()
    for m0, q, ja, jb in slots:
        n = int(hn[m0 + q])
        ka = kps[ja].cpu().numpy().view(_lib.KP_DTYPE).reshape(-1)
        kb = kps[jb].cpu().numpy().view(_lib.KP_DTYPE).reshape(-1)
        pr = pairs[m0 + q, :n].cpu().numpy().astype(np.uint32)
        w = O.arrsac_pairs(ka, kb, pr, cam, cam, thr, n_hyp, scene=q, shuffle=True, seed=0, **kw)
        g_inl = inl2[last][m0 + q, :hninl[m0 + q]].cpu().numpy().view(np.uint32)
        g_pose = pose2[last][m0 + q].cpu().numpy()
        ok = (hbest[m0 + q] == w["best_id"] and np.array_equal(g_inl, w["inliers"])
              and (w["best_id"] == 0xFFFFFFFF or g_pose.tobytes() == w["pose"].tobytes()))
        if not ok:
            bad += 1
            detail.append(f"pair ({ja},{jb}): id {int(hbest[m0 + q])} vs {w['best_id']}, inliers {len(g_inl)} vs {len(w['inliers'])}")
    cpu_s = time.perf_counter() - t0
    valid = hn[:NF] >= 8
    out = {"workload": f"configs[1] batch ({NF} frames of 1920x1080 per step) -> extract -> symmetric better-by-24 match of consecutive "
                       f"frames -> two-view ARRSAC of every pair on the device ({n_hyp} eight-point hypotheses, threshold {thr:g}, "
                       f"{kw['block_size']}-match blocks, candidates 1024 halving per block, SPRT, seeded shuffle)",
           "verified_pairs_per_s": round(NF / dt, 1), "ms_per_step": round(dt * 1e3, 3), "steps": args.verify_steps,
           "mean_matches_per_pair": round(float(hn[:NF].mean()), 1),
           "mean_inliers_per_pair": round(float(hninl[:NF].mean()), 1),
           "pairs_with_a_model": int((hbest[:NF] != 0xFFFFFFFF).sum()),
           "residuals_evaluated_frac": round(float(hst["evaluated"][:NF][valid].sum()) / max(1.0, float(hst["exhaustive"][:NF][valid].sum())), 5),
           "parity": {"scenes_checked": len(slots), "mismatches": bad, "detail": detail[:4], "cpu_s_per_scene": round(cpu_s / max(1, len(slots)), 2),
                      "what": "winner id, pose bits and inlier list of scenes taken from the first / last / odd positions of "
                              "the last step's micro-batches vs oracle/arrsac_oracle.c (orc_arrsac_pairs) on the GPU's own "
                              "keypoints and pair lists"}}
    cons.close()
    return out


def extra_pipeline_register(torch, dev, L, _lib, args, ctx, frames, NF, MB):
    """The loop vslam-sandbox runs on every frame once a reconstruction exists (cv-sfm/src/lib.rs:672, 1452-1542, 1549-1604,
    1619-1622), for whole micro-batches, nothing leaving the device: extract -> hasher.hash_bag -> knn(., 3) of every feature
    against each of the frame's recent views (tracking_recent_frames = 32) -> landmark dedup / three best / unique-match decision
    -> duplicate-landmark filter + FeatureWorldMatch list -> Arrsac + LambdaTwist (vslam-sandbox/src/main.rs:105-111: 16 384
    hypotheses, 1 024 candidates, 256 estimations per block).  cv_amd/registration.py chains the five device-resident entry
    points; the reference's control plane is played by torch on the device: the landmark a stored feature observes is the
    world-canvas cell (4 px, per evolution level) its keypoint falls into, the landmark table the cell centres on the plane the
    panning camera looks at.  value = registered frames per second of the whole chain; sampled frames are held to the oracle
    stage by stage (pair lists, winner, pose bits, inlier lists) and every pose to the motion the frames were rendered with."""
    from cv_amd.registration import Registration
    from oracle import oracle as O
    V = max(1, min(args.register_views, NF - 1))
    CELL, F_CAM, Z0 = 4, 1000.0, 5.0
    cam = (F_CAM, F_CAM, W / 2.0, H / 2.0, 0.0, None)
    wc, hc = (W + 4 * NF) // CELL + 2, (H + 2 * NF) // CELL + 2
    n_world = wc * hc * 16
    keys = torch.arange(n_world, device=dev, dtype=torch.int64)
    cell = keys // 16
    xw = ((cell % wc).to(torch.float64) + 0.5) * CELL
    yw = ((cell // wc).to(torch.float64) + 0.5) * CELL
    P = torch.stack([(xw - W / 2.0) * Z0 / F_CAM, (yw - H / 2.0) * Z0 / F_CAM, torch.full_like(xw, Z0), torch.ones_like(xw)], 1)
    d_world = (P / torch.linalg.norm(P[:, :3], dim=1, keepdim=True)).contiguous()
    del keys, cell, xw, yw, P
    rng = np.random.default_rng(0xC0DE)
    codewords = rng.integers(0, 256, (4096, 64), dtype=np.uint8)        # cv-sfm ships 4096 words (cv-sfm/src/codewords.rs)
    thr, n_hyp, kw = 1e-5, 16384, dict(block_size=64, max_candidates=1024, estimations_per_block=256)
    reg = Registration(torch, CAP, NF, V, codewords, cam, device=dev.index, threshold=thr, n_hypotheses=n_hyp, seed=0, **kw)
    rs_s = torch.cuda.ExternalStream(reg.rs_stream(), device=dev)
    akz_s = torch.cuda.ExternalStream(L.akz_stream(ctx.handle), device=dev)
    z2 = lambda shape, dt: [torch.zeros(shape, dtype=dt, device=dev) for _ in range(2)]
    kps2, descs2, counts2, lm2 = z2((NF, CAP, 28), torch.uint8), z2((NF, CAP, 64), torch.uint8), z2((NF,), torch.int32), z2((NF, CAP), torch.int32)
    gidx = torch.arange(NF, device=dev, dtype=torch.float32).view(NF, 1)
    frame_blocks = list(range(NF))
    view_blocks = [[(j - 1 - v) % NF for v in range(V)] for j in range(NF)]
    glue = torch.cuda.Stream(device=dev)
    done = [torch.cuda.Event(), torch.cuda.Event()]
    n = [0]

    def step():
        p = n[0] & 1
        cur = torch.cuda.current_stream()
        if n[0] >= 2:
            cur.wait_event(done[p])                       # set p's keypoints / descriptors were last read two steps ago
        for m0 in range(0, NF, MB):
            _lib.check(L.akz_extract_batch_device(ctx.handle, frames[m0:m0 + MB].data_ptr(), 0, MB, W, H, kps2[p][m0:m0 + MB].data_ptr(),
                                                  descs2[p][m0:m0 + MB].data_ptr(), CAP, counts2[p][m0:m0 + MB].data_ptr(),
                                                  _lib.wait_handle(cur)), "extract")
        # the caller's bookkeeping: which landmark every feature of every stored view observes
        glue.wait_stream(akz_s)
        with torch.cuda.stream(glue):
            k = kps2[p].view(torch.float32).view(NF, CAP, 7)
            cx = torch.clamp(torch.floor((k[..., 0] + 4.0 * gidx) / CELL), 0, wc - 1).to(torch.int32)
            cy = torch.clamp(torch.floor((k[..., 1] + 2.0 * gidx) / CELL), 0, hc - 1).to(torch.int32)
            cls = kps2[p].view(torch.int32).view(NF, CAP, 7)[..., 6] & 15
            lm2[p].copy_((cy * wc + cx) * 16 + cls)
        reg.enqueue(kps2[p], descs2[p], counts2[p], frame_blocks, view_blocks, lm2[p], d_world, n_world, stream_to_wait=_lib.wait_handle(glue))
        done[p].record(rs_s)
        n[0] += 1

    step(); step()
    reg.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.register_steps):
        step()
    reg.sync(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.register_steps
    last = (n[0] - 1) & 1
    kps = kps2[last].cpu().numpy().view(_lib.KP_DTYPE).reshape(NF, CAP)
    counts = counts2[last].cpu().numpy()
    lms = lm2[last].cpu().numpy().view(np.uint32)
    h_np = reg.npairs.cpu().numpy().view(np.uint32); h_id = reg.best_id.cpu().numpy().view(np.uint32)
    h_ninl = reg.n_inliers.cpu().numpy().view(np.uint32); h_pose = reg.pose.cpu().numpy().reshape(NF, 3, 4)
    h_dec = reg.decision.cpu().numpy().view(np.uint32)
    # every pose against the motion the frames were rendered with: identity rotation, camera at (4 g, 2 g) px on the canvas
    have = h_id != 0xFFFFFFFF
    g = np.arange(NF)
    expect_t = -np.stack([4.0 * g * Z0 / F_CAM, 2.0 * g * Z0 / F_CAM, np.zeros(NF)], 1)
    rot_err = np.abs(h_pose[:, :, :3] - np.eye(3)).max((1, 2))
    t_err = np.abs(h_pose[:, :, 3] - expect_t).max(1)
    pose_ok = have & (rot_err < 0.03) & (t_err < 0.15)
    # sampled frames stage by stage against the oracle, on the GPU's own intermediate data
    world = None
    bad, detail, checked = 0, [], 0
    t0 = time.perf_counter()
    if args.register_check:
        world = d_world.cpu().numpy()
        descs = descs2[last].cpu().numpy()
        for f in sorted({0, NF - 1} | {(i * NF // args.register_check) | 1 for i in range(args.register_check)})[:args.register_check]:
            nq = int(counts[f])
            gk = reg.knn[f].cpu().numpy()
            gnb = np.zeros((V, CAP, 3), _lib.NB_DTYPE)
            gnb["index"] = gk[..., 0]; gnb["distance"] = gk[..., 1]
            v = (f * 7) % V
            tb = view_blocks[f][v]
            wk = O.knn(descs[f, :200], descs[tb, :counts[tb]], 3)
            ok = np.array_equal(gnb["index"][v, :200], wk["index"]) and np.array_equal(gnb["distance"][v, :200], wk["distance"])
            wbest, wdec = O.best_of_views(gnb, nq, lms, np.array(view_blocks[f], np.uint32), counts.astype(np.uint32), 24)
            ok = ok and np.array_equal(reg.best[f, :nq].cpu().numpy().view(np.uint32), wbest) and np.array_equal(h_dec[f, :nq], wdec)
            wpairs = O.landmark_pairs(wbest, wdec, world)
            gp = reg.pairs[f, :h_np[f]].cpu().numpy().view(np.uint32)
            ok = ok and len(wpairs) == h_np[f] and np.array_equal(gp, wpairs)
            want = O.p3p_arrsac_pairs(kps[f], wpairs, world, cam, thr, n_hyp, scene=f, shuffle=True, seed=0, init_blocks=1, halve=True,
                                      sprt=True, **kw)
            g_inl = reg.inliers[f, :h_ninl[f]].cpu().numpy().view(np.uint32)
            ok = ok and h_id[f] == want["best_id"] and np.array_equal(g_inl, want["inliers"]) and \
                (want["best_id"] == 0xFFFFFFFF or h_pose[f].tobytes() == want["pose"].tobytes())
            checked += 1
            if not ok:
                bad += 1
                detail.append(f"frame {f}: pairs {int(h_np[f])} vs {len(wpairs)}, id {int(h_id[f])} vs {want['best_id']}, inliers {int(h_ninl[f])} vs {len(want['inliers'])}")
    cpu_s = time.perf_counter() - t0
    nq_mean = float(counts.mean())
    dist = float(sum(int(counts[j]) * int(counts[view_blocks[j]].sum()) for j in range(NF)))
    out = {"workload": f"configs[1] batch ({NF} frames of 1920x1080 per step) -> extract -> hash_bag (4096 codewords) -> knn(., 3) of every "
                       f"feature against each of {V} recent views ({NF * V} problems of ~{int(nq_mean)}^2) -> best-of-views (better_by 24) -> "
                       f"(feature, landmark) pair lists -> Lambda Twist ARRSAC per frame ({n_hyp} hypotheses, candidates 1024 halving, 256 "
                       f"estimations per block, threshold {thr:g}, seeded shuffle); landmarks = 4-px world-canvas cells per level (synthetic "
                       f"control plane, torch on the device)",
           "registered_frames_per_s": round(NF / dt, 1), "ms_per_step": round(dt * 1e3, 3), "steps": args.register_steps,
           "knn_distances_per_s": round(dist / dt, 1),
           "mean_features_per_frame": round(nq_mean, 1), "mean_unique_matches_per_frame": round(float((h_dec == 1).sum()) / NF, 1),
           "mean_world_matches_per_frame": round(float(h_np.mean()), 1), "mean_inliers_per_frame": round(float(h_ninl.mean()), 1),
           "frames_with_a_model": int(have.sum()),
           "frames_whose_pose_is_the_rendered_motion": int(pose_ok.sum()),
           "pose_error": {"rotation_max_abs": round(float(rot_err[have].max()) if have.any() else -1.0, 6),
                          "translation_max_abs": round(float(t_err[have].max()) if have.any() else -1.0, 6),
                          "bounds": "rotation entries within 0.03 of the identity, translation within 0.15 of the rendered camera position "
                                    "(cell centres stand in for triangulated landmarks: +-2 px at f = 1000 on a plane 5 units away)"},
           "parity": {"frames_checked": checked, "mismatches": bad, "detail": detail[:4], "cpu_s_per_frame": round(cpu_s / max(1, checked), 2),
                      "what": "knn(., 3) of 200 features against one view, best-of-views + decisions of all features, the (feature, "
                              "landmark) pair list, and the consensus (winner id, pose bits, inlier list) of sampled frames vs "
                              "oracle/match_oracle.c + oracle/arrsac_oracle.c (orc_p3p_arrsac_pairs) on the GPU's own intermediate data"}}
    if pose_ok.sum() < 0.9 * NF:
        out["parity"]["mismatches"] += 1
        out["parity"]["detail"].append(f"only {int(pose_ok.sum())} of {NF} poses are the rendered motion")
    reg.close()
    return out


def extra_ransac(n_hyp):
    """BASELINE configs[3] as SURVEY 8d defines it: the scene of eight-point/tests/random.rs with 1 000 matches, 30 %
    outliers, seed 0x5AC, n_hyp eight-sample hypotheses, threshold 1e-7; host buffers in and out."""
    from cv_amd.ransac import EssentialConsensus
    from oracle import oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_parity import _two_view_scene
    rng = np.random.default_rng(0x5AC)
    n, thr = 1000, 1e-7
    a, b = _two_view_scene(rng, n, 0.3)
    samples = np.stack([rng.choice(n, 8, replace=False) for _ in range(n_hyp)]).astype(np.uint32)
    resample = 64                                   # arrsac's estimations_per_block in the full-shape leg
    cons = EssentialConsensus(n, n_hyp + resample * 16)
    cons.model_inliers(a, b, samples, thr)          # warm-up
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        pose, inl, best = cons.model_inliers(a, b, samples, thr)
    dt = (time.perf_counter() - t0) / reps
    counts = cons.counts(n_hyp)
    # oracle: the first `sub` hypotheses in full (per-(hypothesis, pose) inlier counts), and the winning hypothesis
    # on its own (pose bits and inlier set)
    sub = min(1024, n_hyp)
    t0 = time.perf_counter()
    _, _, _, wcounts = O.essential_batch(a, b, samples[:sub], thr)
    cpu_s = time.perf_counter() - t0
    bad = int(not np.array_equal(counts[:sub], wcounts))
    h = best // 4
    wpose, wbest, winl, wc1 = O.essential_batch(a, b, samples[h:h + 1], thr)
    bad += int(wbest != best % 4 or wpose.tobytes() != pose.tobytes() or not np.array_equal(winl, inl))
    bad += int(int(counts.max()) != len(inl) or not np.array_equal(wc1[0], counts[h]))
    # the same scene through the ARRSAC-shaped entry point: samples drawn on the device, block scoring with the exact
    # bound, the candidate cap and the SPRT test (vslam-sandbox's parameters); and with the bound alone
    arr = {}
    full = dict(max_candidates=1024, bound=True, sprt=True, halve=True, estimations_per_block=resample)
    for name, kw in (("bound_cap_sprt", dict(max_candidates=1024, bound=True, sprt=True)),
                     ("bound_only", dict(max_candidates=0, bound=True, sprt=False)),
                     ("halving_cap_sprt_resampling", full)):
        cons.arrsac_model_inliers(a, b, thr, n_hypotheses=n_hyp, seed=0, **kw)
        t0 = time.perf_counter()
        for _ in range(reps):
            apose, ainl, abest, ast = cons.arrsac_model_inliers(a, b, thr, n_hypotheses=n_hyp, seed=0, **kw)
        adt = (time.perf_counter() - t0) / reps
        arr[name] = {"hypotheses_per_s": round(n_hyp / adt, 1), "ms_per_scene": round(adt * 1e3, 3),
                     "residuals_evaluated_frac": round(ast["residuals_evaluated"] / ast["residuals_exhaustive"], 4),
                     "survivors": ast["survivors"], "inliers": int(len(ainl)), "best_id": int(abest)}
    # exhaustive scoring of the device-drawn samples: the bound-only run must give the same winner
    dsamples = cons.arrsac_samples(0, n, n_hyp)
    epose, einl, ebest = cons.model_inliers(a, b, dsamples, thr)
    bad += int(arr["bound_only"]["best_id"] != ebest or arr["bound_only"]["inliers"] != len(einl))
    # the full shape against its specification (oracle/arrsac_oracle.c) at a size the CPU finishes in seconds
    sub_h = min(2048, n_hyp)
    got = cons.arrsac_model_inliers(a, b, thr, n_hypotheses=sub_h, seed=0, **full)
    want = O.arrsac(a, b, thr, sub_h, seed=0, **full)
    spec_bad = int(got[2] != want[2] or got[0].tobytes() != want[0].tobytes() or not np.array_equal(got[1], want[1])
                   or any(got[3][k] != want[3][k] for k in ("survivors", "blocks", "poses", "residuals_evaluated")))
    bad += spec_bad
    arr["spec_parity"] = {"hypotheses": sub_h, "mismatches": spec_bad,
                          "what": "winner id, pose bits, inlier list, survivors, blocks, poses made and residuals "
                                  "evaluated of halving_cap_sprt_resampling vs oracle/arrsac_oracle.c"}
    arr["note"] = ("rs_essential_arrsac, minimal samples drawn on the device (xoshiro256++, seed 0); bound_only is "
                   "checked against exhaustive scoring of the same samples; exhaustive_same_samples_best_id "
                   f"{int(ebest)}, inliers {len(einl)}; halving_cap_sprt_resampling: candidate cap 1024 halving per "
                   f"block, SPRT, {resample} hypotheses re-sampled from the best pose's inliers after every block")
    # The exact statement costs ~2.4 kflop of f64 per (pose, match) — 4x4 design matrix (~250 flops) + cyclic Jacobi (~6 sweeps
    # x 6 rotations x ~60 flops) — but most pairs never reach it: rs_pair_far proves residual >= threshold from the rays'
    # angle to each other's epipolar plane (~100 flops) and whole waves of such pairs skip the eigen-decomposition.  The
    # rate is therefore reported as residual DECISIONS per second, not as a fraction of the f64 peak.
    out = {"workload": f"{n_hyp} eight-point hypotheses x 4 poses x {n} matches (30 % outliers), threshold 1e-7, "
                       "host buffers in and out",
           "hypotheses_per_s": round(n_hyp / dt, 1), "residuals_per_s": round(n_hyp * 4 * n / dt, 1),
           "ms_per_scene": round(dt * 1e3, 3), "inliers": int(len(inl)), "best_id": int(best),
           "roofline": ransac_roofline(n_hyp, n, dt),
           "arrsac": arr,
           "cpu_oracle": {"hypotheses_per_s": round(sub / cpu_s, 1), "cores": 1,
                          "sample": f"first {sub} hypotheses, {cpu_s:.1f} s"},
           "parity": {"hypotheses_checked": sub + 1, "mismatches": bad,
                      "what": "inlier counts of the first hypotheses x 4 poses, and the winning hypothesis' pose bits, "
                              "pose index and inlier set, vs oracle/ransac_oracle.c"}}
    cons.close()
    return out


def extra_criterion(_lib):
    """The reference's own benchmark harness (akaze/benches/criterion.rs:8-52 — the one workload anybody with cargo can
    reproduce): `extract` = Akaze::sparse().extract_from_gray_float_image on res/0000000000.png (1241 x 376, the first KITTI
    fixture), and horizontal_filter / vertical_filter of that image with gaussian_kernel(1.0, 7) and gaussian_kernel(10.0, 71).
    GPU through the C ABI with HOST buffers in and out, as a criterion iteration has them (akz_extract_gray_f32,
    akz_horizontal_filter / akz_vertical_filter); beside it the -O3 -march=native build of the oracle, one thread, on the same
    arrays; outputs compared bit for bit."""
    from cv_amd import akaze as A
    from oracle import oracle as O
    z = np.load(os.path.join(ROOT, "tests", "golden", "kitti_pair.npz"))
    img8 = z["frame0"]
    img = O.u8_to_f32(img8)
    h, w = img.shape
    ak = A.Akaze.sparse()
    ctx = ak.context(w, h, 1)
    fast = O.fast_lib()
    for fn in ("orc_horizontal_filter", "orc_vertical_filter"):
        getattr(fast, fn).argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]

    def best_of(f, reps):
        t = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = f()
            t.append(time.perf_counter() - t0)
        return r, float(np.median(t))
    out = {"image": f"res/0000000000.png ({w} x {h}), tests/golden/kitti_pair.npz", "rows": {}, "mismatches": 0,
           "how": "median wall time per call; GPU = the C ABI with host buffers in and out (one frame per call: launch latency, "
                  "not throughput, is what is measured); CPU = oracle/ at -O3 -march=native, one thread; outputs bit-identical"}
    ak.extract_from_gray_float_image(img)                       # context, tables
    (gk, gd), g_s = best_of(lambda: ak.extract_arrays(img), 20)
    cfg = O.default_config(threshold=0.01)
    O.extract_match_many(img8[None], threads=1, match=False, cfg=cfg)
    cres, c_s = best_of(lambda: O.extract_match_many(img8[None], threads=1, match=False, cfg=cfg), 3)
    same = gk.tobytes() == cres[0][0].tobytes() and np.array_equal(gd, cres[0][1])
    out["rows"]["extract"] = {"gpu_ms": round(g_s * 1e3, 3), "cpu_ms": round(c_s * 1e3, 2), "descriptors": int(len(gd)), "bit_identical": bool(same),
                              "reference": "Akaze::sparse().extract_from_gray_float_image (criterion.rs:8-15); 399 descriptors (estimate_pose.rs:41)"}
    out["mismatches"] += int(not same) + int(len(gd) != 399)
    for kname, (r, n) in (("small_kernel", (1.0, 7)), ("large_kernel", (10.0, 71))):
        k = O.gaussian_kernel(r, n)
        for direction, gfn, cfn in (("horizontal", A.horizontal_filter, fast.orc_horizontal_filter),
                                    ("vertical", A.vertical_filter, fast.orc_vertical_filter)):
            gfn(img, k, ctx)
            g, g_s = best_of(lambda: gfn(img, k, ctx), 20)
            co = np.empty_like(img)
            _, c_s = best_of(lambda: cfn(img.ctypes.data, w, h, k.ctypes.data, len(k), co.ctypes.data), 5)
            same = g.tobytes() == co.tobytes()
            out["rows"][f"{direction}_filter_{kname}"] = {"gpu_ms": round(g_s * 1e3, 3), "cpu_ms": round(c_s * 1e3, 3), "taps": n,
                                                          "bit_identical": bool(same), "reference": f"criterion.rs: gaussian_kernel({r}, {n})"}
            out["mismatches"] += int(not same)
    out["parity"] = {"mismatches": out["mismatches"]}
    return out
