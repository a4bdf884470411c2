#!/usr/bin/env python3
"""Throughput of rs_essential_arrsac_batch_device (or, with --registration, rs_p3p_arrsac_batch_device) alone: S synthetic frame pairs (pixel keypoints of a rigid motion
seen through a pinhole camera, `--matches` pairs each, a fraction joined at random), vslam-sandbox's consensus
parameters (8192 initialisation hypotheses, 1024 candidates; vslam-sandbox/src/main.rs:112-117).  Prints one JSON line.

  python tools/bench_verify.py [--scenes 256] [--matches 4400] [--hyp 8192] [--block 16] [--check 2]
  python tools/bench_verify.py --registration --cap 2048 --matches 1500 --hyp 16384 --block 64 --resample 256 --thr 1e-5
      (the single-view consensus of vslam-sandbox/src/main.rs:104-110: 16 384 hypotheses, 1 024 candidates, 256 estimations
       per block, threshold cv-sfm/src/settings.rs:352-355)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def make_scene(rng, cap, n_pairs, outlier_frac, cam, noise_px):
    from cv_amd._lib import KP_DTYPE
    from test_oracle_ransac import _rot
    R = _rot((rng.random(3) - 0.5) * 0.1)
    t = (rng.random(3) - 0.5) * 0.3
    fx, fy, cx, cy, skew = cam[:5]

    def project(P):
        x, y = P[:, 0] / P[:, 2], P[:, 1] / P[:, 2]
        return np.stack([fx * x + skew * y + cx, fy * y + cy], 1)
    pts = np.stack([rng.uniform(-3, 3, cap), rng.uniform(-1.7, 1.7, cap), rng.uniform(3, 12, cap)], 1)
    ka = np.zeros(cap, KP_DTYPE); kb = np.zeros(cap, KP_DTYPE)
    pa = project(pts) + rng.standard_normal((cap, 2)) * noise_px
    perm = rng.permutation(cap)
    pb = project(pts[perm] @ R.T + t) + rng.standard_normal((cap, 2)) * noise_px
    ka["x"], ka["y"], kb["x"], kb["y"] = pa[:, 0], pa[:, 1], pb[:, 0], pb[:, 1]
    ib = rng.choice(cap, size=n_pairs, replace=False)
    ia = perm[ib].copy()
    bad = rng.random(n_pairs) < outlier_frac
    ia[bad] = rng.integers(0, cap, bad.sum())
    o = np.argsort(ia, kind="stable")
    return ka, kb, np.stack([ia[o], ib[o]], 1).astype(np.uint32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=256)
    ap.add_argument("--matches", type=int, default=4400)
    ap.add_argument("--cap", type=int, default=8192)
    ap.add_argument("--hyp", type=int, default=8192)
    ap.add_argument("--block", type=int, default=16)
    ap.add_argument("--init-blocks", type=int, default=1)
    ap.add_argument("--candidates", type=int, default=1024)
    ap.add_argument("--no-halve", action="store_true")
    ap.add_argument("--no-sprt", action="store_true")
    ap.add_argument("--outliers", type=float, default=0.1)
    ap.add_argument("--noise", type=float, default=0.1)
    ap.add_argument("--thr", type=float, default=1e-7)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--check", type=int, default=2, help="scenes compared with oracle/arrsac_oracle.c")
    ap.add_argument("--distinct", type=int, default=16, help="distinct synthetic scenes (repeated to fill --scenes)")
    ap.add_argument("--registration", action="store_true", help="the P3P registration consensus instead of the two-view one")
    ap.add_argument("--resample", type=int, default=0, help="estimations_per_block")
    args = ap.parse_args()

    import torch
    from cv_amd import build
    build.build()
    from cv_amd.ransac import EssentialConsensus
    dev = torch.device("cuda", 0)
    S, cap = args.scenes, args.cap
    cam = (1000.0, 1000.0, 960.0, 540.0, 0.0, None)
    rng = np.random.default_rng(0xFE11)
    nd = min(args.distinct, S)
    n_world = 4 * cap
    if args.registration:
        from test_oracle_arrsac import _registration_scene
        base, worlds = [], []
        for i in range(nd):
            kps, world, pr, _, _, _ = _registration_scene(rng, cap, n_world, args.matches, args.outliers, cam, args.noise)
            pr = pr.copy(); pr[:, 1] += i * n_world
            base.append((kps, kps, pr))
            worlds.append(world)
        world_all = np.concatenate(worlds)
        d_world = torch.from_numpy(world_all).to(dev)
    else:
        base = [make_scene(rng, cap, args.matches, args.outliers, cam, args.noise) for _ in range(nd)]
    kps_a = np.stack([b[0] for b in base]); kps_b = np.stack([b[1] for b in base])
    pairs = np.zeros((S, cap, 2), np.uint32)
    for s in range(S):
        pairs[s, :args.matches] = base[s % nd][2]
    ia = ib = [s % nd for s in range(S)]
    d_ka = torch.from_numpy(kps_a.view(np.uint8).reshape(nd, cap, 28)).to(dev)
    d_kb = torch.from_numpy(kps_b.view(np.uint8).reshape(nd, cap, 28)).to(dev)
    d_pairs = torch.from_numpy(pairs.view(np.int32)).to(dev)
    d_np = torch.full((S,), args.matches, dtype=torch.int32, device=dev)
    d_pose = torch.zeros((S, 12), dtype=torch.float64, device=dev)
    d_best = torch.zeros((S,), dtype=torch.int32, device=dev)
    d_inl = torch.zeros((S, cap), dtype=torch.int32, device=dev)
    d_ninl = torch.zeros((S,), dtype=torch.int32, device=dev)
    d_stats = torch.zeros((S, 32), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    cons = EssentialConsensus(cap, args.hyp + args.resample * ((cap + args.block - 1) // args.block))
    cons.reserve(S)
    kw = dict(block_size=args.block, init_blocks=args.init_blocks, max_candidates=args.candidates, halve=not args.no_halve,
              sprt=not args.no_sprt, estimations_per_block=args.resample)
    prm = cons.make_params(args.thr, n_hypotheses=args.hyp, seed=0, **kw)
    c = cons.camera(cam)

    def run():
        if args.registration:
            cons.p3p_model_inliers_batch_device(d_ka.data_ptr(), cap, ia, d_pairs.data_ptr(), d_np.data_ptr(), d_world.data_ptr(), d_world.shape[0], c, prm,
                                                d_pose.data_ptr(), d_best.data_ptr(), d_inl.data_ptr(), d_ninl.data_ptr(),
                                                d_stats.data_ptr(), shuffle=True)
            return
        cons.model_inliers_batch_device(d_ka.data_ptr(), d_kb.data_ptr(), cap, ia, ib, d_pairs.data_ptr(), d_np.data_ptr(), c, c, prm,
                                        d_pose.data_ptr(), d_best.data_ptr(), d_inl.data_ptr(), d_ninl.data_ptr(), d_stats.data_ptr(),
                                        shuffle=True)
    run()
    cons.sync()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        run()
    cons.sync()
    dt = (time.perf_counter() - t0) / args.reps
    ninl = d_ninl.cpu().numpy().view(np.uint32)
    st = d_stats.cpu().numpy().view(np.dtype([("poses", "<u4"), ("survivors", "<u4"), ("blocks", "<u4"), ("reserved", "<u4"),
                                               ("evaluated", "<u8"), ("exhaustive", "<u8")])).reshape(S)
    out = {"what": "registration (P3P)" if args.registration else "two-view (eight-point)", "scenes": S, "matches": args.matches, "hypotheses": args.hyp, "params": kw, "ms_per_call": round(dt * 1e3, 3),
           "pairs_per_s": round(S / dt, 1), "hypotheses_per_s": round(S * args.hyp / dt, 1),
           "residuals_per_s": round(float(st["evaluated"].sum()) / dt, 1),
           "mean_inliers": round(float(ninl.mean()), 1), "mean_blocks": float(st["blocks"].mean()),
           "residuals_evaluated_frac": round(float(st["evaluated"].sum()) / float(st["exhaustive"].sum()), 5)}
    if args.check:
        from oracle import oracle as O
        pose = d_pose.cpu().numpy(); best = d_best.cpu().numpy().view(np.uint32); inl = d_inl.cpu().numpy().view(np.uint32)
        bad = 0
        t0 = time.perf_counter()
        for s in np.linspace(0, S - 1, args.check).astype(int):
            ka, kb, pr = base[s % nd]
            if args.registration:
                w = O.p3p_arrsac_pairs(ka, pr, world_all, cam, args.thr, args.hyp, scene=int(s), shuffle=True, seed=0, **kw)
            else:
                w = O.arrsac_pairs(ka, kb, pr, cam, cam, args.thr, args.hyp, scene=int(s), shuffle=True, seed=0, **kw)
            bad += int(best[s] != w["best_id"] or pose[s].tobytes() != w["pose"].tobytes()
                       or not np.array_equal(inl[s, :ninl[s]], w["inliers"]))
        out["parity"] = {"scenes_checked": int(args.check), "mismatches": bad, "cpu_s_per_scene": round((time.perf_counter() - t0) / args.check, 2)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
