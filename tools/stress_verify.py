#!/usr/bin/env python3
"""Randomised parity sweep of the batched two-view verification and (every third call) the batched registration
consensus (GPU box): random scene counts and sizes, outlier fractions, cameras, thresholds and ARRSAC rule sets —
rs_essential_arrsac_batch_device / rs_p3p_arrsac_batch_device vs oracle/arrsac_oracle.c (orc_arrsac_pairs /
orc_p3p_arrsac_pairs): bearings, scoring order, winner id, pose bits, inlier list, survivors / blocks / poses /
residuals evaluated must all be equal.  The oracle runs in a process pool.
usage: python tools/stress_verify.py [--rounds 12] [--seed 1] [--procs 32]"""
import argparse
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

STATS = np.dtype([("poses", "<u4"), ("survivors", "<u4"), ("blocks", "<u4"), ("reserved", "<u4"), ("evaluated", "<u8"),
                  ("exhaustive", "<u8")])


def make_round(r, seed):
    """One call's worth of scenes and parameters, a pure function of (r, seed)."""
    from test_gpu_parity import _pixel_scene
    rng = np.random.default_rng(seed * 104729 + r)
    cap = int(rng.choice([64, 200, 512, 1000]))
    S = int(rng.integers(1, 24))
    cam_a = (float(rng.uniform(600, 1200)), float(rng.uniform(600, 1200)), float(rng.uniform(300, 900)),
             float(rng.uniform(150, 500)), float(rng.choice([0.0, 0.3])), None)
    cam_b = cam_a if rng.random() < 0.5 else (cam_a[0] * 0.97, cam_a[1] * 1.02, cam_a[2] - 11.0, cam_a[3] + 7.0, 0.5,
                                              float(rng.choice([-0.05, 0.02])))
    bs = int(rng.choice([8, 16, 33, 64, 100]))
    kw = dict(block_size=bs, init_blocks=int(rng.integers(1, 4)), max_candidates=int(rng.choice([0, 1, 7, 48, 96, 1024])),
              sprt=bool(rng.integers(2)), halve=bool(rng.integers(2)), bound=bool(rng.integers(2)),
              estimations_per_block=int(rng.choice([0, 0, 5, 16])))
    if rng.random() < 0.3:
        kw.update(sprt_delta=0.02, sprt_ratio=200.0)
    n_hyp = int(rng.choice([8, 64, 192, 500, 1500]))
    thr = float(rng.choice([1e-9, 1e-7, 2e-7, 1e-6, 1e-4, 3e-3, 0.05, 0.2]))
    shuffle = bool(rng.integers(2))
    scenes = []
    for s in range(S):
        n = int(rng.choice([0, 3, 8, 9, bs, bs + 1, cap, int(rng.integers(0, cap + 1))]))
        n = min(n, cap)
        scenes.append(_pixel_scene(rng, cap, cap, n, float(rng.choice([0.0, 0.3, 0.7])), cam_a,
                                   noise_px=float(rng.choice([0.0, 0.3, 2.0]))))
    reg = None
    if r % 3 == 2:
        # every third call is the registration path: (feature, world point) pairs against one table of landmarks
        from test_oracle_arrsac import _registration_scene
        n_world = int(rng.integers(50, 1500))
        reg, worlds = [], []
        for s in range(S):
            n = min(int(rng.choice([0, 2, 3, 4, bs, bs + 1, cap, int(rng.integers(0, cap + 1))])), cap)
            kps, world, pr, _, _, _ = _registration_scene(rng, cap, n_world, n, float(rng.choice([0.0, 0.3, 0.7])), cam_b,
                                                          noise_px=float(rng.choice([0.0, 0.2, 2.0])))
            pr = pr.copy(); pr[:, 1] += s * n_world
            reg.append((kps, pr))
            worlds.append(world)
        reg = dict(scenes=reg, world=np.concatenate(worlds))
        thr = float(rng.choice([1e-7, 1e-6, 1e-4]))
    return dict(cap=cap, cam_a=cam_a, cam_b=cam_b, kw=kw, n_hyp=n_hyp, thr=thr, shuffle=shuffle, scenes=scenes,
                seed=int(rng.integers(1 << 40)), reg=reg)


def oracle_scene(args):
    r, seed, s = args
    from oracle import oracle as O
    R = make_round(r, seed)
    if R["reg"] is not None:
        kps, pr = R["reg"]["scenes"][s]
        w = O.p3p_arrsac_pairs(kps, pr, R["reg"]["world"], R["cam_b"], R["thr"], R["n_hyp"], scene=s, shuffle=R["shuffle"],
                               seed=R["seed"], **R["kw"])
        return r, s, w
    ka, kb, pr = R["scenes"][s]
    w = O.arrsac_pairs(ka, kb, pr, R["cam_a"], R["cam_b"], R["thr"], R["n_hyp"], scene=s, shuffle=R["shuffle"], seed=R["seed"],
                       **R["kw"])
    return r, s, w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--procs", type=int, default=32)
    a = ap.parse_args()
    from oracle import oracle as O
    O.build()
    rounds = [make_round(r, a.seed) for r in range(a.rounds)]
    jobs = [(r, a.seed, s) for r, R in enumerate(rounds) for s in range(len(R["scenes"]))]
    with mp.get_context("spawn").Pool(a.procs) as pool:
        want = {(r, s): w for r, s, w in pool.map(oracle_scene, jobs, chunksize=1)}
    import torch
    from cv_amd import build
    build.build()
    from cv_amd.ransac import EssentialConsensus
    dev = torch.device("cuda", 0)
    bad = 0
    models = 0
    for r, R in enumerate(rounds):
        cap, kw, S = R["cap"], R["kw"], len(R["scenes"])
        reg = R["reg"]
        pairs = np.zeros((S, cap, 2), np.uint32)
        if reg is not None:
            for s, (kps, pr) in enumerate(reg["scenes"]):
                pairs[s, :len(pr)] = pr
            kps_a = kps_b = np.stack([sc[0] for sc in reg["scenes"]])
            npairs = np.array([len(sc[1]) for sc in reg["scenes"]], np.uint32)
        else:
            for s, sc in enumerate(R["scenes"]):
                pairs[s, :len(sc[2])] = sc[2]
            kps_a = np.stack([sc[0] for sc in R["scenes"]]); kps_b = np.stack([sc[1] for sc in R["scenes"]])
            npairs = np.array([len(sc[2]) for sc in R["scenes"]], np.uint32)
        d_ka = torch.from_numpy(kps_a.view(np.uint8).reshape(S, cap, 28)).to(dev)
        d_kb = torch.from_numpy(kps_b.view(np.uint8).reshape(S, cap, 28)).to(dev)
        d_pairs = torch.from_numpy(pairs.view(np.int32)).to(dev)
        d_np = torch.from_numpy(npairs.view(np.int32)).to(dev)
        d_pose = torch.zeros((S, 12), dtype=torch.float64, device=dev)
        d_best = torch.zeros((S,), dtype=torch.int32, device=dev)
        d_inl = torch.zeros((S, cap), dtype=torch.int32, device=dev)
        d_ninl = torch.zeros((S,), dtype=torch.int32, device=dev)
        d_stats = torch.zeros((S, 32), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        blocks_max = (cap + kw["block_size"] - 1) // kw["block_size"]
        cons = EssentialConsensus(cap, R["n_hyp"] + kw["estimations_per_block"] * blocks_max)
        cons.reserve(S)
        prm = cons.make_params(R["thr"], n_hypotheses=R["n_hyp"], seed=R["seed"], **kw)
        ia = list(range(S))
        if reg is not None:
            d_world = torch.from_numpy(reg["world"]).to(dev)
            cons.p3p_model_inliers_batch_device(d_ka.data_ptr(), cap, ia, d_pairs.data_ptr(), d_np.data_ptr(), d_world.data_ptr(),
                                                d_world.shape[0], cons.camera(R["cam_b"]), prm, d_pose.data_ptr(), d_best.data_ptr(),
                                                d_inl.data_ptr(), d_ninl.data_ptr(), d_stats.data_ptr(), shuffle=R["shuffle"])
        else:
            cons.model_inliers_batch_device(d_ka.data_ptr(), d_kb.data_ptr(), cap, ia, ia, d_pairs.data_ptr(), d_np.data_ptr(),
                                            cons.camera(R["cam_a"]), cons.camera(R["cam_b"]), prm, d_pose.data_ptr(),
                                            d_best.data_ptr(), d_inl.data_ptr(), d_ninl.data_ptr(), d_stats.data_ptr(),
                                            shuffle=R["shuffle"])
        cons.sync()
        pose = d_pose.cpu().numpy(); best = d_best.cpu().numpy().view(np.uint32)
        inl = d_inl.cpu().numpy().view(np.uint32); ninl = d_ninl.cpu().numpy().view(np.uint32)
        st = d_stats.cpu().numpy().view(STATS).reshape(S)
        rbad = 0
        for s in range(S):
            w = want[(r, s)]
            if reg is not None:
                ga, gb, go = cons.scene_world(s, cap)
                ok = ga.tobytes() == w["bearings"].tobytes() and gb.tobytes() == w["world"].tobytes()
            else:
                ga, gb, go = cons.scene(s, cap)
                ok = ga.tobytes() == w["bearings_a"].tobytes() and gb.tobytes() == w["bearings_b"].tobytes()
            if R["shuffle"]:
                ok = ok and np.array_equal(go, w["order"])
            ok = ok and best[s] == w["best_id"] and ninl[s] == len(w["inliers"])
            if ok and w["best_id"] != 0xFFFFFFFF:
                models += 1
                ok = pose[s].tobytes() == np.ascontiguousarray(w["pose"]).tobytes() and np.array_equal(inl[s, :ninl[s]], w["inliers"])
                ok = ok and all(int(st[k][s]) == w["stats"][wk] for k, wk in
                                (("survivors", "survivors"), ("blocks", "blocks"), ("poses", "poses"), ("evaluated", "residuals_evaluated")))
            if not ok:
                rbad += 1
                print(f"MISMATCH round {r} scene {s}: n {npairs[s]} best {best[s]} / {w['best_id']} inliers {ninl[s]} / {len(w['inliers'])}",
                      flush=True)
        bad += rbad
        print(f"round {r}{' (registration)' if reg is not None else ''}: {S} scenes cap {cap} hyp {R['n_hyp']} thr {R['thr']:g} shuffle {int(R['shuffle'])} {kw} -> "
              f"{'ok' if not rbad else str(rbad) + ' BAD'}", flush=True)
    print(f"stress_verify seed {a.seed}: {len(jobs)} scenes in {a.rounds} calls, {models} with a model, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
