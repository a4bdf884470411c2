#!/usr/bin/env python3
"""bench.py — frames/s of AKAZE detect + describe + brute-force Hamming match on 1080p (BASELINE.json metric).

One step = one pass of the hot path over one batch of 256 synthetic 1920x1080 frames per GPU
(BASELINE.json configs[1], plus the consecutive-frame symmetric better-by-24 match of configs[2]'s
matcher): Akaze::default() extract of every frame, then frame g is matched against frame g-1.
Inputs are resident in HBM before the timed region.  Multi-GPU: one process per GPU, frames sharded
frame g -> rank g mod N; the only exchange is an RCCL all-gather of the fixed-capacity
descriptor blocks so that the owner of frame g holds frame g-1's descriptors.  Weak scaling: every rank
processes its own 256 frames per step.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      — the FED diffusion step kernel (calculate_step, the dominant kernel): algorithmic bytes per
                  launch (12 B per pixel-step x pixels x frames) / HIP-event time on the library's
                  stream, against the 8 TB/s HBM3E peak.
  cpu_baseline  — the CPU oracle (a restatement of the reference, kind "port") timed on this box's host
                  cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 1920, 1080
FRAMES_PER_STEP = 256
CAP = 8192              # descriptor block capacity per frame (cv-sfm tracking_features, settings.rs:433-434)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FED_BYTES_PER_PIXEL_STEP = 12.0
MFMA_I8_PEAK_TOPS = 3944.0   # dense int8 MFMA, measured ceiling in MI355X_MICROARCH.md (~2x the bf16 rate)
MFMA_FP4_PEAK_TOPS = 10000.0  # dense FP4/FP6 MFMA (MI355X_MICROARCH.md; AMD's 20 PF headline is 2:1 sparse)


def make_world(seed, w, h):
    """Deterministic synthetic 'world' canvas (value noise + rectangles + discs), uint8, numpy."""
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
    density = (w * h) / (1920.0 * 1080.0)
    n_shapes = int(200 * density)
    for _ in range(n_shapes):
        sw, sh = rng.integers(8, 97, 2)
        x, y = rng.integers(0, w), rng.integers(0, h)
        img[y:y + sh, x:x + sw] = rng.integers(0, 256)
    yy, xx = np.mgrid[0:97, 0:97]
    for _ in range(n_shapes):
        r = int(rng.integers(4, 49))
        x, y = int(rng.integers(r, w - r)), int(rng.integers(r, h - r))
        m = (yy[:2 * r + 1, :2 * r + 1] - r) ** 2 + (xx[:2 * r + 1, :2 * r + 1] - r) ** 2 <= r * r
        img[y - r:y + r + 1, x - r:x + r + 1][m] = rng.integers(0, 256)
    return np.clip(img, 0, 255).astype(np.uint8)


def make_frames(torch, device, rank, n_frames, world_size):
    """n_frames 1080p frames for this rank: a camera panning over the world canvas (4 px right, 2 px down
    per GLOBAL frame) plus +-2 sensor noise.  Global frame g = j*world_size + rank."""
    total = n_frames * world_size
    world = make_world(0xA4A2E, W + 4 * total + 64, H + 2 * total + 64)
    wt = torch.from_numpy(world).to(device)
    frames = torch.empty((n_frames, H, W), dtype=torch.uint8, device=device)
    gen = torch.Generator(device=device)
    for j in range(n_frames):
        g = j * world_size + rank
        gen.manual_seed(1000 + g)
        crop = wt[2 * g:2 * g + H, 4 * g:4 * g + W].to(torch.int16)
        noise = torch.randint(-2, 3, (H, W), generator=gen, device=device, dtype=torch.int16)
        frames[j] = (crop + noise).clamp_(0, 255).to(torch.uint8)
    return frames


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=FRAMES_PER_STEP, help="frames per GPU per step")
    ap.add_argument("--micro-batch", type=int, default=128, help="frames per kernel launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=6)
    ap.add_argument("--cpu-procs", type=int, default=64, help="host processes of the all-cores CPU baseline (0 = skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for the "
                    "single-GPU smoke test of the multi-rank path)")
    ap.add_argument("--share-device", action="store_true", help="all ranks use cuda:0 (smoke test of N>1 on one GPU)")
    ap.add_argument("--dump-matches", default=None, help="write per-global-frame keypoint/match counts to this .npy")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    assert torch.cuda.is_available(), "bench.py needs the MI355X (there is no CPU fallback)"
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from cv_amd import build
    if world > 1:               # one rank builds (normally a no-op: the .so travels with the tree), the rest wait
        if rank == 0:
            build.build()
        dist.barrier()
    else:
        build.build()
    from cv_amd import _lib
    from cv_amd.akaze import Akaze
    from cv_amd.knn import Matcher, RULE_STRICT
    from cv_amd.sharding import exchange_predecessors
    L = _lib.lib()

    NF, MB = args.frames, min(args.micro_batch, args.frames)
    assert NF % MB == 0
    frames = make_frames(torch, dev, rank, NF, world)

    ak = Akaze.default()
    ak.device = local_rank
    ak.max_keypoints = CAP
    ctx = ak.context(W, H, MB)
    matcher = Matcher(CAP, device=local_rank)
    akz_stream = torch.cuda.ExternalStream(L.akz_stream(ctx.handle), device=dev)
    hm_stream = torch.cuda.ExternalStream(L.hm_stream(matcher.handle), device=dev)

    # Every output exists twice and consecutive steps alternate between the two sets, so a step never has to
    # wait for the previous step's matcher before its extraction may overwrite descriptors: the stages of
    # neighbouring steps overlap exactly like the stages of neighbouring micro-batches inside a step.
    def zeros2(shape, dtype):
        return [torch.zeros(shape, dtype=dtype, device=dev) for _ in range(2)]
    kps2 = zeros2((NF, CAP, 28), torch.uint8)
    descs2 = zeros2((NF, CAP, 64), torch.uint8)
    counts2 = zeros2((NF,), torch.int32)
    # predecessor descriptor blocks: prev[j] = descriptors of global frame g-1 for local frame j
    prev_descs2 = zeros2((NF, CAP, 64), torch.uint8) if world > 1 else [None, None]
    prev_counts2 = zeros2((NF,), torch.int32) if world > 1 else [None, None]
    pairs2 = zeros2((NF + 2, CAP, 2), torch.int32)   # +2: a micro-batch can carry mb+1 pairs
    npairs2 = zeros2((NF + 2,), torch.int32)
    match_done = [torch.cuda.Event(), torch.cuda.Event()]   # the matcher finished reading output set p
    step_no = [0]
    host_trace = [] if os.environ.get("AKZ_BENCH_TRACE") else None   # (step, m0, ms in extract call, ms in match call)
    if world > 1:
        gath_d = torch.zeros((world, MB, CAP, 64), dtype=torch.uint8, device=dev)
        gath_n = torch.zeros((world, MB), dtype=torch.int32, device=dev)
    # match problems are issued per micro-batch so the VALU-bound matcher of micro-batch m overlaps the
    # HBM-bound scale space of micro-batch m+1: frame j pairs with frame j-1; frame 0 pairs with the step's
    # last frame once that exists.
    def idx(vals):
        return (C.c_uint32 * len(vals))(*vals)

    # multi-rank: the descriptor exchange runs on its own stream, so that the next micro-batch's scale space
    # (which waits on the caller's stream only) does not queue behind the collective of this one
    comm = torch.cuda.Stream(device=dev) if world > 1 else None

    def step():
        cur = torch.cuda.current_stream()
        p = step_no[0] & 1
        kps, descs, counts, pairs, npairs = kps2[p], descs2[p], counts2[p], pairs2[p], npairs2[p]
        prev_descs, prev_counts = prev_descs2[p], prev_counts2[p]
        if step_no[0] >= 2:          # set p was last read by the matcher two steps ago (long finished)
            cur.wait_event(match_done[p])
            if world > 1:
                comm.wait_event(match_done[p])
        for m0 in range(0, NF, MB):
            tA = time.perf_counter()
            _lib.check(L.akz_extract_batch_device(
                ctx.handle, frames[m0:m0 + MB].data_ptr(), 0, MB, W, H, kps[m0:m0 + MB].data_ptr(),
                descs[m0:m0 + MB].data_ptr(), CAP, counts[m0:m0 + MB].data_ptr(), cur.cuda_stream), "extract")
            tB = time.perf_counter()
            js = [j for j in range(m0, m0 + MB) if j > 0]
            if m0 + MB == NF:
                js.append(0)
            if world > 1:
                comm.wait_stream(akz_stream)
                with torch.cuda.stream(comm):
                    js = exchange_predecessors(dist, rank, world, m0, MB, NF, descs[m0:m0 + MB],
                                               counts[m0:m0 + MB], gath_d, gath_n, prev_descs, prev_counts)
                ia, ib, tb, nb, wait = idx(js), idx(js), prev_descs, prev_counts, comm
            else:
                ia, ib, tb, nb, wait = idx(js), idx([(j - 1) % NF for j in js]), descs, counts, akz_stream
            # problem p writes pairs/npairs block p of the view starting at js[0]'s slot; keep them per frame
            _lib.check(L.hm_match_batch_device(
                matcher.handle, descs.data_ptr(), counts.data_ptr(), tb.data_ptr(), nb.data_ptr(), CAP, ia, ib,
                len(js), RULE_STRICT, 24, 0.0, 1, pairs[m0:].data_ptr(), npairs[m0:].data_ptr(), wait.cuda_stream),
                "match")
            if host_trace is not None:
                host_trace.append((step_no[0], m0, round((tB - tA) * 1e3, 2), round((time.perf_counter() - tB) * 1e3, 2)))
        match_done[p].record(hm_stream)
        step_no[0] += 1

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.timing_enable(True)
    ctx.timing_reset()
    _lib.check(L.hm_timing_get(matcher.handle, None, None, 1), "hm_timing_get")
    _lib.check(L.hm_timing_enable(matcher.handle, 1), "hm_timing_enable")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    last = (step_no[0] - 1) & 1
    kps, descs, counts, pairs, npairs = kps2[last], descs2[last], counts2[last], pairs2[last], npairs2[last]
    knn_ms, knn_launches = C.c_double(), C.c_uint64()
    _lib.check(L.hm_timing_get(matcher.handle, C.byref(knn_ms), C.byref(knn_launches), 1), "hm_timing_get")
    _lib.check(L.hm_timing_enable(matcher.handle, 0), "hm_timing_enable")
    _lib.check(L.akz_sync(ctx.handle), "akz_sync")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    fed_ms, fed_launches, fed_units = ctx.timing_get(0)
    ss_ms, _, _ = ctx.timing_get(1)
    all_ms, _, _ = ctx.timing_get(2)
    ctx.timing_enable(False)

    # Isolated pass for the roofline: the same FED launches with nothing else on the GPU (in the timed region
    # above they share the chip with the keypoint and matcher streams of neighbouring micro-batches).
    iso = None
    if rank == 0:
        barrier_local = torch.cuda.synchronize
        barrier_local()
        ctx.timing_enable(True)
        ctx.timing_reset()
        for _ in range(3):
            _lib.check(L.akz_scale_space_device(ctx.handle, frames[:MB].data_ptr(), 0, MB, W, H, None), "scale_space")
            _lib.check(L.akz_sync(ctx.handle), "akz_sync")
        i_ms, i_launches, i_units = ctx.timing_get(0)
        s_ms, _, _ = ctx.timing_get(1)
        ctx.timing_enable(False)
        if i_ms > 0:
            iso = {"achieved": round(FED_BYTES_PER_PIXEL_STEP * i_units / (i_ms * 1e-3) / 1e9, 1),
                   "avg_launch_us": round(i_ms * 1e3 / max(1, i_launches), 2),
                   "scale_space_frames_per_s": round(3 * MB / (s_ms * 1e-3), 1)}
            iso["frac"] = round(iso["achieved"] / HBM_PEAK_GBS, 4)
    if world > 1:
        dist.barrier()

    if args.dump_matches:
        # per GLOBAL frame g = j*world + rank: [keypoints, matches of (g, g-1)]; problem order follows `js`
        # within each micro-batch, so the match count of local frame j is looked up through the same schedule
        per_frame = torch.zeros((NF, 2), dtype=torch.int32, device=dev)
        per_frame[:, 0] = counts
        slot = 0
        for m0 in range(0, NF, MB):
            if world > 1:
                if rank > 0:
                    js = list(range(m0, m0 + MB))
                else:
                    js = list(range(m0 + 1, m0 + MB)) + ([m0] if m0 > 0 else []) + ([0] if m0 + MB == NF else [])
            else:
                js = [j for j in range(m0, m0 + MB) if j > 0] + ([0] if m0 + MB == NF else [])
            for q, j in enumerate(js):
                per_frame[j, 1] = npairs[m0 + q]
        allf = [torch.zeros_like(per_frame) for _ in range(world)] if world > 1 else [per_frame]
        if world > 1:
            dist.all_gather(allf, per_frame)
        if rank == 0:
            glob = np.zeros((NF * world, 2), np.int32)
            for r in range(world):
                glob[r::world] = allf[r].cpu().numpy()
            np.save(args.dump_matches, glob)

    n_kp = counts.float().mean().item()
    n_match = npairs[:NF].float().mean().item()
    if rank == 0 and host_trace is not None:
        print("host ms per call (step, m0, extract, match):", host_trace, file=sys.stderr)
    if rank == 0:
        total_frames = NF * world * args.steps
        fps = total_frames / elapsed
        fed_bytes = FED_BYTES_PER_PIXEL_STEP * fed_units
        achieved = fed_bytes / (fed_ms * 1e-3) / 1e9 if fed_ms > 0 else 0.0
        out = {
            "metric": "frames/sec AKAZE detect+describe+BF-Hamming-match, 1080p",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: batch of 256 synthetic 1920x1080 frames per GPU, "
                                   "Akaze::default() detect+describe, + symmetric better-by-24 BF Hamming match "
                                   "of consecutive frames", "frames_per_gpu_per_step": NF, "micro_batch": MB,
                       "parallelism": f"frame-sharded x{world}", "mean_keypoints_per_frame": round(n_kp, 1),
                       "mean_matches_per_pair": round(n_match, 1)},
            "roofline": {"bound": "hbm", "kernel": "k_fed_pair<T> (calculate_step, two frames per block, up to 8 steps per launch)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic_per_launch(MB),
                         "launches": int(fed_launches),
                         "avg_launch_us": round(fed_ms * 1e3 / max(1, fed_launches), 2),
                         "algorithmic_bytes_per_launch": round(fed_bytes / max(1, fed_launches), 0),
                         "note": "achieved = 12 B x pixel-steps / HIP-event time of the FED launches inside the timed "
                                 "region (they share the GPU with the other two streams); temporal blocking moves "
                                 "fewer HBM bytes than the 12 B/pixel-step contract figure, see traffic",
                         "isolated": iso},
            "phase_ms_per_step": {"fed": round(fed_ms / args.steps, 2), "scale_space": round(ss_ms / args.steps, 2),
                                  "extract": round(all_ms / args.steps, 2)},
        }
        if knn_ms.value > 0:
            # second roofline, for the matcher: 2 directions x nq x nt x 512-bit contractions per frame pair as
            # int8 MACs (2 ops each) over the HIP-event time of the k_knn_mfma launches on the matcher's stream
            pairs_total = NF * args.steps
            macs = 2.0 * pairs_total * (n_kp ** 2) * 512.0
            tops = 2.0 * macs / (knn_ms.value * 1e-3) / 1e12
            fp4 = os.environ.get("AKZ_MATCH_FP4", "1") != "0"
            peak = MFMA_FP4_PEAK_TOPS if fp4 else MFMA_I8_PEAK_TOPS
            out["roofline_matcher"] = {
                "bound": "mfma",
                "kernel": "k_knn_mfma4<2> (v_mfma_scale_f32_32x32x64_f8f6f4, E2M1 operands)" if fp4
                          else "k_knn_mfma<2> (v_mfma_i32_32x32x32_i8)",
                "achieved": round(tops, 1), "peak": peak, "unit": "TOP/s", "frac": round(tops / peak, 4),
                "launches": int(knn_launches.value),
                "avg_launch_us": round(knn_ms.value * 1e3 / max(1, knn_launches.value), 2),
                "note": "ops = 2 x 512 MACs per (query, target) pair with the mean keypoint count; peak = the dense "
                        "FP4 MFMA figure of MI355X_MICROARCH.md (~10 PF; micro-benchmark ceiling 9.1 PF)" if fp4 else
                        "ops = 2 x 512 int8 MACs per (query, target) pair with the mean keypoint count; peak = the "
                        "int8 micro-benchmark ceiling of MI355X_MICROARCH.md (no spec figure is listed for dense I8)"}
        rf = out["roofline"]
        if rf["traffic"] and rf["avg_launch_us"]:
            # what the memory system actually moved (PMC) over the same launch time: the number to hold against
            # the 8 TB/s peak; `achieved` above counts the contract's 12 B per pixel-step and exceeds the peak
            # because up to 8 steps share one pass over HBM
            rf["hbm_side_gbs"] = round(rf["traffic"] / (rf["avg_launch_us"] * 1e-6) / 1e9, 1)
            rf["hbm_side_frac"] = round(rf["hbm_side_gbs"] / HBM_PEAK_GBS, 4)
        out["device"] = device_probe(torch, dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames, args.cpu_frames)
            if args.cpu_procs > 0:
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(frames, args.cpu_procs)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def device_probe(torch, dev):
    """What the numbers were measured on: device name, CU count, memory, and a
    device-to-device copy probe (read + write of 1 GiB, best of 5) as the practical HBM ceiling of this box."""
    p = torch.cuda.get_device_properties(dev)
    n = 1 << 28
    a = torch.empty(n, dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    best = 0.0
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        e1.synchronize()
        best = max(best, 2.0 * 4 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del a, b
    return {"name": p.name, "gcn_arch": getattr(p, "gcnArchName", ""), "compute_units": p.multi_processor_count,
            "hbm_gib": round(p.total_memory / 2**30, 1), "d2d_copy_gbs": round(best, 1),
            "hbm_peak_gbs_spec": HBM_PEAK_GBS}


def pmc_traffic_per_launch(mb=None):
    """HBM bytes per FED launch from the committed rocprofv3 PMC passes (profiles/r01_pmc_traffic.json:
    (2 x FETCH_SIZE + WRITE_SIZE) KiB summed over the k_fed_pair dispatches of one micro-batch / launches;
    FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction for 16-byte coalesced reads)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            fed = json.load(f)["fed"]
        per_frame = fed["hbm_bytes_per_launch"] / fed["frames_per_launch"]   # a launch covers a whole micro-batch
        return round(per_frame * (mb or fed["frames_per_launch"]))
    except Exception:
        return None


def cpu_baseline(frames, n):
    """The CPU oracle (a C restatement of the reference's akaze crate + BF matcher; kind 'port') on the
    first n frames of the same workload, single thread, on this box's host cores."""
    from oracle import oracle as O
    n = max(2, min(n, frames.shape[0]))
    host = frames[:n].cpu().numpy()
    orc = O.Akaze(W, H, O.default_config())
    t0 = time.perf_counter()
    prev = None
    nk = 0
    for i in range(n):
        kp, d = orc.extract(host[i])
        nk += len(d)
        if prev is not None:
            O.match(d, prev, rule=O.RULE_STRICT, param_u=24, symmetric=True)
        prev = d
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
            "host_cores_available": os.cpu_count(),
            "sample": f"first {n} frames of the bench batch: Akaze::default() extract + symmetric match vs previous "
                      f"frame, single thread, {dt:.1f} s, {nk // n} keypoints/frame"}


def _cpu_worker(args):
    """One frame pair on one core: extract both frames with the oracle, match them symmetrically."""
    a, b = args
    from oracle import oracle as O
    orc = O.Akaze(W, H, O.default_config())
    _, da = orc.extract(a)
    _, db = orc.extract(b)
    O.match(db, da, rule=O.RULE_STRICT, param_u=24, symmetric=True)
    return len(da) + len(db)


def cpu_baseline_all_cores(frames, procs):
    """The same oracle on `procs` host cores at once (one process per frame pair, nothing shared): what the
    reference's per-frame parallelism (cv-sfm processes frames independently) could reach on this host."""
    import multiprocessing as mp
    procs = max(1, min(procs, os.cpu_count() or 1, frames.shape[0] // 2))
    host = frames[:2 * procs].cpu().numpy()
    tasks = [(host[2 * i], host[2 * i + 1]) for i in range(procs)]
    ctx = mp.get_context("spawn")            # the parent holds a HIP context: never fork it
    with ctx.Pool(procs) as pool:
        pool.map(_cpu_worker, tasks[:1])     # warm the workers' imports (oracle build check, page-in)
        t0 = time.perf_counter()
        pool.map(_cpu_worker, tasks, chunksize=1)
        dt = time.perf_counter() - t0
    return {"value": round(2 * procs / dt, 2), "unit": "frames/s", "cores": procs, "kind": "port",
            "sample": f"{procs} processes x 2 frames (extract both, one symmetric match), {dt:.1f} s"}


if __name__ == "__main__":
    main()
