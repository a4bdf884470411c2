#!/usr/bin/env python3
"""bench.py — frames/s of AKAZE detect + describe + brute-force Hamming match on 1080p (BASELINE.json metric).

One step = one pass of the hot path over one batch of 256 synthetic 1920x1080 frames per GPU
(BASELINE.json configs[1], plus the consecutive-frame symmetric better-by-24 match of configs[2]'s
matcher): Akaze::default() extract of every frame, then frame g is matched against frame g-1.
Inputs are resident in HBM before the timed region.  Multi-GPU: one process per GPU, frames sharded
frame g -> rank g mod N; the only exchange is a ring shift of the fixed-capacity descriptor
blocks (RCCL send/recv to rank + 1) so that the owner of frame g holds frame g-1's descriptors.  Weak scaling: every rank
processes its own 256 frames per step.

Rank 0 writes the full report (every object named below, ~20 KB) to gpurun_out/bench_detail.json and prints, as the LAST
and only JSON line of stdout, a compact headline (< 4 KB: `headline()` below) that carries the contract fields plus one
`roofline` object, one `cpu_baseline` object, the parity counts and one scalar per extra leg.  The report's objects:
  roofline        — the kernel family with the largest time per step when the GPU is its alone (isolated pass), timed
                    in the run by the launches' own start/stop events: achieved = the kernel's algorithmic bytes
                    (what it must read and write once, given what it fuses: DESIGN.md §5; for the gather kernel the
                    distinct 32-byte sectors its keypoints touch) / its event time, against the 8 TB/s HBM3E peak, so
                    frac <= 1 by construction; `traffic` = PMC HBM bytes per launch from the committed rocprofv3
                    counter passes when they were taken at this micro-batch, else null.
  roofline_top    — the same for the five most expensive kernel families, the matcher (MFMA ops / 10 PF) among them.
  algorithmic_gbs — SURVEY §8d's contract figure (1.0535 GB per 1080p frame) x the isolated scale-space rate: it
                    counts every named pyramid buffer once per consuming stage and therefore exceeds what the fused
                    kernels move; kept for continuity, never used as a roofline fraction.
  parity_checked  — the GPU keypoints / descriptors / match pairs of the cpu_baseline frames compared with what the
                    oracle just computed for them (the run FAILS, rc 1, on any mismatch).
  configs_extra   — BASELINE configs[2] (1 000 x 5 000 Bernoulli descriptors, 999 consecutive pairs) and configs[3]
                    (10 000 eight-point hypotheses on a 1 000-match scene), each oracle-checked on a sample, each with
                    its own roofline triple; the pipeline+verify and pipeline+register legs; the criterion rows.
  cpu_baseline    — the CPU oracle (a restatement of the reference, kind "port") timed on this box's host
                    cores on a bounded sample of the same workload (rank 0, N=1 only).
`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches its own N ranks
(torch.distributed.run on 127.0.0.1); under torch.distributed.run it is one of them.
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
    ap.add_argument("--micro-batch", type=int, default=256, help="frames per library call (measured: 64 -> 6794, 128 -> 6912, 256 -> 6997 frames/s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true", help="A/B: timed region without the per-launch start/stop events "
                    "(no roofline objects)")
    ap.add_argument("--no-isolated", action="store_true", help="skip the isolated scale-space pass (rocprof runs whose kernel "
                    "statistics are to be compared with this line's roofline: every launch is then the pipelined workload's)")
    ap.add_argument("--pmc-run", action="store_true", help="counter passes: exactly --steps steps of the pipeline and nothing "
                    "else on the GPU (no instrumented pass, no isolated pass, no CPU baseline, no extras)")
    ap.add_argument("--cpu-frames", type=int, default=6)
    ap.add_argument("--parity-pairs", type=int, default=5, help="frame pairs from other batch positions held to the oracle "
                    "(2 oracle extractions each, ~1 s per 1080p frame)")
    ap.add_argument("--cpu-procs", type=int, default=64, help="OpenMP threads of the all-cores CPU baseline (0 = skip; each "
                    "thread holds a ~0.4 GB pyramid)")
    ap.add_argument("--no-pipeline", action="store_true", help="one buffer set in the library (AKZ_OPT_NO_PIPELINE): "
                    "consecutive calls do not overlap; for counter passes and serial phase profiles")
    ap.add_argument("--opt", action="append", default=[], help="akz_options field for the context, key=value (A/B runs; "
                    "the defaults are what the headline is quoted on)")
    ap.add_argument("--max-features", type=int, default=0, help="A/B: Akaze.maximum_features (0 = the reference's default, unlimited)")
    ap.add_argument("--matcher-low-priority", action="store_true", help="(the default since round 5: accepted, no effect)")
    ap.add_argument("--matcher-normal-priority", action="store_true", help="A/B: matcher stream at the default priority instead of the lowest "
                    "(measured 8 644 / 8 644 against 8 702 / 8 686 frames/s with the lowest)")
    ap.add_argument("--matcher-cus", type=int, default=0, help="A/B: matcher stream on the last N compute units of every XCD (0 = all)")
    ap.add_argument("--no-extras", action="store_true", help="skip configs_extra (BASELINE configs[2] and [3], pipeline+verify)")
    ap.add_argument("--verify-steps", type=int, default=4, help="timed steps of the pipeline+verify leg (extract + match + "
                    "two-view ARRSAC of every frame pair, device-resident); 0 = skip")
    ap.add_argument("--verify-block", type=int, default=16, help="pipeline+verify: matches per scoring block")
    ap.add_argument("--verify-check", type=int, default=16, help="pipeline+verify: scenes compared with oracle/arrsac_oracle.c")
    ap.add_argument("--register-steps", type=int, default=3, help="timed steps of the pipeline+register leg (extract + hash_bag + "
                    "knn(., 3) against the recent views + best-of-views + Lambda Twist ARRSAC of every frame, device-resident); 0 = skip")
    ap.add_argument("--register-views", type=int, default=32, help="pipeline+register: recent views per frame (cv-sfm tracking_recent_frames)")
    ap.add_argument("--register-check", type=int, default=4, help="pipeline+register: frames compared with the oracle")
    ap.add_argument("--extra-frames", type=int, default=1000, help="frames of the configs[2] matcher workload")
    ap.add_argument("--extra-hyp", type=int, default=10000, help="hypotheses of the configs[3] scene")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for the "
                    "single-GPU smoke test of the multi-rank path)")
    ap.add_argument("--recent", type=int, default=1, help="every frame is matched against its K predecessors g-1 .. g-K "
                    "(cv-sfm tracking_recent_frames: up to 32); 1 = the headline workload (symmetric match with g-1), "
                    "K > 1: LinearKnn::knn(., 2) of every feature against each of the K views (hm_knn_batch_device: one call per step)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "shift", "allgather"], help="how descriptor blocks reach the "
                    "ranks that match against them: ring shift (K = 1) or all-gather of the fixed-capacity blocks (K > 1)")
    ap.add_argument("--comm", default="auto", choices=["auto", "akz", "torch"], help="akz: the library's own RCCL exchange "
                    "(akz_comm_*, C ABI); torch: torch.distributed.  auto = akz with the nccl backend, torch otherwise")
    ap.add_argument("--force-exchange", action="store_true", help="run the exchange code path even with one rank (a rank "
                    "then sends to itself): the single-GPU test of the N > 1 path's collectives")
    ap.add_argument("--share-device", action="store_true", help="all ranks use cuda:0 (smoke test of N>1 on one GPU)")
    ap.add_argument("--detail-stdout", action="store_true", help="also print the full report (one {\"bench_detail\": ...} line) before "
                    "the headline; by default it only goes to gpurun_out/bench_detail.json")
    ap.add_argument("--dump-matches", default=None, help="write per-global-frame keypoint/match counts to this .npy")
    args = ap.parse_args()
    if args.pmc_run:
        args.no_cpu_baseline = args.no_extras = args.no_isolated = True
    if args.share_device and args.gpus > 1 and args.backend == "nccl":
        args.backend = "gloo"         # RCCL refuses two ranks on one device; the smoke test of N > 1 on one GPU runs over gloo

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args, sys.argv[1:])          # (does not return: exec of torch.distributed.run with the same arguments)
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs the MI355X (there is no CPU fallback)"
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded = world > 1 or args.force_exchange      # the exchange step runs (with one rank: to itself)
    comm_kind = args.comm if args.comm != "auto" else ("akz" if args.backend == "nccl" else "torch")
    K = max(1, args.recent)
    use_allgather = args.exchange == "allgather" or (args.exchange == "auto" and K > 1)
    if world > 1 or (sharded and comm_kind == "torch"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
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
    from cv_amd.sharding import AkzExchange, TorchExchange, exchange_predecessors, gathered_block, pred_row, window_views
    L = _lib.lib()

    NF, MB = args.frames, min(args.micro_batch, args.frames)
    assert NF % MB == 0
    frames = make_frames(torch, dev, rank, NF, world)

    ak = Akaze.default()
    ak.device = local_rank
    ak.max_keypoints = CAP
    if args.max_features:                     # A/B: a truncating call takes the (level, tile) visiting order + its sort
        ak.maximum_features = args.max_features
    okw = {}
    for kv in args.opt:                       # A/B runs: akz_options fields by name (cv_amd._lib.make_options)
        key, val = kv.split("=")
        okw[key] = val if key == "contrast" else (bool(int(val)) if key in _lib.BOOL_OPTIONS else int(val))
    if args.no_pipeline:
        okw["pipeline"] = False
    ctx = ak.context(W, H, MB, options=_lib.make_options(**okw) if okw else None)
    matcher = Matcher(CAP, device=local_rank, low_priority=not args.matcher_normal_priority, cus=args.matcher_cus)
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
    # (row NF holds the predecessor of local frame 0 on rank 0: cv_amd/sharding.py)
    shift_mode = sharded and not use_allgather
    prev_descs2 = zeros2((NF + 1, CAP, 64), torch.uint8) if shift_mode else [None, None]
    prev_counts2 = zeros2((NF + 1,), torch.int32) if shift_mode else [None, None]
    # all-gather mode: gathered[m][r][i] = block of local frame m*MB + i of rank r (cv_amd/sharding.py: gathered_block)
    gath_descs2 = zeros2((NF // MB, world, MB, CAP, 64), torch.uint8) if (sharded and use_allgather) else [None, None]
    gath_counts2 = zeros2((NF // MB, world, MB), torch.int32) if (sharded and use_allgather) else [None, None]
    knn_out2 = zeros2((NF, K, CAP, 2, 2), torch.int32) if K > 1 else [None, None]     # [frame][view][query][k]{index, distance}
    pairs2 = zeros2((NF + 2, CAP, 2), torch.int32)   # +2: a micro-batch can carry mb+1 pairs
    npairs2 = zeros2((NF + 2,), torch.int32)
    match_done = [torch.cuda.Event(), torch.cuda.Event()]   # the matcher finished reading output set p
    step_no = [0]
    host_trace = [] if os.environ.get("AKZ_BENCH_TRACE") else None   # (step, m0, ms in extract call, ms in match call)
    # match problems are issued per micro-batch so the VALU-bound matcher of micro-batch m overlaps the
    # HBM-bound scale space of micro-batch m+1: frame j pairs with frame j-1; frame 0 pairs with the step's
    # last frame once that exists.
    def idx(vals):
        return (C.c_uint32 * len(vals))(*vals)

    # multi-rank: the descriptor exchange (a ring shift: every rank sends its block to rank + 1 and receives its
    # predecessor's straight into the rows the matcher reads) runs on its own stream, so that the next micro-batch's
    # scale space (which waits on the caller's stream only) does not queue behind it; ordering contract in
    # cv_amd/sharding.py
    comm = torch.cuda.Stream(device=dev) if sharded else None
    exchange = None
    comm_note = None
    if sharded:
        if comm_kind == "akz":
            # every rank must end up on the same route: agree on whether the library's own RCCL exchange came up everywhere
            try:
                exchange = AkzExchange(dist, rank, world, local_rank)
                ok = 1
            except Exception as e:          # librccl not loadable, communicator refused ...
                exchange, ok, comm_note = None, 0, f"akz_comm unavailable ({e}); torch.distributed used instead"
            if world > 1:
                flag = torch.tensor([ok], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag.item())
            if not ok:
                if exchange is not None:
                    exchange.close()
                comm_note = comm_note or "akz_comm unavailable on another rank; torch.distributed used instead"
                comm_kind = "torch"
                if not dist.is_initialized():
                    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                    os.environ.setdefault("MASTER_PORT", "29511")
                    dist.init_process_group(args.backend, rank=rank, world_size=world, **({"device_id": dev} if args.backend == "nccl" else {}))
                exchange = TorchExchange(dist, rank, world)
        else:
            exchange = TorchExchange(dist, rank, world)
        flush_c_stdio()

    verify = {"on": None}            # pipeline+verify leg: a callable(p, m0, js, prev_js) that enqueues the consensus

    def step():
        cur = torch.cuda.current_stream()
        p = step_no[0] & 1
        kps, descs, counts, pairs, npairs = kps2[p], descs2[p], counts2[p], pairs2[p], npairs2[p]
        prev_descs, prev_counts = prev_descs2[p], prev_counts2[p]
        if step_no[0] >= 2:          # set p was last read by the matcher two steps ago (long finished)
            cur.wait_event(match_done[p])
            if sharded:
                comm.wait_event(match_done[p])
            if verify["on"] is not None and verify.get("armed"):
                cur.wait_event(verify["done"][p])   # ... and its keypoints / pair lists by the consensus
        for m0 in range(0, NF, MB):
            tA = time.perf_counter()
            _lib.check(L.akz_extract_batch_device(
                ctx.handle, frames[m0:m0 + MB].data_ptr(), 0, MB, W, H, kps[m0:m0 + MB].data_ptr(),
                descs[m0:m0 + MB].data_ptr(), CAP, counts[m0:m0 + MB].data_ptr(), cur.cuda_stream), "extract")
            tB = time.perf_counter()
            js = [j for j in range(m0, m0 + MB) if j > 0]
            if m0 + MB == NF:
                js.append(0)
            if K > 1:
                # window mode: the micro-batch's blocks go to every rank (all-gather); matching follows the step's last one
                if sharded:
                    comm.wait_stream(akz_stream)
                    with torch.cuda.stream(comm):
                        exchange.allgather(descs[m0:m0 + MB], counts[m0:m0 + MB], gath_descs2[p][m0 // MB], gath_counts2[p][m0 // MB])
                continue
            if shift_mode:
                comm.wait_stream(akz_stream)
                with torch.cuda.stream(comm):
                    js = exchange_predecessors(dist, rank, world, m0, MB, NF, descs[m0:m0 + MB],
                                               counts[m0:m0 + MB], prev_descs, prev_counts, exchange)
                ia, ib, tb, nb, wait = idx(js), idx([pred_row(rank, j, NF) for j in js]), prev_descs, prev_counts, comm
            elif sharded:
                # K = 1 through the all-gather: the predecessor's block is looked up in the gathered array
                comm.wait_stream(akz_stream)
                with torch.cuda.stream(comm):
                    exchange.allgather(descs[m0:m0 + MB], counts[m0:m0 + MB], gath_descs2[p][m0 // MB], gath_counts2[p][m0 // MB])
                if m0 + MB < NF:
                    continue             # (frame 0's predecessor is the step's last frame: match once everything is gathered)
                js = list(range(NF))
                ia = idx(js)
                ib = idx([gathered_block(window_views(rank, j, world, NF, 1)[0], world, NF, MB) for j in js])
                tb, nb, wait = gath_descs2[p], gath_counts2[p], comm
                m0 = 0                   # the pair lists of all frames, slot = frame
            else:
                ia, ib, tb, nb, wait = idx(js), idx([(j - 1) % NF for j in js]), descs, counts, akz_stream
            # problem p writes pairs/npairs block p of the view starting at js[0]'s slot; keep them per frame
            _lib.check(L.hm_match_batch_device(
                matcher.handle, descs.data_ptr(), counts.data_ptr(), tb.data_ptr(), nb.data_ptr(), CAP, ia, ib,
                len(js), RULE_STRICT, 24, 0.0, 1, pairs[m0:].data_ptr(), npairs[m0:].data_ptr(), wait.cuda_stream),
                "match")
            if verify["on"] is not None:
                verify["on"](p, m0, js, [(j - 1) % NF for j in js])
            if host_trace is not None:
                host_trace.append((step_no[0], m0, round((tB - tA) * 1e3, 2), round((time.perf_counter() - tB) * 1e3, 2)))
        if K > 1:
            # every feature of frame j against each of its K recent views (cv-sfm/src/lib.rs:1468-1486), 2 neighbours each
            views_d = gath_descs2[p] if sharded else descs
            views_n = gath_counts2[p] if sharded else counts
            wait = comm if sharded else akz_stream
            iq, it = [], []
            for j in range(NF):
                gv = window_views(rank, j, world, NF, K)
                iq += [j] * K
                it += [gathered_block(g, world, NF, MB) for g in gv] if sharded else gv
            for p0 in range(0, len(iq), 32768):          # (a call takes up to 65 535 problems)
                p1 = min(len(iq), p0 + 32768)
                _lib.check(L.hm_knn_batch_device(matcher.handle, descs.data_ptr(), counts.data_ptr(), views_d.data_ptr(),
                                                 views_n.data_ptr(), CAP, idx(iq[p0:p1]), idx(it[p0:p1]), p1 - p0, 2,
                                                 knn_out2[p].view(-1, CAP, 2, 2)[p0:].data_ptr(),
                                                 wait.cuda_stream if p0 == 0 else None), "knn_batch")
        match_done[p].record(hm_stream)
        step_no[0] += 1

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    # The timed region runs with the KERNEL timers on (akz_timing_enable 2, hm_timing_enable): every timed launch carries
    # its own start / stop events (hipExtLaunchKernel — the dispatch's begin and end timestamps, what rocprofv3's kernel
    # trace reports), no extra packet enters a stream; --no-kernel-timers is the A/B switch.  The PHASE timers are event
    # brackets on the streams and stay off here: an untimed instrumented pass of two more steps reads them.
    kt = not args.no_kernel_timers
    ctx.timing_enable(2 if kt else 0)
    ctx.timing_reset()
    _lib.check(L.hm_timing_get(matcher.handle, None, None, 1), "hm_timing_get")
    _lib.check(L.hm_timing_enable(matcher.handle, 1 if kt else 0), "hm_timing_enable")
    if exchange is not None:
        exchange.exposed()               # reset the exchange's own timer
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    my_elapsed = elapsed
    exch = exchange.exposed() if exchange is not None else None
    knn_ms, knn_launches = C.c_double(), C.c_uint64()
    _lib.check(L.hm_timing_get(matcher.handle, C.byref(knn_ms), C.byref(knn_launches), 1), "hm_timing_get")
    _lib.check(L.hm_timing_enable(matcher.handle, 0), "hm_timing_enable")
    fam_pipe = read_families(ctx)                 # kernel families over the timed region (all streams busy)
    desc_k_ms, desc_k_launches, _ = ctx.timing_get(25)
    INSTR_STEPS = 0 if args.pmc_run else 2
    fed_ms = ss_ms = all_ms = desc_ms = refine_ms = 0.0
    if INSTR_STEPS:
        ctx.timing_enable(1)
        ctx.timing_reset()
        step()
        step()                          # two steps: both output sets, so `last` below still names the newest one
        barrier()
        fed_ms, _, _ = ctx.timing_get(0)
        ss_ms, _, _ = ctx.timing_get(1)
        all_ms, _, _ = ctx.timing_get(2)
        desc_ms, _, _ = ctx.timing_get(11)
        refine_ms, _, _ = ctx.timing_get(12)
    last = (step_no[0] - 1) & 1
    kps, descs, counts, pairs, npairs = kps2[last], descs2[last], counts2[last], pairs2[last], npairs2[last]
    _lib.check(L.akz_sync(ctx.handle), "akz_sync")
    per_rank = [my_elapsed]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank = [float(x.item()) for x in allt]
        elapsed = max(per_rank)
    ctx.timing_enable(False)

    # Isolated pass: the library's launches with nothing else on the GPU (in the timed region above three streams share the
    # chip).  One whole extraction per repetition — scale space, then the keypoint stage of the same call — with a device
    # synchronisation in between, into the output set the parity checks do NOT read; then the matcher alone on that set.
    fam_iso, iso_fps, iso_match = None, None, None
    ISO_REPS = 3
    if rank == 0 and not args.no_isolated and not args.pmc_run:
        torch.cuda.synchronize()
        spare = 1 - last
        ctx.timing_enable(1)
        ctx.timing_reset()
        for _ in range(ISO_REPS):
            for m0 in range(0, NF, MB):
                _lib.check(L.akz_extract_batch_device(ctx.handle, frames[m0:m0 + MB].data_ptr(), 0, MB, W, H, kps2[spare][m0:m0 + MB].data_ptr(),
                                                      descs2[spare][m0:m0 + MB].data_ptr(), CAP, counts2[spare][m0:m0 + MB].data_ptr(), None), "extract")
                _lib.check(L.akz_sync(ctx.handle), "akz_sync")
        fam_iso = read_families(ctx)
        s_ms, _, _ = ctx.timing_get(1)
        ctx.timing_enable(False)
        iso_fps = ISO_REPS * NF / (s_ms * 1e-3) if s_ms > 0 else None
        if K == 1 and not sharded:
            js_all = list(range(NF))
            _lib.check(L.hm_timing_get(matcher.handle, None, None, 1), "hm_timing_get")
            _lib.check(L.hm_timing_enable(matcher.handle, 1), "hm_timing_enable")
            for _ in range(ISO_REPS):
                _lib.check(L.hm_match_batch_device(matcher.handle, descs2[spare].data_ptr(), counts2[spare].data_ptr(), descs2[spare].data_ptr(),
                                                   counts2[spare].data_ptr(), CAP, idx(js_all), idx([(j - 1) % NF for j in js_all]), NF, RULE_STRICT,
                                                   24, 0.0, 1, pairs2[spare].data_ptr(), npairs2[spare].data_ptr(), None), "match")
                _lib.check(L.hm_sync(matcher.handle), "hm_sync")
            im, il_ = C.c_double(), C.c_uint64()
            _lib.check(L.hm_timing_get(matcher.handle, C.byref(im), C.byref(il_), 1), "hm_timing_get")
            _lib.check(L.hm_timing_enable(matcher.handle, 0), "hm_timing_enable")
            iso_match = (im.value, int(il_.value))
    if world > 1:
        dist.barrier()

    if args.dump_matches:
        # per GLOBAL frame g = j*world + rank: [keypoints, matches of (g, g-1)]; problem order follows `js`
        # within each micro-batch, so the match count of local frame j is looked up through the same schedule
        slot_of = {}
        for m0 in range(0, NF, MB):
            if sharded and not shift_mode:
                js, base = list(range(m0, m0 + MB)), None        # all-gather route: slot = frame
            elif sharded and rank > 0:
                js, base = list(range(m0, m0 + MB)), m0
            elif sharded:
                js, base = list(range(m0 + 1, m0 + MB)) + ([m0] if m0 > 0 else []) + ([0] if m0 + MB == NF else []), m0
            else:
                js, base = [j for j in range(m0, m0 + MB) if j > 0] + ([0] if m0 + MB == NF else []), m0
            for q, j in enumerate(js):
                slot_of[j] = j if base is None else base + q
        per_frame = torch.zeros((NF, 2), dtype=torch.int32, device=dev)
        per_frame[:, 0] = counts
        if K == 1:
            for j in range(NF):
                per_frame[j, 1] = npairs[slot_of[j]]
        allf = [torch.zeros_like(per_frame) for _ in range(world)] if world > 1 else [per_frame]
        if world > 1:
            dist.all_gather(allf, per_frame)
        if rank == 0:
            glob = np.zeros((NF * world, 2), np.int32)
            for r in range(world):
                glob[r::world] = allf[r].cpu().numpy()
            np.save(args.dump_matches, glob)
        # the pair lists (K = 1) or the neighbour lists against the K views, one file per rank, keyed by GLOBAL frame
        if K == 1:
            hp, hn = pairs.cpu().numpy(), npairs.cpu().numpy()
            np.savez(f"{args.dump_matches}.r{rank}.npz",
                     **{f"g{j * world + rank}": hp[slot_of[j], :hn[slot_of[j]]].copy() for j in range(NF)})
        else:
            hk, hc = knn_out2[last].cpu().numpy(), counts.cpu().numpy()
            np.savez(f"{args.dump_matches}.r{rank}.npz",
                     **{f"g{j * world + rank}": hk[j, :, :hc[j]].copy() for j in range(NF)})

    n_kp = counts.float().mean().item()
    n_match = npairs[:NF].float().mean().item()
    if rank == 0 and host_trace is not None:
        print("host ms per call (step, m0, extract, match):", host_trace, file=sys.stderr)
    if rank == 0:
        total_frames = NF * world * args.steps
        fps = total_frames / elapsed
        gather = None
        try:
            gs = sorted({0, NF // 3, NF // 2, NF - 1})
            hk = [kps[j].cpu().numpy().view(_lib.KP_DTYPE).reshape(-1) for j in gs]
            gather = gather_model(ctx, hk, [int(counts[j].item()) for j in gs])
        except Exception as e:                 # the model is reporting only: never fail the run for it
            print(f"bench.py: gather model skipped ({e})", file=sys.stderr)
        tops = roofline_entries(fam_pipe, fam_iso, MB, args.steps, gather, ISO_REPS)
        if knn_ms.value > 0:
            # the matcher as a family of its own: 2 directions x nq x nt x 512-bit contractions per frame pair as MACs (2 ops each)
            ops_pair = 2.0 * 2.0 * (n_kp ** 2) * 512.0
            tops_m = ops_pair * NF * args.steps / (knn_ms.value * 1e-3) / 1e12
            em = {"bound": "mfma", "kernel": "k_knn_mfma4w<2> (v_mfma_scale_f32_32x32x64_f8f6f4, E2M1 operands, 64 resident queries per wave, target tiles by LDS-DMA: exact 2-NN of every descriptor, both directions)",
                  "achieved": round(tops_m, 1), "peak": MFMA_FP4_PEAK_TOPS, "unit": "TOP/s", "frac": round(tops_m / MFMA_FP4_PEAK_TOPS, 4),
                  "frac_is": "2 x 512 MACs per (query, target) pair / kernel time / the 10 PF dense FP4 MFMA peak",
                  "peak_check": "the instruction alone sustains 9 870 TOP/s on this part (tools/ubench/mfma_fp4_rate.hip, "
                                "profiles/r04_mfma_fp4_rate.txt); with 4 / 8 integer VALU instructions behind every MFMA — the key "
                                "epilogue's share is ~4 — the same loop gives 5 880 / 4 140: FP4 MFMA and VALU do not overlap at two "
                                "waves per SIMD, so the kernel's ceiling is MFMA time + epilogue time, ~0.58 of the peak",
                  "traffic": None, "launches": int(knn_launches.value), "avg_launch_us": round(knn_ms.value * 1e3 / max(1, knn_launches.value), 2),
                  "gpu_ms": round(knn_ms.value, 2), "gpu_ms_per_step": round(knn_ms.value / args.steps, 3),
                  "timed": "the k-NN launches' own start/stop events over the timed steps; the matcher's stream has the lowest priority, so "
                           "inside the pipeline its kernels are the ones time-sliced (3-4x their isolated duration)"}
            em["rank_ms_per_step"] = em["gpu_ms_per_step"]
            if iso_match and iso_match[0] > 0:
                it_ = ops_pair * NF * ISO_REPS / (iso_match[0] * 1e-3) / 1e12
                em["isolated"] = {"achieved": round(it_, 1), "frac": round(it_ / MFMA_FP4_PEAK_TOPS, 4),
                                  "avg_launch_us": round(iso_match[0] * 1e3 / max(1, iso_match[1]), 2),
                                  "gpu_ms_per_step": round(iso_match[0] / ISO_REPS, 3)}
                em["rank_ms_per_step"] = em["isolated"]["gpu_ms_per_step"]
            tops.append(em)
        tops.sort(key=lambda e: -e["rank_ms_per_step"])
        traffic = pipeline_traffic(MB, NF)
        out = {
            "metric": "frames/sec AKAZE detect+describe+BF-Hamming-match, 1080p",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: batch of 256 synthetic 1920x1080 frames per GPU (ONE panning "
                                   "camera over a single seeded world canvas, 4 px right / 2 px down per frame, +-2 sensor "
                                   "noise per frame — not SURVEY 8d's per-frame seeds: consecutive frames must overlap for "
                                   "the match to mean something), Akaze::default() detect+describe, + symmetric "
                                   "better-by-24 BF Hamming match of consecutive frames", "frames_per_gpu_per_step": NF, "micro_batch": MB,
                       "parallelism": f"frame-sharded x{world}", "mean_keypoints_per_frame": round(n_kp, 1),
                       "mean_matches_per_pair": round(n_match, 1),
                       "library_options": okw or "defaults"},
            "roofline": tops[0] if tops else None,
            "roofline_top": tops[:5],
            "roofline_rule": "roofline = the kernel family with the largest time per step when the GPU is its alone (isolated pass: "
                             "rank_ms_per_step); roofline_top = the five largest, the matcher among them.  frac is always the family's own "
                             "roofline fraction inside the timed pipeline (bytes / 8 TB/s, or MFMA ops / 10 PF for the matcher), "
                             "isolated.frac the same with nothing else running; valu_frac (VALU issue) and bound beside it",
        }
        if INSTR_STEPS:
            out["phase_ms_per_step"] = {"fed": round(fed_ms / INSTR_STEPS, 2), "scale_space": round(ss_ms / INSTR_STEPS, 2),
                                        "extract": round(all_ms / INSTR_STEPS, 2), "describe": round(desc_ms / INSTR_STEPS, 2),
                                        "refine": round(refine_ms / INSTR_STEPS, 2),
                                        "note": "HIP-event brackets on the library's streams (wall time of a phase, waits for the "
                                                "other streams included) in an instrumented pass of 2 steps run after the timed "
                                                "region (same pipelined workload)"}
        if iso_fps:
            out["scale_space_isolated"] = {
                "frames_per_s": round(iso_fps, 1),
                "algorithmic_gbs": round(iso_fps * CONTRACT_BYTES_PER_FRAME / 1e9, 1),
                "note": "configs[1] 'scale-space kernels only' (A1-A11: the scale-space phase of whole extractions run one at a time, "
                        "nothing else on the GPU); algorithmic_gbs = "
                        "frames/s x SURVEY 8d's 1.0535 GB contract figure, which counts every pyramid buffer once per "
                        "consuming stage: fused kernels and temporal blocking move less, so it is NOT a roofline "
                        "fraction (see roofline / roofline_top for those)"}
        if knn_ms.value > 0:
            # the matcher's roofline: 2 directions x nq x nt x 512-bit contractions per frame pair as MACs (2 ops
            # each) over the HIP-event time of the k-NN launches on the matcher's stream
            pairs_total = NF * args.steps
            macs = 2.0 * pairs_total * (n_kp ** 2) * 512.0
            tops_m = 2.0 * macs / (knn_ms.value * 1e-3) / 1e12
            out["roofline_matcher"] = {
                "bound": "mfma", "kernel": "k_knn_mfma4w<2> (v_mfma_scale_f32_32x32x64_f8f6f4, E2M1 operands)",
                "achieved": round(tops_m, 1), "peak": MFMA_FP4_PEAK_TOPS, "unit": "TOP/s",
                "frac": round(tops_m / MFMA_FP4_PEAK_TOPS, 4), "traffic": None, "launches": int(knn_launches.value),
                "avg_launch_us": round(knn_ms.value * 1e3 / max(1, knn_launches.value), 2),
                "note": "ops = 2 x 512 MACs per (query, target) pair with the mean keypoint count; time = the k-NN "
                        "launches' own start/stop events over the timed region; peak = the dense FP4 MFMA figure "
                        "of MI355X_MICROARCH.md (~10 PF)"}
        if sharded:
            out["multi_gpu"] = {
                "ranks": world,
                "per_rank_frames_per_s": [round(NF * args.steps / t_, 1) for t_ in per_rank],
                "recent_views": K, "exchange": "all-gather of fixed-capacity descriptor blocks" if use_allgather else "ring shift",
                "comm": "akz_comm_* (libakz -> librccl.so.1)" if comm_kind == "akz" else f"torch.distributed ({args.backend})",
                "block_bytes_per_rank_per_step": NF * (CAP * 64 + 4),
                # did RCCL see N ranks?  akz_comm_world() of the library's own communicator (None on the torch.distributed route)
                "rccl_ranks_seen": exchange.ranks_seen() if hasattr(exchange, "ranks_seen") else None}
            if comm_note:
                out["multi_gpu"]["comm_note"] = comm_note
            if exch:
                out["multi_gpu"]["exchange_ms_per_step"] = round(exch[0] / args.steps, 3)
                out["multi_gpu"]["exchange_bytes_per_step"] = int(exch[2] // max(1, args.steps))
                out["multi_gpu"]["note"] = ("exchange_ms_per_step = HIP-event time of the transfers on the exchange stream (rank 0); "
                                            "they overlap the next micro-batch's scale space, so ms_per_step loses less than that")
        if K > 1:
            out["config"]["recent_views"] = K
            out["config"]["workload"] += f"; WINDOW MODE: 2-NN of every feature against each of the {K} preceding frames instead of the symmetric match"
        if traffic:
            out["hbm_traffic_per_frame"] = traffic
            out["end_to_end_hbm_frac"] = round(traffic["bytes"] * fps / world / (HBM_PEAK_GBS * 1e9), 4)
            # the same for the VALU: wave-instructions of the library's kernels per frame x 64 lanes x frames/s against the issue peak
            out["end_to_end_valu_frac"] = round(traffic["valu_insts"] * 64.0 * fps / world / (VALU_ISSUE_PEAK_T * 1e12), 4)
        out["device"] = device_probe(torch, dev)
        rc = 0
        if world == 1 and not args.no_cpu_baseline:
            base, oracle_out = cpu_baseline(frames, args.cpu_frames)
            out["cpu_baseline"] = base
            # the benchmarked configuration under the oracle: what the GPU produced for those same frames in the
            # last timed step (default options, micro-batch MB, pipelined) vs what the oracle just computed
            out["parity_checked"] = parity_check(oracle_out, kps, descs, counts, pairs, npairs, MB)
            if out["parity_checked"]["mismatches"]:
                rc = 1
            # ... and frame pairs from other positions of the batch (middle, odd, the last one of every micro-batch)
            out["parity_checked_spread"] = parity_spread(frames, kps, descs, counts, pairs, npairs, NF, MB, args.parity_pairs)
            if out["parity_checked_spread"]["mismatches"]:
                rc = 1
            if args.cpu_procs > 0:
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(frames, args.cpu_procs)
                out["cpu_baseline_intra_frame"] = cpu_baseline_intra_frame(frames, oracle_out, args.cpu_procs)
        if world == 1 and not args.no_extras:
            out["configs_extra"] = {"configs[2]": extra_match(torch, dev, L, _lib, args.extra_frames),
                                    "configs[3]": extra_ransac(args.extra_hyp)}
            if args.verify_steps > 0:
                out["configs_extra"]["pipeline+verify"] = extra_pipeline_verify(
                    torch, dev, L, _lib, args, step, step_no, barrier, verify, match_done, hm_stream, kps2, pairs2, npairs2, NF, MB)
            out["configs_extra"]["criterion"] = extra_criterion(_lib)
            if args.register_steps > 0 and K == 1 and not sharded:
                out["configs_extra"]["pipeline+register"] = extra_pipeline_register(torch, dev, L, _lib, args, ctx, frames, NF, MB)
            for v in out["configs_extra"].values():
                if v.get("parity", {}).get("mismatches"):
                    rc = 1
        flush_c_stdio()                     # (RCCL writes a version banner through C stdio: it must not follow the line)
        detail = write_detail(out)
        if args.detail_stdout:
            print(json.dumps({"bench_detail": out}), flush=True)
        print(headline(out, detail), flush=True)        # the LAST line of stdout, < 4 KB
        if rc:
            print("bench.py: GPU output differs from the oracle (see parity_checked / configs_extra)", file=sys.stderr)
            sys.exit(1)
    if world > 1:
        dist.destroy_process_group()


HEADLINE_LIMIT = 4096      # bytes: the driver keeps a bounded tail of stdout; a 22 KB line (round 4) was not parseable from it


def _short_roofline(e):
    """One flat roofline object for the headline: the contract's keys + the kernel's own launch statistics."""
    if not e:
        return None
    r = {k: e.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_launch_us")}
    r["kernel"] = str(e.get("kernel", "")).split(" ")[0]
    for k in ("valu_frac", "gpu_ms_per_step"):
        if e.get(k) is not None:
            r[k] = e[k]
    iso = e.get("isolated") or {}
    if iso.get("frac") is not None:
        r["isolated_frac"] = iso["frac"]
        r["isolated_avg_launch_us"] = iso.get("avg_launch_us")
    return r


def headline(out, detail_path=None):
    """The compact last line (< HEADLINE_LIMIT bytes): the contract fields of the task statement, ONE roofline object,
    ONE cpu_baseline object, parity counts and one scalar per extra leg.  Everything else stays in the detail file."""
    cfg = out.get("config", {})
    h = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                 "scaling", "vs_baseline", "dtype", "data")}
    h["config"] = {k: cfg[k] for k in ("workload", "frames_per_gpu_per_step", "micro_batch", "parallelism", "mean_keypoints_per_frame",
                                       "mean_matches_per_pair", "recent_views") if k in cfg}
    h["roofline"] = _short_roofline(out.get("roofline"))
    tops = out.get("roofline_top") or []
    if tops:      # [kernel, frac in the pipeline, frac alone, ms per step alone]
        h["roofline_top"] = [[str(e.get("kernel", "")).split(" ")[0], e.get("frac"), (e.get("isolated") or {}).get("frac"),
                              e.get("rank_ms_per_step")] for e in tops[:5]]
    for k in ("end_to_end_hbm_frac", "end_to_end_valu_frac"):
        if k in out:
            h[k] = out[k]
    iso = out.get("scale_space_isolated")
    if iso:
        h["scale_space_isolated_frames_per_s"] = iso.get("frames_per_s")
    cb = out.get("cpu_baseline")
    if cb:
        h["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind")}
        h["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:160]
        ac = out.get("cpu_baseline_all_cores")
        if ac:
            h["cpu_baseline"]["all_cores"] = {"value": ac.get("value"), "cores": ac.get("cores")}
    pc, ps = out.get("parity_checked"), out.get("parity_checked_spread")
    if pc:
        h["parity"] = {"frames": pc.get("frames", 0) + (ps or {}).get("frames", 0), "pairs": pc.get("pairs", 0) + (ps or {}).get("pairs", 0),
                       "mismatches": pc.get("mismatches", 0) + (ps or {}).get("mismatches", 0), "vs": "oracle/ (bit patterns)"}
    ex = out.get("configs_extra") or {}
    legs = {}
    pick = (("configs[2]", "pairs_per_s"), ("configs[3]", "hypotheses_per_s"), ("pipeline+verify", "verified_pairs_per_s"),
            ("pipeline+register", "registered_frames_per_s"))
    for leg, key in pick:
        if leg in ex:
            e = {key: ex[leg].get(key), "mismatches": (ex[leg].get("parity") or {}).get("mismatches")}
            rf = (ex[leg].get("roofline") or {}).get("frac")
            if rf is not None:
                e["roofline_frac"] = rf
            if "ms_per_step" in ex[leg]:
                e["ms_per_step"] = ex[leg]["ms_per_step"]
            legs[leg] = e
    if "criterion" in ex:
        legs["criterion"] = {"extract_gpu_ms": ex["criterion"].get("rows", {}).get("extract", {}).get("gpu_ms"),
                             "mismatches": ex["criterion"].get("mismatches")}
    if legs:
        h["extras"] = legs
    mg = out.get("multi_gpu")
    if mg:
        h["multi_gpu"] = {k: mg[k] for k in ("ranks", "exchange", "comm", "rccl_ranks_seen", "exchange_ms_per_step", "recent_views",
                                             "per_rank_frames_per_s") if k in mg}
    if detail_path:
        h["detail"] = detail_path
    line = json.dumps(h, separators=(",", ":"))
    for drop in ("roofline_top", "multi_gpu", "extras"):      # never exceed the limit: shed the optional objects first
        if len(line) < HEADLINE_LIMIT:
            break
        h.pop(drop, None)
        line = json.dumps(h, separators=(",", ":"))
    if len(line) >= HEADLINE_LIMIT:
        h["config"]["workload"] = h["config"].get("workload", "")[:200]
        line = json.dumps(h, separators=(",", ":"))
    assert len(line) < HEADLINE_LIMIT, len(line)
    return line


def write_detail(out):
    """The full report as a side file (gpurun_out/ merges back from the GPU box); returns the path relative to the repo."""
    rel = os.path.join("gpurun_out", "bench_detail.json")
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, rel), "w") as f:
            json.dump(out, f)
            f.write("\n")
        return rel
    except OSError as e:
        print(f"bench.py: detail file not written ({e})", file=sys.stderr)
        return None


def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: become the launcher of N ranks of this same command
    on this node (127.0.0.1 rendezvous, a free port), one rank per GPU — or all on cuda:0 with --share-device."""
    import socket
    if not args.share_device:
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s) (--share-device puts all ranks on cuda:0 "
                             "for a smoke test)")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.stdout.flush()
    os.execve(sys.executable, cmd, env)


def flush_c_stdio():
    try:
        C.CDLL(None).fflush(None)
    except Exception:
        pass


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
    profiles/, taken at this micro-batch); `bound` names whichever of the two is larger.  Ordered by a family's time per step
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
                if e["valu_frac"] > e["hbm_frac"]:
                    e["bound"] = "valu"          # (frac stays the bytes fraction; valu_frac is beside it)
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


def parity_check(oracle_out, kps, descs, counts, pairs, npairs, mb):
    """GPU outputs of the first len(oracle_out) frames of the last timed step vs the oracle's (keypoints as raw
    bytes, descriptor bytes, symmetric better-by-24 match pairs of (frame j, frame j-1))."""
    n = len(oracle_out)
    cnt = counts[:n].cpu().numpy()
    bad = []
    for j, (okp, od, om) in enumerate(oracle_out):
        c = int(cnt[j])
        gk = kps[j, :min(c, CAP)].cpu().numpy().tobytes()
        gd = descs[j, :min(c, CAP)].cpu().numpy()
        if c != len(okp) or gk != okp.tobytes() or not np.array_equal(gd, od):
            bad.append(f"frame {j}: keypoints/descriptors ({c} vs {len(okp)})")
        if om is not None and j < mb:          # problem j-1 of micro-batch 0 is (frame j, frame j-1)
            m = int(npairs[j - 1].item())
            gp = pairs[j - 1, :m].cpu().numpy().astype(np.uint32)
            if m != len(om) or not np.array_equal(gp, om.astype(np.uint32)):
                bad.append(f"pair ({j},{j - 1}): matches ({m} vs {len(om)})")
    return {"frames": n, "pairs": sum(1 for o in oracle_out if o[2] is not None), "mismatches": len(bad),
            "what": "keypoint bytes, descriptor bytes and symmetric match pairs of the cpu_baseline frames: last timed "
                    "step's GPU output (default options, pipelined micro-batches) vs the oracle", "detail": bad[:4]}


def parity_spread(frames, kps, descs, counts, pairs, npairs, NF, MB, n_pairs):
    """Frame pairs (j-1, j) spread over the batch — odd and even positions, the last frame of every micro-batch — the
    last timed step's GPU keypoints / descriptors / pair list of each against the oracle."""
    from oracle import oracle as O
    if n_pairs <= 0 or NF < 8:
        return {"frames": 0, "pairs": 0, "mismatches": 0}
    cand = sorted({NF - 1, NF // 2, (NF // 4) | 1, (3 * NF // 4) | 1, NF // 3} | {m0 + MB - 1 for m0 in range(0, NF, MB)})
    cand = [j for j in cand if j >= 7][:n_pairs]           # (frames 0..5 are the cpu_baseline sample's)
    orc = O.Akaze(W, H, O.default_config())
    cnt = counts.cpu().numpy()
    cache, bad = {}, []

    def ext(j):
        if j not in cache:
            cache[j] = orc.extract(frames[j].cpu().numpy())
            okp, od = cache[j]
            c = int(cnt[j])
            if c != len(okp) or kps[j, :min(c, CAP)].cpu().numpy().tobytes() != okp.tobytes() or \
                    not np.array_equal(descs[j, :min(c, CAP)].cpu().numpy(), od):
                bad.append(f"frame {j}: keypoints/descriptors ({c} vs {len(okp)})")
        return cache[j]
    for j in cand:
        (_, dp), (_, dj) = ext(j - 1), ext(j)
        om = O.match(dj, dp, rule=O.RULE_STRICT, param_u=24, symmetric=True).astype(np.uint32)
        m0 = (j // MB) * MB
        slot = j - 1 if m0 == 0 else j                      # position of frame j's problem in its micro-batch's list
        m = int(npairs[slot].item())
        gp = pairs[slot, :m].cpu().numpy().astype(np.uint32)
        if m != len(om) or not np.array_equal(gp, om):
            bad.append(f"pair ({j},{j - 1}): matches ({m} vs {len(om)})")
    return {"frames": len(cache), "pairs": len(cand), "positions": cand, "mismatches": len(bad), "detail": bad[:4],
            "what": "as parity_checked, for frame pairs at other positions of the batch"}


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
                                        shuffle=True, stream_to_wait=hm_stream.cuda_stream)
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
                                                  cur.cuda_stream), "extract")
        # the caller's bookkeeping: which landmark every feature of every stored view observes
        glue.wait_stream(akz_s)
        with torch.cuda.stream(glue):
            k = kps2[p].view(torch.float32).view(NF, CAP, 7)
            cx = torch.clamp(torch.floor((k[..., 0] + 4.0 * gidx) / CELL), 0, wc - 1).to(torch.int32)
            cy = torch.clamp(torch.floor((k[..., 1] + 2.0 * gidx) / CELL), 0, hc - 1).to(torch.int32)
            cls = kps2[p].view(torch.int32).view(NF, CAP, 7)[..., 6] & 15
            lm2[p].copy_((cy * wc + cx) * 16 + cls)
        reg.enqueue(kps2[p], descs2[p], counts2[p], frame_blocks, view_blocks, lm2[p], d_world, n_world, stream_to_wait=glue.cuda_stream)
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


def cpu_baseline(frames, n):
    """The CPU oracle (a C restatement of the reference's akaze crate + BF matcher; kind 'port') on the first n frames of
    the same workload, single thread, on this box's host cores.  Two builds of the same sources: the -O2 CHECKER computes
    what the GPU output is held to (returned: keypoints, descriptors, match pairs per frame); the -O3 -march=native build
    (oracle/Makefile `fast`) is the one that is TIMED, after its outputs have been found bit-identical to the
    checker's on these frames."""
    from oracle import oracle as O
    n = max(2, min(n, frames.shape[0]))
    host = frames[:n].cpu().numpy()
    t0 = time.perf_counter()
    results = O.extract_match_many(host, threads=1, fast=False)
    dt_checker = time.perf_counter() - t0
    O.fast_lib()                                            # (build outside the timed region)
    t0 = time.perf_counter()
    fast = O.extract_match_many(host, threads=1, fast=True)
    dt = time.perf_counter() - t0
    same = all(a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1])
               and ((a[2] is None and b[2] is None) or np.array_equal(a[2], b[2])) for a, b in zip(results, fast))
    if not same:
        raise SystemExit("bench.py: the -O3 -march=native oracle build differs from the -O2 checker")
    nk = sum(len(r[1]) for r in results)
    return ({"value": round(n / dt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
             "host_cores_available": os.cpu_count(),
             "build": "oracle/Makefile fast: gcc -O3 -march=native -fopenmp -ffp-contract=off (no fast-math); outputs "
                      "bit-identical to the -O2 checker on this sample (asserted before timing)",
             "checker_frames_per_s": round(n / dt_checker, 3),
             "sample": f"first {n} frames of the bench batch: Akaze::default() extract + symmetric match vs previous "
                       f"frame, single thread, {dt:.1f} s, {nk // n} keypoints/frame"}, results)


def cpu_baseline_all_cores(frames, threads):
    """The same -O3 -march=native build with OpenMP over frames (one pyramid per thread, dynamic schedule): what the
    reference's per-frame parallelism (cv-sfm extracts frame by frame; rayon inside a frame) could reach on this host."""
    from oracle import oracle as O
    # (the CPUs this process may run on — not omp_get_max_threads(), which an inherited OMP_NUM_THREADS=1 pins to one;
    # the thread count is set explicitly for the call)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(threads if threads > 0 else avail, avail))
    n = min(frames.shape[0], 2 * threads)
    host = frames[:n].cpu().numpy()
    O.extract_match_many(host[:min(n, threads)], threads=threads, fast=True, match=False)     # page the workers' pyramids in
    t0 = time.perf_counter()
    O.extract_match_many(host, threads=threads, fast=True)
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 2), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{n} frames, OpenMP over frames on {threads} threads (extract + symmetric match vs previous frame), {dt:.1f} s; "
                      f"host reports {os.cpu_count()} logical CPUs"}


def cpu_baseline_intra_frame(frames, checker_results, threads):
    """The reference as its own driver runs it: vslam-sandbox feeds VSlam::add_frame ONE frame at a time, and a
    frame's extraction is parallel only where the akaze crate's `rayon` feature makes it so — lib.rs:241-247 (the two simple
    Scharr filters), detector_response.rs:21,54,71-83 (the evolutions, and the five multiscale filters of one), scale_space_
    extrema.rs:352 and descriptors.rs:35 (the keypoints).  The separable filters, the 166 FED steps and the extrema search
    are serial in the reference too, so this is what all the host's cores buy a single frame.  Same -O3 -march=native build
    (ORC_OPT_INTRA); outputs asserted bit-identical to the checker's before they count."""
    from oracle import oracle as O
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(threads if threads > 0 else avail, avail, 32))     # (16 evolutions x nested joins: more threads only spin)
    n = min(len(checker_results), frames.shape[0])
    host = frames[:n].cpu().numpy()
    O.extract_many_intra(host[:1], threads=threads)         # the thread team, the pyramid
    t0 = time.perf_counter()
    got = O.extract_many_intra(host, threads=threads)
    dt = time.perf_counter() - t0
    same = all(a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1]) for a, b in zip(checker_results, got))
    if not same:
        raise SystemExit("bench.py: the intra-frame parallel oracle differs from the -O2 checker")
    return {"value": round(n / dt, 3), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"the same {n} frames, one after the other, each parallel at the akaze crate's four rayon sites only "
                      f"(OpenMP, {threads} threads; extraction only, no matching), {dt:.1f} s; bit-identical to the checker"}


if __name__ == "__main__":
    main()
