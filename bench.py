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
from tools.bench_common import W, H, FRAMES_PER_STEP, CAP, make_world, make_frames  # noqa: E402,F401
from tools.roofline import *  # noqa: E402,F401,F403
from tools.roofline import _short_roofline  # noqa: E402
from tools.bench_extras import (extra_match, extra_pipeline_verify, extra_pipeline_register, extra_ransac,  # noqa: E402
                                extra_criterion)


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
                descs[m0:m0 + MB].data_ptr(), CAP, counts[m0:m0 + MB].data_ptr(), _lib.wait_handle(cur)), "extract")
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
                len(js), RULE_STRICT, 24, 0.0, 1, pairs[m0:].data_ptr(), npairs[m0:].data_ptr(), _lib.wait_handle(wait)),
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
                                                 _lib.wait_handle(wait) if p0 == 0 else None), "knn_batch")
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
