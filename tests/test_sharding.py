"""The N>1 path's exchange step on CPU: world_size 2, 3, 4 and 8 over gloo (8 = BASELINE configs[4]).  Descriptor blocks are
fake but tagged with their global frame index, so the test checks that after the ring shift every local
frame holds exactly its predecessor's block, that the rows a rank does not own are never written, and that every
frame becomes matchable exactly once."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nf, mb, cap, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cv_amd.sharding import exchange_predecessors, global_index, pred_row
    try:
        descs = torch.zeros((nf, cap, 64), dtype=torch.uint8)
        counts = torch.zeros((nf,), dtype=torch.int32)
        prev_descs = torch.full((nf + 1, cap, 64), 255, dtype=torch.uint8)
        prev_counts = torch.full((nf + 1,), -1, dtype=torch.int32)
        ready_all = []
        for m0 in range(0, nf, mb):
            for j in range(m0, m0 + mb):          # "extract": tag the block with its global frame index
                g = global_index(rank, j, world)
                descs[j] = g % 251
                counts[j] = 1000 + g
            ready = exchange_predecessors(dist, rank, world, m0, mb, nf, descs[m0:m0 + mb], counts[m0:m0 + mb],
                                          prev_descs, prev_counts)
            for j in ready:                        # a frame must only be reported once its predecessor is filed
                gp = (global_index(rank, j, world) - 1) % (nf * world)
                row = pred_row(rank, j, nf)
                assert int(prev_counts[row]) == 1000 + gp, (rank, j, int(prev_counts[row]), gp)
                assert bool((prev_descs[row] == gp % 251).all())
            ready_all += ready
        # the spare row is used by rank 0 only (the predecessor of its frame 0), row 0 by every other rank
        untouched = 0 if rank == 0 else nf
        assert int(prev_counts[untouched]) == -1 and bool((prev_descs[untouched] == 255).all())
        assert sorted(ready_all) == list(range(nf)), (rank, sorted(ready_all))
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nf,mb,world", [(8, 4, 2), (6, 2, 2), (4, 4, 2), (3, 1, 2), (6, 3, 3), (4, 2, 4), (4, 2, 8), (2, 1, 8)])
def test_predecessor_exchange(nf, mb, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nf, mb, 16, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def test_owner_mapping():
    from cv_amd.sharding import global_index, owner
    for world in (1, 2, 4, 8):
        for g in range(64):
            r, j = owner(g, world)
            assert global_index(r, j, world) == g


@pytest.mark.parametrize("world,nf,mb,k", [(8, 32, 8, 32), (8, 16, 4, 32), (8, 8, 8, 63), (4, 16, 8, 32), (8, 4, 4, 31)])
def test_world_8_index_arithmetic(world, nf, mb, k):
    """BASELINE configs[4] without the wire: the index arithmetic of an eight-rank job, checked rank by rank in one process.
    gathered_block is a bijection of the step's global frames onto the [nf // mb][world][mb] slab array and agrees with the
    rank-major layout an all-gather of micro-batch m produces; a window of K = 32 views (more than the world: several of
    them live on the SAME rank, and the first frames' windows wrap to the step's end) names exactly g-1 .. g-K; the ring
    shift's receive rows (pred_row) put the predecessor of every local frame of every rank where the matcher reads it."""
    from cv_amd.sharding import exchange_predecessors, gathered_block, global_index, owner, pred_row, window_views
    total = nf * world
    slab = {}
    for m in range(nf // mb):                        # what rank-major all-gathers of every micro-batch leave behind
        for r in range(world):
            for i in range(mb):
                slab[(m * world + r) * mb + i] = global_index(r, m * mb + i, world)
    assert sorted(slab) == list(range(total))
    for g in range(total):
        assert slab[gathered_block(g, world, nf, mb)] == g
    for r in range(world):
        for j in range(nf):
            g = global_index(r, j, world)
            views = window_views(r, j, world, nf, k)
            assert views == [(g - d) % total for d in range(1, k + 1)] and len(set(views)) == min(k, len(views))
            owners = [owner(v, world)[0] for v in views]
            if k >= world:
                assert set(owners) == set(range(world))            # every rank holds one of the frame's views
            assert owners.count(r) == k // world                   # ... and g-world, g-2*world, .. are the frame's own rank's
    # the ring shift, played by all ranks in one process: rank r's block of micro-batch m lands in rank r + 1's rows
    class FakeExchange:
        def __init__(self, r):
            self.r = r

        def shift(self, descs, counts, recv_descs, recv_counts):
            sent[self.r] = (descs, counts)
            recv[self.r] = (recv_descs, recv_counts)

    import torch
    blocks = [torch.tensor([global_index(r, j, world) for j in range(nf)], dtype=torch.int32) for r in range(world)]
    prev = [torch.full((nf + 1,), -1, dtype=torch.int32) for _ in range(world)]
    ready_all = [[] for _ in range(world)]
    for m0 in range(0, nf, mb):
        sent, recv, ready = {}, {}, {}
        for r in range(world):
            ready[r] = exchange_predecessors(None, r, world, m0, mb, nf, blocks[r][m0:m0 + mb], blocks[r][m0:m0 + mb],
                                             prev[r], prev[r], FakeExchange(r))
        for r in range(world):                       # the wire: what rank r - 1 sent is what rank r receives
            recv[r][1].copy_(sent[(r - 1) % world][1])
        for r in range(world):
            for j in ready[r]:
                assert int(prev[r][pred_row(r, j, nf)]) == (global_index(r, j, world) - 1) % total, (r, j)
            ready_all[r] += ready[r]
    for r in range(world):
        assert sorted(ready_all[r]) == list(range(nf))


def _window_worker(rank, world, port, nf, mb, cap, k, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cv_amd.sharding import TorchExchange, gathered_block, global_index, window_views
    try:
        ex = TorchExchange(dist, rank, world)
        descs = torch.zeros((nf, cap, 64), dtype=torch.uint8)
        counts = torch.zeros((nf,), dtype=torch.int32)
        gd = torch.full((nf // mb, world, mb, cap, 64), 255, dtype=torch.uint8)
        gc = torch.full((nf // mb, world, mb), -1, dtype=torch.int32)
        for m0 in range(0, nf, mb):
            for j in range(m0, m0 + mb):
                g = global_index(rank, j, world)
                descs[j] = g % 251
                counts[j] = 1000 + g
            ex.allgather(descs[m0:m0 + mb], counts[m0:m0 + mb], gd[m0 // mb], gc[m0 // mb])
        flat_d, flat_c = gd.reshape(-1, cap, 64), gc.reshape(-1)
        total = nf * world
        seen = set()
        for g in range(total):                         # every global frame sits where gathered_block says
            b = gathered_block(g, world, nf, mb)
            assert int(flat_c[b]) == 1000 + g and bool((flat_d[b] == g % 251).all()), (rank, g, b)
            seen.add(b)
        assert len(seen) == total
        for j in range(nf):                            # the window of every local frame: g-1 .. g-k, wrapping in the step
            g = global_index(rank, j, world)
            assert window_views(rank, j, world, nf, k) == [(g - d) % total for d in range(1, k + 1)]
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nf,mb,world,k", [(8, 4, 2, 4), (6, 2, 3, 4), (4, 4, 2, 1), (6, 3, 3, 7), (4, 2, 4, 6),
                                          (8, 4, 8, 32), (4, 2, 8, 31), (2, 2, 8, 15)])
def test_window_allgather(nf, mb, world, k):
    """All-gather route (SURVEY §8e, K recent views): after the per-micro-batch all-gathers every rank addresses the block
    of ANY global frame through gathered_block, and a frame's window is its K predecessors modulo the step."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_window_worker, args=(r, world, port, nf, mb, 8, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
