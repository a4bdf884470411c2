"""The N>1 path's exchange step on CPU: world_size 2 and 3 over gloo.  Descriptor blocks are
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


@pytest.mark.parametrize("nf,mb,world", [(8, 4, 2), (6, 2, 2), (4, 4, 2), (3, 1, 2), (6, 3, 3)])
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


@pytest.mark.parametrize("nf,mb,world,k", [(8, 4, 2, 4), (6, 2, 3, 4), (4, 4, 2, 1), (6, 3, 3, 7)])
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
