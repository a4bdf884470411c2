"""The N>1 path's exchange step on CPU: world_size 2 over gloo.  Descriptor blocks are
fake but tagged with their global frame index, so the test checks that after the all-gather every local
frame holds exactly its predecessor's block and that every frame becomes matchable exactly once."""
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
    from cv_amd.sharding import exchange_predecessors, global_index
    try:
        descs = torch.zeros((nf, cap, 64), dtype=torch.uint8)
        counts = torch.zeros((nf,), dtype=torch.int32)
        prev_descs = torch.full((nf, cap, 64), 255, dtype=torch.uint8)
        prev_counts = torch.full((nf,), -1, dtype=torch.int32)
        gath_d = torch.zeros((world, mb, cap, 64), dtype=torch.uint8)
        gath_n = torch.zeros((world, mb), dtype=torch.int32)
        ready_all = []
        for m0 in range(0, nf, mb):
            for j in range(m0, m0 + mb):          # "extract": tag the block with its global frame index
                g = global_index(rank, j, world)
                descs[j] = g % 251
                counts[j] = 1000 + g
            ready = exchange_predecessors(dist, rank, world, m0, mb, nf, descs[m0:m0 + mb], counts[m0:m0 + mb],
                                          gath_d, gath_n, prev_descs, prev_counts)
            for j in ready:                        # a frame must only be reported once its predecessor is filed
                gp = (global_index(rank, j, world) - 1) % (nf * world)
                assert int(prev_counts[j]) == 1000 + gp, (rank, j, int(prev_counts[j]), gp)
                assert bool((prev_descs[j] == gp % 251).all())
            ready_all += ready
        assert sorted(ready_all) == list(range(nf)), (rank, sorted(ready_all))
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nf,mb", [(8, 4), (6, 2), (4, 4), (3, 1)])
def test_predecessor_exchange_world2(nf, mb):
    world = 2
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
