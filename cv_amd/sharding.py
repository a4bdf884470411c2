"""Frame sharding across ranks for the multi-GPU front-end.

Frames are independent units for extraction (akaze::Akaze is Copy and stateless, akaze/src/lib.rs:108):
global frame g lives on rank g % world as local frame g // world.  The only exchange the path needs is
the one that brings frame g-1's descriptors to the owner of frame g for cross-frame matching: an
all-gather of the ranks' fixed-capacity descriptor blocks (one RCCL collective per micro-batch).
torch.distributed is plumbing here; no kernel of the hot path depends on it.
"""


def owner(g, world):
    """(rank, local index) of global frame g."""
    return g % world, g // world


def global_index(rank, j, world):
    return j * world + rank


def _all_gather(dist, out, inp):
    """out[r] <- rank r's inp.  One collective; RCCL (backend "nccl") takes the fused single-buffer form, the
    list form is the fallback for backends without it (gloo on device tensors, used only by the 1-GPU smoke
    test of the multi-rank path)."""
    try:
        dist.all_gather_into_tensor(out.view(-1), inp.reshape(-1))
    except (RuntimeError, NotImplementedError):
        # gloo stages device tensors through host memory on its own streams; a full device sync on both
        # sides keeps that staging ordered with the producer / consumer streams of this process (the RCCL
        # path above is stream-ordered and needs none).  Seen without it: roughly one smoke run in fifteen
        # matched a micro-batch against the previous micro-batch's gathered block.
        import torch
        torch.cuda.synchronize()
        dist.all_gather(list(out.unbind(0)), inp.contiguous())
        torch.cuda.synchronize()


def exchange_predecessors(dist, rank, world, m0, mb, nf, descs_mb, counts_mb, gath_d, gath_n, prev_descs,
                          prev_counts):
    """After the local frames [m0, m0+mb) of every rank have been extracted, all-gather their descriptor
    blocks and file each local frame's predecessor (global frame g-1, wrapping inside the step of
    nf*world frames) into prev_descs / prev_counts.

    Returns the local frames whose predecessor block became available with this call (and can be
    matched now).  Tensors: descs_mb [mb,cap,64] u8, counts_mb [mb] i32, gath_d [world,mb,cap,64],
    gath_n [world,mb], prev_descs [nf,cap,64], prev_counts [nf]."""
    _all_gather(dist, gath_d, descs_mb)
    _all_gather(dist, gath_n, counts_mb)
    if rank > 0:
        # predecessor of (rank, j) is (rank-1, j): same micro-batch
        prev_descs[m0:m0 + mb].copy_(gath_d[rank - 1])
        prev_counts[m0:m0 + mb].copy_(gath_n[rank - 1])
        return list(range(m0, m0 + mb))
    # rank 0: predecessor of (0, j) is (world-1, j-1); (0, 0) wraps to (world-1, nf-1)
    ready = []
    if mb > 1:
        prev_descs[m0 + 1:m0 + mb].copy_(gath_d[world - 1, :mb - 1])
        prev_counts[m0 + 1:m0 + mb].copy_(gath_n[world - 1, :mb - 1])
        ready += list(range(m0 + 1, m0 + mb))
    nxt = m0 + mb if m0 + mb < nf else 0
    prev_descs[nxt].copy_(gath_d[world - 1, mb - 1])
    prev_counts[nxt].copy_(gath_n[world - 1, mb - 1])
    if m0 > 0:
        ready.append(m0)      # its predecessor arrived with the previous micro-batch
    if nxt == 0:
        ready.append(0)       # the step's first frame pairs with the step's last frame
    return ready
