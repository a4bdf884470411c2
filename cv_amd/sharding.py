"""Frame sharding across ranks for the multi-GPU front-end.

Frames are independent units for extraction (akaze::Akaze is Copy and stateless, akaze/src/lib.rs:108):
global frame g lives on rank g % world as local frame g // world.  The only exchange the path needs is the
one that brings frame g-1's descriptors to the owner of frame g for cross-frame matching
(cv-sfm/src/settings.rs:213-215: a new frame is matched against the recent ones).  A rank needs exactly ONE
other rank's block — its ring predecessor's — so the exchange is a ring shift, not an all-gather:

    rank r  --descs[m0 : m0+mb], counts[m0 : m0+mb]-->  rank (r + 1) % world

with RCCL a grouped ncclSend / ncclRecv pair per micro-batch (torch.distributed.batch_isend_irecv), i.e. one
block per rank over one xGMI link instead of world blocks landing on every rank, and the received block lands
directly where the matcher reads it (no staging buffer, no second copy):

  * rank r > 0: the predecessor of local frame j is (r-1, j)      -> prev[j]      <- block row j - m0
  * rank 0:     the predecessor of local frame j is (world-1, j-1) -> prev[j]      <- block row j - 1 - m0,
    i.e. the whole block lands one row further down, prev[m0+1 : m0+mb+1]; the predecessor of frame 0 (the
    step's last global frame) therefore lands in row nf: prev_* have nf + 1 rows and frame 0 is matched
    against row nf (`pred_row`).

Stream-ordering contract (explicit events, nothing implied):
  * the exchange is issued on the caller's CURRENT stream (bench.py: a dedicated `comm` stream).  Before the
    call that stream must already wait for (a) the event that marks this micro-batch's descriptors complete
    (the library's keypoint stream: `comm.wait_stream(akz_stream)` records and waits on such an event) and
    (b) the event that marks the matcher's last read of the receive rows (`match_done` of the step that used
    this buffer set before);
  * torch's NCCL process group runs the transfer on its own stream: it makes that stream wait for the current
    one at launch, and `work.wait()` below makes the current stream wait for the transfer (host does not
    block).  Everything enqueued on the current stream after the call — the matcher takes it as
    `stream_to_wait` and waits on an event recorded there — sees the received rows;
  * the send rows may be overwritten only by work ordered after that same event (bench.py: the extraction of
    the step after next waits on `match_done`, which is recorded after the matcher that waited on `comm`).

gloo (CPU tests, and the one-GPU smoke test of the N > 1 path) has no device point-to-point: device tensors
are staged through host memory with a device synchronisation on both sides.
torch.distributed is plumbing here; no kernel of the hot path depends on it.
"""


def owner(g, world):
    """(rank, local index) of global frame g."""
    return g % world, g // world


def global_index(rank, j, world):
    return j * world + rank


def pred_row(rank, j, nf):
    """Row of prev_descs / prev_counts that holds the predecessor of local frame j."""
    return nf if (rank == 0 and j == 0) else j


def _ring_shift(dist, rank, world, send_tensors, recv_tensors):
    """send_tensors[i] -> rank + 1, recv_tensors[i] <- rank - 1, one grouped operation."""
    nxt, prv = (rank + 1) % world, (rank - 1) % world
    staged = send_tensors[0].is_cuda and dist.get_backend() != "nccl"
    if staged:
        import torch
        torch.cuda.synchronize()                       # the producer stream has finished writing the rows
        s_host = [t.cpu() for t in send_tensors]
        r_host = [torch.empty(t.shape, dtype=t.dtype) for t in recv_tensors]
    else:
        s_host, r_host = [t.contiguous() for t in send_tensors], recv_tensors
    ops = []
    for t in s_host:
        ops.append(dist.P2POp(dist.isend, t, nxt))
    for t in r_host:
        ops.append(dist.P2POp(dist.irecv, t, prv))
    for work in dist.batch_isend_irecv(ops):
        work.wait()                                    # NCCL: orders the current stream after the transfer
    if staged:
        for dst, src in zip(recv_tensors, r_host):
            dst.copy_(src)
        torch.cuda.synchronize()


def exchange_predecessors(dist, rank, world, m0, mb, nf, descs_mb, counts_mb, prev_descs, prev_counts):
    """After the local frames [m0, m0+mb) of every rank have been extracted, pass their descriptor blocks one
    rank up the ring so that every rank holds the predecessors (global frame g-1, wrapping inside the step of
    nf*world frames) of its own frames.

    Returns the local frames whose predecessor became available with this call (and can be matched now);
    the predecessor of frame j is row pred_row(rank, j, nf).  Tensors: descs_mb [mb,cap,64] u8, counts_mb
    [mb] i32 (views of the rank's own outputs), prev_descs [nf+1,cap,64], prev_counts [nf+1]."""
    lo = m0 if rank > 0 else m0 + 1                    # rank 0 receives (world-1, j-1): one row further down
    _ring_shift(dist, rank, world, [descs_mb, counts_mb], [prev_descs[lo:lo + mb], prev_counts[lo:lo + mb]])
    if rank > 0:
        return list(range(m0, m0 + mb))                # predecessor of (rank, j) is (rank-1, j): this micro-batch
    ready = list(range(m0 + 1, m0 + mb))               # (0, j) <- (world-1, j-1) for the frames after the first
    if m0 > 0:
        ready.append(m0)                               # its predecessor arrived with the previous micro-batch
    if m0 + mb == nf:
        ready.append(0)                                # row nf: the step's last global frame
    return ready
