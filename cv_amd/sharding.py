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


def exchange_predecessors(dist, rank, world, m0, mb, nf, descs_mb, counts_mb, prev_descs, prev_counts, exchange=None):
    """After the local frames [m0, m0+mb) of every rank have been extracted, pass their descriptor blocks one
    rank up the ring so that every rank holds the predecessors (global frame g-1, wrapping inside the step of
    nf*world frames) of its own frames.

    Returns the local frames whose predecessor became available with this call (and can be matched now);
    the predecessor of frame j is row pred_row(rank, j, nf).  Tensors: descs_mb [mb,cap,64] u8, counts_mb
    [mb] i32 (views of the rank's own outputs), prev_descs [nf+1,cap,64], prev_counts [nf+1]."""
    lo = m0 if rank > 0 else m0 + 1                    # rank 0 receives (world-1, j-1): one row further down
    if exchange is None:
        _ring_shift(dist, rank, world, [descs_mb, counts_mb], [prev_descs[lo:lo + mb], prev_counts[lo:lo + mb]])
    else:
        exchange.shift(descs_mb, counts_mb, prev_descs[lo:lo + mb], prev_counts[lo:lo + mb])
    if rank > 0:
        return list(range(m0, m0 + mb))                # predecessor of (rank, j) is (rank-1, j): this micro-batch
    ready = list(range(m0 + 1, m0 + mb))               # (0, j) <- (world-1, j-1) for the frames after the first
    if m0 > 0:
        ready.append(m0)                               # its predecessor arrived with the previous micro-batch
    if m0 + mb == nf:
        ready.append(0)                                # row nf: the step's last global frame
    return ready


# ---- matching against a window of recent frames (g-1 .. g-k): the all-gather of SURVEY.md §8e -------------------
# cv-sfm matches a new frame against up to tracking_recent_frames = 32 recent views (cv-sfm/src/settings.rs:449-450, the
# loop at cv-sfm/src/lib.rs:1462-1486).  With frame g on rank g % world the views g-1 .. g-k live on the other ranks, all
# of them once k >= world - 1: every micro-batch's blocks are all-gathered, rank-major, into one slab per micro-batch,
#     gathered[m][r][i]  =  block of local frame m*mb + i of rank r  =  global frame (m*mb + i) * world + r
# and a view is addressed by its block index in that array (hm_knn_views_device's view_idx).

def gathered_block(g, world, nf, mb):
    """Index, in the [nf // mb][world][mb] array of gathered blocks, of global frame g (0 <= g < nf * world)."""
    r, j = g % world, g // world
    return ((j // mb) * world + r) * mb + (j % mb)


def window_views(rank, j, world, nf, k):
    """Global frames g-1 .. g-k of local frame j (g = j * world + rank), wrapping inside the step's nf * world frames."""
    total = nf * world
    g = j * world + rank
    return [(g - d) % total for d in range(1, k + 1)]


class TorchExchange:
    """The exchange through torch.distributed (nccl = RCCL on device tensors; gloo stages device tensors through the host)."""

    def __init__(self, dist, rank, world):
        self.dist, self.rank, self.world = dist, rank, world

    def shift(self, descs, counts, recv_descs, recv_counts):
        _ring_shift(self.dist, self.rank, self.world, [descs, counts], [recv_descs, recv_counts])

    def allgather(self, descs, counts, all_descs, all_counts):
        """descs [mb,cap,64], counts [mb] -> all_descs [world,mb,cap,64], all_counts [world,mb] (rank-major)."""
        dist = self.dist
        staged = descs.is_cuda and dist.get_backend() != "nccl"
        if staged:
            import torch
            torch.cuda.synchronize()
            for src, dst in ((descs, all_descs), (counts, all_counts)):
                host = [torch.empty(src.shape, dtype=src.dtype) for _ in range(self.world)]
                dist.all_gather(host, src.cpu())
                for r in range(self.world):
                    dst[r].copy_(host[r])
            torch.cuda.synchronize()
            return
        if descs.is_cuda:
            w1 = dist.all_gather_into_tensor(all_descs, descs.contiguous(), async_op=True)
            w2 = dist.all_gather_into_tensor(all_counts, counts.contiguous(), async_op=True)
            w1.wait(); w2.wait()              # NCCL: orders the current stream after the collectives
        else:
            dist.all_gather([all_descs[r] for r in range(self.world)], descs.contiguous())
            dist.all_gather([all_counts[r] for r in range(self.world)], counts.contiguous())

    def exposed(self):
        return None

    def close(self):
        pass


class AkzExchange:
    """The exchange through the library's own C ABI (akz_comm_*: RCCL loaded by libakz, its own stream).  The unique id
    is made by rank 0 and handed round through torch.distributed's store — the host application's part of the job."""

    def __init__(self, dist, rank, world, device):
        import ctypes as C
        import torch
        from . import _lib
        self._lib, self._C, self._torch = _lib, C, torch
        self.rank, self.world = rank, world
        L = _lib.lib()
        ident = [None]
        if rank == 0:
            buf = (C.c_uint8 * 128)()
            _lib.check(L.akz_comm_unique_id(buf), "akz_comm_unique_id")
            ident[0] = bytes(buf)
        if world > 1:
            dist.broadcast_object_list(ident, src=0)
        self._h = C.c_void_p()
        idb = (C.c_uint8 * 128).from_buffer_copy(ident[0])
        _lib.check(L.akz_comm_create(idb, rank, world, device, C.byref(self._h)), "akz_comm_create")
        self.stream = torch.cuda.ExternalStream(L.akz_comm_stream(self._h), device=torch.device("cuda", device))
        _lib.check(L.akz_comm_timing(self._h, 1, None, None, None, 1), "akz_comm_timing")

    def ranks_seen(self):
        """akz_comm_world(): the world size the library's communicator was created with — what RCCL saw."""
        return int(self._lib.lib().akz_comm_world(self._h))

    def _after(self):
        self._torch.cuda.current_stream().wait_stream(self.stream)    # what follows on the caller's stream sees the rows

    def shift(self, descs, counts, recv_descs, recv_counts):
        cur = self._torch.cuda.current_stream()
        self._lib.check(self._lib.lib().akz_comm_shift_blocks(
            self._h, descs.data_ptr(), counts.data_ptr(), descs.shape[0], descs.shape[1], recv_descs.data_ptr(),
            recv_counts.data_ptr(), self._lib.wait_handle(cur)), "akz_comm_shift_blocks")
        self._after()

    def allgather(self, descs, counts, all_descs, all_counts):
        cur = self._torch.cuda.current_stream()
        self._lib.check(self._lib.lib().akz_comm_allgather_blocks(
            self._h, descs.data_ptr(), counts.data_ptr(), descs.shape[0], descs.shape[1], all_descs.data_ptr(),
            all_counts.data_ptr(), self._lib.wait_handle(cur)), "akz_comm_allgather_blocks")
        self._after()

    def exposed(self):
        """(milliseconds of transfers on the comm stream, calls, bytes) since the last call."""
        C = self._C
        ms, calls, nbytes = C.c_double(), C.c_uint64(), C.c_uint64()
        self._lib.check(self._lib.lib().akz_comm_timing(self._h, 1, C.byref(ms), C.byref(calls), C.byref(nbytes), 1), "akz_comm_timing")
        return ms.value, calls.value, nbytes.value

    def close(self):
        if self._h:
            self._lib.lib().akz_comm_destroy(self._h)
            self._h = self._C.c_void_p()
