"""Row-band partition of an image over ranks, and the two ways the bands come together again.

The trace loop has no exchange between pixels: every rank holds the whole
(tiny) scene and renders its own round-robin bands of rows; the only
communication moves finished 8-bit rows (RCCL when the tensors live on GPUs,
gloo on CPU tensors in the tests):

  BandGather     one frame: one gather of every rank's rows to rank 0.
  FrameExchange  a run of frames: frame f is put together on rank f % N, a
                 whole group of frames moves in one all-to-all.  Every link of
                 the fully connected xGMI mesh then carries 1/N^2 of a frame
                 per frame in each direction, instead of rank 0's links
                 carrying all of it.

The reference has no counterpart: its render() (RayTracer.c:1574) is one loop
on one core.
"""
import numpy as np
import torch
import torch.distributed as dist


def rank_rows(height, band_rows, n_ranks, rank):
    """Image rows owned by `rank`, increasing: band b (rows b*band_rows ...) belongs to rank b % n_ranks."""
    ys = np.arange(height)
    if n_ranks <= 1:
        return ys
    return ys[(ys // band_rows) % n_ranks == rank]


def default_band_rows(height, n_ranks):
    """Fine bands even out the load (the lit part of a mesh scene is a compact blob of rows),
    16 rows = one tile row of the kernel's 16x16 pixel tiles."""
    return 16 if height >= 16 * n_ranks else max(1, height // max(1, n_ranks))


class BandGather:
    """Buffers and index maps for gathering every rank's packed rows into the full image on rank 0.

    `slots` > 1 gives that many independent (local, gathered) buffer pairs so that a caller can render frame
    i + 1 into one slot while the gather of frame i is still in flight on the communicator's stream.
    """

    def __init__(self, height, width, band_rows, world_size, rank, device, slots=1):
        self.height, self.width = height, width
        self.band_rows, self.world_size, self.rank = band_rows, world_size, rank
        self.rows = [rank_rows(height, band_rows, world_size, r) for r in range(world_size)]
        self.max_rows = max(len(r) for r in self.rows)
        self.n_local = len(self.rows[rank])
        self.slots = slots
        self.locals = [torch.zeros((self.max_rows, width, 3), dtype=torch.uint8, device=device) for _ in range(slots)]
        self.local = self.locals[0]
        self.gathereds = [None] * slots
        self.perm = None
        # gloo moves host memory: device buffers are staged through the host (CPU tests, one-GPU rehearsals)
        self.via_host = (world_size > 1 and torch.device(device).type == "cuda" and dist.is_initialized()
                         and dist.get_backend() == "gloo")
        if rank == 0:
            self.gathereds = [torch.zeros((world_size, self.max_rows, width, 3), dtype=torch.uint8,
                                          device="cpu" if self.via_host else device) for _ in range(slots)]
            # image row y comes from packed slot perm[y] of the flattened [world_size * max_rows] rows
            perm = np.empty(height, dtype=np.int64)
            for r, ys in enumerate(self.rows):
                perm[ys] = r * self.max_rows + np.arange(len(ys))
            self.perm = torch.from_numpy(perm).to(device)
        self.gathered = self.gathereds[0]

    def gather_async(self, slot=0):
        """Start the gather of slot `slot`; returns a handle for finish() (None when there is nothing to wait for)."""
        if self.world_size == 1:
            return None
        local = self.locals[slot].cpu() if self.via_host else self.locals[slot]
        if self.rank == 0:
            return dist.gather(local, list(self.gathereds[slot].unbind(0)), dst=0, async_op=True)
        return dist.gather(local, None, dst=0, async_op=True)

    def finish(self, handle, slot=0):
        """Wait for a gather (stream-ordered on GPUs) and return the full image on rank 0, None elsewhere."""
        if self.world_size == 1:
            return self.locals[slot][: self.n_local]
        if handle is not None:
            handle.wait()
        if self.rank != 0:
            return None
        flat = self.gathereds[slot].view(self.world_size * self.max_rows, self.width, 3)
        if self.via_host:
            flat = flat.to(self.perm.device)
        return flat.index_select(0, self.perm)

    def gather(self, slot=0):
        """Collective: returns the full [height, width, 3] image on rank 0, None elsewhere."""
        return self.finish(self.gather_async(slot), slot)


class FrameExchange:
    """Buffers and index maps for a run of frames rendered in row bands by every rank.

    Frames are handled in groups of `rounds * world_size`.  Rank r renders its bands of frame j of a group
    into `target(slot, j)`; `exchange_async(slot, n)` then moves the whole group with ONE all_to_all_single, after
    which frame j is complete on rank j % world_size; `finish()` waits for it (stream-ordered on GPUs) and
    returns this rank's frames as [(j, image[height, width, 3])].  `slots` independent buffer sets let the
    next group render while the previous one is in flight.
    """

    def __init__(self, height, width, band_rows, world_size, rank, device, rounds=4, slots=2):
        self.height, self.width = height, width
        self.band_rows, self.world_size, self.rank = band_rows, world_size, rank
        self.rounds, self.slots = rounds, slots
        self.group = rounds * world_size
        self.rows = [rank_rows(height, band_rows, world_size, r) for r in range(world_size)]
        self.max_rows = max(len(r) for r in self.rows)
        self.n_local = len(self.rows[rank])
        self.via_host = (world_size > 1 and torch.device(device).type == "cuda" and dist.is_initialized()
                         and dist.get_backend() == "gloo")
        shape = (world_size, rounds, self.max_rows, width, 3)
        self.send = [torch.zeros(shape, dtype=torch.uint8, device=device) for _ in range(slots)]
        self.recv = [torch.zeros(shape, dtype=torch.uint8, device="cpu" if self.via_host else device)
                     for _ in range(slots)] if world_size > 1 else self.send
        # image row y is packed row src_row[y] of the rows rank src_rank[y] rendered
        src_rank = np.empty(height, dtype=np.int64)
        src_row = np.empty(height, dtype=np.int64)
        for r, ys in enumerate(self.rows):
            src_rank[ys] = r
            src_row[ys] = np.arange(len(ys))
        idx_dev = "cpu" if self.via_host else device
        self.src_rank = torch.from_numpy(src_rank).to(idx_dev)
        self.src_row = torch.from_numpy(src_row).to(idx_dev)
        self.device = device

    def target(self, slot, j):
        """Where this rank renders its bands of frame j (0 <= j < group) of the group held by `slot`."""
        return self.send[slot][j % self.world_size, j // self.world_size]

    def exchange_async(self, slot, n_frames=None):
        """Start moving the first n_frames frames of the group in `slot` (default: the whole group)."""
        n = self.group if n_frames is None else n_frames
        assert 0 < n <= self.group
        used = (n + self.world_size - 1) // self.world_size  # rounds that hold at least one frame
        if self.world_size == 1:
            return (None, self.send[slot], used, n)
        send, recv = self.send[slot], self.recv[slot]
        if used < self.rounds:  # a short last group: move only the rounds in use
            send = send[:, :used].contiguous()
            recv = torch.empty(send.shape, dtype=torch.uint8, device=recv.device)
        if self.via_host:
            send = send.cpu()
        work = dist.all_to_all_single(recv, send, async_op=True)
        return (work, recv, used, n)

    def finish(self, handle, slot=0):
        """Wait for an exchange; returns [(j, image)] for the frames of the group that live on this rank."""
        work, recv, used, n = handle
        if work is not None:
            work.wait()
        out = []
        for g in range(used):
            j = g * self.world_size + self.rank
            if j < n and self.world_size == 1:
                out.append((j, recv[0, g, : self.n_local]))
            elif j < n:
                img = recv[self.src_rank, g, self.src_row]  # one gather kernel: [height, width, 3]
                out.append((j, img.to(self.device) if self.via_host else img))
        return out
