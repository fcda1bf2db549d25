"""Row-band partition of one image over ranks and the single gather to rank 0.

The trace loop has no exchange between pixels: every rank holds the whole
(tiny) scene and renders its own round-robin bands of rows; the only
communication is one gather of the 8-bit rows to rank 0 (RCCL when the
tensors live on GPUs, gloo on CPU tensors in the tests).  The reference has
no counterpart: its render() (RayTracer.c:1574) is one loop on one core.
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
