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
    """Buffers and index maps for gathering every rank's packed rows into the full image on rank 0."""

    def __init__(self, height, width, band_rows, world_size, rank, device):
        self.height, self.width = height, width
        self.band_rows, self.world_size, self.rank = band_rows, world_size, rank
        self.rows = [rank_rows(height, band_rows, world_size, r) for r in range(world_size)]
        self.max_rows = max(len(r) for r in self.rows)
        self.local = torch.zeros((self.max_rows, width, 3), dtype=torch.uint8, device=device)
        self.n_local = len(self.rows[rank])
        self.gathered = None
        self.perm = None
        if rank == 0:
            self.gathered = torch.zeros((world_size, self.max_rows, width, 3), dtype=torch.uint8, device=device)
            # image row y comes from packed slot perm[y] of the flattened [world_size * max_rows] rows
            perm = np.empty(height, dtype=np.int64)
            for r, ys in enumerate(self.rows):
                perm[ys] = r * self.max_rows + np.arange(len(ys))
            self.perm = torch.from_numpy(perm).to(device)

    def gather(self):
        """Collective: returns the full [height, width, 3] image on rank 0, None elsewhere."""
        if self.world_size == 1:
            return self.local[: self.n_local]
        if self.rank == 0:
            dist.gather(self.local, list(self.gathered.unbind(0)), dst=0)
            flat = self.gathered.view(self.world_size * self.max_rows, self.width, 3)
            return flat.index_select(0, self.perm)
        dist.gather(self.local, None, dst=0)
        return None
