"""Row-tile sharding of one frame over the ranks of a torch.distributed job and
the gather of the tiles to rank 0 -- the multi-GPU form of the row bands of
c/main.c:510-529.  No data-path collective besides that gather."""
import numpy as np

from . import interleaved_job, job_row_indices, job_rows


class FrameShards:
    """Which rows each rank renders (8-row tiles dealt round-robin) and how the
    tiles go back into the frame."""

    def __init__(self, nx, ny, nsamples, world, tile_rows=8, **job_kw):
        self.nx, self.ny, self.world, self.tile_rows = nx, ny, world, tile_rows
        self.jobs = [interleaved_job(nx, ny, nsamples, r, world, tile_rows, **job_kw) for r in range(world)]
        self.rows = [job_row_indices(j) for j in self.jobs]
        self.max_rows = max((len(r) for r in self.rows), default=0)
        self._row_index = {}     # (rank, device) -> index tensor, built once

    def job(self, rank):
        return self.jobs[rank]

    def rows_of(self, rank):
        return self.rows[rank]

    def row_index(self, rank, device):
        import torch
        key = (rank, str(device))
        if key not in self._row_index:
            self._row_index[key] = torch.as_tensor(self.rows[rank], device=device)
        return self._row_index[key]

    def check(self):
        seen = np.concatenate(self.rows) if self.rows else np.zeros(0, np.int64)
        assert len(seen) == self.ny and np.array_equal(np.sort(seen), np.arange(self.ny)), "rows not covered exactly once"
        for j, r in zip(self.jobs, self.rows):
            assert (job_rows(j) if j.n_tiles else 0) == len(r)


def gather_frame(shards, my_tiles, rank, frame=None, gathered=None):
    """my_tiles: torch tensor [max_rows, nx, 3] uint8 holding this rank's tiles
    (padded at the end).  Rank 0 returns the assembled [ny, nx, 3] frame."""
    import torch
    import torch.distributed as dist

    world = shards.world
    if world > 1:
        if rank == 0 and gathered is None:
            gathered = [torch.zeros_like(my_tiles) for _ in range(world)]
        dist.gather(my_tiles, gathered if rank == 0 else None, dst=0)
    else:
        gathered = [my_tiles]
    if rank != 0:
        return None
    if frame is None:
        frame = torch.zeros((shards.ny, shards.nx, 3), dtype=my_tiles.dtype, device=my_tiles.device)
    for q in range(world):
        idx = shards.row_index(q, my_tiles.device)
        if len(idx):
            frame[idx] = gathered[q][: len(idx)]
    return frame
