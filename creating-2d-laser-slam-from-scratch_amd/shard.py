"""Multi-GPU sharding of the batched many-scan mode.

The units (query scans) are independent: rank r of W matches scans [r*B/W, (r+1)*B/W) against
its own replica of the shared correlation grid -- no collective on the data path.  The only
exchanges are one broadcast of the 4 MB grid (optional: every rank can rebuild it from the same
base scans, which is cheaper) and one all_gather of the 112-byte result records.  With the
`nccl` backend these are RCCL collectives over xGMI; the same code runs on `gloo` for the CPU tests.
"""
from __future__ import annotations

import numpy as np

RESULT_BYTES = 112


def shard_range(n_units: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous balanced split: rank r owns [r*n/W, (r+1)*n/W)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return (rank * n_units) // world, ((rank + 1) * n_units) // world


def all_gather_results(results, world: int):
    """results: uint8 tensor [B_local, 112] (device or CPU).  Returns [sum B, 112] on every rank.
    Uneven shards are padded to the largest one for the collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist

    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return results  # (a process group of ONE still runs the collectives: how the 1-GPU box exercises the RCCL path)
    n_local = torch.tensor([results.shape[0]], dtype=torch.int64, device=results.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    counts = [int(c.item()) for c in counts]
    m = max(counts)
    padded = torch.zeros((m, RESULT_BYTES), dtype=torch.uint8, device=results.device)
    padded[: results.shape[0]] = results
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded)
    return torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)


def broadcast_grid(matcher, device, src: int = 0, install_device=None):
    """Replicates rank `src`'s correlation grid (bytes + offset) onto every rank's matcher.
    `matcher` needs GetCorrelationGrid(), grid_info()['offset'] and set_grid(grid, offset) /
    set_grid_dev(ptr, offset).  `device`: where the collective runs (the GPU for RCCL, "cpu" for gloo);
    `install_device`: a GPU to stage a CPU-side result on so that it is still installed through the device entry
    point set_grid_dev (how the 2-rank gloo test on a 1-GPU box exercises the path an RCCL run takes)."""
    import torch
    import torch.distributed as dist

    rank = dist.get_rank()
    gi = matcher.grid_info()
    nbytes = gi["height"] * gi["stride"]
    if rank == src:
        g = torch.from_numpy(np.ascontiguousarray(matcher.GetCorrelationGrid()).reshape(-1)).to(device)
        off = torch.tensor(np.asarray(gi["offset"], dtype=np.float64), device=device)
    else:
        g = torch.empty(nbytes, dtype=torch.uint8, device=device)
        off = torch.zeros(2, dtype=torch.float64, device=device)
    dist.broadcast(g, src)
    dist.broadcast(off, src)
    offset = off.cpu().numpy()
    if not g.is_cuda and install_device is not None:
        g = g.to(install_device)
    if g.is_cuda:
        torch.cuda.synchronize()
        matcher.set_grid_dev(g.data_ptr(), offset)
        matcher.ctx.synchronize()
    else:
        matcher.set_grid(g.numpy(), offset)
    return g


# ------------------------------------------------------------------------------------------------
# Offline map build from known poses (last row): the one path with a real exchange step.
# karto::OccupancyGrid::CreateFromScans depends on the scans only through the union of their boxes (min/max)
# and per-cell hit/pass counts (integer sums), so W ranks build partial grids of disjoint scan subsets and combine
# them exactly: one 4-double all-reduce (the box), one all-reduce(sum) of the two counter planes in a single
# contiguous buffer (at 2005^2 cells: 32 MB; ring all-reduce over xGMI is per-link bound, ~0.4 ms at W=8).
# ------------------------------------------------------------------------------------------------
def merge_boxes(box, device="cpu"):
    """All ranks' (minx, miny, maxx, maxy) -> the union box, bit for bit (negation and max are exact)."""
    import torch
    import torch.distributed as dist

    b = np.asarray(box, dtype=np.float64)
    t = torch.tensor([-b[0], -b[1], b[2], b[3]], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    m = t.cpu().numpy()
    return np.array([-m[0], -m[1], m[2], m[3]])


class GpuOccBackend:
    """The device path of the sharded build: lslam_occgrid_scan_bounds / _create_partial / counters."""

    def __init__(self, ctx, laser_params, device=None):
        self.ctx, self.laser, self.device = ctx, laser_params, device

    def scan_bounds(self, ranges, poses):
        from . import api
        return api.OccupancyGrid.scan_bounds(self.ctx, self.laser, ranges, poses)

    def create_partial(self, ranges, poses, resolution, box):
        from . import api
        return api.OccupancyGrid.CreatePartial(self.ctx, self.laser, ranges, poses, resolution, box)

    def all_reduce_counters(self, part):
        """In place on the grid's device for RCCL (`nccl`); staged through host memory for gloo."""
        import torch
        import torch.distributed as dist

        words = part.counter_words()
        if words == 0:
            return
        if dist.get_backend() == "nccl":
            t = torch.empty(words, dtype=torch.int32, device=self.device)  # two's complement: same sums as uint32
            torch.cuda.synchronize(self.device)
            part.export_counters_dev(t.data_ptr())
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize(self.device)
            part.import_counters_dev(t.data_ptr())
        else:
            c = part.export_counters().reshape(-1)
            t = torch.from_numpy(c.view(np.int32))
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            part.import_counters(c)


def build_occupancy_grid_sharded(backend, ranges_local, poses_local, resolution, device="cpu"):
    """This rank's scans (possibly none) in, the occupancy grid of ALL ranks' scans out, identical on every rank
    and identical to CreateFromScans over the concatenated scans.  None where the reference returns NULL (no
    scans on any rank).  `backend`: GpuOccBackend, or any object with scan_bounds / create_partial /
    all_reduce_counters (the CPU tests use the oracle)."""
    box = merge_boxes(backend.scan_bounds(ranges_local, poses_local), device)
    if not (box[0] <= box[2] and box[1] <= box[3]):
        return None
    part = backend.create_partial(ranges_local, poses_local, resolution, box)
    backend.all_reduce_counters(part)
    return part
