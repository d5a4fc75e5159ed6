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

    if world == 1:
        return results
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


def broadcast_grid(matcher, device, src: int = 0):
    """Replicates rank `src`'s correlation grid (bytes + offset) onto every rank's matcher.
    `matcher` needs GetCorrelationGrid(), grid_info()['offset'] and set_grid(grid, offset) /
    set_grid_dev(ptr, offset)."""
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
    if g.is_cuda:
        torch.cuda.synchronize()
        matcher.set_grid_dev(g.data_ptr(), offset)
        matcher.ctx.synchronize()
    else:
        matcher.set_grid(g.numpy(), offset)
    return g
