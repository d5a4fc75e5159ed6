"""N>1 path on CPU: world_size-2 gloo processes exercise the sharding helpers the multi-GPU bench
uses (scan ranges per rank, all_gather of 112-byte result records, grid broadcast)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

# spawned workers re-import this module without conftest.py: set the import alias up here
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lslam  # noqa: E402,F401
from lslam_amd import shard  # noqa: E402


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 4096, 4097):
        for w in (1, 2, 3, 8):
            parts = [shard.shard_range(n, w, r) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.shard_range(10, 2, 2)


class FakeMatcher:
    """numpy stand-in with the three methods broadcast_grid needs."""

    def __init__(self, rank):
        self.h, self.s = 37, 40
        rng = np.random.default_rng(123)
        self.grid = rng.integers(0, 101, (self.h, self.s), dtype=np.uint8) if rank == 0 else np.zeros((self.h, self.s), np.uint8)
        self.offset = np.array([-50.0, 12.5]) if rank == 0 else np.zeros(2)

    def grid_info(self):
        return {"height": self.h, "stride": self.s, "offset": self.offset}

    def GetCorrelationGrid(self):
        return self.grid

    def set_grid(self, g, off):
        self.grid = np.asarray(g, dtype=np.uint8).reshape(self.h, self.s).copy()
        self.offset = np.asarray(off).copy()


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import lslam  # noqa: F401
    from lslam_amd import shard as sh
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_total = 11  # uneven: 5 + 6
        lo, hi = sh.shard_range(n_total, world, rank)
        rec = torch.zeros((hi - lo, 112), dtype=torch.uint8)
        for i in range(lo, hi):
            rec[i - lo] = i + 1  # every byte of unit i's record = i+1
        allr = sh.all_gather_results(rec, world)
        ok_gather = allr.shape == (n_total, 112) and all(int(allr[i, 5]) == i + 1 for i in range(n_total))
        fm = FakeMatcher(rank)
        sh.broadcast_grid(fm, torch.device("cpu"), src=0)
        ref = FakeMatcher(0)
        ok_bcast = np.array_equal(fm.grid, ref.grid) and np.array_equal(fm.offset, ref.offset)
        q.put((rank, bool(ok_gather), bool(ok_bcast)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_gather_and_broadcast():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert out == [(0, True, True), (1, True, True)]


# ------------------------------------------------------------------------------------------------
# Sharded OccupancyGrid::CreateFromScans (last row): the control flow and the two all-reduces, with the
# CPU oracle standing in for the device kernels.  world_size 2 and 3, uneven shards, one EMPTY shard.
# ------------------------------------------------------------------------------------------------
class OracleOccBackend:
    def __init__(self, port):
        self.port = port

    def scan_bounds(self, ranges, poses):
        return self.port.occgrid_bounds(ranges, poses)

    def create_partial(self, ranges, poses, resolution, box):
        dims, cnt = self.port.occgrid_partial(ranges, poses, resolution, box)
        return {"dims": dims, "counters": cnt, "box": np.asarray(box).copy()}

    def all_reduce_counters(self, part):
        import torch.distributed as dist
        c = part["counters"].reshape(-1)
        if c.size:
            dist.all_reduce(torch.from_numpy(c.view(np.int32)), op=dist.ReduceOp.SUM)


def _occ_worker(rank, world, port, q, n_scans):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import lslam  # noqa: F401
    from lslam_amd import shard as sh, synth
    from oracle import pyoracle as O
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        wl = synth.make_match_workload(n_base=max(n_scans, 1), n_query=1, seed=6)
        ranges, poses = wl.base_ranges[:n_scans], wl.base_poses[:n_scans]
        kport = O.PortKarto(O.default_cfg(), O.laser_struct(wl.laser, 20.0))
        lo, hi = sh.shard_range(n_scans, world, rank)
        part = sh.build_occupancy_grid_sharded(OracleOccBackend(kport), ranges[lo:hi], poses[lo:hi], 0.05)
        if n_scans == 0:
            q.put((rank, part is None))
            return
        exp, off = kport.occgrid_from_scans(ranges, poses, 0.05)
        got = kport.occgrid_update(part["dims"], part["counters"])
        q.put((rank, bool(np.array_equal(got, exp) and np.array_equal(part["box"][:2], off))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_scans", [(2, 9), (3, 2), (2, 0)])  # (3, 2): rank 0 has no scans
def test_sharded_occupancy_grid_equals_whole_build(world, n_scans):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_occ_worker, args=(r, world, port, q, n_scans)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert out == [(r, True) for r in range(world)]
