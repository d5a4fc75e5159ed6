"""Karto hit/pass-counter occupancy grid (lesson6's published map, next-row #1): the GPU build must
equal the CPU oracle cell for cell -- integer counters, exact by construction."""
import os
import sys
import time

import numpy as np
import pytest

# spawned rank workers re-import this module without conftest.py: set the import alias up here
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lslam  # noqa: E402,F401
from lslam_amd import api, synth  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("res,thr,n_scans", [(0.05, 20.0, 24), (0.05, 49.5, 40), (0.1, 12.0, 10)])
def test_create_from_scans_matches_oracle(ctx, oracle_lib, res, thr, n_scans):
    wl = synth.make_match_workload(n_base=n_scans, n_query=1, seed=6)
    laser = wl.laser
    port = oracle_lib.PortKarto(oracle_lib.default_cfg(), oracle_lib.laser_struct(laser, thr))
    ranges = wl.base_ranges.copy()
    ranges[0, 5] = np.nan       # ignored readings
    ranges[1, 7] = 0.05         # below minimum range
    ranges[2, 9] = 70.0         # beyond maximum range
    exp, off = port.occgrid_from_scans(ranges, wl.base_poses, res)
    g = api.OccupancyGrid.CreateFromScans(ctx, api.laser_params(laser, thr), ranges, wl.base_poses, res)
    w, h, goff, gres = g.info()
    assert (h, w) == exp.shape and gres == res
    assert np.array_equal(goff, off)
    got = g.data()
    assert (exp == 100).sum() > 100 and (exp == 255).sum() > 5000
    assert np.array_equal(got, exp)
    ros = g.ros_data()
    assert np.array_equal(ros == -1, exp == 0) and np.array_equal(ros == 100, exp == 100) and np.array_equal(ros == 0, exp == 255)


def test_no_scans_is_null(ctx):
    assert api.OccupancyGrid.CreateFromScans(ctx, api.laser_params(synth.Laser()), np.zeros((0, 1081)), np.zeros((0, 3)), 0.05) is None


# ---- sharded build (SURVEY 8(e) "offline map build from known poses") ----
def _workload(n_scans, thr=20.0):
    wl = synth.make_match_workload(n_base=n_scans, n_query=1, seed=6)
    return wl, api.laser_params(wl.laser, thr)


def test_partials_merge_to_the_whole_build(ctx, oracle_lib):
    """Three disjoint scan subsets (one empty) on the merged box; counters merged on the host path and on the
    device path; cells equal the one-shot build and the oracle."""
    import torch
    wl, lp = _workload(17)
    ranges, poses = wl.base_ranges, wl.base_poses
    whole = api.OccupancyGrid.CreateFromScans(ctx, lp, ranges, poses, 0.05)
    port = oracle_lib.PortKarto(oracle_lib.default_cfg(), oracle_lib.laser_struct(wl.laser, 20.0))
    exp, off = port.occgrid_from_scans(ranges, poses, 0.05)
    cuts = [(0, 6), (6, 6), (6, 17)]
    boxes = [api.OccupancyGrid.scan_bounds(ctx, lp, ranges[a:b], poses[a:b]) for a, b in cuts]
    assert np.array_equal(boxes[1], [1e18, 1e18, -1e18, -1e18])  # BoundingBox2() (Karto.h:2765)
    for (a, b), bx in zip(cuts, boxes):
        if b > a:
            assert np.array_equal(bx, port.occgrid_bounds(ranges[a:b], poses[a:b]))
    box = np.array([min(b[0] for b in boxes), min(b[1] for b in boxes), max(b[2] for b in boxes), max(b[3] for b in boxes)])
    parts = [api.OccupancyGrid.CreatePartial(ctx, lp, ranges[a:b], poses[a:b], 0.05, box) for a, b in cuts]
    assert parts[0].info()[:2] == whole.info()[:2] and np.array_equal(parts[0].info()[2], off)
    assert not parts[1].export_counters().any()
    # counters of a shard = the oracle's for the same shard
    d, cnt = port.occgrid_partial(ranges[6:17], poses[6:17], 0.05, box)
    assert np.array_equal(parts[2].export_counters(), cnt)
    # host path: accumulate exported buffers
    parts[0].import_counters(parts[1].export_counters(), accumulate=True)
    parts[0].import_counters(parts[2].export_counters(), accumulate=True)
    assert np.array_equal(parts[0].export_counters(), whole.export_counters())
    assert np.array_equal(parts[0].data(), exp) and np.array_equal(whole.data(), exp)
    # device path: a torch-owned buffer, as the RCCL all-reduce uses it
    t = torch.empty(parts[2].counter_words(), dtype=torch.int32, device="cuda:0")
    parts[2].export_counters_dev(t.data_ptr())
    torch.cuda.synchronize()
    parts[1].import_counters_dev(t.data_ptr(), accumulate=True)
    assert np.array_equal(parts[1].export_counters(), cnt)
    parts[1].import_counters_dev(t.data_ptr())  # replace
    assert np.array_equal(parts[1].export_counters(), cnt)
    with pytest.raises(ValueError):
        parts[1].import_counters(np.zeros(5, dtype=np.uint32))


def test_partial_rejects_empty_box(ctx):
    wl, lp = _workload(2)
    with pytest.raises(api.LslamError):
        api.OccupancyGrid.CreatePartial(ctx, lp, wl.base_ranges[:0], wl.base_poses[:0], 0.05, [1e18, 1e18, -1e18, -1e18])


def _rank_worker(rank, world, port, q, n_scans):
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import lslam  # noqa: F401
    import torch.distributed as dist
    from lslam_amd import api as A, shard as sh, synth as sy

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # both ranks share the box's one GPU
    try:
        wl = sy.make_match_workload(n_base=n_scans, n_query=1, seed=6)
        c = A.Context(0)
        lp = A.laser_params(wl.laser, 20.0)
        lo, hi = sh.shard_range(n_scans, world, rank)
        g = sh.build_occupancy_grid_sharded(sh.GpuOccBackend(c, lp, "cuda:0"), wl.base_ranges[lo:hi], wl.base_poses[lo:hi], 0.05)
        w, h, off, _ = g.info()
        q.put((rank, w, h, off.tolist(), g.data().tobytes()))
    finally:
        dist.destroy_process_group()


def test_two_ranks_build_the_same_grid_as_one(ctx, oracle_lib):
    """Two gloo ranks on the one GPU of the box: every rank ends with the oracle's grid of ALL scans."""
    import socket
    import torch.multiprocessing as mp
    n_scans = 13
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mctx = mp.get_context("spawn")
    q = mctx.Queue()
    procs = [mctx.Process(target=_rank_worker, args=(r, 2, port, q, n_scans)) for r in range(2)]
    for p in procs:
        p.start()
    out = []
    for _ in range(600):  # a rank that died must fail the test at once, not after the queue timeout
        while not q.empty():
            out.append(q.get())
        if len(out) == 2 or any(p.exitcode not in (None, 0) for p in procs):
            break
        time.sleep(0.2)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    out = sorted(out)
    assert len(out) == 2
    wl, _ = _workload(n_scans)
    kport = oracle_lib.PortKarto(oracle_lib.default_cfg(), oracle_lib.laser_struct(wl.laser, 20.0))
    exp, off = kport.occgrid_from_scans(wl.base_ranges, wl.base_poses, 0.05)
    for rank, w, h, goff, cells in out:
        assert (h, w) == exp.shape and goff == off.tolist()
        assert cells == exp.tobytes()


# ---- RCCL called directly from the C++ host (lslam_occgrid_create_sharded / lslam_pool_occgrid_from_scans) ----
def test_pool_occupancy_grid_rccl_clique_and_merge_path(oracle_lib):
    """One process, the pool's devices: [0] forms an RCCL clique of one (ncclCommInitAll + both all-reduces run, on the
    context stream); [0, 0, 0] cannot (one GPU named three times) and takes the counter-addition path.  Both must equal
    the oracle's grid of all scans.  On an 8-GPU node `MatcherPool(cfg, laser, 0)` is the clique of eight."""
    wl, lp = _workload(19)
    port = oracle_lib.PortKarto(oracle_lib.default_cfg(), oracle_lib.laser_struct(wl.laser, 20.0))
    exp, off = port.occgrid_from_scans(wl.base_ranges, wl.base_poses, 0.05)
    for devices in ([0], [0, 0, 0]):
        pool = api.MatcherPool(api.baseline_config(range_threshold=20.0), lp, devices)
        got, goff = pool.CreateOccupancyGrid(lp, wl.base_ranges, wl.base_poses, 0.05)
        pool.close()
        assert got.shape == exp.shape and np.array_equal(goff, off), devices
        assert np.array_equal(got, exp), devices


def test_create_sharded_with_a_callers_communicator(ctx, oracle_lib):
    """lslam_occgrid_create_sharded with an ncclComm_t made by the CALLER (ncclGetUniqueId + ncclCommInitRank through
    ctypes on librccl, world size 1 on this box; one per process on an 8-GPU node)."""
    import ctypes as C

    rccl = C.CDLL("librccl.so.1", mode=C.RTLD_GLOBAL)

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid = UniqueId()
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    wl, lp = _workload(11)
    port = oracle_lib.PortKarto(oracle_lib.default_cfg(), oracle_lib.laser_struct(wl.laser, 20.0))
    exp, off = port.occgrid_from_scans(wl.base_ranges, wl.base_poses, 0.05)
    g = api.OccupancyGrid.CreateSharded(ctx, lp, wl.base_ranges, wl.base_poses, 0.05, comm.value)
    assert np.array_equal(g.data(), exp)
    g.close()
    # a rank without scans in a world of one: the union box is empty -> the reference's NULL
    with pytest.raises(api.LslamError):
        api.OccupancyGrid.CreateSharded(ctx, lp, np.zeros((0, 1081)), np.zeros((0, 3)), 0.05, comm.value)
    # arguments every rank shares are refused BEFORE the first collective, also on a rank that holds no scans (ADVICE r03:
    # an EMPTY shard used to pass a too-short row stride and wait in ncclAllReduce for peers that had returned) -- and
    # the communicator is still usable afterwards, i.e. nobody entered a collective
    for bad in (dict(ranges=np.zeros((0, 100)), res=0.05), dict(ranges=np.zeros((0, 1081)), res=0.0)):
        with pytest.raises(api.LslamError) as e:
            api.OccupancyGrid.CreateSharded(ctx, lp, bad["ranges"], np.zeros((0, 3)), bad["res"], comm.value)
        assert e.value.code == -1
    g = api.OccupancyGrid.CreateSharded(ctx, lp, wl.base_ranges, wl.base_poses, 0.05, comm.value)
    assert np.array_equal(g.data(), exp)
    g.close()
    rccl.ncclCommDestroy(comm)
