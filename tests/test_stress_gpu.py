"""Randomised parity sweep: different lasers (beam counts not multiples of 64), batch sizes (1 -> the
beam-sliced atomics path, larger -> one wave per (scan, angle)), poses anywhere in the grid
including the rim (rows leaving the flat index range, occupancy columns wrapping), random invalid
readings.  Integer response sums must be bit-exact, results within 1e-9 of the CPU oracle."""
import math

import numpy as np
import pytest

from lslam_amd import api, synth

pytestmark = pytest.mark.gpu


# (2500 beams: more than 64 x 32 per wave, so the chip-filling coarse kernel runs with two beam slices per (scan, angle) --
# the parked-beam bookkeeping of its phase A is per slice -- and the rebuild leaves its LDS-resident form)
@pytest.mark.parametrize("n_ranges,inc_deg,seed", [(360, 1.0, 1), (1081, 0.25, 2), (721, 0.5, 3), (1080, 0.3333, 4),
                                                    (2500, 0.1, 5)])
def test_random_sweep(ctx, oracle_lib, n_ranges, inc_deg, seed):
    rng = np.random.default_rng(seed)
    laser = synth.Laser(n_ranges=n_ranges, angle_min=math.radians(-0.5 * (n_ranges - 1) * inc_deg),
                        angle_increment=math.radians(inc_deg), range_max=40.0)
    thr = 30.0
    port = oracle_lib.PortKarto(oracle_lib.default_cfg(), oracle_lib.laser_struct(laser, thr))
    gm = api.ScanMatcher(ctx, api.baseline_config(range_threshold=thr), api.laser_params(laser, thr))
    world = synth.arena(size=50.0, n_axis=14, n_rot=5, seed=seed)
    wl = synth.make_match_workload(n_base=16, n_query=24, seed=seed, laser=laser, world=world, query_spread=4.0)
    port.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    assert np.array_equal(gm.GetCorrelationGrid(), port.grid())
    ranges, poses = wl.query_ranges.copy(), wl.query_poses.copy()
    # random invalid readings and a few poses pushed to the rim / far corners of the grid
    mask = rng.random(ranges.shape) < 0.03
    ranges[mask] = np.where(rng.random(mask.sum()) < 0.5, np.nan, np.inf)
    half = 0.5 * 0.05 * (gm.grid_info()["roi_w"] - 1)
    for i, (fx, fy) in enumerate([(0.97, 0.0), (-0.97, 0.97), (0.0, -0.985), (0.985, 0.985)]):
        poses[i, 0] = wl.center_pose[0] + fx * half
        poses[i, 1] = wl.center_pose[1] + fy * half
        poses[i, 2] = rng.uniform(-math.pi, math.pi)
    # integer numerators, fast and generic kernels, single scan (beam-sliced atomics)
    for q in (0, 1, 5, 9):
        p = poses[q]
        _, _, _, st, sums = port.correlate_scan(ranges[q], p, p, 0.5, 0.1, 0.349, 0.0349, True, False, want_sums=True)
        assert st == 0
        assert np.array_equal(gm.coarse_sums(ranges[q], p), sums)
        assert np.array_equal(gm.coarse_sums(ranges[q], p, force_generic=True), sums)
    # whole batch, then odd-sized sub-batches (S = 1, 3, 7 take different wave decompositions)
    res = gm.match_batch(ranges, poses)
    exp = [port.match(ranges[q], poses[q]) for q in range(len(ranges))]
    for q, (mean, cov, resp) in enumerate(exp):
        assert res["status"][q] == 0
        assert np.abs(res["pose"][q][:2] - mean[:2]).max() <= 1e-9
        assert abs(math.remainder(res["pose"][q][2] - mean[2], 2 * math.pi)) <= 1e-9
        assert abs(res["response"][q] - resp) <= 1e-12
        assert np.abs(res["covariance"][q] - cov).max() <= 1e-9 * max(1.0, np.abs(cov).max())
    for S in (1, 3, 7):
        sub = gm.match_batch(ranges[:S], poses[:S])
        assert sub.tobytes() == res[:S].tobytes()
    # a batch big enough for the fine pass to switch to the 4x4-block kernel (k_resp_tile3): the same
    # scans tiled 8x must give byte-identical results to the row-kernel batch above
    ctx.profile(True)
    ctx.profile_reset()
    big = gm.match_batch(np.tile(ranges, (8, 1)), np.tile(poses, (8, 1)))
    prof = ctx.profile_read()
    ctx.profile(False)
    assert "resp_tile_fine" in prof and "resp_rows_fine" not in prof
    assert big.tobytes() == np.tile(res, 8).tobytes()


def _kround(v):
    return math.floor(v + 0.5) if v >= 0.0 else math.ceil(v - 0.5)


def test_non_uniform_lattices(ctx, oracle_lib):
    """Search centres whose lattice coordinates sit on cell boundaries: (centre + x_i - offset) * scale
    lands on k + 0.5 up to rounding noise, so neighbouring lattice coordinates round differently and the
    lattice is NOT uniform.  The packed kernels skip such scans and the scan's reduce block computes
    the numerators itself (block_generic_fallback); results must still equal the oracle's."""
    laser = synth.Laser()
    thr = 12.0
    port = oracle_lib.PortKarto(oracle_lib.default_cfg(), oracle_lib.laser_struct(laser, thr))
    gm = api.ScanMatcher(ctx, api.baseline_config(range_threshold=thr), api.laser_params(laser, thr))
    wl = synth.make_match_workload(n_base=20, n_query=24, seed=11, laser=laser)
    port.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    off = gm.grid_info()["offset"]
    res, scale = 0.05, 1.0 / 0.05
    ranges, poses = wl.query_ranges.copy(), wl.query_poses.copy()
    n_uneven = 0
    for q in range(len(poses)):
        # put x (even q) or y (odd q) of the centre half a cell off the grid raster
        ax = q & 1
        k = _kround((poses[q, ax] - off[ax]) * scale)
        poses[q, ax] = off[ax] + (k + 0.5) * res
        # the coarse lattice of this centre, with the device's own arithmetic (k_pass_setup)
        nx = int(_kround(0.5 * 2.0 / (2 * res)) + 1)
        cells = [_kround(((poses[q, ax] + (-0.5 + i * (2 * res))) - off[ax]) * scale) for i in range(nx)]
        steps = {b - a for a, b in zip(cells, cells[1:])}
        n_uneven += steps != {2}
    assert n_uneven >= 4, "the construction should produce non-uniform lattices"
    exp = [port.match(ranges[q], poses[q]) for q in range(len(ranges))]

    def check(res_):
        for q, (mean, cov, resp) in enumerate(exp):
            assert res_["status"][q] == 0
            assert np.abs(res_["pose"][q][:2] - mean[:2]).max() <= 1e-9
            assert abs(math.remainder(res_["pose"][q][2] - mean[2], 2 * math.pi)) <= 1e-9
            assert abs(res_["response"][q] - resp) <= 1e-12
            assert np.abs(res_["covariance"][q] - cov).max() <= 1e-9 * max(1.0, np.abs(cov).max())

    small = gm.match_batch(ranges, poses)  # row kernels with beam slices
    check(small)
    big = gm.match_batch(np.tile(ranges, (9, 1)), np.tile(poses, (9, 1)))  # tiled coarse + 4x4-block fine kernels
    assert big.tobytes() == np.tile(small, 9).tobytes()
    for q in range(4):  # integer numerators of non-uniform lattices, through the debug hook
        p = poses[q]
        _, _, _, st, sums = port.correlate_scan(ranges[q], p, p, 0.5, 0.1, 0.349, 0.0349, True, False, want_sums=True)
        assert st == 0 and np.array_equal(gm.coarse_sums(ranges[q], p), sums)


def test_many_beam_laser(ctx, oracle_lib):
    """A 0.125-degree laser (2881 beams): more than kMaxBeamsPerLane * 64 beams per (scan, angle), so even
    a chip-filling batch splits the beams of a wave (beam slices + integer atomics) to keep the packed
    16-bit partial sums exact; the fine pass packs up to 256 beams per lane."""
    n = 2881
    laser = synth.Laser(n_ranges=n, angle_min=math.radians(-180.0), angle_increment=math.radians(0.125),
                        range_max=30.0)
    thr = 20.0
    port = oracle_lib.PortKarto(oracle_lib.default_cfg(), oracle_lib.laser_struct(laser, thr))
    gm = api.ScanMatcher(ctx, api.baseline_config(range_threshold=thr), api.laser_params(laser, thr))
    world = synth.arena(size=40.0, n_axis=12, n_rot=4, seed=21)
    wl = synth.make_match_workload(n_base=10, n_query=12, seed=21, laser=laser, world=world, query_spread=2.0)
    port.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    assert np.array_equal(gm.GetCorrelationGrid(), port.grid())
    p0 = wl.query_poses[0]
    _, _, _, st, sums = port.correlate_scan(wl.query_ranges[0], p0, p0, 0.5, 0.1, 0.349, 0.0349, True, False,
                                            want_sums=True)
    assert st == 0 and np.array_equal(gm.coarse_sums(wl.query_ranges[0], p0), sums)
    small = gm.match_batch(wl.query_ranges, wl.query_poses)
    for q in range(len(small)):
        mean, cov, resp = port.match(wl.query_ranges[q], wl.query_poses[q])
        assert small["status"][q] == 0
        assert np.abs(small["pose"][q][:2] - mean[:2]).max() <= 1e-9
        assert abs(math.remainder(small["pose"][q][2] - mean[2], 2 * math.pi)) <= 1e-9
        assert abs(small["response"][q] - resp) <= 1e-12
    big = gm.match_batch(np.tile(wl.query_ranges, (17, 1)), np.tile(wl.query_poses, (17, 1)))  # 204 scans
    assert big.tobytes() == np.tile(small, 17).tobytes()


def test_matcher_pool_shards_like_one_matcher(ctx, workload_spread):
    """lslam_pool (C++ host, one process): scans sharded [r*B/W, (r+1)*B/W) over the pool's matchers, grid replicated
    device-to-device (or rebuilt everywhere); byte-identical to one matcher.  On the 1-GPU box the pool holds three
    contexts on device 0; `devices=0` takes every visible GPU."""
    wl = workload_spread
    cfg, lp = api.baseline_config(), api.laser_params(wl.laser)
    gm = api.ScanMatcher(ctx, cfg, lp)
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    want = gm.match_batch(wl.query_ranges, wl.query_poses)
    for devices, rebuild in (([0, 0, 0], False), ([0, 0], True), (0, False)):
        pool = api.MatcherPool(cfg, lp, devices)
        assert pool.devices == (len(devices) if isinstance(devices, list) else pool.devices) >= 1
        pool.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose, rebuild_everywhere=rebuild)
        got = pool.match_batch(wl.query_ranges, wl.query_poses)
        assert got.tobytes() == want.tobytes()
        assert pool.match_batch(wl.query_ranges[:0], wl.query_poses[:0]).shape == (0,)
        pool.close()
    with pytest.raises(api.LslamError):
        api.MatcherPool(cfg, lp, 99)


def test_two_matchers_one_context_interleaved(ctx, workload):
    """The sequential and the loop matcher of karto::Mapper (Mapper.cpp:1964-1968, 862-871) are two ScanMatcher instances
    living side by side: two lslam_matcher handles on ONE context, used alternately, keep their own grids and scratch."""
    wl = workload
    a = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(wl.laser))
    b = api.ScanMatcher(ctx, api.baseline_config(search_size=4.0), api.laser_params(wl.laser))
    a.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    b.AddScans(wl.base_ranges[:10], wl.base_poses[:10], wl.center_pose)
    ra0 = a.match_batch(wl.query_ranges, wl.query_poses)
    rb0 = b.match_batch(wl.query_ranges[:3], wl.query_poses[:3], doPenalize=False, doRefineMatch=False)
    for _ in range(3):
        rb = b.match_batch(wl.query_ranges[:3], wl.query_poses[:3], doPenalize=False, doRefineMatch=False)
        ra = a.match_batch(wl.query_ranges, wl.query_poses)
        assert ra.tobytes() == ra0.tobytes() and rb.tobytes() == rb0.tobytes()


def test_concurrent_use_of_one_matcher_is_refused_not_corrupted(ctx, workload_spread):
    """A ScanMatcher instance is not re-entrant (its grid and workspaces are shared state, Mapper.h:1273-1278).  Two host
    threads hammering ONE matcher must each get either a correct result or LSLAM_ERR_INVALID_ARGUMENT -- never a wrong
    result or a memory fault (round 1's two-stream experiment grew the shared workspaces under running kernels)."""
    import threading

    wl = workload_spread
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(wl.laser))
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    want = [gm.match_batch(wl.query_ranges[:k], wl.query_poses[:k]).tobytes() for k in (7, 40)]
    outcome = {"ok": 0, "refused": 0, "wrong": 0}
    lock = threading.Lock()

    def worker(which):
        k = (7, 40)[which]
        for _ in range(40):
            try:
                got = gm.match_batch(wl.query_ranges[:k], wl.query_poses[:k]).tobytes()
                key = "ok" if got == want[which] else "wrong"
            except api.LslamError as e:
                key = "refused" if e.code == -1 else "wrong"
            with lock:
                outcome[key] += 1

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert outcome["wrong"] == 0 and outcome["ok"] > 0, outcome
    assert gm.match_batch(wl.query_ranges[:7], wl.query_poses[:7]).tobytes() == want[0]  # still healthy afterwards
