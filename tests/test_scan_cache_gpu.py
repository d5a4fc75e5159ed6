"""The device-side scan cache behind seam B1 (lslam_scan_cache_*, lslam_matcher_match_scan_cached): MatchScan with its base
scans named by id must return EXACTLY what the literal entry point lslam_matcher_match_scan returns for the same scans and
poses (same kernels on the same world points and FindValidPoints anchors) -- records compared byte for byte, correlation
grids byte for byte -- through everything Mapper::Process does to a scan's pose: first match (the speculative refresh at
the returned mean), a loop closure re-posing old scans, an anonymous query, and the configurations the clear-free
rebuild does not cover (wide smear kernels: the contiguous fallback).  One sequence is also checked against the CPU
oracle so that "equal to match_scan" is anchored to the reference, not only to ourselves."""
import numpy as np
import pytest

from lslam_amd import api, synth

pytestmark = pytest.mark.gpu


def _record(t):
    resp, pose, cov = t
    return np.concatenate([[resp], pose, cov.ravel()]).tobytes()


def _pair(ctx, laser=synth.Laser(), **cfg_kw):
    rt = cfg_kw.pop("range_threshold", 49.5)
    lp = api.laser_params(laser, rt)
    cfg = api.baseline_config(range_threshold=rt, **cfg_kw)
    return api.ScanMatcher(ctx, cfg, lp), api.ScanMatcher(ctx, cfg, lp), api.ScanCache(ctx, lp)


def _trajectory(n, seed):
    """n scans along a path (truth poses), noisy odometry, synthetic 1081-beam ranges."""
    wl = synth.make_match_workload(n_base=n, n_query=1, seed=seed)
    rng = np.random.default_rng(seed + 100)
    odo = wl.base_poses + rng.uniform(-1, 1, wl.base_poses.shape) * np.array([0.15, 0.15, 0.05])
    return wl.base_ranges, odo


def test_process_like_sequence_equals_match_scan(ctx, oracle_lib):
    """A running window as Mapper::Process drives it (Mapper.cpp:2040-2044): scan k is matched against the <= W scans before
    it at their CORRECTED poses, then takes the returned mean and joins the window."""
    plain, cached, cache = _pair(ctx)
    ranges, odo = _trajectory(40, 11)
    W = 12
    poses = np.zeros_like(odo)
    poses[0] = odo[0]
    cache.put(0, ranges[0])
    port = oracle_lib.PortKarto(oracle_lib.default_cfg(), oracle_lib.laser_struct(synth.Laser()))
    for k in range(1, len(ranges)):
        lo = max(0, k - W)
        ids = np.arange(lo, k)
        a = plain.MatchScan(ranges[k], odo[k], ranges[lo:k], poses[lo:k])
        b = cache.MatchScan(cached, ids, poses[lo:k], odo[k], query_id=k, query_ranges=ranges[k], takes_result_pose=True)
        assert _record(a) == _record(b), k
        if k % 7 == 0:
            assert np.array_equal(plain.GetCorrelationGrid(), cached.GetCorrelationGrid())
        if k in (5, 23):  # anchor to the reference's restatement
            mean, cov, resp = port.match_scan(ranges[lo:k], poses[lo:k], ranges[k], odo[k])
            assert np.abs(b[1] - mean).max() <= 1e-9 and abs(b[0] - resp) <= 1e-12
        poses[k] = b[1]
    c = cache.counters()
    assert c["matches"] == 39 and c["uploads"] == 40 and len(cache) == 40
    # every query's world points were prepared behind its own match at the pose it then took: apart from scan 0 no base
    # scan ever needed a refresh in front of a match
    assert c["speculated"] == 39 and c["refreshed"] == 1, c


def test_reposed_scans_named_twice_and_anonymous_query(ctx):
    plain, cached, cache = _pair(ctx)
    ranges, odo = _trajectory(16, 12)
    for i in range(16):
        cache.put(i, ranges[i])
    ids = np.arange(0, 12)
    a = plain.MatchScan(ranges[12], odo[12], ranges[:12], odo[:12])
    b = cache.MatchScan(cached, ids, odo[:12], odo[12], query_id=12)  # readings already cached: none passed
    assert _record(a) == _record(b)
    assert cache.counters()["refreshed"] == 12
    # a closed loop re-poses some scans (CorrectPoses / SetSensorPose): only those are refreshed
    moved = odo.copy()
    moved[3:7] += np.array([0.07, -0.04, 0.013])
    a = plain.MatchScan(ranges[13], odo[13], ranges[:12], moved[:12])
    b = cache.MatchScan(cached, ids, moved[:12], odo[13], query_id=13)
    assert _record(a) == _record(b)
    assert cache.counters()["refreshed"] == 16
    # the same call again: nothing to refresh, same record
    assert _record(cache.MatchScan(cached, ids, moved[:12], odo[13], query_id=13)) == _record(a)
    assert cache.counters()["refreshed"] == 16
    # a chain that names a scan twice, no penalty / no refinement (TryCloseLoop's coarse call, Mapper.cpp:991)
    twice = np.array([0, 1, 2, 2, 5, 9])
    a = plain.MatchScan(ranges[14], odo[14], ranges[twice], moved[twice], doPenalize=False, doRefineMatch=False)
    b = cache.MatchScan(cached, twice, moved[twice], odo[14], query_id=14, doPenalize=False, doRefineMatch=False)
    assert _record(a) == _record(b)
    # anonymous query (TryCloseLoop's temporary scan, Mapper.cpp:1008-1015): nothing is kept
    n = len(cache)
    a = plain.MatchScan(ranges[15], odo[15], ranges[:12], moved[:12])
    b = cache.MatchScan(cached, ids, moved[:12], odo[15], query_id=-1, query_ranges=ranges[15])
    assert _record(a) == _record(b) and len(cache) == n
    # the query among its own base scans (allowed when its readings are resident)
    a = plain.MatchScan(ranges[5], odo[5], ranges[:12], moved[:12])
    b = cache.MatchScan(cached, ids, moved[:12], odo[5], query_id=5)
    assert _record(a) == _record(b)
    # empty window: the reference matches against an empty grid
    a = plain.MatchScan(ranges[1], odo[1], ranges[:0], odo[:0])
    b = cache.MatchScan(cached, np.zeros(0, dtype=np.int64), odo[:0], odo[1], query_id=1)
    assert _record(a) == _record(b)


def test_errors_and_forget(ctx):
    plain, cached, cache = _pair(ctx)
    ranges, odo = _trajectory(4, 13)
    cache.put(0, ranges[0])
    with pytest.raises(api.LslamError) as e:
        cache.MatchScan(cached, [0, 7], odo[[0, 1]], odo[2], query_id=2, query_ranges=ranges[2])
    assert e.value.code == -1 and "not in the scan cache" in str(e.value)
    with pytest.raises(api.LslamError):
        cache.MatchScan(cached, [0], odo[:1], odo[3], query_id=3)  # neither cached nor given
    with pytest.raises(api.LslamError):
        cache.MatchScan(cached, [0, 0], np.array([odo[0], odo[1]]), odo[3], query_id=-1, query_ranges=ranges[3])  # two poses
    assert 0 in cache and 9 not in cache
    cache.forget(0)
    assert 0 not in cache
    cache.put(5, ranges[1])
    cache.put(6, ranges[2])
    cache.forget()  # everything
    assert len(cache) == 0
    # ids are the caller's: reuse after forget starts clean
    cache.put(0, ranges[1])
    a = plain.MatchScan(ranges[3], odo[3], ranges[1:2], odo[1:2])
    b = cache.MatchScan(cached, [0], odo[1:2], odo[3], query_id=-1, query_ranges=ranges[3])
    assert _record(a) == _record(b)
    # a cache made for another laser is refused
    other = api.ScanCache(ctx, api.laser_params(synth.Laser(n_ranges=360, angle_min=-np.pi, angle_increment=2 * np.pi / 360)))
    with pytest.raises(api.LslamError):
        other.MatchScan(cached, [], odo[:0], odo[3], query_id=-1, query_ranges=ranges[3][:360])


@pytest.mark.parametrize("kw", [dict(resolution=0.05, smear_deviation=0.45, search_size=1.0, range_threshold=20.0),   # 37x37 smear
                                dict(resolution=0.025, smear_deviation=0.03, search_size=0.5, range_threshold=20.0),  # 5x5
                                dict(search_size=8.0, resolution=0.05, range_threshold=20.0)])                        # loop-sized lattice
def test_other_configurations(ctx, kw):
    """Wide smear kernels take the listed scatter path (the cache gathers the window into a contiguous workspace for it);
    the loop matcher's lattice goes through the dense kernels."""
    plain, cached, cache = _pair(ctx, **dict(kw))
    ranges, odo = _trajectory(10, 14)
    for i in range(8):
        cache.put(i, ranges[i])
    ids = np.arange(8)
    for q in (8, 9):
        a = plain.MatchScan(ranges[q], odo[q], ranges[:8], odo[:8], doRefineMatch=kw.get("search_size", 1) < 8)
        b = cache.MatchScan(cached, ids, odo[:8], odo[q], query_id=q, query_ranges=ranges[q],
                            doRefineMatch=kw.get("search_size", 1) < 8, takes_result_pose=True)
        assert _record(a) == _record(b)
    assert np.array_equal(plain.GetCorrelationGrid(), cached.GetCorrelationGrid())


def test_cache_grows_past_its_first_allocation(ctx):
    """More scans than the first 256 slots: the resident arrays are re-allocated and carried over."""
    plain, cached, cache = _pair(ctx)
    ranges, odo = _trajectory(6, 15)
    for i in range(300):
        cache.put(i, ranges[i % 5])
    ids = np.array([0, 1, 2, 3, 4, 299, 257])
    base_r = ranges[[0, 1, 2, 3, 4, 299 % 5, 257 % 5]]
    base_p = np.concatenate([odo[:5], odo[[4, 2]] + 0.01])
    a = plain.MatchScan(ranges[5], odo[5], base_r, base_p)
    b = cache.MatchScan(cached, ids, base_p, odo[5], query_id=1000, query_ranges=ranges[5])
    assert _record(a) == _record(b)
    assert len(cache) == 301 and cache.counters()["resident_bytes"] >= 512 * 1081 * 24


def test_prepare_moves_the_refresh_off_the_next_match(ctx):
    """lslam_scan_cache_prepare: a scan's world points + anchors at the pose the caller EXPECTS it to have next (the drop-in
    layer: the pose AddEdges gives the scan Mapper::Process has just matched), enqueued behind the match.  Right guess: the
    next match needs no refresh; wrong guess: it refreshes again; either way the records equal lslam_matcher_match_scan's."""
    plain, cached, cache = _pair(ctx)
    ranges, odo = _trajectory(8, 16)
    for i in range(6):
        cache.put(i, ranges[i])
    ids = np.arange(6)
    cache.MatchScan(cached, ids, odo[:6], odo[6], query_id=6, query_ranges=ranges[6])
    base = cache.counters()
    assert base["refreshed"] == 6 and base["speculated"] == 0
    moved = odo[:7].copy()
    moved[6] += np.array([1e-3, -2e-3, 5e-4])
    cache.prepare(6, moved[6])                       # right guess
    cache.prepare(6, moved[6])                       # again: nothing to do
    a = plain.MatchScan(ranges[7], odo[7], ranges[:7], moved)
    b = cache.MatchScan(cached, np.arange(7), moved, odo[7], query_id=7, query_ranges=ranges[7])
    c = cache.counters()
    assert _record(a) == _record(b) and c["refreshed"] == 6 and c["speculated"] == 1
    cache.prepare(3, moved[3] + 1e-6)                # wrong guess: scan 3 is still where it was
    b = cache.MatchScan(cached, np.arange(7), moved, odo[7], query_id=7)
    c = cache.counters()
    assert _record(a) == _record(b) and c["refreshed"] == 7 and c["speculated"] == 2
    with pytest.raises(api.LslamError):
        cache.prepare(99, odo[0])


def test_long_scans_take_the_fallback_paths(ctx):
    """2400 beams per scan: the anchor chains no longer fit LDS, so the cache refreshes world points only
    (k_scan_prep), the clear-free rebuild refuses the window and the cache gathers it into a contiguous workspace for the
    listed path -- still byte-identical to lslam_matcher_match_scan."""
    laser = synth.Laser(n_ranges=2400, angle_min=-np.pi * 0.75, angle_increment=1.5 * np.pi / 2400)
    plain, cached, cache = _pair(ctx, laser=laser, range_threshold=20.0)
    assert plain.num_beams == 2400
    wl = synth.make_match_workload(n_base=7, n_query=1, seed=17, laser=laser)
    odo = wl.base_poses + np.array([0.05, -0.03, 0.01])
    for i in range(5):
        cache.put(i, wl.base_ranges[i])
    for q in (5, 6):
        a = plain.MatchScan(wl.base_ranges[q], odo[q], wl.base_ranges[:5], odo[:5])
        b = cache.MatchScan(cached, np.arange(5), odo[:5], odo[q], query_id=q, query_ranges=wl.base_ranges[q], takes_result_pose=True)
        assert _record(a) == _record(b)
    assert np.array_equal(plain.GetCorrelationGrid(), cached.GetCorrelationGrid())
