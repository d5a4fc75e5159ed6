"""The one-launch match of ONE scan (LSLAM_OPT_LONE_KERNEL, k_match_lone): coarse responses -> coarse reduce -> fine
responses -> fine reduce behind device-side hand-overs instead of four dependent launches.  It runs the four-kernel chain's
own wave / block functions on the same beam slices, so its records must be byte-identical to the chain's (which
test_matcher_gpu.py holds to the oracle and the reference): complete MatchScan calls, batches of one, the streaming
front-end scan by scan, with and without penalties, with unreadable beams, at every waves-per-block setting, and over
thousands of consecutive launches (the hand-over counters are monotonic and never reset)."""
import numpy as np
import pytest

from lslam_amd import api, synth

pytestmark = pytest.mark.gpu


def _matcher(ctx, wl, **cfg):
    gm = api.ScanMatcher(ctx, api.baseline_config(**cfg), api.laser_params(wl.laser, cfg.get("range_threshold", 49.5)))
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    return gm


@pytest.mark.parametrize("waves", [4, 8, 16, 104])  # 104: one task wave per block of four
def test_batches_of_one_equal_the_four_kernel_chain(ctx, waves):
    wl = synth.make_match_workload(n_base=30, n_query=40, seed=61, query_spread=2.0)
    wl.query_ranges[np.random.default_rng(2).random(wl.query_ranges.shape) < 0.02] = np.nan
    gm = _matcher(ctx, wl)
    for i in range(40):
        for pen in (True, False):
            gm.set_option("lone_kernel", 0)
            want = gm.match_batch(wl.query_ranges[i:i + 1], wl.query_poses[i:i + 1], doPenalize=pen)
            gm.set_option("lone_kernel", waves)
            before = gm.lone_kernel_launches
            got = gm.match_batch(wl.query_ranges[i:i + 1], wl.query_poses[i:i + 1], doPenalize=pen)
            assert gm.lone_kernel_launches == before + 1
            assert got.tobytes() == want.tobytes(), (i, pen)
    assert (want["status"] == 0).all()
    # what the kernel does not cover keeps the chain, silently: two scans, a coarse-only match
    before = gm.lone_kernel_launches
    two = gm.match_batch(wl.query_ranges[:2], wl.query_poses[:2])
    coarse_only = gm.match_batch(wl.query_ranges[:1], wl.query_poses[:1], doRefineMatch=False)
    assert gm.lone_kernel_launches == before
    gm.set_option("lone_kernel", 0)
    assert two.tobytes() == gm.match_batch(wl.query_ranges[:2], wl.query_poses[:2]).tobytes()
    assert coarse_only.tobytes() == gm.match_batch(wl.query_ranges[:1], wl.query_poses[:1], doRefineMatch=False).tobytes()
    gm.close()


@pytest.mark.parametrize("waves", [4, 16])
def test_complete_match_scan_calls(ctx, waves):
    wl = synth.make_match_workload(n_base=70, n_query=12, seed=62, query_spread=1.5)
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(wl.laser))
    for i in range(12):
        gm.set_option("lone_kernel", 0)
        want = gm.MatchScan(wl.query_ranges[i], wl.query_poses[i], wl.base_ranges, wl.base_poses)
        gm.set_option("lone_kernel", waves)
        before = gm.lone_kernel_launches
        got = gm.MatchScan(wl.query_ranges[i], wl.query_poses[i], wl.base_ranges, wl.base_poses)
        assert gm.lone_kernel_launches == before + 1
        assert got[0] == want[0] and got[1].tobytes() == want[1].tobytes() and got[2].tobytes() == want[2].tobytes(), i
    gm.close()


def test_the_shipped_indoor_lattice(ctx):
    # lesson6/config/mapper_params.yaml without the expansion passes: 16 x 16 x 21, the <4,8> coarse instantiation
    import math
    cfg = dict(search_size=0.3, resolution=0.01, smear_deviation=0.03, range_threshold=12.0,
               distance_variance_penalty=0.5 ** 2, angle_variance_penalty=0.1 ** 2)
    laser = synth.Laser(range_max=30.0)
    world = synth.arena(size=24.0, n_axis=8, n_rot=3, seed=12)
    wl = synth.make_match_workload(n_base=20, n_query=10, seed=12, laser=laser, world=world, err_xy=0.08,
                                   err_th=math.radians(6.0), query_spread=0.5)
    gm = _matcher(ctx, wl, **cfg)
    for i in range(10):
        gm.set_option("lone_kernel", 0)
        want = gm.match_batch(wl.query_ranges[i:i + 1], wl.query_poses[i:i + 1])
        gm.set_option("lone_kernel", 8)
        before = gm.lone_kernel_launches
        got = gm.match_batch(wl.query_ranges[i:i + 1], wl.query_poses[i:i + 1])
        assert gm.lone_kernel_launches == before + 1
        assert got.tobytes() == want.tobytes(), i
    assert (want["status"] == 0).all()
    gm.close()


def test_streaming_front_end_and_thousands_of_launches(ctx):
    laser = synth.Laser()
    path = synth.rings_trajectory(400)
    world = synth.arena_around_path(path, size=100.0, n_axis=30, n_rot=10, seed=6)
    odom = synth.drifting_odometry(path, scale=1.01, sigma_xy=0.004, sigma_th=0.0015, seed=6)
    rng = np.random.default_rng(6)
    scans = [synth.ranges_to_f64(synth.cast_scan(world, p, laser, 0.01, 0.0, rng)) for p in path]
    poses = {}
    for waves in (0, 4):
        gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(laser))
        gm.set_option("lone_kernel", waves)
        fe = api.FrontEnd(gm)
        for rep in range(5 if waves else 1):  # 2000 launches on one set of counters
            fe.reset()
            for r, o in zip(scans, odom):
                fe.Process(r, o)
        poses[waves] = np.stack([fe.scan_pose(i) for i in range(fe.num_scans())])
        if waves:
            assert gm.lone_kernel_launches >= 5 * 390
        fe.close()
        gm.close()
    assert poses[4].tobytes() == poses[0].tobytes()


def test_option_values(ctx):
    wl = synth.make_match_workload(n_base=10, n_query=1, seed=64)
    gm = _matcher(ctx, wl)
    opt = 9  # LSLAM_OPT_LONE_KERNEL
    assert gm.L.lslam_matcher_get_option(gm.h, opt) == 0  # off by default: the four-launch chain is the faster one
    for v in (4, 8, 16, 104, 108, 116, 0):
        gm.set_option("lone_kernel", v)
        assert gm.L.lslam_matcher_get_option(gm.h, opt) == v
    for bad in (1, 5, 32, 101, 132, -4):
        with pytest.raises(api.LslamError):
            gm.set_option("lone_kernel", bad)
    assert gm.L.lslam_matcher_get_option(gm.h, opt) == 0
    gm.close()
