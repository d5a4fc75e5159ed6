"""Speculative anchor chains (k_anchor_chain / anchor_spec_block): the streaming front-end works out FindValidPoints' anchors
(Mapper.cpp:774-787) of the newest scan beside the scan's own match, on the points at the pose the match STARTS from, and
takes them over at the final pose only when that is provably the chain of the final points.  The rows it leaves in the
ring must be the rows of the plain path (LSLAM_FE_SPEC_CHAIN=0: the chain worked out on the final points, the path
test_frontend_gpu.py holds to the reference) for every scan -- with readings that hit nothing (+inf: the reference's
unfiltered points carry them), unreadable beams (NaN), walls a few centimetres from the threshold, pose-graph look-ahead
on and off -- and so must every pose."""
import os

import numpy as np
import pytest

from lslam_amd import api, synth

pytestmark = pytest.mark.gpu


def _scans(n, seed, dropout=0.01, nan_share=0.0):
    laser = synth.Laser()
    path = synth.rings_trajectory(n)
    world = synth.arena_around_path(path, size=100.0, n_axis=30, n_rot=10, seed=seed)
    odom = synth.drifting_odometry(path, scale=1.01, sigma_xy=0.004, sigma_th=0.0015, seed=seed)
    rng = np.random.default_rng(seed)
    scans = []
    for p in path:
        r = synth.ranges_to_f64(synth.cast_scan(world, p, laser, 0.01, dropout, rng))
        if nan_share:
            r[rng.random(r.shape) < nan_share] = np.nan
        scans.append(r)
    return laser, scans, odom


def _run(ctx, laser, scans, odom, spec_chain, many=False, **cfg):
    os.environ["LSLAM_FE_SPEC_CHAIN"] = "1" if spec_chain else "0"
    try:
        gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(laser))
        fe = api.FrontEnd(gm, config=api.frontend_config(**cfg)) if cfg else api.FrontEnd(gm)
    finally:
        del os.environ["LSLAM_FE_SPEC_CHAIN"]
    if many:
        fe.ProcessMany(np.stack(scans), np.stack(odom))
    else:
        for r, o in zip(scans, odom):
            fe.Process(r, o)
    n = fe.num_scans()
    rows = [fe.anchor_row(i) for i in range(n)]
    poses = np.stack([fe.scan_pose(i) for i in range(n)])
    stats = fe.spec_chain_stats()
    fe.close()
    gm.close()
    return rows, poses, stats


@pytest.mark.parametrize("seed,dropout,nan_share", [(6, 0.01, 0.0), (7, 0.05, 0.02), (8, 0.0, 0.0)])
def test_rows_and_poses_equal_the_plain_path(ctx, seed, dropout, nan_share):
    laser, scans, odom = _scans(300, seed, dropout, nan_share)
    assert any(np.isinf(r).any() for r in scans) or dropout == 0.0
    want_rows, want_poses, off = _run(ctx, laser, scans, odom, False)
    got_rows, got_poses, on = _run(ctx, laser, scans, odom, True)
    assert off["handed"] == 0
    assert on["handed"] >= len(got_rows) - 2  # every scan but the first (no match) is handed its chain
    assert on["recomputed"] <= on["handed"] // 4  # ... and nearly all of them can take it over
    assert len(got_rows) == len(want_rows)
    for i, (g, w) in enumerate(zip(got_rows, want_rows)):
        assert np.array_equal(g, w), i
    assert got_poses.tobytes() == want_poses.tobytes()


def test_with_the_pose_graph_and_look_ahead(ctx):
    laser, scans, odom = _scans(400, 9)
    cfg = dict(scan_buffer_size=70, scan_buffer_maximum_scan_distance=20.0, do_loop_closing=1, link_scan_maximum_distance=1.5,
               loop_search_maximum_distance=3.0, loop_match_minimum_chain_size=10)
    want_rows, want_poses, _ = _run(ctx, laser, scans, odom, False, **cfg)
    for many in (False, True):
        got_rows, got_poses, on = _run(ctx, laser, scans, odom, True, many=many, **cfg)
        assert on["handed"] > 300
        for i, (g, w) in enumerate(zip(got_rows, want_rows)):
            assert np.array_equal(g, w), (many, i)
        assert got_poses.tobytes() == want_poses.tobytes()


def test_a_scan_whose_points_change_class_is_recomputed(ctx):
    # every other beam returns nothing: the +inf points' signs depend on the heading, and a heading correction that moves a
    # quadrant boundary across one of them must send the kernel down the plain path (same rows either way)
    laser, scans, odom = _scans(120, 10, dropout=0.5)
    want_rows, want_poses, _ = _run(ctx, laser, scans, odom, False)
    got_rows, got_poses, on = _run(ctx, laser, scans, odom, True)
    for i, (g, w) in enumerate(zip(got_rows, want_rows)):
        assert np.array_equal(g, w), i
    assert got_poses.tobytes() == want_poses.tobytes()
    assert on["handed"] > 100


def test_the_shipped_indoor_configuration_with_response_expansion(ctx):
    # lesson6/config/mapper_params.yaml: 0.01 m cells, 16 x 16 x 21 lattice, expansion passes on, 12 m range threshold (the
    # readings beyond it are points all the same: LocalizedRangeScan::Update's unfiltered list, Karto.h:5379-5404)
    import math
    laser = synth.Laser(range_max=30.0)
    world = synth.arena(size=24.0, n_axis=8, n_rot=3, seed=12)
    path = synth.trajectory(world, 150, step=0.05, max_turn=math.radians(2.0), seed=12)
    odom = synth.drifting_odometry(path, scale=1.005, sigma_xy=0.001, sigma_th=0.0005, seed=12)
    rng = np.random.default_rng(12)
    scans = [synth.ranges_to_f64(synth.cast_scan(world, p, laser, 0.005, 0.01, rng)) for p in path]
    out = {}
    for on in (0, 1):
        os.environ["LSLAM_FE_SPEC_CHAIN"] = str(on)
        try:
            gm = api.ScanMatcher(ctx, api.baseline_config(search_size=0.3, resolution=0.01, smear_deviation=0.03, use_response_expansion=1,
                                                          distance_variance_penalty=0.25, angle_variance_penalty=0.01,
                                                          range_threshold=12.0), api.laser_params(laser, 12.0))
            fe = api.FrontEnd(gm, scan_buffer_size=30, min_travel_distance=0.02, min_travel_heading=math.radians(1.0))
        finally:
            del os.environ["LSLAM_FE_SPEC_CHAIN"]
        for r, o in zip(scans, odom):
            fe.Process(r, o)
        n = fe.num_scans()
        out[on] = ([fe.anchor_row(i) for i in range(n)], np.stack([fe.scan_pose(i) for i in range(n)]), fe.spec_chain_stats())
        fe.close()
        gm.close()
    assert len(out[1][0]) > 100 and out[1][2]["handed"] > 100
    for i, (g, w) in enumerate(zip(out[1][0], out[0][0])):
        assert np.array_equal(g, w), i
    assert out[1][1].tobytes() == out[0][1].tobytes()
