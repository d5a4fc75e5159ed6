"""Streaming front-end (BASELINE config 5 in miniature): per-scan correlative match against the
device-resident running window + incremental log-odds map update, composed as SURVEY.md §7
prescribes -- poses from the Karto Mapper::Process path, map from the Hector update with each
accepted pose -- and compared with the same composition of the CPU oracles."""
import math
import pathlib

import numpy as np
import pytest

from lslam_amd import api, synth

pytestmark = pytest.mark.gpu
G = pathlib.Path(__file__).resolve().parent / "golden"


def test_frontend_reproduces_reference_process_vectors(ctx):
    """Golden vectors produced by karto::Mapper::Process itself (tests/golden/make_golden.py)."""
    d = np.load(G / "karto_frontend_golden.npz")
    laser = synth.Laser()
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(laser))
    fe = api.FrontEnd(gm, scan_buffer_size=8, scan_buffer_max_distance=3.0)
    n_proc = 0
    for r, o, ok, pose in zip(d["ranges"], d["odom"], d["processed"], d["corrected"]):
        ok2, pose2, _, _ = fe.Process(r.astype(np.float64), o)
        assert ok2 == bool(ok)
        assert np.abs(pose2 - pose).max() <= 1e-9
        n_proc += ok2
    assert n_proc > 10 and fe.running_scans() <= 8


def test_streaming_match_plus_map_update(ctx, oracle_lib):
    laser = synth.Laser()
    world = synth.arena(size=60.0, n_axis=16, n_rot=6, seed=23)
    path = synth.trajectory(world, 60, step=0.25, seed=23, bounds=20.0)
    odom = synth.perturb(path, 0.05, math.radians(2.0), 24)
    rng = np.random.default_rng(25)
    cfg = oracle_lib.default_cfg(scan_buffer_size=20, scan_buffer_max_scan_distance=6.0)
    port = oracle_lib.PortKarto(cfg, oracle_lib.laser_struct(laser))
    port.frontend()
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(laser))
    fe = api.FrontEnd(gm, scan_buffer_size=20, scan_buffer_max_distance=6.0)
    n, cell = 2000, 0.025  # config 5 uses 4000x4000 @ 0.025 m; same cell size, a quarter of the area
    off = (n * cell * 0.5, n * cell * 0.5)
    cmap = oracle_lib.PortHector(n, n, cell, off)
    gmap = api.OccGridMap(ctx, n, n, cell, off)
    for m in (cmap, gmap):
        m.setUpdateOccupiedFactor(0.9)
    origin = path[0].copy()
    n_proc = 0
    for t, o in zip(path, odom):
        r32 = synth.cast_scan(world, t, laser, 0.01, 0.01, rng)
        r = synth.ranges_to_f64(r32)
        ok_c, pose_c, cov_c, resp_c = port.process(r, o)
        ok_g, pose_g, cov_g, resp_g = fe.Process(r, o)
        assert ok_c == ok_g
        assert np.abs(pose_c - pose_g).max() <= 1e-9
        if not ok_c:
            continue
        n_proc += 1
        assert abs(resp_c - resp_g) <= 1e-12
        assert np.abs(cov_c - cov_g).max() <= 1e-9 * max(1.0, np.abs(cov_c).max())
        assert port.running_scans() == fe.running_scans()
        # incremental map update with the accepted pose (map frame = first pose at the map centre)
        pts = synth.hector_points(r32, laser, 1.0 / cell, use_max=20.0)
        pose_map = (pose_g - np.array([origin[0], origin[1], 0.0])).astype(np.float32)
        cmap.updateByScan(pts, (0.0, 0.0), (pose_c - np.array([origin[0], origin[1], 0.0])).astype(np.float32))
        gmap.updateByScan(pts, (0.0, 0.0), pose_map)
    assert n_proc >= 50
    a, b = cmap.logodds(), gmap.logodds()
    assert (a != 0).sum() > 50000
    assert a.tobytes() == b.tobytes()
    assert np.array_equal(cmap.occupancy_i8(), gmap.occupancy_i8())
    # drift stays small: matched poses track the truth much better than raw odometry would
    assert np.hypot(*(pose_g[:2] - path[-1][:2])) < 0.5


def _loop_frontend(ctx, d, laser):
    cfg = api.frontend_config(scan_buffer_size=int(d["cfg_scan_buffer_size"]),
                              scan_buffer_maximum_scan_distance=float(d["cfg_scan_buffer_max_scan_distance"]),
                              do_loop_closing=int(d["cfg_do_loop_closing"]),
                              link_scan_maximum_distance=float(d["cfg_link_scan_maximum_distance"]),
                              loop_search_maximum_distance=float(d["cfg_loop_search_maximum_distance"]))
    gm = api.ScanMatcher(ctx, api.baseline_config(range_threshold=float(d["range_threshold"])),
                         api.laser_params(laser, float(d["range_threshold"])))
    return gm, api.FrontEnd(gm, config=cfg)


def test_closed_loop_trajectory_reproduces_reference_graph(ctx):
    """Mapper::Process WITH its pose graph on a trajectory that revisits (golden vectors from karto::Mapper itself):
    LinkNearChains' extra matches enter the weighted mean, TryCloseLoop runs the 81x81x21 loop matcher and re-poses
    the closing scan.  Same poses (<= 1e-9), same edge count after every scan."""
    d = np.load(G / "karto_loop_golden.npz")
    laser = synth.Laser()
    gm, fe = _loop_frontend(ctx, d, laser)
    for i, (r, o) in enumerate(zip(d["ranges"], d["odom"])):
        ok, pose, _, _ = fe.Process(r.astype(np.float64), o)
        assert ok == bool(d["processed"][i])
        assert np.abs(pose - d["corrected"][i]).max() <= 1e-9, (i, pose, d["corrected"][i])
        assert fe.stats()["edges"] == int(d["edges"][i]), i
    st = fe.stats()
    assert st["chain_matches"] > 0 and st["loop_coarse_matches"] > 0 and st["loops_closed"] > 0, st
    final = np.stack([fe.scan_pose(i) for i in range(fe.num_scans())])
    assert np.abs(final - d["final_poses"]).max() <= 1e-9
    # the loop is worth closing: raw odometry has drifted by more than the corrected poses
    assert np.hypot(*(final[-1][:2] - d["truth"][-1][:2])) < 0.1 < np.hypot(*(d["odom"][-1][:2] - d["truth"][-1][:2]))


def test_closed_loop_against_reference_live(ctx, oracle_lib):
    """Same, on a different trajectory, against the reference's own Mapper running beside it (oracle/_ref travels to
    the GPU box) with the lesson6 yaml's graph distances."""
    if not oracle_lib.have_ref():
        pytest.skip("oracle/_ref not built")
    laser = synth.Laser()
    world = synth.arena(size=40.0, n_axis=10, n_rot=4, seed=5)
    path = synth.loop_trajectory(260, w=8.0, h=5.0, step=0.2, origin=(-4.0, -2.5))
    odom = synth.drifting_odometry(path, scale=1.03, seed=9)
    kw = dict(scan_buffer_size=25, scan_buffer_max_scan_distance=6.0, do_loop_closing=1, link_scan_maximum_distance=1.5,
              loop_search_maximum_distance=3.0, loop_match_minimum_chain_size=8)
    ref = oracle_lib.RefKarto(oracle_lib.default_cfg(**kw), oracle_lib.laser_struct(laser, 20.0))
    gm = api.ScanMatcher(ctx, api.baseline_config(range_threshold=20.0), api.laser_params(laser, 20.0))
    fe = api.FrontEnd(gm, config=api.frontend_config(
        scan_buffer_size=25, scan_buffer_maximum_scan_distance=6.0, do_loop_closing=1, link_scan_maximum_distance=1.5,
        loop_search_maximum_distance=3.0, loop_match_minimum_chain_size=8))
    worst = 0.0
    for i, (t, o) in enumerate(zip(path, odom)):
        r = synth.ranges_to_f64(synth.cast_scan(world, t, laser, 0.01, 0.01, np.random.default_rng([43, i])))
        ok_c, pose_c = ref.process(r, o)
        ok_g, pose_g, _, _ = fe.Process(r, o)
        assert ok_c == ok_g
        worst = max(worst, float(np.abs(pose_c - pose_g).max()))
        assert worst <= 1e-9, (i, pose_c, pose_g)
        assert ref.graph_stats()[1] == fe.stats()["edges"], i
    assert fe.stats()["loops_closed"] > 0
    print("closed-loop run vs the reference's Mapper: max pose difference", worst, fe.stats())


def test_rejected_scan_reports_identity_covariance(ctx):
    laser = synth.Laser()
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(laser))
    fe = api.FrontEnd(gm)
    world = synth.square_room(10.0)
    r = synth.ranges_to_f64(synth.cast_scan(world, (0, 0, 0), laser))
    assert fe.Process(r, (0.0, 0.0, 0.0))[0]
    ok, pose, cov, resp = fe.Process(r, (0.01, 0.0, 0.0))  # has not moved enough (Mapper.cpp:2028-2031)
    assert not ok and np.array_equal(cov, np.eye(3)) and resp == 0.0
    ok, _, _, _ = fe.Process(r, (0.01, 0.0, 0.0), time_s=4000.0)  # MinimumTimeInterval (3600 s) has passed
    assert ok


@pytest.mark.parametrize("variant", ["laser_offset", "response_expansion", "sensor_reference_pose"])
def test_graph_frontend_variants_against_reference_live(ctx, oracle_lib, variant):
    """The pose-graph front-end under the parameters a lesson6 deployment changes: a laser mounted off the base centre
    (Sensor::SetOffsetPose, karto_slam.cc:387-389: HasMovedEnough, SetSensorPose and the barycentre all go through it),
    response expansion on (use_response_expansion: true in mapper_params.yaml), and scan reference poses taken from the
    sensor instead of the barycentre (use_scan_barycenter: false) -- each against the reference's Mapper beside it."""
    if not oracle_lib.have_ref():
        pytest.skip("oracle/_ref not built")
    laser = synth.Laser()
    world = synth.arena(size=40.0, n_axis=10, n_rot=4, seed=5)
    path = synth.loop_trajectory(150, w=6.0, h=4.0, step=0.2, origin=(-3.0, -2.0))
    odom = synth.drifting_odometry(path, scale=1.02, seed=11)
    offset = (0.18, -0.05, 0.04) if variant == "laser_offset" else (0.0, 0.0, 0.0)
    expansion = 1 if variant == "response_expansion" else 0
    bary = 0 if variant == "sensor_reference_pose" else 1
    kw = dict(scan_buffer_size=20, scan_buffer_max_scan_distance=5.0, do_loop_closing=1, link_scan_maximum_distance=1.5,
              loop_search_maximum_distance=3.0, loop_match_minimum_chain_size=6, use_scan_barycenter=bary,
              use_response_expansion=expansion)
    ref = oracle_lib.RefKarto(oracle_lib.default_cfg(**kw), oracle_lib.laser_struct(laser, 20.0, offset))
    gm = api.ScanMatcher(ctx, api.baseline_config(range_threshold=20.0, use_response_expansion=expansion),
                         api.laser_params(laser, 20.0, offset))
    fe = api.FrontEnd(gm, config=api.frontend_config(
        scan_buffer_size=20, scan_buffer_maximum_scan_distance=5.0, do_loop_closing=1, link_scan_maximum_distance=1.5,
        loop_search_maximum_distance=3.0, loop_match_minimum_chain_size=6, use_scan_barycenter=bary))
    for i, (t, o) in enumerate(zip(path, odom)):
        # the scan is cast from where the LASER is: robot pose composed with the mounting offset
        c, s_ = math.cos(t[2]), math.sin(t[2])
        lp = (t[0] + c * offset[0] - s_ * offset[1], t[1] + s_ * offset[0] + c * offset[1], t[2] + offset[2])
        r = synth.ranges_to_f64(synth.cast_scan(world, lp, laser, 0.01, 0.01, np.random.default_rng([47, i])))
        ok_c, pose_c = ref.process(r, o)
        ok_g, pose_g, _, _ = fe.Process(r, o)
        assert ok_c == ok_g, i
        assert np.abs(pose_c - pose_g).max() <= 1e-9, (variant, i, pose_c, pose_g)
        assert ref.graph_stats()[1] == fe.stats()["edges"], (variant, i)
    assert fe.stats()["edges"] > fe.stats()["scans"]
