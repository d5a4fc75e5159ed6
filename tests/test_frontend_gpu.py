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
