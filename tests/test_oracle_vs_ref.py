"""Pins the CPU oracle: the plain-C restatement (oracle/karto_oracle.c) against the reference's own
open_karto compiled unmodified (oracle/_ref, built by oracle/Makefile from /root/reference), and
both against the known answers recorded in SURVEY.md §8(c).  CPU only."""
import math

import numpy as np
import pytest

from lslam_amd import synth


@pytest.fixture(scope="module")
def po(oracle_lib):
    if not oracle_lib.have_ref():
        pytest.skip("oracle/_ref/libkarto_ref.so not built (needs /root/reference)")
    return oracle_lib


def pair(po, laser=synth.Laser(), range_threshold=49.5, offset=(0, 0, 0), **cfg):
    c = po.default_cfg(**cfg)
    l = po.laser_struct(laser, range_threshold, offset)
    return po.RefKarto(c, l), po.PortKarto(c, l)


def test_survey_known_answers(po):
    """SURVEY.md §8(c) KATs, reproduced by the reference build itself."""
    laser = synth.Laser()
    ref, port = pair(po)
    gi = ref.grid_info()
    assert (gi["width"], gi["height"], gi["stride"], gi["roi_w"], gi["roi_h"]) == (2005, 2005, 2008, 2001, 2001)
    assert gi["height"] * gi["stride"] == 4026040
    assert ref.num_beams == 1081 == port.num_beams
    assert ref.kernel().tolist() == [[6, 25, 6], [25, 100, 25], [6, 25, 6]]
    assert [ref.round(v) for v in (0.5, -0.5, 2.5)] == [1.0, -1.0, 3.0]
    assert ref.L.kref_sizeof_pose2() == 24 and ref.L.kref_sizeof_matrix3() == 72
    world = synth.square_room(10.0)
    truth = [(0, 0, 0), (0.31, 0.12, 0.06), (0.62, 0.2, 0.15)]
    odom = [(0, 0, 0), (0.25, 0.05, 0.02), (0.55, 0.3, 0.10)]
    expect = [(0, 0, 0), (0.300000, 0.100000, 0.058390), (0.590184, 0.211330, 0.148860)]
    port.frontend()
    for t, o, e in zip(truth, odom, expect):
        r = synth.ranges_to_f64(synth.cast_scan(world, t, laser))
        ok, pose = ref.process(r, o)
        ok2, pose2, _, _ = port.process(r, o)
        assert ok and ok2
        assert np.allclose(pose, e, atol=1e-6)
        assert np.array_equal(pose, pose2)
    occ, _ = ref.occupancy_grid(0.05)
    assert occ.shape == (401, 401)


@pytest.mark.parametrize("res,row", [(0.025, [25, 71, 100, 71, 25]), (0.01, [14, 25, 41, 61, 80, 95, 100, 95, 80, 61, 41, 25, 14])])
def test_smear_kernels(po, res, row):
    ref, port = pair(po, range_threshold=12.0, resolution=res, search_size=0.3)
    k = ref.kernel()
    assert k[len(k) // 2].tolist() == row
    assert np.array_equal(k, port.kernel())
    assert ref.grid_info()["stride"] == port.grid_info()["stride"]
    if res == 0.01:
        assert (ref.grid_info()["width"], ref.grid_info()["stride"]) == (2445, 2448)


def test_point_readings_and_valid_points(po, workload):
    ref, port = pair(po)
    wl = workload
    for s in range(5):
        a = ref.point_readings(wl.base_ranges[s], wl.base_poses[s])
        b = port.point_readings(wl.base_ranges[s], wl.base_poses[s])
        both_nan = np.isnan(a) & np.isnan(b)
        assert np.array_equal(a[~both_nan], b[~both_nan])  # inf ranges give NaN/inf points in both
        vp = wl.query_poses[0][:2]
        va = ref.find_valid_points(wl.base_ranges[s], wl.base_poses[s], vp)
        vb = port.find_valid_points(b, vp)
        assert np.array_equal(va, vb)


def test_match_scan_bit_exact(po, workload):
    ref, port = pair(po)
    wl = workload
    for q in range(len(wl.query_ranges)):
        rp, rc, rr = ref.match(wl.base_ranges, wl.base_poses, wl.query_ranges[q], wl.query_poses[q])
        pp, pc, pr = port.match_scan(wl.base_ranges, wl.base_poses, wl.query_ranges[q], wl.query_poses[q])
        assert np.array_equal(rp, pp) and np.array_equal(rc, pc) and rr == pr
        assert np.array_equal(ref.grid(), port.grid())
        assert np.array_equal(ref.grid_info()["offset"], port.grid_info()["offset"])


def test_tables_probs_and_response_sums(po, workload):
    ref, port = pair(po)
    wl = workload
    q, qp = wl.query_ranges[2].copy(), wl.query_poses[2]
    q[3] = np.nan
    q[900] = np.inf
    ref.match(wl.base_ranges, wl.base_poses, q, qp, True, False)       # coarse only: tables = coarse tables
    port.match_scan(wl.base_ranges, wl.base_poses, q, qp, True, False)
    t_ref, ang = ref.tables()
    t_port = port.compute_offsets(q, qp, qp[2], 0.349, 0.0349)
    assert t_ref.shape == (21, 1081)
    assert np.array_equal(t_ref, t_port)
    assert np.array_equal(ref.probs(), port.probs())
    gi = ref.grid_info()
    rng = np.random.default_rng(0)
    for _ in range(50):
        a = int(rng.integers(0, 21))
        pos = int(rng.integers(0, gi["height"] * gi["stride"]))
        assert ref.get_response(a, pos) == port.response_sum(t_port[a], pos) / (1081 * 100)


def test_shared_grid_mode(po, workload_spread):
    ref, port = pair(po)
    wl = workload_spread
    ref.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
    port.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
    assert np.array_equal(ref.grid(), port.grid())
    _, poses, covs, resp = ref.match_fixed_grid(wl.query_ranges, wl.query_poses)
    for i in range(len(resp)):
        m, c, r = port.match(wl.query_ranges[i], wl.query_poses[i])
        assert np.array_equal(m, poses[i]) and np.array_equal(c, covs[i]) and r == resp[i]


def test_response_expansion_and_flags(po, workload):
    ref, port = pair(po, use_response_expansion=1)
    wl = workload
    n = wl.query_ranges.shape[1]
    rot = wl.truth_poses[0].copy()
    rot[2] = math.remainder(rot[2] + math.radians(45.0), 2 * math.pi)
    for q, qp in ((wl.query_ranges[0], rot), (np.full(n, np.nan), wl.query_poses[1])):
        rp, rc, rr = ref.match(wl.base_ranges, wl.base_poses, q, qp)
        pp, pc, pr = port.match_scan(wl.base_ranges, wl.base_poses, q, qp)
        assert np.array_equal(rp, pp) and np.array_equal(rc, pc) and rr == pr


def test_laser_offset_and_1080_beams(po):
    """A LaserScan's true angle_max makes karto use 1080 of 1081 ranges (Karto.h:4152-4161);
    a mounted-off-centre laser exercises GetSensorAt / SetSensorPose (Karto.h:5280-5313)."""
    laser = synth.Laser(karto_uses_all_ranges=False)
    off = (0.15, -0.04, 0.2)
    ref, port = pair(po, laser=laser, offset=off)
    assert ref.num_beams == port.num_beams == 1080
    wl = synth.make_match_workload(n_base=6, n_query=3, seed=8, laser=laser)
    sposes = np.stack([port.sensor_pose_from_robot(p) for p in wl.base_poses])
    for q in range(3):
        rp, rc, rr = ref.match(wl.base_ranges, wl.base_poses, wl.query_ranges[q], wl.query_poses[q])
        pp, pc, pr = port.match_scan(wl.base_ranges, sposes, wl.query_ranges[q],
                                     port.sensor_pose_from_robot(wl.query_poses[q]))
        assert np.array_equal(rp, pp) and np.array_equal(rc, pc) and rr == pr


def test_streaming_front_end(po):
    """Mapper::Process (travel gating, running window, SetSensorPose) over a short trajectory."""
    laser = synth.Laser()
    ref, port = pair(po, scan_buffer_size=8, scan_buffer_max_scan_distance=3.0)
    port.frontend()
    world = synth.arena()
    path = synth.trajectory(world, 40, step=0.12, seed=9)  # some steps below the 0.2 m travel gate
    odom = synth.perturb(path, 0.04, math.radians(1.5), 10)
    rng = np.random.default_rng(3)
    n_proc = 0
    for t, o in zip(path, odom):
        r = synth.ranges_to_f64(synth.cast_scan(world, t, laser, 0.01, 0.01, rng))
        ok, pose = ref.process(r, o)
        ok2, pose2, _, _ = port.process(r, o)
        assert ok == ok2
        assert np.array_equal(pose, pose2)
        assert ref.running_scans() == port.running_scans()
        n_proc += ok
    assert 10 < n_proc < 40 and port.running_scans() <= 8


def test_occupancy_grid_counters(po, workload):
    """karto::OccupancyGrid::CreateFromScans (next-row #1)."""
    wl = workload
    for thr, res in ((20.0, 0.05), (49.5, 0.05), (12.0, 0.1)):
        ref, port = pair(po, range_threshold=thr)
        a, oa = ref.occgrid_from_scans(wl.base_ranges, wl.base_poses, res)
        b, ob = port.occgrid_from_scans(wl.base_ranges, wl.base_poses, res)
        assert a.shape == b.shape and np.array_equal(oa, ob)
        assert (a == 100).sum() > 100 and np.array_equal(a, b)
