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


# ---------------------------------------------------------------------------------------------------------
# lesson4 (Hector): oracle/hector_oracle.c against the reference's own hector_mapping headers, compiled
# unmodified as oracle/_ref/libhector_ref.so (oracle/hector_ref_driver.cpp; Eigen from oracle/shim/Eigen).
# Everything is compared BIT FOR BIT (float32 bytes).
# ---------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def hpo(oracle_lib):
    if not oracle_lib.have_ref_hector():
        pytest.skip("oracle/_ref/libhector_ref.so not built (needs /root/reference)")
    return oracle_lib


def _hector_scans(n_scans, seed, size=40.0, spread=3.0, use_max=20.0, cell=0.05):
    laser = synth.Laser()
    world = synth.arena(size=size, n_axis=10, n_rot=4, seed=5)
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_scans):
        pose = np.array([rng.uniform(-spread, spread), rng.uniform(-spread, spread), rng.uniform(-math.pi, math.pi)], np.float32)
        r = synth.cast_scan(world, tuple(float(v) for v in pose), laser, 0.01, 0.01, rng)
        out.append((synth.hector_points(r, laser, 1.0 / cell, use_max=use_max), pose, r))
    return out


@pytest.mark.parametrize("p_occ,n,cell", [(0.9, 1000, 0.05), (0.6, 640, 0.1)])
def test_hector_update_by_scan_vs_reference(hpo, p_occ, n, cell):
    """OccGridMapBase::updateByScan (H/map/OccGridMapBase.h:118-168) incl. beams leaving the map (640x0.1 m map is
    smaller than the 20 m the scans reach), the once-per-scan rule and the occupied clamp over repeated scans."""
    off = (n * cell * 0.5, n * cell * 0.5)
    ref, port = hpo.RefHector(n, n, cell, off), hpo.PortHector(n, n, cell, off)
    assert hpo._HrefLib.lib().href_sizeof_cell() == 8  # LogOddsCell {float, int} (SURVEY §8 sizes)
    for m in (ref, port):
        m.setUpdateFreeFactor(0.4)
        m.setUpdateOccupiedFactor(p_occ)
    assert ref.getScaleToMap() == port.getScaleToMap()
    scans = _hector_scans(12, seed=3, cell=cell)
    for k, (pts, pose, _) in enumerate(scans + scans[:3] * 4):  # the repeats drive cells towards the clamp
        origo = (0.0, 0.0) if k % 3 else (1.5, -0.5)            # a laser mounted off the base centre
        ref.updateByScan(pts, origo, pose)
        port.updateByScan(pts, origo, pose)
    a, b = ref.logodds(), port.logodds()
    assert np.count_nonzero(a) > 50000
    assert a.tobytes() == b.tobytes()
    assert np.array_equal(ref.update_index(), port.update_index())
    assert np.array_equal(ref.occupancy_i8(), port.occupancy_i8())


def test_hector_clamp_is_reached(hpo):
    """updateSetOccupied adds only below 50 (H/map/GridMapLogOdds.h:108-114): hammer one wall until it saturates."""
    n, cell = 400, 0.05
    off = (n * cell * 0.5, n * cell * 0.5)
    ref, port = hpo.RefHector(n, n, cell, off), hpo.PortHector(n, n, cell, off)
    for m in (ref, port):
        m.setUpdateOccupiedFactor(0.9)
    pts, pose, _ = _hector_scans(1, seed=9, size=16.0, spread=0.5, use_max=9.0)[0]
    for _ in range(40):
        ref.updateByScan(pts, (0.0, 0.0), pose)
        port.updateByScan(pts, (0.0, 0.0), pose)
    a = ref.logodds()
    assert a.max() > 50.0 and a.max() < 50.0 + 2.2  # crossed the clamp once, then stopped
    assert a.tobytes() == port.logodds().tobytes()


def test_hector_just_once_vs_reference(hpo):
    """updateByScanJustOnce (H/map/OccGridMapBase.h:175-217), the make_hector_map demo: 1600^2 map, points in metres,
    begin cell (800,800) and 1/0.05 hard-coded by the reference."""
    laser = synth.Laser()
    ref, port = hpo.RefHector(1600, 1600, 0.05, (40.0, 40.0)), hpo.PortHector(1600, 1600, 0.05, (40.0, 40.0))
    world = synth.arena(size=60.0, n_axis=10, n_rot=4, seed=8)
    for k in range(3):
        r = synth.cast_scan(world, (0.3 * k, -0.2 * k, 0.1 * k), laser)
        pm = synth.hector_points_metres(r, laser)
        ref.updateByScanJustOnce(pm)
        port.updateByScanJustOnce(pm)
    assert np.count_nonzero(ref.logodds()) > 100000
    assert ref.logodds().tobytes() == port.logodds().tobytes()


def test_hector_hessian_and_level_match_vs_reference(hpo):
    """getCompleteHessianDerivs / interpMapValueWithDerivatives (H/map/OccGridMapUtil.h:77-228) and
    ScanMatcher::matchData on one level (H/matcher/ScanMatcher.h:60-139), incl. starts far enough off that the
    0.2 rad clamp of the search direction fires and points fall outside the map."""
    n, cell = 512, 0.1
    off = (n * cell * 0.5, n * cell * 0.5)
    ref, port = hpo.RefHector(n, n, cell, off), hpo.PortHector(n, n, cell, off)
    for m in (ref, port):
        m.setUpdateOccupiedFactor(0.9)
    scans = _hector_scans(10, seed=11, spread=1.5, cell=cell)
    for pts, pose, _ in scans:
        ref.updateByScan(pts, (0.0, 0.0), pose)
        port.updateByScan(pts, (0.0, 0.0), pose)
    rng = np.random.default_rng(12)
    for pts, pose, _ in scans:
        for scale in (0.02, 0.3, 3.0):
            start = (pose + rng.uniform(-1, 1, 3) * np.array([scale, scale, 0.3 * scale])).astype(np.float32)
            pm = ref.world_to_map_pose(start)
            Hr, dr = ref.hessian_derivs(pts, pm)
            Hp, dp = port.hessian_derivs(pts, pm)
            assert Hr.tobytes() == Hp.tobytes() and dr.tobytes() == dp.tobytes()
            for iters in (0, 3, 5):
                pr, cr = ref.match_level(pts, start, iters)
                pp, cp = port.match_level(pts, start, iters)
                assert pr.tobytes() == pp.tobytes(), (start, iters, pr, pp)
                assert cr.tobytes() == cp.tobytes()
    # empty container: begin estimate comes straight back (ScanMatcher.h:96)
    pr, _ = ref.match_level(np.zeros((0, 2), np.float32), (1.0, 2.0, 0.3), 5)
    pp, _ = port.match_level(np.zeros((0, 2), np.float32), (1.0, 2.0, 0.3), 5)
    assert pr.tolist() == pp.tolist() == [1.0, 2.0, np.float32(0.3)]


def test_hector_pyramid_match_and_update_vs_reference(hpo):
    """MapRepMultiMap (H/slam_main/MapRepMultiMap.h): constructor geometry per level, matchData coarse-to-fine,
    updateByScan on every level from the containers cached by matchData -- and the reference's quirk that
    updateByScan WITHOUT a preceding matchData leaves the levels above 0 on stale (initially empty) containers."""
    laser = synth.Laser()
    n, cell, LV = 1024, 0.05, 3
    rep = hpo.RefHectorRep(cell, n, n, LV, (0.5, 0.5))
    rep.setUpdateFactorFree(0.4)
    rep.setUpdateFactorOccupied(0.9)
    ports = []
    for lv in range(LV):
        sx, sy, c, off = rep.level_info(lv)
        assert (sx, sy) == (n >> lv, n >> lv) and c == np.float32(cell) * np.float32(2 ** lv)
        assert off == (np.float32(cell) * n * np.float32(0.5),) * 2
        p = hpo.PortHector(sx, sy, c, off)
        p.setUpdateFreeFactor(0.4)
        p.setUpdateOccupiedFactor(0.9)
        ports.append(p)
    world = synth.arena(size=40.0, n_axis=10, n_rot=4, seed=5)
    scan0 = synth.hector_points(synth.cast_scan(world, (0.3, -0.2, 0.1), laser), laser, 1.0 / cell)
    # the quirk: no matchData yet -> dataContainers[] are empty -> only level 0 is updated
    rep.updateByScan(scan0, (0.0, 0.0), (0.3, -0.2, 0.1))
    assert rep.cached_points(1) == 0 and np.count_nonzero(rep.logodds(0)) > 0
    assert np.count_nonzero(rep.logodds(1)) == 0 and np.count_nonzero(rep.logodds(2)) == 0
    rep.reset()
    est = np.array([0.3, -0.2, 0.1], np.float32)
    for k in range(25):
        truth = (0.3 + 0.05 * k, -0.2 + 0.02 * k, 0.1 + 0.01 * k)
        pts = synth.hector_points(synth.cast_scan(world, truth, laser), laser, 1.0 / cell)
        guess = est if k else np.array(truth, np.float32)
        e_ref, H_ref = rep.matchData(pts, guess)
        e_port, H_port = hpo.PortHector.match_data(ports, pts, guess)
        assert e_ref.tobytes() == e_port.tobytes(), (k, e_ref, e_port)
        assert H_ref.tobytes() == H_port.tobytes()
        assert rep.cached_points(1) == len(pts) == rep.cached_points(2)
        est = e_ref
        rep.updateByScan(pts, (0.0, 0.0), est)
        rep.onMapUpdated()
        for lv, p in enumerate(ports):
            f = np.float32(hpo.PortHector.level_factor(lv)) if lv else np.float32(1)
            p.updateByScan(pts.astype(np.float32) * f, (0.0, 0.0), est)
    assert np.abs(est - np.array(truth)).max() < 0.05
    for lv in range(LV):
        assert rep.logodds(lv).tobytes() == ports[lv].logodds().tobytes()
        assert np.array_equal(rep.occupancy_i8(lv), ports[lv].occupancy_i8())


def test_hector_processor_loop_vs_restatement(hpo):
    """HectorSlamProcessor::update (H/slam_main/HectorSlamProcessor.h:84-110) over a trajectory: pose chain fed back as
    the next start estimate (hector_slam.cc:200-204), map updated only when poseDifferenceLargerThan says so.  The
    predicate is the reference's own (its unqualified abs(float) resolves to abs(int) with this toolchain:
    H/util/UtilFunctions.h:88, so sub-radian heading changes alone never trigger an update)."""
    laser = synth.Laser()
    n, cell, LV = 1024, 0.05, 3
    proc = hpo.RefHectorProcessor(cell, n, n, (0.5, 0.5), LV, p_free=0.4, p_occ=0.9)
    assert hpo.href_pose_difference_larger_than([0, 0, 0.5], [0, 0, 0], 10.0, 0.13) is False  # abs(int)
    assert hpo.href_pose_difference_larger_than([0, 0, 1.5], [0, 0, 0], 10.0, 0.13) is True
    ports = []
    for lv in range(LV):
        c = np.float32(cell) * np.float32(2 ** lv)
        p = hpo.PortHector(n >> lv, n >> lv, c, (np.float32(cell) * n * np.float32(0.5),) * 2)
        p.setUpdateFreeFactor(0.4)
        p.setUpdateOccupiedFactor(0.9)
        ports.append(p)
    world = synth.arena(size=40.0, n_axis=10, n_rot=4, seed=5)
    est = np.zeros(3, np.float32)
    last_update = np.full(3, np.finfo(np.float32).max, np.float32)
    updates = 0
    for k in range(60):
        truth = (0.04 * k, 0.015 * k, 0.004 * k)
        pts = synth.hector_points(synth.cast_scan(world, truth, laser), laser, 1.0 / cell)
        did = proc.update(pts, est)
        new_ref, cov_ref = proc.last_pose()
        new_port, cov_port = hpo.PortHector.match_data(ports, pts, est)
        assert new_ref.tobytes() == new_port.tobytes(), k
        assert cov_ref.tobytes() == cov_port.tobytes()
        if hpo.href_pose_difference_larger_than(new_port, last_update, 0.4, 0.13):
            for lv, p in enumerate(ports):
                f = np.float32(hpo.PortHector.level_factor(lv)) if lv else np.float32(1)
                p.updateByScan(pts.astype(np.float32) * f, (0.0, 0.0), new_port)
            last_update = new_port.copy()
            updates += 1
            assert did
        else:
            assert not did
        est = new_ref
    assert 4 <= updates <= 10
    assert np.abs(est - np.array(truth)).max() < 0.05
    for lv in range(LV):
        assert proc.logodds(lv).tobytes() == ports[lv].logodds().tobytes()


def test_hector_non_finite_points_vs_reference(hpo):
    """NaN / Inf / out-of-int-range points: the x86 float->int cast gives INT_MIN, the in-map test drops the beam
    (H/map/OccGridMapBase.h:226-238) -- the behaviour the HIP kernel reproduces explicitly."""
    n, cell = 200, 0.05
    ref, port = hpo.RefHector(n, n, cell, (5.0, 5.0)), hpo.PortHector(n, n, cell, (5.0, 5.0))
    nan, inf = np.float32("nan"), np.float32("inf")
    pts = np.array([[40, 0], [nan, 3], [3, nan], [inf, 0], [0, -inf], [3e9, 0], [-3e9, 1], [1e20, 1e20], [0, 30],
                    [nan, nan]], dtype=np.float32)
    for m in (ref, port):
        m.updateByScan(pts, (0.0, 0.0), (0.0, 0.0, 0.2))
    a = ref.logodds()
    assert (a > 0).sum() == 2 and a[0, 0] == 0
    assert a.tobytes() == port.logodds().tobytes()


def test_hector_rep_restatement_vs_reference(hpo):
    """PortHectorRep (the restated MapRepMultiMap incl. its cached containers) against the reference's, with a stale
    container on purpose."""
    laser = synth.Laser()
    n, cell, LV = 512, 0.05, 3
    ref, port = hpo.RefHectorRep(cell, n, n, LV), hpo.PortHectorRep(cell, n, n, LV)
    for m in (ref, port):
        m.setUpdateFactorFree(0.4)
        m.setUpdateFactorOccupied(0.9)
    world = synth.arena(size=20.0, n_axis=4, n_rot=2, seed=2)
    for k in range(6):
        truth = (3.0 + 0.05 * k, 3.0, 0.02 * k)
        pts = synth.hector_points(synth.cast_scan(world, truth, laser), laser, 1.0 / cell, use_max=12.0)
        if k != 3:  # scan 3 is integrated without matching it first -> levels > 0 reuse scan 2's container
            a, b = ref.matchData(pts, truth, (0.5, 0.25)), port.matchData(pts, truth, (0.5, 0.25))
            assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes()
        ref.updateByScan(pts, (0.5, 0.25), truth)
        ref.onMapUpdated()
        port.updateByScan(pts, (0.5, 0.25), truth)
    for lv in range(LV):
        assert np.count_nonzero(ref.logodds(lv)) > 100
        assert ref.logodds(lv).tobytes() == port.logodds(lv).tobytes()
