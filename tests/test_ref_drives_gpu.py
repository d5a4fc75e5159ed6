"""the reference's OWN orchestrators driving the HIP path through the C ABI.

* Karto: oracle/_ref_gpu/libkarto_ref_gpu.so is the reference's Karto.o + Mapper.o with ONE symbol substituted at link
  time -- karto::ScanMatcher::MatchScan is integration/karto_scan_matcher_gpu.cpp, which takes the reference's
  `LocalizedRangeScan*` / `LocalizedRangeScanVector`, builds its `lslam_laser` from the scan's `LaserRangeFinder` and
  calls lslam_matcher_match_scan.  karto::Mapper::Process, MapperGraph::AddEdges / LinkNearChains / TryCloseLoop,
  ScanManager, the dataset: all the reference's compiled code.  Expected: the same poses and the same graph as the pure
  reference (oracle/_ref/libkarto_ref.so) on closed-loop trajectories.
* Hector: oracle/_ref_gpu/libhector_ref_gpu.so is the reference's HectorSlamProcessor with `mapRep` swapped for
  integration/hector_map_rep_gpu.hpp (`HectorMapRepGpu : MapRepresentationInterface`).  Expected: the same map-update
  decisions, poses within the fp32 tolerance of the device Gauss-Newton matcher, maps equal where poses are.

Both twins are built in the container that holds /root/reference (`make -C oracle ref_gpu`) and travel to the GPU box.
"""
import math

import numpy as np
import pytest

from lslam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def po(oracle_lib):
    if not (oracle_lib.have_ref() and oracle_lib.have_ref_gpu()):
        pytest.skip("oracle/_ref or oracle/_ref_gpu not built (needs /root/reference at build time)")
    return oracle_lib


def _run_pair(po, kw, laser, thr, offset, path, odom, world, seed):
    cfg = po.default_cfg(**kw)
    cpu = po.RefKarto(cfg, po.laser_struct(laser, thr, offset))
    gpu = po.RefKarto(cfg, po.laser_struct(laser, thr, offset), gpu=True)
    calls0 = gpu.gpu_match_calls()
    worst = 0.0
    for i, (t, o) in enumerate(zip(path, odom)):
        c, s_ = math.cos(t[2]), math.sin(t[2])
        lp = (t[0] + c * offset[0] - s_ * offset[1], t[1] + s_ * offset[0] + c * offset[1], t[2] + offset[2])
        r = synth.ranges_to_f64(synth.cast_scan(world, lp, laser, 0.01, 0.01, np.random.default_rng([seed, i])))
        ok_c, pose_c = cpu.process(r, o)
        ok_g, pose_g = gpu.process(r, o)
        assert ok_c == ok_g, i
        worst = max(worst, float(np.abs(pose_c - pose_g).max()))
        assert worst <= 1e-9, (i, pose_c, pose_g)
        assert cpu.graph_stats() == gpu.graph_stats(), i
    n = cpu.graph_stats()[0]
    final_c = np.stack([cpu.scan_pose(i) for i in range(n)])
    final_g = np.stack([gpu.scan_pose(i) for i in range(n)])
    assert np.abs(final_c - final_g).max() <= 1e-9
    return cpu, gpu, worst, gpu.gpu_match_calls() - calls0


def test_reference_mapper_on_gpu_matcher_closed_loop(po):
    """karto::Mapper (the reference's, unmodified) + GPU MatchScan == karto::Mapper + its own MatchScan on a trajectory
    that closes loops: sequential matches, near-chain matches and the 81x81x21 loop matcher all go to the device."""
    laser = synth.Laser()
    world = synth.arena(size=40.0, n_axis=10, n_rot=4, seed=5)
    path = synth.loop_trajectory(260, w=8.0, h=5.0, step=0.2, origin=(-4.0, -2.5))
    odom = synth.drifting_odometry(path, scale=1.03, seed=9)
    kw = dict(scan_buffer_size=25, scan_buffer_max_scan_distance=6.0, do_loop_closing=1, link_scan_maximum_distance=1.5,
              loop_search_maximum_distance=3.0, loop_match_minimum_chain_size=8)
    cpu, gpu, worst, calls = _run_pair(po, kw, laser, 20.0, (0.0, 0.0, 0.0), path, odom, world, 43)
    v, e = cpu.graph_stats()
    assert e > v > 100  # loop / near-chain links exist on both sides (equal counts asserted per scan)
    # every processed scan but the first is matched at least once on the device (Mapper.cpp:2040)
    assert calls >= v - 1, (calls, v)
    print("reference Mapper + GPU MatchScan vs pure reference: max pose difference", worst, "device MatchScan calls", calls,
          "vertices/edges", (v, e))
    # seam B2 with the reference's own types: the published map of ALL processed scans (karto_slam.cc:511-512), built from the
    # Mapper's LocalizedRangeScanVector by integration/karto_occupancy_grid_gpu.hpp and returned as a karto::OccupancyGrid
    g_c, off_c = cpu.occupancy_grid(0.05)
    g_g, off_g = gpu.occupancy_grid(0.05)
    assert g_c.shape == g_g.shape and np.array_equal(off_c, off_g)
    assert np.array_equal(g_c, g_g)
    assert (g_c == 100).sum() > 500 and (g_c == 255).sum() > 20000


def test_reference_mapper_uses_the_scan_cache_and_survives_reset(po):
    """Seam B1's device-side scan cache (round 4): the reference's Mapper::Process on the GPU MatchScan sends every scan's
    readings ONCE (not its whole running window per call), the world points of the scan just matched are prepared behind
    its own match, ~ScanMatcher releases the device matcher (Mapper::Reset, Mapper.cpp:1980-1992, deletes and re-creates
    its matchers: no leak per cycle) and forgets the scans the old Mapper numbered; the host grid behind
    GetCorrelationGrid() can be refreshed on request and equals the pure reference's."""
    laser = synth.Laser()
    world = synth.arena(size=40.0, n_axis=10, n_rot=4, seed=5)
    path = synth.loop_trajectory(90, w=6.0, h=4.0, step=0.2, origin=(-3.0, -2.0))
    odom = synth.drifting_odometry(path, scale=1.02, seed=13)
    kw = dict(scan_buffer_size=20, scan_buffer_max_scan_distance=5.0, do_loop_closing=1, link_scan_maximum_distance=1.5,
              loop_search_maximum_distance=3.0, loop_match_minimum_chain_size=6)
    cfg = po.default_cfg(**kw)
    cpu = po.RefKarto(cfg, po.laser_struct(laser, 20.0))
    gpu = po.RefKarto(cfg, po.laser_struct(laser, 20.0), gpu=True)
    scans = [synth.ranges_to_f64(synth.cast_scan(world, t, laser, 0.01, 0.01, np.random.default_rng([51, i]))) for i, t in enumerate(path)]
    alive0 = gpu.gpu_stats()["matchers_alive"]
    for cycle in range(3):
        s0 = gpu.gpu_stats()
        for i, (r, o) in enumerate(zip(scans, odom)):
            ok_c, pose_c = cpu.process(r, o)
            ok_g, pose_g = gpu.process(r, o)
            assert ok_c == ok_g and np.abs(pose_c - pose_g).max() <= 1e-9, (cycle, i)
            assert cpu.graph_stats() == gpu.graph_stats(), (cycle, i)
        s1 = gpu.gpu_stats()
        n = cpu.graph_stats()[0]
        calls = s1["match_calls"] - s0["match_calls"]
        assert calls >= n - 1 and s1["cached_calls"] - s0["cached_calls"] == calls  # every call of the Mapper went through the cache
        # one upload per processed scan (the first scan of a Mapper is never a query: it goes up when first named as a base scan)
        assert s1["scans_uploaded"] - s0["scans_uploaded"] == n, (s0, s1, n)
        assert s1["resident_scans"] == n
        # the scan a Process() call matches is prepared at its new pose behind that match: refreshes in front of a match are
        # the exception (the first scan, scans a closed loop re-posed), not one per call
        assert s1["refreshes"] - s0["refreshes"] <= 0.2 * calls + 2, (s0, s1)
        assert s1["matchers_alive"] <= alive0 + 2  # sequential + loop matcher of THIS Mapper, whatever the cycle
        g_c, off_c = cpu.mapper_grid()
        g_g, off_g = gpu.mapper_grid()
        assert g_c is not None and g_g is not None and g_c.any()
        assert np.array_equal(off_c, off_g) and np.array_equal(g_c, g_g)
        cpu.reset()
        gpu.reset()
        s2 = gpu.gpu_stats()
        assert s2["matchers_alive"] == alive0 and s2["resident_scans"] == 0, s2
    cpu.close()
    gpu.close()


def test_reference_mapper_literal_forwarding_switch(po):
    """LSLAM_KARTO_NO_CACHE=1 (read once per process) selects round 3's literal forwarding; run in a child process and
    compare its poses with the cached path of this process: same records either way."""
    import subprocess
    import sys
    import textwrap

    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r)
        import lslam
        from lslam_amd import synth
        from oracle import pyoracle as po
        laser = synth.Laser()
        world = synth.arena(size=40.0, n_axis=10, n_rot=4, seed=5)
        path = synth.loop_trajectory(40, w=6.0, h=4.0, step=0.2, origin=(-3.0, -2.0))
        odom = synth.drifting_odometry(path, scale=1.02, seed=13)
        gpu = po.RefKarto(po.default_cfg(scan_buffer_size=20, scan_buffer_max_scan_distance=5.0), po.laser_struct(laser, 20.0), gpu=True)
        out = []
        for i, (t, o) in enumerate(zip(path, odom)):
            r = synth.ranges_to_f64(synth.cast_scan(world, t, laser, 0.01, 0.01, np.random.default_rng([52, i])))
            out.append(gpu.process(r, o)[1])
        s = gpu.gpu_stats()
        print("RESULT", s["cached_calls"], s["match_calls"], np.stack(out).tobytes().hex())
    """) % str(__import__("pathlib").Path(__file__).resolve().parent.parent)
    res = {}
    for flag in ("0", "1"):
        env = dict(__import__("os").environ, LSLAM_KARTO_NO_CACHE=flag)
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")][-1].split()
        res[flag] = (int(line[1]), int(line[2]), line[3])
    assert res["0"][0] == res["0"][1] > 30 and res["1"][0] == 0 and res["1"][1] == res["0"][1]
    assert res["0"][2] == res["1"][2]


@pytest.mark.parametrize("variant", ["laser_offset", "response_expansion"])
def test_reference_mapper_on_gpu_matcher_variants(po, variant):
    """The lslam_laser built from the reference's LaserRangeFinder carries the mounting offset (Sensor::SetOffsetPose,
    karto_slam.cc:387-389); response expansion is read from the Mapper parameter like Mapper.cpp:238."""
    laser = synth.Laser()
    world = synth.arena(size=40.0, n_axis=10, n_rot=4, seed=5)
    path = synth.loop_trajectory(120, w=6.0, h=4.0, step=0.2, origin=(-3.0, -2.0))
    odom = synth.drifting_odometry(path, scale=1.02, seed=11)
    offset = (0.18, -0.05, 0.04) if variant == "laser_offset" else (0.0, 0.0, 0.0)
    kw = dict(scan_buffer_size=20, scan_buffer_max_scan_distance=5.0, do_loop_closing=1, link_scan_maximum_distance=1.5,
              loop_search_maximum_distance=3.0, loop_match_minimum_chain_size=6,
              use_response_expansion=1 if variant == "response_expansion" else 0)
    _run_pair(po, kw, laser, 20.0, offset, path, odom, world, 47)


def test_reference_match_scan_entry_on_gpu(po, workload):
    """One ScanMatcher::MatchScan call through the reference's signature (LocalizedRangeScan*, vector, Pose2&, Matrix3&):
    response identical, pose / covariance <= 1e-12, with and without penalties / refinement."""
    wl = workload
    cpu = po.RefKarto(po.default_cfg(), po.laser_struct(wl.laser))
    gpu = po.RefKarto(po.default_cfg(), po.laser_struct(wl.laser), gpu=True)
    for q in range(4):
        for pen, fine in ((True, True), (False, False), (False, True)):
            a = cpu.match(wl.base_ranges, wl.base_poses, wl.query_ranges[q], wl.query_poses[q], pen, fine)
            b = gpu.match(wl.base_ranges, wl.base_poses, wl.query_ranges[q], wl.query_poses[q], pen, fine)
            assert a[2] == b[2]
            assert np.abs(a[0] - b[0]).max() <= 1e-12
            assert np.abs(a[1] - b[1]).max() <= 1e-12 * max(1.0, np.abs(a[1]).max())


def test_reference_hector_processor_on_gpu_map_rep(po):
    """HectorSlamProcessor::update (H/slam_main/HectorSlamProcessor.h:84-110), the reference's own, with mapRep =
    HectorMapRepGpu: 60 scans, pose chain fed back as the next start estimate (hector_slam.cc:200-204).  The device
    Gauss-Newton matcher is fp32 with device libm (tolerance 1e-4); the map-update decisions are the
    reference's own predicate on those poses and must agree; the maps are compared cell for cell."""
    laser = synth.Laser()
    n, cell, LV = 1024, 0.05, 3
    cpu = po.RefHectorProcessor(cell, n, n, (0.5, 0.5), LV, p_free=0.4, p_occ=0.9)
    gpu = po.RefHectorProcessor(cell, n, n, (0.5, 0.5), LV, p_free=0.4, p_occ=0.9, gpu=True)
    world = synth.arena(size=40.0, n_axis=10, n_rot=4, seed=5)
    est_c = np.zeros(3, np.float32)
    est_g = np.zeros(3, np.float32)
    updates, worst = 0, 0.0
    for k in range(60):
        truth = (0.04 * k, 0.015 * k, 0.004 * k)
        pts = synth.hector_points(synth.cast_scan(world, truth, laser), laser, 1.0 / cell)
        did_c = cpu.update(pts, est_c)
        did_g = gpu.update(pts, est_g)
        assert did_c == did_g, k
        updates += did_c
        est_c, cov_c = cpu.last_pose()
        est_g, cov_g = gpu.last_pose()
        worst = max(worst, float(np.abs(est_c - est_g).max()))
        assert worst <= 1e-4, (k, est_c, est_g)
        assert np.abs(cov_c - cov_g).max() <= 2e-3 * max(1.0, float(np.abs(cov_c).max())), k
    assert 4 <= updates <= 10
    assert np.abs(est_g - np.array(truth)).max() < 0.05
    for lv in range(LV):
        a, b = cpu.logodds(lv), gpu.logodds(lv)
        differing = int(np.count_nonzero(a != b))
        # poses differ by <= 1e-4 (fp32 matcher), so a handful of ray end cells may land one cell over
        assert differing <= 0.002 * np.count_nonzero(a), (lv, differing)
    print("reference HectorSlamProcessor + GPU map rep: max pose difference", worst, "map updates", updates)


def test_reference_hector_processor_on_gpu_map_rep_without_matching(po):
    """map_without_matching = true (the mapping-only mode of HectorSlamProcessor::update): no matcher in the loop, so the
    pyramid the reference's processor builds through the GPU map rep is BIT-equal to its own -- level 0 only, because
    levels > 0 are fed from containers cached by matchData (MapRepMultiMap.h:161,186), which never runs here: they stay
    empty on both sides."""
    laser = synth.Laser()
    n, cell, LV = 512, 0.05, 3
    cpu = po.RefHectorProcessor(cell, n, n, (0.5, 0.5), LV, p_free=0.4, p_occ=0.9)
    gpu = po.RefHectorProcessor(cell, n, n, (0.5, 0.5), LV, p_free=0.4, p_occ=0.9, gpu=True)
    world = synth.arena(size=30.0, n_axis=8, n_rot=3, seed=6)
    for k in range(25):
        pose = np.array([0.1 * k - 1.0, 0.05 * k, 0.03 * k], np.float32)
        pts = synth.hector_points(synth.cast_scan(world, pose, laser), laser, 1.0 / cell, use_max=10.0)
        assert cpu.update(pts, pose, map_without_matching=True)
        assert gpu.update(pts, pose, map_without_matching=True)
    for lv in range(LV):
        assert cpu.logodds(lv).tobytes() == gpu.logodds(lv).tobytes(), lv
    assert np.count_nonzero(cpu.logodds(0)) > 1000


def test_reference_occupancy_grid_through_karto_types(po, workload):
    """lslam::CreateOccupancyGridFromScans(LocalizedRangeScanVector, resolution) -> karto::OccupancyGrid*: cells, size and
    CoordinateConverter offset equal to karto::OccupancyGrid::CreateFromScans on the same scans, incl. a laser mounted off
    the base centre and ignored readings; an empty scan list is NULL on both sides."""
    wl = workload
    ranges = wl.base_ranges.copy()
    ranges[0, 5] = np.nan
    ranges[1, 7] = 0.05
    ranges[2, 9] = 70.0
    for offset, res in (((0.0, 0.0, 0.0), 0.05), ((0.2, -0.1, 0.05), 0.1)):
        cpu = po.RefKarto(po.default_cfg(), po.laser_struct(wl.laser, 20.0, offset))
        gpu = po.RefKarto(po.default_cfg(), po.laser_struct(wl.laser, 20.0, offset), gpu=True)
        a, off_a = cpu.occgrid_from_scans(ranges, wl.base_poses, res)
        b, off_b = gpu.occgrid_from_scans(ranges, wl.base_poses, res)
        assert a.shape == b.shape and np.array_equal(off_a, off_b)
        assert np.array_equal(a, b)
        assert cpu.occgrid_from_scans(ranges[:0], wl.base_poses[:0], res) == (None, None)
        assert gpu.occgrid_from_scans(ranges[:0], wl.base_poses[:0], res) == (None, None)
