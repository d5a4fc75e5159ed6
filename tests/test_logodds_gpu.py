"""GPU parity tests of the Hector log-odds Bresenham map update against the CPU restatement
(oracle/hector_oracle.c).  The map is float32: same operations in the same per-cell order, so the
log-odds planes must be BIT-EXACT, and the published int8 occupancy identical."""
import math

import numpy as np
import pytest

from lslam_amd import api, synth

pytestmark = pytest.mark.gpu


def scans_for_map(n_scans=6, seed=3, map_cells=1000, cell=0.05):
    world = synth.arena(size=40.0, n_axis=10, n_rot=4, seed=seed)
    laser = synth.Laser()
    path = synth.trajectory(world, n_scans, step=0.45, seed=seed, bounds=8.0)
    rng = np.random.default_rng(seed)
    out = []
    for p in path:
        r = synth.cast_scan(world, p, laser, 0.01, 0.01, rng)
        out.append((synth.hector_points(r, laser, 1.0 / cell), p.astype(np.float32)))
    return out


def test_update_by_scan_bit_exact(ctx, oracle_lib):
    n, cell = 1000, 0.05
    off = (n * cell * 0.5, n * cell * 0.5)  # MapRepMultiMap mid offset (MapRepMultiMap.h:62-66)
    cpu = oracle_lib.PortHector(n, n, cell, off)
    gpu = api.OccGridMap(ctx, n, n, cell, off, levels=1)
    for m in (cpu, gpu):
        m.setUpdateFreeFactor(0.4)
        m.setUpdateOccupiedFactor(0.9)  # hector_slam.cc:144-145
    assert gpu.getScaleToMap() == cpu.getScaleToMap()
    for pts, pose in scans_for_map():
        assert len(pts) > 500
        cpu.updateByScan(pts, (0.0, 0.0), pose)
        gpu.updateByScan(pts, (0.0, 0.0), pose)
        a, b = cpu.logodds(), gpu.logodds()
        assert (a != 0).sum() > 10000
        assert a.tobytes() == b.tobytes()
    assert np.array_equal(cpu.occupancy_i8(), gpu.occupancy_i8())
    occ = gpu.occupancy_i8()
    assert {int(v) for v in np.unique(occ)} == {-1, 0, 100}


def test_repeated_scans_hit_the_occupied_clamp(ctx, oracle_lib):
    """logOdds < 50 clamp applies to occupied updates only (GridMapLogOdds.h:108-114)."""
    n, cell = 400, 0.05
    off = (10.0, 10.0)
    cpu = oracle_lib.PortHector(n, n, cell, off)
    gpu = api.OccGridMap(ctx, n, n, cell, off)
    for m in (cpu, gpu):
        m.setUpdateOccupiedFactor(0.999)  # ln(999) ~ 6.9 per hit -> clamp after 8 scans
    (pts, pose) = scans_for_map(1, seed=7, map_cells=n)[0]
    pose = np.array([0.0, 0.0, 0.3], dtype=np.float32)
    pts = pts[(np.abs(pts) < 150).all(axis=1)]
    for _ in range(12):
        cpu.updateByScan(pts, (0.0, 0.0), pose)
        gpu.updateByScan(pts, (0.0, 0.0), pose)
    a, b = cpu.logodds(), gpu.logodds()
    assert a.max() >= 50.0 and a.max() < 50.0 + 7.0
    assert a.tobytes() == b.tobytes()


def test_free_then_occupied_same_scan(ctx, oracle_lib):
    """A cell crossed by an earlier beam and hit by a later one gets (v+free)-free then +occ;
    hit first and crossed later gets +occ only (OccGridMapBase.h:316-330)."""
    n, cell = 200, 0.05
    cpu = oracle_lib.PortHector(n, n, cell, (5.0, 5.0))
    gpu = api.OccGridMap(ctx, n, n, cell, (5.0, 5.0))
    # beams along +x of different lengths, both orders, plus duplicates and a zero-length beam
    pts = np.array([[40, 0], [20, 0], [10, 0], [30, 0], [30, 0], [0.2, 0.1], [-15, 25], [-15, 25], [-7.5, 12.5],
                    [500, 0], [0, -60]], dtype=np.float32)
    pose = np.array([0.0, 0.0, 0.0], dtype=np.float32)
    for k in range(3):
        cpu.updateByScan(pts, (0.0, 0.0), pose)
        gpu.updateByScan(pts, (0.0, 0.0), pose)
        assert cpu.logodds().tobytes() == gpu.logodds().tobytes()
        pts = pts[::-1].copy()  # reverse beam order next round


def test_pyramid_levels(ctx, oracle_lib):
    """MapRepMultiMap: level i = size>>i, cell*2^i, points*(1/2^i) from the container CACHED BY matchData
    (MapRepMultiMap.h:57-93,144-191) -- compared with the reference's own MapRepMultiMap where its build is present
    and with the restatement (pinned to it in tests/test_oracle_vs_ref.py) always."""
    n, cell, levels = 512, 0.05, 3
    gpu = api.OccGridMap(ctx, n, n, cell, (n * cell * 0.5, n * cell * 0.5), levels=levels)
    reps = [oracle_lib.PortHectorRep(cell, n, n, levels)]
    if oracle_lib.have_ref_hector():
        reps.append(oracle_lib.RefHectorRep(cell, n, n, levels))
    assert gpu.levels == levels
    scans = scans_for_map(4, seed=9, map_cells=n)
    # the reference's quirk first: updateByScan before any matchData leaves the levels above 0 untouched
    pts0, pose0 = scans[0]
    pts0 = pts0[(np.abs(pts0) < 200).all(axis=1)]
    pose0 = (pose0 * np.float32(0.2)).astype(np.float32)
    for m in reps + [gpu]:
        m.updateByScan(pts0, (1.5, -0.5), pose0)
    assert gpu.cached_points() == 0
    assert np.count_nonzero(gpu.logodds(0)) > 0 and np.count_nonzero(gpu.logodds(1)) == 0
    for k, (pts, pose) in enumerate(scans[1:]):
        pts = pts[(np.abs(pts) < 200).all(axis=1)]
        pose = (pose * np.float32(0.2)).astype(np.float32)
        for r in reps:
            r.matchData(pts, pose, (1.5, -0.5))
        gpu.matchData(pose, pts, (1.5, -0.5))
        assert gpu.cached_points() == len(pts)
        # stale container on purpose in the last round: level 0 gets HALF the scan, the upper levels the cached one
        upd = pts if k < 2 else pts[: len(pts) // 2]
        for m in reps + [gpu]:
            m.updateByScan(upd, (1.5, -0.5), pose)
    for i in range(levels):
        assert gpu.size(i) == (n >> i, n >> i)
        for r in reps:
            assert r.logodds(i).tobytes() == gpu.logodds(i).tobytes(), i


def test_non_finite_and_huge_points_are_dropped(ctx, oracle_lib):
    """The reference only ever sees finite points, but its x86 float->int cast turns NaN / Inf / out-of-range into
    INT_MIN, which the in-map test of updateLineBresenhami rejects (OccGridMapBase.h:226-238): the beam is skipped."""
    n, cell = 200, 0.05
    cpu = oracle_lib.PortHector(n, n, cell, (5.0, 5.0))
    gpu = api.OccGridMap(ctx, n, n, cell, (5.0, 5.0))
    nan, inf = np.float32("nan"), np.float32("inf")
    pts = np.array([[40, 0], [nan, 3], [3, nan], [inf, 0], [0, -inf], [3e9, 0], [-3e9, 1], [1e20, 1e20], [0, 30],
                    [nan, nan]], dtype=np.float32)
    pose = np.array([0.0, 0.0, 0.2], dtype=np.float32)
    cpu.updateByScan(pts, (0.0, 0.0), pose)
    gpu.updateByScan(pts, (0.0, 0.0), pose)
    a = cpu.logodds()
    assert (a > 0).sum() == 2  # only the two finite in-map beams end in a cell
    assert a.tobytes() == gpu.logodds().tobytes()
    assert gpu.logodds()[100, 100] < 0 and gpu.logodds()[0, 0] == 0  # nothing was traced to cell (0,0)


def test_more_than_4096_points(ctx, oracle_lib):
    """The per-cell key carries the beam index in 16 bits: containers of up to 65536 points."""
    n, cell = 600, 0.05
    cpu = oracle_lib.PortHector(n, n, cell, (15.0, 15.0))
    gpu = api.OccGridMap(ctx, n, n, cell, (15.0, 15.0))
    rng = np.random.default_rng(5)
    ang = rng.uniform(-math.pi, math.pi, 9000)
    rad = rng.uniform(20, 280, 9000)
    pts = np.stack([rad * np.cos(ang), rad * np.sin(ang)], axis=1).astype(np.float32)
    pose = np.array([0.3, -0.4, 0.7], dtype=np.float32)
    for _ in range(2):
        cpu.updateByScan(pts, (0.0, 0.0), pose)
        gpu.updateByScan(pts, (0.0, 0.0), pose)
    assert cpu.logodds().tobytes() == gpu.logodds().tobytes()
    with pytest.raises(Exception):
        gpu.updateByScan(np.zeros((70000, 2), np.float32), (0.0, 0.0), pose)


def test_update_just_once_demo_variant(ctx, oracle_lib):
    """lesson4 make_hector_map: fresh 1600x1600 map per scan, begin (800,800), metres/0.05."""
    laser = synth.Laser()
    world = synth.arena(size=40.0, n_axis=10, n_rot=4, seed=2)
    r = synth.cast_scan(world, (1.0, -2.0, 0.4), laser)
    pts = synth.hector_points_metres(r, laser)
    cpu = oracle_lib.PortHector(1600, 1600, 0.05, (0.0, 0.0))
    gpu = api.OccGridMap(ctx, 1600, 1600, 0.05, (0.0, 0.0))
    cpu.updateByScanJustOnce(pts)
    gpu.updateByScanJustOnce(pts)
    a = cpu.logodds()
    assert (a > 0).sum() > 300 and (a < 0).sum() > 10000
    assert a.tobytes() == gpu.logodds().tobytes()


def test_out_of_map_beams_are_dropped(ctx, oracle_lib):
    n, cell = 100, 0.1
    cpu = oracle_lib.PortHector(n, n, cell, (5.0, 5.0))
    gpu = api.OccGridMap(ctx, n, n, cell, (5.0, 5.0))
    pts = np.array([[200, 0], [-200, 3], [10, 10], [0, 49.4], [0, 49.6], [49.4, 0]], dtype=np.float32)
    for pose in ((0, 0, 0), (4.9, 0, 0), (20.0, 0, 0)):  # last: begin cell outside the map
        pose = np.array(pose, dtype=np.float32)
        cpu.updateByScan(pts, (0, 0), pose)
        gpu.updateByScan(pts, (0, 0), pose)
    assert cpu.logodds().tobytes() == gpu.logodds().tobytes()
    # empty scan is legal
    gpu.updateByScan(np.zeros((0, 2), np.float32), (0, 0), np.zeros(3, np.float32))


@pytest.mark.parametrize("ordered", [False, True])
def test_gauss_newton_match_data(ctx, oracle_lib, ordered):
    """Next-row #2: MapRepMultiMap::matchData (coarse-to-fine Gauss-Newton on the 3-level pyramid) --
    the lesson4 front-end loop matchData -> updateByScan, GPU vs the restated oracle (pinned bit for bit to the
    reference's headers).  Default kernel: parallel fp32 tree sums + float32 libm; `ordered_sums`: the reference's
    sequential accumulation order.  Tolerance for both: the north star's 1e-4 (map units are metres / radians)."""
    laser = synth.Laser()
    n, cell, levels = 1024, 0.05, 3
    off = (n * cell * 0.5, n * cell * 0.5)
    world = synth.arena(size=40.0, n_axis=10, n_rot=4, seed=3)
    cpus = [oracle_lib.PortHector(n >> i, n >> i, cell * 2 ** i, off) for i in range(levels)]
    gpu = api.OccGridMap(ctx, n, n, cell, off, levels=levels)
    gpu.set_option("ordered_sums", int(ordered))
    for m in cpus + [gpu]:
        m.setUpdateOccupiedFactor(0.9)
    path = synth.trajectory(world, 14, step=0.3, seed=3, bounds=6.0)
    rng = np.random.default_rng(1)
    worst = 0.0
    for k, t in enumerate(path):
        r = synth.cast_scan(world, t, laser, 0.01, 0.0, rng)
        pts = synth.hector_points(r, laser, 1.0 / cell, use_max=20.0)
        if k == 0:
            pose_c = pose_g = t.astype(np.float32)
            gpu.matchData(pose_c, pts)  # like HectorSlamProcessor::update: match first (caches the containers)
        else:
            hint = (t + np.array([0.08, -0.06, 0.03])).astype(np.float32)
            pose_c, H_c = oracle_lib.PortHector.match_data(cpus, pts, hint)
            pose_g, H_g = gpu.matchData(hint, pts)  # also caches pts for the update below
            d = float(np.abs(pose_c - pose_g).max())
            worst = max(worst, d)
            assert d <= 1e-4, (k, pose_c, pose_g)
            # the last Hessian (what matchData returns as covMatrix) is summed over gradients that jump at cell borders: a
            # 1e-5 pose difference moves it by a few 1e-3 relative
            assert np.abs(H_c - H_g).max() <= (1e-3 if ordered else 1e-2) * max(1.0, float(np.abs(H_c).max()))
            assert np.hypot(*(pose_g[:2] - t[:2])) < 0.03  # it converges to the truth from an 10 cm / 1.7 deg hint
        for i, c in enumerate(cpus):  # keep both maps identical: update both with the ORACLE pose
            f = np.float32(oracle_lib.PortHector.level_factor(i))
            c.updateByScan(pts if i == 0 else pts * f, (0.0, 0.0), pose_c)
        gpu.updateByScan(pts, (0.0, 0.0), pose_c)
    for i, c in enumerate(cpus):
        assert c.logodds().tobytes() == gpu.logodds(i).tobytes()
    print("ordered" if ordered else "parallel", "sums: max |pose_gpu - pose_oracle| =", worst)
    assert worst <= (5e-6 if ordered else 5e-5)  # observed: ~2e-6 / ~2e-5 (what the 1e-4 tolerance is spent on)
    # empty scan: beginEstimateWorld comes back unchanged
    p, _ = gpu.matchData(np.array([1.0, 2.0, 0.3], np.float32), np.zeros((0, 2), np.float32))
    assert np.array_equal(p, np.array([1.0, 2.0, 0.3], np.float32))


@pytest.mark.parametrize("n_points,threads", [(900, "256"), (900, "1024"), (2000, "512"), (9000, "512")])
def test_gauss_newton_kernel_variants_agree(ctx, monkeypatch, n_points, threads):
    """The parallel-sum matcher has three forms -- points in registers (<= 3 per thread at 512 threads: every real scan),
    points in LDS (longer containers), points in memory (beyond 56 KB of LDS) -- and three block sizes (LSLAM_GN_THREADS);
    all must land within the matcher's tolerance of the ORDERED kernel (which equals the CPU restatement bit for bit,
    test_gauss_newton_match_data[True]) on the same map and container, and the cached container must feed the update of
    the levels above 0 identically."""
    laser = synth.Laser()
    n, cell, levels = 1024, 0.05, 3
    off = (n * cell * 0.5, n * cell * 0.5)
    world = synth.arena(size=40.0, n_axis=10, n_rot=4, seed=3)
    rng = np.random.default_rng(7)
    base = synth.hector_points(synth.cast_scan(world, (0.5, 0.2, 0.3), laser, 0.01, 0.0, rng), laser, 1.0 / cell, use_max=20.0)
    reps = -(-n_points // len(base))
    pts = np.concatenate([base + rng.normal(0.0, 0.05, base.shape).astype(np.float32) for _ in range(reps)])[:n_points].astype(np.float32)
    monkeypatch.setenv("LSLAM_GN_THREADS", threads)
    fast = api.OccGridMap(ctx, n, n, cell, off, levels=levels)
    monkeypatch.delenv("LSLAM_GN_THREADS")
    slow = api.OccGridMap(ctx, n, n, cell, off, levels=levels)
    slow.set_option("ordered_sums", 1)
    pose0 = np.array([0.5, 0.2, 0.3], np.float32)
    for m in (fast, slow):
        m.setUpdateOccupiedFactor(0.9)
        m.matchData(pose0, base)               # caches the container for the levels above 0
        m.updateByScan(base, (0.0, 0.0), pose0)
    if n_points > 4266:  # the ordered kernel keeps nine terms per point in LDS: it refuses containers this long
        with pytest.raises(api.LslamError):
            slow.matchData(pose0, pts)
        p_f, H_f = fast.matchData(pose0 + np.array([0.06, -0.05, 0.02], np.float32), pts)
        assert np.abs(p_f - pose0).max() < 0.03 and np.isfinite(H_f).all()
        return
    hint = pose0 + np.array([0.06, -0.05, 0.02], np.float32)
    p_f, H_f = fast.matchData(hint, pts)
    p_s, H_s = slow.matchData(hint, pts)
    assert np.abs(p_f - p_s).max() <= 5e-5, (p_f, p_s)
    assert np.abs(H_f - H_s).max() <= 1e-2 * max(1.0, float(np.abs(H_s).max()))
    assert np.abs(p_f - pose0).max() < 0.03     # and both converge back to the pose the map was built at
    for m in (fast, slow):                      # the cached container (n_points of them) feeds levels 1 and 2
        m.updateByScan(pts, (0.0, 0.0), pose0)
    for lv in range(levels):
        assert fast.logodds(lv).tobytes() == slow.logodds(lv).tobytes(), lv


def test_batched_update_equals_sequential(ctx, oracle_lib):
    """lslam_map_update_batch: K scans marked in parallel + one apply pass == K successive updateByScan calls, bit for
    bit, on every pyramid level -- including repeats that reach the occupied clamp, an empty scan, uneven point counts,
    off-centre origos and more than 64 scans (two groups)."""
    n, cell, levels = 512, 0.05, 3
    gpu_b = api.OccGridMap(ctx, n, n, cell, (n * cell * 0.5, n * cell * 0.5), levels=levels)
    gpu_s = api.OccGridMap(ctx, n, n, cell, (n * cell * 0.5, n * cell * 0.5), levels=levels)
    cpu = oracle_lib.PortHectorRep(cell, n, n, levels)
    for m in (gpu_b, gpu_s):
        m.setUpdateOccupiedFactor(0.9)
    cpu.setUpdateFactorOccupied(0.9)
    scans = scans_for_map(9, seed=21, map_cells=n)
    seq = []
    for rep in range(9):  # 81 scans: the same walls over and over -> clamp at 50
        for k, (pts, pose) in enumerate(scans):
            pts = pts[(np.abs(pts) < 220).all(axis=1)]
            if k == 4 and rep == 1:
                pts = pts[:0]  # an empty container
            elif k % 3 == 1:
                pts = pts[: 200 + 37 * k]
            pose = (pose * np.float32(0.25)).astype(np.float32)
            seq.append((pts, np.array([0.4 * (k % 2), -0.2 * (k % 3)], np.float32), pose))
    for pts, origo, pose in seq:
        for m in (gpu_s, cpu):
            m.matchData(pose if m is gpu_s else pts, pts if m is gpu_s else pose, origo)
            m.updateByScan(pts, origo, pose)
    gpu_b.updateByScans([s[0] for s in seq], np.stack([s[1] for s in seq]), np.stack([s[2] for s in seq]))
    assert gpu_b.cached_points() == len(seq[-1][0])
    for lv in range(levels):
        a = gpu_s.logodds(lv)
        assert np.count_nonzero(a) > 1000
        assert a.tobytes() == cpu.logodds(lv).tobytes(), lv
        assert a.tobytes() == gpu_b.logodds(lv).tobytes(), lv
    # a second batch on top of the first (plane epochs advance), then a plain update in between
    gpu_b.updateByScans([s[0] for s in seq[:5]], np.stack([s[1] for s in seq[:5]]), np.stack([s[2] for s in seq[:5]]))
    for pts, origo, pose in seq[:5]:
        gpu_s.matchData(pose, pts, origo)
        gpu_s.updateByScan(pts, origo, pose)
    assert gpu_s.logodds(0).tobytes() == gpu_b.logodds(0).tobytes()


def test_batched_update_free_then_occupied_order(ctx, oracle_lib):
    """The (v+free)-free rule depends on beam order inside a scan (OccGridMapBase.h:316-330): hand-made beams, both
    orders, duplicates, through the batched path."""
    n, cell = 200, 0.05
    cpu = oracle_lib.PortHector(n, n, cell, (5.0, 5.0))
    gpu = api.OccGridMap(ctx, n, n, cell, (5.0, 5.0))
    pts = np.array([[40, 0], [20, 0], [10, 0], [30, 0], [30, 0], [0.2, 0.1], [-15, 25], [-15, 25], [-7.5, 12.5],
                    [500, 0], [0, -60]], dtype=np.float32)
    pose = np.zeros(3, np.float32)
    batch = []
    for k in range(6):
        cpu.updateByScan(pts, (0.0, 0.0), pose)
        batch.append(pts.copy())
        pts = pts[::-1].copy()
    gpu.updateByScans(batch, (0.0, 0.0), np.zeros((6, 3), np.float32))
    assert cpu.logodds().tobytes() == gpu.logodds().tobytes()


def test_laserscan_to_container_on_device(ctx, oracle_lib):
    """LaserScan -> DataContainer on the device (laser_geometry projection + the node's filters + the
    base_link <- laser transform, hector_slam.cc:193, 320-362) against the host evaluation, for EVERY beam: same points
    bit for bit, same order, same origo; then matchData / updateByScan on the resident container equal the host-fed path."""
    laser = synth.Laser()
    n, cell = 1024, 0.05
    world = synth.arena(size=40.0, n_axis=10, n_rot=4, seed=5)
    gpu_d = api.OccGridMap(ctx, n, n, cell, (n * cell * 0.5,) * 2, levels=3)  # device-projected
    gpu_h = api.OccGridMap(ctx, n, n, cell, (n * cell * 0.5,) * 2, levels=3)  # host-projected
    for m in (gpu_d, gpu_h):
        m.setUpdateOccupiedFactor(0.9)
    rng = np.random.default_rng(4)
    for laser_pose in ((0.0, 0.0, 0.0, 0.0), (0.21, -0.07, 0.35, 0.3)):
        sp = api.hector_scan(laser, laser_pose=laser_pose)
        for k in range(6):
            truth = (0.3 + 0.05 * k, -0.2 + 0.02 * k, 0.1 + 0.01 * k)
            r = synth.cast_scan(world, truth, laser, 0.01, 0.02, rng)
            r[5] = np.float32("nan"); r[17] = np.float32(0.05); r[40] = np.float32(29.99)  # filtered one way or another
            want, origo = synth.hector_project(r, laser, 1.0 / cell, laser_pose=laser_pose)
            n_pts = gpu_d.setScan(r, sp)
            got, got_origo = gpu_d.container()
            assert n_pts == len(want) > 500
            assert got.tobytes() == want.tobytes()
            assert got_origo.tobytes() == origo.tobytes()
            pose = np.asarray(truth, np.float32)
            pd, Hd = gpu_d.matchContainer(pose)
            ph, Hh = gpu_h.matchData(pose, want, origo)
            assert pd.tobytes() == ph.tobytes() and Hd.tobytes() == Hh.tobytes()
            gpu_d.updateByContainer(pose)
            gpu_h.updateByScan(want, origo, pose)
    for lv in range(3):
        assert np.count_nonzero(gpu_d.logodds(lv)) > 1000
        assert gpu_d.logodds(lv).tobytes() == gpu_h.logodds(lv).tobytes()
    assert gpu_d.setScan(np.zeros(0, np.float32), sp) == 0  # an empty LaserScan is legal


def test_pipelined_and_two_kernel_paths_agree(ctx, oracle_lib, monkeypatch):
    """updateByScan is PIPELINED (k_logodds_pipe: the apply of the previous scan and the mark of this one in one launch,
    the last apply flushed by whoever reads); LSLAM_MAP_TWO_KERNELS=1 (read at map creation) keeps mark + apply per call.
    Both must reproduce the sequential reference bit for bit, incl. the order-dependent (v + free) - free cells, the clamp
    and a 3-level pyramid fed from cached containers -- with reads, matchData calls and batched updates interleaved so
    that every flush point is crossed."""
    n, cell, LV = 512, 0.05, 3
    off = (n * cell * 0.5, n * cell * 0.5)
    one = api.OccGridMap(ctx, n, n, cell, off, levels=LV)
    monkeypatch.setenv("LSLAM_MAP_TWO_KERNELS", "1")
    two = api.OccGridMap(ctx, n, n, cell, off, levels=LV)
    monkeypatch.delenv("LSLAM_MAP_TWO_KERNELS")
    cpu = oracle_lib.PortHectorRep(cell, n, n, LV)
    for m in (one, two):
        m.setUpdateOccupiedFactor(0.97)
    cpu.setUpdateFactorOccupied(0.97)
    scans = scans_for_map(10, seed=11, map_cells=n)
    for k, (pts, pose) in enumerate(scans * 2):
        pts = pts[(np.abs(pts) < 230).all(axis=1)]
        if k % 3 == 0:  # every third scan is matched first: the levels above 0 get a fresh cached container
            hint = pose + np.array([0.02, -0.01, 0.01], np.float32)
            for m in (one, two):
                m.matchData(hint, pts)
            cpu.matchData(pts, hint)
        for m in (one, two):
            m.updateByScan(pts, (0.0, 0.0), pose)
        cpu.updateByScan(pts, (0.0, 0.0), pose)
        if k == 4:  # a read in the middle of a run of updates
            assert one.logodds(0).tobytes() == cpu.logodds(0).tobytes()
        if k == 7:  # a batched update between single ones (every level is fed the same scan: match first on the CPU side)
            for m in (one, two):
                m.updateByScans([pts, pts], (0.0, 0.0), np.stack([pose, pose]))
            for _ in range(2):
                cpu.matchData(pts, pose)
                cpu.updateByScan(pts, (0.0, 0.0), pose)
            for m in (one, two):
                m.matchData(pose, pts)
            cpu.matchData(pts, pose)
    ctx.profile(True); ctx.profile_reset()
    for _ in range(2):
        one.updateByScan(scans[0][0], (0.0, 0.0), scans[0][1])
        two.updateByScan(scans[0][0], (0.0, 0.0), scans[0][1])
        cpu.updateByScan(scans[0][0], (0.0, 0.0), scans[0][1])
    ctx.synchronize()  # flushes the pending apply of `one`
    ctx.profile(False)
    prof = ctx.profile_read()
    assert {"logodds_pipe", "logodds_mark", "logodds_apply"} <= set(prof), prof
    # pipelined (round 4: ONE launch for all 3 levels): 2 pipe launches + 1 flushed apply; two-kernel: 6 marks + 6 applies
    assert prof["logodds_pipe"][0] == 2 and prof["logodds_mark"][0] == 6 and prof["logodds_apply"][0] == 7, prof
    for lv in range(LV):
        ref = cpu.logodds(lv)
        assert np.count_nonzero(ref) > 1000
        assert one.logodds(lv).tobytes() == ref.tobytes(), lv
        assert two.logodds(lv).tobytes() == ref.tobytes(), lv


def test_pipelined_update_raw_plane_after_synchronize(ctx, oracle_lib):
    """The raw float plane (lslam_map_cells_dev_ptr) is complete after lslam_synchronize: the context flushes the
    pending apply of every map it owns before it waits."""
    n, cell = 400, 0.05
    off = (10.0, 10.0)
    cpu = oracle_lib.PortHector(n, n, cell, off)
    gpu = api.OccGridMap(ctx, n, n, cell, off)
    ptr = gpu.cells_dev_ptr(0)
    for pts, pose in scans_for_map(3, seed=5, map_cells=n):
        pts = pts[(np.abs(pts) < 150).all(axis=1)]
        pose = np.array([0.0, 0.0, float(pose[2])], np.float32)
        cpu.updateByScan(pts, (0.0, 0.0), pose)
        gpu.updateByScan(pts, (0.0, 0.0), pose)
    ctx.synchronize()
    raw = np.zeros((n, n), np.float32)
    ctx.download(ptr, raw)
    assert raw.tobytes() == cpu.logodds().tobytes()


def test_pipelined_update_across_the_epoch_wrap(ctx, oracle_lib):
    """The key planes carry a 16-bit scan epoch; when it wraps (every 65 535 single-scan updates) the planes are cleared --
    after the pending apply of the pipelined path has been flushed, or the last scan before the wrap would lose its
    update.  66 000 tiny updates (a handful of beams whose cells overlap from scan to scan) against the sequential oracle."""
    n, cell = 64, 0.05
    off = (1.6, 1.6)
    cpu = oracle_lib.PortHector(n, n, cell, off)
    gpu = api.OccGridMap(ctx, n, n, cell, off)
    for m in (cpu, gpu):
        m.setUpdateFreeFactor(0.49)      # tiny steps: 66 000 of them stay far from the +50 clamp and from float saturation
        m.setUpdateOccupiedFactor(0.51)
    rng = np.random.default_rng(9)
    pts_all = (rng.uniform(-25.0, 25.0, (16, 5, 2))).astype(np.float32)
    poses = np.zeros((16, 3), np.float32)
    poses[:, 2] = rng.uniform(-3.0, 3.0, 16)
    total = 66000
    for k in range(total):
        j = k & 15
        cpu.updateByScan(pts_all[j], (0.0, 0.0), poses[j])
        gpu.updateByScan(pts_all[j], (0.0, 0.0), poses[j])
        if k in (65533, 65534, 65535, 65536, 65537):  # reads right around the wrap (each flushes the pending apply)
            assert gpu.logodds().tobytes() == cpu.logodds().tobytes(), k
    a = cpu.logodds()
    assert np.count_nonzero(a) > 100
    assert gpu.logodds().tobytes() == a.tobytes()
