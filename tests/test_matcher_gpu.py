"""GPU parity tests of the correlative scan matcher: the HIP path (through the C ABI) against the
CPU oracle on identical seeded inputs.  Integer/byte/index work (correlation grid, smear kernel,
valid-point mask, lookup tables, response numerators) must be BIT-EXACT; poses, covariances and
responses are fp64 results of the same expression order and must agree to 1e-9 (the north-star
tolerance is 1e-4 m / 1e-4 rad; identical lattice arg-max makes them agree to ~1e-15)."""
import math

import numpy as np
import pytest

from lslam_amd import api, synth

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-9  # m / rad   (north star: 1e-4)
COV_TOL = 1e-9


def make_pair(ctx, oracle_lib, laser=synth.Laser(), cfg_kw=None, range_threshold=49.5, offset=(0, 0, 0)):
    cfg_kw = cfg_kw or {}
    ocfg = oracle_lib.default_cfg(**{k: v for k, v in cfg_kw.items() if k in (
        "search_size", "resolution", "smear_deviation", "use_response_expansion")})
    ol = oracle_lib.laser_struct(laser, range_threshold, offset)
    port = oracle_lib.PortKarto(ocfg, ol)
    gcfg = api.baseline_config(range_threshold=range_threshold, **cfg_kw)
    gm = api.ScanMatcher(ctx, gcfg, api.laser_params(laser, range_threshold, offset))
    return port, gm


def test_geometry_and_kernel(ctx, oracle_lib):
    port, gm = make_pair(ctx, oracle_lib)
    gi, oi = gm.grid_info(), port.grid_info()
    for k in ("width", "height", "stride", "roi_x", "roi_y", "roi_w", "roi_h", "kernel_size"):
        assert gi[k] == oi[k]
    assert (gi["width"], gi["stride"], gi["roi_w"]) == (2005, 2008, 2001)  # SURVEY.md §8 sizes table
    assert gm.num_beams == port.num_beams == 1081
    assert np.array_equal(gm.kernel(), port.kernel())
    assert gm.kernel().tolist() == [[6, 25, 6], [25, 100, 25], [6, 25, 6]]


@pytest.mark.parametrize("res,smear", [(0.025, 0.03), (0.01, 0.03)])
def test_kernel_other_resolutions(ctx, oracle_lib, res, smear):
    port, gm = make_pair(ctx, oracle_lib, cfg_kw=dict(resolution=res, smear_deviation=smear, search_size=0.3),
                         range_threshold=12.0)
    assert np.array_equal(gm.kernel(), port.kernel())
    assert gm.grid_info()["stride"] == port.grid_info()["stride"]


def test_valid_mask_and_grid(ctx, oracle_lib, workload):
    wl = workload
    port, gm = make_pair(ctx, oracle_lib)
    center = wl.query_poses[0]
    for s in range(4):
        pts = port.point_readings(wl.base_ranges[s], wl.base_poses[s])
        valid = port.find_valid_points(pts, center[:2])
        mask = gm.valid_mask(wl.base_ranges[s], wl.base_poses[s], center[:2])
        # the reference keeps an order-preserving subset: compare as point lists
        assert np.array_equal(pts[mask.astype(bool)], valid)
    port.set_base_scans(wl.base_ranges, wl.base_poses, center)
    gm.AddScans(wl.base_ranges, wl.base_poses, center)
    g_gpu, g_cpu = gm.GetCorrelationGrid(), port.grid()
    assert g_cpu.any()
    assert np.array_equal(g_gpu, g_cpu)
    assert np.array_equal(gm.grid_info()["offset"], port.grid_info()["offset"])


@pytest.mark.parametrize("res,smear,ks", [(0.05, 0.03, 3), (0.025, 0.03, 5), (0.05, 0.08, 7), (0.05, 0.45, 37)])
def test_grid_rebuild_all_smear_paths(ctx, oracle_lib, workload, res, smear, ks):
    """AddScans with the smear as a gather pass: the compile-time half sizes (3x3, 5x5), the generic form (7x7) and a
    kernel wider than the gather tabulates (37x37 -> listed scatter path); grid bytes AND a match equal the oracle's."""
    wl = workload
    kw = dict(search_size=1.0 if res >= 0.05 else 0.5, resolution=res, smear_deviation=smear)
    port, gm = make_pair(ctx, oracle_lib, cfg_kw=kw, range_threshold=20.0)
    assert gm.grid_info()["kernel_size"] == ks
    center = wl.query_poses[0]
    port.set_base_scans(wl.base_ranges[:10], wl.base_poses[:10], center)
    gm.AddScans(wl.base_ranges[:10], wl.base_poses[:10], center)
    g_gpu, g_cpu = gm.GetCorrelationGrid(), port.grid()
    assert g_cpu.any() and np.array_equal(g_gpu, g_cpu)
    # the coarse pass reads the parity planes the same pass wrote
    res_g = gm.match_batch(wl.query_ranges[:2], wl.query_poses[:2])
    for q in range(2):
        mean, cov, resp = port.match(wl.query_ranges[q], wl.query_poses[q])
        _assert_result(res_g[q], mean, cov, resp)


def test_lookup_tables_bit_exact(ctx, oracle_lib, workload):
    wl = workload
    port, gm = make_pair(ctx, oracle_lib)
    center = wl.query_poses[0]
    port.set_base_scans(wl.base_ranges[:2], wl.base_poses[:2], center)
    gm.AddScans(wl.base_ranges[:2], wl.base_poses[:2], center)
    for q in range(3):
        r, p = wl.query_ranges[q].copy(), wl.query_poses[q]
        r[5] = np.nan  # INVALID_SCAN entries
        r[77] = np.inf
        for (off, res) in ((0.349, 0.0349), (0.5 * 0.0349, 0.00349), (0.349 + math.radians(20), 0.0349)):
            t_cpu = port.compute_offsets(r, p, p[2], off, res)
            t_gpu = gm.lookup_table(r, p, p[2], off, res)
            assert t_gpu.shape == t_cpu.shape
            assert np.array_equal(t_gpu, t_cpu)
            assert (t_gpu[:, 5] == np.iinfo(np.int32).max).all()


def test_coarse_response_sums_bit_exact(ctx, oracle_lib, workload):
    wl = workload
    port, gm = make_pair(ctx, oracle_lib)
    center = wl.query_poses[0]
    port.set_base_scans(wl.base_ranges, wl.base_poses, center)
    gm.AddScans(wl.base_ranges, wl.base_poses, center)
    for q in range(3):
        r, p = wl.query_ranges[q], wl.query_poses[q]
        _, _, _, st, sums_cpu = port.correlate_scan(r, p, p, 0.5, 0.1, 0.349, 0.0349, True, False, want_sums=True)
        assert st == 0 and sums_cpu.any()
        fast = gm.coarse_sums(r, p, force_generic=False)
        slow = gm.coarse_sums(r, p, force_generic=True)
        assert np.array_equal(fast, sums_cpu)  # packed LDS-reduced lattice kernel
        assert np.array_equal(slow, sums_cpu)  # generic per-candidate kernel


@pytest.mark.parametrize("S", [40, 160, 2304])
def test_coarse_sums_of_whole_batches_bit_exact(ctx, oracle_lib, workload_spread, S):
    """The coarse numerators of EVERY scan and EVERY candidate of a batch (GetResponse, Mapper.cpp:819-856; 11 x 11 x 21 per
    scan), through the launches a match of that batch size takes: 40 scans = beam slices on the linear parity planes, table
    cells on the reference's fp64 tree; 160 = the tiled planes, cells decided on the fp32 estimate (the hot kernel of the
    bench); 2304 = the same behind one prep block per scan.  A match result only shows the best candidate's sum; here a
    wrong sum anywhere on a lattice would show."""
    wl = workload_spread
    port, gm = make_pair(ctx, oracle_lib)
    port.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    nq = len(wl.query_ranges)
    idx = np.arange(S) % nq
    poses = wl.query_poses[idx].copy()
    rng = np.random.default_rng(S)
    far = np.arange(S) >= nq  # the repeats get poses of their own: different lattices, different table cells
    poses[far, :2] += rng.uniform(-0.4, 0.4, size=(int(far.sum()), 2))
    poses[far, 2] += rng.uniform(-0.3, 0.3, size=int(far.sum()))
    ranges = wl.query_ranges[idx].copy()
    ranges[rng.random(ranges.shape) < 0.01] = np.nan
    got = gm.coarse_sums_batch(ranges, poses)
    assert got.shape[0] == S and got.any()
    check = np.unique(np.concatenate([np.arange(min(S, nq)), rng.integers(0, S, size=40)]))
    for q in check:
        _, _, _, st, sums_cpu = port.correlate_scan(ranges[q], poses[q], poses[q], 0.5, 0.1, 0.349, 0.0349, True, False,
                                                    want_sums=True)
        assert st == 0
        assert np.array_equal(got[q], sums_cpu), q
    # the fine pass of the same batches (3 x 3 x 11 around the device's own coarse mean, Mapper.cpp:276-281): k_resp_rows<1,4>
    # with beam slices at 40 scans, the 4x4-block kernel k_resp_tile3 (one / three angles per wave) at 160 / 2304
    fine, centers = gm.fine_sums_batch(ranges, poses)
    n_fine = 0
    for q in check:
        if np.isnan(centers[q]).any():
            continue
        _, _, _, st, fine_cpu = port.correlate_scan(ranges[q], poses[q], centers[q], 0.05, 0.05, 0.5 * 0.0349, 0.00349, True, True,
                                                    want_sums=True)
        if st == 0:
            assert np.array_equal(fine[q], fine_cpu), q
            n_fine += int(fine_cpu.any())
    assert n_fine >= len(check) // 2


@pytest.mark.parametrize("heading", [0.0, math.pi / 2])
def test_beams_on_half_cell_boundaries(ctx, oracle_lib, heading):
    """Phase A of k_resp_rows decides a beam's table cell on an fp32 estimate and parks the beams whose coordinate lies
    within the estimate's error band of a half-integer for the reference's own fp64 expression.  Here EVERY beam sits on
    such a boundary for one candidate heading (scan-frame x or y = (k + 0.5 +- 3e-7) cells: where the reference rounds is
    beyond doubt in fp64 -- it does not hinge on the last ulp of the device's and glibc's sin/cos -- and undecidable in
    fp32), so whole waves of beams take the parked path; the numerators must still be the reference's bit for bit."""
    laser = synth.Laser()
    port, gm = make_pair(ctx, oracle_lib, laser)
    n = laser.n_ranges
    phi = laser.angle_min + np.arange(n) * laser.angle_increment
    b = np.arange(n)
    on_x = np.abs(np.cos(phi)) > 0.6
    off = np.where(b % 2 == 0, 3e-7, -3e-7)  # cells: far inside the fp32 band (~1e-4), far outside libm's last-ulp noise
    r = np.where(on_x, ((100 + b % 40) + 0.5 + off) / 20.0 / np.abs(np.cos(phi)),
                 ((60 + b % 30) + 0.5 + off) / 20.0 / np.maximum(np.abs(np.sin(phi)), 1e-3))
    r[::97] = np.nan  # a few INVALID_SCAN readings among them
    base_poses = np.array([[1.0, 2.0, heading], [1.05, 2.0, heading], [1.0, 1.95, heading]])
    base_ranges = np.stack([r, r, r])
    center = np.array([1.03, 1.98, heading])  # candidate 10 of 21: (heading - 0.349) + 10 * 0.0349 = heading (+- 1 ulp)
    port.set_base_scans(base_ranges, base_poses, center)
    gm.AddScans(base_ranges, base_poses, center)
    _, _, _, st, sums_cpu = port.correlate_scan(r, center, center, 0.5, 0.1, 0.349, 0.0349, True, False, want_sums=True)
    assert st == 0 and sums_cpu.any()
    assert np.array_equal(gm.coarse_sums(r, center, force_generic=False), sums_cpu)
    assert np.array_equal(gm.coarse_sums(r, center, force_generic=True), sums_cpu)
    # the same through the batched kernels (tiled planes, beam slices) and the whole match
    S = 2304
    res = gm.match_batch(np.tile(r, (S, 1)), np.tile(center, (S, 1)))
    mean, cov, resp = port.match(r, center)
    for i in (0, 1, S - 1):
        _assert_result(res[i], mean, cov, resp)
    assert np.ptp(res["response"]) == 0.0


def _assert_result(res, mean, cov, resp):
    assert res["status"] == 0
    assert np.abs(res["pose"][:2] - mean[:2]).max() <= POSE_TOL
    assert abs(math.remainder(res["pose"][2] - mean[2], 2 * math.pi)) <= POSE_TOL
    assert np.abs(res["covariance"] - cov).max() <= COV_TOL * max(1.0, np.abs(cov).max())
    assert abs(res["response"] - resp) <= 1e-12


def test_match_batch_vs_oracle(ctx, oracle_lib, workload_spread):
    """cfg 4 in miniature: independent scans against one shared grid."""
    wl = workload_spread
    port, gm = make_pair(ctx, oracle_lib)
    port.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    res = gm.match_batch(wl.query_ranges, wl.query_poses)
    n_good = 0
    for q in range(len(res)):
        mean, cov, resp = port.match(wl.query_ranges[q], wl.query_poses[q])
        _assert_result(res[q], mean, cov, resp)
        n_good += resp > 0.3
    assert n_good >= len(res) // 2  # the workload is not degenerate


def test_match_batch_no_penalty_no_refine(ctx, oracle_lib, workload_spread):
    wl = workload_spread
    port, gm = make_pair(ctx, oracle_lib)
    port.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    for pen, ref in ((False, True), (True, False), (False, False)):
        res = gm.match_batch(wl.query_ranges[:6], wl.query_poses[:6], doPenalize=pen, doRefineMatch=ref)
        for q in range(6):
            mean, cov, resp = port.match(wl.query_ranges[q], wl.query_poses[q], pen, ref)
            _assert_result(res[q], mean, cov, resp)


def test_match_scan_full(ctx, oracle_lib, workload):
    """cfg 3: the complete MatchScan (grid rebuilt around each query)."""
    wl = workload
    port, gm = make_pair(ctx, oracle_lib)
    errs = []
    for q in range(6):
        mean, cov, resp = port.match_scan(wl.base_ranges, wl.base_poses, wl.query_ranges[q], wl.query_poses[q])
        r, m, c = gm.MatchScan(wl.query_ranges[q], wl.query_poses[q], wl.base_ranges, wl.base_poses)
        assert abs(r - resp) <= 1e-12
        assert np.abs(m - mean).max() <= POSE_TOL
        assert np.abs(c - cov).max() <= COV_TOL * max(1.0, np.abs(cov).max())
        errs.append(np.hypot(*(m[:2] - wl.truth_poses[q][:2])))
    # and the matcher does its job on this workload: most queries land within ~one coarse cell of truth
    assert np.median(errs) < 0.12


def test_device_resident_f32_batch(ctx, oracle_lib, workload_spread):
    """LaserScan-in path: float32 ranges + poses + results resident in HBM (bench.py's path)."""
    wl = workload_spread
    port, gm = make_pair(ctx, oracle_lib)
    port.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    S, n = wl.query_ranges.shape
    r32 = wl.query_ranges.astype(np.float32)
    assert np.array_equal(r32.astype(np.float64), wl.query_ranges)  # synthetic ranges are float32-exact
    d_r, d_p, d_o = ctx.alloc(r32.nbytes), ctx.alloc(S * 24), ctx.alloc(S * 112)
    ctx.upload(d_r, r32)
    ctx.upload(d_p, wl.query_poses)
    gm.match_batch_dev(S, d_r, n, d_p, d_o, dtype="f32")
    out = np.zeros(S, dtype=api.RESULT_DTYPE)
    ctx.synchronize()
    ctx.download(d_o, out)
    ref = gm.match_batch(wl.query_ranges, wl.query_poses)
    assert out.tobytes() == ref.tobytes()
    for p in (d_r, d_p, d_o):
        ctx.free(p)


def test_edge_cases(ctx, oracle_lib, workload):
    wl = workload
    port, gm = make_pair(ctx, oracle_lib)
    center = wl.query_poses[0]
    port.set_base_scans(wl.base_ranges, wl.base_poses, center)
    gm.AddScans(wl.base_ranges, wl.base_poses, center)
    n = wl.query_ranges.shape[1]
    cases = []
    allnan = np.full(n, np.nan)
    cases.append((allnan, wl.query_poses[0]))                    # every reading invalid -> response 0
    cases.append((np.full(n, np.inf), wl.query_poses[0]))
    far = np.full(n, 59.9)
    cases.append((far, wl.query_poses[1]))                       # endpoints off the grid (1-D index check)
    tiny = np.full(n, 0.01)
    cases.append((tiny, wl.query_poses[2]))                      # below minimum range: still in the tables
    mixed = wl.query_ranges[3].copy()
    mixed[::3] = np.nan
    mixed[1::7] = np.inf
    cases.append((mixed, wl.query_poses[3]))
    # sensor pose near the rim of the grid: candidate rows run over the flat-index bounds
    rim = center.copy()
    rim[0] += 49.0
    rim[1] -= 49.0
    cases.append((wl.query_ranges[4], rim))
    ranges = np.stack([c[0] for c in cases])
    poses = np.stack([c[1] for c in cases])
    res = gm.match_batch(ranges, poses)
    for i in range(len(cases)):
        mean, cov, resp = port.match(ranges[i], poses[i])
        _assert_result(res[i], mean, cov, resp)
    assert res["response"][0] == 0.0


def test_out_of_grid_is_reported_not_crashed(ctx, oracle_lib, workload):
    """The reference throws karto::Exception when a candidate cell leaves the grid (Karto.h:4488-4499)."""
    wl = workload
    port, gm = make_pair(ctx, oracle_lib)
    gm.AddScans(wl.base_ranges[:2], wl.base_poses[:2], wl.query_poses[0])
    port.set_base_scans(wl.base_ranges[:2], wl.base_poses[:2], wl.query_poses[0])
    away = wl.query_poses[0].copy()
    away[0] += 500.0
    res = gm.match_batch(wl.query_ranges[:1], away[None, :])
    assert res["status"][0] == -3
    with pytest.raises(RuntimeError):
        port.match(wl.query_ranges[0], away)


def test_response_expansion(ctx, oracle_lib, workload):
    wl = workload
    port, gm = make_pair(ctx, oracle_lib, cfg_kw=dict(use_response_expansion=1))
    center = wl.query_poses[0]
    port.set_base_scans(wl.base_ranges, wl.base_poses, center)
    gm.AddScans(wl.base_ranges, wl.base_poses, center)
    # a scan rotated by 60 deg: nothing matches inside +-20 deg, expansion finds it
    rot = wl.query_poses[0].copy()
    rot[2] = math.remainder(wl.truth_poses[0][2] + math.radians(45.0), 2 * math.pi)
    rot[:2] = wl.truth_poses[0][:2]
    ranges = np.stack([wl.query_ranges[0], wl.query_ranges[1], np.full(wl.query_ranges.shape[1], np.nan)])
    poses = np.stack([rot, wl.query_poses[1], wl.query_poses[2]])
    res = gm.match_batch(ranges, poses)
    for i in range(3):
        mean, cov, resp = port.match(ranges[i], poses[i])
        _assert_result(res[i], mean, cov, resp)
    assert res["flags"][2] & 1  # all-NaN scan: response stayed 0 through every expansion


def test_invalid_create_parameters(ctx):
    lp = api.laser_params(synth.Laser())
    assert api.ScanMatcher.Create(ctx, api.baseline_config(resolution=0.0), lp) is None  # Mapper.cpp:130-145
    assert api.ScanMatcher.Create(ctx, api.baseline_config(search_size=-1.0), lp) is None
    with pytest.raises(api.LslamError) as e:  # CalculateKernel throws (Mapper.h:1041-1053)
        api.ScanMatcher(ctx, api.baseline_config(smear_deviation=0.001), lp)
    assert e.value.code == -7


def test_laser_offset_pose_helpers(ctx, oracle_lib):
    laser = synth.Laser()
    port, gm = make_pair(ctx, oracle_lib, laser=laser, offset=(0.2, -0.05, 0.3))
    rng = np.random.default_rng(0)
    for _ in range(20):
        robot = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(-3.1, 3.1)])
        s_cpu, s_gpu = port.sensor_pose_from_robot(robot), gm.sensor_pose_from_robot(robot)
        assert np.array_equal(s_cpu, s_gpu)
        assert np.array_equal(port.robot_pose_from_sensor(s_cpu), gm.robot_pose_from_sensor(s_cpu))


def test_full_size_properties(ctx):
    """BASELINE size (1081 beams, 2005x2005 grid, 4096-scan batch is bench.py's job; here 512):
    size-independent properties -- a batch equals its halves, scans are independent of their
    batch neighbours, and matching a base scan from its own pose returns that pose."""
    wl = synth.make_match_workload(n_base=70, n_query=8, seed=11, query_spread=1.5)
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(wl.laser))
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    S = 512
    idx = np.arange(S) % len(wl.query_ranges)
    ranges, poses = wl.query_ranges[idx], wl.query_poses[idx]
    full = gm.match_batch(ranges, poses)
    assert (full["status"] == 0).all()
    for i in range(len(wl.query_ranges), S):  # periodic input -> periodic output, bit for bit
        assert full[i].tobytes() == full[i % len(wl.query_ranges)].tobytes()
    half = gm.match_batch(ranges[: S // 2][::-1], poses[: S // 2][::-1])[::-1]
    assert half.tobytes() == full[: S // 2].tobytes()
    # the full BASELINE batch (4096 scans: tiled parity planes + 4x4-block fine kernel) equals the same
    # scans matched 64 at a time (linear planes + row kernels): the layouts change nothing
    idx4 = np.arange(4096) % len(wl.query_ranges)
    pert = synth.perturb(wl.query_poses[idx4], 0.2, math.radians(6.0), 99)  # 4096 different search centres
    big = gm.match_batch(wl.query_ranges[idx4], pert)
    assert (big["status"] == 0).all()
    for lo in (0, 1984, 4032):
        small = gm.match_batch(wl.query_ranges[idx4][lo:lo + 64], pert[lo:lo + 64])
        assert small.tobytes() == big[lo:lo + 64].tobytes()
    # self-match: the last base scan matched from its own pose stays within one fine cell
    own = gm.match_batch(wl.base_ranges[-1:], wl.base_poses[-1:])
    assert own["response"][0] > 0.5
    assert np.abs(own["pose"][0][:2] - wl.base_poses[-1][:2]).max() <= 0.05 + 1e-9


def test_reference_indoor_default_config(ctx, oracle_lib):
    """The configuration the reference ships (lesson6/config/mapper_params.yaml): 0.01 m cells,
    0.3 m search space, 12 m range threshold -> 2445x2445 grid, 13x13 smear kernel, 16x16x21 coarse
    lattice, response expansion on.  Exercises the <4,8> packed kernel, the wide smear and 31-byte rows."""
    laser = synth.Laser(range_max=30.0)
    kw = dict(search_size=0.3, resolution=0.01, smear_deviation=0.03, use_response_expansion=1)
    port, gm = make_pair(ctx, oracle_lib, laser=laser, cfg_kw=kw, range_threshold=12.0)
    assert gm.grid_info()["width"] == 2445 and gm.grid_info()["kernel_size"] == 13
    world = synth.arena(size=24.0, n_axis=8, n_rot=3, seed=12)
    wl = synth.make_match_workload(n_base=12, n_query=6, seed=12, laser=laser, world=world, err_xy=0.08,
                                   err_th=math.radians(6.0), query_spread=0.5)
    port.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    assert np.array_equal(gm.GetCorrelationGrid(), port.grid())
    p0 = wl.query_poses[0]
    _, _, _, st, sums = port.correlate_scan(wl.query_ranges[0], p0, p0, 0.15, 0.02, 0.349, 0.0349, True, False,
                                            want_sums=True)
    assert st == 0
    assert np.array_equal(gm.coarse_sums(wl.query_ranges[0], p0), sums)
    res = gm.match_batch(wl.query_ranges, wl.query_poses)
    good = 0
    for q in range(len(res)):
        mean, cov, resp = port.match(wl.query_ranges[q], wl.query_poses[q])
        _assert_result(res[q], mean, cov, resp)
        good += resp > 0.3
    assert good >= 3
    # the same scans 20x: enough waves for the tiled parity planes (<4,8,true>, two 8-row passes) and the
    # 4x4-block fine kernel, with the expansion passes in between -- byte-identical to the small batch
    big = gm.match_batch(np.tile(wl.query_ranges, (20, 1)), np.tile(wl.query_poses, (20, 1)))
    assert big.tobytes() == np.tile(res, 20).tobytes()
    # 2052 scans: the chip-filling variants (128-thread reduce blocks with more lattice cells than threads, one
    # scan_prep block per scan) must give the same bytes again
    huge = gm.match_batch(np.tile(wl.query_ranges, (342, 1)), np.tile(wl.query_poses, (342, 1)))
    assert huge.tobytes() == np.tile(res, 342).tobytes()
    # full MatchScan too (grid rebuilt with the 13x13 smear around the query)
    mean, cov, resp = port.match_scan(wl.base_ranges, wl.base_poses, wl.query_ranges[1], wl.query_poses[1])
    r, m, c = gm.MatchScan(wl.query_ranges[1], wl.query_poses[1], wl.base_ranges, wl.base_poses)
    assert abs(r - resp) <= 1e-12 and np.abs(m - mean).max() <= POSE_TOL


@pytest.mark.parametrize("search_size,nx", [(8.0, 81), (10.0, 101)])
def test_loop_closure_size_lattice(ctx, oracle_lib, search_size, nx):
    """Next-row #3: the loop-closure matcher instance (Mapper.cpp:862-871): search space 8 m (library
    default) / 10 m (mapper_params.yaml) at 0.05 m -> 81^2 / 101^2 positions x 21 angles, coarse pass
    only and no odometry penalty (TryCloseLoop, Mapper.cpp:991).  Dense big-lattice kernel."""
    laser = synth.Laser(range_max=30.0)
    kw = dict(search_size=search_size, resolution=0.05, smear_deviation=0.03)
    port, gm = make_pair(ctx, oracle_lib, laser=laser, cfg_kw=kw, range_threshold=12.0)
    world = synth.arena(size=40.0, n_axis=12, n_rot=4, seed=14)
    wl = synth.make_match_workload(n_base=16, n_query=3, seed=14, laser=laser, world=world, err_xy=2.0,
                                   err_th=math.radians(12.0), query_spread=1.0)
    port.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    assert np.array_equal(gm.GetCorrelationGrid(), port.grid())
    p0 = wl.query_poses[0]
    off = 0.5 * (round(search_size / 0.05)) * 0.05
    _, _, _, st, sums = port.correlate_scan(wl.query_ranges[0], p0, p0, off, 0.1, 0.349, 0.0349, False, False,
                                            want_sums=True)
    assert st == 0 and sums.shape == (nx, nx, 21) and sums.any()
    assert np.array_equal(gm.coarse_sums(wl.query_ranges[0], p0), sums)
    res = gm.match_batch(wl.query_ranges, wl.query_poses, doPenalize=False, doRefineMatch=False)
    for q in range(len(res)):
        mean, cov, resp = port.match(wl.query_ranges[q], wl.query_poses[q], False, False)
        _assert_result(res[q], mean, cov, resp)
    # the dense kernel's zero-window pruning (8x8-block occupancy -> one live mask per beam and row tile) is exact: with
    # it switched off every numerator and every record is the same; and a search centre near the grid's edge -- windows
    # that hang over the first row / wrap past widthStep stay live -- still equals the oracle
    gm.set_option("row_occupancy", 0)
    assert np.array_equal(gm.coarse_sums(wl.query_ranges[0], p0), sums)
    res_off = gm.match_batch(wl.query_ranges, wl.query_poses, doPenalize=False, doRefineMatch=False)
    gm.set_option("row_occupancy", 1)
    assert res_off.tobytes() == res.tobytes()
    gi = gm.grid_info()
    edge = np.array([gi["offset"][0] + 0.5 * search_size + 0.3, gi["offset"][1] + 0.5 * search_size + 0.2, p0[2]])
    _, _, _, st_e, sums_e = port.correlate_scan(wl.query_ranges[0], edge, edge, off, 0.1, 0.349, 0.0349, False, False,
                                                want_sums=True)
    if st_e == 0:
        assert np.array_equal(gm.coarse_sums(wl.query_ranges[0], edge), sums_e)
    # a batch large enough for the 4-row-slot form of the kernel (other tile geometry, same masks' meaning)
    many_r = np.repeat(wl.query_ranges, 24, axis=0)
    many_p = np.repeat(wl.query_poses, 24, axis=0) + np.linspace(0, 0.4, 72)[:, None] * np.array([1.0, -0.5, 0.02])
    res_m = gm.match_batch(many_r, many_p, doPenalize=False, doRefineMatch=False)
    for q in (0, 17, 40, 71):
        mean, cov, resp = port.match(many_r[q], many_p[q], False, False)
        _assert_result(res_m[q], mean, cov, resp)
    # with refinement too (TryCloseLoop's second, fine MatchScan uses the sequential matcher; this only checks
    # that a fine pass behind a big coarse lattice works)
    res = gm.match_batch(wl.query_ranges[:1], wl.query_poses[:1], doPenalize=False, doRefineMatch=True)
    mean, cov, resp = port.match(wl.query_ranges[0], wl.query_poses[0], False, True)
    _assert_result(res[0], mean, cov, resp)
    # an all-zero response surface: every one of the 138 k / 214 k candidates ties with the best (0), far more than the
    # tie list holds -> the bitmap path; the mean is the average over the whole lattice, the covariance the maximum
    blind = np.full_like(wl.query_ranges[:1], np.nan)
    res = gm.match_batch(blind, wl.query_poses[:1], doPenalize=False, doRefineMatch=False)
    mean, cov, resp = port.match(blind[0], wl.query_poses[0], False, False)
    assert resp == 0.0
    _assert_result(res[0], mean, cov, resp)
