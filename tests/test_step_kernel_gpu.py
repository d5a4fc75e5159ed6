"""The scan-resident workgroup kernel (LSLAM_OPT_STEP_KERNEL, k_match_step): ONE launch per batched match -- a workgroup
of 3 or 4 waves takes a scan through scan_prep -> coarse responses -> coarse reduce -> fine responses -> fine reduce, the
response numerators in LDS.  It runs the five-kernel path's own device functions in the same order, so its records must be
byte-identical to that path's (which test_matcher_gpu.py holds to the oracle and the reference), for every batch size, for
scans whose beams all take the parked (fp64) path, for non-uniform lattices (the reduce block's generic fallback), for
unreadable beams, with and without penalties, at both waves-per-scan settings -- and every candidate's numerator must be
the restatement's (Mapper.cpp:819-856), not only the best one's."""
import math

import numpy as np
import pytest
import torch

from lslam_amd import api, synth

pytestmark = pytest.mark.gpu


def _matcher(ctx, wl, **cfg):
    gm = api.ScanMatcher(ctx, api.baseline_config(**cfg), api.laser_params(wl.laser, cfg.get("range_threshold", 49.5)))
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    return gm


@pytest.mark.parametrize("waves", [3, 4])
def test_records_equal_the_five_kernel_path(ctx, waves):
    wl = synth.make_match_workload(n_base=20, n_query=2100, seed=41, query_spread=2.0)
    wl.query_ranges[np.random.default_rng(1).random(wl.query_ranges.shape) < 0.01] = np.nan
    gm = _matcher(ctx, wl)
    for n in (64, 65, 300, 2100):
        for pen in (True, False):
            gm.set_option("step_kernel", 0)
            want = gm.match_batch(wl.query_ranges[:n], wl.query_poses[:n], doPenalize=pen)
            gm.set_option("step_kernel", waves)
            before = gm.step_kernel_launches
            got = gm.match_batch(wl.query_ranges[:n], wl.query_poses[:n], doPenalize=pen)
            assert gm.step_kernel_launches == before + 1, (n, pen)
            assert (want["status"] == 0).sum() > n // 2
            assert got.tobytes() == want.tobytes(), (n, pen)
    # below the minimum batch, without refinement: the five-kernel path, silently
    before = gm.step_kernel_launches
    small = gm.match_batch(wl.query_ranges[:8], wl.query_poses[:8])
    coarse_only = gm.match_batch(wl.query_ranges[:100], wl.query_poses[:100], doRefineMatch=False)
    assert gm.step_kernel_launches == before
    gm.set_option("step_kernel", 0)
    assert small.tobytes() == gm.match_batch(wl.query_ranges[:8], wl.query_poses[:8]).tobytes()
    assert coarse_only.tobytes() == gm.match_batch(wl.query_ranges[:100], wl.query_poses[:100], doRefineMatch=False).tobytes()
    gm.close()


@pytest.mark.parametrize("waves", [3, 4])
def test_device_entry_points_f32_f64_and_pipelined(ctx, waves):
    wl = synth.make_match_workload(n_base=20, n_query=700, seed=42, query_spread=2.0)
    gm = _matcher(ctx, wl)
    dev = torch.device("cuda", 0)
    r32 = torch.from_numpy(np.ascontiguousarray(wl.query_ranges.astype(np.float32))).to(dev)
    r64 = torch.from_numpy(np.ascontiguousarray(wl.query_ranges.astype(np.float32).astype(np.float64))).to(dev)
    p = torch.from_numpy(np.ascontiguousarray(wl.query_poses)).to(dev)

    def run(rt, dtype, depth, n_steps=1):
        gm.set_option("pipeline_depth", depth)
        outs = [torch.empty((700, 112), dtype=torch.uint8, device=dev) for _ in range(n_steps)]
        torch.cuda.synchronize()
        for o in outs:
            gm.match_batch_dev(700, rt.data_ptr(), rt.shape[1], p.data_ptr(), o.data_ptr(), dtype=dtype)
        ctx.synchronize()
        return [o.cpu().numpy().tobytes() for o in outs]

    gm.set_option("step_kernel", 0)
    want = run(r32, "f32", 1)[0]
    assert run(r64, "f64", 1)[0] == want
    gm.set_option("step_kernel", waves)
    assert run(r32, "f32", 1)[0] == want
    assert run(r64, "f64", 1)[0] == want
    for got in run(r32, "f32", 2, n_steps=5):
        assert got == want
    gm.set_option("pipeline_depth", 1)
    gm.close()


def test_every_numerator_of_whole_batches(ctx, oracle_lib, workload_spread):
    """Coarse 11 x 11 x 21 and fine 3 x 3 x 11 numerators of every scan of a batch, copied out of the kernel's LDS."""
    wl = workload_spread
    port = oracle_lib.PortKarto(oracle_lib.default_cfg(), oracle_lib.laser_struct(wl.laser))
    port.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
    gm = _matcher(ctx, wl)
    S, nq = 320, len(wl.query_ranges)
    idx = np.arange(S) % nq
    rng = np.random.default_rng(7)
    poses = wl.query_poses[idx].copy()
    far = np.arange(S) >= nq
    poses[far, :2] += rng.uniform(-0.4, 0.4, size=(int(far.sum()), 2))
    poses[far, 2] += rng.uniform(-0.3, 0.3, size=int(far.sum()))
    ranges = wl.query_ranges[idx].copy()
    ranges[rng.random(ranges.shape) < 0.01] = np.nan
    for waves in (3, 4):
        gm.set_option("step_kernel", waves)
        before = gm.step_kernel_launches
        got = gm.coarse_sums_batch(ranges, poses)
        fine, centers = gm.fine_sums_batch(ranges, poses)
        assert gm.step_kernel_launches == before + 2
        gm.set_option("step_kernel", 0)
        assert np.array_equal(got, gm.coarse_sums_batch(ranges, poses))
        fine5, centers5 = gm.fine_sums_batch(ranges, poses)
        assert np.array_equal(fine, fine5) and centers.tobytes() == centers5.tobytes()
        n_fine = 0
        for q in np.unique(np.concatenate([np.arange(nq), rng.integers(0, S, size=24)])):
            _, _, _, st, sums_cpu = port.correlate_scan(ranges[q], poses[q], poses[q], 0.5, 0.1, 0.349, 0.0349, True, False,
                                                        want_sums=True)
            assert st == 0 and np.array_equal(got[q], sums_cpu), q
            if np.isnan(centers[q]).any():
                continue
            _, _, _, st, fine_cpu = port.correlate_scan(ranges[q], poses[q], centers[q], 0.05, 0.05, 0.5 * 0.0349, 0.00349, True,
                                                        True, want_sums=True)
            if st == 0:
                assert np.array_equal(fine[q], fine_cpu), q
                n_fine += int(fine_cpu.any())
        assert n_fine >= nq // 2
    gm.close()


@pytest.mark.parametrize("heading", [0.0, math.pi / 2])
def test_all_beams_on_the_parked_path(ctx, oracle_lib, heading):
    """test_matcher_gpu.py::test_beams_on_half_cell_boundaries' scan -- every beam inside the fp32 estimate's error band for
    one candidate heading -- through the step kernel, whose parked beams are lane masks per block of 64 (not a list)."""
    laser = synth.Laser()
    port = oracle_lib.PortKarto(oracle_lib.default_cfg(), oracle_lib.laser_struct(laser))
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(laser))
    n = laser.n_ranges
    phi = laser.angle_min + np.arange(n) * laser.angle_increment
    b = np.arange(n)
    on_x = np.abs(np.cos(phi)) > 0.6
    off = np.where(b % 2 == 0, 3e-7, -3e-7)
    r = np.where(on_x, ((100 + b % 40) + 0.5 + off) / 20.0 / np.abs(np.cos(phi)),
                 ((60 + b % 30) + 0.5 + off) / 20.0 / np.maximum(np.abs(np.sin(phi)), 1e-3))
    r[::97] = np.nan
    base_poses = np.array([[1.0, 2.0, heading], [1.05, 2.0, heading], [1.0, 1.95, heading]])
    center = np.array([1.03, 1.98, heading])
    port.set_base_scans(np.stack([r, r, r]), base_poses, center)
    gm.AddScans(np.stack([r, r, r]), base_poses, center)
    _, _, _, st, sums_cpu = port.correlate_scan(r, center, center, 0.5, 0.1, 0.349, 0.0349, True, False, want_sums=True)
    assert st == 0 and sums_cpu.any()
    S = 192
    want = gm.match_batch(np.tile(r, (S, 1)), np.tile(center, (S, 1)))
    mean, cov, resp = port.match(r, center)
    assert np.abs(want["pose"][0] - mean).max() <= 1e-9 and abs(want["response"][0] - resp) <= 1e-12
    for waves in (3, 4):
        gm.set_option("step_kernel", waves)
        got = gm.match_batch(np.tile(r, (S, 1)), np.tile(center, (S, 1)))
        assert got.tobytes() == want.tobytes()
        sums = gm.coarse_sums_batch(np.tile(r, (S, 1)), np.tile(center, (S, 1)))
        assert np.array_equal(sums[0], sums_cpu) and np.array_equal(sums[S - 1], sums_cpu)
    gm.close()


def test_non_uniform_lattices_take_the_fallback(ctx):
    """Centres half a cell off the raster: the packed response functions skip the scan and its reduce phases compute the
    numerators themselves (block_generic_fallback) -- inside the step kernel, into LDS."""
    laser = synth.Laser()
    thr = 12.0
    wl = synth.make_match_workload(n_base=20, n_query=24, seed=11, laser=laser)
    gm = api.ScanMatcher(ctx, api.baseline_config(range_threshold=thr), api.laser_params(laser, thr))
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    off = gm.grid_info()["offset"]
    poses = wl.query_poses.copy()
    for q in range(len(poses)):
        ax = q & 1
        v = (poses[q, ax] - off[ax]) * 20.0
        k = math.floor(v + 0.5) if v >= 0 else math.ceil(v - 0.5)
        poses[q, ax] = off[ax] + (k + 0.5) * 0.05
    ranges, poses = np.tile(wl.query_ranges, (4, 1)), np.tile(poses, (4, 1))
    want = gm.match_batch(ranges, poses)
    assert (want["status"] == 0).all()
    for waves in (3, 4):
        gm.set_option("step_kernel", waves)
        before = gm.step_kernel_launches
        got = gm.match_batch(ranges, poses)
        assert gm.step_kernel_launches == before + 1
        assert got.tobytes() == want.tobytes()
    gm.close()


def test_reference_indoor_default_config_takes_the_five_kernel_path(ctx):
    """mapper_params.yaml (res 0.01, response expansion on): expansion passes keep the five-kernel path; same records."""
    laser = synth.Laser()
    wl = synth.make_match_workload(n_base=12, n_query=80, seed=43, laser=laser, query_spread=0.5)
    cfg = dict(search_size=0.3, resolution=0.01, smear_deviation=0.03, range_threshold=12.0, use_response_expansion=1)
    gm = api.ScanMatcher(ctx, api.baseline_config(**cfg), api.laser_params(laser, 12.0))
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    want = gm.match_batch(wl.query_ranges, wl.query_poses)
    gm.set_option("step_kernel", 3)
    before = gm.step_kernel_launches
    got = gm.match_batch(wl.query_ranges, wl.query_poses)
    assert gm.step_kernel_launches == before and got.tobytes() == want.tobytes()
    gm.close()
    # the same configuration without the expansion: the <4,8> instantiation of the step kernel (16 x 16 x 21 lattice)
    cfg["use_response_expansion"] = 0
    gm = api.ScanMatcher(ctx, api.baseline_config(**cfg), api.laser_params(laser, 12.0))
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    want = gm.match_batch(wl.query_ranges, wl.query_poses)
    for waves in (3, 4):
        gm.set_option("step_kernel", waves)
        before = gm.step_kernel_launches
        got = gm.match_batch(wl.query_ranges, wl.query_poses)
        assert gm.step_kernel_launches == before + 1 and got.tobytes() == want.tobytes()
        gm.set_option("step_kernel", 0)
    gm.close()
