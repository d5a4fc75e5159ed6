"""Soak: random worlds, window sizes, odometry errors and invalid readings; 256-scan batches (large enough
for the tiled coarse planes and the 4x4-block fine kernel) compared with the reference itself where
oracle/_ref is available (else with the pinned restatement).  Responses must be exact, poses and
covariances equal to rounding."""
import math

import numpy as np
import pytest

from lslam_amd import api, synth

pytestmark = pytest.mark.gpu


import os

# (seed, batch): 256 = the wide-block reduce kernels, 2304 = the chip-filling variants (128 / 64-thread reduce blocks,
# one scan_prep block per scan).  LSLAM_SOAK_SEEDS="10,11,12" adds 2304-scan batches for those seeds (ad-hoc soak).
_CASES = [(0, 256), (1, 256), (2, 256), (3, 2304)] + [(int(x), 2304) for x in os.environ.get("LSLAM_SOAK_SEEDS", "").split(",") if x]


@pytest.mark.parametrize("seed,B", _CASES)
def test_random_world_batches_vs_reference(ctx, oracle_lib, seed, B):
    po = oracle_lib
    rng = np.random.default_rng(100 + seed)
    laser = synth.Laser()
    world = synth.arena(size=rng.uniform(30, 90), n_axis=int(rng.integers(6, 30)), n_rot=int(rng.integers(2, 10)),
                        seed=200 + seed)
    wl = synth.make_match_workload(n_base=int(rng.integers(10, 70)), n_query=32, seed=300 + seed, laser=laser,
                                   world=world, query_spread=rng.uniform(0.5, 4.0))
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(laser))
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    idx = np.arange(B) % 32
    poses = synth.perturb(wl.truth_poses[idx], rng.uniform(0.05, 0.45), math.radians(rng.uniform(2, 18)), 400 + seed)
    ranges = wl.query_ranges[idx].copy()
    ranges[rng.random(ranges.shape) < 0.01] = np.inf
    res = gm.match_batch(ranges, poses)
    if po.have_ref():
        ref = po.RefKarto(po.default_cfg(), po.laser_struct(laser))
        ref.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
        _, c_poses, c_covs, c_resp = ref.match_fixed_grid(ranges, poses)
        c_covs = np.asarray(c_covs).reshape(B, 9)
    else:
        port = po.PortKarto(po.default_cfg(), po.laser_struct(laser))
        port.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
        c_poses, c_resp, c_covs = np.zeros((B, 3)), np.zeros(B), np.zeros((B, 9))
        for i in range(B):
            m, c, r = port.match(ranges[i], poses[i])
            c_poses[i], c_resp[i], c_covs[i] = m, r, np.asarray(c).reshape(-1)
    assert (res["status"] == 0).all()
    d = res["pose"] - c_poses
    d[:, 2] = np.arctan2(np.sin(d[:, 2]), np.cos(d[:, 2]))
    assert np.abs(d).max() <= 1e-12
    assert np.abs(res["response"] - c_resp).max() == 0.0
    assert np.abs(res["covariance"].reshape(B, 9) - c_covs).max() <= 1e-12
    assert res["response"].mean() > 0.3  # the matches are real ones


@pytest.mark.parametrize("B", [192, 320, 1000])
def test_match_tail_on_and_off_give_identical_records(ctx, B):
    """Batches of 192 .. 2047 scans run coarse reduce + fine pass + fine reduce as ONE launch (k_match_tail, one block per
    scan); LSLAM_OPT_MATCH_TAIL = 0 keeps the three kernels.  Same device functions, so the result records must be
    byte-identical -- with search windows outside the grid (status != 0), lattices on a cell boundary (non-uniform: the
    reduce blocks compute those scans' numerators themselves), invalid readings, and without penalties."""
    rng = np.random.default_rng(900 + B)
    laser = synth.Laser()
    world = synth.arena(size=60.0, n_axis=14, n_rot=5, seed=901)
    wl = synth.make_match_workload(n_base=40, n_query=48, seed=902, laser=laser, world=world, query_spread=3.0)
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(laser))
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    idx = np.arange(B) % 48
    poses = synth.perturb(wl.truth_poses[idx], 0.3, math.radians(10.0), 903)
    poses[:6, :2] += 70.0                       # search window off the grid: Grid::GridIndex throws in the reference
    gi = gm.grid_info()
    res = 0.05
    # exactly on a cell boundary of the lattice: coordinates round differently along the lattice -> non-uniform steps
    poses[6:12, 0] = gi["offset"][0] + res * (np.round((poses[6:12, 0] - gi["offset"][0]) / res) + 0.5)
    ranges = wl.query_ranges[idx].copy()
    ranges[rng.random(ranges.shape) < 0.02] = np.inf
    for pen in (True, False):
        gm.set_option("match_tail", 1)
        ctx.profile(True); ctx.profile_reset()
        on = gm.match_batch(ranges, poses, doPenalize=pen)
        ctx.profile(False)
        names = set(ctx.profile_read())
        assert "match_tail" in names and "reduce_fine" not in names, names
        gm.set_option("match_tail", 0)
        ctx.profile(True); ctx.profile_reset()
        off = gm.match_batch(ranges, poses, doPenalize=pen)
        ctx.profile(False)
        assert "match_tail" not in set(ctx.profile_read())
        gm.set_option("match_tail", 1)
        assert on.tobytes() == off.tobytes(), pen
    assert (on["status"][:6] != 0).all() and (on["status"][12:] == 0).all()
