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
# LSLAM_SOAK_DEPTH=2..4 runs them as PIPELINED sub-batches (LSLAM_OPT_PIPELINE_DEPTH; lslam_matcher_match_batch splits the batch).
# Seeds 45 and 46 are the pinned regression of round 5's extended soak (34 x 2304 matches against the reference): the two
# batches in which the PENALISED response of a refined match came out one ulp off (1 of 2304 each; tools/soak_probe.py) --
# they keep the <= 2 ulp / <= 0.2 % clause below honest: without them no case of the suite exercises it.
_CASES = ([(0, 256), (1, 256), (2, 256), (3, 2304), (45, 2304), (46, 2304)] +
          [(int(x), 2304) for x in os.environ.get("LSLAM_SOAK_SEEDS", "").split(",") if x])


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
    if os.environ.get("LSLAM_SOAK_DEPTH"):
        gm.set_option("pipeline_depth", int(os.environ["LSLAM_SOAK_DEPTH"]))
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
    # The response numerators are integers and equal bit for bit (the un-penalised check below).  The PENALISED response of a
    # refined match can differ in its last bit: the fine pass is centred on the coarse mean, whose heading goes through
    # atan2 / sin / cos -- the device libm's differ from glibc's in the last bit for ~1.5 % of the headings (4.4e-16 rad), and
    # once in a few thousand matches the angle penalty (Mapper.cpp:408-411) of the best fine candidate then rounds the other
    # way (tools/soak_probe.py 45: 1 of 2304).  One ulp of a value in (0.5, 1] is 1.1e-16.
    dr = np.abs(res["response"] - c_resp)
    assert dr.max() <= 2.3e-16 and np.count_nonzero(dr) <= max(1, B // 500), (dr.max(), np.count_nonzero(dr))
    if po.have_ref():  # without the odometry penalty the response IS the integer ratio: exact, refined or not
        res0 = gm.match_batch(ranges, poses, doPenalize=False)
        _, _, _, c_resp0 = ref.match_fixed_grid(ranges, poses, do_penalize=False)
        assert np.abs(res0["response"] - c_resp0).max() == 0.0
    assert np.abs(res["covariance"].reshape(B, 9) - c_covs).max() <= 1e-12
    assert res["response"].mean() > 0.3  # the matches are real ones
