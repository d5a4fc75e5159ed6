#!/usr/bin/env python3
"""Where a soak case's 1-ulp response difference comes from: the batch of tests/test_soak_gpu.py for one seed, matched with and
without the odometry penalty, GPU against the reference itself.  usage: soak_probe.py SEED [BATCH]"""
import math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lslam  # noqa
from lslam_amd import api, synth
from oracle import pyoracle as po

seed = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 2304
rng = np.random.default_rng(100 + seed)
laser = synth.Laser()
world = synth.arena(size=rng.uniform(30, 90), n_axis=int(rng.integers(6, 30)), n_rot=int(rng.integers(2, 10)), seed=200 + seed)
wl = synth.make_match_workload(n_base=int(rng.integers(10, 70)), n_query=32, seed=300 + seed, laser=laser, world=world,
                               query_spread=rng.uniform(0.5, 4.0))
ctx = api.Context(0)
gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(laser))
gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
idx = np.arange(B) % 32
poses = synth.perturb(wl.truth_poses[idx], rng.uniform(0.05, 0.45), math.radians(rng.uniform(2, 18)), 400 + seed)
ranges = wl.query_ranges[idx].copy()
ranges[rng.random(ranges.shape) < 0.01] = np.inf
ref = po.RefKarto(po.default_cfg(), po.laser_struct(laser))
ref.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
for pen in (True, False):
    for refine in (True, False):
        res = gm.match_batch(ranges, poses, doPenalize=pen, doRefineMatch=refine)
        _, c_poses, c_covs, c_resp = ref.match_fixed_grid(ranges, poses, do_penalize=pen, do_refine=refine)
        dr = np.abs(res["response"] - c_resp)
        dp = res["pose"] - c_poses
        bad = np.nonzero(dr)[0]
        print("penalty %-5s refine %-5s: responses differing %d of %d (max %.3g); poses differing in the last bits: %d (max %.3g)" %
              (pen, refine, len(bad), B, dr.max(), int(np.count_nonzero(np.abs(dp).max(axis=1))), np.abs(dp).max()))
        for i in bad[:4]:
            print("   scan %d: response gpu %.17g ref %.17g; pose diff %s; heading gpu %.17g ref %.17g" %
                  (i, res["response"][i], c_resp[i], dp[i], res["pose"][i, 2], c_poses[i, 2]))
