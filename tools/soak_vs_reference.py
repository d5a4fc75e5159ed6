import sys, math, numpy as np
sys.path.insert(0, '.')
import lslam
from lslam_amd import api, synth
from oracle import pyoracle as po
ctx = api.Context(0)
worst = 0.0; worst_r = 0.0; n = 0
for seed in range(6):
    rng = np.random.default_rng(100 + seed)
    laser = synth.Laser()
    world = synth.arena(size=rng.uniform(30, 90), n_axis=int(rng.integers(6, 30)), n_rot=int(rng.integers(2, 10)), seed=200 + seed)
    wl = synth.make_match_workload(n_base=int(rng.integers(10, 70)), n_query=32, seed=300 + seed, laser=laser, world=world, query_spread=rng.uniform(0.5, 4.0))
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(laser))
    ref = po.RefKarto(po.default_cfg(), po.laser_struct(laser)) if po.have_ref() else None
    port = po.PortKarto(po.default_cfg(), po.laser_struct(laser))
    o = ref or port
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    o.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
    B = 256
    idx = np.arange(B) % 32
    poses = synth.perturb(wl.truth_poses[idx], rng.uniform(0.05, 0.45), math.radians(rng.uniform(2, 18)), 400 + seed)
    ranges = wl.query_ranges[idx].copy()
    bad = rng.random(ranges.shape) < 0.01
    ranges[bad] = np.inf
    res = gm.match_batch(ranges, poses)
    if ref:
        sec, c_poses, c_covs, c_resp = ref.match_fixed_grid(ranges, poses)
    else:
        c_poses = np.zeros((B, 3)); c_resp = np.zeros(B); c_covs = np.zeros((B, 9))
        for i in range(B):
            m, c, r = port.match(ranges[i], poses[i]); c_poses[i] = m; c_resp[i] = r; c_covs[i] = np.asarray(c).reshape(-1)
    d = res["pose"] - c_poses
    d[:, 2] = np.arctan2(np.sin(d[:, 2]), np.cos(d[:, 2]))
    worst = max(worst, np.abs(d).max()); worst_r = max(worst_r, np.abs(res["response"] - c_resp).max())
    cerr = np.abs(res["covariance"].reshape(B, 9) - np.asarray(c_covs).reshape(B, 9)).max()
    n += B
    print(f"seed {seed}: statuses ok {int((res['status']==0).sum())}/{B} max pose diff {np.abs(d).max():.2e} resp diff {np.abs(res['response']-c_resp).max():.2e} cov diff {cerr:.2e} mean resp {res['response'].mean():.3f}")
print("TOTAL", n, "worst pose", worst, "worst resp", worst_r, "kind", "reference" if po.have_ref() else "port")
