#!/usr/bin/env python3
"""Config 5 at full size, front-end only (pose graph on, no map update): Mapper::Process scan by scan against
lslam_frontend_process_many (one scan of look-ahead), same scans, same graph."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lslam  # noqa
from lslam_amd import api, synth
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 64
with_map = len(sys.argv) > 3 and sys.argv[3] == "map"  # + the 4000 x 4000 log-odds map updated after every call (own context)
laser = synth.Laser()
path = synth.rings_trajectory(n)
world = synth.arena_around_path(path, size=100.0, n_axis=30, n_rot=10, seed=6)
odom = synth.drifting_odometry(path, scale=1.01, sigma_xy=0.004, sigma_th=0.0015, seed=6)
scans32 = bench.cast_scans(world, laser, path, 0, 6, max(1, min(32, os.cpu_count() or 1)))
r64 = np.stack([synth.ranges_to_f64(r) for r in scans32])
ctx = api.Context(0)
cfg = api.frontend_config(scan_buffer_size=70, scan_buffer_maximum_scan_distance=20.0, do_loop_closing=1,
                          link_scan_maximum_distance=1.5, loop_search_maximum_distance=3.0, loop_match_minimum_chain_size=10)
res = {}
gmap = pts_all = None
if with_map:
    ctx_map = api.Context(0)
    gmap = api.OccGridMap(ctx_map, 4000, 4000, 0.025, (50.0, 50.0))
    gmap.setUpdateOccupiedFactor(0.9)
    pts_all = [synth.hector_points(r, laser, 40.0, use_max=20.0) for r in scans32]
for mode in ("process", "process_many", "process", "process_many"):
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(laser))
    fe = api.FrontEnd(gm, config=cfg)
    fe.Process(r64[0], odom[0]); fe.Process(r64[1], odom[1]); fe.reset()
    ctx.synchronize()
    t0 = time.perf_counter()
    poses = np.zeros((n, 3))
    if mode == "process":
        for i in range(n):
            _, poses[i], _, _ = fe.Process(r64[i], odom[i])
    else:
        for i0 in range(0, n, chunk):
            i1 = min(n, i0 + chunk)
            ok, poses[i0:i1], _, _ = fe.ProcessMany(r64[i0:i1], odom[i0:i1])
            if gmap is not None:
                idx = [i for i in range(i0, i1) if ok[i - i0]]
                if idx:
                    gmap.updateByScans([pts_all[i] for i in idx], (0.0, 0.0), poses[idx].astype(np.float32))
    ctx.synchronize()
    if gmap is not None:
        ctx_map.synchronize(); gmap.reset()
    dt = time.perf_counter() - t0
    st = fe.stats()
    print("%-13s %8.1f scans/s  %6.1f us/scan  edges %d loops %d  look-ahead %s" % (mode, n / dt, 1e6 * dt / n, st["edges"], st["loops_closed"],
          fe.lookahead_stats()), flush=True)
    if mode in res:
        pass
    res.setdefault(mode, poses)
    fe.close(); gm.close()
print("max pose difference process vs process_many: %.3g" % np.abs(res["process"] - res["process_many"]).max())
