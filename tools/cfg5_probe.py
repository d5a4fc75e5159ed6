#!/usr/bin/env python3
"""Per-1000-scan wall time of the cfg 5 front-end run (pose graph on, no map update): shows what grows with the number of
resident scans (the host-side graph walks) against what does not (the device chain)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lslam  # noqa
from lslam_amd import api, synth
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
laser = synth.Laser()
path = synth.rings_trajectory(n)
world = synth.arena_around_path(path, size=100.0, n_axis=30, n_rot=10, seed=6)
odom = synth.drifting_odometry(path, scale=1.01, sigma_xy=0.004, sigma_th=0.0015, seed=6)
scans32 = bench.cast_scans(world, laser, path, 0, 6, max(1, min(32, os.cpu_count() or 1)))
r64 = [synth.ranges_to_f64(r) for r in scans32]
ctx = api.Context(0)
gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(laser))
fe = api.FrontEnd(gm, config=api.frontend_config(scan_buffer_size=70, scan_buffer_maximum_scan_distance=20.0, do_loop_closing=1,
                                                 link_scan_maximum_distance=1.5, loop_search_maximum_distance=3.0,
                                                 loop_match_minimum_chain_size=10))
fe.Process(r64[0], odom[0]); fe.Process(r64[1], odom[1]); fe.reset()
t0 = time.perf_counter(); last = t0; prev = fe.stats()
for i in range(n):
    fe.Process(r64[i], odom[i])
    if i % 1000 == 999:
        now = time.perf_counter(); st = fe.stats()
        print("scans %5d-%5d: %6.1f us/scan   loop coarse %5d  chain %4d" % (i - 999, i, 1e6 * (now - last) / 1000,
              st["loop_coarse_matches"] - prev["loop_coarse_matches"], st["chain_matches"] - prev["chain_matches"]), flush=True)
        last, prev = now, st
print("total %.1f us/scan" % (1e6 * (time.perf_counter() - t0) / n))
