#!/usr/bin/env python3
"""Per-kernel breakdown of the streamed per-scan chain (lslam_frontend_process, pose graph off): one pass with HIP
events around every kernel (ctx.profile), one plain pass for the throughput.

  python tools/chain_profile.py [--scans 600]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lslam  # noqa: E402,F401
from lslam_amd import api, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=600)
    ap.add_argument("--loop", action="store_true", help="pose graph with loop closing on (cfg 5's settings)")
    ap.add_argument("--lone-kernel", type=int, default=-1, help="LSLAM_OPT_LONE_KERNEL (0, 4, 8, 16); -1 = the library's default")
    args = ap.parse_args()
    import bench
    laser = synth.Laser()
    path = synth.rings_trajectory(args.scans)
    world = synth.arena_around_path(path, size=100.0, n_axis=30, n_rot=10, seed=6)
    odom = synth.drifting_odometry(path, scale=1.01, sigma_xy=0.004, sigma_th=0.0015, seed=6)
    scans32 = bench.cast_scans(world, laser, path, 0, 6, max(1, min(32, os.cpu_count() or 1)))
    r64 = [synth.ranges_to_f64(r) for r in scans32]
    ctx = api.Context(0)
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(laser))
    if args.lone_kernel >= 0:
        gm.set_option("lone_kernel", args.lone_kernel)
    if args.loop:
        fe = api.FrontEnd(gm, config=api.frontend_config(scan_buffer_size=70, scan_buffer_maximum_scan_distance=20.0, do_loop_closing=1,
                                                         link_scan_maximum_distance=1.5, loop_search_maximum_distance=3.0,
                                                         loop_match_minimum_chain_size=10))
    else:
        fe = api.FrontEnd(gm)  # defaults: graph bookkeeping on, loop closing off
    for r, o in zip(r64[:80], odom[:80]):
        fe.Process(r, o)
    fe.reset()
    ctx.synchronize()
    t0 = time.perf_counter()
    for r, o in zip(r64, odom):
        fe.Process(r, o)
    ctx.synchronize()
    plain = time.perf_counter() - t0
    fe.reset()
    ctx.profile(True); ctx.profile_reset()
    for r, o in zip(r64, odom):
        fe.Process(r, o)
    ctx.synchronize()
    ctx.profile(False)
    prof = ctx.profile_read()
    n = args.scans
    out = {"scans": n, "us_per_scan": round(1e6 * plain / n, 1),
           "kernel_us_per_scan": {k: round(1e3 * v[1] / n, 2) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])},
           "launches_per_scan": round(sum(v[0] for v in prof.values()) / n, 1)}
    out["kernel_us_total"] = round(sum(out["kernel_us_per_scan"].values()), 1)
    out["launches"] = {k: v[0] for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}
    out["graph"] = fe.stats()
    out["lone_kernel"] = args.lone_kernel
    out["lone_kernel_launches"] = gm.lone_kernel_launches
    import hashlib
    out["poses_sha256"] = hashlib.sha256(np.stack([fe.scan_pose(i) for i in range(fe.num_scans())]).tobytes()).hexdigest()[:16]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
