#!/usr/bin/env python3
"""Secondary measurements for the BASELINE configs that are parity cases rather than the headline
bench line (bench.py): config 2 (log-odds update), config 3 (single-scan MatchScan), config 5
(streaming match + map update).  Each prints one JSON line with the GPU number and the CPU oracle
timed beside it on the same inputs -- the oracle is used exactly as in bench.py's cpu_baseline leg: as the timed
CPU baseline and as the checker of the GPU results, never as a producer of them.
Usage: python tools/bench_extra.py [--scans N]"""
import argparse
import json
import math
import os
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import lslam  # noqa: E402,F401
from lslam_amd import api, synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def cfg2_map_update(ctx, n_poses, n=1000, cell=0.05):
    """lesson4 Hector-style log-odds update: 1081-beam scans into a 1000x1000 @ 0.05 m grid (BASELINE configs[1]); with
    n = 4000, cell = 0.025 the same scans into config 5's map, whose planes (64 MB each) no longer fit the L2."""
    laser = synth.Laser()
    off = (n * cell * 0.5, n * cell * 0.5)
    world = synth.arena(size=44.0, n_axis=12, n_rot=4, seed=3)
    rng = np.random.default_rng(3)
    poses = []
    while len(poses) < n_poses:
        x, y = rng.uniform(-4, 4, 2)
        if synth.point_is_free(world, x, y, 0.8):
            poses.append((x, y, rng.uniform(-math.pi, math.pi)))
    scans = [(synth.hector_points(synth.cast_scan(world, p, laser, 0.01, 0.01, rng), laser, 1.0 / cell, use_max=20.0),
              np.asarray(p, dtype=np.float32)) for p in poses]
    gmap = api.OccGridMap(ctx, n, n, cell, off)
    cmap = po.PortHector(n, n, cell, off)
    for m in (gmap, cmap):
        m.setUpdateOccupiedFactor(0.9)
    # device-resident points, asynchronous updates
    ptrs = []
    for pts, _ in scans:
        d = ctx.alloc(pts.nbytes)
        ctx.upload(d, pts)
        ptrs.append(d)
    for (pts, pose), d in list(zip(scans, ptrs))[:5]:
        gmap.updateByScan_dev(d, len(pts), (0.0, 0.0), pose)
    ctx.synchronize()
    # pass 1, profiled (HIP events around every kernel): kernel times; pass 2, plain: throughput
    gmap.reset()
    ctx.profile(True); ctx.profile_reset()
    for (pts, pose), d in zip(scans, ptrs):
        gmap.updateByScan_dev(d, len(pts), (0.0, 0.0), pose)
    ctx.synchronize()
    ctx.profile(False)
    prof = ctx.profile_read()
    gmap.reset()
    ctx.synchronize()
    t0 = time.perf_counter()
    for (pts, pose), d in zip(scans, ptrs):
        gmap.updateByScan_dev(d, len(pts), (0.0, 0.0), pose)
    ctx.synchronize()
    gpu_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    visits = 0
    for pts, pose in scans:
        cmap.updateByScan(pts, (0.0, 0.0), pose)
        visits += cmap.last_cell_visits()
    cpu_s = time.perf_counter() - t0
    same = gmap.logodds().tobytes() == cmap.logodds().tobytes()
    for d in ptrs:
        ctx.free(d)
    alg_bytes = visits * 16 + sum(len(p) for p, _ in scans) * 8
    k_ms = sum(v[1] for v in prof.values())
    # the batched entry: the same scans, 64 per call, points resident in HBM back to back
    allpts = np.ascontiguousarray(np.concatenate([p for p, _ in scans]), dtype=np.float32)
    counts = np.array([len(p) for p, _ in scans], dtype=np.int32)
    allposes = np.stack([q for _, q in scans])
    d_all = ctx.alloc(allpts.nbytes)
    ctx.upload(d_all, allpts)
    bmap = api.OccGridMap(ctx, n, n, cell, off)
    bmap.setUpdateOccupiedFactor(0.9)
    # points resident in HBM: the host never sees them, so it is told what the node knows -- no point beyond use_max = 20 m
    bmap.set_option("batch_radius_cells", int(np.ceil(20.0 / cell)) + 1)
    bmap.updateByScans_dev(d_all, counts[:64], (0.0, 0.0), allposes[:64])  # warm-up: allocates the planes
    bmap.reset()
    ctx.synchronize()
    ctx.profile(True); ctx.profile_reset()
    bmap.updateByScans_dev(d_all, counts, (0.0, 0.0), allposes)
    ctx.synchronize()
    ctx.profile(False)
    bprof = ctx.profile_read()
    bmap.reset()
    ctx.synchronize()
    t0 = time.perf_counter()
    bmap.updateByScans_dev(d_all, counts, (0.0, 0.0), allposes)
    ctx.synchronize()
    b_s = time.perf_counter() - t0
    b_same = bmap.logodds().tobytes() == cmap.logodds().tobytes()
    b_stats = bmap.batch_stats()
    ctx.free(d_all)
    bk_ms = sum(v[1] for v in bprof.values())
    return {"config": "cfg2 log-odds update, 1081-beam scans into %dx%d@%gm" % (n, n, cell), "scans": n_poses,
            "algorithmic_bytes_per_scan": round(alg_bytes / n_poses),
            "gpu_scans_per_s": round(n_poses / gpu_s, 1), "gpu_cell_updates_per_s": round(visits / gpu_s),
            "kernel_ms_total": round(k_ms, 3), "kernel_algorithmic_GBs": round(alg_bytes / (k_ms * 1e-3) / 1e9, 2),
            "batched": {"scans_per_call": 64, "gpu_scans_per_s": round(n_poses / b_s, 1),
                        "gpu_cell_updates_per_s": round(visits / b_s), "bit_exact": bool(b_same),
                        "scratch_bytes": b_stats["scratch_bytes"], "scratch_rounds": b_stats["rounds"], "window_misses": b_stats["window_misses"],
                        "kernel_ms": {k: round(v[1], 3) for k, v in sorted(bprof.items())},
                        "kernel_algorithmic_GBs": round(alg_bytes / (bk_ms * 1e-3) / 1e9, 2),
                        "whole_call_algorithmic_GBs": round(alg_bytes / b_s / 1e9, 2),
                        "frac_of_hbm_peak": round(alg_bytes / b_s / 1e9 / 8000.0, 4)},
            "cpu_port_scans_per_s": round(n_poses / cpu_s, 1), "cpu_cores": 1, "bit_exact": bool(same),
            "cell_visits_per_scan": round(visits / n_poses)}


def cfg3_single_scan(ctx, n_queries):
    """lesson6 MatchScan, one scan at a time: grid rebuilt from a 70-scan window per call."""
    wl = synth.make_match_workload(n_base=70, n_query=16, seed=4)
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(wl.laser))
    idx = np.arange(n_queries) % 16
    gm.MatchScan(wl.query_ranges[0], wl.query_poses[0], wl.base_ranges, wl.base_poses)
    t0 = time.perf_counter()
    out = [gm.MatchScan(wl.query_ranges[i], wl.query_poses[i], wl.base_ranges, wl.base_poses) for i in idx]
    gpu_s = time.perf_counter() - t0
    res = {"config": "cfg3 single-scan MatchScan (AddScans of 70 scans + search), host API incl. PCIe upload of the window",
           "queries": n_queries, "gpu_ms_per_match": round(1e3 * gpu_s / n_queries, 4)}
    if po.have_ref():
        ref = po.RefKarto(po.default_cfg(), po.laser_struct(wl.laser))
        k = min(n_queries, 200)
        t0 = time.perf_counter()
        cpu = [ref.match(wl.base_ranges, wl.base_poses, wl.query_ranges[i], wl.query_poses[i]) for i in idx[:k]]
        cpu_s = time.perf_counter() - t0
        res["cpu_reference_ms_per_match"] = round(1e3 * cpu_s / k, 4)
        res["max_pose_err"] = float(max(np.abs(out[i][1] - cpu[i][0]).max() for i in range(k)))
    return res


def cfg5_streaming(ctx, n_scans, ref_scans):
    """BASELINE config 5 as SURVEY.md 8(d) writes it: a 10 k-scan synthetic trajectory with CLOSED LOOPS inside a
    100 m x 100 m arena (0.25 m / <= 5 deg steps, drifting odometry), per-scan correlative match with the pose graph
    (LinkNearChains + TryCloseLoop) + incremental log-odds update of a 4000x4000 @ 0.025 m map."""
    import bench  # the pooled numpy ray caster

    laser = synth.Laser()
    path = synth.rings_trajectory(n_scans)
    world = synth.arena_around_path(path, size=100.0, n_axis=30, n_rot=10, seed=6)
    odom = synth.drifting_odometry(path, scale=1.01, sigma_xy=0.004, sigma_th=0.0015, seed=6)
    t0 = time.perf_counter()
    scans32 = bench.cast_scans(world, laser, path, 0, 6, max(1, min(32, os.cpu_count() or 1)))
    gen_s = time.perf_counter() - t0
    n, cell = 4000, 0.025
    off = (n * cell * 0.5, n * cell * 0.5)
    graph = dict(scan_buffer_size=70, scan_buffer_maximum_scan_distance=20.0, do_loop_closing=1,
                 link_scan_maximum_distance=1.5, loop_search_maximum_distance=3.0, loop_match_minimum_chain_size=10)
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(laser))
    fe = api.FrontEnd(gm, config=api.frontend_config(**graph))
    gmap = api.OccGridMap(ctx, n, n, cell, off)
    gmap.setUpdateOccupiedFactor(0.9)
    pts_all = [synth.hector_points(r, laser, 1.0 / cell, use_max=20.0) for r in scans32]
    r64 = [synth.ranges_to_f64(r) for r in scans32]
    # warm-up: kernels + plane allocation, on throw-away objects' first calls
    fe.Process(r64[0], odom[0]); fe.Process(r64[1], odom[1]); fe.reset()
    gmap.updateByScans(pts_all[:64], (0.0, 0.0), np.zeros((64, 3), np.float32)); gmap.reset()
    ctx.synchronize()
    t0 = time.perf_counter()
    g_poses, pend_pts, pend_pose, upd = [], [], [], []
    for r, o, pts in zip(r64, odom, pts_all):
        ok, pose, _, _ = fe.Process(r, o)
        g_poses.append(pose)
        if ok:  # the Karto matcher does not read this map: its updates are deferred and applied 64 scans at a time
            pend_pts.append(pts); pend_pose.append(pose.astype(np.float32)); upd.append(len(g_poses) - 1)
            if len(pend_pts) == 64:
                gmap.updateByScans(pend_pts, (0.0, 0.0), np.stack(pend_pose)); pend_pts, pend_pose = [], []
    if pend_pts:
        gmap.updateByScans(pend_pts, (0.0, 0.0), np.stack(pend_pose))
    ctx.synchronize()
    gpu_s = time.perf_counter() - t0
    st = fe.stats()
    g_poses = np.array(g_poses)
    res = {"config": "cfg5 streaming (SURVEY 8(d)): %d-scan closed-loop trajectory in a 100 m arena, per-scan correlative match "
                     "(70-scan running window) + pose graph (LinkNearChains, TryCloseLoop on the 81x81x21 loop matcher) + "
                     "log-odds update of a 4000x4000@0.025m map (batched, 64 scans per call)" % n_scans,
           "scans": n_scans, "gpu_scans_per_s": round(n_scans / gpu_s, 1), "gpu_us_per_scan": round(1e6 * gpu_s / n_scans, 1),
           "graph": st, "workload_gen_s": round(gen_s, 1),
           "max_pose_err_vs_truth_xy": float(np.hypot(*(g_poses[:, :2] - path[:, :2]).T).max()),
           "odometry_drift_at_end_xy": float(np.hypot(*(odom[-1, :2] - path[-1, :2])))}
    # map parity: the restated Hector update (pinned to the reference's headers) fed the same poses, every scan
    cmap = po.PortHector(n, n, cell, off)
    cmap.setUpdateOccupiedFactor(0.9)
    t0 = time.perf_counter()
    for i in upd:
        cmap.updateByScan(pts_all[i], (0.0, 0.0), g_poses[i].astype(np.float32))
    res["cpu_port_map_update_s"] = round(time.perf_counter() - t0, 1)
    res["map_bit_exact"] = bool(gmap.logodds().tobytes() == cmap.logodds().tobytes())
    res["map_cells_touched"] = int(np.count_nonzero(cmap.logodds()))
    # pose parity: the reference's own Mapper::Process (its graph, its loop matcher) beside it
    if po.have_ref() and ref_scans > 0:
        k = min(n_scans, ref_scans)
        ref = po.RefKarto(po.default_cfg(scan_buffer_size=70, scan_buffer_max_scan_distance=20.0, do_loop_closing=1,
                                         link_scan_maximum_distance=1.5, loop_search_maximum_distance=3.0,
                                         loop_match_minimum_chain_size=10), po.laser_struct(laser))
        t0 = time.perf_counter()
        worst = 0.0
        for i in range(k):
            _, pose = ref.process(r64[i], odom[i])
            worst = max(worst, float(np.abs(pose - g_poses[i]).max()))
        cpu_s = time.perf_counter() - t0
        res.update({"cpu_reference_scans_per_s": round(k / cpu_s, 1), "cpu_reference_scans": k, "cpu_cores": 1,
                    "max_pose_err_vs_reference": worst, "reference_graph_edges": ref.graph_stats()[1]})
    return res


def hector_front_end(ctx, n_scans):
    """Next-row #2: lesson4's loop matchData -> updateByScan (Gauss-Newton on a 3-level 1024^2 pyramid)."""
    laser = synth.Laser()
    n, cell, levels = 1024, 0.05, 3
    off = (n * cell * 0.5, n * cell * 0.5)
    world = synth.arena(size=40.0, n_axis=10, n_rot=4, seed=3)
    path = synth.trajectory(world, n_scans, step=0.05, seed=3, bounds=6.0)
    rng = np.random.default_rng(1)
    pts_all = [synth.hector_points(synth.cast_scan(world, t, laser, 0.01, 0.0, rng), laser, 1.0 / cell, use_max=20.0)
               for t in path]
    hints = [(t + np.array([0.05, -0.04, 0.02])).astype(np.float32) for t in path]

    def run(match, update):
        t0 = time.perf_counter()
        poses = []
        for k, (pts, hint) in enumerate(zip(pts_all, hints)):
            # like HectorSlamProcessor::update (HectorSlamProcessor.h:84-110): always match first -- on the empty map
            # of scan 0 that returns the start estimate unchanged, and it caches the containers of the levels above 0
            pose = match(pts, path[0].astype(np.float32) if k == 0 else hint)
            update(pts, pose)
            poses.append(pose)
        return time.perf_counter() - t0, np.array(poses)

    gpu = api.OccGridMap(ctx, n, n, cell, off, levels=levels)
    gpu.setUpdateOccupiedFactor(0.9)
    g_s, g_poses = run(lambda pts, h: gpu.matchData(h, pts)[0], lambda pts, pose: gpu.updateByScan(pts, (0.0, 0.0), pose))
    ctx.synchronize()
    cpu = po.PortHectorRep(cell, n, n, levels)
    cpu.setUpdateFactorOccupied(0.9)
    c_s, c_poses = run(lambda pts, h: cpu.matchData(pts, h)[0], lambda pts, pose: cpu.updateByScan(pts, (0.0, 0.0), pose))
    return {"config": "lesson4 front-end loop: matchData (Gauss-Newton, 3-level 1024^2 pyramid) + updateByScan per scan",
            "scans": n_scans, "gpu_scans_per_s": round(n_scans / g_s, 1), "cpu_port_scans_per_s": round(n_scans / c_s, 1),
            "cpu_cores": 1, "max_pose_diff_vs_port": float(np.abs(g_poses - c_poses).max()),
            "max_pose_err_vs_truth_xy": float(np.hypot(*(g_poses[:, :2] - path[:, :2]).T).max())}


def occgrid_from_scans(ctx, n_scans):
    """Next-row #1: lesson6's published map, OccupancyGrid::CreateFromScans over all scans (karto_slam.cc:507-512)."""
    laser = synth.Laser()
    world = synth.arena(size=60.0, n_axis=18, n_rot=6, seed=9)
    path = synth.trajectory(world, n_scans, step=0.2, seed=9, bounds=20.0)
    rng = np.random.default_rng(9)
    ranges = np.stack([synth.ranges_to_f64(synth.cast_scan(world, t, laser, 0.01, 0.01, rng)) for t in path])
    thr = 20.0
    lp = api.laser_params(laser, thr)
    g = api.OccupancyGrid.CreateFromScans(ctx, lp, ranges[:8], path[:8], 0.05)  # warm-up
    t0 = time.perf_counter()
    g = api.OccupancyGrid.CreateFromScans(ctx, lp, ranges, path, 0.05)
    got = g.data()
    gpu_s = time.perf_counter() - t0
    res = {"config": "Karto OccupancyGrid::CreateFromScans (hit/pass counters, TraceLine) over a whole trajectory, host API "
                     "incl. upload of the ranges and download of the grid", "scans": n_scans,
           "grid": list(got.shape), "gpu_ms": round(gpu_s * 1e3, 2)}
    if po.have_ref():
        ref = po.RefKarto(po.default_cfg(), po.laser_struct(laser, thr))
        t0 = time.perf_counter()
        exp, _ = ref.occgrid_from_scans(ranges, path, 0.05)
        res["cpu_reference_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
        res["cells_equal"] = bool(np.array_equal(got, exp))
    return res


def loop_closure(ctx, n_queries):
    """Next-row #3: loop-closure matcher instance, 10 m search space @0.05 m -> 101x101x21 candidates,
    coarse pass only, no penalty (TryCloseLoop, Mapper.cpp:991)."""
    laser = synth.Laser(range_max=30.0)
    world = synth.arena(size=40.0, n_axis=12, n_rot=4, seed=14)
    wl = synth.make_match_workload(n_base=30, n_query=8, seed=14, laser=laser, world=world, err_xy=2.0,
                                   err_th=math.radians(12.0), query_spread=1.0)
    cfg = api.baseline_config(search_size=10.0, range_threshold=12.0)
    gm = api.ScanMatcher(ctx, cfg, api.laser_params(laser, 12.0))
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    idx = np.arange(n_queries) % 8
    r, p = wl.query_ranges[idx], wl.query_poses[idx]
    gm.match_batch(r[:2], p[:2], doPenalize=False, doRefineMatch=False)
    t0 = time.perf_counter()
    one = [gm.match_batch(r[i:i + 1], p[i:i + 1], doPenalize=False, doRefineMatch=False)[0] for i in range(min(16, n_queries))]
    lat_s = (time.perf_counter() - t0) / min(16, n_queries)
    ctx.profile(True); ctx.profile_reset()
    gm.match_batch(r[:1], p[:1], doPenalize=False, doRefineMatch=False)
    ctx.profile(False)
    single_prof = {k: round(v[1], 4) for k, v in sorted(ctx.profile_read().items())}
    t0 = time.perf_counter()
    res = gm.match_batch(r, p, doPenalize=False, doRefineMatch=False)
    gpu_s = time.perf_counter() - t0
    out = {"config": "loop-closure coarse match: 101x101x21 candidates x 1081 beams (231 M byte gathers per match)",
           "queries": n_queries, "gpu_batched_ms_per_match": round(1e3 * gpu_s / n_queries, 4),
           "gpu_single_ms_per_match": round(1e3 * lat_s, 4), "single_match_kernel_ms": single_prof}
    port = po.PortKarto(po.default_cfg(search_size=10.0), po.laser_struct(laser, 12.0))
    port.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
    k = min(4, n_queries)
    t0 = time.perf_counter()
    cpu = [port.match(r[i], p[i], False, False) for i in range(k)]
    out["cpu_port_ms_per_match"] = round(1e3 * (time.perf_counter() - t0) / k, 2)
    out["max_pose_err"] = float(max(np.abs(res["pose"][i] - cpu[i][0]).max() for i in range(k)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--map-scans", type=int, default=1000)
    ap.add_argument("--single", type=int, default=200)
    ap.add_argument("--stream", type=int, default=10000)
    ap.add_argument("--stream-ref", type=int, default=1500, help="scans the reference's Mapper::Process is run on beside it")
    ap.add_argument("--loop", type=int, default=64)
    ap.add_argument("--hector", type=int, default=300)
    ap.add_argument("--occgrid", type=int, default=500)
    ap.add_argument("--map-size", type=int, default=1000, help="cfg2: map side in cells")
    ap.add_argument("--map-cell", type=float, default=0.05, help="cfg2: cell length [m]")
    ap.add_argument("--only", default="", help="comma list of: cfg2,cfg3,cfg5,loop,hector,occgrid")
    args = ap.parse_args()
    only = set(filter(None, args.only.split(",")))
    po.build("restate")
    ctx = api.Context(0)
    jobs = [("cfg2", lambda: cfg2_map_update(ctx, args.map_scans, args.map_size, args.map_cell)), ("cfg3", lambda: cfg3_single_scan(ctx, args.single)),
            ("cfg5", lambda: cfg5_streaming(ctx, args.stream, args.stream_ref)), ("loop", lambda: loop_closure(ctx, args.loop)),
            ("hector", lambda: hector_front_end(ctx, args.hector)), ("occgrid", lambda: occgrid_from_scans(ctx, args.occgrid))]
    for name, fn in jobs:
        if not only or name in only:
            print(json.dumps(fn()), flush=True)


if __name__ == "__main__":
    main()
