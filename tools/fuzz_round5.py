#!/usr/bin/env python3
"""Randomised checks of round 5's two structurally new paths, many seeds, on an MI355X:

  windows   lslam_map_update_batch(_dev) -- per-scan tile WINDOWS in a slot pool, rounds under a scratch budget -- against the
            CPU restatement's sequential updateByScan (H/map/OccGridMapBase.h:118-330): planes bit for bit.  Random map sizes
            and cell lengths, batch sizes 1..64, budgets 1..64 MB, poses incl. near and beyond the map's edges, empty
            containers, device-resident points with sufficient / absent radius hints.
  lookahead lslam_frontend_process_many against one lslam_frontend_process call per scan (Mapper::Process, Mapper.cpp:1999-2079):
            every pose bit for bit, same graph.  Random closed-loop trajectories (rings of random size and count -> different
            loop structure), random chunk sizes 1..300, random loop-search parameters, pools of 1 / 2 / 4 loop matchers.

  sums      lslam_matcher_debug_coarse_sums_batch -- the coarse numerators of every candidate of every scan of a batch through
            the launches a match of that size takes (beam slices + fp64 table cells / tiled planes + fp32-estimate cells) --
            against the restatement's GetResponse sums (Mapper.cpp:819-856): integers, bit for bit.  Random worlds, windows,
            batch sizes 8..1600, pose errors up to the search window and beyond, 1 % unreadable beams.

  dense     the same for the loop-closure matcher's lattices (Mapper.cpp:862-871, 976-1051: search space 4-10 m -> 41^2..101^2
            positions x 21 angles, k_resp_dense in its lone and batched forms), 2 scans per case against the restatement.

  specchain the streaming front-end with speculative anchor chains (round 6) against the plain path: every scan's anchors and pose.

usage: fuzz_round5.py [windows|lookahead|sums|dense|specchain|all] [N_CASES] [FIRST_SEED]   (prints one line per case, exit code 1 on a mismatch)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lslam  # noqa: F401  (import alias of the package directory)
from lslam_amd import api, synth
from oracle import pyoracle as po

what = sys.argv[1] if len(sys.argv) > 1 else "all"
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
bad = 0


def fuzz_windows(ctx, seed):
    import torch

    rng = np.random.default_rng(seed)
    n = int(rng.choice([400, 777, 1000, 1536, 2048, 3001]))
    cell = float(rng.choice([0.025, 0.05, 0.1]))
    K = int(rng.integers(1, 65))
    budget = int(rng.choice([1, 2, 4, 16, 64]))
    use_max = float(rng.uniform(3.0, 25.0))
    side = n * cell
    off = (side * rng.uniform(0.3, 0.7), side * rng.uniform(0.3, 0.7))
    world = synth.arena(size=max(20.0, min(60.0, side)), n_axis=8, n_rot=3, seed=seed)
    laser = synth.Laser()
    path = synth.trajectory(world, K, step=rng.uniform(0.1, 1.0), seed=seed, bounds=max(3.0, min(25.0, side * 0.5)))
    # some poses pushed towards / beyond the map's edge: windows clip, beams leave the map
    shift = np.zeros((K, 3))
    far = rng.random(K) < 0.3
    shift[far, :2] = rng.uniform(-0.6, 0.6, size=(int(far.sum()), 2)) * side
    path = path + shift
    scans = []
    for p in path:
        r = synth.cast_scan(world, p, laser, 0.01, 0.01, rng)
        pts = synth.hector_points(r, laser, 1.0 / cell, use_max=use_max)
        if rng.random() < 0.08:
            pts = pts[:0]
        scans.append((pts, p.astype(np.float32)))
    cpu = po.PortHector(n, n, cell, off)
    gpu = api.OccGridMap(ctx, n, n, cell, off, levels=1)
    dev = api.OccGridMap(ctx, n, n, cell, off, levels=1)
    for m in (cpu, gpu, dev):
        m.setUpdateFreeFactor(0.4)
        m.setUpdateOccupiedFactor(0.9)
    gpu.set_option("batch_scratch_mb", budget)
    dev.set_option("batch_scratch_mb", budget)
    hinted = rng.random() < 0.6
    if hinted:
        dev.set_option("batch_radius_cells", int(use_max / cell) + 2)
    origo = (float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.5, 0.5)))
    reps = int(rng.integers(1, 3))
    for _ in range(reps):
        for pts, pose in scans:
            cpu.updateByScan(pts, origo, pose)
        gpu.updateByScans([p for p, _ in scans], origo, np.stack([q for _, q in scans]))
        allp = np.ascontiguousarray(np.concatenate([p for p, _ in scans]) if sum(len(p) for p, _ in scans) else np.zeros((0, 2)),
                                    dtype=np.float32)
        counts = np.array([len(p) for p, _ in scans], np.int32)
        d_pts = torch.from_numpy(allp if len(allp) else np.zeros((1, 2), np.float32)).to("cuda:0")
        torch.cuda.synchronize()
        dev.updateByScans_dev(d_pts.data_ptr(), counts, origo, np.stack([q for _, q in scans]))
    want = cpu.logodds().tobytes()
    a, b = gpu.logodds().tobytes() == want, dev.logodds().tobytes() == want
    st, sd = gpu.batch_stats(), dev.batch_stats()
    ok = a and b and st["window_misses"] == 0 and sd["window_misses"] == 0
    print("windows   seed %d: %dx%d @ %.3f, K %2d, budget %2d MB, reach %4.1f m, %s: host rounds %d scratch %.1f MB, dev rounds %d%s, "
          "cells touched %d -> %s" % (seed, n, n, cell, K, budget, use_max, "x%d" % reps, st["rounds"], st["scratch_bytes"] / 2**20,
                                      sd["rounds"], " (hint)" if hinted else "", int(np.count_nonzero(cpu.logodds())),
                                      "equal" if ok else "MISMATCH host=%s dev=%s misses %d/%d" % (a, b, st["window_misses"], sd["window_misses"])),
          flush=True)
    gpu.close(); dev.close()
    return ok


def fuzz_lookahead(ctx, seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(400, 1400))
    laser = synth.Laser()
    # a different loop structure per seed: two to four small concentric rings, one to three laps each
    outer = rng.uniform(8.0, 16.0)
    halves = tuple(outer - 2.5 * k for k in range(int(rng.integers(2, 5))))
    path = synth.rings_trajectory(n, half_sizes=halves, laps=int(rng.integers(1, 4)), radius=2.0, change_len=6.0)
    world = synth.arena_around_path(path, size=2.0 * (outer + 8.0), n_axis=int(rng.integers(8, 20)), n_rot=int(rng.integers(3, 8)),
                                    seed=seed)
    odom = synth.drifting_odometry(path, scale=rng.uniform(0.995, 1.02), sigma_xy=rng.uniform(0.001, 0.006),
                                   sigma_th=rng.uniform(0.0005, 0.002), seed=seed)
    import bench

    scans32 = bench.cast_scans(world, laser, path, 0, seed, max(1, min(16, os.cpu_count() or 1)))
    r64 = np.stack([synth.ranges_to_f64(r) for r in scans32])
    if rng.random() < 0.3:  # some unreadable scans: all-NaN readings
        r64[rng.integers(0, n, size=3)] = np.nan
    cfg = api.frontend_config(scan_buffer_size=int(rng.choice([20, 70])), scan_buffer_maximum_scan_distance=float(rng.choice([10.0, 20.0])),
                              do_loop_closing=1, link_scan_maximum_distance=float(rng.uniform(1.0, 2.0)),
                              loop_search_maximum_distance=float(rng.uniform(2.0, 4.5)),
                              loop_match_minimum_chain_size=int(rng.integers(5, 14)))
    pool = int(rng.choice([1, 2, 4]))
    os.environ["LSLAM_FE_LOOP_POOL"] = str(pool)
    out = {}
    for mode in ("process", "many"):
        gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(laser))
        fe = api.FrontEnd(gm, config=cfg)
        poses, oks = np.zeros((n, 3)), np.zeros(n, bool)
        if mode == "process":
            for i in range(n):
                oks[i], poses[i], _, _ = fe.Process(r64[i], odom[i])
        else:
            i0 = 0
            while i0 < n:
                i1 = min(n, i0 + int(rng.integers(1, 300)))
                oks[i0:i1], poses[i0:i1], _, _ = fe.ProcessMany(r64[i0:i1], odom[i0:i1])
                i0 = i1
        ctx.synchronize()
        final = np.stack([fe.scan_pose(i) for i in range(fe.num_scans())]) if fe.num_scans() else np.zeros((0, 3))
        out[mode] = (poses, oks, fe.stats(), final, fe.lookahead_stats())
        fe.close(); gm.close()
    a, b = out["process"], out["many"]
    same = (np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1]) and a[2] == b[2] and np.array_equal(a[3], b[3], equal_nan=True))
    print("lookahead seed %d: %4d scans, pool %d, graph %s, look-ahead %s -> %s" %
          (seed, n, pool, {k: a[2][k] for k in ("edges", "loop_coarse_matches", "loops_closed")}, b[4], "equal" if same else "MISMATCH"),
          flush=True)
    return same


def fuzz_specchain(ctx, seed):
    """Speculative anchor chains (round 6, k_anchor_chain / anchor_spec_block) against the plain path (LSLAM_FE_SPEC_CHAIN=0):
    every scan's anchor row and every pose bit for bit.  Random trajectories, worlds, shares of +inf / NaN readings, windows."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(150, 500))
    laser = synth.Laser()
    outer = rng.uniform(8.0, 30.0)
    halves = tuple(outer - 2.5 * k for k in range(int(rng.integers(1, 4))))
    path = synth.rings_trajectory(n, half_sizes=halves, laps=int(rng.integers(1, 3)), radius=2.0, change_len=6.0)
    world = synth.arena_around_path(path, size=2.0 * (outer + rng.uniform(4.0, 40.0)), n_axis=int(rng.integers(4, 30)),
                                    n_rot=int(rng.integers(0, 10)), seed=seed)
    odom = synth.drifting_odometry(path, scale=rng.uniform(0.99, 1.03), sigma_xy=rng.uniform(0.001, 0.01),
                                   sigma_th=rng.uniform(0.0005, 0.004), seed=seed)
    dropout = float(rng.choice([0.0, 0.01, 0.1, 0.5]))
    nan_share = float(rng.choice([0.0, 0.0, 0.02]))
    r64 = []
    for q in path:
        r = synth.ranges_to_f64(synth.cast_scan(world, q, laser, float(rng.choice([0.0, 0.01])), dropout, rng))
        if nan_share:
            r[rng.random(r.shape) < nan_share] = np.nan
        r64.append(r)
    loop = rng.random() < 0.4
    cfg = api.frontend_config(scan_buffer_size=int(rng.choice([10, 70])), scan_buffer_maximum_scan_distance=float(rng.choice([10.0, 20.0])),
                              do_loop_closing=int(loop), link_scan_maximum_distance=1.5, loop_search_maximum_distance=3.0,
                              loop_match_minimum_chain_size=10)
    out = {}
    for on in (0, 1):
        os.environ["LSLAM_FE_SPEC_CHAIN"] = str(on)
        gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(laser))
        fe = api.FrontEnd(gm, config=cfg)
        del os.environ["LSLAM_FE_SPEC_CHAIN"]
        for r, o in zip(r64, odom):
            fe.Process(r, o)
        k = fe.num_scans()
        out[on] = ([fe.anchor_row(i) for i in range(k)], np.stack([fe.scan_pose(i) for i in range(k)]), fe.spec_chain_stats())
        fe.close(); gm.close()
    same = (len(out[0][0]) == len(out[1][0]) and all(np.array_equal(a, b) for a, b in zip(out[0][0], out[1][0])) and
            out[0][1].tobytes() == out[1][1].tobytes())
    print("specchain seed %d: %4d scans, +inf share %.2f, NaN share %.2f, loop closing %d, chains %s -> %s" %
          (seed, n, dropout, nan_share, int(loop), out[1][2], "equal" if same else "MISMATCH"), flush=True)
    return same


def fuzz_sums(ctx, seed):
    import math

    rng = np.random.default_rng(seed)
    laser = synth.Laser()
    world = synth.arena(size=rng.uniform(25, 90), n_axis=int(rng.integers(6, 30)), n_rot=int(rng.integers(2, 10)), seed=seed)
    wl = synth.make_match_workload(n_base=int(rng.integers(5, 70)), n_query=32, seed=seed + 1, laser=laser, world=world,
                                   query_spread=rng.uniform(0.3, 4.0))
    S = int(rng.choice([8, 40, 100, 128, 256, 600, 1600]))
    idx = np.arange(S) % 32
    poses = synth.perturb(wl.truth_poses[idx], rng.uniform(0.02, 0.8), math.radians(rng.uniform(1, 30)), seed + 2)
    ranges = wl.query_ranges[idx].copy()
    ranges[rng.random(ranges.shape) < 0.01] = np.inf
    ranges[rng.random(ranges.shape) < 0.003] = np.nan
    port = po.PortKarto(po.default_cfg(), po.laser_struct(laser))
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(laser))
    port.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    if os.environ.get("LSLAM_FUZZ_STEP_KERNEL"):  # the same batches through the scan-resident workgroup kernel (3 / 4 waves)
        gm.set_option("step_kernel", int(os.environ["LSLAM_FUZZ_STEP_KERNEL"]))
    got = gm.coarse_sums_batch(ranges, poses)
    got_f, centers = gm.fine_sums_batch(ranges, poses)  # fine pass (Mapper.cpp:276-281) around the device's own coarse mean
    check = np.unique(rng.integers(0, S, size=min(S, 48)))
    n_bad, n_bad_f, total, total_f = 0, 0, 0, 0
    for q in check:
        _, _, _, st, sums_cpu = port.correlate_scan(ranges[q], poses[q], poses[q], 0.5, 0.1, 0.349, 0.0349, True, False, want_sums=True)
        if st != 0:
            continue
        total += int(sums_cpu.sum() > 0)
        n_bad += not np.array_equal(got[q], sums_cpu)
        if np.isnan(centers[q]).any():
            continue
        _, _, _, st, fine_cpu = port.correlate_scan(ranges[q], poses[q], centers[q], 0.05, 0.05, 0.5 * 0.0349, 0.00349, True, True,
                                                    want_sums=True)
        if st != 0:
            continue
        total_f += int(fine_cpu.sum() > 0)
        n_bad_f += not np.array_equal(got_f[q], fine_cpu)
    ok = n_bad == 0 and n_bad_f == 0
    print("sums      seed %d: batch %3d, %d scans checked: coarse %d non-zero lattices x 2541 candidates, fine %d x 99 -> %s" %
          (seed, S, len(check), total, total_f, "equal" if ok else "MISMATCH in %d coarse / %d fine lattices" % (n_bad, n_bad_f)), flush=True)
    gm.close()
    return ok


def fuzz_dense(ctx, seed):
    import math

    rng = np.random.default_rng(seed)
    laser = synth.Laser(range_max=30.0)
    search = float(rng.choice([4.0, 6.0, 8.0, 10.0]))
    kw = dict(search_size=search, resolution=0.05, smear_deviation=0.03)
    port = po.PortKarto(po.default_cfg(**kw), po.laser_struct(laser, 12.0, (0, 0, 0)))
    gm = api.ScanMatcher(ctx, api.baseline_config(range_threshold=12.0, **kw), api.laser_params(laser, 12.0, (0, 0, 0)))
    world = synth.arena(size=rng.uniform(30, 60), n_axis=int(rng.integers(6, 20)), n_rot=int(rng.integers(2, 8)), seed=seed)
    wl = synth.make_match_workload(n_base=int(rng.integers(8, 40)), n_query=8, seed=seed + 1, laser=laser, world=world,
                                   err_xy=rng.uniform(0.2, 0.45 * search), err_th=math.radians(rng.uniform(2, 18)), query_spread=1.0)
    port.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    S = int(rng.choice([1, 3, 8, 24, 72]))
    idx = np.arange(S) % 8
    poses = wl.query_poses[idx] + rng.uniform(-0.2, 0.2, size=(S, 3)) * np.array([1.0, 1.0, 0.1])
    ranges = wl.query_ranges[idx].copy()
    ranges[rng.random(ranges.shape) < 0.01] = np.nan
    got = gm.coarse_sums_batch(ranges, poses) if S > 1 else gm.coarse_sums(ranges[0], poses[0])[None]
    off = 0.5 * (round(search / 0.05)) * 0.05
    n_bad, n_chk = 0, 0
    for q in np.unique(rng.integers(0, S, size=2)):
        _, _, _, st, sums = port.correlate_scan(ranges[q], poses[q], poses[q], off, 0.1, 0.349, 0.0349, False, False, want_sums=True)
        if st != 0:
            continue
        n_chk += 1
        n_bad += not np.array_equal(got[q], sums)
    ok = n_bad == 0
    print("dense     seed %d: search %4.1f m (%d^2 x 21), batch %2d, %d scans checked -> %s" %
          (seed, search, got.shape[1], S, n_chk, "equal" if ok else "MISMATCH in %d" % n_bad), flush=True)
    gm.close()
    return ok


import torch  # (initialised before the library's own HIP context, like bench.py and the tests do)

torch.cuda.init()
ctx = api.Context(0)
t0 = time.time()
for k in range(n_cases):
    if what in ("windows", "all"):
        bad += not fuzz_windows(ctx, seed0 + k)
    if what in ("lookahead", "all"):
        bad += not fuzz_lookahead(ctx, seed0 + k)
    if what in ("sums", "all"):
        bad += not fuzz_sums(ctx, seed0 + k)
    if what in ("dense", "all"):
        bad += not fuzz_dense(ctx, seed0 + k)
    if what in ("specchain", "all"):
        bad += not fuzz_specchain(ctx, seed0 + k)
print("%d case(s) differ; %.0f s" % (bad, time.time() - t0))
sys.exit(1 if bad else 0)
