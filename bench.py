#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: scan-matches/sec (1081-beam scans vs a 2000x2000-class grid).

One "step" = one pass of the hot path over one batch: the coarse+fine correlative search
(karto::ScanMatcher::MatchScan's search part, Mapper.cpp:227-290) of `--batch` independent,
DISTINCT synthetic 1081-beam scans (SURVEY.md §8(d) cfg 4: poses uniform over the free space around
the window, each with its own odometry error) against one shared, HBM-resident 2005x2005 uint8
correlation grid (BASELINE.json configs[3], the configuration the metric is quoted on).  float32
ranges, fp64 poses and the 112-byte result records are resident in HBM before the timed region.

Multi-GPU: `python bench.py --gpus N` starts N ranks itself (re-exec through torch.distributed.run;
under an existing torchrun it just joins).  The scans are independent units, so the path shards with NO
data-path collective.  Default `--scaling strong` is SURVEY §8(e)'s partition: the `--batch` (4096)
scans are split [r*B/W, (r+1)*B/W) over the ranks (shard.shard_range); `--scaling weak` gives every
rank its own `--batch` scans.  Every rank builds the same grid from the same seeded base scans
(cheaper than the 4 MB broadcast, which exists as --broadcast-grid).  RCCL carries the timing barrier,
the max-over-ranks reduction and the all_gather of the 112-byte result records ("poses out"), which is
timed separately and reported as `gather_ms`.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed on the stream it
runs on) and `cpu_baseline` (the reference's own CorrelateScan, oracle/_ref, on this box's host
cores; falls back to the plain-C port where _ref is absent).

At N=1 the same run also measures the other north-star configs (`secondary`, ~10 s of GPU work): config 2
(log-odds update, one scan per call and the batched entry, each against the HBM peak), config 3 (ms per complete
MatchScan) and a closed-loop slice of config 5 (streaming front-end with its pose graph on the 4000x4000@0.025 m
map), every one with the reference's CPU figure beside it and checked against it; plus a `sustained` leg of the
headline step (>= 10 s by default: clocks ramped, visible to a 5-s busy sampler), config 5 AT ITS STATED SIZE (10 000 scans,
checked in the run against the record of the reference in tests/golden/karto_cfg5_golden.npz), the SURVEY 8(f) rows
(loop-closure lattice, lesson4 loop, CreateFromScans) and the DROP-IN legs: the reference's own Mapper::Process /
HectorSlamProcessor (oracle/_ref_gpu, test infrastructure compiled from /root/reference) driving the HIP path through the
C ABI -- labelled integration legs, never the headline.  The CPU legs run in worker processes forked BEFORE the GPU is touched; they
are released only after the headline's timed region, so the headline number is measured on a quiet host.  A compact
copy of the secondary numbers is also nested under `roofline.secondary` / `cpu_baseline.secondary`.
"""
from __future__ import annotations

import argparse
import json
import os
import pathlib
import socket
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_CU, SIMD_PER_CU = 256, 4  # MI355X: 256 CUs x 4 SIMDs
PEAK_CLOCK_HZ = 2.4e9       # peak engine clock
# a wave64 VALU instruction occupies its SIMD for >= 2 cycles (MI355X_MICROARCH.md), so the chip cannot
# issue more than SIMDs * clk / 2 of them per second -- the bound the hot kernel is priced against
VALU_ISSUE_PEAK = N_CU * SIMD_PER_CU * PEAK_CLOCK_HZ / 2.0
L2_PEAK_GBS = 34500.0       # aggregate L2 bandwidth (MI355X_MICROARCH.md: 4 MiB per XCD, ~34.5 TB/s)
N_BEAMS = 1081


def algorithmic_bytes(nx, ny, na_c, fx, fy, na_f, n):
    """SURVEY.md §8(d) / BASELINE.md §4 accounting, per scan-match."""
    coarse = nx * ny * na_c * n + na_c * n * 4           # grid gathers + lookup table of the coarse pass
    fine = (fx * fy * na_f + na_f) * n + na_f * n * 4    # fine pass + angular-covariance responses
    io = n * 4 + 112                                      # float32 ranges in, result record out
    return coarse, coarse + fine + io


# ------------------------------------------------------------------------------------------------------
# workload (pure numpy, generated BEFORE CUDA / torch.distributed are touched so that forking is safe)
# ------------------------------------------------------------------------------------------------------
def _cast_chunk(job):
    from lslam_amd import synth

    world, laser, poses, first, seed = job
    out = np.empty((len(poses), laser.n_ranges), dtype=np.float32)
    for i, p in enumerate(poses):
        rng = np.random.default_rng([seed, first + i])  # per-scan stream: independent of the sharding
        out[i] = synth.cast_scan(world, p, laser, 0.01, 0.01, rng)
    return out


def cast_scans(world, laser, poses, first, seed, procs):
    """float32 ranges [len(poses), n] -- numpy ray casting, fanned out over host processes."""
    import multiprocessing as mp

    if len(poses) == 0:
        return np.empty((0, laser.n_ranges), dtype=np.float32)
    procs = max(1, min(procs, len(poses)))
    bounds = np.linspace(0, len(poses), procs * 4 + 1).astype(int)
    jobs = [(world, laser, poses[a:b], first + a, seed) for a, b in zip(bounds[:-1], bounds[1:]) if b > a]
    if procs == 1:
        return np.concatenate([_cast_chunk(j) for j in jobs])
    with mp.get_context("fork").Pool(procs) as pool:
        return np.concatenate(pool.map(_cast_chunk, jobs))


def query_poses(world, anchor, n, spread, seed):
    """n truth poses uniform over the free space of a disc around the window end (cfg 4)."""
    from lslam_amd import synth

    rng = np.random.default_rng(seed)
    out = np.empty((n, 3))
    for i in range(n):
        while True:
            r = spread * np.sqrt(rng.random())
            a = rng.uniform(-np.pi, np.pi)
            x, y = anchor[0] + r * np.cos(a), anchor[1] + r * np.sin(a)
            if synth.point_is_free(world, x, y, 0.8):
                break
        out[i] = (x, y, anchor[2] + rng.uniform(-0.3, 0.3))
    return out


# ------------------------------------------------------------------------------------------------------
# CPU legs: worker processes forked before CUDA / HIP are touched.  Inputs travel by fork (module global
# _JOB), results by pickle.  Every task waits for GO, which the main process sets after the headline's
# timed region.  The reference library (oracle/_ref) chats on std::cout, also at exit: workers' fd 1 goes
# to /dev/null.
# ------------------------------------------------------------------------------------------------------
GO = None
GO_MULTI = None  # the all-core leg starts only after the single-core headline baseline has finished
_JOB = {}


def _quiet_worker():
    dn = os.open(os.devnull, os.O_WRONLY)
    os.dup2(dn, 1)


def _cpu_cfg4_share(_):
    """One host core: the reference's own CorrelateScan (oracle/_ref) on 200 of the sample's scans."""
    GO_MULTI.wait()
    from oracle import pyoracle as po

    wl, q_r, q_p = _JOB["wl"], _JOB["q_r"][:200], _JOB["q_p"][:200]
    ref = po.RefKarto(po.default_cfg(), po.laser_struct(wl.laser))
    ref.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
    sec, _, _, _ = ref.match_fixed_grid(q_r, q_p)
    return sec * len(q_r), len(q_r)


def _cpu_cfg4_single(_):
    """cpu_baseline of the headline: coarse+fine CorrelateScan of the sample against the shared grid, one core."""
    GO.wait()
    from oracle import pyoracle as po

    wl, q_r, q_p = _JOB["wl"], _JOB["q_r"], _JOB["q_p"]
    sample = len(q_r)
    if po.have_ref():
        ref = po.RefKarto(po.default_cfg(), po.laser_struct(wl.laser))
        ref.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
        sec, c_poses, _, c_resp = ref.match_fixed_grid(q_r, q_p)
        kind = "reference"
    else:
        po.build("restate")
        port = po.PortKarto(po.default_cfg(), po.laser_struct(wl.laser))
        port.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
        t1 = time.perf_counter()
        c_poses, c_resp = np.zeros((sample, 3)), np.zeros(sample)
        for i in range(sample):
            m, _, r = port.match(q_r[i], q_p[i])
            c_poses[i], c_resp[i] = m, r
        sec = (time.perf_counter() - t1) / sample
        kind = "port"
    return {"sec_per_match": sec, "poses": c_poses, "resp": c_resp, "kind": kind}


def _sha(a):
    import hashlib

    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _cpu_cfg2(_):
    """config 2 on one host core: lesson4's own OccGridMapBase::updateByScan (oracle/_ref/libhector_ref.so) over the
    same 1081-beam scans; the restated oracle beside it counts the Bresenham cell visits (algorithmic bytes)."""
    GO.wait()
    from oracle import pyoracle as po

    d = _JOB["cfg2"]
    n, cell, off = d["n"], d["cell"], d["off"]
    po.build("restate")
    port = po.PortHector(n, n, cell, off)
    port.setUpdateOccupiedFactor(0.9)
    visits = 0
    t0 = time.perf_counter()
    for pts, pose in zip(d["pts"], d["poses"]):
        port.updateByScan(pts, (0.0, 0.0), pose)
        visits += port.last_cell_visits()
    port_s = time.perf_counter() - t0
    out = {"visits": int(visits), "port_scans_per_s": len(d["pts"]) / port_s, "map_sha": _sha(port.logodds()), "kind": "port",
           "scans_per_s": len(d["pts"]) / port_s}
    if po.have_ref_hector():
        ref = po.RefHector(n, n, cell, off)
        ref.setUpdateOccupiedFactor(0.9)
        t0 = time.perf_counter()
        for pts, pose in zip(d["pts"], d["poses"]):
            ref.updateByScan(pts, (0.0, 0.0), pose)
        ref_s = time.perf_counter() - t0
        out.update({"kind": "reference", "scans_per_s": len(d["pts"]) / ref_s, "ref_equals_port": _sha(ref.logodds()) == out["map_sha"]})
    return out


def _cpu_cfg3(_):
    """config 3 on one host core: the reference's complete MatchScan (grid rebuilt from the 70-scan window per call)."""
    GO.wait()
    from oracle import pyoracle as po

    wl, idx = _JOB["cfg3"]["wl"], _JOB["cfg3"]["idx"]
    if not po.have_ref():
        return None
    ref = po.RefKarto(po.default_cfg(), po.laser_struct(wl.laser))
    t0 = time.perf_counter()
    out = [ref.match(wl.base_ranges, wl.base_poses, wl.query_ranges[i], wl.query_poses[i]) for i in idx]
    sec = (time.perf_counter() - t0) / len(idx)
    return {"ms_per_match": 1e3 * sec, "poses": np.stack([o[0] for o in out]), "resp": np.array([o[2] for o in out])}


CFG5_GRAPH = dict(scan_buffer_size=70, do_loop_closing=1, link_scan_maximum_distance=1.5, loop_search_maximum_distance=3.0,
                  loop_match_minimum_chain_size=10)


def _cpu_cfg5(_):
    """config 5 slice on one host core: the reference's own karto::Mapper::Process with its pose graph."""
    GO.wait()
    from oracle import pyoracle as po

    d = _JOB["cfg5"]
    if not po.have_ref():
        return None
    ref = po.RefKarto(po.default_cfg(scan_buffer_max_scan_distance=20.0, **CFG5_GRAPH), po.laser_struct(d["laser"]))
    poses = np.zeros((len(d["r64"]), 3))
    t0 = time.perf_counter()
    for i, (r, o) in enumerate(zip(d["r64"], d["odom"])):
        _, poses[i] = ref.process(r, o)
    sec = time.perf_counter() - t0
    v, e = ref.graph_stats()
    return {"scans_per_s": len(poses) / sec, "seconds": sec, "poses": poses, "vertices": int(v), "edges": int(e)}


# ------------------------------------------------------------------------------------------------------
# secondary configs: workloads (numpy, before the fork) and GPU legs
# ------------------------------------------------------------------------------------------------------
def secondary_workloads(n_map_scans, n_single, n_stream, procs):
    """cfg 2 / 3 / 5 inputs as SURVEY.md 8(d) describes them (seeds fixed)."""
    import math

    from lslam_amd import synth

    laser = synth.Laser()
    out = {}
    # cfg 2: 1081-beam scans from distinct poses into a 1000x1000 @ 0.05 m map, p_free 0.4 / p_occ 0.9 (seed 3)
    n, cell = 1000, 0.05
    world = synth.arena(size=44.0, n_axis=12, n_rot=4, seed=3)
    rng = np.random.default_rng(3)
    poses = []
    while len(poses) < n_map_scans:
        x, y = rng.uniform(-4, 4, 2)
        if synth.point_is_free(world, x, y, 0.8):
            poses.append((x, y, rng.uniform(-math.pi, math.pi)))
    poses = np.asarray(poses)
    r32 = cast_scans(world, laser, poses, 0, 3, procs)
    out["cfg2"] = {"n": n, "cell": cell, "off": (n * cell * 0.5, n * cell * 0.5),
                   "pts": [synth.hector_points(r, laser, 1.0 / cell, use_max=20.0) for r in r32],
                   "poses": poses.astype(np.float32)}
    # cfg 3: 70-scan running window (seed 4), query = next pose + odometry error
    out["cfg3"] = {"wl": synth.make_match_workload(n_base=70, n_query=16, seed=4), "idx": np.arange(n_single) % 16}
    # cfg 5 slice: closed loops inside the 100 m arena (one ring of the full trajectory's family, driven > 2 laps so
    # that near chains link and loops close within the slice), 0.25 m steps, drifting odometry (seed 6)
    path = synth.rings_trajectory(n_stream, half_sizes=(9.0,), laps=3)
    world5 = synth.arena_around_path(path, size=100.0, n_axis=30, n_rot=10, seed=6)
    odom = synth.drifting_odometry(path, scale=1.01, sigma_xy=0.004, sigma_th=0.0015, seed=6)
    s32 = cast_scans(world5, laser, path, 0, 6, procs)
    out["cfg5"] = {"laser": laser, "path": path, "odom": odom, "r64": [synth.ranges_to_f64(r) for r in s32],
                   "pts": [synth.hector_points(r, laser, 1.0 / 0.025, use_max=20.0) for r in s32]}
    return out


# mapper_params.yaml as karto_slam.cc applies it (setParamDistanceVariancePenalty / ...AngleVariancePenalty square their
# argument, Mapper.cpp:1919-1927; `use_scan_range` becomes the laser's range threshold, karto_slam.cc:98,395)
INDOOR = {"search_size": 0.3, "resolution": 0.01, "smear_deviation": 0.03, "use_response_expansion": 1,
          "distance_variance_penalty": 0.5 ** 2, "angle_variance_penalty": 0.1 ** 2, "range_threshold": 12.0}


def batch_leg_workloads(n_scans, procs):
    """Two more batched-match workloads, each `n_scans` DISTINCT scans against one shared grid:
    `indoor`  the configuration the reference SHIPS (lesson6/config/mapper_params.yaml: 0.01 m cells, 0.3 m search space,
              12 m range threshold, response expansion on -> 2445 x 2445 grid, 13 x 13 smear kernel, 16 x 16 x 21 coarse
              lattice) in a cluttered 24 m world;
    `dense`   the headline's configuration in a world of 140 obstacles per 60 m x 60 m (median range 3 m instead of 24 m):
              what the exact zero-row pruning is worth when the grid is NOT mostly empty."""
    import math

    from lslam_amd import synth

    out = {}
    laser_in = synth.Laser(range_max=30.0)
    world_in = synth.arena(size=24.0, n_axis=8, n_rot=3, seed=12)
    wl = synth.make_match_workload(n_base=40, n_query=1, seed=12, laser=laser_in, world=world_in, err_xy=0.08,
                                   err_th=math.radians(6.0), query_spread=0.5)
    truth = query_poses(world_in, wl.center_pose, n_scans, 1.5, seed=121)
    out["indoor"] = {"wl": wl, "q_r32": cast_scans(world_in, laser_in, truth, 0, 123, procs),
                     "q_p": synth.perturb(truth, 0.08, math.radians(6.0), 122), "cfg": dict(INDOOR),
                     "what": "lesson6/config/mapper_params.yaml: res 0.01, search 0.3, range 12, smear 0.03, response expansion on "
                             "(2445x2445 grid, 13x13 smear, coarse 16x16x21 + fine 3x3x11), %d distinct scans in a cluttered 24 m "
                             "world, 40-scan window" % n_scans}
    world_d = synth.arena(size=60.0, n_axis=100, n_rot=40, seed=21)
    wld = synth.make_match_workload(n_base=70, n_query=1, seed=21, query_spread=3.0, world=world_d)
    truth = query_poses(world_d, wld.center_pose, n_scans, 3.0, seed=211)
    out["dense"] = {"wl": wld, "q_r32": cast_scans(world_d, synth.Laser(), truth, 0, 213, procs),
                    "q_p": synth.perturb(truth, 0.3, np.deg2rad(10.0), 212), "cfg": {},
                    "what": "the headline's configuration (BASELINE configs[3]) in a DENSE world: 140 obstacles on 60 m x 60 m "
                            "(median range ~3 m; the headline's arena: 32 on 80 m x 80 m, ~24 m), %d distinct scans, 70-scan window"
                            % n_scans}
    return out


def _cpu_batch_share(job):
    """One host core: the reference's CorrelateScan (+ MatchScan's response expansion) on every `cores`-th scan of a batch leg."""
    GO_MULTI.wait()
    from oracle import pyoracle as po

    name, i, cores = job
    d = _JOB[name]
    wl, cfg = d["wl"], dict(d["cfg"])
    thr = cfg.pop("range_threshold", 49.5)
    from lslam_amd import synth

    ref = po.RefKarto(po.default_cfg(**cfg), po.laser_struct(wl.laser, thr))
    ref.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
    idx = np.arange(i, len(d["q_p"]), cores)
    sec, poses, _, resp = ref.match_fixed_grid(synth.ranges_to_f64(d["q_r32"][idx]), d["q_p"][idx])
    return {"idx": idx, "busy_s": sec * len(idx), "poses": poses, "resp": resp}


def gpu_batch_leg(ctx, api, d, steps=60):
    """A batch leg on the GPU: plain steps of all the job's scans against its grid (float32 ranges, poses and records resident
    in HBM), the exact pruning's statistics, the step with pruning off, the records of one step."""
    import torch

    wl, cfg = d["wl"], dict(d["cfg"])
    thr = cfg.pop("range_threshold", 49.5)
    gm = api.ScanMatcher(ctx, api.baseline_config(range_threshold=thr, **cfg), api.laser_params(wl.laser, thr))
    gi = gm.grid_info()
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    dev = torch.device("cuda", ctx.device)
    r = torch.from_numpy(np.ascontiguousarray(d["q_r32"])).to(dev)
    p = torch.from_numpy(np.ascontiguousarray(d["q_p"])).to(dev)
    n = r.shape[0]
    out = torch.empty((n, 112), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def step():
        gm.match_batch_dev(n, r.data_ptr(), r.shape[1], p.data_ptr(), out.data_ptr(), dtype="f32")

    def timed(k):
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        ctx.synchronize()
        return (time.perf_counter() - t0) / k

    step(); step()
    t = timed(steps)
    rec = out.cpu().numpy().view(api.RESULT_DTYPE).reshape(-1).copy()
    ctx.profile(True); ctx.profile_only(None); ctx.profile_reset()
    step()
    prof = ctx.profile_read()
    ctx.profile(False)
    gm.set_option("collect_stats", 1)
    step(); ctx.synchronize()
    st = gm.read_stats()
    gm.set_option("collect_stats", 0)
    gm.set_option("row_occupancy", 0)
    step()
    t_off = timed(max(4, steps // 4))
    rec_off = out.cpu().numpy().view(api.RESULT_DTYPE).reshape(-1).copy()
    gm.set_option("row_occupancy", 1)
    gm.close()
    return {"n": n, "grid": [gi["width"], gi["height"]], "kernel_size": gi["kernel_size"], "s_per_step": t, "s_per_step_pruning_off": t_off,
            "records": rec, "pruning_off_identical": bool(rec.tobytes() == rec_off.tobytes()),
            "pruned_row_fraction": round(1.0 - st["rows_live"] / max(st["rows_in_range"], 1), 5),
            "beam_angles_with_no_live_row": round(1.0 - st["beam_angles_queued"] / max(st["beam_angles"], 1), 5),
            "kernel_ms": {k: round(v[1], 4) for k, v in sorted(prof.items())}}


def gpu_cfg2(ctx, api, d):
    """config 2 on the GPU: one scan per call (points resident in HBM, asynchronous updates) and the batched entry."""
    n, cell, off = d["n"], d["cell"], d["off"]
    scans = list(zip(d["pts"], d["poses"]))
    gmap = api.OccGridMap(ctx, n, n, cell, off)
    gmap.setUpdateOccupiedFactor(0.9)
    ptrs = []
    for pts, _ in scans:
        p = ctx.alloc(max(pts.nbytes, 8))
        ctx.upload(p, pts)
        ptrs.append(p)
    for (pts, pose), p in list(zip(scans, ptrs))[:5]:
        gmap.updateByScan_dev(p, len(pts), (0.0, 0.0), pose)
    gmap.reset()
    ctx.synchronize()
    t0 = time.perf_counter()
    for (pts, pose), p in zip(scans, ptrs):
        gmap.updateByScan_dev(p, len(pts), (0.0, 0.0), pose)
    ctx.synchronize()
    single_s = time.perf_counter() - t0
    single_sha = _sha(gmap.logodds())
    gmap.reset()
    ctx.profile(True); ctx.profile_only(None); ctx.profile_reset()
    for (pts, pose), p in zip(scans, ptrs):
        gmap.updateByScan_dev(p, len(pts), (0.0, 0.0), pose)
    ctx.synchronize()
    ctx.profile(False)
    single_prof = ctx.profile_read()
    for p in ptrs:
        ctx.free(p)
    gmap.close()
    allpts = np.ascontiguousarray(np.concatenate([p for p, _ in scans]), dtype=np.float32)
    counts = np.array([len(p) for p, _ in scans], dtype=np.int32)
    d_all = ctx.alloc(allpts.nbytes)
    ctx.upload(d_all, allpts)
    bmap = api.OccGridMap(ctx, n, n, cell, off)
    bmap.setUpdateOccupiedFactor(0.9)
    # points resident in HBM: the host never sees them, so it is told what the node knows -- no point beyond use_max = 20 m
    bmap.set_option("batch_radius_cells", int(np.ceil(20.0 / cell)) + 1)
    bmap.updateByScans_dev(d_all, counts[:64], (0.0, 0.0), d["poses"][:64])  # allocates the byte planes
    bmap.reset()
    ctx.synchronize()
    t0 = time.perf_counter()
    bmap.updateByScans_dev(d_all, counts, (0.0, 0.0), d["poses"])
    ctx.synchronize()
    batch_s = time.perf_counter() - t0
    batch_sha = _sha(bmap.logodds())
    bmap.reset()
    ctx.profile(True); ctx.profile_reset()
    bmap.updateByScans_dev(d_all, counts, (0.0, 0.0), d["poses"])
    ctx.synchronize()
    ctx.profile(False)
    batch_prof = ctx.profile_read()
    batch_stats = bmap.batch_stats()
    ctx.free(d_all)
    bmap.close()
    return {"single_s": single_s, "single_sha": single_sha, "single_prof": single_prof, "batch_s": batch_s,
            "batch_sha": batch_sha, "batch_prof": batch_prof, "points": int(counts.sum()), "batch_stats": batch_stats}


def gpu_cfg3(ctx, api, d):
    wl, idx = d["wl"], d["idx"]
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(wl.laser))
    gm.MatchScan(wl.query_ranges[0], wl.query_poses[0], wl.base_ranges, wl.base_poses)
    t0 = time.perf_counter()
    out = [gm.MatchScan(wl.query_ranges[i], wl.query_poses[i], wl.base_ranges, wl.base_poses) for i in idx]
    sec = (time.perf_counter() - t0) / len(idx)
    # the same calls with the window named by id in a device-side scan cache (seam B1, round 4): 24 bytes of pose per base
    # scan instead of its 8.6 KB of readings; records must be identical
    cache = api.ScanCache(ctx, api.laser_params(wl.laser))
    for k in range(len(wl.base_ranges)):
        cache.put(k, wl.base_ranges[k])
    ids = np.arange(len(wl.base_ranges))
    cache.MatchScan(gm, ids, wl.base_poses, wl.query_poses[0], query_id=-1, query_ranges=wl.query_ranges[0])
    t0 = time.perf_counter()
    out_c = [cache.MatchScan(gm, ids, wl.base_poses, wl.query_poses[i], query_id=-1, query_ranges=wl.query_ranges[i]) for i in idx]
    sec_c = (time.perf_counter() - t0) / len(idx)
    same = all(a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) for a, b in zip(out, out_c))
    cache.close()
    gm.close()
    return {"ms_per_match": 1e3 * sec, "poses": np.stack([o[1] for o in out]), "resp": np.array([o[0] for o in out]),
            "ms_per_match_cached": 1e3 * sec_c, "cached_identical": bool(same)}


def gpu_cfg5(ctx, api, d):
    """Streaming front-end: Mapper::Process with its pose graph per scan + log-odds update of the 4000x4000@0.025 m map
    (the Karto matcher does not read that map, so its updates are applied 64 accepted scans at a time)."""
    n, cell = 4000, 0.025
    off = (n * cell * 0.5, n * cell * 0.5)
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(d["laser"]))
    fe = api.FrontEnd(gm, config=api.frontend_config(scan_buffer_maximum_scan_distance=20.0, **CFG5_GRAPH))
    gmap = api.OccGridMap(ctx, n, n, cell, off)
    gmap.setUpdateOccupiedFactor(0.9)
    r64, odom, pts_all = d["r64"], d["odom"], d["pts"]
    fe.Process(r64[0], odom[0]); fe.Process(r64[1], odom[1]); fe.reset()
    gmap.updateByScans(pts_all[:64], (0.0, 0.0), np.zeros((64, 3), np.float32)); gmap.reset()
    ctx.synchronize()

    def run():
        poses, pend_pts, pend_pose, upd = [], [], [], []
        t0 = time.perf_counter()
        for i, (r, o, pts) in enumerate(zip(r64, odom, pts_all)):
            ok, pose, _, _ = fe.Process(r, o)
            poses.append(pose)
            if ok:
                pend_pts.append(pts); pend_pose.append(pose.astype(np.float32)); upd.append(i)
                if len(pend_pts) == 64:
                    gmap.updateByScans(pend_pts, (0.0, 0.0), np.stack(pend_pose)); pend_pts, pend_pose = [], []
        if pend_pts:
            gmap.updateByScans(pend_pts, (0.0, 0.0), np.stack(pend_pose))
        ctx.synchronize()
        return time.perf_counter() - t0, np.array(poses), upd

    sec, poses, upd = run()
    st = fe.stats()
    logodds = gmap.logodds()
    # per-kernel view of the same run (HIP events around every launch; slower, so not the throughput figure)
    fe.reset(); gmap.reset()
    ctx.profile(True); ctx.profile_only(None); ctx.profile_reset()
    run()
    ctx.profile(False)
    prof = ctx.profile_read()
    gmap.close(); fe.close(); gm.close()
    return {"seconds": sec, "poses": poses, "updated": upd, "stats": st, "logodds": logodds, "prof": prof, "n": n, "cell": cell, "off": off}


# ------------------------------------------------------------------------------------------------------
# drop-in legs: the reference's OWN orchestrators (compiled from /root/reference by `make -C oracle ref_gpu`, test
# infrastructure) driving the HIP path through the C ABI -- karto::Mapper::Process with MatchScan substituted at link time
# (integration/karto_scan_matcher_gpu.cpp), HectorSlamProcessor with HectorMapRepGpu as its mapRep
# (integration/hector_map_rep_gpu.hpp).  Labelled integration legs: never the headline.
# ------------------------------------------------------------------------------------------------------
def dropin_karto(d, repeat_reset=False):
    """The cfg 5 slice through the reference's Mapper::Process + GPU MatchScan.  Returns scans/s, the integration layer's
    counters and a host profile (share of the wall time spent inside MatchScan / inside the device call)."""
    from oracle import pyoracle as po

    if not po.have_ref_gpu():
        return {"error": "oracle/_ref_gpu not built (needs /root/reference at build time)"}
    ref = po.RefKarto(po.default_cfg(scan_buffer_max_scan_distance=20.0, **CFG5_GRAPH), po.laser_struct(d["laser"]), gpu=True)
    r64, odom = d["r64"], d["odom"]
    for i in range(3):  # device matcher + cache creation, first-use allocations
        ref.process(r64[i], odom[i])
    ref.reset()
    s0 = ref.gpu_stats()
    poses = np.zeros((len(r64), 3))
    t0 = time.perf_counter()
    for i, (r, o) in enumerate(zip(r64, odom)):
        _, poses[i] = ref.process(r, o)
    sec = time.perf_counter() - t0
    s1 = ref.gpu_stats()
    v, e = ref.graph_stats()
    calls = s1["match_calls"] - s0["match_calls"]
    in_match = (s1["ns_in_match_scan"] - s0["ns_in_match_scan"]) * 1e-9
    in_dev = (s1["ns_in_device_call"] - s0["ns_in_device_call"]) * 1e-9
    out = {"scans": len(r64), "seconds": sec, "scans_per_s": len(r64) / sec, "poses": poses, "vertices": int(v), "edges": int(e),
           "device_match_calls": calls, "cached_calls": s1["cached_calls"] - s0["cached_calls"],
           "scans_uploaded": s1["scans_uploaded"] - s0["scans_uploaded"], "refreshes_before_a_match": s1["refreshes"] - s0["refreshes"],
           "us_per_match_call": 1e6 * in_dev / max(calls, 1),
           "host_profile": {"wall_s": sec, "inside_MatchScan_s": in_match, "inside_device_call_s": in_dev,
                            "reference_host_code_s": sec - in_match,
                            "share_reference_host_code": (sec - in_match) / sec, "share_device_call": in_dev / sec}}
    ref.close()
    return out


def dropin_hector(n_scans=300, update_every_scan=False):
    """lesson4's loop through the reference's HectorSlamProcessor::update (H/slam_main/HectorSlamProcessor.h:84-110) with
    mapRep = HectorMapRepGpu, beside the same processor on its own MapRepMultiMap (1 host core)."""
    from lslam_amd import synth
    from oracle import pyoracle as po

    if not (po.have_ref_gpu() and po.have_ref_hector()):
        return {"error": "oracle/_ref_gpu not built"}
    laser = synth.Laser()
    n, cell, levels = 1024, 0.05, 3
    world = synth.arena(size=40.0, n_axis=10, n_rot=4, seed=5)
    # a gentle arc (4 cm / 0.23 deg per scan) from the map origin: the processor feeds its own estimate back as the next
    # start (hector_slam.cc:200-204), which the reference's matcher can only follow on a path without sharp turns
    k = np.arange(n_scans)
    path = np.stack([0.04 * k, 0.015 * k, 0.004 * k], axis=1)
    rng = np.random.default_rng(1)
    pts_all = [synth.hector_points(synth.cast_scan(world, t, laser, 0.01, 0.0, rng), laser, 1.0 / cell, use_max=20.0) for t in path]

    def run(gpu):
        proc = po.RefHectorProcessor(cell, n, n, (0.5, 0.5), levels, p_free=0.4, p_occ=0.9, gpu=gpu)
        if update_every_scan:
            proc.L.href_proc_set_update_thresholds(proc.h, 0.0, 0.0)
        est = np.zeros(3, np.float32)  # the pose chain is fed back as the next start estimate (hector_slam.cc:200-204)
        poses, upd = [], []
        t0 = time.perf_counter()
        for pts in pts_all:
            upd.append(bool(proc.update(pts, est)))
            est, _ = proc.last_pose()
            poses.append(est.copy())
        sec = time.perf_counter() - t0
        lo = proc.logodds(0)
        proc.close()
        return sec, np.array(poses), np.array(upd), lo

    g_s, g_p, g_u, g_lo = run(True)
    c_s, c_p, c_u, c_lo = run(False)
    # HectorSlamProcessor updates its map when the pose moved 0.4 m / 0.9 rad since the last update (HectorSlamProcessor.h:
    # 100-108): a 1e-5 pose difference can move such a decision by one scan, after which the two runs match against
    # DIFFERENT maps and are two valid trajectories of the same algorithm, not a parity measurement any more.  Parity is
    # what happens up to the first differing decision.
    differ = np.nonzero(g_u != c_u)[0]
    k = int(differ[0]) if len(differ) else n_scans
    # both against the TRUE motion (the path starts at the map origin, the processor's frame)
    err_g = float(np.hypot(*(g_p[:, :2] - path[:, :2]).T).max())
    err_c = float(np.hypot(*(c_p[:, :2] - path[:, :2]).T).max())
    return {"scans": n_scans, "gpu_scans_per_s": n_scans / g_s, "cpu_reference_scans_per_s": n_scans / c_s, "cpu_cores": 1,
            "map_updates_gpu": int(g_u.sum()), "map_updates_reference": int(c_u.sum()),
            "scans_until_first_differing_update_decision": k,
            "max_pose_diff_vs_reference_until_then": float(np.abs(g_p[:k] - c_p[:k]).max()) if k else 0.0,
            "max_pose_diff_vs_reference": float(np.abs(g_p - c_p).max()),
            "max_err_vs_truth_xy_gpu": err_g, "max_err_vs_truth_xy_reference": err_c,
            "map_cells_differing": int(np.count_nonzero(g_lo != c_lo)), "map_cells_touched": int(np.count_nonzero(c_lo)),
            "update_every_scan": bool(update_every_scan)}


def gpu_cfg5_full(ctx, api, n_scans=10000):
    """BASELINE configs[4] AT ITS STATED SIZE: 10 000-scan closed-loop trajectory, pose graph on, 4000x4000@0.025 m map,
    checked inside the run against the record of the reference's own Mapper::Process over the same scans
    (tests/golden/karto_cfg5_golden.npz: 39 min of one host core, tests/golden/make_cfg5_golden.py)."""
    import hashlib
    import importlib.util

    from lslam_amd import synth

    G = ROOT / "tests" / "golden"
    d = np.load(G / "karto_cfg5_golden.npz", allow_pickle=False)
    spec = importlib.util.spec_from_file_location("make_cfg5_golden", G / "make_cfg5_golden.py")
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    n = min(int(d["n"]), n_scans)
    t0 = time.perf_counter()
    laser, path, odom, scans32 = mk.workload(n=int(d["n"]))
    gen_s = time.perf_counter() - t0
    same_input = hashlib.sha256(scans32.tobytes()).hexdigest() == str(d["ranges_sha256"])
    size, cell = 4000, 0.025
    off = (size * cell * 0.5, size * cell * 0.5)
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(laser))
    fe = api.FrontEnd(gm, config=api.frontend_config(scan_buffer_maximum_scan_distance=20.0, **CFG5_GRAPH))
    # the map on a context (= stream) of its own: the Karto matcher never reads this map, so its batched updates need not
    # queue up in front of the next scan's match on the front-end's stream
    ctx_map = api.Context(ctx.device)
    gmap = api.OccGridMap(ctx_map, size, size, cell, off)
    gmap.setUpdateOccupiedFactor(0.9)
    r64 = [synth.ranges_to_f64(r) for r in scans32[:n]]
    pts_all = [synth.hector_points(r, laser, 1.0 / cell, use_max=20.0) for r in scans32[:n]]
    fe.Process(r64[0], odom[0]); fe.Process(r64[1], odom[1]); fe.reset()
    gmap.updateByScans(pts_all[:64], (0.0, 0.0), np.zeros((64, 3), np.float32)); gmap.reset()
    ctx.synchronize()
    # (a) scan by scan, Mapper::Process as the node calls it -- timed first, then everything is reset
    poses_1, pend_pts, pend_pose = np.zeros((n, 3)), [], []
    t0 = time.perf_counter()
    for i in range(n):
        ok, poses_1[i], _, _ = fe.Process(r64[i], odom[i])
        if ok:
            pend_pts.append(pts_all[i]); pend_pose.append(poses_1[i].astype(np.float32))
            if len(pend_pts) == 64:
                gmap.updateByScans(pend_pts, (0.0, 0.0), np.stack(pend_pose)); pend_pts, pend_pose = [], []
    if pend_pts:
        gmap.updateByScans(pend_pts, (0.0, 0.0), np.stack(pend_pose))
    ctx.synchronize(); ctx_map.synchronize()
    sec_scan_by_scan = time.perf_counter() - t0
    fe.reset(); gmap.reset()
    ctx.synchronize(); ctx_map.synchronize()
    # (b) the recorded trajectory handed over 256 scans at a time (lslam_frontend_process_many: one scan of look-ahead -- the
    # running-window match of scan t + 1 goes out under the loop search of scan t; same poses, same graph), the map updated
    # after every call with the poses it returned.  THIS run is the one checked against the reference's record below.
    r64m = np.stack(r64)
    poses, ok_all, upd = np.zeros((n, 3)), np.zeros(n, bool), []
    t0 = time.perf_counter()
    for i0 in range(0, n, 256):
        i1 = min(n, i0 + 256)
        ok_all[i0:i1], poses[i0:i1], _, _ = fe.ProcessMany(r64m[i0:i1], odom[i0:i1])
        idx = [i for i in range(i0, i1) if ok_all[i]]
        if idx:
            gmap.updateByScans([pts_all[i] for i in idx], (0.0, 0.0), poses[idx].astype(np.float32))
            upd.extend(idx)
    ctx.synchronize(); ctx_map.synchronize()
    sec = time.perf_counter() - t0
    look = fe.lookahead_stats()
    st = fe.stats()
    final = np.stack([fe.scan_pose(i) for i in range(fe.num_scans())])
    out = {"scans": n, "seconds": sec, "scans_per_s": n / sec, "scan_by_scan_scans_per_s": n / sec_scan_by_scan,
           "scan_by_scan_equals_look_ahead": bool(np.array_equal(poses_1, poses)), "look_ahead": look,
           "graph": st, "workload_gen_s": gen_s, "same_input_as_record": same_input,
           "map_sha256": _sha(gmap.logodds()), "map_cells_touched": int(np.count_nonzero(gmap.logodds())),
           "max_pose_err_vs_truth_xy": float(np.hypot(*(poses[:, :2] - path[:n, :2]).T).max())}
    if same_input:
        out["processed_flags_equal"] = bool(np.array_equal(ok_all, d["processed"][:n].astype(bool)))
        out["max_pose_err_vs_reference_record"] = float(np.abs(poses - d["corrected"][:n]).max())
        out["reference_edges"] = int(d["edges"][n - 1])
        out["edges_equal"] = bool(st["edges"] == int(d["edges"][n - 1]))
        if n == int(d["n"]):
            out["max_final_pose_err_vs_reference_record"] = float(np.abs(final - d["final_poses"]).max())
    gmap.close(); fe.close(); gm.close(); ctx_map.close()
    return out


def gpu_next_rows(ctx, api):
    """SURVEY 8(f) rows, each beside and checked against its CPU leg (tools/bench_extra.py holds the leg functions)."""
    sys.path.insert(0, str(ROOT / "tools"))
    import bench_extra as bx

    out = {}
    for name, fn in (("loop_closure", lambda: bx.loop_closure(ctx, 64)), ("lesson4_loop", lambda: bx.hector_front_end(ctx, 300)),
                     ("create_from_scans", lambda: bx.occgrid_from_scans(ctx, 500))):
        try:
            out[name] = fn()
        except Exception as e:
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def dropin_legs(d, args):
    """dropin_karto (scan cache) in this process, its literal-forwarding twin in a child (LSLAM_KARTO_NO_CACHE is read once
    per process), dropin_hector with the reference's own update thresholds and with a map update per scan."""
    import subprocess
    import tempfile

    out = {}
    try:
        a = dropin_karto(d)
        out["karto"] = a
        with tempfile.TemporaryDirectory() as td:
            f = os.path.join(td, "cfg5.npz")
            np.savez(f, r64=np.stack(d["r64"]), odom=d["odom"])
            env = dict(os.environ, LSLAM_KARTO_NO_CACHE="1")
            p = subprocess.run([sys.executable, str(ROOT / "tools" / "dropin_bench.py"), "--child", f], env=env, capture_output=True,
                               text=True, timeout=300)
            child = [l for l in p.stdout.splitlines() if l.startswith("CHILD ")]
            if p.returncode == 0 and child:
                b = json.loads(child[-1][6:])
                b["poses_identical_to_cached_path"] = bool(np.array_equal(np.load(f + ".poses.npy"), a["poses"]))
                out["karto_literal_forwarding"] = b
            else:
                out["karto_literal_forwarding"] = {"error": (p.stderr or "no output")[-300:]}
    except Exception as e:
        out["karto"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    for key, every in (("hector", False), ("hector_update_every_scan", True)):
        try:
            out[key] = dropin_hector(300, update_every_scan=every)
        except Exception as e:
            out[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def _source_sha(path):
    import hashlib

    try:
        return hashlib.sha256(pathlib.Path(path).read_bytes()).hexdigest()[:16]
    except OSError:
        return None


def _traffic(name):
    """PMC figures per launch from profiles/traffic.json (static: `rocprofv3 --pmc` passes, tools/pmc_passes.sh)."""
    try:
        return json.loads((ROOT / "profiles" / "traffic.json").read_text()).get(name, {})
    except Exception:
        return {}


def build_secondary(gpu, cpu, job, args):
    """The `secondary` block: cfg 2 / 3 / 5 GPU figures, each beside and checked against its CPU leg."""
    from oracle import pyoracle as po

    out = {"note": "measured in this run after the headline's timed region; CPU legs = the reference's own code "
                   "(oracle/_ref) on one host core each, in worker processes running beside the GPU legs"}
    roof, cpus = {}, {}
    # ---- cfg 2: Hector log-odds update, 1081-beam scans into 1000x1000 @ 0.05 m --------------------------------------
    g, c = gpu.get("cfg2", {}), cpu.get("cfg2")
    if "error" in g or not c:
        out["cfg2"] = g or {"error": "no CPU leg"}
    else:
        n = len(job["cfg2"]["pts"])
        alg = c["visits"] * 16 + g["points"] * 8  # (8 B read + 8 B write) per traversed cell + 8 B per point in
        k_single = sum(v[1] for v in g["single_prof"].values()) * 1e-3
        k_batch = sum(v[1] for v in g["batch_prof"].values()) * 1e-3
        tr = _traffic("logodds_batched")
        out["cfg2"] = {
            "config": "BASELINE configs[1]: Hector log-odds update, %d 1081-beam scans into 1000x1000@0.05 m (H/map/OccGridMapBase.h:118-330)" % n,
            "cell_visits_per_scan": round(c["visits"] / n), "algorithmic_bytes_per_scan": round(alg / n),
            "single_scan_per_call": {
                "scans_per_s": round(n / g["single_s"], 1), "us_per_scan": round(1e6 * g["single_s"] / n, 2),
                "cell_updates_per_s": round(c["visits"] / g["single_s"]),
                "algorithmic_GBs": round(alg / g["single_s"] / 1e9, 2), "frac_of_hbm_peak": round(alg / g["single_s"] / 1e9 / HBM_PEAK_GBS, 5),
                "kernel_us_per_scan": {k: round(1e3 * v[1] / n, 2) for k, v in sorted(g["single_prof"].items())},
                "launches_per_scan": round(sum(v[0] for v in g["single_prof"].values()) / n, 2),
                "kernel_algorithmic_GBs": round(alg / k_single / 1e9, 2) if k_single else None,
                "measured_hbm_bytes_per_scan": _traffic("logodds_pipe").get("hbm_bytes_per_launch"),
                "measured_hbm_source": _traffic("logodds_pipe").get("source"),
                "note": "pipelined: one launch per scan in steady state (apply of the previous scan + mark of this one)",
                "bit_exact_vs_cpu": g["single_sha"] == c["map_sha"]},
            "batched_64_per_call": {
                "scans_per_s": round(n / g["batch_s"], 1), "cell_updates_per_s": round(c["visits"] / g["batch_s"]),
                "algorithmic_GBs": round(alg / g["batch_s"] / 1e9, 2), "frac_of_hbm_peak": round(alg / g["batch_s"] / 1e9 / HBM_PEAK_GBS, 5),
                "kernel_ms": {k: round(v[1], 3) for k, v in sorted(g["batch_prof"].items())},
                "kernel_algorithmic_GBs": round(alg / k_batch / 1e9, 2) if k_batch else None,
                "measured_hbm_bytes_per_64_scan_call": tr.get("hbm_bytes_per_launch"),
                "measured_hbm_source": tr.get("source"),
                # the call's scratch: per-scan windows of 8x8-cell tiles in a slot pool (+ 64 flag bytes per tile), not 64
                # byte planes of the whole map; `rounds` = passes the last 64-scan group needed under the budget
                "scratch_bytes": g.get("batch_stats", {}).get("scratch_bytes"),
                "scratch_rounds": g.get("batch_stats", {}).get("rounds"),
                "window_misses": g.get("batch_stats", {}).get("window_misses"),
                "bit_exact_vs_cpu": g["batch_sha"] == c["map_sha"]},
            "cpu": {"scans_per_s": round(c["scans_per_s"], 1), "kind": c["kind"], "cores": 1,
                    "restated_oracle_scans_per_s": round(c["port_scans_per_s"], 1), "reference_equals_restatement": c.get("ref_equals_port")},
        }
        roof["cfg2_single_frac_of_hbm"] = out["cfg2"]["single_scan_per_call"]["frac_of_hbm_peak"]
        roof["cfg2_single_scans_per_s"] = out["cfg2"]["single_scan_per_call"]["scans_per_s"]
        roof["cfg2_batched_frac_of_hbm"] = out["cfg2"]["batched_64_per_call"]["frac_of_hbm_peak"]
        roof["cfg2_batched_scans_per_s"] = out["cfg2"]["batched_64_per_call"]["scans_per_s"]
        roof["cfg2_batched_scratch_bytes"] = out["cfg2"]["batched_64_per_call"]["scratch_bytes"]
        roof["cfg2_bit_exact"] = bool(g["single_sha"] == c["map_sha"] and g["batch_sha"] == c["map_sha"])
        cpus["cfg2_scans_per_s"] = round(c["scans_per_s"], 1)
        cpus["cfg2_kind"] = c["kind"]
    # ---- cfg 3: one complete MatchScan per call ------------------------------------------------------------------
    g, c = gpu.get("cfg3", {}), cpu.get("cfg3")
    if "error" in g:
        out["cfg3"] = g
    else:
        out["cfg3"] = {"config": "BASELINE configs[2]: single-scan MatchScan (AddScans of the 70-scan window + coarse/fine search, "
                                 "Mapper.cpp:184-291), host entry point incl. the PCIe upload of the window",
                       "calls": len(job["cfg3"]["idx"]), "gpu_ms_per_match": round(g["ms_per_match"], 4),
                       "gpu_ms_per_match_window_in_scan_cache": round(g["ms_per_match_cached"], 4),
                       "scan_cache_records_identical": g["cached_identical"]}
        if c:
            out["cfg3"].update({"cpu_reference_ms_per_match": round(c["ms_per_match"], 4), "cpu_cores": 1,
                                "max_pose_err_vs_reference": float(np.abs(g["poses"] - c["poses"]).max()),
                                "max_response_err_vs_reference": float(np.abs(g["resp"] - c["resp"]).max())})
            cpus["cfg3_ms_per_match"] = out["cfg3"]["cpu_reference_ms_per_match"]
            roof["cfg3_max_pose_err"] = out["cfg3"]["max_pose_err_vs_reference"]
        roof["cfg3_ms_per_match"] = out["cfg3"]["gpu_ms_per_match"]
        roof["cfg3_ms_per_match_scan_cache"] = out["cfg3"]["gpu_ms_per_match_window_in_scan_cache"]
    # ---- cfg 5: streaming front-end slice ------------------------------------------------------------------------
    g, c = gpu.get("cfg5", {}), cpu.get("cfg5")
    if "error" in g:
        out["cfg5"] = g
    else:
        d = job["cfg5"]
        n = len(d["r64"])
        st = g["stats"]
        # map parity: the restated Hector update (pinned to the reference's headers) fed the GPU's poses
        po.build("restate")
        cmap = po.PortHector(g["n"], g["n"], g["cell"], g["off"])
        cmap.setUpdateOccupiedFactor(0.9)
        for i in g["updated"]:
            cmap.updateByScan(d["pts"][i], (0.0, 0.0), g["poses"][i].astype(np.float32))
        ref_map = cmap.logodds()
        chain = {k: round(1e3 * v[1] / n, 2) for k, v in sorted(g["prof"].items(), key=lambda kv: -kv[1][1])}
        out["cfg5"] = {
            "config": "BASELINE configs[4] slice: %d-scan closed-loop trajectory (0.25 m steps, drifting odometry) in the 100 m arena; "
                      "per scan Mapper::Process with its pose graph (70-scan running window, LinkNearChains, TryCloseLoop on the "
                      "81x81x21 loop matcher; Mapper.cpp:1999-2079) + log-odds update of the 4000x4000@0.025 m map, 64 scans per call" % n,
            "scans": n, "gpu_scans_per_s": round(n / g["seconds"], 1), "gpu_us_per_scan": round(1e6 * g["seconds"] / n, 1),
            "graph": st, "kernel_us_per_scan": dict(list(chain.items())[:12]), "kernel_us_per_scan_total": round(sum(chain.values()), 1),
            "map_bit_exact": bool(g["logodds"].tobytes() == ref_map.tobytes()), "map_cells_touched": int(np.count_nonzero(ref_map)),
            "max_pose_err_vs_truth_xy": float(np.hypot(*(g["poses"][:, :2] - d["path"][:, :2]).T).max()),
            "odometry_drift_max_xy": float(np.hypot(*(d["odom"][:, :2] - d["path"][:, :2]).T).max()),
        }
        if c:
            out["cfg5"].update({"cpu_reference_scans_per_s": round(c["scans_per_s"], 2), "cpu_reference_seconds": round(c["seconds"], 1),
                                "cpu_cores": 1, "max_pose_err_vs_reference": float(np.abs(g["poses"] - c["poses"]).max()),
                                "reference_graph_edges": c["edges"], "edges_equal": bool(c["edges"] == st["edges"])})
            cpus["cfg5_scans_per_s"] = out["cfg5"]["cpu_reference_scans_per_s"]
            roof["cfg5_max_pose_err"] = out["cfg5"]["max_pose_err_vs_reference"]
            roof["cfg5_edges_equal"] = out["cfg5"]["edges_equal"]
        roof["cfg5_scans_per_s"] = out["cfg5"]["gpu_scans_per_s"]
        roof["cfg5_loops_closed"] = st.get("loops_closed")
        roof["cfg5_map_bit_exact"] = out["cfg5"]["map_bit_exact"]
    # ---- cfg 5 at its stated size ---------------------------------------------------------------------------------
    g = gpu.get("cfg5_full")
    if g is not None:
        if "error" in g:
            out["cfg5_full"] = g
        else:
            out["cfg5_full"] = {
                "config": "BASELINE configs[4] at its stated size: %d-scan closed-loop trajectory, pose graph on, 4000x4000@0.025 m map; "
                          "scans_per_s = the recorded trajectory handed over 256 scans per call (lslam_frontend_process_many: the "
                          "running-window match of scan t+1 enqueued under the loop search of scan t), scan_by_scan_scans_per_s = one "
                          "Mapper::Process call per scan; both the same poses; "
                          "checked against the record of the reference's Mapper::Process over the same scans "
                          "(tests/golden/karto_cfg5_golden.npz; the reference needs ~39 min of one host core for it)" % g["scans"],
                **{k: (round(v, 4) if isinstance(v, float) and k not in ("max_pose_err_vs_reference_record", "max_final_pose_err_vs_reference_record") else v)
                   for k, v in g.items()}}
            roof["cfg5_full_scans_per_s"] = round(g["scans_per_s"], 1)  # lslam_frontend_process_many (one scan of look-ahead)
            roof["cfg5_full_scan_by_scan_scans_per_s"] = round(g["scan_by_scan_scans_per_s"], 1)  # Mapper::Process, call by call
            roof["cfg5_full_scan_by_scan_equals_look_ahead"] = g.get("scan_by_scan_equals_look_ahead")
            roof["cfg5_full_max_pose_err_vs_reference_record"] = g.get("max_pose_err_vs_reference_record")
            roof["cfg5_full_edges_equal"] = g.get("edges_equal")
            cpus["cfg5_full_reference_scans_per_s"] = 4.3  # tests/golden/make_cfg5_golden.py: 10 000 scans in 2 321 s (recorded, not re-run)
    # ---- SURVEY 8(f) rows ----------------------------------------------------------------------------------------
    nr = gpu.get("next_rows")
    if nr is not None:
        out["next_rows"] = nr
        lc, l4, cs = nr.get("loop_closure", {}), nr.get("lesson4_loop", {}), nr.get("create_from_scans", {})
        roof["loop_closure_ms_single"] = lc.get("gpu_single_ms_per_match")
        roof["loop_closure_ms_batched"] = lc.get("gpu_batched_ms_per_match")
        roof["lesson4_loop_scans_per_s"] = l4.get("gpu_scans_per_s")
        roof["create_from_scans_ms"] = cs.get("gpu_ms")
        cpus["loop_closure_ms"] = lc.get("cpu_port_ms_per_match")
        cpus["lesson4_loop_scans_per_s"] = l4.get("cpu_port_scans_per_s")
        cpus["create_from_scans_ms"] = cs.get("cpu_reference_ms")
    # ---- drop-in legs --------------------------------------------------------------------------------------------
    dr = gpu.get("dropin")
    if dr is not None:
        o = {"note": "the reference's OWN orchestrators (oracle/_ref_gpu: Karto.o + Mapper.o with MatchScan / ~ScanMatcher substituted "
                     "at link time; HectorSlamProcessor with mapRep = HectorMapRepGpu) driving the HIP path through the C ABI; "
                     "integration legs, never the headline"}
        k, c5, g5 = dr.get("karto", {}), cpu.get("cfg5"), gpu.get("cfg5", {})
        if "error" in k or not k:
            o["karto"] = k
        else:
            hp = k["host_profile"]
            o["karto"] = {
                "config": "reference karto::Mapper::Process + GPU MatchScan through the device-side scan cache, the cfg 5 slice",
                "scans": k["scans"], "scans_per_s": round(k["scans_per_s"], 1), "device_match_calls": k["device_match_calls"],
                "us_per_device_match_call": round(k["us_per_match_call"], 2), "scans_uploaded": k["scans_uploaded"],
                "refreshes_before_a_match": k["refreshes_before_a_match"], "edges": k["edges"],
                "host_profile": {kk: round(v, 4) for kk, v in hp.items()},
                "native_frontend_scans_per_s": round(len(job["cfg5"]["r64"]) / g5["seconds"], 1) if "seconds" in g5 else None,
                "max_pose_diff_vs_native_frontend": float(np.abs(g5["poses"] - k["poses"]).max()) if "poses" in g5 else None}
            if c5:
                o["karto"].update({"pure_reference_scans_per_s": round(c5["scans_per_s"], 2),
                                   "max_pose_err_vs_pure_reference": float(np.abs(c5["poses"] - k["poses"]).max()),
                                   "edges_equal_pure_reference": bool(c5["edges"] == k["edges"])})
            lf = dr.get("karto_literal_forwarding", {})
            if "scans_per_s" in lf:
                o["karto"]["literal_forwarding"] = {"scans_per_s": round(lf["scans_per_s"], 1), "us_per_device_match_call": lf["us_per_match_call"],
                                                    "poses_identical_to_cached_path": lf.get("poses_identical_to_cached_path"),
                                                    "host_profile": lf.get("host_profile")}
                o["karto"]["speedup_from_scan_cache"] = round(k["scans_per_s"] / lf["scans_per_s"], 3)
            else:
                o["karto"]["literal_forwarding"] = lf
            roof["dropin_karto_scans_per_s"] = o["karto"]["scans_per_s"]
            roof["dropin_karto_share_reference_host_code"] = round(hp["share_reference_host_code"], 3)
        for key in ("hector", "hector_update_every_scan"):
            h = dr.get(key, {})
            o[key] = {kk: (float("%.4g" % v) if isinstance(v, float) else v) for kk, v in h.items()}
            if "gpu_scans_per_s" in h:
                roof["dropin_%s_scans_per_s" % key] = round(h["gpu_scans_per_s"], 1)
                cpus["dropin_%s_reference_scans_per_s" % key] = round(h["cpu_reference_scans_per_s"], 1)
        out["dropin"] = o
    # ---- the two extra batched-match legs: every record against the reference's CorrelateScan (all host cores) --------
    for leg, key in (("indoor", "cfg_indoor_default"), ("dense", "headline_dense_world")):
        g, c = gpu.get(leg), cpu.get(leg)
        if g is None:
            continue
        if "error" in g or not isinstance(c, list):
            out[key] = g if "error" in g else {"error": "no CPU leg", "gpu_scans_per_s": round(g["n"] / g["s_per_step"], 1)}
            continue
        n = g["n"]
        rec = g["records"]
        pose_err, resp_err, checked = 0.0, 0.0, 0
        for part in c:
            gi = rec[part["idx"]]
            ok = gi["status"] == 0
            d = np.abs(gi["pose"][ok] - part["poses"][ok])
            d[:, 2] = np.abs(np.remainder(d[:, 2] + np.pi, 2 * np.pi) - np.pi)
            pose_err = max(pose_err, float(d.max()) if d.size else 0.0)
            resp_err = max(resp_err, float(np.abs(gi["response"][ok] - part["resp"][ok]).max()) if ok.any() else 0.0)
            checked += int(ok.sum())
        busy = [p["busy_s"] for p in c]
        per_core = [len(p["idx"]) / p["busy_s"] for p in c if p["busy_s"] > 0]
        out[key] = {
            "config": job[leg]["what"], "scans_per_step": n, "grid": g["grid"], "smear_kernel": g["kernel_size"],
            "ms_per_step": round(1e3 * g["s_per_step"], 4), "scans_per_s": round(n / g["s_per_step"], 1),
            "ms_per_step_pruning_off": round(1e3 * g["s_per_step_pruning_off"], 4),
            "scans_per_s_pruning_off": round(n / g["s_per_step_pruning_off"], 1),
            "pruned_row_fraction": g["pruned_row_fraction"], "beam_angles_with_no_live_row": g["beam_angles_with_no_live_row"],
            "results_identical_pruning_off": g["pruning_off_identical"], "kernel_ms_one_instrumented_step": g["kernel_ms"],
            "records_ok": int((rec["status"] == 0).sum()), "zero_response_records": int((rec["response"] == 0.0).sum()),
            "records_checked_vs_reference": checked, "max_pose_err_vs_reference": pose_err, "max_response_err_vs_reference": resp_err,
            "cpu": {"kind": "reference", "scans_per_s_per_core": round(float(np.median(per_core)), 2), "cores": len(c),
                    "scans_per_s_all_cores": round(n / max(busy), 1),
                    "note": "oracle/_ref: the reference's CorrelateScan (coarse, MatchScan's response expansion, fine) vs the same "
                            "grid, every %d-th scan per host process" % len(c)},
        }
        short = "indoor_default" if leg == "indoor" else "dense_world"
        roof[short + "_scans_per_s"] = out[key]["scans_per_s"]
        roof[short + "_pruned_row_fraction"] = out[key]["pruned_row_fraction"]
        roof[short + "_scans_per_s_pruning_off"] = out[key]["scans_per_s_pruning_off"]
        roof[short + "_max_pose_err"] = pose_err
        cpus[short + "_reference_scans_per_s_per_core"] = out[key]["cpu"]["scans_per_s_per_core"]
    out["roofline_summary"], out["cpu_summary"] = roof, cpus
    return out


def ms_per_step_guess(nb, n_full, sec_full):
    """rough step time [ms] of a batch of nb scans from the measured full batch (floor: the five-kernel latency chain)"""
    return max(0.08, 1e3 * sec_full * nb / max(n_full, 1))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _visible_gpus():
    import torch

    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000, help="timed steps (default: ~0.75 s of GPU time at N=1)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096, help="scans per step: in total (strong) / per GPU (weak)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--n-base", type=int, default=70, help="running-window scans rasterised into the grid")
    ap.add_argument("--spread", type=float, default=3.0, help="radius [m] of the disc the query poses are drawn from")
    ap.add_argument("--unique", type=int, default=0, help="distinct ray-cast scans (0 = all of them; 64 = round 1's tiling)")
    ap.add_argument("--cpu-sample", type=int, default=4096, help="scan-matches timed on the host for cpu_baseline")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-cores", type=int, default=0, help="also time the reference on this many host cores (0 = min(nproc, 64))")
    ap.add_argument("--broadcast-grid", action="store_true", help="build the grid on rank 0 and RCCL-broadcast it")
    ap.add_argument("--no-diagnostics", action="store_true", help="skip the untimed pruning statistics / pruning-off step")
    ap.add_argument("--dump-results", default="", help="write the gathered 112-byte result records of the last step to this .npy")
    ap.add_argument("--no-secondary", action="store_true", help="skip the cfg 2 / 3 / 5 legs (N=1 only anyway)")
    ap.add_argument("--map-scans", type=int, default=1000, help="cfg 2: scans integrated into the 1000x1000 map")
    ap.add_argument("--single", type=int, default=200, help="cfg 3: complete MatchScan calls")
    ap.add_argument("--stream-scans", type=int, default=600, help="cfg 5 slice: scans of the closed-loop trajectory")
    ap.add_argument("--sustained-s", type=float, default=10.0, help="seconds of the untimed-by-contract sustained leg (0 = off)")
    ap.add_argument("--no-full-cfg5", action="store_true", help="skip config 5 at its stated size (10 000 scans, ~2 s of GPU time)")
    ap.add_argument("--no-dropin", action="store_true", help="skip the drop-in legs (reference orchestrators on the HIP path)")
    ap.add_argument("--no-next-rows", action="store_true", help="skip the SURVEY 8(f) rows")
    ap.add_argument("--no-batch-legs", action="store_true",
                    help="skip the two extra batched-match legs (the reference's shipped indoor configuration; the dense world)")
    ap.add_argument("--pipeline-depth", type=int, default=1,
                    help="LSLAM_OPT_PIPELINE_DEPTH of the contract's timed region (default 1: one step after the other, the "
                         "library's default and what a caller that consumes each step's poses gets).")
    ap.add_argument("--pipelined-leg-depth", type=int, default=2,
                    help="depth of the separate, labelled `pipelined_leg` (independent steps overlapping on the matcher's internal "
                         "streams: the product's option for callers whose steps do not depend on each other); 0 = skip")
    ap.add_argument("--plain-steps", type=int, default=0, help="steps of the plain (depth-1) roofline leg (0 = min(--steps, 300))")
    args = ap.parse_args()
    backend = os.environ.get("LSLAM_BENCH_BACKEND", "nccl")  # gloo: the N>1 control flow on a 1-GPU box (tests)
    # LSLAM_BENCH_FORCE_DIST=1: initialise torch.distributed (and run every collective of the N > 1 line) at world size 1
    # too -- how the 1-GPU test box executes the RCCL branch before an 8-GPU node does
    force_dist = os.environ.get("LSLAM_BENCH_FORCE_DIST", "0") not in ("", "0")
    depth = max(1, min(4, args.pipeline_depth))
    leg_depth = max(0, min(4, args.pipelined_leg_depth))
    # LSLAM_BENCH_SHARE_GPU=1 (tests only): ranks beyond the visible GPUs wrap around -- with nccl this asks RCCL for a
    # communicator with two ranks on one device, which it may refuse
    share_gpu = os.environ.get("LSLAM_BENCH_SHARE_GPU", "0") not in ("", "0")

    # ---- `python bench.py --gpus N`: start the N ranks ourselves ------------------------------------------------
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        have = _visible_gpus()
        if backend == "nccl" and have < args.gpus and not share_gpu:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(pathlib.Path(__file__).resolve()),
               *sys.argv[1:]]
        os.execv(sys.executable, cmd)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_size}")
    distributed = world_size > 1 or force_dist
    if distributed and "WORLD_SIZE" not in os.environ:  # forced at N = 1 outside torchrun: a rendezvous of one
        os.environ.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                           "MASTER_PORT": str(_free_port())})

    # stdout carries the ONE JSON line and nothing else: the reference library behind the CPU baseline
    # (oracle/_ref) chats on std::cout ("Registering sensor ...", also at exit), so fd 1 is pointed at stderr
    # for everything but the final print
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import lslam  # noqa: F401
    from lslam_amd import shard, synth

    # ---- synthetic workload (SURVEY.md §8(d) cfg 4): same grid on every rank, own scans per rank ----------------
    laser = synth.Laser()
    world = synth.arena()
    wl = synth.make_match_workload(n_base=args.n_base, n_query=1, seed=5, query_spread=args.spread, world=world)
    B = args.batch
    n_total = B if args.scaling == "strong" else B * world_size
    lo, hi = shard.shard_range(n_total, world_size, rank) if args.scaling == "strong" else (rank * B, (rank + 1) * B)
    n_unique = n_total if args.unique <= 0 else min(args.unique, n_total)
    truth_u = query_poses(world, wl.center_pose, n_unique, args.spread, seed=55)
    idx = np.arange(n_total) % n_unique
    truth = truth_u[idx]
    odom = synth.perturb(truth, 0.3, np.deg2rad(10.0), 77)  # every problem has its own search centre
    t_gen = time.perf_counter()
    procs = max(1, min(32, (os.cpu_count() or 1) // world_size))
    if n_unique == n_total:
        my_ranges = cast_scans(world, laser, truth[lo:hi], lo, 555, procs)
    else:
        my_ranges = cast_scans(world, laser, truth_u, 0, 555, procs)[idx[lo:hi]]
    t_gen = time.perf_counter() - t_gen
    my_odom = np.ascontiguousarray(odom[lo:hi])
    n_mine = hi - lo

    # ---- CPU legs: fork the workers now (no HIP / CUDA state in this process yet); they start when GO is set ------
    global GO, GO_MULTI
    do_cpu = (not args.no_cpu) and rank == 0  # N > 1: rank 0 times the 1-core leg too (released after the timed region)
    do_secondary = do_cpu and not args.no_secondary and world_size == 1
    pool, tasks = None, {}
    cores = args.cpu_cores or min(os.cpu_count() or 1, 64)
    if do_cpu:
        import multiprocessing as mp

        from oracle import pyoracle as po

        sample = max(8, min(args.cpu_sample if world_size == 1 else min(args.cpu_sample, 2048), n_mine))
        _JOB.update({"wl": wl, "q_r": synth.ranges_to_f64(my_ranges[:sample]), "q_p": my_odom[:sample]})
        if do_secondary:
            t_sec = time.perf_counter()
            _JOB.update(secondary_workloads(args.map_scans, args.single, args.stream_scans, procs))
            if not args.no_batch_legs:
                _JOB.update(batch_leg_workloads(min(args.batch, 4096), procs))
            t_gen_secondary = time.perf_counter() - t_sec
        fork = mp.get_context("fork")
        GO, GO_MULTI = fork.Event(), fork.Event()
        multi = po.have_ref() and world_size == 1  # the all-core leg would compete with the other ranks' host threads
        n_workers = 1 + (cores if multi else 0) + (3 if do_secondary else 0)
        pool = fork.Pool(n_workers, initializer=_quiet_worker)
        if do_secondary:  # longest first
            tasks["cfg5"] = pool.apply_async(_cpu_cfg5, (0,))
        tasks["cfg4"] = pool.apply_async(_cpu_cfg4_single, (0,))
        if do_secondary:
            tasks["cfg3"] = pool.apply_async(_cpu_cfg3, (0,))
            tasks["cfg2"] = pool.apply_async(_cpu_cfg2, (0,))
        if multi:
            tasks["cfg4_multi"] = [pool.apply_async(_cpu_cfg4_share, (i,)) for i in range(cores)]
            for leg in ("indoor", "dense"):  # behind the cfg-4 shares on the same worker processes
                if leg in _JOB:
                    tasks[leg] = [pool.apply_async(_cpu_batch_share, ((leg, i, cores),)) for i in range(cores)]

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X GPU (no CPU fallback)")
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and local_rank >= n_dev and not share_gpu:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {n_dev} GPU(s) visible")
    if backend != "nccl" or share_gpu:
        local_rank = local_rank % n_dev  # test mode only: several ranks share one GPU
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    from lslam_amd import api

    dev = torch.device("cuda", local_rank)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    ctx = api.Context(local_rank)
    cfg = api.baseline_config()
    gm = api.ScanMatcher(ctx, cfg, api.laser_params(laser))
    assert gm.num_beams == N_BEAMS
    if args.broadcast_grid and distributed:
        if rank == 0:
            gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
        shard.broadcast_grid(gm, coll_dev, src=0, install_device=dev)  # RCCL broadcast of the 4 MB grid
    else:
        gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)  # redundant build: cheaper than the broadcast
    ranges32 = torch.from_numpy(np.ascontiguousarray(my_ranges)).to(dev)
    poses = torch.from_numpy(my_odom).to(dev)
    # one record buffer per step in flight: pipelined steps must not share an output buffer (lslam_gpu.h)
    results_ring = [torch.zeros((max(n_mine, 1), 112), dtype=torch.uint8, device=dev) for _ in range(max(depth, leg_depth, 1))]
    results = results_ring[0]
    torch.cuda.synchronize()
    step_no, cur_depth = [0], [1]

    def step():
        if n_mine:
            out = results_ring[step_no[0] % cur_depth[0]]  # depth 1: always `results`
            step_no[0] += 1
            gm.match_batch_dev(n_mine, ranges32.data_ptr(), N_BEAMS, poses.data_ptr(), out.data_ptr(), dtype="f32")

    def set_depth(d):
        gm.set_option("pipeline_depth", d)  # joins whatever is in flight
        step_no[0], cur_depth[0] = 0, d

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()

    def timed(n_steps):
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step()
        barrier()  # stream + device sync, then the RCCL barrier: the timed region is bracketed on both sides
        el = time.perf_counter() - t0
        timed.local = el  # this rank's own clock over the region (reported per rank in the N > 1 line)
        if distributed:
            t = torch.tensor([el], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    # ---- untimed: one PLAIN step with HIP events around EVERY kernel (the per-kernel table, the dominant kernel's name)
    step()  # first use: allocations, the grid's derived views
    ctx.profile(True)
    ctx.profile_only(None)
    ctx.profile_reset()
    step()
    prof_all = ctx.profile_read()
    ctx.profile(False)
    dom_name = max(prof_all, key=lambda k: prof_all[k][1]) if prof_all else None
    # ---- THE CONTRACT: W untimed warm-up steps, then exactly K timed steps between barriers -- the product's pipelined
    # steps (depth D in flight on the matcher's internal streams, byte-identical records), no event anywhere in the region
    set_depth(depth)
    for _ in range(args.warmup):
        step()
    elapsed = timed(args.steps)
    local_elapsed = timed.local
    # ---- the roofline leg: the PLAIN step (depth 1, one kernel after the other on one stream -- the per-kernel
    # bookkeeping `roofline` prices), HIP events around the dominant kernel only (two events per launch cost ~3 us of
    # stream time), timed like the region above
    set_depth(1)
    step()
    n_plain = args.plain_steps or min(args.steps, 300)
    ctx.profile(True)
    ctx.profile_only(dom_name)
    ctx.profile_reset()
    clk0 = ctx.clock_sample()
    elapsed_plain = timed(n_plain)
    clk1 = ctx.clock_sample()
    clock_ghz = ctx.clock_ghz(clk0, clk1)  # the shader clock the leg ran at (s_memtime / s_memrealtime), not an assumed 2.4 GHz
    ctx.profile(False)
    prof = ctx.profile_read()  # the dominant kernel, timed live over the plain leg
    ctx.profile_only(None)
    # ---- the pipelined leg (labelled, never `value` unless --pipeline-depth asks for it): the same step with
    # `leg_depth` steps in flight on the matcher's internal streams
    pipelined_leg = None
    if leg_depth > 1 and n_mine:
        set_depth(leg_depth)
        for _ in range(max(args.warmup, leg_depth)):
            step()
        el_leg = timed(args.steps)
        pipelined_leg = {"depth": leg_depth, "steps": args.steps, "ms_per_step": round(1e3 * el_leg / args.steps, 4),
                         "value": round(n_total * args.steps / el_leg, 1), "unit": "scan-matches/s",
                         "note": "INDEPENDENT steps overlapping (LSLAM_OPT_PIPELINE_DEPTH): every step matches the same inputs into "
                                 "its own record buffer; a caller whose step t + 1 needs the poses of step t cannot use it"}
    set_depth(depth)
    per_rank_ms = None
    if distributed:  # every rank's own time per step: a measured scaling curve can be diagnosed (stragglers, small-batch floor)
        t = torch.tensor([local_elapsed], dtype=torch.float64, device=coll_dev)
        parts = [torch.zeros_like(t) for _ in range(world_size)]
        dist.all_gather(parts, t)
        per_rank_ms = [round(1e3 * float(x.item()) / args.steps, 4) for x in parts]
    # who ran: the communicator's size as the backend reports it, and every rank's device (name + PCI bus id)
    rccl_ranks, device_names = None, [f"{torch.cuda.get_device_name(local_rank)} [cuda:{local_rank}]"]
    if distributed:
        rccl_ranks = dist.get_world_size() if backend == "nccl" else None
        mine = f"{torch.cuda.get_device_name(local_rank)} [cuda:{local_rank}]".encode()[:64].ljust(64)
        t = torch.tensor(list(mine), dtype=torch.uint8, device=coll_dev)
        parts = [torch.zeros_like(t) for _ in range(world_size)]
        dist.all_gather(parts, t)
        device_names = [bytes(x.cpu().tolist()).decode(errors="replace").strip() for x in parts]

    # ---- sustained leg: the same step for >= --sustained-s seconds (clocks ramped, visible to a 1 Hz busy sampler).
    # Reported beside `value`, never as it: `value` is the K steps of the contract above.
    sustained = None
    if args.sustained_s > 0 and n_mine:
        n_sus = max(args.steps, int(args.sustained_s / max(elapsed / args.steps, 1e-6)) + 1)
        el_sus = timed(n_sus)
        sustained = {"steps": n_sus, "seconds": round(el_sus, 3), "ms_per_step": round(1e3 * el_sus / n_sus, 4),
                     "value": round(n_total * n_sus / el_sus, 1), "unit": "scan-matches/s",
                     "note": "the same step repeated for >= --sustained-s seconds after the contract's region; `value` at the top "
                             "of this line is the K = --steps timed steps of the contract, not this leg"}
    # Config 5 at its stated size is the one GPU leg whose rate is the HOST thread's (10 000 calls, a spin-waiting front-end):
    # it runs before the CPU legs are released (beside them it reads ~12 % lower; every other GPU leg runs beside them).
    sec_gpu = {}
    if do_secondary and not args.no_full_cfg5:
        try:
            sec_gpu["cfg5_full"] = gpu_cfg5_full(ctx, api)
            sec_gpu["cfg5_full"]["host"] = "before the CPU legs of this run are released"
        except Exception as e:
            sec_gpu["cfg5_full"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if GO is not None:
        GO.set()  # the host is free now: release the CPU legs

    # ---- "poses out": gather every rank's result records (112 B/scan), timed on its own --------------------------
    barrier()
    # every buffer of the ring was written by a pipelined step of the regions above: they must all hold the same records
    ring_identical = all(bool(torch.equal(r, results)) for r in results_ring[1:])
    res_local = results[:n_mine]
    res_np = res_local.cpu().numpy().view(api.RESULT_DTYPE).reshape(-1)
    gather_ms = None
    if distributed:
        src = res_local if backend == "nccl" else res_local.cpu()
        gathered = shard.all_gather_results(src, world_size)  # warm-up (communicator set-up)
        barrier()
        t0 = time.perf_counter()
        for _ in range(5):
            gathered = shard.all_gather_results(src, world_size)
        barrier()
        gather_ms = (time.perf_counter() - t0) / 5 * 1e3
        all_np = gathered.cpu().numpy().view(api.RESULT_DTYPE).reshape(-1)
    else:
        all_np = res_np
    n_ok_all = int((all_np["status"] == 0).sum())

    # ---- untimed diagnostics: how much the exact zero-row pruning removes, and the step time without it ----------
    pruning = None
    set_depth(1)  # every diagnostic below is a plain-step measurement unless it says otherwise
    if not args.no_diagnostics:
        gm.set_option("collect_stats", 1)
        step()
        ctx.synchronize()
        st = gm.read_stats()
        gm.set_option("collect_stats", 0)
        gm.set_option("row_occupancy", 0)
        step()
        n_off = max(2, min(200, args.steps // 3))
        el_off = timed(n_off)
        gm.set_option("row_occupancy", 1)
        step()
        off_np = results[:n_mine].cpu().numpy().view(api.RESULT_DTYPE).reshape(-1)
        same = bool(np.array_equal(off_np["pose"], res_np["pose"]) and np.array_equal(off_np["response"], res_np["response"]))
        pruning = {
            "pruned_row_fraction": round(1.0 - st["rows_live"] / max(st["rows_in_range"], 1), 5),
            "beam_angles_with_no_live_row": round(1.0 - st["beam_angles_queued"] / max(st["beam_angles"], 1), 5),
            # beams that have no live row in ANY of their scan's 21 coarse angles: what compacting each scan's beam list
            # once, in front of the hot kernel, could remove from it (worth doing from 25 % up)
            "beams_with_no_live_row_in_any_angle": round(1.0 - st["beams_live_in_some_angle"] / max(st["beams_readable"], 1), 5),
            "ms_per_step_pruning_off": round(1e3 * el_off / n_off, 4),
            "results_identical_pruning_off": same,
            "note": "coarse pass of this rank's scans; pruning is exact (a pruned row is provably all zero), the "
                    "pruning-off step time is the worst case over world sparsity",
        }

    # ---- untimed-by-contract: the LDS-staged variant of the hot kernel (north star: "pyramid staged in LDS tiles"; built as
    # an experiment, measured here, not used): phase B of the coarse pass reads its rows from per-drain patches of the
    # linear parity planes staged in LDS instead of gathering from the L2-resident tiled planes
    lds_experiment = None
    if not args.no_diagnostics and n_mine >= 2048:
        try:
            gm.set_option("lds_staged", 1)
            gm.set_option("collect_stats", 1)
            step(); ctx.synchronize()
            st_l = gm.read_stats()
            gm.set_option("collect_stats", 0)
            step()
            n_l = max(2, min(100, args.steps // 5))
            ctx.profile(True); ctx.profile_only("resp_rows_coarse"); ctx.profile_reset()
            el_l = timed(n_l)
            ctx.profile(False)
            pl = ctx.profile_read().get("resp_rows_coarse", (0, 0.0))
            ctx.profile_only(None)
            l_np = results[:n_mine].cpu().numpy().view(api.RESULT_DTYPE).reshape(-1)
            gm.set_option("lds_staged", 0)
            step(); ctx.synchronize()
            lds_experiment = {
                "ms_per_step": round(1e3 * el_l / n_l, 4), "resp_rows_coarse_ms_per_launch": round(pl[1] / max(pl[0], 1), 4),
                "drains_staged_in_lds": st_l["lds_drains_staged"], "drains_on_the_global_path": st_l["lds_drains_global"],
                "results_identical": bool(np.array_equal(l_np["pose"], res_np["pose"]) and np.array_equal(l_np["response"], res_np["response"])),
                "note": "k_resp_rows<3,11,false,false,true>: 6 KB LDS patch per parity plane per drain of 64 queued beams "
                        "(LSLAM_OPT_LDS_STAGED); compare ms_per_step / roofline.avg_launch_ms of the shipped kernel"}
        except Exception as e:  # pragma: no cover
            lds_experiment = {"error": f"{type(e).__name__}: {e}"[:200]}
            try:
                gm.set_option("lds_staged", 0); gm.set_option("collect_stats", 0)
            except Exception:
                pass

    # ---- untimed-by-contract: what the pipelined steps buy at the per-GPU batch sizes of a strong-scaling run.  The same
    # matcher, LSLAM_OPT_PIPELINE_DEPTH 1..4, batches of B/8, B/4, B/2, B scans: ms per step (wall time / steps) and the
    # efficiency an N-GPU run of the timed region's depth would reach if nothing but the per-GPU step time mattered:
    # T_1(B) / (N * T(B/N)).  Records of every depth are compared with the plain step's.
    pipelined = None
    if not args.no_diagnostics and world_size == 1 and n_mine >= 512:
        try:
            pipelined = {"note": "one ScanMatcher, LSLAM_OPT_PIPELINE_DEPTH = d: d steps in flight on the matcher's internal "
                                 "streams, own per-step workspaces, shared grid; ms = wall time / steps; depth 1 = plain",
                         "timed_region_depth": depth, "predicted_efficiency_depth": (leg_depth if leg_depth > 1 else depth)}
            sizes = sorted({max(64, n_mine // 8), max(64, n_mine // 4), max(64, n_mine // 2), n_mine})
            ring4 = [torch.zeros((n_mine, 112), dtype=torch.uint8, device=dev) for _ in range(4)]
            torch.cuda.synchronize()  # torch's fill kernels run on torch's stream: done before the library's streams write
            table, same_all = {}, True
            for nb in sizes:
                row = {}
                n_st = max(24, min(400, int(0.2 / (1e-3 * ms_per_step_guess(nb, n_mine, elapsed_plain / n_plain)))))
                for d in (1, 2, 3, 4):
                    gm.set_option("pipeline_depth", d)
                    for k in range(4):
                        gm.match_batch_dev(nb, ranges32.data_ptr(), N_BEAMS, poses.data_ptr(), ring4[k % d].data_ptr(), dtype="f32")
                    ctx.synchronize()
                    t0 = time.perf_counter()
                    for k in range(n_st):
                        gm.match_batch_dev(nb, ranges32.data_ptr(), N_BEAMS, poses.data_ptr(), ring4[k % d].data_ptr(), dtype="f32")
                    t_enq = time.perf_counter() - t0  # the host's share: every call has returned, nothing awaited yet
                    ctx.synchronize()
                    row["depth_%d_ms" % d] = round(1e3 * (time.perf_counter() - t0) / n_st, 4)
                    row["depth_%d_host_enqueue_ms" % d] = round(1e3 * t_enq / n_st, 4)
                    same_all = same_all and all(bool(torch.equal(ring4[k][:nb], results[:nb])) for k in range(d))
                row["steps"] = n_st
                table["batch_%d" % nb] = row
            pipelined["ms_per_step"] = table
            pipelined["results_identical"] = same_all
            key = "depth_%d_ms" % (leg_depth if leg_depth > 1 else depth)  # the pipelined prediction; `_plain` below is depth 1
            t_full = table["batch_%d" % n_mine][key]
            pipelined["predicted_strong_scaling_efficiency"] = {
                str(n): round(t_full / (n * table["batch_%d" % max(64, n_mine // n)][key]), 4) for n in (2, 4, 8)
                if "batch_%d" % max(64, n_mine // n) in table}
            pipelined["predicted_strong_scaling_efficiency_plain"] = {
                str(n): round(table["batch_%d" % n_mine]["depth_1_ms"] / (n * table["batch_%d" % max(64, n_mine // n)]["depth_1_ms"]), 4)
                for n in (2, 4, 8) if "batch_%d" % max(64, n_mine // n) in table}
            gm.set_option("pipeline_depth", 1)
        except Exception as e:  # pragma: no cover
            pipelined = {"error": f"{type(e).__name__}: {e}"[:200]}
            try:
                gm.set_option("pipeline_depth", 1)
            except Exception:
                pass

    if args.dump_results and rank == 0:
        np.save(args.dump_results, all_np.view(np.uint8).reshape(-1, 112))
    if rank != 0:
        gm.close()
        ctx.close()
        if distributed:
            dist.destroy_process_group()
        return

    # ---- secondary configs on the GPU (the CPU legs are running beside them in their own processes) --------------
    if do_secondary:
        for name, fn in (("cfg2", gpu_cfg2), ("cfg3", gpu_cfg3), ("cfg5", gpu_cfg5)):
            try:
                sec_gpu[name] = fn(ctx, api, _JOB[name])
            except Exception as e:  # a failed leg is reported, it does not take the headline down
                sec_gpu[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        for leg in ("indoor", "dense"):
            if leg in _JOB:
                try:
                    sec_gpu[leg] = gpu_batch_leg(ctx, api, _JOB[leg])
                except Exception as e:
                    sec_gpu[leg] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if not args.no_next_rows:
            sec_gpu["next_rows"] = gpu_next_rows(ctx, api)
        if not args.no_dropin:
            sec_gpu["dropin"] = dropin_legs(_JOB["cfg5"], args)

    total_matches = n_total * args.steps
    value = total_matches / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # ---- roofline of the dominant kernel (HIP events on the context stream) --------------------------------------
    side = int(np.floor(cfg.search_size / cfg.resolution + 0.5) + 1)
    nx = ny = int(np.floor(0.5 * (side - 1) * 2.0 / 2.0 + 0.5) + 1)
    na_c = int(np.floor(cfg.coarse_search_angle_offset * 2.0 / cfg.coarse_angle_resolution + 0.5) + 1)
    na_f = int(np.floor(0.5 * cfg.coarse_angle_resolution * 2.0 / cfg.fine_search_angle_offset + 0.5) + 1)
    coarse_bytes, match_bytes = algorithmic_bytes(nx, ny, na_c, 3, 3, na_f, N_BEAMS)
    roofline = None
    if dom_name and dom_name in prof:
        launches, total_ms = prof[dom_name]
        avg_ms = total_ms / max(launches, 1)
        per_launch_bytes = (coarse_bytes if dom_name == "resp_rows_coarse" else match_bytes) * n_mine
        hbm_alg = per_launch_bytes / (avg_ms * 1e-3) / 1e9
        rec = {}
        tfile = ROOT / "profiles" / "traffic.json"  # PMC figures per launch of `python bench.py` (profiles/README.md)
        if tfile.exists():
            try:
                rec = json.loads(tfile.read_text()).get(dom_name, {})
            except Exception:
                rec = {}
        # The kernel gathers from an L2-resident 4 MB grid, four candidates per dword, and skips provably-zero rows:
        # HBM is not what bounds it (algorithmic bytes / time EXCEEDS the HBM peak).  What does is VALU issue: the
        # PMC pass counts the wave64 VALU instructions of one launch; the chip issues at most SIMDs*clk/2 per second.
        insts = rec.get("valu_insts_per_launch")
        scans_ref = rec.get("scans_per_launch")
        scale = (n_mine / scans_ref) if scans_ref else None  # same kernel, same per-scan work: PMC figures scale with the scans
        ceil, mix = {}, {}
        try:
            ceil = json.loads((ROOT / "profiles" / "ceilings.json").read_text())
            mix = json.loads((ROOT / "profiles" / "valu_mix.json").read_text()).get(dom_name, {})
        except Exception:
            pass
        # the launch in CLOCKS, live: its HIP-event time x the shader clock the plain leg ran at (lslam_clock_sample around
        # the leg: s_memtime / s_memrealtime) -- no 2.4 GHz assumed anywhere below except in `peak` / `frac`, which keep the
        # guide's convention (2 cycles per wave64 VALU instruction at the peak clock) so that rounds stay comparable
        cycles_live = avg_ms * 1e-3 * clock_ghz * 1e9 if clock_ghz else None
        hbm_frac = round(hbm_alg / HBM_PEAK_GBS, 5)
        meas_hbm = (round(rec["hbm_bytes_per_launch"] * scale / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                    if rec.get("hbm_bytes_per_launch") and scale else None)
        worst = round(n_total / (pruning["ms_per_step_pruning_off"] * 1e-3), 1) if pruning else None
        if insts and scale:
            insts_here = insts * scale
            achieved = insts_here / (avg_ms * 1e-3)
            waves_here = n_mine * na_c  # one wave64 per (scan, coarse angle) from 98 scans up (no beam slices)
            valu_mix_frac = l1_rate = l1_frac = gather_model = None
            if cycles_live and mix.get("floor_cycles_per_simd_w8") and mix.get("scans_per_launch"):
                valu_mix_frac = round(mix["floor_cycles_per_simd_w8"] * n_mine / mix["scans_per_launch"] / cycles_live, 4)
            if cycles_live and rec.get("l1_line_lookups_per_launch"):
                look = rec["l1_line_lookups_per_launch"] * scale
                l1_rate = look / N_CU / cycles_live
                if ceil.get("l1_hit_lookups_per_cu_clk_max"):
                    l1_frac = round(l1_rate / ceil["l1_hit_lookups_per_cu_clk_max"], 4)
                if rec.get("l2_read_requests_per_launch") and ceil.get("l1_hit_lookups_per_cu_clk_max") and ceil.get("l1_miss_lines_per_cu_clk"):
                    miss = rec["l2_read_requests_per_launch"] * scale
                    gather_model = round(((look - miss) / ceil["l1_hit_lookups_per_cu_clk_max"] + miss / ceil["l1_miss_lines_per_cu_clk"])
                                         / N_CU / cycles_live, 4)
            # the contract's keys first, scalars before anything nested (the driver's record keeps the head of the object)
            roofline = {
                "bound": "valu_issue", "kernel": dom_name,
                "frac": round(achieved / VALU_ISSUE_PEAK, 4),
                "achieved": round(achieved / 1e12, 4), "peak": round(VALU_ISSUE_PEAK / 1e12, 4), "unit": "T wave64-VALU-instructions/s",
                "traffic": rec.get("hbm_bytes_per_launch"),
                "hbm_algorithmic_frac": hbm_frac, "measured_hbm_frac": meas_hbm,
                "avg_launch_ms": round(avg_ms, 4),
                "valu_insts_per_wave": round(insts_here / max(waves_here, 1), 1),
                "valu_mix_frac": valu_mix_frac,
                "l1_lookup_frac": l1_frac,
                "worst_case_value": worst,
                "pmc_inputs_stale": None,  # filled in below
                "gather_pipe_model_frac": gather_model,
                "gather_unit_busy": rec.get("gather_unit_busy"),
                "l1_lookups_per_cu_clk": round(l1_rate, 4) if l1_rate else None,
                "l1_miss_ratio": rec.get("l1_miss_ratio"),
                "shader_clock_ghz": round(clock_ghz, 4) if clock_ghz else None,
                "launch_cycles": round(cycles_live, 0) if cycles_live else None,
                "valu_mix_frac_w4": (round(mix["floor_cycles_per_simd_w4"] * n_mine / mix["scans_per_launch"] / cycles_live, 4)
                                     if cycles_live and mix.get("floor_cycles_per_simd_w4") else None),
                "avg_issue_cycles_per_valu_inst": mix.get("avg_cycles_per_valu_inst_w8"),
                "l2_hit": rec.get("l2_hit"),
                "l2_read_frac": (round(rec["l2_read_requests_per_launch"] * 128.0 * scale / (avg_ms * 1e-3) / 1e9 / L2_PEAK_GBS, 4)
                                 if rec.get("l2_read_requests_per_launch") else None),
                "l1_line_lookups_per_vmem_read": rec.get("l1_line_lookups_per_vmem_read"),
                "valu_insts_per_launch": int(insts_here),
                "definitions": {
                    "frac": "wave64 VALU instructions per second of the launch (PMC SQ_INSTS_VALU / live HIP-event time) / (256 CUs x 4 "
                            "SIMDs x 2.4 GHz / 2 cycles): the guide's convention, kept so that rounds compare; it UNDERSTATES how busy the "
                            "issue ports are because only v_fma/mul/add_f32, v_add/sub_u32, v_and/or/xor, v_mov, v_lshrrev issue in ~2 cycles "
                            "-- every VOP3-only opcode (v_perm_b32 ...), DPP / SDWA form, conversion, compare, and ANY instruction with an "
                            "SGPR source takes 4.1 (tools/micro/valu_rate.hip, cycle counters: profiles/r06/micro_valu_rate.txt)",
                    "valu_mix_frac": "the cycles the SIMDs' issue ports NEED for this kernel's instruction mix at their best measured "
                                     "rates (8 waves per SIMD; _w4: at the kernel's own 4) / the launch's cycles (live time x live shader "
                                     "clock) -- tools/valu_mix_floor.py, profiles/valu_mix.json; <= 1 by construction",
                    "l1_lookup_frac": "TCP_TOTAL_CACHE_ACCESSES per CU per clock / the highest rate any tools/micro/ta_rate.hip pattern "
                                      "reaches (profiles/ceilings.json: %s, L1 hits spread over the tag banks)" % ceil.get("l1_hit_lookups_per_cu_clk_max"),
                    "gather_pipe_model_frac": "(L1 hits / %s per clock + L1 misses / %s per clock) / launch cycles per CU: what the CU's "
                                              "address + L1 pipe needs for this kernel's lookups at the measured hit and miss rates; agrees with "
                                              "gather_unit_busy (TA_TA_BUSY per CU per clock of the PMC pass; the saturating micro patterns read "
                                              "%s)" % (ceil.get("l1_hit_lookups_per_cu_clk_max"), ceil.get("l1_miss_lines_per_cu_clk"), ceil.get("ta_busy_max")),
                    "traffic": "HBM bytes per launch, 2 x FETCH_SIZE + WRITE_SIZE of the PMC passes (gfx950 correction of the guide)",
                    "hbm_algorithmic_frac": "SURVEY 8(d) algorithmic bytes / launch time / 8 TB/s: > 1, not a bound for a gather from an "
                                            "L2-resident 4 MB grid with exact zero-row pruning",
                    "worst_case_value": "scan-matches/s with the exact zero-row pruning switched off (every in-range row gathered): the "
                                        "floor over world sparsity, measured in this run",
                },
                "valu_insts_source": rec.get("source", "profiles/traffic.json") + " (rocprofv3 --pmc passes of this command; static "
                                     "properties of kernel + workload, scaled to this launch's scans, not re-measured in this run)",
                "instruction_mix": rec.get("instruction_mix"),
            }
        else:
            roofline = {"bound": "hbm", "kernel": dom_name, "frac": hbm_frac, "achieved": round(hbm_alg, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "traffic": rec.get("hbm_bytes_per_launch"), "hbm_algorithmic_frac": hbm_frac,
                        "measured_hbm_frac": meas_hbm, "avg_launch_ms": round(avg_ms, 4), "worst_case_value": worst}
        # the PMC inputs are static files: tools/make_traffic.py records the hash of the kernel source they were
        # collected from; a different source today means `frac` / `traffic` describe an older kernel
        meta = {}
        try:
            meta = json.loads(tfile.read_text()).get("_meta", {})
        except Exception:
            pass
        # every source the hot kernel is compiled from (scan_matcher.hip includes the .hpp files)
        csrc = ROOT / "creating-2d-laser-slam-from-scratch_amd" / "csrc"
        hashed = meta.get("source_sha256", {})
        need = ("scan_matcher.hip", "common.hpp", "karto_math.hpp", "scan_cache_impl.hpp", "frontend_impl.hpp")
        stale = [f for f in need if hashed.get(f) != _source_sha(csrc / f)]
        roofline["pmc_inputs_stale"] = bool(stale)
        roofline.update({
            "pmc_inputs_stale_files": stale,
            "pmc_inputs_source_sha256": {f: hashed.get(f) for f in need},
            "leg": "plain steps (LSLAM_OPT_PIPELINE_DEPTH 1: one kernel after the other on one stream), %d of them timed like the "
                   "contract's region right after it, HIP events around this kernel only" % n_plain,
            "plain_ms_per_step": round(1e3 * elapsed_plain / n_plain, 4),
            "plain_value": round(n_total * n_plain / elapsed_plain, 1),
            "traffic_source": (rec.get("source", "profiles/traffic.json") + ": rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                               "this command (tools/pmc_passes.sh), per launch of %s scans; NOT measured in this run"
                               % rec.get("scans_per_launch")) if rec.get("hbm_bytes_per_launch") else None,
            # SURVEY 8(d)'s convention, kept for the record: algorithmic bytes / time against the HBM peak.  It is
            # NOT a bound for this kernel (see above); the measured HBM traffic is ~1 % of the algorithmic bytes.
            "hbm_algorithmic": {
                "achieved": round(hbm_alg, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac,
                "algorithmic_bytes_per_launch": per_launch_bytes,
                "whole_match_GBs": round(value / world_size * match_bytes / 1e9, 2),
                "whole_match_frac": round(value / world_size * match_bytes / 1e9 / HBM_PEAK_GBS, 5),
                "measured_hbm_frac": meas_hbm,
            },
        })

    # ---- CPU baseline on this box's host cores (rank 0, N=1 only): collect the worker results -------------------
    cpu_baseline = None
    secondary = None
    if do_cpu:
        c4 = tasks["cfg4"].get()
        GO_MULTI.set()
        sec, kind, sample = c4["sec_per_match"], c4["kind"], len(c4["poses"])
        g = res_np[:sample]  # the same run doubles as a parity check of the timed GPU results
        cpu_baseline = {
            "value": round(1.0 / sec, 2), "unit": "scan-matches/s", "cores": 1, "kind": kind,
            "sample": f"{sample} of the batch's scans (coarse+fine CorrelateScan vs the same shared grid), "
                      f"{sec * sample:.1f} s on 1 of {os.cpu_count()} host cores",
            "max_pose_err_vs_gpu": float(np.abs(g["pose"] - c4["poses"]).max()),
            "max_response_err_vs_gpu": float(np.abs(g["response"] - c4["resp"]).max()),
        }
        if "cfg4_multi" in tasks:
            try:  # all-core figure beside the single-core one; never the headline baseline
                out = [t.get() for t in tasks["cfg4_multi"]]
                busy = max(o[0] for o in out)
                cpu_baseline["multicore"] = {"value": round(sum(o[1] for o in out) / busy, 1), "unit": "scan-matches/s",
                                             "cores": cores, "sample": f"200 scan-matches per process, slowest process {busy:.2f} s"}
                # the same two numbers as scalars (the driver's record of this line keeps scalar keys only)
                cpu_baseline["multicore_value"] = cpu_baseline["multicore"]["value"]
                cpu_baseline["multicore_cores"] = cores
            except Exception as e:  # pragma: no cover
                cpu_baseline["multicore"] = {"error": str(e)[:120]}
    if do_secondary:
        cpu_legs = {k: tasks[k].get() for k in ("cfg2", "cfg3", "cfg5")}
        for leg in ("indoor", "dense"):
            if leg in tasks:
                try:
                    cpu_legs[leg] = [t.get() for t in tasks[leg]]
                except Exception as e:  # pragma: no cover
                    cpu_legs[leg] = {"error": str(e)[:200]}
        secondary = build_secondary(sec_gpu, cpu_legs, _JOB, args)
        secondary["workload_gen_s"] = round(t_gen_secondary, 2)
        # compact copies where the driver's record keeps nested objects
        if roofline is not None:
            roofline["secondary"] = secondary.get("roofline_summary")
        cpu_baseline["secondary"] = secondary.get("cpu_summary")
    if pool is not None:
        pool.close()
        pool.join()

    line = {
        "metric": "scan-matches/sec (1081-beam vs 2000x2000 grid)",
        "value": round(value, 1),
        "unit": "scan-matches/s",
        "n_gpus": world_size,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[3]: batched correlative scan-match, %d independent DISTINCT 1081-beam scans per step "
                        "(%d distinct ray casts; %s scaling: %d per GPU) vs one shared 2005x2005@0.05m uint8 grid (%d-scan "
                        "window), coarse 11x11x21 + fine 3x3x11 + angular covariance, +-0.5 m / +-20 deg"
                        % (n_total, n_unique, args.scaling, n_mine, args.n_base),
            "scans_per_step": n_total, "scans_per_gpu_per_step": n_mine, "distinct_scans": n_unique, "beams": N_BEAMS,
            "grid": [2005, 2005],
            "sharding": "scans [r*B/W,(r+1)*B/W) per rank (SURVEY 8(e)), no data-path collective; all_gather of 112-B "
                        "results timed separately (gather_ms)",
        },
        "results_ok": n_ok_all,
        "gather_ms": None if gather_ms is None else round(gather_ms, 4),
        "per_rank_ms_per_step": per_rank_ms,
        "workload_gen_s": round(t_gen, 2),
        # every kernel of ONE untimed plain step with HIP events around every launch (the events themselves cost ~3 us of
        # stream time per launch, so the sum exceeds a step of the timed region); the dominant kernel's figure in `roofline`
        # is the live average over the plain leg
        "kernel_ms_one_instrumented_plain_step": {k: round(v[1], 4) for k, v in sorted(prof_all.items())},
        "pruning": pruning,
        "pipelined": pipelined,
        "lds_staged_experiment": lds_experiment,
        "value_leg": ("the %d timed steps of the contract (barrier + synchronize on both sides) at LSLAM_OPT_PIPELINE_DEPTH %d%s; "
                      "`pipelined_leg` is the same step with independent steps overlapping on the matcher's internal streams (a "
                      "labelled option, not `value`), `sustained` the timed region's step for >= 10 s"
                      % (args.steps, depth, " = the library's default: one step after the other" if depth == 1 else "")),
        "pipeline_depth": depth,
        "pipelined_records_identical": ring_identical,
        "pipelined_leg": pipelined_leg,
        "pipelined_value": pipelined_leg["value"] if pipelined_leg else None,
        "pipelined_ms_per_step": pipelined_leg["ms_per_step"] if pipelined_leg else None,
        "plain": {"steps": n_plain, "ms_per_step": round(1e3 * elapsed_plain / n_plain, 4),
                  "value": round(n_total * n_plain / elapsed_plain, 1), "unit": "scan-matches/s",
                  "note": "the roofline leg: plain steps again, HIP events around the dominant kernel only"},
        "backend": (backend if distributed else None),
        "rccl_ranks": rccl_ranks,
        "devices": device_names,
        "sustained": sustained,
        "roofline": roofline,
        "cpu_baseline": cpu_baseline,
        "secondary": secondary,
    }
    json_out.write(json.dumps(line) + "\n")
    json_out.flush()
    os.dup2(os.open(os.devnull, os.O_WRONLY), 1)  # the reference library's exit chatter (std::cout) goes nowhere
    gm.close()   # device handles are released while everything they point into is still alive
    ctx.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
