#!/usr/bin/env python3
"""Prices exact candidate-level pruning of the coarse pass (VERDICT r04 item 3 / SURVEY §7's max-pooled pyramid) on the
bench batch, on the CPU, before anything is built.

What the reference's outputs can see of a coarse candidate (x, y, theta):
  * the maximum and its ties within 1e-6 (Mapper.cpp:431-483),
  * ComputePositionalCovariance (Mapper.cpp:535-630): only cells of the max-over-angle grid with response >= best - 0.1.
So a candidate whose (penalised) response is provably < best - 0.1 need not be evaluated exactly.  An upper bound for a
GROUP of candidates is  U = sum_b max_{c in group} grid[pos_c + table[a][b]]  -- one gather per beam from a max-pooled
copy of the grid instead of one per candidate.  This script measures, over a sample of the bench batch:
  (i)   the fraction of the 2 541 candidates with response >= best - 0.1 (nothing can do better than keep those),
  (ii)  the fraction of candidates left after bounds over groups of  {whole 11x11 lattice, 4x4, 2x2 lattice points} per
        angle, with the lower bound L on `best` taken (a) as best itself (optimistic) and (b) as the exact response of
        the lattice's centre cell over all angles (what a first cheap exact pass would know),
and prints the work a two-stage search would do relative to the exhaustive one (gathers per beam x angle).

Uses the plain-C restatement (oracle/karto_oracle.c) for the lookup tables and the exhaustive sums: test infrastructure
pricing a design, not product code.  ~1 minute for 128 scans.
"""
from __future__ import annotations

import argparse
import json
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import lslam  # noqa: E402,F401
from lslam_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

sys.path.insert(0, str(ROOT))
import bench  # noqa: E402  (query_poses / cast_scans: the bench batch itself)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=128)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    po.build("restate")
    laser, world = synth.Laser(), synth.arena()
    wl = synth.make_match_workload(n_base=70, n_query=1, seed=5, query_spread=3.0, world=world)
    n = args.scans
    truth = bench.query_poses(world, wl.center_pose, 4096, 3.0, seed=55)[:n]
    odom = synth.perturb(bench.query_poses(world, wl.center_pose, 4096, 3.0, seed=55), 0.3, np.deg2rad(10.0), 77)[:n]
    ranges = synth.ranges_to_f64(bench.cast_scans(world, laser, truth, 0, 555, 8))
    cfg = po.default_cfg()
    port = po.PortKarto(cfg, po.laser_struct(laser))
    port.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
    grid = port.grid().reshape(-1).astype(np.uint8)
    gi = port.grid_info()
    stride, border = gi["stride"], gi["roi_x"]
    res = cfg.resolution
    side = int(np.floor(cfg.search_size / res + 0.5) + 1)
    off = 0.5 * (side - 1) * res
    nx = int(np.floor(off * 2.0 / (2 * res) + 0.5) + 1)
    ang_off, ang_res = cfg.coarse_angle_offset, cfg.coarse_angle_resolution
    na = int(np.floor(ang_off * 2.0 / ang_res + 0.5) + 1)
    N = port.num_beams
    denom = N * 100.0
    lat = -off + np.arange(nx) * (2 * res)
    sd = lat[:, None] ** 2 + lat[None, :] ** 2  # [y, x]
    dp = np.maximum(1.0 - 0.2 * sd / cfg.distance_variance_penalty, cfg.minimum_distance_penalty)
    gpad = np.concatenate([grid, np.zeros(1, np.uint8)])  # index data_size = "outside" -> 0
    data_size = grid.size

    def groups(k):
        edges = list(range(0, nx, k)) + [nx]
        return [(a, b) for a, b in zip(edges[:-1], edges[1:])]

    acc = {"cand_ge": [], "cells_ge": [], "best": []}
    for mode in ("best", "centre"):
        for k in (nx, 4, 2):
            acc[f"kept_{mode}_g{k}"] = []
            acc[f"groups_kept_{mode}_g{k}"] = []
    for s in range(n):
        pose = odom[s]
        table = port.compute_offsets(ranges[s], pose, pose[2], ang_off, ang_res).astype(np.int64)  # [na, N]
        invalid = table == np.iinfo(np.int32).max
        # lattice cell flat indices (Mapper.cpp:385-388)
        gx = np.floor((pose[0] + lat - gi["offset"][0]) / res + 0.5).astype(np.int64) + border
        gy = np.floor((pose[1] + lat - gi["offset"][1]) / res + 0.5).astype(np.int64) + border
        pos = gx[None, :] + gy[:, None] * stride  # [y, x]
        idx = pos[None, None, :, :] + table[:, :, None, None]  # [na, N, y, x]
        bad = (idx < 0) | (idx >= data_size) | invalid[:, :, None, None]
        idx = np.where(bad, data_size, idx)
        V = gpad[idx]  # uint8 [na, N, y, x]
        sums = V.sum(axis=1, dtype=np.int64)  # [na, y, x]
        angle = (pose[2] - ang_off) + np.arange(na) * ang_res
        apn = np.maximum(1.0 - 0.2 * (angle - pose[2]) ** 2 / cfg.angle_variance_penalty, cfg.minimum_angle_penalty)
        resp = sums / denom
        resp = np.where(resp == 0.0, 0.0, resp * (dp[None] * apn[:, None, None]))
        best = resp.max()
        thr = best - 0.1
        acc["best"].append(best)
        acc["cand_ge"].append(float((resp >= thr).mean()))
        acc["cells_ge"].append(float((resp.max(axis=0) >= thr).mean()))
        centre_L = resp[:, nx // 2, nx // 2].max()
        for mode, L in (("best", best), ("centre", centre_L)):
            for k in (nx, 4, 2):
                kept = 0
                gk = 0
                gtot = 0
                for (y0, y1) in groups(k):
                    for (x0, x1) in groups(k):
                        U = V[:, :, y0:y1, x0:x1].max(axis=(2, 3)).sum(axis=1, dtype=np.int64) / denom  # [na]
                        U = U * apn * dp[y0:y1, x0:x1].max()
                        alive = U >= L - 0.1
                        kept += int(alive.sum()) * (y1 - y0) * (x1 - x0)
                        gk += int(alive.sum())
                        gtot += na
                acc[f"kept_{mode}_g{k}"].append(kept / (na * nx * nx))
                acc[f"groups_kept_{mode}_g{k}"].append(gk / gtot)
    out = {"scans": n, "lattice": [nx, nx, na], "beams": N,
           "mean_best_response": float(np.mean(acc["best"])),
           "candidates_with_response_ge_best_minus_0.1": float(np.mean(acc["cand_ge"])),
           "cells_with_max_over_angle_ge_best_minus_0.1": float(np.mean(acc["cells_ge"]))}
    for mode in ("best", "centre"):
        for k in (nx, 4, 2):
            ng = len(groups(k)) ** 2
            kept = float(np.mean(acc[f"kept_{mode}_g{k}"]))
            out[f"L={mode},group={k}x{k}"] = {
                "candidates_kept": round(kept, 4),
                "groups_kept": round(float(np.mean(acc[f"groups_kept_{mode}_g{k}"])), 4),
                # gathers per beam x angle: one per group for the bound pass + the kept candidates' exact gathers, against
                # nx*nx for the exhaustive pass (before row pruning, which both would keep)
                "relative_gather_work": round((ng + kept * nx * nx) / (nx * nx), 4)}
    print(json.dumps(out, indent=1))
    if args.out:
        pathlib.Path(args.out).write_text(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
