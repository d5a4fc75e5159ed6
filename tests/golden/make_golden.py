#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE ITSELF (oracle/_ref = the reference's open_karto and
lesson4 hector_mapping headers compiled unmodified from /root/reference; only possible in the build container).

    python tests/golden/make_golden.py

karto_match_golden.npz : seeded base window + queries (float32 ranges as a LaserScan carries them)
    and what karto::ScanMatcher produced for them: full MatchScan per query, shared-grid
    CorrelateScan results, the shared correlation grid (sparse), coarse lookup-table and
    search-space-probability digests.
karto_frontend_golden.npz : a 30-scan trajectory through karto::Mapper::Process (corrected poses).
karto_loop_golden.npz : a 190-scan CLOSED-LOOP trajectory (a rectangle driven 1.5 times, drifting odometry) through
    karto::Mapper::Process with loop closing on: corrected pose per scan, graph edge count after each scan, final
    poses of all vertices.
hector_golden.npz : from the reference's own lesson4 hector_mapping headers, compiled unmodified as
    oracle/_ref/libhector_ref.so (Eigen from oracle/shim/Eigen): the log-odds map of 4 scans through
    OccGridMapBase::updateByScan, and a 40-scan run of HectorSlamProcessor::update on a 3-level pyramid
    (start estimates, matched poses, Hessians, which scans updated the map, per-level map digests).
"""
import hashlib
import math
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import lslam  # noqa: E402,F401
from lslam_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

OUT = pathlib.Path(__file__).resolve().parent


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    po.build("all")
    assert po.have_ref(), "needs /root/reference to build oracle/_ref"
    laser = synth.Laser()
    wl = synth.make_match_workload(n_base=10, n_query=6, seed=31, query_spread=1.5)
    cfg, ls = po.default_cfg(), po.laser_struct(laser)
    ref = po.RefKarto(cfg, ls)
    full = [ref.match(wl.base_ranges, wl.base_poses, wl.query_ranges[q], wl.query_poses[q]) for q in range(6)]
    ref.set_base_scans(wl.base_ranges, wl.base_poses, wl.center_pose)
    grid = ref.grid()
    nz = np.flatnonzero(grid.reshape(-1)).astype(np.int32)
    _, s_poses, s_covs, s_resp = ref.match_fixed_grid(wl.query_ranges, wl.query_poses)
    # coarse tables / probs of query 0 against the shared grid (coarse pass only)
    ref.match_fixed_grid(wl.query_ranges[:1], wl.query_poses[:1], True, False)
    tables, _ = ref.tables()
    probs = ref.probs()
    np.savez_compressed(
        OUT / "karto_match_golden.npz",
        base_ranges=wl.base_ranges.astype(np.float32), base_poses=wl.base_poses,
        query_ranges=wl.query_ranges.astype(np.float32), query_poses=wl.query_poses, center_pose=wl.center_pose,
        full_pose=np.stack([f[0] for f in full]), full_cov=np.stack([f[1] for f in full]),
        full_resp=np.array([f[2] for f in full]),
        shared_pose=s_poses, shared_cov=s_covs, shared_resp=s_resp,
        grid_nz_index=nz, grid_nz_value=grid.reshape(-1)[nz], grid_offset=ref.grid_info()["offset"],
        grid_sha256=np.array(sha(grid)), coarse_table_q0_sha256=np.array(sha(tables)),
        coarse_table_q0_head=tables[:, :16].copy(), probs_q0=probs,
    )
    # streaming front-end
    ref2 = po.RefKarto(po.default_cfg(scan_buffer_size=8, scan_buffer_max_scan_distance=3.0), ls)
    world = synth.arena()
    path = synth.trajectory(world, 30, step=0.15, seed=13)
    odom = synth.perturb(path, 0.04, math.radians(1.5), 14)
    rng = np.random.default_rng(15)
    ranges, processed, poses = [], [], []
    for t, o in zip(path, odom):
        r = synth.cast_scan(world, t, laser, 0.01, 0.01, rng)
        ok, pose = ref2.process(synth.ranges_to_f64(r), o)
        ranges.append(r); processed.append(ok); poses.append(pose)
    np.savez_compressed(OUT / "karto_frontend_golden.npz", ranges=np.stack(ranges), odom=odom,
                        processed=np.array(processed), corrected=np.stack(poses))
    # closed-loop trajectory through karto::Mapper::Process WITH its pose graph: LinkNearChains on the revisits and
    # TryCloseLoop (no solver attached, like the library itself)
    lp_cfg = dict(scan_buffer_size=30, scan_buffer_max_scan_distance=8.0, do_loop_closing=1,
                  link_scan_maximum_distance=1.5, loop_search_maximum_distance=3.0)
    ref3 = po.RefKarto(po.default_cfg(**lp_cfg), po.laser_struct(laser, 20.0))
    lworld = synth.arena(size=40.0, n_axis=10, n_rot=4, seed=5)
    lpath = synth.loop_trajectory(190)
    lodom = synth.drifting_odometry(lpath)
    lranges, lproc, lposes, ledges = [], [], [], []
    for i, (t, o) in enumerate(zip(lpath, lodom)):
        r = synth.cast_scan(lworld, t, laser, 0.01, 0.01, np.random.default_rng([41, i]))
        ok, pose = ref3.process(synth.ranges_to_f64(r), o)
        lranges.append(r); lproc.append(ok); lposes.append(pose); ledges.append(ref3.graph_stats()[1])
    nv, ne = ref3.graph_stats()
    assert ne > nv + 2, (nv, ne)  # the revisit added links beyond one per scan
    final = np.stack([ref3.scan_pose(i) for i in range(nv)])
    np.savez_compressed(OUT / "karto_loop_golden.npz", ranges=np.stack(lranges), odom=lodom, truth=lpath,
                        processed=np.array(lproc), corrected=np.stack(lposes), edges=np.array(ledges),
                        final_poses=final, range_threshold=np.float64(20.0),
                        **{"cfg_" + k: np.float64(v) for k, v in lp_cfg.items()})
    # hector: from the reference's own hector_mapping headers (oracle/_ref/libhector_ref.so)
    assert po.have_ref_hector(), "needs /root/reference to build oracle/_ref/libhector_ref.so"
    n, cell = 600, 0.05
    off = (n * cell * 0.5, n * cell * 0.5)
    hm = po.RefHector(n, n, cell, off)
    hm.setUpdateOccupiedFactor(0.9)
    hworld = synth.arena(size=28.0, n_axis=6, n_rot=3, seed=17)
    hpath = synth.trajectory(hworld, 4, step=0.5, seed=17, bounds=4.0)
    hr = np.stack([synth.cast_scan(hworld, p, laser) for p in hpath])
    for r, p in zip(hr, hpath):
        hm.updateByScan(synth.hector_points(r, laser, 1.0 / cell, use_max=14.0), (0.0, 0.0), p.astype(np.float32))
    lo = hm.logodds()
    hnz = np.flatnonzero(lo.reshape(-1)).astype(np.int32)
    # lesson4 loop through HectorSlamProcessor::update: 3-level 1024^2 pyramid, 40 scans
    pn, LV = 1024, 3
    proc = po.RefHectorProcessor(cell, pn, pn, (0.5, 0.5), LV, p_free=0.4, p_occ=0.9)
    est = np.zeros(3, np.float32)
    p_ranges, p_start, p_pose, p_cov, p_upd = [], [], [], [], []
    pworld = synth.arena(size=40.0, n_axis=10, n_rot=4, seed=5)
    for k in range(40):
        truth = (0.04 * k, 0.015 * k, 0.004 * k)
        r = synth.cast_scan(pworld, truth, laser)
        pts = synth.hector_points(r, laser, 1.0 / cell, use_max=20.0)
        p_start.append(est.copy())
        p_upd.append(proc.update(pts, est))
        est, cov = proc.last_pose()
        p_ranges.append(r); p_pose.append(est.copy()); p_cov.append(cov.copy())
    plo = [proc.logodds(lv) for lv in range(LV)]
    np.savez_compressed(OUT / "hector_golden.npz", ranges=hr, poses=hpath.astype(np.float32), size=np.array([n, n]),
                        cell=np.float32(cell), offset=np.array(off, dtype=np.float32), use_max=np.float32(14.0),
                        nz_index=hnz, nz_value=lo.reshape(-1)[hnz], occupancy_sha256=np.array(sha(hm.occupancy_i8())),
                        proc_size=np.array([pn, LV]), proc_use_max=np.float32(20.0), proc_ranges=np.stack(p_ranges),
                        proc_start=np.stack(p_start), proc_pose=np.stack(p_pose), proc_cov=np.stack(p_cov),
                        proc_updated=np.array(p_upd), proc_logodds_sha256=np.array([sha(a) for a in plo]),
                        proc_nonzero=np.array([np.count_nonzero(a) for a in plo]),
                        proc_truth_last=np.array(truth, np.float32))
    assert np.abs(est - np.array(truth)).max() < 0.05, (est, truth)
    for f in sorted(OUT.glob("*.npz")):
        print(f.name, f.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
