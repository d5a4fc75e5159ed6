#!/usr/bin/env python3
"""tests/golden/karto_cfg5_golden.npz: the REFERENCE's own karto::Mapper::Process (oracle/_ref, compiled unmodified from
/root/reference) over the first N (default: all) scans of BASELINE config 5's 10 000-scan closed-loop trajectory -- the
workload tools/bench_extra.py uses (synth.rings_trajectory(10000), 100 m arena seed 6, drifting odometry seed 6, ray
casts seeded per scan [6, i]) -- with the pose graph and loop closing on: corrected pose of every scan when it was
processed, edge count after every scan, final poses of all vertices.

    python tests/golden/make_cfg5_golden.py [N]        # N = 10000: ~39 min of one CPU core (3600: ~6 min)

The ranges themselves (N x 1081 float32 = 15 MB) are NOT stored: the generator is deterministic, the test regenerates
them and checks their SHA-256 against the one recorded here before comparing anything.
"""
import hashlib
import os
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import lslam  # noqa: E402,F401
import bench  # noqa: E402  (the pooled numpy ray caster)
from lslam_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

OUT = pathlib.Path(__file__).resolve().parent
GRAPH = dict(scan_buffer_size=70, scan_buffer_max_scan_distance=20.0, do_loop_closing=1, link_scan_maximum_distance=1.5,
             loop_search_maximum_distance=3.0, loop_match_minimum_chain_size=10)


def workload(n_total=10000, n=10000):
    laser = synth.Laser()
    path = synth.rings_trajectory(n_total)
    world = synth.arena_around_path(path, size=100.0, n_axis=30, n_rot=10, seed=6)
    odom = synth.drifting_odometry(path, scale=1.01, sigma_xy=0.004, sigma_th=0.0015, seed=6)
    scans32 = bench.cast_scans(world, laser, path[:n], 0, 6, max(1, min(32, os.cpu_count() or 1)))
    return laser, path[:n], odom[:n], scans32


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    po.build("ref")
    assert po.have_ref(), "needs /root/reference to build oracle/_ref"
    laser, path, odom, scans32 = workload(n=n)
    ref = po.RefKarto(po.default_cfg(**GRAPH), po.laser_struct(laser))
    poses = np.zeros((n, 3))
    processed = np.zeros(n, dtype=np.uint8)
    edges = np.zeros(n, dtype=np.int32)
    t0 = time.perf_counter()
    for i in range(n):
        ok, poses[i] = ref.process(synth.ranges_to_f64(scans32[i]), odom[i])
        processed[i] = ok
        edges[i] = ref.graph_stats()[1]
        if i % 500 == 499:
            print(i + 1, "scans", round(time.perf_counter() - t0, 1), "s, edges", edges[i], flush=True)
    nv = ref.graph_stats()[0]
    final = np.stack([ref.scan_pose(i) for i in range(nv)])
    np.savez_compressed(OUT / "karto_cfg5_golden.npz", n=n, ranges_sha256=hashlib.sha256(scans32.tobytes()).hexdigest(),
                        corrected=poses, processed=processed, edges=edges, final_poses=final,
                        cpu_seconds=time.perf_counter() - t0, graph=np.array(sorted(GRAPH.items()), dtype=object).astype(str))
    print("wrote", OUT / "karto_cfg5_golden.npz", "edges", int(edges[-1]), "vertices", nv)


if __name__ == "__main__":
    main()
