#!/usr/bin/env python3
"""Where INSIDE a kernel the time goes: runs a workload against a DIAGNOSTIC build of the library (-DLSLAM_PHASE_STAMPS:
csrc/common.hpp's PhaseClock; waves add s_memtime deltas between phase marks into a [kernel][phase] table) and prints the
table -- cycles per visit, visits, and the share of each phase in its kernel.

  python tools/ab_variants.py build stamps:-DLSLAM_PHASE_STAMPS        # here (hipcc cross-compiles)
  LSLAM_GPU_LIB=$PWD/creating-2d-laser-slam-from-scratch_amd/_variants/stamps.so python tools/phase_stamps.py rays   # on the GPU box
  ... phase_stamps.py lone    # the single-scan MatchScan chain (streaming front-end, 600 scans)
"""
import ctypes as C
import json
import math
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import lslam  # noqa: E402,F401
from lslam_amd import api, synth  # noqa: E402

MAP_KERNELS = {0: ("k_lo_batch_rays", ["scan header", "beam line (point load + transform)", "cell addresses (closed-form Bresenham)",
                                       "plane bytes back (loads)", "marks + flags issued"])}
MATCHER_KERNELS = {
    0: ("k_find_valid (base-scan blocks)", ["world points of the scan into LDS", "anchors' side tests, runs marked valid",
                                            "valid[] out, cells tagged in the mark plane"]),
    1: ("k_smear_gather", ["smear kernel into LDS", "own bytes + neighbour rows of the mark plane", "grid + parity planes stored"]),
    2: ("k_anchor_chain", ["world points (fp64 sincos per beam)", "successor of every point", "pointer doubling",
                           "ordered compaction of the anchors"]),
    3: ("reduce_coarse_lds_block", ["penalty tables, cells, clears", "numerators -> responses (division), cell maxima", "block maximum",
                                    "probabilities merged, tie candidates, mask words", "ordered tie average (thread 0)",
                                    "fine lattice | covariance terms compacted", "ordered covariance sums + record (thread 0)"]),
    4: ("reduce_fine_block", ["penalised responses", "block maximum", "tie mask + ordered tie average (thread 0)",
                              "angular-covariance numerators", "angular covariance + record + ticket (thread 0)"]),
    5: ("resp_rows_wave (one wave per block)", ["lattice record, cos/sin", "phases A + B", "wave reduction + stores"]),
    6: ("scan_prep_block", ["first world point | transform (thread 0) | lattice (last wave)", "barrier: the slowest of the three",
                            "scan-frame points"]),
    7: ("k_match_lone (thread 0 of every block)", ["coarse tasks", "block barrier, release fence, arrival", "coarse reduce (last block) | wait + acquire",
                                                    "fine tasks", "block barrier, release fence, arrival"]),
}


def lone(ctx):
    """The streamed single-scan chain: lslam_frontend_process, pose graph bookkeeping on, loop closing off, 600 scans."""
    import bench

    L = api.lib()
    L.lslam_debug_matcher_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    n = 600
    laser = synth.Laser()
    path = synth.rings_trajectory(n)
    world = synth.arena_around_path(path, size=100.0, n_axis=30, n_rot=10, seed=6)
    odom = synth.drifting_odometry(path, scale=1.01, sigma_xy=0.004, sigma_th=0.0015, seed=6)
    import os

    scans32 = bench.cast_scans(world, laser, path, 0, 6, max(1, min(32, os.cpu_count() or 1)))
    r64 = [synth.ranges_to_f64(r) for r in scans32]
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(laser))
    if os.environ.get("LSLAM_LONE_KERNEL"):
        gm.set_option("lone_kernel", int(os.environ["LSLAM_LONE_KERNEL"]))
    fe = api.FrontEnd(gm)
    for r, o in zip(r64[:80], odom[:80]):
        fe.Process(r, o)
    fe.reset()
    ctx.synchronize()
    read(ctx, L.lslam_debug_matcher_stamps)
    import time

    t0 = time.perf_counter()
    for r, o in zip(r64, odom):
        fe.Process(r, o)
    ctx.synchronize()
    wall = time.perf_counter() - t0
    cyc, vis = read(ctx, L.lslam_debug_matcher_stamps)
    return {"workload": "streaming front-end, 600 scans, 70-scan window, loop closing off (diagnostic build: the stamps cost time)",
            "us_per_scan_instrumented": round(1e6 * wall / n, 1), "scans": n, "kernels": table(cyc, vis, MATCHER_KERNELS)}


def read(ctx, fn, reset=True):
    out = (C.c_ulonglong * 128)()
    ctx.check(fn(ctx.h, out, int(reset)))
    a = np.frombuffer(out, dtype=np.uint64).reshape(8, 16).astype(np.float64)
    return a[:, :8], a[:, 8:]


def table(cyc, vis, names, clock_ghz=None):
    rows = {}
    for k, (kname, phases) in names.items():
        tot = cyc[k].sum()
        if tot == 0:
            continue
        rows[kname] = {"phases": [{"phase": phases[i] if i < len(phases) else f"phase {i}", "visits": int(vis[k][i]),
                                   "cycles_per_visit": round(cyc[k][i] / max(vis[k][i], 1), 1),
                                   "share": round(cyc[k][i] / tot, 4)} for i in range(8) if vis[k][i] > 0],
                       "cycles_total": tot}
    return rows


def rays(ctx):
    L = api.lib()
    L.lslam_debug_map_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    laser = synth.Laser()
    n, cell = 1000, 0.05
    off = (n * cell * 0.5, n * cell * 0.5)
    world = synth.arena(size=44.0, n_axis=12, n_rot=4, seed=3)
    rng = np.random.default_rng(3)
    poses = []
    while len(poses) < 256:
        x, y = rng.uniform(-4, 4, 2)
        if synth.point_is_free(world, x, y, 0.8):
            poses.append((x, y, rng.uniform(-math.pi, math.pi)))
    scans = [(synth.hector_points(synth.cast_scan(world, p, laser, 0.01, 0.01, rng), laser, 1.0 / cell, use_max=20.0),
              np.asarray(p, dtype=np.float32)) for p in poses]
    gmap = api.OccGridMap(ctx, n, n, cell, off)
    gmap.setUpdateOccupiedFactor(0.9)
    pts = [p for p, _ in scans]
    ps = np.stack([q for _, q in scans])
    gmap.updateByScans(pts[:64], (0.0, 0.0), ps[:64])
    ctx.synchronize()
    read(ctx, L.lslam_debug_map_stamps)
    c0 = ctx.clock_sample()
    for k in range(0, 256, 64):
        gmap.updateByScans(pts[k:k + 64], (0.0, 0.0), ps[k:k + 64])
    ctx.synchronize()
    c1 = ctx.clock_sample()
    cyc, vis = read(ctx, L.lslam_debug_map_stamps)
    return {"workload": "cfg 2 batched: 4 calls x 64 scans into 1000x1000@0.05", "shader_clock_ghz": ctx.clock_ghz(c0, c1),
            "kernels": table(cyc, vis, MAP_KERNELS)}


def batch(ctx, n_scans=4096, steps=10):
    """The headline step: `n_scans` distinct scans vs the shared grid (bench.py's workload), plain steps."""
    import bench
    import torch

    L = api.lib()
    L.lslam_debug_matcher_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    laser, world = synth.Laser(), synth.arena()
    wl = synth.make_match_workload(n_base=70, n_query=1, seed=5, query_spread=3.0, world=world)
    truth = bench.query_poses(world, wl.center_pose, n_scans, 3.0, seed=55)
    odom = synth.perturb(truth, 0.3, np.deg2rad(10.0), 77)
    ranges = bench.cast_scans(world, laser, truth, 0, 555, 8)
    gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(laser))
    gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
    dev = torch.device("cuda", 0)
    r = torch.from_numpy(np.ascontiguousarray(ranges)).to(dev)
    p = torch.from_numpy(np.ascontiguousarray(odom)).to(dev)
    out = torch.empty((n_scans, 112), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for _ in range(3):
        gm.match_batch_dev(n_scans, r.data_ptr(), r.shape[1], p.data_ptr(), out.data_ptr(), dtype="f32")
    ctx.synchronize()
    read(ctx, L.lslam_debug_matcher_stamps)
    c0 = ctx.clock_sample()
    for _ in range(steps):
        gm.match_batch_dev(n_scans, r.data_ptr(), r.shape[1], p.data_ptr(), out.data_ptr(), dtype="f32")
    ctx.synchronize()
    c1 = ctx.clock_sample()
    cyc, vis = read(ctx, L.lslam_debug_matcher_stamps)
    return {"workload": "bench.py's step: %d distinct scans vs the shared 2005^2 grid, %d plain steps (diagnostic build)" % (n_scans, steps),
            "shader_clock_ghz": ctx.clock_ghz(c0, c1), "kernels": table(cyc, vis, MATCHER_KERNELS)}


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "rays"
    import torch  # (initialised before the library's own HIP context, like bench.py and the tests do)

    torch.cuda.init()
    ctx = api.Context(0)
    out = rays(ctx) if what == "rays" else batch(ctx, int(sys.argv[2]) if len(sys.argv) > 2 else 4096) if what == "batch" else lone(ctx)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
