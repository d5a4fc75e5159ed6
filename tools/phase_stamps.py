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
MATCHER_KERNELS = {}


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


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "rays"
    ctx = api.Context(0)
    out = rays(ctx) if what == "rays" else lone(ctx)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
