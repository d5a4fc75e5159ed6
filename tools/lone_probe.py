#!/usr/bin/env python3
"""One match of one scan through k_match_lone, timed, with the hand-over words dumped (diagnostics)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lslam  # noqa: E402,F401
from lslam_amd import api, synth  # noqa: E402

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
wl = synth.make_match_workload(n_base=30, n_query=8, seed=61, query_spread=2.0)
ctx = api.Context(0)
gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(wl.laser))
gm.AddScans(wl.base_ranges, wl.base_poses, wl.center_pose)
want = [gm.match_batch(wl.query_ranges[i:i + 1], wl.query_poses[i:i + 1]) for i in range(n)]
gm.set_option("lone_kernel", mode)
L = api.lib()
L.lslam_debug_lone_sync.argtypes = [C.c_void_p, C.c_void_p]
for i in range(n):
    t0 = time.perf_counter()
    got = gm.match_batch(wl.query_ranges[i:i + 1], wl.query_poses[i:i + 1])
    dt = time.perf_counter() - t0
    ring = np.zeros(128, dtype=np.uint32)
    rc = L.lslam_debug_lone_sync(gm.h, ring.ctypes.data)
    print(f"mode {mode} match {i}: {dt * 1e3:.2f} ms status {int(got['status'][0])} equal {got.tobytes() == want[i].tobytes()} rc {rc}", flush=True)
    print("  ring", ring.reshape(16, 8)[:4, :8].tolist(), flush=True)
