#!/usr/bin/env python3
"""Where the time of ONE complete host-API MatchScan goes (BASELINE config 3): the whole call, AddScans alone, the search alone,
the 605 KB window upload alone."""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lslam
from lslam_amd import api, synth
ctx = api.Context(0)
wl = synth.make_match_workload(n_base=70, n_query=16, seed=4)
gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(wl.laser))
n = gm.num_beams
def t(fn, reps=300):
    fn(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
q = wl.query_ranges[0]; qp = wl.query_poses[0]
print("beams", n, "stride", wl.base_ranges.shape[1])
print("MatchScan (stride 1081):      %.1f us" % t(lambda: gm.MatchScan(q, qp, wl.base_ranges, wl.base_poses)))
b1080 = np.ascontiguousarray(wl.base_ranges[:, :n]); q1080 = np.ascontiguousarray(q[:n])
print("MatchScan (stride == beams):  %.1f us" % t(lambda: gm.MatchScan(q1080, qp, b1080, wl.base_poses)))
print("AddScans alone (upload+prep+rebuild+sync): %.1f us" % t(lambda: gm.AddScans(wl.base_ranges, wl.base_poses, qp)))
print("AddScans, stride == beams:    %.1f us" % t(lambda: gm.AddScans(b1080, wl.base_poses, qp)))
print("match_batch(1) alone:         %.1f us" % t(lambda: gm.match_batch(q[None, :], qp[None, :])))
d = ctx.alloc(b1080.nbytes)
print("upload 605 KB pageable (dev_upload, sync): %.1f us" % t(lambda: ctx.upload(d, b1080)))
