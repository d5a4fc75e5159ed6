"""Where the lesson4 loop (matchData + updateByScan per scan, 3-level 1024^2 pyramid) spends its time: loop / match-only
microseconds per scan and the HIP-event time of every kernel.  LSLAM_GN_THREADS=256|512|1024, LSLAM_GN_ORDERED=1 select the
Gauss-Newton kernel variant.  Usage (GPU box): python tools/gn_profile.py"""
import sys, time, numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent))
import lslam
from lslam_amd import api, synth
ctx = api.Context(0)
laser = synth.Laser()
n, cell, levels = 1024, 0.05, 3
off = (n * cell * 0.5, n * cell * 0.5)
world = synth.arena(size=40.0, n_axis=10, n_rot=4, seed=3)
N = 200
path = synth.trajectory(world, N, step=0.05, seed=3, bounds=6.0)
rng = np.random.default_rng(1)
pts_all = [synth.hector_points(synth.cast_scan(world, t, laser, 0.01, 0.0, rng), laser, 1.0 / cell, use_max=20.0) for t in path]
hints = [(t + np.array([0.05, -0.04, 0.02])).astype(np.float32) for t in path]
gpu = api.OccGridMap(ctx, n, n, cell, off, levels=levels)
gpu.setUpdateOccupiedFactor(0.9)
def loop(update=True):
    t0 = time.perf_counter()
    for k, (pts, hint) in enumerate(zip(pts_all, hints)):
        pose = gpu.matchData(path[0].astype(np.float32) if k == 0 else hint, pts)[0]
        if update: gpu.updateByScan(pts, (0.0, 0.0), pose)
    ctx.synchronize()
    return time.perf_counter() - t0
loop(); gpu.reset()
print("loop us/scan", 1e6 * loop() / N)
print("match only us/scan (map kept)", 1e6 * loop(False) / N)
gpu.reset()
ctx.profile(True); ctx.profile_only(None); ctx.profile_reset()
loop()
ctx.profile(False)
for k, v in sorted(ctx.profile_read().items(), key=lambda kv: -kv[1][1]):
    print("%-20s launches %5d  us/scan %8.2f  us/launch %8.2f" % (k, v[0], 1e3 * v[1] / N, 1e3 * v[1] / max(v[0], 1)))
