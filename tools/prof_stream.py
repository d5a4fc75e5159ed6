import sys, math, time, numpy as np
sys.path.insert(0, '.')
import lslam
from lslam_amd import api, synth
ctx = api.Context(0)
laser = synth.Laser()
world = synth.arena(size=100.0, n_axis=30, n_rot=10, seed=6)
n_scans = 300
path = synth.trajectory(world, n_scans, step=0.25, seed=6, bounds=40.0)
odom = synth.perturb(path, 0.05, math.radians(2.0), 7)
rng = np.random.default_rng(8)
scans32 = [synth.cast_scan(world, t, laser, 0.01, 0.01, rng) for t in path]
gm = api.ScanMatcher(ctx, api.baseline_config(), api.laser_params(laser))
fe = api.FrontEnd(gm, scan_buffer_size=70, scan_buffer_max_distance=20.0)
r64 = [synth.ranges_to_f64(r) for r in scans32]
for r, o in zip(r64[:100], odom[:100]): fe.Process(r, o)
ctx.synchronize()
t0 = time.perf_counter()
for r, o in zip(r64[100:], odom[100:]): fe.Process(r, o)
ctx.synchronize()
print("plain: %.1f us/scan" % ((time.perf_counter() - t0) / 200 * 1e6))
fe2 = api.FrontEnd(gm, scan_buffer_size=70, scan_buffer_max_distance=20.0)
for r, o in zip(r64[:100], odom[:100]): fe2.Process(r, o)
ctx.profile(True); ctx.profile_reset()
t0 = time.perf_counter()
for r, o in zip(r64[100:], odom[100:]): fe2.Process(r, o)
ctx.synchronize()
print("profiled: %.1f us/scan" % ((time.perf_counter() - t0) / 200 * 1e6))
pr = ctx.profile_read()
tot = 0
for k, (n, ms) in sorted(pr.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:24s} launches/scan {n/200:5.2f}  us/scan {ms*1e3/200:7.1f}"); tot += ms
print("kernel total us/scan %.1f" % (tot * 1e3 / 200))
