"""lesson5 lidar motion de-skew on the device (LidarUndistortion::CorrectLaserScan,
lesson5/src/lidar_undistortion.cc:339-447) against the numpy restatement of the same statements (tests/deskew_restatement.py), which tests/test_deskew_pin.py pins
against the reference's own compiled lidar_undistortion.cc (bit-equal on 99.97-100 % of the coordinates, <= 4.8e-7 m); the
same file compares the device path with the reference directly.  The float32 cos/sin of the Euler angles come from the
device libm, so the comparison carries a 2e-6 m tolerance."""
import math

import numpy as np
import pytest

from lslam_amd import api, synth

pytestmark = pytest.mark.gpu
f32 = np.float32


from deskew_restatement import restated_deskew  # noqa: E402


def test_deskew_matches_the_restated_reference(ctx):
    laser = synth.Laser()
    world = synth.arena(size=40.0, n_axis=10, n_rot=4, seed=5)
    r = synth.cast_scan(world, (0.3, -0.2, 0.1), laser, 0.01, 0.02, np.random.default_rng(3))
    r[7] = np.float32("nan")
    r[0] = np.float32(0.01)  # the first beams are invalid: the reference anchors on the first VALID beam
    t0, dur = 1000.0, 0.1
    p = api.DeskewParams(laser.angle_min, laser.angle_increment, laser.range_min, 30.0, t0, dur / len(r), 1, 1,
                         t0 - 0.004, t0 + dur + 0.006, 0.05, 0.012, 0.0, 0.0)
    # integrated gyro samples (lidar_undistortion.cc:205-243): 100 Hz, first one just before the scan starts
    times = [t0 - 0.003 + 0.01 * k for k in range(11)]
    gyro = np.array([[0.02 * math.sin(k), -0.03, 0.6 + 0.05 * k] for k in range(11)])
    rot = [[0.0, 0.0, 0.0]]
    for k in range(1, 11):
        rot.append(list(np.array(rot[-1]) + gyro[k] * (times[k] - times[k - 1])))
    want, want_valid = restated_deskew(r, p, times, rot)
    got, got_valid = api.deskew_scan(ctx, r, p, times, rot)
    assert np.array_equal(got_valid, want_valid) and want_valid.sum() > 500 and not want_valid[0] and not want_valid[7]
    assert np.abs(got - want).max() <= 2e-6
    assert np.all(got[~want_valid] == 0)
    # what it is for: with a pure yaw rate the correction of beam i is a rotation by yaw(t_i) - yaw(t_first)
    p2 = api.DeskewParams(laser.angle_min, laser.angle_increment, laser.range_min, 30.0, t0, dur / len(r), 1, 0,
                          0.0, 1.0, 0.0, 0.0, 0.0, 0.0)
    rot2 = [[0.0, 0.0, 0.5 * (t - times[0])] for t in times]
    got2, v2 = api.deskew_scan(ctx, r, p2, times, rot2)
    first = int(np.flatnonzero(v2)[0])
    i = int(np.flatnonzero(v2 & (np.arange(len(r)) < 900))[-1])  # inside the IMU samples' time span (no clamping at the end)
    yaw = 0.5 * ((t0 + i * dur / len(r)) - (t0 + first * dur / len(r)))
    a = laser.angle_min + i * laser.angle_increment
    x, y = float(r[i]) * math.cos(a), float(r[i]) * math.sin(a)
    assert abs(got2[i, 0] - (math.cos(yaw) * x - math.sin(yaw) * y)) < 1e-4
    assert abs(got2[i, 1] - (math.sin(yaw) * x + math.cos(yaw) * y)) < 1e-4
    assert got2[i, 2] == 1.0  # the reference transforms (x, y, 1.0)
