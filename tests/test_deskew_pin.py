"""Pins the lesson5 de-skew: the reference's OWN LidarUndistortion (lesson5/src/lidar_undistortion.cc compiled unmodified into
oracle/_ref/liblesson5_ref.so behind oracle/lesson5_ref_driver.cpp; ROS / tf / PCL / Eigen replaced by the stand-ins under
oracle/shim, which define PCL's getTransformation formula and Eigen's evaluation orders and nothing of the reference's own
statements) against the numpy restatement (tests/deskew_restatement.py) -- CPU -- and, under the gpu marker, against
lslam_deskew_scan.  The messages go through the reference's own callbacks: ImuCallback / OdomCallback fill its queues,
PruneImuDeque integrates the gyro samples, PruneOdomDeque derives the odometry increment (tf getRPY, Affine3f inverse and
product, getTranslationAndEulerAngles), CorrectLaserScan de-skews; the state CorrectLaserScan read is what the C ABI's
lslam_deskew_scan takes as inputs (lidar_undistortion.cc:339-447 is the part on the device)."""
import math

import numpy as np
import pytest

from lslam_amd import api, synth

from deskew_restatement import restated_deskew

f32 = np.float32


@pytest.fixture(scope="module")
def po5(oracle_lib):
    if not oracle_lib.have_ref_lesson5():
        pytest.skip("oracle/_ref/liblesson5_ref.so not built (needs /root/reference)")
    return oracle_lib


def _quat_yaw(yaw):
    return (0.0, 0.0, math.sin(0.5 * yaw), math.cos(0.5 * yaw))


def _run_reference(po5, use_imu, use_odom, seed):
    """Two scans through the reference (it corrects a scan when the NEXT one arrives), 100 Hz IMU + 50 Hz odometry around them."""
    laser = synth.Laser()
    world = synth.arena(size=40.0, n_axis=10, n_rot=4, seed=5)
    rng = np.random.default_rng(seed)
    r = synth.cast_scan(world, (0.3, -0.2, 0.1), laser, 0.01, 0.02, rng)
    r[7] = np.float32("nan")
    r[0] = np.float32(0.01)       # below range_min: the reference anchors its start transform on the first VALID beam
    r[40] = np.float32(45.0)      # above range_max
    t0, dur = 1000.0, 0.1
    n = len(r)
    node = po5.RefLesson5(use_imu, use_odom)
    for k in range(-3, 40):       # IMU 100 Hz from 30 ms before the scan to well after the second one
        t = t0 - 0.003 + 0.01 * k
        node.add_imu(t, (0.02 * math.sin(k), -0.03 + 0.001 * k, 0.6 + 0.05 * math.cos(0.3 * k)))
    for k in range(-3, 20):       # odometry 50 Hz: a gentle arc
        t = t0 - 0.004 + 0.02 * k
        yaw = 0.3 + 0.06 * k
        node.add_odom(t, (1.0 + 0.01 * k, 2.0 + 0.004 * k * k * 0.1, 0.0), _quat_yaw(yaw))
    args = (laser.angle_min, laser.angle_increment, dur / n, laser.range_min, 30.0)
    assert node.scan(t0, *args, r) is None                  # queued
    out = node.scan(t0 + 0.12, *args, r[::-1].copy())       # the first scan is corrected now
    assert out is not None
    return laser, r, out


@pytest.mark.parametrize("use_imu,use_odom", [(True, True), (True, False), (False, True)])
def test_restatement_equals_the_reference(po5, use_imu, use_odom):
    laser, r, ref = _run_reference(po5, use_imu, use_odom, 3)
    assert np.array_equal(ref["ranges"], r, equal_nan=True) and ref["valid"].sum() > 500
    assert not ref["valid"][0] and not ref["valid"][7] and not ref["valid"][40]
    if use_imu:
        assert len(ref["imu_time"]) >= 9 and ref["imu_time"][0] < ref["scan_time_start"]
        assert np.abs(ref["imu_rot"][-1]).max() > 1e-3
    if use_odom:
        assert np.abs(ref["odom_incre"][:2]).max() > 1e-3 and ref["start_odom_time"] < ref["scan_time_start"] < ref["end_odom_time"]
    p = api.DeskewParams(ref["angle_min"], ref["angle_increment"], ref["range_min"], ref["range_max"], ref["scan_time_start"],
                         ref["time_increment"], int(use_imu), int(use_odom), ref["start_odom_time"], ref["end_odom_time"],
                         float(ref["odom_incre"][0]), float(ref["odom_incre"][1]), float(ref["odom_incre"][2]), 0.0)
    times = list(ref["imu_time"]) if use_imu else [0.0]
    rots = [list(v) for v in ref["imu_rot"]] if use_imu else [[0.0, 0.0, 0.0]]
    want, want_valid = restated_deskew(r, p, times, rots)
    assert np.array_equal(want_valid, ref["valid"])
    # same statements, same float32 / float64 types: equal up to glibc cosf / sinf vs the restatement's rounded double
    # cos / sin in getTransformation (a last-bit difference of an Euler angle's cosine moves a 30 m point by ~2e-6 m)
    d = np.abs(want - ref["xyz"])
    assert d.max() <= 1e-6, d.max()         # observed: 0.0 (IMU only / odometry only), 4.8e-7 (both)
    assert np.mean(d == 0) > 0.99           # and practically every coordinate is bit-equal (observed 99.97 % / 100 %)
    assert np.all(ref["xyz"][~ref["valid"]] == 0)


@pytest.mark.gpu
@pytest.mark.parametrize("use_imu,use_odom", [(True, True), (True, False), (False, True)])
def test_device_deskew_equals_the_reference(ctx, po5, use_imu, use_odom):
    """lslam_deskew_scan fed the state the reference's own Prune* steps produced == the reference's CorrectLaserScan
    (float32 cos / sin of the Euler angles come from the device libm: 4e-6 m)."""
    laser, r, ref = _run_reference(po5, use_imu, use_odom, 4)
    p = api.DeskewParams(ref["angle_min"], ref["angle_increment"], ref["range_min"], ref["range_max"], ref["scan_time_start"],
                         ref["time_increment"], int(use_imu), int(use_odom), ref["start_odom_time"], ref["end_odom_time"],
                         float(ref["odom_incre"][0]), float(ref["odom_incre"][1]), float(ref["odom_incre"][2]), 0.0)
    times = list(ref["imu_time"]) if use_imu else [0.0]
    rots = [list(v) for v in ref["imu_rot"]] if use_imu else [[0.0, 0.0, 0.0]]
    got, got_valid = api.deskew_scan(ctx, r, p, times, rots)
    assert np.array_equal(got_valid, ref["valid"])
    assert np.abs(got - ref["xyz"]).max() <= 4e-6
    assert np.all(got[~ref["valid"]] == 0)
