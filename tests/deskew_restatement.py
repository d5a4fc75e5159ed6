"""A numpy restatement of lesson5's de-skew arithmetic (LidarUndistortion::CorrectLaserScan / ComputeRotation /
ComputePosition, lesson5/src/lidar_undistortion.cc:339-447) with pcl::getTransformation's published formula and Eigen
3.3's evaluation orders -- pinned against the reference's own compiled code by tests/test_deskew_pin.py (round 4), used by
tests/test_deskew_gpu.py as a second, GPU-box-independent checker."""
import math

import numpy as np

f32 = np.float32


def get_transformation(x, y, z, roll, pitch, yaw):
    A, B, Cc, D = f32(math.cos(yaw)), f32(math.sin(yaw)), f32(math.cos(pitch)), f32(math.sin(pitch))
    E, F = f32(math.cos(roll)), f32(math.sin(roll))
    DE, DF = D * E, D * F
    L = np.array([[A * Cc, A * DF - B * E, B * F + A * DE], [B * Cc, A * E + B * DF, B * DE - A * F], [-D, Cc * F, Cc * E]],
                 dtype=np.float32)
    return L, np.array([x, y, z], dtype=np.float32)


def cof(m, i, j):
    i1, i2, j1, j2 = (i + 1) % 3, (i + 2) % 3, (j + 1) % 3, (j + 2) % 3
    return m[i1, j1] * m[i2, j2] - m[i1, j2] * m[i2, j1]


def inverse(L, t):
    c0, c1, c2 = cof(L, 0, 0), cof(L, 1, 0), cof(L, 2, 0)
    det = c0 * L[0, 0] + (c1 * L[1, 0] + c2 * L[2, 0])
    inv = f32(1.0) / det
    R = np.array([[c0 * inv, c1 * inv, c2 * inv], [cof(L, 0, 1) * inv, cof(L, 1, 1) * inv, cof(L, 2, 1) * inv],
                  [cof(L, 0, 2) * inv, cof(L, 1, 2) * inv, cof(L, 2, 2) * inv]], dtype=np.float32)
    tt = np.array([(-R[i, 0]) * t[0] + ((-R[i, 1]) * t[1] + (-R[i, 2]) * t[2]) for i in range(3)], dtype=np.float32)
    return R, tt


def mul(a, b):
    (La, ta), (Lb, tb) = a, b
    L = np.array([[La[i, 0] * Lb[0, j] + (La[i, 1] * Lb[1, j] + La[i, 2] * Lb[2, j]) for j in range(3)] for i in range(3)],
                 dtype=np.float32)
    t = np.array([(La[i, 0] * tb[0] + (La[i, 1] * tb[1] + La[i, 2] * tb[2])) + ta[i] for i in range(3)], dtype=np.float32)
    return L, t


def restated_deskew(r, p, imu_time, imu_rot):
    n = len(r)
    out = np.zeros((n, 3), np.float32)
    valid = np.zeros(n, bool)
    last = len(imu_time) - 1

    def transform_at(i):
        t = p.scan_time_start + i * p.time_increment
        rot = [f32(0)] * 3
        pos = [f32(0)] * 3
        if p.use_imu:
            f = 0
            while f < last:
                if t < imu_time[f]:
                    break
                f += 1
            if t > imu_time[f] or f == 0:
                rot = [f32(imu_rot[f][k]) for k in range(3)]
            else:
                b = f - 1
                rf = (t - imu_time[b]) / (imu_time[f] - imu_time[b])
                rb = (imu_time[f] - t) / (imu_time[f] - imu_time[b])
                rot = [f32(imu_rot[f][k] * rf + imu_rot[b][k] * rb) for k in range(3)]
        if p.use_odom:
            rf = (t - p.start_odom_time) / (p.end_odom_time - p.start_odom_time)
            pos = [f32(float(v) * rf) for v in (p.odom_incre_x, p.odom_incre_y, p.odom_incre_z)]
        return get_transformation(pos[0], pos[1], pos[2], float(rot[0]), float(rot[1]), float(rot[2]))

    start_inv = None
    for i in range(n):
        ri = r[i]
        if not np.isfinite(ri) or ri < f32(p.range_min) or ri > f32(p.range_max):
            continue
        valid[i] = True
        a = float(f32(f32(p.angle_min) + f32(i) * f32(p.angle_increment)))  # float32 like CreateAngleCache (:169)
        px, py, pz = float(ri) * math.cos(a), float(ri) * math.sin(a), 1.0
        cur = transform_at(i)
        if start_inv is None:
            start_inv = inverse(*cur)
        L, t = mul(start_inv, cur)
        L, t = L.astype(np.float64), t.astype(np.float64)
        out[i] = [f32(((L[k, 0] * px + L[k, 1] * py) + L[k, 2] * pz) + t[k]) for k in range(3)]
    return out, valid
