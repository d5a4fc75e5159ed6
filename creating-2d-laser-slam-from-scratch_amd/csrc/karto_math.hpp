// Scalar fp64 building blocks of the correlative scan matcher, usable from host and device.
// Everything here is bit-relevant: lattice cells and lookup-table indices are *rounded* values,
// so the operation order follows the reference expression by expression and the translation
// unit is built with -ffp-contract=off (no FMA contraction, like the reference's x86-64 build).
// Citations: Math.h / Karto.h = lesson6/lib/open_karto/include/open_karto/...
#pragma once

#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <cstdint>

namespace lslam {

#define LSLAM_HD __host__ __device__ __forceinline__

constexpr double kPi = 3.14159265358979323846;    // KT_PI    (Math.h:30)
constexpr double k2Pi = 6.28318530717958647692;   // KT_2PI   (Math.h:31)
constexpr double kPi180 = 0.01745329251994329577; // KT_PI_180 (Math.h:35)
constexpr double kTol = 1e-06;                    // KT_TOLERANCE (Math.h:41)
constexpr int32_t kInvalidScan = INT_MAX;         // INVALID_SCAN (Math.h:47)
constexpr double kMaxVariance = 500.0;            // MAX_VARIANCE (Mapper.cpp:36)
constexpr double kDistPenaltyGain = 0.2;          // Mapper.cpp:37
constexpr double kAnglePenaltyGain = 0.2;         // Mapper.cpp:38
constexpr int kOccupied = 100;                    // GridStates_Occupied (Karto.h:4196)

// math::Round -- half away from zero (Math.h:87-90)
LSLAM_HD double kround(double v) { return v >= 0.0 ? floor(v + 0.5) : ceil(v - 0.5); }
LSLAM_HD double ksq(double v) { return v * v; }
// math::DoubleEqual (Math.h:135-139)
LSLAM_HD bool double_equal(double a, double b) {
  double d = a - b;
  return d < 0.0 ? d >= -kTol : d <= kTol;
}
// math::NormalizeAngle (Math.h:182-211)
LSLAM_HD double normalize_angle(double angle) {
  while (angle < -kPi) {
    if (angle < -k2Pi)
      angle += (uint32_t)(angle / -k2Pi) * k2Pi;
    else
      angle += k2Pi;
  }
  while (angle > kPi) {
    if (angle > k2Pi)
      angle -= (uint32_t)(angle / k2Pi) * k2Pi;
    else
      angle -= k2Pi;
  }
  return angle;
}
// math::NormalizeAngleDifference (Math.h:221-234)
LSLAM_HD double normalize_angle_difference(double minuend, double subtrahend) {
  while (minuend - subtrahend < -kPi) minuend += k2Pi;
  while (minuend - subtrahend > kPi) minuend -= k2Pi;
  return minuend;
}
// CoordinateConverter::WorldToGrid, one axis (Karto.h:4237-4252)
LSLAM_HD int world_to_grid(double w, double offset, double scale) {
  return (int)kround((w - offset) * scale);
}
// number of angles / lattice points: (kt_int32u)(Round(off*2/res)+1) (Mapper.cpp:339-361, Karto.h:6417)
LSLAM_HD int lattice_count(double offset, double resolution) {
  return (int)(uint32_t)(kround(offset * 2.0 / resolution) + 1);
}

// Rows 0,1 of Matrix3::FromAxisAngle(0,0,1,radians) (Karto.h:2392-2417), spelled out so the
// zero terms are rounded exactly as the reference rounds them.
struct Rot2 {
  double m00, m01, m02, m10, m11, m12;
};
LSLAM_HD Rot2 rot_z(double radians) {
  const double x = 0.0, y = 0.0, z = 1.0;
  double c = cos(radians), s = sin(radians), omc = 1.0 - c;
  double xyM = x * y * omc, xzM = x * z * omc, yzM = y * z * omc;
  double xS = x * s, yS = y * s, zS = z * s;
  Rot2 r;
  r.m00 = x * x * omc + c;
  r.m01 = xyM - zS;
  r.m02 = xzM + yS;
  r.m10 = xyM + zS;
  r.m11 = y * y * omc + c;
  r.m12 = yzM - xS;
  return r;
}
LSLAM_HD Rot2 rot_identity() { return Rot2{1.0, 0.0, 0.0, 0.0, 1.0, 0.0}; }
// Matrix3 * Pose2, rows x and y (Karto.h:2574-2583)
LSLAM_HD void rot_apply(const Rot2& r, double x, double y, double h, double& ox, double& oy) {
  ox = r.m00 * x + r.m01 * y + r.m02 * h;
  oy = r.m10 * x + r.m11 * y + r.m12 * h;
}

// karto::Transform(Pose2) = SetTransform(origin, pose) (Karto.h:2860-2863,2909-2935)
struct SensorXform {
  Rot2 rot, inv;
  double tx, ty, th;
};
LSLAM_HD SensorXform sensor_xform(double px, double py, double ph) {
  SensorXform t;
  if (px == 0.0 && py == 0.0 && ph == 0.0) {  // rPose1 == rPose2 (Karto.h:2911-2917)
    t.rot = rot_identity();
    t.inv = rot_identity();
    t.tx = t.ty = t.th = 0.0;
    return t;
  }
  t.rot = rot_z(ph - 0.0);
  t.inv = rot_z(0.0 - ph);
  t.tx = px;  // rPose1 is the origin -> newPosition = rPose2 (Karto.h:2929-2932)
  t.ty = py;
  t.th = ph - 0.0;
  return t;
}

// One beam of LocalizedRangeScan::Update (Karto.h:5384-5388 == :5394-5398)
LSLAM_HD void beam_world_point(double sx, double sy, double sh, double min_angle, double ang_res,
                               uint32_t beam, double r, double& px, double& py) {
  double angle = sh + min_angle + beam * ang_res;
#if defined(__HIP_DEVICE_COMPILE__)
  double sn, cs;  // one shared argument reduction; ocml's sincos returns the same values as sin and cos
  sincos(angle, &sn, &cs);
  px = sx + (r * cs);
  py = sy + (r * sn);
#else
  px = sx + (r * cos(angle));
  py = sy + (r * sin(angle));
#endif
}

// One lookup-table entry of GridIndexLookup::ComputeOffsets (Karto.h:6486-6496): local point
// rotated by the candidate angle, WorldToGrid(offset + rGridOffset), base-class GridIndex
// (no ROI, no bounds check).
LSLAM_HD void lookup_cell(double lx, double ly, double cosine, double sine, double off_x, double off_y,
                          double scale, int& gx, int& gy) {
  double ox = cosine * lx - sine * ly;
  double oy = sine * lx + cosine * ly;
  gx = world_to_grid(ox + off_x, off_x, scale);
  gy = world_to_grid(oy + off_y, off_y, scale);
}
// Same cell through a cheaper rounding: (int)math::Round(v) == sign(v) * (int)(|v| + 0.5) -- the
// conversion truncates, |v| + 0.5 >= 0, and ceil(v - 0.5) == -floor(-v + 0.5) (round-to-nearest is
// symmetric) -- identical to lookup_cell for |v| < 2^31.
LSLAM_HD int kround_i32(double v) {
  const int i = (int)(fabs(v) + 0.5);
  return v < 0.0 ? -i : i;
}
LSLAM_HD void lookup_cell_i32(double lx, double ly, double cosine, double sine, double off_x, double off_y,
                              double scale, int& gx, int& gy) {
  double ox = cosine * lx - sine * ly;
  double oy = sine * lx + cosine * ly;
  gx = kround_i32(((ox + off_x) - off_x) * scale);
  gy = kround_i32(((oy + off_y) - off_y) * scale);
}
LSLAM_HD int32_t lookup_offset(double lx, double ly, double cosine, double sine, double off_x,
                               double off_y, double scale, int stride) {
  int gx, gy;
  lookup_cell(lx, ly, cosine, sine, off_x, off_y, scale, gx, gy);
  return gx + gy * stride;
}

// karto::Transform(rPose1, rPose2) (Karto.h:2870-2873, 2909-2935) and TransformPose (:2881-2887)
struct PoseXform {
  Rot2 rot;
  double tx, ty, th;
};
LSLAM_HD PoseXform pose_xform(const double p1[3], const double p2[3]) {
  PoseXform t;
  if (p1[0] == p2[0] && p1[1] == p2[1] && p1[2] == p2[2]) {  // :2911-2917
    t.rot = rot_identity();
    t.tx = t.ty = t.th = 0.0;
    return t;
  }
  t.rot = rot_z(p2[2] - p1[2]);  // :2920
  if (p1[0] != 0.0 || p1[1] != 0.0) {  // :2925-2928: rPose2 - m_Rotation * rPose1
    double rx, ry;
    rot_apply(t.rot, p1[0], p1[1], p1[2], rx, ry);
    t.tx = p2[0] - rx;
    t.ty = p2[1] - ry;
  } else {
    t.tx = p2[0];
    t.ty = p2[1];
  }
  t.th = p2[2] - p1[2];  // :2934
  return t;
}
LSLAM_HD void pose_xform_apply(const PoseXform& t, const double src[3], double out[3]) {
  double rx, ry;
  rot_apply(t.rot, src[0], src[1], src[2], rx, ry);
  out[0] = t.tx + rx;
  out[1] = t.ty + ry;
  out[2] = normalize_angle(src[2] + t.th);
}

// Matrix3::InverseFast by cofactors with Inverse()'s 1e-14 tolerance (Karto.h:2445-2493)
LSLAM_HD bool mat3_inverse(const double m[9], double inv[9]) {
  inv[0] = m[4] * m[8] - m[5] * m[7];
  inv[1] = m[2] * m[7] - m[1] * m[8];
  inv[2] = m[1] * m[5] - m[2] * m[4];
  inv[3] = m[5] * m[6] - m[3] * m[8];
  inv[4] = m[0] * m[8] - m[2] * m[6];
  inv[5] = m[2] * m[3] - m[0] * m[5];
  inv[6] = m[3] * m[7] - m[4] * m[6];
  inv[7] = m[1] * m[6] - m[0] * m[7];
  inv[8] = m[0] * m[4] - m[1] * m[3];
  double det = m[0] * inv[0] + m[1] * inv[3] + m[2] * inv[6];
  if (fabs(det) <= 1e-14) return false;
  double id = 1.0 / det;
  for (int i = 0; i < 9; i++) inv[i] *= id;
  return true;
}
// Matrix3 operator* (Karto.h:2552-2568)
LSLAM_HD void mat3_mul(const double a[9], const double b[9], double out[9]) {
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++)
      out[3 * r + c] = a[3 * r] * b[c] + a[3 * r + 1] * b[3 + c] + a[3 * r + 2] * b[6 + c];
}
// MapperGraph::ComputeWeightedMean for ONE (mean, covariance) pair (Mapper.cpp:1288-1330): what
// AddEdges' closing SetSensorPose(ComputeWeightedMean(...)) (Mapper.cpp:969-972) applies when no
// near chain contributes.
LSLAM_HD void weighted_mean_single(const double mean[3], const double cov[9], double out[3]) {
  double inv[9], ios[9], w[9];
  mat3_inverse(cov, inv);
  double sum[9];
  for (int i = 0; i < 9; i++) sum[i] = 0.0 + inv[i];  // zero matrix += inverse (Mapper.cpp:1296-1303)
  mat3_inverse(sum, ios);
  double tx = cos(mean[2]), ty = sin(mean[2]);
  mat3_mul(ios, inv, w);
  double wx = w[0] * mean[0] + w[1] * mean[1] + w[2] * mean[2];  // Matrix3 * Pose2 (Karto.h:2574-2583)
  double wy = w[3] * mean[0] + w[4] * mean[1] + w[5] * mean[2];
  out[0] = 0.0 + wx;  // Pose2 operator+= on a default Pose2 (Karto.h:2117-2121)
  out[1] = 0.0 + wy;
  tx /= 1;
  ty /= 1;
  out[2] = atan2(ty, tx);
}

// MapperGraph::ComputeWeightedMean for n (mean, covariance) pairs, same accumulation order (Mapper.cpp:1288-1330):
// the running-scan match plus whatever LinkNearChains contributed.  means: n*3, covs: n*9.
inline void weighted_mean(int n, const double* means, const double* covs, double out[3]) {
  double sum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  double* inv = new double[(size_t)9 * n];
  for (int i = 0; i < n; i++) {
    mat3_inverse(covs + 9 * i, inv + 9 * i);  // Matrix3::Inverse, 1e-14 tolerance (release build: no assert)
    for (int k = 0; k < 9; k++) sum[k] += inv[9 * i + k];
  }
  double ios[9];
  mat3_inverse(sum, ios);
  double ax = 0.0, ay = 0.0, tx = 0.0, ty = 0.0;
  for (int i = 0; i < n; i++) {
    const double* m = means + 3 * i;
    tx += cos(m[2]);
    ty += sin(m[2]);
    double w[9];
    mat3_mul(ios, inv + 9 * i, w);
    ax += w[0] * m[0] + w[1] * m[1] + w[2] * m[2];  // Matrix3 * Pose2, Pose2 += (Karto.h:2574-2583, 2117-2121)
    ay += w[3] * m[0] + w[4] * m[1] + w[5] * m[2];
  }
  delete[] inv;
  tx /= n;
  ty /= n;
  out[0] = ax;
  out[1] = ay;
  out[2] = atan2(ty, tx);
}

}  // namespace lslam
