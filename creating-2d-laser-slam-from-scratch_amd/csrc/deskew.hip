// lesson5 lidar motion de-skew on the device: LidarUndistortion::CorrectLaserScan with its
// ComputeRotation / ComputePosition helpers (lesson5/src/lidar_undistortion.cc:339-396, 397-447).  Thread per beam:
// time of the beam, IMU rotation interpolated from the integrated gyro samples, odometry translation interpolated over
// the scan, pcl::getTransformation of both (float32 Euler -> affine), the transform of the FIRST valid beam inverted once
// per scan, transBt = transStartInverse * transFinal, point = transBt * (r cos a, r sin a, 1.0) in the reference's mixed
// float/double arithmetic (the z = 1.0 is the reference's).
// The reference's arithmetic goes through PCL (getTransformation) and Eigen (Affine3f inverse and product) inside a ROS node
// class.  This kernel follows the published PCL formula (pcl/common/impl/eigen.hpp) and Eigen 3.3's evaluation orders;
// round 4 pins it against the reference's own source compiled in place behind stand-ins for those libraries
// (tests/test_deskew_pin.py: <= 4e-6 m, the device's float32 cos / sin being the difference).
#include <cmath>

#include "common.hpp"

using namespace lslam;

namespace {

struct Aff3 {  // Eigen::Affine3f: linear (row-major here) + translation
  float l[9], t[3];
};

// pcl::getTransformation(x, y, z, roll, pitch, yaw) for float
__device__ __forceinline__ Aff3 get_transformation(float x, float y, float z, float roll, float pitch, float yaw) {
  const float A = cosf(yaw), B = sinf(yaw), C = cosf(pitch), D = sinf(pitch), E = cosf(roll), F = sinf(roll);
  const float DE = D * E, DF = D * F;
  Aff3 t;
  t.l[0] = A * C;  t.l[1] = A * DF - B * E;  t.l[2] = B * F + A * DE;  t.t[0] = x;
  t.l[3] = B * C;  t.l[4] = A * E + B * DF;  t.l[5] = B * DE - A * F;  t.t[1] = y;
  t.l[6] = -D;     t.l[7] = C * F;           t.l[8] = C * E;           t.t[2] = z;
  return t;
}
__device__ __forceinline__ float cof3(const float* m, int i, int j) {  // Eigen cofactor_3x3<i,j>
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m[3 * i1 + j1] * m[3 * i2 + j2] - m[3 * i1 + j2] * m[3 * i2 + j1];
}
// Transform::inverse(Affine): linear by the cofactor inverse (LU/InverseImpl.h), translation = (-Linv) * t
__device__ __forceinline__ Aff3 inverse(const Aff3& a) {
  const float* m = a.l;
  const float c0 = cof3(m, 0, 0), c1 = cof3(m, 1, 0), c2 = cof3(m, 2, 0);
  const float det = c0 * m[0] + (c1 * m[3] + c2 * m[6]);
  const float invdet = 1.0f / det;
  Aff3 r;
  r.l[0] = c0 * invdet; r.l[1] = c1 * invdet; r.l[2] = c2 * invdet;
  r.l[3] = cof3(m, 0, 1) * invdet; r.l[4] = cof3(m, 1, 1) * invdet; r.l[5] = cof3(m, 2, 1) * invdet;
  r.l[6] = cof3(m, 0, 2) * invdet; r.l[7] = cof3(m, 1, 2) * invdet; r.l[8] = cof3(m, 2, 2) * invdet;
  for (int i = 0; i < 3; i++)
    r.t[i] = (-r.l[3 * i]) * a.t[0] + ((-r.l[3 * i + 1]) * a.t[1] + (-r.l[3 * i + 2]) * a.t[2]);
  return r;
}
// Affine * Affine (Geometry/Transform.h): linear = L1 L2, translation = L1 t2 + t1; 3-term sums a0 + (a1 + a2)
__device__ __forceinline__ Aff3 mul(const Aff3& a, const Aff3& b) {
  Aff3 r;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++)
      r.l[3 * i + j] = a.l[3 * i] * b.l[j] + (a.l[3 * i + 1] * b.l[3 + j] + a.l[3 * i + 2] * b.l[6 + j]);
    r.t[i] = (a.l[3 * i] * b.t[0] + (a.l[3 * i + 1] * b.t[1] + a.l[3 * i + 2] * b.t[2])) + a.t[i];
  }
  return r;
}

struct DeskewCfg {
  int n, n_imu_last;  // n_imu_last = current_imu_index_ (index of the last integrated IMU sample)
  float range_min, range_max;
  float angle_min, angle_inc;  // the LaserScan message's float32 fields
  double t0, dt;
  int use_imu, use_odom;
  double odom_t0, odom_t1;
  float odom_dx, odom_dy, odom_dz;
};

__device__ __forceinline__ Aff3 transform_at(const DeskewCfg& c, int i, const double* imu_time, const double* rx,
                                             const double* ry, const double* rz) {
  const double t = c.t0 + i * c.dt;  // :358
  float rotX = 0, rotY = 0, rotZ = 0, posX = 0, posY = 0, posZ = 0;
  if (c.use_imu) {  // ComputeRotation (:397-434)
    int f = 0;
    while (f < c.n_imu_last) {
      if (t < imu_time[f]) break;
      ++f;
    }
    if (t > imu_time[f] || f == 0) {
      rotX = (float)rx[f]; rotY = (float)ry[f]; rotZ = (float)rz[f];
    } else {
      const int b = f - 1;
      const double rf = (t - imu_time[b]) / (imu_time[f] - imu_time[b]);
      const double rb = (imu_time[f] - t) / (imu_time[f] - imu_time[b]);
      rotX = (float)(rx[f] * rf + rx[b] * rb);
      rotY = (float)(ry[f] * rf + ry[b] * rb);
      rotZ = (float)(rz[f] * rf + rz[b] * rb);
    }
  }
  if (c.use_odom) {  // ComputePosition (:437-447)
    const double rf = (t - c.odom_t0) / (c.odom_t1 - c.odom_t0);
    posX = (float)((double)c.odom_dx * rf);
    posY = (float)((double)c.odom_dy * rf);
    posZ = (float)((double)c.odom_dz * rf);
  }
  return get_transformation(posX, posY, posZ, rotX, rotY, rotZ);
}

__global__ void __launch_bounds__(256)
k_deskew(DeskewCfg c, const float* __restrict__ ranges, const double* __restrict__ imu_time, const double* __restrict__ rx,
         const double* __restrict__ ry, const double* __restrict__ rz, float* __restrict__ out_xyz, uint8_t* __restrict__ valid) {
  __shared__ int s_first;
  __shared__ Aff3 s_start_inv;
  if (threadIdx.x == 0) s_first = c.n;
  __syncthreads();
  auto is_valid = [&](int i) {
    const float r = ranges[i];
    return !(!isfinite(r) || r < c.range_min || r > c.range_max);  // :350-353
  };
  for (int i = threadIdx.x; i < c.n; i += blockDim.x)
    if (is_valid(i)) atomicMin(&s_first, i);
  __syncthreads();
  if (threadIdx.x == 0 && s_first < c.n) s_start_inv = inverse(transform_at(c, s_first, imu_time, rx, ry, rz));  // :377-383
  __syncthreads();
  for (int i = threadIdx.x; i < c.n; i += blockDim.x) {
    const bool ok = is_valid(i);
    valid[i] = ok ? 1 : 0;
    float x = 0.f, y = 0.f, z = 0.f;
    if (ok) {
      // CreateAngleCache (:164-173): the angle is evaluated in FLOAT32 (angle_min + i * angle_increment on the message's
      // float fields), then widened for the double cos / sin
      const float af = c.angle_min + (float)i * c.angle_inc;
      const double a = (double)af;
      const double px = (double)ranges[i] * cos(a), py = (double)ranges[i] * sin(a), pz = 1.0;  // :361-362, :343
      const Aff3 bt = mul(s_start_inv, transform_at(c, i, imu_time, rx, ry, rz));             // :386-390
      x = (float)((((double)bt.l[0] * px + (double)bt.l[1] * py) + (double)bt.l[2] * pz) + (double)bt.t[0]);  // :394-396
      y = (float)((((double)bt.l[3] * px + (double)bt.l[4] * py) + (double)bt.l[5] * pz) + (double)bt.t[1]);
      z = (float)((((double)bt.l[6] * px + (double)bt.l[7] * py) + (double)bt.l[8] * pz) + (double)bt.t[2]);
    }
    out_xyz[3 * i] = x; out_xyz[3 * i + 1] = y; out_xyz[3 * i + 2] = z;
  }
}

}  // namespace

extern "C" {

int lslam_deskew_scan(lslam_context* ctx, const float* ranges, int n, const lslam_deskew_params* p, const double* imu_time,
                      const double* imu_rot_x, const double* imu_rot_y, const double* imu_rot_z, int n_imu,
                      float* out_xyz, uint8_t* out_valid) {
  if (!ctx || n < 0 || (n > 0 && (!ranges || !out_xyz || !out_valid)) || !p) return LSLAM_ERR_INVALID_ARGUMENT;
  if (p->use_imu && (n_imu < 1 || !imu_time || !imu_rot_x || !imu_rot_y || !imu_rot_z)) return LSLAM_ERR_INVALID_ARGUMENT;
  if (n == 0) return LSLAM_OK;
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  const int ni = p->use_imu ? n_imu : 1;
  float* d_r = nullptr;
  double* d_imu = nullptr;
  float* d_out = nullptr;
  uint8_t* d_v = nullptr;
  auto cleanup = [&]() {
    if (d_r) (void)hipFree(d_r);
    if (d_imu) (void)hipFree(d_imu);
    if (d_out) (void)hipFree(d_out);
    if (d_v) (void)hipFree(d_v);
  };
  if (hipMalloc((void**)&d_r, (size_t)n * sizeof(float)) != hipSuccess ||
      hipMalloc((void**)&d_imu, (size_t)4 * ni * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&d_out, (size_t)3 * n * sizeof(float)) != hipSuccess || hipMalloc((void**)&d_v, (size_t)n) != hipSuccess) {
    cleanup();
    return ctx->fail(LSLAM_ERR_HIP, "lslam_deskew_scan: out of device memory");
  }
  hipError_t e = hipMemcpyAsync(d_r, ranges, (size_t)n * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
  if (p->use_imu) {
    const double* src[4] = {imu_time, imu_rot_x, imu_rot_y, imu_rot_z};
    for (int k = 0; k < 4 && e == hipSuccess; k++)
      e = hipMemcpyAsync(d_imu + (size_t)k * ni, src[k], (size_t)ni * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  }
  DeskewCfg c;
  c.n = n; c.n_imu_last = ni - 1;
  c.range_min = p->range_min; c.range_max = p->range_max;
  c.angle_min = p->angle_min; c.angle_inc = p->angle_increment;
  c.t0 = p->scan_time_start; c.dt = p->time_increment;
  c.use_imu = p->use_imu; c.use_odom = p->use_odom;
  c.odom_t0 = p->start_odom_time; c.odom_t1 = p->end_odom_time;
  c.odom_dx = p->odom_incre_x; c.odom_dy = p->odom_incre_y; c.odom_dz = p->odom_incre_z;
  if (e == hipSuccess) {
    launch(ctx, "deskew", k_deskew, dim3(1), dim3(256), 0, c, (const float*)d_r, (const double*)d_imu, (const double*)(d_imu + ni),
           (const double*)(d_imu + 2 * (size_t)ni), (const double*)(d_imu + 3 * (size_t)ni), d_out, d_v);
    e = hipMemcpyAsync(out_xyz, d_out, (size_t)3 * n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out_valid, d_v, (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  cleanup();
  if (e != hipSuccess) return ctx->fail(LSLAM_ERR_HIP, "lslam_deskew_scan: %s", hipGetErrorString(e));
  return LSLAM_OK;
}

}  // extern "C"
