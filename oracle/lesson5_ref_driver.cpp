// TEST INFRASTRUCTURE ONLY -- C entry points around the reference's UNMODIFIED lesson5 de-skew node class
// (lesson5/src/lidar_undistortion.cc + lesson5/include/lesson5/lidar_undistortion.h, compiled in place from
// /root/reference, never copied), so that tests can run LidarUndistortion::CacheLaserScan / PruneImuDeque /
// PruneOdomDeque / CorrectLaserScan themselves.  ROS, tf, PCL and Eigen are not in this image: the source compiles against
// the stand-ins in oracle/shim/ros_pcl (what they define is stated in lslam_ros_pcl_shim.hpp) and oracle/shim/Eigen.
// Built by `make -C oracle ref_lesson5` into oracle/_ref/liblesson5_ref.so.
//
// Only tests/ may load it.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>

// the node class keeps its state and its steps private; the driver needs both (same device as the other ref drivers)
#define private public
#define main lesson5_node_main_unused
#include "lidar_undistortion.cc"  // -I$(REF)/lesson5/src, -I$(REF)/lesson5/include
#undef main
#undef private

extern "C" {

int l5_abi(void) { return 1; }

void* l5_create(int use_imu, int use_odom) {
  ros::NodeHandle::lslam_use_imu() = use_imu;
  ros::NodeHandle::lslam_use_odom() = use_odom;
  return new LidarUndistortion();
}
void l5_destroy(void* h) { delete (LidarUndistortion*)h; }

// sensor_msgs/Imu: stamp + angular velocity (ImuCallback, :82-87)
void l5_add_imu(void* h, double stamp, double wx, double wy, double wz) {
  auto m = std::make_shared<sensor_msgs::Imu>();
  m->header.stamp = ros::Time(stamp);
  m->angular_velocity.x = wx;
  m->angular_velocity.y = wy;
  m->angular_velocity.z = wz;
  ((LidarUndistortion*)h)->ImuCallback(m);
}
// nav_msgs/Odometry: stamp + pose (OdomCallback, :89-94)
void l5_add_odom(void* h, double stamp, double x, double y, double z, double qx, double qy, double qz, double qw) {
  auto m = std::make_shared<nav_msgs::Odometry>();
  m->header.stamp = ros::Time(stamp);
  m->pose.pose.position.x = x;
  m->pose.pose.position.y = y;
  m->pose.pose.position.z = z;
  m->pose.pose.orientation.x = qx;
  m->pose.pose.orientation.y = qy;
  m->pose.pose.orientation.z = qz;
  m->pose.pose.orientation.w = qw;
  ((LidarUndistortion*)h)->OdomCallback(m);
}

// One LaserScan through ScanCallback's own sequence (:96-125), step by step so that the state the de-skew reads can be
// captured between PruneOdomDeque and CorrectLaserScan.  The reference corrects the scan that arrived BEFORE this one (it
// keeps one scan queued, :146-152).  Returns 1 when a scan was corrected, 0 while the first scan is queued, < 0 when a
// Prune* step refused (waiting for IMU / odometry data).
//   out_xyz[3 n], out_valid[n]: corrected_pointcloud_ and the validity rule of :351-354 (points the reference skips stay 0)
//   state[8]: current_scan_time_start_, current_scan_time_increment_, start_odom_time_, end_odom_time_, odom_incre_x/y/z_,
//             current_imu_index_;  imu[4 * cap]: imu_time_, imu_rot_x_, imu_rot_y_, imu_rot_z_ (first current_imu_index_+1 each)
//   scan_hdr[4]: angle_min, angle_increment, range_min, range_max of the scan that WAS corrected
int l5_scan(void* h, double stamp, float angle_min, float angle_increment, float time_increment, float range_min, float range_max,
            const float* ranges, int n, float* out_xyz, uint8_t* out_valid, double* state, double* imu, int cap, float* scan_hdr,
            float* corrected_ranges) {
  LidarUndistortion* u = (LidarUndistortion*)h;
  auto m = std::make_shared<sensor_msgs::LaserScan>();
  m->header.stamp = ros::Time(stamp);
  m->angle_min = angle_min;
  m->angle_increment = angle_increment;
  m->angle_max = angle_min + angle_increment * (float)(n - 1);
  m->time_increment = time_increment;
  m->range_min = range_min;
  m->range_max = range_max;
  m->ranges.assign(ranges, ranges + n);
  if (!u->CacheLaserScan(m)) return 0;
  if (u->use_imu_ && !u->PruneImuDeque()) return -1;
  if (u->use_odom_ && !u->PruneOdomDeque()) return -2;
  state[0] = u->current_scan_time_start_;
  state[1] = u->current_scan_time_increment_;
  state[2] = u->start_odom_time_;
  state[3] = u->end_odom_time_;
  state[4] = u->odom_incre_x_;
  state[5] = u->odom_incre_y_;
  state[6] = u->odom_incre_z_;
  state[7] = u->current_imu_index_;
  for (int i = 0; i <= u->current_imu_index_ && i < cap; i++) {
    imu[i] = u->imu_time_[i];
    imu[cap + i] = u->imu_rot_x_[i];
    imu[2 * cap + i] = u->imu_rot_y_[i];
    imu[3 * cap + i] = u->imu_rot_z_[i];
  }
  const sensor_msgs::LaserScan& cur = u->current_laserscan_;
  scan_hdr[0] = cur.angle_min;
  scan_hdr[1] = cur.angle_increment;
  scan_hdr[2] = cur.range_min;
  scan_hdr[3] = cur.range_max;
  u->CorrectLaserScan();
  const int cnt = (int)u->scan_count_;
  for (int i = 0; i < n && i < cnt; i++) {
    const pcl::PointXYZ& p = u->corrected_pointcloud_->points[i];
    out_xyz[3 * i] = p.x;
    out_xyz[3 * i + 1] = p.y;
    out_xyz[3 * i + 2] = p.z;
    const float r = cur.ranges[i];
    corrected_ranges[i] = r;
    out_valid[i] = (std::isfinite(r) && !(r < cur.range_min) && !(r > cur.range_max)) ? 1 : 0;  // :351-354
  }
  u->PublishCorrectedPointCloud();
  u->ResetParameters();
  return 1;
}

}  // extern "C"
