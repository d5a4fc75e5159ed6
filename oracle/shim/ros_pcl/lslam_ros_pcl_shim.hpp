// TEST INFRASTRUCTURE ONLY -- stand-ins for the ROS / tf / PCL headers lesson5's lidar_undistortion.{h,cc} include, so that
// the reference's source compiles UNMODIFIED, in place, into oracle/_ref/liblesson5_ref.so (oracle/Makefile, driver
// oracle/lesson5_ref_driver.cpp) and tests can run LidarUndistortion::CorrectLaserScan itself.  None of ROS, tf or PCL is in
// this image.  What these stand-ins DEFINE (everything else -- validity rule, time stamps, the IMU / odometry
// interpolation, the output statements -- is the reference's own compiled code):
//   * message structs with the fields the reference reads (sensor_msgs/LaserScan, Imu, nav_msgs/Odometry, std_msgs/Header);
//   * ros::NodeHandle & friends as inert objects (no middleware: the driver calls the callbacks directly);
//   * pcl::getTransformation / getTranslationAndEulerAngles: PCL's published formulas (common/impl/eigen.hpp: the 3x3 block
//     A*C, A*DF - B*E, B*F + A*DE / B*C, A*E + B*DF, B*DE - A*F / -D, C*F, C*E with A = cos(yaw) ... F = sin(roll) evaluated
//     in the Scalar type, float here; roll = atan2(t(2,1), t(2,2)), pitch = asin(-t(2,0)), yaw = atan2(t(1,0), t(0,0)));
//   * tf::Matrix3x3(q).getRPY: tf's setRotation + getEulerYPR (LinearMath/Matrix3x3.h), double;
//   * Eigen::Affine3f products / inverse: oracle/shim/Eigen (Eigen 3.3's evaluation orders, stated there).
#ifndef LSLAM_ORACLE_ROS_PCL_SHIM
#define LSLAM_ORACLE_ROS_PCL_SHIM
#include <cmath>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include <Eigen/Core>
#include <Eigen/Geometry>

#define ROS_INFO_STREAM(x) do { } while (0)
#define ROS_INFO(...) do { } while (0)
#define ROS_WARN(...) do { } while (0)
#define ROS_ERROR(...) do { } while (0)

namespace ros {
struct Time {
  double sec = 0.0;
  Time() {}
  explicit Time(double s) : sec(s) {}
  double toSec() const { return sec; }
};
struct TransportHints {
  TransportHints& tcpNoDelay() { return *this; }
};
struct Subscriber {};
struct Publisher {
  template <class M>
  void publish(const M&) const {}
};
struct NodeHandle {
  NodeHandle(const std::string& = std::string()) {}
  template <class M, class T>
  Subscriber subscribe(const std::string&, uint32_t, void (T::*)(const std::shared_ptr<M const>&), T*, const TransportHints& = TransportHints()) {
    return Subscriber();
  }
  template <class M, class X>
  Publisher advertise(const std::string&, uint32_t, X) {
    return Publisher();
  }
  template <class T>
  bool param(const std::string& name, T& value, const T& dflt) const {
    value = dflt;
    if (name == "use_imu" && lslam_use_imu() >= 0) value = (T)(lslam_use_imu() != 0);
    if (name == "use_odometry" && lslam_use_odom() >= 0) value = (T)(lslam_use_odom() != 0);
    return true;
  }
  // the two node parameters the reference reads (lidar_undistortion.cc:27-28), set by the driver before construction
  static int& lslam_use_imu() { static int v = -1; return v; }
  static int& lslam_use_odom() { static int v = -1; return v; }
};
inline void init(int&, char**, const std::string&) {}
struct AsyncSpinner {
  explicit AsyncSpinner(int) {}
  void start() {}
};
inline void waitForShutdown() {}
}  // namespace ros

namespace std_msgs {
struct Header {
  uint32_t seq = 0;
  ros::Time stamp;
  std::string frame_id;
};
}  // namespace std_msgs

namespace geometry_msgs {
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct PoseWithCovariance { Pose pose; double covariance[36] = {0}; };
}  // namespace geometry_msgs

namespace sensor_msgs {
struct LaserScan {
  typedef std::shared_ptr<LaserScan const> ConstPtr;
  std_msgs::Header header;
  float angle_min = 0, angle_max = 0, angle_increment = 0, time_increment = 0, scan_time = 0, range_min = 0, range_max = 0;
  std::vector<float> ranges, intensities;
};
struct Imu {
  typedef std::shared_ptr<Imu const> ConstPtr;
  std_msgs::Header header;
  geometry_msgs::Quaternion orientation;
  geometry_msgs::Vector3 angular_velocity, linear_acceleration;
};
struct PointCloud2 {};
}  // namespace sensor_msgs

namespace nav_msgs {
struct Odometry {
  typedef std::shared_ptr<Odometry const> ConstPtr;
  std_msgs::Header header;
  std::string child_frame_id;
  geometry_msgs::PoseWithCovariance pose;
};
}  // namespace nav_msgs

namespace tf {
struct Quaternion {
  double x_ = 0, y_ = 0, z_ = 0, w_ = 1;
  Quaternion() {}
  Quaternion(double x, double y, double z, double w) : x_(x), y_(y), z_(z), w_(w) {}
};
inline void quaternionMsgToTF(const geometry_msgs::Quaternion& m, Quaternion& q) { q = Quaternion(m.x, m.y, m.z, m.w); }
struct Matrix3x3 {
  double m[3][3];
  explicit Matrix3x3(const Quaternion& q) {  // setRotation (LinearMath/Matrix3x3.h)
    const double d = q.x_ * q.x_ + q.y_ * q.y_ + q.z_ * q.z_ + q.w_ * q.w_;
    const double s = 2.0 / d;
    const double xs = q.x_ * s, ys = q.y_ * s, zs = q.z_ * s;
    const double wx = q.w_ * xs, wy = q.w_ * ys, wz = q.w_ * zs;
    const double xx = q.x_ * xs, xy = q.x_ * ys, xz = q.x_ * zs;
    const double yy = q.y_ * ys, yz = q.y_ * zs, zz = q.z_ * zs;
    m[0][0] = 1.0 - (yy + zz); m[0][1] = xy - wz; m[0][2] = xz + wy;
    m[1][0] = xy + wz; m[1][1] = 1.0 - (xx + zz); m[1][2] = yz - wx;
    m[2][0] = xz - wy; m[2][1] = yz + wx; m[2][2] = 1.0 - (xx + yy);
  }
  void getRPY(double& roll, double& pitch, double& yaw) const {  // getEulerYPR, solution 1
    if (std::fabs(m[2][0]) >= 1.0) {
      yaw = 0.0;
      const double delta = std::atan2(m[2][1], m[2][2]);
      if (m[2][0] < 0) {
        pitch = M_PI / 2.0;
        roll = delta;
      } else {
        pitch = -M_PI / 2.0;
        roll = delta;
      }
    } else {
      pitch = -std::asin(m[2][0]);
      roll = std::atan2(m[2][1] / std::cos(pitch), m[2][2] / std::cos(pitch));
      yaw = std::atan2(m[1][0] / std::cos(pitch), m[0][0] / std::cos(pitch));
    }
  }
};
}  // namespace tf

namespace pcl {
struct PCLHeader {
  uint32_t seq = 0;
  uint64_t stamp = 0;
  std::string frame_id;
};
struct PointXYZ {
  float x = 0.f, y = 0.f, z = 0.f;
};
template <class PointT>
struct PointCloud {
  typedef std::shared_ptr<PointCloud<PointT>> Ptr;
  PCLHeader header;
  std::vector<PointT> points;
  uint32_t width = 0, height = 0;
  bool is_dense = true;
};
// common/impl/eigen.hpp, Scalar = float
inline Eigen::Affine3f getTransformation(float x, float y, float z, float roll, float pitch, float yaw) {
  const float A = std::cos(yaw), B = std::sin(yaw), C = std::cos(pitch), D = std::sin(pitch), E = std::cos(roll), F = std::sin(roll),
              DE = D * E, DF = D * F;
  Eigen::Affine3f t;
  t.linear()(0, 0) = A * C; t.linear()(0, 1) = A * DF - B * E; t.linear()(0, 2) = B * F + A * DE; t.translation()[0] = x;
  t.linear()(1, 0) = B * C; t.linear()(1, 1) = A * E + B * DF; t.linear()(1, 2) = B * DE - A * F; t.translation()[1] = y;
  t.linear()(2, 0) = -D;    t.linear()(2, 1) = C * F;          t.linear()(2, 2) = C * E;          t.translation()[2] = z;
  return t;
}
inline void getTranslationAndEulerAngles(const Eigen::Affine3f& t, float& x, float& y, float& z, float& roll, float& pitch, float& yaw) {
  x = t(0, 3);
  y = t(1, 3);
  z = t(2, 3);
  roll = std::atan2(t(2, 1), t(2, 2));
  pitch = std::asin(-t(2, 0));
  yaw = std::atan2(t(1, 0), t(0, 0));
}
}  // namespace pcl

namespace pcl_conversions {
inline void toPCL(const std_msgs::Header& h, pcl::PCLHeader& o) {
  o.seq = h.seq;
  o.stamp = (uint64_t)(h.stamp.toSec() * 1e6);
  o.frame_id = h.frame_id;
}
}  // namespace pcl_conversions
#endif
