// lslam_adapters.hpp -- header-only C++ host layer above the C ABI (include/lslam_gpu.h).
//
// The reference's seams for this path are C++ classes, not an FFI; these two
// adapters mirror their names, argument meaning and error behaviour so a maintainer of the
// reference can swap call sites one for one:
//   lslam::GpuScanMatcher  <-> karto::ScanMatcher                 (Mapper.h:1127-1279)
//   lslam::MapRepGpu       <-> hectorslam::MapRepresentationInterface (H/slam_main/
//                              MapRepresentationInterface.h:44-69), update side
// Only PODs appear here so the header builds without open_karto / Eigen; INTEGRATION.md shows the
// few lines that convert karto::LocalizedRangeScan / hectorslam::DataContainer to these PODs.
#pragma once

#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "lslam_gpu.h"

namespace lslam {

struct Pose2 {  // karto::Pose2 (Karto.h:1959-2168): x, y, heading
  double x = 0, y = 0, heading = 0;
};
struct Matrix3 {  // karto::Matrix3 (Karto.h:2344-2613), row-major
  double m[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  double& operator()(int r, int c) { return m[r][c]; }
  double operator()(int r, int c) const { return m[r][c]; }
};
// what MatchScan reads from a karto::LocalizedRangeScan: GetRangeReadings() + GetSensorPose()
struct RangeScan {
  const double* ranges = nullptr;  // >= num_beams readings
  Pose2 sensor_pose;
};

// the exceptions the reference throws on this path (Mapper.cpp:444-447,484-487; Karto.h:4488-4499)
struct MatcherError : std::runtime_error {
  int status;
  MatcherError(int s, const std::string& w) : std::runtime_error(w), status(s) {}
};

class GpuScanMatcher {
 public:
  // ScanMatcher::Create (Mapper.h:1139-1143): nullptr on invalid parameters, throws
  // std::runtime_error for a bad smear deviation (Mapper.h:1041-1053)
  static GpuScanMatcher* Create(lslam_context* ctx, const lslam_matcher_config& cfg, const lslam_laser& laser) {
    lslam_matcher* h = nullptr;
    int rc = lslam_matcher_create(ctx, &cfg, &laser, &h);
    if (rc == LSLAM_ERR_INVALID_ARGUMENT) return nullptr;
    if (rc != LSLAM_OK) throw MatcherError(rc, lslam_last_error(ctx));
    return new GpuScanMatcher(ctx, h, laser);
  }
  ~GpuScanMatcher() { lslam_matcher_destroy(h_); }
  GpuScanMatcher(const GpuScanMatcher&) = delete;
  GpuScanMatcher& operator=(const GpuScanMatcher&) = delete;

  // kt_double MatchScan(pScan, rBaseScans, rMean, rCovariance, doPenalize, doRefineMatch)
  // (Mapper.h:1155-1159)
  double MatchScan(const RangeScan& scan, const std::vector<RangeScan>& baseScans, Pose2& rMean,
                   Matrix3& rCovariance, bool doPenalize = true, bool doRefineMatch = true) {
    const int n = lslam_matcher_num_beams(h_);
    const int stride = n > 0 ? n : 1;
    ranges_.resize(baseScans.size() * (size_t)stride);
    poses_.resize(baseScans.size() * 3);
    for (size_t i = 0; i < baseScans.size(); i++) {
      std::memcpy(&ranges_[i * stride], baseScans[i].ranges, sizeof(double) * (size_t)n);
      poses_[3 * i] = baseScans[i].sensor_pose.x;
      poses_[3 * i + 1] = baseScans[i].sensor_pose.y;
      poses_[3 * i + 2] = baseScans[i].sensor_pose.heading;
    }
    const double q[3] = {scan.sensor_pose.x, scan.sensor_pose.y, scan.sensor_pose.heading};
    lslam_match_result r;
    check(lslam_matcher_match_scan(h_, (int)baseScans.size(), ranges_.data(), stride, poses_.data(), scan.ranges, q,
                                   doPenalize, doRefineMatch, &r));
    return unpack(r, rMean, rCovariance);
  }

  // AddScans (Mapper.cpp:699-708) around an explicit centre + the search against the current
  // grid for many independent scans (the batched mode; no equivalent single call in the reference)
  void AddScans(const std::vector<RangeScan>& baseScans, const Pose2& center) {
    const int n = lslam_matcher_num_beams(h_);
    const int stride = n > 0 ? n : 1;
    ranges_.resize(baseScans.size() * (size_t)stride);
    poses_.resize(baseScans.size() * 3);
    for (size_t i = 0; i < baseScans.size(); i++) {
      std::memcpy(&ranges_[i * stride], baseScans[i].ranges, sizeof(double) * (size_t)n);
      poses_[3 * i] = baseScans[i].sensor_pose.x;
      poses_[3 * i + 1] = baseScans[i].sensor_pose.y;
      poses_[3 * i + 2] = baseScans[i].sensor_pose.heading;
    }
    const double c[3] = {center.x, center.y, center.heading};
    check(lslam_matcher_set_base_scans(h_, (int)baseScans.size(), ranges_.data(), stride, poses_.data(), c));
  }
  std::vector<lslam_match_result> MatchBatch(const std::vector<RangeScan>& scans, bool doPenalize = true,
                                             bool doRefineMatch = true) {
    const int n = lslam_matcher_num_beams(h_);
    const int stride = n > 0 ? n : 1;
    ranges_.resize(scans.size() * (size_t)stride);
    poses_.resize(scans.size() * 3);
    for (size_t i = 0; i < scans.size(); i++) {
      std::memcpy(&ranges_[i * stride], scans[i].ranges, sizeof(double) * (size_t)n);
      poses_[3 * i] = scans[i].sensor_pose.x;
      poses_[3 * i + 1] = scans[i].sensor_pose.y;
      poses_[3 * i + 2] = scans[i].sensor_pose.heading;
    }
    std::vector<lslam_match_result> out(scans.size());
    check(lslam_matcher_match_batch(h_, (int)scans.size(), ranges_.data(), stride, poses_.data(), doPenalize,
                                    doRefineMatch, out.data()));
    return out;
  }

  // MatchBatch as `depth` pipelined sub-batches (LSLAM_OPT_PIPELINE_DEPTH, 1..4; lslam_gpu.h): the upload of one
  // runs under the kernels of the other, their response kernels fill each other's tails.  Same records.  No reference counterpart.
  void SetPipelineDepth(int depth) { check(lslam_matcher_set_option(h_, LSLAM_OPT_PIPELINE_DEPTH, depth)); }

  // LocalizedRangeScan::GetSensorAt / SetSensorPose (Karto.h:5280-5313)
  Pose2 SensorPoseFromRobot(const Pose2& robot) const {
    const double r[3] = {robot.x, robot.y, robot.heading};
    double s[3];
    lslam_sensor_pose_from_robot(&laser_, r, s);
    return Pose2{s[0], s[1], s[2]};
  }
  Pose2 RobotPoseFromSensor(const Pose2& sensor) const {
    const double s[3] = {sensor.x, sensor.y, sensor.heading};
    double r[3];
    lslam_robot_pose_from_sensor(&laser_, s, r);
    return Pose2{r[0], r[1], r[2]};
  }
  lslam_matcher* handle() { return h_; }

 private:
  GpuScanMatcher(lslam_context* ctx, lslam_matcher* h, const lslam_laser& laser) : ctx_(ctx), h_(h), laser_(laser) {}
  void check(int rc) {
    if (rc != LSLAM_OK) throw MatcherError(rc, lslam_last_error(ctx_));
  }
  static double unpack(const lslam_match_result& r, Pose2& mean, Matrix3& cov) {
    if (r.status != LSLAM_OK) throw MatcherError(r.status, "scan matcher: the reference throws here");
    mean = Pose2{r.pose[0], r.pose[1], r.pose[2]};
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) cov.m[i][j] = r.covariance[3 * i + j];
    return r.response;
  }
  lslam_context* ctx_;
  lslam_matcher* h_;
  lslam_laser laser_;
  std::vector<double> ranges_, poses_;
};

// Pose path of karto::Mapper::Process (Mapper.cpp:1999-2079) with a device-resident running-scan
// window: what SlamKarto::addScan calls per LaserScan (karto_slam.cc:444).  The pose graph
// (vertices, edges, loop closure) stays with the reference's host code.
class GpuFrontEnd {
 public:
  // scanBufferSize / scanBufferMaximumScanDistance / minimumTravelDistance / minimumTravelHeading:
  // the Mapper parameters of the same names (Mapper.cpp:1480-1515)
  GpuFrontEnd(lslam_context* ctx, GpuScanMatcher& matcher, int scanBufferSize, double scanBufferMaximumScanDistance,
              double minimumTravelDistance, double minimumTravelHeading)
      : ctx_(ctx) {
    int rc = lslam_frontend_create(matcher.handle(), scanBufferSize, scanBufferMaximumScanDistance,
                                   minimumTravelDistance, minimumTravelHeading, &h_);
    if (rc != LSLAM_OK) throw std::runtime_error(lslam_last_error(ctx));
  }
  ~GpuFrontEnd() { lslam_frontend_destroy(h_); }
  GpuFrontEnd(const GpuFrontEnd&) = delete;
  GpuFrontEnd& operator=(const GpuFrontEnd&) = delete;

  // kt_bool Process(LocalizedRangeScan*): ranges + odometric robot pose in; returns whether the scan
  // was processed (HasMovedEnough) and its corrected robot pose (GetCorrectedPose) / covariance
  bool Process(const double* ranges, int nRanges, const Pose2& odometricPose, Pose2& correctedPose,
               Matrix3* covariance = nullptr, double* response = nullptr) {
    const double o[3] = {odometricPose.x, odometricPose.y, odometricPose.heading};
    double c[3], cov[9];
    int processed = 0;
    int rc = lslam_frontend_process(h_, ranges, nRanges, o, &processed, c, cov, response);
    if (rc != LSLAM_OK) throw MatcherError(rc, lslam_last_error(ctx_));
    correctedPose = Pose2{c[0], c[1], c[2]};
    if (covariance && processed)
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) covariance->m[i][j] = cov[3 * i + j];
    return processed != 0;
  }
  // Process for a recorded trajectory (nScans scans already at hand, rows of rangesStride doubles, poses as x, y, heading
  // triples): one scan of look-ahead, same results (lslam_frontend_process_many).  processed[i] / corrected[3 i ..] as Process.
  void ProcessMany(const double* ranges, int rangesStride, const double* odometricPoses, int nScans, std::vector<int32_t>& processed,
                   std::vector<double>& corrected) {
    processed.assign((size_t)nScans, 0);
    corrected.assign((size_t)nScans * 3, 0.0);
    int rc = lslam_frontend_process_many(h_, nScans, ranges, rangesStride, odometricPoses, nullptr, processed.data(), corrected.data(),
                                         nullptr, nullptr);
    if (rc != LSLAM_OK) throw MatcherError(rc, lslam_last_error(ctx_));
  }
  int RunningScans() const { return lslam_frontend_running_scans(h_); }

 private:
  lslam_context* ctx_;
  lslam_frontend* h_ = nullptr;
};

// Batched many-scan mode over every GPU of the node (one process): scans sharded [r*B/W, (r+1)*B/W), shared grid
// replicated over xGMI, results in scan order
class GpuMatcherPool {
 public:
  GpuMatcherPool(int nDevices, const lslam_matcher_config& cfg, const lslam_laser& laser) {
    int rc = lslam_pool_create(nDevices, &cfg, &laser, &h_);
    if (rc != LSLAM_OK) throw std::runtime_error(lslam_pool_last_error(nullptr));
  }
  GpuMatcherPool(const std::vector<int>& devices, const lslam_matcher_config& cfg, const lslam_laser& laser) {
    int rc = lslam_pool_create_on(devices.data(), (int)devices.size(), &cfg, &laser, &h_);
    if (rc != LSLAM_OK) throw std::runtime_error(lslam_pool_last_error(nullptr));
  }
  ~GpuMatcherPool() { lslam_pool_destroy(h_); }
  GpuMatcherPool(const GpuMatcherPool&) = delete;
  GpuMatcherPool& operator=(const GpuMatcherPool&) = delete;
  int devices() const { return lslam_pool_devices(h_); }
  // ScanMatcher::AddScans once; the grid travels device-to-device
  void AddScans(const std::vector<RangeScan>& baseScans, int numBeams, const Pose2& center, bool rebuildEverywhere = false) {
    std::vector<double> r(baseScans.size() * (size_t)numBeams), p(baseScans.size() * 3);
    for (size_t i = 0; i < baseScans.size(); i++) {
      std::copy(baseScans[i].ranges, baseScans[i].ranges + numBeams, r.begin() + i * (size_t)numBeams);
      p[3 * i] = baseScans[i].sensor_pose.x; p[3 * i + 1] = baseScans[i].sensor_pose.y; p[3 * i + 2] = baseScans[i].sensor_pose.heading;
    }
    const double c[3] = {center.x, center.y, center.heading};
    int rc = lslam_pool_set_base_scans(h_, (int)baseScans.size(), r.data(), numBeams, p.data(), c, rebuildEverywhere ? 1 : 0);
    if (rc != LSLAM_OK) throw MatcherError(rc, lslam_pool_last_error(h_));
  }
  // the search part of MatchScan for n independent scans (ranges: n rows of numBeams doubles, poses: n*3)
  std::vector<lslam_match_result> MatchBatch(const double* ranges, int numBeams, const double* sensorPoses, int n,
                                             bool doPenalize = true, bool doRefineMatch = true) {
    std::vector<lslam_match_result> out((size_t)n);
    int rc = lslam_pool_match_batch(h_, n, ranges, numBeams, sensorPoses, doPenalize, doRefineMatch, out.data());
    if (rc != LSLAM_OK) throw MatcherError(rc, lslam_pool_last_error(h_));
    return out;
  }

 private:
  lslam_pool* h_ = nullptr;
};

// hectorslam::MapRepresentationInterface on the GPU: matchData + updateByScan
class MapRepGpu {
 public:
  // MapRepMultiMap(mapResolution, mapSizeX, mapSizeY, numDepth, startCoords)
  // (H/slam_main/MapRepMultiMap.h:57-93): offset = totalMapSize * startCoords
  MapRepGpu(lslam_context* ctx, float mapResolution, int mapSizeX, int mapSizeY, unsigned numDepth, float startX,
            float startY)
      : ctx_(ctx) {
    float offX = (mapResolution * static_cast<float>(mapSizeX)) * startX;
    float offY = (mapResolution * static_cast<float>(mapSizeY)) * startY;
    int rc = lslam_map_create(ctx, mapSizeX, mapSizeY, mapResolution, offX, offY, (int)numDepth, &h_);
    if (rc != LSLAM_OK) throw std::runtime_error(lslam_last_error(ctx));
  }
  ~MapRepGpu() { lslam_map_destroy(h_); }
  MapRepGpu(const MapRepGpu&) = delete;
  MapRepGpu& operator=(const MapRepGpu&) = delete;

  void reset() { lslam_map_reset(h_); }
  float getScaleToMap() const { return lslam_map_scale_to_map(h_, 0); }
  int getMapLevels() const { return lslam_map_levels(h_); }
  void setUpdateFactorFree(float f) { lslam_map_set_update_factor_free(h_, f); }
  void setUpdateFactorOccupied(float f) { lslam_map_set_update_factor_occupied(h_, f); }
  // updateByScan(dataContainer, robotPoseWorld): points = DataContainer entries (x,y pairs, map-cell
  // units), origo = DataContainer::getOrigo()
  void updateByScan(const float* pointsXY, int n, const float origo[2], const float robotPoseWorld[3]) {
    int rc = lslam_map_update_by_scan(h_, pointsXY, n, origo, robotPoseWorld);
    if (rc != LSLAM_OK) throw std::runtime_error(lslam_last_error(ctx_));
  }
  // Eigen::Vector3f matchData(beginEstimateWorld, dataContainer, covMatrix)
  // (the container is cached for the levels above 0 of the next updateByScan, like MapRepMultiMap.h:161)
  void matchData(const float beginEstimateWorld[3], const float* pointsXY, int n, const float origo[2], float outPose[3],
                 float outCov[9]) {
    int rc = lslam_map_match_data(h_, pointsXY, n, origo, beginEstimateWorld, outPose, outCov);
    if (rc != LSLAM_OK) throw std::runtime_error(lslam_last_error(ctx_));
  }
  // getGridMap(level) contents: log-odds plane / the int8 data of nav_msgs::OccupancyGrid
  void readLogOdds(int level, float* out) { lslam_map_read_logodds(h_, level, out); }
  void readOccupancy(int level, int8_t* out) { lslam_map_read_occupancy_i8(h_, level, out); }
  lslam_map* handle() { return h_; }

 private:
  lslam_context* ctx_;
  lslam_map* h_ = nullptr;
};

}  // namespace lslam
