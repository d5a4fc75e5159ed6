// hector_map_rep_gpu.hpp -- hectorslam::MapRepresentationInterface implemented on the MI355X path.
//
// lesson4's HectorSlamProcessor holds a plain `MapRepresentationInterface* mapRep` (H/slam_main/
// HectorSlamProcessor.h:141) and drives it through matchData / updateByScan / onMapUpdated (:91,:104-105).  This class
// is that interface (H/slam_main/MapRepresentationInterface.h:44-69) on top of the C ABI of include/lslam_gpu.h, with
// the constructor signature of the reference's own implementation, MapRepMultiMap (H/slam_main/MapRepMultiMap.h:57-59):
// replacing `new MapRepMultiMap(...)` at HectorSlamProcessor.h:61 by `new lslam::HectorMapRepGpu(ctx, ...)` is the
// whole integration.  Compiles against the reference's headers (and therefore Eigen); it is
// header-only like everything under hector_mapping/.
//
//   matchData      -> lslam_map_match_data   (Gauss-Newton on the pyramid, k_gn_match; caches the container like :161)
//   updateByScan   -> lslam_map_update_by_scan (Bresenham log-odds update of every level, levels > 0 from the cached
//                     container exactly like :186)
//   getGridMap(l)  -> a HOST mirror hectorslam::GridMap per level, refreshed from HBM on demand (the publisher thread of
//                     hector_slam.cc:263-311 reads cells through isFree / isOccupied; LogOddsCell::updateIndex is a
//                     per-scan scratch mark of the CPU algorithm and is not mirrored)
//
// Threading (hector_slam.cc:263-311): the node's publisher thread calls getGridMap(i) and THEN takes the level's
// MapLockerInterface to read the cells, while the main thread matches and updates.  An lslam_map is not thread-safe (its
// pipelined update keeps a pending apply that every reader flushes), so every call into it -- and the stale_ flags -- sit
// behind ONE internal mutex; the rewrite of a host mirror additionally holds that level's MapLockerInterface, the lock the
// publisher holds while it reads the mirror (MapProcContainer::updateByScan takes the same lock around the CPU update,
// H/slam_main/MapProcContainer.h:80-91).  getGridMap is called before the publisher locks (hector_slam.cc:263), so taking
// the non-recursive lock inside it cannot deadlock in the reference's flow.
#pragma once

#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "lesson4/hector_mapping/slam_main/MapRepMultiMap.h"

#include "lslam_gpu.h"

namespace lslam {

class HectorMapRepGpu : public hectorslam::MapRepresentationInterface {
 public:
  // same arguments as MapRepMultiMap (H/slam_main/MapRepMultiMap.h:57-93) + the device context
  HectorMapRepGpu(lslam_context* ctx, float mapResolution, int mapSizeX, int mapSizeY, unsigned int numDepth,
                  const Eigen::Vector2f& startCoords)
      : ctx_(ctx) {
    const float totalMapSizeX = mapResolution * static_cast<float>(mapSizeX);
    const float mid_offset_x = totalMapSizeX * startCoords.x();
    const float totalMapSizeY = mapResolution * static_cast<float>(mapSizeY);
    const float mid_offset_y = totalMapSizeY * startCoords.y();
    check(lslam_map_create(ctx, mapSizeX, mapSizeY, mapResolution, mid_offset_x, mid_offset_y, static_cast<int>(numDepth),
                           &map_));
    Eigen::Vector2i resolution(mapSizeX, mapSizeY);
    for (unsigned int i = 0; i < numDepth; ++i) {
      mirrors_.push_back(new hectorslam::GridMap(mapResolution, resolution, Eigen::Vector2f(mid_offset_x, mid_offset_y)));
      stale_.push_back(false);
      mutexes_.push_back(0);
      resolution /= 2;
      mapResolution *= 2.0f;
    }
  }
  virtual ~HectorMapRepGpu() {
    for (size_t i = 0; i < mirrors_.size(); ++i) {
      delete mirrors_[i];
      if (mutexes_[i]) delete mutexes_[i];  // MapProcContainer::cleanup owns its mutex the same way (:62-65)
    }
    lslam_map_destroy(map_);
  }
  HectorMapRepGpu(const HectorMapRepGpu&) = delete;
  HectorMapRepGpu& operator=(const HectorMapRepGpu&) = delete;

  virtual void reset() {
    std::lock_guard<std::mutex> lock(mu_);
    check(lslam_map_reset(map_));
    for (size_t i = 0; i < mirrors_.size(); ++i) {
      mirrors_[i]->reset();
      stale_[i] = false;
    }
  }
  virtual float getScaleToMap() const { return lslam_map_scale_to_map(map_, 0); }
  virtual int getMapLevels() const { return static_cast<int>(mirrors_.size()); }
  virtual const hectorslam::GridMap& getGridMap(int mapLevel = 0) const {
    std::lock_guard<std::mutex> lock(mu_);
    if (stale_[mapLevel]) {
      hectorslam::GridMap& g = *mirrors_[mapLevel];
      const int n = g.getSizeX() * g.getSizeY();
      scratch_.resize(static_cast<size_t>(n));
      check(lslam_map_read_logodds(map_, mapLevel, scratch_.data()));
      MapLockerInterface* cells = mutexes_[mapLevel];  // a reader of the mirror holds this one
      if (cells) cells->lockMap();
      for (int i = 0; i < n; ++i) g.getCell(i).logOddsVal = scratch_[static_cast<size_t>(i)];
      if (cells) cells->unlockMap();
      stale_[mapLevel] = false;
    }
    return *mirrors_[mapLevel];
  }
  virtual void addMapMutex(int i, MapLockerInterface* mapMutex) {
    std::lock_guard<std::mutex> lock(mu_);
    if (mutexes_[i]) delete mutexes_[i];
    mutexes_[i] = mapMutex;
  }
  virtual MapLockerInterface* getMapMutex(int i) { return mutexes_[i]; }
  virtual void onMapUpdated() {}  // the device matcher keeps no interpolation cache (MapRepMultiMap.h:127-135 resets the CPU one)

  virtual Eigen::Vector3f matchData(const Eigen::Vector3f& beginEstimateWorld, const hectorslam::DataContainer& dataContainer,
                                    Eigen::Matrix3f& covMatrix) {
    std::lock_guard<std::mutex> lock(mu_);
    flatten(dataContainer);
    const float begin[3] = {beginEstimateWorld[0], beginEstimateWorld[1], beginEstimateWorld[2]};
    float pose[3], cov[9];
    check(lslam_map_match_data(map_, pts_.data(), dataContainer.getSize(), origo_, begin, pose, cov));
    if (dataContainer.getSize() != 0)  // ScanMatcher.h:96: an empty container leaves covMatrix untouched
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) covMatrix(i, j) = cov[3 * i + j];
    return Eigen::Vector3f(pose[0], pose[1], pose[2]);
  }
  virtual void updateByScan(const hectorslam::DataContainer& dataContainer, const Eigen::Vector3f& robotPoseWorld) {
    std::lock_guard<std::mutex> lock(mu_);
    flatten(dataContainer);
    const float pose[3] = {robotPoseWorld[0], robotPoseWorld[1], robotPoseWorld[2]};
    check(lslam_map_update_by_scan(map_, pts_.data(), dataContainer.getSize(), origo_, pose));
    for (size_t i = 0; i < mirrors_.size(); ++i) {
      stale_[i] = true;
      mirrors_[i]->setUpdated();  // GridMapBase::setUpdated (H/map/GridMapBase.h:333): drives the node's re-publish
    }
    ++updates_;
  }
  // number of updateByScan calls so far (what GridMapBase::getUpdateIndex tells a caller of the CPU map, without pulling
  // the map out of HBM the way getGridMap() must)
  long updateCount() const {
    std::lock_guard<std::mutex> lock(mu_);
    return updates_;
  }
  virtual void setUpdateFactorFree(float free_factor) {
    std::lock_guard<std::mutex> lock(mu_);
    check(lslam_map_set_update_factor_free(map_, free_factor));
    for (size_t i = 0; i < mirrors_.size(); ++i) mirrors_[i]->setUpdateFreeFactor(free_factor);
  }
  virtual void setUpdateFactorOccupied(float occupied_factor) {
    std::lock_guard<std::mutex> lock(mu_);
    check(lslam_map_set_update_factor_occupied(map_, occupied_factor));
    for (size_t i = 0; i < mirrors_.size(); ++i) mirrors_[i]->setUpdateOccupiedFactor(occupied_factor);
  }
  lslam_map* handle() { return map_; }

 private:
  void check(int rc) const {
    if (rc != LSLAM_OK) throw std::runtime_error(std::string("lslam: ") + lslam_last_error(ctx_));
  }
  void flatten(const hectorslam::DataContainer& dc) {
    const int n = dc.getSize();
    pts_.resize(static_cast<size_t>(2 * (n > 0 ? n : 1)));
    for (int i = 0; i < n; ++i) {
      const Eigen::Vector2f& p = dc.getVecEntry(i);
      pts_[static_cast<size_t>(2 * i)] = p[0];
      pts_[static_cast<size_t>(2 * i + 1)] = p[1];
    }
    const Eigen::Vector2f o = dc.getOrigo();
    origo_[0] = o[0];
    origo_[1] = o[1];
  }
  lslam_context* ctx_;
  mutable std::mutex mu_;  // every call into map_ and every access to stale_ / scratch_ / pts_
  lslam_map* map_ = nullptr;
  long updates_ = 0;
  std::vector<hectorslam::GridMap*> mirrors_;
  mutable std::vector<bool> stale_;
  std::vector<MapLockerInterface*> mutexes_;
  std::vector<float> pts_;
  mutable std::vector<float> scratch_;
  float origo_[2] = {0.f, 0.f};
};

}  // namespace lslam
