// karto_occupancy_grid_gpu.hpp -- karto::OccupancyGrid::CreateFromScans (Karto.h:5659-5673) on the MI355X path, with the
// reference's OWN types on both sides: a `LocalizedRangeScanVector` in, a heap `karto::OccupancyGrid*` out (the caller
// deletes it, like karto_slam.cc:578).  This is seam B2: SlamKarto::updateMap (karto_slam.cc:507-581) rebuilds its published
// map from ALL processed scans every map_update_interval with exactly this call; replacing it is one line there
// (INTEGRATION.md §2d):
//     karto::OccupancyGrid* occ_grid = lslam::CreateOccupancyGridFromScans(ctx, mapper_->GetAllProcessedScans(), resolution_);
// The reference's function is a static member DEFINED IN THE HEADER (inline), so it cannot be substituted at link time
// the way ScanMatcher::MatchScan is; it is a call-site change instead.
//
// The returned grid holds the cell states (GridStates: 0 unknown, 100 occupied, 255 free), width, height and the
// CoordinateConverter (offset, scale) the reference computes; its hit / pass COUNTER grids stay empty -- they live in HBM
// and nothing in the reference reads them after the build (IsFree, GetValue, Clone work as usual).
#pragma once

#include <cstring>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "open_karto/Karto.h"

#include "lslam_gpu.h"

namespace lslam {

// karto::LaserRangeFinder -> lslam_laser (Karto.h:3985-4137: the getters behind karto_slam.cc:384-395)
inline lslam_laser LaserFrom(karto::LaserRangeFinder* lrf) {
  lslam_laser l;
  l.minimum_angle = lrf->GetMinimumAngle();
  l.maximum_angle = lrf->GetMaximumAngle();
  l.angular_resolution = lrf->GetAngularResolution();
  l.minimum_range = lrf->GetMinimumRange();
  l.maximum_range = lrf->GetMaximumRange();
  l.range_threshold = lrf->GetRangeThreshold();
  const karto::Pose2 off = lrf->GetOffsetPose();
  l.offset_x = off.GetX();
  l.offset_y = off.GetY();
  l.offset_heading = off.GetHeading();
  return l;
}

inline karto::OccupancyGrid* CreateOccupancyGridFromScans(lslam_context* ctx, const karto::LocalizedRangeScanVector& rScans,
                                                          kt_double resolution) {
  if (rScans.empty()) return NULL;  // Karto.h:5661-5664
  if (karto::math::DoubleEqual(resolution, 0.0)) throw karto::Exception("Resolution cannot be 0");  // the ctor's check (:5627-5630)
  karto::LaserRangeFinder* lrf = rScans[0]->GetLaserRangeFinder();
  const lslam_laser laser = LaserFrom(lrf);
  const size_t n = lrf->GetNumberOfRangeReadings();  // Karto.h:4152-4161: round((max - min) / res)
  const size_t stride = n > 0 ? n : 1;
  std::vector<double> ranges(rScans.size() * stride), poses(rScans.size() * 3);
  for (size_t i = 0; i < rScans.size(); i++) {
    karto::LocalizedRangeScan* s = rScans[i];
    if (s->GetLaserRangeFinder() != lrf) {
      const lslam_laser other = LaserFrom(s->GetLaserRangeFinder());
      if (std::memcmp(&other, &laser, sizeof laser) != 0) throw std::runtime_error("lslam: scans of different LaserRangeFinders");
    }
    if (s->GetNumberOfRangeReadings() < n) throw std::runtime_error("lslam: scan has too few readings");
    std::memcpy(&ranges[i * stride], s->GetRangeReadings(), sizeof(double) * n);
    const karto::Pose2 sp = s->GetSensorPose();
    poses[3 * i] = sp.GetX();
    poses[3 * i + 1] = sp.GetY();
    poses[3 * i + 2] = sp.GetHeading();
  }
  lslam_occgrid* og = nullptr;
  int rc = lslam_occgrid_create_from_scans(ctx, &laser, static_cast<int>(rScans.size()), ranges.data(), static_cast<int>(stride),
                                           poses.data(), resolution, &og);
  if (rc != LSLAM_OK) throw std::runtime_error(std::string("lslam_occgrid_create_from_scans: ") + lslam_last_error(ctx));
  int32_t dims[2];
  double off[2], res = 0.0;
  lslam_occgrid_info(og, dims, off, &res);
  std::vector<uint8_t> cells(static_cast<size_t>(dims[0]) * dims[1]);
  rc = lslam_occgrid_read_u8(og, cells.data());
  lslam_occgrid_destroy(og);
  if (rc != LSLAM_OK) throw std::runtime_error(std::string("lslam_occgrid_read_u8: ") + lslam_last_error(ctx));
  karto::OccupancyGrid* grid = new karto::OccupancyGrid(dims[0], dims[1], karto::Vector2<kt_double>(off[0], off[1]), resolution);
  kt_int8u* data = grid->GetDataPointer();
  const size_t step = static_cast<size_t>(grid->GetWidthStep());
  for (int y = 0; y < dims[1]; y++) std::memcpy(data + y * step, &cells[static_cast<size_t>(y) * dims[0]], static_cast<size_t>(dims[0]));
  return grid;
}

}  // namespace lslam
