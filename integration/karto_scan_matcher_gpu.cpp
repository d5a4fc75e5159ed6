// karto_scan_matcher_gpu.cpp -- the reference's OWN karto::Mapper driving the MI355X scan matcher.
//
// This translation unit DEFINES karto::ScanMatcher::MatchScan (Mapper.h:1155-1159; the reference's body is
// Mapper.cpp:184-291) on top of the C ABI of include/lslam_gpu.h.  Linked into open_karto in place of the reference's
// definition, every caller of MatchScan inside the unmodified library -- Mapper::Process (Mapper.cpp:2040),
// MapperGraph::AddEdges (:942), LinkNearChains (:1140), TryCloseLoop (:991 coarse on the loop matcher, :1015 fine) --
// runs its correlative search on the GPU while the pose graph, the scan manager and the dataset stay the
// reference's own code.  Nothing else of ScanMatcher is replaced: Create (Mapper.cpp:126-172) still allocates the
// host-side CorrelationGrid / search-space grid, which this file only reads for their GEOMETRY.
//
// How a maintainer of the reference uses it:
//   * either delete the body of ScanMatcher::MatchScan from Mapper.cpp and add this file to the library's sources,
//   * or, without touching any reference file, mark the reference's definition weak in the compiled object
//     (objcopy --weaken-symbol=<mangled MatchScan> Mapper.o) and link this file's strong definition next to it:
//     that is what oracle/Makefile's `ref_gpu` target does for the parity tests (link-time substitution).
//
// MatchScan is a member of ScanMatcher, and ScanMatcher is a friend of Mapper (Mapper.h:1745), so the nine Mapper
// parameters are read exactly where the reference reads them (Mapper.cpp:206,238-256,279-280,405-411); no access
// hack is needed.  The karto::LaserRangeFinder -> lslam_laser conversion the C ABI asks for is laser_from() below
// (what karto_slam.cc:384-395 configures per sensor).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "open_karto/Mapper.h"

#include "karto_occupancy_grid_gpu.hpp"  // lslam::LaserFrom (karto::LaserRangeFinder -> lslam_laser); includes lslam_gpu.h

namespace lslam_karto {

// one HIP context per process, on device $LSLAM_DEVICE (default 0)
static lslam_context* context() {
  static lslam_context* ctx = nullptr;
  if (!ctx) {
    const char* e = std::getenv("LSLAM_DEVICE");
    int rc = lslam_create(e ? std::atoi(e) : 0, &ctx);
    if (rc != LSLAM_OK) throw std::runtime_error(std::string("lslam_create: ") + lslam_last_error(nullptr));
  }
  return ctx;
}

static lslam_laser laser_from(karto::LaserRangeFinder* lrf) { return lslam::LaserFrom(lrf); }

struct GpuMatcher {
  lslam_matcher* h = nullptr;
  lslam_matcher_config cfg;
  lslam_laser laser;
  int grid_w = 0, grid_h = 0;  // geometry of the karto::CorrelationGrid it mirrors
  std::vector<double> ranges, poses;
  long long calls = 0;
};

// keyed by the karto::ScanMatcher instance (a library that adopts this file would hold the handle as a member)
static std::map<const karto::ScanMatcher*, GpuMatcher>& registry() {
  static std::map<const karto::ScanMatcher*, GpuMatcher> r;
  return r;
}

static bool same_laser(const lslam_laser& a, const lslam_laser& b) { return std::memcmp(&a, &b, sizeof a) == 0; }
static bool same_cfg(const lslam_matcher_config& a, const lslam_matcher_config& b) {
  return std::memcmp(&a, &b, sizeof a) == 0;
}

long long gpu_match_calls() {  // for the tests: how many MatchScan calls really ran on the device
  long long n = 0;
  for (auto& kv : registry()) n += kv.second.calls;
  return n;
}

void release_all() {
  for (auto& kv : registry())
    if (kv.second.h) lslam_matcher_destroy(kv.second.h);
  registry().clear();
}

}  // namespace lslam_karto

extern "C" long long lslam_karto_gpu_match_calls(void) { return lslam_karto::gpu_match_calls(); }
extern "C" void lslam_karto_gpu_release(void) { lslam_karto::release_all(); }

namespace karto {

static const kt_double kMaxVariance = 500.0;  // MAX_VARIANCE, file-local in the reference (Mapper.cpp:35)

kt_double ScanMatcher::MatchScan(LocalizedRangeScan* pScan, const LocalizedRangeScanVector& rBaseScans, Pose2& rMean,
                                 Matrix3& rCovariance, kt_bool doPenalize, kt_bool doRefineMatch) {
  using namespace lslam_karto;
  // Mapper.cpp:195-209: a scan without readings cannot be matched
  Pose2 scanPose = pScan->GetSensorPose();
  if (pScan->GetNumberOfRangeReadings() == 0) {
    rMean = scanPose;
    rCovariance(0, 0) = kMaxVariance;
    rCovariance(1, 1) = kMaxVariance;
    rCovariance(2, 2) = 4 * math::Square(m_pMapper->m_pCoarseAngleResolution->GetValue());
    return 0.0;
  }

  // ---- which parameter set built this matcher (Mapper.cpp:1964-1968 sequential, :865-867 loop) ----------------------
  const kt_double resolution = m_pCorrelationGrid->GetResolution();
  const kt_int32s side = m_pSearchSpaceProbs->GetWidth();
  lslam_matcher_config cfg;
  lslam_matcher_config_defaults(&cfg);
  struct Set { kt_double dim, res, smear; };
  const Set sets[2] = {{m_pMapper->m_pCorrelationSearchSpaceDimension->GetValue(),
                        m_pMapper->m_pCorrelationSearchSpaceResolution->GetValue(),
                        m_pMapper->m_pCorrelationSearchSpaceSmearDeviation->GetValue()},
                       {m_pMapper->m_pLoopSearchSpaceDimension->GetValue(), m_pMapper->m_pLoopSearchSpaceResolution->GetValue(),
                        m_pMapper->m_pLoopSearchSpaceSmearDeviation->GetValue()}};
  int pick = -1;
  const int first = (this == m_pMapper->m_pSequentialScanMatcher) ? 0 : 1;
  for (int t = 0; t < 2 && pick < 0; t++) {
    const Set& s = sets[(first + t) & 1];
    if (s.res == resolution && static_cast<kt_int32s>(math::Round(s.dim / s.res) + 1) == side) pick = (first + t) & 1;
  }
  if (pick < 0) throw std::runtime_error("lslam: ScanMatcher was not created from the Mapper's search-space parameters");
  cfg.search_size = sets[pick].dim;
  cfg.resolution = sets[pick].res;
  cfg.smear_deviation = sets[pick].smear;
  LaserRangeFinder* lrf = pScan->GetLaserRangeFinder();
  const lslam_laser laser = laser_from(lrf);
  cfg.range_threshold = laser.range_threshold;
  cfg.coarse_search_angle_offset = m_pMapper->m_pCoarseSearchAngleOffset->GetValue();
  cfg.coarse_angle_resolution = m_pMapper->m_pCoarseAngleResolution->GetValue();
  cfg.fine_search_angle_offset = m_pMapper->m_pFineSearchAngleOffset->GetValue();
  cfg.distance_variance_penalty = m_pMapper->m_pDistanceVariancePenalty->GetValue();
  cfg.angle_variance_penalty = m_pMapper->m_pAngleVariancePenalty->GetValue();
  cfg.minimum_distance_penalty = m_pMapper->m_pMinimumDistancePenalty->GetValue();
  cfg.minimum_angle_penalty = m_pMapper->m_pMinimumAnglePenalty->GetValue();
  cfg.use_response_expansion = m_pMapper->m_pUseResponseExpansion->GetValue() ? 1 : 0;

  // ---- the device matcher mirroring this instance (re-created when a parameter or the sensor changed) ---------------
  GpuMatcher& g = registry()[this];
  if (!g.h || !same_cfg(g.cfg, cfg) || !same_laser(g.laser, laser) || g.grid_w != m_pCorrelationGrid->GetWidth() ||
      g.grid_h != m_pCorrelationGrid->GetHeight()) {
    if (g.h) lslam_matcher_destroy(g.h);
    g.h = nullptr;
    int rc = lslam_matcher_create(context(), &cfg, &laser, &g.h);
    if (rc != LSLAM_OK) throw std::runtime_error(std::string("lslam_matcher_create: ") + lslam_last_error(context()));
    int32_t info[8];
    double off[2];
    lslam_matcher_grid_info(g.h, info, off);
    // same correlation grid as ScanMatcher::Create built on the host (Mapper.cpp:147-171): the range threshold the
    // reference passed to Create is the laser's (Mapper.cpp:2011), so the two must agree
    if (info[0] != m_pCorrelationGrid->GetWidth() || info[1] != m_pCorrelationGrid->GetHeight() ||
        info[2] != m_pCorrelationGrid->GetWidthStep()) {
      lslam_matcher_destroy(g.h);
      g.h = nullptr;
      throw std::runtime_error("lslam: device correlation grid geometry differs from karto::CorrelationGrid");
    }
    g.cfg = cfg;
    g.laser = laser;
    g.grid_w = info[0];
    g.grid_h = info[1];
  }

  // ---- LocalizedRangeScan* -> what the C ABI takes: raw readings + SENSOR poses (Karto.h:5020,5280) ------------------
  const int nb = lslam_matcher_num_beams(g.h);
  const size_t stride = static_cast<size_t>(nb > 0 ? nb : 1);
  const size_t n_base = rBaseScans.size();
  g.ranges.resize(n_base * stride);
  g.poses.resize(n_base * 3);
  for (size_t i = 0; i < n_base; i++) {
    LocalizedRangeScan* b = rBaseScans[i];
    if (b->GetLaserRangeFinder() != lrf && !same_laser(laser_from(b->GetLaserRangeFinder()), laser))
      throw std::runtime_error("lslam: base scans of a different LaserRangeFinder are not supported");
    if (static_cast<int>(b->GetNumberOfRangeReadings()) < nb) throw std::runtime_error("lslam: base scan has too few readings");
    std::memcpy(&g.ranges[i * stride], b->GetRangeReadings(), sizeof(double) * static_cast<size_t>(nb));
    const Pose2 sp = b->GetSensorPose();
    g.poses[3 * i] = sp.GetX();
    g.poses[3 * i + 1] = sp.GetY();
    g.poses[3 * i + 2] = sp.GetHeading();
  }
  if (static_cast<int>(pScan->GetNumberOfRangeReadings()) < nb) throw std::runtime_error("lslam: scan has too few readings");
  const double q[3] = {scanPose.GetX(), scanPose.GetY(), scanPose.GetHeading()};
  lslam_match_result r;
  int rc = lslam_matcher_match_scan(g.h, static_cast<int>(n_base), g.ranges.data(), static_cast<int>(stride), g.poses.data(),
                                    pScan->GetRangeReadings(), q, doPenalize ? 1 : 0, doRefineMatch ? 1 : 0, &r);
  if (rc != LSLAM_OK) throw std::runtime_error(std::string("lslam_matcher_match_scan: ") + lslam_last_error(context()));
  g.calls++;
  // the reference throws from inside CorrelateScan (Mapper.cpp:444-447, 484-487)
  if (r.status == LSLAM_ERR_PROBABILITY_SEARCH)
    throw std::runtime_error("Mapper FATAL ERROR - Index out of range in probability search!");
  if (r.status == LSLAM_ERR_NO_BEST_POSE) throw std::runtime_error("Mapper FATAL ERROR - Unable to find best position");
  if (r.status != LSLAM_OK) throw std::runtime_error("lslam: scan match failed");
  rMean = Pose2(r.pose[0], r.pose[1], r.pose[2]);
  // the entries CorrelateScan writes (Mapper.cpp:535-692); the others keep what the caller put there
  rCovariance(0, 0) = r.covariance[0];
  rCovariance(0, 1) = r.covariance[1];
  rCovariance(1, 0) = r.covariance[3];
  rCovariance(1, 1) = r.covariance[4];
  rCovariance(2, 2) = r.covariance[8];
  return r.response;
}

}  // namespace karto
